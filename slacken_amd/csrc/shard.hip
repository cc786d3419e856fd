// shard.hip -- the owner side of table-sharded classification (BASELINE configs[3]): point lookups of the keys
// received from the other ranks, and the scatter of the answers a rank got back to the slots its fragments read them from.
// The lookup is the left join + spanToHit's otherwise(NONE) (S/slacken/Classifier.scala:84, KeyValueIndex.scala:176-185) for
// keys that arrive without their fragment.
#include <hip/hip_runtime.h>

#include "engine.h"

namespace slk {

namespace {

constexpr int SW = 4;  // waves per block

__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 64 keys per wave iteration: lane i hashes key i; FOUR lanes then read one 64-byte bucket (4 x 16 B = one HBM line),
// sixteen buckets per wave instruction, four instructions in flight -- the access shape of the classify kernel's probe.
// A key whose home bucket is full without holding it (about 3 % at the usual load) continues alone in the next buckets.
__global__ void __launch_bounds__(SW * 64) lookup_coop_kernel(TableView T, const int64_t *__restrict__ keys, uint64_t n,
                                                              int32_t *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint4 stash_all[SW][64];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint4 *stash = stash_all[wib];
  const uint64_t nwaves = (uint64_t)gridDim.x * SW;
  const uint64_t tmask = (1ULL << T.taxon_bits) - 1;
  const int g = lane >> 2, c = lane & 3;
  const char *cellbase = (const char *)T.cells + c * 16;
  for (uint64_t base = ((uint64_t)blockIdx.x * SW + wib) * 64; base < n; base += nwaves * 64) {
    const uint64_t i = base + lane;
    const bool in = i < n;
    const uint64_t key = in ? (uint64_t)keys[i] : 0;
    const uint64_t h = fmix64(key);
    const uint64_t tag = in ? ((h & T.rem_mask) << T.disp_bits) : ~0ULL;  // displacement 0
    uint4 st;
    st.x = (uint32_t)((h >> T.shift) & T.bucket_mask);
    st.y = 0;                // taxon found
    st.z = (uint32_t)tag;
    st.w = (uint32_t)(tag >> 32);
    stash[lane] = st;
    wsync();
    ulonglong2 cell[4];
#pragma unroll
    for (int s = 0; s < 4; s++) cell[s] = *(const ulonglong2 *)(cellbase + ((uint64_t)stash[s * 16 + g].x << 6));
    uint32_t unresolved = 0;  // bit s: this lane's group of step s found neither its key nor an empty cell
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const uint64_t want = ((uint64_t)stash[s * 16 + g].w << 32) | stash[s * 16 + g].z;
      const bool act = want != ~0ULL;
      const bool e0 = cell[s].x == 0, e1 = cell[s].y == 0;
      const bool m0 = act && !e0 && (cell[s].x >> T.taxon_bits) == want;
      const bool m1 = act && !e1 && (cell[s].y >> T.taxon_bits) == want;
      if (m0 || m1) stash[s * 16 + g].y = (uint32_t)((m0 ? cell[s].x : cell[s].y) & tmask);
      const uint64_t B = __ballot(m0 || m1 || e0 || e1 || !act);
      if (((B >> (g * 4)) & 0xF) == 0) unresolved |= 1u << s;
    }
    wsync();
    int32_t taxon = (int32_t)stash[lane].y;
    // was entry `lane` unresolved?  its group was g' = lane & 15 of step s' = lane >> 4: ask lane 4 * g'
    const uint32_t ur = (uint32_t)__shfl((int)unresolved, (lane & 15) * 4);
    if (in && ((ur >> (lane >> 4)) & 1)) {
      // bucket-level linear probing continues in the next buckets (cells are never freed: stop at an empty cell)
      const uint64_t home = h >> T.shift;
      const uint64_t rem_hi = (h & T.rem_mask) << T.disp_bits;
      for (int d = 1; d <= T.max_disp; d++) {
        const ulonglong2 *b = (const ulonglong2 *)(T.cells + (((home + d) & T.bucket_mask) << 3));
        const uint64_t want = rem_hi | (uint64_t)d;
        bool has_empty = false;
        int32_t found = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          ulonglong2 v = b[q];
          has_empty |= (v.x == 0) | (v.y == 0);
          if (v.x != 0 && (v.x >> T.taxon_bits) == want) found = (int32_t)(v.x & tmask);
          if (v.y != 0 && (v.y >> T.taxon_bits) == want) found = (int32_t)(v.y & tmask);
        }
        if (found) { taxon = found; break; }
        if (has_empty) break;
      }
    }
    if (in) out[i] = ext_taxon(T, taxon);
    wsync();
  }
}

__global__ void __launch_bounds__(256) scatter_taxa_kernel(const uint64_t *__restrict__ slots, const int32_t *__restrict__ taxa,
                                                           uint64_t n, int32_t *__restrict__ by_slot,
                                                           const int32_t *__restrict__ to_dense, int32_t n_to_dense) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    int32_t t = taxa[i];
    if (to_dense != nullptr && t > 0) t = t < n_to_dense ? to_dense[t] : 0;  // (a taxon outside the taxonomy cannot be in this rank's table either)
    by_slot[slots[i]] = t;
  }
}

}  // namespace

void launch_lookup_coop(const TableView &t, const int64_t *keys, uint64_t n, int32_t *out, hipStream_t s) {
  if (n == 0) return;
  uint64_t blocks = std::min<uint64_t>((n + SW * 64 - 1) / (SW * 64), 256 * 16);
  hipLaunchKernelGGL(lookup_coop_kernel, dim3((unsigned)blocks), dim3(SW * 64), 0, s, t, keys, n, out);
}
void launch_scatter_taxa(const uint64_t *slots, const int32_t *taxa, uint64_t n, int32_t *taxa_by_slot, const int32_t *to_dense,
                         int32_t n_to_dense, hipStream_t s) {
  if (n == 0) return;
  uint64_t blocks = std::min<uint64_t>((n + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(scatter_taxa_kernel, dim3((unsigned)blocks), dim3(256), 0, s, slots, taxa, n, taxa_by_slot, to_dense, n_to_dense);
}

}  // namespace slk

// shard.hip -- the owner side of table-sharded classification (BASELINE configs[3]) as a kernel of its own: point
// lookups of the keys received from the other ranks, for the steps of the pipeline that have no scan to carry them (its first and
// last batches; lane.hip's step kernel answers the others inside the scan) and for the staged route.
// The lookup is the left join + spanToHit's otherwise(NONE) (S/slacken/Classifier.scala:84, KeyValueIndex.scala:176-185) for
// keys that arrive without their fragment.
#include <hip/hip_runtime.h>

#include "engine.h"

#include <cstdlib>

namespace slk {

namespace {

constexpr int SW = 4;  // waves per block

__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 64 keys per wave iteration: lane i hashes key i; LPB lanes then read one bucket together (LPB x 16 B), 64 / LPB buckets per
// wave instruction, LPB instructions in flight -- the access shape of the classify kernel's probe.  A key whose home bucket
// is full without holding it, and has overflowed (its flag), continues alone in the next buckets.
__global__ void __launch_bounds__(SW * 64) lookup_coop_kernel(TableView T, const int64_t *__restrict__ keys, uint64_t n,
                                                              int32_t *__restrict__ out) {
  constexpr int PG = 64 / LPB;
  __shared__ __attribute__((aligned(16))) uint4 stash_all[SW][64];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint4 *stash = stash_all[wib];
  const uint64_t nwaves = (uint64_t)gridDim.x * SW;
  const uint64_t tmask = (1ULL << T.g.taxon_bits) - 1;
  const int g = lane / LPB, c = lane % LPB;
  const char *cellbase = (const char *)T.cells + c * 16;
  const uint64_t units = (n + 63) / 64;
  for (uint64_t unit = (uint64_t)blockIdx.x * SW + wib; unit < units; unit += nwaves) {
    const uint64_t base = unit * 64;
    const uint64_t i = base + lane;
    const bool in = i < n;
    const uint64_t key = in ? (uint64_t)keys[i] : 0;
    const uint64_t h = fmix64(key);
    uint32_t home;
    uint64_t rem_hi;
    table_slot(T.g, h, home, rem_hi);
    const uint64_t tag = in ? rem_hi : ~0ULL;  // displacement 0
    uint4 st;
    st.x = home;
    st.y = 0;                // taxon found
    st.z = (uint32_t)tag;
    st.w = (uint32_t)(tag >> 32);
    stash[lane] = st;
    wsync();
    ulonglong2 cell[LPB];
#pragma unroll
    for (int s = 0; s < LPB; s++) cell[s] = *(const ulonglong2 *)(cellbase + ((uint64_t)stash[s * PG + g].x << BUCKET_SHIFT));
    uint32_t unresolved = 0;  // bit s: this lane's group of step s found neither its key nor the end of its probe sequence
#pragma unroll
    for (int s = 0; s < LPB; s++) {
      const uint64_t want = ((uint64_t)stash[s * PG + g].w << 32) | stash[s * PG + g].z;
      const bool act = want != ~0ULL;
      const bool e0 = cell[s].x == 0, e1 = cell[s].y == 0;
      const bool m0 = act && !e0 && cell_tag(T.g, cell[s].x) == want;
      const bool m1 = act && !e1 && cell_tag(T.g, cell[s].y) == want;
      if (m0 || m1) stash[s * PG + g].y = (uint32_t)((m0 ? cell[s].x : cell[s].y) & tmask);
      const bool closed = T.g.flag != 0 && c == 0 && (cell[s].x & T.g.flag) == 0;   // full, but no record ever went past it
      const uint64_t B = __ballot(m0 || m1 || e0 || e1 || !act || closed);
      if (((B >> (g * LPB)) & ((1u << LPB) - 1)) == 0) unresolved |= 1u << s;
    }
    wsync();
    int32_t taxon = (int32_t)stash[lane].y;
    // was entry `lane` unresolved?  its group was g' = lane % PG of step s' = lane / PG: ask lane LPB * g'
    const uint32_t ur = (uint32_t)__shfl((int)unresolved, (lane % PG) * LPB);
    if (in && ((ur >> (lane / PG)) & 1)) {
      // bucket-level linear probing continues in the next buckets (cells are never freed: stop at an empty cell)
      for (int d = 1; d <= T.max_disp; d++) {
        const ulonglong2 *b = (const ulonglong2 *)(T.cells + ((uint64_t)table_bucket(T.g, home, (uint32_t)d) * CELLS));
        const uint64_t want = rem_hi | (uint64_t)d;
        bool has_empty = false, closed = false;
        int32_t found = 0;
#pragma unroll
        for (int q = 0; q < LPB; q++) {
          ulonglong2 v = b[q];
          has_empty |= (v.x == 0) | (v.y == 0);
          if (q == 0) closed = T.g.flag != 0 && (v.x & T.g.flag) == 0;
          if (v.x != 0 && cell_tag(T.g, v.x) == want) found = (int32_t)(v.x & tmask);
          if (v.y != 0 && cell_tag(T.g, v.y) == want) found = (int32_t)(v.y & tmask);
        }
        if (found) { taxon = found; break; }
        if (has_empty || closed) break;
      }
    }
    if (in) out[i] = ext_taxon(T, taxon);
    wsync();
  }
}

}  // namespace

void launch_lookup_coop(const TableView &t, const int64_t *keys, uint64_t n, int32_t *out, hipStream_t s) {
  if (n == 0) return;
  // The kernel is bound by the memory system's request rate and, per wave, by two dependent round trips (the key, then its bucket):
  // it wants every wave slot it can get (64 waves per CU: 12.8 ms per 3.9e8 keys with 8 waves per CU, alone, against a 68 GiB table).
  // Round 2 ran it BESIDE the scans of the next batch with 8 waves per CU (SLK_LOOKUP_BLOCKS_PER_CU=2), which was best against a
  // table small enough for the Infinity Cache; now the stages of the sharded pipeline follow each other on one stream.
  static const int bpc = getenv("SLK_LOOKUP_BLOCKS_PER_CU") ? std::max(1, atoi(getenv("SLK_LOOKUP_BLOCKS_PER_CU"))) : 16;
  uint64_t blocks = std::min<uint64_t>((n + SW * 64 - 1) / (SW * 64), (uint64_t)256 * bpc);
  hipLaunchKernelGGL(lookup_coop_kernel, dim3((unsigned)blocks), dim3(SW * 64), 0, s, t, keys, n, out);
}
}  // namespace slk

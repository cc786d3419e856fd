// build.hip -- library construction on the device: minimizers of taxon-labelled sequences, merged by LCA, straight into the
// HBM record table.  Replaces, for one batch of sequences, the reference's
//   SplitterMinimizers.find            S/slacken/Minimizers.scala:43-76   (every super-mer's minimizer, labelled with the taxon)
//   groupBy(id columns).agg(TaxonLCA)  S/slacken/KeyValueIndex.scala:85-93, S/slacken/LowestCommonAncestor.scala:152-170
// (S/ = src/main/scala/com/jnpersson/ in the reference).  The group-by needs no sort here: the table is the group-by.  A
// record is inserted with atomicCAS, and a key that is already present has its taxon replaced by LCA(old, new) in a CAS loop.
// LCA is associative and commutative on a rooted tree, so the final taxon of every key is independent of the order in which
// lanes, waves and batches arrive; only the cell a record occupies may differ between runs, which no lookup can observe.
//
// Work decomposition: the host cuts every sequence into chunks of CHUNK_WINDOWS k-mer windows that overlap by k-1 bases (the
// same overlap the reference's indexed-FASTA reader uses, FileInputs.scala:240-262): the SET of window minimizers is unchanged.
// One lane per chunk, 64 chunks per wave in lockstep; minimizers go to a wave-shared LDS queue and are inserted 64 at a time,
// one lane per record, so that the latency of the atomics is paid once per 64 records.
#include <hip/hip_runtime.h>

#include "engine.h"

namespace slk {

namespace {

constexpr int BW = 4;  // waves per block

// Insert (key, taxon) or merge the taxon into the existing record.  Returns 1 if a new record was created, 0 if merged (or
// the key belongs to another rank's shard of the table), -1 if no cell could be found within the displacement limit.
__device__ int insert_merge(const TableBuild &t, const int32_t *parents, int32_t ntax, uint64_t key, int32_t taxon, int &max_d) {
  const uint64_t h = fmix64(key);
  if (!shard_keeps(t, h)) return 0;
  uint32_t home;
  uint64_t rem_hi;
  table_slot(t.g, h, home, rem_hi);
  const unsigned long long tmask = (1ULL << t.g.taxon_bits) - 1;
  for (int d = 0; d <= t.disp_limit; d++) {
    unsigned long long *bucket = (unsigned long long *)(t.cells + ((uint64_t)table_bucket(t.g, home, (uint32_t)d) * CELLS));
    const unsigned long long tag = rem_hi | (uint64_t)d;
    const unsigned long long val = (tag << t.g.taxon_bits) | (uint32_t)taxon;
    unsigned long long first = 0;
    for (int c = 0; c < CELLS; c++) {
      unsigned long long cur = __hip_atomic_load(&bucket[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == 0) {
        unsigned long long old = atomicCAS(&bucket[c], 0ULL, val);
        if (old == 0) {
          max_d = max(max_d, d);
          return 1;
        }
        cur = old;  // somebody else took this cell: it may be this very key
      }
      if (c == 0) first = cur;
      if (cell_tag(t.g, cur) == tag) {
        for (;;) {
          int32_t old_taxon = (int32_t)(cur & tmask);
          int32_t merged = tax_lca(parents, ntax, old_taxon, taxon);
          if (merged == old_taxon) return 0;
          unsigned long long want = (cur & ~tmask) | (uint32_t)merged;   // (the bucket flag in a first cell's top bit stays)
          unsigned long long prev = atomicCAS(&bucket[c], cur, want);
          if (prev == cur) return 0;
          cur = prev;    // (the taxon was merged by another lane, or the bucket flag was raised meanwhile: again)
        }
      }
    }
    // full, and the key is not here: the record goes on, and the bucket says so from now on (engine.h: TableGeom.flag)
    if (t.g.flag && !(first & t.g.flag)) atomicOr(&bucket[0], (unsigned long long)t.g.flag);
  }
  return -1;
}

struct BuildLds {
  uint64_t q_key[BW][128];
  int32_t q_tax[BW][128];
};

__device__ __forceinline__ int code_of(uint32_t c) {  // BitRepresentation.charToTwobit :127-135; 5 = not a nucleotide
  const uint32_t VM = (1u << 1) | (1u << 3) | (1u << 7) | (1u << 20) | (1u << 21);  // A C G T U, either case
  bool ok = ((c & 0xC0) == 0x40) && ((VM >> (c & 31)) & 1);
  uint32_t t = (c >> 1) & 3;
  t ^= t >> 1;
  return ok ? (int)t : 5;
}

__global__ void __launch_bounds__(BW * 64) build_kernel(ScanParams P, TableBuild T, const int32_t *__restrict__ parents,
                                                        int32_t ntax, const uint8_t *__restrict__ bases, uint64_t total_bases,
                                                        const uint64_t *__restrict__ chunk_start,
                                                        const uint32_t *__restrict__ chunk_len,
                                                        const int32_t *__restrict__ chunk_taxon, uint64_t nchunks) {
  extern __shared__ uint64_t ring[];  // [w][BW*64]: the last w keys of every lane
  __shared__ BuildLds L;
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wib = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t stride = BW * 64;
  const int w = P.w, k = P.k, m = P.m;
  const uint64_t c = (uint64_t)blockIdx.x * stride + tid;
  const uint32_t len = c < nchunks ? chunk_len[c] : 0;
  const uint64_t start = c < nchunks ? chunk_start[c] : 0;
  const int32_t taxon = c < nchunks ? chunk_taxon[c] : 0;
  uint32_t maxlen = len;
  for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, o));
  maxlen = __builtin_amdgcn_readfirstlane(maxlen);

  uint64_t fwd = 0, rc = 0, minv = ~0ULL, cur_val = 0, lo = 0, hi = 0;
  uint32_t nvalid = 0;
  int head = w - 1, minage = 0;
  bool have_cur = false;
  uint32_t qhead = 0, qn = 0;             // wave-uniform
  int created = 0, failed = 0, max_d = 0;  // per lane

  auto flush = [&](uint32_t cnt) {        // insert cnt (<= 64) queued records, one per lane
    if (lane < cnt) {
      uint32_t e = (qhead + lane) & 127;
      int r = insert_merge(T, parents, ntax, L.q_key[wib][e], L.q_tax[wib][e], max_d);
      created += (r == 1);
      failed += (r < 0);
    }
    qhead = (qhead + cnt) & 127;
    qn -= cnt;
  };

  for (uint32_t step = 0; step < maxlen; step++) {
    if ((step & 15) == 0) {  // 16 bytes per lane every 16 steps (the buffer's last block: byte loads, engine.h)
      uint4 v = make_uint4(0, 0, 0, 0);
      if (step < len) v = load_block16(bases + start + step, clamp_room(total_bases - start - step));
      lo = ((uint64_t)v.y << 32) | v.x;
      hi = ((uint64_t)v.w << 32) | v.z;
    }
    uint32_t ch = (uint32_t)(lo & 0xff);
    lo = (lo >> 8) | (hi << 56);
    hi >>= 8;
    int t = (step < len) ? code_of(ch) : 5;
    bool emit = false;
    if (t >= 4) {  // InputReader.removeInvalid (InputReader.scala:60-72): sequences are split around anything else
      nvalid = 0; fwd = 0; rc = 0; head = w - 1; minage = 0; minv = ~0ULL; have_cur = false;
    } else {
      nvalid++;
      fwd = (fwd << 2) | ((uint64_t)t << P.sh);
      rc = ((rc >> 2) | ((uint64_t)(3 - t) << 62)) & P.keep;
      if (nvalid >= (uint32_t)m) {
        uint64_t canon = (P.canonical && rc < fwd) ? rc : fwd;
        uint64_t key = (canon ^ P.xmask) & P.smask;
        head = (head + 1 == w) ? 0 : head + 1;
        ring[(uint32_t)head * stride + tid] = key;
        if (key <= minv) { minv = key; minage = 0; }
        else if (++minage >= w) {
          int slot = (head + 1 == w) ? 0 : head + 1;
          minv = ~0ULL;
          for (int a = w - 1; a >= 0; a--) {
            uint64_t v = ring[(uint32_t)slot * stride + tid];
            if (v <= minv) { minv = v; minage = a; }
            slot = (slot + 1 == w) ? 0 : slot + 1;
          }
        }
        if (nvalid >= (uint32_t)k && (!have_cur || minv != cur_val)) {  // a new super-mer starts: one record candidate
          have_cur = true;
          cur_val = minv;
          emit = true;
        }
      }
    }
    uint64_t mask = __ballot(emit);
    if (mask) {
      uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
      if (emit) {
        uint32_t e = (qhead + qn + before) & 127;
        L.q_key[wib][e] = cur_val;
        L.q_tax[wib][e] = taxon;
      }
      qn += (uint32_t)__popcll(mask);
      if (qn >= 64) flush(64);
    }
  }
  if (qn) flush(qn);  // qn < 64 here

  for (int o = 32; o > 0; o >>= 1) {
    created += __shfl_xor(created, o);
    failed += __shfl_xor(failed, o);
    max_d = max(max_d, __shfl_xor(max_d, o));
  }
  if (lane == 0) {
    if (created) atomicAdd(T.n_inserted, (unsigned long long)created);
    if (failed) atomicAdd(T.n_overflow, (unsigned long long)failed);
    if (max_d) atomicMax(T.max_disp, max_d);
  }
}

__host__ __device__ inline uint64_t fmix64_inverse(uint64_t x) {
  x ^= x >> 33; x *= 0x9cb4b2f8129337dbULL;  // inverse of 0xc4ceb9fe1a85ec53 mod 2^64
  x ^= x >> 33; x *= 0x4f74430c22a54005ULL;  // inverse of 0xff51afd7ed558ccd mod 2^64
  x ^= x >> 33;
  return x;
}

// Every occupied cell back to its (key, taxon) record: the cell holds the hash remainder and its displacement, the bucket
// index gives the home bucket, and both the range reduction (engine.h: table_hash_of) and fmix64 are invertible.
__global__ void __launch_bounds__(256) export_kernel(TableView T, uint64_t cell0, uint64_t ncells, int64_t *__restrict__ keys,
                                                     int32_t *__restrict__ taxa, uint64_t capacity,
                                                     unsigned long long *__restrict__ counter) {
  uint64_t i;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t tmask = (1ULL << T.g.taxon_bits) - 1, dmask = (1ULL << T.g.disp_bits) - 1;
  for (uint64_t base = cell0 + (uint64_t)blockIdx.x * blockDim.x; base < ncells; base += step) {  // wave-uniform trip count
    i = base + threadIdx.x;
    uint64_t cell = i < ncells ? T.cells[i] : 0;
    const bool has = cell != 0;
    const uint64_t mask = __ballot(has);
    if (mask == 0) continue;
    // one atomic per wave: the lanes' output slots are consecutive
    unsigned long long first = 0;
    const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
    if (has && before == 0) first = atomicAdd(counter, (unsigned long long)__popcll(mask));
    const int leader = __ffsll((long long)mask) - 1;
    first = ((unsigned long long)(uint32_t)__shfl((int)(first >> 32), leader) << 32) | (uint32_t)__shfl((int)first, leader);
    if (!has) continue;
    const uint64_t tag = cell_tag(T.g, cell);
    const uint64_t bucket = i / CELLS, d = tag & dmask;
    const uint32_t home = (uint32_t)(bucket >= d ? bucket - d : bucket + T.g.nbuckets - d);
    const uint64_t h = table_hash_of(T.g, home, tag >> T.g.disp_bits);
    unsigned long long slot = first + before;
    if (slot < capacity) {
      keys[slot] = (int64_t)fmix64_inverse(h);
      taxa[slot] = ext_taxon(T, (int32_t)(cell & tmask));
    }
  }
}

}  // namespace

void launch_build(const ScanParams &P, const TableBuild &T, const int32_t *parents, int32_t ntax, const uint8_t *bases,
                  uint64_t total_bases, const uint64_t *chunk_start, const uint32_t *chunk_len, const int32_t *chunk_taxon, uint64_t nchunks,
                  hipStream_t s) {
  if (nchunks == 0) return;
  const unsigned block = BW * 64;
  size_t lds = (size_t)block * P.w * 8;
  uint64_t blocks = (nchunks + block - 1) / block;
  hipLaunchKernelGGL(build_kernel, dim3((unsigned)blocks), dim3(block), lds, s, P, T, parents, ntax, bases, total_bases,
                     chunk_start, chunk_len, chunk_taxon, nchunks);
}

void launch_export(const TableView &T, uint64_t nbuckets, int64_t *keys, int32_t *taxa, uint64_t capacity,
                   unsigned long long *counter, hipStream_t s) {
  launch_export_range(T, 0, nbuckets, keys, taxa, capacity, counter, s);
}
// the records of buckets [bucket0, bucket1) (a table that is being moved to a larger one goes there piece by piece)
void launch_export_range(const TableView &T, uint64_t bucket0, uint64_t bucket1, int64_t *keys, int32_t *taxa, uint64_t capacity,
                         unsigned long long *counter, hipStream_t s) {
  if (bucket1 <= bucket0) return;
  const uint64_t cell0 = bucket0 * CELLS, ncells = bucket1 * CELLS;
  uint64_t blocks = std::min<uint64_t>((ncells - cell0 + 255) / 256, 256 * 64);
  hipLaunchKernelGGL(export_kernel, dim3((unsigned)blocks), dim3(256), 0, s, T, cell0, ncells, keys, taxa, capacity, counter);
}

}  // namespace slk

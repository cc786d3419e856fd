// wide.hip -- the classify path for minimizers wider than 32 nt (m <= 128: 2..4 id columns; KeyValueIndex.scala:49,
// NTBitArray with several longs, MinimizerPriorities over several words).  The staged shape of kernels.hip with W-word keys:
// a lane-per-fragment scan into per-fragment span regions, one lane per span for the lookup, and the classify kernel of
// kernels.hip unchanged.  Correctness first: the libraries the reference documents all use m = 31, which the fused kernels
// serve; this path exists so that no valid splitter is refused.
//   scan     Supermers.splitFragment / spans (S/slacken/Supermers.scala:49-125) over MinSplitter.splitRead
//            (S/kmers/minimizer/MinSplitter.scala:133-172): rolling forward and reverse-complement m-mers over W words
//            (NTBitArray.shiftAddBP, NTBitArray.scala:140-150,418-421), canonical orientation = the smaller of the two as
//            left-aligned numbers (writeCanonical :258-266), XOR and space masks per word, sliding minimum by value over
//            the last w keys (PosRankWindow), run-length merge of equal minima.
//   lookup   the left join on (id1..idW) (Classifier.scala:84).
#include <hip/hip_runtime.h>

#include "engine.h"

namespace slk {

namespace {

template <int W> struct Key {
  uint64_t v[W];
};
template <int W> __device__ __forceinline__ bool key_less(const Key<W> &a, const Key<W> &b) {  // unsigned, word 0 first
#pragma unroll
  for (int i = 0; i < W; i++) {
    if (a.v[i] != b.v[i]) return a.v[i] < b.v[i];
  }
  return false;
}
template <int W> __device__ __forceinline__ bool key_eq(const Key<W> &a, const Key<W> &b) {
  bool e = true;
#pragma unroll
  for (int i = 0; i < W; i++) e = e && a.v[i] == b.v[i];
  return e;
}
template <int W> __device__ __forceinline__ uint64_t key_hash(const uint64_t *k) {
  uint64_t h = 0;
#pragma unroll
  for (int i = 0; i < W; i++) h = fmix64(h ^ k[i]) + 0x9E3779B97F4A7C15ULL * (uint64_t)(i + 1);
  return fmix64(h);
}

__device__ __forceinline__ int wide_code(uint8_t c) {  // BitRepresentation.charToTwobit :127-135; 5 = not a nucleotide
  const uint32_t VM = (1u << 1) | (1u << 3) | (1u << 7) | (1u << 20) | (1u << 21);
  bool ok = ((c & 0xC0) == 0x40) && ((VM >> (c & 31)) & 1);
  uint32_t t = (c >> 1) & 3;
  t ^= t >> 1;
  return ok ? (int)t : 5;
}

template <int W> struct WideWriter {
  uint64_t *keys;
  int32_t *meta;
  uint64_t base;
  int32_t n;
  bool first, have_last;
  Key<W> last;
  __device__ __forceinline__ void emit(const Key<W> &key, int32_t kmers, int32_t flag) {
    bool seqlike = flag == 1;
    bool distinct = seqlike && (first || !(have_last && key_eq<W>(key, last)));  // Supermers.spans :84-90
    if (seqlike) { last = key; have_last = true; }
    first = false;
#pragma unroll
    for (int i = 0; i < W; i++) keys[(base + n) * W + i] = seqlike ? key.v[i] : 0;
    meta[base + n] = pack_meta(kmers, flag, distinct ? 1 : 0);
    n++;
  }
};

// One mate (the W-word counterpart of kernels.hip scan_mate).  ring[(slot * W + i) * blockDim.x + threadIdx.x].
template <int W>
__device__ void wide_scan_mate(const WideParams &P, const uint8_t *__restrict__ seq, uint32_t n, uint64_t *ring, WideWriter<W> &out) {
  const int k = P.k, m = P.m, w = P.w;
  const uint32_t stride = blockDim.x, tid = threadIdx.x;
  const int ins_word = (m - 1) >> 5, ins_shift = 62 - 2 * ((m - 1) & 31);
  const uint64_t keep_last = P.last_sh == 0 ? ~0ULL : (~0ULL << P.last_sh);
  int run_class = 0;
  uint32_t run_len = 0, nvalid = 0;
  Key<W> fwd{}, rc{}, minv{}, cur_val{};
  int head = 0, minage = 0;
  int32_t cur_run = 0;
  Key<W> zero{};
  for (uint32_t i = 0; i <= n; i++) {
    int t = 5, cls = -1;
    if (i < n) { t = wide_code(seq[i]); cls = t < 4 ? 1 : 0; }
    if (run_len > 0 && cls != run_class) {
      if (run_class == 1 && nvalid >= (uint32_t)k) out.emit(cur_val, cur_run, 1);
      else if (run_len >= (uint32_t)k) out.emit(zero, (int32_t)run_len - (k - 1), 2);  // Supermers.scala:116-119
      run_len = 0;
    }
    if (i == n) break;
    if (run_len == 0) {
      run_class = cls; nvalid = 0; head = w - 1; minage = 0; cur_run = 0;
#pragma unroll
      for (int j = 0; j < W; j++) { fwd.v[j] = 0; rc.v[j] = 0; minv.v[j] = ~0ULL; }
    }
    run_len++;
    if (t < 4) {
      nvalid++;
      // forward: the W-word number moves left by one nucleotide, the new one enters at nucleotide position m-1
#pragma unroll
      for (int j = 0; j < W; j++) fwd.v[j] = (fwd.v[j] << 2) | (j + 1 < W ? fwd.v[j + 1] >> 62 : 0);
      fwd.v[ins_word] |= (uint64_t)t << ins_shift;
      // reverse complement: moves right, the complement enters at position 0; bits beyond position m-1 are dropped
#pragma unroll
      for (int j = W - 1; j >= 0; j--) rc.v[j] = (rc.v[j] >> 2) | (j > 0 ? rc.v[j - 1] << 62 : 0);
      rc.v[0] |= (uint64_t)(3 - t) << 62;
      rc.v[W - 1] &= keep_last;
      if (nvalid >= (uint32_t)m) {
        const bool use_rc = P.canonical && key_less<W>(rc, fwd);
        Key<W> key;
#pragma unroll
        for (int j = 0; j < W; j++) key.v[j] = ((use_rc ? rc.v[j] : fwd.v[j]) ^ P.xmask[j]) & P.smask[j];
        head = (head + 1 == w) ? 0 : head + 1;
#pragma unroll
        for (int j = 0; j < W; j++) ring[((uint32_t)head * W + j) * stride + tid] = key.v[j];
        if (!key_less<W>(minv, key)) { minv = key; minage = 0; }   // key <= minv
        else if (++minage >= w) {  // the minimum left the window: rescan the last w keys (oldest first)
          int slot = (head + 1 == w) ? 0 : head + 1;
#pragma unroll
          for (int j = 0; j < W; j++) minv.v[j] = ~0ULL;
          for (int a = w - 1; a >= 0; a--) {
            Key<W> v;
#pragma unroll
            for (int j = 0; j < W; j++) v.v[j] = ring[((uint32_t)slot * W + j) * stride + tid];
            if (!key_less<W>(minv, v)) { minv = v; minage = a; }
            slot = (slot + 1 == w) ? 0 : slot + 1;
          }
        }
        if (nvalid >= (uint32_t)k) {
          if (cur_run == 0) { cur_val = minv; cur_run = 1; }
          else if (key_eq<W>(minv, cur_val)) cur_run++;               // MinSplitter.splitRead :154-158
          else { out.emit(cur_val, cur_run, 1); cur_val = minv; cur_run = 1; }
        }
      }
    }
  }
}

template <int W>
__global__ void __launch_bounds__(64) wide_scan_kernel(WideParams P, const uint8_t *__restrict__ bases,
                                                       const uint64_t *__restrict__ offsets,
                                                       const uint8_t *__restrict__ mate_bases,
                                                       const uint64_t *__restrict__ mate_offsets, uint64_t R,
                                                       uint64_t *__restrict__ span_keys, int32_t *__restrict__ span_meta,
                                                       int32_t *__restrict__ span_count) {
  extern __shared__ uint64_t wide_ring[];
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  WideWriter<W> out;
  out.keys = span_keys; out.meta = span_meta;
  out.base = span_region(offsets, mate_offsets, r);
  out.n = 0; out.first = true; out.have_last = false;
  uint64_t o0 = offsets[r];
  wide_scan_mate<W>(P, bases + o0, (uint32_t)(offsets[r + 1] - o0), wide_ring, out);
  if (mate_bases) {
    Key<W> zero{};
    out.emit(zero, -(P.k - 1), 3);  // MATE_PAIR_BORDER pseudo-span (Supermers.scala:53-57)
    uint64_t m0 = mate_offsets[r];
    wide_scan_mate<W>(P, mate_bases + m0, (uint32_t)(mate_offsets[r + 1] - m0), wide_ring, out);
  }
  span_count[r] = out.n;
}

template <int W>
__device__ __forceinline__ int32_t wide_find(const WideTable &t, const uint64_t *key) {
  uint64_t slot = key_hash<W>(key) & t.mask;
  for (uint64_t step = 0; step <= t.mask; step++) {
    int32_t taxon = t.taxa[slot];
    if (taxon == 0) return 0;  // slots are never freed: an empty slot ends the probe sequence
    bool eq = true;
#pragma unroll
    for (int i = 0; i < W; i++) eq = eq && t.keys[slot * W + i] == key[i];
    if (eq) return taxon;
    slot = (slot + 1) & t.mask;
  }
  return 0;
}

template <int W>
__global__ void __launch_bounds__(256) wide_insert_kernel(WideTable t, const int64_t *__restrict__ keys,
                                                          const int32_t *__restrict__ taxa, uint64_t n,
                                                          unsigned long long *__restrict__ counters) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  int n_ins = 0, n_ovf = 0;
  for (; i < n; i += stride) {
    const int32_t taxon = taxa[i];
    if (taxon == 0) continue;  // a record with taxon NONE is indistinguishable from a miss
    uint64_t k[W];
#pragma unroll
    for (int j = 0; j < W; j++) k[j] = (uint64_t)keys[i * W + j];
    uint64_t slot = key_hash<W>(k) & t.mask;
    bool done = false;
    for (uint64_t step = 0; step <= t.mask && !done; step++) {
      // Keys are unique (makeRecords' groupBy): claiming a slot needs no look at the keys other records are still writing.
      if (atomicCAS((int *)&t.taxa[slot], 0, taxon) == 0) {
#pragma unroll
        for (int j = 0; j < W; j++) t.keys[slot * W + j] = k[j];
        done = true;
        n_ins++;
      } else {
        slot = (slot + 1) & t.mask;
      }
    }
    if (!done) n_ovf++;
  }
  for (int o = 32; o > 0; o >>= 1) { n_ins += __shfl_xor(n_ins, o); n_ovf += __shfl_xor(n_ovf, o); }
  if ((threadIdx.x & 63) == 0) {
    if (n_ins) atomicAdd(&counters[0], (unsigned long long)n_ins);
    if (n_ovf) atomicAdd(&counters[2], (unsigned long long)n_ovf);
  }
}

template <int W>
__global__ void __launch_bounds__(256) wide_lookup_kernel(WideTable t, const int64_t *__restrict__ keys, uint64_t n,
                                                          int32_t *__restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t k[W];
#pragma unroll
    for (int j = 0; j < W; j++) k[j] = (uint64_t)keys[i * W + j];
    out[i] = wide_find<W>(t, k);
  }
}

template <int W>
__global__ void __launch_bounds__(256) wide_probe_kernel(WideTable t, const uint64_t *__restrict__ offsets,
                                                         const uint64_t *__restrict__ mate_offsets, uint64_t R,
                                                         const uint64_t *__restrict__ span_keys,
                                                         const int32_t *__restrict__ span_meta,
                                                         const int32_t *__restrict__ span_count,
                                                         int32_t *__restrict__ span_taxon) {
  const uint32_t lane = threadIdx.x & 63;
  uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t r = wave; r < R; r += nwaves) {
    const uint64_t base = span_region(offsets, mate_offsets, r);
    const int32_t n = span_count[r];
    for (int32_t j = lane; j < n; j += 64) {
      const int32_t flag = meta_flag(span_meta[base + j]);
      int32_t taxon;
      if (flag == 2) taxon = -1;        // spanToHit (KeyValueIndex.scala:176-185): the flag wins over any record
      else if (flag == 3) taxon = -2;
      else {
        uint64_t k[W];
#pragma unroll
        for (int i = 0; i < W; i++) k[i] = span_keys[(base + j) * W + i];
        taxon = wide_find<W>(t, k);
      }
      span_taxon[base + j] = taxon;
    }
  }
}

// ---- library construction with W-word keys (slk_index_add_sequences; KeyValueIndex.makeRecords, KeyValueIndex.scala:85-93) ----
// The SEQUENCE-flag spans of the scanned chunks are the super-mers' minimizers (SplitterMinimizers.find, Minimizers.scala:43-76:
// library sequences are split around anything that is not a nucleotide, which is what the scan's run splitting does; runs it
// flags as ambiguous carry no minimizer).  One lane per span inserts (key, taxon) or merges the taxon into the record that is there
// by LCA (TaxonLCA, LowestCommonAncestor.scala:152-170).  A slot is claimed by its taxon word: 0 -> CLAIMED, key words written,
// then the taxon published; a lane that meets a CLAIMED slot comes back to it in its next round -- nobody waits inside a round, so
// lanes of one wave cannot hold each other up.
constexpr int32_t WIDE_CLAIMED = -1;
template <int W>
__global__ void __launch_bounds__(256) wide_build_insert_kernel(WideTable t, const int32_t *__restrict__ parents, int32_t ntax,
                                                                const uint64_t *__restrict__ offsets, uint64_t R,
                                                                const uint64_t *__restrict__ span_keys, const int32_t *__restrict__ span_meta,
                                                                const int32_t *__restrict__ span_count, const int32_t *__restrict__ chunk_taxon,
                                                                unsigned long long *__restrict__ counters) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  int created = 0, failed = 0;
  for (uint64_t r = wave; r < R; r += nwaves) {
    const uint64_t base = offsets[r];
    const int32_t n = span_count[r], taxon = chunk_taxon[r];
    for (int32_t j0 = 0; j0 < n; j0 += 64) {
      const int32_t j = j0 + (int32_t)lane;
      bool todo = j < n && meta_flag(span_meta[base + j]) == 1;
      uint64_t k[W];
#pragma unroll
      for (int i = 0; i < W; i++) k[i] = todo ? span_keys[(base + j) * W + i] : 0;
      uint64_t slot = key_hash<W>(k) & t.mask, steps = 0;
      while (__ballot(todo) != 0) {
        if (todo) {
          int32_t cur = __hip_atomic_load(&t.taxa[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          if (cur == 0) {
            if (atomicCAS((int *)&t.taxa[slot], 0, WIDE_CLAIMED) == 0) {
#pragma unroll
              for (int i = 0; i < W; i++) t.keys[slot * W + i] = k[i];
              __hip_atomic_store(&t.taxa[slot], taxon, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
              created++;
              todo = false;
            }                              // (lost the race: look at the slot again next round)
          } else if (cur != WIDE_CLAIMED) {
            bool eq = true;
#pragma unroll
            for (int i = 0; i < W; i++) eq = eq && __hip_atomic_load(&t.keys[slot * W + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == k[i];
            if (eq) {
              for (;;) {                   // (the word holds a published taxon from here on: only LCA merges change it)
                const int32_t merged = tax_lca(parents, ntax, cur, taxon);
                if (merged == cur) break;
                const int32_t prev = atomicCAS((int *)&t.taxa[slot], cur, merged);
                if (prev == cur) break;
                cur = prev;
              }
              todo = false;
            } else {
              slot = (slot + 1) & t.mask;
              if (++steps > t.mask) { failed++; todo = false; }
            }
          }
        }
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) { created += __shfl_xor(created, o); failed += __shfl_xor(failed, o); }
  if (lane == 0) {
    if (created) atomicAdd(&counters[0], (unsigned long long)created);
    if (failed) atomicAdd(&counters[2], (unsigned long long)failed);
  }
}

// the table's records as (W key words, taxon) rows (slk_index_export)
template <int W>
__global__ void __launch_bounds__(256) wide_export_kernel(WideTable t, int64_t *__restrict__ keys, int32_t *__restrict__ taxa, uint64_t capacity,
                                                          unsigned long long *__restrict__ counter) {
  const uint64_t nslots = t.mask + 1;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (uint64_t)gridDim.x * blockDim.x) {
    const int32_t taxon = t.taxa[i];
    if (taxon == 0) continue;
    const unsigned long long at = atomicAdd(counter, 1ULL);
    if (at < capacity) {
#pragma unroll
      for (int j = 0; j < W; j++) keys[at * W + j] = (int64_t)t.keys[i * W + j];
      taxa[at] = taxon;
    }
  }
}

// span slots -> the caller's dense arrays (slk_spans_batch_wide): the slk_span records (key = id1) and the key rows beside them
struct WideSpanOut { int64_t key; int32_t kmers; int8_t flag; uint8_t distinct; uint16_t pad; };
template <int W>
__global__ void __launch_bounds__(256) wide_gather_spans_kernel(const uint64_t *__restrict__ offsets, const uint64_t *__restrict__ mate_offsets,
                                                                uint64_t R, const uint64_t *__restrict__ span_keys,
                                                                const int32_t *__restrict__ span_meta, const uint64_t *__restrict__ out_offsets,
                                                                WideSpanOut *__restrict__ out, int64_t *__restrict__ out_keys) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t r = wave; r < R; r += nwaves) {
    const uint64_t base = span_region(offsets, mate_offsets, r), o = out_offsets[r];
    const uint64_t n = out_offsets[r + 1] - o;
    for (uint64_t j = lane; j < n; j += 64) {
      const int32_t m = span_meta[base + j];
      WideSpanOut s;
      s.key = (int64_t)span_keys[(base + j) * W];
      s.kmers = meta_kmers(m); s.flag = (int8_t)meta_flag(m); s.distinct = (uint8_t)meta_distinct(m); s.pad = 0;
      out[o + j] = s;
#pragma unroll
      for (int i = 0; i < W; i++) out_keys[(o + j) * W + i] = (int64_t)span_keys[(base + j) * W + i];
    }
  }
}

}  // namespace

#define WIDE_DISPATCH(W, CALL) \
  switch (W) {                 \
    case 2: { constexpr int WW = 2; CALL; break; } \
    case 3: { constexpr int WW = 3; CALL; break; } \
    default: { constexpr int WW = 4; CALL; break; } \
  }

void launch_wide_insert(const WideTable &t, int W, const int64_t *keys, const int32_t *taxa, uint64_t n, unsigned long long *counters,
                        hipStream_t s) {
  if (n == 0) return;
  uint64_t blocks = std::min<uint64_t>((n + 255) / 256, 8192);
  WIDE_DISPATCH(W, hipLaunchKernelGGL(wide_insert_kernel<WW>, dim3((unsigned)blocks), dim3(256), 0, s, t, keys, taxa, n, counters));
}
void launch_wide_lookup(const WideTable &t, int W, const int64_t *keys, uint64_t n, int32_t *out, hipStream_t s) {
  if (n == 0) return;
  uint64_t blocks = std::min<uint64_t>((n + 255) / 256, 8192);
  WIDE_DISPATCH(W, hipLaunchKernelGGL(wide_lookup_kernel<WW>, dim3((unsigned)blocks), dim3(256), 0, s, t, keys, n, out));
}
void launch_wide_scan(const WideParams &P, const uint8_t *bases, const uint64_t *offsets, const uint8_t *mate_bases,
                      const uint64_t *mate_offsets, uint64_t R, uint64_t *span_keys, int32_t *span_meta, int32_t *span_count,
                      hipStream_t s) {
  if (R == 0) return;
  const unsigned block = 64;
  size_t lds = (size_t)block * P.w * P.W * 8;  // <= 64 KiB: w * W <= 128 (checked at index creation)
  uint64_t blocks = (R + block - 1) / block;
  WIDE_DISPATCH(P.W, hipLaunchKernelGGL(wide_scan_kernel<WW>, dim3((unsigned)blocks), dim3(block), lds, s, P, bases, offsets,
                                        mate_bases, mate_offsets, R, span_keys, span_meta, span_count));
}
void launch_wide_probe(const WideTable &t, int W, const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R,
                       const uint64_t *span_keys, const int32_t *span_meta, const int32_t *span_count, int32_t *span_taxon,
                       hipStream_t s) {
  if (R == 0) return;
  uint64_t blocks = std::min<uint64_t>((R + 3) / 4, 256 * 32);
  WIDE_DISPATCH(W, hipLaunchKernelGGL(wide_probe_kernel<WW>, dim3((unsigned)blocks), dim3(256), 0, s, t, offsets, mate_offsets, R,
                                      span_keys, span_meta, span_count, span_taxon));
}

}  // namespace slk

namespace slk {
void launch_wide_build_insert(const WideTable &t, int W, const int32_t *parents, int32_t ntax, const uint64_t *offsets, uint64_t R,
                              const uint64_t *span_keys, const int32_t *span_meta, const int32_t *span_count, const int32_t *chunk_taxon,
                              unsigned long long *counters, hipStream_t s) {
  if (R == 0) return;
  uint64_t blocks = std::min<uint64_t>((R + 3) / 4, 256 * 32);
  WIDE_DISPATCH(W, hipLaunchKernelGGL(wide_build_insert_kernel<WW>, dim3((unsigned)blocks), dim3(256), 0, s, t, parents, ntax, offsets, R,
                                      span_keys, span_meta, span_count, chunk_taxon, counters));
}
void launch_wide_export(const WideTable &t, int W, int64_t *keys, int32_t *taxa, uint64_t capacity, unsigned long long *counter, hipStream_t s) {
  uint64_t blocks = std::min<uint64_t>((t.mask + 256) / 256, 256 * 32);
  WIDE_DISPATCH(W, hipLaunchKernelGGL(wide_export_kernel<WW>, dim3((unsigned)blocks), dim3(256), 0, s, t, keys, taxa, capacity, counter));
}
void launch_wide_gather_spans(int W, const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R, const uint64_t *span_keys,
                              const int32_t *span_meta, const uint64_t *out_offsets, void *out, int64_t *out_keys, hipStream_t s) {
  if (R == 0) return;
  uint64_t blocks = std::min<uint64_t>((R + 3) / 4, 8192);
  WIDE_DISPATCH(W, hipLaunchKernelGGL(wide_gather_spans_kernel<WW>, dim3((unsigned)blocks), dim3(256), 0, s, offsets, mate_offsets, R,
                                      span_keys, span_meta, out_offsets, (WideSpanOut *)out, out_keys));
}
}  // namespace slk

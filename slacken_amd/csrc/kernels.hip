// kernels.hip -- CDNA4 (gfx950) kernels of the classify path, baseline generation ("v1": one lane per read for the
// scan and for the per-read LCA, one wave per read for the table probes).  Integer / byte work, HBM-bound: no MFMA.
//
// Stage <-> reference map (S/ = src/main/scala/com/jnpersson/ under /root/reference):
//   scan_kernel      KeyValueIndex.getSpans (S/slacken/KeyValueIndex.scala:163-173): Supermers.splitByAmbiguity /
//                    splitFragment / spans (S/slacken/Supermers.scala:49-125,150-189) over ShiftScanner.allMatches
//                    (S/kmers/minimizer/ShiftScanner.scala:90-159), SpacedSeed/RandomXOR priorities
//                    (S/kmers/minimizer/MinimizerPriorities.scala:144-179,282-321), PosRankWindow + MinSplitter.splitRead
//                    (PosRankWindow.scala:33-97, MinSplitter.scala:133-172) in their observable form: window minimum by
//                    value + run-length merge of equal minima (the equivalence is a property test under tests/).
//   probe_kernel     the left equi-join + spanToHit (S/slacken/Classifier.scala:84-88, KeyValueIndex.scala:176-185).
//   classify_kernel  Classifier.classify (Classifier.scala:439-454), TaxonCounts.toMap/totalKmers
//                    (S/slacken/TaxonCounts.scala:70-87), LowestCommonAncestor.apply/resolveTree
//                    (S/slacken/LowestCommonAncestor.scala:49-146), Taxonomy.hasAncestor (S/slacken/Taxonomy.scala:236-244).
#include "engine.h"

#include <algorithm>

namespace slk {

// ---------------------------------------------------------------------------------------------------------------
// table build / lookup
// ---------------------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) table_insert_kernel(TableBuild t, const int64_t *__restrict__ keys,
                                                           const int32_t *__restrict__ taxa, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  // The counters are summed per lane and added once per wave at the end: 1e10 atomics on one address would bound the build.
  int n_ins = 0, n_dup = 0, n_ovf = 0, max_d = 0;
  for (; i < n; i += stride) {
    int32_t taxon = taxa[i];
    if (taxon == 0) continue;  // a record with taxon NONE is indistinguishable from a miss
    uint64_t h = fmix64((uint64_t)keys[i]);
    if (!shard_keeps(t, h)) continue;  // table-sharded library: another rank's record
    uint32_t home;
    uint64_t rem_hi;
    table_slot(t.g, h, home, rem_hi);
    bool done = false;
    for (int d = 0; d <= t.disp_limit && !done; d++) {
      unsigned long long *bucket = (unsigned long long *)(t.cells + ((uint64_t)table_bucket(t.g, home, (uint32_t)d) * CELLS));
      uint64_t tag = rem_hi | (uint64_t)d;
      unsigned long long val = (tag << t.g.taxon_bits) | (uint32_t)taxon;
      unsigned long long first = 0;
      for (int c = 0; c < CELLS && !done; c++) {
        unsigned long long cur = __hip_atomic_load(&bucket[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
          unsigned long long old = atomicCAS(&bucket[c], 0ULL, val);
          if (old == 0) {
            done = true;
            n_ins++;
            max_d = max(max_d, d);
            break;
          }
          cur = old;
        }
        if (c == 0) first = cur;
        if (cell_tag(t.g, cur) == tag) {  // same key already present: contract violation, keep the first
          done = true;
          n_dup++;
        }
      }
      // the bucket is full and does not hold the key: the record goes on, and the bucket says so from now on
      if (!done && t.g.flag && !(first & t.g.flag)) atomicOr(&bucket[0], (unsigned long long)t.g.flag);
    }
    if (!done) n_ovf++;
  }
  for (int o = 32; o > 0; o >>= 1) {
    n_ins += __shfl_xor(n_ins, o);
    n_dup += __shfl_xor(n_dup, o);
    n_ovf += __shfl_xor(n_ovf, o);
    max_d = max(max_d, __shfl_xor(max_d, o));
  }
  if ((threadIdx.x & 63) == 0) {
    if (n_ins) atomicAdd(t.n_inserted, (unsigned long long)n_ins);
    if (n_dup) atomicAdd(t.n_duplicate, (unsigned long long)n_dup);
    if (n_ovf) atomicAdd(t.n_overflow, (unsigned long long)n_ovf);
    if (max_d) atomicMax(t.max_disp, max_d);
  }
}

// One bucket per probe step; returns the stored taxon or 0 (NONE).
__device__ __forceinline__ int32_t table_lookup(const TableView &t, uint64_t key) {
  uint64_t h = fmix64(key);
  uint32_t home;
  uint64_t rem_hi;
  table_slot(t.g, h, home, rem_hi);
  uint64_t tmask = (1ULL << t.g.taxon_bits) - 1;
  for (int d = 0; d <= t.max_disp; d++) {
    const ulonglong2 *b = (const ulonglong2 *)(t.cells + ((uint64_t)table_bucket(t.g, home, (uint32_t)d) * CELLS));
    uint64_t tag = rem_hi | (uint64_t)d;
    uint64_t cells[CELLS];
#pragma unroll
    for (int c = 0; c < LPB; c++) { ulonglong2 v = b[c]; cells[2 * c] = v.x; cells[2 * c + 1] = v.y; }
    bool has_empty = false;
    int32_t found = 0;
#pragma unroll
    for (int c = 0; c < CELLS; c++) {
      has_empty |= (cells[c] == 0);
      if (cells[c] != 0 && cell_tag(t.g, cells[c]) == tag) found = (int32_t)(cells[c] & tmask);
    }
    if (found) return found;
    if (has_empty) return 0;
    if (t.g.flag && !(cells[0] & t.g.flag)) return 0;  // full, but no record ever went past it
  }
  return 0;
}

__global__ void __launch_bounds__(256) table_lookup_kernel(TableView t, const int64_t *__restrict__ keys, uint64_t n,
                                                           int32_t *__restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = ext_taxon(t, table_lookup(t, (uint64_t)keys[i]));
}

// ---------------------------------------------------------------------------------------------------------------
// stage 1: scan (one lane per fragment)
// ---------------------------------------------------------------------------------------------------------------

// BitRepresentation.charToTwobitWithInvalid (S/kmers/util/BitRepresentation.scala:150-158): 0..3, 4 = whitespace, 5 = invalid
__device__ __forceinline__ int char_code(uint8_t c) {
  if (c == '\n' || c == '\r') return 4;
  switch (c | 0x20) {
    case 'a': return 0;
    case 'c': return 1;
    case 'g': return 2;
    case 't': return 3;
    case 'u': return 3;
    default: return 5;
  }
}

struct SpanWriter {
  uint64_t *keys;
  int32_t *meta;
  uint64_t base;
  int32_t n;
  bool first;       // Supermers.spans :72
  bool have_last;   // lastMinimizer = Array[Long]() initially (:73)
  uint64_t last;

  __device__ __forceinline__ void emit(uint64_t key, int32_t kmers, int32_t flag) {
    bool seqlike = (flag == 1);
    bool distinct = seqlike && (first || !(have_last && key == last));  // :84-86
    if (seqlike) { last = key; have_last = true; }                      // :88-90
    first = false;
    keys[base + n] = seqlike ? key : 0;
    meta[base + n] = pack_meta(kmers, flag, distinct ? 1 : 0);
    n++;
  }
};

// One mate: Supermers.splitFragment(NTSeq) :113-125 fused with splitByAmbiguity :150-178 and the m-mer scan.
// ring: this lane's last w keys, ring[slot * blockDim.x + threadIdx.x].
__device__ void scan_mate(const ScanParams &P, const uint8_t *__restrict__ seq, uint32_t n, uint64_t *ring, SpanWriter &out) {
  const int k = P.k, m = P.m, w = P.w;
  const uint32_t stride = blockDim.x;
  int run_class = 0;        // 1 = characters of [actguACTGU\n\r], 0 = anything else
  uint32_t run_len = 0;     // string length of the current run (incl. whitespace)
  uint32_t nvalid = 0;      // valid nucleotides in the current run
  uint64_t fwd = 0, rc = 0;
  int head = 0;             // ring slot of the newest key
  uint64_t minv = 0;        // minimum of the last <= w keys
  int minage = 0;           // pushes since the newest key equal to minv
  uint64_t cur_val = 0;     // value of the open super-mer
  int32_t cur_run = 0;      // its number of k-mer windows (0 = none open)

  for (uint32_t i = 0; i <= n; i++) {
    int t = 5, cls = -1;
    if (i < n) {
      t = char_code(seq[i]);
      cls = (t < 5) ? 1 : 0;
    }
    if (run_len > 0 && cls != run_class) {  // the run [.., i) ends
      if (run_class == 1 && nvalid >= (uint32_t)k) {
        out.emit(cur_val, cur_run, 1);      // last super-mer of a SEQUENCE_FLAG run
      } else if (run_len >= (uint32_t)k) {
        // AMBIGUOUS_FLAG run of string length >= k: ONE span with kmers = length - (k-1) (Supermers.scala:116-119)
        out.emit(0, (int32_t)run_len - (k - 1), 2);
      }                                     // runs shorter than k vanish (:116)
      run_len = 0;
    }
    if (i == n) break;
    if (run_len == 0) {
      run_class = cls;
      nvalid = 0; fwd = 0; rc = 0; head = w - 1; minage = 0; minv = ~0ULL; cur_run = 0;
    }
    run_len++;
    if (t < 4) {
      nvalid++;
      // left-aligned rolling m-mer (NTBitArray.shiftLongArrayKmerLeft, NTBitArray.scala:140-150) and its reverse complement
      fwd = (fwd << 2) | ((uint64_t)t << P.sh);
      rc = ((rc >> 2) | ((uint64_t)(3 - t) << 62)) & P.keep;
      if (nvalid >= (uint32_t)m) {
        uint64_t canon = (P.canonical && rc < fwd) ? rc : fwd;  // NTBitArray.writeCanonical :258-266 == unsigned min
        uint64_t key = (canon ^ P.xmask) & P.smask;            // RandomXOR then SpacedSeed (MinimizerPriorities.scala:165-175,308-312)
        head = (head + 1 == w) ? 0 : head + 1;
        ring[(uint32_t)head * stride + threadIdx.x] = key;
        if (key <= minv) { minv = key; minage = 0; }
        else if (++minage >= w) {  // the minimum left the window: rescan the last w keys (oldest first)
          int slot = (head + 1 == w) ? 0 : head + 1;
          minv = ~0ULL;
          for (int a = w - 1; a >= 0; a--) {
            uint64_t v = ring[(uint32_t)slot * stride + threadIdx.x];
            if (v <= minv) { minv = v; minage = a; }
            slot = (slot + 1 == w) ? 0 : slot + 1;
          }
        }
        if (nvalid >= (uint32_t)k) {  // one k-mer window is complete: its minimizer VALUE is minv
          if (cur_run == 0) { cur_val = minv; cur_run = 1; }
          else if (minv == cur_val) cur_run++;                 // MinSplitter.splitRead :154-158 (equal value, any position)
          else { out.emit(cur_val, cur_run, 1); cur_val = minv; cur_run = 1; }
        }
      }
    }
  }
}

__global__ void scan_kernel(ScanParams P, const uint8_t *__restrict__ bases, const uint64_t *__restrict__ offsets,
                            const uint8_t *__restrict__ mate_bases, const uint64_t *__restrict__ mate_offsets, uint64_t R,
                            uint64_t *__restrict__ span_keys, int32_t *__restrict__ span_meta,
                            int32_t *__restrict__ span_count) {
  extern __shared__ uint64_t ring[];
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  SpanWriter out;
  out.keys = span_keys; out.meta = span_meta;
  out.base = span_region(offsets, mate_offsets, r);
  out.n = 0; out.first = true; out.have_last = false; out.last = 0;
  uint64_t o0 = offsets[r];
  scan_mate(P, bases + o0, (uint32_t)(offsets[r + 1] - o0), ring, out);
  if (mate_bases) {
    out.emit(0, -(P.k - 1), 3);  // MATE_PAIR_BORDER pseudo-span: empty super-mer, kmers = 0 - (k-1) (Supermers.scala:53-57)
    uint64_t m0 = mate_offsets[r];
    scan_mate(P, mate_bases + m0, (uint32_t)(mate_offsets[r + 1] - m0), ring, out);
  }
  span_count[r] = out.n;
}

// ---------------------------------------------------------------------------------------------------------------
// stage 2: probe (one wave per fragment, one lane per span)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) probe_kernel(TableView T, const uint64_t *__restrict__ offsets,
                                                    const uint64_t *__restrict__ mate_offsets, uint64_t R,
                                                    const uint64_t *__restrict__ span_keys,
                                                    const int32_t *__restrict__ span_meta,
                                                    const int32_t *__restrict__ span_count,
                                                    int32_t *__restrict__ span_taxon) {
  const uint32_t lane = threadIdx.x & 63;
  uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t r = wave; r < R; r += nwaves) {
    uint64_t base = span_region(offsets, mate_offsets, r);
    int32_t n = span_count[r];
    for (int32_t j = lane; j < n; j += 64) {
      int32_t flag = meta_flag(span_meta[base + j]);
      int32_t taxon;
      // spanToHit (KeyValueIndex.scala:176-185): flag wins over any record; unmatched -> NONE
      if (flag == 2) taxon = -1;
      else if (flag == 3) taxon = -2;
      else taxon = ext_taxon(T, table_lookup(T, span_keys[base + j]));
      span_taxon[base + j] = taxon;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// stage 3: classify (one lane per fragment)
// ---------------------------------------------------------------------------------------------------------------
struct Tax {
  const int32_t *parents;
  const uint4 *nodes;   // {parent, tin, tout, -} per id from a depth-first tour (capi.hip: build_tax_nodes), or null
  int32_t T;
  __device__ __forceinline__ int32_t parent(int32_t t) const { return ((uint32_t)t < (uint32_t)T) ? parents[t] : 0; }
  __device__ __forceinline__ uint4 node(int32_t t) const {   // an id outside the taxonomy is a tree of its own
    return ((uint32_t)t < (uint32_t)T) ? nodes[t] : make_uint4(0u, 0x40000000u + (uint32_t)t, 0x40000000u + (uint32_t)t, 0u);
  }
};

// LowestCommonAncestor.apply :49-78 without the path buffer: first node on b's path that lies on a's path.
__device__ int32_t lca_pair(const Tax &tx, int32_t a, int32_t b) {
  if (a == 0 || b == 0) return b == 0 ? a : b;
  for (int32_t y = b; y != 0; y = tx.parent(y))
    for (int32_t x = a; x != 0; x = tx.parent(x))
      if (x == y) return y;
  return 1;  // ROOT
}

struct MapView {  // insertion-ordered taxon -> count map (Int2IntArrayMap) in this fragment's scratch slots
  int2 *e;
  int32_t n;
  __device__ __forceinline__ int32_t get(int32_t t) const {
    for (int32_t i = 0; i < n; i++) { int2 v = e[i]; if (v.x == t) return v.y; }
    return 0;
  }
};

__global__ void __launch_bounds__(256) classify_kernel(Tax tx, const uint64_t *__restrict__ offsets,
                                                       const uint64_t *__restrict__ mate_offsets, uint64_t R,
                                                       const int32_t *__restrict__ span_meta,
                                                       const int32_t *__restrict__ span_taxon,
                                                       const int32_t *__restrict__ span_count,
                                                       uint64_t *__restrict__ map_scratch, int32_t min_hit_groups,
                                                       Thresholds thr, int32_t C, uint64_t out_stride,
                                                       int32_t *__restrict__ out_taxon, uint8_t *__restrict__ out_classified,
                                                       int32_t *__restrict__ out_num_distinct,
                                                       int32_t *__restrict__ out_total_kmers,
                                                       int32_t *__restrict__ out_num_hits,
                                                       int32_t *__restrict__ out_num_probes) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  uint64_t base = span_region(offsets, mate_offsets, r);
  int32_t n = span_count[r];
  MapView map;
  map.e = (int2 *)(map_scratch + base);
  map.n = 0;
  int32_t total = 0, nd = 0, nprobe = 0;
  for (int32_t j = 0; j < n; j++) {
    int32_t meta = span_meta[base + j];
    int32_t taxon = span_taxon[base + j];
    if (meta_flag(meta) == 2) taxon = -1;       // spanToHit: the flag wins over any record (KeyValueIndex.scala:176-185)
    else if (meta_flag(meta) == 3) taxon = -2;
    int32_t count = meta_kmers(meta);
    if (taxon != -2) total += count;                         // TaxonCounts.totalKmers :84-87
    if (meta_distinct(meta) && taxon != 0) nd++;              // Classifier.scala:94 (distinct is false for flagged spans)
    if (taxon == -1 || taxon == -2) continue;                 // TaxonCounts.toMap :70-81
    nprobe++;
    int32_t i = 0;
    for (; i < map.n; i++) if (map.e[i].x == taxon) break;
    if (i == map.n) { map.e[i] = make_int2(taxon, count); map.n++; }
    else map.e[i].y += count;
  }

  if (tx.nodes != nullptr) {
    // resolveTree on the taxonomy's Euler-tour intervals (a is an ancestor-or-self of b iff tin[a] <= tin[b] <= tout[a]; lane.hip
    // and fused.hip do the same on their LDS maps): no walks of root paths.  The intervals are loaded again where
    // they are needed -- this map lives in HBM and has no room for them; the loads hit the L2.  NONE is kept in the map
    // (TaxonCounts.toMap) and is on no root path and in no clade.
    int32_t maxTaxon = 0, best = 0, sum_all = 0;
    uint32_t m_in = 0, m_out = 0;
    for (int32_t it = 0; it < map.n; it++) {           // step 1 (:101-123)
      const int2 a = map.e[it];
      if (a.x == 0) continue;
      sum_all += a.y;
      const uint4 na = tx.node(a.x);
      int32_t score = 0;
      for (int32_t j = 0; j < map.n; j++) {
        const int2 b = map.e[j];
        if (b.x == 0) continue;
        const uint4 nb = tx.node(b.x);
        score += (nb.y <= na.y && na.y <= nb.z) ? b.y : 0;
      }
      if (score > best) { maxTaxon = a.x; best = score; m_in = na.y; m_out = na.z; }
      else if (score == best) {                          // LowestCommonAncestor.apply :49-78 of (maxTaxon, a.x)
        if (m_in <= na.y && na.y <= m_out) {
        } else if (na.y <= m_in && m_in <= na.z) { maxTaxon = a.x; m_in = na.y; m_out = na.z; }
        else {
          int32_t x = (int32_t)tx.node(maxTaxon).x;
          uint4 nx = make_uint4(0, 0, 0, 0);
          while (x != 0) { nx = tx.node(x); if (nx.y <= na.y && na.y <= nx.z) break; x = (int32_t)nx.x; }
          if (x == 0) { x = 1; nx = tx.node(1); }
          maxTaxon = x; m_in = nx.y; m_out = nx.z;
        }
      }
    }
    for (int32_t c = 0; c < C; c++) {                    // step 2 (:125-144), jumping from map taxon to map taxon
      const double required = ceil(__dmul_rn(thr.v[c], (double)total));
      int32_t mt = maxTaxon;
      uint32_t cin = m_in, cout = m_out;
      uint4 cur = make_uint4(0, 0, 0, 0);
      bool have_cur = false;
      while (mt != 0) {
        int32_t ms = 0, up_taxon = 0;
        bool side = false;
        uint32_t up_in = 0, up_out = 0;
        for (int32_t j = 0; j < map.n; j++) {
          const int2 b = map.e[j];
          if (b.x == 0) continue;
          const uint4 nb = tx.node(b.x);
          const bool inside = cin <= nb.y && nb.y <= cout;
          const bool above = !inside && nb.y <= cin && cin <= nb.z;
          ms += inside ? b.y : 0;
          side = side || (!inside && !above);
          if (above && (up_taxon == 0 || nb.y > up_in)) { up_taxon = b.x; up_in = nb.y; up_out = nb.z; }
        }
        if ((double)ms >= required) break;
        if (ms == sum_all) { mt = 0; break; }
        if (!side) { mt = up_taxon; cin = up_in; cout = up_out; have_cur = false; }
        else {
          if (!have_cur) cur = tx.node(mt);
          mt = (int32_t)cur.x;
          if (mt != 0) { cur = tx.node(mt); have_cur = true; cin = cur.y; cout = cur.z; }
        }
      }
      const bool classified = (mt != 0) && (nd >= min_hit_groups);   // Classifier.scala:445
      out_taxon[(uint64_t)c * out_stride + r] = classified ? mt : 0;
      out_classified[(uint64_t)c * out_stride + r] = classified ? 1 : 0;
    }
    if (out_num_distinct) out_num_distinct[r] = nd;
    if (out_total_kmers) out_total_kmers[r] = total;
    if (out_num_hits) out_num_hits[r] = n;
    if (out_num_probes) out_num_probes[r] = nprobe;
    return;
  }
  // (no Euler tour -- a taxonomy of more than 2^26 ids: the reference's walks)
  // resolveTree step 1 (:101-123): LCA of all taxa with the maximal root-path score; threshold independent
  int32_t maxTaxon = 0, maxScore = 0;
  for (int32_t it = 0; it < map.n; it++) {
    int32_t taxon = map.e[it].x;
    int32_t score = 0;
    for (int32_t node = taxon; node != 0; node = tx.parent(node)) score += map.get(node);
    if (score > maxScore) { maxTaxon = taxon; maxScore = score; }
    else if (score == maxScore) maxTaxon = lca_pair(tx, maxTaxon, taxon);
  }

  for (int32_t c = 0; c < C; c++) {
    // Math.ceil(confidenceThreshold * totalKmers) in binary64 (:94)
    double required = ceil(__dmul_rn(thr.v[c], (double)total));
    int32_t mt = maxTaxon;
    int32_t ms = map.get(mt);                                // :125
    while (mt != 0 && (double)ms < required) {               // :126-144
      ms = 0;
      for (int32_t it = 0; it < map.n; it++) {
        int2 v = map.e[it];
        bool in_clade = false;                               // Taxonomy.hasAncestor(v.x, mt) :236-244
        for (int32_t x = v.x; x != 0; x = tx.parent(x)) if (x == mt) { in_clade = true; break; }
        if (in_clade) ms += v.y;
      }
      if ((double)ms >= required) break;
      mt = tx.parent(mt);
    }
    bool classified = (mt != 0) && (nd >= min_hit_groups);   // Classifier.scala:445
    out_taxon[(uint64_t)c * out_stride + r] = classified ? mt : 0;
    out_classified[(uint64_t)c * out_stride + r] = classified ? 1 : 0;
  }
  if (out_num_distinct) out_num_distinct[r] = nd;
  if (out_total_kmers) out_total_kmers[r] = total;
  if (out_num_hits) out_num_hits[r] = n;
  if (out_num_probes) out_num_probes[r] = nprobe;
}

// ---------------------------------------------------------------------------------------------------------------
// compaction of the sparse per-read regions for the host-facing outputs
// ---------------------------------------------------------------------------------------------------------------
struct SpanOut { int64_t key; int32_t kmers; int8_t flag; uint8_t distinct; uint16_t pad; };
struct HitOut { int32_t taxon; int32_t count; };

__global__ void __launch_bounds__(256) gather_spans_kernel(const uint64_t *__restrict__ offsets,
                                                           const uint64_t *__restrict__ mate_offsets, uint64_t R,
                                                           const uint64_t *__restrict__ span_keys,
                                                           const int32_t *__restrict__ span_meta,
                                                           const uint64_t *__restrict__ out_offsets,
                                                           SpanOut *__restrict__ out) {
  const uint32_t lane = threadIdx.x & 63;
  uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t r = wave; r < R; r += nwaves) {
    uint64_t base = span_region(offsets, mate_offsets, r);
    uint64_t o = out_offsets[r];
    int32_t n = (int32_t)(out_offsets[r + 1] - o);
    for (int32_t j = lane; j < n; j += 64) {
      int32_t meta = span_meta[base + j];
      SpanOut s;
      s.key = (int64_t)span_keys[base + j];
      s.kmers = meta_kmers(meta);
      s.flag = (int8_t)meta_flag(meta);
      s.distinct = (uint8_t)meta_distinct(meta);
      s.pad = 0;
      out[o + j] = s;
    }
  }
}

__global__ void __launch_bounds__(256) gather_hits_kernel(const uint64_t *__restrict__ offsets,
                                                          const uint64_t *__restrict__ mate_offsets, uint64_t R,
                                                          const int32_t *__restrict__ span_meta,
                                                          const int32_t *__restrict__ span_taxon,
                                                          const uint64_t *__restrict__ out_offsets,
                                                          HitOut *__restrict__ out) {
  const uint32_t lane = threadIdx.x & 63;
  uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t r = wave; r < R; r += nwaves) {
    uint64_t base = span_region(offsets, mate_offsets, r);
    uint64_t o = out_offsets[r];
    int32_t n = (int32_t)(out_offsets[r + 1] - o);
    for (int32_t j = lane; j < n; j += 64) {
      HitOut h;
      h.taxon = span_taxon[base + j];
      h.count = meta_kmers(span_meta[base + j]);
      out[o + j] = h;
    }
  }
}

// The hit lists as TaxonCounts.fromHits merges them (S/slacken/TaxonCounts.scala:31-48): adjacent entries of one taxon become one entry
// with the sum of their k-mer counts -- what ClassifiedRead.outputLine prints (pairsInOrderString :94-110), and a fifth to a tenth of
// the bytes the un-merged lists send over the link (slk_stream_set_merged_hits).  A wave per fragment, 64 entries at a time; the run
// that is open at the end of a 64 is carried (taxon, sum) and written when the next 64 close it or the fragment ends.
// COUNT: merged[r] = entries of fragment r's merged list; otherwise the entries go to out[moffs[r] ..).
template <bool COUNT>
__global__ void __launch_bounds__(256) merged_hits_kernel(const uint64_t *__restrict__ offsets, const uint64_t *__restrict__ mate_offsets, uint64_t R,
                                                          const int32_t *__restrict__ span_meta, const int32_t *__restrict__ span_taxon,
                                                          const int32_t *__restrict__ nh, const uint64_t *__restrict__ moffs,
                                                          int32_t *__restrict__ merged, HitOut *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t r = wave; r < R; r += nwaves) {
    const uint64_t base = span_region(offsets, mate_offsets, r);
    const int32_t n = nh[r];
    const uint64_t o = COUNT ? 0 : moffs[r];
    int32_t carry_t = 0, carry_c = 0;
    bool have_carry = false;
    uint32_t runs = 0;   // runs begun so far; the open one is run runs - 1
    for (int32_t j0 = 0; j0 < n; j0 += 64) {
      const int32_t j = j0 + lane;
      const bool valid = j < n;
      const int32_t t = valid ? span_taxon[base + j] : 0;
      int32_t left = __shfl_up(t, 1);
      if (lane == 0) left = carry_t;
      const bool start = valid && ((lane == 0 && !have_carry) || t != left);
      const uint64_t S = __ballot(start);
      const int nvalid = min(64, n - j0);
      const int nstart = __popcll(S);
      const int last = nvalid - 1;
      if (COUNT) {
        runs += (uint32_t)nstart;
        have_carry = true;
        carry_t = __shfl(t, last);
        continue;
      }
      const int32_t c = valid ? meta_kmers(span_meta[base + j]) : 0;
      int32_t P = c;   // inclusive prefix sums of the counts over the lanes
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int32_t q = __shfl_up(P, d);
        if (lane >= d) P += q;
      }
      const int first = S ? __ffsll((long long)S) - 1 : nvalid;   // lanes [0, first) continue the carried run
      if (first > 0) carry_c += __shfl(P, first - 1);
      if (S != 0) {
        if (have_carry && lane == 0) { HitOut h; h.taxon = carry_t; h.count = carry_c; out[o + runs - 1] = h; }
        // a run begun at lane i ends before the next start (or with the 64): its sum from the prefix sums
        const uint64_t later = lane < 63 ? (S >> (lane + 1)) : 0;
        const int e = later ? lane + 1 + (__ffsll((long long)later) - 1) : nvalid;
        const int32_t Pe = __shfl(P, e - 1), Pb = __shfl_up(P, 1);
        const int32_t seg = Pe - (lane == 0 ? 0 : Pb);
        const int L = 63 - __clzll((long long)S);                 // the last start: its run stays open
        if (start && lane != L) {
          HitOut h; h.taxon = t; h.count = seg;
          out[o + runs + (uint32_t)__popcll(S & ((1ULL << lane) - 1))] = h;
        }
        carry_t = __shfl(t, L);
        carry_c = __shfl(seg, L);
        runs += (uint32_t)nstart;
        have_carry = true;
      }
    }
    if (COUNT) { if (lane == 0) merged[r] = (int32_t)runs; }
    else if (have_carry && lane == 0) { HitOut h; h.taxon = carry_t; h.count = carry_c; out[o + runs - 1] = h; }
  }
}

// ---- launchers (called from capi.hip) ----
// slk_index_finalize, dense taxon ids: the taxon field of every cell becomes to_dense[taxon] (cells whose taxon has no dense
// id -- not a node of the taxonomy -- are counted; the caller first counts, and rewrites only if there are none)
__global__ void __launch_bounds__(256) remap_cells_kernel(uint64_t *__restrict__ cells, uint64_t ncells, int32_t taxon_bits,
                                                          const int32_t *__restrict__ to_dense, int32_t n_to_dense,
                                                          unsigned long long *__restrict__ undefined, bool apply) {
  const uint64_t tmask = (1ULL << taxon_bits) - 1;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  int bad = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncells; i += step) {
    const uint64_t c = cells[i];
    if (c == 0) continue;
    const uint64_t t = c & tmask;   // (the bits above the taxon field -- tag, bucket flag -- stay as they are)
    const int32_t d = t < (uint64_t)n_to_dense ? to_dense[t] : 0;
    if (d == 0) bad++;
    else if (apply) cells[i] = (c & ~tmask) | (uint64_t)(uint32_t)d;
  }
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(undefined, (unsigned long long)bad);
}
void launch_remap_cells(uint64_t *cells, uint64_t ncells, int32_t taxon_bits, const int32_t *to_dense, int32_t n_to_dense,
                        unsigned long long *undefined, bool apply, hipStream_t s) {
  if (ncells == 0) return;
  uint64_t blocks = std::min<uint64_t>((ncells + 255) / 256, 256 * 64);
  hipLaunchKernelGGL(remap_cells_kernel, dim3((unsigned)blocks), dim3(256), 0, s, cells, ncells, taxon_bits, to_dense, n_to_dense,
                     undefined, apply);
}

void launch_table_insert(const TableBuild &t, const int64_t *keys, const int32_t *taxa, uint64_t n, hipStream_t s) {
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) return;
  hipLaunchKernelGGL(table_insert_kernel, dim3((unsigned)blocks), dim3(256), 0, s, t, keys, taxa, n);
}
void launch_table_lookup(const TableView &t, const int64_t *keys, uint64_t n, int32_t *out, hipStream_t s) {
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) return;
  hipLaunchKernelGGL(table_lookup_kernel, dim3((unsigned)blocks), dim3(256), 0, s, t, keys, n, out);
}
void launch_scan(const ScanParams &P, const uint8_t *bases, const uint64_t *offsets, const uint8_t *mate_bases,
                 const uint64_t *mate_offsets, uint64_t R, uint64_t *span_keys, int32_t *span_meta, int32_t *span_count,
                 hipStream_t s) {
  if (R == 0) return;
  // one lane per fragment; the w-key ring lives in LDS: blockDim * w * 8 bytes <= 64 KiB
  int block = 256;
  while (block > 64 && (size_t)block * P.w * 8 > 65536) block >>= 1;
  size_t lds = (size_t)block * P.w * 8;
  uint64_t blocks = (R + block - 1) / block;
  hipLaunchKernelGGL(scan_kernel, dim3((unsigned)blocks), dim3(block), lds, s, P, bases, offsets, mate_bases,
                     mate_offsets, R, span_keys, span_meta, span_count);
}
void launch_probe(const TableView &T, const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R,
                  const uint64_t *span_keys, const int32_t *span_meta, const int32_t *span_count, int32_t *span_taxon,
                  hipStream_t s) {
  if (R == 0) return;
  uint64_t blocks = (R + 3) / 4;  // 4 waves per block, one fragment per wave per iteration
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(probe_kernel, dim3((unsigned)blocks), dim3(256), 0, s, T, offsets, mate_offsets, R, span_keys,
                     span_meta, span_count, span_taxon);
}
void launch_classify(const int32_t *parents, const uint4 *nodes, int32_t T, const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R,
                     const int32_t *span_meta, const int32_t *span_taxon, const int32_t *span_count,
                     uint64_t *map_scratch, int32_t min_hit_groups, const Thresholds &thr, int32_t C, uint64_t out_stride,
                     int32_t *out_taxon, uint8_t *out_classified, int32_t *out_num_distinct, int32_t *out_total_kmers,
                     int32_t *out_num_hits, int32_t *out_num_probes, hipStream_t s) {
  if (R == 0) return;
  Tax tx{parents, nodes, T};
  uint64_t blocks = (R + 255) / 256;
  hipLaunchKernelGGL(classify_kernel, dim3((unsigned)blocks), dim3(256), 0, s, tx, offsets, mate_offsets, R, span_meta,
                     span_taxon, span_count, map_scratch, min_hit_groups, thr, C, out_stride, out_taxon, out_classified,
                     out_num_distinct, out_total_kmers, out_num_hits, out_num_probes);
}
void launch_gather_spans(const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R, const uint64_t *span_keys,
                         const int32_t *span_meta, const uint64_t *out_offsets, void *out, hipStream_t s) {
  if (R == 0) return;
  uint64_t blocks = (R + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gather_spans_kernel, dim3((unsigned)blocks), dim3(256), 0, s, offsets, mate_offsets, R, span_keys,
                     span_meta, out_offsets, (SpanOut *)out);
}
// Exclusive prefix sums of the per-fragment span / hit counts (the offsets of the caller's lists) on the device: chunk sums, their
// scan by one block, the fill.  (The counts used to go down, be summed by one host thread and go up again as offsets: 3.4 ms of a
// 4 M-read call's 49.)
constexpr int SCAN_T = 256, SCAN_PER = 8, SCAN_CHUNK = SCAN_T * SCAN_PER;
__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t *lds /* SCAN_T / 64 words */, uint64_t &block_total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up((unsigned long long)incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  uint64_t before = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_T / 64; w++) { if (w < wave) before += lds[w]; tot += lds[w]; }
  block_total = tot;
  __syncthreads();
  return before + incl - v;
}
__global__ void __launch_bounds__(SCAN_T) scan_sums_kernel(const int32_t *__restrict__ counts, uint64_t n, uint64_t *__restrict__ sums) {
  __shared__ uint64_t lds[SCAN_T / 64];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_PER;
  uint64_t v = 0;
#pragma unroll
  for (int j = 0; j < SCAN_PER; j++) if (base + j < n) v += (uint64_t)(uint32_t)counts[base + j];
  uint64_t tot;
  (void)block_exclusive_scan(v, lds, tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(SCAN_T) scan_chunks_kernel(uint64_t *__restrict__ sums, uint64_t nb) {   // one block: sums -> their exclusive scan, sums[nb] = all
  __shared__ uint64_t lds[SCAN_T / 64];
  uint64_t carry = 0;
  for (uint64_t b0 = 0; b0 < nb; b0 += SCAN_T) {
    const uint64_t i = b0 + threadIdx.x;
    const uint64_t v = i < nb ? sums[i] : 0;
    uint64_t tot;
    const uint64_t ex = block_exclusive_scan(v, lds, tot);
    if (i < nb) sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) sums[nb] = carry;
}
__global__ void __launch_bounds__(SCAN_T) scan_fill_kernel(const int32_t *__restrict__ counts, uint64_t n, const uint64_t *__restrict__ sums,
                                                           uint64_t *__restrict__ out) {
  __shared__ uint64_t lds[SCAN_T / 64];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_PER;
  uint32_t c[SCAN_PER];
  uint64_t v = 0;
#pragma unroll
  for (int j = 0; j < SCAN_PER; j++) { c[j] = base + j < n ? (uint32_t)counts[base + j] : 0u; v += c[j]; }
  uint64_t tot;
  uint64_t at = sums[blockIdx.x] + block_exclusive_scan(v, lds, tot);
#pragma unroll
  for (int j = 0; j < SCAN_PER; j++) { if (base + j < n) out[base + j] = at; at += c[j]; }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = sums[gridDim.x];
}
void launch_counts_to_offsets(const int32_t *counts, uint64_t n, uint64_t *out, uint64_t *tmp, hipStream_t s) {
  const uint64_t nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;   // (n >= 1)
  hipLaunchKernelGGL(scan_sums_kernel, dim3((unsigned)nb), dim3(SCAN_T), 0, s, counts, n, tmp);
  hipLaunchKernelGGL(scan_chunks_kernel, dim3(1), dim3(SCAN_T), 0, s, tmp, nb);
  hipLaunchKernelGGL(scan_fill_kernel, dim3((unsigned)nb), dim3(SCAN_T), 0, s, counts, n, tmp, out);
}

void launch_merged_hits(bool count, const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R, const int32_t *span_meta, const int32_t *span_taxon,
                        const int32_t *nh, const uint64_t *moffs, int32_t *merged, void *out, hipStream_t s) {
  if (R == 0) return;
  const uint64_t blocks = std::min<uint64_t>((R + 3) / 4, 8192);
  if (count) hipLaunchKernelGGL(merged_hits_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, offsets, mate_offsets, R, span_meta, span_taxon, nh, moffs, merged, (HitOut *)out);
  else hipLaunchKernelGGL(merged_hits_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, offsets, mate_offsets, R, span_meta, span_taxon, nh, moffs, merged, (HitOut *)out);
}
void launch_gather_hits(const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R, const int32_t *span_meta,
                        const int32_t *span_taxon, const uint64_t *out_offsets, void *out, hipStream_t s) {
  if (R == 0) return;
  uint64_t blocks = (R + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gather_hits_kernel, dim3((unsigned)blocks), dim3(256), 0, s, offsets, mate_offsets, R, span_meta,
                     span_taxon, out_offsets, (HitOut *)out);
}

// Reads that arrive in the engine's 3-bit form (slk_classify_batch_packed; host/pack.hpp): back to one character per base -- "ACGT"
// by code where the validity bit is set, 'N' elsewhere -- so that every kernel of the classify path reads them as it reads ASCII.
// One thread per word of 16 bases, one 16-byte store; words [w0, w1).  The classify kernels see valid / invalid and the code of a
// base, nothing else of a character, so the results are those of the original text.  Streaming: 6 B in, 16 B out per word.
__global__ void __launch_bounds__(256) unpack_bases_kernel(const uint32_t *__restrict__ codes, const uint16_t *__restrict__ valid, uint64_t w0,
                                                           uint64_t w1, uint8_t *__restrict__ out) {
  for (uint64_t w = w0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < w1; w += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = codes[w];
    const uint32_t v = valid[w];
    uint32_t o[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
      uint32_t x = 0;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int j = 4 * d + b;
        const uint32_t code = (c >> (2 * j)) & 3u;
        const uint32_t ch = ((v >> j) & 1u) ? ((0x54474341u >> (8 * code)) & 0xFFu) : (uint32_t)'N';   // "ACGT"
        x |= ch << (8 * b);
      }
      o[d] = x;
    }
    *(uint4 *)(out + w * 16) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
// The wave pass's work list: the hand-on lists of its long fragments, longest class first, then list 5, one behind the other
// (engine.h: FusedArgs.hand_hdr).  A few hundred thousand entries at most; the counts are only known on the device.
__global__ void order_wave_list_kernel(unsigned long long *hdr, uint32_t *lists, uint64_t stride, uint64_t long_cap) {
  uint64_t start[HandOn::WAVE_CLASSES + 2];   // of list WAVE0 + j (j = WAVE_CLASSES: list REST) in the output
  start[0] = 0;
#pragma unroll
  for (int j = 0; j < HandOn::WAVE_CLASSES; j++) start[j + 1] = start[j] + hdr[HandOn::N_WAVE0 + j];
  start[HandOn::WAVE_CLASSES + 1] = start[HandOn::WAVE_CLASSES] + hdr[HandOn::REST];
  const uint64_t n = start[HandOn::WAVE_CLASSES + 1];
  uint32_t *const out = lists + HandOn::ordered_at(stride, long_cap);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    int j = 0;
#pragma unroll
    for (int q = 1; q <= HandOn::WAVE_CLASSES; q++) j += i >= start[q];
    const uint64_t from = HandOn::list_at(j < HandOn::WAVE_CLASSES ? HandOn::WAVE0 + j : (int)HandOn::REST, stride, long_cap);
    uint64_t st = 0;
#pragma unroll
    for (int q = 0; q <= HandOn::WAVE_CLASSES; q++) if (q == j) st = start[q];
    out[i] = lists[from + (i - st)];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr[HandOn::ORDERED] = n;
}
void launch_order_wave_list(const FusedArgs &A, hipStream_t s) {
  order_wave_list_kernel<<<256, 256, 0, s>>>(A.hand_hdr, A.hand_lists, A.hand_stride, A.hand_long_cap);
}

void launch_unpack_bases(const uint32_t *codes, const uint16_t *valid, uint64_t w0, uint64_t w1, uint8_t *out, hipStream_t s) {
  if (w1 <= w0) return;
  const uint64_t blocks = std::min<uint64_t>((w1 - w0 + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(unpack_bases_kernel, dim3((unsigned)blocks), dim3(256), 0, s, codes, valid, w0, w1, out);
}

}  // namespace slk

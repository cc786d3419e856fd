// fused.hip -- ONE WAVEFRONT PER FRAGMENT (the lane-per-fragment kernel of lane.hip takes the short fragments; this one takes
// what that one hands back -- fragments over 1000 bases or with more than 12 distinct taxa -- and the span output; long
// unpaired fragments go on to segment_kernel below, one lane per segment): scan -> probe -> per-read LCA fused in one launch, no
// HBM intermediates (spans, hits and the per-read taxon map live in registers / LDS).  gfx950, wave64.  Integer/byte
// work bounded by random 64-byte HBM probes: no MFMA.
//
//   scan   2-bit packing of 256 bases per step (coalesced dword loads, SWAR code/pack) into TWO LDS rings, the bases and
//          their reverse complement (mirrored), so that every lane extracts its forward and reverse m-mer with one
//          funnel shift each; canonical/XOR/space mask; width-w window minimum by DPP lane shifts; run-length merge of
//          equal minima by ballot + bit scans.
//          Reference: KeyValueIndex.getSpans (S/slacken/KeyValueIndex.scala:163-173) = Supermers.splitFragment/spans
//          (S/slacken/Supermers.scala:49-125) over MinSplitter.splitEncode (S/kmers/minimizer/MinSplitter.scala:98-172),
//          ShiftScanner.allMatches (ShiftScanner.scala:90-159), RandomXOR/SpacedSeed (MinimizerPriorities.scala:144-321).
//   probe  4 lanes read one 64-byte bucket (4 x 16 B, one HBM line per probe), 16 probes per wave instruction, all four
//          instructions' loads in flight before the first compare.  Reference: the left join + spanToHit
//          (S/slacken/Classifier.scala:84-88, KeyValueIndex.scala:176-185).
//   LCA    a fragment whose hits name ONE taxon (the common case) is resolved without touching the tree; otherwise the
//          hits are folded into a 128-slot LDS hash map (taxon -> k-mer count) and resolveTree runs with one lane per
//          distinct taxon.  Reference: TaxonCounts.toMap/totalKmers (S/slacken/TaxonCounts.scala:70-87),
//          LowestCommonAncestor.apply/resolveTree (S/slacken/LowestCommonAncestor.scala:49-146), Classifier.classify
//          (Classifier.scala:439-454).
// Control flow per fragment: a producer state machine (START -> FAST rounds over the mate, or over its runs of valid
// characters one after the other (RUNS) when it holds others -> NEXT mate) fills the LDS span buffer; ONE flush site probes and
// folds whatever is buffered; repeat until the fragment is exhausted.
#include "engine.h"

#include <cstdlib>

namespace slk {

constexpr int FW = 4;          // waves (fragments in flight) per block
constexpr int SPAN_CAP = 128;  // buffered spans per wave before a flush
constexpr int MAP_CAP = 128;   // taxon map slots per wave (power of two)
constexpr int32_t MAP_EMPTY = -1;  // AMBIGUOUS_SPAN is never inserted, so -1 is free

// the wave-per-fragment scan's staged bases (fused_kernel only: the segment kernel's lanes stream their bases themselves)
struct __attribute__((aligned(16))) RingLds {
  uint32_t fwd_ring[32 + 4];     // 2-bit bases, MSB first inside each dword, ring of 512 bases (two 256-base blocks); words 32..34
                                 // repeat words 0..2, so that the three words an m-mer straddles are read without wrapping around
  uint32_t rc_ring[32 + 4];      // complement of base p at ring position 511 - p
};
struct __attribute__((aligned(16))) WaveLds {
  uint64_t span_key[SPAN_CAP];
  int32_t span_meta[SPAN_CAP];
  uint64_t stash[128];           // probe: (bucket byte offset, tag) per span of the chunk; resolveTree: dense (taxon,count)
  int32_t result[64];
  int32_t map_key[MAP_CAP];
  int32_t map_cnt[MAP_CAP];
};

// ---- wave helpers ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() {  // order this wave's LDS traffic (lanes of one wave exchange data via LDS)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
  uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}
// lane i <- lane i+1 / lane i-1 (DPP wave shifts; the vacated end lane reads 0 -- bound_ctrl, so that no register has to be
// zeroed for it beforehand)
__device__ __forceinline__ uint64_t from_next(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x130, 0xF, 0xF, true);
  uint32_t hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x130, 0xF, 0xF, true);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t from_prev(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x138, 0xF, 0xF, true);
  uint32_t hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, true);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int src) {  // src wave-uniform; result wave-uniform
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t bcast64(uint64_t v) {  // lane 0 -> all, result is wave-uniform
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ int lanes_below(uint64_t mask) {  // popcount(mask & lanes lower than this one)
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// the next unit of a hand-on list for this wave (one atomic per wave; every lane gets the value)
__device__ __forceinline__ uint64_t next_unit(unsigned long long *counter) {
  unsigned long long t = 0;
  if ((threadIdx.x & 63) == 0) t = atomicAdd(counter, 1ULL);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)t), hi = __builtin_amdgcn_readfirstlane((uint32_t)(t >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// ---- staging: 256 characters -> 2-bit codes in the two rings ----------------------------------------------------------
// returns true in lanes that saw a character outside ACGTUacgtu (BitRepresentation.isValid, BitRepresentation.scala:140-143)
__device__ __forceinline__ bool stage_block(RingLds *L, const uint8_t *seq, uint32_t n, uint32_t blk, int lane) {
  uint32_t pos = blk * 256 + lane * 4;
  uint32_t nchars = pos < n ? min(4u, n - pos) : 0u;
  uint32_t v = 0x41414141u;  // missing characters read as 'A' (never used by a window, never "bad")
  if (nchars == 4) {
    __builtin_memcpy(&v, seq + pos, 4);  // unaligned dword load
  } else if (nchars > 0) {               // the fragment's last 1-3 characters: byte loads, nothing past the end is touched
    uint32_t t = 0;
    for (uint32_t j = 0; j < nchars; j++) t |= (uint32_t)seq[pos + j] << (8 * j);
    v = (v & ~((1u << (8 * nchars)) - 1)) | t;
  }
  // valid <=> (c & 0xC0) == 0x40 and (c & 0x1F) in {1,3,7,20,21}  (A C G T U, either case)
  const uint32_t VM = (1u << 1) | (1u << 3) | (1u << 7) | (1u << 20) | (1u << 21);
  uint32_t okbits = (VM >> (v & 31)) & (VM >> ((v >> 8) & 31)) & (VM >> ((v >> 16) & 31)) & (VM >> ((v >> 24) & 31)) & 1u;
  bool bad = ((v & 0xC0C0C0C0u) != 0x40404040u) || okbits == 0;
  // (c >> 1) & 3 maps A,C,T/U,G (either case) to 0,1,2,3; x ^ (x >> 1) turns that into A=0 C=1 G=2 T=3
  uint32_t t = (v >> 1) & 0x03030303u;
  t ^= (t >> 1) & 0x01010101u;
  uint32_t pack = (t * 0x40100401u) >> 24;  // code0<<6 | code1<<4 | code2<<2 | code3
  uint32_t g = (blk * 64 + lane) & 127;     // group of 4 bases within the 512-base ring
  const uint32_t fi = (g & ~3u) | (3u - (g & 3u));                        // MSB-first inside each dword
  ((uint8_t *)L->fwd_ring)[fi] = (uint8_t)pack;
  if (fi < 12u) ((uint8_t *)L->fwd_ring)[fi + 128u] = (uint8_t)pack;      // (the ring's first three words again behind its end)
  // reverse complement of the group: complement, reverse the four 2-bit codes
  uint32_t x = __brev(~pack) >> 24;                         // bit-reversed byte
  x = ((x >> 1) & 0x55u) | ((x & 0x55u) << 1);              // swap the bits of each pair back
  uint32_t gr = 127 - g;
  const uint32_t ri = (gr & ~3u) | (3u - (gr & 3u));
  ((uint8_t *)L->rc_ring)[ri] = (uint8_t)x;
  if (ri < 12u) ((uint8_t *)L->rc_ring)[ri + 128u] = (uint8_t)x;
  return bad;
}

// 64 bits of a ring starting at base position p (mod 512), left-aligned
__device__ __forceinline__ uint64_t ring_bits(const uint32_t *ring, uint32_t p) {
  const uint32_t *w = ring + ((p >> 4) & 31);   // (words 32..34 repeat 0..2: no wrap-around inside the three)
  const uint32_t s = (p & 15) * 2;
  const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
  uint32_t hi = (uint32_t)(((((uint64_t)w0 << 32) | w1) << s) >> 32);
  uint32_t lo = (uint32_t)(((((uint64_t)w1 << 32) | w2) << s) >> 32);
  return ((uint64_t)hi << 32) | lo;
}

// left-aligned key of the m-mer starting at base q (NTBitArray layout; SpacedSeed(RandomXOR) priority)
// (keep / xmask / smask: the splitter's masks, which the caller holds in VECTOR registers although they are wave-uniform -- the
//  kernel's scalar registers are oversubscribed, and as scalars these six were spilled and read back lane by lane every round)
__device__ __forceinline__ uint64_t key_at(const RingLds *L, const ScanParams &P, uint32_t q, uint64_t keep, uint64_t xmask, uint64_t smask) {
  uint64_t fwd = ring_bits(L->fwd_ring, q) & keep;
  uint64_t canon = fwd;
  if (P.canonical) {
    uint64_t rc = ring_bits(L->rc_ring, 512u - ((q + P.m) & 511u)) & keep;
    canon = umin64(fwd, rc);  // NTBitArray.writeCanonical :258-266 == unsigned minimum of the two orientations
  }
  return (canon ^ xmask) & smask;
}

__device__ __forceinline__ void put_span(WaveLds *L, int slot, uint64_t key, int32_t kmers, int32_t flag, bool distinct) {
  L->span_key[slot] = key;
  L->span_meta[slot] = pack_meta(kmers, flag, distinct ? 1 : 0);
}

// ---- probe + fold ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void map_insert(WaveLds *L, int32_t taxon, int32_t count, int32_t *status) {
  uint32_t slot = ((uint32_t)taxon * 0x9E3779B1u) >> 25;
  int probe = 0;
  for (; probe < MAP_CAP; probe++) {
    int32_t old = atomicCAS(&L->map_key[slot], MAP_EMPTY, taxon);
    if (old == MAP_EMPTY || old == taxon) {
      atomicAdd(&L->map_cnt[slot], count);
      break;
    }
    slot = (slot + 1) & (MAP_CAP - 1);
  }
  if (probe == MAP_CAP) atomicOr(status, 1);
}
__device__ __forceinline__ int32_t map_get(const WaveLds *L, int32_t taxon) {
  uint32_t slot = ((uint32_t)taxon * 0x9E3779B1u) >> 25;
  for (int probe = 0; probe < MAP_CAP; probe++) {
    int32_t kk = L->map_key[slot];
    if (kk == taxon) return L->map_cnt[slot];
    if (kk == MAP_EMPTY) return 0;
    slot = (slot + 1) & (MAP_CAP - 1);
  }
  return 0;
}

// Look up the (<= 64) buffered spans [s0, s0+cnt) and return this lane's taxon (lane l <-> span s0+l).  The access shape of
// lane.hip's probe_batch: LPB lanes read one bucket together (LPB x 16 B), 64 / LPB probes per wave instruction, all LPB
// instructions in flight before the first compare.
__device__ __forceinline__ int32_t probe_chunk(WaveLds *L, const TableView &T, int s0, int cnt, int lane, int32_t meta) {
  constexpr int PG = 64 / LPB;
  const uint64_t NO_TAG = ~0ULL;  // a real tag has at most 64 - taxon_bits significant bits
  uint64_t key = (lane < cnt) ? L->span_key[s0 + lane] : 0;
  int32_t flag = meta_flag(meta);
  bool seq = (lane < cnt) && flag == 1;
  uint64_t h = fmix64(key);
  uint32_t home;
  uint64_t rem_hi;
  table_slot(T.g, h, home, rem_hi);
  ulonglong2 st;
  st.x = home;                                                     // home bucket
  st.y = seq ? rem_hi : NO_TAG;                                    // tag at displacement 0
  ((ulonglong2 *)L->stash)[lane] = st;
  L->result[lane] = 0;
  wave_sync();
  const int g = lane / LPB, c = lane % LPB;                        // 64 / LPB groups of LPB lanes
  const uint64_t tmask = (1ULL << T.g.taxon_bits) - 1;
  const uint32_t gmask = (1u << LPB) - 1;
  const char *cellbase = (const char *)T.cells + c * 16;
  uint32_t more = 0;  // bit s: my group's span s*PG+g overflowed its home bucket and is still unresolved
  {
    ulonglong2 cell[LPB];
#pragma unroll
    for (int s = 0; s < LPB; s++) {
      ulonglong2 e = ((const ulonglong2 *)L->stash)[s * PG + g];
      cell[s] = make_ulonglong2(0, 0);
      if (e.y != NO_TAG) cell[s] = *(const ulonglong2 *)(cellbase + (e.x << BUCKET_SHIFT));
    }
#pragma unroll
    for (int s = 0; s < LPB; s++) {
      const uint64_t want = L->stash[2 * (s * PG + g) + 1];
      const bool act = want != NO_TAG;
      const bool e0 = cell[s].x == 0, e1 = cell[s].y == 0;
      const bool m0 = act && !e0 && cell_tag(T.g, cell[s].x) == want;
      const bool m1 = act && !e1 && cell_tag(T.g, cell[s].y) == want;
      if (m0 || m1) L->result[s * PG + g] = (int32_t)((m0 ? cell[s].x : cell[s].y) & tmask);
      // a group is resolved once one of its lanes matched or saw an empty cell (cells are never freed), or when the bucket is
      // full but no record ever went past it (the flag in its first cell)
      const bool closed = T.g.flag != 0 && c == 0 && (cell[s].x & T.g.flag) == 0;
      const uint64_t B = __ballot(m0 || m1 || e0 || e1 || !act || closed);
      if (((B >> (g * LPB)) & gmask) == 0) more |= 1u << s;
    }
  }
  if (__ballot(more != 0) != 0) {  // rare: bucket-level linear probing
    for (int d = 1; d <= T.max_disp && __ballot(more != 0) != 0; d++) {
      for (int s = 0; s < LPB; s++) {
        const bool act = (more >> s) & 1;
        ulonglong2 cl = make_ulonglong2(0, 0);
        uint64_t want = 0;
        if (act) {
          ulonglong2 e = ((const ulonglong2 *)L->stash)[s * PG + g];
          want = e.y + (uint64_t)d;
          cl = *(const ulonglong2 *)(cellbase + ((uint64_t)table_bucket(T.g, (uint32_t)e.x, (uint32_t)d) << BUCKET_SHIFT));
        }
        const bool e0 = cl.x == 0, e1 = cl.y == 0;
        const bool m0 = act && !e0 && cell_tag(T.g, cl.x) == want;
        const bool m1 = act && !e1 && cell_tag(T.g, cl.y) == want;
        if (m0 || m1) L->result[s * PG + g] = (int32_t)((m0 ? cl.x : cl.y) & tmask);
        const bool closed = act && T.g.flag != 0 && c == 0 && (cl.x & T.g.flag) == 0;
        const uint64_t B = __ballot(act && (m0 || m1 || e0 || e1 || closed));
        if (act && ((B >> (g * LPB)) & gmask) != 0) more &= ~(1u << s);
      }
    }
  }
  wave_sync();
  int32_t taxon = L->result[lane];
  if (flag == 2) taxon = -1;       // spanToHit: the flag wins over any record (KeyValueIndex.scala:176-185)
  else if (flag == 3) taxon = -2;
  return taxon;
}

__device__ __forceinline__ int char_code2(uint8_t c) {  // 0..3 nucleotide, 5 anything else (whitespace is not expected here)
  uint32_t x = ((uint32_t)(c & 0xDF)) - 0x41;
  switch (x) {
    case 0: return 0;
    case 2: return 1;
    case 6: return 2;
    case 0x13: return 3;
    case 0x14: return 3;
    default: return 5;
  }
}

// LowestCommonAncestor.apply :49-78 (wave-uniform arguments and control flow)
__device__ int32_t lca_uniform(const int32_t *parents, int32_t ntax, int32_t a, int32_t b) {
  if (a == 0 || b == 0) return b == 0 ? a : b;
  for (int32_t y = b; y != 0; y = ((uint32_t)y < (uint32_t)ntax) ? parents[y] : 0)
    for (int32_t x = a; x != 0; x = ((uint32_t)x < (uint32_t)ntax) ? parents[x] : 0)
      if (x == y) return y;
  return 1;
}

// {parent, tin, tout, -} of taxon t (engine.h: FusedArgs.nodes); wave-uniform t: a scalar load.  An id outside the taxonomy is a
// tree of its own.
__device__ __forceinline__ uint4 tax_node(const FusedArgs &A, int32_t t) {
  return ((uint32_t)t < (uint32_t)A.ntax) ? A.nodes[t] : make_uint4(0u, 0x40000000u + (uint32_t)t, 0x40000000u + (uint32_t)t, 0u);
}

// resolveTree over the LDS map on the taxonomy's Euler-tour intervals (lane.hip has the same for its 12-slot maps):
// one lane per distinct taxon.  Each loads its taxon's interval once; a taxon's root-path score is the count of the entries whose
// interval holds its tin, a candidate's clade sum that of the entries whose tin its interval holds; the confidence walk jumps to
// the nearest map taxon above the candidate when nothing else is left outside its clade, and ends when the clade holds the whole
// map.  No walk of parent pointers except towards a tie's LCA and past side branches.  (The intervals live in the span buffer's
// key array, empty by now.)
__device__ __forceinline__ void resolve_map_intervals(WaveLds *L, const FusedArgs &A, uint64_t r, int lane, int32_t total, int32_t nd,
                                                      const int2 *dense, int D) {
  uint32_t *const tin = (uint32_t *)L->span_key, *const tout = tin + MAP_CAP;
  static_assert(SPAN_CAP * sizeof(uint64_t) >= 2 * MAP_CAP * sizeof(uint32_t), "the intervals alias the span keys");
  for (int b0 = 0; b0 < D; b0 += 64) {
    const int i = b0 + lane;
    if (i < D) {
      const uint4 n = tax_node(A, dense[i].x);
      tin[i] = n.y; tout[i] = n.z;
    }
  }
  wave_sync();
  // step 1 (:101-123): the LCA of the taxa with the maximal root-path score
  int32_t maxTaxon = 0, best = 0, sum_all = 0;
  uint32_t m_in = 0, m_out = 0;
  for (int b0 = 0; b0 < D; b0 += 64) {
    const int i = b0 + lane;
    const int32_t t = i < D ? dense[i].x : 0;
    const bool act = t != 0;   // (the map of this kernel keeps NONE, as TaxonCounts.toMap does: it is on no root path and in no clade)
    const uint32_t ain = act ? tin[i] : 0u, aout = act ? tout[i] : 0u;
    int32_t score = 0;
    for (int j = 0; j < D; j++) score += (act && tin[j] <= ain && ain <= tout[j]) ? dense[j].y : 0;   // (uniform j: LDS broadcasts)
    sum_all += wave_sum(act ? dense[i].y : 0);
    const int32_t mx = wave_max(act ? score : -1);
    if (mx > best) { best = mx; maxTaxon = 0; }
    if (mx == best && best > 0) {
      uint64_t tie = __ballot(act && score == best && t != 0);
      while (tie) {
        const int bl = __builtin_ctzll(tie);
        tie &= tie - 1;
        const int32_t tt = __builtin_amdgcn_readlane(t, bl);
        const uint32_t tin_t = __builtin_amdgcn_readlane(ain, bl), tout_t = __builtin_amdgcn_readlane(aout, bl);
        if (maxTaxon == 0 || (tin_t <= m_in && m_in <= tout_t)) {   // LowestCommonAncestor.apply :49-78 by intervals
          maxTaxon = tt; m_in = tin_t; m_out = tout_t;
        } else if (!(m_in <= tin_t && tin_t <= m_out)) {          // neither holds the other: the first node above that holds tt
          int32_t x = (int32_t)tax_node(A, maxTaxon).x;
          uint4 nx = make_uint4(0, 0, 0, 0);
          while (x != 0) {
            nx = tax_node(A, x);
            if (nx.y <= tin_t && tin_t <= nx.z) break;
            x = (int32_t)nx.x;
          }
          if (x == 0) { x = 1; nx = tax_node(A, 1); }             // no common node: ROOT (:77)
          maxTaxon = x; m_in = nx.y; m_out = nx.z;
        }
      }
    }
  }
  for (int32_t c = 0; c < A.C; c++) {
    const double required = ceil(__dmul_rn(A.thr.v[c], (double)total));  // Math.ceil(confidence * totalKmers) :94
    int32_t mt = maxTaxon;
    uint32_t cin = m_in, cout = m_out;
    uint4 cur = make_uint4(0, 0, 0, 0);
    bool have_cur = false;
    while (mt != 0) {  // :125-144
      int32_t sum = 0;
      bool side = false;
      uint32_t up_in = 0;
      for (int b0 = 0; b0 < D; b0 += 64) {
        const int i = b0 + lane;
        if (i < D && dense[i].x != 0) {
          const uint32_t jin = tin[i], jout = tout[i];
          const bool inside = cin <= jin && jin <= cout;
          const bool above = !inside && jin <= cin && cin <= jout;
          sum += inside ? dense[i].y : 0;
          side = side || (!inside && !above);
          if (above && jin > up_in) up_in = jin;   // (deeper on the candidate's root path = later in the tour)
        }
      }
      const int32_t ms = wave_sum(sum);
      if ((double)ms >= required) break;
      if (ms == sum_all) { mt = 0; break; }        // the clade holds the whole map: no ancestor can do better
      if (__ballot(side) == 0) {
        // everything left lies above the candidate, on its root path: the next clade that differs is the nearest of them
        const uint32_t nearest = (uint32_t)wave_max((int)up_in);
        for (int b0 = 0; b0 < D; b0 += 64) {
          const int i = b0 + lane;
          if (i < D && dense[i].x != 0 && tin[i] == nearest) { L->result[0] = dense[i].x; L->result[1] = (int32_t)tout[i]; }
        }
        wave_sync();
        mt = L->result[0]; cin = nearest; cout = (uint32_t)L->result[1];
        have_cur = false;
        wave_sync();
      } else {
        if (!have_cur) cur = tax_node(A, mt);
        mt = (int32_t)cur.x;                       // Taxonomy.parents
        if (mt != 0) { cur = tax_node(A, mt); have_cur = true; cin = cur.y; cout = cur.z; }
      }
    }
    const bool classified = (mt != 0) && (nd >= A.min_hit_groups);  // Classifier.scala:445
    if (lane == 0) {
      A.out_taxon[(uint64_t)c * A.out_stride + r] = classified ? ext_taxon(A.T, mt) : 0;
      A.out_classified[(uint64_t)c * A.out_stride + r] = classified ? 1 : 0;
    }
  }
}

// resolveTree over the LDS map (the general case: at least two distinct non-NONE taxa)
__device__ __forceinline__ void resolve_map(WaveLds *L, const FusedArgs &A, uint64_t r, int lane, int32_t total, int32_t nd) {
  int2 *dense = (int2 *)L->stash;
  int D = 0;
#pragma unroll
  for (int half = 0; half < 2; half++) {
    int slot = lane + 64 * half;
    int32_t kk = L->map_key[slot];
    bool occ = kk != MAP_EMPTY;
    uint64_t mask = __ballot(occ);
    if (occ) dense[D + lanes_below(mask)] = make_int2(kk, L->map_cnt[slot]);
    D += __popcll(mask);
  }
  wave_sync();
  if (A.nodes != nullptr) { resolve_map_intervals(L, A, r, lane, total, nd, dense, D); return; }
  // (no Euler tour: a taxonomy of more than 2^22 ids whose records could not be renumbered -- the walks of the reference)
  // step 1 (:101-123): LCA of the taxa with the maximal root-path score
  int32_t maxTaxon = 0, best = 0;
  for (int b0 = 0; b0 < D; b0 += 64) {
    int i = b0 + lane;
    bool act = i < D;
    int32_t t = act ? dense[i].x : 0;
    int32_t score = 0;
    for (int32_t node = t; node != 0; node = ((uint32_t)node < (uint32_t)A.ntax) ? A.parents[node] : 0)
      score += map_get(L, node);
    int32_t mx = wave_max(act ? score : -1);
    if (mx > best) { best = mx; maxTaxon = 0; }
    if (mx == best && best > 0) {
      uint64_t tie = __ballot(act && score == best && t != 0);
      while (tie) {
        int bl = __builtin_ctzll(tie);
        tie &= tie - 1;
        maxTaxon = lca_uniform(A.parents, A.ntax, maxTaxon, __builtin_amdgcn_readlane(t, bl));
      }
    }
  }
  for (int32_t c = 0; c < A.C; c++) {
    double required = ceil(__dmul_rn(A.thr.v[c], (double)total));  // Math.ceil(confidence * totalKmers) :94
    int32_t mt = maxTaxon;
    int32_t ms = map_get(L, mt);  // :125
    while (mt != 0 && (double)ms < required) {  // :126-144
      int32_t sum = 0;
      for (int b0 = 0; b0 < D; b0 += 64) {
        int i = b0 + lane;
        if (i < D) {
          int2 e = dense[i];
          for (int32_t x = e.x; x != 0; x = ((uint32_t)x < (uint32_t)A.ntax) ? A.parents[x] : 0)
            if (x == mt) { sum += e.y; break; }  // Taxonomy.hasAncestor :236-244
        }
      }
      ms = wave_sum(sum);
      if ((double)ms >= required) break;
      mt = ((uint32_t)mt < (uint32_t)A.ntax) ? A.parents[mt] : 0;
    }
    bool classified = (mt != 0) && (nd >= A.min_hit_groups);  // Classifier.scala:445
    if (lane == 0) {
      A.out_taxon[(uint64_t)c * A.out_stride + r] = classified ? ext_taxon(A.T, mt) : 0;
      A.out_classified[(uint64_t)c * A.out_stride + r] = classified ? 1 : 0;
    }
  }
}

enum { PH_START = 0, PH_FAST = 1, PH_RUNS = 2, PH_NEXT = 3 };

// SLK_WPS: waves per SIMD the register allocator must leave room for (k blocks of 256 threads per CU <=> k waves/SIMD)
#ifndef SLK_WPS
#define SLK_WPS 0
#endif
#if SLK_WPS > 0
#define FUSED_BOUNDS __launch_bounds__(FW * 64, SLK_WPS)
#else
#define FUSED_BOUNDS __launch_bounds__(FW * 64)
#endif

// W5: the window of the default splitter (k = 35, m = 31: five m-mers), its minimum as a fixed network of lane shifts
template <int MODE, bool W5>
__global__ void FUSED_BOUNDS fused_kernel(FusedArgs A) {
  __shared__ WaveLds lds[FW];
  __shared__ RingLds rings[FW];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps per-read state in SGPRs
  WaveLds *L = &lds[wib];
  RingLds *G = &rings[wib];
  const ScanParams &P = A.P;
  uint64_t v_keep = P.keep, v_xmask = P.xmask, v_smask = P.smask;
  asm volatile("" : "+v"(v_keep), "+v"(v_xmask), "+v"(v_smask));   // (vector registers: see key_at)
  const int w = W5 ? 5 : P.w;
  const uint32_t STEP = 64 - (w - 1);  // k-mer windows resolved per round
  const uint64_t nwaves = (uint64_t)gridDim.x * FW;
  // Work list: every fragment, or (work_list) the fragments the lane kernel deferred -- it appends them to the list itself --
  // one per wave iteration.  (They used to be found 64 flags at a time by ballot, each wave working through its 64 one after
  // the other: with most fragments deferred -- a batch of long reads -- that left most of the chip idle.)
  const uint64_t nunits = A.work_list ? (uint64_t)*A.work_count : A.R;
  for (uint64_t unit = (uint64_t)blockIdx.x * FW + wib, it = 0;; it++) {
    // a hand-on list: every unit is work and they differ a hundredfold in length, so the (fixed) grid draws them from a counter --
    // from its second unit on: a wave's first is the one of its own number, so that a list shorter than the grid (the usual case:
    // an empty one) costs no atomics (8 192 waves drawing from one address took 0.28 ms to find an empty list empty)
    if (it) unit = A.work_draw ? nwaves + next_unit(A.work_draw) : unit + nwaves;
    if (unit >= nunits) break;
   {
    const uint64_t r = A.work_list ? (uint64_t)A.work_list[unit] : unit;
    // ---- per-fragment state (wave-uniform unless noted) ----
    int nbuf = 0, n_out = 0;
    bool first = true, have_last = false;   // Supermers.spans :72-73
    uint64_t last_key = 0;
    const uint64_t base = span_region(A.offsets, A.mate_offsets, r);
    int32_t total = 0, nd = 0, np = 0, t0 = 0;
    bool map_mode = false;
    int32_t acc_t0 = 0, acc_none = 0;       // per-lane partial sums while !map_mode
    if (MODE != MODE_SPANS) {
      L->map_key[lane] = MAP_EMPTY; L->map_key[lane + 64] = MAP_EMPTY;
      L->map_cnt[lane] = 0; L->map_cnt[lane + 64] = 0;
    }
    int mate = 0, phase = PH_START;
    bool done = false;
    const uint8_t *seq = nullptr;
    uint32_t n = 0;
    // fast-path state: the stretch of valid characters being scanned is [.., fast_n) -- the whole mate, or one of its runs
    uint32_t i0 = 0, nwin = 0, staged = 0, blk = 0, fast_n = 0;
    bool runs_mode = false;   // the mate holds characters outside ACGTU: it is taken run by run (PH_RUNS)
    uint32_t rpos = 0;        // ... and this is where the next run starts
    uint64_t carry_val = 0;
    int carry_run = 0;
    bool seg_first = true;

    while (true) {
      // ================= producer: fill the span buffer =================
      while (!done && nbuf <= SPAN_CAP - 66) {
        // (wave-uniform state, said so: the compiler's analysis loses that through the state machine's merges and then runs the
        //  machine as divergent control flow, with lane masks and register copies at every join)
        phase = __builtin_amdgcn_readfirstlane(phase); nbuf = __builtin_amdgcn_readfirstlane(nbuf);
        carry_run = __builtin_amdgcn_readfirstlane(carry_run);
        i0 = __builtin_amdgcn_readfirstlane(i0); nwin = __builtin_amdgcn_readfirstlane(nwin);
        staged = __builtin_amdgcn_readfirstlane(staged); blk = __builtin_amdgcn_readfirstlane(blk);
        fast_n = __builtin_amdgcn_readfirstlane(fast_n); n = __builtin_amdgcn_readfirstlane(n);
        if (phase == PH_START) {
          if (mate == 0) { uint64_t o = A.offsets[r]; seq = A.bases + o; n = (uint32_t)(A.offsets[r + 1] - o); }
          else { uint64_t o = A.mate_offsets[r]; seq = A.mate_bases + o; n = (uint32_t)(A.mate_offsets[r + 1] - o); }
          // pre-scan for characters outside ACGTUacgtu; blocks are staged in descending order, so the ring's two halves
          // end up holding blocks 0 and 1
          bool bad = false;
          uint32_t nblk = (n + 255) / 256;
          for (uint32_t b = nblk; b-- > 0;) bad |= stage_block(G, seq, n, b, lane);
          if (__ballot(bad) != 0) {
            // Supermers.splitByAmbiguity :150-178: the mate is a sequence of runs of valid and of other characters; a valid run
            // of >= k characters is scanned like a whole clean mate, another run of >= k characters is one ambiguous span
            runs_mode = true;
            rpos = 0;
            staged = min(nblk, 2u) * 256; blk = staged / 256;
            phase = PH_RUNS;
          } else if (n >= (uint32_t)P.k) {
            runs_mode = false;
            fast_n = n;
            nwin = n - P.k + 1;
            total += (int32_t)nwin;  // the super-mers of a run partition its k-mer windows
            i0 = 0; staged = min(nblk, 2u) * 256; blk = staged / 256;
            carry_val = 0; carry_run = 0; seg_first = true;
            phase = PH_FAST;
          } else {
            phase = PH_NEXT;  // an all-valid mate shorter than k yields nothing (Supermers.scala:116)
          }
        } else if (phase == PH_FAST) {
         // rounds of STEP windows, one after the other while the span buffer has room (a loop of its own: the state machine around
         // it is not consulted between two rounds of one stretch)
         do {
          if (i0 >= nwin) {  // the segment's last super-mer
            if (lane == 0)
              put_span(L, nbuf, carry_val, carry_run, 1, seg_first ? (first || !have_last || carry_val != last_key) : true);
            nbuf += 1;
            first = false; have_last = true; last_key = carry_val;
            phase = runs_mode ? PH_RUNS : PH_NEXT;
            break;
          }
          uint32_t need = min(n, i0 + 64 + P.m - 1);
          while (staged < need) { stage_block(G, seq, n, blk, lane); blk++; staged += 256; }
          wave_sync();
          uint32_t q = i0 + lane;
          uint64_t key = (q + P.m <= fast_n) ? key_at(G, P, q, v_keep, v_xmask, v_smask) : ~0ULL;
          // minimum over lanes [l, l+w): doubling, then one overlapping step (PosRankWindow's observable result)
          uint64_t res;
          if (W5) {
            const uint64_t c2 = umin64(key, from_next(key));                  // lanes l, l + 1
            const uint64_t c4 = umin64(c2, from_next(from_next(c2)));         // l .. l + 3
            res = umin64(c4, from_next(c4));                                  // l .. l + 4
          } else {
            uint64_t cur = key;
            int covered = 1;
            if (w >= 2) { cur = umin64(cur, from_next(cur)); covered = 2; }
            for (; covered * 2 <= w; covered *= 2) cur = umin64(cur, shfl64(cur, lane + covered));
            res = cur;
            if (covered < w) {
              int d = w - covered;
              res = umin64(cur, d == 1 ? from_next(cur) : shfl64(cur, lane + d));
            }
          }
          int nw = (int)min(STEP, nwin - i0);
          i0 += STEP;
          uint64_t prev = from_prev(res);
          bool is_start = (lane < nw) && (lane == 0 ? (carry_run == 0 || res != carry_val) : (res != prev));
          uint64_t S = __ballot(is_start);
          if (S == 0) { carry_run += nw; continue; }  // MinSplitter.splitRead :154-158: equal value => same super-mer
          int firstl = __builtin_ctzll(S), lastl = 63 - __builtin_clzll(S);
          bool special = first || !have_last;         // distinct test of the segment's first span (Supermers.spans :84-86)
          int nclose = 0;
          if (carry_run > 0) {
            if (lane == 0)
              put_span(L, nbuf, carry_val, carry_run + firstl, 1, seg_first ? (special || carry_val != last_key) : true);
            nclose = 1;
            seg_first = false;
          }
          if (is_start && lane != lastl) {
            int d = __builtin_ctzll(S >> (lane + 1)) + 1;
            bool dist = (seg_first && lane == firstl) ? (special || res != last_key) : true;
            put_span(L, nbuf + nclose + lanes_below(S), res, d, 1, dist);
          }
          int emitted = nclose + __popcll(S) - 1;
          if (emitted > 0) { seg_first = false; first = false; }
          nbuf += emitted;
          carry_val = readlane64(res, lastl);
          carry_run = nw - lastl;
         } while (nbuf <= SPAN_CAP - 66);
        } else if (phase == PH_RUNS) {
          if (rpos >= n) { phase = PH_NEXT; continue; }
          // the run that starts at rpos: its class and its end (512 characters per step, 8 per lane)
          const uint32_t a = rpos;
          const bool valid_run = char_code2(seq[a]) < 4;
          uint32_t b = n;
          for (uint32_t p = a; p < n; p += 512) {
            const uint32_t q = p + 8 * lane;
            uint64_t chunk = 0;
            if (q < n) __builtin_memcpy(&chunk, seq + q, 8);  // (readable: 16 bytes of padding follow the last read)
            int firstdiff = 8;
#pragma unroll
            for (int j = 7; j >= 0; j--) {
              const bool v = char_code2((uint8_t)(chunk >> (8 * j))) < 4;
              if (q + j < n && v != valid_run) firstdiff = j;
            }
            const uint64_t D = __ballot(firstdiff < 8);
            if (D != 0) {
              const int fl = __builtin_ctzll(D);
              b = p + 8 * fl + (uint32_t)__builtin_amdgcn_readlane(firstdiff, fl);
              break;
            }
          }
          rpos = b;
          const uint32_t len = b - a;
          if (valid_run) {
            if (len >= (uint32_t)P.k) {
              fast_n = b;
              nwin = b - P.k + 1;            // (window starts are absolute positions in the mate)
              total += (int32_t)(len - P.k + 1);
              i0 = a;
              if (a / 256 + 2 < blk || a / 256 >= blk) { blk = a / 256; staged = blk * 256; }  // not in the ring: stage from a's block
              carry_val = 0; carry_run = 0; seg_first = true;
              phase = PH_FAST;
            }
          } else if (len >= (uint32_t)P.k) {
            if (lane == 0) put_span(L, nbuf, 0, (int32_t)len - (P.k - 1), 2, false);  // Supermers.scala:116-119
            nbuf += 1;
            total += (int32_t)len - (P.k - 1);
            first = false;
          }
        } else {  // PH_NEXT
          if (mate == 0 && A.mate_bases) {
            if (lane == 0) put_span(L, nbuf, 0, -(P.k - 1), 3, false);  // MATE_PAIR_BORDER (Supermers.scala:53-57)
            nbuf += 1;
            first = false;
            mate = 1;
            phase = PH_START;
          } else {
            done = true;
          }
        }
      }

      // ================= consumer: the one flush site =================
      wave_sync();
      for (int s0 = 0; s0 < nbuf; s0 += 64) {
        int cnt = min(64, nbuf - s0);
        bool in = lane < cnt;
        if (MODE == MODE_SPANS) {
          if (in) {
            A.span_keys[base + n_out + lane] = L->span_key[s0 + lane];
            A.span_meta[base + n_out + lane] = L->span_meta[s0 + lane];
          }
        } else {
          int32_t meta = in ? L->span_meta[s0 + lane] : 0;
          int32_t taxon = probe_chunk(L, A.T, s0, cnt, lane, meta);
          int32_t count = meta_kmers(meta);
          bool real = in && taxon >= 0;                                           // TaxonCounts.toMap :70-81 keeps these
          nd += __popcll(__ballot(in && meta_distinct(meta) && taxon != 0));      // Classifier.scala:94
          np += __popcll(__ballot(real));
          if (!map_mode) {
            uint64_t nz = __ballot(real && taxon != 0);
            if (nz != 0) {
              if (t0 == 0) t0 = __builtin_amdgcn_readlane(taxon, __builtin_ctzll(nz));
              if (__ballot(real && taxon != 0 && taxon != t0) != 0) {
                // a second taxon: move the single-taxon summary into the LDS map and continue there
                int32_t c0 = wave_sum(acc_t0), cn = wave_sum(acc_none);
                if (lane == 0 && c0 != 0) map_insert(L, t0, c0, A.status);
                if (lane == 1 && cn != 0) map_insert(L, 0, cn, A.status);
                map_mode = true;
              }
            }
          }
          if (map_mode) {
            if (real) map_insert(L, taxon, count, A.status);
          } else if (real) {
            if (taxon != 0) acc_t0 += count;
            else acc_none += count;
          }
          if (MODE == MODE_HITS && in) {
            A.span_meta[base + n_out + lane] = meta;
            A.span_taxon[base + n_out + lane] = ext_taxon(A.T, taxon);
          }
        }
        n_out += cnt;
      }
      nbuf = 0;
      wave_sync();
      if (done) break;
    }

    if (MODE != MODE_CLASSIFY) {
      if (lane == 0) A.span_count[r] = n_out;
    }
    if (MODE == MODE_SPANS) { wave_sync(); continue; }

    // ---- per-read classification -------------------------------------------------------------------------------------
    if (map_mode) {
      resolve_map(L, A, r, lane, total, nd);
    } else {
      // Every non-NONE hit names t0 (or there is none): resolveTree's first loop yields t0; lifting it changes nothing
      // because NONE is in no clade, so the result is t0 iff its own k-mer count reaches the required score (:125-146).
      bool need_count = false;
      for (int32_t c = 0; c < A.C; c++) need_count |= A.thr.v[c] > 0.0;
      int32_t c0 = (need_count && t0 != 0) ? wave_sum(acc_t0) : 0;
      for (int32_t c = 0; c < A.C; c++) {
        double required = ceil(__dmul_rn(A.thr.v[c], (double)total));
        int32_t mt = (t0 != 0 && !((double)c0 < required)) ? t0 : 0;
        bool classified = (mt != 0) && (nd >= A.min_hit_groups);
        if (lane == 0) {
          A.out_taxon[(uint64_t)c * A.out_stride + r] = classified ? ext_taxon(A.T, mt) : 0;
          A.out_classified[(uint64_t)c * A.out_stride + r] = classified ? 1 : 0;
        }
      }
    }
    if (lane == 0) {
      if (A.out_nd) A.out_nd[r] = nd;
      if (A.out_tk) A.out_tk[r] = total;
      if (A.out_nh) A.out_nh[r] = n_out;
      if (A.out_np) A.out_np[r] = np;
    }
    wave_sync();
   }
  }
}


// ---- long unpaired fragments: ONE WAVE PER FRAGMENT, ONE LANE PER SEGMENT ---------------------------------------------------
// The wave-wide scan above spends a few hundred instructions per 60 windows; a lane that walks its own stretch of the read
// with the lane kernel's state machine (lane.hip) needs a fraction of that per base, and 64 of them run side by side.  The
// fragment's k-mer windows are cut into 64 contiguous segments (of at least 64 windows), lane j scans the bases of segment j
// (its windows + k - 1) and pushes its super-mers into the wave's span buffer; probing, the taxon map and resolveTree are the
// wave kernel's.  Everything the classification needs is additive over windows -- a window's minimizer depends on its own
// bases only, super-mers and ambiguous spans partition the windows of their runs (Supermers.scala:116-119) -- except what
// looks across a segment border, which is settled at the end from what each lane saw at its two ends:
//   * `distinct` of a segment's first super-mer (Supermers.spans :84-90) compares with the last super-mer BEFORE it, which
//     another lane produced: it is counted as distinct first and taken back if the keys turn out equal;
//   * a super-mer or an ambiguous span cut by a border was counted twice in the number of spans (out_nh).
// Window width 5 (the register window of lane.hip).  The wave's span buffer does not keep the spans in order; for the HIT LISTS
// (HITS: TaxonHit per span in ordinal order, TaxonCounts.scala:94-121 is formatted from them) every span therefore carries its
// place -- lane j's i-th span goes to slot w0(j) + i of the fragment's span region, which a lane cannot overrun: a span holds at
// least one of its windows -- in a scratch copy of the region (A.span_keys, 8 bytes per slot: taxon, meta), and once the borders
// are settled the lanes' stretches are moved up against each other into span_taxon / span_meta: lane j's entries start at the
// sum of the earlier lanes' counts less their merges, and an entry that a border had cut in two (a super-mer, an ambiguous span)
// adds its k-mers to the last entry before it.
constexpr int SEG_SBLK = 5;       // 16-byte sub-blocks fetched per refill of a lane's read stream (as lane.hip)
constexpr uint32_t SEG_MIN_WINDOWS = 64;
struct __attribute__((aligned(16))) SegLds {
  uint4 sbuf[(SEG_SBLK - 1) * 64];
  uint64_t first_key[64], last_key[64];
  uint32_t flags[64];
};
enum { SEGF_HAS = 1, SEGF_ENDS_OPEN = 2, SEGF_END_AMB = 4 };
// HITS: a lane's spans get their taxa in the wave's probe chunks, out of any order a memory system would like -- consecutive entries
// of a chunk belong to different lanes, i.e. to places of the span region far apart: written where they belong one by one they were
// four scattered requests per span, and the hit lists came out at 32-47 Gbp/s where the wave kernel's run at 43-75.  So every lane
// collects ITS entries (taxon, meta: 8 bytes) in a queue of its own in LDS, in span order, and writes them to its stretch of the
// scratch region eight at a time: 64 contiguous bytes.  A lane that gets more than the queue holds within one chunk (one segment
// producing nearly all spans of a stretch of the fragment) has the surplus written directly, and empties its queue after the chunk.
constexpr int SEG_QL = 8;
template <bool HITS> struct SegHitLds {};
template <> struct __attribute__((aligned(16))) SegHitLds<true> {
  uint2 q[SEG_QL][64];        // [entry number mod SEG_QL][lane]
  uint32_t span_dst[SPAN_CAP];  // whose the buffered span is (lane << 26) and its number among that lane's spans
  uint32_t delivered[64];     // entries of the lane that have their taxon (they arrive in order)
  uint32_t flushed[64];       // entries of the lane that are in the scratch region
};

// (room: bytes from seq to the end of the caller's buffer -- the buffer's last block is assembled from byte loads)
__device__ __forceinline__ uint4 seg_refill(SegLds *G, int lane, const uint8_t *seq, uint32_t p, uint32_t n, uint32_t room) {
  uint4 v[SEG_SBLK];
#pragma unroll
  for (int i = 0; i < SEG_SBLK; i++) {
    v[i] = make_uint4(0, 0, 0, 0);
    if (p + 16u * i < n) v[i] = load_block16(seq + p + 16u * i, room - (p + 16u * i));
  }
#pragma unroll
  for (int i = 1; i < SEG_SBLK; i++) G->sbuf[(i - 1) * 64 + lane] = v[i];
  return v[0];
}

// A lane writes the entries of its queue that have arrived to its stretch of the scratch region (`mine`): whole groups of SEG_QG
// consecutive entries (32 contiguous bytes) as long as the queue did not overflow in the last chunk; everything, entry by entry, when
// it did (the surplus is in the region already) or at the end (ALL).
constexpr uint32_t SEG_QG = 4;
template <bool ALL>
__device__ __forceinline__ void seg_flush_queue(SegHitLds<true> *Q, uint2 *mine, int lane) {
  const uint32_t D = Q->delivered[lane];
  uint32_t F = Q->flushed[lane];
  if (ALL || D > F + (uint32_t)SEG_QL) {
    const uint32_t upto = D < F + (uint32_t)SEG_QL ? D : F + (uint32_t)SEG_QL;
    for (uint32_t j = F; j < upto; j++) mine[j] = Q->q[j % SEG_QL][lane];
    F = D;
  } else {
    while (D - F >= SEG_QG) {
      uint2 e[SEG_QG];
#pragma unroll
      for (uint32_t j = 0; j < SEG_QG; j++) e[j] = Q->q[(F + j) % SEG_QL][lane];
#pragma unroll
      for (uint32_t j = 0; j < SEG_QG; j += 2) *(uint4 *)&mine[F + j] = make_uint4(e[j].x, e[j].y, e[j + 1].x, e[j + 1].y);
      F += SEG_QG;
    }
  }
  Q->flushed[lane] = F;
}
template <bool ALL>
__device__ __forceinline__ void seg_flush_queue(SegHitLds<false> *, uint2 *, int) {}

template <bool HITS>
__global__ void FUSED_BOUNDS segment_kernel(FusedArgs A) {
  __shared__ WaveLds lds[FW];
  __shared__ SegLds seg[FW];
  __shared__ SegHitLds<HITS> hitq[FW];
  // (four blocks per CU at 37 888 bytes; at 39 936 -- four times that is still under 160 KB -- the part held three, and the kernel
  //  lost 12 %: what only the hit lists need lives in SegHitLds)
  static_assert(HITS || FW * (sizeof(WaveLds) + sizeof(SegLds)) <= 37888, "the plain variant's LDS block grew: four blocks per CU no longer fit");
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  WaveLds *L = &lds[wib];
  SegLds *G = &seg[wib];
  SegHitLds<HITS> *Q = &hitq[wib];
  const ScanParams P = A.P;
  const int k = P.k, m = P.m;
  const uint64_t nwaves = (uint64_t)gridDim.x * FW;
  const uint64_t nunits = (uint64_t)*A.work_count;
  const uint64_t bases_end = A.offsets[A.R];
  const uint32_t VM = (1u << 1) | (1u << 3) | (1u << 7) | (1u << 20) | (1u << 21);  // A C G T U, either case
  for (uint64_t unit = (uint64_t)blockIdx.x * FW + wib, it = 0;; it++) {
    if (it) unit = A.work_draw ? nwaves + next_unit(A.work_draw) : unit + nwaves;   // (as in fused_kernel)
    if (unit >= nunits) break;
    const uint64_t r = A.work_list[unit];
    const uint64_t o = A.offsets[r];
    const uint32_t n_all = (uint32_t)(A.offsets[r + 1] - o);
    // ---- this lane's segment ----
    const uint32_t nwin = n_all - (uint32_t)k + 1;  // (seg_min_len > k)
    const uint32_t S = max(SEG_MIN_WINDOWS, (nwin + 63) / 64);
    const uint32_t w0 = (uint32_t)lane * S;
    const bool exists = w0 < nwin;
    const uint8_t *seq = A.bases + o + (exists ? w0 : 0);
    const uint32_t n = exists ? (min(nwin, w0 + S) - w0) + (uint32_t)k - 1 : 0;
    const uint32_t room = clamp_room(bases_end - o - (exists ? w0 : 0));  // bytes from seq to the end of the caller's buffer
    uint2 *const prov = HITS ? (uint2 *)A.span_keys + o : nullptr;        // (unpaired: the fragment's span region starts at offsets[r])
    uint32_t lcount = 0;                                                   // spans of this lane so far
    if constexpr (HITS) { Q->delivered[lane] = 0; Q->flushed[lane] = 0; }
    // ---- wave state (as fused_kernel) ----
    int nbuf = 0, n_out = 0;
    int32_t nd = 0, np = 0, t0 = 0;
    bool map_mode = false;
    int32_t acc_t0 = 0, acc_none = 0;
    L->map_key[lane] = MAP_EMPTY; L->map_key[lane + 64] = MAP_EMPTY;
    L->map_cnt[lane] = 0; L->map_cnt[lane + 64] = 0;
    // ---- lane state (as lane_kernel, window width 5) ----
    bool fin = !exists;
    uint32_t pos = 0;
    uint32_t cur = 0, b1 = 0, b2 = 0, b3 = 0;
    int sb = 1;
    if (!fin) {
      uint4 v = seg_refill(G, lane, seq, 0, n, room);
      cur = v.x; b1 = v.y; b2 = v.z; b3 = v.w;
    }
    int run_class = 0;
    uint32_t run_len = 0, nvalid = 0;
    uint64_t fwd = 0, rc = 0;
    uint64_t k1 = ~0ULL, p1 = ~0ULL, p2 = ~0ULL, m41 = ~0ULL;
    uint64_t cur_val = 0;
    int32_t cur_run = 0;
    bool have_last = false;
    uint64_t last_key = 0, first_key = 0;
    int32_t total = 0, namb = 0;
    // what the borders need
    bool seen_win = false, starts_seq = false, first_run_done = false, first_run_amb = false;
    bool ends_open = false, end_amb = false;
    int first_slot = -1;       // where this lane's first super-mer sits in the span buffer until its taxon is known
    bool first_nz = false;     // ... and whether that taxon is a real one
    bool done = false;

    while (true) {
      // ================= producer: every lane takes one step; at most 64 spans per step =================
      while (!done && nbuf <= SPAN_CAP - 64) {
        if (__ballot(!fin) == 0) { done = true; break; }
        const bool act = !fin;
        const bool is_end = pos >= n;
        const uint32_t c = cur & 0xFF;
        const bool okc = ((c & 0xC0) == 0x40) && ((VM >> (c & 31)) & 1);
        uint32_t t = (c >> 1) & 3;
        t ^= t >> 1;
        const int cls = is_end ? -1 : (okc ? 1 : 0);
        const bool run_end = act && run_len > 0 && cls != run_class;
        const bool seqrun = run_class == 1 && nvalid >= (uint32_t)k;
        const bool seq_close = run_end && seqrun;
        const bool amb_close = run_end && !seqrun && run_len >= (uint32_t)k;
        const int32_t amb_kmers = (int32_t)run_len - (k - 1);
        total += amb_close ? amb_kmers : 0;
        namb += amb_close ? 1 : 0;
        first_run_amb = (run_end && !first_run_done) ? amb_close : first_run_amb;
        first_run_done = first_run_done || run_end;
        ends_open = (act && is_end) ? seq_close : ends_open;
        end_amb = (act && is_end) ? amb_close : end_amb;
        const uint64_t ekey = cur_val;
        const int32_t ekmers = cur_run;
        run_len = run_end ? 0u : run_len;
        const bool proc = act && !is_end;
        const bool new_run = proc && run_len == 0;
        run_class = new_run ? cls : run_class;
        nvalid = new_run ? 0u : nvalid;
        cur_run = (new_run || seq_close) ? 0 : cur_run;
        run_len += proc ? 1u : 0u;
        const bool nt = proc && okc;
        nvalid += nt ? 1u : 0u;
        fwd = (fwd << 2) | ((uint64_t)t << P.sh);
        rc = ((rc >> 2) | ((uint64_t)(3 - t) << 62)) & P.keep;
        const bool havekey = nt && nvalid >= (uint32_t)m;
        const uint64_t canon = (P.canonical && rc < fwd) ? rc : fwd;
        const uint64_t key = (canon ^ P.xmask) & P.smask;
        const uint64_t pm = umin64(key, k1);
        const uint64_t m4 = umin64(pm, p2);
        const uint64_t minv = umin64(key, m41);
        k1 = key; p2 = p1; p1 = pm; m41 = m4;
        const bool havewin = havekey && nvalid >= (uint32_t)k;
        starts_seq = (havewin && !seen_win) ? (pos == (uint32_t)k - 1) : starts_seq;  // the segment's first window is a k-mer
        seen_win = seen_win || havewin;
        const bool start = havewin && cur_run == 0;
        const bool same = havewin && cur_run != 0 && minv == cur_val;
        const bool change = havewin && cur_run != 0 && minv != cur_val;
        const bool emit = seq_close || change;
        cur_val = (start || change) ? minv : cur_val;
        cur_run = (start || change) ? 1 : (same ? cur_run + 1 : cur_run);
        pos += proc ? 1u : 0u;
        const bool nextdw = proc && (pos & 3) == 0;
        cur = nextdw ? b1 : (proc ? (cur >> 8) : cur);
        b1 = nextdw ? b2 : b1;
        b2 = nextdw ? b3 : b2;
        const bool refill = proc && (pos & 15) == 0;
        if (__ballot(refill) != 0) {
          if (refill && pos < n) {
            uint4 v;
            if (sb < SEG_SBLK) { v = G->sbuf[(sb - 1) * 64 + lane]; sb++; }
            else { v = seg_refill(G, lane, seq, pos, n, room); sb = 1; }
            cur = v.x; b1 = v.y; b2 = v.z; b3 = v.w;
          }
        }
        fin = fin || (act && is_end);
        const bool distinct = emit && !(have_last && ekey == last_key);  // (a segment's first: provisionally distinct)
        const bool is_first = emit && !have_last;
        first_key = is_first ? ekey : first_key;
        last_key = emit ? ekey : last_key;
        have_last = have_last || emit;
        total += emit ? ekmers : 0;
        // (HITS: the ambiguous spans take their place in the buffer too -- the hit list is in span order, and a lane's entries reach
        //  its queue in the order of the buffer; probe_chunk gives them their special taxon, nothing else looks at them)
        const bool push = emit || (HITS && amb_close);
        const uint64_t E = __ballot(push);
        if (E != 0) {
          if (push) {
            const int slot = nbuf + lanes_below(E);
            if (emit) put_span(L, slot, ekey, ekmers, 1, distinct);
            else put_span(L, slot, 0, amb_kmers, 2, false);
            if constexpr (HITS) Q->span_dst[slot] = ((uint32_t)lane << 26) | lcount;     // whose span, and its number there (a lane has fewer than 2^25 windows)
            if (is_first && lane > 0) first_slot = slot;
          }
          lcount += push ? 1u : 0u;
          nbuf += __popcll(E);
        }
      }

      // ================= consumer: whole chunks of 64 (everything at the end) =================
      wave_sync();
      const int nflush = done ? nbuf : (nbuf & ~63);
      for (int s0 = 0; s0 < nflush; s0 += 64) {
        const int cnt = min(64, nflush - s0);
        const bool in = lane < cnt;
        const int32_t meta = in ? L->span_meta[s0 + lane] : 0;
        const int32_t taxon = probe_chunk(L, A.T, s0, cnt, lane, meta);
        const int32_t count = meta_kmers(meta);
        const bool real = in && taxon >= 0;
        if constexpr (HITS) {
          if (in) {   // the entry to its lane's queue (or, beyond what the queue holds, straight to its place)
            const uint32_t dst = Q->span_dst[s0 + lane], p = dst >> 26, i = dst & 0x3FFFFFFu;
            const uint2 e = make_uint2((uint32_t)ext_taxon(A.T, taxon), (uint32_t)meta);
            if (i < Q->flushed[p] + (uint32_t)SEG_QL) Q->q[i % SEG_QL][p] = e;
            else prov[(uint64_t)p * S + i] = e;
            atomicMax(&Q->delivered[p], i + 1);
          }
          wave_sync();
          seg_flush_queue<false>(Q, prov + w0, lane);
          wave_sync();
        }
        nd += __popcll(__ballot(in && meta_distinct(meta) && taxon != 0));
        np += __popcll(__ballot(real));
        {  // the lanes whose first super-mer is in this chunk learn its taxon
          const int src = first_slot - s0;
          const bool mine = first_slot >= 0 && src >= 0 && src < cnt;
          const int32_t tf = __shfl(taxon, mine ? src : 0);
          if (mine) { first_nz = tf != 0; first_slot = -1; }
        }
        if (!map_mode) {
          uint64_t nz = __ballot(real && taxon != 0);
          if (nz != 0) {
            if (t0 == 0) t0 = __builtin_amdgcn_readlane(taxon, __builtin_ctzll(nz));
            if (__ballot(real && taxon != 0 && taxon != t0) != 0) {
              int32_t c0 = wave_sum(acc_t0), cn = wave_sum(acc_none);
              if (lane == 0 && c0 != 0) map_insert(L, t0, c0, A.status);
              if (lane == 1 && cn != 0) map_insert(L, 0, cn, A.status);
              map_mode = true;
            }
          }
        }
        if (map_mode) {
          if (real) map_insert(L, taxon, count, A.status);
        } else if (real) {
          if (taxon != 0) acc_t0 += count;
          else acc_none += count;
        }
        n_out += cnt;
      }
      wave_sync();
      if (done) break;
      // the spans short of a whole chunk move to the front of the buffer
      const int rest = nbuf - nflush;
      if (nflush > 0 && rest > 0) {
        uint64_t kk = 0; int32_t mm = 0; uint32_t dd = 0;
        if (lane < rest) { kk = L->span_key[nflush + lane]; mm = L->span_meta[nflush + lane]; if constexpr (HITS) dd = Q->span_dst[nflush + lane]; }
        wave_sync();
        if (lane < rest) { L->span_key[lane] = kk; L->span_meta[lane] = mm; if constexpr (HITS) Q->span_dst[lane] = dd; }
        if (first_slot >= 0) first_slot -= nflush;
        wave_sync();
      }
      nbuf = rest;
    }

    // ---- the borders ----
    G->first_key[lane] = first_key;
    G->last_key[lane] = last_key;
    G->flags[lane] = (have_last ? SEGF_HAS : 0) | (ends_open ? SEGF_ENDS_OPEN : 0) | (end_amb ? SEGF_END_AMB : 0);
    wave_sync();
    bool undo_distinct = false;
    int merged = 0;
    if (lane > 0 && exists) {
      if (have_last && first_nz) {  // the super-mer before this segment's first one: the nearest earlier lane that has any
        int q = lane - 1;
        while (q >= 0 && !(G->flags[q] & SEGF_HAS)) q--;
        undo_distinct = q >= 0 && G->last_key[q] == first_key;
      }
      const uint32_t fp = G->flags[lane - 1];
      if ((fp & SEGF_ENDS_OPEN) && starts_seq && G->last_key[lane - 1] == first_key) merged = 1;  // one super-mer, cut
      if ((fp & SEGF_END_AMB) && first_run_amb) merged = 1;                                      // one ambiguous span, cut
    }
    nd -= __popcll(__ballot(undo_distinct));
    n_out += (HITS ? 0 : wave_sum(namb)) - wave_sum(merged);   // (HITS: the ambiguous spans went through the buffer and are counted)
    total = wave_sum(total);
    if constexpr (HITS) {
      // the hit list in ordinal order: lane j's entries start where the earlier lanes' end
      const uint32_t mine = lcount - (uint32_t)merged;        // (merged implies lcount >= 1: the cut span is this lane's first)
      uint32_t incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += up;
      }
      const uint32_t off = incl - mine;
      seg_flush_queue<true>(Q, prov + w0, lane);              // what is left in the queues
      wave_sync();
      Q->delivered[lane] = incl;                               // (the queues are done with: where each lane's entries end in the list,
      Q->flushed[lane] = (uint32_t)merged;                     //  and whether its first one went into the lane before)
      // (the scratch entries were written by other lanes of THIS wave: workgroup scope -- an agent-scope fence writes the XCD's L2
      //  back, twice per fragment: measured at 4x the kernel's time)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      wave_sync();
      // the copy by the whole wave, 64 consecutive places of the list at a time (whole lines written, and the reads run along the
      // lanes' stretches): a lane finds whose entry belongs at its place by bisection over the 64 ends.  (A lane copying its own
      // stretch writes 64 places far apart with every store: three scattered requests per entry, 9 ms per Gbp.)
      const uint32_t n_list = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      for (uint32_t t = lane; t < n_list; t += 64) {
        int j = 0;
#pragma unroll
        for (int b = 32; b > 0; b >>= 1) j += (Q->delivered[j + b - 1] <= t) ? b : 0;   // the first lane whose end lies beyond t
        const uint32_t begin = j > 0 ? Q->delivered[j - 1] : 0u;
        const uint2 e = prov[(uint64_t)j * S + Q->flushed[j] + (t - begin)];
        A.span_taxon[o + t] = (int32_t)e.x;
        A.span_meta[o + t] = (int32_t)e.y;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      wave_sync();
      if (merged) atomicAdd(&A.span_meta[o + off - 1], meta_kmers((int32_t)prov[w0].y) << 4);   // (engine.h pack_meta: k-mers << 4)
      if (lane == 0) A.span_count[r] = n_out;
    }

    // ---- per-read classification (as fused_kernel) ----
    if (map_mode) {
      resolve_map(L, A, r, lane, total, nd);
    } else {
      bool need_count = false;
      for (int32_t c = 0; c < A.C; c++) need_count |= A.thr.v[c] > 0.0;
      int32_t c0 = (need_count && t0 != 0) ? wave_sum(acc_t0) : 0;
      for (int32_t c = 0; c < A.C; c++) {
        double required = ceil(__dmul_rn(A.thr.v[c], (double)total));
        int32_t mt = (t0 != 0 && !((double)c0 < required)) ? t0 : 0;
        bool classified = (mt != 0) && (nd >= A.min_hit_groups);
        if (lane == 0) {
          A.out_taxon[(uint64_t)c * A.out_stride + r] = classified ? ext_taxon(A.T, mt) : 0;
          A.out_classified[(uint64_t)c * A.out_stride + r] = classified ? 1 : 0;
        }
      }
    }
    if (lane == 0) {
      if (A.out_nd) A.out_nd[r] = nd;
      if (A.out_tk) A.out_tk[r] = total;
      if (A.out_nh) A.out_nh[r] = n_out;
      if (A.out_np) A.out_np[r] = np;
    }
    wave_sync();
  }
}

// over a hand-on list (A.work_list): unpaired, window width 5; A.span_taxon set: the hit lists too (A.span_keys is their scratch)
void launch_segments(const FusedArgs &A, hipStream_t s) {
  if (A.span_taxon) hipLaunchKernelGGL(segment_kernel<true>, dim3(256 * 8), dim3(FW * 64), 0, s, A);
  else hipLaunchKernelGGL(segment_kernel<false>, dim3(256 * 8), dim3(FW * 64), 0, s, A);
}

void launch_fused(int mode, const FusedArgs &A, hipStream_t s) {
  if (A.R == 0) return;
  // a persistent grid (measured: one block per four fragments is 10 % slower for 150-base reads -- the per-wave set-up is
  // not free here, unlike in the lane kernel); the deferral pass, whose length is only known on the device, likewise
  uint64_t blocks = (A.R + FW - 1) / FW;
  static const int bpc = getenv("SLK_FUSED_BLOCKS_PER_CU") ? atoi(getenv("SLK_FUSED_BLOCKS_PER_CU")) : 8;  // (tuning experiment)
  const uint64_t cap = (A.work_list || bpc <= 0) ? 256 * 8 : (uint64_t)256 * bpc;
  if (blocks > cap) blocks = cap;
  dim3 g((unsigned)blocks), b(FW * 64);
  const bool w5 = A.P.w == 5;
  if (mode == MODE_SPANS) { if (w5) hipLaunchKernelGGL((fused_kernel<MODE_SPANS, true>), g, b, 0, s, A); else hipLaunchKernelGGL((fused_kernel<MODE_SPANS, false>), g, b, 0, s, A); }
  else if (mode == MODE_HITS) { if (w5) hipLaunchKernelGGL((fused_kernel<MODE_HITS, true>), g, b, 0, s, A); else hipLaunchKernelGGL((fused_kernel<MODE_HITS, false>), g, b, 0, s, A); }
  else { if (w5) hipLaunchKernelGGL((fused_kernel<MODE_CLASSIFY, true>), g, b, 0, s, A); else hipLaunchKernelGGL((fused_kernel<MODE_CLASSIFY, false>), g, b, 0, s, A); }
}

}  // namespace slk

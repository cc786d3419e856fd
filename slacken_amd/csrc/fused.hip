// fused.hip -- the production kernels: ONE WAVEFRONT PER FRAGMENT, scan -> probe -> per-read LCA fused in one launch, no
// HBM intermediates (spans, hits and the per-read taxon map live in registers / LDS).  gfx950, wave64.  Integer/byte
// work bounded by random 64-byte HBM probes: no MFMA.
//
//   scan   2-bit packing of 256 bases per step (coalesced dword loads, SWAR code/pack), per-lane m-mer extraction
//          from the packed LDS ring (funnel shift), reverse complement by bit reversal, canonical/XOR/space mask,
//          width-w window minimum by log-step lane shuffles, run-length merge of equal minima by ballot + bit scans.
//          Reference: KeyValueIndex.getSpans (S/slacken/KeyValueIndex.scala:163-173) = Supermers.splitFragment/spans
//          (S/slacken/Supermers.scala:49-125) over MinSplitter.splitEncode (S/kmers/minimizer/MinSplitter.scala:98-172),
//          ShiftScanner.allMatches (ShiftScanner.scala:90-159), RandomXOR/SpacedSeed (MinimizerPriorities.scala:144-321).
//   probe  8 lanes read one 64-byte bucket (8 x 8 B, one HBM line per probe), 8 probes per wave instruction, all of a
//          chunk's loads in flight before the first compare.  Reference: the left join + spanToHit
//          (S/slacken/Classifier.scala:84-88, KeyValueIndex.scala:176-185).
//   LCA    hits folded into a 128-slot LDS hash map (taxon -> k-mer count); resolveTree with one lane per distinct
//          taxon.  Reference: TaxonCounts.toMap/totalKmers (S/slacken/TaxonCounts.scala:70-87),
//          LowestCommonAncestor.apply/resolveTree (S/slacken/LowestCommonAncestor.scala:49-146), Classifier.classify
//          (Classifier.scala:439-454).
// Fragments containing a non-ACGTU character take a sequential single-lane scan (same state machine as kernels.hip).
#include "engine.h"

namespace slk {

constexpr int FW = 4;          // waves (fragments in flight) per block
constexpr int SPAN_CAP = 128;  // buffered spans per wave before a flush
constexpr int MAP_CAP = 128;   // taxon map slots per wave (power of two)
constexpr int32_t MAP_EMPTY = -1;  // AMBIGUOUS_SPAN is never inserted, so -1 is free

struct __attribute__((aligned(16))) WaveLds {
  uint64_t packed[16];           // 2-bit bases, MSB first, ring of 512 bases (two 256-base blocks)
  uint64_t span_key[SPAN_CAP];
  int32_t span_meta[SPAN_CAP];
  uint64_t stash[128];           // probe: (bucket, tag) per span of the chunk; resolveTree: dense (taxon,count) list
  int32_t result[64];
  int32_t map_key[MAP_CAP];
  int32_t map_cnt[MAP_CAP];
  uint64_t seq_ring[64];         // window ring of the sequential (slow-path) scanner, w <= 64
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

struct Ctx {
  WaveLds *L;
  int lane;
  int nbuf;        // spans buffered in L->span_*           (wave-uniform)
  int n_out;       // spans already flushed for this read   (wave-uniform)
  bool first;      // Supermers.spans :72                    (wave-uniform)
  bool have_last;  // lastMinimizer != Array()               (wave-uniform)
  uint64_t last_key;
  uint64_t base;   // span_region of this read
  int32_t part_total, part_nd, part_np;  // per-lane partial sums over this read's hits
};

// ---- staging: 256 characters -> 2-bit codes in the packed ring -------------------------------------------------------
// Characters 4*lane .. 4*lane+3 of block `blk`; only aligned dwords holding at least one wanted byte are touched.
__device__ __forceinline__ uint32_t load4(const uint8_t *seq, uint32_t n, uint32_t pos, uint32_t &nchars) {
  nchars = pos < n ? min(4u, n - pos) : 0u;
  if (nchars == 0) return 0;
  uintptr_t a = (uintptr_t)(seq + pos);
  const uint32_t *al = (const uint32_t *)(a & ~(uintptr_t)3);
  uint32_t mis = (uint32_t)(a & 3);
  uint32_t d0 = al[0], d1 = 0;
  if (mis != 0 && nchars > 4 - mis) d1 = al[1];
  uint32_t v = mis ? __builtin_amdgcn_alignbyte(d1, d0, mis) : d0;
  if (nchars < 4) v &= (1u << (nchars * 8)) - 1;
  return v;
}

// returns true in lanes that saw a character outside ACGTUacgtu (BitRepresentation.isValid, BitRepresentation.scala:140-143)
__device__ __forceinline__ bool stage_block(WaveLds *L, const uint8_t *seq, uint32_t n, uint32_t blk, int lane) {
  uint32_t nchars;
  uint32_t v = load4(seq, n, blk * 256 + lane * 4, nchars);
  bool bad = false;
  const uint32_t VMASK = 1u | (1u << 2) | (1u << 6) | (1u << 0x13) | (1u << 0x14);  // A C G T U minus 'A'
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t x = (((v >> (8 * j)) & 0xDF) - 0x41);
    bool ok = x < 32 && ((VMASK >> x) & 1);
    bad |= (j < (int)nchars) && !ok;
  }
  // (c >> 1) & 3 maps A,C,T/U,G (either case) to 0,1,2,3; x ^ (x >> 1) turns that into A=0 C=1 G=2 T=3
  uint32_t t = (v >> 1) & 0x03030303u;
  t ^= (t >> 1) & 0x01010101u;
  uint32_t pack = (t * 0x40100401u) >> 24;  // code0<<6 | code1<<4 | code2<<2 | code3
  uint32_t g = (blk * 64 + lane) & 127;     // group of 4 bases within the 512-base ring
  ((uint8_t *)L->packed)[(g & ~7u) | (7u - (g & 7u))] = (uint8_t)pack;  // MSB-first inside each 64-bit word
  return bad;
}

// left-aligned key of the m-mer starting at base q (NTBitArray layout; SpacedSeed(RandomXOR) priority)
__device__ __forceinline__ uint64_t key_at(const WaveLds *L, const ScanParams &P, uint32_t q) {
  uint32_t wq = (q >> 5) & 15, o = (q & 31) * 2;
  uint64_t a = L->packed[wq], b = L->packed[(wq + 1) & 15];
  uint64_t fwd = o ? ((a << o) | (b >> (64 - o))) : a;
  fwd &= P.keep;
  uint64_t canon = fwd;
  if (P.canonical) {
    // reverse complement: complement, reverse all 64 bits, swap the two bits of every pair back, drop the padding
    uint64_t x = __brevll(~fwd);
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    uint64_t rc = x << P.sh;
    canon = umin64(fwd, rc);  // NTBitArray.writeCanonical :258-266
  }
  return (canon ^ P.xmask) & P.smask;
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

// Look up the (<= 64) buffered spans [s0, s0+cnt) and return this lane's taxon (lane l <-> span s0+l).
__device__ __forceinline__ int32_t probe_chunk(WaveLds *L, const TableView &T, int s0, int cnt, int lane) {
  const uint64_t NO_TAG = ~0ULL;  // a real tag has at most 64 - taxon_bits significant bits
  uint64_t key = 0;
  int32_t meta = 0;
  if (lane < cnt) { key = L->span_key[s0 + lane]; meta = L->span_meta[s0 + lane]; }
  int32_t flag = meta_flag(meta);
  bool seq = (lane < cnt) && flag == 1;
  uint64_t h = fmix64(key);
  uint64_t bucket = h >> T.shift;
  uint64_t tag = seq ? ((h & T.rem_mask) << T.disp_bits) : NO_TAG;
  L->stash[2 * lane] = bucket;
  L->stash[2 * lane + 1] = tag;
  L->result[lane] = 0;
  wave_sync();
  const int g = lane >> 3, c = lane & 7;
  const uint64_t tmask = (1ULL << T.taxon_bits) - 1;
  // displacement 0: all eight steps' loads in flight before the first compare
  uint64_t cell[8];
#pragma unroll
  for (int s = 0; s < 8; s++) {
    int idx = s * 8 + g;
    uint64_t bkt = L->stash[2 * idx], want = L->stash[2 * idx + 1];
    cell[s] = 0;
    if (want != NO_TAG) cell[s] = T.cells[((bkt & T.bucket_mask) << 3) + c];
  }
  uint32_t more = 0;  // bit s set: group's span s*8+g must look at the next bucket
#pragma unroll
  for (int s = 0; s < 8; s++) {
    uint64_t want = L->stash[2 * (s * 8 + g) + 1];
    bool act = want != NO_TAG;
    bool match = act && cell[s] != 0 && (cell[s] >> T.taxon_bits) == want;
    bool empty = cell[s] == 0;
    uint64_t mm = __ballot(match), me = __ballot(empty);
    uint32_t gm = (uint32_t)(mm >> (g * 8)) & 0xFF, ge = (uint32_t)(me >> (g * 8)) & 0xFF;
    if (match) L->result[s * 8 + g] = (int32_t)(cell[s] & tmask);
    if (act && gm == 0 && ge == 0) more |= 1u << s;  // bucket full, key not in it: overflowed to a later bucket
  }
  // rare: follow bucket-level linear probing (cells are never freed, so the first non-full bucket ends the search)
  for (int d = 1; d <= T.max_disp && __ballot(more != 0) != 0; d++) {
#pragma unroll
    for (int s = 0; s < 8; s++) {
      bool act = (more >> s) & 1;
      uint64_t cl = 0, want = 0;
      if (act) {
        uint64_t bkt = L->stash[2 * (s * 8 + g)];
        want = L->stash[2 * (s * 8 + g) + 1] + (uint64_t)d;
        cl = T.cells[(((bkt + d) & T.bucket_mask) << 3) + c];
      }
      bool match = act && cl != 0 && (cl >> T.taxon_bits) == want;
      bool empty = cl == 0;
      uint64_t mm = __ballot(match), me = __ballot(empty);
      uint32_t gm = (uint32_t)(mm >> (g * 8)) & 0xFF, ge = (uint32_t)(me >> (g * 8)) & 0xFF;
      if (match) L->result[s * 8 + g] = (int32_t)(cl & tmask);
      if (act && (gm != 0 || ge != 0)) more &= ~(1u << s);
    }
  }
  wave_sync();
  int32_t taxon = L->result[lane];
  if (flag == 2) taxon = -1;       // spanToHit: the flag wins over any record (KeyValueIndex.scala:176-185)
  else if (flag == 3) taxon = -2;
  return taxon;
}

template <int MODE>
__device__ __forceinline__ void flush(Ctx &X, const FusedArgs &A) {
  WaveLds *L = X.L;
  const int lane = X.lane;
  wave_sync();
  for (int s0 = 0; s0 < X.nbuf; s0 += 64) {
    int cnt = min(64, X.nbuf - s0);
    if (MODE == MODE_SPANS) {
      if (lane < cnt) {
        A.span_keys[X.base + X.n_out + lane] = L->span_key[s0 + lane];
        A.span_meta[X.base + X.n_out + lane] = L->span_meta[s0 + lane];
      }
    } else {
      int32_t taxon = probe_chunk(L, A.T, s0, cnt, lane);
      if (lane < cnt) {
        int32_t meta = L->span_meta[s0 + lane];
        int32_t count = meta_kmers(meta);
        if (taxon != -2) X.part_total += count;                  // TaxonCounts.totalKmers :84-87
        if (meta_distinct(meta) && taxon != 0) X.part_nd++;       // Classifier.scala:94
        if (taxon != -1 && taxon != -2) {                         // TaxonCounts.toMap :70-81
          X.part_np++;
          map_insert(L, taxon, count, A.status);
        }
        if (MODE == MODE_HITS) {
          A.span_meta[X.base + X.n_out + lane] = meta;
          A.span_taxon[X.base + X.n_out + lane] = taxon;
        }
      }
    }
    X.n_out += cnt;
  }
  X.nbuf = 0;
  wave_sync();
}

// ---- scan, fast path: every character of the mate is a nucleotide ----------------------------------------------------
template <int MODE>
__device__ void scan_mate_fast(Ctx &X, const FusedArgs &A, const uint8_t *seq, uint32_t n, uint32_t staged) {
  const ScanParams &P = A.P;
  WaveLds *L = X.L;
  const int lane = X.lane;
  const int w = P.w;
  const uint32_t nwin = n - P.k + 1;           // caller guarantees n >= k
  const uint32_t STEP = 64 - (w - 1);          // windows per round
  uint32_t blk = staged / 256;
  uint64_t carry_val = 0;
  int carry_run = 0;
  bool seg_first = true;                       // no span of this segment emitted yet
  for (uint32_t i0 = 0; i0 < nwin; i0 += STEP) {
    uint32_t need = min(n, i0 + 64 + P.m - 1);
    while (staged < need) {
      stage_block(L, seq, n, blk, lane);
      blk++;
      staged += 256;
    }
    if (X.nbuf > SPAN_CAP - 66) flush<MODE>(X, A);
    wave_sync();
    uint32_t q = i0 + lane;
    uint64_t key = (q + P.m <= n) ? key_at(L, P, q) : ~0ULL;
    // minimum over lanes [l, l+w): doubling, then one overlapping step (PosRankWindow's observable result)
    uint64_t cur = key;
    int covered = 1;
    for (; covered * 2 <= w; covered *= 2) cur = umin64(cur, shfl64(cur, lane + covered));
    uint64_t res = (covered < w) ? umin64(cur, shfl64(cur, lane + (w - covered))) : cur;
    int nw = (int)min(STEP, nwin - i0);
    bool valid = lane < nw;
    uint64_t prev = shfl64(res, lane - 1);
    bool is_start = valid && (lane == 0 ? (carry_run == 0 || res != carry_val) : (res != prev));
    uint64_t S = __ballot(is_start);
    if (S == 0) { carry_run += nw; continue; }  // MinSplitter.splitRead :154-158: equal value => same super-mer
    int firstl = __builtin_ctzll(S), lastl = 63 - __builtin_clzll(S);
    bool special = X.first || !X.have_last;     // distinct test of the segment's first span (Supermers.spans :84-86)
    int nclose = 0;
    if (carry_run > 0) {
      if (lane == 0) put_span(L, X.nbuf, carry_val, carry_run + firstl, 1, seg_first ? (special || carry_val != X.last_key) : true);
      nclose = 1;
      seg_first = false;
    }
    int rank = __popcll(S & ((1ULL << lane) - 1));
    if (is_start && lane != lastl) {
      uint64_t rest = S >> (lane + 1);
      int d = __builtin_ctzll(rest) + 1;
      bool dist = (seg_first && lane == firstl) ? (special || res != X.last_key) : true;
      put_span(L, X.nbuf + nclose + rank, res, d, 1, dist);
    }
    int emitted = nclose + __popcll(S) - 1;
    if (emitted > 0) { seg_first = false; X.first = false; }
    X.nbuf += emitted;
    carry_val = shfl64(res, lastl);
    carry_run = nw - lastl;
  }
  // the segment's last super-mer
  if (lane == 0) put_span(L, X.nbuf, carry_val, carry_run, 1, seg_first ? (X.first || !X.have_last || carry_val != X.last_key) : true);
  X.nbuf += 1;
  X.first = false;
  X.have_last = true;
  X.last_key = carry_val;
}

// ---- scan, slow path: one lane walks the mate character by character (mates with ambiguous characters) ------------------
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

template <int MODE>
__device__ void scan_mate_slow(Ctx &X, const FusedArgs &A, const uint8_t *seq, uint32_t n) {
  const ScanParams &P = A.P;
  WaveLds *L = X.L;
  const int k = P.k, m = P.m, w = P.w;
  // sequential state (meaningful in lane 0 only)
  uint32_t i = 0, run_len = 0, nvalid = 0;
  int run_class = 0, head = 0, minage = 0;
  uint64_t fwd = 0, rc = 0, minv = 0, cur_val = 0;
  int32_t cur_run = 0;
  int nbuf = X.nbuf;
  bool first = X.first, have_last = X.have_last;
  uint64_t last_key = X.last_key;
  auto emit = [&](uint64_t key, int32_t kmers, int32_t flag) {
    bool seqlike = flag == 1;
    bool distinct = seqlike && (first || !(have_last && key == last_key));
    if (seqlike) { last_key = key; have_last = true; }
    first = false;
    put_span(L, nbuf, seqlike ? key : 0, kmers, flag, distinct);
    nbuf++;
  };
  bool done = false;
  while (!done) {
    if (X.lane == 0) {
      while (i <= n && nbuf < SPAN_CAP - 2) {
        int t = 5, cls = -1;
        if (i < n) { t = char_code2(seq[i]); cls = (t < 4) ? 1 : 0; }
        if (run_len > 0 && cls != run_class) {
          if (run_class == 1 && nvalid >= (uint32_t)k) emit(cur_val, cur_run, 1);
          else if (run_len >= (uint32_t)k) emit(0, (int32_t)run_len - (k - 1), 2);  // Supermers.scala:116-119
          run_len = 0;
        }
        if (i == n) { i++; break; }
        if (run_len == 0) { run_class = cls; nvalid = 0; fwd = 0; rc = 0; head = w - 1; minage = 0; minv = ~0ULL; cur_run = 0; }
        run_len++;
        if (t < 4) {
          nvalid++;
          fwd = (fwd << 2) | ((uint64_t)t << P.sh);
          rc = ((rc >> 2) | ((uint64_t)(3 - t) << 62)) & P.keep;
          if (nvalid >= (uint32_t)m) {
            uint64_t canon = (P.canonical && rc < fwd) ? rc : fwd;
            uint64_t key = (canon ^ P.xmask) & P.smask;
            head = (head + 1 == w) ? 0 : head + 1;
            L->seq_ring[head] = key;
            if (key <= minv) { minv = key; minage = 0; }
            else if (++minage >= w) {
              int slot = (head + 1 == w) ? 0 : head + 1;
              minv = ~0ULL;
              for (int a = w - 1; a >= 0; a--) {
                uint64_t v = L->seq_ring[slot];
                if (v <= minv) { minv = v; minage = a; }
                slot = (slot + 1 == w) ? 0 : slot + 1;
              }
            }
            if (nvalid >= (uint32_t)k) {
              if (cur_run == 0) { cur_val = minv; cur_run = 1; }
              else if (minv == cur_val) cur_run++;
              else { emit(cur_val, cur_run, 1); cur_val = minv; cur_run = 1; }
            }
          }
        }
        i++;
      }
    }
    X.nbuf = __builtin_amdgcn_readfirstlane(nbuf);
    done = __builtin_amdgcn_readfirstlane((int)(i > n)) != 0;
    if (!done) flush<MODE>(X, A);
    nbuf = X.nbuf;
  }
  X.first = __builtin_amdgcn_readfirstlane((int)first) != 0;
  X.have_last = __builtin_amdgcn_readfirstlane((int)have_last) != 0;
  X.last_key = bcast64(last_key);
}

template <int MODE>
__device__ void scan_mate(Ctx &X, const FusedArgs &A, const uint8_t *seq, uint32_t n) {
  // pre-scan: any character outside ACGTUacgtu? (stages block 0 as a side effect)
  bool bad = false;
  uint32_t nblk = (n + 255) / 256;
  for (uint32_t b = nblk; b-- > 0;) bad |= stage_block(X.L, seq, n, b, X.lane);  // block 0 last: it stays staged
  bool any_bad = __ballot(bad) != 0;
  if (any_bad) {
    scan_mate_slow<MODE>(X, A, seq, n);
  } else if (n >= (uint32_t)A.P.k) {
    // blocks were staged in descending order, so the ring's two halves end up holding blocks 0 and 1
    scan_mate_fast<MODE>(X, A, seq, n, min(nblk, 2u) * 256);
  }  // an all-valid mate shorter than k yields nothing (Supermers.scala:116)
}

// LowestCommonAncestor.apply :49-78 (wave-uniform arguments and control flow)
__device__ int32_t lca_uniform(const int32_t *parents, int32_t ntax, int32_t a, int32_t b) {
  if (a == 0 || b == 0) return b == 0 ? a : b;
  for (int32_t y = b; y != 0; y = ((uint32_t)y < (uint32_t)ntax) ? parents[y] : 0)
    for (int32_t x = a; x != 0; x = ((uint32_t)x < (uint32_t)ntax) ? parents[x] : 0)
      if (x == y) return y;
  return 1;
}

template <int MODE>
__global__ void __launch_bounds__(FW * 64) fused_kernel(FusedArgs A) {
  __shared__ WaveLds lds[FW];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  WaveLds *L = &lds[wib];
  const uint64_t nwaves = (uint64_t)gridDim.x * FW;
  for (uint64_t r = (uint64_t)blockIdx.x * FW + wib; r < A.R; r += nwaves) {
    Ctx X;
    X.L = L; X.lane = lane; X.nbuf = 0; X.n_out = 0;
    X.first = true; X.have_last = false; X.last_key = 0;
    X.base = span_region(A.offsets, A.mate_offsets, r);
    X.part_total = 0; X.part_nd = 0; X.part_np = 0;
    if (MODE != MODE_SPANS) {
      L->map_key[lane] = MAP_EMPTY; L->map_key[lane + 64] = MAP_EMPTY;
      L->map_cnt[lane] = 0; L->map_cnt[lane + 64] = 0;
    }
    uint64_t o0 = A.offsets[r];
    scan_mate<MODE>(X, A, A.bases + o0, (uint32_t)(A.offsets[r + 1] - o0));
    if (A.mate_bases) {
      if (X.nbuf > SPAN_CAP - 2) flush<MODE>(X, A);
      if (lane == 0) put_span(L, X.nbuf, 0, -(A.P.k - 1), 3, false);  // MATE_PAIR_BORDER (Supermers.scala:53-57)
      X.nbuf += 1;
      X.first = false;
      uint64_t m0 = A.mate_offsets[r];
      scan_mate<MODE>(X, A, A.mate_bases + m0, (uint32_t)(A.mate_offsets[r + 1] - m0));
    }
    flush<MODE>(X, A);
    if (MODE != MODE_CLASSIFY) {
      if (lane == 0) A.span_count[r] = X.n_out;
    }
    if (MODE == MODE_SPANS) continue;

    // ---- per-read classification -------------------------------------------------------------------------------------
    int32_t total = wave_sum(X.part_total), nd = wave_sum(X.part_nd), np = wave_sum(X.part_np);
    // dense list of the map's entries
    int2 *dense = (int2 *)L->stash;
    int D = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      int slot = lane + 64 * half;
      int32_t kk = L->map_key[slot];
      bool occ = kk != MAP_EMPTY;
      uint64_t mask = __ballot(occ);
      if (occ) dense[D + __popcll(mask & ((1ULL << lane) - 1))] = make_int2(kk, L->map_cnt[slot]);
      D += __popcll(mask);
    }
    wave_sync();
    // resolveTree step 1 (:101-123): LCA of the taxa with the maximal root-path score
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
          maxTaxon = lca_uniform(A.parents, A.ntax, maxTaxon, __shfl(t, bl));
        }
      }
    }
    for (int32_t c = 0; c < A.C; c++) {
      double required = ceil(__dmul_rn(A.thresholds[c], (double)total));  // Math.ceil(confidence * totalKmers) :94
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
        A.out_taxon[(uint64_t)c * A.R + r] = classified ? mt : 0;
        A.out_classified[(uint64_t)c * A.R + r] = classified ? 1 : 0;
      }
    }
    if (lane == 0) {
      if (A.out_nd) A.out_nd[r] = nd;
      if (A.out_tk) A.out_tk[r] = total;
      if (A.out_nh) A.out_nh[r] = X.n_out;
      if (A.out_np) A.out_np[r] = np;
    }
    wave_sync();
  }
}

void launch_fused(int mode, const FusedArgs &A, hipStream_t s) {
  if (A.R == 0) return;
  uint64_t blocks = (A.R + FW - 1) / FW;
  if (blocks > 256 * 8) blocks = 256 * 8;
  dim3 g((unsigned)blocks), b(FW * 64);
  if (mode == MODE_SPANS) hipLaunchKernelGGL(fused_kernel<MODE_SPANS>, g, b, 0, s, A);
  else if (mode == MODE_HITS) hipLaunchKernelGGL(fused_kernel<MODE_HITS>, g, b, 0, s, A);
  else hipLaunchKernelGGL(fused_kernel<MODE_CLASSIFY>, g, b, 0, s, A);
}

}  // namespace slk

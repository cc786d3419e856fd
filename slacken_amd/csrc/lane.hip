// lane.hip -- the hot-path kernel for short fragments: ONE LANE PER FRAGMENT (64 fragments per wavefront, in lockstep),
// scan -> probe -> per-read LCA in one launch with no HBM intermediates.  gfx950, wave64; integer/byte work, no MFMA.
//
// Why this shape: the per-read control flow of the classify path (ambiguity runs, run-length merging, distinct tracking,
// the taxon map) is cheap per lane but expensive per wave; with one lane per read it costs nothing in scalar instructions
// and every vector instruction does 64 reads' worth of work.  What a lane cannot do efficiently alone is the table probe
// (a whole 64-byte bucket per lane costs 4x the address-unit work), so minimizers are pushed into a wave-shared LDS queue
// and probed cooperatively: 4 lanes read one bucket (4 x 16 B = one HBM line), 16 probes per wave instruction, four
// instructions in flight per batch of 64.
//
//   scan    each lane consumes its read 16 bytes at a time (fetched 80 at a time, see stream_refill), rolls the forward and
//           reverse-complement m-mer,
//           takes the canonical / XOR / spaced-seed key, a width-w sliding minimum (registers for w = 5, a van-Herk
//           prefix/suffix ring in LDS otherwise) and merges equal consecutive minima into super-mer spans.
//           Reference: KeyValueIndex.getSpans (S/slacken/KeyValueIndex.scala:163-173) = Supermers.splitByAmbiguity /
//           splitFragment / spans (S/slacken/Supermers.scala:49-125,150-189), MinSplitter.splitRead
//           (S/kmers/minimizer/MinSplitter.scala:133-172), PosRankWindow (PosRankWindow.scala:33-97), ShiftScanner.allMatches
//           (ShiftScanner.scala:90-159), RandomXOR/SpacedSeed (MinimizerPriorities.scala:144-321).
//   probe   the left join + spanToHit (S/slacken/Classifier.scala:84-88, KeyValueIndex.scala:176-185).
//   LCA     hits are folded (LDS atomics) into a 12-slot taxon->count map per read; NONE hits are never needed by
//           resolveTree and are only counted; one distinct taxon is resolved without touching the tree, several from the
//           Euler-tour intervals of the map's taxa (one 16-byte load each; no walk of parent pointers on a lineage).
//           Reference: TaxonCounts.toMap/totalKmers (S/slacken/TaxonCounts.scala:70-87), LowestCommonAncestor
//           (S/slacken/LowestCommonAncestor.scala:49-146), Classifier.classify (S/slacken/Classifier.scala:439-454).
// Fragments longer than LANE_MAX_LEN, and fragments whose 12-slot map overflows, are flagged in `defer` and re-done by the
// wave-per-read kernel of fused.hip in the same stream.  With A.span_taxon set the kernel also writes the un-merged hit lists
// (TaxonHit per span, ordinal order) into the fragments' span regions, the layout fused.hip's MODE_HITS uses.
#include "engine.h"

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>

namespace slk {

#ifndef SLK_LANE_LW
#define SLK_LANE_LW 4
#endif
constexpr int LW = SLK_LANE_LW;   // waves per block
constexpr int QCAP = 128;         // probe queue entries per wave (64 buffered + at most 64 pushed per step)
#ifndef SLK_SBLK
#define SLK_SBLK 5
#endif
constexpr int SBLK = SLK_SBLK;           // read stream: 16-byte sub-blocks fetched per refill (48 bytes per lane)
#ifndef SLK_OMAP
#define SLK_OMAP 12
#endif
constexpr int OMAP = SLK_OMAP;    // taxon map slots per fragment (any number; 12 is what the LDS left at four blocks per CU holds)
constexpr int OMAP_CNT_BITS = 10; // a fragment the lane kernel takes has <= LANE_MAX_LEN (1000) k-mers; taxon ids need <= 22 bits
constexpr uint32_t OMAP_CNT_MASK = (1u << OMAP_CNT_BITS) - 1;

struct __attribute__((aligned(16))) LaneLds {
  uint64_t q_key[QCAP];
  uint32_t q_meta[QCAP];          // owner lane (6 bits) | distinct << 6 | kmers << 7 (13 bits) | displacement << 20 (8 bits)
  uint64_t stash[128];            // (home bucket, -, tag) 16 bytes per queue entry of the batch
  uint32_t found[64];             // taxon found per entry of the batch (a word array of its own: read by all 64 lanes at once, and
                                  // as the second word of the 16-byte stash entries that read was a 4-way bank conflict)
  uint4 sbuf[(SBLK - 1) * 64];    // read stream: the staged 16-byte sub-blocks 1.. of every lane, [sub-block - 1][lane]
  uint32_t omap[OMAP * 64];       // [slot][owner lane]: taxon << 10 | k-mer count; 0 = empty (NONE hits are not stored)
                                  // (LONG variant: the taxon alone; the counts are a second array behind the per-wave block)
  uint32_t o_flags[64];           // low bits: hits with distinct && taxon != NONE (Classifier.scala:94); bit 31: map overflow
  uint64_t rb[64];                // hit-list output only: every lane's span region (engine.h span_region)
  uint16_t q_ord[QCAP];           // hit-list output only: the queue entry's ordinal among its fragment's spans
};

// Timing experiments (probes off, map updates off, everything served from the L2 ...) change what the kernel computes: they
// exist only in a build with -DSLK_TUNING (make EXTRA=-DSLK_TUNING), where SLK_DEBUG_ABLATE selects them.  The shipped
// library has no such switch.
#ifdef SLK_TUNING
#define SLK_TUNE(...) __VA_ARGS__
#define SLK_TUNE_ON(bit) ((dbg & (bit)) != 0)
#else
#define SLK_TUNE(...)
#define SLK_TUNE_ON(bit) false
#endif

#ifndef SLK_PROBE_NT
#define SLK_PROBE_NT 0
#endif
__device__ __forceinline__ ulonglong2 SLK_PROBE_LOAD(const ulonglong2 *p) {
#if SLK_PROBE_NT
  ulonglong2 v;
  v.x = __builtin_nontemporal_load(&p->x);
  v.y = __builtin_nontemporal_load(&p->y);
  return v;
#else
  return *p;
#endif
}

__device__ __forceinline__ void lane_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint64_t lmin64(uint64_t a, uint64_t b) { return a < b ? a : b; }

// Fold entry `meta`'s hit into its owner's 12-slot map: ONE round of LDS atomics per batch (NONE hits carry no information
// for resolveTree and are dropped).
template <bool LONG>
__device__ __forceinline__ void fold_hit(LaneLds *L, uint32_t *ocnt, bool in, uint32_t meta, int32_t taxon) {
  if (in && taxon != 0) {
    const int owner = meta & 63;
    const int32_t kmers = (int32_t)((meta >> 7) & 0x1FFF);
    if (meta & 64) atomicAdd(&L->o_flags[owner], 1u);  // distinct && taxon != NONE (Classifier.scala:94)
    uint32_t slot = (uint32_t)(((uint64_t)((uint32_t)taxon * 0x9E3779B1u) * (uint32_t)OMAP) >> 32);  // hash -> [0, OMAP)
    int p = 0;
    if (LONG) {  // fragments of up to 8191 bases: 32-bit taxon and 32-bit count in two words
      for (; p < OMAP; p++) {
        uint32_t old = atomicCAS(&L->omap[slot * 64 + owner], 0u, (uint32_t)taxon);
        if (old == 0u || old == (uint32_t)taxon) { atomicAdd(&ocnt[slot * 64 + owner], (uint32_t)kmers); break; }
        slot = (slot + 1 == (uint32_t)OMAP) ? 0u : slot + 1;
      }
    } else {
      const uint32_t fresh = ((uint32_t)taxon << OMAP_CNT_BITS) | (uint32_t)kmers;
      for (; p < OMAP; p++) {
        uint32_t old = atomicCAS(&L->omap[slot * 64 + owner], 0u, fresh);
        if (old == 0u) break;
        if ((old >> OMAP_CNT_BITS) == (uint32_t)taxon) { atomicAdd(&L->omap[slot * 64 + owner], (uint32_t)kmers); break; }
        slot = (slot + 1 == (uint32_t)OMAP) ? 0u : slot + 1;
      }
    }
    if (p == OMAP) atomicOr(&L->o_flags[owner], 0x80000000u);
  }
}

// Probe `cnt` (<= 64) entries of the ring queue starting at `qhead` and fold the hits into their owners' maps.
//   1. lane i hashes entry i and parks (bucket, tag) in LDS;
//   2. LPB lanes read one bucket together (LPB x 16 B: one 64-byte access granule with 8-cell buckets, one 128-byte line with
//      16-cell ones), 64 / LPB probes per wave instruction, all LPB instructions of the batch in flight before the first compare;
//      the lane that finds the key writes the taxon to found[entry];
//   3. lane i folds entry i's hit into its owner's 12-slot map: ONE round of LDS atomics per batch (NONE hits carry no
//      information for resolveTree and are dropped).
constexpr int PG = 64 / LPB;   // probes per wave instruction
template <bool HITS, bool LONG>
__device__ __forceinline__ int probe_batch(LaneLds *L, uint32_t *ocnt, const TableView &T, int qhead, int qn, int cnt, int lane, int dbg,
                                           int32_t *hit_meta, int32_t *hit_taxon) {
  const uint64_t NO_TAG = ~0ULL;
  const bool in = lane < cnt;
  const int qi = (qhead + lane) & (QCAP - 1);
  const uint64_t key = L->q_key[qi];
  const uint32_t meta = L->q_meta[qi];
  const uint32_t ord = HITS ? L->q_ord[qi] : 0u;  // (in a register now: re-queued entries may wrap onto this batch's slots)
  const uint64_t h = fmix64(key);
  uint32_t home;
  uint64_t rem_hi;
  table_slot(T.g, h, home, rem_hi);
  uint4 st;
  const uint32_t disp = (meta >> 20) & 255;                               // > 0 for an entry re-queued after a full bucket
  st.x = table_bucket(T.g, home, disp);                                   // bucket to read
  st.y = 0;
  const uint64_t tag = in ? (rem_hi + disp) : NO_TAG;                     // the tag's low bits are the displacement
  st.z = (uint32_t)tag; st.w = (uint32_t)(tag >> 32);
  ((uint4 *)L->stash)[lane] = st;
  L->found[lane] = 0;
  lane_wave_sync();
  const int g = lane / LPB, c = lane % LPB;                               // 64 / LPB groups of LPB lanes
  const uint64_t tmask = (1ULL << T.g.taxon_bits) - 1;
  const char *cellbase = (const char *)T.cells + c * 16;
  ulonglong2 cell[LPB];
#pragma unroll
  for (int s = 0; s < LPB; s++) {
    uint32_t bkt = ((const uint4 *)L->stash)[s * PG + g].x;
    SLK_TUNE(if (dbg & 4) bkt &= 1023u;)                                  // (timing experiment 4: every probe hits the L2)
    // (inactive entries read some bucket: harmless)
    cell[s] = SLK_PROBE_LOAD((const ulonglong2 *)(cellbase + ((uint64_t)bkt << BUCKET_SHIFT)));
  }
  constexpr uint64_t GROUP_LSB = LPB == 4 ? 0x1111111111111111ULL : 0x0101010101010101ULL;
  int requeued = 0;
#pragma unroll
  for (int s = 0; s < LPB; s++) {
    const uint2 tg = *(const uint2 *)&((const uint4 *)L->stash)[s * PG + g].z;
    const uint64_t want = ((uint64_t)tg.y << 32) | tg.x;
    const bool act = want != NO_TAG;
    const bool e0 = cell[s].x == 0, e1 = cell[s].y == 0;
    const bool m0 = act && !e0 && cell_tag(T.g, cell[s].x) == want;
    const bool m1 = act && !e1 && cell_tag(T.g, cell[s].y) == want;
    if (m0 || m1) L->found[s * PG + g] = (uint32_t)((m0 ? cell[s].x : cell[s].y) & tmask);
    // a group is resolved by a match, by an empty cell, or by a full bucket that no record ever went past (its first cell's flag)
    const bool closed = T.g.flag != 0 && c == 0 && (cell[s].x & T.g.flag) == 0;
    const uint64_t B = __ballot(m0 || m1 || e0 || e1 || !act || closed);
    uint64_t any = B;
#pragma unroll
    for (int i = 1; i < LPB; i <<= 1) any |= any >> i;
    if ((any & GROUP_LSB) != GROUP_LSB) {
      // Some bucket was full without holding its key and has overflowed: bucket-level linear probing continues in the
      // next bucket.  Rather than following it here with a dependent load, the entry goes back to the tail of the
      // queue with its displacement raised, and takes an ordinary slot of a later batch.
      const int qj = (qhead + s * PG + g) & (QCAP - 1);
      const uint32_t m_old = L->q_meta[qj];
      const bool again = c == 0 && ((B >> (g * LPB)) & ((1u << LPB) - 1)) == 0 && (int)((m_old >> 20) & 255) < T.max_disp;
      const uint64_t k_old = L->q_key[qj];
      const uint16_t o_old = HITS ? L->q_ord[qj] : (uint16_t)0;
      const uint64_t R = __ballot(again);
      lane_wave_sync();  // every key is in registers before a tail slot (which may wrap onto this batch) is written
      if (again) {
        const int slot = (qhead + qn + requeued + __builtin_amdgcn_mbcnt_hi((uint32_t)(R >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)R, 0))) & (QCAP - 1);
        L->q_key[slot] = k_old;
        L->q_meta[slot] = m_old + (1u << 20);
        if (HITS) L->q_ord[slot] = o_old;
      }
      requeued += __popcll(R);
    }
  }
  lane_wave_sync();
  // step 3: one lane per entry
  const int32_t taxon = (int32_t)L->found[lane];
  if (!SLK_TUNE_ON(2)) fold_hit<LONG>(L, ocnt, in, meta, taxon);
  if (HITS && in) {
    // the un-merged hit list (TaxonHit, KeyValueIndex.scala:436-441) in the fragment's span region.  An entry handed back
    // to the queue writes NONE here and its final taxon when a later batch resolves it.
    const uint64_t at = L->rb[meta & 63] + ord;
    hit_taxon[at] = ext_taxon(T, taxon);
    hit_meta[at] = pack_meta((int32_t)((meta >> 7) & 0x1FFF), 1, (meta >> 6) & 1);
  }
  lane_wave_sync();
  return requeued;
}

// The lists of the table-sharded mode are streams: written once, read once, a step later, 12 GB of them per 10 M reads, beside a
// table whose lines are not reused either and a taxonomy that IS (resolve_lane's node records).  SLK_STREAM_NT=1 marks their loads
// and stores nontemporal.
#ifndef SLK_STREAM_NT
#define SLK_STREAM_NT 0
#endif
#ifndef SLK_APPLY_ROWS
#define SLK_APPLY_ROWS 4
#endif
template <class T> __device__ __forceinline__ T SLK_STREAM_LOAD(const T *p) {
#if SLK_STREAM_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <class T> __device__ __forceinline__ void SLK_STREAM_STORE(T *p, T v) {
#if SLK_STREAM_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// ---- table-sharded mode (engine.h: ShardIO, ApplyJob) -------------------------------------------------------------------
// The step kernel runs the local kernel's scan, so it forms the same probe batches in the same order; instead of probing them it
// appends a batch's keys to the send regions of their owners and logs where each owner's group went.  Nothing but 8-byte keys,
// 4-byte taxa and the log touches HBM: no per-probe slot addresses, no scatter of the answers, no compaction of the lists.
__device__ __forceinline__ uint64_t lane_readlane64(uint64_t v, int src) {
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t shard_owner(uint64_t key, uint32_t ns) {  // == slk_shard_of
  const uint64_t h = fmix64(key);
  return (ns & (ns - 1)) == 0 ? (uint32_t)(h & (ns - 1)) : (uint32_t)(h % ns);
}
// One key of an earlier batch's lookups per lane, loaded a batch ahead of its use (engine.h: ShardIO.side_*)
__device__ __forceinline__ uint64_t side_load(const ShardIO &S, uint64_t batch, int lane) {
  const uint64_t i = batch * 64 + (uint64_t)lane;
  return i < S.side_n ? (uint64_t)SLK_STREAM_LOAD(&S.side_keys[i]) : 0;
}
// LOOKUP job: 64 of the keys this rank received for an earlier batch, probed with probe_batch's access shape (LPB lanes per bucket,
// all of the batch's loads in flight before the first compare); a key whose bucket is full, flagged and does not hold it goes on
// alone (as in shard.hip's lookup_coop_kernel).  `key` was loaded a batch ago (side_load).
__device__ __forceinline__ void side_probe(LaneLds *L, const TableView &T, const ShardIO &S, int lane, uint64_t batch, uint64_t key, int dbg) {
  const uint64_t base = batch * 64;
  if (base >= S.side_n) return;
  const uint64_t i = base + (uint64_t)lane;
  const bool in = i < S.side_n;
  const uint64_t h = fmix64(key);
  uint32_t home;
  uint64_t rem_hi;
  table_slot(T.g, h, home, rem_hi);
  const uint64_t tag = in ? rem_hi : ~0ULL;
  uint4 st;
  st.x = home; st.y = 0; st.z = (uint32_t)tag; st.w = (uint32_t)(tag >> 32);
  ((uint4 *)L->stash)[lane] = st;
  L->found[lane] = 0;
  lane_wave_sync();
  const int g = lane / LPB, c = lane % LPB;
  const uint64_t tmask = (1ULL << T.g.taxon_bits) - 1;
  const char *cellbase = (const char *)T.cells + c * 16;
  ulonglong2 cell[LPB];
#pragma unroll
  for (int s = 0; s < LPB; s++) {
    uint32_t bkt = ((const uint4 *)L->stash)[s * PG + g].x;
    SLK_TUNE(if (dbg & 4) bkt &= 1023u;)                                  // (timing experiment 4: every probe hits the L2)
    cell[s] = SLK_PROBE_LOAD((const ulonglong2 *)(cellbase + ((uint64_t)bkt << BUCKET_SHIFT)));
  }
  uint32_t unresolved = 0;
#pragma unroll
  for (int s = 0; s < LPB; s++) {
    const uint2 tg = *(const uint2 *)&((const uint4 *)L->stash)[s * PG + g].z;
    const uint64_t want = ((uint64_t)tg.y << 32) | tg.x;
    const bool act = want != ~0ULL;
    const bool e0 = cell[s].x == 0, e1 = cell[s].y == 0;
    const bool m0 = act && !e0 && cell_tag(T.g, cell[s].x) == want;
    const bool m1 = act && !e1 && cell_tag(T.g, cell[s].y) == want;
    if (m0 || m1) L->found[s * PG + g] = (uint32_t)((m0 ? cell[s].x : cell[s].y) & tmask);
    const bool closed = T.g.flag != 0 && c == 0 && (cell[s].x & T.g.flag) == 0;
    const uint64_t B = __ballot(m0 || m1 || e0 || e1 || !act || closed);
    if (((B >> (g * LPB)) & ((1u << LPB) - 1)) == 0) unresolved |= 1u << s;
  }
  lane_wave_sync();
  int32_t taxon = (int32_t)L->found[lane];
  const uint32_t ur = (uint32_t)__shfl((int)unresolved, (lane % PG) * LPB);   // entry `lane` was group lane % PG of step lane / PG
  if (in && ((ur >> (lane / PG)) & 1)) {
    for (int d = 1; d <= T.max_disp; d++) {
      const ulonglong2 *b = (const ulonglong2 *)(T.cells + ((uint64_t)table_bucket(T.g, home, (uint32_t)d) * CELLS));
      const uint64_t want = rem_hi | (uint64_t)d;
      bool has_empty = false, closed = false;
      int32_t hit = 0;
#pragma unroll
      for (int q = 0; q < LPB; q++) {
        const ulonglong2 v = b[q];
        has_empty |= (v.x == 0) | (v.y == 0);
        if (q == 0) closed = T.g.flag != 0 && (v.x & T.g.flag) == 0;
        if (v.x != 0 && cell_tag(T.g, v.x) == want) hit = (int32_t)(v.x & tmask);
        if (v.y != 0 && cell_tag(T.g, v.y) == want) hit = (int32_t)(v.y & tmask);
      }
      if (hit) { taxon = hit; break; }
      if (has_empty || closed) break;
    }
  }
  if (in && !SLK_TUNE_ON(128)) SLK_STREAM_STORE(&S.side_out[i], ext_taxon(T, taxon));   // (128: timing experiment, the answers are not written)
  lane_wave_sync();
}

// APPLY job: the tile's rows of the log, four at a time -- lane i takes the i-th entry of a logged probe batch in (owner, rank)
// order, reads its span metadata from this rank's meta region and its taxon from the owners' answers (which lie where the keys
// lay), and folds it into the owner lane's map exactly as the local kernel does.  The rows' loads are independent of each other
// (log entry -> meta and taxon), so four rows' worth are in flight before the first fold: a row at a time the replay is two
// dependent round trips per 64 probes, which is what it cost when it rode along batch by batch (15.3 ms per step against 12.9
// without it; profiles/r04_table_sharded_v3_first.json).
template <bool HITS>
__device__ __forceinline__ void apply_rows(LaneLds *L, const ApplyJob &J, int lane, uint64_t row0, uint32_t nrows) {
  const uint32_t ns = (uint32_t)J.n_shards;
  constexpr int U = SLK_APPLY_ROWS;
  for (uint32_t b = 0; b < nrows; b += U) {
    uint64_t at[U];
    bool in[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      in[u] = false;
      at[u] = 0;
      if (b + u < nrows) {
        uint32_t off = 0;
        for (uint32_t sh = 0; sh < ns; sh++) {
          const uint4 e = J.batch_log[(row0 + b + u) * ns + sh];        // (wave-uniform loads)
          const uint32_t cnt = e.z;
          if (cnt == 0) continue;
          if ((uint32_t)lane >= off && (uint32_t)lane < off + cnt) {
            in[u] = true;
            const uint32_t i = (uint32_t)lane - off;
            at[u] = (uint64_t)sh * J.cap + (i < e.w ? e.x + i : e.y + (i - e.w));
          }
          off += cnt;
        }
      }
    }
    uint32_t meta[U];
    int32_t taxon[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      meta[u] = in[u] ? SLK_STREAM_LOAD(&J.send_meta[at[u]]) : 0u;
      taxon[u] = in[u] ? SLK_STREAM_LOAD(&J.taxa[at[u]]) : 0;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (b + u < nrows) {   // (wave-uniform)
        int32_t t = taxon[u];
        if (in[u]) {
          if (HITS) {   // the un-merged hit list, as probe_batch writes it (the owners answer in the caller's ids)
            const uint64_t hat = L->rb[meta[u] & 63] + (meta[u] >> 20);
            J.A.span_taxon[hat] = t;
            J.A.span_meta[hat] = pack_meta((int32_t)((meta[u] >> 7) & 0x1FFF), 1, (meta[u] >> 6) & 1);
          }
          if (J.to_dense != nullptr && t > 0) t = t < J.n_to_dense ? J.to_dense[t] : 0;  // (owners answer in the caller's ids)
        }
        fold_hit<false>(L, nullptr, in[u], meta[u], t);
      }
    }
    lane_wave_sync();
  }
}

// EMIT job, one probe batch: its minimizers go to their owners' send regions, the span metadata the APPLY needs (owner lane,
// distinct, k-mers) to a region of the same shape, which stays on this rank.  Lane sh keeps the chunk of owner sh's region this wave
// is filling (ch_pos .. ch_end); a batch whose keys for that owner do not fit what is left of it takes the rest of the chunk AND the
// head of a freshly reserved one (one atomic per chunk and owner), so the regions have no holes but the chunk tails at the kernel's
// end.  Between the atomic and the use of its answer runs the batch's side job -- 64 lookups of an earlier batch -- so the cursor's
// round trip and the lookups' share one wait.
template <bool HITS>
__device__ __forceinline__ void emit_batch(LaneLds *L, const FusedArgs &A, const ShardIO &S, const ApplyJob &J, int qhead, int cnt, int lane,
                                           uint64_t row, uint64_t tile, uint32_t &ch_pos, uint32_t &ch_end, uint32_t &side_j, uint64_t &side_key, int dbg) {
  SLK_TUNE(if (dbg & 256) return;)                        // (timing experiment 256: the scan alone, its keys dropped)
  const bool in = lane < cnt;
  const int qi = (qhead + lane) & (QCAP - 1);
  const uint64_t key = L->q_key[qi];
  // (nothing is re-queued in this mode, so the displacement field of the entry is free: with hit lists it carries the span's
  //  ordinal -- a fragment this kernel takes has fewer than 1000 spans -- for the APPLY to write the hit where it belongs)
  const uint32_t meta = L->q_meta[qi] | (HITS ? (uint32_t)L->q_ord[qi] << 20 : 0u);
  const uint32_t ns = (uint32_t)S.n_shards;
  const uint32_t g = ns == 1 ? 0u : shard_owner(key, ns);
  uint32_t mycount = 0, rank = 0;
  for (uint32_t sh = 0; sh < ns; sh++) {
    const uint64_t m = __ballot(in && g == sh);
    if ((uint32_t)lane == sh) mycount = (uint32_t)__popcll(m);
    if (g == sh) rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
  }
  const uint32_t pos = ch_pos, room = ch_end - ch_pos;   // (lanes < ns: this wave's chunk of owner `lane`)
  const bool need = (uint32_t)lane < ns && mycount > room;
  unsigned long long got = 0;
  if (need) got = atomicAdd(&S.cursors[lane], (unsigned long long)S.chunk);
  if (side_j < S.side_per_tile) {   // LOOKUP job: one batch of an earlier batch's lookups per batch of keys sent off
    const uint64_t batch = tile * S.side_per_tile + side_j;
    const uint64_t k_now = side_key;
    side_j++;
    if (side_j < S.side_per_tile) side_key = side_load(S, batch + 1, lane);   // (in flight while this batch is probed)
    side_probe(L, A.T, S, lane, batch, k_now, dbg);
  }
  uint32_t fresh = 0xFFFFFFFFu;
  if (need) {
    if (got + S.chunk <= S.cap) fresh = (uint32_t)got;
    else atomicOr(A.status, 2);                       // the region is full: the host emits the batch again with larger regions
  }
  if ((uint32_t)lane < ns) {
    S.batch_log[row * ns + (uint32_t)lane] = make_uint4(pos, fresh, mycount, room);
    if (need) { ch_pos = fresh == 0xFFFFFFFFu ? 0u : fresh + (mycount - room); ch_end = fresh == 0xFFFFFFFFu ? 0u : fresh + S.chunk; }
    else ch_pos = pos + mycount;
  }
  const uint32_t gpos = (uint32_t)__shfl((int)pos, (int)g), groom = (uint32_t)__shfl((int)room, (int)g), gfresh = (uint32_t)__shfl((int)fresh, (int)g);
  if (in) {
    const bool tail = rank >= groom;                  // beyond what the old chunk had left: the fresh one
    if ((!tail || gfresh != 0xFFFFFFFFu) && !SLK_TUNE_ON(32)) {   // (32: timing experiment, keys and metadata are not written)
      const uint64_t at = (uint64_t)g * S.cap + (tail ? gfresh + (rank - groom) : gpos + rank);
      SLK_STREAM_STORE(&S.send_keys[at], (int64_t)key);
      SLK_STREAM_STORE(&S.send_meta[at], meta);
    }
  }
  lane_wave_sync();
}

// {parent, tin, tout, -} of taxon t (engine.h: FusedArgs.nodes); an id outside the taxonomy is a tree of its own
__device__ __forceinline__ uint4 lane_node(const uint4 *nodes, int32_t ntax, int32_t t) {
  return ((uint32_t)t < (uint32_t)ntax) ? nodes[t] : make_uint4(0u, 0x40000000u + (uint32_t)t, 0x40000000u + (uint32_t)t, 0u);
}
// BitRepresentation.charToTwobit (BitRepresentation.scala:127-135) for one character: 0..3, or 5 for anything else
__device__ __forceinline__ int lane_code(uint32_t c) {
  const uint32_t VM = (1u << 1) | (1u << 3) | (1u << 7) | (1u << 20) | (1u << 21);  // A C G T U, either case
  bool ok = ((c & 0xC0) == 0x40) && ((VM >> (c & 31)) & 1);
  uint32_t t = (c >> 1) & 3;  // A,C,T/U,G -> 0,1,2,3
  t ^= t >> 1;                // -> A=0 C=1 G=2 T=3
  return ok ? (int)t : 5;
}

// Read stream.  A lane consumes its read 16 bytes at a time; fetching those 16 bytes alone every 16 steps asks the L2 for
// every 64-byte line about four times, and with 20 waves x 64 lanes per CU the lines do not survive in the L2 between two
// requests (PMC: ~1.0e8 of 5.2e8 fabric reads per launch were re-fetched read bytes).  So a refill fetches SBLK sub-blocks
// back to back (the L1 merges requests to a line that is already on its way), keeps the first in registers and parks the
// others in the lane's own LDS slots.  Measured per 10 M x 150 bp launch: 16 bytes per refill 9.28 ms, 48: 9.0, 64: 8.8,
// 80: 8.5 (two refills per 150-base read), 96: 8.6, 112 and more: slower (the LDS they take costs resident waves).  Sub-blocks
// starting at or beyond the end of the read are not fetched, and the block that holds the last bytes of the caller's buffer
// is assembled from byte loads (load_block16; `room` = bytes from seq to the end of the buffer): nothing outside the buffer
// is touched.
__device__ __forceinline__ uint4 stream_refill(LaneLds *L, int lane, const uint8_t *seq, uint32_t p, uint32_t n, uint32_t room) {
  uint4 v[SBLK];
#pragma unroll
  for (int i = 0; i < SBLK; i++) {
    v[i] = make_uint4(0, 0, 0, 0);
    if (p + 16u * i < n) v[i] = load_block16(seq + p + 16u * i, room - (p + 16u * i));
  }
#pragma unroll
  for (int i = 1; i < SBLK; i++) L->sbuf[(i - 1) * 64 + lane] = v[i];
  return v[0];
}

// ---- the read stream as whole 128-byte lines (SLK_PACKED_STREAM) ----------------------------------------------------------------
// The lanes of a tile read 64 fragments that lie back to back in the caller's buffer.  Fetched lane by lane (stream_refill) those
// bytes reach the memory system as one 64-byte request per sector a lane touches: 2.3e7 of the 4.2e8 requests of a 10 M x 150 bp
// launch, each at the price of a table probe (the part serves requests, not bytes: DESIGN.md 4).  A tile whose fragments span at most
// PACK_MAX_SPAN bytes (64 fragments of up to 170 bases) is instead fetched by the whole wave at once -- 16 bytes per lane, 1 KiB per
// instruction, whole lines, half the requests -- and kept for the length of the scan in the LDS slots of the lane-wise stream, which
// it fits as 3 bits per base: 2-bit codes (BitRepresentation.charToTwobit, BitRepresentation.scala:127-135) and a validity bit, 16
// bases per word.  A lane then takes 16 bases per refill from the two words its position straddles; the step body reads a code
// and a validity bit where the lane-wise stream decodes a character.  Longer tiles, pairs and the long variant keep the lane-wise
// stream.
#ifndef SLK_PACKED_STREAM
#define SLK_PACKED_STREAM 1
#endif
#ifndef SLK_PACK_ROUNDS
#define SLK_PACK_ROUNDS 2
#endif
constexpr uint32_t PACK_MAX_SPAN = 10880;                       // bytes of a tile that can be staged (680 words of 16 bases)
constexpr uint32_t PACK_WORDS = PACK_MAX_SPAN / 16 + 1;         // + one word of padding: a refill reads two words
constexpr int PACK_LOADS = (PACK_MAX_SPAN + 1023) / 1024;       // wave-wide 1 KiB loads per tile, at most
static_assert((size_t)PACK_WORDS * 6 <= (size_t)(SBLK - 1) * 64 * 16, "the packed tile lives in the lane-wise stream's LDS slots");
// 16 characters -> (2-bit codes, char i in bits 2i..2i+1; validity bits, char i in bit i)
__device__ __forceinline__ void pack16(const uint4 v, uint32_t &codes, uint32_t &valid) {
  const uint32_t VM = (1u << 1) | (1u << 3) | (1u << 7) | (1u << 20) | (1u << 21);  // A C G T U, either case
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  codes = 0; valid = 0;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    uint32_t t = (w[d] >> 1) & 0x03030303u;       // A,C,T/U,G -> 0,1,2,3 per byte
    t ^= (t >> 1) & 0x01010101u;                  // -> A=0 C=1 G=2 T=3 (BitRepresentation.scala:35-39)
    t = (t | (t >> 6) | (t >> 12) | (t >> 18)) & 0xFFu;
    codes |= t << (8 * d);
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const uint32_t c = (w[d] >> (8 * b)) & 0xFFu;
      const uint32_t ok = (((c & 0xC0u) == 0x40u) ? 1u : 0u) & (VM >> (c & 31u));   // BitRepresentation.isValid :140-143
      valid |= (ok & 1u) << (4 * d + b);
    }
  }
}
// the 16 bases from position s of the packed tile on: codes in .x, validity bits in .y
__device__ __forceinline__ uint2 packed_take(const uint32_t *pcodes, const uint16_t *pvalid, uint32_t s) {
  const uint32_t wi = s >> 4, bo = s & 15u;
  const uint64_t c64 = (uint64_t)pcodes[wi] | ((uint64_t)pcodes[wi + 1] << 32);
  const uint32_t v32 = (uint32_t)pvalid[wi] | ((uint32_t)pvalid[wi + 1] << 16);
  return make_uint2((uint32_t)(c64 >> (2 * bo)), (v32 >> bo) & 0xFFFFu);
}

#ifndef SLK_LANE_WPS
#define SLK_LANE_WPS 0
#endif
#if SLK_LANE_WPS > 0
#define LANE_BOUNDS __launch_bounds__(LW * 64, SLK_LANE_WPS)
#else
// (at least four waves per SIMD -- what the LDS footprint allows anyway --, i.e. at most 128 VGPRs: the emit variant with its side
//  job came out at 129 and lost a quarter of its resident waves)
#define LANE_BOUNDS __launch_bounds__(LW * 64) __attribute__((amdgpu_waves_per_eu(4)))
#endif

// A wave of the first pass (or of the long variant) hands fragments on: straight into the hand-on list of the kernel that takes such
// a fragment (the order inside a list does not matter); one atomic per wave and list that gets something.
// hand-on list of a fragment the first pass does not take for its length (engine.h: FusedArgs.hand_hdr)
__device__ __forceinline__ int route_by_length(uint64_t len, uint32_t long_max, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t seg_min_len,
                                               uint32_t wave_min, uint32_t wave_ratio_q10) {
  if (long_max != 0 && len <= long_max) return (len > b0) + (len > b1) + (len > b2);
  if (seg_min_len != 0 && len >= seg_min_len) return HandOn::SEG;
  int c = HandOn::WAVE_CLASSES - 1;                         // the wave kernel's long fragments, by length: a geometric ladder
  for (uint64_t b = ((uint64_t)wave_min * wave_ratio_q10) >> 10; c > 0 && len >= b; b = (b * wave_ratio_q10) >> 10) c--;
  return HandOn::WAVE0 + c;
}
__device__ __attribute__((noinline)) void hand_on(unsigned long long *hdr, uint32_t *lists, uint64_t stride, uint64_t long_cap, uint32_t long_max,
                                                  uint32_t b0, uint32_t b1, uint32_t b2, uint32_t seg_min_len, uint32_t wave_min,
                                                  uint32_t wave_ratio_q10, const uint64_t *offsets, const uint64_t *mate_offsets, bool dfr,
                                                  bool too_long, uint64_t r, int lane, bool from_long) {
  const uint64_t DM = __ballot(dfr);
  int route = from_long ? HandOn::LATE : HandOn::REST;      // map overflow (of the long variant | of the first pass): the wave kernel
  if (!from_long && dfr && too_long) {
    uint64_t len = offsets[r + 1] - offsets[r];
    if (mate_offsets) len += mate_offsets[r + 1] - mate_offsets[r];
    route = route_by_length(len, long_max, b0, b1, b2, seg_min_len, wave_min, wave_ratio_q10);
  }
  if (!from_long && lane == 0) atomicAdd(&hdr[HandOn::HANDED], (unsigned long long)__popcll(DM));
  for (int l = from_long ? HandOn::LATE : 0; l <= HandOn::LATE; l++) {
    const uint64_t M = __ballot(dfr && route == l);
    if (M == 0) continue;
    const int leader = __ffsll((long long)M) - 1;
    unsigned long long at = 0;
    if (lane == leader) at = atomicAdd(&hdr[HandOn::count_word(l)], (unsigned long long)__popcll(M));
    at = lane_readlane64(at, leader);
    if (dfr && route == l)
      lists[HandOn::list_at(l, stride, long_cap) + at + __builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0))] = (uint32_t)r;
  }
}

// The routing kernel (engine.h: FusedArgs.route_first): what a first pass does with the fragments it does not take, for all of
// them.  A block sorts 2048 fragments at a time: places inside the block from LDS counters, one atomic per list and block on the
// header (a wave at a time, like hand_on, the 3 500 waves of a 224 k-fragment batch queued up at eight addresses for 0.26 ms).
__global__ void __launch_bounds__(256) route_kernel(FusedArgs A) {
  constexpr int PER = 8, NL = HandOn::LISTS;
  __shared__ uint32_t cnt[NL];
  __shared__ unsigned long long base[NL];
  const uint64_t chunk = 256 * PER, chunks = (A.R + chunk - 1) / chunk;
  for (uint64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
    if (threadIdx.x < NL) cnt[threadIdx.x] = 0;
    __syncthreads();
    int route[PER];
    uint32_t rank[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const uint64_t r = c * chunk + (uint64_t)j * 256 + threadIdx.x;
      route[j] = -1;
      rank[j] = 0;
      if (r < A.R) {
        uint64_t len = A.offsets[r + 1] - A.offsets[r];
        if (A.mate_offsets) len += A.mate_offsets[r + 1] - A.mate_offsets[r];
        route[j] = len <= 1000 ? (int)HandOn::SHORT : route_by_length(len, A.long_max, A.long_bound[0], A.long_bound[1], A.long_bound[2],
                                                                      A.seg_min_len, A.wave_min, A.wave_ratio_q10);
        rank[j] = atomicAdd(&cnt[route[j]], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < NL && cnt[threadIdx.x] != 0)
      base[threadIdx.x] = atomicAdd(&A.hand_hdr[HandOn::count_word((int)threadIdx.x)], (unsigned long long)cnt[threadIdx.x]);
    if (threadIdx.x == 0) {
      uint32_t handed = 0;
      for (int l = 0; l < NL; l++) if (l != HandOn::SHORT) handed += cnt[l];
      if (handed) atomicAdd(&A.hand_hdr[HandOn::HANDED], (unsigned long long)handed);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; j++)
      if (route[j] >= 0)
        A.hand_lists[HandOn::list_at(route[j], A.hand_stride, A.hand_long_cap) + base[route[j]] + rank[j]] =
            (uint32_t)(c * chunk + (uint64_t)j * 256 + threadIdx.x);
    __syncthreads();
  }
}
void launch_route(const FusedArgs &A, hipStream_t s) {
  if (A.R == 0) return;
  hipLaunchKernelGGL(route_kernel, dim3((unsigned)std::min<uint64_t>((A.R + 2047) / 2048, 2048)), dim3(256), 0, s, A);
}

// Per-read classification, one lane per read: the map the probe batches (or the APPLY job's replay) folded for the lane's fragment
// -> TaxonCounts / resolveTree / the minHitGroups test -> the fragment's output rows.  oflags = the lane's o_flags word.
template <bool HITS, bool LONG>
__device__ __forceinline__ void resolve_lane(LaneLds *L, uint32_t *ocnt, const FusedArgs &A, int lane, uint64_t r, uint32_t oflags, int32_t total,
                                             int32_t nhits, int32_t np, int dbg) {
  const int32_t nd = (int32_t)oflags;
  // The map's entries move to the front of the lane's column (their hash order has served its purpose): entry j of D.
  uint32_t *const ecnt = LONG ? ocnt : L->omap;
  int D = 0;
  for (int s = 0; s < OMAP; s++) {
    const uint32_t e = L->omap[s * 64 + lane];
    if (e != 0) {
      if (LONG) ocnt[D * 64 + lane] = ocnt[s * 64 + lane];
      L->omap[D * 64 + lane] = e;
      D++;
    }
  }
#define ENT_TAXON(j) (LONG ? (int32_t)L->omap[(j) * 64 + lane] : (int32_t)(L->omap[(j) * 64 + lane] >> OMAP_CNT_BITS))
#define ENT_COUNT(j) (LONG ? (int32_t)ecnt[(j) * 64 + lane] : (int32_t)(ecnt[(j) * 64 + lane] & OMAP_CNT_MASK))
  int32_t maxTaxon = D ? ENT_TAXON(0) : 0;  // D <= 1: the single taxon (or NONE)
  const int32_t c0 = D ? ENT_COUNT(0) : 0;
  // D >= 2: resolveTree on Euler-tour intervals (engine.h: FusedArgs.nodes).  One 16-byte load per map taxon, issued back
  // to back, brings its interval; "is a an ancestor-or-self of b" is then two compares, for the root-path scores (step 1)
  // as for the clade sums of the confidence walk (step 2).  The intervals live in the probe queue's LDS, idle by now.
  uint32_t *const tin = (uint32_t *)L, *const tout = tin + OMAP * 64;
  static_assert(offsetof(LaneLds, omap) >= 2 * OMAP * 64 * sizeof(uint32_t), "the intervals alias the queue and the read stream's slots");
  uint32_t m_in = 0, m_out = 0;   // maxTaxon's interval
  int32_t sum_all = c0;
  if (D >= 2 && !SLK_TUNE_ON(16)) {  // (16: timing experiment)
#pragma unroll
    for (int j = 0; j < OMAP; j++) {
      if (j < D) {
        const uint4 nj = lane_node(A.nodes, A.ntax, ENT_TAXON(j));
        tin[j * 64 + lane] = nj.y;
        tout[j * 64 + lane] = nj.z;
      }
    }
    // step 1 (:101-123): the LCA of the taxa with the maximal root-path score -- a taxon's score is the k-mer count of the
    // map taxa on its root path, i.e. of the entries whose interval holds its tin
    maxTaxon = 0;
    sum_all = 0;
    int32_t best = 0;
    for (int a = 0; a < D; a++) {
      const uint32_t ain = tin[a * 64 + lane], aout = tout[a * 64 + lane];
      int32_t score = 0;
      for (int b = 0; b < D; b++)
        score += (tin[b * 64 + lane] <= ain && ain <= tout[b * 64 + lane]) ? ENT_COUNT(b) : 0;
      sum_all += ENT_COUNT(a);
      if (score > best) {
        maxTaxon = ENT_TAXON(a); best = score; m_in = ain; m_out = aout;
      } else if (score == best) {   // LowestCommonAncestor.apply :49-78 of (maxTaxon, this taxon)
        if (m_in <= ain && ain <= m_out) {
          // maxTaxon is an ancestor-or-self of this taxon: it stays
        } else if (ain <= m_in && m_in <= aout) {
          maxTaxon = ENT_TAXON(a); m_in = ain; m_out = aout;
        } else {                   // neither: the first node above maxTaxon whose interval holds this taxon
          int32_t x = (int32_t)lane_node(A.nodes, A.ntax, maxTaxon).x;
          uint4 nx = make_uint4(0, 0, 0, 0);
          while (x != 0) {
            nx = lane_node(A.nodes, A.ntax, x);
            if (nx.y <= ain && ain <= nx.z) break;
            x = (int32_t)nx.x;
          }
          if (x == 0) { x = 1; nx = lane_node(A.nodes, A.ntax, 1); }   // no common node: ROOT (:77)
          maxTaxon = x; m_in = nx.y; m_out = nx.z;
        }
      }
    }
  }
  for (int32_t c = 0; c < A.C; c++) {
    const double required = ceil(__dmul_rn(A.thr.v[c], (double)total));  // Math.ceil(confidence * totalKmers) :94
    int32_t mt = maxTaxon;
    if (D < 2 || SLK_TUNE_ON(16)) {
      // one taxon: its clade sum never grows (NONE is in no clade), so it is the call or there is none (:125-144)
      if ((double)c0 < required) mt = 0;
    } else {
      // step 2 (:125-144): from maxTaxon towards the root until the clade of the candidate holds `required` k-mers of the
      // map.  The clade sum only changes where the candidate's interval comes to hold another map taxon, and once it holds
      // them all no ancestor can do better: the walk ends there (the reference goes on to the root and finds nothing).
      uint32_t cin = m_in, cout = m_out;
      uint4 cur = make_uint4(0, 0, 0, 0);
      bool have_cur = false;
      while (mt != 0) {
        int32_t ms = 0;
        bool side = false;          // a map taxon outside the clade that is NOT an ancestor of the candidate
        int up = -1;                // the nearest map taxon above the candidate
        uint32_t up_in = 0;
        for (int j = 0; j < D; j++) {
          const uint32_t jin = tin[j * 64 + lane], jout = tout[j * 64 + lane];
          const bool inside = cin <= jin && jin <= cout;
          ms += inside ? ENT_COUNT(j) : 0;
          const bool above = !inside && jin <= cin && cin <= jout;
          side = side || (!inside && !above);
          if (above && (up < 0 || jin > up_in)) { up = j; up_in = jin; }   // (deeper on one root path = later in the tour)
        }
        if ((double)ms >= required) break;
        if (ms == sum_all) { mt = 0; break; }
        if (!side) {
          // everything left lies above the candidate, on its root path: the next clade that differs is the nearest of them
          mt = ENT_TAXON(up); cin = up_in; cout = tout[up * 64 + lane];
          have_cur = false;
        } else {
          if (!have_cur) cur = lane_node(A.nodes, A.ntax, mt);
          mt = (int32_t)cur.x;                                           // Taxonomy.parents
          if (mt != 0) { cur = lane_node(A.nodes, A.ntax, mt); have_cur = true; cin = cur.y; cout = cur.z; }
        }
      }
    }
    bool classified = (mt != 0) && (nd >= A.min_hit_groups);            // Classifier.scala:445
    A.out_taxon[(uint64_t)c * A.out_stride + r] = classified ? ext_taxon(A.T, mt) : 0;
    A.out_classified[(uint64_t)c * A.out_stride + r] = classified ? 1 : 0;
  }
#undef ENT_TAXON
#undef ENT_COUNT
  if (A.out_nd) A.out_nd[r] = nd;
  if (A.out_tk) A.out_tk[r] = total;
  if (A.out_nh) A.out_nh[r] = nhits;
  if (HITS) A.span_count[r] = nhits;
  if (A.out_np) A.out_np[r] = np;
}

// LONG: the second pass, over the fragments of 1 001 .. A.long_max bases that the first one handed on (hand-on lists 0..3, one per
// length class), with a map of full 32-bit counts (such a fragment has up to 4 965 k-mers for one taxon; the one-word map of the hot
// variant counts to 1 023).  What overflows its map too goes on to the wave kernel's list.  The hot variant is untouched by it.
// MODE == LANE_EMIT (lane_step_kernel): the table-sharded step -- the scan sends its minimizers off instead of probing (EMIT job of
// batch A), and an earlier batch's lookups (S.side_*) and a yet earlier batch's classification (J) ride along (engine.h: ShardIO).
template <bool W5, int MODE, bool HITS, bool LONG>
__device__ __forceinline__ void lane_body(const FusedArgs &A, const ShardIO &S, const ApplyJob *Jp, int32_t *defer, uint32_t max_len, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const ScanParams P = A.P;
  const int k = P.k, m = P.m, w = P.w;
  // per-wave LDS: fixed part, then (generic w only) the window array, [w][64]: the keys of the current w-block in its first
  // rows, the suffix minima of the previous block in the rows the current block has not reached yet
  // (the hit-list fields are the struct's tail: kernels that do not write hit lists leave them out of their footprint)
  const size_t fixed = HITS ? sizeof(LaneLds) : offsetof(LaneLds, rb);
  const size_t win_bytes = W5 ? 0 : (size_t)w * 64 * sizeof(uint64_t);
  const size_t per_wave = fixed + win_bytes + (LONG ? (size_t)OMAP * 64 * sizeof(uint32_t) : 0);
  LaneLds *L = (LaneLds *)(lds_raw + (size_t)wib * per_wave);
  uint64_t *win = (uint64_t *)((unsigned char *)L + fixed);
  uint32_t *ocnt = LONG ? (uint32_t *)((unsigned char *)L + fixed + win_bytes) : nullptr;   // LONG: k-mer counts of the map's slots
  const bool paired = A.mate_bases != nullptr;
  const bool scans = MODE != LANE_EMIT || A.R != 0;   // (a step may carry side jobs only)
  const uint64_t bases_end = scans ? A.offsets[A.R] : 0, mates_end = (scans && paired) ? A.mate_offsets[A.R] : 0;  // (wave-uniform loads)
  // LONG: tiles of the four class lists one after the other (a tile never mixes classes; class 0 here: list 15, the fragments
  // of at most 1000 bases of a batch without a first pass); the first pass / the routing kernel has finished (same stream)
  uint64_t cls_n[5] = {0, 0, 0, 0, 0}, cls_tiles[5] = {0, 0, 0, 0, 0};
  uint64_t ntiles = (A.R + 63) / 64;
  if (LONG) {
    ntiles = 0;
#pragma unroll
    for (int c = 0; c < 5; c++) {
      cls_n[c] = c == 0 ? (A.route_first ? A.hand_hdr[HandOn::N_SHORT] : 0ULL) : A.hand_hdr[c - 1];
      cls_tiles[c] = (cls_n[c] + 63) / 64;
      ntiles += cls_tiles[c];
    }
  }
  const uint64_t nwaves = (uint64_t)gridDim.x * LW;
  // EMIT: tiles of the scan and tiles of the APPLY job (an earlier batch: its tile t rides with the scan's tile t; whichever batch
  // has more tiles, the rest go alone); the chunk of every owner's send region this wave is filling, in lane `owner`
  const uint64_t etiles = ntiles;
  const uint64_t atiles = MODE == LANE_EMIT ? (Jp->A.R + 63) / 64 : 0;
  if (MODE == LANE_EMIT) ntiles = max(etiles, atiles);
  uint32_t ch_pos = 0, ch_end = 0;

  // (The 64 lanes of a wave run in lockstep, so a tile lasts as long as its longest fragment.  Two ways of handing the tiles
  // fragments of similar length were measured in round 2 and dropped, profiles/r02_mixed_lengths.json: sorting a block's 256
  // fragments over its four waves gains nothing, because a block's LDS and wave slots are held until its longest wave ends;
  // a length-bucketed order inside windows of 16 384 fragments gains 4 % on lengths uniform in 50..250 -- where a globally
  // sorted input gains 20 % -- because every lane then pays scattered loads of its offsets and scattered stores of its results.)
  for (uint64_t tile = (uint64_t)blockIdx.x * LW + wib, it = 0;; it++) {
    if (LONG) {
      // The number of hand-ons is only known here, so the grid is a fixed one and the waves draw their tiles from a counter:
      // with a strided walk most waves would take floor(tiles / waves) tiles and a few one more, and everyone waits for those.
      // (A wave's first tile is the one of its own number: fewer tiles than waves -- usually none -- cost no atomics.)
      if (it) {
        unsigned long long t = 0;
        if (lane == 0) t = atomicAdd(&A.hand_hdr[HandOn::LONG_DRAW], 1ULL);
        tile = nwaves + lane_readlane64(t, 0);
      }
    } else if (MODE == LANE_EMIT) {
      // persistent waves (they carry their chunks from tile to tile), tiles drawn from a counter as above
      if (it) {
        unsigned long long t = 0;
        if (lane == 0) t = atomicAdd(&S.cursors[S.n_shards], 1ULL);
        tile = nwaves + lane_readlane64(t, 0);
      }
    } else if (it) {
      tile += nwaves;
    }
    if (tile >= ntiles) break;
    uint64_t unit = tile * 64 + lane;
    bool have = unit < A.R;
    uint64_t r = unit;
    if (LONG) {
      uint64_t local = tile;   // (the longest class first: the last tiles to start are the shortest)
      int c = 4;
#pragma unroll
      for (int i = 4; i > 0; i--) if (c == i && local >= cls_tiles[i]) { local -= cls_tiles[i]; c = i - 1; }
      unit = local * 64 + lane;
      have = unit < cls_n[c];
      const uint32_t *const cl = A.hand_lists + HandOn::list_at(c == 0 ? (int)HandOn::SHORT : c - 1, A.hand_stride, A.hand_long_cap);
      r = have ? (uint64_t)cl[unit] : 0;
    }
    // ---- fragment descriptor ----
    const uint8_t *seq = A.bases;
    uint32_t n = 0, n2 = 0;
    uint32_t room = 0;  // bytes from the fragment's first base to the end of the caller's buffer
    if (have) {
      uint64_t o = A.offsets[r];
      seq = A.bases + o;
      n = (uint32_t)(A.offsets[r + 1] - o);
      room = clamp_room(bases_end - o);
      SLK_TUNE(if (dbg & 8) { seq = A.bases + A.offsets[r & 1023]; room = clamp_room(bases_end - A.offsets[r & 1023]); })  // (timing experiment 8: the read stream comes from the L2)
      if (paired) n2 = (uint32_t)(A.mate_offsets[r + 1] - A.mate_offsets[r]);  // (the mate's place is read again when the scan gets there)
    }
    bool too_long = have && ((uint64_t)n + n2 > max_len);
    bool fin = !have || too_long;
    // EMIT: the tile's first row in the batch log.  A tile's probes number at most its bases, and tile t starts at
    // row floor(span_region(first fragment of t) / 64) + t: rows of different tiles never overlap (capi.hip: slk_shard_batch_rows)
    uint64_t row = 0, row0 = 0;
    if (MODE == LANE_EMIT && tile < etiles) row = row0 = (span_region(A.offsets, A.mate_offsets, tile * 64) >> 6) + tile;
    const uint64_t rbase = (HITS && have) ? span_region(A.offsets, A.mate_offsets, r) : 0;
    if (HITS && MODE != LANE_EMIT) L->rb[lane] = rbase;
    // ---- per-lane LDS state ----
#pragma unroll
    for (int s = 0; s < OMAP; s++) L->omap[s * 64 + lane] = 0;
    if (LONG) {
#pragma unroll
      for (int s = 0; s < OMAP; s++) ocnt[s * 64 + lane] = 0;
    }
    L->o_flags[lane] = 0;
    if (MODE == LANE_EMIT && tile < atiles) {
      // APPLY job first: the earlier batch's tile of the same number is replayed and classified before this tile's scan starts.  The
      // replay is bound by memory latency, the scan by instruction issue; the waves of a CU are at different points of their tiles,
      // so the one hides behind the other ACROSS waves, where riding along batch by batch made every wave wait three times per batch.
      const uint64_t r2 = tile * 64 + lane;
      const bool have2 = r2 < Jp->A.R;
      const uint2 tr = Jp->tile_rows[tile];   // (wave-uniform load) {first row of the tile in the log, rows}
      if (HITS) L->rb[lane] = have2 ? span_region(Jp->A.offsets, Jp->A.mate_offsets, r2) : 0;
      lane_wave_sync();                       // (the map's reset above and the span regions are seen by every lane)
      apply_rows<HITS>(L, *Jp, lane, tr.x, tr.y);
      const bool dfr2 = have2 && (Jp->defer[r2] != 0 || (L->o_flags[lane] & 0x80000000u));   // not taken by its EMIT, or its map overflowed just now
      const uint64_t DM = __ballot(dfr2);
      if (DM != 0 && lane == 0) atomicAdd(Jp->n_deferred, (unsigned long long)__popcll(DM));
      if (have2) {
        if (dfr2) {
          // (the caller routes this fragment through the staged kernels; until then it has no spans on file)
          Jp->defer[r2] = 1;
          if (Jp->A.out_nh) Jp->A.out_nh[r2] = 0;
          if (HITS) Jp->A.span_count[r2] = 0;
        } else {
          const int2 ri = Jp->read_info[r2];
          resolve_lane<HITS, false>(L, nullptr, Jp->A, lane, r2, L->o_flags[lane], ri.x, ri.y, 0, dbg);
        }
      }
      lane_wave_sync();                       // (resolve_lane's intervals lay over the queue and the read stream's slots: the scan below stages its first bytes there)
    }
    // ---- scan state ----
    uint32_t pos = 0;
    int mate = 0;
    uint32_t cur = 0, b1 = 0, b2 = 0, b3 = 0;  // 16 buffered characters
    int sb = 1;  // next staged sub-block (SBLK: none left, fetch)
    // a short unpaired tile is fetched by the whole wave and staged as codes (above); `cur` then holds 16 codes, `b1` 16 validity bits
    bool packed = false;
    uint32_t s0 = 0;   // packed: where this lane's fragment starts in the tile
    uint32_t *const pcodes = (uint32_t *)L->sbuf;
    uint16_t *const pvalid = (uint16_t *)(pcodes + PACK_WORDS);
    constexpr bool CAN_PACK = SLK_PACKED_STREAM && !LONG && MODE == LANE_LOCAL;   // (the sharded scan has no registers to spare for a second stream)
    if (CAN_PACK && !paired) {
      const uint64_t t0 = tile * 64, t1 = min(A.R, t0 + 64);
      const uint64_t span_o = A.offsets[t0], span = A.offsets[t1] - span_o;   // (wave-uniform loads)
      if (span <= PACK_MAX_SPAN && __ballot(!fin) != 0) {
        packed = true;
        s0 = have ? (uint32_t)(A.offsets[r] - span_o) : 0;
        const uint8_t *tb = A.bases + span_o;
        const uint32_t troom = clamp_room(bases_end - span_o), sp = (uint32_t)span;
        // (in SLK_PACK_ROUNDS rounds of loads: all eleven in flight at once cost the kernel 44 registers it does not have)
        constexpr int PG0 = (PACK_LOADS + SLK_PACK_ROUNDS - 1) / SLK_PACK_ROUNDS;
#pragma unroll
        for (int j0 = 0; j0 < PACK_LOADS; j0 += PG0) {
          uint4 v[PG0];
#pragma unroll
          for (int j = 0; j < PG0; j++) {
            const uint32_t cb = (uint32_t)(j0 + j) * 1024u + (uint32_t)lane * 16u;
            v[j] = make_uint4(0, 0, 0, 0);
            if (j0 + j < PACK_LOADS && cb < sp) v[j] = load_block16(tb + cb, troom - cb);
          }
#pragma unroll
          for (int j = 0; j < PG0; j++) {
            const uint32_t cb = (uint32_t)(j0 + j) * 1024u + (uint32_t)lane * 16u;
            if (j0 + j < PACK_LOADS && cb < sp + 16u) {       // (and the word behind the last one: a refill reads two)
              uint32_t cw, vw;
              pack16(v[j], cw, vw);
              pcodes[cb >> 4] = cw;
              pvalid[cb >> 4] = (uint16_t)vw;
            }
          }
        }
        lane_wave_sync();
      }
    }
    if (!fin && n > 0) {
      if (CAN_PACK && packed) {
        const uint2 pk = packed_take(pcodes, pvalid, s0);
        cur = pk.x; b1 = pk.y;
      } else {
        uint4 v = stream_refill(L, lane, seq, 0, n, room);
        cur = v.x; b1 = v.y; b2 = v.z; b3 = v.w;
      }
    }
    int run_class = 0;
    uint32_t run_len = 0, nvalid = 0;
    uint64_t fwd = 0, rc = 0;
    uint64_t k1 = ~0ULL, p1 = ~0ULL, p2 = ~0ULL, m41 = ~0ULL;  // W5: previous key, pair minima, previous 4-minimum
    uint64_t pre = ~0ULL;                                      // generic: prefix minimum of the current block
    uint64_t cur_val = 0;
    int32_t cur_run = 0;
    bool first = true, have_last = false;
    uint64_t last_key = 0;
    int32_t total = 0, np = 0, nhits = 0;
    uint32_t side_j = (MODE == LANE_EMIT && S.side_n != 0 && tile < etiles) ? 0u : 0xFFFFFFFFu;   // batches of the LOOKUP job done by this tile (wave-uniform)
    uint64_t side_key = (MODE == LANE_EMIT && S.side_n != 0 && S.side_per_tile != 0 && tile < etiles) ? side_load(S, tile * S.side_per_tile, lane) : 0;
    int qn = 0;       // queue fill (wave-uniform)
    int qhead = 0;    // ring position of the oldest queued entry (wave-uniform)
    int tphase = 0;   // generic window: step mod w (wave-uniform)

    const uint32_t VM = (1u << 1) | (1u << 3) | (1u << 7) | (1u << 20) | (1u << 21);  // A C G T U, either case
    while (__ballot(!fin) != 0) {
      // One event per lane per step: a character, or the end of a mate.  Straight-line predicated code: the per-read
      // control flow (Supermers.splitByAmbiguity :150-178, MinSplitter.splitRead :133-172) is data, not branches.
      if (!W5 && tphase == 0) pre = ~0ULL;  // a new w-block starts: empty prefix
      const bool act = !fin;
      const bool is_end = pos >= n;
      bool okc;
      uint32_t t;
      if (CAN_PACK && packed) {   // (wave-uniform) the tile was staged as codes and validity bits
        t = cur & 3u;
        okc = (b1 & 1u) != 0;
      } else {
        const uint32_t c = cur & 0xFF;
        okc = ((c & 0xC0) == 0x40) && ((VM >> (c & 31)) & 1);  // BitRepresentation.isValid :140-143
        t = (c >> 1) & 3;  // A,C,T/U,G -> 0,1,2,3
        t ^= t >> 1;       // -> A=0 C=1 G=2 T=3 (BitRepresentation.scala:35-39)
      }
      const int cls = is_end ? -1 : (okc ? 1 : 0);
      // -- does the current run end here?
      const bool run_end = act && run_len > 0 && cls != run_class;
      const bool seqrun = run_class == 1 && nvalid >= (uint32_t)k;
      const bool seq_close = run_end && seqrun;                              // last super-mer of a SEQUENCE_FLAG run
      const bool amb_close = run_end && !seqrun && run_len >= (uint32_t)k;   // AMBIGUOUS_FLAG run >= k: one span, no lookup
      const int32_t ord0 = nhits;  // ordinal of the span (if any) that closes in this step
      total += amb_close ? (int32_t)run_len - (k - 1) : 0;                   // Supermers.scala:116-119
      if (HITS && amb_close) {
        A.span_meta[rbase + ord0] = pack_meta((int32_t)run_len - (k - 1), 2, 0);
        A.span_taxon[rbase + ord0] = -1;                                     // spanToHit: AMBIGUOUS_SPAN
      }
      nhits += amb_close ? 1 : 0;
      first = first && !amb_close;
      const uint64_t ekey = cur_val;   // what a super-mer closing in this step carries
      const int32_t ekmers = cur_run;
      run_len = run_end ? 0u : run_len;
      // -- consume the character
      const bool proc = act && !is_end;
      const bool new_run = proc && run_len == 0;
      run_class = new_run ? cls : run_class;
      nvalid = new_run ? 0u : nvalid;
      cur_run = (new_run || seq_close) ? 0 : cur_run;
      // (the window state needs no reset: a window is only used once w keys of the new run have been seen)
      run_len += proc ? 1u : 0u;
      const bool nt = proc && okc;
      nvalid += nt ? 1u : 0u;
      // Rolled unconditionally: whatever a non-nucleotide step shifts in has left the m-mer again before the next key is
      // taken (a key needs m valid characters in a row, and both words hold exactly m bases).
      fwd = (fwd << 2) | ((uint64_t)t << P.sh);                               // NTBitArray.shiftLongArrayKmerLeft :140-150
      rc = ((rc >> 2) | ((uint64_t)(3 - t) << 62)) & P.keep;
      const bool havekey = nt && nvalid >= (uint32_t)m;
      const uint64_t canon = (P.canonical && rc < fwd) ? rc : fwd;             // NTBitArray.writeCanonical :258-266
      const uint64_t key = (canon ^ P.xmask) & P.smask;                        // RandomXOR, then SpacedSeed
      uint64_t minv;
      if (W5) {
        const uint64_t p = lmin64(key, k1);
        const uint64_t m4 = lmin64(p, p2);
        minv = lmin64(key, m41);  // minimum of the last five keys
        // pushed unconditionally: a window is only used once the last five pushes were keys of the current run
        k1 = key; p2 = p1; p1 = p; m41 = m4;
      } else {
        // van Herk: window = suffix of the previous w-block  U  prefix of the current one (blocks on the step counter)
        const uint64_t kk = havekey ? key : ~0ULL;
        pre = lmin64(pre, kk);
        minv = (tphase == w - 1) ? pre : lmin64(pre, win[(tphase + 1) * 64 + lane]);  // row tphase + 1: still the previous block's
        win[tphase * 64 + lane] = kk;                                                  // row tphase: its suffix minimum was used last step
      }
      const bool havewin = havekey && nvalid >= (uint32_t)k;  // a k-mer window is complete: its minimizer VALUE is minv
      const bool start = havewin && cur_run == 0;
      const bool same = havewin && cur_run != 0 && minv == cur_val;            // MinSplitter.splitRead :154-158
      const bool change = havewin && cur_run != 0 && minv != cur_val;
      const bool emit = seq_close || change;
      cur_val = (start || change) ? minv : cur_val;
      cur_run = (start || change) ? 1 : (same ? cur_run + 1 : cur_run);
      // -- advance the character stream
      pos += proc ? 1u : 0u;
      const bool refill = proc && (pos & 15) == 0;
      if (CAN_PACK && packed) {
        cur = proc ? (cur >> 2) : cur;
        b1 = proc ? (b1 >> 1) : b1;
        if (__ballot(refill) != 0) {
          if (refill && pos < n) {
            const uint2 pk = packed_take(pcodes, pvalid, s0 + pos);
            cur = pk.x; b1 = pk.y;
          }
        }
      } else {
        const bool nextdw = proc && (pos & 3) == 0;
        cur = nextdw ? b1 : (proc ? (cur >> 8) : cur);
        b1 = nextdw ? b2 : b1;
        b2 = nextdw ? b3 : b2;
        if (__ballot(refill) != 0) {
          if (refill) {
            if (pos < n) {
              uint4 v;
              if (sb < SBLK) { v = L->sbuf[(sb - 1) * 64 + lane]; sb++; }
              else { v = stream_refill(L, lane, seq, pos, n, room); sb = 1; }
              cur = v.x; b1 = v.y; b2 = v.z; b3 = v.w;
            }
          }
        }
      }
      // -- end of a mate (rare; wave-uniform for equal-length reads)
      const bool at_end = act && is_end;
      if (__ballot(at_end) != 0) {
        if (at_end) {
          if (mate == 0 && paired) {  // MATE_PAIR_BORDER pseudo-span (Supermers.scala:53-57): no k-mers, no lookup
            if (HITS) {  // it follows the span that closes in this step, if there is one
              const int32_t ordb = ord0 + ((amb_close || emit) ? 1 : 0);
              A.span_meta[rbase + ordb] = pack_meta(-(k - 1), 3, 0);
              A.span_taxon[rbase + ordb] = -2;
            }
            nhits++;
            first = false;
            mate = 1;
            const uint64_t o2 = A.mate_offsets[r];
            seq = A.mate_bases + o2; n = (uint32_t)(A.mate_offsets[r + 1] - o2); room = clamp_room(mates_end - o2); pos = 0;
            if (n > 0) {
              uint4 v = stream_refill(L, lane, seq, 0, n, room);
              sb = 1;
              cur = v.x; b1 = v.y; b2 = v.z; b3 = v.w;
            }
          } else {
            fin = true;
          }
        }
      }
      if (!W5) {  // end of a w-block: rebuild the suffix minima of the block just completed (wave-uniform control flow)
        if (tphase == w - 1) {
          uint64_t sm = ~0ULL;
          for (int j = w - 1; j >= 0; j--) { sm = lmin64(sm, win[j * 64 + lane]); win[j * 64 + lane] = sm; }  // in place
          tphase = 0;
        } else {
          tphase++;
        }
      }
      // ---- queue the emitted sequence spans ----
      const bool distinct = emit && (first || !(have_last && ekey == last_key));  // Supermers.spans :84-90
      last_key = emit ? ekey : last_key;
      have_last = have_last || emit;
      first = first && !emit;
      total += emit ? ekmers : 0;
      np += emit ? 1 : 0;
      nhits += emit ? 1 : 0;
      uint64_t E = __ballot(emit);
      if (E != 0) {
        if (emit) {
          int slot = (qhead + qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(E >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)E, 0))) & (QCAP - 1);
          L->q_key[slot] = ekey;
          if (HITS) L->q_ord[slot] = (uint16_t)ord0;
          L->q_meta[slot] = (uint32_t)lane | (distinct ? 64u : 0u) | ((uint32_t)ekmers << 7);
        }
        qn += __popcll(E);
        // (a loop, not an if: a batch can hand entries back -- bucket overflows re-queued with a larger displacement -- and
        // the next step may push 64 more; fewer than 64 must be left so that the 128-entry ring cannot overflow)
        while (qn >= 64) {
          lane_wave_sync();
          int back = 0;
          if (MODE == LANE_EMIT) emit_batch<HITS>(L, A, S, *Jp, qhead, 64, lane, row++, tile, ch_pos, ch_end, side_j, side_key, dbg);
          else if (!SLK_TUNE_ON(1)) back = probe_batch<HITS, LONG>(L, ocnt, A.T, qhead, qn, 64, lane, dbg, A.span_meta, A.span_taxon);
          qhead = (qhead + 64) & (QCAP - 1);
          qn += back - 64;
        }
      }
    }
    while (qn > 0) {  // drain, including entries re-queued by the batches themselves
      lane_wave_sync();
      const int cnt = min(qn, 64);
      int back = 0;
      if (MODE == LANE_EMIT) emit_batch<HITS>(L, A, S, *Jp, qhead, cnt, lane, row++, tile, ch_pos, ch_end, side_j, side_key, dbg);
      else if (!SLK_TUNE_ON(1)) back = probe_batch<HITS, LONG>(L, ocnt, A.T, qhead, qn, cnt, lane, dbg, A.span_meta, A.span_taxon);
      qhead = (qhead + cnt) & (QCAP - 1);
      qn += back - cnt;
    }
    lane_wave_sync();
    if (MODE == LANE_EMIT) {
      if (tile < etiles) {   // what the APPLY cannot recompute without scanning again
        if (have) S.read_info[r] = make_int2(total, nhits);
        if (lane == 0) S.tile_rows[tile] = make_uint2((uint32_t)row0, (uint32_t)(row - row0));
      }
      // a tile that sent off fewer batches than it owns of the LOOKUP job finishes its share now (a few at most: the shares are
      // dealt out by the batch's average)
      while (side_j < S.side_per_tile) {
        const uint64_t batch = tile * S.side_per_tile + side_j;
        const uint64_t k_now = side_key;
        side_j++;
        if (side_j < S.side_per_tile) side_key = side_load(S, batch + 1, lane);
        side_probe(L, A.T, S, lane, batch, k_now, dbg);
      }
    }

    // ---- per-read classification (one lane per read) ------------------------------------------------------------------
    if (MODE == LANE_LOCAL) {
      const uint32_t oflags = have ? L->o_flags[lane] : 0u;
      const bool dfr = have && (too_long || (oflags & 0x80000000u));  // re-done by the wave-per-read / segment kernels
      if (__ballot(dfr) != 0)   // (rare, and kept out of line: the hot loop's registers and schedule are not to know about it)
        hand_on(A.hand_hdr, A.hand_lists, A.hand_stride, A.hand_long_cap, A.long_max, A.long_bound[0], A.long_bound[1], A.long_bound[2],
                A.seg_min_len, A.wave_min, A.wave_ratio_q10, A.offsets, A.mate_offsets, dfr, too_long, r, lane, LONG);
      if (have && !dfr) resolve_lane<HITS, LONG>(L, ocnt, A, lane, r, oflags, total, nhits, np, dbg);
    } else {
      if (have && too_long) defer[r] = 1;   // EMIT job: fragments this kernel does not take (the caller routes them)
    }
    lane_wave_sync();
  }
  if (MODE == LANE_EMIT) {
    // the tails of the chunks this wave still holds are never written: zero keys (they travel and are answered; nobody reads the answers)
    for (uint32_t sh = 0; sh < (uint32_t)S.n_shards; sh++) {
      const uint32_t p0 = __builtin_amdgcn_readlane(ch_pos, sh), p1 = __builtin_amdgcn_readlane(ch_end, sh);
      for (uint32_t i = p0 + (uint32_t)lane; i < p1; i += 64) S.send_keys[(uint64_t)sh * S.cap + i] = 0;
    }
  }
}

template <bool W5, int MODE, bool HITS, bool LONG>
__global__ void LANE_BOUNDS lane_kernel(FusedArgs A, ShardIO S, int32_t *defer, uint32_t max_len, int dbg) {
  lane_body<W5, MODE, HITS, LONG>(A, S, nullptr, defer, max_len, dbg);
}
// the table-sharded step (engine.h: ShardIO): EMIT job of batch A + LOOKUP job (S.side_*) + APPLY job J of earlier batches
template <bool W5, bool HITS>
__global__ void LANE_BOUNDS lane_step_kernel(FusedArgs A, ShardIO S, ApplyJob J, int32_t *defer, uint32_t max_len, int dbg) {
  lane_body<W5, LANE_EMIT, HITS, false>(A, S, &J, defer, max_len, dbg);
}

static size_t lane_lds_per_wave(bool hits, bool w5, int w, bool lng) {
  return (hits ? sizeof(LaneLds) : offsetof(LaneLds, rb)) + (w5 ? 0 : (size_t)w * 64 * sizeof(uint64_t)) + (lng ? (size_t)OMAP * 64 * sizeof(uint32_t) : 0);
}

template <bool HITS, bool LONG>
static void launch_lane_mode(const FusedArgs &A, int32_t *defer, uint32_t max_len, hipStream_t s) {
  if (A.R == 0) return;
  const bool w5 = A.P.w == 5;
  const size_t per_wave = lane_lds_per_wave(HITS, w5, A.P.w, LONG);
  static const int extra_lds = getenv("SLK_LANE_EXTRA_LDS") ? atoi(getenv("SLK_LANE_EXTRA_LDS")) : 0;  // (occupancy experiment)
  size_t lds = per_wave * LW + (size_t)extra_lds;
  uint64_t tiles = (A.R + 63) / 64;
  uint64_t blocks = (tiles + LW - 1) / LW;
  static const int bpc = getenv("SLK_LANE_BLOCKS_PER_CU") ? atoi(getenv("SLK_LANE_BLOCKS_PER_CU")) : 0;  // (tuning experiment)
  if (bpc > 0 && blocks > (uint64_t)256 * bpc) blocks = (uint64_t)256 * bpc;
  if (LONG && blocks > 256 * 5) blocks = 256 * 5;   // (the number of hand-ons is only known on the device: the waves loop over them)
  dim3 g((unsigned)blocks), b(LW * 64);
#ifdef SLK_TUNING
  static const int dbg = getenv("SLK_DEBUG_ABLATE") ? atoi(getenv("SLK_DEBUG_ABLATE")) : 0;  // timing experiments only: 1 = no probes, 2 = no map updates
#else
  const int dbg = 0;
#endif
  static bool occ_printed = false;
  if (getenv("SLK_DEBUG_OCC") && !occ_printed) {  // (tuning aid)
    occ_printed = true;
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, w5 ? (const void *)lane_kernel<true, LANE_LOCAL, HITS, LONG> : (const void *)lane_kernel<false, LANE_LOCAL, HITS, LONG>, LW * 64, lds);
    fprintf(stderr, "[slk] lane kernel: %zu B LDS per block, %d blocks (%d waves) resident per CU\n", lds, nb, nb * LW);
  }
  if (w5) hipLaunchKernelGGL((lane_kernel<true, LANE_LOCAL, HITS, LONG>), g, b, lds, s, A, ShardIO{}, defer, max_len, dbg);
  else hipLaunchKernelGGL((lane_kernel<false, LANE_LOCAL, HITS, LONG>), g, b, lds, s, A, ShardIO{}, defer, max_len, dbg);
}

// A.span_taxon set: the hit lists are written too (span_meta / span_taxon / span_count, the layout of MODE_HITS)
void launch_lane(const FusedArgs &A, int32_t *defer, uint32_t max_len, hipStream_t s) {
  if (A.span_taxon) launch_lane_mode<true, false>(A, defer, max_len, s);
  else launch_lane_mode<false, false>(A, defer, max_len, s);
}
// the pass over the first one's hand-ons of 1 001 .. A.long_max bases (hand-on lists 0..3)
void launch_lane_long(const FusedArgs &A, uint32_t max_len, hipStream_t s) {
  if (A.span_taxon) launch_lane_mode<true, true>(A, nullptr, max_len, s);
  else launch_lane_mode<false, true>(A, nullptr, max_len, s);
}

// The table-sharded step: any of the three jobs may be absent (A.R == 0: no scan; S.side_n == 0: no lookups -- they need a scan to ride
// in; J.A.R == 0: no replay).  Hit lists are written when the batches carry span arrays (A.span_taxon / J.A.span_taxon: the
// flagged spans by the EMIT, the hits by the APPLY).  The grid is what the part holds at once: the waves are persistent, they keep
// the chunks of the send regions they are filling from tile to tile and draw their tiles from S.cursors[n_shards] (zero beforehand).
template <bool W5, bool HITS>
static void launch_step_mode(const FusedArgs &A, const ShardIO &S, const ApplyJob &J, int32_t *defer, uint32_t max_len, hipStream_t s) {
  const uint64_t tiles = std::max((A.R + 63) / 64, (J.A.R + 63) / 64);
  if (tiles == 0) return;
  const size_t lds = lane_lds_per_wave(HITS, W5, A.P.w, false) * LW;
  static int resident[2][2] = {{0, 0}, {0, 0}};   // blocks per CU by (W5, HITS) at the default window; other windows: asked every time
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  int nb = W5 ? resident[1][HITS] : 0;
  if (nb == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)lane_step_kernel<W5, HITS>, LW * 64, lds) != hipSuccess || nb <= 0) nb = 4;
    if (W5) resident[1][HITS] = nb;
  }
  static const int bpc = getenv("SLK_STEP_BLOCKS_PER_CU") ? atoi(getenv("SLK_STEP_BLOCKS_PER_CU")) : 0;  // (tuning experiment)
  if (bpc > 0) nb = bpc;
  const uint64_t blocks = std::min<uint64_t>((tiles + LW - 1) / LW, (uint64_t)cus * nb);
#ifdef SLK_TUNING
  static const int dbg = getenv("SLK_DEBUG_ABLATE") ? atoi(getenv("SLK_DEBUG_ABLATE")) : 0;  // timing experiments only (4, 32, 128: see side_probe / emit_batch)
#else
  const int dbg = 0;
#endif
  hipLaunchKernelGGL((lane_step_kernel<W5, HITS>), dim3((unsigned)blocks), dim3(LW * 64), lds, s, A, S, J, defer, max_len, dbg);
}
void launch_lane_step(const FusedArgs &A, const ShardIO &S, const ApplyJob &J, int32_t *defer, uint32_t max_len, hipStream_t s) {
  const bool hits = A.R ? A.span_taxon != nullptr : J.A.span_taxon != nullptr;
  if (A.P.w == 5) { if (hits) launch_step_mode<true, true>(A, S, J, defer, max_len, s); else launch_step_mode<true, false>(A, S, J, defer, max_len, s); }
  else { if (hits) launch_step_mode<false, true>(A, S, J, defer, max_len, s); else launch_step_mode<false, false>(A, S, J, defer, max_len, s); }
}

}  // namespace slk

// engine.h -- internal device/host structures of the classify engine (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slk {

// Splitter constants in the reference's left-aligned key space.
struct ScanParams {
  int32_t k, m, w;    // w = k - m + 1 = m-mers per k-mer window (PosRankWindow.scala:43-45,70)
  int32_t canonical;
  int32_t sh;         // (32 - m) * 2: left shift that aligns a right-aligned m-mer to the MSB
  uint64_t xmask;     // RandomXOR.mask word 0 (MinimizerPriorities.scala:146-160)
  uint64_t smask;     // SpacedSeed.spaceMask word 0 (:285-300)
  uint64_t keep;      // ~0 << sh: the 2m bits an m-mer occupies
};

// Minimizers wider than 32 nt (2..4 id columns, KeyValueIndex.scala:49; NTBitArray with several longs): wide.hip
constexpr int WIDE_MAXW = 4;
struct WideParams {
  int32_t k, m, w, canonical;
  int32_t W;                     // 64-bit words per m-mer = ceil(m / 32)
  int32_t last_sh;               // (32 - m % 32) % 32 * 2: unused low bits of the last word
  uint64_t xmask[WIDE_MAXW];     // RandomXOR.mask (MinimizerPriorities.scala:146-160)
  uint64_t smask[WIDE_MAXW];     // SpacedSeed.spaceMask (:285-300)
};
// Records with W-word keys: open addressing, linear probing by slot; a slot is taken when its taxon is non-zero.
struct WideTable {
  uint64_t *keys;   // [capacity][W]
  int32_t *taxa;    // [capacity]
  uint64_t mask;    // capacity - 1 (capacity is a power of two, load <= 0.5)
};
void launch_wide_insert(const WideTable &t, int W, const int64_t *keys, const int32_t *taxa, uint64_t n, unsigned long long *counters,
                        hipStream_t s);
void launch_wide_lookup(const WideTable &t, int W, const int64_t *keys, uint64_t n, int32_t *out, hipStream_t s);
void launch_wide_scan(const WideParams &P, const uint8_t *bases, const uint64_t *offsets, const uint8_t *mate_bases,
                      const uint64_t *mate_offsets, uint64_t R, uint64_t *span_keys, int32_t *span_meta, int32_t *span_count,
                      hipStream_t s);
void launch_wide_build_insert(const WideTable &t, int W, const int32_t *parents, int32_t ntax, const uint64_t *offsets, uint64_t R,
                              const uint64_t *span_keys, const int32_t *span_meta, const int32_t *span_count, const int32_t *chunk_taxon,
                              unsigned long long *counters, hipStream_t s);
void launch_wide_export(const WideTable &t, int W, int64_t *keys, int32_t *taxa, uint64_t capacity, unsigned long long *counter, hipStream_t s);
void launch_wide_gather_spans(int W, const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R, const uint64_t *span_keys,
                              const int32_t *span_meta, const uint64_t *out_offsets, void *out, int64_t *out_keys, hipStream_t s);
void launch_wide_probe(const WideTable &t, int W, const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R,
                       const uint64_t *span_keys, const int32_t *span_meta, const int32_t *span_count, int32_t *span_taxon,
                       hipStream_t s);

// HBM-resident record table: buckets of CELLS 8-byte cells, bucket-level linear probing.  SLK_BUCKET_CELLS = 8 (the default):
// 64-byte buckets, one HBM access granule, four lanes x 16 B per probe.  16: 128-byte buckets, one L2 line, eight lanes per probe
// -- a gather microbenchmark serves random requests of either size at the same rate, but the classify kernel with 128-byte
// buckets measured 3-4 % SLOWER at every load factor (twice the load instructions and bytes per probe for a tenth of the
// second-bucket probes; profiles/r03_bucket_geometry.txt), so that variant is a build option kept for the record.
// The number of buckets is ANY number (not a power of two: a 1.2e10-record library must not cost twice the memory of a
// 1.0e10-record one), chosen by a multiply-shift range reduction of the hash's top q bits:
//   h      = fmix64(key)                       (bijective, so (home bucket, remainder) identifies the key: lossless)
//   x      = h >> (64 - q),  q = ceil(log2(nbuckets))
//   home   = (x * nbuckets) >> q               (at most two consecutive x share a home bucket, since 2^(q-1) < nbuckets <= 2^q)
//   extra  = ((x * nbuckets) mod 2^q) >= nbuckets      (1 for the second of two x that share a home: tells them apart)
//   rem    = extra << (64 - q) | (h mod 2^(64 - q))
//   cell   = flag << 63 | ((rem << disp_bits | displacement) << taxon_bits) | taxon        (0 = empty; taxon != 0)
// A record lives in the first bucket home+d (d <= max_disp, wrapping at nbuckets) that had a free cell when it was inserted;
// cells are never freed, so a lookup may stop at the first bucket that still has an empty cell.  The top bit of a bucket's
// FIRST cell (where the cell layout leaves a bit: TableGeom.flag) says that some record found this bucket full and went on to the
// next: a lookup that finds its home bucket full but unflagged is a miss without a second probe.
#ifndef SLK_BUCKET_CELLS
#define SLK_BUCKET_CELLS 8
#endif
constexpr int CELLS = SLK_BUCKET_CELLS;               // 8-byte cells per bucket
constexpr int LPB = CELLS / 2;                        // lanes (16 bytes each) that read one bucket together
constexpr int BUCKET_SHIFT = CELLS == 16 ? 7 : 6;     // log2(bytes per bucket)
static_assert(CELLS == 8 || CELLS == 16, "buckets of 64 or 128 bytes");
struct TableGeom {
  uint64_t nbuckets;   // 32 .. 2^32
  uint64_t rem_mask;   // 2^(64 - q) - 1
  uint64_t flag;       // 1 << 63, or 0 when the cells have no bit to spare
  int32_t q;           // ceil(log2(nbuckets)), 5 .. 32
  int32_t taxon_bits;
  int32_t disp_bits;
  int32_t pad;
};
// home bucket of hash h and the remainder field of its cells, already shifted past the displacement field
__host__ __device__ __forceinline__ void table_slot(const TableGeom &g, uint64_t h, uint32_t &home, uint64_t &rem_hi) {
  // (q <= 32, so x is a 32-bit number; nbuckets is one too unless it is 2^32 itself: one 32 x 32 -> 64 multiply)
  const uint32_t x = (uint32_t)(h >> (64 - g.q));
  const uint64_t prod = (g.nbuckets >> 32) != 0 ? (uint64_t)x << 32 : (uint64_t)x * (uint32_t)g.nbuckets;
  home = (uint32_t)(prod >> g.q);
  const uint64_t extra = (prod & ((1ULL << g.q) - 1)) >= g.nbuckets ? 1 : 0;
  rem_hi = ((h & g.rem_mask) | (extra << (64 - g.q))) << g.disp_bits;
}
__host__ __device__ __forceinline__ uint32_t table_bucket(const TableGeom &g, uint32_t home, uint32_t d) {   // home + d, wrapping
  const uint64_t b = (uint64_t)home + d;
  return (uint32_t)(b >= g.nbuckets ? b - g.nbuckets : b);
}
// the (remainder, displacement) tag of a cell: what a probe compares
__host__ __device__ __forceinline__ uint64_t cell_tag(const TableGeom &g, uint64_t cell) { return (cell & ~g.flag) >> g.taxon_bits; }
// table_slot's inverse (export_kernel): the hash whose home bucket and remainder these are
__host__ __device__ inline uint64_t table_hash_of(const TableGeom &g, uint32_t home, uint64_t rem) {
  const uint64_t extra = rem >> (64 - g.q);
  const uint64_t num = (uint64_t)home << g.q;                          // first x with home(x) == home: ceil(home * 2^q / nbuckets)
  const uint64_t x = (num + g.nbuckets - 1) / g.nbuckets + extra;      // (home < 2^32, q <= 32; the sum cannot wrap: num <= 2^64 - 2^32)
  return (x << (64 - g.q)) | (rem & g.rem_mask);
}
struct TableView {
  const uint64_t *cells;
  TableGeom g;
  int32_t max_disp;   // largest displacement in use
  // Dense taxon ids (slk_index_finalize): when the caller's ids need more than 22 bits, the cells hold the rank of the taxon
  // among the taxonomy's nodes instead (the lane kernel packs taxon << 10 | count into one LDS word), the kernels walk a
  // parents array in those ranks, and ids are translated back wherever a taxon leaves the engine.  nullptr: ids as given.
  const int32_t *to_orig;
};
__device__ __forceinline__ int32_t ext_taxon(const TableView &T, int32_t t) { return (T.to_orig != nullptr && t > 0) ? T.to_orig[t] : t; }

__host__ __device__ inline uint64_t fmix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

__device__ __forceinline__ int32_t tax_parent(const int32_t *parents, int32_t ntax, int32_t t) {
  return ((uint32_t)t < (uint32_t)ntax) ? parents[t] : 0;
}

// LowestCommonAncestor.apply (LowestCommonAncestor.scala:49-78): the first node of b's path to the root that lies on a's
// path; ROOT when the paths never meet; NONE is the identity.  Computed by levelling the depths (O(depth) loads).
__device__ inline int32_t tax_lca(const int32_t *parents, int32_t ntax, int32_t a, int32_t b) {
  if (a == 0 || b == 0) return b == 0 ? a : b;
  if (a == b) return a;
  int da = 0, db = 0;
  for (int32_t x = a; x != 0; x = tax_parent(parents, ntax, x)) da++;
  for (int32_t y = b; y != 0; y = tax_parent(parents, ntax, y)) db++;
  for (; da > db; da--) a = tax_parent(parents, ntax, a);
  for (; db > da; db--) b = tax_parent(parents, ntax, b);
  while (a != b && a != 0) {
    a = tax_parent(parents, ntax, a);
    b = tax_parent(parents, ntax, b);
  }
  return a != 0 ? a : 1;
}

// span_meta packing: kmers (signed) << 4 | flag << 1 | distinct
__host__ __device__ inline int32_t pack_meta(int32_t kmers, int32_t flag, int32_t distinct) {
  return (int32_t)(((uint32_t)kmers << 4) | ((uint32_t)flag << 1) | (uint32_t)distinct);
}
__host__ __device__ inline int32_t meta_kmers(int32_t m) { return m >> 4; }
__host__ __device__ inline int32_t meta_flag(int32_t m) { return (m >> 1) & 7; }
__host__ __device__ inline int32_t meta_distinct(int32_t m) { return m & 1; }

// Where read r's span slots start in the per-batch scratch.  A fragment yields at most
// max(0, L1-k+1) [+ 1 + max(0, L2-k+1)] spans, so [offsets[r] (+ mate_offsets[r] + r)] regions never overlap.
__device__ inline uint64_t span_region(const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t r) {
  return offsets[r] + (mate_offsets ? mate_offsets[r] + r : 0);
}

// 16 bytes of a read stream at `src`.  The kernels consume sequences in 16-byte blocks; a block may reach past the end of its
// sequence (into the next one -- harmless, those bytes are never used), but it must not reach past the end of the caller's
// BUFFER: `room` = bytes between src and the end of the buffer.  Only the last block of a buffer takes the byte loads.
__device__ __forceinline__ uint4 load_block16(const uint8_t *src, uint32_t room) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (room >= 16u) {
    __builtin_memcpy(&v, src, 16);
  } else {
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t j = 0; j < 15; j++)
      if (j < room) w[j >> 2] |= (uint32_t)src[j] << (8 * (j & 3));
    v = make_uint4(w[0], w[1], w[2], w[3]);
  }
  return v;
}
__device__ __forceinline__ uint32_t clamp_room(uint64_t bytes) { return bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)bytes; }

// build-time view of the table (table_insert_kernel, build_kernel)
struct TableBuild {
  uint64_t *cells;
  TableGeom g;
  int32_t disp_limit;          // (1 << disp_bits) - 1
  // Table-sharded libraries (slk_index_set_shard): this index keeps the records with fmix64(key) mod n_shards == shard and drops
  // the others where they arrive, so every rank can be fed the same record stream / the same genomes.  n_shards <= 1: keeps all.
  uint32_t shard, n_shards;
  int32_t *max_disp;           // device: running maximum displacement
  unsigned long long *n_inserted, *n_duplicate, *n_overflow;
};
__host__ __device__ __forceinline__ bool shard_keeps(const TableBuild &t, uint64_t h) {
  return t.n_shards <= 1 || (uint32_t)(h % t.n_shards) == t.shard;
}

// Confidence thresholds travel BY VALUE in the kernel arguments: several classify calls may be queued on a stream, each with
// its own list, and none of them shares a staging buffer with another.
constexpr int MAX_THRESHOLDS = 16;
struct Thresholds { double v[MAX_THRESHOLDS]; };

// Layout of the hand-on lists and their header (FusedArgs.hand_*)
struct HandOn {
  enum { SEG = 4, REST = 5, WAVE0 = 6, WAVE_CLASSES = 8, LATE = WAVE0 + WAVE_CLASSES, SHORT = LATE + 1, LISTS = SHORT + 1 };
  // header words: [l] = entries of list l for l < 6, then
  enum { LONG_DRAW = 6, SEG_DRAW = 7, WAVE_DRAW = 8,  // tile / unit draws of the long, segment and wave pass
         HANDED = 9,                                  // fragments handed on by the first pass (or: routed to another list than 15)
         ORDERED = 10,                                // entries of the wave pass's list as launch_order_wave_list put it together
         N_LATE = 11, LATE_DRAW = 12, N_SHORT = 13, N_WAVE0 = 16, WORDS = 32 };
  static __host__ __device__ __forceinline__ int count_word(int l) {
    return l < WAVE0 ? l : l < LATE ? N_WAVE0 + (l - WAVE0) : l == LATE ? N_LATE : N_SHORT;
  }
  // where list l starts (entries from hand_lists); lists 0..5 and SHORT hold up to `stride` entries, the others `long_cap`
  static __host__ __device__ __forceinline__ uint64_t list_at(int l, uint64_t stride, uint64_t long_cap) {
    return l < WAVE0 ? (uint64_t)l * stride : l < LATE ? WAVE0 * stride + (uint64_t)(l - WAVE0) * long_cap
           : l == LATE ? (WAVE0 + 1) * stride + WAVE_CLASSES * long_cap : (WAVE0 + 1) * stride + (WAVE_CLASSES + 1) * long_cap;
  }
  static __host__ __device__ __forceinline__ uint64_t ordered_at(uint64_t stride, uint64_t long_cap) {   // (`stride` entries)
    return WAVE0 * stride + WAVE_CLASSES * long_cap;
  }
  static __host__ __device__ __forceinline__ uint64_t entries(uint64_t stride, uint64_t long_cap, bool with_short) {
    return (WAVE0 + 1 + (with_short ? 1 : 0)) * stride + (WAVE_CLASSES + 1) * long_cap;
  }
};


// arguments of the fused wave-per-read kernels (fused.hip)
struct FusedArgs {
  ScanParams P;
  TableView T;
  const int32_t *parents;
  int32_t ntax;
  // the same forest with an Euler tour, [ntax] x {parent, tin, tout, 0}: a is an ancestor-or-self of b iff tin[a] <= tin[b] <= tout[a].
  // The lane kernel's resolveTree asks that question of pairs of map taxa instead of walking root paths (capi.hip: build_tax_nodes)
  const uint4 *nodes;
  const uint8_t *bases;
  const uint64_t *offsets;
  const uint8_t *mate_bases;
  const uint64_t *mate_offsets;
  uint64_t R;
  uint64_t out_stride;   // out_taxon / out_classified are [C][out_stride] (= R unless this launch is a sub-batch of a larger call)
  int32_t min_hit_groups;
  Thresholds thr;
  int32_t C;
  int32_t *out_taxon;
  uint8_t *out_classified;
  int32_t *out_nd, *out_tk, *out_nh, *out_np;
  uint64_t *span_keys;   // MODE_SPANS output (sparse per-read regions, see span_region)
  int32_t *span_meta;    // MODE_SPANS / MODE_HITS output
  int32_t *span_taxon;   // MODE_HITS output
  int32_t *span_count;   // MODE_SPANS / MODE_HITS output
  int32_t *status;       // device error bits: 1 = taxon map overflow
  // Hand-on lists (uint32 fragment indices; HandOn below is their layout).  The lane kernel's first pass sorts the fragments it does
  // not take by the kernel that will: lists 0..3 = the four length classes of its own long variant (tiles of similar length: the 64
  // lanes of a wave run in lockstep), 4 = the lane-per-segment kernel, 5 = the wave kernel's short ones (map overflows), 6..13 = the
  // wave kernel's long fragments in eight length classes, longest first: that kernel takes a fragment per wave, so a long fragment
  // started late is what the whole pass then waits for, and it walks 6, 7, .., 13, 5 -- longest first, whatever the order of the
  // batch.  (A small kernel strings the lists together before the pass, launch_order_wave_list: the wave kernel itself, tuned to
  // its registers, walks one list as before -- drawing from four inside it cost it a tenth of its rate.)  14 = what the long
  // variant hands on in turn (map overflows; a second launch of the wave kernel takes them).
  // A batch of mostly long fragments has no first pass: its few short fragments, spread over all the tiles, would keep every tile
  // going for up to 1000 steps with a tenth of its lanes (1.1 of the 10.3 ms of a nanopore-like batch).  A routing kernel fills the
  // lists instead (launch_route), the fragments of at most 1000 bases go to list 15, and the long variant takes them as a fifth
  // class, the last to start.
  unsigned long long *hand_hdr;          // HandOn::WORDS words
  uint32_t *hand_lists;
  uint64_t hand_stride;                  // entries of a list that may hold every fragment (= R)
  uint64_t hand_long_cap;                // entries of a list of fragments over 1000 bases (a batch holds at most total bases / 1001)
  uint32_t route_first;                  // 1: the routing kernel stands for the first pass (list 15 exists)
  uint32_t wave_min, wave_ratio_q10;     // wave classes: 13 = up to wave_min * ratio, 12 = up to wave_min * ratio^2, ...
  uint32_t long_max;                     // the long variant takes fragments of up to this many bases (0: there is no such pass)
  uint32_t long_bound[3];                // borders of its length classes
  uint32_t seg_min_len;                  // unpaired fragments of at least this many bases belong to the lane-per-segment kernel
                                         // (launch_segments), 0 = none
  // what a pass kernel (long lane, segment, wave) walks: work_list[0 .. *work_count), drawing units from *work_draw if set
  const uint32_t *work_list;
  const unsigned long long *work_count;
  unsigned long long *work_draw;
};

// APPLY job of the table-sharded step kernel: the lists and the log an EARLIER batch's EMIT left on this rank, the owners' answers,
// and where that batch's results go (A: R, outputs, thresholds, taxonomy; offsets only when hit lists are written).  A.R == 0: no job.
struct ApplyJob {
  FusedArgs A;
  int32_t n_shards;
  uint64_t cap;
  const uint32_t *send_meta;
  const uint4 *batch_log;
  const uint2 *tile_rows;
  const int2 *read_info;
  const int32_t *taxa;               // [n_shards][cap]: the owners' answers at the positions of the keys (the caller's ids)
  const int32_t *to_dense;           // caller's id -> the table's dense id (nullptr: ids as given)
  int32_t n_to_dense;
  int32_t *defer;                    // [R] in: 1 = the EMIT did not take the fragment (too long); out: also map overflows
  unsigned long long *n_deferred;    // += the fragments flagged in defer[] (one atomic per wave that has any)
};

// Table-sharded classification (BASELINE configs[3]).  A batch takes three jobs, each of them riding in the ONE kernel
// launched per pipeline step (lane_step_kernel, lane.hip) beside the jobs of its neighbours:
//   EMIT    (step t)      scans the fragments and, instead of probing, appends every minimizer to the send region of the rank that
//                         owns it (fmix64(key) mod n_shards), logging per probe batch where each owner's group of keys went;
//   LOOKUP  (step t + 2)  the OWNER's side: the keys this rank received are answered with the local kernel's cooperative probe,
//                         64 of them whenever a wave has sent off 64 keys of its own;
//   APPLY   (step t + 4)  no second scan: the tile's probe batches are replayed from the log, each probe's taxon is read from the
//                         owners' answers at the logged position, folded into the fragment's map and resolved as the local kernel does.
// Send regions: ONE contiguous region per owner, [n_shards][cap] -- what the exchange sends is region[0 .. cursor) as it stands, no
// compaction pass.  A wave reserves `chunk` entries of an owner's region at a time (one atomic per chunk: a cursor per owner bumped
// once per probe batch serialised the appends, 73 ms per 10 M reads) and keeps the chunk it is filling from tile to tile (the
// kernel's waves are persistent and draw tiles from a counter), so the only entries never written are the tails of the chunks the
// waves hold when the kernel ends (zeroed there: they travel and are answered like keys, nobody reads their answers).
struct ShardIO {
  int32_t n_shards;
  uint32_t chunk;                    // entries a wave reserves at a time (a power of two >= 64)
  uint64_t cap;                      // entries per owner region (a multiple of chunk, < 2^32)
  int64_t *send_keys;                // [n_shards][cap]
  unsigned long long *cursors;       // [n_shards + 1]: entries reserved per owner (multiples of chunk; beyond cap: status bit 2);
                                     // [n_shards]: tile draw of the launch
  uint32_t *send_meta;               // [n_shards][cap]: the keys' span metadata (owner lane | distinct | k-mers | ordinal); stays on this rank
  uint4 *batch_log;                  // [rows][n_shards]: {position of the owner's group, start of a freshly reserved chunk, keys, room left
                                     // at the position}: key i of the group sits at position + i while i < room, else at fresh + i - room
  uint2 *tile_rows;                  // [tiles]: {first row of the tile in the log, rows it used}
  int2 *read_info;                   // [R]: total k-mers, spans of a fragment (TaxonCounts.totalKmers; the "no span, no row" test)
  // LOOKUP job (optional: side_n != 0).  The 64-key batches of side_keys[0 .. side_n) are dealt out statically -- tile t of the scan
  // owns batches [t * side_per_tile, (t + 1) * side_per_tile) (a shared cursor would be one atomic address for six million draws:
  // measured, 73 ms) -- and a tile that sends off fewer batches than it owns finishes its share at its end.
  const int64_t *side_keys;
  uint64_t side_n;
  uint32_t side_per_tile;
  int32_t *side_out;
};
enum { LANE_LOCAL = 0, LANE_EMIT = 1 };
void launch_lane_step(const FusedArgs &A, const ShardIO &S, const ApplyJob &J, int32_t *defer, uint32_t max_len, hipStream_t s);
// cooperative point lookups (4 lanes x 16 B per bucket) and the scatter of returned taxa to their slots (shard.hip)
void launch_lookup_coop(const TableView &t, const int64_t *keys, uint64_t n, int32_t *out, hipStream_t s);
// cells' taxon field -> to_dense[taxon]; *undefined counts the cells whose taxon has no dense id; apply = false: count only
void launch_remap_cells(uint64_t *cells, uint64_t ncells, int32_t taxon_bits, const int32_t *to_dense, int32_t n_to_dense,
                        unsigned long long *undefined, bool apply, hipStream_t s);

enum { MODE_SPANS = 0, MODE_CLASSIFY = 1, MODE_HITS = 2 };
void launch_fused(int mode, const FusedArgs &A, hipStream_t s);
// strings the wave pass's hand-on lists together, long fragments first (FusedArgs.hand_hdr); behind the passes that fill them
void launch_order_wave_list(const FusedArgs &A, hipStream_t s);
// list[0 .. *count) = the indices r < R with flags[r] != 0 (in no particular order); *count must be zero beforehand
void launch_segments(const FusedArgs &A, hipStream_t s);
// lane-per-fragment classify kernel (lane.hip); fragments it cannot take are flagged in defer[] for launch_fused
void launch_lane(const FusedArgs &A, int32_t *defer, uint32_t max_len, hipStream_t s);
void launch_lane_long(const FusedArgs &A, uint32_t max_len, hipStream_t s);
// instead of a first pass (FusedArgs.route_first): every fragment goes to the hand-on list of the kernel that takes it
void launch_route(const FusedArgs &A, hipStream_t s);

// launchers (kernels.hip)
void launch_table_insert(const TableBuild &t, const int64_t *keys, const int32_t *taxa, uint64_t n, hipStream_t s);
// build.hip: minimizers of taxon-labelled sequence chunks merged by LCA straight into the table; table -> records
constexpr uint32_t BUILD_CHUNK_WINDOWS = 512;
constexpr int BUILD_MAX_W = 28;
void launch_build(const ScanParams &P, const TableBuild &T, const int32_t *parents, int32_t ntax, const uint8_t *bases,
                  uint64_t total_bases, const uint64_t *chunk_start, const uint32_t *chunk_len, const int32_t *chunk_taxon, uint64_t nchunks,
                  hipStream_t s);
void launch_export(const TableView &T, uint64_t nbuckets, int64_t *keys, int32_t *taxa, uint64_t capacity,
                   unsigned long long *counter, hipStream_t s);
void launch_export_range(const TableView &T, uint64_t bucket0, uint64_t bucket1, int64_t *keys, int32_t *taxa, uint64_t capacity,
                         unsigned long long *counter, hipStream_t s);
void launch_table_lookup(const TableView &t, const int64_t *keys, uint64_t n, int32_t *out, hipStream_t s);
void launch_scan(const ScanParams &P, const uint8_t *bases, const uint64_t *offsets, const uint8_t *mate_bases,
                 const uint64_t *mate_offsets, uint64_t R, uint64_t *span_keys, int32_t *span_meta, int32_t *span_count,
                 hipStream_t s);
void launch_probe(const TableView &T, const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R,
                  const uint64_t *span_keys, const int32_t *span_meta, const int32_t *span_count, int32_t *span_taxon,
                  hipStream_t s);
void launch_classify(const int32_t *parents, const uint4 *nodes, int32_t T, const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R,
                     const int32_t *span_meta, const int32_t *span_taxon, const int32_t *span_count,
                     uint64_t *map_scratch, int32_t min_hit_groups, const Thresholds &thr, int32_t C, uint64_t out_stride,
                     int32_t *out_taxon, uint8_t *out_classified, int32_t *out_num_distinct, int32_t *out_total_kmers,
                     int32_t *out_num_hits, int32_t *out_num_probes, hipStream_t s);
void launch_gather_spans(const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R, const uint64_t *span_keys,
                         const int32_t *span_meta, const uint64_t *out_offsets, void *out, hipStream_t s);
// out[i] = counts[0] + .. + counts[i - 1] for i = 0 .. n (n + 1 values); tmp holds n / 2048 + 2 words
void launch_counts_to_offsets(const int32_t *counts, uint64_t n, uint64_t *out, uint64_t *tmp, hipStream_t s);
// the hit lists merged as TaxonCounts.fromHits merges them: count = true -> merged[r] = entries of fragment r; else -> out[moffs[r] ..)
void launch_merged_hits(bool count, const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R, const int32_t *span_meta, const int32_t *span_taxon,
                        const int32_t *nh, const uint64_t *moffs, int32_t *merged, void *out, hipStream_t s);
void launch_gather_hits(const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R, const int32_t *span_meta,
                        const int32_t *span_taxon, const uint64_t *out_offsets, void *out, hipStream_t s);
// words [w0, w1) of a packed read stream (host/pack.hpp) -> out[16 w0 .. 16 w1): "ACGT" by code, 'N' where the validity bit is clear
void launch_unpack_bases(const uint32_t *codes, const uint16_t *valid, uint64_t w0, uint64_t w1, uint8_t *out, hipStream_t s);

}  // namespace slk

/*
 * slacken_amd.h -- C ABI of the MI355X-native classify engine (libslacken_amd.so).
 *
 * Drop-in boundary for the hot path of JNP-Solutions/Slacken:
 *   minimizer scan -> minimizer->taxon lookup -> per-read LCA ("resolveTree") classification.
 *
 * The reference has NO native/FFI interface (it is Scala on Spark).  The seam this ABI sits
 * behind is therefore method-level; each entry point cites the reference method(s) it replaces, relative to
 * /root/reference/src/main/scala/com/jnpersson/ (abbreviated S/).  INTEGRATION.md shows the JNI binding a
 * maintainer would add on the Scala side.
 *
 * Conventions
 *   - every function returns int32 status: 0 = ok, negative = error (SLK_E_*); slk_last_error() gives the text
 *     (thread-local);
 *   - no exceptions, no callbacks, no ownership transfer: the caller owns every buffer it passes; the library owns
 *     only the opaque handles it created;
 *   - plain pointers and sizes only.  "host" entry points take host pointers and do their own H2D/D2H; "_device"
 *     entry points take pointers to memory already resident on the index's GPU and are asynchronous on the
 *     slk_stream (call slk_stream_synchronize before reading results);
 *   - a buffer of bases is exactly offsets[R] (mate_offsets[R]) bytes: nothing past it is read, no padding is needed
 *     (the kernels stream 16-byte blocks and assemble a buffer's last block from byte loads);
 *   - an slk_index is immutable after slk_index_finalize() and may be shared by many threads; an slk_stream holds
 *     one HIP stream plus the scratch of ONE in-flight batch: use one per calling thread.
 *   - minimizers of up to 128 nt (id_longs = ceil(m/32) <= 4 key words per record, row-major) are supported by
 *     slk_index_create / append / lookup / add_sequences / export, slk_spans_batch_wide and the classify entry points; the staged
 *     device entries and the sharded ones take one-word keys (m <= 32) and return SLK_E_UNSUPPORTED otherwise.
 */
#ifndef SLACKEN_AMD_H
#define SLACKEN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLK_OK 0
#define SLK_E_INVALID (-1)     /* bad argument */
#define SLK_E_UNSUPPORTED (-2) /* e.g. m > 32 */
#define SLK_E_HIP (-3)         /* HIP runtime error (text in slk_last_error) */
#define SLK_E_NO_GPU (-4)      /* no usable gfx950 device: there is NO CPU fallback */
#define SLK_E_CAPACITY (-5)    /* output buffer or table capacity exceeded */
#define SLK_E_STATE (-6)       /* call order violated (e.g. classify before finalize / taxonomy) */

/* S/slacken/package.scala:30-39 and S/slacken/Taxonomy.scala:30-31 */
#define SLK_TAXON_NONE 0
#define SLK_TAXON_ROOT 1
#define SLK_TAXON_AMBIGUOUS (-1)
#define SLK_TAXON_MATE_PAIR_BORDER (-2)
#define SLK_FLAG_SEQUENCE 1
#define SLK_FLAG_AMBIGUOUS 2
#define SLK_FLAG_MATE_PAIR_BORDER 3
/* S/kmers/minimizer/package.scala:32 */
#define SLK_DEFAULT_TOGGLE_MASK 0xe37e28c4271b5a2dULL

typedef struct slk_index slk_index;
typedef struct slk_stream slk_stream;

/* The splitter parameters of an index: S/kmers/IndexParams.scala:30-47 (k, m), S/kmers/SplitterFormat.scala:42-64
 * (randomXOR mask, canonical, minimizerSpaces).  Defaults of `slacken build`: k=35, m=31, spaces=7,
 * xor_mask=SLK_DEFAULT_TOGGLE_MASK, canonical=1 (S/slacken/Slacken.scala:126-136). */
typedef struct {
  int32_t k;
  int32_t m;
  int32_t spaces;
  int32_t canonical;
  uint64_t xor_mask;
  int32_t id_longs; /* ceil(m/32) <= 4, S/slacken/KeyValueIndex.scala:49: key words per record */
  int32_t reserved;
} slk_params;

/* Sizing of the HBM-resident record table (the engine's replacement for the bucketed Parquet table scanned by the
 * join at S/slacken/Classifier.scala:84). */
typedef struct {
  uint64_t expected_records; /* upper bound on records that will be appended */
  int32_t max_taxon;         /* largest taxon id that will be appended (0 = derive from nothing: 2^22-1) */
  float load_factor;         /* target cells-used fraction, 0 = default: 0.55 while the table then takes at most 55 % of the device's
                                FREE memory, more for larger libraries (0.85 at most); less for tables whose cells leave a short
                                displacement field */
} slk_table_config;

typedef struct {
  uint64_t records;       /* records stored (taxon != NONE) */
  uint64_t buckets;       /* buckets of bucket_cells 8-byte cells (any number: not a power of two) */
  uint64_t table_bytes;
  int32_t bucket_bits, taxon_bits, disp_bits; /* bucket_bits = ceil(log2(buckets)) */
  int32_t max_displacement; /* largest bucket displacement in use (0 = every record in its home bucket) */
  uint64_t duplicate_keys;  /* appended records whose key was already present (contract violation; first kept) */
  int32_t taxonomy_size;
  int32_t device;
  int32_t dense_taxa;       /* > 0: taxon ids beyond 22 bits were renumbered internally at slk_index_finalize (the number of
                               taxonomy nodes); every taxon that crosses this ABI is still the caller's id */
  int32_t bucket_cells;     /* 8-byte cells per bucket: 8 (64-byte buckets; a build option makes it 16) */
  float load_factor;        /* the load the table was sized for: slk_table_config.load_factor, or what the default chose from the
                               device's free memory (runs on parts with different memory are comparable by this) */
  int32_t grown;            /* times the table moved to a larger geometry because a record found no cell within reach
                               (more records than expected_records: the load never fails for that) */
} slk_index_info;

/* OrdinalSpan (S/slacken/package.scala:61-62) without the title; ordinal = position in the read's span list.
 * For flag != SLK_FLAG_SEQUENCE the reference draws a random minimizer (S/slacken/Supermers.scala:34-42,53-56)
 * that can never be observed; this engine writes key = 0. */
typedef struct {
  int64_t key; /* left-aligned, exactly the value of Parquet column id1 */
  int32_t kmers;
  int8_t flag;
  uint8_t distinct;
  uint16_t pad;
} slk_span;

/* TaxonHit (S/slacken/KeyValueIndex.scala:436-441); distinct/ordinal are implied by position. */
typedef struct {
  int32_t taxon;
  int32_t count;
} slk_hit;

int32_t slk_device_count(void);
const char *slk_last_error(void);
const char *slk_version(void);

/* Pinned host memory.  The host entry points accept any host memory; buffers that come from slk_host_alloc (or were
 * registered with slk_host_register: long-lived caller buffers, e.g. a JVM direct ByteBuffer) are DMA'd from and to
 * directly, at PCIe rate, instead of being copied through the stream's staging buffers.  A pointer anywhere inside such a
 * buffer qualifies.  slk_host_free takes the pointer slk_host_alloc returned / slk_host_register was given. */
int32_t slk_host_alloc(size_t bytes, void **out);
int32_t slk_host_register(void *ptr, size_t bytes);
int32_t slk_host_free(void *ptr);

/* ---- index: replaces KeyValueIndex.load / loadRecords (S/slacken/KeyValueIndex.scala:150-159,413-426) ---- */
int32_t slk_index_create(const slk_params *params, const slk_table_config *cfg, int32_t device, slk_index **out);
/* Append records (id1: int64 left-aligned minimizer, taxon: int32) -- the rows of the Parquet table, in any order
 * and any chunking (one call per bucket file is the intended use).  Keys are unique (guaranteed by makeRecords'
 * groupBy, KeyValueIndex.scala:85-93).  Records with taxon == NONE are skipped (indistinguishable from a miss). */
int32_t slk_index_append(slk_index *ix, const int64_t *keys, const int32_t *taxa, uint64_t n);
int32_t slk_index_append_device(slk_index *ix, const int64_t *d_keys, const int32_t *d_taxa, uint64_t n);
/* Table-sharded libraries (a table beyond one GPU's memory; BASELINE.json configs[3], the exchange that replaces the shuffle of
 * the join at S/slacken/Classifier.scala:84): this index keeps only the records whose key falls to `shard` of `n_shards`
 * (slk_shard_of) and drops the others where they arrive, in slk_index_append[_device] and slk_index_add_sequences[_device] alike --
 * every rank / device is handed the same record stream or the same genomes and ends up with its share.  Call before the first
 * record; slk_table_config.expected_records then bounds this shard's records. */
int32_t slk_index_set_shard(slk_index *ix, uint32_t shard, uint32_t n_shards);
/* parents[t] = parent taxon, parents[ROOT] = NONE, unused ids = NONE: Taxonomy.parents (S/slacken/Taxonomy.scala:81-109,159);
 * replaces the bcTaxonomy broadcast (KeyValueIndex.scala:44-47). */
int32_t slk_index_set_taxonomy(slk_index *ix, const int32_t *parents, int32_t T);
/* Library construction from taxon-labelled sequences (BASELINE config 5, the dynamic 2-step library): every super-mer's
 * minimizer of every sequence, labelled with the sequence's taxon, merged per minimizer by LCA -- SplitterMinimizers.find
 * (S/slacken/Minimizers.scala:43-76) + groupBy(id).agg(TaxonLCA) (KeyValueIndex.makeRecords, KeyValueIndex.scala:85-93;
 * LowestCommonAncestor.scala:152-170).  Sequences are split around anything that is not ACGTU (either case), as
 * InputReader.removeInvalid does for library input (S/kmers/input/InputReader.scala:60-72); they must be free of
 * whitespace.  Sequence i is bases[offsets[i] .. offsets[i+1]) with taxon taxa[i]; sequences with taxon NONE are
 * skipped (the reference filters on Taxonomy.isDefined, KeyValueIndex.scala:118-120 -- the caller applies that filter).
 * Needs the taxonomy (slk_index_set_taxonomy) and an unfinalized index whose slk_table_config.expected_records bounds
 * the number of distinct minimizers.  May be called any number of times, also mixed with slk_index_append of records
 * whose keys do not occur in the sequences; the result does not depend on the order or batching of the calls. */
int32_t slk_index_add_sequences(slk_index *ix, const uint8_t *bases, const uint64_t *offsets, const int32_t *taxa,
                                uint64_t n_sequences);
/* The same with the bases already resident on the index's GPU (d_bases: device pointer to offsets[n_sequences] bytes;
 * offsets and taxa: host arrays).  The sequences are scanned where they lie. */
int32_t slk_index_add_sequences_device(slk_index *ix, const uint8_t *d_bases, const uint64_t *offsets, const int32_t *taxa,
                                       uint64_t n_sequences);
/* Ends the build.  If the taxon ids need more than 22 bits (slk_table_config.max_taxon >= 2^22), the taxonomy has been set
 * and every record's taxon is one of its nodes, the table is renumbered here to dense internal ids (one pass over the
 * cells) so that the fast kernels apply; ids crossing the ABI are unaffected.  Set the taxonomy BEFORE finalizing to get
 * this; it cannot be replaced afterwards on such an index. */
int32_t slk_index_finalize(slk_index *ix);
/* The table's records as (key, taxon) arrays -- what KeyValueIndex.writeRecords would persist (KeyValueIndex.scala:125-139);
 * keys: id_longs words per record, row-major.  The order is unspecified (a set).  *n_records receives the number of records; if it exceeds capacity only the first
 * `capacity` were written and SLK_E_CAPACITY is returned.  keys/taxa may be NULL with capacity 0 to query the count. */
int32_t slk_index_export(const slk_index *ix, int64_t *keys, int32_t *taxa, uint64_t capacity, uint64_t *n_records);
int32_t slk_index_get_info(const slk_index *ix, slk_index_info *out);
/* The table's bucket choice as host arithmetic (no GPU): the range reduction of a 64-bit hash onto ANY number of buckets
 * (32 <= nbuckets <= 2^32) -- home = (top q bits of hash) * nbuckets >> q, q = ceil(log2(nbuckets)) -- with the remainder a cell
 * keeps so that (home, rem) still identifies the hash (the table is lossless, unlike a compact hash table with truncated keys),
 * and its inverse.  For tests and for tools that lay out tables offline. */
int32_t slk_table_slot(uint64_t nbuckets, uint64_t hash, uint32_t *home, uint64_t *rem);
int32_t slk_table_hash_of(uint64_t nbuckets, uint32_t home, uint64_t rem, uint64_t *hash);
/* Point lookups (host arrays), for tests and tooling: taxon or NONE per key -- the left join + spanToHit's
 * otherwise(NONE) (KeyValueIndex.scala:176-185). */
int32_t slk_index_lookup(const slk_index *ix, const int64_t *keys, uint64_t n, int32_t *out_taxa);
void slk_index_destroy(slk_index *ix);

int32_t slk_stream_create(slk_index *ix, slk_stream **out);
int32_t slk_stream_synchronize(slk_stream *st);
/* on != 0: the hit lists slk_classify_batch / slk_classify_batch_packed return on this stream are merged as TaxonCounts.fromHits
 * merges them (S/slacken/TaxonCounts.scala:31-48) -- adjacent TaxonHits of one taxon (AMBIGUOUS included; a MATE_PAIR_BORDER stands
 * alone) become one entry whose count is the sum of theirs -- and out_hit_offsets index the merged entries.  It is the list
 * ClassifiedRead.outputLine prints (pairsInOrderString :94-110, lengthString) at a fifth to a tenth of the bytes over the link;
 * ordinals and `distinct` of single spans are no longer implied by position, so lists that will be regrouped by title
 * (slk_classify_hits) or counted per minimizer must stay un-merged (the default).  A call that asks for the offsets alone
 * (out_hits == NULL) still counts the single spans. */
int32_t slk_stream_set_merged_hits(slk_stream *st, int32_t on);
void *slk_stream_hip_stream(slk_stream *st); /* the hipStream_t, for event timing by the caller */
void slk_stream_destroy(slk_stream *st);

/* ---- kernel-1-only entry: replaces KeyValueIndex.getSpans (S/slacken/KeyValueIndex.scala:163-173), i.e.
 * Supermers.splitFragment + Supermers.spans (S/slacken/Supermers.scala:49-97,113-125) over
 * MinSplitter.splitEncode (S/kmers/minimizer/MinSplitter.scala:98-101).
 * bases: concatenated ASCII reads WITHOUT whitespace (getSpans' precondition, KeyValueIndex.scala:162);
 * offsets[R+1]; mate_bases/mate_offsets: second mates with the same indexing, or NULL for single-end.
 * out_span_offsets[R+1]; out_spans[spans_capacity] in read order then ordinal order. */
int32_t slk_spans_batch(slk_index *ix, slk_stream *st, const uint8_t *bases, const uint64_t *offsets,
                        const uint8_t *mate_bases, const uint64_t *mate_offsets, uint64_t R,
                        uint64_t *out_span_offsets, slk_span *out_spans, uint64_t spans_capacity);

/* The same for any number of id columns (KeyValueIndex.scala:49: idLongs = ceil(m / 32)): out_keys[spans * id_longs] receives
 * OrdinalSpan.minimizer row by row (id1..idN, left-aligned words), out_spans[i].key = id1.  With one id column it equals
 * slk_spans_batch plus the copy of the keys. */
int32_t slk_spans_batch_wide(slk_index *ix, slk_stream *st, const uint8_t *bases, const uint64_t *offsets,
                             const uint8_t *mate_bases, const uint64_t *mate_offsets, uint64_t R,
                             uint64_t *out_span_offsets, slk_span *out_spans, int64_t *out_keys, uint64_t spans_capacity);

/* ---- the hot path: replaces Classifier.classify (S/slacken/Classifier.scala:114-121) =
 * collectHitsBySequence (:70-96: getSpans -> join -> spanToHit -> group) + classifyHits (:124-147) ->
 * Classifier.classify (object, :439-454) -> TaxonCounts (S/slacken/TaxonCounts.scala:31-87) ->
 * LowestCommonAncestor.resolveTree (S/slacken/LowestCommonAncestor.scala:91-146).
 * One call = one batch of R fragments and C confidence thresholds.
 *   out_taxon[C*R], out_classified[C*R]  (threshold-major): ClassifiedRead.taxon / .classified
 *   out_num_distinct[R]  hits with distinct && taxon != NONE        (Classifier.scala:94)
 *   out_total_kmers[R]   TaxonCounts.totalKmers                      (TaxonCounts.scala:84-87)
 *   out_hit_offsets[R+1], out_hits[hits_capacity] (both nullable): un-merged TaxonHits in ordinal order incl.
 *     the -1 / -2 entries.  A read with out_hit_offsets[r+1] == out_hit_offsets[r] produced no span: the reference
 *     emits NO row for it (grouping is over span rows, Classifier.scala:92) -- the host must drop it.
 *     out_hit_offsets without out_hits: the number of spans per read only (offsets[r+1] - offsets[r]); the lists are then
 *     neither built nor copied, and the call takes the kernels that do not keep span order (the fastest route).
 * The caller's buffers may be any host memory: the copies go through pinned staging buffers of the stream.
 * Sample-id regex, titles, duplicate-title merging and text formatting stay on the host. */
int32_t slk_classify_batch(slk_index *ix, slk_stream *st, const uint8_t *bases, const uint64_t *offsets,
                           const uint8_t *mate_bases, const uint64_t *mate_offsets, uint64_t R,
                           int32_t min_hit_groups, const double *thresholds, int32_t C, int32_t *out_taxon,
                           uint8_t *out_classified, int32_t *out_num_distinct, int32_t *out_total_kmers,
                           uint64_t *out_hit_offsets, slk_hit *out_hits, uint64_t hits_capacity);

/* The same call with the reads in the engine's own 3-bit form -- InputFragment.nucleotides (S/kmers/minimizer/MinSplitter.scala:31-32)
 * already encoded as BitRepresentation.charToTwobit encodes them (S/kmers/util/BitRepresentation.scala:127-135: A = 0, C = 1, G = 2,
 * T / U = 3, either case) with the isValid test (:140-143) as one bit per base -- so that a batch costs 6 bytes per 16 bases on the
 * PCIe link instead of 16: the call a JNI shim makes is bound by that link (slk_classify_batch from pinned memory: 305 M reads/s of
 * 150 bp against the kernels' 1 100 M).  Base p of the concatenated reads (the p that offsets[] counts) is described by bits
 * 2 (p % 16) .. 2 (p % 16) + 1 of codes[p / 16] and bit p % 16 of valid[p / 16]; codes of invalid bases are ignored; both arrays
 * hold ceil(offsets[R] / 16) words (mates likewise).  Results are those of slk_classify_batch on any text with the same codes and
 * validity.  slk_pack_bases makes the form from text (AVX2 + BMI2 where the CPU has them, several threads for large inputs). */
int32_t slk_classify_batch_packed(slk_index *ix, slk_stream *st, const uint32_t *codes, const uint16_t *valid, const uint64_t *offsets,
                                  const uint32_t *mate_codes, const uint16_t *mate_valid, const uint64_t *mate_offsets, uint64_t R,
                                  int32_t min_hit_groups, const double *thresholds, int32_t C, int32_t *out_taxon,
                                  uint8_t *out_classified, int32_t *out_num_distinct, int32_t *out_total_kmers,
                                  uint64_t *out_hit_offsets, slk_hit *out_hits, uint64_t hits_capacity);
int32_t slk_pack_bases(const uint8_t *bases, uint64_t n, uint32_t *codes, uint16_t *valid);

/* ---- kernel-3-only entry: Classifier.classify (object, S/slacken/Classifier.scala:439-454) on hit lists the caller holds
 * (host pointers; synchronous): TaxonCounts.toMap/totalKmers + resolveTree + the minHitGroups test for R lists of un-merged
 * hits in ordinal order.  This is what the host uses to reproduce the reference's regrouping by TITLE
 * (groupBy("seqTitle") + collect_list, Classifier.scala:92; sorted by ordinal :136): fragments that share a title are one
 * read there, so the host concatenates their hit lists, sorts them stably by ordinal and classifies the merged list here.
 *   hits[hit_offsets[r] .. hit_offsets[r+1])   list r; taxon AMBIGUOUS / MATE_PAIR_BORDER entries as slk_classify_batch
 *                                              returns them
 *   distinct[i] (nullable)                     OrdinalSpan.distinct of hit i (slk_span.distinct of the same ordinal);
 *                                              NULL = no hit counts as distinct
 *   out_num_distinct[R], out_total_kmers[R]    nullable */
int32_t slk_classify_hits(slk_index *ix, slk_stream *st, uint64_t R, const uint64_t *hit_offsets, const slk_hit *hits,
                          const uint8_t *distinct, int32_t min_hit_groups, const double *thresholds, int32_t C,
                          int32_t *out_taxon, uint8_t *out_classified, int32_t *out_num_distinct, int32_t *out_total_kmers);

/* Same computation with every pointer (bases, offsets, mates, thresholds excepted: host) resident on the index's
 * GPU; asynchronous on st.  d_out_num_hits[R] (nullable) receives the span count per read (0 => no row);
 * d_out_num_probes[R] (nullable) the number of SEQUENCE_FLAG spans = table lookups (the P_r of SURVEY.md 8d).
 * total_bases / total_mate_bases = offsets[R] / mate_offsets[R] (the caller knows them; avoids a D2H sync). */
int32_t slk_classify_batch_device(slk_index *ix, slk_stream *st, const uint8_t *d_bases, const uint64_t *d_offsets,
                                  const uint8_t *d_mate_bases, const uint64_t *d_mate_offsets, uint64_t R,
                                  uint64_t total_bases, uint64_t total_mate_bases, int32_t min_hit_groups,
                                  const double *thresholds, int32_t C, int32_t *d_out_taxon,
                                  uint8_t *d_out_classified, int32_t *d_out_num_distinct,
                                  int32_t *d_out_total_kmers, int32_t *d_out_num_hits,
                                  int32_t *d_out_num_probes);

/* ---- staged device entry points: the same three steps as separate calls, for the TABLE-SHARDED mode (the record
 * table exceeds one GPU's HBM: each rank holds the records whose hash falls to it, minimizers are exchanged with an
 * all-to-all between slk_scan_device and slk_classify_hits_device; SURVEY.md 8e, BASELINE.json configs[3]) and for an
 * index build from sequences.  All pointers are device pointers; asynchronous on st.
 *   span slot of fragment r, span j:  offsets[r] (+ mate_offsets[r] + r when paired) + j   (at most L-k+1 [+1+L2-k+1]
 *   spans per fragment, so slots never overlap); d_span_* must hold total_bases (+ total_mate_bases + R) + 1 entries. */
/* getSpans (KeyValueIndex.scala:163-173): d_span_keys (left-aligned minimizers, 0 for flagged spans),
 * d_span_meta (kmers << 4 | flag << 1 | distinct), d_span_count[R]. */
int32_t slk_scan_device(slk_index *ix, slk_stream *st, const uint8_t *d_bases, const uint64_t *d_offsets,
                        const uint8_t *d_mate_bases, const uint64_t *d_mate_offsets, uint64_t R,
                        uint64_t *d_span_keys, int32_t *d_span_meta, int32_t *d_span_count);
/* the join (Classifier.scala:84): taxon or NONE for n minimizers against THIS index's records */
int32_t slk_lookup_device(slk_index *ix, slk_stream *st, const int64_t *d_keys, uint64_t n, int32_t *d_out_taxa);
/* Which rank owns a minimizer in table-sharded mode: fmix64(key) mod n_shards (host helper; the device side of the
 * Python host uses the same bijective mixer) */
uint32_t slk_shard_of(int64_t key, uint32_t n_shards);
/* The fast form of the table-sharded path for fragments of up to 1000 bases (both mates together): ONE scan per batch, and nothing
 * but 8-byte keys and 4-byte taxa on the links -- no span arrays in HBM, no per-probe return addresses, no compaction of the lists.
 * It replaces the shuffle behind the reference's join (S/slacken/Classifier.scala:84-95).  A batch passes through three JOBS, and one
 * slk_shard_step_device call launches up to three jobs of three different batches as ONE kernel, so that the latency-bound ones hide
 * behind the scan the way the probes of slk_classify_batch_device do:
 *   EMIT    scans the batch's fragments and appends every minimizer to the send region of its owner (slk_shard_of): d_send_keys is
 *           [n_shards][capacity_per_owner], filled from the front without holes -- what travels to owner g is
 *           d_send_keys[g * capacity .. + d_cursors[g]) as it stands (d_cursors[0 .. n_shards) after the step; a multiple of
 *           slk_shard_chunk(n_shards): the last few entries are zero keys whose answers nobody reads).  A cursor beyond the capacity
 *           means the region was too small: slk_stream_synchronize reports SLK_E_CAPACITY, emit the batch again with larger regions.
 *           What the APPLY of the same batch needs stays on this rank, in the other arrays of slk_shard_lists.
 *   LOOKUP  the OWNER's side: taxon or NONE for the n keys this rank received (any batch, any ranks), as slk_lookup_device.
 *   APPLY   replays the batch's probe log -- no second scan --, takes each probe's taxon from d_taxa ([n_shards][capacity_per_owner]:
 *           the owners' answers at the positions the keys had) and classifies: the outputs of slk_classify_batch_device.
 * (between the jobs: all-to-all of keys, all-to-all of taxa back into the same positions -- the caller's, e.g. RCCL.)
 * Any of emit / lookup / apply may be NULL.  A step without an EMIT runs the other jobs as kernels of their own.
 * d_defer[R] of a batch (zeroed by its EMIT) is 1 after its APPLY for the fragments this path does not take (longer than 1000
 * bases, more than 12 distinct taxa): classify those with the staged calls above.  SLK_E_UNSUPPORTED if the index's splitter is
 * outside the fused kernel's range (window wider than 32 m-mers, or taxon ids beyond 22 bits that slk_index_finalize could not
 * renumber): use the staged calls. */
typedef struct {
  const uint8_t *d_bases;       /* the batch (EMIT only; the APPLY reads d_offsets / d_mate_offsets when hit lists are written) */
  const uint64_t *d_offsets;
  const uint8_t *d_mate_bases;  /* nullable */
  const uint64_t *d_mate_offsets;
  uint64_t R;
  uint64_t total_bases;         /* d_offsets[R] / d_mate_offsets[R]: the caller knows them (spares a copy back to the host) */
  uint64_t total_mate_bases;
  uint32_t n_shards;
  uint32_t reserved;
  uint64_t capacity_per_owner;  /* entries of each owner's region: a multiple of slk_shard_chunk(n_shards), below 2^32 */
  int64_t *d_send_keys;         /* [n_shards][capacity_per_owner] */
  uint32_t *d_send_meta;        /* [n_shards][capacity_per_owner]: the span behind every key */
  uint64_t *d_cursors;          /* [n_shards + 3] (zeroed by the EMIT): [g] = entries to send to owner g; [n_shards + 2] = after the
                                   APPLY, the number of fragments flagged in d_defer */
  uint32_t *d_batch_log;        /* [slk_shard_batch_rows(..)][n_shards][4]: per batch of 64 probes, where each owner's keys went */
  uint32_t *d_tile_rows;        /* [ceil(R / 64)][2] */
  int32_t *d_read_info;         /* [R][2]: k-mers and spans of a fragment */
  int32_t *d_defer;             /* [R] */
  int32_t *d_span_meta;         /* hit lists (all three or none; span slots as for slk_scan_device): the EMIT writes the flagged spans, */
  int32_t *d_span_taxon;        /* the APPLY the hits; gather them with d_span_count[r] entries per fragment */
  int32_t *d_span_count;
} slk_shard_lists;
typedef struct {
  const int64_t *d_keys;
  uint64_t n;
  int32_t *d_out_taxa;
} slk_shard_lookup;
typedef struct {
  const int32_t *d_taxa;        /* [n_shards][capacity_per_owner] of the batch's lists */
  int32_t min_hit_groups;
  int32_t C;
  const double *thresholds;     /* host, [C] */
  int32_t *d_out_taxon;         /* [C * R] threshold-major */
  uint8_t *d_out_classified;
  int32_t *d_out_num_distinct;  /* nullable */
  int32_t *d_out_total_kmers;   /* nullable */
  int32_t *d_out_num_hits;      /* nullable: spans per fragment (0 => no row) */
} slk_shard_results;
uint32_t slk_shard_chunk(uint32_t n_shards);
uint64_t slk_shard_batch_rows(uint64_t total_bases, uint64_t total_mate_bases, uint64_t R, int32_t paired);
int32_t slk_shard_step_device(slk_index *ix, slk_stream *st, const slk_shard_lists *emit, const slk_shard_lookup *lookup,
                              const slk_shard_lists *apply_lists, const slk_shard_results *apply);
/* classifyHits (Classifier.scala:124-147, 439-454) over span slots whose taxa have been filled in (d_span_taxon: record
 * taxon / NONE for SEQUENCE spans; AMBIGUOUS and MATE_PAIR_BORDER spans are recognised from d_span_meta).  d_scratch:
 * as many 8-byte entries as span slots. */
int32_t slk_classify_hits_device(slk_index *ix, slk_stream *st, const uint64_t *d_offsets,
                                 const uint64_t *d_mate_offsets, uint64_t R, const int32_t *d_span_meta,
                                 const int32_t *d_span_taxon, const int32_t *d_span_count, uint64_t *d_scratch,
                                 int32_t min_hit_groups, const double *thresholds, int32_t C, int32_t *d_out_taxon,
                                 uint8_t *d_out_classified, int32_t *d_out_num_distinct, int32_t *d_out_total_kmers,
                                 int32_t *d_out_num_hits);

/* ---- table-sharded classification in ONE process (BASELINE.json configs[3]: a library whose table exceeds one GPU's HBM).
 * The host counterpart of the exchange the reference's join implies (S/slacken/Classifier.scala:84-95: the span rows are shuffled to
 * the records by minimizer, the hits shuffled back and regrouped by title).  n indices -- member g created on its device with
 * slk_index_set_shard(g, n), fed the whole record stream, finalized, all with the same splitter and taxonomy -- form a set; one
 * slk_shardset_classify call is one ROUND: member g classifies batches[g] (R may be 0: the member still answers the others' keys).
 * Every member scans its own fragments; 8-byte minimizers travel to their owners, 4-byte taxa travel back, nothing else moves.
 * The exchange is RCCL's (ncclSend / ncclRecv grouped over the members' streams; librccl is loaded at run time) when every member
 * has a device of its own, device-to-device copies otherwise (SLK_EXCHANGE_AUTO chooses) -- several members on ONE device is
 * how a one-GPU box tests this path.  A batch's arguments mean what slk_classify_batch's do (host pointers; hit lists optional),
 * and so do the results: bit-identical to the replicated mode.  One thread per set at a time; several sets may share members. */
typedef struct slk_shardset slk_shardset;
#define SLK_EXCHANGE_AUTO 0
#define SLK_EXCHANGE_RCCL 1
#define SLK_EXCHANGE_COPY 2
typedef struct {
  const uint8_t *bases;
  const uint64_t *offsets;
  const uint8_t *mate_bases;    /* nullable (single-end) */
  const uint64_t *mate_offsets; /* nullable */
  uint64_t R;
  int32_t *out_taxon;           /* [C*R] threshold-major */
  uint8_t *out_classified;      /* [C*R] */
  int32_t *out_num_distinct;    /* [R] nullable */
  int32_t *out_total_kmers;     /* [R] nullable */
  uint64_t *out_hit_offsets;    /* [R+1] nullable */
  slk_hit *out_hits;            /* nullable */
  uint64_t hits_capacity;
} slk_shard_batch;
int32_t slk_shardset_create(slk_index *const *members, int32_t n_members, int32_t exchange, slk_shardset **out);
int32_t slk_shardset_classify(slk_shardset *set, slk_shard_batch *batches /* [n_members] */, int32_t min_hit_groups,
                              const double *thresholds, int32_t C);
/* n_rounds rounds in one call, batches[round * n_members + member], PIPELINED: per step every member launches one kernel that scans
 * round t, answers the keys it received for round t - 2 and classifies round t - 4, while the keys of round t - 1 and the taxa of
 * round t - 3 travel on the members' exchange streams; the host's only wait per step is for the split sizes of the round emitted a
 * step earlier.  Results as n_rounds calls of slk_shardset_classify.  device_resident != 0: every pointer of the batches is a DEVICE
 * pointer on its member's GPU (reads and results stay in HBM; no hit lists: out_hit_offsets / out_hits must be NULL). */
int32_t slk_shardset_classify_rounds(slk_shardset *set, slk_shard_batch *batches /* [n_rounds][n_members] */, int32_t n_rounds,
                                     int32_t device_resident, int32_t min_hit_groups, const double *thresholds, int32_t C);
int32_t slk_shardset_exchange_mode(const slk_shardset *set); /* SLK_EXCHANGE_RCCL or SLK_EXCHANGE_COPY: what AUTO chose */
void slk_shardset_destroy(slk_shardset *set);

/* Per-stage device timing of the last slk_classify_batch_device call on st, in milliseconds (HIP events on the
 * stream the kernels ran on): [0]=scan, [1]=probe, [2]=classify.  Synchronises st. */
int32_t slk_stream_last_stage_ms(slk_stream *st, float out_ms[3]);
/* How many fragments of the last classify call on st the lane-per-fragment kernel handed to the wave-per-fragment kernel
 * (longer than 1000 bases, or more than 12 distinct taxa); 0 if that call did not take the lane kernel.  Synchronises st. */
int32_t slk_stream_last_deferred(slk_stream *st, uint64_t *out_count);

#ifdef __cplusplus
}
#endif
#endif

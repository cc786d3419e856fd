/*
 * slacken_oracle.h -- CPU restatement of the Slacken classify hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker for the HIP
 * engine in slacken_amd/.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path never links,
 * imports or calls anything in oracle/.
 *
 * PARITY PIN STATUS: partially pinned.  The reference (Scala/Spark) cannot be
 * compiled or run in the build container (no JVM), and its repository holds no
 * golden classify output.  This restatement is pinned against everything the
 * reference's own tests hold for this path:
 *   - the super-mer known-answer test   (MinSplitterTest.scala:25-33)
 *   - the spaced-seed documentation KAT (MinimizerPriorities.scala:274-277)
 *   - constants (toggle mask, special taxa, flags)
 *   - the reference's property specs, restated in tests/ (MinSplitterProps,
 *     SupermersProps, NTBitArrayProps, LowestCommonAncestorProps incl. its
 *     independent `correctClassification` spec)
 * End-to-end per-read output against the Spark path itself is UNPINNED.
 *
 * All file:line citations are relative to /root/reference/ and use
 *   S/ = src/main/scala/com/jnpersson/
 */
#ifndef SLACKEN_ORACLE_H
#define SLACKEN_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXW 4 /* 64-bit words per minimizer: m <= 128 */

/* S/slacken/package.scala:30-39 */
#define ORC_AMBIGUOUS_SPAN (-1)
#define ORC_MATE_PAIR_BORDER (-2)
#define ORC_SEQUENCE_FLAG 1
#define ORC_AMBIGUOUS_FLAG 2
#define ORC_MATE_PAIR_BORDER_FLAG 3
/* S/slacken/Taxonomy.scala:30-31 */
#define ORC_NONE 0
#define ORC_ROOT 1
/* S/kmers/minimizer/package.scala:32 */
#define ORC_DEFAULT_TOGGLE_MASK 0xe37e28c4271b5a2dULL

typedef struct {
  int k, m, spaces, canonical, W;
  uint64_t xor_mask;
  uint64_t mask[ORC_MAXW];  /* RandomXOR.mask, MinimizerPriorities.scala:146-160 */
  uint64_t space[ORC_MAXW]; /* SpacedSeed.spaceMask, :285-300 (all ones if spaces == 0) */
} orc_params;

/* OrdinalSpan, S/slacken/package.scala:61-62 (title omitted) */
typedef struct {
  uint64_t key[ORC_MAXW];
  int32_t kmers;
  int32_t flag;
  int32_t ordinal;
  int32_t distinct;
} orc_span;

/* Supermer (rank, start, length); MinSplitter.scala:63-72 */
typedef struct {
  uint64_t key[ORC_MAXW];
  int32_t start;
  int32_t length;
} orc_supermer;

/* TaxonHit, KeyValueIndex.scala:436-441 */
typedef struct {
  int32_t taxon;
  int32_t count;
} orc_hit;

typedef struct orc_index orc_index; /* (key words, taxon) records: the "join" side */

int orc_params_init(orc_params *p, int k, int m, int spaces, uint64_t xor_mask, int canonical);

/* BitRepresentation.charToTwobitWithInvalid :150-158 -> 0..3, 4 = whitespace, 5 = invalid */
int orc_char_to_twobit(int c);

/* priority of one left-aligned m-mer: SpacedSeed.writePriorityOf -> RandomXOR.writePriorityOf */
void orc_priority(const orc_params *p, const uint64_t *mmer, uint64_t *out);

/* left-aligned 2-bit encoding of an ACGTU string of length n <= 32*ORC_MAXW (NTBitArray.encode) */
void orc_encode(const char *s, int n, uint64_t *out);
void orc_reverse_complement(const uint64_t *in, int size, uint64_t *out);
void orc_canonical(const uint64_t *in, int size, uint64_t *out);

/* MinSplitter.splitEncode: returns the number of super-mers, or <0 on error. cap = capacity of out. */
int orc_split_encode(const orc_params *p, const char *seq, int n, orc_supermer *out, int cap);

/* Supermers.splitByAmbiguity: writes (start,len,flag) triples; returns the count. */
int orc_split_by_ambiguity(const char *seq, int n, int k, int32_t *starts, int32_t *lens, int32_t *flags, int cap);

/* Supermers.splitFragment + Supermers.spans for one (optionally paired) fragment.
 * seq2 == NULL for single reads. Returns the number of spans or <0 on error. */
int orc_spans(const orc_params *p, const char *seq1, int n1, const char *seq2, int n2, orc_span *out, int cap);

/* All SEQUENCE_FLAG span keys (word 0 only, W must be 1) of one sequence, in order: the (id1) side of
 * SplitterMinimizers.find (S/slacken/Minimizers.scala:43-76) used by the index build. Returns the count or <0. */
long orc_minimizer_keys(const orc_params *p, const char *seq, long n, int64_t *out_keys, long cap);
/* library construction (KeyValueIndex.makeRecords :85-93): minimizers of one library sequence; LCA-merged records of many */
long orc_library_minimizers(const orc_params *p, const char *seq, long n, int64_t *out_keys, long cap);
long orc_build_records(const orc_params *p, const int32_t *parents, int32_t T, const char *bases, const uint64_t *offsets,
                       const int32_t *taxa, long S, int64_t *out_keys, int32_t *out_taxa, long cap);

/* index ("records" table): keys are W words per record, left-aligned as in the Parquet id columns */
orc_index *orc_index_create(int W, const int64_t *keys, const int32_t *taxa, size_t n);
void orc_index_destroy(orc_index *ix);
/* returns taxon, or ORC_NONE when there is no record (left join + otherwise(NONE)) */
int32_t orc_index_lookup(const orc_index *ix, const uint64_t *key);

/* LowestCommonAncestor.apply :49-78 */
int32_t orc_lca(const int32_t *parents, int32_t T, int32_t a, int32_t b);
/* LowestCommonAncestor.resolveTree(Int2IntMap, Double) :101-146 over an insertion-ordered map */
int32_t orc_resolve_tree(const int32_t *parents, int32_t T, const int32_t *map_taxa, const int32_t *map_counts,
                         int n, double required_score);

/* Result of Classifier.classify (object) :439-454 for one read */
typedef struct {
  int32_t taxon;        /* reportTaxon */
  int32_t classified;   /* 0/1 */
  int32_t num_distinct; /* count_if(distinct && taxon != NONE), Classifier.scala:94 */
  int32_t total_kmers;  /* TaxonCounts.totalKmers */
  int32_t num_hits;     /* number of spans (0 => the read vanishes from the output) */
} orc_read_result;

/* Full per-read path: spans -> spanToHit -> sort by ordinal -> TaxonCounts -> resolveTree.
 * hits_out (nullable) receives the un-merged hits in ordinal order (cap entries). */
int orc_classify_read(const orc_params *p, const orc_index *ix, const int32_t *parents, int32_t T,
                      const char *seq1, int n1, const char *seq2, int n2, int min_hit_groups, double confidence,
                      orc_read_result *res, orc_hit *hits_out, int cap);

/* Kraken-style strings: TaxonCounts.lengthString :114-121 and pairsInOrderString :94-110 from
 * un-merged ordinal-ordered hits. Return the string length (excluding NUL) or <0 if cap is too small. */
int orc_length_string(const orc_hit *hits, int n, int k, char *out, int cap);
/* Classifier.classify (object, Classifier.scala:439-454) on a given list of hits in ordinal order (distinct[i] =
 * OrdinalSpan.distinct of hit i): used for rows merged from fragments that share a title (Classifier.scala:92,136). */
int orc_classify_hits(const int32_t *parents, int32_t T, const orc_hit *hits, const uint8_t *distinct, int n,
                      int min_hit_groups, double confidence, orc_read_result *res);
int orc_pairs_in_order_string(const orc_hit *hits, int n, char *out, int cap);

/* Batch form with the same argument meaning as slk_classify_batch (include/slacken_amd.h).
 * OpenMP over reads when built with -fopenmp; returns the number of threads used (>=1) or <0. */
int orc_classify_batch(const orc_params *p, const orc_index *ix, const int32_t *parents, int32_t T,
                       const uint8_t *bases, const uint64_t *offsets, const uint8_t *mate_bases,
                       const uint64_t *mate_offsets, size_t R, int min_hit_groups, const double *thresholds, int C,
                       int32_t *out_taxon, uint8_t *out_classified, int32_t *out_num_distinct,
                       int32_t *out_total_kmers, int32_t *out_num_hits);

#ifdef __cplusplus
}
#endif
#endif

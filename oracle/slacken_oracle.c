/*
 * slacken_oracle.c -- CPU restatement of the Slacken classify hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see slacken_oracle.h): parity checker for the HIP engine;
 * never linked or called by the product path.  PARITY PIN STATUS: partially pinned
 * (reference KATs and property specs; no runnable reference, no golden classify output).
 *
 * Each function cites the reference file:line it restates
 * (S/ = src/main/scala/com/jnpersson/, relative to /root/reference/).
 * The restatement is deliberately literal (deque window, outside-in canonical test,
 * insertion-ordered taxon map) so that the shortcuts taken by the GPU kernels
 * (window-min + RLE, min(fwd, rc)) are checked against the reference's own control flow.
 */
#include "slacken_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Java `x << n` / `x >>> n` on long use n mod 64 */
static inline uint64_t jshl(uint64_t x, int n) { return x << (n & 63); }
static inline uint64_t jshr(uint64_t x, int n) { return x >> (n & 63); }

/* NTBitArray.longsForSize, S/kmers/util/NTBitArray.scala:124-125 */
static inline int longs_for_size(int size) { return (size % 32 == 0) ? (size >> 5) : ((size >> 5) + 1); }

/* BitRepresentation.charToTwobitWithInvalid, S/kmers/util/BitRepresentation.scala:150-158 */
int orc_char_to_twobit(int c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case 'U': case 'u': return 3;
    case '\n': case '\r': return 4; /* WHITESPACE */
    default: return 5;              /* INVALID */
  }
}

/* BitRepresentation.isValid :140-143 */
static inline int is_valid_char(int c) { return orc_char_to_twobit(c) < 4; }

/* NTBitArray.apply :454-459 */
static inline int nt_at(const uint64_t *data, int pos) {
  return (int)((data[pos / 32] >> (2 * (31 - pos % 32))) & 3);
}

/* NTBitArray.shiftLongArrayKmerLeft :140-150 */
static inline void shift_add_bp(uint64_t *data, int W, int add, int k) {
  int i = 0;
  for (; i < W - 1; i++) data[i] = (data[i] << 2) | (data[i + 1] >> 62);
  int kmod32 = k & 31;
  data[i] = (data[i] << 2) | jshl((uint64_t)add, (32 - kmod32) * 2);
}

/* BitRepresentation.swapNTSequence :60-73 */
static inline uint64_t swap_nt_sequence(uint64_t kmer) {
  kmer = ((kmer & 0xCCCCCCCCCCCCCCCCULL) >> 2) | ((kmer & 0x3333333333333333ULL) << 2);
  kmer = ((kmer & 0xF0F0F0F0F0F0F0F0ULL) >> 4) | ((kmer & 0x0F0F0F0F0F0F0F0FULL) << 4);
  kmer = ((kmer & 0xFF00FF00FF00FF00ULL) >> 8) | ((kmer & 0x00FF00FF00FF00FFULL) << 8);
  kmer = ((kmer & 0xFFFF0000FFFF0000ULL) >> 16) | ((kmer & 0x0000FFFF0000FFFFULL) << 16);
  return (kmer >> 32) | (kmer << 32);
}

/* NTBitArray.writeReverseComplement :231-247 (reverseComplementLeftAligned with mask -1L, BitRepresentation :83-84) */
void orc_reverse_complement(const uint64_t *in, int size, uint64_t *out) {
  int l = longs_for_size(size);
  int shiftAmt = (size % 32) * 2;
  out[0] = swap_nt_sequence(in[l - 1]) ^ ~0ULL;
  for (int i = 1; i < l; i++) {
    out[i] = swap_nt_sequence(in[l - 1 - i]) ^ ~0ULL;
    if (shiftAmt > 0) out[i - 1] = jshl(out[i - 1], 64 - shiftAmt) | jshr(out[i], shiftAmt);
  }
  out[l - 1] = jshl(out[l - 1], 64 - shiftAmt);
}

/* NTBitArray.sliceIsForwardOrientation :437-452 with pos = 0 */
static int is_forward_orientation(const uint64_t *data, int size) {
  int st = 0, end = size - 1;
  while (st < end) {
    int a = nt_at(data, st);
    int b = (~nt_at(data, end)) & 3; /* complementOne, BitRepresentation :47 */
    if (a < b) return 1;
    if (a > b) return 0;
    st++;
    end--;
  }
  return nt_at(data, st) < 2; /* apply(st) < G */
}

/* NTBitArray.writeCanonical :258-266 */
void orc_canonical(const uint64_t *in, int size, uint64_t *out) {
  int l = longs_for_size(size);
  if (is_forward_orientation(in, size)) memcpy(out, in, sizeof(uint64_t) * l);
  else orc_reverse_complement(in, size, out);
}

/* NTBitArray.encode :78-98 (left-aligned, MSB first, A-padded) */
void orc_encode(const char *s, int n, uint64_t *out) {
  int l = longs_for_size(n);
  for (int i = 0; i < l; i++) out[i] = 0;
  for (int i = 0; i < n; i++) out[i / 32] |= (uint64_t)orc_char_to_twobit(s[i]) << (2 * (31 - i % 32));
}

/* NTBitArray.<<= :339-362 */
static void bitarray_shl(uint64_t *data, int len, int bits) {
  int write = 0, shift = bits, read = 0;
  while (shift > 63) { read++; shift -= 64; }
  while (write < len) {
    if (read < len - 1 && shift > 0) data[write] = (data[read] << shift) | (data[read + 1] >> (64 - shift));
    else if (read < len) data[write] = data[read] << shift;
    else data[write] = 0;
    write++;
    read++;
  }
}

int orc_params_init(orc_params *p, int k, int m, int spaces, uint64_t xor_mask, int canonical) {
  if (m < 1 || k < m || spaces < 0 || spaces > m / 2) return -1; /* assert (s <= inner.width / 2), :282 */
  int W = longs_for_size(m);
  if (W > ORC_MAXW) return -2;
  memset(p, 0, sizeof(*p));
  p->k = k; p->m = m; p->spaces = spaces; p->canonical = canonical; p->W = W; p->xor_mask = xor_mask;
  /* RandomXOR.mask, S/kmers/minimizer/MinimizerPriorities.scala:146-160 */
  for (int i = 0; i < W; i++) {
    if (i == W - 1 && (m % 32 != 0)) p->mask[i] = jshl(xor_mask, 64 - (m % 32) * 2);
    else p->mask[i] = xor_mask;
  }
  /* SpacedSeed.spaceMask :285-300: NTBitArray.fill(-1, width) (NTBitArray.scala:105-112), then s x (<<= 4; |= finalBits) */
  for (int i = 0; i < W; i++) p->space[i] = ~0ULL;
  if (m % 32 != 0) p->space[W - 1] &= jshl(~0ULL, 64 - (m % 32) * 2);
  uint64_t finalBits = jshl(3ULL, 64 - (m % 32) * 2);
  for (int i = 0; i < spaces; i++) {
    bitarray_shl(p->space, W, 4);
    p->space[W - 1] |= finalBits;
  }
  return 0;
}

/* SpacedSeed.writePriorityOf :308-312 -> RandomXOR.writePriorityOf :165-175 */
void orc_priority(const orc_params *p, const uint64_t *mmer, uint64_t *out) {
  if (p->canonical) orc_canonical(mmer, p->m, out);
  else memcpy(out, mmer, sizeof(uint64_t) * p->W);
  for (int i = 0; i < p->W; i++) out[i] = (out[i] ^ p->mask[i]) & p->space[i];
}

/* ---- growable scratch (one per thread) ---- */
typedef struct {
  uint64_t *keys; /* MinimizerPositions data: nvalid x W */
  uint8_t *valid; /* MinimizerPositions valid tag */
  size_t cap;
  orc_supermer *sm;
  size_t sm_cap;
  orc_span *spans;
  size_t spans_cap;
} scratch_t;

static void scratch_reserve(scratch_t *s, size_t n, int W) {
  if (n > s->cap) {
    s->cap = n * 2 + 64;
    s->keys = (uint64_t *)realloc(s->keys, s->cap * ORC_MAXW * sizeof(uint64_t));
    s->valid = (uint8_t *)realloc(s->valid, s->cap);
    s->sm_cap = s->cap;
    s->sm = (orc_supermer *)realloc(s->sm, s->sm_cap * sizeof(orc_supermer));
  }
  (void)W;
}
static void scratch_free(scratch_t *s) {
  free(s->keys); free(s->valid); free(s->sm); free(s->spans);
  memset(s, 0, sizeof(*s));
}

/* MinimizerPositions.compare :55-63 (Long.compareUnsigned per word) */
static inline int mp_compare(const uint64_t *keys, int W, int p1, int p2) {
  for (int i = 0; i < W; i++) {
    uint64_t a = keys[(size_t)p1 * W + i], b = keys[(size_t)p2 * W + i];
    if (a != b) return a < b ? -1 : 1;
  }
  return 0;
}

/* ShiftScanner.allMatches(data, size) :90-159. Returns the number of valid characters (= matches.length),
 * or -1 on an invalid character (InvalidNucleotideException). */
static int all_matches(const orc_params *p, const char *seq, int n, uint64_t *keys, uint8_t *valid) {
  int W = p->W, width = p->m;
  uint64_t window[ORC_MAXW] = {0};
  int validSize = 0, pos = 0;
  while (validSize < width - 1 && pos < n) {
    int x = orc_char_to_twobit((unsigned char)seq[pos]);
    if (x == 5) return -1;
    if (x != 4) {
      for (int i = 0; i < W; i++) keys[(size_t)validSize * W + i] = 0; /* invalidMinimizer */
      valid[validSize] = 0;
      shift_add_bp(window, W, x, width);
      validSize++;
    }
    pos++;
  }
  while (pos < n) {
    int x = orc_char_to_twobit((unsigned char)seq[pos]);
    if (x == 5) return -1;
    if (x != 4) {
      shift_add_bp(window, W, x, width);
      orc_priority(p, window, &keys[(size_t)validSize * W]);
      valid[validSize] = 1; /* RandomXOR never returns `empty` */
      validSize++;
    }
    pos++;
  }
  return validSize;
}

/* The scanner's matches by themselves (what ShiftScannerProps.scala:28-58 looks at): keys[i * W ..] / valid[i] for the i-th VALID
 * character of seq; the caller's arrays hold n entries.  Returns their number, -1 on an invalid character. */
int orc_all_matches(const orc_params *p, const char *seq, int n, uint64_t *keys, uint8_t *valid) {
  return all_matches(p, seq, n, keys, valid);
}

/* PosRankWindow, S/kmers/minimizer/PosRankWindow.scala:33-97 */
typedef struct {
  int m, k, W, length;
  int leftBound, rightBound;
  const uint64_t *keys;
  uint8_t *valid;
} prw_t;

static void prw_advance(prw_t *w) { /* advanceWindow :47-75 */
  w->rightBound++;
  if (w->rightBound > w->length) return;
  int inserted = w->rightBound - 1;
  if (w->valid[inserted]) {
    int test = w->rightBound - 2;
    while (test >= w->leftBound + 1 && (!w->valid[test] || mp_compare(w->keys, w->W, test, inserted) > 0)) {
      w->valid[test] = 0;
      test--;
    }
    if (!w->valid[w->leftBound] || mp_compare(w->keys, w->W, inserted, w->leftBound) < 0) w->leftBound++;
  }
  while (w->rightBound - w->leftBound > w->k - (w->m - 1) ||
         (w->leftBound < w->length && !w->valid[w->leftBound]))
    w->leftBound++;
}

/* MinSplitter.splitRead(encoded, matches) :133-172 over MinimizerPositions from all_matches */
static int split_read(const orc_params *p, const uint64_t *keys, uint8_t *valid, int nvalid, orc_supermer *out, int cap) {
  prw_t w = {p->m, p->k, p->W, nvalid, 0, 0, keys, valid};
  while (w.rightBound < p->k) prw_advance(&w); /* :42-44 */
  int regionStart = 0, count = 0;
  while (w.rightBound <= w.length) { /* hasNext :96 */
    int pos = w.leftBound;          /* next :86-93 */
    if (pos >= w.length) return -3;
    prw_advance(&w);
    if (!valid[pos]) return -4;
    int consumed = 1;
    while (w.rightBound <= w.length && (w.leftBound == pos || mp_compare(keys, p->W, w.leftBound, pos) == 0)) {
      /* window.next */
      if (w.leftBound >= w.length) return -3;
      prw_advance(&w);
      consumed++;
    }
    int thisStart = regionStart;
    regionStart += consumed;
    if (count >= cap) return -5;
    memcpy(out[count].key, &keys[(size_t)pos * p->W], sizeof(uint64_t) * p->W);
    for (int i = p->W; i < ORC_MAXW; i++) out[count].key[i] = 0;
    out[count].start = thisStart;
    out[count].length = (w.rightBound <= w.length) ? consumed + (p->k - 1) : nvalid - thisStart;
    count++;
  }
  return count;
}

static int split_encode_scratch(const orc_params *p, const char *seq, int n, scratch_t *s, orc_supermer *out, int cap) {
  scratch_reserve(s, (size_t)n + 1, p->W);
  int nvalid = all_matches(p, seq, n, s->keys, s->valid);
  if (nvalid < 0) return nvalid;
  return split_read(p, s->keys, s->valid, nvalid, out, cap);
}

/* MinSplitter.splitEncode :98-101 */
int orc_split_encode(const orc_params *p, const char *seq, int n, orc_supermer *out, int cap) {
  scratch_t s = {0};
  int r = split_encode_scratch(p, seq, n, &s, out, cap);
  scratch_free(&s);
  return r;
}

/* regex [actguACTGU\n\r], S/slacken/Supermers.scala:141 */
static inline int in_nonambig_class(int c) { return orc_char_to_twobit(c) < 5; }

/* Supermers.enoughValidChars :180-189 */
static int enough_valid_chars(const char *s, int n, int min) {
  int c = 0;
  for (int i = 0; i < n; i++) {
    if (is_valid_char((unsigned char)s[i])) c++;
    if (c == min) return 1;
  }
  return 0;
}

/* Supermers.splitByAmbiguity :150-178 */
int orc_split_by_ambiguity(const char *seq, int n, int k, int32_t *starts, int32_t *lens, int32_t *flags, int cap) {
  int at = 0, count = 0;
  while (at < n) {
    int end = at;
    int matching = in_nonambig_class((unsigned char)seq[at]);
    while (end < n && in_nonambig_class((unsigned char)seq[end]) == matching) end++;
    if (count >= cap) return -5;
    starts[count] = at;
    lens[count] = end - at;
    if (matching) flags[count] = enough_valid_chars(seq + at, end - at, k) ? ORC_SEQUENCE_FLAG : ORC_AMBIGUOUS_FLAG;
    else flags[count] = ORC_AMBIGUOUS_FLAG;
    count++;
    at = end;
  }
  return count;
}

typedef struct {
  int first;
  int have_last;
  uint64_t last[ORC_MAXW];
  int ordinal;
} span_state;

static int push_span(orc_span **out, int *count, int cap, span_state *st, const uint64_t *key, int W, int size,
                     int flag, int k) {
  if (*count >= cap) return -5;
  orc_span *sp = &(*out)[*count];
  /* Supermers.spans :70-97 */
  int seqlike = (flag != ORC_AMBIGUOUS_FLAG && flag != ORC_MATE_PAIR_BORDER_FLAG);
  int equal_last = st->have_last && memcmp(key, st->last, sizeof(uint64_t) * W) == 0;
  sp->distinct = seqlike && (st->first || !equal_last);
  if (seqlike) {
    memcpy(st->last, key, sizeof(uint64_t) * W);
    st->have_last = 1;
  }
  st->first = 0;
  for (int i = 0; i < ORC_MAXW; i++) sp->key[i] = (i < W && seqlike) ? key[i] : 0; /* random id for flagged spans: unobservable */
  sp->kmers = size - (k - 1);
  sp->flag = flag;
  sp->ordinal = st->ordinal++;
  (*count)++;
  return 0;
}

/* Supermers.splitFragment(NTSeq) :113-125 feeding Supermers.spans */
static int spans_one(const orc_params *p, const char *seq, int n, scratch_t *s, orc_span **out, int *count, int cap,
                     span_state *st) {
  int at = 0;
  while (at < n) { /* splitByAmbiguity, inlined so no per-read segment array is needed */
    int end = at;
    int matching = in_nonambig_class((unsigned char)seq[at]);
    while (end < n && in_nonambig_class((unsigned char)seq[end]) == matching) end++;
    int len = end - at;
    int flag = (matching && enough_valid_chars(seq + at, len, p->k)) ? ORC_SEQUENCE_FLAG : ORC_AMBIGUOUS_FLAG;
    if (len >= p->k) { /* `if ntseq.length >= k` :116 */
      if (flag == ORC_AMBIGUOUS_FLAG) {
        uint64_t zero[ORC_MAXW] = {0};
        int r = push_span(out, count, cap, st, zero, p->W, len, ORC_AMBIGUOUS_FLAG, p->k);
        if (r < 0) return r;
      } else {
        scratch_reserve(s, (size_t)len + 1, p->W);
        int ns = split_encode_scratch(p, seq + at, len, s, s->sm, (int)s->sm_cap);
        if (ns < 0) return ns;
        for (int i = 0; i < ns; i++) {
          int r = push_span(out, count, cap, st, s->sm[i].key, p->W, s->sm[i].length, ORC_SEQUENCE_FLAG, p->k);
          if (r < 0) return r;
        }
      }
    }
    at = end;
  }
  return 0;
}

/* Supermers.splitFragment(InputFragment) :49-66 + spans :70-97 */
static int spans_scratch(const orc_params *p, const char *seq1, int n1, const char *seq2, int n2, scratch_t *s,
                         orc_span *out, int cap) {
  span_state st = {1, 0, {0}, 0};
  int count = 0;
  int r = spans_one(p, seq1, n1, s, &out, &count, cap, &st);
  if (r < 0) return r;
  if (seq2) {
    uint64_t zero[ORC_MAXW] = {0};
    r = push_span(&out, &count, cap, &st, zero, p->W, 0, ORC_MATE_PAIR_BORDER_FLAG, p->k); /* emptySupermer :53-56 */
    if (r < 0) return r;
    r = spans_one(p, seq2, n2, s, &out, &count, cap, &st);
    if (r < 0) return r;
  }
  return count;
}

int orc_spans(const orc_params *p, const char *seq1, int n1, const char *seq2, int n2, orc_span *out, int cap) {
  scratch_t s = {0};
  int r = spans_scratch(p, seq1, n1, seq2, n2, &s, out, cap);
  scratch_free(&s);
  return r;
}

long orc_minimizer_keys(const orc_params *p, const char *seq, long n, int64_t *out_keys, long cap) {
  if (p->W != 1) return -2;
  orc_span *sp = (orc_span *)malloc(sizeof(orc_span) * (size_t)(n + 2));
  scratch_t s = {0};
  int ns = spans_scratch(p, seq, (int)n, NULL, 0, &s, sp, (int)(n + 2));
  long cnt = 0;
  if (ns >= 0) {
    for (int i = 0; i < ns; i++)
      if (sp[i].flag == ORC_SEQUENCE_FLAG) {
        if (cnt >= cap) { cnt = -5; break; }
        out_keys[cnt++] = (int64_t)sp[i].key[0];
      }
  } else cnt = ns;
  s.spans = NULL;
  scratch_free(&s);
  free(sp);
  return cnt;
}

/* ---- library construction (BASELINE config 5): KeyValueIndex.makeRecords, S/slacken/KeyValueIndex.scala:85-93 ---- */

/* One (taxon, sequence) row of SplitterMinimizers.find (S/slacken/Minimizers.scala:43-56) after InputReader.removeInvalid
 * (S/kmers/input/InputReader.scala:60-72): every match of the regex [ACTGUactgu][ACTGUactgu\n\r]* is one fragment, and
 * superkmerPositions (MinSplitter.scala:114-117,180-216; same window walk as splitRead) yields one minimizer per super-mer. */
long orc_library_minimizers(const orc_params *p, const char *seq, long n, int64_t *out_keys, long cap) {
  if (p->W != 1) return -2;
  long cnt = 0, pos = 0;
  scratch_t s = {0};
  orc_supermer *sm = NULL;
  long sm_cap = 0;
  while (pos < n) {
    if (orc_char_to_twobit((unsigned char)seq[pos]) >= 4) { pos++; continue; } /* a match starts at a nucleotide */
    long end = pos + 1;
    while (end < n && orc_char_to_twobit((unsigned char)seq[end]) < 5) end++;   /* nucleotides and \n \r */
    long len = end - pos;
    if (len + 2 > sm_cap) { free(sm); sm_cap = len + 2; sm = (orc_supermer *)malloc(sizeof(orc_supermer) * (size_t)sm_cap); }
    int ns = split_encode_scratch(p, seq + pos, (int)len, &s, sm, (int)sm_cap);
    if (ns < 0) { cnt = ns; break; }
    for (int i = 0; i < ns; i++) {
      if (cnt >= cap) { cnt = -5; break; }
      out_keys[cnt++] = (int64_t)sm[i].key[0];
    }
    if (cnt < 0) break;
    pos = end;
  }
  free(sm);
  scratch_free(&s);
  return cnt;
}

typedef struct { int64_t key; int32_t taxon; long order; } build_pair;
static int build_pair_cmp(const void *a, const void *b) {
  const build_pair *x = (const build_pair *)a, *y = (const build_pair *)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->order < y->order ? -1 : (x->order > y->order);
}

/* minimizersTaxa.groupBy(id).agg(TaxonLCA) (KeyValueIndex.scala:85-93; TaxonLCA: zero = NONE, reduce = lca,
 * LowestCommonAncestor.scala:152-170).  Output sorted by key (signed).  Returns the number of records, -5 if cap is short. */
long orc_build_records(const orc_params *p, const int32_t *parents, int32_t T, const char *bases, const uint64_t *offsets,
                       const int32_t *taxa, long S, int64_t *out_keys, int32_t *out_taxa, long cap) {
  long total = 0, np = 0;
  for (long i = 0; i < S; i++) total += (long)(offsets[i + 1] - offsets[i]);
  build_pair *pairs = (build_pair *)malloc(sizeof(build_pair) * (size_t)(total + 1));
  int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(total + 1));
  long rc = 0;
  for (long i = 0; i < S && rc >= 0; i++) {
    long len = (long)(offsets[i + 1] - offsets[i]);
    long n = orc_library_minimizers(p, bases + offsets[i], len, tmp, len + 1);
    if (n < 0) { rc = n; break; }
    for (long j = 0; j < n; j++) { pairs[np].key = tmp[j]; pairs[np].taxon = taxa[i]; pairs[np].order = np; np++; }
  }
  long cnt = 0;
  if (rc >= 0) {
    qsort(pairs, (size_t)np, sizeof(build_pair), build_pair_cmp);
    for (long i = 0; i < np;) {
      int32_t acc = ORC_NONE; /* zero */
      long j = i;
      for (; j < np && pairs[j].key == pairs[i].key; j++) acc = orc_lca(parents, T, acc, pairs[j].taxon);
      if (cnt >= cap) { cnt = -5; break; }
      out_keys[cnt] = pairs[i].key;
      out_taxa[cnt] = acc;
      cnt++;
      i = j;
    }
  } else cnt = rc;
  free(pairs);
  free(tmp);
  return cnt;
}

/* ---- index: the records side of `taggedSpans.join(index.records, idColumnNames, "left")`, Classifier.scala:84 ---- */
struct orc_index {
  int W;
  size_t cap; /* power of two */
  uint64_t *keys;
  int32_t *taxa;
  uint8_t *used;
};

static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
static inline uint64_t hash_key(const uint64_t *key, int W) {
  uint64_t h = 0x9e3779b97f4a7c15ULL;
  for (int i = 0; i < W; i++) h = mix64(h ^ key[i]);
  return h;
}

orc_index *orc_index_create(int W, const int64_t *keys, const int32_t *taxa, size_t n) {
  orc_index *ix = (orc_index *)calloc(1, sizeof(orc_index));
  ix->W = W;
  size_t cap = 16;
  while (cap < 2 * n + 1) cap <<= 1;
  ix->cap = cap;
  ix->keys = (uint64_t *)calloc(cap * W, sizeof(uint64_t));
  ix->taxa = (int32_t *)calloc(cap, sizeof(int32_t));
  ix->used = (uint8_t *)calloc(cap, 1);
  for (size_t r = 0; r < n; r++) {
    const uint64_t *key = (const uint64_t *)&keys[r * W];
    size_t h = hash_key(key, W) & (cap - 1);
    while (ix->used[h]) {
      if (memcmp(&ix->keys[h * W], key, sizeof(uint64_t) * W) == 0) break; /* keys are unique by contract; keep the first */
      h = (h + 1) & (cap - 1);
    }
    if (!ix->used[h]) {
      ix->used[h] = 1;
      memcpy(&ix->keys[h * W], key, sizeof(uint64_t) * W);
      ix->taxa[h] = taxa[r];
    }
  }
  return ix;
}

void orc_index_destroy(orc_index *ix) {
  if (!ix) return;
  free(ix->keys); free(ix->taxa); free(ix->used); free(ix);
}

int32_t orc_index_lookup(const orc_index *ix, const uint64_t *key) {
  size_t h = hash_key(key, ix->W) & (ix->cap - 1);
  while (ix->used[h]) {
    if (memcmp(&ix->keys[h * ix->W], key, sizeof(uint64_t) * ix->W) == 0) return ix->taxa[h];
    h = (h + 1) & (ix->cap - 1);
  }
  return ORC_NONE; /* otherwise(lit(Taxonomy.NONE)), KeyValueIndex.scala:176-185 */
}

/* ---- taxonomy / LCA ---- */
#define PATH_MAX_LENGTH 256 /* LowestCommonAncestor.scala:34 */

static inline int32_t parent_of(const int32_t *parents, int32_t T, int32_t t) {
  return (t >= 0 && t < T) ? parents[t] : ORC_NONE; /* out-of-range would throw in the reference */
}

/* LowestCommonAncestor.apply :49-78 */
int32_t orc_lca(const int32_t *parents, int32_t T, int32_t tax1, int32_t tax2) {
  if (tax1 == ORC_NONE || tax2 == ORC_NONE) return tax2 == ORC_NONE ? tax1 : tax2;
  int32_t path[PATH_MAX_LENGTH + 1];
  int32_t a = tax1;
  int i = 0;
  while (a != ORC_NONE && i < PATH_MAX_LENGTH) {
    path[i++] = a;
    a = parent_of(parents, T, a);
  }
  path[i] = ORC_NONE;
  int32_t b = tax2;
  while (b != ORC_NONE) {
    for (i = 0; path[i] != ORC_NONE; i++)
      if (path[i] == b) return b;
    b = parent_of(parents, T, b);
  }
  return ORC_ROOT;
}

static inline int32_t map_get(const int32_t *map_taxa, const int32_t *map_counts, int n, int32_t t) {
  for (int i = 0; i < n; i++)
    if (map_taxa[i] == t) return map_counts[i];
  return 0; /* Int2IntMap default value */
}

/* Taxonomy.hasAncestor :236-237 via stepsToAncestor :241-244 */
static int has_ancestor(const int32_t *parents, int32_t T, int32_t tax, int32_t ancestor) {
  int32_t t = tax; /* pathToRoot :204-215: empty for NONE */
  while (t != ORC_NONE) {
    if (t == ancestor) return 1;
    t = parent_of(parents, T, t);
  }
  return 0;
}

/* LowestCommonAncestor.resolveTree(hitCounts, requiredScore) :101-146 */
int32_t orc_resolve_tree(const int32_t *parents, int32_t T, const int32_t *map_taxa, const int32_t *map_counts,
                         int n, double requiredScore) {
  int32_t maxTaxon = 0;
  int maxScore = 0;
  for (int it = 0; it < n; it++) {
    int32_t taxon = map_taxa[it];
    int32_t node = taxon;
    int score = 0;
    while (node != ORC_NONE) {
      score += map_get(map_taxa, map_counts, n, node);
      node = parent_of(parents, T, node);
    }
    if (score > maxScore) {
      maxTaxon = taxon;
      maxScore = score;
    } else if (score == maxScore) {
      maxTaxon = orc_lca(parents, T, maxTaxon, taxon);
    }
  }
  maxScore = map_get(map_taxa, map_counts, n, maxTaxon);
  while (maxTaxon != ORC_NONE && (double)maxScore < requiredScore) {
    maxScore = 0;
    for (int it = 0; it < n; it++)
      if (has_ancestor(parents, T, map_taxa[it], maxTaxon)) maxScore += map_counts[it];
    if ((double)maxScore >= requiredScore) return maxTaxon;
    maxTaxon = parent_of(parents, T, maxTaxon);
  }
  return maxTaxon;
}

/* TaxonCounts.fromHits :31-48 + toMap :70-81 + totalKmers :84-87, then resolveTree :91-96 and Classifier.classify :439-454 */
static void classify_hits(const int32_t *parents, int32_t T, const orc_hit *hits, const uint8_t *distinct, int n,
                          int min_hit_groups, double confidence, int32_t *mt, int32_t *mc, orc_read_result *res) {
  /* fromHits: merge adjacent equal taxa -> (mt, mc)[0..nm) */
  int nm = 0;
  for (int i = 0; i < n; i++) {
    if (nm > 0 && mt[nm - 1] == hits[i].taxon) mc[nm - 1] += hits[i].count;
    else { mt[nm] = hits[i].taxon; mc[nm] = hits[i].count; nm++; }
  }
  /* toMap (insertion ordered Int2IntArrayMap, omitting AMBIGUOUS and MATE_PAIR_BORDER) and totalKmers */
  int32_t *kt = mt + n, *kc = mc + n;
  int nk = 0, total = 0;
  for (int i = 0; i < nm; i++) {
    if (mt[i] != ORC_MATE_PAIR_BORDER) total += mc[i];
    if (mt[i] == ORC_AMBIGUOUS_SPAN || mt[i] == ORC_MATE_PAIR_BORDER) continue;
    int j = 0;
    for (; j < nk; j++) if (kt[j] == mt[i]) break;
    if (j == nk) { kt[nk] = mt[i]; kc[nk] = 0; nk++; }
    kc[j] += mc[i];
  }
  double requiredScore = ceil(confidence * (double)total); /* Math.ceil(confidenceThreshold * totalKmers) :94 */
  int32_t taxon = orc_resolve_tree(parents, T, kt, kc, nk, requiredScore);
  int nd = 0;
  for (int i = 0; i < n; i++) nd += (distinct[i] && hits[i].taxon != ORC_NONE); /* Classifier.scala:94 */
  int classified = taxon != ORC_NONE && nd >= min_hit_groups; /* :445 */
  res->taxon = classified ? taxon : ORC_NONE;
  res->classified = classified;
  res->num_distinct = nd;
  res->total_kmers = total;
  res->num_hits = n;
}

typedef struct {
  scratch_t s;
  orc_hit *hits;
  uint8_t *distinct;
  int32_t *mt, *mc;
  size_t cap;
} cls_scratch;

static void cls_reserve(cls_scratch *c, size_t nspans) {
  if (nspans > c->cap) {
    c->cap = nspans * 2 + 64;
    c->s.spans = (orc_span *)realloc(c->s.spans, c->cap * sizeof(orc_span));
    c->s.spans_cap = c->cap;
    c->hits = (orc_hit *)realloc(c->hits, c->cap * sizeof(orc_hit));
    c->distinct = (uint8_t *)realloc(c->distinct, c->cap);
    c->mt = (int32_t *)realloc(c->mt, 2 * c->cap * sizeof(int32_t));
    c->mc = (int32_t *)realloc(c->mc, 2 * c->cap * sizeof(int32_t));
  }
}
static void cls_free(cls_scratch *c) {
  scratch_free(&c->s);
  free(c->hits); free(c->distinct); free(c->mt); free(c->mc);
  memset(c, 0, sizeof(*c));
}

static int classify_read_scratch(const orc_params *p, const orc_index *ix, const int32_t *parents, int32_t T,
                                 const char *seq1, int n1, const char *seq2, int n2, int min_hit_groups,
                                 const double *thresholds, int C, cls_scratch *c, orc_read_result *res) {
  cls_reserve(c, (size_t)n1 + (size_t)(seq2 ? n2 : 0) + 2);
  int ns = spans_scratch(p, seq1, n1, seq2, n2, &c->s, c->s.spans, (int)c->s.spans_cap);
  if (ns < 0) return ns;
  for (int i = 0; i < ns; i++) { /* spanToHit, KeyValueIndex.scala:176-185 */
    const orc_span *sp = &c->s.spans[i];
    int32_t taxon;
    if (sp->flag == ORC_AMBIGUOUS_FLAG) taxon = ORC_AMBIGUOUS_SPAN;
    else if (sp->flag == ORC_MATE_PAIR_BORDER_FLAG) taxon = ORC_MATE_PAIR_BORDER;
    else taxon = orc_index_lookup(ix, sp->key);
    c->hits[i].taxon = taxon;
    c->hits[i].count = sp->kmers;
    c->distinct[i] = (uint8_t)sp->distinct;
  }
  /* hits are produced in ordinal order, so Arrays.sort(hits, hitsComparator) (Classifier.scala:136) is the identity */
  for (int t = 0; t < C; t++)
    classify_hits(parents, T, c->hits, c->distinct, ns, min_hit_groups, thresholds[t], c->mt, c->mc, &res[t]);
  return ns;
}

int orc_classify_read(const orc_params *p, const orc_index *ix, const int32_t *parents, int32_t T, const char *seq1,
                      int n1, const char *seq2, int n2, int min_hit_groups, double confidence, orc_read_result *res,
                      orc_hit *hits_out, int cap) {
  cls_scratch c = {0};
  int ns = classify_read_scratch(p, ix, parents, T, seq1, n1, seq2, n2, min_hit_groups, &confidence, 1, &c, res);
  if (ns >= 0 && hits_out) {
    if (ns > cap) ns = -5;
    else memcpy(hits_out, c.hits, sizeof(orc_hit) * ns);
  }
  cls_free(&c);
  return ns;
}

/* TaxonCounts.lengthString :114-121, on the fromHits-merged (taxa, counts) as the reference computes it.  A single fragment
   has at most one border; a row merged from several fragments that share a title (Classifier.scala:92) can have several, and
   adjacent ones (equal ordinals) collapse into ONE merged entry before indexOf / take / drop are applied. */
int orc_length_string(const orc_hit *hits, int n, int k, char *out, int cap) {
  int32_t *mt = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  long *mc = (long *)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1));
  int nm = 0;
  for (int i = 0; i < n; i++) { /* fromHits :31-48 */
    if (nm > 0 && mt[nm - 1] == hits[i].taxon) mc[nm - 1] += hits[i].count;
    else { mt[nm] = hits[i].taxon; mc[nm] = hits[i].count; nm++; }
  }
  int border = -1;
  for (int i = 0; i < nm; i++) if (mt[i] == ORC_MATE_PAIR_BORDER) { border = i; break; } /* taxa.indexOf */
  long a = 0, b = 0;
  int r;
  if (border == -1) {
    for (int i = 0; i < nm; i++) a += mc[i];                       /* counts.sum */
    r = snprintf(out, cap, "%ld", a + (k - 1));
  } else {
    for (int i = 0; i < border; i++) a += mc[i];                   /* counts.take(border).sum */
    for (int i = border + 1; i < nm; i++) b += mc[i];              /* counts.drop(border + 1).sum */
    r = snprintf(out, cap, "%ld|%ld", a + (k - 1), b + (k - 1));
  }
  free(mt); free(mc);
  return (r < 0 || r >= cap) ? -5 : r;
}

/* Classifier.classify (object, :439-454) on a caller-assembled hit list: what classifyHits does with the hits that
   groupBy("seqTitle") collected (:92) once they are sorted by ordinal (:136). */
int orc_classify_hits(const int32_t *parents, int32_t T, const orc_hit *hits, const uint8_t *distinct, int n,
                      int min_hit_groups, double confidence, orc_read_result *res) {
  int32_t *mt = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * n + 2));
  int32_t *mc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * n + 2));
  if (!mt || !mc) { free(mt); free(mc); return -1; }
  classify_hits(parents, T, hits, distinct, n, min_hit_groups, confidence, mt, mc, res);
  free(mt); free(mc);
  return 0;
}

/* TaxonCounts.pairsInOrderString :94-110 over fromHits-merged pairs */
int orc_pairs_in_order_string(const orc_hit *hits, int n, char *out, int cap) {
  int len = 0, i = 0;
  if (cap < 1) return -5;
  out[0] = 0;
  while (i < n) {
    int32_t t = hits[i].taxon;
    long c = 0;
    int j = i;
    while (j < n && hits[j].taxon == t) { c += hits[j].count; j++; } /* fromHits merge */
    int r;
    if (t == ORC_MATE_PAIR_BORDER) r = snprintf(out + len, cap - len, "|:|");
    else if (t == ORC_AMBIGUOUS_SPAN) r = snprintf(out + len, cap - len, "A:%ld", c);
    else r = snprintf(out + len, cap - len, "%d:%ld", t, c);
    if (r < 0 || r >= cap - len) return -5;
    len += r;
    if (j < n) {
      if (len + 1 >= cap) return -5;
      out[len++] = ' ';
      out[len] = 0;
    }
    i = j;
  }
  return len;
}

int orc_classify_batch(const orc_params *p, const orc_index *ix, const int32_t *parents, int32_t T,
                       const uint8_t *bases, const uint64_t *offsets, const uint8_t *mate_bases,
                       const uint64_t *mate_offsets, size_t R, int min_hit_groups, const double *thresholds, int C,
                       int32_t *out_taxon, uint8_t *out_classified, int32_t *out_num_distinct,
                       int32_t *out_total_kmers, int32_t *out_num_hits) {
  int nthreads = 1;
  int err = 0;
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
#ifdef _OPENMP
#pragma omp single
    nthreads = omp_get_num_threads();
#endif
    cls_scratch c = {0};
    orc_read_result res[16];
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1024)
#endif
    for (long long r = 0; r < (long long)R; r++) {
      const char *s1 = (const char *)bases + offsets[r];
      int n1 = (int)(offsets[r + 1] - offsets[r]);
      const char *s2 = mate_bases ? (const char *)mate_bases + mate_offsets[r] : NULL;
      int n2 = mate_bases ? (int)(mate_offsets[r + 1] - mate_offsets[r]) : 0;
      int ns = classify_read_scratch(p, ix, parents, T, s1, n1, s2, n2, min_hit_groups, thresholds, C > 16 ? 16 : C,
                                     &c, res);
      if (ns < 0) { err = ns; continue; }
      for (int t = 0; t < C && t < 16; t++) {
        out_taxon[(size_t)t * R + r] = res[t].taxon;
        out_classified[(size_t)t * R + r] = (uint8_t)res[t].classified;
      }
      if (out_num_distinct) out_num_distinct[r] = res[0].num_distinct;
      if (out_total_kmers) out_total_kmers[r] = res[0].total_kmers;
      if (out_num_hits) out_num_hits[r] = ns;
    }
    cls_free(&c);
  }
  if (C > 16) return -6;
  return err < 0 ? err : nthreads;
}

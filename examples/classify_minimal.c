/* classify_minimal.c -- the C ABI from plain C: build a table from two labelled sequences on the device, classify three reads,
 * print taxon / classified / hit list per read.   cc -Iinclude examples/classify_minimal.c -Lslacken_amd/lib -lslacken_amd */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "slacken_amd.h"

#define CHECK(x) do { if ((x) != SLK_OK) { fprintf(stderr, "%s: %s\n", #x, slk_last_error()); return 1; } } while (0)

int main(void) {
  /* taxonomy: 1 = root, 2 = a genus, 3 and 4 = its species */
  const int32_t parents[5] = {0, 0, 1, 2, 2};
  slk_params p = {35, 31, 7, 1, SLK_DEFAULT_TOGGLE_MASK, 1, 0};
  slk_table_config cfg = {4096, 4, 0.0f};
  slk_index *ix = NULL;
  CHECK(slk_index_create(&p, &cfg, 0, &ix));
  CHECK(slk_index_set_taxonomy(ix, parents, 5));

  /* two "genomes" that share their first 60 bases: those minimizers end up at the genus (LCA) */
  const char *shared = "ACGTTGCATGCCGATAGGCTTAACGGATCGATTACAGGCATCGATCGGATCGATCGTAGC";
  const char *g3 = "TAGGATCGATCGATCGGGATTTACGGCGATCTTAGGCTAGCTAGGCTTCGATATCGCGGCTATTAGCCGATTCGGA";
  const char *g4 = "CCATGGCTAGCTTAGGCGCGATATTCGGCTAGGATCCTAGGAGCTTCGAGGCTATATCGGCGGATTCGATCGTTAG";
  char seqs[512];
  uint64_t offsets[3] = {0, 0, 0};
  snprintf(seqs, sizeof seqs, "%s%s", shared, g3);
  offsets[1] = strlen(seqs);
  snprintf(seqs + offsets[1], sizeof seqs - offsets[1], "%s%s", shared, g4);
  offsets[2] = strlen(seqs);
  const int32_t taxa[2] = {3, 4};
  CHECK(slk_index_add_sequences(ix, (const uint8_t *)seqs, offsets, taxa, 2));
  CHECK(slk_index_finalize(ix));
  slk_index_info info;
  CHECK(slk_index_get_info(ix, &info));
  printf("records %llu\n", (unsigned long long)info.records);

  /* reads: the shared part (genus), species 3's own part, something unrelated */
  char reads[512];
  uint64_t roffs[4] = {0, 0, 0, 0};
  snprintf(reads, sizeof reads, "%s", shared);
  roffs[1] = strlen(reads);
  snprintf(reads + roffs[1], sizeof reads - roffs[1], "%s", g3);
  roffs[2] = strlen(reads);
  snprintf(reads + roffs[2], sizeof reads - roffs[2], "%s", "GGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGG");
  roffs[3] = strlen(reads);
  slk_stream *st = NULL;
  CHECK(slk_stream_create(ix, &st));
  const double thresholds[1] = {0.0};
  int32_t taxon[3], nd[3], tk[3];
  uint8_t cls[3];
  uint64_t hit_offs[4];
  slk_hit hits[512];
  CHECK(slk_classify_batch(ix, st, (const uint8_t *)reads, roffs, NULL, NULL, 3, 1, thresholds, 1, taxon, cls, nd, tk, hit_offs, hits, 512));
  for (int r = 0; r < 3; r++) {
    printf("read %d: taxon %d classified %d kmers %d hits", r, taxon[r], cls[r], tk[r]);
    for (uint64_t j = hit_offs[r]; j < hit_offs[r + 1]; j++) printf(" %d:%d", hits[j].taxon, hits[j].count);
    printf("\n");
  }
  slk_stream_destroy(st);
  slk_index_destroy(ix);
  return 0;
}

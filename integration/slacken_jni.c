/* slacken_jni.c -- JNI shim between the reference (Scala on Spark) and libslacken_amd.so.
 * Class: com.jnpersson.slacken.gpu.Native (integration/GpuClassifier.scala).  NOT COMPILED IN THIS REPOSITORY'S BUILD IMAGE
 * (it has no JDK, hence no jni.h); build where the reference builds:
 *   cc -O2 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude integration/slacken_jni.c \
 *      -Lslacken_amd/lib -lslacken_amd -o libslacken_jni.so
 * Rules (include/slacken_amd.h): plain pointers and sizes, int32 status + thread-local slk_last_error(), no callbacks, no
 * ownership transfer; one slk_index per executor JVM shared by all task threads, one slk_stream per task thread. */
#include <jni.h>
#include <stdint.h>

#include "slacken_amd.h"

static void throw_state(JNIEnv *e) {
  (*e)->ThrowNew(e, (*e)->FindClass(e, "java/lang/IllegalStateException"), slk_last_error());
}
#define H(x) ((slk_index *)(intptr_t)(x))
#define S(x) ((slk_stream *)(intptr_t)(x))

JNIEXPORT jint JNICALL Java_com_jnpersson_slacken_gpu_Native_deviceCount(JNIEnv *e, jclass c) { return slk_device_count(); }

/* KeyValueIndex.load: parameters of IndexParams + MinSplitter (k, m, spaces, XOR mask, canonical) */
JNIEXPORT jlong JNICALL Java_com_jnpersson_slacken_gpu_Native_indexCreate(JNIEnv *e, jclass c, jint k, jint m, jint spaces,
                                                                         jlong xorMask, jboolean canonical,
                                                                         jlong expectedRecords, jint maxTaxon, jint device) {
  slk_params p = {k, m, spaces, canonical ? 1 : 0, (uint64_t)xorMask, (m + 31) / 32, 0};
  slk_table_config cfg = {(uint64_t)expectedRecords, maxTaxon, 0.0f};
  slk_index *ix = NULL;
  if (slk_index_create(&p, &cfg, device, &ix) != SLK_OK) { throw_state(e); return 0; }
  return (jlong)(intptr_t)ix;
}

/* one Parquet bucket file = one call; ids holds n rows of idLongs words (row-major), pinned, not copied */
JNIEXPORT void JNICALL Java_com_jnpersson_slacken_gpu_Native_indexAppend(JNIEnv *e, jclass c, jlong h, jlongArray ids,
                                                                        jintArray taxa, jint n) {
  jlong *k = (*e)->GetPrimitiveArrayCritical(e, ids, NULL);
  jint *t = (*e)->GetPrimitiveArrayCritical(e, taxa, NULL);
  int32_t rc = slk_index_append(H(h), (const int64_t *)k, (const int32_t *)t, (uint64_t)n);
  (*e)->ReleasePrimitiveArrayCritical(e, taxa, t, JNI_ABORT);
  (*e)->ReleasePrimitiveArrayCritical(e, ids, k, JNI_ABORT);
  if (rc != SLK_OK) throw_state(e);
}

/* Taxonomy.parents (Taxonomy.scala:159) */
JNIEXPORT void JNICALL Java_com_jnpersson_slacken_gpu_Native_setTaxonomy(JNIEnv *e, jclass c, jlong h, jintArray parents) {
  jsize T = (*e)->GetArrayLength(e, parents);
  jint *p = (*e)->GetPrimitiveArrayCritical(e, parents, NULL);
  int32_t rc = slk_index_set_taxonomy(H(h), (const int32_t *)p, (int32_t)T);
  (*e)->ReleasePrimitiveArrayCritical(e, parents, p, JNI_ABORT);
  if (rc != SLK_OK) throw_state(e);
}

JNIEXPORT void JNICALL Java_com_jnpersson_slacken_gpu_Native_indexFinalize(JNIEnv *e, jclass c, jlong h) {
  if (slk_index_finalize(H(h)) != SLK_OK) throw_state(e);
}
JNIEXPORT void JNICALL Java_com_jnpersson_slacken_gpu_Native_indexDestroy(JNIEnv *e, jclass c, jlong h) { slk_index_destroy(H(h)); }

/* KeyValueIndex.makeRecords(library, Some(taxonSet)) for the dynamic library (Dynamic.scala:362-374): bases = direct
 * ByteBuffer of concatenated, whitespace-free sequences */
JNIEXPORT void JNICALL Java_com_jnpersson_slacken_gpu_Native_addSequences(JNIEnv *e, jclass c, jlong h, jobject bases,
                                                                         jlongArray offsets, jintArray taxa, jint n) {
  const uint8_t *b = (const uint8_t *)(*e)->GetDirectBufferAddress(e, bases);
  jlong *o = (*e)->GetPrimitiveArrayCritical(e, offsets, NULL);
  jint *t = (*e)->GetPrimitiveArrayCritical(e, taxa, NULL);
  int32_t rc = slk_index_add_sequences(H(h), b, (const uint64_t *)o, (const int32_t *)t, (uint64_t)n);
  (*e)->ReleasePrimitiveArrayCritical(e, taxa, t, JNI_ABORT);
  (*e)->ReleasePrimitiveArrayCritical(e, offsets, o, JNI_ABORT);
  if (rc != SLK_OK) throw_state(e);
}

JNIEXPORT jlong JNICALL Java_com_jnpersson_slacken_gpu_Native_streamCreate(JNIEnv *e, jclass c, jlong h) {
  slk_stream *st = NULL;
  if (slk_stream_create(H(h), &st) != SLK_OK) { throw_state(e); return 0; }
  return (jlong)(intptr_t)st;
}
JNIEXPORT void JNICALL Java_com_jnpersson_slacken_gpu_Native_streamDestroy(JNIEnv *e, jclass c, jlong s) { slk_stream_destroy(S(s)); }
/* slk_stream_set_merged_hits: the hit lists of classifyBatch / classifyBatchPacked as TaxonCounts.fromHits merges them (what
 * outputLine prints), a sixth of the bytes on the way back; not for lists that are regrouped by title afterwards */
JNIEXPORT void JNICALL Java_com_jnpersson_slacken_gpu_Native_streamSetMergedHits(JNIEnv *e, jclass c, jlong s, jboolean on) {
  if (slk_stream_set_merged_hits(S(s), on ? 1 : 0) != SLK_OK) throw_state(e);
}

/* Classifier.classify for one batch.  bases / mateBases: direct ByteBuffers (mateBases null for single-end); offsets,
 * mateOffsets: long[R+1]; thresholds: double[C]; outputs are caller-owned arrays: outTaxon int[C*R], outClassified byte[C*R],
 * outNumDistinct / outTotalKmers int[R], outHitOffsets long[R+1] (null: no hit lists), outHits direct ByteBuffer of
 * {int taxon; int count} with room for hitsCapacity entries.  A fragment r with outHitOffsets[r+1] == outHitOffsets[r]
 * produced no span: the reference emits no row for it. */
JNIEXPORT void JNICALL Java_com_jnpersson_slacken_gpu_Native_classifyBatch(
    JNIEnv *e, jclass c, jlong h, jlong s, jobject bases, jlongArray offsets, jobject mateBases, jlongArray mateOffsets, jint R,
    jint minHitGroups, jdoubleArray thresholds, jintArray outTaxon, jbyteArray outClassified, jintArray outNumDistinct,
    jintArray outTotalKmers, jlongArray outHitOffsets, jobject outHits, jlong hitsCapacity) {
  const uint8_t *b = (const uint8_t *)(*e)->GetDirectBufferAddress(e, bases);
  const uint8_t *mb = mateBases ? (const uint8_t *)(*e)->GetDirectBufferAddress(e, mateBases) : NULL;
  slk_hit *hits = outHits ? (slk_hit *)(*e)->GetDirectBufferAddress(e, outHits) : NULL;
  jsize C = (*e)->GetArrayLength(e, thresholds);
  jdouble thr[16];
  if (C > 16) C = 16;
  (*e)->GetDoubleArrayRegion(e, thresholds, 0, C, thr);
  /* the call below blocks on the GPU: no critical sections are held across it; primitive arrays are copied in and out */
  jlong *o = (*e)->GetLongArrayElements(e, offsets, NULL);
  jlong *mo = mateOffsets ? (*e)->GetLongArrayElements(e, mateOffsets, NULL) : NULL;
  jint *t = (*e)->GetIntArrayElements(e, outTaxon, NULL);
  jbyte *cl = (*e)->GetByteArrayElements(e, outClassified, NULL);
  jint *nd = (*e)->GetIntArrayElements(e, outNumDistinct, NULL);
  jint *tk = (*e)->GetIntArrayElements(e, outTotalKmers, NULL);
  jlong *ho = outHitOffsets ? (*e)->GetLongArrayElements(e, outHitOffsets, NULL) : NULL;
  int32_t rc = slk_classify_batch(H(h), S(s), b, (const uint64_t *)o, mb, (const uint64_t *)mo, (uint64_t)R, minHitGroups, thr, C,
                                  (int32_t *)t, (uint8_t *)cl, (int32_t *)nd, (int32_t *)tk, (uint64_t *)ho, hits,
                                  (uint64_t)hitsCapacity);
  if (ho) (*e)->ReleaseLongArrayElements(e, outHitOffsets, ho, 0);
  (*e)->ReleaseIntArrayElements(e, outTotalKmers, tk, 0);
  (*e)->ReleaseIntArrayElements(e, outNumDistinct, nd, 0);
  (*e)->ReleaseByteArrayElements(e, outClassified, cl, 0);
  (*e)->ReleaseIntArrayElements(e, outTaxon, t, 0);
  if (mo) (*e)->ReleaseLongArrayElements(e, mateOffsets, mo, JNI_ABORT);
  (*e)->ReleaseLongArrayElements(e, offsets, o, JNI_ABORT);
  if (rc != SLK_OK) throw_state(e);
}

/* Classifier.classify for one batch whose reads are PACKED -- 2-bit codes and validity bits, 16 bases per word, both in ONE direct
 * buffer of slk_host_alloc memory: codes (4 bytes per 16 bases) from byte 0, validity (2 bytes per 16 bases) from byte
 * validOffset -- so that 64 instead of 158 bytes per 150-base read cross the PCIe link (slk_classify_batch_packed: 643 against
 * 327 M reads/s).  Reports-only shape: no hit lists.  packBases below fills such a buffer from text; a caller that copies its reads
 * out of JVM strings anyway packs while it does (GpuClassifier.packInto). */
JNIEXPORT void JNICALL Java_com_jnpersson_slacken_gpu_Native_classifyBatchPacked(
    JNIEnv *e, jclass c, jlong h, jlong s, jobject packed, jlong validOffset, jlongArray offsets, jint R, jint minHitGroups,
    jdoubleArray thresholds, jintArray outTaxon, jbyteArray outClassified, jintArray outNumDistinct, jintArray outTotalKmers,
    jlongArray outHitOffsets) {
  (void)c;
  const uint8_t *buf = (const uint8_t *)(*e)->GetDirectBufferAddress(e, packed);
  jsize C = (*e)->GetArrayLength(e, thresholds);
  jdouble thr[16];
  if (C > 16) C = 16;
  (*e)->GetDoubleArrayRegion(e, thresholds, 0, C, thr);
  jlong *o = (*e)->GetLongArrayElements(e, offsets, NULL);
  jint *t = (*e)->GetIntArrayElements(e, outTaxon, NULL);
  jbyte *cl = (*e)->GetByteArrayElements(e, outClassified, NULL);
  jint *nd = (*e)->GetIntArrayElements(e, outNumDistinct, NULL);
  jint *tk = (*e)->GetIntArrayElements(e, outTotalKmers, NULL);
  jlong *ho = outHitOffsets ? (*e)->GetLongArrayElements(e, outHitOffsets, NULL) : NULL;   /* span counts only: a read without a span has no row */
  int32_t rc = slk_classify_batch_packed(H(h), S(s), (const uint32_t *)buf, (const uint16_t *)(buf + validOffset), (const uint64_t *)o, NULL, NULL,
                                         NULL, (uint64_t)R, minHitGroups, thr, C, (int32_t *)t, (uint8_t *)cl, (int32_t *)nd, (int32_t *)tk,
                                         (uint64_t *)ho, NULL, 0);
  if (ho) (*e)->ReleaseLongArrayElements(e, outHitOffsets, ho, 0);
  (*e)->ReleaseIntArrayElements(e, outTotalKmers, tk, 0);
  (*e)->ReleaseIntArrayElements(e, outNumDistinct, nd, 0);
  (*e)->ReleaseByteArrayElements(e, outClassified, cl, 0);
  (*e)->ReleaseIntArrayElements(e, outTaxon, t, 0);
  (*e)->ReleaseLongArrayElements(e, offsets, o, JNI_ABORT);
  if (rc != SLK_OK) throw_state(e);
}

/* slk_pack_bases: text (direct buffer) -> the packed form in another direct buffer (layout as above) */
JNIEXPORT void JNICALL Java_com_jnpersson_slacken_gpu_Native_packBases(JNIEnv *e, jclass c, jobject text, jlong n, jobject packed, jlong validOffset) {
  (void)c;
  uint8_t *buf = (uint8_t *)(*e)->GetDirectBufferAddress(e, packed);
  if (slk_pack_bases((const uint8_t *)(*e)->GetDirectBufferAddress(e, text), (uint64_t)n, (uint32_t *)buf, (uint16_t *)(buf + validOffset)) != SLK_OK)
    throw_state(e);
}

/* slk_spans_batch: OrdinalSpan per fragment (KeyValueIndex.getSpans :163-173) -- here for the `distinct` flags of fragments whose title
 * repeats (GpuClassifier.regroup).  outSpans: direct buffer of slk_span {long key; int kmers; byte flag; byte distinct; short pad} */
JNIEXPORT void JNICALL Java_com_jnpersson_slacken_gpu_Native_spansBatch(JNIEnv *e, jclass c, jlong h, jlong s, jobject bases, jlongArray offsets,
                                                                       jobject mateBases, jlongArray mateOffsets, jint R, jlongArray outSpanOffsets,
                                                                       jobject outSpans, jlong capacity) {
  (void)c;
  jlong *o = (*e)->GetLongArrayElements(e, offsets, NULL);
  jlong *mo = mateOffsets ? (*e)->GetLongArrayElements(e, mateOffsets, NULL) : NULL;
  jlong *so = (*e)->GetLongArrayElements(e, outSpanOffsets, NULL);
  int32_t rc = slk_spans_batch(H(h), S(s), (const uint8_t *)(*e)->GetDirectBufferAddress(e, bases), (const uint64_t *)o,
                               mateBases ? (const uint8_t *)(*e)->GetDirectBufferAddress(e, mateBases) : NULL, (const uint64_t *)mo, (uint64_t)R,
                               (uint64_t *)so, (slk_span *)(*e)->GetDirectBufferAddress(e, outSpans), (uint64_t)capacity);
  (*e)->ReleaseLongArrayElements(e, outSpanOffsets, so, 0);
  if (mo) (*e)->ReleaseLongArrayElements(e, mateOffsets, mo, JNI_ABORT);
  (*e)->ReleaseLongArrayElements(e, offsets, o, JNI_ABORT);
  if (rc != SLK_OK) throw_state(e);
}

/* slk_classify_hits: hit lists merged by the caller (fragments that share a title, Classifier.scala:92,136) */
JNIEXPORT void JNICALL Java_com_jnpersson_slacken_gpu_Native_classifyHits(JNIEnv *e, jclass c, jlong h, jlong s, jint r, jlongArray hitOffsets,
                                                                          jobject hits, jbyteArray distinct, jint minHitGroups,
                                                                          jdoubleArray thresholds, jintArray outTaxon, jbyteArray outClassified) {
  (void)c;
  jsize C = (*e)->GetArrayLength(e, thresholds);
  jlong *ho = (*e)->GetPrimitiveArrayCritical(e, hitOffsets, NULL);
  jbyte *di = distinct ? (*e)->GetPrimitiveArrayCritical(e, distinct, NULL) : NULL;
  jdouble *thr = (*e)->GetPrimitiveArrayCritical(e, thresholds, NULL);
  jint *tx = (*e)->GetPrimitiveArrayCritical(e, outTaxon, NULL);
  jbyte *cl = (*e)->GetPrimitiveArrayCritical(e, outClassified, NULL);
  int32_t rc = slk_classify_hits((slk_index *)(intptr_t)h, (slk_stream *)(intptr_t)s, (uint64_t)r, (const uint64_t *)ho,
                                 (const slk_hit *)(*e)->GetDirectBufferAddress(e, hits), (const uint8_t *)di, minHitGroups, thr, (int32_t)C,
                                 (int32_t *)tx, (uint8_t *)cl, NULL, NULL);
  (*e)->ReleasePrimitiveArrayCritical(e, outClassified, cl, 0);
  (*e)->ReleasePrimitiveArrayCritical(e, outTaxon, tx, 0);
  (*e)->ReleasePrimitiveArrayCritical(e, thresholds, thr, JNI_ABORT);
  if (di) (*e)->ReleasePrimitiveArrayCritical(e, distinct, di, JNI_ABORT);
  (*e)->ReleasePrimitiveArrayCritical(e, hitOffsets, ho, JNI_ABORT);
  if (rc != SLK_OK) (*e)->ThrowNew(e, (*e)->FindClass(e, "java/lang/IllegalStateException"), slk_last_error());
}

/* slk_host_alloc as a direct ByteBuffer: batch buffers the library DMAs from and to without a staging copy (reuse them; free with
 * slk_host_free(GetDirectBufferAddress(buffer)) when the executor shuts down) */
JNIEXPORT jobject JNICALL Java_com_jnpersson_slacken_gpu_Native_allocPinned(JNIEnv *e, jclass c, jlong bytes) {
  (void)c;
  void *p = NULL;
  if (slk_host_alloc((size_t)bytes, &p) != SLK_OK) {
    (*e)->ThrowNew(e, (*e)->FindClass(e, "java/lang/OutOfMemoryError"), slk_last_error());
    return NULL;
  }
  return (*e)->NewDirectByteBuffer(e, p, bytes);
}

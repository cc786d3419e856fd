/* GpuClassifier.scala -- what a Slacken maintainer adds to the reference tree (package com.jnpersson.slacken.gpu) to run
 * Classifier.classify on libslacken_amd.so.  NOT COMPILED HERE (no JVM in this repository's build image); written against the
 * reference's own types: KeyValueIndex, InputFragment, ClassifyParams, ClassifiedRead, TaxonHit, TaxonCounts
 * (src/main/scala/com/jnpersson/slacken/{KeyValueIndex,Classifier,TaxonCounts}.scala). */
package com.jnpersson.slacken.gpu

import java.nio.{ByteBuffer, ByteOrder}

import com.jnpersson.kmers.minimizer.InputFragment
import com.jnpersson.slacken._
import org.apache.spark.sql.{Dataset, SparkSession}

object Native {
  System.loadLibrary("slacken_jni")
  @native def deviceCount(): Int
  @native def indexCreate(k: Int, m: Int, spaces: Int, xorMask: Long, canonical: Boolean, expectedRecords: Long, maxTaxon: Int,
                          device: Int): Long
  @native def indexAppend(h: Long, ids: Array[Long], taxa: Array[Int], n: Int): Unit
  @native def setTaxonomy(h: Long, parents: Array[Int]): Unit
  @native def indexFinalize(h: Long): Unit
  @native def indexDestroy(h: Long): Unit
  @native def addSequences(h: Long, bases: ByteBuffer, offsets: Array[Long], taxa: Array[Int], n: Int): Unit
  @native def streamCreate(h: Long): Long
  @native def streamDestroy(s: Long): Unit
  @native def classifyBatch(h: Long, s: Long, bases: ByteBuffer, offsets: Array[Long], mateBases: ByteBuffer,
                            mateOffsets: Array[Long], r: Int, minHitGroups: Int, thresholds: Array[Double], outTaxon: Array[Int],
                            outClassified: Array[Byte], outNumDistinct: Array[Int], outTotalKmers: Array[Int],
                            outHitOffsets: Array[Long], outHits: ByteBuffer, hitsCapacity: Long): Unit
  /** slk_classify_hits: Classifier.classify (object, Classifier.scala:439-454) on hit lists the caller merged itself */
  @native def classifyHits(h: Long, s: Long, r: Int, hitOffsets: Array[Long], hits: ByteBuffer, distinct: Array[Byte],
                           minHitGroups: Int, thresholds: Array[Double], outTaxon: Array[Int], outClassified: Array[Byte]): Unit
  /** slk_host_alloc as a direct buffer (NewDirectByteBuffer): the library DMAs from and to it without a staging copy */
  @native def allocPinned(bytes: Long): ByteBuffer
}

/** One table per executor JVM: the records are loaded once (replacing the per-query Parquet scan + join of
 * Classifier.spansToGroupedHits) and stay resident in HBM; task threads share the handle. */
object GpuIndexHolder {
  private var handle = 0L
  def get(index: KeyValueIndex, splitterParams: (Int, Int, Int, Long, Boolean), records: Iterator[(Array[Long], Int)],
          recordCount: Long): Long = synchronized {
    if (handle == 0L) {
      val (k, m, spaces, xorMask, canonical) = splitterParams
      val h = Native.indexCreate(k, m, spaces, xorMask, canonical, recordCount, index.taxonomy.size - 1, 0)
      val idLongs = (m + 31) / 32
      records.grouped(1 << 20).foreach { chunk =>
        val ids = new Array[Long](chunk.size * idLongs)
        val taxa = new Array[Int](chunk.size)
        var i = 0
        for ((id, t) <- chunk) { System.arraycopy(id, 0, ids, i * idLongs, idLongs); taxa(i) = t; i += 1 }
        Native.indexAppend(h, ids, taxa, chunk.size)
      }
      Native.setTaxonomy(h, index.taxonomy.parents)
      Native.indexFinalize(h)
      handle = h
    }
    handle
  }
}

final class GpuClassifier(index: KeyValueIndex, handle: => Long)(implicit spark: SparkSession) {
  import spark.implicits._

  /** Drop-in for Classifier.classify (Classifier.scala:114-121) with per-read output.
   * Titles: the reference groups the span rows of ALL fragments by title (Classifier.scala:92), so fragments that share a title are
   * one read.  This method classifies fragments individually, which is the same thing for every title that occurs once.  For inputs
   * where titles repeat, follow it with the reference's own grouping: `rows.groupByKey(_.title)`, leave groups of one alone, and for
   * the others concatenate the fragments' hits, sort them stably by ordinal (Classifier.scala:136) and classify the merged list with
   * Native.classifyHits (the per-hit `distinct` flags come from a getSpans-style call, slk_spans_batch) -- what the stand-alone
   * host does in slacken_cli.cpp: resolve_repeated_titles.  Batch buffers should come from Native.allocPinned and be reused. */
  def classify(subjects: Dataset[InputFragment], cpar: ClassifyParams, threshold: Double): Dataset[ClassifiedRead] = {
    val k = index.params.k
    val sre = cpar.sampleRegex.map(_.r)
    subjects.mapPartitions { frags =>
      val h = handle
      val st = Native.streamCreate(h)
      frags.grouped(1 << 20).flatMap { batch =>
        val r = batch.size
        val paired = batch.head.nucleotides2.nonEmpty
        def pack(seqs: Seq[String]): (ByteBuffer, Array[Long]) = {
          val offsets = seqs.scanLeft(0L)(_ + _.length).toArray
          val buf = ByteBuffer.allocateDirect(offsets.last.toInt + 16)
          seqs.foreach(s => buf.put(s.getBytes("ISO-8859-1")))
          (buf, offsets)
        }
        // getSpans' precondition (KeyValueIndex.scala:162): no whitespace in the nucleotides
        val (bases, offsets) = pack(batch.map(_.nucleotides))
        val (mbases, moffsets) = if (paired) pack(batch.map(_.nucleotides2.get)) else (null, null)
        val cap = offsets.last + (if (paired) moffsets.last else 0L) + r + 1
        val outTaxon = new Array[Int](r); val outCls = new Array[Byte](r)
        val nd = new Array[Int](r); val tk = new Array[Int](r); val ho = new Array[Long](r + 1)
        val hits = ByteBuffer.allocateDirect((cap * 8).toInt).order(ByteOrder.nativeOrder())
        Native.classifyBatch(h, st, bases, offsets, mbases, moffsets, r, cpar.minHitGroups, Array(threshold), outTaxon, outCls,
          nd, tk, ho, hits, cap)
        batch.iterator.zipWithIndex.flatMap { case (f, i) =>
          val n = (ho(i + 1) - ho(i)).toInt
          if (n == 0) None // no span => no row (Classifier.scala:92 groups span rows)
          else {
            val th = Array.tabulate(n) { j =>
              val p = ((ho(i) + j) * 8).toInt
              TaxonHit(distinct = false, j, hits.getInt(p), hits.getInt(p + 4)) // (distinct is already folded into nd)
            }
            val tc = TaxonCounts.fromHits(th)
            val sample = sre match {
              case Some(re) => re.findFirstMatchIn(f.header).map(_.group(1)).getOrElse("other")
              case _ => "all"
            }
            Some(ClassifiedRead(sample, outCls(i) != 0, f.header, outTaxon(i), th, tc.lengthString(k), tc.pairsInOrderString))
          }
        }
      } ++ { Native.streamDestroy(st); Iterator.empty }
    }
  }
}

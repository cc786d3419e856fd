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
  /** slk_stream_set_merged_hits: hit lists come back as TaxonCounts.fromHits would merge them (for output lines only) */
  @native def streamSetMergedHits(s: Long, on: Boolean): Unit
  @native def classifyBatch(h: Long, s: Long, bases: ByteBuffer, offsets: Array[Long], mateBases: ByteBuffer,
                            mateOffsets: Array[Long], r: Int, minHitGroups: Int, thresholds: Array[Double], outTaxon: Array[Int],
                            outClassified: Array[Byte], outNumDistinct: Array[Int], outTotalKmers: Array[Int],
                            outHitOffsets: Array[Long], outHits: ByteBuffer, hitsCapacity: Long): Unit
  /** slk_classify_hits: Classifier.classify (object, Classifier.scala:439-454) on hit lists the caller merged itself */
  @native def classifyHits(h: Long, s: Long, r: Int, hitOffsets: Array[Long], hits: ByteBuffer, distinct: Array[Byte],
                           minHitGroups: Int, thresholds: Array[Double], outTaxon: Array[Int], outClassified: Array[Byte]): Unit
  /** slk_host_alloc as a direct buffer (NewDirectByteBuffer): the library DMAs from and to it without a staging copy */
  @native def allocPinned(bytes: Long): ByteBuffer
  /** slk_classify_batch_packed: the reads as 2-bit codes + validity bits (16 bases per word; codes from byte 0 of `packed`, validity
   * from byte validOffset) -- 64 instead of 158 bytes per 150-base read over PCIe.  outHitOffsets: span counts only (0 => no row). */
  @native def classifyBatchPacked(h: Long, s: Long, packed: ByteBuffer, validOffset: Long, offsets: Array[Long], r: Int,
                                  minHitGroups: Int, thresholds: Array[Double], outTaxon: Array[Int], outClassified: Array[Byte],
                                  outNumDistinct: Array[Int], outTotalKmers: Array[Int], outHitOffsets: Array[Long]): Unit
  @native def packBases(text: ByteBuffer, n: Long, packed: ByteBuffer, validOffset: Long): Unit
  /** slk_spans_batch (getSpans): outSpans holds {long key; int kmers; byte flag; byte distinct; short pad} per span */
  @native def spansBatch(h: Long, s: Long, bases: ByteBuffer, offsets: Array[Long], mateBases: ByteBuffer, mateOffsets: Array[Long],
                         r: Int, outSpanOffsets: Array[Long], outSpans: ByteBuffer, capacity: Long): Unit
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

  /** Drop-in for Classifier.classify (Classifier.scala:114-121) with per-read output: classifyFragments, then regroup for the
   * titles that occur more than once -- the reference groups the span rows of ALL fragments by title (Classifier.scala:92), so
   * fragments that share a title are ONE read there. */
  def classify(subjects: Dataset[InputFragment], cpar: ClassifyParams, threshold: Double): Dataset[ClassifiedRead] = {
    val perFragment = classifyFragments(subjects, cpar, threshold)
    val repeated = subjects.groupBy($"header").count().where($"count" > 1).select($"header".as[String])   // (a handful, or none)
    if (repeated.isEmpty) perFragment
    else {
      val rep = spark.sparkContext.broadcast(repeated.collect().toSet)
      perFragment.filter(r => !rep.value.contains(r.title)).union(
        regroup(subjects.filter(f => rep.value.contains(f.header)), cpar, threshold))
    }
  }

  /** The fragments of repeated titles as the reference treats them: all span rows of a title in one group, its hits sorted stably by
   * ordinal (Classifier.scala:92,136; equal ordinals in input order -- Spark leaves that order undefined), classified from the merged
   * list by slk_classify_hits.  The per-hit `distinct` flags (OrdinalSpan.distinct) come from slk_spans_batch, whose spans are the
   * hits' spans one to one.  What slacken_cli.cpp: resolve_repeated_titles does on the stand-alone host. */
  def regroup(fragments: Dataset[InputFragment], cpar: ClassifyParams, threshold: Double): Dataset[ClassifiedRead] = {
    val k = index.params.k
    val sre = cpar.sampleRegex.map(_.r)
    fragments.groupByKey(_.header).mapGroups { (title, group) =>
      val frags = group.toArray
      val h = handle
      val st = Native.streamCreate(h)
      try {
        val r = frags.length
        val paired = frags.head.nucleotides2.nonEmpty
        def pack(seqs: Seq[String]): (ByteBuffer, Array[Long]) = {
          val offsets = seqs.scanLeft(0L)(_ + _.length).toArray
          val buf = ByteBuffer.allocateDirect(offsets.last.toInt + 16)
          seqs.foreach(s => buf.put(s.getBytes("ISO-8859-1")))
          (buf, offsets)
        }
        val (bases, offsets) = pack(frags.map(_.nucleotides))
        val (mbases, moffsets) = if (paired) pack(frags.map(_.nucleotides2.get)) else (null, null)
        val cap = offsets.last + (if (paired) moffsets.last else 0L) + r + 1
        val ho = new Array[Long](r + 1); val so = new Array[Long](r + 1)
        val hits = ByteBuffer.allocateDirect((cap * 8).toInt).order(ByteOrder.nativeOrder())
        val spans = ByteBuffer.allocateDirect((cap * 16).toInt).order(ByteOrder.nativeOrder())
        Native.classifyBatch(h, st, bases, offsets, mbases, moffsets, r, cpar.minHitGroups, Array(threshold), new Array[Int](r),
          new Array[Byte](r), new Array[Int](r), new Array[Int](r), ho, hits, cap)
        Native.spansBatch(h, st, bases, offsets, mbases, moffsets, r, so, spans, cap)
        // (ordinal, fragment, taxon, count, distinct) of every hit of every fragment, sorted stably by ordinal
        val rows = for {i <- 0 until r; j <- 0 until (ho(i + 1) - ho(i)).toInt} yield {
          val p = ((ho(i) + j) * 8).toInt
          val q = ((so(i) + j) * 16).toInt
          (j, i, hits.getInt(p), hits.getInt(p + 4), spans.get(q + 13))
        }
        val sorted = rows.sortBy(_._1)   // (stable)
        val n = sorted.length
        val mergedHits = ByteBuffer.allocateDirect(math.max(n, 1) * 8).order(ByteOrder.nativeOrder())
        sorted.foreach { x => mergedHits.putInt(x._3); mergedHits.putInt(x._4) }
        val outTaxon = new Array[Int](1); val outCls = new Array[Byte](1)
        Native.classifyHits(h, st, 1, Array(0L, n.toLong), mergedHits, sorted.map(_._5).toArray, cpar.minHitGroups, Array(threshold), outTaxon, outCls)
        val th = sorted.zipWithIndex.map { case (x, ord) => TaxonHit(x._5 != 0, ord, x._3, x._4) }.toArray
        val tc = TaxonCounts.fromHits(th)
        val sample = sre match {
          case Some(re) => re.findFirstMatchIn(title).map(_.group(1)).getOrElse("other")
          case _ => "all"
        }
        ClassifiedRead(sample, outCls(0) != 0, title, outTaxon(0), th, tc.lengthString(k), tc.pairsInOrderString)
      } finally Native.streamDestroy(st)
    }.filter(_.hits.nonEmpty)   // (a title none of whose fragments has a span: no row)
  }

  /** Reports only (the reference's --nodetailed, Classifier.classifySimple :458-467): (title, classified, taxon) per read through the
   * PACKED entry -- the reads are packed while they are copied out of the JVM's strings (packInto), into pinned memory, and cross
   * PCIe at 3 bits per base.  Single-end; titles that repeat go through regroup as above. */
  def classifySimple(subjects: Dataset[InputFragment], cpar: ClassifyParams, threshold: Double): Dataset[(String, Boolean, Int)] =
    subjects.mapPartitions { frags =>
      val h = handle
      val st = Native.streamCreate(h)
      frags.grouped(1 << 20).flatMap { batch =>
        val r = batch.size
        val offsets = batch.scanLeft(0L)(_ + _.nucleotides.length).toArray
        val words = (offsets.last + 15) / 16
        val validOffset = (words * 4 + 63) / 64 * 64
        val packed = Native.allocPinned(validOffset + words * 2 + 64).order(ByteOrder.LITTLE_ENDIAN)   // (reuse per task in real code)
        GpuClassifier.packInto(batch.map(_.nucleotides), packed, validOffset)
        val outTaxon = new Array[Int](r); val outCls = new Array[Byte](r)
        val ho = new Array[Long](r + 1)
        Native.classifyBatchPacked(h, st, packed, validOffset, offsets, r, cpar.minHitGroups, Array(threshold), outTaxon, outCls,
          new Array[Int](r), new Array[Int](r), ho)
        batch.iterator.zipWithIndex.collect { case (f, i) if ho(i + 1) > ho(i) => (f.header, outCls(i) != 0, outTaxon(i)) }
      } ++ { Native.streamDestroy(st); Iterator.empty }
    }

  /** Classifies every fragment on its own (exact for titles that occur once).  Batch buffers should come from Native.allocPinned
   * and be reused. */
  def classifyFragments(subjects: Dataset[InputFragment], cpar: ClassifyParams, threshold: Double): Dataset[ClassifiedRead] = {
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

object GpuClassifier {
  /** BitRepresentation.charToTwobit / isValid (BitRepresentation.scala:127-143) for a batch of reads, straight into the packed form
   * slk_classify_batch_packed takes: base p of the concatenation -> bits 2 (p % 16).. of code word p / 16, bit p % 16 of validity
   * word p / 16 (little-endian words). */
  def packInto(seqs: Seq[String], packed: ByteBuffer, validOffset: Long): Unit = {
    var p = 0L; var codes = 0; var valid = 0
    def flush(): Unit = {
      val w = (p - 1) / 16
      packed.putInt((w * 4).toInt, codes); packed.putShort((validOffset + w * 2).toInt, valid.toShort)
      codes = 0; valid = 0
    }
    for (s <- seqs; ch <- s) {
      val code = ch match {
        case 'A' | 'a' => 0; case 'C' | 'c' => 1; case 'G' | 'g' => 2; case 'T' | 't' | 'U' | 'u' => 3; case _ => -1
      }
      val j = (p % 16).toInt
      if (code >= 0) { codes |= code << (2 * j); valid |= 1 << j }
      p += 1
      if (p % 16 == 0) flush()
    }
    if (p % 16 != 0) flush()
  }
}

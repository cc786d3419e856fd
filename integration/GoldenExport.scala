/* GoldenExport.scala -- a ScalaTest for the REFERENCE's tree (drop it into src/test/scala/com/jnpersson/slacken/ and run
 * `sbt "testOnly com.jnpersson.slacken.GoldenExport"`): it produces, with the reference's OWN KeyValueIndex / Classifier, the
 * per-read output lines for the two read files this repository's configs[0] tests run on, so that the engine's results can be
 * compared with the Spark path itself instead of with the build's CPU restatement of it.
 *
 * NOT COMPILED HERE (no JVM, no sbt, no network in this repository's build image) -- the same status as integration/slacken_jni.c.
 * Until someone with a JVM has run it and committed its output, tests/test_config1.py says "reference golden absent".
 *
 * Inputs (made by this repository, all data):
 *   SLK_GOLDEN_LIB   a library in Slacken's on-disk layout -- <loc>.properties, <loc>/*.parquet (id1: long, taxon: int),
 *                    <loc>_taxonomy/{nodes,names}.dmp -- written from tests/golden/config1_library.npz by
 *                    `python tools/export_config1_library.py <loc>`: the three stand-in genomes of tests/golden/make_config1.py under
 *                    the reference's hard-coded test taxonomy (TestData.taxonomy, Testing.scala:147-156), k = 35, m = 31, s = 7
 *   testData/SRR094926_10k.fasta, testData/ERR599052_10k.fastq   the reference's own read files
 * Output:
 *   SLK_GOLDEN_OUT (default config1_reference_lines.txt.gz): for each (file, confidence) in
 *   {SRR094926_10k.fasta, ERR599052_10k.fastq} x {0.0, 0.15} a header line "# <file> c=<confidence>" followed by the
 *   ClassifiedRead.outputLine of every read (Classifier.scala:41-44), sorted by title (Spark's row order is not defined; titles
 *   are unique in both files).  Commit it as tests/golden/config1_reference_lines.txt.gz. */
package com.jnpersson.slacken

import java.io.{FileOutputStream, OutputStreamWriter, PrintWriter}
import java.util.zip.GZIPOutputStream

import com.jnpersson.kmers.{IndexParams, SparkSessionTestWrapper}
import com.jnpersson.kmers.input.FileInputs
import org.apache.spark.sql.SparkSession
import org.scalatest.funsuite.AnyFunSuite

class GoldenExport extends AnyFunSuite with SparkSessionTestWrapper {
  implicit val sp: SparkSession = spark
  import spark.sqlContext.implicits._

  test("export per-read classifications of the configs[0] read files") {
    val lib = sys.env.getOrElse("SLK_GOLDEN_LIB", "testData/slacken/config1_library")
    val out = sys.env.getOrElse("SLK_GOLDEN_OUT", "config1_reference_lines.txt.gz")
    // KeyValueIndex.load (KeyValueIndex.scala:413-426): IndexParams.read + Taxonomy.load + the Parquet records
    val index = KeyValueIndex.load(lib)
    val k = index.params.k
    val cls = new Classifier(index)
    val cpar = ClassifyParams(2, withUnclassified = true, List(0.0, 0.15), None, perReadOutput = true)
    val w = new PrintWriter(new OutputStreamWriter(new GZIPOutputStream(new FileOutputStream(out)), "UTF-8"))
    try {
      for {file <- List("testData/SRR094926_10k.fasta", "testData/ERR599052_10k.fastq")} {
        // the CLI's own input route (Slacken.scala: inputReader(files, k, paired)): FileInputs.getInputFragments(withAmbiguous = true)
        val inputs = new FileInputs(List(file), k, 10000000)
        val reads = inputs.getInputFragments(withAmbiguous = true)
        // the hits are collected once and classified at both thresholds, as classifyAndWrite does (Classifier.scala:105-112)
        val hits = cls.collectHitsBySequence(reads).cache()
        for {c <- cpar.thresholds} {
          val lines = cls.classifyHits(hits, cpar, c).map(_.outputLine).collect().sortBy(_.split("\t")(1))
          w.println(s"# ${file.split("/").last} c=$c")
          lines.foreach(l => w.println(l))
        }
        hits.unpersist()
      }
    } finally w.close()
  }
}

// slacken_cli.cpp -- `slacken-amd classify`: host side of the classify path above the C ABI (include/slacken_amd.h),
// mirroring the reference's CLI surface, input handling and outputs (S/ = src/main/scala/com/jnpersson/ in /root/reference):
//   flags            S/slacken/Slacken.scala:66-100,186-196 (classify: -i, -o, --min-hits, -p, --[no]unclassified,
//                    --[no]detailed, -c, --sample-regex, files, @list)
//   index params     S/kmers/IndexParams.scala:30-47, S/kmers/SplitterFormat.scala:42-64 (<idx>.properties)
//   taxonomy         S/slacken/Taxonomy.scala:116-137 (<idx>_taxonomy/{nodes,names,merged}.dmp)
//   records          the Parquet table (id1: int64, taxon: int32), read natively (parquet_source.cpp) or, converted once by tools/parquet_to_slkrec.py, from
//                    the flat <idx>.slkrec
//   inputs           S/kmers/input/FileInputs.scala:64-85,156-221, InputReader.scala:105-131 (FASTA, FASTQ, gz, pairing)
//   per-read output  S/slacken/Classifier.scala:41-44,124-147,184-227, S/slacken/TaxonCounts.scala:94-121
//   report           S/slacken/KrakenReport.scala (taxonomy.hpp)
// Host-only subcommands (`report`, `parse`, `props`) exist so that this layer can be tested without a GPU.
#include <atomic>
#include <chrono>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <set>
#include <unordered_map>
#include <algorithm>

#include "../../include/slacken_amd.h"
#include "output.hpp"
#include "pack.hpp"
#include "parquet_source.hpp"
#include "seqio.hpp"
#include "taxonomy.hpp"

using namespace slk_host;
namespace fs = std::filesystem;

// wall-clock per sub-task, as Dynamic.Timer prints it (Dynamic.scala:46-54)
struct Timer {
  std::string task;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit Timer(std::string t) : task(std::move(t)) {}
  ~Timer() {
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << "Finish task: " << task << " [" << s << " s]" << std::endl;
  }
};

[[noreturn]] static void die(const std::string &msg) {
  std::cerr << "slacken-amd: " << msg << std::endl;
  exit(2);
}
#define SLK_CALL(x) do { if ((x) != SLK_OK) die(std::string(#x) + ": " + slk_last_error()); } while (0)

// ---- Java .properties (the subset HDFSUtil.writeProperties produces) ----
static std::map<std::string, std::string> read_properties(const std::string &path) {
  std::ifstream f(path);
  if (!f) die("cannot open " + path);
  std::map<std::string, std::string> p;
  std::string line;
  while (std::getline(f, line)) {
    line = trim(line);
    if (line.empty() || line[0] == '#' || line[0] == '!') continue;
    size_t eq = line.find_first_of("=:");
    if (eq == std::string::npos) continue;
    std::string k = trim(line.substr(0, eq)), v = trim(line.substr(eq + 1));
    std::string u;
    for (size_t i = 0; i < v.size(); i++) { if (v[i] == '\\' && i + 1 < v.size()) i++; u.push_back(v[i]); }
    p[k] = u;
  }
  return p;
}

struct IndexParams { int k, m, spaces; uint64_t xorMask; bool canonical; };
static IndexParams read_index_params(const std::string &location) {  // IndexParams.read + RandomXORFormat.read + decorate
  auto p = read_properties(location + ".properties");
  auto get = [&](const char *k, const char *def) { auto it = p.find(k); return it == p.end() ? std::string(def ? def : "") : it->second; };
  if (!p.count("k") || !p.count("m") || !p.count("version")) die("Unable to read index parameters for " + location);
  if (std::stoi(get("version", "1")) > 1) die("A newer version of this software is needed to read " + location);
  std::string splitter = get("splitter", "standard");
  if (splitter != "randomXOR") die("splitter '" + splitter + "' is not supported by this engine (randomXOR only)");
  IndexParams ip;
  ip.k = std::stoi(get("k", nullptr));
  ip.m = std::stoi(get("m", nullptr));
  ip.spaces = std::stoi(get("minimizerSpaces", "0"));
  ip.xorMask = p.count("XORmask") ? (uint64_t)std::stoll(get("XORmask", nullptr)) : SLK_DEFAULT_TOGGLE_MASK;  // signed decimal long
  ip.canonical = get("canonical", "true") == "true";
  return ip;
}

// ---- records (<idx>.slkrec written by tools/parquet_to_slkrec.py) ----
// <idx>.slkrec: "SLKREC1\0", u64 n, u32 id columns W, u32 largest taxon (0 = not recorded), int64 keys[n][W], int32 taxa[n].
// Streamed into the device table in chunks: a standard library is ~120 GB of records, which must not need as much host memory.
struct RecordFile {
  FILE *f = nullptr;
  std::string path;
  uint64_t n = 0;
  uint32_t max_taxon = 0, id_columns = 1;
  RecordFile(const std::string &location, int expect_columns) : path(location + ".slkrec") {
    f = fopen(path.c_str(), "rb");
    if (!f) die("cannot open " + path + " (this build reads Parquet " + (parquet_available() ? "natively, but " + location + "/ holds no *.parquet" : "only through tools/parquet_to_slkrec.py " + location) + ")");
    char magic[8];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "SLKREC1", 8) != 0) die(path + ": bad magic");
    if (fread(&n, 8, 1, f) != 1 || fread(&id_columns, 4, 1, f) != 1 || fread(&max_taxon, 4, 1, f) != 1) die(path + ": truncated header");
    if ((int)id_columns != expect_columns)
      die(path + ": " + std::to_string(id_columns) + " id columns, the index parameters imply " + std::to_string(expect_columns));
  }
  ~RecordFile() { if (f) fclose(f); }
  void read_at(uint64_t off, void *dst, size_t bytes) {
    if (fseeko(f, (off_t)off, SEEK_SET) != 0 || fread(dst, 1, bytes, f) != bytes) die(path + ": truncated");
  }
  static constexpr uint64_t CHUNK = 1ull << 24;
  template <class F> void for_each_chunk(bool with_keys, F fn) {  // fn(keys or null, taxa, count)
    const uint64_t W = id_columns;
    std::vector<int64_t> keys(with_keys ? std::min(n, CHUNK) * W : 0);
    std::vector<int32_t> taxa(std::min(n, CHUNK));
    for (uint64_t o = 0; o < n; o += CHUNK) {
      uint64_t c = std::min(CHUNK, n - o);
      if (with_keys) read_at(24 + o * 8 * W, keys.data(), c * 8 * W);
      read_at(24 + n * 8 * W + o * 4, taxa.data(), c * 4);
      fn(with_keys ? keys.data() : nullptr, taxa.data(), c);
    }
  }
};

static int cmd_report(int argc, char **argv) {  // report <taxonomy dir> <counts.tsv: taxon \t count>
  if (argc < 2) die("usage: report TAXONOMY_DIR COUNTS_TSV");
  Taxonomy tax = Taxonomy::load(argv[0]);
  std::ifstream f(argv[1]);
  std::vector<std::pair<Taxon, long>> counts;
  Taxon t; long c;
  while (f >> t >> c) counts.emplace_back(t, c);
  KrakenReport(tax, counts).print(std::cout);
  return 0;
}
// taxonomy <taxonomy dir> <rank> <threshold> <counts.tsv>: the dynamic library's taxon selection on its own (Dynamic.scala
// CountFilter :174-185 + Taxonomy.taxaWithDescendants :304-311): line 1 = the kept taxa, line 2 = with descendants
static int cmd_taxonomy(int argc, char **argv) {
  if (argc < 4) die("usage: taxonomy TAXONOMY_DIR RANK THRESHOLD COUNTS_TSV");
  Taxonomy tax = Taxonomy::load(argv[0]);
  int rank = rank_index(argv[1]);
  if (rank == NO_RANK) die(std::string("unknown rank ") + argv[1]);
  long threshold = std::stol(argv[2]);
  std::ifstream f(argv[3]);
  std::vector<std::pair<Taxon, long>> counts;
  Taxon t; long c;
  while (f >> t >> c) counts.emplace_back(t, c);
  KrakenReport agg(tax, counts);
  std::vector<Taxon> keep;
  for (auto &kv : agg.taxonCounts)
    if (tax.depth(kv.first) >= rank - 1 && agg.clade(kv.first) >= threshold) keep.push_back(kv.first);
  for (Taxon k : keep) std::cout << k << ' ';
  std::cout << '\n';
  auto in = tax.withDescendants(keep);
  for (Taxon i = 0; i < tax.size(); i++) if (in[i]) std::cout << i << ' ';
  std::cout << '\n';
  return 0;
}
// gunzip FILE: the file's bytes as the input layer sees them (ByteSource: zlib, libbz2 or the parallel inflate of pargz.hpp)
static int cmd_gunzip(int argc, char **argv) {
  if (argc < 1) die("usage: gunzip FILE");
  ByteSource src(argv[0]);
  std::vector<char> buf((size_t)4 << 20);
  uint64_t total = 0;
  auto t0 = std::chrono::steady_clock::now();
  const bool quiet = argc >= 2 && std::string(argv[1]) == "--count";
  while (size_t n = src.read(buf.data(), buf.size())) {
    if (!quiet && fwrite(buf.data(), 1, n, stdout) != n) die("write error");
    total += n;
  }
  if (quiet) {
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << total << " bytes, " << dt << " s, " << total / dt / 1e9 << " GB/s\n";
  }
  return 0;
}
static int cmd_parse(int argc, char **argv) {  // parse <file> [<file2>]: header \t nucleotides [\t nucleotides2]
  if (argc < 1) die("usage: parse [--count] FILE [MATE_FILE]");
  if (std::string(argv[0]) == "--count") {  // read through the batch reader only: fragments, bases, a checksum, seconds
    std::vector<std::string> files(argv + 1, argv + std::min(argc, 3));
    auto t0 = std::chrono::steady_clock::now();
    BatchPrefetcher pf(files, files.size() >= 2);
    uint64_t n = 0, nb = 0, sum = 0;
    while (auto b = pf.next()) {
      n += b->size();
      nb += b->bases.size() + b->mate_bases.size();
      for (size_t i = 0; i < b->size(); i += 97) sum = sum * 31 + std::hash<std::string_view>()(b->title(i)) + b->seq(i).size();
    }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << n << " fragments, " << nb << " bases, checksum " << sum << ", " << dt << " s\n";
    return 0;
  }
  std::vector<std::string> files(argv, argv + std::min(argc, 2));
  FragmentSource src(files, argc >= 2);
  for (;;) {
    FragmentBatchPtr bp;
    if (!src.fill(bp, 4096, (size_t)64 << 20)) break;
    const FragmentBatch &b = *bp;
    for (size_t i = 0; i < b.size(); i++) {
      std::cout << b.title(i) << '\t' << b.seq(i);
      if (b.paired) std::cout << '\t' << b.mate(i);
      std::cout << '\n';
    }
  }
  return 0;
}
// records <idx>: how many records the library holds and a checksum of them, from the Parquet files (native reader) and
// from <idx>.slkrec when present -- lets the readers be tested without a GPU
static int cmd_records(int argc, char **argv) {
  if (argc < 1) die("usage: records INDEX_LOCATION");
  std::string location = argv[0];
  const int W = (read_index_params(location).m + 31) / 32;
  auto report = [](const char *src, uint64_t n, uint64_t kx, int64_t ts, int32_t mt) {
    std::cout << src << " n=" << n << " key_xor=" << kx << " taxon_sum=" << ts << " max_taxon=" << mt << '\n';
  };
  if (parquet_available() && fs::is_directory(location)) {
    uint64_t n = 0, kx = 0; int64_t ts = 0; int32_t mt = 0;
    int64_t stat_max = -1;
    uint64_t rows = parquet_count_rows(location, W, &stat_max);
    parquet_for_each_batch(location, W, [&](const int64_t *k, const int32_t *t, uint64_t c) {
      for (uint64_t i = 0; i < c; i++) { ts += t[i]; mt = std::max(mt, t[i]); }
      for (uint64_t i = 0; i < c * W; i++) kx ^= ((uint64_t)k[i] + i % W) * 0x9E3779B97F4A7C15ull;
      n += c;
    });
    if (rows != n) die("row count of the footers differs from the rows read");
    if (stat_max >= 0 && stat_max != mt) die("column statistics disagree with the data");
    report("parquet", n, kx, ts, mt);
  }
  if (fs::exists(location + ".slkrec")) {
    RecordFile rec(location, W);
    uint64_t n = 0, kx = 0; int64_t ts = 0; int32_t mt = 0;
    rec.for_each_chunk(true, [&](const int64_t *k, const int32_t *t, uint64_t c) {
      for (uint64_t i = 0; i < c; i++) { ts += t[i]; mt = std::max(mt, t[i]); }
      for (uint64_t i = 0; i < c * W; i++) kx ^= ((uint64_t)k[i] + i % W) * 0x9E3779B97F4A7C15ull;
      n += c;
    });
    report("slkrec", n, kx, ts, mt);
  }
  return 0;
}

// repeated [-p] FILES: the read titles the classify command would regroup (titles.hpp) -- those that occur more than once among the
// fragments, and for paired input those whose header repeats inside one file of a pair -- one per line, sorted.  Host only: lets
// the detection be tested without a GPU (every fragment is taken to produce a row).
static int cmd_repeated(int argc, char **argv) {
  bool paired = false;
  std::vector<std::string> files;
  for (int i = 0; i < argc; i++) {
    if (std::string(argv[i]) == "-p") paired = true;
    else files.push_back(argv[i]);
  }
  if (files.empty() || (paired && files.size() % 2)) die("usage: repeated [-p] FILES...");
  RepeatedTitles rep;
  ConcurrentTitleSet titles;
  const size_t unit = paired ? 2 : 1;
  for (size_t u = 0; u + unit <= files.size(); u += unit) {
    FragmentSource src(std::vector<std::string>(files.begin() + u, files.begin() + u + unit), paired, &rep);
    for (;;) {
      FragmentBatchPtr bp;
      if (!src.fill(bp, 4096, (size_t)64 << 20)) break;
      std::vector<uint64_t> hs, again;
      for (size_t i = 0; i < bp->size(); i++) hs.push_back(title_hash(bp->title(i)));
      titles.insert_many(hs, again);
      rep.add(again);
    }
  }
  rep.settle_unmatched([&](uint64_t h) { return titles.contains(h); });
  const FlatHashSet<0> D = rep.to_set();
  std::set<std::string> out;
  std::string_view h, sq;
  for (size_t i = 0; i < files.size(); i++) {
    AsyncRecordStream rs(files[i]);
    while (rs.next(h, sq)) {
      if (paired) h = remove_suffix(h, i % 2 == 0 ? "/1" : "/2");
      if (D.contains(title_hash(h))) out.insert(std::string(h));
    }
  }
  for (const std::string &t : out) std::cout << t << "\n";
  return 0;
}

static int cmd_props(int argc, char **argv) {
  if (argc < 1) die("usage: props INDEX_LOCATION");
  IndexParams ip = read_index_params(argv[0]);
  std::cout << "k=" << ip.k << " m=" << ip.m << " spaces=" << ip.spaces << " xorMask=" << (long long)ip.xorMask << " canonical=" << ip.canonical << '\n';
  return 0;
}

// ---- options shared by classify and classify2 (ClassifyCommand, Slacken.scala:66-100) ----
struct ClassifyOpts {
  std::string index, output, sample_regex;
  int min_hits = 2;
  bool paired = false, with_unclassified = true, detailed = true;
  std::vector<double> thresholds;
  std::vector<std::string> files;
  std::vector<int> devices{0};  // --devices: the GPUs that share the reads (table replicated on each)
  bool shard_table = false;     // --shard-table: the table is spread over the devices instead (a library beyond one GPU's memory)
  // classify2 (Slacken.scala:199-260)
  std::string library, rank = "species";
  int min_count = -1, min_distinct = -1, reads = -1;
  double init_confidence = 0.15;
  // GoldSetOptions (Dynamic.scala:62): a user-supplied taxon set to compare the detected set with, or to build the library from
  std::string gold_set, promote_rank;
  bool classify_with_gold = false;
};

static ClassifyOpts parse_classify_opts(int argc, char **argv, bool two_step) {
  ClassifyOpts o;
  for (int i = 0; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() { if (i + 1 >= argc) die("missing value for " + a); return std::string(argv[++i]); };
    if (a == "-i" || a == "--index") o.index = next();
    else if (a == "-o" || a == "--output") o.output = next();
    else if (a == "--min-hits") o.min_hits = std::stoi(next());
    else if (a == "-p" || a == "--paired") o.paired = true;
    else if (a == "--unclassified") o.with_unclassified = true;
    else if (a == "--nounclassified") o.with_unclassified = false;
    else if (a == "--detailed") o.detailed = true;
    else if (a == "--nodetailed") o.detailed = false;
    else if (a == "-c" || a == "--confidence") { while (i + 1 < argc && (isdigit(argv[i + 1][0]) || argv[i + 1][0] == '.')) o.thresholds.push_back(std::stod(argv[++i])); }
    else if (a == "--sample-regex") o.sample_regex = next();
    else if (a == "--devices") {
      std::string v = next();
      o.devices.clear();
      if (v == "all") {
        for (int d = 0; d < slk_device_count(); d++) o.devices.push_back(d);
        if (o.devices.empty()) die("--devices all: no GPU visible");
      } else {
        size_t p0 = 0;
        while (p0 <= v.size()) {
          size_t p1 = v.find(',', p0);
          if (p1 == std::string::npos) p1 = v.size();
          if (p1 == p0 || !isdigit((unsigned char)v[p0])) die("--devices wants `all` or a comma-separated list of device numbers");
          o.devices.push_back(std::stoi(v.substr(p0, p1 - p0)));
          p0 = p1 + 1;
        }
      }
    }
    else if (a == "--shard-table") { if (two_step) die("--shard-table is for classify (the dynamic library of classify2 is small)"); o.shard_table = true; }
    else if (two_step && (a == "-l" || a == "--library")) o.library = next();
    else if (two_step && a == "--rank") o.rank = next();
    else if (two_step && (a == "-C" || a == "--min-count")) o.min_count = std::stoi(next());
    else if (two_step && (a == "-D" || a == "--min-distinct")) o.min_distinct = std::stoi(next());
    else if (two_step && (a == "-R" || a == "--reads")) o.reads = std::stoi(next());
    else if (two_step && a == "--init-confidence") o.init_confidence = std::stod(next());
    else if (two_step && (a == "-g" || a == "--gold-set")) o.gold_set = next();
    else if (two_step && a == "--classify-with-gold") o.classify_with_gold = true;
    else if (two_step && a == "--promote-gold-set") o.promote_rank = next();
    else if (two_step && (a == "--bracken-length" || a == "--index-reports"))
      die(a + " is not supported by this engine (Bracken weights and index reports are outside the classify path)");
    else if (!a.empty() && a[0] == '@') { std::ifstream lf(a.substr(1)); std::string l; while (std::getline(lf, l)) if (!trim(l).empty()) o.files.push_back(trim(l)); }
    else if (!a.empty() && a[0] == '-') die("unknown option " + a);
    else o.files.push_back(a);
  }
  if (o.index.empty() || o.output.empty() || o.files.empty() || (two_step && o.library.empty()))
    die(two_step ? "usage: classify2 -i INDEX -o OUTPUT --library DIR [--rank R] [-R N | -C N | -D N] [--init-confidence X] [classify options] FILES"
                 : "usage: classify -i INDEX -o OUTPUT [-p] [-c T...] [--min-hits N] [--sample-regex RE] FILES");
  if (two_step && o.gold_set.empty() && (o.classify_with_gold || !o.promote_rank.empty()))
    die("--classify-with-gold and --promote-gold-set qualify a gold set: give one with -g FILE");
  if (o.thresholds.empty()) o.thresholds.push_back(0.0);
  for (double t : o.thresholds) if (t < 0 || t > 1) die("confidence must be in [0, 1]");
  if (o.paired && o.files.size() % 2 != 0)
    die("For paired end mode, please supply pairs of files (even number). " + std::to_string(o.files.size()) + " files were supplied");
  if ((o.min_count >= 0) + (o.min_distinct >= 0) + (o.reads >= 0) > 1) die("--min-count, --min-distinct and --reads are mutually exclusive");
  if (o.init_confidence < 0 || o.init_confidence > 1) die("--read-confidence must be >=0 and <= 1");
  return o;
}

// ---- the device-side index as the host sees it ----
// The table is REPLICATED on every device of --devices and the reads are shared out between them batch by batch (SURVEY 8e;
// the reference's counterpart is the fan-out of the span rows over the partitions, KeyValueIndex.scala:169-172): there is
// no exchange between devices, only the host-side merge of the per-taxon counts that the report is made of.  The same
// device may be listed more than once (two tables on it): that is how the multi-device path is tested on a one-GPU box.
// --shard-table (BASELINE configs[3]): a library whose table does not fit one GPU is SPREAD over the devices
// instead -- device i keeps the records whose minimizer falls to it (slk_index_set_shard; every device is handed the whole record
// stream and drops the rest), and the batches are classified in rounds of one batch per device by slk_shardset_classify: minimizers
// travel to their owners and taxa back (RCCL, or copies when devices repeat).  The output is byte for byte that of the other mode.
struct DeviceIndex {
  std::vector<slk_index *> ixs;   // one per device of the list
  slk_index *ix = nullptr;        // = ixs[0]
  slk_stream *st = nullptr;       // a stream on ixs[0]
  std::vector<int> devices{0};
  bool sharded = false;
  std::vector<slk_shardset *> sets;   // sharded: the rounds of several host threads overlap, each on a set (streams, buffers) of its own
  ~DeviceIndex() { reset(); }
  void reset() {
    for (slk_shardset *s : sets) slk_shardset_destroy(s);
    sets.clear();
    if (st) slk_stream_destroy(st);
    for (slk_index *i : ixs) slk_index_destroy(i);
    ixs.clear(); st = nullptr; ix = nullptr;
  }
  // slk_classify_batch on this library, whichever way it is laid out (single caller: the passes after the stream of batches)
  void classify_one(const uint8_t *bases, const uint64_t *offs, const uint8_t *mb, const uint64_t *mo, uint64_t n, int min_hits,
                    const double *thr, int C, int32_t *taxon, uint8_t *cls, int32_t *nd, int32_t *tk, uint64_t *hit_offs, slk_hit *hits, uint64_t cap) {
    if (!sharded) {
      SLK_CALL(slk_classify_batch(ix, st, bases, offs, mb, mo, n, min_hits, thr, C, taxon, cls, nd, tk, hit_offs, hits, cap));
      return;
    }
    std::vector<slk_shard_batch> round(ixs.size(), slk_shard_batch{});
    round[0] = slk_shard_batch{bases, offs, mb, mo, n, taxon, cls, nd, tk, hit_offs, hits, cap};
    SLK_CALL(slk_shardset_classify(sets[0], round.data(), min_hits, thr, C));
  }
  template <class F> void on_each(F f) {  // f(index) on every replica, side by side
    if (ixs.size() == 1) { f(ixs[0]); return; }
    std::vector<std::thread> th;
    std::vector<std::string> err(ixs.size());
    for (size_t i = 0; i < ixs.size(); i++)
      th.emplace_back([&, i] { if (f(ixs[i]) != SLK_OK) err[i] = slk_last_error(); });  // (slk_last_error is per thread)
    for (auto &t : th) t.join();
    for (auto &e : err) if (!e.empty()) die(e);
  }
  void create(const IndexParams &ip, const Taxonomy &tax, uint64_t expected_records, int32_t max_taxon) {
    slk_params sp{ip.k, ip.m, ip.spaces, ip.canonical ? 1 : 0, ip.xorMask, (ip.m + 31) / 32, 0};
    // (sharded: a device's share of the records, with room for the hash's unevenness)
    const uint64_t share = sharded ? expected_records / devices.size() + expected_records / (4 * devices.size()) + 4096 : expected_records;
    slk_table_config cfg{share, max_taxon, 0.0f};
    std::vector<int32_t> parents(tax.parents.begin(), tax.parents.end());
    if (max_taxon + 1 > (int32_t)parents.size()) parents.resize(max_taxon + 1, 0);
    for (int d : devices) {
      slk_index *one = nullptr;
      SLK_CALL(slk_index_create(&sp, &cfg, d, &one));
      if (sharded) SLK_CALL(slk_index_set_shard(one, (uint32_t)ixs.size(), (uint32_t)devices.size()));
      ixs.push_back(one);
      SLK_CALL(slk_index_set_taxonomy(one, parents.data(), (int32_t)parents.size()));
    }
    ix = ixs[0];
  }
  void append(const int64_t *keys, const int32_t *taxa, uint64_t n) {
    if (ixs.size() == 1) { SLK_CALL(slk_index_append(ix, keys, taxa, n)); return; }
    on_each([&](slk_index *i) { return slk_index_append(i, keys, taxa, n); });
  }
  int32_t add_sequences(const uint8_t *bases, const uint64_t *offsets, const int32_t *taxa, uint64_t n) {
    // (every replica builds the same records: the result does not depend on insertion order)
    std::vector<int32_t> rc(ixs.size(), SLK_OK);
    std::vector<std::string> err(ixs.size());
    std::vector<std::thread> th;
    for (size_t i = 0; i < ixs.size(); i++)
      th.emplace_back([&, i] { rc[i] = slk_index_add_sequences(ixs[i], bases, offsets, taxa, n); if (rc[i]) err[i] = slk_last_error(); });
    for (auto &t : th) t.join();
    for (size_t i = 0; i < ixs.size(); i++) if (rc[i] != SLK_OK) { last_error = err[i]; return rc[i]; }
    return SLK_OK;
  }
  std::string last_error;
  void finalize() {
    for (slk_index *i : ixs) SLK_CALL(slk_index_finalize(i));
    SLK_CALL(slk_stream_create(ix, &st));
    if (sharded) {
      // Two sets (two host threads whose rounds overlap) where the exchange is copies; ONE where it is RCCL's: several
      // communicators over the same devices, driven by threads that do not agree on an order, are NCCL / RCCL's documented way
      // into a deadlock (the library serialises its grouped calls besides), and no multi-device run has measured a gain from two.
      const char *e = getenv("SLK_SHARD_SETS");
      slk_shardset *first = nullptr;
      SLK_CALL(slk_shardset_create(ixs.data(), (int32_t)ixs.size(), SLK_EXCHANGE_AUTO, &first));
      sets.push_back(first);
      const bool rccl = slk_shardset_exchange_mode(first) == SLK_EXCHANGE_RCCL;
      const size_t n_sets = std::max<size_t>(1, std::min<size_t>(4, e ? (size_t)atol(e) : (rccl ? 1 : 2)));
      for (size_t i = 1; i < n_sets; i++) {
        slk_shardset *s = nullptr;
        SLK_CALL(slk_shardset_create(ixs.data(), (int32_t)ixs.size(), SLK_EXCHANGE_AUTO, &s));
        sets.push_back(s);
      }
      std::cerr << "table sharded over " << ixs.size() << " device table(s), exchange by "
                << (slk_shardset_exchange_mode(sets[0]) == SLK_EXCHANGE_RCCL ? "RCCL" : "device-to-device copies") << std::endl;
    }
  }
};

// One pass of the hot path over all input fragments: batches are parsed ahead on a reader thread, classified here, and
// handed to f (shared ownership: output formatting keeps them alive on its own threads).
template <class F>
static void classify_stream(DeviceIndex &dev, const std::vector<std::string> &files, bool paired, int min_hits,
                            const std::vector<double> &thresholds, bool want_spans, bool want_hits, F f, RepeatedTitles *rep = nullptr,
                            const std::function<void(const FragmentBatch &)> &pre = nullptr, bool merged_hits = false) {
  // merged_hits: the hit lists come merged as TaxonCounts.fromHits merges them (slk_stream_set_merged_hits) -- for a consumer that only
  // prints them (OutputSink), a sixth of the bytes on the way back; never with spans (their lists go by position) nor on a shard set
  // Several input files (or pairs) are read side by side, each on its own threads (a gzip file on the cores' share of it, pargz.hpp), and
  // their batches are taken in turn: the order of the output is deterministic, though interleaved between files at batch
  // granularity (the reference's output order is whatever Spark's partitions give).
  const size_t unit = paired ? 2 : 1;
  const size_t nsrc = files.size() / unit;
  const char *cenv = getenv("SLK_INPUT_STREAMS");
  const size_t conc = std::max<size_t>(1, std::min<size_t>(nsrc, cenv ? (size_t)atol(cenv) : 8));
  gz_concurrent_files() = (int)(conc * unit);
  std::vector<std::unique_ptr<BatchPrefetcher>> active;
  size_t next_src = 0;
  auto open_next = [&]() -> std::unique_ptr<BatchPrefetcher> {
    if (next_src >= nsrc) return nullptr;
    std::vector<std::string> fs(files.begin() + next_src * unit, files.begin() + (next_src + 1) * unit);
    next_src++;
    return std::make_unique<BatchPrefetcher>(fs, paired, rep);
  };
  while (active.size() < conc) { auto r = open_next(); if (!r) break; active.push_back(std::move(r)); }
  size_t turn = 0;
  auto next_batch = [&]() -> FragmentBatchPtr {
    while (!active.empty()) {
      if (turn >= active.size()) turn = 0;
      auto b = active[turn]->next();
      if (b) { turn++; return b; }
      auto r = open_next();  // this file is exhausted: the next unopened one takes its place in the rotation
      if (r) active[turn] = std::move(r);
      else active.erase(active.begin() + turn);
    }
    return nullptr;
  };
  // The batches are classified by a few worker threads, each with a stream of its own (its scratch, its staging buffers, its
  // HIP stream): one worker's copies overlap another's kernels.  Batches are taken and handed to f in input order.
  const int C = (int)thresholds.size();
  const char *wenv = getenv("SLK_CLASSIFY_THREADS");
  // (SLK_CLASSIFY_THREADS: threads per device; with several devices worker i drives device i mod N, each on its own table)
  const size_t per_dev = std::max<size_t>(1, std::min<size_t>(8, wenv ? (size_t)atol(wenv) : 2));
  const size_t n_workers = per_dev * dev.ixs.size();
  std::vector<slk_stream *> streams(n_workers, nullptr);
  std::vector<slk_index *> stream_ix(n_workers, nullptr);
  for (size_t i = 0; i < n_workers; i++) stream_ix[i] = dev.ixs[i % dev.ixs.size()];
  streams[0] = dev.st;
  for (size_t i = 1; i < n_workers && !dev.sharded; i++) SLK_CALL(slk_stream_create(stream_ix[i], &streams[i]));
  static const bool no_merge = getenv("SLK_CLI_MERGED_HITS") && getenv("SLK_CLI_MERGED_HITS")[0] == '0';   // (A/B switch)
  const bool merge = merged_hits && want_hits && !want_spans && !dev.sharded && !no_merge;
  for (size_t i = 0; i < n_workers && !dev.sharded; i++) SLK_CALL(slk_stream_set_merged_hits(streams[i], merge ? 1 : 0));
  std::atomic<size_t> total{0}, n_batches{0};
  const bool timing = getenv("SLK_HOST_TIMING") != nullptr;  // where the wall clock of the workers goes, by stage
  std::mutex mu_in, mu_out, mu_stat;
  std::condition_variable cv_out;
  size_t next_ticket = 0, next_out = 0;
  double t_input = 0, t_device = 0, t_hand_over = 0;
  std::exception_ptr failure;
  std::atomic<bool> failed{false};
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  // SLK_CLI_PACKED=1: reports-only calls send the reads packed (3 bits per base).  Off by default: this host is bound by parsing, not
  // by the link -- 10 M reads from a FASTQ file, reports only: 0.83 s as text, 0.87 s packed (the packing is the workers' time;
  // profiles/r04_cli_packed_ab.txt) -- the packed entry pays where the caller's reads are packed already or the link is the limit.
  const bool packed_calls = getenv("SLK_CLI_PACKED") && getenv("SLK_CLI_PACKED")[0] == '1';
  auto work = [&](slk_index *wix, slk_stream *st) {
    std::vector<int32_t> nd, tk;
    std::vector<uint32_t> pk_codes, pk_mcodes;
    std::vector<uint16_t> pk_valid, pk_mvalid;
    double w_input = 0, w_device = 0, w_hand_over = 0;
    try {
      for (;;) {
        double t0 = now();
        FragmentBatchPtr frags;
        size_t ticket;
        if (failed) break;
        {
          std::lock_guard<std::mutex> lk(mu_in);
          frags = next_batch();
          ticket = next_ticket;
          if (frags) next_ticket++;
        }
        if (!frags) break;
        double t1 = now();
        w_input += t1 - t0;
        n_batches++;
        auto b = new_classified_batch();
        b->frags = std::move(frags);
        b->C = C;
        const FragmentBatch &fb = *b->frags;
        const size_t n = fb.size();
        total += n;
        if (pre) pre(fb);   // (on the worker's own time, not under the output's lock)
        b->taxon.resize((size_t)C * n); b->classified.resize((size_t)C * n); nd.resize(n); tk.resize(n);
        b->hit_offs.resize(n + 1);
        const size_t cap = fb.bases.size() + fb.mate_bases.size() + n + 1;
        if (want_hits) b->reserve_hits(cap);
        const uint8_t *mb = fb.paired ? fb.mate_bases.data() : nullptr;
        const uint64_t *mo = fb.paired ? fb.mate_offs.data() : nullptr;
        if (!want_hits && packed_calls) {
          // reports only: nothing but the reads crosses the link, so they cross it in the engine's 3-bit form -- packed here, on the
          // worker's own time (pack.hpp: AVX2 + BMI2), 6 bytes per 16 bases instead of 16
          pk_codes.resize((fb.bases.size() + 15) / 16 + 1); pk_valid.resize(pk_codes.size());
          slk::pack_bases(fb.bases.data(), fb.bases.size(), pk_codes.data(), pk_valid.data());
          if (fb.paired) {
            pk_mcodes.resize((fb.mate_bases.size() + 15) / 16 + 1); pk_mvalid.resize(pk_mcodes.size());
            slk::pack_bases(fb.mate_bases.data(), fb.mate_bases.size(), pk_mcodes.data(), pk_mvalid.data());
          }
          SLK_CALL(slk_classify_batch_packed(wix, st, pk_codes.data(), pk_valid.data(), fb.offs.data(), fb.paired ? pk_mcodes.data() : nullptr,
                                             fb.paired ? pk_mvalid.data() : nullptr, mo, n, min_hits, thresholds.data(), C, b->taxon.data(),
                                             b->classified.data(), nd.data(), tk.data(), b->hit_offs.data(), nullptr, cap));
        } else {
          SLK_CALL(slk_classify_batch(wix, st, fb.bases.data(), fb.offs.data(), mb, mo, n, min_hits, thresholds.data(), C,
                                      b->taxon.data(), b->classified.data(), nd.data(), tk.data(), b->hit_offs.data(), want_hits ? b->hits.get() : nullptr, cap));
        }
        if (want_spans) {
          b->span_offs.resize(n + 1);
          b->spans.resize(cap);
          SLK_CALL(slk_spans_batch(wix, st, fb.bases.data(), fb.offs.data(), mb, mo, n, b->span_offs.data(), b->spans.data(), cap));
        }
        double t2 = now();
        w_device += t2 - t1;
        {
          std::unique_lock<std::mutex> lk(mu_out);
          cv_out.wait(lk, [&] { return next_out == ticket || failure; });
          if (!failure) f(std::shared_ptr<const ClassifiedBatch>(b));
          next_out = ticket + 1;
          cv_out.notify_all();
        }
        w_hand_over += now() - t2;
      }
    } catch (...) {
      std::lock_guard<std::mutex> lk(mu_out);
      if (!failure) failure = std::current_exception();
      failed = true;
      next_out = (size_t)-1;  // (nobody waits for a ticket any more)
      cv_out.notify_all();
    }
    std::lock_guard<std::mutex> lk(mu_stat);
    t_input += w_input; t_device += w_device; t_hand_over += w_hand_over;
  };
  // --shard-table: a worker takes up to one batch per device table, classifies them as ONE round of its shard set, and hands them
  // on in input order; a second worker's round (its own set) overlaps the first one's copies and host work
  auto work_sharded = [&](slk_shardset *set, slk_stream *span_st) {
    const size_t W = dev.ixs.size();
    std::vector<int32_t> nd, tk;
    try {
      for (;;) {
        std::vector<FragmentBatchPtr> in;
        size_t ticket0;
        if (failed) break;
        {
          std::lock_guard<std::mutex> lk(mu_in);
          ticket0 = next_ticket;
          while (in.size() < W) {
            FragmentBatchPtr fb = next_batch();
            if (!fb) break;
            in.push_back(std::move(fb));
            next_ticket++;
          }
        }
        if (in.empty()) break;
        std::vector<std::shared_ptr<ClassifiedBatch>> out;
        std::vector<slk_shard_batch> round(W, slk_shard_batch{});
        std::vector<std::vector<int32_t>> nds(in.size()), tks(in.size());
        for (size_t g = 0; g < in.size(); g++) {
          n_batches++;
          auto b = new_classified_batch();
          b->frags = std::move(in[g]);
          b->C = C;
          const FragmentBatch &fb = *b->frags;
          const size_t n = fb.size();
          total += n;
          if (pre) pre(fb);
          b->taxon.resize((size_t)C * n); b->classified.resize((size_t)C * n); nds[g].resize(n); tks[g].resize(n);
          b->hit_offs.resize(n + 1);
          const size_t cap = fb.bases.size() + fb.mate_bases.size() + n + 1;
          if (want_hits) b->reserve_hits(cap);
          round[g] = slk_shard_batch{fb.bases.data(), fb.offs.data(), fb.paired ? fb.mate_bases.data() : nullptr, fb.paired ? fb.mate_offs.data() : nullptr,
                                     n, b->taxon.data(), b->classified.data(), nds[g].data(), tks[g].data(), b->hit_offs.data(),
                                     want_hits ? b->hits.get() : nullptr, cap};
          out.push_back(b);
        }
        SLK_CALL(slk_shardset_classify(set, round.data(), min_hits, thresholds.data(), C));
        for (size_t g = 0; g < out.size(); g++) {
          std::shared_ptr<ClassifiedBatch> b = out[g];
          if (want_spans) {
            const FragmentBatch &fb = *b->frags;
            const size_t n = fb.size(), cap = fb.bases.size() + fb.mate_bases.size() + n + 1;
            b->span_offs.resize(n + 1);
            b->spans.resize(cap);
            SLK_CALL(slk_spans_batch(dev.ix, span_st, fb.bases.data(), fb.offs.data(), fb.paired ? fb.mate_bases.data() : nullptr,
                                     fb.paired ? fb.mate_offs.data() : nullptr, n, b->span_offs.data(), b->spans.data(), cap));
          }
          std::unique_lock<std::mutex> lk(mu_out);
          cv_out.wait(lk, [&] { return next_out == ticket0 + g || failure; });
          if (!failure) f(std::shared_ptr<const ClassifiedBatch>(b));
          next_out = ticket0 + g + 1;
          cv_out.notify_all();
        }
      }
    } catch (...) {
      std::lock_guard<std::mutex> lk(mu_out);
      if (!failure) failure = std::current_exception();
      failed = true;
      next_out = (size_t)-1;
      cv_out.notify_all();
    }
  };
  std::vector<std::thread> workers;
  if (dev.sharded) {
    std::vector<slk_stream *> span_streams(dev.sets.size(), nullptr);
    span_streams[0] = dev.st;
    for (size_t i = 1; i < dev.sets.size(); i++) SLK_CALL(slk_stream_create(dev.ix, &span_streams[i]));
    for (size_t i = 1; i < dev.sets.size(); i++) workers.emplace_back(work_sharded, dev.sets[i], span_streams[i]);
    work_sharded(dev.sets[0], span_streams[0]);
    for (auto &t : workers) t.join();
    for (size_t i = 1; i < dev.sets.size(); i++) slk_stream_destroy(span_streams[i]);
  } else {
    for (size_t i = 1; i < n_workers; i++) workers.emplace_back(work, stream_ix[i], streams[i]);
    work(stream_ix[0], streams[0]);
    for (auto &t : workers) t.join();
  }
  for (size_t i = 1; i < n_workers; i++) slk_stream_destroy(streams[i]);
  if (!dev.sharded) (void)slk_stream_set_merged_hits(dev.st, 0);   // (the device's own stream serves other callers: un-merged lists again)
  if (failure) std::rethrow_exception(failure);
  if (timing)
    std::cerr << "host timing: " << n_batches << " batches on " << n_workers << " classify thread(s) over " << dev.ixs.size() << " device table(s); summed over them: waiting for input "
              << t_input << " s, upload+kernels+download " << t_device << " s, waiting for their turn and handing over to the output threads "
              << t_hand_over << " s" << std::endl;
  std::cerr << total << " fragments" << std::endl;
}

static size_t host_threads() {
  const char *e = getenv("SLK_HOST_THREADS");
  long v = e ? atol(e) : 0;
  if (v > 0) return (size_t)v;
  unsigned hc = std::thread::hardware_concurrency();
  return std::min<size_t>(32, std::max<unsigned>(2, hc) - 1);
}

// ---- titles that occur more than once ----
// The reference regroups the hits of ALL fragments by title (groupBy("seqTitle") + collect_list, Classifier.scala:92; the
// same in SQLClassifier :281-290) and sorts each group by ordinal (:136, a stable sort): fragments that share a title are
// ONE read -- one row, one classification of the merged hit list.  Its paired reader is an inner join on the header
// (InputReader.scala:104-119), so a header that repeats inside a file of a pair multiplies before that grouping.
// The first pass streams the input once and treats every fragment on its own -- exact for every title that occurs once.
// The titles whose hash was seen twice (OutputSink, FragmentSource) are settled here: their records are read again, joined
// as the reference joins them, classified with hit lists, merged per title, classified again from the merged list
// (slk_classify_hits), and their rows and counts of the first pass are replaced.  The order of equal ordinals in a merged
// list is not defined by the reference (collect_list after a shuffle); here it is input order.
struct RepeatFragment { std::string title, seq, mate; };
struct RepeatResult {
  std::vector<slk_hit> hits;
  std::vector<uint8_t> distinct;
  std::vector<int32_t> taxon;        // per threshold
  std::vector<uint8_t> classified;   // per threshold
};

static std::vector<RepeatResult> classify_fragments(DeviceIndex &dev, const std::vector<RepeatFragment> &frags, const std::vector<size_t> &pick,
                                                    bool paired, int min_hits, const std::vector<double> &thresholds, bool want_distinct) {
  const int C = (int)thresholds.size();
  std::vector<RepeatResult> out(pick.size());
  size_t i0 = 0;
  while (i0 < pick.size()) {
    FragmentBatch fb;
    fb.paired = paired;
    size_t i1 = i0;
    while (i1 < pick.size() && i1 - i0 < ((size_t)1 << 16) && fb.bases.size() + fb.mate_bases.size() < ((size_t)256 << 20)) {
      const RepeatFragment &f = frags[pick[i1]];
      std::string_view m(f.mate);
      fb.add(f.title, f.seq, paired ? &m : nullptr);
      i1++;
    }
    const size_t n = i1 - i0, cap = fb.bases.size() + fb.mate_bases.size() + n + 1;
    std::vector<int32_t> taxon((size_t)C * n), nd(n), tk(n);
    std::vector<uint8_t> cls((size_t)C * n);
    std::vector<uint64_t> hit_offs(n + 1), span_offs(n + 1);
    std::vector<slk_hit> hits(cap);
    std::vector<slk_span> spans(want_distinct ? cap : 0);
    const uint8_t *mb = paired ? fb.mate_bases.data() : nullptr;
    const uint64_t *mo = paired ? fb.mate_offs.data() : nullptr;
    dev.classify_one(fb.bases.data(), fb.offs.data(), mb, mo, n, min_hits, thresholds.data(), C, taxon.data(), cls.data(), nd.data(), tk.data(),
                     hit_offs.data(), hits.data(), cap);
    if (want_distinct) SLK_CALL(slk_spans_batch(dev.ix, dev.st, fb.bases.data(), fb.offs.data(), mb, mo, n, span_offs.data(), spans.data(), cap));
    for (size_t i = 0; i < n; i++) {
      RepeatResult &r = out[i0 + i];
      r.hits.assign(hits.begin() + hit_offs[i], hits.begin() + hit_offs[i + 1]);
      if (want_distinct) {
        if (span_offs[i + 1] - span_offs[i] != hit_offs[i + 1] - hit_offs[i]) die("internal: span and hit lists differ in length");
        for (size_t j = span_offs[i]; j < span_offs[i + 1]; j++) r.distinct.push_back(spans[j].distinct);
      }
      for (int c = 0; c < C; c++) { r.taxon.push_back(taxon[(size_t)c * n + i]); r.classified.push_back(cls[(size_t)c * n + i]); }
    }
    i0 = i1;
  }
  return out;
}

// What the regrouping yields: per title that occurs more than once, the merged hit list and its classification per threshold
struct Regrouped {
  std::vector<std::string> titles;
  std::vector<uint64_t> moffs{0};
  std::vector<slk_hit> mhits;
  std::vector<int32_t> mtaxon;     // [C][titles]
  std::vector<uint8_t> mcls;
};

// D: hashes of the titles seen more than once.  uncount(title, result) is called for every fragment the FIRST pass made of such a
// title (its row and its count are what the merged row replaces).
template <class Uncount>
static Regrouped regroup_repeated_titles(DeviceIndex &dev, const std::vector<std::string> &files, bool is_paired, int min_hits,
                                         const std::vector<double> &thresholds, const FlatHashSet<0> &D, Uncount uncount) {
  Regrouped out;
  const size_t unit = is_paired ? 2 : 1;
  std::vector<RepeatFragment> joined;   // the fragments of the reference's reader for these titles
  std::vector<RepeatFragment> first;    // paired: the fragments the first pass made of them (its rows are what gets replaced)
  std::string_view h, sq;
  for (size_t u = 0; u + unit <= files.size(); u += unit) {
    if (!is_paired) {
      AsyncRecordStream rs(files[u]);
      while (rs.next(h, sq)) if (D.contains(title_hash(h))) joined.push_back({std::string(h), std::string(sq), std::string()});
      continue;
    }
    // PairedInputReader.getFragments: every record of file 1 with every record of file 2 of the same header.  Three readers at
    // once -- file 1, file 2, and the pairing walk of the first pass (whose fragments are what the merged rows replace) --, each
    // with its own stream: one pass of wall time over the pair, not three one after the other.
    std::vector<std::string> order;
    std::unordered_map<std::string, std::pair<std::vector<std::string>, std::vector<std::string>>> lists;
    std::unordered_map<std::string, std::vector<std::string>> second;
    std::exception_ptr err1, err2;
    std::thread t1([&] {
      try {
        std::string_view h1, s1;
        AsyncRecordStream r1(files[u]);
        while (r1.next(h1, s1)) {
          h1 = remove_suffix(h1, "/1");
          if (!D.contains(title_hash(h1))) continue;
          auto it = lists.try_emplace(std::string(h1)).first;
          if (it->second.first.empty()) order.push_back(it->first);
          it->second.first.emplace_back(s1);
        }
      } catch (...) { err1 = std::current_exception(); }
    });
    std::thread t2([&] {
      try {
        std::string_view h2, s2;
        AsyncRecordStream r2(files[u + 1]);
        while (r2.next(h2, s2)) {
          h2 = remove_suffix(h2, "/2");
          if (D.contains(title_hash(h2))) second[std::string(h2)].emplace_back(s2);
        }
      } catch (...) { err2 = std::current_exception(); }
    });
    std::exception_ptr err0;
    try {
      FragmentSource src({files[u], files[u + 1]}, true);
      for (;;) {
        FragmentBatchPtr bp;
        if (!src.fill(bp, (size_t)1 << 17, (size_t)512 << 20)) break;
        for (size_t i = 0; i < bp->size(); i++)
          if (D.contains(title_hash(bp->title(i)))) first.push_back({std::string(bp->title(i)), std::string(bp->seq(i)), std::string(bp->mate(i))});
      }
    } catch (...) { err0 = std::current_exception(); }
    t1.join();
    t2.join();
    for (std::exception_ptr e : {err0, err1, err2}) if (e) std::rethrow_exception(e);
    for (auto &kv : second) {   // (a header of file 2 alone joins nothing)
      auto it = lists.find(kv.first);
      if (it != lists.end()) it->second.second = std::move(kv.second);
    }
    for (const std::string &title : order) {
      auto &l = lists[title];
      for (const std::string &s1 : l.first) for (const std::string &s2 : l.second) joined.push_back({title, s1, s2});
    }
  }
  // titles (compared as strings) with more than one fragment
  std::unordered_map<std::string_view, std::vector<size_t>> groups;
  std::vector<std::string_view> group_order;
  for (size_t i = 0; i < joined.size(); i++) {
    auto &g = groups[joined[i].title];
    if (g.empty()) group_order.push_back(joined[i].title);
    g.push_back(i);
  }
  std::vector<size_t> pick;
  std::vector<std::string_view> merged_titles;
  for (std::string_view title : group_order) {
    const auto &g = groups[title];
    if (g.size() < 2) continue;   // (a hash collision, or a header that repeats on one side of a pair without a partner)
    merged_titles.push_back(title);
    pick.insert(pick.end(), g.begin(), g.end());
  }
  if (merged_titles.empty()) return out;
  std::cerr << merged_titles.size() << " read titles occur more than once (" << pick.size() << " fragments): their hits are regrouped by title" << std::endl;
  const int C = (int)thresholds.size();
  std::vector<RepeatResult> res = classify_fragments(dev, joined, pick, is_paired, min_hits, thresholds, true);
  // what the first pass counted (and wrote) for these titles
  if (!is_paired) {
    for (size_t i = 0; i < pick.size(); i++) uncount(joined[pick[i]].title, res[i]);
  } else {
    std::vector<size_t> pick1;
    for (size_t i = 0; i < first.size(); i++) {
      auto it = groups.find(first[i].title);
      if (it != groups.end() && it->second.size() >= 2) pick1.push_back(i);
    }
    std::vector<RepeatResult> res1 = classify_fragments(dev, first, pick1, true, min_hits, thresholds, false);
    for (size_t i = 0; i < pick1.size(); i++) uncount(first[pick1[i]].title, res1[i]);
  }
  // merged hit lists: concatenation in input order, stable sort by ordinal (Classifier.scala:136)
  std::vector<uint8_t> mdistinct;
  {
    size_t at = 0;
    struct Ref { uint32_t ordinal; uint32_t member; };
    std::vector<Ref> refs;
    for (std::string_view title : merged_titles) {
      const size_t gn = groups[title].size();
      refs.clear();
      for (size_t m = 0; m < gn; m++)
        for (size_t j = 0; j < res[at + m].hits.size(); j++) refs.push_back({(uint32_t)j, (uint32_t)m});
      std::stable_sort(refs.begin(), refs.end(), [](const Ref &a, const Ref &b) { return a.ordinal < b.ordinal; });
      for (const Ref &r : refs) {
        out.mhits.push_back(res[at + r.member].hits[r.ordinal]);
        mdistinct.push_back(res[at + r.member].distinct[r.ordinal]);
      }
      out.moffs.push_back(out.mhits.size());
      out.titles.emplace_back(title);
      at += gn;
    }
  }
  const size_t R = merged_titles.size();
  out.mtaxon.resize((size_t)C * R);
  out.mcls.resize((size_t)C * R);
  for (size_t r0 = 0; r0 < R;) {   // (bounded calls: a merged list per title, a few million hits per call)
    size_t r1 = r0 + 1;
    while (r1 < R && r1 - r0 < ((size_t)1 << 18) && out.moffs[r1 + 1] - out.moffs[r0] < ((size_t)1 << 23)) r1++;
    const size_t n = r1 - r0;
    std::vector<int32_t> tx((size_t)C * n);
    std::vector<uint8_t> cl((size_t)C * n);
    SLK_CALL(slk_classify_hits(dev.ix, dev.st, n, out.moffs.data() + r0, out.mhits.data(), mdistinct.data(), min_hits, thresholds.data(), C,
                               tx.data(), cl.data(), nullptr, nullptr));
    for (int c = 0; c < C; c++)
      for (size_t i = 0; i < n; i++) { out.mtaxon[(size_t)c * R + r0 + i] = tx[(size_t)c * n + i]; out.mcls[(size_t)c * R + r0 + i] = cl[(size_t)c * n + i]; }
    r0 = r1;
  }
  return out;
}

static void resolve_repeated_titles(DeviceIndex &dev, const IndexParams &ip, const ClassifyOpts &o, OutputSink &sink) {
  Timer t("Regroup repeated titles");
  const FlatHashSet<0> D = sink.repeated().to_set();
  const int C = (int)o.thresholds.size();
  const Regrouped g = regroup_repeated_titles(dev, o.files, o.paired, o.min_hits, o.thresholds, D, [&](const std::string &title, const RepeatResult &r) {
    if (r.hits.empty()) return;  // no span, no row
    const std::string sample = sink.sample_of(title);
    for (int c = 0; c < C; c++)
      if (r.classified[c] || o.with_unclassified) sink.adjust_count(c, sample, r.taxon[c], -1);
  });
  const size_t R = g.titles.size();
  if (R == 0) return;
  std::map<std::pair<int, std::string>, std::string> extra;
  for (size_t r = 0; r < R; r++) {
    const size_t n = g.moffs[r + 1] - g.moffs[r];
    if (n == 0) continue;   // none of the fragments had a span: no row
    const std::string sample = sink.sample_of(g.titles[r]);
    for (int c = 0; c < C; c++) {
      const bool classified = g.mcls[(size_t)c * R + r] != 0;
      if (!classified && !o.with_unclassified) continue;
      const int32_t tx = g.mtaxon[(size_t)c * R + r];
      sink.adjust_count(c, sample, tx, +1);
      if (o.detailed) OutputSink::append_output_line(extra[{c, sample}], classified, g.titles[r], tx, g.mhits.data() + g.moffs[r], n, ip.k, true);
    }
  }
  std::unordered_map<std::string_view, bool> drop;
  for (const std::string &title : g.titles) drop[title] = true;
  sink.replace_rows([&](std::string_view title) { return drop.count(title) != 0; }, extra);
}

// Classifier.classifyHitsAndWrite / writePerSampleOutput (Classifier.scala:156-227): per-read lines and Kraken reports
static void classify_and_write(DeviceIndex &dev, const IndexParams &ip, const Taxonomy &tax, const ClassifyOpts &o) {
  OutputOptions oo;
  oo.output = o.output; oo.sample_regex = o.sample_regex; oo.thresholds = o.thresholds;
  oo.with_unclassified = o.with_unclassified; oo.detailed = o.detailed; oo.k = ip.k;
  Timer t("Classify reads");
  OutputSink sink(oo, tax, host_threads());
  classify_stream(dev, o.files, o.paired, o.min_hits, o.thresholds, false, o.detailed,   // (hit lists only feed the per-read lines)
                  [&](std::shared_ptr<const ClassifiedBatch> b) { sink.submit(std::move(b)); }, &sink.repeated(), nullptr, true);
  sink.drain();
  sink.repeated().settle_unmatched([&](uint64_t h) { return sink.has_title(h); });
  if (!sink.repeated().empty()) resolve_repeated_titles(dev, ip, o, sink);
  sink.finish();
}

// KeyValueIndex.load (KeyValueIndex.scala:413-426): parameters, taxonomy and records into HBM
static void load_index(const std::string &location, IndexParams &ip, Taxonomy &tax, DeviceIndex &dev) {
  Timer t("Load index " + location);
  ip = read_index_params(location);
  tax = Taxonomy::load(location + "_taxonomy");
  // records: the flat <idx>.slkrec if it exists, else Slacken's Parquet table itself
  const int W = (ip.m + 31) / 32;   // id columns (KeyValueIndex.scala:49)
  uint64_t n_records = 0;
  if (!fs::exists(location + ".slkrec") && parquet_available() && fs::is_directory(location)) {
    int64_t mt = -1;
    const double tl0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    n_records = parquet_count_rows(location, W, &mt);
    const double tl1 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    int32_t max_taxon = std::max<int32_t>(tax.size() - 1, (int32_t)std::max<int64_t>(mt, 0));
    if (mt < 0)  // no column statistics: one pass over the taxon column
      parquet_for_each_batch(location, W, [&](const int64_t *, const int32_t *taxa, uint64_t c) { for (uint64_t i = 0; i < c; i++) max_taxon = std::max(max_taxon, taxa[i]); });
    dev.create(ip, tax, n_records, max_taxon);
    const double tl2 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (getenv("SLK_HOST_TIMING")) std::cerr << "host timing: library load: footers of the bucket files " << tl1 - tl0 << " s, device table " << tl2 - tl1 << " s\n";
    // bucket files are decoded on several threads (whole files: a bucket file of a standard library is ~60 MB) and appended
    // here in file order
    struct FileRecords { std::vector<int64_t> keys; std::vector<int32_t> taxa; };
    ThreadPool pool(host_threads());
    std::deque<std::future<FileRecords>> pending;
    const bool timing = getenv("SLK_HOST_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_wait = 0, t_append = 0;
    auto drain_one = [&]() {
      const double t0 = now();
      FileRecords fr = pending.front().get();
      pending.pop_front();
      const double t1 = now();
      dev.append(fr.keys.data(), fr.taxa.data(), fr.taxa.size());
      t_wait += t1 - t0; t_append += now() - t1;
    };
    for (auto &file : parquet_list_files(location)) {
      pending.push_back(pool.submit([file, W]() {
        FileRecords fr;
        parquet_read_file(file, W, [&](const int64_t *keys, const int32_t *taxa, uint64_t c) {
          fr.keys.insert(fr.keys.end(), keys, keys + c * W);
          fr.taxa.insert(fr.taxa.end(), taxa, taxa + c);
        });
        return fr;
      }));
      while (pending.size() >= 2 * pool.size()) drain_one();
    }
    while (!pending.empty()) drain_one();
    if (timing) std::cerr << "host timing: library load: waiting for decoded bucket files " << t_wait << " s, appending them to the table " << t_append << " s (" << pool.size() << " decoding threads)\n";
  } else {
    RecordFile rec(location, W);
    n_records = rec.n;
    int32_t max_taxon = std::max<int32_t>(tax.size() - 1, (int32_t)rec.max_taxon);
    if (rec.max_taxon == 0)  // an older file without the recorded maximum: one pass over the taxon column
      rec.for_each_chunk(false, [&](const int64_t *, const int32_t *taxa, uint64_t c) { for (uint64_t i = 0; i < c; i++) max_taxon = std::max(max_taxon, taxa[i]); });
    dev.create(ip, tax, rec.n, max_taxon);
    rec.for_each_chunk(true, [&](const int64_t *keys, const int32_t *taxa, uint64_t c) { dev.append(keys, taxa, c); });
  }
  {
    const double tf0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    dev.finalize();
    if (getenv("SLK_HOST_TIMING"))
      std::cerr << "host timing: library load: finalize " << std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - tf0 << " s\n";
  }
  std::cerr << "index: " << n_records << " records, k=" << ip.k << " m=" << ip.m << " spaces=" << ip.spaces << std::endl;
}

static int cmd_classify(int argc, char **argv) {
  ClassifyOpts o = parse_classify_opts(argc, argv, false);
  IndexParams ip;
  Taxonomy tax;
  DeviceIndex dev;
  dev.devices = o.devices;
  dev.sharded = o.shard_table;
  load_index(o.index, ip, tax, dev);
  classify_and_write(dev, ip, tax, o);
  return 0;
}

// ---- classify2: two-step classification with a dynamic library (Dynamic.scala; Slacken.scala:199-260) ----
static void find_fna(const fs::path &dir, std::vector<std::string> &out) {  // HDFSUtil.findFiles(location + "/library", ".fna")
  if (!fs::exists(dir)) die("no such directory: " + dir.string());
  for (auto &e : fs::recursive_directory_iterator(dir))
    if (e.is_regular_file() && ends_with(e.path().string(), ".fna")) out.push_back(e.path().string());
  std::sort(out.begin(), out.end());
}

// "%.2f%%".format(d * 100) (Helpers.formatPerc, S/kmers/package.scala:60) with java.util.Formatter's rounding
static std::string format_perc(double d) {
  if (std::isnan(d)) return "NaN%";
  std::string v = java_format_6_2f(d * 100);
  return v.substr(v.find_first_not_of(' ')) + "%";
}
static std::string rank_name(int rank_index_) {   // a Rank's toString: the case object's name (Taxonomy.scala:39-48)
  if (rank_index_ == NO_RANK) return "null";
  std::string t = rank_titles(rank_index_);
  if (!t.empty()) t[0] = (char)toupper((unsigned char)t[0]);
  return t;
}

// Dynamic.readGoldSet (Dynamic.scala:284-310): the taxa of the file (one per line, first CSV column; secondary ids mapped to their
// primaries by merged.dmp), those without sequence in the library replaced by -- in the reference's words "promoted to" -- their
// nearest ancestor that has, everything filtered at the reclassification rank; the promoted ones are kept down to
// --promote-gold-set RANK if that is given.  in_library: GenomeLibrary.taxonSet (:35-44), the labelled taxa with their ancestors.
// The messages are the reference's (println: standard output).
static std::vector<Taxon> read_gold_set(const Taxonomy &tax, const std::string &file, const std::string &promote_rank, int rank_depth,
                                        const std::string &rank_title, const std::vector<uint8_t> &in_library) {
  std::ifstream f(file);
  if (!f) die("cannot open the gold set " + file);
  std::set<Taxon> gold;
  std::string l;
  while (std::getline(f, l)) {
    if (!l.empty() && l.back() == '\r') l.pop_back();
    if (l.empty()) continue;   // (spark.read.csv drops empty lines)
    std::string c0 = l.substr(0, l.find(','));
    if (c0.size() >= 2 && c0.front() == '"' && c0.back() == '"') c0 = c0.substr(1, c0.size() - 2);
    size_t used = 0;
    int t = 0;
    try { t = std::stoi(c0, &used); } catch (...) { used = 0; }
    if (used != c0.size() || c0.empty()) die("gold set " + file + ": not a taxon id: " + l);   // (x.getString(0).toInt throws)
    if (t < 0 || t >= (int)tax.primary.size()) die("gold set " + file + ": taxon " + c0 + " is outside the taxonomy");
    gold.insert(tax.primary[t]);
  }
  std::cout << "Gold set contained " << gold.size() << " taxa" << std::endl;
  auto in_lib = [&](Taxon t) { return t >= 0 && t < (Taxon)in_library.size() && in_library[t]; };
  std::set<Taxon> not_found, promoted;
  for (Taxon t : gold) if (!in_lib(t)) not_found.insert(t);
  for (Taxon t : not_found)
    for (Taxon p = t; p != NONE; p = (p >= 0 && p < tax.size()) ? tax.parents[p] : NONE)   // Taxonomy.pathToRoot :204-215
      if (in_lib(p)) { promoted.insert(p); break; }
  std::cout << not_found.size() << " taxa from gold set not found in library, promoted to " << promoted.size() << " taxa." << std::endl;
  {
    std::map<int, int> by_depth;   // (depth -> taxa; Rank orders by depth)
    for (Taxon t : promoted) by_depth[tax.depth(t)]++;
    std::cout << "Promoted to levels: ArrayBuffer(";
    bool first = true;
    for (auto &kv : by_depth) {
      std::cout << (first ? "" : ", ") << "(" << rank_name(kv.first >= 0 && kv.first <= 8 ? kv.first + 1 : NO_RANK) << "," << kv.second << ")";
      first = false;
    }
    std::cout << ")" << std::endl;
  }
  std::set<Taxon> kept;
  if (!promote_rank.empty()) {
    const int pr = rank_index(promote_rank);
    if (pr == NO_RANK) die("unknown rank " + promote_rank);
    for (Taxon t : promoted) if (tax.depth(t) >= pr - 1) kept.insert(t);
    std::cout << "Keeping " << kept.size() << " taxa at rank " << rank_name(pr) << " and below from promoted set" << std::endl;
  }
  std::set<Taxon> total = gold;
  total.insert(promoted.begin(), promoted.end());
  std::set<Taxon> filtered = kept;
  for (Taxon t : total) if (tax.depth(t) >= rank_depth) filtered.insert(t);
  std::cout << "Initial adjusted gold set size " << total.size() << ", filtered at " << rank_title << " to " << filtered.size() << std::endl;
  return std::vector<Taxon>(filtered.begin(), filtered.end());
}

static int cmd_classify2(int argc, char **argv) {
  ClassifyOpts o = parse_classify_opts(argc, argv, true);
  int rank = rank_index(o.rank);  // Taxonomy.rankOrNull
  if (rank == NO_RANK) die("unknown rank " + o.rank);
  const int rank_depth = rank - 1;
  IndexParams ip;
  Taxonomy tax;
  std::vector<std::pair<Taxon, long>> counts;  // (inputs: getInputFragments(withAmbiguous = true), Dynamic.scala:323)
  int32_t max_taxon;
  // GenomeLibrary.getTaxonLabels: TSV header \t taxon
  std::vector<std::pair<std::string, Taxon>> all_labels;
  {
    std::ifstream lf(o.library + "/seqid2taxid.map");
    if (!lf) die("cannot open " + o.library + "/seqid2taxid.map");
    std::string l;
    while (std::getline(lf, l)) {
      size_t tab = l.find('\t');
      if (tab == std::string::npos) continue;
      all_labels.emplace_back(l.substr(0, tab), (Taxon)std::stoi(l.substr(tab + 1)));
    }
  }
  std::vector<Taxon> gold;          // readGoldSet's result, if a gold set was given
  const bool with_gold = !o.gold_set.empty() && o.classify_with_gold;   // makeRecords :366-369: the library is built from it, nothing is detected
  {
    DeviceIndex base;
    base.devices = o.devices;
    load_index(o.index, ip, tax, base);
    slk_index_info info;
    SLK_CALL(slk_index_get_info(base.ix, &info));
    max_taxon = info.taxonomy_size - 1;
    if (!o.gold_set.empty()) {
      // GenomeLibrary.taxonSet (:35-44): the labelled taxa and their ancestors (Taxonomy.taxaWithAncestors :306-310)
      std::vector<uint8_t> in_library((size_t)tax.size(), 0);
      for (auto &lb : all_labels)
        for (Taxon p = lb.second; p > 0 && p < tax.size() && !in_library[p]; p = tax.parents[p]) in_library[p] = 1;
      gold = read_gold_set(tax, o.gold_set, o.promote_rank, rank_depth, rank_name(rank), in_library);
    }
    // step 1: per-taxon support in the sample (Dynamic.findTaxonSet :213-243)
    std::map<Taxon, long> m;
    if (with_gold) {
      // (no detection pass)
    } else if (o.min_count >= 0 || o.min_distinct >= 0) {
      // MinimizerTotalCount / MinimizerDistinctCount: hits with a true taxon at depth >= rank (minimizersInSubjects :73-86)
      std::vector<std::pair<Taxon, int64_t>> pairs;
      classify_stream(base, o.files, o.paired, o.min_hits, {0.0}, o.min_distinct >= 0, true, [&](std::shared_ptr<const ClassifiedBatch> b) {
        for (size_t i = 0; i < b->frags->size(); i++)
          for (size_t j = b->hit_offs[i]; j < b->hit_offs[i + 1]; j++) {
            Taxon t = b->hits[j].taxon;
            if (t == SLK_TAXON_AMBIGUOUS || t == SLK_TAXON_MATE_PAIR_BORDER || tax.depth(t) < rank_depth) continue;
            if (o.min_distinct >= 0) pairs.emplace_back(t, b->spans[b->span_offs[i] + (j - b->hit_offs[i])].key);
            else m[t] += 1;
          }
      });
      if (o.min_distinct >= 0) {
        std::sort(pairs.begin(), pairs.end());
        pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
        for (auto &pr : pairs) m[pr.first] += 1;
      }
    } else {
      // ClassifiedReadCount(threshold, confidence): classified reads per taxon (classifiedReadsPerTaxon :133-141).  That count goes
      // through Classifier.classify, which regroups the hits by title (Classifier.scala:92): fragments that share a title are one
      // read here too -- the titles are tracked on the way, and those that repeat are settled as in the final classification.
      ConcurrentTitleSet seen;
      RepeatedTitles rep;
      classify_stream(base, o.files, o.paired, o.min_hits, {o.init_confidence}, false, false, [&](std::shared_ptr<const ClassifiedBatch> b) {
        for (size_t i = 0; i < b->frags->size(); i++)
          if (b->hit_offs[i + 1] > b->hit_offs[i] && b->classified[i]) m[b->taxon[i]] += 1;
      }, &rep, [&](const FragmentBatch &fb) {
        std::vector<uint64_t> hs(fb.size()), again;
        for (size_t i = 0; i < fb.size(); i++) hs[i] = title_hash(fb.title(i));
        seen.insert_many(hs, again);
        rep.add(again);
      });
      rep.settle_unmatched([&](uint64_t h) { return seen.contains(h); });
      if (!rep.empty()) {
        const std::vector<double> thr{o.init_confidence};
        const Regrouped g = regroup_repeated_titles(base, o.files, o.paired, o.min_hits, thr, rep.to_set(), [&](const std::string &, const RepeatResult &r) {
          if (!r.hits.empty() && r.classified[0]) m[r.taxon[0]] -= 1;
        });
        for (size_t r = 0; r < g.titles.size(); r++)
          if (g.moffs[r + 1] > g.moffs[r] && g.mcls[r]) m[g.mtaxon[r]] += 1;
        for (auto it = m.begin(); it != m.end();) it = it->second == 0 ? m.erase(it) : std::next(it);
      }
    }
    counts.assign(m.begin(), m.end());
  }  // the base index leaves HBM here
  const long threshold = o.min_count >= 0 ? o.min_count : o.min_distinct >= 0 ? o.min_distinct : o.reads >= 0 ? o.reads : 100;
  // CountFilter (Dynamic.scala:174-185): keys at depth >= rank whose clade total reaches the threshold
  KrakenReport agg(tax, counts);
  std::vector<Taxon> keep;
  for (auto &kv : agg.taxonCounts)
    if (tax.depth(kv.first) >= rank_depth && agg.clade(kv.first) >= threshold) keep.push_back(kv.first);
  if (with_gold) {
    keep = gold;   // Dynamic.makeRecords :366-369: taxonomy.taxaWithDescendants(goldSet); no _taxonSet.txt, nothing was detected
  } else {
    std::ofstream ts(o.output + "_taxonSet.txt");  // HDFSUtil.writeTextLines, Dynamic.scala:224-225 (BitSet order = ascending)
    for (Taxon t : keep) ts << t << "\n";
  }
  if (!o.gold_set.empty() && !with_gold) {
    // findTaxonSet :262-274: the detected set against the gold set
    std::set<Taxon> g(gold.begin(), gold.end());
    size_t tp = 0;
    for (Taxon t : keep) tp += g.count(t);
    const size_t fp = keep.size() - tp, fn = g.size() - tp;
    std::cout << "Comparing detected set with supplied gold set. True Positives: " << tp << ", False Positives: " << fp << ", False Negatives: " << fn
              << ", Precision: " << format_perc((double)tp / (double)(tp + fp)) << ", Recall: " << format_perc((double)tp / (double)g.size()) << std::endl;
  }
  std::vector<uint8_t> in_set = tax.withDescendants(keep);
  size_t n_set = 0;
  for (uint8_t b : in_set) n_set += b;
  if (with_gold) std::cerr << "Gold set: " << keep.size() << " taxa at rank " << o.rank << ", expanded with descendants to " << n_set << std::endl;
  else std::cerr << "Detected set: initial scan produced " << keep.size() << " taxa at rank " << o.rank << ", expanded with descendants to " << n_set << std::endl;

  // step 2: KeyValueIndex.makeRecords(library, Some(taxonSet)) :100-122 -- sequences whose label is in the set
  std::unordered_map<std::string, Taxon> labels;
  for (auto &lb : all_labels) {
    const Taxon t = lb.second;
    if (t >= 0 && t < tax.size() && in_set[t] && tax.isDefined(t)) labels[lb.first] = t;
  }
  std::vector<std::string> fna;
  find_fna(fs::path(o.library) / "library", fna);
  std::vector<uint8_t> bases;
  std::vector<uint64_t> offsets(1, 0);
  std::vector<int32_t> taxa;
  size_t n_titles = 0;
  for (auto &file : fna) {
    AsyncRecordStream rs(file);  // (plain .fna files are parsed on several threads)
    std::string_view h, sq;
    while (rs.next(h, sq)) {
      auto it = labels.find(std::string(h));
      if (it == labels.end()) continue;
      bases.insert(bases.end(), sq.begin(), sq.end());
      offsets.push_back(bases.size());
      taxa.push_back(it->second);
      n_titles++;
    }
  }
  std::cerr << "Construct dynamic records from: " << n_titles << " sequences, " << bases.size() << " bases" << std::endl;
  // distinct minimizers <= super-mers: about 2/(w+1) per k-mer window on random sequence, at most one per window
  const int w = ip.k - ip.m + 1;
  uint64_t expected = (uint64_t)((double)bases.size() * std::min(1.0, 2.5 / (w + 1))) + 1024;
  DeviceIndex dyn;
  dyn.devices = o.devices;
  for (int attempt = 0;; attempt++) {
    dyn.create(ip, tax, expected, max_taxon);
    int32_t rc = dyn.add_sequences(bases.data(), offsets.data(), taxa.data(), taxa.size());
    if (rc == SLK_OK) break;
    if (rc != SLK_E_CAPACITY || attempt == 1) die("slk_index_add_sequences: " + dyn.last_error);
    dyn.reset();  // low-complexity sequence: retry with one record per base
    expected = bases.size() + 1024;
  }
  dyn.finalize();
  slk_index_info info;
  SLK_CALL(slk_index_get_info(dyn.ix, &info));
  std::cerr << "dynamic index: " << info.records << " records" << std::endl;
  classify_and_write(dyn, ip, tax, o);
  return 0;
}

static const char *HELP =
    "slacken-amd -- Slacken's classify path on an MI355X (libslacken_amd.so)\n"
    "  slacken-amd [--partitions N] classify  -i INDEX -o OUTPUT [options] FILES...\n"
    "  slacken-amd [--partitions N] classify2 -i INDEX -o OUTPUT --library DIR [options] FILES...\n"
    "options of both (the reference's `classify`, Slacken.scala:66-100):\n"
    "  -i, --index LOC        library location: LOC.properties, LOC_taxonomy/{nodes,names}.dmp, LOC/*.parquet (or LOC.slkrec)\n"
    "  -o, --output PREFIX    writes PREFIX_c<threshold>/sample=<id>/part-*.txt.gz and PREFIX_c<threshold>/<id>_kreport.txt\n"
    "  -c, --confidence T...  confidence thresholds in [0, 1] (default 0.0)\n"
    "      --min-hits N       distinct minimizer hits needed to classify (default 2)\n"
    "  -p, --paired           FILES are pairs (file_1 file_2 ...), joined by read id without /1 /2\n"
    "      --sample-regex RE  group 1 of the first match in the read id names the sample (\"other\" without a match).  The dialect is\n"
    "                         std::regex's ECMAScript, not java.util.regex: no possessive quantifiers, look-behind, \\p{..} or named\n"
    "                         groups; character classes, alternation, greedy and lazy quantifiers, look-ahead and back-references\n"
    "                         behave alike.  A match in which group 1 took no part names the sample \"null\", as the reference does\n"
    "      --[no]unclassified keep (default) or drop unclassified reads\n"
    "      --[no]detailed     per-read output (default) or reports only\n"
    "      --devices LIST     GPUs that share the reads, `all` or e.g. 0,1,2,3 (default 0); the library is replicated on each\n"
    "      --shard-table      classify: spread the library over the devices instead (each holds the records whose minimizer falls to it;\n"
    "                         minimizers travel to their owners and taxa back): for a library beyond one GPU's memory\n"
    "  FILES                  FASTA / FASTQ, plain, .gz or .bz2; @list.txt names a file of file names\n"
    "options of classify2 (Slacken.scala:199-260): --library DIR (DIR/library/**/*.fna, DIR/seqid2taxid.map), --rank RANK (species),\n"
    "  -R, --reads N (100) | -C, --min-count N | -D, --min-distinct N, --init-confidence X (0.15)\n"
    "  -g, --gold-set FILE (a taxon per line: the detected set is compared with it), --classify-with-gold (the dynamic library is built\n"
    "  from the gold set instead of a detected one), --promote-gold-set RANK (gold taxa without sequence in the library: keep the\n"
    "  ancestors they are promoted to down to RANK)\n"
    "host-only helpers: report TAXONOMY_DIR COUNTS_TSV | parse FILE [MATE_FILE] | props INDEX | records INDEX | repeated [-p] FILES\n"
    "environment: SLK_HOST_THREADS (formatting/decoding threads), SLK_INPUT_STREAMS (input files read side by side, default 8),\n"
    "             SLK_PARSE_THREADS (threads parsing one plain input file, default min(8, cores/2)), SLK_GZIP_LEVEL (1..9, default zlib's),\n"
    "             SLK_GZ_THREADS (threads inflating one gzip input file, default min(16, cores / files read side by side); 0: zlib),\n"
    "             SLK_CLASSIFY_THREADS (threads classifying batches, each with its own stream, default 2),\n"
    "             SLK_HOST_TIMING (report where the wall clock of the classify loop went)\n";

int main(int argc, char **argv) {
  int i = 1;
  while (i < argc && std::string(argv[i]) == "--partitions") i += 2;  // global Spark option of the reference: accepted, unused
  if (i >= argc) die("usage: slacken-amd [--partitions N] classify|classify2|report|parse|props|records ... (--help for the options)");
  std::string cmd = argv[i++];
  if (cmd == "--help" || cmd == "-h" || cmd == "help") { std::cout << HELP; return 0; }
  if (cmd == "--version") { std::cout << slk_version() << "\n"; return 0; }
  try {
    if (cmd == "classify") return cmd_classify(argc - i, argv + i);
    if (cmd == "classify2") return cmd_classify2(argc - i, argv + i);
    if (cmd == "report") return cmd_report(argc - i, argv + i);
    if (cmd == "parse") return cmd_parse(argc - i, argv + i);
    if (cmd == "gunzip") return cmd_gunzip(argc - i, argv + i);
    if (cmd == "props") return cmd_props(argc - i, argv + i);
    if (cmd == "records") return cmd_records(argc - i, argv + i);
    if (cmd == "repeated") return cmd_repeated(argc - i, argv + i);
    if (cmd == "taxonomy") return cmd_taxonomy(argc - i, argv + i);
  } catch (const std::exception &e) {
    die(e.what());
  }
  die("unknown command " + cmd + " (this engine implements `classify` and `classify2`; the reference's other subcommands are out of scope)");
}

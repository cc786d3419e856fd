// slacken_cli.cpp -- `slacken-amd classify`: host side of the classify path above the C ABI (include/slacken_amd.h),
// mirroring the reference's CLI surface, input handling and outputs (S/ = src/main/scala/com/jnpersson/ in /root/reference):
//   flags            S/slacken/Slacken.scala:66-100,186-196 (classify: -i, -o, --min-hits, -p, --[no]unclassified,
//                    --[no]detailed, -c, --sample-regex, files, @list)
//   index params     S/kmers/IndexParams.scala:30-47, S/kmers/SplitterFormat.scala:42-64 (<idx>.properties)
//   taxonomy         S/slacken/Taxonomy.scala:116-137 (<idx>_taxonomy/{nodes,names,merged}.dmp)
//   records          the Parquet table (id1: int64, taxon: int32) converted once by tools/parquet_to_slkrec.py into
//                    <idx>.slkrec (this image has no Arrow C++ development package; pyarrow does the conversion)
//   inputs           S/kmers/input/FileInputs.scala:64-85,156-221, InputReader.scala:105-131 (FASTA, FASTQ, gz, pairing)
//   per-read output  S/slacken/Classifier.scala:41-44,124-147,184-227, S/slacken/TaxonCounts.scala:94-121
//   report           S/slacken/KrakenReport.scala (taxonomy.hpp)
// Host-only subcommands (`report`, `parse`, `props`) exist so that this layer can be tested without a GPU.
#include <dlfcn.h>
#include <zlib.h>

#include <cstring>
#include <filesystem>
#include <iostream>
#include <regex>
#include <unordered_map>

#include "../../include/slacken_amd.h"
#include "taxonomy.hpp"

using namespace slk_host;
namespace fs = std::filesystem;

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
struct Records { std::vector<int64_t> keys; std::vector<int32_t> taxa; };
static Records read_records(const std::string &location) {
  std::string path = location + ".slkrec";
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) die("cannot open " + path + " (convert the Parquet table once with tools/parquet_to_slkrec.py " + location + ")");
  char magic[8];
  uint64_t n; uint32_t idl, rsv;
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "SLKREC1", 8) != 0) die(path + ": bad magic");
  if (fread(&n, 8, 1, f) != 1 || fread(&idl, 4, 1, f) != 1 || fread(&rsv, 4, 1, f) != 1) die(path + ": truncated header");
  if (idl != 1) die(path + ": " + std::to_string(idl) + " id columns; this engine supports minimizers up to 32 nt (one column)");
  Records r;
  r.keys.resize(n); r.taxa.resize(n);
  if (n && (fread(r.keys.data(), 8, n, f) != n || fread(r.taxa.data(), 4, n, f) != n)) die(path + ": truncated");
  fclose(f);
  return r;
}

// ---- sequence input ----
struct Fragment { std::string header, nucleotides; };
static bool ends_with(const std::string &s, const char *suf);
// bzip2 through the system's libbz2 (no development header in this image: the three stable high-level entry points are
// declared here and resolved at run time)
static std::string slurp_bz2(const std::string &path) {
  void *h = dlopen("libbz2.so.1", RTLD_NOW);
  if (!h) h = dlopen("libbz2.so.1.0", RTLD_NOW);
  if (!h) die("bzip2 input needs libbz2.so.1: " + path);
  auto bzopen = (void *(*)(const char *, const char *))dlsym(h, "BZ2_bzopen");
  auto bzread = (int (*)(void *, void *, int))dlsym(h, "BZ2_bzread");
  auto bzclose = (void (*)(void *))dlsym(h, "BZ2_bzclose");
  if (!bzopen || !bzread || !bzclose) die("libbz2 lacks BZ2_bzopen/BZ2_bzread/BZ2_bzclose");
  void *b = bzopen(path.c_str(), "rb");
  if (!b) die("cannot open " + path);
  std::string out;
  std::vector<char> buf(1 << 20);
  int n;
  while ((n = bzread(b, buf.data(), (int)buf.size())) > 0) out.append(buf.data(), n);
  bzclose(b);
  return out;
}
static std::string slurp(const std::string &path) {  // gzread also reads plain files
  if (ends_with(path, ".bz2")) return slurp_bz2(path);
  gzFile g = gzopen(path.c_str(), "rb");
  if (!g) die("cannot open " + path);
  std::string out;
  std::vector<char> buf(1 << 20);
  int n;
  while ((n = gzread(g, buf.data(), (unsigned)buf.size())) > 0) out.append(buf.data(), n);
  gzclose(g);
  return out;
}
static std::string first_token(const std::string &s) { return s.substr(0, s.find(' ')); }  // headerLine.split(" ")(0)

// FastaTextInput (FileInputs.scala:156-183): records separated by '>', lines by [\n\r]+, records with < 2 lines are skipped
static std::vector<Fragment> read_fasta(const std::string &path) {
  std::string all = slurp(path);
  std::vector<Fragment> out;
  size_t pos = 0;
  while (pos <= all.size()) {
    size_t end = all.find('>', pos);
    if (end == std::string::npos) end = all.size();
    std::vector<std::string> lines;
    size_t i = pos;
    while (i < end) {
      size_t j = i;
      while (j < end && all[j] != '\n' && all[j] != '\r') j++;
      if (j > i || lines.empty()) lines.emplace_back(all, i, j - i);   // split keeps a leading empty string only
      while (j < end && (all[j] == '\n' || all[j] == '\r')) j++;
      i = j;
    }
    if (lines.size() >= 2) {
      Fragment f;
      f.header = first_token(lines[0]);
      for (size_t l = 1; l < lines.size(); l++) f.nucleotides += lines[l];
      out.push_back(std::move(f));
    }
    pos = end + 1;
  }
  return out;
}
// FastqTextInput (:188-221): every 4-line window whose 1st line starts with '@' and 3rd with '+'
static std::vector<Fragment> read_fastq(const std::string &path) {
  std::string all = slurp(path);
  std::vector<std::string> lines;  // Spark's text reader: lines end with \n, \r\n or \r
  size_t i = 0;
  while (i < all.size()) {
    size_t j = all.find_first_of("\n\r", i);
    if (j == std::string::npos) j = all.size();
    lines.emplace_back(all, i, j - i);
    i = (j + 1 < all.size() && all[j] == '\r' && all[j + 1] == '\n') ? j + 2 : j + 1;
  }
  std::vector<Fragment> out;
  for (size_t l = 0; l + 2 < lines.size(); l++) {  // the window may be cut short at the end of the file; it needs 3 lines
    if (!lines[l].empty() && lines[l][0] == '@' && !lines[l + 2].empty() && lines[l + 2][0] == '+') {
      Fragment f;
      f.header = first_token(lines[l]).substr(1);
      f.nucleotides = lines[l + 1];
      out.push_back(std::move(f));
    }
  }
  return out;
}
static std::string lower(std::string s) { for (auto &c : s) c = (char)tolower(c); return s; }
static bool ends_with(const std::string &s, const char *suf) { size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }
static std::vector<Fragment> read_file(const std::string &file) {  // FileInputs.forFile :64-85
  std::string lo = lower(file);
  if (ends_with(lo, "fq") || ends_with(lo, "fastq") || ends_with(lo, ".fq.gz") || ends_with(lo, ".fastq.gz") ||
      ends_with(lo, ".fq.bz2") || ends_with(lo, ".fastq.bz2")) return read_fastq(file);
  return read_fasta(file);  // (.fai-indexed long-sequence reading is a library-build input, not a classify input)
}
static std::string remove_suffix(const std::string &h, const char *suf) { return ends_with(h, suf) ? h.substr(0, h.size() - strlen(suf)) : h; }

// Double.toString for the values a confidence list can hold (shortest repr that round-trips; at least one decimal)
static std::string java_double_to_string(double d) {
  char b[64];
  for (int prec = 1; prec <= 17; prec++) {
    snprintf(b, sizeof b, "%.*g", prec, d);
    if (strtod(b, nullptr) == d) break;
  }
  std::string s = b;
  if (s.find('e') != std::string::npos) return s;  // (scientific notation: not reachable for sensible thresholds)
  if (s.find('.') == std::string::npos) s += ".0";
  return s;
}

// TaxonCounts.lengthString :114-121 and pairsInOrderString :94-110 over un-merged hits
static std::string length_string(const slk_hit *h, size_t n, int k) {
  long a = 0, b = 0;
  size_t border = n;
  for (size_t i = 0; i < n; i++) if (h[i].taxon == SLK_TAXON_MATE_PAIR_BORDER) { border = i; break; }
  for (size_t i = 0; i < border; i++) a += h[i].count;
  if (border == n) return std::to_string(a + (k - 1));
  for (size_t i = border + 1; i < n; i++) b += h[i].count;
  return std::to_string(a + (k - 1)) + "|" + std::to_string(b + (k - 1));
}
static std::string pairs_in_order(const slk_hit *h, size_t n) {
  std::string s;
  size_t i = 0;
  while (i < n) {
    size_t j = i;
    long c = 0;
    while (j < n && h[j].taxon == h[i].taxon) { c += h[j].count; j++; }  // TaxonCounts.fromHits merges adjacent equals
    if (h[i].taxon == SLK_TAXON_MATE_PAIR_BORDER) s += "|:|";
    else if (h[i].taxon == SLK_TAXON_AMBIGUOUS) s += "A:" + std::to_string(c);
    else s += std::to_string(h[i].taxon) + ":" + std::to_string(c);
    if (j < n) s += " ";
    i = j;
  }
  return s;
}

struct GzWriter {
  gzFile g = nullptr;
  void open(const std::string &path) { g = gzopen(path.c_str(), "wb"); if (!g) die("cannot write " + path); }
  void line(const std::string &s) { gzwrite(g, s.data(), (unsigned)s.size()); gzputc(g, '\n'); }
  ~GzWriter() { if (g) gzclose(g); }
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
static int cmd_parse(int argc, char **argv) {  // parse <file> [<file2>]: header \t nucleotides [\t nucleotides2]
  if (argc < 1) die("usage: parse FILE [MATE_FILE]");
  auto a = read_file(argv[0]);
  if (argc == 1) { for (auto &f : a) std::cout << f.header << '\t' << f.nucleotides << '\n'; return 0; }
  auto b = read_file(argv[1]);
  std::unordered_map<std::string, size_t> idx;
  for (size_t i = 0; i < b.size(); i++) idx.emplace(remove_suffix(b[i].header, "/2"), i);
  for (auto &f : a) {
    auto it = idx.find(remove_suffix(f.header, "/1"));
    if (it != idx.end()) std::cout << remove_suffix(f.header, "/1") << '\t' << f.nucleotides << '\t' << b[it->second].nucleotides << '\n';
  }
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
  // classify2 (Slacken.scala:199-260)
  std::string library, rank = "species";
  int min_count = -1, min_distinct = -1, reads = -1;
  double init_confidence = 0.15;
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
    else if (two_step && (a == "-l" || a == "--library")) o.library = next();
    else if (two_step && a == "--rank") o.rank = next();
    else if (two_step && (a == "-C" || a == "--min-count")) o.min_count = std::stoi(next());
    else if (two_step && (a == "-D" || a == "--min-distinct")) o.min_distinct = std::stoi(next());
    else if (two_step && (a == "-R" || a == "--reads")) o.reads = std::stoi(next());
    else if (two_step && a == "--init-confidence") o.init_confidence = std::stod(next());
    else if (two_step && (a == "--bracken-length" || a == "--index-reports" || a == "-g" || a == "--gold-set" || a == "--classify-with-gold" || a == "--promote-gold-set"))
      die(a + " is not supported by this engine (Bracken weights, index reports and gold sets are outside the classify path)");
    else if (!a.empty() && a[0] == '@') { std::ifstream lf(a.substr(1)); std::string l; while (std::getline(lf, l)) if (!trim(l).empty()) o.files.push_back(trim(l)); }
    else if (!a.empty() && a[0] == '-') die("unknown option " + a);
    else o.files.push_back(a);
  }
  if (o.index.empty() || o.output.empty() || o.files.empty() || (two_step && o.library.empty()))
    die(two_step ? "usage: classify2 -i INDEX -o OUTPUT --library DIR [--rank R] [-R N | -C N | -D N] [--init-confidence X] [classify options] FILES"
                 : "usage: classify -i INDEX -o OUTPUT [-p] [-c T...] [--min-hits N] [--sample-regex RE] FILES");
  if (o.thresholds.empty()) o.thresholds.push_back(0.0);
  for (double t : o.thresholds) if (t < 0 || t > 1) die("confidence must be in [0, 1]");
  if (o.paired && o.files.size() % 2 != 0)
    die("For paired end mode, please supply pairs of files (even number). " + std::to_string(o.files.size()) + " files were supplied");
  if ((o.min_count >= 0) + (o.min_distinct >= 0) + (o.reads >= 0) > 1) die("--min-count, --min-distinct and --reads are mutually exclusive");
  if (o.init_confidence < 0 || o.init_confidence > 1) die("--read-confidence must be >=0 and <= 1");
  return o;
}

// ---- the device-side index as the host sees it ----
struct DeviceIndex {
  slk_index *ix = nullptr;
  slk_stream *st = nullptr;
  ~DeviceIndex() { reset(); }
  void reset() { if (st) slk_stream_destroy(st); if (ix) slk_index_destroy(ix); st = nullptr; ix = nullptr; }
  void create(const IndexParams &ip, const Taxonomy &tax, uint64_t expected_records, int32_t max_taxon) {
    slk_params sp{ip.k, ip.m, ip.spaces, ip.canonical ? 1 : 0, ip.xorMask, (ip.m + 31) / 32, 0};
    slk_table_config cfg{expected_records, max_taxon, 0.0f};
    SLK_CALL(slk_index_create(&sp, &cfg, 0, &ix));
    std::vector<int32_t> parents(tax.parents.begin(), tax.parents.end());
    if (max_taxon + 1 > (int32_t)parents.size()) parents.resize(max_taxon + 1, 0);
    SLK_CALL(slk_index_set_taxonomy(ix, parents.data(), (int32_t)parents.size()));
  }
  void finalize() {
    SLK_CALL(slk_index_finalize(ix));
    SLK_CALL(slk_stream_create(ix, &st));
  }
};

struct Inputs { std::vector<Fragment> frags, mates; bool paired = false; };
// paired: join by header after stripping /1 and /2 (InputReader.scala:105-131)
static Inputs read_inputs(const std::vector<std::string> &files, bool paired) {
  Inputs in;
  in.paired = paired;
  if (!paired) for (auto &f : files) { auto v = read_file(f); in.frags.insert(in.frags.end(), std::make_move_iterator(v.begin()), std::make_move_iterator(v.end())); }
  else for (size_t i = 0; i < files.size(); i += 2) {
    auto a = read_file(files[i]), b = read_file(files[i + 1]);
    std::unordered_map<std::string, size_t> idx;
    for (size_t j = 0; j < b.size(); j++) idx.emplace(remove_suffix(b[j].header, "/2"), j);
    for (auto &f : a) {
      std::string h = remove_suffix(f.header, "/1");
      auto it = idx.find(h);
      if (it == idx.end()) continue;
      in.frags.push_back({h, f.nucleotides});
      in.mates.push_back({h, b[it->second].nucleotides});
    }
  }
  return in;
}

// One pass of the hot path over all fragments, in batches; f(i, taxon[C], classified[C], hits, n_hits, spans or null)
// is called for every fragment that produced at least one span (the others yield no row, Classifier.scala:92).
template <class F>
static void classify_all(DeviceIndex &dev, const Inputs &in, int min_hits, const std::vector<double> &thresholds, bool want_spans, F f) {
  const size_t R = in.frags.size(), BATCH = 1 << 20;
  const int C = (int)thresholds.size();
  std::vector<uint8_t> bases, mbases, cls;
  std::vector<uint64_t> offs, moffs, hit_offs, span_offs;
  std::vector<int32_t> taxon, nd, tk, t1(C);
  std::vector<uint8_t> c1(C);
  std::vector<slk_hit> hits;
  std::vector<slk_span> spans;
  for (size_t b0 = 0; b0 < R; b0 += BATCH) {
    size_t n = std::min(BATCH, R - b0);
    bases.clear(); mbases.clear();
    offs.assign(1, 0); moffs.assign(1, 0);
    for (size_t i = 0; i < n; i++) {
      const std::string &s = in.frags[b0 + i].nucleotides;
      bases.insert(bases.end(), s.begin(), s.end());
      offs.push_back(bases.size());
      if (in.paired) { const std::string &t = in.mates[b0 + i].nucleotides; mbases.insert(mbases.end(), t.begin(), t.end()); moffs.push_back(mbases.size()); }
    }
    taxon.resize((size_t)C * n); cls.resize((size_t)C * n); nd.resize(n); tk.resize(n);
    hit_offs.resize(n + 1);
    size_t cap = bases.size() + mbases.size() + n + 1;
    hits.resize(cap);
    const uint8_t *mb = in.paired ? mbases.data() : nullptr;
    const uint64_t *mo = in.paired ? moffs.data() : nullptr;
    SLK_CALL(slk_classify_batch(dev.ix, dev.st, bases.data(), offs.data(), mb, mo, n, min_hits, thresholds.data(), C, taxon.data(),
                                cls.data(), nd.data(), tk.data(), hit_offs.data(), hits.data(), cap));
    if (want_spans) {
      span_offs.resize(n + 1);
      spans.resize(cap);
      SLK_CALL(slk_spans_batch(dev.ix, dev.st, bases.data(), offs.data(), mb, mo, n, span_offs.data(), spans.data(), cap));
    }
    for (size_t i = 0; i < n; i++) {
      size_t h0 = hit_offs[i], h1 = hit_offs[i + 1];
      if (h1 == h0) continue;
      for (int c = 0; c < C; c++) { t1[c] = taxon[(size_t)c * n + i]; c1[c] = cls[(size_t)c * n + i]; }
      f(b0 + i, t1.data(), c1.data(), &hits[h0], h1 - h0, want_spans ? &spans[span_offs[i]] : nullptr);
    }
  }
}

// Classifier.classifyHitsAndWrite / writePerSampleOutput (Classifier.scala:156-227): per-read lines and Kraken reports
static void classify_and_write(DeviceIndex &dev, const IndexParams &ip, const Taxonomy &tax, const Inputs &in, const ClassifyOpts &o) {
  size_t max_dec = 0;  // thresholds' directory names (:189-191)
  for (double t : o.thresholds) { std::string s = java_double_to_string(t); max_dec = std::max(max_dec, s.size() - s.find('.') - 1); }
  std::regex re;
  if (!o.sample_regex.empty()) re = std::regex(o.sample_regex);
  const int C = (int)o.thresholds.size();
  struct SampleOut { std::vector<GzWriter> writers; std::vector<std::map<Taxon, long>> counts; };
  std::map<std::string, SampleOut> samples;
  std::vector<std::string> locations(C);
  for (int c = 0; c < C; c++) {
    char b[64];
    snprintf(b, sizeof b, "%.*f", (int)max_dec, o.thresholds[c]);
    locations[c] = o.output + "_c" + b;
    fs::create_directories(locations[c]);
  }
  auto sample_of = [&](const std::string &title) -> std::string {  // Classifier.scala:138-142
    if (o.sample_regex.empty()) return "all";
    std::smatch m;
    if (std::regex_search(title, m, re) && m.size() > 1) return m[1].str();
    return "other";
  };
  auto out_for = [&](const std::string &sample) -> SampleOut & {
    auto it = samples.find(sample);
    if (it != samples.end()) return it->second;
    SampleOut &so = samples[sample];
    so.writers.resize(C);
    so.counts.resize(C);
    if (o.detailed) for (int c = 0; c < C; c++) {
      std::string dir = locations[c] + "/sample=" + sample;  // Classifier.perReadOutputsLocation :415-416
      fs::create_directories(dir);
      so.writers[c].open(dir + "/part-00000.txt.gz");
    }
    return so;
  };
  classify_all(dev, in, o.min_hits, o.thresholds, false,
               [&](size_t i, const int32_t *taxon, const uint8_t *cls, const slk_hit *hits, size_t nh, const slk_span *) {
    const std::string &title = in.frags[i].header;
    SampleOut &so = out_for(sample_of(title));
    for (int c = 0; c < C; c++) {
      bool classified = cls[c] != 0;
      if (!classified && !o.with_unclassified) continue;
      so.counts[c][taxon[c]] += 1;
      if (o.detailed)  // ClassifiedRead.outputLine, Classifier.scala:41-44
        so.writers[c].line(std::string(classified ? "C" : "U") + "\t" + title + "\t" + std::to_string(taxon[c]) + "\t" +
                           length_string(hits, nh, ip.k) + "\t" + pairs_in_order(hits, nh));
    }
  });
  for (auto &kv : samples)
    for (int c = 0; c < C; c++) {
      std::vector<std::pair<Taxon, long>> counts(kv.second.counts[c].begin(), kv.second.counts[c].end());
      std::ofstream rep(locations[c] + "/" + kv.first + "_kreport.txt");  // Classifier.reportOutputLocation :419-420
      KrakenReport(tax, counts).print(rep);
    }
}

// KeyValueIndex.load (KeyValueIndex.scala:413-426): parameters, taxonomy and records into HBM
static void load_index(const std::string &location, IndexParams &ip, Taxonomy &tax, DeviceIndex &dev) {
  ip = read_index_params(location);
  tax = Taxonomy::load(location + "_taxonomy");
  Records rec = read_records(location);
  int32_t max_taxon = tax.size() - 1;
  for (int32_t t : rec.taxa) max_taxon = std::max(max_taxon, t);
  dev.create(ip, tax, rec.keys.size(), max_taxon);
  SLK_CALL(slk_index_append(dev.ix, rec.keys.data(), rec.taxa.data(), rec.keys.size()));
  dev.finalize();
  std::cerr << "index: " << rec.keys.size() << " records, k=" << ip.k << " m=" << ip.m << " spaces=" << ip.spaces << std::endl;
}

static int cmd_classify(int argc, char **argv) {
  ClassifyOpts o = parse_classify_opts(argc, argv, false);
  IndexParams ip;
  Taxonomy tax;
  DeviceIndex dev;
  load_index(o.index, ip, tax, dev);
  Inputs in = read_inputs(o.files, o.paired);
  std::cerr << in.frags.size() << " fragments" << std::endl;
  classify_and_write(dev, ip, tax, in, o);
  return 0;
}

// ---- classify2: two-step classification with a dynamic library (Dynamic.scala; Slacken.scala:199-260) ----
static void find_fna(const fs::path &dir, std::vector<std::string> &out) {  // HDFSUtil.findFiles(location + "/library", ".fna")
  if (!fs::exists(dir)) die("no such directory: " + dir.string());
  for (auto &e : fs::recursive_directory_iterator(dir))
    if (e.is_regular_file() && ends_with(e.path().string(), ".fna")) out.push_back(e.path().string());
  std::sort(out.begin(), out.end());
}

static int cmd_classify2(int argc, char **argv) {
  ClassifyOpts o = parse_classify_opts(argc, argv, true);
  int rank = rank_index(o.rank);  // Taxonomy.rankOrNull
  if (rank == NO_RANK) die("unknown rank " + o.rank);
  const int rank_depth = rank - 1;
  IndexParams ip;
  Taxonomy tax;
  std::vector<std::pair<Taxon, long>> counts;
  Inputs in = read_inputs(o.files, o.paired);  // getInputFragments(withAmbiguous = true), Dynamic.scala:323
  std::cerr << in.frags.size() << " fragments" << std::endl;
  int32_t max_taxon;
  {
    DeviceIndex base;
    load_index(o.index, ip, tax, base);
    slk_index_info info;
    SLK_CALL(slk_index_get_info(base.ix, &info));
    max_taxon = info.taxonomy_size - 1;
    // step 1: per-taxon support in the sample (Dynamic.findTaxonSet :213-243)
    std::map<Taxon, long> m;
    if (o.min_count >= 0 || o.min_distinct >= 0) {
      // MinimizerTotalCount / MinimizerDistinctCount: hits with a true taxon at depth >= rank (minimizersInSubjects :73-86)
      std::vector<std::pair<Taxon, int64_t>> pairs;
      classify_all(base, in, o.min_hits, {0.0}, o.min_distinct >= 0,
                   [&](size_t, const int32_t *, const uint8_t *, const slk_hit *hits, size_t nh, const slk_span *spans) {
        for (size_t j = 0; j < nh; j++) {
          Taxon t = hits[j].taxon;
          if (t == SLK_TAXON_AMBIGUOUS || t == SLK_TAXON_MATE_PAIR_BORDER || tax.depth(t) < rank_depth) continue;
          if (o.min_distinct >= 0) pairs.emplace_back(t, spans[j].key);
          else m[t] += 1;
        }
      });
      if (o.min_distinct >= 0) {
        std::sort(pairs.begin(), pairs.end());
        pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
        for (auto &pr : pairs) m[pr.first] += 1;
      }
    } else {
      // ClassifiedReadCount(threshold, confidence): classified reads per taxon (classifiedReadsPerTaxon :133-141)
      classify_all(base, in, o.min_hits, {o.init_confidence}, false,
                   [&](size_t, const int32_t *taxon, const uint8_t *cls, const slk_hit *, size_t, const slk_span *) {
        if (cls[0]) m[taxon[0]] += 1;
      });
    }
    counts.assign(m.begin(), m.end());
  }  // the base index leaves HBM here
  const long threshold = o.min_count >= 0 ? o.min_count : o.min_distinct >= 0 ? o.min_distinct : o.reads >= 0 ? o.reads : 100;
  // CountFilter (Dynamic.scala:174-185): keys at depth >= rank whose clade total reaches the threshold
  KrakenReport agg(tax, counts);
  std::vector<Taxon> keep;
  for (auto &kv : agg.taxonCounts)
    if (tax.depth(kv.first) >= rank_depth && agg.clade(kv.first) >= threshold) keep.push_back(kv.first);
  {
    std::ofstream ts(o.output + "_taxonSet.txt");  // HDFSUtil.writeTextLines, Dynamic.scala:224-225 (BitSet order = ascending)
    for (Taxon t : keep) ts << t << "\n";
  }
  std::vector<uint8_t> in_set = tax.withDescendants(keep);
  size_t n_set = 0;
  for (uint8_t b : in_set) n_set += b;
  std::cerr << "Detected set: initial scan produced " << keep.size() << " taxa at rank " << o.rank << ", expanded with descendants to " << n_set << std::endl;

  // step 2: KeyValueIndex.makeRecords(library, Some(taxonSet)) :100-122 -- sequences whose label is in the set
  std::unordered_map<std::string, Taxon> labels;  // GenomeLibrary.getTaxonLabels: TSV header \t taxon
  {
    std::ifstream lf(o.library + "/seqid2taxid.map");
    if (!lf) die("cannot open " + o.library + "/seqid2taxid.map");
    std::string l;
    while (std::getline(lf, l)) {
      size_t tab = l.find('\t');
      if (tab == std::string::npos) continue;
      Taxon t = (Taxon)std::stoi(l.substr(tab + 1));
      if (t >= 0 && t < tax.size() && in_set[t] && tax.isDefined(t)) labels[l.substr(0, tab)] = t;
    }
  }
  std::vector<std::string> fna;
  find_fna(fs::path(o.library) / "library", fna);
  std::vector<uint8_t> bases;
  std::vector<uint64_t> offsets(1, 0);
  std::vector<int32_t> taxa;
  size_t n_titles = 0;
  for (auto &file : fna)
    for (auto &fr : read_fasta(file)) {
      auto it = labels.find(fr.header);
      if (it == labels.end()) continue;
      bases.insert(bases.end(), fr.nucleotides.begin(), fr.nucleotides.end());
      offsets.push_back(bases.size());
      taxa.push_back(it->second);
      n_titles++;
    }
  std::cerr << "Construct dynamic records from: " << n_titles << " sequences, " << bases.size() << " bases" << std::endl;
  // distinct minimizers <= super-mers: about 2/(w+1) per k-mer window on random sequence, at most one per window
  const int w = ip.k - ip.m + 1;
  uint64_t expected = (uint64_t)((double)bases.size() * std::min(1.0, 2.5 / (w + 1))) + 1024;
  DeviceIndex dyn;
  for (int attempt = 0;; attempt++) {
    dyn.create(ip, tax, expected, max_taxon);
    int32_t rc = slk_index_add_sequences(dyn.ix, bases.data(), offsets.data(), taxa.data(), taxa.size());
    if (rc == SLK_OK) break;
    if (rc != SLK_E_CAPACITY || attempt == 1) die(std::string("slk_index_add_sequences: ") + slk_last_error());
    dyn.reset();  // low-complexity sequence: retry with one record per base
    expected = bases.size() + 1024;
  }
  dyn.finalize();
  slk_index_info info;
  SLK_CALL(slk_index_get_info(dyn.ix, &info));
  std::cerr << "dynamic index: " << info.records << " records" << std::endl;
  classify_and_write(dyn, ip, tax, in, o);
  return 0;
}

int main(int argc, char **argv) {
  int i = 1;
  while (i < argc && std::string(argv[i]) == "--partitions") i += 2;  // global Spark option of the reference: accepted, unused
  if (i >= argc) die("usage: slacken-amd [--partitions N] classify|classify2|report|parse|props ...");
  std::string cmd = argv[i++];
  try {
    if (cmd == "classify") return cmd_classify(argc - i, argv + i);
    if (cmd == "classify2") return cmd_classify2(argc - i, argv + i);
    if (cmd == "report") return cmd_report(argc - i, argv + i);
    if (cmd == "parse") return cmd_parse(argc - i, argv + i);
    if (cmd == "props") return cmd_props(argc - i, argv + i);
  } catch (const std::exception &e) {
    die(e.what());
  }
  die("unknown command " + cmd + " (this engine implements `classify` and `classify2`; the reference's other subcommands are out of scope)");
}

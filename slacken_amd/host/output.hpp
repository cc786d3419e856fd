// output.hpp -- per-read output lines, gzip part files and Kraken reports of the host layer (C++17, header only).
// Mirrors (S/ = src/main/scala/com/jnpersson/ in the reference):
//   ClassifiedRead.outputLine                         S/slacken/Classifier.scala:41-44
//   TaxonCounts.lengthString / pairsInOrderString     S/slacken/TaxonCounts.scala:94-121
//   Classifier.classifyHits (sample id)               S/slacken/Classifier.scala:124-147
//   Classifier.writePerSampleOutput                   S/slacken/Classifier.scala:184-227, locations :415-420
// Formatting and compression run on a small thread pool: a batch is cut into slices, each slice is formatted and deflated
// into one gzip MEMBER per (threshold, sample), and the members are appended to the part file in read order (a
// concatenation of gzip members is a gzip file).
#pragma once
#include <dlfcn.h>
#include <zlib.h>

#include <filesystem>
#include <functional>
#include <future>
#include <map>
#include <queue>
#include <regex>

#include "../../include/slacken_amd.h"
#include "seqio.hpp"
#include "taxonomy.hpp"
#include "titles.hpp"

namespace slk_host {

// Double.toString for the values a confidence list can hold (shortest repr that round-trips; at least one decimal)
inline std::string java_double_to_string(double d) {
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

inline void append_int(std::string &s, long v) {
  char b[24];
  int n = snprintf(b, sizeof b, "%ld", v);
  s.append(b, (size_t)n);
}
// TaxonCounts.lengthString :114-121 over un-merged hits
inline void append_length_string(std::string &s, const slk_hit *h, size_t n, int k) {
  long a = 0, b = 0;
  size_t border = n;
  for (size_t i = 0; i < n; i++) if (h[i].taxon == SLK_TAXON_MATE_PAIR_BORDER) { border = i; break; }
  for (size_t i = 0; i < border; i++) a += h[i].count;
  append_int(s, a + (k - 1));
  if (border == n) return;
  for (size_t i = border + 1; i < n; i++) b += h[i].count;
  s.push_back('|');
  append_int(s, b + (k - 1));
}
// The same on the fromHits-merged (taxa, counts), literally: needed for a row merged from several fragments that share a title
// (Classifier.scala:92), which can hold several borders -- adjacent ones (equal ordinals) collapse into one merged entry
// before indexOf / take / drop apply.  For a single fragment (at most one border) both forms agree.
inline void append_length_string_merged(std::string &s, const slk_hit *h, size_t n, int k) {
  std::vector<std::pair<int32_t, long>> m;
  for (size_t i = 0; i < n; i++) {
    if (!m.empty() && m.back().first == h[i].taxon) m.back().second += h[i].count;
    else m.emplace_back(h[i].taxon, h[i].count);
  }
  size_t border = m.size();
  for (size_t i = 0; i < m.size(); i++) if (m[i].first == SLK_TAXON_MATE_PAIR_BORDER) { border = i; break; }
  long a = 0, b = 0;
  for (size_t i = 0; i < border; i++) a += m[i].second;
  append_int(s, a + (k - 1));
  if (border == m.size()) return;
  for (size_t i = border + 1; i < m.size(); i++) b += m[i].second;
  s.push_back('|');
  append_int(s, b + (k - 1));
}
// TaxonCounts.pairsInOrderString :94-110 (TaxonCounts.fromHits :31-48 merges adjacent equal taxa)
inline void append_pairs_in_order(std::string &s, const slk_hit *h, size_t n) {
  size_t i = 0;
  while (i < n) {
    size_t j = i;
    long c = 0;
    while (j < n && h[j].taxon == h[i].taxon) { c += h[j].count; j++; }
    if (h[i].taxon == SLK_TAXON_MATE_PAIR_BORDER) s.append("|:|");
    else if (h[i].taxon == SLK_TAXON_AMBIGUOUS) { s.append("A:"); append_int(s, c); }
    else { append_int(s, h[i].taxon); s.push_back(':'); append_int(s, c); }
    if (j < n) s.push_back(' ');
    i = j;
  }
}

inline int gzip_level() {  // SLK_GZIP_LEVEL: 1..9; default = zlib's default level, as java.util.zip's Deflater behind Hadoop's GzipCodec
  static const int level = [] {
    const char *e = getenv("SLK_GZIP_LEVEL");
    int v = e ? atoi(e) : 0;
    return v >= 1 && v <= 9 ? v : Z_DEFAULT_COMPRESSION;
  }();
  return level;
}

// One-shot gzip members compress about twice as fast with libdeflate as with zlib at the same level; this image has the
// library without its header, so its four stable entry points are declared here and resolved at run time (as libbz2 in
// seqio.hpp).  Absent, or with SLK_GZIP_IMPL=zlib, zlib does the work: either way the member is ordinary gzip.
struct LibDeflate {
  void *(*alloc)(int) = nullptr;
  size_t (*compress)(void *, const void *, size_t, void *, size_t) = nullptr;
  size_t (*bound)(void *, size_t) = nullptr;
  void (*release)(void *) = nullptr;
  bool ok = false;
  LibDeflate() {
    const char *impl = getenv("SLK_GZIP_IMPL");
    if (impl && std::string(impl) == "zlib") return;
    void *h = dlopen("libdeflate.so.0", RTLD_NOW);
    if (!h) return;
    alloc = (void *(*)(int))dlsym(h, "libdeflate_alloc_compressor");
    compress = (size_t (*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_gzip_compress");
    bound = (size_t (*)(void *, size_t))dlsym(h, "libdeflate_gzip_compress_bound");
    release = (void (*)(void *))dlsym(h, "libdeflate_free_compressor");
    ok = alloc && compress && bound && release;
  }
};
inline const LibDeflate &libdeflate() { static const LibDeflate l; return l; }

inline std::string gzip_member(const std::string &text) {
  if (libdeflate().ok) {
    const LibDeflate &ld = libdeflate();
    struct PerThread {  // (a compressor is reusable, but serves one thread at a time)
      void *c = nullptr;
      ~PerThread() { if (c) libdeflate().release(c); }
    };
    static thread_local PerThread t;
    if (!t.c) t.c = ld.alloc(gzip_level() == Z_DEFAULT_COMPRESSION ? 6 : gzip_level());
    if (t.c) {
      std::string out;
      out.resize(ld.bound(t.c, text.size()));
      const size_t n = ld.compress(t.c, text.data(), text.size(), &out[0], out.size());
      if (n > 0) { out.resize(n); return out; }
    }
  }
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (deflateInit2(&zs, gzip_level(), Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
  std::string out;
  out.resize(deflateBound(&zs, (uLong)text.size()) + 32);
  zs.next_in = (Bytef *)text.data();
  zs.avail_in = (uInt)text.size();
  zs.next_out = (Bytef *)&out[0];
  zs.avail_out = (uInt)out.size();
  int rc = deflate(&zs, Z_FINISH);
  if (rc != Z_STREAM_END) { deflateEnd(&zs); throw std::runtime_error("deflate failed"); }
  out.resize(zs.total_out);
  deflateEnd(&zs);
  return out;
}

class ThreadPool {
  std::vector<std::thread> th_;
  std::queue<std::function<void()>> q_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool stop_ = false;

 public:
  explicit ThreadPool(size_t n) {
    for (size_t i = 0; i < n; i++)
      th_.emplace_back([this] {
        for (;;) {
          std::function<void()> f;
          {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
            if (q_.empty()) return;
            f = std::move(q_.front());
            q_.pop();
          }
          f();
        }
      });
  }
  ~ThreadPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  size_t size() const { return th_.size(); }
  template <class F> auto submit(F f) -> std::future<decltype(f())> {
    auto task = std::make_shared<std::packaged_task<decltype(f())()>>(std::move(f));
    auto fut = task->get_future();
    { std::lock_guard<std::mutex> lk(mu_); q_.push([task] { (*task)(); }); }
    cv_.notify_one();
    return fut;
  }
};

// One classified batch: the fragments and what slk_classify_batch returned for them (threshold-major taxon / classified).
struct ClassifiedBatch {
  FragmentBatchPtr frags;
  int C = 0;
  std::vector<int32_t> taxon;
  std::vector<uint8_t> classified;
  std::vector<uint64_t> hit_offs;
  std::unique_ptr<slk_hit[]> hits;  // hit_offs[n] entries (allocated uninitialised: zero-filling 8 bytes per base per batch showed)
  size_t hits_capacity = 0;
  std::vector<uint64_t> span_offs;  // only when spans were requested
  std::vector<slk_span> spans;
  void reserve_hits(size_t cap) {  // (the untouched tail of the allocation is never paged in)
    if (cap <= hits_capacity) return;
    hits_capacity = cap + cap / 8;
    hits.reset(new slk_hit[hits_capacity]);
  }
  void recycle() { frags.reset(); }  // the fragments go back to their own pool; the result buffers keep their memory
  size_t footprint() const { return hits_capacity * sizeof(slk_hit) / 8 + spans.capacity() * sizeof(slk_span); }
};
inline std::shared_ptr<ClassifiedBatch> new_classified_batch() {
  static auto *pool = new Recycler<ClassifiedBatch>(32, (size_t)256 << 20);
  return std::shared_ptr<ClassifiedBatch>(pool->acquire(), [](ClassifiedBatch *b) { pool->release(b); });
}

struct OutputOptions {
  std::string output, sample_regex;
  std::vector<double> thresholds;
  bool with_unclassified = true, detailed = true;
  int k = 35;
};

class OutputSink {
  using Key = std::pair<int, std::string>;  // (threshold index, sample id)
  struct SliceOut {
    std::map<Key, std::string> gz;
    std::map<Key, std::map<Taxon, long>> counts;
    std::vector<uint64_t> repeated;  // hashes of titles that had been seen before (titles.hpp)
  };
  const OutputOptions o_;
  const Taxonomy &tax_;
  std::vector<std::string> locations_;
  std::regex re_;
  ThreadPool pool_;
  std::deque<std::future<SliceOut>> pending_;
  std::map<Key, FILE *> files_;
  std::map<Key, std::map<Taxon, long>> counts_;
  mutable ConcurrentTitleSet titles_;  // every title that produced a row
  RepeatedTitles repeated_;
  bool drained_ = false;

 public:
  // Per-read output: group 1 of the first match, else "other" (Classifier.classifyHits, Classifier.scala:138-142).
  // Reports only (--nodetailed): the reference takes the SQL route, ifnull(regexp_extract(title, re, 1), "other")
  // (SQLClassifier, Classifier.scala:297-300) -- and regexp_extract yields "" rather than null without a match, so there the
  // unmatched titles form the sample "" (report file "_kreport.txt").  Mirrored as is.
  //
  // A match in which group 1 took no part: Regex.Match.group(1) is null there, and so is ClassifiedRead.sampleId.  What becomes of
  // it is Spark's doing (org.apache.spark:spark-sql_2.12:3.5.6, not under /root/reference; parity unpinned): the per-read rows are
  // written with partitionBy("sample") (Classifier.scala:207-211), whose directory for a null OR EMPTY value is
  // sample=__HIVE_DEFAULT_PARTITION__ and which %-escapes the characters a path cannot hold (ExternalCatalogUtils.
  // getPartitionPathString / escapePathName), and the report's name is read back from that directory's name
  // (makeReportsFromClassifications :232-241).  The reports-only route's regexp_extract gives "" for such a group as for no match.
  // (std::regex is ECMAScript, not java.util.regex: see --help.)
  static std::string spark_partition_value(const std::string &v, bool is_null) {
    if (is_null || v.empty()) return "__HIVE_DEFAULT_PARTITION__";
    std::string out;
    for (unsigned char c : v) {
      const bool esc = (c >= 0x01 && c <= 0x1F) || c == 0x7F || strchr("\"#%'*/:=?\\{[]^", c) != nullptr;
      if (esc) { char b[8]; snprintf(b, sizeof b, "%%%02X", c); out += b; }
      else out.push_back((char)c);
    }
    return out;
  }
  std::string sample_of(std::string_view title) const {
    if (o_.sample_regex.empty()) return "all";
    std::cmatch m;
    if (std::regex_search(title.data(), title.data() + title.size(), m, re_) && m.size() > 1)
      return o_.detailed ? spark_partition_value(m[1].str(), !m[1].matched) : m[1].str();
    return o_.detailed ? "other" : "";
  }
  // ClassifiedRead.outputLine, Classifier.scala:41-44
  static void append_output_line(std::string &s, bool classified, std::string_view title, int32_t taxon, const slk_hit *h, size_t n,
                                 int k, bool merged_row) {
    s.push_back(classified ? 'C' : 'U');
    s.push_back('\t');
    s.append(title);
    s.push_back('\t');
    append_int(s, taxon);
    s.push_back('\t');
    if (merged_row) append_length_string_merged(s, h, n, k);
    else append_length_string(s, h, n, k);
    s.push_back('\t');
    append_pairs_in_order(s, h, n);
    s.push_back('\n');
  }

 private:
  SliceOut do_slice(std::shared_ptr<const ClassifiedBatch> b, size_t i0, size_t i1) const {
    SliceOut out;
    std::map<Key, std::string> text;
    const size_t n = b->frags->size();
    std::vector<uint64_t> hashes;
    hashes.reserve(i1 - i0);
    for (size_t i = i0; i < i1; i++) {
      const size_t h0 = b->hit_offs[i], h1 = b->hit_offs[i + 1];
      std::string_view title = b->frags->title(i);
      hashes.push_back(title_hash(title));   // (every fragment, with or without a row: titles.hpp)
      if (h1 == h0) continue;  // no span => no row at all (grouping is over span rows, Classifier.scala:92)
      std::string sample = sample_of(title);
      for (int c = 0; c < b->C; c++) {
        const bool classified = b->classified[(size_t)c * n + i] != 0;
        if (!classified && !o_.with_unclassified) continue;
        const int32_t t = b->taxon[(size_t)c * n + i];
        Key key(c, sample);
        out.counts[key][t] += 1;
        if (!o_.detailed) continue;
        append_output_line(text[key], classified, title, t, &b->hits[h0], h1 - h0, o_.k, false);
      }
    }
    titles_.insert_many(hashes, out.repeated);
    for (auto &kv : text) out.gz[kv.first] = gzip_member(kv.second);
    return out;
  }

  void commit(SliceOut so) {
    for (auto &kv : so.gz) {
      FILE *&f = files_[kv.first];
      if (!f) {
        std::string dir = locations_[kv.first.first] + "/sample=" + kv.first.second;  // Classifier.perReadOutputsLocation :415-416
        std::filesystem::create_directories(dir);
        f = fopen((dir + "/part-00000.txt.gz").c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write under " + dir);
      }
      if (fwrite(kv.second.data(), 1, kv.second.size(), f) != kv.second.size()) throw std::runtime_error("write failed");
    }
    for (auto &kv : so.counts) {
      auto &dst = counts_[kv.first];
      for (auto &tc : kv.second) dst[tc.first] += tc.second;
    }
    repeated_.add(so.repeated);
  }

 public:
  OutputSink(const OutputOptions &o, const Taxonomy &tax, size_t threads)
      : o_(o), tax_(tax), pool_(std::max<size_t>(1, threads)) {
    size_t max_dec = 0;  // thresholds' directory names (Classifier.scala:189-191)
    for (double t : o_.thresholds) { std::string s = java_double_to_string(t); max_dec = std::max(max_dec, s.size() - s.find('.') - 1); }
    for (double t : o_.thresholds) {
      char b[64];
      snprintf(b, sizeof b, "%.*f", (int)max_dec, t);
      locations_.push_back(o_.output + "_c" + b);
      std::filesystem::create_directories(locations_.back());
    }
    if (!o_.sample_regex.empty()) re_ = std::regex(o_.sample_regex);
  }
  ~OutputSink() { for (auto &kv : files_) if (kv.second) fclose(kv.second); }

  void submit(std::shared_ptr<const ClassifiedBatch> b) {
    const size_t n = b->frags->size(), SL = 32768;
    for (size_t i0 = 0; i0 < n; i0 += SL) {
      size_t i1 = std::min(n, i0 + SL);
      pending_.push_back(pool_.submit([this, b, i0, i1] { return do_slice(b, i0, i1); }));
    }
    while (pending_.size() > 8 * pool_.size()) { commit(pending_.front().get()); pending_.pop_front(); }
  }

  // All rows of the first pass are on disk and counted once this returns.
  void drain() {
    while (!pending_.empty()) { commit(pending_.front().get()); pending_.pop_front(); }
    for (auto &kv : files_) if (kv.second) { fclose(kv.second); kv.second = nullptr; }
    drained_ = true;
  }
  RepeatedTitles &repeated() { return repeated_; }
  bool has_title(uint64_t h) { return titles_.contains(h); }   // (after drain(): a fragment of the run had this title)
  const OutputOptions &options() const { return o_; }

  // ---- corrections for titles that occur more than once (slacken_cli.cpp: resolve_repeated_titles); after drain() ----
  void adjust_count(int c, const std::string &sample, Taxon t, long delta) {
    auto &m = counts_[Key(c, sample)];
    if ((m[t] += delta) == 0) m.erase(t);
  }
  // Drops the per-read lines whose title is in `titles` from the part files and appends `extra` (text per (threshold, sample)).
  void replace_rows(const std::function<bool(std::string_view)> &in_titles, const std::map<std::pair<int, std::string>, std::string> &extra) {
    if (!o_.detailed) return;
    std::map<Key, bool> todo;
    for (auto &kv : files_) todo[kv.first] = true;
    for (auto &kv : extra) todo[kv.first] = true;
    for (auto &kt : todo) {
      const Key &key = kt.first;
      const std::string dir = locations_[key.first] + "/sample=" + key.second;
      const std::string path = dir + "/part-00000.txt.gz", tmp = path + ".tmp";
      std::filesystem::create_directories(dir);
      FILE *out = fopen(tmp.c_str(), "wb");
      if (!out) throw std::runtime_error("cannot write under " + dir);
      std::string text;
      bool wrote_any = false;
      auto flush = [&](bool all) {
        if (text.empty() || (!all && text.size() < ((size_t)8 << 20))) return;
        std::string gz = gzip_member(text);
        if (fwrite(gz.data(), 1, gz.size(), out) != gz.size()) throw std::runtime_error("write failed");
        text.clear();
        wrote_any = true;
      };
      if (std::filesystem::exists(path)) {
        gzFile in = gzopen(path.c_str(), "rb");
        if (!in) throw std::runtime_error("cannot read " + path);
        gzbuffer(in, 1 << 20);
        std::string line;
        std::vector<char> buf(1 << 16);
        bool open_line = false;
        while (gzgets(in, buf.data(), (int)buf.size())) {
          const size_t len = strlen(buf.data());
          if (!open_line) line.clear();
          line.append(buf.data(), len);
          open_line = len == 0 || buf[len - 1] != '\n';
          if (open_line) continue;
          const size_t t0 = line.find('\t'), t1 = t0 == std::string::npos ? t0 : line.find('\t', t0 + 1);
          if (t1 != std::string::npos && in_titles(std::string_view(line).substr(t0 + 1, t1 - t0 - 1))) continue;
          text.append(line);
          flush(false);
        }
        gzclose(in);
      }
      auto ex = extra.find(key);
      if (ex != extra.end()) text.append(ex->second);
      const bool any_text = wrote_any || !text.empty();
      flush(true);
      if (fclose(out) != 0) { std::filesystem::remove(tmp); throw std::runtime_error("write failed: " + tmp); }   // (the old part file stays)
      if (any_text) {
        std::filesystem::rename(tmp, path);
      } else {
        // every row of this sample went and none came: the reference would have no such file (and a 0-byte .gz is not a gzip file)
        std::error_code ec;
        std::filesystem::remove(tmp, ec);
        std::filesystem::remove(path, ec);
        const auto dir = std::filesystem::path(path).parent_path();
        if (std::filesystem::is_empty(dir, ec)) std::filesystem::remove(dir, ec);
      }
    }
  }

  void finish() {  // Classifier.reportOutputLocation :419-420 + KrakenReport
    if (!drained_) drain();
    for (auto &kv : counts_) {
      if (kv.second.empty()) continue;
      std::vector<std::pair<Taxon, long>> counts(kv.second.begin(), kv.second.end());
      std::ofstream rep(locations_[kv.first.first] + "/" + kv.first.second + "_kreport.txt");
      KrakenReport(tax_, counts).print(rep);
    }
  }
};

}  // namespace slk_host

// seqio.hpp -- streaming sequence input of the host layer (C++17, header only): FASTA / FASTQ text, optionally gzip or bzip2
// compressed, single or paired, delivered as batches laid out the way slk_classify_batch wants them.
// Mirrors (S/ = src/main/scala/com/jnpersson/ in the reference):
//   FileInputs.forFile      S/kmers/input/FileInputs.scala:64-85    format by file name
//   FastaTextInput          S/kmers/input/FileInputs.scala:155-183  records separated by '>', lines by [\n\r]+, >= 2 lines
//   FastqTextInput          S/kmers/input/FileInputs.scala:188-221  every window of 4 lines (sliding by ONE line) whose 1st
//                                                                   line starts with '@' and 3rd with '+'
//   PairedInputReader       S/kmers/input/InputReader.scala:105-131 inner join on the header without /1 and /2
// Files are read in chunks; nothing here holds a whole file in memory except the un-joined remainder of a mate file whose
// records are not in the same order as the first file's.
#pragma once
#include <dlfcn.h>
#include <zlib.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

namespace slk_host {

inline bool ends_with(const std::string &s, const char *suf) {
  size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}
inline std::string lower(std::string s) {
  for (auto &c : s) c = (char)tolower(c);
  return s;
}

// plain / gzip (zlib reads both) / bzip2 (through the system's libbz2: this image has no bzlib.h, so the three stable
// high-level entry points are declared here and resolved at run time)
class ByteSource {
  gzFile g_ = nullptr;
  void *bz_ = nullptr;
  int (*bzread_)(void *, void *, int) = nullptr;
  void (*bzclose_)(void *) = nullptr;

 public:
  explicit ByteSource(const std::string &path) {
    if (ends_with(path, ".bz2")) {
      void *h = dlopen("libbz2.so.1", RTLD_NOW);
      if (!h) h = dlopen("libbz2.so.1.0", RTLD_NOW);
      if (!h) throw std::runtime_error("bzip2 input needs libbz2.so.1: " + path);
      auto bzopen = (void *(*)(const char *, const char *))dlsym(h, "BZ2_bzopen");
      bzread_ = (int (*)(void *, void *, int))dlsym(h, "BZ2_bzread");
      bzclose_ = (void (*)(void *))dlsym(h, "BZ2_bzclose");
      if (!bzopen || !bzread_ || !bzclose_) throw std::runtime_error("libbz2 lacks BZ2_bzopen/BZ2_bzread/BZ2_bzclose");
      bz_ = bzopen(path.c_str(), "rb");
      if (!bz_) throw std::runtime_error("cannot open " + path);
    } else {
      g_ = gzopen(path.c_str(), "rb");
      if (!g_) throw std::runtime_error("cannot open " + path);
      gzbuffer(g_, 1 << 20);
    }
  }
  ByteSource(const ByteSource &) = delete;
  ~ByteSource() {
    if (g_) gzclose(g_);
    if (bz_) bzclose_(bz_);
  }
  size_t read(char *dst, size_t cap) {  // 0 at the end of the file
    int n = g_ ? gzread(g_, dst, (unsigned)cap) : bzread_(bz_, dst, (int)cap);
    if (n < 0) throw std::runtime_error("read error (corrupt compressed input?)");
    return (size_t)n;
  }
};

inline size_t io_chunk_bytes() {  // SLK_IO_CHUNK: chunk size override, used by the tests to put chunk borders everywhere
  const char *e = getenv("SLK_IO_CHUNK");
  long v = e ? atol(e) : 0;
  return v > 0 ? (size_t)v : (size_t)8 << 20;
}

// (header token, nucleotides) of one file, in file order.  The views stay valid until the next call.
class RecordStream {
  ByteSource src_;
  bool fastq_;
  std::string buf_;
  size_t pos_ = 0;  // first unconsumed byte of buf_
  bool eof_ = false;
  std::string seq_tmp_;
  const size_t chunk_;

  bool fill() {  // drop the consumed prefix, append one chunk; false if nothing more could be read
    if (eof_) return false;
    if (pos_ > 0) { buf_.erase(0, pos_); pos_ = 0; }
    size_t old = buf_.size();
    buf_.resize(old + chunk_);
    size_t n = src_.read(&buf_[old], chunk_);
    buf_.resize(old + n);
    if (n == 0) eof_ = true;
    return n > 0;
  }
  static std::string_view first_token(std::string_view s) { return s.substr(0, s.find(' ')); }  // headerLine.split(" ")(0)

  // The line starting at p (Spark's text reader: lines end with \n, \r\n or \r).  0 = no line at p (end of file),
  // 1 = line found, -1 = its end cannot be decided yet (more data needed).
  int line_at(size_t p, std::string_view &line, size_t &next) const {
    if (p >= buf_.size()) return eof_ ? 0 : -1;
    const char *d = buf_.data();
    const char *a = (const char *)memchr(d + p, '\n', buf_.size() - p);
    size_t j = a ? (size_t)(a - d) : buf_.size();
    const char *b = (const char *)memchr(d + p, '\r', j - p);
    if (b) j = (size_t)(b - d);
    if (j == buf_.size()) j = std::string::npos;
    if (j == std::string::npos) {
      if (!eof_) return -1;
      line = std::string_view(buf_).substr(p);
      next = buf_.size();
      return 1;
    }
    if (buf_[j] == '\r' && j + 1 == buf_.size() && !eof_) return -1;  // \r\n may straddle the chunk border
    line = std::string_view(buf_).substr(p, j - p);
    next = (buf_[j] == '\r' && j + 1 < buf_.size() && buf_[j + 1] == '\n') ? j + 2 : j + 1;
    return 1;
  }

  bool next_fastq(std::string_view &header, std::string_view &seq) {
    for (;;) {
      std::string_view l0, l1, l2;
      size_t n0 = 0, n1 = 0, n2 = 0;
      int r0 = line_at(pos_, l0, n0);
      int r1 = r0 == 1 ? line_at(n0, l1, n1) : r0;
      int r2 = r1 == 1 ? line_at(n1, l2, n2) : r1;
      if (r0 == -1 || r1 == -1 || r2 == -1) {
        fill();  // (sets eof_ when there is nothing more, which turns every -1 into 0 or 1)
        continue;
      }
      if (r0 == 0) return false;
      if (r1 == 0 || r2 == 0) { pos_ = n0; continue; }  // fewer than 3 lines left: no window can start here
      const bool rec = !l0.empty() && l0[0] == '@' && !l2.empty() && l2[0] == '+';
      pos_ = n0;  // the window slides by one line
      if (rec) {
        header = first_token(l0).substr(1);
        seq = l1;
        return true;
      }
    }
  }

  bool next_fasta(std::string_view &header, std::string_view &seq) {
    for (;;) {
      if (pos_ > buf_.size()) return false;
      size_t end = buf_.find('>', pos_);
      if (end == std::string::npos) {
        if (!eof_) { fill(); continue; }
        end = buf_.size();
      }
      // String.split("[\n\r]+"): a leading empty string is kept, trailing ones are dropped
      std::string_view rec = std::string_view(buf_).substr(pos_, end - pos_);
      const bool last = end == buf_.size();
      size_t i = 0, nlines = 0;
      std::string_view first;
      seq_tmp_.clear();
      std::string_view only;
      while (i < rec.size()) {
        size_t j = rec.find_first_of("\n\r", i);
        if (j == std::string_view::npos) j = rec.size();
        if (j > i || nlines == 0) {
          std::string_view line = rec.substr(i, j - i);
          if (nlines == 0) first = line;
          else if (nlines == 1) only = line;
          else {
            if (nlines == 2) seq_tmp_.assign(only);
            seq_tmp_.append(line);
          }
          nlines++;
        }
        while (j < rec.size() && (rec[j] == '\n' || rec[j] == '\r')) j++;
        i = j;
      }
      pos_ = end + 1;
      if (nlines >= 2) {
        header = first_token(first);
        seq = nlines == 2 ? only : std::string_view(seq_tmp_);
        return true;
      }
      if (last) { pos_ = buf_.size() + 1; return false; }
    }
  }

 public:
  static bool is_fastq_name(const std::string &file) {  // FileInputs.forFile :64-85
    std::string lo = lower(file);
    return ends_with(lo, "fq") || ends_with(lo, "fastq") || ends_with(lo, ".fq.gz") || ends_with(lo, ".fastq.gz") ||
           ends_with(lo, ".fq.bz2") || ends_with(lo, ".fastq.bz2");
  }
  explicit RecordStream(const std::string &file) : src_(file), fastq_(is_fastq_name(file)), chunk_(io_chunk_bytes()) {}
  bool next(std::string_view &header, std::string_view &seq) { return fastq_ ? next_fastq(header, seq) : next_fasta(header, seq); }
};

// A RecordStream read ahead on its own thread (decompression and line splitting of one file), handed over in chunks of a few
// thousand records.  Paired input runs two of these side by side; a gzip stream inflates at a few hundred MB/s on one core,
// so the two files of a pair are best inflated concurrently.  The views stay valid until the next call.
class AsyncRecordStream {
  struct Chunk {
    std::string blob;
    std::vector<uint32_t> pos;  // 4 per record: header offset, header length, sequence offset, sequence length
  };
  std::deque<std::unique_ptr<Chunk>> q_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool done_ = false, stop_ = false;
  std::string error_;
  std::unique_ptr<Chunk> cur_;
  size_t cur_i_ = 0;
  std::thread th_;

  void run(std::string file) {
    try {
      RecordStream rs(file);
      std::string_view h, sq;
      auto c = std::make_unique<Chunk>();
      auto flush = [&]() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return q_.size() < 4 || stop_; });
        if (stop_) return false;
        q_.push_back(std::move(c));
        cv_.notify_all();
        c = std::make_unique<Chunk>();
        return true;
      };
      while (rs.next(h, sq)) {
        if (c->blob.size() + h.size() + sq.size() > 0xF0000000u) throw std::runtime_error("record too large: " + file);
        c->pos.push_back((uint32_t)c->blob.size()); c->pos.push_back((uint32_t)h.size());
        c->blob.append(h);
        c->pos.push_back((uint32_t)c->blob.size()); c->pos.push_back((uint32_t)sq.size());
        c->blob.append(sq);
        if (c->pos.size() >= 4 * 4096 || c->blob.size() >= ((size_t)4 << 20)) if (!flush()) return;
      }
      if (!c->pos.empty()) flush();
    } catch (const std::exception &e) {
      std::lock_guard<std::mutex> lk(mu_);
      error_ = e.what();
    }
    std::lock_guard<std::mutex> lk(mu_);
    done_ = true;
    cv_.notify_all();
  }

 public:
  explicit AsyncRecordStream(const std::string &file) : th_([this, file] { run(file); }) {}
  ~AsyncRecordStream() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; cv_.notify_all(); }
    th_.join();
  }
  bool next(std::string_view &header, std::string_view &seq) {
    if (!cur_ || cur_i_ * 4 >= cur_->pos.size()) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return !q_.empty() || done_; });
      if (!error_.empty()) throw std::runtime_error(error_);
      if (q_.empty()) { cur_.reset(); return false; }
      cur_ = std::move(q_.front());
      q_.pop_front();
      cur_i_ = 0;
      cv_.notify_all();
    }
    const uint32_t *p = &cur_->pos[cur_i_ * 4];
    header = std::string_view(cur_->blob).substr(p[0], p[1]);
    seq = std::string_view(cur_->blob).substr(p[2], p[3]);
    cur_i_++;
    return true;
  }
};

inline std::string_view remove_suffix(std::string_view h, const char *suf) {  // header.replaceAll(suffix + "$", "")
  size_t n = strlen(suf);
  return (h.size() >= n && h.compare(h.size() - n, n, suf) == 0) ? h.substr(0, h.size() - n) : h;
}

// A batch of fragments in the layout of slk_classify_batch: concatenated bases + offsets (mates likewise), titles.
struct FragmentBatch {
  std::vector<uint8_t> bases, mate_bases;
  std::vector<uint64_t> offs{0}, mate_offs{0};
  std::string titles;
  std::vector<uint64_t> title_off{0};
  bool paired = false;
  size_t size() const { return offs.size() - 1; }
  std::string_view title(size_t i) const { return std::string_view(titles).substr(title_off[i], title_off[i + 1] - title_off[i]); }
  std::string_view seq(size_t i) const { return std::string_view((const char *)bases.data() + offs[i], offs[i + 1] - offs[i]); }
  std::string_view mate(size_t i) const { return std::string_view((const char *)mate_bases.data() + mate_offs[i], mate_offs[i + 1] - mate_offs[i]); }
  void add(std::string_view title, std::string_view s, const std::string_view *m) {
    titles.append(title);
    title_off.push_back(titles.size());
    bases.insert(bases.end(), s.begin(), s.end());
    offs.push_back(bases.size());
    if (m) {
      mate_bases.insert(mate_bases.end(), m->begin(), m->end());
      mate_offs.push_back(mate_bases.size());
    }
  }
};

// All fragments of a list of input files (or of pairs of files), in file order.
class FragmentSource {
  std::vector<std::string> files_;
  bool paired_;
  size_t next_file_ = 0;
  std::unique_ptr<AsyncRecordStream> s1_, s2_;
  bool joined_ = false;  // the rest of the mate file has been loaded into mates_ (its order differs from the first file's)
  std::unordered_map<std::string, std::string> mates_;
  std::string h1_, seq1_;

  bool open_next() {
    if (next_file_ >= files_.size()) return false;
    s1_ = std::make_unique<AsyncRecordStream>(files_[next_file_]);
    if (paired_) s2_ = std::make_unique<AsyncRecordStream>(files_[next_file_ + 1]);
    next_file_ += paired_ ? 2 : 1;
    joined_ = false;
    mates_.clear();
    return true;
  }

 public:
  FragmentSource(std::vector<std::string> files, bool paired) : files_(std::move(files)), paired_(paired) {}

  // Appends up to max_fragments (and about max_bases) to b; false when every file is exhausted and nothing was added.
  bool fill(FragmentBatch &b, size_t max_fragments, size_t max_bases) {
    b.paired = paired_;
    size_t added = 0;
    while (added < max_fragments && b.bases.size() + b.mate_bases.size() < max_bases) {
      if (!s1_ && !open_next()) break;
      std::string_view h, s;
      if (!s1_->next(h, s)) { s1_.reset(); s2_.reset(); continue; }
      if (!paired_) { b.add(h, s, nullptr); added++; continue; }
      // Paired: the reference joins the two files on the header.  While both files list their records in the same order
      // the join is a lockstep walk; at the first disagreement the rest of the mate file becomes the join's hash side.
      h1_.assign(remove_suffix(h, "/1"));
      seq1_.assign(s);
      if (!joined_) {
        std::string_view h2, m;
        if (s2_->next(h2, m)) {
          if (remove_suffix(h2, "/2") == std::string_view(h1_)) { b.add(h1_, seq1_, &m); added++; continue; }
          mates_.emplace(std::string(remove_suffix(h2, "/2")), std::string(m));
        }
        while (s2_->next(h2, m)) mates_.emplace(std::string(remove_suffix(h2, "/2")), std::string(m));
        joined_ = true;
      }
      auto it = mates_.find(h1_);
      if (it == mates_.end()) continue;  // inner join: no mate, no fragment
      std::string_view m(it->second);
      b.add(h1_, seq1_, &m);
      added++;
    }
    return added > 0;
  }
};

// Reads batches on its own thread, at most `depth` ahead of the consumer.
class BatchPrefetcher {
  FragmentSource src_;
  size_t max_fragments_, max_bases_, depth_;
  std::deque<std::unique_ptr<FragmentBatch>> q_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool done_ = false, stop_ = false;
  std::string error_;
  std::thread th_;

  void run() {
    try {
      for (;;) {
        auto b = std::make_unique<FragmentBatch>();
        if (!src_.fill(*b, max_fragments_, max_bases_)) break;
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return q_.size() < depth_ || stop_; });
        if (stop_) return;
        q_.push_back(std::move(b));
        cv_.notify_all();
      }
    } catch (const std::exception &e) {
      std::lock_guard<std::mutex> lk(mu_);
      error_ = e.what();
    }
    std::lock_guard<std::mutex> lk(mu_);
    done_ = true;
    cv_.notify_all();
  }

 public:
  BatchPrefetcher(std::vector<std::string> files, bool paired, size_t max_fragments = (size_t)1 << 20,
                  size_t max_bases = (size_t)512 << 20, size_t depth = 2)
      : src_(std::move(files), paired), max_fragments_(max_fragments), max_bases_(max_bases), depth_(depth), th_([this] { run(); }) {}
  ~BatchPrefetcher() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; cv_.notify_all(); }
    th_.join();
  }
  std::unique_ptr<FragmentBatch> next() {  // nullptr at the end of the input
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !q_.empty() || done_; });
    if (!error_.empty()) throw std::runtime_error(error_);
    if (q_.empty()) return nullptr;
    auto b = std::move(q_.front());
    q_.pop_front();
    cv_.notify_all();
    return b;
  }
};

}  // namespace slk_host

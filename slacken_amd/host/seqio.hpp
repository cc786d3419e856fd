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
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include "parbz2.hpp"
#include "pargz.hpp"
#include "titles.hpp"

#include <condition_variable>
#include <cstring>
#include <atomic>
#include <deque>
#include <map>
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

// How many compressed input files are being read side by side (set by whoever opens them): a gzip file's inflate threads are
// the cores' share of it.
inline std::atomic<int> &gz_concurrent_files() { static std::atomic<int> v{1}; return v; }
// the cores this process may really use: the machine's, its affinity mask, its cgroup's CPU quota -- whichever is least
inline unsigned effective_cpus() {
  static const unsigned n = [] {
    unsigned v = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) v = std::min<unsigned>(v, std::max(1, CPU_COUNT(&set)));
    long quota = -1, period = 100000;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "max 100000" or "<quota> <period>"
      char q[64] = {0};
      if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atol(q);
      fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
      if (fscanf(g, "%ld", &quota) != 1) quota = -1;
      fclose(g);
      if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%ld", &period) != 1) period = 100000; fclose(h); }
    }
    if (quota > 0 && period > 0) v = std::min<unsigned>(v, (unsigned)std::max<long>(1, (quota + period - 1) / period));
    return v;
  }();
  return n;
}
inline int gz_threads() {  // SLK_GZ_THREADS: threads inflating ONE gzip file (1: this decoder without the guessing; 0: zlib's gzread)
  const char *e = getenv("SLK_GZ_THREADS");
  if (e && atol(e) >= 0 && (e[0] >= '0' && e[0] <= '9')) return (int)atol(e);
  // (under four threads per file the guessing costs about what it gains -- twice the CPU per byte: one thread then, which decodes
  // straight on and is still faster than zlib)
  const unsigned share = effective_cpus() / (unsigned)std::max(1, gz_concurrent_files().load());
  return share < 4 ? 1 : (int)std::min<unsigned>(16, share);
}
inline size_t gz_group() {  // SLK_GZ_GROUP: compressed chunks whose text is parsed as one segment (= one batch for the device)
  const char *e = getenv("SLK_GZ_GROUP");
  return e && atol(e) > 0 ? (size_t)atol(e) : 4;
}
inline size_t gz_chunk_bytes() {  // SLK_GZ_CHUNK: compressed bytes per chunk (the tests put chunk borders everywhere with it)
  const char *e = getenv("SLK_GZ_CHUNK");
  return e && atol(e) > 0 ? (size_t)atol(e) : (size_t)1 << 20;
}

// bzip2 from a pipe (or with SLK_GZ_THREADS=0): libbz2's streaming decoder, stream after stream -- a .bz2 file may be several
// concatenated streams (pbzip2 writes such files; Hadoop's codec reads them through), which the library's BZ2_bzread stops after the
// first of, silently.  This image has no bzlib.h: bz_stream is declared here as bzlib.h 1.0 declares it and the three entry points
// are resolved at run time.
class Bz2Stream {
  struct bz_stream {
    char *next_in; unsigned int avail_in, total_in_lo32, total_in_hi32;
    char *next_out; unsigned int avail_out, total_out_lo32, total_out_hi32;
    void *state;
    void *(*bzalloc)(void *, int, int); void (*bzfree)(void *, void *); void *opaque;
  };
  int (*init_)(bz_stream *, int, int) = nullptr;
  int (*run_)(bz_stream *) = nullptr;
  int (*end_)(bz_stream *) = nullptr;
  FILE *f_ = nullptr;
  bz_stream z_{};
  bool open_ = false, eof_ = false, any_stream_ = false;
  std::vector<char> in_;

 public:
  explicit Bz2Stream(const std::string &path) : in_((size_t)1 << 20) {
    void *h = dlopen("libbz2.so.1", RTLD_NOW);
    if (!h) h = dlopen("libbz2.so.1.0", RTLD_NOW);
    if (!h) throw std::runtime_error("bzip2 input needs libbz2.so.1: " + path);
    init_ = (int (*)(bz_stream *, int, int))dlsym(h, "BZ2_bzDecompressInit");
    run_ = (int (*)(bz_stream *))dlsym(h, "BZ2_bzDecompress");
    end_ = (int (*)(bz_stream *))dlsym(h, "BZ2_bzDecompressEnd");
    if (!init_ || !run_ || !end_) throw std::runtime_error("libbz2 lacks BZ2_bzDecompressInit/BZ2_bzDecompress/BZ2_bzDecompressEnd");
    f_ = fopen(path.c_str(), "rb");
    if (!f_) throw std::runtime_error("cannot open " + path);
  }
  Bz2Stream(const Bz2Stream &) = delete;
  ~Bz2Stream() {
    if (open_) end_(&z_);
    if (f_) fclose(f_);
  }
  size_t read(char *dst, size_t cap) {
    size_t got = 0;
    while (got < cap) {
      if (z_.avail_in == 0 && !eof_) {
        const size_t k = fread(in_.data(), 1, in_.size(), f_);
        if (k == 0) eof_ = true;
        z_.next_in = in_.data();
        z_.avail_in = (unsigned int)k;
      }
      if (!open_) {
        if (z_.avail_in == 0) {   // no further stream
          if (!any_stream_) throw std::runtime_error("read error (corrupt compressed input?): not a bzip2 file");
          break;
        }
        char *ni = z_.next_in;
        const unsigned int ai = z_.avail_in;
        z_ = bz_stream{};
        z_.next_in = ni; z_.avail_in = ai;
        if (init_(&z_, 0, 0) != 0) throw std::runtime_error("read error (corrupt compressed input?): libbz2 initialisation");
        open_ = true;
      }
      z_.next_out = dst + got;
      z_.avail_out = (unsigned int)std::min<size_t>(cap - got, 0x7FFFFFFF);
      const unsigned int before = z_.avail_out;
      const int rc = run_(&z_);
      got += before - z_.avail_out;
      if (rc == 4 /* BZ_STREAM_END */) {
        end_(&z_);
        open_ = false;
        any_stream_ = true;
        continue;
      }
      if (rc != 0) throw std::runtime_error("read error (corrupt compressed input?): bzip2 data");
      if (eof_ && z_.avail_in == 0 && before == z_.avail_out) throw std::runtime_error("read error (corrupt compressed input?): unexpected end of the bzip2 data");
    }
    return got;
  }
};

// plain / gzip (zlib reads both) / bzip2.  A gzip or bzip2 FILE is decompressed on several threads (pargz.hpp, parbz2.hpp);
// pipes, small gzip files and SLK_GZ_THREADS=0 keep zlib's gzread and libbz2's streaming decoder.
class ByteSource {
  gzFile g_ = nullptr;
  std::unique_ptr<Bz2Stream> bz_;
  std::unique_ptr<slk::pargz::Reader> pz_;
  std::unique_ptr<slk::parbz2::Reader> pb_;

 public:
  static size_t bz2_chunk_bytes() {  // SLK_BZ2_CHUNK: compressed bytes per chunk of a bzip2 file (tests: chunk borders everywhere)
    const char *e = getenv("SLK_BZ2_CHUNK");
    return e && atol(e) > 0 ? (size_t)atol(e) : (size_t)2 << 20;
  }
  static bool regular_file(const std::string &path) {
    struct stat sb;
    return stat(path.c_str(), &sb) == 0 && S_ISREG(sb.st_mode);
  }
  static bool gzip_file_worth_threads(const std::string &path, size_t chunk) {
    struct stat sb;
    if (stat(path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode) || (size_t)sb.st_size < 2 * chunk) return false;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    unsigned char m[3] = {0, 0, 0};
    const size_t k = fread(m, 1, 3, f);
    fclose(f);
    return k == 3 && m[0] == 0x1f && m[1] == 0x8b && m[2] == 8;
  }

  explicit ByteSource(const std::string &path) {
    if (ends_with(path, ".bz2") && gz_threads() >= 1 && regular_file(path)) {
      // blocks of a bzip2 file are independent: several threads (parbz2.hpp), as Hadoop's splittable codec gives the reference
      pb_ = std::make_unique<slk::parbz2::Reader>(path, gz_threads(), bz2_chunk_bytes());
    } else if (ends_with(path, ".bz2")) {
      bz_ = std::make_unique<Bz2Stream>(path);
    } else if (gz_threads() >= 1 && gzip_file_worth_threads(path, gz_chunk_bytes())) {
      pz_ = std::make_unique<slk::pargz::Reader>(path, gz_threads(), gz_chunk_bytes());
    } else {
      g_ = gzopen(path.c_str(), "rb");
      if (!g_) throw std::runtime_error("cannot open " + path);
      gzbuffer(g_, 1 << 20);
    }
  }
  ByteSource(const ByteSource &) = delete;
  ~ByteSource() {
    if (g_) gzclose(g_);
  }
  size_t read(char *dst, size_t cap) {  // 0 at the end of the file
    if (pz_) return pz_->read(dst, cap);
    if (pb_) return pb_->read(dst, cap);
    if (bz_) return bz_->read(dst, cap);
    int n = gzread(g_, dst, (unsigned)cap);
    if (n < 0) throw std::runtime_error("read error (corrupt compressed input?)");
    if ((size_t)n < cap) {   // a gzip file that ends inside a member: gzread hands out what there was and says so only here
      int err = Z_OK;
      (void)gzerror(g_, &err);
      if (err == Z_BUF_ERROR) throw std::runtime_error("read error (corrupt compressed input?): unexpected end of the gzip data");
    }
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
  int pool = 0;  // which recycler it came from: 0 = batches assembled from runs of records, 1 = parsed chunks
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
  // records [i, i + n) of c as one run; strip: a suffix to drop from the titles (the "/1" of paired input) or nullptr
  void append(const FragmentBatch &c, size_t i, size_t n, const char *strip) {
    if (!strip) {
      const uint64_t t0 = c.title_off[i], shift = titles.size() - t0;
      titles.append(c.titles, t0, c.title_off[i + n] - t0);
      for (size_t r = 1; r <= n; r++) title_off.push_back(c.title_off[i + r] + shift);
    } else {
      for (size_t r = 0; r < n; r++) {
        titles.append(remove_suffix(c.title(i + r), strip));
        title_off.push_back(titles.size());
      }
    }
    const uint64_t b0 = c.offs[i], shift = bases.size() - b0;
    bases.insert(bases.end(), c.bases.begin() + b0, c.bases.begin() + c.offs[i + n]);
    for (size_t r = 1; r <= n; r++) offs.push_back(c.offs[i + r] + shift);
  }
  void append_mates(const FragmentBatch &c, size_t i, size_t n) {  // the sequences of c's records [i, i + n) as mates
    const uint64_t b0 = c.offs[i], shift = mate_bases.size() - b0;
    mate_bases.insert(mate_bases.end(), c.bases.begin() + b0, c.bases.begin() + c.offs[i + n]);
    for (size_t r = 1; r <= n; r++) mate_offs.push_back(c.offs[i + r] + shift);
  }
  void recycle() {  // empty, with the memory kept
    bases.clear(); mate_bases.clear(); titles.clear();
    offs.assign(1, 0); mate_offs.assign(1, 0); title_off.assign(1, 0);
    paired = false;
  }
  size_t footprint() const { return bases.capacity() + mate_bases.capacity() + titles.capacity() + 8 * (offs.capacity() + mate_offs.capacity() + title_off.capacity()); }
};

// Batches travel from the parsing threads through the classify loop to the output threads and die there; their buffers are
// tens of MB, which the allocator would map, fault in page by page and unmap again for every batch.  Finished objects come
// back here instead and are handed out again with their memory (and its pages) in place.
template <class T>
class Recycler {
  std::mutex mu_;
  std::vector<T *> free_;
  const size_t max_objects_, max_footprint_;

 public:
  Recycler(size_t max_objects, size_t max_footprint) : max_objects_(max_objects), max_footprint_(max_footprint) {}
  T *acquire() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (!free_.empty()) { T *p = free_.back(); free_.pop_back(); return p; }
    }
    return new T();
  }
  void release(T *p) {
    if (!p) return;
    if (p->footprint() <= max_footprint_) {
      p->recycle();
      std::lock_guard<std::mutex> lk(mu_);
      if (free_.size() < max_objects_) { free_.push_back(p); return; }
    }
    delete p;
  }
};
// (two pools, as the two kinds differ in size; never destroyed: threads may still hand objects back during exit())
inline Recycler<FragmentBatch> &fragment_batches(int pool) {
  static Recycler<FragmentBatch> *r[2] = {new Recycler<FragmentBatch>(24, (size_t)256 << 20), new Recycler<FragmentBatch>(48, (size_t)256 << 20)};
  return *r[pool];
}
struct FragmentBatchReturn { void operator()(FragmentBatch *b) const { if (b) fragment_batches(b->pool).release(b); } };
using FragmentBatchPtr = std::unique_ptr<FragmentBatch, FragmentBatchReturn>;
inline FragmentBatchPtr new_fragment_batch(int pool = 0) {
  FragmentBatch *b = fragment_batches(pool).acquire();
  b->pool = pool;
  return FragmentBatchPtr(b);
}

// ---- plain (uncompressed) files: mapped into memory and cut into segments that are parsed on several threads ----
// Both record rules of the reference are local -- a FASTQ record starts at every line that begins with '@' and whose second
// successor begins with '+' (the sliding window of next_fastq above), a FASTA record at the start of the file and after every
// '>' -- so a file can be cut anywhere: a segment owns the records that START inside it and reads as far past its end as they
// reach.  The records of the segments, taken in segment order, are those of the serial reader.
// (final = false: the buffer is the part of a file that has arrived so far -- a gzip file being inflated, pargz.hpp; parse() then
// says whether the records of the segment were all there, and the caller asks again when more has arrived.)
class PlainSegmentParser {
  const char *d_;
  size_t n_;
  bool fastq_;
  bool final_ = true;
  struct Line { size_t s, e, next; };  // [s, e) without its terminator; next = start of the following line

  Line line(size_t p) const {  // p < n_
    const char *a = (const char *)memchr(d_ + p, '\n', n_ - p);
    size_t j = a ? (size_t)(a - d_) : n_;
    const char *b = (const char *)memchr(d_ + p, '\r', j - p);
    if (b) j = (size_t)(b - d_);
    if (j == n_) return {p, n_, n_};
    return {p, j, (d_[j] == '\r' && j + 1 < n_ && d_[j + 1] == '\n') ? j + 2 : j + 1};
  }
  bool is_line_start(size_t p) const { return p == 0 || d_[p - 1] == '\n' || (d_[p - 1] == '\r' && d_[p] != '\n'); }
  static std::string_view first_token(std::string_view s) { return s.substr(0, s.find(' ')); }

  bool fastq(size_t a, size_t b, FragmentBatch &out) const {  // false: the data ended before the segment's records did (!final_)
    if (a >= n_) return true;
    const size_t p = is_line_start(a) ? a : line(a).next;  // (from inside a line, line() finds where that line ends)
    if (p >= b) return true;
    if (p >= n_) return final_;
    Line l0 = line(p);
    if (l0.next >= n_) return final_;
    Line l1 = line(l0.next);
    while (l1.next < n_) {  // (fewer than three lines left: no window can start here or later)
      Line l2 = line(l1.next);
      if (l0.e > l0.s && d_[l0.s] == '@' && l2.e > l2.s && d_[l2.s] == '+')
        out.add(first_token(std::string_view(d_ + l0.s, l0.e - l0.s)).substr(1), std::string_view(d_ + l1.s, l1.e - l1.s), nullptr);
      l0 = l1;
      l1 = l2;
      if (l0.s >= b) return true;  // the window slides by one line; the next segment owns this one
    }
    return final_;   // (a line that touches the end of what has arrived may go on, or be followed by the \n of a \r\n)
  }

  bool fasta(size_t a, size_t b, FragmentBatch &out) const {
    size_t s = 0;
    if (a > 0) {
      const char *q = (const char *)memchr(d_ + a - 1, '>', n_ - (a - 1));
      if (!q) return n_ >= b || final_;   // no record starts in [a, b) -- if [a, b) has arrived
      s = (size_t)(q - d_) + 1;
    }
    while (s < b) {
      const char *q = (const char *)memchr(d_ + s, '>', n_ - s);
      if (!q && !final_) return false;   // the record may go on beyond what has arrived
      const size_t end = q ? (size_t)(q - d_) : n_;
      // String.split("[\n\r]+"): a leading empty string is kept, trailing ones are dropped; the first line is the header
      size_t i = s, nlines = 0;
      std::string_view first;
      while (i < end) {
        const char *x = (const char *)memchr(d_ + i, '\n', end - i);
        size_t j = x ? (size_t)(x - d_) : end;
        const char *y = (const char *)memchr(d_ + i, '\r', j - i);
        if (y) j = (size_t)(y - d_);
        if (j > i || nlines == 0) {
          if (nlines == 0) first = std::string_view(d_ + i, j - i);
          else out.bases.insert(out.bases.end(), d_ + i, d_ + j);
          nlines++;
        }
        while (j < end && (d_[j] == '\n' || d_[j] == '\r')) j++;
        i = j;
      }
      if (nlines >= 2) {
        out.titles.append(first_token(first));
        out.title_off.push_back(out.titles.size());
        out.offs.push_back(out.bases.size());
      }
      if (end == n_) return true;
      s = end + 1;
    }
    return true;
  }

 public:
  PlainSegmentParser(const char *d, size_t n, bool fastq, bool final = true) : d_(d), n_(n), fastq_(fastq), final_(final) {}
  bool parse(size_t a, size_t b, FragmentBatch &out) const {  // the records starting in [a, b)
    return fastq_ ? fastq(a, b, out) : fasta(a, b, out);
  }
};

inline size_t parse_threads() {  // SLK_PARSE_THREADS: threads that parse one plain input file
  const char *e = getenv("SLK_PARSE_THREADS");
  long v = e ? atol(e) : 0;
  if (v > 0) return (size_t)v;
  return std::min<size_t>(8, std::max<unsigned>(2, effective_cpus() / 2));
}

// The records of one file, read ahead and handed over in chunks, in file order.  Compressed input is inflated and split on one
// thread (a gzip stream is serial; paired input runs two of these side by side); a plain file is mapped and its segments are
// parsed on several threads.  The views of next() stay valid until the next call.
class AsyncRecordStream {
  std::deque<FragmentBatchPtr> q_;  // serial producer: finished chunks
  std::map<size_t, FragmentBatchPtr> done_;  // parallel producers: finished segments by number
  size_t nseg_ = 0, seg_bytes_ = 0, next_claim_ = 0, next_out_ = 0, depth_ = 0;
  std::mutex mu_;
  std::condition_variable cv_;
  bool done_flag_ = false, stop_ = false, parallel_ = false;
  std::string error_;
  FragmentBatchPtr cur_;
  size_t cur_i_ = 0;
  std::vector<std::thread> th_;
  const char *map_ = nullptr;
  size_t map_len_ = 0;

  void run_serial(std::string file) {
    try {
      RecordStream rs(file);
      std::string_view h, sq;
      auto c = new_fragment_batch(1);
      auto flush = [&]() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return q_.size() < 4 || stop_; });
        if (stop_) return false;
        q_.push_back(std::move(c));
        cv_.notify_all();
        c = new_fragment_batch(1);
        return true;
      };
      while (rs.next(h, sq)) {
        c->add(h, sq, nullptr);
        if (c->size() >= 4096 || c->bases.size() >= ((size_t)4 << 20)) if (!flush()) return;
      }
      if (c->size()) flush();
    } catch (const std::exception &e) {
      std::lock_guard<std::mutex> lk(mu_);
      error_ = e.what();
    }
    std::lock_guard<std::mutex> lk(mu_);
    done_flag_ = true;
    cv_.notify_all();
  }

  void run_segments(bool fastq) {
    PlainSegmentParser parser(map_, map_len_, fastq);
    for (;;) {
      size_t i;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || next_claim_ >= nseg_ || next_claim_ < next_out_ + depth_; });
        if (stop_ || next_claim_ >= nseg_) return;
        i = next_claim_++;
      }
      auto c = new_fragment_batch(1);
      try {
        const size_t a = i * seg_bytes_, b = std::min(map_len_, a + seg_bytes_);
        c->bases.reserve((b - a) / 2 + 256);
        parser.parse(a, b, *c);
      } catch (const std::exception &e) {
        std::lock_guard<std::mutex> lk(mu_);
        error_ = e.what();
      }
      std::lock_guard<std::mutex> lk(mu_);
      done_[i] = std::move(c);
      cv_.notify_all();
    }
  }

  // A gzip file inflated on several threads into one buffer (pargz.hpp, region mode): segment i = the text of compressed chunk
  // i, parsed in place by the plain file's segment parser as soon as it and its successor have arrived.
  std::unique_ptr<slk::pargz::Reader> gz_;
  size_t gz_group_ = 1;
  void run_gz_segments(bool fastq) {
    for (;;) {
      size_t i;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || next_claim_ >= nseg_ || next_claim_ < next_out_ + depth_; });
        if (stop_ || next_claim_ >= nseg_) return;
        i = next_claim_++;
      }
      auto c = new_fragment_batch(1);
      try {
        slk::pargz::Reader::View v;
        const size_t first = i * gz_group_, last = std::min(gz_->segments(), first + gz_group_) - 1;
        bool ok = gz_->wait_segment(last, v, first);
        while (ok) {
          c->bases.reserve((v.end - v.begin) / 2 + 256);
          PlainSegmentParser parser(gz_->base(), v.avail, fastq, v.eof);
          if (parser.parse(v.begin, v.end, *c) || v.eof) break;
          c = new_fragment_batch(1);   // a record reached beyond what had arrived: again, with more
          ok = gz_->wait_more(last, v.avail, v, first);
        }
        if (!ok) return;   // (the stream is being closed)
        for (size_t k = first; k <= last; k++) gz_->segment_parsed(k);
      } catch (const std::exception &e) {
        std::lock_guard<std::mutex> lk(mu_);
        error_ = e.what();
      }
      std::lock_guard<std::mutex> lk(mu_);
      done_[i] = std::move(c);
      cv_.notify_all();
    }
  }

  static bool is_compressed(const std::string &file) {  // by name, or by the gzip magic (zlib would inflate it either way)
    if (ends_with(file, ".gz") || ends_with(file, ".bz2")) return true;
    FILE *f = fopen(file.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + file);
    unsigned char m[2] = {0, 0};
    size_t k = fread(m, 1, 2, f);
    fclose(f);
    return k == 2 && m[0] == 0x1f && m[1] == 0x8b;
  }

  FragmentBatchPtr pop() {  // the next non-empty chunk, nullptr at the end
    for (;;) {
      std::unique_lock<std::mutex> lk(mu_);
      FragmentBatchPtr c;
      if (parallel_) {
        cv_.wait(lk, [&] { return !error_.empty() || next_out_ >= nseg_ || done_.count(next_out_); });
        if (!error_.empty()) throw std::runtime_error(error_);
        if (next_out_ >= nseg_) return nullptr;
        auto it = done_.find(next_out_);
        c = std::move(it->second);
        done_.erase(it);
        next_out_++;
      } else {
        cv_.wait(lk, [&] { return !q_.empty() || done_flag_; });
        if (!error_.empty()) throw std::runtime_error(error_);
        if (q_.empty()) return nullptr;
        c = std::move(q_.front());
        q_.pop_front();
      }
      cv_.notify_all();
      if (c->size()) return c;
    }
  }

 public:
  static bool regular_file(const std::string &file) {  // (a pipe -- /dev/stdin, <(zcat x) -- can neither be mapped nor peeked into)
    struct stat sb;
    return stat(file.c_str(), &sb) == 0 && S_ISREG(sb.st_mode);
  }

  explicit AsyncRecordStream(const std::string &file) {
    if (regular_file(file) && !ends_with(file, ".bz2") && ByteSource::gzip_file_worth_threads(file, gz_chunk_bytes()) && gz_threads() >= 1) {
      try {
        gz_ = std::make_unique<slk::pargz::Reader>(file, gz_threads(), gz_chunk_bytes(), true);
      } catch (const slk::pargz::ReserveError &) {
        // the address space for the inflated file could not be reserved (ulimit -v, overcommit = 2): zlib reads the file as well
        th_.emplace_back([this, file] { run_serial(file); });
        return;
      }
      parallel_ = true;
      gz_group_ = gz_group();
      gz_->set_consumer_span(gz_group_);
      nseg_ = (gz_->segments() + gz_group_ - 1) / gz_group_;
      const size_t nt = std::max<size_t>(1, std::min(parse_threads(), nseg_));
      depth_ = nt + 2;
      const bool fastq = RecordStream::is_fastq_name(file);
      for (size_t t = 0; t < nt; t++) th_.emplace_back([this, fastq] { run_gz_segments(fastq); });
      return;
    }
    if (!regular_file(file) || is_compressed(file)) {
      th_.emplace_back([this, file] { run_serial(file); });
      return;
    }
    parallel_ = true;
    int fd = open(file.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + file);
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); throw std::runtime_error("cannot stat " + file); }
    map_len_ = (size_t)sb.st_size;
    if (map_len_) {
      void *m = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd, 0);
      close(fd);
      if (m == MAP_FAILED) throw std::runtime_error("cannot map " + file);
      madvise(m, map_len_, MADV_SEQUENTIAL);
      map_ = (const char *)m;
    } else {
      close(fd);
    }
    const char *e = getenv("SLK_IO_CHUNK");  // (the tests put segment borders everywhere with it)
    seg_bytes_ = e && atol(e) > 0 ? (size_t)atol(e) : (size_t)16 << 20;
    nseg_ = (map_len_ + seg_bytes_ - 1) / seg_bytes_;
    const size_t nt = std::max<size_t>(1, std::min(parse_threads(), nseg_));
    depth_ = nt + 2;
    const bool fastq = RecordStream::is_fastq_name(file);
    for (size_t t = 0; t < nt; t++) th_.emplace_back([this, fastq] { run_segments(fastq); });
  }
  AsyncRecordStream(const AsyncRecordStream &) = delete;
  ~AsyncRecordStream() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; cv_.notify_all(); }
    if (gz_) gz_->shutdown();   // (wakes the parsers that wait for data)
    for (auto &t : th_) t.join();
    if (map_) munmap((void *)map_, map_len_);
  }
  bool next(std::string_view &header, std::string_view &seq) {
    if (!cur_ || cur_i_ >= cur_->size()) {
      cur_ = pop();
      cur_i_ = 0;
      if (!cur_) return false;
    }
    header = cur_->title(cur_i_);
    seq = cur_->seq(cur_i_);
    cur_i_++;
    return true;
  }
  // Chunk-wise access: the current chunk and the position in it (nullptr at the end of the file); advance(n) consumes n of its
  // records, take() the whole chunk when nothing of it has been consumed.
  const FragmentBatch *current(size_t &i) {
    if (!cur_ || cur_i_ >= cur_->size()) {
      cur_ = pop();
      cur_i_ = 0;
    }
    i = cur_i_;
    return cur_.get();
  }
  void advance(size_t n) { cur_i_ += n; }
  FragmentBatchPtr take() { cur_i_ = 0; return std::move(cur_); }
  bool whole_chunks() const { return parallel_; }  // chunks are batch-sized (the segments of a plain file)
};

// All fragments of a list of input files (or of pairs of files), in file order.
class FragmentSource {
  std::vector<std::string> files_;
  bool paired_;
  size_t next_file_ = 0;
  std::unique_ptr<AsyncRecordStream> s1_, s2_;
  bool joined_ = false;  // the rest of the mate file has been loaded into mates_ (its order differs from the first file's)
  // Paired input is an inner join on the header in the reference (InputReader.scala:104-119): a header that occurs twice in
  // one of the files multiplies.  The walk below pairs every header once.  Repeats among the fragments it makes are noticed
  // where all fragment titles are (OutputSink); what it must report itself is the records that became NO fragment -- a record of
  // file 1 without a partner left, records of file 2 that nobody claimed or that repeat a header of file 2: if their header is
  // also the title of a fragment of the run, the reference's join has more products for it than this walk made (titles.hpp).
  bool done_ = false;
  RepeatedTitles *rep_;
  struct Mate { std::string seq; bool claimed = false; };
  std::unordered_map<std::string, Mate> mates_;
  void unclaimed_mates() {
    if (rep_) for (auto &kv : mates_) if (!kv.second.claimed) rep_->add_unmatched(title_hash(kv.first));
  }

  bool open_next() {
    if (next_file_ >= files_.size()) return false;
    s1_ = std::make_unique<AsyncRecordStream>(files_[next_file_]);
    if (paired_) s2_ = std::make_unique<AsyncRecordStream>(files_[next_file_ + 1]);
    next_file_ += paired_ ? 2 : 1;
    unclaimed_mates();
    joined_ = false;
    mates_.clear();
    return true;
  }

 public:
  FragmentSource(std::vector<std::string> files, bool paired, RepeatedTitles *rep = nullptr)
      : files_(std::move(files)), paired_(paired), rep_(rep) {}

  // Appends up to max_fragments (and about max_bases) to b; false when every file is exhausted and nothing was added.
  // Records move a run at a time (one copy of the bases of the run, not one per record).
  bool fill(FragmentBatchPtr &bp, size_t max_fragments, size_t max_bases) {
    if (!bp) bp = new_fragment_batch();
    bp->paired = paired_;
    size_t added = 0;
    while (added < max_fragments && bp->bases.size() + bp->mate_bases.size() < max_bases) {
      if (!s1_ && !open_next()) { if (!done_) { unclaimed_mates(); mates_.clear(); done_ = true; } break; }
      size_t i1 = 0, i2 = 0;
      const FragmentBatch *c1 = s1_->current(i1);
      if (!c1) { s1_.reset(); s2_.reset(); continue; }
      if (!paired_) {
        if (bp->size() == 0 && i1 == 0 && s1_->whole_chunks()) {  // a parsed segment of a plain file is a batch as it stands
          bp = s1_->take();
          bp->paired = false;
          return true;
        }
        const size_t n = std::min(c1->size() - i1, max_fragments - added);
        bp->append(*c1, i1, n, nullptr);
        s1_->advance(n);
        added += n;
        continue;
      }
      // Paired: the reference joins the two files on the header.  While both files list their records in the same order
      // the join is a lockstep walk; at the first disagreement the rest of the mate file becomes the join's hash side.
      if (!joined_) {
        const FragmentBatch *c2 = s2_->current(i2);
        if (c2) {
          const size_t n = std::min({c1->size() - i1, c2->size() - i2, max_fragments - added});
          size_t ok = 0;
          while (ok < n && remove_suffix(c1->title(i1 + ok), "/1") == remove_suffix(c2->title(i2 + ok), "/2")) ok++;
          if (ok) {
            bp->append(*c1, i1, ok, "/1");
            bp->append_mates(*c2, i2, ok);
            s1_->advance(ok);
            s2_->advance(ok);
            added += ok;
          }
          if (ok == n) continue;
        }
        std::string_view h2, m;
        while (s2_->next(h2, m)) {
          auto ins = mates_.emplace(std::string(remove_suffix(h2, "/2")), Mate{std::string(m), false});
          if (!ins.second && rep_) rep_->add(title_hash(ins.first->first));   // the header repeats inside file 2
        }
        joined_ = true;
        continue;
      }
      std::string_view h, sq;
      s1_->next(h, sq);
      h = remove_suffix(h, "/1");
      auto it = mates_.find(std::string(h));
      if (it == mates_.end()) {  // inner join: no mate, no fragment
        if (rep_) rep_->add_unmatched(title_hash(h));
        continue;
      }
      it->second.claimed = true;
      std::string_view m(it->second.seq);
      bp->add(h, sq, &m);
      added++;
    }
    return added > 0;
  }
};

// Reads batches on its own thread, at most `depth` ahead of the consumer.
class BatchPrefetcher {
  FragmentSource src_;
  size_t max_fragments_, max_bases_, depth_;
  std::deque<FragmentBatchPtr> q_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool done_ = false, stop_ = false;
  std::string error_;
  std::thread th_;

  void run() {
    try {
      for (;;) {
        auto b = new_fragment_batch();
        if (!src_.fill(b, max_fragments_, max_bases_)) break;
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
  BatchPrefetcher(std::vector<std::string> files, bool paired, RepeatedTitles *rep = nullptr, size_t max_fragments = (size_t)1 << 17,
                  size_t max_bases = (size_t)512 << 20, size_t depth = 2)
      : src_(std::move(files), paired, rep), max_fragments_(max_fragments), max_bases_(max_bases), depth_(depth), th_([this] { run(); }) {}
  ~BatchPrefetcher() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; cv_.notify_all(); }
    th_.join();
  }
  FragmentBatchPtr next() {  // nullptr at the end of the input
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

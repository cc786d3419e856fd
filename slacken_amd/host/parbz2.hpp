// parbz2.hpp -- one bzip2 FILE decompressed on several threads (C++17, header only; the blocks themselves are decoded by the
// system's libbz2, resolved at run time as in seqio.hpp -- this image has no bzlib.h).
//
// Hadoop splits bzip2 input (BZip2Codec is a SplittableCompressionCodec), so the reference reads a .bz2 file on as many tasks as it
// has splits (FileInputs.scala:64-85 accepts .fq.bz2 / .fastq.bz2 / .bz2).  A bzip2 stream is a sequence of independent blocks,
// each introduced by the 48-bit magic 0x314159265359 at ANY bit position and closed by the next one or by the end-of-stream magic
// 0x177245385090; a block carries its own CRC.  So: the file is cut into chunks, a worker scans its chunk for block magics bit by
// bit, and for every block that STARTS in its chunk builds a one-block stream in memory -- "BZh9", the block's bits shifted to a
// byte boundary, the end-of-stream magic, the block's CRC as the stream's -- and hands it to BZ2_bzBuffToBuffDecompress.  (A
// false magic inside compressed data has probability 2^-48 per bit position -- about 0.003 per 100 GB of file -- and fails loudly, on
// the block's CRC; SLK_GZ_THREADS=0 reads such a file through libbz2's own streaming decoder.)  Concatenated streams are just more
// blocks.  read() hands the text out in file order, and checks on the way what a serial decoder checks between the blocks: every
// stream's COMBINED CRC (the 32 bits behind its end-of-stream magic against the fold of its blocks' CRCs, c = rotl(c, 1) ^ crc),
// that a stream header stands wherever a stream starts, that the file ends behind an end-of-stream marker and nowhere else -- a
// file cut at a block boundary, a stream with a block missing and bytes behind the last stream are errors here as they are for
// libbz2's streaming decoder (the other route), and so is an empty file.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace slk {
namespace parbz2 {

constexpr uint64_t BLOCK_MAGIC = 0x314159265359ull, EOS_MAGIC = 0x177245385090ull, MASK48 = (1ull << 48) - 1;

class Reader {
  typedef int (*decompress_t)(char *, unsigned int *, char *, unsigned int, int, int);
  struct Magic {            // a block or end-of-stream marker found at bit position `bit`, with the 32 bits behind it
    uint64_t bit;
    uint32_t crc;           // the block's CRC / the stream's combined CRC
    bool is_block;
  };
  struct Slot {
    std::vector<uint8_t> out;
    std::vector<Magic> magics;   // those that START in this chunk, in file order
    bool done = false;
  };
  const uint8_t *p_ = nullptr;
  size_t n_ = 0;
  int fd_ = -1;
  size_t chunk_bytes_, nchunks_ = 0, lookahead_;
  decompress_t decompress_ = nullptr;
  std::vector<Slot> slots_;
  std::mutex mu_;
  std::condition_variable cv_;
  size_t next_claim_ = 0, consumed_ = 0;
  bool stop_ = false;
  std::string error_;
  std::vector<std::thread> th_;
  size_t out_chunk_ = 0, out_off_ = 0;
  // consumer side: the stream structure, checked chunk by chunk in file order
  bool entered_ = false;        // the current chunk's markers have been checked
  bool in_stream_ = false;      // between a stream header and its end-of-stream marker
  uint32_t combined_ = 0;       // fold of the block CRCs of the current stream
  uint64_t header_byte_ = 0;    // where the next stream header must stand (valid when !in_stream_)
  bool finished_ = false;

  // bit i of the file (most significant bit of a byte first, as bzip2 writes them)
  inline uint64_t bits48_at(uint64_t bit) const {   // the 48 bits starting at `bit` (zero beyond the end)
    uint64_t v = 0;
    const size_t b = (size_t)(bit >> 3);
    for (int i = 0; i < 8; i++) v = (v << 8) | (b + i < n_ ? p_[b + i] : 0);
    return (v >> (16 - (bit & 7))) & MASK48;
  }
  // the next block or end-of-stream magic at or after `from` and before `limit` (bit positions); limit if none
  uint64_t next_magic(uint64_t from, uint64_t limit, bool *is_block) const {
    // a rolling window over the bits: the window's low 48 bits are compared after every bit
    size_t byte = (size_t)(from >> 3);
    uint64_t w = 0;
    int have = 0;   // bits in w that count
    uint64_t pos = (uint64_t)byte * 8;   // bit position just after the last bit shifted in
    while (byte < n_) {
      const uint8_t c = p_[byte++];
      for (int k = 7; k >= 0; k--) {
        w = (w << 1) | ((c >> k) & 1);
        pos++;
        have = have < 48 ? have + 1 : 48;
        if (have == 48 && pos - 48 >= from) {
          const uint64_t v = w & MASK48;
          if (v == BLOCK_MAGIC || v == EOS_MAGIC) {
            if (pos - 48 >= limit) return limit;
            *is_block = v == BLOCK_MAGIC;
            return pos - 48;
          }
        }
      }
      if (pos >= limit + 48) return limit;
    }
    return limit;
  }

  void decode_block(uint64_t start, uint64_t end, std::vector<uint8_t> &out, std::vector<uint8_t> &tmp) {
    // one-block stream: "BZh9" + bits [start, end) + EOS magic + the block's CRC (the 32 bits behind its magic) + padding
    const uint64_t nbits = end - start;
    const size_t nbytes = 4 + (size_t)((nbits + 48 + 32 + 7) / 8) + 8;
    tmp.assign(nbytes, 0);
    memcpy(tmp.data(), "BZh9", 4);
    uint64_t wpos = 32;   // bit position in tmp
    auto put_bits = [&](uint64_t v, int k) {   // k <= 56
      for (int i = k - 1; i >= 0; i--, wpos++)
        if ((v >> i) & 1) tmp[(size_t)(wpos >> 3)] |= (uint8_t)(0x80u >> (wpos & 7));
    };
    // the block's bits, a byte at a time once the source is read at its shift
    const int sh = (int)(start & 7);
    size_t sb = (size_t)(start >> 3);
    uint64_t left = nbits;
    uint8_t *dst = tmp.data() + 4;
    while (left >= 8) {
      const uint8_t a = p_[sb], b = sb + 1 < n_ ? p_[sb + 1] : 0;
      *dst++ = (uint8_t)((a << sh) | (sh ? b >> (8 - sh) : 0));
      sb++;
      left -= 8;
    }
    wpos = 32 + (nbits - left);
    if (left) {
      const uint8_t a = p_[sb], b = sb + 1 < n_ ? p_[sb + 1] : 0;
      const uint8_t v = (uint8_t)((a << sh) | (sh ? b >> (8 - sh) : 0));
      put_bits(v >> (8 - left), (int)left);
    }
    const uint64_t crc = (bits48_at(start + 48) >> 16) & 0xFFFFFFFFull;
    put_bits(EOS_MAGIC, 48);
    put_bits(crc, 32);
    const size_t src_len = (size_t)((wpos + 7) / 8);
    // a block holds up to 900 000 bytes AFTER the first run-length stage, which can stand for ~51 times as many
    size_t cap = std::max<size_t>(out.size() + 1024, out.size() + 1200000);
    for (;;) {
      const size_t at = out.size();
      out.resize(cap);
      unsigned int got = (unsigned int)std::min<size_t>(cap - at, 0xFFFFFFF0u);
      const int rc = decompress_((char *)out.data() + at, &got, (char *)tmp.data(), (unsigned int)src_len, 0, 0);
      if (rc == 0) { out.resize(at + got); return; }
      out.resize(at);
      if (rc != -8 /* BZ_OUTBUFF_FULL */) throw std::runtime_error("corrupt bzip2 block");
      cap = at + (cap - at) * 4;
      if (cap - at > ((size_t)1 << 30)) throw std::runtime_error("corrupt bzip2 block");
    }
  }

  void worker() {
    std::vector<uint8_t> tmp;
    try {
      for (;;) {
        size_t i;
        {
          std::unique_lock<std::mutex> lk(mu_);
          cv_.wait(lk, [&] { return stop_ || next_claim_ >= nchunks_ || next_claim_ < consumed_ + lookahead_; });
          if (stop_ || next_claim_ >= nchunks_) return;
          i = next_claim_++;
          slots_[i % lookahead_].done = false;
        }
        Slot &s = slots_[i % lookahead_];
        s.out.clear();
        s.magics.clear();
        const uint64_t lo = (uint64_t)std::min(n_, i * chunk_bytes_) * 8, hi = (uint64_t)std::min(n_, (i + 1) * chunk_bytes_) * 8;
        const uint64_t file_end = (uint64_t)n_ * 8;
        bool is_block = false;
        uint64_t at = next_magic(lo, hi, &is_block);
        while (at < hi) {
          bool next_is_block = false;
          const uint64_t next = next_magic(at + 48, file_end, &next_is_block);   // the block runs to the next magic (or the end)
          s.magics.push_back(Magic{at, (uint32_t)((bits48_at(at + 48) >> 16) & 0xFFFFFFFFull), is_block});
          if (is_block) decode_block(at, next, s.out, tmp);
          if (next >= hi) break;
          at = next;
          is_block = next_is_block;
        }
        std::lock_guard<std::mutex> lk(mu_);
        s.done = true;
        cv_.notify_all();
      }
    } catch (const std::exception &e) {
      std::lock_guard<std::mutex> lk(mu_);
      if (error_.empty()) error_ = e.what();
      stop_ = true;
      cv_.notify_all();
    }
  }

 public:
  Reader(const std::string &path, int threads, size_t chunk_bytes) : chunk_bytes_(std::max<size_t>(chunk_bytes, 64)) {
    void *h = dlopen("libbz2.so.1", RTLD_NOW);
    if (!h) h = dlopen("libbz2.so.1.0", RTLD_NOW);
    if (!h) throw std::runtime_error("bzip2 input needs libbz2.so.1: " + path);
    decompress_ = (decompress_t)dlsym(h, "BZ2_bzBuffToBuffDecompress");
    if (!decompress_) throw std::runtime_error("libbz2 lacks BZ2_bzBuffToBuffDecompress");
    fd_ = open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw std::runtime_error("cannot open " + path);
    struct stat sb;
    if (fstat(fd_, &sb) != 0) { ::close(fd_); throw std::runtime_error("cannot stat " + path); }
    n_ = (size_t)sb.st_size;
    if (n_) {
      void *m = mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd_, 0);
      if (m == MAP_FAILED) { ::close(fd_); throw std::runtime_error("cannot map " + path); }
      madvise(m, n_, MADV_SEQUENTIAL);
      p_ = (const uint8_t *)m;
    }
    if (n_ < 4 || p_[0] != 'B' || p_[1] != 'Z' || p_[2] != 'h' || p_[3] < '1' || p_[3] > '9') {   // (an empty file too)
      if (p_) munmap((void *)p_, n_);
      ::close(fd_);
      throw std::runtime_error("read error (corrupt compressed input?): not a bzip2 file: " + path);
    }
    nchunks_ = (n_ + chunk_bytes_ - 1) / chunk_bytes_;
    const size_t nt = std::min<size_t>((size_t)std::max(1, threads), std::max<size_t>(1, nchunks_));
    lookahead_ = nt + 3;
    slots_.resize(lookahead_);
    th_.reserve(nt);
    for (size_t t = 0; t < nt; t++) th_.emplace_back([this] { worker(); });
  }
  Reader(const Reader &) = delete;
  ~Reader() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; cv_.notify_all(); }
    for (auto &t : th_) t.join();
    if (p_) munmap((void *)p_, n_);
    if (fd_ >= 0) ::close(fd_);
  }

  // what a serial decoder checks between the blocks (header comment), for the markers of one chunk
  void check_structure(const std::vector<Magic> &magics) {
    auto bad = [](const char *what) { throw std::runtime_error(std::string("read error (corrupt compressed input?): ") + what); };
    for (const Magic &m : magics) {
      if (!in_stream_) {   // a stream starts: "BZh1".."BZh9" at header_byte_, its first marker right behind
        const size_t h = (size_t)header_byte_;
        if (m.bit != (uint64_t)h * 8 + 32 || h + 4 > n_ || p_[h] != 'B' || p_[h + 1] != 'Z' || p_[h + 2] != 'h' || p_[h + 3] < '1' || p_[h + 3] > '9')
          bad("bzip2 data (bytes that are no bzip2 stream)");
        in_stream_ = true;
        combined_ = 0;
      }
      if (m.is_block) {
        combined_ = ((combined_ << 1) | (combined_ >> 31)) ^ m.crc;
      } else {
        if (combined_ != m.crc) bad("bzip2 data (a stream's combined CRC does not match its blocks)");
        in_stream_ = false;
        header_byte_ = (m.bit + 48 + 32 + 7) / 8;
      }
    }
  }
  void check_end() {
    if (finished_) return;
    finished_ = true;
    if (in_stream_) throw std::runtime_error("read error (corrupt compressed input?): unexpected end of the bzip2 data");
    if (header_byte_ != (uint64_t)n_) throw std::runtime_error("read error (corrupt compressed input?): bzip2 data (bytes behind the last stream)");
  }

  // the next bytes of the decompressed file; 0 at its end
  size_t read(char *dst, size_t cap) {
    size_t got = 0;
    while (got < cap && out_chunk_ < nchunks_) {
      Slot &s = slots_[out_chunk_ % lookahead_];
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return !error_.empty() || (next_claim_ > out_chunk_ && s.done); });
        if (!error_.empty()) throw std::runtime_error("read error (corrupt compressed input?): " + error_);
      }
      if (!entered_) { check_structure(s.magics); entered_ = true; }
      const size_t k = std::min(cap - got, s.out.size() - out_off_);
      memcpy(dst + got, s.out.data() + out_off_, k);
      got += k;
      out_off_ += k;
      if (out_off_ >= s.out.size()) {
        std::lock_guard<std::mutex> lk(mu_);
        out_chunk_++;
        out_off_ = 0;
        entered_ = false;
        consumed_ = out_chunk_;
        cv_.notify_all();
      }
    }
    if (out_chunk_ >= nchunks_) check_end();
    return got;
  }
};

}  // namespace parbz2
}  // namespace slk

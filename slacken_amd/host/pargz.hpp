// pargz.hpp -- one gzip FILE inflated on several threads (C++17, header only; zlib only for crc32).
//
// The reference reads .gz input through Hadoop's gzip codec, one thread per file (FileInputs.scala:64-85 picks the reader by file
// name; a gzip file is not splittable).  Here the inflate thread of a file was what bounded `slacken-amd classify` on gzip input:
// zlib delivers ~0.45 GB/s of text, 1.4 M reads/s, while the kernels behind it take 1 100 M.
//
// A deflate stream can be entered in the middle without what came before -- at a block boundary, if one accepts that
// back-references may reach into the 32 KiB of output that precede the entry point.  So (the scheme of pugz / rapidgzip):
//   * the compressed file is cut into chunks; a worker looks for the first boundary in its chunk -- a dynamic-Huffman block
//     header (found by testing bit positions: the header has to describe two complete prefix codes, and the block behind it has
//     to decode) or the header of a further gzip member (BGZF files are thousands of small members) -- and decodes from there to
//     the first boundary at or beyond the end of its chunk, into 16-bit symbols: a byte, or a MARKER naming a position of the
//     unknown 32 KiB window;
//   * the chunks are chained in file order: if the decoder of chunk i - 1 stopped exactly where chunk i was entered, chunk i's
//     symbols are what a serial inflate would have produced, and its markers are replaced from the last 32 KiB of chunk i - 1
//     (the tail first, so that chunk i + 1 can follow at once); otherwise -- no boundary found, a false one, or a block that
//     spans several chunks -- the worker decodes its chunk again from where its predecessor stopped, with the window known.  No
//     guess survives unchecked: the chain only accepts positions the true decode reaches.
//   * per member CRC-32 and ISIZE are verified as zlib's gzread does (parts per chunk, crc32_combine in file order).
// read() hands the bytes out in file order (memory: `lookahead` chunks in flight).  In REGION mode the workers write the text at its
// final offset into one reserved stretch of address space instead, and the consumers -- the segment parsers of seqio.hpp --
// treat it like a mapped plain file that fills while they read it (wait_segment / wait_more / segment_parsed).
#pragma once
#include <zlib.h>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace slk {
namespace pargz {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
struct ReserveError : std::runtime_error { using std::runtime_error::runtime_error; };   // region mode: no address space (not the file's fault)

// ---- bits -------------------------------------------------------------------------------------------------------------------
struct BitReader {
  const uint8_t *p = nullptr;
  size_t n = 0;
  uint64_t buf = 0;
  int cnt = 0;       // valid bits in buf (negative after reading past the end)
  size_t next = 0;   // next byte to load
  void seek(uint64_t bit) {
    next = (size_t)(bit >> 3);
    buf = 0; cnt = 0;
    refill();
    const int s = (int)(bit & 7);
    buf >>= s; cnt -= s;
  }
  inline void refill() {
    if (next + 8 <= n) {   // (bits above cnt are those of the bytes at `next`: OR-ing them in again is idempotent)
      uint64_t w;
      memcpy(&w, p + next, 8);
      buf |= w << cnt;
      const int adv = (63 - cnt) >> 3;
      next += (size_t)adv;
      cnt += adv << 3;
    } else {
      while (cnt <= 56 && next < n) { buf |= (uint64_t)p[next++] << cnt; cnt += 8; }
    }
  }
  inline uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1)); }
  inline void drop(int k) { buf >>= k; cnt -= k; }
  inline uint32_t take(int k) { uint32_t v = peek(k); drop(k); return v; }
  uint64_t position() const { return (uint64_t)next * 8 - (uint64_t)cnt; }
  bool overrun() const { return cnt < 0; }
};

// ---- prefix codes -----------------------------------------------------------------------------------------------------------
// entry: bits 0..7 code length (for a link to a second-level table: its index width), 8..15 kind, 16..31 value
enum : uint32_t { K_INVALID = 0, K_LIT = 1, K_LEN = 2, K_EOB = 3, K_SUB = 4, K_DIST = 5 };
inline uint32_t mk(uint32_t len, uint32_t kind, uint32_t val) { return len | (kind << 8) | (val << 16); }
inline uint32_t e_len(uint32_t e) { return e & 255; }
inline uint32_t e_kind(uint32_t e) { return (e >> 8) & 255; }
inline uint32_t e_val(uint32_t e) { return e >> 16; }

constexpr int LIT_P = 10, DIST_P = 8;
static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t bitrev(uint32_t v, int len) {
  uint32_t r = 0;
  for (int i = 0; i < len; i++) { r = (r << 1) | (v & 1); v >>= 1; }
  return r;
}

// Kraft sum of the lengths in units of 2^-15: 32768 = complete; symbols = number of lengths != 0
inline uint32_t kraft(const uint8_t *lens, int n, int *symbols) {
  uint32_t s = 0;
  int c = 0;
  for (int i = 0; i < n; i++) if (lens[i]) { s += 1u << (15 - lens[i]); c++; }
  *symbols = c;
  return s;
}

struct Tables {
  uint32_t lit[(1 << LIT_P) + 288 * 32];
  uint32_t dist[(1 << DIST_P) + 32 * 128];
};

// canonical code of `lens` into a two-level table of primary width P; entry of symbol s from `entry(s, len)`.  The caller has
// checked that the code is not over-subscribed; slots no code reaches stay K_INVALID.
template <class F>
inline void build_table(uint32_t *tab, size_t cap, int P, const uint8_t *lens, int n, F entry) {
  uint16_t count[16] = {0}, nextc[16];
  for (int i = 0; i < n; i++) count[lens[i]]++;
  count[0] = 0;
  uint32_t code = 0;
  for (int l = 1; l <= 15; l++) { code = (code + count[l - 1]) << 1; nextc[l] = (uint16_t)code; }
  const uint32_t psize = 1u << P;
  std::fill(tab, tab + psize, mk(0, K_INVALID, 0));
  uint8_t sub_bits[1 << LIT_P];   // per primary slot: width of its second-level table (0: none)
  memset(sub_bits, 0, psize);
  uint32_t rev[288];
  for (int s = 0; s < n; s++) {
    const int l = lens[s];
    if (!l) continue;
    rev[s] = bitrev(nextc[l]++, l);
    if (l <= P) {
      const uint32_t e = entry(s, l);
      for (uint32_t i = rev[s]; i < psize; i += 1u << l) tab[i] = e;
    } else {
      uint8_t &b = sub_bits[rev[s] & (psize - 1)];
      b = (uint8_t)std::max<int>(b, l - P);
    }
  }
  size_t at = psize;
  for (uint32_t i = 0; i < psize; i++) {
    if (!sub_bits[i]) continue;
    const size_t sz = (size_t)1 << sub_bits[i];
    if (at + sz > cap) throw Error("prefix code table overflow");
    std::fill(tab + at, tab + at + sz, mk(0, K_INVALID, 0));
    tab[i] = mk(sub_bits[i], K_SUB, (uint32_t)at);
    at += sz;
  }
  for (int s = 0; s < n; s++) {
    const int l = lens[s];
    if (l <= P) continue;
    const uint32_t link = tab[rev[s] & (psize - 1)];
    const uint32_t e = entry(s, l);
    uint32_t *sub = tab + e_val(link);
    for (uint32_t i = rev[s] >> P; i < (1u << e_len(link)); i += 1u << (l - P)) sub[i] = e;
  }
}

inline uint32_t lit_entry(int s, int l) {
  if (s < 256) return mk((uint32_t)l, K_LIT, (uint32_t)s);
  if (s == 256) return mk((uint32_t)l, K_EOB, 0);
  if (s <= 285) return mk((uint32_t)l, K_LEN, (uint32_t)(s - 257));
  return mk((uint32_t)l, K_INVALID, 0);
}
inline uint32_t dist_entry(int s, int l) { return s < 30 ? mk((uint32_t)l, K_DIST, (uint32_t)s) : mk((uint32_t)l, K_INVALID, 0); }

inline void build_fixed(Tables &T) {
  uint8_t l[288];
  for (int i = 0; i < 144; i++) l[i] = 8;
  for (int i = 144; i < 256; i++) l[i] = 9;
  for (int i = 256; i < 280; i++) l[i] = 7;
  for (int i = 280; i < 288; i++) l[i] = 8;
  build_table(T.lit, sizeof(T.lit) / 4, LIT_P, l, 288, lit_entry);
  uint8_t d[32];
  for (int i = 0; i < 32; i++) d[i] = 5;
  build_table(T.dist, sizeof(T.dist) / 4, DIST_P, d, 32, dist_entry);
}

// The header of a dynamic block (RFC 1951, 3.2.7) behind BFINAL/BTYPE -> code lengths.  strict: what the boundary search asks of
// a candidate (both codes complete, as every compressor writes them; an end-of-block code).  Otherwise what zlib accepts.
inline bool read_dynamic_header(BitReader &br, uint8_t *lens /* 320 */, int &hlit, int &hdist, bool strict) {
  static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  br.refill();
  hlit = (int)br.take(5) + 257;
  hdist = (int)br.take(5) + 1;
  const int hclen = (int)br.take(4) + 4;
  if (hlit > 286 || hdist > 30) return false;
  uint8_t cl[19] = {0};
  br.refill();
  for (int i = 0; i < hclen; i++) {
    if (i == 12) br.refill();
    cl[ORDER[i]] = (uint8_t)br.take(3);
  }
  if (br.overrun()) return false;
  int syms;
  const uint32_t ks = kraft(cl, 19, &syms);
  if (ks != 32768u) return false;   // (zlib too: an incomplete code-length code is an error)
  (void)syms;
  // 7-bit single-level table
  uint16_t tab[128];
  {
    uint16_t count[8] = {0}, nextc[8];
    for (int i = 0; i < 19; i++) count[cl[i]]++;
    count[0] = 0;
    uint32_t code = 0;
    for (int l = 1; l <= 7; l++) { code = (code + count[l - 1]) << 1; nextc[l] = (uint16_t)code; }
    for (int i = 0; i < 128; i++) tab[i] = 0xFFFF;
    for (int s = 0; s < 19; s++) {
      const int l = cl[s];
      if (!l) continue;
      const uint32_t r = bitrev(nextc[l]++, l);
      for (uint32_t i = r; i < 128; i += 1u << l) tab[i] = (uint16_t)(s | (l << 8));
    }
  }
  const int total = hlit + hdist;
  int i = 0;
  while (i < total) {
    br.refill();
    const uint16_t e = tab[br.peek(7)];
    if (e == 0xFFFF) return false;
    br.drop(e >> 8);
    const int s = e & 255;
    if (s < 16) { lens[i++] = (uint8_t)s; continue; }
    int rep, v = 0;
    if (s == 16) { if (i == 0) return false; v = lens[i - 1]; rep = 3 + (int)br.take(2); }
    else if (s == 17) rep = 3 + (int)br.take(3);
    else rep = 11 + (int)br.take(7);
    if (i + rep > total) return false;
    while (rep--) lens[i++] = (uint8_t)v;
  }
  if (br.overrun()) return false;
  if (lens[256] == 0) return false;   // no end-of-block code
  int ls, ds;
  const uint32_t kl = kraft(lens, hlit, &ls), kd = kraft(lens + hlit, hdist, &ds);
  if (kl > 32768u || kd > 32768u) return false;
  if (kl != 32768u && (strict || ls != 1)) return false;
  if (kd != 32768u && ds > 1) return false;             // (no distance code, or a single one, may be incomplete)
  return true;
}

// ---- gzip framing -----------------------------------------------------------------------------------------------------------
// the member header at byte b (RFC 1952); true and b = first byte of the deflate data, or false (not a header / cut short)
inline bool parse_gzip_header(const uint8_t *p, size_t n, size_t &b) {
  size_t i = b;
  if (i + 10 > n || p[i] != 0x1f || p[i + 1] != 0x8b || p[i + 2] != 8 || (p[i + 3] & 0xE0)) return false;
  const int flg = p[i + 3];
  i += 10;
  if (flg & 4) {
    if (i + 2 > n) return false;
    const size_t xlen = p[i] | (p[i + 1] << 8);
    i += 2 + xlen;
    if (i > n) return false;
  }
  for (int f = 8; f <= 16; f <<= 1)   // FNAME, FCOMMENT: zero-terminated
    if (flg & f) {
      while (i < n && p[i]) i++;
      if (i >= n) return false;
      i++;
    }
  if (flg & 2) i += 2;
  if (i > n) return false;
  b = i;
  return true;
}

struct Boundary {
  uint64_t bit = 0;
  bool member = false;   // a gzip member header starts here (bit = 8 * byte), not a deflate block
  bool operator==(const Boundary &o) const { return bit == o.bit && member == o.member; }
};
struct MemberEnd { uint64_t out_pos; uint32_t crc, isize; };

// ---- the decoder ------------------------------------------------------------------------------------------------------------
// T = uint16_t: speculative (window unknown: sources before the chunk become markers 256 + position in the 32 KiB window);
// T = uint8_t: the window is known (`win`, its last `win_len` bytes valid).
template <class T>
struct Inflater {
  static constexpr bool SPEC = sizeof(T) == 2;
  BitReader br;
  std::vector<T> &out;
  size_t pos = 0;             // symbols written
  int64_t lowest;             // smallest source position a back-reference may name (member start, or -window)
  const uint8_t *win = nullptr;   // known window: byte j of the 32 KiB before the chunk at win[j]
  std::vector<MemberEnd> &ends;
  Tables &tab;                // scratch (dynamic blocks)
  const Tables &fixed;
  size_t max_out;             // speculation gives up beyond this
  int blocks_done = 0;
  size_t marker_end = 0;      // speculative: no marker at or beyond this position (they die out as the matches move on)

  Inflater(const uint8_t *p, size_t n, std::vector<T> &o, std::vector<MemberEnd> &e, Tables &t, const Tables &f, size_t cap)
      : out(o), ends(e), tab(t), fixed(f), max_out(cap) { br.p = p; br.n = n; lowest = 0; }

  inline void room(size_t k) {
    if (pos + k > out.size()) {
      if (pos + k > max_out) throw Error("speculative output limit");
      out.resize(std::max(out.size() * 2, pos + k + (1u << 16)));
    }
  }
  // a match whose source begins before the chunk: markers (speculative) or the known window
  inline void copy_before(T *o, size_t &ps, uint32_t length, int64_t s) {
    for (uint32_t k = 0; k < length; k++, ps++, s++) {
      if (s >= 0) o[ps] = o[s];
      else if (SPEC) o[ps] = (T)(256 + 32768 + s);
      else o[ps] = (T)win[32768 + s];
    }
  }
  void huffman_block(const Tables &t) {
    BitReader b = br;   // (locals: the loop keeps them in registers)
    T *o = out.data();
    size_t cap = out.size(), ps = pos;
    constexpr size_t SLACK = 300;   // a match of 258 and the overshoot of the 8-element copies
    for (;;) {
      if (ps + SLACK > cap) {
        pos = ps;
        room(SLACK);
        o = out.data();
        cap = out.size();
      }
      if (b.cnt < 0) throw Error("unexpected end of data");
      b.refill();
      uint32_t e = t.lit[b.buf & ((1u << LIT_P) - 1)];
      if (e_kind(e) == K_LIT) {   // up to three literals of the first-level table per refill (text is mostly literals)
        b.drop((int)e_len(e)); o[ps++] = (T)e_val(e);
        e = t.lit[b.buf & ((1u << LIT_P) - 1)];
        if (e_kind(e) == K_LIT) {
          b.drop((int)e_len(e)); o[ps++] = (T)e_val(e);
          e = t.lit[b.buf & ((1u << LIT_P) - 1)];
          if (e_kind(e) == K_LIT) { b.drop((int)e_len(e)); o[ps++] = (T)e_val(e); continue; }
        }
        b.refill();   // (leaves the bits already looked at where they are)
      }
      if (e_kind(e) == K_SUB) e = t.lit[e_val(e) + ((uint32_t)(b.buf >> LIT_P) & ((1u << e_len(e)) - 1))];
      b.drop((int)e_len(e));
      const uint32_t kind = e_kind(e);
      if (kind == K_LEN) {
        const uint32_t li = e_val(e);
        const uint32_t length = LEN_BASE[li] + b.take(LEN_EXTRA[li]);
        uint32_t d = t.dist[b.buf & ((1u << DIST_P) - 1)];
        if (e_kind(d) == K_SUB) d = t.dist[e_val(d) + ((uint32_t)(b.buf >> DIST_P) & ((1u << e_len(d)) - 1))];
        if (e_kind(d) != K_DIST) throw Error("invalid distance code");
        b.drop((int)e_len(d));
        const uint32_t ds = e_val(d);
        const uint32_t distance = DIST_BASE[ds] + b.take(DIST_EXTRA[ds]);
        if (b.cnt < 0) throw Error("unexpected end of data");
        const int64_t src = (int64_t)ps - (int64_t)distance;
        if (src < lowest) throw Error("invalid distance");
        if (src < 0) { copy_before(o, ps, length, src); if (SPEC) marker_end = ps; continue; }
        if (SPEC && (size_t)src < marker_end) marker_end = ps + length;   // (it may copy markers)
        if (distance >= 8) {   // 8 elements at a time; may write up to 7 beyond the match (SLACK)
          const T *sp = o + src;
          T *dp = o + ps, *const end = dp + length;
          do { memcpy(dp, sp, 8 * sizeof(T)); dp += 8; sp += 8; } while (dp < end);
        } else {
          for (uint32_t k = 0; k < length; k++) o[ps + k] = o[ps + k - distance];
        }
        ps += length;
        continue;
      }
      if (kind == K_LIT) { o[ps++] = (T)e_val(e); continue; }   // (a literal with a long code)
      if (kind == K_EOB) {
        if (b.cnt < 0) throw Error("unexpected end of data");
        br = b;
        pos = ps;
        return;
      }
      throw Error("invalid literal/length code");
    }
  }
  // one block at the reader's position; true if it was the member's last
  bool block(bool strict_header) {
    br.refill();
    const bool final = br.take(1);
    const uint32_t type = br.take(2);
    if (type == 0) {
      br.drop(br.cnt & 7);
      br.refill();
      const uint32_t len = br.take(16), nlen = br.take(16);
      if (br.overrun() || (len ^ nlen) != 0xFFFF) throw Error("invalid stored block");
      size_t at = (size_t)(br.position() >> 3);
      if (at + len > br.n) throw Error("unexpected end of data");
      room(len);
      for (uint32_t i = 0; i < len; i++) out[pos++] = (T)br.p[at + i];
      br.seek((uint64_t)(at + len) * 8);
    } else if (type == 1) {
      huffman_block(fixed);
    } else if (type == 2) {
      uint8_t lens[320];
      int hlit, hdist;
      if (!read_dynamic_header(br, lens, hlit, hdist, strict_header)) throw Error("invalid dynamic block header");
      build_table(tab.lit, sizeof(tab.lit) / 4, LIT_P, lens, hlit, lit_entry);
      build_table(tab.dist, sizeof(tab.dist) / 4, DIST_P, lens + hlit, hdist, dist_entry);
      huffman_block(tab);
    } else {
      throw Error("invalid block type");
    }
    blocks_done++;
    return final;
  }
  // From boundary `from` to the first boundary at or beyond `stop_bit` (or the end of the gzip data).  -> where it stopped;
  // stream_end: nothing but (ignored) garbage or nothing at all follows.
  Boundary run(Boundary from, uint64_t stop_bit, bool &stream_end) {
    stream_end = false;
    Boundary at = from;
    bool first = true;
    for (;;) {
      if (!first && at.bit >= stop_bit) return at;
      if (at.member) {
        size_t b = (size_t)(at.bit >> 3);
        if (!parse_gzip_header(br.p, br.n, b)) {
          if (first && (SPEC || at.bit == 0)) throw Error("not a gzip header");   // a wrong guess, or no gzip file at all
          stream_end = true;   // (zlib's gzread ignores what follows the last member if it is no gzip header)
          return at;
        }
        lowest = (int64_t)pos;   // nothing before the member can be referenced
        br.seek((uint64_t)b * 8);
        at.member = false;
        at.bit = (uint64_t)b * 8;
        first = false;
        continue;   // (the data of a member start at a block boundary, which may already be beyond stop_bit)
      }
      br.seek(at.bit);
      const bool final = block(SPEC && blocks_done == 0);
      first = false;
      if (final) {
        br.drop(br.cnt & 7);
        size_t b = (size_t)(br.position() >> 3);
        if (b + 8 > br.n) throw Error("unexpected end of data (gzip trailer)");
        MemberEnd me;
        me.out_pos = pos;
        me.crc = (uint32_t)br.p[b] | ((uint32_t)br.p[b + 1] << 8) | ((uint32_t)br.p[b + 2] << 16) | ((uint32_t)br.p[b + 3] << 24);
        me.isize = (uint32_t)br.p[b + 4] | ((uint32_t)br.p[b + 5] << 8) | ((uint32_t)br.p[b + 6] << 16) | ((uint32_t)br.p[b + 7] << 24);
        ends.push_back(me);
        b += 8;
        at.member = true;
        at.bit = (uint64_t)b * 8;
        if (b >= br.n) { stream_end = true; return at; }
      } else {
        at.bit = br.position();
      }
    }
  }
};

// ---- the reader -------------------------------------------------------------------------------------------------------------
// CRC-32 of a buffer: libdeflate's (carry-less multiplication, several GB/s) where the system has the library -- this image
// ships libdeflate.so.0 without its header, so the one entry point is resolved at run time --, else zlib's (~0.8 GB/s)
inline uint32_t fast_crc32(const uint8_t *p, size_t n) {
  typedef uint32_t (*fn_t)(uint32_t, const void *, size_t);
  static const fn_t fn = [] {
    void *h = dlopen("libdeflate.so.0", RTLD_NOW);
    return h ? (fn_t)dlsym(h, "libdeflate_crc32") : (fn_t) nullptr;
  }();
  return fn ? fn(0, p, n) : (uint32_t)crc32_z(0L, p, n);
}

struct Timing {   // SLK_GZ_TIMING: where the workers' time goes (seconds summed over threads), printed when the reader closes
  std::atomic<uint64_t> search{0}, decode{0}, resolve{0}, redo{0}, crc{0}, wait{0}, marked{0}, total{0};
};
inline uint64_t now_ns() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

class Reader {
  struct Seg { size_t len; uint32_t crc; bool ends_member; uint32_t want_crc, want_isize; };
  struct Chunk {
    std::vector<uint16_t> spec;
    std::vector<uint8_t> bytes;
    size_t nbytes = 0;
    std::vector<MemberEnd> ends;
    std::vector<Seg> segs;
    bool found = false;
    Boundary F, specE;
    bool spec_end = false, spec_bytes = false;   // spec_bytes: the speculative output is in `bytes` already (entered at a member start)
    size_t spec_len = 0, marker_end = 0;
    bool done = false;
  };
  struct Chain {   // what chunk i + 1 needs of chunk i
    Boundary E;
    bool stream_end = false;
    std::array<uint8_t, 32768> win;
    size_t win_len = 0;
    uint64_t total = 0;   // inflated bytes up to and including this chunk
  };
  struct Info {   // region mode: what stays known of a chunk after its slot has moved on
    uint64_t begin = 0, end = 0;
    bool done = false, parsed = false;
    std::vector<Seg> segs;
  };

  const uint8_t *p_ = nullptr;
  size_t n_ = 0;
  int fd_ = -1;
  size_t chunk_bytes_, nchunks_ = 0, lookahead_;
  std::vector<Chunk> slots_;
  std::vector<Chain> chains_;   // ring, same indexing as slots_
  std::mutex mu_;
  std::condition_variable cv_;
  size_t next_claim_ = 0, chained_ = 0, consumed_ = 0;   // chunks claimed / with their chain published (a prefix) / handed out
  bool stop_ = false;
  std::string error_;
  std::vector<std::thread> th_;
  size_t nthreads_ = 1;   // (fixed before the first worker starts: they read it)
  Tables fixed_;
  Chain start_;
  Timing tm_;
  const bool timing_ = getenv("SLK_GZ_TIMING") != nullptr;
  // consumer state
  size_t out_chunk_ = 0, out_off_ = 0;
  uint32_t run_crc_ = 0;
  uint64_t run_len_ = 0;
  // region mode: the inflated file as ONE stretch of address space (reserved up front, made writable as the data arrive, given
  // back behind the consumers), so that the consumers can treat it like a mapped plain file
  const bool region_mode_;
  uint8_t *region_ = nullptr;
  size_t region_cap_ = 0, rw_end_ = 0, dropped_ = 0;
  std::vector<Info> info_;
  size_t verified_ = 0;          // chunks whose CRC parts are folded into the member checks (a prefix of the done ones)
  bool last_stream_end_ = false;
  uint64_t written_ = 0;         // inflated bytes up to the last chained chunk
  size_t span_ = 1;              // chunks a consumer takes as one segment
  int starving_ = 0;             // consumers waiting for data beyond the budget (a record longer than it)
  const uint64_t region_budget_ = (uint64_t)512 << 20;

  void ensure_writable(uint64_t upto) {
    std::lock_guard<std::mutex> lk(mu_);
    if (upto > region_cap_) throw Error("inflated data exceed the reserved address space");
    if (upto > rw_end_) {
      const size_t step = (size_t)64 << 20;
      const size_t to = std::min(region_cap_, (size_t)((upto + step - 1) / step * step));
      if (mprotect(region_ + rw_end_, to - rw_end_, PROT_READ | PROT_WRITE) != 0) throw Error("cannot make the inflate buffer writable");
      rw_end_ = to;
    }
  }
  // (mu_ held) fold the CRC parts of the chunks that have finished, in file order; member checks as gzread makes them
  void verify_prefix() {
    while (verified_ < nchunks_ && info_[verified_].done && error_.empty()) {
      for (const Seg &sg : info_[verified_].segs) {
        run_crc_ = (uint32_t)crc32_combine(run_crc_, sg.crc, (z_off_t)sg.len);
        run_len_ += sg.len;
        if (sg.ends_member) {
          if (run_crc_ != sg.want_crc || (uint32_t)run_len_ != sg.want_isize) error_ = "gzip CRC / length mismatch";
          run_crc_ = 0; run_len_ = 0;
        }
      }
      verified_++;
      if (verified_ == nchunks_ && error_.empty() && (!last_stream_end_ || run_len_ != 0)) error_ = "unexpected end of data";
    }
    if (!error_.empty()) { stop_ = true; cv_.notify_all(); }
  }

  // slots that may be taken again: the chunk is handed out and its successor has read its chain.  (Region mode: a FINISHED chunk
  // -- its worker may still be replacing markers when its successor has long chained -- lives in the region, not in its slot;
  // what bounds the inflaters there is the memory between them and the slowest consumer.)
  size_t released() const { return std::min(region_mode_ ? verified_ : consumed_, chained_ ? chained_ - 1 : 0); }
  bool may_claim() const {
    if (next_claim_ >= released() + lookahead_) return false;
    if (!region_mode_ || starving_ > 0 || next_claim_ < consumed_ + span_ + 2) return true;   // (the slowest consumer's own chunks: always)
    const uint64_t low = consumed_ ? info_[consumed_ - 1].end : 0;
    return written_ - low <= region_budget_;
  }
  uint64_t lo_bit(size_t i) const { return (uint64_t)std::min(n_, i * chunk_bytes_) * 8; }

  // the first boundary in [lo, hi) that a decoder accepts, decoded speculatively to the first boundary at or beyond `hi`
  void speculate(size_t i, Chunk &c, Tables &tab) {
    c.found = false;
    const uint64_t lo = lo_bit(i), hi = lo_bit(i + 1);
    const size_t cap = (size_t)24 * chunk_bytes_ + ((size_t)8 << 20);   // (text inflates 3-6 fold; what goes far beyond is left to the serial route)
    auto attempt = [&](Boundary from) {
      const uint64_t t0 = timing_ ? now_ns() : 0;
      struct Stop { Timing &tm; uint64_t t0; bool on; ~Stop() { if (on) tm.decode += now_ns() - t0; } } stopwatch{tm_, t0, timing_};
      c.ends.clear();
      if (from.member) {
        // a chunk entered where a gzip member starts has nothing unknown before it: bytes at once, no markers to replace
        // (every chunk of a BGZF file -- bgzip, bcl2fastq -- is such a chunk)
        Inflater<uint8_t> inf(p_, n_, c.bytes, c.ends, tab, fixed_, cap);
        try {
          c.specE = inf.run(from, hi, c.spec_end);
          c.spec_len = inf.pos;
          c.spec_bytes = true;
          c.F = from;
          c.found = true;
          return 2;
        } catch (const Error &) {
          return inf.blocks_done >= 2 ? 1 : 0;
        }
      }
      Inflater<uint16_t> inf(p_, n_, c.spec, c.ends, tab, fixed_, cap);
      inf.lowest = -32768;
      try {
        c.specE = inf.run(from, hi, c.spec_end);
        c.spec_len = inf.pos;
        c.spec_bytes = false;
        c.marker_end = std::min(inf.marker_end, inf.pos);
        c.F = from;
        c.found = true;
        return 2;
      } catch (const Error &) {
        return inf.blocks_done >= 2 ? 1 : 0;   // 1: it looked like a stream for a while: do not search on
      }
    };
    if (i == 0) {   // the file's first member: known
      Boundary b0;
      b0.bit = 0; b0.member = true;
      attempt(b0);
      return;
    }
    const uint8_t *p = p_;
    for (uint64_t byte = lo >> 3; byte < (hi >> 3); byte++) {
      // a member header?  (checked before the bit positions of this byte: it is the smaller position)
      if (p[byte] == 0x1f && byte + 10 <= n_ && p[byte + 1] == 0x8b && p[byte + 2] == 8 && !(p[byte + 3] & 0xE0)) {
        Boundary b;
        b.bit = byte * 8; b.member = true;
        const int r = attempt(b);
        if (r) return;
      }
      uint32_t w = 0;
      memcpy(&w, p + byte, std::min<size_t>(4, n_ - byte));
      for (int s = 0; s < 8; s++) {
        const uint32_t v = w >> s;
        if ((v & 7) != 4) continue;                          // BFINAL = 0, BTYPE = 10
        if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) continue;   // HLIT, HDIST
        BitReader br;
        br.p = p_; br.n = n_;
        br.seek(byte * 8 + s + 3);
        uint8_t lens[320];
        int hlit, hdist;
        if (!read_dynamic_header(br, lens, hlit, hdist, true)) continue;
        Boundary b;
        b.bit = byte * 8 + s; b.member = false;
        const int r = attempt(b);
        if (r) return;
      }
    }
  }

  static void window_after(const Chain &prev, const uint8_t *bytes, size_t len, size_t member_start /* npos: none in this chunk */,
                           Chain &out) {
    if (member_start != (size_t)-1) {
      const size_t have = std::min<size_t>(32768, len - member_start);
      memcpy(out.win.data() + 32768 - have, bytes + len - have, have);
      out.win_len = have;
    } else if (len >= 32768) {
      memcpy(out.win.data(), bytes + len - 32768, 32768);
      out.win_len = std::min<size_t>(32768, prev.win_len + len);
    } else {
      memmove(out.win.data(), prev.win.data() + len, 32768 - len);   // (out may be prev: chunks without output pass it on)
      memcpy(out.win.data() + 32768 - len, bytes, len);
      out.win_len = std::min<size_t>(32768, prev.win_len + len);
    }
  }

  void finish_segments(Chunk &c, const uint8_t *data) {   // CRC parts, cut at the member ends
    c.segs.clear();
    size_t a = 0;
    for (const MemberEnd &me : c.ends) {
      Seg s;
      s.len = (size_t)me.out_pos - a;
      s.crc = fast_crc32(data + a, s.len);
      s.ends_member = true; s.want_crc = me.crc; s.want_isize = me.isize;
      c.segs.push_back(s);
      a = (size_t)me.out_pos;
    }
    if (a < c.nbytes || c.segs.empty()) {
      Seg s;
      s.len = c.nbytes - a;
      s.crc = fast_crc32(data + a, s.len);
      s.ends_member = false; s.want_crc = 0; s.want_isize = 0;
      c.segs.push_back(s);
    }
  }

  void worker() {
    auto tab = std::make_unique<Tables>();
    try {
      for (;;) {
        size_t i;
        {
          std::unique_lock<std::mutex> lk(mu_);
          cv_.wait(lk, [&] { return stop_ || next_claim_ >= nchunks_ || may_claim(); });
          if (stop_ || next_claim_ >= nchunks_) return;
          i = next_claim_++;
          slots_[i % lookahead_].done = false;   // (under the lock: the consumer tells this chunk from the slot's last one by it)
        }
        Chunk &c = slots_[i % lookahead_];
        Chain &mine = chains_[i % lookahead_];
        c.nbytes = 0;
        c.ends.clear();
        if (nthreads_ == 1 && i > 0) {
          c.found = false;   // one worker: it would only guess at what it is about to know -- it decodes on from where it stopped
        } else {
          const uint64_t t0 = timing_ ? now_ns() : 0;
          speculate(i, c, *tab);
          if (timing_) tm_.search += now_ns() - t0;   // (includes the decode attempts: subtracted when printed)
        }
        const uint64_t t_wait = timing_ ? now_ns() : 0;
        const Chain *prev = &start_;   // (before chunk 0: the file's first member header, no window)
        if (i > 0) {
          std::unique_lock<std::mutex> lk(mu_);
          cv_.wait(lk, [&] { return stop_ || chained_ >= i; });
          if (stop_) return;
          prev = &chains_[(i - 1) % lookahead_];
        }
        if (timing_) tm_.wait += now_ns() - t_wait;
        const uint64_t t_fin = timing_ ? now_ns() : 0;
        bool redone = false;
        const uint64_t hi = lo_bit(i + 1);
        Chain next;
        bool early = false;   // chain already published
        auto publish = [&]() {
          std::lock_guard<std::mutex> lk(mu_);
          mine.E = next.E; mine.stream_end = next.stream_end; mine.win = next.win; mine.win_len = next.win_len;
          mine.total = next.total;
          written_ = next.total;
          chained_ = i + 1;
          cv_.notify_all();
        };
        if (prev->stream_end || prev->E.bit >= hi) {
          // the data ended before this chunk, or a block of the predecessor runs through all of it: nothing to add
          next.E = prev->E; next.stream_end = prev->stream_end; next.win = prev->win; next.win_len = prev->win_len;
          next.total = prev->total;
          c.ends.clear();
        } else if (c.found && c.F == prev->E && c.spec_bytes) {
          // entered at a member start and decoded to bytes: they are the truth as they stand
          const size_t len = c.spec_len;
          c.nbytes = len;
          next.total = prev->total + len;
          next.E = c.specE; next.stream_end = c.spec_end;
          const size_t member_start = c.ends.empty() ? 0 : (size_t)c.ends.back().out_pos;
          window_after(*prev, c.bytes.data(), len, member_start, next);
          const uint64_t at = prev->total;   // (read before the chain is published: the predecessor's slot may move on after that)
          publish();
          early = true;
          if (region_mode_) {
            ensure_writable(next.total);
            memcpy(region_ + at, c.bytes.data(), len);
          }
        } else if (c.found && c.F == prev->E) {
          // the speculative symbols are the truth: markers <- the predecessor's window, the tail first
          const size_t len = c.spec_len;
          if (region_mode_) ensure_writable(prev->total + len);
          else if (c.bytes.size() < len) c.bytes.resize(len);
          c.nbytes = len;
          next.total = prev->total + len;
          std::vector<uint8_t> lut(256 + 32768);
          for (int v = 0; v < 256; v++) lut[v] = (uint8_t)v;
          memcpy(lut.data() + 256, prev->win.data(), 32768);
          const size_t valid = prev->win_len;
          const uint16_t *sp = c.spec.data();
          uint8_t *by = region_mode_ ? region_ + prev->total : c.bytes.data();
          const uint16_t min_marker = (uint16_t)(256 + 32768 - valid);
          if (timing_) { tm_.marked += c.marker_end; tm_.total += len; }
          // 16 symbols at a time: all bytes -> narrowed; else through the table.  (On FASTQ most groups hold a marker for the
          // whole length of a chunk: the short matches of such text keep copying them forward.)  A marker can only be wrong --
          // name a byte before the start of the gzip member -- while the member is younger than 32 KiB: checked only then.
          const bool check = valid < 32768;
          auto resolve = [&](size_t a, size_t b) {
            size_t k = a;
            if (check) {
              uint16_t low = 0xFFFF;
              for (; k < b; k++) {
                const uint16_t v = sp[k];
                by[k] = lut[v];
                if (v >= 256 && v < low) low = v;
              }
              if (low < min_marker) throw Error("invalid distance (before the start of the gzip member)");
              return;
            }
            const uint8_t *const lt = lut.data();
#if defined(__SSE2__)
            const __m128i hi_bits = _mm_set1_epi16((short)0xFF00), zero = _mm_setzero_si128();
            for (; k + 16 <= b; k += 16) {
              const __m128i x = _mm_loadu_si128((const __m128i *)(sp + k)), y = _mm_loadu_si128((const __m128i *)(sp + k + 8));
              // bytes narrow (a marker saturates to 255 for the moment); then the markers of the group, one by one
              _mm_storeu_si128((__m128i *)(by + k), _mm_packus_epi16(x, y));
              const __m128i mx = _mm_cmpeq_epi16(_mm_and_si128(x, hi_bits), zero), my = _mm_cmpeq_epi16(_mm_and_si128(y, hi_bits), zero);
              unsigned m = ~(unsigned)_mm_movemask_epi8(_mm_packs_epi16(mx, my)) & 0xFFFFu;
              while (m) {
                const int j = __builtin_ctz(m);
                m &= m - 1;
                by[k + j] = lt[sp[k + j]];
              }
            }
#endif
            for (; k < b; k++) by[k] = lt[sp[k]];
          };
          const size_t tail = len > 32768 ? len - 32768 : 0;
          resolve(tail, len);
          size_t member_start = (size_t)-1;
          if (!c.ends.empty()) {
            member_start = (size_t)c.ends.back().out_pos;   // (a member that ended here: what follows belongs to the next one)
          } else if (c.F.member) {
            member_start = 0;
          }
          next.E = c.specE; next.stream_end = c.spec_end;
          window_after(*prev, by, len, member_start, next);
          publish();
          early = true;
          resolve(0, tail);
        } else {
          // decode again from where the predecessor stopped, the window known
          redone = true;
          std::vector<MemberEnd> ends;
          Inflater<uint8_t> inf(p_, n_, c.bytes, ends, *tab, fixed_, (size_t)-1);
          inf.win = prev->win.data();
          inf.lowest = -(int64_t)prev->win_len;
          bool se = false;
          next.E = inf.run(prev->E, hi, se);
          next.stream_end = se;
          c.nbytes = inf.pos;
          c.ends = std::move(ends);
          next.total = prev->total + c.nbytes;
          if (region_mode_) {
            ensure_writable(next.total);
            memcpy(region_ + prev->total, c.bytes.data(), c.nbytes);
          }
          size_t member_start = c.ends.empty() ? (size_t)-1 : (size_t)c.ends.back().out_pos;
          if (prev->E.member && member_start == (size_t)-1) member_start = 0;
          window_after(*prev, c.bytes.data(), c.nbytes, member_start, next);
        }
        if (!early) publish();
        const uint64_t t_crc = timing_ ? now_ns() : 0;
        if (timing_) (redone ? tm_.redo : tm_.resolve) += t_crc - t_fin;
        finish_segments(c, region_mode_ ? region_ + (next.total - c.nbytes) : c.bytes.data());
        if (timing_) tm_.crc += now_ns() - t_crc;
        {
          std::lock_guard<std::mutex> lk(mu_);
          c.done = true;
          if (region_mode_) {
            Info &in = info_[i];
            in.begin = next.total - c.nbytes; in.end = next.total; in.segs = c.segs; in.done = true;
            if (i + 1 == nchunks_) last_stream_end_ = next.stream_end;
            verify_prefix();
          }
          cv_.notify_all();
        }
      }
    } catch (const std::exception &e) {
      std::lock_guard<std::mutex> lk(mu_);
      if (error_.empty()) error_ = e.what();
      stop_ = true;
      cv_.notify_all();
    }
  }

 public:
  // threads >= 1; chunk_bytes: compressed bytes per chunk; region: the consumers parse the inflated file in place (segments
  // below) instead of taking it through read()
  Reader(const std::string &path, int threads, size_t chunk_bytes, bool region = false)
      : chunk_bytes_(std::max<size_t>(chunk_bytes, 64)), region_mode_(region) {
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
    build_fixed(fixed_);
    start_.E.bit = 0; start_.E.member = true; start_.win_len = 0;
    nchunks_ = (n_ + chunk_bytes_ - 1) / chunk_bytes_;
    threads = std::max(1, threads);
    lookahead_ = (size_t)threads + 3;
    slots_.resize(lookahead_);
    chains_.resize(lookahead_);
    if (region_mode_) {
      info_.resize(nchunks_);
      // deflate expands at most 1032-fold; the reservation costs address space only
      // (all of it or nothing: a smaller stretch would turn a valid file that inflates beyond it into an error half way through;
      //  the caller reads the file through zlib instead when this throws.  SLK_GZ_RESERVE_LIMIT: a cap in bytes, for tests.)
      const size_t want = std::min<size_t>((size_t)8 << 40, n_ * 1032 + ((size_t)64 << 20));
      const char *lim = getenv("SLK_GZ_RESERVE_LIMIT");
      void *m = (lim && (size_t)atoll(lim) < want) ? MAP_FAILED : mmap(nullptr, want, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (m != MAP_FAILED) { region_ = (uint8_t *)m; region_cap_ = want; }
      if (!region_) { if (p_) munmap((void *)p_, n_); ::close(fd_); throw ReserveError("cannot reserve address space to inflate " + path); }
    }
    nthreads_ = std::min<size_t>((size_t)threads, std::max<size_t>(1, nchunks_));
    th_.reserve(nthreads_);
    for (size_t t = 0; t < nthreads_; t++) th_.emplace_back([this] { worker(); });
  }
  Reader(const Reader &) = delete;
  void shutdown() {   // stops the workers and wakes every waiter (they return false)
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; cv_.notify_all(); }
    for (auto &t : th_) if (t.joinable()) t.join();
  }
  ~Reader() {
    shutdown();
    if (timing_)
      fprintf(stderr, "[pargz] %zu chunks of %zu bytes, %zu threads; thread-seconds: boundary search %.3f, speculative decode %.3f, "
              "waiting for the predecessor %.3f, markers %.3f (%.1f %% of the symbols may be markers), decoding again %.3f, crc %.3f\n", nchunks_, chunk_bytes_, nthreads_,
              (tm_.search - tm_.decode) / 1e9, tm_.decode / 1e9, tm_.wait / 1e9, tm_.resolve / 1e9, 100.0 * tm_.marked / std::max<uint64_t>(1, tm_.total), tm_.redo / 1e9, tm_.crc / 1e9);
    if (p_) munmap((void *)p_, n_);
    if (region_) munmap(region_, region_cap_);
    if (fd_ >= 0) ::close(fd_);
  }

  // ---- region mode: segment i = the inflated bytes of chunk i, [begin, end) of ONE buffer that starts at base() ----
  struct View { size_t begin = 0, end = 0, avail = 0; bool eof = false; };   // avail: bytes of the file inflated and checked so far
  size_t segments() const { return nchunks_; }
  void set_consumer_span(size_t k) { std::lock_guard<std::mutex> lk(mu_); span_ = std::max<size_t>(1, k); cv_.notify_all(); }
  const char *base() const { return (const char *)region_; }
  // Blocks until segment i and the one behind it (the records that start in i end there, usually) have arrived.  Throws on
  // corrupt input; false if the reader is being closed.
  // (first .. i: several segments taken as one, [begin of first, end of i))
  bool wait_segment(size_t i, View &v, size_t first = (size_t)-1) {
    if (first == (size_t)-1) first = i;
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !error_.empty() || stop_ || verified_ >= std::min(nchunks_, i + 2); });
    if (!error_.empty()) throw std::runtime_error("read error (corrupt compressed input?): " + error_);
    if (verified_ < std::min(nchunks_, i + 2)) return false;
    v.begin = info_[first].begin; v.end = info_[i].end;
    v.avail = info_[verified_ - 1].end;
    v.eof = verified_ == nchunks_;
    return true;
  }
  // ... and until more than `avail_known` bytes have (a record of segment i reached beyond what was there)
  bool wait_more(size_t i, size_t avail_known, View &v, size_t first = (size_t)-1) {
    if (first == (size_t)-1) first = i;
    std::unique_lock<std::mutex> lk(mu_);
    starving_++;
    cv_.notify_all();
    cv_.wait(lk, [&] { return !error_.empty() || stop_ || verified_ == nchunks_ || info_[verified_ - 1].end > avail_known; });
    starving_--;
    if (!error_.empty()) throw std::runtime_error("read error (corrupt compressed input?): " + error_);
    if (!(verified_ == nchunks_ || info_[verified_ - 1].end > avail_known)) return false;
    v.begin = info_[first].begin; v.end = info_[i].end;
    v.avail = info_[verified_ - 1].end;
    v.eof = verified_ == nchunks_;
    return true;
  }
  // The consumer is through with segment i: the inflaters may move on, the memory behind the slowest consumer goes back.
  void segment_parsed(size_t i) {
    size_t drop_from = 0, drop_to = 0;
    {
      std::lock_guard<std::mutex> lk(mu_);
      info_[i].parsed = true;
      while (consumed_ < nchunks_ && info_[consumed_].parsed) consumed_++;
      const size_t low = consumed_ ? (size_t)info_[consumed_ - 1].end : 0;   // first byte someone may still look at (and 2 before it)
      const size_t keep = low > 8192 ? (low - 8192) & ~(size_t)4095 : 0;
      if (keep > dropped_ + ((size_t)8 << 20)) { drop_from = dropped_; drop_to = keep; dropped_ = keep; }
      cv_.notify_all();
    }
    if (drop_to > drop_from) madvise(region_ + drop_from, drop_to - drop_from, MADV_DONTNEED);
  }

  // the next bytes of the inflated file; 0 at its end.  Throws on corrupt input, as zlib's gzread reports it.
  size_t read(char *dst, size_t cap) {
    size_t got = 0;
    while (got < cap) {
      if (out_chunk_ >= nchunks_) {
        if (run_len_ != 0) throw std::runtime_error("read error (corrupt compressed input?): gzip member cut short");
        break;
      }
      Chunk &c = slots_[out_chunk_ % lookahead_];
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return !error_.empty() || (consumed_ == out_chunk_ && next_claim_ > out_chunk_ && c.done); });
        if (!error_.empty()) throw std::runtime_error("read error (corrupt compressed input?): " + error_);
      }
      if (out_off_ == 0) {   // first visit: the member checks of this chunk
        for (const Seg &s : c.segs) {
          run_crc_ = (uint32_t)crc32_combine(run_crc_, s.crc, (z_off_t)s.len);
          run_len_ += s.len;
          if (s.ends_member) {
            if (run_crc_ != s.want_crc || (uint32_t)run_len_ != s.want_isize)
              throw std::runtime_error("read error (corrupt compressed input?): gzip CRC / length mismatch");
            run_crc_ = 0; run_len_ = 0;
          }
        }
      }
      const size_t k = std::min(cap - got, c.nbytes - out_off_);
      memcpy(dst + got, c.bytes.data() + out_off_, k);
      got += k;
      out_off_ += k;
      if (out_off_ >= c.nbytes) {
        const bool last = out_chunk_ + 1 >= nchunks_;
        if (last) {
          // the last chunk's chain tells whether the data ended where a member ended
          std::lock_guard<std::mutex> lk(mu_);
          const Chain &ch = chains_[out_chunk_ % lookahead_];
          if (!ch.stream_end) error_ = "unexpected end of data";
        }
        {
          std::lock_guard<std::mutex> lk(mu_);
          consumed_ = out_chunk_ + 1;
          cv_.notify_all();
        }
        out_chunk_++;
        out_off_ = 0;
        if (!error_.empty()) throw std::runtime_error("read error (corrupt compressed input?): " + error_);
      }
    }
    return got;
  }
};

}  // namespace pargz
}  // namespace slk

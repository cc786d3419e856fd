// titles.hpp -- finding read titles that occur more than once in the input (C++17, header only).
// The reference regroups the span rows of ALL fragments by title string (groupBy("seqTitle"), S/slacken/Classifier.scala:92),
// so fragments that share a title are ONE read there; and its paired reader joins the two files on the header
// (S/kmers/input/InputReader.scala:104-119), so a header that repeats inside a file multiplies.  Real inputs have neither,
// and the host streams them once; to stay exact on the inputs that do, the first pass remembers a 64-bit hash of every title
// it sees (8 bytes per read) and reports the hashes that repeat.  Those titles -- and only those -- are then re-read and
// regrouped exactly as the reference does (slacken_cli.cpp: resolve_repeated_titles).  A hash collision only adds a title to
// the re-read set, where titles are compared as strings.
#pragma once
#include <algorithm>
#include <cstdint>
#include <mutex>
#include <string_view>
#include <vector>

namespace slk_host {

inline uint64_t title_hash(std::string_view t) {
  uint64_t x = std::hash<std::string_view>()(t);
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;   // (the set below uses both ends of the word)
  return x ? x : 1;  // 0 marks an empty cell
}

// open addressing, linear probing, no deletion; the low `TAG` bits of a stored word are flags, the rest identifies the title
template <int TAG>
class FlatHashSet {
  std::vector<uint64_t> cells_;
  size_t n_ = 0;
  static constexpr uint64_t FLAGS = (1ull << TAG) - 1;
  void grow() {
    std::vector<uint64_t> old;
    old.swap(cells_);
    cells_.assign(old.empty() ? 1024 : old.size() * 2, 0);
    n_ = 0;
    for (uint64_t c : old) if (c) insert(c | FLAGS, (unsigned)(c & FLAGS));
  }

 public:
  // Adds `flags` to the entry of h (creating it).  Returns the flags the entry had before: 0 = new.
  unsigned insert(uint64_t h, unsigned flags) {
    if ((n_ + 1) * 2 > cells_.size()) grow();
    const uint64_t id = h | FLAGS;  // (never 0)
    const size_t mask = cells_.size() - 1;
    for (size_t i = (size_t)(h >> 20) & mask;; i = (i + 1) & mask) {
      uint64_t &c = cells_[i];
      if (c == 0) { c = (id & ~FLAGS) | flags; n_++; return 0; }
      if ((c | FLAGS) == id) { const unsigned before = (unsigned)(c & FLAGS); c |= flags; return before ? before : (TAG ? 0u : 1u); }
    }
  }
  void prefetch(uint64_t h) const {
    if (!cells_.empty()) __builtin_prefetch(&cells_[(size_t)(h >> 20) & (cells_.size() - 1)]);
  }
  bool contains(uint64_t h) const {
    if (cells_.empty()) return false;
    const uint64_t id = h | FLAGS;
    const size_t mask = cells_.size() - 1;
    for (size_t i = (size_t)(h >> 20) & mask;; i = (i + 1) & mask) {
      if (cells_[i] == 0) return false;
      if ((cells_[i] | FLAGS) == id) return true;
    }
  }
  size_t size() const { return n_; }
};

// hashes reported as repeated, from any thread
class RepeatedTitles {
  std::mutex mu_;
  std::vector<uint64_t> h_;
  std::vector<uint64_t> unmatched_;   // headers of paired records that became no fragment (seqio.hpp: FragmentSource)

 public:
  void add_unmatched(uint64_t h) { std::lock_guard<std::mutex> lk(mu_); unmatched_.push_back(h); }
  // once every fragment title of the run is known: an unmatched record whose header is one of them makes that title repeated
  template <class F> void settle_unmatched(F is_fragment_title) {
    std::lock_guard<std::mutex> lk(mu_);
    for (uint64_t h : unmatched_) if (is_fragment_title(h)) h_.push_back(h);
    unmatched_.clear();
  }
  void add(uint64_t h) { std::lock_guard<std::mutex> lk(mu_); h_.push_back(h); }
  void add(const std::vector<uint64_t> &v) { if (!v.empty()) { std::lock_guard<std::mutex> lk(mu_); h_.insert(h_.end(), v.begin(), v.end()); } }
  bool empty() { std::lock_guard<std::mutex> lk(mu_); return h_.empty(); }
  FlatHashSet<0> to_set() {
    std::lock_guard<std::mutex> lk(mu_);
    FlatHashSet<0> s;
    for (uint64_t h : h_) s.insert(h, 0);
    return s;
  }
};

// every fragment title of the run, inserted from the formatting threads: 256 independently locked shards
class ConcurrentTitleSet {
  struct Shard { std::mutex mu; FlatHashSet<0> set; };
  std::vector<Shard> shards_;

 public:
  ConcurrentTitleSet() : shards_(256) {}
  bool insert(uint64_t h) {  // true: seen before
    Shard &s = shards_[h & 255];
    std::lock_guard<std::mutex> lk(s.mu);
    return s.set.insert(h, 0) != 0;
  }
  bool contains(uint64_t h) {
    Shard &s = shards_[h & 255];
    std::lock_guard<std::mutex> lk(s.mu);
    return s.set.contains(h);
  }
  // A slice's worth of hashes at once: grouped by shard (one lock per shard and slice) and with the cells prefetched a few
  // inserts ahead -- an insert is otherwise one cache miss in a table of 16 bytes per read.  `hs` is reordered.  The hashes that
  // had been seen before are appended to `repeated`.
  void insert_many(std::vector<uint64_t> &hs, std::vector<uint64_t> &repeated) {
    if (hs.empty()) return;
    uint32_t start[257] = {0};
    for (uint64_t h : hs) start[(h & 255) + 1]++;
    for (int i = 0; i < 256; i++) start[i + 1] += start[i];
    std::vector<uint64_t> by(hs.size());
    {
      uint32_t at[256];
      for (int i = 0; i < 256; i++) at[i] = start[i];
      for (uint64_t h : hs) by[at[h & 255]++] = h;
    }
    hs.swap(by);
    for (int sh = 0; sh < 256; sh++) {
      const uint32_t a = start[sh], b = start[sh + 1];
      if (a == b) continue;
      Shard &s = shards_[sh];
      std::lock_guard<std::mutex> lk(s.mu);
      const uint32_t AHEAD = 8;
      for (uint32_t i = a; i < std::min(b, a + AHEAD); i++) s.set.prefetch(hs[i]);
      for (uint32_t i = a; i < b; i++) {
        if (i + AHEAD < b) s.set.prefetch(hs[i + AHEAD]);
        if (s.set.insert(hs[i], 0) != 0) repeated.push_back(hs[i]);
      }
    }
  }
};

}  // namespace slk_host

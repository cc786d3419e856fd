// hostside.h -- what the host-side translation units of the library share (capi.hip: the C ABI of one index / one stream;
// shardset.hip: the table-sharded set of indices): the handles' structures, error reporting, the staged copies between caller
// memory and HBM.  Internal: not part of the C ABI.
#pragma once
#include "../../include/slacken_amd.h"
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace slk;

inline thread_local std::string g_err;

inline int32_t fail(int32_t code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

// A failed HIP call leaves its code behind as the thread's "last error"; it is read back (cleared) here so that it cannot
// be mistaken for the result of a later kernel launch that is checked with hipGetLastError().
#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      (void)hipGetLastError();                                                                         \
      return fail(SLK_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                                  \
  } while (0)


struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T *as() const { return (T *)p; }
};

// ---- host side of the copies ----------------------------------------------------------------------------------------------
// A few threads that move bytes between the caller's (pageable) memory and the pinned staging buffers: one thread copies
// at 10-12 GB/s, PCIe Gen5 x16 takes ~55.  Started on first use; SLK_COPY_THREADS (default 6, 1 = the calling thread only).
class HostPool {
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(size_t)> *fn_ = nullptr;
  size_t next_ = 0, n_ = 0, active_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
  std::mutex call_mu_;  // one parallel_for at a time (others wait: the pool is for memory-bound copies)

  void worker() {
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return stop_ || (gen_ != seen && fn_); });
      if (stop_) return;
      seen = gen_;
      while (fn_ && next_ < n_) {
        size_t i = next_++;
        active_++;
        const std::function<void(size_t)> *f = fn_;
        lk.unlock();
        (*f)(i);
        lk.lock();
        active_--;
      }
      done_.notify_all();
    }
  }

 public:
  explicit HostPool(size_t n) { for (size_t i = 0; i + 1 < n; i++) th_.emplace_back([this] { worker(); }); }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  size_t size() const { return th_.size() + 1; }
  void parallel_for(size_t n, const std::function<void(size_t)> &f) {  // f(0) .. f(n-1), the caller takes part
    if (n == 0) return;
    if (th_.empty() || n == 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::lock_guard<std::mutex> call(call_mu_);
    std::unique_lock<std::mutex> lk(mu_);
    fn_ = &f; next_ = 0; n_ = n; gen_++;
    cv_.notify_all();
    while (next_ < n_) {
      size_t i = next_++;
      active_++;
      lk.unlock();
      f(i);
      lk.lock();
      active_--;
    }
    done_.wait(lk, [&] { return active_ == 0; });
    fn_ = nullptr;
  }
};
inline HostPool &host_pool() {
  static HostPool *pool = [] {
    const char *e = getenv("SLK_COPY_THREADS");
    long n = e ? atol(e) : 6;
    unsigned hc = std::thread::hardware_concurrency();
    if (hc && n > (long)hc) n = hc;
    return new HostPool((size_t)std::max<long>(1, n));  // (never destroyed: worker threads must not be joined at process exit)
  }();
  return *pool;
}
inline void parallel_memcpy(void *dst, const void *src, size_t n) {
  const size_t SLICE = (size_t)1 << 20;
  if (n <= 2 * SLICE) { memcpy(dst, src, n); return; }
  const size_t parts = std::min(host_pool().size(), (n + SLICE - 1) / SLICE);
  const size_t per = ((n + parts - 1) / parts + 63) & ~(size_t)63;
  host_pool().parallel_for(parts, [&](size_t i) {
    const size_t a = i * per, b = std::min(n, a + per);
    if (a < b) memcpy((char *)dst + a, (const char *)src + a, b - a);
  });
}

// Host memory the library has pinned (slk_host_alloc / slk_host_register): copies from and to it are DMA'd directly.
struct PinnedRanges {
  std::mutex mu;
  std::map<uintptr_t, std::pair<size_t, bool>> ranges;  // start -> (bytes, allocated by us)
  void add(void *p, size_t n, bool owned) { std::lock_guard<std::mutex> lk(mu); ranges[(uintptr_t)p] = {n, owned}; }
  bool remove(void *p, bool *owned) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = ranges.find((uintptr_t)p);
    if (it == ranges.end()) return false;
    *owned = it->second.second;
    ranges.erase(it);
    return true;
  }
  bool covers(const void *p, size_t n) {
    std::lock_guard<std::mutex> lk(mu);
    if (ranges.empty()) return false;
    auto it = ranges.upper_bound((uintptr_t)p);
    if (it == ranges.begin()) return false;
    --it;
    return (uintptr_t)p + n <= it->first + it->second.first;
  }
};
inline PinnedRanges &pinned() { static PinnedRanges *r = new PinnedRanges(); return *r; }

// Pinned staging buffers: every copy between PAGEABLE caller memory and HBM goes through them (copy_in / copy_out), so the
// runtime never has to pin the caller's pages for DMA -- that path took tens of ms per call once an application with many
// threads was mapping and unmapping memory around it.  Memory from slk_host_alloc / slk_host_register skips them.
constexpr int N_STAGE = 3;
struct Staging {
  void *buf[N_STAGE] = {};
  hipEvent_t ev[N_STAGE] = {};
  bool busy[N_STAGE] = {};
  int next = 0;
  void release() {
    for (int i = 0; i < N_STAGE; i++) {
      if (buf[i]) (void)hipHostFree(buf[i]);
      if (ev[i]) (void)hipEventDestroy(ev[i]);
      buf[i] = nullptr; ev[i] = nullptr; busy[i] = false;
    }
  }
};

struct slk_index {
  int32_t device = 0;
  slk_params params{};
  ScanParams sp{};
  uint64_t *cells = nullptr;
  uint64_t nbuckets = 0;
  int32_t bucket_bits = 0, taxon_bits = 0, disp_bits = 0;   // bucket_bits = ceil(log2(nbuckets)): the hash bits that choose the bucket
  bool bucket_flag = false;        // the cells keep their top bit for the buckets' "a record went past" flag (engine.h: TableGeom.flag)
  uint32_t shard = 0, n_shards = 0;  // slk_index_set_shard: keep only the records of this shard
  int32_t *d_max_disp = nullptr;
  unsigned long long *d_counters = nullptr;  // inserted, duplicate, overflow
  int32_t *d_parents = nullptr;   // the taxonomy as given (ids of the caller)
  int32_t T = 0;
  std::vector<int32_t> h_parents;  // host copy, for the dense renumbering at finalize
  // dense taxon ids (engine.h: TableView.to_orig): set up by slk_index_finalize when the caller's ids need more than 22 bits
  int32_t *d_parents_dense = nullptr, *d_to_orig = nullptr, *d_to_dense = nullptr;
  uint4 *d_nodes = nullptr;        // kernel_parents() with an Euler tour (engine.h: FusedArgs.nodes); null: more than 2^22 ids, no lane kernel
  uint4 *d_nodes_orig = nullptr;   // the same for the taxonomy as given (the staged classify kernel works in the caller's ids); may BE d_nodes
  int32_t D = 0;                   // nodes of the taxonomy = largest dense id (0: ids are stored as given)
  bool finalized = false;
  int32_t max_disp = 0;
  uint64_t records = 0, dups = 0;
  uint64_t unplaced = 0;           // records of the last insert that found no cell within reach of the displacement field (capi.hip: insert_growing)
  uint32_t grown = 0;              // times the table was moved to a larger one because of that
  float load_target = 0;           // the load factor the table was sized for (given, or chosen by the free memory: slk_index_create)
  hipStream_t build_stream = nullptr;
  DevBuf stage_keys, stage_taxa;
  Staging staging;    // host -> HBM copies of the build calls
  int W = 1;          // id columns; > 1: the wide path (wide.hip) with its own table
  WideParams wp{};
  WideTable wt{};

  TableGeom geom() const {
    TableGeom g{};
    g.nbuckets = nbuckets;
    g.q = bucket_bits;
    g.rem_mask = (1ULL << (64 - bucket_bits)) - 1;
    g.flag = bucket_flag ? (1ULL << 63) : 0;
    g.taxon_bits = taxon_bits;
    g.disp_bits = disp_bits;
    return g;
  }
  TableView view() const {
    TableView v;
    v.cells = cells;
    v.g = geom();
    v.max_disp = max_disp;
    v.to_orig = d_to_orig;
    return v;
  }
  // what the fused kernels walk: the parents array in the ids the cells hold
  const int32_t *kernel_parents() const { return D ? d_parents_dense : d_parents; }
  const uint4 *kernel_nodes() const { return d_nodes; }
  int32_t kernel_ntax() const { return D ? D + 1 : T; }
  int internal_taxon_bits() const {
    if (!D) return taxon_bits;
    int b = 1;
    while ((1LL << b) <= (long long)D) b++;
    return b;
  }
};

struct slk_stream {
  slk_index *ix = nullptr;
  int32_t device = 0;  // copy: the stream may be destroyed after its index
  hipStream_t s = nullptr;
  DevBuf span_keys, span_meta, span_taxon, span_count;  // per-batch scratch (sparse per-read regions)
  DevBuf bases, offsets, mate_bases, mate_offsets;      // staging for the host-pointer entry points
  DevBuf out_taxon, out_cls, out_nd, out_tk, out_nh, out_offsets, out_items, defer_list, scan_tmp;
  bool merged_hits = false;                 // slk_stream_set_merged_hits: hit lists as TaxonCounts.fromHits merges them
  DevBuf pk_codes, pk_valid, pk_mate_codes, pk_mate_valid;   // slk_classify_batch_packed: the reads as they arrive (3 bits per base)
  hipEvent_t ev_unpack = nullptr;
  hipStream_t s2 = nullptr;                 // the segment pass and the wave pass run here, beside the long-lane pass on s
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int32_t *d_status = nullptr;     // device error bits of the fused kernels
  int32_t *h_status = nullptr;     // pinned copy, refreshed after every classify launch
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  Staging staging;
  // large host-pointer calls: the reads go up on a second stream, sub-batch by sub-batch, while the kernels of the
  // sub-batches before run on s (slk_classify_batch)
  hipStream_t cs = nullptr;
  Staging staging_c;
  std::vector<hipEvent_t> up_ev;
  // ... and the results of sub-batch i come down on a third stream while sub-batch i + 1 runs (pinned result buffers only)
  hipStream_t ds = nullptr;
  std::vector<hipEvent_t> dn_ev;
  bool reran = false;           // check_status classified queued calls again (a taxon map had overflowed): results on the host are stale
  bool timed = false;
  bool last_used_lane = false;  // the last classify call ran the lane kernel (defer_list[0] is its deferral count)
  // The arguments of every classify call queued since the stream was last synchronised, for the unbounded re-run
  // (check_status): a taxon-map overflow is only seen at the next synchronisation, and by then several calls may have gone by.
  struct LastCall {
    bool valid = false, want_hits = false;
    Thresholds thr{};
    const uint8_t *bases = nullptr, *mate_bases = nullptr;
    const uint64_t *offsets = nullptr, *mate_offsets = nullptr;
    uint64_t R = 0, total = 0, mate_total = 0, out_stride = 0;
    uint64_t span_shift = 0;   // slots the span arrays are moved by for this call (a sub-batch of a larger host call: run_classify)
    int32_t min_hit_groups = 0, C = 0;
    int32_t *out_taxon = nullptr, *out_nd = nullptr, *out_tk = nullptr, *out_nh = nullptr, *out_np = nullptr;
    uint8_t *out_cls = nullptr;
  };
  std::vector<LastCall> queued;
};

extern "C" __attribute__((visibility("hidden"))) int32_t check_status(slk_stream *st);

constexpr size_t STAGE_BYTES = (size_t)8 << 20;

inline int32_t stage_ready(Staging *g) {
  if (g->buf[0]) return SLK_OK;
  for (int i = 0; i < N_STAGE; i++) {
    HIPCHK(hipHostMalloc(&g->buf[i], STAGE_BYTES, hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&g->ev[i], hipEventDisableTiming));
  }
  return SLK_OK;
}

// caller memory -> HBM, ordered on s.  The caller's buffer is free on return unless it is pinned memory of the library
// (then the DMA reads it directly and is complete when s has been synchronised -- every host entry point does before it
// returns); the last DMA may still be in flight.
inline int32_t copy_in(Staging *g, hipStream_t s, void *d_dst, const void *h_src, size_t n) {
  if (n == 0) return SLK_OK;
  if (pinned().covers(h_src, n)) {
    HIPCHK(hipMemcpyAsync(d_dst, h_src, n, hipMemcpyHostToDevice, s));
    return SLK_OK;
  }
  int32_t rc = stage_ready(g);
  if (rc) return rc;
  for (size_t o = 0; o < n; o += STAGE_BYTES) {
    const size_t len = std::min(STAGE_BYTES, n - o);
    const int b = g->next;
    g->next = (g->next + 1) % N_STAGE;
    if (g->busy[b]) HIPCHK(hipEventSynchronize(g->ev[b]));  // the DMA that last used this buffer
    parallel_memcpy(g->buf[b], (const char *)h_src + o, len);
    HIPCHK(hipMemcpyAsync((char *)d_dst + o, g->buf[b], len, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(g->ev[b], s));
    g->busy[b] = true;
  }
  return SLK_OK;
}

// HBM -> caller memory, after everything queued on s; complete on return.  The DMA of one piece overlaps the copy of the
// piece before it into the caller's buffer.
inline int32_t copy_out(Staging *g, hipStream_t s, void *h_dst, const void *d_src, size_t n) {
  if (n == 0) return SLK_OK;
  if (pinned().covers(h_dst, n)) {
    HIPCHK(hipMemcpyAsync(h_dst, d_src, n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return SLK_OK;
  }
  int32_t rc = stage_ready(g);
  if (rc) return rc;
  size_t prev_o = 0, prev_len = 0;
  int prev_b = -1;
  for (size_t o = 0; o < n; o += STAGE_BYTES) {
    const size_t len = std::min(STAGE_BYTES, n - o);
    const int b = prev_b < 0 ? 0 : (prev_b + 1) % N_STAGE;
    // (stream order protects the buffer: an earlier copy_in DMA out of it is queued before this write into it)
    HIPCHK(hipMemcpyAsync(g->buf[b], (const char *)d_src + o, len, hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(g->ev[b], s));
    g->busy[b] = true;
    if (prev_b >= 0) {
      HIPCHK(hipEventSynchronize(g->ev[prev_b]));
      parallel_memcpy((char *)h_dst + prev_o, g->buf[prev_b], prev_len);
    }
    prev_b = b; prev_o = o; prev_len = len;
  }
  if (prev_b >= 0) {
    HIPCHK(hipEventSynchronize(g->ev[prev_b]));
    parallel_memcpy((char *)h_dst + prev_o, g->buf[prev_b], prev_len);
  }
  return SLK_OK;
}
inline int32_t copy_in(slk_stream *st, void *d_dst, const void *h_src, size_t n) { return copy_in(&st->staging, st->s, d_dst, h_src, n); }
inline int32_t copy_out(slk_stream *st, void *h_dst, const void *d_src, size_t n) { return copy_out(&st->staging, st->s, h_dst, d_src, n); }

// Every entry point that launches kernels starts here: select the index's device and drop whatever error code an earlier,
// unrelated HIP call of this thread (this library's or the application's) left behind, so that the hipGetLastError()
// after a launch reports that launch.
inline int32_t set_device(const slk_index *ix) {
  HIPCHK(hipSetDevice(ix->device));
  (void)hipGetLastError();
  return SLK_OK;
}


// defined in capi.hip (inside its extern "C" block), used by shardset.hip too; not exported
#define SLK_INTERNAL extern "C" __attribute__((visibility("hidden")))
SLK_INTERNAL uint64_t span_slots(uint64_t total_bases, uint64_t total_mate_bases, uint64_t R, bool paired);
SLK_INTERNAL int32_t ensure_scratch(slk_stream *st, uint64_t slots, uint64_t R);
SLK_INTERNAL int32_t check_ready(const slk_index *ix, const slk_stream *st, bool need_tax);
SLK_INTERNAL bool lane_path_ok(const slk_index *ix);
SLK_INTERNAL int32_t upload_reads(slk_stream *st, const uint8_t *bases, const uint64_t *offsets, const uint8_t *mate_bases,
                                  const uint64_t *mate_offsets, uint64_t R, uint64_t *total, uint64_t *mate_total);
SLK_INTERNAL int32_t counts_to_offsets(slk_stream *st, const int32_t *d_counts, uint64_t R, uint64_t *out_offsets, uint64_t capacity);

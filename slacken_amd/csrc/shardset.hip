// shardset.hip -- table-sharded classification in ONE process (BASELINE.json configs[3]): a library whose
// record table exceeds one GPU's HBM is spread over several indices -- member g holds the records with fmix64(key) mod n == g
// (slk_index_set_shard) -- and a ROUND classifies one batch of fragments per member.  A round passes through three jobs,
//   EMIT    every member scans ITS fragments and appends their minimizers to per-owner send regions        (lane.hip: step kernel)
//   LOOKUP  every member answers the keys it received from its shard of the table                          (step kernel / shard.hip)
//   APPLY   every member folds the answers into its fragments' taxon maps and resolves                     (step kernel)
// with two exchanges between them (8-byte keys to their owners, 4-byte taxa back into the positions the keys had), and the rounds
// of a call are PIPELINED: step t launches, per member, ONE kernel that carries EMIT(round t), LOOKUP(round t - 2) and
// APPLY(round t - 4), while the exchanges of rounds t - 1 and t - 3 run on the members' exchange streams beside it.  The host
// never waits for a stage: its only wait per step is for the cursors of the round emitted one step earlier (the exchange's split
// sizes), by which time the next step is already queued.
// This replaces the shuffle behind the reference's join (S/slacken/Classifier.scala:84-95: spans JOIN records ON id, then regrouped
// by title) for the one case where data must move; with a table that fits one GPU the replicated mode needs no exchange at all.
// Fragments the lane kernel does not take (over 1000 bases, more than 12 distinct taxa) make a second, staged round of the same
// shape once the pipeline has drained: wave-per-fragment scan into span arrays, keys collected by owner, the same exchange,
// unbounded classify kernel.
//
// The exchange is RCCL's (ncclSend / ncclRecv between the members' exchange streams, one communicator per member from
// ncclCommInitAll; the library is loaded at run time) when the members sit on distinct devices, and device-to-device copies ordered
// by events otherwise -- several members on ONE device is how this path is tested on a one-GPU box.  Copies between devices that
// cannot reach each other's memory (hipDeviceCanAccessPeer) are staged through pinned host memory.
#include "hostside.h"

#include <dlfcn.h>

#include <memory>

namespace {

// ---- the few RCCL entry points, resolved at run time (no link-time dependency; PyTorch processes carry their own copy) ----
typedef struct ncclComm *ncclComm_t;
enum { ncclInt32 = 2, ncclInt64 = 4 };   // ncclDataType_t values of the two element types that travel (rccl.h)
struct Rccl {
  void *h = nullptr;
  bool tried = false, ok = false;
  int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool load() {
    if (tried) return ok;   // (the verdict of the first attempt: a library that lacked a symbol is not asked again)
    tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
    GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
    Send = (decltype(Send))dlsym(h, "ncclSend");
    Recv = (decltype(Recv))dlsym(h, "ncclRecv");
    GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
    ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
    return ok;
  }
};
Rccl &rccl() { static Rccl r; return r; }
// One grouped exchange at a time in the process: several sets may live on the same devices (the CLI's host threads each drive one),
// and grouped calls of several communicators issued by unsynchronised threads may reach a device in different orders -- the
// documented way to deadlock NCCL / RCCL.
std::mutex &rccl_mu() { static std::mutex m; return m; }

// ---- kernels of the staged round: the SEQUENCE-flag spans of a batch, by owner ----
// pass 1 (fill == false): counts[owner] += 1 per such span; pass 2: keys and span slots to owner_start[owner] + cursor (order
// inside an owner's list is arbitrary: the slot comes along); flagged spans get their special taxon, the others NONE for now.
__global__ void __launch_bounds__(256) collect_keys_kernel(const uint64_t *__restrict__ offsets, const uint64_t *__restrict__ mate_offsets,
                                                           uint64_t R, const uint64_t *__restrict__ span_keys, const int32_t *__restrict__ span_meta,
                                                           const int32_t *__restrict__ span_count, uint32_t n_shards, bool fill,
                                                           unsigned long long *__restrict__ counts, const uint64_t *__restrict__ owner_start,
                                                           int64_t *__restrict__ out_keys, uint64_t *__restrict__ out_slots,
                                                           int32_t *__restrict__ span_taxon) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t r = wave; r < R; r += nwaves) {
    const uint64_t base = span_region(offsets, mate_offsets, r);
    const int32_t n = span_count[r];
    for (int32_t j = (int32_t)lane; j < n; j += 64) {
      const int32_t flag = meta_flag(span_meta[base + j]);
      if (flag != 1) {
        if (fill) span_taxon[base + j] = flag == 2 ? -1 : -2;   // spanToHit: AMBIGUOUS_SPAN / MATE_PAIR_BORDER (KeyValueIndex.scala:176-185)
        continue;
      }
      const uint64_t key = span_keys[base + j];
      const uint32_t owner = (uint32_t)(fmix64(key) % n_shards);
      const unsigned long long at = atomicAdd(&counts[owner], 1ULL);
      if (fill) {
        out_keys[owner_start[owner] + at] = (int64_t)key;
        out_slots[owner_start[owner] + at] = base + j;
        span_taxon[base + j] = 0;
      }
    }
  }
}
__global__ void __launch_bounds__(256) scatter_taxa_kernel(const uint64_t *__restrict__ slots, const int32_t *__restrict__ taxa, uint64_t n,
                                                           int32_t *__restrict__ span_taxon) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) span_taxon[slots[i]] = taxa[i];
}

// Rounds in flight on a member: EMIT(t) .. APPLY(t + 4), and one more whose results are on their way to the caller.  The large
// transient buffers (keys out, keys in, answers) live for three steps and are kept in rings of three.
constexpr int NSLOT = 6, NBIG = 3;

struct Slot {   // what a round leaves on a member between its EMIT and its APPLY
  DevBuf bases, offsets, mate_bases, mate_offsets;                       // the reads (host-pointer rounds)
  DevBuf send_meta, cursors, log, tile_rows, read_info, defer, taxa;     // engine.h: ShardIO / ApplyJob
  DevBuf span_meta, span_taxon, span_count, out_offsets, out_items;      // hit lists
  DevBuf out_taxon, out_cls, out_nd, out_tk, out_nh;                     // results (host-pointer rounds)
  uint64_t *h_cursors = nullptr;                                          // pinned [n + 3]
  hipEvent_t ev_up = nullptr, ev_keys = nullptr, ev_taxa = nullptr;       // reads uploaded; the round's keys arrived HERE; its taxa came back
  slk_shard_lists lists{};
  uint64_t total = 0, mate_total = 0;
  bool failed = false;                                                    // a send region overflowed: the whole batch takes the staged route
  void release() {
    for (DevBuf *b : {&bases, &offsets, &mate_bases, &mate_offsets, &send_meta, &cursors, &log, &tile_rows, &read_info, &defer, &taxa, &span_meta,
                      &span_taxon, &span_count, &out_offsets, &out_items, &out_taxon, &out_cls, &out_nd, &out_tk, &out_nh})
      b->release();
    if (h_cursors) (void)hipHostFree(h_cursors);
    for (hipEvent_t *e : {&ev_up, &ev_keys, &ev_taxa}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
    h_cursors = nullptr;
  }
};
struct Big {    // send_keys: EMIT(t) .. exchange after step t; recv_keys / found: exchange .. LOOKUP(t + 2) .. exchange after step t + 2
  DevBuf send_keys, recv_keys, found;
  void release() { send_keys.release(); recv_keys.release(); found.release(); }
};

struct Member {
  slk_index *ix = nullptr;
  int32_t device = 0;           // copy: the set may be destroyed after its members' indices
  slk_stream *st = nullptr;     // this set's compute stream on the member (HIP stream, staging buffers, scratch of the staged round)
  hipStream_t xs = nullptr;     // the exchanges
  hipStream_t us = nullptr;     // reads up (host-pointer rounds)
  hipStream_t ds = nullptr;     // results down
  Staging staging_u, staging_d;
  ncclComm_t comm = nullptr;
  Slot slot[NSLOT];
  Big big[NBIG];
  hipEvent_t ev_step[NSLOT] = {};   // step t's kernel (and the copy of its cursors) has finished
  // the staged round's buffers
  DevBuf counts, starts, out_keys, slots, recv_keys, found, taxa;
  uint64_t *h_counts = nullptr;     // pinned [n]: keys per owner
  hipEvent_t ev_sent = nullptr, ev_found = nullptr, ev_bounce = nullptr;
  void *h_bounce = nullptr;         // pinned: copies to devices that cannot reach this one's memory
  size_t bounce_cap = 0;
  void release() {
    for (Slot &s : slot) s.release();
    for (Big &b : big) b.release();
    for (DevBuf *b : {&counts, &starts, &out_keys, &slots, &recv_keys, &found, &taxa}) b->release();
    staging_u.release(); staging_d.release();
    if (h_counts) (void)hipHostFree(h_counts);
    if (h_bounce) (void)hipHostFree(h_bounce);
    for (hipEvent_t &e : ev_step) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (ev_sent) (void)hipEventDestroy(ev_sent);
    if (ev_found) (void)hipEventDestroy(ev_found);
    if (ev_bounce) (void)hipEventDestroy(ev_bounce);
    for (hipStream_t *s : {&xs, &us, &ds}) { if (*s) { (void)hipStreamSynchronize(*s); (void)hipStreamDestroy(*s); } *s = nullptr; }
    h_counts = nullptr; h_bounce = nullptr; bounce_cap = 0; ev_sent = ev_found = ev_bounce = nullptr;
  }
};

}  // namespace

struct slk_shardset {
  int n = 0;
  int mode = SLK_EXCHANGE_COPY;
  std::vector<Member> m;
  // cnt[a][b]: elements member a sends to member b in the current exchange
  std::vector<std::vector<uint64_t>> cnt;
  // peer[a][b]: device of a can read the memory of b's device (or it is the same device); otherwise copies are staged through the host
  std::vector<std::vector<char>> peer;
};

namespace {

int32_t use(const Member &mb) {
  HIPCHK(hipSetDevice(mb.device));
  (void)hipGetLastError();
  return SLK_OK;
}

// Moves, for every pair (a, b), n[a][b] elements of esz bytes from src[a] + src_off(a, b) to dst[b] + dst_off(b, a), on the streams
// str[.].  The source of a was produced behind ev[a]; what arrives at b is consumed behind b's stream str[b].
template <class SrcOff, class DstOff>
int32_t exchange(slk_shardset *set, const std::vector<const void *> &src, const std::vector<void *> &dst, const std::vector<std::vector<uint64_t>> &n,
                 size_t esz, int nccl_type, const std::vector<hipStream_t> &str, const std::vector<hipEvent_t> &ev, SrcOff src_off, DstOff dst_off) {
  const int W = set->n;
  // every member's stream first falls in behind its own event: what it sends is ready there, and whatever still read the buffers it
  // is about to receive into has finished (RCCL orders the transfers themselves; the copies below wait for their sources' events)
  for (int a = 0; a < W; a++) {
    int32_t rc = use(set->m[a]);
    if (rc) return rc;
    HIPCHK(hipStreamWaitEvent(str[a], ev[a], 0));
  }
  if (set->mode == SLK_EXCHANGE_RCCL) {
    std::lock_guard<std::mutex> lk(rccl_mu());
    int e = rccl().GroupStart();
    if (e != 0) return fail(SLK_E_HIP, "ncclGroupStart failed: %s", rccl().GetErrorString(e));
    int32_t rc = SLK_OK;
    for (int a = 0; a < W && !e && !rc; a++) {
      rc = use(set->m[a]);
      for (int b = 0; b < W && !e && !rc; b++) {
        if (n[a][b]) e = rccl().Send((const char *)src[a] + src_off(a, b) * esz, n[a][b], nccl_type, b, set->m[a].comm, str[a]);
        if (!e && n[b][a]) e = rccl().Recv((char *)dst[a] + dst_off(a, b) * esz, n[b][a], nccl_type, b, set->m[a].comm, str[a]);
      }
    }
    const int e2 = rccl().GroupEnd();   // (always closed: a group left open would swallow the thread's later calls)
    if (rc) return rc;
    if (e || e2) return fail(SLK_E_HIP, "RCCL exchange failed: %s", rccl().GetErrorString(e ? e : e2));
    return SLK_OK;
  }
  for (int b = 0; b < W; b++) {
    for (int a = 0; a < W; a++) {
      if (!n[a][b]) continue;
      const size_t bytes = n[a][b] * esz;
      const char *from = (const char *)src[a] + src_off(a, b) * esz;
      char *to = (char *)dst[b] + dst_off(b, a) * esz;
      if (set->peer[b][a]) {
        int32_t rc = use(set->m[b]);
        if (rc) return rc;
        HIPCHK(hipStreamWaitEvent(str[b], ev[a], 0));
        HIPCHK(hipMemcpyAsync(to, from, bytes, hipMemcpyDefault, str[b]));
      } else {
        // no peer access between the two devices: down to a's pinned bounce buffer on a's stream, up from there on b's.  (The
        // buffer is a's; a region per destination, so that the pairs of one exchange do not wait for each other.)
        Member &ma = set->m[a];
        int32_t rc = use(ma);
        if (rc) return rc;
        uint64_t row = 0, off = 0;
        for (int d = 0; d < W; d++) { if (d == b) off = row; row += n[a][d]; }
        if (row * esz > ma.bounce_cap) {
          HIPCHK(hipStreamSynchronize(str[a]));   // (the copies still using the old buffer)
          if (ma.h_bounce) HIPCHK(hipHostFree(ma.h_bounce));
          ma.h_bounce = nullptr; ma.bounce_cap = 0;
          HIPCHK(hipHostMalloc(&ma.h_bounce, row * esz + row * esz / 4 + 4096, hipHostMallocPortable));
          ma.bounce_cap = row * esz + row * esz / 4 + 4096;
        }
        HIPCHK(hipStreamWaitEvent(str[a], ev[a], 0));
        HIPCHK(hipMemcpyAsync((char *)ma.h_bounce + off * esz, from, bytes, hipMemcpyDeviceToHost, str[a]));
        HIPCHK(hipEventRecord(ma.ev_bounce, str[a]));
        rc = use(set->m[b]);
        if (rc) return rc;
        HIPCHK(hipStreamWaitEvent(str[b], ma.ev_bounce, 0));
        HIPCHK(hipMemcpyAsync(to, (const char *)ma.h_bounce + off * esz, bytes, hipMemcpyHostToDevice, str[b]));
      }
    }
  }
  return SLK_OK;
}

struct RoundArgs {
  slk_shard_batch *batches;
  int32_t min_hit_groups;
  const double *thresholds;
  int32_t C;
};

// exclusive prefix sums over the exchange matrix: where a's elements for b start in a's send buffer, and in b's receive buffer
struct Layout {
  std::vector<std::vector<uint64_t>> soff, roff;
  std::vector<uint64_t> sent, received;
  explicit Layout(const std::vector<std::vector<uint64_t>> &n) {
    const size_t W = n.size();
    soff.assign(W, std::vector<uint64_t>(W, 0)); roff = soff;
    sent.assign(W, 0); received.assign(W, 0);
    for (size_t a = 0; a < W; a++)
      for (size_t b = 0; b < W; b++) { soff[a][b] = sent[a]; sent[a] += n[a][b]; }
    for (size_t b = 0; b < W; b++)
      for (size_t a = 0; a < W; a++) { roff[b][a] = received[b]; received[b] += n[a][b]; }
  }
};

int32_t sync_all(slk_shardset *set) {
  for (Member &mb : set->m) {
    int32_t rc = use(mb);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(mb.st->s));
  }
  return SLK_OK;
}

// ---- the staged round (synchronous): everything the lane kernel does not take, and splitters outside its range ----------------
// keys -> owners -> lookup -> taxa back on the compute streams: src_keys[a] holds a's keys sorted by owner (set->cnt[a][b] of them
// for b); afterwards m[a].taxa holds the answers in the same order (on a's stream).
int32_t staged_lookup(slk_shardset *set, const std::vector<const void *> &src_keys) {
  const int W = set->n;
  const Layout lay(set->cnt);
  std::vector<void *> recv(W), taxa(W);
  std::vector<const void *> found(W);
  std::vector<hipEvent_t> ev_sent(W), ev_found(W);
  std::vector<hipStream_t> str(W);
  for (int a = 0; a < W; a++) {
    Member &mb = set->m[a];
    int32_t rc = use(mb);
    if (rc) return rc;
    HIPCHK(mb.recv_keys.ensure(std::max<uint64_t>(lay.received[a], 1) * 8));
    HIPCHK(mb.found.ensure(std::max<uint64_t>(lay.received[a], 1) * 4));
    HIPCHK(mb.taxa.ensure(std::max<uint64_t>(lay.sent[a], 1) * 4));
    recv[a] = mb.recv_keys.p; found[a] = mb.found.p; taxa[a] = mb.taxa.p; str[a] = mb.st->s;
    HIPCHK(hipEventRecord(mb.ev_sent, mb.st->s));
    ev_sent[a] = mb.ev_sent; ev_found[a] = mb.ev_found;
  }
  // (the bounce path records ev_sent itself: give it events of its own to wait for)
  int32_t rc = exchange(set, src_keys, recv, set->cnt, 8, ncclInt64, str, ev_sent, [&](int a, int b) { return lay.soff[a][b]; },
                        [&](int b, int a) { return lay.roff[b][a]; });
  if (rc) return rc;
  for (int b = 0; b < W; b++) {
    Member &mb = set->m[b];
    rc = use(mb);
    if (rc) return rc;
    launch_lookup_coop(mb.ix->view(), mb.recv_keys.as<int64_t>(), lay.received[b], mb.found.as<int32_t>(), mb.st->s);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(mb.ev_found, mb.st->s));
  }
  // the way back is the transposed exchange: b returns to a what a sent, to the place it was sent from
  std::vector<std::vector<uint64_t>> back(W, std::vector<uint64_t>(W, 0));
  for (int a = 0; a < W; a++)
    for (int b = 0; b < W; b++) back[b][a] = set->cnt[a][b];
  return exchange(set, found, taxa, back, 4, ncclInt32, str, ev_found, [&](int b, int a) { return lay.roff[b][a]; },
                  [&](int a, int b) { return lay.soff[a][b]; });
}

int32_t ensure_outputs(slk_stream *st, uint64_t R, int32_t C) {
  HIPCHK(st->out_taxon.ensure((size_t)C * R * 4));
  HIPCHK(st->out_cls.ensure((size_t)C * R));
  HIPCHK(st->out_nd.ensure(R * 4));
  HIPCHK(st->out_tk.ensure(R * 4));
  HIPCHK(st->out_nh.ensure(R * 4));
  return SLK_OK;
}

// results of a member's staged batch to the caller's arrays (everything queued on its stream has been synchronised)
int32_t download_staged(Member &mb, const slk_shard_batch &B, int32_t C) {
  slk_stream *st = mb.st;
  const uint64_t R = B.R;
  int32_t rc = copy_out(st, B.out_taxon, st->out_taxon.p, (size_t)C * R * 4);
  if (!rc) rc = copy_out(st, B.out_classified, st->out_cls.p, (size_t)C * R);
  if (!rc && B.out_num_distinct) rc = copy_out(st, B.out_num_distinct, st->out_nd.p, R * 4);
  if (!rc && B.out_total_kmers) rc = copy_out(st, B.out_total_kmers, st->out_tk.p, R * 4);
  if (rc) return rc;
  if (B.out_hit_offsets) {
    rc = counts_to_offsets(st, st->out_nh.as<int32_t>(), R, B.out_hit_offsets, B.out_hits ? B.hits_capacity : ~0ULL);
    if (rc) return rc;
    const uint64_t n = B.out_hit_offsets[R];
    if (n && B.out_hits) {
      HIPCHK(st->out_items.ensure(n * sizeof(slk_hit)));
      launch_gather_hits(st->offsets.as<uint64_t>(), B.mate_offsets ? st->mate_offsets.as<uint64_t>() : nullptr, R, st->span_meta.as<int32_t>(),
                         st->span_taxon.as<int32_t>(), st->out_offsets.as<uint64_t>(), st->out_items.p, st->s);
      HIPCHK(hipGetLastError());
      rc = copy_out(st, B.out_hits, st->out_items.p, n * sizeof(slk_hit));
      if (rc) return rc;
    }
  }
  HIPCHK(hipStreamSynchronize(st->s));
  return SLK_OK;
}

int32_t staged_round(slk_shardset *set, const RoundArgs &A) {
  const int W = set->n;
  std::vector<uint64_t> total(W, 0), mate_total(W, 0);
  Thresholds thr{};
  memcpy(thr.v, A.thresholds, A.C * sizeof(double));
  // scan into span arrays, count the keys per owner
  for (int g = 0; g < W; g++) {
    Member &mb = set->m[g];
    const slk_shard_batch &B = A.batches[g];
    for (int d = 0; d < W; d++) mb.h_counts[d] = 0;
    if (B.R == 0) continue;
    int32_t rc = use(mb);
    if (rc) return rc;
    slk_stream *st = mb.st;
    rc = upload_reads(st, B.bases, B.offsets, B.mate_bases, B.mate_offsets, B.R, &total[g], &mate_total[g]);
    if (!rc) rc = ensure_scratch(st, span_slots(total[g], mate_total[g], B.R, B.mate_offsets != nullptr), B.R);
    if (!rc) rc = ensure_outputs(st, B.R, A.C);
    if (rc) return rc;
    const bool paired = B.mate_offsets != nullptr;
    FusedArgs F{};
    F.P = mb.ix->sp; F.bases = st->bases.as<uint8_t>(); F.offsets = st->offsets.as<uint64_t>();
    F.mate_bases = paired ? st->mate_bases.as<uint8_t>() : nullptr; F.mate_offsets = paired ? st->mate_offsets.as<uint64_t>() : nullptr;
    F.R = B.R; F.span_keys = st->span_keys.as<uint64_t>(); F.span_meta = st->span_meta.as<int32_t>(); F.span_count = st->span_count.as<int32_t>();
    F.status = st->d_status;
    if (mb.ix->sp.w <= 32) launch_fused(MODE_SPANS, F, st->s);
    else launch_scan(mb.ix->sp, F.bases, F.offsets, F.mate_bases, F.mate_offsets, B.R, F.span_keys, F.span_meta, F.span_count, st->s);
    HIPCHK(hipGetLastError());
    HIPCHK(mb.counts.ensure((size_t)W * 8));
    HIPCHK(mb.starts.ensure((size_t)W * 8));
    HIPCHK(hipMemsetAsync(mb.counts.p, 0, (size_t)W * 8, st->s));
    const unsigned blocks = (unsigned)std::min<uint64_t>((B.R + 3) / 4, 8192);
    hipLaunchKernelGGL(collect_keys_kernel, dim3(blocks), dim3(256), 0, st->s, F.offsets, F.mate_offsets, B.R, F.span_keys, F.span_meta,
                       F.span_count, (uint32_t)W, false, mb.counts.as<unsigned long long>(), (const uint64_t *)nullptr, (int64_t *)nullptr,
                       (uint64_t *)nullptr, (int32_t *)nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(mb.h_counts, mb.counts.p, (size_t)W * 8, hipMemcpyDeviceToHost, st->s));
  }
  int32_t rc = sync_all(set);
  if (rc) return rc;
  std::vector<const void *> src(W, nullptr);
  for (int g = 0; g < W; g++) {
    Member &mb = set->m[g];
    const slk_shard_batch &B = A.batches[g];
    uint64_t n = 0;
    std::vector<uint64_t> starts(W);
    for (int d = 0; d < W; d++) { set->cnt[g][d] = mb.h_counts[d]; starts[d] = n; n += mb.h_counts[d]; }
    rc = use(mb);
    if (rc) return rc;
    HIPCHK(mb.out_keys.ensure(std::max<uint64_t>(n, 1) * 8));
    HIPCHK(mb.slots.ensure(std::max<uint64_t>(n, 1) * 8));
    src[g] = mb.out_keys.p;
    if (B.R == 0) continue;
    slk_stream *st = mb.st;
    const bool paired = B.mate_offsets != nullptr;
    rc = copy_in(st, mb.starts.p, starts.data(), (size_t)W * 8);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(mb.counts.p, 0, (size_t)W * 8, st->s));
    const unsigned blocks = (unsigned)std::min<uint64_t>((B.R + 3) / 4, 8192);
    hipLaunchKernelGGL(collect_keys_kernel, dim3(blocks), dim3(256), 0, st->s, st->offsets.as<uint64_t>(),
                       paired ? st->mate_offsets.as<uint64_t>() : (const uint64_t *)nullptr, B.R, st->span_keys.as<uint64_t>(),
                       st->span_meta.as<int32_t>(), st->span_count.as<int32_t>(), (uint32_t)W, true, mb.counts.as<unsigned long long>(),
                       mb.starts.as<uint64_t>(), mb.out_keys.as<int64_t>(), mb.slots.as<uint64_t>(), st->span_taxon.as<int32_t>());
    HIPCHK(hipGetLastError());
  }
  rc = staged_lookup(set, src);
  if (rc) return rc;
  for (int g = 0; g < W; g++) {
    Member &mb = set->m[g];
    const slk_shard_batch &B = A.batches[g];
    if (B.R == 0) continue;
    rc = use(mb);
    if (rc) return rc;
    slk_stream *st = mb.st;
    const bool paired = B.mate_offsets != nullptr;
    uint64_t n = 0;
    for (int d = 0; d < W; d++) n += set->cnt[g][d];
    if (n) {
      hipLaunchKernelGGL(scatter_taxa_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, st->s, mb.slots.as<uint64_t>(),
                         mb.taxa.as<int32_t>(), n, st->span_taxon.as<int32_t>());
      HIPCHK(hipGetLastError());
    }
    // the key slots are dead after the exchange: the unbounded per-fragment taxon map of the classify kernel reuses them
    launch_classify(mb.ix->d_parents, mb.ix->d_nodes_orig, mb.ix->T, st->offsets.as<uint64_t>(), paired ? st->mate_offsets.as<uint64_t>() : nullptr,
                    B.R, st->span_meta.as<int32_t>(), st->span_taxon.as<int32_t>(), st->span_count.as<int32_t>(), st->span_keys.as<uint64_t>(),
                    A.min_hit_groups, thr, A.C, B.R, st->out_taxon.as<int32_t>(), st->out_cls.as<uint8_t>(), st->out_nd.as<int32_t>(),
                    st->out_tk.as<int32_t>(), st->out_nh.as<int32_t>(), nullptr, st->s);
    HIPCHK(hipGetLastError());
  }
  rc = sync_all(set);
  if (rc) return rc;
  for (int g = 0; g < W; g++) {
    if (A.batches[g].R == 0) continue;
    rc = use(set->m[g]);
    if (!rc) rc = download_staged(set->m[g], A.batches[g], A.C);
    if (rc) return rc;
  }
  return SLK_OK;
}

// ---- the pipelined rounds -----------------------------------------------------------------------------------------------------
// Entries of an owner's send region for a batch: the expected number of super-mers of random sequence -- 2 / (w + 1) per k-mer
// window -- spread evenly by the hash, a fifth on top, and a chunk per wave that may hold one half filled at the kernel's end.  A
// region that is too small all the same (sequence built to change its minimizer at every window has up to one per window) sends
// its batch to the staged route.
uint64_t region_capacity(const slk_index *ix, uint64_t bases, uint64_t R, int W) {
  const uint32_t chunk = slk_shard_chunk((uint32_t)W);
  const uint64_t shortfall = R * (uint64_t)(ix->sp.k - 1);
  const uint64_t windows = bases > shortfall ? bases - shortfall : 0;
  const double expect = 2.0 / (ix->sp.w + 1) * (double)windows + (double)R;
  const uint64_t tiles = (R + 63) / 64;
  uint64_t cap = (uint64_t)(expect / W * 1.2) + 4096 + (uint64_t)chunk * std::min<uint64_t>(tiles, 8192);
  cap = (cap + chunk - 1) / chunk * chunk;
  const uint64_t limit = ((1ull << 32) - 1) / chunk * chunk - chunk;
  return std::min(cap, limit);
}

struct Pipe {
  slk_shardset *set;
  slk_shard_batch *batches;   // [rounds][W]
  int rounds;
  bool on_device;             // the batches' pointers are device pointers on the members' devices (no hit lists)
  int32_t min_hit_groups, C;
  const double *thresholds;
  int W;
  std::vector<std::vector<uint32_t>> deferred;   // [round * W + g]
  const slk_shard_batch &B(int r, int g) const { return batches[(size_t)r * W + g]; }
  bool hits(int r, int g) const { const slk_shard_batch &b = B(r, g); return !on_device && b.out_hit_offsets && b.out_hits; }
};

// reads of round r to the members' slots (host-pointer rounds), on the upload streams
int32_t upload_round(Pipe &P, int r) {
  for (int g = 0; g < P.W; g++) {
    Member &mb = P.set->m[g];
    Slot &sl = mb.slot[r % NSLOT];
    const slk_shard_batch &B = P.B(r, g);
    sl.failed = false;
    sl.lists = slk_shard_lists{};
    sl.lists.R = B.R;
    if (B.R == 0) continue;
    int32_t rc = use(mb);
    if (rc) return rc;
    const bool paired = B.mate_offsets != nullptr;
    if (P.on_device) {
      sl.lists.d_bases = B.bases; sl.lists.d_offsets = B.offsets; sl.lists.d_mate_bases = B.mate_bases; sl.lists.d_mate_offsets = B.mate_offsets;
      // (the totals are the last offsets: two words per batch, read here once)
      HIPCHK(hipMemcpyAsync(&sl.h_cursors[0], B.offsets + B.R, 8, hipMemcpyDeviceToHost, mb.us));
      if (paired) HIPCHK(hipMemcpyAsync(&sl.h_cursors[1], B.mate_offsets + B.R, 8, hipMemcpyDeviceToHost, mb.us));
      HIPCHK(hipStreamSynchronize(mb.us));
      sl.total = sl.h_cursors[0]; sl.mate_total = paired ? sl.h_cursors[1] : 0;
    } else {
      for (uint64_t i = 0; i < B.R; i++)
        if (B.offsets[i + 1] < B.offsets[i] || B.offsets[i + 1] - B.offsets[i] > 0x7fffffffULL ||
            (paired && (B.mate_offsets[i + 1] < B.mate_offsets[i] || B.mate_offsets[i + 1] - B.mate_offsets[i] > 0x7fffffffULL)))
          return fail(SLK_E_INVALID, "offsets (and mate_offsets) must be non-decreasing with reads shorter than 2^31 (member %d, read %llu)", g, (unsigned long long)i);
      sl.total = B.offsets[B.R]; sl.mate_total = paired ? B.mate_offsets[B.R] : 0;
      HIPCHK(sl.bases.ensure(std::max<uint64_t>(sl.total, 1)));
      HIPCHK(sl.offsets.ensure((B.R + 1) * 8));
      rc = copy_in(&mb.staging_u, mb.us, sl.bases.p, B.bases, sl.total);
      if (!rc) rc = copy_in(&mb.staging_u, mb.us, sl.offsets.p, B.offsets, (B.R + 1) * 8);
      if (!rc && paired) {
        HIPCHK(sl.mate_bases.ensure(std::max<uint64_t>(sl.mate_total, 1)));
        HIPCHK(sl.mate_offsets.ensure((B.R + 1) * 8));
        rc = copy_in(&mb.staging_u, mb.us, sl.mate_bases.p, B.mate_bases, sl.mate_total);
        if (!rc) rc = copy_in(&mb.staging_u, mb.us, sl.mate_offsets.p, B.mate_offsets, (B.R + 1) * 8);
      }
      if (rc) return rc;
      sl.lists.d_bases = sl.bases.as<uint8_t>(); sl.lists.d_offsets = sl.offsets.as<uint64_t>();
      sl.lists.d_mate_bases = paired ? sl.mate_bases.as<uint8_t>() : nullptr; sl.lists.d_mate_offsets = paired ? sl.mate_offsets.as<uint64_t>() : nullptr;
    }
    HIPCHK(hipEventRecord(sl.ev_up, mb.us));
  }
  return SLK_OK;
}

// the slot's lists for round r's EMIT (sizes follow the batch)
int32_t prepare_lists(Pipe &P, int r, int g) {
  Member &mb = P.set->m[g];
  Slot &sl = mb.slot[r % NSLOT];
  Big &bg = mb.big[r % NBIG];
  const slk_shard_batch &B = P.B(r, g);
  const int W = P.W;
  slk_shard_lists &L = sl.lists;
  const bool paired = B.mate_offsets != nullptr;
  L.n_shards = (uint32_t)W;
  L.total_bases = sl.total; L.total_mate_bases = sl.mate_total;
  L.capacity_per_owner = region_capacity(mb.ix, sl.total + sl.mate_total, B.R, W);
  const uint64_t ents = (uint64_t)W * L.capacity_per_owner;
  const uint64_t rows = slk_shard_batch_rows(sl.total, sl.mate_total, B.R, paired);
  HIPCHK(bg.send_keys.ensure(ents * 8));
  HIPCHK(sl.send_meta.ensure(ents * 4));
  HIPCHK(sl.taxa.ensure(ents * 4));
  HIPCHK(sl.cursors.ensure(((size_t)W + 3) * 8));   // (+ tile draws and the count of handed-back fragments)
  HIPCHK(sl.log.ensure(rows * W * 16));
  HIPCHK(sl.tile_rows.ensure(((B.R + 63) / 64 + 1) * 8));
  HIPCHK(sl.read_info.ensure(std::max<uint64_t>(B.R, 1) * 8));
  HIPCHK(sl.defer.ensure(std::max<uint64_t>(B.R, 1) * 4));
  L.d_send_keys = bg.send_keys.as<int64_t>(); L.d_send_meta = sl.send_meta.as<uint32_t>(); L.d_cursors = sl.cursors.as<uint64_t>();
  L.d_batch_log = sl.log.as<uint32_t>(); L.d_tile_rows = sl.tile_rows.as<uint32_t>(); L.d_read_info = sl.read_info.as<int32_t>();
  L.d_defer = sl.defer.as<int32_t>();
  if (P.hits(r, g)) {
    const uint64_t slots = span_slots(sl.total, sl.mate_total, B.R, paired);
    HIPCHK(sl.span_meta.ensure(slots * 4));
    HIPCHK(sl.span_taxon.ensure(slots * 4));
    HIPCHK(sl.span_count.ensure((B.R + 1) * 4));
    L.d_span_meta = sl.span_meta.as<int32_t>(); L.d_span_taxon = sl.span_taxon.as<int32_t>(); L.d_span_count = sl.span_count.as<int32_t>();
  }
  if (!P.on_device) {
    HIPCHK(sl.out_taxon.ensure((size_t)P.C * B.R * 4));
    HIPCHK(sl.out_cls.ensure((size_t)P.C * B.R));
    HIPCHK(sl.out_nd.ensure(B.R * 4));
    HIPCHK(sl.out_tk.ensure(B.R * 4));
  }
  HIPCHK(sl.out_nh.ensure(B.R * 4));
  return SLK_OK;
}

// Step t on every member: EMIT(round t) + LOOKUP(round t - 2) + APPLY(round t - 4) as ONE kernel on the member's compute stream.
int32_t launch_step(Pipe &P, int t, const std::vector<std::vector<uint64_t>> &received) {
  const int W = P.W;
  for (int g = 0; g < W; g++) {
    Member &mb = P.set->m[g];
    int32_t rc = use(mb);
    if (rc) return rc;
    const slk_shard_lists *emit = nullptr, *alists = nullptr;
    slk_shard_lookup lk{};
    slk_shard_results res{};
    const slk_shard_lookup *lookup = nullptr;
    const slk_shard_results *apply = nullptr;
    if (t < P.rounds && P.B(t, g).R) {
      Slot &sl = mb.slot[t % NSLOT];
      rc = prepare_lists(P, t, g);
      if (rc) return rc;
      HIPCHK(hipStreamWaitEvent(mb.st->s, sl.ev_up, 0));
      emit = &sl.lists;
    }
    const int rl = t - 2, ra = t - 4;
    if (rl >= 0 && rl < P.rounds) {
      // every member's exchange of round rl has been issued: this member's keys are here behind ITS ev_keys, and nobody reads
      // the rings' older contents any more behind the others'
      for (int d = 0; d < W; d++) HIPCHK(hipStreamWaitEvent(mb.st->s, P.set->m[d].slot[rl % NSLOT].ev_keys, 0));
      if (received[rl][g]) {
        Big &bg = mb.big[rl % NBIG];
        lk.d_keys = bg.recv_keys.as<int64_t>(); lk.n = received[rl][g]; lk.d_out_taxa = bg.found.as<int32_t>();
        if (W == 1) {   // a set of one: nothing travels -- the keys are answered where they were written, the answers land where the replay reads them
          lk.d_keys = bg.send_keys.as<int64_t>();
          lk.d_out_taxa = mb.slot[rl % NSLOT].taxa.as<int32_t>();
        }
        lookup = &lk;
      }
    }
    if (ra >= 0 && ra < P.rounds) {
      for (int d = 0; d < W; d++) HIPCHK(hipStreamWaitEvent(mb.st->s, P.set->m[d].slot[ra % NSLOT].ev_taxa, 0));
      Slot &sl = mb.slot[ra % NSLOT];
      const slk_shard_batch &B = P.B(ra, g);
      if (B.R && !sl.failed) {
        alists = &sl.lists;
        res.d_taxa = sl.taxa.as<int32_t>(); res.min_hit_groups = P.min_hit_groups; res.C = P.C; res.thresholds = P.thresholds;
        res.d_out_taxon = P.on_device ? B.out_taxon : sl.out_taxon.as<int32_t>();
        res.d_out_classified = P.on_device ? B.out_classified : sl.out_cls.as<uint8_t>();
        res.d_out_num_distinct = P.on_device ? B.out_num_distinct : sl.out_nd.as<int32_t>();
        res.d_out_total_kmers = P.on_device ? B.out_total_kmers : sl.out_tk.as<int32_t>();
        res.d_out_num_hits = sl.out_nh.as<int32_t>();
        apply = &res;
      }
    }
    if (emit && apply && (emit->d_span_meta == nullptr) != (alists->d_span_meta == nullptr)) {
      // (one step's kernel writes hit lists for both its batches or for neither: the replay of a batch that differs goes alone)
      rc = slk_shard_step_device(mb.ix, mb.st, nullptr, nullptr, alists, apply);
      if (rc) return rc;
      alists = nullptr; apply = nullptr;
    }
    if (emit || lookup || apply) {
      rc = slk_shard_step_device(mb.ix, mb.st, emit, lookup, alists, apply);
      if (rc) return rc;
    }
    if (emit) {
      Slot &sl = mb.slot[t % NSLOT];
      HIPCHK(hipMemcpyAsync(sl.h_cursors, sl.cursors.p, ((size_t)W + 3) * 8, hipMemcpyDeviceToHost, mb.st->s));
    }
    HIPCHK(hipEventRecord(mb.ev_step[t % NSLOT], mb.st->s));
  }
  return SLK_OK;
}

// keys of round r to their owners, taxa of round r - 2 back, on the exchange streams (behind step r's / step r's kernels)
int32_t exchange_after_step(Pipe &P, int t, std::vector<std::vector<uint64_t>> &received, std::vector<std::vector<std::vector<uint64_t>>> &sent) {
  const int W = P.W;
  slk_shardset *set = P.set;
  std::vector<hipStream_t> str(W);
  std::vector<hipEvent_t> ev(W);
  for (int g = 0; g < W; g++) {
    str[g] = set->m[g].xs; ev[g] = set->m[g].ev_step[t % NSLOT];
    int32_t rc = use(set->m[g]);
    if (rc) return rc;
    HIPCHK(hipEventSynchronize(ev[g]));   // the step's one host wait (the next step is queued behind it already)
  }
  if (t < P.rounds) {   // keys of round t: split sizes from the cursors
    const int r = t;
    for (int g = 0; g < W; g++) {
      Member &mb = set->m[g];
      Slot &sl = mb.slot[r % NSLOT];
      int32_t rc = use(mb);
      if (rc) return rc;
      for (int d = 0; d < W; d++) sent[r][g][d] = 0;
      if (P.B(r, g).R == 0) continue;
      bool over = false;
      for (int d = 0; d < W; d++) over = over || sl.h_cursors[d] > sl.lists.capacity_per_owner;
      if (over) {   // a region was too small: nothing of this batch travels; all of it takes the staged route
        sl.failed = true;
        mb.st->queued.clear();
        *mb.st->h_status = 0;
        HIPCHK(hipMemsetAsync(mb.st->d_status, 0, sizeof(int32_t), mb.xs));
        continue;
      }
      for (int d = 0; d < W; d++) sent[r][g][d] = sl.h_cursors[d];
    }
    const Layout lay(sent[r]);
    std::vector<const void *> src(W);
    std::vector<void *> dst(W);
    for (int g = 0; g < W; g++) {
      Member &mb = set->m[g];
      int32_t rc = use(mb);
      if (rc) return rc;
      Big &bg = mb.big[r % NBIG];
      HIPCHK(bg.recv_keys.ensure(std::max<uint64_t>(lay.received[g], 1) * 8));
      HIPCHK(bg.found.ensure(std::max<uint64_t>(lay.received[g], 1) * 4));
      src[g] = bg.send_keys.p; dst[g] = bg.recv_keys.p;
      received[r][g] = lay.received[g];
    }
    set->cnt = sent[r];
    int32_t rc = SLK_OK;
    if (W > 1)
      rc = exchange(set, src, dst, sent[r], 8, ncclInt64, str, ev,
                    [&](int a, int b) { return (uint64_t)b * set->m[a].slot[r % NSLOT].lists.capacity_per_owner; },
                    [&](int b, int a) { return lay.roff[b][a]; });
    else { rc = use(set->m[0]); if (!rc) HIPCHK(hipStreamWaitEvent(str[0], ev[0], 0)); }
    if (rc) return rc;
    for (int g = 0; g < W; g++) {
      int32_t rc2 = use(set->m[g]);
      if (rc2) return rc2;
      HIPCHK(hipEventRecord(set->m[g].slot[r % NSLOT].ev_keys, set->m[g].xs));
    }
  }
  const int rb = t - 2;   // its lookups ran in step t
  if (rb >= 0 && rb < P.rounds) {
    const Layout lay(sent[rb]);
    std::vector<std::vector<uint64_t>> back(W, std::vector<uint64_t>(W, 0));
    for (int a = 0; a < W; a++)
      for (int b = 0; b < W; b++) back[b][a] = sent[rb][a][b];
    std::vector<const void *> src(W);
    std::vector<void *> dst(W);
    for (int g = 0; g < W; g++) { src[g] = set->m[g].big[rb % NBIG].found.p; dst[g] = set->m[g].slot[rb % NSLOT].taxa.p; }
    int32_t rc = SLK_OK;
    if (W > 1)
      rc = exchange(set, src, dst, back, 4, ncclInt32, str, ev, [&](int b, int a) { return lay.roff[b][a]; },
                    [&](int a, int b) { return (uint64_t)b * set->m[a].slot[rb % NSLOT].lists.capacity_per_owner; });
    else { rc = use(set->m[0]); if (!rc) HIPCHK(hipStreamWaitEvent(str[0], ev[0], 0)); }
    if (rc) return rc;
    for (int g = 0; g < W; g++) {
      int32_t rc2 = use(set->m[g]);
      if (rc2) return rc2;
      HIPCHK(hipEventRecord(set->m[g].slot[rb % NSLOT].ev_taxa, set->m[g].xs));
    }
  }
  return SLK_OK;
}

// Round r's APPLY has finished (its step's event was waited for): which fragments it handed back, and -- host-pointer rounds -- its
// results to the caller's arrays, on the download stream beside the running step.
int32_t collect_round(Pipe &P, int r) {
  const int W = P.W;
  for (int g = 0; g < W; g++) {
    Member &mb = P.set->m[g];
    Slot &sl = mb.slot[r % NSLOT];
    slk_shard_batch &B = P.batches[(size_t)r * W + g];
    std::vector<uint32_t> &dl = P.deferred[(size_t)r * W + g];
    dl.clear();
    if (B.R == 0) continue;
    int32_t rc = use(mb);
    if (rc) return rc;
    if (sl.failed) {
      for (uint64_t i = 0; i < B.R; i++) dl.push_back((uint32_t)i);
      if (!P.on_device && B.out_hit_offsets) memset(B.out_hit_offsets, 0, (B.R + 1) * 8);
      continue;
    }
    uint64_t n_def = 0;
    HIPCHK(hipMemcpyAsync(&n_def, sl.cursors.as<uint64_t>() + W + 2, 8, hipMemcpyDeviceToHost, mb.ds));
    HIPCHK(hipStreamSynchronize(mb.ds));
    if (n_def) {
      std::vector<int32_t> defer(B.R);
      rc = copy_out(&mb.staging_d, mb.ds, defer.data(), sl.defer.p, B.R * 4);
      if (rc) return rc;
      for (uint64_t i = 0; i < B.R; i++)
        if (defer[i]) dl.push_back((uint32_t)i);
    }
    if (P.on_device) continue;
    rc = copy_out(&mb.staging_d, mb.ds, B.out_taxon, sl.out_taxon.p, (size_t)P.C * B.R * 4);
    if (!rc) rc = copy_out(&mb.staging_d, mb.ds, B.out_classified, sl.out_cls.p, (size_t)P.C * B.R);
    if (!rc && B.out_num_distinct) rc = copy_out(&mb.staging_d, mb.ds, B.out_num_distinct, sl.out_nd.p, B.R * 4);
    if (!rc && B.out_total_kmers) rc = copy_out(&mb.staging_d, mb.ds, B.out_total_kmers, sl.out_tk.p, B.R * 4);
    if (rc) return rc;
    if (B.out_hit_offsets) {
      std::vector<int32_t> counts(B.R);
      rc = copy_out(&mb.staging_d, mb.ds, counts.data(), sl.out_nh.p, B.R * 4);
      if (rc) return rc;
      B.out_hit_offsets[0] = 0;
      for (uint64_t i = 0; i < B.R; i++) B.out_hit_offsets[i + 1] = B.out_hit_offsets[i] + (uint64_t)counts[i];
      const uint64_t n = B.out_hit_offsets[B.R];
      if (B.out_hits && n > B.hits_capacity)
        return fail(SLK_E_CAPACITY, "output needs %llu entries, capacity is %llu", (unsigned long long)n, (unsigned long long)B.hits_capacity);
      if (n && B.out_hits) {
        HIPCHK(sl.out_offsets.ensure((B.R + 1) * 8));
        HIPCHK(sl.out_items.ensure(n * sizeof(slk_hit)));
        rc = copy_in(&mb.staging_d, mb.ds, sl.out_offsets.p, B.out_hit_offsets, (B.R + 1) * 8);
        if (rc) return rc;
        launch_gather_hits(sl.lists.d_offsets, sl.lists.d_mate_offsets, B.R, sl.span_meta.as<int32_t>(), sl.span_taxon.as<int32_t>(),
                           sl.out_offsets.as<uint64_t>(), sl.out_items.p, mb.ds);
        HIPCHK(hipGetLastError());
        rc = copy_out(&mb.staging_d, mb.ds, B.out_hits, sl.out_items.p, n * sizeof(slk_hit));
        if (rc) return rc;
      }
    }
  }
  return SLK_OK;
}

// The fragments round r's lane kernels handed back, as a batch of their own per member, through the staged round; their rows replace
// what the pipeline left in the caller's arrays (hit lists are rebuilt with the longer lists spliced in).
int32_t finish_deferred(Pipe &P, int r) {
  const int W = P.W;
  const int32_t C = P.C;
  bool any = false;
  for (int g = 0; g < W; g++) any = any || !P.deferred[(size_t)r * W + g].empty();
  if (!any) return SLK_OK;
  struct Sub {
    std::vector<uint8_t> bases, mates, cls;
    std::vector<uint64_t> offs, moffs, hit_offs;
    std::vector<int32_t> taxon, nd, tk;
    std::vector<slk_hit> hits;
  };
  std::vector<Sub> sub(W);
  std::vector<slk_shard_batch> sb(W);
  // device-resident rounds: the handed-back fragments come down first (their offsets, then their bases piece by piece: they are few)
  std::vector<std::vector<uint64_t>> h_offs(W), h_moffs(W);
  for (int g = 0; g < W; g++) {
    const slk_shard_batch &B = P.B(r, g);
    const std::vector<uint32_t> &dl = P.deferred[(size_t)r * W + g];
    Sub &s = sub[g];
    const size_t n = dl.size();
    sb[g] = slk_shard_batch{};
    if (!n) continue;
    const bool paired = B.mate_offsets != nullptr;
    const uint64_t *offs = B.offsets, *moffs = B.mate_offsets;
    if (P.on_device) {
      int32_t rc = use(P.set->m[g]);
      if (rc) return rc;
      h_offs[g].resize(B.R + 1);
      HIPCHK(hipMemcpy(h_offs[g].data(), B.offsets, (B.R + 1) * 8, hipMemcpyDeviceToHost));
      offs = h_offs[g].data();
      if (paired) {
        h_moffs[g].resize(B.R + 1);
        HIPCHK(hipMemcpy(h_moffs[g].data(), B.mate_offsets, (B.R + 1) * 8, hipMemcpyDeviceToHost));
        moffs = h_moffs[g].data();
      }
    }
    s.offs.assign(1, 0);
    if (paired) s.moffs.assign(1, 0);
    for (uint32_t i : dl) {
      const size_t at = s.bases.size(), len = offs[i + 1] - offs[i];
      s.bases.resize(at + len);
      if (len) {
        if (P.on_device) HIPCHK(hipMemcpy(s.bases.data() + at, B.bases + offs[i], len, hipMemcpyDeviceToHost));
        else memcpy(s.bases.data() + at, B.bases + offs[i], len);
      }
      s.offs.push_back(s.bases.size());
      if (paired) {
        const size_t mat = s.mates.size(), mlen = moffs[i + 1] - moffs[i];
        s.mates.resize(mat + mlen);
        if (mlen) {
          if (P.on_device) HIPCHK(hipMemcpy(s.mates.data() + mat, B.mate_bases + moffs[i], mlen, hipMemcpyDeviceToHost));
          else memcpy(s.mates.data() + mat, B.mate_bases + moffs[i], mlen);
        }
        s.moffs.push_back(s.mates.size());
      }
    }
    if (s.bases.empty()) s.bases.push_back('N');
    if (paired && s.mates.empty()) s.mates.push_back('N');
    s.taxon.resize((size_t)C * n); s.cls.resize((size_t)C * n); s.nd.resize(n); s.tk.resize(n); s.hit_offs.resize(n + 1);
    const bool want_hits = P.hits(r, g);
    const size_t cap = s.bases.size() + s.mates.size() + n + 1;
    if (want_hits) s.hits.resize(cap);
    sb[g] = slk_shard_batch{s.bases.data(), s.offs.data(), paired ? s.mates.data() : nullptr, paired ? s.moffs.data() : nullptr, n,
                            s.taxon.data(), s.cls.data(), s.nd.data(), s.tk.data(), (!P.on_device && B.out_hit_offsets) ? s.hit_offs.data() : nullptr,
                            want_hits ? s.hits.data() : nullptr, cap};
  }
  RoundArgs A2{sb.data(), P.min_hit_groups, P.thresholds, C};
  int32_t rc = staged_round(P.set, A2);
  if (rc) return rc;
  for (int g = 0; g < W; g++) {
    slk_shard_batch &B = P.batches[(size_t)r * W + g];
    const std::vector<uint32_t> &dl = P.deferred[(size_t)r * W + g];
    const Sub &s = sub[g];
    const size_t n = dl.size();
    if (!n) continue;
    if (P.on_device) {   // (row by row: the handed-back fragments are few)
      rc = use(P.set->m[g]);
      if (rc) return rc;
      for (size_t i = 0; i < n; i++) {
        const uint32_t q = dl[i];
        for (int32_t c = 0; c < C; c++) {
          HIPCHK(hipMemcpy(B.out_taxon + (size_t)c * B.R + q, &s.taxon[(size_t)c * n + i], 4, hipMemcpyHostToDevice));
          HIPCHK(hipMemcpy(B.out_classified + (size_t)c * B.R + q, &s.cls[(size_t)c * n + i], 1, hipMemcpyHostToDevice));
        }
        if (B.out_num_distinct) HIPCHK(hipMemcpy(B.out_num_distinct + q, &s.nd[i], 4, hipMemcpyHostToDevice));
        if (B.out_total_kmers) HIPCHK(hipMemcpy(B.out_total_kmers + q, &s.tk[i], 4, hipMemcpyHostToDevice));
      }
      continue;
    }
    for (size_t i = 0; i < n; i++) {
      const uint32_t q = dl[i];
      for (int32_t c = 0; c < C; c++) {
        B.out_taxon[(size_t)c * B.R + q] = s.taxon[(size_t)c * n + i];
        B.out_classified[(size_t)c * B.R + q] = s.cls[(size_t)c * n + i];
      }
      if (B.out_num_distinct) B.out_num_distinct[q] = s.nd[i];
      if (B.out_total_kmers) B.out_total_kmers[q] = s.tk[i];
    }
    if (!B.out_hit_offsets) continue;
    // hit lists: the pipeline left the handed-back fragments without spans; their lists from the staged round are spliced in
    std::vector<uint64_t> offs(B.R + 1, 0);
    {
      size_t i = 0;
      for (uint64_t q = 0; q < B.R; q++) {
        uint64_t len = B.out_hit_offsets[q + 1] - B.out_hit_offsets[q];
        if (i < n && dl[i] == q) { len = s.hit_offs[i + 1] - s.hit_offs[i]; i++; }
        offs[q + 1] = offs[q] + len;
      }
    }
    if (B.out_hits) {
      if (offs[B.R] > B.hits_capacity)
        return fail(SLK_E_CAPACITY, "output needs %llu entries, capacity is %llu", (unsigned long long)offs[B.R], (unsigned long long)B.hits_capacity);
      std::vector<slk_hit> merged(offs[B.R]);
      size_t i = 0;
      for (uint64_t q = 0; q < B.R; q++) {
        const uint64_t len = offs[q + 1] - offs[q];
        if (i < n && dl[i] == q) { memcpy(merged.data() + offs[q], s.hits.data() + s.hit_offs[i], len * sizeof(slk_hit)); i++; }
        else if (len) memcpy(merged.data() + offs[q], B.out_hits + B.out_hit_offsets[q], len * sizeof(slk_hit));
      }
      if (!merged.empty()) memcpy(B.out_hits, merged.data(), merged.size() * sizeof(slk_hit));
    }
    memcpy(B.out_hit_offsets, offs.data(), (B.R + 1) * sizeof(uint64_t));
  }
  return SLK_OK;
}

int32_t run_rounds(slk_shardset *set, slk_shard_batch *batches, int rounds, bool on_device, int32_t min_hit_groups, const double *thresholds, int32_t C) {
  const int W = set->n;
  Pipe P{set, batches, rounds, on_device, min_hit_groups, C, thresholds, W, {}};
  P.deferred.assign((size_t)rounds * W, {});
  bool fast = true;
  for (int g = 0; g < W; g++) fast = fast && lane_path_ok(set->m[g].ix);
  if (!fast) {   // a splitter outside the lane kernel's range: everything takes the staged kernels, round by round
    if (on_device) return fail(SLK_E_UNSUPPORTED, "device-resident rounds need a splitter the lane kernel takes");
    for (int r = 0; r < rounds; r++) {
      RoundArgs A{batches + (size_t)r * W, min_hit_groups, thresholds, C};
      int32_t rc = staged_round(set, A);
      if (rc) return rc;
    }
    return SLK_OK;
  }
  std::vector<std::vector<uint64_t>> received(rounds, std::vector<uint64_t>(W, 0));
  std::vector<std::vector<std::vector<uint64_t>>> sent(rounds, std::vector<std::vector<uint64_t>>(W, std::vector<uint64_t>(W, 0)));
  int32_t rc = upload_round(P, 0);
  if (rc) return rc;
  for (int t = 0; t < rounds + 4; t++) {
    rc = launch_step(P, t, received);
    if (rc) return rc;
    if (t >= 1) {
      rc = exchange_after_step(P, t - 1, received, sent);   // waits for step t - 1, with step t queued behind it
      if (rc) return rc;
      const int done = t - 1 - 4;                            // APPLY(done) ran in step t - 1
      if (done >= 0 && done < rounds) {
        rc = collect_round(P, done);
        if (rc) return rc;
      }
    }
    if (t + 1 < rounds) {   // (beside step t; the slot it takes held round t + 1 - NSLOT, collected just now at the latest)
      rc = upload_round(P, t + 1);
      if (rc) return rc;
    }
  }
  // (the last step, rounds + 3, carried the last APPLY)
  for (Member &mb : set->m) {
    rc = use(mb);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(mb.st->s));
    HIPCHK(hipStreamSynchronize(mb.xs));
  }
  rc = collect_round(P, rounds - 1);
  if (rc) return rc;
  for (Member &mb : set->m) {   // (the step kernels are not re-runnable by check_status: deferrals are settled below)
    rc = use(mb);
    if (rc) return rc;
    mb.st->queued.clear();
    if (*mb.st->h_status & ~2) return fail(SLK_E_HIP, "device status %d", *mb.st->h_status);
    *mb.st->h_status = 0;
    HIPCHK(hipMemsetAsync(mb.st->d_status, 0, sizeof(int32_t), mb.st->s));
    HIPCHK(hipStreamSynchronize(mb.st->s));
  }
  for (int r = 0; r < rounds; r++) {
    rc = finish_deferred(P, r);
    if (rc) return rc;
  }
  return SLK_OK;
}

int32_t check_batches(slk_shardset *set, slk_shard_batch *batches, int32_t rounds, bool on_device, const double *thresholds, int32_t C) {
  if (!set || !batches || rounds < 1) return fail(SLK_E_INVALID, "null argument");
  if (C < 1 || C > MAX_THRESHOLDS || !thresholds) return fail(SLK_E_INVALID, "need 1..%d thresholds", MAX_THRESHOLDS);
  for (int64_t i = 0; i < (int64_t)rounds * set->n; i++) {
    slk_shard_batch &B = batches[i];
    const int g = (int)(i % set->n);
    if (B.R && (!B.bases || !B.offsets || !B.out_taxon || !B.out_classified)) return fail(SLK_E_INVALID, "null argument (member %d)", g);
    if ((B.mate_bases == nullptr) != (B.mate_offsets == nullptr)) return fail(SLK_E_INVALID, "mate_bases and mate_offsets must be given together");
    if (B.R >= 0xFFFFFFFFull) return fail(SLK_E_INVALID, "a batch holds fewer than 2^32 fragments");
    if (on_device && (B.out_hit_offsets || B.out_hits)) return fail(SLK_E_UNSUPPORTED, "device-resident rounds return no hit lists");
    if (!on_device && B.out_hit_offsets) B.out_hit_offsets[0] = 0;
  }
  return SLK_OK;
}

}  // namespace

extern "C" {

int32_t slk_shardset_create(slk_index *const *members, int32_t n_members, int32_t exchange, slk_shardset **out) {
  if (!members || !out || n_members < 1 || n_members > 64) return fail(SLK_E_INVALID, "1..64 members");
  if (exchange != SLK_EXCHANGE_AUTO && exchange != SLK_EXCHANGE_RCCL && exchange != SLK_EXCHANGE_COPY) return fail(SLK_E_INVALID, "exchange mode %d", exchange);
  *out = nullptr;
  bool distinct = true;
  for (int g = 0; g < n_members; g++) {
    slk_index *ix = members[g];
    if (!ix) return fail(SLK_E_INVALID, "null member");
    if (!ix->finalized || !ix->d_parents) return fail(SLK_E_STATE, "member %d is not finalized or has no taxonomy", g);
    if (ix->W > 1) return fail(SLK_E_UNSUPPORTED, "the sharded entry points support minimizers of up to 32 nt (one id column)");
    if (ix->n_shards != (uint32_t)n_members || ix->shard != (uint32_t)g)
      return fail(SLK_E_INVALID, "member %d must be shard %d of %d (slk_index_set_shard); it is shard %u of %u", g, g, n_members, ix->shard, ix->n_shards);
    if (memcmp(&ix->params, &members[0]->params, sizeof(slk_params)) != 0 || ix->T != members[0]->T)
      return fail(SLK_E_INVALID, "member %d differs from member 0 in its splitter or taxonomy", g);
    for (int h = 0; h < g; h++) distinct = distinct && members[h]->device != ix->device;
  }
  int mode = exchange;
  if (mode == SLK_EXCHANGE_AUTO) mode = (distinct && n_members > 1 && rccl().load()) ? SLK_EXCHANGE_RCCL : SLK_EXCHANGE_COPY;
  if (mode == SLK_EXCHANGE_RCCL) {
    if (!distinct) return fail(SLK_E_INVALID, "RCCL needs every member on a device of its own");
    if (!rccl().load()) return fail(SLK_E_UNSUPPORTED, "librccl.so could not be loaded");
  }
  std::unique_ptr<slk_shardset> set(new slk_shardset());
  set->n = n_members;
  set->mode = mode;
  set->m.resize(n_members);
  set->cnt.assign(n_members, std::vector<uint64_t>(n_members, 0));
  set->peer.assign(n_members, std::vector<char>(n_members, 1));
  auto cleanup = [&]() {
    for (Member &mb : set->m) {
      if (mb.ix) (void)hipSetDevice(mb.device);
      if (mb.comm) (void)rccl().CommDestroy(mb.comm);
      if (mb.st) slk_stream_destroy(mb.st);
      mb.release();
    }
  };
  for (int g = 0; g < n_members; g++) {
    Member &mb = set->m[g];
    mb.ix = members[g];
    mb.device = members[g]->device;
    int32_t rc = slk_stream_create(mb.ix, &mb.st);
    if (rc) { cleanup(); return rc; }
    bool ok = hipHostMalloc((void **)&mb.h_counts, ((size_t)n_members + 3) * 8, hipHostMallocDefault) == hipSuccess &&
              hipEventCreateWithFlags(&mb.ev_sent, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&mb.ev_found, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&mb.ev_bounce, hipEventDisableTiming) == hipSuccess &&
              hipStreamCreateWithFlags(&mb.xs, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&mb.us, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&mb.ds, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < NSLOT; i++) {
      Slot &sl = mb.slot[i];
      ok = hipEventCreateWithFlags(&mb.ev_step[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&sl.ev_up, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&sl.ev_keys, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&sl.ev_taxa, hipEventDisableTiming) == hipSuccess &&
           hipHostMalloc((void **)&sl.h_cursors, ((size_t)n_members + 3) * 8, hipHostMallocDefault) == hipSuccess;
    }
    if (!ok) {
      (void)hipGetLastError();
      cleanup();
      return fail(SLK_E_HIP, "could not set up member %d", g);
    }
  }
  // Which members' devices reach each other's memory.  Copies between two that do not are staged through pinned host memory (said
  // once, here, with the reason) instead of failing in the first round.
  for (int a = 0; a < n_members; a++)
    for (int b = 0; b < n_members; b++) {
      const int da = set->m[a].device, db = set->m[b].device;
      if (da == db) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess) { (void)hipGetLastError(); can = 0; }
      if (can) {
        (void)hipSetDevice(da);
        const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
        (void)hipGetLastError();
      }
      set->peer[a][b] = (char)can;
      if (!can && mode == SLK_EXCHANGE_COPY)
        fprintf(stderr, "[slacken_amd] shard set: device %d cannot access the memory of device %d (hipDeviceCanAccessPeer): copies between "
                        "members %d and %d are staged through pinned host memory\n", da, db, a, b);
    }
  if (mode == SLK_EXCHANGE_RCCL) {
    std::vector<int> devs(n_members);
    std::vector<ncclComm_t> comms(n_members, nullptr);
    for (int g = 0; g < n_members; g++) devs[g] = members[g]->device;
    std::lock_guard<std::mutex> lk(rccl_mu());
    const int e = rccl().CommInitAll(comms.data(), n_members, devs.data());
    if (e != 0) { cleanup(); return fail(SLK_E_HIP, "ncclCommInitAll failed: %s", rccl().GetErrorString(e)); }
    for (int g = 0; g < n_members; g++) set->m[g].comm = comms[g];
  }
  *out = set.release();
  return SLK_OK;
}

int32_t slk_shardset_exchange_mode(const slk_shardset *set) { return set ? set->mode : SLK_E_INVALID; }

int32_t slk_shardset_classify(slk_shardset *set, slk_shard_batch *batches, int32_t min_hit_groups, const double *thresholds, int32_t C) {
  int32_t rc = check_batches(set, batches, 1, false, thresholds, C);
  if (rc) return rc;
  return run_rounds(set, batches, 1, false, min_hit_groups, thresholds, C);
}

int32_t slk_shardset_classify_rounds(slk_shardset *set, slk_shard_batch *batches, int32_t n_rounds, int32_t device_resident, int32_t min_hit_groups,
                                     const double *thresholds, int32_t C) {
  int32_t rc = check_batches(set, batches, n_rounds, device_resident != 0, thresholds, C);
  if (rc) return rc;
  return run_rounds(set, batches, n_rounds, device_resident != 0, min_hit_groups, thresholds, C);
}

void slk_shardset_destroy(slk_shardset *set) {
  if (!set) return;
  for (Member &mb : set->m) {
    (void)hipSetDevice(mb.device);   // (the member's own copy: its index may be gone already)
    if (mb.st) (void)hipStreamSynchronize(mb.st->s);
    if (mb.comm) (void)rccl().CommDestroy(mb.comm);
    if (mb.st) slk_stream_destroy(mb.st);
    mb.release();
  }
  delete set;
}

}  // extern "C"

// shardset.hip -- table-sharded classification in ONE process (BASELINE.json configs[3]): a library whose
// record table exceeds one GPU's HBM is spread over several indices -- member g holds the records with fmix64(key) mod n == g
// (slk_index_set_shard) -- and a ROUND classifies one batch of fragments per member:
//   scan     every member scans ITS fragments and sorts their minimizers by owner           (lane_kernel<EMIT>, list compaction)
//   exchange 8-byte keys to their owners                                                    (RCCL send/recv in one group, or copies)
//   lookup   every member answers the keys it received from its shard of the table          (lookup_coop_kernel)
//   exchange 4-byte taxa back, in the order the keys were sent
//   classify every member folds the answers into its fragments' taxon maps and resolves     (lane_kernel<APPLY>)
// This replaces the shuffle behind the reference's join (S/slacken/Classifier.scala:84-95: spans JOIN records ON id, then regrouped
// by title) for the one case where data must move; with a table that fits one GPU the replicated mode needs no exchange at all.
// One host thread drives all members: every stage is launched on every member's stream before the next stage's host-side
// bookkeeping, so the GPUs work side by side; the only host wait inside a round is for the split sizes of the exchange.
// Fragments the lane kernel does not take (over 1000 bases, more than 12 distinct taxa) make a second, staged round of the same
// shape: wave-per-fragment scan into span arrays, keys collected by owner, the same exchange, unbounded classify kernel.
//
// The exchange is RCCL's (ncclSend / ncclRecv between the members' streams, one communicator per member from ncclCommInitAll; the
// library is loaded at run time) when the members sit on distinct devices, and device-to-device copies ordered by events otherwise
// -- several members on ONE device is how this path is tested on a one-GPU box.
#include "hostside.h"

#include <dlfcn.h>

#include <memory>

namespace {

// ---- the few RCCL entry points, resolved at run time (no link-time dependency; PyTorch processes carry their own copy) ----
typedef struct ncclComm *ncclComm_t;
enum { ncclInt32 = 2, ncclInt64 = 4 };   // ncclDataType_t values of the two element types that travel (rccl.h)
struct Rccl {
  void *h = nullptr;
  int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool load() {
    if (h) return true;
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
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
  }
};
Rccl &rccl() { static Rccl r; return r; }

#define NCCLCHK(expr)                                                                                             \
  do {                                                                                                            \
    int e_ = (expr);                                                                                              \
    if (e_ != 0) return fail(SLK_E_HIP, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

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

struct Member {
  slk_index *ix = nullptr;
  slk_stream *st = nullptr;   // this set's stream on the member: HIP stream, staging buffers, read / span / result buffers
  ncclComm_t comm = nullptr;
  // the fast round's lists (engine.h: ShardIO) and the exchange buffers
  DevBuf send_keys, send_meta, counts, batch_base, tile_rows, read_info, defer, out_keys, list_off, owner_counts;
  DevBuf recv_keys, found, taxa, slots, starts;
  uint64_t *h_counts = nullptr;    // pinned: [n + 1] keys per owner, lists that overflowed; staged round: [n] keys per owner
  hipEvent_t ev_sent = nullptr, ev_found = nullptr;
  uint32_t sub = 1;
  uint64_t cap = 0;
  void release() {
    for (DevBuf *b : {&send_keys, &send_meta, &counts, &batch_base, &tile_rows, &read_info, &defer, &out_keys, &list_off, &owner_counts,
                      &recv_keys, &found, &taxa, &slots, &starts})
      b->release();
    if (h_counts) (void)hipHostFree(h_counts);
    if (ev_sent) (void)hipEventDestroy(ev_sent);
    if (ev_found) (void)hipEventDestroy(ev_found);
    h_counts = nullptr; ev_sent = ev_found = nullptr;
  }
};

}  // namespace

struct slk_shardset {
  int n = 0;
  int mode = SLK_EXCHANGE_COPY;
  std::vector<Member> m;
  // cnt[a][b]: elements member a sends to member b in the current exchange
  std::vector<std::vector<uint64_t>> cnt;
};

namespace {

int32_t use(const Member &mb) { return set_device(mb.ix); }

// Moves, for every pair (a, b), n[a][b] elements of esz bytes from src[a] + src_off(a, b) to dst[b] + dst_off(b, a).  The source of a
// was produced on a's stream (ev[a] recorded behind it); what arrives at b is consumed on b's stream.
template <class SrcOff, class DstOff>
int32_t exchange(slk_shardset *set, const std::vector<const void *> &src, const std::vector<void *> &dst, const std::vector<std::vector<uint64_t>> &n,
                 size_t esz, int nccl_type, const std::vector<hipEvent_t> &ev, SrcOff src_off, DstOff dst_off) {
  const int W = set->n;
  if (set->mode == SLK_EXCHANGE_RCCL) {
    NCCLCHK(rccl().GroupStart());
    for (int a = 0; a < W; a++) {
      int32_t rc = use(set->m[a]);
      if (rc) return rc;
      for (int b = 0; b < W; b++) {
        if (n[a][b]) NCCLCHK(rccl().Send((const char *)src[a] + src_off(a, b) * esz, n[a][b], nccl_type, b, set->m[a].comm, set->m[a].st->s));
        if (n[b][a]) NCCLCHK(rccl().Recv((char *)dst[a] + dst_off(a, b) * esz, n[b][a], nccl_type, b, set->m[a].comm, set->m[a].st->s));
      }
    }
    NCCLCHK(rccl().GroupEnd());
    return SLK_OK;
  }
  for (int b = 0; b < W; b++) {
    int32_t rc = use(set->m[b]);
    if (rc) return rc;
    for (int a = 0; a < W; a++) {
      if (!n[a][b]) continue;
      if (a != b) HIPCHK(hipStreamWaitEvent(set->m[b].st->s, ev[a], 0));
      HIPCHK(hipMemcpyAsync((char *)dst[b] + dst_off(b, a) * esz, (const char *)src[a] + src_off(a, b) * esz, n[a][b] * esz, hipMemcpyDefault,
                            set->m[b].st->s));
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

// keys -> owners -> lookup -> taxa back: src_keys[a] holds a's keys sorted by owner (set->cnt[a][b] of them for b); afterwards
// m[a].taxa holds the answers in the same order (on a's stream).
int32_t lookup_round(slk_shardset *set, const std::vector<const void *> &src_keys) {
  const int W = set->n;
  const Layout lay(set->cnt);
  std::vector<void *> recv(W), taxa(W);
  std::vector<const void *> found(W);
  std::vector<hipEvent_t> ev_sent(W), ev_found(W);
  for (int a = 0; a < W; a++) {
    Member &mb = set->m[a];
    int32_t rc = use(mb);
    if (rc) return rc;
    HIPCHK(mb.recv_keys.ensure(std::max<uint64_t>(lay.received[a], 1) * 8));
    HIPCHK(mb.found.ensure(std::max<uint64_t>(lay.received[a], 1) * 4));
    HIPCHK(mb.taxa.ensure(std::max<uint64_t>(lay.sent[a], 1) * 4));
    recv[a] = mb.recv_keys.p; found[a] = mb.found.p; taxa[a] = mb.taxa.p;
    HIPCHK(hipEventRecord(mb.ev_sent, mb.st->s));
    ev_sent[a] = mb.ev_sent; ev_found[a] = mb.ev_found;
  }
  int32_t rc = exchange(set, src_keys, recv, set->cnt, 8, ncclInt64, ev_sent, [&](int a, int b) { return lay.soff[a][b]; },
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
  return exchange(set, found, taxa, back, 4, ncclInt32, ev_found, [&](int b, int a) { return lay.roff[b][a]; },
                  [&](int a, int b) { return lay.soff[a][b]; });
}

int32_t sync_all(slk_shardset *set) {
  for (Member &mb : set->m) {
    int32_t rc = use(mb);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(mb.st->s));
  }
  return SLK_OK;
}

int32_t ensure_outputs(Member &mb, uint64_t R, int32_t C) {
  slk_stream *st = mb.st;
  HIPCHK(st->out_taxon.ensure((size_t)C * R * 4));
  HIPCHK(st->out_cls.ensure((size_t)C * R));
  HIPCHK(st->out_nd.ensure(R * 4));
  HIPCHK(st->out_tk.ensure(R * 4));
  HIPCHK(st->out_nh.ensure(R * 4));
  return SLK_OK;
}

// results of member g's batch to the caller's arrays (everything queued on its stream has been synchronised)
int32_t download(Member &mb, const slk_shard_batch &B, int32_t C) {
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

// ---- the staged round: everything the lane kernel does not take (and splitters outside its range) -----------------------------
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
    if (!rc) rc = ensure_outputs(mb, B.R, A.C);
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
  rc = lookup_round(set, src);
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
    if (!rc) rc = download(set->m[g], A.batches[g], A.C);
    if (rc) return rc;
  }
  return SLK_OK;
}

// ---- the fast round -----------------------------------------------------------------------------------------------------------
int32_t emit_member(Member &mb, const slk_shard_batch &B, int W, uint64_t total, uint64_t mate_total, uint64_t scale, bool want_hits) {
  slk_stream *st = mb.st;
  const bool paired = B.mate_offsets != nullptr;
  // about 0.26 probes per base on random sequence, spread evenly by the hash over W owners and by the tile index over `sub`
  // sub-lists per owner (each fed by at least 64 tiles, so that the spread holds): 0.6 / (W * sub) per base leaves 2x headroom;
  // a list that overflows all the same is reported by the compaction, and the batch is emitted again with more room
  const uint64_t tiles = (B.R + 63) / 64;
  uint32_t sub = 1;
  while (sub < 256 && (uint64_t)sub * 2 * 64 <= tiles) sub *= 2;
  const uint64_t cap = ((uint64_t)((double)(total + mate_total) * 0.6 / ((double)W * sub)) + 4096) * scale;
  if (cap >= (1ull << 25)) return fail(SLK_E_CAPACITY, "a batch of %llu bases is too large for the sharded lists: use smaller batches", (unsigned long long)(total + mate_total));
  mb.sub = sub; mb.cap = cap;
  const uint64_t rows = slk_shard_batch_rows(total, mate_total, B.R, paired);
  const uint64_t lists = (uint64_t)W * sub;
  HIPCHK(mb.send_keys.ensure(lists * cap * 8));
  HIPCHK(mb.send_meta.ensure(lists * cap * 4));
  HIPCHK(mb.out_keys.ensure(lists * cap * 8));
  HIPCHK(mb.counts.ensure(lists * 8));
  HIPCHK(mb.list_off.ensure((lists + 1) * 8));
  HIPCHK(mb.owner_counts.ensure(((size_t)W + 1) * 8));
  HIPCHK(mb.batch_base.ensure(rows * W * 4));
  HIPCHK(mb.tile_rows.ensure((tiles + 1) * 4));
  HIPCHK(mb.read_info.ensure(B.R * 8));
  HIPCHK(mb.defer.ensure(B.R * 4));
  HIPCHK(hipMemsetAsync(mb.counts.p, 0, lists * 8, st->s));
  HIPCHK(hipMemsetAsync(mb.defer.p, 0, B.R * 4, st->s));
  FusedArgs F{};
  F.P = mb.ix->sp; F.bases = st->bases.as<uint8_t>(); F.offsets = st->offsets.as<uint64_t>();
  F.mate_bases = paired ? st->mate_bases.as<uint8_t>() : nullptr; F.mate_offsets = paired ? st->mate_offsets.as<uint64_t>() : nullptr;
  F.R = B.R; F.status = st->d_status;
  if (want_hits) { F.span_meta = st->span_meta.as<int32_t>(); F.span_taxon = st->span_taxon.as<int32_t>(); F.span_count = st->span_count.as<int32_t>(); }
  ShardIO S{};
  S.n_shards = W; S.n_sub = (int32_t)sub; S.cap = cap; S.send_keys = mb.send_keys.as<int64_t>();
  S.send_counts = mb.counts.as<unsigned long long>(); S.batch_base = mb.batch_base.as<uint32_t>(); S.send_meta = mb.send_meta.as<uint32_t>();
  S.tile_rows = mb.tile_rows.as<uint32_t>(); S.read_info = (int2 *)mb.read_info.p;
  launch_lane_sharded(LANE_EMIT, F, S, mb.defer.as<int32_t>(), 1000, st->s);
  HIPCHK(hipGetLastError());
  launch_compact_lists(mb.send_keys.as<int64_t>(), mb.counts.as<unsigned long long>(), (uint32_t)W, sub, cap, mb.out_keys.as<int64_t>(),
                       mb.list_off.as<uint64_t>(), mb.owner_counts.as<uint64_t>(), st->s);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(mb.h_counts, mb.owner_counts.p, ((size_t)W + 1) * 8, hipMemcpyDeviceToHost, st->s));
  return SLK_OK;
}

int32_t fast_round(slk_shardset *set, const RoundArgs &A, std::vector<std::vector<uint32_t>> &deferred) {
  const int W = set->n;
  std::vector<uint64_t> total(W, 0), mate_total(W, 0);
  std::vector<bool> hits(W, false);
  Thresholds thr{};
  memcpy(thr.v, A.thresholds, A.C * sizeof(double));
  for (int g = 0; g < W; g++) {
    Member &mb = set->m[g];
    const slk_shard_batch &B = A.batches[g];
    for (int d = 0; d <= W; d++) mb.h_counts[d] = 0;
    if (B.R == 0) continue;
    int32_t rc = use(mb);
    if (rc) return rc;
    hits[g] = B.out_hit_offsets != nullptr && B.out_hits != nullptr;
    rc = upload_reads(mb.st, B.bases, B.offsets, B.mate_bases, B.mate_offsets, B.R, &total[g], &mate_total[g]);
    if (!rc) rc = ensure_outputs(mb, B.R, A.C);
    if (!rc && hits[g]) rc = ensure_scratch(mb.st, span_slots(total[g], mate_total[g], B.R, B.mate_offsets != nullptr), B.R);
    if (!rc) rc = emit_member(mb, B, W, total[g], mate_total[g], 1, hits[g]);
    if (rc) return rc;
  }
  // the round's one host wait: the split sizes (and whether a send list overflowed: that member emits again with longer lists)
  for (int g = 0; g < W; g++) {
    Member &mb = set->m[g];
    const slk_shard_batch &B = A.batches[g];
    if (B.R == 0) continue;
    int32_t rc = use(mb);
    if (rc) return rc;
    for (uint64_t scale = 2;; scale *= 2) {
      HIPCHK(hipStreamSynchronize(mb.st->s));
      if (mb.h_counts[W] == 0) break;
      if (scale > 64) return fail(SLK_E_CAPACITY, "the send lists of a batch overflowed at 64 times their estimated size");
      mb.st->queued.clear();
      *mb.st->h_status = 0;
      HIPCHK(hipMemsetAsync(mb.st->d_status, 0, sizeof(int32_t), mb.st->s));
      rc = emit_member(mb, B, W, total[g], mate_total[g], scale, hits[g]);
      if (rc) return rc;
    }
  }
  std::vector<const void *> src(W, nullptr);
  for (int g = 0; g < W; g++) {
    for (int d = 0; d < W; d++) set->cnt[g][d] = set->m[g].h_counts[d];
    src[g] = set->m[g].out_keys.p;
  }
  int32_t rc = lookup_round(set, src);
  if (rc) return rc;
  for (int g = 0; g < W; g++) {
    Member &mb = set->m[g];
    const slk_shard_batch &B = A.batches[g];
    if (B.R == 0) continue;
    rc = use(mb);
    if (rc) return rc;
    slk_stream *st = mb.st;
    const bool paired = B.mate_offsets != nullptr;
    FusedArgs F{};
    F.P = mb.ix->sp; F.T = mb.ix->view(); F.parents = mb.ix->kernel_parents(); F.ntax = mb.ix->kernel_ntax(); F.nodes = mb.ix->kernel_nodes();
    F.bases = st->bases.as<uint8_t>(); F.offsets = st->offsets.as<uint64_t>();
    F.mate_bases = paired ? st->mate_bases.as<uint8_t>() : nullptr; F.mate_offsets = paired ? st->mate_offsets.as<uint64_t>() : nullptr;
    F.R = B.R; F.out_stride = B.R; F.min_hit_groups = A.min_hit_groups; F.thr = thr; F.C = A.C;
    F.out_taxon = st->out_taxon.as<int32_t>(); F.out_classified = st->out_cls.as<uint8_t>(); F.out_nd = st->out_nd.as<int32_t>();
    F.out_tk = st->out_tk.as<int32_t>(); F.out_nh = st->out_nh.as<int32_t>(); F.status = st->d_status;
    if (hits[g]) { F.span_meta = st->span_meta.as<int32_t>(); F.span_taxon = st->span_taxon.as<int32_t>(); F.span_count = st->span_count.as<int32_t>(); }
    ShardIO S{};
    S.n_shards = W; S.n_sub = (int32_t)mb.sub; S.cap = mb.cap; S.batch_base = mb.batch_base.as<uint32_t>(); S.list_off = mb.list_off.as<uint64_t>();
    S.taxa = mb.taxa.as<int32_t>(); S.send_meta = mb.send_meta.as<uint32_t>(); S.tile_rows = mb.tile_rows.as<uint32_t>();
    S.read_info = (int2 *)mb.read_info.p; S.to_dense = mb.ix->d_to_dense; S.n_to_dense = mb.ix->T;
    launch_lane_sharded(LANE_APPLY, F, S, mb.defer.as<int32_t>(), 1000, st->s);
    HIPCHK(hipGetLastError());
  }
  rc = sync_all(set);
  if (rc) return rc;
  for (int g = 0; g < W; g++) {
    Member &mb = set->m[g];
    const slk_shard_batch &B = A.batches[g];
    deferred[g].clear();
    if (B.R == 0) continue;
    rc = use(mb);
    if (rc) return rc;
    mb.st->queued.clear();   // (these launches are not re-runnable by check_status: deferrals are settled below)
    *mb.st->h_status = 0;
    HIPCHK(hipMemsetAsync(mb.st->d_status, 0, sizeof(int32_t), mb.st->s));
    std::vector<int32_t> defer(B.R);
    rc = copy_out(mb.st, defer.data(), mb.defer.p, B.R * 4);
    if (!rc) rc = download(mb, B, A.C);
    if (rc) return rc;
    for (uint64_t r = 0; r < B.R; r++)
      if (defer[r]) deferred[g].push_back((uint32_t)r);
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
  auto cleanup = [&]() {
    for (Member &mb : set->m) {
      if (mb.ix) (void)hipSetDevice(mb.ix->device);
      if (mb.comm) (void)rccl().CommDestroy(mb.comm);
      if (mb.st) slk_stream_destroy(mb.st);
      mb.release();
    }
  };
  for (int g = 0; g < n_members; g++) {
    Member &mb = set->m[g];
    mb.ix = members[g];
    int32_t rc = slk_stream_create(mb.ix, &mb.st);
    if (rc) { cleanup(); return rc; }
    if (hipHostMalloc((void **)&mb.h_counts, ((size_t)n_members + 1) * 8, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&mb.ev_sent, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&mb.ev_found, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      cleanup();
      return fail(SLK_E_HIP, "could not set up member %d", g);
    }
  }
  if (mode == SLK_EXCHANGE_RCCL) {
    std::vector<int> devs(n_members);
    std::vector<ncclComm_t> comms(n_members, nullptr);
    for (int g = 0; g < n_members; g++) devs[g] = members[g]->device;
    const int e = rccl().CommInitAll(comms.data(), n_members, devs.data());
    if (e != 0) { cleanup(); return fail(SLK_E_HIP, "ncclCommInitAll failed: %s", rccl().GetErrorString(e)); }
    for (int g = 0; g < n_members; g++) set->m[g].comm = comms[g];
  }
  *out = set.release();
  return SLK_OK;
}

int32_t slk_shardset_exchange_mode(const slk_shardset *set) { return set ? set->mode : SLK_E_INVALID; }

int32_t slk_shardset_classify(slk_shardset *set, slk_shard_batch *batches, int32_t min_hit_groups, const double *thresholds, int32_t C) {
  if (!set || !batches) return fail(SLK_E_INVALID, "null argument");
  if (C < 1 || C > MAX_THRESHOLDS || !thresholds) return fail(SLK_E_INVALID, "need 1..%d thresholds", MAX_THRESHOLDS);
  const int W = set->n;
  bool fast = true;
  for (int g = 0; g < W; g++) {
    const slk_shard_batch &B = batches[g];
    if (B.R && (!B.bases || !B.offsets || !B.out_taxon || !B.out_classified)) return fail(SLK_E_INVALID, "null argument (member %d)", g);
    if ((B.mate_bases == nullptr) != (B.mate_offsets == nullptr)) return fail(SLK_E_INVALID, "mate_bases and mate_offsets must be given together");
    if (B.R >= 0xFFFFFFFFull) return fail(SLK_E_INVALID, "a batch holds fewer than 2^32 fragments");
    if (B.out_hit_offsets) B.out_hit_offsets[0] = 0;
    fast = fast && lane_path_ok(set->m[g].ix);
  }
  RoundArgs A{batches, min_hit_groups, thresholds, C};
  if (!fast) return staged_round(set, A);   // (a splitter outside the lane kernel's range: everything takes the staged kernels)
  std::vector<std::vector<uint32_t>> deferred(W);
  int32_t rc = fast_round(set, A, deferred);
  if (rc) return rc;
  bool any = false;
  for (int g = 0; g < W; g++) any = any || !deferred[g].empty();
  if (!any) return SLK_OK;
  // The fragments the lane kernel handed back, as a batch of their own per member, through the staged round; their rows replace
  // what the fast round left in the caller's arrays (hit lists are rebuilt with the longer lists spliced in).
  struct Sub {
    std::vector<uint8_t> bases, mates, cls;
    std::vector<uint64_t> offs, moffs, hit_offs;
    std::vector<int32_t> taxon, nd, tk;
    std::vector<slk_hit> hits;
  };
  std::vector<Sub> sub(W);
  std::vector<slk_shard_batch> sb(W);
  for (int g = 0; g < W; g++) {
    const slk_shard_batch &B = batches[g];
    Sub &s = sub[g];
    const size_t n = deferred[g].size();
    sb[g] = slk_shard_batch{};
    if (!n) continue;
    const bool paired = B.mate_offsets != nullptr;
    s.offs.assign(1, 0);
    if (paired) s.moffs.assign(1, 0);
    for (uint32_t r : deferred[g]) {
      s.bases.insert(s.bases.end(), B.bases + B.offsets[r], B.bases + B.offsets[r + 1]);
      s.offs.push_back(s.bases.size());
      if (paired) {
        s.mates.insert(s.mates.end(), B.mate_bases + B.mate_offsets[r], B.mate_bases + B.mate_offsets[r + 1]);
        s.moffs.push_back(s.mates.size());
      }
    }
    if (s.bases.empty()) s.bases.push_back('N');
    if (paired && s.mates.empty()) s.mates.push_back('N');
    s.taxon.resize((size_t)C * n); s.cls.resize((size_t)C * n); s.nd.resize(n); s.tk.resize(n); s.hit_offs.resize(n + 1);
    const bool want_hits = B.out_hit_offsets && B.out_hits;
    const size_t cap = s.bases.size() + s.mates.size() + n + 1;
    if (want_hits) s.hits.resize(cap);
    sb[g] = slk_shard_batch{s.bases.data(), s.offs.data(), paired ? s.mates.data() : nullptr, paired ? s.moffs.data() : nullptr, n,
                            s.taxon.data(), s.cls.data(), s.nd.data(), s.tk.data(), B.out_hit_offsets ? s.hit_offs.data() : nullptr,
                            want_hits ? s.hits.data() : nullptr, cap};
  }
  RoundArgs A2{sb.data(), min_hit_groups, thresholds, C};
  rc = staged_round(set, A2);
  if (rc) return rc;
  for (int g = 0; g < W; g++) {
    slk_shard_batch &B = batches[g];
    const Sub &s = sub[g];
    const size_t n = deferred[g].size();
    if (!n) continue;
    for (size_t i = 0; i < n; i++) {
      const uint32_t r = deferred[g][i];
      for (int32_t c = 0; c < C; c++) {
        B.out_taxon[(size_t)c * B.R + r] = s.taxon[(size_t)c * n + i];
        B.out_classified[(size_t)c * B.R + r] = s.cls[(size_t)c * n + i];
      }
      if (B.out_num_distinct) B.out_num_distinct[r] = s.nd[i];
      if (B.out_total_kmers) B.out_total_kmers[r] = s.tk[i];
    }
    if (!B.out_hit_offsets) continue;
    // hit lists: the fast round left the deferred fragments without spans; their lists from the staged round are spliced in
    std::vector<uint64_t> offs(B.R + 1, 0);
    {
      size_t i = 0;
      for (uint64_t r = 0; r < B.R; r++) {
        uint64_t len = B.out_hit_offsets[r + 1] - B.out_hit_offsets[r];
        if (i < n && deferred[g][i] == r) { len = s.hit_offs[i + 1] - s.hit_offs[i]; i++; }
        offs[r + 1] = offs[r] + len;
      }
    }
    if (B.out_hits) {
      if (offs[B.R] > B.hits_capacity)
        return fail(SLK_E_CAPACITY, "output needs %llu entries, capacity is %llu", (unsigned long long)offs[B.R], (unsigned long long)B.hits_capacity);
      std::vector<slk_hit> merged(offs[B.R]);
      size_t i = 0;
      for (uint64_t r = 0; r < B.R; r++) {
        const uint64_t len = offs[r + 1] - offs[r];
        if (i < n && deferred[g][i] == r) { memcpy(merged.data() + offs[r], s.hits.data() + s.hit_offs[i], len * sizeof(slk_hit)); i++; }
        else if (len) memcpy(merged.data() + offs[r], B.out_hits + B.out_hit_offsets[r], len * sizeof(slk_hit));
      }
      if (!merged.empty()) memcpy(B.out_hits, merged.data(), merged.size() * sizeof(slk_hit));
    }
    memcpy(B.out_hit_offsets, offs.data(), (B.R + 1) * sizeof(uint64_t));
  }
  return SLK_OK;
}

void slk_shardset_destroy(slk_shardset *set) {
  if (!set) return;
  for (Member &mb : set->m) {
    (void)hipSetDevice(mb.ix->device);
    if (mb.st) (void)hipStreamSynchronize(mb.st->s);
    if (mb.comm) (void)rccl().CommDestroy(mb.comm);
    if (mb.st) slk_stream_destroy(mb.st);
    mb.release();
  }
  delete set;
}

}  // extern "C"

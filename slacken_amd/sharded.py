"""Table-sharded classify (BASELINE.json configs[3]): the record table is too large for one GPU's HBM, so
rank g holds the records with fmix64(key) mod G == g.  Per batch every rank scans ITS reads, sends each minimizer to
its owner (all-to-all of 8-byte keys over RCCL/xGMI), owners look them up in their shard, the taxa return by the inverse
all-to-all (4 bytes each), and every rank finishes its own reads' LCA locally.  This replaces the reference's shuffle +
join (S/slacken/Classifier.scala:84) for the one case where data must move; with a table that fits, use the replicated
mode (no collective at all).

Host plumbing only: torch for device buffers and the collectives, the engine for all compute.  Two routes:
  fast    slk_shard_step_device: per pipeline step ONE kernel scans batch t into per-owner send regions (sent as they stand: no
          compaction), answers the keys received for batch t - 2 and replays + classifies batch t - 4, while the all-to-alls of
          batches t - 1 (keys) and t - 3 (taxa) run on a second stream; nothing but 8-byte keys and 4-byte taxa moves and the
          fragments are scanned once; takes fragments of up to 1000 bases with at most 12 distinct taxa;
  staged  slk_scan_device / slk_lookup_device / slk_classify_hits_device with the exchange lists built by torch ops: takes
          everything, and the fragments the fast route hands back (`defer`)."""
import sys

import numpy as np

_C1 = 0xff51afd7ed558ccd - (1 << 64)
_C2 = 0xc4ceb9fe1a85ec53 - (1 << 64)


def fmix64_torch(x):
    """The engine's bijective 64-bit mixer (engine.h fmix64) on an int64 tensor (two's-complement wrap-around)."""
    m31 = (1 << 31) - 1
    x = x ^ ((x >> 33) & m31)
    x = x * _C1
    x = x ^ ((x >> 33) & m31)
    x = x * _C2
    x = x ^ ((x >> 33) & m31)
    return x


def shard_of_torch(keys, world):
    """fmix64(key) mod world as unsigned arithmetic, int64 tensor in -> int64 tensor out (== slk_shard_of)."""
    h = fmix64_torch(keys)
    if world & (world - 1) == 0:
        return h & (world - 1)
    half = (h >> 1) & ((1 << 63) - 1)          # floor(h_unsigned / 2)
    return ((half % world) * 2 + (h & 1)) % world


def shard_of_numpy(keys, world):
    k = np.asarray(keys).astype(np.int64).view(np.uint64)
    x = k.copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xff51afd7ed558ccd)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xc4ceb9fe1a85ec53)
        x ^= x >> np.uint64(33)
    return (x % np.uint64(world)).astype(np.int64)


class _StageSpan:
    """with-block that brackets a stage with two timing events on its stream and files them in the batch's profile list (nothing
    if that is None).  A plain module-level class without a per-call type: nothing here may end up in a reference cycle that only the
    cycle collector frees (see ShardedClassifier.close)."""
    __slots__ = ("torch", "prof", "name", "ext", "e0")

    def __init__(self, torch, prof, name, ext):
        self.torch, self.prof, self.name, self.ext, self.e0 = torch, prof, name, ext, None

    def __enter__(self):
        if self.prof is not None:
            self.e0 = self.torch.cuda.Event(enable_timing=True)
            self.e0.record(self.ext)
        return self

    def __exit__(self, exc_type, exc, tb):
        if self.prof is not None and exc_type is None:
            e1 = self.torch.cuda.Event(enable_timing=True)
            e1.record(self.ext)
            self.prof.append((self.name, self.e0, e1))
        return False


class Exchange:
    """The two all-to-all(v) steps of the table-sharded mode and nothing else: no engine, no GPU needed (bench.py --dry-run and the
    CPU tests run it over gloo).  Keys go to their owners sorted by destination rank, answers come back in the same order.  The
    split sizes travel first, and with them one flag word per rank (a send region overflowed) -- so the decision every
    rank must take alike costs no collective of its own and nothing touches torch's default stream."""

    def __init__(self, rank=0, world=1, dist=None, device=None, on_cpu=False, force=False):
        import torch
        self.torch, self.rank, self.world, self.dist = torch, rank, world, dist
        self.force = force   # a world of ONE rank goes through the collectives all the same (it sends to itself): runs that code on one GPU
        self.device = device if device is not None else torch.device("cpu")
        self.on_cpu = on_cpu or self.device.type == "cpu"  # gloo has no device all-to-all: stage through host memory (tests)

    @property
    def single(self):
        return self.dist is None or (self.world == 1 and not self.force)

    def split_sizes(self, send_counts, flag=False):
        """-> (recv_counts, any rank's flag).  One all-to-all of [count for that peer, my flag] pairs."""
        if self.single:
            return list(send_counts), bool(flag)
        torch = self.torch
        dev = "cpu" if self.on_cpu else self.device
        sc = torch.tensor([[int(c), 1 if flag else 0] for c in send_counts], dtype=torch.int64, device=dev)
        rc = torch.empty_like(sc)
        self.dist.all_to_all_single(rc, sc)
        rc = rc.tolist()
        return [int(r[0]) for r in rc], any(int(r[1]) for r in rc) or bool(flag)

    def all_to_all(self, send, send_counts, recv_counts):
        """send: tensor sorted by destination rank -> what the other ranks sent here, sorted by source rank"""
        if self.single:
            return send
        torch = self.torch
        dev = "cpu" if self.on_cpu else self.device
        src = send.to(dev)
        out = torch.empty(sum(recv_counts), dtype=send.dtype, device=dev)
        self.dist.all_to_all_single(out, src, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts))
        return out.to(self.device)

    def regions_out(self, regions, recv_counts):
        """regions[d]: the (contiguous) tensor this rank sends to rank d -- views of per-owner send regions, sent as they stand, no
        packing pass -> what the ranks sent here, one contiguous tensor sorted by source rank."""
        torch = self.torch
        if self.single:
            return regions[0]
        if self.on_cpu:
            return self.all_to_all(torch.cat([r.reshape(-1) for r in regions]), [int(r.numel()) for r in regions], recv_counts)
        out = torch.empty(max(sum(recv_counts), 1), dtype=regions[0].dtype, device=self.device)
        outs = list(out[:sum(recv_counts)].split(list(recv_counts)))
        self.dist.all_to_all(outs, [r.contiguous() for r in regions])
        return out[:sum(recv_counts)]

    def regions_back(self, send, send_counts, regions):
        """the way back: send (sorted by destination = the order the keys arrived in) is written INTO regions[d] (views of the
        per-owner answer regions: the answers land at the positions their keys had)."""
        if self.single:
            if regions[0].data_ptr() != send.data_ptr():
                regions[0].copy_(send[:regions[0].numel()])
            return
        if self.on_cpu:
            back = self.all_to_all(send, send_counts, [int(r.numel()) for r in regions])
            o = 0
            for r in regions:
                r.copy_(back[o:o + r.numel()])
                o += r.numel()
            return
        self.dist.all_to_all(list(regions), list(send[:sum(send_counts)].split(list(send_counts))))

    def any_rank(self, flag):
        """logical OR of a host flag over all ranks (every rank must take the same sequence of collectives)"""
        if self.single:
            return bool(flag)
        t = self.torch.tensor([1 if flag else 0], dtype=self.torch.int64, device="cpu" if self.on_cpu else self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return bool(int(t.item()))


class ShardedClassifier:
    """index: this rank's slacken_amd.Index holding ONLY the records with shard_of(key) == rank (plus the taxonomy).
    Call close() when done (or use it as a context manager): the batches' device tensors live on the ENGINE's streams, which torch
    sees as external streams, and must be gone -- with torch's cache of their blocks -- before those streams are destroyed."""

    def __init__(self, index, rank=0, world=1, dist=None, device=None, exchange_on_cpu=False, force_collectives=False):
        import torch
        self.torch, self.ix, self.st = torch, index, index.stream()
        self.rank, self.world, self.dist = rank, world, dist
        self.device = device if device is not None else torch.device("cuda", 0)
        self.on_cpu = exchange_on_cpu
        self.ex = Exchange(rank, world, dist, self.device, exchange_on_cpu, force_collectives)
        self.stage_ms = None   # classify_many(profile=True): per-stage device times of the batches
        self._xst = None
        self._ext = self._xext = None
        self._closed = False

    # ---- lifetime -----------------------------------------------------------------------------------------------------------------
    def close(self):
        """Ordered teardown: nothing of a batch is held any more, both streams are drained, torch's caching allocator gives back the
        blocks it keeps for them (a cached block belongs to the stream it was allocated on), the torch views of the streams go, and
        only then are the engine's streams destroyed.  Results handed out earlier stay valid: they were allocated on the CALLER's
        stream, not on the engine's."""
        if self._closed:
            return
        self._closed = True
        torch = self.torch
        for st in (self.st, self._xst):
            if st is not None and getattr(st, "h", None):
                try:
                    st.synchronize()
                except Exception:
                    pass
        if torch.cuda.is_available():
            torch.cuda.synchronize(self.device)
            torch.cuda.empty_cache()
            # (the pinned-memory allocator too keeps events on the streams its blocks were used on; this classifier's own pinned
            #  buffers are the library's, but a caller may have copied results with non_blocking=True on one of these streams)
            host_empty = getattr(torch._C, "_host_emptyCache", None)
            if host_empty is not None:
                try:
                    host_empty()
                except Exception:
                    pass
        self.__dict__.pop("_cursor_pool", None)
        self._ext = self._xext = None     # (the ExternalStream wrappers: Stream.close refuses while one is alive)
        for st in (self._xst, self.st):
            if st is not None:
                st.close()
        self._xst = self.st = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _all_to_all(self, send, send_counts):
        """send: tensor sorted by destination rank; returns (received tensor, recv_counts list)."""
        recv_counts, _ = self.ex.split_sizes(send_counts)
        return self.ex.all_to_all(send, send_counts, recv_counts), recv_counts

    def classify(self, d_bases, d_offsets, R, total_bases, thresholds=(0.0,), min_hit_groups=2, fast=True, d_mate_bases=None,
                 d_mate_offsets=None, total_mate_bases=0):
        """d_bases uint8 [total_bases], d_offsets int64 [R+1] (device tensors); second mates likewise (optional)."""
        mates = (d_mate_bases, d_mate_offsets, total_mate_bases) if d_mate_bases is not None else None
        if fast:
            outs = self.classify_many([(d_bases, d_offsets, R, total_bases, mates)], thresholds, min_hit_groups)
            if outs is not None:
                return outs[0]
        return self._classify_staged(d_bases, d_offsets, R, total_bases, thresholds, min_hit_groups, mates)

    def _any_rank(self, flag):
        """logical OR of a host flag over all ranks, on the exchange stream (never torch's default stream, see _streams)"""
        if self.ex.single:
            return bool(flag)
        _, xext = self._streams()
        with self.torch.cuda.stream(xext):
            return self.ex.any_rank(flag)

    # ---- the fast route -------------------------------------------------------------------------------------------------------
    # A batch passes through three jobs -- EMIT (scan, keys into per-owner send regions), LOOKUP (on the owners), APPLY (replay of
    # the probe log with the owners' answers, classification) -- with an exchange between them, and step t of classify_many launches
    # ONE kernel on the compute stream that carries EMIT(t), LOOKUP(t - 2) and APPLY(t - 4) (slk_shard_step_device): the lookups'
    # and the replay's memory latency hides behind the scan the way the local kernel's probes do.  The exchanges of batches t - 1
    # (keys) and t - 3 (taxa) run beside it on a second stream.  The host's one wait per step is for the cursors of the batch
    # emitted a step earlier (the exchange's split sizes); by then the next step is queued.
    # Both streams are the ENGINE's, wrapped as torch ExternalStreams so that torch kernels, collectives and engine kernels are
    # ordered by the streams themselves.  (Nothing here touches torch's default stream: the engine's streams are blocking streams,
    # and an event recorded on the legacy null stream waits for all of them and holds back whatever is issued after it.)
    def _streams(self):
        if self._ext is None:
            torch = self.torch
            self._ext = self.st.external_stream(torch, self.device)
            self._xst = self.ix.stream()
            self._xext = self._xst.external_stream(torch, self.device)
        return self._ext, self._xext

    def _ev(self, b, name, ext):
        """(profile) an event pair around a stage of batch b on stream ext: a context manager"""
        return _StageSpan(self.torch, b["prof"] if b is not None else None, name, ext)

    def _region_capacity(self, total_bases, R):
        """entries of an owner's send region: the expected super-mers of random sequence (2 / (w + 1) per k-mer window) spread evenly
        by the hash, a fifth on top, and a chunk per wave that may hold one half filled when the kernel ends"""
        from slacken_amd import capi
        W = self.world
        chunk = int(capi.lib().slk_shard_chunk(W))
        w = self.ix.k - self.ix.m + 1
        windows = max(0, total_bases - R * (self.ix.k - 1))
        expect = 2.0 / (w + 1) * windows + R
        cap = int(expect / W * 1.2) + 4096 + chunk * min((R + 63) // 64, 8192)
        cap = -(-cap // chunk) * chunk
        return min(cap, ((1 << 32) - 1) // chunk * chunk - chunk)

    def _emit_state(self, batch, profile):
        """device state of a batch for its three jobs (allocated on the compute stream)"""
        from slacken_amd import capi
        torch, dev, W = self.torch, self.device, self.world
        ext, _ = self._streams()
        d_bases, d_offsets, R, total_bases, mates = batch
        mb, mo, mtotal = mates if mates is not None else (None, None, 0)
        cap = self._region_capacity(total_bases + mtotal, R)
        rows = int(capi.lib().slk_shard_batch_rows(total_bases, mtotal, R, 1 if mates is not None else 0))
        tiles = (R + 63) // 64
        with torch.cuda.stream(ext):
            t = dict(send_keys=torch.empty(W * cap, dtype=torch.int64, device=dev),
                     send_meta=torch.empty(W * cap, dtype=torch.int32, device=dev),
                     cursors=torch.empty(W + 3, dtype=torch.int64, device=dev),
                     log=torch.empty(max(rows, 1) * W * 4, dtype=torch.int32, device=dev),
                     tile_rows=torch.empty(2 * tiles + 2, dtype=torch.int32, device=dev),
                     read_info=torch.empty(2 * max(R, 1), dtype=torch.int32, device=dev),
                     defer=torch.zeros(max(R, 1), dtype=torch.int32, device=dev))
        L = capi.ShardLists(d_bases.data_ptr(), d_offsets.data_ptr(), mb.data_ptr() if mb is not None else None,
                            mo.data_ptr() if mo is not None else None, R, total_bases, mtotal, W, 0, cap, t["send_keys"].data_ptr(), t["send_meta"].data_ptr(),
                            t["cursors"].data_ptr(), t["log"].data_ptr(), t["tile_rows"].data_ptr(), t["read_info"].data_ptr(),
                            t["defer"].data_ptr(), None, None, None)
        # (pinned by the LIBRARY, not by torch: torch's pinned-memory allocator keeps an event per block on the stream that used it
        #  and asks for it at some later allocation -- by then the engine's stream that the event was recorded on may be gone:
        #  "operation not permitted on an event last recorded in a capturing stream" in the first pinned allocation of a later
        #  classifier, once in 3 000 tests.  Memory torch does not own has no such afterlife.)
        h = torch.from_numpy(self._cursor_buffer(W + 3))
        return dict(R=R, cap=cap, lists=L, h_cursors=h, prof=[] if profile else None, failed=False, batch=batch, **t)

    def _cursor_buffer(self, n):
        """a pinned int64 array of n words from this classifier's small pool (slk_host_alloc; a pipeline has at most six batches in flight)"""
        from slacken_amd import capi
        pool = self.__dict__.setdefault("_cursor_pool", [])
        at = self.__dict__.get("_cursor_next", 0)
        if len(pool) < 8:
            pool.append(capi.pinned_array((n,), np.int64))
            self._cursor_next = 0
            return pool[-1]
        self._cursor_next = (at + 1) % len(pool)
        if pool[at].size < n:
            pool[at] = capi.pinned_array((n,), np.int64)
        return pool[at][:n]

    def classify_many(self, batches, thresholds=(0.0,), min_hit_groups=2, profile=False):
        """The fast route over several batches [(d_bases, d_offsets, R, total_bases, mates or None)], pipelined (see above).
        Every rank must pass the same number of batches.  Returns the list of result dicts, or None if the splitter only has the
        staged route.  profile: self.stage_ms = {stage: mean device ms per batch} from events around every stage (the step kernel is
        ONE stage: its three jobs belong to three batches)."""
        import ctypes as C
        import slacken_amd
        from slacken_amd import capi
        torch, dev, W = self.torch, self.device, self.world
        ext, xext = self._streams()
        n = len(batches)
        Cn = len(thresholds)
        thr = (C.c_double * Cn)(*thresholds)
        cur = torch.cuda.current_stream()
        # results: on the CALLER's stream (they outlive this call, and must not belong to the engine's streams, see close())
        outs = []
        for (_, _, R, _, _) in batches:
            outs.append(dict(taxon=torch.zeros(Cn * max(R, 1), dtype=torch.int32, device=dev),
                             classified=torch.zeros(Cn * max(R, 1), dtype=torch.uint8, device=dev),
                             num_distinct=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                             total_kmers=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                             num_hits=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                             exchanged_keys=0, looked_up_keys=0, sent_remote_keys=0))
        ext.wait_stream(cur)      # (the caller's tensors and the results' memory were produced / last used on ITS stream)
        states = [None] * n
        steps = []                # (profile) one event pair per step kernel
        unsupported = False

        def launch(t):
            nonlocal unsupported
            emit = lk = al = ap = None
            b = None
            if t < n:
                b = states[t] = self._emit_state(batches[t], profile)
                if batches[t][2]:
                    emit = b["lists"]
            if 0 <= t - 2 < n:
                s2 = states[t - 2]
                ext.wait_event(s2["keys_here"])
                if s2["looked_up"] and not s2["failed"]:
                    lk = capi.ShardLookup(s2["recv_keys"].data_ptr(), s2["looked_up"], s2["found"].data_ptr())
            if 0 <= t - 4 < n:
                s4 = states[t - 4]
                ext.wait_event(s4["taxa_here"])
                if s4["R"] and not s4["failed"]:
                    o = outs[t - 4]
                    al = s4["lists"]
                    ap = capi.ShardResults(s4["taxa"].data_ptr(), min_hit_groups, Cn, thr, o["taxon"].data_ptr(), o["classified"].data_ptr(),
                                           o["num_distinct"].data_ptr(), o["total_kmers"].data_ptr(), o["num_hits"].data_ptr())
            with torch.cuda.stream(ext):
                with _StageSpan(torch, steps if profile else None, "step", ext):
                    if emit is not None or lk is not None or ap is not None:
                        try:
                            self.st.shard_step(emit, lk, al, ap)
                        except slacken_amd.SlackenError as e:
                            if e.code != capi.E_UNSUPPORTED or t != 0:
                                raise
                            unsupported = True
                if b is not None:
                    if emit is None:
                        b["cursors"].zero_()
                    b["h_cursors"].copy_(b["cursors"], non_blocking=True)
                    b["stepped"] = torch.cuda.Event()
                    b["stepped"].record(ext)
                if 0 <= t - 2 < n:
                    states[t - 2]["looked"] = torch.cuda.Event()
                    states[t - 2]["looked"].record(ext)
                if 0 <= t - 4 < n:       # the batch is through: its lists go back to the allocator
                    s4 = states[t - 4]
                    for k in ("send_meta", "log", "tile_rows", "read_info", "taxa", "cursors"):
                        s4.pop(k, None)

        def exchange_keys(j):
            """batch j was emitted a step ago: the host's wait for its cursors, then its keys travel (exchange stream)"""
            b = states[j]
            b["stepped"].synchronize()
            hc = [int(v) for v in b["h_cursors"].tolist()]
            send_counts = hc[:W]
            over = any(c > b["cap"] for c in send_counts)
            with torch.cuda.stream(xext):
                xext.wait_event(b["stepped"])
                recv_counts, any_over = self.ex.split_sizes([0] * W if over else send_counts, over)
                if any_over:   # a region was too small somewhere: this batch takes the staged route on every rank
                    b["failed"] = True
                    send_counts, recv_counts = [0] * W, [0] * W
                n_send, n_recv = sum(send_counts), sum(recv_counts)
                regions = [b["send_keys"][g * b["cap"]:g * b["cap"] + send_counts[g]] for g in range(W)]
                with self._ev(b, "exchange_keys", xext):
                    b["recv_keys"] = self.ex.regions_out(regions, recv_counts)
                if self.ex.single:          # the answers are written where the apply reads them: no copy on the way back
                    b["taxa"] = torch.empty(W * b["cap"], dtype=torch.int32, device=dev)
                    b["found"] = b["taxa"]
                else:
                    b["found"] = torch.empty(max(n_recv, 1), dtype=torch.int32, device=dev)
                # (made on one stream, used on the other: the allocator must not hand the blocks out again before that use is over)
                b["send_keys"].record_stream(xext)
                for k in ("recv_keys", "found", "taxa"):
                    if k in b:
                        b[k].record_stream(ext)
                b["keys_here"] = torch.cuda.Event()
                b["keys_here"].record(xext)
            b["looked_up"], b["send_counts"], b["recv_counts"] = n_recv, send_counts, recv_counts
            o = outs[j]
            o["exchanged_keys"], o["looked_up_keys"], o["sent_remote_keys"] = n_send, n_recv, n_send - send_counts[self.rank]
            if not self.ex.single:
                b.pop("send_keys", None)

        def exchange_taxa(j):
            """batch j's lookups ran in the step just finished: the taxa go back (exchange stream)"""
            b = states[j]
            with torch.cuda.stream(xext):
                xext.wait_event(b["looked"])
                if not self.ex.single:
                    b["taxa"] = torch.empty(W * b["cap"], dtype=torch.int32, device=dev)
                    regions = [b["taxa"][g * b["cap"]:g * b["cap"] + b["send_counts"][g]] for g in range(W)]
                    b["taxa"].record_stream(ext)
                    with self._ev(b, "exchange_taxa", xext):
                        self.ex.regions_back(b["found"], b["recv_counts"], regions)
                b["taxa_here"] = torch.cuda.Event()
                b["taxa_here"].record(xext)
            for k in ("recv_keys", "found", "send_keys"):
                b.pop(k, None)

        import os
        import time
        trace = os.environ.get("SLK_SHARDED_TRACE") == "1"     # (tuning aid: where the host's time goes, per step)
        tw = time.perf_counter()
        try:
            for t in range(n + 4):
                t0 = time.perf_counter()
                launch(t)
                t1 = time.perf_counter()
                if t == 0 and self._any_rank(unsupported):
                    states.clear()
                    return None
                if 1 <= t <= n:
                    exchange_keys(t - 1)
                t2 = time.perf_counter()
                if 3 <= t <= n + 2:
                    exchange_taxa(t - 3)
                if trace:
                    print(f"[sharded] step {t}: launch {1e3 * (t1 - t0):.2f} ms, wait + keys {1e3 * (t2 - t1):.2f}, taxa {1e3 * (time.perf_counter() - t2):.2f}",
                          file=sys.stderr, flush=True)
        except BaseException:
            states.clear()     # (nothing of a batch may outlive this call: see close())
            raise
        overflowed = any(s["failed"] for s in states)
        for st in (self.st, self._xst):
            try:
                st.synchronize()
            except slacken_amd.SlackenError as e:
                if not (overflowed and e.code == capi.E_CAPACITY):   # (the engine's own note of an overflow that was handled above)
                    raise
        if profile:
            acc = {}
            for sb in states:
                for name, e0, e1 in sb["prof"]:
                    acc.setdefault(name, []).append(e0.elapsed_time(e1))
            for name, e0, e1 in steps:
                acc.setdefault(name, []).append(e0.elapsed_time(e1))
            self.stage_ms = {k: float(np.mean(v)) for k, v in acc.items()}
            self.step_ms = [e0.elapsed_time(e1) for _, e0, e1 in steps]
        cur.wait_stream(ext)
        if trace:
            print(f"[sharded] pipeline of {n} batches: {1e3 * (time.perf_counter() - tw):.2f} ms", file=sys.stderr, flush=True)
        for i, (b, batch) in enumerate(zip(states, batches)):
            d_bases, d_offsets, R, total_bases, mates = batch
            if b["failed"]:      # the whole batch through the staged route
                outs[i] = dict(self._classify_staged(d_bases, d_offsets, R, total_bases, thresholds, min_hit_groups, mates), deferred=R)
                continue
            outs[i] = self._finish_deferred(outs[i], b, d_bases, d_offsets, R, thresholds, min_hit_groups, mates)
        states.clear()
        return outs

    def jobs_alone(self, batch, thresholds=(0.0,), min_hit_groups=2):
        """world = 1 only: the three jobs of one batch each as a step of its own (the host waits in between), and then together in
        ONE kernel as the pipeline runs them: what riding along buys.  -> {jobs: ms}"""
        import ctypes as C
        import time
        from slacken_amd import capi
        torch, dev = self.torch, self.device
        assert self.ex.single and self.world == 1
        ext, _ = self._streams()
        Cn = len(thresholds)
        thr = (C.c_double * Cn)(*thresholds)
        R = batch[2]

        def timed(f):
            torch.cuda.synchronize()
            self.st.synchronize()
            t0 = time.perf_counter()
            f()
            self.st.synchronize()
            return (time.perf_counter() - t0) * 1e3

        out = {}
        with torch.cuda.stream(ext):
            b0 = self._emit_state(batch, False)
            out["emit"] = timed(lambda: self.st.shard_step(b0["lists"], None, None, None))
            n = int(b0["cursors"][0].item())
            taxa0 = torch.empty(b0["cap"], dtype=torch.int32, device=dev)
            lk = capi.ShardLookup(b0["send_keys"].data_ptr(), n, taxa0.data_ptr())
            out["lookup"] = timed(lambda: self.st.shard_step(None, lk, None, None))
            o0 = [torch.zeros(Cn * max(R, 1), dtype=torch.int32, device=dev), torch.zeros(Cn * max(R, 1), dtype=torch.uint8, device=dev)]
            ap = capi.ShardResults(taxa0.data_ptr(), min_hit_groups, Cn, thr, o0[0].data_ptr(), o0[1].data_ptr(), None, None, None)
            out["apply"] = timed(lambda: self.st.shard_step(None, None, b0["lists"], ap))
            b1 = self._emit_state(batch, False)
            out["emit+lookup"] = timed(lambda: self.st.shard_step(b1["lists"], lk, None, None))
            b2 = self._emit_state(batch, False)
            out["emit+apply"] = timed(lambda: self.st.shard_step(b2["lists"], None, b0["lists"], ap))
            b3 = self._emit_state(batch, False)
            out["emit+lookup+apply"] = timed(lambda: self.st.shard_step(b3["lists"], lk, b0["lists"], ap))
            del b0, b1, b2, b3, taxa0, o0
        return {k: round(v, 3) for k, v in out.items()}

    def _finish_deferred(self, out, b, d_bases, d_offsets, R, thresholds, min_hit_groups, mates):
        """fragments the fused kernel does not take: a compacted batch through the staged route (all ranks, also with none)"""
        torch, dev = self.torch, self.device
        C = len(thresholds)
        mb, mo, mtotal = mates if mates is not None else (None, None, 0)
        idx = torch.nonzero(b["defer"][:R]).flatten() if R else torch.zeros(0, dtype=torch.int64, device=dev)
        out["deferred"] = int(idx.numel())
        if self._any_rank(idx.numel() > 0):
            def compact(bases, offsets):
                lens = offsets[idx + 1] - offsets[idx]
                sub_off = torch.zeros(idx.numel() + 1, dtype=torch.int64, device=dev)
                sub_off[1:] = torch.cumsum(lens, 0)
                sub_total = int(sub_off[-1].item()) if idx.numel() else 0
                src = torch.repeat_interleave(offsets[idx], lens) + (torch.arange(sub_total, device=dev) -
                                                                     torch.repeat_interleave(sub_off[:-1], lens))
                return (bases[src] if sub_total else torch.zeros(1, dtype=torch.uint8, device=dev)), sub_off, sub_total
            sub_bases, sub_off, sub_total = compact(d_bases, d_offsets)
            sub_mates = compact(mb, mo) if mates is not None else None
            sub = self._classify_staged(sub_bases, sub_off, int(idx.numel()), sub_total, thresholds, min_hit_groups, sub_mates)
            n = int(idx.numel())
            if n:
                for c in range(C):
                    out["taxon"][c * R + idx] = sub["taxon"][c * n:(c + 1) * n]
                    out["classified"][c * R + idx] = sub["classified"][c * n:(c + 1) * n]
                for k in ("num_distinct", "total_kmers", "num_hits"):
                    out[k][idx] = sub[k][:n]
            out["exchanged_keys"] += sub["exchanged_keys"]
        return out

    def _classify_staged(self, d_bases, d_offsets, R, total_bases, thresholds=(0.0,), min_hit_groups=2, mates=None):
        torch, dev = self.torch, self.device
        mb, mo, mtotal = mates if mates is not None else (None, None, 0)
        mkw = dict(d_mate_bases=mb.data_ptr(), d_mate_offsets=mo.data_ptr()) if mates is not None else {}
        slots = total_bases + mtotal + R + 1
        keys = torch.empty(slots, dtype=torch.int64, device=dev)
        meta = torch.empty(slots, dtype=torch.int32, device=dev)
        count = torch.zeros(max(R, 1), dtype=torch.int32, device=dev)
        taxon = torch.zeros(slots, dtype=torch.int32, device=dev)
        torch.cuda.current_stream().synchronize()    # (the engine's stream is not torch's: order the two explicitly)
        self.st.scan_device(d_bases.data_ptr(), d_offsets.data_ptr(), R, keys.data_ptr(), meta.data_ptr(), count.data_ptr(), **mkw)
        self.st.synchronize()
        cnt = count[:R].long()
        total = int(cnt.sum().item())
        starts = torch.cumsum(cnt, 0) - cnt
        region = d_offsets[:R] if mates is None else d_offsets[:R] + mo[:R] + torch.arange(R, device=dev)  # engine.h span_region
        slot = torch.repeat_interleave(region, cnt) + (torch.arange(total, device=dev) - torch.repeat_interleave(starts, cnt))
        seq = ((meta[slot] >> 1) & 7) == 1
        seq_slot = slot[seq]
        k = keys[seq_slot]
        owner = shard_of_torch(k, self.world)
        order = torch.argsort(owner, stable=True)
        send_counts = torch.bincount(owner, minlength=self.world).tolist()
        recv_keys, recv_counts = self._all_to_all(k[order].contiguous(), send_counts)
        found = torch.zeros(max(recv_keys.numel(), 1), dtype=torch.int32, device=dev)
        if recv_keys.numel():
            recv_keys = recv_keys.contiguous()
            torch.cuda.current_stream().synchronize()
            self.st.lookup_device(recv_keys.data_ptr(), recv_keys.numel(), found.data_ptr())
            self.st.synchronize()
        back, _ = self._all_to_all(found[:recv_keys.numel()].contiguous(), recv_counts)
        taxa_seq = torch.empty_like(back)
        taxa_seq[order] = back
        taxon[seq_slot] = taxa_seq
        C = len(thresholds)
        out = dict(taxon=torch.zeros(C * max(R, 1), dtype=torch.int32, device=dev),
                   classified=torch.zeros(C * max(R, 1), dtype=torch.uint8, device=dev),
                   num_distinct=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                   total_kmers=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                   num_hits=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                   exchanged_keys=int(k.numel()))
        scratch = keys  # the key slots are dead after the exchange
        torch.cuda.current_stream().synchronize()
        self.st.classify_hits_device(d_offsets.data_ptr(), R, meta.data_ptr(), taxon.data_ptr(), count.data_ptr(),
                                     scratch.data_ptr(), out["taxon"].data_ptr(), out["classified"].data_ptr(),
                                     out["num_distinct"].data_ptr(), out["total_kmers"].data_ptr(),
                                     out["num_hits"].data_ptr(), min_hit_groups=min_hit_groups, thresholds=thresholds,
                                     **({"d_mate_offsets": mo.data_ptr()} if mates is not None else {}))
        self.st.synchronize()
        return out

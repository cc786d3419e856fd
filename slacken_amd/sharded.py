"""Table-sharded classify (BASELINE.json configs[3]): the record table is too large for one GPU's HBM, so
rank g holds the records with fmix64(key) mod G == g.  Per batch every rank scans ITS reads, sends each minimizer to
its owner (all-to-all of 8-byte keys over RCCL/xGMI), owners look them up in their shard, the taxa return by the inverse
all-to-all (4 bytes each), and every rank finishes its own reads' LCA locally.  This replaces the reference's shuffle +
join (S/slacken/Classifier.scala:84) for the one case where data must move; with a table that fits, use the replicated
mode (no collective at all).

Host plumbing only: torch for device buffers and the collectives, the engine for all compute.  Two routes:
  fast    slk_shard_emit_device -> slk_shard_compact_device -> all-to-all -> slk_lookup_device -> all-to-all ->
          slk_shard_apply_device: the fused lane-per-fragment kernel runs on both sides of the exchange, nothing but 8-byte keys
          and 4-byte taxa moves, and the fragments are scanned once (the apply replays the emit's log); takes fragments of up to 1000
          bases with at most 12 distinct taxa; classify_many runs the scans on one stream and the memory-bound stages on another;
  staged  slk_scan_device / slk_lookup_device / slk_classify_hits_device with the exchange lists built by torch ops: takes
          everything, and the fragments the fast route hands back (`defer`)."""
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
    if that is None).  A plain module-level class: the batch's tensors live on the ENGINE's streams, and must die by reference
    count, before their stream does -- an object that only the cycle collector frees (a class made per call is one) would keep
    them past it."""
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
    split sizes travel first, and with them one flag word per rank (a send list overflowed: emit again) -- so the decision every
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

    def any_rank(self, flag):
        """logical OR of a host flag over all ranks (every rank must take the same sequence of collectives)"""
        if self.single:
            return bool(flag)
        t = self.torch.tensor([1 if flag else 0], dtype=self.torch.int64, device="cpu" if self.on_cpu else self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return bool(int(t.item()))


class ShardedClassifier:
    """index: this rank's slacken_amd.Index holding ONLY the records with shard_of(key) == rank (plus the taxonomy)."""

    def __init__(self, index, rank=0, world=1, dist=None, device=None, exchange_on_cpu=False, force_collectives=False):
        import torch
        self.torch, self.ix, self.st = torch, index, index.stream()
        self.rank, self.world, self.dist = rank, world, dist
        self.device = device if device is not None else torch.device("cuda", 0)
        self.on_cpu = exchange_on_cpu
        self.ex = Exchange(rank, world, dist, self.device, exchange_on_cpu, force_collectives)
        self.stage_ms = None   # classify_many(profile=True): per-stage device times of the batches

    def _all_to_all(self, send, send_counts):
        """send: tensor sorted by destination rank; returns (received tensor, recv_counts list)."""
        recv_counts, _ = self.ex.split_sizes(send_counts)
        return self.ex.all_to_all(send, send_counts, recv_counts), recv_counts

    def classify(self, d_bases, d_offsets, R, total_bases, thresholds=(0.0,), min_hit_groups=2, fast=True, d_mate_bases=None,
                 d_mate_offsets=None, total_mate_bases=0):
        """d_bases uint8 [total_bases], d_offsets int64 [R+1] (device tensors); second mates likewise (optional)."""
        mates = (d_mate_bases, d_mate_offsets, total_mate_bases) if d_mate_bases is not None else None
        if fast:
            out = self._classify_fast(d_bases, d_offsets, R, total_bases, thresholds, min_hit_groups, mates)
            if out is not None:
                return out
        return self._classify_staged(d_bases, d_offsets, R, total_bases, thresholds, min_hit_groups, mates)

    def _any_rank(self, flag):
        """logical OR of a host flag over all ranks, on the memory stream (never torch's default stream, see _two_streams)"""
        if self.ex.single:
            return bool(flag)
        _, (_, ext) = self._two_streams()
        with self.torch.cuda.stream(ext):
            return self.ex.any_rank(flag)

    # ---- the fast route, in stages that are interleaved between batches (classify_many) -------------------------------------
    # The stages of a batch are bound by different things: the scan (emit) by instruction issue, compaction, the owners' lookup and
    # the replay of the batches (apply) by HBM, the exchange by the links.  So the scans of all batches go to ONE engine stream and
    # the other stages to ANOTHER, each wrapped as a torch ExternalStream so that torch kernels, collectives and engine kernels are ordered
    # by the streams themselves, with an event where the memory stream needs the scan's result.  In steady state the scan stream
    # runs emit(k+1) while the memory stream runs compact(k), lookup(k), apply(k).
    # (Nothing here touches torch's default stream: the engine's streams are blocking streams, and an event recorded on the legacy
    # null stream waits for all of them and holds back whatever is issued after it -- measured: it serialised the pipeline.)
    def _two_streams(self):
        if not hasattr(self, "_scan"):
            torch = self.torch
            self._scan = (self.st, torch.cuda.ExternalStream(self.st.hip_stream, device=self.device))
            mem = self.ix.stream()
            self._mem = (mem, torch.cuda.ExternalStream(mem.hip_stream, device=self.device))
        return self._scan, self._mem

    def _compute(self):
        """(engine stream, torch view of it) the memory-bound stages -- compaction, lookups, apply -- are launched on: the memory
        stream beside the scans (round 2's pipeline), or the scan stream itself, one kernel after the other (the fused pipeline: on a
        table of HBM size kernels side by side cost more than they hide)"""
        scan, mem = self._two_streams()
        return scan if getattr(self, "_serial_compute", False) else mem

    def _xstream(self):
        """the stream the exchanges are issued on: the memory stream (beside the compute stream in the fused pipeline)"""
        return self._two_streams()[1][1]

    def _ev(self, b, name, ext):
        """(profile) an event pair around a stage of batch b on stream ext: a context manager"""
        return _StageSpan(self.torch, b["prof"] if b is not None else None, name, ext)

    def _fast_emit(self, batch, cap_scale=1, profile=False, side=None):
        """stage 1 (scan stream, asynchronous): scan + send lists.  None if this index's splitter only has the staged route.
        side: the state of an EARLIER batch whose received keys are answered on the way (slk_shard_emit_lookup_device)."""
        import slacken_amd
        from slacken_amd import capi
        torch, dev, W = self.torch, self.device, self.world
        (st, ext), _ = self._two_streams()
        d_bases, d_offsets, R, total_bases, mates = batch
        mb, mo, mtotal = mates if mates is not None else (None, None, 0)
        mkw = dict(d_mate_bases=mb.data_ptr(), d_mate_offsets=mo.data_ptr()) if mates is not None else {}
        # about 0.26 probes per base on random sequence, spread evenly by the hash over W owners and by the tile index over SUB
        # sub-lists per owner (each fed by at least 64 tiles, so that the spread holds): 0.6 / (W * SUB) per base leaves 2x
        # headroom; a list that overflows all the same is reported by the compaction, and the batch is emitted again with more room
        tiles = (R + 63) // 64
        SUB = 1
        while SUB < 256 and SUB * 2 * 64 <= tiles:
            SUB *= 2
        cap = (int((total_bases + mtotal) * 0.6 / (W * SUB)) + (1 << 12)) * cap_scale
        rows = int(capi.lib().slk_shard_batch_rows(total_bases, mtotal, R, 1 if mates is not None else 0))
        cur = torch.cuda.current_stream()
        if cur != torch.cuda.default_stream(self.device):     # (the caller's tensors were produced on ITS stream)
            ext.wait_stream(cur)
        b = dict(R=R, SUB=SUB, cap=cap, mkw=mkw, batch=batch, prof=[] if profile else None)
        side_args = None
        if side is not None and side["looked_up"] and R:
            ext.wait_event(side["keys_here"])         # (its keys have arrived)
            # the keys' batches of 64 are dealt out to this scan's tiles; what a tile does not get to is on file in side["done"]
            side["per_tile"], side["tiles"] = -(-(-(-side["looked_up"] // 64)) // tiles), tiles
            with torch.cuda.stream(ext):
                side["done"] = torch.empty(tiles, dtype=torch.int32, device=dev)
            side_args = (side["recv_keys"].data_ptr(), side["looked_up"], side["per_tile"], side["done"].data_ptr(), side["found"].data_ptr())
        with torch.cuda.stream(ext):
            defer = torch.empty(max(R, 1), dtype=torch.int32, device=dev)
            batch_base = torch.empty(rows * W, dtype=torch.int32, device=dev)
            send_keys = torch.empty(W * SUB * cap, dtype=torch.int64, device=dev)
            send_meta = torch.empty(W * SUB * cap, dtype=torch.int32, device=dev)
            counts = torch.empty(W * SUB, dtype=torch.int64, device=dev)
            tile_rows = torch.empty(tiles + 1, dtype=torch.int32, device=dev)
            read_info = torch.empty(2 * max(R, 1), dtype=torch.int32, device=dev)
            try:
                with self._ev(b, "emit+lookup" if side_args else "emit", ext):
                    st.shard_emit_device(d_bases.data_ptr(), d_offsets.data_ptr(), R, W, SUB, send_keys.data_ptr(), send_meta.data_ptr(), cap,
                                         counts.data_ptr(), batch_base.data_ptr(), tile_rows.data_ptr(), read_info.data_ptr(),
                                         defer.data_ptr(), side=side_args, **mkw)
            except slacken_amd.SlackenError as e:
                if e.code != capi.E_UNSUPPORTED:
                    raise
                return None
            emitted = torch.cuda.Event()
            emitted.record(ext)
        b.update(defer=defer, batch_base=batch_base, send_keys=send_keys, send_meta=send_meta, counts=counts,
                 tile_rows=tile_rows, read_info=read_info, emitted=emitted)
        return b

    def _fast_compact(self, b):
        """stage 2 (asynchronous): the send lists back to back; the split sizes on their way to the host"""
        torch, dev, W = self.torch, self.device, self.world
        st, ext = self._compute()
        with torch.cuda.stream(ext):
            ext.wait_event(b["emitted"])
            b["out_keys"] = torch.empty_like(b["send_keys"])   # (room for every list at its capacity; the used prefix is what is sent)
            b["list_off"] = torch.empty(W * b["SUB"] + 1, dtype=torch.int64, device=dev)
            owner_counts = torch.empty(W + 1, dtype=torch.int64, device=dev)
            with self._ev(b, "compact", ext):
                st.shard_compact_device(b["send_keys"].data_ptr(), W, b["SUB"], b["cap"], b["counts"].data_ptr(), b["out_keys"].data_ptr(),
                                        b["list_off"].data_ptr(), owner_counts.data_ptr())
            b["h_counts"] = torch.empty(W + 1, dtype=torch.int64, pin_memory=True)
            b["h_counts"].copy_(owner_counts, non_blocking=True)
            b["ready"] = torch.cuda.Event()
            b["ready"].record(ext)

    def _exchange_keys(self, b, send_counts, recv_counts):
        """stage 3a (exchange stream): keys to their owners"""
        torch = self.torch
        ext = self._xstream()
        n_send, n_recv = sum(send_counts), sum(recv_counts)
        cext = self._compute()[1]
        with torch.cuda.stream(ext):
            ext.wait_event(b["ready"])
            b["out_keys"].record_stream(ext)      # (made on the compute stream, read here: the allocator must not hand it out before)
            with self._ev(b, "exchange_keys", ext):
                recv_keys = self.ex.all_to_all(b["out_keys"][:n_send], send_counts, recv_counts)
            b["recv_keys"] = recv_keys.contiguous() if n_recv else torch.zeros(1, dtype=torch.int64, device=self.device)
            b["found"] = torch.empty(max(n_recv, 1), dtype=torch.int32, device=self.device)
            for tns in (b["recv_keys"], b["found"]):   # (made here, used by the lookups on the compute stream -- and the scan's side job)
                tns.record_stream(cext)
                tns.record_stream(self._two_streams()[0][1])
            b["keys_here"] = torch.cuda.Event()
            b["keys_here"].record(ext)
        b["exchanged"], b["looked_up"] = n_send, n_recv
        b["send_counts"], b["recv_counts"] = send_counts, recv_counts
        b["sent_remote"] = n_send - send_counts[self.rank]
        del b["send_keys"], b["out_keys"]

    def _lookup_and_return(self, b, after=None):
        """stage 3b: the keys a scan's side job has not answered (all of them if none ran) on the compute stream, then the taxa back
        on the exchange stream.  after: the event of the emit launch that carried the side job."""
        torch = self.torch
        st, ext = self._compute()
        xs = self._xstream()
        n_recv = b["looked_up"]
        with torch.cuda.stream(ext):
            ext.wait_event(b["keys_here"])
            if after is not None:
                ext.wait_event(after)
            if n_recv and "done" in b:      # a later scan answered most of them: the rest
                with self._ev(b, "lookup_rest", ext):
                    st.lookup_rest_device(b["recv_keys"].data_ptr(), n_recv, b["per_tile"], b["tiles"], b["done"].data_ptr(), b["found"].data_ptr())
            elif n_recv:
                with self._ev(b, "lookup", ext):
                    st.lookup_device(b["recv_keys"].data_ptr(), n_recv, b["found"].data_ptr())
            looked = torch.cuda.Event()
            looked.record(ext)
        with torch.cuda.stream(xs):
            xs.wait_event(looked)
            with self._ev(b, "exchange_taxa", xs):
                back = self.ex.all_to_all(b["found"][:n_recv], b["recv_counts"], b["send_counts"])
            b["taxa"] = back.contiguous() if back.numel() else torch.zeros(1, dtype=torch.int32, device=self.device)
            b["taxa"].record_stream(ext)          # (read by the apply on the compute stream)
            b["taxa_here"] = torch.cuda.Event()
            b["taxa_here"].record(xs)
        for k in ("recv_keys", "found", "done"):
            b.pop(k, None)

    def _fast_exchange(self, b, send_counts, recv_counts):
        """stage 3 (memory stream): keys to their owners, lookup, taxa back"""
        self._exchange_keys(b, send_counts, recv_counts)
        self._lookup_and_return(b)

    def _fast_apply(self, b, thresholds, min_hit_groups):
        """stage 4 (memory stream, behind the lookup; asynchronous): the batches of probes are replayed from the emit's log and
        folded with the owners' answers -- no second scan.  The result tensors are valid once that stream has been synchronised."""
        torch, dev, R, W = self.torch, self.device, b["R"], self.world
        st, ext = self._compute()
        d_bases, d_offsets = b["batch"][0], b["batch"][1]
        C = len(thresholds)
        with torch.cuda.stream(ext):
            ext.wait_event(b["taxa_here"])
            out = dict(taxon=torch.zeros(C * max(R, 1), dtype=torch.int32, device=dev),
                       classified=torch.zeros(C * max(R, 1), dtype=torch.uint8, device=dev),
                       num_distinct=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                       total_kmers=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                       num_hits=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                       exchanged_keys=b["exchanged"], looked_up_keys=b["looked_up"], sent_remote_keys=b["sent_remote"])
            with self._ev(b, "apply", ext):
                st.shard_apply_device(d_bases.data_ptr(), d_offsets.data_ptr(), R, W, b["SUB"], b["cap"], b["taxa"].data_ptr(),
                                      b["list_off"].data_ptr(), b["send_meta"].data_ptr(), b["batch_base"].data_ptr(),
                                      b["tile_rows"].data_ptr(), b["read_info"].data_ptr(), out["taxon"].data_ptr(),
                                      out["classified"].data_ptr(), b["defer"].data_ptr(), out["num_distinct"].data_ptr(),
                                      out["total_kmers"].data_ptr(), out["num_hits"].data_ptr(), min_hit_groups=min_hit_groups,
                                      thresholds=thresholds, **b["mkw"])
            b["applied"] = torch.cuda.Event()
            b["applied"].record(ext)
        return out

    def _classify_fast(self, d_bases, d_offsets, R, total_bases, thresholds, min_hit_groups, mates=None):
        outs = self.classify_many([(d_bases, d_offsets, R, total_bases, mates)], thresholds, min_hit_groups)
        return None if outs is None else outs[0]

    def classify_many(self, batches, thresholds=(0.0,), min_hit_groups=2, profile=False, fused_lookup=True):
        """The fast route over several batches [(d_bases, d_offsets, R, total_bases, mates or None)], two in flight (see above).
        fused_lookup (the default): the keys a rank RECEIVES for batch t - 2 are answered inside the scan of batch t
        (_classify_many_fused); False: by a lookup kernel of their own beside the scan of batch t + 1, as in round 2.
        Every rank must pass the same number of batches.  Returns the list of result dicts, or None if the splitter only has the
        staged route.  profile: self.stage_ms = {stage: mean device ms per batch} from events around every stage (the stages of
        neighbouring batches overlap on the two streams: these are their durations IN the pipeline, not alone)."""
        import slacken_amd
        from slacken_amd import capi
        torch = self.torch
        if fused_lookup:
            return self._classify_many_fused(batches, thresholds, min_hit_groups, profile)
        states, outs = [], []
        overflowed = False
        _, (_, mem_ext) = self._two_streams()

        def settle(j):   # batch j: its one host wait, then exchange + lookup + apply on the memory stream
            nonlocal overflowed
            b, scale = states[j], 1
            while True:
                b["ready"].synchronize()
                send_counts = [int(v) for v in b["h_counts"][:self.world].tolist()]
                # (the split sizes carry every rank's overflow flag: all ranks take the same branch without a collective of its own)
                with torch.cuda.stream(mem_ext):
                    recv_counts, over = self.ex.split_sizes(send_counts, int(b["h_counts"][self.world]) != 0)
                if not over:
                    break
                overflowed = True                      # (rare: a send list overflowed somewhere -- every rank emits again)
                scale *= 2
                b = states[j] = self._fast_emit(batches[j], scale, profile)
                self._fast_compact(b)
            self._fast_exchange(b, send_counts, recv_counts)
            outs.append(self._fast_apply(b, thresholds, min_hit_groups))

        def release_finished():   # batches whose apply has run give their device memory back (a long run holds a few, not all)
            for sb in states:
                if "applied" in sb and "taxa" in sb and sb["applied"].query():
                    for k in ("taxa", "batch_base", "list_off", "counts", "send_meta", "tile_rows", "read_info"):
                        sb.pop(k, None)

        for i, batch in enumerate(batches):
            release_finished()
            b = self._fast_emit(batch, 1, profile)     # scan stream
            if i == 0 and self._any_rank(b is None):
                return None
            states.append(b)
            if i >= 1:
                settle(i - 1)                          # lookup(i-1) and apply(i-1) run beside emit(i)
            self._fast_compact(b)                      # memory stream, behind apply(i-1)
        if states:
            settle(len(states) - 1)
        (scan_st, _), (mem_st, _) = self._two_streams()
        for st in (scan_st, mem_st):
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
            self.stage_ms = {k: float(np.mean(v)) for k, v in acc.items()}
        for i, (b, batch) in enumerate(zip(states, batches)):
            for k in ("taxa", "batch_base", "list_off", "counts", "send_meta", "tile_rows", "read_info"):
                b.pop(k, None)
            d_bases, d_offsets, R, total_bases, mates = batch
            outs[i] = self._finish_deferred(outs[i], b, d_bases, d_offsets, R, thresholds, min_hit_groups, mates)
        return outs

    def _classify_many_fused(self, batches, thresholds, min_hit_groups, profile):
        """Kernels side by side cost more than they hide on a table of HBM size: the scan and a lookup kernel share one request rate
        and the wave slots (lookup 8-13 ms alone, 14 beside the scan, the scan 6.5 alone and 8-10 beside it).  The local
        kernel does not have that problem -- its probes hide behind its own scan -- so the sharded scan takes the owner's lookups
        along the same way: the scan of batch t answers the keys this rank received for batch t - 2 (their exchange ran during the
        scan of batch t - 1), and the other stages follow it on the SAME stream, one kernel after the other.  Per iteration t:
          compute stream   emit(t) + lookups(t - 2) | what is left of lookups(t - 2) | compact(t) | apply(t - 2)
          host             waits for the split sizes of batch t - 1 (while emit(t) runs)
          exchange stream  keys(t - 1) to their owners (beside emit(t)) | taxa(t - 2) back (beside compact(t))
        Results come out two batches late; the last two batches' lookups run as a kernel of their own."""
        import slacken_amd
        from slacken_amd import capi
        torch = self.torch
        n = len(batches)
        states, outs = [None] * n, [None] * n
        _, (_, mem_ext) = self._two_streams()
        self._serial_compute = True
        try:
            return self._fused_loop(batches, thresholds, min_hit_groups, profile, states, outs, mem_ext)
        finally:
            self._serial_compute = False

    def _fused_loop(self, batches, thresholds, min_hit_groups, profile, states, outs, mem_ext):
        import slacken_amd
        from slacken_amd import capi
        torch = self.torch
        n = len(batches)
        overflowed = False

        def settle(j):   # batch j: the host's wait for its split sizes, then its keys travel
            nonlocal overflowed
            b, scale = states[j], 1
            while True:
                b["ready"].synchronize()
                send_counts = [int(v) for v in b["h_counts"][:self.world].tolist()]
                with torch.cuda.stream(mem_ext):
                    recv_counts, over = self.ex.split_sizes(send_counts, int(b["h_counts"][self.world]) != 0)
                if not over:
                    break
                overflowed = True                      # (rare: a send list overflowed somewhere -- every rank emits again, without a side job)
                scale *= 2
                b = states[j] = self._fast_emit(batches[j], scale, profile)
                self._fast_compact(b)
            self._exchange_keys(b, send_counts, recv_counts)

        for t in range(n + 2):
            if t < n:
                b = self._fast_emit(batches[t], 1, profile, side=states[t - 2] if t >= 2 else None)
                if t == 0 and self._any_rank(b is None):
                    return None
                states[t] = b
            if 1 <= t <= n:
                settle(t - 1)
            if t >= 2:
                self._lookup_and_return(states[t - 2], after=states[t]["emitted"] if t < n else None)
            if t < n:
                self._fast_compact(states[t])                 # (beside the taxa's way back)
            if t >= 2:
                outs[t - 2] = self._fast_apply(states[t - 2], thresholds, min_hit_groups)
            for sb in states:   # batches whose apply has run give their device memory back
                if sb is not None and "applied" in sb and "taxa" in sb and sb["applied"].query():
                    for k in ("taxa", "batch_base", "list_off", "counts", "send_meta", "tile_rows", "read_info"):
                        sb.pop(k, None)
        (scan_st, _), (mem_st, _) = self._two_streams()
        for st in (scan_st, mem_st):
            try:
                st.synchronize()
            except slacken_amd.SlackenError as e:
                if not (overflowed and e.code == capi.E_CAPACITY):
                    raise
        if profile:
            acc = {}
            for sb in states:
                for name, e0, e1 in sb["prof"]:
                    acc.setdefault(name, []).append(e0.elapsed_time(e1))
            self.stage_ms = {k: float(np.mean(v)) for k, v in acc.items()}
        for i, (b, batch) in enumerate(zip(states, batches)):
            for k in ("taxa", "batch_base", "list_off", "counts", "send_meta", "tile_rows", "read_info"):
                b.pop(k, None)
            d_bases, d_offsets, R, total_bases, mates = batch
            outs[i] = self._finish_deferred(outs[i], b, d_bases, d_offsets, R, thresholds, min_hit_groups, mates)
        return outs

    def stage_times_alone(self, batch, thresholds=(0.0,), min_hit_groups=2):
        """Every stage of one batch run ALONE (a synchronisation between stages; world = 1 only): what the stages cost without
        each other's company, next to their times in the pipeline (classify_many(profile=True)).  -> {stage: ms}"""
        import time
        torch = self.torch
        assert self.ex.single
        (scan_st, _), (mem_st, mem_ext) = self._two_streams()

        def timed(f):
            torch.cuda.synchronize(); scan_st.synchronize(); mem_st.synchronize()
            t0 = time.perf_counter()
            r = f()
            scan_st.synchronize(); mem_st.synchronize(); torch.cuda.synchronize()
            return r, (time.perf_counter() - t0) * 1e3

        out = {}
        b0, out["emit"] = timed(lambda: self._fast_emit(batch))
        _, out["compact"] = timed(lambda: self._fast_compact(b0))
        counts = [int(v) for v in b0["h_counts"][:1].tolist()]
        self._exchange_keys(b0, counts, counts)
        b1, out["emit+lookup"] = timed(lambda: self._fast_emit(batch, side=b0))
        self._fast_compact(b1)
        self._exchange_keys(b1, counts, counts)
        _, out["lookup_rest"] = timed(lambda: self._lookup_and_return(b0))
        _, out["apply"] = timed(lambda: self._fast_apply(b0, thresholds, min_hit_groups))
        _, out["lookup"] = timed(lambda: self._lookup_and_return(b1))
        self._fast_apply(b1, thresholds, min_hit_groups)
        scan_st.synchronize(); mem_st.synchronize()
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

"""Table-sharded classify (BASELINE.json configs[3]): the record table is too large for one GPU's HBM, so
rank g holds the records with fmix64(key) mod G == g.  Per batch every rank scans ITS reads, sends each minimizer to
its owner (all-to-all of 8-byte keys over RCCL/xGMI), owners look them up in their shard, the taxa return by the inverse
all-to-all (4 bytes each), and every rank finishes its own reads' LCA locally.  This replaces the reference's shuffle +
join (S/slacken/Classifier.scala:84) for the one case where data must move; with a table that fits, use the replicated
mode (no collective at all).

Host plumbing only: torch for device buffers and the collectives, the engine for all compute.  Two routes:
  fast    slk_shard_emit_device -> slk_shard_compact_device -> all-to-all -> slk_lookup_device -> all-to-all ->
          slk_shard_apply_device: the fused lane-per-fragment kernel runs on both sides of the exchange, nothing but 8-byte keys
          and 4-byte taxa moves; takes fragments of up to 1000 bases with at most 12 distinct taxa; classify_many keeps two
          batches in flight so that the exchange of one overlaps the scans of its neighbours;
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


class ShardedClassifier:
    """index: this rank's slacken_amd.Index holding ONLY the records with shard_of(key) == rank (plus the taxonomy)."""

    def __init__(self, index, rank=0, world=1, dist=None, device=None, exchange_on_cpu=False):
        import torch
        self.torch, self.ix, self.st = torch, index, index.stream()
        self.rank, self.world, self.dist = rank, world, dist
        self.device = device if device is not None else torch.device("cuda", 0)
        self.on_cpu = exchange_on_cpu  # gloo has no device all-to-all: stage the exchange through host memory (tests)

    def _all_to_all(self, send, send_counts):
        """send: tensor sorted by destination rank; returns (received tensor, recv_counts list)."""
        torch = self.torch
        if self.world == 1 or self.dist is None:
            return send, list(send_counts)
        dev = "cpu" if self.on_cpu else self.device
        sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
        rcnt = torch.empty(self.world, dtype=torch.int64, device=dev)
        self.dist.all_to_all_single(rcnt, sc)
        recv_counts = [int(v) for v in rcnt.tolist()]
        src = send.to(dev)
        out = torch.empty(sum(recv_counts), dtype=send.dtype, device=dev)
        self.dist.all_to_all_single(out, src, output_split_sizes=recv_counts, input_split_sizes=list(send_counts))
        return out.to(self.device), recv_counts

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
        """logical OR of a host flag over all ranks (every rank must take the same sequence of collectives)"""
        if self.world == 1 or self.dist is None:
            return bool(flag)
        t = self.torch.tensor([1 if flag else 0], dtype=self.torch.int64, device="cpu" if self.on_cpu else self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return bool(int(t.item()))

    # ---- the fast route, in stages that are interleaved between batches (classify_many) -------------------------------------
    # All torch work of a batch is issued on the ENGINE's stream (wrapped as a torch ExternalStream): torch kernels,
    # collectives and engine kernels of one batch are ordered by that stream itself, whatever stream the caller works on.
    # Two batches on two engine streams overlap where the hardware has room: emit and apply are bound by instruction issue,
    # the owners' lookup by HBM requests, the exchange by the links.
    N_STREAMS = int(__import__("os").environ.get("SLK_SHARD_STREAMS", "2"))   # batches in flight (classify_many)

    def _stream(self, which=0):
        if not hasattr(self, "_streams"):
            self._streams = [self.st] + [self.ix.stream() for _ in range(self.N_STREAMS - 1)]
            self._ext = [self.torch.cuda.ExternalStream(s.hip_stream, device=self.device) for s in self._streams]
        return self._streams[which], self._ext[which]

    def _fast_emit(self, which, d_bases, d_offsets, R, total_bases, mates, cap_scale=1):
        """stage 1 (asynchronous): scan + send lists + their compaction.  Returns the batch's state, or None if this index's
        splitter only has the staged route."""
        import slacken_amd
        from slacken_amd import capi
        torch, dev, W = self.torch, self.device, self.world
        st, ext = self._stream(which)
        mb, mo, mtotal = mates if mates is not None else (None, None, 0)
        mkw = dict(d_mate_bases=mb.data_ptr(), d_mate_offsets=mo.data_ptr()) if mates is not None else {}
        # about 0.26 probes per base on random sequence, spread evenly by the hash over W owners and by the tile index over SUB
        # sub-lists per owner (each fed by at least 64 tiles, so that the spread holds): 0.6 / (W * SUB) per base leaves 2x
        # headroom; a list that overflows all the same makes the engine say so, and the batch is emitted again with more room
        tiles = (R + 63) // 64
        SUB = 1
        while SUB < 256 and SUB * 2 * 64 <= tiles:
            SUB *= 2
        cap = (int((total_bases + mtotal) * 0.6 / (W * SUB)) + (1 << 12)) * cap_scale
        rows = int(capi.lib().slk_shard_batch_rows(total_bases, mtotal, R, 1 if mates is not None else 0))
        ext.wait_stream(torch.cuda.current_stream())    # the caller's tensors were produced on ITS stream
        with torch.cuda.stream(ext):
            defer = torch.empty(max(R, 1), dtype=torch.int32, device=dev)
            batch_base = torch.empty(rows * W, dtype=torch.int32, device=dev)
            send_keys = torch.empty(W * SUB * cap, dtype=torch.int64, device=dev)
            counts = torch.empty(W * SUB, dtype=torch.int64, device=dev)
            try:
                st.shard_emit_device(d_bases.data_ptr(), d_offsets.data_ptr(), R, W, SUB, send_keys.data_ptr(), cap, counts.data_ptr(),
                                     batch_base.data_ptr(), defer.data_ptr(), **mkw)
            except slacken_amd.SlackenError as e:
                if e.code != capi.E_UNSUPPORTED:
                    raise
                return None
            out_keys = torch.empty_like(send_keys)   # (room for every list at its capacity; the used prefix is what is sent)
            list_off = torch.empty(W * SUB + 1, dtype=torch.int64, device=dev)
            owner_counts = torch.empty(W, dtype=torch.int64, device=dev)
            st.shard_compact_device(send_keys.data_ptr(), W, SUB, cap, counts.data_ptr(), out_keys.data_ptr(), list_off.data_ptr(),
                                    owner_counts.data_ptr())
            h_counts = torch.empty(W, dtype=torch.int64, pin_memory=True)
            h_counts.copy_(owner_counts, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(ext)
        return dict(st=st, ext=ext, R=R, SUB=SUB, defer=defer, batch_base=batch_base, send_keys=send_keys, counts=counts,
                    out_keys=out_keys, list_off=list_off, h_counts=h_counts, ready=ready, mkw=mkw, d_bases=d_bases,
                    d_offsets=d_offsets, mates=mates)

    def _fast_ready(self, b):
        """the batch's only host wait: W numbers (the exchange's split sizes) and the engine's status word.  False: a send list
        overflowed its capacity."""
        import slacken_amd
        from slacken_amd import capi
        b["ready"].synchronize()
        try:
            b["st"].synchronize()
        except slacken_amd.SlackenError as e:
            if e.code == capi.E_CAPACITY:
                return False
            raise
        return True

    def _fast_exchange(self, b):
        """stage 2 (asynchronous but for the collectives' own host side): keys to their owners, lookup, taxa back"""
        torch = self.torch
        send_counts = [int(v) for v in b["h_counts"].tolist()]
        n_send = sum(send_counts)
        with torch.cuda.stream(b["ext"]):
            recv_keys, recv_counts = self._all_to_all(b["out_keys"][:n_send], send_counts)
            found = torch.empty(max(recv_keys.numel(), 1), dtype=torch.int32, device=self.device)
            if recv_keys.numel():
                recv_keys = recv_keys.contiguous()
                b["st"].lookup_device(recv_keys.data_ptr(), recv_keys.numel(), found.data_ptr())
            back, _ = self._all_to_all(found[:recv_keys.numel()], recv_counts)
            b["taxa"] = back.contiguous() if back.numel() else torch.zeros(1, dtype=torch.int32, device=self.device)
        b["exchanged"] = n_send
        b["recv_keys"] = recv_keys          # (kept alive until the lookup has run)
        del b["send_keys"], b["out_keys"]

    def _fast_apply(self, b, thresholds, min_hit_groups):
        """stage 3 (asynchronous): the second scan.  The result tensors are valid once the batch's stream has been synchronised."""
        torch, dev, R, W = self.torch, self.device, b["R"], self.world
        C = len(thresholds)
        with torch.cuda.stream(b["ext"]):
            out = dict(taxon=torch.zeros(C * max(R, 1), dtype=torch.int32, device=dev),
                       classified=torch.zeros(C * max(R, 1), dtype=torch.uint8, device=dev),
                       num_distinct=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                       total_kmers=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                       num_hits=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                       exchanged_keys=b["exchanged"])
            b["st"].shard_apply_device(b["d_bases"].data_ptr(), b["d_offsets"].data_ptr(), R, W, b["SUB"], b["taxa"].data_ptr(),
                                       b["list_off"].data_ptr(), b["batch_base"].data_ptr(), out["taxon"].data_ptr(),
                                       out["classified"].data_ptr(), b["defer"].data_ptr(), out["num_distinct"].data_ptr(),
                                       out["total_kmers"].data_ptr(), out["num_hits"].data_ptr(), min_hit_groups=min_hit_groups,
                                       thresholds=thresholds, **b["mkw"])
            for k in ("taxa", "batch_base", "list_off", "counts", "recv_keys"):   # (their memory goes back to this stream's pool:
                b.pop(k, None)                                                     #  whatever reuses it is ordered after the apply)
        return out

    def _classify_fast(self, d_bases, d_offsets, R, total_bases, thresholds, min_hit_groups, mates=None):
        outs = self.classify_many([(d_bases, d_offsets, R, total_bases, mates)], thresholds, min_hit_groups)
        return None if outs is None else outs[0]

    def classify_many(self, batches, thresholds=(0.0,), min_hit_groups=2):
        """The fast route over several batches [(d_bases, d_offsets, R, total_bases, mates or None)], two in flight on two engine
        streams: while batch i is scanned (emit), batch i-1's keys are exchanged and looked up and its answers applied.  Every rank
        must pass the same number of batches.  Returns the list of result dicts, or None if the splitter only has the staged route."""
        states, outs = [], []

        def settle(j):   # batch j: its one host wait, then exchange + lookup + apply issued on its stream
            b, scale = states[j], 1
            while self._any_rank(not self._fast_ready(b)):      # (rare: a send list overflowed somewhere -- every rank emits again)
                scale *= 2
                d_bases, d_offsets, R, total_bases, mates = batches[j]
                b = states[j] = self._fast_emit(j % self.N_STREAMS, d_bases, d_offsets, R, total_bases, mates, scale)
            self._fast_exchange(b)
            outs.append(self._fast_apply(b, thresholds, min_hit_groups))

        for i, (d_bases, d_offsets, R, total_bases, mates) in enumerate(batches):
            b = self._fast_emit(i % self.N_STREAMS, d_bases, d_offsets, R, total_bases, mates)    # asynchronous
            if i == 0 and self._any_rank(b is None):
                return None
            states.append(b)
            if i >= self.N_STREAMS - 1:
                settle(i - (self.N_STREAMS - 1))     # (the emits of the batches after it are running meanwhile on the other streams)
        for j in range(max(0, len(states) - (self.N_STREAMS - 1)), len(states)):
            settle(j)
        for i, (b, batch) in enumerate(zip(states, batches)):
            b["st"].synchronize()
            d_bases, d_offsets, R, total_bases, mates = batch
            outs[i] = self._finish_deferred(outs[i], b, d_bases, d_offsets, R, thresholds, min_hit_groups, mates)
        return outs

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

"""Table-sharded classify (BASELINE.json configs[3]): the record table is too large for one GPU's HBM, so
rank g holds the records with fmix64(key) mod G == g.  Per batch every rank scans ITS reads, sends each minimizer to
its owner (all-to-all of 8-byte keys over RCCL/xGMI), owners look them up in their shard, the taxa return by the inverse
all-to-all (4 bytes each), and every rank finishes its own reads' LCA locally.  This replaces the reference's shuffle +
join (S/slacken/Classifier.scala:84) for the one case where data must move; with a table that fits, use the replicated
mode (no collective at all).

Host plumbing only: torch for device buffers and the collectives, the engine for all compute.  Two routes:
  fast    slk_shard_emit_device -> all-to-all -> slk_lookup_device -> all-to-all -> slk_shard_scatter_device ->
          slk_shard_apply_device: the fused lane-per-fragment kernel runs on both sides of the exchange, no span arrays;
          takes fragments of up to 1000 bases with at most 12 distinct taxa;
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

    def _classify_fast(self, d_bases, d_offsets, R, total_bases, thresholds, min_hit_groups, mates=None):
        import slacken_amd
        from slacken_amd import capi
        torch, dev, W = self.torch, self.device, self.world
        # about 0.26 probes per base on random sequence, spread evenly by the hash over W owners and by the wave index over
        # SUB sub-lists per owner (each fed by at least 64 waves, so that the spread holds): 0.6 / (W * SUB) per base leaves 2x
        # headroom; a list that overflows all the same makes the engine say so, and the batch is emitted again with twice the room
        mb, mo, mtotal = mates if mates is not None else (None, None, 0)
        mkw = dict(d_mate_bases=mb.data_ptr(), d_mate_offsets=mo.data_ptr()) if mates is not None else {}
        tiles = (R + 63) // 64
        SUB = 1
        while SUB < 256 and SUB * 2 * 64 <= tiles:
            SUB *= 2
        cap = int((total_bases + mtotal) * 0.6 / (W * SUB)) + (1 << 12)
        defer = torch.empty(max(R, 1), dtype=torch.int32, device=dev)
        unsupported = False
        while True:
            send_keys = torch.empty(W * SUB * cap, dtype=torch.int64, device=dev)
            send_slots = torch.empty(W * SUB * cap, dtype=torch.int64, device=dev)
            counts = torch.zeros(W * SUB, dtype=torch.int64, device=dev)
            try:
                self.st.shard_emit_device(d_bases.data_ptr(), d_offsets.data_ptr(), R, W, SUB, send_keys.data_ptr(),
                                          send_slots.data_ptr(), cap, counts.data_ptr(), defer.data_ptr(), **mkw)
                self.st.synchronize()
                break
            except slacken_amd.SlackenError as e:
                if e.code == capi.E_CAPACITY:
                    del send_keys, send_slots
                    cap *= 2
                    continue
                if e.code != capi.E_UNSUPPORTED:   # (unsupported: this splitter only has the staged route)
                    raise
                unsupported = True
                break
        if self._any_rank(unsupported):
            return None
        sub_counts = [int(v) for v in counts.tolist()]          # [owner][sub-list], owner-major: concatenation order
        send_counts = [sum(sub_counts[g * SUB:(g + 1) * SUB]) for g in range(W)]
        keys = torch.cat([send_keys[i * cap:i * cap + n] for i, n in enumerate(sub_counts)])
        slots = torch.cat([send_slots[i * cap:i * cap + n] for i, n in enumerate(sub_counts)])
        del send_keys, send_slots
        recv_keys, recv_counts = self._all_to_all(keys, send_counts)
        found = torch.zeros(max(recv_keys.numel(), 1), dtype=torch.int32, device=dev)
        if recv_keys.numel():
            recv_keys = recv_keys.contiguous()
            self.st.lookup_device(recv_keys.data_ptr(), recv_keys.numel(), found.data_ptr())
            self.st.synchronize()
        back, _ = self._all_to_all(found[:recv_keys.numel()].contiguous(), recv_counts)
        by_slot = torch.empty(total_bases + mtotal + R + 1, dtype=torch.int32, device=dev)
        back = back.contiguous()
        self.st.shard_scatter_device(slots.data_ptr(), back.data_ptr(), back.numel(), by_slot.data_ptr())
        C = len(thresholds)
        out = dict(taxon=torch.zeros(C * max(R, 1), dtype=torch.int32, device=dev),
                   classified=torch.zeros(C * max(R, 1), dtype=torch.uint8, device=dev),
                   num_distinct=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                   total_kmers=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                   num_hits=torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
                   exchanged_keys=int(keys.numel()))
        self.st.shard_apply_device(d_bases.data_ptr(), d_offsets.data_ptr(), R, by_slot.data_ptr(), out["taxon"].data_ptr(),
                                   out["classified"].data_ptr(), defer.data_ptr(), out["num_distinct"].data_ptr(),
                                   out["total_kmers"].data_ptr(), out["num_hits"].data_ptr(), min_hit_groups=min_hit_groups,
                                   thresholds=thresholds, **mkw)
        self.st.synchronize()
        # fragments the fused kernel does not take: a compacted batch through the staged route (all ranks, also with none)
        idx = torch.nonzero(defer[:R]).flatten() if R else torch.zeros(0, dtype=torch.int64, device=dev)
        out["deferred"] = int(idx.numel())
        if self._any_rank(idx.numel() > 0):
            def compact(bases, offsets):
                lens = offsets[idx + 1] - offsets[idx]
                sub_off = torch.zeros(idx.numel() + 1, dtype=torch.int64, device=dev)
                sub_off[1:] = torch.cumsum(lens, 0)
                sub_total = int(sub_off[-1].item()) if idx.numel() else 0
                src = torch.repeat_interleave(offsets[idx], lens) + (torch.arange(sub_total, device=dev) -
                                                                     torch.repeat_interleave(sub_off[:-1], lens))
                return torch.cat([bases[src], torch.zeros(64, dtype=torch.uint8, device=dev)]), sub_off, sub_total
            sub_bases, sub_off, sub_total = compact(d_bases, d_offsets)
            sub_mates = None
            if mates is not None:
                smb, smo, smt = compact(mb, mo)
                sub_mates = (smb, smo, smt)
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
        self.st.classify_hits_device(d_offsets.data_ptr(), R, meta.data_ptr(), taxon.data_ptr(), count.data_ptr(),
                                     scratch.data_ptr(), out["taxon"].data_ptr(), out["classified"].data_ptr(),
                                     out["num_distinct"].data_ptr(), out["total_kmers"].data_ptr(),
                                     out["num_hits"].data_ptr(), min_hit_groups=min_hit_groups, thresholds=thresholds,
                                     **({"d_mate_offsets": mo.data_ptr()} if mates is not None else {}))
        self.st.synchronize()
        return out

#!/usr/bin/env python3
"""Hot-path throughput on a library with phylogenetic structure: genomes evolve along the taxonomy (each child = its parent with
a few per cent substitutions), so minimizers carry LCA taxa at every rank and a read hits several distinct taxa -- the per-read
resolveTree really walks the tree, unlike on bench.py's independent random genomes (one taxon per read).  GPU box."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import slacken_amd
    rng = np.random.default_rng(11)
    # a regular tree: 8 ranks, fan-out 2 below the root's 2 children => 256 leaves
    # CHAIN=n: n unary nodes above every branching node (NCBI lineages are 25-40 nodes deep, most of them without siblings that
    # matter to a given read): the tree walks get longer, the LCA taxa of the records stay at the branching nodes
    chain = int(os.environ.get("CHAIN", 0))
    parents = [0, 0]
    level = [1]
    evolves = {1}
    for d in range(8):
        nxt = []
        for p in level:
            for _ in range(2):
                up = p
                for _ in range(chain):
                    parents.append(up)
                    up = len(parents) - 1
                parents.append(up)
                nxt.append(len(parents) - 1)
                evolves.add(len(parents) - 1)
        level = nxt
    parents = np.array(parents, np.int32)
    leaves = level
    L = 1 << 18
    acgt = np.frombuffer(b"ACGT", np.uint8)
    genome = {1: acgt[rng.integers(0, 4, L, dtype=np.uint8)]}
    order = [t for t in range(2, len(parents))]
    rate = float(os.environ.get("RATE", 0.02))
    for t in order:                               # parents precede children by construction
        if t not in evolves:                      # a unary node of a chain: its parent's genome
            genome[t] = genome[int(parents[t])]
            continue
        g = genome[int(parents[t])].copy()
        sub = rng.random(L) < rate
        g[sub] = acgt[rng.integers(0, 4, int(sub.sum()), dtype=np.uint8)]
        genome[t] = g
    bases = np.concatenate([genome[t] for t in leaves])
    offsets = np.arange(len(leaves) + 1, dtype=np.uint64) * np.uint64(L)
    pad = int(float(os.environ.get("PAD", 0)))     # PAD=1e10: random records up to the standard-library scale (128 GiB table)
    ix = slacken_amd.Index(expected_records=max(int(len(leaves) * L * 0.4), pad), max_taxon=len(parents) - 1)
    ix.set_taxonomy(parents)
    ix.add_sequences(bases, offsets, np.array(leaves, np.int32))
    tmp = slacken_amd.Index(expected_records=int(len(leaves) * L * 0.4), max_taxon=len(parents) - 1)
    tmp.set_taxonomy(parents)
    tmp.add_sequences(bases, offsets, np.array(leaves, np.int32))
    _, taxa = tmp.export()
    tmp.close()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    smask = (((2**62 - 1) & ~0x0CCCCCCC) << 2) - (1 << 64)
    for s0 in range(0, max(0, pad - len(taxa)), 1 << 27):
        n = min(1 << 27, pad - len(taxa) - s0)
        keys = ((torch.randint(0, 2**32, (n,), generator=gen, device="cuda", dtype=torch.int64) << 32) |
                torch.randint(0, 2**32, (n,), generator=gen, device="cuda", dtype=torch.int64)) & smask
        tx = torch.randint(2, len(parents), (n,), generator=gen, device="cuda", dtype=torch.int32)
        torch.cuda.synchronize()
        ix.append_device(keys.data_ptr(), tx.data_ptr(), n)
        del keys, tx
    ix.finalize()
    depth = np.zeros(len(parents), np.int32)
    for t in range(2, len(parents)):
        depth[t] = depth[parents[t]] + 1
    hist = np.bincount(depth[taxa] // (chain + 1), minlength=9)
    R = 4_000_000
    d_all = torch.from_numpy(bases).cuda()
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    stt = torch.randint(0, len(bases) - 150, (R,), generator=g, device="cuda")
    d_b = torch.cat([d_all[(stt[:, None] + torch.arange(150, device="cuda")[None, :]).reshape(-1)], torch.zeros(64, dtype=torch.uint8, device="cuda")])
    d_o = torch.arange(0, (R + 1) * 150, 150, dtype=torch.int64, device="cuda")
    d_t = torch.zeros(2 * R, dtype=torch.int32, device="cuda")
    d_c = torch.zeros(2 * R, dtype=torch.uint8, device="cuda")
    d_nd = torch.zeros(R, dtype=torch.int32, device="cuda")
    st = ix.stream()
    out = {}
    for thr in ((0.0,), (0.0, 0.15)):
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st.classify_batch_device(d_b.data_ptr(), d_o.data_ptr(), R, R * 150, d_t.data_ptr(), d_c.data_ptr(), d_nd.data_ptr(), thresholds=thr)
            st.synchronize()
            dt = time.perf_counter() - t0
        out[f"thresholds_{len(thr)}"] = dict(ms=round(dt * 1e3, 2), M_reads_per_s=round(R / dt / 1e6, 1), deferred=st.last_deferred())
    lvl = depth[d_t[:R].cpu().numpy()] // (chain + 1)
    # long reads (the wave-per-read kernel: 128-slot map, resolveTree with one lane per distinct taxon)
    for L_read, R2 in ((1001, 1_000_000), (10_000, 100_000)):
        stt2 = torch.randint(0, len(bases) - L_read, (R2,), generator=g, device="cuda")
        b2 = torch.cat([d_all[(stt2[:, None] + torch.arange(L_read, device="cuda")[None, :]).reshape(-1)], torch.zeros(64, dtype=torch.uint8, device="cuda")])
        o2 = torch.arange(0, (R2 + 1) * L_read, L_read, dtype=torch.int64, device="cuda")
        for thr in ((0.0,), (0.0, 0.15)):
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st.classify_batch_device(b2.data_ptr(), o2.data_ptr(), R2, R2 * L_read, d_t.data_ptr(), d_c.data_ptr(), d_nd.data_ptr(), thresholds=thr)
                st.synchronize()
                dt = time.perf_counter() - t0
            out[f"reads_{L_read}bp" + ("" if len(thr) == 1 else "_two_thresholds")] = dict(ms=round(dt * 1e3, 2), Gbp_per_s=round(R2 * L_read / dt / 1e9, 1))
        del b2, o2
    out.update(records=int(ix.info().records), records_by_depth=hist.tolist(), classified=float(d_c[:R].float().mean()),
               calls_by_depth=np.bincount(lvl, minlength=9).tolist(), substitution_rate_per_level=rate, chain=chain,
               tree_depth=int(depth.max()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()

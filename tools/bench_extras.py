#!/usr/bin/env python3
"""Secondary measurements (not the bench.py contract line): library construction on the device (config 5's builder) and the
host-pointer classify entry (PCIe-inclusive).  Prints one JSON object; run on the GPU box."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401  (before the engine's library: one HIP runtime per process)
    import slacken_amd
    import taxgen
    rng = np.random.default_rng(5)
    parents = taxgen.taxonomy(8 * 1024, rng)
    taxa = np.array(taxgen.defined_taxa(parents))
    out = {}
    # ---- library construction: G genomes of L bases, host -> device -> table ----
    G, L = int(os.environ.get("G", 256)), int(os.environ.get("L", 4 << 20))
    acgt = np.frombuffer(b"ACGT", np.uint8)
    bases = acgt[rng.integers(0, 4, G * L, dtype=np.uint8)]
    offsets = (np.arange(G + 1, dtype=np.uint64) * np.uint64(L))
    gt = rng.choice(taxa[len(taxa) // 2:], G).astype(np.int32)
    ix = slacken_amd.Index(expected_records=int(G * L * 0.4), max_taxon=len(parents) - 1)
    ix.set_taxonomy(parents)
    ix.add_sequences(bases[:L], offsets[:2], gt[:1])          # warm-up (allocations, code load)
    t0 = time.perf_counter()
    ix.add_sequences(bases, offsets, gt)
    dt = time.perf_counter() - t0
    info = ix.info()
    out["build"] = dict(bases=G * L, seconds=round(dt, 4), Gbp_per_s=round(G * L / dt / 1e9, 3), records=int(info.records),
                        table_GiB=info.table_bytes / 2**30, max_displacement=info.max_displacement,
                        note="slk_index_add_sequences from host memory (includes the H2D copy of the bases)")
    t0 = time.perf_counter()
    ix.add_sequences(bases, offsets, gt)                       # same sequences again: every insert is a merge
    out["build"]["seconds_all_merges"] = round(time.perf_counter() - t0, 4)
    ix.finalize()
    # ---- host-pointer classify entry: reads in host memory, results back in host memory ----
    R = 4_000_000
    starts = rng.integers(0, G * L - 150, R)
    idx = (starts[:, None] + np.arange(150)[None, :]).reshape(-1)
    rb = bases[idx]
    ro = (np.arange(R + 1, dtype=np.uint64) * np.uint64(150))
    st = ix.stream()
    st.classify_batch(rb[:150 * 1000], ro[:1001], with_hits=False)
    t0 = time.perf_counter()
    res = st.classify_batch(rb, ro, with_hits=False)
    dt = time.perf_counter() - t0
    out["host_entry"] = dict(reads=R, seconds=round(dt, 4), M_reads_per_s=round(R / dt / 1e6, 1),
                             classified=float(res["classified"][0].mean()),
                             note="slk_classify_batch with pageable host buffers: H2D of 150 B/read + kernels + D2H, one call")
    # ---- long reads (device entry): every fragment is longer than the lane kernel takes, so the wave-per-read kernel runs ----
    import torch
    for L_read, R in ((10_000, 100_000), (1000, 1_000_000), (1001, 1_000_000)):
        starts = rng.integers(0, G * L - L_read, R)
        d_all = torch.from_numpy(bases).cuda()
        idx = torch.from_numpy(starts).cuda()[:, None] + torch.arange(L_read, device="cuda")[None, :]
        d_b = torch.cat([d_all[idx.reshape(-1)], torch.zeros(64, dtype=torch.uint8, device="cuda")])
        del idx
        d_o = torch.arange(0, (R + 1) * L_read, L_read, dtype=torch.int64, device="cuda")
        d_t = torch.zeros(R, dtype=torch.int32, device="cuda")
        d_c = torch.zeros(R, dtype=torch.uint8, device="cuda")
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st.classify_batch_device(d_b.data_ptr(), d_o.data_ptr(), R, R * L_read, d_t.data_ptr(), d_c.data_ptr())
            st.synchronize()
            dt = time.perf_counter() - t0
        out[f"reads_{L_read}bp"] = dict(reads=R, ms=round(dt * 1e3, 2), Gbp_per_s=round(R * L_read / dt / 1e9, 1),
                                        classified=float(d_c.float().mean().item()))
        del d_b, d_o, d_all
    # ---- paired 2 x 150 and other read lengths on the hot path (device entry) ----
    d_all = torch.from_numpy(bases).cuda()

    def make(R, L_read, seed):
        g = torch.Generator(device="cuda")
        g.manual_seed(seed)
        stt = torch.randint(0, G * L - L_read, (R,), generator=g, device="cuda")
        idx = stt[:, None] + torch.arange(L_read, device="cuda")[None, :]
        b = torch.cat([d_all[idx.reshape(-1)], torch.zeros(64, dtype=torch.uint8, device="cuda")])
        o = torch.arange(0, (R + 1) * L_read, L_read, dtype=torch.int64, device="cuda")
        return b, o

    R = 4_000_000
    d_t = torch.zeros(R, dtype=torch.int32, device="cuda")
    d_c = torch.zeros(R, dtype=torch.uint8, device="cuda")
    for name, L1, L2 in (("single_100", 100, 0), ("single_150", 150, 0), ("single_250", 250, 0), ("paired_2x150", 150, 150), ("paired_2x100", 100, 100)):
        b1, o1 = make(R, L1, 1)
        kw = {}
        if L2:
            b2, o2 = make(R, L2, 2)
            kw = dict(d_mate_bases=b2.data_ptr(), d_mate_offsets=o2.data_ptr(), total_mate_bases=R * L2)
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st.classify_batch_device(b1.data_ptr(), o1.data_ptr(), R, R * L1, d_t.data_ptr(), d_c.data_ptr(), **kw)
            st.synchronize()
            dt = time.perf_counter() - t0
        out[name] = dict(fragments=R, ms=round(dt * 1e3, 2), M_fragments_per_s=round(R / dt / 1e6, 1),
                         Gbp_per_s=round(R * (L1 + L2) / dt / 1e9, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()

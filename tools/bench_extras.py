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
    from slacken_amd import capi
    import threading

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best

    res = {}
    dt = timed(lambda: res.update(st.classify_batch(rb, ro, with_hits=False)))
    out["host_entry"] = dict(reads=R, seconds=round(dt, 4), M_reads_per_s=round(R / dt / 1e6, 1),
                             classified=float(res["classified"][0].mean()),
                             note="slk_classify_batch with pageable host buffers: H2D of 150 B/read + kernels + D2H, one call "
                                  "(best of 3); the upload is staged by the library's copy threads and pipelined with the kernels")
    prb = capi.pinned_array(rb.shape, np.uint8); prb[:] = rb
    pro = capi.pinned_array(ro.shape, np.uint64); pro[:] = ro
    pout = dict(taxon=capi.pinned_array((1, R), np.int32), classified=capi.pinned_array((1, R), np.uint8),
                num_distinct=capi.pinned_array((R,), np.int32), total_kmers=capi.pinned_array((R,), np.int32))
    dt = timed(lambda: st.classify_batch(prb, pro, with_hits=False, out=pout))
    assert np.array_equal(pout["taxon"], res["taxon"])
    out["host_entry_pinned"] = dict(reads=R, seconds=round(dt, 4), M_reads_per_s=round(R / dt / 1e6, 1),
                                    GB_per_s_up=round((rb.nbytes + ro.nbytes) / dt / 1e9, 1),
                                    note="the same call with input and output buffers from slk_host_alloc: direct DMA, one call (best of 3)")
    # the same reads in the engine's 3-bit form (slk_classify_batch_packed): 6 bytes per 16 bases over the link
    t0 = time.perf_counter()
    pk_codes, pk_valid = capi.pack_bases(rb, pinned=True)
    out["pack_bases"] = dict(bases=int(rb.size), seconds=round(time.perf_counter() - t0, 4),
                             note="slk_pack_bases (AVX2 + BMI2, the library's copy threads) incl. first touch of the pinned output")
    t0 = time.perf_counter()
    capi.lib().slk_pack_bases(rb.ctypes.data, rb.size, pk_codes.ctypes.data, pk_valid.ctypes.data)
    dtp = time.perf_counter() - t0
    out["pack_bases"].update(seconds_warm=round(dtp, 4), GB_per_s=round(rb.size / dtp / 1e9, 1))
    dt = timed(lambda: st.classify_batch(None, pro, with_hits=False, out=pout, packed=(pk_codes, pk_valid)))
    assert np.array_equal(pout["taxon"], res["taxon"])
    out["host_entry_packed_pinned"] = dict(reads=R, seconds=round(dt, 4), M_reads_per_s=round(R / dt / 1e6, 1),
                                           GB_per_s_up=round((pk_codes.nbytes + pk_valid.nbytes + ro.nbytes) / dt / 1e9, 1),
                                           bytes_per_read_up=round((pk_codes.nbytes + pk_valid.nbytes + ro.nbytes) / R, 1),
                                           note="slk_classify_batch_packed, reads packed beforehand into buffers of slk_host_alloc: one call (best of 3)")
    dt = timed(lambda: st.classify_batch(None, ro, with_hits=False, packed=(np.array(pk_codes), np.array(pk_valid))))
    out["host_entry_packed_pageable"] = dict(reads=R, seconds=round(dt, 4), M_reads_per_s=round(R / dt / 1e6, 1),
                                             note="the same from pageable buffers (staged by the library's copy threads)")
    # several caller threads, one stream each (the intended use: one slk_stream per Spark task thread)
    for nthreads, pinned_bufs in ((3, False), (3, True)):
        streams = [ix.stream() for _ in range(nthreads)]
        bufs = []
        for _ in range(nthreads):
            if pinned_bufs:
                o = dict(taxon=capi.pinned_array((1, R), np.int32), classified=capi.pinned_array((1, R), np.uint8),
                         num_distinct=capi.pinned_array((R,), np.int32), total_kmers=capi.pinned_array((R,), np.int32))
            else:
                o = None
            bufs.append(o)
        calls = 3

        def worker(i):
            for _ in range(calls):
                streams[i].classify_batch(prb if pinned_bufs else rb, pro if pinned_bufs else ro, with_hits=False, out=bufs[i])

        for i in range(nthreads):
            worker(i) if False else None
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.perf_counter() - t0
        out[f"host_entry_{nthreads}threads_{'pinned' if pinned_bufs else 'pageable'}"] = dict(
            reads=R * calls * nthreads, seconds=round(dt, 4), M_reads_per_s=round(R * calls * nthreads / dt / 1e6, 1),
            note=f"{nthreads} threads x {calls} calls of {R} reads, a stream each")
        del streams
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

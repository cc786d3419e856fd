#!/usr/bin/env python3
"""bench.py -- classify throughput of the MI355X engine on BASELINE.json's headline configuration.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (configs[1] of BASELINE.json; synthetic because standard-224 is an S3 download):
  * "standard-224-scale" library: --records (default 1.0e10) unique (minimizer, taxon) records, k=35 m=31 s=7:
    the minimizers of G synthetic genomes (found with the engine's own scan kernel and merged by LCA) padded with
    uniformly random 48-significant-bit keys; taxonomy 8 ranks x 8192 nodes relabelled onto ids < 3 080 008.
  * reads: --reads (default 1.0e7) single-end 150 bp per GPU, 80 % drawn from the genomes (random strand, 1 %
    substitutions), 20 % uniform random, 0.5 % with one N, 0.05 % with a 40-N run.  Resident in HBM before timing.
  * a step = one pass of scan -> probe -> classify over the whole read batch (slk_classify_batch_device).
Multi-GPU: table replicated, reads sharded (each rank its own batch), no data-path collective => "weak" scaling.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the table probe); `cpu_baseline` times the CPU
restatement under oracle/ (OpenMP, all host cores) on a bounded sample -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GATHER_CEILING_GLPS = 48.5
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
K, M, SPACES = 35, 31, 7
READ_LEN = 150
TAX_EXTENT = 3080008  # README.md:374 of the reference (NCBI taxonomy array extent)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def build_taxonomy(seed=2240):
    import taxgen
    rng = np.random.default_rng(seed)
    small = taxgen.taxonomy(8 * 8192, rng)
    parents, _ = taxgen.sparse_relabel(small, TAX_EXTENT, rng)
    taxa = np.nonzero(parents)[0].astype(np.int32)
    is_parent = np.zeros(len(parents), bool)
    is_parent[parents[taxa]] = True
    leaves = taxa[~is_parent[taxa]]
    return parents, taxa, leaves


def genome_records(slacken_amd, genomes, genome_taxa, parents, device):
    """(key, LCA taxon) records of the genomes, built on the device by the config-5 builder (slk_index_add_sequences:
    minimizers of every genome, duplicates across genomes merged by LCA -- KeyValueIndex.makeRecords, KeyValueIndex.scala:85-93)
    and exported sorted by key."""
    lens = np.array([len(g) for g in genomes], np.uint64)
    offsets = np.zeros(len(genomes) + 1, np.uint64)
    np.cumsum(lens, out=offsets[1:])
    tmp = slacken_amd.Index(k=K, m=M, spaces=SPACES, expected_records=int(offsets[-1]) // 2 + 1024, max_taxon=TAX_EXTENT - 1,
                            device=device)
    tmp.set_taxonomy(parents)
    tmp.add_sequences(np.concatenate(genomes), offsets, genome_taxa.astype(np.int32))
    keys, tax = tmp.export()
    tmp.close()
    return keys, tax


def make_reads_device(torch, genome_cat, genome_len, n_genomes, n_reads, seed, device):
    """uint8 tensor [n_reads * READ_LEN] on the GPU + uint64-compatible offsets (int64 tensor)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    flat = torch.full((n_reads * READ_LEN + 64,), ord("A"), dtype=torch.uint8, device=device)  # 16+ readable pad bytes
    out = flat[:n_reads * READ_LEN].view(n_reads, READ_LEN)
    ar = torch.arange(READ_LEN, device=device)
    CH = 1 << 20
    for s in range(0, n_reads, CH):
        n = min(CH, n_reads - s)
        gi = torch.randint(0, n_genomes, (n,), generator=g, device=device)
        st = torch.randint(0, genome_len - READ_LEN + 1, (n,), generator=g, device=device)
        idx = (gi * genome_len + st)[:, None] + ar[None, :]
        r = genome_cat[idx]
        flip = torch.rand(n, generator=g, device=device) < 0.5
        rc = comp[r.long()].flip(1)
        r = torch.where(flip[:, None], rc, r)
        subs = torch.rand((n, READ_LEN), generator=g, device=device) < 0.01
        rnd = acgt[torch.randint(0, 4, (n, READ_LEN), generator=g, device=device)]
        r = torch.where(subs, rnd, r)
        is_random = torch.rand(n, generator=g, device=device) < 0.2
        r = torch.where(is_random[:, None], rnd, r)
        one_n = torch.rand(n, generator=g, device=device) < 0.005
        pos = torch.randint(0, READ_LEN, (n,), generator=g, device=device)
        r = torch.where(one_n[:, None] & (ar[None, :] == pos[:, None]), torch.full_like(r, ord("N")), r)
        run_n = torch.rand(n, generator=g, device=device) < 0.0005
        pos = torch.randint(0, READ_LEN - 40, (n,), generator=g, device=device)
        in_run = (ar[None, :] >= pos[:, None]) & (ar[None, :] < pos[:, None] + 40)
        r = torch.where(run_n[:, None] & in_run, torch.full_like(r, ord("N")), r)
        out[s:s + n] = r
    offsets = torch.arange(0, (n_reads + 1) * READ_LEN, READ_LEN, dtype=torch.int64, device=device)
    return flat, offsets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--records", type=float, default=1.0e10, help="library records (standard-224-scale)")
    ap.add_argument("--reads", type=float, default=1.0e7, help="150 bp reads per GPU per step")
    ap.add_argument("--genomes", type=int, default=2048)
    ap.add_argument("--genome-len", type=int, default=16384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from slacken_amd import dist as sdist
    rank, world, local_rank = sdist.env_rank_world()
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    import slacken_amd
    n_records, n_reads = int(args.records), int(args.reads)
    t0 = time.time()
    parents, taxa, leaves = build_taxonomy()
    rng = np.random.default_rng(224)
    G = min(args.genomes, len(leaves))
    genome_taxa = rng.choice(leaves, size=G, replace=False)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    genomes = [acgt[rng.integers(0, 4, args.genome_len)] for _ in range(G)]
    gkeys, gtax = genome_records(slacken_amd, genomes, genome_taxa, parents, local_rank)
    log(f"rank {rank}: taxonomy {len(taxa) + 1} nodes, {G} genomes -> {len(gkeys)} records ({time.time() - t0:.1f}s)")

    # ---- HBM-resident table: genome records + random padding generated on the device
    ix = slacken_amd.Index(k=K, m=M, spaces=SPACES, expected_records=max(n_records, len(gkeys)),
                           max_taxon=TAX_EXTENT - 1, device=local_rank)
    ix.append(gkeys, gtax)
    smask = ((2**62 - 1) & ~0x0CCCCCCC) << 2  # SpacedSeed mask for m=31, s=7 (48 significant bits), left-aligned
    smask_i64 = smask - (1 << 64) if smask >= (1 << 63) else smask
    d_taxa = torch.from_numpy(taxa).to(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(224 + 7)
    pad, CH = max(0, n_records - len(gkeys)), 1 << 27
    for s in range(0, pad, CH):
        n = min(CH, pad - s)
        hi = torch.randint(0, 2**32, (n,), generator=gen, device=device, dtype=torch.int64)
        lo = torch.randint(0, 2**32, (n,), generator=gen, device=device, dtype=torch.int64)
        keys = ((hi << 32) | lo) & smask_i64
        del hi, lo
        tx = d_taxa[torch.randint(0, len(taxa), (n,), generator=gen, device=device)]
        torch.cuda.synchronize()
        ix.append_device(keys.data_ptr(), tx.data_ptr(), n)
        del keys, tx
    ix.set_taxonomy(parents)
    ix.finalize()
    info = ix.info()
    torch.cuda.empty_cache()
    log(f"rank {rank}: table {info.records} records in {info.table_bytes / 2**30:.1f} GiB "
        f"(2^{info.bucket_bits} buckets, load {info.records / (info.buckets * 8):.2f}, max displacement "
        f"{info.max_displacement}, {info.duplicate_keys} duplicate pad keys dropped) ({time.time() - t0:.1f}s)")

    # ---- reads resident in HBM
    genome_cat = torch.from_numpy(np.concatenate(genomes)).to(device)
    d_bases, d_offsets = make_reads_device(torch, genome_cat, args.genome_len, G, n_reads, 150 + rank, device)
    total_bases = n_reads * READ_LEN
    d_taxon = torch.zeros(n_reads, dtype=torch.int32, device=device)
    d_cls = torch.zeros(n_reads, dtype=torch.uint8, device=device)
    d_nd = torch.zeros(n_reads, dtype=torch.int32, device=device)
    d_tk = torch.zeros(n_reads, dtype=torch.int32, device=device)
    d_nh = torch.zeros(n_reads, dtype=torch.int32, device=device)
    d_np = torch.zeros(n_reads, dtype=torch.int32, device=device)
    st = ix.stream()
    torch.cuda.synchronize()
    log(f"rank {rank}: {n_reads} reads resident ({time.time() - t0:.1f}s)")

    def step():
        st.classify_batch_device(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, total_bases, d_taxon.data_ptr(),
                                 d_cls.data_ptr(), d_nd.data_ptr(), d_tk.data_ptr(), d_nh.data_ptr(), d_np.data_ptr(),
                                 min_hit_groups=2, thresholds=(0.0,))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    stage_ms = np.zeros(3)
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
        # HIP events on the engine's own stream bracket each stage kernel (read after the timed region)
    barrier()
    elapsed = time.perf_counter() - t_start
    # per-stage device time: re-read the events of the LAST timed step (every step launches the same three kernels)
    stage_ms = np.array(st.last_stage_ms())
    elapsed = sdist.max_over_ranks(elapsed, dist, device)
    ms_per_step = elapsed / args.steps * 1e3
    reads_per_s = world * n_reads / (elapsed / args.steps)

    # ---- algorithmic bytes: B(r) = L_r + 64 * P_r + 8
    probes = int(d_np.sum().item())
    classified = float(d_cls.float().mean().item())
    bytes_per_launch = total_bases + 64 * probes + 8 * n_reads
    fused = float(stage_ms[1]) < 0.05 and float(stage_ms[2]) < 0.05  # one fused launch: [fused, 0, ~0]
    dom_ms = float(stage_ms[0]) if fused else float(stage_ms[1])
    dom_name = ("slk::lane_kernel<true> (scan+probe+LCA fused, lane per read; followed by the near-empty deferral "
                "launches of segment_kernel and fused_kernel<1>)") if fused else "slk::probe_kernel"
    achieved = bytes_per_launch / (dom_ms * 1e-3) / 1e9
    path_achieved = bytes_per_launch / (float(stage_ms.sum()) * 1e-3) / 1e9

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if os.path.exists(tpath):  # FETCH_SIZE + WRITE_SIZE of the dominant kernel, rocprofv3 --pmc passes of this same command
        tj = json.load(open(tpath))
        if tj.get("reads_per_launch") == n_reads and tj.get("records") == int(args.records):
            traffic = tj["hbm_bytes_per_launch"]

    out = {
        "metric": "classify_throughput_150bp_standard224scale",
        "value": round(reads_per_s / 1e6, 3),
        "unit": "M reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": "standard-224-scale synthetic library (k=35,m=31,s=7), synthetic 150 bp single-end reads, "
                        "full table resident in HBM (BASELINE.json configs[1])",
            "records": int(info.records), "table_GiB": round(info.table_bytes / 2**30, 1),
            "reads_per_gpu_per_step": n_reads, "read_len": READ_LEN, "parallelism": f"read-sharded x{world}, table replicated",
            "probes_per_read": round(probes / n_reads, 3), "classified_fraction": round(classified, 4),
            "deferred_to_wave_kernel": st.last_deferred(),
            "stage_ms": ({"fused": round(dom_ms, 3)} if fused else
                         {"scan": round(float(stage_ms[0]), 3), "probe": round(float(stage_ms[1]), 3),
                          "classify": round(float(stage_ms[2]), 3)}),
            "path_GBps_all_kernels": round(path_achieved, 1),
            "path_frac_all_kernels": round(path_achieved / HBM_PEAK_GBPS, 4),
        },
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": bytes_per_launch, "kernel_ms": round(dom_ms, 3),
                     # the part's measured rate of random 64-byte line reads on a 128 GiB table (tools/gather_bench.hip,
                     # profiles/r01_gather_microbench.txt): what a hash-table probe can reach, as opposed to the streaming peak
                     "random_line_ceiling_Glines_per_s": GATHER_CEILING_GLPS,
                     "probe_lines_per_s_G": round(probes / (dom_ms * 1e-3) / 1e9, 2),
                     "frac_of_random_line_ceiling": round(probes / (dom_ms * 1e-3) / 1e9 / GATHER_CEILING_GLPS, 3)},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, gkeys, gtax, parents, d_bases, n_reads, rng)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(args, gkeys, gtax, parents, d_bases, n_reads, rng):
    """The CPU restatement (oracle/, OpenMP over reads) on a bounded sample of the same reads (sized for roughly 12 s),
    against the genome records plus random padding up to 2^24 records (hits come from the genome records; the padding
    keeps the hash table far larger than the CPU caches, as the full library would)."""
    ncpu = len(os.sched_getaffinity(0))
    os.environ.setdefault("OMP_NUM_THREADS", str(ncpu))
    from oracle import oracle
    p = oracle.params(k=K, m=M, spaces=SPACES)
    pad = max(0, (1 << 24) - len(gkeys))
    pkeys = (rng.integers(0, 2**63, pad, dtype=np.uint64) * np.uint64(2)) & np.uint64(p.space[0])
    ptax = np.full(pad, 1, np.int32)
    oix = oracle.Index(1, np.concatenate([gkeys, pkeys.view(np.int64)]), np.concatenate([gtax, ptax]))
    probe_n = min(n_reads, 200000)
    bases = d_bases[:probe_n * READ_LEN].cpu().numpy()
    offsets = np.arange(0, (probe_n + 1) * READ_LEN, READ_LEN, dtype=np.uint64)
    t = time.perf_counter()
    res = oracle.classify_batch(p, oix, parents, bases, offsets)  # also warms the table
    rate = probe_n / (time.perf_counter() - t)
    S = int(min(n_reads, max(probe_n, rate * 12.0)))
    bases = d_bases[:S * READ_LEN].cpu().numpy()
    offsets = np.arange(0, (S + 1) * READ_LEN, READ_LEN, dtype=np.uint64)
    passes, dt = 0, 0.0
    while dt < 10.0 and passes < 8:  # about 10-30 s of wall time on the host cores
        t = time.perf_counter()
        res = oracle.classify_batch(p, oix, parents, bases, offsets)
        dt += time.perf_counter() - t
        passes += 1
    S *= passes
    return {"value": round(S / dt / 1e6, 4), "unit": "M reads/s", "cores": int(res["threads"]), "kind": "port",
            "sample": f"{passes} pass(es) over the first {S // passes} of the step's reads, CPU restatement (oracle/, OpenMP, {res['threads']} threads on "
                      f"{ncpu} usable CPUs) vs {len(gkeys)} genome records + random padding to 2^24 records in a DRAM "
                      f"hash table; {dt:.1f} s; the reference's own Spark path cannot run here (no JVM)"}


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- classify throughput of the MI355X engine on BASELINE.json's headline configuration.

  python bench.py --gpus N --steps K --warmup W

N > 1: if the process was not started by torch.distributed.run (no WORLD_SIZE in the environment) it starts the N ranks
itself -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>`
as a CHILD process, before anything in this process has touched the GPU -- and exits with the child's code.  Started by
torch.distributed.run (the driver's way) it is one rank of the job.  A WORLD_SIZE that differs from --gpus is an error.

Workload (configs[1] of BASELINE.json; synthetic because standard-224 is an S3 download):
  * "standard-224-scale" library: --records (default 1.0e10) unique (minimizer, taxon) records, k=35 m=31 s=7:
    the minimizers of G synthetic genomes (--genomes x --genome-len; default 8192 x 1 Mbp, one per leaf taxon, as SURVEY
    8d specifies), found and LCA-merged by the engine's own library builder straight into the table
    (slk_index_add_sequences_device), padded with uniformly random 48-significant-bit keys; taxonomy 8 ranks x 8192 nodes
    relabelled onto ids < 3 080 008.
  * reads: --reads (default 1.0e7) single-end 150 bp per GPU, 80 % drawn from the genomes (random strand, 1 %
    substitutions), 20 % uniform random, 0.5 % with one N, 0.05 % with a 40-N run.  Resident in HBM before timing.
  * a step = one pass of scan -> probe -> classify over the whole read batch (slk_classify_batch_device).
Multi-GPU: table replicated, reads sharded (each rank its own batch), no data-path collective => "weak" scaling.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fused lane kernel, bound by its table probes);
`cpu_baseline` times the CPU restatement under oracle/ (OpenMP, all host cores) on a bounded sample -- a reported
baseline, not the target.

--table-sharded (BASELINE.json configs[3]): the library is larger than one GPU's HBM, so rank g holds only the records whose key
falls to it (fmix64(key) mod N; slk_index_set_shard) -- by default 5.0e9 records per rank, 4.0e10 at N = 8 (> 288 GB of table) --
and a step is one 10 M-read batch per rank through scan -> keys to their owners (all-to-all over RCCL / xGMI) -> lookup -> taxa
back -> per-read LCA (slacken_amd/sharded.py).  The K timed steps are K batches in the pipeline (K + 4 kernel launches).  The line carries the
stages' device times, the bytes exchanged per read, the per-link rate against the 153 GB/s xGMI link and the lookup stage's share
of the part's random-request rate.

--dry-run: no GPU, no engine: every rank times a trivial numpy step and goes through the same rendezvous (gloo), barrier,
max-over-ranks and reporting code, so that the N-rank plumbing can be tested on a CPU box.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# random 64-byte requests/s this part sustains against a 128 GiB table (tools/gather_bench2.hip, profiles/r02_gather_experiments.txt):
# the FALLBACK only -- a run measures the rate itself after its timed region (measure_request_ceiling) and says which it quotes
GATHER_CEILING_GLPS = 48.4
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
K, M, SPACES = 35, 31, 7
READ_LEN = 150
TAX_EXTENT = 3080008  # README.md:374 of the reference (NCBI taxonomy array extent)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", os.environ.get("SLK_TRAFFIC_FILE", "r04_traffic.json"))


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)
    progress(" ".join(str(x) for x in a)[:120])


# ---- watchdog: a rank that makes no progress for WATCHDOG_S seconds says where it is and exits non-zero (never a re-exec) ----
WATCHDOG_S = float(os.environ.get("SLK_BENCH_WATCHDOG_S", "300"))
_progress = {"stage": "start", "t": time.monotonic(), "armed": False}


def progress(stage):
    """called at every milestone (every log line is one): the stage a hung rank would be reported in"""
    _progress["stage"], _progress["t"] = stage, time.monotonic()


def arm_watchdog(rank):
    import threading
    if _progress["armed"] or WATCHDOG_S <= 0:
        return
    _progress["armed"] = True

    def watch():
        while True:
            time.sleep(min(5.0, max(0.05, WATCHDOG_S / 4)))
            idle = time.monotonic() - _progress["t"]
            if idle > WATCHDOG_S:
                print(f"[bench] rank {rank}: WATCHDOG: no progress for {idle:.0f} s in stage: {_progress['stage']} -- giving up",
                      file=sys.stderr, flush=True)
                os._exit(3)

    threading.Thread(target=watch, name="bench-watchdog", daemon=True).start()


def measure_request_ceiling(torch, device, table_bytes):
    """The part's random-request rate measured in THIS run, on this box (boxes differ by ~5 %): uniformly random aligned 64-byte
    and 128-byte requests with the probe's access shape over a buffer of the table's size (tools/gather_rate.hip ->
    slacken_amd/lib/libslk_gather.so, built by `make all`).  Call after the index has been closed.  -> dict or None."""
    import ctypes
    path = os.path.join(ROOT, "slacken_amd", "lib", "libslk_gather.so")
    if not os.path.exists(path):
        return None
    try:
        G = ctypes.CDLL(path)
        G.slk_gather_rate.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                                      ctypes.POINTER(ctypes.c_float)]
        G.slk_gather_rate.restype = ctypes.c_int
        torch.cuda.empty_cache()
        nbytes = min(int(table_bytes), 128 << 30) // 4096 * 4096
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)   # (contents irrelevant)
        torch.cuda.synchronize()
        out = {"bytes": nbytes}
        for lanes, name in ((4, "64B"), (8, "128B")):
            g, ms = ctypes.c_double(0), ctypes.c_float(0)
            best = 0.0
            for _ in range(2):
                rc = G.slk_gather_rate(buf.data_ptr(), nbytes, lanes, 1500.0, ctypes.byref(g), ctypes.byref(ms))
                if rc != 0:
                    return None
                best = max(best, g.value)
            out[name] = round(best, 2)
        del buf
        torch.cuda.empty_cache()
        return out
    except Exception as e:   # (a measurement aid: the bench line says so instead of failing)
        log("request-rate ceiling not measured:", repr(e))
        return None


def sharded_traffic(world, records_per_rank, reads_per_step):
    """`traffic` of the table-sharded line: HBM bytes of a full step from the counters on file (profiles/r04_traffic_sharded.json),
    quoted only for the kernel sources and the workload they were taken from -- else null, as the contract allows."""
    path = os.path.join(ROOT, "profiles", "r04_traffic_sharded.json")
    try:
        tj = json.load(open(path))
    except (OSError, ValueError):
        return {"traffic": None}
    if tj.get("kernel_source_hash") == kernel_source_hash() and tj.get("world") == world and \
            tj.get("records_per_rank") == records_per_rank and tj.get("reads_per_step") == reads_per_step:
        return {"traffic": tj["hbm_bytes_per_step"], "traffic_source": "profiles/r04_traffic_sharded.json (a full step; same kernel sources, same workload)",
                "l2_miss_requests_per_step": int(tj["l2_miss_requests_per_step"])}
    return {"traffic": None, "traffic_source": "profiles/r04_traffic_sharded.json is for other kernel sources or another workload: not quoted"}


def kernel_source_hash():
    """sha256 over the sources of the dominant kernel (lane.hip and the structures it shares, engine.h): a counter-derived
    figure on file is only quoted for the code it was taken from."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "slacken_amd", "csrc")
    for name in ("engine.h", "lane.hip"):
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """Parent of an N-rank run: nothing here has touched (or will touch) the GPU; the ranks are children."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting", n, "ranks:", " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


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


def make_genomes_device(torch, n_genomes, genome_len, seed, device):
    """uint8 tensor [n_genomes * genome_len] of uniform ACGT on the GPU."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    total = n_genomes * genome_len
    out = torch.empty((total,), dtype=torch.uint8, device=device)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    CH = 1 << 28
    for s in range(0, total, CH):
        n = min(CH, total - s)
        out[s:s + n] = acgt[torch.randint(0, 4, (n,), generator=g, device=device)]
    return out


def make_reads_device(torch, genome_cat, genome_len, n_genomes, n_reads, seed, device):
    """uint8 tensor [n_reads * READ_LEN] on the GPU + uint64-compatible offsets (int64 tensor)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    flat = torch.empty((n_reads * READ_LEN,), dtype=torch.uint8, device=device)  # exactly offsets[R] bytes: no padding needed
    out = flat.view(n_reads, READ_LEN)
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


def dry_run(args, rank, world):
    """The N-rank plumbing without a GPU: gloo rendezvous, barrier-bracketed timed loop, max over ranks, one line."""
    from slacken_amd import dist as sdist
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    n_reads = int(args.reads)
    x = np.arange(1 << 16, dtype=np.uint64)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        x = x * np.uint64(3) + np.uint64(1)
    if os.environ.get("SLK_BENCH_TEST_HANG_RANK") == str(rank):   # (tests/test_dist_cpu.py: what the watchdog does about a stuck rank)
        progress("test hang before the first barrier")
        time.sleep(3600)
    progress("dry run: barrier before the timed steps")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = x * np.uint64(3) + np.uint64(1)
    barrier()
    elapsed = sdist.max_over_ranks(time.perf_counter() - t0, dist, None)
    counts = sdist.allreduce_counts(np.array([n_reads], np.int64), dist, None)
    if rank == 0:
        print(json.dumps({"metric": "classify_throughput_150bp_standard224scale", "value": None, "unit": "M reads/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / max(1, args.steps) * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "dry run: no GPU work, rank plumbing only",
                                     "reads_all_ranks_per_step": int(counts[0])}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


XGMI_LINK_GBPS = 153.0   # per direction and link; a GPU has 7 (one to each peer of an 8-GPU node)


def dry_run_sharded(args, rank, world):
    """The table-sharded exchange without a GPU: every rank owns the keys that fall to it (slacken_amd.sharded.shard_of_numpy),
    sends its queries to their owners through the same Exchange the GPU path uses (split sizes with the overflow flag riding on
    them, keys out, answers back in the same order; gloo here), and checks every answer against the whole table."""
    import torch
    from slacken_amd import dist as sdist
    from slacken_amd import sharded
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    ex = sharded.Exchange(rank, world, dist, torch.device("cpu"))
    rng = np.random.default_rng(224)
    table_keys = np.unique(rng.integers(-2**62, 2**62, 50000, dtype=np.int64))
    table_taxa = rng.integers(1, 1000, len(table_keys)).astype(np.int32)
    mine = sharded.shard_of_numpy(table_keys, world) == rank
    my_keys, my_taxa = table_keys[mine], table_taxa[mine]          # (sorted: np.unique)
    n_reads = int(args.reads)
    qrng = np.random.default_rng(150 + rank)

    def step(flag):
        q = np.where(qrng.random(n_reads) < 0.7, qrng.choice(table_keys, n_reads), qrng.integers(-2**62, 2**62, n_reads, dtype=np.int64))
        owner = sharded.shard_of_numpy(q, world)
        order = np.argsort(owner, kind="stable")
        send_counts = np.bincount(owner, minlength=world).tolist()
        recv_counts, any_flag = ex.split_sizes(send_counts, flag)
        got = ex.all_to_all(torch.from_numpy(q[order]), send_counts, recv_counts).numpy()
        at = np.searchsorted(my_keys, got)
        at[at >= len(my_keys)] = 0
        ans = np.where(len(my_keys) and my_keys[at] == got, my_taxa[at], 0).astype(np.int32) if len(my_keys) else np.zeros(len(got), np.int32)
        back = ex.all_to_all(torch.from_numpy(ans), recv_counts, send_counts).numpy()
        taxa = np.empty(n_reads, np.int32)
        taxa[order] = back
        ref_at = np.searchsorted(table_keys, q)
        ref_at[ref_at >= len(table_keys)] = 0
        want = np.where(table_keys[ref_at] == q, table_taxa[ref_at], 0)
        if not np.array_equal(taxa, want):
            raise SystemExit(f"rank {rank}: the exchange returned wrong answers")
        return any_flag, len(q)

    def barrier():
        if dist is not None:
            dist.barrier()

    flags = []
    for i in range(args.warmup):
        flags.append(step(rank == world - 1 and i == 0)[0])      # the last rank raises its flag once: every rank must see it
    barrier()
    t0 = time.perf_counter()
    sent = 0
    for _ in range(args.steps):
        f, n = step(False)
        flags.append(f)
        sent += n
    barrier()
    elapsed = sdist.max_over_ranks(time.perf_counter() - t0, dist, None)
    counts = sdist.allreduce_counts(np.array([n_reads, sent], np.int64), dist, None)
    if args.warmup and flags[:1] != [True] or any(flags[1:]):
        raise SystemExit(f"rank {rank}: overflow flags {flags}")
    if rank == 0:
        print(json.dumps({"metric": "classify_throughput_150bp_standard224scale", "value": None, "unit": "M reads/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / max(1, args.steps) * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "dry run of the table-sharded exchange: no GPU work, every answer checked",
                                     "parallelism": f"table-sharded x{world}", "reads_all_ranks_per_step": int(counts[0]),
                                     "keys_exchanged_all_ranks": int(counts[1])}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def table_sharded(args, rank, world, local_rank):
    """BASELINE.json configs[3] (module docstring).  One rank = one GPU = 1/N of the table."""
    import torch
    from slacken_amd import dist as sdist
    from slacken_amd import sharded
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.collectives_at_one_rank:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)     # (RCCL refuses two ranks on one GPU; the exchange is staged through host memory)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    import slacken_amd
    n_reads, per_rank = int(args.reads), int(args.records_per_rank)
    t0 = time.time()
    parents, taxa, leaves = build_taxonomy()
    rng = np.random.default_rng(224)
    G = min(args.genomes, len(leaves))
    genome_taxa = rng.choice(leaves, size=G, replace=False).astype(np.int32)
    genome_cat = make_genomes_device(torch, G, args.genome_len, 224 + 1, device)     # (the same genomes on every rank)
    g_offsets = np.arange(0, (G + 1) * args.genome_len, args.genome_len, dtype=np.uint64)

    # ---- this rank's shard of the table: every rank scans all genomes and draws the same padding keys, and keeps its share
    expect_genome = int(G * args.genome_len * 0.36 / world * 1.1) + 4096
    ix = slacken_amd.Index(k=K, m=M, spaces=SPACES, expected_records=max(per_rank, expect_genome), max_taxon=TAX_EXTENT - 1,
                           load_factor=args.load_factor, device=local_rank)
    ix.set_shard(rank, world)
    ix.set_taxonomy(parents)
    ix.add_sequences_device(genome_cat.data_ptr(), g_offsets, genome_taxa)
    n_genome_records = int(ix.info().records)
    log(f"rank {rank}: {n_genome_records} genome records of this shard ({time.time() - t0:.1f}s)")
    smask = ((2**62 - 1) & ~0x0CCCCCCC) << 2
    smask_i64 = smask - (1 << 64) if smask >= (1 << 63) else smask
    d_taxa = torch.from_numpy(taxa).to(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(224 + 7)                                   # (one global key stream; a rank keeps the keys that fall to it)
    CH = 1 << 27
    while int(ix.info().records) < per_rank:
        hi = torch.randint(0, 2**32, (CH,), generator=gen, device=device, dtype=torch.int64)
        lo = torch.randint(0, 2**32, (CH,), generator=gen, device=device, dtype=torch.int64)
        keys = ((hi << 32) | lo) & smask_i64
        del hi, lo
        tx = d_taxa[torch.randint(0, len(taxa), (CH,), generator=gen, device=device)]
        if world > 1:                                          # (the engine would drop the others itself; this spares it the copy)
            keep = sharded.shard_of_torch(keys, world) == rank
            keys, tx = keys[keep].contiguous(), tx[keep].contiguous()
        n = min(int(keys.numel()), per_rank - int(ix.info().records))
        torch.cuda.synchronize()
        ix.append_device(keys.data_ptr(), tx.data_ptr(), n)
        del keys, tx
    ix.finalize()
    info = ix.info()
    torch.cuda.empty_cache()
    log(f"rank {rank}: shard {rank}/{world}: {info.records} records in {info.table_bytes / 2**30:.1f} GiB, load "
        f"{info.records / (info.buckets * info.bucket_cells):.2f} ({time.time() - t0:.1f}s)")

    d_bases, d_offsets = make_reads_device(torch, genome_cat, args.genome_len, G, n_reads, 150 + rank, device)
    del genome_cat
    torch.cuda.empty_cache()
    total_bases = n_reads * READ_LEN
    sc = sharded.ShardedClassifier(ix, rank, world, dist, device, exchange_on_cpu=args.rehearse_on_one_gpu,
                                   force_collectives=args.collectives_at_one_rank)
    batch = (d_bases, d_offsets, n_reads, total_bases, None)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    progress("table-sharded: warm-up batches")
    outs = sc.classify_many([batch] * max(1, args.warmup), thresholds=(0.0,), min_hit_groups=2)
    if outs is None:
        raise SystemExit("the fast sharded route does not take this splitter")
    barrier()
    progress("table-sharded: timed batches")
    t_start = time.perf_counter()
    outs = sc.classify_many([batch] * args.steps, thresholds=(0.0,), min_hit_groups=2, profile=True)   # K steps = K batches in the pipeline
    barrier()
    elapsed = time.perf_counter() - t_start
    progress("table-sharded: jobs alone")
    step_ms = list(getattr(sc, "step_ms", []))
    alone = sc.jobs_alone(batch) if world == 1 and not args.collectives_at_one_rank else None   # (after the timed region)
    elapsed = sdist.max_over_ranks(elapsed, dist, None if args.rehearse_on_one_gpu else device)
    ms_per_step = elapsed / args.steps * 1e3
    reads_per_s = world * n_reads / (elapsed / args.steps)
    o = outs[-1]
    stage = sc.stage_ms or {}
    keys_per_batch, remote = int(o["exchanged_keys"]), int(o["sent_remote_keys"])
    looked_up = int(o["looked_up_keys"])
    classified = float(o["classified"][:n_reads].float().mean().item())
    n_deferred = int(o.get("deferred", 0))
    x_ms = stage.get("exchange_keys", 0.0)
    per_link = (remote / max(1, world - 1)) * 8 / (x_ms * 1e-3) / 1e9 if world > 1 and x_ms > 0 else None
    # the owner's lookups ride inside the step kernel: their rate is taken over a steady-state step (one that carries all three jobs:
    # steps 4 .. K - 1 of the K + 4 of a run)
    steady = step_ms[4:args.steps] if len(step_ms) >= args.steps and args.steps > 4 else step_ms
    steady_ms = float(np.median(steady)) if steady else 0.0
    lookup_rate = looked_up / (steady_ms * 1e-3) / 1e9 if steady_ms > 0 else None
    # algorithmic bytes of a step on one rank (SURVEY 8d's B(r), with the probes served by whichever rank owns them) against the
    # WHOLE pipeline's time per step
    bytes_per_step = total_bases + 64 * looked_up + 8 * n_reads
    # the part's random-request rate on THIS box, measured now that the timed region is over (the shard is closed for it)
    ceiling, ceiling_src, ceil_meas = GATHER_CEILING_GLPS, "constant of round 2 (FALLBACK: not measured in this run)", None
    if rank == 0 and not args.no_ceiling and not args.rehearse_on_one_gpu:
        progress("request-rate ceiling")
        table_bytes = int(info.table_bytes)
        del outs, o, batch, d_bases, d_offsets
        sc.close()
        ix.close()
        ceil_meas = measure_request_ceiling(torch, device, table_bytes)
        if ceil_meas:
            ceiling = ceil_meas["64B"]
            ceiling_src = f"measured in this run: uniformly random 64-byte requests over {ceil_meas['bytes'] / 2**30:.0f} GiB, tools/gather_rate.hip"
    out = {
        "metric": "classify_throughput_150bp_standard224scale", "value": round(reads_per_s / 1e6, 3), "unit": "M reads/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        **({"rehearsal": "all ranks on GPU 0, exchange through host memory (gloo): not a measurement"} if args.rehearse_on_one_gpu else {}),
        **({"collectives": "one rank sends to itself through the RCCL all-to-all: the exchange stages are device-local copies"}
           if args.collectives_at_one_rank and world == 1 else {}),
        "config": {
            "workload": "oversized custom library, hash-sharded over the GPUs (k=35,m=31,s=7), synthetic 150 bp single-end reads; "
                        "minimizers to their owners and taxa back by all-to-all (BASELINE.json configs[3])",
            "parallelism": f"table-sharded x{world}", "pipeline": "step t = one kernel: EMIT(t) + LOOKUP(t - 2) + APPLY(t - 4); exchanges of t - 1 and t - 3 beside it", "records_per_rank": int(info.records), "records_all_ranks": int(info.records) * world,
            "table_GiB_per_rank": round(info.table_bytes / 2**30, 1), "table_GB_all_ranks": round(info.table_bytes * world / 1e9, 1),
            "exceeds_one_gpu_288GB": bool(info.table_bytes * world > 288e9),
            "bucket_bytes": int(info.bucket_cells) * 8, "table_load": round(info.records / (info.buckets * info.bucket_cells), 3),
            "genomes": G, "genome_len": args.genome_len, "reads_per_gpu_per_step": n_reads, "read_len": READ_LEN,
            "classified_fraction": round(classified, 4), "deferred_to_staged_route": n_deferred,
            "stage_ms_in_pipeline": {k: round(v, 3) for k, v in stage.items()},
            "steady_state_step_ms": round(steady_ms, 3), "step_ms_all": [round(v, 2) for v in step_ms],
            **({"job_ms_alone": alone} if alone else {}),
            "keys_per_read": round(keys_per_batch / n_reads, 3),
            "exchanged_bytes_per_read": round(12.0 * keys_per_batch / n_reads, 1),     # 8-byte key out, 4-byte taxon back
            "remote_bytes_per_read": round(12.0 * remote / n_reads, 1),
            "xgmi_link_GBps_keys": None if per_link is None else round(per_link, 1), "xgmi_link_peak_GBps": XGMI_LINK_GBPS,
            "xgmi_link_frac": None if per_link is None else round(per_link / XGMI_LINK_GBPS, 3),
        },
        "roofline": {"bound": "hbm", "kernel": "the sharded pipeline: lane_step_kernel (EMIT of batch t, LOOKUP of batch t - 2 and APPLY of batch t - 4 in one launch) "
                                               "beside the exchanges; K batches take K + 4 steps", "achieved": round(bytes_per_step / (ms_per_step * 1e-3) / 1e9, 1),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(bytes_per_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     **sharded_traffic(world, per_rank, n_reads), "algorithmic_bytes_per_launch": bytes_per_step, "kernel_ms": round(ms_per_step, 3),
                     "lookup_stage_Grequests_per_s": None if lookup_rate is None else round(lookup_rate, 2),
                     "random_line_ceiling_Glines_per_s": ceiling, "random_line_ceiling_source": ceiling_src,
                     "lookup_stage_frac_of_request_rate_ceiling": None if lookup_rate is None else round(lookup_rate / ceiling, 3)},
    }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--records", type=float, default=1.0e10, help="library records (standard-224-scale)")
    ap.add_argument("--reads", type=float, default=1.0e7, help="150 bp reads per GPU per step")
    ap.add_argument("--reads-total", type=float, default=0.0,
                    help="150 bp reads per step over ALL GPUs (overrides --reads: each rank takes reads_total / N; BASELINE.json "
                         "configs[2] as written is --gpus 8 --reads-total 1e8 = 12.5 M per GPU)")
    ap.add_argument("--genomes", type=int, default=8192)
    ap.add_argument("--genome-len", type=int, default=1 << 20)
    ap.add_argument("--load-factor", type=float, default=0.0, help="cells-used fraction of the table (0: the engine's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ceiling", action="store_true", help="do not measure the random-request rate after the timed region")
    ap.add_argument("--table-sharded", action="store_true",
                    help="BASELINE configs[3]: every rank holds 1/N of the table, minimizers and taxa cross the links (RCCL all-to-all)")
    ap.add_argument("--records-per-rank", type=float, default=5.0e9, help="--table-sharded: records of one rank's shard")
    ap.add_argument("--collectives-at-one-rank", action="store_true",
                    help="--table-sharded with ONE rank: the keys and taxa go through the RCCL all-to-all all the same (the rank sends to "
                         "itself), so that the `nccl` process group and its device collectives run on a one-GPU box")
    ap.add_argument("--dry-run", action="store_true", help="rank plumbing only (gloo, no GPU)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N ranks that all use GPU 0 (gloo for the rendezvous): rehearses the N-rank GPU path on a one-GPU box; "
                         "not a measurement")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))  # (no GPU call has been made in this process)
    from slacken_amd import dist as sdist
    rank, world, local_rank = sdist.env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}; start it as `python bench.py --gpus N` or "
                         f"under torch.distributed.run with --nproc-per-node equal to --gpus")
    arm_watchdog(rank)
    if args.reads_total > 0:
        args.reads = float(int(args.reads_total) // world)
    if args.dry_run:
        return dry_run_sharded(args, rank, world) if args.table_sharded else dry_run(args, rank, world)
    if args.table_sharded:
        return table_sharded(args, rank, world, local_rank)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")     # (RCCL refuses two ranks on one GPU)
        else:
            dist.init_process_group("nccl", device_id=device)

    import slacken_amd
    n_records, n_reads = int(args.records), int(args.reads)
    t0 = time.time()
    parents, taxa, leaves = build_taxonomy()
    rng = np.random.default_rng(224)
    G = min(args.genomes, len(leaves))
    genome_taxa = rng.choice(leaves, size=G, replace=False).astype(np.int32)
    genome_cat = make_genomes_device(torch, G, args.genome_len, 224 + 1, device)
    g_offsets = np.arange(0, (G + 1) * args.genome_len, args.genome_len, dtype=np.uint64)
    torch.cuda.synchronize()
    log(f"rank {rank}: taxonomy {len(taxa) + 1} nodes, {G} genomes x {args.genome_len} bp on the device ({time.time() - t0:.1f}s)")

    # ---- HBM-resident table: the genomes' minimizers (LCA-merged by the engine's library builder) + random padding
    # distinct minimizers of the genomes: about 2/(w+1) per k-mer window on random sequence (w = 5)
    expect_genome = int(G * args.genome_len * 0.36) + 1024
    ix = slacken_amd.Index(k=K, m=M, spaces=SPACES, expected_records=max(n_records, expect_genome),
                           max_taxon=TAX_EXTENT - 1, load_factor=args.load_factor, device=local_rank)
    ix.set_taxonomy(parents)
    ix.add_sequences_device(genome_cat.data_ptr(), g_offsets, genome_taxa)
    n_genome_records = int(ix.info().records)
    log(f"rank {rank}: {n_genome_records} genome records ({time.time() - t0:.1f}s)")
    smask = ((2**62 - 1) & ~0x0CCCCCCC) << 2  # SpacedSeed mask for m=31, s=7 (48 significant bits), left-aligned
    smask_i64 = smask - (1 << 64) if smask >= (1 << 63) else smask
    d_taxa = torch.from_numpy(taxa).to(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(224 + 7)
    pad, CH = max(0, n_records - n_genome_records), 1 << 27
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
    ix.finalize()
    info = ix.info()
    if not info.bucket_cells:   # (an A/B library of an earlier round: 64-byte buckets, no such field)
        info.bucket_cells = 8
    torch.cuda.empty_cache()
    log(f"rank {rank}: table {info.records} records in {info.table_bytes / 2**30:.1f} GiB "
        f"({info.buckets} buckets of {info.bucket_cells * 8} bytes, load {info.records / (info.buckets * info.bucket_cells):.2f}, max displacement "
        f"{info.max_displacement}, {info.duplicate_keys} duplicate pad keys dropped) ({time.time() - t0:.1f}s)")

    # ---- reads resident in HBM
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
    # Dominant kernel: HIP events recorded on the ENGINE's stream (torch sees it as an external stream) bracket every timed
    # step's launches -- no host synchronisation inside the timed region; read back after it.  A step's bracket holds the
    # lane kernel and the near-empty launches behind it (the passes over its hand-on lists: a few microseconds each).
    ext = torch.cuda.ExternalStream(st.hip_stream, device=device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t_start = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(ext)
        step()
        ev[i][1].record(ext)
    barrier()
    elapsed = time.perf_counter() - t_start
    st.synchronize()   # (the engine's own check of the queued calls: raises if a device-side error was flagged)
    step_dev_ms = np.array([a.elapsed_time(b) for a, b in ev]) if args.steps else np.zeros(1)
    stage_ms = np.array(st.last_stage_ms())  # the engine's own events of the LAST timed step: [fused, 0, ~0] on the hot path
    elapsed = sdist.max_over_ranks(elapsed, dist, None if args.rehearse_on_one_gpu else device)
    ms_per_step = elapsed / args.steps * 1e3
    reads_per_s = world * n_reads / (elapsed / args.steps)

    # ---- algorithmic bytes: B(r) = L_r + 64 * P_r + 8
    probes = int(d_np.sum().item())
    classified = float(d_cls.float().mean().item())
    bytes_per_launch = total_bases + 64 * probes + 8 * n_reads
    fused = float(stage_ms[1]) < 0.05 and float(stage_ms[2]) < 0.05  # one fused launch: [fused, 0, ~0]
    # hot path: mean device time of a step's launches over the K timed steps; staged path (other splitters): the probe kernel
    dom_ms = float(step_dev_ms.mean()) if fused else float(stage_ms[1])
    dom_name = ("slk::lane_kernel<true> (scan+probe+LCA fused, lane per read; followed by the near-empty launches of the passes "
                "over its hand-on lists: lane_kernel<LONG>, segment_kernel, order_wave_list_kernel, fused_kernel<1> twice)") if fused else "slk::probe_kernel"
    achieved = bytes_per_launch / (dom_ms * 1e-3) / 1e9
    path_achieved = bytes_per_launch / (float(stage_ms.sum()) * 1e-3) / 1e9

    # HBM traffic from the PMC counters is collected by separate rocprofv3 --pmc passes of this command
    # (tools/profile.sh) and kept on file with the hash of the kernel sources it was measured on: quoted only on a match.
    traffic, traffic_note, miss_requests = None, "no counter file for this build", None
    if os.path.exists(TRAFFIC_FILE):
        tj = json.load(open(TRAFFIC_FILE))
        same = (tj.get("reads_per_launch") == n_reads and tj.get("records") == int(args.records)
                and tj.get("genomes") == [G, args.genome_len])
        if same and tj.get("kernel_source_hash") == kernel_source_hash():
            traffic, traffic_note = tj["hbm_bytes_per_launch"], f"profiles/{os.path.basename(TRAFFIC_FILE)} (same kernel sources, same workload)"
            if tj.get("tcc_miss_x64_bytes"):
                miss_requests = tj["tcc_miss_x64_bytes"] // 64
        else:
            traffic_note = f"profiles/{os.path.basename(TRAFFIC_FILE)} is for other kernel sources or another workload: not quoted"

    # ---- after the timed region: the CPU baseline, then the part's random-request rate on THIS box (the index is closed for it)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        progress("cpu baseline")
        cpu = cpu_baseline(torch, slacken_amd, args, parents, genome_cat, genome_taxa, G, device)
    deferred = st.last_deferred()
    ceiling, ceiling_src, ceil_meas = GATHER_CEILING_GLPS, "constant of round 2 (FALLBACK: not measured in this run)", None
    if rank == 0 and not args.no_ceiling:
        progress("request-rate ceiling")
        table_bytes = int(info.table_bytes)
        del d_bases, d_offsets, genome_cat
        st.close()
        ix.close()
        ceil_meas = measure_request_ceiling(torch, device, table_bytes)
        if ceil_meas:
            ceiling = ceil_meas["64B"]
            ceiling_src = (f"measured in this run: uniformly random 64-byte requests (4 lanes x 16 B) over {ceil_meas['bytes'] / 2**30:.0f} GiB, "
                           f"tools/gather_rate.hip")
    log(f"rank {rank}: request-rate ceiling {ceiling} G/s ({ceiling_src})")

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
        "reads_total": world * n_reads,
        **({"rehearsal": "all ranks on GPU 0: not a measurement"} if args.rehearse_on_one_gpu else {}),
        "config": {
            "workload": "standard-224-scale synthetic library (k=35,m=31,s=7), synthetic 150 bp single-end reads, "
                        "full table resident in HBM (BASELINE.json configs[1])",
            "records": int(info.records), "table_GiB": round(info.table_bytes / 2**30, 1), "bucket_bytes": int(info.bucket_cells) * 8,
            "table_load": round(info.records / (info.buckets * info.bucket_cells), 3), "max_displacement": int(info.max_displacement),
            "genomes": G, "genome_len": args.genome_len, "genome_records": n_genome_records,
            "reads_per_gpu_per_step": n_reads, "reads_total_per_step": world * n_reads, "read_len": READ_LEN,
            "parallelism": f"read-sharded x{world}, table replicated",
            "scaling_note": (f"weak: every rank classifies its own {n_reads} reads per step, value = {world} x {n_reads} reads / the slowest rank's time"
                             + ("; --reads-total given: the step's reads are BASELINE.json configs[2]'s total cut into equal shares" if args.reads_total > 0
                                else "; BASELINE.json configs[2] as written (100 M reads over 8 GPUs) is --gpus 8 --reads-total 1e8")),
            "probes_per_read": round(probes / n_reads, 3), "classified_fraction": round(classified, 4),
            "deferred_to_wave_kernel": deferred,
            "stage_ms": ({"fused_last_step": round(float(stage_ms[0]), 3), "step_device_ms_min": round(float(step_dev_ms.min()), 3),
                          "step_device_ms_max": round(float(step_dev_ms.max()), 3)} if fused else
                         {"scan": round(float(stage_ms[0]), 3), "probe": round(float(stage_ms[1]), 3),
                          "classify": round(float(stage_ms[2]), 3)}),
            "path_GBps_all_kernels": round(path_achieved, 1),
            "path_frac_all_kernels": round(path_achieved / HBM_PEAK_GBPS, 4),
        },
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "traffic_source": traffic_note,
                     "algorithmic_bytes_per_launch": bytes_per_launch, "kernel_ms": round(dom_ms, 3),
                     # the part's rate of random 64-byte requests over a buffer of the table's size: what a hash-table probe can
                     # reach, as opposed to the streaming peak
                     "random_line_ceiling_Glines_per_s": ceiling,
                     "random_line_ceiling_source": ceiling_src,
                     "random_128B_request_ceiling_G_per_s": ceil_meas["128B"] if ceil_meas else None,
                     "probe_lines_per_s_G": round(probes / (dom_ms * 1e-3) / 1e9, 2),
                     "frac_of_random_line_ceiling": round(probes / (dom_ms * 1e-3) / 1e9 / ceiling, 3),
                     # every L2 miss of the kernel (probes, second-bucket probes, the read stream's lines, outputs; TCC_MISS_sum of the
                     # counter file) against the same ceiling: the request rate is what bounds the path
                     "l2_miss_requests_per_launch": miss_requests,
                     "frac_of_request_rate_ceiling": (round(miss_requests / (dom_ms * 1e-3) / 1e9 / ceiling, 3)
                                                      if miss_requests else None)},
    }
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(torch, slacken_amd, args, parents, genome_cat, genome_taxa, G, device):
    """The CPU restatement (oracle/, OpenMP over reads) on a bounded sample of the same workload (sized for roughly 12 s):
    reads of the same generator drawn from the first 64 genomes, against those genomes' records plus random padding up to
    2^24 records (hits come from the genome records; the padding keeps the hash table far larger than the CPU caches, as
    the full library would) -- a CPU table of the full 1.0e10 records would take longer to build than the whole bench run."""
    ncpu = len(os.sched_getaffinity(0))
    os.environ.setdefault("OMP_NUM_THREADS", str(ncpu))
    from oracle import oracle
    Gc = min(G, 64)
    L = args.genome_len
    tmp = slacken_amd.Index(k=K, m=M, spaces=SPACES, expected_records=int(Gc * L * 0.5) + 1024, max_taxon=TAX_EXTENT - 1,
                            device=device.index)
    tmp.set_taxonomy(parents)
    tmp.add_sequences_device(genome_cat.data_ptr(), np.arange(0, (Gc + 1) * L, L, dtype=np.uint64), genome_taxa[:Gc])
    gkeys, gtax = tmp.export()
    tmp.close()
    p = oracle.params(k=K, m=M, spaces=SPACES)
    rng = np.random.default_rng(99)
    pad = max(0, (1 << 24) - len(gkeys))
    pkeys = (rng.integers(0, 2**63, pad, dtype=np.uint64) * np.uint64(2)) & np.uint64(p.space[0])
    ptax = np.full(pad, 1, np.int32)
    oix = oracle.Index(1, np.concatenate([gkeys, pkeys.view(np.int64)]), np.concatenate([gtax, ptax]))
    n_sample = 2_000_000
    d_bases, _ = make_reads_device(torch, genome_cat, L, Gc, n_sample, 151, device)
    h_bases = d_bases[:n_sample * READ_LEN].cpu().numpy()
    del d_bases
    probe_n = 200000
    offsets = np.arange(0, (probe_n + 1) * READ_LEN, READ_LEN, dtype=np.uint64)
    t = time.perf_counter()
    res = oracle.classify_batch(p, oix, parents, h_bases[:probe_n * READ_LEN], offsets)  # also warms the table
    rate = probe_n / (time.perf_counter() - t)
    S = int(min(n_sample, max(probe_n, rate * 12.0)))
    bases = h_bases[:S * READ_LEN]
    offsets = np.arange(0, (S + 1) * READ_LEN, READ_LEN, dtype=np.uint64)
    passes, dt = 0, 0.0
    while dt < 10.0 and passes < 8:  # about 10-30 s of wall time on the host cores
        t = time.perf_counter()
        res = oracle.classify_batch(p, oix, parents, bases, offsets)
        dt += time.perf_counter() - t
        passes += 1
    return {"value": round(S * passes / dt / 1e6, 4), "unit": "M reads/s", "cores": int(res["threads"]), "kind": "port",
            "sample": f"{passes} pass(es) over {S} reads of the step's generator drawn from the first {Gc} genomes, CPU restatement "
                      f"(oracle/, OpenMP, {res['threads']} threads on {ncpu} usable CPUs) vs their {len(gkeys)} records + random "
                      f"padding to 2^24 records in a DRAM hash table; {dt:.1f} s; the reference's own Spark path cannot run "
                      f"here (no JVM)"}


if __name__ == "__main__":
    main()

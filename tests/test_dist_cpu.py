"""The N>1 path on CPU: two gloo ranks shard the reads, classify nothing themselves (no GPU here) but run exactly the
sharding, count-merging and timing code bench.py and a multi-GPU host use."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from slacken_amd import dist as sdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, taxa, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sdist.shard_range(len(taxa), rank, world)
    counts = sdist.taxon_read_counts(taxa[lo:hi], 64)
    total = sdist.allreduce_counts(counts, dist)
    slowest = sdist.max_over_ranks(1.0 + rank, dist)
    if rank == 0:
        np.save(out, np.concatenate([total, [int(slowest * 1000), lo, hi]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_read_sharding(tmp_path):
    rng = np.random.default_rng(5)
    taxa = rng.integers(0, 64, 10001)
    out = str(tmp_path / "r0.npy")
    mp.spawn(_worker, args=(2, _free_port(), taxa, out), nprocs=2, join=True)
    got = np.load(out)
    assert np.array_equal(got[:64], np.bincount(taxa, minlength=64))  # merged counts == unsharded counts
    assert got[64] == 2000                                            # max over ranks of the per-rank time
    assert (got[65], got[66]) == (0, 5000)


def test_shard_ranges_partition():
    for n in (0, 1, 7, 10**7 + 3):
        for world in (1, 2, 3, 8):
            r = [sdist.shard_range(n, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1


def _bench(*args, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], capture_output=True, text=True, env=e, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, [json.loads(ln) for ln in lines]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun around it) must itself start two ranks and print ONE line with n_gpus = 2
    (dry mode: gloo, no GPU work; the launch, rendezvous, barrier, max-over-ranks and reporting code are the real ones)."""
    p, lines = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--reads", "1000", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    assert lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 3 and lines[0]["dry_run"] is True
    assert lines[0]["config"]["reads_all_ranks_per_step"] == 2000   # summed over both ranks


def test_bench_refuses_a_world_size_mismatch():
    """Under a launcher whose WORLD_SIZE differs from --gpus the bench exits non-zero instead of reporting a 1-rank number as N."""
    p, lines = _bench("--gpus", "4", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and not lines
    assert "WORLD_SIZE=1 but --gpus 4" in p.stderr


def test_bench_single_rank_dry_line():
    p, lines = _bench("--dry-run", "--steps", "2")
    assert p.returncode == 0 and len(lines) == 1 and lines[0]["n_gpus"] == 1

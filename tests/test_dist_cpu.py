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


def test_bench_table_sharded_dry_two_ranks():
    """configs[3] without a GPU: `bench.py --gpus 2 --table-sharded --dry-run` starts its two ranks, each owns the keys that fall to
    it, queries travel to their owners and the answers back through slacken_amd.sharded.Exchange (the GPU path's own exchange code,
    over gloo), every answer is checked against the whole table, and the overflow flag that rides on the split sizes reaches
    every rank."""
    p, lines = _bench("--gpus", "2", "--table-sharded", "--steps", "3", "--warmup", "1", "--reads", "4000", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["config"]["parallelism"] == "table-sharded x2"
    assert line["config"]["reads_all_ranks_per_step"] == 8000 and line["config"]["keys_exchanged_all_ranks"] == 3 * 8000


def test_bench_table_sharded_dry_single_rank():
    p, lines = _bench("--table-sharded", "--steps", "2", "--warmup", "1", "--reads", "1000", "--dry-run")
    assert p.returncode == 0 and len(lines) == 1 and lines[0]["config"]["parallelism"] == "table-sharded x1"


def _exchange_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from slacken_amd import sharded
    ex = sharded.Exchange(rank, world, dist, torch.device("cpu"))
    # rank r sends (r + 1) * (d + 1) items to rank d, each item = 1000 * r + d
    send_counts = [(rank + 1) * (d + 1) for d in range(world)]
    send = torch.cat([torch.full((c,), 1000 * rank + d, dtype=torch.int64) for d, c in enumerate(send_counts)])
    recv_counts, flag = ex.split_sizes(send_counts, flag=(rank == 1))
    assert recv_counts == [(s + 1) * (rank + 1) for s in range(world)] and flag is True
    got = ex.all_to_all(send, send_counts, recv_counts)
    want = torch.cat([torch.full(((s + 1) * (rank + 1),), 1000 * s + rank, dtype=torch.int64) for s in range(world)])
    assert torch.equal(got, want)
    back = ex.all_to_all((got * 2).to(torch.int32), recv_counts, send_counts)     # answers return in the order the keys were sent
    assert torch.equal(back.long(), send * 2)
    _, flag2 = ex.split_sizes(send_counts, flag=False)
    assert flag2 is False and ex.any_rank(rank == 2) is True and ex.any_rank(False) is False
    if rank == 0:
        np.save(out, np.array([1]))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_three_ranks(tmp_path):
    """Exchange on three gloo ranks with uneven split sizes (also a non-power-of-two world): sizes, order of the received items,
    the way back, and the flag that rides on the split sizes."""
    out = str(tmp_path / "ok.npy")
    mp.spawn(_exchange_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    assert os.path.exists(out)


def test_bench_watchdog_names_the_stage_of_a_stuck_rank():
    """A rank that stops making progress (here: rank 1 sleeps before the first barrier, so rank 0 hangs IN the barrier) is not
    waited for forever: every rank's watchdog prints the stage it is in and exits non-zero; nothing is re-executed."""
    p, lines = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "1000", "--dry-run",
                      env={"SLK_BENCH_WATCHDOG_S": "10", "SLK_BENCH_TEST_HANG_RANK": "1"})
    assert p.returncode != 0 and not lines
    assert "WATCHDOG: no progress" in p.stderr
    # (the launcher ends the other rank as soon as the first one has given up: one of the two lines is certain, usually both)
    assert ("stage: test hang before the first barrier" in p.stderr            # the stuck rank
            or "stage: dry run: barrier before the timed steps" in p.stderr)   # the rank that waits for it


def test_bench_reads_total_is_cut_into_equal_shares():
    """BASELINE.json configs[2] is a TOTAL (100 M reads over 8 GPUs): --reads-total N gives every rank N / world reads."""
    p, lines = _bench("--gpus", "4", "--steps", "2", "--warmup", "1", "--reads-total", "10000", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    assert lines[0]["n_gpus"] == 4 and lines[0]["config"]["reads_all_ranks_per_step"] == 10000


def test_bench_table_sharded_dry_eight_ranks():
    """The table-sharded exchange protocol at the node's real world size (8 gloo ranks on CPU): every answer checked."""
    p, lines = _bench("--gpus", "8", "--table-sharded", "--steps", "2", "--warmup", "1", "--reads", "3000", "--dry-run")
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8 and lines[0]["config"]["parallelism"] == "table-sharded x8"
    assert lines[0]["config"]["reads_all_ranks_per_step"] == 8 * 3000


def _zero_split_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from slacken_amd import sharded
    ex = sharded.Exchange(rank, world, dist, torch.device("cpu"))
    # rank r sends NOTHING to rank (r + 1) % world and to itself, 5 + r items to the others; rank 2 sends nothing at all
    send_counts = [0 if (d == rank or d == (rank + 1) % world or rank == 2) else 5 + rank for d in range(world)]
    send = torch.cat([torch.full((c,), 100 * rank + d, dtype=torch.int64) for d, c in enumerate(send_counts)] or [torch.zeros(0, dtype=torch.int64)])
    recv_counts, _ = ex.split_sizes(send_counts)
    want_counts = [0 if (rank == s or rank == (s + 1) % world or s == 2) else 5 + s for s in range(world)]
    assert recv_counts == want_counts
    got = ex.all_to_all(send, send_counts, recv_counts)
    want = torch.cat([torch.full((c,), 100 * s + rank, dtype=torch.int64) for s, c in enumerate(want_counts)])
    assert torch.equal(got, want)
    back = ex.all_to_all((got + 1).to(torch.int32), recv_counts, send_counts)
    assert torch.equal(back.long(), send + 1)
    if rank == 0:
        np.save(out, np.array([1]))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_with_zero_length_splits(tmp_path):
    """Peers that get nothing (and a rank that sends nothing at all): the all-to-all(v) with zero-length splits in both directions."""
    out = str(tmp_path / "ok.npy")
    mp.spawn(_zero_split_worker, args=(4, _free_port(), out), nprocs=4, join=True)
    assert os.path.exists(out)

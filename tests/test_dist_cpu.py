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

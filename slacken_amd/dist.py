"""Multi-GPU plumbing for the classify path: one process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm,
"gloo" on CPU for tests).  The path shards by READS with the table replicated: rank g classifies
fragments [g*R/G, (g+1)*R/G); there is NO data-path collective.  The only cross-rank steps are the report's per-taxon
read counts (one small all-reduce, the GPU counterpart of the groupBy(taxon).count behind KrakenReport,
S/slacken/Classifier.scala:245-251) and the max-over-ranks timing used by bench.py."""
import os

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(n, rank, world):
    """Contiguous read range of `rank`: [rank*n/world, (rank+1)*n/world)."""
    return (rank * n) // world, ((rank + 1) * n) // world


def taxon_read_counts(taxa, size):
    """Per-taxon read counts of one shard (index = taxon id, 0 = unclassified)."""
    return np.bincount(np.asarray(taxa, dtype=np.int64), minlength=size).astype(np.int64)


def allreduce_counts(counts, dist=None, device=None):
    """Sum the shards' per-taxon counts (identity when not distributed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return counts
    import torch
    t = torch.as_tensor(counts, dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def max_over_ranks(seconds, dist=None, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

"""Synthetic taxonomies modelled on the reference's generator (T/slacken/Testing.scala:62-83): 8 ranks below the
root, `level_size` nodes per rank with contiguous ids, each node's parent uniform among all lower ids >= 1."""
import numpy as np

N_RANKS = 8  # Superkingdom .. Species, Taxonomy.scala:40-48


def taxonomy(size, rng):
    level_size = size // N_RANKS + 1
    n = N_RANKS * level_size + 2
    parents = np.zeros(n, np.int32)
    for depth in range(1, N_RANKS + 1):
        max_parent = (depth - 1) * level_size + 1
        for tid in range((depth - 1) * level_size + 2, depth * level_size + 2):
            parents[tid] = rng.integers(1, max_parent + 1)
    parents[1] = 0  # Taxonomy.fromNodesAndNames: parents(ROOT) = NONE
    return parents


def sparse_relabel(parents, extent, rng):
    """Spread the defined taxa over ids < extent (NCBI-like sparse id space); ROOT stays 1."""
    defined = [t for t in range(2, len(parents)) if parents[t] != 0]
    new_ids = rng.choice(np.arange(2, extent), size=len(defined), replace=False)
    remap = {0: 0, 1: 1}
    remap.update({old: int(new) for old, new in zip(defined, new_ids)})
    out = np.zeros(extent, np.int32)
    for old in defined:
        out[remap[old]] = remap[int(parents[old])]
    return out, remap


def defined_taxa(parents):
    return [t for t in range(1, len(parents)) if parents[t] != 0 or t == 1]


def path_to_root(parents, t):
    out = []
    while t != 0:
        out.append(int(t))
        t = parents[t]
    return out

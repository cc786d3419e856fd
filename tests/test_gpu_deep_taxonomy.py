"""resolveTree on taxonomies shaped like NCBI's -- lineages tens of nodes deep, most nodes without a sibling that matters to a
read, several trees (ids whose parent is NONE besides ROOT) -- and on reads that hit many taxa: on one root path, on side
branches, tied scores whose LCA is a node no record names.  The lane kernel answers ancestor questions from an Euler tour of
the taxonomy (engine.h: FusedArgs.nodes) and jumps its confidence walk from one map taxon to the next; the oracle walks parent
pointers one by one as LowestCommonAncestor.scala:49-146 does.  Every output bit for bit, several thresholds."""
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

THRESHOLDS = (0.0, 0.05, 0.15, 0.3, 0.6, 1.0)


def deep_forest(rng, levels=6, fan=2, chain_max=4, extra_trees=1):
    """parents[] of a tree with `fan`-way branching on `levels` levels and 0..chain_max unary nodes above every branching node,
    ids shuffled (a child's id may be smaller than its parent's); extra_trees more such trees hang from ids whose parent is 0."""
    parents = [0, 0]
    tops = [1]
    for _ in range(extra_trees):
        parents.append(0)                      # the root of a detached tree
        tops.append(len(parents) - 1)
    branching, leaves = [], []
    for top in tops:
        level = [top]
        for d in range(levels if top == 1 else max(2, levels - 3)):
            nxt = []
            for p in level:
                for _ in range(fan):
                    up = p
                    for _ in range(int(rng.integers(0, chain_max + 1))):
                        parents.append(up)
                        up = len(parents) - 1
                    parents.append(up)
                    nxt.append(len(parents) - 1)
                    branching.append(len(parents) - 1)
            level = nxt
        leaves += level
    parents = np.array(parents, np.int64)
    n = len(parents)
    perm = np.concatenate([[0, 1], 2 + rng.permutation(n - 2)])     # old id -> new id; NONE and ROOT keep theirs
    out = np.zeros(n, np.int32)
    out[perm] = perm[parents]
    return out, [int(perm[x]) for x in branching], [int(perm[x]) for x in leaves], [int(perm[x]) for x in tops]


def evolve_genomes(rng, parents, leaves, tops, genome_len, rate):
    """a genome per leaf: the genome of the tree's top with `rate` substitutions per node of the leaf's root path"""
    memo = {}

    def genome(t):
        if t in memo:
            return memo[t]
        if t in tops:
            g = synth.random_dna(genome_len, rng)
        else:
            g = genome(int(parents[t])).copy()
            sub = rng.random(genome_len) < rate
            g[sub] = synth.random_dna(int(sub.sum()), rng)
        memo[t] = g
        return g

    return [genome(t) for t in leaves]


@pytest.fixture(scope="module", params=[(0, 1), (4, 1), (9, 2)], ids=["flat", "chains", "long-chains-forest"])
def deep_world(orc, request):
    import slacken_amd
    chain_max, extra = request.param
    rng = np.random.default_rng(4000 + chain_max)
    p = orc.params()
    parents, branching, leaves, tops = deep_forest(rng, levels=6, fan=2, chain_max=chain_max, extra_trees=extra)
    genomes = evolve_genomes(rng, parents, leaves, tops, 6000, 0.006)
    bases, offsets = synth.pack(genomes)
    keys, taxa = orc.build_records(p, parents, bases, offsets, leaves)        # (key, LCA taxon) records at every branching level
    assert len(set(taxa.tolist())) > len(leaves) // 2
    ix = slacken_amd.Index(expected_records=len(keys), max_taxon=len(parents) - 1)
    ix.append(keys, taxa)
    ix.set_taxonomy(parents)
    ix.finalize()
    depth = max(len(path_to_root(parents, t)) for t in leaves)
    return dict(p=p, parents=parents, genomes=genomes, leaves=leaves, st=ix.stream(), oix=orc.Index(1, keys, taxa), ix=ix, depth=depth)


def path_to_root(parents, t):
    out = []
    while t != 0:
        out.append(t)
        t = int(parents[t])
    return out


def chimeras(world, rng, n, length=150, pieces=(1, 2, 3, 5)):
    reads = []
    G = world["genomes"]
    for _ in range(n):
        k = int(rng.choice(pieces))
        cuts = np.sort(rng.integers(0, length, k - 1)) if k > 1 else np.array([], np.int64)
        bounds = np.concatenate([[0], cuts, [length]])
        parts = []
        a = int(rng.integers(0, len(G[0]) - length))      # the same locus of several genomes: homologous minimizers, tied scores
        for i in range(k):
            g = G[int(rng.integers(0, len(G)))]
            parts.append(g[a + bounds[i]:a + bounds[i + 1]])
        r = np.concatenate(parts).copy()
        if rng.random() < 0.3:
            r = synth.revcomp(r).copy()
        reads.append(r)
    return reads


def check(orc, world, reads, mates=None, min_hit_groups=2):
    bases, offsets = synth.pack(reads)
    mb = mo = None
    if mates is not None:
        mb, mo = synth.pack(mates)
    got = world["st"].classify_batch(bases, offsets, mb, mo, thresholds=THRESHOLDS, min_hit_groups=min_hit_groups, with_hits=False,
                                     with_num_hits=True)
    want = orc.classify_batch(world["p"], world["oix"], world["parents"], bases, offsets, mb, mo, min_hit_groups=min_hit_groups,
                              thresholds=THRESHOLDS)
    for key in ("total_kmers", "num_hits", "num_distinct", "taxon", "classified"):
        bad = np.nonzero(np.atleast_2d(got[key] != want[key]).any(axis=0))[0]
        assert bad.size == 0, (key, bad[:5].tolist(), np.atleast_2d(got[key])[:, bad[:5]].tolist(), np.atleast_2d(want[key])[:, bad[:5]].tolist())
    return got


def test_short_reads_hitting_many_taxa(orc, deep_world):
    rng = np.random.default_rng(1)
    reads = chimeras(deep_world, rng, 6000)
    reads += chimeras(deep_world, rng, 1500, length=400, pieces=(2, 4, 8))     # more taxa per read, some past the 12-slot map
    got = check(orc, deep_world, reads)
    assert deep_world["st"].last_deferred() < len(reads) // 2                  # it IS the lane kernel that answered
    # the calls really sit at many levels of the tree, the walk really moves: thresholds change them
    assert len(set(got["taxon"][0].tolist())) > 20
    assert (got["taxon"][0] != got["taxon"][3]).mean() > 0.05


def test_pairs_and_long_reads(orc, deep_world):
    rng = np.random.default_rng(2)
    reads = chimeras(deep_world, rng, 1500)
    mates = chimeras(deep_world, rng, 1500, length=120)
    check(orc, deep_world, reads, mates, min_hit_groups=1)
    long_reads = chimeras(deep_world, rng, 300, length=1800, pieces=(3, 6))    # the lane kernel's long variant
    long_reads += chimeras(deep_world, rng, 40, length=5600, pieces=(4, 9))    # segment kernel / wave kernel
    check(orc, deep_world, long_reads)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SLK_DEEP_SEEDS", 3))))   # (SLK_DEEP_SEEDS: soak runs)
def test_random_shapes(orc, seed):
    """taxonomy shape, library, reads, thresholds and minHitGroups drawn at random"""
    import slacken_amd
    rng = np.random.default_rng(7000 + seed)
    p = orc.params()
    parents, branching, leaves, tops = deep_forest(rng, levels=int(rng.integers(3, 7)), fan=int(rng.integers(2, 4)),
                                                   chain_max=int(rng.integers(0, 12)), extra_trees=int(rng.integers(0, 3)))
    if len(leaves) > 160:
        keep = rng.choice(len(leaves), 160, replace=False)
        leaves = [leaves[i] for i in keep]
    genomes = evolve_genomes(rng, parents, leaves, tops, int(rng.integers(2500, 5000)), float(rng.choice([0.002, 0.006, 0.02])))
    bases, offsets = synth.pack(genomes)
    keys, taxa = orc.build_records(p, parents, bases, offsets, leaves)
    ix = slacken_amd.Index(expected_records=len(keys), max_taxon=len(parents) - 1)
    ix.append(keys, taxa)
    ix.set_taxonomy(parents)
    ix.finalize()
    world = dict(p=p, parents=parents, genomes=genomes, leaves=leaves, st=ix.stream(), oix=orc.Index(1, keys, taxa), ix=ix)
    reads = chimeras(world, rng, 2500, length=int(rng.integers(60, 260)), pieces=(1, 2, 3, 4, 6))
    thr = tuple(sorted(float(x) for x in rng.choice([0.0, 0.01, 0.05, 0.1, 0.15, 0.25, 0.4, 0.7, 1.0], size=4, replace=False)))
    bases, offsets = synth.pack(reads)
    mhg = int(rng.integers(1, 4))
    got = world["st"].classify_batch(bases, offsets, thresholds=thr, min_hit_groups=mhg, with_hits=False, with_num_hits=True)
    want = orc.classify_batch(p, world["oix"], parents, bases, offsets, None, None, min_hit_groups=mhg, thresholds=thr)
    for key in ("total_kmers", "num_hits", "num_distinct", "taxon", "classified"):
        assert np.array_equal(got[key], want[key]), (seed, key)


def test_classify_hits_entry_on_deep_taxonomies(orc, deep_world):
    """slk_classify_hits (kernel 3 alone, the staged classify kernel: Euler-tour intervals in the caller's ids) on hit lists whose
    taxa sit on one lineage, on side branches and in another tree, NONE among them, against the oracle's walks."""
    import slacken_amd
    rng = np.random.default_rng(17)
    parents = deep_world["parents"]
    leaves = deep_world["leaves"]
    lists, flags = [], []
    for r in range(1500):
        kind = rng.random()
        pool = []
        for _ in range(int(rng.integers(1, 4))):                       # one to three lineages
            path = path_to_root(parents, int(rng.choice(leaves)))
            pool += [path[i] for i in sorted(rng.choice(len(path), size=min(len(path), int(rng.integers(1, 6))), replace=False))]
        if kind < 0.1:
            pool = pool[:1]
        hits, dis = [], []
        for _ in range(int(rng.integers(1, 60))):
            u = rng.random()
            if u < 0.05:
                hits.append((-1, int(rng.integers(1, 40)))); dis.append(0)
            elif u < 0.4:
                hits.append((0, int(rng.integers(1, 8)))); dis.append(int(rng.random() < 0.7))
            else:
                hits.append((int(rng.choice(pool)), int(rng.integers(1, 6)))); dis.append(int(rng.random() < 0.7))
        lists.append(hits)
        flags.append(dis)
    offs = np.cumsum([0] + [len(h) for h in lists]).astype(np.uint64)
    flat = np.array([h for hs in lists for h in hs], slacken_amd.capi.HIT_DTYPE)
    dflat = np.array([d for ds in flags for d in ds], np.uint8)
    got = deep_world["st"].classify_hits(offs, flat, dflat, min_hit_groups=1, thresholds=THRESHOLDS)
    for c, t in enumerate(THRESHOLDS):
        for r, (hits, dis) in enumerate(zip(lists, flags)):
            want = orc.classify_hits(parents, hits, dis, 1, t)
            assert (int(got["taxon"][c, r]), bool(got["classified"][c, r])) == (want["taxon"], want["classified"]), (t, r, hits)

"""resolveTree / LCA against the reference's own independent spec `correctClassification`
(T/slacken/LowestCommonAncestorProps.scala:66-91), restated in Python, on the reference's taxonomy generator."""
import math

import numpy as np
from hypothesis import given, settings, strategies as st

import taxgen

SET = settings(max_examples=120, deadline=None)


def has_ancestor(parents, tax, anc):  # Taxonomy.hasAncestor :236-244
    return anc in taxgen.path_to_root(parents, tax)


def py_lca(parents, a, b):  # LowestCommonAncestor.apply :49-78
    if a == 0 or b == 0:
        return a if b == 0 else b
    pa = taxgen.path_to_root(parents, a)
    for x in taxgen.path_to_root(parents, b):
        if x in pa:
            return x
    return 1


def correct_classification(parents, hits, threshold):
    """hits: list of (taxon, count). LowestCommonAncestorProps.correctClassification :66-91."""
    total = sum(c for _, c in hits)
    distinct = []
    for t, _ in hits:
        if t != 0 and t not in distinct:
            distinct.append(t)

    def frac_above(t):
        return 0 if not hits else sum(c for h, c in hits if has_ancestor(parents, t, h)) / total

    def frac_below(t):
        return 0 if not hits else sum(c for h, c in hits if has_ancestor(parents, h, t)) / total

    best = sorted(((frac_above(t), t) for t in distinct), key=lambda x: x[0], reverse=True)
    if not best:
        return 0
    best_frac, best_taxon = best[0]
    for f, t in best[1:]:
        if f != best_frac:
            break
        best_taxon = py_lca(parents, best_taxon, t)
    for tax in taxgen.path_to_root(parents, best_taxon):
        if not frac_below(tax) < threshold:
            return tax
    return 0


def to_map(hits):  # TaxonCounts.fromHits + toMap (insertion ordered)
    taxa, counts = [], []
    for t, c in hits:
        if t in (-1, -2):
            continue
        if t in taxa:
            counts[taxa.index(t)] += c
        else:
            taxa.append(t)
            counts.append(c)
    return taxa, counts


@st.composite
def tax_and_hits(draw):
    seed = draw(st.integers(0, 2**31))
    rng = np.random.default_rng(seed)
    parents = taxgen.taxonomy(draw(st.sampled_from([8, 30, 100])), rng)
    taxa = taxgen.defined_taxa(parents)
    kmers = draw(st.integers(10, 200))
    invalid = int(math.floor(kmers * draw(st.floats(0, 1))))
    hits = []
    for pool, total in ((taxa, kmers - invalid), ([0], invalid)):  # Testing.pseudoRead / readHits :35-46
        while total > 0:
            c = min(total, int(rng.integers(1, 11)))
            hits.append((int(pool[rng.integers(0, len(pool))]), c))
            total -= c
    order = rng.permutation(len(hits))
    return parents, [hits[i] for i in order], draw(st.floats(0, 1))


@SET
@given(tax_and_hits())
def test_resolve_tree_matches_reference_spec(orc, c):  # LowestCommonAncestorProps "resolveTree" :93-107
    parents, hits, threshold = c
    taxa, counts = to_map(hits)
    total = sum(cnt for _, cnt in hits)
    got = orc.resolve_tree(parents, taxa, counts, math.ceil(threshold * total))
    assert got == correct_classification(parents, hits, threshold)


@SET
@given(st.integers(0, 2**31))
def test_lca_props(orc, seed):  # TaxonomyProps.scala LCA invariants
    rng = np.random.default_rng(seed)
    parents = taxgen.taxonomy(60, rng)
    taxa = taxgen.defined_taxa(parents)
    for _ in range(20):
        a, b, c = (int(taxa[i]) for i in rng.integers(0, len(taxa), 3))
        l = orc.lca(parents, a, b)
        assert l == py_lca(parents, a, b) == orc.lca(parents, b, a)
        assert has_ancestor(parents, a, l) and has_ancestor(parents, b, l)
        assert orc.lca(parents, a, 0) == a and orc.lca(parents, 0, b) == b and orc.lca(parents, a, a) == a
        assert orc.lca(parents, orc.lca(parents, a, b), c) == orc.lca(parents, a, orc.lca(parents, b, c))


def test_resolve_tree_edge_cases(orc):
    #        1
    #      2   3
    #     4 5   6      7 is undefined (parent NONE)
    parents = np.array([0, 0, 1, 1, 2, 2, 3, 0], np.int32)
    rt = lambda taxa, counts, req: orc.resolve_tree(parents, taxa, counts, req)
    assert rt([], [], 0.0) == 0
    assert rt([0], [50], 0.0) == 0                 # only NONE hits
    assert rt([4, 5], [3, 3], 0.0) == 2            # tie -> LCA
    assert rt([4, 5, 6], [3, 3, 3], 0.0) == 1      # three-way tie across clades -> root
    assert rt([4, 2], [3, 1], 0.0) == 4            # path score 4 (4+2) beats 2's own 1
    assert rt([4, 5, 0], [3, 2, 5], 4.0) == 2      # lifted once: clade(2) = 5 >= 4
    assert rt([4, 5, 0], [3, 2, 5], 6.0) == 0      # runs off the root
    assert rt([4, 6], [3, 2], 5.0) == 1            # root clade covers both
    assert rt([7, 4], [5, 1], 0.0) == 7            # undefined taxon: path is itself only
    assert rt([7, 4], [2, 2], 0.0) == 1            # tie with a taxon outside the tree -> ROOT
    assert math.ceil(0.07 * 100) == 8              # binary64 product 7.000000000000001: required score is 8, not 7
    assert rt([4, 0], [7, 93], math.ceil(0.07 * 100)) == 0 and rt([4, 0], [8, 92], math.ceil(0.07 * 100)) == 4

"""Device library construction (slk_index_add_sequences, build.hip) against the oracle's restatement of
KeyValueIndex.makeRecords, bit for bit on the exported records, and classification with the built index."""
import numpy as np
import pytest

import synth
import taxgen
from test_oracle_build import messy_genome

pytestmark = pytest.mark.gpu


def pack(seqs):
    bases = np.frombuffer("".join(seqs).encode(), np.uint8)
    offsets = np.zeros(len(seqs) + 1, np.uint64)
    np.cumsum([len(s) for s in seqs], out=offsets[1:])
    return bases, offsets


def genomes(rng, parents, n, lo, hi, shared=400):
    taxa = np.array(taxgen.defined_taxa(parents))
    base = messy_genome(rng, 4000)
    seqs, tx = [], []
    for _ in range(n):
        g = list(messy_genome(rng, int(rng.integers(lo, hi))))
        a = int(rng.integers(0, 3000))
        cut = min(shared, len(g))
        g[:cut] = base[a:a + cut]
        seqs.append("".join(g))
        tx.append(int(taxa[rng.integers(1, len(taxa))]))
    return seqs, tx


def _build_param_sets():
    sets = [(35, 31, 7), (31, 12, 0), (35, 31, 0), (40, 32, 4)]
    rng = np.random.default_rng(4242)
    for _ in range(int(__import__("os").environ.get("SLK_FUZZ_SEEDS", 0)) // 4):   # soak: random splitters
        m = int(rng.integers(8, 33))
        sets.append((int(rng.integers(m, m + 28)), m, int(rng.integers(0, m // 2 + 1))))
    return sets


@pytest.mark.parametrize("k,m,spaces", _build_param_sets())
def test_build_matches_oracle(orc, k, m, spaces):
    import slacken_amd
    rng = np.random.default_rng(k * 100 + m)
    parents = taxgen.taxonomy(8 * 16, rng)
    p = orc.params(k=k, m=m, spaces=spaces)
    seqs, tx = genomes(rng, parents, 40, 10, 6000)
    seqs += ["", "ACGT", "N" * 100, synth.random_dna(k, rng).tobytes().decode(), synth.random_dna(k - 1, rng).tobytes().decode()]
    tx += [5, 5, 5, 5, 5]
    seqs.append(synth.random_dna(3000, rng).tobytes().decode())     # taxon NONE: skipped
    tx.append(0)
    bases, offsets = pack(seqs)
    keep = [i for i, t in enumerate(tx) if t != 0]
    wb, wo = pack([seqs[i] for i in keep])
    want_k, want_t = orc.build_records(p, parents, wb, wo, [tx[i] for i in keep])

    def build(parts):
        ix = slacken_amd.Index(k=k, m=m, spaces=spaces, expected_records=int(offsets[-1]) + 1000, max_taxon=len(parents) - 1)
        ix.set_taxonomy(parents)
        for idx in parts:
            b, o = pack([seqs[i] for i in idx])
            ix.add_sequences(b, o, [tx[i] for i in idx])
        return ix

    ix = build([list(range(len(seqs)))])
    got_k, got_t = ix.export()
    assert np.array_equal(got_k, want_k) and np.array_equal(got_t, want_t)
    assert ix.info().records == len(want_k)
    # any batching / order of the calls gives the same records
    perm = rng.permutation(len(seqs)).tolist()
    ix2 = build([perm[:7], perm[7:30], perm[30:]])
    k2, t2 = ix2.export()
    assert np.array_equal(k2, want_k) and np.array_equal(t2, want_t)
    ix.finalize()
    assert np.array_equal(ix.lookup(want_k), want_t)


def test_long_sequences_many_chunks(orc):
    import slacken_amd
    rng = np.random.default_rng(77)
    parents = taxgen.taxonomy(8 * 8, rng)
    p = orc.params()
    seqs = [synth.random_dna(300_000, rng).tobytes().decode(), messy_genome(rng, 150_000)]
    seqs.append(seqs[0][100_000:180_000])      # a second taxon shares 80 kbp with the first
    tx = [10, 20, 30]
    bases, offsets = pack(seqs)
    want_k, want_t = orc.build_records(p, parents, bases, offsets, tx)
    ix = slacken_amd.Index(expected_records=len(want_k) * 2, max_taxon=len(parents) - 1)
    ix.set_taxonomy(parents)
    ix.add_sequences(bases, offsets, tx)
    got_k, got_t = ix.export()
    assert np.array_equal(got_k, want_k) and np.array_equal(got_t, want_t)
    lca = orc.lca(parents, 10, 30)
    assert (got_t == lca).sum() > 10_000


def test_built_index_classifies_like_oracle(orc):
    """two-step shape: records built on the device stay in HBM and are classified against directly"""
    import slacken_amd
    rng = np.random.default_rng(5)
    parents = taxgen.taxonomy(8 * 32, rng)
    p = orc.params()
    lib = synth.Library(orc, p, parents, n_genomes=8, genome_len=20000)
    seqs = [g.tobytes().decode() for g in lib.genomes]
    bases, offsets = pack(seqs)
    ix = slacken_amd.Index(expected_records=len(lib.keys) * 2, max_taxon=len(parents) - 1)
    ix.set_taxonomy(parents)
    ix.add_sequences(bases, offsets, lib.genome_taxa)
    got_k, got_t = ix.export()
    order = np.argsort(lib.keys)
    assert np.array_equal(got_k, lib.keys[order]) and np.array_equal(got_t, lib.taxa[order])
    ix.finalize()
    reads = synth.make_reads(lib, 3000, rng)
    rb, ro = synth.pack(reads)
    want = orc.classify_batch(p, orc.Index(1, lib.keys, lib.taxa), parents, rb, ro, thresholds=(0.0, 0.15))
    got = ix.stream().classify_batch(rb, ro, thresholds=(0.0, 0.15), with_hits=False)
    for key in ("taxon", "classified", "num_distinct", "total_kmers"):
        assert np.array_equal(got[key], want[key])


def test_append_then_add_sequences_and_errors(orc):
    import slacken_amd
    rng = np.random.default_rng(6)
    parents = taxgen.taxonomy(8 * 8, rng)
    p = orc.params()
    g = synth.random_dna(4000, rng).tobytes().decode()
    bases, offsets = pack([g])
    ix = slacken_amd.Index(expected_records=5000, max_taxon=len(parents) - 1)
    with pytest.raises(slacken_amd.SlackenError):          # needs the taxonomy
        ix.add_sequences(bases, offsets, [3])
    ix.set_taxonomy(parents)
    with pytest.raises(slacken_amd.SlackenError):          # taxon out of range
        ix.add_sequences(bases, offsets, [len(parents) + 100000])
    ix.append([12345 << 16], [7])
    ix.add_sequences(bases, offsets, [3])
    k1, t1 = ix.export()
    wk, wt = orc.build_records(p, parents, bases, offsets, [3])
    assert sorted(k1.tolist()) == sorted(wk.tolist() + [12345 << 16])
    ix.finalize()
    with pytest.raises(slacken_amd.SlackenError):
        ix.add_sequences(bases, offsets, [3])


@pytest.mark.parametrize("seed", range(max(6, int(__import__("os").environ.get("SLK_FUZZ_SEEDS", 0)) // 4)))
def test_random_genomes_classify_to_ancestor_or_self(orc, seed):
    """The reference's end-to-end property (T/slacken/ClassifierTest.scala:75-124): ~100 random genomes (1-10 kb) on leaf
    taxa, library built from them (here: on the device), 1000 simulated 200 bp reads, minHitGroups = 1, confidence 0; random
    k, m, spaces: every CLASSIFIED read's taxon is the true taxon or one of its ancestors.  (m is limited to the engine's
    32; the device results are additionally compared with the oracle.)"""
    import slacken_amd
    rng = np.random.default_rng(1000 + seed)
    m = int(rng.integers(15, 33))
    k = int(rng.integers(m, m + 28))
    spaces = int(rng.integers(0, m // 2 + 1))
    parents = taxgen.taxonomy(100 * 8, rng)
    taxa = np.array(taxgen.defined_taxa(parents))
    leaves = np.setdiff1d(taxa, parents[taxa])
    leaves = leaves[leaves != 1]
    gs = [synth.random_dna(int(rng.integers(1000, 10000)), rng) for _ in leaves]
    bases = np.concatenate(gs)
    offsets = np.zeros(len(gs) + 1, np.uint64)
    np.cumsum([len(g) for g in gs], out=offsets[1:])
    ix = slacken_amd.Index(k=k, m=m, spaces=spaces, expected_records=int(offsets[-1]), max_taxon=len(parents) - 1)
    ix.set_taxonomy(parents)
    ix.add_sequences(bases, offsets, leaves.astype(np.int32))
    rec_k, rec_t = ix.export()
    ix.finalize()
    reads, truth = [], []
    for _ in range(1000):
        g = int(rng.integers(0, len(gs)))
        a = int(rng.integers(0, len(gs[g]) - 200))
        reads.append(gs[g][a:a + 200])
        truth.append(int(leaves[g]))
    rb, ro = synth.pack(reads)
    got = ix.stream().classify_batch(rb, ro, thresholds=(0.0,), min_hit_groups=1, with_hits=False)
    assert got["classified"][0].sum() > 900
    for r in np.nonzero(got["classified"][0])[0]:
        assert int(got["taxon"][0][r]) in taxgen.path_to_root(parents, truth[r]), (k, m, spaces, r)
    p = orc.params(k=k, m=m, spaces=spaces)
    want = orc.classify_batch(p, orc.Index(1, rec_k, rec_t), parents, rb, ro, thresholds=(0.0,), min_hit_groups=1)
    assert np.array_equal(got["taxon"], want["taxon"]) and np.array_equal(got["classified"], want["classified"])


def test_real_contig_with_long_n_runs(orc):
    """The reference's testData/Akashinriki_10k.fasta (one 599 940-base barley contig, 112 822 of them N in long runs; kept as
    data under tests/golden): device library construction equals the oracle's, and the contig classifies identically as ONE
    fragment (wave-per-read path, ~150 000 spans incl. ambiguous ones of thousands of k-mers) and cut into pieces."""
    import gzip
    import os
    import hostmodel
    import slacken_amd
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    (_, contig), = hostmodel.parse_fasta(gzip.open(os.path.join(gold, "akashinriki_10k.fasta.gz"), "rt").read())
    assert len(contig) == 599940 and contig.count("N") == 112822
    rng = np.random.default_rng(31)
    parents = taxgen.taxonomy(8 * 8, rng)
    p = orc.params()
    other = bytearray(contig[100_000:400_000].encode())          # a relative: a stretch of the contig with 2 % substitutions
    for i in rng.choice(len(other), len(other) // 50, replace=False):
        other[i] = b"ACGT"[rng.integers(0, 4)]
    seqs = [contig, other.decode()]
    tx = [10, 20]
    bases, offsets = pack(seqs)
    want_k, want_t = orc.build_records(p, parents, bases, offsets, tx)
    ix = slacken_amd.Index(expected_records=len(want_k) * 2, max_taxon=len(parents) - 1)
    ix.set_taxonomy(parents)
    ix.add_sequences(bases, offsets, tx)
    got_k, got_t = ix.export()
    assert np.array_equal(got_k, want_k) and np.array_equal(got_t, want_t)
    ix.finalize()
    oix = orc.Index(1, want_k, want_t)
    st = ix.stream()
    cuts = np.sort(rng.choice(len(contig), 400, replace=False))
    pieces = [contig[a:b] for a, b in zip(np.concatenate([[0], cuts]), np.concatenate([cuts, [len(contig)]]))]
    frags = [contig] + pieces + [contig[200_000:260_000].lower()]
    fb, fo = pack(frags)
    want = orc.classify_batch(p, oix, parents, fb, fo, thresholds=(0.0, 0.3))
    for with_hits in (True, False):
        got = st.classify_batch(fb, fo, thresholds=(0.0, 0.3), with_hits=with_hits)
        for key in ("taxon", "classified", "num_distinct", "total_kmers"):
            assert np.array_equal(got[key], want[key]), (key, with_hits)
    full = st.classify_batch(fb, fo, thresholds=(0.0,), with_hits=True)
    assert np.array_equal(full["num_hits"], want["num_hits"]) and want["num_hits"][0] > 100_000
    ho = full["hit_offsets"].astype(np.int64)
    for i in (0, 1, 57, len(frags) - 1):
        _, hits = orc.classify_read(p, oix, parents, frags[i], None, 2, 0.0)
        g = full["hits"][ho[i]:ho[i + 1]]
        assert [(int(t), int(c)) for t, c in zip(g["taxon"], g["count"])] == hits
        if i == 0:
            assert any(t == -1 and c > 1000 for t, c in hits)       # an ambiguous span of thousands of k-mers

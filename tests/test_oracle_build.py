"""Library construction in the oracle (orc_library_minimizers / orc_build_records: KeyValueIndex.makeRecords,
S/slacken/KeyValueIndex.scala:85-93) against the pure-Python scan model and a dict-based LCA merge, plus the properties the
device builder relies on.  Reference pins available offline: '#records == #distinct minimizers'
(T/slacken/KeyValueIndexTest.scala:35-60, restated below on synthetic genomes -- the reference's tiny genome file is not in the
mount); everything else here is restated semantics: parity unpinned."""
import numpy as np

import pymodel
import synth
import taxgen

K, M, S = 35, 31, 7


def py_keys(p, seq):
    out = []
    for key, _, _ in pymodel.supermers(seq, K, M, S, 0xe37e28c4271b5a2d, True):
        v = pymodel.left_align(key, M)[0]
        out.append(v - (1 << 64) if v >= (1 << 63) else v)
    return out


def messy_genome(rng, n):
    g = synth.random_dna(n, rng).tobytes().decode()
    g = list(g)
    for _ in range(max(1, n // 400)):      # single Ns, N runs, IUPAC codes, lower case
        a = int(rng.integers(0, n))
        g[a] = "NRYKM"[int(rng.integers(0, 5))]
    a = int(rng.integers(0, max(1, n - 60)))
    g[a:a + 50] = "N" * min(50, n - a)
    for _ in range(n // 50):
        a = int(rng.integers(0, n))
        g[a] = g[a].lower()
    return "".join(g)


def test_library_minimizers_split_around_invalid(orc):
    rng = np.random.default_rng(11)
    p = orc.params()
    for n in (10, 34, 35, 36, 200, 1500):
        g = messy_genome(rng, n)
        want = []
        import re
        for mt in re.finditer("[ACTGUactgu][ACTGUactgu\n\r]*", g):   # InputReader.removeInvalid
            want += py_keys(p, mt.group(0).upper())
        assert orc.library_minimizers(p, g).tolist() == want
    # whitespace inside a fragment is skipped, not a separator (multi-line FASTA records, InputReader.scala:57-58)
    g = synth.random_dna(300, rng).tobytes().decode()
    assert orc.library_minimizers(p, g[:100] + "\n" + g[100:217] + "\r\n" + g[217:]).tolist() == py_keys(p, g)


def test_chunking_preserves_the_minimizer_set(orc):
    """Chunks of CW k-mer windows overlapping by k-1 bases (what build.hip scans per lane, and what the reference's indexed
    FASTA reader produces) have the same minimizer SET as the whole sequence."""
    rng = np.random.default_rng(12)
    p = orc.params()
    g = synth.random_dna(5000, rng).tobytes().decode()
    whole = set(orc.library_minimizers(p, g).tolist())
    for cw in (1, 7, 64, 512):
        got = set()
        for w0 in range(0, len(g) - K + 1, cw):
            got |= set(orc.library_minimizers(p, g[w0:w0 + cw + K - 1]).tolist())
        assert got == whole


def test_build_records_lca_merge(orc):
    rng = np.random.default_rng(13)
    p = orc.params()
    parents = taxgen.taxonomy(8 * 16, rng)
    taxa = np.array(taxgen.defined_taxa(parents))
    seqs, tx = [], []
    base = messy_genome(rng, 3000)
    for i in range(12):
        g = list(messy_genome(rng, int(rng.integers(20, 2500))))
        a = int(rng.integers(0, 2000))
        g[:400] = base[a:a + 400][:len(g[:400])]      # shared stretches => keys with several taxa
        seqs.append("".join(g))
        tx.append(int(taxa[rng.integers(1, len(taxa))]))
    bases = np.frombuffer("".join(seqs).encode(), np.uint8)
    offsets = np.zeros(len(seqs) + 1, np.uint64)
    np.cumsum([len(s) for s in seqs], out=offsets[1:])
    keys, rt = orc.build_records(p, parents, bases, offsets, tx)
    want = {}
    for s, t in zip(seqs, tx):
        for key in orc.library_minimizers(p, s).tolist():
            want[key] = orc.lca(parents, want.get(key, 0), t)
    assert keys.tolist() == sorted(want) and rt.tolist() == [want[k] for k in keys.tolist()]
    # '#records == #distinct minimizers' (KeyValueIndexTest.scala:35-60)
    assert len(keys) == len({k for s in seqs for k in orc.library_minimizers(p, s).tolist()})
    assert any(t not in tx for t in rt.tolist())       # some records really are merged above the input taxa
    # order independence (what lets the device merge with atomics)
    perm = rng.permutation(len(seqs))
    b2 = np.frombuffer("".join(seqs[i] for i in perm).encode(), np.uint8)
    o2 = np.zeros(len(seqs) + 1, np.uint64)
    np.cumsum([len(seqs[i]) for i in perm], out=o2[1:])
    k2, t2 = orc.build_records(p, parents, b2, o2, [tx[i] for i in perm])
    assert np.array_equal(k2, keys) and np.array_equal(t2, rt)


import pytest


@pytest.mark.parametrize("k,m,spaces,canonical,xor", [(35, 31, 7, True, 0xe37e28c4271b5a2d), (15, 15, 0, True, 0), (40, 22, 5, False, 0x1234567),
                                                       (61, 32, 10, True, 0xe37e28c4271b5a2d), (28, 17, 3, True, 0xdeadbeefcafe)])
def test_reads_classify_to_their_genome_or_an_ancestor(orc, k, m, spaces, canonical, xor):
    """The reference's one end-to-end property of the classify path, ClassifierTest.scala:74-117 ("Classify with random genomes"):
    random genomes of 1 000 .. 10 000 bases at the leaves of a generated taxonomy (Testing.taxonomies(800)), the library
    makeRecords builds from them, 1 000 reads of 200 bases cut from them, minHitGroups = 1, confidence 0: every read that is
    classified is classified to its genome's taxon or one of its ancestors.  Here on the oracle; tests/test_gpu_build.py holds the
    engine to the same."""
    rng = np.random.default_rng(k * 1000 + m)
    parents = taxgen.taxonomy(100 * 8, rng)
    has_child = np.zeros(len(parents), bool)
    has_child[parents[parents > 0]] = True
    leaves = [t for t in taxgen.defined_taxa(parents) if not has_child[t] and t != 1]
    genomes = [synth.random_dna(int(rng.integers(1000, 10001)), rng) for _ in leaves]
    bases = np.concatenate(genomes)
    offsets = np.cumsum([0] + [len(g) for g in genomes]).astype(np.uint64)
    p = orc.params(k=k, m=m, spaces=spaces, xor_mask=xor, canonical=canonical)
    keys, tx = orc.build_records(p, parents, bases, offsets, np.array(leaves, np.int32))
    oix = orc.Index(1, keys, tx)
    which = rng.integers(0, len(genomes), 1000)
    reads = []
    for g in which:
        a = int(rng.integers(0, len(genomes[g]) - 200))
        reads.append(genomes[g][a:a + 200])
    rb, ro = synth.pack(reads)
    res = orc.classify_batch(p, oix, parents, rb, ro, min_hit_groups=1, thresholds=(0.0,))
    wrong = [i for i in range(1000) if res["classified"][0][i] and int(res["taxon"][0][i]) not in taxgen.path_to_root(parents, leaves[which[i]])]
    assert not wrong, wrong[:5]
    assert res["classified"][0].mean() > 0.95       # (not in the reference's test: a read of 200 bases of its genome does classify)

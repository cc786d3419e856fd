"""Golden vectors under tests/golden/ (inputs: reads from the reference's own testData; expected values: SELF-GENERATED
by the CPU oracle -- see tests/golden/make_golden.py for why no reference output exists).  The CPU test pins the oracle
against regressions; the GPU test checks the engine, through the C ABI, against the same vectors incl. the Kraken-style
output lines (ClassifiedRead.outputLine, Classifier.scala:41-44)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    g = json.load(open(os.path.join(GOLD, "golden_classify.json")))
    reads = [line.rstrip("\n").split("\t") for line in open(os.path.join(GOLD, "reads.tsv"))]
    lib = np.load(os.path.join(GOLD, "library.npz"))
    return g, reads, lib


def test_oracle_reproduces_golden(orc):
    g, reads, lib = load()
    p = orc.params(k=g["k"], m=g["m"], spaces=g["spaces"])
    oix = orc.Index(1, lib["keys"], lib["taxa"])
    assert [t for t, _ in reads] == [r["title"] for r in g["reads"]]
    for (title, seq), want in zip(reads, g["reads"]):
        for thr in g["thresholds"]:
            res, hits = orc.classify_read(p, oix, lib["parents"], seq, None, g["min_hit_groups"], thr)
            assert [res["taxon"], int(res["classified"])] == want[f"c{thr}"]
        assert [list(h) for h in hits] == want["hits"]
        assert (res["num_distinct"], res["total_kmers"]) == (want["num_distinct"], want["total_kmers"])
        if hits:
            assert orc.output_line(bool(want["c0.0"][1]), title, want["c0.0"][0], hits, g["k"]) == want["line"]
    for i, want in zip(range(0, 300, 2), g["pairs"]):
        res, hits = orc.classify_read(p, oix, lib["parents"], reads[i][1], reads[i + 1][1], 2, 0.0)
        assert orc.output_line(bool(res["classified"]), reads[i][0], res["taxon"], hits, g["k"]) == want["line"]


@pytest.mark.gpu
def test_engine_matches_golden(orc):
    import slacken_amd
    g, reads, lib = load()
    ix = slacken_amd.Index(k=g["k"], m=g["m"], spaces=g["spaces"], expected_records=len(lib["keys"]),
                           max_taxon=len(lib["parents"]) - 1)
    ix.append(lib["keys"], lib["taxa"])
    ix.set_taxonomy(lib["parents"])
    ix.finalize()
    st = ix.stream()
    seqs = [np.frombuffer(s.encode(), np.uint8) for _, s in reads]
    offsets = np.zeros(len(seqs) + 1, np.uint64)
    np.cumsum([len(s) for s in seqs], out=offsets[1:])
    bases = np.concatenate(seqs)
    for with_hits in (True, False):
        got = st.classify_batch(bases, offsets, thresholds=g["thresholds"], min_hit_groups=g["min_hit_groups"],
                                with_hits=with_hits)
        for ci, thr in enumerate(g["thresholds"]):
            want = np.array([r[f"c{thr}"] for r in g["reads"]])
            assert np.array_equal(got["taxon"][ci], want[:, 0]) and np.array_equal(got["classified"][ci], want[:, 1])
        assert got["num_distinct"].tolist() == [r["num_distinct"] for r in g["reads"]]
        assert got["total_kmers"].tolist() == [r["total_kmers"] for r in g["reads"]]
    ho = got = st.classify_batch(bases, offsets, thresholds=[0.0])
    off = ho["hit_offsets"].astype(np.int64)
    for i, ((title, _), want) in enumerate(zip(reads, g["reads"])):
        h = ho["hits"][off[i]:off[i + 1]]
        hits = [(int(t), int(c)) for t, c in zip(h["taxon"], h["count"])]
        assert [list(x) for x in hits] == want["hits"]
        if hits:  # a read without spans produces no output row at all (SURVEY 3.3 i)
            line = orc.output_line(bool(ho["classified"][0][i]), title, int(ho["taxon"][0][i]), hits, g["k"])
            assert line == want["line"]
    # paired
    m1 = [seqs[i] for i in range(0, 300, 2)]
    m2 = [seqs[i + 1] for i in range(0, 300, 2)]
    o1 = np.zeros(len(m1) + 1, np.uint64); np.cumsum([len(s) for s in m1], out=o1[1:])
    o2 = np.zeros(len(m2) + 1, np.uint64); np.cumsum([len(s) for s in m2], out=o2[1:])
    pg = st.classify_batch(np.concatenate(m1), o1, np.concatenate(m2), o2, thresholds=[0.0])
    off = pg["hit_offsets"].astype(np.int64)
    for j, want in enumerate(g["pairs"]):
        h = pg["hits"][off[j]:off[j + 1]]
        hits = [(int(t), int(c)) for t, c in zip(h["taxon"], h["count"])]
        line = orc.output_line(bool(pg["classified"][0][j]), want["title"], int(pg["taxon"][0][j]), hits, g["k"])
        assert line == want["line"]

"""Regrouping by title (Classifier.scala:92,136) and the paired reader's join (InputReader.scala:104-119): the restatement in
hostmodel.py and the oracle's Classifier.classify on merged hit lists, on CPU.  The engine's side of it is tested through the
CLI in test_host_classify_gpu.py.  The reference holds no fixture for repeated titles: parity unpinned."""
import json
import os

import numpy as np

import hostmodel
from oracle import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_merge_is_a_stable_sort_by_ordinal():
    a = [(5, 3), (0, 2), (5, 1)]
    b = [(7, 4), (-2, -34), (7, 2), (9, 9)]
    (title, hits, distinct), = hostmodel.merge_by_title([("x", a, [1, 0, 1]), ("x", b, [1, 0, 0, 1])])
    assert title == "x"
    assert hits == [(5, 3), (7, 4), (0, 2), (-2, -34), (5, 1), (7, 2), (9, 9)]
    assert distinct == [1, 1, 0, 0, 1, 0, 1]
    # fragments without spans take no part; order of first appearance
    rows = hostmodel.merge_by_title([("b", [], []), ("a", a, [1, 1, 1]), ("b", b, [0] * 4), ("c", [], [])])
    assert [r[0] for r in rows] == ["a", "b"] and rows[1][1] == b


def test_paired_join_multiplies_repeated_headers():
    r1 = [("x/1", "A"), ("y/1", "C"), ("x/1", "G"), ("z/1", "T")]
    r2 = [("y/2", "c"), ("x/2", "a"), ("x/2", "g"), ("w/2", "t")]
    assert hostmodel.paired_join(r1, r2) == [("x", "A", "a"), ("x", "A", "g"), ("y", "C", "c"), ("x", "G", "a"), ("x", "G", "g")]


def test_length_string_is_taken_from_the_merged_counts():
    """TaxonCounts.lengthString (TaxonCounts.scala:114-121) works on the fromHits-merged arrays: two fragments' borders have
    the same ordinal only if their first mates have equally many spans -- then they are adjacent and collapse."""
    k = 35
    one = [(5, 3), (-2, -34), (7, 4)]
    assert oracle.length_string(one, k) == "37|38"
    adjacent = [(5, 3), (5, 2), (-2, -34), (-2, -34), (7, 4), (7, 1)]
    assert oracle.length_string(adjacent, k) == "39|39"            # the collapsed border entry is skipped as a whole
    apart = [(5, 3), (-2, -34), (7, 4), (-2, -34), (8, 1)]
    assert oracle.length_string(apart, k) == "37|5"                # the second border's count is part of drop(border + 1).sum
    assert oracle.pairs_in_order_string(adjacent) == "5:5 |:| 7:5"


def test_classify_hits_equals_classify_read_on_a_single_fragment():
    g = json.load(open(os.path.join(GOLD, "golden_classify.json")))
    lib = np.load(os.path.join(GOLD, "library.npz"))
    p = oracle.params(k=g["k"], m=g["m"], spaces=g["spaces"])
    oix = oracle.Index(1, lib["keys"], lib["taxa"])
    reads = [line.rstrip("\n").split("\t") for line in open(os.path.join(GOLD, "reads.tsv"))][:60]
    for thr in (0.0, 0.5):
        for (t1, s1), (t2, s2) in zip(reads[::2], reads[1::2]):
            res, hits = oracle.classify_read(p, oix, lib["parents"], s1, s2, confidence=thr)
            distinct = [x["distinct"] for x in oracle.spans(p, s1, s2)]
            again = oracle.classify_hits(lib["parents"], hits, distinct, 2, thr)
            assert again == res
    # merging two fragments: counts add up, distinct hits add up
    (_, s1), (_, s2) = reads[0], reads[1]
    r1, h1 = oracle.classify_read(p, oix, lib["parents"], s1)
    r2, h2 = oracle.classify_read(p, oix, lib["parents"], s2)
    d1 = [x["distinct"] for x in oracle.spans(p, s1)]
    d2 = [x["distinct"] for x in oracle.spans(p, s2)]
    (_, hits, distinct), = hostmodel.merge_by_title([("t", h1, d1), ("t", h2, d2)])
    m = oracle.classify_hits(lib["parents"], hits, distinct)
    assert m["total_kmers"] == r1["total_kmers"] + r2["total_kmers"]
    assert m["num_distinct"] == r1["num_distinct"] + r2["num_distinct"]
    assert m["num_hits"] == len(h1) + len(h2)


# ---- the host's detection of the titles it must regroup (`slacken-amd repeated`, no GPU): every title whose fragments the
# reference would merge -- or whose join has more products than the streaming walk makes -- must be found
def _repeated(tmp_path, recs1, recs2=None):
    import subprocess
    from test_host_cli import CLI
    f1 = tmp_path / "a_1.fq"
    with open(f1, "w") as f:
        for h, q in recs1:
            f.write(f"@{h}\n{q}\n+\n{'I' * len(q)}\n")
    args = [str(f1)]
    if recs2 is not None:
        f2 = tmp_path / "a_2.fq"
        with open(f2, "w") as f:
            for h, q in recs2:
                f.write(f"@{h}\n{q}\n+\n{'I' * len(q)}\n")
        args = ["-p", str(f1), str(f2)]
    r = subprocess.run([CLI, "repeated", *args], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r.stdout.split()


def test_host_finds_the_titles_to_regroup(tmp_path):
    import random
    rnd = random.Random(4)
    seq = lambda: "".join(rnd.choice("ACGT") for _ in range(60))
    # unpaired: titles that occur twice or three times, among many that occur once
    recs = [(f"r{i}", seq()) for i in range(3000)] + [("r7", seq()), ("r2999", seq()), ("r7", seq())]
    rnd.shuffle(recs)
    assert _repeated(tmp_path, recs) == sorted(["r7", "r2999"])
    assert _repeated(tmp_path, [(f"u{i}", seq()) for i in range(500)]) == []
    # paired: every shape of a repeated header, with the two files in the same order and in different orders
    for trial in range(6):
        n = 400
        r1 = [(f"p{i}", seq()) for i in range(n)]
        r2 = [(f"p{i}", seq()) for i in range(n)]
        r1.insert(120, ("p5", seq()))                       # 2 x 1: file 1 repeats a header
        r2.insert(300, ("p9", seq()))                       # 1 x 2: file 2 repeats a header
        r1.insert(10, ("p200", seq())); r2.insert(350, ("p200", seq()))   # 2 x 2
        r1.append(("only1", seq())); r2.append(("only2", seq()))          # no partner: no fragment, nothing to regroup
        r1.append(("dup_no_mate", seq())); r1.append(("dup_no_mate", seq()))   # repeated, but no product at all
        if trial % 3 == 1:
            rnd.shuffle(r2)
        elif trial % 3 == 2:
            rnd.shuffle(r1)
        joined = hostmodel.paired_join([(h + "/1", q) for h, q in r1], [(h + "/2", q) for h, q in r2])
        count = {}
        for t, _, _ in joined:
            count[t] = count.get(t, 0) + 1
        want = sorted(t for t, c in count.items() if c >= 2)
        assert want == ["p200", "p5", "p9"]
        got = _repeated(tmp_path, [(h + "/1", q) for h, q in r1], [(h + "/2", q) for h, q in r2])
        assert set(want) <= set(got) <= set(want) | {"dup_no_mate"}, (trial, got)


def test_the_merged_list_helper_of_the_gpu_tests_is_from_hits():
    """tests/test_gpu_parity.py: merged_lists (numpy) is what the GPU tests hold the device-merged hit lists against; here it is held
    against TaxonCounts.fromHits (TaxonCounts.scala:31-48) written out plainly: adjacent hits of one taxon become one entry with
    the sum of their counts -- per fragment, borders (-2) and ambiguous spans (-1) like any other taxon."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gpu_parity_helpers", os.path.join(os.path.dirname(__file__), "test_gpu_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    dt = np.dtype([("taxon", np.int32), ("count", np.int32)])
    rng = np.random.default_rng(31)
    for _ in range(50):
        lists = []
        for _r in range(int(rng.integers(1, 40))):
            n = int(rng.integers(0, 12)) if rng.random() < 0.8 else 0
            lists.append([(int(rng.choice([-2, -1, 0, 5, 5, 5, 9, 9, 12])), int(rng.integers(1, 30))) for _ in range(n)])
        offs = np.cumsum([0] + [len(h) for h in lists]).astype(np.uint64)
        flat = np.array([h for hs in lists for h in hs], dt) if offs[-1] else np.zeros(0, dt)
        m_off, m_hits = mod.merged_lists(offs, flat)
        want = []
        for hs in lists:
            out = []
            for t, c in hs:
                if out and out[-1][0] == t:
                    out[-1] = (t, out[-1][1] + c)
                else:
                    out.append((t, c))
            want.append(out)
        assert m_off.tolist() == np.cumsum([0] + [len(h) for h in want]).tolist()
        assert [(int(t), int(c)) for t, c in zip(m_hits["taxon"], m_hits["count"])] == [h for hs in want for h in hs]

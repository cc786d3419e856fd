"""The reference's property specs for the scan, restated with hypothesis against the C oracle
(T/kmers/minimizer/MinSplitterProps.scala, T/slacken/SupermersProps.scala, T/kmers/util/NTBitArrayProps.scala)."""
import random

from hypothesis import given, settings, strategies as st

import pymodel

TOGGLE = 0xe37e28c4271b5a2d
SET = settings(max_examples=150, deadline=None)


@st.composite
def m_k(draw, max_k=91):
    # TestGenerators.mAndKPairs: k odd in [1, 91], m in [1, k]
    k = draw(st.integers(1, max_k).filter(lambda x: x % 2 == 1))
    m = draw(st.integers(1, k))
    return m, k


@st.composite
def priorities(draw, m, canonical_only=False):
    # TestGenerators.minimizerPriorities: RandomXOR(m, DEFAULT_TOGGLE_MASK, canonical = true) (+ the MinTable
    # lexicographic ordering for m <= 10, here the identity ordering), with SpacedSeed s in [0, m/2]
    s = draw(st.integers(0, m // 2))
    if m <= 10 and not canonical_only and draw(st.booleans()):
        return dict(m=m, spaces=s, xor_mask=0, canonical=False)
    return dict(m=m, spaces=s, xor_mask=TOGGLE, canonical=True)


def dna(min_len, max_len=200, alphabet="ACTG"):
    return st.text(alphabet=alphabet, min_size=min_len, max_size=max(min_len, max_len))


@st.composite
def case_mkx(draw, canonical_only=False, alphabet="ACTG"):
    m, k = draw(m_k())
    pri = draw(priorities(m, canonical_only))
    x = draw(dna(k, 200, alphabet))
    return k, pri, x


@SET
@given(case_mkx())
def test_splitting_preserves_all_data(orc, c):  # MinSplitterProps.scala:29-45
    k, pri, x = c
    sm = orc.split_encode(orc.params(k=k, **pri), x)
    parts = [x[s:s + l] for _, s, l in sm]
    assert parts[0] + "".join(p[k - 1:] for p in parts[1:]) == x


@SET
@given(case_mkx())
def test_adjacent_minimizers_differ(orc, c):  # :47-61
    k, pri, x = c
    sm = orc.split_encode(orc.params(k=k, **pri), x)
    for a, b in zip(sm, sm[1:]):
        assert a[0] != b[0]


@SET
@given(case_mkx())
def test_minimizers_are_minimal_mmers(orc, c):  # :63-79
    k, pri, x = c
    p = orc.params(k=k, **pri)
    for key, s, l in orc.split_encode(p, x):
        region = x[s:s + l]
        ks = [orc.priority(p, orc.encode(region[i:i + p.m])) for i in range(len(region) - p.m + 1)]
        assert list(key) == min(ks)


@SET
@given(m_k().flatmap(lambda mk: st.tuples(st.just(mk), priorities(mk[0]), dna(0, max(0, mk[1] - 1)))))
def test_too_short_has_no_minimizers(orc, c):  # :81-89
    (m, k), pri, x = c
    x = x[:k - 1]
    assert orc.split_encode(orc.params(k=k, **pri), x) == []


@SET
@given(st.integers(1, 63).flatmap(lambda m: st.tuples(priorities(m, True), dna(m, m))))
def test_canonical_priority_equal_for_rc(orc, c):  # :91-99
    pri, x = c
    p = orc.params(k=pri["m"], **pri)
    e = orc.encode(x)
    assert orc.priority(p, e) == orc.priority(p, orc.reverse_complement(e, p.m))


def revcomp(s):
    return s.upper().translate(str.maketrans("ACGTU", "TGCAA"))[::-1]


@SET
@given(case_mkx(canonical_only=True))
def test_supermers_invariant_under_rc(orc, c):  # :101-114
    k, pri, x = c
    p = orc.params(k=k, **pri)
    fwd = [key for key, _, _ in orc.split_encode(p, x)]
    rev = [key for key, _, _ in orc.split_encode(p, revcomp(x))]
    assert fwd == rev[::-1]


@SET
@given(case_mkx(alphabet="ACTGUactgu"))
def test_literal_window_equals_min_rle_model(orc, c):
    # SURVEY 3.2 claim: the deque/tie-break machinery is unobservable; super-mers == RLE of window minima by value.
    k, pri, x = c
    p = orc.params(k=k, **pri)
    got = orc.split_encode(p, x)
    want = [(pymodel.left_align(v, p.m), s, l) for v, s, l in pymodel.supermers(x, k, **pri)]
    assert got == want


@SET
@given(st.integers(1, 128).flatmap(lambda n: dna(n, n)))
def test_bitarray_props(orc, x):  # NTBitArrayProps.scala:94-128, BitRepresentationProps.scala:42-72
    n = len(x)
    e = orc.encode(x)
    assert orc.decode(e, n) == x
    r = orc.reverse_complement(e, n)
    assert orc.decode(r, n) == revcomp(x)
    assert orc.reverse_complement(r, n) == e
    can = orc.canonical(e, n)
    assert can == min(e, r) and not (can > e)


@SET
@given(st.integers(1, 60), st.text(alphabet="ACTGNactgn\n", min_size=0, max_size=300))
def test_split_by_ambiguity(orc, k, x):  # SupermersProps.scala:40-57 + Supermers.scala:150-189
    segs = orc.split_by_ambiguity(x, k)
    assert "".join(x[s:s + l] for s, l, _ in segs) == x
    for (s, l, f), nxt in zip(segs, segs[1:] + [None]):
        run = x[s:s + l]
        clean = all(ch in "ACTGactg\n\r" for ch in run)
        nvalid = sum(ch in "ACTGactg" for ch in run)
        assert f == (orc.SEQUENCE_FLAG if clean and nvalid >= k else orc.AMBIGUOUS_FLAG)
        if nxt is not None:  # maximal alternating runs
            assert clean != all(ch in "ACTGactg\n\r" for ch in x[nxt[0]:nxt[0] + nxt[1]])


def model_spans(orc, p, seq1, seq2=None):
    """Supermers.splitFragment + spans from SURVEY 3.3, on top of pymodel.supermers (whitespace-free input)."""
    out, state = [], dict(first=True, last=None)

    def emit(key, size, flag):
        seqlike = flag == orc.SEQUENCE_FLAG
        d = seqlike and (state["first"] or key != state["last"])
        if seqlike:
            state["last"] = key
        state["first"] = False
        out.append(dict(key=key if seqlike else (0,) * p.W, kmers=size - (p.k - 1), flag=flag, ordinal=len(out),
                        distinct=d))

    def one(seq):
        i = 0
        while i < len(seq):
            j = i
            ok = seq[i] in "ACGTUacgtu"
            while j < len(seq) and (seq[j] in "ACGTUacgtu") == ok:
                j += 1
            if j - i >= p.k:
                if ok:
                    for v, s, l in pymodel.supermers(seq[i:j], p.k, p.m, p.spaces, p.xor_mask, bool(p.canonical)):
                        emit(pymodel.left_align(v, p.m), l, orc.SEQUENCE_FLAG)
                else:
                    emit(None, j - i, orc.AMBIGUOUS_FLAG)
            i = j

    one(seq1)
    if seq2 is not None:
        emit(None, 0, orc.MATE_PAIR_BORDER_FLAG)
        one(seq2)
    return out


@SET
@given(st.sampled_from([(35, 31, 7), (31, 31, 0), (15, 7, 3), (21, 12, 0), (45, 33, 5)]),
       st.text(alphabet="ACGT" * 12 + "Nacgtu", min_size=0, max_size=400),
       st.one_of(st.none(), st.text(alphabet="ACGT" * 12 + "Nn", min_size=0, max_size=200)))
def test_spans_match_model(orc, kms, s1, s2):
    k, m, s = kms
    p = orc.params(k=k, m=m, spaces=s)
    assert orc.spans(p, s1, s2) == model_spans(orc, p, s1, s2)


def test_spans_quirks(orc):
    # SURVEY 3.3 (i)-(iii)
    p = orc.params()
    rnd = random.Random(7)
    r = "".join(rnd.choice("ACGT") for _ in range(150))
    assert orc.spans(p, r[:34]) == []                      # shorter than k: no rows at all
    assert orc.spans(p, r[:20] + "N" + r[20:40]) == []     # only short runs: vanishes
    sp = orc.spans(p, r[:60] + "N" * 40 + r[60:])          # ambiguous run >= k: ONE span, kmers = L-(k-1)
    amb = [x for x in sp if x["flag"] == orc.AMBIGUOUS_FLAG]
    assert len(amb) == 1 and amb[0]["kmers"] == 40 - 34 and not amb[0]["distinct"]
    assert [x["ordinal"] for x in sp] == list(range(len(sp)))
    sp = orc.spans(p, r[:60] + "N" + r[61:])               # single N: dropped, no span for it
    assert all(x["flag"] == orc.SEQUENCE_FLAG for x in sp)
    assert sum(x["kmers"] for x in sp) == (60 - 34) + (89 - 34)
    pair = orc.spans(p, r, r)                              # mate border: kmers = -(k-1), never distinct
    b = [x for x in pair if x["flag"] == orc.MATE_PAIR_BORDER_FLAG]
    assert len(b) == 1 and b[0]["kmers"] == -34 and not b[0]["distinct"]
    n1 = b[0]["ordinal"]
    assert pair[n1 + 1]["distinct"] == (pair[n1 + 1]["key"] != pair[n1 - 1]["key"])  # lastMinimizer survives border
    assert orc.spans(p, "", "") == [dict(key=(0,), kmers=-34, flag=3, ordinal=0, distinct=False)]


@settings(max_examples=60, deadline=None)
@given(st.integers(97, 128).flatmap(lambda m: st.tuples(st.just(m), st.integers(m, m + 31), priorities(m, True), dna(m + 31, 400))))
def test_four_word_minimizers_match_the_python_model(orc, c):
    # m in 97..128 (four id columns; the reference's ClassifierTest draws m up to 128): the C restatement's multi-word
    # arithmetic against the pure-Python-integer model
    m, k, pri, x = c
    p = orc.params(k=k, **pri)
    got = orc.split_encode(p, x)
    want = [(pymodel.left_align(v, p.m), s, l) for v, s, l in pymodel.supermers(x, k, **pri)]
    assert got == want


@SET
@given(st.integers(1, 40).flatmap(lambda m: st.tuples(st.just(m), st.text(alphabet="ACTGactg", min_size=m, max_size=200))))
def test_scanner_finds_all_mmers(orc, c):  # ShiftScannerProps.scala:28-58 "Find all m-mers", :60-68 "Encoding of NT sequence"
    """With a priority that is the identity on the encoded m-mer (no toggle mask, no spaces, forward orientation only -- the reference
    test uses a table that permits every m-mer) the scanner's match at every position from m - 1 on IS the m-mer that ends there,
    upper case; the first m - 1 positions carry no match; and with the canonical orientation it is the smaller of the m-mer and its
    reverse complement (the scan of the reverse strand that the reference test makes separately)."""
    m, x = c
    p = orc.params(k=m, m=m, spaces=0, xor_mask=0, canonical=False)
    got = orc.all_matches(p, x)
    assert len(got) == len(x) and all(not v for _, v in got[:m - 1]) and all(v for _, v in got[m - 1:])
    up = x.upper()
    assert [orc.decode(list(k), m) for k, _ in got[m - 1:]] == [up[i:i + m] for i in range(len(x) - m + 1)]
    pc = orc.params(k=m, m=m, spaces=0, xor_mask=0, canonical=True)
    can = orc.all_matches(pc, x)
    assert [orc.decode(list(k), m) for k, _ in can[m - 1:]] == \
        [min(up[i:i + m], revcomp(up[i:i + m]), key=lambda s: orc.encode(s)) for i in range(len(x) - m + 1)]
    assert orc.all_matches(p, x[:m // 2] + "#" + x[m // 2:]) is None            # (InvalidNucleotideException, ShiftScanner.scala:101)

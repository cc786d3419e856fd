"""Long unpaired fragments on the hot path run with one lane per SEGMENT of the fragment (fused.hip: segment_kernel): what a
lane cannot see -- whether its first super-mer equals the last one before it, whether a super-mer or an ambiguous span was cut
by a segment border -- is settled at the end.  Reads built to hit those borders, against the oracle: taxon, classified,
distinct hit groups, k-mer total and the number of spans -- and the un-merged hit lists in ordinal order, which the segment kernel
can put together from its lanes' stretches once the borders are settled (a span cut by a border is ONE hit with the k-mers of both
parts; SLK_SEG_HITS=1: by default the hit lists of long reads stay with the wave kernel, the faster route for them)."""
import os

import numpy as np
import pytest

import synth
import taxgen

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def segments_from_1001_bases(monkeypatch):
    # the engine's default hands the fragments that are long for their batch to the segment kernel (those of 1001..4999 to the lane kernel's
    # long variant, those in between to the wave kernel); here everything over 1000 goes to the segment kernel, which makes the segments short (64 windows) and the
    # borders many
    monkeypatch.setenv("SLK_SEG_MIN_LEN", "1001")
    monkeypatch.setenv("SLK_LANE_LONG_MAX", "0")
    # hit lists too come from the segment kernel here (the engine's default keeps them on the wave kernel, which measured faster)
    monkeypatch.setenv("SLK_SEG_HITS", "1")


@pytest.fixture(scope="module")
def world(orc):
    import slacken_amd
    p = orc.params()
    rng = np.random.default_rng(77)
    parents = taxgen.taxonomy(8 * 32, rng)
    lib = synth.Library(orc, p, parents, n_genomes=12, genome_len=30000, pad_records=20000)
    # low-complexity sequence is in the library too, so that long super-mers with equal keys carry real taxa
    units = [synth.random_dna(int(rng.integers(1, 7)), rng) for _ in range(24)] + [np.frombuffer(b"AC", np.uint8), np.frombuffer(b"ACGGT", np.uint8)]
    keys, taxa = [lib.keys], [lib.taxa]
    for u in units:
        kk = np.setdiff1d(np.unique(orc.minimizer_keys(p, np.tile(u, 40).tobytes())), np.concatenate(keys))
        keys.append(kk)
        taxa.append(np.full(len(kk), int(rng.choice(lib.genome_taxa)), np.int32))
    keys, taxa = np.concatenate(keys), np.concatenate(taxa).astype(np.int32)
    lib.units = units
    ix = slacken_amd.Index(expected_records=len(keys), max_taxon=len(parents) - 1)
    ix.append(keys, taxa)
    ix.set_taxonomy(parents)
    ix.finalize()
    return dict(p=p, lib=lib, parents=parents, st=ix.stream(), oix=orc.Index(1, keys, taxa), ix=ix)


def long_read(lib, rng, length):
    """pieces of several genomes (several taxa per read), substitutions, single Ns, N runs shorter and longer than k,
    repeats whose windows share one minimizer for hundreds of bases"""
    parts, have = [], 0
    while have < length:
        kind = rng.random()
        if kind < 0.70:
            g = lib.genomes[rng.integers(0, len(lib.genomes))]
            L = int(rng.integers(100, 4000))
            a = int(rng.integers(0, len(g) - L))
            piece = g[a:a + L].copy()
            if rng.random() < 0.5:
                piece = synth.revcomp(piece).copy()
            subs = rng.random(L) < 0.01
            piece[subs] = synth.random_dna(int(subs.sum()), rng)
        elif kind < 0.80:
            piece = np.full(int(rng.choice([1, 2, 5, 30, 34, 35, 36, 40, 64, 70, 100, 200])), ord("N"), np.uint8)
        elif kind < 0.90:
            unit = lib.units[rng.integers(0, len(lib.units))]
            piece = np.tile(unit, int(rng.integers(20, 400)))
        else:
            piece = synth.random_dna(int(rng.integers(10, 500)), rng)
        parts.append(piece)
        have += len(piece)
    return np.concatenate(parts)[:length].copy()


def check(orc, world, reads, thresholds=(0.0, 0.1, 0.5), min_hit_groups=2):
    bases, offsets = synth.pack(reads)
    got = world["st"].classify_batch(bases, offsets, thresholds=thresholds, min_hit_groups=min_hit_groups, with_hits=False,
                                     with_num_hits=True)
    want = orc.classify_batch(world["p"], world["oix"], world["parents"], bases, offsets, None, None,
                              min_hit_groups=min_hit_groups, thresholds=thresholds)
    for key in ("total_kmers", "num_hits", "num_distinct", "taxon", "classified"):
        bad = np.nonzero(np.atleast_2d(got[key] != want[key]).any(axis=0))[0]
        assert bad.size == 0, (key, bad[:5].tolist(), [len(reads[i]) for i in bad[:5]],
                               np.atleast_2d(got[key])[:, bad[:5]].tolist(), np.atleast_2d(want[key])[:, bad[:5]].tolist())
    # the same batch with every handed-on fragment on the wave kernel, and with the engine's default routes (1001..4999 bases: the
    # lane kernel's long variant, the long ones of the batch: the segment kernel, the rest and map overflows: the wave kernel)
    # -- the latter with a first pass of the lane kernel and with the routing kernel in its place (the engine's choice follows the
    # batch's mean fragment length), where the fragments of up to 1000 bases are a fifth class of the long variant
    saved = {v: os.environ.get(v) for v in ("SLK_SEG_MIN_LEN", "SLK_LANE_LONG_MAX", "SLK_ROUTE_FIRST")}
    try:
        for name, env in (("wave kernel", dict(SLK_SEG_MIN_LEN="0", SLK_LANE_LONG_MAX="0")), ("default routes", {}),
                          ("default routes behind a first pass", dict(SLK_ROUTE_FIRST="0")),
                          ("default routes behind the routing kernel", dict(SLK_ROUTE_FIRST="1"))):
            for v in saved:
                os.environ.pop(v, None)
            os.environ.update(env)
            other = world["st"].classify_batch(bases, offsets, thresholds=thresholds, min_hit_groups=min_hit_groups, with_hits=False,
                                               with_num_hits=True)
            for key in ("total_kmers", "num_hits", "num_distinct", "taxon", "classified"):
                assert np.array_equal(other[key], got[key]), name + " vs segment kernel: " + key
    finally:
        for v, x in saved.items():
            os.environ.pop(v, None)
            if x is not None:
                os.environ[v] = x
    return got


def check_hits(orc, world, reads, every=1):
    """hit lists (TaxonHit per span, ordinal order; TaxonCounts.scala:94-121 is formatted from them) of a batch on the segment
    route against the oracle's, and against the wave kernel's for the same batch"""
    bases, offsets = synth.pack(reads)
    got = world["st"].classify_batch(bases, offsets, thresholds=(0.0, 0.2), with_hits=True)
    want = orc.classify_batch(world["p"], world["oix"], world["parents"], bases, offsets, None, None, min_hit_groups=2, thresholds=(0.0, 0.2))
    for key in ("total_kmers", "num_hits", "num_distinct", "taxon", "classified"):
        assert np.array_equal(got[key], want[key]), key
    ho = got["hit_offsets"].astype(np.int64)
    for i in range(0, len(reads), every):
        _, hits = orc.classify_read(world["p"], world["oix"], world["parents"], reads[i].tobytes(), None, 2, 0.0)
        g = got["hits"][ho[i]:ho[i + 1]]
        assert [(int(t), int(c)) for t, c in zip(g["taxon"], g["count"])] == hits, (i, len(reads[i]))
    saved = {v: os.environ.get(v) for v in ("SLK_SEG_MIN_LEN", "SLK_LANE_LONG_MAX", "SLK_ROUTE_FIRST")}
    try:
        # the wave kernel alone; the default routes behind a first pass and behind the routing kernel (long variant with hit lists)
        for name, env in (("wave kernel", dict(SLK_SEG_MIN_LEN="0", SLK_LANE_LONG_MAX="0")),
                          ("first pass", dict(SLK_ROUTE_FIRST="0")), ("routing kernel", dict(SLK_ROUTE_FIRST="1"))):
            for v in saved:
                os.environ.pop(v, None)
            os.environ.update(env)
            other = world["st"].classify_batch(bases, offsets, thresholds=(0.0, 0.2), with_hits=True)
            assert np.array_equal(other["hit_offsets"], got["hit_offsets"]) and np.array_equal(other["hits"], got["hits"]), name
            for key in ("total_kmers", "num_hits", "num_distinct", "taxon", "classified"):
                assert np.array_equal(other[key], got[key]), name + ": " + key
    finally:
        for v, x in saved.items():
            os.environ.pop(v, None)
            if x is not None:
                os.environ[v] = x


@pytest.mark.parametrize("seed", range(int(os.environ.get("SLK_SEG_SEEDS", 4))))   # (SLK_SEG_SEEDS: soak runs)
def test_hit_lists_from_the_segment_kernel(orc, world, seed):
    rng = np.random.default_rng(900 + seed)
    reads = [long_read(world["lib"], rng, int(rng.integers(1001, 4200))) for _ in range(60)]
    reads += [long_read(world["lib"], rng, int(rng.integers(4200, 30000))) for _ in range(15)]
    reads += synth.make_reads(world["lib"], 100, rng, vary_length=True)
    order = rng.permutation(len(reads))
    check_hits(orc, world, [reads[i] for i in order])


@pytest.mark.parametrize("seed", range(int(os.environ.get("SLK_SEG_SEEDS", 4))))   # (SLK_SEG_SEEDS: soak runs)
def test_long_reads_with_cut_spans(orc, world, seed):
    rng = np.random.default_rng(500 + seed)
    # up to 4200 bases: segments of 64 windows, a border every 64 bases; longer reads: longer segments
    reads = [long_read(world["lib"], rng, int(rng.integers(1001, 4200))) for _ in range(120)]
    reads += [long_read(world["lib"], rng, int(rng.integers(4200, 30000))) for _ in range(40)]
    reads += synth.make_reads(world["lib"], 300, rng, vary_length=True)      # short ones in the same batch (lane kernel)
    order = rng.permutation(len(reads))
    got = check(orc, world, [reads[i] for i in order], min_hit_groups=int(rng.integers(1, 4)))
    assert got["classified"][0].mean() > 0.3


def test_segment_borders_one_by_one(orc, world):
    """One read, then the same read with a run of Ns (shorter than, equal to, longer than k) or a repeat slid base by base
    across a segment border."""
    rng = np.random.default_rng(9)
    g = world["lib"].genomes[0]
    base = g[:3300].copy()
    reads = [base]
    for ins_len in (1, 34, 35, 36, 70):
        for at in range(1000, 1000 + 70, 3):
            r = base.copy()
            r[at:at + ins_len] = ord("N")
            reads.append(r)
    unit = np.frombuffer(b"AC", np.uint8)
    for at in range(1900, 1900 + 70, 5):
        r = base.copy()
        r[at:at + 300] = np.tile(unit, 150)
        reads.append(r)
    reads.append(np.full(5000, ord("N"), np.uint8))                   # nothing but one ambiguous span, cut 63 times
    reads.append(np.tile(np.frombuffer(b"ACGGT", np.uint8), 1000))    # one minimizer value throughout
    reads.append(np.full(3000, ord("A"), np.uint8))
    check(orc, world, reads, thresholds=(0.0, 0.3))
    check_hits(orc, world, reads)       # spans cut by one border, by sixty-three, N runs slid across a border: one hit each


def test_hit_lists_when_few_lanes_hold_all_the_sequence(orc, world):
    """The hit-list variant queues a lane's entries in LDS, eight at a time; a chunk of 64 spans that comes from two or three lanes
    -- islands of sequence in a sea of Ns, or one lane's stretch of a long fragment with a new minimizer in every window while its
    neighbours sit in a repeat -- overruns the queues, and the surplus goes straight to its place."""
    rng = np.random.default_rng(41)
    lib = world["lib"]
    reads = []
    for n_islands, island, sea in ((1, 120, 3000), (2, 90, 2500), (3, 200, 4000), (1, 64 + 34, 6000), (5, 70, 900)):
        parts = [np.full(sea, ord("N"), np.uint8)]
        for _ in range(n_islands):
            g = lib.genomes[rng.integers(0, len(lib.genomes))]
            a = int(rng.integers(0, len(g) - island))
            parts += [g[a:a + island].copy(), np.full(int(rng.integers(sea // 2, sea)), ord("N"), np.uint8)]
        reads.append(np.concatenate(parts))
    # repeats (one span per hundreds of bases) around a stretch of genome (a span every few bases)
    for unit_len, mid in ((2, 300), (5, 150), (3, 1000)):
        unit = synth.random_dna(unit_len, rng)
        g = lib.genomes[rng.integers(0, len(lib.genomes))]
        a = int(rng.integers(0, len(g) - mid))
        reads.append(np.concatenate([np.tile(unit, 4000 // unit_len), g[a:a + mid], np.tile(unit, 5000 // unit_len)]))
    check(orc, world, reads)
    check_hits(orc, world, reads)


def test_very_long_read(orc, world):
    rng = np.random.default_rng(3)
    reads = [long_read(world["lib"], rng, 400_000), long_read(world["lib"], rng, 70_000)]
    check(orc, world, reads)
    check_hits(orc, world, reads)


def test_default_threshold_splits_the_work_between_the_two_kernels(orc, world, monkeypatch):
    """Default routes: the segment kernel's threshold follows the batch (1/16384 of its bases, at least 16 000 -- which is what a batch
    of this size gets --, at most 250 000); shorter long fragments, clean or with characters outside ACGTU (which the wave kernel
    takes run by run), stay on the wave kernel (three lists by length, longest first) or the lane kernel's long variant, short ones
    on the lane kernel -- one batch."""
    monkeypatch.delenv("SLK_SEG_MIN_LEN")
    monkeypatch.delenv("SLK_LANE_LONG_MAX")
    rng = np.random.default_rng(12)
    reads = []
    for _ in range(150):
        r = long_read(world["lib"], rng, int(rng.integers(1001, 5000)))
        if rng.random() < 0.5:
            r = np.where(r == ord("N"), ord("A"), r).astype(np.uint8)    # a clean one
        reads.append(r)
    reads += synth.make_reads(world["lib"], 200, rng, vary_length=True)
    reads += [long_read(world["lib"], rng, int(rng.integers(5000, 16000))) for _ in range(12)]
    reads += [long_read(world["lib"], rng, int(rng.integers(16000, 40000))) for _ in range(12)]
    reads += [long_read(world["lib"], rng, n) for n in (9999, 10000, 10001, 15999, 16000, 16001, 250001)]
    bases, offsets = synth.pack(reads)
    got = world["st"].classify_batch(bases, offsets, thresholds=(0.0, 0.2), with_hits=False, with_num_hits=True)
    want = orc.classify_batch(world["p"], world["oix"], world["parents"], bases, offsets, None, None, min_hit_groups=2,
                              thresholds=(0.0, 0.2))
    for key in ("total_kmers", "num_hits", "num_distinct", "taxon", "classified"):
        assert np.array_equal(got[key], want[key]), key


def many_slices(lib, rng, n, lo, hi):
    """n fragments of lo..hi bases cut from the library's genomes (vectorised: these batches hold 10^4 .. 10^5.5 fragments)"""
    cat = np.concatenate(lib.genomes)
    glen = len(lib.genomes[0])
    lens = rng.integers(lo, hi + 1, n)
    offsets = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=offsets[1:])
    starts = rng.integers(0, len(lib.genomes), n) * glen + rng.integers(0, glen - hi, n)
    rid = np.repeat(np.arange(n), lens)
    bases = cat[starts[rid] + (np.arange(int(offsets[-1])) - offsets[:-1].astype(np.int64)[rid])].copy()
    subs = rng.random(len(bases)) < 0.01
    bases[subs] = synth.random_dna(int(subs.sum()), rng)
    bases[rng.random(len(bases)) < 0.0005] = ord("N")
    return bases, offsets


KEYS = ("total_kmers", "num_hits", "num_distinct", "taxon", "classified")


@pytest.mark.parametrize("route", ["wave", "segment", "long"])
def test_hand_on_lists_longer_than_the_grids(orc, world, monkeypatch, route):
    """The kernels behind the first pass have fixed grids (8 192 waves; the long lane variant at most 5 120) and the number of
    hand-ons is only known on the device: a wave takes the unit of its own number first and draws further ones from a counter.
    More hand-ons than waves on each route, against the oracle."""
    monkeypatch.setenv("SLK_SEG_MIN_LEN", {"wave": "0", "segment": "1001", "long": "5000"}[route])
    monkeypatch.setenv("SLK_LANE_LONG_MAX", "4999" if route == "long" else "0")
    rng = np.random.default_rng(77)
    # the long variant's grid is one wave per 64 fragments of the BATCH, capped at 5 120: its draw needs over 327 680 hand-ons
    n = 340_000 if route == "long" else 20_000
    bases, offsets = many_slices(world["lib"], rng, n, 1001, 1100 if route == "long" else 1300)
    import torch
    st = world["st"]
    d_b, d_o = torch.from_numpy(bases).cuda(), torch.from_numpy(offsets.astype(np.int64)).cuda()   # one undivided batch: the device entry
    outs = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(3)]
    d_t, d_c = torch.zeros(2 * n, dtype=torch.int32, device="cuda"), torch.zeros(2 * n, dtype=torch.uint8, device="cuda")
    st.classify_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, int(offsets[-1]), d_t.data_ptr(), d_c.data_ptr(), outs[0].data_ptr(),
                             outs[1].data_ptr(), outs[2].data_ptr(), min_hit_groups=2, thresholds=(0.0, 0.2))
    st.synchronize()
    assert st.last_deferred() == n
    want = orc.classify_batch(world["p"], world["oix"], world["parents"], bases, offsets, None, None, min_hit_groups=2,
                              thresholds=(0.0, 0.2))
    got = dict(taxon=d_t.cpu().numpy().reshape(2, n), classified=d_c.cpu().numpy().reshape(2, n), num_distinct=outs[0].cpu().numpy(),
               total_kmers=outs[1].cpu().numpy(), num_hits=outs[2].cpu().numpy())
    for key in KEYS:
        assert np.array_equal(got[key], want[key]), key


@pytest.mark.parametrize("paired", [False, True])
def test_hit_lists_of_long_reads_with_ambiguous_runs(orc, world, paired):
    """Hit lists of long PAIRED fragments come from the wave kernel, which takes a mate holding characters outside ACGTU run by run
    (valid runs on the 64-lane path, other runs of >= k characters as one ambiguous span each); unpaired ones from the segment
    kernel: per-read results and the un-merged hit lists against the oracle."""
    rng = np.random.default_rng(40 + paired)
    reads = [long_read(world["lib"], rng, int(rng.integers(1001, 9000))) for _ in range(80)]
    reads += [np.full(2000, ord("N"), np.uint8), np.concatenate([np.full(40, ord("N"), np.uint8), world["lib"].genomes[0][:1500]])]
    mates = None
    if paired:
        mates = [long_read(world["lib"], rng, int(rng.integers(200, 3000))) for _ in reads]
    bases, offsets = synth.pack(reads)
    mb = mo = None
    if paired:
        mb, mo = synth.pack(mates)
    got = world["st"].classify_batch(bases, offsets, mb, mo, thresholds=(0.0, 0.15), min_hit_groups=2)
    want = orc.classify_batch(world["p"], world["oix"], world["parents"], bases, offsets, mb, mo, min_hit_groups=2,
                              thresholds=(0.0, 0.15))
    for key in ("total_kmers", "num_hits", "num_distinct", "taxon", "classified"):
        assert np.array_equal(got[key], want[key]), key
    ho = got["hit_offsets"].astype(np.int64)
    for i in range(len(reads)):
        _, hits = orc.classify_read(world["p"], world["oix"], world["parents"], reads[i].tobytes(),
                                    None if mates is None else mates[i].tobytes(), 2, 0.0)
        g = got["hits"][ho[i]:ho[i + 1]]
        assert [(int(t), int(c)) for t, c in zip(g["taxon"], g["count"])] == hits, i


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("with_hits", [False, True])
def test_batches_of_mostly_long_fragments_have_no_first_pass(orc, world, monkeypatch, paired, with_hits):
    """A batch whose fragments average over 1000 bases is sorted by a routing kernel instead of a first pass of the lane kernel, and
    its fragments of up to 1000 bases -- empty ones, ones shorter than k, ones of exactly 1000 -- are a fifth class of the lane
    kernel's long variant (capi.hip: route_first; engine.h: HandOn): the engine's default routes on such a batch, pairs too (a pair
    counts with both mates' bases), against the oracle, and the same with a first pass."""
    for v in ("SLK_SEG_MIN_LEN", "SLK_LANE_LONG_MAX", "SLK_SEG_HITS"):
        monkeypatch.delenv(v)
    rng = np.random.default_rng(77 + 2 * paired + with_hits)
    lib = world["lib"]
    reads = [long_read(lib, rng, int(n)) for n in np.exp(rng.uniform(np.log(1001), np.log(40000), 140)).astype(np.int64)]
    reads += [long_read(lib, rng, n) for n in (1000, 1001, 4999, 5000, 5001, 8750, 8751, 999)]
    reads += synth.make_reads(lib, 60, rng, vary_length=True)                              # short ones among them
    reads += [np.zeros(0, np.uint8), synth.random_dna(34, rng), synth.random_dna(35, rng), np.full(1000, ord("N"), np.uint8)]
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    mates = mb = mo = None
    if paired:
        mates = [long_read(lib, rng, int(rng.integers(1, 2500))) if rng.random() < 0.8 else np.zeros(0, np.uint8) for _ in reads]
        mb, mo = synth.pack(mates)
    bases, offsets = synth.pack(reads)
    assert (len(bases) + (len(mb) if paired else 0)) // 1000 > len(reads)                  # (the engine's criterion)
    want = orc.classify_batch(world["p"], world["oix"], world["parents"], bases, offsets, mb, mo, min_hit_groups=2, thresholds=(0.0, 0.2))
    first = None
    for route_first in (None, "1", "0"):
        if route_first is None:
            monkeypatch.delenv("SLK_ROUTE_FIRST", raising=False)
        else:
            monkeypatch.setenv("SLK_ROUTE_FIRST", route_first)
        got = world["st"].classify_batch(bases, offsets, mb, mo, thresholds=(0.0, 0.2), min_hit_groups=2, with_hits=with_hits,
                                         with_num_hits=True)
        for key in ("total_kmers", "num_hits", "num_distinct", "taxon", "classified"):
            bad = np.nonzero(np.atleast_2d(got[key] != want[key]).any(axis=0))[0]
            assert bad.size == 0, (route_first, key, bad[:5].tolist(), [len(reads[i]) for i in bad[:5]])
        if with_hits:
            if first is None:
                first = got
                ho = got["hit_offsets"].astype(np.int64)
                for i in range(0, len(reads), 7):
                    _, hits = orc.classify_read(world["p"], world["oix"], world["parents"], reads[i].tobytes(),
                                                None if mates is None else mates[i].tobytes(), 2, 0.0)
                    g = got["hits"][ho[i]:ho[i + 1]]
                    assert [(int(t), int(c)) for t, c in zip(g["taxon"], g["count"])] == hits, (i, len(reads[i]))
            else:
                assert np.array_equal(got["hit_offsets"], first["hit_offsets"]) and np.array_equal(got["hits"], first["hits"]), route_first

"""GPU parity: every result of the HIP engine, obtained through the C ABI, is compared bit for bit with the CPU oracle
on the same seeded inputs (integer/byte/index work => exact equality; the only floating point is one binary64
multiply + ceil per read and threshold, also compared exactly)."""
import os

import numpy as np
import pytest

import synth
import taxgen

pytestmark = pytest.mark.gpu

PARAM_SETS = [
    dict(k=35, m=31, spaces=7),                                   # `slacken build` defaults (Slacken.scala:126-136)
    dict(k=35, m=31, spaces=0),
    dict(k=31, m=31, spaces=0),                                   # window of one m-mer
    dict(k=5, m=2, spaces=0, xor_mask=0, canonical=False),        # the reference's KAT configuration
    dict(k=21, m=12, spaces=5),
    dict(k=45, m=32, spaces=16),                                  # full 64-bit key, widest space mask
    dict(k=63, m=9, spaces=2, canonical=False),                   # wide window (55 m-mers)
]


def full(ps):
    d = dict(xor_mask=0xe37e28c4271b5a2d, canonical=True)
    d.update(ps)
    return d


@pytest.fixture(scope="module")
def world(orc):
    """A small synthetic library + taxonomy + engine index for the default parameters."""
    import slacken_amd
    rng = np.random.default_rng(2240)
    parents = taxgen.taxonomy(8 * 128, rng)
    p = orc.params()
    lib = synth.Library(orc, p, parents, n_genomes=12, genome_len=20000, pad_records=50000)
    ix = slacken_amd.Index(expected_records=len(lib.keys), max_taxon=len(parents) - 1)
    half = len(lib.keys) // 2
    ix.append(lib.keys[:half], lib.taxa[:half])          # chunked append, as one call per Parquet bucket file would
    ix.append(lib.keys[half:], lib.taxa[half:])
    ix.set_taxonomy(parents)
    ix.finalize()
    oix = orc.Index(1, lib.keys, lib.taxa)
    return dict(p=p, lib=lib, ix=ix, oix=oix, parents=parents, st=ix.stream())


def test_index_lookup_is_lossless(world):
    lib, ix = world["lib"], world["ix"]
    info = ix.info()
    assert info.records == len(lib.keys) and info.duplicate_keys == 0
    assert np.array_equal(ix.lookup(lib.keys), lib.taxa)
    rng = np.random.default_rng(1)
    # near misses: flip one bit of stored keys (must not alias: the table is lossless, unlike a Kraken 2 CHT)
    probe = lib.keys[rng.integers(0, len(lib.keys), 20000)] ^ (np.int64(1) << rng.integers(2, 62, 20000))
    expect = np.array([world["oix"].lookup([int(k) & (2**64 - 1)]) for k in probe], np.int32)
    assert np.array_equal(ix.lookup(probe), expect)
    assert ix.lookup(np.zeros(0, np.int64)).size == 0


def oracle_spans(orc, p, reads, mates=None):
    out, off = [], [0]
    for i, r in enumerate(reads):
        sp = orc.spans(p, r.tobytes(), None if mates is None else mates[i].tobytes())
        out += [(np.int64(np.uint64(s["key"][0])), s["kmers"], s["flag"], int(s["distinct"])) for s in sp]
        off.append(len(out))
    return off, out


@pytest.mark.parametrize("ps", PARAM_SETS, ids=lambda ps: f"k{ps['k']}m{ps['m']}s{ps['spaces']}")
def test_spans_parity(orc, ps):
    import slacken_amd
    ps = full(ps)
    p = orc.params(**ps)
    rng = np.random.default_rng(ps["k"] * 100 + ps["m"])
    parents = taxgen.taxonomy(64, rng)
    lib = synth.Library(orc, p, parents, n_genomes=3, genome_len=3000) if p.W == 1 else None
    reads = synth.make_reads(lib, 700, rng, vary_length=True, n_single=0.15, n_run=0.1, lowercase=0.1, short=0.05)
    reads += [np.frombuffer(s, np.uint8) for s in
              (b"", b"N", b"ACGT" * 50, b"A" * 200, b"ACGTU" * 30, b"ACGT" * 9 + b"-" + b"TTGCA" * 20,
               b"N" * 100, b"AATTTACTTTAGTTAC")]
    ix = slacken_amd.Index(expected_records=16, max_taxon=7, **ps)
    ix.finalize()
    st = ix.stream()
    bases, offsets = synth.pack(reads)
    got_off, got = st.spans_batch(bases, offsets)
    want_off, want = oracle_spans(orc, p, reads)
    assert got_off.tolist() == want_off
    assert [(int(s["key"]), int(s["kmers"]), int(s["flag"]), int(s["distinct"])) for s in got] == \
           [(int(a), b, c, d) for a, b, c, d in want]
    # paired: mate border span + distinct tracking across mates
    mates = [reads[(i * 7 + 3) % len(reads)] for i in range(len(reads))]
    mb, mo = synth.pack(mates)
    got_off, got = st.spans_batch(bases, offsets, mb, mo)
    want_off, want = oracle_spans(orc, p, reads, mates)
    assert got_off.tolist() == want_off
    assert [(int(s["key"]), int(s["kmers"]), int(s["flag"]), int(s["distinct"])) for s in got] == \
           [(int(a), b, c, d) for a, b, c, d in want]


def test_reference_kat_on_gpu():
    # T/kmers/minimizer/MinSplitterTest.scala:25-33 through the C ABI
    import slacken_amd
    ix = slacken_amd.Index(k=5, m=2, spaces=0, xor_mask=0, canonical=False, expected_records=16, max_taxon=7)
    ix.finalize()
    seq = b"AATTTACTTTAGTTAC"
    off, sp = ix.stream().spans_batch(np.frombuffer(seq, np.uint8), np.array([0, len(seq)], np.uint64))
    starts = np.concatenate([[0], np.cumsum(sp["kmers"])[:-1]])
    assert [seq[s:s + n + 4].decode() for s, n in zip(starts, sp["kmers"])] == \
           ["AATTT", "ATTTA", "TTTACTTT", "CTTTA", "TTTAGTTA", "GTTAC"]


def merged_lists(hit_offsets, hits):
    """TaxonCounts.fromHits (TaxonCounts.scala:31-48) over every fragment's list: adjacent entries of one taxon become one, counts summed"""
    ho = hit_offsets.astype(np.int64)
    n = len(hits)
    if n == 0:
        return np.zeros(len(ho), np.uint64), hits[:0]
    first_of_read = np.zeros(n, bool)
    first_of_read[ho[:-1][ho[:-1] < n]] = True
    start = first_of_read.copy()
    start[1:] |= hits["taxon"][1:] != hits["taxon"][:-1]
    run = np.cumsum(start) - 1
    out = np.zeros(int(run[-1]) + 1, hits.dtype)
    out["taxon"] = hits["taxon"][start]
    out["count"] = np.bincount(run, weights=hits["count"].astype(np.float64)).astype(np.int64)
    runs_before = np.concatenate([[0], np.cumsum(start)])            # runs begun before entry i
    return runs_before[ho].astype(np.uint64), out


def check_classify(orc, world, reads, mates=None, thresholds=(0.0, 0.07, 0.15, 0.5, 1.0), min_hit_groups=2):
    p, st = world["p"], world["st"]
    bases, offsets = synth.pack(reads)
    mb = mo = None
    if mates is not None:
        mb, mo = synth.pack(mates)
    got = st.classify_batch(bases, offsets, mb, mo, min_hit_groups=min_hit_groups, thresholds=thresholds)
    want = orc.classify_batch(p, world["oix"], world["parents"], bases, offsets, mb, mo,
                              min_hit_groups=min_hit_groups, thresholds=thresholds)
    for key in ("taxon", "classified", "num_distinct", "total_kmers", "num_hits"):
        assert np.array_equal(got[key], want[key]), key
    # without the hit lists the engine takes its hot path (lane-per-read kernel + deferral): same answers required
    fast = st.classify_batch(bases, offsets, mb, mo, min_hit_groups=min_hit_groups, thresholds=thresholds,
                             with_hits=False, with_num_hits=True)
    for key in ("taxon", "classified", "num_distinct", "total_kmers", "num_hits"):
        assert np.array_equal(fast[key], want[key]), "hot path: " + key
    # the same reads in the engine's 3-bit form (slk_pack_bases -> slk_classify_batch_packed): identical in everything, hit lists too
    pk = st.classify_batch(bases, offsets, mb, mo, min_hit_groups=min_hit_groups, thresholds=thresholds, packed=True)
    for key in ("taxon", "classified", "num_distinct", "total_kmers", "num_hits", "hit_offsets", "hits"):
        assert np.array_equal(pk[key], got[key]), "packed: " + key
    pkf = st.classify_batch(bases, offsets, mb, mo, min_hit_groups=min_hit_groups, thresholds=thresholds, with_hits=False,
                            with_num_hits=True, packed=True)
    for key in ("taxon", "classified", "num_distinct", "total_kmers", "num_hits"):
        assert np.array_equal(pkf[key], want[key]), "packed, hot path: " + key
    # the same lists merged on the device as TaxonCounts.fromHits merges them (slk_stream_set_merged_hits), text and packed
    want_off, want_hits = merged_lists(got["hit_offsets"], got["hits"])
    st.set_merged_hits(True)
    try:
        for packed in (False, True):
            mg = st.classify_batch(bases, offsets, mb, mo, min_hit_groups=min_hit_groups, thresholds=thresholds, packed=packed)
            for key in ("taxon", "classified", "num_distinct", "total_kmers"):
                assert np.array_equal(mg[key], want[key]), "merged lists: " + key
            assert np.array_equal(mg["hit_offsets"], want_off), "merged lists: offsets"
            assert np.array_equal(mg["hits"]["taxon"], want_hits["taxon"]) and np.array_equal(mg["hits"]["count"], want_hits["count"]), "merged lists"
    finally:
        st.set_merged_hits(False)
    # un-merged hit lists (what hitDetails / lengthString are formatted from)
    ho = got["hit_offsets"].astype(np.int64)
    for i in range(0, len(reads), max(1, len(reads) // 300)):
        _, hits = orc.classify_read(p, world["oix"], world["parents"], reads[i].tobytes(),
                                    None if mates is None else mates[i].tobytes(), min_hit_groups, thresholds[0])
        g = got["hits"][ho[i]:ho[i + 1]]
        assert [(int(t), int(c)) for t, c in zip(g["taxon"], g["count"])] == hits
    return got


def test_classify_parity_single(orc, world):
    rng = np.random.default_rng(150)
    reads = synth.make_reads(world["lib"], 6000, rng)
    got = check_classify(orc, world, reads)
    assert got["classified"][0].mean() > 0.5          # most genome-derived reads classify at confidence 0
    assert (got["num_hits"] == 0).any()                # some reads vanish (SURVEY 3.3 i)


def test_classify_parity_varied_lengths(orc, world):
    rng = np.random.default_rng(151)
    reads = synth.make_reads(world["lib"], 3000, rng, vary_length=True, n_single=0.2, n_run=0.1)
    check_classify(orc, world, reads, min_hit_groups=1)
    check_classify(orc, world, reads, min_hit_groups=3, thresholds=(0.3,))


def test_classify_parity_paired(orc, world):
    rng = np.random.default_rng(152)
    r1 = synth.make_reads(world["lib"], 3000, rng)
    r2 = synth.make_reads(world["lib"], 3000, rng, vary_length=True)
    got = check_classify(orc, world, r1, r2)
    assert (got["num_hits"] >= 1).all()                # the mate border span always exists


def test_classify_edge_batches(orc, world):
    st = world["st"]
    empty = st.classify_batch(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert empty["taxon"].shape == (1, 0) and empty["hit_offsets"].tolist() == [0]
    reads = [np.zeros(0, np.uint8)] * 5 + [np.frombuffer(b"N" * 150, np.uint8)] + [np.frombuffer(b"ACGT" * 8, np.uint8)]
    check_classify(orc, world, reads)
    check_classify(orc, world, reads, reads[::-1])
    long_read = synth.make_reads(world["lib"], 1, np.random.default_rng(9), length=15000, short=0)
    check_classify(orc, world, long_read * 3)


def test_api_errors(world):
    import slacken_amd
    with pytest.raises(slacken_amd.SlackenError) as e:
        slacken_amd.Index(k=150, m=140, spaces=0)      # five id columns (m <= 128 is served, tests/test_gpu_wide.py)
    assert e.value.code == -2
    with pytest.raises(slacken_amd.SlackenError):
        slacken_amd.Index(k=10, m=12)
    ix = slacken_amd.Index(expected_records=16, max_taxon=7)
    with pytest.raises(slacken_amd.SlackenError):  # not finalized
        ix.stream().classify_batch(np.frombuffer(b"ACGT", np.uint8), np.array([0, 4], np.uint64))
    with pytest.raises(slacken_amd.SlackenError):  # taxon above max_taxon
        ix.append(np.array([1], np.int64), np.array([9], np.int32))
    with pytest.raises(slacken_amd.SlackenError):  # cycle
        ix.set_taxonomy(np.array([0, 0, 3, 2], np.int32))
    st = world["st"]
    reads = synth.make_reads(world["lib"], 50, np.random.default_rng(3))
    bases, offsets = synth.pack(reads)
    with pytest.raises(slacken_amd.SlackenError) as e:
        st.classify_batch(bases, offsets, hits_capacity=3)
    assert e.value.code == -5


@pytest.mark.parametrize("ps", [dict(k=31, m=31, spaces=0), dict(k=21, m=12, spaces=5), dict(k=45, m=32, spaces=16),
                                dict(k=35, m=31, spaces=7, canonical=False), dict(k=40, m=25, spaces=3),
                                dict(k=31, m=15, spaces=0), dict(k=51, m=20, spaces=2)],
                         ids=lambda ps: f"k{ps['k']}m{ps['m']}s{ps['spaces']}")
def test_classify_parity_other_splitters(orc, ps):
    """Window widths other than 5 run the van-Herk variant of the lane kernel (w <= 32); non-canonical and full-width
    keys exercise the key arithmetic.  Hot path and hit-list path against the oracle."""
    import slacken_amd
    ps = full(ps)
    p = orc.params(**ps)
    rng = np.random.default_rng(ps["k"] * 7 + ps["m"])
    parents = taxgen.taxonomy(8 * 16, rng)
    lib = synth.Library(orc, p, parents, n_genomes=6, genome_len=6000, pad_records=3000)
    ix = slacken_amd.Index(expected_records=len(lib.keys), max_taxon=len(parents) - 1, **ps)
    ix.append(lib.keys, lib.taxa)
    ix.set_taxonomy(parents)
    ix.finalize()
    world = dict(p=p, st=ix.stream(), oix=orc.Index(1, lib.keys, lib.taxa), parents=parents)
    reads = synth.make_reads(lib, 1500, rng, vary_length=True, n_single=0.1, n_run=0.05)
    check_classify(orc, world, reads, thresholds=(0.0, 0.2))
    check_classify(orc, world, reads[:600], reads[600:1200], thresholds=(0.0,))


def test_many_taxa_per_read_take_the_deferred_path(orc):
    """More than 12 distinct taxa in one fragment overflow the lane kernel's per-read map: the fragment is re-done by the
    wave-per-read kernel (128-slot map).  More than 128 sends the batch through the staged kernels (unbounded map)."""
    import slacken_amd
    p = orc.params()
    rng = np.random.default_rng(99)
    parents = taxgen.taxonomy(8 * 64, rng)
    taxa = np.array(taxgen.defined_taxa(parents))
    reads = [synth.random_dna(150, rng) for _ in range(300)] + [synth.random_dna(400, rng) for _ in range(20)]
    keys, tx = [], []
    for r in reads:  # every minimizer of every read is a record with its own random taxon
        kk = orc.minimizer_keys(p, r.tobytes())
        keys.append(kk)
        tx.append(rng.choice(taxa, size=len(kk)))
    keys, idx = np.unique(np.concatenate(keys), return_index=True)
    tx = np.concatenate(tx)[idx].astype(np.int32)
    ix = slacken_amd.Index(expected_records=len(keys), max_taxon=len(parents) - 1)
    ix.append(keys, tx)
    ix.set_taxonomy(parents)
    ix.finalize()
    world = dict(p=p, st=ix.stream(), oix=orc.Index(1, keys, tx), parents=parents)
    got = check_classify(orc, world, reads, thresholds=(0.0, 0.3))
    assert (got["num_distinct"][:300] > 12).all()
    assert world["st"].last_deferred() >= 300        # (of the last call: the hot path without hit lists)
    # > 128 distinct taxa in one fragment overflow the wave kernel's map too: the batch is classified again by the staged
    # kernels with an unbounded map (check_status -> run_unbounded), with and without hit lists, sync and async entry
    big = [synth.random_dna(3000, rng), synth.random_dna(150, rng), synth.random_dna(6000, rng)] + reads[:50]
    kk = np.unique(np.concatenate([orc.minimizer_keys(p, b.tobytes()) for b in big]))
    many = rng.choice(taxa, size=len(kk), replace=len(kk) > len(taxa)).astype(np.int32)
    ix2 = slacken_amd.Index(expected_records=len(kk), max_taxon=len(parents) - 1)
    ix2.append(kk, many)
    ix2.set_taxonomy(parents)
    ix2.finalize()
    world2 = dict(p=p, st=ix2.stream(), oix=orc.Index(1, kk, many), parents=parents)
    got = check_classify(orc, world2, big, thresholds=(0.0, 0.1, 0.5))
    assert got["num_distinct"][0] > 128 and got["num_distinct"][2] > 128
    check_classify(orc, world2, big[:25], big[25:50], thresholds=(0.0,))
    # ... and as sub-batches of one host call (each sub-batch is classified again, into its own part of the batch's span arrays)
    old_sub = os.environ.get("SLK_HOST_SUBBATCH")
    os.environ["SLK_HOST_SUBBATCH"] = "7"
    try:
        check_classify(orc, world2, big, thresholds=(0.0, 0.5))
        check_classify(orc, world2, big[:25], big[25:50], thresholds=(0.0,))
    finally:
        os.environ.pop("SLK_HOST_SUBBATCH", None)
        if old_sub is not None:
            os.environ["SLK_HOST_SUBBATCH"] = old_sub
    # the same stream keeps working on ordinary batches afterwards
    check_classify(orc, world2, reads[:100], thresholds=(0.0,))
    import torch
    bases, offsets = synth.pack(big)
    d_b = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).cuda()
    d_o = torch.from_numpy(offsets.astype(np.int64)).cuda()
    R = len(big)
    outs = [torch.zeros(R, dtype=torch.int32, device="cuda") for _ in range(5)]
    d_c = torch.zeros(R, dtype=torch.uint8, device="cuda")
    st = world2["st"]
    st.classify_batch_device(d_b.data_ptr(), d_o.data_ptr(), R, int(offsets[-1]), outs[0].data_ptr(), d_c.data_ptr(),
                             outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), outs[4].data_ptr(),
                             min_hit_groups=2, thresholds=(0.0,))
    st.synchronize()
    want = orc.classify_batch(p, world2["oix"], parents, bases, offsets, thresholds=(0.0,))
    assert np.array_equal(outs[0].cpu().numpy(), want["taxon"][0]) and np.array_equal(d_c.cpu().numpy(), want["classified"][0])
    assert np.array_equal(outs[1].cpu().numpy(), want["num_distinct"]) and np.array_equal(outs[3].cpu().numpy(), want["num_hits"])
    # two calls queued before one synchronisation, the FIRST holding the overflowing fragments: both are re-run, each with its
    # own thresholds and into its own outputs
    b2, o2 = synth.pack(reads[:64])
    d_b2 = torch.from_numpy(b2).cuda()
    d_o2 = torch.from_numpy(o2.astype(np.int64)).cuda()
    t2 = torch.zeros(2 * 64, dtype=torch.int32, device="cuda")
    c2 = torch.zeros(2 * 64, dtype=torch.uint8, device="cuda")
    for t in outs:
        t.zero_()
    st.classify_batch_device(d_b.data_ptr(), d_o.data_ptr(), R, int(offsets[-1]), outs[0].data_ptr(), d_c.data_ptr(),
                             outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), outs[4].data_ptr(), thresholds=(0.0,))
    st.classify_batch_device(d_b2.data_ptr(), d_o2.data_ptr(), 64, int(o2[-1]), t2.data_ptr(), c2.data_ptr(), thresholds=(0.0, 0.3))
    st.synchronize()
    want2 = orc.classify_batch(p, world2["oix"], parents, b2, o2, thresholds=(0.0, 0.3))
    assert np.array_equal(outs[0].cpu().numpy(), want["taxon"][0]) and np.array_equal(outs[1].cpu().numpy(), want["num_distinct"])
    assert np.array_equal(t2.cpu().numpy().reshape(2, 64), want2["taxon"]) and np.array_equal(c2.cpu().numpy().reshape(2, 64), want2["classified"])
    # the host entry cut into sub-batches (the overflow is in the first of four): the re-run addresses its span scratch by the
    # batch's absolute offsets
    old = os.environ.get("SLK_HOST_SUBBATCH")
    os.environ["SLK_HOST_SUBBATCH"] = "16"
    try:
        order = list(range(3, R)) + [0, 1, 2]      # ... and once with the overflowing fragments in the LAST sub-batch
        for sel in (list(range(R)), order):
            bb, oo = synth.pack([big[i] for i in sel])
            got = world2["st"].classify_batch(bb, oo, thresholds=(0.0, 0.5), with_hits=False, with_num_hits=True)
            w = orc.classify_batch(p, world2["oix"], parents, bb, oo, thresholds=(0.0, 0.5))
            for key in ("taxon", "classified", "num_distinct", "total_kmers", "num_hits"):
                assert np.array_equal(got[key], w[key]), key
    finally:
        if old is None:
            os.environ.pop("SLK_HOST_SUBBATCH", None)
        else:
            os.environ["SLK_HOST_SUBBATCH"] = old


def test_streams_on_threads_share_one_index(orc, world):
    """The reference calls the path from many task threads with shared read-only state: one finalized index, one
    stream per thread, concurrent slk_classify_batch calls (ctypes drops the GIL) -- every thread gets the oracle's answers."""
    import threading
    rng = np.random.default_rng(77)
    ix = world["ix"]
    jobs = []
    for t in range(6):
        reads = synth.make_reads(world["lib"], 1500 + 100 * t, rng, vary_length=bool(t % 2))
        bases, offsets = synth.pack(reads)
        want = orc.classify_batch(world["p"], world["oix"], world["parents"], bases, offsets, thresholds=(0.0, 0.15))
        jobs.append((bases, offsets, want))
    errors = []

    def work(job, with_hits):
        try:
            st = ix.stream()
            for _ in range(5):
                got = st.classify_batch(job[0], job[1], thresholds=(0.0, 0.15), with_hits=with_hits)
                for key in ("taxon", "classified", "num_distinct", "total_kmers"):
                    assert np.array_equal(got[key], job[2][key]), key
            st.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(job, i % 2 == 0)) for i, job in enumerate(jobs)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


@pytest.mark.parametrize("taxonomy_first", [False, True], ids=["ids-as-given", "dense-ids"])
def test_taxon_ids_beyond_22_bits(orc, taxonomy_first):
    """The lane kernel packs (taxon << 10 | count) into one LDS word.  With the taxonomy set before slk_index_finalize the table
    is renumbered to dense internal ids and the lane kernel serves ids of 2^22 and more at full speed (nothing deferred);
    without it (taxonomy set after the records were finalized) the ids stay as given and the wave kernel takes every fragment.
    Either way every taxon that crosses the ABI -- calls, hit lists, lookups, exported records -- is the caller's id."""
    import slacken_amd
    rng = np.random.default_rng(2222)
    small = taxgen.taxonomy(8 * 16, rng)
    parents, remap = taxgen.sparse_relabel(small, 6_000_000, rng)
    p = orc.params()

    class L:
        pass
    L.genomes = [synth.random_dna(6000, rng) for _ in range(5)]
    big_taxa = np.nonzero(parents)[0]
    big_taxa = big_taxa[big_taxa > (1 << 22)]
    assert len(big_taxa) >= 5
    keys, tx = [], []
    for g, t in zip(L.genomes, big_taxa[:5]):
        kk = orc.minimizer_keys(p, g.tobytes())
        keys.append(kk)
        tx.append(np.full(len(kk), t, np.int32))
    keys, idx = np.unique(np.concatenate(keys), return_index=True)
    tx = np.concatenate(tx)[idx]
    # give some minimizers an inner node (an LCA, as a real library has): reads then hit several taxa of a lineage
    inner = np.array([t for t in np.nonzero(parents)[0] if t > (1 << 22) and (parents == t).any()], np.int32)
    tx[::7] = inner[np.arange(len(tx[::7])) % len(inner)]
    ix = slacken_amd.Index(expected_records=len(keys), max_taxon=len(parents) - 1)
    assert ix.info().taxon_bits > 22
    ix.append(keys, tx)
    if taxonomy_first:
        ix.set_taxonomy(parents)
        ix.finalize()
        assert ix.info().dense_taxa == len(taxgen.defined_taxa(parents))
        with pytest.raises(slacken_amd.SlackenError):
            ix.set_taxonomy(parents)          # the dense ids derive from it
    else:
        ix.finalize()
        ix.set_taxonomy(parents)
        assert ix.info().dense_taxa == 0
    reads = synth.make_reads(L, 800, rng)
    world = dict(p=p, st=ix.stream(), oix=orc.Index(1, keys, tx), parents=parents)
    got = check_classify(orc, world, reads, thresholds=(0.0, 0.2))
    assert (got["taxon"][0] > (1 << 22)).any()
    assert world["st"].last_deferred() == 0     # (dense ids: the lane kernel kept every fragment; ids as given: it did not run)
    assert np.array_equal(ix.lookup(keys), tx)
    ek, et = ix.export()
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ek, keys[order]) and np.array_equal(et, tx[order])


def test_c_example_runs(tmp_path):
    """examples/classify_minimal.c: the ABI from plain C (device library construction + classify with hit lists)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "classify_minimal"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "classify_minimal.c"), "-L", os.path.join(root, "slacken_amd", "lib"),
                           "-lslacken_amd", "-Wl,-rpath," + os.path.join(root, "slacken_amd", "lib"), "-o", str(exe)])
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    assert out[1].startswith("read 0: taxon 2 classified 1")      # the shared stretch: the genus (LCA of both species)
    assert out[2].startswith("read 1: taxon 3 classified 1")      # species 3's own sequence
    assert out[3].startswith("read 2: taxon 0 classified 0")


def test_classify_hits_entry(orc, world):
    """slk_classify_hits (kernel 3 alone: Classifier.classify, Classifier.scala:439-454) on caller-assembled hit lists vs the
    oracle: lists of the kind the host builds when it regroups fragments by title -- several borders, ambiguous spans,
    NONE hits, taxa in any order, an empty list."""
    import slacken_amd
    rng = np.random.default_rng(92)
    parents = world["parents"]
    defined = np.array(taxgen.defined_taxa(parents), np.int32)
    lists, flags = [], []
    for r in range(400):
        n = int(rng.integers(0, 90)) if r else 0
        pool = rng.choice(defined, size=int(rng.integers(1, 9)))
        hits, dis = [], []
        for _ in range(n):
            u = rng.random()
            if u < 0.05:
                hits.append((-2, -34)); dis.append(0)
            elif u < 0.10:
                hits.append((-1, int(rng.integers(1, 40)))); dis.append(0)
            elif u < 0.35:
                hits.append((0, int(rng.integers(1, 6)))); dis.append(int(rng.random() < 0.7))
            else:
                hits.append((int(rng.choice(pool)), int(rng.integers(1, 6)))); dis.append(int(rng.random() < 0.7))
        lists.append(hits)
        flags.append(dis)
    offs = np.cumsum([0] + [len(h) for h in lists]).astype(np.uint64)
    flat = np.array([h for hs in lists for h in hs] or [(0, 0)], slacken_amd.capi.HIT_DTYPE)[:int(offs[-1])]
    dflat = np.array([d for ds in flags for d in ds], np.uint8)
    thr = (0.0, 0.15, 0.5, 1.0)
    for mhg in (1, 2, 5):
        got = world["st"].classify_hits(offs, flat, dflat, min_hit_groups=mhg, thresholds=thr)
        for c, t in enumerate(thr):
            for r, (hits, dis) in enumerate(zip(lists, flags)):
                want = orc.classify_hits(parents, hits, dis, mhg, t)
                assert (int(got["taxon"][c, r]), bool(got["classified"][c, r])) == (want["taxon"], want["classified"]), (mhg, t, r)
                assert int(got["num_distinct"][r]) == want["num_distinct"] and int(got["total_kmers"][r]) == want["total_kmers"]
    none = world["st"].classify_hits(offs, flat, None)   # no distinct flags: nothing counts as a hit group
    assert not none["classified"].any() and not none["num_distinct"].any()


def test_host_entry_subbatches_and_pinned_buffers(orc, world):
    """The host-pointer entry cuts a large call into sub-batches (upload of one overlapping the kernels of the one before) and
    DMAs straight from and to buffers of slk_host_alloc: neither may change a result.  SLK_HOST_SUBBATCH moves the sub-batch
    size so that a small batch takes the pipelined route."""
    import slacken_amd
    from slacken_amd import capi
    rng = np.random.default_rng(77)
    reads = synth.make_reads(world["lib"], 3001, rng, vary_length=True, n_single=0.1, n_run=0.05, short=0.05)
    mates = synth.make_reads(world["lib"], 3001, rng, vary_length=True, short=0.1)
    bases, offsets = synth.pack(reads)
    mb, mo = synth.pack(mates)
    thr = (0.0, 0.3)
    keys = ("taxon", "classified", "num_distinct", "total_kmers")
    for m_b, m_o in ((None, None), (mb, mo)):
        want = orc.classify_batch(world["p"], world["oix"], world["parents"], bases, offsets, m_b, m_o, thresholds=thr)
        old = os.environ.get("SLK_HOST_SUBBATCH")
        try:
            os.environ["SLK_HOST_SUBBATCH"] = "1000000"
            whole = world["st"].classify_batch(bases, offsets, m_b, m_o, thresholds=thr, with_hits=True)   # (in one piece)
            m_off, m_hits = merged_lists(whole["hit_offsets"], whole["hits"])
            for sub in ("256", "1000", "1500"):   # 12, 4 and 3 sub-batches (the last ones ragged)
                os.environ["SLK_HOST_SUBBATCH"] = sub
                got = world["st"].classify_batch(bases, offsets, m_b, m_o, thresholds=thr, with_hits=False, with_num_hits=True)
                for k in keys + ("num_hits",):
                    assert np.array_equal(got[k], want[k]), (sub, k)
                # with hit lists: the sub-batches' spans lie in the batch's span arrays (pairs: regions by fragment number), the lists
                # are put together at the end -- as they come, merged, and from packed reads
                for packed in (False, True):
                    got = world["st"].classify_batch(bases, offsets, m_b, m_o, thresholds=thr, with_hits=True, packed=packed)
                    for k in keys + ("num_hits", "hit_offsets", "hits"):
                        assert np.array_equal(got[k], whole[k] if k in ("hit_offsets", "hits") else want[k]), (sub, packed, k)
                world["st"].set_merged_hits(True)
                try:
                    got = world["st"].classify_batch(bases, offsets, m_b, m_o, thresholds=thr, with_hits=True)
                finally:
                    world["st"].set_merged_hits(False)
                assert np.array_equal(got["hit_offsets"], m_off) and np.array_equal(got["hits"], m_hits), (sub, "merged")
            # pinned input and output buffers, pipelined and in one piece
            pb = capi.pinned_array(bases.shape, np.uint8); pb[:] = bases
            po = capi.pinned_array(offsets.shape, np.uint64); po[:] = offsets
            out = dict(taxon=capi.pinned_array((2, len(reads)), np.int32), classified=capi.pinned_array((2, len(reads)), np.uint8),
                       num_distinct=capi.pinned_array((len(reads),), np.int32), total_kmers=capi.pinned_array((len(reads),), np.int32))
            for sub in ("700", "100000"):
                os.environ["SLK_HOST_SUBBATCH"] = sub
                for a in out.values():
                    a[...] = 0
                world["st"].classify_batch(pb, po, m_b, m_o, thresholds=thr, with_hits=False, out=out)
                for k in keys:
                    assert np.array_equal(out[k], want[k]), (sub, k)
            full = world["st"].classify_batch(pb[3:], offsets[:1], with_hits=True)   # a pointer INSIDE a pinned buffer, no reads
            assert full["taxon"].shape == (1, 0)
        finally:
            if old is None:
                os.environ.pop("SLK_HOST_SUBBATCH", None)
            else:
                os.environ["SLK_HOST_SUBBATCH"] = old
    lib = slacken_amd.lib()
    assert lib.slk_host_free(12345) != 0 and b"slk_host_alloc" in lib.slk_last_error()


def test_mixed_lengths_take_the_length_bucketed_tile_order(orc, world):
    """Batches of 4096 fragments and more are classified in a length-bucketed tile order when they mix lengths (the 64 lanes
    of a wave run in lockstep): every fragment's results must still land in its own place -- single, paired, with fragments
    the lane kernel hands on (over 1000 bases), empty ones, and a ragged last window."""
    rng = np.random.default_rng(1616)
    n = 16384 + 4096 + 37
    reads = synth.make_reads(world["lib"], n, rng, vary_length=True, n_single=0.05, n_run=0.02, short=0.05)
    for i in rng.integers(0, n, 60):
        reads[i] = synth.make_reads(world["lib"], 1, rng, length=int(rng.integers(1001, 2600)), short=0)[0]
    reads[5] = np.zeros(0, np.uint8)
    check_classify(orc, world, reads, thresholds=(0.0, 0.15))
    assert world["st"].last_deferred() >= 50
    mates = synth.make_reads(world["lib"], n, rng, vary_length=True, short=0.1)
    check_classify(orc, world, reads, mates, thresholds=(0.0,))


def test_tiles_staged_by_the_whole_wave(orc, world):
    """The hot kernel fetches a tile of 64 short fragments with wave-wide loads and scans it from 2-bit codes + validity bits in LDS
    when the tile spans at most 10 880 bytes (lane.hip: SLK_PACKED_STREAM), and lane by lane otherwise.  Tiles right at that border
    (10 879 / 10 880 / 10 881 bytes), tiles of empty fragments, every byte value but the line breaks as a character (only ACGTU in
    either case are nucleotides: BitRepresentation.scala:127-143), a last tile of fewer than 64 fragments that ends with the buffer, fragments that
    start at every offset within a 16-byte word: without hit lists (the hot path) and with them, against the oracle."""
    rng = np.random.default_rng(170)
    lib = world["lib"]

    def tile_of(span):   # 64 fragments whose lengths sum to `span`
        lens = rng.integers(100, 240, 64)
        lens = np.maximum(1, (lens * (span / lens.sum())).astype(np.int64))
        lens[-1] += span - lens.sum()
        assert lens.sum() == span and lens.min() >= 1
        return synth.make_reads(lib, 64, rng, short=0.0)[:0] + [synth.make_reads(lib, 1, rng, length=int(L), short=0.0)[0] for L in lens]

    reads = []
    for span in (10879, 10880, 10881, 10880, 9600, 64):
        reads += tile_of(span)
    reads += [np.zeros(0, np.uint8)] * 64                                        # a tile of empty fragments
    odd = synth.make_reads(lib, 64, rng, length=150, short=0.0)
    allbytes = np.arange(256, dtype=np.uint8)
    allbytes[[10, 13]] = ord("N")      # (line breaks inside a read are outside the contract: getSpans takes whitespace-free reads, KeyValueIndex.scala:162)
    for i in range(0, 64, 4):                                                    # every (other) byte value inside genome-derived reads
        odd[i] = np.concatenate([odd[i][:60], allbytes[(i * 4) % 256:(i * 4) % 256 + 16], odd[i][60:]])
    odd[1] = np.frombuffer(b"acgtuACGTU" * 15, np.uint8)
    odd[2] = np.frombuffer(b"N" * 150, np.uint8)
    odd[3] = np.frombuffer(b"-" * 40 + b"ACGT" * 30, np.uint8)
    reads += odd
    reads += synth.make_reads(lib, 37, rng, vary_length=True, n_single=0.3, n_run=0.2)   # the last tile: 37 fragments, then the buffer ends
    got = check_classify(orc, world, reads, thresholds=(0.0, 0.2))
    assert got["classified"][0].mean() > 0.5
    # the same fragments shifted through the tiles: every alignment of a fragment's start within a word, tiles that mix the two routes
    for shift in (1, 7, 33):
        check_classify(orc, world, reads[shift:] + reads[:shift], thresholds=(0.0,))


def test_packed_reads_any_text(orc, world):
    """slk_classify_batch_packed on text with everything a FASTQ line can hold -- lowercase, U / u, IUPAC codes, N runs shorter than,
    equal to and longer than k, '-', digits, bytes above 127 --, packed by slk_pack_bases and, for comparison, by a plain Python
    restatement of the 3-bit form; single reads and pairs, in one piece and through the sub-batch pipeline (words of 16 bases
    straddle the sub-batches' borders), from pinned and from pageable buffers."""
    from slacken_amd import capi
    rng = np.random.default_rng(404)
    reads = synth.make_reads(world["lib"], 2500, rng, vary_length=True, n_single=0.2, n_run=0.1, short=0.05)
    junk = np.frombuffer(b"acgtuURYKMSWBDHVNn-.*0123\x80\xff", np.uint8)
    for i in rng.integers(0, len(reads), 600):
        r = reads[i].copy()
        if len(r):
            at = rng.integers(0, len(r), max(1, len(r) // 12))
            r[at] = rng.choice(junk, len(at))
            if i % 3 == 0:
                r = np.frombuffer(r.tobytes().lower(), np.uint8)
            reads[i] = r
    mates = synth.make_reads(world["lib"], 2500, rng, vary_length=True, short=0.1)
    bases, offsets = synth.pack(reads)
    mb, mo = synth.pack(mates)

    def py_pack(b):
        lut = np.full(256, 255, np.uint8)
        for ch, code in zip(b"ACGTUacgtu", [0, 1, 2, 3, 3, 0, 1, 2, 3, 3]):
            lut[ch] = code
        x = lut[b]
        ok = x != 255
        pad = (-len(b)) % 16
        xx = np.concatenate([np.where(ok, x, 0), np.zeros(pad, np.uint8)]).reshape(-1, 16).astype(np.uint32)
        codes = (xx << (2 * np.arange(16, dtype=np.uint32))).sum(1).astype(np.uint32)
        valid = (np.concatenate([ok, np.zeros(pad, bool)]).reshape(-1, 16).astype(np.uint32) << np.arange(16, dtype=np.uint32)).sum(1).astype(np.uint16)
        return codes, valid

    for b in (bases, mb):
        c, v = capi.pack_bases(b)
        pc, pv = py_pack(b)
        assert np.array_equal(c[:len(pc)], pc) and np.array_equal(v[:len(pv)], pv)
    thr = (0.0, 0.2)
    old = os.environ.get("SLK_HOST_SUBBATCH")
    try:
        for m_b, m_o in ((None, None), (mb, mo)):
            want = orc.classify_batch(world["p"], world["oix"], world["parents"], bases, offsets, m_b, m_o, thresholds=thr)
            for sub in ("300", "1000", "1000000"):
                os.environ["SLK_HOST_SUBBATCH"] = sub
                for pinned in (False, True):
                    pk = capi.pack_bases(bases, pinned=pinned) + (capi.pack_bases(m_b, pinned=pinned) if m_b is not None else ())
                    got = world["st"].classify_batch(None, offsets, None, m_o, thresholds=thr, with_hits=False, with_num_hits=True, packed=pk)
                    for k in ("taxon", "classified", "num_distinct", "total_kmers", "num_hits"):
                        assert np.array_equal(got[k], want[k]), (sub, pinned, k)
    finally:
        if old is None:
            os.environ.pop("SLK_HOST_SUBBATCH", None)
        else:
            os.environ["SLK_HOST_SUBBATCH"] = old
    empty = world["st"].classify_batch(None, np.zeros(1, np.uint64), packed=(np.zeros(1, np.uint32), np.zeros(1, np.uint16)))
    assert empty["taxon"].shape == (1, 0)


def test_a_library_that_outgrows_its_table_makes_it_grow(orc):
    """expected_records far below what is appended: the load passes what the cells' displacement field can count, and records find
    no cell within reach.  That used to fail the load (SLK_E_CAPACITY, after hours of Parquet streaming for a real library); now
    the table moves to a larger geometry and the insert that hit the limit runs again.  Every record present, every lookup exact,
    duplicate keys counted once each -- also those in the batch that ran twice --, the export equal to the input; with records and
    with sequences (the library builder), on the device and (SLK_TEST_HOST_GROW) through host memory."""
    import slacken_amd
    rng = np.random.default_rng(808)
    n = 400_000
    keys = np.unique(rng.integers(-2**62, 2**62, n, dtype=np.int64) & ~np.int64(0x33333333))
    taxa = rng.integers(1, 2000, len(keys)).astype(np.int32)
    dup_at = rng.choice(len(keys) // 2, 700, replace=False)
    for via_host in ("0", "1"):
        os.environ["SLK_GROW_VIA_HOST"] = via_host
        try:
            _grow_with_records(keys, taxa, dup_at, rng)
        finally:
            os.environ.pop("SLK_GROW_VIA_HOST", None)
    _grow_with_sequences(orc, rng)


def _grow_with_records(keys, taxa, dup_at, rng):
    import slacken_amd
    ix = slacken_amd.Index(expected_records=len(keys) // 5, max_taxon=2047)
    b0 = ix.info().buckets
    cuts = [0, len(keys) // 3, len(keys) // 2, len(keys)]
    for a, b in zip(cuts[:-1], cuts[1:]):
        k, t = keys[a:b], taxa[a:b]
        if a == cuts[2]:     # the last call repeats 700 keys of the first half (with other taxa: the first stays)
            k, t = np.concatenate([k, keys[dup_at]]), np.concatenate([t, np.full(700, 7, np.int32)])
            order = rng.permutation(len(k))
            k, t = k[order], t[order]
        ix.append(k, t)
    info = ix.info()
    assert info.buckets > b0, "the table did not have to grow: the test no longer tests anything"
    assert info.records == len(keys) and info.duplicate_keys == 700
    parents = np.zeros(2048, np.int32)
    parents[2:] = 1
    ix.set_taxonomy(parents)
    ix.finalize()
    assert np.array_equal(ix.lookup(keys), taxa)
    assert not ix.lookup(np.setdiff1d(keys[:5000] ^ np.int64(1 << 40), keys)).any()
    gk, gt = ix.export()
    assert np.array_equal(gk, keys) and np.array_equal(gt, taxa)
    ix.close()


def _grow_with_sequences(orc, rng):
    """the library builder: sequences whose minimizers outnumber expected_records several times"""
    import slacken_amd
    p = orc.params()
    parents = taxgen.taxonomy(8 * 16, rng)
    leaves = np.array(taxgen.defined_taxa(parents))[-20:]
    genomes = [synth.random_dna(30000, rng) for _ in range(12)]
    gt_ = rng.choice(leaves, len(genomes)).astype(np.int32)
    bases, offsets = synth.pack(genomes)
    want_k, want_t = orc.build_records(p, parents, bases, offsets, gt_)
    ix = slacken_amd.Index(expected_records=len(want_k) // 6, max_taxon=len(parents) - 1)
    b0 = ix.info().buckets
    ix.set_taxonomy(parents)
    ix.add_sequences(bases[:int(offsets[5])], offsets[:6], gt_[:5])
    ix.add_sequences(bases[int(offsets[5]):], offsets[5:] - offsets[5], gt_[5:])
    assert ix.info().buckets > b0
    ix.finalize()
    gk, gtx = ix.export()
    assert np.array_equal(gk, want_k) and np.array_equal(gtx, want_t)
    ix.close()

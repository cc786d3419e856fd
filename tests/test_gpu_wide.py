"""Minimizers wider than 32 nt (2..4 id columns): the wide path (wide.hip) against the oracle, which restates the
reference's multi-word NTBitArray arithmetic literally (W <= 4).  The reference's own end-to-end test draws m up to 128
(T/slacken/ClassifierTest.scala:75-130)."""
import numpy as np
import pytest

import synth
import taxgen

pytestmark = pytest.mark.gpu


def build_world(orc, k, m, spaces, canonical, rng, n_genomes=6, genome_len=4000):
    p = orc.params(k=k, m=m, spaces=spaces, canonical=canonical)
    W = (m + 31) // 32
    parents = taxgen.taxonomy(8 * 16, rng)
    taxa = np.array(taxgen.defined_taxa(parents))
    leaves = np.setdiff1d(taxa, parents[taxa])
    genomes = [synth.random_dna(genome_len, rng) for _ in range(n_genomes)]
    for g in range(1, n_genomes):
        genomes[g][500:1200] = genomes[g - 1][500:1200]          # shared stretches => LCA records
    recs = {}
    for g, t in zip(genomes, rng.choice(leaves[leaves != 1], n_genomes, replace=False)):
        for sp in orc.spans(p, g.tobytes()):
            if sp["flag"] == 1:
                key = tuple(sp["key"][:W])
                recs[key] = orc.lca(parents, recs.get(key, 0), int(t))
    keys = np.array(sorted(recs), dtype=np.uint64).reshape(-1, W).view(np.int64)
    tx = np.array([recs[tuple(int(x) for x in row)] for row in keys.view(np.uint64)], np.int32)
    return p, W, parents, genomes, keys, tx


def _param_sets():
    import os
    sets = [(40, 33, 0, True), (45, 40, 7, True), (70, 63, 10, True), (64, 64, 0, True), (80, 65, 16, False), (110, 100, 20, True),
            (158, 128, 64, True), (130, 128, 0, True), (96, 96, 48, True)]
    rng = np.random.default_rng(99)
    for _ in range(int(os.environ.get("SLK_FUZZ_SEEDS", 0)) // 8):      # soak: random wide splitters
        m = int(rng.integers(33, 129))
        W = (m + 31) // 32
        sets.append((int(rng.integers(m, m + 128 // W)), m, int(rng.integers(0, m // 2 + 1)), bool(rng.integers(0, 4))))
    return sets


@pytest.mark.parametrize("k,m,spaces,canonical", _param_sets())
def test_wide_classify_parity(orc, k, m, spaces, canonical):
    import slacken_amd
    rng = np.random.default_rng(k * 1000 + m)
    p, W, parents, genomes, keys, tx = build_world(orc, k, m, spaces, canonical, rng)
    ix = slacken_amd.Index(k=k, m=m, spaces=spaces, canonical=canonical, expected_records=len(tx), max_taxon=len(parents) - 1)
    assert ix.W == W
    ix.append(keys[:len(tx) // 2], tx[:len(tx) // 2])
    ix.append(keys[len(tx) // 2:], tx[len(tx) // 2:])
    ix.set_taxonomy(parents)
    ix.finalize()
    assert ix.info().records == len(tx)
    assert np.array_equal(ix.lookup(keys), tx)
    probe = keys.copy()                     # the lowest used bit of some word flipped: mostly absent keys, now and then a neighbour
    probe[:, -1] ^= 1 << 40                 # that is in the library too
    known = {tuple(r): int(t) for r, t in zip(keys.tolist(), tx)}
    want_probe = np.array([known.get(tuple(r), 0) for r in probe.tolist()], np.int32)
    assert np.array_equal(ix.lookup(probe), want_probe) and (want_probe == 0).mean() > 0.9
    oix = orc.Index(W, keys, tx)

    class L:
        pass
    L.genomes = genomes
    reads = synth.make_reads(L, 600, rng, length=int(rng.integers(k + 5, 3 * k + 100)), vary_length=True, n_single=0.1, n_run=0.05)
    mates = synth.make_reads(L, 600, rng, length=int(rng.integers(k, 2 * k + 50)), vary_length=True, short=0.1)
    st = ix.stream()
    for mb_mo in (None, synth.pack(mates)):
        bases, offsets = synth.pack(reads)
        mb, mo = mb_mo if mb_mo else (None, None)
        want = orc.classify_batch(p, oix, parents, bases, offsets, mb, mo, min_hit_groups=2, thresholds=(0.0, 0.1, 0.5))
        for with_hits in (True, False):
            got = st.classify_batch(bases, offsets, mb, mo, min_hit_groups=2, thresholds=(0.0, 0.1, 0.5), with_hits=with_hits)
            for key in ("taxon", "classified", "num_distinct", "total_kmers"):
                assert np.array_equal(got[key], want[key]), (key, with_hits, mb is not None)
        assert np.array_equal(got["num_hits"] if "num_hits" in got else want["num_hits"], want["num_hits"])
        full = st.classify_batch(bases, offsets, mb, mo, thresholds=(0.0,), with_hits=True)
        ho = full["hit_offsets"].astype(np.int64)
        for i in range(0, len(reads), 25):
            _, hits = orc.classify_read(p, oix, parents, reads[i].tobytes(), None if mb is None else mates[i].tobytes(), 2, 0.0)
            g = full["hits"][ho[i]:ho[i + 1]]
            assert [(int(t), int(c)) for t, c in zip(g["taxon"], g["count"])] == hits
        # the lists merged on the device (slk_stream_set_merged_hits) on the staged route of the wide minimizers too
        from test_gpu_parity import merged_lists
        m_off, m_hits = merged_lists(full["hit_offsets"], full["hits"])
        st.set_merged_hits(True)
        try:
            mg = st.classify_batch(bases, offsets, mb, mo, thresholds=(0.0,), with_hits=True)
        finally:
            st.set_merged_hits(False)
        assert np.array_equal(mg["hit_offsets"], m_off) and np.array_equal(mg["hits"], m_hits)
    assert want["classified"][0].mean() > 0.05


def test_wide_limits(orc):
    import slacken_amd
    with pytest.raises(slacken_amd.SlackenError) as e:       # five id columns
        slacken_amd.Index(k=140, m=130)
    assert e.value.code == slacken_amd.capi.E_UNSUPPORTED
    with pytest.raises(slacken_amd.SlackenError) as e:       # window of 40 m-mers with 4 id columns
        slacken_amd.Index(k=139, m=100)
    assert e.value.code == slacken_amd.capi.E_UNSUPPORTED
    ix = slacken_amd.Index(k=50, m=40, expected_records=16, max_taxon=7)
    ix.set_taxonomy(np.array([0, 0, 1, 1], np.int32))
    ix.finalize()
    with pytest.raises(slacken_amd.SlackenError):            # one key word per span: the wide entry is slk_spans_batch_wide
        ix.stream().spans_batch(np.frombuffer(b"ACGT" * 20, np.uint8), np.array([0, 80], np.uint64))
    with pytest.raises(slacken_amd.SlackenError):            # the sharded entries are one-column only
        slacken_amd.Index(k=50, m=40, expected_records=16, max_taxon=7).set_shard(0, 2)


@pytest.mark.parametrize("k,m,spaces,canonical", [(45, 40, 7, True), (70, 63, 10, True), (80, 65, 16, False), (110, 100, 20, True), (158, 128, 64, True)])
def test_wide_library_construction_spans_and_export(orc, k, m, spaces, canonical):
    """The rest of the reference's surface for idLongs > 1 (KeyValueIndex.scala:49 carries the id columns everywhere): library
    construction from taxon-labelled sequences (makeRecords: minimizers of the sequences split around non-nucleotides, LCA per
    minimizer), the records back out of the table, and getSpans with all key words -- against the oracle's multi-word arithmetic."""
    import slacken_amd
    rng = np.random.default_rng(7 * k + m)
    p, W, parents, genomes, keys, tx = build_world(orc, k, m, spaces, canonical, rng, n_genomes=5, genome_len=3000)
    taxa = np.array(taxgen.defined_taxa(parents))
    leaves = np.setdiff1d(taxa, parents[taxa])
    g_taxa = rng.choice(leaves[leaves != 1], len(genomes), replace=False).astype(np.int32)
    seqs = []
    for g in genomes:                       # Ns, a run of them, lower case: the sequences are split there
        s = g.copy()
        s[rng.integers(0, len(s), 3)] = ord("N")
        a = int(rng.integers(100, len(s) - 200))
        s[a:a + 50] = ord("N")
        seqs.append(np.frombuffer(s.tobytes().lower(), np.uint8) if rng.random() < 0.3 else s)
    seqs.append(synth.random_dna(k - 1, rng))      # shorter than k: nothing
    seq_taxa = list(g_taxa) + [int(g_taxa[0])]
    recs = {}
    for s, t in zip(seqs, seq_taxa):
        for sp in orc.spans(p, s.tobytes()):
            if sp["flag"] == 1:
                key = tuple(sp["key"][:W])
                recs[key] = orc.lca(parents, recs.get(key, 0), int(t))
    want_keys = np.array(sorted(recs), dtype=np.uint64).reshape(-1, W)
    want_tx = np.array([recs[tuple(int(x) for x in row)] for row in want_keys], np.int32)
    bases, offsets = synth.pack(seqs)
    for split in (len(seqs), 2):            # one call, and the sequences over several calls (same records)
        ix = slacken_amd.Index(k=k, m=m, spaces=spaces, canonical=canonical, expected_records=len(want_tx) + 64, max_taxon=len(parents) - 1)
        ix.set_taxonomy(parents)
        for a in range(0, len(seqs), split):
            b = min(len(seqs), a + split)
            ix.add_sequences(bases[int(offsets[a]):int(offsets[b])], offsets[a:b + 1] - offsets[a], seq_taxa[a:b])
        ix.finalize()
        gk, gt = ix.export()
        assert ix.info().records == len(want_tx)
        assert np.array_equal(gk.view(np.uint64), want_keys) and np.array_equal(gt, want_tx)
        assert np.array_equal(ix.lookup(want_keys.view(np.int64)), want_tx)
    # getSpans with every key word, single and paired
    class L:
        pass
    L.genomes = genomes
    reads = synth.make_reads(L, 150, rng, length=2 * k, vary_length=True, n_single=0.2, n_run=0.1)
    mates = synth.make_reads(L, 150, rng, length=k + 30, vary_length=True, short=0.1)
    st = ix.stream()
    for mb_mo in (None, synth.pack(mates)):
        rb, ro = synth.pack(reads)
        mb, mo = mb_mo if mb_mo else (None, None)
        so, spans, skeys = st.spans_batch_wide(rb, ro, mb, mo)
        so = so.astype(np.int64)
        for i in range(len(reads)):
            want = orc.spans(p, reads[i].tobytes(), None if mb is None else mates[i].tobytes())
            got = spans[so[i]:so[i + 1]]
            assert len(got) == len(want)
            for j, w in enumerate(want):
                assert (int(got["kmers"][j]), int(got["flag"][j]), bool(got["distinct"][j])) == (w["kmers"], w["flag"], w["distinct"])
                if w["flag"] == 1:
                    assert tuple(int(x) for x in skeys[so[i] + j].view(np.uint64)) == tuple(w["key"][:W])
                    assert int(got["key"][j]) == int(skeys[so[i] + j][0])


def test_wide_library_through_the_cli(orc, tmp_path):
    """A library with two id columns (m = 40) in Slacken's on-disk layout, classified by `slacken-amd classify`: the Parquet
    reader delivers (id1, id2) rows, the engine takes its wide path, the lines equal the oracle's."""
    import os
    import sys
    from test_host_cli import ROOT, write_taxonomy
    from test_host_classify_gpu import classify, read_out
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import parquet_to_slkrec as conv
    rng = np.random.default_rng(640)
    k, m, spaces = 47, 40, 6
    p, W, parents, genomes, keys, tx = build_world(orc, k, m, spaces, True, rng)
    loc = str(tmp_path / "wide_lib")
    conv.write_parquet_dir(loc, keys, tx, buckets=3)
    with open(loc + ".properties", "w") as f:
        f.write(f"k={k}\nm={m}\nbuckets=3\nversion=1\nsplitter=randomXOR\nminimizerSpaces={spaces}\ncanonical=true\n")
    write_taxonomy(loc + "_taxonomy", parents, np.random.default_rng(1))

    class L:
        pass
    L.genomes = genomes
    reads = synth.make_reads(L, 400, rng, length=140, vary_length=True, n_single=0.1)
    fq = tmp_path / "r.fq"
    with open(fq, "w") as f:
        for i, r in enumerate(reads):
            s = r.tobytes().decode()
            f.write(f"@w{i}\n{s}\n+\n{'I' * len(s)}\n")
    out = tmp_path / "wide_out"
    classify("-i", loc, "-o", out, "-c", "0.1", fq)
    oix = orc.Index(W, keys, tx)
    want = []
    for i, r in enumerate(reads):
        res, hits = orc.classify_read(p, oix, parents, r.tobytes(), None, 2, 0.1)
        if hits:
            want.append(orc.output_line(res["classified"], f"w{i}", res["taxon"], hits, k))
    assert read_out(f"{out}_c0.1") == want and any(l.startswith("C") for l in want)

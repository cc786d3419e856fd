"""Size-independent properties of the classify path at BASELINE.json's full sizes, which the CPU oracle cannot reach: a
1.0e10-record table (128 GiB) and 1.0e7 x 150 bp reads (SLK_FULLSIZE=0: a 2^30-record table and 2M reads).  Integer work => exact equality everywhere.
  * determinism and batch-split invariance
  * reverse-complement invariance (canonical minimizers: MinSplitterProps.scala:101-114 lifted to the whole path)
  * the two independently written kernels (lane-per-read hot path, wave-per-read path) agree read for read
  * a poly-A read is one super-mer of L-k+1 k-mers; per-taxon read counts add up to the classified reads
and ONE comparison with the oracle itself at full size: a sample of the reads (the N-bearing ones among them) is classified by the
oracle against a mini-index that holds, for exactly the sample's minimizers, what POINT LOOKUPS of the big table returned (a different
kernel from the classify kernels' probe) -- so bucket addressing at 2^31 buckets / 128 GiB offsets is checked against the oracle too."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    import torch
    import bench
    import slacken_amd
    full = os.environ.get("SLK_FULLSIZE", "1") == "1"
    n_records, n_reads = (int(1e10), int(1e7)) if full else (1 << 30, 2_000_000)
    # (SLK_FULLSIZE_RECORDS=2e10: the largest table the bench has a line for -- 213 GiB, 3.6e9 buckets, load 0.70, chains of 32
    #  buckets; run once per round as a soak, gpurun_out/fullsize_2e10.log)
    n_records = int(float(os.environ.get("SLK_FULLSIZE_RECORDS", n_records)))
    dev = torch.device("cuda", 0)
    parents, taxa, leaves = bench.build_taxonomy()
    rng = np.random.default_rng(224)
    G, GL = 256, 1 << 20    # 2.7e8 genome bases: hit-bearing records spread over the whole table, as in the bench
    genome_taxa = rng.choice(leaves, size=G, replace=False).astype(np.int32)
    genome_cat = bench.make_genomes_device(torch, G, GL, 225, dev)
    ix = slacken_amd.Index(expected_records=n_records, max_taxon=bench.TAX_EXTENT - 1)
    ix.set_shard(0, 1)      # (shard 0 of 1 = the whole library: lets the C-ABI shard set take this index as its one member)
    ix.set_taxonomy(parents)
    ix.add_sequences_device(genome_cat.data_ptr(), np.arange(0, (G + 1) * GL, GL, dtype=np.uint64), genome_taxa)
    n_genome = int(ix.info().records)
    smask = ((2**62 - 1) & ~0x0CCCCCCC) << 2
    smask -= 1 << 64
    d_taxa = torch.from_numpy(taxa).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    for s in range(0, n_records - n_genome, 1 << 27):
        n = min(1 << 27, n_records - n_genome - s)
        keys = ((torch.randint(0, 2**32, (n,), generator=gen, device=dev, dtype=torch.int64) << 32) |
                torch.randint(0, 2**32, (n,), generator=gen, device=dev, dtype=torch.int64)) & smask
        tx = d_taxa[torch.randint(0, len(taxa), (n,), generator=gen, device=dev)]
        torch.cuda.synchronize()
        ix.append_device(keys.data_ptr(), tx.data_ptr(), n)
        del keys, tx
    ix.finalize()
    info = ix.info()
    print(f"[fullsize] {info.records} records, {info.table_bytes / 2**30:.1f} GiB, max displacement {info.max_displacement}", file=sys.stderr)
    bases, offsets = bench.make_reads_device(torch, genome_cat, GL, G, n_reads, 150, dev)
    return dict(torch=torch, ix=ix, st=ix.stream(), bases=bases, offsets=offsets, R=n_reads, dev=dev,
                ntax=bench.TAX_EXTENT, parents=parents)


def run(big, bases, offsets, R, thresholds=(0.0, 0.1)):
    torch = big["torch"]
    C = len(thresholds)
    out = dict(taxon=torch.zeros(C * R, dtype=torch.int32, device=big["dev"]),
               cls=torch.zeros(C * R, dtype=torch.uint8, device=big["dev"]),
               nd=torch.zeros(R, dtype=torch.int32, device=big["dev"]),
               tk=torch.zeros(R, dtype=torch.int32, device=big["dev"]),
               nh=torch.zeros(R, dtype=torch.int32, device=big["dev"]),
               np_=torch.zeros(R, dtype=torch.int32, device=big["dev"]))
    big["st"].classify_batch_device(bases.data_ptr(), offsets.data_ptr(), R, R * 150, out["taxon"].data_ptr(),
                                    out["cls"].data_ptr(), out["nd"].data_ptr(), out["tk"].data_ptr(),
                                    out["nh"].data_ptr(), out["np_"].data_ptr(), thresholds=thresholds)
    big["st"].synchronize()
    return out


def same(a, b, keys=("taxon", "cls", "nd", "tk", "nh", "np_")):
    return all(bool((a[k] == b[k]).all()) for k in keys)


def test_deterministic_and_split_invariant(big):
    torch = big["torch"]
    R = big["R"]
    a = run(big, big["bases"], big["offsets"], R)
    b = run(big, big["bases"], big["offsets"], R)
    assert same(a, b)
    # four sub-batches give the same per-read answers as one batch
    q = R // 4
    for j in range(4):
        sub_b = big["bases"][j * q * 150:]
        sub_o = big["offsets"][:q + 1].clone()
        part = run(big, sub_b, sub_o, q, thresholds=(0.0,))
        assert bool((part["taxon"] == a["taxon"][j * q:(j + 1) * q]).all())
        assert bool((part["nd"] == a["nd"][j * q:(j + 1) * q]).all())
    # accounting: k-mers of an all-valid 150 bp read = 116; probes <= spans; classified <=> taxon != 0
    valid = a["nh"] > 0
    assert int(a["tk"][valid].max()) <= 116 and int(a["tk"].min()) >= 0
    assert bool((a["np_"] <= a["nh"]).all())
    assert bool(((a["taxon"][:R] != 0) == (a["cls"][:R] != 0)).all())
    counts = torch.bincount(a["taxon"][:R].long(), minlength=big["ntax"])
    assert int(counts[1:].sum()) == int(a["cls"][:R].sum())


def test_reverse_complement_invariance(big):
    torch = big["torch"]
    R = min(big["R"], 1_000_000)
    b = big["bases"][:R * 150].view(R, 150)
    comp = torch.arange(256, dtype=torch.uint8, device=big["dev"])
    for x, y in zip(b"ACGT", b"TGCA"):
        comp[x] = y
    rc = torch.cat([comp[b.long()].flip(1).reshape(-1), torch.full((64,), 65, dtype=torch.uint8, device=big["dev"])])
    fwd = run(big, big["bases"], big["offsets"][:R + 1], R)
    rev = run(big, rc, big["offsets"][:R + 1], R)
    assert same(fwd, rev, keys=("taxon", "cls", "nd", "tk", "nh", "np_"))


def test_lane_and_wave_kernels_agree(big):
    # host-pointer entry with hit lists runs the wave-per-read kernel; compare with the device-pointer hot path
    R = 200_000
    host_b = big["bases"][:R * 150].cpu().numpy()
    host_o = np.arange(0, (R + 1) * 150, 150, dtype=np.uint64)
    wave = big["st"].classify_batch(host_b, host_o, thresholds=(0.0, 0.1), with_hits=True)
    lane = run(big, big["bases"], big["offsets"][:R + 1], R)
    assert np.array_equal(wave["taxon"].reshape(-1), lane["taxon"].cpu().numpy())
    assert np.array_equal(wave["classified"].reshape(-1), lane["cls"].cpu().numpy())
    assert np.array_equal(wave["num_distinct"], lane["nd"].cpu().numpy())
    assert np.array_equal(wave["total_kmers"], lane["tk"].cpu().numpy())
    assert np.array_equal(wave["num_hits"], lane["nh"].cpu().numpy())


def test_long_reads_of_mixed_lengths_all_routes_agree(big, orc, monkeypatch):
    """A batch of ~1 Gbp of fragments of 200 .. 50 000 bases (log-normal, as a nanopore run's) takes the routing kernel, the long
    variant of the lane kernel, the wave kernel in length order and -- when its threshold is moved down -- the segment kernel, three of
    them side by side on two streams: at that size the passes really overlap.  Every arrangement of the routes must give the same rows
    (integer work: exact), and a sample of the fragments is classified by the oracle through point lookups of the big table."""
    torch = big["torch"]
    dev = big["dev"]
    full = os.environ.get("SLK_FULLSIZE", "1") == "1"
    rng = np.random.default_rng(41)
    target = 1_000_000_000 if full else 200_000_000
    lens = np.clip(rng.lognormal(np.log(3000), 0.9, 2_000_000), 200, 50000).astype(np.int64)
    lens = lens[:np.searchsorted(np.cumsum(lens), target)]
    lens[:8] = (1000, 1001, 4999, 5000, 35, 34, 8750, 50000)
    R = len(lens)
    offs = np.zeros(R + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    total = int(offs[-1])
    # the fragments are stretches of the 150-base reads' buffer (reads of the library's genomes and random ones, some with Ns), cut
    # anywhere: a long fragment then runs over several source reads -- several taxa, N runs, misses
    src_len = big["R"] * 150
    starts = torch.from_numpy(rng.integers(0, src_len - 50001, R)).to(dev)
    d_offs = torch.from_numpy(offs).to(dev)
    pos = torch.arange(total, device=dev)
    rid = torch.searchsorted(d_offs, pos, right=True) - 1
    d_b = big["bases"][starts[rid] + (pos - d_offs[rid])].contiguous()
    del pos, rid

    def classify(env):
        for v in ("SLK_SEG_MIN_LEN", "SLK_LANE_LONG_MAX", "SLK_ROUTE_FIRST", "SLK_FORCE_WAVE"):
            monkeypatch.delenv(v, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = dict(taxon=torch.zeros(2 * R, dtype=torch.int32, device=dev), cls=torch.zeros(2 * R, dtype=torch.uint8, device=dev),
                   nd=torch.zeros(R, dtype=torch.int32, device=dev), tk=torch.zeros(R, dtype=torch.int32, device=dev),
                   nh=torch.zeros(R, dtype=torch.int32, device=dev))
        big["st"].classify_batch_device(d_b.data_ptr(), d_offs.data_ptr(), R, total, out["taxon"].data_ptr(), out["cls"].data_ptr(),
                                        out["nd"].data_ptr(), out["tk"].data_ptr(), out["nh"].data_ptr(), thresholds=(0.0, 0.1))
        big["st"].synchronize()
        return out

    keys = ("taxon", "cls", "nd", "tk", "nh")
    base = classify({})                                             # default routes: routing kernel first
    again = classify({})
    assert same(base, again, keys)
    for name, env in (("first pass instead of the routing kernel", dict(SLK_ROUTE_FIRST="0")),
                      ("segment kernel from 12 000 bases", dict(SLK_SEG_MIN_LEN="12000")),
                      ("wave kernel for everything over 1000 bases", dict(SLK_SEG_MIN_LEN="0", SLK_LANE_LONG_MAX="0")),
                      ("no long variant, segment kernel from 5 000", dict(SLK_LANE_LONG_MAX="0", SLK_SEG_MIN_LEN="5000"))):
        other = classify(env)
        for k in keys:
            bad = torch.nonzero(other[k][:R] != base[k][:R]).flatten()[:5].cpu().tolist()
            assert not bad and bool((other[k] == base[k]).all()), (name, k, bad, [int(lens[i]) for i in bad])
    # a sample against the oracle (its minimizers looked up in the big table by the point-lookup kernel)
    pick = np.unique(np.concatenate([np.arange(8), rng.choice(R, 300, replace=False)]))
    h_reads = [d_b[int(offs[i]):int(offs[i + 1])].cpu().numpy() for i in pick]
    p = orc.params()
    mk = np.unique(np.concatenate([orc.minimizer_keys(p, r.tobytes()) for r in h_reads if len(r) >= 35] or [np.zeros(0, np.int64)]))
    taxa = big["ix"].lookup(mk)
    oix = orc.Index(1, mk[taxa != 0], taxa[taxa != 0])
    sb = np.concatenate(h_reads)
    so = np.cumsum([0] + [len(r) for r in h_reads]).astype(np.uint64)
    want = orc.classify_batch(p, oix, big["parents"], sb, so, thresholds=(0.0, 0.1))
    idx = torch.from_numpy(pick).to(dev)
    for c in range(2):
        assert np.array_equal(base["taxon"][c * R + idx].cpu().numpy(), want["taxon"][c])
        assert np.array_equal(base["cls"][c * R + idx].cpu().numpy(), want["classified"][c])
    assert np.array_equal(base["nd"][idx].cpu().numpy(), want["num_distinct"])
    assert np.array_equal(base["tk"][idx].cpu().numpy(), want["total_kmers"])
    assert np.array_equal(base["nh"][idx].cpu().numpy(), want["num_hits"])
    assert int((want["num_distinct"] > 1).sum()) > 50          # (fragments over several source reads: several taxa each)


def test_pairs_lane_and_wave_kernels_agree_and_a_sample_against_the_oracle(big, orc):
    """2 x 150-base pairs at full table size: the device entry (lane kernel, both mates in one lane) against the host entry with hit
    lists (wave kernel) pair for pair, and a sample of the pairs classified by the oracle through point lookups of the big table."""
    torch = big["torch"]
    dev = big["dev"]
    R = min(big["R"] // 2, 400_000)
    d_b1 = big["bases"][:R * 150]
    d_b2 = big["bases"][R * 150:2 * R * 150].contiguous()
    d_o = big["offsets"][:R + 1]
    thr = (0.0, 0.1)
    out = dict(taxon=torch.zeros(2 * R, dtype=torch.int32, device=dev), cls=torch.zeros(2 * R, dtype=torch.uint8, device=dev),
               nd=torch.zeros(R, dtype=torch.int32, device=dev), tk=torch.zeros(R, dtype=torch.int32, device=dev),
               nh=torch.zeros(R, dtype=torch.int32, device=dev))
    big["st"].classify_batch_device(d_b1.data_ptr(), d_o.data_ptr(), R, R * 150, out["taxon"].data_ptr(), out["cls"].data_ptr(),
                                    out["nd"].data_ptr(), out["tk"].data_ptr(), out["nh"].data_ptr(), d_mate_bases=d_b2.data_ptr(),
                                    d_mate_offsets=d_o.data_ptr(), total_mate_bases=R * 150, thresholds=thr)
    big["st"].synchronize()
    h1, h2 = d_b1.cpu().numpy(), d_b2.cpu().numpy()
    ho = np.arange(0, (R + 1) * 150, 150, dtype=np.uint64)
    wave = big["st"].classify_batch(h1, ho, h2, ho, thresholds=thr, with_hits=True)
    assert np.array_equal(wave["taxon"].reshape(-1), out["taxon"].cpu().numpy())
    assert np.array_equal(wave["classified"].reshape(-1), out["cls"].cpu().numpy())
    assert np.array_equal(wave["num_distinct"], out["nd"].cpu().numpy()) and np.array_equal(wave["total_kmers"], out["tk"].cpu().numpy())
    assert np.array_equal(wave["num_hits"], out["nh"].cpu().numpy())
    rng = np.random.default_rng(8)
    pick = np.sort(rng.choice(R, 3000, replace=False))
    r1 = h1.reshape(R, 150)[pick]
    r2 = h2.reshape(R, 150)[pick]
    p = orc.params()
    mk = np.unique(np.concatenate([orc.minimizer_keys(p, bytes(x)) for x in r1] + [orc.minimizer_keys(p, bytes(x)) for x in r2]))
    taxa = big["ix"].lookup(mk)
    oix = orc.Index(1, mk[taxa != 0], taxa[taxa != 0])
    so = np.arange(0, (len(pick) + 1) * 150, 150, dtype=np.uint64)
    want = orc.classify_batch(p, oix, big["parents"], r1.reshape(-1), so, r2.reshape(-1), so, thresholds=thr)
    for c in range(2):
        assert np.array_equal(wave["taxon"][c][pick], want["taxon"][c]) and np.array_equal(wave["classified"][c][pick], want["classified"][c])
    assert np.array_equal(wave["num_distinct"][pick], want["num_distinct"]) and np.array_equal(wave["total_kmers"][pick], want["total_kmers"])
    assert np.array_equal(wave["num_hits"][pick], want["num_hits"])


def test_host_entry_at_its_default_subbatch_size(big):
    """1.5 M reads through slk_classify_batch and slk_classify_batch_packed -- cut into sub-batches of 2^19 whose upload overlaps the
    kernels of the one before and whose rows come down beside the next one's -- from pageable and from pinned caller buffers: the
    same answers as the device entry on the resident copy of the reads."""
    from slacken_amd import capi
    R = min(big["R"], 1_500_000)
    host_b = big["bases"][:R * 150].cpu().numpy()
    host_o = np.arange(0, (R + 1) * 150, 150, dtype=np.uint64)
    dev = run(big, big["bases"], big["offsets"][:R + 1], R)
    got = big["st"].classify_batch(host_b, host_o, thresholds=(0.0, 0.1), with_hits=False, with_num_hits=True)
    pb = capi.pinned_array(host_b.shape, np.uint8); pb[:] = host_b
    po = capi.pinned_array(host_o.shape, np.uint64); po[:] = host_o
    out = dict(taxon=capi.pinned_array((2, R), np.int32), classified=capi.pinned_array((2, R), np.uint8),
               num_distinct=capi.pinned_array((R,), np.int32), total_kmers=capi.pinned_array((R,), np.int32))
    big["st"].classify_batch(pb, po, thresholds=(0.0, 0.1), with_hits=False, out=out)
    out2 = dict(taxon=capi.pinned_array((2, R), np.int32), classified=capi.pinned_array((2, R), np.uint8),
                num_distinct=capi.pinned_array((R,), np.int32), total_kmers=capi.pinned_array((R,), np.int32))
    big["st"].classify_batch(None, po, thresholds=(0.0, 0.1), with_hits=False, out=out2, packed=capi.pack_bases(host_b, pinned=True))
    for res in (got, out, out2):
        assert np.array_equal(res["taxon"].reshape(-1), dev["taxon"].cpu().numpy())
        assert np.array_equal(res["classified"].reshape(-1), dev["cls"].cpu().numpy())
        assert np.array_equal(res["num_distinct"], dev["nd"].cpu().numpy())
        assert np.array_equal(res["total_kmers"], dev["tk"].cpu().numpy())
    assert np.array_equal(got["num_hits"], dev["nh"].cpu().numpy())


@pytest.mark.parametrize("paired", [False, True])
def test_host_entry_with_hit_lists_at_its_default_subbatch_size(big, monkeypatch, paired):
    """1.3 M fragments with hit lists through slk_classify_batch, un-merged and merged on the device (slk_stream_set_merged_hits): the
    sub-batches of 2^19 leave their spans in the batch's span arrays (pairs: regions by the fragment's number in the BATCH) and the
    lists are put together at the end -- the same rows and lists as the call in one piece, and the rows of the device entry."""
    from test_gpu_parity import merged_lists
    R = min(big["R"] // 2, 1_300_000)
    host_b = big["bases"][:R * 150].cpu().numpy()
    host_o = np.arange(0, (R + 1) * 150, 150, dtype=np.uint64)
    mb = mo = None
    if paired:   # (mates of 100 bases: stretches of the second half of the reads)
        mb = big["bases"][R * 150:R * 150 + R * 100].cpu().numpy()
        mo = np.arange(0, (R + 1) * 100, 100, dtype=np.uint64)
    st = big["st"]
    monkeypatch.setenv("SLK_HOST_SUBBATCH", "100000000")
    whole = st.classify_batch(host_b, host_o, mb, mo, thresholds=(0.0, 0.1), with_hits=True)
    monkeypatch.delenv("SLK_HOST_SUBBATCH")
    parts = st.classify_batch(host_b, host_o, mb, mo, thresholds=(0.0, 0.1), with_hits=True)
    for k in ("taxon", "classified", "num_distinct", "total_kmers", "hit_offsets", "hits"):
        assert np.array_equal(parts[k], whole[k]), k
    m_off, m_hits = merged_lists(whole["hit_offsets"], whole["hits"])
    st.set_merged_hits(True)
    try:
        mg = st.classify_batch(host_b, host_o, mb, mo, thresholds=(0.0, 0.1), with_hits=True)
    finally:
        st.set_merged_hits(False)
    assert np.array_equal(mg["hit_offsets"], m_off) and np.array_equal(mg["hits"], m_hits)
    if not paired:
        dev = run(big, big["bases"], big["offsets"][:R + 1], R)
        assert np.array_equal(parts["taxon"].reshape(-1), dev["taxon"].cpu().numpy()) and np.array_equal(parts["num_hits"], dev["nh"].cpu().numpy())


def test_sharded_pipeline_of_several_batches_against_the_local_kernel(big):
    """Seven batches of 1.4 M reads (the last one short) through the table-sharded pipeline at world = 1 -- one step kernel per batch
    carrying EMIT(t), LOOKUP(t - 2) and APPLY(t - 4), batches of different sizes in flight at once -- against the local kernel's rows
    for the same reads in the 1.0e10-record table."""
    from slacken_amd.sharded import ShardedClassifier
    torch = big["torch"]
    R = big["R"]
    thr = (0.0, 0.1)
    local = run(big, big["bases"], big["offsets"], R, thresholds=thr)
    per = 1_400_000 if R >= 9_000_000 else R // 7 + 1
    cuts = list(range(0, R, per)) + [R]
    batches = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        n = b - a
        d_o = torch.arange(0, (n + 1) * 150, 150, dtype=torch.int64, device=big["dev"])
        batches.append((big["bases"][a * 150:b * 150 if b < R else None], d_o, n, n * 150, None))
    sc = ShardedClassifier(big["ix"], 0, 1, None, big["dev"])
    try:
        outs = sc.classify_many(batches, thresholds=thr)
        assert outs is not None and len(outs) == len(batches)
        for (a, b), o in zip(zip(cuts[:-1], cuts[1:]), outs):
            n = b - a
            assert int(o["deferred"]) == 0 if "deferred" in o else True
            for c in range(2):
                assert bool((o["taxon"][c * n:(c + 1) * n] == local["taxon"][c * R + a:c * R + b]).all()), (a, c)
                assert bool((o["classified"][c * n:(c + 1) * n] == local["cls"][c * R + a:c * R + b]).all()), (a, c)
            assert bool((o["num_distinct"][:n] == local["nd"][a:b]).all()) and bool((o["total_kmers"][:n] == local["tk"][a:b]).all())
    finally:
        sc.close()


def test_shard_set_rounds_against_the_local_kernel(big):
    """The C-ABI route of the table-sharded mode (slk_shardset_classify_rounds, device-resident batches, one member): seven rounds of
    different sizes in one pipelined call against the local kernel's rows."""
    from slacken_amd import capi
    torch = big["torch"]
    R = big["R"]
    thr = (0.0, 0.1)
    local = run(big, big["bases"], big["offsets"], R, thresholds=thr)
    per = 1_400_000 if R >= 9_000_000 else R // 7 + 1
    cuts = list(range(0, R, per)) + [R]
    rounds, keep = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        n = b - a
        d_o = torch.arange(0, (n + 1) * 150, 150, dtype=torch.int64, device=big["dev"])
        d_b = big["bases"][a * 150:b * 150].contiguous()          # (exactly offsets[n] bytes)
        out = dict(taxon=torch.zeros(2 * n, dtype=torch.int32, device=big["dev"]), cls=torch.zeros(2 * n, dtype=torch.uint8, device=big["dev"]),
                   nd=torch.zeros(n, dtype=torch.int32, device=big["dev"]), tk=torch.zeros(n, dtype=torch.int32, device=big["dev"]))
        keep.append((d_b, d_o, out))
        rounds.append([dict(bases=d_b.data_ptr(), offsets=d_o.data_ptr(), R=n, out_taxon=out["taxon"].data_ptr(), out_classified=out["cls"].data_ptr(),
                            out_num_distinct=out["nd"].data_ptr(), out_total_kmers=out["tk"].data_ptr())])
    torch.cuda.synchronize()
    ss = capi.ShardSet([big["ix"]])
    try:
        ss.classify_rounds_device(rounds, thresholds=thr)
        torch.cuda.synchronize()
        for (a, b), (_, _, o) in zip(zip(cuts[:-1], cuts[1:]), keep):
            n = b - a
            for c in range(2):
                assert bool((o["taxon"][c * n:(c + 1) * n] == local["taxon"][c * R + a:c * R + b]).all()), (a, c)
                assert bool((o["cls"][c * n:(c + 1) * n] == local["cls"][c * R + a:c * R + b]).all()), (a, c)
            assert bool((o["nd"] == local["nd"][a:b]).all()) and bool((o["tk"] == local["tk"][a:b]).all())
    finally:
        ss.close() if hasattr(ss, "close") else None


def test_poly_a_known_answer(big):
    torch = big["torch"]
    R = 4096
    bases = torch.full((R * 150 + 64,), ord("A"), dtype=torch.uint8, device=big["dev"])
    out = run(big, bases, big["offsets"][:R + 1], R)
    assert bool((out["nh"] == 1).all()) and bool((out["tk"] == 116).all()) and bool((out["np_"] == 1).all())
    assert bool((out["taxon"] == out["taxon"][0]).all())


def test_sampled_reads_against_the_oracle_through_point_lookups(big, orc):
    """configs[1] pinned to the oracle at full size (Classifier.classify, Classifier.scala:439-454).  >= 20 000 of the 10 M reads --
    a random draw plus reads that hold an N or a run of Ns -- are scanned by the ORACLE; the SEQUENCE-span minimizers it finds are
    looked up in the 1.0e10-record table with slk_index_lookup (table_lookup_kernel: one lane per key, not the cooperative probe
    of the classify kernels); an oracle index is built from exactly those (key, taxon) answers; the oracle classifies the sample
    against it, two thresholds; and the lane kernel's rows for the same reads (taken from its pass over ALL reads) must be
    identical: taxon, classified, distinct hit groups, total k-mers, number of spans.  The table-sharded route at world = 1 (emit ->
    compact -> lookup_coop -> apply) classifies the same sample and must agree as well."""
    torch = big["torch"]
    R = big["R"]
    thr = (0.0, 0.1)
    full = run(big, big["bases"], big["offsets"], R, thresholds=thr)
    rows = big["bases"][:R * 150].view(R, 150)
    rng = np.random.default_rng(20)
    with_n = torch.nonzero((rows[:2_000_000] == ord("N")).any(1)).flatten().cpu().numpy()
    pick = np.unique(np.concatenate([rng.choice(R, 20000, replace=False), with_n[:3000]]))
    assert len(pick) >= 20000 and len(with_n) >= 100
    sample = rows[torch.from_numpy(pick).to(big["dev"])].cpu().numpy()          # [S, 150]
    S = len(pick)
    p = orc.params()
    keys = np.unique(np.concatenate([orc.minimizer_keys(p, bytes(sample[i])) for i in range(S)]))
    taxa = big["ix"].lookup(keys)
    hit = taxa != 0
    assert 0.3 < hit.mean() < 1.0                   # genome-derived reads hit, random ones do not
    oix = orc.Index(1, keys[hit], taxa[hit])
    s_bases = sample.reshape(-1)
    s_off = np.arange(0, (S + 1) * 150, 150, dtype=np.uint64)
    want = orc.classify_batch(p, oix, big["parents"], s_bases, s_off, thresholds=thr)
    idx = torch.from_numpy(pick).to(big["dev"])
    for c in range(2):
        assert np.array_equal(full["taxon"][c * R + idx].cpu().numpy(), want["taxon"][c])
        assert np.array_equal(full["cls"][c * R + idx].cpu().numpy(), want["classified"][c])
    assert np.array_equal(full["nd"][idx].cpu().numpy(), want["num_distinct"])
    assert np.array_equal(full["tk"][idx].cpu().numpy(), want["total_kmers"])
    assert np.array_equal(full["nh"][idx].cpu().numpy(), want["num_hits"])
    assert int(want["classified"][0].sum()) > S // 2 and int((want["num_hits"] > 1).sum()) > S // 2
    # the sharded route, one rank (no exchange): same table, other kernels (lookup_coop_kernel, lane_kernel<EMIT / APPLY>)
    from slacken_amd.sharded import ShardedClassifier
    sc = ShardedClassifier(big["ix"], 0, 1, None, big["dev"])
    d_b = torch.from_numpy(s_bases).to(big["dev"])
    d_o = torch.from_numpy(s_off.astype(np.int64)).to(big["dev"])
    got = sc.classify(d_b, d_o, S, S * 150, thresholds=thr, fast=True)
    assert got["deferred"] == 0
    assert np.array_equal(got["taxon"].cpu().numpy().reshape(2, S), want["taxon"])
    assert np.array_equal(got["classified"].cpu().numpy().reshape(2, S), want["classified"])
    assert np.array_equal(got["num_distinct"].cpu().numpy(), want["num_distinct"])
    assert np.array_equal(got["total_kmers"].cpu().numpy(), want["total_kmers"])
    sc.close()


def _run_bench(*args, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return lines[0]


def test_bench_table_sharded_two_ranks_rehearsal():
    """BASELINE configs[3] end to end: `python bench.py --gpus 2 --table-sharded` starts two ranks, each builds ITS HALF of the
    table (slk_index_set_shard: both scan all genomes and draw the same padding keys, each keeps what falls to it), classifies its
    own batches through EMIT -> all-to-all -> LOOKUP -> all-to-all -> APPLY (slk_shard_step_device), and rank 0 prints ONE line.  Both ranks share GPU 0
    here (gloo, exchange through host memory): a rehearsal of the N-rank path, not a measurement."""
    line = _run_bench("--gpus", "2", "--table-sharded", "--steps", "3", "--warmup", "1", "--records-per-rank", "1e8", "--reads", "1e6",
                      "--genomes", "64", "--genome-len", "262144", "--rehearse-on-one-gpu")
    cfg = line["config"]
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and cfg["parallelism"] == "table-sharded x2"
    assert cfg["records_per_rank"] == 100_000_000 and cfg["reads_per_gpu_per_step"] == 1_000_000
    assert 30 < cfg["keys_per_read"] < 45 and 0.3 < cfg["remote_bytes_per_read"] / cfg["exchanged_bytes_per_read"] < 0.7
    assert cfg["classified_fraction"] > 0.5         # the other rank's half of the genome records answers too
    for stage in ("step", "exchange_keys", "exchange_taxa"):
        assert cfg["stage_ms_in_pipeline"][stage] > 0
    assert abs(line["value"] - 2 * 1.0 / (line["ms_per_step"] * 1e-3)) < 0.01 * line["value"]


def test_bench_table_sharded_one_rank_through_rccl():
    """the `nccl` process group and its device all_to_all_single, on the one GPU there is: a world of one rank that sends its keys
    and taxa to itself through RCCL"""
    line = _run_bench("--table-sharded", "--collectives-at-one-rank", "--steps", "3", "--warmup", "1", "--records-per-rank", "1e8", "--reads", "1e6",
                      "--genomes", "64", "--genome-len", "262144")
    cfg = line["config"]
    assert line["n_gpus"] == 1 and "collectives" in line and cfg["classified_fraction"] > 0.5
    assert cfg["stage_ms_in_pipeline"]["exchange_keys"] > 0.01 and cfg["stage_ms_in_pipeline"]["exchange_taxa"] > 0.005


def test_bench_table_sharded_one_rank():
    line = _run_bench("--table-sharded", "--steps", "3", "--warmup", "1", "--records-per-rank", "2e8", "--reads", "1e6",
                      "--genomes", "64", "--genome-len", "262144")
    cfg = line["config"]
    assert line["n_gpus"] == 1 and cfg["parallelism"] == "table-sharded x1" and cfg["remote_bytes_per_read"] == 0
    assert cfg["xgmi_link_GBps_keys"] is None and line["roofline"]["lookup_stage_frac_of_request_rate_ceiling"] > 0


def test_bench_two_ranks_rehearsal():
    """`python bench.py --gpus 2` end to end on the GPU: the parent starts two ranks itself, each builds its table, classifies its
    own batch, and rank 0 prints ONE line with n_gpus = 2 and the sum of both ranks' reads.  (Both ranks share GPU 0 here --
    an 8-GPU node is the driver's -- so the rendezvous is gloo and the figure is not a measurement.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--records", "2e8",
                        "--reads", "1e6", "--genomes", "64", "--genome-len", "262144", "--rehearse-on-one-gpu"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "cpu_baseline" not in line
    assert line["config"]["reads_per_gpu_per_step"] == 1_000_000
    assert abs(line["value"] - 2 * 1.0 / (line["ms_per_step"] * 1e-3)) < 0.01 * line["value"]     # whole-job reads / max-rank time
    assert line["roofline"]["frac"] > 0.0     # (two processes time-share the GPU here: the event timings mean nothing)

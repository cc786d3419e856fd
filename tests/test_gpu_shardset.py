"""Table-sharded classification in ONE process through the C ABI (slk_shardset_*; BASELINE configs[3]): the
library is spread over several indices by slk_index_set_shard -- every member is fed ALL records and keeps its share -- and a round
classifies one batch per member: scan, keys to their owners, lookup, taxa back, per-read LCA.  On the one-GPU box all members sit
on device 0 and the exchange is device-to-device copies ordered by events (the RCCL leg needs a device per member: compile-checked,
first run when a node is there); results must be bit-identical to the oracle and to the replicated mode, hit lists included."""
import numpy as np
import pytest

import synth
import taxgen

pytestmark = pytest.mark.gpu


def make_members(lib, parents, n, **kw):
    import slacken_amd
    members = []
    for g in range(n):
        ix = slacken_amd.Index(expected_records=max(len(lib.keys) // n * 2, 64), max_taxon=len(parents) - 1, **kw)
        ix.set_shard(g, n)
        ix.set_taxonomy(parents)
        for a in range(0, len(lib.keys), 7001):       # the whole record stream, in chunks, to every member
            ix.append(lib.keys[a:a + 7001], lib.taxa[a:a + 7001])
        ix.finalize()
        members.append(ix)
    return members


def same(got, want, with_hits=True):
    for key in ("taxon", "classified", "num_distinct", "total_kmers"):
        assert np.array_equal(got[key], want[key]), key
    if with_hits:
        assert np.array_equal(got["num_hits"], want["num_hits"])


def same_hits(orc, world, got, reads, mates=None, step=7):
    """un-merged hit lists (TaxonHit per span, ordinal order) against the oracle's, read by read"""
    ho = got["hit_offsets"].astype(np.int64)
    for i in range(0, len(reads), step):
        _, hits = orc.classify_read(world["p"], world["oix"], world["parents"], reads[i].tobytes(),
                                    None if mates is None else mates[i].tobytes(), 2, 0.0)
        g = got["hits"][ho[i]:ho[i + 1]]
        assert [(int(t), int(c)) for t, c in zip(g["taxon"], g["count"])] == hits


@pytest.fixture(scope="module")
def world(orc):
    rng = np.random.default_rng(77)
    parents = taxgen.taxonomy(8 * 32, rng)
    p = orc.params()
    lib = synth.Library(orc, p, parents, n_genomes=10, genome_len=12000, pad_records=30000)
    return dict(p=p, parents=parents, lib=lib, oix=orc.Index(1, lib.keys, lib.taxa), rng=rng)


@pytest.mark.parametrize("n", [2, 3])
def test_rounds_equal_the_oracle_and_the_replicated_mode(orc, world, n):
    import slacken_amd
    from slacken_amd import capi
    lib, parents, p, rng = world["lib"], world["parents"], world["p"], world["rng"]
    members = make_members(lib, parents, n)
    recs = [int(m.info().records) for m in members]
    assert sum(recs) == len(lib.keys) and min(recs) > len(lib.keys) // n // 2        # every record once, spread by the hash
    # each member's table holds exactly its share
    for g, m in enumerate(members):
        k, t = m.export()
        mine = np.array([slacken_amd.lib().slk_shard_of(int(x), n) for x in lib.keys[:2000]]) == g
        assert np.isin(lib.keys[:2000][mine], k).all() and not np.isin(lib.keys[:2000][~mine], k).any()
    sset = capi.ShardSet(members)
    assert sset.exchange_mode == capi.EXCHANGE_COPY          # (all members on device 0)
    whole = slacken_amd.Index(expected_records=len(lib.keys), max_taxon=len(parents) - 1)
    whole.append(lib.keys, lib.taxa)
    whole.set_taxonomy(parents)
    whole.finalize()
    wst = whole.stream()
    thr = (0.0, 0.2)
    for rnd in range(3):
        # uneven batches, one member without any in the second round; long fragments (the staged round) in the third
        sizes = [int(rng.integers(200, 900)) for _ in range(n)]
        if rnd == 1:
            sizes[0] = 0
        batches, wants, reps, all_reads = [], [], [], []
        for g in range(n):
            if sizes[g] == 0:
                batches.append(None); wants.append(None); reps.append(None); all_reads.append(None)
                continue
            reads = synth.make_reads(lib, sizes[g], rng, n_single=0.1, n_run=0.05, vary_length=True)
            if rnd == 2:
                reads += synth.make_reads(lib, 15, rng, length=1800, short=0)
                reads = [reads[i] for i in rng.permutation(len(reads))]
            bases, offsets = synth.pack(reads)
            batches.append((bases, offsets))
            all_reads.append(reads)
            wants.append(orc.classify_batch(p, world["oix"], parents, bases, offsets, thresholds=thr))
            reps.append(wst.classify_batch(bases, offsets, thresholds=thr, with_hits=True))
        for with_hits in (True, False):
            outs = sset.classify(batches, thresholds=thr, with_hits=with_hits)
            for g in range(n):
                if batches[g] is None:
                    assert outs[g] is None
                    continue
                same(outs[g], wants[g], with_hits)
                for key in ("taxon", "classified", "num_distinct", "total_kmers"):
                    assert np.array_equal(outs[g][key], reps[g][key])
                if with_hits:
                    assert np.array_equal(outs[g]["hits"], reps[g]["hits"]) and np.array_equal(outs[g]["hit_offsets"], reps[g]["hit_offsets"])
                    same_hits(orc, world, outs[g], all_reads[g])
    sset.close()


def test_paired_rounds_and_many_taxa(orc, world):
    """pairs (the mate border's pseudo-span in the hit lists) and fragments with more than 12 distinct taxa (handed back by the
    lane kernel: staged round) through a set of two"""
    from slacken_amd import capi
    lib, parents, p, rng = world["lib"], world["parents"], world["p"], world["rng"]
    members = make_members(lib, parents, 2)
    sset = capi.ShardSet(members, exchange=capi.EXCHANGE_COPY)
    thr = (0.0, 0.15)
    batches, wants, pairs = [], [], []
    for g in range(2):
        r1 = synth.make_reads(lib, 300, rng, n_single=0.1, n_run=0.05, vary_length=True)
        r2 = synth.make_reads(lib, 300, rng, n_single=0.1, n_run=0.05, vary_length=True)
        b1, o1 = synth.pack(r1)
        b2, o2 = synth.pack(r2)
        batches.append((b1, o1, b2, o2))
        pairs.append((r1, r2))
        wants.append(orc.classify_batch(p, world["oix"], parents, b1, o1, b2, o2, thresholds=thr))
    outs = sset.classify(batches, thresholds=thr)
    for g in range(2):
        same(outs[g], wants[g])
        same_hits(orc, world, outs[g], pairs[g][0], pairs[g][1], step=5)
    sset.close()
    # every minimizer of every read its own record with its own taxon: > 12 distinct taxa per read
    taxa = np.array(taxgen.defined_taxa(parents))
    reads = [synth.random_dna(150, rng) for _ in range(120)] + [synth.random_dna(500, rng) for _ in range(10)]
    keys = np.unique(np.concatenate([orc.minimizer_keys(p, r.tobytes()) for r in reads]))
    tx = rng.choice(taxa, size=len(keys)).astype(np.int32)

    class L2:
        pass
    l2 = L2()
    l2.keys, l2.taxa = keys, tx
    members = make_members(l2, parents, 2)
    sset = capi.ShardSet(members)
    bases, offsets = synth.pack(reads)
    half = len(reads) // 2
    b0, o0 = synth.pack(reads[:half])
    b1, o1 = synth.pack(reads[half:])
    oix = orc.Index(1, keys, tx)
    outs = sset.classify([(b0, o0), (b1, o1)], thresholds=(0.0, 0.3))
    same(outs[0], orc.classify_batch(p, oix, parents, b0, o0, thresholds=(0.0, 0.3)))
    same(outs[1], orc.classify_batch(p, oix, parents, b1, o1, thresholds=(0.0, 0.3)))
    assert (outs[0]["num_distinct"] > 12).all()
    sset.close()


def test_set_refuses_what_is_not_a_sharding(orc, world):
    import slacken_amd
    from slacken_amd import capi
    lib, parents = world["lib"], world["parents"]
    members = make_members(lib, parents, 2)
    with pytest.raises(slacken_amd.SlackenError):
        capi.ShardSet(members[::-1])                       # shard 1 in place 0
    with pytest.raises(slacken_amd.SlackenError):
        capi.ShardSet(members[:1])                         # shard 0 of 2 alone
    with pytest.raises(slacken_amd.SlackenError):
        capi.ShardSet(members, exchange=capi.EXCHANGE_RCCL)   # two members on one device: RCCL cannot
    ix = slacken_amd.Index(expected_records=100, max_taxon=len(parents) - 1)
    ix.append(lib.keys[:10], lib.taxa[:10])
    with pytest.raises(slacken_amd.SlackenError):
        ix.set_shard(0, 2)                                 # after the first record


RCCL_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import torch  # (as in the test session: PyTorch's HIP runtime first)
import slacken_amd, synth, taxgen
from slacken_amd import capi
rng = np.random.default_rng(5)
parents = taxgen.taxonomy(8 * 16, rng)
taxa = np.array(taxgen.defined_taxa(parents))
reads = [synth.random_dna(int(rng.integers(60, 300)), rng) for _ in range(800)] + [synth.random_dna(1500, rng) for _ in range(5)]
keys = np.unique((rng.integers(0, 2**62, 20000, dtype=np.int64) * 4) & ~np.int64(0x33333330))
tx = rng.choice(taxa, size=len(keys)).astype(np.int32)
ix = slacken_amd.Index(expected_records=len(keys), max_taxon=len(parents) - 1)
ix.set_shard(0, 1)
ix.append(keys, tx); ix.set_taxonomy(parents); ix.finalize()
# the reads' own minimizers cannot be had without the oracle here: plant records by classifying spans instead
st = ix.stream()
bases, offsets = synth.pack(reads)
so, spans = st.spans_batch(bases, offsets)
rk = np.unique(spans["key"][spans["flag"] == 1])
ix2 = slacken_amd.Index(expected_records=len(rk) + len(keys), max_taxon=len(parents) - 1)
ix2.set_shard(0, 1)
ix2.append(rk, rng.choice(taxa, size=len(rk)).astype(np.int32)); ix2.set_taxonomy(parents); ix2.finalize()
want = ix2.stream().classify_batch(bases, offsets, thresholds=(0.0, 0.2))
sset = capi.ShardSet([ix2], exchange=capi.EXCHANGE_RCCL)
assert sset.exchange_mode == capi.EXCHANGE_RCCL
for _ in range(2):
    got = sset.classify([(bases, offsets)], thresholds=(0.0, 0.2))[0]
    for k in ("taxon", "classified", "num_distinct", "total_kmers", "hit_offsets", "hits"):
        assert np.array_equal(got[k], want[k]), k
assert int(want["classified"][0].sum()) > 700
sset.close()
print("RCCL-OK")
"""


def test_rccl_leg_runs_with_one_member(tmp_path):
    """The RCCL leg of the exchange -- librccl loaded at run time, ncclCommInitAll, grouped ncclSend / ncclRecv on the member's
    stream -- cannot span devices on a one-GPU box, but it can RUN: a set of one member sends its keys to itself through RCCL.
    Both rounds (fast, and staged for the long fragments) must give what the plain classify call gives.  In a child process with a
    time limit: a collective library that does not come up must not take the test session with it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_one.py"
    script.write_text(RCCL_SCRIPT.format(root=root))
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=240)
    assert p.returncode == 0 and "RCCL-OK" in p.stdout, p.stderr[-3000:]


def test_splitter_outside_the_lane_kernels_range_takes_the_staged_round(orc):
    """A window of 40 m-mers (k = 70, m = 31): neither the lane kernel nor the wave kernel take it -- every fragment of every member
    goes through the staged round (scan kernel, keys collected by owner, exchange, unbounded classify kernel)."""
    import slacken_amd
    from slacken_amd import capi
    rng = np.random.default_rng(21)
    parents = taxgen.taxonomy(8 * 16, rng)
    p = orc.params(k=70, m=31, spaces=5)
    lib = synth.Library(orc, p, parents, n_genomes=6, genome_len=6000, pad_records=5000)
    members = []
    for g in range(2):
        ix = slacken_amd.Index(k=70, m=31, spaces=5, expected_records=len(lib.keys), max_taxon=len(parents) - 1)
        ix.set_shard(g, 2)
        ix.append(lib.keys, lib.taxa)
        ix.set_taxonomy(parents)
        ix.finalize()
        members.append(ix)
    sset = capi.ShardSet(members)
    oix = orc.Index(1, lib.keys, lib.taxa)
    batches, wants, all_reads = [], [], []
    for g in range(2):
        reads = synth.make_reads(lib, 300, rng, length=200, vary_length=True, n_single=0.1, n_run=0.05)
        bases, offsets = synth.pack(reads)
        batches.append((bases, offsets))
        all_reads.append(reads)
        wants.append(orc.classify_batch(p, oix, parents, bases, offsets, thresholds=(0.0, 0.1)))
    outs = sset.classify(batches, thresholds=(0.0, 0.1))
    world = dict(p=p, oix=oix, parents=parents)
    for g in range(2):
        same(outs[g], wants[g])
        same_hits(orc, world, outs[g], all_reads[g], step=9)
    assert wants[0]["classified"][0].mean() > 0.3
    sset.close()

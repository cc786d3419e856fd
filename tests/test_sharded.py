"""Table-sharded mode (configs[3]): ownership function on CPU; on the GPU box two gloo ranks share the card, each holds
half of the records, classifies half of the reads through the key/taxon all-to-all, and the union must equal the oracle."""
import os
import socket

import numpy as np
import pytest
import torch

import slacken_amd
from slacken_amd import sharded


def test_shard_function_agrees_everywhere():
    L = slacken_amd.lib()
    rng = np.random.default_rng(3)
    keys = rng.integers(-2**63, 2**63 - 1, 5000, dtype=np.int64)
    for world in (1, 2, 3, 7, 8):
        want = np.array([L.slk_shard_of(int(k), world) for k in keys])
        assert np.array_equal(sharded.shard_of_numpy(keys, world), want)
        assert np.array_equal(sharded.shard_of_torch(torch.from_numpy(keys), world).numpy(), want)
        assert len(np.unique(want)) == world or world == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, payload, outdir, fast):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keys, taxa, parents, bases, offsets = payload
    mine = sharded.shard_of_numpy(keys, world) == rank
    ix = slacken_amd.Index(expected_records=max(int(mine.sum()), 16), max_taxon=len(parents) - 1)
    ix.append(keys[mine], taxa[mine])
    ix.set_taxonomy(parents)
    ix.finalize()
    R = len(offsets) - 1
    lo, hi = (rank * R) // world, ((rank + 1) * R) // world
    dev = torch.device("cuda", 0)
    b0, b1 = int(offsets[lo]), int(offsets[hi])
    d_bases = torch.from_numpy(bases[b0:b1]).to(dev)
    d_off = torch.from_numpy((offsets[lo:hi + 1] - offsets[lo]).astype(np.int64)).to(dev)
    sc = sharded.ShardedClassifier(ix, rank, world, dist, dev, exchange_on_cpu=True)
    out = sc.classify(d_bases, d_off, hi - lo, b1 - b0, thresholds=(0.0, 0.2), fast=fast)
    assert ("deferred" in out) == fast
    np.savez(os.path.join(outdir, f"r{rank}.npz"), lo=lo, hi=hi, **{k: v.cpu().numpy() for k, v in out.items()
                                                                     if hasattr(v, "cpu")})
    del out
    sc.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("fast", [True, False], ids=["fast", "staged"])
def test_two_ranks_half_table_each(orc, tmp_path, fast):
    import torch.multiprocessing as mp
    import synth
    import taxgen
    rng = np.random.default_rng(41)
    parents = taxgen.taxonomy(8 * 32, rng)
    p = orc.params()
    lib = synth.Library(orc, p, parents, n_genomes=8, genome_len=10000, pad_records=20000)
    reads = synth.make_reads(lib, 3000, rng, n_single=0.1, n_run=0.05, vary_length=True)
    reads += synth.make_reads(lib, 40, rng, length=1500, short=0)     # longer than the fused kernel takes: deferred
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    bases, offsets = synth.pack(reads)
    want = orc.classify_batch(p, orc.Index(1, lib.keys, lib.taxa), parents, bases, offsets, thresholds=(0.0, 0.2))
    mp.spawn(_rank, args=(2, _free_port(), (lib.keys, lib.taxa, parents, bases, offsets), str(tmp_path), fast), nprocs=2, join=True)
    R = len(reads)
    got = {k: np.zeros((2, R), np.int64) if k in ("taxon", "classified") else np.zeros(R, np.int64)
           for k in ("taxon", "classified", "num_distinct", "total_kmers", "num_hits")}
    for rank in range(2):
        z = np.load(os.path.join(str(tmp_path), f"r{rank}.npz"))
        lo, hi = int(z["lo"]), int(z["hi"])
        n = hi - lo
        for c in range(2):
            got["taxon"][c, lo:hi] = z["taxon"][c * n:(c + 1) * n]
            got["classified"][c, lo:hi] = z["classified"][c * n:(c + 1) * n]
        for k in ("num_distinct", "total_kmers", "num_hits"):
            got[k][lo:hi] = z[k][:n]
    for k in got:
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("big_ids", [False, True], ids=["ids<2^22", "dense-ids"])
def test_single_rank_fast_route_with_deferrals(orc, big_ids):
    """world = 1: no exchange, but the same EMIT -> LOOKUP -> APPLY jobs of the step kernel, incl. fragments the fused kernel
    hands back (longer than 1000 bases; more than 12 distinct taxa) and empty / vanishing fragments.  With taxon ids beyond
    22 bits the table holds dense internal ids: the owners' answers (caller's ids) are translated by the apply kernel."""
    import synth
    import taxgen
    rng = np.random.default_rng(43)
    parents = taxgen.taxonomy(8 * 64, rng)
    if big_ids:
        parents, _ = taxgen.sparse_relabel(parents, 5_000_000, rng)
    taxa = np.array(taxgen.defined_taxa(parents))
    p = orc.params()
    reads = synth.make_reads(synth.Library(orc, p, parents, n_genomes=4, genome_len=8000), 500, rng, vary_length=True)
    reads += [synth.random_dna(400, rng) for _ in range(30)] + [synth.random_dna(2500, rng) for _ in range(5)]
    reads += [np.zeros(0, np.uint8), np.frombuffer(b"ACGT" * 5, np.uint8)]
    keys = np.unique(np.concatenate([orc.minimizer_keys(p, r.tobytes()) for r in reads]))
    tx = rng.choice(taxa, size=len(keys)).astype(np.int32)       # every minimizer its own random taxon: many taxa per read
    ix = slacken_amd.Index(expected_records=len(keys), max_taxon=len(parents) - 1)
    ix.append(keys, tx)
    ix.set_taxonomy(parents)
    ix.finalize()
    assert (ix.info().dense_taxa > 0) == big_ids
    bases, offsets = synth.pack(reads)
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(bases).to(dev)
    d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    sc = sharded.ShardedClassifier(ix, 0, 1, None, dev)
    R = len(reads)
    want = orc.classify_batch(p, orc.Index(1, keys, tx), parents, bases, offsets, thresholds=(0.0, 0.1))
    for fast in (True, False):
        out = sc.classify(d_bases, d_off, R, int(offsets[-1]), thresholds=(0.0, 0.1), fast=fast)
        assert np.array_equal(out["taxon"].cpu().numpy().reshape(2, R), want["taxon"]), fast
        assert np.array_equal(out["classified"].cpu().numpy().reshape(2, R), want["classified"]), fast
        for k in ("num_distinct", "total_kmers", "num_hits"):
            assert np.array_equal(out[k].cpu().numpy()[:R], want[k]), (k, fast)
        if fast:
            assert out["deferred"] >= 35
    sc.close()


@pytest.mark.gpu
def test_single_rank_paired(orc):
    """paired fragments through both sharded routes: the slot of a probe is offsets[r] + mate_offsets[r] + r + ordinal"""
    import synth
    import taxgen
    rng = np.random.default_rng(44)
    parents = taxgen.taxonomy(8 * 32, rng)
    p = orc.params()
    lib = synth.Library(orc, p, parents, n_genomes=6, genome_len=9000, pad_records=5000)
    r1 = synth.make_reads(lib, 1200, rng, n_single=0.1)
    r2 = synth.make_reads(lib, 1200, rng, vary_length=True, short=0.1)
    r1[7], r2[7] = synth.make_reads(lib, 1, rng, length=900, short=0)[0], synth.make_reads(lib, 1, rng, length=700, short=0)[0]  # > 1000 together
    ix = slacken_amd.Index(expected_records=len(lib.keys), max_taxon=len(parents) - 1)
    ix.append(lib.keys, lib.taxa)
    ix.set_taxonomy(parents)
    ix.finalize()
    b1, o1 = synth.pack(r1)
    b2, o2 = synth.pack(r2)
    dev = torch.device("cuda", 0)
    # (device buffers are exactly offsets[R] bytes: the kernels read nothing past them)
    d = [torch.from_numpy(b1).to(dev), torch.from_numpy(o1.astype(np.int64)).to(dev),
         torch.from_numpy(b2).to(dev), torch.from_numpy(o2.astype(np.int64)).to(dev)]
    R = len(r1)
    want = orc.classify_batch(p, orc.Index(1, lib.keys, lib.taxa), parents, b1, o1, b2, o2, thresholds=(0.0, 0.15))
    sc = sharded.ShardedClassifier(ix, 0, 1, None, dev)
    for fast in (True, False):
        out = sc.classify(d[0], d[1], R, int(o1[-1]), thresholds=(0.0, 0.15), fast=fast, d_mate_bases=d[2], d_mate_offsets=d[3],
                          total_mate_bases=int(o2[-1]))
        assert np.array_equal(out["taxon"].cpu().numpy().reshape(2, R), want["taxon"]), fast
        assert np.array_equal(out["classified"].cpu().numpy().reshape(2, R), want["classified"]), fast
        for k in ("num_distinct", "total_kmers", "num_hits"):
            assert np.array_equal(out[k].cpu().numpy()[:R], want[k]), (k, fast)
        if fast:
            assert out["deferred"] >= 1
    sc.close()


def _rank_many(rank, world, port, payload, outdir):
    """several batches through classify_many: the step kernel carries the scan of batch t, the lookups of batch t - 2 and the replay
    of batch t - 4; twice over (the second pass runs on the buffers torch has cached from the first)"""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keys, taxa, parents, batches = payload
    ix = slacken_amd.Index(expected_records=len(keys), max_taxon=len(parents) - 1)
    ix.set_shard(rank, world)               # (fed everything, keeps its share)
    ix.append(keys, taxa)
    ix.set_taxonomy(parents)
    ix.finalize()
    dev = torch.device("cuda", 0)
    mine = [b for i, b in enumerate(batches) if i % world == rank]
    dbs = [(torch.from_numpy(b).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev), len(o) - 1, int(o[-1]), None) for b, o in mine]
    sc = sharded.ShardedClassifier(ix, rank, world, dist, dev, exchange_on_cpu=True)
    res = {}
    for tag in ("f", "s"):
        outs = sc.classify_many(dbs, thresholds=(0.0, 0.2))
        for j, o in enumerate(outs):
            for k in ("taxon", "classified", "num_distinct", "total_kmers", "num_hits"):
                res[f"{tag}_{j}_{k}"] = o[k].cpu().numpy()
    np.savez(os.path.join(outdir, f"many{rank}.npz"), **res)
    del outs
    sc.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_many_batches_with_the_lookups_inside_later_scans(orc, tmp_path, world):
    """classify_many over five batches per rank (uneven sizes, one of a single fragment, one with fragments the lane kernel hands
    back): the keys a rank receives for batch t are answered inside the scan of batch t + 2 and batch t is replayed and classified
    inside the scan of batch t + 4 -- the first steps' side jobs find nothing to do, the last batches' lookups and replays run in
    steps without a scan, a scan with fewer tiles than the batch it carries leaves the rest to tiles without fragments.  Every batch
    against the oracle."""
    import torch.multiprocessing as mp
    import synth
    import taxgen
    rng = np.random.default_rng(57)
    parents = taxgen.taxonomy(8 * 32, rng)
    p = orc.params()
    lib = synth.Library(orc, p, parents, n_genomes=8, genome_len=10000, pad_records=20000)
    sizes = [1500, 1, 2200, 90, 1700, 2600, 300, 1200, 64, 2000][:5 * world]
    batches, wants = [], []
    for i, n in enumerate(sizes):
        reads = synth.make_reads(lib, n, rng, n_single=0.1, n_run=0.05, vary_length=True)
        if i == 2:
            reads += synth.make_reads(lib, 12, rng, length=1600, short=0)
        bases, offsets = synth.pack(reads)
        batches.append((bases, offsets))
        wants.append(orc.classify_batch(p, orc.Index(1, lib.keys, lib.taxa), parents, bases, offsets, thresholds=(0.0, 0.2)))
    if world == 1:
        _rank_many(0, 1, _free_port(), (lib.keys, lib.taxa, parents, batches), str(tmp_path))
    else:
        mp.spawn(_rank_many, args=(world, _free_port(), (lib.keys, lib.taxa, parents, batches), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), f"many{rank}.npz"))
        mine = [i for i in range(len(batches)) if i % world == rank]
        for j, i in enumerate(mine):
            R = len(batches[i][1]) - 1
            for tag in ("f", "s"):
                assert np.array_equal(z[f"{tag}_{j}_taxon"].reshape(2, -1)[:, :R], wants[i]["taxon"]), (tag, i)
                assert np.array_equal(z[f"{tag}_{j}_classified"].reshape(2, -1)[:, :R], wants[i]["classified"]), (tag, i)
                for k in ("num_distinct", "total_kmers", "num_hits"):
                    assert np.array_equal(z[f"{tag}_{j}_{k}"][:R], wants[i][k]), (tag, i, k)


@pytest.mark.gpu
def test_results_held_by_a_reference_cycle_outlive_the_classifier(tmp_path):
    """Round 3's segfault: tensors of the sharded pipeline, allocated on torch ExternalStreams that wrap the ENGINE's streams, were
    kept by a reference cycle until the cycle collector ran -- after the streams had been destroyed.  Now the results are
    allocated on the caller's stream, nothing of a batch outlives classify_many, close() drains, empties torch's cache and drops
    the stream views BEFORE it destroys the streams, and an engine stream is never destroyed while a torch view of it is alive
    (capi.Stream.close parks it).  Here: a cycle holds a batch's outputs AND a stream view; the classifier is closed; the
    collector runs; the process must exit cleanly (run as a child: a crash at interpreter exit would not show in this one)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import gc, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
import numpy as np, torch
import slacken_amd
from slacken_amd import sharded, capi
from oracle import oracle as orc
import synth, taxgen
rng = np.random.default_rng(5)
parents = taxgen.taxonomy(8 * 16, rng)
p = orc.params()
lib = synth.Library(orc, p, parents, n_genomes=4, genome_len=6000, pad_records=2000)
ix = slacken_amd.Index(expected_records=len(lib.keys), max_taxon=len(parents) - 1)
ix.append(lib.keys, lib.taxa); ix.set_taxonomy(parents); ix.finalize()
reads = synth.make_reads(lib, 700, rng)
bases, offsets = synth.pack(reads)
dev = torch.device('cuda', 0)
d_b, d_o = torch.from_numpy(bases).to(dev), torch.from_numpy(offsets.astype(np.int64)).to(dev)
sc = sharded.ShardedClassifier(ix, 0, 1, None, dev)
outs = sc.classify_many([(d_b, d_o, len(reads), int(offsets[-1]), None)] * 6)
want = orc.classify_batch(p, orc.Index(1, lib.keys, lib.taxa), parents, bases, offsets)
assert np.array_equal(outs[3]['taxon'].cpu().numpy()[:len(reads)], want['taxon'][0])
view = sc.st.external_stream(torch, dev)          # a torch view of the engine's stream that the caller holds on to
cycle = {{'outs': outs, 'view': view}}
cycle['self'] = cycle                              # only the cycle collector frees these
del outs, view
sc.close()                                         # the stream with the live view is parked, not destroyed
assert len(capi._deferred_streams) == 1
del cycle
gc.collect()
torch.cuda.synchronize()
assert capi.release_deferred_streams() == 0        # the view has died: now the stream goes
ix.close()
print('clean exit')
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "clean exit" in r.stdout, (r.returncode, r.stderr[-2000:])

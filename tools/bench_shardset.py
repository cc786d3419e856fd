#!/usr/bin/env python3
"""bench_shardset.py -- device rate of the table-sharded mode behind the C ABI (slk_shardset_classify_rounds, device-resident
batches): the set a JVM executor that owns its host's GPUs would drive.  One process; member g = shard g of n on device
devices[g] (the same device may be listed several times: how a one-GPU box runs it -- the members then share the card's
memory system, so n members on one card measure the protocol, not n cards).

  python tools/bench_shardset.py --devices 0 --records-per-member 5e9 --reads 1e7 --rounds 20

Prints one JSON line: reads/s over all members, ms per round, the fraction of the 8 TB/s HBM roofline by SURVEY 8d's bytes."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402  (the workload generators of bench.py)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="0", help="comma-separated device of every member (repeat a device for several members on it)")
    ap.add_argument("--records-per-member", type=float, default=5.0e9)
    ap.add_argument("--reads", type=float, default=1.0e7, help="150 bp reads per member per round")
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--genomes", type=int, default=8192)
    ap.add_argument("--genome-len", type=int, default=1 << 20)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import slacken_amd
    from slacken_amd import capi, sharded
    devs = [int(x) for x in args.devices.split(",")]
    W = len(devs)
    n_reads, per = int(args.reads), int(args.records_per_member)
    t0 = time.time()
    parents, taxa, leaves = bench.build_taxonomy()
    rng = np.random.default_rng(224)
    G = min(args.genomes, len(leaves))
    genome_taxa = rng.choice(leaves, size=G, replace=False).astype(np.int32)
    g_offsets = np.arange(0, (G + 1) * args.genome_len, args.genome_len, dtype=np.uint64)
    smask = ((2**62 - 1) & ~0x0CCCCCCC) << 2
    smask_i64 = smask - (1 << 64) if smask >= (1 << 63) else smask
    members, batches, keep = [], [], []
    for g, d in enumerate(devs):
        torch.cuda.set_device(d)
        device = torch.device("cuda", d)
        genome_cat = bench.make_genomes_device(torch, G, args.genome_len, 224 + 1, device)
        ix = slacken_amd.Index(k=bench.K, m=bench.M, spaces=bench.SPACES, expected_records=max(per, int(G * args.genome_len * 0.4 / W) + 4096),
                               max_taxon=bench.TAX_EXTENT - 1, device=d)
        ix.set_shard(g, W)
        ix.set_taxonomy(parents)
        ix.add_sequences_device(genome_cat.data_ptr(), g_offsets, genome_taxa)
        d_taxa = torch.from_numpy(taxa).to(device)
        gen = torch.Generator(device=device)
        gen.manual_seed(224 + 7)
        CH = 1 << 27
        while int(ix.info().records) < per:
            hi = torch.randint(0, 2**32, (CH,), generator=gen, device=device, dtype=torch.int64)
            lo = torch.randint(0, 2**32, (CH,), generator=gen, device=device, dtype=torch.int64)
            keys = ((hi << 32) | lo) & smask_i64
            del hi, lo
            tx = d_taxa[torch.randint(0, len(taxa), (CH,), generator=gen, device=device)]
            if W > 1:
                k = sharded.shard_of_torch(keys, W) == g
                keys, tx = keys[k].contiguous(), tx[k].contiguous()
            n = min(int(keys.numel()), per - int(ix.info().records))
            torch.cuda.synchronize()
            ix.append_device(keys.data_ptr(), tx.data_ptr(), n)
            del keys, tx
        ix.finalize()
        info = ix.info()
        print(f"[shardset] member {g} on device {d}: {info.records} records in {info.table_bytes / 2**30:.1f} GiB ({time.time() - t0:.1f}s)",
              file=sys.stderr, flush=True)
        d_bases, d_offsets = bench.make_reads_device(torch, genome_cat, args.genome_len, G, n_reads, 150 + g, device)
        del genome_cat
        torch.cuda.empty_cache()
        out = dict(taxon=torch.zeros(n_reads, dtype=torch.int32, device=device), cls=torch.zeros(n_reads, dtype=torch.uint8, device=device),
                   nd=torch.zeros(n_reads, dtype=torch.int32, device=device), tk=torch.zeros(n_reads, dtype=torch.int32, device=device))
        keep.append((d_bases, d_offsets, out))
        members.append(ix)
        batches.append(dict(bases=d_bases.data_ptr(), offsets=d_offsets.data_ptr(), R=n_reads, out_taxon=out["taxon"].data_ptr(),
                            out_classified=out["cls"].data_ptr(), out_num_distinct=out["nd"].data_ptr(), out_total_kmers=out["tk"].data_ptr()))
    torch.cuda.synchronize()
    ss = capi.ShardSet(members)
    ss.classify_rounds_device([batches] * args.warmup)
    for d in set(devs):
        torch.cuda.synchronize(d)
    t = time.perf_counter()
    ss.classify_rounds_device([batches] * args.rounds)
    for d in set(devs):
        torch.cuda.synchronize(d)
    el = time.perf_counter() - t
    classified = float(keep[0][2]["cls"].float().mean().item())
    probes_per_read = 38.67
    ms = el / args.rounds * 1e3
    bytes_per_round_member = n_reads * (150 + 64 * probes_per_read + 8)
    line = {"metric": "classify_throughput_150bp_table_sharded_shardset", "value": round(W * n_reads / (el / args.rounds) / 1e6, 3), "unit": "M reads/s",
            "members": W, "devices": devs, "rounds": args.rounds, "warmup": args.warmup, "ms_per_round": round(ms, 3),
            "reads_per_member_per_round": n_reads, "records_per_member": int(members[0].info().records),
            "table_GiB_per_member": round(members[0].info().table_bytes / 2**30, 1), "classified_fraction": round(classified, 4),
            "exchange": "RCCL" if ss.exchange_mode == capi.EXCHANGE_RCCL else "device-to-device copies",
            "entry": "slk_shardset_classify_rounds(device_resident = 1): rounds pipelined, one kernel per member and step",
            "roofline": {"bound": "hbm", "achieved": round(bytes_per_round_member / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(bytes_per_round_member / (ms * 1e-3) / 1e9 / 8000.0, 4),
                         "note": "per member; members that share a device share its memory system"}}
    print(json.dumps(line), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            f.write(json.dumps(line) + "\n")
    ss.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Where the table-sharded step kernel's time goes: jobs_alone on a 5.0e9-record shard with parts of the kernel switched off
(a -DSLK_TUNING build: SLACKEN_AMD_LIB=build_ab/lib_tuning.so, SLK_DEBUG_ABLATE = 4: every lookup hits the L2; 32: keys and
metadata are not written; 128: the answers are not written).  The results of such a run are wrong by construction: timing only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json, os
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np, torch
import bench, slacken_amd
from slacken_amd import sharded
dev = torch.device("cuda", 0)
parents, taxa, leaves = bench.build_taxonomy()
rng = np.random.default_rng(224)
G, GL = 8192, 1 << 20
genome_taxa = rng.choice(leaves, size=G, replace=False).astype(np.int32)
genome_cat = bench.make_genomes_device(torch, G, GL, 225, dev)
ix = slacken_amd.Index(k=35, m=31, spaces=7, expected_records=int(5e9), max_taxon=bench.TAX_EXTENT - 1)
ix.set_taxonomy(parents)
ix.add_sequences_device(genome_cat.data_ptr(), np.arange(0, (G + 1) * GL, GL, dtype=np.uint64), genome_taxa)
smask = ((2**62 - 1) & ~0x0CCCCCCC) << 2
smask = smask - (1 << 64) if smask >= (1 << 63) else smask
d_taxa = torch.from_numpy(taxa).to(dev)
gen = torch.Generator(device=dev); gen.manual_seed(231)
while int(ix.info().records) < int(5e9):
    CH = 1 << 27
    keys = ((torch.randint(0, 2**32, (CH,), generator=gen, device=dev, dtype=torch.int64) << 32) | torch.randint(0, 2**32, (CH,), generator=gen, device=dev, dtype=torch.int64)) & smask
    tx = d_taxa[torch.randint(0, len(taxa), (CH,), generator=gen, device=dev)]
    torch.cuda.synchronize()
    ix.append_device(keys.data_ptr(), tx.data_ptr(), min(CH, int(5e9) - int(ix.info().records)))
ix.finalize()
d_b, d_o = bench.make_reads_device(torch, genome_cat, GL, G, 10_000_000, 150, dev)
del genome_cat
sc = sharded.ShardedClassifier(ix, 0, 1, None, dev)
batch = (d_b, d_o, 10_000_000, 1_500_000_000, None)
sc.jobs_alone(batch)
print(json.dumps(sc.jobs_alone(batch)))
sc.close()
''' % (ROOT, ROOT)

out = {}
SETS = (("all on", "0"), ("lookups from the L2 (4)", "4"), ("no key / meta stores (32)", "32"), ("no answer stores (128)", "128"),
        ("no list stores at all (160)", "160"), ("L2 lookups, no stores (164)", "164"), ("no EMIT at all: the scan drops its keys (256)", "256"))
if len(sys.argv) > 1:   # only the experiments named (by their number)
    SETS = tuple(x for x in SETS if x[1] in sys.argv[1:])
for name, abl in SETS:
    env = dict(os.environ, SLACKEN_AMD_LIB=os.path.join(ROOT, "build_ab", "lib_tuning.so"), SLK_DEBUG_ABLATE=abl)
    r = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, env=env, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    out[name] = json.loads(line[-1]) if line else r.stderr[-300:]
    print(f"{name:32s} {out[name]}", flush=True)

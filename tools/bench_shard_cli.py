#!/usr/bin/env python3
"""`slacken-amd classify` end to end with the library replicated (one table; two tables on the GPU) and SPREAD over two device tables
(--shard-table: rounds through slk_shardset_classify, exchange by device-to-device copies on the one-GPU box): wall clock, and that the
files are the same.  GPU box; one JSON object."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import slacken_amd
    import taxgen
    import parquet_to_slkrec as conv
    from test_host_classify2_gpu import write_ranked_taxonomy
    R = int(float(os.environ.get("R", 5e6)))
    rng = np.random.default_rng(3)
    parents = taxgen.taxonomy(8 * 64, rng)
    taxa = np.array(taxgen.defined_taxa(parents))
    G, L = 128, 1 << 20
    acgt = np.frombuffer(b"ACGT", np.uint8)
    bases = acgt[rng.integers(0, 4, G * L, dtype=np.uint8)]
    ix = slacken_amd.Index(expected_records=G * L // 2, max_taxon=len(parents) - 1)
    ix.set_taxonomy(parents)
    ix.add_sequences(bases, np.arange(G + 1, dtype=np.uint64) * np.uint64(L), rng.choice(taxa[len(taxa) // 2:], G).astype(np.int32))
    keys, tx = ix.export()
    ix.close()
    d = tempfile.mkdtemp(prefix="slkshard_")
    loc = os.path.join(d, "lib")
    conv.write_slkrec(loc + ".slkrec", keys, tx)
    with open(loc + ".properties", "w") as f:
        f.write("k=35\nm=31\nversion=1\nsplitter=randomXOR\nminimizerSpaces=7\n")
    write_ranked_taxonomy(loc + "_taxonomy", parents)
    starts = rng.integers(0, G * L - 150, R)
    fq = os.path.join(d, "reads.fq")
    with open(fq, "wb") as f:
        qual = b"I" * 150
        for s in range(0, R, 100000):
            e = min(R, s + 100000)
            blk = bases[(starts[s:e, None] + np.arange(150)[None, :])]
            f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (s + i, blk[i].tobytes(), qual) for i in range(e - s)))
    out = dict(reads=R, records=int(len(keys)))
    digests = {}
    for name, extra in (("one_table", []), ("two_tables_replicated", ["--devices", "0,0"]), ("two_tables_sharded", ["--devices", "0,0", "--shard-table"]),
                        ("four_tables_sharded", ["--devices", "0,0,0,0", "--shard-table"])):
        for detailed in (True, False):
            o = os.path.join(d, name + ("_d" if detailed else "_r"))
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                r = subprocess.run([os.path.join(ROOT, "slacken_amd", "bin", "slacken-amd"), "classify", "-i", loc, "-o", o, "-c", "0.15", *extra,
                                    *([] if detailed else ["--nodetailed"]), fq], capture_output=True, text=True)
                assert r.returncode == 0, r.stderr
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out[name + ("_per_read_lines" if detailed else "_reports_only")] = dict(wall_s=round(best, 2), M_reads_per_s=round(R / best / 1e6, 2))
            digests[(name, detailed)] = hashlib.sha256(open(o + "_c0.15/all_kreport.txt", "rb").read()).hexdigest()
    out["reports_identical"] = len(set(digests.values())) == 1
    print(json.dumps(out))


if __name__ == "__main__":
    main()

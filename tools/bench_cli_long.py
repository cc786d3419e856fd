#!/usr/bin/env python3
"""`slacken-amd classify` end to end on long reads (10 kbp, one in a hundred with a run of Ns): per-read lines (hit lists from the
wave kernel) and reports only (segment kernel).  Run on the GPU box; prints one JSON object."""
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
    R, LEN = int(os.environ.get("R", 100_000)), int(os.environ.get("LEN", 10_000))
    rng = np.random.default_rng(3)
    parents = taxgen.taxonomy(8 * 64, rng)
    taxa = np.array(taxgen.defined_taxa(parents))
    G, L = 64, 1 << 20
    acgt = np.frombuffer(b"ACGT", np.uint8)
    bases = acgt[rng.integers(0, 4, G * L, dtype=np.uint8)]
    ix = slacken_amd.Index(expected_records=G * L // 2, max_taxon=len(parents) - 1)
    ix.set_taxonomy(parents)
    ix.add_sequences(bases, np.arange(G + 1, dtype=np.uint64) * np.uint64(L), rng.choice(taxa[len(taxa) // 2:], G).astype(np.int32))
    keys, tx = ix.export()
    ix.close()
    d = tempfile.mkdtemp(prefix="slkclilong_")
    loc = os.path.join(d, "lib")
    conv.write_slkrec(loc + ".slkrec", keys, tx)
    with open(loc + ".properties", "w") as f:
        f.write("k=35\nm=31\nversion=1\nsplitter=randomXOR\nminimizerSpaces=7\n")
    write_ranked_taxonomy(loc + "_taxonomy", parents)
    starts = rng.integers(0, G * L - LEN, R)
    fq = os.path.join(d, "long.fq")
    qual = b"I" * LEN
    with open(fq, "wb") as f:
        for i in range(R):
            s = bases[starts[i]:starts[i] + LEN].tobytes()
            if i % 100 == 0:
                s = s[:LEN // 3] + b"N" * 50 + s[LEN // 3 + 50:]
            f.write(b"@long%d\n%s\n+\n%s\n" % (i, s, qual))
    out = dict(reads=R, read_len=LEN, fastq_MB=round(os.path.getsize(fq) / 1e6))
    os.environ["SLK_HOST_TIMING"] = "1"
    for name, extra in (("detailed", []), ("reports_only", ["--nodetailed"])):
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(ROOT, "slacken_amd", "bin", "slacken-amd"), "classify", "-i", loc, "-o",
                            os.path.join(d, "out_" + name), *extra, fq], capture_output=True, text=True)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr
        out[name] = dict(seconds=round(dt, 2), Gbp_per_s=round(R * LEN / dt / 1e9, 2),
                         log=[l for l in r.stderr.split("\n") if "task" in l or "host timing" in l])
    print(json.dumps(out))


if __name__ == "__main__":
    main()

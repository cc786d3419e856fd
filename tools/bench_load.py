#!/usr/bin/env python3
"""Library load time of the CLI from Slacken's Parquet layout (snappy bucket files): N records in F files.  GPU box."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import taxgen
    from test_host_classify2_gpu import write_ranked_taxonomy
    N, F = int(float(os.environ.get("N", 2e8))), int(os.environ.get("F", 64))
    rng = np.random.default_rng(1)
    d = tempfile.mkdtemp(prefix="slkload_")
    loc = os.path.join(d, "lib")
    os.makedirs(loc)
    parents = taxgen.taxonomy(8 * 64, rng)
    per = N // F
    for b in range(F):
        keys = rng.integers(-2**62, 2**62, per, dtype=np.int64) & ~np.int64(0x33333333)
        taxa = rng.integers(2, len(parents), per).astype(np.int32)
        pq.write_table(pa.table({"id1": keys, "taxon": taxa}), os.path.join(loc, f"part-00000-x_{b:05d}.c000.snappy.parquet"),
                       compression="snappy")
    with open(loc + ".properties", "w") as f:
        f.write("k=35\nm=31\nversion=1\nsplitter=randomXOR\nminimizerSpaces=7\n")
    write_ranked_taxonomy(loc + "_taxonomy", parents)
    fq = os.path.join(d, "r.fq")
    open(fq, "w").write("@a\n" + "ACGT" * 30 + "\n+\n" + "I" * 120 + "\n")
    size = sum(os.path.getsize(os.path.join(loc, f)) for f in os.listdir(loc))
    out = {}
    for threads in (1, 0):
        env = dict(os.environ)
        if threads:
            env["SLK_HOST_THREADS"] = str(threads)
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(ROOT, "slacken_amd", "bin", "slacken-amd"), "classify", "-i", loc, "-o", os.path.join(d, "o"), fq],
                           capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        load = [l for l in r.stderr.split("\n") if "Load index" in l][0]
        tim = [l for l in r.stderr.split("\n") if "library load" in l]
        out["threads_%s" % (threads or "default")] = dict(wall=round(time.perf_counter() - t0, 2), load=load.split("[")[1].rstrip("]"), timing=tim)
    out.update(records=per * F, files=F, parquet_GB=round(size / 1e9, 2))
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/stress_cli.py [RUNS] -- the host pipeline of `slacken-amd classify` under random thread counts and chunk sizes: the same
inputs (plain FASTQ, one gzip file, a BGZF-like file, pairs as two gzip files; the reads repeat the golden reads many times over,
so that many batches are in flight) must give byte-identical per-read lines and reports every time.  GPU box."""
import glob
import gzip
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def digest(prefix):
    h = hashlib.md5()
    for d in sorted(glob.glob(prefix + "_c*")):
        for fn in sorted(glob.glob(os.path.join(d, "sample=*", "part-*.txt.gz"))):
            h.update(gzip.open(fn, "rb").read())
        for fn in sorted(glob.glob(os.path.join(d, "*_kreport.txt"))):
            h.update(open(fn, "rb").read())
    return h.hexdigest()


def main():
    from pathlib import Path
    from test_host_classify_gpu import make_library
    from test_host_cli import CLI
    from test_pargz import bgzf
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    d = Path(tempfile.mkdtemp(prefix="slkstress_"))
    g, loc, tax, reads = make_library(d)
    rng = np.random.default_rng(1)
    text = "".join(f"@{t}_{rep} x\n{s}\n+\n{'I' * len(s)}\n" for rep in range(60) for t, s in reads).encode()
    half = [(t, s) for t, s in reads if len(s) > 60]
    p1 = "".join(f"@{t}_{rep}/1\n{s[:len(s) // 2]}\n+\n{'I' * (len(s) // 2)}\n" for rep in range(40) for t, s in half).encode()
    p2 = "".join(f"@{t}_{rep}/2\n{s[len(s) // 2:]}\n+\n{'I' * (len(s) - len(s) // 2)}\n" for rep in range(40) for t, s in half).encode()
    files = {"plain": [d / "r.fq"], "gz": [d / "r1.fq.gz"], "bgzf": [d / "r2.fq.gz"], "pairs_gz": [d / "p_1.fq.gz", d / "p_2.fq.gz"]}
    open(files["plain"][0], "wb").write(text)
    open(files["gz"][0], "wb").write(gzip.compress(text, 6))
    open(files["bgzf"][0], "wb").write(bgzf(text))
    open(files["pairs_gz"][0], "wb").write(gzip.compress(p1, 6))
    open(files["pairs_gz"][1], "wb").write(gzip.compress(p2, 6))
    want = {}
    bad = 0
    for run in range(runs):
        for name, inputs in files.items():
            env = dict(os.environ)
            if run:   # (run 0: the defaults)
                env.update(SLK_CLASSIFY_THREADS=str(rng.integers(1, 7)), SLK_HOST_THREADS=str(rng.integers(1, 17)),
                           SLK_PARSE_THREADS=str(rng.integers(1, 9)), SLK_GZ_THREADS=str(rng.integers(0, 17)),
                           SLK_GZ_CHUNK=str(rng.choice([20_000, 100_000, 1 << 20])), SLK_GZ_GROUP=str(rng.integers(1, 6)),
                           SLK_IO_CHUNK=str(rng.choice([50_000, 400_000, 16 << 20])))
            out = str(d / f"out_{name}")
            for old in glob.glob(out + "_c*"):
                subprocess.call(["rm", "-rf", old])
            extra = ["-p"] if name == "pairs_gz" else []
            if run and rng.random() < 0.3:
                extra += ["--devices", "0,0,0" if rng.random() < 0.5 else "0,0"]   # (the same GPU as several devices: the multi-device host)
            r = subprocess.run([CLI, "classify", "-i", loc, "-o", out, "-c", "0.0", "0.15", *extra, *map(str, inputs)], capture_output=True, text=True, env=env)
            if r.returncode != 0:
                bad += 1
                print("FAILED", run, name, {k: v for k, v in env.items() if k.startswith("SLK_")}, r.stderr[-300:])
                continue
            dg = digest(out)
            key = "single" if name != "pairs_gz" else "pairs"   # plain, gz and bgzf hold the same reads
            if want.setdefault(key, dg) != dg:
                bad += 1
                print("DIFFERENT OUTPUT", run, name, {k: v for k, v in env.items() if k.startswith("SLK_")})
    print("runs", runs, "bad", bad, "digests", want)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

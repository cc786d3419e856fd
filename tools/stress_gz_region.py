"""tools/stress_gz_region.py FROM TO -- seeds FROM..TO-1: random FASTQ as one gzip member, as BGZF blocks or as a few members, through
`slacken-amd parse` with zlib (SLK_GZ_THREADS=0) and with the parallel inflate in place (random threads, chunk size, parse threads,
group size): the records must be the same.  Several ranges can run side by side."""
import gzip, hashlib, io, os, subprocess, sys, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_pargz import fastq_text, bgzf
CLI = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slacken_amd", "bin", "slacken-amd")
bad = 0
d = "/tmp/stress_region"; os.makedirs(d, exist_ok=True)
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    text = fastq_text(rng, int(rng.integers(200, 6000)), read_len=int(rng.integers(30, 400)))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        blob = gzip.compress(text, int(rng.integers(1, 10)))
    elif kind == 1:
        blob = bgzf(text, block=int(rng.integers(2000, 65000)))
    else:
        cuts = sorted(rng.integers(0, len(text), int(rng.integers(1, 8))).tolist())
        parts = [text[a:b] for a, b in zip([0] + cuts, cuts + [len(text)])]
        blob = b"".join(gzip.compress(p, 6) for p in parts)
    path = f"{d}/s{sys.argv[1]}_{seed % 8}.fq.gz"
    open(path, "wb").write(blob)
    outs = []
    for threads in (0, int(rng.integers(1, 17))):
        chunk = int(rng.choice([300, 1000, 5000, 30000]))
        if len(blob) < 2 * chunk: chunk = max(64, len(blob) // 4)
        env = dict(os.environ, SLK_GZ_THREADS=str(threads), SLK_GZ_CHUNK=str(chunk), SLK_PARSE_THREADS=str(int(rng.integers(1, 7))), SLK_GZ_GROUP=str(int(rng.integers(1, 6))))
        p = subprocess.run([CLI, "parse", path], env=env, capture_output=True, timeout=120)
        outs.append((p.returncode, hashlib.md5(p.stdout).hexdigest(), p.stderr[-200:]))
    if outs[0][:2] != outs[1][:2] or outs[0][0] != 0:
        bad += 1
        print("MISMATCH seed", seed, kind, outs)
print("done", sys.argv[1], sys.argv[2], "bad", bad)

#!/usr/bin/env python3
"""End-to-end timing of `slacken-amd classify` (parse + H2D + kernels + D2H + formatting + gzip) on synthetic reads against a
small synthetic library.  Run on the GPU box; prints one JSON object."""
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
    R = int(os.environ.get("R", 2_000_000))
    rng = np.random.default_rng(3)
    parents = taxgen.taxonomy(8 * 64, rng)
    taxa = np.array(taxgen.defined_taxa(parents))
    G, L = 64, 1 << 20
    acgt = np.frombuffer(b"ACGT", np.uint8)
    bases = acgt[rng.integers(0, 4, G * L, dtype=np.uint8)]
    offsets = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
    ix = slacken_amd.Index(expected_records=G * L // 2, max_taxon=len(parents) - 1)
    ix.set_taxonomy(parents)
    ix.add_sequences(bases, offsets, rng.choice(taxa[len(taxa) // 2:], G).astype(np.int32))
    keys, tx = ix.export()
    ix.close()
    d = tempfile.mkdtemp(prefix="slkcli_")
    loc = os.path.join(d, "lib")
    conv.write_slkrec(loc + ".slkrec", keys, tx)
    with open(loc + ".properties", "w") as f:
        f.write("k=35\nm=31\nversion=1\nsplitter=randomXOR\nminimizerSpaces=7\n")
    write_ranked_taxonomy(loc + "_taxonomy", parents)
    starts = rng.integers(0, G * L - 150, R)
    fq = os.path.join(d, "reads.fq")
    with open(fq, "wb") as f:
        CH = 100000
        qual = b"I" * 150
        for s in range(0, R, CH):
            e = min(R, s + CH)
            blk = bases[(starts[s:e, None] + np.arange(150)[None, :])]
            f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (s + i, blk[i].tobytes(), qual) for i in range(e - s)))
    out = {}
    os.environ["SLK_HOST_TIMING"] = "1"
    # a paired sample: the first half of the reads as /1, the second half as /2
    half = R // 2
    p1, p2 = os.path.join(d, "pair_1.fq"), os.path.join(d, "pair_2.fq")
    with open(p1, "wb") as f1, open(p2, "wb") as f2:
        CH = 100000
        qual = b"I" * 150
        for s in range(0, half, CH):
            e = min(half, s + CH)
            blk = bases[(starts[s:e, None] + np.arange(150)[None, :])]
            blk2 = bases[(starts[half + s:half + e, None] + np.arange(150)[None, :])]
            f1.write(b"".join(b"@r%d/1\n%s\n+\n%s\n" % (s + i, blk[i].tobytes(), qual) for i in range(e - s)))
            f2.write(b"".join(b"@r%d/2\n%s\n+\n%s\n" % (s + i, blk2[i].tobytes(), qual) for i in range(e - s)))
    cases = os.environ.get("SLK_CLI_CASES", "detailed,reports_only,paired_detailed,gz,classify2").split(",")
    for name, extra, inputs, n in (("detailed", [], [fq], R), ("reports_only", ["--nodetailed"], [fq], R),
                                   ("paired_detailed", ["-p"], [p1, p2], half)):
        if name not in cases:
            continue
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(ROOT, "slacken_amd", "bin", "slacken-amd"), "classify", "-i", loc, "-o",
                            os.path.join(d, "out_" + name), *extra, *inputs], capture_output=True, text=True)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr
        out[name + "_log"] = [l for l in r.stderr.split("\n") if "task" in l or "fragments" in l or "host timing" in l]
        calls = [l for l in r.stderr.split("\n") if l.startswith("slk_classify_batch")]
        if calls:
            out[name + "_calls"] = calls[:3] + calls[3::max(1, len(calls) // 16)]
        out[name] = dict(fragments=n, seconds=round(dt, 2), M_fragments_per_s=round(n / dt / 1e6, 3))
    if "gz" not in cases and "classify2" not in cases:
        print(json.dumps(out))
        return
    # the same reads as one gzip file and as eight: input files are inflated and parsed side by side
    import gzip
    parts = [os.path.join(d, f"part{i}.fq.gz") for i in range(8)]
    lines = open(fq, "rb").read().split(b"\n")
    per = (len(lines) // 4 + 7) // 8 * 4
    for i, pth in enumerate(parts):
        with gzip.open(pth, "wb", compresslevel=1) as f:
            f.write(b"\n".join(lines[i * per:(i + 1) * per]) + b"\n")
    one = os.path.join(d, "all.fq.gz")
    with gzip.open(one, "wb", compresslevel=1) as f:
        f.write(open(fq, "rb").read())
    # each with zlib on one thread per file (SLK_GZ_THREADS=0: how the input was read until round 2's pargz.hpp) and with the
    # parallel inflate (default); SLK_CLI_GZ_VARIANTS='{"name": {"ENV": "value"}, ...}' adds settings; best of two runs each
    variants = {"zlib_one_thread_per_file": {"SLK_GZ_THREADS": "0"}, "": {}}
    variants.update(json.loads(os.environ.get("SLK_CLI_GZ_VARIANTS", "{}")))
    for name, inputs in (("gz_one_file", [one]), ("gz_eight_files", parts)):
        for vname, venv in variants.items():
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                r = subprocess.run([os.path.join(ROOT, "slacken_amd", "bin", "slacken-amd"), "classify", "-i", loc, "-o",
                                    os.path.join(d, "out_" + name), *inputs], capture_output=True, text=True, env=dict(os.environ, **venv))
                dt = time.perf_counter() - t0
                assert r.returncode == 0, r.stderr
                if best is None or dt < best[0]:
                    best = (dt, [l for l in r.stderr.split("\n") if "host timing" in l])
            key = name + ("_" + vname if vname else "")
            out[key + "_log"] = best[1]
            out[key] = dict(reads=R, seconds=round(best[0], 2), M_reads_per_s=round(R / best[0] / 1e6, 3))
    # the paired sample as two gzip files (how paired-end runs usually arrive): two parallel inflates side by side; same report
    if "paired_detailed" in cases:
        pz = []
        for src in (p1, p2):
            dst = src + ".gz"
            with gzip.open(dst, "wb", compresslevel=1) as f:
                f.write(open(src, "rb").read())
            pz.append(dst)
        for vname, venv in (("zlib_one_thread_per_file", {"SLK_GZ_THREADS": "0"}), ("", {})):
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                r = subprocess.run([os.path.join(ROOT, "slacken_amd", "bin", "slacken-amd"), "classify", "-i", loc, "-o",
                                    os.path.join(d, "out_paired_gz"), "-p", *pz], capture_output=True, text=True, env=dict(os.environ, **venv))
                dt = time.perf_counter() - t0
                assert r.returncode == 0, r.stderr
                best = dt if best is None else min(best, dt)
            a = open(os.path.join(d, "out_paired_gz_c0.0", "all_kreport.txt")).read()
            b = open(os.path.join(d, "out_paired_detailed_c0.0", "all_kreport.txt")).read()
            assert a == b, "paired gzip input: the report differs from the plain files'"
            out["paired_gz" + ("_" + vname if vname else "")] = dict(fragments=half, seconds=round(best, 2), M_fragments_per_s=round(half / best / 1e6, 3))
    # classify2: the two-step run with a dynamic library built on the device from the genome FASTA files
    libdir = os.path.join(d, "k2")
    os.makedirs(os.path.join(libdir, "library", "bacteria"))
    gtaxa = rng.choice(taxa[len(taxa) // 2:], G).astype(np.int32)   # (labels only: the base index above has its own)
    with open(os.path.join(libdir, "library", "bacteria", "library.fna"), "wb") as f, open(os.path.join(libdir, "seqid2taxid.map"), "w") as mp:
        for gi in range(G):
            f.write(b">G%d synthetic genome\n" % gi)
            g = bases[gi * L:(gi + 1) * L]
            f.write(b"\n".join(g[j:j + 80].tobytes() for j in range(0, L, 80)) + b"\n")
            mp.write(f"G{gi}\t{int(gtaxa[gi])}\n")
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(ROOT, "slacken_amd", "bin", "slacken-amd"), "classify2", "-i", loc, "-o", os.path.join(d, "out2"),
                        "--library", libdir, "-R", "100", "--rank", "superkingdom", fq], capture_output=True, text=True)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr
    out["classify2"] = dict(reads=R, library_Mbp=G * L // 1000000, seconds=round(dt, 2), M_reads_per_s=round(R / dt / 1e6, 3),
                            log=[l for l in r.stderr.split("\n") if "task" in l or "Detected" in l or "dynamic" in l or "Construct" in l])
    out["fastq_MB"] = round(os.path.getsize(fq) / 1e6)
    out["records"] = len(keys)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

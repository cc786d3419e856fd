#!/usr/bin/env python3
"""tools/make_traffic.py <profile dir> <bench json of the same command> -> profiles/r03_traffic.json
HBM traffic of the dominant kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh (separate rocprofv3 --pmc runs,
per-dispatch means), stamped with the hash of the kernel sources it was measured on: bench.py quotes it only on a match."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

out_dir, bench_json = sys.argv[1], sys.argv[2]
vals = {}
for f in glob.glob(os.path.join(out_dir, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    acc, cnt = {}, {}
    for r in csv.DictReader(open(f)):
        if not r["Kernel_Name"].replace(" ", "").startswith("voidslk::lane_kernel<true,0,false,false>"):   # the hot variant only
            continue
        c = r["Counter_Name"]
        acc[c] = acc.get(c, 0.0) + float(r["Counter_Value"])
        cnt[c] = cnt.get(c, 0) + 1
    for c in acc:
        vals[c] = acc[c] / cnt[c]
line = [json.loads(l) for l in open(bench_json) if l.startswith("{")][-1]
cfg = line["config"]
fetch_kb, write_kb = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
hbm = (fetch_kb + write_kb) * 1024
doc = {
    "kernel": "slk::lane_kernel<true, 0, false, false>",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --no-cpu-baseline --steps 2 --warmup 1`, "
              "per-dispatch mean (tools/profile.sh, tools/make_traffic.py)",
    "kernel_source_hash": bench.kernel_source_hash(),
    "fetch_size_kb": fetch_kb, "write_size_kb": write_kb, "hbm_bytes_per_launch": int(hbm),
    "tcc_miss_x64_bytes": int(vals["TCC_MISS_sum"] * 64) if "TCC_MISS_sum" in vals else None,
    "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"],
    "ratio_to_algorithmic": round(hbm / line["roofline"]["algorithmic_bytes_per_launch"], 4),
    "note": "FETCH_SIZE needs no x2 correction for this access shape (64-byte gathers, 4 lanes x 16 B): TCC_MISS_sum x 64 B is its "
            "cross-check; the half-count of MI355X_MICROARCH.md applies to wide coalesced streams.  Infinity-Cache hits are counted "
            "as traffic by these counters.",
    "reads_per_launch": cfg["reads_per_gpu_per_step"], "records": int(round(cfg["records"], -5)) if False else None,
    "genomes": [cfg["genomes"], cfg["genome_len"]],
}
doc["records"] = int(float(os.environ.get("SLK_BENCH_RECORDS", "1e10")))
json.dump(doc, open(os.path.join(ROOT, "profiles", "r03_traffic.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))

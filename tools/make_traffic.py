#!/usr/bin/env python3
"""tools/make_traffic.py <profile dir> <bench json of the same command> -> profiles/r04_traffic.json
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
newest = {}
for f in glob.glob(os.path.join(out_dir, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    d = os.path.relpath(f, out_dir).split(os.sep)[0]      # (a directory merged from several sessions holds one file per session: the last)
    if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
        newest[d] = f
for f in newest.values():
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
# The read stream is fetched with wave-wide coalesced loads (lane.hip: SLK_PACKED_STREAM; every tile of this workload: 64 x 150 bases
# <= 10 880).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports exactly half of the bytes of such a read (128-byte requests tallied
# at 64): that half is added back.  The probes (4 lanes x 16 B = 64-byte requests) need no correction.
stream_bytes = cfg["reads_per_gpu_per_step"] * cfg["read_len"]
hbm = (fetch_kb + write_kb) * 1024 + stream_bytes // 2
doc = {
    "kernel": "slk::lane_kernel<true, 0, false, false>",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --no-cpu-baseline --steps 2 --warmup 1`, "
              "per-dispatch mean (tools/profile.sh, tools/make_traffic.py)",
    "kernel_source_hash": bench.kernel_source_hash(),
    "fetch_size_kb": fetch_kb, "write_size_kb": write_kb, "coalesced_stream_bytes": int(stream_bytes),
    "fetch_size_correction_bytes": int(stream_bytes // 2), "hbm_bytes_per_launch": int(hbm),
    "tcc_miss_x64_bytes": int(vals["TCC_MISS_sum"] * 64) if "TCC_MISS_sum" in vals else None,
    "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"],
    "ratio_to_algorithmic": round(hbm / line["roofline"]["algorithmic_bytes_per_launch"], 4),
    "note": "hbm_bytes_per_launch = (FETCH_SIZE + WRITE_SIZE) x 1024 + half the read stream's bytes: the stream is fetched with wide "
            "coalesced loads, of which gfx950's FETCH_SIZE reports exactly half (MI355X_MICROARCH.md); the probes (64-byte gathers, 4 lanes x "
            "16 B) need no correction.  TCC_MISS_sum counts requests, a 128-byte one once.  Infinity-Cache hits are counted as traffic by "
            "these counters.",
    "reads_per_launch": cfg["reads_per_gpu_per_step"], "records": int(round(cfg["records"], -5)) if False else None,
    "genomes": [cfg["genomes"], cfg["genome_len"]],
}
doc["records"] = int(float(os.environ.get("SLK_BENCH_RECORDS", "1e10")))
json.dump(doc, open(os.path.join(ROOT, "profiles", "r04_traffic.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))

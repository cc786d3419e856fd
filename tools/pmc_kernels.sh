#!/bin/bash
# tools/pmc_kernels.sh TAG "COUNTERS" [bench args] -- one rocprofv3 PMC pass of bench.py; per-dispatch means per lane kernel variant
TAG=$1; PMC=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --pmc $PMC --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $OUT/bench.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
python3 - <<PY
import csv, glob, collections, json
acc=collections.defaultdict(float); cnt=collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lane_kernel" in r["Kernel_Name"]:
            k=(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
            acc[k]+=float(r["Counter_Value"]); cnt[k]+=1
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("$TAG", d["value"], d["ms_per_step"])
for k in sorted(acc):
    if acc[k]/cnt[k] > 1e6: print("  ", k[0], k[1], round(acc[k]/cnt[k]))
PY

#!/bin/bash
# tools/pmc_one.sh TAG "COUNTERS" [bench args]  -- one rocprofv3 PMC pass of bench.py; prints the lane kernel's per-dispatch means
TAG=$1; PMC=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --pmc $PMC --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $OUT/bench.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(float); cnt=collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lane_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
print("$TAG", {k: round(v/cnt[k]) for k,v in acc.items()})
PY

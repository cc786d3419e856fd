#!/bin/bash
# LDS conflict share of the hot kernel with and without the fold round (timing build)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
for AB in 0 2 1; do
  OUT=gpurun_out/r02/pmc_lds_$AB
  rm -rf $OUT
  SLACKEN_AMD_LIB=$PWD/build_ab/libslacken_tuning.so SLK_DEBUG_ABLATE=$AB rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $OUT.json 2> $OUT.err || { tail -3 $OUT.err; exit 1; }
  python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(float); cnt=collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].replace(" ","").startswith("voidslk::lane_kernel<true,0,false,false>"):
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
print("ablate=$AB", {k: round(v/cnt[k]) for k,v in acc.items()})
PY
  rm -rf $OUT
done

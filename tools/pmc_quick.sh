#!/bin/bash
# tools/pmc_quick.sh <tag> -- instruction-count and wait-cycle PMC passes only (faster than tools/profile.sh)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $OUT/bench_pmc$i.json 2> $OUT/pmc$i.err
  echo "pmc$i rc=$?"
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -E "lane_kernel" $OUT/summary.txt

#!/bin/bash
# tools/profile.sh <tag> [bench args...] -- rocprofv3 kernel trace + PMC passes of bench.py (run on the GPU box via gpurun).
# Summaries land in gpurun_out/prof_<tag>/; copy what should be judged into profiles/.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
# (20 timed steps: the one warm-up launch, cold and a tenth slower, then weighs 1/21 in the trace's average per launch)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 1 "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 bench.py --no-cpu-baseline --no-ceiling --steps 2 --warmup 1 "$@" > $OUT/bench_pmc$i.json 2> $OUT/pmc$i.err
  echo "pmc$i ($PMC) rc=$?"
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
python3 tools/make_traffic.py $OUT $OUT/bench_trace.json > $OUT/traffic.log 2>&1 && cp profiles/r04_traffic.json $OUT/ || tail -3 $OUT/traffic.log

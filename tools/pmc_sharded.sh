#!/bin/bash
# tools/pmc_sharded.sh -- counters of the table-sharded step kernel (lane_step_kernel) on bench.py --table-sharded (world 1)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_sharded; rm -rf $OUT; mkdir -p $OUT
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 bench.py --table-sharded --steps 8 --warmup 5 --no-ceiling "$@" > $OUT/b$i.json 2> $OUT/pmc$i.err
  echo "pmc$i rc=$?"
done
python3 tools/summarize_prof.py $OUT 2>&1 | grep "lane_step_kernel\|lookup_coop"

#!/bin/bash
# tools/pmc_long.sh -- counters of the long-read kernels on tools/bench_long.py (LENGTHS, SLK_SEG_MIN_LEN from the environment)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_long; rm -rf $OUT; mkdir -p $OUT
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 tools/bench_long.py > $OUT/b$i.json 2> $OUT/pmc$i.err
  echo "pmc$i rc=$?"
done
python3 tools/summarize_prof.py $OUT 2>&1 | grep -v "at::\|rocclr\|build_kernel\|insert" 

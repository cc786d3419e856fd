#!/bin/bash
# round-3 soak on the GPU box: the whole GPU suite with many seeds, the sizing soak, long reads with hit lists on both routes
export SLK_FUZZ_SEEDS=${SLK_FUZZ_SEEDS:-400} SLK_SEG_SEEDS=${SLK_SEG_SEEDS:-80} SLK_DEEP_SEEDS=${SLK_DEEP_SEEDS:-100} SLK_TITLE_SEEDS=${SLK_TITLE_SEEDS:-12}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py -p no:cacheprovider > gpurun_out/soak_r03.log 2>&1; echo "soak rc=$?"; tail -4 gpurun_out/soak_r03.log
timeout -k 10 300 python tools/soak_sizing.py > gpurun_out/soak_sizing_r03.log 2>&1; echo "sizing rc=$?"; tail -3 gpurun_out/soak_sizing_r03.log

timeout -k 10 600 python -m pytest tests/test_gpu_segments.py -m gpu -x -q > gpurun_out/t12.log 2>&1; echo "seg tests rc=$?"; tail -3 gpurun_out/t12.log
timeout -k 10 300 python tools/bench_long_hits.py > gpurun_out/r03_long_hits_segment.json 2> gpurun_out/r03_long_hits_segment.err; echo "rc=$?"; cat gpurun_out/r03_long_hits_segment.json
SLK_SEG_MIN_LEN=0 timeout -k 10 300 python tools/bench_long_hits.py > gpurun_out/r03_long_hits_wave.json 2> gpurun_out/r03_long_hits_wave.err; echo "rc=$?"; cat gpurun_out/r03_long_hits_wave.json
timeout -k 10 400 python tools/soak_sizing.py 300 > gpurun_out/soak_sizing_r03.log 2>&1; echo "sizing rc=$?"; tail -3 gpurun_out/soak_sizing_r03.log

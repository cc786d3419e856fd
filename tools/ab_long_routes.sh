#!/bin/bash
# Long unpaired reads by length on each of the two kernels that take them (tools/bench_long.py, 1 Gbp per batch, bases resident):
# the lane-per-segment kernel from 1001 bases on (SLK_SEG_MIN_LEN=5001: everything here) and the wave-per-fragment kernel alone.
export LENGTHS=${LENGTHS:-5000,7500,10000,15000,20000,30000,60000,100000,300000}
fmt='import json,sys; d=json.loads(sys.stdin.read()); print({k:v["Gbp_per_s"] for k,v in d.items() if isinstance(v, dict) and "with_N" not in k})'
echo "== segment kernel (SLK_SEG_MIN_LEN=5000), Gbp/s"; SLK_SEG_MIN_LEN=5000 timeout -k 10 300 python tools/bench_long.py 2>gpurun_out/routes.err | python -c "$fmt" || exit 1
echo "== wave kernel (SLK_SEG_MIN_LEN=0), Gbp/s"; SLK_SEG_MIN_LEN=0 timeout -k 10 300 python tools/bench_long.py 2>gpurun_out/routes.err | python -c "$fmt" || exit 1
echo "== default routes, Gbp/s"; timeout -k 10 300 python tools/bench_long.py 2>gpurun_out/routes.err | python -c "$fmt" || exit 1

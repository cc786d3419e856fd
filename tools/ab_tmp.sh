#!/bin/bash
export LENGTHS=1500,10000 SLK_SEG_MIN_LEN=5000
fmt='import json,sys; d=json.loads(sys.stdin.read()); print({k:v["Gbp_per_s"] for k,v in d.items() if isinstance(v, dict) and "with_N" not in k})'
for lf in 0.35 0.45 0.5 0.55; do echo "== load $lf"; LOAD=$lf timeout -k 10 300 python tools/bench_long.py 2>gpurun_out/x.err | python -c "$fmt" || exit 1; done
echo "== pow2 0.5"; EXPECTED=134217722 LOAD=0.5 timeout -k 10 300 python tools/bench_long.py 2>gpurun_out/x.err | python -c "$fmt" || exit 1
echo "== pow2 0.25"; EXPECTED=134217722 LOAD=0.25 timeout -k 10 300 python tools/bench_long.py 2>gpurun_out/x.err | python -c "$fmt" || exit 1
echo "== r02"; ( cd build_ab/r02 && timeout -k 10 200 python tools/bench_long.py 2>/dev/null | python -c "$fmt" )

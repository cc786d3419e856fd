#!/bin/bash
# tools/ab_libs.sh RECORDS ROUNDS lib1.so lib2.so ... -- the headline bench per library build, ROUNDS times round-robin on one box
rec=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    name=$(basename $lib .so)
    SLACKEN_AMD_LIB=$PWD/$lib timeout -k 10 500 python bench.py --records $rec --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/abx_$name.json 2> gpurun_out/abx_$name.err || exit 1
    python - <<PY
import json
d=json.loads(open("gpurun_out/abx_$name.json").read().strip().splitlines()[-1])
print("round $r", "$name", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
  done
done

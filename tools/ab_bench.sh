#!/bin/bash
# A/B of library builds on the GPU box: tools/ab_bench.sh RECORDS lib1.so lib2.so ...  (results under gpurun_out/ab_*.json)
rec=$1; shift
for lib in "$@"; do
  name=$(basename $lib .so)
  SLACKEN_AMD_LIB=$PWD/$lib timeout -k 10 500 python bench.py --records $rec --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
done

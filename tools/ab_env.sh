#!/bin/bash
# A/B of environment settings on the GPU box: tools/ab_env.sh RECORDS "VAR=1" "VAR=2 OTHER=x" ...
rec=$1; shift
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 500 python bench.py --records $rec --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/abenv_$i.json 2> gpurun_out/abenv_$i.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/abenv_$i.json").read().strip().splitlines()[-1])
print("$e", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
done

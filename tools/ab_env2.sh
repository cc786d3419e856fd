#!/bin/bash
# tools/ab_env2.sh RECORDS -- the headline bench with the current library, with the segment passes off, with another build of the
# library (slacken_amd/lib/libslacken_amd_oldlane.so, e.g. an older lane.hip compiled aside; skipped when absent), and again
rec=$1
run() { name=$1; shift; env "$@" timeout -k 10 500 python bench.py --records $rec --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
}
run cur1 X=1
run noseg SLK_SEG_MIN_LEN=0
if [ -f slacken_amd/lib/libslacken_amd_oldlane.so ]; then
  run oldlane SLACKEN_AMD_LIB=$PWD/slacken_amd/lib/libslacken_amd_oldlane.so
  run oldlane_noseg SLACKEN_AMD_LIB=$PWD/slacken_amd/lib/libslacken_amd_oldlane.so SLK_SEG_MIN_LEN=0
fi
run cur2 X=1
run noseg2 SLK_SEG_MIN_LEN=0

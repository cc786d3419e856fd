#!/bin/bash
# tools/ab_geometry.sh RECORDS "lf1 lf2 .." lib1.so lib2.so ... -- the headline bench per library build and table load factor
# (bucket size is a compile-time constant: make LIB=build_ab/c8.so EXTRA=-DSLK_BUCKET_CELLS=8 build_ab/c8.so)
rec=$1; lfs=$2; shift 2
for lib in "$@"; do
  for lf in $lfs; do
    name=$(basename $lib .so)_lf$lf
    SLACKEN_AMD_LIB=$PWD/$lib timeout -k 10 400 python bench.py --records $rec --load-factor $lf --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/abg_$name.json 2> gpurun_out/abg_$name.err || { tail -5 gpurun_out/abg_$name.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/abg_$name.json").read().strip().splitlines()[-1])
c=d["config"]
print("$name", "M reads/s", d["value"], "ms", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "table GiB", c["table_GiB"], "load", c.get("table_load"), "maxdisp", c.get("max_displacement"))
PY
  done
done

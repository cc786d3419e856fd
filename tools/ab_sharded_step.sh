#!/bin/bash
# A/B of build variants of the table-sharded step kernel (build_ab/lib_<name>.so, built with -DSLK_STREAM_NT / -DSLK_APPLY_ROWS):
# bench.py --table-sharded on each, the shipped library first and last.  Run on the GPU box; writes gpurun_out/ab_sharded_step.txt
cd "$(dirname "$0")/.."
out=gpurun_out/ab_sharded_step.txt
: > $out
for lib in "" build_ab/lib_nt1_u4.so build_ab/lib_nt0_u8.so build_ab/lib_nt1_u8.so ""; do
  name=${lib:-shipped}
  SLACKEN_AMD_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python bench.py --table-sharded --steps 12 --warmup 5 --no-ceiling > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.log || { echo "$name FAILED" >> $out; tail -3 gpurun_out/ab_tmp.log >> $out; continue; }
  python - "$name" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
c = d["config"]
print(f"{sys.argv[1]:28s} {d['value']:8.1f} M reads/s  {d['ms_per_step']:7.3f} ms/batch  steady step {c['steady_state_step_ms']:7.3f} ms  alone {c['job_ms_alone']}")
PY
done
cat $out

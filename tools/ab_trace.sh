#!/bin/bash
# tools/ab_trace.sh lib1.so lib2.so ... -- rocprofv3 kernel trace of the headline bench per library build (per-kernel durations side by side)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for lib in "$@"; do
  name=$(basename $lib .so)
  OUT=$ROOT/gpurun_out/abt_$name
  rm -rf $OUT; mkdir -p $OUT
  export SLACKEN_AMD_LIB=$ROOT/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
  echo "== $name"; tail -1 $OUT/bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  python3 tools/summarize_prof.py $OUT | grep -E "lane_kernel|fused_kernel|segment_kernel|memset|fill" 
done

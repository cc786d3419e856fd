#!/bin/bash
# tools/ab_phylo.sh lib1.so lib2.so ... -- bench_phylo.py (structured library) and bench.py (headline, 1e10) per library build
for lib in "$@"; do
  name=$(basename $lib .so)
  SLACKEN_AMD_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_phylo.py 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$name phylo', d['thresholds_1'])"
  SLACKEN_AMD_LIB=$PWD/$lib SLK_DEBUG_OCC=1 timeout -k 10 300 python bench.py --no-cpu-baseline 2>gpurun_out/ab_occ.err | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$name headline', d['value'], d['ms_per_step'])"
  grep "slk. lane" gpurun_out/ab_occ.err | head -1
done

#!/bin/bash
# tools/ab_hot.sh LIB... -- the default bench line (no CPU baseline, no ceiling run) with each library, the shipped one first and last
# (boxes differ by 5 %, so a change to the hot path is only judged on one box, in one call).  GPU box.
cd "$(dirname "$0")/.."
for lib in "" "$@" ""; do
  name=${lib:-shipped}
  SLACKEN_AMD_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python bench.py --no-cpu-baseline --no-ceiling 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-24s %8.1f M reads/s  %.3f ms/step  kernel %.3f ms' % ('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" || exit 1
done

#!/bin/bash
# tools/trace_long_mixed.sh [SLK_SEG_MIN_LEN] -- kernel trace of tools/bench_long_mixed.py (which kernel takes how long in a batch of
# mixed long reads); GPU box.  Prints the engine's kernels of the trace.
export TMPDIR=/tmp
OUT=gpurun_out/trace_long_mixed_${1:-default}
[ -n "$1" ] && export SLK_SEG_MIN_LEN=$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bench_long_mixed.py > $OUT/bench.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
cut -c1-230 $OUT/bench.json
python3 - $OUT <<'PY'
import csv, glob, os, re, sys
f = max(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if 'slk::' in r['Kernel_Name'] and 'build' not in r['Kernel_Name'] and 'insert' not in r['Kernel_Name']]
first = lambda n: 'route_kernel' in n or re.search(r'lane_kernel<[^>]*, false>', n) is not None   # a call starts with one of these
calls = []
for r in rows:
    if first(r['Kernel_Name']) or not calls:
        calls.append([])
    calls[-1].append(r)
for ci in (2, 5):            # the third timed call of the first two workloads (nanopore-like: random order, sorted)
    if ci >= len(calls): break
    c = calls[ci]
    t0 = int(c[0]['Start_Timestamp'])
    print(f'-- call {ci}')
    for r in c:
        print(f"  {r['Kernel_Name'][:60]:60s} start {(int(r['Start_Timestamp'])-t0)/1e6:8.3f} ms  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:8.3f} ms")
PY

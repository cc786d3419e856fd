#!/bin/bash
# tools/ab_long.sh lib1.so lib2.so ... -- tools/bench_long.py per library, segment kernel on (default) and off
for lib in "$@"; do
  name=$(basename $lib .so)
  SLACKEN_AMD_LIB=$PWD/$lib timeout -k 10 400 python tools/bench_long.py > gpurun_out/abl_${name}_seg.json 2> gpurun_out/abl_${name}_seg.err || exit 1
  SLK_SEG_MIN_LEN=0 SLACKEN_AMD_LIB=$PWD/$lib timeout -k 10 400 python tools/bench_long.py > gpurun_out/abl_${name}_wave.json 2> gpurun_out/abl_${name}_wave.err || exit 1
done
python - "$@" <<'PY'
import json, os, sys
names = [os.path.basename(l)[:-3] for l in sys.argv[1:]]
rows = {}
for n in names:
    for mode in ("seg", "wave"):
        d = json.load(open(f"gpurun_out/abl_{n}_{mode}.json"))
        for k, v in d.items():
            if k != "seg_min_len":
                rows.setdefault(k, {})[(n, mode)] = v["Gbp_per_s"]
print("length".ljust(24), *[f"{n[-8:]}/{m}".rjust(14) for n in names for m in ("seg", "wave")])
for k, r in rows.items():
    print(k.ljust(24), *[f"{r[(n, m)]:14.1f}" for n in names for m in ("seg", "wave")])
PY

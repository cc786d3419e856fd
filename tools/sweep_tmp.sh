for mode in "" "--separate-lookup"; do
  timeout -k 10 300 python bench.py --table-sharded --steps 10 --warmup 3 $mode > gpurun_out/ts_$mode.json 2> gpurun_out/ts_$mode.err; echo "rc=$?"; tail -2 gpurun_out/ts_$mode.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ts_$mode.json").read().strip().splitlines()[-1])
print("mode [$mode]", d["value"], "M reads/s", d["ms_per_step"], "ms", d["config"]["stage_ms_in_pipeline"], "ALONE", d["config"].get("stage_ms_alone"))
PY
done
timeout -k 10 300 python bench.py --table-sharded --steps 40 --warmup 3 > gpurun_out/ts_40.json 2> gpurun_out/ts_40.err; python - <<PY
import json
d=json.loads(open("gpurun_out/ts_40.json").read().strip().splitlines()[-1])
print("40 steps", d["value"], "M reads/s", d["ms_per_step"], "ms", d["config"]["stage_ms_in_pipeline"])
PY

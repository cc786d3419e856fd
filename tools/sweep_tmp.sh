for b in 2 4 8 16; do
  SLK_LOOKUP_BLOCKS_PER_CU=$b timeout -k 10 300 python bench.py --table-sharded --steps 8 --warmup 2 > gpurun_out/sw_lookup_$b.json 2> gpurun_out/sw_lookup_$b.err || { tail -3 gpurun_out/sw_lookup_$b.err; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/sw_lookup_$b.json").read().strip().splitlines()[-1])
print("blocks/CU $b", d["value"], d["ms_per_step"], d["config"]["stage_ms_in_pipeline"])
PY
done
timeout -k 10 300 python bench.py --records 2.0e10 --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r03_bench_records_2.0e10.json 2> gpurun_out/r03_bench_records_2.0e10.err; echo "records 2e10 rc=$?"; tail -3 gpurun_out/r03_bench_records_2.0e10.err; tail -c 900 gpurun_out/r03_bench_records_2.0e10.json
timeout -k 10 300 python bench.py --records 1.5e10 --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r03_bench_records_1.5e10.json 2> gpurun_out/r03_bench_records_1.5e10.err; echo "records 1.5e10 rc=$?"; tail -2 gpurun_out/r03_bench_records_1.5e10.err

mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -q -x > gpurun_out/gpu_tests_18.log 2>&1; tail -3 gpurun_out/gpu_tests_18.log
for s in 32; do timeout -k 10 300 python bench.py --steps 128 --warmup 2 --no-cpu-baseline --streams $s > gpurun_out/bench_s$s.json 2> gpurun_out/bench_s$s.err; tail -2 gpurun_out/bench_s$s.err; python - <<PY
import json
d=json.load(open("gpurun_out/bench_s$s.json")); print("users/batch=$s", round(d["value"],1), "items/s", round(d["ms_per_step"],2), "ms/user", {k: round(v,2) for k,v in d["per_user"].items()})
PY
done

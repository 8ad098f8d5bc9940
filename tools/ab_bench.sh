#!/bin/bash
# A/B of an env switch on the default (batched) bench line, alternating values on ONE box.  usage: tools/ab_bench.sh ENVVAR v1 v2 [v1 v2 ...]
var=$1; shift
i=0
for v in "$@"; do
  i=$((i+1))
  env $var=$v python bench.py --no-cpu-baseline --aligned-resid-scale "" --single-stream-users 0 $AB_ARGS 2>/dev/null > gpurun_out/abb_$i.json
  python - <<PY
import json
d=json.load(open("gpurun_out/abb_$i.json"))
print("$var=$v", round(d["value"],1), "items/s; all GEMMs", round(d["roofline"]["target_forward"]["all_gemms_tflops"],1), "TF; per user", {k: round(x,3) for k,x in d["per_user"].items()})
PY
done

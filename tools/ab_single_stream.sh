#!/bin/bash
# A/B of an env switch on the single-user-stream pass of bench.py.  usage: tools/ab_single_stream.sh ENVVAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --no-cpu-baseline --steps 1 --aligned-resid-scale "" --single-stream-users 12 2>/dev/null > gpurun_out/ab_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json")); s=d["single_user_stream"]
print("$var=$v", "batched", round(d["value"],1), "single-stream items/s", round(s["items_per_s"],1), "ms/user", round(s["ms_per_user"],2), "GB/s", round(s["roofline"]["achieved"]))
PY
done

#!/bin/bash
# kernel summaries of the short batched bench under several env settings.  usage: tools/kstats_ab.sh "VAR=a" "VAR=b VAR2=c" ...  ("" = defaults)
for e in "$@"; do echo "--- ${e:-defaults}"; env $e bash tools/kstats.sh | grep -E "attn|rope|items|value" ; python -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/kstats.log').read().strip().splitlines()[-2] if False else [l for l in open('gpurun_out/kstats.log') if l.startswith('{')][-1]); print('items/s', round(d['value'],1))
except Exception as ex: print('no bench line', ex)
"; done

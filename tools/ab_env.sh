#!/bin/bash
# A/B of env switches on a tool.  usage: tools/ab_env.sh "python tools/x.py" "VAR=a VAR2=b" "VAR=c" ...   ("" = defaults)
cmd=$1; shift
for e in "$@"; do echo "--- ${e:-defaults}"; env $e timeout -k 10 300 $cmd 2>&1 | grep -v amdgpu.ids || exit 1; done

#!/bin/bash
# A/B of an env switch on the one-user-at-a-time loop (no profiler), alternating on ONE box.  usage: tools/ab_single_user.sh ENVVAR v1 v2 [v1 v2 ...]
var=$1; shift
for v in "$@"; do printf "%s=%s  " $var $v; env $var=$v timeout -k 10 120 python tools/single_user_run.py ${SU_USERS:-12} 2>&1 | grep MARK || exit 1; done

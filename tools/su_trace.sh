#!/bin/bash
# kernel trace of the one-user-at-a-time loop + busy/gap summary -> gpurun_out/su_gaps.txt  (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/su
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/su -o su --output-format csv -- python3 tools/single_user_run.py ${1:-6} > gpurun_out/su.log 2>&1 || exit 1
python tools/trace_gaps.py $(find gpurun_out/su -name "*kernel_trace.csv") 0.6 > gpurun_out/su_gaps.txt
grep MARK gpurun_out/su.log; cat gpurun_out/su_gaps.txt; rm -rf gpurun_out/su

#!/bin/bash
# kernel trace of a short lock-step bench run + busy/gap summary of its last batches -> gpurun_out/batch_gaps.txt  (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/bt
timeout -k 10 500 rocprofv3 --kernel-trace -d gpurun_out/bt -o bt --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --aligned-resid-scale '' --single-stream-users 0 --no-configs --no-latency-curve "$@" > gpurun_out/bt.log 2>&1 || exit 1
python tools/trace_gaps.py $(find gpurun_out/bt -name "*kernel_trace.csv") ${FRAC:-0.5} 40 > gpurun_out/batch_gaps.txt
cat gpurun_out/batch_gaps.txt; rm -rf gpurun_out/bt

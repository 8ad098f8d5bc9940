#!/bin/bash
# kernel trace + rocprofv3 --stats of the small lock-step batches (1 / 4 / 16 users, BSSD and plain beam search) -> gpurun_out/band_*.txt and
# gpurun_out/band_stats_*.csv (copy the ones to be judged into profiles/).  Run on the GPU box from the repo root.
# usage: tools/band_trace.sh "1 4 16" "bssd tg" [tag]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${3:-band}
for u in ${1:-1 4 16}; do for mode in ${2:-bssd}; do
  rm -rf gpurun_out/bt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/bt -o bt --output-format csv -- python3 tools/batch_run.py $u $mode 6 > gpurun_out/${tag}_${mode}_u${u}.log 2>&1 || { tail -5 gpurun_out/${tag}_${mode}_u${u}.log; exit 1; }
  python tools/trace_gaps.py $(find gpurun_out/bt -name "*kernel_trace.csv") 0.6 > gpurun_out/${tag}_${mode}_u${u}_gaps.txt
  cp $(find gpurun_out/bt -name "*kernel_stats.csv") gpurun_out/${tag}_stats_${mode}_u${u}.csv
  grep MARK gpurun_out/${tag}_${mode}_u${u}.log; head -16 gpurun_out/${tag}_${mode}_u${u}_gaps.txt
done; done
rm -rf gpurun_out/bt

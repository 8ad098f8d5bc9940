# run on the GPU box: default bench line, kernel stats, PMC traffic passes -> gpurun_out/final (copy the summaries into profiles/ afterwards)
# usage (from the build container):  gpurun -- 'ATSPEED_COMMIT=<short sha> bash tools/profile_round.sh'
set -x
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err || exit 1
cp gpurun_out/bench_detail.json gpurun_out/final/bench_detail.json      # the full report of THIS run (the passes below overwrite gpurun_out/bench_detail.json)
cd /tmp && export TMPDIR=/tmp
# the same command under the kernel trace (sub-passes off: the summary is the headline workload's)
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --aligned-resid-scale '' --single-stream-users 0 --no-configs --no-latency-curve > $GRAFT_REPO_ROOT/gpurun_out/final/stats.log 2>&1 || exit 1
# HBM-side traffic: FETCH_SIZE and WRITE_SIZE in separate passes, no trace domains next to --pmc
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --aligned-resid-scale '' --single-stream-users 0 --no-configs --no-latency-curve > $GRAFT_REPO_ROOT/gpurun_out/final/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --aligned-resid-scale '' --single-stream-users 0 --no-configs --no-latency-curve > $GRAFT_REPO_ROOT/gpurun_out/final/pmc_write.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
find gpurun_out/final -name "*kernel_stats.csv" | head -3
# keep what is merged back small: the per-dispatch counter CSVs are summarised here, the traces dropped
python tools/pmc_summary.py gpurun_out/final gpurun_out/final/pmc_traffic.json > gpurun_out/final/pmc_summary.log 2>&1
cp $(find gpurun_out/final/stats -name "*kernel_stats.csv" | head -1) gpurun_out/final/kernel_stats.csv
rm -rf gpurun_out/final/stats gpurun_out/final/pmc_fetch gpurun_out/final/pmc_write
ls -la gpurun_out/final

#!/bin/bash
# kernel summary of a short batched bench run: top kernels by time -> gpurun_out/kstats.txt   (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kstats
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kstats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --aligned-resid-scale '' --single-stream-users 0 "$@" > $GRAFT_REPO_ROOT/gpurun_out/kstats.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kstats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("gpurun_out/kstats.txt", "w") as o:
    for r in rows[:14]:
        line = f"{r['Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:64]:64s} {int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:9.1f} us  {float(r['Percentage']):6.2f} %"
        print(line); o.write(line + "\n")
PY
tail -1 gpurun_out/kstats.log | cut -c1-200
rm -rf gpurun_out/kstats

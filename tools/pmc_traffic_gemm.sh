#!/bin/bash
# HBM-side traffic (L2 misses to the fabric) of ONE ring-GEMM shape per launch, for env-selected variants: separate rocprofv3 --pmc passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace off.  usage: tools/pmc_traffic_gemm.sh TAG M N K [EPI]   (env passes through; EPI 0 store / 2 residual / 3 SwiGLU)
set -e
TAG=$1; M=$2; N=$3; K=$4
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmct_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $OUT/f -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/one_gemm.py $M $N $K $5 > $OUT.f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/w -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/one_gemm.py $M $N $K $5 > $OUT.w.log 2>&1
python3 - <<PY
import csv, glob
def avg(d):
    v = [float(r["Counter_Value"]) for f in glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "gemm_ring" in r["Kernel_Name"]]
    return sum(v) / max(1, len(v)), len(v)
f, nf = avg("f"); w, nw = avg("w")
alg = ($N * $K + $M * $K + $M * $N) * 2
print("$TAG M=$M N=$N K=$K: FETCH_SIZE %.0f KiB x2 (gfx950) + WRITE_SIZE %.0f KiB = %.3f GB per launch (%d/%d dispatches); algorithmic %.3f GB; ratio %.2f" % (f, w, (2 * f + w) * 1024 / 1e9, nf, nw, alg / 1e9, (2 * f + w) * 1024 / alg))
PY

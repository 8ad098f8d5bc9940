#!/bin/bash
# round-6 measurement calls (run through gpurun): $1 = step name
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
case "$1" in
a)
  timeout -k 10 600 python -m pytest tests/test_closures_gpu.py -x -q -m gpu -k "one_user_fp8_rope or rope_and_kv or graph_replay" > gpurun_out/r06a_t1.log 2>&1 && \
  for v in 1 0; do ATSPEED_FUSE_QKV_ROPE=$v timeout -k 10 300 python tools/batch_run.py 1 bssd 12 none fp8 2>&1 | grep MARK | sed "s/^/rope=$v /" >> gpurun_out/r06a_ab.log || exit 1; done && \
  for v in 1 0; do ATSPEED_FUSE_QKV_ROPE=$v timeout -k 10 300 python tools/batch_run.py 1 bssd 12 3e-6 fp8 2>&1 | grep MARK | sed "s/^/rope=$v /" >> gpurun_out/r06a_ab.log || exit 1; done && \
  for u in "16 tg" "4 bssd" "16 bssd" "4 tg"; do for v in 0 1 2; do ATSPEED_GEMM_KCUT=$v timeout -k 10 300 python tools/batch_run.py $u 6 2>&1 | grep MARK | sed "s/^/kcut=$v /" >> gpurun_out/r06a_ab.log || exit 1; done; done && \
  timeout -k 10 900 python -m pytest tests/test_fp8_gpu.py tests/test_kernels_gpu.py tests/test_bssd_gpu.py -x -q -m gpu -k "fp8 or stream_k or panel or unaligned or two_streams or residual or packed_equals or inference_cli" > gpurun_out/r06a_t2.log 2>&1
  ;;
b)  # (the fragment double-buffering A/B of gemm_wdma_kernel: profiles/r06_wdma_db_ab.txt; the kernel variant lost and is not in the tree,
    #  profiles/r06_wdma_db_experiment.patch has it)
  echo "step b needs the patch profiles/r06_wdma_db_experiment.patch applied"; exit 1
  ;;
suite)  # the full -m gpu suite with its slowest tests: profiles/r06_gpu_suite_durations.txt (tests/test_bench_contract.py holds it under 600 s)
  timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=40 > gpurun_out/r06_gpu_suite.log 2>&1; rc=$?
  grep -A42 "slowest 40" gpurun_out/r06_gpu_suite.log > gpurun_out/r06_gpu_suite_durations.txt; tail -3 gpurun_out/r06_gpu_suite.log >> gpurun_out/r06_gpu_suite_durations.txt
  exit $rc
  ;;
raster)  # one raster experiment on the headline GEMM (gate_up, 26 112 tokens, SwiGLU epilogue): band height of the XCD tile groups.
  # probe libraries: hipcc -DATS_RING_GM=<n> of gemm.hip linked with the product objects (built in the container, tools/probe/libatspeed_gm<n>.so)
  : > gpurun_out/r06_raster.log
  for g in 4 6 2; do
    L=$PWD/tools/probe/libatspeed_gm$g.so; [ $g = 4 ] && L=$PWD/atspeed_amd/lib/libatspeed_hip.so
    echo "== GM=$g ($L)" >> gpurun_out/r06_raster.log
    ATSPEED_LIB=$L GEMM_AB_EPI=3 timeout -k 10 300 python tools/gemm_ab.py 26112 2>&1 | grep gate_up >> gpurun_out/r06_raster.log || exit 1
    ATSPEED_LIB=$L GEMM_AB_EPI=3 timeout -k 10 300 python tools/gemm_ab.py 26112 2>&1 | grep gate_up >> gpurun_out/r06_raster.log || exit 1
    ATSPEED_LIB=$L bash tools/pmc_traffic_gemm.sh gm$g 26112 22016 4096 3 >> gpurun_out/r06_raster.log 2>&1 || exit 1
  done
  ;;
msplit)  # (A/B of the M-split form of gemm_wdma_kernel at 129-256 tokens: profiles/r06_wdma_msplit_ab.txt; no gain, not in the tree,
         #  profiles/r06_wdma_msplit_experiment.patch has it)
  echo "step msplit needs profiles/r06_wdma_msplit_experiment.patch applied"; exit 1
  ;;
d)  # the tiled kernel's qkv slabs summed by RoPE (<= 32 tokens, draft): equality tests, A/B on one user per call
  timeout -k 10 600 python -m pytest tests/test_closures_gpu.py tests/test_bssd_gpu.py tests/test_from_hf_gpu.py -x -q -m gpu -k "qkv_slabs or lossless or from_hf" > gpurun_out/r06d_t1.log 2>&1 && \
  : > gpurun_out/r06d_ab.log && \
  for rs in none 3e-6; do for v in 1 0 1 0; do ATSPEED_FUSE_QKV_REDUCE=$v timeout -k 10 300 python tools/batch_run.py 1 bssd 12 $rs 2>&1 | grep MARK | sed "s/^/fuse_qkv_reduce=$v /" >> gpurun_out/r06d_ab.log || exit 1; done; done
  ;;
profile)  # the round's profiles/ artefacts: default bench line + kernel stats + PMC traffic (tools/profile_round.sh), then the one-user W8A8 trace
  ATSPEED_COMMIT=${ATSPEED_COMMIT:-unknown} bash tools/profile_round.sh > gpurun_out/r06_profile_round.log 2>&1 || { tail -20 gpurun_out/r06_profile_round.log; exit 1; }
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  for cfg in "fp8:none fp8" "bf16:none"; do
    tag=${cfg%%:*}; args=${cfg#*:}
    rm -rf gpurun_out/bt
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/bt -o bt --output-format csv -- python3 tools/batch_run.py 1 bssd 6 $args > gpurun_out/r06_band_bssd_u1_${tag}.log 2>&1 || { tail -5 gpurun_out/r06_band_bssd_u1_${tag}.log; exit 1; }
    python tools/trace_gaps.py $(find gpurun_out/bt -name "*kernel_trace.csv") 0.6 > gpurun_out/r06_band_bssd_u1_${tag}_gaps.txt
    cp $(find gpurun_out/bt -name "*kernel_stats.csv") gpurun_out/r06_band_bssd_u1_${tag}_kernel_stats.csv
    grep MARK gpurun_out/r06_band_bssd_u1_${tag}.log
  done
  rm -rf gpurun_out/bt
  ;;
fulldims)  # the K = 20 full-dims test with its prints (fp64 arbiter numbers)
  timeout -k 10 600 python -m pytest tests/test_fulldims_gpu.py -x -q -s -m gpu -k "equals_oracle" > gpurun_out/r06_fulldims_k20.log 2>&1; rc=$?
  grep -n "fp64\|near ties\|checked exactly\|user " gpurun_out/r06_fulldims_k20.log | tail -30; exit $rc
  ;;
pmc8)  # counters of the W8A8 weight-streaming kernel at 228 tokens (gate_up, down): matrix pipe busy, waits, LDS, HBM-side bytes -> profiles/r06_pmc_wdma8.json
  bash tools/pmc_wdma8.sh gate_up_228 228 22016 4096 3 > gpurun_out/r06_pmc8_gate_up.log 2>&1 || { tail -5 gpurun_out/r06_pmc8_gate_up.log; exit 1; }
  bash tools/pmc_wdma8.sh down_228 228 4096 11008 2 > gpurun_out/r06_pmc8_down.log 2>&1 || { tail -5 gpurun_out/r06_pmc8_down.log; exit 1; }
  bash tools/pmc_wdma8.sh gate_up_32 20 22016 4096 3 > gpurun_out/r06_pmc8_gate_up20.log 2>&1 || { tail -5 gpurun_out/r06_pmc8_gate_up20.log; exit 1; }
  python3 - <<'PY'
import json
out = {"_what": "rocprofv3 --pmc passes (tools/pmc_wdma8.sh, round 6: + SQ_VALU_MFMA_BUSY_CYCLES / SQ_WAIT_INST_LDS / SQ_WAIT_INST_ANY) of the W8A8 weight-streaming kernel on engine-like launches (packed e4m3 operands, own epilogue, six weight copies in rotation); per-launch averages, FETCH doubled per the gfx950 note"}
for tag in ("gate_up_228", "down_228", "gate_up_32"):
    out[tag] = json.load(open(f"gpurun_out/pmc_{tag}.json"))
json.dump(out, open("gpurun_out/r06_pmc_wdma8.json", "w"), indent=1)
for tag, d in out.items():
    if tag.startswith("_"): continue
    for k, v in d.items():
        if isinstance(v, dict) and "counters" in v:
            print(tag, k, {x: (round(y, 4) if isinstance(y, float) else y) for x, y in v.items() if x != "counters"})
PY
  ;;
esac

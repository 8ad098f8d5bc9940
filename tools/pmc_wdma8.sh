#!/bin/bash
# Counters of the W8A8 weight-streaming kernel (gemm_wdma_kernel<..., F8>) on one shape: LDS bank conflicts (the chunk g / 4 + g fragment map),
# HBM-side bytes per launch (FETCH_SIZE, WRITE_SIZE: separate passes, no trace domains beside --pmc).  usage: tools/pmc_wdma8.sh TAG M N K EPI
set -e
TAG=$1; M=$2; N=$3; K=$4; EPI=${5:-0}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
# (round 6: + the wait / matrix-pipe pass VERDICT r5 asked for -- SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAIT_INST_LDS, SQ_WAIT_INST_ANY)
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d" " -f1)
  rocprofv3 --pmc $pass -d $OUT/$tag -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/one_gemm_fp8.py $M $N $K $EPI > $OUT.$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - "$OUT" "$M" "$N" "$K" "$EPI" <<'PY'
import collections, csv, glob, json, os, re, sys
root, m, n, k, epi = sys.argv[1], *map(int, sys.argv[2:6])
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")) + glob.glob(os.path.join(root, "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "gemm_wdma_kernel" not in kn and "splitk_" not in kn: continue
        mm = re.search(r"(gemm_wdma_kernel<[^>]*>|splitk_\w+<[^>]*>)", kn)
        key = mm.group(1) if mm else kn[:80]
        a = agg[key][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {"_method": "rocprofv3 --pmc, separate passes (tools/pmc_wdma8.sh), per-launch averages; FETCH_SIZE doubled per the gfx950 note (128-B requests counted at 64 B)",
       "shape": dict(m=m, n=n, k=k, epilogue=epi), "algorithmic_weight_bytes": n * k, "algorithmic_x_bytes": m * k}
for kern, cs in agg.items():
    c = {name: v[1] / v[0] for name, v in cs.items()}
    d = {"counters": c}
    if "FETCH_SIZE" in c: d["hbm_read_bytes_per_launch"] = 2 * c["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in c: d["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
    if "SQ_LDS_BANK_CONFLICT" in c: d["lds_bank_conflict_cycles"] = c["SQ_LDS_BANK_CONFLICT"]; d["lds_active_cycles"] = c.get("SQ_LDS_IDX_ACTIVE")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("SQ_BUSY_CYCLES"):
        # SQ_BUSY_CYCLES sums the busy cycles of the shader engines' sequencers; the matrix pipes' busy cycles against the wave-SIMD time the launch occupied
        # (SQ_WAVE_CYCLES counts 4-cycle quanta per wave; waves_per_simd of them share one SIMD's pipe)
        wps = 2 if "4, true>" in kern or ", 4, false>" in kern else 1
        d["mfma_pipe_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["SQ_WAVE_CYCLES"] * 4 / wps) if c.get("SQ_WAVE_CYCLES") else None
        d["wave_time_waiting_fraction"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None
        d["wave_time_waiting_on_lds_fraction"] = c.get("SQ_WAIT_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None
    out[kern] = d
json.dump(out, open(root + ".json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT

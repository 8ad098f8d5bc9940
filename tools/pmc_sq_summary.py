#!/usr/bin/env python3
"""Summarise the two SQ-counter passes of tools/pmc_gemm.sh into one JSON (per-launch averages of the ring GEMM kernel).
usage: pmc_sq_summary.py gpurun_out/pmc_TAG "description" > profiles/xxx.json"""
import collections, csv, glob, json, os, sys
root, what = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0]); kern = None
for f in glob.glob(os.path.join(root, "p*", "*", "*_counter_collection.csv")) + glob.glob(os.path.join(root, "p*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "gemm_ring" not in r["Kernel_Name"]:
            continue
        kern = r["Kernel_Name"][:120]
        a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
c = {k: v[1] / v[0] for k, v in agg.items()}
out = {"_method": "rocprofv3 --pmc, two separate passes (tools/pmc_gemm.sh), per-launch averages of the ring GEMM: " + what +
       ".  SQ_WAVE_CYCLES counts 4-cycle quanta per wave; two waves share a SIMD, so MFMA pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_WAVE_CYCLES * 4 / 2).",
       "kernel": kern, "counters": c}
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_WAVE_CYCLES" in c:
    out["mfma_pipe_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["SQ_WAVE_CYCLES"] * 4 / 2)
if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
    out["wave_time_parked_in_waitcnt_or_barrier"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
if "SQ_LDS_BANK_CONFLICT" in c:
    out["lds_bank_conflict_cycles"] = c["SQ_LDS_BANK_CONFLICT"]
print(json.dumps(out, indent=1))

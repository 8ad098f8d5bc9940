#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json.

HBM-side bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and, on gfx950,
FETCH_SIZE counts 128-B fabric read requests of wide coalesced streams (global_load_dwordx4 and LDS-DMA alike) at 64 B,
i.e. exactly half the bytes (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B-per-lane stores."""
import collections, csv, glob, json, os, subprocess, sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/final"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/pmc_traffic.json"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
KINDS = {"gate_up": "gemm_ring_kernel<3, 8, false, false", "qkv": "gemm_ring_kernel<0, 8, false, false", "qkv+rope": "gemm_ring_kernel<5, 8, false, false", "o_proj+down": "gemm_ring_kernel<2, 8, false, false", "lm_head+lse": "gemm_ring_kernel<4, 8, false, false",
         "lm_head": "gemm_ring_kernel<1, 8, false, false", "gate_up_128": "gemm_ring_kernel<3, 4, false, false", "attention": "tree_attn32_kernel<128, 4>",
         "lse": "lse_rows_kernel"}

def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(path, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg

fetch, write = per_kernel(os.path.join(root, "pmc_fetch")), per_kernel(os.path.join(root, "pmc_write"))
res = {"_method": __doc__.strip()}
# provenance: bench.py quotes a traffic figure only when it was recorded for the GEMM sources of the tree it runs in (kernel_sha)
import bench
res["kernel_sha"] = bench.kernel_sha()
res["commit"] = os.environ.get("ATSPEED_COMMIT") or "unknown"      # the GPU box has no .git: tools/profile_round.sh is given the commit
res["workload"] = os.environ.get("ATSPEED_PMC_WORKLOAD", "python bench.py --steps 2 --warmup 0 (Beauty, 256 users per lock-step batch, bf16)")
# average M of each GEMM kind over the same workload's launches, from the bench line of the same script run (hipEvent brackets, rows / count)
avg_m = {}
try:
    line = json.loads(open(os.path.join(root, "bench_default.json")).read().strip().splitlines()[-1])
    k = line["roofline"]["kernel"]
    avg_m["gate_up"] = (line["roofline"].get("avg_m") or (float(k.split("avg_M=")[1].split()[0]) if "avg_M=" in k else None)) if ("<3, 8" in k or "gate_up" in k) else None
except Exception:
    pass
for kind, pat in KINDS.items():
    fk = [k for k in fetch if pat in k]
    wk = [k for k in write if pat in k]
    if not fk or not wk:
        continue
    # every instantiation the pattern covers (round 4: the ring kernel exists without and with its split-K tail, `..., 4, false>` / `..., 4, true>`,
    # chosen per launch), dispatch-weighted
    fc, fv = sum(fetch[k][0] for k in fk), sum(fetch[k][1] for k in fk)
    wc, wv = sum(write[k][0] for k in wk), sum(write[k][1] for k in wk)
    res[kind] = {"kernel": pat, "instantiations": sorted(k.split("(")[0][-60:] for k in fk), "dispatches": fc, "FETCH_SIZE_KiB_per_launch": fv / fc, "WRITE_SIZE_KiB_per_launch": wv / wc,
                 "hbm_bytes_per_launch": 2 * fv / fc * 1024 + wv / wc * 1024, "avg_m": avg_m.get(kind)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "_method"}, indent=1))

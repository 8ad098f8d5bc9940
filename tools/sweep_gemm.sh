# NOTE (round 6): ATSPEED_GEMM_TARGET_WGS / ATSPEED_GEMM_BN are constants in the product since round 6; this round-1 sweep runs as written from commit 60a2317.
for t in 192 512 768 1024; do echo "== target_wgs=$t"; ATSPEED_GEMM_TARGET_WGS=$t timeout -k 10 120 python tools/gemm_bench.py 20,100,228 2>/dev/null | grep -v amdgpu; done
echo "== BN=64 target 512"; ATSPEED_GEMM_BN=64 ATSPEED_GEMM_TARGET_WGS=512 timeout -k 10 120 python tools/gemm_bench.py 100,228 2>/dev/null | grep -v amdgpu

echo "== big kernel (M>=768)"; timeout -k 10 200 python tools/gemm_bench.py 1024,2048,3648 2>/dev/null | grep -v amdgpu
echo "== LDS-tiled kernel only"; ATSPEED_GEMM_BIG_MIN_M=1000000 timeout -k 10 200 python tools/gemm_bench.py 1024,2048,3648 2>/dev/null | grep -v amdgpu

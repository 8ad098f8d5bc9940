timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k gemm 2>&1 | tail -2
for d in 0 3; do echo "== DBG=$d (0 = two tiles in flight, 3 = simple loop)"; ATSPEED_GEMM_BIG_DBG=$d timeout -k 10 200 python tools/gemm_bench.py 1024,3648,7296 2>/dev/null | grep -v amdgpu; done

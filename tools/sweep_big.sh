timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k gemm 2>&1 | tail -2
timeout -k 10 200 python tools/gemm_bench.py 640,1280,1920,3200 2>/dev/null | grep -v amdgpu

mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_bssd_gpu.py -m gpu -q -x -k "games or batch" > gpurun_out/gpu_tests_16.log 2>&1; tail -5 gpurun_out/gpu_tests_16.log
# N=1 through the distributed launcher: exercises the RCCL init + all_gather path of bench.py
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 32 --warmup 2 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; tail -2 gpurun_out/bench_torchrun1.err | cut -c1-200; python -c "
import json; d=json.loads(open('gpurun_out/bench_torchrun1.json').read().strip().splitlines()[-1]); print('torchrun n=1:', round(d['value'],1), d['n_gpus'], d['config']['parallelism'])"

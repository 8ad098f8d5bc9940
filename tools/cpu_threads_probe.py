import torch, time, os
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("no cpu.max", e)
a=torch.randn(228,4096); w=torch.randn(11008,4096)
for n in (128, 64, 32, 16, 8):
    torch.set_num_threads(n)
    for _ in range(2): a@w.T
    t=time.perf_counter()
    for _ in range(10): a@w.T
    print(n, "threads:", round((time.perf_counter()-t)/10*1e3,2), "ms per 228x4096x11008 matmul")

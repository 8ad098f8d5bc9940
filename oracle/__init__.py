"""ORACLE package (test infrastructure, not product code): CPU restatements of the reference's beam-SD path.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import it."""


def use_allotted_cpu_threads() -> int:
    """torch's intra-op thread count := the CPUs this process is really allotted (cgroup v2 `cpu.max` quota, affinity mask), and return it.
    torch sizes its pool from the machine (128 threads on the GPU box's 256-CPU host) while a one-GPU box's share is 16 CPUs: the oversubscribed
    fp32 matmuls of the oracle ran 3 x slower (228 x 4096 x 11008: 49.8 ms at 128 threads, 16.8 ms at 16; tools/cpu_threads_probe.py)."""
    import os
    import torch
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    n = max(1, min(n, torch.get_num_threads()))
    torch.set_num_threads(n)
    return n

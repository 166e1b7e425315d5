#!/usr/bin/env python3
"""What the host side of a GPU box really offers: affinity, cgroup CPU quota, and PyTorch-CPU conv throughput versus
thread count (output kept as profiles/r02_host_cpu_probe.txt)."""
import os, time, torch, torch.nn.functional as F
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print(open("/proc/self/status").read().split("Cpus_allowed_list:")[1].split("\n")[0])
x = torch.randn(2, 256, 52, 52); w = torch.randn(512, 256, 3, 3)
for n in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(n)
    F.conv2d(x, w, padding=1)
    t = time.time()
    for _ in range(5): F.conv2d(x, w, padding=1)
    dt = (time.time() - t) / 5
    print(f"threads {n:4d}: {dt*1e3:8.2f} ms  {2*2*512*256*9*52*52/dt/1e9:8.1f} GFLOP/s", flush=True)

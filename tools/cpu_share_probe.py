#!/usr/bin/env python3
"""How many host CPUs does this box really give us?  (nproc / affinity / cgroup quota, and a scaling measurement.)"""
import os, time, threading
import numpy as np
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(p, open(p).read().strip())
    except OSError:
        pass
print("loadavg", open("/proc/loadavg").read().strip())


def work(out, i):
    a = np.random.rand(300, 300)
    t = time.time(); n = 0
    while time.time() - t < 2.0:
        a @ a; n += 1
    out[i] = n


for nt in (1, 8, 16, 32, 64, 128):
    out = [0] * nt
    th = [threading.Thread(target=work, args=(out, i)) for i in range(nt)]
    [t.start() for t in th]; [t.join() for t in th]
    print(nt, "threads: matmuls in 2 s", sum(out), "per thread", sum(out) // nt, flush=True)

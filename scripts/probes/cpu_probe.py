import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, "n/a")
print("loadavg", open("/proc/loadavg").read().strip())
x = torch.randn(1000000, 768); q = torch.randn(256, 768)
for th in (256, 128, 64, 32, 16, 8):
    torch.set_num_threads(th)
    for block in (16384, 65536, 262144):
        best = 1e9
        for rep in range(2):
            t = time.perf_counter()
            for c0 in range(0, 1000000, block):
                s = q @ x[c0:c0 + block].T
                v, i = torch.topk(s, 10, dim=1)
            best = min(best, time.perf_counter() - t)
        print(th, block, round(best, 3), flush=True)

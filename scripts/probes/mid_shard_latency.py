import os, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from vietnamese_qa_system_amd.index import DeviceIndex
rng = np.random.default_rng(0)
d = 768
for n in (131072, 200000, 262144, 524288, 1000000):
    x = torch.randn((n, d), device="cuda", dtype=torch.float16)
    q = rng.standard_normal((1, d)).astype(np.float32)
    ix = DeviceIndex(x, id_base=1, dtype="fp16", device=0)
    for k in (1, 10):
        for _ in range(10): ix.search_host(q, k, normalize=True)
        ts = []
        for _ in range(100):
            t0 = time.perf_counter(); ix.search_host(q, k, normalize=True); ts.append(time.perf_counter() - t0)
        print(f"n={n} k={k}: {np.median(ts)*1e6:.1f} us  ({n*d*2/np.median(ts)/1e12:.2f} TB/s of rows)", flush=True)
    ix.close(); del x

"""Dev probe: a small fp32 index (BASELINE configs[0]'s shape) -- device step of the exact f32 MFMA scan against the number of live
queries (the f32 loop multiplies only the live query column groups), fp16 beside it."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vietnamese_qa_system_amd.index import DeviceIndex
rng = np.random.default_rng(0)
for n in (1000, 50000):
    x = rng.standard_normal((n, 768)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = torch.from_numpy(rng.standard_normal((256, 768)).astype(np.float32)).cuda()
    for dt in ("fp32", "fp16"):
        ix = DeviceIndex(x, dtype=dt, device=0)
        out = []
        for b in (1, 16, 17, 64, 65, 256):
            for _ in range(5): ix.search(q[:b], 10)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ix.search(q[:b], 10)
            e1.record(); torch.cuda.synchronize()
            out.append(f"B={b}: {e0.elapsed_time(e1) / 20:.3f} ms")
        print(f"{n} x 768 {dt}: " + "  ".join(out), flush=True)
        ix.close()

"""Dev probe: is the one-question forward's ~5 us per launch a clock effect?  Times the B = 1, L = 32 forward back to back in
runs of 30 / 300 / 3000 calls, and right behind a heavy kernel (a large GEMM keeps the chip's clock up)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import encoder as E
from vietnamese_qa_system_amd.encoder import QuestionEncoder
cfg = dict(E.PHOBERT_BASE)
w = E.synthetic_weights(cfg, seed=0)
ids, mask = E.synthetic_tokens(cfg, 1, 32, seed=1)
enc = QuestionEncoder(w, cfg, max_tokens=64)
ids_t, mask_t = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
for _ in range(5): enc.forward(ids_t, mask_t)
torch.cuda.synchronize()
for n in (30, 300, 3000):
    t0 = time.perf_counter()
    for _ in range(n): enc.forward(ids_t, mask_t)
    torch.cuda.synchronize()
    print(f"{n:5d} forwards back to back: {(time.perf_counter() - t0) / n * 1e3:.4f} ms each", flush=True)
a = torch.randn((8192, 8192), device="cuda", dtype=torch.float16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(20):
    for _ in range(3): torch.mm(a, a)
    e0.record(); enc.forward(ids_t, mask_t); e1.record()
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(f"one forward right behind three 8192^3 GEMMs: median {np.median(ts):.4f} ms")
time.sleep(0.5)
ts = []
for _ in range(20):
    time.sleep(0.02)
    e0.record(); enc.forward(ids_t, mask_t); e1.record()
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(f"one forward on an idle chip (20 ms pauses): median {np.median(ts):.4f} ms")

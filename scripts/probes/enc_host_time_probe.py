import os, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from vietnamese_qa_system_amd.encoder import MINILM_L12, XLMR_BASE
device = torch.device("cuda", 0)
for name, cfg in (("minilm", MINILM_L12), ("xlmr", XLMR_BASE)):
    enc, ids, mask, lens, g = bench.make_encoder(torch, device, 0, 1, 32, max_tokens=64, cfg=cfg)
    mask = torch.ones_like(mask)
    for _ in range(10): enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=0)
    torch.cuda.synchronize()
    th, tt = [], []
    for _ in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=0)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        th.append(t1 - t0); tt.append(t2 - t0)
    print(f"{name}: host time of the call {np.median(th)*1e3:.3f} ms, until the device is done {np.median(tt)*1e3:.3f} ms")
    ids_h, mask_h = ids.cpu().numpy(), mask.cpu().numpy()
    out = torch.empty((1, cfg["hidden"]), dtype=torch.float32, device=device)
    th, tt = [], []
    for _ in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enc.forward_host(ids_h, mask_h, out, pooling="mean", normalize=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        th.append(t1 - t0); tt.append(t2 - t0)
    print(f"{name}: forward_host host time {np.median(th)*1e3:.3f} ms, until the device is done {np.median(tt)*1e3:.3f} ms")

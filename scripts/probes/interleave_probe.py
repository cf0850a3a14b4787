"""Dev probe: why does the packed encoder forward take 2.2 ms inside bench.py's end-to-end sequence and 1.65 ms back to back?
Times the forward (event pairs) when each call follows (a) another forward, (b) the 10M-row fp16 search, (c) a plain read of the
same 15 GB (no MFMA), (d) an MFMA-heavy GEMM loop of similar length with no HBM stream."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import encoder as E
from vietnamese_qa_system_amd.encoder import QuestionEncoder
from vietnamese_qa_system_amd.index import DeviceIndex
dev = torch.device("cuda", 0)
cfg = dict(E.PHOBERT_BASE)
w = E.synthetic_weights(cfg, seed=0)
ids, mask = E.synthetic_tokens(cfg, 256, 32, seed=1)
enc = QuestionEncoder(w, cfg, max_tokens=256 * 32)
ids_t, mask_t = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
real = int(mask.sum())
g = torch.Generator(device=dev); g.manual_seed(1)
n = 10_000_000
buf = torch.empty((n, 768), dtype=torch.float16, device=dev)
for c0 in range(0, n, 1 << 18):
    x = torch.randn((min(n, c0 + (1 << 18)) - c0, 768), generator=g, device=dev)
    buf[c0:c0 + x.shape[0]] = (x / x.norm(dim=1, keepdim=True)).half()
ix = DeviceIndex(buf, dtype="fp16")
q = torch.randn((256, 768), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).half()
a = torch.randn((8192, 8192), generator=g, device=dev).half(); b = torch.randn((8192, 8192), generator=g, device=dev).half()
flat = buf.view(-1)
between = {"forward only": lambda: None, "after the 10M-row search": lambda: ix.search(q, 10),
           "after a plain read of the 15 GB (max over the shard)": lambda: flat.max(),
           "after 3 x hipBLASLt 8192^3 (no HBM stream, ~2.5 ms of MFMA)": lambda: [torch.mm(a, b) for _ in range(3)]}
for name, fn in between.items():
    for _ in range(3):
        fn(); enc.forward(ids_t, mask_t, real_tokens=real)
    ev = []
    for _ in range(20):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); enc.forward(ids_t, mask_t, real_tokens=real); e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    t = np.array([e0.elapsed_time(e1) for e0, e1 in ev])
    print(f"{name:62s}: forward median {np.median(t):.3f} ms (p10 {np.percentile(t, 10):.3f}, p90 {np.percentile(t, 90):.3f})", flush=True)

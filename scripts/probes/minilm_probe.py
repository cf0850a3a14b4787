"""Dev probe: the reference's OTHER encoder (heavy_ranker.py:80: paraphrase-multilingual-MiniLM-L12-v2 -- hidden 384, 12 heads of 32,
FFN 1536) by its shape, random weights: 256 ragged questions packed, and one 32-token question; wall / event times.  Under
`rocprofv3 --kernel-trace --stats` the kernel table says which launches the time goes to."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vietnamese_qa_system_amd.encoder import MINILM_L12, XLMR_BASE
device = torch.device("cuda", 0)
which = os.environ.get("MODEL", "minilm")
cfg = MINILM_L12 if which == "minilm" else XLMR_BASE
for b, L in ((256, 32), (1, 32)):
    enc, ids, mask, lens, g = bench.make_encoder(torch, device, 0, b, L, max_tokens=256 * 32, cfg=cfg)
    real = int(lens.sum())
    for _ in range(5):
        enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=real)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for e0, e1 in ev:
        e0.record(); enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=real); e1.record()
    torch.cuda.synchronize()
    print(f"{which} B={b} L={L} real tokens {real}: forward {np.median([a.elapsed_time(c) for a, c in ev]):.4f} ms", flush=True)
    enc.close()

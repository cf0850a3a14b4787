"""Dev helper: time the encoder forward (PhoBERT-base shape, random weights)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import encoder as E
from vietnamese_qa_system_amd.encoder import QuestionEncoder
b, l = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = dict(E.PHOBERT_BASE)
w = E.synthetic_weights(cfg, seed=0)
ids, mask = E.synthetic_tokens(cfg, b, l, seed=1)
enc = QuestionEncoder(w, cfg, max_tokens=b * l)
ids_t, mask_t = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
for _ in range(3): enc.forward(ids_t, mask_t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): enc.forward(ids_t, mask_t)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
flops = b * l * 12 * (2 * (768 * 2304 + 768 * 768 + 2 * 768 * 3072) + 4 * l * 768)
print(f"encoder B={b} L={l}: {ms:.3f} ms, {flops / ms / 1e9:.0f} TFLOP/s")

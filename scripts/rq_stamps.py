"""Dev helper: cycle timeline of scan_regq.hip's rolling form from a stamped diagnostic build
(`build_variants.py --only=scan_regq.hip rqst:VQA_RQ_STAMPS=1,VQA_RQ_ABLATE=8`, VQA_LIB selects it): for each wave of one workgroup during
one half tile (12 steps = 3 barrier groups), cycles per step, cycles the wave spent issuing its DMA piece, and the group-end waits."""
import argparse, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vietnamese_qa_system_amd.index import DeviceIndex
from vietnamese_qa_system_amd import _native as N

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--d", type=int, default=768)
ap.add_argument("--b", type=int, default=256)
args = ap.parse_args()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
buf = torch.empty((args.n, args.d), dtype=torch.float16, device=dev)
for c0 in range(0, args.n, 1 << 18):
    c1 = min(args.n, c0 + (1 << 18))
    x = torch.randn((c1 - c0, args.d), generator=gen, device=dev)
    x /= x.norm(dim=1, keepdim=True)
    buf[c0:c1] = x.to(buf.dtype)
ix = DeviceIndex(buf, dtype="fp16")
q = torch.randn((args.b, args.d), generator=gen, device=dev)
q = (q / q.norm(dim=1, keepdim=True)).to(buf.dtype)
for _ in range(12):
    ix.search(q, 10)
torch.cuda.synchronize()
lib = N.load()
out = np.zeros((8, 3, 16), dtype=np.uint64)
lib.vqa_debug_read_rq_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.vqa_debug_read_rq_stamps(out.ctypes.data, out.size) == 0
t = out.astype(np.int64)
t0 = t[:, 0, 0][t[:, 0, 0] > 0].min()
print(f"B = {args.b}; one half tile of workgroup 5 (12 steps); cycles (s_memtime)")
print("wave | per step: start->next start (12) | DMA issue (12) | per group: last step start->waits, waits, barrier")
for w in range(8):
    if t[w, 0, 0] == 0:
        print(f"{w:4d} | no live queries")
        continue
    starts = np.array([t[w, g, 3 * j] for g in range(3) for j in range(4)])
    dma = np.array([t[w, g, 3 * j + 2] - t[w, g, 3 * j + 1] for g in range(3) for j in range(4)])
    steps = np.diff(np.append(starts, t[w, 2, 14]))
    grp = [(t[w, g, 12] - t[w, g, 9], t[w, g, 13] - t[w, g, 12], t[w, g, 14] - t[w, g, 13]) for g in range(3)]
    print(f"{w:4d} | " + " ".join(f"{x:4d}" for x in steps) + " | " + " ".join(f"{x:4d}" for x in dma) + " | " + "  ".join(f"{a}/{b}/{c}" for a, b, c in grp)
          + f" | half tile {t[w, 2, 14] - t[w, 0, 0]} cycles, start at {t[w, 0, 0] - t0}")

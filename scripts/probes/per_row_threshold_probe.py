"""Dev probe: from which centre norm on does the per-row form of the sketch bound beat the centre split?  3M x 768 fp16 rows sharing one
common component of weight w (mean cosine w^2 / (1 + w^2)), queries drawn alike; per weight: the split form (VQA_SKETCH_PER_ROW=0) and the
per-row form (=1): candidate pairs, step time; the exact scan beside them."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vietnamese_qa_system_amd.index import DeviceIndex
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
n, d, b, k = 3_000_000, 768, 256, 10
c = torch.randn((1, d), generator=g, device=dev); c /= c.norm()
def draw(m, w):
    v = torch.randn((m, d), generator=g, device=dev)
    v = w * c + v / v.norm(dim=1, keepdim=True)
    return (v / v.norm(dim=1, keepdim=True)).half()
def timed(ix, q):
    for _ in range(3): ix.search(q, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(15): ix.search(q, k)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 15 * 1e3
for w in (1.0, 1.5, 2.0, 2.5, 3.0, 4.0):
    x = torch.cat([draw(1 << 19, w) for _ in range(0, n, 1 << 19)])[:n]
    q = draw(b, w)
    out = []
    for form in ("0", "1"):
        ix = DeviceIndex(x, dtype="fp16", sketch=True, options={"sketch_per_row": int(form), "sketch_profit": 0.0})
        ix.search(q, k); torch.cuda.synchronize()
        st = ix.sketch_stats()
        out.append(f"{'per-row' if ix.sketch_split(0)[3] else 'split'}: pairs {st['rescored_pairs']:8d} overflow {st['overflow']} step {timed(ix, q):.3f} ms")
        ix.close()
    ref = DeviceIndex(x, dtype="fp16", sketch=False)
    print(f"weight {w} (mean cosine {w * w / (1 + w * w):.2f}): " + " | ".join(out) + f" | exact {timed(ref, q):.3f} ms", flush=True)
    ref.close(); del x

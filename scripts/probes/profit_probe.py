import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vietnamese_qa_system_amd.index import DeviceIndex
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
def timed(ix, q, k, n=20):
    for _ in range(3): ix.search(q, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ix.search(q, k)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for n, d in ((1_200_000, 768), (1_200_000, 1024), (3_000_000, 768), (3_000_000, 1024), (1_200_000, 384)):
    x = torch.randn((n, d), generator=g, device=dev); x = (x / x.norm(dim=1, keepdim=True)).half()
    ske = DeviceIndex(x, dtype="fp16", sketch=True, options={"sketch_profit": 0.0})
    ref = DeviceIndex(x, dtype="fp16", sketch=False)
    for b in (256, 1):
        q = torch.randn((b, d), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).half()
        for k in (10, 13, 32, 64, 100, 128):
            ske.search(q, k); torch.cuda.synchronize(); st = ske.sketch_stats()
            ts, te = timed(ske, q, k), timed(ref, q, k)
            print(f"n={n} d={d} b={b} k={k}: sketch {ts:.3f} ms exact {te:.3f} ms ratio {te/ts:.2f}  pairs x256/b / n = {st['rescored_pairs'] * 256 / b / n:.2f} state {ske.sketch_state()}", flush=True)
    ske.close(); ref.close(); del x

import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vietnamese_qa_system_amd.index import DeviceIndex
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
n, d = 10_000_000, 768
x = torch.empty((n, d), dtype=torch.float16, device=dev)
for c0 in range(0, n, 1 << 19):
    r = torch.randn((min(n, c0 + (1 << 19)) - c0, d), generator=g, device=dev)
    x[c0:c0 + r.shape[0]] = (r / r.norm(dim=1, keepdim=True)).half()
for env in ({"sketch": 0}, {"sketch_rotate": 0}, {}):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix = DeviceIndex(x, dtype="fp16", options=env)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(env or "default (sketch, rotated, centred, copy)", f"build {t:.2f} s, device bytes {ix.device_bytes() / 1e9:.1f} GB", flush=True)
    ix.close()

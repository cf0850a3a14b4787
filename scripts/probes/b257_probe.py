import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vietnamese_qa_system_amd.index import DeviceIndex
from vietnamese_qa_system_amd import _native as N
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
n, d = 3_000_000, 768
x = torch.randn((n, d), generator=g, device=dev); x = (x / x.norm(dim=1, keepdim=True)).half()
ix = DeviceIndex(x, dtype="fp16")
lib = N.load()
for b in (256, 1, 257, 2, 300):
    q = torch.randn((b, d), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).half()
    ix.search(q, 10); torch.cuda.synchronize()
    st = ix.sketch_stats()
    print("B", b, "pairs", st["last_scan_pairs"], "max region", st["largest_region"], "max sublist", st["longest_sublist"], "overflow", st["overflow"], flush=True)

import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["VQA_LIB"] = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "vietnamese_qa_system_amd/lib/libvqa_retrieval_dev.so")
from vietnamese_qa_system_amd.index import DeviceIndex
from vietnamese_qa_system_amd import _native as N
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
n, d = 3_000_000, 768
x = torch.randn((n, d), generator=g, device=dev); x = (x / x.norm(dim=1, keepdim=True)).half()
ix = DeviceIndex(x, dtype="fp16")
lib = N.load()
lib.vqa_dev_sketch_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
for b in (256, 1, 257, 2, 300):
    q = torch.randn((b, d), generator=g, device=dev); q = (q / q.norm(dim=1, keepdim=True)).half()
    ix.search(q, 10); torch.cuda.synchronize()
    out = (ctypes.c_longlong * 4)()
    lib.vqa_dev_sketch_stats(ix._handle, out)
    print("B", b, "pairs", out[0], "max region", out[1], "max sublist", out[2], "overflow", out[3], flush=True)

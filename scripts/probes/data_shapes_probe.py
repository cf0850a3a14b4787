"""Dev probe: how does the sketch search's pruning hold up on data that is not isotropic?  3M x 768 fp16 rows of several shapes,
256 queries drawn like the rows; per shape: candidate pairs of the main scan (vqa_index_sketch_stats), whether the
search stayed on the sketch, step time against the exact scan."""
import ctypes, os, sys, time
import torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from vietnamese_qa_system_amd.index import DeviceIndex
from vietnamese_qa_system_amd import _native as N
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(11)
n, d, b, k = 3_000_000, 768, 256, 10
lib = N.load()


def unit(x):
    return x / x.norm(dim=1, keepdim=True)


_shared = {}


def shared(name, fn):  # structure the rows and the queries of a shape share (cluster centres, the low-rank basis)
    if name not in _shared:
        _shared[name] = fn()
    return _shared[name]


def make(kind, m):
    x = torch.randn((m, d), generator=g, device=dev)
    if kind == "isotropic":
        pass
    elif kind == "two outlier dimensions (20x)":
        x[:, 77] *= 20; x[:, 588] *= 20
    elif kind == "two constant outlier dimensions (mean cosine 0.5)":
        x = unit(x)
        x[:, 77] += 0.7; x[:, 588] += 0.7
    elif kind == "common component (mean cosine 0.5)":
        x = unit(x) + torch.ones(d, device=dev) / d ** 0.5
    elif kind.startswith("common component (mean cosine 0.9)"):
        x = unit(x) + 3.0 * torch.ones(d, device=dev) / d ** 0.5
    elif kind == "1000 clusters, within-cluster sigma 0.3":
        c = shared(kind, lambda: unit(torch.randn((1000, d), generator=g, device=dev)))
        x = c[torch.randint(0, 1000, (m,), generator=g, device=dev)] + 0.3 * x / d ** 0.5
    elif kind == "50 tight clusters, sigma 0.05":
        c = shared(kind, lambda: unit(torch.randn((50, d), generator=g, device=dev)))
        x = c[torch.randint(0, 50, (m,), generator=g, device=dev)] + 0.05 * x / d ** 0.5
    elif kind == "low rank 32 + 10 % noise":
        p = shared(kind, lambda: torch.randn((32, d), generator=g, device=dev))
        x = torch.randn((m, 32), generator=g, device=dev) @ p + 0.1 * x
    return unit(x).half()


def timed(ix, q):
    for _ in range(3):
        ix.search(q, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        ix.search(q, k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3


for kind in ("isotropic", "two outlier dimensions (20x)", "two constant outlier dimensions (mean cosine 0.5)", "common component (mean cosine 0.5)",
             "common component (mean cosine 0.9)", "common component (mean cosine 0.9), per-row form off", "common component (mean cosine 0.9), split off too",
             "1000 clusters, within-cluster sigma 0.3", "50 tight clusters, sigma 0.05", "low rank 32 + 10 % noise"):
    x = torch.cat([make(kind, 1 << 19) for _ in range(0, n, 1 << 19)])[:n]
    q = make(kind, b)
    opts = {}
    if "form off" in kind or "split off" in kind:
        opts["sketch_per_row"] = 0
    if "split off" in kind:
        opts["sketch_split"] = 0
    ske = DeviceIndex(x, dtype="fp16", sketch=True, options=opts)
    s1, i1, _ = ske.search(q, k); torch.cuda.synchronize()
    st = ske.sketch_stats(); out = (st['last_scan_pairs'], st['largest_region'], st['longest_sublist'], st['overflow'])
    state = ske.sketch_state()
    t_s = timed(ske, q)
    if ske.sketch_split(0)[3]:
        kind += " [per-row form]"
    ske.close()
    ref = DeviceIndex(x, dtype="fp16", sketch=False)
    s0, i0, _ = ref.search(q, k); torch.cuda.synchronize()
    t_e = timed(ref, q)
    ref.close()
    same = bool(torch.equal(i0, i1))
    print(f"{kind:62s}: main-scan pairs {out[0]:9d}  overflow {out[3]}  state after {state:3d}  sketch {t_s:.3f} ms  exact {t_e:.3f} ms  same ids {same}  max |ds| {float((s0 - s1).abs().max()):.1e}", flush=True)
    del x

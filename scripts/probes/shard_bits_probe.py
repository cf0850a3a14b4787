"""Dev probe: do the shards of a corpus return, for the rows they hold, the same score bits as the single index of the whole corpus?
(the 8-rank sketch-shards case of tests/sharded_worker.py, in one process)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vietnamese_qa_system_amd import index as _index_mod
from vietnamese_qa_system_amd.index import DeviceIndex
_index_mod.DEFAULT_OPTIONS["stage_min_tiles"] = 2
from vietnamese_qa_system_amd.sharded import shard_bounds
world, k = 8, 10
rng = np.random.default_rng(5)
def unit(rng, n, d):
    v = rng.standard_normal((n, d)).astype(np.float32)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float16)
n3 = 140_000 * world + 3
x3 = unit(rng, n3, 64)
b3 = [shard_bounds(n3, world, r) for r in range(world)]
for lo, _ in b3[1:]:
    x3[lo - 1] = x3[lo] = x3[lo + 65_536] = x3[123]
q3 = unit(rng, 24, 64); q3[0] = x3[123]
q = torch.from_numpy(q3).cuda()
if os.environ.get("VQA_POISON"):  # freed device memory full of 0xFF / NaN patterns: what a long-lived process hands hipMalloc
    for val in (0xFF, 0x7F):
        t = torch.full((8 << 30,), val, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); del t; torch.cuda.empty_cache()
if os.environ.get("VQA_SHARD_FIRST"):  # the order of tests/sharded_worker.py: the rank's shard index exists (and has searched) before the single one
    lo, hi = b3[int(os.environ["VQA_SHARD_FIRST"])]
    first = DeviceIndex(x3[lo:hi], dtype="fp16", id_base=lo)
    first.search(q, k); torch.cuda.synchronize()
    print("shard first: state", first.sketch_state(), first.sketch_stats())
one = DeviceIndex(x3, dtype="fp16")
for rep in range(3):
    s1, i1, p1 = one.search(q, int(os.environ.get("K1", 50)), return_positions=True); torch.cuda.synchronize()
    print("one: rep", rep, "state", one.sketch_state(), one.sketch_stats(), "has copy", one.device_bytes())
full = {}
s1 = s1.cpu().numpy(); p1 = p1.cpu().numpy()
for r, (lo, hi) in enumerate(b3):
    sh = DeviceIndex(x3[lo:hi], dtype="fp16", id_base=lo)
    s, i, p = sh.search(q, k, return_positions=True); torch.cuda.synchronize()
    st = sh.sketch_stats(); state = sh.sketch_state()
    s = s.cpu().numpy(); i = i.cpu().numpy()
    bad = 0
    for row in range(24):
        for j in range(k):
            m = np.nonzero(p1[row] == i[row, j])[0]
            if len(m) and s1[row, m[0]] != s[row, j]:
                bad += 1
    print("shard", r, "rows", hi - lo, "state", state, "overflow", st["overflow"], "pairs", st["rescored_pairs"], "score bits differing from the single index:", bad, flush=True)
    sh.close()

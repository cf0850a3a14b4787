import time, numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
from vietnamese_qa_system_amd import Embeddings
rng = np.random.default_rng(0)
x0 = rng.standard_normal((1000, 768)).astype(np.float32); q0 = rng.standard_normal((256, 768)).astype(np.float32)
for dt in ("fp32", "fp16"):
    emb = Embeddings(dtype=dt, device=0); emb.index_vectors(list(range(1, 1001)), x0)
    def t(fn, n=30):
        for _ in range(5): fn()
        t0 = time.perf_counter()
        for _ in range(n): r = fn()
        return (time.perf_counter() - t0) / n * 1e3
    print(dt, "batchsearch(256,10) %.3f ms" % t(lambda: emb.batchsearch(q0, 10)),
          "| search_host(256) %.3f" % t(lambda: emb._index.search_host(q0, 10, normalize=True)),
          "| search_host(64) %.3f" % t(lambda: emb._index.search_host(q0[:64], 10, normalize=True)),
          "| _format %.3f" % t(lambda s=emb._index.search_host(q0, 10, normalize=True): emb._format(*s)))
    qd = torch.from_numpy(q0).cuda()
    def dev():
        r = emb._index.search(qd, 10); torch.cuda.synchronize(); return r
    print("   device search + sync %.3f" % t(dev))

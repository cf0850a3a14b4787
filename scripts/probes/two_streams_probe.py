"""Dev probe: do the one-question forwards of two encoders overlap when each is given a stream of its own?  XLM-R-base and MiniLM-L12
shapes, replayed graphs and eager launches; time until both are done against the sum of the two alone."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vietnamese_qa_system_amd import encoder as ENC
device = torch.device("cuda", 0)
for graphs in (1,):
    ENC.DEFAULT_OPTIONS["graphs"] = graphs
    encs = []
    for cfg in (ENC.MINILM_L12, ENC.XLMR_BASE):
        enc, ids, mask, lens, g = bench.make_encoder(torch, device, 0, 1, 32, max_tokens=64, cfg=cfg)
        encs.append((enc, ids, torch.ones_like(mask)))
    mode = os.environ.get("STREAMS", "default")
    if mode == "priority":
        streams = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]
    elif mode == "many":  # skip a few of the pool's streams between the two
        pool = [torch.cuda.Stream() for _ in range(8)]
        streams = [pool[0], pool[int(os.environ.get("SECOND", "3"))]]
    else:
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def run(which, same_stream=False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in which:
            enc, ids, mask = encs[i]
            with torch.cuda.stream(streams[0 if same_stream else i]):
                enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=0)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for _ in range(10):
        run((0, 1))
    res = {}
    for name, which, same in (("minilm alone", (0,), False), ("xlmr alone", (1,), False), ("both, one stream", (0, 1), True), ("both, two streams", (0, 1), False)):
        res[name] = np.median([run(which, same) for _ in range(50)]) * 1e3
    print(f"graphs={graphs}: " + " | ".join(f"{k} {v:.3f} ms" for k, v in res.items()), flush=True)
    for enc, *_ in encs:
        enc.close()

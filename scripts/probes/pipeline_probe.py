"""Dev probe: does the question encoder of batch i + 1 overlap with the search of batch i when each is given a stream of its own?
10M x 768 fp16 shard (sketch search), PhoBERT-base-shaped encoder on 256 ragged questions (packed); queries/s over 40 batches:
in sequence on one stream, and pipelined on two streams (two query buffers; stream priorities: encoder high / search high / equal)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vietnamese_qa_system_amd.index import DeviceIndex

device = torch.device("cuda", 0)
n = int(os.environ.get("ROWS", "10000000"))
shard = bench.build_shard(torch, n, 768, 1234, device, "fp16")
ix = DeviceIndex(shard, id_base=1, dtype="fp16", device=0, sketch=True)
del shard
torch.cuda.empty_cache()
b, L, k, steps = 256, 32, 10, 40
enc, ids, mask, lens, g = bench.make_encoder(torch, device, 0, b, L)
real = int(mask.sum())
for _ in range(5):
    ix.search(enc.forward(ids, mask, pooling="cls", normalize=True, real_tokens=real), k)
torch.cuda.synchronize()


def sequential():
    t0 = time.perf_counter()
    for _ in range(steps):
        ix.search(enc.forward(ids, mask, pooling="cls", normalize=True, real_tokens=real), k)
    torch.cuda.synchronize()
    return b * steps / (time.perf_counter() - t0)


def pipelined(pe, ps):
    se, ss = torch.cuda.Stream(priority=pe), torch.cuda.Stream(priority=ps)
    enc_done = [torch.cuda.Event() for _ in range(steps)]
    srch_done = [torch.cuda.Event() for _ in range(steps)]
    qs = [None, None]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(se):
            if i >= 2:
                se.wait_event(srch_done[i - 2])  # the buffer this forward's output replaces has been searched
            qs[i & 1] = enc.forward(ids, mask, pooling="cls", normalize=True, real_tokens=real)
            enc_done[i].record(se)
        with torch.cuda.stream(ss):
            ss.wait_event(enc_done[i])
            qs[i & 1].record_stream(ss)
            res = ix.search(qs[i & 1], k)
            srch_done[i].record(ss)
    torch.cuda.synchronize()
    return b * steps / (time.perf_counter() - t0)


print(f"in sequence, one stream: {np.median([sequential() for _ in range(3)]):.0f} q/s", flush=True)
for pe, ps, name in ((-1, 0, "encoder high priority"), (0, -1, "search high priority"), (0, 0, "equal priorities")):
    print(f"pipelined, {name}: {np.median([pipelined(pe, ps) for _ in range(3)]):.0f} q/s", flush=True)

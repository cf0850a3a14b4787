"""Dev probe: one question (B = 1, L tokens) through the one-launch forward (encoder_persist_kernel) against the launches (persistent = 0),
over the number of resident workgroups.  Event time per call (back to back on one stream) and host wall time of call + synchronise."""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import encoder as E
from vietnamese_qa_system_amd.encoder import QuestionEncoder

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="phobert")
ap.add_argument("--l", default="32")
ap.add_argument("--grids", default="-1,0,24,48,64,96,144,192,256")
ap.add_argument("--reps", type=int, default=300)
args = ap.parse_args()
cfg = dict({"phobert": E.PHOBERT_BASE, "xlmr": E.XLMR_BASE, "minilm": E.MINILM_L12}[args.model], vocab_size=8000)
w = E.synthetic_weights(cfg, seed=3, layers=cfg["layers"])
for L in [int(v) for v in args.l.split(",")]:
    ids, mask = E.synthetic_tokens(cfg, 1, L, seed=5, min_len=L)
    ids_d, mask_d = torch.from_numpy(ids).cuda().int(), torch.from_numpy(mask).cuda().int()
    ref = None
    for g in [int(v) for v in args.grids.split(",")]:
        enc = QuestionEncoder(w, cfg, max_tokens=64, options={"persistent": 0} if g < 0 else {"persistent": 1, "persistent_grid": g})
        for _ in range(20):
            out = enc.forward(ids_d, mask_d, pooling="mean", real_tokens=0)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        if ref is None:
            ref = got
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            enc.forward(ids_d, mask_d, pooling="mean", real_tokens=0)
        e1.record()
        torch.cuda.synchronize()
        ev = e0.elapsed_time(e1) / args.reps
        walls = []
        for _ in range(100):
            t0 = time.perf_counter()
            enc.forward(ids_d, mask_d, pooling="mean", real_tokens=0)
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
        print(f"{args.model} L={L:3d}  {'launches (persistent=0)' if g < 0 else f'one launch, grid {g if g else chr(39)+chr(100)+chr(101)+chr(102)+chr(39)}':28s} "
              f"event {ev * 1e3:7.1f} us per call   wall+sync median {np.median(walls) * 1e6:7.1f} us   same bits as the launches: {np.array_equal(got, ref)}", flush=True)
        enc.close()

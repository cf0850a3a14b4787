"""Generates the retrieval golden vectors under tests/golden/ (run once in the build container; the
.npz files are committed, this script is kept so they can be regenerated).

The reference holds NO golden vectors for this path (txtai/faiss are absent, no tests),
so expected outputs come from two independent third-party implementations that ARE installed here:

* ``torch.nn.functional.cosine_similarity`` -- the call the reference itself makes to score a question
  against a context (``/root/reference/src/test.py:104``) -- followed by ``torch.topk``;
* scikit-learn ``NearestNeighbors(metric="cosine", algorithm="brute")``.

Neither the oracle (``oracle/``) nor the product package is imported here: the fixtures pin the oracle,
they are not produced by it.

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn.functional as F
from sklearn.neighbors import NearestNeighbors

HERE = os.path.dirname(os.path.abspath(__file__))
K = 10


def retr_1k() -> None:
    """BASELINE.json configs[0]: 1k-doc x 768-d random-embedding index, brute-force cosine top-10."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1000, 768)).astype(np.float32)  # raw (un-normalised) document embeddings
    q = rng.standard_normal((256, 768)).astype(np.float32)   # raw query embeddings
    ids = np.arange(1, 1001, dtype=np.int64)                 # sqlite AUTOINCREMENT rowids (setup_db.py:14)
    tx, tq = torch.from_numpy(x), torch.from_numpy(q)
    cos = torch.stack([F.cosine_similarity(tq[i:i + 1], tx, dim=-1) for i in range(tq.shape[0])])  # [B, N]
    ts, tp = torch.topk(cos, K, dim=1, largest=True, sorted=True)
    nn = NearestNeighbors(n_neighbors=K, metric="cosine", algorithm="brute").fit(x)
    dist, sp = nn.kneighbors(q)
    np.savez(os.path.join(HERE, "retr_1k.npz"), x=x, q=q, ids=ids,
             torch_scores=ts.numpy().astype(np.float32), torch_ids=ids[tp.numpy()],
             sk_scores=(1.0 - dist).astype(np.float32), sk_ids=ids[sp])


def retr_ties() -> None:
    """Duplicate rows pin the tie order this project DEFINES (score desc, row position asc); expected output is
    produced by a plain Python sort over exact fp64 dot products of small-integer vectors (no rounding)."""
    rng = np.random.default_rng(1)
    base = rng.integers(-3, 4, size=(40, 64)).astype(np.float32)
    rows = np.concatenate([base, base[:25], base[5:20], base[:10]])  # 90 rows, many exact duplicates
    perm = rng.permutation(rows.shape[0])
    x = rows[perm]
    q = rng.integers(-3, 4, size=(32, 64)).astype(np.float32)
    ids = (np.arange(x.shape[0], dtype=np.int64) * 7 + 100)  # arbitrary, monotonic external ids
    exp_pos = np.zeros((q.shape[0], K), dtype=np.int64)
    exp_sc = np.zeros((q.shape[0], K), dtype=np.float32)
    for i in range(q.shape[0]):
        sc = [float(np.dot(q[i].astype(np.float64), x[j].astype(np.float64))) for j in range(x.shape[0])]
        order = sorted(range(x.shape[0]), key=lambda j: (-sc[j], j))[:K]
        exp_pos[i] = order
        exp_sc[i] = [sc[j] for j in order]
    np.savez(os.path.join(HERE, "retr_ties.npz"), x=x, q=q, ids=ids, exp_pos=exp_pos, exp_ids=ids[exp_pos],
             exp_scores=exp_sc)


def retr_same_stored() -> None:
    """Same-stored-value fixtures: the inputs ARE the values each index type keeps in HBM (fp16 values, OCP e4m3 codes of
    16 * x, fp32 values), and the expected top-(K + 1) comes from torch float64 scoring of exactly those values (torch's own
    float8_e4m3fn cast is the codec) -- neither ``oracle/`` nor the product is involved.  The GPU tests compare ids bit for
    bit wherever the fp64 gap to the neighbouring ranks exceeds the fp32 accumulation error (2e-6), K + 1 scores are kept
    for that purpose."""
    g = torch.Generator().manual_seed(20260)
    out = {}

    def unit(n, d):
        v = torch.randn((n, d), generator=g, dtype=torch.float32)
        return v / v.norm(dim=1, keepdim=True)

    def expect(qv: torch.Tensor, xv: torch.Tensor, scale: float = 1.0):
        s = (qv.double() @ xv.double().T) * scale
        order = torch.argsort(-s, dim=1, stable=True)[:, :K + 1]  # score desc, position asc
        return order.numpy().astype(np.int64), torch.gather(s, 1, order).numpy()

    # fp16 index (BASELINE configs[2] storage type): 2000 x 128 stored halves
    x16, q16 = unit(2000, 128).half(), unit(64, 128).half()
    out["f16_x"], out["f16_q"] = x16.numpy(), q16.numpy()
    out["f16_pos"], out["f16_scores"] = expect(q16, x16)
    # fp8 index (configs[4]): codes of 16 * x for rows and queries; score = <dec(q), dec(x)> / 256
    x8 = (unit(2000, 128) * 16).to(torch.float8_e4m3fn)
    q8 = (unit(64, 128) * 16).to(torch.float8_e4m3fn)
    out["fp8_x_codes"], out["fp8_q_codes"] = x8.view(torch.uint8).numpy(), q8.view(torch.uint8).numpy()
    out["fp8_pos"], out["fp8_scores"] = expect(q8.float(), x8.float(), 1.0 / 256)
    # fp32 index (configs[1]): 1000 x 128 stored floats
    x32, q32 = unit(1000, 128), unit(64, 128)
    out["f32_x"], out["f32_q"] = x32.numpy(), q32.numpy()
    out["f32_pos"], out["f32_scores"] = expect(q32, x32)
    np.savez(os.path.join(HERE, "retr_same_stored.npz"), **out)


if __name__ == "__main__":
    import sys
    which = sys.argv[1:] or ["retr_1k", "retr_ties", "retr_same_stored"]
    for name in which:
        {"retr_1k": retr_1k, "retr_ties": retr_ties, "retr_same_stored": retr_same_stored}[name]()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))

"""Dev probe (torch arithmetic only, no product kernels in the counting): how many (query, row) pairs would survive the sketch bound
on data that is NOT isotropic, under four forms of the bound --
  (i)   the product's: y = T(x - mu) sketched per tile, z = Tq sketched per query;
  (ii)  query split along the centre direction u = mu / |mu|, rows projected off u, the rank-one term alpha * beta_x bounded per TILE
        (alpha = q.u, beta_x = u.(x - mu); needs two more floats per tile and one per query, nothing per row);
  (iii) the same with the exact per-ROW term alpha * beta_x (needs a per-row value inside the scan's epilogue);
  (iv)  the product's sketch, only the slack term |q . x_lo| split into |alpha| * max_tile|u . x_lo| + |q_r| * max_tile|x_lo|.
Rows: this build's own encoder outputs (random init: collapsed onto one direction), a common component (mean cosine 0.5), 1000 clusters,
isotropic.  Prints candidates per query for k = 10; the k-th best score comes from the exact scores of the same rows."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device("cuda", 0)
d, B, K, TILE = 768, 256, 10, 256
gen = torch.Generator(device=dev)
gen.manual_seed(5)
signs = (torch.randint(0, 2, (d,), generator=gen, device=dev) * 2 - 1).float()


def fwht(v):
    """normalised Walsh-Hadamard transform of the 512- and the 256-block of every row (d = 768)"""
    out = []
    for lo, n in ((0, 512), (512, 256)):
        a = v[:, lo:lo + n].clone()
        h = 1
        while h < n:
            a = a.view(-1, n // (2 * h), 2, h)
            a = torch.stack((a[:, :, 0] + a[:, :, 1], a[:, :, 0] - a[:, :, 1]), dim=2).reshape(-1, n)
            h *= 2
        out.append(a / n ** 0.5)
    return torch.cat(out, dim=1)


def T(v):
    return fwht(v * signs)


def sketch_rows(y):
    """per-tile scale, codes, hi / lo parts"""
    n = y.shape[0]
    yt = y.view(n // TILE, TILE, d)
    s = yt.abs().amax(dim=(1, 2)).clamp_min(1e-30) / 127
    code = torch.clamp(torch.round(yt / s[:, None, None]), -127, 127)
    hi = code * s[:, None, None]
    lo = yt - hi
    return code.view(n, d), s, hi.norm(dim=2).amax(1), lo.norm(dim=2).amax(1), lo.view(n, d)


def sketch_q(z):
    s = z.abs().amax(1).clamp_min(1e-30) / 127
    code = torch.clamp(torch.round(z / s[:, None]), -127, 127)
    return code, s, (z - code * s[:, None]).norm(dim=1)


def count(x, q, name):
    n = x.shape[0] // TILE * TILE
    x = x[:n].float()
    q = q.float()
    scores = q @ x.T
    theta = scores.topk(K, dim=1).values[:, -1]
    mu = x[:: max(1, n // 65536)].mean(0)
    u = mu / mu.norm()
    qmu = q @ mu
    tiles = n // TILE

    def survivors(bound):
        return int((bound >= theta[:, None]).sum())

    res = {}
    # (i) the product's form
    code, s_t, A, Bt, lo = sketch_rows(T(x - mu))
    qc, s_q, qlo = sketch_q(T(q))
    D = (qc @ code.T).view(B, tiles, TILE)
    base = qmu[:, None, None] + s_q[:, None, None] * s_t[None, :, None] * D
    res["(i) product"] = survivors((base + (qlo[:, None] * A[None, :] + q.norm(dim=1)[:, None] * Bt[None, :])[:, :, None]).view(B, n))
    # (iv) slack term split along u
    Tu = T(u[None, :])[0]
    C = (lo @ Tu).abs().view(tiles, TILE).amax(1)
    alpha = q @ u
    qr = q - alpha[:, None] * u[None, :]
    res["(iv) product sketch, q.x_lo split"] = survivors(
        (base + (qlo[:, None] * A[None, :] + alpha.abs()[:, None] * C[None, :] + qr.norm(dim=1)[:, None] * Bt[None, :])[:, :, None]).view(B, n))
    del D, base
    # (ii), (iii) query split, rows projected off u
    beta = (x - mu) @ u
    r = x - mu - beta[:, None] * u[None, :]
    code, s_t, A, Bt, _ = sketch_rows(T(r))
    qc, s_q, qlo = sketch_q(T(qr))
    D = (qc @ code.T).view(B, tiles, TILE)
    slack = (qlo[:, None] * A[None, :] + qr.norm(dim=1)[:, None] * Bt[None, :])[:, :, None]
    core = qmu[:, None, None] + s_q[:, None, None] * s_t[None, :, None] * D + slack
    bt = beta.view(tiles, TILE)
    bmax, bmin = bt.amax(1), bt.amin(1)
    rank1_tile = torch.where(alpha[:, None] >= 0, alpha[:, None] * bmax[None, :], alpha[:, None] * bmin[None, :])
    res["(ii) split, per-tile beta"] = survivors((core + rank1_tile[:, :, None]).view(B, n))
    full = (core + alpha[:, None, None] * bt[None, :, :]).view(B, n)
    res["(iii) split, per-row beta"] = survivors(full)
    # the cascade's thresholds: the k-th best score of the first 10 % / 30 % of the rows instead of the final one
    for frac in (0.1, 0.3):
        m = int(n * frac) // TILE * TILE
        th = scores[:, :m].topk(K, dim=1).values[:, -1]
        res[f"(iii) against the k-th best of the first {int(frac * 100)} % of the rows"] = int((full >= th[:, None]).sum())
        gap = (theta - th)
        res[f"      (that threshold lies {gap.mean().item():.5f} below the final one; slack of the bound {slack.mean().item():.5f})"] = 0
    sig = scores.std().item()
    print(f"{name}: rows {n}, |mu| {mu.norm():.4f}, score sigma {sig:.5f}, (k-th best - mean) / sigma {((theta - scores.mean(1)).mean() / sig):.2f}, "
          f"alpha {alpha.mean():.3f}, |q_r| {qr.norm(dim=1).mean():.3f}, beta sigma {beta.std():.5f} (within tiles: max - min {(bmax - bmin).mean():.5f}), "
          f"sigma of q_r.r_x {(qr @ r.T).std():.5f}")
    for kname, v in res.items():
        print(f"    {kname:40s} {v / B:10.1f} candidates per query ({v / (B * n) * 100:.4f} % of the pairs)")
    sys.stdout.flush()


def unit(v):
    return v / v.norm(dim=1, keepdim=True)


def main():
    n = 1 << 20
    # own encoder outputs
    import bench
    import numpy as np
    S, L = 1024, 32
    enc, _, _, _, g = bench.make_encoder(torch, dev, 0, S, L, max_tokens=S * L)
    from vietnamese_qa_system_amd.encoder import PHOBERT_BASE
    x = torch.empty((n, d), dtype=torch.float16, device=dev)
    for c0 in range(0, n, S):
        ids, mask, lens = bench.make_tokens(torch, dev, g, PHOBERT_BASE, S, L)
        x[c0:c0 + S] = enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=int(lens.sum())).to(torch.float16)
    ids, mask, lens = bench.make_tokens(torch, dev, g, PHOBERT_BASE, B, L)
    q = enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=int(lens.sum())).to(torch.float16)
    enc.close()
    count(x, q, "own encoder outputs (random init, mean pooling)")
    del x
    c = unit(torch.randn((1, d), generator=gen, device=dev))
    for w, label in ((1.0, "common component, mean cosine 0.5"), (3.0, "common component, mean cosine 0.9")):
        draw = lambda m: unit(w * c + unit(torch.randn((m, d), generator=gen, device=dev))).half()
        count(draw(n), draw(B), label)
    centres = unit(torch.randn((1000, d), generator=gen, device=dev))
    draw = lambda m: unit(centres[torch.randint(0, 1000, (m,), generator=gen, device=dev)] + 0.3 * torch.randn((m, d), generator=gen, device=dev) / d ** 0.5).half()
    count(draw(n), draw(B), "1000 clusters, sigma 0.3")
    draw = lambda m: unit(torch.randn((m, d), generator=gen, device=dev)).half()
    count(draw(n), draw(B), "isotropic")


main()

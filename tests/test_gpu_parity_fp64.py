"""Oracle-INDEPENDENT parity at BASELINE geometry -- needs an MI355X; nothing under oracle/ is imported.

tests/golden/retr_same_stored.npz pins a toy geometry (2000 x 128: 8 tiles, one per workgroup).  Here the expected top-11
come from torch float64 on the host at test time (seeded recipe, no 150 MB fixture) for shapes that drive the deep
machinery of the scoring kernel: d = 768 (24 K-steps per tile), hundreds of tiles (many tiles per workgroup: the X ring
runs across tile boundaries, lists spill and compact), the seed pass at its cap, the two-stage search (first-stage
thresholds handed to the main launch) and exact duplicate rows in both stages.  The values scored are the very values the
index stores (fp16 values; e4m3 codes by torch's own float8 codec), so ids must match bit for bit wherever the fp64 gap
to both neighbouring ranks exceeds the accumulation error; inside a near-tie group the returned row must belong to it,
and rows with bit-equal scores must come back in ascending position (the tie order this build defines).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
K = 10


def _unit(gen, n, d):
    x = torch.randn((n, d), generator=gen, dtype=torch.float32)
    return x / x.norm(dim=1, keepdim=True)


def _fp64_top(values_x: torch.Tensor, values_q: torch.Tensor, k1: int, chunk: int = 25000):
    """Top-k1 of q . x in float64 over all rows, (scores [B, k1], positions [B, k1]), score desc / position asc."""
    q64 = values_q.double()
    best_s = best_p = None
    for c0 in range(0, values_x.shape[0], chunk):
        s = q64 @ values_x[c0:c0 + chunk].double().T
        kk = min(k1, s.shape[1])
        ts, tp = torch.topk(s, kk, dim=1)
        tp = tp + c0
        if best_s is not None:
            ts, tp = torch.cat([best_s, ts], 1), torch.cat([best_p, tp], 1)
        # score desc, position asc (lexsort: last key is primary)
        order = torch.from_numpy(np.lexsort((tp.numpy(), -ts.numpy()), axis=1))[:, :k1]
        best_s, best_p = torch.gather(ts, 1, order), torch.gather(tp, 1, order)
    return best_s.numpy(), best_p.numpy()


def _check(s, p, exp_sc, exp_pos, tie_tol, score_tol, min_clear):
    assert np.abs(s - exp_sc[:, :K]).max() <= score_tol, np.abs(s - exp_sc[:, :K]).max()
    gaps = exp_sc[:, :-1] - exp_sc[:, 1:]  # between rank j and j + 1, j = 0 .. K - 1
    clear_below = gaps[:, :K] > tie_tol
    clear_above = np.concatenate([np.ones((s.shape[0], 1), bool), gaps[:, :K - 1] > tie_tol], axis=1)
    clear = clear_above & clear_below
    assert np.array_equal(p[clear], exp_pos[:, :K][clear]), "a row outside every near-tie band differs from the fp64 expectation"
    for b, j in zip(*np.nonzero(~clear)):  # near tie: the row must belong to the tie group around this rank
        group = {int(exp_pos[b, t]) for t in range(K + 1) if abs(exp_sc[b, t] - exp_sc[b, j]) <= tie_tol}
        assert int(p[b, j]) in group, (b, j)
    assert clear.mean() >= min_clear, clear.mean()
    # bit-equal scores come back in ascending row position
    same = s[:, 1:] == s[:, :-1]
    assert (p[:, 1:][same] > p[:, :-1][same]).all()


def _search(x16, q16, dtype, env=None):
    from vietnamese_qa_system_amd.index import DeviceIndex
    # (env: the round-4 vocabulary of these tests; VQA_STAGE_MIN is vqa_index_options.stage_min_tiles, read when the index is created)
    options = {{"VQA_STAGE_MIN": "stage_min_tiles"}[k]: int(v) for k, v in (env or {}).items()}
    ix = DeviceIndex(x16, id_base=0, dtype=dtype, device=0, options=options)
    s, _, p = ix.search(q16.cuda(), K, return_positions=True)
    torch.cuda.synchronize()
    info = ix.launch_info(q16.shape[0], K)
    return s.cpu().numpy(), p.cpu().numpy(), ix, info


def test_fp16_100k_x_768_against_fp64(native_lib):
    """BASELINE configs[2] geometry (d = 768, B = 256, k = 10) on 100 000 rows: 391 tiles over 256 workgroups."""
    gen = torch.Generator().manual_seed(20260301)
    x16, q16 = _unit(gen, 100_000, 768).half(), _unit(gen, 256, 768).half()
    x16[70_001] = x16[123]
    x16[99_999] = x16[123]  # the shard's last row (ragged tile): an exact three-way tie for query 0
    q16[0] = x16[123]
    exp_sc, exp_pos = _fp64_top(x16, q16, K + 1)
    s, p, ix, info = _search(x16, q16, "fp16")
    ix.close()
    assert info.seed_tiles >= 24 and info.first_stage_rows == 0
    _check(s, p, exp_sc, exp_pos, tie_tol=2e-6, score_tol=1e-5, min_clear=0.97)
    assert p[0, :3].tolist() == [123, 70_001, 99_999]


@pytest.mark.parametrize("dtype", ["fp16", "fp8"])
def test_two_stage_300001_x_64_against_fp64(native_lib, dtype):
    """300 001 x 64 with VQA_STAGE_MIN=2: 1172 tiles, the first 256 scored by the first-stage launch whose exact k-th best
    scores seed the main launch.  Runs of exact duplicates sit in both stages and across the stage boundary (row 65 536)."""
    gen = torch.Generator().manual_seed(20260302)
    x, q = _unit(gen, 300_001, 64), _unit(gen, 256, 64)
    if dtype == "fp8":
        # the codes an fp8 index stores, by torch's own codec; value = code / 16 is exact in fp16, so the index holds these codes
        x16 = ((x * 16).to(torch.float8_e4m3fn).float() / 16).half()
        q16 = ((q * 16).to(torch.float8_e4m3fn).float() / 16).half()
    else:
        x16, q16 = x.half(), q.half()
    for j, rows in enumerate(([5, 6, 7, 40_000], [65_535, 65_536, 65_537], [100_000, 250_000, 300_000], [1000, 200_001])):
        for r in rows[1:]:
            x16[r] = x16[rows[0]]
        q16[j] = x16[rows[0]]
    exp_sc, exp_pos = _fp64_top(x16, q16, K + 1)
    s, p, ix, info = _search(x16, q16, dtype, env={"VQA_STAGE_MIN": "2"})
    if dtype == "fp8":
        codes, _ = ix.get_rows(0, 4096)
        assert np.array_equal(codes, (x16[:4096].float() * 16).to(torch.float8_e4m3fn).view(torch.uint8).numpy())
    ix.close()
    assert info.first_stage_rows == 256 * 256, info.first_stage_rows
    # the fp8 MFMA accumulates with 2^-15 relative error: a wider band, as in test_gpu_golden_same_stored.py
    tol = dict(tie_tol=2e-6, score_tol=1e-5, min_clear=0.95) if dtype == "fp16" else dict(tie_tol=1e-4, score_tol=5e-5, min_clear=0.75)
    _check(s, p, exp_sc, exp_pos, **tol)
    assert p[0, :4].tolist() == [5, 6, 7, 40_000] and p[1, :3].tolist() == [65_535, 65_536, 65_537]
    assert p[2, :3].tolist() == [100_000, 250_000, 300_000] and p[3, :2].tolist() == [1000, 200_001]
    # the same search in one stage returns the same rows; the same bits too for fp8 -- an fp16 shard of this size runs its main
    # launch over the int8 sketch and scores the surviving pairs in a summation order of its own (last-bit differences)
    s1, p1, ix1, info1 = _search(x16, q16, dtype, env={"VQA_STAGE_MIN": "0"})
    ix1.close()
    assert info1.first_stage_rows == 0 and np.array_equal(p, p1)
    assert info.sketch_scan == (1 if dtype == "fp16" else 0) and info1.sketch_scan == 0
    assert np.array_equal(s, s1) if dtype == "fp8" else np.abs(s - s1).max() <= 3e-7

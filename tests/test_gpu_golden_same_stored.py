"""HIP path against oracle-INDEPENDENT golden vectors on the same stored values -- needs an MI355X.

tests/golden/retr_same_stored.npz (made by tests/golden/make_golden.py: torch float64 scoring, torch's own float8 codec)
holds the very values each index type keeps in HBM and the expected top-11.  Nothing under oracle/ is imported here: ids
must match bit for bit wherever the fp64 gap to both neighbouring ranks exceeds the fp32 accumulation error; inside a
near-tie group the returned row must still be one of the group's rows.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

K = 10
# (index dtype, tie tolerance = gap below which the accumulation error may reorder, score tolerance, share of (query, rank)
# pairs of the fixture whose gaps are clear of the tie tolerance: those must match bit for bit).  The fp8 MFMA accumulates
# with 2^-15 relative error, hence its wider band.
CASES = {"f16": ("fp16", 2e-6, 1e-5, 0.98), "f32": ("fp32", 2e-6, 1e-5, 0.98), "fp8": ("fp8", 1e-4, 5e-5, 0.9)}
E4M3 = None


def _e4m3_values():
    """code -> value through torch's codec (the one that made the fixture)"""
    global E4M3
    if E4M3 is None:
        E4M3 = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float().numpy()
    return E4M3


def _inputs(g, key):
    if key == "fp8":  # rows whose e4m3(16 x) codes are exactly the fixture's: x = decode(code) / 16 is exact in fp16
        x = (_e4m3_values()[g["fp8_x_codes"]] / 16.0).astype(np.float16)
        q = (_e4m3_values()[g["fp8_q_codes"]] / 16.0).astype(np.float16)
        return x, q
    return g[f"{key}_x"], g[f"{key}_q"]


@pytest.mark.parametrize("key", ["f16", "f32", "fp8"])
def test_ids_bit_exact_against_fp64_golden(native_lib, golden_dir, key):
    from vietnamese_qa_system_amd.index import DeviceIndex
    g = np.load(f"{golden_dir}/retr_same_stored.npz")
    dtype, tie_tol, score_tol, min_clear = CASES[key]
    x, q = _inputs(g, key)
    ix = DeviceIndex(x, id_base=0, dtype=dtype, device=0)
    if key == "fp8":  # the index must hold exactly the fixture's codes
        codes, _ = ix.get_rows()
        assert np.array_equal(codes, g["fp8_x_codes"])
    s, i, p = ix.search(torch.from_numpy(q).cuda(), K, return_positions=True)
    torch.cuda.synchronize()
    s, p = s.cpu().numpy(), p.cpu().numpy()
    ix.close()
    exp_pos, exp_sc = g[f"{key}_pos"], g[f"{key}_scores"]  # [B, K + 1]
    assert np.abs(s - exp_sc[:, :K]).max() <= score_tol
    gaps = exp_sc[:, :-1] - exp_sc[:, 1:]  # gap between rank j and j + 1, j = 0 .. K - 1
    exact = 0
    for b in range(q.shape[0]):
        for j in range(K):
            clear_above = j == 0 or gaps[b, j - 1] > tie_tol
            clear_below = gaps[b, j] > tie_tol
            if clear_above and clear_below:
                assert p[b, j] == exp_pos[b, j], f"query {b} rank {j}: row {p[b, j]} != golden {exp_pos[b, j]}"
                exact += 1
            else:  # near tie: the row must belong to the tie group around this rank
                group = {int(exp_pos[b, t]) for t in range(K + 1) if abs(exp_sc[b, t] - exp_sc[b, j]) <= tie_tol}
                assert int(p[b, j]) in group
    assert exact >= min_clear * q.shape[0] * K  # the fixture is almost free of near ties: the exact branch carries the test

"""The CPU oracle against its pins (no GPU): golden vectors produced by torch F.cosine_similarity (the reference's own
scoring call, src/test.py:104) and scikit-learn brute force, the defined tie order, the independent C restatement,
the fp8 codec, and the shard-merge identity."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import retrieval as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def flat_ip():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libflat_ip.so"))
    lib.flat_ip_search.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                   ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

    def search(q, x, dtype, k):
        q = np.ascontiguousarray(q, dtype=np.float32)
        x = np.ascontiguousarray(x)
        s = np.empty((q.shape[0], k), np.float32)
        p = np.empty((q.shape[0], k), np.int64)
        assert lib.flat_ip_search(q.ctypes.data, q.shape[0], q.shape[1], x.ctypes.data, dtype, x.shape[0], k,
                                  s.ctypes.data, p.ctypes.data) == 0
        return s, p
    return search


def test_golden_1k_torch_and_sklearn(golden_dir):
    """BASELINE.json configs[0]: 1k x 768 random index, brute-force cosine top-10."""
    g = np.load(f"{golden_dir}/retr_1k.npz")
    s, i, p = R.search(R.l2_normalize(g["q"]), R.l2_normalize(g["x"]), 10, ids=g["ids"])
    assert np.array_equal(i, g["torch_ids"])
    assert np.array_equal(i, g["sk_ids"])
    assert np.abs(s - g["torch_scores"]).max() < 5e-7
    assert np.abs(s - g["sk_scores"]).max() < 5e-7
    assert np.array_equal(i, p + 1)  # sqlite rowids: id = position + 1


def test_golden_ties_order(golden_dir):
    g = np.load(f"{golden_dir}/retr_ties.npz")
    s, i, p = R.search(g["q"], g["x"], 10, ids=g["ids"], chunk=17)
    assert np.array_equal(p, g["exp_pos"]) and np.array_equal(i, g["exp_ids"]) and np.array_equal(s, g["exp_scores"])


@pytest.mark.parametrize("chunk", [7, 64, 100000])
def test_chunking_is_invisible(chunk):
    rng = np.random.default_rng(2)
    x = R.l2_normalize(rng.standard_normal((3000, 48)).astype(np.float32))
    q = R.l2_normalize(rng.standard_normal((9, 48)).astype(np.float32))
    ref = R.search(q, x, 10, chunk=1 << 20)
    got = R.search(q, x, 10, chunk=chunk)
    assert np.array_equal(ref[2], got[2]) and np.allclose(ref[0], got[0], atol=1e-6)


@pytest.mark.parametrize("dtype", [R.DTYPE_F32, R.DTYPE_F16, R.DTYPE_FP8_E4M3])
def test_c_restatement_agrees(flat_ip, dtype):
    rng = np.random.default_rng(3)
    x = R.quantize_rows(R.l2_normalize(rng.standard_normal((2000, 96)).astype(np.float32)), dtype)
    q = R.l2_normalize(rng.standard_normal((8, 96)).astype(np.float32))
    s, _, p = R.search(q, x, 10, dtype=dtype)
    cs, cp = flat_ip(q, x, dtype, 10)
    R.check_topk(cs, cp, R.full_scores(q, x, dtype), 10, score_tol=2e-6, tie_tol=2e-6)
    assert (cp == p).mean() > 0.99


def test_c_restatement_ties_and_short(flat_ip, golden_dir):
    g = np.load(f"{golden_dir}/retr_ties.npz")
    cs, cp = flat_ip(g["q"], g["x"], R.DTYPE_F32, 10)
    assert np.array_equal(cp, g["exp_pos"]) and np.array_equal(cs, g["exp_scores"])
    cs, cp = flat_ip(g["q"][:2], g["x"][:4], R.DTYPE_F32, 10)
    assert np.all(cp[:, 4:] == -1) and np.all(np.isneginf(cs[:, 4:]))


def test_e4m3_codec_matches_torch():
    rng = np.random.default_rng(4)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-12, 6, 200000))).astype(np.float32)
    x = x[np.abs(x) <= 448]
    x = np.concatenate([x, np.array([0.0, -0.0, 448, -448, 2 ** -9, 2 ** -10, 1.5 * 2 ** -9, 2 ** -6, 0.0625], np.float32)])
    codes = R.e4m3_encode(x)
    t = torch.from_numpy(x).to(torch.float8_e4m3fn)
    assert np.array_equal(codes & 0x7F, t.view(torch.uint8).numpy() & 0x7F)  # -0.0 sign aside
    assert np.array_equal(R.e4m3_decode(codes), t.float().numpy())
    assert R.e4m3_decode(R.e4m3_encode(np.array([1e6, -1e6], np.float32))).tolist() == [448.0, -448.0]  # saturating


@pytest.mark.parametrize("shards", [2, 4, 8])
def test_sharded_merge_equals_unsharded(golden_dir, shards):
    g = np.load(f"{golden_dir}/retr_1k.npz")
    x, q = R.l2_normalize(g["x"]), R.l2_normalize(g["q"])[:40]
    ref_s, ref_i, _ = R.search(q, x, 10, ids=g["ids"])
    n = x.shape[0]
    parts_s, parts_i = [], []
    for r in range(shards):
        lo, hi = r * n // shards, (r + 1) * n // shards
        s, i, _ = R.search(q, x[lo:hi], 10, ids=g["ids"][lo:hi])
        parts_s.append(s)
        parts_i.append(i)
    ms, mi = R.merge_shards(np.stack(parts_s), np.stack(parts_i), 10)
    assert np.array_equal(mi, ref_i) and np.array_equal(ms, ref_s)


def test_merge_ties_prefer_lower_rank_then_slot():
    s = np.array([[[1.0, 0.5]], [[1.0, 0.5]], [[2.0, 1.0]]], np.float32)  # R=3, B=1, k=2
    i = np.array([[[10, 11]], [[20, 21]], [[30, 31]]], np.int64)
    ms, mi = R.merge_shards(s, i, 4)
    assert mi.tolist() == [[30, 10, 20, 31]] and ms.tolist() == [[2.0, 1.0, 1.0, 1.0]]


def test_comparator_catches_errors():
    rng = np.random.default_rng(5)
    x = R.l2_normalize(rng.standard_normal((500, 32)).astype(np.float32))
    q = R.l2_normalize(rng.standard_normal((4, 32)).astype(np.float32))
    s, _, p = R.search(q, x, 5)
    full = R.full_scores(q, x)
    R.check_topk(s, p, full, 5)
    bad = p.copy()
    bad[0, 0] = int(np.argmin(full[0]))
    with pytest.raises(AssertionError):
        R.check_topk(s, bad, full, 5)
    with pytest.raises(AssertionError):
        R.check_topk(s + 1e-3, p, full, 5)
    swapped = p.copy()
    swapped[1, [0, 1]] = swapped[1, [1, 0]]
    with pytest.raises(AssertionError):
        R.check_topk(s, swapped, full, 5)


def test_recall_at_k():
    a = np.array([[1, 2, 3], [4, 5, 6]])
    assert R.recall_at_k(a, a) == 1.0
    assert abs(R.recall_at_k(a, np.array([[1, 2, 9], [7, 8, 9]])) - (2 / 3 + 0) / 2) < 1e-12


@pytest.mark.parametrize("key,dtype", [("f16", R.DTYPE_F16), ("f32", R.DTYPE_F32), ("fp8", R.DTYPE_FP8_E4M3)])
def test_oracle_against_same_stored_fp64_golden(golden_dir, key, dtype):
    """The oracle scores the stored values of every index type like torch float64 does (fixture made without the oracle;
    for fp8 the codes come from torch's own float8_e4m3fn cast, which also pins the oracle's codec)."""
    g = np.load(f"{golden_dir}/retr_same_stored.npz")
    if key == "fp8":
        e4m3 = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float().numpy()
        assert np.array_equal(np.nan_to_num(R.e4m3_decode(np.arange(256, dtype=np.uint8)), nan=7e7), np.nan_to_num(e4m3, nan=7e7))
        x = g["fp8_x_codes"]
        q = R.e4m3_decode(g["fp8_q_codes"]) / 256.0
        # the oracle's encoder reproduces the codes from the decoded values
        assert np.array_equal(R.e4m3_encode(R.e4m3_decode(x)), x)
    else:
        x, q = g[f"{key}_x"], g[f"{key}_q"].astype(np.float32)
    s, _, p = R.search(q, x, 10, dtype=dtype, acc="f64")
    exp_pos, exp_sc = g[f"{key}_pos"], g[f"{key}_scores"]
    assert np.abs(s - exp_sc[:, :10]).max() < 1e-6
    gaps = exp_sc[:, :-1] - exp_sc[:, 1:]
    clear = (gaps[:, :10] > 1e-9) & np.concatenate([np.ones((gaps.shape[0], 1), bool), gaps[:, :9] > 1e-9], axis=1)
    assert np.array_equal(p[clear], exp_pos[:, :10][clear]) and clear.mean() > 0.99


def test_fast_e4m3_encoder_equals_the_reference_definition():
    """oracle.e4m3_encode_fast (integer passes on the fp32 bits; what bench.py uses over millions of rows) against
    oracle.e4m3_encode (the float64 definition pinned to torch's codec above): every code point, every midpoint between
    neighbouring code points and its two fp32 neighbours (round half to even), the saturation edge, non-finite values, and
    random data over 27 binades."""
    tab = R.e4m3_decode(np.arange(256, dtype=np.uint8)).astype(np.float64)
    pos = np.sort(tab[np.isfinite(tab) & (tab >= 0)])
    mids = ((pos[:-1] + pos[1:]) / 2).astype(np.float32)
    pts = np.concatenate([pos.astype(np.float32), mids, np.nextafter(mids, np.float32(0)), np.nextafter(mids, np.float32(1e9)),
                          np.array([448, 449, 463.9, 464, 464.1, 480, 500, 1e9, np.inf, np.nan, 1e-10, 0.0, 2 ** -10,
                                    2 ** -10 * 1.0001, 2 ** -9, 2 ** -7], dtype=np.float32)])
    pts = np.concatenate([pts, -pts])
    with np.errstate(invalid="ignore"):
        assert np.array_equal(R.e4m3_encode(pts), R.e4m3_encode_fast(pts))
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(300_000) * np.exp(rng.uniform(-12, 7, 300_000))).astype(np.float32)
    assert np.array_equal(R.e4m3_encode(x), R.e4m3_encode_fast(x))

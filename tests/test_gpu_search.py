"""Parity of the HIP scoring + top-k path (through the C ABI) against the CPU oracle -- needs an MI355X.

Bar: ids exact wherever the oracle's neighbouring score gaps exceed tie_tol = 2e-6, set
membership inside near-tie groups; scores within 1e-5 absolute (fp16 index, fp32 accumulate).  The oracle
scores the SAME stored fp16 values in fp64.
"""
import numpy as np
import pytest
import torch

from conftest import set_option

from oracle import retrieval as R

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-5
TIE_TOL = 2e-6


def _mk(n, d, b, seed):
    rng = np.random.default_rng(seed)
    x = R.l2_normalize(rng.standard_normal((n, d)).astype(np.float32)).astype(np.float16)
    q = R.l2_normalize(rng.standard_normal((b, d)).astype(np.float32)).astype(np.float16)
    return x, q


def _search(x16, q16, k, ids=None, id_base=0):
    from vietnamese_qa_system_amd.index import DeviceIndex
    ix = DeviceIndex(x16, ids=ids, id_base=id_base, dtype="fp16", device=0)
    s, i, p = ix.search(torch.from_numpy(q16).cuda(), k, return_positions=True)
    torch.cuda.synchronize()
    out = s.cpu().numpy(), i.cpu().numpy(), p.cpu().numpy()
    ix.close()
    return out


@pytest.mark.parametrize("n,d,b,k", [
    (1000, 768, 256, 10),   # BASELINE configs[0] shape
    (1, 64, 3, 1),
    (7, 64, 1, 10),         # fewer rows than k: padded tail
    (255, 128, 5, 10), (256, 128, 5, 10), (257, 128, 5, 10),
    (5000, 768, 37, 12),
    (4097, 100, 9, 3),      # d not a multiple of 64: zero padded
    (70001, 64, 300, 10),   # more tiles than workgroups + two query tiles + ragged tail
    (3001, 100, 257, 10),   # query tiles of equal size (129 + 128): the second tile starts at an 8-byte aligned row
    (2000, 33, 259, 5),     # ... and at a 4-byte aligned one (66-byte rows); 130 + 129
    (1500, 64, 777, 3),     # four tiles of 195 / 195 / 195 / 192
])
def test_search_matches_oracle(native_lib, n, d, b, k):
    x, q = _mk(n, d, b, seed=n + d + b)
    s, i, p = _search(x, q, k, id_base=1)
    kk = min(k, n)
    s_full = R.full_scores(q.astype(np.float32), x, R.DTYPE_F16)
    R.check_topk(s[:, :kk], p[:, :kk], s_full, k, score_tol=SCORE_TOL, tie_tol=TIE_TOL)
    assert np.array_equal(i[:, :kk], p[:, :kk] + 1)
    if kk < k:
        assert np.all(np.isneginf(s[:, kk:])) and np.all(i[:, kk:] == -1) and np.all(p[:, kk:] == -1)


def test_golden_1k(native_lib, golden_dir):
    g = np.load(f"{golden_dir}/retr_1k.npz")
    x16 = R.l2_normalize(g["x"]).astype(np.float16)
    q16 = R.l2_normalize(g["q"]).astype(np.float16)
    s, i, p = _search(x16, q16, 10, ids=g["ids"])
    s_full = R.full_scores(q16.astype(np.float32), x16, R.DTYPE_F16)
    R.check_topk(s, p, s_full, 10, score_tol=SCORE_TOL, tie_tol=TIE_TOL)
    # against the fp32 torch/sklearn golden: fp16 storage may only reorder near-ties
    assert R.recall_at_k(i, g["torch_ids"]) >= 0.99
    assert np.abs(s - g["torch_scores"]).max() < 2e-3


def test_golden_ties_bit_exact(native_lib, golden_dir):
    """Small-integer vectors: every product and sum is exact in fp16 x fp16 -> fp32, so ids, positions AND scores
    must be bit-exact, including the defined tie order (score desc, row position asc)."""
    g = np.load(f"{golden_dir}/retr_ties.npz")
    s, i, p = _search(g["x"].astype(np.float16), g["q"].astype(np.float16), 10, ids=g["ids"])
    assert np.array_equal(p, g["exp_pos"])
    assert np.array_equal(i, g["exp_ids"])
    assert np.array_equal(s, g["exp_scores"])


def test_all_rows_identical_ties_by_position(native_lib):
    x = np.tile(np.arange(64, dtype=np.float16)[None, :] / 64, (1500, 1))
    q = np.ones((4, 64), dtype=np.float16)
    s, i, p = _search(x, q, 10)
    assert np.array_equal(p, np.tile(np.arange(10), (4, 1)))


def test_ascending_scores_force_list_overflow(native_lib):
    """Every later row beats every earlier one: each tile floods the candidate lists (refusal / retry path)."""
    n, d = 2000, 64  # n / 2048 is exact in fp16 for n <= 2048, so all scores are distinct
    u = np.zeros(d, dtype=np.float32)
    u[0] = 1.0
    x = (np.arange(1, n + 1, dtype=np.float32)[:, None] / 2048.0 * u[None, :]).astype(np.float16)
    q = np.tile(u[None, :], (3, 1)).astype(np.float16)
    q[1] *= -1  # descending for this query
    s, i, p = _search(x, q, 10)
    s_full = R.full_scores(q.astype(np.float32), x, R.DTYPE_F16)
    R.check_topk(s, p, s_full, 10, score_tol=0, tie_tol=0)
    assert np.array_equal(p[0], np.arange(n - 1, n - 11, -1))
    assert np.array_equal(p[1], np.arange(10))


def test_two_pass_large(native_lib):
    """Enough tiles (>= 8 per workgroup) to take the threshold-seeded two-pass schedule."""
    n, d, b, k = 600_000, 64, 32, 10
    x, q = _mk(n, d, b, seed=5)
    s, i, p = _search(x, q, k, id_base=1)
    s_full = R.full_scores(q.astype(np.float32), x, R.DTYPE_F16)
    R.check_topk(s, p, s_full, k, score_tol=SCORE_TOL, tie_tol=TIE_TOL)
    so, io, po = R.search(q.astype(np.float32), x, k, dtype=R.DTYPE_F16, id_base=1)
    assert R.recall_at_k(i, io) == 1.0


@pytest.mark.parametrize("k", [1, 10, 12, 30])
def test_two_stage_search_equals_one_stage(native_lib, monkeypatch, k):
    """Shards of at least 24 tiles per workgroup are searched in two stages (the first 10 % of the tiles give the main launch
    its thresholds; capi.hip plan_launch).  VQA_STAGE_MIN brings the switch-over down to a size the oracle handles: the
    result must be the oracle's and bit-identical to the one-stage search's (VQA_STAGE_MIN=0).  (Exact fp16 main launch:
    the int8 sketch pre-pass of large shards, tests/test_gpu_sketch.py, is switched off here.)"""
    set_option(monkeypatch, "VQA_SKETCH", "0")
    from vietnamese_qa_system_amd.index import DeviceIndex
    n, d, b = 300_001, 64, 41
    x, q = _mk(n, d, b, seed=11)
    x[1000:1040] = x[7]  # a run of ties that straddles nothing special in stage one ...
    x[250_000:250_040] = x[7]  # ... and the same rows again in the main stage
    s_full = R.full_scores(q.astype(np.float32), x, R.DTYPE_F16)
    out = []
    for stage_min in ("2", "0"):
        set_option(monkeypatch, "VQA_STAGE_MIN", stage_min)
        # (final_rescore = 0: the two-stage plan otherwise ends in the re-scoring arithmetic -- tests/test_gpu_determinism.py -- and the
        # one-stage plan of a small shard does not; this test is about the stages)
        ix = DeviceIndex(x, id_base=1, dtype="fp16", device=0, options={"final_rescore": 0})
        info = ix.launch_info(b, k)
        assert (info.first_stage_rows > 0) == (stage_min == "2" and k <= 12)
        assert info.rows_per_launch == n - info.first_stage_rows
        s, i, p = ix.search(torch.from_numpy(q).cuda(), k, return_positions=True)
        torch.cuda.synchronize()
        out.append((s.cpu().numpy(), p.cpu().numpy()))
        ix.close()
        R.check_topk(out[-1][0], out[-1][1], s_full, k, score_tol=SCORE_TOL, tie_tol=TIE_TOL)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_argument_errors(native_lib):
    from vietnamese_qa_system_amd.index import DeviceIndex
    x, q = _mk(100, 64, 2, seed=0)
    ix = DeviceIndex(x, dtype="fp16")
    with pytest.raises(ValueError):
        ix.search(torch.from_numpy(q).cuda(), 5000)
    with pytest.raises(ValueError):
        ix.search(torch.zeros(2, 32, dtype=torch.float16).cuda(), 5)
    with pytest.raises(ValueError):
        ix.search(torch.from_numpy(q).cuda().to(torch.int32), 5)
    ix.close()
    with pytest.raises(RuntimeError):
        ix.search(torch.from_numpy(q).cuda(), 5)


def test_merge_topk_matches_oracle(native_lib):
    from vietnamese_qa_system_amd.index import merge_topk
    rng = np.random.default_rng(3)
    r, b, k = 8, 50, 10
    sc = np.sort(rng.standard_normal((r, b, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    sc[:, :, 5:] = sc[:, :, 4:5]  # ties inside and across shards
    ids = rng.integers(0, 1 << 40, size=(r, b, k)).astype(np.int64)
    sc[3, :, 7:] = -np.inf
    ids[3, :, 7:] = -1  # a short shard
    es, ei = R.merge_shards(np.where(ids < 0, -np.inf, sc), ids, k)
    gs, gi = merge_topk(torch.from_numpy(sc).cuda(), torch.from_numpy(ids).cuda(), k)
    assert np.array_equal(gs.cpu().numpy(), es)
    assert np.array_equal(gi.cpu().numpy(), ei)


def test_sharded_searcher_packs_one_gather_buffer(native_lib):
    """The searcher's send/receive buffer layout ([ids int64 | scores f32] per rank) through the strided merge: fill the
    receive buffer as an all-gather over 3 ranks would and compare with the oracle's shard merge; world = 1 end to end."""
    from vietnamese_qa_system_amd.index import DeviceIndex, merge_topk
    from vietnamese_qa_system_amd.sharded import ShardedSearcher, sharded_index_searcher
    rng = np.random.default_rng(5)
    world, b, k = 3, 7, 5  # b * k odd: the per-rank block is padded to 8 bytes
    sc = np.sort(rng.standard_normal((world, b, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    ids = rng.integers(0, 1 << 40, size=(world, b, k)).astype(np.int64)
    s = ShardedSearcher(lambda *a: None, merge_topk)
    s.world = world
    send, recv, (send_s, send_i, recv_s, recv_i) = s._buffers(b, k, torch.device("cuda", 0))
    assert send.numel() % 8 == 0 and recv.shape == (world, send.numel())
    assert not recv_s.is_contiguous() and recv_s.shape == (world, b, k) and recv_i.shape == (world, b, k)
    for r in range(world):  # what all_gather_into_tensor delivers: rank r's send buffer in row r
        send_s.copy_(torch.from_numpy(sc[r]))
        send_i.copy_(torch.from_numpy(ids[r]))
        recv[r].copy_(send)
    gs, gi = merge_topk(recv_s, recv_i, k)
    es, ei = R.merge_shards(sc, ids, k)
    assert np.array_equal(gs.cpu().numpy(), es) and np.array_equal(gi.cpu().numpy(), ei)
    # world = 1 through a real index: results land in the send buffer and come back as copies
    x, q = _mk(3000, 64, 9, seed=21)
    ix = DeviceIndex(x, id_base=1, dtype="fp16", device=0)
    one = sharded_index_searcher(ix)
    s1, i1 = one.search(torch.from_numpy(q).cuda(), 10)
    s2, i2, _ = ix.search(torch.from_numpy(q).cuda(), 10)
    assert torch.equal(s1, s2) and torch.equal(i1, i2)
    piped = one.search_pipelined([torch.from_numpy(q).cuda(), torch.from_numpy(q[:4]).cuda()], 10)
    assert torch.equal(piped[0][0], s2) and torch.equal(piped[1][1], i2[:4])
    ix.close()


@pytest.mark.parametrize("n,d,b,k", [(5000, 128, 9, 30), (2000, 64, 3, 100), (40, 64, 2, 50), (70001, 64, 257, 13)])
def test_large_k_runs_continuation_passes(native_lib, n, d, b, k):
    """k > 12: ceil(k / 12) passes, each strictly below the last key of the previous one; exact, incl. exhaustion."""
    x, q = _mk(n, d, b, seed=k)
    s, i, p = _search(x, q, k, id_base=1)
    kk = min(k, n)
    R.check_topk(s[:, :kk], p[:, :kk], R.full_scores(q.astype(np.float32), x, R.DTYPE_F16), k, score_tol=SCORE_TOL,
                 tie_tol=TIE_TOL)
    assert np.array_equal(i[:, :kk], p[:, :kk] + 1)
    if kk < k:
        assert np.all(np.isneginf(s[:, kk:])) and np.all(i[:, kk:] == -1)


def test_large_k_clustered_rows_take_the_gated_fallback(native_lib, monkeypatch):
    """k > 12 first tries ONE pass (every workgroup keeps its local top 12) and verifies it on the device; when one
    workgroup owns more than 12 of the top-k the gated continuation passes must take over -- same exact result.  Here the
    40 best rows of every query sit in one 256-row tile.  The multi-pass path alone (VQA_WIDE_K=0) must agree too."""
    from vietnamese_qa_system_amd.index import DeviceIndex
    n, d, b, k = 70001, 64, 5, 40
    x, _ = _mk(n, d, b, seed=77)
    rng = np.random.default_rng(5)
    u = rng.standard_normal(d).astype(np.float32)
    u /= np.linalg.norm(u)
    q = R.l2_normalize(u + 0.03 * rng.standard_normal((b, d)).astype(np.float32)).astype(np.float16)  # queries around u
    x[256:316] = R.l2_normalize(u + 0.03 * rng.standard_normal((60, d)).astype(np.float32)).astype(np.float16)  # so are 60 rows of tile 1
    full = R.full_scores(q.astype(np.float32), x, R.DTYPE_F16)
    top = np.argsort(-full, axis=1)[:, :k]
    assert ((top >= 256) & (top < 316)).all()  # the whole top-40 of every query sits in one tile = one workgroup
    results = []
    for wide in ("1", "0"):
        set_option(monkeypatch, "VQA_WIDE_K", wide)
        ix = DeviceIndex(x, id_base=1, dtype="fp16", device=0)
        s, i, p = ix.search(torch.from_numpy(q).cuda(), k, return_positions=True)
        torch.cuda.synchronize()
        s, i, p = s.cpu().numpy(), i.cpu().numpy(), p.cpu().numpy()
        ix.close()
        R.check_topk(s, p, full, k, score_tol=SCORE_TOL, tie_tol=TIE_TOL)
        assert np.array_equal(i, p + 1)
        results.append((s, p))
    assert np.array_equal(results[0][1], results[1][1]) and np.array_equal(results[0][0], results[1][0])


def test_large_k_ties_across_pass_boundaries(native_lib, golden_dir):
    g = np.load(f"{golden_dir}/retr_ties.npz")
    x, q = g["x"].astype(np.float16), g["q"].astype(np.float16)
    s, i, p = _search(x, q, 40)
    full = R.full_scores(q.astype(np.float32), x, R.DTYPE_F16)
    order = np.argsort(-full, axis=1, kind="stable")[:, :40]  # score desc, position asc: exact integers, many duplicates
    assert np.array_equal(p, order)
    assert np.array_equal(s, np.take_along_axis(full, order, axis=1).astype(np.float32))

"""Worst-case inputs for the score bound of the int8 sketch search -- needs an MI355X; nothing under oracle/ is imported.

The sketch search (include/vqa_retrieval.h VQA_INDEX_SKETCH) may drop a (query, row) pair only when

    s_q s_t D + ||q_lo|| max||x_hi|| + ||q|| max||x_lo||  <  theta          (Cauchy-Schwarz on the two quantisation residues)

Random rows never come near the slack: their residue is uncorrelated with the query.  Here the residue of the planted rows is
ALIGNED with the query (every component's rounding error has the sign of the query's component, the query's components all
have one magnitude: equality in Cauchy-Schwarz), so their true score sits at the very top of what the bound allows, and their
scores are placed AT the thresholds the cascade derives (k-th largest exact seed, exact k-th best of the first stage).  A bound
that is too tight anywhere -- tile maxima, per-query constants, the fp32 threshold line of the scan -- drops such a row, and
then either the result differs from the float64 expectation or the candidate lists run short and the search overflows into
its exact fallback (``sketch_state() != 0``): both are asserted against.

* rotation off (VQA_SKETCH_ROTATE=0): rows on an exact power-of-two grid ``s (m + 0.4375 sign(q_j))``, representable in
  fp16; every product and partial sum is exact in fp32, so scores are bit-equal to float64 whatever the summation order and
  the expectation is strict (score desc, position asc).
* rotation on (VQA_SKETCH_CENTER=0): the same construction in the rotated space, mapped back through T^-1 = D H (restated here
  from the header's description: random signs from a fixed hash, normalised Walsh-Hadamard blocks); fp16 rounding of the rows
  perturbs the grid by ~1 % of a step, scores are compared with torch float64 of the stored values, near-tie aware.
* the device's own sketch of a tile (codes, scale, maxima: ``vqa_index_get_sketch_tile``) against a float64 restatement: the
  maxima must dominate the true residue norms of every row -- what the bound rests on.
* the exact fallback at k = 1, 2 on rows the bound cannot prune, distinct scores (thresholds and scan in different summation
  orders must not drop the row that defines the threshold), unfilled tiles, a B > 256 call whose first tile overflows, and a
  hipGraph-captured search replayed on data that overflows.
"""
import numpy as np
import pytest
import torch

from conftest import set_option

pytestmark = pytest.mark.gpu

S = 2.0 ** -10      # grid step of the planted tiles = their sketch scale
C = 2.0 ** -4       # magnitude of every query component
FRAC = 0.4375       # residue of every planted component, in steps, with the sign of the query's component


def _index(x, monkeypatch, sketch, env=None, dtype="fp16", **kw):
    from vietnamese_qa_system_amd.index import DeviceIndex
    set_option(monkeypatch, "VQA_STAGE_MIN", "2")
    for key, v in (env or {}).items():
        set_option(monkeypatch, key, v)
    return DeviceIndex(x, dtype=dtype, device=0, sketch=sketch, **kw)


def _search(ix, q, k):
    s, _, p = ix.search(torch.from_numpy(q).cuda(), k, return_positions=True)
    torch.cuda.synchronize()
    return s.cpu().numpy(), p.cpu().numpy()


def _fp64_topk(x, q, k):
    """(scores, positions) of the k best rows per query in float64: score desc, position asc."""
    s = torch.from_numpy(q).double() @ torch.from_numpy(x).double().T
    s = s.numpy()
    order = np.lexsort((np.broadcast_to(np.arange(s.shape[1]), s.shape), -s), axis=1)[:, :k]
    return np.take_along_axis(s, order, axis=1), order, s


def _sign_hash(d8):
    """D of T = H D (csrc/convert.hip sketch_rotate): the sign of element j from a fixed 32-bit mix of j."""
    j = np.arange(d8, dtype=np.uint64)
    h = (j + 0x9E3779B9) & 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return np.where(h & 1, -1.0, 1.0)


def _hadamard(n):
    h = np.ones((1, 1))
    while h.shape[0] < n:
        h = np.block([[h, h], [h, -h]])
    return h


def _rotation(d8):
    """T as a dense float64 matrix: per power-of-two block of the padded row (768 = 512 + 256), H / sqrt(B) after the signs."""
    t = np.zeros((d8, d8))
    off, rem = 0, d8
    while rem:
        b = 1 << (rem.bit_length() - 1)
        t[off:off + b, off:off + b] = _hadamard(b) / np.sqrt(b)
        off, rem = off + b, rem - b
    return t * _sign_hash(d8)[None, :]  # (T x)_i = sum_j H_ij D_j x_j


def _planted_corpus(rng, n, d, sigma):
    """Filler rows on the grid (integers in [-40, 40] steps; row 0 of every tile carries one component of 127 steps, so every
    tile's scale is exactly the step) and the integer amplitudes / scores of the planted rows.  Returns (grid rows in STEPS as
    float64 [n, d], dict position -> delta) -- the caller plants ``sigma * (100 + FRAC)`` with `delta` steps added on component 0."""
    y = rng.integers(-40, 41, size=(n, d)).astype(np.float64)
    y[::256, 0] = 127.0
    return y


def _plant(y, sigma, pos, delta, amp):
    y[pos] = sigma * (amp + FRAC)
    y[pos, 0] += sigma[0] * delta  # q . x moves by delta grid steps of C * S


# rows at theta + {2, 1, 0} steps in the first stage (tiles 0 .. 255: each in a 128-row half of its own, so each is a seed and the
# k-th largest seed IS the 10th best score), rows at theta + 1, theta, theta - 1 in the main stage and in the last tile
_FIRST = {5: 2, 300: 2, 1_000: 2, 20_000: 2, 2_700: 1, 33_000: 1, 60_000: 1, 40_100: 0, 50_000: 0, 65_500: 0, 7_000: -1, 64_001: -1}
_MAIN = {65_536: 1, 200_123: 1, 70_000: 0, 250_000: 0, 100_000: -1, 299_999: -1}


@pytest.mark.parametrize("k", [10, 12, 1])
def test_aligned_residues_at_the_thresholds_without_rotation(native_lib, monkeypatch, k):
    n, d, b = 300_000, 128, 8
    rng = np.random.default_rng(17)
    sigma = rng.choice([-1.0, 1.0], size=(b, d))
    y = _planted_corpus(rng, n, d, sigma[0])
    for pos, delta in {**_FIRST, **_MAIN}.items():
        _plant(y, sigma[0], pos, delta, 100.0)
    x = (y * S).astype(np.float16)
    assert np.array_equal(x.astype(np.float64), y * S), "the grid must be exactly representable in fp16"
    q = (sigma * C).astype(np.float16)
    exp_s, exp_p, full = _fp64_topk(x, q, k)
    assert np.array_equal(exp_s.astype(np.float32).astype(np.float64), exp_s), "scores must be exact in fp32"
    ske = _index(x, monkeypatch, sketch=True, env={"VQA_SKETCH_ROTATE": "0"})
    assert ske.launch_info(b, k).sketch_scan == 1
    # the planted tile's sketch is what the construction assumes: scale = the step, codes = the integer parts, residue 0.4375
    codes, info, _ = ske.sketch_tile(0)
    assert info[3] == np.float32(S) and info[2] == np.float32(1.0 / S)
    assert np.array_equal(codes[5].astype(np.float64), sigma[0] * 100.0 + np.where(np.arange(d) == 0, sigma[0, 0] * 2, 0.0))
    assert info[1] >= FRAC * S * np.sqrt(d) and info[1] <= FRAC * S * np.sqrt(d) * 1.0001
    s1, p1 = _search(ske, q, k)
    stats, state = ske.sketch_stats(), ske.sketch_state()
    s2, p2 = _search(ske, q, k)
    ske.close()
    assert stats["overflow"] == 0 and state == 0, (stats, state)
    assert np.array_equal(p1, exp_p), (p1[0], exp_p[0])
    assert np.array_equal(s1.astype(np.float64), exp_s)
    assert np.array_equal(p1, p2) and np.array_equal(s1, s2)
    if k == 10:  # query 0: 4 rows at +2, 5 at +1 (three of the first stage, two of the main stage), the earliest row at theta
        assert p1[0].tolist() == [5, 300, 1_000, 20_000, 2_700, 33_000, 60_000, 65_536, 200_123, 40_100]


@pytest.mark.parametrize("d,center", [(128, "0"), (256, "0"), (128, "1")])
def test_aligned_residues_at_the_thresholds_in_the_rotated_space(native_lib, monkeypatch, d, center):
    """The same planting in the space the sketch is cut from: y = T x on the grid, x = T^-1 y rounded to fp16.  With the centre on,
    mu (the mean of the strided sample of the first fill) moves every row off the grid by the same vector: alignment is lost for
    the rows but the bound's q . mu bookkeeping (threshold shift, its fp margin) is in play."""
    n, b, k, amp = 300_000, 8, 10, 50.0
    rng = np.random.default_rng(23 + d)
    sigma = rng.choice([-1.0, 1.0], size=(b, d))
    y = _planted_corpus(rng, n, d, sigma[0])
    for pos, delta in {**_FIRST, **_MAIN}.items():
        _plant(y, sigma[0], pos, 3 * delta, amp)  # 3 steps apart: clear of the fp16 rounding of the rows
    t = _rotation(d)
    x = ((y * S) @ t).astype(np.float16)      # x = T^-1 y = T' y (T orthogonal): row vectors, so y @ T
    q = ((sigma * C) @ t).astype(np.float16)
    # the stored rows, rotated back, sit on the grid to within a few % of a step
    back = x[[5, 40_100, 70_000]].astype(np.float64) @ t.T / S
    assert np.abs(back - y[[5, 40_100, 70_000]]).max() < 0.06
    exp_s, exp_p, full = _fp64_topk(x, q, k + 4)
    ske = _index(x, monkeypatch, sketch=True, env={"VQA_SKETCH_CENTER": center})
    assert ske.launch_info(b, k).sketch_scan == 1
    s1, p1 = _search(ske, q, k)
    stats, state = ske.sketch_stats(), ske.sketch_state()
    ske.close()
    assert stats["overflow"] == 0 and state == 0, (stats, state)
    assert np.abs(s1 - exp_s[:, :k]).max() <= 1e-5
    tol = 2e-6
    for row in range(b):  # every row whose fp64 score clears the returned k-th by more than the accumulation error must be returned
        kth = full[row, p1[row, k - 1]]
        must = set(np.nonzero(full[row] > kth + tol)[0].tolist())
        assert must <= set(p1[row].tolist()), (row, sorted(must - set(p1[row].tolist())))
        assert np.all(full[row, p1[row]] >= exp_s[row, k - 1] - tol)
    planted = {**_FIRST, **_MAIN}
    best = sorted(planted, key=lambda pos: (-planted[pos], pos))[:k]
    assert p1[0].tolist() == best


def _same_rows_outside_near_ties(s1, p1, ref, q, k, tol=1e-6):
    """rows of (s1, p1) against the exact scan `ref` asked for k + 1 rows: equal wherever a rank's score is further than 2 tol from
    both neighbours of the exact list (the k-th against the (k+1)-th too); scores within tol everywhere (fp16 shards: 1e-6 -- the
    products are exact, only the summation order differs; fp32 shards: the products round too)"""
    s0, p0 = _search(ref, q, k + 1)
    assert np.abs(s1 - s0[:, :k]).max() < tol
    gap = np.minimum(np.abs(np.diff(s0, axis=1, prepend=np.inf)), np.abs(np.diff(s0, axis=1, append=-np.inf)))[:, :k]
    clear = gap > 2 * tol
    assert clear.mean() > 0.8 and np.array_equal(p1[clear], p0[:, :k][clear]), (clear.mean(), int((p1 != p0[:, :k])[clear].sum()))


@pytest.mark.parametrize("env,d", [({"VQA_SKETCH_ROTATE": "0"}, 200), ({}, 768), ({"VQA_SKETCH_CENTER": "0"}, 96)])
def test_device_sketch_of_a_tile_dominates_the_true_residues(native_lib, monkeypatch, env, d):
    """codes, scale and the two maxima of a tile as the DEVICE holds them, against float64: with y = T (x - mu) (T, mu as the mode
    says), every row r of the tile must satisfy ||s c_r|| <= info[0] and ||y_r - s c_r|| <= info[1] -- then
    |q . x - q . mu - s_q s (q_int . c_r)| <= ||q_lo|| info[0] + ||q|| info[1] for every query, whatever its own quantisation."""
    n = 140_000
    rng = np.random.default_rng(d)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x[:, 3] *= 9.0  # an outlier dimension: what the rotation is for
    x += 0.4
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float16)
    ske = _index(x, monkeypatch, sketch=True, env=env)
    d8 = (d + 127) // 128 * 128
    rot = env.get("VQA_SKETCH_ROTATE", "1") != "0"
    t = _rotation(d8) if rot else np.eye(d8)
    for tile in (0, 17, (n - 1) // 256):
        codes, info, mu = ske.sketch_tile(tile)
        rows = np.zeros((256, d8))
        chunk = x[tile * 256:(tile + 1) * 256].astype(np.float64)
        rows[:chunk.shape[0], :d] = chunk
        centred = rot and env.get("VQA_SKETCH_CENTER", "1") != "0"
        assert centred == bool(np.any(mu != 0))
        valid = chunk.shape[0]
        # (the scale is the maximum over all 256 rows of the tile as stored: rows past the shard's end read as zeros, T (0 - mu);
        # codes and maxima exist for the shard's own rows only)
        yv = (rows - mu.astype(np.float64)) @ t.T
        s = float(info[3])
        assert abs(s - np.abs(yv).max() / 127.0) <= 1e-5 * s and info[2] == np.float32(1.0) / info[3]
        yv, codes = yv[:valid], codes[:valid]
        if rot:
            want = np.clip(np.rint(yv / s), -127, 127)
            diff = np.abs(codes.astype(np.float64) - want)
            assert diff.max() <= 1 and (diff > 0).mean() < 2e-3  # a code may differ where y / s is within fp32 rounding of a half
        else:  # no rotation: the device's own fp32 division, bit for bit
            want = np.clip(np.rint(yv.astype(np.float32) / info[3]), -127, 127)
            assert np.array_equal(codes.astype(np.float32), want)
        hi = np.linalg.norm(s * codes.astype(np.float64), axis=1).max()
        lo = np.linalg.norm(yv - s * codes.astype(np.float64), axis=1).max()
        slack = 14 * 2.0 ** -24 * np.linalg.norm(yv, axis=1).max() if rot else 0.0  # the rotation's own fp32 rounding (joins the margin)
        assert hi <= info[0] <= hi * 1.0001 + 1e-12, (hi, info[0])
        assert lo - slack <= info[1] <= (lo + slack) * 1.0001 + 1e-12, (lo, info[1])
        # the split of the slack term |z . x_lo| along w = T mu / ||T mu||: the tile's max |w . x_lo| must dominate the float64 value
        # (any w and any C >= max |w . x_lo| keep the bound valid: |z . x_lo| <= |alpha| C + ||z - alpha w|| ||x_lo|| for every alpha)
        c, w, _, per_row = ske.sketch_split(tile)
        assert not per_row
        if centred:
            tw = t @ mu.astype(np.float64)
            assert np.abs(w - tw / np.linalg.norm(tw)).max() < 1e-5 and abs(np.linalg.norm(w.astype(np.float64)) - 1.0) < 1e-5
            cw = np.abs((yv - s * codes.astype(np.float64)) @ w.astype(np.float64)).max()
            assert cw - slack <= c <= (cw + slack) * 1.001 + 2e-4 * lo, (cw, c)
            assert c < 0.5 * info[1]  # a quantisation residue is nearly orthogonal to a fixed direction: that is the point of the split
        else:
            assert c == 0.0 and not np.any(w)
    ske.close()


def test_split_slack_term_prunes_collapsed_embeddings_and_changes_no_result(native_lib, monkeypatch):
    """Embeddings that share one large common component (mean cosine 0.9: what an untrained encoder emits): the queries are mostly
    alpha w, so splitting the slack term |z . x_lo| along w removes most of ||q|| ||x_lo||.  Same rows and scores as without the
    split and as the exact scan; fewer candidate pairs."""
    n, d, b, k = 300_000, 768, 64, 10
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    c = torch.randn((1, d), generator=g, device="cuda")
    c /= c.norm()

    def draw(m):
        v = torch.randn((m, d), generator=g, device="cuda")
        v = 3.0 * c + v / v.norm(dim=1, keepdim=True)
        return (v / v.norm(dim=1, keepdim=True)).half().cpu().numpy()

    x, q = draw(n), draw(b)
    pairs = {}
    res = {}
    for name, env in (("split", {}), ("plain", {"VQA_SKETCH_SPLIT": "0"})):
        ske = _index(x, monkeypatch, sketch=True, env=env)
        res[name] = _search(ske, q.astype(np.float32), k)
        st = ske.sketch_stats()
        assert st["overflow"] == 0 and ske.sketch_state() == 0, (name, st)
        pairs[name] = st["rescored_pairs"]
        ske.close()
        set_option(monkeypatch, "VQA_SKETCH_SPLIT", None)
    assert np.array_equal(res["split"][0], res["plain"][0]) and np.array_equal(res["split"][1], res["plain"][1])
    ref = _index(x, monkeypatch, sketch=False)
    _same_rows_outside_near_ties(res["split"][0], res["split"][1], ref, q.astype(np.float32), k)  # re-scoring vs MFMA summation order
    ref.close()
    assert pairs["split"] < 0.5 * pairs["plain"], pairs


@pytest.mark.parametrize("k", [1, 2, 10])
def test_small_k_fallback_keeps_the_row_that_defines_theta(native_lib, monkeypatch, k):
    """Rows the bound cannot prune (200 000 distinct near-duplicates of the query, all within the slack) with DISTINCT scores whose
    best rows sit in the first stage: the cascade overflows and its gated exact pass runs.  Its threshold must come from the
    scan's own arithmetic: a threshold summed in the re-scoring kernel's order could exceed the MFMA sum of the very row that
    defines it by an ulp and leave fewer than k results (k = 1: (-inf, -1) for a query with a real best row)."""
    n, d, b = 200_000, 192, 4   # 50 000 near-duplicates per query: its 16 sub-lists of 2048 candidates overflow
    rng = np.random.default_rng(99)
    v = rng.standard_normal((b, d)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    owner = rng.integers(0, b, n)
    x = v[owner] + 0.002 * rng.standard_normal((n, d)).astype(np.float32)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float16)
    q = v.astype(np.float16)
    for row in range(b):  # the best row of every query: inside the first stage, not a duplicate of anything
        pos = 1_000 + 1_500 * row
        x[pos] = q[row]
    ref = _index(x, monkeypatch, sketch=False)
    ske = _index(x, monkeypatch, sketch=True)
    s0, p0 = _search(ref, q, k)
    s1, p1 = _search(ske, q, k)
    stats = ske.sketch_stats()
    ref.close()
    ske.close()
    assert stats["overflow"] == 1, "this input is meant to overflow the candidate buffers"
    assert np.all(p1 >= 0) and np.all(np.isfinite(s1)), "the fallback returned fewer than k rows"
    assert np.array_equal(p1, p0) and np.array_equal(s1, s0)  # the exact scan's bits
    assert p1[:, 0].tolist() == [1_000 + 1_500 * row for row in range(b)]


def test_unfilled_tiles_and_a_partly_filled_shard(native_lib, monkeypatch):
    """A shard created empty and filled only in part: tiles no call has touched keep an all-zero tile_info, every pair of theirs is
    a candidate, the search overflows and falls back -- the exact scan's result over rows that read as zeros."""
    from vietnamese_qa_system_amd.index import DeviceIndex
    n, d, b, k = 300_000, 64, 32, 10
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float16)
    q = rng.standard_normal((b, d)).astype(np.float32)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float16)
    set_option(monkeypatch, "VQA_STAGE_MIN", "2")
    set_option(monkeypatch, "VQA_SKETCH_COOLDOWN", "0")  # no pause after the overflow: the search behind the second fill is a sketch search
    filled = 200_100
    ske = DeviceIndex.empty(n, d, dtype="fp16", device=0, sketch=True)
    ske.set_rows(0, x[:filled])
    x_now = x.copy()
    x_now[filled:] = 0
    exp_s, exp_p, full = _fp64_topk(x_now, q, k)
    s1, p1 = _search(ske, q, k)
    assert ske.sketch_stats()["overflow"] == 1
    ske.set_rows(filled, x[filled:])  # the rest arrives: now a plain sketch search
    s2, p2 = _search(ske, q, k)
    stats = ske.sketch_stats()
    ske.close()
    assert np.abs(s1 - exp_s).max() <= 1e-5 and np.array_equal(p1, exp_p)
    exp_s2, exp_p2, _ = _fp64_topk(x, q, k)
    assert np.abs(s2 - exp_s2).max() <= 1e-5 and np.array_equal(p2, exp_p2) and stats["overflow"] == 0


def test_overflow_in_the_first_query_tile_of_a_long_batch_is_reported(native_lib, monkeypatch):
    """B = 300 runs as two query tiles of 150 (capi.hip search_impl: tiles of equal size): the first (150 copies of one row: every pair a
    candidate) overflows, the last (150 random queries) does not.  The overflow of the EARLIER tile must still reach the host (the pinned
    mirror carries the OR over the call's tiles), and every query gets the exact scan's rows."""
    n, d, k = 180_000, 64, 10
    rng = np.random.default_rng(8)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float16)
    # 100 000 duplicates of one row, all behind the first stage (the cascade's first 65 536 rows): queries equal to that row cannot
    # be pruned; for the others the first stage's k-th best score keeps the duplicates out (duplicates INSIDE the first stage would
    # tie with every query's seed threshold and overflow every list)
    x[70_000:170_000] = x[7]
    q = rng.standard_normal((300, d)).astype(np.float32)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float16)
    t0 = 150  # queries of the first tile
    q[:t0] = x[7]
    ref = _index(x, monkeypatch, sketch=False)
    ske = _index(x, monkeypatch, sketch=True)
    s0, p0 = _search(ref, q, k)
    assert ske.sketch_state() == 0
    s1, p1 = _search(ske, q, k)
    stats, state = ske.sketch_stats(), ske.sketch_state()
    s2, p2 = _search(ske, q, k)   # inside the pause: the exact scan for every tile
    ref.close()
    ske.close()
    assert stats["overflow"] == 0 and stats["overflow_earlier_tiles"] == 1 and state > 0, (stats, state)
    assert np.array_equal(p1[:t0], p0[:t0]) and np.array_equal(s1[:t0], s0[:t0])
    assert np.abs(s1[t0:] - s0[t0:]).max() <= 3e-7 and np.array_equal(p1[t0:], p0[t0:])
    assert np.array_equal(p2, p0) and np.array_equal(s2, s0)


def test_pause_doubles_while_the_data_keeps_overflowing(native_lib, monkeypatch):
    """Data the bound never prunes: overflow -> pause of VQA_SKETCH_COOLDOWN searches on the exact scan -> one more sketch search ->
    overflow again -> a pause twice as long.  Results never change."""
    n, d, b, k = 150_000, 64, 16, 5
    rng = np.random.default_rng(5)
    v = rng.standard_normal((1, d)).astype(np.float32)
    v = (v / np.linalg.norm(v)).astype(np.float16)
    x = np.repeat(v, n, axis=0)
    x[123] = -v[0]
    q = np.repeat(v, b, axis=0)
    ske = _index(x, monkeypatch, sketch=True, env={"VQA_SKETCH_COOLDOWN": "3"})
    states, first = [], None
    for _ in range(14):
        s, p = _search(ske, q, k)
        first = (s, p) if first is None else first
        assert np.array_equal(p, first[1]) and np.array_equal(s, first[0])
        states.append(ske.sketch_state())
    ske.close()
    assert first[1][0].tolist() == [0, 1, 2, 3, 4]
    # search 1 overflows (state: a pause of 3 pending); searches 2-4 run the exact scan; search 5 is a sketch search again and
    # overflows: the next pause is 6 searches long (6-11), search 12 tries again, then 12
    # (the search that finds the report starts the pause, the following ones count it down; the one that reaches 0 is a sketch search)
    assert states[:12] == [3, 3, 2, 1, 6, 6, 5, 4, 3, 2, 1, 12], states


def test_captured_search_replays_on_data_that_overflows(native_lib, monkeypatch):
    """A search captured into a hipGraph replays the launches the host chose at capture time (include/vqa_retrieval.h): here the
    sketch search WITH its gated exact fallback behind it.  Replayed on queries the bound prunes and on queries it cannot prune
    (the overflow happens inside the replay, no host code runs), the outputs are those of an eager search."""
    n, d, b, k = 200_000, 64, 64, 10
    rng = np.random.default_rng(13)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float16)
    x[50_000:190_000] = x[9]
    q_easy = rng.standard_normal((b, d)).astype(np.float32)
    q_easy = (q_easy / np.linalg.norm(q_easy, axis=1, keepdims=True)).astype(np.float16)
    q_hard = q_easy.copy()
    q_hard[:8] = x[9]
    ref = _index(x, monkeypatch, sketch=False)
    ske = _index(x, monkeypatch, sketch=True)
    want = {name: _search(ref, qq, k) for name, qq in (("easy", q_easy), ("hard", q_hard))}
    ref.close()
    dev = torch.device("cuda", 0)
    q_buf = torch.from_numpy(q_easy).to(dev)
    out_s = torch.empty((b, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((b, k), dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        ske.search(q_buf, k, out=(out_s, out_i))  # warm-up outside the capture: one-time kernel attributes
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        ske.search(q_buf, k, out=(out_s, out_i))
    for name, qq in (("easy", q_easy), ("hard", q_hard), ("easy", q_easy), ("hard", q_hard)):
        q_buf.copy_(torch.from_numpy(qq))
        out_s.fill_(-1.0)
        graph.replay()
        torch.cuda.synchronize()
        s, i = out_s.cpu().numpy(), out_i.cpu().numpy()
        s0, p0 = want[name]
        assert np.array_equal(i, p0), name          # ids = positions (id_base 0)
        assert np.abs(s - s0).max() <= 3e-7, name
    ske.close()


@pytest.mark.parametrize("pattern", ["0xCB", "0x7F", "0xFF"])
def test_dirty_workspaces_and_a_partial_query_tile(native_lib, monkeypatch, pattern):
    """hipMalloc hands a long-lived process recycled, dirty memory.  VQA_POISON_WORKSPACE fills every workspace a search must write
    before it reads with a byte pattern (0xCB...: large negative floats, 0x7F7F...: 3.4e38, 0xFF...: NaN): a batch of 24 queries --
    232 padded query rows whose per-query constants nobody writes -- and batches across the tile limit must still take the sketch
    search (round 4: a negative garbage 1 / s_q turned the padded queries' +inf threshold into -inf and flooded every region)."""
    n, d, k = 200_000, 64, 10
    rng = np.random.default_rng(44)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float16)
    ref = _index(x, monkeypatch, sketch=False)
    ske = _index(x, monkeypatch, sketch=True, env={"VQA_POISON_WORKSPACE": pattern})
    for b in (24, 1, 256, 300, 13):
        q = rng.standard_normal((b, d)).astype(np.float32)
        q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float16)
        s0, p0 = _search(ref, q, k)
        s1, p1 = _search(ske, q, k)
        st, state = ske.sketch_stats(), ske.sketch_state()
        assert st["overflow"] == 0 and st["overflow_earlier_tiles"] == 0 and state == 0, (b, st, state)
        assert np.array_equal(p1, p0) and np.abs(s1 - s0).max() <= 3e-7, b
    ref.close()
    ske.close()


def _collapsed(n, b, d, weight, seed):
    """unit rows sharing one large common component: mean cosine weight^2 / (1 + weight^2) (3.0: 0.9)"""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    c = torch.randn((1, d), generator=g, device="cuda")
    c /= c.norm()

    def draw(m):
        v = torch.randn((m, d), generator=g, device="cuda")
        v = weight * c + v / v.norm(dim=1, keepdim=True)
        return (v / v.norm(dim=1, keepdim=True)).half().cpu().numpy()

    return draw(n), draw(b)


@pytest.mark.parametrize("pattern", ["", "0xCB"])
def test_per_row_form_on_collapsed_embeddings(native_lib, monkeypatch, pattern):
    """Rows collapsed onto their centre direction (mean cosine 0.9 >= 0.6) switch the shard to the per-row form at its first fill: rows
    and queries are projected off w before they are sketched, the scan adds alpha beta per (query, row).  (a) what the device keeps
    for a tile, against float64: beta = w . y, the codes / maxima are those of y - beta w, the tile's c >= max |beta| -- with these the
    bound follows for every query by z . y = alpha (w . y) + z_r . y, z_r . y = beta (z_r . w) + z_r . y_r; (b) the search returns the
    exact scan's rows with fewer candidates than the split form; 300 queries = a second, partial query tile; poisoned workspaces."""
    n, d, b, k = 300_000, 768, 300, 10
    x, q = _collapsed(n, b, d, 3.0, 91)
    if pattern:
        set_option(monkeypatch, "VQA_POISON_WORKSPACE", pattern)
    ske = _index(x, monkeypatch, sketch=True)
    t = _rotation(d)
    for tile in (0, 500, (n - 1) // 256):
        codes, info, mu = ske.sketch_tile(tile)
        c, w, beta, per_row = ske.sketch_split(tile)
        assert per_row
        rows = x[tile * 256:(tile + 1) * 256].astype(np.float64)
        valid = rows.shape[0]
        y = (rows - mu.astype(np.float64)) @ t.T
        w64 = w.astype(np.float64)
        assert np.abs(beta[:valid] - y @ w64).max() < 2e-6 and c >= np.abs(beta[:valid]).max()
        yr = y - beta[:valid, None].astype(np.float64) * w64[None, :]  # with the DEVICE's beta: the identity y = beta w + y_r is what counts
        s = float(info[3])
        sc = s * codes[:valid].astype(np.float64)
        slack = 14 * 2.0 ** -24 * np.linalg.norm(y, axis=1).max()
        assert np.linalg.norm(sc, axis=1).max() <= info[0] * (1 + 1e-6)
        assert np.linalg.norm(yr - sc, axis=1).max() <= info[1] + slack
        assert np.abs(codes).max() <= 127 and np.linalg.norm(yr, axis=1).mean() < 0.5  # a third of the unit norm is left to sketch
    s1, p1 = _search(ske, q.astype(np.float32), k)
    st = ske.sketch_stats()
    assert st["overflow"] == 0 and st["overflow_earlier_tiles"] == 0 and ske.sketch_state() == 0, st
    pairs = st["rescored_pairs"]
    ske.close()
    set_option(monkeypatch, "VQA_POISON_WORKSPACE", None)
    split = _index(x, monkeypatch, sketch=True, env={"VQA_SKETCH_PER_ROW": "0"})
    assert not split.sketch_split(0)[3]
    s2, p2 = _search(split, q.astype(np.float32), k)
    pairs_split = split.sketch_stats()["rescored_pairs"]
    split.close()
    set_option(monkeypatch, "VQA_SKETCH_PER_ROW", None)
    assert np.array_equal(s1, s2) and np.array_equal(p1, p2)  # both score the survivors with the same kernel
    assert pairs < 0.6 * pairs_split, (pairs, pairs_split)
    ref = _index(x, monkeypatch, sketch=False)
    _same_rows_outside_near_ties(s1, p1, ref, q.astype(np.float32), k)
    ref.close()


def test_per_row_form_needs_six_k_steps(native_lib, monkeypatch):
    """rows of fewer than six 64-element K-steps keep the split form however collapsed they are (the tile's betas land in LDS behind
    the counted waits of K-steps 1-4)"""
    n, d, b, k = 200_000, 256, 64, 10
    x, q = _collapsed(n, b, d, 3.0, 92)
    ske = _index(x, monkeypatch, sketch=True, env={"VQA_SKETCH_PER_ROW": "1"})
    assert not ske.sketch_split(0)[3]
    s1, p1 = _search(ske, q.astype(np.float32), k)
    assert ske.sketch_stats()["overflow"] == 0
    ske.close()
    set_option(monkeypatch, "VQA_SKETCH_PER_ROW", None)
    ref = _index(x, monkeypatch, sketch=False)
    _same_rows_outside_near_ties(s1, p1, ref, q.astype(np.float32), k)
    ref.close()


def test_a_search_that_scores_too_many_pairs_pauses_the_sketch_without_overflowing(native_lib, monkeypatch):
    """100 tight clusters in a shard of natural sketch size (1.1M rows: no VQA_STAGE_MIN): every query has 11 000 near-duplicates.  No
    candidate buffer fills up -- the search stands, exact -- but it scores 2.8M pairs exactly where the exact scan would have been
    cheaper: the handle reports the pause an overflow would have started (VQA_SKETCH_PROFIT=0: stays on the sketch)."""
    from vietnamese_qa_system_amd.index import DeviceIndex
    n, d, b, k = 1_100_000, 64, 256, 10
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    centres = torch.randn((100, d), generator=g, device="cuda")
    centres /= centres.norm(dim=1, keepdim=True)

    def draw(m):
        v = centres[torch.randint(0, 100, (m,), generator=g, device="cuda")] + 0.02 * torch.randn((m, d), generator=g, device="cuda") / d ** 0.5
        return (v / v.norm(dim=1, keepdim=True)).half()

    x, q = draw(n), draw(b)
    set_option(monkeypatch, "VQA_STAGE_MIN", None)
    ref = DeviceIndex(x, dtype="fp16", device=0, sketch=False)
    s0, _, p0 = ref.search(q, k, return_positions=True)
    ref.close()
    for profit, want_pause in (("", True), ("0", False)):
        if profit:
            set_option(monkeypatch, "VQA_SKETCH_PROFIT", profit)
        ske = DeviceIndex(x, dtype="fp16", device=0, sketch=True)
        assert ske.launch_info(b, k).sketch_scan == 1
        s1, _, p1 = ske.search(q, k, return_positions=True)
        torch.cuda.synchronize()
        st = ske.sketch_stats()
        assert st["overflow"] == 0 and st["rescored_pairs"] > 0.4 * n, st
        assert (ske.sketch_state() > 0) == want_pause, (profit, ske.sketch_state())
        assert (s1 - s0).abs().max().item() < 1e-6
        s2, _, p2 = ske.search(q, k, return_positions=True)  # inside the pause: the exact scan
        torch.cuda.synchronize()
        if want_pause:
            # (both exact scans end in the re-scoring arithmetic: the same bits -- tests/test_gpu_determinism.py; against the SKETCH search the
            # 11 000 near-duplicates per query are more near-ties of the k-th row than the k + 2 rows an exact scan re-ranks can cover)
            assert torch.equal(s2, s0) and torch.equal(p2, p0) and (s2 - s1).abs().max().item() < 1e-6
        else:
            assert torch.equal(s2, s1) and torch.equal(p2, p1)
        ske.close()
        set_option(monkeypatch, "VQA_SKETCH_PROFIT", None)


def test_per_row_form_fp32_shard_filled_in_unaligned_chunks(native_lib, monkeypatch):
    """The per-row form on an fp32 shard filled by four calls that start and end inside tiles (the first call fixes the centre and the
    form): every tile's betas, codes and maxima are those of its rows as stored, whichever call wrote them; same rows as the exact scan."""
    from vietnamese_qa_system_amd.index import DeviceIndex
    n, d, b, k = 150_003, 768, 32, 10
    x, q = _collapsed(n, b, d, 3.0, 93)
    x, q = x.astype(np.float32), q.astype(np.float32)
    set_option(monkeypatch, "VQA_STAGE_MIN", "2")
    ske = DeviceIndex.empty(n, d, dtype="fp32", device=0, sketch=True)
    for lo, hi in ((100_000, 150_003), (0, 777), (777, 40_001), (40_001, 100_000)):
        ske.set_rows(lo, x[lo:hi])
    assert ske.launch_info(b, k).sketch_scan == 1
    t = _rotation(d)
    for tile in (3, 156, 390, (n - 1) // 256):  # 3, 156 and 390 hold rows of two calls
        codes, info, mu = ske.sketch_tile(tile)
        c, w, beta, per_row = ske.sketch_split(tile)
        assert per_row
        rows = x[tile * 256:(tile + 1) * 256].astype(np.float64)
        valid = rows.shape[0]
        y = (rows - mu.astype(np.float64)) @ t.T
        w64 = w.astype(np.float64)
        assert np.abs(beta[:valid] - y @ w64).max() < 2e-6 and c >= np.abs(beta[:valid]).max()
        yr = y - beta[:valid, None].astype(np.float64) * w64[None, :]
        sc = float(info[3]) * codes[:valid].astype(np.float64)
        slack = 14 * 2.0 ** -24 * np.linalg.norm(y, axis=1).max()
        assert np.linalg.norm(sc, axis=1).max() <= info[0] * (1 + 1e-6) and np.linalg.norm(yr - sc, axis=1).max() <= info[1] + slack
    s1, p1 = _search(ske, q, k)
    assert ske.sketch_stats()["overflow"] == 0 and ske.sketch_state() == 0
    ske.close()
    ref = _index(x, monkeypatch, sketch=False, dtype="fp32")
    _same_rows_outside_near_ties(s1, p1, ref, q, k, tol=3e-6)
    ref.close()


def test_a_pause_started_by_a_large_k_leaves_small_k_on_the_sketch(native_lib, monkeypatch):
    """1.2M x 768 rows, natural sketch size: 256 queries at k = 128 score more pairs exactly than the exact forms cost -> the handle
    pauses the sketch, but only for searches of k >= 64: a k = 10 search inside the pause still runs on the sketch (its candidate
    counts replace those of the k = 128 search), a k = 64 search takes the exact scan (the counts stay)."""
    from vietnamese_qa_system_amd.index import DeviceIndex
    n, d, b = 1_200_000, 768, 256
    g = torch.Generator(device="cuda")
    g.manual_seed(17)
    x = torch.cat([torch.randn((1 << 18, d), generator=g, device="cuda") for _ in range(0, n, 1 << 18)])[:n]
    x = (x / x.norm(dim=1, keepdim=True)).half()
    q = torch.randn((b, d), generator=g, device="cuda")
    q = (q / q.norm(dim=1, keepdim=True)).half()
    set_option(monkeypatch, "VQA_STAGE_MIN", None)
    ske = DeviceIndex(x, dtype="fp16", device=0, sketch=True)
    ref = DeviceIndex(x, dtype="fp16", device=0, sketch=False)
    ske.search(q, 128)
    torch.cuda.synchronize()
    wide = ske.sketch_stats()
    assert wide["overflow"] == 0 and wide["rescored_pairs"] > 0.75 * n - 4e5 and ske.sketch_state() > 0, (wide, ske.sketch_state())
    s64, _, p64 = ske.search(q, 64, return_positions=True)  # the pause applies: exact scan, the candidate buffers are not touched
    torch.cuda.synchronize()
    assert ske.sketch_stats()["rescored_pairs"] == wide["rescored_pairs"]
    e64, _, ep64 = ref.search(q, 64, return_positions=True)
    assert torch.equal(s64, e64) and torch.equal(p64, ep64)
    s10, _, p10 = ske.search(q, 10, return_positions=True)  # k < 64: still the sketch search
    torch.cuda.synchronize()
    narrow = ske.sketch_stats()
    assert 0 < narrow["rescored_pairs"] < 0.5 * wide["rescored_pairs"] and narrow["overflow"] == 0, (narrow, wide)
    e10, _, ep10 = ref.search(q, 10, return_positions=True)
    assert (s10 - e10).abs().max().item() < 1e-6 and ske.sketch_state() > 0
    ske.close()
    ref.close()

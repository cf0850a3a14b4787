"""The int8 sketch pre-pass of large fp16 shards (include/vqa_retrieval.h VQA_INDEX_SKETCH) -- needs an MI355X.

A shard large enough for the two-stage search runs its main launch over an int8 sketch of the rows (v_mfma_i32_16x16x64_i8:
half the bytes, twice the matrix rate per row) with a RIGOROUS upper bound on every (query, row) score, and scores exactly
-- fp16 rows, fp32 accumulation -- only the pairs the bound cannot exclude.  The results must be those of the exact scan:
the oracle's rows (tie-aware), the exact scan's rows position for position, scores within the last bits of fp32, exact
duplicates bit-equal and in position order whichever stage they sit in.  VQA_STAGE_MIN brings the switch-over down to sizes
the oracle handles (131 072 rows on 256 compute units)."""
import numpy as np
import pytest
import torch

from conftest import set_option

from oracle import retrieval as R

pytestmark = pytest.mark.gpu
SCORE_TOL, TIE_TOL = 1e-5, 2e-6


def _unit(rng, n, d):
    return R.l2_normalize(rng.standard_normal((n, d)).astype(np.float32)).astype(np.float16)


def _index(x, monkeypatch, sketch, stage_min="2", ids=None, id_base=0, dtype="fp16", rescore_copy=None):
    from vietnamese_qa_system_amd.index import DeviceIndex
    set_option(monkeypatch, "VQA_STAGE_MIN", stage_min)
    return DeviceIndex(x, ids=ids, id_base=id_base, dtype=dtype, device=0, sketch=sketch, rescore_copy=rescore_copy)


def _search(ix, q, k):
    s, i, p = ix.search(torch.from_numpy(q).cuda(), k, return_positions=True)
    torch.cuda.synchronize()
    return s.cpu().numpy(), i.cpu().numpy(), p.cpu().numpy()


@pytest.mark.parametrize("n,d,b,k", [(300_001, 64, 41, 10), (200_000, 100, 256, 12), (150_000, 768, 64, 1), (262_144, 128, 7, 10),
                                     (200_000, 100, 257, 10), (150_000, 768, 300, 10)])  # (two query tiles of equal size: 129 + 128, 150 + 150)
def test_sketch_search_equals_the_exact_scan(native_lib, monkeypatch, n, d, b, k):
    rng = np.random.default_rng(n + d)
    x, q = _unit(rng, n, d), _unit(rng, b, d)
    dup = [3, 40_000, 65_535, 65_536, n - 1]  # exact duplicates in the first stage, astride its end (256 tiles) and in the last tile
    for r in dup[1:]:
        x[r] = x[dup[0]]
    q[0] = x[dup[0]]
    ids = np.arange(n, dtype=np.int64) * 5 + 9
    ref = _index(x, monkeypatch, sketch=False, ids=ids)
    ske = _index(x, monkeypatch, sketch=True, ids=ids)
    assert ref.launch_info(b, k).sketch_scan == 0 and ske.launch_info(b, k).sketch_scan == 1
    assert ske.launch_info(b, k).first_stage_rows == 256 * 256
    assert ske.launch_info(b, k).bytes_per_launch == (n - 65536) * d  # one byte per element
    s0, i0, p0 = _search(ref, q, k)
    s1, i1, p1 = _search(ske, q, k)
    s2, _, p2 = _search(ske, q, k)  # the handle's candidate buffers are reused: same bits again
    ref.close()
    ske.close()
    assert np.array_equal(p1, p2) and np.array_equal(s1, s2)
    s_full = R.full_scores(q.astype(np.float32), x, R.DTYPE_F16)
    R.check_topk(s1, p1, s_full, k, score_tol=SCORE_TOL, tie_tol=TIE_TOL)
    assert np.array_equal(p1, p0), "the sketch search returns other rows than the exact scan"
    assert np.array_equal(i1, ids[p1]) and np.abs(s1 - s0).max() <= 3e-7
    nd = min(k, len(dup))
    assert p1[0, :nd].tolist() == dup[:nd] and len(set(s1[0, :nd].tolist())) == 1  # bit-equal scores, position order


@pytest.mark.parametrize("n,d,b,k", [(300_001, 64, 41, 30), (200_000, 128, 256, 13), (180_000, 768, 32, 32), (250_000, 96, 17, 64),
                                     (320_000, 128, 33, 100), (200_000, 64, 256, 128)])
def test_wide_sketch_search_equals_the_exact_scan(native_lib, monkeypatch, n, d, b, k):
    """12 < k <= 128 (the dense leg of a hybrid search asks for 10 x limit rows) take the same cascade as k <= 12: exact seeds ->
    sketch scan of the first stage against their threshold -> exact k-th best of its survivors -> sketch scan of the rest.  Same
    rows as the exact large-k path, duplicates (inside the first stage, astride its end, in the last tile) bit-equal in
    position order."""
    rng = np.random.default_rng(n + d + k)
    x, q = _unit(rng, n, d), _unit(rng, b, d)
    dup = [3, 40_000, 65_535, 65_536, n - 1]
    for r in dup[1:]:
        x[r] = x[dup[0]]
    q[0] = x[dup[0]]
    ids = np.arange(n, dtype=np.int64) * 3 + 1
    ref = _index(x, monkeypatch, sketch=False, ids=ids)
    ske = _index(x, monkeypatch, sketch=True, ids=ids)
    li = ske.launch_info(b, k)
    assert ref.launch_info(b, k).sketch_scan == 0 and li.sketch_scan == 1 and ske.launch_info(b, 129).sketch_scan == 0
    # k >= 16: a second cascade stage of twice the first one's tiles stands in front of the main launch
    staged = 65536 * (3 if k >= 16 and (n + 255) // 256 >= 4 * 256 else 1)  # (where the shard leaves the main launch a tile per workgroup)
    assert li.first_stage_rows == staged and li.rows_per_launch == n - staged and li.bytes_per_launch == (n - staged) * d
    s0, i0, p0 = _search(ref, q, k)
    s1, i1, p1 = _search(ske, q, k)
    s2, _, p2 = _search(ske, q, k)
    s3, _, p3 = _search(ske, q, 10)  # the narrow form on the same handle, then the wide one again
    s4, _, p4 = _search(ske, q, k)
    ref.close()
    ske.close()
    assert np.array_equal(p1, p2) and np.array_equal(s1, s2) and np.array_equal(p1, p4) and np.array_equal(s1, s4)
    assert np.array_equal(p3, p1[:, :10]) and np.array_equal(s3, s1[:, :10])
    s_full = R.full_scores(q.astype(np.float32), x, R.DTYPE_F16)
    R.check_topk(s1, p1, s_full, k, score_tol=SCORE_TOL, tie_tol=TIE_TOL)
    # (32 768 results at k = 128: rows may swap ranks with the exact scan's inside near-tie groups -- the two paths add up a score in
    # different orders --, never elsewhere)
    diff = p1 != p0
    if k <= 64:  # these shapes return the exact scan's rows in the exact scan's order: the strict form stays their regression guard
        assert not diff.any(), "the wide sketch search returns other rows than the exact scan"
    assert np.all(np.abs(s1[diff] - s0[diff]) <= TIE_TOL) and diff.mean() < 1e-3, "the wide sketch search returns other rows than the exact scan"
    assert np.array_equal(i1, ids[p1]) and np.abs(s1 - s0).max() <= 3e-7
    assert p1[0, :5].tolist() == dup and len(set(s1[0, :5].tolist())) == 1


def test_wide_sketch_overflow_takes_the_exact_passes(native_lib, monkeypatch):
    """k = 30 on rows the bound cannot prune: the gated exact passes (three of them) overwrite the result -- the exact path's bits."""
    n, d, b, k = 200_000, 64, 24, 30
    rng = np.random.default_rng(2)
    v = _unit(rng, 1, d)
    x = np.repeat(v, n, axis=0)
    x[5] = _unit(rng, 1, d)[0]
    q = np.repeat(v, b, axis=0)
    set_option(monkeypatch, "VQA_WIDE_K", "0")  # the reference handle: plain exact passes
    ref = _index(x, monkeypatch, sketch=False)
    ske = _index(x, monkeypatch, sketch=True)
    s0, _, p0 = _search(ref, q, k)
    s1, _, p1 = _search(ske, q, k)
    ref.close()
    ske.close()
    assert np.array_equal(p1, p0) and np.array_equal(s1, s0)
    assert p1[0].tolist() == [r for r in range(k + 1) if r != 5]  # ties by position


def test_rows_the_bound_cannot_prune_take_the_exact_fallback(native_lib, monkeypatch):
    """Every row equal to every query: each (query, row) pair is a candidate, the scan's regions fill up, the overflow flag
    sends the search through the exact main launch (gated on the flag, no host round trip) -- same bits as the exact scan."""
    n, d, b, k = 200_000, 64, 48, 10
    rng = np.random.default_rng(1)
    v = _unit(rng, 1, d)
    x = np.repeat(v, n, axis=0)
    x[77_777] = _unit(rng, 1, d)[0]  # one row that is not a duplicate
    q = np.repeat(v, b, axis=0)
    ref = _index(x, monkeypatch, sketch=False)
    ske = _index(x, monkeypatch, sketch=True)
    s0, _, p0 = _search(ref, q, k)
    s1, _, p1 = _search(ske, q, k)
    ref.close()
    ske.close()
    assert np.array_equal(p1, p0) and np.array_equal(s1, s0)
    assert p1[0].tolist() == list(range(k))  # ties by position


def test_outlier_rows_unnormalised_rows_and_zero_rows(native_lib, monkeypatch):
    """The bound holds for any fp16 rows: a tile whose scale one huge component sets (coarse codes for its other rows), rows of
    norm 30, zero rows, a zero query."""
    n, d, b, k = 180_000, 96, 32, 10
    rng = np.random.default_rng(5)
    x, q = _unit(rng, n, d), _unit(rng, b, d)
    x[70_000, 5] = np.float16(900.0)        # one component of one row: the scale of its tile grows 5000-fold
    x[70_001] = (x[70_001].astype(np.float32) * 30).astype(np.float16)
    x[100_000:100_300] = 0
    x[150_000] = (q[3].astype(np.float32) * 0.5).astype(np.float16)
    q[1] = 0
    q[2, 5] = np.float16(1.0)               # looks along the outlier component
    ref = _index(x, monkeypatch, sketch=False)
    ske = _index(x, monkeypatch, sketch=True)
    s0, _, p0 = _search(ref, q, k)
    s1, _, p1 = _search(ske, q, k)
    ref.close()
    ske.close()
    s_full = R.full_scores(q.astype(np.float32), x, R.DTYPE_F16)
    R.check_topk(s1, p1, s_full, k, score_tol=2e-3, tie_tol=1e-3)  # scores up to 900: the tolerance is relative to them
    keep = np.arange(b) != 1  # (the zero query scores 0 everywhere: any rows are a valid answer, by position)
    assert np.array_equal(p1[keep], p0[keep]) and p1[2, 0] == 70_000 and p1[1].tolist() == list(range(k))
    assert np.abs(s1 - s0).max() <= 1e-4 * max(1.0, np.abs(s0[np.isfinite(s0)]).max())


def test_shard_filled_in_unaligned_chunks_has_a_consistent_sketch(native_lib, monkeypatch):
    """vqa_index_set_rows re-derives the sketch of every tile a call touches from the stored rows, so chunks that start and end
    inside tiles leave the same sketch as one call -- and the same results, bit for bit."""
    from vietnamese_qa_system_amd.index import DeviceIndex
    n, d, b, k = 150_003, 64, 16, 10
    rng = np.random.default_rng(9)
    x, q = _unit(rng, n, d), _unit(rng, b, d)
    whole = _index(x, monkeypatch, sketch=True)
    set_option(monkeypatch, "VQA_STAGE_MIN", "2")
    parts = DeviceIndex.empty(n, d, dtype="fp16", device=0, sketch=True)
    for lo, hi in ((100_000, 150_003), (0, 777), (777, 40_001), (40_001, 100_000)):
        parts.set_rows(lo, x[lo:hi])
    assert parts.launch_info(b, k).sketch_scan == 1
    s0, _, p0 = _search(whole, q, k)
    s1, _, p1 = _search(parts, q, k)
    whole.close()
    parts.close()
    assert np.array_equal(p0, p1) and np.array_equal(s0, s1)


def test_fp32_shards_take_the_sketch_too(native_lib, monkeypatch):
    """BASELINE configs[1]'s storage type: the exact scan runs on the f32 MFMA (1/16 of the fp16 rate), the sketch scan on the same
    int8 MFMA as for fp16 shards; survivors are re-scored in fp32 from the stored fp32 rows.  Same rows as the exact scan, scores
    within fp32 rounding of a different summation order, duplicates bit-equal in position order."""
    n, d, b, k = 200_000, 768, 48, 10
    rng = np.random.default_rng(21)
    x = R.l2_normalize(rng.standard_normal((n, d)).astype(np.float32))
    q = R.l2_normalize(rng.standard_normal((b, d)).astype(np.float32))
    dup = [9, 65_535, 65_536, 150_000]
    for r in dup[1:]:
        x[r] = x[dup[0]]
    q[0] = x[dup[0]]
    ref = _index(x, monkeypatch, sketch=False, dtype="fp32")
    ske = _index(x, monkeypatch, sketch=True, dtype="fp32")
    assert ref.launch_info(b, k).sketch_scan == 0 and ske.launch_info(b, k).sketch_scan == 1
    s0, _, p0 = _search(ref, q, k)
    s1, _, p1 = _search(ske, q, k)
    ref.close()
    ske.close()
    s_full = R.full_scores(q, x, R.DTYPE_F32)
    R.check_topk(s1, p1, s_full, k, score_tol=SCORE_TOL, tie_tol=TIE_TOL)
    assert np.array_equal(p1, p0) and np.abs(s1 - s0).max() <= 2e-6
    assert p1[0, :4].tolist() == dup and len(set(s1[0, :4].tolist())) == 1


def test_clustered_rows_and_repeated_searches(native_lib, monkeypatch):
    """Embeddings of real text cluster: 50 tight clusters, queries next to their centres, so thousands of rows sit within the
    bound's slack of every threshold.  Whatever share of the pairs survives (or overflows into the exact fallback and switches
    the sketch off for the following searches), every search returns the exact scan's rows."""
    n, d, b, k = 220_000, 128, 64, 10
    rng = np.random.default_rng(31)
    centres = R.l2_normalize(rng.standard_normal((50, d)).astype(np.float32))
    x = R.l2_normalize(centres[rng.integers(0, 50, n)] + 0.05 * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float16)
    q = R.l2_normalize(centres[rng.integers(0, 50, b)] + 0.02 * rng.standard_normal((b, d)).astype(np.float32)).astype(np.float16)
    ref = _index(x, monkeypatch, sketch=False)
    ske = _index(x, monkeypatch, sketch=True)
    s0, _, p0 = _search(ref, q, k)
    s_full = R.full_scores(q.astype(np.float32), x, R.DTYPE_F16)
    for rep in range(4):  # overflow -> fallback -> cool-down searches without the sketch: same rows every time
        s1, _, p1 = _search(ske, q, k)
        R.check_topk(s1, p1, s_full, k, score_tol=SCORE_TOL, tie_tol=TIE_TOL)
        assert np.abs(s1 - s0).max() <= 3e-7, rep
        # rows may differ from the exact scan's only inside near-tie groups (clustered data has them)
        diff = p1 != p0
        assert np.all(np.abs(s1[diff] - s0[diff]) <= TIE_TOL)
    ref.close()
    ske.close()


def test_random_geometries_against_the_exact_scan(native_lib, monkeypatch):
    """A seeded sweep over shard geometries the fixed cases do not hit: ragged row counts, dimensions that pad differently in the
    fp16 rows (64-element steps) and the int8 sketch (128), k on both sides of 12, batches beyond one query tile, both storage
    types, planted duplicates.  Every search: the exact scan's rows, scores within the summation-order tolerance."""
    rng = np.random.default_rng(2026)
    for case in range(16):
        n = int(rng.integers(140_000, 330_000))
        d = int(rng.choice([32, 40, 96, 200, 384, 520, 768, 1000]))
        k = int(rng.choice([1, 5, 10, 12, 13, 24, 32, 64]))
        b = int(rng.choice([1, 3, 64, 255, 256, 257, 300]))
        dtype = "fp32" if case % 4 == 3 else "fp16"
        if dtype == "fp32":
            n = min(n, 200_000)
        x = R.l2_normalize(rng.standard_normal((n, d)).astype(np.float32))
        q = R.l2_normalize(rng.standard_normal((b, d)).astype(np.float32))
        if dtype == "fp16":
            x, q = x.astype(np.float16), q.astype(np.float16)
        planted = rng.integers(0, n, 6)
        x[planted[1:]] = x[planted[0]]
        q[0] = x[planted[0]]
        ref = _index(x, monkeypatch, sketch=False, dtype=dtype)
        ske = _index(x, monkeypatch, sketch=True, dtype=dtype)
        assert ske.launch_info(b, k).sketch_scan == 1, (case, n, d, k, b, dtype)
        s0, _, p0 = _search(ref, q, k)
        s1, _, p1 = _search(ske, q, k)
        ref.close()
        ske.close()
        tol = 3e-7 if dtype == "fp16" else 2e-6
        assert np.abs(s1 - s0).max() <= tol, (case, n, d, k, b, dtype)
        diff = p1 != p0  # other rows only inside near-tie groups (fp32 products round: order-dependent last bits)
        assert np.all(np.abs(s1[diff] - s0[diff]) <= TIE_TOL) and diff.mean() < 0.01, (case, n, d, k, b, dtype, int(diff.sum()))
        nd = min(k, len(set(planted.tolist())))
        assert sorted(p1[0, :nd].tolist()) == sorted(set(planted.tolist()))[:nd], (case, n, d, k, b, dtype)


def test_third_cascade_level_changes_no_bit(native_lib, monkeypatch):
    """k >= 16 on a shard large enough: [first stage | a second stage of twice its tiles | the rest], the second stage against the first
    one's exact k-th best, the rest against the k-th best of both.  Same keys in the same lists as with two levels (only fewer
    of them): rows and scores bit for bit; exact duplicates in all three stages."""
    n, d, b, k = 330_000, 128, 19, 30
    rng = np.random.default_rng(12)
    x, q = _unit(rng, n, d), _unit(rng, b, d)
    for r in (70_000, 196_607, 196_608, 300_000):  # first stage: rows < 65 536, second: < 196 608
        x[r] = x[5]
    q[0] = x[5]
    three = _index(x, monkeypatch, sketch=True)
    assert three.launch_info(b, k).first_stage_rows == 3 * 65536 and three.launch_info(b, 10).first_stage_rows == 65536
    s3, _, p3 = _search(three, q, k)
    assert three.sketch_stats()["overflow"] == 0
    three.close()
    set_option(monkeypatch, "VQA_SKETCH_MID_K", "0")
    two = _index(x, monkeypatch, sketch=True)
    assert two.launch_info(b, k).first_stage_rows == 65536
    s2, _, p2 = _search(two, q, k)
    two.close()
    set_option(monkeypatch, "VQA_SKETCH_MID_K", None)
    assert np.array_equal(p3, p2) and np.array_equal(s3, s2)
    assert p3[0, :5].tolist() == [5, 70_000, 196_607, 196_608, 300_000] and len(set(s3[0, :5].tolist())) == 1


@pytest.mark.parametrize("dtype,d,k", [("fp16", 200, 10), ("fp16", 768, 30), ("fp32", 96, 12)])
def test_row_major_rescoring_copy_changes_no_bit(native_lib, monkeypatch, dtype, d, k):
    """VQA_INDEX_RESCORE_ROWS: the sketch search scores its surviving pairs from a row-major copy of the stored rows (whole
    cache lines) instead of the scan's tiled layout (64-byte pieces) -- same values, same summation order, so positions AND
    scores are bit-equal with and without it; a shard filled in unaligned chunks keeps the copy consistent."""
    from vietnamese_qa_system_amd.index import DeviceIndex
    n, b = 170_003, 96
    rng = np.random.default_rng(77 + d)
    x = R.l2_normalize(rng.standard_normal((n, d)).astype(np.float32))
    q = R.l2_normalize(rng.standard_normal((b, d)).astype(np.float32))
    if dtype == "fp16":
        x, q = x.astype(np.float16), q.astype(np.float16)
    x[[5, 70_000, n - 1]] = x[5]
    q[0] = x[5]
    plain = _index(x, monkeypatch, sketch=True, dtype=dtype, rescore_copy=False)
    set_option(monkeypatch, "VQA_STAGE_MIN", "2")
    copy = DeviceIndex.empty(n, d, dtype=dtype, device=0, sketch=True, rescore_copy=True)
    for lo, hi in ((90_001, n), (0, 513), (513, 90_001)):
        copy.set_rows(lo, x[lo:hi])
    esize = 2 if dtype == "fp16" else 4
    d_pad = (d * esize + 127) // 128 * 128 // esize
    assert copy.device_bytes() - plain.device_bytes() == (n + 255) // 256 * 256 * d_pad * esize
    assert copy.launch_info(b, k).sketch_scan == 1 and plain.launch_info(b, k).sketch_scan == 1
    s0, _, p0 = _search(plain, q, k)
    s1, _, p1 = _search(copy, q, k)
    plain.close()
    copy.close()
    assert np.array_equal(p0, p1) and np.array_equal(s0, s1)
    assert p1[0, :3].tolist() == [5, 70_000, n - 1] and len(set(s1[0, :3].tolist())) == 1


def test_first_form_with_an_exact_first_stage_still_agrees(native_lib, monkeypatch):
    """VQA_SKETCH_CASCADE=0 keeps the round's first form of the search (exact fp16 first stage -> theta -> one sketch scan) as an
    A / B switch: same rows and bit-equal scores as the cascade (every returned score comes from the same re-scoring kernel)."""
    n, d, b, k = 210_000, 128, 50, 10
    rng = np.random.default_rng(4)
    x, q = _unit(rng, n, d), _unit(rng, b, d)
    x[[7, 65_536, n - 2]] = x[7]
    q[0] = x[7]
    cascade = _index(x, monkeypatch, sketch=True)
    set_option(monkeypatch, "VQA_SKETCH_CASCADE", "0")
    first = _index(x, monkeypatch, sketch=True)
    s0, _, p0 = _search(cascade, q, k)
    s1, _, p1 = _search(first, q, k)
    cascade.close()
    first.close()
    assert np.array_equal(p0, p1) and np.array_equal(s0, s1)
    assert p0[0, :3].tolist() == [7, 65_536, n - 2]


@pytest.mark.parametrize("b", [1, 3])
def test_one_query_with_thousands_of_candidates_stays_on_the_sketch(native_lib, monkeypatch, b):
    """A single query whose neighbourhood holds 5000 near-duplicates scattered over the shard: every one of them is a candidate.
    The candidate list of a query is 16 sub-lists of 1024; they must fill evenly whatever the batch size (a first form picked
    the sub-list by the pair's position in its scan region alone: with one query every region's first pairs landed in the same
    three sub-lists and a 10M-row shard fell back to the exact scan for B = 1 and for the last query tile of B = 257)."""
    n, d, k = 300_000, 64, 10
    rng = np.random.default_rng(12)
    x = _unit(rng, n, d)
    q = _unit(rng, b, d)
    near = rng.choice(n, 5000, replace=False)
    x[near] = R.l2_normalize(q[0].astype(np.float32)[None, :] + 0.01 * rng.standard_normal((5000, d)).astype(np.float32)).astype(np.float16)
    ref = _index(x, monkeypatch, sketch=False)
    ske = _index(x, monkeypatch, sketch=True)
    assert ske.sketch_state() == 0 and ref.sketch_state() == -1
    s0, _, p0 = _search(ref, q, k)
    s1, _, p1 = _search(ske, q, k)
    state = ske.sketch_state()
    ref.close()
    ske.close()
    assert state == 0, "the sketch search overflowed into its exact fallback"
    assert np.abs(s1 - s0).max() <= 3e-7
    diff = p1 != p0
    assert np.all(np.abs(s1[diff] - s0[diff]) <= TIE_TOL)


@pytest.mark.parametrize("shape", ["common component", "constant outlier dimensions", "wide outlier dimensions"])
def test_anisotropic_rows_stay_on_the_sketch_and_agree(native_lib, monkeypatch, shape):
    """Embeddings of one encoder are not isotropic: a large common component (mean cosine 0.5), a few dimensions with huge constant
    values, or with huge variance.  The sketch is cut from the centred, rotated rows (convert.hip: sketch_rotate; the centre is
    the mean of the shard's first fill), which keeps the quantiser's range on what distinguishes rows: such shards stay on the
    sketch search -- and return the exact scan's rows, whatever the chunking of the fill."""
    from vietnamese_qa_system_amd.index import DeviceIndex
    n, d, b, k = 260_000, 256, 64, 10
    rng = np.random.default_rng(41)
    def make(m):
        x = R.l2_normalize(rng.standard_normal((m, d)).astype(np.float32))
        if shape == "common component":
            x = x + np.ones(d, np.float32) / np.sqrt(d)
        elif shape == "constant outlier dimensions":
            x[:, 7] += 0.7
            x[:, 200] -= 0.7
        else:
            x[:, 7] *= 20
            x[:, 200] *= 20
        return R.l2_normalize(x).astype(np.float16)
    x, q = make(n), make(b)
    x[[11, 70_000, n - 3]] = x[11]
    q[0] = x[11]
    ref = _index(x, monkeypatch, sketch=False)
    set_option(monkeypatch, "VQA_STAGE_MIN", "2")
    ske = DeviceIndex.empty(n, d, dtype="fp16", device=0, sketch=True)
    for lo, hi in ((0, 100_001), (100_001, n)):  # the centre comes from the first fill; the second chunk is cut against it too
        ske.set_rows(lo, x[lo:hi])
    s0, _, p0 = _search(ref, q, k)
    s1, _, p1 = _search(ske, q, k)
    state = ske.sketch_state()
    ref.close()
    ske.close()
    assert state == 0, "the sketch search overflowed into its exact fallback"
    assert np.abs(s1 - s0).max() <= 6e-7
    diff = p1 != p0
    assert np.all(np.abs(s1[diff] - s0[diff]) <= TIE_TOL) and diff.mean() < 0.01
    assert p1[0, :3].tolist() == [11, 70_000, n - 3] and len(set(s1[0, :3].tolist())) == 1

"""Both ways a large shard can answer a search return the same BITS (VERDICT r5 "two paths, last-bit different, chosen asynchronously").

A shard that keeps a sketch answers through the sketch cascade (scores from the re-scoring kernel's fma chain) or -- while the sketch is
paused after an unprofitable / overflowing search, or with the sketch off -- through the exact scan (MFMA sums).  Since round 6 every
exact path of such a shard ends by scoring its rows in the re-scoring arithmetic and re-ranking them (csrc/sketch.hip
final_rescore_kernel), over the scan's k + 2 best rows where a pass has room (k <= 10).  Reference call site:
inference_pipeline/db_utils/heavy_ranker.py:98-101 -- the scores and their order are what `embeddings.search` hands back."""
import numpy as np
import pytest
import torch

from conftest import set_option

from oracle import retrieval as R

pytestmark = pytest.mark.gpu


def _rows(rng, n, d):
    x = R.l2_normalize(rng.standard_normal((n, d)).astype(np.float32)).astype(np.float16)
    # a cluster of near-duplicates of one row: fp16 neighbours in a few coordinates, so a query close to it sees a run of scores a few
    # 1e-7 apart -- the gaps inside which MFMA sums and the fma chain order rows differently
    base = x[17].copy()
    for j, r in enumerate(range(1000, 1000 + 24)):
        y = base.copy()
        idx = rng.integers(0, d, size=3)
        y[idx] = np.nextafter(y[idx], np.float16(np.inf if j % 2 else -np.inf))
        x[r] = y
    x[n - 5] = base  # an exact duplicate far away (last tile)
    return x, base


def _search(ix, q, k):
    s, i, p = ix.search(torch.from_numpy(q).cuda(), k, return_positions=True)
    torch.cuda.synchronize()
    return s.cpu().numpy(), i.cpu().numpy(), p.cpu().numpy()


@pytest.mark.parametrize("n,d,b,k", [(150_000, 768, 64, 10), (200_000, 128, 256, 10), (140_000, 384, 33, 1), (160_000, 768, 16, 12),
                                     (180_000, 256, 24, 30)])
def test_paused_sketch_returns_the_sketch_paths_bits(native_lib, monkeypatch, n, d, b, k):
    from vietnamese_qa_system_amd.index import DeviceIndex
    rng = np.random.default_rng(n + d + k)
    x, base = _rows(rng, n, d)
    q = R.l2_normalize(rng.standard_normal((b, d)).astype(np.float32)).astype(np.float16)
    q[0] = base
    q[1] = R.l2_normalize((base.astype(np.float32) + 0.02 * rng.standard_normal(d).astype(np.float32))[None])[0].astype(np.float16)
    set_option(monkeypatch, "VQA_STAGE_MIN", "2")
    ske = DeviceIndex(x, dtype="fp16", device=0, sketch=True)
    assert ske.launch_info(b, k).sketch_scan == 1 and ske.sketch_state() == 0
    s1, i1, p1 = _search(ske, q, k)          # the sketch cascade
    assert ske.sketch_stats()["overflow"] == 0
    ske.sketch_pause(2)                      # what the handle does by itself behind an overflow / an unprofitable search: the exact scan
    assert ske.sketch_state() > 0
    s2, i2, p2 = _search(ske, q, k)
    s3, i3, p3 = _search(ske, q, k)
    assert np.array_equal(p1, p2) and np.array_equal(s1, s2) and np.array_equal(i1, i2), "the paused (exact) search returns other bits than the sketch search"
    assert np.array_equal(p2, p3) and np.array_equal(s2, s3)
    ske.sketch_pause(0)                      # ... and back: flipping the path mid-sequence never shows in the results
    assert ske.sketch_state() == 0
    s6, i6, p6 = _search(ske, q, k)
    assert np.array_equal(p1, p6) and np.array_equal(s1, s6)
    # a shard without a sketch on the same plan: the exact scan again, the same bits
    ref = DeviceIndex(x, dtype="fp16", device=0, sketch=False)
    s4, i4, p4 = _search(ref, q, k)
    assert np.array_equal(p1, p4) and np.array_equal(s1, s4)
    # ... and with the final re-scoring off it is the MFMA sums: same rows up to near-ties, scores within the last bits
    raw = DeviceIndex(x, dtype="fp16", device=0, sketch=False, options={"final_rescore": 0})
    s5, _, p5 = _search(raw, q, k)
    for ix in (ske, ref, raw):
        ix.close()
    assert np.abs(s5 - s1).max() <= 5e-7
    diff = p5 != p1
    assert np.all(np.abs(s5[diff] - s1[diff]) <= 2e-6)
    # the oracle agrees (tie-aware), and the planted duplicates lead query 0 in position order with equal bits
    s_full = R.full_scores(q.astype(np.float32), x, R.DTYPE_F16)
    R.check_topk(s1, p1, s_full, k, score_tol=1e-5, tie_tol=2e-6)
    row = p1[0].tolist()  # (the near-duplicates that were nudged upwards score a hair above the row itself and may fill a short list)
    if 17 in row and n - 5 in row:
        a, c = row.index(17), row.index(n - 5)
        # exact duplicates: equal bits, position order; only rows with that very score (a near-duplicate whose nudges cancel) may sit between
        assert a < c and np.all(s1[0, a:c + 1] == s1[0, a]) and row[a:c + 1] == sorted(row[a:c + 1])
    assert k < 30 or (17 in row and n - 5 in row)


def test_fp32_shard_paused_and_unpaused(native_lib, monkeypatch):
    from vietnamese_qa_system_amd.index import DeviceIndex
    n, d, b, k = 140_000, 768, 40, 10
    rng = np.random.default_rng(5)
    x = R.l2_normalize(rng.standard_normal((n, d)).astype(np.float32))
    x[70_000] = x[3]
    q = R.l2_normalize(rng.standard_normal((b, d)).astype(np.float32))
    q[0] = x[3]
    set_option(monkeypatch, "VQA_STAGE_MIN", "2")
    ske = DeviceIndex(x, dtype="fp32", device=0, sketch=True)
    assert ske.launch_info(b, k).sketch_scan == 1
    s1, _, p1 = _search(ske, q, k)
    ske.sketch_pause(1)
    s2, _, p2 = _search(ske, q, k)
    ske.close()
    assert np.array_equal(p1, p2) and np.array_equal(s1, s2)
    assert p1[0, 0] == 3 and p1[0, 1] == 70_000 and s1[0, 0] == s1[0, 1]

"""BASELINE.json's full single-GPU size (configs[2]: 10M x 768 fp16, batch 256, top-10) through size-independent
properties -- the oracle cannot score 10M rows x 256 queries inside a test, so what is checked here does not need it to:

  * planted needles: rows equal to a query (and a duplicate of it further down) must come back first, in position order;
  * shard invariance: searching two half shards and merging (``vqa_merge_topk``) is bit-identical to searching the whole
    -- a row's score does not depend on which workgroup or launch computed it;
  * sampled exactness: the oracle scores a random 200k-row sample for every query; no sampled row may beat the returned
    k-th score without being in the result (a missed candidate anywhere in the shard has a 2 % chance per row of showing).

``VQA_FULLSIZE_ROWS`` shrinks the shard for a quick run (default 10,000,000 rows = 15.36 GB).
"""
import os

import numpy as np
import pytest
import torch

from oracle import retrieval as R

pytestmark = pytest.mark.gpu

N = int(os.environ.get("VQA_FULLSIZE_ROWS", "10000000"))
D, B, K = 768, 256, 10


@pytest.fixture(scope="module")
def shard():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321)
    x = torch.empty((N, D), dtype=torch.float16, device=dev)
    for c0 in range(0, N, 1 << 18):
        c1 = min(N, c0 + (1 << 18))
        y = torch.randn((c1 - c0, D), generator=gen, device=dev)
        x[c0:c1] = (y / y.norm(dim=1, keepdim=True)).half()
    q = torch.randn((B, D), generator=gen, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).half()
    # needles: query i at row pos[i] and again at row dup[i] > pos[i] (queries 0..31)
    rng = np.random.default_rng(17)
    pos = np.sort(rng.choice(N // 2, size=32, replace=False))
    dup = N // 2 + np.sort(rng.choice(N - N // 2, size=32, replace=False))
    x[torch.from_numpy(pos).to(dev)] = q[:32]
    x[torch.from_numpy(dup).to(dev)] = q[:32]
    yield x, q, pos, dup
    del x


def test_full_size_fp8_index_properties(native_lib, shard):
    """The same shard as an fp8 (e4m3) index (7.68 GB): needles first in position order, shard invariance bit for bit,
    and sampled exactness against the oracle scoring the SAME e4m3 codes (parity bar of the fp8 MFMA: 5e-5 / 1e-4)."""
    from vietnamese_qa_system_amd.index import DeviceIndex, merge_topk
    x, q, pos, dup = shard
    xf = x  # fp16 source values; the index quantises 16 * x to e4m3 itself
    full = DeviceIndex(xf, id_base=1, dtype="fp8", device=0)
    s, i, p = full.search(q, K, return_positions=True)
    torch.cuda.synchronize()
    s_h, p_h = s.cpu().numpy(), p.cpu().numpy()
    assert np.all(np.diff(s_h, axis=1) <= 0) and np.array_equal(i.cpu().numpy(), p_h + 1)
    assert np.array_equal(p_h[:32, 0], pos) and np.array_equal(p_h[:32, 1], dup) and np.array_equal(s_h[:32, 0], s_h[:32, 1])
    half = N // 2
    lo = DeviceIndex(xf[:half], id_base=1, dtype="fp8", device=0)
    s0, i0, _ = lo.search(q, K)
    lo.close()
    hi = DeviceIndex(xf[half:], id_base=1 + half, dtype="fp8", device=0)
    s1, i1, _ = hi.search(q, K)
    hi.close()
    ms, mi = merge_topk(torch.stack([s0, s1]), torch.stack([i0, i1]), K)
    torch.cuda.synchronize()
    assert torch.equal(mi, i) and torch.equal(ms, s)
    rng = np.random.default_rng(29)
    sample = np.sort(rng.choice(N, size=min(N, 50_000), replace=False))
    xs = x[torch.from_numpy(sample).to(x.device)].float().cpu().numpy()
    codes = R.e4m3_encode(xs * 16.0)
    qd = R.e4m3_decode(R.e4m3_encode(q.float().cpu().numpy() * 16.0)) / 256.0  # what the kernel multiplies, scores / 256
    ref = R.full_scores(qd, codes, R.DTYPE_FP8_E4M3)
    kth = s_h[:, K - 1][:, None].astype(np.float64)
    beat = ref > kth + 1e-4
    for b in range(B):
        missing = set(sample[beat[b]].tolist()) - set(p_h[b].tolist())
        assert not missing, f"query {b}: rows {sorted(missing)[:5]} beat the returned k-th score but are not in the result"
    full.close()


def test_full_size_properties(native_lib, shard):
    from vietnamese_qa_system_amd.index import DeviceIndex, merge_topk
    x, q, pos, dup = shard
    full = DeviceIndex(x, id_base=1, dtype="fp16", device=0)
    s, i, p = full.search(q, K, return_positions=True)
    torch.cuda.synchronize()
    s_h, i_h, p_h = s.cpu().numpy(), i.cpu().numpy(), p.cpu().numpy()

    # ---- well-formed: sorted by (score desc, position asc), ids = positions + id_base, no duplicates
    assert np.all(np.diff(s_h, axis=1) <= 0)
    tie = np.diff(s_h, axis=1) == 0
    assert np.all(np.diff(p_h, axis=1)[tie] > 0)
    assert np.array_equal(i_h, p_h + 1)
    assert all(len(set(row)) == K for row in p_h.tolist())

    # ---- run to run: the append order inside a candidate list depends on wave timing, the result must not
    for _ in range(3):
        s2, i2, _ = full.search(q, K)
        assert torch.equal(s2, s) and torch.equal(i2, i)

    # ---- the shard is searched through its int8 sketch, and neither the full batch nor a single query, a ragged last query tile
    # or a wide k overflowed into the exact fallback (which would silently switch the sketch off for the next searches)
    assert full.launch_info(q.shape[0], K).sketch_scan == 1
    ragged = torch.cat([q, q[:1]])  # one query in the last tile
    # (k = 30: three cascade levels by k; k = 100: four -- the first stage's leading quarter as a stage of its own -- and the radix
    # selection of the merges; the full batch at k = 10 takes three levels by the shard's size)
    for qs, kk, rows in ((q[:1], K, [0]), (ragged, K, list(range(q.shape[0])) + [0]), (q[:64], 30, list(range(64))),
                         (q[:48], 100, list(range(48)))):
        sq, iq, _ = full.search(qs, kk)
        torch.cuda.synchronize()
        assert torch.equal(sq[:, :K], s[rows]) and torch.equal(iq[:, :K], i[rows])
        assert bool((sq[:, 1:] <= sq[:, :-1]).all()) and all(len(set(r)) == kk for r in iq.tolist())
    assert full.sketch_state() == 0

    # ---- planted needles: self-score = sum of squares of the stored fp16 values; the copy further down ties with it
    qq = q[:32].float().cpu().numpy()
    self_score = (qq.astype(np.float64) ** 2).sum(1)
    assert np.array_equal(p_h[:32, 0], pos) and np.array_equal(p_h[:32, 1], dup)
    assert np.abs(s_h[:32, 0] - self_score).max() < 1e-5 and np.array_equal(s_h[:32, 0], s_h[:32, 1])

    # ---- shard invariance: two half shards + merge == the whole, bit for bit
    half = N // 2
    lo = DeviceIndex(x[:half], id_base=1, dtype="fp16", device=0)
    s0, i0, _ = lo.search(q, K)
    lo.close()
    hi = DeviceIndex(x[half:], id_base=1 + half, dtype="fp16", device=0)
    s1, i1, _ = hi.search(q, K)
    hi.close()
    ms, mi = merge_topk(torch.stack([s0, s1]), torch.stack([i0, i1]), K)
    torch.cuda.synchronize()
    assert torch.equal(mi, i) and torch.equal(ms, s)

    # ---- sampled exactness against the oracle (same stored values, fp64 accumulation)
    rng = np.random.default_rng(23)
    sample = np.sort(rng.choice(N, size=min(N, 200_000), replace=False))
    xs = x[torch.from_numpy(sample).to(x.device)].cpu().numpy()
    ref = R.full_scores(q.float().cpu().numpy(), xs, R.DTYPE_F16)  # [B, sample]
    kth = s_h[:, K - 1][:, None].astype(np.float64)
    beat = ref > kth + 2e-6  # clearly above the k-th returned score: must have been returned
    for b in range(B):
        missing = set(sample[beat[b]].tolist()) - set(p_h[b].tolist())
        assert not missing, f"query {b}: rows {sorted(missing)[:5]} beat the returned k-th score but are not in the result"
    # and the scores of returned rows that fall in the sample agree with the oracle
    where = {int(r): j for j, r in enumerate(sample.tolist())}
    for b in range(B):
        for j in range(K):
            col = where.get(int(p_h[b, j]))
            if col is not None:
                assert abs(float(s_h[b, j]) - ref[b, col]) < 1e-5
    full.close()

"""fp8 (OCP e4m3) and fp32 index storage: parity of the HIP path against the oracle scoring the SAME stored values
(BASELINE.json configs[1] fp32 index, configs[4] fp8 index), and the recall of an fp8 index against the fp32 reference."""
import numpy as np
import pytest
import torch

from oracle import retrieval as R

pytestmark = pytest.mark.gpu

FP8_SCALE = 16.0


def _unit(n, d, seed):
    return R.l2_normalize(np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32))


@pytest.mark.parametrize("n,d,b,k", [(3000, 768, 40, 10), (257, 100, 3, 5), (70001, 128, 260, 10)])
def test_fp8_index_matches_oracle_on_same_codes(native_lib, n, d, b, k):
    from vietnamese_qa_system_amd.index import DeviceIndex
    x, q = _unit(n, d, 1), _unit(b, d, 2)
    ix = DeviceIndex(x, dtype="fp8", id_base=1)
    codes, _ = ix.get_rows()
    ref_codes = R.e4m3_encode(x * FP8_SCALE)
    assert codes.dtype == np.uint8 and np.array_equal(codes, ref_codes)  # device codec == oracle codec, bit for bit
    s, i, p = ix.search(torch.from_numpy(q).cuda(), k, return_positions=True)
    torch.cuda.synchronize()
    qd = R.e4m3_decode(R.e4m3_encode(q * FP8_SCALE)).astype(np.float64)
    full = qd @ R.e4m3_decode(ref_codes).astype(np.float64).T / (FP8_SCALE * FP8_SCALE)
    # v_mfma_f32_16x16x32_fp8_fp8 does not accumulate at full fp32 precision: measured relative score error ~4e-5
    # (2^-15) on MI355X, so the bar for fp8 is 5e-5 absolute (scores <= 1) and ranks are compared with tie_tol 1e-4
    R.check_topk(s.cpu().numpy(), p.cpu().numpy(), full, k, score_tol=5e-5, tie_tol=1e-4)
    assert np.array_equal(i.cpu().numpy(), p.cpu().numpy() + 1)
    ix.close()


@pytest.mark.parametrize("n,d,b,k", [(3000, 768, 40, 10), (300, 50, 5, 12), (40000, 64, 257, 10)])
def test_fp32_index_matches_oracle(native_lib, n, d, b, k):
    from vietnamese_qa_system_amd.index import DeviceIndex
    x, q = _unit(n, d, 3), _unit(b, d, 4)
    ix = DeviceIndex(x, dtype="fp32")
    rows, _ = ix.get_rows()
    assert rows.dtype == np.float32 and np.array_equal(rows, x)
    s, i, p = ix.search(torch.from_numpy(q).cuda(), k, return_positions=True)
    torch.cuda.synchronize()
    R.check_topk(s.cpu().numpy(), p.cpu().numpy(), R.full_scores(q, x, R.DTYPE_F32), k, score_tol=1e-5, tie_tol=2e-6)
    ix.close()


def test_fp8_recall_against_fp32_reference(native_lib):
    """configs[4] reports recall@10 of the fp8 index against the fp32 index (expected < 1; stated, not asserted tight)."""
    from vietnamese_qa_system_amd.index import DeviceIndex
    x, q = _unit(50000, 768, 5), _unit(64, 768, 6)
    _, ref_ids, _ = R.search(q, x, 10)
    ix = DeviceIndex(x, dtype="fp8")
    _, i, _ = ix.search(torch.from_numpy(q).cuda(), 10)
    torch.cuda.synchronize()
    recall = R.recall_at_k(i.cpu().numpy(), ref_ids)
    assert 0.6 < recall <= 1.0, recall
    ix.close()


def test_embeddings_fp8_and_fp32_save_load(native_lib, tmp_path):
    from vietnamese_qa_system_amd import Embeddings
    x, q = _unit(500, 96, 7), _unit(5, 96, 8)
    for dtype in ("fp8", "fp32"):
        emb = Embeddings(dtype=dtype, min_score=None)
        emb.index_vectors(list(range(1, 501)), x)
        before = emb.batchsearch(q, 4)
        emb.save(str(tmp_path / dtype))
        after = Embeddings().load(str(tmp_path / dtype))
        after.min_score = None
        assert after.dtype == dtype and after.batchsearch(q, 4) == before

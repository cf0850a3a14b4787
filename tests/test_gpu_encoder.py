"""HIP question encoder (through the C ABI) against the numpy oracle, which is itself pinned to HF transformers by
tests/test_oracle_encoder.py.  Tolerance (fp16 storage, fp32 accumulation, stated per SURVEY.md section 7): pooled vectors
within 2e-2 absolute and cosine >= 0.999 of the fp64 oracle; retrieval with the encoded queries returns the oracle's
top-1 ids."""
import numpy as np
import pytest
import torch

from oracle import encoder as E
from oracle import retrieval as R

pytestmark = pytest.mark.gpu

TINY = dict(vocab_size=100, hidden=64, layers=2, heads=4, ffn=128, max_pos=40, type_vocab=1, pad_id=1, ln_eps=1e-5)


def _cos(a, b):
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def test_tiny_fixture_all_poolings(native_lib, golden_dir):
    from vietnamese_qa_system_amd.encoder import QuestionEncoder
    g = np.load(f"{golden_dir}/enc_tiny.npz")
    w = {k[2:]: g[k] for k in g.files if k.startswith("w.")}
    enc = QuestionEncoder(w, TINY, max_tokens=64)
    ids, mask = g["input_ids"], g["attention_mask"]
    cls_raw = enc.forward(ids, mask, pooling="cls", normalize=False).cpu().numpy()
    assert np.abs(cls_raw - g["dpr_pooler_output"]).max() < 2e-2      # HF DPR pooler_output
    assert _cos(cls_raw, g["dpr_pooler_output"]).min() > 0.9995
    for pooling in ("cls", "mean"):
        got = enc.forward(ids, mask, pooling=pooling, normalize=True).cpu().numpy()
        ref = E.encode(w, TINY, ids, mask, pooling=pooling)
        assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-5
        assert _cos(got, ref).min() > 0.9995 and np.abs(got - ref).max() < 5e-3
    with pytest.raises(ValueError):
        enc.forward(np.zeros((40, 12), np.int32), np.ones((40, 12), np.int32))  # 480 tokens > max_tokens
    with pytest.raises(ValueError):
        enc.forward(ids, mask, pooling="max")
    enc.close()


# (2, 40, 32) and (1, 64, 16): >= 1024 tokens run the 256 x 128 LDS-DMA GEMM (1280 tokens: a ragged last row tile);
# (12, 1, 32), (2, 2, 32), (2, 3, 9): <= 64 tokens run the skinny GEMM (single query; full 4 token tiles; ragged 27 tokens)
# (2, 37, 32): 1184 tokens = a ragged last row tile whose rows past M are computed and stored into the workspace's padding rows
@pytest.mark.parametrize("layers,b,l", [(2, 8, 32), (12, 12, 24), (2, 3, 80), (1, 2, 256), (2, 40, 32), (1, 64, 16), (2, 37, 32),
                                        (12, 1, 32), (2, 2, 32), (2, 3, 9)])
def test_phobert_base_shape_vs_oracle(native_lib, layers, b, l):
    from vietnamese_qa_system_amd.encoder import QuestionEncoder
    cfg = dict(E.PHOBERT_BASE, layers=layers)
    w = E.synthetic_weights(cfg, seed=5, layers=layers)
    ids, mask = E.synthetic_tokens(cfg, b, l, seed=9)
    enc = QuestionEncoder(w, cfg, max_tokens=b * l)
    for pooling in ("cls", "mean"):
        got = enc.forward(ids, mask, pooling=pooling).cpu().numpy()
        ref = E.encode(w, cfg, ids, mask, pooling=pooling)
        assert _cos(got, ref).min() > 0.999, (pooling, _cos(got, ref).min())
        assert np.abs(got - ref).max() < 2e-2
    enc.close()


def test_baseline_config1_encoder_batch_at_full_size(native_lib):
    """BASELINE configs[1]'s encoder batch at full size: PhoBERT-base shape, 12 layers, B = 256, L = 32 (8192 token rows,
    the persistent LDS-DMA GEMM path) against the fp64 oracle -- every pooled vector, both poolings."""
    from vietnamese_qa_system_amd.encoder import QuestionEncoder
    cfg = dict(E.PHOBERT_BASE)
    w = E.synthetic_weights(cfg, seed=0)
    ids, mask = E.synthetic_tokens(cfg, 256, 32, seed=1)
    enc = QuestionEncoder(w, cfg, max_tokens=256 * 32)
    for pooling in ("cls", "mean"):
        got = enc.forward(ids, mask, pooling=pooling).cpu().numpy()
        ref = E.encode(w, cfg, ids, mask, pooling=pooling)
        assert _cos(got, ref).min() > 0.999, (pooling, _cos(got, ref).min())
        assert np.abs(got - ref).max() < 2e-2
        assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-5
    enc.close()


def test_small_batches_replay_a_graph(native_lib):
    """B * L <= 1024 tokens: the first call of a shape runs eagerly, the second captures a hipGraph over the encoder's
    staging buffers, later ones replay it -- every call must see ITS inputs and agree with the oracle."""
    from vietnamese_qa_system_amd.encoder import QuestionEncoder
    cfg = dict(E.PHOBERT_BASE, layers=2)
    w = E.synthetic_weights(cfg, seed=3, layers=2)
    enc = QuestionEncoder(w, cfg, max_tokens=4 * 24)
    outs = []
    for seed in (1, 2, 3, 4):  # same shape, different tokens: eager, capture + replay, replay, replay
        ids, mask = E.synthetic_tokens(cfg, 4, 24, seed=seed)
        got = enc.forward(ids, mask, pooling="mean").cpu().numpy()
        ref = E.encode(w, cfg, ids, mask, pooling="mean")
        assert _cos(got, ref).min() > 0.999 and np.abs(got - ref).max() < 2e-2, seed
        outs.append(got)
    assert not np.allclose(outs[1], outs[2])  # a replay that ignored its inputs would repeat the captured call's output
    ids, mask = E.synthetic_tokens(cfg, 4, 24, seed=2)
    again = enc.forward(ids, mask, pooling="mean").cpu().numpy()
    assert np.array_equal(again, outs[1])  # same inputs through the replayed graph: bit-identical
    enc.close()


def test_encoder_feeds_retrieval(native_lib):
    """configs[1] in miniature: encoder forward -> fp16 index search; ids must match the all-oracle pipeline."""
    from vietnamese_qa_system_amd import Embeddings
    from vietnamese_qa_system_amd.encoder import QuestionEncoder, TextEncoder
    cfg = dict(E.PHOBERT_BASE, layers=2)
    w = E.synthetic_weights(cfg, seed=2, layers=2)
    ids, mask = E.synthetic_tokens(cfg, 48, 16, seed=4)
    table = {f"q{i}": (ids[i], mask[i]) for i in range(48)}

    def tokenizer(texts):
        return np.stack([table[t][0] for t in texts]), np.stack([table[t][1] for t in texts])

    enc = QuestionEncoder(w, cfg, max_tokens=48 * 16)
    ref_q = E.encode(w, cfg, ids, mask, pooling="mean").astype(np.float32)
    rng = np.random.default_rng(0)
    # corpus = slightly noisy copies of the query embeddings (cosine ~0.998 to their own query, <= 0.96 to any other --
    # random-weight embeddings are close to each other) + distractors: every query has a clear nearest document
    docs = np.concatenate([ref_q + 0.002 * rng.standard_normal(ref_q.shape).astype(np.float32),
                           rng.standard_normal((2000, 768)).astype(np.float32) / 28])
    emb = Embeddings(encoder=TextEncoder(tokenizer, enc, pooling="mean"), min_score=None)
    emb.index_vectors(list(range(1, docs.shape[0] + 1)), docs)
    res = emb.batchsearch([f"q{i}" for i in range(48)], 1)
    x16 = R.l2_normalize(docs).astype(np.float16)
    _, ref_ids, _ = R.search(R.l2_normalize(ref_q).astype(np.float16).astype(np.float32), x16, 1, dtype=R.DTYPE_F16, id_base=1)
    assert [r[0][0] for r in res] == ref_ids[:, 0].tolist() == list(range(1, 49))
    enc.close()


def test_sequence_packing_matches_the_padded_form(native_lib):
    """Ragged batch (8-32 real tokens of 32): with a right-padded mask only the real tokens are computed (sequence packing,
    include/vqa_retrieval.h: real_tokens); the pooled vectors equal the padded computation's and the fp64 oracle's."""
    from vietnamese_qa_system_amd.encoder import QuestionEncoder
    cfg = dict(E.PHOBERT_BASE, layers=2)
    w = E.synthetic_weights(cfg, seed=5, layers=2)
    b, l = 192, 32  # 6144 positions: above the graph threshold, so the packed path runs
    ids, mask = E.synthetic_tokens(cfg, b, l, seed=21)
    assert 0.4 < mask.mean() < 0.9 and mask[:, 0].all()
    enc = QuestionEncoder(w, cfg, max_tokens=b * l)
    ids_d, mask_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    for pooling in ("cls", "mean"):
        padded = enc.forward(ids_d, mask_d, pooling=pooling).cpu().numpy()            # device mask, no count: every position
        packed = enc.forward(ids, mask, pooling=pooling).cpu().numpy()                # host mask: counted and packed
        explicit = enc.forward(ids_d, mask_d, pooling=pooling, real_tokens=int(mask.sum())).cpu().numpy()
        # packed and padded row counts take different GEMM tile shapes, so the row statistics of the folded LayerNorms are
        # added up in a different order: a last-bit change of rstd flips the odd fp16 rounding of a stored activation
        assert np.abs(packed - padded).max() < 2e-4 and np.array_equal(packed, explicit)
        ref = E.encode(w, cfg, ids[:6], mask[:6], pooling=pooling)
        assert _cos(packed[:6], ref).min() > 0.999
    # a mask with a hole is not right-padded: host-side masks fall back to the padded form ...
    holed = mask.copy()
    holed[3, 2] = 0
    got = enc.forward(ids, holed, pooling="mean").cpu().numpy()
    ref = E.encode(w, cfg, ids[3:4], holed[3:4], pooling="mean")
    assert _cos(got[3:4], ref).min() > 0.999
    # ... and announcing it as packable is caught on the device and reported by the next call
    enc.forward(ids_d, torch.from_numpy(holed).cuda(), pooling="mean", real_tokens=int(holed.sum()))
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="right-padded"):
        enc.forward(ids, mask)
    enc.forward(ids, mask)  # the flag was consumed
    # too few tokens announced: also caught
    enc.forward(ids_d, mask_d, real_tokens=int(mask.sum()) - 5)
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="announced"):
        enc.forward(ids, mask)
    # too MANY announced (the GEMMs would run on stale workspace rows past the packed ones): caught the same way
    enc.forward(ids_d, mask_d, real_tokens=int(mask.sum()) + 7)
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="announced"):
        enc.forward(ids, mask)
    enc.forward(ids, mask)
    enc.close()


def test_cls_pooling_prunes_the_last_layer_to_first_rows(native_lib, monkeypatch):
    """CLS pooling of a large batch runs the last layer's out-projection / FFN / LayerNorms on the B first-token rows only
    (encoder.hip encoder_launch).  Same vectors as the unpruned sequence (VQA_ENC_FIRST_ROWS=0 at create) up to the fp16
    rounding of a different GEMM kernel, padded and packed; mean pooling never prunes (bit-identical either way)."""
    from vietnamese_qa_system_amd.encoder import QuestionEncoder
    cfg = dict(E.PHOBERT_BASE, layers=2)
    w = E.synthetic_weights(cfg, seed=5, layers=2)
    b, l = 160, 32
    ids, mask = E.synthetic_tokens(cfg, b, l, seed=33)
    ids_d, mask_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    out = {}
    for on in ("1", "0"):
        monkeypatch.setenv("VQA_ENC_FIRST_ROWS", on)
        enc = QuestionEncoder(w, cfg, max_tokens=b * l)
        out[on] = {(p, r): enc.forward(ids_d, mask_d, pooling=p, real_tokens=int(mask.sum()) if r else 0).cpu().numpy()
                   for p in ("cls", "mean") for r in (False, True)}
        enc.close()
    ref = E.encode(w, cfg, ids[:8], mask[:8], pooling="cls")
    for r in (False, True):
        assert np.abs(out["1"]["cls", r] - out["0"]["cls", r]).max() < 2e-3
        assert _cos(out["1"]["cls", r][:8], ref).min() > 0.999
        assert np.array_equal(out["1"]["mean", r], out["0"]["mean", r])


def test_folded_layernorms_match_the_layernorm_kernels(native_lib, monkeypatch):
    """Calls of at least 1024 tokens run without LayerNorm kernels: the GEMMs carry raw rows plus per-row statistics and the
    LayerNorm is applied inside the GEMM epilogues / folded into gamma-scaled weights (encoder.hip FoldArgs).  Same vectors as
    the LayerNorm-kernel sequence (VQA_ENC_FOLD=0 at create) up to fp16 rounding -- both poolings, padded and packed, with and
    without the first-row pruning of the last layer -- and the fp64 oracle's."""
    from vietnamese_qa_system_amd.encoder import QuestionEncoder
    cfg = dict(E.PHOBERT_BASE, layers=3)
    w = E.synthetic_weights(cfg, seed=7, layers=3)
    # LayerNorm parameters away from (1, 0) so that the folded gamma / beta terms carry weight
    rng = np.random.default_rng(3)
    for k in list(w):
        if "LayerNorm.weight" in k:
            w[k] = (1.0 + 0.3 * rng.standard_normal(w[k].shape)).astype(np.float32)
        elif "LayerNorm.bias" in k:
            w[k] = (0.2 * rng.standard_normal(w[k].shape)).astype(np.float32)
    b, l = 96, 32
    ids, mask = E.synthetic_tokens(cfg, b, l, seed=35)
    ids_d, mask_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    out = {}
    for fold, first in (("1", "1"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("VQA_ENC_FOLD", fold)
        monkeypatch.setenv("VQA_ENC_FIRST_ROWS", first)
        enc = QuestionEncoder(w, cfg, max_tokens=b * l)
        out[fold, first] = {(p, r): enc.forward(ids_d, mask_d, pooling=p, real_tokens=int(mask.sum()) if r else 0).cpu().numpy()
                            for p in ("cls", "mean") for r in (False, True)}
        enc.close()
    for p in ("cls", "mean"):
        ref = E.encode(w, cfg, ids[:6], mask[:6], pooling=p)
        for r in (False, True):
            for key in (("1", "1"), ("1", "0")):
                assert np.abs(out[key][p, r] - out["0", "1"][p, r]).max() < 3e-3, (p, r, key)
                assert _cos(out[key][p, r][:6], ref).min() > 0.999
                assert np.abs(out[key][p, r][:6] - ref).max() < 2e-2


def test_folded_layernorms_with_outlier_dimensions_and_wide_gammas(native_lib, monkeypatch):
    """Trained RoBERTa / PhoBERT checkpoints are not N(0, 0.02): a few hidden dimensions carry activations tens of times the
    rest, LayerNorm gammas span two orders of magnitude and the rows have a mean far from zero.  The folded form takes the
    variance as E[x^2] - mean^2 of fp16-rounded rows and multiplies by fp16(W gamma): this input stresses exactly those
    cancellations.  The fold must stay as close to the fp64 oracle as the LayerNorm-kernel sequence does (VQA_ENC_FOLD=0)."""
    from vietnamese_qa_system_amd.encoder import QuestionEncoder
    cfg = dict(E.PHOBERT_BASE, layers=3)
    w = E.synthetic_weights(cfg, seed=17, layers=3)
    rng = np.random.default_rng(11)
    h = cfg["hidden"]
    outlier = np.array([7, 300, 588])
    for k in list(w):
        if "LayerNorm.weight" in k:
            g = np.exp(rng.uniform(np.log(0.05), np.log(4.0), h)).astype(np.float32)  # 0.05 .. 4
            g[outlier] = 12.0
            w[k] = g
        elif "LayerNorm.bias" in k:
            bta = (0.5 * rng.standard_normal(h) + 0.4).astype(np.float32)  # rows with a mean away from zero
            bta[outlier] = np.array([6.0, -5.0, 8.0], np.float32)
            w[k] = bta
        elif k.endswith("output.dense.bias"):  # both residual-producing projections: push the outlier dimensions and the row mean
            w[k] = w[k].copy()
            w[k][outlier] += np.array([15.0, -12.0, 20.0], np.float32)
            w[k] += 0.5
    w["embeddings.word_embeddings.weight"] = w["embeddings.word_embeddings.weight"].copy()
    w["embeddings.word_embeddings.weight"][:, outlier] *= 30.0
    b, l = 48, 32
    ids, mask = E.synthetic_tokens(cfg, b, l, seed=36)
    ids_d, mask_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    out = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("VQA_ENC_FOLD", fold)
        enc = QuestionEncoder(w, cfg, max_tokens=b * l)
        out[fold] = {(p, r): enc.forward(ids_d, mask_d, pooling=p, real_tokens=int(mask.sum()) if r else 0).cpu().numpy()
                     for p in ("cls", "mean") for r in (False, True)}
        enc.close()
    nref = 8
    for p in ("cls", "mean"):
        ref = E.encode(w, cfg, ids[:nref], mask[:nref], pooling=p)
        for r in (False, True):
            err_fold = np.abs(out["1"][p, r][:nref] - ref).max()
            err_plain = np.abs(out["0"][p, r][:nref] - ref).max()
            print(f"outlier fold test: pooling={p} packed={r}: |fold - oracle| = {err_fold:.2e}, |plain - oracle| = {err_plain:.2e}")
            assert _cos(out["1"][p, r][:nref], ref).min() > 0.999, (p, r, _cos(out["1"][p, r][:nref], ref).min())
            assert err_fold < max(2.0 * err_plain, 5e-3), (p, r, err_fold, err_plain)


# ---- the reference's own two model shapes (heavy_ranker.py:80,83): MiniLM-L12 (a BERT: hidden 384, 12 heads of 32, FFN 1536,
# absolute position ids, mean pooling) and XLM-R base (vocab 250 002, 514 positions, mean pooling) ------------------------------
BERT_TINY = dict(vocab_size=120, hidden=64, layers=2, heads=2, ffn=128, max_pos=48, type_vocab=2, pad_id=0, ln_eps=1e-12,
                 position_ids="absolute")


def test_tiny_bert_fixture_head_size_32(native_lib, golden_dir):
    """HF BertModel's own output (tests/golden/enc_bert_tiny.npz: head size 32, absolute positions, two token types) through the HIP
    encoder: sentence-transformers mean pooling of the last hidden state."""
    from vietnamese_qa_system_amd.encoder import QuestionEncoder
    g = np.load(f"{golden_dir}/enc_bert_tiny.npz")
    w = {k[2:]: g[k] for k in g.files if k.startswith("w.")}
    enc = QuestionEncoder(w, BERT_TINY, max_tokens=64)
    ids, mask = g["input_ids"], g["attention_mask"]
    got = enc.forward(ids, mask, pooling="mean", normalize=False).cpu().numpy()
    assert np.abs(got - g["mean_pooled"]).max() < 2e-2 and _cos(got, g["mean_pooled"]).min() > 0.9995
    cls = enc.forward(ids, mask, pooling="cls", normalize=False).cpu().numpy()
    assert np.abs(cls - g["last_hidden_state"][:, 0]).max() < 2e-2
    enc.close()
    # the RoBERTa position rule on the same weights is measurably another model
    enc = QuestionEncoder(w, dict(BERT_TINY, position_ids="roberta"), max_tokens=64)
    other = enc.forward(ids, mask, pooling="mean", normalize=False).cpu().numpy()
    enc.close()
    assert np.abs(other - g["mean_pooled"]).max() > 2e-2


# (12, 6, 32): small call (register-staged GEMM: K = 384 is no multiple of 256), head size 32 at one query block; (2, 64, 32): 2048
# tokens on the LDS-DMA tile kernel; (1, 3, 128): four query blocks; (2, 2, 256): eight; (2, 5, 77): ragged
@pytest.mark.parametrize("layers,b,l,vocab", [(12, 6, 32, 250037), (2, 64, 32, 30000), (1, 3, 128, 30000), (2, 2, 256, 30000),
                                              (2, 5, 77, 30000)])
def test_minilm_l12_shape_vs_oracle(native_lib, layers, b, l, vocab):
    from vietnamese_qa_system_amd.encoder import MINILM_L12, QuestionEncoder
    assert MINILM_L12 == E.MINILM_L12
    cfg = dict(E.MINILM_L12, layers=layers, vocab_size=vocab)
    w = E.synthetic_weights(cfg, seed=6, layers=layers)
    ids, mask = E.synthetic_tokens(cfg, b, l, seed=10)
    ids[:, -1] = np.where(mask[:, -1] == 1, vocab - 1, ids[:, -1])  # the table's last row is reachable
    enc = QuestionEncoder(w, cfg, max_tokens=b * l)
    nref = min(b, 6)
    for pooling in ("mean", "cls"):
        got = enc.forward(ids, mask, pooling=pooling).cpu().numpy()
        ref = E.encode(w, cfg, ids[:nref], mask[:nref], pooling=pooling)
        assert _cos(got[:nref], ref).min() > 0.999, (pooling, _cos(got[:nref], ref).min())
        assert np.abs(got[:nref] - ref).max() < 2e-2
        assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-5
    enc.close()


def test_minilm_packed_batch_matches_the_padded_form(native_lib):
    """Head size 32 through the packed (ragged) path: B = 192, L = 32 -- the matrix-core attention kernel at DH = 32 takes every
    sequence's length from the packing offsets; same vectors as the padded form and the fp64 oracle."""
    from vietnamese_qa_system_amd.encoder import QuestionEncoder
    cfg = dict(E.MINILM_L12, layers=3, vocab_size=20000)
    w = E.synthetic_weights(cfg, seed=8, layers=3)
    b, l = 192, 32
    ids, mask = E.synthetic_tokens(cfg, b, l, seed=22)
    enc = QuestionEncoder(w, cfg, max_tokens=b * l)
    ids_d, mask_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    for pooling in ("mean", "cls"):
        padded = enc.forward(ids_d, mask_d, pooling=pooling).cpu().numpy()
        packed = enc.forward(ids, mask, pooling=pooling).cpu().numpy()
        assert np.abs(packed - padded).max() < 5e-4
        ref = E.encode(w, cfg, ids[:6], mask[:6], pooling=pooling)
        assert _cos(packed[:6], ref).min() > 0.999 and np.abs(packed[:6] - ref).max() < 2e-2
    enc.close()


@pytest.mark.parametrize("layers,b,l", [(12, 4, 128), (2, 2, 256), (2, 48, 40)])
def test_xlmr_base_shape_vs_oracle(native_lib, layers, b, l):
    """paraphrase-multilingual-mpnet-base-v2 (heavy_ranker.py:83) is an XLM-RoBERTa base: the PhoBERT-base layer shape behind a
    250 002-row vocabulary and 514 positions, mean pooling, sentences of up to 128 tokens and more."""
    from vietnamese_qa_system_amd.encoder import XLMR_BASE, QuestionEncoder
    assert XLMR_BASE == E.XLMR_BASE
    cfg = dict(E.XLMR_BASE, layers=layers)
    w = E.synthetic_weights(cfg, seed=12, layers=layers)
    ids, mask = E.synthetic_tokens(cfg, b, l, seed=14)
    ids[0, 1] = cfg["vocab_size"] - 1
    enc = QuestionEncoder(w, cfg, max_tokens=max(b * l, 520))
    nref = min(b, 4)
    for pooling in ("mean", "cls"):
        got = enc.forward(ids, mask, pooling=pooling).cpu().numpy()
        ref = E.encode(w, cfg, ids[:nref], mask[:nref], pooling=pooling)
        assert _cos(got[:nref], ref).min() > 0.999, (pooling, _cos(got[:nref], ref).min())
        assert np.abs(got[:nref] - ref).max() < 2e-2
    with pytest.raises(ValueError, match="position"):
        enc.forward(np.zeros((1, 513), np.int32) + 5, np.ones((1, 513), np.int32))  # needs position 514 of a 514-row table
    enc.close()

"""The encoder oracle (plain numpy, no transformers import) against the golden vectors HF RobertaModel /
DPRQuestionEncoder produced in the build container (tests/golden/make_golden_encoder.py)."""
import numpy as np

from oracle import encoder as E

TINY = dict(vocab_size=100, hidden=64, layers=2, heads=4, ffn=128, max_pos=40, type_vocab=1, pad_id=1, ln_eps=1e-5)


def _tiny(golden_dir):
    g = np.load(f"{golden_dir}/enc_tiny.npz")
    return g, {k[2:]: g[k] for k in g.files if k.startswith("w.")}


def test_tiny_roberta_last_hidden_state(golden_dir):
    g, w = _tiny(golden_dir)
    out = E.forward(w, TINY, g["input_ids"], g["attention_mask"])
    assert np.abs(out - g["last_hidden_state"]).max() < 5e-6


def test_tiny_dpr_pooler_output_is_cls(golden_dir):
    g, w = _tiny(golden_dir)
    pooled = E.encode(w, TINY, g["input_ids"], g["attention_mask"], pooling="cls", l2=False)
    assert np.abs(pooled - g["dpr_pooler_output"]).max() < 5e-6


def test_position_ids_skip_padding():
    ids = np.array([[0, 5, 6, 2, 1, 1], [0, 2, 1, 1, 1, 1]])
    assert E.position_ids(ids, 1).tolist() == [[2, 3, 4, 5, 1, 1], [2, 3, 1, 1, 1, 1]]


def test_phobert_shaped_layer(golden_dir):
    p = np.load(f"{golden_dir}/enc_phobert_layer.npz")
    cfg = dict(E.PHOBERT_BASE, layers=1)
    w = {k: v.astype(np.float64) for k, v in E.synthetic_weights(cfg, seed=1234, layers=1).items()}
    out = E.layer_forward(w, cfg, 0, p["hidden_in"].astype(np.float64), p["attention_mask"])
    assert np.abs(out - p["hidden_out"]).max() < 5e-6


def test_mean_pool_and_normalize():
    h = np.arange(24, dtype=np.float64).reshape(2, 3, 4)
    m = np.array([[1, 1, 0], [1, 0, 0]])
    assert np.allclose(E.pool(h, m, "mean"), [[2, 3, 4, 5], [12, 13, 14, 15]])
    n = E.normalize(np.array([[3.0, 4.0], [0.0, 0.0]]))
    assert np.allclose(n, [[0.6, 0.8], [0, 0]])


def test_synthetic_tokens_shape():
    ids, mask = E.synthetic_tokens(E.PHOBERT_BASE, 16, 32, seed=3)
    assert ids.shape == (16, 32) and (ids[:, 0] == 0).all()
    lens = mask.sum(1)
    assert lens.min() >= 8 and lens.max() <= 32
    assert all(ids[i, lens[i] - 1] == 2 and (ids[i, lens[i]:] == 1).all() for i in range(16))


BERT_TINY = dict(vocab_size=120, hidden=64, layers=2, heads=2, ffn=128, max_pos=48, type_vocab=2, pad_id=0, ln_eps=1e-12,
                 position_ids="absolute")


def test_tiny_bert_head_size_32_absolute_positions_mean_pooling(golden_dir):
    """HF BertModel (what paraphrase-multilingual-MiniLM-L12-v2 is, heavy_ranker.py:80) at head size 32: absolute position ids, two
    token types, pad id 0, LayerNorm eps 1e-12; sentence-transformers mean pooling."""
    g = np.load(f"{golden_dir}/enc_bert_tiny.npz")
    w = {k[2:]: g[k] for k in g.files if k.startswith("w.")}
    out = E.forward(w, BERT_TINY, g["input_ids"], g["attention_mask"])
    real = g["attention_mask"].astype(bool)
    assert np.abs(out - g["last_hidden_state"])[real].max() < 5e-6
    pooled = E.encode(w, BERT_TINY, g["input_ids"], g["attention_mask"], pooling="mean", l2=False)
    assert np.abs(pooled - g["mean_pooled"]).max() < 5e-6
    # the RoBERTa position rule on the same weights gives another answer: the mode is pinned, not incidental
    other = E.forward(w, dict(BERT_TINY, position_ids="roberta"), g["input_ids"], g["attention_mask"])
    assert np.abs(other - g["last_hidden_state"])[real].max() > 1e-2


def test_minilm_shaped_layer(golden_dir):
    p = np.load(f"{golden_dir}/enc_minilm_layer.npz")
    cfg = dict(E.MINILM_L12, layers=1, vocab_size=64, max_pos=16)
    w = {k: v.astype(np.float64) for k, v in E.synthetic_weights(cfg, seed=4321, layers=1).items()}
    out = E.layer_forward(w, cfg, 0, p["hidden_in"].astype(np.float64), p["attention_mask"])
    real = p["attention_mask"].astype(bool)
    assert np.abs(out - p["hidden_out"])[real].max() < 5e-6


def test_two_layer_models_from_token_ids_every_hidden_state(golden_dir):
    """HF output_hidden_states of 2-layer PhoBERT-base- and MiniLM-L12-shaped models run from token ids (12 x 32 ragged tokens; the
    first sequences stored): the oracle's embedding output and both layers, real positions."""
    for name, base, seed in (("enc_phobert_hidden.npz", E.PHOBERT_BASE, 2024), ("enc_minilm_hidden.npz", E.MINILM_L12, 2025)):
        g = np.load(f"{golden_dir}/{name}")
        cfg = dict(base, layers=2, vocab_size=2000)
        w = {k: v.astype(np.float64) for k, v in E.synthetic_weights(cfg, seed=seed, layers=2).items()}
        keep = g["hidden_0"].shape[0]
        ids, mask = g["input_ids"][:keep], g["attention_mask"][:keep]
        real = mask.astype(bool)
        x = E.embed(w, cfg, ids)
        assert np.abs(x - g["hidden_0"])[real].max() < 5e-6
        for i in (0, 1):
            x = E.layer_forward(w, cfg, i, x, mask)
            assert np.abs(x - g[f"hidden_{i + 1}"])[real].max() < 2e-5, (name, i)

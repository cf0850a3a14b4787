"""CPU oracle for the question-encoder forward (TEST INFRASTRUCTURE -- NOT PRODUCT CODE).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

What it restates: the transformer forward + pooling + L2-normalise that txtai runs for every query inside
``embeddings.search`` (reference call sites ``inference_pipeline/db_utils/heavy_ranker.py:98-101``; model chosen by
``path=`` at ``:80,83``; the DPR form the reference experimented with is ``q_model(input_ids).pooler_output`` at
``src/test.py:84-86``).  The arithmetic lives in HF ``transformers`` (pinned 4.33.1 in the reference's
``requirements.txt:62``; 5.15 is installed in the build container), so this file follows the published RoBERTa /
DPR algorithm in plain numpy -- it imports nothing from ``transformers``:

* embeddings = word + token_type(0) + position, ``position_ids = cumsum(ids != pad) * (ids != pad) + pad``
  (``transformers/models/roberta/modeling_roberta.py`` ``create_position_ids_from_input_ids``), then LayerNorm;
* per layer (post-LN): q/k/v Linear -> heads -> ``softmax(q k^T / sqrt(dh) + additive_mask) v`` -> out Linear +
  residual + LayerNorm -> Linear + GELU(erf) -> Linear + residual + LayerNorm;
* pooling: CLS row of the last layer (``modeling_dpr.py`` ``DPREncoder.forward``: ``sequence_output[:, 0, :]``, no
  projection when ``projection_dim = 0``) or sentence-transformers masked mean; then ``x / ||x||``.

PARITY STATUS: **parity unpinned by the reference** -- it holds no golden vectors for the encoder.  The oracle is checked
against outputs of HF ``RobertaModel`` / ``DPRQuestionEncoder`` of transformers 5.15 (what the build container has, NOT the
reference's pinned 4.33.1) run here (``tests/golden/make_golden_encoder.py`` -> ``enc_tiny.npz``, ``enc_phobert_layer.npz``):
a third-party stand-in, not a pin to the reference.
"""
from __future__ import annotations

import numpy as np
from scipy.special import erf

PREFIX = ""  # weights use HF RobertaModel state_dict names without a leading "roberta."


def position_ids(input_ids: np.ndarray, pad_id: int) -> np.ndarray:
    mask = (input_ids != pad_id).astype(np.int64)
    return np.cumsum(mask, axis=1) * mask + pad_id


def layer_norm(x: np.ndarray, g: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def gelu(x: np.ndarray) -> np.ndarray:
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def linear(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    return x @ w.T + b  # torch Linear: weight [out, in]


def embed(w: dict, cfg: dict, input_ids: np.ndarray) -> np.ndarray:
    # RoBERTa / XLM-R: pad-offset ids; BERT (``transformers/models/bert/modeling_bert.py`` ``BertEmbeddings``: the registered buffer
    # ``position_ids = arange(max_pos)[:, :L]``): the token's index -- paraphrase-multilingual-MiniLM-L12-v2 (heavy_ranker.py:80)
    if cfg.get("position_ids", "roberta") == "absolute":
        pos = np.broadcast_to(np.arange(input_ids.shape[1]), input_ids.shape)
    else:
        pos = position_ids(input_ids, cfg["pad_id"])
    x = (w["embeddings.word_embeddings.weight"][input_ids] + w["embeddings.position_embeddings.weight"][pos]
         + w["embeddings.token_type_embeddings.weight"][0])
    return layer_norm(x, w["embeddings.LayerNorm.weight"], w["embeddings.LayerNorm.bias"], cfg["ln_eps"])


def layer_forward(w: dict, cfg: dict, i: int, x: np.ndarray, attention_mask: np.ndarray) -> np.ndarray:
    p = f"encoder.layer.{i}."
    b, l, h = x.shape
    nh = cfg["heads"]
    dh = h // nh

    def heads(t):
        return t.reshape(b, l, nh, dh).transpose(0, 2, 1, 3)

    q = heads(linear(x, w[p + "attention.self.query.weight"], w[p + "attention.self.query.bias"]))
    k = heads(linear(x, w[p + "attention.self.key.weight"], w[p + "attention.self.key.bias"]))
    v = heads(linear(x, w[p + "attention.self.value.weight"], w[p + "attention.self.value.bias"]))
    s = q @ k.transpose(0, 1, 3, 2) / np.sqrt(dh)
    s = s + (1.0 - attention_mask[:, None, None, :].astype(x.dtype)) * np.finfo(np.float32).min
    s = s - s.max(axis=-1, keepdims=True)
    e = np.exp(s)
    ctx = (e / e.sum(axis=-1, keepdims=True)) @ v
    ctx = ctx.transpose(0, 2, 1, 3).reshape(b, l, h)
    a = linear(ctx, w[p + "attention.output.dense.weight"], w[p + "attention.output.dense.bias"])
    x = layer_norm(a + x, w[p + "attention.output.LayerNorm.weight"], w[p + "attention.output.LayerNorm.bias"], cfg["ln_eps"])
    f = gelu(linear(x, w[p + "intermediate.dense.weight"], w[p + "intermediate.dense.bias"]))
    f = linear(f, w[p + "output.dense.weight"], w[p + "output.dense.bias"])
    return layer_norm(f + x, w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], cfg["ln_eps"])


def forward(w: dict, cfg: dict, input_ids: np.ndarray, attention_mask: np.ndarray, dtype=np.float64) -> np.ndarray:
    """last_hidden_state [B, L, H]; computed in ``dtype`` (float64 by default: the GPU tolerance is stated against it)."""
    w = {k: np.asarray(v, dtype=dtype) for k, v in w.items()}
    x = embed(w, cfg, np.asarray(input_ids))
    for i in range(cfg["layers"]):
        x = layer_forward(w, cfg, i, x, np.asarray(attention_mask))
    return x


def pool(last_hidden: np.ndarray, attention_mask: np.ndarray, pooling: str) -> np.ndarray:
    if pooling == "cls":
        return last_hidden[:, 0, :]
    if pooling == "mean":
        m = attention_mask[:, :, None].astype(last_hidden.dtype)
        return (last_hidden * m).sum(axis=1) / np.clip(m.sum(axis=1), 1e-9, None)
    raise ValueError(pooling)


def normalize(x: np.ndarray) -> np.ndarray:
    n = np.linalg.norm(x, axis=-1, keepdims=True)
    return x / np.where(n == 0, 1.0, n)


def encode(w: dict, cfg: dict, input_ids, attention_mask, pooling: str = "cls", l2: bool = True, dtype=np.float64) -> np.ndarray:
    out = pool(forward(w, cfg, input_ids, attention_mask, dtype), np.asarray(attention_mask), pooling)
    return normalize(out) if l2 else out


# ---- seeded synthetic weights (PhoBERT-base shape by default): the SAME recipe is used by tests, bench and the fixture
# generator, so only inputs/outputs need to be committed -------------------------------------------------------------
PHOBERT_BASE = dict(vocab_size=64001, hidden=768, layers=12, heads=12, ffn=3072, max_pos=258, type_vocab=1, pad_id=1,
                    ln_eps=1e-5)
# the models heavy_ranker.py:80,83 load, by their published configs (XLM-RoBERTa base; a BERT with 12 heads of 32)
XLMR_BASE = dict(vocab_size=250002, hidden=768, layers=12, heads=12, ffn=3072, max_pos=514, type_vocab=1, pad_id=1, ln_eps=1e-5)
MINILM_L12 = dict(vocab_size=250037, hidden=384, layers=12, heads=12, ffn=1536, max_pos=512, type_vocab=2, pad_id=0, ln_eps=1e-12,
                  position_ids="absolute")


def synthetic_weights(cfg: dict, seed: int = 0, layers=None, std: float = 0.02) -> dict:
    """normal(0, std) matrices/embeddings, small random biases, LayerNorm gains around 1 (numpy PCG64: stable)."""
    rng = np.random.default_rng(seed)
    h, f = cfg["hidden"], cfg["ffn"]

    def mat(*shape):
        return (rng.standard_normal(shape) * std).astype(np.float32)

    w = {"embeddings.word_embeddings.weight": mat(cfg["vocab_size"], h),
         "embeddings.position_embeddings.weight": mat(cfg["max_pos"], h),
         "embeddings.token_type_embeddings.weight": mat(cfg["type_vocab"], h),
         "embeddings.LayerNorm.weight": (1.0 + mat(h)).astype(np.float32), "embeddings.LayerNorm.bias": mat(h)}
    for i in range(cfg["layers"] if layers is None else layers):
        p = f"encoder.layer.{i}."
        for name, shape in (("attention.self.query", (h, h)), ("attention.self.key", (h, h)), ("attention.self.value", (h, h)),
                            ("attention.output.dense", (h, h)), ("intermediate.dense", (f, h)), ("output.dense", (h, f))):
            w[p + name + ".weight"] = mat(*shape)
            w[p + name + ".bias"] = mat(shape[0])
        for name in ("attention.output.LayerNorm", "output.LayerNorm"):
            w[p + name + ".weight"] = (1.0 + mat(h)).astype(np.float32)
            w[p + name + ".bias"] = mat(h)
    return w


def synthetic_tokens(cfg: dict, b: int, l: int, seed: int = 0, min_len: int = 8):
    """ids uniform in [3, vocab), first token 0 (<s>), last real token 2 (</s>), ragged lengths
    uniform in [min_len, l] padded with pad_id."""
    rng = np.random.default_rng(seed)
    ids = rng.integers(3, cfg["vocab_size"], size=(b, l)).astype(np.int32)
    lens = rng.integers(min(min_len, l), l + 1, size=b)
    mask = (np.arange(l)[None, :] < lens[:, None]).astype(np.int32)
    ids[:, 0] = 0
    ids[np.arange(b), lens - 1] = 2
    ids = np.where(mask == 1, ids, cfg["pad_id"]).astype(np.int32)
    return ids, mask

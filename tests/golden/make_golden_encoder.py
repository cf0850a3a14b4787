"""Generates the encoder golden vectors by importing HF ``transformers`` ONCE in the build container (it cannot travel
to the GPU box; only these inputs/outputs and this script are committed).

* ``enc_tiny.npz``: a tiny RoBERTa (2 layers, hidden 64, 4 heads, FFN 128, vocab 100, 40 positions, pad 1) with
  seed-0 random weights: full state_dict, ragged ``input_ids`` / ``attention_mask``, HF ``last_hidden_state``, and the
  ``pooler_output`` of a ``DPRQuestionEncoder`` (projection_dim 0) loaded with the same weights mapped into its BERT.
* ``enc_phobert_layer.npz``: ONE PhoBERT-base-shaped layer (hidden 768, 12 heads, FFN 3072): ``hidden_in`` /
  ``attention_mask`` / HF ``hidden_out``.  Its weights are NOT stored: they come from ``oracle.encoder.synthetic_weights``
  (numpy PCG64 recipe, seed 1234), which tests re-run.

* ``enc_bert_tiny.npz`` (round 4): a tiny BERT -- what ``paraphrase-multilingual-MiniLM-L12-v2`` (``heavy_ranker.py:80``) is: absolute
  position ids, two token types, pad id 0, LayerNorm eps 1e-12 -- with HEAD SIZE 32 (hidden 64, 2 heads; 2 layers, FFN 128, vocab
  120, 48 positions): full state_dict, ragged inputs, HF ``BertModel`` ``last_hidden_state`` and the sentence-transformers masked
  mean of it.
* ``enc_minilm_layer.npz`` (round 4): ONE MiniLM-L12-shaped layer (hidden 384, 12 heads of 32, FFN 1536) of HF ``BertModel``:
  ``hidden_in`` / ``attention_mask`` / ``hidden_out``; weights from ``oracle.encoder.synthetic_weights`` (seed 4321), not stored.

* ``enc_phobert_hidden.npz`` / ``enc_minilm_hidden.npz`` (round 5): 2-layer models of the two shapes from TOKEN IDS (12 x 32 ragged
  tokens), HF ``output_hidden_states`` of the first sequences -- the per-layer golden of the HIP encoder (``hidden_states``).

    python tests/golden/make_golden_encoder.py [--hidden-only]
"""
import os
import sys

import numpy as np
import torch
from transformers import BertConfig, BertModel, DPRConfig, DPRQuestionEncoder, RobertaConfig, RobertaModel

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import encoder as E  # noqa: E402  (only for the seeded weight/token recipes, not for expected outputs)


def tiny():
    torch.manual_seed(0)
    cfg = RobertaConfig(vocab_size=100, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                        max_position_embeddings=40, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5,
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = RobertaModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():  # make biases / LayerNorm parameters non-trivial
        for name, p in model.named_parameters():
            if name.endswith("bias"):
                p.normal_(0, 0.05)
            elif "LayerNorm.weight" in name:
                p.normal_(1.0, 0.05)
    ids = torch.tensor([[0, 5, 17, 33, 8, 99, 41, 2, 1, 1, 1, 1],
                        [0, 9, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1],
                        [0, 7, 7, 7, 64, 12, 88, 3, 55, 21, 60, 2],
                        [0, 45, 31, 2, 1, 1, 1, 1, 1, 1, 1, 1]])
    mask = (ids != 1).long()
    with torch.no_grad():
        out = model(input_ids=ids, attention_mask=mask).last_hidden_state
    sd = {k: v.numpy() for k, v in model.state_dict().items() if "position_ids" not in k and "token_type_ids" not in k}
    # DPR question encoder = BERT encoder + CLS pooling (src/test.py:84-86 `.pooler_output`).  BERT positions are plain
    # arange, so feed the DPR model the RoBERTa position embedding rows it would pick (pad + cumsum) via position_ids.
    dcfg = DPRConfig(vocab_size=100, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                     max_position_embeddings=40, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5, projection_dim=0,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    dpr = DPRQuestionEncoder(dcfg).eval()
    bert = dpr.question_encoder.bert_model
    missing, unexpected = bert.load_state_dict(model.state_dict(), strict=False)
    assert not [m for m in missing if "pooler" not in m and "position_ids" not in m and "token_type_ids" not in m], missing
    pos = (torch.cumsum(mask, dim=1) * mask + 1)
    with torch.no_grad():
        hidden = bert(input_ids=ids, attention_mask=mask, position_ids=pos).last_hidden_state
        dpr_pooled = hidden[:, 0, :]  # DPREncoder.forward: pooled_output = sequence_output[:, 0, :]
    assert torch.allclose(hidden, out, atol=1e-5), "DPR's BERT and RoBERTa disagree on the same weights"
    np.savez(os.path.join(HERE, "enc_tiny.npz"), input_ids=ids.numpy().astype(np.int32), attention_mask=mask.numpy().astype(np.int32),
             last_hidden_state=out.numpy(), dpr_pooler_output=dpr_pooled.numpy(),
             **{"w." + k: v for k, v in sd.items()})


def phobert_layer():
    cfg = dict(E.PHOBERT_BASE, layers=1)
    w = E.synthetic_weights(cfg, seed=1234, layers=1)
    hf_cfg = RobertaConfig(vocab_size=cfg["vocab_size"], hidden_size=768, num_hidden_layers=1, num_attention_heads=12,
                           intermediate_size=3072, max_position_embeddings=258, type_vocab_size=1, pad_token_id=1,
                           layer_norm_eps=1e-5, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = RobertaModel(hf_cfg, add_pooling_layer=False).eval()
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in m or "token_type_ids" in m for m in missing), (missing, unexpected)
    rng = np.random.default_rng(77)
    hidden_in = rng.standard_normal((2, 16, 768)).astype(np.float32)
    mask = np.ones((2, 16), dtype=np.int32)
    mask[1, 11:] = 0
    layer = model.encoder.layer[0]
    ext = (1.0 - torch.from_numpy(mask)[:, None, None, :].float()) * torch.finfo(torch.float32).min
    with torch.no_grad():
        out = layer(torch.from_numpy(hidden_in), attention_mask=ext)
        out = out[0] if isinstance(out, tuple) else out
    np.savez(os.path.join(HERE, "enc_phobert_layer.npz"), hidden_in=hidden_in, attention_mask=mask, hidden_out=out.numpy())


def bert_tiny():
    torch.manual_seed(1)
    cfg = BertConfig(vocab_size=120, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
                     max_position_embeddings=48, type_vocab_size=2, pad_token_id=0, layer_norm_eps=1e-12,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = BertModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("bias"):
                p.normal_(0, 0.05)
            elif "LayerNorm.weight" in name:
                p.normal_(1.0, 0.05)
    ids = torch.tensor([[101, 5, 17, 33, 8, 99, 41, 102, 0, 0, 0, 0, 0, 0],
                        [101, 9, 102, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
                        [101, 7, 7, 7, 64, 12, 88, 3, 55, 21, 60, 119, 4, 102],
                        [101, 45, 31, 6, 102, 0, 0, 0, 0, 0, 0, 0, 0, 0]])
    mask = (ids != 0).long()
    with torch.no_grad():
        out = model(input_ids=ids, attention_mask=mask).last_hidden_state  # token_type_ids default to 0, position_ids to arange
    m = mask[:, :, None].float()
    mean = (out * m).sum(1) / m.sum(1).clamp(min=1e-9)  # sentence-transformers Pooling (mean of the real tokens)
    sd = {k: v.numpy() for k, v in model.state_dict().items() if "position_ids" not in k and "token_type_ids" not in k}
    np.savez(os.path.join(HERE, "enc_bert_tiny.npz"), input_ids=ids.numpy().astype(np.int32), attention_mask=mask.numpy().astype(np.int32),
             last_hidden_state=out.numpy(), mean_pooled=mean.numpy(), **{"w." + k: v for k, v in sd.items()})


def minilm_layer():
    cfg = dict(E.MINILM_L12, layers=1, vocab_size=64, max_pos=16)  # (the embedding tables play no part in a layer's in/out)
    w = E.synthetic_weights(cfg, seed=4321, layers=1)
    hf_cfg = BertConfig(vocab_size=64, hidden_size=384, num_hidden_layers=1, num_attention_heads=12, intermediate_size=1536,
                        max_position_embeddings=16, type_vocab_size=2, pad_token_id=0, layer_norm_eps=1e-12,
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = BertModel(hf_cfg, add_pooling_layer=False).eval()
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    assert not unexpected and all("position_ids" in m or "token_type_ids" in m for m in missing), (missing, unexpected)
    rng = np.random.default_rng(78)
    hidden_in = rng.standard_normal((2, 16, 384)).astype(np.float32)
    mask = np.ones((2, 16), dtype=np.int32)
    mask[0, 9:] = 0
    ext = (1.0 - torch.from_numpy(mask)[:, None, None, :].float()) * torch.finfo(torch.float32).min
    with torch.no_grad():
        out = model.encoder.layer[0](torch.from_numpy(hidden_in), attention_mask=ext)
        out = out[0] if isinstance(out, tuple) else out
    np.savez(os.path.join(HERE, "enc_minilm_layer.npz"), hidden_in=hidden_in, attention_mask=mask, hidden_out=out.numpy())


def hidden_states(name, cfg, seed, token_seed, b, l, keep, hf_model, hf_cfg):
    """Round 5: a 2-layer model of a reference shape FROM TOKEN IDS, HF ``output_hidden_states`` -- what the HIP encoder's
    ``vqa_encoder_forward_hidden`` is held to (the layer goldens above start from arbitrary hidden_in rows, which no sequence of
    token ids produces).  b x l tokens (384: the HIP side runs its LDS-DMA tile GEMMs with the folded LayerNorms); the hidden
    states of the first ``keep`` sequences are stored (fp32), weights come from the seeded recipe and are not stored."""
    w = E.synthetic_weights(cfg, seed=seed, layers=2)
    model = hf_model(hf_cfg, add_pooling_layer=False).eval()
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    assert not unexpected and all("position_ids" in m or "token_type_ids" in m for m in missing), (missing, unexpected)
    ids, mask = E.synthetic_tokens(cfg, b, l, seed=token_seed)
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids).long(), attention_mask=torch.from_numpy(mask).long(), output_hidden_states=True)
    hs = [h.numpy()[:keep] for h in out.hidden_states]
    assert len(hs) == 3 and np.array_equal(hs[2], out.last_hidden_state.numpy()[:keep])
    np.savez(os.path.join(HERE, name), input_ids=ids, attention_mask=mask, hidden_0=hs[0], hidden_1=hs[1], hidden_2=hs[2])


def phobert_hidden():
    cfg = dict(E.PHOBERT_BASE, layers=2, vocab_size=2000)
    hidden_states("enc_phobert_hidden.npz", cfg, 2024, 5, 12, 32, 3, RobertaModel,
                  RobertaConfig(vocab_size=2000, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072,
                                max_position_embeddings=258, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5,
                                hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))


def minilm_hidden():
    cfg = dict(E.MINILM_L12, layers=2, vocab_size=2000)
    hidden_states("enc_minilm_hidden.npz", cfg, 2025, 6, 12, 32, 4, BertModel,
                  BertConfig(vocab_size=2000, hidden_size=384, num_hidden_layers=2, num_attention_heads=12, intermediate_size=1536,
                             max_position_embeddings=512, type_vocab_size=2, pad_token_id=0, layer_norm_eps=1e-12,
                             hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))


if __name__ == "__main__":
    if "--hidden-only" not in sys.argv:
        tiny()
        phobert_layer()
        bert_tiny()
        minilm_layer()
    phobert_hidden()
    minilm_hidden()
    for f in sorted(os.listdir(HERE)):
        if f.startswith("enc_"):
            print(f, os.path.getsize(os.path.join(HERE, f)))

"""Writes two tiny model DIRECTORIES in the layouts ``Embeddings(path=...)`` has to read (heavy_ranker.py:78-83), from the weights
already committed in ``enc_tiny.npz`` / ``enc_bert_tiny.npz`` -- so the expected outputs are those files' HF outputs:

* ``hf_tiny_roberta_st/``: sentence-transformers layout (``modules.json`` -> Transformer at the root, ``1_Pooling/config.json``
  mean tokens, ``2_Normalize``), ``model.safetensors`` written by HF ``save_pretrained`` (names without a prefix);
* ``hf_tiny_bert_bin/``: a plain HF directory: ``config.json`` + ``pytorch_model.bin`` whose names carry the ``bert.`` prefix a
  task-head checkpoint has (plus a pooler and a head tensor the loader must skip).

Data only (config / weight files of 0.3 MB each); transformers is imported HERE, in the build container, never by the tests.

    python tests/golden/make_golden_checkpoints.py
"""
import json
import os
import shutil

import numpy as np
import torch
from transformers import BertConfig, BertModel, RobertaConfig, RobertaModel

HERE = os.path.dirname(os.path.abspath(__file__))


def roberta_st():
    g = np.load(os.path.join(HERE, "enc_tiny.npz"))
    cfg = RobertaConfig(vocab_size=100, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                        max_position_embeddings=40, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5,
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = RobertaModel(cfg, add_pooling_layer=False).eval()
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in m or "token_type_ids" in m for m in missing), (missing, unexpected)
    out = os.path.join(HERE, "hf_tiny_roberta_st")
    shutil.rmtree(out, ignore_errors=True)
    model.save_pretrained(out, safe_serialization=True)
    assert os.path.exists(os.path.join(out, "model.safetensors"))
    with open(os.path.join(out, "modules.json"), "w") as f:
        json.dump([{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
                   {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"},
                   {"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"}], f, indent=1)
    os.makedirs(os.path.join(out, "1_Pooling"))
    os.makedirs(os.path.join(out, "2_Normalize"))
    with open(os.path.join(out, "1_Pooling", "config.json"), "w") as f:
        json.dump({"word_embedding_dimension": 64, "pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True,
                   "pooling_mode_max_tokens": False, "pooling_mode_mean_sqrt_len_tokens": False}, f, indent=1)
    open(os.path.join(out, "2_Normalize", ".keep"), "w").close()


def bert_bin():
    g = np.load(os.path.join(HERE, "enc_bert_tiny.npz"))
    cfg = BertConfig(vocab_size=120, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
                     max_position_embeddings=48, type_vocab_size=2, pad_token_id=0, layer_norm_eps=1e-12,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    out = os.path.join(HERE, "hf_tiny_bert_bin")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    cfg.save_pretrained(out)
    sd = {"bert." + k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    sd["bert.embeddings.position_ids"] = torch.arange(48)[None]
    sd["bert.pooler.dense.weight"] = torch.zeros(64, 64)
    sd["cls.predictions.bias"] = torch.zeros(120)
    torch.save(sd, os.path.join(out, "pytorch_model.bin"))


if __name__ == "__main__":
    roberta_st()
    bert_bin()
    for root, _, files in os.walk(HERE):
        for f in files:
            if "hf_tiny" in root:
                print(os.path.relpath(os.path.join(root, f), HERE), os.path.getsize(os.path.join(root, f)))

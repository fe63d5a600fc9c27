"""Encoder golden vectors from transformers' local BertModel class (fixture generation, this container only).

    python -m oracle.gen_fixtures encoder

The model is built from a BertConfig (no download, no checkpoint: none exists offline) and loaded
with the seeded weights of oracle.bert_oracle.make_blob, so the committed fixture only needs the
inputs and the expected pooled outputs; tests regenerate the same weights from the seed.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

from . import bert_oracle as bo

GOLDEN = Path(__file__).resolve().parent.parent / "tests" / "golden"

CASES = {
    # name: (cfg overrides, seed, style, B, S, lens)
    "tiny": (dict(vocab=500, hidden=128, layers=2, heads=2, ffn=256, max_pos=64), 101, "test", 3, 32, [32, 17, 5]),
    "base1": (dict(layers=1), 102, "test", 2, 32, [32, 20]),
    "base12": (dict(), 103, "test", 2, 64, [64, 41]),
    "base12_bench_weights": (dict(), 0, "bench", 2, 128, [128, 77]),
}


def hf_forward(cfg: dict, blob: np.ndarray, ids: np.ndarray, lens: np.ndarray) -> np.ndarray:
    import torch
    from transformers import BertConfig
    from transformers.models.bert.modeling_bert import BertModel

    hc = BertConfig(vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"],
                    num_attention_heads=cfg["heads"], intermediate_size=cfg["ffn"], max_position_embeddings=cfg["max_pos"],
                    type_vocab_size=cfg["type_vocab"], layer_norm_eps=cfg["ln_eps"], hidden_act="gelu",
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = BertModel(hc, add_pooling_layer=False).eval()
    missing, unexpected = model.load_state_dict(bo.to_hf_state_dict(cfg, blob), strict=False)
    assert not unexpected and all("position_ids" in m or "token_type_ids" in m for m in missing), (missing, unexpected)
    S = ids.shape[1]
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
    with torch.no_grad():
        h = model(input_ids=torch.from_numpy(ids.astype(np.int64)), attention_mask=torch.from_numpy(mask)).last_hidden_state
    m = torch.from_numpy(mask).to(h.dtype)[:, :, None]
    return ((h * m).sum(1) / m.sum(1)).numpy().astype(np.float32)


def gen_encoder() -> None:
    out, meta = {}, {}
    for name, (over, seed, style, B, S, lens) in CASES.items():
        cfg = dict(bo.BERT_BASE)
        cfg.update(over)
        rng = np.random.default_rng(seed)
        ids = rng.integers(1, cfg["vocab"], size=(B, S)).astype(np.int32)
        lens = np.asarray(lens, dtype=np.int32)
        blob = bo.make_blob(cfg, seed, style)
        want = hf_forward(cfg, blob, ids, lens)
        got = bo.forward(cfg, blob, ids, lens, dtype=np.float64)
        err = float(np.abs(got - want).max())
        print(f"{name}: |numpy restatement - transformers|_max = {err:.3e}, |out|_max = {np.abs(want).max():.3f}")
        assert err < 2e-5, name  # the restatement is pinned by the independent implementation
        out[f"{name}_ids"], out[f"{name}_lens"], out[f"{name}_pooled"] = ids, lens, want
        meta[name] = {"cfg": cfg, "seed": seed, "style": style}
    np.savez_compressed(GOLDEN / "encoder_golden.npz", **out)
    (GOLDEN / "encoder_golden.json").write_text(json.dumps(meta, indent=1))
    print("wrote encoder_golden.npz / .json")

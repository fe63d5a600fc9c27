"""torch-CPU fp32 restatement of oracle/bert_oracle.forward (test infrastructure: bench.py's `cpu_baseline` leg and its test).

Same arithmetic as the numpy restatement -- which follows transformers' BertModel (embeddings -> LN; per layer Q/K/V linear + bias,
softmax(QK^T / sqrt(d_h) + additive key mask) V, out-proj + residual + LN, FFN1 + erf-GELU, FFN2 + residual + LN; masked mean
pooling), the forward llama.cpp runs behind LlamaCppEmbeddings.embed_documents (reference src/semcode/embeddings/providers.py:69-100,
called at src/semcode/services/indexer.py:150) -- but every matmul is ONE [tokens, K] x [K, N] sgemm over all chunks of the call and
every element-wise step a threaded torch op, so that the host's cores are actually used (CPU path
timed on all host cores).  Never imported by the product."""
from __future__ import annotations

import numpy as np

from . import bert_oracle as bo


def forward(cfg: dict, blob: np.ndarray, ids: np.ndarray, lens: np.ndarray, threads: int | None = None, chunk: int = 64) -> np.ndarray:
    """ids [B,S] int, lens [B] -> pooled [B,H] f32.  Attention is evaluated `chunk` chunks at a time (the [B, heads, S, S] score
    tensor of 256 chunks would be 800 MB); everything else runs on all B*S token rows at once."""
    import torch
    import torch.nn.functional as F

    if threads:
        torch.set_num_threads(int(threads))
    if cfg.get("alibi") or cfg.get("geglu"):
        raise ValueError("bert_torch.forward restates the BERT configuration only")
    W = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in bo.unpack(cfg, blob).items()}
    ids_t = torch.from_numpy(np.asarray(ids, dtype=np.int64))
    lens_t = torch.from_numpy(np.clip(np.asarray(lens), 1, ids.shape[1]).astype(np.int64))
    B, S = ids_t.shape
    H, nh = cfg["hidden"], cfg["heads"]
    dh = H // nh
    eps = cfg["ln_eps"]
    with torch.no_grad():
        x = W["word_emb"][ids_t] + W["pos_emb"][:S][None] + W["type_emb"][0][None, None]
        x = F.layer_norm(x, (H,), W["emb_ln_g"], W["emb_ln_b"], eps).reshape(B * S, H)
        key_ok = torch.arange(S)[None, :] < lens_t[:, None]  # [B,S]
        mask = torch.zeros((B, 1, 1, S), dtype=torch.float32).masked_fill(~key_ok[:, None, None, :], float("-inf"))
        for l in range(cfg["layers"]):
            p = f"l{l}."
            wqkv = torch.cat([W[p + "wq"], W[p + "wk"], W[p + "wv"]], 0)
            bqkv = torch.cat([W[p + "bq"], W[p + "bk"], W[p + "bv"]], 0)
            qkv = F.linear(x, wqkv, bqkv).reshape(B, S, 3, nh, dh)
            ctx = torch.empty((B, S, H), dtype=torch.float32)
            for b0 in range(0, B, chunk):
                q = qkv[b0:b0 + chunk, :, 0].permute(0, 2, 1, 3)
                k = qkv[b0:b0 + chunk, :, 1].permute(0, 2, 1, 3)
                v = qkv[b0:b0 + chunk, :, 2].permute(0, 2, 1, 3)
                s = torch.softmax(q @ k.transpose(-1, -2) / float(np.sqrt(dh)) + mask[b0:b0 + chunk], dim=-1)
                ctx[b0:b0 + chunk] = (s @ v).permute(0, 2, 1, 3).reshape(-1, S, H)
            x = F.layer_norm(F.linear(ctx.reshape(B * S, H), W[p + "wo"], W[p + "bo"]) + x, (H,), W[p + "ln1_g"], W[p + "ln1_b"], eps)
            h = F.gelu(F.linear(x, W[p + "w1"], W[p + "b1"]))  # erf form
            x = F.layer_norm(F.linear(h, W[p + "w2"], W[p + "b2"]) + x, (H,), W[p + "ln2_g"], W[p + "ln2_b"], eps)
        x = x.reshape(B, S, H)
        pooled = (x * key_ok[:, :, None]).sum(1) / lens_t[:, None]
    return pooled.numpy().astype(np.float32)

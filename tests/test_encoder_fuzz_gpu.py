"""GPU: the encoder forward on random model shapes, batch sizes, sequence buckets and chunk lengths, both pipelines, against the
numpy restatement (oracle/bert_oracle.py) at the tolerance of tests/test_encoder_gpu.py (cosine >= 0.999, |err| <= 2e-2 -- relative to the element where it exceeds 1, 5e-2 for chunks of fewer than 8 tokens).

The fixed cases of test_encoder_gpu.py sit on the shapes the bench uses; this sweep walks the edges between them: hidden sizes whose
tile counts are odd (384 = 1.5 tiles of 256), FFN widths that are multiples of 128 only, batches of 1 / 2 / 3 / 255 / 257 chunks,
chunks of one token next to full ones, every sequence bucket, ALiBi / GEGLU switches, and the pipeline the planner would not pick."""
import numpy as np
import pytest

from oracle import bert_oracle as bo
from semcode_amd import _native

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    r = _native.Runtime(device=0)
    yield r
    r.close()


def draw(rng):
    heads = int(rng.choice([2, 4, 6, 8, 12]))
    hidden = heads * 64
    ffn = int(rng.choice([128, 256, 384, 640, 1024, 1536, 3072]))
    layers = int(rng.integers(1, 4))
    S = int(rng.choice([32, 64, 128, 256, 512, 1024]))
    budget = 12_000 if hidden <= 384 else 5_000  # tokens: keeps the numpy forward of a case at a few seconds
    bmax = max(1, budget // S)
    B = int(rng.choice([b for b in (1, 2, 3, 7, 16, 33, 64, 65, 129, 255, 256, 257, 300) if b <= bmax] or [1]))
    alibi = bool(rng.random() < 0.3) or S > 512
    geglu = bool(rng.random() < 0.3)
    cfg = dict(bo.BERT_BASE, vocab=300, hidden=hidden, layers=layers, heads=heads, ffn=ffn, max_pos=max(64, S if not alibi else 64), alibi=alibi, geglu=geglu)
    lens = rng.integers(1, S + 1, size=B)
    lens[rng.integers(0, B)] = S
    lens[rng.integers(0, B)] = int(rng.choice([1, 2, S - 1, S // 2 + 1]))
    path = str(rng.choice(["small", "batch", "auto"]))
    return cfg, B, S, lens.astype(np.int32), path


@pytest.mark.parametrize("seed", range(24))
def test_random_encoder_shapes_match_restatement(rt, seed):
    rng = np.random.default_rng(1000 + seed)
    cfg, B, S, lens, path = draw(rng)
    blob = bo.make_blob(cfg, seed, "test")
    ids = rng.integers(1, cfg["vocab"], size=(B, S)).astype(np.int32)
    want = bo.forward(cfg, blob, ids, lens)
    enc = _native.Encoder(rt, cfg, weights=blob)
    enc.set_path(path)
    got = enc.embed_ids(ids, lens)
    enc.close()
    tag = f"hidden {cfg['hidden']} ffn {cfg['ffn']} layers {cfg['layers']} B {B} S {S} alibi {cfg['alibi']} geglu {cfg['geglu']} path {path}"
    assert np.isfinite(got).all(), tag
    cos = (got * want).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    assert cos.min() >= 0.999, (tag, float(cos.min()), int(cos.argmin()), int(lens[cos.argmin()]))
    # 2e-2 is stated for unit-scale outputs; make_blob(style="test") has LayerNorm gains that put single elements of short chunks at
    # 2 - 4, where one bf16 rounding alone is 2^-7 ... 2^-6: the bound scales with the element beyond 1
    err = np.abs(got - want)
    # ... and chunks of fewer than 8 tokens get 5e-2: the mean pool averages the per-token bf16 noise of the residual stream down by
    # sqrt(len); a chunk of one or two tokens shows it whole (measured: 0.022 - 0.048 at len 1 - 2, cosine still >= 0.999)
    bound = np.where(lens[:, None] < 8, 5e-2, 2e-2) * np.maximum(1.0, np.abs(want))
    at = np.unravel_index(np.argmax(err - bound), err.shape)
    assert (err <= bound).all(), (tag, float(err[at]), float(want[at]), int(lens[at[0]]))

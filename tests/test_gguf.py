"""GGUF weights (the reference's local model format: settings.embedding_llamacpp_model_path, providers.py:77-99).

No GGUF file and no llama.cpp exist offline ("parity unpinned" against a real jina / BERT GGUF): the test writes files with
llama.cpp's tensor names and metadata keys for the bert and jina-bert-v2 architectures from the oracle's seeded blob, and the
loader must give back that blob -- bit for bit for F32, rounded exactly as the file rounds for F16 / BF16 -- plus the
vocabulary in vocab.txt's convention.  On the GPU the encoder built from the file equals the encoder built from the blob."""
import struct

import numpy as np
import pytest

from oracle import bert_oracle as bo
from semcode_amd.embeddings import gguf
from semcode_amd.embeddings.providers import load_weight_blob

TINY = dict(bo.BERT_BASE, vocab=300, hidden=128, layers=2, heads=2, ffn=256, max_pos=64)


def gguf_tensors(cfg, blob):
    """the oracle blob as llama.cpp-named tensors (the inverse of gguf_to_blob's mapping, written independently of it)"""
    u = bo.unpack(cfg, blob)
    F = cfg["ffn"]
    t = {"token_embd.weight": u["word_emb"], "token_types.weight": u["type_emb"], "token_embd_norm.weight": u["emb_ln_g"], "token_embd_norm.bias": u["emb_ln_b"]}
    if not cfg.get("alibi"):
        t["position_embd.weight"] = u["pos_emb"]
    for l in range(cfg["layers"]):
        p, b = f"l{l}.", f"blk.{l}."
        for ours, theirs in (("q", "attn_q"), ("k", "attn_k"), ("v", "attn_v"), ("o", "attn_output")):
            t[b + theirs + ".weight"], t[b + theirs + ".bias"] = u[p + "w" + ours], u[p + "b" + ours]
        t[b + "attn_output_norm.weight"], t[b + "attn_output_norm.bias"] = u[p + "ln1_g"], u[p + "ln1_b"]
        if cfg.get("geglu"):
            t[b + "ffn_gate.weight"], t[b + "ffn_up.weight"] = u[p + "w1"][:F], u[p + "w1"][F:]
        else:
            t[b + "ffn_up.weight"], t[b + "ffn_up.bias"] = u[p + "w1"], u[p + "b1"]
        t[b + "ffn_down.weight"], t[b + "ffn_down.bias"] = u[p + "w2"], u[p + "b2"]
        t[b + "layer_output_norm.weight"], t[b + "layer_output_norm.bias"] = u[p + "ln2_g"], u[p + "ln2_b"]
    return t


def gguf_meta(cfg, tokens=None):
    arch = "jina-bert-v2" if cfg.get("geglu") else "bert"
    m = {"general.architecture": arch, "general.name": "test", f"{arch}.block_count": cfg["layers"], f"{arch}.embedding_length": cfg["hidden"],
         f"{arch}.feed_forward_length": cfg["ffn"], f"{arch}.attention.head_count": cfg["heads"], f"{arch}.context_length": cfg["max_pos"],
         f"{arch}.attention.layer_norm_epsilon": float(cfg["ln_eps"]), f"{arch}.attention.causal": False}
    if tokens is not None:
        m["tokenizer.ggml.model"] = "bert"
        m["tokenizer.ggml.tokens"] = tokens
        m["tokenizer.ggml.token_type"] = [3 if t.startswith("[") else 1 for t in tokens]
    return m


@pytest.mark.parametrize("arch", ["bert", "jina"])
@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_gguf_file_gives_the_source_blob_back(tmp_path, arch, dtype):
    cfg = dict(TINY, alibi=True, geglu=True) if arch == "jina" else dict(TINY)
    blob = bo.make_blob(cfg, 11, "test")
    if arch == "jina":  # the gated feed-forward has no bias in the file: the loader fills zeros
        u = bo.unpack(cfg, blob)
        for l in range(cfg["layers"]):
            u[f"l{l}.b1"][:] = 0.0
    path = tmp_path / f"m-{arch}-{dtype}.gguf"
    gguf.write_gguf(path, gguf_meta(cfg), gguf_tensors(cfg, blob), dtype=dtype)
    got = load_weight_blob(path, cfg["layers"], cfg)
    if dtype == "f32":
        want = blob
    elif dtype == "f16":
        want = blob.astype(np.float16).astype(np.float32)
    else:
        u32 = blob.view(np.uint32).astype(np.uint64)
        want = (((u32 + 0x7FFF + ((u32 >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
    assert got.dtype == np.float32 and got.shape == blob.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    meta, _ = gguf.read_gguf(path)
    fcfg = gguf.gguf_config(meta)
    assert (fcfg["hidden"], fcfg["layers"], fcfg["heads"], fcfg["ffn"], fcfg["alibi"], fcfg["geglu"]) == (128, 2, 2, 256, arch == "jina", arch == "jina")


def test_gguf_vocabulary_comes_back_in_vocab_txt_convention(tmp_path):
    wordpiece = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "def", "##ine", "return", "##s", "x", "(", ")", "λ", "##λ"]
    stored = [t if t.startswith("[") else (t[2:] if t.startswith("##") else "▁" + t) for t in wordpiece]  # llama.cpp's converter
    cfg = dict(TINY, vocab=len(wordpiece))
    path = tmp_path / "v.gguf"
    gguf.write_gguf(path, gguf_meta(cfg, stored), gguf_tensors(cfg, bo.make_blob(cfg, 3, "test")))
    meta, _ = gguf.read_gguf(path)
    assert gguf.gguf_vocab(meta) == wordpiece


def test_gguf_reader_refuses_what_it_cannot_read(tmp_path):
    cfg = dict(TINY)
    blob = bo.make_blob(cfg, 5, "test")
    good = tmp_path / "g.gguf"
    gguf.write_gguf(good, gguf_meta(cfg), gguf_tensors(cfg, blob))
    raw = bytearray(good.read_bytes())
    # not GGUF / future version / truncated
    (tmp_path / "a.gguf").write_bytes(b"GGML" + bytes(raw[4:]))
    with pytest.raises(gguf.GGUFError):
        gguf.read_gguf(tmp_path / "a.gguf")
    (tmp_path / "b.gguf").write_bytes(bytes(raw[:4]) + struct.pack("<I", 9) + bytes(raw[8:]))
    with pytest.raises(gguf.GGUFError):
        gguf.read_gguf(tmp_path / "b.gguf")
    (tmp_path / "c.gguf").write_bytes(bytes(raw[: len(raw) // 2]))
    with pytest.raises(gguf.GGUFError):
        gguf.read_gguf(tmp_path / "c.gguf")
    # a quantised tensor type (Q4_0 = 2) is refused by name
    t = gguf_tensors(cfg, blob)
    gguf.write_gguf(tmp_path / "q.gguf", gguf_meta(cfg), t)
    q = bytearray((tmp_path / "q.gguf").read_bytes())
    name = b"token_embd.weight"
    i = q.index(name) + len(name) + 4 + 16  # n_dims u32 + two u64 dims -> the type field
    q[i:i + 4] = struct.pack("<I", 2)
    (tmp_path / "q.gguf").write_bytes(bytes(q))
    with pytest.raises(gguf.GGUFError, match="ggml type 2"):
        gguf.read_gguf(tmp_path / "q.gguf")
    # architecture / shape disagreements with the encoder configuration
    with pytest.raises(ValueError, match="layers"):
        load_weight_blob(good, 3, dict(cfg, layers=3))
    with pytest.raises(ValueError, match="BERT"):
        load_weight_blob(good, 2, dict(cfg, alibi=True, geglu=True))
    del t["blk.1.ffn_down.bias"]
    gguf.write_gguf(tmp_path / "m.gguf", gguf_meta(cfg), t)
    with pytest.raises(gguf.GGUFError, match="ffn_down.bias"):
        load_weight_blob(tmp_path / "m.gguf", 2, cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ["bert", "jina"])
def test_encoder_from_gguf_equals_encoder_from_blob(tmp_path, rt, arch):
    """The provider pointed at a .gguf (through the reference's own setting name) builds the same encoder as the blob path:
    identical output bits, configuration and vocabulary taken from the file."""
    from semcode_amd import _native
    from semcode_amd.embeddings.providers import MI355XEmbeddings

    cfg = dict(TINY, alibi=True, geglu=True) if arch == "jina" else dict(TINY)
    blob = bo.make_blob(cfg, 17, "test")
    if arch == "jina":
        u = bo.unpack(cfg, blob)
        for l in range(cfg["layers"]):
            u[f"l{l}.b1"][:] = 0.0
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"w{i}" for i in range(200)] + [f"##s{i}" for i in range(95)]
    stored = [t if t.startswith("[") else (t[2:] if t.startswith("##") else "▁" + t) for t in words]
    path = tmp_path / f"enc-{arch}.gguf"
    gguf.write_gguf(path, gguf_meta(cfg, stored), gguf_tensors(cfg, blob))
    emb = MI355XEmbeddings(weights=path, runtime=rt)
    assert emb._cfg["hidden"] == 128 and emb._cfg["layers"] == 2 and bool(emb._cfg.get("alibi")) == (arch == "jina")
    assert emb.tokenizer.vocab["w7"] == 12 and emb.tokenizer.vocab["##s3"] == 208
    ref = _native.Encoder(rt, cfg, weights=blob)
    ids = np.random.default_rng(1).integers(5, 300, size=(6, 32)).astype(np.int32)
    lens = np.array([32, 32, 17, 9, 2, 1], np.int32)
    got, want = emb.embed_ids_array(ids, lens), ref.embed_ids(ids, lens)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    vec = emb.embed_documents(["w1 w2s3 w9", "w100"])
    assert len(vec) == 2 and len(vec[0]) == 128
    emb.close()
    ref.close()

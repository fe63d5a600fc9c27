"""Weights-only GGUF reader for the MI355X encoder.

Reference: the local model path of semcode is a GGUF file -- `settings.embedding_llamacpp_model_path`
(src/semcode/settings.py:51) handed to llama.cpp as `model_path` (src/semcode/embeddings/providers.py:77-99).  A checkout
that points that setting at `jina-embeddings-v2-base-en.gguf` can keep the file: this module reads the tensors (and the
WordPiece vocabulary stored next to them) and lays them out in the encoder's blob order (include/semcode_hip.h,
sc_encoder_blob_bytes).

The file is parsed as data and nothing in it is executed: header, key/value metadata, tensor infos, then raw little-endian
tensor bytes reached through a read-only numpy memmap.  Tensor types F32, F16 and BF16 only -- quantised GGUF types
(Q4_0, Q8_0, K-quants ...) are refused: the encoder computes in bf16 from f32 masters and de-quantising llama.cpp's block
formats is a different loader.

Format (GGUF v2 / v3, the ggml specification): magic "GGUF", u32 version, u64 tensor count, u64 KV count; KV = string key,
u32 value type, value; string = u64 length + UTF-8 bytes; array = u32 element type + u64 count + elements; tensor info = string
name, u32 n_dims, u64 dims[n_dims] (dims[0] is the FASTEST varying: a torch [out, in] matrix is stored as dims = {in, out}), u32
ggml type, u64 offset from the start of the data section; the data section starts at the next multiple of `general.alignment`
(32 when absent) after the last tensor info.

Tensor names: llama.cpp's `bert` / `jina-bert-v2` architectures (gguf-py tensor_mapping): token_embd, position_embd,
token_types, token_embd_norm, blk.N.{attn_q, attn_k, attn_v, attn_output, attn_output_norm, ffn_up, ffn_gate, ffn_down,
layer_output_norm}.  jina-bert-v2 has no position table (ALiBi) and a gated feed-forward without bias: the converter splits the
checkpoint's `mlp.gated_layers` [2F, H] into ffn_gate (first F rows, the activated half) and ffn_up; the blob wants them
stacked again, gate first.  PARITY STATUS: no GGUF file and no llama.cpp exist offline, so the name mapping and the vocabulary
convention below are restated from the published converter and exercised only on files this repository writes
(tests/test_gguf.py) -- "parity unpinned" against a real jina / BERT GGUF.
"""
from __future__ import annotations

import struct
from pathlib import Path
from typing import Any, Optional

import numpy as np

GGUF_MAGIC = b"GGUF"
GGML_F32, GGML_F16, GGML_BF16 = 0, 1, 30
_SCALAR = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<?", 10: "<Q", 11: "<q", 12: "<d"}
_T_STRING, _T_ARRAY = 8, 9
_MAX_STRING = 1 << 26  # a corrupt length must not turn into a 2^60-byte read


class GGUFError(ValueError):
    pass


class _Reader:
    def __init__(self, buf: memoryview):
        self.b, self.o = buf, 0

    def take(self, n: int) -> memoryview:
        if n < 0 or self.o + n > len(self.b):
            raise GGUFError("GGUF file truncated")
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def scalar(self, fmt: str):
        return struct.unpack(fmt, self.take(struct.calcsize(fmt)))[0]

    def string(self) -> str:
        n = self.scalar("<Q")
        if n > _MAX_STRING:
            raise GGUFError(f"GGUF string of {n} bytes")
        return bytes(self.take(n)).decode("utf-8", errors="replace")

    def value(self, t: int, depth: int = 0):
        if t in _SCALAR:
            return self.scalar(_SCALAR[t])
        if t == _T_STRING:
            return self.string()
        if t == _T_ARRAY:
            if depth > 2:
                raise GGUFError("GGUF arrays nested too deeply")
            et, n = self.scalar("<I"), self.scalar("<Q")
            if n > (1 << 28):
                raise GGUFError(f"GGUF array of {n} elements")
            if et in _SCALAR and et != 7:  # numeric arrays in one read
                fmt = _SCALAR[et]
                return np.frombuffer(self.take(n * struct.calcsize(fmt)), dtype=np.dtype(fmt)).copy()
            return [self.value(et, depth + 1) for _ in range(n)]
        raise GGUFError(f"unknown GGUF value type {t}")


def read_gguf(path: "str | Path") -> "tuple[dict, dict]":
    """(metadata, tensors): metadata = the KV section as a dict; tensors[name] = (ggml type, numpy shape (slowest first), array view
    into a read-only memmap of the file, in the file's own dtype: float32, float16, or uint16 holding bf16 bits)."""
    path = Path(path)
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    r = _Reader(memoryview(mm))
    if bytes(r.take(4)) != GGUF_MAGIC:
        raise GGUFError(f"{path}: not a GGUF file")
    version = r.scalar("<I")
    if version not in (2, 3):
        raise GGUFError(f"{path}: GGUF version {version} is not supported (2 and 3 are)")
    n_tensors, n_kv = r.scalar("<Q"), r.scalar("<Q")
    if n_tensors > (1 << 20) or n_kv > (1 << 20):
        raise GGUFError(f"{path}: implausible counts ({n_tensors} tensors, {n_kv} keys)")
    meta: dict = {}
    for _ in range(n_kv):
        key = r.string()
        meta[key] = r.value(r.scalar("<I"))
    infos = []
    for _ in range(n_tensors):
        name = r.string()
        nd = r.scalar("<I")
        if nd > 4:
            raise GGUFError(f"{path}: tensor {name!r} has {nd} dimensions")
        dims = [r.scalar("<Q") for _ in range(nd)]
        infos.append((name, dims, r.scalar("<I"), r.scalar("<Q")))
    align = int(meta.get("general.alignment", 32)) or 32
    data0 = (r.o + align - 1) // align * align
    tensors: dict = {}
    for name, dims, gtype, off in infos:
        if gtype not in (GGML_F32, GGML_F16, GGML_BF16):
            raise GGUFError(f"{path}: tensor {name!r} has ggml type {gtype}; only F32 (0), F16 (1) and BF16 (30) are read -- "
                            "convert the model with --outtype f32 / f16 / bf16")
        dt = {GGML_F32: np.dtype("<f4"), GGML_F16: np.dtype("<f2"), GGML_BF16: np.dtype("<u2")}[gtype]
        count = int(np.prod(dims, dtype=np.uint64)) if dims else 1
        lo = data0 + off
        if lo + count * dt.itemsize > mm.size:
            raise GGUFError(f"{path}: tensor {name!r} reaches beyond the end of the file")
        arr = np.frombuffer(mm, dtype=dt, count=count, offset=lo)
        tensors[name] = (gtype, tuple(int(d) for d in reversed(dims)), arr)
    meta["_version"] = version
    return meta, tensors


def tensor_f32(tensors: dict, name: str) -> np.ndarray:
    gtype, shape, arr = tensors[name]
    if gtype == GGML_BF16:
        out = (arr.astype(np.uint32) << 16).view(np.float32)
    else:
        out = arr.astype(np.float32)
    return out.reshape(shape)


def gguf_config(meta: dict) -> dict:
    """Encoder configuration stated by the file's metadata (keys of llama.cpp's bert / jina-bert-v2 architectures)."""
    arch = str(meta.get("general.architecture", ""))
    if arch not in ("bert", "jina-bert-v2", "nomic-bert"):
        raise GGUFError(f"GGUF architecture {arch!r} is not an encoder this backend runs (bert, jina-bert-v2)")
    if arch == "nomic-bert":
        raise GGUFError("nomic-bert (rotary positions, SwiGLU) is not implemented by this backend")

    def need(key: str):
        if f"{arch}.{key}" not in meta:
            raise GGUFError(f"GGUF metadata lacks {arch}.{key}")
        return meta[f"{arch}.{key}"]

    jina = arch == "jina-bert-v2"
    cfg = {"hidden": int(need("embedding_length")), "layers": int(need("block_count")), "heads": int(need("attention.head_count")),
           "ffn": int(need("feed_forward_length")), "max_pos": int(meta.get(f"{arch}.context_length", 512)),
           "ln_eps": float(meta.get(f"{arch}.attention.layer_norm_epsilon", 1e-12)), "alibi": jina, "geglu": jina}
    return cfg


def gguf_vocab(meta: dict) -> Optional[list]:
    """The WordPiece vocabulary, one token per id, in vocab.txt's convention.  llama.cpp's converter stores BERT vocabularies
    'phantom-space' style: a word-initial piece gets a leading U+2581, a continuation piece loses its '##'; special tokens
    ([PAD], [CLS] ...: token type 3 = control) are stored as they are."""
    toks = meta.get("tokenizer.ggml.tokens")
    if toks is None:
        return None
    types = meta.get("tokenizer.ggml.token_type")
    out = []
    for i, t in enumerate(toks):
        control = types is not None and i < len(types) and int(types[i]) == 3
        if control or (t.startswith("[") and t.endswith("]") and len(t) > 2):
            out.append(t)
        elif t.startswith("▁"):
            out.append(t[1:])
        else:
            out.append("##" + t)
    return out


def gguf_to_blob(path: "str | Path", cfg: Optional[dict] = None) -> "tuple[np.ndarray, dict, dict]":
    """(flat f32 blob in ABI order, configuration derived from the file, metadata).  cfg: the configuration the caller is about to
    build the encoder with -- every field the file states must agree with it."""
    meta, T = read_gguf(path)
    fcfg = gguf_config(meta)
    jina = fcfg["geglu"]

    def get(name: str) -> np.ndarray:
        if name not in T:
            raise GGUFError(f"{path}: tensor {name!r} not found")
        return tensor_f32(T, name)

    word = get("token_embd.weight")
    fcfg["vocab"] = int(word.shape[0])
    types = get("token_types.weight")
    fcfg["type_vocab"] = int(types.shape[0])
    if cfg is not None:
        for key in ("hidden", "layers", "heads", "ffn", "vocab", "type_vocab"):
            if key in cfg and int(cfg[key]) != int(fcfg[key]):
                raise ValueError(f"{path}: the file says {key} = {fcfg[key]}, the encoder configuration {cfg[key]}")
        if bool(cfg.get("alibi")) != jina or bool(cfg.get("geglu")) != jina:
            raise ValueError(f"{path} is a {'jina-bert-v2 (ALiBi + GEGLU)' if jina else 'BERT'} model but the encoder configuration says "
                             f"alibi={bool(cfg.get('alibi'))}, geglu={bool(cfg.get('geglu'))}")
    H, F = fcfg["hidden"], fcfg["ffn"]
    parts = [word.reshape(-1)]
    if not jina:
        pos = get("position_embd.weight")
        want = int(cfg["max_pos"]) if cfg and "max_pos" in cfg else int(pos.shape[0])
        if pos.shape[0] < want:
            raise ValueError(f"{path}: position table has {pos.shape[0]} rows, the encoder configuration wants {want}")
        fcfg["max_pos"] = want
        parts.append(pos[:want].reshape(-1))
    parts += [types.reshape(-1), get("token_embd_norm.weight").reshape(-1), get("token_embd_norm.bias").reshape(-1)]
    for l in range(fcfg["layers"]):
        p = f"blk.{l}."
        for n in ("attn_q", "attn_k", "attn_v", "attn_output"):
            parts += [get(p + n + ".weight").reshape(-1), get(p + n + ".bias").reshape(-1)]
        parts += [get(p + "attn_output_norm.weight").reshape(-1), get(p + "attn_output_norm.bias").reshape(-1)]
        if jina:  # gate (the activated half) first, then up; no bias
            parts += [get(p + "ffn_gate.weight").reshape(-1), get(p + "ffn_up.weight").reshape(-1), np.zeros(2 * F, np.float32)]
        else:
            parts += [get(p + "ffn_up.weight").reshape(-1), get(p + "ffn_up.bias").reshape(-1)]
        parts += [get(p + "ffn_down.weight").reshape(-1), get(p + "ffn_down.bias").reshape(-1),
                  get(p + "layer_output_norm.weight").reshape(-1), get(p + "layer_output_norm.bias").reshape(-1)]
    blob = np.concatenate(parts).astype(np.float32, copy=False)
    return blob, fcfg, meta


# ------------------------------------------------------------------------------------------------ writer (tests, conversions)

def write_gguf(path: "str | Path", meta: dict, tensors: "dict[str, np.ndarray]", dtype: str = "f32", alignment: int = 32) -> None:
    """Minimal GGUF v3 writer: metadata values may be str, bool, int, float, or lists of str / int / float; tensors are numpy arrays
    (numpy shape = slowest dimension first) stored as F32, F16 or BF16.  Used by the tests and to convert a checkpoint that exists in
    another format; a real GGUF comes from llama.cpp's converter."""
    gtype = {"f32": GGML_F32, "f16": GGML_F16, "bf16": GGML_BF16}[dtype]

    def s(x: str) -> bytes:
        b = x.encode("utf-8")
        return struct.pack("<Q", len(b)) + b

    def val(v: Any) -> bytes:
        if isinstance(v, bool):
            return struct.pack("<I?", 7, v)
        if isinstance(v, (int, np.integer)):
            return struct.pack("<II", 4, int(v)) if 0 <= int(v) < (1 << 32) else struct.pack("<Iq", 11, int(v))
        if isinstance(v, (float, np.floating)):
            return struct.pack("<If", 6, float(v))
        if isinstance(v, str):
            return struct.pack("<I", _T_STRING) + s(v)
        if isinstance(v, (list, tuple, np.ndarray)):
            v = list(v)
            if all(isinstance(e, str) for e in v):
                return struct.pack("<IIQ", _T_ARRAY, _T_STRING, len(v)) + b"".join(s(e) for e in v)
            if all(isinstance(e, (int, np.integer)) for e in v):
                return struct.pack("<IIQ", _T_ARRAY, 5, len(v)) + np.asarray(v, "<i4").tobytes()
            return struct.pack("<IIQ", _T_ARRAY, 6, len(v)) + np.asarray(v, "<f4").tobytes()
        raise TypeError(f"cannot store {type(v)} in GGUF metadata")

    meta = dict(meta)
    meta.setdefault("general.alignment", alignment)
    head = GGUF_MAGIC + struct.pack("<IQQ", 3, len(tensors), len(meta))
    kv = b"".join(s(k) + val(v) for k, v in meta.items())
    infos, blobs, off = b"", [], 0
    for name, a in tensors.items():
        a = np.ascontiguousarray(a, dtype=np.float32)
        if gtype == GGML_F32:
            raw = a.astype("<f4").tobytes()
        elif gtype == GGML_F16:
            raw = a.astype("<f2").tobytes()
        else:  # bf16, round to nearest even
            u = a.view(np.uint32).astype(np.uint64)
            raw = (((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype("<u2")).tobytes()
        dims = list(reversed(a.shape))
        infos += s(name) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims) + struct.pack("<IQ", gtype, off)
        pad = (-len(raw)) % alignment
        blobs.append(raw + b"\0" * pad)
        off += len(raw) + pad
    body = head + kv + infos
    body += b"\0" * ((-len(body)) % alignment)
    Path(path).write_bytes(body + b"".join(blobs))

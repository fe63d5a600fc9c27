"""Embedding provider factory with the MI355X encoder as a provider.

Reference: src/semcode/embeddings/providers.py:21-104.  `EmbeddingPayload` and
`EmbeddingProviderFactory.create(provider=None, model=None)` keep their names, signature, dispatch on the
lower-cased provider name and error types (NotImplementedError for unknown providers, ValueError when
the llama.cpp model path is unset, RuntimeError when llama-cpp-python is missing), so
IndexerService._embedding_client_instance (src/semcode/services/indexer.py:180-183) and
SemanticSearchPipeline._embedding_client (src/semcode/rag/pipeline.py:298-301) work unchanged.  New
provider names "mi355x" / "hip" select the HIP encoder (settings.embedding_provider = "mi355x").

The returned object has LangChain's Embeddings surface -- embed_documents(list[str]) -> list[list[float]],
embed_query(str) -> list[float] (duck-typed in reference tests/integration/test_indexer_service.py:7-12)
-- plus array fast paths that skip the list-of-float boxing.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Any, List, Optional, Sequence

import numpy as np

from ..settings import resolve as _resolve_settings
from .payload import EmbeddingPayload
from .tokenizer import HashTokenizer, WordPieceTokenizer, pack

log = logging.getLogger(__name__)

__all__ = ["EmbeddingPayload", "EmbeddingProviderFactory", "MI355XEmbeddings"]

MI355X_PROVIDER_NAMES = {"mi355x", "hip", "rocm"}

# HF BertModel checkpoint names -> blob order of include/semcode_hip.h (sc_encoder_blob_bytes)
_HF_HEAD = ["embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight", "embeddings.token_type_embeddings.weight",
            "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"]
_HF_LAYER = ["attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight", "attention.self.key.bias",
             "attention.self.value.weight", "attention.self.value.bias", "attention.output.dense.weight", "attention.output.dense.bias",
             "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias", "intermediate.dense.weight", "intermediate.dense.bias",
             "output.dense.weight", "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias"]


# jina-bert-v2 (jina-embeddings-v2-*) checkpoint names: no position table (ALiBi), GLU feed-forward `mlp.gated_layers` [2F, H]
# without bias (gate = first F rows), `mlp.wo`, `mlp.layernorm`.  Written from the published module layout; no such checkpoint
# exists offline, so the mapping is only exercised on a file this repo writes with those names (parity unpinned).
_JINA_HEAD = ["embeddings.word_embeddings.weight", "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"]
_JINA_LAYER = ["attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight", "attention.self.key.bias",
               "attention.self.value.weight", "attention.self.value.bias", "attention.output.dense.weight", "attention.output.dense.bias",
               "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias", "mlp.gated_layers.weight", None,  # None: zero bias
               "mlp.wo.weight", "mlp.wo.bias", "mlp.layernorm.weight", "mlp.layernorm.bias"]


def tensors_rows(tensors: dict, name: str) -> int:
    """Number of rows (output features) of a 2-D checkpoint tensor, with or without the "bert." prefix."""
    for key in (name, "bert." + name):
        if key in tensors:
            return int(np.asarray(tensors[key]).shape[0])
    raise KeyError(name)


def load_weight_blob(path: "str | Path", layers: int, cfg: Optional[dict] = None) -> np.ndarray:
    """Flat f32 blob in ABI order from `.npy` (already flat), `.gguf` (llama.cpp's bert / jina-bert-v2 tensor names, F32 / F16 /
    BF16; embeddings/gguf.py) or `.safetensors` (HF BERT names, or jina-bert-v2 names when the
    file holds `mlp.gated_layers`; optional "bert." prefix).  cfg: the encoder configuration the blob is for (checked
    against the checkpoint's architecture: a jina file needs alibi + geglu, a BERT file neither)."""
    path = Path(path)
    if path.suffix == ".npy":
        return np.load(path, allow_pickle=False).astype(np.float32).reshape(-1)
    if path.suffix == ".safetensors":
        from safetensors.numpy import load_file

        tensors = load_file(str(path))

        def get(name: str) -> np.ndarray:
            for key in (name, "bert." + name):
                if key in tensors:
                    return np.asarray(tensors[key], dtype=np.float32).reshape(-1)
            raise KeyError(f"{path}: tensor {name!r} not found")

        jina = any(k.endswith("encoder.layer.0.mlp.gated_layers.weight") for k in tensors)
        if cfg is not None and (bool(cfg.get("alibi")) != jina or bool(cfg.get("geglu")) != jina):
            raise ValueError(f"{path} is a {'jina-bert-v2 (ALiBi + GEGLU)' if jina else 'BERT'} checkpoint but the encoder configuration says "
                             f"alibi={bool(cfg.get('alibi'))}, geglu={bool(cfg.get('geglu'))}")
        head, layer = (_JINA_HEAD, _JINA_LAYER) if jina else (_HF_HEAD, _HF_LAYER)
        parts = [get(n) for n in head]
        for l in range(layers):
            for n in layer:
                if n is None:  # gated_layers has no bias: 2F zeros
                    parts.append(np.zeros(tensors_rows(tensors, f"encoder.layer.{l}.mlp.gated_layers.weight"), np.float32))
                else:
                    parts.append(get(f"encoder.layer.{l}.{n}"))
        return np.concatenate(parts)
    if path.suffix == ".gguf":  # the reference's local model format (settings.embedding_llamacpp_model_path): F32 / F16 / BF16 tensors
        from .gguf import gguf_to_blob

        blob, fcfg, _ = gguf_to_blob(path, cfg)
        if fcfg["layers"] != layers:
            raise ValueError(f"{path} has {fcfg['layers']} layers, the encoder configuration {layers}")
        return blob
    raise ValueError(f"unsupported weight file {path} (use .npy blob, .safetensors or .gguf)")


class MI355XEmbeddings:
    """LangChain-Embeddings-shaped client whose forward runs in libsemcode_hip on the MI355X."""

    def __init__(self, model: Optional[str] = None, *, cfg: Optional[dict] = None, weights: "np.ndarray | str | Path | None" = None,
                 vocab: "dict | str | Path | None" = None, device: Optional[int] = None, max_tokens: Optional[int] = None,
                 normalize: bool = False, batch_size: int = 256, runtime: Any = None, synth_seed: int = 0,
                 allow_synthetic: Optional[bool] = None) -> None:
        from .. import _native  # raises loudly when libsemcode_hip.so is missing: there is no CPU fallback

        settings = _resolve_settings()
        self.model = model or getattr(settings, "embedding_model", None)
        self.max_tokens = int(max_tokens or getattr(settings, "mi355x_max_tokens", 512))
        self.batch_size = int(batch_size)
        # Like the reference's llama.cpp branch, which refuses to start without a model path (providers.py:77-81), a client
        # without weights or without a vocabulary is an error: random-init weights and the hash tokenizer produce vectors that
        # index and search without any error and mean nothing.  Benchmarks and tests opt in explicitly.
        if allow_synthetic is None:
            allow_synthetic = bool(getattr(settings, "mi355x_allow_synthetic", False))
        weights = weights if weights is not None else getattr(settings, "mi355x_weights_path", None)
        vocab = vocab if vocab is not None else getattr(settings, "mi355x_vocab_path", None)
        # the reference's own local-model setting (settings.py:51, providers.py:77-99): a checkout that points it at a GGUF file keeps it
        gguf_path = None
        if weights is None:
            cand = getattr(settings, "embedding_llamacpp_model_path", None)
            if cand and str(cand).lower().endswith(".gguf"):
                weights = cand
        if isinstance(weights, (str, Path)) and str(weights).lower().endswith(".gguf"):
            gguf_path = Path(weights)
        if weights is None and not allow_synthetic:
            raise ValueError("Set SEMCODE_MI355X_WEIGHTS_PATH (.safetensors, .gguf or .npy blob; a .gguf under SEMCODE_EMBEDDING_LLAMACPP_MODEL_PATH is taken too) when using the mi355x embedding provider "
                             "(or SEMCODE_MI355X_ALLOW_SYNTHETIC=1 for random-init benchmark weights).")
        self._native = _native
        self._cfg = dict(_native.BERT_BASE)
        self._vocab_tmp = None
        if gguf_path is not None:  # the file states its own architecture, and carries the WordPiece vocabulary
            from .gguf import gguf_config, gguf_vocab, read_gguf

            meta, tens = read_gguf(gguf_path)
            fcfg = gguf_config(meta)
            fcfg["vocab"] = int(tens["token_embd.weight"][1][0])
            fcfg["type_vocab"] = int(tens["token_types.weight"][1][0])
            if fcfg["alibi"]:
                fcfg["max_pos"] = max(int(fcfg["max_pos"]), int(self._cfg["max_pos"]))
            elif "position_embd.weight" in tens:
                fcfg["max_pos"] = int(tens["position_embd.weight"][1][0])
            self._cfg.update(fcfg)
            if vocab is None:
                toks = gguf_vocab(meta)
                if toks is not None:
                    import tempfile

                    self._vocab_tmp = tempfile.NamedTemporaryFile("w", suffix=".vocab.txt", encoding="utf-8", delete=False)
                    self._vocab_tmp.write("\n".join(toks) + "\n")
                    self._vocab_tmp.close()
                    vocab = self._vocab_tmp.name
        self._cfg.update(cfg or {})
        if vocab is None and not allow_synthetic:  # (a GGUF file brings its own vocabulary: checked after it was read)
            raise ValueError("Set SEMCODE_MI355X_VOCAB_PATH (vocab.txt of the model) when using the mi355x embedding provider "
                             "(or SEMCODE_MI355X_ALLOW_SYNTHETIC=1 for the hash-tokenizer stand-in).")
        self.max_tokens = min(self.max_tokens, self._cfg["max_pos"])
        self._fast_tokenizer: Any = None
        self._owns_runtime = False  # an explicit runtime is the caller's; the default one is shared with the vector store
        self._runtime = runtime or _native.shared_runtime(int(device if device is not None else getattr(settings, "mi355x_device", 0)))
        if isinstance(weights, (str, Path)):
            weights = load_weight_blob(weights, self._cfg["layers"], self._cfg)
        if weights is None:
            log.warning("mi355x embeddings: no weights configured (SEMCODE_MI355X_WEIGHTS_PATH); using random-init weights seed=%d", synth_seed)
        self._encoder = _native.Encoder(self._runtime, self._cfg, weights=weights, normalize=normalize, synth_seed=synth_seed)
        if vocab is None:
            log.warning("mi355x embeddings: no vocab.txt configured (SEMCODE_MI355X_VOCAB_PATH); using the hash tokenizer stand-in")
            self.tokenizer: Any = HashTokenizer(self._cfg["vocab"])
        else:
            self.tokenizer = WordPieceTokenizer(vocab)
            if isinstance(vocab, (str, Path)):  # the C++ tokenizer (ASCII fast path + full Unicode normalisation, multi-threaded)
                self._fast_tokenizer = _native.NativeTokenizer(vocab)
        self.dimension = self._cfg["hidden"]
        # Texts that reached max_tokens lost their tail.  The reference's default chunker emits up to 200 lines / 6 000 characters
        # (tree_sitter_chunker.py:64-65: 1.5-2k word pieces) and only shrinks chunks for llamacpp / lmstudio (ingestion/manager.py:68-79),
        # so with a 512-position BERT most real chunks are cut: counted here and logged (once per power of two), never silent.
        self.truncated_texts = 0
        self.total_texts = 0

    def _note_truncation(self, lens: np.ndarray) -> None:
        cut = int((np.asarray(lens) >= self.max_tokens).sum())
        self.total_texts += int(len(lens))
        if cut:
            before = self.truncated_texts
            self.truncated_texts += cut
            if before == 0 or (self.truncated_texts.bit_length() != before.bit_length()):
                log.warning("mi355x embeddings: %d of %d texts so far reached max_tokens=%d and were truncated (raise SEMCODE_MI355X_MAX_TOKENS up to "
                            "the model's context -- 1024 / 2048 need a position-free (ALiBi) encoder -- or shrink the chunker's max_lines)",
                            self.truncated_texts, self.total_texts, self.max_tokens)

    # ---- LangChain Embeddings surface (lists of Python floats)
    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        return self.embed_documents_array(texts).tolist()

    def embed_query(self, text: str) -> List[float]:
        return self.embed_documents_array([text])[0].tolist()

    # ---- array fast paths
    def embed_documents_array(self, texts: Sequence[str]) -> np.ndarray:
        """texts -> [n, hidden] f32; tokenise on the host, forward on the device in batches."""
        out = np.empty((len(texts), self.dimension), dtype=np.float32)
        for start in range(0, len(texts), self.batch_size):
            ids, lens = self.tokenize(texts[start:start + self.batch_size])
            out[start:start + len(lens)] = self._encoder.embed_ids(ids, lens)
        return out

    def tokenize(self, texts: Sequence[str]) -> "tuple[np.ndarray, np.ndarray]":
        """texts -> (ids [n, S] int32 padded to the smallest sequence bucket that fits, lens [n])."""
        if self._fast_tokenizer is None:
            toks = [self.tokenizer.encode(t, self.max_tokens) for t in texts]
            ids, lens = pack(toks, getattr(self.tokenizer, "pad_id", 0), self.max_tokens)
            self._note_truncation(lens)
            return ids, lens
        from .tokenizer import bucket_for

        smax = bucket_for(10 ** 9, self.max_tokens)
        ids, lens, fallback = self._fast_tokenizer.encode_batch(texts, self.max_tokens, smax)
        for i in np.nonzero(fallback)[0]:  # only input that is not valid UTF-8 (cannot come out of a Python str)
            t = self.tokenizer.encode(texts[i], min(self.max_tokens, smax))
            ids[i, : len(t)] = t
            ids[i, len(t):] = self.tokenizer.pad_id
            lens[i] = len(t)
        S = bucket_for(int(lens.max()) if len(lens) else 1, self.max_tokens)
        self._note_truncation(lens)
        return np.ascontiguousarray(ids[:, :S]), lens

    def embed_ids_array(self, ids: np.ndarray, lens: np.ndarray) -> np.ndarray:
        """Pre-tokenised input: ids [B, S] (S one of 32/64/128/256/512/1024/2048), lens [B] -> [B, hidden] f32."""
        return self._encoder.embed_ids(ids, lens)

    def embed_ids_into(self, store: Any, ids: np.ndarray, lens: np.ndarray, rows: np.ndarray, want_host: bool = False,
                       wait: bool = True) -> "np.ndarray | None":
        """Embed pre-tokenised input and write the vectors into `store`'s device index at `rows` (store.plan_rows),
        device to device.  The store must sit on this client's runtime (the default for both).  wait=False enqueues the
        batch and returns (at most two in flight); call wait() before relying on its completion."""
        index = getattr(store, "_collection", None)
        if index is None:
            raise RuntimeError("Milvus collection is not initialized. Call connect() first.")
        if getattr(store, "_runtime", None) is not self._runtime:
            raise RuntimeError("embed_ids_into: the vector store and the embedding client use different runtimes")
        return self._encoder.embed_ids_into(ids, lens, index, rows, want_host=want_host, wait=wait)

    def wait(self) -> None:
        """Block until every batch enqueued with embed_ids_into(..., wait=False) has finished."""
        self._encoder.wait()

    def close(self) -> None:
        self._encoder.close()
        if self._owns_runtime:
            self._runtime.close()


class EmbeddingProviderFactory:
    """Factory that returns embedding clients based on configuration."""

    @staticmethod
    def create(provider: "str | None" = None, model: "str | None" = None) -> Any:
        settings = _resolve_settings()
        provider_name = (provider or settings.embedding_provider).lower()

        if provider_name in MI355X_PROVIDER_NAMES:
            embed_model = model or settings.embedding_model
            log.info("initializing_mi355x_embeddings model=%s", embed_model)
            return MI355XEmbeddings(model=embed_model)

        if provider_name in {"openai", "lmstudio"} or provider_name.startswith("openai"):
            from langchain_openai import OpenAIEmbeddings  # type: ignore

            embed_model = model or settings.embedding_model
            kwargs: dict[str, Any] = {"model": embed_model, "encoding_format": "float"}
            if settings.embedding_api_base:
                kwargs["base_url"] = settings.embedding_api_base
            if settings.embedding_api_key:
                kwargs["api_key"] = settings.embedding_api_key
            if provider_name != "openai" or not settings.embedding_use_tiktoken:
                kwargs["tiktoken_enabled"] = False
            return OpenAIEmbeddings(**kwargs)

        if provider_name == "jina":
            from langchain_community.embeddings import JinaEmbeddings  # type: ignore

            embed_model = model or settings.embedding_model or "jina-embeddings-v2-base-en"
            jina_kwargs: dict[str, Any] = {"model_name": embed_model}
            if settings.embedding_api_key:
                jina_kwargs["jina_api_key"] = settings.embedding_api_key
            return JinaEmbeddings(**jina_kwargs)

        if provider_name in {"llamacpp", "llama.cpp"}:
            try:
                from langchain_community.embeddings import LlamaCppEmbeddings  # type: ignore
            except ImportError as exc:  # pragma: no cover - optional dependency
                raise RuntimeError(
                    "llama-cpp-python is required for llama.cpp embeddings. "
                    "Install it or select a different embedding provider."
                ) from exc
            model_path = settings.embedding_llamacpp_model_path
            if not model_path:
                raise ValueError("Set SEMCODE_EMBEDDING_LLAMACPP_MODEL_PATH when using the llama.cpp embedding provider.")
            # every argument the reference passes (providers.py:85-99): this branch is untouched by the MI355X backend and a
            # checkout that selects it must construct the very same client
            llama_kwargs: dict[str, Any] = {"model_path": str(model_path), "n_ctx": settings.embedding_llamacpp_n_ctx,
                                            "n_threads": settings.embedding_llamacpp_n_threads, "n_parts": -1, "seed": 0, "f16_kv": True,
                                            "logits_all": False, "vocab_only": False, "use_mlock": False,
                                            "n_batch": settings.embedding_llamacpp_batch_size, "n_gpu_layers": 0, "verbose": False,
                                            "device": "cpu"}
            return LlamaCppEmbeddings(**llama_kwargs)

        raise NotImplementedError(f"Embedding provider not yet supported: {provider_name}")

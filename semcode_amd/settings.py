"""Settings read at the two seams -- same field names and SEMCODE_ env prefix as the reference.

Reference: AppSettings, src/semcode/settings.py:30-82 (pydantic-settings, env prefix SEMCODE_,
extra="allow").  When this package is dropped into a semcode checkout the real
`semcode.settings.settings` object is used instead (see `resolve()`); this stand-alone twin keeps the
seams usable (and testable) without pydantic_settings, which is absent in the build image.

Backend-specific keys ride on the reference's extra="allow" (settings.py:36):
    mi355x_device, mi355x_metric, mi355x_index_type, mi355x_nlist, mi355x_nprobe,
    mi355x_weights_path, mi355x_vocab_path, mi355x_allow_synthetic, mi355x_max_tokens, mi355x_store_path, mi355x_ingest_batch
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Any

_DEFAULTS: dict[str, Any] = {
    # --- names the reference seams read (settings.py:40-76) ---
    "workspace_root": Path("./workspace"),
    "milvus_uri": "http://localhost:19530",
    "milvus_username": None,
    "milvus_password": None,
    "embedding_provider": "openai",
    "embedding_model": "text-embedding-3-large",
    "embedding_dimension": 3072,
    "embedding_api_base": None,
    "embedding_api_key": None,
    "embedding_use_tiktoken": True,
    "embedding_llamacpp_model_path": None,
    "embedding_llamacpp_n_ctx": 2048,
    "embedding_llamacpp_n_threads": 4,
    "embedding_llamacpp_batch_size": 256,
    "embedding_batch_size": 64,
    "rag_max_context_sources": 5,
    "milvus_upsert_batch_size": 128,
    # --- MI355X backend keys (extra="allow") ---
    "mi355x_device": 0,
    "mi355x_metric": "IP",            # milvus_store.py:78-82
    "mi355x_index_type": "IVF_FLAT",  # milvus_store.py:80
    "mi355x_nlist": 128,              # milvus_store.py:81
    "mi355x_nprobe": 16,              # milvus_store.py:144
    "mi355x_weights_path": None,
    "mi355x_vocab_path": None,
    "mi355x_allow_synthetic": False,  # random-init weights / hash tokenizer when the two paths above are unset (benchmarks only)
    "mi355x_max_tokens": 512,
    "mi355x_store_path": None,
    "mi355x_ingest_batch": 256,       # chunks per embed+upsert batch of services.indexer.ingest_chunks
}


def _coerce(raw: str, like: Any) -> Any:
    if isinstance(like, bool):
        return raw.strip().lower() in {"1", "true", "yes", "on"}
    if isinstance(like, int):
        return int(raw)
    if isinstance(like, float):
        return float(raw)
    if isinstance(like, Path):
        return Path(raw)
    return raw


class Settings:
    """Attribute bag with SEMCODE_<NAME> environment overrides; unknown attributes may be set (extra="allow")."""

    def __init__(self, **overrides: Any) -> None:
        for name, default in _DEFAULTS.items():
            raw = os.environ.get("SEMCODE_" + name.upper())
            setattr(self, name, _coerce(raw, default) if raw is not None else default)
        for name, value in overrides.items():
            setattr(self, name, value)


def resolve() -> Any:
    """semcode's own settings object when running inside a semcode checkout, else the twin above."""
    try:  # pragma: no cover - semcode is not importable in the build image
        from semcode.settings import settings as real

        return real
    except Exception:
        return settings


settings = Settings()

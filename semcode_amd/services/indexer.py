"""The embed batch loop and chunk-id rule of IndexerService, restated for the MI355X backend.

Reference: src/semcode/services/indexer.py:135-188 (`_build_payloads`, `_embedding_batch_size`,
`_make_chunk_id`).  The reference's IndexerService itself runs unchanged on top of the two seams
(it only needs EmbeddingProviderFactory.create() and MilvusVectorStore); these functions exist so
that the same behaviour -- batch size, progress protocol, payload mapping, md5 ids -- can be used
(and tested) without the reference's ingestion/chunking stack, which is out of scope and not
importable in the build image (tree_sitter absent).

`chunk` objects are duck-typed like the reference's CodeChunk
(src/semcode/chunking/tree_sitter_chunker.py:48-57): .content, .path (absolute Path inside the
workspace copy), .language, .start_line, .end_line, .symbol.
"""
from __future__ import annotations

import hashlib
from pathlib import Path
from typing import Any, Callable, List, Optional, Sequence

from ..embeddings.payload import EmbeddingPayload
from ..settings import resolve as _resolve_settings


def embedding_batch_size() -> int:
    """indexer.py:175-178: settings.embedding_batch_size (default 64), at least 1."""
    size = getattr(_resolve_settings(), "embedding_batch_size", 64)
    return max(1, size)


def make_chunk_id(repo: str, path: Path, start: int, end: int) -> str:
    """indexer.py:185-188: md5 hex of "repo:path:start:end" -- the collection's primary key."""
    return hashlib.md5(f"{repo}:{path}:{start}:{end}".encode("utf-8")).hexdigest()


def build_payloads(repo_name: str, repo_path: Path, chunks: Sequence[Any], embedding_client: Any,
                   progress: Optional[Callable[[int, int], None]] = None) -> List[EmbeddingPayload]:
    """indexer.py:135-173: embed chunk contents in batches, report progress, zip into payloads.

    progress protocol: (0, total) first; then cumulative (done, total) after every batch; for
    total == 0 only (0, 0) and the embedding client is never called.
    """
    contents = [chunk.content for chunk in chunks]
    total = len(contents)
    if progress:
        progress(0, total)
    vectors: List[List[float]] = []
    if total:
        batch_size = embedding_batch_size()
        for start in range(0, total, batch_size):
            batch = contents[start:start + batch_size]
            vectors.extend(embedding_client.embed_documents(batch))
            if progress:
                progress(len(vectors), total)
    payloads: List[EmbeddingPayload] = []
    for chunk, vector in zip(chunks, vectors):
        chunk_id = make_chunk_id(repo_name, chunk.path, chunk.start_line, chunk.end_line)
        payloads.append(
            EmbeddingPayload(
                id=chunk_id,
                text=chunk.content,
                vector=vector,
                metadata={
                    "repo": repo_name,
                    "path": str(Path(chunk.path).relative_to(repo_path)),
                    "language": chunk.language,
                    "start_line": chunk.start_line,
                    "end_line": chunk.end_line,
                    "symbol": chunk.symbol,
                },
            )
        )
    return payloads

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
import queue
import threading
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


def ingest_chunks(repo_name: str, repo_path: Path, chunks: Sequence[Any], embedding_client: Any, vector_store: Any,
                  embed_progress: Optional[Callable[[int, int], None]] = None,
                  upsert_progress: Optional[Callable[[int, int], None]] = None, batch_size: Optional[int] = None) -> int:
    """`_build_payloads` + `upsert_embeddings` (indexer.py:94-114) as one pipelined pass for the MI355X backend.

    What differs from the reference's two loops, and only in speed: (1) no `list[float]` is ever built -- a batch goes
    from the encoder's device buffer into the index rows (MilvusVectorStore.upsert_encoded); (2) a producer thread
    tokenises batch i+1 (C++ tokenizer, GIL released) while the device embeds batch i; (3) the batch is
    settings.mi355x_ingest_batch chunks (default 256) rather than embedding_batch_size = 64, since 64 x 256 tokens do not
    fill the chip; (4) batches are enqueued without waiting (two in flight), so the primary-key bookkeeping of batch i+1
    overlaps the forward of batch i.  What does not differ: chunk ids (make_chunk_id), payload mapping, replace-by-primary-key, and both
    progress protocols -- (0, total) first, then the cumulative count after every batch; total == 0 reports (0, 0) only
    and calls nothing.  A primary key that occurs more than once keeps its LAST chunk, as sequential upserts would.
    Returns the number of chunks embedded and stored.
    """
    total = len(chunks)
    for cb in (embed_progress, upsert_progress):
        if cb:
            cb(0, total)
    if total == 0:
        return 0
    ids = [make_chunk_id(repo_name, c.path, c.start_line, c.end_line) for c in chunks]
    last = {pk: i for i, pk in enumerate(ids)}
    keep = sorted(last.values())  # later duplicates replace earlier ones; order otherwise preserved
    bs = int(batch_size or getattr(_resolve_settings(), "mi355x_ingest_batch", 256))
    bs = max(1, bs)
    batches = [keep[i:i + bs] for i in range(0, len(keep), bs)]
    q: "queue.Queue" = queue.Queue(maxsize=2)

    def produce() -> None:
        try:
            for idx in batches:
                q.put((idx, embedding_client.tokenize([chunks[i].content for i in idx])))
        except BaseException as exc:  # hand the failure to the consumer instead of dying silently
            q.put(exc)
        else:
            q.put(None)

    worker = threading.Thread(target=produce, name="semcode-tokenize", daemon=True)
    worker.start()
    done = 0
    skipped = total - len(keep)
    try:
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            idx, (tok, lens) = item
            metas = [{"repo": repo_name, "path": str(Path(chunks[i].path).relative_to(repo_path)), "language": chunks[i].language,
                      "start_line": chunks[i].start_line, "end_line": chunks[i].end_line, "symbol": chunks[i].symbol} for i in idx]
            vector_store.upsert_encoded([ids[i] for i in idx], tok, lens, [chunks[i].content for i in idx], metas, embedding_client, wait=False)
            done += len(idx)
            if done == len(keep):
                embedding_client.wait()  # the last report means "stored"
            reported = done + (skipped if done == len(keep) else 0)  # superseded duplicates count as done at the end
            for cb in (embed_progress, upsert_progress):
                if cb:
                    cb(reported, total)
    finally:
        try:
            embedding_client.wait()  # never leave enqueued batches behind, whatever happened above
        except Exception:  # pragma: no cover - the original error is the one to surface
            pass
        while worker.is_alive():  # unblock the producer if we stopped early
            try:
                q.get_nowait()
            except queue.Empty:
                worker.join(0.05)
    return len(keep)

"""End-to-end ingest throughput, strings in -> searchable rows: tokenizer + PCIe + encoder + index append.

Two ways through the same seams on the same synthetic "source code" chunks (about 256 WordPiece tokens each, BERT-base
shaped encoder with random-init weights, synthetic 30522-word vocabulary -- no real vocabulary or weights exist offline):
  reference loops : build_payloads (embedding_batch_size = 64, list[float] vectors) + upsert_embeddings (batch 128)
                    = what IndexerService does through the two seams (indexer.py:94-114)
  fused           : services.ingest_chunks (batch 256, tokenizer thread, device-to-device rows)
These are the PCIe- and host-inclusive rates DESIGN.md quotes beside bench.py's HBM-resident `value`.
"""
import sys
import tempfile
import time
from dataclasses import dataclass
from pathlib import Path

sys.path.insert(0, ".")
import numpy as np

from semcode_amd.embeddings.providers import MI355XEmbeddings
from semcode_amd.services import build_payloads, ingest_chunks
from semcode_amd.storage import MilvusVectorStore


@dataclass
class Chunk:
    content: str
    path: Path
    language: str
    start_line: int
    end_line: int
    symbol: "str | None" = None


def make_vocab(path: Path, n: int = 30522) -> list:
    rng = np.random.default_rng(0)
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list("(){}[]:;,.=+-*/<>!&|%#@\"'_")
    letters = np.array(list("abcdefghijklmnopqrstuvwxyz"))
    seen = set(words)
    while len(words) < n:
        w = "".join(rng.choice(letters, size=int(rng.integers(2, 9))))
        w = w if rng.random() < 0.7 else "##" + w
        if w not in seen:
            seen.add(w)
            words.append(w)
    path.write_text("\n".join(words) + "\n", encoding="utf-8")
    return [w for w in words if w.isalpha()]


def main() -> None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    tmp = Path(tempfile.mkdtemp())
    plain = make_vocab(tmp / "vocab.txt")
    rng = np.random.default_rng(1)
    root = Path("/w/demo")
    chunks = []
    for i in range(n):
        toks = rng.choice(len(plain), size=215)
        body = " ".join(plain[t] if j % 9 else plain[t] + "(x):" for j, t in enumerate(toks))  # ~254 word pieces with the punctuation
        chunks.append(Chunk(body, root / "src" / f"m{i}.py", "python", 1, 40))
    emb = MI355XEmbeddings(vocab=tmp / "vocab.txt", max_tokens=256, allow_synthetic=True)  # random-init benchmark weights
    ids, lens = emb.tokenize([c.content for c in chunks[:256]])
    print(f"chunks {n}, tokens per chunk: mean {lens.mean():.0f} max {lens.max()} (bucket {ids.shape[1]})", flush=True)
    t0 = time.perf_counter()
    for s in range(0, n, 256):
        emb.tokenize([c.content for c in chunks[s:s + 256]])
    t_tok = time.perf_counter() - t0
    print(f"tokenizer alone      : {n / t_tok:9.0f} chunks/s", flush=True)
    ingest_chunks("demo", root, chunks[:512], emb, _store())  # warm-up (workspace allocation, first-touch)

    slow = _store()
    t0 = time.perf_counter()
    payloads = build_payloads("demo", root, chunks, emb)
    t_embed = time.perf_counter() - t0
    slow.upsert_embeddings(payloads)
    t_slow = time.perf_counter() - t0
    print(f"reference loops      : {n / t_slow:9.0f} chunks/s  (embed {t_embed:.2f} s + upsert {t_slow - t_embed:.2f} s)", flush=True)

    fast = _store()
    t0 = time.perf_counter()
    ingest_chunks("demo", root, chunks, emb, fast)
    t_fast = time.perf_counter() - t0
    print(f"fused ingest_chunks  : {n / t_fast:9.0f} chunks/s  ({t_fast:.2f} s)", flush=True)
    q = emb.embed_documents_array([c.content for c in chunks[:64]])
    (_, rf), (_, rs) = fast.search_batch(q, top_k=3), slow.search_batch(q, top_k=3)
    print("same top-3 rows through both paths:", bool(np.array_equal(rf, rs)), " self-hit:", rf[:, 0].tolist() == list(range(64)))
    for s_ in (slow, fast):
        s_.close()
    emb.close()


def _store() -> MilvusVectorStore:
    s = MilvusVectorStore(dim=768, metric="COSINE", index_type="FLAT")
    s.connect()
    return s


if __name__ == "__main__":
    main()

"""Host-side tokenisation feeding sc_encoder_embed_ids (first cut in Python).

The reference hands raw strings to its provider (src/semcode/services/indexer.py:141,150) and the
provider's library tokenises internally; here that step is explicit and stays on the host side of the
C ABI.  `WordPieceTokenizer` is BERT's scheme (basic split + greedy longest-match-first word pieces)
over any `vocab.txt`; it is checked against transformers' BertTokenizer on a synthetic vocabulary in
tests/test_tokenizer.py.  No vocabulary file exists offline, so when none is configured the
`HashTokenizer` stand-in maps words to stable pseudo-ids (fine for throughput work and plumbing,
meaningless for retrieval quality -- a warning says so).
"""
from __future__ import annotations

import unicodedata
from pathlib import Path
from typing import Iterable, List, Sequence

import numpy as np

SEQ_BUCKETS = (32, 64, 128, 256, 512, 1024, 2048)


def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


def basic_split(text: str, lowercase: bool = True) -> List[str]:
    """BERT BasicTokenizer: clean, space CJK, whitespace split, lower + strip accents, split punctuation."""
    out = []
    for ch in text:
        cp = ord(ch)
        if cp == 0 or cp == 0xFFFD or (unicodedata.category(ch).startswith("C") and ch not in "\t\n\r"):  # Cc, Cf, Cn, Co, Cs
            continue
        if _is_cjk(cp):
            out.append(f" {ch} ")
        elif ch in "\t\n\r" or unicodedata.category(ch) == "Zs":
            out.append(" ")
        else:
            out.append(ch)
    words: List[str] = []
    for tok in "".join(out).split():
        if lowercase:
            # code point by code point, as the Rust `tokenizers` normaliser and the C++ tokenizer do: str.lower() on the whole word
            # applies the Greek final-sigma rule ("ΟΔΟΣ" -> "οδος"), they do not ("οδοσ")
            tok = "".join(c.lower() for c in tok)
            tok = "".join(c for c in unicodedata.normalize("NFD", tok) if unicodedata.category(c) != "Mn")
        cur = ""
        for ch in tok:
            if _is_punct(ch):
                if cur:
                    words.append(cur)
                    cur = ""
                words.append(ch)
            else:
                cur += ch
        if cur:
            words.append(cur)
    return words


class WordPieceTokenizer:
    def __init__(self, vocab: "dict[str, int] | str | Path", lowercase: bool = True, unk: str = "[UNK]", cls: str = "[CLS]",
                 sep: str = "[SEP]", pad: str = "[PAD]", max_chars_per_word: int = 100) -> None:
        if not isinstance(vocab, dict):
            lines = Path(vocab).read_text(encoding="utf-8").split("\n")
            vocab = {tok.rstrip("\r"): i for i, tok in enumerate(lines) if tok.rstrip("\r") != "" or i < len(lines) - 1}
        self.vocab = vocab
        self.lowercase = lowercase
        self.unk_id, self.cls_id, self.sep_id = vocab[unk], vocab[cls], vocab[sep]
        self.pad_id = vocab.get(pad, 0)
        self.max_chars_per_word = max_chars_per_word

    @property
    def vocab_size(self) -> int:
        return max(self.vocab.values()) + 1

    def _wordpiece(self, word: str) -> List[int]:
        if len(word) > self.max_chars_per_word:
            return [self.unk_id]
        ids, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                piece = word[start:end] if start == 0 else "##" + word[start:end]
                if piece in self.vocab:
                    cur = self.vocab[piece]
                    break
                end -= 1
            if cur is None:
                return [self.unk_id]
            ids.append(cur)
            start = end
        return ids

    def encode(self, text: str, max_tokens: int = 512) -> List[int]:
        """[CLS] pieces... [SEP], truncated to max_tokens."""
        ids = [self.cls_id]
        for w in basic_split(text, self.lowercase):
            ids.extend(self._wordpiece(w))
            if len(ids) >= max_tokens - 1:
                break
        return ids[: max_tokens - 1] + [self.sep_id]


class HashTokenizer:
    """Stand-in when no vocab.txt is configured: stable pseudo-ids from an FNV-1a hash of each word."""

    def __init__(self, vocab_size: int = 30522, lowercase: bool = True) -> None:
        self.vocab_size = vocab_size
        self.lowercase = lowercase
        self.pad_id, self.unk_id, self.cls_id, self.sep_id = 0, 100, 101, 102  # BERT's conventional ids

    def encode(self, text: str, max_tokens: int = 512) -> List[int]:
        ids = [self.cls_id]
        for w in basic_split(text, self.lowercase):
            h = 0xCBF29CE484222325
            for b in w.encode("utf-8"):
                h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
            ids.append(1000 + h % (self.vocab_size - 1000))
            if len(ids) >= max_tokens - 1:
                break
        return ids[: max_tokens - 1] + [self.sep_id]


def bucket_for(n: int, max_tokens: int = 512) -> int:
    for b in SEQ_BUCKETS:
        if n <= b and b <= max(max_tokens, SEQ_BUCKETS[0]):
            return b
    return max(b for b in SEQ_BUCKETS if b <= max(max_tokens, SEQ_BUCKETS[0]))


def pack(token_lists: Sequence[Sequence[int]], pad_id: int = 0, max_tokens: int = 512) -> "tuple[np.ndarray, np.ndarray]":
    """Ragged id lists -> (ids [B, S] int32 padded to the smallest bucket that fits, lens [B] int32)."""
    limit = bucket_for(10 ** 9, max_tokens)
    lens = np.asarray([min(len(t), limit) for t in token_lists], dtype=np.int32)
    S = bucket_for(int(lens.max()) if len(lens) else 1, max_tokens)
    ids = np.full((len(token_lists), S), pad_id, dtype=np.int32)
    for i, t in enumerate(token_lists):
        ids[i, : lens[i]] = t[: lens[i]]
    return ids, lens

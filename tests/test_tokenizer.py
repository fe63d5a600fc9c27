"""CPU: the host tokenizer against transformers' BertTokenizer on a synthetic vocabulary, and the factory's dispatch."""
import json

import numpy as np
import pytest

from semcode_amd.embeddings import EmbeddingProviderFactory
from semcode_amd.embeddings.tokenizer import HashTokenizer, WordPieceTokenizer, basic_split, bucket_for, pack

WORDS = ["def", "return", "class", "self", "print", "hello", "world", "name", "str", "int", "for", "in", "range", "if", "else",
         "greet", "the", "quick", "brown", "fox", "jump", "##s", "##ed", "##ing", "##er", "un", "##believ", "##able", "cafe", "##x",
         "f", "x", "y", "a", "b", "c", "0", "1", "2", "10", "##0", "_", "(", ")", ":", ",", ".", "=", "+", "-", ">", "\"", "'", "{", "}",
         "[", "]", "#", "中", "文", "naive", "resume", "##_", "__", "init", "##__"]


@pytest.fixture(scope="module")
def vocab_file(tmp_path_factory):
    toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS
    p = tmp_path_factory.mktemp("vocab") / "vocab.txt"
    p.write_text("\n".join(toks) + "\n", encoding="utf-8")
    return p


TEXTS = [
    'def greet(name: str) -> str:\n    return f"Hello {name}"\n',
    "The quick brown fox jumps, jumped; jumping... unbelievable!",
    "class A(B):\n\tdef __init__(self):\n\t\tself.x = 10 + y[0]",
    "café naïve résumé 中文 mixed space​ zero-width",
    "",
    "x" * 150 + " ok",
    "for i in range(10): print(i) # comment",
]


def test_wordpiece_matches_transformers(vocab_file):
    transformers = pytest.importorskip("transformers")
    hf = transformers.BertTokenizer(str(vocab_file), do_lower_case=True)
    mine = WordPieceTokenizer(vocab_file)
    for t in TEXTS:
        want = hf.encode(t, add_special_tokens=True, truncation=True, max_length=64)
        assert mine.encode(t, 64) == want, t
    long = " ".join(["hello world"] * 300)
    assert mine.encode(long, 32) == hf.encode(long, add_special_tokens=True, truncation=True, max_length=32)


def test_basic_split_and_hash_tokenizer_are_deterministic():
    assert basic_split("Hello, World! x=1") == ["hello", ",", "world", "!", "x", "=", "1"]
    h = HashTokenizer(30522)
    a, b = h.encode("def greet(name): return name"), h.encode("def greet(name): return name")
    assert a == b and a[0] == 101 and a[-1] == 102 and all(0 <= i < 30522 for i in a)
    assert h.encode("") == [101, 102]


def test_native_tokenizer_reproduces_committed_berttokenizer_ids(golden, tmp_path):
    """tests/golden/tokenizer_unicode.json (oracle/gen_tokenizer_fixtures.py): 20 texts over Latin accents, CJK, Cyrillic, Greek,
    Hangul, Arabic / Hebrew / Indic / Thai, emoji, control and format characters, full-width forms, > 100-character words --
    ids of the in-container transformers BertTokenizer, uncased and cased, with and without truncation."""
    import json

    from semcode_amd import _native

    g = json.loads((golden / "tokenizer_unicode.json").read_text())
    vp = tmp_path / "vocab.txt"
    vp.write_text("\n".join(g["vocab"]) + "\n", encoding="utf-8")
    for name, want in g["ids"].items():
        nat = _native.NativeTokenizer(vp, lowercase=name.startswith("uncased"))
        max_tokens = int(name.split("_")[1])
        ids, lens, fb = nat.encode_batch(g["texts"], max_tokens, 512, threads=2)
        assert not fb.any()
        for i, w in enumerate(want):
            assert ids[i, : lens[i]].tolist() == w, (name, i, g["texts"][i][:40])
        nat.close()


def test_python_tokenizer_reproduces_the_same_committed_ids(golden):
    """The Python WordPieceTokenizer (taken when the vocabulary is handed over as a dict) must give the ids the C++ tokenizer and
    BertTokenizer give: it used to lower-case with str.lower(), whose Greek final-sigma rule ("ΟΔΟΣ" -> "οδος") neither of them
    applies, and to keep private-use code points they drop -- the same text got different ids depending on how the provider was built."""
    g = json.loads((golden / "tokenizer_unicode.json").read_text())
    vocab = {t: i for i, t in enumerate(g["vocab"])}
    for key, lower in (("uncased_512", True), ("cased_512", False), ("uncased_16", True), ("cased_16", False)):
        tk = WordPieceTokenizer(vocab, lowercase=lower)
        limit = int(key.split("_")[1])
        for text, want in zip(g["texts"], g["ids"][key]):
            assert tk.encode(text, limit) == want, (key, text)
    tk = WordPieceTokenizer({"[PAD]": 0, "[UNK]": 1, "[CLS]": 2, "[SEP]": 3, "οδοσ": 4, "οδος": 5})
    assert tk.encode("ΟΔΟΣ") == [2, 4, 3]  # code point by code point: no final sigma


def test_native_tokenizer_equals_berttokenizer_on_every_code_point(tmp_path):
    """Live check against the in-container tokenizer: every code point (surrogates excepted), embedded between two letters, must
    normalise and split exactly as BertTokenizer does -- the property the generated tables were built to have, checked end to end
    through sc_tokenizer_encode with a vocabulary that makes each outcome visible."""
    transformers = pytest.importorskip("transformers")
    from semcode_amd import _native

    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "a", "b", "##b", "ab", "##a"]
    vp = tmp_path / "vocab.txt"
    vp.write_text("\n".join(vocab) + "\n", encoding="utf-8")
    for lower in (True, False):
        hf = transformers.BertTokenizer(str(vp), do_lower_case=lower)
        nat = _native.NativeTokenizer(vp, lowercase=lower)
        cps = [cp for cp in range(0x80, 0x110000, 1 if lower else 7) if not 0xD800 <= cp <= 0xDFFF]
        for start in range(0, len(cps), 20000):
            texts = ["a" + chr(cp) + "b" for cp in cps[start:start + 20000]]
            want = hf(texts, truncation=True, max_length=16)["input_ids"]
            ids, lens, fb = nat.encode_batch(texts, 16, 16)
            assert not fb.any()
            bad = [hex(cps[start + i]) for i, w in enumerate(want) if ids[i, : lens[i]].tolist() != w]
            assert not bad, (lower, bad[:20])
        nat.close()
    # malformed UTF-8 cannot come out of a Python str; through the C ABI it is flagged, not guessed at
    import ctypes as C

    nat = _native.NativeTokenizer(vp)
    blob, offs = b"ok \xff\xfe bad", np.array([0, 10], np.int64)
    ids, lens, fb = np.zeros((1, 16), np.int32), np.zeros(1, np.int32), np.zeros(1, np.uint8)
    st = _native.lib().sc_tokenizer_encode(nat._h, blob, offs.ctypes.data_as(C.c_void_p), 1, 16, 16, ids.ctypes.data_as(C.c_void_p),
                                           lens.ctypes.data_as(C.c_void_p), fb.ctypes.data_as(C.c_void_p), 1)
    assert st == 0 and fb[0] == 1 and lens[0] == 0
    nat.close()


def test_pack_buckets():
    assert [bucket_for(n) for n in (1, 32, 33, 64, 200, 256, 257, 512, 9999)] == [32, 32, 64, 64, 256, 256, 512, 512, 512]
    assert bucket_for(300, max_tokens=256) == 256
    ids, lens = pack([[5, 6, 7], list(range(40))], pad_id=0)
    assert ids.shape == (2, 64) and lens.tolist() == [3, 40] and ids[0, 3:].sum() == 0 and ids.dtype == np.int32
    ids, lens = pack([list(range(700))], max_tokens=512)
    assert ids.shape == (1, 512) and lens.tolist() == [512]


def test_factory_dispatch_and_error_types(monkeypatch):
    # reference providers.py:102-104
    with pytest.raises(NotImplementedError, match="Embedding provider not yet supported: nope"):
        EmbeddingProviderFactory.create(provider="nope")
    # reference providers.py:79-82: llama.cpp path unset -> ValueError (when langchain_community is importable),
    # or the RuntimeError of providers.py:73-76 when it is not
    with pytest.raises((ValueError, RuntimeError)):
        EmbeddingProviderFactory.create(provider="llamacpp")
    # the MI355X provider without weights / vocabulary configured: ValueError, as for the unset llama.cpp path
    monkeypatch.delenv("SEMCODE_MI355X_WEIGHTS_PATH", raising=False)
    monkeypatch.delenv("SEMCODE_MI355X_ALLOW_SYNTHETIC", raising=False)
    with pytest.raises(ValueError, match="SEMCODE_MI355X_WEIGHTS_PATH"):
        EmbeddingProviderFactory.create(provider="mi355x")
    # and it needs the device: without one the error surfaces (no silent CPU fallback)
    import torch

    if not torch.cuda.is_available():
        from semcode_amd import _native

        from semcode_amd.settings import settings

        monkeypatch.setattr(settings, "mi355x_allow_synthetic", True)  # the settings object is read once at import, like the reference's
        with pytest.raises(_native.ScError):
            EmbeddingProviderFactory.create(provider="mi355x")


def test_native_tokenizer_matches_python_on_ascii_and_handles_unicode(vocab_file):
    import time

    from semcode_amd import _native

    nat = _native.NativeTokenizer(vocab_file)
    py = WordPieceTokenizer(vocab_file)
    assert (nat.vocab_size, nat.pad_id, nat.unk_id, nat.cls_id, nat.sep_id) == (py.vocab_size, py.pad_id, py.unk_id, py.cls_id, py.sep_id)
    ascii_texts = [t for t in TEXTS if t.isascii()] + ["a\x00b\x01c\x7f d", "\t\n  ", "x=y+1;print(x)  #done", " ".join(["hello world"] * 300)]
    for max_tokens, S in ((64, 64), (32, 64), (512, 512)):
        ids, lens, fb = nat.encode_batch(ascii_texts, max_tokens, S, threads=3)
        assert not fb.any()
        for i, t in enumerate(ascii_texts):
            want = py.encode(t, min(max_tokens, S))
            assert ids[i, : lens[i]].tolist() == want, (t[:40], max_tokens)
            assert (ids[i, lens[i]:] == nat.pad_id).all()
    # non-ASCII texts are tokenised natively too (NFD accent stripping, CJK spacing, Unicode punctuation: unicode_tables.h)
    mixed = ["plain ascii", "café", "中文", "naive", TEXTS[3]]
    ids, lens, fb = nat.encode_batch(mixed, 64, 64)
    assert not fb.any()
    for i, t in enumerate(mixed):
        assert ids[i, : lens[i]].tolist() == py.encode(t, 64), t
    assert ids[1, : lens[1]].tolist() == nat.encode_batch(["cafe"], 64, 64)[0][0, : lens[1]].tolist()  # accents are stripped
    # throughput sanity: the C++ path is much faster than the per-character Python path
    big = ["def f%d(x):\\n    return x + %d  # helper number %d\\n" % (i, i, i) * 20 for i in range(400)]
    t0 = time.perf_counter(); nat.encode_batch(big, 256, 256); t_nat = time.perf_counter() - t0
    t0 = time.perf_counter(); [py.encode(t, 256) for t in big[:40]]; t_py = (time.perf_counter() - t0) * 10
    assert t_nat < t_py / 5, (t_nat, t_py)
    nat.close()

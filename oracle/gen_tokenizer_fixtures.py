"""Generate tests/golden/tokenizer_unicode.json: texts with non-ASCII content, a synthetic vocabulary, and the ids the in-container
`transformers` BertTokenizer produces for them (uncased and cased).  The C++ tokenizer (sc_tokenizer_encode) must reproduce them.

    python -m oracle.gen_tokenizer_fixtures

The reference hands raw strings to its provider's library (src/semcode/services/indexer.py:141,150), whose tokenizer is not
available offline; BertTokenizer is the scheme the BERT-family encoder this repo loads is trained with."""
import json
import random
import tempfile
from pathlib import Path

from transformers import BertTokenizer

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tests" / "golden" / "tokenizer_unicode.json"

TEXTS = [
    "def café(naïve): return 'Ünïcödé' # déjà vu",
    "// 日本語のコメント: 変数を初期化する\nint x = 0; /* 中文注释 */",
    "print(\"Привет, мир!\")  # комментарий на русском языке",
    "ΟΔΥΣΣΕΥΣ είναι ΑΣ και Σ; ΐ ΰ",
    "İstanbul'da ǅungla ẞtraße ß ﬁne Å㎏",
    "한글 주석: 값을 반환한다",
    "emoji 🚀 in a string 👩‍💻 and ½ ² ³ № ™",
    "zero​width soft­hyphen line sep para sep nb sp ideographic　space",
    "“curly quotes” — em-dash … ellipsis ‹guillemets› «x» ¿qué? ¡sí! § ¶ † •",
    "arabic: مرحبا بالعالم  hebrew: שלום עולם  hindi: नमस्ते दुनिया  thai: สวัสดี",
    "math: ∀x∈ℝ, x² ≥ 0 ⇒ √x ≤ ∞ ≠ ≈ ± × ÷",
    "control\x01chars\x7f\x00 tab\there\r\nnewline � replacement  private",
    "ｆｕｌｌｗｉｄｔｈ ＡＢＣ １２３ ﹙small﹚ ︵vertical︶",
    "combining: é ǟ ộ कि กิ",
    "A" * 101 + " é" * 3 + " " + "ü" * 100 + " " + "ü" * 101,
    "snake_case camelCase kebab-case $var @decorator #include <vector> a->b a::b x<<=1",
    "",
    "   \t\n  ",
    "ǆ Ǆ ǅ ﬃ ŉ ǰ ΐ ẖ ẗ ẘ",
    "CJK ext: 𠀀𪜀 compat: 豈更 radicals: ⺀⼀ kana: ｱｲｳ ゙ ゚",
]


def build_vocab(texts, rng):
    tok = BertTokenizer(_vocab_file(["[PAD]", "[UNK]", "[CLS]", "[SEP]"]), do_lower_case=True)
    bt = tok.backend_tokenizer
    chars, words = set(), set()
    for lower in (True, False):
        t = BertTokenizer(_vocab_file(["[PAD]", "[UNK]", "[CLS]", "[SEP]"]), do_lower_case=lower).backend_tokenizer
        for text in texts:
            for w, _ in t.pre_tokenizer.pre_tokenize_str(t.normalizer.normalize_str(text)):
                words.add(w)
                chars.update(w)
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    chars = sorted(chars)
    rng.shuffle(chars)
    keep = chars[: int(len(chars) * 0.9)]  # a tenth of the characters stay out of the vocabulary -> [UNK] words
    vocab += sorted(keep) + sorted("##" + c for c in keep)
    for w in sorted(words):
        if 2 <= len(w) <= 12 and rng.random() < 0.5:
            cut = rng.randint(1, len(w) - 1)
            vocab += [w[:cut], "##" + w[cut:]]
        if len(w) <= 8 and rng.random() < 0.3:
            vocab.append(w)
    seen, out = set(), []
    for v in vocab:
        if v not in seen:
            seen.add(v)
            out.append(v)
    return out


def _vocab_file(vocab):
    d = tempfile.mkdtemp()
    p = Path(d) / "vocab.txt"
    p.write_text("\n".join(vocab) + "\n", encoding="utf-8")
    return str(p)


def main():
    rng = random.Random(7)
    vocab = build_vocab(TEXTS, rng)
    vf = _vocab_file(vocab)
    cases = {}
    for lower in (True, False):
        tok = BertTokenizer(vf, do_lower_case=lower)
        for max_tokens in (512, 16):
            ids = [tok(t, truncation=True, max_length=max_tokens)["input_ids"] for t in TEXTS]
            cases[f"{'uncased' if lower else 'cased'}_{max_tokens}"] = ids
    import tokenizers
    import transformers

    OUT.write_text(json.dumps({"generator": "oracle/gen_tokenizer_fixtures.py", "transformers": transformers.__version__,
                               "tokenizers": tokenizers.__version__, "vocab": vocab, "texts": TEXTS, "ids": cases}, ensure_ascii=True, indent=0))
    print("wrote", OUT, OUT.stat().st_size, "bytes;", len(vocab), "vocab entries")


if __name__ == "__main__":
    main()

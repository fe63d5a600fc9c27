"""Generate semcode_amd/csrc/unicode_tables.h: the per-code-point behaviour of BERT's text normalisation, as the C++ WordPiece
tokenizer (sc_tokenizer.cpp) needs it for non-ASCII text.

    python scripts/gen_unicode_tables.py            # writes semcode_amd/csrc/unicode_tables.h

Source of truth: the in-container `transformers` BertTokenizer (tokenizers' BertNormalizer(clean_text, handle_chinese_chars,
strip_accents = lowercase, lowercase) + BertPreTokenizer), probed one code point at a time -- that is the tokenizer a BERT-family
checkpoint is used with, and its Unicode tables are a mix of versions that no single `unicodedata` reproduces (its category
tables predate Python's; its NFD tables are newer).  The reference hands raw strings to its provider's library
(src/semcode/services/indexer.py:141,150); this table is what lets the C++ tokenizer accept the non-ASCII ones.

Two tables:
  NORM  cp -> what the normaliser emits for it: nothing (control / format / private use / U+FFFD / stripped accent), a space,
        the character wrapped in spaces (CJK ideographs), or a replacement sequence (NFD + accent stripping + lower-casing;
        identity is not stored).  Two versions of the replacement: lower-casing on (uncased models) and off.
  PRE   cp -> punctuation (split into a token of its own) | whitespace | other, as the pre-tokenizer sees a normalised character.
Hangul syllables (U+AC00..U+D7A3) decompose algorithmically and are not stored."""
import os
import sys
import tempfile
from pathlib import Path

from transformers import BertTokenizer

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "semcode_amd" / "csrc" / "unicode_tables.h"

d = tempfile.mkdtemp()
vp = os.path.join(d, "vocab.txt")
open(vp, "w").write("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "a", "b"]) + "\n")


def probe(lowercase):
    tok = BertTokenizer(vp, do_lower_case=lowercase)
    bt = tok.backend_tokenizer
    return bt.normalizer, bt.pre_tokenizer


REMOVED, SPACE, CJK, MAPPED = 1, 2, 3, 4


def norm_table(norm):
    cls, repl = {}, {}
    for cp in range(0x80, 0x110000):
        if 0xD800 <= cp <= 0xDFFF or 0xAC00 <= cp <= 0xD7A3:
            continue
        ch = chr(cp)
        out = norm.normalize_str("a" + ch + "b")
        assert out[0] == "a" and out[-1] == "b", (hex(cp), out)
        mid = out[1:-1]
        if mid == "":
            cls[cp] = REMOVED
        elif mid == " ":
            cls[cp] = SPACE
        elif len(mid) >= 3 and mid[0] == " " and mid[-1] == " " and mid[1:-1] == ch:
            cls[cp] = CJK
        elif mid != ch:  # includes compatibility ideographs: wrapped in spaces AND decomposed -> the spaces are part of the replacement
            cls[cp] = MAPPED
            repl[cp] = [ord(c) for c in mid]
    return cls, repl


def pre_table(pre):
    punct, space = [], []
    for cp in range(0x80, 0x110000):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        pieces = [t for t, _ in pre.pre_tokenize_str("a" + chr(cp) + "b")]
        if len(pieces) == 3:
            punct.append(cp)
        elif len(pieces) == 2:
            space.append(cp)
        else:
            assert len(pieces) == 1, (hex(cp), pieces)
    return punct, space


def ranges(cps):
    out, start, prev = [], None, None
    for cp in cps:
        if start is None:
            start = prev = cp
        elif cp == prev + 1:
            prev = cp
        else:
            out.append((start, prev))
            start = prev = cp
    if start is not None:
        out.append((start, prev))
    return out


def emit_ranges(f, name, cps):
    r = ranges(sorted(cps))
    f.write(f"static const uint32_t {name}[][2] = {{\n")
    for i in range(0, len(r), 6):
        f.write("    " + " ".join(f"{{0x{a:X}, 0x{b:X}}}," for a, b in r[i:i + 6]) + "\n")
    f.write("};\n")
    f.write(f"static const int {name}_N = {len(r)};\n\n")


def emit_map(f, name, repl):
    keys = sorted(repl)
    pool, index = [], []
    for cp in keys:
        index.append((cp, len(pool), len(repl[cp])))
        pool.extend(repl[cp])
    f.write(f"static const uint32_t {name}_KEY[] = {{\n")
    for i in range(0, len(index), 10):
        f.write("    " + " ".join(f"0x{cp:X}," for cp, _, _ in index[i:i + 10]) + "\n")
    f.write("};\n")
    f.write(f"static const uint32_t {name}_POS[] = {{  /* (offset << 3) | length into {name}_POOL */\n")
    for i in range(0, len(index), 10):
        f.write("    " + " ".join(f"0x{(off << 3) | n:X}," for _, off, n in index[i:i + 10]) + "\n")
    f.write("};\n")
    f.write(f"static const uint32_t {name}_POOL[] = {{\n")
    for i in range(0, len(pool), 12):
        f.write("    " + " ".join(f"0x{c:X}," for c in pool[i:i + 12]) + "\n")
    f.write("};\n")
    f.write(f"static const int {name}_N = {len(index)};\n\n")
    assert max(n for _, _, n in index) < 8


def main():
    import tokenizers
    import transformers

    norm_lc, pre = probe(True)
    norm_cs, _ = probe(False)
    cls_lc, repl_lc = norm_table(norm_lc)
    cls_cs, repl_cs = norm_table(norm_cs)
    punct, space = pre_table(pre)
    with open(OUT, "w") as f:
        f.write("// unicode_tables.h -- GENERATED by scripts/gen_unicode_tables.py; do not edit.\n")
        f.write(f"// Probed from transformers {transformers.__version__} / tokenizers {tokenizers.__version__} BertNormalizer + BertPreTokenizer, one code point at a time\n")
        f.write("// (code points >= 0x80; ASCII is handled in code; Hangul syllables decompose algorithmically).\n")
        f.write("#pragma once\n#include <stdint.h>\n\n")
        f.write("// ---- normaliser, lowercase = true (uncased models: NFD, accents stripped, lower-cased)\n")
        emit_ranges(f, "UT_LC_REMOVED", [cp for cp, c in cls_lc.items() if c == REMOVED])
        emit_map(f, "UT_LC_MAP", repl_lc)
        f.write("// ---- normaliser, lowercase = false (cased models: clean-up and CJK spacing only)\n")
        emit_ranges(f, "UT_CS_REMOVED", [cp for cp, c in cls_cs.items() if c == REMOVED])
        if repl_cs:
            emit_map(f, "UT_CS_MAP", repl_cs)
        else:
            f.write("static const int UT_CS_MAP_N = 0;\nstatic const uint32_t UT_CS_MAP_KEY[1] = {0}, UT_CS_MAP_POS[1] = {0}, UT_CS_MAP_POOL[1] = {0};\n\n")
        f.write("// ---- both: characters replaced by a space, ideographs wrapped in spaces\n")
        assert {cp for cp, c in cls_lc.items() if c == SPACE} == {cp for cp, c in cls_cs.items() if c == SPACE}
        emit_ranges(f, "UT_SPACE", [cp for cp, c in cls_lc.items() if c == SPACE])
        # wrapped in spaces as they are (compatibility ideographs that the uncased model also decomposes sit in UT_LC_MAP, spaces included)
        emit_ranges(f, "UT_LC_CJK", [cp for cp, c in cls_lc.items() if c == CJK])
        emit_ranges(f, "UT_CS_CJK", [cp for cp, c in cls_cs.items() if c == CJK])
        f.write("// ---- pre-tokenizer: punctuation (a token of its own), whitespace (after normalisation only U+0020 occurs; kept for completeness)\n")
        emit_ranges(f, "UT_PUNCT", punct)
        emit_ranges(f, "UT_PRE_SPACE", space)
    print("wrote", OUT, OUT.stat().st_size, "bytes;", len(repl_lc), "lower-case mappings,", len(repl_cs), "cased mappings,", len(punct), "punctuation code points")


if __name__ == "__main__":
    sys.exit(main())

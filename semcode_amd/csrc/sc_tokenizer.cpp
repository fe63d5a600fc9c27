// sc_tokenizer.cpp -- host-side WordPiece tokenizer (BERT scheme) behind the C ABI, multi-threaded.
//
// Replaces (reference): the tokenisation hidden inside the embedding provider's library
// (llama.cpp's tokenizer behind LlamaCppEmbeddings.embed_documents, src/semcode/embeddings/providers.py:69-100;
// the reference hands raw strings over at src/semcode/services/indexer.py:150).
//
// Scope: the ASCII path of BERT's BasicTokenizer + WordPiece, exactly (clean control characters, split on
// whitespace, lower-case, split punctuation, greedy longest-match-first pieces, [UNK] for unmatched or
// > 100-character words, [CLS] ... [SEP] with truncation).  Source code is overwhelmingly ASCII; a text with any
// byte >= 0x80 is flagged instead of tokenised, and the Python tokenizer (semcode_amd/embeddings/tokenizer.py,
// which carries the Unicode rules: NFD accent stripping, CJK spacing, Unicode punctuation) handles it.
#include <atomic>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "sc_internal.h"

struct sc_tokenizer {
    std::unordered_map<std::string, int32_t> vocab;
    int32_t vocab_size = 0, pad = 0, unk = -1, cls = -1, sep = -1;
    bool lowercase = true;
    size_t max_piece = 0;
};

extern "C" sc_status sc_tokenizer_create(const char* vocab_utf8, size_t nbytes, int32_t lowercase, sc_tokenizer** out) {
    if (!vocab_utf8 || !out) return sc_fail(SC_ERR_INVALID, "sc_tokenizer_create: NULL argument");
    *out = nullptr;
    sc_tokenizer* t = new (std::nothrow) sc_tokenizer();
    if (!t) return sc_fail(SC_ERR_NOMEM, "out of host memory");
    t->lowercase = lowercase != 0;
    int32_t id = 0;
    size_t start = 0;
    for (size_t i = 0; i <= nbytes; ++i) {
        if (i == nbytes || vocab_utf8[i] == '\n') {
            if (i == nbytes && start == i) break;  // no trailing empty line
            size_t end = i;
            if (end > start && vocab_utf8[end - 1] == '\r') --end;
            std::string tok(vocab_utf8 + start, end - start);
            t->vocab.emplace(tok, id);  // first occurrence wins, like a dict built in file order would keep the last; vocab files have no duplicates
            if (tok.size() > t->max_piece) t->max_piece = tok.size();
            ++id;
            start = i + 1;
        }
    }
    t->vocab_size = id;
    auto find = [&](const char* s) { auto it = t->vocab.find(s); return it == t->vocab.end() ? -1 : it->second; };
    t->unk = find("[UNK]"); t->cls = find("[CLS]"); t->sep = find("[SEP]");
    const int32_t pad = find("[PAD]");
    t->pad = pad < 0 ? 0 : pad;
    if (t->unk < 0 || t->cls < 0 || t->sep < 0) {
        delete t;
        return sc_fail(SC_ERR_INVALID, "sc_tokenizer_create: vocabulary lacks [UNK], [CLS] or [SEP]");
    }
    *out = t;
    return SC_OK;
}

extern "C" sc_status sc_tokenizer_destroy(sc_tokenizer* t) {
    delete t;
    return SC_OK;
}

extern "C" sc_status sc_tokenizer_info(sc_tokenizer* t, int32_t* vocab_size, int32_t* pad, int32_t* unk, int32_t* cls, int32_t* sep) {
    if (!t) return sc_fail(SC_ERR_INVALID, "tokenizer is NULL");
    if (vocab_size) *vocab_size = t->vocab_size;
    if (pad) *pad = t->pad;
    if (unk) *unk = t->unk;
    if (cls) *cls = t->cls;
    if (sep) *sep = t->sep;
    return SC_OK;
}

static inline bool ascii_punct(unsigned char c) { return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126); }
static inline bool ascii_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
// str.split() also splits on \x0b \x0c \x1c-\x1f, but BasicTokenizer._clean_text removes those (category Cc) first
static inline bool ascii_dropped(unsigned char c) { return c == 0 || (c < 32 && !ascii_space(c)) || c == 127; }

// appends the pieces of one word; false when the id budget is reached
static void wordpiece(const sc_tokenizer* t, const std::string& w, std::vector<int32_t>& ids, std::string& scratch) {
    if (w.size() > 100) { ids.push_back(t->unk); return; }
    const size_t mark = ids.size();
    size_t start = 0;
    while (start < w.size()) {
        size_t end = w.size();
        int32_t cur = -1;
        while (start < end) {
            scratch.clear();
            if (start > 0) scratch = "##";
            scratch.append(w, start, end - start);
            auto it = t->vocab.find(scratch);
            if (it != t->vocab.end()) { cur = it->second; break; }
            --end;
        }
        if (cur < 0) { ids.resize(mark); ids.push_back(t->unk); return; }
        ids.push_back(cur);
        start = end;
    }
}

static void encode_one(const sc_tokenizer* t, const char* s, size_t n, int max_tokens, std::vector<int32_t>& ids, bool& non_ascii) {
    ids.clear();
    non_ascii = false;
    for (size_t i = 0; i < n; ++i)
        if ((unsigned char)s[i] >= 0x80) { non_ascii = true; return; }
    ids.push_back(t->cls);
    std::string word, scratch;
    bool full = false;
    auto flush = [&]() {
        if (!word.empty() && !full) {
            wordpiece(t, word, ids, scratch);
            if ((int)ids.size() >= max_tokens - 1) full = true;  // same stopping rule as tokenizer.py encode()
        }
        word.clear();
    };
    for (size_t i = 0; i < n && !full; ++i) {
        unsigned char c = (unsigned char)s[i];
        if (ascii_dropped(c)) continue;
        if (ascii_space(c)) { flush(); continue; }
        if (ascii_punct(c)) {
            flush();
            if (!full) { word.assign(1, (char)c); flush(); }
            continue;
        }
        if (t->lowercase && c >= 'A' && c <= 'Z') c = (unsigned char)(c + 32);
        word.push_back((char)c);
    }
    flush();
    if ((int)ids.size() > max_tokens - 1) ids.resize((size_t)(max_tokens - 1));
    ids.push_back(t->sep);
}

extern "C" sc_status sc_tokenizer_encode(sc_tokenizer* t, const char* bytes, const int64_t* offsets, int32_t n, int32_t max_tokens, int32_t S,
                                         int32_t* ids, int32_t* lens, uint8_t* needs_fallback, int32_t threads) {
    if (!t || (!bytes && n > 0) || !offsets || !ids || !lens || !needs_fallback) return sc_fail(SC_ERR_INVALID, "sc_tokenizer_encode: NULL argument");
    if (n < 0 || max_tokens < 2 || S < 2) return sc_fail(SC_ERR_INVALID, "sc_tokenizer_encode: need n >= 0, max_tokens >= 2, S >= 2");
    const int limit = max_tokens < S ? max_tokens : S;
    int nthreads = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if (nthreads > n) nthreads = n > 0 ? n : 1;
    std::atomic<int32_t> next(0);
    auto work = [&]() {
        std::vector<int32_t> v;
        for (;;) {
            const int32_t i = next.fetch_add(1);
            if (i >= n) break;
            bool na = false;
            encode_one(t, bytes + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), limit, v, na);
            needs_fallback[i] = na ? 1 : 0;
            int32_t* row = ids + (size_t)i * S;
            const int32_t len = na ? 0 : (int32_t)v.size();
            for (int32_t j = 0; j < S; ++j) row[j] = j < len ? v[(size_t)j] : t->pad;
            lens[i] = len;
        }
    };
    std::vector<std::thread> pool;
    for (int k = 1; k < nthreads; ++k) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    return SC_OK;
}

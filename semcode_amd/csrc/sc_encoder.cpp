// sc_encoder.cpp -- host side of the transformer-encoder entry points (include/semcode_hip.h).
//
// Mirrors (reference): the object EmbeddingProviderFactory.create() returns and its
// embed_documents / embed_query (src/semcode/embeddings/providers.py:34-104; third-party forward in
// llama.cpp): token ids in, one pooled vector per chunk out.  Tokenisation stays on the host side of
// the boundary (semcode_amd/embeddings/tokenizer.py).
//
// Forward per layer (BERT, post-LN):  QKV = X Wqkv^T + b  ->  attention  ->  Y = ctx Wo^T + bo + X
//   -> X1 = LN(Y) -> Hm = gelu(X1 W1^T + b1) -> Y2 = Hm W2^T + b2 + X1 -> X = LN(Y2); pooled = masked mean.
// Activations bf16 in HBM ([tokens, H] row-major, tokens padded to 128), weights bf16 [out, in],
// biases / LayerNorm parameters / embedding tables f32.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "sc_internal.h"

// gemm_bf16.hip / encoder_ops.hip
bool sc_gemm_bf16_supported(int M, int N, int K);
// splitk_scratch (optional, f32): lets small-M GEMMs run as tile x K-slice workgroups + a reduce/epilogue kernel (gemm_bf16.hip)
void sc_launch_gemm_bf16(int epi, const void* A, int lda, const void* W, int ldw, const float* bias, const void* R, int ldr, void* C,
                         int ldc, int M, int N, int K, hipStream_t s, void* splitk_scratch = nullptr, size_t splitk_scratch_bytes = 0);
int sc_gemm_splitk_factor(int M, int N, int K, int cus);
void sc_launch_embed_ln(const int32_t* ids, int tokens, int S, int H, int vocab, int max_pos, const float* wemb, const float* pemb,
                        const float* temb, const float* g, const float* b, float eps, void* out, hipStream_t s);
void sc_launch_layernorm(const void* in, int tokens, int H, const float* g, const float* b, float eps, void* out, hipStream_t s);
bool sc_attention_supported(int S, int H, int heads);
void sc_launch_attention(const void* qkv, const int32_t* lens, int B, int S, int H, const float* slopes, void* ctx, hipStream_t s, int blocked = 0);
void sc_launch_geglu(const void* h, int64_t tokens, int F, void* out, hipStream_t s);
// LayerNorm-folded batch pipeline (gemm_bf16.hip EPI_LNA_* / EPI_RESLN_STATS, encoder_ops.hip)
bool sc_gemm_ln_supported(int M, int N, int K);
void sc_launch_gemm_bf16_ln(int epi, const void* A, int lda, const void* W, int ldw, const float* bias, const void* R, int ldr, void* C, int ldc, int M,
                            int N, int K, hipStream_t s, const float* c1, const float* stats_in, float* fin, const float* gam, float* stats_out, float eps);
void sc_launch_embed_raw(const int32_t* ids, int tokens, int tokens_pad, int S, int H, int vocab, int max_pos, const float* wemb, const float* pemb,
                         const float* temb, void* out, float* stats, int slots, hipStream_t s);
void sc_launch_fold_ln_weights(const float* W, const float* gamma, const float* beta, const float* bias, int N, int K, void* Wf, float* c1, float* c2,
                               hipStream_t s);
void sc_launch_add_vectors(const float* a, const float* b, float* out, int n, hipStream_t s);
void sc_launch_mean_pool_ln(const void* y, const float* stats, int slots, int tokens_pad, const float* gamma, const float* beta, float eps,
                            const int32_t* lens, int B, int S, int H, float* out, hipStream_t s);
void sc_launch_mean_pool(const void* x, const int32_t* lens, int B, int S, int H, int normalize, float* out, hipStream_t s);
void sc_launch_f32_to_bf16(const float* in, void* out, int64_t n, hipStream_t s);
void sc_launch_synth_scaled(float* out, int64_t n, uint64_t seed, float scale, float offset, hipStream_t s);
void sc_launch_bf16_to_f32(const void* in, float* out, int64_t n, hipStream_t s);

void sc_gemm_set_debug(int v);
void sc_gemm_set_order(int v);
void sc_gemm_set_pp(int v);
void sc_gemm_set_trace(unsigned long long* dev);
void sc_gemm_force_tile128(bool on);

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RES = 2, EPI_LNA_BIAS = 3, EPI_LNA_GELU = 4, EPI_RESLN_STATS = 5 };

struct LayerW {
    void* wqkv;  // bf16 [3H, H]
    float* bqkv; // [3H]
    void* wo;    // bf16 [H, H]
    float* bo;
    float *ln1g, *ln1b;
    void* w1;    // bf16 [F, H]
    float* b1;
    void* w2;    // bf16 [H, F]
    float* b2;
    float *ln2g, *ln2b;
    // LayerNorm-folded copies for the batch pipeline (forward_folded): the LayerNorm in FRONT of a GEMM lives in its weights --
    // Wqkv' = Wqkv diag(gamma of the previous LayerNorm), W1' = W1 diag(ln1 gamma) -- and in two vectors per GEMM (gemm_bf16.hip)
    void *wqkv_f = nullptr, *w1_f = nullptr;
    float *c1q = nullptr, *c2q = nullptr, *c1f = nullptr, *c2f = nullptr;
    const float* g_prev = nullptr;  // gamma of the LayerNorm that produces this layer's input (embedding LN or the previous layer's ln2)
    float *bb_o = nullptr, *bb_2 = nullptr;  // bo + beta_prev, b2 + ln1 beta: the residual epilogues add the normalised residual's beta with the bias
};

struct sc_encoder {
    sc_runtime* rt = nullptr;
    sc_encoder_cfg cfg{};
    char* params = nullptr;  // one device allocation holding every parameter
    size_t params_bytes = 0;
    float *wemb = nullptr, *pemb = nullptr, *temb = nullptr, *embg = nullptr, *embb = nullptr;
    float* slopes = nullptr;  // ALiBi head slopes (device) or NULL
    std::vector<LayerW> layers;
    // workspace for `ws_tokens` (multiple of 256) tokens
    int64_t ws_tokens = 0;
    char* ws = nullptr;
    void *x = nullptr, *x1 = nullptr, *y = nullptr, *qkv = nullptr, *ctx = nullptr, *hm = nullptr, *hg = nullptr;
    int32_t* ids = nullptr;
    int32_t* lens = nullptr;
    float* pooled = nullptr;
    int64_t ws_batch = 0;
    float *stat_a = nullptr, *stat_b = nullptr;  // [slots][ws_tokens][2] partial row sums of the two pre-LayerNorm tensors (folded pipeline)
    float *fin_a = nullptr, *fin_b = nullptr;    // [ws_tokens][2] their finalised (mu, rs)
    bool foldable = false;                       // shapes allow the folded pipeline (256-tile GEMMs, K % 256 == 0)
    int path = 0;                                // sc_encoder_set_path: 0 auto, 1 batch pipeline always, 2 small-batch pipeline always
    void* splitk = nullptr;  // f32 partial products of the split-K GEMMs (batches of <= 1024 tokens), allocated on first use
    static constexpr size_t SPLITK_BYTES = 64u << 20;
    // pinned host staging for the asynchronous embed -> index path: ids | lens | rows of one batch per slot
    struct PinSlot {
        char* host = nullptr;
        size_t cap = 0;
        hipEvent_t done = nullptr;
        bool busy = false;
    } pin[2];
    int pin_next = 0;
    std::mutex mu;
};

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// number of f32 values in the weight blob, in blob order (see include/semcode_hip.h)
static int64_t blob_floats(const sc_encoder_cfg& c) {
    const int64_t H = c.hidden, F = c.ffn, F1 = c.ffn_type == 1 ? 2 * F : F;
    int64_t n = (int64_t)c.vocab * H + (c.pos_type == 1 ? 0 : (int64_t)c.max_pos * H) + (int64_t)c.type_vocab * H + 2 * H;
    n += (int64_t)c.layers * (4 * (H * H + H) + 2 * H + (F1 * H + F1) + (H * F + H) + 2 * H);
    return n;
}

extern "C" sc_status sc_encoder_blob_bytes(const sc_encoder_cfg* cfg, int64_t* out) {
    if (!cfg || !out) return sc_fail(SC_ERR_INVALID, "sc_encoder_blob_bytes: NULL argument");
    *out = blob_floats(*cfg) * 4;
    return SC_OK;
}

static sc_status check_cfg(const sc_encoder_cfg& c) {
    if (c.vocab < 1 || c.hidden < 1 || c.layers < 1 || c.heads < 1 || c.ffn < 1 || c.max_pos < 1 || c.type_vocab < 1)
        return sc_fail(SC_ERR_INVALID, "sc_encoder_create: non-positive model dimension");
    if (c.hidden != c.heads * 64) return sc_fail(SC_ERR_UNSUPPORTED, "sc_encoder_create: head dimension must be 64 (hidden=%d heads=%d)", c.hidden, c.heads);
    if (c.hidden % 128 || c.ffn % 128 || c.hidden > 2048)
        return sc_fail(SC_ERR_UNSUPPORTED, "sc_encoder_create: hidden (<=2048) and ffn must be multiples of 128 (got %d, %d)", c.hidden, c.ffn);
    if (!(c.ln_eps > 0.f)) return sc_fail(SC_ERR_INVALID, "sc_encoder_create: ln_eps must be > 0");
    if (c.pos_type < 0 || c.pos_type > 1 || c.ffn_type < 0 || c.ffn_type > 1) return sc_fail(SC_ERR_INVALID, "sc_encoder_create: unknown pos_type / ffn_type");
    return SC_OK;
}

extern "C" sc_status sc_encoder_create(sc_runtime* rt, const sc_encoder_cfg* cfg, const void* weights_blob, size_t nbytes, sc_encoder** out) {
    if (!rt || !cfg || !out) return sc_fail(SC_ERR_INVALID, "sc_encoder_create: NULL argument");
    *out = nullptr;
    sc_status st = check_cfg(*cfg);
    if (st) return st;
    const int64_t nfl = blob_floats(*cfg);
    if (weights_blob && (int64_t)nbytes != nfl * 4)
        return sc_fail(SC_ERR_INVALID, "sc_encoder_create: weight blob is %zu bytes, config needs %lld", nbytes, (long long)(nfl * 4));
    SC_HIP(hipSetDevice(rt->device));
    hipStream_t s = rt->stream;
    sc_encoder* e = new (std::nothrow) sc_encoder();
    if (!e) return sc_fail(SC_ERR_NOMEM, "out of host memory");
    e->rt = rt;
    sc_runtime_retain(rt);
    e->cfg = *cfg;
    const int64_t H = cfg->hidden, F = cfg->ffn, L = cfg->layers, F1 = cfg->ffn_type == 1 ? 2 * F : F;
    const bool alibi = cfg->pos_type == 1;

    // staging copy of the f32 blob on device (freed after conversion)
    float* blob = nullptr;
    hipError_t he = hipMalloc((void**)&blob, (size_t)nfl * 4);
    if (he != hipSuccess) {
        delete e;
        sc_runtime_release(rt);
        return sc_fail(SC_ERR_NOMEM, "hipMalloc weight staging (%lld B) failed: %s", (long long)nfl * 4, hipGetErrorString(he));
    }
    auto fail = [&](sc_status code) {
        hipFree(blob);
        hipFree(e->params);
        delete e;
        sc_runtime_release(rt);
        return code;
    };
    // parameter arena: f32 tables + per-layer {bf16 matrices, f32 vectors}
    size_t total = 0;
    auto reserve = [&](size_t bytes) { size_t o = total; total += align256(bytes); return o; };
    const size_t o_wemb = reserve((size_t)cfg->vocab * H * 4), o_pemb = reserve(alibi ? 16 : (size_t)cfg->max_pos * H * 4),
                 o_slopes = reserve((size_t)cfg->heads * 4),
                 o_temb = reserve((size_t)cfg->type_vocab * H * 4), o_eg = reserve(H * 4), o_eb = reserve(H * 4);
    struct LO { size_t wqkv, bqkv, wo, bo, l1g, l1b, w1, b1, w2, b2, l2g, l2b, wqkv_f, c1q, c2q, w1_f, c1f, c2f, bb_o, bb_2; };
    std::vector<LO> lo(L);
    // the LayerNorm-folded batch pipeline needs every GEMM on the 256 x 256 tile and the statistics in whole 256-column slots
    e->foldable = (H % 256) == 0 && (F % 256) == 0 && (F1 % 256) == 0 && ((3 * H) % 256) == 0;
    for (int64_t l = 0; l < L; ++l) {
        lo[l].wqkv = reserve(3 * H * H * 2); lo[l].bqkv = reserve(3 * H * 4); lo[l].wo = reserve(H * H * 2); lo[l].bo = reserve(H * 4);
        lo[l].l1g = reserve(H * 4); lo[l].l1b = reserve(H * 4); lo[l].w1 = reserve(F1 * H * 2); lo[l].b1 = reserve(F1 * 4);
        lo[l].w2 = reserve(H * F * 2); lo[l].b2 = reserve(H * 4); lo[l].l2g = reserve(H * 4); lo[l].l2b = reserve(H * 4);
        if (e->foldable) {
            lo[l].wqkv_f = reserve(3 * H * H * 2); lo[l].c1q = reserve(3 * H * 4); lo[l].c2q = reserve(3 * H * 4);
            lo[l].w1_f = reserve(F1 * H * 2); lo[l].c1f = reserve(F1 * 4); lo[l].c2f = reserve(F1 * 4);
            lo[l].bb_o = reserve(H * 4); lo[l].bb_2 = reserve(H * 4);
        }
    }
    he = hipMalloc((void**)&e->params, total);
    if (he != hipSuccess) return fail(sc_fail(SC_ERR_NOMEM, "hipMalloc parameters (%zu B) failed: %s", total, hipGetErrorString(he)));
    e->params_bytes = total;

    if (weights_blob) {
        he = hipMemcpyAsync(blob, weights_blob, (size_t)nfl * 4, hipMemcpyHostToDevice, s);
        if (he != hipSuccess) return fail(sc_fail(SC_ERR_HIP, "weight upload failed: %s", hipGetErrorString(he)));
    } else {
        // synthetic weights: every tensor ~ 0.02 * N(0,1) over the blob index, then LayerNorm gamma = 1,
        // beta = 0 and biases = 0 are overwritten below (same rule in oracle/bert_oracle.py synth_weights)
        sc_launch_synth_scaled(blob, nfl, cfg->synth_seed, 0.02f, 0.0f, s);
    }
    // walk the blob in its documented order
    int64_t off = 0;
    auto take = [&](int64_t n) { float* p = blob + off; off += n; return p; };
    auto put_f32 = [&](size_t dst, const float* src, int64_t n, int fill /*0 copy, 1 ones, 2 zeros*/) {
        float* d = (float*)(e->params + dst);
        if (!weights_blob && fill == 1) sc_launch_synth_scaled(d, n, 0, 0.0f, 1.0f, s);
        else if (!weights_blob && fill == 2) hipMemsetAsync(d, 0, (size_t)n * 4, s);
        else hipMemcpyAsync(d, src, (size_t)n * 4, hipMemcpyDeviceToDevice, s);
        return d;
    };
    auto put_bf16 = [&](size_t dst, const float* src, int64_t n) {
        void* d = e->params + dst;
        sc_launch_f32_to_bf16(src, d, n, s);
        return d;
    };
    e->wemb = put_f32(o_wemb, take((int64_t)cfg->vocab * H), (int64_t)cfg->vocab * H, 0);
    e->pemb = alibi ? nullptr : put_f32(o_pemb, take((int64_t)cfg->max_pos * H), (int64_t)cfg->max_pos * H, 0);
    if (alibi) {  // ALiBi head slopes (Press et al.; the non-power-of-two rule of the jina-bert implementation)
        std::vector<float> sl((size_t)cfg->heads);
        auto pow2_slopes = [](int n, std::vector<float>& out, int take, int stride) {
            const double start = std::pow(2.0, -std::pow(2.0, -(std::log2((double)n) - 3.0)));
            double v = start;
            for (int i = 0, got = 0; i < n && got < take; ++i, v *= start)
                if (i % stride == 0) { out.push_back((float)v); ++got; }
        };
        std::vector<float> tmp;
        int p2 = 1;
        while (p2 * 2 <= cfg->heads) p2 *= 2;
        pow2_slopes(p2, tmp, p2, 1);
        if (p2 < cfg->heads) pow2_slopes(2 * p2, tmp, cfg->heads - p2, 2);
        for (int h = 0; h < cfg->heads; ++h) sl[(size_t)h] = tmp[(size_t)h];
        e->slopes = (float*)(e->params + o_slopes);
        hipMemcpyAsync(e->slopes, sl.data(), sl.size() * 4, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);  // sl goes out of scope
    }
    e->temb = put_f32(o_temb, take((int64_t)cfg->type_vocab * H), (int64_t)cfg->type_vocab * H, 0);
    e->embg = put_f32(o_eg, take(H), H, 1);
    e->embb = put_f32(o_eb, take(H), H, 2);
    e->layers.resize(L);
    for (int64_t l = 0; l < L; ++l) {
        LayerW& w = e->layers[l];
        // blob order: Wq bq Wk bk Wv bv Wo bo ln1g ln1b W1 b1 W2 b2 ln2g ln2b ; device: Wqkv = [Wq; Wk; Wv]
        char* wqkv = e->params + lo[l].wqkv;
        float* bqkv = (float*)(e->params + lo[l].bqkv);
        const float* wqkv_f32[3];
        for (int p = 0; p < 3; ++p) {
            const float* wp = take(H * H);
            wqkv_f32[p] = wp;
            sc_launch_f32_to_bf16(wp, wqkv + (size_t)p * H * H * 2, H * H, s);
            const float* bp = take(H);
            if (weights_blob) hipMemcpyAsync(bqkv + p * H, bp, H * 4, hipMemcpyDeviceToDevice, s);
            else hipMemsetAsync(bqkv + p * H, 0, H * 4, s);
        }
        w.wqkv = wqkv;
        w.bqkv = bqkv;
        w.wo = put_bf16(lo[l].wo, take(H * H), H * H);
        w.bo = put_f32(lo[l].bo, take(H), H, 2);
        w.ln1g = put_f32(lo[l].l1g, take(H), H, 1);
        w.ln1b = put_f32(lo[l].l1b, take(H), H, 2);
        const float* w1_f32 = take(F1 * H);
        w.w1 = put_bf16(lo[l].w1, w1_f32, F1 * H);
        w.b1 = put_f32(lo[l].b1, take(F1), F1, 2);
        w.w2 = put_bf16(lo[l].w2, take(H * F), H * F);
        w.b2 = put_f32(lo[l].b2, take(H), H, 2);
        w.ln2g = put_f32(lo[l].l2g, take(H), H, 1);
        w.ln2b = put_f32(lo[l].l2b, take(H), H, 2);
        if (e->foldable) {
            // the LayerNorm in front of this layer: the embeddings' for layer 0, else the previous layer's second one
            const float* gp = l == 0 ? e->embg : e->layers[l - 1].ln2g;
            const float* bp = l == 0 ? e->embb : e->layers[l - 1].ln2b;
            w.g_prev = gp;
            w.wqkv_f = e->params + lo[l].wqkv_f;
            w.c1q = (float*)(e->params + lo[l].c1q); w.c2q = (float*)(e->params + lo[l].c2q);
            for (int p = 0; p < 3; ++p)  // Wq, Wk, Wv are separate tensors of the blob
                sc_launch_fold_ln_weights(wqkv_f32[p], gp, bp, bqkv + p * H, (int)H, (int)H, (char*)w.wqkv_f + (size_t)p * H * H * 2, w.c1q + p * H, w.c2q + p * H, s);
            w.w1_f = e->params + lo[l].w1_f;
            w.c1f = (float*)(e->params + lo[l].c1f); w.c2f = (float*)(e->params + lo[l].c2f);
            sc_launch_fold_ln_weights(w1_f32, w.ln1g, w.ln1b, w.b1, (int)F1, (int)H, w.w1_f, w.c1f, w.c2f, s);
            w.bb_o = (float*)(e->params + lo[l].bb_o); w.bb_2 = (float*)(e->params + lo[l].bb_2);
            sc_launch_add_vectors(w.bo, bp, w.bb_o, (int)H, s);
            sc_launch_add_vectors(w.b2, w.ln1b, w.bb_2, (int)H, s);
        }
    }
    he = hipStreamSynchronize(s);
    if (he == hipSuccess) he = hipGetLastError();
    if (he != hipSuccess) return fail(sc_fail(SC_ERR_HIP, "encoder parameter setup failed: %s", hipGetErrorString(he)));
    hipFree(blob);
    *out = e;
    return SC_OK;
}

extern "C" sc_status sc_encoder_destroy(sc_encoder* e) {
    if (!e) return SC_OK;
    hipSetDevice(e->rt->device);
    hipStreamSynchronize(e->rt->stream);
    hipFree(e->params);
    hipFree(e->ws);
    hipFree(e->splitk);
    for (auto& slot : e->pin) {
        if (slot.host) hipHostFree(slot.host);
        if (slot.done) hipEventDestroy(slot.done);
    }
    sc_runtime* rt = e->rt;
    delete e;
    sc_runtime_release(rt);
    return SC_OK;
}

static sc_status ensure_ws(sc_encoder* e, int64_t B, int64_t S) {
    const int64_t tokens = (B * S + 255) / 256 * 256;  // GEMM tiles are 256 rows
    if (tokens <= e->ws_tokens && B <= e->ws_batch) return SC_OK;
    SC_HIP(hipStreamSynchronize(e->rt->stream));
    hipFree(e->ws);
    e->ws = nullptr;
    e->ws_tokens = 0;
    e->ws_batch = 0;
    const int64_t H = e->cfg.hidden, F = e->cfg.ffn, F1 = e->cfg.ffn_type == 1 ? 2 * F : F;
    const int64_t nb = B > e->ws_batch ? B : e->ws_batch;
    size_t total = 0;
    auto reserve = [&](size_t bytes) { size_t o = total; total += align256(bytes); return o; };
    const size_t ox = reserve(tokens * H * 2), ox1 = reserve(tokens * H * 2), oy = reserve(tokens * H * 2), oqkv = reserve(tokens * 3 * H * 2),
                 octx = reserve(tokens * H * 2), ohm = reserve(tokens * F1 * 2), ohg = reserve(e->cfg.ffn_type == 1 ? tokens * F * 2 : 16),
                 oids = reserve(tokens * 4), olens = reserve(nb * 4), opool = reserve(nb * H * 4);
    const size_t slots = (size_t)H / 256;
    const size_t osa = reserve(e->foldable ? slots * tokens * 8 : 16), osb = reserve(e->foldable ? slots * tokens * 8 : 16),
                 ofa = reserve(e->foldable ? tokens * 8 : 16), ofb = reserve(e->foldable ? tokens * 8 : 16);
    hipError_t he = hipMalloc((void**)&e->ws, total);
    if (he != hipSuccess) return sc_fail(SC_ERR_NOMEM, "hipMalloc encoder workspace (%zu B) failed: %s", total, hipGetErrorString(he));
    SC_HIP(hipMemsetAsync(e->ws, 0, total, e->rt->stream));  // padded rows must hold finite values
    e->x = e->ws + ox; e->x1 = e->ws + ox1; e->y = e->ws + oy; e->qkv = e->ws + oqkv; e->ctx = e->ws + octx; e->hm = e->ws + ohm; e->hg = e->ws + ohg;
    e->ids = (int32_t*)(e->ws + oids); e->lens = (int32_t*)(e->ws + olens); e->pooled = (float*)(e->ws + opool);
    e->stat_a = (float*)(e->ws + osa); e->stat_b = (float*)(e->ws + osb); e->fin_a = (float*)(e->ws + ofa); e->fin_b = (float*)(e->ws + ofb);
    e->ws_tokens = tokens;
    e->ws_batch = nb;
    return SC_OK;
}

// The batch pipeline with every LayerNorm folded into its neighbours (no layernorm_kernel launch, no LayerNorm round trip
// through HBM: 24 x 33 us and 24 x 200 MB per step at 256 x 256 tokens).  Activations between layers are the PRE-LayerNorm
// tensors (raw bf16 rows) together with their row statistics:
//   embed_raw                     -> x  (raw) + stat_b            x = word + position + type embeddings
//   per layer, LN_p = the LayerNorm that belongs in front of it (embeddings' / previous layer's second):
//     QKV   = LN_p(x) Wqkv^T + b   as  rs (x Wqkv'^T - mu c1) + c2            EPI_LNA_BIAS   (stat_b -> fin_b)
//     y     = ctx Wo^T + bo + LN_p(x)    residual normalised on the fly, row sums of y out   EPI_RESLN_STATS (fin_b -> stat_a)
//     hm    = gelu(LN_1(y) W1^T + b1)    folded the same way                                 EPI_LNA_GELU   (stat_a -> fin_a)
//     x     = hm W2^T + b2 + LN_1(y)                                                         EPI_RESLN_STATS (fin_a -> stat_b)
//   pooled = masked mean of LN_2(x) of the last layer, normalised on the fly (mean_pool_ln)
// Statistics are taken of the bf16-ROUNDED rows, i.e. of exactly what the consumer reads; all reductions run in a fixed order.
static sc_status forward_folded_locked(sc_encoder* e, const int32_t* ids_dev, const int32_t* lens_dev, int32_t B, int32_t S, float* out_dev) {
    const sc_encoder_cfg& c = e->cfg;
    sc_runtime* rt = e->rt;
    hipStream_t s = rt->stream;
    const int H = c.hidden, F = c.ffn;
    const int tokens = B * S;
    const int M = (tokens + 255) / 256 * 256;
    const int slots = H / 256;
    // statistics buffers are laid out for ws_tokens rows; this call uses the first M rows of every slot: slot stride must be M, so
    // they are addressed as [slots][M][2] inside the (larger or equal) allocation
    static const char* env_fb = getenv("SC_FFN_BLOCKED");
    const bool ffn_blocked = c.ffn_type != 1 && !(env_fb && env_fb[0] == '0');
    sc_launch_embed_raw(ids_dev, tokens, M, S, H, c.vocab, c.max_pos, e->wemb, e->pemb, e->temb, e->x, e->stat_b, slots, s);
    for (int l = 0; l < c.layers; ++l) {
        const LayerW& w = e->layers[l];
        hipEvent_t g0, g1;
        sc_prof_begin(rt, SC_PROF_GEMM, &g0, &g1);
        sc_launch_gemm_bf16_ln(EPI_LNA_BIAS, e->x, H, w.wqkv_f, H, w.c2q, nullptr, 0, e->qkv, SC_LDC_BLOCKED64, M, 3 * H, H, s, w.c1q, e->stat_b, e->fin_b, nullptr,
                               nullptr, c.ln_eps);
        sc_prof_end(rt, SC_PROF_GEMM, g0, g1);
        hipEvent_t a0, a1;
        sc_prof_begin(rt, SC_PROF_ATTN, &a0, &a1);
        sc_launch_attention(e->qkv, lens_dev, B, S, H, e->slopes, e->ctx, s, M);
        sc_prof_end(rt, SC_PROF_ATTN, a0, a1);
        sc_prof_begin(rt, SC_PROF_GEMM, &g0, &g1);
        sc_launch_gemm_bf16_ln(EPI_RESLN_STATS, e->ctx, H, w.wo, H, w.bb_o, e->x, H, e->y, H, M, H, H, s, nullptr, nullptr, e->fin_b, w.g_prev, e->stat_a, c.ln_eps);
        sc_prof_end(rt, SC_PROF_GEMM, g0, g1);
        const void* ffn_in = e->hm;
        sc_prof_begin(rt, SC_PROF_GEMM, &g0, &g1);
        if (c.ffn_type == 1)
            sc_launch_gemm_bf16_ln(EPI_LNA_BIAS, e->y, H, w.w1_f, H, w.c2f, nullptr, 0, e->hm, 2 * F, M, 2 * F, H, s, w.c1f, e->stat_a, e->fin_a, nullptr, nullptr, c.ln_eps);
        else
            sc_launch_gemm_bf16_ln(EPI_LNA_GELU, e->y, H, w.w1_f, H, w.c2f, nullptr, 0, e->hm, ffn_blocked ? SC_LDC_BLOCKED64 : F, M, F, H, s, w.c1f, e->stat_a,
                                   e->fin_a, nullptr, nullptr, c.ln_eps);
        sc_prof_end(rt, SC_PROF_GEMM, g0, g1);
        if (c.ffn_type == 1) {
            sc_launch_geglu(e->hm, M, F, e->hg, s);
            ffn_in = e->hg;
        }
        sc_prof_begin(rt, SC_PROF_GEMM, &g0, &g1);
        sc_launch_gemm_bf16_ln(EPI_RESLN_STATS, ffn_in, (ffn_blocked && c.ffn_type != 1) ? SC_LDC_BLOCKED64 : F, w.w2, F, w.bb_2, e->y, H, e->x, H, M, H, F, s, nullptr,
                               nullptr, e->fin_a, w.ln1g, e->stat_b, c.ln_eps);
        sc_prof_end(rt, SC_PROF_GEMM, g0, g1);
    }
    const LayerW& last = e->layers[c.layers - 1];
    if (c.normalize) {  // L2-normalised output: the plain pooling kernel does it; give it the normalised rows
        sc_launch_layernorm(e->x, tokens, H, last.ln2g, last.ln2b, c.ln_eps, e->x1, s);
        sc_launch_mean_pool(e->x1, lens_dev, B, S, H, c.normalize, out_dev, s);
    } else {
        sc_launch_mean_pool_ln(e->x, e->stat_b, slots, M, last.ln2g, last.ln2b, c.ln_eps, lens_dev, B, S, H, out_dev, s);
    }
    SC_HIP(hipGetLastError());
    return SC_OK;
}

// ids_dev [B,S], lens_dev [B] device pointers; out_dev [B,H] f32 device.  Caller holds e->mu.
static sc_status forward_locked(sc_encoder* e, const int32_t* ids_dev, const int32_t* lens_dev, int32_t B, int32_t S, float* out_dev) {
    const sc_encoder_cfg& c = e->cfg;
    sc_runtime* rt = e->rt;
    hipStream_t s = rt->stream;
    const int H = c.hidden, F = c.ffn;
    const int tokens = B * S;
    const int M = (tokens + 255) / 256 * 256;
    // batches beyond 1024 token rows (or sc_encoder_set_path 1) take the LayerNorm-folded pipeline where the model's shapes allow it
    static const char* env_fold = getenv("SC_ENC_FOLD");  // SC_ENC_FOLD=0: same-box A/B against the stand-alone LayerNorm kernels
    if (e->foldable && e->path != 2 && (M > 1024 || e->path == 1) && !(env_fold && env_fold[0] == '0'))
        return forward_folded_locked(e, ids_dev, lens_dev, B, S, out_dev);
    void* sk = nullptr;
    if (M <= 1024 && e->path != 1) {  // a query or a few chunks: too few tiles for the chip, split K (gemm_bf16.hip)
        if (!e->splitk) SC_HIP(hipMalloc(&e->splitk, sc_encoder::SPLITK_BYTES));
        sk = e->splitk;
    }
    const size_t skb = sk ? sc_encoder::SPLITK_BYTES : 0;
    // batch steps: the FFN hidden activations travel in 64-column blocks ([F / 64][M][64]) between FFN1's epilogue and FFN2's
    // K loop, so that a wave's 16 x 64 output block and a K-tile of an A row panel are contiguous (row-major: 128-byte pieces 6 KB
    // apart).  Only the 256-tile kernel reads that layout: not with split-K (small M), not on the GEGLU path (element-wise kernel
    // in between), and only when F divides into 256-column tiles.
    static const char* env_fb = getenv("SC_FFN_BLOCKED");
    const bool ffn_blocked = !sk && c.ffn_type != 1 && (F % 256) == 0 && (H % 256) == 0 && !(env_fb && env_fb[0] == '0');
    sc_launch_embed_ln(ids_dev, tokens, S, H, c.vocab, c.max_pos, e->wemb, e->pemb, e->temb, e->embg, e->embb, c.ln_eps, e->x, s);
    for (int l = 0; l < c.layers; ++l) {
        const LayerW& w = e->layers[l];
        hipEvent_t g0, g1;
        sc_prof_begin(rt, SC_PROF_GEMM, &g0, &g1);
        // QKV in 64-column blocks, i.e. [3 heads][tokens][64]: attention reads each (chunk, head) operand as one contiguous block
        sc_launch_gemm_bf16(EPI_BIAS, e->x, H, w.wqkv, H, w.bqkv, nullptr, 0, e->qkv, SC_LDC_BLOCKED64, M, 3 * H, H, s, sk, skb);
        sc_prof_end(rt, SC_PROF_GEMM, g0, g1);
        hipEvent_t a0, a1;
        sc_prof_begin(rt, SC_PROF_ATTN, &a0, &a1);
        sc_launch_attention(e->qkv, lens_dev, B, S, H, e->slopes, e->ctx, s, M);
        sc_prof_end(rt, SC_PROF_ATTN, a0, a1);
        sc_prof_begin(rt, SC_PROF_GEMM, &g0, &g1);
        sc_launch_gemm_bf16(EPI_BIAS_RES, e->ctx, H, w.wo, H, w.bo, e->x, H, e->y, H, M, H, H, s, sk, skb);
        sc_prof_end(rt, SC_PROF_GEMM, g0, g1);
        sc_launch_layernorm(e->y, tokens, H, w.ln1g, w.ln1b, c.ln_eps, e->x1, s);
        const void* ffn_in = e->hm;
        sc_prof_begin(rt, SC_PROF_GEMM, &g0, &g1);
        if (c.ffn_type == 1) sc_launch_gemm_bf16(EPI_BIAS, e->x1, H, w.w1, H, w.b1, nullptr, 0, e->hm, 2 * F, M, 2 * F, H, s, sk, skb);
        else sc_launch_gemm_bf16(EPI_BIAS_GELU, e->x1, H, w.w1, H, w.b1, nullptr, 0, e->hm, ffn_blocked ? SC_LDC_BLOCKED64 : F, M, F, H, s, sk, skb);
        sc_prof_end(rt, SC_PROF_GEMM, g0, g1);
        if (c.ffn_type == 1) {  // GEGLU: gelu(gate) * up
            sc_launch_geglu(e->hm, M, F, e->hg, s);
            ffn_in = e->hg;
        }
        sc_prof_begin(rt, SC_PROF_GEMM, &g0, &g1);
        sc_launch_gemm_bf16(EPI_BIAS_RES, ffn_in, ffn_blocked ? SC_LDC_BLOCKED64 : F, w.w2, F, w.b2, e->x1, H, e->y, H, M, H, F, s, sk, skb);
        sc_prof_end(rt, SC_PROF_GEMM, g0, g1);
        sc_launch_layernorm(e->y, tokens, H, w.ln2g, w.ln2b, c.ln_eps, e->x, s);
    }
    sc_launch_mean_pool(e->x, lens_dev, B, S, H, c.normalize, out_dev, s);
    SC_HIP(hipGetLastError());
    return SC_OK;
}

static sc_status check_embed_args(sc_encoder* e, const void* ids, const void* lens, int32_t B, int32_t S, const void* out) {
    if (!e || !ids || !lens || !out) return sc_fail(SC_ERR_INVALID, "embed: NULL argument");
    if (B < 1 || B > 65536) return sc_fail(SC_ERR_INVALID, "embed: batch %d out of range", B);
    if (!sc_attention_supported(S, e->cfg.hidden, e->cfg.heads))
        return sc_fail(SC_ERR_UNSUPPORTED, "embed: sequence length %d not in {32,64,128,256,512,1024,2048} (pad on the host)", S);
    if (S > e->cfg.max_pos && e->cfg.pos_type == 0) return sc_fail(SC_ERR_INVALID, "embed: sequence length %d exceeds max_pos %d", S, e->cfg.max_pos);
    return SC_OK;
}

extern "C" sc_status sc_encoder_embed_ids_dev(sc_encoder* e, const int32_t* ids_dev, const int32_t* lens_dev, int32_t B, int32_t S,
                                              float* out_dev) {
    sc_status st = check_embed_args(e, ids_dev, lens_dev, B, S, out_dev);
    if (st) return st;
    std::lock_guard<std::mutex> g(e->mu);
    SC_HIP(hipSetDevice(e->rt->device));
    st = ensure_ws(e, B, S);
    if (st) return st;
    return forward_locked(e, ids_dev, lens_dev, B, S, out_dev);
}

extern "C" sc_status sc_encoder_embed_ids(sc_encoder* e, const int32_t* ids, const int32_t* lens, int32_t B, int32_t S, float* out) {
    sc_status st = check_embed_args(e, ids, lens, B, S, out);
    if (st) return st;
    std::lock_guard<std::mutex> g(e->mu);
    SC_HIP(hipSetDevice(e->rt->device));
    st = ensure_ws(e, B, S);
    if (st) return st;
    hipStream_t s = e->rt->stream;
    SC_HIP(hipMemcpyAsync(e->ids, ids, (size_t)B * S * 4, hipMemcpyHostToDevice, s));
    SC_HIP(hipMemcpyAsync(e->lens, lens, (size_t)B * 4, hipMemcpyHostToDevice, s));
    st = forward_locked(e, e->ids, e->lens, B, S, e->pooled);
    if (st) return st;
    SC_HIP(hipMemcpyAsync(out, e->pooled, (size_t)B * e->cfg.hidden * 4, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    return SC_OK;
}

extern "C" sc_status sc_encoder_embed_ids_into(sc_encoder* e, const int32_t* ids, const int32_t* lens, int32_t B, int32_t S, sc_index* ix,
                                               const int64_t* rows, float* out) {
    if (!ix || !rows) return sc_fail(SC_ERR_INVALID, "sc_encoder_embed_ids_into: NULL index / rows");
    sc_status st = check_embed_args(e, ids, lens, B, S, rows);
    if (st) return st;
    if (ix->rt != e->rt) return sc_fail(SC_ERR_INVALID, "sc_encoder_embed_ids_into: encoder and index belong to different runtimes");
    if (ix->dim != e->cfg.hidden) return sc_fail(SC_ERR_INVALID, "sc_encoder_embed_ids_into: index dim %d != encoder hidden %d", ix->dim, e->cfg.hidden);
    std::lock_guard<std::mutex> g(e->mu);
    SC_HIP(hipSetDevice(e->rt->device));
    st = ensure_ws(e, B, S);
    if (st) return st;
    hipStream_t s = e->rt->stream;
    SC_HIP(hipMemcpyAsync(e->ids, ids, (size_t)B * S * 4, hipMemcpyHostToDevice, s));
    SC_HIP(hipMemcpyAsync(e->lens, lens, (size_t)B * 4, hipMemcpyHostToDevice, s));
    st = forward_locked(e, e->ids, e->lens, B, S, e->pooled);
    if (st) return st;
    {
        std::lock_guard<std::mutex> gi(ix->mu);  // lock order: encoder, then index (nothing takes them the other way round)
        st = sc_index_put_rows_locked(ix, e->pooled, true, rows, B, "sc_encoder_embed_ids_into");
    }
    if (st) {
        hipStreamSynchronize(s);
        return st;
    }
    if (out) SC_HIP(hipMemcpyAsync(out, e->pooled, (size_t)B * e->cfg.hidden * 4, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    return SC_OK;
}

// wait for the batch that last used `slot` (its inputs may be overwritten afterwards); reports a failure of that batch's device work
static sc_status pin_slot_wait(sc_encoder::PinSlot& slot) {
    if (!slot.busy) return SC_OK;
    slot.busy = false;
    hipError_t he = hipEventSynchronize(slot.done);
    if (he != hipSuccess) return sc_fail(SC_ERR_HIP, "asynchronous embed batch failed: %s", hipGetErrorString(he));
    return SC_OK;
}

extern "C" sc_status sc_encoder_embed_ids_into_async(sc_encoder* e, const int32_t* ids, const int32_t* lens, int32_t B, int32_t S, sc_index* ix,
                                                     const int64_t* rows) {
    if (!ix || !rows) return sc_fail(SC_ERR_INVALID, "sc_encoder_embed_ids_into_async: NULL index / rows");
    sc_status st = check_embed_args(e, ids, lens, B, S, rows);
    if (st) return st;
    if (ix->rt != e->rt) return sc_fail(SC_ERR_INVALID, "sc_encoder_embed_ids_into_async: encoder and index belong to different runtimes");
    if (ix->dim != e->cfg.hidden) return sc_fail(SC_ERR_INVALID, "sc_encoder_embed_ids_into_async: index dim %d != encoder hidden %d", ix->dim, e->cfg.hidden);
    std::lock_guard<std::mutex> g(e->mu);
    SC_HIP(hipSetDevice(e->rt->device));
    hipStream_t s = e->rt->stream;
    sc_encoder::PinSlot& slot = e->pin[e->pin_next];
    st = pin_slot_wait(slot);  // two batches may be in flight; the third waits for the first
    if (st) return st;
    const size_t ids_b = (size_t)B * S * 4, lens_b = align256((size_t)B * 4), rows_b = (size_t)B * 8, need = align256(ids_b) + lens_b + rows_b;
    if (need > slot.cap) {
        if (slot.host) hipHostFree(slot.host);
        slot.host = nullptr;
        slot.cap = 0;
        SC_HIP(hipHostMalloc((void**)&slot.host, need, hipHostMallocDefault));
        slot.cap = need;
    }
    if (!slot.done) SC_HIP(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    char* h_ids = slot.host;
    char* h_lens = slot.host + align256(ids_b);
    int64_t* h_rows = (int64_t*)(h_lens + lens_b);
    memcpy(h_ids, ids, ids_b);
    memcpy(h_lens, lens, (size_t)B * 4);
    memcpy(h_rows, rows, rows_b);
    st = ensure_ws(e, B, S);
    if (st) return st;
    SC_HIP(hipMemcpyAsync(e->ids, h_ids, ids_b, hipMemcpyHostToDevice, s));
    SC_HIP(hipMemcpyAsync(e->lens, h_lens, (size_t)B * 4, hipMemcpyHostToDevice, s));
    st = forward_locked(e, e->ids, e->lens, B, S, e->pooled);
    if (st) return st;
    {
        std::lock_guard<std::mutex> gi(ix->mu);
        st = sc_index_put_rows_locked(ix, e->pooled, true, h_rows, B, "sc_encoder_embed_ids_into_async");
    }
    if (st) return st;
    SC_HIP(hipEventRecord(slot.done, s));
    slot.busy = true;
    e->pin_next ^= 1;
    return SC_OK;
}

extern "C" sc_status sc_encoder_wait(sc_encoder* e) {
    if (!e) return sc_fail(SC_ERR_INVALID, "sc_encoder_wait: NULL encoder");
    std::lock_guard<std::mutex> g(e->mu);
    SC_HIP(hipSetDevice(e->rt->device));
    sc_status first = SC_OK;
    for (auto& slot : e->pin) {
        sc_status st = pin_slot_wait(slot);
        if (st && !first) first = st;
    }
    return first;
}

extern "C" sc_status sc_encoder_set_path(sc_encoder* e, int32_t path) {
    if (!e || path < 0 || path > 2) return sc_fail(SC_ERR_INVALID, "sc_encoder_set_path: path must be 0 (auto), 1 (batch pipeline) or 2 (small-batch pipeline)");
    std::lock_guard<std::mutex> g(e->mu);
    e->path = path;
    return SC_OK;
}

// Diagnostic: copy one workspace buffer of the last forward to the host (0 x, 1 y, 2 qkv, 3 ctx, 4 hm, 5 stat_a, 6 stat_b, 7 fin_a, 8 fin_b).
extern "C" sc_status sc_diag_encoder_read(sc_encoder* e, int32_t which, void* out, size_t nbytes) {
    if (!e || !out) return sc_fail(SC_ERR_INVALID, "sc_diag_encoder_read: NULL argument");
    std::lock_guard<std::mutex> g(e->mu);
    const void* src[9] = {e->x, e->y, e->qkv, e->ctx, e->hm, e->stat_a, e->stat_b, e->fin_a, e->fin_b};
    if (which < 0 || which > 8 || !src[which]) return sc_fail(SC_ERR_INVALID, "sc_diag_encoder_read: no such buffer");
    SC_HIP(hipSetDevice(e->rt->device));
    SC_HIP(hipStreamSynchronize(e->rt->stream));
    SC_HIP(hipMemcpy(out, src[which], nbytes, hipMemcpyDeviceToHost));
    return SC_OK;
}

extern "C" sc_status sc_encoder_info(sc_encoder* e, sc_encoder_cfg* cfg_out) {
    if (!e || !cfg_out) return sc_fail(SC_ERR_INVALID, "sc_encoder_info: NULL argument");
    *cfg_out = e->cfg;
    return SC_OK;
}

// ------------------------------------------------------------------ diagnostics (single-kernel parity tests)

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
};
// host f32 [n] -> device bf16 (through a device f32 staging buffer)
sc_status upload_bf16(const float* host, int64_t n, DevBuf& f32buf, DevBuf& out, hipStream_t s) {
    if (f32buf.alloc((size_t)n * 4) != hipSuccess || out.alloc((size_t)n * 2) != hipSuccess) return sc_fail(SC_ERR_NOMEM, "diag: hipMalloc failed");
    SC_HIP(hipMemcpyAsync(f32buf.p, host, (size_t)n * 4, hipMemcpyHostToDevice, s));
    sc_launch_f32_to_bf16((const float*)f32buf.p, out.p, n, s);
    return SC_OK;
}
}  // namespace

extern "C" sc_status sc_diag_gemm_bf16(sc_runtime* rt, int32_t epi, const float* A, const float* W, const float* bias, const float* R,
                                       int32_t M, int32_t N, int32_t K, float* out) {
    if (!rt || !A || !W || !bias || !out) return sc_fail(SC_ERR_INVALID, "sc_diag_gemm_bf16: NULL argument");
    const bool allow_splitk = epi >= 0 && (epi & 16);
    if (epi >= 0) epi &= 15;
    if (epi < 0 || epi > 2 || (epi == EPI_BIAS_RES && !R)) return sc_fail(SC_ERR_INVALID, "sc_diag_gemm_bf16: bad epilogue / missing residual");
    if (!sc_gemm_bf16_supported(M, N, K)) return sc_fail(SC_ERR_UNSUPPORTED, "sc_diag_gemm_bf16: need M%%128==0, N%%128==0, K%%64==0");
    SC_HIP(hipSetDevice(rt->device));
    hipStream_t s = rt->stream;
    DevBuf fa, fw, fr, da, dw, dr, db, dc, fo, sk;
    size_t sk_bytes = 0;
    if (allow_splitk) {
        sk_bytes = (size_t)sc_gemm_splitk_factor(M, N, K, rt->cus) * M * N * 4;
        if (sk.alloc(sk_bytes) != hipSuccess) return sc_fail(SC_ERR_NOMEM, "diag: hipMalloc failed");
    }
    sc_status st = upload_bf16(A, (int64_t)M * K, fa, da, s);
    if (st) return st;
    st = upload_bf16(W, (int64_t)N * K, fw, dw, s);
    if (st) return st;
    if (R) {
        st = upload_bf16(R, (int64_t)M * N, fr, dr, s);
        if (st) return st;
    }
    if (db.alloc((size_t)N * 4) != hipSuccess || dc.alloc((size_t)M * N * 2) != hipSuccess || fo.alloc((size_t)M * N * 4) != hipSuccess)
        return sc_fail(SC_ERR_NOMEM, "diag: hipMalloc failed");
    SC_HIP(hipMemcpyAsync(db.p, bias, (size_t)N * 4, hipMemcpyHostToDevice, s));
    sc_launch_gemm_bf16(epi, da.p, K, dw.p, K, (const float*)db.p, dr.p, N, dc.p, N, M, N, K, s, allow_splitk ? sk.p : nullptr, sk_bytes);
    sc_launch_bf16_to_f32(dc.p, (float*)fo.p, (int64_t)M * N, s);
    SC_HIP(hipGetLastError());
    SC_HIP(hipMemcpyAsync(out, fo.p, (size_t)M * N * 4, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    return SC_OK;
}

void sc_launch_gemm_i8_diag(const void* A, const void* W, void* C, int M, int N, int K, hipStream_t s);

void sc_scan_set_coarse_workgroups(int v);
void sc_scan_set_coarse_persistent(int v);
void sc_ivf_set_refresh_nomem(int v);
void sc_ivf_set_refine_cap(int v);
void sc_ivf_set_coarse_nomem(int v);
void sc_set_collect_pass(int v);
void sc_set_tighten(int v);
void sc_set_wide_force(int v);
void sc_set_ivf_tail_rows(int v);
extern "C" sc_status sc_diag_set_option(const char* name, int32_t value) {
    if (!name) return sc_fail(SC_ERR_INVALID, "sc_diag_set_option: NULL name");
    if (!strcmp(name, "coarse_workgroups")) sc_scan_set_coarse_workgroups(value);
    else if (!strcmp(name, "coarse_persistent")) sc_scan_set_coarse_persistent(value);
    else if (!strcmp(name, "gemm_pp")) sc_gemm_set_pp(value);
    else if (!strcmp(name, "ivf_refresh_nomem")) sc_ivf_set_refresh_nomem(value);
    else if (!strcmp(name, "ivf_refine_cap")) sc_ivf_set_refine_cap(value);
    else if (!strcmp(name, "ivf_coarse_nomem")) sc_ivf_set_coarse_nomem(value);
    else if (!strcmp(name, "collect_pass")) sc_set_collect_pass(value);
    else if (!strcmp(name, "tighten")) sc_set_tighten(value);
    else if (!strcmp(name, "wide_candidates")) sc_set_wide_force(value);
    else if (!strcmp(name, "ivf_tail_rows")) sc_set_ivf_tail_rows(value);
    else return sc_fail(SC_ERR_INVALID, "sc_diag_set_option: unknown option '%s'", name);
    return SC_OK;
}

extern "C" sc_status sc_diag_gemm_i8(sc_runtime* rt, const int8_t* A, const int8_t* W, int32_t M, int32_t N, int32_t K, int32_t* out) {
    if (!rt || !A || !W || !out) return sc_fail(SC_ERR_INVALID, "sc_diag_gemm_i8: NULL argument");
    if (M <= 0 || N <= 0 || K <= 0 || (M % 256) || (N % 256) || (K % 128)) return sc_fail(SC_ERR_UNSUPPORTED, "sc_diag_gemm_i8: need M%%256==0, N%%256==0, K%%128==0");
    SC_HIP(hipSetDevice(rt->device));
    hipStream_t s = rt->stream;
    DevBuf da, dw, dc;
    if (da.alloc((size_t)M * K) != hipSuccess || dw.alloc((size_t)N * K) != hipSuccess || dc.alloc((size_t)M * N * 4) != hipSuccess)
        return sc_fail(SC_ERR_NOMEM, "diag: hipMalloc failed");
    SC_HIP(hipMemcpyAsync(da.p, A, (size_t)M * K, hipMemcpyHostToDevice, s));
    SC_HIP(hipMemcpyAsync(dw.p, W, (size_t)N * K, hipMemcpyHostToDevice, s));
    sc_launch_gemm_i8_diag(da.p, dw.p, dc.p, M, N, K, s);
    SC_HIP(hipGetLastError());
    SC_HIP(hipMemcpyAsync(out, dc.p, (size_t)M * N * 4, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    return SC_OK;
}

extern "C" sc_status sc_diag_attention(sc_runtime* rt, const float* qkv, const int32_t* lens, int32_t B, int32_t S, int32_t heads, float* out) {
    if (!rt || !qkv || !lens || !out || B < 1) return sc_fail(SC_ERR_INVALID, "sc_diag_attention: bad argument");
    const int H = heads * 64;
    if (!sc_attention_supported(S, H, heads)) return sc_fail(SC_ERR_UNSUPPORTED, "sc_diag_attention: S must be one of 32,64,128,256,512,1024,2048");
    SC_HIP(hipSetDevice(rt->device));
    hipStream_t s = rt->stream;
    const int64_t tokens = (int64_t)B * S;
    DevBuf fq, dq, dl, dc, fo;
    sc_status st = upload_bf16(qkv, tokens * 3 * H, fq, dq, s);
    if (st) return st;
    if (dl.alloc((size_t)B * 4) != hipSuccess || dc.alloc((size_t)tokens * H * 2) != hipSuccess || fo.alloc((size_t)tokens * H * 4) != hipSuccess)
        return sc_fail(SC_ERR_NOMEM, "diag: hipMalloc failed");
    SC_HIP(hipMemcpyAsync(dl.p, lens, (size_t)B * 4, hipMemcpyHostToDevice, s));
    sc_launch_attention(dq.p, (const int32_t*)dl.p, B, S, H, nullptr, dc.p, s);
    sc_launch_bf16_to_f32(dc.p, (float*)fo.p, tokens * H, s);
    SC_HIP(hipGetLastError());
    SC_HIP(hipMemcpyAsync(out, fo.p, (size_t)tokens * H * 4, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    return SC_OK;
}

// Time `iters` launches of one GEMM shape on device-resident synthetic bf16 data (hipEvents on the
// runtime's stream).  variant: 0 = product kernel; 1/2/4/5 = diagnostic ablations of the 256-tile kernel
// (no in-loop LDS-DMA / no MFMA / no epilogue / no DMA + no epilogue); 128 = force the 128x128 tile.
extern "C" sc_status sc_diag_gemm_trace(sc_runtime* rt, int32_t epi, int32_t M, int32_t N, int32_t K, uint64_t* out, int64_t cap_words) {
    if (!rt || !out) return sc_fail(SC_ERR_INVALID, "sc_diag_gemm_trace: bad argument");
    if (M <= 0 || N <= 0 || K <= 0 || (M % 256) || (N % 256) || (K % 64)) return sc_fail(SC_ERR_UNSUPPORTED, "sc_diag_gemm_trace: need M%%256==0, N%%256==0, K%%64==0");
    const int64_t ntiles = (int64_t)(M / 256) * (N / 256);
    const int64_t launches = cap_words / (ntiles * 8);  // back-to-back traced launches, [launch][tile][8]
    if (launches < 1) return sc_fail(SC_ERR_INVALID, "sc_diag_gemm_trace: out too small (need 8 words per tile)");
    SC_HIP(hipSetDevice(rt->device));
    hipStream_t s = rt->stream;
    DevBuf fa, da, fw, dw, db, dr, dc, tr;
    const int64_t na = (int64_t)M * K, nw = (int64_t)N * K, nc = (int64_t)M * N;
    const size_t trace_bytes = (size_t)launches * ntiles * 64;
    if (fa.alloc(na * 4) != hipSuccess || da.alloc(na * 2) != hipSuccess || fw.alloc(nw * 4) != hipSuccess || dw.alloc(nw * 2) != hipSuccess ||
        db.alloc((size_t)N * 4) != hipSuccess || dr.alloc(nc * 2) != hipSuccess || dc.alloc(nc * 2) != hipSuccess || tr.alloc(trace_bytes) != hipSuccess)
        return sc_fail(SC_ERR_NOMEM, "diag: hipMalloc failed");
    sc_launch_synth_scaled((float*)fa.p, na, 1, 1.0f, 0.0f, s);
    sc_launch_synth_scaled((float*)fw.p, nw, 2, 0.05f, 0.0f, s);
    sc_launch_f32_to_bf16((const float*)fa.p, da.p, na, s);
    sc_launch_f32_to_bf16((const float*)fw.p, dw.p, nw, s);
    SC_HIP(hipMemsetAsync(db.p, 0, (size_t)N * 4, s));
    SC_HIP(hipMemsetAsync(dr.p, 0, (size_t)nc * 2, s));
    SC_HIP(hipMemsetAsync(tr.p, 0, trace_bytes, s));
    if (const char* e = getenv("SC_GEMM_TRACE_DBG")) sc_gemm_set_debug(atoi(e));  // stamps around an ablated main loop (scripts/gemm_clock.py)
    for (int i = 0; i < 2; ++i) sc_launch_gemm_bf16(epi, da.p, K, dw.p, K, (const float*)db.p, dr.p, N, dc.p, N, M, N, K, s);
    for (int64_t l = 0; l < launches; ++l) {
        sc_gemm_set_trace((unsigned long long*)tr.p + l * ntiles * 8);
        sc_launch_gemm_bf16(epi, da.p, K, dw.p, K, (const float*)db.p, dr.p, N, dc.p, N, M, N, K, s);
    }
    sc_gemm_set_trace(nullptr);
    sc_gemm_set_debug(0);
    hipError_t he = hipStreamSynchronize(s);
    if (he != hipSuccess) return sc_fail(SC_ERR_HIP, "diag gemm trace failed: %s", hipGetErrorString(he));
    SC_HIP(hipMemcpy(out, tr.p, trace_bytes, hipMemcpyDeviceToHost));
    return SC_OK;
}

extern "C" sc_status sc_diag_gemm_bench(sc_runtime* rt, int32_t epi, int32_t M, int32_t N, int32_t K, int32_t iters, int32_t variant,
                                        double* ms_per_launch) {
    if (!rt || !ms_per_launch || iters < 1) return sc_fail(SC_ERR_INVALID, "sc_diag_gemm_bench: bad argument");
    if (!sc_gemm_bf16_supported(M, N, K)) return sc_fail(SC_ERR_UNSUPPORTED, "sc_diag_gemm_bench: need M%%128==0, N%%128==0, K%%64==0");
    SC_HIP(hipSetDevice(rt->device));
    hipStream_t s = rt->stream;
    DevBuf fa, da, fw, dw, db, dr, dc;
    const int64_t na = (int64_t)M * K, nw = (int64_t)N * K, nc = (int64_t)M * N;
    if (fa.alloc(na * 4) != hipSuccess || da.alloc(na * 2) != hipSuccess || fw.alloc(nw * 4) != hipSuccess || dw.alloc(nw * 2) != hipSuccess ||
        db.alloc((size_t)N * 4) != hipSuccess || dr.alloc(nc * 2) != hipSuccess || dc.alloc(nc * 2) != hipSuccess)
        return sc_fail(SC_ERR_NOMEM, "diag: hipMalloc failed");
    sc_launch_synth_scaled((float*)fa.p, na, 1, 1.0f, 0.0f, s);
    sc_launch_synth_scaled((float*)fw.p, nw, 2, 0.05f, 0.0f, s);
    sc_launch_f32_to_bf16((const float*)fa.p, da.p, na, s);
    sc_launch_f32_to_bf16((const float*)fw.p, dw.p, nw, s);
    SC_HIP(hipMemsetAsync(db.p, 0, (size_t)N * 4, s));
    SC_HIP(hipMemsetAsync(dr.p, 0, (size_t)nc * 2, s));
    // variant = 100000 * pp + v: pp = main loop of the 256-tile kernel (0 = as configured, 1 = one barrier per K-tile, 2..5 = ping-pong depth)
    const int pp = variant / 100000;
    variant %= 100000;
    sc_gemm_set_pp(pp == 0 ? -1 : pp == 1 ? 0 : pp);
    sc_gemm_force_tile128(variant == 128);
    sc_gemm_set_debug((variant == 128 || variant >= 1000) ? 0 : variant);
    sc_gemm_set_order(variant >= 1000 ? variant - 1000 : 16);  // variants 1000+o: tile order o with the real epilogue
    hipEvent_t e0, e1;
    SC_HIP(hipEventCreate(&e0));
    SC_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) sc_launch_gemm_bf16(epi, da.p, K, dw.p, K, (const float*)db.p, dr.p, N, dc.p, N, M, N, K, s);
    hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) sc_launch_gemm_bf16(epi, da.p, K, dw.p, K, (const float*)db.p, dr.p, N, dc.p, N, M, N, K, s);
    hipEventRecord(e1, s);
    hipError_t he = hipStreamSynchronize(s);
    sc_gemm_force_tile128(false);
    sc_gemm_set_debug(0);
    sc_gemm_set_order(16);
    sc_gemm_set_pp(-1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (he != hipSuccess) return sc_fail(SC_ERR_HIP, "diag gemm bench failed: %s", hipGetErrorString(he));
    *ms_per_launch = ms / iters;
    return SC_OK;
}

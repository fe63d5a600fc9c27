// encoder_ops.hip -- the non-GEMM kernels of the transformer-encoder forward (gfx950).
//
// Replaces (reference): what llama.cpp's BERT graph does around the matmuls behind
// LlamaCppEmbeddings.embed_documents / embed_query (src/semcode/embeddings/providers.py:69-100;
// call sites src/semcode/services/indexer.py:150, src/semcode/rag/pipeline.py:171-175):
// token/position/type embedding + LayerNorm, post-attention / post-FFN LayerNorm, masked softmax
// attention, masked mean pooling.  Arithmetic follows BertModel (post-LN, eps inside sqrt, biased
// variance, erf-GELU, additive key mask), restated on CPU in oracle/bert_oracle.py.
//
//   embed_ln_kernel   HBM-bound  : 1 wave / token, f32 tables -> bf16 activations
//   layernorm_kernel  HBM-bound  : 1 wave / token, bf16 -> bf16 (the residual add is fused into the
//                                  producing GEMM's epilogue)
//   attention_kernel  MFMA + LDS : 1 workgroup / (chunk, head); K and V of the head stay in LDS,
//                                  S^T = K Q^T on v_mfma_f32_32x32x16_bf16 so a lane owns one query
//                                  column, softmax in registers, P feeds P V straight from the
//                                  accumulator registers, V fragments by ds_read_b64_tr_b16
//   mean_pool_kernel  HBM-bound  : 1 workgroup / chunk, masked mean (+ optional L2 normalise) -> f32
#include <cstdlib>

#include "gemm_tile.h"  // bf16 helpers, LDS-DMA pointer types

typedef short s16x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ float bf2f(bf16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
static __device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

#define LN_MAXJ 8  // hidden <= 2048

// ------------------------------------------------------------------ embeddings + LayerNorm
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t* __restrict__ ids, int tokens, int S, int H, int vocab, int max_pos,
                                                        const float* __restrict__ wemb, const float* __restrict__ pemb,
                                                        const float* __restrict__ temb, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= tokens) return;
    int id = ids[tok];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    int pos = tok % S;
    pos = pos >= max_pos ? max_pos - 1 : pos;
    const float* we = wemb + (size_t)id * H;
    const float* pe = pemb ? pemb + (size_t)pos * H : nullptr;  // NULL: no position table (ALiBi models)
    f32x4 v[LN_MAXJ];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
        const int k0 = 4 * lane + 256 * j;
        if (k0 < H) {
            v[j] = *reinterpret_cast<const f32x4*>(we + k0);
            if (pe) v[j] += *reinterpret_cast<const f32x4*>(pe + k0);
            v[j] += *reinterpret_cast<const f32x4*>(temb + k0);
            sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
    }
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
        const int k0 = 4 * lane + 256 * j;
        if (k0 < H) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d = v[j][c] - mean;
                sq = fmaf(d, d, sq);
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)H + eps);
    bf16_t* o = out + (size_t)tok * H;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
        const int k0 = 4 * lane + 256 * j;
        if (k0 < H) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + k0);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + k0);
            u16x4 r;
#pragma unroll
            for (int c = 0; c < 4; ++c) r[c] = f2bf((v[j][c] - mean) * rstd * g[c] + b[c]);
            *reinterpret_cast<u16x4*>(o + k0) = r;
        }
    }
}

// ------------------------------------------------------------------ LayerNorm-folded batch pipeline (sc_encoder.cpp, gemm_bf16.hip EPI_LNA_* / EPI_RESLN_STATS)
// Embeddings WITHOUT their LayerNorm: the raw sum (bf16) plus the row statistics the consuming GEMM folds in -- slot 0 of
// stats [slots][tokens_pad][2] gets (sum, sum of squares) of the bf16-ROUNDED row, the other slots zero (a GEMM-produced tensor
// fills one slot per 256 columns).
__global__ __launch_bounds__(256) void embed_raw_kernel(const int32_t* __restrict__ ids, int tokens, int tokens_pad, int S, int H, int vocab, int max_pos,
                                                         const float* __restrict__ wemb, const float* __restrict__ pemb,
                                                         const float* __restrict__ temb, bf16_t* __restrict__ out, float* __restrict__ stats, int slots) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= tokens_pad) return;
    float s1 = 0.f, s2 = 0.f;
    if (tok < tokens) {
        int id = ids[tok];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        int pos = tok % S;
        pos = pos >= max_pos ? max_pos - 1 : pos;
        const float* we = wemb + (size_t)id * H;
        const float* pe = pemb ? pemb + (size_t)pos * H : nullptr;
        bf16_t* o = out + (size_t)tok * H;
#pragma unroll
        for (int j = 0; j < LN_MAXJ; ++j) {
            const int k0 = 4 * lane + 256 * j;
            if (k0 < H) {
                f32x4 v = *reinterpret_cast<const f32x4*>(we + k0);
                if (pe) v += *reinterpret_cast<const f32x4*>(pe + k0);
                v += *reinterpret_cast<const f32x4*>(temb + k0);
                u16x4 r;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    r[c] = f2bf(v[c]);
                    const float y = bf2f(r[c]);
                    s1 += y;
                    s2 = fmaf(y, y, s2);
                }
                *reinterpret_cast<u16x4*>(o + k0) = r;
            }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
    } else {
        // padding rows (tokens .. tokens_pad, never read by attention or pooling): zeros with statistics (0, 0), so that whatever the
        // row-independent GEMMs compute for them stays finite (mu 0, rs 1/sqrt(eps), times zero)
        for (int k0 = 4 * lane; k0 < H; k0 += 256) *reinterpret_cast<u16x4*>(out + (size_t)tok * H + k0) = u16x4{0, 0, 0, 0};
    }
    if (lane < slots) {
        float* p = stats + ((size_t)lane * tokens_pad + tok) * 2;
        p[0] = lane == 0 ? s1 : 0.f;
        p[1] = lane == 0 ? s2 : 0.f;
    }
}

// W' = bf16(W diag(gamma)), c1[n] = sum_k W'[n,k] (of the ROUNDED values: it cancels what the MFMA accumulates),
// c2[n] = bias[n] + sum_k beta[k] W[n,k].  One wave per output row n.
__global__ __launch_bounds__(256) void fold_ln_weights_kernel(const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ bias, int N, int K, bf16_t* __restrict__ Wf,
                                                               float* __restrict__ c1, float* __restrict__ c2) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float s1 = 0.f, s2 = 0.f;
    for (int k0 = 4 * lane; k0 < K; k0 += 256) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + k0);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + k0);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + k0);
        u16x4 r;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            r[c] = f2bf(w[c] * g[c]);
            s1 += bf2f(r[c]);
            s2 = fmaf(b[c], w[c], s2);
        }
        *reinterpret_cast<u16x4*>(Wf + (size_t)n * K + k0) = r;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        c1[n] = s1;
        c2[n] = (bias ? bias[n] : 0.f) + s2;
    }
}
__global__ __launch_bounds__(256) void add_vectors_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// masked mean pooling of LayerNorm(y) computed on the fly from the raw rows and their partial statistics:
// mean_t LN(y_t) = gamma * (sum_t rs_t y_t - sum_t rs_t mu_t) / len + beta.  One workgroup per (chunk, 256 columns).
__global__ __launch_bounds__(256) void mean_pool_ln_kernel(const bf16_t* __restrict__ y, const float* __restrict__ stats, int slots, int tokens_pad,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            const int32_t* __restrict__ lens, int S, int H, float* __restrict__ out) {
    __shared__ float part[8][32][9];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int b = blockIdx.y, cc = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int k0 = blockIdx.x * 256 + cc * 8;
    int len = lens[b];
    len = len < 1 ? 1 : (len > S ? S : len);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float accmu = 0.f;
    const float inv_h = 1.0f / (float)H;
    if (k0 < H) {
        const bf16_t* p = y + (size_t)b * S * H + k0;
#pragma unroll 2
        for (int s0 = rg; s0 < len; s0 += 8) {
            const size_t tok = (size_t)b * S + s0;
            float s1 = 0.f, s2 = 0.f;
            for (int t = 0; t < slots; ++t) {
                s1 += stats[((size_t)t * tokens_pad + tok) * 2];
                s2 += stats[((size_t)t * tokens_pad + tok) * 2 + 1];
            }
            const float mu = s1 * inv_h;
            const float rs = 1.0f / sqrtf(fmaxf(s2 * inv_h - mu * mu, 0.f) + eps);
            const u32x4 raw = *reinterpret_cast<const u32x4*>(p + (size_t)s0 * H);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[2 * c] = fmaf(rs, __builtin_bit_cast(float, raw[c] << 16), acc[2 * c]);
                acc[2 * c + 1] = fmaf(rs, __builtin_bit_cast(float, raw[c] & 0xFFFF0000u), acc[2 * c + 1]);
            }
            accmu = fmaf(rs, mu, accmu);
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) part[rg][cc][c] = acc[c];
    part[rg][cc][8] = accmu;
    __syncthreads();
    if (rg == 0 && k0 < H) {
        const float inv = 1.0f / (float)len;
        float m = part[0][cc][8];
#pragma unroll
        for (int g = 1; g < 8; ++g) m += part[g][cc][8];
        float o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float t = part[0][cc][c];
#pragma unroll
            for (int g = 1; g < 8; ++g) t += part[g][cc][c];
            o[c] = fmaf((t - m) * inv, gamma[k0 + c], beta[k0 + c]);
        }
        *reinterpret_cast<f32x4*>(out + (size_t)b * H + k0) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(out + (size_t)b * H + k0 + 4) = f32x4{o[4], o[5], o[6], o[7]};
    }
}

// ------------------------------------------------------------------ LayerNorm (input already holds x + sublayer(x))
// HBM-bound (2 x tokens x H x 2 B).  Half a wave per token: lane l of the half owns the 16-byte chunks l, l + 32, ... of
// the row (8 bf16 each; H = 768 -> 3 per lane), so one load instruction moves 2 x 512 B.  Workgroups walk the rows with
// a grid stride and keep the NEXT row's loads in flight while the current one is reduced (two dependent half-wave
// reductions: mean, then the centred sum of squares, as BertModel computes it), gamma / beta stay in registers.
// The first version (one wave per row, 8-byte loads, one row per wave) ran at 3.65 TB/s.
template <int NC>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ in, int tokens, int H, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, bf16_t* __restrict__ out) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int l32 = threadIdx.x & 31, half = (threadIdx.x >> 5) & 1;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    f32x4 g[NC][2], b[NC][2];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int k0 = (l32 + 32 * j) * 8;
        if (k0 < H) {
            g[j][0] = *reinterpret_cast<const f32x4*>(gamma + k0);
            g[j][1] = *reinterpret_cast<const f32x4*>(gamma + k0 + 4);
            b[j][0] = *reinterpret_cast<const f32x4*>(beta + k0);
            b[j][1] = *reinterpret_cast<const f32x4*>(beta + k0 + 4);
        }
    }
    const float inv_h = 1.0f / (float)H;
    u32x4 nxt[NC];
    int tok = wave * 2 + half;
    if (tok < tokens) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int k0 = (l32 + 32 * j) * 8;
            if (k0 < H) nxt[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + (size_t)tok * H + k0));
        }
    }
    for (int base = wave * 2; base < tokens; base += nwaves * 2) {
        tok = base + half;
        const bool live = tok < tokens;
        u32x4 raw[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) raw[j] = nxt[j];
        const int ntok = tok + nwaves * 2;
        if (ntok < tokens) {
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int k0 = (l32 + 32 * j) * 8;
                if (k0 < H) nxt[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + (size_t)ntok * H + k0));
            }
        }
        float v[NC][8];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const bool on = live && (l32 + 32 * j) * 8 < H;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[j][2 * c] = on ? __builtin_bit_cast(float, raw[j][c] << 16) : 0.f;
                v[j][2 * c + 1] = on ? __builtin_bit_cast(float, raw[j][c] & 0xFFFF0000u) : 0.f;
            }
            sum += ((v[j][0] + v[j][1]) + (v[j][2] + v[j][3])) + ((v[j][4] + v[j][5]) + (v[j][6] + v[j][7]));
        }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float mean = sum * inv_h;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const bool on = (l32 + 32 * j) * 8 < H;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float d = on ? v[j][c] - mean : 0.f;
                sq = fmaf(d, d, sq);
            }
        }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
        const float rstd = 1.0f / sqrtf(sq * inv_h + eps);
        if (live) {
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int k0 = (l32 + 32 * j) * 8;
                if (k0 < H) {
                    u32x4 r;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float lo = (v[j][2 * c] - mean) * rstd * g[j][c >> 1][(2 * c) & 3] + b[j][c >> 1][(2 * c) & 3];
                        const float hi = (v[j][2 * c + 1] - mean) * rstd * g[j][c >> 1][(2 * c + 1) & 3] + b[j][c >> 1][(2 * c + 1) & 3];
                        r[c] = pack_bf16x2(lo, hi);
                    }
                    *reinterpret_cast<u32x4*>(out + (size_t)tok * H + k0) = r;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ attention (head dim 64)
// qkv: [tokens, 3H] bf16 rows = [Q | K | V], head h at columns h*64 of each third.  ctx: [tokens, H].
// KT = S / 32 key tiles.  LDS: K image [S][64] with chunk ^= (row>>1)&7, V image [S][64] with
// chunk ^= ((row>>1)&1)<<2 (conflict-free for ds_read_b128 rows / ds_read_b64_tr_b16 blocks).
// One 32-row query block of one wave, keys processed in groups of 4 tiles (128 keys) with an online softmax, so
// that only 64 score registers are live (S = 256 runs 2 workgroups per CU, S = 512 no longer spills).
// FULL = every key of the padded sequence is real (len == S): no per-tile guards or masks.
// ALIBI: scores get the symmetric linear bias -slope_head * |query - key| (jina-bert-v2 style encoders, no position table).
// The online-softmax state of one 32-row query block: running maximum, running denominator, unnormalised output.  A sequence
// longer than the 512 keys whose K and V fit the LDS is attended segment by segment (attention_long_kernel): the state is
// carried from one segment of keys to the next.
struct AttnState {
    float m_run, l_run;
    f32x16 o0, o1;
};
static __device__ __forceinline__ void attn_state_init(AttnState& a) {
    a.m_run = -3.0e38f;
    a.l_run = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a.o0[r] = 0.f; a.o1[r] = 0.f; }
}
// One segment of keys (the KT tiles in Kl / Vl = keys key0 .. key0 + 32 KT of the sequence; len / nkt count inside the segment)
// folded into the state.  first: the state is fresh (no rescale of O before the first group).
template <int KT, bool FULL, bool ALIBI, int GKMAX = 4>
static __device__ __forceinline__ void attention_qblock_core(const bf16x8 (&qf)[4], const char* Kl, const char* Vl, float* xch, int len, int nkt, int lane,
                                                             int qbase, float slope2, AttnState& S_, bool first, int key0) {
    constexpr int GK = KT < GKMAX ? KT : GKMAX;   // key tiles per group
    constexpr int NG = (KT + GK - 1) / GK;
    const int l31 = lane & 31, hh = lane >> 5;
    const float sl2 = 0.125f * 1.44269504088896340736f;  // 1/sqrt(64) * log2(e)
    const int tail = len & 31;                           // != 0: the last real key tile is partially masked
    float m_run = S_.m_run, l_run = S_.l_run;
    f32x16 o0 = S_.o0, o1 = S_.o1;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (!FULL && g * GK >= nkt) continue;  // wave-uniform: nothing real in this group
        // S^T tiles of the group: st[i][r] = score(key = 32 t + (r&3) + 8 (r>>2) + 4 hh, query = l31), t = g*GK + i
        f32x16 st[GK];
#pragma unroll
        for (int i = 0; i < GK; ++i) {
            const int t = g * GK + i;
            if (FULL || t < nkt) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const int krow = 32 * t + l31;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int c = 2 * ks + hh;
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kl + krow * 128 + ((c ^ ((krow >> 1) & 7)) << 4));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], acc, 0, 0, 0);
                }
                if (ALIBI) {  // work in the exp2 domain from here on: v = s * sl2 - slope2 * |q - key|
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float dist = (float)(key0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hh - (qbase + l31));
                        acc[r] = fmaf(acc[r], sl2, -slope2 * fabsf(dist));
                    }
                }
                st[i] = acc;
            }
        }
        const float sc2 = ALIBI ? 1.0f : sl2;  // scores already scaled when ALIBI
        float mx = m_run;
#pragma unroll
        for (int i = 0; i < GK; ++i) {
            const int t = g * GK + i;
            if (FULL || t < nkt) {
                if (!FULL && t == nkt - 1 && tail) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (32 * t + (r & 3) + 8 * (r >> 2) + 4 * hh < len) mx = fmaxf(mx, st[i][r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[i][r]);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mb = mx * sc2;
        const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run, sc2, -mb));  // first group: exp2(-huge) = 0, and O, l are 0 anyway
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < GK; ++i) {
            const int t = g * GK + i;
            if (FULL || t < nkt) {
                const bool masked = !FULL && (t == nkt - 1) && tail;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = __builtin_amdgcn_exp2f(fmaf(st[i][r], sc2, -mb));
                    if (masked && !(32 * t + (r & 3) + 8 * (r >> 2) + 4 * hh < len)) p = 0.f;
                    st[i][r] = p;
                    sum += p;
                }
            }
        }
        sum += __shfl_xor(sum, 32, 64);
        l_run = fmaf(l_run, alpha, sum);
        m_run = mx;
        if (g > 0 || !first) {  // rescale O: its rows are queries (r&3) + 8 (r>>2) + 4 hh, alpha lives on lane q -> exchange through LDS
            xch[l31] = alpha;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 al = *reinterpret_cast<const f32x4*>(xch + 8 * g4 + 4 * hh);
#pragma unroll
                for (int c = 0; c < 4; ++c) { o0[4 * g4 + c] *= al[c]; o1[4 * g4 + c] *= al[c]; }
            }
        }
        // O += P V: A operand = P straight from the score registers (k order of step s:
        // key = 32 t + 16 s + 8 (j>>2) + 4 hh + (j&3)), B operand = V by transposed LDS reads
#pragma unroll
        for (int i = 0; i < GK; ++i) {
            const int t = g * GK + i;
            if (FULL || t < nkt) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 pp;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pp[j] = pack_bf16x2(st[i][8 * s + 2 * j], st[i][8 * s + 2 * j + 1]);
                    const bf16x8 pf = __builtin_bit_cast(bf16x8, pp);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        bf16x8 vf;
#pragma unroll
                        for (int piece = 0; piece < 2; ++piece) {
                            const int row = 32 * t + 16 * s + 8 * piece + 4 * hh + ((lane & 15) >> 2);
                            const int d0 = 32 * dt + 16 * ((lane >> 4) & 1);
                            const int chunk = (d0 >> 3) + ((lane & 3) >> 1);
                            const int sw = chunk ^ (((row >> 1) & 1) << 2);
                            typedef __attribute__((address_space(3))) s16x4* lds_s16x4p;
                            const s16x4 got = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4p)(Vl + row * 128 + sw * 16 + 8 * (lane & 1)));
                            vf[4 * piece + 0] = got[0];
                            vf[4 * piece + 1] = got[1];
                            vf[4 * piece + 2] = got[2];
                            vf[4 * piece + 3] = got[3];
                        }
                        if (dt == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, vf, o0, 0, 0, 0);
                        else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, vf, o1, 0, 0, 0);
                    }
                }
            }
        }
    }
    S_.m_run = m_run;
    S_.l_run = l_run;
    S_.o0 = o0;
    S_.o1 = o1;
}
// o[r] = O[q = (r&3) + 8 (r>>2) + 4 hh][d = 32 dt + l31]: normalise by 1/l[q], stage as bf16 [q][d] in LDS 8 query rows
// at a time (1 KiB, wave-private; LDS runs a wave's instructions in order), then leave as whole 128-byte rows.
static __device__ __forceinline__ void attn_state_store(const AttnState& S_, float* xch, char* ostg, bf16_t* obase, int H, int lane) {
    const int l31 = lane & 31, hh = lane >> 5;
    const float l_run = S_.l_run;
    const f32x16 o0 = S_.o0, o1 = S_.o1;
    xch[l31] = 1.0f / l_run;
    bf16_t* og = reinterpret_cast<bf16_t*>(ostg);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 il = *reinterpret_cast<const f32x4*>(xch + 8 * g4 + 4 * hh);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ql = 4 * hh + c;  // row inside this block of 8
            og[ql * 64 + l31] = (bf16_t)(pack_bf16x2(o0[4 * g4 + c] * il[c], 0.f) & 0xFFFFu);
            og[ql * 64 + 32 + l31] = (bf16_t)(pack_bf16x2(o1[4 * g4 + c] * il[c], 0.f) & 0xFFFFu);
        }
        const int ql = lane >> 3, c8 = lane & 7;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(ostg + ql * 128 + c8 * 16);
        *reinterpret_cast<bf16x8*>(obase + (size_t)(8 * g4 + ql) * H + c8 * 8) = v;
    }
}

template <int KT, bool FULL, bool ALIBI, int GKMAX = 4>
static __device__ __forceinline__ void attention_qblock(const bf16x8 (&qf)[4], const char* Kl, const char* Vl, float* xch, char* ostg,
                                                        bf16_t* obase, int H, int len, int nkt, int lane, int qbase, float slope2) {
    AttnState st;
    attn_state_init(st);
    attention_qblock_core<KT, FULL, ALIBI, GKMAX>(qf, Kl, Vl, xch, len, nkt, lane, qbase, slope2, st, true, 0);
    attn_state_store(st, xch, ostg, obase, H, lane);
}

// NW waves per workgroup (4, or 8 for S = 256: one query block per wave, two key-tile groups of 2 -> half the registers, so that
// 4 waves instead of 2 share a SIMD and cover each other's MFMA -> softmax -> MFMA dependency stalls)
template <int KT, bool ALIBI, int NW = 4>
__global__ __launch_bounds__(NW * 64, (KT <= 8 ? 2 : 1)) void attention_kernel(const bf16_t* __restrict__ qkv, const int32_t* __restrict__ lens, int H,
                                                            const float* __restrict__ slopes, bf16_t* __restrict__ ctx, int blocked) {
    constexpr int S = KT * 32;
    constexpr int NQB = (KT + NW - 1) / NW;  // query blocks per wave
    constexpr int GKMAX = (NW == 8 && KT <= 8) ? 2 : 4;  // S = 512 runs one workgroup per CU anyway (LDS): registers are not its limit
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kl = smem;
    char* Vl = smem + S * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* xch = reinterpret_cast<float*>(smem + 2 * S * 128) + w * 32;  // [NW waves][32] alpha / 1/l exchange
    char* ostg = smem + 2 * S * 128 + NW * 128 + w * 1024;               // per-wave [8 q][64 d] bf16 output staging
    const int head = blockIdx.x, b = blockIdx.y;
    // row-major QKV [tokens][3H] (Q | K | V, head h at column 64 h), or -- blocked -- [3 heads][tokens][64] as the QKV projection
    // writes it (gemm_bf16.hip c_index; `blocked` = tokens padded to that GEMM's M, 0 = row-major): every operand of this (chunk, head) is then one contiguous [S][64] block
    const int ld = blocked ? 64 : 3 * H;
    const size_t T = (size_t)blocked, nh = (size_t)(H >> 6);  // blocked = the row count (M) of the projection that wrote the blocks
    const bf16_t* base = blocked ? qkv + ((size_t)head * T + (size_t)b * S) * 64 : qkv + (size_t)b * S * ld + head * 64;
    const size_t koff = blocked ? nh * T * 64 : (size_t)H, voff = 2 * koff;
    int len = lens[b];
    len = len < 1 ? 1 : (len > S ? S : len);
    len = __builtin_amdgcn_readfirstlane(len);
    const int nkt = (len + 31) >> 5;  // key tiles holding at least one real key; later tiles are skipped entirely

    // ---- stage K then V rows [0, 32 nkt) of this (chunk, head): pieces of 8 rows x 128 B = 1 KiB
    for (int piece = w; piece < nkt * 4; piece += NW) {
        const int p = piece * 64 + lane;
        const int r = p >> 3, ck = (p & 7) ^ ((r >> 1) & 7);
        __builtin_amdgcn_global_load_lds((gbl_vptr)(base + (size_t)r * ld + koff + ck * 8), (lds_vptr)(Kl + piece * 1024), 16, 0, 0);
    }
    for (int piece = w; piece < nkt * 4; piece += NW) {
        const int p = piece * 64 + lane;
        const int r = p >> 3, cv = (p & 7) ^ (((r >> 1) & 1) << 2);
        __builtin_amdgcn_global_load_lds((gbl_vptr)(base + (size_t)r * ld + voff + cv * 8), (lds_vptr)(Vl + piece * 1024), 16, 0, 0);
    }
    // Q fragments of all of this wave's query blocks (B operand: lane (q = lane&31, hh) holds Q[q][16 ks + 8 hh .. +7]),
    // requested now so that their latency hides behind the K/V staging
    const int l31 = lane & 31, hh = lane >> 5;
    bf16x8 qf[NQB][4];
#pragma unroll
    for (int i = 0; i < NQB; ++i) {
        const int qb = w + NW * i;
        if (qb < KT) {
            const bf16_t* qrow = base + (size_t)(qb * 32 + l31) * ld;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[i][ks] = *reinterpret_cast<const bf16x8*>(qrow + 16 * ks + 8 * hh);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll
    for (int i = 0; i < NQB; ++i) {
        const int qb = w + NW * i;
        if (qb < KT) {
            bf16_t* obase = ctx + (size_t)(b * S + qb * 32) * H + head * 64;
            const float slope2 = ALIBI ? slopes[head] * 1.44269504088896340736f : 0.f;
            if (nkt == KT && (len & 31) == 0) attention_qblock<KT, true, ALIBI, GKMAX>(qf[i], Kl, Vl, xch, ostg, obase, H, len, nkt, lane, qb * 32, slope2);
            else attention_qblock<KT, false, ALIBI, GKMAX>(qf[i], Kl, Vl, xch, ostg, obase, H, len, nkt, lane, qb * 32, slope2);
        }
    }
}

// ---- sequences of 1 024 and 2 048 tokens (the reference's default chunker emits up to 200 lines / 6 000 characters per chunk --
// src/semcode/chunking/tree_sitter_chunker.py:64-65 -- i.e. 1.5-2k word pieces; jina-embeddings-v2 is an 8k-context ALiBi
// model).  K and V of 512 keys fill the LDS (128 KiB), so the keys are attended in SEGMENTS of 512: stage a segment, every wave
// folds it into the online-softmax state of its ONE 32-row query block (AttnState, 34 registers), next segment.  A workgroup
// takes 8 query blocks (256 queries) of one (chunk, head); the S / 256 workgroups of a (chunk, head) each stage all its keys
// (from L2 after the first).  Same arithmetic per (query, key) as attention_kernel: a sequence of <= 512 real tokens padded to
// 1 024 gives the bits the S = 512 kernel gives.
template <bool ALIBI>
__global__ __launch_bounds__(512, 1) void attention_long_kernel(const bf16_t* __restrict__ qkv, const int32_t* __restrict__ lens, int S, int H,
                                                                const float* __restrict__ slopes, bf16_t* __restrict__ ctx, int blocked) {
    constexpr int KT = 16, NW = 8, SEG = 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kl = smem;
    char* Vl = smem + SEG * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* xch = reinterpret_cast<float*>(smem + 2 * SEG * 128) + w * 32;
    char* ostg = smem + 2 * SEG * 128 + NW * 128 + w * 1024;
    const int head = blockIdx.x, b = blockIdx.y, qb = (int)blockIdx.z * NW + w;  // this wave's 32-row query block
    const int ld = blocked ? 64 : 3 * H;
    const size_t T = (size_t)blocked, nh = (size_t)(H >> 6);
    const bf16_t* base = blocked ? qkv + ((size_t)head * T + (size_t)b * S) * 64 : qkv + (size_t)b * S * ld + head * 64;
    const size_t koff = blocked ? nh * T * 64 : (size_t)H, voff = 2 * koff;
    int len = lens[b];
    len = len < 1 ? 1 : (len > S ? S : len);
    len = __builtin_amdgcn_readfirstlane(len);
    const int l31 = lane & 31, hh = lane >> 5;
    bf16x8 qf[4];
    {
        const bf16_t* qrow = base + (size_t)(qb * 32 + l31) * ld;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + 16 * ks + 8 * hh);
    }
    AttnState st;
    attn_state_init(st);
    const float slope2 = ALIBI ? slopes[head] * 1.44269504088896340736f : 0.f;
    const int nseg = (len + SEG - 1) / SEG;
#pragma unroll 1
    for (int seg = 0; seg < nseg; ++seg) {
        const int key0 = seg * SEG;
        const int slen = len - key0 < SEG ? len - key0 : SEG;  // real keys in this segment
        const int nkt = (slen + 31) >> 5;
        if (seg) __syncthreads();  // every wave is done with the previous segment's K / V
        for (int piece = w; piece < nkt * 4; piece += NW) {
            const int p = piece * 64 + lane;
            const int r = p >> 3, ck = (p & 7) ^ ((r >> 1) & 7);
            __builtin_amdgcn_global_load_lds((gbl_vptr)(base + (size_t)(key0 + r) * ld + koff + ck * 8), (lds_vptr)(Kl + piece * 1024), 16, 0, 0);
        }
        for (int piece = w; piece < nkt * 4; piece += NW) {
            const int p = piece * 64 + lane;
            const int r = p >> 3, cv = (p & 7) ^ (((r >> 1) & 1) << 2);
            __builtin_amdgcn_global_load_lds((gbl_vptr)(base + (size_t)(key0 + r) * ld + voff + cv * 8), (lds_vptr)(Vl + piece * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nkt == KT && (slen & 31) == 0) attention_qblock_core<KT, true, ALIBI, 4>(qf, Kl, Vl, xch, slen, nkt, lane, qb * 32, slope2, st, seg == 0, key0);
        else attention_qblock_core<KT, false, ALIBI, 4>(qf, Kl, Vl, xch, slen, nkt, lane, qb * 32, slope2, st, seg == 0, key0);
    }
    attn_state_store(st, xch, ostg, ctx + (size_t)(b * S + qb * 32) * H + head * 64, H, lane);
}

// ------------------------------------------------------------------ masked mean pooling
__global__ __launch_bounds__(256) void mean_pool_kernel(const bf16_t* __restrict__ x, const int32_t* __restrict__ lens, int S, int H,
                                                         int normalize, float* __restrict__ out) {
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    int len = lens[b];
    len = len < 1 ? 1 : (len > S ? S : len);
    const float inv = 1.0f / (float)len;
    float ss = 0.f;
    for (int k0 = 4 * tid; k0 < H; k0 += 1024) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bf16_t* p = x + (size_t)b * S * H + k0;
        for (int s = 0; s < len; ++s) {
            const u16x4 raw = *reinterpret_cast<const u16x4*>(p + (size_t)s * H);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += bf2f(raw[c]);
        }
        acc *= inv;
        *reinterpret_cast<f32x4*>(out + (size_t)b * H + k0) = acc;
        ss += (acc[0] * acc[0] + acc[1] * acc[1]) + (acc[2] * acc[2] + acc[3] * acc[3]);
    }
    if (normalize) {
        red[tid] = ss;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        const float scale = 1.0f / fmaxf(sqrtf(red[0]), 1e-12f);
        for (int k0 = 4 * tid; k0 < H; k0 += 1024) {
            f32x4 v = *reinterpret_cast<f32x4*>(out + (size_t)b * H + k0);
            v *= scale;
            *reinterpret_cast<f32x4*>(out + (size_t)b * H + k0) = v;
        }
    }
}

// ------------------------------------------------------------------ f32 -> bf16 weight conversion, fills
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = f2bf(in[i]);
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = bf2f(in[i]);
}
__global__ __launch_bounds__(256) void synth_scaled_kernel(float* __restrict__ out, int64_t n, uint64_t key, float scale, float offset) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = offset + scale * sc_synth_value(key, (uint64_t)i, 0u, 1u);
}

// ------------------------------------------------------------------ launchers
void sc_launch_embed_ln(const int32_t* ids, int tokens, int S, int H, int vocab, int max_pos, const float* wemb, const float* pemb,
                        const float* temb, const float* g, const float* b, float eps, void* out, hipStream_t s) {
    hipLaunchKernelGGL(embed_ln_kernel, dim3((unsigned)((tokens + 3) / 4)), dim3(256), 0, s, ids, tokens, S, H, vocab, max_pos, wemb, pemb, temb,
                       g, b, eps, (bf16_t*)out);
}
void sc_launch_layernorm(const void* in, int tokens, int H, const float* g, const float* b, float eps, void* out, hipStream_t s) {
    int blocks = (tokens + 7) / 8;  // 8 rows per workgroup per sweep
    if (blocks > 256 * 8) blocks = 256 * 8;
    const dim3 grid((unsigned)blocks), block(256);
    if (H <= 768) hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, s, (const bf16_t*)in, tokens, H, g, b, eps, (bf16_t*)out);
    else if (H <= 1024) hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, (const bf16_t*)in, tokens, H, g, b, eps, (bf16_t*)out);
    else hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, (const bf16_t*)in, tokens, H, g, b, eps, (bf16_t*)out);
}
template <int KT, bool ALIBI>
static void launch_attn(const void* qkv, const int32_t* lens, int B, int H, const float* slopes, void* ctx, int blocked, hipStream_t s) {
    static const char* env = getenv("SC_ATTN_WAVES");  // A/B aid
    constexpr int NWD = KT >= 8 ? 8 : 4;
    const int nw = (env && KT >= 8) ? atoi(env) : NWD;
    static ScDeviceOnce once;  // per instantiation and device
    sc_device_once(once, [&] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<KT, ALIBI, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, KT * 32 * 256 + 4 * 1152);
        if (KT >= 8) hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<KT, ALIBI, NWD>), hipFuncAttributeMaxDynamicSharedMemorySize, KT * 32 * 256 + NWD * 1152);
    });
    const dim3 grid((unsigned)(H / 64), (unsigned)B);
    if (KT >= 8 && nw == 8)
        hipLaunchKernelGGL((attention_kernel<KT, ALIBI, NWD>), grid, dim3(NWD * 64), (size_t)KT * 32 * 256 + NWD * 1152, s, (const bf16_t*)qkv, lens, H, slopes, (bf16_t*)ctx, blocked);
    else
        hipLaunchKernelGGL((attention_kernel<KT, ALIBI, 4>), grid, dim3(256), (size_t)KT * 32 * 256 + 4 * 1152, s, (const bf16_t*)qkv, lens, H, slopes, (bf16_t*)ctx, blocked);
}
bool sc_attention_supported(int S, int H, int heads) {
    return heads > 0 && H == heads * 64 && (S == 32 || S == 64 || S == 128 || S == 256 || S == 512 || S == 1024 || S == 2048);
}
// slopes: NULL = plain attention; else [heads] ALiBi slopes (device)
void sc_launch_attention(const void* qkv, const int32_t* lens, int B, int S, int H, const float* slopes, void* ctx, hipStream_t s, int blocked) {
#define SC_ATTN_CASE(SS, KK)                                                      \
    case SS:                                                                      \
        if (slopes) launch_attn<KK, true>(qkv, lens, B, H, slopes, ctx, blocked, s); \
        else launch_attn<KK, false>(qkv, lens, B, H, nullptr, ctx, blocked, s);   \
        break;
    if (S > 512) {  // K / V streamed through the LDS in segments of 512 keys
        static ScDeviceOnce once;
        const int lds = 2 * 512 * 128 + 8 * 128 + 8 * 1024;
        sc_device_once(once, [&] {
            hipFuncSetAttribute(reinterpret_cast<const void*>(attention_long_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipFuncSetAttribute(reinterpret_cast<const void*>(attention_long_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        });
        const dim3 grid((unsigned)(H / 64), (unsigned)B, (unsigned)(S / 256));
        if (slopes) hipLaunchKernelGGL(attention_long_kernel<true>, grid, dim3(512), (size_t)lds, s, (const bf16_t*)qkv, lens, S, H, slopes, (bf16_t*)ctx, blocked);
        else hipLaunchKernelGGL(attention_long_kernel<false>, grid, dim3(512), (size_t)lds, s, (const bf16_t*)qkv, lens, S, H, (const float*)nullptr, (bf16_t*)ctx, blocked);
        return;
    }
    switch (S) {
        SC_ATTN_CASE(32, 1)
        SC_ATTN_CASE(64, 2)
        SC_ATTN_CASE(128, 4)
        SC_ATTN_CASE(256, 8)
        SC_ATTN_CASE(512, 16)
        default: break;
    }
#undef SC_ATTN_CASE
}
// GEGLU (jina-bert-v2 GLUMLP): out[m][j] = gelu(h[m][j]) * h[m][F + j], h = [tokens, 2F] bf16 -> out [tokens, F] bf16
__global__ __launch_bounds__(256) void geglu_kernel(const bf16_t* __restrict__ h, int64_t tokens, int F, bf16_t* __restrict__ out) {
    const int64_t total = tokens * (int64_t)(F / 8);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / (F / 8);
        const int j = (int)(i - m * (F / 8)) * 8;
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(h + m * 2 * F + j);
        const bf16x8 u = *reinterpret_cast<const bf16x8*>(h + m * 2 * F + F + j);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x2 gg = gelu_erf_fast2(f32x2{bf2f((bf16_t)g[2 * c]), bf2f((bf16_t)g[2 * c + 1])});
            o[c] = pack_bf16x2(gg[0] * bf2f((bf16_t)u[2 * c]), gg[1] * bf2f((bf16_t)u[2 * c + 1]));
        }
        *reinterpret_cast<u32x4*>(out + m * F + j) = o;
    }
}
void sc_launch_geglu(const void* h, int64_t tokens, int F, void* out, hipStream_t s) {
    int64_t blocks = (tokens * (F / 8) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)h, tokens, F, (bf16_t*)out);
}
// Unnormalised pooling with more parallelism than one workgroup per chunk (which ran at 1.3 TB/s): a workgroup owns 256
// columns of one chunk, thread (cc, rg) sums the rows rg, rg + 8, ... of 8 columns with 16-byte loads, LDS combines the 8 row groups
// in a fixed order.
__global__ __launch_bounds__(256) void mean_pool_sliced_kernel(const bf16_t* __restrict__ x, const int32_t* __restrict__ lens, int S, int H,
                                                                float* __restrict__ out) {
    __shared__ float part[8][32][8];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int b = blockIdx.y, cc = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int k0 = blockIdx.x * 256 + cc * 8;
    int len = lens[b];
    len = len < 1 ? 1 : (len > S ? S : len);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (k0 < H) {
        const bf16_t* p = x + (size_t)b * S * H + k0;
#pragma unroll 4
        for (int s0 = rg; s0 < len; s0 += 8) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(p + (size_t)s0 * H);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[2 * c] += __builtin_bit_cast(float, raw[c] << 16);
                acc[2 * c + 1] += __builtin_bit_cast(float, raw[c] & 0xFFFF0000u);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) part[rg][cc][c] = acc[c];
    __syncthreads();
    if (rg == 0 && k0 < H) {
        const float inv = 1.0f / (float)len;
        f32x4 lo, hi;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float t = part[0][cc][c];
#pragma unroll
            for (int g = 1; g < 8; ++g) t += part[g][cc][c];
            if (c < 4) lo[c] = t * inv;
            else hi[c - 4] = t * inv;
        }
        *reinterpret_cast<f32x4*>(out + (size_t)b * H + k0) = lo;
        *reinterpret_cast<f32x4*>(out + (size_t)b * H + k0 + 4) = hi;
    }
}

void sc_launch_mean_pool(const void* x, const int32_t* lens, int B, int S, int H, int normalize, float* out, hipStream_t s) {
    if (!normalize && (H % 8) == 0)
        hipLaunchKernelGGL(mean_pool_sliced_kernel, dim3((unsigned)((H + 255) / 256), (unsigned)B), dim3(256), 0, s, (const bf16_t*)x, lens, S, H, out);
    else
        hipLaunchKernelGGL(mean_pool_kernel, dim3((unsigned)B), dim3(256), 0, s, (const bf16_t*)x, lens, S, H, normalize, out);
}
void sc_launch_embed_raw(const int32_t* ids, int tokens, int tokens_pad, int S, int H, int vocab, int max_pos, const float* wemb, const float* pemb,
                         const float* temb, void* out, float* stats, int slots, hipStream_t s) {
    hipLaunchKernelGGL(embed_raw_kernel, dim3((unsigned)((tokens_pad + 3) / 4)), dim3(256), 0, s, ids, tokens, tokens_pad, S, H, vocab, max_pos, wemb, pemb, temb,
                       (bf16_t*)out, stats, slots);
}
void sc_launch_fold_ln_weights(const float* W, const float* gamma, const float* beta, const float* bias, int N, int K, void* Wf, float* c1, float* c2,
                               hipStream_t s) {
    hipLaunchKernelGGL(fold_ln_weights_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, W, gamma, beta, bias, N, K, (bf16_t*)Wf, c1, c2);
}
void sc_launch_add_vectors(const float* a, const float* b, float* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(add_vectors_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, out, n);
}
void sc_launch_mean_pool_ln(const void* y, const float* stats, int slots, int tokens_pad, const float* gamma, const float* beta, float eps,
                            const int32_t* lens, int B, int S, int H, float* out, hipStream_t s) {
    hipLaunchKernelGGL(mean_pool_ln_kernel, dim3((unsigned)((H + 255) / 256), (unsigned)B), dim3(256), 0, s, (const bf16_t*)y, stats, slots, tokens_pad, gamma,
                       beta, eps, lens, S, H, out);
}
void sc_launch_f32_to_bf16(const float* in, void* out, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, (bf16_t*)out, n);
}
void sc_launch_bf16_to_f32(const void* in, float* out, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)in, out, n);
}
void sc_launch_synth_scaled(float* out, int64_t n, uint64_t seed, float scale, float offset, hipStream_t s) {
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(synth_scaled_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, n, sc_synth_key(seed), scale, offset);
}

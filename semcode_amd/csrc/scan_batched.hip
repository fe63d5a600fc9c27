// scan_batched.hip -- large query batches: bf16 MFMA coarse scores fused with threshold filtering, then an
// exact f32 re-rank with a per-query certificate (gfx950).
//
// Replaces (reference): the server side of Collection.search for a FLAT scan
// (src/semcode/storage/milvus_store.py:141-147), for batches the reference never sends (it issues one
// query per call); BASELINE.json's configs[2] (batch-1024 over 10M x 768) is this path.
//
// Why two stages: at Q = 1024 the scan is 2*N*d*Q = 1.6e13 FLOP per batch -- 101 ms on the f32 matrix
// pipe but 6.3 ms at the bf16 MFMA peak, against 4.9 ms to stream the f32 corpus.
// So the corpus keeps a bf16 shadow copy in HBM and the batch runs as a bf16 GEMM (gemm_tile.h) whose
// epilogue never stores scores: it compares each of them with a per-query threshold and appends the
// rare survivors (64-bit key = coarse score | row) to a per-query list.  Thresholds tighten between
// "phases" of geometrically growing row ranges (1 Ki, 4 Ki, 16 Ki, ... rows, x4 each): after each phase a
// small kernel keeps the k' best coarse keys per query and publishes the new threshold.
//
// Exactness: the k' (= 128) coarse candidates of a query are re-scored in f32 in the canonical
// summation order of scan_exact.hip / oracle/sc_oracle.c, so every returned distance is bit-identical
// to the exact path.  |coarse - exact| <= eps_q (bf16 input rounding, bound below), therefore a row that
// is NOT a candidate has exact score >= tau_q - eps_q; if the k-th exact candidate score is strictly
// below that, the top-k is proven complete and correctly ordered.  Queries that fail the test (or whose
// survivor list overflowed) are flagged and re-run through the exact scan by the host code.
//
// Roofline: MFMA bf16; algorithmic FLOPs = 2 * rows * ld * Qpad per phase launch.
#include <map>
#include <vector>
#include <algorithm>
#include <cstdio>
#include <algorithm>
#include <cstdlib>

#include "gemm_tile.h"

#define KPRIME 128         // coarse candidates kept per query (one re-rank thread each)
#define KPRIME8 512        // the same for the int8 coarse stage, whose error bound is ~7x wider (see below); with 256 the certificate
                           // fails for 51 of 1 024 queries at 10M x 768 and the bf16 stage they go to costs more than is saved: 84k -> 63k QPS
#define SEL_THREADS 256

// ---- int8 coarse stage -------------------------------------------------------------------------------------------------
// At Q = 1024 the bf16 GEMM is the whole cost (15 ms at 1.0 PF; 6.3 ms even at the nominal peak).  v_mfma_i32_16x16x64_i8 runs
// at twice the bf16 rate, so the coarse scores can also be taken from an int8 shadow: row r is stored as q_r = round(x_r / s_r),
// s_r = max|x_r| / 127 (7.7 GB at 10M x 768), the query likewise, and <x,q> ~ s_r s_q <q_r,q_q> with an EXACT integer dot.
// Nothing else changes: survivors below a per-query threshold, the KPRIME8 best kept between phases, every candidate re-scored in
// f32 in the canonical order (returned distances stay bit-identical to the exact path), and the same certificate
//     kth exact score + eps_q < tau_q      with eps_q from Cauchy-Schwarz on the ACTUAL rounding residuals
//     |<x,q> - s_r s_q <q_r,q_q>| <= max_r|x_r - s_r q_r| |q| + max_r|s_r q_r| |q - s_q q_q| .
// int8 residuals are ~7x those of bf16 (step max|x|/127 vs 2^-9 relative), so tau_q has to sit further out: 512 candidates
// instead of 128 (on N(0,1) data at 768 dimensions eps ~ 24 in squared-L2 units against a gap of ~48 between the 10th and the
// 512th neighbour).  A query that fails the int8 certificate is re-run on the bf16 stage, and only then exactly.

// ------------------------------------------------------------------ bf16 shadow + max norm + max rounding residual
// One wave per row: Xb = bf16(X); res_bits = max over rows of |x - bf16(x)|^2 (the certificate's error bound uses the
// actual rounding residual, by Cauchy-Schwarz |<x,q> - <xb,qb>| <= |x - xb| |q| + |xb| |q - qb|).
__global__ __launch_bounds__(256) void shadow_kernel(const float* __restrict__ X, const float* __restrict__ xnorm, int64_t first, int64_t n, int ld,
                                                      bf16_t* __restrict__ Xb, unsigned* __restrict__ res_bits) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    float worst = 0.f, worst_rel = 0.f;
    for (int64_t r = wave0; r < n; r += nwaves) {
        const float* x = X + (first + r) * (int64_t)ld;
        bf16_t* o = Xb + (first + r) * (int64_t)ld;
        float res = 0.f;
        for (int k0 = 4 * lane; k0 < ld; k0 += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + k0);
            u16x4 b;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                b[c] = f32_to_bf16(v[c]);
                const float d = v[c] - bf16_to_f32(b[c]);
                res = fmaf(d, d, res);
            }
            *reinterpret_cast<u16x4*>(o + k0) = b;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) res += __shfl_xor(res, off, 64);
        worst = fmaxf(worst, res);
        const float xn = xnorm[first + r];
        if (xn > 0.f) worst_rel = fmaxf(worst_rel, res / xn);
    }
    if (lane == 0 && worst > 0.f) {
        atomicMax(res_bits, __builtin_bit_cast(unsigned, worst * 1.0001f));          // max |x - xb|^2
        atomicMax(res_bits + 1, __builtin_bit_cast(unsigned, worst_rel * 1.0001f));  // max |x - xb|^2 / |x|^2
    }
}
__global__ __launch_bounds__(256) void norm_max_kernel(const float* __restrict__ xnorm, int64_t n, unsigned* __restrict__ out_bits) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, xnorm[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __builtin_bit_cast(unsigned, m));  // non-negative floats order like their bits
}
// f32 padded queries [Q, ld] -> bf16 [Qpad, ld] (rows >= Q zero) and qres[q] = |q - bf16(q)|^2 ; one wave per query row
__global__ __launch_bounds__(256) void query_bf16_kernel(const float* __restrict__ Qp, int Q, int Qpad, int ld, bf16_t* __restrict__ Qb,
                                                          float* __restrict__ qres) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Qpad) return;
    float res = 0.f;
    for (int k0 = 4 * lane; k0 < ld; k0 += 256) {
        u16x4 b = {0, 0, 0, 0};
        if (q < Q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(Qp + (size_t)q * ld + k0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                b[c] = f32_to_bf16(v[c]);
                const float d = v[c] - bf16_to_f32(b[c]);
                res = fmaf(d, d, res);
            }
        }
        *reinterpret_cast<u16x4*>(Qb + (size_t)q * ld + k0) = b;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) res += __shfl_xor(res, off, 64);
    if (lane == 0 && q < Q) qres[q] = res * 1.0001f;
}

// int8 shadow: one wave per row.  Xq [rows][ld8] int8 (ld8 = ld rounded up to 128, zero padded), xscale[row] = s_r,
// res_bits[1] / [2] = max |x - s q|^2 and max |x - s q|^2 / |x|^2 over the rows (the certificate's bound).
template <bool QUERY>
__global__ __launch_bounds__(256) void shadow8_kernel(const float* __restrict__ X, const float* __restrict__ xnorm, int64_t first, int64_t n, int64_t nout,
                                                       int ld, int ld8, int8_t* __restrict__ Xq, float* __restrict__ xscale,
                                                       unsigned* __restrict__ res_bits, float* __restrict__ qres,
                                                       const unsigned* __restrict__ common_absmax_bits) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    float worst = 0.f, worst_rel = 0.f;
    for (int64_t r = wave0; r < nout; r += nwaves) {  // QUERY: rows n .. nout are padding (zeros, scale 1)
        const bool real = r < n;
        const float* x = X + (first + r) * (int64_t)ld;
        float m = 0.f;
        if (real)
            for (int k0 = 4 * lane; k0 < ld; k0 += 256) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + k0);
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if (QUERY) m = __builtin_bit_cast(float, *common_absmax_bits);  // queries: ONE scale for the batch (the epilogue's prefilter relies on it)
        const float sc = m > 0.f ? m * (1.0f / 127.0f) : 1.0f;
        const float inv = 1.0f / sc;
        float res = 0.f;
        int8_t* o = Xq + (first + r) * (int64_t)ld8;
        for (int k0 = 16 * lane; k0 < ld8; k0 += 1024) {
            u32x4 packed = {0u, 0u, 0u, 0u};
            if (real && k0 < ld) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(x + k0 + 4 * j);
                    uint32_t word = 0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float qf = fminf(fmaxf(rintf(v[c] * inv), -127.0f), 127.0f);
                        const float d = fmaf(-sc, qf, v[c]);
                        res = fmaf(d, d, res);
                        word |= ((uint32_t)(int)qf & 0xFFu) << (8 * c);
                    }
                    packed[j] = word;
                }
            }
            *reinterpret_cast<u32x4*>(o + k0) = packed;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) res += __shfl_xor(res, off, 64);
        if (lane == 0) xscale[first + r] = sc;
        if (QUERY) {
            if (lane == 0 && real) qres[r] = res * 1.0001f;
        } else {
            worst = fmaxf(worst, res);
            const float xn = xnorm[first + r];
            if (xn > 0.f) worst_rel = fmaxf(worst_rel, res / xn);
        }
    }
    if (!QUERY && lane == 0 && worst > 0.f) {
        atomicMax(res_bits, __builtin_bit_cast(unsigned, worst * 1.0001f));
        atomicMax(res_bits + 1, __builtin_bit_cast(unsigned, worst_rel * 1.0001f));
    }
}

// max |q_i| over the whole query batch (non-negative floats order like their bits)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ Qp, int64_t n, unsigned* __restrict__ out_bits) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(Qp[i]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __builtin_bit_cast(unsigned, m));
}

// ------------------------------------------------------------------ coarse GEMM + filter
struct CoarseArgs {
    const bf16_t* Xb;      // [rows padded to 128, ld]
    const float* xnorm;    // [n]
    int64_t row0, row1;    // phase row range [row0, row1), row0 % 128 == 0
    int ld;
    const bf16_t* Qb;      // [Qpad, ld]
    const float* qnorm;    // [Q]
    int Q, qtiles;
    const float* thr;      // [Q] threshold in v-space (v = score for L2, -score otherwise), +inf at start
    const float* thr_fast; // [Q] pre-adjusted threshold of the cheap test (superset of v <= thr)
    uint64_t* surv;        // [Q][cap]
    unsigned* count;       // [Q]
    int cap;
    int ntiles;
    const float* xscale;   // int8 stage: s_r per corpus row (rows padded to 256 hold anything finite)
    const float* qscale;   // int8 stage: s_q per query [Qpad]
    unsigned long long* trace;  // DBG = 2 (SC_COARSE_TRACE): [ntiles][4] = HW_ID | XCC_ID << 32, t_entry, t_mainloop_done, t_end (100 MHz)
};

template <int METRIC>
__global__ __launch_bounds__(256) void scan_coarse_kernel(CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int tile = xcd_remap(blockIdx.x, a.ntiles);
    const int rt = tile / a.qtiles, qt = tile - rt * a.qtiles;  // consecutive tiles share the corpus row panel
    const int64_t m0 = a.row0 + (int64_t)rt * G_BM;
    const int n0 = qt * G_BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_tile_mainloop(a.Xb + m0 * a.ld, a.ld, 0, a.Qb, a.ld, n0, a.ld, smem, acc, w, lane);

    // acc[ni][mi][r] = <x[m0 + wm*64 + mi*16 + fr], q[n0 + wn*64 + ni*16 + 4*fq + r]>  (bf16 inputs)
    const int fr = lane & 15, fq = lane >> 4;
    float xn[4], xs[4];
    int64_t rows[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        rows[mi] = m0 + wm * 64 + mi * 16 + fr;
        xn[mi] = rows[mi] < a.row1 ? a.xnorm[rows[mi]] : 0.f;
        xs[mi] = (METRIC == SC_METRIC_COSINE) ? 1.0f / sqrtf(xn[mi]) : 0.f;
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int q0 = n0 + wn * 64 + ni * 16 + 4 * fq;
        if (q0 >= a.Q) continue;
        const f32x4 tf = *reinterpret_cast<const f32x4*>(a.thr_fast + q0);  // thr_fast is allocated padded to 128
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            if (rows[mi] >= a.row1) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dot = acc[ni][mi][r];
                float t;
                if (METRIC == SC_METRIC_L2) t = fmaf(-2.0f, dot, xn[mi]);
                else if (METRIC == SC_METRIC_COSINE) t = -dot * xs[mi];
                else t = -dot;
                if (t <= tf[r]) {
                    const int q = q0 + r;
                    if (q < a.Q) {
                        const float sc = sc_score<METRIC>(dot, xn[mi], a.qnorm[q]);
                        const float v = (METRIC == SC_METRIC_L2) ? sc : -sc;
                        if (v <= a.thr[q]) {
                            const unsigned pos = atomicAdd(a.count + q, 1u);
                            if (pos < (unsigned)a.cap) a.surv[(size_t)q * a.cap + pos] = sc_make_key<METRIC>(sc, (uint32_t)rows[mi]);
                        }
                    }
                }
            }
        }
    }
}

// 256 x 256 tile variant (gemm_tile.h, second half): 8 waves, each 128 corpus rows x 64 queries.
// Per-query thresholds / norms of the workgroup's 256 queries sit in LDS behind the pipeline buffers, and a
// lane queues its (rare) hits in registers so that the global atomics that allocate list slots are issued
// back to back and their latency is paid once per tile, not once per hit.
#define COARSE_QLDS (4 * T_TILE_BYTES)  // byte offset of {thr_fast[256], thr[256], qnorm[256], xnorm[256], qscale[256], xscale[256], row bounds [256][2]} in LDS
#define COARSE_LDS_BYTES (4 * T_TILE_BYTES + 8 * 256 * 4)
// per-tile staging of the workgroup's 256 query thresholds / norms and the tile's 256 row norms (visible to everyone after
// the main loop's barriers)
// ROWS / QUERIES: which half of the staging a call does (the persistent kernel stages the query side only when its query tile
// changes and feeds the row side from registers it loaded a tile earlier)
// int8 stage: the two per-row constants of the epilogue's prefilter.  The fast test t <= tf_q is  acc >= (base_r - tf_q) R_r  with
// base_r = |x|^2 (L2) or 0, R_r = 1 / (c_r s_r s_q), c_r = 2 (L2), 1 (IP), 1/|x| (cosine) and ONE query scale s_q per batch; with the
// loosest tf of a lane's 16 queries the integer bound of a row is  Ti = (int)(A_r - tfmax' B_r)  -- one FMA per row block in the
// epilogue instead of a reciprocal, the slack arithmetic and the clamps, which every lane of every tile recomputed (8 waves x 8 row
// blocks; the epilogue is issue bound: 2.1-3.4 us per tile, profiles/r3k_coarse_trace.log).  A_r already carries every slack of the
// row side (4e-6 relative, 2 units for the bound's own rounding, 1 for the truncation), B_r > 0 always; rows beyond the phase's end
// (scale 0) get a bound nothing passes.
template <int METRIC>
static __device__ __forceinline__ void coarse_row_bound(float xn, float sx, float sq0, float& A, float& B) {
    if (!(sx > 0.f)) { A = 3.0e38f; B = 1.0e-30f; return; }
    const float c = (METRIC == SC_METRIC_L2) ? 2.0f * sx : (METRIC == SC_METRIC_COSINE) ? sx / sqrtf(xn) : sx;
    float R = __builtin_amdgcn_rcpf(c * sq0);
    if (!(R < 3.0e38f)) R = 3.0e38f;   // (cosine, |x| = 0: c = inf -> R = 0 is fine; c = 0 cannot happen with sx > 0 and finite xn)
    if (!(R > 1.0e-30f)) R = 1.0e-30f;
    float base = (METRIC == SC_METRIC_L2) ? xn * R : 0.f;
    if (!(fabsf(base) < 3.0e38f)) base = -3.0e38f;  // inf / NaN: let everything through to the precise test
    A = base - fabsf(base) * 4e-6f - 3.0f;
    B = R;
}
template <bool I8 = false, bool ROWS = true, bool QUERIES = true, int METRIC = SC_METRIC_L2>
static __device__ __forceinline__ void coarse256_stage(const CoarseArgs& a, int64_t m0, int n0, char* smem, int tid) {
    float* q_tf = reinterpret_cast<float*>(smem + COARSE_QLDS);
    if (ROWS && tid >= 256) {
        const int64_t row = m0 + (tid - 256);
        const float xn = row < a.row1 ? a.xnorm[row] : 1.0f;
        q_tf[768 + tid - 256] = xn;
        if (I8) {
            const float sx = row < a.row1 ? a.xscale[row] : 0.0f;
            q_tf[1280 + tid - 256] = sx;
            float A, B;
            coarse_row_bound<METRIC>(xn, sx, a.qscale[0], A, B);
            *reinterpret_cast<f32x2*>(q_tf + 1536 + 2 * (tid - 256)) = f32x2{A, B};
        }
    }
    if (QUERIES && tid < 256) {
        const int q = n0 + tid;
        q_tf[tid] = q < a.Q ? a.thr_fast[q] : -__builtin_inff();  // padding never passes (and does not loosen the lane's prefilter bound)
        q_tf[256 + tid] = q < a.Q ? a.thr[q] : -__builtin_inff();
        q_tf[512 + tid] = q < a.Q ? a.qnorm[q] : 1.0f;
        if (I8) q_tf[1024 + tid] = a.qscale[q];  // padded to Qpad
    }
}
// tile (rt, qt) of logical tile index `tile`: column-major walk inside groups of 8 row panels (as the encoder GEMMs once did)
static __device__ __forceinline__ void coarse256_coords(const CoarseArgs& a, int tile, int64_t& m0, int& n0) {
    const int G = 8, rtiles = a.ntiles / a.qtiles;
    const int gsz = G * a.qtiles, g = tile / gsz, r = tile - g * gsz;
    const int rows_here = (g * G + G <= rtiles) ? G : rtiles - g * G;
    m0 = a.row0 + (int64_t)(g * G + r % rows_here) * T_BM;
    n0 = (r / rows_here) * T_BN;
}
// Epilogue.  One compare per score against a per-row bound, a wave-uniform branch per group of 4 scores around the precise test.
// A variant without that branch -- every lane that passes the bound runs the precise test under its own exec mask and parks
// its key in a per-wave LDS list flushed once per tile -- was measured on the same box and lost (int8 stage 11.25 -> 12.28 ms,
// bf16 15.05 -> 16.2 ms per step, gpurun_out/r2f_scan_*.log): the divergent bodies cost more than the uniform branch saves.
// So did a two-phase form (branch-free bound tests into a 32-bit group mask, DPP OR over the wave, then a loop over the set bits
// with the precise test present once and the accumulators fetched by a switch): the per-workgroup stamps (SC_COARSE_TRACE,
// profiles/r2j_coarse_trace.log) put this version at 3.4 / 4.4 us of epilogue per tile with 10 / 35 of 256 groups entering the
// precise test (0.46 us of that is wave skew, ~1.9 us the 32 bound tests: a wave64 VALU instruction issues over 4 cycles and two
// waves share a SIMD), the two-phase form at 3.4-3.7 / 5.5-6.2 us -- its loop body costs more per entered group than 32 unrolled
// copies do.  Removing the returning atomics changed nothing (4.3 vs 4.4 us).
template <int METRIC, bool I8 = false, bool TRACE = false, bool NOATOM = false, bool NOPRECISE = false, bool ROWTEST_OFF = false>
static __device__ __forceinline__ void coarse256_epilogue(const CoarseArgs& a, const f32x4 (&acc)[4][8], int64_t m0, int n0, char* smem, int w,
                                                          int lane) {
    const int wm = w >> 2, wn = w & 3;
    asm volatile("" : "+v"(lane));
    const float* q_tf = reinterpret_cast<const float*>(smem + COARSE_QLDS);
    const float* q_thr = q_tf + 256;
    const float* q_qn = q_tf + 512;
    const float* x_xn = q_tf + 768;  // |x|^2 of the tile's 256 corpus rows
    // acc[ni][mi][r] = <x[m0 + wm*128 + mi*16 + fr], q[n0 + wn*64 + ni*16 + 4*fq + r]>  (bf16 inputs)
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 tf[4], sq[4];  // sq: int8 stage only, the query scales of this lane's 16 columns
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        tf[ni] = *reinterpret_cast<const f32x4*>(q_tf + wn * 64 + ni * 16 + 4 * fq);
        if (I8) sq[ni] = *reinterpret_cast<const f32x4*>(q_tf + 1024 + wn * 64 + ni * 16 + 4 * fq);
    }
    float tfmax = -__builtin_inff();  // loosest fast threshold among this lane's 16 queries
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) tfmax = fmaxf(fmaxf(tfmax, fmaxf(tf[ni][0], tf[ni][1])), fmaxf(tf[ni][2], tf[ni][3]));
    // int8: the lane's side of the row bound (coarse_row_bound): tfmax with its own relative slack, rounded up
    const float tfm = fabsf(tfmax) < 3.0e38f ? fmaf(fabsf(tfmax), 4e-6f, tfmax) : tfmax;
    const f32x2* rowb = reinterpret_cast<const f32x2*>(q_tf + 1536);
    // hits of this lane: up to 4 queued (local query index, key); a 5th and later ones are flushed directly
    int nh = 0;
    int hq0 = 0, hq1 = 0, hq2 = 0, hq3 = 0;
    uint64_t hk0 = 0, hk1 = 0, hk2 = 0, hk3 = 0;
    // int8: a first, branch-free pass over the 8 row blocks -- bound, maximum of the lane's 16 scores, one ballot each -- collects
    // which row blocks hold anything at all; in the late phases of a batch (most of its tiles) none does and the epilogue ends here.
    // The per-row-block form below is ~2 KB of code per block, nearly all of it cold: its hot path hopped over 17 000 instructions
    // in eight jumps (2.1-2.3 us per tile with NOTHING passing, profiles/r3k_coarse_trace.log); this pass is ~150 contiguous ones.
    unsigned blockmask = 0xFFu;
    int TiA[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (I8 && !ROWTEST_OFF) {
        blockmask = 0u;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const f32x2 ab = rowb[wm * 128 + mi * 16 + fr];
            const float tl = fmaf(-tfm, ab[1], ab[0]);
            TiA[mi] = (int)__builtin_amdgcn_fmed3f(tl, -2.0e9f, 2.0e9f);
            int mx = max(max(__float_as_int(acc[0][mi][0]), __float_as_int(acc[0][mi][1])), max(__float_as_int(acc[0][mi][2]), __float_as_int(acc[0][mi][3])));
#pragma unroll
            for (int ni = 1; ni < 4; ++ni)
                mx = max(max(mx, max(__float_as_int(acc[ni][mi][0]), __float_as_int(acc[ni][mi][1]))), max(__float_as_int(acc[ni][mi][2]), __float_as_int(acc[ni][mi][3])));
            blockmask |= __any(mx >= TiA[mi]) ? (1u << mi) : 0u;
        }
        blockmask = __builtin_amdgcn_readfirstlane(blockmask);
        if (blockmask == 0u || a.cap == -2) return;
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        if (I8 && !(blockmask & (1u << mi))) continue;
        const int rl = wm * 128 + mi * 16 + fr;
        int Ti = 0;
        if (I8) {
            if (ROWTEST_OFF) {
                const f32x2 ab = rowb[rl];
                Ti = (int)__builtin_amdgcn_fmed3f(fmaf(-tfm, ab[1], ab[0]), -2.0e9f, 2.0e9f);
            } else Ti = TiA[mi];
        }
        const float xn = x_xn[rl];  // staged at kernel start; rows >= row1 hold +inf (L2) / 0 scale so that they never pass
        const float xs = (METRIC == SC_METRIC_COSINE) ? 1.0f / sqrtf(xn) : 0.f;  // exact: the precise test below uses it too
        const float sx = I8 ? q_tf[1280 + rl] : 0.f;  // int8 stage: the integer dot is scaled by s_r s_q
        const float ar = (METRIC == SC_METRIC_L2) ? -2.0f * sx : (METRIC == SC_METRIC_COSINE) ? -sx * xs : -sx;
        // Prefilter: ONE compare per score.  The fast test t <= tf_q is  dot >= (base_r - tf_q) / c_r  with base_r = |x|^2 (L2) or
        // 0 and c_r = 2 (L2), 1 (IP), 1/|x| (cosine), times s_r s_q in the int8 stage, where every query of the batch shares one
        // scale (sc_launch_query_i8) -- so the right-hand side differs between this lane's 16 queries only through tf_q, and with
        // tfmax = max of those it is bounded below by a per-row constant.  Anything that passes is tested precisely below.
        float Tlb = 0.f;
        if (!I8) {
            Tlb = (METRIC == SC_METRIC_L2) ? 0.5f * (xn - tfmax) : (METRIC == SC_METRIC_COSINE) ? -tfmax / xs : -tfmax;
            Tlb = Tlb - fabsf(Tlb) * 4e-6f;  // rounding of this bound itself (the precise test has its own slack)
            if (!(Tlb == Tlb)) Tlb = -__builtin_inff();  // NaN (0 * inf on an all-zero row): let the precise test decide
        }
        // Round 3: first ONE test per row block -- the maximum of the lane's 16 scores (8 v_max3) against the bound, one wave-uniform
        // branch per 16 x 64 scores instead of four; the per-group tests below run only for the row blocks that pass (15-45 % of
        // them).  The bound tests were 1.9 us of a 3.4-4.4 us epilogue at ~28 vector instructions per row block (a wave64
        // instruction issues over 4 cycles, two waves share a SIMD); the common path is now 9.
        if (!I8) {
            float mx = fmaxf(fmaxf(acc[0][mi][0], acc[0][mi][1]), fmaxf(acc[0][mi][2], acc[0][mi][3]));
#pragma unroll
            for (int ni = 1; ni < 4; ++ni) mx = fmaxf(fmaxf(mx, fmaxf(acc[ni][mi][0], acc[ni][mi][1])), fmaxf(acc[ni][mi][2], acc[ni][mi][3]));
            if (!__any(mx >= Tlb) && !ROWTEST_OFF) continue;
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            // one uniform branch per group of 4 scores; taken by ~1 group in 500 once thresholds are tight
            bool g;
            if (I8) {
                // (__float_as_int, not __builtin_bit_cast(int, acc[ni][mi][r]): hipcc lowers the bit_cast of a vector-ELEMENT lvalue as
                // element 0 -- the first int8 build tested the wrong accumulators for three queries in four)
                // the group's maximum against the bound: 3 instructions instead of 4 compares and 3 ORs (a wave64 instruction issues
                // over 4 cycles, two waves share a SIMD, and a third of all row blocks get here: SC_COARSE_EXPERIMENT)
                const int gm = max(max(__float_as_int(acc[ni][mi][0]), __float_as_int(acc[ni][mi][1])),
                                   max(__float_as_int(acc[ni][mi][2]), __float_as_int(acc[ni][mi][3])));
                g = gm >= Ti;
            } else {
                g = (acc[ni][mi][0] >= Tlb) | (acc[ni][mi][1] >= Tlb) | (acc[ni][mi][2] >= Tlb) | (acc[ni][mi][3] >= Tlb);
            }
            if (!__any(g) || NOPRECISE || a.cap == -3) continue;  // NOPRECISE / cap -3: diagnostic (results invalid), what the bound tests alone cost
            if (TRACE && lane == 0) atomicAdd(&a.trace[(size_t)blockIdx.x * 8 + 4], 1ull);
            f32x4 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (I8) {
                    const float av = (float)__float_as_int(acc[ni][mi][r]) * ar;
                    t[r] = (METRIC == SC_METRIC_L2) ? fmaf(av, sq[ni][r], xn) : av * sq[ni][r];
                } else {
                    const float dot = acc[ni][mi][r];
                    if (METRIC == SC_METRIC_L2) t[r] = fmaf(-2.0f, dot, xn);
                    else if (METRIC == SC_METRIC_COSINE) t[r] = -dot * xs;
                    else t[r] = -dot;
                }
            }
            {
                const int64_t row = m0 + rl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (int8: entering the precise test on the integer bound alone -- it is a superset of this float test -- saved these
                    // 12 instructions per entered group but ran the precise test 1.3x as often: 7.94 -> 7.95 ms of kernels per step)
                    if (row < a.row1 && t[r] <= tf[ni][r]) {
                        if (TRACE) atomicAdd(&a.trace[(size_t)blockIdx.x * 8 + 5], 1ull);
                        const int ql = wn * 64 + ni * 16 + 4 * fq + r;
                        const float accv = acc[ni][mi][r];
                        const float dotv = I8 ? (float)__float_as_int(accv) * (sx * sq[ni][r]) : accv;
                        const float sc = sc_score<METRIC>(dotv, xn, q_qn[ql]);
                        const float v = (METRIC == SC_METRIC_L2) ? sc : -sc;
                        if (v <= q_thr[ql]) {  // q_thr = -inf for padded queries
                            if (TRACE) atomicAdd(&a.trace[(size_t)blockIdx.x * 8 + 6], 1ull);
                            const uint64_t key = sc_make_key<METRIC>(sc, (uint32_t)row);
                            if (nh == 0) { hq0 = ql; hk0 = key; }
                            else if (nh == 1) { hq1 = ql; hk1 = key; }
                            else if (nh == 2) { hq2 = ql; hk2 = key; }
                            else if (nh == 3) { hq3 = ql; hk3 = key; }
                            else {
                                const unsigned pos = atomicAdd(a.count + n0 + ql, 1u);
                                if (pos < (unsigned)a.cap) a.surv[(size_t)(n0 + ql) * a.cap + pos] = key;
                            }
                            ++nh;
                        }
                    }
                }
            }
        }
    }
    if (!__any(nh > 0)) return;
    // allocate the queued hits' list slots with back-to-back atomics, then store
    unsigned p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    if (NOATOM) {  // diagnostic (results invalid): what the returning atomics cost
        p0 = lane; p1 = 64 + lane; p2 = 128 + lane; p3 = 192 + lane;
    } else {
        if (nh > 0) p0 = atomicAdd(a.count + n0 + hq0, 1u);
        if (nh > 1) p1 = atomicAdd(a.count + n0 + hq1, 1u);
        if (nh > 2) p2 = atomicAdd(a.count + n0 + hq2, 1u);
        if (nh > 3) p3 = atomicAdd(a.count + n0 + hq3, 1u);
    }
    if (nh > 0 && p0 < (unsigned)a.cap) a.surv[(size_t)(n0 + hq0) * a.cap + p0] = hk0;
    if (nh > 1 && p1 < (unsigned)a.cap) a.surv[(size_t)(n0 + hq1) * a.cap + p1] = hk1;
    if (nh > 2 && p2 < (unsigned)a.cap) a.surv[(size_t)(n0 + hq2) * a.cap + p2] = hk2;
    if (nh > 3 && p3 < (unsigned)a.cap) a.surv[(size_t)(n0 + hq3) * a.cap + p3] = hk3;
}

// Epilogue for tiles in which MOST scores survive (the first phases of a batch: thresholds are still +inf or loose, a 256 x 256
// tile yields thousands of survivors).  The epilogue above allocates a list slot per hit with a returning global atomic -- four
// queued per lane, the rest one round trip each: the first six launches of a 10M-row batch (3.5 % of the rows) took 2.5 of its
// 10.9 ms (gpurun_out/r3l: 500, 644, 411, 243, 286, 407 us; one 256-row tile with 9 400 hits 230 us).  Here a tile makes ONE global
// atomic per query: pass 1 counts the hits per query in LDS (the pipeline buffers are dead), 256 threads reserve [base, base + n)
// of each query's list, pass 2 re-evaluates the same tests and writes each hit at base + an LDS ticket.  Same survivor set as
// the sparse epilogue (same tests, same arithmetic); the order inside a list differs, which the selection does not see.
// Those six launches now take 86, 86, 64, 58, 198, 553 us (gpurun_out/r3n), the step 10.7 -> 9.1 ms.
template <int METRIC, bool I8>
static __device__ __forceinline__ void coarse256_epilogue_dense(const CoarseArgs& a, const f32x4 (&acc)[4][8], int64_t m0, int n0, char* smem, int w,
                                                                int lane, int tid) {
    const int wm = w >> 2, wn = w & 3;
    asm volatile("" : "+v"(lane));
    const float* q_tf = reinterpret_cast<const float*>(smem + COARSE_QLDS);
    const float* q_thr = q_tf + 256;
    const float* q_qn = q_tf + 512;
    const float* x_xn = q_tf + 768;
    unsigned* cnt = reinterpret_cast<unsigned*>(smem);  // [256] hits per local query, then the tickets of pass 2
    unsigned* base = cnt + 256;                         // [256] first list slot of this tile's hits
    const int fr = lane & 15, fq = lane >> 4;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is out of the main loop: the ring is free
    if (tid < 256) cnt[tid] = 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    f32x4 tf[4], sq[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        tf[ni] = *reinterpret_cast<const f32x4*>(q_tf + wn * 64 + ni * 16 + 4 * fq);
        if (I8) sq[ni] = *reinterpret_cast<const f32x4*>(q_tf + 1024 + wn * 64 + ni * 16 + 4 * fq);
    }
    // the tests of coarse256_epilogue, score by score: fast test, then the precise one; key valid when it returns true
    auto hit = [&](int mi, int ni, int r, uint64_t& key) -> bool {
        const int rl = wm * 128 + mi * 16 + fr;
        const int64_t row = m0 + rl;
        const float xn = x_xn[rl];
        const float xs = (METRIC == SC_METRIC_COSINE) ? 1.0f / sqrtf(xn) : 0.f;
        const float sx = I8 ? q_tf[1280 + rl] : 0.f;
        const float ar = (METRIC == SC_METRIC_L2) ? -2.0f * sx : (METRIC == SC_METRIC_COSINE) ? -sx * xs : -sx;
        const float accv = acc[ni][mi][r];
        float t;
        if (I8) {
            const float av = (float)__float_as_int(accv) * ar;
            t = (METRIC == SC_METRIC_L2) ? fmaf(av, sq[ni][r], xn) : av * sq[ni][r];
        } else {
            if (METRIC == SC_METRIC_L2) t = fmaf(-2.0f, accv, xn);
            else if (METRIC == SC_METRIC_COSINE) t = -accv * xs;
            else t = -accv;
        }
        if (!(row < a.row1 && t <= tf[ni][r])) return false;
        const int ql = wn * 64 + ni * 16 + 4 * fq + r;
        const float dotv = I8 ? (float)__float_as_int(accv) * (sx * sq[ni][r]) : accv;
        const float sc = sc_score<METRIC>(dotv, xn, q_qn[ql]);
        const float v = (METRIC == SC_METRIC_L2) ? sc : -sc;
        if (!(v <= q_thr[ql])) return false;
        key = sc_make_key<METRIC>(sc, (uint32_t)row);
        return true;
    };
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                uint64_t key;
                if (hit(mi, ni, r, key)) __hip_atomic_fetch_add(&cnt[wn * 64 + ni * 16 + 4 * fq + r], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < 256) {
        const unsigned c = cnt[tid];
        unsigned b = 0;
        if (c) b = atomicAdd(a.count + n0 + tid, c);  // (padded queries never hit: their threshold is -inf)
        base[tid] = b;
        cnt[tid] = 0;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                uint64_t key;
                if (hit(mi, ni, r, key)) {
                    const int ql = wn * 64 + ni * 16 + 4 * fq + r;
                    const unsigned pos = base[ql] + __hip_atomic_fetch_add(&cnt[ql], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (pos < (unsigned)a.cap) a.surv[(size_t)(n0 + ql) * a.cap + pos] = key;
                }
            }
}

// PP: 0 = one barrier per K-tile, 2..5 = the ping-pong main loop with that many half-tiles in flight (gemm_tile.h)
// DENSE: the two-pass epilogue above (launches whose tiles are expected to keep hundreds of survivors)
template <int METRIC, int DBG = 0, bool I8 = false, int PP = 4, bool DENSE = false>
__global__ __launch_bounds__(512) void scan_coarse256_kernel(CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int64_t m0;
    int n0;
    if ((DBG & 2) && tid == 0) {
        a.trace[(size_t)blockIdx.x * 8 + 0] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        a.trace[(size_t)blockIdx.x * 8 + 1] = (unsigned long long)wall_clock64();
    }
    coarse256_coords(a, xcd_remap(blockIdx.x, a.ntiles), m0, n0);
    // (requesting the first two K-tiles before this staging -- so that the two memory round trips overlap -- measured no change:
    // entry -> main loop done stayed at 11.2 us per int8 tile)
    coarse256_stage<I8, true, true, METRIC>(a, m0, n0, smem, tid);
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // int8 stage: a.ld counts PAIRS of int8 (the tile machinery addresses 2-byte elements), zero accumulators are zero i32 bits
    if constexpr (PP > 0) gemm_tile256_mainloop_pp<PP, 0, NoTailHook, I8>(a.Xb + m0 * a.ld, a.ld, 0, a.Qb, a.ld, n0, a.ld, smem, acc, w, lane);
    else gemm_tile256_mainloop<0, NoTailHook, I8>(a.Xb + m0 * a.ld, a.ld, 0, a.Qb, a.ld, n0, a.ld, smem, acc, w, lane);
    asm volatile("" ::: "memory");  // keep the epilogue's loads out of the register-tight main loop
    __builtin_amdgcn_sched_barrier(0);
    if ((DBG & 2) && tid == 0) a.trace[(size_t)blockIdx.x * 8 + 2] = (unsigned long long)wall_clock64();
    if (DBG & 1) {  // diagnostic: main loop only
        float sink = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) sink += acc[i][j][0] + acc[i][j][3];
        if (sink == 12345.678f) a.count[0] = 1;
        if (DBG & 2) {
            __syncthreads();
            if (tid == 0) a.trace[(size_t)blockIdx.x * 8 + 3] = (unsigned long long)wall_clock64();
        }
        return;
    }
    if constexpr (DENSE) coarse256_epilogue_dense<METRIC, I8>(a, acc, m0, n0, smem, w, lane, tid);
    else coarse256_epilogue<METRIC, I8, (DBG & 2) != 0, (DBG & 4) != 0, (DBG & 8) != 0>(a, acc, m0, n0, smem, w, lane);
    if (DBG & 2) {
        __syncthreads();
        if (tid == 0) a.trace[(size_t)blockIdx.x * 8 + 3] = (unsigned long long)wall_clock64();
    }
}

// A persistent variant (one workgroup per CU walking tiles b, b + grid, ..., the tail hook of tile t requesting K-tiles 0 and 1 of
// tile t + grid under its last 32 MFMAs) was built and measured in a same-box A/B: 64.7k -> 60.4k QPS.  Like both persistent walks
// tried on the encoder GEMMs it loses to hardware dispatch of one workgroup per tile; DESIGN.md section 8.

// ------------------------------------------------------------------ per-phase selection: keep the kp (KPRIME or KPRIME8) best coarse keys
// best [Q][kp] keys in no particular order (SC_KEY_MAX padded).  One workgroup per query.
//
// The kp-th smallest of the n <= cap + kp keys (survivors of this phase + the previous best) is found by a radix select over
// the 64-bit keys, 12 bits per pass from the top (4096-bin histogram in LDS, atomics; after the first pass only the keys of one
// bin are still in play), then every key <= that pivot is kept -- exactly kp of them, keys being unique (row id in the low
// word).  The rank sort this replaces was quadratic in n: 0.31 ms per step at kp = 128 and 2.6 ms at kp = 512
// (gpurun_out/r2c_scan_i8.log); the re-rank never needed the list sorted.
#define SEL_BINS 4096
template <int METRIC>
__global__ __launch_bounds__(SEL_THREADS) void scan_select_kernel(uint64_t* __restrict__ surv, unsigned* __restrict__ count, int cap,
                                                                   uint64_t* __restrict__ best, const float* __restrict__ qnorm,
                                                                   float* __restrict__ thr, float* __restrict__ thr_fast,
                                                                   int* __restrict__ overflow, int kp) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sel_lds[];  // [cap + kp] candidates | hist[SEL_BINS] | scan[SEL_THREADS] | misc
    uint64_t* cand = sel_lds;
    unsigned* hist = reinterpret_cast<unsigned*>(sel_lds + cap + kp);
    unsigned* part = hist + SEL_BINS;          // per-thread partial sums of the bin scan
    unsigned* misc = part + SEL_THREADS;       // [0] n, [1] chosen bin, [2] rank inside it, [3] output cursor
    const int q = blockIdx.x, tid = threadIdx.x;
    unsigned c = count[q];
    if (c > (unsigned)cap) {
        if (tid == 0) overflow[q] = 1;
        c = cap;
    }
    if (tid == 0) { misc[0] = 0; misc[3] = 0; }
    __syncthreads();
    // gather the real keys (the previous best is padded with SC_KEY_MAX)
    for (int i = tid; i < (int)c + kp; i += SEL_THREADS) {
        const uint64_t key = i < (int)c ? surv[(size_t)q * cap + i] : best[(size_t)q * kp + (i - (int)c)];
        if (key != SC_KEY_MAX) cand[atomicAdd(&misc[0], 1u)] = key;
    }
    __syncthreads();
    const int n = (int)misc[0];
    uint64_t pivot = SC_KEY_MAX;  // keep every key <= pivot
    bool have_kth = false;
    if (n > kp) {
        // radix select of rank kp - 1 (0-based): fixed high bits `prefix` (width 64 - shift - 12 ... ), remaining rank `want`
        uint64_t prefix = 0;
        unsigned want = (unsigned)(kp - 1);
        for (int shift = 52; shift >= -8; shift -= 12) {  // 52, 40, 28, 16, 4, then the last 4 bits (shift -8 -> 4-bit digit)
            const int sh = shift < 0 ? 0 : shift;
            const int bits = shift < 0 ? 4 : 12;
            const unsigned mask = (1u << bits) - 1u;
            for (int i = tid; i < SEL_BINS; i += SEL_THREADS) hist[i] = 0;
            __syncthreads();
            const int hi_shift = sh + bits;  // bits above the current digit must equal prefix
            for (int i = tid; i < n; i += SEL_THREADS) {
                const uint64_t key = cand[i];
                if (hi_shift >= 64 || (key >> hi_shift) == prefix) atomicAdd(&hist[(unsigned)(key >> sh) & mask], 1u);
            }
            __syncthreads();
            // bin scan: thread t owns bins [16 t, 16 t + 16); exclusive prefix over the threads by wave shuffles + 4 wave totals
            unsigned local = 0;
            for (int j = 0; j < SEL_BINS / SEL_THREADS; ++j) local += hist[tid * (SEL_BINS / SEL_THREADS) + j];
            unsigned incl = local;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl_up(incl, off, 64);
                if ((tid & 63) >= off) incl += o;
            }
            if ((tid & 63) == 63) part[tid >> 6] = incl;
            __syncthreads();
            unsigned pre = incl - local;
            for (int w = 0; w < (tid >> 6); ++w) pre += part[w];
            if (pre <= want && want < pre + local) {  // exactly one thread: the wanted rank falls into its 16 bins
                unsigned acc = pre;
                int bin = tid * (SEL_BINS / SEL_THREADS);
                for (;; ++bin) {
                    if (acc + hist[bin] > want) break;
                    acc += hist[bin];
                }
                misc[1] = (unsigned)bin;
                misc[2] = want - acc;
            }
            __syncthreads();
            prefix = (prefix << bits) | (uint64_t)misc[1];
            want = misc[2];
            __syncthreads();
        }
        pivot = prefix;
        have_kth = true;
    } else if (n == kp) {
        // exactly kp keys: all stay, the threshold is their maximum
        uint64_t m = 0;
        for (int i = tid; i < n; i += SEL_THREADS) m = cand[i] > m ? cand[i] : m;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const uint64_t o = ((uint64_t)__shfl_xor((unsigned)(m >> 32), off, 64) << 32) | (uint64_t)__shfl_xor((unsigned)m, off, 64);
            m = o > m ? o : m;
        }
        uint64_t* wmax = reinterpret_cast<uint64_t*>(hist);
        if ((tid & 63) == 0) wmax[tid >> 6] = m;
        __syncthreads();
        pivot = wmax[0];
        for (int w = 1; w < SEL_THREADS / 64; ++w) pivot = wmax[w] > pivot ? wmax[w] : pivot;
        have_kth = true;
        __syncthreads();
    }
    for (int i = tid; i < n; i += SEL_THREADS) {
        const uint64_t key = cand[i];
        if (key <= pivot) best[(size_t)q * kp + atomicAdd(&misc[3], 1u)] = key;
    }
    __syncthreads();
    const int kept = (int)misc[3];  // == min(n, kp)
    for (int i = kept + tid; i < kp; i += SEL_THREADS) best[(size_t)q * kp + i] = SC_KEY_MAX;
    if (tid == 0) {
        count[q] = 0;
        float t = __builtin_inff(), tf = __builtin_inff();
        if (have_kth) {
            const float sc = sc_key_score(METRIC, pivot);
            t = (METRIC == SC_METRIC_L2) ? sc : -sc;
            const float slack = 1e-3f * fabsf(t) + 1e-6f;
            const float qn = qnorm[q];
            if (METRIC == SC_METRIC_L2) tf = (t - qn) + slack + 1e-3f * fabsf(qn);
            else if (METRIC == SC_METRIC_COSINE) tf = (t + slack) * sqrtf(qn) + 1e-5f * sqrtf(qn);  // -dot/|x| <= (t+slack)*|q|  (|q| > 0)
            else tf = t + slack;
        }
        thr[q] = t;
        thr_fast[q] = tf;
    }
}

// ------------------------------------------------------------------ exact re-rank + certificate
// One 128-thread workgroup per query: thread j re-scores candidate j with the canonical f32 fmaf chain.  The 128 candidate rows
// are scattered over the corpus; read row-per-thread, every load instruction touched 64 different cache lines and the kernel
// took 0.34 ms per 1024 queries.  So the rows are fetched cooperatively, 64 floats of all 128 rows at a time (16 lanes x 16 B
// = one 256-byte row segment per load), through an LDS tile [128][64 + 4] (the pad keeps the row-per-thread ds_read_b128 of
// the chain conflict-free), the next tile's loads in flight while this one is multiplied.  The chain itself is unchanged:
// k = 16 t + 4 g + c, c outer, g inner -- bit-identical scores.
#define RR_STRIDE 68  // floats per LDS row

// The certificate: is the k-th exact key provably better than every row that is NOT a candidate?  Such a row has coarse
// v-score >= tau, hence exact v-score >= tau - eps with
//   |coarse dot - exact dot| <= |x - xc| |q| + |xc| |q - qc| + accumulation noise   (Cauchy-Schwarz on the ACTUAL rounding
// residuals of the coarse operands xc, qc -- bf16 or scaled int8; bits = {max |x|^2, max |x - xc|^2, max |x - xc|^2 / |x|^2}
// over the corpus, qres = |q - qc|^2).  The int8 dot itself is exact; its scaling to f32 rounds three times, which the
// ld * 1.2e-7 term (sized for the f32 accumulation of the bf16 stage) covers many times over.
template <int METRIC>
static __device__ __forceinline__ float certificate_eps(const unsigned* __restrict__ bits, float qnorm2, float qres2, int ld) {
    const float xmax = sqrtf(__builtin_bit_cast(float, bits[0]));
    const float xres = sqrtf(__builtin_bit_cast(float, bits[1]));
    const float qn = sqrtf(qnorm2);
    const float ddot = xres * qn + (xmax + xres) * sqrtf(qres2) + (float)ld * 1.2e-7f * (xmax + xres) * qn;
    float eps;
    if (METRIC == SC_METRIC_L2) eps = 2.0f * ddot;
    else if (METRIC == SC_METRIC_COSINE) {  // relative form: |x - xc|/|x| + (1 + .)|q - qc|/|q|
        const float relx = sqrtf(__builtin_bit_cast(float, bits[2]));
        eps = relx + (1.0f + relx) * (sqrtf(qres2) / fmaxf(qn, 1e-30f)) + (float)ld * 1.2e-7f * (1.0f + relx);
    } else eps = ddot;
    return eps * 1.01f + 1e-6f;
}
template <int METRIC>
static __device__ __forceinline__ bool certified(uint64_t kth_key, const unsigned* __restrict__ bits, float qnorm2, float qres2, int ld, float tau) {
    const float sc = sc_key_score(METRIC, kth_key);
    const float vk = (METRIC == SC_METRIC_L2) ? sc : -sc;
    return vk + certificate_eps<METRIC>(bits, qnorm2, qres2, ld) < tau;
}
// SPLIT (int8 stage, KPRIME8 candidates): blockIdx.y selects a block of 128 candidates, the exact keys go to ekeys [Q][kp] and
// scan_finalize_kernel sorts them and evaluates the certificate.
template <int METRIC, bool SPLIT = false>
__global__ __launch_bounds__(KPRIME) void scan_rerank_kernel(const float* __restrict__ X, const float* __restrict__ xnorm, int ld,
                                                          const float* __restrict__ Qp, const float* __restrict__ qnorm,
                                                          const uint64_t* __restrict__ best, const float* __restrict__ thr,
                                                          const unsigned* __restrict__ xnorm_max_bits, const float* __restrict__ qres,
                                                          const int* __restrict__ overflow, int k,
                                                          int64_t row_base, const uint32_t* __restrict__ perm, float* __restrict__ out_dist,
                                                          int64_t* __restrict__ out_rows, int* __restrict__ flags, int kp = KPRIME,
                                                          uint64_t* __restrict__ ekeys = nullptr, const int* __restrict__ ncand = nullptr) {
    static_assert(KPRIME == 128, "the cooperative tile load assumes 128 candidates = 128 threads");
    __shared__ uint64_t keys[KPRIME];
    __shared__ uint32_t rowid[KPRIME];
    __shared__ __attribute__((aligned(16))) float tile[KPRIME * RR_STRIDE];
    __shared__ __attribute__((aligned(16))) float qch[64];
    const int q = blockIdx.x, lane = threadIdx.x;
    // ncand (SPLIT; the widened re-rank of the IVF coarse stage): only the first ncand[q] of the kp slots hold keys
    const int nc = (SPLIT && ncand) ? ncand[q] : kp;
    if (SPLIT && (int)blockIdx.y * KPRIME >= nc) return;
    const uint64_t ck = (SPLIT ? (int)blockIdx.y * KPRIME : 0) + lane < nc ? best[(size_t)q * kp + (SPLIT ? blockIdx.y * KPRIME : 0) + lane] : SC_KEY_MAX;
    rowid[lane] = ck != SC_KEY_MAX ? (uint32_t)ck : 0u;  // padding slots re-score row 0 and are discarded below
    __syncthreads();
    const int seg = lane & 15, r0 = lane >> 4;  // this thread fetches 16-byte piece `seg` of rows r0, r0 + 8, ...
    const float* qv = Qp + (size_t)q * ld;
    f32x4 nxt[16];
    float nq = 0.f;
    auto fetch = [&](int ch) {
#pragma unroll
        for (int i = 0; i < 16; ++i) nxt[i] = *reinterpret_cast<const f32x4*>(X + (size_t)rowid[r0 + 8 * i] * ld + ch * 64 + seg * 4);
        if (lane < 64) nq = qv[ch * 64 + lane];
    };
    fetch(0);
    float acc = 0.f;
    const int nch = ld >> 6;
    for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();  // everyone is done with the previous tile
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4*>(tile + (r0 + 8 * i) * RR_STRIDE + seg * 4) = nxt[i];
        if (lane < 64) qch[lane] = nq;
        __syncthreads();
        if (ch + 1 < nch) fetch(ch + 1);
        const float* xr = tile + lane * RR_STRIDE;
#pragma unroll
        for (int t = 0; t < 64; t += 16) {
            f32x4 xa[4], qa[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                xa[g] = *reinterpret_cast<const f32x4*>(xr + t + 4 * g);
                qa[g] = *reinterpret_cast<const f32x4*>(qch + t + 4 * g);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc = fmaf(xa[g][c], qa[g][c], acc);
        }
    }
    uint64_t ek = SC_KEY_MAX;
    if (ck != SC_KEY_MAX) {
        const uint32_t row = (uint32_t)ck;
        ek = sc_make_key<METRIC>(sc_score<METRIC>(acc, xnorm[row], qnorm[q]), perm ? perm[row] : row);  // ties: reported row id
    }
    if (SPLIT) {
        ekeys[(size_t)q * kp + blockIdx.y * KPRIME + lane] = ek;
        return;
    }
    keys[lane] = ek;
    __syncthreads();
    int rank = 0;
    int have = 0;
    for (int j = 0; j < KPRIME; ++j) {
        const uint64_t o = keys[j];
        rank += (o < ek) ? 1 : 0;
        have += (o != SC_KEY_MAX) ? 1 : 0;
    }
    __syncthreads();
    if (ek != SC_KEY_MAX) keys[rank] = ek;  // unique keys -> a permutation of the first `have` slots
    __syncthreads();
    if (lane < k) {
        const size_t o = (size_t)q * k + lane;
        if (lane < have) {
            out_dist[o] = sc_key_score(METRIC, keys[lane]);
            out_rows[o] = row_base + (int64_t)(uint32_t)keys[lane];
        } else {
            out_dist[o] = (METRIC == SC_METRIC_L2) ? __builtin_inff() : -__builtin_inff();
            out_rows[o] = -1;
        }
    }
    if (lane == 0) {
        int bad = overflow[q];
        if (have == KPRIME) {  // otherwise every row of the corpus is a candidate: nothing can be missing
            const int kk = k < have ? k : have;
            if (!certified<METRIC>(keys[kk - 1], xnorm_max_bits, qnorm[q], qres[q], ld, thr[q])) bad = 1;
        }
        flags[q] = bad;
    }
}

// int8 stage: the KPRIME8 exact keys of one query (scan_rerank_kernel<.., SPLIT>) -> sorted top-k + certificate
template <int METRIC>
__global__ __launch_bounds__(256) void scan_finalize_kernel(const uint64_t* __restrict__ ekeys, int kp, const float* __restrict__ qnorm,
                                                             const float* __restrict__ thr, const unsigned* __restrict__ xnorm_max_bits,
                                                             const float* __restrict__ qres, const int* __restrict__ overflow, int ld, int k,
                                                             int64_t row_base, float* __restrict__ out_dist, int64_t* __restrict__ out_rows,
                                                             int* __restrict__ flags) {
    __shared__ uint64_t keys[KPRIME8], sorted[KPRIME8];
    __shared__ int s_have;
    const int q = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_have = 0;
    for (int i = tid; i < kp; i += 256) {
        keys[i] = ekeys[(size_t)q * kp + i];
        sorted[i] = SC_KEY_MAX;
    }
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < kp; i += 256) {
        const uint64_t key = keys[i];
        if (key == SC_KEY_MAX) continue;
        int rank = 0;
        for (int j = 0; j < kp; ++j) rank += (keys[j] < key) ? 1 : 0;  // exact keys are unique (row id in the low word)
        sorted[rank] = key;
        ++mine;
    }
    if (mine) atomicAdd(&s_have, mine);
    __syncthreads();
    const int have = s_have;
    if (tid < k) {
        const size_t o = (size_t)q * k + tid;
        if (tid < have) {
            out_dist[o] = sc_key_score(METRIC, sorted[tid]);
            out_rows[o] = row_base + (int64_t)(uint32_t)sorted[tid];
        } else {
            out_dist[o] = (METRIC == SC_METRIC_L2) ? __builtin_inff() : -__builtin_inff();
            out_rows[o] = -1;
        }
    }
    if (tid == 0) {
        int bad = overflow[q];
        if (have == kp) {
            const int kk = k < have ? k : have;
            if (!certified<METRIC>(sorted[kk - 1], xnorm_max_bits, qnorm[q], qres[q], ld, thr[q])) bad = 1;
        }
        flags[q] = bad;
    }
}

// ---- the collect pass (second chance of a query whose certificate failed) ------------------------------------------------------------
// The certificate compares the k-th exact score with the kp-th best COARSE score; it fails when more than kp rows are within the
// coarse error of the k-th neighbour (clustered corpora: a whole cluster is).  The failed pass still leaves a true upper bound of
// the k-th score: vk, the k-th exact score among its candidates.  A row can only belong to the result if its exact v-score is
// <= vk, i.e. its coarse v-score <= vk + eps =: T.  The collect pass runs the same coarse kernel once more over all rows with the
// FIXED threshold T (no selection, no shrinking), keeps every row that passes (up to BATCH_CAP per query), re-scores them all exactly
// and takes the exact top-k: correct by construction -- no candidate count to exceed.  Only a query with more than BATCH_CAP rows
// within T goes on to the next stage.  10M x 768 in 4096 clusters of spread 0.1: every query used to end in the exact scan
// (347 ms per 1024-query batch, profiles/r3t_clustered_probe.log); see DESIGN.md section 4 for what it takes now.
template <int METRIC>
__global__ __launch_bounds__(256) void scan_collect_bound_kernel(const float* __restrict__ prev_dist, int k, const float* __restrict__ qnorm,
                                                                  const float* __restrict__ qres, const unsigned* __restrict__ bits, int ld,
                                                                  float* __restrict__ thr, float* __restrict__ thr_fast, int* __restrict__ flags, int Q) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    const float d = prev_dist[(size_t)q * k + (k - 1)];
    const float vk = (METRIC == SC_METRIC_L2) ? d : -d;
    const bool ok = fabsf(vk) < 3.0e38f;  // fewer than k candidates (+-inf) or NaN: nothing to bound with
    float t = -__builtin_inff(), tf = -__builtin_inff();
    if (ok) {
        t = vk + certificate_eps<METRIC>(bits, qnorm[q], qres[q], ld);
        const float slack = 1e-3f * fabsf(t) + 1e-6f;  // the fast test must pass whatever the precise one passes (scan_select_kernel)
        const float qn = qnorm[q];
        if (METRIC == SC_METRIC_L2) tf = (t - qn) + slack + 1e-3f * fabsf(qn);
        else if (METRIC == SC_METRIC_COSINE) tf = (t + slack) * sqrtf(qn) + 1e-5f * sqrtf(qn);
        else tf = t + slack;
    }
    thr[q] = t;
    thr_fast[q] = tf;
    flags[q] = ok ? 0 : 1;
}
// ---- thresholds from EXACT scores before the large phases ---------------------------------------------------------------------------
// Between phases the threshold of a query is its kp-th best coarse key; a phase of 3 r0 new rows then yields 3 kp survivors per
// query (kp = 512: 1 536), each of which costs the threshold epilogue a trip through its precise test -- 1.3 of the 7.9 ms of coarse
// kernels per step, most of it in the last two phases (94 % of the rows).  But a row matters only if its coarse score is within
// eps of the k-th EXACT score, and far fewer than kp rows are (~150 on the Gaussian benchmark: what the certificate needs kp = 512
// for is the worst query, not the typical one).  So before a large phase the 128 best of the kp candidates are re-scored exactly
// (0.4 GB of scattered rows) and the threshold becomes  min(kp-th coarse key, k-th exact score + eps (+ a hair))  -- eps is the
// certificate's own bound, so the cut is one the certificate accepts: a row dropped by it has coarse > v_k + eps, i.e. exact > v_k.
// thr_cut keeps the smallest cut ever applied to a query; the certificate compares with min(final kp-th key, thr_cut).
template <int METRIC>
__global__ __launch_bounds__(128) void scan_tighten_kernel(const uint64_t* __restrict__ ekeys, int kp, int k, const float* __restrict__ qnorm,
                                                            const float* __restrict__ qres, const unsigned* __restrict__ bits, int ld,
                                                            float* __restrict__ thr, float* __restrict__ thr_fast, float* __restrict__ thr_cut) {
    __shared__ uint64_t keys[512];  // kp = 128 (k <= 64) or 512 (k <= 256: every candidate of the int8 stage)
    __shared__ float s_v;
    const int q = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < kp; i += 128) keys[i] = ekeys[(size_t)q * kp + i];
    if (tid == 0) s_v = __builtin_inff();
    __syncthreads();
    for (int i = tid; i < kp; i += 128) {
        const uint64_t key = keys[i];
        if (key == SC_KEY_MAX) continue;
        int rank = 0;
        for (int j = 0; j < kp; ++j) rank += keys[j] < key ? 1 : 0;  // exact keys are unique (row id in the low word)
        if (rank == k - 1) {
            const float sc = sc_key_score(METRIC, key);
            s_v = (METRIC == SC_METRIC_L2) ? sc : -sc;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float t = thr[q];
        const float vk = s_v;
        if (vk < 3.0e38f) {
            const float eps = certificate_eps<METRIC>(bits, qnorm[q], qres[q], ld);
            const float cut = (vk + eps) + fabsf(vk + eps) * 2e-6f + 1e-30f;  // strictly above v_k + eps: the certificate's "<" must hold when v_k does not improve
            t = fminf(t, cut);
        }
        t = fminf(t, thr_cut[q]);
        thr_cut[q] = t;
        thr[q] = t;
        float tf = __builtin_inff();
        if (t < __builtin_inff()) {  // scan_select_kernel's fast-test form of the same threshold
            const float slack = 1e-3f * fabsf(t) + 1e-6f;
            const float qn = qnorm[q];
            if (METRIC == SC_METRIC_L2) tf = (t - qn) + slack + 1e-3f * fabsf(qn);
            else if (METRIC == SC_METRIC_COSINE) tf = (t + slack) * sqrtf(qn) + 1e-5f * sqrtf(qn);
            else tf = t + slack;
        }
        thr_fast[q] = tf;
    }
}
// ---- the wide candidate set (corpora whose certificate fails at kp candidates: clusters) ----------------------------------------------
// With thresholds from exact scores the selection between phases need not truncate at kp: it keeps EVERY key within the cut (up to
// kcap = 4 096) and at least the kp best.  Nothing within the coarse error of the k-th exact score is ever dropped then, so the
// certificate holds by construction and the batch is answered in this one pass -- where the kp-candidate form sent a clustered
// batch through a second pass (the collect pass) or to the bf16 stage.  best [Q][kcap] holds nbest[q] keys (no padding).
template <int METRIC>
__global__ __launch_bounds__(SEL_THREADS) void scan_select_wide_kernel(const uint64_t* __restrict__ surv, unsigned* __restrict__ count, int cap,
                                                                        uint64_t* __restrict__ best, unsigned* __restrict__ nbest, int kcap, int kp,
                                                                        const float* __restrict__ qnorm, float* __restrict__ thr, float* __restrict__ thr_fast,
                                                                        const float* __restrict__ thr_cut, int* __restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sel_lds[];  // [cap + kcap] candidates | hist[SEL_BINS] | scan[SEL_THREADS] | misc
    uint64_t* cand = sel_lds;
    unsigned* hist = reinterpret_cast<unsigned*>(sel_lds + cap + kcap);
    unsigned* part = hist + SEL_BINS;
    unsigned* misc = part + SEL_THREADS;  // [0] n, [1] chosen bin, [2] rank inside it, [3] output cursor, [4] keys within the cut
    const int q = blockIdx.x, tid = threadIdx.x;
    unsigned c = count[q];
    if (c > (unsigned)cap) {
        if (tid == 0) overflow[q] = 1;
        c = cap;
    }
    const unsigned nb = nbest[q];
    const float cutv = thr_cut[q];
    const bool have_cut = cutv < __builtin_inff();
    if (tid == 0) { misc[0] = 0; misc[3] = 0; misc[4] = 0; }
    __syncthreads();
    for (unsigned i = tid; i < c + nb; i += SEL_THREADS) {
        const uint64_t key = i < c ? surv[(size_t)q * cap + i] : best[(size_t)q * kcap + (i - c)];
        if (key == SC_KEY_MAX) continue;
        cand[atomicAdd(&misc[0], 1u)] = key;
        if (have_cut) {
            const float sc = sc_key_score(METRIC, key);
            if (((METRIC == SC_METRIC_L2) ? sc : -sc) <= cutv) atomicAdd(&misc[4], 1u);
        }
    }
    __syncthreads();
    const int n = (int)misc[0];
    const int cle = (int)misc[4];
    if (cle > kcap && tid == 0) overflow[q] = 1;  // more keys within the cut than the set holds: the query goes on to the next stage
    const int keep = (n < kp ? n : kp) > (cle < kcap ? cle : kcap) ? (n < kp ? n : kp) : (cle < kcap ? cle : kcap);
    uint64_t pivot = SC_KEY_MAX;  // keep every key <= pivot
    if (n > keep) {
        uint64_t prefix = 0;
        unsigned want = (unsigned)(keep - 1);
        for (int shift = 52; shift >= -8; shift -= 12) {
            const int sh = shift < 0 ? 0 : shift;
            const int bits = shift < 0 ? 4 : 12;
            const unsigned mask = (1u << bits) - 1u;
            for (int i = tid; i < SEL_BINS; i += SEL_THREADS) hist[i] = 0;
            __syncthreads();
            const int hi_shift = sh + bits;
            for (int i = tid; i < n; i += SEL_THREADS) {
                const uint64_t key = cand[i];
                if (hi_shift >= 64 || (key >> hi_shift) == prefix) atomicAdd(&hist[(unsigned)(key >> sh) & mask], 1u);
            }
            __syncthreads();
            unsigned local = 0;
            for (int j = 0; j < SEL_BINS / SEL_THREADS; ++j) local += hist[tid * (SEL_BINS / SEL_THREADS) + j];
            unsigned incl = local;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl_up(incl, off, 64);
                if ((tid & 63) >= off) incl += o;
            }
            if ((tid & 63) == 63) part[tid >> 6] = incl;
            __syncthreads();
            unsigned pre = incl - local;
            for (int w = 0; w < (tid >> 6); ++w) pre += part[w];
            if (pre <= want && want < pre + local) {
                unsigned acc = pre;
                int bin = tid * (SEL_BINS / SEL_THREADS);
                for (;; ++bin) {
                    if (acc + hist[bin] > want) break;
                    acc += hist[bin];
                }
                misc[1] = (unsigned)bin;
                misc[2] = want - acc;
            }
            __syncthreads();
            prefix = (prefix << bits) | (uint64_t)misc[1];
            want = misc[2];
            __syncthreads();
        }
        pivot = prefix;
    }
    for (int i = tid; i < n; i += SEL_THREADS) {
        const uint64_t key = cand[i];
        if (key <= pivot) best[(size_t)q * kcap + atomicAdd(&misc[3], 1u)] = key;
    }
    __syncthreads();
    if (tid == 0) {
        nbest[q] = misc[3];  // == keep
        count[q] = 0;
        // what the rows dropped here (and by the phase's own test) exceed: the cut when every key within it was kept, else the kp-th key
        float t = __builtin_inff();
        if (have_cut && cle >= kp && cle <= kcap) t = cutv;
        else if (n > keep && keep >= kp) {
            const float sc = sc_key_score(METRIC, pivot);
            t = (METRIC == SC_METRIC_L2) ? sc : -sc;
        } else if (n == keep && keep >= kp && n == kp) {  // exactly kp keys: all stay, the threshold is their maximum (as scan_select_kernel has it)
            uint64_t m = 0;
            for (int i = 0; i < n; ++i) m = cand[i] > m ? cand[i] : m;
            const float sc = sc_key_score(METRIC, m);
            t = (METRIC == SC_METRIC_L2) ? sc : -sc;
        }
        float tf = __builtin_inff();
        if (t < __builtin_inff()) {
            const float slack = 1e-3f * fabsf(t) + 1e-6f;
            const float qn = qnorm[q];
            if (METRIC == SC_METRIC_L2) tf = (t - qn) + slack + 1e-3f * fabsf(qn);
            else if (METRIC == SC_METRIC_COSINE) tf = (t + slack) * sqrtf(qn) + 1e-5f * sqrtf(qn);
            else tf = t + slack;
        }
        thr[q] = t;
        thr_fast[q] = tf;
    }
}
// the keys of best within the final threshold, compacted: what the exact re-score has to look at
template <int METRIC>
__global__ __launch_bounds__(256) void scan_wide_compact_kernel(const uint64_t* __restrict__ best, const unsigned* __restrict__ nbest, int kcap,
                                                                 const float* __restrict__ thr, uint64_t* __restrict__ cand, int* __restrict__ ncand) {
    __shared__ unsigned s_n;
    const int q = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const unsigned nb = nbest[q];
    const float tau = thr[q];
    for (unsigned i = tid; i < nb; i += 256) {
        const uint64_t key = best[(size_t)q * kcap + i];
        if (key == SC_KEY_MAX) continue;  // (the plain form pads its kp slots)
        const float sc = sc_key_score(METRIC, key);
        if (((METRIC == SC_METRIC_L2) ? sc : -sc) <= tau) cand[(size_t)q * kcap + atomicAdd(&s_n, 1u)] = key;
    }
    __syncthreads();
    if (tid == 0) ncand[q] = (int)s_n;
}
// the certificate of the wide form: the k-th exact score (out_dist, already final) + eps below the threshold, or nothing was ever dropped
template <int METRIC>
__global__ __launch_bounds__(256) void scan_wide_certify_kernel(const float* __restrict__ out_dist, int k, const float* __restrict__ qnorm, const float* __restrict__ qres,
                                                                 const unsigned* __restrict__ bits, int ld, const float* __restrict__ thr,
                                                                 const int* __restrict__ overflow, int* __restrict__ flags, int Q) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    int bad = overflow[q];
    const float tau = thr[q];
    if (tau < __builtin_inff()) {
        const float d = out_dist[(size_t)q * k + (k - 1)];
        const float vk = (METRIC == SC_METRIC_L2) ? d : -d;
        if (!(vk + certificate_eps<METRIC>(bits, qnorm[q], qres[q], ld) < tau)) bad = 1;
    }
    flags[q] = bad;
}

// thr[q] = min(thr[q], thr_cut[q]) before the certificate (the last selection may have left +inf: fewer than kp keys)
__global__ __launch_bounds__(256) void scan_thr_min_kernel(float* __restrict__ thr, const float* __restrict__ thr_cut, int Q) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < Q) thr[q] = fminf(thr[q], thr_cut[q]);
}
__global__ __launch_bounds__(256) void scan_fill_u32_kernel(unsigned* __restrict__ p, unsigned v, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ __launch_bounds__(256) void scan_collect_counts_kernel(const unsigned* __restrict__ count, int cap, int* __restrict__ ncand, int* __restrict__ flags, int Q) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    const unsigned c = count[q];
    const bool fits = c <= (unsigned)cap && !flags[q];
    ncand[q] = fits ? (int)c : 0;
    if (!fits) flags[q] = 1;
}

__global__ __launch_bounds__(256) void scan_batched_init_kernel(float* thr, float* thr_fast, int qpad, uint64_t* best, unsigned* count,
                                                                 int* overflow, int Q, int kp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < qpad) { thr[i] = __builtin_inff(); thr_fast[i] = __builtin_inff(); }
    if (i < Q) { count[i] = 0u; overflow[i] = 0; }
    if (i < Q * kp) best[i] = SC_KEY_MAX;
}

// ------------------------------------------------------------------ launchers
void sc_launch_scan_batched_init(float* thr, float* thr_fast, int qpad, uint64_t* best, unsigned* count, int* overflow, int Q, int kp, hipStream_t s) {
    const int n = Q * kp > qpad ? Q * kp : qpad;
    hipLaunchKernelGGL(scan_batched_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, thr, thr_fast, qpad, best, count, overflow, Q, kp);
}
void sc_launch_shadow(const float* X, const float* xnorm, int64_t first, int64_t n, int ld, void* Xb, unsigned* res_bits, hipStream_t s) {
    if (n <= 0) return;
    int64_t blocks = (n + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(shadow_kernel, dim3((unsigned)blocks), dim3(256), 0, s, X, xnorm, first, n, ld, (bf16_t*)Xb, res_bits);
}
void sc_launch_shadow8(const float* X, const float* xnorm, int64_t first, int64_t n, int ld, int ld8, void* Xq, float* xscale, unsigned* res_bits,
                       hipStream_t s) {
    if (n <= 0) return;
    int64_t blocks = (n + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(shadow8_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, X, xnorm, first, n, n, ld, ld8, (int8_t*)Xq, xscale, res_bits, (float*)nullptr,
                       (const unsigned*)nullptr);
}
void sc_launch_norm_max(const float* xnorm, int64_t n, unsigned* out_bits, hipStream_t s) {
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(norm_max_kernel, dim3((unsigned)blocks), dim3(256), 0, s, xnorm, n, out_bits);
}
void sc_launch_query_bf16(const float* Qp, int Q, int Qpad, int ld, void* Qb, float* qres, hipStream_t s) {
    hipLaunchKernelGGL(query_bf16_kernel, dim3((unsigned)((Qpad + 3) / 4)), dim3(256), 0, s, Qp, Q, Qpad, ld, (bf16_t*)Qb, qres);
}
// f32 padded queries [Q, ld] -> int8 [Qpad, ld8] (rows >= Q zero) with ONE scale s = max |q_i| / 127 for the whole batch
// (qscale [Qpad] all equal), qres[q] = |q - s q_q|^2; absmax_bits: 4 bytes of device scratch
void sc_launch_query_i8(const float* Qp, int Q, int Qpad, int ld, int ld8, void* Qq, float* qscale, float* qres, unsigned* absmax_bits, hipStream_t s) {
    hipMemsetAsync(absmax_bits, 0, 4, s);
    const int64_t n = (int64_t)Q * ld;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0, s, Qp, n, absmax_bits);
    hipLaunchKernelGGL(shadow8_kernel<true>, dim3((unsigned)((Qpad + 3) / 4)), dim3(256), 0, s, Qp, (const float*)nullptr, (int64_t)0, (int64_t)Q, (int64_t)Qpad, ld,
                       ld8, (int8_t*)Qq, qscale, (unsigned*)nullptr, qres, (const unsigned*)absmax_bits);
}

int sc_batched_kprime(void) { return KPRIME; }
int sc_batched_kprime8(void) { return KPRIME8; }

// ---- persistent form: one workgroup per CU walks its share of the tiles, and the LDS ring never drains between them: the last
// phases of a tile request the first two K-tiles of the NEXT tile into the slots that fall free (gemm_tile.h, PPNextTileHook), so
// a tile's first bytes (HBM latency: the corpus rows are read once) and a third of its fill travel under the previous tile's
// tail and threshold epilogue.  A tile is 6 (int8) or 12 (bf16) K-tiles at 768 dimensions: with one workgroup per launch slot the
// stamps read entry -> main loop done 11.6 us for 6.1 us of MFMAs, epilogue 4.2, 0.5 to the next workgroup
// (profiles/r3g_coarse_trace.log).  Tiles: XCD x owns a contiguous range of logical tiles (as xcd_remap gives it); its G / 8
// workgroups take consecutive tiles of it per round, i.e. one group of 8 row panels x 4 query tiles runs on one XCD at a time, as
// under hardware dispatch.  The per-tile thresholds / norms in LDS are double buffered (a wave may be a whole epilogue ahead).
static int g_coarse_wgs = 0, g_coarse_persistent = 1;  // sc_diag_set_option
void sc_scan_set_coarse_workgroups(int v) { g_coarse_wgs = v; }
void sc_scan_set_coarse_persistent(int v) { g_coarse_persistent = v; }
#define COARSE_QLDS_BYTES (8 * 256 * 4)
#define COARSEP_LDS_BYTES (4 * T_TILE_BYTES + 2 * COARSE_QLDS_BYTES)
template <int METRIC, bool I8, bool TRACE = false>
__global__ __launch_bounds__(512) void scan_coarse256p_kernel(CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // this workgroup's tiles: lo + idx, lo + idx + per_round, ... below hi
    const int G = (int)gridDim.x, x = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3, per_round = G >> 3;
    const int q8 = a.ntiles >> 3, r8 = a.ntiles & 7;
    const int lo = x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8, hi = lo + (x < r8 ? q8 + 1 : q8);
    int tile = lo + idx;
    if (tile >= hi) return;
    const int nk = a.ld / G_BK;
    uint32_t va[2], vw[2];
    pp_piece_offsets(a.ld, a.ld, w, lane, va, vw);
    PPNextTileHook hook;
    hook.a_kbytes = (uint32_t)(G_BK * 2); hook.a1_off = (uint32_t)(64 * a.ld * 2); hook.w1_off = (uint32_t)(32 * a.ld * 2);
    hook.w = w; hook.smem = smem;
    int64_t m0;
    int n0;
    coarse256_coords(a, tile, m0, n0);
    hook.ra = __builtin_amdgcn_make_buffer_rsrc((void*)(a.Xb + m0 * a.ld), 0, -1, 0x00020000);
    hook.rw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.Qb + (size_t)n0 * a.ld), 0, -1, 0x00020000);
#pragma unroll
    for (int h = 0; h < 8; ++h) hook.coop(h, h, va, vw);  // the first tile's first two K-tiles, in ring order from parity 0
    int par = 0, it = 0;
    // Staging without a global round trip in front of every tile: the row side (|x|^2 and the int8 row scale of the tile's 256 rows,
    // threads 256..511) is loaded one tile ahead into two registers and only WRITTEN to LDS here (the compiler guards that write with
    // s_waitcnt vmcnt(0) in waves 4-7, which also waits for the prefetched half-tiles; an LDS-DMA form of this staging without that
    // wait was built and measured: no gain -- 6.98 -> 6.92 ms of kernels with the epilogue switched off, slower with it); the query side (thresholds, norms,
    // scales of the 256 queries) is staged again only when the query tile changes -- a workgroup's tiles are one round (a
    // multiple of the group of 8 row panels x all query tiles, at 1 024 queries and 256 CUs) apart, so it never does there.
    float* q_cur = reinterpret_cast<float*>(smem + COARSE_QLDS);
    float xn_next = 1.0f, xs_next = 0.0f;
    const float sq0_batch = I8 ? a.qscale[0] : 1.0f;  // every query of a batch shares one scale (sc_launch_query_i8)
    if (tid >= 256) {
        const int64_t row = m0 + (tid - 256);
        xn_next = row < a.row1 ? a.xnorm[row] : 1.0f;
        if (I8) xs_next = row < a.row1 ? a.xscale[row] : 0.0f;
    }
    int n0_even = -1, n0_odd = -1;  // query tile staged in either copy
#pragma unroll 1
    for (;; ++it) {
        char* smem_q = smem + (it & 1) * COARSE_QLDS_BYTES;  // this tile's thresholds / norms (the other copy may still be read)
        q_cur = reinterpret_cast<float*>(smem_q + COARSE_QLDS);
        if (tid >= 256) {
            q_cur[768 + tid - 256] = xn_next;
            if (I8) {
                q_cur[1280 + tid - 256] = xs_next;
                float A, B;
                coarse_row_bound<METRIC>(xn_next, xs_next, sq0_batch, A, B);
                *reinterpret_cast<f32x2*>(q_cur + 1536 + 2 * (tid - 256)) = f32x2{A, B};
            }
        }
        if (((it & 1) ? n0_odd : n0_even) != n0) {
            coarse256_stage<I8, false, true>(a, m0, n0, smem_q, tid);
            if (it & 1) n0_odd = n0;
            else n0_even = n0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int next = tile + per_round;
        const bool more = next < hi;
        int64_t m0n = m0;
        int n0n = n0;
        if (more) coarse256_coords(a, next, m0n, n0n);  // the last tile "prefetches" itself: the request counts of the loop stay what they are
        if (more && tid >= 256) {  // the next tile's row side: in flight under this tile's main loop
            const int64_t row = m0n + (tid - 256);
            xn_next = row < a.row1 ? a.xnorm[row] : 1.0f;
            if (I8) xs_next = row < a.row1 ? a.xscale[row] : 0.0f;
        }
        hook.ra = __builtin_amdgcn_make_buffer_rsrc((void*)(a.Xb + m0n * a.ld), 0, -1, 0x00020000);
        hook.rw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.Qb + (size_t)n0n * a.ld), 0, -1, 0x00020000);
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (TRACE && tid == 0) {
            a.trace[(size_t)tile * 8 + 0] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
            a.trace[(size_t)tile * 8 + 1] = (unsigned long long)wall_clock64();
        }
        gemm_tile256_mainloop_pp<4, 0, PPNextTileHook, I8, true>(a.Xb + m0 * a.ld, a.ld, 0, a.Qb, a.ld, n0, a.ld, smem, acc, w, lane, hook, G_BK, par);
        par ^= nk & 1;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (TRACE && tid == 0) a.trace[(size_t)tile * 8 + 2] = (unsigned long long)wall_clock64();
        if (a.cap != -1) coarse256_epilogue<METRIC, I8>(a, acc, m0, n0, smem_q, w, lane);  // cap -1 / -2: SC_COARSE_EXPERIMENT (results invalid)
        if (TRACE && tid == 0) a.trace[(size_t)tile * 8 + 3] = (unsigned long long)wall_clock64();
        if (!more) break;
        tile = next;
        m0 = m0n;
        n0 = n0n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last tile's self-prefetch must not outlive the workgroup's LDS
}

// ---- narrow streaming form for small batches (17 .. 64 queries, int8 shadow) -----------------------------------------------
// A 256 x 256 tile spends 6.1 us of MFMAs per 256 corpus rows whatever the batch holds; below ~100 queries the bound is not the
// matrix pipe but the memory system: the 7.7 GB int8 shadow streams in 1.2 ms.  This kernel is shaped for that: a persistent
// workgroup (8 waves) walks row tiles of 256 rows x 64 query slots and sees the corpus as ONE stream of stages (tile, K-tile) --
// 32 KiB of rows + 8 KiB of queries each -- through a three-deep LDS ring: two stages are always in flight, across tile
// boundaries too, so neither the first bytes of a tile nor the threshold epilogue interrupts the stream.  Wave w multiplies rows
// 32 w .. 32 w + 31 by all 64 slots (16 MFMAs per stage: the pipe idles, by design).  One barrier per stage.  Survivors (rare
// in the phases this kernel serves) are appended to a per-wave list in global memory -- no atomics, nothing to wait for -- and a
// small kernel scatters the lists into the per-query survivor lists afterwards.
// 10M x 768, 32 queries: 2.6 ms through the 256-query tiles -> see profiles/r3*_q_sweep.log.
#define C64_STAGE_A (256 * 128)
#define C64_STAGE_W (64 * 128)
#define C64_STAGE (C64_STAGE_A + C64_STAGE_W)
#define C64_RING (3 * C64_STAGE)
#define C64_LDS_BYTES (C64_RING + 3 * 1024 * 4 + 1024 * 4 + 3 * 512 * 4 + 4 * 64 * 4)
// GROUPED (IVF_FLAT coarse stage, ivf_coarse.hip): a work item is one 256-row tile of a LIST PART against one group of up to 64
// query SLOTS (the (query, list) pairs that probe the list): rows from the centred int8 shadow, slots from the per-pair centred
// queries, thresholds / norms / scales / query ids per slot.  Same stream of stages, same tests; a hit names the slot's query.
struct GroupItem {
    long long row0;  // first stored position of the tile
    int rows;        // valid rows (<= 256)
    int slot_base;   // first of the group's 64 slots
};
struct GroupedArgs {
    const GroupItem* items;
    int nitems;
    const float *slot_tf, *slot_thr, *slot_qn, *slot_qs;
    const int32_t* slot_q;
    const float *slot_qb, *slot_qd;  // |q'|, |q' - qq|: the pair factors of the row-wise error bound
    const f32x4* xrow;               // per row {|x'|^2, scale, 2 |dx|, 2 (|x'| + |dx|)}
    const int32_t* slot_dst;         // DENSE: survivor-list position of the slot's list row r = (uint32)(r + slot_dst) (mod 2^32)
};
// DENSE (GROUPED; phase A of the IVF coarse stage): the thresholds are still +inf, EVERY row of the list survives -- so a row's place in
// the query's survivor list is known in advance (its offset in the list + what the query's earlier lists hold) and the key goes
// straight there: no test, no hit list, no atomics (the hit lists + their scatter cost 1.3 ms of a config-5 batch, all of it phase A).
template <int METRIC, bool GROUPED = false, bool DENSE = false>
__global__ __launch_bounds__(512) void scan_coarse64s_kernel(CoarseArgs a, u32x4_t* __restrict__ hitlist, unsigned* __restrict__ hitcount, int hitcap, GroupedArgs ga) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld8 = a.ld * 2;  // bytes per int8 row (a.ld counts 2-byte elements)
    const int nk = ld8 >> 7;   // stages per tile
    const int64_t ntiles = GROUPED ? (int64_t)ga.nitems : (a.row1 - a.row0 + 255) >> 8;
    const int64_t first = blockIdx.x;
    if (first >= ntiles) {
        if (tid < 8) hitcount[(size_t)blockIdx.x * 8 + tid] = 0;
        return;
    }
    const int64_t mine = (ntiles - first + gridDim.x - 1) / gridDim.x, S = mine * nk;
    // [3][xnorm 256 | xscale 256]: a tile's row side is requested two stages ahead, i.e. (one stage per tile, ld8 = 128) while the
    // epilogue of the tile two before it is still reading its copy -- three copies; the slot side (GROUPED) likewise
    // GROUPED: the row side is one 16-byte record per row ([3][256] f32x4 + 4 KiB that take the second copy waves 4-7 request so that
    // every wave issues the same number of pieces)
    float* rowlds = reinterpret_cast<float*>(smem + C64_RING);
    float* slotlds = rowlds + 3 * 1024 + 1024;  // GROUPED: [3][8][64]: tf | thr | qn | qs | query id | |q'| | |dq| | (unused)
    float* qlds = slotlds + 3 * 512;            // flat: thr_fast[64] | thr[64] | qnorm[64] | qscale[64]
    if (!GROUPED && tid < 64) {
        const bool real = tid < a.Q;
        qlds[tid] = real ? a.thr_fast[tid] : -__builtin_inff();
        qlds[64 + tid] = real ? a.thr[tid] : -__builtin_inff();
        qlds[128 + tid] = real ? a.qnorm[tid] : 1.0f;
        qlds[192 + tid] = a.qscale[tid];  // padded to Qpad
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // per-lane source offsets: a piece = 8 rows x 128 B; the 16-byte chunk of row r at position pos comes from chunk pos ^ ((r >> 1) & 7)
    uint32_t va[4], vw;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (4 * w + i) * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
        va[i] = (uint32_t)(r * ld8 + c * 16);
    }
    {
        const int r = w * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
        vw = (uint32_t)(r * ld8 + c * 16);
    }
    // row side: |x|^2 (waves 0-3) and the int8 row scale (waves 4-7), 64 rows per wave and tile; rows beyond the tile's end read as 0
    // (the descriptor ends there)
    const float* rowsrc = w < 4 ? a.xnorm : a.xscale;
    const uint32_t vrow = (uint32_t)(((w & 3) * 64 + lane) * 4);
    const float* slotsrc = nullptr;  // GROUPED: wave w requests slot array w % 5 of the item (256 B)
    if (GROUPED)
        slotsrc = w == 0 ? ga.slot_tf : w == 1 ? ga.slot_thr : w == 2 ? ga.slot_qn : w == 3 ? ga.slot_qs : w == 4 ? reinterpret_cast<const float*>(ga.slot_q)
                  : w == 5 ? ga.slot_qb : w == 6 ? ga.slot_qd : reinterpret_cast<const float*>(ga.slot_dst);

    int64_t t_i = first;  // issue side: tile and K-tile of the next stage to request
    int kt_i = 0, j_i = 0;
    int64_t m0_i = 0;
    int rows_i = 256, sb_i = 0;
    auto issue = [&](int slot) {
        char* dst = smem + slot * C64_STAGE;
        if (kt_i == 0) {
            if (GROUPED) {
                const GroupItem it = ga.items[t_i];
                m0_i = it.row0;
                rows_i = it.rows;
                sb_i = it.slot_base;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(slotsrc + sb_i), 0, 256, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(reinterpret_cast<char*>(slotlds) + ((j_i % 3) * 512 + w * 64) * 4), 4, (uint32_t)(lane * 4), 0, 0, 0);
            } else {
                m0_i = a.row0 + (t_i << 8);
                const int64_t left = a.row1 - m0_i;
                rows_i = (int)(left >= 256 ? 256 : left > 0 ? left : 0);
            }
            if (GROUPED) {  // 64 row records of 16 B per wave (waves 4-7: the same rows again, into the spare 4 KiB)
                const __amdgpu_buffer_rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc((void*)(ga.xrow + m0_i), 0, rows_i * 16, 0x00020000);
                char* rdst = reinterpret_cast<char*>(rowlds) + (w < 4 ? (j_i % 3) * 4096 : 3 * 4096) + (w & 3) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rrow, (lds_vptr)rdst, 16, (uint32_t)(((w & 3) * 64 + lane) * 16), 0, 0, 0);
            } else {
                const __amdgpu_buffer_rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc((void*)(rowsrc + m0_i), 0, rows_i * 4, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rrow, (lds_vptr)(reinterpret_cast<char*>(rowlds) + ((j_i % 3) * 512 + (w >> 2) * 256 + (w & 3) * 64) * 4), 4, vrow, 0, 0, 0);
            }
        }
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(a.Xb) + m0_i * (int64_t)ld8), 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(a.Qb) + (int64_t)sb_i * ld8), 0, -1, 0x00020000);
        const uint32_t so = (uint32_t)kt_i * 128u;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_vptr)(dst + (4 * w + i) * 1024), 16, va[i], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_vptr)(dst + C64_STAGE_A + w * 1024), 16, vw, so, 0, 0);
        if (++kt_i == nk) {
            kt_i = 0;
            t_i += gridDim.x;
            ++j_i;
        }
    };
    issue(0);
    if (S > 1) issue(1);

    const int fr = lane & 15, fq = lane >> 4;
    const int sw = (fr >> 1) & 7;
    // fragment addresses inside a stage (k-step 0; k-step 1 = ^ 64): rows 32 w + 16 mi + fr of A, 16 ni + fr of W
    const uint32_t a_rd = (uint32_t)((32 * w + fr) * 128 + ((fq ^ sw) << 4));
    const uint32_t w_rd = (uint32_t)(C64_STAGE_A + fr * 128 + ((fq ^ sw) << 4));
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    u32x4_t* hlist = hitlist + ((size_t)blockIdx.x * 8 + w) * (size_t)hitcap;
    int hn = 0;
    int64_t t_c = first;
    int kt_c = 0, j_c = 0;
    constexpr int P0 = GROUPED ? 7 : 6;  // pieces of a stage that opens a tile: 5 + the row side (+ the slot side)
#pragma unroll 1
    for (int64_t s = 0; s < S; ++s) {
        // stage s has landed (this wave's pieces); the stage behind it may fly (5 pieces, P0 when it opens a tile)
        if (s + 1 >= S) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (kt_c + 1 == nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P0) : "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");  // ... and everyone's; every wave is done with stage s - 1, whose slot stage s + 2 takes
        if (s + 2 < S) issue((int)((s + 2) % 3));
        const char* st = smem + (int)(s % 3) * C64_STAGE;
        bf16x8 af[2][2], wf[4][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) af[mi][ks] = *reinterpret_cast<const bf16x8*>(st + ((a_rd ^ (uint32_t)(ks << 6)) + mi * 2048));
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) wf[ni][ks] = *reinterpret_cast<const bf16x8*>(st + ((w_rd ^ (uint32_t)(ks << 6)) + ni * 2048));
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = pp_mma<true>(wf[ni][ks], af[mi][ks], acc[ni][mi]);
        if (kt_c + 1 < nk) {
            ++kt_c;
            continue;
        }
        // ---- the tile is complete: thresholds (the tests of coarse256_epilogue), survivors appended to this wave's list
        int64_t m0;
        int rows_c;
        if (GROUPED) {
            const GroupItem it = ga.items[t_c];
            m0 = it.row0;
            rows_c = it.rows;
        } else {
            m0 = a.row0 + (t_c << 8);
            const int64_t left = a.row1 - m0;
            rows_c = (int)(left >= 256 ? 256 : left);
        }
        const float* rs = rowlds + (j_c % 3) * (GROUPED ? 1024 : 512);
        const float* sl = GROUPED ? slotlds + (j_c % 3) * 512 : qlds;  // tf | thr | qn | qs (| query id | |q'| | |dq|)
        f32x4 tf[4], sq[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            tf[ni] = *reinterpret_cast<const f32x4*>(sl + ni * 16 + 4 * fq);
            sq[ni] = *reinterpret_cast<const f32x4*>(sl + 192 + ni * 16 + 4 * fq);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int rl = 32 * w + 16 * mi + fr;
            const int64_t row = m0 + rl;
            const bool rowok = rl < rows_c;
            float xn, sx, ea = 0.f, ec = 0.f;  // GROUPED: ea = 2 |dx_r|, ec = 2 (|x'_r| + |dx_r|)
            if (GROUPED) {
                const f32x4 rec = *reinterpret_cast<const f32x4*>(rs + 4 * rl);
                xn = rec[0]; sx = rec[1]; ea = rec[2]; ec = rec[3];
            } else {
                xn = rs[rl];
                sx = rs[256 + rl];
            }
            const float xs = (METRIC == SC_METRIC_COSINE) ? 1.0f / sqrtf(xn) : 0.f;
            const float ar = (METRIC == SC_METRIC_L2) ? -2.0f * sx : (METRIC == SC_METRIC_COSINE) ? -sx * xs : -sx;
            if (GROUPED && DENSE) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ql = 16 * ni + 4 * fq + r;
                        const int qid = __float_as_int(sl[256 + ql]);
                        if (rowok && qid >= 0) {
                            const float dotv = (float)__float_as_int(acc[ni][mi][r]) * (sx * sq[ni][r]);
                            float sc;  // L2: lower bound of the distance; IP: upper bound of the inner product (slot constant = -(<c, q> + allowance))
                            if (METRIC == SC_METRIC_L2) sc = fmaf(-ec, sl[384 + ql], fmaf(-ea, sl[320 + ql], sc_score<METRIC>(dotv, xn, sl[128 + ql])));
                            else sc = fmaf(ec, sl[384 + ql], fmaf(ea, sl[320 + ql], dotv - sl[128 + ql]));
                            const uint32_t pos = (uint32_t)row + (uint32_t)__float_as_int(sl[448 + ql]);
                            if (pos < (uint32_t)a.cap) a.surv[(size_t)qid * a.cap + pos] = sc_make_key<METRIC>(sc, (uint32_t)row);
                        }
                    }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            // (the row factor as a real register pair: -s_r taken straight from the odd half of the row record came out of hipcc as
            // v_pk_mul_f32 ... op_sel:[1,0], the encoding tests/test_isa.py keeps out of every kernel)
            f32x2 ar2 = {ar, ar};
            asm volatile("" : "+v"(ar2));
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                f32x4 t;
                bool g = false;
                const f32x2 av01 = f32x2{(float)__float_as_int(acc[ni][mi][0]), (float)__float_as_int(acc[ni][mi][1])} * ar2;
                const f32x2 av23 = f32x2{(float)__float_as_int(acc[ni][mi][2]), (float)__float_as_int(acc[ni][mi][3])} * ar2;
                const f32x4 av4 = {av01[0], av01[1], av23[0], av23[1]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float av = av4[r];
                    t[r] = (METRIC == SC_METRIC_L2) ? fmaf(av, sq[ni][r], xn) : av * sq[ni][r];
                    g |= t[r] <= tf[ni][r];
                }
                if (!__any(g && rowok)) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bool h = false;
                    uint64_t key = 0;
                    const int ql = 16 * ni + 4 * fq + r;
                    if (rowok && t[r] <= tf[ni][r]) {
                        const float dotv = (float)__float_as_int(acc[ni][mi][r]) * (sx * sq[ni][r]);
                        float sc = sc_score<METRIC>(dotv, xn, sl[128 + ql]);
                        if (GROUPED) {  // the row's own error bound: a lower bound of the exact distance (L2) / an upper bound of the inner product (IP)
                            if (METRIC == SC_METRIC_L2) sc = fmaf(-ec, sl[384 + ql], fmaf(-ea, sl[320 + ql], sc));
                            else sc = fmaf(ec, sl[384 + ql], fmaf(ea, sl[320 + ql], dotv - sl[128 + ql]));
                        }
                        const float v = (METRIC == SC_METRIC_L2) ? sc : -sc;
                        if (v <= sl[64 + ql]) {  // -inf for padded slots
                            h = true;
                            key = sc_make_key<METRIC>(sc, (uint32_t)row);
                        }
                    }
                    const uint64_t m = __ballot(h);
                    if (m) {
                        const int off = hn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (h) {
                            const uint32_t qid = GROUPED ? (uint32_t)__float_as_int(sl[256 + ql]) : (uint32_t)ql;
                            if (off < hitcap) {
                                hlist[off] = u32x4_t{(uint32_t)key, (uint32_t)(key >> 32), qid, 0u};
                            } else {  // list full: allocate the slot here
                                const unsigned pos = atomicAdd(a.count + qid, 1u);
                                if (pos < (unsigned)a.cap) a.surv[(size_t)qid * a.cap + pos] = key;
                            }
                        }
                        hn += (int)__popcll(m);
                    }
                }
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        hn = __builtin_amdgcn_readfirstlane(hn);
        kt_c = 0;
        t_c += gridDim.x;
        ++j_c;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) hitcount[(size_t)blockIdx.x * 8 + w] = (unsigned)(hn < hitcap ? hn : hitcap);
}

// the per-wave hit lists of scan_coarse64s_kernel -> the per-query survivor lists (one wave per list)
__global__ __launch_bounds__(64) void scan_hits_scatter_kernel(const u32x4_t* __restrict__ hitlist, const unsigned* __restrict__ hitcount, int hitcap,
                                                               unsigned* __restrict__ count, uint64_t* __restrict__ surv, int cap) {
    const unsigned n = hitcount[blockIdx.x];
    const u32x4_t* l = hitlist + (size_t)blockIdx.x * (size_t)hitcap;
    for (unsigned i = threadIdx.x; i < n; i += 64) {
        const u32x4_t e = l[i];
        const unsigned pos = atomicAdd(count + e[2], 1u);
        if (pos < (unsigned)cap) surv[(size_t)e[2] * cap + pos] = ((uint64_t)e[1] << 32) | e[0];
    }
}

template <int METRIC>
static void launch_coarse64s(const CoarseArgs& a, hipStream_t s, void* hit_scratch, size_t hit_bytes) {
    static ScDeviceOnce once;
    sc_device_once(once, [&] { hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse64s_kernel<METRIC, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C64_LDS_BYTES); });
    const int64_t ntiles = (a.row1 - a.row0 + 255) >> 8;
    const int cus = sc_device_cus();
    const int wgs = (int)std::min<int64_t>(ntiles, g_coarse_wgs > 0 ? g_coarse_wgs : cus);
    const size_t lists = (size_t)wgs * 8, off = (lists * 4 + 255) & ~(size_t)255;
    unsigned* hitcount = (unsigned*)hit_scratch;
    u32x4_t* hitlist = (u32x4_t*)((char*)hit_scratch + off);
    const int hitcap = (int)std::min<size_t>((hit_bytes - off) / (lists * 16), (size_t)1 << 20);
    hipLaunchKernelGGL((scan_coarse64s_kernel<METRIC, false>), dim3((unsigned)wgs), dim3(512), C64_LDS_BYTES, s, a, hitlist, hitcount, hitcap, GroupedArgs{});
    hipLaunchKernelGGL(scan_hits_scatter_kernel, dim3((unsigned)lists), dim3(64), 0, s, (const u32x4_t*)hitlist, (const unsigned*)hitcount, hitcap, a.count, a.surv, a.cap);
}
template <int METRIC>
static void launch_ivf_coarse_m(const CoarseArgs& a, const GroupedArgs& ga, int wgs, u32x4_t* hitlist, unsigned* hitcount, int hitcap, size_t lists, bool dense,
                                unsigned* count, uint64_t* surv, int cap, hipStream_t s) {
    static ScDeviceOnce once;
    sc_device_once(once, [&] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse64s_kernel<METRIC, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C64_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse64s_kernel<METRIC, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C64_LDS_BYTES);
    });
    if (dense) {  // keys go straight to their places
        hipLaunchKernelGGL((scan_coarse64s_kernel<METRIC, true, true>), dim3((unsigned)wgs), dim3(512), C64_LDS_BYTES, s, a, hitlist, hitcount, hitcap, ga);
        return;
    }
    hipLaunchKernelGGL((scan_coarse64s_kernel<METRIC, true>), dim3((unsigned)wgs), dim3(512), C64_LDS_BYTES, s, a, hitlist, hitcount, hitcap, ga);
    hipLaunchKernelGGL(scan_hits_scatter_kernel, dim3((unsigned)lists), dim3(64), 0, s, (const u32x4_t*)hitlist, (const unsigned*)hitcount, hitcap, count, surv, cap);
}
// IVF_FLAT coarse stage (L2 / IP): `nitems` work items {row0, rows, slot_base} over the centred int8 shadow Xc8 / per-pair queries Qc8; per-slot
// thresholds etc.; hits go to the survivor lists of the slots' queries.  hit_scratch as for the flat form.
void sc_launch_ivf_coarse(const void* Xc8, const float* xrow, int ld8, const void* Qc8, const void* items, int nitems, const float* slot_tf,
                          const float* slot_thr, const float* slot_qn, const float* slot_qs, const int32_t* slot_q, const float* slot_qb, const float* slot_qd,
                          uint64_t* surv, unsigned* count, int cap, void* hit_scratch, size_t hit_bytes, hipStream_t s, const int32_t* slot_dst, int metric) {
    if (nitems <= 0) return;
    CoarseArgs a;
    a.Xb = (const bf16_t*)Xc8; a.xnorm = nullptr; a.xscale = nullptr; a.row0 = 0; a.row1 = 0; a.ld = ld8 / 2; a.Qb = (const bf16_t*)Qc8; a.qnorm = nullptr; a.Q = 0; a.qtiles = 1;
    a.thr = nullptr; a.thr_fast = nullptr; a.surv = surv; a.count = count; a.cap = cap; a.ntiles = nitems; a.qscale = nullptr; a.trace = nullptr;
    GroupedArgs ga;
    ga.items = (const GroupItem*)items; ga.nitems = nitems; ga.slot_tf = slot_tf; ga.slot_thr = slot_thr; ga.slot_qn = slot_qn; ga.slot_qs = slot_qs; ga.slot_q = slot_q; ga.slot_qb = slot_qb; ga.slot_qd = slot_qd; ga.xrow = (const f32x4*)xrow;
    ga.slot_dst = slot_dst ? slot_dst : slot_q;  // (wave 7 requests it either way)
    const int cus = sc_device_cus();
    const int wgs = std::min(nitems, g_coarse_wgs > 0 ? g_coarse_wgs : cus);
    const size_t lists = (size_t)wgs * 8, off = (lists * 4 + 255) & ~(size_t)255;
    unsigned* hitcount = (unsigned*)hit_scratch;
    u32x4_t* hitlist = (u32x4_t*)((char*)hit_scratch + off);
    const int hitcap = (int)std::min<size_t>((hit_bytes - off) / (lists * 16), (size_t)1 << 20);
    if (metric == SC_METRIC_L2) launch_ivf_coarse_m<SC_METRIC_L2>(a, ga, wgs, hitlist, hitcount, hitcap, lists, slot_dst != nullptr, count, surv, cap, s);
    else launch_ivf_coarse_m<SC_METRIC_IP>(a, ga, wgs, hitlist, hitcount, hitcap, lists, slot_dst != nullptr, count, surv, cap, s);
}
bool sc_scan_coarse64_supported(int Q, int ld8, size_t hit_bytes) { return Q >= 1 && Q <= 64 && ld8 >= 128 && (ld8 % 128) == 0 && hit_bytes >= (size_t)2048 * (4 + 64 * 16) + 256; }

template <int METRIC, bool I8>
static void launch_coarse256(const CoarseArgs& a, hipStream_t s, bool dense) {
    static const char* envd = getenv("SC_COARSE_DENSE");  // A/B: 0 = the sparse epilogue everywhere
    static const bool dense_ok = envd ? atoi(envd) != 0 : true;
    if (dense && dense_ok && a.ld >= 2 * G_BK) {
        static ScDeviceOnce once_d;
        sc_device_once(once_d, [&] { hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse256_kernel<METRIC, 0, I8, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSE_LDS_BYTES); });
        hipLaunchKernelGGL((scan_coarse256_kernel<METRIC, 0, I8, 4, true>), dim3((unsigned)a.ntiles), dim3(512), COARSE_LDS_BYTES, s, a);
        return;
    }
    static const char* env = getenv("SC_COARSE_PP");  // A/B: 0 = the one-barrier main loop
    static const bool pp = (env ? atoi(env) : 4) != 0;
    static ScDeviceOnce once;  // per instantiation and device
    sc_device_once(once, [&] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse256_kernel<METRIC, 0, I8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSE_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse256_kernel<METRIC, 0, I8, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSE_LDS_BYTES);
        hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse256p_kernel<METRIC, I8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSEP_LDS_BYTES);
    });
    static const char* envp = getenv("SC_COARSE_PERSIST");  // A/B: 0 = one workgroup per tile
    static const bool persist_env = envp ? atoi(envp) != 0 : true;
    if (pp && persist_env && g_coarse_persistent && a.ld >= 3 * G_BK) {
        const int cus = sc_device_cus(), cus8 = cus >= 8 ? (cus & ~7) : 8;
        const int wgs = g_coarse_wgs > 0 ? ((g_coarse_wgs + 7) & ~7) : cus8;  // a multiple of 8: blocks b, b + 8, ... share an XCD
        hipLaunchKernelGGL((scan_coarse256p_kernel<METRIC, I8>), dim3((unsigned)wgs), dim3(512), COARSEP_LDS_BYTES, s, a);
        return;
    }
    if (pp && a.ld >= 2 * G_BK) hipLaunchKernelGGL((scan_coarse256_kernel<METRIC, 0, I8, 4>), dim3((unsigned)a.ntiles), dim3(512), COARSE_LDS_BYTES, s, a);
    else hipLaunchKernelGGL((scan_coarse256_kernel<METRIC, 0, I8, 0>), dim3((unsigned)a.ntiles), dim3(512), COARSE_LDS_BYTES, s, a);
}

// i8: Xb / Qb are the int8 shadows with rows of ld8 bytes (`ld` is then ld8), xscale / qscale their per-row scales; the batch must be
// padded to 256 queries (the int8 stage only exists on the 256 x 256 tile)
// SC_COARSE_TRACE: one traced launch, summarised on stderr (mean us per workgroup: entry -> main loop done -> end, and the idle gap
// between consecutive workgroups of one CU)
static void coarse256_trace(CoarseArgs a, bool i8, hipStream_t s) {
    unsigned long long* dev = nullptr;
    const size_t words = (size_t)a.ntiles * 8;
    if (hipMalloc(&dev, words * 8) != hipSuccess) return;
    (void)hipMemsetAsync(dev, 0, words * 8, s);
    a.trace = dev;
    static const int mode = atoi(getenv("SC_COARSE_TRACE"));  // 1 trace, 2 trace without the returning atomics (results invalid)
    if (i8 && mode == 5) {  // the persistent kernel: stamps of wave 0 per tile (entry = its own loop start)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse256p_kernel<SC_METRIC_L2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSEP_LDS_BYTES);
        hipLaunchKernelGGL((scan_coarse256p_kernel<SC_METRIC_L2, true, true>), dim3(256), dim3(512), COARSEP_LDS_BYTES, s, a);
    } else if (i8 && mode == 3) {  // main loop only
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse256_kernel<SC_METRIC_L2, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSE_LDS_BYTES);
        hipLaunchKernelGGL((scan_coarse256_kernel<SC_METRIC_L2, 3, true>), dim3((unsigned)a.ntiles), dim3(512), COARSE_LDS_BYTES, s, a);
    } else if (i8 && mode == 4) {  // bound tests only
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse256_kernel<SC_METRIC_L2, 10, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSE_LDS_BYTES);
        hipLaunchKernelGGL((scan_coarse256_kernel<SC_METRIC_L2, 10, true>), dim3((unsigned)a.ntiles), dim3(512), COARSE_LDS_BYTES, s, a);
    } else if (i8 && mode == 2) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse256_kernel<SC_METRIC_L2, 6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSE_LDS_BYTES);
        hipLaunchKernelGGL((scan_coarse256_kernel<SC_METRIC_L2, 6, true>), dim3((unsigned)a.ntiles), dim3(512), COARSE_LDS_BYTES, s, a);
    } else if (i8) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse256_kernel<SC_METRIC_L2, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSE_LDS_BYTES);
        hipLaunchKernelGGL((scan_coarse256_kernel<SC_METRIC_L2, 2, true>), dim3((unsigned)a.ntiles), dim3(512), COARSE_LDS_BYTES, s, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse256_kernel<SC_METRIC_L2, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSE_LDS_BYTES);
        hipLaunchKernelGGL((scan_coarse256_kernel<SC_METRIC_L2, 2, false>), dim3((unsigned)a.ntiles), dim3(512), COARSE_LDS_BYTES, s, a);
    }
    std::vector<unsigned long long> h(words);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), dev, words * 8, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> per_cu;
    double ml = 0, ep = 0, taken = 0, fastpass = 0, hits = 0;
    unsigned long long t_first = ~0ull, t_last = 0;
    for (int t = 0; t < a.ntiles; ++t) {
        const unsigned long long* r = &h[(size_t)t * 8];
        taken += (double)r[4]; fastpass += (double)r[5]; hits += (double)r[6];
        ml += (double)(r[2] - r[1]);
        ep += (double)(r[3] - r[2]);
        // HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13; XCC_ID low bits of the high word
        per_cu[(r[0] >> 32) << 16 | ((r[0] >> 8) & 0xFF)].push_back({r[1], r[3]});
        t_first = r[1] < t_first ? r[1] : t_first;
        t_last = r[3] > t_last ? r[3] : t_last;
    }
    double gap = 0;
    size_t gaps = 0;
    for (auto& kv : per_cu) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end());
        for (size_t i = 1; i < v.size(); ++i) { gap += (double)((long long)v[i].first - (long long)v[i - 1].second); ++gaps; }
    }
    fprintf(stderr, "[coarse trace] %s tiles %d on %zu CUs, launch %.1f us: per tile entry->mainloop done %.2f us, epilogue %.2f us, gap to the next workgroup of the CU %.2f us; per tile: %.1f of 256 wave-groups entered the precise test, %.1f scores passed the fast test, %.1f survivors\n",
            i8 ? "int8" : "bf16", a.ntiles, per_cu.size(), (double)(t_last - t_first) / 100.0, ml / a.ntiles / 100.0, ep / a.ntiles / 100.0, gaps ? gap / gaps / 100.0 : 0.0, taken / a.ntiles, fastpass / a.ntiles, hits / a.ntiles);
}

void sc_launch_scan_coarse(int metric, const void* Xb, const float* xnorm, int64_t row0, int64_t row1, int ld, const void* Qb,
                           const float* qnorm, int Q, int Qpad, const float* thr, const float* thr_fast, uint64_t* surv, unsigned* count,
                           int cap, hipStream_t s, bool i8, const float* xscale, const float* qscale, bool dense, void* hit_scratch, size_t hit_bytes) {
    CoarseArgs a;
    a.Xb = (const bf16_t*)Xb; a.xnorm = xnorm; a.row0 = row0; a.row1 = row1; a.ld = i8 ? ld / 2 : ld; a.Qb = (const bf16_t*)Qb; a.qnorm = qnorm;
    a.Q = Q; a.thr = thr; a.thr_fast = thr_fast; a.surv = surv; a.count = count; a.cap = cap; a.xscale = xscale; a.qscale = qscale;
    // SC_COARSE_EXPERIMENT=1: the persistent kernel skips its sparse epilogue, 2: runs only the epilogue's first pass (timing experiments; results invalid)
    static const int experiment = [] { const char* e = getenv("SC_COARSE_EXPERIMENT"); return e ? atoi(e) : 0; }();
    if (experiment && !dense && row0 > 0) a.cap = -experiment;
    static const char* env64 = getenv("SC_COARSE64");  // A/B: 0 = small batches through the 256-query tiles
    if (i8 && !dense && hit_scratch && (row0 % T_BM) == 0 && sc_scan_coarse64_supported(Q, ld, hit_bytes) && !(env64 && env64[0] == '0') && !getenv("SC_COARSE_TRACE") &&
        !getenv("SC_COARSE_DBG")) {
        a.qtiles = 1;
        a.ntiles = (int)((row1 - row0 + T_BM - 1) / T_BM);
        a.trace = nullptr;
        if (metric == SC_METRIC_L2) launch_coarse64s<SC_METRIC_L2>(a, s, hit_scratch, hit_bytes);
        else if (metric == SC_METRIC_COSINE) launch_coarse64s<SC_METRIC_COSINE>(a, s, hit_scratch, hit_bytes);
        else launch_coarse64s<SC_METRIC_IP>(a, s, hit_scratch, hit_bytes);
        return;
    }
    if ((Qpad % T_BN) == 0 && (row0 % T_BM) == 0) {  // large batches: 256 x 256 tiles (corpus rows are padded to 256)
        a.qtiles = Qpad / T_BN;
        a.ntiles = (int)(((row1 - row0 + T_BM - 1) / T_BM) * a.qtiles);
        a.trace = nullptr;
        static const bool trace = getenv("SC_COARSE_TRACE") != nullptr;  // diagnostic: per-workgroup time stamps of the large L2 launches -> stderr
        static const int trace_min = [] { const char* e = getenv("SC_COARSE_TRACE_MIN"); return e ? atoi(e) : 20000; }();
        if (trace && metric == SC_METRIC_L2 && a.ntiles >= trace_min) {
            coarse256_trace(a, i8, s);
            return;
        }
        static const bool dbg = getenv("SC_COARSE_DBG") != nullptr;  // diagnostic: time the main loop alone (results invalid)
        if (dbg) {
            if (i8) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse256_kernel<SC_METRIC_L2, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSE_LDS_BYTES);
                hipLaunchKernelGGL((scan_coarse256_kernel<SC_METRIC_L2, 1, true>), dim3((unsigned)a.ntiles), dim3(512), COARSE_LDS_BYTES, s, a);
            } else {
                hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse256_kernel<SC_METRIC_L2, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COARSE_LDS_BYTES);
                hipLaunchKernelGGL((scan_coarse256_kernel<SC_METRIC_L2, 1, false>), dim3((unsigned)a.ntiles), dim3(512), COARSE_LDS_BYTES, s, a);
            }
            return;
        }
        if (i8) {
            if (metric == SC_METRIC_L2) launch_coarse256<SC_METRIC_L2, true>(a, s, dense);
            else if (metric == SC_METRIC_COSINE) launch_coarse256<SC_METRIC_COSINE, true>(a, s, dense);
            else launch_coarse256<SC_METRIC_IP, true>(a, s, dense);
        } else {
            if (metric == SC_METRIC_L2) launch_coarse256<SC_METRIC_L2, false>(a, s, dense);
            else if (metric == SC_METRIC_COSINE) launch_coarse256<SC_METRIC_COSINE, false>(a, s, dense);
            else launch_coarse256<SC_METRIC_IP, false>(a, s, dense);
        }
        return;
    }
    a.qtiles = Qpad / G_BN;
    const int64_t rtiles = (row1 - row0 + G_BM - 1) / G_BM;
    a.ntiles = (int)(rtiles * a.qtiles);
    const size_t lds = 4 * G_TILE_BYTES;
    static ScDeviceOnce once128;
    sc_device_once(once128, [&] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse_kernel<SC_METRIC_IP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse_kernel<SC_METRIC_L2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(scan_coarse_kernel<SC_METRIC_COSINE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    dim3 grid((unsigned)a.ntiles), block(256);
    if (metric == SC_METRIC_L2) hipLaunchKernelGGL(scan_coarse_kernel<SC_METRIC_L2>, grid, block, lds, s, a);
    else if (metric == SC_METRIC_COSINE) hipLaunchKernelGGL(scan_coarse_kernel<SC_METRIC_COSINE>, grid, block, lds, s, a);
    else hipLaunchKernelGGL(scan_coarse_kernel<SC_METRIC_IP>, grid, block, lds, s, a);
}

void sc_launch_scan_select(int metric, uint64_t* surv, unsigned* count, int cap, uint64_t* best, const float* qnorm, float* thr,
                           float* thr_fast, int* overflow, int Q, int kp, hipStream_t s) {
    const size_t lds = (size_t)(cap + kp) * 8 + (SEL_BINS + SEL_THREADS + 8) * 4;
    static ScDeviceOnce once_sel;
    sc_device_once(once_sel, [&] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(scan_select_kernel<SC_METRIC_IP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(scan_select_kernel<SC_METRIC_L2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(scan_select_kernel<SC_METRIC_COSINE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    dim3 grid((unsigned)Q), block(SEL_THREADS);
    if (metric == SC_METRIC_L2) hipLaunchKernelGGL(scan_select_kernel<SC_METRIC_L2>, grid, block, lds, s, surv, count, cap, best, qnorm, thr, thr_fast, overflow, kp);
    else if (metric == SC_METRIC_COSINE) hipLaunchKernelGGL(scan_select_kernel<SC_METRIC_COSINE>, grid, block, lds, s, surv, count, cap, best, qnorm, thr, thr_fast, overflow, kp);
    else hipLaunchKernelGGL(scan_select_kernel<SC_METRIC_IP>, grid, block, lds, s, surv, count, cap, best, qnorm, thr, thr_fast, overflow, kp);
}

template <int METRIC>
static void launch_rerank(const float* X, const float* xnorm, int ld, const float* Qp, const float* qnorm, const uint64_t* best, const float* thr,
                          const unsigned* bits, const float* qres, const int* overflow, int Q, int k, int64_t row_base, const uint32_t* perm,
                          float* out_dist, int64_t* out_rows, int* flags, int kp, uint64_t* ekeys, hipStream_t s) {
    if (kp == KPRIME) {
        hipLaunchKernelGGL((scan_rerank_kernel<METRIC, false>), dim3((unsigned)Q), dim3(KPRIME), 0, s, X, xnorm, ld, Qp, qnorm, best, thr, bits, qres, overflow, k,
                           row_base, perm, out_dist, out_rows, flags, kp, (uint64_t*)nullptr);
        return;
    }
    hipLaunchKernelGGL((scan_rerank_kernel<METRIC, true>), dim3((unsigned)Q, (unsigned)(kp / KPRIME)), dim3(KPRIME), 0, s, X, xnorm, ld, Qp, qnorm, best, thr, bits,
                       qres, overflow, k, row_base, perm, out_dist, out_rows, flags, kp, ekeys);
    hipLaunchKernelGGL(scan_finalize_kernel<METRIC>, dim3((unsigned)Q), dim3(256), 0, s, ekeys, kp, qnorm, thr, bits, qres, overflow, ld, k, row_base, out_dist,
                       out_rows, flags);
}

// exact keys of cand [Q][kp] (the first ncand[q] slots of query q) -> ekeys [Q][kp]; nothing else (no sort, no certificate)
template <int METRIC>
static void launch_rerank_keys(const float* X, const float* xnorm, int ld, const float* Qp, const float* qnorm, const uint64_t* cand, const int* ncand, int kp,
                               const uint32_t* perm, uint64_t* ekeys, int Q, hipStream_t s) {
    hipLaunchKernelGGL((scan_rerank_kernel<METRIC, true>), dim3((unsigned)Q, (unsigned)(kp / KPRIME)), dim3(KPRIME), 0, s, X, xnorm, ld, Qp, qnorm, cand,
                       (const float*)nullptr, (const unsigned*)nullptr, (const float*)nullptr, (const int*)nullptr, 0, (int64_t)0, perm, (float*)nullptr, (int64_t*)nullptr,
                       (int*)nullptr, kp, ekeys, ncand);
}
void sc_launch_scan_rerank_keys(int metric, const float* X, const float* xnorm, int ld, const float* Qp, const float* qnorm, const uint64_t* cand, const int* ncand, int kp,
                                const uint32_t* perm, uint64_t* ekeys, int Q, hipStream_t s) {
    if (metric == SC_METRIC_L2) launch_rerank_keys<SC_METRIC_L2>(X, xnorm, ld, Qp, qnorm, cand, ncand, kp, perm, ekeys, Q, s);
    else if (metric == SC_METRIC_COSINE) launch_rerank_keys<SC_METRIC_COSINE>(X, xnorm, ld, Qp, qnorm, cand, ncand, kp, perm, ekeys, Q, s);
    else launch_rerank_keys<SC_METRIC_IP>(X, xnorm, ld, Qp, qnorm, cand, ncand, kp, perm, ekeys, Q, s);
}
void sc_launch_scan_rerank_keys_l2(const float* X, const float* xnorm, int ld, const float* Qp, const float* qnorm, const uint64_t* cand, const int* ncand, int kp,
                                   const uint32_t* perm, uint64_t* ekeys, int Q, hipStream_t s) {
    launch_rerank_keys<SC_METRIC_L2>(X, xnorm, ld, Qp, qnorm, cand, ncand, kp, perm, ekeys, Q, s);
}
void sc_launch_scan_collect_bound(int metric, const float* prev_dist, int k, const float* qnorm, const float* qres, const unsigned* bits, int ld, float* thr,
                                  float* thr_fast, int* flags, int Q, hipStream_t s) {
    const dim3 grid((unsigned)((Q + 255) / 256)), block(256);
    if (metric == SC_METRIC_L2) hipLaunchKernelGGL(scan_collect_bound_kernel<SC_METRIC_L2>, grid, block, 0, s, prev_dist, k, qnorm, qres, bits, ld, thr, thr_fast, flags, Q);
    else if (metric == SC_METRIC_COSINE) hipLaunchKernelGGL(scan_collect_bound_kernel<SC_METRIC_COSINE>, grid, block, 0, s, prev_dist, k, qnorm, qres, bits, ld, thr, thr_fast, flags, Q);
    else hipLaunchKernelGGL(scan_collect_bound_kernel<SC_METRIC_IP>, grid, block, 0, s, prev_dist, k, qnorm, qres, bits, ld, thr, thr_fast, flags, Q);
}
void sc_launch_scan_tighten(int metric, const uint64_t* ekeys, int kp, int k, const float* qnorm, const float* qres, const unsigned* bits, int ld, float* thr,
                            float* thr_fast, float* thr_cut, int Q, hipStream_t s) {
    const dim3 grid((unsigned)Q), block(128);
    if (metric == SC_METRIC_L2) hipLaunchKernelGGL(scan_tighten_kernel<SC_METRIC_L2>, grid, block, 0, s, ekeys, kp, k, qnorm, qres, bits, ld, thr, thr_fast, thr_cut);
    else if (metric == SC_METRIC_COSINE) hipLaunchKernelGGL(scan_tighten_kernel<SC_METRIC_COSINE>, grid, block, 0, s, ekeys, kp, k, qnorm, qres, bits, ld, thr, thr_fast, thr_cut);
    else hipLaunchKernelGGL(scan_tighten_kernel<SC_METRIC_IP>, grid, block, 0, s, ekeys, kp, k, qnorm, qres, bits, ld, thr, thr_fast, thr_cut);
}
void sc_launch_scan_select_wide(int metric, const uint64_t* surv, unsigned* count, int cap, uint64_t* best, unsigned* nbest, int kcap, int kp, const float* qnorm,
                                float* thr, float* thr_fast, const float* thr_cut, int* overflow, int Q, hipStream_t s) {
    const size_t lds = (size_t)(cap + kcap) * 8 + (SEL_BINS + SEL_THREADS + 8) * 4;
    static ScDeviceOnce once;
    sc_device_once(once, [&] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(scan_select_wide_kernel<SC_METRIC_IP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(scan_select_wide_kernel<SC_METRIC_L2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(scan_select_wide_kernel<SC_METRIC_COSINE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const dim3 grid((unsigned)Q), block(SEL_THREADS);
    if (metric == SC_METRIC_L2) hipLaunchKernelGGL(scan_select_wide_kernel<SC_METRIC_L2>, grid, block, lds, s, surv, count, cap, best, nbest, kcap, kp, qnorm, thr, thr_fast, thr_cut, overflow);
    else if (metric == SC_METRIC_COSINE) hipLaunchKernelGGL(scan_select_wide_kernel<SC_METRIC_COSINE>, grid, block, lds, s, surv, count, cap, best, nbest, kcap, kp, qnorm, thr, thr_fast, thr_cut, overflow);
    else hipLaunchKernelGGL(scan_select_wide_kernel<SC_METRIC_IP>, grid, block, lds, s, surv, count, cap, best, nbest, kcap, kp, qnorm, thr, thr_fast, thr_cut, overflow);
}
void sc_launch_scan_wide_compact(int metric, const uint64_t* best, const unsigned* nbest, int kcap, const float* thr, uint64_t* cand, int* ncand, int Q, hipStream_t s) {
    const dim3 grid((unsigned)Q), block(256);
    if (metric == SC_METRIC_L2) hipLaunchKernelGGL(scan_wide_compact_kernel<SC_METRIC_L2>, grid, block, 0, s, best, nbest, kcap, thr, cand, ncand);
    else if (metric == SC_METRIC_COSINE) hipLaunchKernelGGL(scan_wide_compact_kernel<SC_METRIC_COSINE>, grid, block, 0, s, best, nbest, kcap, thr, cand, ncand);
    else hipLaunchKernelGGL(scan_wide_compact_kernel<SC_METRIC_IP>, grid, block, 0, s, best, nbest, kcap, thr, cand, ncand);
}
void sc_launch_scan_wide_certify(int metric, const float* out_dist, int k, const float* qnorm, const float* qres, const unsigned* bits, int ld, const float* thr,
                                 const int* overflow, int* flags, int Q, hipStream_t s) {
    const dim3 grid((unsigned)((Q + 255) / 256)), block(256);
    if (metric == SC_METRIC_L2) hipLaunchKernelGGL(scan_wide_certify_kernel<SC_METRIC_L2>, grid, block, 0, s, out_dist, k, qnorm, qres, bits, ld, thr, overflow, flags, Q);
    else if (metric == SC_METRIC_COSINE) hipLaunchKernelGGL(scan_wide_certify_kernel<SC_METRIC_COSINE>, grid, block, 0, s, out_dist, k, qnorm, qres, bits, ld, thr, overflow, flags, Q);
    else hipLaunchKernelGGL(scan_wide_certify_kernel<SC_METRIC_IP>, grid, block, 0, s, out_dist, k, qnorm, qres, bits, ld, thr, overflow, flags, Q);
}
void sc_launch_scan_thr_min(float* thr, const float* thr_cut, int Q, hipStream_t s) {
    hipLaunchKernelGGL(scan_thr_min_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, s, thr, thr_cut, Q);
}
void sc_launch_fill_u32(unsigned* p, unsigned v, int n, hipStream_t s) {
    hipLaunchKernelGGL(scan_fill_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, v, n);
}
void sc_launch_scan_collect_counts(const unsigned* count, int cap, int* ncand, int* flags, int Q, hipStream_t s) {
    hipLaunchKernelGGL(scan_collect_counts_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, s, count, cap, ncand, flags, Q);
}

// kp = KPRIME: one fused kernel; kp = KPRIME8 (int8 stage): blocks of 128 candidates re-scored into ekeys [Q][kp], then sorted
void sc_launch_scan_rerank(int metric, const float* X, const float* xnorm, int ld, const float* Qp, const float* qnorm, const uint64_t* best,
                           const float* thr, const unsigned* xnorm_max_bits, const float* qres, const int* overflow, int Q, int k,
                           int64_t row_base, const uint32_t* perm, float* out_dist, int64_t* out_rows, int* flags, hipStream_t s, int kp,
                           uint64_t* ekeys) {
    if (metric == SC_METRIC_L2) launch_rerank<SC_METRIC_L2>(X, xnorm, ld, Qp, qnorm, best, thr, xnorm_max_bits, qres, overflow, Q, k, row_base, perm, out_dist, out_rows, flags, kp, ekeys, s);
    else if (metric == SC_METRIC_COSINE) launch_rerank<SC_METRIC_COSINE>(X, xnorm, ld, Qp, qnorm, best, thr, xnorm_max_bits, qres, overflow, Q, k, row_base, perm, out_dist, out_rows, flags, kp, ekeys, s);
    else launch_rerank<SC_METRIC_IP>(X, xnorm, ld, Qp, qnorm, best, thr, xnorm_max_bits, qres, overflow, Q, k, row_base, perm, out_dist, out_rows, flags, kp, ekeys, s);
}

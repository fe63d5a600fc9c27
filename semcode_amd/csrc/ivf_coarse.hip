// ivf_coarse.hip -- data preparation for the int8 coarse stage of IVF_FLAT probing (L2): centred shadows and per-pair queries.
//
// Replaces (reference): the scan Milvus' IVF_FLAT runs over the probed lists behind MilvusVectorStore.search
// (src/semcode/storage/milvus_store.py:135-148; index parameters :76-83).  List-major probing scores every (query, probed row)
// pair exactly in f32 on the matrix cores -- 9.8e11 FLOP per 1 024-query batch at config 5, bound by the f32 MFMA rate and by
// re-streaming popular lists (29 ms, profiles/r3a_bench.json.log).  A low-precision coarse pass in front of it (as on the
// exhaustive path) fails on exactly the data an IVF index is built for: inside a tight cluster the Cauchy-Schwarz bound on the
// quantisation error, which scales with |x| |q|, exceeds the gap between the k-th and the kp-th neighbour.
//
// L2 distances do not change when x and q are shifted by the same vector.  So every list is quantised RELATIVE TO ITS CENTROID:
//     x' = x - c_list,  q' = q - c_list   (one q' per (query, probed list) pair),   |x - q|^2 = |x'|^2 + |q'|^2 - 2 <x', q'>
// and the error of the int8 dot product scales with |x'| |q'| -- the spread of the cluster, not its distance from the origin.
// The coarse score of a pair is turned into a LOWER bound of the exact distance by subtracting the pair's own error bound
// eps(list, pair); rows whose lower bound cannot beat the query's threshold are dropped, the survivors are re-scored exactly from
// the original f32 rows (canonical fmaf chain: the returned distances are those of the exact path), and the result is certified
// when the k-th exact distance lies below the final threshold -- no unseen row of the probed lists can then be closer.
//
//   ivf_center_shadow_kernel : Xc8 [rows][ld8] int8 of x', per row {|x'|^2, scale, 2 |dx|, 2 (|x'| + |dx|)}, per-list maxima of |dx|^2 and |x'|^2
//   ivf_pair_query_kernel    : per slot (query, list): int8 q', scale, |q'|^2 - eps  (the lower-bound form of the query norm)
//   ivf_slot_thr_kernel      : per slot thresholds from the per-query ones, before every phase
//   ivf_bound / ivf_candidates / ivf_refine_finalize : bound and refine (below): the exact re-rank of every row whose lower bound does not
//                              exceed an upper bound of the k-th distance
#include "sc_common.h"

// lists[nlist + 1] ascending: the list whose range holds stored position r
static __device__ __forceinline__ int list_of_pos(const int64_t* __restrict__ list_off, int nlist, int64_t r) {
    int lo = 0, hi = nlist;  // invariant: list_off[lo] <= r < list_off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (list_off[mid] <= r) lo = mid;
        else hi = mid;
    }
    return lo;
}

// one wave per stored row; list_stats[list] = {bits of max |x' - s q|^2, bits of max |x'|^2}
// UNIT (COSINE): cos(x, q) = <x / |x|, q / |q|>, the inner product of the normalised vectors, and that form may be shifted by ANY vector
// per list: x' = x / |x| - c / |c| (the normalised centroid keeps |x'| small for lists built by cosine).  The coarse stage then runs
// its IP form on unit vectors; the exact re-score uses the canonical cosine of the original rows as everywhere else.
template <bool UNIT>
__global__ __launch_bounds__(256) void ivf_center_shadow_kernel(const float* __restrict__ X, int64_t rows, int ld, int ld8, const float* __restrict__ C, int ldc,
                                                                 const int64_t* __restrict__ list_off, int nlist, int8_t* __restrict__ Xc8,
                                                                 f32x4* __restrict__ xrow, unsigned* __restrict__ list_stats,
                                                                 const float* __restrict__ xnorm, const float* __restrict__ cnorm,
                                                                 const int64_t* __restrict__ only) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t i = wave0; i < rows; i += nwaves) {
        const int64_t r = only ? only[i] : i;  // (only: the stored positions of rows overwritten in place, `rows` of them)
        const int l = list_of_pos(list_off, nlist, r);
        const float* x = X + r * (int64_t)ld;
        const float* c = C + (int64_t)l * ldc;
        float rx = 1.0f, rc = 1.0f;
        if (UNIT) {
            const float xn = xnorm[r], cn = cnorm[l];
            rx = xn > 0.f ? 1.0f / sqrtf(xn) : 0.f;
            rc = cn > 0.f ? 1.0f / sqrtf(cn) : 0.f;
        }
        float m = 0.f, nn = 0.f;
        for (int k0 = 4 * lane; k0 < ld; k0 += 256) {
            const f32x4 v = UNIT ? *reinterpret_cast<const f32x4*>(x + k0) * rx - *reinterpret_cast<const f32x4*>(c + k0) * rc
                                 : *reinterpret_cast<const f32x4*>(x + k0) - *reinterpret_cast<const f32x4*>(c + k0);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            nn = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], nn))));
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            m = fmaxf(m, __shfl_xor(m, off, 64));
            nn += __shfl_xor(nn, off, 64);
        }
        const float sc = m > 0.f ? m * (1.0f / 127.0f) : 1.0f, inv = 1.0f / sc;
        float res = 0.f;
        int8_t* o = Xc8 + r * (int64_t)ld8;
        for (int k0 = 16 * lane; k0 < ld8; k0 += 1024) {
            u32x4 packed = {0u, 0u, 0u, 0u};
            if (k0 < ld) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = UNIT ? *reinterpret_cast<const f32x4*>(x + k0 + 4 * j) * rx - *reinterpret_cast<const f32x4*>(c + k0 + 4 * j) * rc
                                         : *reinterpret_cast<const f32x4*>(x + k0 + 4 * j) - *reinterpret_cast<const f32x4*>(c + k0 + 4 * j);
                    uint32_t word = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float qf = fminf(fmaxf(rintf(v[e] * inv), -127.0f), 127.0f);
                        const float d = fmaf(-sc, qf, v[e]);
                        res = fmaf(d, d, res);
                        word |= ((uint32_t)(int)qf & 0xFFu) << (8 * e);
                    }
                    packed[j] = word;
                }
            }
            *reinterpret_cast<u32x4*>(o + k0) = packed;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) res += __shfl_xor(res, off, 64);
        if (lane == 0) {
            // the row's record for the coarse kernel: |x'|^2, scale, and the two row factors of its error bound
            //   eps(row, pair) = 2 |dx_r| |q'| + 2 (|x'_r| + |dx_r|) |dq|      (Cauchy-Schwarz on the actual rounding residuals)
            const float dx = sqrtf(res * 1.0001f), xm = sqrtf(nn * 1.0001f);
            xrow[r] = f32x4{nn, sc, 2.0f * dx, 2.0f * (xm + dx)};
            atomicMax(list_stats + 2 * l, __builtin_bit_cast(unsigned, res * 1.0001f));
            atomicMax(list_stats + 2 * l + 1, __builtin_bit_cast(unsigned, nn * 1.0001f));
        }
    }
}

// One wave per slot.  slot_q / slot_l: the pair (-1 = padding slot: zeros, never passes).  Outputs: Qc8 [slot][ld8], slot_qs (scale),
// slot_qnlb = |q'|^2 minus the f32 rounding allowance of the coarse score's own terms, slot_qb = |q'|, slot_qd = |q' - qq| (the pair
// factors of the per-row error bound the kernel subtracts: score - 2 |dx_r| |q'| - 2 (|x'_r| + |dx_r|) |dq| is a lower bound of
// |x - q|^2 up to the f32 rounding of the exact path, which the certificate of the re-rank adds), and slot_eps = that bound with the
// LIST's maxima in place of the row's: the fast test of the kernel must pass whatever any row of the list could pass.
// IP (template parameter false): <x, q> = <x', q> + <c, q>: only the rows are centred.  The slot's query is q itself (int8, its own
// scale), the slot constant is <c, q> (plus its rounding allowance: the kernel adds it to the coarse score, which must stay an UPPER
// bound of the exact inner product), and the pair factors are halved because the row record carries the factor 2 of the L2 form:
//   score_ub = s_r s_q <x'q, qq> + <c, q> + |dx_r| |q| + (|x'_r| + |dx_r|) |dq|.
// slot_qnlb = -(<c, q> + allowance): ivf_slot_thr_kernel's "threshold - query norm" then reads T + <c, q>, the bound on -<x', q>.
// UNIT (COSINE; with L2 = false): the IP form on q / |q| and c / |c| (ivf_center_shadow_kernel<true>).
template <bool L2, bool UNIT = false>
__global__ __launch_bounds__(256) void ivf_pair_query_kernel(const float* __restrict__ Qp, int ld, int ld8, const float* __restrict__ C, int ldc,
                                                              const int32_t* __restrict__ slot_q, const int32_t* __restrict__ slot_l, int nslots,
                                                              const unsigned* __restrict__ list_stats, int8_t* __restrict__ Qc8, float* __restrict__ slot_qs,
                                                              float* __restrict__ slot_qnlb, float* __restrict__ slot_qb, float* __restrict__ slot_qd,
                                                              float* __restrict__ slot_eps, const float* __restrict__ qnorm, const float* __restrict__ cnorm) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int slot = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= nslots) return;
    const int q = slot_q[slot], l = slot_l[slot];
    int8_t* o = Qc8 + (int64_t)slot * ld8;
    if (q < 0) {
        for (int k0 = 16 * lane; k0 < ld8; k0 += 1024) *reinterpret_cast<u32x4*>(o + k0) = u32x4{0u, 0u, 0u, 0u};
        if (lane == 0) { slot_qs[slot] = 0.f; slot_qnlb[slot] = 0.f; slot_qb[slot] = 0.f; slot_qd[slot] = 0.f; slot_eps[slot] = 0.f; }
        return;
    }
    const float* x = Qp + (int64_t)q * ld;
    const float* c = C + (int64_t)l * ldc;
    float rq = 1.0f, rc = 1.0f;
    if (UNIT) {
        const float qn2 = qnorm[q], cn2 = cnorm[l];
        rq = qn2 > 0.f ? 1.0f / sqrtf(qn2) : 0.f;
        rc = cn2 > 0.f ? 1.0f / sqrtf(cn2) : 0.f;
    }
    float m = 0.f, nn = 0.f, cq = 0.f, cn = 0.f;
    f32x4 cq4 = {0.f, 0.f, 0.f, 0.f}, cn4 = {0.f, 0.f, 0.f, 0.f};  // (element-wise accumulators: a scalar fmaf chain over the elements came out
                                                                   // of hipcc as v_pk_fma_f32 with op_sel -- tests/test_isa.py)
    for (int k0 = 4 * lane; k0 < ld; k0 += 256) {
        f32x4 xv = *reinterpret_cast<const f32x4*>(x + k0), cv = *reinterpret_cast<const f32x4*>(c + k0);
        if (UNIT) { xv = xv * rq; cv = cv * rc; }
        const f32x4 v = L2 ? xv - cv : xv;
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        nn = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], nn))));
        if (!L2) {
            cq4 = __builtin_elementwise_fma(cv, xv, cq4);
            cn4 = __builtin_elementwise_fma(cv, cv, cn4);
        }
    }
    if (!L2) {
        cq = (cq4[0] + cq4[1]) + (cq4[2] + cq4[3]);
        cn = (cn4[0] + cn4[1]) + (cn4[2] + cn4[3]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        m = fmaxf(m, __shfl_xor(m, off, 64));
        nn += __shfl_xor(nn, off, 64);
        if (!L2) {
            cq += __shfl_xor(cq, off, 64);
            cn += __shfl_xor(cn, off, 64);
        }
    }
    const float sc = m > 0.f ? m * (1.0f / 127.0f) : 1.0f, inv = 1.0f / sc;
    float res = 0.f;
    for (int k0 = 16 * lane; k0 < ld8; k0 += 1024) {
        u32x4 packed = {0u, 0u, 0u, 0u};
        if (k0 < ld) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = *reinterpret_cast<const f32x4*>(x + k0 + 4 * j);
                if (UNIT) v = v * rq;
                if (L2) v = v - *reinterpret_cast<const f32x4*>(c + k0 + 4 * j);
                uint32_t word = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float qf = fminf(fmaxf(rintf(v[e] * inv), -127.0f), 127.0f);
                    const float d = fmaf(-sc, qf, v[e]);
                    res = fmaf(d, d, res);
                    word |= ((uint32_t)(int)qf & 0xFFu) << (8 * e);
                }
                packed[j] = word;
            }
        }
        *reinterpret_cast<u32x4*>(o + k0) = packed;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) res += __shfl_xor(res, off, 64);
    if (lane == 0) {
        const float dx = sqrtf(__builtin_bit_cast(float, list_stats[2 * l])), xm = sqrtf(__builtin_bit_cast(float, list_stats[2 * l + 1]));
        const float qn = sqrtf(nn * 1.0001f), dq = sqrtf(res * 1.0001f);
        slot_qs[slot] = sc;
        if (L2) {
            // f32 rounding of the coarse score's own terms (stored norms, scaling of the integer dot): taken off the query norm
            const float round_eps = (4.0e-6f * (xm * xm + nn) + (float)ld * 2.4e-7f * (xm + dx) * qn) * 1.01f + 1e-6f;
            slot_qnlb[slot] = nn - round_eps;
            slot_qb[slot] = qn * 1.01f;
            slot_qd[slot] = dq * 1.01f;
            slot_eps[slot] = (2.0f * (dx * qn + (xm + dx) * dq)) * 1.03f + 1e-6f;  // >= the kernel's row-wise bound for every row of the list
        } else {
            // rounding of the scaled integer dot and of <c, q> itself: an f32 sum of ld / 64 fmas per lane, a 6-level tree over the lanes
            // and 3 adds -- its error is below (ld / 64 + 9) 2^-24 sum |c_i q_i| <= ... |c| |q|; four times that is allowed for
            const float round_eps = (4.0e-6f * ((xm + dx) * qn + fabsf(cq)) + (float)(ld / 64 + 16) * 2.4e-7f * (sqrtf(cn) + xm + dx) * qn) * 1.01f + 1e-6f;
            slot_qnlb[slot] = -(cq + round_eps);
            slot_qb[slot] = 0.5f * qn * 1.01f;
            slot_qd[slot] = 0.5f * dq * 1.01f;
            slot_eps[slot] = (dx * qn + (xm + dx) * dq) * 1.03f + 1e-6f;
        }
    }
}

// per slot: thr / thr_fast of its query for this phase (scan_select_kernel's L2 formulas with the slot's lower-bound query norm)
__global__ __launch_bounds__(256) void ivf_slot_thr_kernel(const int32_t* __restrict__ slot_q, const float* __restrict__ slot_qnlb, const float* __restrict__ slot_eps,
                                                            const float* __restrict__ thr, int nslots, float* __restrict__ slot_thr, float* __restrict__ slot_tf) {
    const int s = (int)blockIdx.x * 256 + threadIdx.x;
    if (s >= nslots) return;
    const int q = slot_q[s];
    float t = -__builtin_inff(), tf = -__builtin_inff();
    if (q >= 0) {
        t = thr[q];
        const float qn = slot_qnlb[s];
        // the fast test (score - qn <= tf) must pass whatever the precise one (score - row-wise bound <= t) can pass
        tf = (t - qn) + slot_eps[s] + (1e-3f * fabsf(t) + 1e-6f) + 1e-3f * fabsf(qn);
        if (!(t < __builtin_inff())) tf = __builtin_inff();
    }
    slot_thr[s] = t;
    slot_tf[s] = tf;
}

void sc_launch_ivf_center_shadow(const float* X, int64_t rows, int ld, int ld8, const float* C, int ldc, const int64_t* list_off, int nlist, void* Xc8,
                                 float* xrow, unsigned* list_stats, hipStream_t s, const float* xnorm, const float* cnorm, const int64_t* only) {
    if (rows <= 0) return;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (xnorm && cnorm)
        hipLaunchKernelGGL(ivf_center_shadow_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, X, rows, ld, ld8, C, ldc, list_off, nlist, (int8_t*)Xc8, (f32x4*)xrow,
                           list_stats, xnorm, cnorm, only);
    else
        hipLaunchKernelGGL(ivf_center_shadow_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, X, rows, ld, ld8, C, ldc, list_off, nlist, (int8_t*)Xc8, (f32x4*)xrow,
                           list_stats, (const float*)nullptr, (const float*)nullptr, only);
}
void sc_launch_ivf_pair_query(const float* Qp, int ld, int ld8, const float* C, int ldc, const int32_t* slot_q, const int32_t* slot_l, int nslots,
                              const unsigned* list_stats, void* Qc8, float* slot_qs, float* slot_qnlb, float* slot_qb, float* slot_qd, float* slot_eps, hipStream_t s,
                              int metric, const float* qnorm, const float* cnorm) {
    if (nslots <= 0) return;
    const dim3 grid((unsigned)((nslots + 3) / 4)), block(256);
    if (metric == SC_METRIC_L2)
        hipLaunchKernelGGL((ivf_pair_query_kernel<true, false>), grid, block, 0, s, Qp, ld, ld8, C, ldc, slot_q, slot_l, nslots, list_stats, (int8_t*)Qc8, slot_qs, slot_qnlb,
                           slot_qb, slot_qd, slot_eps, (const float*)nullptr, (const float*)nullptr);
    else if (metric == SC_METRIC_COSINE)
        hipLaunchKernelGGL((ivf_pair_query_kernel<false, true>), grid, block, 0, s, Qp, ld, ld8, C, ldc, slot_q, slot_l, nslots, list_stats, (int8_t*)Qc8, slot_qs, slot_qnlb,
                           slot_qb, slot_qd, slot_eps, qnorm, cnorm);
    else
        hipLaunchKernelGGL((ivf_pair_query_kernel<false, false>), grid, block, 0, s, Qp, ld, ld8, C, ldc, slot_q, slot_l, nslots, list_stats, (int8_t*)Qc8, slot_qs, slot_qnlb,
                           slot_qb, slot_qd, slot_eps, (const float*)nullptr, (const float*)nullptr);
}
void sc_launch_ivf_slot_thr(const int32_t* slot_q, const float* slot_qnlb, const float* slot_eps, const float* thr, int nslots, float* slot_thr, float* slot_tf,
                            hipStream_t s) {
    if (nslots <= 0) return;
    hipLaunchKernelGGL(ivf_slot_thr_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, s, slot_q, slot_qnlb, slot_eps, thr, nslots, slot_thr, slot_tf);
}

// ---- bound and refine ---------------------------------------------------------------------------------------------------------------
// Round-3 first form kept the kp = 512 best lower bounds of a query and certified the result when the k-th exact distance lay below
// the 512th lower bound; 124 .. 175 of 1024 queries at config 5 failed that test (more than 512 rows of their own cluster have lower
// bounds below d_k: int8 cannot tell them apart) and their exact re-probe cost 6.0 of the batch's 14.2 ms.  The bound that matters is
// not the kp-th LOWER bound but an UPPER bound of d_k, and phase A provides one almost for free:
//   phase A   every row of the query's nearest list(s): lower-bound keys, all kept (survA; dense, no test);
//             its kpa = 128 best are re-scored exactly: their k-th exact distance dA >= d_k, and T = dA + allowance;
//   phase B   the other lists: a row survives iff lower bound <= T (survB);
//   refine    S = {survA \ the 128 already done : lower bound <= T} + survB is re-scored exactly; exact top-k of (the 128 + S).
// Every probed row outside S has exact distance >= lower bound > T >= d_k: the result is the exact probe's, bit for bit, with no
// candidate count to exceed -- only |S| > IVFW_CAP or an overflowing survivor list send a query to the exact probe.
// The allowance is the certificate's own (f32 rounding of the exact scores; scan_batched.hip certified<>).
#define IVFW_CAP 4096
int sc_ivf_widen_cap(void) { return IVFW_CAP; }

static __device__ __forceinline__ float ivf_rounding_allowance(int metric, const unsigned* __restrict__ xmax_bits, float qnorm2, int ld) {
    if (metric == SC_METRIC_COSINE) return ((float)ld * 1.2e-7f) * 1.01f + 1e-6f;  // scores of unit vectors (the certificate's relative form)
    const float xmax = sqrtf(__builtin_bit_cast(float, xmax_bits[0])), qn = sqrtf(qnorm2);
    return ((metric == SC_METRIC_L2 ? 2.0f : 1.0f) * (float)ld * 1.2e-7f * xmax * qn) * 1.01f + 1e-6f;
}
// v-space (smaller = better) value of a key: the distance (L2) or minus the inner product (IP)
static __device__ __forceinline__ float ivf_key_v(int metric, uint64_t key) {
    const float sc = sc_key_score(metric, key);
    return metric == SC_METRIC_L2 ? sc : -sc;
}

// T[q] = (k-th smallest exact distance among the kpa re-scored phase-A candidates) + allowance; +inf if there are fewer than k
// ekeysT (optional): a second sample of kpa re-scored rows (the two-level bound of long lists); the k-th of the union
__global__ __launch_bounds__(128) void ivf_bound_kernel(int metric, const uint64_t* __restrict__ ekeysA, int kpa_one, int k, const float* __restrict__ qnorm,
                                                         const unsigned* __restrict__ xmax_bits, int ld, float* __restrict__ thr,
                                                         const uint64_t* __restrict__ ekeysT, int keep_min) {
    __shared__ uint64_t keys[1024];
    __shared__ float s_t;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int kpa = ekeysT ? 2 * kpa_one : kpa_one;
    for (int i = tid; i < kpa; i += 128) keys[i] = i < kpa_one ? ekeysA[(size_t)q * kpa_one + i] : ekeysT[(size_t)q * kpa_one + (i - kpa_one)];
    if (tid == 0) s_t = __builtin_inff();
    __syncthreads();
    for (int i = tid; i < kpa; i += 128) {
        const uint64_t key = keys[i];
        if (key == SC_KEY_MAX) continue;
        int rank = 0;
        for (int j = 0; j < kpa; ++j) rank += keys[j] < key ? 1 : 0;  // exact keys are unique (row id in the low word; the two samples are disjoint rows)
        if (rank == k - 1) s_t = ivf_key_v(metric, key) + ivf_rounding_allowance(metric, xmax_bits, qnorm[q], ld);
    }
    __syncthreads();
    if (tid == 0) thr[q] = keep_min ? fminf(thr[q], s_t) : s_t;  // (every such bound is valid, so is their minimum)
}

// S of one query -> cand [q][IVFW_CAP], ncand[q]; flags[q] = 1 (exact probe) when a survivor list overflowed or S does not fit
__global__ __launch_bounds__(256) void ivf_candidates_kernel(int metric, const uint64_t* __restrict__ survA, const unsigned* __restrict__ cntA, const uint64_t* __restrict__ bestA, int kpa,
                                                              const uint64_t* __restrict__ survB, const unsigned* __restrict__ cntB, int cap,
                                                              const float* __restrict__ thr, uint64_t* __restrict__ cand, int* __restrict__ ncand,
                                                              int* __restrict__ flags, int wcap, int capB) {
    __shared__ unsigned s_n;
    __shared__ uint64_t wmax[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_n = 0;
    const unsigned nA = cntA[q], nB = cntB[q];
    const float T = thr[q];
    if (nA > (unsigned)cap || nB > (unsigned)capB || !(T < __builtin_inff())) {  // uniform over the workgroup
        if (tid == 0) { ncand[q] = 0; flags[q] = 1; }
        return;
    }
    // the largest key of bestA: survA keys up to it are the kpa that have been re-scored already
    uint64_t m = 0;
    for (int i = tid; i < kpa; i += 256) {
        const uint64_t key = bestA[(size_t)q * kpa + i];
        if (key != SC_KEY_MAX && key > m) m = key;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint64_t o = ((uint64_t)__shfl_xor((unsigned)(m >> 32), off, 64) << 32) | (uint64_t)__shfl_xor((unsigned)m, off, 64);
        m = o > m ? o : m;
    }
    if ((tid & 63) == 0) wmax[tid >> 6] = m;
    __syncthreads();
    uint64_t pivot = wmax[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) pivot = wmax[w] > pivot ? wmax[w] : pivot;
    for (unsigned i = tid; i < nA + nB; i += 256) {
        const bool fromA = i < nA;
        const uint64_t key = fromA ? survA[(size_t)q * cap + i] : survB[(size_t)q * capB + (i - nA)];
        if (fromA && key <= pivot) continue;
        if (ivf_key_v(metric, key) <= T) {
            const unsigned pos = atomicAdd(&s_n, 1u);
            if (pos < (unsigned)wcap) cand[(size_t)q * IVFW_CAP + pos] = key;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const bool fits = s_n <= (unsigned)wcap;
        ncand[q] = fits ? (int)s_n : 0;
        flags[q] = fits ? 0 : 1;
    }
}

// exact keys of the kpa + ncand[q] re-scored rows -> the k best, in order (flagged queries are left to the exact probe)
__global__ __launch_bounds__(256) void ivf_refine_finalize_kernel(int metric, const uint64_t* __restrict__ ekeysA, int kpa, const uint64_t* __restrict__ ekeys,
                                                                   const int* __restrict__ ncand, const int* __restrict__ flags, int k, int64_t row_base,
                                                                   float* __restrict__ out_dist, int64_t* __restrict__ out_rows, int stride) {
    __shared__ uint64_t keys[IVFW_CAP + 512];
    __shared__ uint64_t wmin[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    if (flags[q]) return;
    const int n = kpa + ncand[q];
    for (int i = tid; i < n; i += 256) keys[i] = i < kpa ? ekeysA[(size_t)q * kpa + i] : ekeys[(size_t)q * stride + (i - kpa)];
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        uint64_t m = SC_KEY_MAX;
        for (int i = tid; i < n; i += 256) m = keys[i] < m ? keys[i] : m;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const uint64_t o = ((uint64_t)__shfl_xor((unsigned)(m >> 32), off, 64) << 32) | (uint64_t)__shfl_xor((unsigned)m, off, 64);
            m = o < m ? o : m;
        }
        if ((tid & 63) == 0) wmin[tid >> 6] = m;
        __syncthreads();
        uint64_t b = wmin[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) b = wmin[w] < b ? wmin[w] : b;
        if (tid == 0) {
            const size_t o = (size_t)q * k + j;
            if (b != SC_KEY_MAX) {
                out_dist[o] = sc_key_score(metric, b);
                out_rows[o] = row_base + (int64_t)(uint32_t)b;
            } else {
                out_dist[o] = (metric == SC_METRIC_L2) ? __builtin_inff() : -__builtin_inff();
                out_rows[o] = -1;
            }
        }
        if (b != SC_KEY_MAX)
            for (int i = tid; i < n; i += 256)
                if (keys[i] == b) keys[i] = SC_KEY_MAX;
        __syncthreads();
    }
}

void sc_launch_ivf_bound(int metric, const uint64_t* ekeysA, int kpa, int k, const float* qnorm, const unsigned* xmax_bits, int ld, float* thr, int Q, hipStream_t s,
                         const uint64_t* ekeysT, bool keep_min) {
    hipLaunchKernelGGL(ivf_bound_kernel, dim3((unsigned)Q), dim3(128), 0, s, metric, ekeysA, kpa, k, qnorm, xmax_bits, ld, thr, ekeysT, keep_min ? 1 : 0);
}
void sc_launch_ivf_candidates(int metric, const uint64_t* survA, const unsigned* cntA, const uint64_t* bestA, int kpa, const uint64_t* survB, const unsigned* cntB, int cap,
                              const float* thr, uint64_t* cand, int* ncand, int* flags, int Q, int wcap, hipStream_t s, int capB) {
    hipLaunchKernelGGL(ivf_candidates_kernel, dim3((unsigned)Q), dim3(256), 0, s, metric, survA, cntA, bestA, kpa, survB, cntB, cap, thr, cand, ncand, flags,
                       wcap < IVFW_CAP ? wcap : IVFW_CAP, capB > 0 ? capB : cap);
}
void sc_launch_ivf_refine_finalize(int metric, const uint64_t* ekeysA, int kpa, const uint64_t* ekeys, const int* ncand, const int* flags, int k, int64_t row_base,
                                   float* out_dist, int64_t* out_rows, int Q, hipStream_t s) {
    hipLaunchKernelGGL(ivf_refine_finalize_kernel, dim3((unsigned)Q), dim3(256), 0, s, metric, ekeysA, kpa, ekeys, ncand, flags, k, row_base, out_dist, out_rows, IVFW_CAP);
}
// the same for the exhaustive path (collect pass, wide / compacted final step; scan_batched.hip): ekeys [Q][stride] (0: sc_ivf_widen_cap()), any metric, no pre-scored block
void sc_launch_refine_finalize(int metric, const uint64_t* ekeys, const int* ncand, const int* flags, int k, int64_t row_base, float* out_dist, int64_t* out_rows,
                               int Q, hipStream_t s, int stride) {
    hipLaunchKernelGGL(ivf_refine_finalize_kernel, dim3((unsigned)Q), dim3(256), 0, s, metric, (const uint64_t*)nullptr, 0, ekeys, ncand, flags, k, row_base, out_dist,
                       out_rows, stride > 0 ? stride : IVFW_CAP);
}

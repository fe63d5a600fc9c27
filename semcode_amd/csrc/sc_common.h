// sc_common.h -- shared host/device definitions for libsemcode_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/semcode_hip.h"

#include <mutex>

#define SC_WAVE 64
#define SC_LD_ALIGN 64            // corpus row stride is a multiple of 64 floats (256 B)
#define SC_KEY_MAX 0xFFFFFFFFFFFFFFFFull

// sc_launch_gemm_bf16: ldc value that selects the 64-column-blocked output layout C[N / 64][M][64] (gemm_bf16.hip c_index);
// sc_launch_attention reads a QKV buffer of that layout when `blocked` (= that GEMM's M) is non-zero
#define SC_LDC_BLOCKED64 (-64)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ static inline uint64_t sc_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Irwin-Hall(12) integer-hash N(0,1) approximation; bit-exact twin: oracle/sc_oracle.c sc_oracle_synth.
__host__ __device__ static inline float sc_synth_value(uint64_t key, uint64_t row, uint32_t col, uint32_t dim) {
    const uint64_t ctr = (row * (uint64_t)dim + col) * 3ull;
    int32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        uint64_t h = sc_mix64(key ^ ((ctr + (uint64_t)j) * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull));
        sum += (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) + (int32_t)((h >> 32) & 0xFFFF) + (int32_t)(h >> 48);
    }
    return (float)(sum - 393210) * (1.0f / 65536.0f);
}
__host__ __device__ static inline uint64_t sc_synth_key(uint64_t seed) { return sc_mix64(seed + 0x9E3779B97F4A7C15ull); }

// score from the exact dot product and the two squared norms (metric: sc_metric)
template <int METRIC>
__host__ __device__ static inline float sc_score(float dot, float xn, float qn) {
    if (METRIC == SC_METRIC_L2) return fmaf(-2.0f, dot, xn + qn);
    if (METRIC == SC_METRIC_COSINE) return dot / (sqrtf(xn) * sqrtf(qn));
    return dot;
}

// 64-bit total order, smaller = better: (order-preserving float map of +-score) << 32 | row
template <int METRIC>
__host__ __device__ static inline uint64_t sc_make_key(float score, uint32_t row) {
    float v = (METRIC == SC_METRIC_L2) ? score : -score;
    v = v + 0.0f;
    uint32_t u = __builtin_bit_cast(uint32_t, v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((uint64_t)u << 32) | row;
}
__host__ __device__ static inline float sc_key_score(int metric, uint64_t key) {
    uint32_t u = (uint32_t)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float v = __builtin_bit_cast(float, u);
    return (metric == SC_METRIC_L2) ? v : -v;
}

// One-time, PER-DEVICE setup at a launch site.  hipFuncSetAttribute (the dynamic-LDS limit of a kernel) applies to the calling
// thread's current device only, so a process-wide "done" flag leaves every other device at the 64 KiB default: one process could
// not drive two devices (a service that owns a sharded collection).  Usage: static ScDeviceOnce once; sc_device_once(once, [&] { ... });
struct ScDeviceOnce {
    std::mutex mu;
    uint64_t done[4] = {0, 0, 0, 0};
};
template <class F>
static inline void sc_device_once(ScDeviceOnce& o, F&& f) {
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 255;
    std::lock_guard<std::mutex> lk(o.mu);
    if (!((o.done[d >> 6] >> (d & 63)) & 1)) {
        f();
        o.done[d >> 6] |= 1ull << (d & 63);
    }
}
// compute units of the calling thread's current device (cached per device)
static inline int sc_device_cus() {
    static std::mutex mu;
    static int cus[256] = {0};
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 255;
    std::lock_guard<std::mutex> lk(mu);
    if (!cus[d]) {
        hipDeviceProp_t prop;
        cus[d] = hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus[d];
}

// ---- launchers implemented in the .hip files (all enqueue on `s`, never synchronise) ----

// rows.hip: out [rows, ld] synthetic (+ optional |x|^2); tight [n, dim] -> padded rows (+ |x|^2); rows -> tight
void sc_launch_synth_fill(float* out, int64_t rows, int dim, int ld, uint64_t seed, int64_t first_row, float* xnorm, hipStream_t s);
void sc_launch_synth_clustered(float* out, int64_t rows, int dim, int ld, uint64_t seed, int64_t first_row, int nclusters, float spread,
                               float* xnorm, hipStream_t s);
void sc_launch_ingest_rows(const float* src, const int64_t* rows, int64_t first, int64_t n, int dim, float* dst, int ld,
                           float* xnorm, hipStream_t s);
void sc_launch_gather_rows(const float* src, int ld, int64_t first, int64_t n, int dim, float* dst, hipStream_t s);
// rows of row_bytes (a multiple of 4) by index: gather dst[i] = src[idx[i]], or scatter dst[idx[i]] = src[i]
void sc_launch_copy_rows_indexed(const void* src, void* dst, const int32_t* idx_dev, int n, size_t row_bytes, bool scatter, hipStream_t s);

struct ScanPlan {
    int qt;          // queries per group (<=16)
    int qstream;     // 0: queries resident in LDS; else streamed per k-chunk through a shared LDS ring that many stages deep
    int groups;      // ceil(Q / qt)
    int gstride;     // slots per group in qmap / partial when that differs from qt (0 = qt)
    int nwg;         // workgroups per group (grid.x)
    int cap;         // per-(wave,query) candidate capacity
    int lists;       // partial lists per query = nwg * 4
    size_t lds;      // dynamic LDS bytes
    size_t partial_bytes;
};
// returns false when (ld, k) cannot be served by the exact kernel
// n_rows > 0: size the grid for that many rows (small corpora), else one workgroup per CU
bool sc_scan_exact_plan(int ld, int Q, int k, int cus, ScanPlan* p, int force_qt = 0, int nprobe = 0, int64_t n_rows = 0);
// X [n, ld], xnorm [n]; Qp [Q, ld] zero padded, qnorm [Q]; partial: plan.partial_bytes
// perm: stored position -> reported row (NULL = identity); seg_*: IVF probe ranges per group (NULL = all rows);
// qmap: query slot -> row of Qp / qnorm (NULL = identity), see scan_exact.hip ScanArgs
void sc_launch_scan_exact(int metric, const float* X, const float* xnorm, int64_t n, int ld, const float* Qp,
                          const float* qnorm, int Q, int k, const ScanPlan& p, uint64_t* partial, const uint32_t* perm,
                          const int* seg_base, const int64_t* seg_rows, int nprobe, hipStream_t s, const int32_t* qmap = nullptr);
// scan_exact.hip, wide groups of list-major IVF probing: one workgroup = one list part [seg_rows[2g], seg_rows[2g+1]) x up to
// `width` (32 or 64) queries (qmap [groups][width], valid slots a prefix, -1 beyond); partial [groups][width][k] sorted keys
bool sc_scan_listgemm_supported(int ld, int k);
void sc_launch_scan_listgemm(int metric, int width, const float* X, const float* xnorm, int ld, const float* Qp, const float* qnorm, int k, int groups,
                             uint64_t* partial, const uint32_t* perm, const int64_t* seg_rows, const int32_t* qmap, hipStream_t s);
// partial [groups][lists][qt][k] sorted keys -> out_dist [Q,k], out_rows [Q,k]
// more lists than one LDS tree merge holds (2 * lists * k keys > 128 KiB) are merged in levels whose intermediate k-lists live right
// behind the partial lists: that many extra bytes (included in ScanPlan::partial_bytes)
size_t sc_topk_merge_scratch_bytes(int lists, int Q, int k);
void sc_launch_topk_merge(int metric, const uint64_t* partial, int groups, int lists, int qt, int Q, int k,
                          int64_t row_base, float* out_dist, int64_t* out_rows, hipStream_t s);
// lists_per_query sorted k-lists per query, list j of query q = partial[src[q * lists_per_query + j] * k ...] (src < 0: none);
// needs 2 * lists_per_query * k * 8 <= 128 KiB (sc_topk_gather_merge_supported)
bool sc_topk_gather_merge_supported(int lists_per_query, int k);
void sc_launch_topk_merge2(int metric, const float* d1, const int64_t* r1, const float* d2, const int64_t* r2, int k, float* out_dist, int64_t* out_rows, int Q,
                           hipStream_t s);
void sc_launch_topk_gather_merge(int metric, const uint64_t* partial, const int32_t* src, int lists_per_query, int Q, int k,
                                 int64_t row_base, float* out_dist, int64_t* out_rows, hipStream_t s);

// scan_batched.hip: bf16 / int8 shadows, coarse GEMM + filter phases, selection, exact re-rank
int sc_batched_kprime(void);   // candidates kept per query: bf16 stage
int sc_batched_kprime8(void);  // int8 stage
void sc_launch_shadow(const float* X, const float* xnorm, int64_t first, int64_t n, int ld, void* Xb, unsigned* res_bits, hipStream_t s);
void sc_launch_shadow8(const float* X, const float* xnorm, int64_t first, int64_t n, int ld, int ld8, void* Xq, float* xscale, unsigned* res_bits,
                       hipStream_t s);
void sc_launch_norm_max(const float* xnorm, int64_t n, unsigned* out_bits, hipStream_t s);
void sc_launch_query_bf16(const float* Qp, int Q, int Qpad, int ld, void* Qb, float* qres, hipStream_t s);
void sc_launch_query_i8(const float* Qp, int Q, int Qpad, int ld, int ld8, void* Qq, float* qscale, float* qres, unsigned* absmax_bits, hipStream_t s);
void sc_launch_scan_batched_init(float* thr, float* thr_fast, int qpad, uint64_t* best, unsigned* count, int* overflow, int Q, int kp, hipStream_t s);
void sc_launch_scan_coarse(int metric, const void* Xb, const float* xnorm, int64_t row0, int64_t row1, int ld, const void* Qb,
                           const float* qnorm, int Q, int Qpad, const float* thr, const float* thr_fast, uint64_t* surv, unsigned* count,
                           int cap, hipStream_t s, bool i8 = false, const float* xscale = nullptr, const float* qscale = nullptr,
                           bool dense = false,  // dense: most scores of this row range are expected to survive (loose thresholds)
                           void* hit_scratch = nullptr, size_t hit_bytes = 0);  // per-wave hit lists of the narrow kernel (batches of <= 64 queries)
void sc_launch_scan_select(int metric, uint64_t* surv, unsigned* count, int cap, uint64_t* best, const float* qnorm, float* thr,
                           float* thr_fast, int* overflow, int Q, int kp, hipStream_t s);
void sc_launch_scan_rerank(int metric, const float* X, const float* xnorm, int ld, const float* Qp, const float* qnorm, const uint64_t* best,
                           const float* thr, const unsigned* xnorm_max_bits, const float* qres, const int* overflow, int Q, int k,
                           int64_t row_base, const uint32_t* perm, float* out_dist, int64_t* out_rows, int* flags, hipStream_t s,
                           int kp, uint64_t* ekeys);

// ivf_coarse.hip + scan_batched.hip: the int8 coarse stage of list-major IVF probing (L2)
void sc_launch_ivf_center_shadow(const float* X, int64_t rows, int ld, int ld8, const float* C, int ldc, const int64_t* list_off, int nlist, void* Xc8,
                                 float* xrow, unsigned* list_stats, hipStream_t s, const float* xnorm = nullptr, const float* cnorm = nullptr, const int64_t* only = nullptr);  // norms: COSINE
void sc_launch_ivf_pair_query(const float* Qp, int ld, int ld8, const float* C, int ldc, const int32_t* slot_q, const int32_t* slot_l, int nslots,
                              const unsigned* list_stats, void* Qc8, float* slot_qs, float* slot_qnlb, float* slot_qb, float* slot_qd, float* slot_eps, hipStream_t s,
                              int metric = SC_METRIC_L2, const float* qnorm = nullptr, const float* cnorm = nullptr);
void sc_launch_ivf_slot_thr(const int32_t* slot_q, const float* slot_qnlb, const float* slot_eps, const float* thr, int nslots, float* slot_thr, float* slot_tf,
                            hipStream_t s);
void sc_launch_ivf_coarse(const void* Xc8, const float* xrow, int ld8, const void* Qc8, const void* items, int nitems, const float* slot_tf,
                          const float* slot_thr, const float* slot_qn, const float* slot_qs, const int32_t* slot_q, const float* slot_qb, const float* slot_qd,
                          uint64_t* surv, unsigned* count, int cap, void* hit_scratch, size_t hit_bytes, hipStream_t s,
                          const int32_t* slot_dst = nullptr, int metric = SC_METRIC_L2);  // non-null: dense phase (every row survives; count[] preset by the caller)
void sc_launch_scan_rerank_keys_l2(const float* X, const float* xnorm, int ld, const float* Qp, const float* qnorm, const uint64_t* cand, const int* ncand, int kp,
                                   const uint32_t* perm, uint64_t* ekeys, int Q, hipStream_t s);
void sc_launch_scan_rerank_keys(int metric, const float* X, const float* xnorm, int ld, const float* Qp, const float* qnorm, const uint64_t* cand, const int* ncand, int kp,
                                const uint32_t* perm, uint64_t* ekeys, int Q, hipStream_t s);
// the collect pass of the exhaustive batched path (scan_batched.hip): thresholds from the k-th exact score of the failed pass; survivor counts -> candidate counts
void sc_launch_scan_collect_bound(int metric, const float* prev_dist, int k, const float* qnorm, const float* qres, const unsigned* bits, int ld, float* thr,
                                  float* thr_fast, int* flags, int Q, hipStream_t s);
void sc_launch_scan_collect_counts(const unsigned* count, int cap, int* ncand, int* flags, int Q, hipStream_t s);
// thresholds from exact scores before the large phases (scan_batched.hip): ekeys [Q][kp <= 128] exact keys of the best coarse candidates
void sc_launch_scan_tighten(int metric, const uint64_t* ekeys, int kp, int k, const float* qnorm, const float* qres, const unsigned* bits, int ld, float* thr,
                            float* thr_fast, float* thr_cut, int Q, hipStream_t s);
void sc_launch_scan_thr_min(float* thr, const float* thr_cut, int Q, hipStream_t s);
// the wide candidate set (scan_batched.hip): best [Q][kcap] with nbest[q] keys, every key within thr_cut kept
void sc_launch_scan_select_wide(int metric, const uint64_t* surv, unsigned* count, int cap, uint64_t* best, unsigned* nbest, int kcap, int kp, const float* qnorm,
                                float* thr, float* thr_fast, const float* thr_cut, int* overflow, int Q, hipStream_t s);
void sc_launch_scan_wide_compact(int metric, const uint64_t* best, const unsigned* nbest, int kcap, const float* thr, uint64_t* cand, int* ncand, int Q, hipStream_t s);
void sc_launch_scan_wide_certify(int metric, const float* out_dist, int k, const float* qnorm, const float* qres, const unsigned* bits, int ld, const float* thr,
                                 const int* overflow, int* flags, int Q, hipStream_t s);
void sc_launch_fill_u32(unsigned* p, unsigned v, int n, hipStream_t s);
void sc_launch_refine_finalize(int metric, const uint64_t* ekeys, const int* ncand, const int* flags, int k, int64_t row_base, float* out_dist, int64_t* out_rows,
                               int Q, hipStream_t s, int stride = 0);
int sc_ivf_widen_cap(void);
void sc_launch_ivf_bound(int metric, const uint64_t* ekeysA, int kpa, int k, const float* qnorm, const unsigned* xmax_bits, int ld, float* thr, int Q, hipStream_t s,
                         const uint64_t* ekeysT = nullptr, bool keep_min = false);
void sc_launch_ivf_candidates(int metric, const uint64_t* survA, const unsigned* cntA, const uint64_t* bestA, int kpa, const uint64_t* survB, const unsigned* cntB, int cap,
                              const float* thr, uint64_t* cand, int* ncand, int* flags, int Q, int wcap, hipStream_t s, int capB = 0);
void sc_launch_ivf_refine_finalize(int metric, const uint64_t* ekeysA, int kpa, const uint64_t* ekeys, const int* ncand, const int* flags, int k, int64_t row_base,
                                   float* out_dist, int64_t* out_rows, int Q, hipStream_t s);

// ivf.hip
void sc_launch_ivf_plan(const int64_t* probe_rows, int Q, int nprobe, const int64_t* list_off, int nlist, int* seg_base, int64_t* seg_rows,
                        hipStream_t s);
void sc_launch_centroid_mean(const float* X, int ld, int dim, const int64_t* members, const int64_t* member_off, int nlist, float* C_tight,
                             const float* C_old, int ldc_old, hipStream_t s);
void sc_launch_permute_rows(const float* X, const float* xnorm, const uint32_t* perm, int64_t n, int ld, float* Xo, float* xnorm_o, hipStream_t s);
void sc_launch_rows_to_sample(const float* X, int ld, int dim, const int64_t* rows, int64_t n, float* out_tight, hipStream_t s);
void sc_launch_reseed_centroids(float* C_tight, int dim, const int32_t* moves_dev, int m, hipStream_t s);

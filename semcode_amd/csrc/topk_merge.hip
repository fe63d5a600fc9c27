// topk_merge.hip -- reduce per-wave sorted top-k key lists to the final [Q, k] result (gfx950).
//
// Input  : partial[group][list][slot][k] 64-bit keys (sc_common.h sc_make_key), each list sorted
//          ascending (= best first), padded with SC_KEY_MAX.
// Output : out_dist[q][j] (score decoded from the key), out_rows[q][j] = row_base + row, best first;
//          unfilled positions get row -1 and +inf (L2) / -inf (IP, COSINE).
// One workgroup per (query, chunk of lists): tree merge in LDS, in levels when the lists do not fit at once.
//
// Replaces (reference): the reduce step inside Milvus' segment search (server side of
// src/semcode/storage/milvus_store.py:141-147).  Latency-bound; not a roofline kernel.
#include "sc_common.h"

#define MERGE_THREADS 256

// Tree merge in LDS: all `lists` sorted k-lists of one query are loaded once, then halved log2(lists) times; in each
// round a thread merges two sorted lists with two pointers (k steps).  Used whenever 2 * lists * k keys fit in LDS.
// lsrc != NULL (list-major IVF probing): list l of query q is the k keys at partial + lsrc[q * lists + l] * k, or empty if negative.
__global__ __launch_bounds__(MERGE_THREADS) void topk_tree_merge_kernel(int metric, const uint64_t* __restrict__ partial, int lists, int qt, int Q,
                                                                         int k, int64_t row_base, float* __restrict__ out_dist,
                                                                         int64_t* __restrict__ out_rows, const int32_t* __restrict__ lsrc,
                                                                         int chunk, uint64_t* __restrict__ out_keys) {
    extern __shared__ __attribute__((aligned(16))) uint64_t tm_lds[];  // [2][nl * k]
    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const int grp = q / qt, slot = q - grp * qt;
    // chunk > 0 (a level of the multi-level merge, blockIdx.y = chunk index): merge lists [l0, l0 + nl) and write the k keys to
    // out_keys[q][blockIdx.y][k] -- the layout the next level reads with qt = 1
    const int l0 = chunk > 0 ? (int)blockIdx.y * chunk : 0;
    const int nl = chunk > 0 ? (lists - l0 < chunk ? lists - l0 : chunk) : lists;
    const uint64_t* base = partial + ((size_t)grp * lists * qt + slot) * (size_t)k;
    const size_t lstride = (size_t)qt * k;
    uint64_t* buf0 = tm_lds;
    uint64_t* buf1 = tm_lds + (size_t)nl * k;
    for (int i = tid; i < nl * k; i += MERGE_THREADS) {
        const int l = i / k, j = i - l * k;
        if (lsrc) {
            const int v = lsrc[(size_t)q * lists + l0 + l];
            buf0[i] = v >= 0 ? partial[(size_t)v * k + j] : SC_KEY_MAX;
        } else {
            buf0[i] = base[(size_t)(l0 + l) * lstride + j];
        }
    }
    __syncthreads();
    int n = nl;
    uint64_t* src = buf0;
    uint64_t* dst = buf1;
    while (n > 1) {
        const int pairs = (n + 1) >> 1;
        for (int t = tid; t < pairs; t += MERGE_THREADS) {
            const uint64_t* A = src + (size_t)(2 * t) * k;
            uint64_t* O = dst + (size_t)t * k;
            if (2 * t + 1 >= n) {
                for (int j = 0; j < k; ++j) O[j] = A[j];
            } else {
                const uint64_t* B = A + k;
                int ia = 0, ib = 0;
                for (int j = 0; j < k; ++j) {  // both lists hold k entries (SC_KEY_MAX padded), so neither index runs past k before j does
                    const uint64_t x = A[ia], y = B[ib];
                    const bool ta = x <= y;
                    O[j] = ta ? x : y;
                    ia += ta ? 1 : 0;
                    ib += ta ? 0 : 1;
                }
            }
        }
        __syncthreads();
        uint64_t* tmp = src; src = dst; dst = tmp;
        n = pairs;
    }
    if (out_keys) {
        uint64_t* ok = out_keys + ((size_t)q * gridDim.y + blockIdx.y) * (size_t)k;
        for (int j = tid; j < k; j += MERGE_THREADS) ok[j] = nl > 0 ? src[j] : SC_KEY_MAX;
        return;
    }
    for (int j = tid; j < k; j += MERGE_THREADS) {
        const size_t o = (size_t)q * k + j;
        const uint64_t key = nl > 0 ? src[j] : SC_KEY_MAX;
        if (key != SC_KEY_MAX) {
            out_dist[o] = sc_key_score(metric, key);
            out_rows[o] = row_base + (int64_t)(uint32_t)key;
        } else {
            out_dist[o] = (metric == SC_METRIC_L2) ? __builtin_inff() : -__builtin_inff();
            out_rows[o] = -1;
        }
    }
}

static ScDeviceOnce g_tree_attr_once;

// Lists whose 2 x k keys fit the LDS of one tree merge; more lists go through levels of such merges (chunks of SC_MERGE_FIT lists ->
// one k-list each, written behind the partial lists: sc_topk_merge_scratch_bytes), which replaced a serial fallback that made
// k = 33 twice and k = 1024 fifty times slower than k = 32.
static inline int merge_fit(int k) { const int f = (128 * 1024) / (2 * 8 * k); return f < 2 ? 2 : f; }
size_t sc_topk_merge_scratch_bytes(int lists, int Q, int k) {
    size_t extra = 0;
    const int fit = merge_fit(k);
    for (int cl = lists; cl > fit;) {
        const int ch = (cl + fit - 1) / fit;
        extra += (size_t)Q * ch * k * sizeof(uint64_t);
        cl = ch;
    }
    return extra;
}
void sc_launch_topk_merge(int metric, const uint64_t* partial, int groups, int lists, int qt, int Q, int k, int64_t row_base,
                          float* out_dist, int64_t* out_rows, hipStream_t s) {
    sc_device_once(g_tree_attr_once, [&] { hipFuncSetAttribute(reinterpret_cast<const void*>(topk_tree_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); });
    const int fit = merge_fit(k);
    const uint64_t* src = partial;
    // scratch of the levels: right behind the [groups][lists][qt][k] partial lists (the plan's partial_bytes includes it)
    uint64_t* scratch = const_cast<uint64_t*>(partial) + (size_t)groups * lists * qt * k;
    int cl = lists, cqt = qt;
    while (cl > fit) {
        const int ch = (cl + fit - 1) / fit;
        hipLaunchKernelGGL(topk_tree_merge_kernel, dim3((unsigned)Q, (unsigned)ch), dim3(MERGE_THREADS), (size_t)2 * fit * k * sizeof(uint64_t), s, metric,
                           src, cl, cqt, Q, k, row_base, out_dist, out_rows, (const int32_t*)nullptr, fit, scratch);
        src = scratch;
        scratch += (size_t)Q * ch * k;
        cl = ch;
        cqt = 1;
    }
    hipLaunchKernelGGL(topk_tree_merge_kernel, dim3((unsigned)Q), dim3(MERGE_THREADS), (size_t)2 * (cl > 0 ? cl : 1) * k * sizeof(uint64_t), s, metric, src,
                       cl, cqt, Q, k, row_base, out_dist, out_rows, (const int32_t*)nullptr, 0, (uint64_t*)nullptr);
}

bool sc_topk_gather_merge_supported(int lists_per_query, int k) {
    return lists_per_query >= 1 && k >= 1 && (size_t)2 * lists_per_query * k * sizeof(uint64_t) <= 128 * 1024;
}

void sc_launch_topk_gather_merge(int metric, const uint64_t* partial, const int32_t* src, int lists_per_query, int Q, int k, int64_t row_base,
                                 float* out_dist, int64_t* out_rows, hipStream_t s) {
    sc_device_once(g_tree_attr_once, [&] { hipFuncSetAttribute(reinterpret_cast<const void*>(topk_tree_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); });
    const size_t tree_lds = (size_t)2 * lists_per_query * k * sizeof(uint64_t);
    hipLaunchKernelGGL(topk_tree_merge_kernel, dim3((unsigned)Q), dim3(MERGE_THREADS), tree_lds, s, metric, partial, lists_per_query, 1, Q, k,
                       row_base, out_dist, out_rows, src, 0, (uint64_t*)nullptr);
}

// Two result sets of one batch, each [Q][k] best first (unfilled: row -1), over DISJOINT rows -> the best k of their union, same order
// rule (score, then row id).  One thread per query: the lists are short.  Used where a trained IVF index answers from its lists and
// from the tail of rows appended since the lists were laid out (sc_api.cpp).
__global__ __launch_bounds__(64) void topk_merge2_kernel(int metric, const float* __restrict__ d1, const int64_t* __restrict__ r1, const float* __restrict__ d2,
                                                          const int64_t* __restrict__ r2, int k, float* __restrict__ out_dist, int64_t* __restrict__ out_rows, int Q) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= Q) return;
    const size_t o = (size_t)q * k;
    int a = 0, b = 0;
    for (int j = 0; j < k; ++j) {
        const bool ha = a < k && r1[o + a] >= 0, hb = b < k && r2[o + b] >= 0;
        bool take_a;
        if (ha && hb) {
            const float x = d1[o + a], y = d2[o + b];
            const bool better = metric == SC_METRIC_L2 ? x < y : x > y;
            take_a = better || (x == y && r1[o + a] < r2[o + b]);
        } else {
            take_a = ha;
        }
        if (!ha && !hb) {
            out_dist[o + j] = (metric == SC_METRIC_L2) ? __builtin_inff() : -__builtin_inff();
            out_rows[o + j] = -1;
        } else if (take_a) {
            out_dist[o + j] = d1[o + a];
            out_rows[o + j] = r1[o + a];
            ++a;
        } else {
            out_dist[o + j] = d2[o + b];
            out_rows[o + j] = r2[o + b];
            ++b;
        }
    }
}
void sc_launch_topk_merge2(int metric, const float* d1, const int64_t* r1, const float* d2, const int64_t* r2, int k, float* out_dist, int64_t* out_rows, int Q,
                           hipStream_t s) {
    hipLaunchKernelGGL(topk_merge2_kernel, dim3((unsigned)((Q + 63) / 64)), dim3(64), 0, s, metric, d1, r1, d2, r2, k, out_dist, out_rows, Q);
}

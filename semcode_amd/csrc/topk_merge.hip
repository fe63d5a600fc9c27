// topk_merge.hip -- reduce per-wave sorted top-k key lists to the final [Q, k] result (gfx950).
//
// Input  : partial[group][list][slot][k] 64-bit keys (sc_common.h sc_make_key), each list sorted
//          ascending (= best first), padded with SC_KEY_MAX.
// Output : out_dist[q][j] (score decoded from the key), out_rows[q][j] = row_base + row, best first;
//          unfilled positions get row -1 and +inf (L2) / -inf (IP, COSINE).
// One workgroup per query.  The k-th key of ANY full list bounds the global k-th key from above, so a
// first pass takes thr0 = min over lists of list[k-1] and the second pass only appends keys <= thr0
// (typically a few dozen); a block-wide rank sort orders the survivors.  If the append buffer ever
// fills, it is compacted (rank sort, keep k, tighten the threshold) and the pass continues.
//
// Replaces (reference): the reduce step inside Milvus' segment search (server side of
// src/semcode/storage/milvus_store.py:141-147).  Latency-bound; not a roofline kernel.
#include "sc_common.h"

#define MERGE_THREADS 256
#define MERGE_CAP 2048

static __device__ void block_compact(volatile uint64_t* cand, volatile uint64_t* tmp, volatile unsigned* cnt, volatile uint64_t* thr,
                                     int k, int tid) {
    __syncthreads();
    const int n = (int)*cnt;
    for (int e = tid; e < n; e += MERGE_THREADS) {
        const uint64_t key = cand[e];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (cand[j] < key) ? 1 : 0;
        if (rank < k) tmp[rank] = key;
    }
    __syncthreads();
    const int m = n < k ? n : k;
    for (int e = tid; e < m; e += MERGE_THREADS) cand[e] = tmp[e];
    if (tid == 0) {
        *cnt = (unsigned)m;
        if (n >= k) *thr = tmp[k - 1];
    }
    __syncthreads();
}

__global__ __launch_bounds__(MERGE_THREADS) void topk_merge_kernel(int metric, const uint64_t* __restrict__ partial, int lists, int qt,
                                                                    int Q, int k, int64_t row_base, float* __restrict__ out_dist,
                                                                    int64_t* __restrict__ out_rows) {
    __shared__ uint64_t cand[MERGE_CAP];
    __shared__ uint64_t tmp[MERGE_CAP];
    __shared__ uint64_t red[MERGE_THREADS];
    __shared__ uint64_t thr_s;
    __shared__ unsigned cnt_s;
    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const int grp = q / qt, slot = q - grp * qt;
    const uint64_t* base = partial + ((size_t)grp * lists * qt + slot) * (size_t)k;
    const size_t lstride = (size_t)qt * k;

    // pass 1: thr0 = min_l list_l[k-1]
    uint64_t m = SC_KEY_MAX;
    for (int l = tid; l < lists; l += MERGE_THREADS) {
        const uint64_t v = base[(size_t)l * lstride + (k - 1)];
        m = v < m ? v : m;
    }
    red[tid] = m;
    if (tid == 0) cnt_s = 0;
    __syncthreads();
    for (int s = MERGE_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = red[tid + s] < red[tid] ? red[tid + s] : red[tid];
        __syncthreads();
    }
    if (tid == 0) thr_s = red[0];
    __syncthreads();

    // pass 2: append survivors
    const int64_t total = (int64_t)lists * k;
    for (int64_t i0 = 0; i0 < total; i0 += MERGE_THREADS) {
        const int64_t i = i0 + tid;
        if (i < total) {
            const int l = (int)(i / k), j = (int)(i - (int64_t)l * k);
            const uint64_t key = base[(size_t)l * lstride + j];
            if (key != SC_KEY_MAX && key <= *(volatile uint64_t*)&thr_s) {
                const unsigned pos = atomicAdd(&cnt_s, 1u);
                cand[pos] = key;
            }
        }
        __syncthreads();
        if (*(volatile unsigned*)&cnt_s > MERGE_CAP - MERGE_THREADS) block_compact(cand, tmp, &cnt_s, &thr_s, k, tid);
    }
    block_compact(cand, tmp, &cnt_s, &thr_s, k, tid);
    const int have = (int)cnt_s;
    for (int j = tid; j < k; j += MERGE_THREADS) {
        const size_t o = (size_t)q * k + j;
        if (j < have) {
            const uint64_t key = cand[j];
            out_dist[o] = sc_key_score(metric, key);
            out_rows[o] = row_base + (int64_t)(uint32_t)key;
        } else {
            out_dist[o] = (metric == SC_METRIC_L2) ? __builtin_inff() : -__builtin_inff();
            out_rows[o] = -1;
        }
    }
}

// Tree merge in LDS: all `lists` sorted k-lists of one query are loaded once, then halved log2(lists) times; in each
// round a thread merges two sorted lists with two pointers (k steps).  Used whenever 2 * lists * k keys fit in LDS.
// lsrc != NULL (list-major IVF probing): list l of query q is the k keys at partial + lsrc[q * lists + l] * k, or empty if negative.
__global__ __launch_bounds__(MERGE_THREADS) void topk_tree_merge_kernel(int metric, const uint64_t* __restrict__ partial, int lists, int qt, int Q,
                                                                         int k, int64_t row_base, float* __restrict__ out_dist,
                                                                         int64_t* __restrict__ out_rows, const int32_t* __restrict__ lsrc) {
    extern __shared__ __attribute__((aligned(16))) uint64_t tm_lds[];  // [2][lists * k]
    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const int grp = q / qt, slot = q - grp * qt;
    const uint64_t* base = partial + ((size_t)grp * lists * qt + slot) * (size_t)k;
    const size_t lstride = (size_t)qt * k;
    uint64_t* buf0 = tm_lds;
    uint64_t* buf1 = tm_lds + (size_t)lists * k;
    for (int i = tid; i < lists * k; i += MERGE_THREADS) {
        const int l = i / k, j = i - l * k;
        if (lsrc) {
            const int v = lsrc[(size_t)q * lists + l];
            buf0[i] = v >= 0 ? partial[(size_t)v * k + j] : SC_KEY_MAX;
        } else {
            buf0[i] = base[(size_t)l * lstride + j];
        }
    }
    __syncthreads();
    int n = lists;
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
    for (int j = tid; j < k; j += MERGE_THREADS) {
        const size_t o = (size_t)q * k + j;
        const uint64_t key = lists > 0 ? src[j] : SC_KEY_MAX;
        if (key != SC_KEY_MAX) {
            out_dist[o] = sc_key_score(metric, key);
            out_rows[o] = row_base + (int64_t)(uint32_t)key;
        } else {
            out_dist[o] = (metric == SC_METRIC_L2) ? __builtin_inff() : -__builtin_inff();
            out_rows[o] = -1;
        }
    }
}

static bool g_tree_attr_done = false;

void sc_launch_topk_merge(int metric, const uint64_t* partial, int groups, int lists, int qt, int Q, int k, int64_t row_base,
                          float* out_dist, int64_t* out_rows, hipStream_t s) {
    (void)groups;
    const size_t tree_lds = (size_t)2 * lists * k * sizeof(uint64_t);
    if (lists > 0 && tree_lds <= 128 * 1024) {
        if (!g_tree_attr_done) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(topk_tree_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            g_tree_attr_done = true;
        }
        hipLaunchKernelGGL(topk_tree_merge_kernel, dim3((unsigned)Q), dim3(MERGE_THREADS), tree_lds, s, metric, partial, lists, qt, Q, k, row_base,
                           out_dist, out_rows, (const int32_t*)nullptr);
        return;
    }
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)Q), dim3(MERGE_THREADS), 0, s, metric, partial, lists, qt, Q, k, row_base,
                       out_dist, out_rows);
}

bool sc_topk_gather_merge_supported(int lists_per_query, int k) {
    return lists_per_query >= 1 && k >= 1 && (size_t)2 * lists_per_query * k * sizeof(uint64_t) <= 128 * 1024;
}

void sc_launch_topk_gather_merge(int metric, const uint64_t* partial, const int32_t* src, int lists_per_query, int Q, int k, int64_t row_base,
                                 float* out_dist, int64_t* out_rows, hipStream_t s) {
    if (!g_tree_attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(topk_tree_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        g_tree_attr_done = true;
    }
    const size_t tree_lds = (size_t)2 * lists_per_query * k * sizeof(uint64_t);
    hipLaunchKernelGGL(topk_tree_merge_kernel, dim3((unsigned)Q), dim3(MERGE_THREADS), tree_lds, s, metric, partial, lists_per_query, 1, Q, k,
                       row_base, out_dist, out_rows, src);
}

// ivf.hip -- IVF_FLAT support kernels (gfx950): probe planning, centroid update, list-major reordering.
//
// Replaces (reference): what Milvus does server-side for
//   create_index(IVF_FLAT, nlist)  src/semcode/storage/milvus_store.py:76-84   (k-means + list build)
//   search(params={nprobe})        src/semcode/storage/milvus_store.py:141-147 (probe + list scan)
// The distance work itself reuses scan_exact.hip (segment mode) and scan_batched.hip (row assignment);
// these kernels are bookkeeping and are bandwidth- or latency-bound.
#include "sc_common.h"

// one thread per query: turn its probed list ids into row ranges and consecutive 16-row tile ordinals
__global__ __launch_bounds__(256) void ivf_plan_kernel(const int64_t* __restrict__ probe_rows, int Q, int nprobe,
                                                        const int64_t* __restrict__ list_off, int nlist, int* __restrict__ seg_base,
                                                        int64_t* __restrict__ seg_rows) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    int t = 0;
    for (int j = 0; j < nprobe; ++j) {
        const int64_t l = probe_rows[(size_t)q * nprobe + j];
        int64_t first = 0, end = 0;
        if (l >= 0 && l < nlist) { first = list_off[l]; end = list_off[l + 1]; }
        seg_base[(size_t)q * (nprobe + 1) + j] = t;
        seg_rows[((size_t)q * nprobe + j) * 2] = first;
        seg_rows[((size_t)q * nprobe + j) * 2 + 1] = end;
        t += (int)((end - first + 15) >> 4);
    }
    seg_base[(size_t)q * (nprobe + 1) + nprobe] = t;
}

// centroid c = mean of its member rows, summed sequentially in f32 in member order (deterministic;
// restated by oracle/sc_oracle.c sc_oracle_centroid_mean); an empty cluster keeps its old centroid.
__global__ __launch_bounds__(256) void centroid_mean_kernel(const float* __restrict__ X, int ld, int dim, const int64_t* __restrict__ members,
                                                             const int64_t* __restrict__ member_off, float* __restrict__ C,
                                                             const float* __restrict__ C_old, int ldc_old) {
    const int c = blockIdx.x;
    const int64_t m0 = member_off[c], m1 = member_off[c + 1];
    for (int col = threadIdx.x; col < dim; col += 256) {
        float acc = 0.f;
        for (int64_t i = m0; i < m1; ++i) acc += X[members[i] * (int64_t)ld + col];
        C[(size_t)c * dim + col] = (m1 > m0) ? acc / (float)(m1 - m0) : C_old[(size_t)c * ldc_old + col];
    }
}

// Xo[pos] = X[perm[pos]] (one wave per row), xnorm likewise
__global__ __launch_bounds__(256) void permute_rows_kernel(const float* __restrict__ X, const float* __restrict__ xnorm,
                                                            const uint32_t* __restrict__ perm, int64_t n, int ld, float* __restrict__ Xo,
                                                            float* __restrict__ xnorm_o) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t pos = wave0; pos < n; pos += nwaves) {
        const int64_t src = perm[pos];
        const float* in = X + src * (int64_t)ld;
        float* out = Xo + pos * (int64_t)ld;
        for (int k0 = 4 * lane; k0 < ld; k0 += 256) *reinterpret_cast<f32x4*>(out + k0) = *reinterpret_cast<const f32x4*>(in + k0);
        if (lane == 0) xnorm_o[pos] = xnorm[src];
    }
}

// out[i, :dim] = X[rows[i], :dim] (tight)
__global__ __launch_bounds__(256) void rows_to_sample_kernel(const float* __restrict__ X, int ld, int dim, const int64_t* __restrict__ rows,
                                                              int64_t n, float* __restrict__ out) {
    const int64_t total = n * (int64_t)dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / dim;
        const int c = (int)(i - r * dim);
        out[i] = X[rows[r] * (int64_t)ld + c];
    }
}

void sc_launch_ivf_plan(const int64_t* probe_rows, int Q, int nprobe, const int64_t* list_off, int nlist, int* seg_base, int64_t* seg_rows,
                        hipStream_t s) {
    hipLaunchKernelGGL(ivf_plan_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, s, probe_rows, Q, nprobe, list_off, nlist, seg_base, seg_rows);
}
void sc_launch_centroid_mean(const float* X, int ld, int dim, const int64_t* members, const int64_t* member_off, int nlist, float* C_tight,
                             const float* C_old, int ldc_old, hipStream_t s) {
    hipLaunchKernelGGL(centroid_mean_kernel, dim3((unsigned)nlist), dim3(256), 0, s, X, ld, dim, members, member_off, C_tight, C_old, ldc_old);
}
void sc_launch_permute_rows(const float* X, const float* xnorm, const uint32_t* perm, int64_t n, int ld, float* Xo, float* xnorm_o, hipStream_t s) {
    if (n <= 0) return;
    int64_t blocks = (n + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(permute_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, X, xnorm, perm, n, ld, Xo, xnorm_o);
}
void sc_launch_rows_to_sample(const float* X, int ld, int dim, const int64_t* rows, int64_t n, float* out_tight, hipStream_t s) {
    if (n <= 0) return;
    int64_t blocks = (n * (int64_t)dim + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(rows_to_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, s, X, ld, dim, rows, n, out_tight);
}

// ---- re-seeding between Lloyd iterations (sc_ivf.cpp): move m copies centroid b onto the slot of a starved centroid e,
// the two copies pushed apart by the factor (1 +- 2^-10) with alternating sign over the dimensions.  Moves are applied
// strictly in order (a large centroid may be split several times), one workgroup, threads over the dimensions.
__global__ __launch_bounds__(256) void reseed_centroids_kernel(float* __restrict__ C, int dim, const int32_t* __restrict__ moves, int m) {
    const float eps = 1.0f / 1024.0f;
    for (int i = 0; i < m; ++i) {
        const int e = moves[2 * i], b = moves[2 * i + 1];
        float* ce = C + (size_t)e * dim;
        float* cb = C + (size_t)b * dim;
        for (int j = threadIdx.x; j < dim; j += 256) {
            const float sg = (j & 1) ? -eps : eps;
            const float v = cb[j];
            ce[j] = v * (1.0f + sg);
            cb[j] = v * (1.0f - sg);
        }
        __syncthreads();
    }
}

void sc_launch_reseed_centroids(float* C_tight, int dim, const int32_t* moves_dev, int m, hipStream_t s) {
    if (m > 0) hipLaunchKernelGGL(reseed_centroids_kernel, dim3(1), dim3(256), 0, s, C_tight, dim, moves_dev, m);
}

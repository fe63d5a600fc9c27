// rows.hip -- row ingestion kernels for the HBM-resident corpus (gfx950).
//
// Corpus layout: X[row * ld + c], f32, ld = dim rounded up to 64 floats (256 B) and zero padded,
// plus xnorm[row] = |x|^2.  Every kernel here is "one wave per row": lane l owns the four floats
// k = 256 j + 4 l + c (c = 0..3) of every 1 KiB slice j, which makes the row write one coalesced
// 16 B/lane store and fixes the summation order of |x|^2:
//     p[l][c] = fmaf chain over j,   s[l] = (p0 + p1) + (p2 + p3),   xor-butterfly 32,16,8,4,2,1.
// oracle/sc_oracle.c sc_oracle_sqnorm restates exactly this order.
//
// Replaces (reference): the vector column of Collection.upsert, src/semcode/storage/milvus_store.py:110-130.
#include "sc_common.h"

static __device__ __forceinline__ float wave_butterfly_sum(float s) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_xor(s, off, 64);
    return s;
}

// ------------------------------------------------------------------ synthetic rows
__global__ __launch_bounds__(256) void synth_rows_kernel(float* __restrict__ out, int64_t rows, int dim, int ld,
                                                          uint64_t key, int64_t first_row, float* __restrict__ xnorm) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave0; r < rows; r += nwaves) {
        float* o = out + r * (int64_t)ld;
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
        for (int k0 = 4 * lane; k0 < ld; k0 += 256) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                v[c] = (k0 + c < dim) ? sc_synth_value(key, (uint64_t)(first_row + r), (uint32_t)(k0 + c), (uint32_t)dim) : 0.0f;
            *reinterpret_cast<f32x4*>(o + k0) = v;
            p0 = fmaf(v[0], v[0], p0);
            p1 = fmaf(v[1], v[1], p1);
            p2 = fmaf(v[2], v[2], p2);
            p3 = fmaf(v[3], v[3], p3);
        }
        if (xnorm) {
            float s = wave_butterfly_sum((p0 + p1) + (p2 + p3));
            if (lane == 0) xnorm[r] = s;
        }
    }
}

// clustered variant: row r = centre(cluster(r)) + spread * noise(r); cluster(r) = hash(seed, r) % nclusters,
// centre values and noise from the same integer-hash generator (restated by the CPU oracle's synth_clustered)
__global__ __launch_bounds__(256) void synth_clustered_rows_kernel(float* __restrict__ out, int64_t rows, int dim, int ld, uint64_t key,
                                                                    uint64_t ckey, int64_t first_row, int nclusters, float spread,
                                                                    float* __restrict__ xnorm) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const uint64_t gr = (uint64_t)(first_row + r);
        const uint64_t cl = sc_mix64(ckey ^ (gr * 0x9E3779B97F4A7C15ull)) % (uint64_t)nclusters;
        float* o = out + r * (int64_t)ld;
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
        for (int k0 = 4 * lane; k0 < ld; k0 += 256) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                v[c] = (k0 + c < dim) ? fmaf(spread, sc_synth_value(key, gr, (uint32_t)(k0 + c), (uint32_t)dim),
                                             sc_synth_value(ckey, cl, (uint32_t)(k0 + c), (uint32_t)dim))
                                      : 0.0f;
            *reinterpret_cast<f32x4*>(o + k0) = v;
            p0 = fmaf(v[0], v[0], p0);
            p1 = fmaf(v[1], v[1], p1);
            p2 = fmaf(v[2], v[2], p2);
            p3 = fmaf(v[3], v[3], p3);
        }
        if (xnorm) {
            float s = wave_butterfly_sum((p0 + p1) + (p2 + p3));
            if (lane == 0) xnorm[r] = s;
        }
    }
}

void sc_launch_synth_clustered(float* out, int64_t rows, int dim, int ld, uint64_t seed, int64_t first_row, int nclusters, float spread,
                               float* xnorm, hipStream_t s) {
    if (rows <= 0) return;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(synth_clustered_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, rows, dim, ld, sc_synth_key(seed),
                       sc_synth_key(seed ^ 0xC1057E25ull), first_row, nclusters, spread, xnorm);
}

void sc_launch_synth_fill(float* out, int64_t rows, int dim, int ld, uint64_t seed, int64_t first_row, float* xnorm,
                          hipStream_t s) {
    if (rows <= 0) return;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(synth_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, rows, dim, ld, sc_synth_key(seed),
                       first_row, xnorm);
}

// ------------------------------------------------------------------ ingest (pad + norm, optional scatter)
// src: tight [n, dim]; destination row = rows ? rows[i] : first + i.
__global__ __launch_bounds__(256) void ingest_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ rows,
                                                           int64_t first, int64_t n, int dim, float* __restrict__ dst, int ld,
                                                           float* __restrict__ xnorm) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t i = wave0; i < n; i += nwaves) {
        const int64_t r = rows ? rows[i] : first + i;
        const float* in = src + i * (int64_t)dim;
        float* o = dst + r * (int64_t)ld;
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
        for (int k0 = 4 * lane; k0 < ld; k0 += 256) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (k0 + c < dim) ? in[k0 + c] : 0.0f;
            *reinterpret_cast<f32x4*>(o + k0) = v;
            p0 = fmaf(v[0], v[0], p0);
            p1 = fmaf(v[1], v[1], p1);
            p2 = fmaf(v[2], v[2], p2);
            p3 = fmaf(v[3], v[3], p3);
        }
        float s = wave_butterfly_sum((p0 + p1) + (p2 + p3));
        if (lane == 0 && xnorm) xnorm[r] = s;
    }
}

void sc_launch_ingest_rows(const float* src, const int64_t* rows, int64_t first, int64_t n, int dim, float* dst, int ld,
                           float* xnorm, hipStream_t s) {
    if (n <= 0) return;
    int64_t blocks = (n + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(ingest_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, rows, first, n, dim, dst, ld, xnorm);
}

// ------------------------------------------------------------------ gather back to a tight [n, dim] buffer
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int ld, int64_t first, int64_t n, int dim,
                                                           float* __restrict__ dst) {
    const int64_t total = n * (int64_t)dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / dim;
        const int c = (int)(i - r * dim);
        dst[i] = src[(first + r) * (int64_t)ld + c];
    }
}

void sc_launch_gather_rows(const float* src, int ld, int64_t first, int64_t n, int dim, float* dst, hipStream_t s) {
    if (n <= 0) return;
    int64_t blocks = (n * (int64_t)dim + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, ld, first, n, dim, dst);
}

// Rows of `words` 4-byte words moved by index: gather  dst[i] = src[idx[i]]  or scatter  dst[idx[i]] = src[i].  The sub-batches of
// uncertified queries (and their results on the way back) used to travel as one hipMemcpyAsync per query: ~9 us each, 3.3 ms for the
// 124 queries the IVF coarse stage hands back at config 5.
__global__ __launch_bounds__(256) void copy_rows_indexed_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const int32_t* __restrict__ idx, int n,
                                                                 int words, int scatter) {
    const int64_t total = (int64_t)n * words;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / words;
        const int c = (int)(i - r * words);
        const int64_t o = (int64_t)idx[r] * words + c;
        if (scatter) dst[o] = src[i];
        else dst[i] = src[o];
    }
}
void sc_launch_copy_rows_indexed(const void* src, void* dst, const int32_t* idx_dev, int n, size_t row_bytes, bool scatter, hipStream_t s) {
    if (n <= 0 || row_bytes == 0) return;
    const int words = (int)(row_bytes / 4);
    int64_t blocks = ((int64_t)n * words + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(copy_rows_indexed_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, idx_dev, n, words, scatter ? 1 : 0);
}

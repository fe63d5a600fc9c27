// gemm_bf16.hip -- C[M,N] = A[M,K] * W[N,K]^T (+ bias, + GELU | + residual), bf16 in, f32 accumulate,
// bf16 out, on the CDNA4 matrix cores (v_mfma_f32_16x16x32_bf16).  gfx950 only.
//
// Replaces (reference): the per-layer matmuls llama.cpp runs on CPU threads behind
// LlamaCppEmbeddings.embed_documents (src/semcode/embeddings/providers.py:69-100, called at
// src/semcode/services/indexer.py:150).  Both operands are "K-contiguous" (activations [M,K] and
// torch-Linear weights [N,K]), so A and W tiles are staged and read identically.
//
// Roofline: MFMA bf16 (2.5 PFLOP/s dense).  Algorithmic FLOPs = 2*M*N*K.
//
// Structure (128x128x64 tile, 4 waves as 2x2, each wave 64x64 = 4x4 MFMA tiles):
//   * both tiles are staged global -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 B/lane), two LDS
//     buffers, the load of K-step t+1 is issued before the MFMAs of K-step t, one barrier per K-step;
//   * LDS rows are 128 B (64 bf16); the 16-byte chunk c of row r is stored at chunk c ^ ((r>>1)&7)
//     (applied on the per-lane SOURCE address, because LDS-DMA writes linearly) which makes every
//     ds_read_b128 fragment read bank-conflict free;
//   * operands are swapped (MFMA A = weight rows, B = activation rows) so that each lane ends up with
//     4 consecutive output columns of one output row -> 8-byte bf16 stores, bias as one float4;
//   * workgroup ids are remapped so that the tiles of one 128-row panel run on the same XCD (shared A
//     panel stays in that XCD's L2).
#include <cstdlib>

#include "gemm_tile.h"

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RES = 2 };

struct GemmArgs {
    const bf16_t* A;   // [M, lda]
    const bf16_t* W;   // [N, ldw]
    const float* bias; // [N]
    const bf16_t* R;   // [M, ldr] residual (EPI_BIAS_RES)
    bf16_t* C;         // [M, ldc]
    int M, N, K, lda, ldw, ldr, ldc;
    int tiles_n;       // N / 128
    int ntiles;        // (M/128) * tiles_n
    int order;         // 256-tile kernels: tile walk (see gemm256_tile)
};

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A tile | W tile]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;

    const int tile = xcd_remap(blockIdx.x, a.ntiles);
    const int mt = tile / a.tiles_n, nt = tile - mt * a.tiles_n;
    const int m0 = mt * G_BM, n0 = nt * G_BN;

    f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_tile_mainloop(a.A, a.lda, m0, a.W, a.ldw, n0, a.K, smem, acc, w, lane);
    const int fr = lane & 15, fq = lane >> 4;

    // epilogue: acc[ni][mi][r] = C[m0 + wm*64 + mi*16 + fr][n0 + wn*64 + ni*16 + 4*fq + r]
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wn * 64 + ni * 16 + 4 * fq;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wm * 64 + mi * 16 + fr;
            f32x4 v = acc[ni][mi] + bv;
            if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752440f));
            }
            if (EPI == EPI_BIAS_RES) {
                const u16x4 rv = *reinterpret_cast<const u16x4*>(a.R + (size_t)m * a.ldr + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bf16_to_f32(rv[r]);
            }
            u16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f32_to_bf16(v[r]);
            *reinterpret_cast<u16x4*>(a.C + (size_t)m * a.ldc + n) = o;
        }
    }
}


// ---- 256 x 256 x 64 tile variant (gemm_tile.h, second half); used when M % 256 == 0 and N % 256 == 0.
// PERSIST: one workgroup per CU walks tiles blockIdx.x, + gridDim.x, ...: the (fire-and-forget) epilogue stores of
// one tile drain while the LDS-DMA of the next tile is already in flight.  The K loop is register-tight (256
// VGPRs), so everything the epilogue needs per lane is re-derived after the loop from an opaque copy of the lane id;
// otherwise hipcc hoists that address arithmetic above the K loop and spills inside it.
template <int EPI, int DBG>
static __device__ __forceinline__ void gemm256_tile(const GemmArgs& a, int vt, char* smem, int w, int lane) {
    const int wm = w >> 2, wn = w & 3;
    constexpr int STG_ROW = 64 * 4 + 16;  // bytes: 64 f32 + 16 B pad (conflict-free b128 writes)
    {
        // tile order: bit 0 of a.order = skip the XCD remap; a.order >> 1 = G: walk column-major inside groups of G row panels
        const int tile = (a.order & 1) ? vt : xcd_remap(vt, a.ntiles);
        int mt = tile / a.tiles_n, nt = tile - mt * a.tiles_n;
        const int G = a.order >> 1;
        if (G > 1) {
            const int tiles_m = a.ntiles / a.tiles_n;
            const int gsz = G * a.tiles_n, g = tile / gsz, r = tile - g * gsz;
            const int rows_here = (g * G + G <= tiles_m) ? G : tiles_m - g * G;  // last group may be short
            mt = g * G + r % rows_here;
            nt = r / rows_here;
        }
        const int m0 = __builtin_amdgcn_readfirstlane(mt * T_BM), n0 = __builtin_amdgcn_readfirstlane(nt * T_BN);

        f32x4 acc[4][8];  // [ni][mi]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm_tile256_mainloop<DBG & 3>(a.A, a.lda, m0, a.W, a.ldw, n0, a.K, smem, acc, w, lane);
        if (DBG & 4) {  // diagnostic: no epilogue, keep the accumulators alive
            float sink = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) sink += acc[i][j][0] + acc[i][j][3];
            if (sink == 12345.678f) a.C[0] = 1;
            return;
        }
        // ---- epilogue through LDS: the MFMA layout gives each lane 4 columns of 16 different rows (32-byte row
        // segments per store instruction); staging the wave's 128 x 64 f32 block in 4 passes of 32 rows lets it
        // leave as whole 128-byte row segments, 16 B per lane.  The pipeline buffers are free here: after the
        // main loop's last barrier no wave reads them again.  Staging is wave-private (LDS executes a wave's
        // instructions in order), so no barrier is needed inside the epilogue.
        int ln = lane;
        asm volatile("" : "+v"(ln));  // opaque: nothing below can be computed before the K loop
        __builtin_amdgcn_sched_barrier(0);
        const int fr = ln & 15, fq = ln >> 4;
        const int prow = ln >> 3, c8 = (ln & 7) * 8;
        char* stg = smem + w * (32 * STG_ROW);
        f32x4 bias4[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bias4[ni] = *reinterpret_cast<const f32x4*>(a.bias + n0 + wn * 64 + ni * 16 + 4 * fq);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    f32x4 v = acc[ni][2 * p + h] + bias4[ni];
                    if (EPI == EPI_BIAS_GELU) {
                        const f32x2 g0 = gelu_erf_fast2(f32x2{v[0], v[1]}), g1 = gelu_erf_fast2(f32x2{v[2], v[3]});
                        v = f32x4{g0[0], g0[1], g1[0], g1[1]};
                    }
                    *reinterpret_cast<f32x4*>(stg + (h * 16 + fr) * STG_ROW + (ni * 16 + 4 * fq) * 4) = v;
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rl = j * 8 + prow;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + rl * STG_ROW + c8 * 4);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + rl * STG_ROW + c8 * 4 + 16);
                const size_t m = (size_t)(m0 + wm * 128 + p * 32 + rl);
                const int n = n0 + wn * 64 + c8;
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (EPI == EPI_BIAS_RES) {
                    const bf16x8 rv = *reinterpret_cast<const bf16x8*>(a.R + m * a.ldr + n);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += bf16_to_f32((bf16_t)rv[r]);
                }
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                u32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = pack_bf16x2(v[2 * r], v[2 * r + 1]);
                *reinterpret_cast<u32x4*>(a.C + m * a.ldc + n) = o;
            }
        }
    }
}

template <int EPI, int DBG = 0>
__global__ __launch_bounds__(512) void gemm256_bf16_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 128 KiB
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    gemm256_tile<EPI, DBG>(a, blockIdx.x, smem, w, lane);
}

// one workgroup per CU walking tiles blockIdx.x, + gridDim.x, ...
template <int EPI>
__global__ __launch_bounds__(512) void gemm256_persistent_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 128 KiB
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll 1
    for (int vt = blockIdx.x; vt < a.ntiles; vt += gridDim.x) {
        if (vt != (int)blockIdx.x) __syncthreads();  // every wave is done reading its epilogue staging of the previous tile
        gemm256_tile<EPI, 0>(a, vt, smem, w, lane);
    }
}

// M, N multiples of 128; K multiple of 64; all leading dimensions multiples of 8 elements.
bool sc_gemm_bf16_supported(int M, int N, int K) { return M > 0 && N > 0 && K > 0 && (M % G_BM) == 0 && (N % G_BN) == 0 && (K % G_BK) == 0; }

static int g_gemm_order = 16;  // XCD remap + column-major walk inside groups of 8 row panels (profiles/r1n_gemm_tile_order.log)
void sc_gemm_set_order(int v) { g_gemm_order = v; }
static int g_gemm_dbg = 0;
static bool g_gemm_persist = false, g_gemm_nopersist = false;
void sc_gemm_set_debug(int v) { g_gemm_persist = (v == 8); g_gemm_nopersist = (v == 9); g_gemm_dbg = (v == 8 || v == 9) ? 0 : v; }
static bool g_force_tile128 = false;
void sc_gemm_force_tile128(bool on) { g_force_tile128 = on; }

void sc_launch_gemm_bf16(int epi, const void* A, int lda, const void* W, int ldw, const float* bias, const void* R, int ldr, void* C,
                         int ldc, int M, int N, int K, hipStream_t s) {
    GemmArgs a;
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.bias = bias; a.R = (const bf16_t*)R; a.C = (bf16_t*)C;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldr = ldr; a.ldc = ldc;
    static const char* env_order = getenv("SC_GEMM_ORDER");  // A/B experiments
    a.order = env_order ? atoi(env_order) : g_gemm_order;
    static bool attr_done = false;
    if ((M % T_BM) == 0 && (N % T_BN) == 0 && !g_force_tile128) {
        a.tiles_n = N / T_BN;
        a.ntiles = (M / T_BM) * a.tiles_n;
        const size_t lds256 = 4 * T_TILE_BYTES;  // 128 KiB
        static bool attr256 = false;
        if (!attr256) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_bf16_kernel<EPI_BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_bf16_kernel<EPI_BIAS_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_bf16_kernel<EPI_BIAS_RES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
            attr256 = true;
        }
        dim3 grid((unsigned)a.ntiles), block(512);
        if (g_gemm_dbg) {  // diagnostic variants (sc_diag_gemm_bench); results are meaningless
            static bool attrd = false;
            if (!attrd) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_bf16_kernel<EPI_BIAS, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_bf16_kernel<EPI_BIAS, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_bf16_kernel<EPI_BIAS, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_bf16_kernel<EPI_BIAS, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);

                attrd = true;
            }
            if (g_gemm_dbg == 1) hipLaunchKernelGGL((gemm256_bf16_kernel<EPI_BIAS, 1>), grid, block, lds256, s, a);
            else if (g_gemm_dbg == 2) hipLaunchKernelGGL((gemm256_bf16_kernel<EPI_BIAS, 2>), grid, block, lds256, s, a);
            else if (g_gemm_dbg == 4) hipLaunchKernelGGL((gemm256_bf16_kernel<EPI_BIAS, 4>), grid, block, lds256, s, a);

            else hipLaunchKernelGGL((gemm256_bf16_kernel<EPI_BIAS, 5>), grid, block, lds256, s, a);
            return;
        }
        // The persistent walk (one workgroup per CU) is kept for experiments only (SC_GEMM_PERSIST=1): in the isolated
        // microbench it wins 7-17 % on the GELU / residual epilogues (profiles/r1m_gemm_microbench.log), but inside the
        // encoder pipeline, A/B on one device, it loses 6 % end to end (16.95k -> 15.97k chunks/s).
        static const bool env_persist = getenv("SC_GEMM_PERSIST") != nullptr;
        if (g_gemm_persist || env_persist) {
            static int n_cus = 0;
            static bool attrp = false;
            if (!attrp) {
                int dev = 0;
                hipDeviceProp_t prop;
                n_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_persistent_kernel<EPI_BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_persistent_kernel<EPI_BIAS_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_persistent_kernel<EPI_BIAS_RES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
                attrp = true;
            }
            dim3 pgrid((unsigned)(a.ntiles < n_cus ? a.ntiles : n_cus));
            if (epi == EPI_BIAS_GELU) hipLaunchKernelGGL((gemm256_persistent_kernel<EPI_BIAS_GELU>), pgrid, block, lds256, s, a);
            else if (epi == EPI_BIAS_RES) hipLaunchKernelGGL((gemm256_persistent_kernel<EPI_BIAS_RES>), pgrid, block, lds256, s, a);
            else hipLaunchKernelGGL((gemm256_persistent_kernel<EPI_BIAS>), pgrid, block, lds256, s, a);
            return;
        }
        if (epi == EPI_BIAS_GELU) hipLaunchKernelGGL(gemm256_bf16_kernel<EPI_BIAS_GELU>, grid, block, lds256, s, a);
        else if (epi == EPI_BIAS_RES) hipLaunchKernelGGL(gemm256_bf16_kernel<EPI_BIAS_RES>, grid, block, lds256, s, a);
        else hipLaunchKernelGGL(gemm256_bf16_kernel<EPI_BIAS>, grid, block, lds256, s, a);
        return;
    }
    a.tiles_n = N / G_BN;
    a.ntiles = (M / G_BM) * a.tiles_n;
    const size_t lds = 4 * G_TILE_BYTES;  // 64 KiB
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BIAS_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BIAS_RES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid((unsigned)a.ntiles), block(256);
    if (epi == EPI_BIAS_GELU) hipLaunchKernelGGL(gemm_bf16_kernel<EPI_BIAS_GELU>, grid, block, lds, s, a);
    else if (epi == EPI_BIAS_RES) hipLaunchKernelGGL(gemm_bf16_kernel<EPI_BIAS_RES>, grid, block, lds, s, a);
    else hipLaunchKernelGGL(gemm_bf16_kernel<EPI_BIAS>, grid, block, lds, s, a);
}

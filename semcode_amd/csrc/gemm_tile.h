// gemm_tile.h -- the 128x128x64 bf16 MFMA tile machinery shared by gemm_bf16.hip (encoder GEMMs) and
// scan_batched.hip (coarse distance GEMM).  See gemm_bf16.hip for the design notes.
#pragma once
#include "sc_common.h"

#define G_BM 128
#define G_BN 128
#define G_BK 64
#define G_TILE_BYTES (128 * 64 * 2)  // 16 KiB per operand tile

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gbl_vptr;
typedef unsigned short bf16_t;
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
static __device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    // round to nearest even; NaN stays NaN
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}


// stage one 128 x 64 bf16 tile: 16 LDS-DMA pieces of 1 KiB, 4 per wave
static __device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ src, int ld, int row0, int k0, char* lds_tile, int w, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w * 4 + i;
        const int p = piece * 64 + lane;  // 16-byte chunk index in the tile image
        const int r = p >> 3, pos = p & 7;
        const int c = pos ^ ((r >> 1) & 7);
        const bf16_t* g = src + (size_t)(row0 + r) * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((gbl_vptr)g, (lds_vptr)(lds_tile + piece * 1024), 16, 0, 0);
    }
}

// acc[ni][mi] += W_tile(n0.., K) x A_tile(m0.., K): lane ends up with
// acc[ni][mi][r] = sum_k A[m0 + wm*64 + mi*16 + (lane&15)][k] * W[n0 + wn*64 + ni*16 + 4*(lane>>4) + r][k].
// smem: 64 KiB ([2][A tile | W tile]).  All 256 threads of the workgroup must call it.
static __device__ __forceinline__ void gemm_tile_mainloop(const bf16_t* __restrict__ A, int lda, int m0, const bf16_t* __restrict__ W, int ldw,
                                                           int n0, int K, char* smem, f32x4 (&acc)[4][4], int w, int lane) {
    const int wm = w >> 1, wn = w & 1;
    const int nk = K / G_BK;
    stage_tile(A, lda, m0, 0, smem, w, lane);
    stage_tile(W, ldw, n0, 0, smem + G_TILE_BYTES, w, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * (2 * G_TILE_BYTES);
        char* nxt = smem + ((kt + 1) & 1) * (2 * G_TILE_BYTES);
        if (kt + 1 < nk) {
            stage_tile(A, lda, m0, (kt + 1) * G_BK, nxt, w, lane);
            stage_tile(W, ldw, n0, (kt + 1) * G_BK, nxt + G_TILE_BYTES, w, lane);
        }
        const char* At = cur;
        const char* Wt = cur + G_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], wf[4];
            const int c = 4 * ks + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 64 + i * 16 + fr;
                const int rw = wn * 64 + i * 16 + fr;
                af[i] = *reinterpret_cast<const bf16x8*>(At + ra * 128 + ((c ^ ((ra >> 1) & 7)) << 4));
                wf[i] = *reinterpret_cast<const bf16x8*>(Wt + rw * 128 + ((c ^ ((rw >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// XCD-aware, bijective block remap: blocks b, b+8, ... share an XCD; each XCD gets a contiguous range
// of logical tiles (consecutive logical tiles share an operand panel, which then stays in that L2).
static __device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

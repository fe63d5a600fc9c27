// gemm_tile.h -- the 128x128x64 bf16 MFMA tile machinery shared by gemm_bf16.hip (encoder GEMMs) and
// scan_batched.hip (coarse distance GEMM).  See gemm_bf16.hip for the design notes.
#pragma once
#include <type_traits>

#include "sc_common.h"

#define G_BM 128
#define G_BN 128
#define G_BK 64
#define G_TILE_BYTES (128 * 64 * 2)  // 16 KiB per operand tile

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gbl_vptr;
typedef unsigned short bf16_t;
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two f32 -> packed bf16 pair (round to nearest even, NaN preserved): one v_cvt_pk_bf16_f32
static __device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hwbf16x2));
}
// erf-GELU with the Abramowitz-Stegun 7.1.26 rational erf (|abs err| <= 1.5e-7, far below bf16 resolution)
static __device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = p * t * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);  // 1 - erf(z)
    const float erf_abs = 1.0f - e;
    return 0.5f * x * (1.0f + (x < 0.f ? -erf_abs : erf_abs));
}

// two erf-GELUs at once with packed f32 math (v_pk_fma_f32 / v_pk_mul_f32); used in the GEMM epilogues, where the
// vector ALU is the bound (two waves per SIMD, 128 values per lane each): measured 6 us per 256x256 tile for the
// A&S 7.1.26 form (16 packed ops + 2 rcp + 2 exp per pair, profiles/r1o_gemm_trace.log).  This form needs 10 packed ops
// and one exp2 per value:
//     gelu(x) = max(x, 0) - |h| erfc(sqrt2 |h|),  h = x / 2,     erfc(sqrt2 y) = 2^Q(y)
// with Q a degree-7 fit of log2 erfc(sqrt2 y) on [0, 6] weighted by the sensitivity y erfc of the result (so the
// tails are relatively accurate too) and monotonically decreasing beyond the fit range (2^Q -> 0, no clamp needed).
// Evaluated in f32: |error| <= 2.5e-7 everywhere (the f32 rounding of the sum), <= 0.015 bf16 ulp wherever |gelu| > 1e-3 and <= 0.14 ulp down to 1e-5;
// tests/test_encoder_gpu.py::test_gelu_epilogue_accuracy checks it through the kernel.
static __device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x) {
    const f32x2 h = x * 0.5f;
    const f32x2 ah = {fabsf(h[0]), fabsf(h[1])};
    f32x2 q = __builtin_elementwise_fma(ah, (f32x2){-0.000139362935f, -0.000139362935f}, (f32x2){0.00303643686f, 0.00303643686f});
    q = __builtin_elementwise_fma(q, ah, (f32x2){-0.0268522501f, -0.0268522501f});
    q = __builtin_elementwise_fma(q, ah, (f32x2){0.131913051f, 0.131913051f});
    q = __builtin_elementwise_fma(q, ah, (f32x2){-0.428876668f, -0.428876668f});
    q = __builtin_elementwise_fma(q, ah, (f32x2){-1.83460581f, -1.83460581f});
    q = __builtin_elementwise_fma(q, ah, (f32x2){-2.30246782f, -2.30246782f});
    q = __builtin_elementwise_fma(q, ah, (f32x2){9.69476332e-06f, 9.69476332e-06f});
    const f32x2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    return __builtin_elementwise_fma(-ah, e, h + ah);
}

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

// =====================================================================================================
// 256 x 256 x 64 tile, 8 waves (2 along M x 4 along N, 128 x 64 outputs per wave), 1 workgroup per CU.
//
// Why: the 128^2 tile moves 1 byte L2->LDS per 64 FLOP and, with one K-tile in flight, ran at the
// latency of that fill (measured 630 TF).  This tile needs half the bytes per FLOP, keeps the NEXT TWO
// K-tiles in flight (2 LDS buffers + the tile being consumed lives in registers) and overlaps every LDS
// fragment read with MFMAs:
//
//     F0 = fragments of k-step 0 of tile t (already in registers)
//     loop t:  issue ds_reads  F1 <- k-step 1 of tile t            (LDS[t&1])
//              32 MFMAs on F0
//              wait my LDS reads (F1) and my LDS-DMA of tile t+1;  ONE barrier
//                    -> every wave is done with LDS[t&1] and tile t+1 is visible
//              issue LDS-DMA of tile t+2 -> LDS[t&1]               (lands during the next 64 MFMAs)
//              issue ds_reads  F0 <- k-step 0 of tile t+1          (LDS[(t+1)&1])
//              32 MFMAs on F1
// =====================================================================================================
#define T_BM 256
#define T_BN 256
#define T_TILE_BYTES (256 * 64 * 2)  // 32 KiB per operand tile
#define T_EPI_ROW 144                // epilogue staging row: 64 bf16 + 16 B pad
#define T_EPI_BYTES (8 * 16 * T_EPI_ROW)  // 8 waves x 16 rows, placed BEHIND the 128 KiB pipeline buffers
#define T_LDS_BYTES (4 * T_TILE_BYTES + T_EPI_BYTES)


template <int AUX = 0>  // cache policy of the LDS-DMA (2 = nt: a stream that should not displace what the other operand keeps in L2)
static __device__ __forceinline__ void stage_tile256(const bf16_t* __restrict__ src, int ld, int row0, int k0, char* lds_tile, int w, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w * 4 + i;      // 32 pieces of 1 KiB, 4 per wave
        const int p = piece * 64 + lane;  // 16-byte chunk index in the tile image
        const int r = p >> 3, pos = p & 7;
        const int c = pos ^ ((r >> 1) & 7);
        const bf16_t* g = src + (size_t)(row0 + r) * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((gbl_vptr)g, (lds_vptr)(lds_tile + piece * 1024), 16, 0, AUX);
    }
}

// fragments of one k-step (32 deep) for this wave: 4 W-side (MFMA A operand) + 8 A-side (MFMA B operand)
struct Frag256 {
    bf16x8 wf[4];
    bf16x8 af[8];
};
static __device__ __forceinline__ void read_frags256(const char* At, const char* Wt, int wm, int wn, int fr, int fq, int ks, Frag256& f) {
    const int c = 4 * ks + fq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rw = wn * 64 + i * 16 + fr;
        f.wf[i] = *reinterpret_cast<const bf16x8*>(Wt + rw * 128 + ((c ^ ((rw >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ra = wm * 128 + i * 16 + fr;
        f.af[i] = *reinterpret_cast<const bf16x8*>(At + ra * 128 + ((c ^ ((ra >> 1) & 7)) << 4));
    }
}
// I8: the operands are int8 (16 per lane instead of 8 bf16: the same 16 bytes, so LDS image, swizzle and fragment reads are
// byte for byte those of the bf16 tile with K counted in pairs of int8) and the accumulators hold i32 bit patterns:
// v_mfma_i32_16x16x64_i8 takes the cycles of the bf16 form at twice the K (MI355X guide, matrix-core table).
typedef int i32x4_t __attribute__((ext_vector_type(4)));
template <bool I8 = false>
static __device__ __forceinline__ void mfma_frags256(const Frag256& f, f32x4 (&acc)[4][8]) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            if (I8)
                acc[ni][mi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_t, f.wf[ni]), __builtin_bit_cast(i32x4_t, f.af[mi]),
                                                                                            __builtin_bit_cast(i32x4_t, acc[ni][mi]), 0, 0, 0));
            else
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wf[ni], f.af[mi], acc[ni][mi], 0, 0, 0);
        }
}

// acc[ni][mi][r] = sum_k A[m0 + wm*128 + mi*16 + (lane&15)][k] * W[n0 + wn*64 + ni*16 + 4*(lane>>4) + r][k]
// smem: 128 KiB ([2][A tile | W tile]).  All 512 threads of the workgroup must call it.  K % 64 == 0.
// a_kstep: elements from one K-tile of A to the next -- 64 for row-major A[M][lda]; M * 64 (with lda = 64) when A is stored in
// 64-column blocks [K / 64][M][64], as the producing GEMM's blocked output layout writes it (gemm_bf16.hip c_index).
// DBG (diagnostic builds only): bit 0 = skip the in-loop LDS-DMA, bit 1 = skip the MFMAs, bit 2 = skip the fragment reads
// (MFMAs run on whatever the registers hold: the pure matrix-pipe ceiling of this loop).  Two variants with valid results were
// A/B-tested on one box and dropped (gpurun_out/r2e: 20.8k -> 20.1k chunks/s each): s_setprio(1) around the MFMA blocks, and the
// A operand's in-loop LDS-DMA with the nt cache policy.
// Round 2 (profiles/r2n_gemm_k32_ab.log): this loop serialises more than it overlaps -- at 8k^3 it takes 884 us, its LDS-DMA requests,
// barriers and fragment reads alone 484, its MFMAs alone 582: one 64 KiB request is in flight between two barriers and its round
// trip (~1.3 us) exceeds the MFMA time of a K-tile (1.0 us).  A variant with 32-deep K-tiles in four stages (two requests in
// flight, 64-byte rows read without a swizzle, one barrier per 32 MFMAs) was built and is correct, but its own skeleton is slower
// (593 us: every 128-byte line is requested twice, twice the barriers): 8k^3 1.31 -> 1.17 PF, encoder 20.1k -> 15.6k chunks/s.
// Not kept.  Three stages of full 64-deep tiles do not fit 160 KiB next to the epilogue staging.
// Round 2 (profiles/r2n_gemm_k32_ab.log): at 8k^3 this loop takes 800 us without its epilogue; without its LDS-DMA 655, its MFMAs
// alone 574 (1.92 PF: the matrix pipe at the clock the part holds under this load), its LDS-DMA requests, barriers and fragment
// reads alone 477.  The fill is therefore not hidden -- it adds 145 us to the 655 -- although it would fit under them twice over.
// Two rearrangements were built, are correct on every encoder test, and did not change that: 32-deep K-tiles in four stages (two
// requests in flight, 64-byte rows read without a swizzle, a barrier per 32 MFMAs: 8k^3 1.31 -> 1.17 PF, encoder 20.1k -> 15.6k
// chunks/s -- its own skeleton is slower, every 128-byte line being requested twice), and this loop with the request moved to the
// start of the iteration behind a second barrier (1.5 iterations of lead instead of 1.0: 8k^3 +2 %, FFN2 -2.5 %, encoder -1.3 %).
// So the stall is neither a lack of requests in flight nor of lead time; what the MFMAs and the LDS-DMA writes share is the LDS
// itself (192 KiB of fragment reads + 64 KiB of DMA writes per K-tile) and the issue slots of the waves that request the pieces.
// Neither variant was kept.
// tail: called by every wave right after the LAST barrier of the loop, before the final 32 MFMAs.  From there on no wave
// reads the 128 KiB of pipeline buffers again, so the hook may start LDS-DMA into them for the epilogue (the residual
// tile, gemm_bf16.hip) and have it land under those MFMAs.
struct NoTailHook {
    static constexpr bool kCoop = false;
    __device__ __forceinline__ void operator()() const {}
    __device__ __forceinline__ void coop(int, int, const uint32_t (&)[2], const uint32_t (&)[2]) const {}
    __device__ __forceinline__ void prepare() {}
};
// request K-tiles 0 and 1 of a tile (16 LDS-DMA pieces per wave)
static __device__ __forceinline__ void gemm_tile256_prologue_issue(const bf16_t* __restrict__ A, int lda, int m0, const bf16_t* __restrict__ W,
                                                                    int ldw, int n0, int K, char* smem, int w, int lane, size_t a_kstep = G_BK) {
    stage_tile256(A, lda, m0, 0, smem, w, lane);
    stage_tile256(W, ldw, n0, 0, smem + T_TILE_BYTES, w, lane);
    if (K > G_BK) {
        stage_tile256(A + a_kstep, lda, m0, 0, smem + 2 * T_TILE_BYTES, w, lane);
        stage_tile256(W, ldw, n0, G_BK, smem + 3 * T_TILE_BYTES, w, lane);
    }
}
template <int DBG = 0, class TailHook = NoTailHook, bool I8 = false>
static __device__ __forceinline__ void gemm_tile256_mainloop(const bf16_t* __restrict__ A, int lda, int m0, const bf16_t* __restrict__ W, int ldw,
                                                              int n0, int K, char* smem, f32x4 (&acc)[4][8], int w, int lane,
                                                              TailHook tail = TailHook{}, size_t a_kstep = G_BK) {
    const int wm = w >> 2, wn = w & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int nk = K / G_BK;
    gemm_tile256_prologue_issue(A, lda, m0, W, ldw, n0, K, smem, w, lane, a_kstep);
    // tile 0 landed, tile 1 (8 pieces per wave) may still be in flight.
    // Raw barrier: __syncthreads() would put an s_waitcnt vmcnt(0) in front of it and wait for tile 1 as well.
    if (nk > 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    Frag256 f0, f1;
    using T = std::true_type;
    using F = std::false_type;
    read_frags256(smem, smem + T_TILE_BYTES, wm, wn, fr, fq, 0, f0);
    if (DBG & 4) read_frags256(smem, smem + T_TILE_BYTES, wm, wn, fr, fq, 1, f1);
    // One K-tile.  STAGE: a tile kt+2 exists and is requested; NEXT: a tile kt+1 exists and its k-step-0
    // fragments are fetched.  sched_barrier(0) pins the order hipcc would otherwise relax (it sinks the
    // register-only MFMAs below the barrier and the LDS-DMA issue below the next MFMA block, which halves
    // the time the DMA has to land).
    auto ktile = [&](int kt, auto stage_c, auto next_c) {
        constexpr bool STAGE = decltype(stage_c)::value, NEXT = decltype(next_c)::value;
        char* cur = smem + (kt & 1) * (2 * T_TILE_BYTES);
        char* nxt = smem + ((kt + 1) & 1) * (2 * T_TILE_BYTES);
        if (!(DBG & 4)) read_frags256(cur, cur + T_TILE_BYTES, wm, wn, fr, fq, 1, f1);
        if (!(DBG & 2)) mfma_frags256<I8>(f0, acc);
        else asm volatile("" ::"v"(f0.wf[0]), "v"(f0.af[0]), "v"(f0.wf[3]), "v"(f0.af[7]));
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my LDS-DMA of tile kt+1 (issued one K-tile ago)
        __syncthreads();                                   // + every wave's reads of `cur` are complete
        if (!STAGE && !NEXT) tail();
        if (STAGE && !(DBG & 1)) {
            stage_tile256(A + (size_t)(kt + 2) * a_kstep, lda, m0, 0, cur, w, lane);
            stage_tile256(W, ldw, n0, (kt + 2) * G_BK, cur + T_TILE_BYTES, w, lane);
        }
        if (NEXT && !(DBG & 4)) read_frags256(nxt, nxt + T_TILE_BYTES, wm, wn, fr, fq, 0, f0);
        if (!(DBG & 2)) mfma_frags256<I8>(f1, acc);
        else asm volatile("" ::"v"(f1.wf[0]), "v"(f1.af[0]), "v"(f1.wf[3]), "v"(f1.af[7]));
        if (STAGE && !(DBG & 1) && NEXT && !(DBG & 2) && !(DBG & 4)) {
            // spread the 8 LDS-DMA issues (each ~100 issue cycles with its address arithmetic) and the 12 fragment reads
            // between the 32 MFMAs instead of bursting them right after the barrier, where both waves of a SIMD would
            // stall the matrix pipe together: groups of {1 VMEM, 1-2 DS reads, 4 MFMA}
#define SC_SGB3(NDS)                                        \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   /* VMEM (LDS-DMA) */ \
    __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0); /* DS read */        \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   /* MFMA */
            SC_SGB3(2) SC_SGB3(2) SC_SGB3(2) SC_SGB3(2) SC_SGB3(1) SC_SGB3(1) SC_SGB3(1) SC_SGB3(1)
#undef SC_SGB3
        }
        if (!STAGE && !NEXT && !std::is_same<TailHook, NoTailHook>::value && !(DBG & 7)) {
            // the tail hook's 16 LDS-DMA issues, spread over the last 32 MFMAs the same way
#define SC_SGB2                                                              \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); /* VMEM (LDS-DMA) */ \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); /* MFMA */
            SC_SGB2 SC_SGB2 SC_SGB2 SC_SGB2 SC_SGB2 SC_SGB2 SC_SGB2 SC_SGB2 SC_SGB2 SC_SGB2 SC_SGB2 SC_SGB2 SC_SGB2 SC_SGB2 SC_SGB2 SC_SGB2
#undef SC_SGB2
        }
    };
    int kt = 0;
#pragma unroll 1
    for (; kt + 2 < nk; ++kt) ktile(kt, T{}, T{});
    if (kt + 1 < nk) { ktile(kt, F{}, T{}); ++kt; }
    ktile(kt, F{}, F{});
}

// =====================================================================================================
// Ping-pong form of the same 256 x 256 x 64 tile (round 3).  Same LDS budget, same accumulator layout, same order of
// accumulation per output (bit-identical results), different schedule:
//
//   * the two waves of a SIMD (w and w + 4: wm = 0 / 1) run ONE barrier apart: while one multiplies a quadrant of its
//     128 x 64 outputs (16 MFMAs) the other reads fragments and requests LDS-DMA pieces, then they swap.  In the
//     one-barrier loop above both waves of a SIMD are in the same place at the same time: both burst their LDS-DMA
//     requests (each blocks its wave's issue for 60-180 cycles) and both wait at the barrier, and the matrix pipe idles
//     (profiles/r2n_gemm_k32_ab.log: the fill adds 145 us to 655 at 8k^3 although it would fit under the MFMAs).
//   * a K-tile travels as FOUR half-tiles of 16 KiB, in the order they are read:
//         j = 0: W0 = the W rows of every wave's ni 0..1     (rows wn * 64 + [ 0, 32))
//         j = 1: A0 = the A rows of every wave's mi 0..3     (rows wm * 128 + [ 0, 64))
//         j = 2: A1 = ... mi 4..7                            (rows wm * 128 + [64, 128))
//         j = 3: W1 = ... ni 2..3                            (rows wn * 64 + [32, 64))
//     through a ring of 8 slots (the same 128 KiB).  Half-tile h = 4 t + j is read in phase h - 1 and requested in phase
//     h - D - 2: D half-tiles (2 D pieces per wave) are always in flight, waited for with a counted vmcnt once per phase.
//   * phases of K-tile t (P = 4 t + p), each {LOAD segment | barrier | MFMA segment | barrier}:
//         p  reads (ds_read_b128)      multiplies                    requests
//         0  A0(t)      8              Q(0,0) = A0 x W0              half-tile P + D + 2
//         1  A1(t)      8              Q(1,0) = A1 x W0                ''
//         2  W1(t)      4              Q(1,1) = A1 x W1                ''
//         3  W0(t + 1)  4              Q(0,1) = A0 x W1                ''
//     every fragment register is loaded in a LOAD segment in which its previous contents are dead: 96 fragment VGPRs, no
//     double buffering.
//   * hazards.  RAW: half-tile h was requested by all waves in phase h - D - 2; every wave waits for its own pieces
//     (vmcnt(2 D) after requesting in phase h - 2) before the barrier that ends that LOAD segment, wave group 1 one barrier
//     later than group 0, and the first read is in phase h - 1, behind both.  WAR: slot (h mod 8) is rewritten in phase
//     h + 6 - D at the earliest (D <= 5: two segments after wave group 1 read it, whose reads have returned before its
//     MFMA segment starts).
//   * tail hooks.  tail() runs in the LOAD segment of the very last phase, when no wave reads the ring any more (wave-private
//     LDS-DMA into the wave's own 16 KiB, as in the one-barrier loop).  A hook with kCoop instead continues the ring: once the
//     K-tiles' own requests have ended, slot after slot falls free (one per phase, in the order they were read) and
//     coop(slot) -- called by EVERY wave in each of the last 7 phases and once behind the loop -- requests that wave's two
//     pieces of whatever the epilogue wants in that slot (gemm_bf16.hip: the residual tile, slot s = the 128 x 64 part of wave
//     s).  Same 2 pieces per wave and phase as the loop's own requests, in flight while the last 1.5 K-tiles multiply; the
//     caller waits vmcnt(0) and passes ONE more barrier (all 8 waves) before reading, because every wave's pieces are in
//     every slot.  The counted waits include the hook's pieces (vmcnt counts in issue order).
//   * persistent use (PREF, scan_batched.hip).  A workgroup that walks several output tiles hands PPNextTileHook to the loop: its
//     coop() requests half-tiles 0 .. 7 (the first TWO K-tiles) of the NEXT output tile into the falling-free slots, in ring order
//     (they are the slots the next tile's loop expects when it starts with buffer parity par ^ (nk & 1)).  The next call (PREF)
//     then has no prologue requests: it waits for half-tiles 0 and 1 (vmcnt(12): six younger half-tiles; anything the caller's
//     epilogue issued in between is younger still and only makes the wait conservative), and its K-tile 0 requests nothing before
//     half-tile 8.  The memory latency of a tile's first bytes and two K-tiles of fill then hide under the previous tile's tail and
//     epilogue instead of opening every tile (coarse scan, 6 K-tiles per tile: profiles/r3g_coarse_trace.log).  K >= 192 there.
// K >= 128 (two K-tiles); the launchers fall back to the one-barrier loop otherwise.
// =====================================================================================================
// requests half-tile h (0 .. 7) of an output tile into `slot`: the cooperative hook of a persistent caller (see above)
struct PPNextTileHook {
    static constexpr bool kCoop = true;
    __amdgpu_buffer_rsrc_t ra, rw;  // the next tile's A rows / W rows
    uint32_t a_kbytes, a1_off, w1_off;
    int w;
    char* smem;
    __device__ __forceinline__ void coop(int slot, int h, const uint32_t (&va)[2], const uint32_t (&vw)[2]) const {
        const int t = h >> 2, j = h & 3;
        char* dst = smem + (slot * 16384 + 2 * w * 1024);
        if (j == 1 || j == 2) {
            const uint32_t so = (uint32_t)t * a_kbytes + (j == 2 ? a1_off : 0u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vptr)dst, 16, va[0], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vptr)(dst + 1024), 16, va[1], so, 0, 0);
        } else {
            const uint32_t so = (uint32_t)t * (uint32_t)(G_BK * 2) + (j == 3 ? w1_off : 0u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)dst, 16, vw[0], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)(dst + 1024), 16, vw[1], so, 0, 0);
        }
    }
    __device__ __forceinline__ void operator()() const {}
    __device__ __forceinline__ void prepare() {}
};
// per-lane source offsets of a wave's two pieces of an A / W half-tile (shared by the loop and by a persistent caller's first prefetch)
static __device__ __forceinline__ void pp_piece_offsets(int lda, int ldw, int w, int lane, uint32_t (&va)[2], uint32_t (&vw)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = (2 * w + i) * 8 + (lane >> 3), pos = lane & 7;
        const int c = pos ^ ((lr >> 1) & 7);  // LDS-DMA writes linearly: the swizzle is applied on the source address
        va[i] = (uint32_t)(((lr >> 6) * 128 + (lr & 63)) * lda * 2 + c * 16);
        vw[i] = (uint32_t)(((lr >> 5) * 64 + (lr & 31)) * ldw * 2 + c * 16);
    }
}
template <bool I8>
static __device__ __forceinline__ f32x4 pp_mma(const bf16x8& wv, const bf16x8& av, const f32x4& c) {
    if (I8)
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_t, wv), __builtin_bit_cast(i32x4_t, av),
                                                                              __builtin_bit_cast(i32x4_t, c), 0, 0, 0));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, av, c, 0, 0, 0);
}

#define SC_PP_BARRIER_VM(N) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory")
#define SC_PP_BARRIER() asm volatile("s_barrier" ::: "memory")

// PREF: half-tiles 0 .. 7 of this tile were requested before the call (K >= 192); par: ring parity of K-tile 0 (0 unless the caller chains
// tiles with an odd number of K-tiles: the next tile then starts at par ^ 1)
template <int D, int DBG = 0, class TailHook = NoTailHook, bool I8 = false, bool PREF = false>
static __device__ __forceinline__ void gemm_tile256_mainloop_pp(const bf16_t* __restrict__ A, int lda, int m0, const bf16_t* __restrict__ W, int ldw,
                                                                 int n0, int K, char* smem, f32x4 (&acc)[4][8], int w, int lane,
                                                                 TailHook tail = TailHook{}, size_t a_kstep = G_BK, int par = 0) {
    static_assert(D >= 2 && D <= 5, "half-tiles in flight");
    constexpr bool HOOK = !std::is_same<TailHook, NoTailHook>::value;
    constexpr bool COOP = TailHook::kCoop;
    const int wm = w >> 2, wn = w & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int nk = K / G_BK;
    const int coop0 = ((nk + par) & 1) * 4 + 7;  // slot read in phase Q - 2 of the last two K-tiles: (coop0 + Q) & 7
    // sources: buffer descriptors (base + 32-bit per-lane offset + scalar K offset: no 64-bit vector arithmetic per piece)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * lda), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * ldw), 0, -1, 0x00020000);
    const uint32_t a_kbytes = (uint32_t)(a_kstep * 2), a1_off = (uint32_t)(64 * lda * 2), w1_off = (uint32_t)(32 * ldw * 2);
    uint32_t va[2], vw[2];  // byte offsets of this lane's 16 bytes in the wave's two pieces (8 rows x 128 B each) of an A / W half-tile
    pp_piece_offsets(lda, ldw, w, lane, va, vw);
    // fragment rows of this lane inside an A / W half-tile image (k-step 0; k-step 1 = the same address ^ 64)
    const uint32_t rd_sw = (uint32_t)(fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4));
    const uint32_t a_rd = (uint32_t)(wm * 64 * 128) + rd_sw, w_rd = (uint32_t)(wn * 32 * 128) + rd_sw;

    bf16x8 fa0[4][2], fa1[4][2], fw0[2][2], fw1[2][2];
    auto stage = [&](int t, int j) {  // request this wave's two pieces of half-tile j of K-tile t
        if (DBG & 1) return;
        char* dst = smem + ((((t + par) & 1) * 4 + j) * 16384 + 2 * w * 1024);
        if (j == 1 || j == 2) {
            const uint32_t so = (uint32_t)t * a_kbytes + (j == 2 ? a1_off : 0u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vptr)dst, 16, va[0], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vptr)(dst + 1024), 16, va[1], so, 0, 0);
        } else {
            const uint32_t so = (uint32_t)t * (uint32_t)(G_BK * 2) + (j == 3 ? w1_off : 0u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)dst, 16, vw[0], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)(dst + 1024), 16, vw[1], so, 0, 0);
        }
    };
    auto rdA = [&](bf16x8 (&dst)[4][2], int t, int j) {
        if (DBG & 4) return;
        const char* b = smem + (uint32_t)(((t + par) & 1) << 16);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) dst[mi][ks] = *reinterpret_cast<const bf16x8*>(b + ((a_rd ^ (uint32_t)(ks << 6)) + j * 16384 + mi * 2048));
    };
    auto rdW = [&](bf16x8 (&dst)[2][2], int t, int j) {
        if (DBG & 4) return;
        const char* b = smem + (uint32_t)(((t + par) & 1) << 16);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) dst[ni][ks] = *reinterpret_cast<const bf16x8*>(b + ((w_rd ^ (uint32_t)(ks << 6)) + j * 16384 + ni * 2048));
    };
    auto quad = [&](const bf16x8 (&fa)[4][2], const bf16x8 (&fw)[2][2], int mh, int nh) {
        if (DBG & 2) {
            asm volatile("" ::"v"(fa[0][0]), "v"(fa[3][1]), "v"(fw[0][0]), "v"(fw[1][1]));
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[nh * 2 + ni][mh * 4 + mi] = pp_mma<I8>(fw[ni][ks], fa[mi][ks], acc[nh * 2 + ni][mh * 4 + mi]);
        __builtin_amdgcn_s_setprio(0);
    };

    // prologue: half-tiles 0 .. D + 1 requested, 0 and 1 (W0, A0 of K-tile 0) landed and visible, W0(0) in registers
    if (!PREF) {
#pragma unroll
        for (int h = 0; h <= D + 1; ++h) {
            const bool keep = DBG & 1;  // (the ablation drops the in-loop requests only)
            if (!keep) stage(h >> 2, h & 3);
            else {
                char* dst = smem + (((h >> 2) & 1) * 4 + (h & 3)) * 16384 + 2 * w * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)dst, 16, vw[0], 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vptr)(dst + 1024), 16, vw[1], 0, 0, 0);
            }
        }
        SC_PP_BARRIER_VM(2 * D);
    } else {
        SC_PP_BARRIER_VM(12);  // half-tiles 2 .. 7 may fly
    }
    rdW(fw0, 0, 0);
    if (wm) SC_PP_BARRIER();  // wave group 1 runs one barrier behind group 0 from here on
    __builtin_amdgcn_sched_barrier(0);

    // MODE 0: K-tiles 0 .. nk - 3 (every request exists); 1: K-tile nk - 2; 2: K-tile nk - 1; 3: K-tile 0 behind a prefetch (PREF)
    auto ktile = [&](int t, auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;
        auto phase = [&](auto p_c) {
            constexpr int p = decltype(p_c)::value;
            constexpr int hs = p + D + 2;  // the half-tile requested in this phase, relative to 4 t
            constexpr bool ST = MODE == 0 || (MODE == 1 && hs < 8) || (MODE == 3 && hs >= 8);
            // Q: phase number within the last two K-tiles.  Pairs of pieces that may stay in flight behind this phase's wait (the
            // half-tile read in phase Q + 1 was requested in phase Q - D and must have landed; everything younger may fly): the
            // K-tiles' own requests of phases Q - D + 1 .. Q (they end with phase 5 - D) and, with a cooperative hook, its
            // requests of phases max(1, Q - D) .. min(Q, 7).
            constexpr int Q = (MODE == 0 || MODE == 3) ? -8 : MODE == 1 ? p : 4 + p;
            constexpr int NST = MODE == 0 ? D : MODE == 3 ? (5 - p > D ? 5 - p : D) : ((Q < 5 - D ? Q : 5 - D) - Q + D > 0 ? (Q < 5 - D ? Q : 5 - D) - Q + D : 0);
            constexpr int RLO = Q - D > 1 ? Q - D : 1, RHI = Q < 7 ? Q : 7;
            constexpr int NRES = (COOP && (MODE == 1 || MODE == 2) && RHI >= RLO) ? RHI - RLO + 1 : 0;
            constexpr int FL = NST + NRES;
            // ---- LOAD segment
            if (p == 0) rdA(fa0, t, 1);
            if (p == 1) rdA(fa1, t, 2);
            if (p == 2) rdW(fw1, t, 3);
            if (p == 3 && MODE != 2) rdW(fw0, t + 1, 0);
            if (ST) stage(t + (hs >> 2), hs & 3);
            if (COOP && Q >= 1) tail.coop((coop0 + Q) & 7, Q - 1, va, vw);
            if (MODE == 2 && p == 3) tail();
            if (MODE == 2 && p == 2 && HOOK) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the hook rewrites slots other waves read
            if (MODE == 2 && p >= 2) SC_PP_BARRIER();  // (nothing of the K-tiles is in flight any more)
            else SC_PP_BARRIER_VM(2 * FL);
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA segment
            if (p == 0) quad(fa0, fw0, 0, 0);
            if (p == 1) quad(fa1, fw0, 1, 0);
            if (p == 2) quad(fa1, fw1, 1, 1);
            if (p == 3) quad(fa0, fw1, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 2 && p == 3) {
                if (!wm) SC_PP_BARRIER();  // group 1 is one barrier behind: this is group 0's last
            } else {
                SC_PP_BARRIER();
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        phase(std::integral_constant<int, 0>{});
        phase(std::integral_constant<int, 1>{});
        phase(std::integral_constant<int, 2>{});
        phase(std::integral_constant<int, 3>{});
    };
    int t = 0;
    if (PREF) {
        ktile(0, std::integral_constant<int, 3>{});
        t = 1;
    }
#pragma unroll 1
    for (; t + 2 < nk; ++t) ktile(t, std::integral_constant<int, 0>{});
    tail.prepare();
    ktile(t, std::integral_constant<int, 1>{});
    ktile(t + 1, std::integral_constant<int, 2>{});
    if (COOP) tail.coop((coop0 + 8) & 7, 7, va, vw);  // the slot of the last half-tile read
}

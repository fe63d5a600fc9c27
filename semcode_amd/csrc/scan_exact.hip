// scan_exact.hip -- exact f32 distance scan fused with per-wave top-k (gfx950 / CDNA4).
//
// Replaces (reference): the server side of Collection.search(...) for a FLAT scan,
// src/semcode/storage/milvus_store.py:141-147 (one query per call there; up to 16 per pass here).
//
// Roofline: HBM.  Algorithmic bytes = n * dim * 4 per pass of <= qt queries; the arithmetic
// (v_mfma_f32_16x16x4_f32, 32 cycles per 64 corpus floats per SIMD) sustains ~19 TB/s chip-wide, so
// for <= 16 queries the corpus stream is the only bound.
//
// Structure (one workgroup = 4 independent waves, 1 workgroup per CU):
//   * each wave owns 16-row tiles of a contiguous row range; a tile is streamed as ld/64 stages of
//     16 rows x 64 floats (4 KiB) by LDS-DMA (global_load_lds_dwordx4, four 1 KiB pieces per stage)
//     into a private 4-deep LDS ring -- no barrier in the loop, ordering by counted s_waitcnt vmcnt;
//   * the LDS image is XOR-swizzled through the per-lane SOURCE address (LDS-DMA writes linearly):
//     16-byte chunk ch of row r lives in slot ch ^ r, which makes the ds_read_b128 A-fragment reads
//     conflict free;
//   * the <= 16 queries sit in LDS for the whole kernel (row pad 32 B -> conflict-free B reads);
//   * MFMA: A = corpus (row = lane & 15, k-group = lane >> 4), B = queries (col = lane & 15), one
//     fmaf chain per (row, query) in the canonical k order of oracle/sc_oracle.c sc_oracle_dot;
//   * per tile each lane holds 4 scores of one query: threshold filter against the wave's running
//     k-th best key, rare LDS append, rank-sort compaction when a candidate buffer fills;
//   * at the end the workgroup merges its 4 waves' lists; topk_merge.hip reduces the per-workgroup lists.
//
// Streamed-query variant (QS != 0), for rows so long that 16 resident queries no longer fit beside the ring (ld > ~1400:
// 3 072-d vectors left room for 6, so a list wanted by 16 queries was streamed three times): the queries are not resident.
// The four waves walk the k-chunks of their tiles in lock step and share a ring of three query stages, each TWO k-chunks of the
// 16 queries (2 x 4 KiB, the same XOR swizzle); wave w LDS-DMAs quarter w of every query stage from L2 (the query block is
// 16 x ld x 4 B and stays cache resident), and one raw s_barrier per query stage -- per two row stages -- publishes it (after the
// issuing wave's counted vmcnt) and frees the slot the next request overwrites.  Same MFMA chain per (row, query), so results are
// bit-identical to the resident variant.  (One k-chunk per query stage and a barrier per row stage streamed 5 % slower.)
#include "sc_common.h"
#include <stdlib.h>

#define SCAN_WAVES 4
#define SCAN_NSTAGE 4
#define SCAN_NORM_SLOTS 4
#define SCAN_STAGE_BYTES 4096
#define SCAN_NORM_BYTES 256
#define SCAN_QPAD 8  // floats

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gbl_vptr;
// volatile accesses must carry the LDS address space explicitly: address-space inference skips
// volatile operations, and a FLAT access counts on vmcnt, which would drain the LDS-DMA ring.
typedef __attribute__((address_space(3))) volatile uint64_t* lds_u64p;
typedef __attribute__((address_space(3))) volatile unsigned* lds_u32p;

struct ScanArgs {
    const float* X;
    const float* xnorm;
    int64_t n;
    int ld;
    const float* Qp;
    const float* qnorm;
    int Q;
    int qt;
    int k;
    int cap;
    int tiles_per_wg;
    uint64_t* partial;
    // optional: stored position -> reported row id (list-major IVF storage); NULL = identity
    const uint32_t* perm;
    // optional segment mode (IVF probe, blockIdx.y = group g): the group's queries scan nprobe row ranges
    // seg_rows[g][j] = {first, end}, whose 16-row tiles are numbered consecutively; seg_base[g][j] = first
    // tile ordinal of range j, seg_base[g][nprobe] = total tiles.  Per-query probing: qt == 1, group = query, ranges =
    // its probed lists.  List-major probing: group = up to qt queries that all probe one list, one range = that list.
    const int* seg_base;
    const int64_t* seg_rows;
    int nprobe;
    // optional: query slot -> row of Qp / qnorm (list-major probing gathers each group's queries); a group's valid slots
    // are a prefix, the first negative entry ends it.  NULL = identity.
    const int32_t* qmap;
    // slots per group in qmap / partial (>= qt; 0 = qt): list-major probing runs groups of few queries on the resident variant
    // and the others on the streamed one, with one numbering of (group, slot)
    int gstride;
};

struct ScanLds {
    unsigned ring, norms, qs, qn, thr, cnt, cand, tmp, seg, total;
};
// qstream: 0 = resident queries, else the number of ring stages of the streamed-query variant (SCAN_NSTAGE)
__host__ __device__ static inline ScanLds scan_lds_layout(int ld, int qt, int cap, int nprobe = 0, int qstream = 0) {
    const unsigned nst = qstream ? (unsigned)qstream : (unsigned)SCAN_NSTAGE;
    ScanLds L;
    unsigned o = 0;
    L.ring = o; o += SCAN_WAVES * nst * SCAN_STAGE_BYTES;
    L.norms = o; o += SCAN_WAVES * SCAN_NORM_SLOTS * SCAN_NORM_BYTES;
    L.qs = o; o += qstream ? 3u * 2u * SCAN_STAGE_BYTES : (unsigned)qt * (unsigned)(ld + SCAN_QPAD) * 4u;  // 3 query stages of 2 k-chunks
    L.qn = o; o += 64;
    L.thr = o; o += SCAN_WAVES * 16 * 8;
    L.cnt = o; o += SCAN_WAVES * 16 * 4;
    L.cand = o; o += (unsigned)SCAN_WAVES * (unsigned)qt * (unsigned)cap * 8u;
    L.tmp = o; o += (unsigned)SCAN_WAVES * (unsigned)cap * 8u;
    L.seg = o; o += nprobe ? (((unsigned)(nprobe + 1) * 4u + 15u) & ~15u) + (unsigned)nprobe * 16u : 0u;
    L.total = o;
    return L;
}

// Rank-sort the n (<= cap) candidate keys of one query slot, keep the k smallest in order, and
// refresh the pruning threshold.  Executed by one whole wave; all traffic through volatile LDS.
static __device__ __forceinline__ void wave_compact(lds_u64p cand, lds_u64p tmp, lds_u32p cnt, lds_u64p thr, int k, int lane) {
    const int n = (int)*cnt;
    if (n <= 64) {
        // one key per lane, ranked against the others by lane broadcasts (v_readlane): no LDS traffic in the n-step loop.  The LDS
        // form below costs a dependent ds_read_b64 per step; with the wide IVF groups restarting their thresholds per list part
        // it was a quarter of scan_listgemm_kernel's time.
        const uint64_t key = lane < n ? cand[lane] : SC_KEY_MAX;
        const unsigned klo = (unsigned)key, khi = (unsigned)(key >> 32);
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const uint64_t kj = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)khi, j) << 32) | (unsigned)__builtin_amdgcn_readlane((int)klo, j);
            rank += kj < key ? 1 : 0;
        }
        if (lane < n && rank < k) cand[rank] = key;  // keys are unique (row id in the low word): ranks are a permutation
        if (n >= k && lane < n && rank == k - 1) *thr = key;
        if (lane == 0) *cnt = (unsigned)(n < k ? n : k);
        return;
    }
    for (int e = lane; e < n; e += 64) {
        const uint64_t key = cand[e];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (cand[j] < key) ? 1 : 0;
        if (rank < k) tmp[rank] = key;
    }
    const int m = n < k ? n : k;
    for (int e = lane; e < m; e += 64) cand[e] = tmp[e];
    if (lane == 0) {
        *cnt = (unsigned)m;
        if (n >= k) *thr = tmp[k - 1];
    }
}

template <int METRIC, int QS>
__global__ __launch_bounds__(256) void scan_exact_kernel(ScanArgs a) {
    constexpr int NST = QS ? QS : SCAN_NSTAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15;  // A row / B column (query slot)
    const int g = lane >> 4;    // k-group
    const ScanLds L = scan_lds_layout(a.ld, a.qt, a.cap, a.seg_base ? a.nprobe : 0, QS);
    const int ld = a.ld;
    const int spt = ld >> 6;  // stages per tile
    const int grp = blockIdx.y;
    const int gstride = a.gstride ? a.gstride : a.qt;
    const int q0 = grp * gstride;
    int nq = min(a.qt, a.Q - q0);
    if (a.qmap) {
        int c = 0;
        while (c < nq && a.qmap[q0 + c] >= 0) ++c;
        nq = c;
    }

    // ---- queries -> LDS (once), thresholds / counters
    {
        const int qstride = ld + SCAN_QPAD;
        float* qs = reinterpret_cast<float*>(smem + L.qs);
        for (int c = 0; c < (QS ? 0 : a.qt); ++c) {
            const bool have = c < nq;
            const int qrow = a.qmap ? (have ? a.qmap[q0 + c] : 0) : q0 + (have ? c : 0);
            const float* src = a.Qp + (int64_t)qrow * ld;
            for (int kk = tid * 4; kk < ld; kk += 1024) {
                f32x4 v = have ? *reinterpret_cast<const f32x4*>(src + kk) : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(qs + c * qstride + kk) = v;
            }
        }
        if (tid < 16) reinterpret_cast<float*>(smem + L.qn)[tid] = (tid < nq) ? a.qnorm[a.qmap ? a.qmap[q0 + tid] : q0 + tid] : 1.0f;
        if (tid < SCAN_WAVES * 16) {
            reinterpret_cast<uint64_t*>(smem + L.thr)[tid] = SC_KEY_MAX;
            reinterpret_cast<unsigned*>(smem + L.cnt)[tid] = 0u;
        }
        if (a.seg_base) {
            int* sb = reinterpret_cast<int*>(smem + L.seg);
            int64_t* sr = reinterpret_cast<int64_t*>(smem + L.seg + (((unsigned)(a.nprobe + 1) * 4u + 15u) & ~15u));
            for (int j = tid; j <= a.nprobe; j += 256) sb[j] = a.seg_base[(size_t)grp * (a.nprobe + 1) + j];
            for (int j = tid; j < 2 * a.nprobe; j += 256) sr[j] = a.seg_rows[(size_t)grp * 2 * a.nprobe + j];
        }
    }
    __syncthreads();
    const int* seg_b = reinterpret_cast<const int*>(smem + L.seg);
    const int64_t* seg_r = reinterpret_cast<const int64_t*>(smem + L.seg + (((unsigned)(a.nprobe + 1) * 4u + 15u) & ~15u));
    const bool segmode = a.seg_base != nullptr;

    // ---- this wave's tiles: wg range [t0, t1), wave takes t0 + w, t0 + w + 4, ...
    const int64_t total_tiles = segmode ? (int64_t)seg_b[a.nprobe] : (a.n + 15) >> 4;
    const int tiles_per_wg = segmode ? (int)((total_tiles + gridDim.x - 1) / gridDim.x) : a.tiles_per_wg;
    const int64_t t0 = (int64_t)blockIdx.x * tiles_per_wg;
    int64_t t1 = t0 + tiles_per_wg;
    if (t1 > total_tiles) t1 = total_tiles;
    int ntiles = 0;
    if (t0 + w < t1) ntiles = (int)((t1 - (t0 + w) + SCAN_WAVES - 1) / SCAN_WAVES);
    // QS: every wave runs the stage count of wave 0 (the barrier and the shared query ring need all four); a wave past its
    // last tile streams clamped rows and drops the scores
    const int max_tiles = t0 < t1 ? (int)((t1 - t0 + SCAN_WAVES - 1) / SCAN_WAVES) : 0;
    const int total_stages = (QS ? max_tiles : ntiles) * spt;

    char* ring = smem + L.ring + w * (NST * SCAN_STAGE_BYTES);
    char* nrm = smem + L.norms + w * (SCAN_NORM_SLOTS * SCAN_NORM_BYTES);
    const char* qsb = smem + L.qs + (size_t)(r16 < a.qt ? r16 : a.qt - 1) * (size_t)(ld + SCAN_QPAD) * 4u + (size_t)g * 16u;
    lds_u64p thr_w = (lds_u64p)(smem + L.thr) + w * 16;
    lds_u32p cnt_w = (lds_u32p)(smem + L.cnt) + w * 16;
    lds_u64p cand_w = (lds_u64p)(smem + L.cand) + w * a.qt * a.cap;
    lds_u64p tmp_w = (lds_u64p)(smem + L.tmp) + w * a.cap;
    const float qn_mine = reinterpret_cast<const float*>(smem + L.qn)[r16];
    const int64_t last_row = a.n - 1;

    // tile ordinal -> first row and last valid row; `j` is a monotone cursor over the probe ranges
    auto tile_span = [&](int64_t ord, int& j, int64_t& row0, int64_t& last) {
        if (!segmode) {
            row0 = ord << 4;
            last = last_row;
        } else {
            while (j + 1 < a.nprobe && ord >= seg_b[j + 1]) ++j;
            row0 = seg_r[2 * j] + ((ord - seg_b[j]) << 4);
            last = seg_r[2 * j + 1] - 1;
        }
    };
    int iss_j = 0, con_j = 0;
    // issue side: (tile ordinal, k-chunk) of the next stage to request
    int iss = 0, iss_tile = 0, iss_kc = 0, iss_slot = 0;
    // per-lane source geometry of one LDS-DMA piece: row-in-piece = lane>>4, slot = lane&15
    const int prow = lane >> 4, pslot = lane & 15;

    auto issue_stage = [&]() {
        int64_t row0, tlast;
        tile_span(t0 + w + (int64_t)iss_tile * SCAN_WAVES, iss_j, row0, tlast);
        const int slot = iss_slot;
        if (++iss_slot == NST) iss_slot = 0;
        if (iss_kc == 0) {  // tile norms first: older than the tile's data in the vmcnt queue
            int64_t rr = row0 + r16;
            rr = rr > tlast ? tlast : rr;
            // one dword per lane: lanes 0-15 fetch the 16 row norms, lanes 16-31 the 16 reported row ids when the storage is
            // permuted (list-major IVF) -- the candidate path then never touches global memory (a lookup there cost a full
            // round trip with the ring drained, several times per tile while the thresholds are still loose)
            const float* nsrc = (a.perm && (lane & 48) == 16) ? reinterpret_cast<const float*>(a.perm + rr) : a.xnorm + rr;
            __builtin_amdgcn_global_load_lds((gbl_vptr)nsrc, (lds_vptr)(nrm + (iss_tile & (SCAN_NORM_SLOTS - 1)) * SCAN_NORM_BYTES), 4, 0, 0);
        }
        char* dst = ring + slot * SCAN_STAGE_BYTES;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = 4 * p + prow;
            int64_t rr = row0 + r;
            rr = rr > tlast ? tlast : rr;
            const float* src = a.X + rr * (int64_t)ld + (iss_kc << 6) + ((pslot ^ r) << 2);
            __builtin_amdgcn_global_load_lds((gbl_vptr)src, (lds_vptr)(dst + p * 1024), 16, 0, 2);
        }
        ++iss;
        if (++iss_kc == spt) { iss_kc = 0; ++iss_tile; }
    };
    // QS: a query stage holds TWO k-chunks of the 16 queries (8 KiB: chunk image | chunk image), three stages in the ring; this
    // wave requests its quarter (queries 4w .. 4w+3) of both chunks.  qiss counts query stages: stage t covers row stages 2t, 2t+1.
    int qiss = 0, qiss_kc = 0, qiss_slot = 0;
    const float* qsrc = a.Qp;
    if (QS) {
        const int c = 4 * w + prow;
        const int cc = c < nq ? c : 0;
        const int qrow = nq > 0 ? (a.qmap ? a.qmap[q0 + cc] : q0 + cc) : 0;
        qsrc = a.Qp + (int64_t)qrow * ld + ((pslot ^ c) << 2);
    }
    char* qring = smem + L.qs;
    auto issue_q = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // always two requests (the count below relies on it); past the end the chunk index just wraps
            __builtin_amdgcn_global_load_lds((gbl_vptr)(qsrc + (qiss_kc << 6)),
                                             (lds_vptr)(qring + qiss_slot * (2 * SCAN_STAGE_BYTES) + h * SCAN_STAGE_BYTES + w * 1024), 16, 0, 0);
            if (++qiss_kc == spt) qiss_kc = 0;
        }
        if (++qiss_slot == 3) qiss_slot = 0;
        ++qiss;
    };

#pragma unroll 1
    for (int j = 0; j < NST - 1; ++j)
        if (iss < total_stages) {
            issue_stage();
            if (QS && j < 2 && 2 * qiss < total_stages) issue_q();  // queue: rows(0) q(0) rows(1) q(1) rows(2)
        }

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int con_tile = 0, con_kc = 0, con_slot = 0, con_qslot = 0;
#pragma unroll 1
    for (int si = 0; si < total_stages; ++si) {
        if (iss < total_stages) issue_stage();
        // younger stages in flight behind stage si: each is >= 4 LDS-DMA instructions
        const int pend = iss - si - 1;
        if (!QS) {
            if (pend >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (pend == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (pend == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // Steady state (six more stages exist, so every request counted here was made).  Even si = 2j needs rows(si) and
            // q(j); younger than both: rows(si+1) q(j+1) rows(si+2) rows(si+3) = 4+2+4+4.  Odd si = 2j+1 needs rows(si); younger:
            // q(j+1) rows(si+1) rows(si+2) q(j+2) rows(si+3) = 2+4+4+2+4.  Near the end the wait is simply complete.
            if (si + 6 < total_stages) {
                if (si & 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        // this wave's row fragments need only its own counted wait: in the streamed variant they are requested BEFORE the
        // barrier, so their LDS latency runs under it
        const char* st = ring + con_slot * SCAN_STAGE_BYTES + r16 * 256;
        f32x4 av[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) av[t] = *reinterpret_cast<const f32x4*>(st + (((4 * t + g) ^ r16) << 4));
        if (QS && !(si & 1)) {
            // One barrier per query stage = per TWO row stages.  Raw barrier (__syncthreads() would drain vmcnt to 0 and with it
            // the ring).  lgkmcnt(4): everything older than the four fragment reads above has returned -- in particular stage
            // si-1's reads of the query slot that the request below overwrites (LDS returns in order; an outstanding scalar load
            // only makes the wait stricter).  After it query stage si/2 is visible to every wave and stage si/2 - 1 is dead.
            asm volatile("s_waitcnt lgkmcnt(4)\n\ts_barrier" ::: "memory");
            if (2 * qiss < total_stages) issue_q();  // q(si/2 + 2) -> the slot of q(si/2 - 1)
        }
        const char* qb = QS ? qring + con_qslot * (2 * SCAN_STAGE_BYTES) + (si & 1) * SCAN_STAGE_BYTES + r16 * 256 : qsb + (size_t)con_kc * 256u;
        if (QS && (si & 1) && ++con_qslot == 3) con_qslot = 0;
        if (++con_slot == NST) con_slot = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(QS ? qb + (((4 * t + g) ^ r16) << 4) : qb + t * 64);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][c], bv[c], acc, 0, 0, 0);
        }

        if (++con_kc == spt && con_tile >= ntiles) {  // QS only: a wave past its last tile
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            con_kc = 0;
            ++con_tile;
        } else if (con_kc == spt) {
            // ---- tile done: lane holds query r16, rows row0 + 4g + {0..3}
            int64_t row0, tlast;
            tile_span(t0 + w + (int64_t)con_tile * SCAN_WAVES, con_j, row0, tlast);
            const f32x4 xn = *reinterpret_cast<const f32x4*>(nrm + (con_tile & (SCAN_NORM_SLOTS - 1)) * SCAN_NORM_BYTES + g * 16);
            const u32x4_t pid = *reinterpret_cast<const u32x4_t*>(nrm + (con_tile & (SCAN_NORM_SLOTS - 1)) * SCAN_NORM_BYTES + 64 + g * 16);
            const uint64_t thr = thr_w[r16];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t row = row0 + 4 * g + c;
                const float sc = sc_score<METRIC>(acc[c], xn[c], qn_mine);
                const uint64_t key = sc_make_key<METRIC>(sc, a.perm ? pid[c] : (uint32_t)row);
                const bool cand = r16 < nq && row <= tlast;
                if (cand && key < thr) {
                    // inline asm: a compiler-visible LDS write here would get an s_waitcnt vmcnt(0) in front
                    // of it (it may alias the in-flight LDS-DMA as far as hipcc knows) and drain the ring.
                    unsigned pos;
                    const unsigned cnt_addr = (unsigned)(uintptr_t)(cnt_w + r16), one = 1u;
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pos) : "v"(cnt_addr), "v"(one) : "memory");
                    const unsigned slot_addr = (unsigned)(uintptr_t)(cand_w + r16 * a.cap + pos);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(slot_addr), "v"(key) : "memory");
                }
            }
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            const bool full = cnt_w[r16] > (unsigned)(a.cap - 16);
            if (__any(full)) {
                for (int c = 0; c < nq; ++c)
                    if (cnt_w[c] > (unsigned)(a.cap - 16))
                        wave_compact(cand_w + c * a.cap, tmp_w, cnt_w + c, thr_w + c, a.k, lane);
            }
            con_kc = 0;
            ++con_tile;
        }
    }

    // ---- flush: every wave sorts its slots; the workgroup then merges its 4 lists per slot (rank among the union,
    // keys are unique) and writes ONE sorted list: partial[grp][wg][slot][k]
    for (int c = 0; c < nq; ++c) wave_compact(cand_w + c * a.cap, tmp_w, cnt_w + c, thr_w + c, a.k, lane);
    __syncthreads();
    uint64_t* out = a.partial + ((size_t)grp * gridDim.x + blockIdx.x) * (size_t)gstride * a.k;
    lds_u64p cand_all = (lds_u64p)(smem + L.cand);
    lds_u32p cnt_all = (lds_u32p)(smem + L.cnt);
    for (int c = 0; c < a.qt; ++c) {
        int m[SCAN_WAVES], total = 0;
#pragma unroll
        for (int v = 0; v < SCAN_WAVES; ++v) {
            m[v] = c < nq ? (int)cnt_all[v * 16 + c] : 0;
            total += m[v];
        }
        for (int e = tid; e < total; e += 256) {
            int v = 0, idx = e;
            while (idx >= m[v]) { idx -= m[v]; ++v; }
            const uint64_t key = cand_all[(v * a.qt + c) * a.cap + idx];
            int rank = 0;
#pragma unroll
            for (int u = 0; u < SCAN_WAVES; ++u)
                for (int j = 0; j < m[u]; ++j) rank += (cand_all[(u * a.qt + c) * a.cap + j] < key) ? 1 : 0;
            if (rank < a.k) out[(size_t)c * a.k + rank] = key;
        }
        const int have = total < a.k ? total : a.k;
        for (int e = have + tid; e < a.k; e += 256) out[(size_t)c * a.k + e] = SC_KEY_MAX;
    }
}

// ------------------------------------------------------------------ wide groups: one list part x 32 or 64 queries per workgroup
// List-major IVF probing streams a list once per GROUP of queries that probe it.  With 16 queries per group the scan above is
// bound by its LDS-DMA row stream (4.8-5.1 TB/s at 3 072 dimensions) while the matrix pipe idles (16 MFMAs = 512 cycles per 4 KiB
// row stage, 28 % busy), and a popular list is streamed again and again: config 5 read 271 GB for 37.8 GB of distinct lists
// (profiles/r2i_bench.json.log).  This kernel is the same arithmetic organised as a GEMM: a workgroup takes 64 rows x 16 QB
// queries (QB = 2 or 4 query blocks) per k-chunk of 64 floats -- rows AND queries staged through a three-deep LDS ring by LDS-DMA
// (16 + 4 QB KiB per stage, the XOR swizzle of the scan above), one raw barrier per k-chunk -- and wave w multiplies the 16 QB
// rows of its row group by the 16 queries of its query block (QB accumulators, 16 QB MFMAs per stage): QB times the arithmetic
// per streamed row byte, so a list wanted by 64 queries is streamed once.  Per (row, query) the MFMA chain is the canonical one
// (k-chunks in order, t then c inside), so scores are bit-identical to scan_exact_kernel.  Candidate lists are wave-private (16
// query slots x the rows of the wave's row group); the 4 / QB waves that share a query block merge their sorted lists at the end
// (nothing to merge at QB = 4).
// Measured (2M x 3072, nlist 1024, 1 024 queries x nprobe 64, scripts/ivf_wide_ab.py; gpurun_out/ivf_wide_ab*.log): list-major
// probing 40.1 -> 21.8 ms, of which this kernel takes 14.7 ms for 41.9 GB of rows x 64 query slots.  Compile-time ablations of it
// on one box: without the LDS-DMA requests 12.0 ms (MFMAs + candidate handling; 64 MFMAs x 32 cycles per stage put the floor at
// 10.2 ms at 2.0 GHz; PMC: matrix pipe busy 0.61, profiles/r2p_pmc_mfma_ivf.json), requests and barriers alone 7.7 ms.  None of
// these moved it: spreading the requests between the MFMAs branch-free (22.9 -> 22.7 ms end to end, kept), taking the queries out of the
// LDS-DMA stream (each lane loading its own B fragments into registers and parking them in wave-private LDS: 23.5 ms, dropped), a
// register-only rank sort for the candidate lists (wave_compact above: +-0, kept), two waves per SIMD (22.9 -> 21.8 ms, kept
// where the lists fit: k <= 32).  What is left is arithmetic on empty query slots (groups of 33 .. 63 queries) and the 16-query
// classes next to it.
#define LG_ROWS 64
#define LG_NST 3
// candidates per (wave, query slot): k <= cap - 16 (a row block appends at most 16 per slot); 64 with 4 waves, 48 with 8 (LDS)
#define LG_CAP(NW) ((NW) == 8 ? 48 : 64)
#define LG_NORM_SLOTS 4

struct ListGemmArgs {
    const float* X;
    const float* xnorm;
    int ld;
    const float* Qp;        // padded queries [Q][ld]
    const float* qnorm;     // [Q]
    int k;
    uint64_t* partial;      // [groups][16 QB][k] sorted keys
    const uint32_t* perm;   // stored position -> reported row id, or NULL
    const int64_t* seg_rows;  // [groups][2] = {first, end} stored positions of the group's list part
    const int32_t* qmap;      // [groups][16 QB] query rows, valid slots a prefix, -1 beyond
};
template <int QB, int NW>
struct LgLds {
    static constexpr unsigned stage = 16384u + 4096u * QB;
    static constexpr unsigned ring = 0;
    static constexpr unsigned norms = LG_NST * stage;
    static constexpr unsigned thr = norms + LG_NORM_SLOTS * 4 * 256;
    static constexpr unsigned cnt = thr + NW * 16 * 8;
    static constexpr unsigned cand = cnt + NW * 16 * 4;
    static constexpr unsigned tmp = cand + NW * 16 * LG_CAP(NW) * 8;
    static constexpr unsigned total = tmp + NW * LG_CAP(NW) * 8;
};

// NW waves per workgroup: 4 (one per SIMD) or 8 (two per SIMD: one wave's barrier wait and fragment reads run under the other's
// MFMAs; candidate lists of 48 keys, so k <= 32)
template <int METRIC, int QB, int NW>
__global__ __launch_bounds__(NW * 64) void scan_listgemm_kernel(ListGemmArgs a) {
    using L = LgLds<QB, NW>;
    constexpr int CAP = LG_CAP(NW);
    constexpr int QW = 16 * QB;       // query slots of the group
    constexpr int RG = NW / QB;       // row groups (waves per query block)
    constexpr int RBW = 4 / RG;       // 16-row blocks per wave
    constexpr int RP = 16 / NW;       // row pieces (1 KiB = 4 rows) a wave requests per stage
    constexpr int QP = 4 * QB / NW;   // query pieces a wave requests per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qb = w % QB, rg = w / QB;  // this wave: query block qb x row blocks [RBW rg, RBW (rg + 1)) of every tile
    const int r16 = lane & 15, g = lane >> 4;
    const int grp = blockIdx.x;
    const int ld = a.ld, spt = ld >> 6;
    const int64_t first = a.seg_rows[2 * (size_t)grp], end = a.seg_rows[2 * (size_t)grp + 1];
    const int ntile = (int)((end - first + LG_ROWS - 1) / LG_ROWS);
    const int total = ntile * spt;
    const int32_t* qm = a.qmap + (size_t)grp * QW;
    const int nq = __popcll(__ballot(lane < QW && qm[lane < QW ? lane : 0] >= 0));  // valid slots are a prefix
    // wave-private selection state: 16 query slots
    lds_u64p thr_w = (lds_u64p)(smem + L::thr) + w * 16;
    lds_u32p cnt_w = (lds_u32p)(smem + L::cnt) + w * 16;
    lds_u64p cand_w = (lds_u64p)(smem + L::cand) + (size_t)w * 16 * CAP;
    lds_u64p tmp_w = (lds_u64p)(smem + L::tmp) + w * CAP;
    if (lane < 16) {
        thr_w[lane] = SC_KEY_MAX;
        cnt_w[lane] = 0u;
    }
    const int myslot = 16 * qb + r16;
    const float qn_mine = myslot < nq ? a.qnorm[qm[myslot]] : 1.0f;

    // ---- LDS-DMA geometry.  The 16 row pieces of a stage (piece P = rows 4 (P % 4) + prow of block P / 4) are dealt out RP per
    // wave, P = w RP + i; the 4 QB query pieces QP per wave the same way.
    const int prow = lane >> 4, pslot = lane & 15;
    const float* qsrc[QP];
#pragma unroll
    for (int i = 0; i < QP; ++i) {
        const int p = w * QP + i, r = 4 * (p & 3) + prow, sl = 16 * (p >> 2) + r;
        const int qrow = nq > 0 ? qm[sl < nq ? sl : 0] : 0;
        qsrc[i] = a.Qp + (int64_t)qrow * ld + ((pslot ^ r) << 2);
    }
    char* ringb = smem + L::ring;
    char* nrm = smem + L::norms;
    // Row addresses are kept per piece and advanced by one tile (64 rows) when a tile's first chunk is requested -- no multiply and,
    // apart from the norms piece that opens a tile, no branch in the request code, so that hipcc can spread it between the MFMAs;
    // rows past the end of the part are clamped to its last row through the pointer (their scores are dropped by row < end).
    int iss_tile = 0, iss_kc = 0, iss_slot = 0;
    const char* rsrc[RP];
    const char* rlast[RP];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        const int P = w * RP + i, r = 4 * (P & 3) + prow;
        rlast[i] = reinterpret_cast<const char*>(a.X + (end - 1) * (int64_t)ld + ((pslot ^ r) << 2));
        const int64_t rr = first + 16 * (P >> 2) + r - LG_ROWS;  // one tile before the first: the first request advances it
        rsrc[i] = reinterpret_cast<const char*>(a.X + rr * (int64_t)ld + ((pslot ^ r) << 2));
    }
    const int64_t tile_bytes = (int64_t)LG_ROWS * ld * 4;
    // The request of one stage is cut into four quarters that the main loop places between its MFMA groups: quarter j carries row
    // piece j (while j < RP) and query piece j - (4 - QP) (the last QP quarters), the first one also the norms piece of a new tile
    // (waves 0 .. 3: block w).  Stages are requested unconditionally, two past the last real one included (clamped rows into ring
    // slots that are dead by then; the flush's barrier waits for them): no branch, and one wait count for every stage.
    auto issue_quarter = [&](int j) {
        char* dst = ringb + iss_slot * L::stage;
        if (j == 0 && iss_kc == 0 && w < 4) {  // block w's 16 row norms (lanes 0-15) and reported ids (lanes 16-31): older than the tile's data
            int64_t rr = first + (int64_t)iss_tile * LG_ROWS + 16 * w + r16;
            rr = rr > end - 1 ? end - 1 : rr;
            const float* nsrc = (a.perm && (lane & 48) == 16) ? reinterpret_cast<const float*>(a.perm + rr) : a.xnorm + rr;
            __builtin_amdgcn_global_load_lds((gbl_vptr)nsrc, (lds_vptr)(nrm + ((iss_tile & (LG_NORM_SLOTS - 1)) * 4 + w) * 256), 4, 0, 0);
        }
        if (j < RP) {
            const int i = j < RP ? j : 0, P = w * RP + i;
            const char* adv = rsrc[i] + (iss_kc == 0 ? tile_bytes : (int64_t)0);
            rsrc[i] = adv;
            const char* src = (adv > rlast[i] ? rlast[i] : adv) + ((int64_t)iss_kc << 8);
            __builtin_amdgcn_global_load_lds((gbl_vptr)src, (lds_vptr)(dst + P * 1024), 16, 0, 2);
        }
        if (j >= 4 - QP) {
            const int i = j >= 4 - QP ? j - (4 - QP) : 0, p = w * QP + i;
            __builtin_amdgcn_global_load_lds((gbl_vptr)(qsrc[i] + (iss_kc << 6)), (lds_vptr)(dst + 16384 + p * 1024), 16, 0, 0);
        }
        if (j == 3) {
            if (++iss_slot == LG_NST) iss_slot = 0;
            if (++iss_kc == spt) { iss_kc = 0; ++iss_tile; }
        }
    };
    for (int pre = 0; pre < 2; ++pre) {
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_quarter(j);
    }

    f32x4 acc[RBW];
#pragma unroll
    for (int i = 0; i < RBW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int con_tile = 0, con_kc = 0, con_slot = 0;
#pragma unroll 1
    for (int si = 0; si < total; ++si) {
        // stage si + 1 is the younger one in flight: RP + QP pieces, one more when it opens a tile (the norms piece; waves 0 .. 3)
        {
            constexpr int N = RP + QP;
            static_assert(N == 8 || N == 6 || N == 4, "wait counts below");
            if (con_kc + 1 == spt && w < 4) {
                if (N == 8) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                else if (N == 6) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            } else {
                if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
        }
        // raw barrier: everyone's pieces of stage si have landed, and everyone is done with stage si - 1, whose slot the requests
        // below overwrite (a wave's fragment reads have returned before the MFMAs that consumed them were issued)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const char* st = ringb + con_slot * L::stage;
        const char* rst = st + (rg * RBW) * 4096 + r16 * 256;
        const char* qst = st + 16384 + qb * 4096 + r16 * 256;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int off = ((4 * t + g) ^ r16) << 4;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(qst + off);
            f32x4 av[RBW];
#pragma unroll
            for (int i = 0; i < RBW; ++i) av[i] = *reinterpret_cast<const f32x4*>(rst + i * 4096 + off);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int i = 0; i < RBW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][c], bv[c], acc[i], 0, 0, 0);
                if (c == 1) issue_quarter(t);  // LDS-DMA pieces in the middle of this group's MFMAs
            }
        }
        if (++con_slot == LG_NST) con_slot = 0;
        if (++con_kc == spt) {
            // ---- tile done: this lane holds query slot 16 qb + r16, rows (rg RBW + i) * 16 + 4 g + {0..3} of the tile
            const int64_t row0 = first + (int64_t)con_tile * LG_ROWS;
#pragma unroll
            for (int i = 0; i < RBW; ++i) {
                const int rb = rg * RBW + i;
                const char* nb = nrm + ((con_tile & (LG_NORM_SLOTS - 1)) * 4 + rb) * 256;
                const f32x4 xn = *reinterpret_cast<const f32x4*>(nb + g * 16);
                const u32x4_t pid = *reinterpret_cast<const u32x4_t*>(nb + 64 + g * 16);
                const uint64_t thr = thr_w[r16];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int64_t row = row0 + rb * 16 + 4 * g + c;
                    const float sc = sc_score<METRIC>(acc[i][c], xn[c], qn_mine);
                    const uint64_t key = sc_make_key<METRIC>(sc, a.perm ? pid[c] : (uint32_t)row);
                    if (myslot < nq && row < end && key < thr) {
                        // inline asm, as in scan_exact_kernel: a compiler-visible LDS write would drain the LDS-DMA ring first
                        unsigned pos;
                        const unsigned cnt_addr = (unsigned)(uintptr_t)(cnt_w + r16), one = 1u;
                        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pos) : "v"(cnt_addr), "v"(one) : "memory");
                        const unsigned slot_addr = (unsigned)(uintptr_t)(cand_w + r16 * CAP + pos);
                        asm volatile("ds_write_b64 %0, %1" ::"v"(slot_addr), "v"(key) : "memory");
                    }
                }
                acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                const bool full = cnt_w[r16] > (unsigned)(CAP - 16);
                if (__any(full)) {
#pragma unroll 1
                    for (int c = 0; c < 16; ++c)
                        if (cnt_w[c] > (unsigned)(CAP - 16)) wave_compact(cand_w + c * CAP, tmp_w, cnt_w + c, thr_w + c, a.k, lane);
                }
            }
            con_kc = 0;
            ++con_tile;
        }
    }
    // ---- flush: every wave sorts its 16 slots; the RG waves of a query block then merge their lists per slot (rank among the
    // union, keys are unique) and write ONE sorted k-list: partial[grp][slot][k]
#pragma unroll 1
    for (int c = 0; c < 16; ++c) wave_compact(cand_w + c * CAP, tmp_w, cnt_w + c, thr_w + c, a.k, lane);
    __syncthreads();  // (also waits for the two stages requested past the end)
    uint64_t* out = a.partial + (size_t)grp * QW * (size_t)a.k;
    lds_u64p cand_all = (lds_u64p)(smem + L::cand);
    lds_u32p cnt_all = (lds_u32p)(smem + L::cnt);
    // slot 16 b + c is shared by waves b + QB v (v < RG): the 16 slots of block qb are dealt out over its RG waves
#pragma unroll 1
    for (int c = rg; c < 16; c += RG) {
        const int slot = 16 * qb + c;
        int m[RG], tot = 0;
#pragma unroll
        for (int v = 0; v < RG; ++v) {
            m[v] = slot < nq ? (int)cnt_all[(qb + QB * v) * 16 + c] : 0;
            tot += m[v];
        }
        for (int e = lane; e < tot; e += 64) {
            int v = 0, idx = e;
            while (idx >= m[v]) { idx -= m[v]; ++v; }
            const uint64_t key = cand_all[((size_t)(qb + QB * v) * 16 + c) * CAP + idx];
            int rank = 0;
#pragma unroll
            for (int u = 0; u < RG; ++u)
                for (int j = 0; j < m[u]; ++j) rank += (cand_all[((size_t)(qb + QB * u) * 16 + c) * CAP + j] < key) ? 1 : 0;
            if (rank < a.k) out[(size_t)slot * a.k + rank] = key;
        }
        const int have = tot < a.k ? tot : a.k;
        for (int e = have + lane; e < a.k; e += 64) out[(size_t)slot * a.k + e] = SC_KEY_MAX;
    }
}

bool sc_scan_listgemm_supported(int ld, int k) { return ld > 0 && (ld % SC_LD_ALIGN) == 0 && k >= 1 && k <= LG_CAP(4) - 16; }

template <int METRIC, int QB, int NW>
static void launch_scan_listgemm(const ListGemmArgs& a, int groups, hipStream_t s) {
    static ScDeviceOnce once;  // per instantiation and device
    sc_device_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(scan_listgemm_kernel<METRIC, QB, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    constexpr unsigned lds = LgLds<QB, NW>::total;
    hipLaunchKernelGGL((scan_listgemm_kernel<METRIC, QB, NW>), dim3((unsigned)groups), dim3(NW * 64), lds, s, a);
}
template <int METRIC>
static void launch_scan_listgemm_m(const ListGemmArgs& a, int width, int groups, hipStream_t s) {
    // 64-query groups: two waves per SIMD where the candidate lists fit (k <= 32); SC_IVF_WAVES=4 forces one per SIMD (A/B)
    static const char* env = getenv("SC_IVF_WAVES");
    const bool eight = a.k <= LG_CAP(8) - 16 && !(env && env[0] == '4');
    if (width == 64 && eight) launch_scan_listgemm<METRIC, 4, 8>(a, groups, s);
    else if (width == 64) launch_scan_listgemm<METRIC, 4, 4>(a, groups, s);
    else launch_scan_listgemm<METRIC, 2, 4>(a, groups, s);
}
// width: 32 or 64 query slots per group
void sc_launch_scan_listgemm(int metric, int width, const float* X, const float* xnorm, int ld, const float* Qp, const float* qnorm, int k, int groups,
                             uint64_t* partial, const uint32_t* perm, const int64_t* seg_rows, const int32_t* qmap, hipStream_t s) {
    if (groups <= 0) return;
    ListGemmArgs a;
    a.X = X; a.xnorm = xnorm; a.ld = ld; a.Qp = Qp; a.qnorm = qnorm; a.k = k; a.partial = partial; a.perm = perm; a.seg_rows = seg_rows; a.qmap = qmap;
    if (metric == SC_METRIC_L2) launch_scan_listgemm_m<SC_METRIC_L2>(a, width, groups, s);
    else if (metric == SC_METRIC_COSINE) launch_scan_listgemm_m<SC_METRIC_COSINE>(a, width, groups, s);
    else launch_scan_listgemm_m<SC_METRIC_IP>(a, width, groups, s);
}

bool sc_scan_exact_plan(int ld, int Q, int k, int cus, ScanPlan* p, int force_qt, int nprobe, int64_t n_rows) {
    if (ld <= 0 || (ld % SC_LD_ALIGN) != 0 || k < 1 || k > 1024 || Q < 1) return false;
    const int cap = ((k + 16 + 63) / 64) * 64;  // >= k + 16, multiple of 64
    const unsigned budget = 160 * 1024;
    const int want = Q < 16 ? Q : 16;
    int qt = want;
    if (force_qt > 0 && force_qt < qt) qt = force_qt;
    while (qt >= 1 && scan_lds_layout(ld, qt, cap, nprobe).total > budget) --qt;
    if (qt < 1) return false;
    // Streamed queries: when the resident layout cannot hold the queries of one pass (long rows), 16 slots with the query
    // block streamed from L2 save whole passes over the rows (2M x 3072, 16 queries: 12.3 -> 5.2 ms).  The lock step costs
    // ~20 % of the streaming rate (4.7 vs 5.9 TB/s; a 6-deep ring measured the same as 4), so it is used only when it saves a
    // pass.  SC_SCAN_QSTREAM=0 forces the resident variant, anything else the streamed one wherever it fits (A/B runs, tests).
    const bool fits = force_qt <= 0 && ld >= 128 && scan_lds_layout(ld, 16, cap, nprobe, SCAN_NSTAGE).total <= budget;
    int qs = (qt < want && ld >= 512 && fits) ? SCAN_NSTAGE : 0;
    if (const char* e = getenv("SC_SCAN_QSTREAM")) qs = (e[0] != '0' && fits) ? SCAN_NSTAGE : 0;
    if (qs) qt = 16;
    p->qstream = qs;
    p->gstride = 0;
    p->qt = qt;
    p->groups = (Q + qt - 1) / qt;
    p->nwg = cus > 0 ? cus : 256;
    // a small corpus (an IVF quantizer's centroids: a few hundred 16-row tiles) gets one tile per wave and no idle workgroups: every
    // workgroup adds a k-list to the merge, and 256 lists of nprobe = 64 keys no longer fit the LDS tree merge (197 us vs 13 us)
    if (n_rows > 0) {
        const int64_t wgs = ((n_rows + 15) / 16 + SCAN_WAVES - 1) / SCAN_WAVES;
        if (wgs < p->nwg) {
            p->nwg = (int)(wgs < 1 ? 1 : wgs);
            const int fit = 8192 / k;  // lists whose 2 x k keys fit the tree merge's 128 KiB
            if (p->nwg > fit) p->nwg = fit < 1 ? 1 : fit;
        }
    }
    p->cap = cap;
    p->lists = p->nwg;  // one merged list per workgroup
    p->lds = scan_lds_layout(ld, qt, cap, nprobe, qs).total;
    p->partial_bytes = (size_t)p->groups * p->lists * qt * (size_t)k * sizeof(uint64_t) + sc_topk_merge_scratch_bytes(p->lists, p->groups * qt, k);
    return true;
}

template <int METRIC, int QS>
static void launch_scan_exact(const ScanArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    static ScDeviceOnce once;  // per instantiation and device
    sc_device_once(once, [&] { hipFuncSetAttribute(reinterpret_cast<const void*>(scan_exact_kernel<METRIC, QS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL((scan_exact_kernel<METRIC, QS>), grid, dim3(256), lds, s, a);
}

void sc_launch_scan_exact(int metric, const float* X, const float* xnorm, int64_t n, int ld, const float* Qp, const float* qnorm,
                          int Q, int k, const ScanPlan& p, uint64_t* partial, const uint32_t* perm, const int* seg_base,
                          const int64_t* seg_rows, int nprobe, hipStream_t s, const int32_t* qmap) {
    ScanArgs a;
    a.X = X; a.xnorm = xnorm; a.n = n; a.ld = ld; a.Qp = Qp; a.qnorm = qnorm; a.Q = Q; a.qt = p.qt; a.k = k; a.cap = p.cap;
    const int64_t tiles = (n + 15) / 16;
    a.tiles_per_wg = (int)((tiles + p.nwg - 1) / p.nwg);
    a.partial = partial;
    a.perm = perm; a.seg_base = seg_base; a.seg_rows = seg_rows; a.nprobe = nprobe; a.qmap = qmap; a.gstride = p.gstride;
    dim3 grid((unsigned)p.nwg, (unsigned)p.groups);
    if (p.qstream) {
        if (metric == SC_METRIC_L2) launch_scan_exact<SC_METRIC_L2, 4>(a, grid, p.lds, s);
        else if (metric == SC_METRIC_COSINE) launch_scan_exact<SC_METRIC_COSINE, 4>(a, grid, p.lds, s);
        else launch_scan_exact<SC_METRIC_IP, 4>(a, grid, p.lds, s);
    } else {
        if (metric == SC_METRIC_L2) launch_scan_exact<SC_METRIC_L2, 0>(a, grid, p.lds, s);
        else if (metric == SC_METRIC_COSINE) launch_scan_exact<SC_METRIC_COSINE, 0>(a, grid, p.lds, s);
        else launch_scan_exact<SC_METRIC_IP, 0>(a, grid, p.lds, s);
    }
}

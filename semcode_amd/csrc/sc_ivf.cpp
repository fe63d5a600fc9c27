// sc_ivf.cpp -- IVF_FLAT index build and probe search (host orchestration).
//
// Mirrors (reference): Collection.create_index(IVF_FLAT, metric, nlist) + load()
// (src/semcode/storage/milvus_store.py:76-84) and the nprobe parameter of Collection.search
// (src/semcode/storage/milvus_store.py:141-147).  Milvus' own k-means (Knowhere/faiss: random sample,
// random init) is not reproducible offline; this build is deterministic instead and is restated by
// oracle/ivf_oracle.py:
//   sample   : ns = min(n, 256 * nlist) rows, row floor(i * n / ns)
//   init     : centroid c = sample row floor(c * ns / nlist)
//   iterate  : niter x { assign every sample row to its nearest centroid (exact scores, ties -> lower
//              centroid id); centroid = f32 mean of its members summed in sample order; empty cluster
//              keeps its centroid }
//   assign   : L2 for metric L2 and IP (Voronoi cells), cosine for COSINE
//   lists    : every row goes to its nearest centroid; storage is re-ordered list-major (stable by row id)
//   probe    : per query the nprobe best centroids under the INDEX metric (IP: largest inner product),
//              then an exact scan of those lists (scan_exact.hip segment mode)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "sc_internal.h"

static const int ASSIGN_CHUNK = 8192;

static sc_metric assign_metric(sc_metric m) { return m == SC_METRIC_COSINE ? SC_METRIC_COSINE : SC_METRIC_L2; }

// nearest centroid (k = 1) for rows given as a tight [n, dim] device matrix; out: host vector of centroid ids
static sc_status assign_rows(sc_index* ix, const float* q_dev_tight, int64_t n, std::vector<int32_t>& out) {
    sc_index* qz = ix->quant;
    hipStream_t s = ix->rt->stream;
    out.resize((size_t)n);
    sc_status st = sc_grow(ix, &ix->ivf_scratch, &ix->ivf_scratch_cap, (size_t)ASSIGN_CHUNK * 12);
    if (st) return st;
    float* dd = (float*)ix->ivf_scratch;
    int64_t* dr = (int64_t*)((char*)ix->ivf_scratch + (size_t)ASSIGN_CHUNK * 4);
    std::vector<int64_t> host((size_t)ASSIGN_CHUNK);
    for (int64_t r0 = 0; r0 < n; r0 += ASSIGN_CHUNK) {
        const int m = (int)std::min<int64_t>(ASSIGN_CHUNK, n - r0);
        std::lock_guard<std::mutex> g(qz->mu);
        const int saved_mode = qz->search_mode, saved_coarse = qz->coarse_mode;
        qz->search_mode = 2;  // thousands of queries against few centroids: the MFMA path, certified exact
        // ... starting at the bf16 stage: against a few thousand centroids the int8 stage saves nothing in the coarse pass and
        // re-ranks 512 candidates per row instead of 128 (a 10M x 3072 build: 6.4 s vs 16.4 s, profiles/r2i_kernel_stats.csv)
        if (qz->coarse_mode == 0) qz->coarse_mode = 16;
        st = sc_search_flat_locked(qz, q_dev_tight + r0 * ix->dim, m, 1, dd, dr);
        qz->search_mode = saved_mode;
        qz->coarse_mode = saved_coarse;
        if (st) return st;
        SC_HIP(hipMemcpyAsync(host.data(), dr, (size_t)m * 8, hipMemcpyDeviceToHost, s));
        SC_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < m; ++i) out[(size_t)(r0 + i)] = (int32_t)host[(size_t)i];
    }
    return SC_OK;
}

namespace {
struct Dev {  // device allocation released on scope exit unless handed over with take()
    void* p = nullptr;
    ~Dev() { hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
    template <class T> T* take() { T* q = (T*)p; p = nullptr; return q; }
};
}  // namespace

void sc_ivf_drop_lists_locked(sc_index* ix) {
    if (ix->perm) hipStreamSynchronize(ix->rt->stream);
    hipFree(ix->perm);
    hipFree(ix->list_off);
    ix->perm = nullptr;
    ix->perm_rows = 0;
    ix->list_off = nullptr;
    ix->inv_h.clear();
    ix->list_off_h.clear();
    ix->assign_h.clear();
    ix->dirty_rows.clear();
    ix->ivf_rows = 0;
    ix->trained = false;
    ix->shadow_rows = 0;
    ix->shadow8_rows = 0;
    ix->shadowc_rows = 0;  // the centred shadow of the coarse stage mirrors the lists
    ix->ivfc_off = false;
}

// Xo[pos] = X[g[pos]] for the n stored rows, into fresh corpus-sized buffers that replace X / xnorm on success.
// Needs a second copy of the corpus for the duration of the move: the rebuildable buffers (bf16 shadow) are freed first.
static sc_status ivf_move_rows_locked(sc_index* ix, const std::vector<uint32_t>& g, Dev& d_g) {
    hipStream_t s = ix->rt->stream;
    const int64_t n = ix->n;
    SC_HIP(hipStreamSynchronize(s));
    hipFree(ix->Xb);  // the layout changes: the shadows are rebuilt anyway
    hipFree(ix->Xq);
    ix->Xb = ix->Xq = nullptr;
    ix->xb_cap = ix->xq_cap = 0;
    ix->shadow_rows = ix->shadow8_rows = 0;
    Dev nx, nn;
    if (nx.alloc((size_t)ix->capacity * ix->ld * sizeof(float)) != hipSuccess || nn.alloc((size_t)ix->capacity * sizeof(float)) != hipSuccess ||
        d_g.alloc((size_t)n * 4) != hipSuccess)
        return sc_fail(SC_ERR_NOMEM, "ivf: hipMalloc of the re-ordered corpus (%lld rows x %d) failed", (long long)ix->capacity, ix->ld);
    SC_HIP(hipMemcpyAsync(d_g.p, g.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    sc_launch_permute_rows(ix->X, ix->xnorm, (const uint32_t*)d_g.p, n, ix->ld, (float*)nx.p, (float*)nn.p, s);
    SC_HIP(hipGetLastError());
    SC_HIP(hipStreamSynchronize(s));
    hipFree(ix->X);
    hipFree(ix->xnorm);
    ix->X = nx.take<float>();
    ix->xnorm = nn.take<float>();
    return SC_OK;
}

sc_status sc_ivf_untrain_locked(sc_index* ix) {
    if (!ix->perm) {
        ix->trained = false;
        return SC_OK;
    }
    if (ix->n > 0) {  // Xo[row] = X[position of row]
        std::vector<uint32_t> g((size_t)ix->n);
        for (int64_t r = 0; r < ix->n; ++r) g[(size_t)r] = (uint32_t)sc_ivf_pos(ix, r);
        Dev d_g;
        sc_status st = ivf_move_rows_locked(ix, g, d_g);
        if (st) return st;  // nothing was changed: the lists stay valid
    }
    sc_ivf_drop_lists_locked(ix);
    return SC_OK;
}

// Given a quantizer already installed in ix->quant and the list of every stored row (by row id), re-order the corpus list-major
// (stable by row id inside a list).  Works from whatever layout is current: insertion order (fresh build) or an older list-major
// layout with appended rows behind it (incremental refresh).  Nothing of the index is modified unless every step succeeded.
static sc_status ivf_install_lists_locked(sc_index* ix, int nlist, std::vector<int32_t>&& assign) {
    hipStream_t s = ix->rt->stream;
    const int64_t n = ix->n;
    std::vector<int64_t> off((size_t)nlist + 1, 0);
    for (int64_t i = 0; i < n; ++i) off[(size_t)assign[(size_t)i] + 1]++;
    for (int c = 0; c < nlist; ++c) off[(size_t)c + 1] += off[(size_t)c];
    std::vector<uint32_t> perm((size_t)n), inv((size_t)n), g((size_t)n);
    {
        std::vector<int64_t> cur(off.begin(), off.end() - 1);
        for (int64_t i = 0; i < n; ++i) {
            const int64_t pos = cur[(size_t)assign[(size_t)i]]++;
            perm[(size_t)pos] = (uint32_t)i;
            inv[(size_t)i] = (uint32_t)pos;
            g[(size_t)pos] = ix->perm ? (uint32_t)sc_ivf_pos(ix, i) : (uint32_t)i;  // where row i sits now
        }
    }
    Dev d_perm, d_off, d_g;
    if (d_perm.alloc((size_t)n * 4) != hipSuccess || d_off.alloc((size_t)(nlist + 1) * 8) != hipSuccess)
        return sc_fail(SC_ERR_NOMEM, "ivf: hipMalloc of the list tables failed");
    SC_HIP(hipMemcpyAsync(d_perm.p, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    SC_HIP(hipMemcpyAsync(d_off.p, off.data(), (size_t)(nlist + 1) * 8, hipMemcpyHostToDevice, s));
    sc_status st = ivf_move_rows_locked(ix, g, d_g);  // synchronises: the uploads above are complete as well
    if (st) return st;
    hipFree(ix->perm);
    hipFree(ix->list_off);
    ix->perm = d_perm.take<uint32_t>();
    ix->perm_rows = 0;
    ix->list_off = d_off.take<int64_t>();
    ix->inv_h.swap(inv);
    ix->list_off_h.swap(off);
    ix->assign_h = std::move(assign);
    ix->ivf_rows = n;
    ix->dirty_rows.clear();
    ix->nlist_trained = nlist;
    ix->shadow_rows = ix->shadow8_rows = 0;
    ix->shadowc_rows = 0;  // new lists (a re-train over the same rows included): the centred shadow is rebuilt on the next coarse probe
    ix->ivfc_off = false;
    ix->uncert_frac = -1.0;
    ix->trained = true;
    return SC_OK;
}

// Incremental upsert: rows appended or overwritten since the lists were built are assigned to the EXISTING centroids and the
// corpus is re-ordered once (one pass over the corpus, no k-means).  The result is exactly what sc_index_assign_lists would
// build from scratch for these centroids.  Called at the start of every search; caller holds ix->mu.
static int g_ivf_refresh_nomem = 0;  // sc_diag_set_option("ivf_refresh_nomem", 1): tests of the fallback below
void sc_ivf_set_refresh_nomem(int v) { g_ivf_refresh_nomem = v; }
static int g_ivf_refine_cap = 1 << 30;  // sc_diag_set_option("ivf_refine_cap", n): the coarse stage's refine step takes on at most n rows per query (tests of the exact re-probe)
void sc_ivf_set_refine_cap(int v) { g_ivf_refine_cap = v < 0 ? (1 << 30) : v; }

sc_status sc_ivf_cover_tail_locked(sc_index* ix) {
    if (!ix->perm) return SC_OK;
    const int64_t have = ix->perm_rows > 0 ? ix->perm_rows : ix->ivf_rows;
    if (have >= ix->n) return SC_OK;
    hipStream_t s = ix->rt->stream;
    Dev d_new;
    if (d_new.alloc((size_t)ix->n * 4) != hipSuccess) return sc_fail(SC_ERR_NOMEM, "ivf: hipMalloc of the extended row map failed");
    SC_HIP(hipMemcpyAsync(d_new.p, ix->perm, (size_t)have * 4, hipMemcpyDeviceToDevice, s));
    std::vector<uint32_t> tail((size_t)(ix->n - have));
    for (int64_t r = have; r < ix->n; ++r) tail[(size_t)(r - have)] = (uint32_t)r;
    SC_HIP(hipMemcpyAsync((uint32_t*)d_new.p + have, tail.data(), tail.size() * 4, hipMemcpyHostToDevice, s));
    SC_HIP(hipStreamSynchronize(s));
    hipFree(ix->perm);
    ix->perm = d_new.take<uint32_t>();
    ix->perm_rows = ix->n;
    return SC_OK;
}

sc_status sc_ivf_refresh_locked(sc_index* ix, bool keep_tail) {
    if (!ix->perm || !ix->quant || (ix->ivf_rows == ix->n && ix->dirty_rows.empty())) return SC_OK;
    if (keep_tail && ix->dirty_rows.empty()) return SC_OK;
    if (g_ivf_refresh_nomem) return sc_fail(SC_ERR_NOMEM, "ivf refresh: out of device memory (forced by sc_diag_set_option)");
    hipStream_t s = ix->rt->stream;
    std::vector<int64_t> rows(ix->dirty_rows);
    std::sort(rows.begin(), rows.end());
    rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
    if (!keep_tail)
        for (int64_t r = ix->ivf_rows; r < ix->n; ++r) rows.push_back(r);
    // (the new lists of the rows are collected first: the copy of the whole assignment -- 40 MB at 10M rows, most of what a refresh that
    // moves nothing used to cost -- is made only when something does move)
    std::vector<int32_t> new_list(rows.size());
    bool changed = !keep_tail && ix->n > ix->ivf_rows;
    const int64_t CH = 65536;
    Dev d_pos, d_tight;
    const int64_t chmax = std::min<int64_t>(CH, (int64_t)rows.size());
    if (d_pos.alloc((size_t)chmax * 8) != hipSuccess || d_tight.alloc((size_t)chmax * ix->dim * 4) != hipSuccess)
        return sc_fail(SC_ERR_NOMEM, "ivf refresh: hipMalloc failed");
    std::vector<int64_t> pos((size_t)chmax);
    std::vector<int32_t> out;
    for (int64_t c0 = 0; c0 < (int64_t)rows.size(); c0 += CH) {
        const int64_t m = std::min<int64_t>(CH, (int64_t)rows.size() - c0);
        for (int64_t i = 0; i < m; ++i) pos[(size_t)i] = sc_ivf_pos(ix, rows[(size_t)(c0 + i)]);
        SC_HIP(hipMemcpyAsync(d_pos.p, pos.data(), (size_t)m * 8, hipMemcpyHostToDevice, s));
        sc_launch_rows_to_sample(ix->X, ix->ld, ix->dim, (const int64_t*)d_pos.p, m, (float*)d_tight.p, s);
        SC_HIP(hipGetLastError());
        sc_status st = assign_rows(ix, (const float*)d_tight.p, m, out);  // synchronises
        if (st) return st;
        for (int64_t i = 0; i < m; ++i) {
            const int64_t r = rows[(size_t)(c0 + i)];
            if (r >= (int64_t)ix->assign_h.size() || ix->assign_h[(size_t)r] != out[(size_t)i]) changed = true;
            new_list[(size_t)(c0 + i)] = out[(size_t)i];
        }
    }
    if (!changed) {  // overwritten rows all stayed in their lists: nothing moves
        ix->dirty_rows.clear();
        return SC_OK;
    }
    if (keep_tail) return sc_ivf_refresh_locked(ix, false);  // a row left its list: the layout is rebuilt, the tail joins it
    std::vector<int32_t> assign(ix->assign_h);
    assign.resize((size_t)ix->n, -1);
    for (size_t i = 0; i < rows.size(); ++i) assign[(size_t)rows[i]] = new_list[i];
    return ivf_install_lists_locked(ix, ix->nlist_trained, std::move(assign));
}

// Quantizer installed in ix->quant: assign every stored row to its nearest centroid and re-order the corpus list-major.
static sc_status ivf_assign_all_and_install_locked(sc_index* ix, int nlist) {
    hipStream_t s = ix->rt->stream;
    const int64_t n = ix->n;
    Dev d_tight;
    const float* all_tight = ix->X;
    if (ix->ld != ix->dim) {
        SC_HIP(hipMalloc(&d_tight.p, (size_t)n * ix->dim * 4));
        sc_launch_gather_rows(ix->X, ix->ld, 0, n, ix->dim, (float*)d_tight.p, s);
        all_tight = (const float*)d_tight.p;
    }
    std::vector<int32_t> assign;
    sc_status st = assign_rows(ix, all_tight, n, assign);
    if (st) return st;
    return ivf_install_lists_locked(ix, nlist, std::move(assign));
}

extern "C" sc_status sc_index_train(sc_index* ix, int32_t niter, uint64_t seed) {
    (void)seed;  // the build is deterministic; kept for ABI stability
    if (!ix) return sc_fail(SC_ERR_INVALID, "index is NULL");
    if (niter < 0 || niter > 1000) return sc_fail(SC_ERR_INVALID, "sc_index_train: niter out of range");
    std::lock_guard<std::mutex> g(ix->mu);
    if (ix->kind != SC_INDEX_IVF_FLAT) return sc_fail(SC_ERR_STATE, "sc_index_train: index kind is not IVF_FLAT");
    if (ix->n < 1) return sc_fail(SC_ERR_STATE, "sc_index_train: the index is empty");
    SC_HIP(hipSetDevice(ix->rt->device));
    hipStream_t s = ix->rt->stream;
    sc_status st = sc_ivf_untrain_locked(ix);
    if (st) return st;
    const int64_t n = ix->n;
    const int dim = ix->dim, ld = ix->ld;
    const int nlist = (int)std::min<int64_t>(ix->nlist, n);
    const int64_t ns = std::min<int64_t>(n, 256ll * nlist);

    // ---- sample (tight [ns, dim]) and initial centroids
    std::vector<int64_t> srows((size_t)ns);
    for (int64_t i = 0; i < ns; ++i) srows[(size_t)i] = (int64_t)(((__int128)i * n) / ns);
    Dev d_srows, d_sample, d_cinit, d_members, d_moff, d_cnew;
    SC_HIP(hipMalloc(&d_srows.p, (size_t)ns * 8));
    SC_HIP(hipMalloc(&d_sample.p, (size_t)ns * dim * 4));
    SC_HIP(hipMemcpyAsync(d_srows.p, srows.data(), (size_t)ns * 8, hipMemcpyHostToDevice, s));
    sc_launch_rows_to_sample(ix->X, ld, dim, (const int64_t*)d_srows.p, ns, (float*)d_sample.p, s);
    std::vector<int64_t> crow((size_t)nlist);
    for (int c = 0; c < nlist; ++c) crow[(size_t)c] = (int64_t)(((__int128)c * ns) / nlist);
    SC_HIP(hipMalloc(&d_cinit.p, (size_t)nlist * 8));
    SC_HIP(hipMalloc(&d_cnew.p, (size_t)nlist * dim * 4));
    SC_HIP(hipMemcpyAsync(d_cinit.p, crow.data(), (size_t)nlist * 8, hipMemcpyHostToDevice, s));
    sc_launch_rows_to_sample((const float*)d_sample.p, dim, dim, (const int64_t*)d_cinit.p, nlist, (float*)d_cnew.p, s);
    SC_HIP(hipStreamSynchronize(s));

    // ---- quantizer = flat index over the centroids
    if (ix->quant) {
        sc_index_destroy(ix->quant);
        ix->quant = nullptr;
    }
    st = sc_index_create(ix->rt, dim, assign_metric(ix->metric), SC_INDEX_FLAT, 0, 0, &ix->quant);
    if (st) return st;
    sc_index* qz = ix->quant;
    {
        std::lock_guard<std::mutex> gq(qz->mu);
        qz->n = 0;
    }
    st = sc_index_reserve(qz, nlist);
    if (st) return st;
    auto set_centroids = [&](const float* c_tight) {
        std::lock_guard<std::mutex> gq(qz->mu);
        sc_launch_ingest_rows(c_tight, nullptr, 0, nlist, dim, qz->X, qz->ld, qz->xnorm, s);
        qz->n = nlist;
        qz->shadow_rows = qz->shadow8_rows = 0;  // the centroids changed: both coarse shadows are stale
    };
    set_centroids((const float*)d_cnew.p);

    // ---- Lloyd iterations on the sample
    SC_HIP(hipMalloc(&d_members.p, (size_t)ns * 8));
    SC_HIP(hipMalloc(&d_moff.p, (size_t)(nlist + 1) * 8));
    std::vector<int32_t> assign;
    std::vector<int64_t> members((size_t)ns), moff((size_t)nlist + 1);
    for (int it = 0; it < niter; ++it) {
        st = assign_rows(ix, (const float*)d_sample.p, ns, assign);
        if (st) return st;
        std::fill(moff.begin(), moff.end(), 0);
        for (int64_t i = 0; i < ns; ++i) moff[(size_t)assign[(size_t)i] + 1]++;
        for (int c = 0; c < nlist; ++c) moff[(size_t)c + 1] += moff[(size_t)c];
        std::vector<int64_t> cur(moff.begin(), moff.end() - 1);
        for (int64_t i = 0; i < ns; ++i) members[(size_t)cur[(size_t)assign[(size_t)i]]++] = i;  // sample order inside a cluster
        SC_HIP(hipMemcpyAsync(d_members.p, members.data(), (size_t)ns * 8, hipMemcpyHostToDevice, s));
        SC_HIP(hipMemcpyAsync(d_moff.p, moff.data(), (size_t)(nlist + 1) * 8, hipMemcpyHostToDevice, s));
        sc_launch_centroid_mean((const float*)d_sample.p, dim, dim, (const int64_t*)d_members.p, (const int64_t*)d_moff.p, nlist,
                                (float*)d_cnew.p, qz->X, qz->ld, s);
        SC_HIP(hipGetLastError());
        // Re-seeding (every iteration but the last, so that final centroids are plain means).  Strided initialisation leaves
        // some true clusters without a centroid and gives others two; Lloyd iterations cannot repair that, and in high
        // dimension the orphaned clusters all fall to one "hub" centroid near the global mean (measured on config 5: one list
        // of 124k rows at a median of 2.4k, list-major probing reading the corpus 26 times over).  So starved centroids
        // (count < 0.75 average, smallest first) are moved onto the largest ones (count > 2 average, largest first, sizes
        // halved on every split), as faiss does for empty clusters.  Integer rule + in-order moves: restated exactly by
        // oracle/ivf_oracle.py.
        if (it + 1 < niter) {
            std::vector<int64_t> cnt((size_t)nlist);
            for (int c = 0; c < nlist; ++c) cnt[(size_t)c] = moff[(size_t)c + 1] - moff[(size_t)c];
            std::vector<int> donors((size_t)nlist);
            for (int c = 0; c < nlist; ++c) donors[(size_t)c] = c;
            std::stable_sort(donors.begin(), donors.end(), [&](int a, int b) { return cnt[(size_t)a] < cnt[(size_t)b]; });
            auto less_big = [](const std::pair<int64_t, int>& a, const std::pair<int64_t, int>& b) {
                return a.first != b.first ? a.first < b.first : a.second > b.second;  // max-heap: larger size, then lower id
            };
            std::vector<std::pair<int64_t, int>> heap;
            for (int c = 0; c < nlist; ++c)
                if (cnt[(size_t)c] * nlist > 2 * ns) heap.emplace_back(cnt[(size_t)c], c);
            std::make_heap(heap.begin(), heap.end(), less_big);
            std::vector<int32_t> moves;
            for (int di = 0; di < nlist && !heap.empty(); ++di) {
                const int e = donors[(size_t)di];
                if (cnt[(size_t)e] * 4 * nlist >= 3 * ns) break;
                std::pop_heap(heap.begin(), heap.end(), less_big);
                const std::pair<int64_t, int> big = heap.back();
                heap.pop_back();
                if (big.first * nlist <= 2 * ns) break;
                moves.push_back(e);
                moves.push_back(big.second);
                const int64_t half = big.first / 2;
                heap.emplace_back(big.first - half, big.second);
                std::push_heap(heap.begin(), heap.end(), less_big);
                heap.emplace_back(half, e);
                std::push_heap(heap.begin(), heap.end(), less_big);
            }
            if (!moves.empty()) {
                Dev d_moves;
                SC_HIP(hipMalloc(&d_moves.p, moves.size() * 4));
                SC_HIP(hipMemcpyAsync(d_moves.p, moves.data(), moves.size() * 4, hipMemcpyHostToDevice, s));
                sc_launch_reseed_centroids((float*)d_cnew.p, dim, (const int32_t*)d_moves.p, (int)(moves.size() / 2), s);
                SC_HIP(hipGetLastError());
                SC_HIP(hipStreamSynchronize(s));
            }
        }
        SC_HIP(hipStreamSynchronize(s));
        set_centroids((const float*)d_cnew.p);
    }

    return ivf_assign_all_and_install_locked(ix, nlist);
}

bool sc_ivf_applicable(const sc_index* ix, int Q, int nprobe) {
    if (ix->kind != SC_INDEX_IVF_FLAT || !ix->trained || !ix->quant) return false;
    if (nprobe < 1 || nprobe > 512 || ix->search_mode == 1 || ix->search_mode == 2 || ix->search_mode == 4 || ix->search_mode == 5) return false;
    if (nprobe >= ix->nlist_trained) return false;  // probing every list = the exhaustive scan
    if (ix->search_mode == 3) return true;
    // one pass per query over nprobe/nlist of the corpus vs one exhaustive pass per 16 queries (or the
    // batched path): probe only while it reads less than a single full pass
    return (int64_t)Q * nprobe < (int64_t)ix->nlist_trained;
}

sc_status sc_ivf_search_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe, float* out_dist, int64_t* out_rows) {
    sc_runtime* rt = ix->rt;
    hipStream_t s = rt->stream;
    sc_index* qz = ix->quant;
    const int nlist = ix->nlist_trained;
    ScanPlan plan;
    if (!sc_scan_exact_plan(ix->ld, Q, k, rt->cus, &plan, 1, nprobe))
        return sc_fail(SC_ERR_UNSUPPORTED, "ivf search: k=%d / dim=%d / nprobe=%d not supported", k, ix->dim, nprobe);
    // scratch: probe results + plan tables
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_pd = carve((size_t)Q * nprobe * 4), o_pr = carve((size_t)Q * nprobe * 8), o_sb = carve((size_t)Q * (nprobe + 1) * 4),
                 o_sr = carve((size_t)Q * nprobe * 16);
    sc_status st = sc_grow(ix, &ix->ivf_scratch, &ix->ivf_scratch_cap, off);
    if (st) return st;
    char* b = (char*)ix->ivf_scratch;
    float* pd = (float*)(b + o_pd);
    int64_t* pr = (int64_t*)(b + o_pr);
    int* sb = (int*)(b + o_sb);
    int64_t* sr = (int64_t*)(b + o_sr);
    {   // coarse probe under the index metric
        std::lock_guard<std::mutex> gq(qz->mu);
        const sc_metric saved = qz->metric;
        qz->metric = ix->metric;
        st = sc_search_flat_locked(qz, q_dev, Q, nprobe, pd, pr);
        qz->metric = saved;
        if (st) return st;
    }
    sc_launch_ivf_plan(pr, Q, nprobe, ix->list_off, nlist, sb, sr, s);
    st = sc_grow(ix, (void**)&ix->qpad, &ix->qpad_cap, (size_t)Q * ix->ld * 4);
    if (st) return st;
    st = sc_grow(ix, (void**)&ix->qnorm, &ix->qnorm_cap, (size_t)Q * 4);
    if (st) return st;
    st = sc_grow(ix, (void**)&ix->partial, &ix->partial_cap, std::max<size_t>(plan.partial_bytes, 16));
    if (st) return st;
    sc_launch_ingest_rows(q_dev, nullptr, 0, Q, ix->dim, ix->qpad, ix->ld, ix->qnorm, s);
    hipEvent_t e0, e1;
    sc_prof_begin(rt, SC_PROF_SCAN, &e0, &e1);
    sc_launch_scan_exact((int)ix->metric, ix->X, ix->xnorm, ix->n, ix->ld, ix->qpad, ix->qnorm, Q, k, plan, ix->partial, ix->perm, sb, sr, nprobe, s);
    sc_prof_end(rt, SC_PROF_SCAN, e0, e1);
    sc_launch_topk_merge((int)ix->metric, ix->partial, plan.groups, plan.lists, plan.qt, Q, k, ix->row_base, out_dist, out_rows, s);
    SC_HIP(hipGetLastError());
    ix->last_path = 3;
    ix->last_probed_lists = nprobe;
    return SC_OK;
}

// ---- list-major probing for query batches ------------------------------------------------------------------------
// Per-query probing streams nprobe lists once PER QUERY; with Q * nprobe >= nlist probes every list is wanted by several
// queries, so the batch is turned round: the (query, list) pairs are bucketed by list, every list is streamed once per
// group of up to qt queries that probe it (scan_exact_kernel, one row range per group, the group's queries gathered
// through qmap), and a query's top-k is merged from the nprobe (group, slot) lists it took part in.  Same lists, same exact
// f32 scores and tie rule as per-query probing, so the results are identical to it.
static bool ivf_listmajor_plan(const sc_index* ix, int k, int nprobe, ScanPlan* plan) {
    return sc_scan_exact_plan(ix->ld, 16, k, ix->rt->cus, plan, 0, 1) && sc_topk_gather_merge_supported(nprobe, k);
}

bool sc_ivf_listmajor_applicable(const sc_index* ix, int Q, int k, int nprobe, bool flat_is_batched) {
    if (ix->kind != SC_INDEX_IVF_FLAT || !ix->trained || !ix->quant || Q < 2) return false;
    if (nprobe < 1 || nprobe > 512 || nprobe >= ix->nlist_trained || (ix->search_mode >= 1 && ix->search_mode <= 3)) return false;
    ScanPlan plan;
    if (!ivf_listmajor_plan(ix, k, nprobe, &plan)) return false;
    if (ix->search_mode == 4 || ix->search_mode == 5) return true;  // (5: the coarse stage was asked for and could not run)
    // auto: probe while that is estimated to be cheaper than the exhaustive paths (which return exact results).
    // Queries follow the data, so a list is expected to receive (query, list) pairs in proportion to its length:
    // work = sum over lists of len * groups(len).  The batched exhaustive path re-runs uncertified queries through the exact
    // scan, which on clustered data (what an IVF index is trained on) can be most of them: its share at the last such search
    // on this index is fed back (10 % assumed before the first).  Constants measured on MI355X (profiles/r1q_ivf_*.log):
    // list-major streams 5.5 TB/s after 1.5 ms of coarse probe + planning; batched exhaustive 1.0 PFLOP/s + 1 ms; exact
    // scan 6 TB/s per pass of qt queries.
    const double n = (double)ix->n, row_bytes = (double)ix->ld * 4.0, pairs = (double)Q * nprobe;
    // a list expected to be wanted by m queries: chunks of 64 on the GEMM-shaped kernel (f32 MFMA bound: 2 * 64 * ld FLOP per row at
    // ~100 TFLOP/s, profiles/r2m_kernel_stats.csv), a remainder of <= 16 on the 16-query scan (row stream at 5.5 TB/s)
    const bool wide_ok = sc_scan_listgemm_supported(ix->ld, k) && plan.qt == 16;
    const double t_row_narrow = row_bytes / 5.5e12, t_row_wide = 2.0 * 64.0 * (double)ix->ld / 1.0e14;
    double t_scan = 0.0;
    for (int l = 0; l < ix->nlist_trained; ++l) {
        const double len = (double)(ix->list_off_h[(size_t)l + 1] - ix->list_off_h[(size_t)l]);
        if (len <= 0) continue;
        double m = std::max(1.0, std::ceil(pairs * len / n));
        if (wide_ok && m > plan.qt) {
            const double chunks = std::floor(m / 64.0), rest = m - 64.0 * chunks;
            t_scan += len * t_row_wide * (chunks + (rest > plan.qt ? (rest > 32.0 ? 1.0 : 0.5) : 0.0));
            m = rest > plan.qt ? 0.0 : rest;
        }
        t_scan += len * t_row_narrow * std::ceil(m / plan.qt);
    }
    const double t_lm = t_scan + 1.5e-3;
    const double t_pass = n * row_bytes / 6.0e12;
    double t_flat;
    if (flat_is_batched) {
        const double redo = (ix->uncert_frac < 0 ? 0.1 : ix->uncert_frac) * Q;
        t_flat = std::max(2.0 * n * ix->ld * Q / 1.0e15, n * ix->ld * 2.0 / 5.0e12) + 1.0e-3 + std::ceil(redo / plan.qt) * t_pass;
    } else {
        t_flat = std::ceil((double)Q / plan.qt) * t_pass;
    }
    return t_lm < t_flat;
}

sc_status sc_ivf_search_listmajor_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe, float* out_dist,
                                         int64_t* out_rows) {
    sc_runtime* rt = ix->rt;
    hipStream_t s = rt->stream;
    sc_index* qz = ix->quant;
    const int nlist = ix->nlist_trained;
    ScanPlan plan;
    if (!ivf_listmajor_plan(ix, k, nprobe, &plan)) return sc_fail(SC_ERR_UNSUPPORTED, "ivf list-major search: k=%d / dim=%d / nprobe=%d not supported", k, ix->dim, nprobe);
    const int qt = plan.qt;
    const size_t npairs = (size_t)Q * nprobe;
    // 1. coarse probe under the index metric -> host
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_pd = carve(npairs * 4), o_pr = carve(npairs * 8);
    sc_status st = sc_grow(ix, &ix->ivf_scratch, &ix->ivf_scratch_cap, off);
    if (st) return st;
    char* b = (char*)ix->ivf_scratch;
    {
        std::lock_guard<std::mutex> gq(qz->mu);
        const sc_metric saved = qz->metric;
        qz->metric = ix->metric;
        st = sc_search_flat_locked(qz, q_dev, Q, nprobe, (float*)(b + o_pd), (int64_t*)(b + o_pr));
        qz->metric = saved;
        if (st) return st;
    }
    static const bool trace = getenv("SC_IVF_TRACE") != nullptr;  // tuning aid: host-side phase times on stderr
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    std::vector<int64_t> probes(npairs);
    SC_HIP(hipMemcpyAsync(probes.data(), b + o_pr, npairs * 8, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    const double t_probe = since();
    // 2. bucket the pairs by list (counting sort keeps the queries of a list in ascending order: deterministic groups)
    std::vector<int> start((size_t)nlist + 1, 0);
    for (size_t i = 0; i < npairs; ++i) {
        const int64_t l = probes[i];
        if (l >= 0 && l < nlist) ++start[(size_t)l + 1];
    }
    for (int l = 0; l < nlist; ++l) start[(size_t)l + 1] += start[(size_t)l];
    std::vector<int> fill(start.begin(), start.end() - 1);
    std::vector<int32_t> pair_of((size_t)start[(size_t)nlist]);
    for (size_t i = 0; i < npairs; ++i) {
        const int64_t l = probes[i];
        if (l >= 0 && l < nlist) pair_of[(size_t)fill[(size_t)l]++] = (int32_t)i;
    }
    // long lists are cut into parts of at most `target` rows (each part its own group) so that no single workgroup streams a
    // 20k-row list while the others idle; a query then merges up to nprobe * maxparts partial lists
    // Wide groups (scan_listgemm_kernel): a list probed by more than qt queries is cut into chunks of up to 64 queries (a last
    // chunk of 17 .. 32 takes the 32-query form of the kernel: half the time), each streamed ONCE with 2-4 x the arithmetic per row
    // byte (the matrix pipe, not the row stream, bounds it); a remainder of at most qt queries joins the narrow classes.  SC_IVF_WIDE=0 switches the class off (A/B, tests of the narrow classes).
    const int qw = 64, qw2 = 32;  // the two group widths of scan_listgemm_kernel
    bool wide_ok = sc_scan_listgemm_supported(ix->ld, k) && qt == 16;
    if (const char* e = getenv("SC_IVF_WIDE"))
        if (e[0] == '0') wide_ok = false;
    int64_t work_rows = 0, work_rows_w = 0, longest = 0;
    for (int l = 0; l < nlist; ++l) {
        const int m = start[(size_t)l + 1] - start[(size_t)l];
        const int64_t len = ix->list_off_h[(size_t)l + 1] - ix->list_off_h[(size_t)l];
        if (m == 0 || len <= 0) continue;
        int mw = 0, c = 0;  // wide chunks of this list, in units of a 64-query chunk's time (a 32-query one takes half)
        while (wide_ok && m - c > qt) { mw += m - c > qw2 ? 2 : 1; c += std::min(qw, m - c); }
        work_rows_w += len * mw / 2;
        work_rows += len * ((m - c + qt - 1) / qt);
        longest = std::max(longest, len);
    }
    // parts: ~8 narrow groups per CU (each streams at the LDS-DMA rate), ~6 wide ones (each 2.5 x longer per row)
    int64_t target = std::min<int64_t>(8192, std::max<int64_t>(512, work_rows / ((int64_t)rt->cus * 8)));
    target = (target + 15) & ~(int64_t)15;
    int64_t target_w = std::min<int64_t>(8192, std::max<int64_t>(512, work_rows_w / ((int64_t)rt->cus * 6)));
    target_w = (target_w + 63) & ~(int64_t)63;
    auto parts_of = [&](int64_t len) { return (int)std::max<int64_t>((len + target - 1) / target, wide_ok ? (len + target_w - 1) / target_w : 0); };
    int maxparts = std::max(1, parts_of(longest));
    while (maxparts > 1 && !sc_topk_gather_merge_supported(nprobe * maxparts, k)) {  // merge capacity: fewer, longer parts
        target *= 2;
        target_w *= 2;
        maxparts = std::max(1, parts_of(longest));
    }
    const int L = nprobe * maxparts;
    std::vector<int32_t> src((size_t)Q * L, -1), qmap, qmap_w, qmap_w2;
    std::vector<int> sb;
    std::vector<int64_t> sr, sr_w, sr_w2;
    // With the streamed-query scan (long rows, qt = 16) a group of few queries is still better off on the resident variant,
    // which streams ~30 % faster: groups are numbered in classes -- the wide ones first (64, then 32 query slots), then those
    // with more queries than fit resident (qt_res), then the small ones -- and each class gets its own launch.
    ScanPlan plan_res = plan;
    if (plan.qstream && !sc_scan_exact_plan(ix->ld, 16, k, rt->cus, &plan_res, 16, 1)) plan_res = plan;
    int qt_res = plan_res.qstream ? qt : std::min(qt, plan_res.qt);
    if (const char* e = getenv("SC_SCAN_QSTREAM"))
        if (plan.qstream && e[0] != '0') qt_res = 0;  // forced: every group on the streamed variant (tests, A/B)
    // Within a class the longest parts go first: one workgroup streams one group, and a 100 MB part that starts in the last
    // round would leave the other CUs idle for its whole length (stable sort: the numbering stays deterministic).
    struct GroupDesc { int l, c, nqg, part; int64_t p0, p1; };
    std::vector<GroupDesc> descs;
    int G = 0, G_big = 0, Gw = 0, Gw2 = 0;  // narrow groups (G_big of them on the streamed variant), wide groups of 64 / 32 slots
    int64_t lists_w = 0;                    // k-lists of the wide classes: they come first in `partial`
    for (int cls = -2; cls < 2; ++cls) {    // -2 wide 64, -1 wide 32, 0 narrow / streamed queries, 1 narrow / resident queries
        if (cls == 0) lists_w = (int64_t)Gw * qw + (int64_t)Gw2 * qw2;
        if (cls == 1) G_big = G;
        if (cls < 0 && !wide_ok) continue;
        descs.clear();
        for (int l = 0; l < nlist; ++l) {
            const int64_t first = ix->list_off_h[(size_t)l], end = ix->list_off_h[(size_t)l + 1];
            const int m = start[(size_t)l + 1] - start[(size_t)l];
            if (m == 0 || end <= first) continue;  // nobody probes it / empty list
            // the chunks of this list's queries: wide ones of up to 64 while more than qt remain (17 .. 32 left: the 32-slot
            // form), then narrow ones of up to qt
            int c = 0;
            while (c < m) {
                const bool wide = wide_ok && m - c > qt;
                const int nqg = std::min(wide ? qw : qt, m - c);
                const int want = wide ? (nqg > qw2 ? -2 : -1) : (nqg > qt_res ? 0 : 1);
                if (want == cls) {
                    const int64_t tg = wide ? target_w : target;
                    const int parts = (int)((end - first + tg - 1) / tg);
                    for (int part = 0; part < parts; ++part) {
                        const int64_t p0 = first + (int64_t)part * tg;
                        descs.push_back({l, c, nqg, part, p0, std::min(end, p0 + tg)});
                    }
                }
                c += nqg;
            }
        }
        std::stable_sort(descs.begin(), descs.end(), [](const GroupDesc& x, const GroupDesc& y) { return x.p1 - x.p0 > y.p1 - y.p0; });
        for (const GroupDesc& d : descs) {
            const int stride = cls == -2 ? qw : cls == -1 ? qw2 : qt;
            const int64_t base = cls == -2 ? (int64_t)Gw * qw : cls == -1 ? (int64_t)Gw * qw + (int64_t)Gw2 * qw2 : lists_w + (int64_t)G * qt;
            std::vector<int32_t>& qm = cls == -2 ? qmap_w : cls == -1 ? qmap_w2 : qmap;
            for (int sl = 0; sl < stride; ++sl) {
                if (sl < d.nqg) {
                    const int32_t pair = pair_of[(size_t)start[(size_t)d.l] + d.c + sl];
                    const int q = pair / nprobe, j = pair - q * nprobe;
                    qm.push_back(q);
                    src[(size_t)q * L + (size_t)j * maxparts + d.part] = (int32_t)(base + sl);
                } else {
                    qm.push_back(-1);
                }
            }
            if (cls < 0) {
                std::vector<int64_t>& srw = cls == -2 ? sr_w : sr_w2;
                srw.push_back(d.p0);
                srw.push_back(d.p1);
                if (cls == -2) ++Gw; else ++Gw2;
            } else {
                sb.push_back(0);
                sb.push_back((int)((d.p1 - d.p0 + 15) >> 4));
                sr.push_back(d.p0);
                sr.push_back(d.p1);
                ++G;
            }
        }
    }
    const double t_plan = since();
    // 3. plan tables -> device (they replace the probe results in the scratch buffer), queries padded + normed
    off = 0;
    const size_t o_src = carve(src.size() * 4), o_qmap = carve((size_t)G * qt * 4 + 16), o_sb = carve((size_t)G * 2 * 4 + 16),
                 o_sr = carve((size_t)G * 2 * 8 + 16), o_qmap_w = carve((size_t)Gw * qw * 4 + 16), o_sr_w = carve((size_t)Gw * 2 * 8 + 16),
                 o_qmap_w2 = carve((size_t)Gw2 * qw2 * 4 + 16), o_sr_w2 = carve((size_t)Gw2 * 2 * 8 + 16);
    st = sc_grow(ix, &ix->ivf_scratch, &ix->ivf_scratch_cap, off);
    if (st) return st;
    b = (char*)ix->ivf_scratch;
    SC_HIP(hipMemcpyAsync(b + o_src, src.data(), src.size() * 4, hipMemcpyHostToDevice, s));
    if (G > 0) {
        SC_HIP(hipMemcpyAsync(b + o_qmap, qmap.data(), (size_t)G * qt * 4, hipMemcpyHostToDevice, s));
        SC_HIP(hipMemcpyAsync(b + o_sb, sb.data(), (size_t)G * 2 * 4, hipMemcpyHostToDevice, s));
        SC_HIP(hipMemcpyAsync(b + o_sr, sr.data(), (size_t)G * 2 * 8, hipMemcpyHostToDevice, s));
    }
    if (Gw > 0) {
        SC_HIP(hipMemcpyAsync(b + o_qmap_w, qmap_w.data(), (size_t)Gw * qw * 4, hipMemcpyHostToDevice, s));
        SC_HIP(hipMemcpyAsync(b + o_sr_w, sr_w.data(), (size_t)Gw * 2 * 8, hipMemcpyHostToDevice, s));
    }
    if (Gw2 > 0) {
        SC_HIP(hipMemcpyAsync(b + o_qmap_w2, qmap_w2.data(), (size_t)Gw2 * qw2 * 4, hipMemcpyHostToDevice, s));
        SC_HIP(hipMemcpyAsync(b + o_sr_w2, sr_w2.data(), (size_t)Gw2 * 2 * 8, hipMemcpyHostToDevice, s));
    }
    st = sc_grow(ix, (void**)&ix->qpad, &ix->qpad_cap, (size_t)Q * ix->ld * 4);
    if (st) return st;
    st = sc_grow(ix, (void**)&ix->qnorm, &ix->qnorm_cap, (size_t)Q * 4);
    if (st) return st;
    st = sc_grow(ix, (void**)&ix->partial, &ix->partial_cap, std::max<size_t>(((size_t)lists_w + (size_t)G * qt) * k * 8, 16));
    if (st) return st;
    sc_launch_ingest_rows(q_dev, nullptr, 0, Q, ix->dim, ix->qpad, ix->ld, ix->qnorm, s);
    // 4. one workgroup per group (grid.y is limited to 65535 groups per launch); the wide class goes first: its workgroups are
    // the long ones
    hipEvent_t e0, e1;
    sc_prof_begin(rt, SC_PROF_SCAN, &e0, &e1);
    sc_launch_scan_listgemm((int)ix->metric, qw, ix->X, ix->xnorm, ix->ld, ix->qpad, ix->qnorm, k, Gw, ix->partial, ix->perm,
                            (const int64_t*)(b + o_sr_w), (const int32_t*)(b + o_qmap_w), s);
    sc_launch_scan_listgemm((int)ix->metric, qw2, ix->X, ix->xnorm, ix->ld, ix->qpad, ix->qnorm, k, Gw2, ix->partial + (size_t)Gw * qw * k, ix->perm,
                            (const int64_t*)(b + o_sr_w2), (const int32_t*)(b + o_qmap_w2), s);
    for (int cls = 0; cls < 2; ++cls)
      for (int g0 = cls ? G_big : 0, hi = cls ? G : G_big; g0 < hi; g0 += 65535) {
        const int gn = std::min(65535, hi - g0);
        ScanPlan p = cls ? plan_res : plan;
        p.groups = gn;
        p.nwg = 1;
        p.lists = 1;
        p.gstride = qt;
        sc_launch_scan_exact((int)ix->metric, ix->X, ix->xnorm, ix->n, ix->ld, ix->qpad, ix->qnorm, gn * qt, k, p,
                             ix->partial + ((size_t)lists_w + (size_t)g0 * qt) * k, ix->perm, (const int*)(b + o_sb) + (size_t)g0 * 2,
                             (const int64_t*)(b + o_sr) + (size_t)g0 * 2, 1, s, (const int32_t*)(b + o_qmap) + (size_t)g0 * qt);
    }
    sc_prof_end(rt, SC_PROF_SCAN, e0, e1);
    // 5. a query's result = merge of the nprobe (group, slot) lists it took part in
    sc_launch_topk_gather_merge((int)ix->metric, ix->partial, (const int32_t*)(b + o_src), L, Q, k, ix->row_base, out_dist, out_rows, s);
    SC_HIP(hipGetLastError());
    SC_HIP(hipStreamSynchronize(s));  // the host plan vectors go out of scope
    if (trace) {
        double streamed = 0.0, streamed_big = 0.0;
        for (size_t g = 0; g < sr.size(); g += 2) {
            streamed += (double)(sr[g + 1] - sr[g]);
            if ((int)(g / 2) < G_big) streamed_big += (double)(sr[g + 1] - sr[g]);
        }
        double streamed_wide = 0.0, streamed_wide2 = 0.0;
        for (size_t g = 0; g < sr_w.size(); g += 2) streamed_wide += (double)(sr_w[g + 1] - sr_w[g]);
        for (size_t g = 0; g < sr_w2.size(); g += 2) streamed_wide2 += (double)(sr_w2[g + 1] - sr_w2[g]);
        streamed += streamed_wide + streamed_wide2;
        fprintf(stderr, "[ivf list-major] wide groups: %d of <= 64 queries = %.1f GB, %d of <= 32 = %.1f GB\n", Gw, streamed_wide * (double)ix->ld * 4.0 / 1e9, Gw2,
                streamed_wide2 * (double)ix->ld * 4.0 / 1e9);
        streamed *= (double)ix->ld * 4.0;
        streamed_big *= (double)ix->ld * 4.0;
        const double t_scan = since() - t_plan;
        fprintf(stderr, "[ivf list-major] Q=%d nprobe=%d qt=%d groups=%d (%d = %.1f GB on streamed queries) parts<=%d target=%lld / %lld rows | D2H of probes %.3f ms, host plan %.3f ms, H2D + scan + merge %.3f ms = %.1f GB at %.2f TB/s\n",
                Q, nprobe, qt, G, plan.qstream ? G_big : 0, plan.qstream ? streamed_big / 1e9 : 0.0, maxparts, (long long)target, (long long)target_w, t_probe, t_plan - t_probe, t_scan, streamed / 1e9, streamed / 1e9 / t_scan);
    }
    ix->last_path = 4;
    ix->last_probed_lists = nprobe;
    ix->last_groups = G + Gw + Gw2;
    for (size_t g = 0; g < sr.size(); g += 2) ix->last_streamed_rows += sr[g + 1] - sr[g];
    for (size_t g = 0; g < sr_w.size(); g += 2) ix->last_streamed_rows += sr_w[g + 1] - sr_w[g];
    for (size_t g = 0; g < sr_w2.size(); g += 2) ix->last_streamed_rows += sr_w2[g + 1] - sr_w2[g];
    for (int l = 0; l < nlist; ++l)
        if (start[(size_t)l + 1] > start[(size_t)l]) ix->last_unique_rows += ix->list_off_h[(size_t)l + 1] - ix->list_off_h[(size_t)l];
    return SC_OK;
}

// List of every row, in insertion order (persistence: together with the centroids this restores the lists without k-means).
extern "C" sc_status sc_index_ivf_assignments(sc_index* ix, int32_t* out) {
    if (!ix || !out) return sc_fail(SC_ERR_INVALID, "sc_index_ivf_assignments: NULL argument");
    std::lock_guard<std::mutex> g(ix->mu);
    if (!ix->trained) return sc_fail(SC_ERR_STATE, "sc_index_ivf_assignments: the index is not trained");
    SC_HIP(hipSetDevice(ix->rt->device));
    sc_status st = sc_ivf_refresh_locked(ix);  // rows upserted since the build get their list first
    if (st) return st;
    memcpy(out, ix->assign_h.data(), (size_t)ix->n * sizeof(int32_t));
    return SC_OK;
}

// Build the lists for GIVEN centroids (no k-means): every row goes to its nearest centroid under the assignment metric.  Multi-GPU
// IVF: one rank trains, broadcasts its centroids, every rank calls this on its shard -- probing then means the same lists on every
// shard, and the merged result equals that of one index over the whole corpus with these centroids.
extern "C" sc_status sc_index_assign_lists(sc_index* ix, const float* centroids, int32_t nlist) {
    if (!ix || !centroids || nlist < 1) return sc_fail(SC_ERR_INVALID, "sc_index_assign_lists: bad argument");
    std::lock_guard<std::mutex> g(ix->mu);
    if (ix->kind != SC_INDEX_IVF_FLAT) return sc_fail(SC_ERR_STATE, "sc_index_assign_lists: index kind is not IVF_FLAT");
    if (ix->n < 1) return sc_fail(SC_ERR_STATE, "sc_index_assign_lists: the index is empty");
    SC_HIP(hipSetDevice(ix->rt->device));
    sc_status st = sc_ivf_untrain_locked(ix);
    if (st) return st;
    if (ix->quant) {
        sc_index_destroy(ix->quant);
        ix->quant = nullptr;
    }
    st = sc_index_create(ix->rt, ix->dim, assign_metric(ix->metric), SC_INDEX_FLAT, 0, 0, &ix->quant);
    if (st) return st;
    st = sc_index_add(ix->quant, centroids, nlist);
    if (st) return st;
    return ivf_assign_all_and_install_locked(ix, nlist);
}

// Install a previously trained IVF structure: centroids [nlist, dim] (tight f32) and the list of every row.
extern "C" sc_status sc_index_set_ivf(sc_index* ix, const float* centroids, const int32_t* assign, int32_t nlist) {
    if (!ix || !centroids || !assign || nlist < 1) return sc_fail(SC_ERR_INVALID, "sc_index_set_ivf: bad argument");
    std::lock_guard<std::mutex> g(ix->mu);
    if (ix->kind != SC_INDEX_IVF_FLAT) return sc_fail(SC_ERR_STATE, "sc_index_set_ivf: index kind is not IVF_FLAT");
    if (ix->n < 1) return sc_fail(SC_ERR_STATE, "sc_index_set_ivf: the index is empty");
    for (int64_t i = 0; i < ix->n; ++i)
        if (assign[i] < 0 || assign[i] >= nlist) return sc_fail(SC_ERR_INVALID, "sc_index_set_ivf: assign[%lld] = %d outside [0,%d)", (long long)i, assign[i], nlist);
    SC_HIP(hipSetDevice(ix->rt->device));
    sc_status st = sc_ivf_untrain_locked(ix);
    if (st) return st;
    if (ix->quant) {
        sc_index_destroy(ix->quant);
        ix->quant = nullptr;
    }
    st = sc_index_create(ix->rt, ix->dim, assign_metric(ix->metric), SC_INDEX_FLAT, 0, 0, &ix->quant);
    if (st) return st;
    st = sc_index_add(ix->quant, centroids, nlist);
    if (st) return st;
    return ivf_install_lists_locked(ix, nlist, std::vector<int32_t>(assign, assign + ix->n));
}

// Centroids (tight [nlist, dim]) and list sizes back to the host (tests, persistence).
extern "C" sc_status sc_index_ivf_info(sc_index* ix, int32_t* nlist, float* centroids, int64_t* list_sizes) {
    if (!ix) return sc_fail(SC_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> g(ix->mu);
    if (!ix->trained || !ix->quant) {
        if (nlist) *nlist = 0;
        return SC_OK;
    }
    SC_HIP(hipSetDevice(ix->rt->device));
    {
        sc_status st = sc_ivf_refresh_locked(ix);
        if (st) return st;
    }
    if (nlist) *nlist = ix->nlist_trained;
    if (centroids) {
        sc_status st = sc_index_get_rows(ix->quant, 0, ix->nlist_trained, centroids);
        if (st) return st;
    }
    if (list_sizes)
        for (int c = 0; c < ix->nlist_trained; ++c) list_sizes[c] = ix->list_off_h[(size_t)c + 1] - ix->list_off_h[(size_t)c];
    return SC_OK;
}

// ---- list-major probing behind an int8 coarse stage (L2) ---------------------------------------------------------------------------
// ivf_coarse.hip has the idea: every list quantised relative to its centroid, one centred int8 query per (query, probed list)
// pair, coarse scores turned into lower bounds of the exact distance, then bound and refine.  Here: the plan (pairs bucketed by
// list, groups of up to 64 query slots, one work item per 256-row tile of a list x group) and the sequence
//   phase A  every query's NEAREST list(s) through the grouped streaming kernel (scan_coarse64s_kernel<GROUPED, DENSE>): all lower-bound
//            keys kept; the 128 best re-scored exactly (canonical fmaf chain from the original rows) -> T = k-th exact distance + allowance;
//   phase B  the other nprobe - 1 lists: rows with lower bound <= T survive;
//   refine   all survivors of both phases within T re-scored exactly, exact top-k.
// No probed row outside the re-scored set can be closer than the k-th result, so the results are those of the exact list-major path,
// bit for bit.  Queries whose survivor lists overflow or whose refine set exceeds 4096 rows are probed again exactly
// (scan_listgemm / scan_exact kernels).
static const int IVFC_CAP = 8192;  // survivors per query, phase A (everything it sees: 2 KP + one prefix at most)
static const int IVFC_CAPB = 16384; // ... and phase B (the rest of a long nearest list against the prefix's bound can be thousands)
static const int64_t IVFC_PREFIX = 4096;  // rows of one list that phase A takes (all of them are kept: cap > 2 KP + prefix)
static int ivfc_ld8(const sc_index* ix) { return (ix->ld + 127) / 128 * 128; }

bool sc_ivf_coarse_applicable(const sc_index* ix, int Q, int k, int nprobe) {
    static const bool env_off = [] { const char* e = getenv("SC_IVF_COARSE"); return e && e[0] == '0'; }();
    if (ix->kind != SC_INDEX_IVF_FLAT || !ix->trained || !ix->quant || !ix->perm || (ix->metric != SC_METRIC_L2 && ix->metric != SC_METRIC_IP && ix->metric != SC_METRIC_COSINE)) return false;
    if (nprobe < 2 || nprobe > 512 || nprobe >= ix->nlist_trained || k < 1 || k > sc_batched_kprime8() / 2) return false;
    if (ix->search_mode == 5) return Q >= 1;
    if (ix->search_mode != 0 || env_off || ix->ivfc_off) return false;
    // auto: any batch over a corpus worth a shadow.  The first rule here (Q >= 64 and Q nprobe >= nlist: "every list wanted by several
    // queries") was list-major thinking -- the stage's gain is the int8 bytes, not the sharing: at config 5 a batch of 32 queries
    // takes 22.9 ms through the per-query probe, 4.9 ms list-major exact and 1.9 ms here; of 2 queries 1.5 / 1.4 / 0.7 ms
    // (scripts/ivf_small_batch.py, profiles/r3z_ivf_small_batch*.log).  Fewer than 8 queries take it when ONE probe would stream a
    // gigabyte of f32 rows or more (config 5, one query: 0.50 ms against 0.83 through the per-query probe; 10M x 768 at the
    // reference's nlist 128 / nprobe 16: 0.51 against 0.67); below that the plan's round trip to the host costs more than the
    // bytes saved (1M x 768, 2 queries: 0.32 ms against 0.14).
    if (ix->n < 100000) return false;
    if (Q >= 8) return true;
    const double probed_bytes = (double)nprobe * ((double)ix->n / (double)ix->nlist_trained) * (double)ix->ld * 4.0;
    return probed_bytes >= 1.0e9;
}

static int g_ivfc_nomem = 0;  // sc_diag_set_option("ivf_coarse_nomem", 1): the centred shadow cannot be allocated (tests of the fallback to the exact probe)
void sc_ivf_set_coarse_nomem(int v) { g_ivfc_nomem = v; }
static sc_status ivfc_ensure_shadow(sc_index* ix) {
    if (ix->shadowc_rows == ix->ivf_rows && ix->Xc8) {
        if (ix->dirty_c8.empty()) return SC_OK;
        // rows of the lists overwritten in place and still in their lists (sc_ivf_refresh_locked found nothing to move, or the layout
        // would have been rebuilt and this shadow with it): their shadow rows alone; the per-list maxima keep accumulating
        hipStream_t s = ix->rt->stream;
        std::vector<int64_t>& d = ix->dirty_c8;
        std::sort(d.begin(), d.end());
        d.erase(std::unique(d.begin(), d.end()), d.end());
        while (!d.empty() && d.back() >= ix->shadowc_rows) d.pop_back();
        if (!d.empty()) {
            sc_status st = sc_grow(ix, (void**)&ix->stage, &ix->stage_cap, d.size() * 8);
            if (st) return st;
            SC_HIP(hipMemcpyAsync(ix->stage, d.data(), d.size() * 8, hipMemcpyHostToDevice, s));
            const bool unit = ix->metric == SC_METRIC_COSINE;
            sc_launch_ivf_center_shadow(ix->X, (int64_t)d.size(), ix->ld, ivfc_ld8(ix), ix->quant->X, ix->quant->ld, ix->list_off, ix->nlist_trained, ix->Xc8, ix->xcs,
                                        ix->list_stats, s, unit ? ix->xnorm : nullptr, unit ? ix->quant->xnorm : nullptr, (const int64_t*)ix->stage);
            sc_launch_norm_max(ix->xnorm, ix->shadowc_rows, ix->list_stats + (size_t)ix->nlist * 2, s);
            SC_HIP(hipGetLastError());
            SC_HIP(hipStreamSynchronize(s));  // (the position list is a host temporary behind an asynchronous copy)
        }
        d.clear();
        return SC_OK;
    }
    ix->dirty_c8.clear();  // a full build covers them
    if (g_ivfc_nomem) return sc_fail(SC_ERR_NOMEM, "ivf coarse stage: out of device memory (forced by sc_diag_set_option)");
    hipStream_t s = ix->rt->stream;
    const int ld8 = ivfc_ld8(ix), nlist = ix->nlist_trained;
    const int64_t rows = ix->ivf_rows, rows_pad = (rows + 255) / 256 * 256 + 256;  // (a list's last tile reads up to 255 rows beyond its end)
    sc_status st = sc_grow(ix, &ix->Xc8, &ix->xc8_cap, (size_t)rows_pad * ld8);
    if (st) return st;
    st = sc_grow(ix, (void**)&ix->xcs, &ix->xcsn_cap, (size_t)rows_pad * 16);
    if (st) return st;
    if (!ix->list_stats) SC_HIP(hipMalloc((void**)&ix->list_stats, ((size_t)ix->nlist * 2 + 4) * 4));
    SC_HIP(hipMemsetAsync(ix->list_stats, 0, ((size_t)ix->nlist * 2 + 4) * 4, s));
    SC_HIP(hipMemsetAsync((char*)ix->Xc8 + (size_t)rows * ld8, 0, (size_t)(rows_pad - rows) * ld8, s));
    SC_HIP(hipMemsetAsync(ix->xcs + rows * 4, 0, (size_t)(rows_pad - rows) * 16, s));
    const bool unit = ix->metric == SC_METRIC_COSINE;  // the IP form on normalised rows and centroids
    sc_launch_ivf_center_shadow(ix->X, rows, ix->ld, ld8, ix->quant->X, ix->quant->ld, ix->list_off, nlist, ix->Xc8, ix->xcs, ix->list_stats, s,
                                unit ? ix->xnorm : nullptr, unit ? ix->quant->xnorm : nullptr);
    sc_launch_norm_max(ix->xnorm, rows, ix->list_stats + (size_t)ix->nlist * 2, s);  // bits of max |x|^2: the re-rank's rounding allowance
    SC_HIP(hipGetLastError());
    ix->shadowc_rows = rows;
    return SC_OK;
}

sc_status sc_ivf_search_coarse_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe, float* out_dist, int64_t* out_rows) {
    sc_runtime* rt = ix->rt;
    hipStream_t s = rt->stream;
    sc_index* qz = ix->quant;
    const int nlist = ix->nlist_trained, ld = ix->ld, ld8 = ivfc_ld8(ix), KP = sc_batched_kprime8();
    sc_status st = ivfc_ensure_shadow(ix);
    if (st) return st;
    const size_t npairs = (size_t)Q * nprobe;
    // 1. coarse probe (the quantizer's own search) -> host
    {
        size_t off = 0;
        auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
        const size_t o_pd = carve(npairs * 4), o_pr = carve(npairs * 8);
        st = sc_grow(ix, &ix->ivf_scratch, &ix->ivf_scratch_cap, off);
        if (st) return st;
        char* b = (char*)ix->ivf_scratch;
        std::lock_guard<std::mutex> gq(qz->mu);
        const sc_metric saved = qz->metric;
        qz->metric = ix->metric;
        st = sc_search_flat_locked(qz, q_dev, Q, nprobe, (float*)(b + o_pd), (int64_t*)(b + o_pr));
        qz->metric = saved;
        if (st) return st;
    }
    std::vector<int64_t> probes(npairs);
    SC_HIP(hipMemcpyAsync(probes.data(), (char*)ix->ivf_scratch + ((npairs * 4 + 255) & ~(size_t)255), npairs * 8, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    // 2. plan.  Phase A = every query's nearest list(s) -- of a long list only its first IVFC_PREFIX rows: any subset gives a valid
    // bound, and phase A keeps every row it sees --, phase B = the other lists and what is left of the phase-A lists; per kind the pairs are
    // bucketed by list (queries in ascending order: deterministic), cut into groups of 64 slots, and every group meets every 256-row
    // tile of its row range.
    struct Item { long long row0; int rows; int slot_base; };
    std::vector<int32_t> slot_q, slot_l, slot_dst;  // slot_dst (phase A): where the list's rows go in the query's survivor list
    std::vector<unsigned> cntA((size_t)Q, 0u);       // rows of every query's phase-A ranges
    std::vector<Item> items[3];  // phase A | the rest of long phase-A lists | the other lists
    int64_t streamed_rows = 0, unique_rows = 0;
    std::vector<char> touched((size_t)nlist, 0);
    auto list_len = [&](int64_t l) { return ix->list_off_h[(size_t)l + 1] - ix->list_off_h[(size_t)l]; };
    // phase A of a query = its nearest lists until they hold 2 KP rows (one list unless the lists are small): enough candidates
    // for a tight bound, which phase B needs -- at +inf every row of the other lists would survive
    std::vector<int> ja((size_t)Q, 1);
    for (int q = 0; q < Q; ++q) {
        int64_t cum = 0;
        int j = 0;
        while (j < nprobe && cum < 2 * (int64_t)KP) {
            const int64_t l = probes[(size_t)q * nprobe + j];
            if (l >= 0 && l < nlist) cum += std::min<int64_t>(list_len(l), IVFC_PREFIX);
            ++j;
        }
        ja[(size_t)q] = j;
    }
    std::vector<int> group_bases;
    // the three kinds of (query, list) entries -- 0: phase A (prefix of the list); 1: phase B, whole list; 2: phase B, the rest of a
    // phase-A list -- bucketed by list in two passes over the probe table (count, fill)
    std::vector<int> start3[3];
    std::vector<int32_t> qs3[3];
    for (int kind = 0; kind < 3; ++kind) start3[kind].assign((size_t)nlist + 1, 0);
    for (int q = 0; q < Q; ++q) {
        const int64_t* pq = probes.data() + (size_t)q * nprobe;
        const int jaq = ja[(size_t)q];
        for (int j = 0; j < nprobe; ++j) {
            const int64_t l = pq[j];
            if (l < 0 || l >= nlist) continue;
            if (j < jaq) {
                ++start3[0][(size_t)l + 1];
                if (list_len(l) > IVFC_PREFIX) ++start3[2][(size_t)l + 1];
            } else {
                ++start3[1][(size_t)l + 1];
            }
        }
    }
    std::vector<int> fill3[3];
    for (int kind = 0; kind < 3; ++kind) {
        for (int l = 0; l < nlist; ++l) start3[kind][(size_t)l + 1] += start3[kind][(size_t)l];
        fill3[kind].assign(start3[kind].begin(), start3[kind].end() - 1);
        qs3[kind].resize((size_t)start3[kind][(size_t)nlist]);
    }
    for (int q = 0; q < Q; ++q) {
        const int64_t* pq = probes.data() + (size_t)q * nprobe;
        const int jaq = ja[(size_t)q];
        for (int j = 0; j < nprobe; ++j) {
            const int64_t l = pq[j];
            if (l < 0 || l >= nlist) continue;
            if (j < jaq) {
                qs3[0][(size_t)fill3[0][(size_t)l]++] = q;
                if (list_len(l) > IVFC_PREFIX) qs3[2][(size_t)fill3[2][(size_t)l]++] = q;
            } else {
                qs3[1][(size_t)fill3[1][(size_t)l]++] = q;
            }
        }
    }
    slot_q.reserve(npairs * 2 + 64);
    slot_l.reserve(npairs * 2 + 64);
    slot_dst.reserve(npairs * 2 + 64);
    int64_t rows_kind[3] = {0, 0, 0};
    for (int kind : {0, 2, 1}) {  // (in this order in memory: the tails form a launch of their own when the bound has two levels)
        const std::vector<int>& start = start3[kind];
        const std::vector<int32_t>& qs = qs3[kind];
        if (start[(size_t)nlist] == 0) continue;
        const int ph = kind == 0 ? 0 : kind == 2 ? 1 : 2;
        for (int l = 0; l < nlist; ++l) {
            const int m = start[(size_t)l + 1] - start[(size_t)l];
            const int64_t lfirst = ix->list_off_h[(size_t)l], lend = ix->list_off_h[(size_t)l + 1];
            const int64_t first = kind == 2 ? lfirst + IVFC_PREFIX : lfirst, end = kind == 0 ? std::min(lend, lfirst + IVFC_PREFIX) : lend;
            if (m == 0 || end <= first) continue;
            if (!touched[(size_t)l]) { touched[(size_t)l] = 1; unique_rows += lend - lfirst; }
            group_bases.clear();
            for (int c = 0; c < m; c += 64) {
                const int nq = std::min(64, m - c);
                const int slot_base = (int)slot_q.size();
                for (int sl = 0; sl < 64; ++sl) {
                    const int q = sl < nq ? qs[(size_t)start[(size_t)l] + c + sl] : -1;
                    slot_q.push_back(q);
                    slot_l.push_back(sl < nq ? l : -1);
                    int32_t dst = 0;
                    if (ph == 0 && q >= 0) {
                        dst = (int32_t)((uint32_t)cntA[(size_t)q] - (uint32_t)first);  // position of stored row r: (uint32)(r + dst)
                        cntA[(size_t)q] = (unsigned)std::min<int64_t>((int64_t)cntA[(size_t)q] + (end - first), (int64_t)1 << 30);
                    }
                    slot_dst.push_back(dst);
                }
                group_bases.push_back(slot_base);
                streamed_rows += end - first;
                rows_kind[kind] += (end - first) * nq;
            }
            // items of one list: blocks of 8 row tiles, every group's copy of a block right behind the previous group's -- the persistent
            // grid hands item i to workgroup i mod grid, workgroups 8 apart share an XCD, so the groups of a popular list stream a
            // tile through the same L2 at the same time instead of one XCD after the other
            for (int64_t b0 = first; b0 < end; b0 += 8 * 256)
                for (const int base : group_bases)
                    for (int64_t r0 = b0; r0 < end && r0 < b0 + 8 * 256; r0 += 256) items[ph].push_back({(long long)r0, (int)std::min<int64_t>(256, end - r0), base});
        }
    }
    const int nslots = (int)slot_q.size();
    const size_t nitems = items[0].size() + items[1].size() + items[2].size();
    // Long lists (the reference's nlist = 128: tens of thousands of rows each): the bound from a 4 096-row prefix of the nearest list is
    // loose, and against it the REST of that list -- where most neighbours are -- overflows the survivor lists (10M x 768, nlist 128:
    // 65 of 1 024 queries).  Then the bound gets a second level: the tails run first, their 128 best lower bounds are re-scored too,
    // and the other lists meet the bound of both samples.
    const bool two_level = !items[1].empty() && rows_kind[2] * 4 >= rows_kind[0];
    // 3. scratch: slot tables, per-pair queries, per-query state, survivor lists of both phases, hit lists, the refine stage's sets
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const int Qpad = (Q + 255) / 256 * 256;
    const int kpa = k <= sc_batched_kprime() / 2 ? sc_batched_kprime() : KP;  // phase-A candidates re-scored for the bound (128; 512 for k > 64)
    const int WCAP = sc_ivf_widen_cap();
    const size_t o_sq = carve((size_t)nslots * 4), o_sl = carve((size_t)nslots * 4), o_qs = carve((size_t)nslots * 4), o_qn = carve((size_t)nslots * 4),
                 o_st = carve((size_t)nslots * 4), o_stf = carve((size_t)nslots * 4), o_qb = carve((size_t)nslots * 4), o_qd = carve((size_t)nslots * 4),
                 o_se = carve((size_t)nslots * 4), o_sd = carve((size_t)nslots * 4), o_items = carve(nitems * sizeof(Item)), o_qc = carve((size_t)nslots * ld8),
                 o_thr = carve((size_t)Qpad * 4), o_tf = carve((size_t)Qpad * 4), o_cnt = carve((size_t)Q * 4), o_cntA = carve((size_t)Q * 4), o_ovf = carve((size_t)Q * 4),
                 o_flag = carve((size_t)Q * 4), o_nc = carve((size_t)Q * 4), o_best = carve((size_t)Q * kpa * 8), o_ekA = carve((size_t)Q * kpa * 8),
                 o_bestT = carve((size_t)Q * kpa * 8), o_ekT = carve((size_t)Q * kpa * 8), o_cntT = carve((size_t)Q * 4), o_thrT = carve((size_t)Qpad * 4), o_tfT = carve((size_t)Qpad * 4),
                 o_survA = carve((size_t)Q * IVFC_CAP * 8), o_survB = carve((size_t)Q * IVFC_CAPB * 8), o_cand = carve((size_t)Q * WCAP * 8),
                 o_ek2 = carve((size_t)Q * WCAP * 8);
    const size_t hit_bytes = (size_t)2048 * (4 + 8192 * 16) + 256;
    const size_t o_hits = carve(hit_bytes);
    st = sc_grow(ix, &ix->ivfc_scratch, &ix->ivfc_scratch_cap, off);
    if (st) return st;
    char* b = (char*)ix->ivfc_scratch;
    int32_t *d_sq = (int32_t*)(b + o_sq), *d_sl = (int32_t*)(b + o_sl), *d_sd = (int32_t*)(b + o_sd);
    float *d_qs = (float*)(b + o_qs), *d_qn = (float*)(b + o_qn), *d_st = (float*)(b + o_st), *d_stf = (float*)(b + o_stf), *d_qb = (float*)(b + o_qb),
          *d_qd = (float*)(b + o_qd), *d_se = (float*)(b + o_se);
    float *thr = (float*)(b + o_thr), *tf = (float*)(b + o_tf);
    unsigned *cnt = (unsigned*)(b + o_cnt), *cntA_d = (unsigned*)(b + o_cntA);
    int *ovf = (int*)(b + o_ovf), *flags = (int*)(b + o_flag), *ncand = (int*)(b + o_nc);
    uint64_t *best = (uint64_t*)(b + o_best), *ekeysA = (uint64_t*)(b + o_ekA), *survA = (uint64_t*)(b + o_survA), *survB = (uint64_t*)(b + o_survB),
             *cand2 = (uint64_t*)(b + o_cand), *ekeys2 = (uint64_t*)(b + o_ek2);
    SC_HIP(hipMemcpyAsync(d_sq, slot_q.data(), (size_t)nslots * 4, hipMemcpyHostToDevice, s));
    SC_HIP(hipMemcpyAsync(d_sl, slot_l.data(), (size_t)nslots * 4, hipMemcpyHostToDevice, s));
    SC_HIP(hipMemcpyAsync(d_sd, slot_dst.data(), (size_t)nslots * 4, hipMemcpyHostToDevice, s));
    if (!items[0].empty()) SC_HIP(hipMemcpyAsync(b + o_items, items[0].data(), items[0].size() * sizeof(Item), hipMemcpyHostToDevice, s));
    if (!items[1].empty())
        SC_HIP(hipMemcpyAsync(b + o_items + items[0].size() * sizeof(Item), items[1].data(), items[1].size() * sizeof(Item), hipMemcpyHostToDevice, s));
    if (!items[2].empty())
        SC_HIP(hipMemcpyAsync(b + o_items + (items[0].size() + items[1].size()) * sizeof(Item), items[2].data(), items[2].size() * sizeof(Item), hipMemcpyHostToDevice, s));
    st = sc_grow(ix, (void**)&ix->qpad, &ix->qpad_cap, (size_t)Q * ld * 4);
    if (st) return st;
    st = sc_grow(ix, (void**)&ix->qnorm, &ix->qnorm_cap, (size_t)Q * 4);
    if (st) return st;
    sc_launch_ingest_rows(q_dev, nullptr, 0, Q, ix->dim, ix->qpad, ld, ix->qnorm, s);
    const int metric = (int)ix->metric;  // L2; IP (rows centred only); COSINE (the IP form on unit vectors) -- ivf_coarse.hip
    const int kmetric = metric == SC_METRIC_COSINE ? (int)SC_METRIC_IP : metric;  // what the streaming kernel computes
    sc_launch_ivf_pair_query(ix->qpad, ld, ld8, qz->X, qz->ld, d_sq, d_sl, nslots, ix->list_stats, b + o_qc, d_qs, d_qn, d_qb, d_qd, d_se, s, metric, ix->qnorm, qz->xnorm);
    sc_launch_scan_batched_init(thr, tf, Qpad, best, cnt, ovf, Q, kpa, s);
    // phase A is dense: every row of its lists survives, at a known place of survA; the counts are known here
    SC_HIP(hipMemcpyAsync(cnt, cntA.data(), (size_t)Q * 4, hipMemcpyHostToDevice, s));
    SC_HIP(hipMemcpyAsync(cntA_d, cntA.data(), (size_t)Q * 4, hipMemcpyHostToDevice, s));
    const unsigned* xmax_bits = ix->list_stats + (size_t)ix->nlist * 2;  // bits of max |x|^2: the rounding allowance of the exact scores
    hipEvent_t e0, e1;
    // 4. phase A: lower bounds of the nearest list(s) -> the kpa best re-scored exactly -> T = their k-th exact distance + allowance
    if (!items[0].empty()) {
        sc_launch_ivf_slot_thr(d_sq, d_qn, d_se, thr, nslots, d_st, d_stf, s);  // (+inf everywhere; the dense form tests nothing)
        sc_prof_begin(rt, SC_PROF_SCAN, &e0, &e1);
        sc_launch_ivf_coarse(ix->Xc8, ix->xcs, ld8, b + o_qc, b + o_items, (int)items[0].size(), d_stf, d_st, d_qn, d_qs, d_sq, d_qb, d_qd, survA, cnt, IVFC_CAP,
                             b + o_hits, hit_bytes, s, d_sd, kmetric);
        sc_prof_end(rt, SC_PROF_SCAN, e0, e1);
    }
    sc_prof_begin(rt, SC_PROF_MERGE, &e0, &e1);
    sc_launch_scan_select(metric, survA, cnt, IVFC_CAP, best, ix->qnorm, thr, tf, ovf, Q, kpa, s);  // (resets cnt: phase B counts from 0)
    sc_launch_scan_rerank_keys(metric, ix->X, ix->xnorm, ld, ix->qpad, ix->qnorm, best, nullptr, kpa, ix->perm, ekeysA, Q, s);
    sc_launch_ivf_bound(metric, ekeysA, kpa, k, ix->qnorm, xmax_bits, ld, thr, Q, s);
    sc_prof_end(rt, SC_PROF_MERGE, e0, e1);
    // 5. phase B: the other lists (and the rest of long phase-A lists) against T
    size_t b_first = items[0].size(), b_count = items[1].size() + items[2].size();
    // (two-level bounds) the 128 best lower bounds among phase B's survivors so far, re-scored: T = min(T, k-th of that sample + phase A's)
    auto tighten = [&]() -> sc_status {
        uint64_t *bestT = (uint64_t*)(b + o_bestT), *ekeysT = (uint64_t*)(b + o_ekT);
        unsigned* cntT = (unsigned*)(b + o_cntT);
        sc_prof_begin(rt, SC_PROF_MERGE, &e0, &e1);
        SC_HIP(hipMemcpyAsync(cntT, cnt, (size_t)Q * 4, hipMemcpyDeviceToDevice, s));  // (the selection resets the counts; the survivors stay where they are)
        SC_HIP(hipMemsetAsync(bestT, 0xFF, (size_t)Q * kpa * 8, s));
        sc_launch_scan_select(metric, survB, cnt, IVFC_CAPB, bestT, ix->qnorm, (float*)(b + o_thrT), (float*)(b + o_tfT), ovf, Q, kpa, s);
        SC_HIP(hipMemcpyAsync(cnt, cntT, (size_t)Q * 4, hipMemcpyDeviceToDevice, s));
        sc_launch_scan_rerank_keys(metric, ix->X, ix->xnorm, ld, ix->qpad, ix->qnorm, bestT, nullptr, kpa, ix->perm, ekeysT, Q, s);
        sc_launch_ivf_bound(metric, ekeysA, kpa, k, ix->qnorm, xmax_bits, ld, thr, Q, s, ekeysT, true);
        sc_prof_end(rt, SC_PROF_MERGE, e0, e1);
        return SC_OK;
    };
    if (two_level) {  // the tails first: most neighbours live there, and what they yield tightens T for the other lists
        sc_launch_ivf_slot_thr(d_sq, d_qn, d_se, thr, nslots, d_st, d_stf, s);
        sc_prof_begin(rt, SC_PROF_SCAN, &e0, &e1);
        sc_launch_ivf_coarse(ix->Xc8, ix->xcs, ld8, b + o_qc, b + o_items + b_first * sizeof(Item), (int)items[1].size(), d_stf, d_st, d_qn, d_qs, d_sq, d_qb, d_qd, survB,
                             cnt, IVFC_CAPB, b + o_hits, hit_bytes, s, nullptr, kmetric);
        sc_prof_end(rt, SC_PROF_SCAN, e0, e1);
        st = tighten();
        if (st) return st;
        b_first += items[1].size();
        b_count = items[2].size();
    }
    if (b_count > 0) {
        sc_launch_ivf_slot_thr(d_sq, d_qn, d_se, thr, nslots, d_st, d_stf, s);
        sc_prof_begin(rt, SC_PROF_SCAN, &e0, &e1);
        sc_launch_ivf_coarse(ix->Xc8, ix->xcs, ld8, b + o_qc, b + o_items + b_first * sizeof(Item), (int)b_count, d_stf, d_st, d_qn, d_qs, d_sq, d_qb, d_qd, survB, cnt,
                             IVFC_CAPB, b + o_hits, hit_bytes, s, nullptr, kmetric);
        sc_prof_end(rt, SC_PROF_SCAN, e0, e1);
    }
    if (two_level && b_count > 0) {  // ... and once more over everything phase B kept: a query between two clusters finds its neighbours in the other lists
        st = tighten();
        if (st) return st;
    }
    // 6. refine: every row whose lower bound is within T, re-scored exactly; exact top-k
    sc_prof_begin(rt, SC_PROF_MERGE, &e0, &e1);
    sc_launch_ivf_candidates(metric, survA, cntA_d, best, kpa, survB, cnt, IVFC_CAP, thr, cand2, ncand, flags, Q, g_ivf_refine_cap < WCAP ? g_ivf_refine_cap : WCAP, s, IVFC_CAPB);
    sc_launch_scan_rerank_keys(metric, ix->X, ix->xnorm, ld, ix->qpad, ix->qnorm, cand2, ncand, WCAP, ix->perm, ekeys2, Q, s);
    sc_launch_ivf_refine_finalize(metric, ekeysA, kpa, ekeys2, ncand, flags, k, ix->row_base, out_dist, out_rows, Q, s);
    sc_prof_end(rt, SC_PROF_MERGE, e0, e1);
    SC_HIP(hipGetLastError());
    static const bool trace_c = getenv("SC_IVF_TRACE") != nullptr;  // tuning aid
    std::vector<int> hflags(Q);
    SC_HIP(hipMemcpyAsync(hflags.data(), flags, (size_t)Q * 4, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    std::vector<int> redo;
    for (int i = 0; i < Q; ++i)
        if (hflags[i]) redo.push_back(i);
    const int R = (int)redo.size();
    ix->last_ivfc_uncertified = R;
    ix->last_uncertified = R;
    if (trace_c) {
        std::vector<int> hnc(Q);
        std::vector<unsigned> hcb(Q);
        std::vector<float> hthr(Q);
        SC_HIP(hipMemcpy(hnc.data(), ncand, (size_t)Q * 4, hipMemcpyDeviceToHost));
        SC_HIP(hipMemcpy(hcb.data(), cnt, (size_t)Q * 4, hipMemcpyDeviceToHost));
        SC_HIP(hipMemcpy(hthr.data(), thr, (size_t)Q * 4, hipMemcpyDeviceToHost));
        int novf = 0, ninf = 0, nc_max = 0, ja_max = 0;
        double nc_sum = 0, cb_sum = 0, ca_sum = 0, ja_sum = 0;
        for (int i = 0; i < Q; ++i) {
            novf += hcb[i] > (unsigned)IVFC_CAPB || cntA[(size_t)i] > (unsigned)IVFC_CAP;
            ninf += !(hthr[i] < 1e30f);
            nc_max = std::max(nc_max, hnc[i]);
            nc_sum += hnc[i];
            cb_sum += hcb[i];
            ca_sum += cntA[(size_t)i];
            ja_max = std::max(ja_max, ja[(size_t)i]);
            ja_sum += ja[(size_t)i];
        }
        fprintf(stderr, "[ivf coarse] Q %d: to the exact probe %d (survivor overflow %d, bound +inf %d); re-scored per query %d + avg %.1f max %d; phase A rows avg %.0f, "
                        "phase B survivors avg %.1f; phase A lists per query avg %.2f max %d; items %zu + %zu%s + %zu, slots %d; streamed %.1f GB int8\n",
                Q, R, novf, ninf, kpa, nc_sum / Q, nc_max, ca_sum / Q, cb_sum / Q, ja_sum / Q, ja_max, items[0].size(), items[1].size(), two_level ? " (own level)" : "",
                items[2].size(), nslots,
                (double)streamed_rows * ld8 / 1e9);
    }
    if (ix->search_mode == 0 && Q >= 32 && R * 4 > Q) ix->ivfc_off = true;  // this index does not quantise well enough: later batches probe exactly
    ix->last_probed_lists = nprobe;
    ix->last_unique_rows = unique_rows;
    ix->last_streamed_rows = streamed_rows;
    ix->last_groups = (int)(nslots / 64);
    if (R > 0) {  // probed again exactly: the sub-batch gets its own staging (queries + results)
        const size_t qb = ((size_t)R * ix->dim * 4 + 255) & ~(size_t)255, db = ((size_t)R * k * 4 + 255) & ~(size_t)255, rb = ((size_t)R * k * 8 + 255) & ~(size_t)255;
        st = sc_grow(ix, &ix->fb, &ix->fb_cap, qb + db + rb + (size_t)R * 4);
        if (st) return st;
        float* fq = (float*)ix->fb;
        float* fd = (float*)((char*)ix->fb + qb);
        int64_t* frw = (int64_t*)((char*)ix->fb + qb + db);
        int32_t* fidx = (int32_t*)((char*)ix->fb + qb + db + rb);
        SC_HIP(hipMemcpyAsync(fidx, redo.data(), (size_t)R * 4, hipMemcpyHostToDevice, s));
        sc_launch_copy_rows_indexed(q_dev, fq, fidx, R, (size_t)ix->dim * 4, false, s);
        const int saved_mode = ix->search_mode;
        ix->search_mode = 4;
        if (R >= 2) st = sc_ivf_search_listmajor_locked(ix, fq, R, k, nprobe, fd, frw);
        else st = sc_ivf_search_locked(ix, fq, R, k, nprobe, fd, frw);
        ix->search_mode = saved_mode;
        if (st) return st;
        sc_launch_copy_rows_indexed(fd, out_dist, fidx, R, (size_t)k * 4, true, s);
        sc_launch_copy_rows_indexed(frw, out_rows, fidx, R, (size_t)k * 8, true, s);
        SC_HIP(hipStreamSynchronize(s));
        ix->last_ivfc_uncertified = R;
        ix->last_uncertified = R;
    }
    ix->last_path = 5;
    return SC_OK;
}

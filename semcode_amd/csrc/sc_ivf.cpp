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
        const int saved_mode = qz->search_mode;
        qz->search_mode = 2;  // thousands of queries against few centroids: the MFMA path, certified exact
        st = sc_search_flat_locked(qz, q_dev_tight + r0 * ix->dim, m, 1, dd, dr);
        qz->search_mode = saved_mode;
        if (st) return st;
        SC_HIP(hipMemcpyAsync(host.data(), dr, (size_t)m * 8, hipMemcpyDeviceToHost, s));
        SC_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < m; ++i) out[(size_t)(r0 + i)] = (int32_t)host[(size_t)i];
    }
    return SC_OK;
}

sc_status sc_ivf_untrain_locked(sc_index* ix) {
    if (!ix->perm) {
        ix->trained = false;
        return SC_OK;
    }
    hipStream_t s = ix->rt->stream;
    // Xo[row] = X[inv[row]]
    float *nx = nullptr, *nn = nullptr;
    hipError_t e = hipMalloc((void**)&nx, (size_t)ix->capacity * ix->ld * sizeof(float));
    if (e != hipSuccess) return sc_fail(SC_ERR_NOMEM, "untrain: hipMalloc failed: %s", hipGetErrorString(e));
    e = hipMalloc((void**)&nn, (size_t)ix->capacity * sizeof(float));
    if (e != hipSuccess) {
        hipFree(nx);
        return sc_fail(SC_ERR_NOMEM, "untrain: hipMalloc failed: %s", hipGetErrorString(e));
    }
    sc_launch_permute_rows(ix->X, ix->xnorm, ix->inv, ix->n, ix->ld, nx, nn, s);
    SC_HIP(hipStreamSynchronize(s));
    hipFree(ix->X);
    hipFree(ix->xnorm);
    ix->X = nx;
    ix->xnorm = nn;
    hipFree(ix->perm);
    hipFree(ix->inv);
    hipFree(ix->list_off);
    ix->perm = ix->inv = nullptr;
    ix->list_off = nullptr;
    ix->inv_h.clear();
    ix->list_off_h.clear();
    ix->trained = false;
    ix->shadow_rows = 0;
    return SC_OK;
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
    struct Dev {
        void* p = nullptr;
        ~Dev() { hipFree(p); }
    } d_srows, d_sample, d_cinit, d_members, d_moff, d_cnew, d_tight;
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
        qz->shadow_rows = 0;
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
        SC_HIP(hipStreamSynchronize(s));
        set_centroids((const float*)d_cnew.p);
    }

    // ---- assign every row, build list-major order
    const float* all_tight = ix->X;
    if (ld != dim) {
        SC_HIP(hipMalloc(&d_tight.p, (size_t)n * dim * 4));
        sc_launch_gather_rows(ix->X, ld, 0, n, dim, (float*)d_tight.p, s);
        all_tight = (const float*)d_tight.p;
    }
    st = assign_rows(ix, all_tight, n, assign);
    if (st) return st;
    std::vector<int64_t> off((size_t)nlist + 1, 0);
    for (int64_t i = 0; i < n; ++i) off[(size_t)assign[(size_t)i] + 1]++;
    for (int c = 0; c < nlist; ++c) off[(size_t)c + 1] += off[(size_t)c];
    std::vector<uint32_t> perm((size_t)n), inv((size_t)n);
    {
        std::vector<int64_t> cur(off.begin(), off.end() - 1);
        for (int64_t i = 0; i < n; ++i) {
            const int64_t pos = cur[(size_t)assign[(size_t)i]]++;
            perm[(size_t)pos] = (uint32_t)i;
            inv[(size_t)i] = (uint32_t)pos;
        }
    }
    float *nx = nullptr, *nn = nullptr;
    hipError_t e = hipMalloc((void**)&nx, (size_t)ix->capacity * ld * sizeof(float));
    if (e != hipSuccess) return sc_fail(SC_ERR_NOMEM, "train: hipMalloc list-major corpus failed: %s", hipGetErrorString(e));
    e = hipMalloc((void**)&nn, (size_t)ix->capacity * sizeof(float));
    if (e != hipSuccess) {
        hipFree(nx);
        return sc_fail(SC_ERR_NOMEM, "train: hipMalloc failed: %s", hipGetErrorString(e));
    }
    SC_HIP(hipMalloc((void**)&ix->perm, (size_t)n * 4));
    SC_HIP(hipMalloc((void**)&ix->inv, (size_t)n * 4));
    SC_HIP(hipMalloc((void**)&ix->list_off, (size_t)(nlist + 1) * 8));
    SC_HIP(hipMemcpyAsync(ix->perm, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    SC_HIP(hipMemcpyAsync(ix->inv, inv.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    SC_HIP(hipMemcpyAsync(ix->list_off, off.data(), (size_t)(nlist + 1) * 8, hipMemcpyHostToDevice, s));
    sc_launch_permute_rows(ix->X, ix->xnorm, ix->perm, n, ld, nx, nn, s);
    SC_HIP(hipGetLastError());
    SC_HIP(hipStreamSynchronize(s));
    hipFree(ix->X);
    hipFree(ix->xnorm);
    ix->X = nx;
    ix->xnorm = nn;
    ix->inv_h.swap(inv);
    ix->list_off_h.swap(off);
    ix->nlist_trained = nlist;
    ix->shadow_rows = 0;
    ix->trained = true;
    return SC_OK;
}

bool sc_ivf_applicable(const sc_index* ix, int Q, int nprobe) {
    if (ix->kind != SC_INDEX_IVF_FLAT || !ix->trained || !ix->quant) return false;
    if (nprobe < 1 || nprobe > 512 || ix->search_mode == 1 || ix->search_mode == 2) return false;
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

// Centroids (tight [nlist, dim]) and list sizes back to the host (tests, persistence).
extern "C" sc_status sc_index_ivf_info(sc_index* ix, int32_t* nlist, float* centroids, int64_t* list_sizes) {
    if (!ix) return sc_fail(SC_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> g(ix->mu);
    if (!ix->trained || !ix->quant) {
        if (nlist) *nlist = 0;
        return SC_OK;
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

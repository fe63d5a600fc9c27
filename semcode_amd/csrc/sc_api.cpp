// sc_api.cpp -- host side of the C ABI declared in include/semcode_hip.h (runtime + vector index).
//
// Mirrors (reference): MilvusVectorStore's use of pymilvus -- connect / create collection + index /
// upsert / search -- src/semcode/storage/milvus_store.py:39-148.  Error behaviour: every failure is
// a negative sc_status plus a thread-local message, so the Python seam can raise ordinary
// exceptions that IndexerService / SemanticSearchPipeline already catch (indexer.py:57-63,
// pipeline.py:95-110).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <chrono>
#include <vector>

#include "sc_internal.h"

static thread_local std::string g_err;

sc_status sc_fail(sc_status code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

extern "C" const char* sc_version(void) { return "semcode_hip 0.1 (gfx950)"; }

extern "C" sc_status sc_last_error(char* buf, size_t n) {
    if (!buf || n == 0) return SC_ERR_INVALID;
    snprintf(buf, n, "%s", g_err.c_str());
    return SC_OK;
}

// ------------------------------------------------------------------ runtime

extern "C" sc_status sc_runtime_create(const sc_runtime_cfg* cfg, sc_runtime** out) {
    if (!out) return sc_fail(SC_ERR_INVALID, "sc_runtime_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return sc_fail(SC_ERR_HIP, "sc_runtime_create: no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    const int dev = cfg ? cfg->device : 0;
    if (dev < 0 || dev >= ndev) return sc_fail(SC_ERR_INVALID, "sc_runtime_create: device %d out of range [0,%d)", dev, ndev);
    SC_HIP(hipSetDevice(dev));
    sc_runtime* rt = new (std::nothrow) sc_runtime();
    if (!rt) return sc_fail(SC_ERR_NOMEM, "sc_runtime_create: out of host memory");
    rt->device = dev;
    if (cfg && cfg->stream) {
        rt->stream = (hipStream_t)cfg->stream;
        rt->own_stream = false;
    } else {
        e = hipStreamCreateWithFlags(&rt->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete rt;
            return sc_fail(SC_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        rt->own_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
        snprintf(rt->name, sizeof rt->name, "%s (%s)", prop.name, prop.gcnArchName);
        rt->cus = prop.multiProcessorCount;
        rt->hbm = (int64_t)prop.totalGlobalMem;
    }
    *out = rt;
    return SC_OK;
}

static void prof_clear(sc_runtime* rt) {
    for (int c = 0; c < SC_PROF_CLASSES; ++c) {
        for (auto& p : rt->prof[c]) {
            hipEventDestroy(p.first);
            hipEventDestroy(p.second);
        }
        rt->prof[c].clear();
    }
}

void sc_runtime_retain(sc_runtime* rt) { rt->refs.fetch_add(1, std::memory_order_relaxed); }

void sc_runtime_release(sc_runtime* rt) {
    if (rt->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    hipSetDevice(rt->device);
    hipStreamSynchronize(rt->stream);
    prof_clear(rt);
    if (rt->own_stream) hipStreamDestroy(rt->stream);
    delete rt;
}

// Drops the creator's reference.  Indexes / encoders still alive keep the runtime (device binding + stream) alive until
// they are destroyed themselves: their destroy paths synchronise on rt->stream, which must not be a freed handle.
extern "C" sc_status sc_runtime_destroy(sc_runtime* rt) {
    if (!rt) return SC_OK;
    sc_runtime_release(rt);
    return SC_OK;
}

extern "C" sc_status sc_runtime_set_stream(sc_runtime* rt, void* stream) {
    if (!rt) return sc_fail(SC_ERR_INVALID, "runtime is NULL");
    std::lock_guard<std::mutex> g(rt->mu);
    if (rt->own_stream) {
        hipStreamSynchronize(rt->stream);
        hipStreamDestroy(rt->stream);
        rt->own_stream = false;
    }
    rt->stream = (hipStream_t)stream;
    return SC_OK;
}

extern "C" sc_status sc_runtime_synchronize(sc_runtime* rt) {
    if (!rt) return sc_fail(SC_ERR_INVALID, "runtime is NULL");
    SC_HIP(hipSetDevice(rt->device));
    SC_HIP(hipStreamSynchronize(rt->stream));
    return SC_OK;
}

extern "C" sc_status sc_runtime_device_info(sc_runtime* rt, char* name, size_t n, int32_t* cus, int64_t* hbm_bytes) {
    if (!rt) return sc_fail(SC_ERR_INVALID, "runtime is NULL");
    if (name && n) snprintf(name, n, "%s", rt->name);
    if (cus) *cus = rt->cus;
    if (hbm_bytes) *hbm_bytes = rt->hbm;
    return SC_OK;
}

extern "C" sc_status sc_runtime_set_profiling(sc_runtime* rt, int32_t enabled) {
    if (!rt) return sc_fail(SC_ERR_INVALID, "runtime is NULL");
    rt->profiling = enabled < 0 ? 0 : enabled;
    return SC_OK;
}

void sc_prof_begin(sc_runtime* rt, int which, hipEvent_t* a, hipEvent_t* b) {
    *a = *b = nullptr;
    if (!rt->profiling || rt->prof[which].size() >= 8192) return;
    // the encoder launches ~60 kernels of these classes per step: an event pair around each costs 1.6 % of the step, so they
    // are sampled (bench.py passes a stride co-prime with the 4 GEMM shapes per layer, so every shape is sampled equally)
    if (which >= SC_PROF_GEMM && rt->profiling > 1 && (rt->prof_seen[which]++ % (unsigned)rt->profiling) != 0) return;
    if (hipEventCreate(a) != hipSuccess) { *a = nullptr; return; }
    if (hipEventCreate(b) != hipSuccess) { hipEventDestroy(*a); *a = *b = nullptr; return; }
    hipEventRecord(*a, rt->stream);
}
void sc_prof_end(sc_runtime* rt, int which, hipEvent_t a, hipEvent_t b) {
    if (!a) return;
    hipEventRecord(b, rt->stream);
    std::lock_guard<std::mutex> g(rt->mu);
    rt->prof[which].push_back({a, b});
}

extern "C" sc_status sc_runtime_profile_read(sc_runtime* rt, int32_t which, double* total_ms, int64_t* launches) {
    if (!rt || which < 0 || which >= SC_PROF_CLASSES) return sc_fail(SC_ERR_INVALID, "bad profile class");
    SC_HIP(hipSetDevice(rt->device));
    SC_HIP(hipStreamSynchronize(rt->stream));
    double t = 0;
    std::lock_guard<std::mutex> g(rt->mu);
    for (auto& p : rt->prof[which]) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) t += ms;
    }
    if (total_ms) *total_ms = t;
    if (launches) *launches = (int64_t)rt->prof[which].size();
    return SC_OK;
}

extern "C" sc_status sc_runtime_profile_reset(sc_runtime* rt) {
    if (!rt) return sc_fail(SC_ERR_INVALID, "runtime is NULL");
    SC_HIP(hipSetDevice(rt->device));
    SC_HIP(hipStreamSynchronize(rt->stream));
    std::lock_guard<std::mutex> g(rt->mu);
    prof_clear(rt);
    for (unsigned& c : rt->prof_seen) c = 0;
    return SC_OK;
}

extern "C" sc_status sc_synth_fill_dev(sc_runtime* rt, float* out, int64_t rows, int32_t dim, int32_t ld, uint64_t seed,
                                       int64_t first_row) {
    if (!rt || !out) return sc_fail(SC_ERR_INVALID, "sc_synth_fill_dev: NULL argument");
    if (rows < 0 || dim <= 0 || ld < dim || (ld & 3)) return sc_fail(SC_ERR_INVALID, "sc_synth_fill_dev: need rows>=0, 0<dim<=ld, ld%%4==0");
    SC_HIP(hipSetDevice(rt->device));
    sc_launch_synth_fill(out, rows, dim, ld, seed, first_row, nullptr, rt->stream);
    SC_HIP(hipGetLastError());
    return SC_OK;
}

// ------------------------------------------------------------------ index

static inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

extern "C" sc_status sc_index_create(sc_runtime* rt, int32_t dim, sc_metric metric, sc_index_kind kind, int32_t nlist,
                                     int64_t row_base, sc_index** out) {
    if (!rt || !out) return sc_fail(SC_ERR_INVALID, "sc_index_create: NULL argument");
    *out = nullptr;
    if (dim <= 0 || dim > 65536) return sc_fail(SC_ERR_INVALID, "sc_index_create: dim %d out of range", dim);
    if (metric != SC_METRIC_IP && metric != SC_METRIC_L2 && metric != SC_METRIC_COSINE)
        return sc_fail(SC_ERR_INVALID, "sc_index_create: unknown metric %d", (int)metric);
    if (kind != SC_INDEX_FLAT && kind != SC_INDEX_IVF_FLAT) return sc_fail(SC_ERR_INVALID, "sc_index_create: unknown kind %d", (int)kind);
    if (kind == SC_INDEX_IVF_FLAT && nlist < 1) return sc_fail(SC_ERR_INVALID, "sc_index_create: IVF_FLAT needs nlist >= 1");
    sc_index* ix = new (std::nothrow) sc_index();
    if (!ix) return sc_fail(SC_ERR_NOMEM, "out of host memory");
    ix->rt = rt;
    sc_runtime_retain(rt);
    ix->dim = dim;
    ix->ld = round_up(dim, SC_LD_ALIGN);
    ix->metric = metric;
    ix->kind = kind;
    ix->nlist = nlist;
    ix->row_base = row_base;
    *out = ix;
    return SC_OK;
}

extern "C" sc_status sc_index_destroy(sc_index* ix) {
    if (!ix) return SC_OK;
    hipSetDevice(ix->rt->device);
    hipStreamSynchronize(ix->rt->stream);
    hipFree(ix->X);
    hipFree(ix->xnorm);
    hipFree(ix->stage);
    hipFree(ix->qpad);
    hipFree(ix->qnorm);
    hipFree(ix->partial);
    hipFree(ix->io);
    hipFree(ix->Xb);
    hipFree(ix->xnorm_max);
    hipFree(ix->Xq);
    hipFree(ix->xscale);
    hipFree(ix->xnorm_max8);
    hipFree(ix->bscratch);
    hipFree(ix->fb);
    hipFree(ix->fb2);
    hipFree(ix->tailbuf);
    hipFree(ix->perm);
    hipFree(ix->list_off);
    hipFree(ix->ivf_scratch);
    hipFree(ix->Xc8);
    hipFree(ix->xcs);
    hipFree(ix->list_stats);
    hipFree(ix->ivfc_scratch);
    if (ix->quant) sc_index_destroy(ix->quant);
    sc_runtime* rt = ix->rt;
    delete ix;
    sc_runtime_release(rt);
    return SC_OK;
}

extern "C" sc_status sc_index_info(sc_index* ix, int64_t* rows, int32_t* dim, int32_t* ld) {
    if (!ix) return sc_fail(SC_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> g(ix->mu);
    if (rows) *rows = ix->n;
    if (dim) *dim = ix->dim;
    if (ld) *ld = ix->ld;
    return SC_OK;
}

// grow a device scratch buffer (contents not preserved)
sc_status sc_grow(sc_index* ix, void** p, size_t* cap, size_t need) {
    if (need <= *cap) return SC_OK;
    SC_HIP(hipStreamSynchronize(ix->rt->stream));
    if (*p) hipFree(*p);
    *p = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess) return sc_fail(SC_ERR_NOMEM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
    *cap = need;
    return SC_OK;
}

// make room for `rows` rows, preserving the first ix->n
static sc_status ensure_rows(sc_index* ix, int64_t rows, bool exact) {
    if (rows <= ix->capacity) return SC_OK;
    int64_t cap = rows;
    if (!exact) cap = std::max<int64_t>(rows, std::max<int64_t>(ix->capacity * 2, 4096));
    float *nx = nullptr, *nn = nullptr;
    hipError_t e = hipMalloc((void**)&nx, (size_t)cap * ix->ld * sizeof(float));
    if (e != hipSuccess) return sc_fail(SC_ERR_NOMEM, "hipMalloc corpus (%lld rows x %d) failed: %s", (long long)cap, ix->ld, hipGetErrorString(e));
    e = hipMalloc((void**)&nn, (size_t)cap * sizeof(float));
    if (e != hipSuccess) {
        hipFree(nx);
        return sc_fail(SC_ERR_NOMEM, "hipMalloc norms failed: %s", hipGetErrorString(e));
    }
    hipStream_t s = ix->rt->stream;
    if (ix->n > 0) {
        SC_HIP(hipMemcpyAsync(nx, ix->X, (size_t)ix->n * ix->ld * sizeof(float), hipMemcpyDeviceToDevice, s));
        SC_HIP(hipMemcpyAsync(nn, ix->xnorm, (size_t)ix->n * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    SC_HIP(hipStreamSynchronize(s));
    hipFree(ix->X);
    hipFree(ix->xnorm);
    ix->X = nx;
    ix->xnorm = nn;
    ix->capacity = cap;
    return SC_OK;
}

extern "C" sc_status sc_index_reserve(sc_index* ix, int64_t rows) {
    if (!ix || rows < 0) return sc_fail(SC_ERR_INVALID, "sc_index_reserve: bad argument");
    std::lock_guard<std::mutex> g(ix->mu);
    SC_HIP(hipSetDevice(ix->rt->device));
    return ensure_rows(ix, rows, true);
}

static const int64_t STAGE_ROWS_BYTES = 64ll << 20;

extern "C" sc_status sc_index_add(sc_index* ix, const float* vecs, int64_t n) {
    if (!ix || n < 0 || (n > 0 && !vecs)) return sc_fail(SC_ERR_INVALID, "sc_index_add: bad argument");
    if (n == 0) return SC_OK;
    std::lock_guard<std::mutex> g(ix->mu);
    SC_HIP(hipSetDevice(ix->rt->device));
    if (ix->n + n > 0xFFFFFFF0ll) return sc_fail(SC_ERR_UNSUPPORTED, "sc_index_add: more than 2^32 rows per shard");
    // a trained index keeps its lists: the new rows wait behind them (position == row id) for sc_ivf_refresh_locked
    sc_status st = ensure_rows(ix, ix->n + n, false);
    if (st) return st;
    const int64_t chunk = std::max<int64_t>(1, STAGE_ROWS_BYTES / ((int64_t)ix->dim * 4));
    hipStream_t s = ix->rt->stream;
    for (int64_t off = 0; off < n; off += chunk) {
        const int64_t m = std::min(chunk, n - off);
        st = sc_grow(ix, (void**)&ix->stage, &ix->stage_cap, (size_t)m * ix->dim * 4);
        if (st) return st;
        SC_HIP(hipMemcpyAsync(ix->stage, vecs + off * ix->dim, (size_t)m * ix->dim * 4, hipMemcpyHostToDevice, s));
        sc_launch_ingest_rows((const float*)ix->stage, nullptr, ix->n + off, m, ix->dim, ix->X, ix->ld, ix->xnorm, s);
        SC_HIP(hipGetLastError());
        SC_HIP(hipStreamSynchronize(s));  // staging buffer is reused by the next chunk
    }
    ix->n += n;
    if (!ix->perm) ix->trained = false;
    return SC_OK;  // rows [shadow_rows, n) get their bf16 shadow lazily
}

extern "C" sc_status sc_index_overwrite(sc_index* ix, const float* vecs, const int64_t* rows, int64_t n) {
    if (!ix || n < 0 || (n > 0 && (!vecs || !rows))) return sc_fail(SC_ERR_INVALID, "sc_index_overwrite: bad argument");
    if (n == 0) return SC_OK;
    std::lock_guard<std::mutex> g(ix->mu);
    SC_HIP(hipSetDevice(ix->rt->device));
    for (int64_t i = 0; i < n; ++i)
        if (rows[i] < 0 || rows[i] >= ix->n) return sc_fail(SC_ERR_INVALID, "sc_index_overwrite: row %lld out of range [0,%lld)", (long long)rows[i], (long long)ix->n);
    return sc_index_put_rows_locked(ix, vecs, false, rows, n, "sc_index_overwrite");
}

// rows[i] <- vecs[i], appends allowed (see include/semcode_hip.h sc_index_put_rows).  vecs: host or device [n, dim].
sc_status sc_index_put_rows_locked(sc_index* ix, const float* vecs, bool vecs_on_device, const int64_t* rows, int64_t n, const char* who) {
    int64_t next = ix->n, min_old = INT64_MAX;
    const int64_t old_n = ix->n;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = rows[i];
        if (r == next) ++next;
        else if (r >= 0 && r < ix->n) min_old = std::min(min_old, r);
        else return sc_fail(SC_ERR_INVALID, "%s: rows[%lld] = %lld is neither an existing row [0,%lld) nor the next free row %lld", who, (long long)i,
                            (long long)r, (long long)ix->n, (long long)next);
    }
    if (next > 0xFFFFFFF0ll) return sc_fail(SC_ERR_UNSUPPORTED, "%s: more than 2^32 rows per shard", who);
    {
        std::vector<int64_t> sorted(rows, rows + n);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return sc_fail(SC_ERR_INVALID, "%s: row numbers must be distinct", who);
    }
    sc_status st = ensure_rows(ix, next, false);
    if (st) return st;
    hipStream_t s = ix->rt->stream;
    // Trained layout: a row of the lists lives at inv_h[row] (replaced in place, re-assigned to its list at the next search);
    // appended rows go behind the lists at position == row id.  The translated positions are a temporary: synchronise
    // before it goes out of scope (upserts into a trained index are not asynchronous).
    std::vector<int64_t> pos;
    const int64_t* row_ids = rows;  // the caller's row numbers (rows is redirected to stored positions below)
    if (ix->perm) {
        pos.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            pos[(size_t)i] = sc_ivf_pos(ix, rows[i]);
        }
        min_old = INT64_MAX;  // the shadow is indexed by stored position
        for (int64_t i = 0; i < n; ++i)
            if (rows[i] < ix->n) min_old = std::min(min_old, pos[(size_t)i]);
        rows = pos.data();
    }
    if (vecs_on_device) {
        st = sc_grow(ix, (void**)&ix->stage, &ix->stage_cap, (size_t)n * 8);
        if (st) return st;
        SC_HIP(hipMemcpyAsync(ix->stage, rows, (size_t)n * 8, hipMemcpyHostToDevice, s));
        sc_launch_ingest_rows(vecs, (const int64_t*)ix->stage, 0, n, ix->dim, ix->X, ix->ld, ix->xnorm, s);
        SC_HIP(hipGetLastError());
        if (ix->perm) SC_HIP(hipStreamSynchronize(s));
    } else {
        const int64_t chunk = std::max<int64_t>(1, STAGE_ROWS_BYTES / ((int64_t)ix->dim * 4 + 8));
        for (int64_t off = 0; off < n; off += chunk) {
            const int64_t m = std::min(chunk, n - off);
            const size_t vbytes = ((size_t)m * ix->dim * 4 + 15) & ~(size_t)15;
            st = sc_grow(ix, (void**)&ix->stage, &ix->stage_cap, vbytes + (size_t)m * 8);
            if (st) return st;
            int64_t* drows = (int64_t*)((char*)ix->stage + vbytes);
            SC_HIP(hipMemcpyAsync(ix->stage, vecs + off * ix->dim, (size_t)m * ix->dim * 4, hipMemcpyHostToDevice, s));
            SC_HIP(hipMemcpyAsync(drows, rows + off, (size_t)m * 8, hipMemcpyHostToDevice, s));
            sc_launch_ingest_rows((const float*)ix->stage, drows, 0, m, ix->dim, ix->X, ix->ld, ix->xnorm, s);
            SC_HIP(hipGetLastError());
            SC_HIP(hipStreamSynchronize(s));  // staging buffer is reused by the next chunk
        }
    }
    if (ix->perm && row_ids) {  // rows of the lists whose vectors changed: re-assigned at the next refresh (recorded once, after the write)
        const size_t before = ix->dirty_rows.size();
        for (int64_t i = 0; i < n; ++i)
            if (row_ids[i] < ix->ivf_rows) ix->dirty_rows.push_back(row_ids[i]);
        if (ix->dirty_rows.size() > before && ix->dirty_rows.size() > 1024) {
            std::sort(ix->dirty_rows.begin(), ix->dirty_rows.end());
            ix->dirty_rows.erase(std::unique(ix->dirty_rows.begin(), ix->dirty_rows.end()), ix->dirty_rows.end());
        }
    }
    ix->n = next;
    if (!ix->perm) ix->trained = false;
    // replaced rows: their shadow rows are stale.  Remembered by stored position and re-built alone before the next search that reads
    // the shadow (ensure_shadow / ensure_shadow8 / ivfc_ensure_shadow); appended rows get theirs lazily as before.  `rows` holds
    // stored positions here, old_n the row count before this call.
    if (min_old != INT64_MAX) {
        auto note = [&](std::vector<int64_t>& dirty, int64_t& covered) {
            if (covered == 0) return;
            for (int64_t i = 0; i < n; ++i)
                if (row_ids[i] < old_n && rows[i] < covered) dirty.push_back(rows[i]);
            if ((int64_t)dirty.size() > sc_index::SC_SHADOW_DIRTY_MAX) {  // too many single rows: the whole shadow in one pass is cheaper
                dirty.clear();
                covered = 0;
            }
        };
        note(ix->dirty_b16, ix->shadow_rows);
        note(ix->dirty_i8, ix->shadow8_rows);
        note(ix->dirty_c8, ix->shadowc_rows);
    }
    return SC_OK;
}

extern "C" sc_status sc_index_put_rows(sc_index* ix, const float* vecs, const int64_t* rows, int64_t n) {
    if (!ix || n < 0 || (n > 0 && (!vecs || !rows))) return sc_fail(SC_ERR_INVALID, "sc_index_put_rows: bad argument");
    if (n == 0) return SC_OK;
    std::lock_guard<std::mutex> g(ix->mu);
    SC_HIP(hipSetDevice(ix->rt->device));
    return sc_index_put_rows_locked(ix, vecs, false, rows, n, "sc_index_put_rows");
}

extern "C" sc_status sc_index_put_rows_dev(sc_index* ix, const float* vecs_dev, const int64_t* rows, int64_t n) {
    if (!ix || n < 0 || (n > 0 && (!vecs_dev || !rows))) return sc_fail(SC_ERR_INVALID, "sc_index_put_rows_dev: bad argument");
    if (n == 0) return SC_OK;
    std::lock_guard<std::mutex> g(ix->mu);
    SC_HIP(hipSetDevice(ix->rt->device));
    sc_status st = sc_index_put_rows_locked(ix, vecs_dev, true, rows, n, "sc_index_put_rows_dev");
    if (st) return st;
    SC_HIP(hipStreamSynchronize(ix->rt->stream));  // `rows` is the caller's (pageable) memory: do not return while its copy may be pending
    return SC_OK;
}

extern "C" sc_status sc_index_get_rows(sc_index* ix, int64_t first, int64_t n, float* out) {
    if (!ix || n < 0 || first < 0 || (n > 0 && !out)) return sc_fail(SC_ERR_INVALID, "sc_index_get_rows: bad argument");
    std::lock_guard<std::mutex> g(ix->mu);
    if (first + n > ix->n) return sc_fail(SC_ERR_INVALID, "sc_index_get_rows: [%lld,%lld) exceeds %lld rows", (long long)first, (long long)(first + n), (long long)ix->n);
    if (n == 0) return SC_OK;
    SC_HIP(hipSetDevice(ix->rt->device));
    const int64_t chunk = std::max<int64_t>(1, STAGE_ROWS_BYTES / ((int64_t)ix->dim * 4));
    hipStream_t s = ix->rt->stream;
    for (int64_t off = 0; off < n; off += chunk) {
        const int64_t m = std::min(chunk, n - off);
        sc_status st = sc_grow(ix, (void**)&ix->stage, &ix->stage_cap, (size_t)m * ix->dim * 4);
        if (st) return st;
        if (ix->perm) {  // list-major storage: fetch row ids first+off .. through the inverse permutation
            std::vector<int64_t> pos((size_t)m);
            for (int64_t i = 0; i < m; ++i) pos[(size_t)i] = sc_ivf_pos(ix, first + off + i);
            st = sc_grow(ix, &ix->ivf_scratch, &ix->ivf_scratch_cap, (size_t)m * 8);
            if (st) return st;
            SC_HIP(hipMemcpyAsync(ix->ivf_scratch, pos.data(), (size_t)m * 8, hipMemcpyHostToDevice, s));
            sc_launch_rows_to_sample(ix->X, ix->ld, ix->dim, (const int64_t*)ix->ivf_scratch, m, (float*)ix->stage, s);
            SC_HIP(hipStreamSynchronize(s));  // pos goes out of scope
        } else {
            sc_launch_gather_rows(ix->X, ix->ld, first + off, m, ix->dim, (float*)ix->stage, s);
        }
        SC_HIP(hipGetLastError());
        SC_HIP(hipMemcpyAsync(out + off * ix->dim, ix->stage, (size_t)m * ix->dim * 4, hipMemcpyDeviceToHost, s));
        SC_HIP(hipStreamSynchronize(s));
    }
    return SC_OK;
}

extern "C" sc_status sc_index_fill_synthetic(sc_index* ix, int64_t n, uint64_t seed, int64_t first_row) {
    if (!ix || n < 0) return sc_fail(SC_ERR_INVALID, "sc_index_fill_synthetic: bad argument");
    std::lock_guard<std::mutex> g(ix->mu);
    SC_HIP(hipSetDevice(ix->rt->device));
    if (n > 0xFFFFFFF0ll) return sc_fail(SC_ERR_UNSUPPORTED, "more than 2^32 rows per shard");
    sc_ivf_drop_lists_locked(ix);
    ix->n = 0;  // nothing to preserve
    sc_status st = ensure_rows(ix, n, true);
    if (st) return st;
    sc_launch_synth_fill(ix->X, n, ix->dim, ix->ld, seed, first_row, ix->xnorm, ix->rt->stream);
    SC_HIP(hipGetLastError());
    ix->n = n;
    ix->trained = false;
    ix->shadow_rows = 0;
    ix->shadow8_rows = 0;
    ix->i8_off = false;
    ix->wide_i8 = false;
    ix->i8_sticky = false;
    ix->cost_i8_first = 0.0;
    ix->collect_off8 = ix->collect_off16 = false;
    return SC_OK;
}

extern "C" sc_status sc_index_fill_synthetic_clustered(sc_index* ix, int64_t n, uint64_t seed, int64_t first_row, int32_t nclusters,
                                                       float spread) {
    if (!ix || n < 0 || nclusters < 1) return sc_fail(SC_ERR_INVALID, "sc_index_fill_synthetic_clustered: bad argument");
    std::lock_guard<std::mutex> g(ix->mu);
    SC_HIP(hipSetDevice(ix->rt->device));
    if (n > 0xFFFFFFF0ll) return sc_fail(SC_ERR_UNSUPPORTED, "more than 2^32 rows per shard");
    sc_ivf_drop_lists_locked(ix);
    ix->n = 0;
    sc_status st = ensure_rows(ix, n, true);
    if (st) return st;
    sc_launch_synth_clustered(ix->X, n, ix->dim, ix->ld, seed, first_row, nclusters, spread, ix->xnorm, ix->rt->stream);
    SC_HIP(hipGetLastError());
    ix->n = n;
    ix->trained = false;
    ix->shadow_rows = 0;
    ix->shadow8_rows = 0;
    ix->i8_off = false;
    ix->wide_i8 = false;
    ix->i8_sticky = false;
    ix->cost_i8_first = 0.0;
    ix->collect_off8 = ix->collect_off16 = false;
    return SC_OK;
}

// Free everything that can be rebuilt (bf16 shadow, search scratch): for corpora close to the HBM capacity.
extern "C" sc_status sc_index_release_scratch(sc_index* ix) {
    if (!ix) return sc_fail(SC_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> g(ix->mu);
    SC_HIP(hipSetDevice(ix->rt->device));
    SC_HIP(hipStreamSynchronize(ix->rt->stream));
    hipFree(ix->Xb); ix->Xb = nullptr; ix->xb_cap = 0; ix->shadow_rows = 0;
    hipFree(ix->Xq); ix->Xq = nullptr; ix->xq_cap = 0; ix->shadow8_rows = 0;
    hipFree(ix->xscale); ix->xscale = nullptr; ix->xscale_cap = 0;
    hipFree(ix->bscratch); ix->bscratch = nullptr; ix->bscratch_cap = 0;
    hipFree(ix->partial); ix->partial = nullptr; ix->partial_cap = 0;
    hipFree(ix->stage); ix->stage = nullptr; ix->stage_cap = 0;
    hipFree(ix->fb); ix->fb = nullptr; ix->fb_cap = 0;
    hipFree(ix->fb2); ix->fb2 = nullptr; ix->fb2_cap = 0;
    hipFree(ix->tailbuf); ix->tailbuf = nullptr; ix->tailbuf_cap = 0;
    hipFree(ix->ivf_scratch); ix->ivf_scratch = nullptr; ix->ivf_scratch_cap = 0;
    return SC_OK;
}

// q_dev: tight [Q, dim] device; outputs device.  Caller holds ix->mu.
static sc_status search_exact_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe, float* out_dist,
                                   int64_t* out_rows) {
    (void)nprobe;
    sc_runtime* rt = ix->rt;
    hipStream_t s = rt->stream;
    ScanPlan plan;
    if (!sc_scan_exact_plan(ix->ld, Q, k, rt->cus, &plan, 0, 0, ix->n))
        return sc_fail(SC_ERR_UNSUPPORTED, "search: k=%d (1..1024) / dim=%d not supported by the exact scan", k, ix->dim);
    sc_status st = sc_grow(ix, (void**)&ix->qpad, &ix->qpad_cap, (size_t)Q * ix->ld * 4);
    if (st) return st;
    st = sc_grow(ix, (void**)&ix->qnorm, &ix->qnorm_cap, (size_t)Q * 4);
    if (st) return st;
    st = sc_grow(ix, (void**)&ix->partial, &ix->partial_cap, std::max<size_t>(plan.partial_bytes, 16));
    if (st) return st;
    sc_launch_ingest_rows(q_dev, nullptr, 0, Q, ix->dim, ix->qpad, ix->ld, ix->qnorm, s);
    int lists = 0;
    if (ix->n > 0) {
        hipEvent_t e0, e1;
        sc_prof_begin(rt, SC_PROF_SCAN, &e0, &e1);
        sc_launch_scan_exact((int)ix->metric, ix->X, ix->xnorm, ix->n, ix->ld, ix->qpad, ix->qnorm, Q, k, plan, ix->partial, ix->perm, nullptr, nullptr, 0, s);
        sc_prof_end(rt, SC_PROF_SCAN, e0, e1);
        lists = plan.lists;
    }
    {
        hipEvent_t e0, e1;
        sc_prof_begin(rt, SC_PROF_MERGE, &e0, &e1);
        sc_launch_topk_merge((int)ix->metric, ix->partial, plan.groups, lists, plan.qt, Q, k, ix->row_base, out_dist, out_rows, s);
        sc_prof_end(rt, SC_PROF_MERGE, e0, e1);
    }
    SC_HIP(hipGetLastError());
    ix->last_path = 1;
    return SC_OK;
}


// the exact scan over stored rows [first, first + nrows) only: positions there equal row ids (the tail behind the lists of a trained
// IVF index)
static sc_status search_exact_range_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int64_t first, int64_t nrows, float* out_dist,
                                           int64_t* out_rows) {
    sc_runtime* rt = ix->rt;
    hipStream_t s = rt->stream;
    ScanPlan plan;
    if (!sc_scan_exact_plan(ix->ld, Q, k, rt->cus, &plan, 0, 0, nrows))
        return sc_fail(SC_ERR_UNSUPPORTED, "search: k=%d (1..1024) / dim=%d not supported by the exact scan", k, ix->dim);
    sc_status st = sc_grow(ix, (void**)&ix->qpad, &ix->qpad_cap, (size_t)Q * ix->ld * 4);
    if (st) return st;
    st = sc_grow(ix, (void**)&ix->qnorm, &ix->qnorm_cap, (size_t)Q * 4);
    if (st) return st;
    st = sc_grow(ix, (void**)&ix->partial, &ix->partial_cap, std::max<size_t>(plan.partial_bytes, 16));
    if (st) return st;
    sc_launch_ingest_rows(q_dev, nullptr, 0, Q, ix->dim, ix->qpad, ix->ld, ix->qnorm, s);
    hipEvent_t e0, e1;
    sc_prof_begin(rt, SC_PROF_SCAN, &e0, &e1);
    sc_launch_scan_exact((int)ix->metric, ix->X + (size_t)first * ix->ld, ix->xnorm + first, nrows, ix->ld, ix->qpad, ix->qnorm, Q, k, plan, ix->partial, nullptr, nullptr,
                         nullptr, 0, s);
    sc_prof_end(rt, SC_PROF_SCAN, e0, e1);
    sc_prof_begin(rt, SC_PROF_MERGE, &e0, &e1);
    sc_launch_topk_merge((int)ix->metric, ix->partial, plan.groups, plan.lists, plan.qt, Q, k, ix->row_base + first, out_dist, out_rows, s);
    sc_prof_end(rt, SC_PROF_MERGE, e0, e1);
    SC_HIP(hipGetLastError());
    return SC_OK;
}

// ---- batched path (scan_batched.hip): bf16 shadow + coarse GEMM phases + exact re-rank + certified fallback

// the rows of a dirty list as runs (first, count) of stored positions, ascending; neighbours up to 32 rows apart share a run (re-building
// a clean row in between changes nothing).  Empties the list.
static std::vector<std::pair<int64_t, int64_t>> dirty_runs(std::vector<int64_t>& dirty, int64_t covered) {
    std::sort(dirty.begin(), dirty.end());
    dirty.erase(std::unique(dirty.begin(), dirty.end()), dirty.end());
    std::vector<std::pair<int64_t, int64_t>> runs;
    for (const int64_t r : dirty) {
        if (r >= covered) break;
        if (!runs.empty() && r < runs.back().first + runs.back().second + 32) runs.back().second = r + 1 - runs.back().first;
        else runs.emplace_back(r, 1);
    }
    dirty.clear();
    return runs;
}

static sc_status ensure_shadow(sc_index* ix) {
    hipStream_t s = ix->rt->stream;
    const int64_t rows_pad = (ix->n + 255) / 256 * 256;
    const size_t need = (size_t)rows_pad * ix->ld * 2;
    if (need > ix->xb_cap) {
        SC_HIP(hipStreamSynchronize(s));
        hipFree(ix->Xb);
        ix->Xb = nullptr;
        ix->xb_cap = 0;
        const size_t cap_rows = (size_t)((ix->capacity + 255) / 256 * 256);
        const size_t want = std::max(need, cap_rows * ix->ld * 2);
        hipError_t e = hipMalloc(&ix->Xb, want);
        if (e != hipSuccess) return sc_fail(SC_ERR_NOMEM, "hipMalloc bf16 shadow (%zu B) failed: %s", want, hipGetErrorString(e));
        ix->xb_cap = want;
        ix->shadow_rows = 0;
    }
    if (!ix->xnorm_max) {
        hipError_t e = hipMalloc((void**)&ix->xnorm_max, 16);
        if (e != hipSuccess) return sc_fail(SC_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
        ix->shadow_rows = 0;
    }
    if (ix->shadow_rows < ix->n) {
        // xnorm_max = float bits of {max |x|^2, max |x - bf16(x)|^2, max |x - bf16(x)|^2 / |x|^2} over the rows
        if (ix->shadow_rows == 0) SC_HIP(hipMemsetAsync(ix->xnorm_max, 0, 16, s));
        sc_launch_shadow(ix->X, ix->xnorm, ix->shadow_rows, ix->n - ix->shadow_rows, ix->ld, ix->Xb, ix->xnorm_max + 1, s);
        if (rows_pad > ix->n)  // the last row tile reads these rows: keep them finite
            SC_HIP(hipMemsetAsync((char*)ix->Xb + (size_t)ix->n * ix->ld * 2, 0, (size_t)(rows_pad - ix->n) * ix->ld * 2, s));
        sc_launch_norm_max(ix->xnorm, ix->n, ix->xnorm_max, s);
        ix->shadow_rows = ix->n;
        SC_HIP(hipGetLastError());
    }
    if (!ix->dirty_b16.empty()) {  // rows overwritten since: their shadow rows alone (the maxima keep accumulating)
        for (const auto& run : dirty_runs(ix->dirty_b16, ix->shadow_rows))
            sc_launch_shadow(ix->X, ix->xnorm, run.first, run.second, ix->ld, ix->Xb, ix->xnorm_max + 1, s);
        sc_launch_norm_max(ix->xnorm, ix->n, ix->xnorm_max, s);
        SC_HIP(hipGetLastError());
    }
    return SC_OK;
}

static inline int ld8_of(const sc_index* ix) { return (ix->ld + 127) / 128 * 128; }  // int8 row stride: whole 128-byte K-tiles

static sc_status ensure_shadow8(sc_index* ix) {
    hipStream_t s = ix->rt->stream;
    const int64_t rows_pad = (ix->n + 255) / 256 * 256;
    const int ld8 = ld8_of(ix);
    const size_t need = (size_t)rows_pad * ld8;
    if (need > ix->xq_cap || (size_t)rows_pad * 4 > ix->xscale_cap) {
        SC_HIP(hipStreamSynchronize(s));
        hipFree(ix->Xq);
        hipFree(ix->xscale);
        ix->Xq = nullptr;
        ix->xscale = nullptr;
        ix->xq_cap = ix->xscale_cap = 0;
        ix->shadow8_rows = 0;
        const size_t cap_rows = (size_t)((std::max(ix->capacity, ix->n) + 255) / 256 * 256);
        hipError_t e = hipMalloc(&ix->Xq, cap_rows * ld8);
        if (e != hipSuccess) return sc_fail(SC_ERR_NOMEM, "hipMalloc int8 shadow (%zu B) failed: %s", cap_rows * ld8, hipGetErrorString(e));
        ix->xq_cap = cap_rows * ld8;
        e = hipMalloc((void**)&ix->xscale, cap_rows * 4);
        if (e != hipSuccess) return sc_fail(SC_ERR_NOMEM, "hipMalloc int8 scales failed: %s", hipGetErrorString(e));
        ix->xscale_cap = cap_rows * 4;
    }
    if (!ix->xnorm_max8) {
        hipError_t e = hipMalloc((void**)&ix->xnorm_max8, 16);
        if (e != hipSuccess) return sc_fail(SC_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
        ix->shadow8_rows = 0;
    }
    if (ix->shadow8_rows < ix->n) {
        if (ix->shadow8_rows == 0) SC_HIP(hipMemsetAsync(ix->xnorm_max8, 0, 16, s));
        sc_launch_shadow8(ix->X, ix->xnorm, ix->shadow8_rows, ix->n - ix->shadow8_rows, ix->ld, ld8, ix->Xq, ix->xscale, ix->xnorm_max8 + 1, s);
        if (rows_pad > ix->n) {  // the last row tile reads these rows
            SC_HIP(hipMemsetAsync((char*)ix->Xq + (size_t)ix->n * ld8, 0, (size_t)(rows_pad - ix->n) * ld8, s));
            SC_HIP(hipMemsetAsync(ix->xscale + ix->n, 0, (size_t)(rows_pad - ix->n) * 4, s));
        }
        sc_launch_norm_max(ix->xnorm, ix->n, ix->xnorm_max8, s);
        ix->shadow8_rows = ix->n;
        SC_HIP(hipGetLastError());
    }
    if (!ix->dirty_i8.empty()) {
        for (const auto& run : dirty_runs(ix->dirty_i8, ix->shadow8_rows))
            sc_launch_shadow8(ix->X, ix->xnorm, run.first, run.second, ix->ld, ld8, ix->Xq, ix->xscale, ix->xnorm_max8 + 1, s);
        sc_launch_norm_max(ix->xnorm, ix->n, ix->xnorm_max8, s);
        SC_HIP(hipGetLastError());
    }
    return SC_OK;
}

static const int BATCH_CAP = 4096;        // survivors kept per query and phase
// first phase: every row of it survives (thresholds start at +inf), so it must stay well below BATCH_CAP; each next phase covers 4x
// more rows.  256 rows when the selection was a quadratic rank sort; with the radix select and the two-pass epilogue of the dense
// phases 2 048 saves two launches + selections per batch (10M rows: 9 -> 7 phases).  SC_PHASE0 overrides (A/B).
static int64_t phase0_rows() {
    static const int64_t v = [] {
        const char* e = getenv("SC_PHASE0");
        const long long x = e ? atoll(e) : 2048;
        return (int64_t)(x >= 256 && x <= 2048 ? (x / 256) * 256 : 2048);
    }();
    return v;
}

static int coarse_pin(const sc_index* ix);
static int64_t i8_min_rows() {  // corpora below this never build an int8 shadow (SC_I8_MINROWS: A/B)
    static const int64_t v = [] { const char* e = getenv("SC_I8_MINROWS"); return e ? (int64_t)atoll(e) : ((int64_t)1 << 20); }();
    return v;
}
static int i8_min_queries() {
    static const int v = [] { const char* e = getenv("SC_I8_MINQ"); return e ? atoi(e) : 1; }();
    return v;
}
static bool batched_applicable(const sc_index* ix, int Q, int k) {
    if (ix->search_mode == 1) return false;
    if (ix->n < 1) return false;
    // top_k beyond 64: only the int8 stage has the candidates for it (512: k <= 256); without it the exact scan answers -- one pass
    // per 16 queries, the cliff this removes where the int8 stage may run (10M x 768, 256 queries, top-100: see profiles/r3z_k100.log)
    if (k > sc_batched_kprime() / 2) {
        // (up to 128: beyond, the keys within the cut outgrow the wide set's 4 096 -- 256 queries, top-256 overflowed for most of them)
        const bool i8_ok = k <= sc_batched_kprime8() / 4 && !ix->i8_off && coarse_pin(ix) != 16 && (ix->n >= i8_min_rows() || ix->search_mode == 2);
        if (!i8_ok) return false;
        return ix->search_mode == 2 || Q >= i8_min_queries();
    }
    if (ix->search_mode == 2) return true;
    static const int64_t min_rows = [] { const char* e = getenv("SC_BATCHED_MINROWS"); return e ? (int64_t)atoll(e) : (int64_t)4096; }();  // A/B
    if (Q > 16) return ix->n >= min_rows;
    // 16 queries and fewer: the exact scan reads the f32 rows once (10M x 768: 4.8 ms); where the int8 stage may run, its narrow
    // streaming kernel reads a quarter of the bytes and the certificate still makes the result exact: 1.7 ms for one query, 1.9 for
    // 16 (scripts/q_sweep.py, profiles/r3z_q_small.log).  SC_I8_MINQ > 1 restores the exact scan below that many queries (A/B).
    return ix->n >= i8_min_rows() && !ix->i8_off && coarse_pin(ix) != 16 && Q >= i8_min_queries();
}

// The int8 stage is tried first (twice the MFMA rate, half the shadow bytes); what it cannot certify goes to the bf16 stage, and
// only what that cannot certify either to the exact scan.  SC_COARSE=bf16 | i8 pins the stage (A/B runs, tests).
static int coarse_env();
static int coarse_pin(const sc_index* ix) { return ix->coarse_mode ? ix->coarse_mode : coarse_env(); }
static int coarse_env() {
    static const int v = [] {
        const char* e = getenv("SC_COARSE");
        if (!e) return 0;
        return (e[0] == 'b' || e[0] == 'B') ? 16 : (e[0] == 'i' || e[0] == 'I') ? 8 : 0;
    }();
    return v;
}

static int g_wide_force = 0;  // sc_diag_set_option("wide_candidates", 1): the int8 stage runs its wide form wherever it can (tests)
void sc_set_wide_force(int v) { g_wide_force = v; }
static int g_tighten = 1;  // sc_diag_set_option("tighten", 0): thresholds stay the kp-th coarse keys (tests, A/B)
void sc_set_tighten(int v) { g_tighten = v; }
static int g_collect_pass = 1;  // sc_diag_set_option("collect_pass", 0): uncertified queries go straight to the next stage (tests, A/B)
void sc_set_collect_pass(int v) { g_collect_pass = v; }

// The collect pass (scan_batched.hip, "the collect pass"): the sub-batch `fq` [R][dim] of queries a stage could not certify, with
// that stage's results in fd / fr [R][k] (fd's k-th column bounds the k-th score).  Resolved queries get their final results
// written into fd / fr; `left` receives the sub-batch positions of those that still need the next stage (more than BATCH_CAP rows
// within the bound, or no bound).  Uses the same scratch as the stage that called it (which is done with it).
static sc_status search_collect_locked(sc_index* ix, const float* fq, int R, int k, float* fd, int64_t* fr, bool i8, std::vector<int>& left) {
    sc_runtime* rt = ix->rt;
    hipStream_t s = rt->stream;
    const int metric = (int)ix->metric, ld = ix->ld, ld8 = ld8_of(ix);
    const int Qpad = (i8 || R > 64) ? (R + 255) / 256 * 256 : 128;
    static_assert(BATCH_CAP == 4096, "the refine kernels' candidate stride (sc_ivf_widen_cap) is the survivor cap");
    if (sc_ivf_widen_cap() != BATCH_CAP) return sc_fail(SC_ERR_STATE, "collect pass: candidate stride mismatch");
    sc_status st = sc_grow(ix, (void**)&ix->qpad, &ix->qpad_cap, (size_t)R * ld * 4);
    if (st) return st;
    st = sc_grow(ix, (void**)&ix->qnorm, &ix->qnorm_cap, (size_t)R * 4);
    if (st) return st;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_qres = carve((size_t)R * 4), o_amax = carve(16);
    const size_t o_qb = carve(i8 ? (size_t)Qpad * ld8 : (size_t)Qpad * ld * 2), o_qs = carve((size_t)Qpad * 4), o_thr = carve((size_t)Qpad * 4),
                 o_tf = carve((size_t)Qpad * 4), o_cnt = carve((size_t)R * 4), o_ovf = carve((size_t)R * 4), o_flag = carve((size_t)R * 4), o_nc = carve((size_t)R * 4),
                 o_surv = carve((size_t)R * BATCH_CAP * 8), o_ek = carve((size_t)R * BATCH_CAP * 8);
    const size_t hit_bytes = (i8 && R <= 64) ? (size_t)2048 * (4 + 1024 * 16) + 256 : 0;
    const size_t o_hits = carve(hit_bytes ? hit_bytes : 16);
    st = sc_grow(ix, &ix->bscratch, &ix->bscratch_cap, off);
    if (st) return st;
    char* b = (char*)ix->bscratch;
    void* Qb = b + o_qb;
    float *qres = (float*)(b + o_qres), *qscale = (float*)(b + o_qs), *thr = (float*)(b + o_thr), *tf = (float*)(b + o_tf);
    unsigned* cnt = (unsigned*)(b + o_cnt);
    int *ovf = (int*)(b + o_ovf), *flags = (int*)(b + o_flag), *ncand = (int*)(b + o_nc);
    uint64_t *surv = (uint64_t*)(b + o_surv), *ekeys = (uint64_t*)(b + o_ek);
    sc_launch_ingest_rows(fq, nullptr, 0, R, ix->dim, ix->qpad, ld, ix->qnorm, s);
    if (i8) sc_launch_query_i8(ix->qpad, R, Qpad, ld, ld8, Qb, qscale, qres, (unsigned*)(b + o_amax), s);
    else sc_launch_query_bf16(ix->qpad, R, Qpad, ld, Qb, qres, s);
    sc_launch_scan_batched_init(thr, tf, Qpad, nullptr, cnt, ovf, R, 0, s);
    sc_launch_scan_collect_bound(metric, fd, k, ix->qnorm, qres, i8 ? ix->xnorm_max8 : ix->xnorm_max, ld, thr, tf, flags, R, s);
    hipEvent_t e0, e1;
    sc_prof_begin(rt, SC_PROF_SCAN, &e0, &e1);
    if (i8) sc_launch_scan_coarse(metric, ix->Xq, ix->xnorm, 0, ix->n, ld8, Qb, ix->qnorm, R, Qpad, thr, tf, surv, cnt, BATCH_CAP, s, true, ix->xscale, qscale, false, hit_bytes ? (void*)(b + o_hits) : nullptr, hit_bytes);
    else sc_launch_scan_coarse(metric, ix->Xb, ix->xnorm, 0, ix->n, ld, Qb, ix->qnorm, R, Qpad, thr, tf, surv, cnt, BATCH_CAP, s, false, nullptr, nullptr, false);
    sc_prof_end(rt, SC_PROF_SCAN, e0, e1);
    sc_prof_begin(rt, SC_PROF_MERGE, &e0, &e1);
    sc_launch_scan_collect_counts(cnt, BATCH_CAP, ncand, flags, R, s);
    sc_launch_scan_rerank_keys(metric, ix->X, ix->xnorm, ld, ix->qpad, ix->qnorm, surv, ncand, BATCH_CAP, ix->perm, ekeys, R, s);
    sc_launch_refine_finalize(metric, ekeys, ncand, flags, k, ix->row_base, fd, fr, R, s);
    sc_prof_end(rt, SC_PROF_MERGE, e0, e1);
    SC_HIP(hipGetLastError());
    std::vector<int> hflags(R);
    SC_HIP(hipMemcpyAsync(hflags.data(), flags, (size_t)R * 4, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    left.clear();
    for (int j = 0; j < R; ++j)
        if (hflags[j]) left.push_back(j);
    return SC_OK;
}

static sc_status search_batched_stage_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, float* out_dist, int64_t* out_rows, bool i8, int depth,
                                             int Q_top) {
    sc_runtime* rt = ix->rt;
    hipStream_t s = rt->stream;
    const int metric = (int)ix->metric, ld = ix->ld, KP = i8 ? sc_batched_kprime8() : sc_batched_kprime();
    const int ld8 = ld8_of(ix);
    // 256-wide query tiles for batches above 64 queries (always for the int8 stage); 65 .. 128 queries used to take the 128-query tiles:
    // 1M x 768, 65 queries 2.02 ms there against 0.86 ms for 256 queries on the 256-wide tiles (profiles/r3z_q_rows.log)
    const int Qpad = (i8 || Q > 64) ? (Q + 255) / 256 * 256 : 128;
    // the wide candidate set (scan_batched.hip): on corpora whose certificate fails at kp candidates the int8 stage keeps every key
    // within its exact-score cut -- needs the cuts (tightening: 2 k <= 128, a corpus beyond 2^17 rows) and 64 KiB of keys per query
    static const bool tighten_env = [] { const char* e = getenv("SC_TIGHTEN"); return !(e && e[0] == '0'); }();  // A/B
    // (the cut needs the k-th exact score among re-scored candidates: the 128 best for k <= 64, all 512 of the int8 stage beyond)
    const bool tighten = tighten_env && g_tighten && (2 * k <= 128 || (i8 && 2 * k <= KP));
    const int TK = 2 * k <= 128 ? 128 : KP;
    const int WB = BATCH_CAP;  // capacity of the wide set
    // (top_k beyond 64 goes straight to the wide form: the 512-candidate certificate is hopeless there -- 256 queries, top-100 over 10M x 768:
    // 5.9 ms, against 211 ms through the exact scan, profiles/r3z_k100.log)
    const bool big_k = k > sc_batched_kprime() / 2;
    const bool wide = i8 && tighten && (ix->wide_i8 || g_wide_force || big_k) && depth == 0 && Q <= 16384 && ix->n > ((int64_t)1 << 18);
    const int KB = wide ? WB : KP;  // row stride of `best`
    sc_status st = i8 ? ensure_shadow8(ix) : ensure_shadow(ix);
    if (st) return st;
    st = sc_grow(ix, (void**)&ix->qpad, &ix->qpad_cap, (size_t)Q * ld * 4);
    if (st) return st;
    st = sc_grow(ix, (void**)&ix->qnorm, &ix->qnorm_cap, (size_t)Q * 4);
    if (st) return st;
    // scratch layout
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_qres = carve((size_t)Q * 4), o_amax = carve(16);
    const size_t o_qb = carve(i8 ? (size_t)Qpad * ld8 : (size_t)Qpad * ld * 2), o_qs = carve((size_t)Qpad * 4), o_thr = carve((size_t)Qpad * 4),
                 o_tf = carve((size_t)Qpad * 4), o_cnt = carve((size_t)Q * 4), o_ovf = carve((size_t)Q * 4), o_flag = carve((size_t)Q * 4),
                 o_best = carve((size_t)Q * KB * 8), o_ek = carve(i8 ? (size_t)Q * KB * 8 : 16), o_surv = carve((size_t)Q * BATCH_CAP * 8),
                 o_nbest = carve((size_t)Q * 4), o_wcand = carve((size_t)Q * KB * 8), o_wnc = carve((size_t)Q * 4),
                 o_b128 = carve((size_t)Q * 512 * 8), o_e128 = carve((size_t)Q * 512 * 8), o_cut = carve((size_t)Qpad * 4), o_cnt2 = carve((size_t)Q * 4),
                 o_thrT = carve((size_t)Qpad * 4), o_tfT = carve((size_t)Qpad * 4);
    // per-wave hit lists of the narrow int8 kernel (batches of <= 64 queries): 2048 lists x 1024 entries of 16 B
    const size_t hit_bytes = (i8 && Q <= 64) ? (size_t)2048 * (4 + 1024 * 16) + 256 : 0;
    const size_t o_hits = carve(hit_bytes ? hit_bytes : 16);
    // the fallback sub-batch (depth 1) runs while the caller's scratch is no longer needed: one buffer serves both
    st = sc_grow(ix, &ix->bscratch, &ix->bscratch_cap, off);
    if (st) return st;
    char* b = (char*)ix->bscratch;
    void* Qb = b + o_qb;
    float* qres = (float*)(b + o_qres);
    float *qscale = (float*)(b + o_qs), *thr = (float*)(b + o_thr), *tf = (float*)(b + o_tf);
    unsigned* cnt = (unsigned*)(b + o_cnt);
    int *ovf = (int*)(b + o_ovf), *flags = (int*)(b + o_flag);
    uint64_t *best = (uint64_t*)(b + o_best), *ekeys = (uint64_t*)(b + o_ek), *surv = (uint64_t*)(b + o_surv);

    sc_launch_ingest_rows(q_dev, nullptr, 0, Q, ix->dim, ix->qpad, ld, ix->qnorm, s);
    if (i8) sc_launch_query_i8(ix->qpad, Q, Qpad, ld, ld8, Qb, qscale, qres, (unsigned*)(b + o_amax), s);
    else sc_launch_query_bf16(ix->qpad, Q, Qpad, ld, Qb, qres, s);
    sc_launch_scan_batched_init(thr, tf, Qpad, best, cnt, ovf, Q, wide ? 0 : KP, s);  // (wide: `best` carries its own counts, no padding)
    unsigned* nbest = (unsigned*)(b + o_nbest);
    if (wide) SC_HIP(hipMemsetAsync(nbest, 0, (size_t)Q * 4, s));
    ix->last_wide = wide ? 1 : 0;
    // thresholds from exact scores before the large phases (scan_batched.hip, scan_tighten_kernel): from 2^17 rows seen on
    // (10M x 768 x 1024, same box: from 2^19 8.45 ms per step, 2^17 8.38, 2^15 8.36; without 8.80)
    float* thr_cut = (float*)(b + o_cut);
    bool cut_used = false;
    if (tighten) sc_launch_fill_u32((unsigned*)thr_cut, 0x7F800000u, Qpad, s);  // +inf
    int64_t r0 = 0, span = phase0_rows();
    while (r0 < ix->n) {
        const int64_t r1 = std::min(ix->n, r0 + span);
        hipEvent_t e0, e1;
        static const int64_t tighten_from = [] { const char* e = getenv("SC_TIGHTEN_FROM"); return e ? (int64_t)atoll(e) : ((int64_t)1 << 17); }();  // A/B
        // (wide form: a cut before every phase but the first -- nothing may be truncated at kp while keys within reach of the k-th exact
        // score can still arrive; the first selection keeps all of its 2 048 rows)
        if (tighten && r0 >= (wide ? (int64_t)1 : tighten_from)) {
            uint64_t *b128 = (uint64_t*)(b + o_b128), *e128 = (uint64_t*)(b + o_e128);
            unsigned* cnt2 = (unsigned*)(b + o_cnt2);
            sc_prof_begin(rt, SC_PROF_MERGE, &e0, &e1);
            const uint64_t* from = best;
            if (wide || KP > TK) {  // the TK best of the candidates (a selection over `best` as if it were a survivor list)
                if (wide) SC_HIP(hipMemcpyAsync(cnt2, nbest, (size_t)Q * 4, hipMemcpyDeviceToDevice, s));
                else sc_launch_fill_u32(cnt2, (unsigned)KP, Q, s);
                SC_HIP(hipMemsetAsync(b128, 0xFF, (size_t)Q * TK * 8, s));
                sc_launch_scan_select(metric, best, cnt2, KB, b128, ix->qnorm, (float*)(b + o_thrT), (float*)(b + o_tfT), (int*)(b + o_wnc), Q, TK, s);
                from = b128;
            }
            sc_launch_scan_rerank_keys(metric, ix->X, ix->xnorm, ld, ix->qpad, ix->qnorm, from, nullptr, TK, ix->perm, e128, Q, s);
            sc_launch_scan_tighten(metric, e128, TK, k, ix->qnorm, qres, i8 ? ix->xnorm_max8 : ix->xnorm_max, ld, thr, tf, thr_cut, Q, s);
            sc_prof_end(rt, SC_PROF_MERGE, e0, e1);
            cut_used = true;
        }
        sc_prof_begin(rt, SC_PROF_SCAN, &e0, &e1);
        // a query keeps about KP of the r0 rows seen so far: a 256 x 256 tile of this phase about 65536 KP / r0 survivors -- above a few
        // hundred the two-pass epilogue (one list-slot atomic per query and tile instead of one per survivor)
        const bool dense = r0 < (int64_t)256 * KP;
        if (i8) sc_launch_scan_coarse(metric, ix->Xq, ix->xnorm, r0, r1, ld8, Qb, ix->qnorm, Q, Qpad, thr, tf, surv, cnt, BATCH_CAP, s, true, ix->xscale, qscale, dense, hit_bytes ? (void*)(b + o_hits) : nullptr, hit_bytes);
        else sc_launch_scan_coarse(metric, ix->Xb, ix->xnorm, r0, r1, ld, Qb, ix->qnorm, Q, Qpad, thr, tf, surv, cnt, BATCH_CAP, s, false, nullptr, nullptr, dense);
        sc_prof_end(rt, SC_PROF_SCAN, e0, e1);
        sc_prof_begin(rt, SC_PROF_MERGE, &e0, &e1);
        if (wide) sc_launch_scan_select_wide(metric, surv, cnt, BATCH_CAP, best, nbest, WB, cut_used ? KP : WB, ix->qnorm, thr, tf, thr_cut, ovf, Q, s);
        else sc_launch_scan_select(metric, surv, cnt, BATCH_CAP, best, ix->qnorm, thr, tf, ovf, Q, KP, s);
        sc_prof_end(rt, SC_PROF_MERGE, e0, e1);
        r0 = r1;
        span *= 4;
    }
    if (cut_used) sc_launch_scan_thr_min(thr, thr_cut, Q, s);  // the certificate's threshold: no looser than any cut that was applied
    // (the plain form can take the same final step over its kp slots -- SC_FINAL_COMPACT=1 -- but gains nothing from it: on the Gaussian
    // benchmark ~200 of the 512 lie within the final threshold, and the step measures 8.47 ms either way)
    static const bool compact_env = [] { const char* e = getenv("SC_FINAL_COMPACT"); return e && e[0] == '1'; }();
    const bool compact_final = !wide && i8 && cut_used && compact_env;
    if (wide || compact_final) {  // the keys within the final threshold, re-scored exactly; exact top-k; the certificate as a kernel of its own
        uint64_t* wcand = (uint64_t*)(b + o_wcand);
        int* wnc = (int*)(b + o_wnc);
        hipEvent_t e0, e1;
        sc_prof_begin(rt, SC_PROF_MERGE, &e0, &e1);
        if (!wide) sc_launch_fill_u32(nbest, (unsigned)KP, Q, s);
        sc_launch_scan_wide_compact(metric, best, nbest, KB, thr, wcand, wnc, Q, s);
        sc_launch_scan_rerank_keys(metric, ix->X, ix->xnorm, ld, ix->qpad, ix->qnorm, wcand, wnc, KB, ix->perm, ekeys, Q, s);
        SC_HIP(hipMemsetAsync(flags, 0, (size_t)Q * 4, s));
        sc_launch_refine_finalize(metric, ekeys, wnc, flags, k, ix->row_base, out_dist, out_rows, Q, s, KB);
        sc_launch_scan_wide_certify(metric, out_dist, k, ix->qnorm, qres, ix->xnorm_max8, ld, thr, ovf, flags, Q, s);
        sc_prof_end(rt, SC_PROF_MERGE, e0, e1);
    } else {
        sc_launch_scan_rerank(metric, ix->X, ix->xnorm, ld, ix->qpad, ix->qnorm, best, thr, i8 ? ix->xnorm_max8 : ix->xnorm_max, qres, ovf, Q, k, ix->row_base,
                              ix->perm, out_dist, out_rows, flags, s, KP, ekeys);
    }
    SC_HIP(hipGetLastError());
    // uncertified queries: hand them to the next stage (int8 -> bf16 -> exact scan)
    std::vector<int> hflags(Q);
    SC_HIP(hipMemcpyAsync(hflags.data(), flags, (size_t)Q * 4, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    std::vector<int> redo;
    for (int i = 0; i < Q; ++i)
        if (hflags[i]) redo.push_back(i);
    int R = (int)redo.size();
    if (i8) {
        // most of a real batch uncertified: this corpus does not quantise well enough (tight clusters, outlier dimensions) --
        // later searches start at the bf16 stage until the rows are replaced wholesale
        // (first resort: the wide candidate set -- the next batch keeps every key within the exact-score cut; if that fails too, bf16)
        if (depth == 0 && coarse_pin(ix) != 8 && !ix->i8_sticky && !big_k) {  // (a large top_k says nothing about the corpus)
            const bool wide_possible = tighten && Q <= 16384 && ix->n > ((int64_t)1 << 18);
            // the wide form is never wrong and costs a few percent where it is not needed: a small batch that fails is evidence enough for it
            // (one clustered query: 3.1 ms through plain form + collect pass, 2.0 ms wide); giving up on int8 takes a real batch
            if (!wide && wide_possible && !ix->wide_i8) {
                if (R * 2 > Q) ix->wide_i8 = true;
            } else if (Q >= 32 && R * 4 > Q) {
                ix->i8_off = true;
            }
        }
    }
    // second chance at this stage's precision: the collect pass (every row within the coarse error of the k-th exact score found)
    bool& collect_off = i8 ? ix->collect_off8 : ix->collect_off16;
    if (R > 0 && g_collect_pass && !collect_off) {
        void** buf = depth == 0 ? &ix->fb : &ix->fb2;
        size_t* cap = depth == 0 ? &ix->fb_cap : &ix->fb2_cap;
        const size_t qb = ((size_t)R * ix->dim * 4 + 255) & ~(size_t)255, db = ((size_t)R * k * 4 + 255) & ~(size_t)255, rb = ((size_t)R * k * 8 + 255) & ~(size_t)255;
        st = sc_grow(ix, buf, cap, qb + db + rb + (size_t)R * 4);
        if (st) return st;
        float* fq = (float*)*buf;
        float* fd = (float*)((char*)*buf + qb);
        int64_t* fr = (int64_t*)((char*)*buf + qb + db);
        int32_t* fidx = (int32_t*)((char*)*buf + qb + db + rb);
        SC_HIP(hipMemcpyAsync(fidx, redo.data(), (size_t)R * 4, hipMemcpyHostToDevice, s));
        sc_launch_copy_rows_indexed(q_dev, fq, fidx, R, (size_t)ix->dim * 4, false, s);
        sc_launch_copy_rows_indexed(out_dist, fd, fidx, R, (size_t)k * 4, false, s);  // the failed pass's results: their k-th score is the bound
        sc_launch_copy_rows_indexed(out_rows, fr, fidx, R, (size_t)k * 8, false, s);
        std::vector<int> left;
        st = search_collect_locked(ix, fq, R, k, fd, fr, i8, left);
        if (st) return st;
        sc_launch_copy_rows_indexed(fd, out_dist, fidx, R, (size_t)k * 4, true, s);
        sc_launch_copy_rows_indexed(fr, out_rows, fidx, R, (size_t)k * 8, true, s);
        SC_HIP(hipStreamSynchronize(s));
        ix->last_collect_tried += R;
        ix->last_collect_resolved += R - (int)left.size();
        if (R >= 32 && (int)left.size() * 2 > R) collect_off = true;
        std::vector<int> still;
        for (int j : left) still.push_back(redo[(size_t)j]);
        redo.swap(still);
        R = (int)redo.size();
    }
    if (i8) ix->last_uncert_i8 = R;
    // (a handful of queries is one pass of the exact scan: not worth a bf16 shadow; top_k beyond 64 is beyond the bf16 stage's 128 candidates)
    const bool to_bf16 = i8 && coarse_pin(ix) != 8 && R > 16 && k <= sc_batched_kprime() / 2;
    if (!to_bf16) {  // what is left goes to the exact scan
        ix->last_uncertified = R;
        ix->uncert_frac = (double)R / (double)Q_top;
    }
    if (R > 0) {
        // the sub-batch gets its own staging (queries + results); nested stages each need one: fb for the first, fb2 for the second
        void** buf = depth == 0 ? &ix->fb : &ix->fb2;
        size_t* cap = depth == 0 ? &ix->fb_cap : &ix->fb2_cap;
        const size_t qb = ((size_t)R * ix->dim * 4 + 255) & ~(size_t)255, db = ((size_t)R * k * 4 + 255) & ~(size_t)255, rb = ((size_t)R * k * 8 + 255) & ~(size_t)255;
        st = sc_grow(ix, buf, cap, qb + db + rb + (size_t)R * 4);
        if (st) return st;
        float* fq = (float*)*buf;
        float* fd = (float*)((char*)*buf + qb);
        int64_t* fr = (int64_t*)((char*)*buf + qb + db);
        int32_t* fidx = (int32_t*)((char*)*buf + qb + db + rb);
        // one gather and two scatters by query index (one hipMemcpyAsync per query cost ~9 us each)
        SC_HIP(hipMemcpyAsync(fidx, redo.data(), (size_t)R * 4, hipMemcpyHostToDevice, s));
        sc_launch_copy_rows_indexed(q_dev, fq, fidx, R, (size_t)ix->dim * 4, false, s);
        if (to_bf16) st = search_batched_stage_locked(ix, fq, R, k, fd, fr, false, depth + 1, Q_top);
        else st = search_exact_locked(ix, fq, R, k, 0, fd, fr);
        if (st) return st;
        sc_launch_copy_rows_indexed(fd, out_dist, fidx, R, (size_t)k * 4, true, s);
        sc_launch_copy_rows_indexed(fr, out_rows, fidx, R, (size_t)k * 8, true, s);
        SC_HIP(hipStreamSynchronize(s));  // `redo` is on this stack frame
    }
    ix->last_path = 2;
    return SC_OK;
}

static sc_status search_batched_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, float* out_dist, int64_t* out_rows) {
    const int env = coarse_pin(ix);
    // The int8 stage halves the coarse GEMM but re-ranks 512 candidates per query instead of 128 (512 * ld * 4 B of scattered rows
    // each): it pays from about a million rows up (10M x 768: coarse 12.9 -> 7.0 ms against +0.3 ms of re-rank).  Small corpora --
    // above all the IVF quantizer, whose nearest-centroid searches of a build went 6.4 -> 16.4 s through it at 4096 x 3072
    // (profiles/r2i_kernel_stats.csv: scan_rerank_kernel 7.4 s) -- start at the bf16 stage.
    // (search mode 2, "batched whenever supported", is the tests' switch: it keeps the int8 stage eligible at any size.)
    // ... and from 129 queries up: the int8 stage always runs 256-query tiles, and a batch of 32 spends 4.05 ms in them against the
    // 3.8 ms of the bf16 stage's 128-query tiles (profiles/r2p_bench.json.log sweep vs r1v)
    // (round 3: the persistent int8 kernel answers a 32-query batch in 3.1 ms where the bf16 stage's 128-query tiles take 3.9: the
    // int8 stage now starts at 17 queries; SC_I8_MINQ restores any other limit for A/B runs)
    const int i8_minq = i8_min_queries();  // (round 3, later: from one query on -- batched_applicable)
    const bool i8 = env == 8 || (env == 0 && !ix->i8_off && ((ix->n >= i8_min_rows() && Q >= i8_minq) || (Q > 16 && Q <= 64 && ix->n >= 65536) || ix->search_mode == 2));
    // (17 .. 64 queries from 65 536 rows on: that batch size is the narrow streaming kernel's -- 1M x 768, 64 queries: 0.71 ms against
    // 2.01 through the bf16 stage's 128-query tiles; 300k rows: 0.51 against 1.74 -- profiles/r3z_q_rows.log)
    ix->last_coarse_bits = i8 ? 8 : 16;
    ix->last_uncert_i8 = 0;
    ix->last_uncertified = 0;
    ix->last_collect_tried = ix->last_collect_resolved = 0;
    // Which stage to START at on a corpus the int8 certificate fails on is settled by the clock: the int8 stage switches itself off
    // when it fails for a quarter of a batch (above); the cost per query of that batch (int8 pass + its collect pass + whatever went
    // on) is remembered, and if the bf16-first batch that follows costs more (tight clusters: bf16 needs its collect pass too, at
    // twice the bytes and half the MFMA rate), the int8 stage is switched back on for good.  10M x 768, 4096 clusters of spread
    // 0.1: 26.2 ms bf16-first, 16.9 ms int8-first (profiles/r3z_clustered_probe.log).
    const bool was_off = ix->i8_off;
    const auto t0 = std::chrono::steady_clock::now();
    const sc_status st = search_batched_stage_locked(ix, q_dev, Q, k, out_dist, out_rows, i8, 0, Q);
    if (st == SC_OK && env == 0 && Q >= 64 && !ix->i8_sticky) {
        const double per_q = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / Q;
        if (i8 && !was_off && ix->i8_off) ix->cost_i8_first = per_q;  // the batch that switched the int8 stage off
        else if (!i8 && was_off && ix->cost_i8_first > 0.0 && ix->last_collect_tried * 4 > Q) {
            if (ix->cost_i8_first < 0.85 * per_q) {
                ix->i8_off = false;
                ix->i8_sticky = true;
            }
            ix->cost_i8_first = 0.0;  // decided either way
        }
    }
    return st;
}

sc_status sc_search_flat_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, float* out_dist, int64_t* out_rows) {
    ix->last_uncertified = 0;
    if (batched_applicable(ix, Q, k)) return search_batched_locked(ix, q_dev, Q, k, out_dist, out_rows);
    return search_exact_locked(ix, q_dev, Q, k, 0, out_dist, out_rows);
}

static int64_t g_ivf_tail_rows = 65536;  // sc_diag_set_option("ivf_tail_rows", n): appended rows a trained index leaves behind its lists (0: fold them in at once)
void sc_set_ivf_tail_rows(int v) { g_ivf_tail_rows = v < 0 ? 65536 : v; }

static sc_status probe_dispatch_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe, float* out_dist, int64_t* out_rows);

static sc_status search_dev_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe, float* out_dist,
                                   int64_t* out_rows) {
    ix->last_probed_lists = 0;
    ix->last_unique_rows = ix->last_streamed_rows = 0;
    ix->last_groups = 0;
    ix->last_tail_rows = 0;
    // Rows APPENDED to a trained index since its lists were laid out.  Folding them in means re-ordering the corpus (a second copy
    // of it, the shadows that mirror the layout): 1.9 - 4.1 s per search at 10M x 768 when searches and upserts alternate
    // (scripts/upsert_search_interleave.py).  Milvus answers from its growing segment by brute force; the same here: up to 65 536
    // appended rows stay behind the lists as a tail, a probe answers from the lists AND from an exact scan of the tail (positions
    // there are row ids), merged; beyond that, or when listed rows were overwritten (their list may have changed), or for an
    // exhaustive search, the lists are refreshed as before.  A tail row is always seen -- the probed lists plus the whole tail --
    // so recall can only be higher than after the refresh.
    {
        int64_t tail = (ix->kind == SC_INDEX_IVF_FLAT && ix->trained && ix->perm) ? ix->n - ix->ivf_rows : 0;
        if (tail > 0 && tail <= g_ivf_tail_rows && !ix->dirty_rows.empty() && ix->search_mode != 1 && ix->search_mode != 2) {
            // listed rows were overwritten as well (a re-index: known chunks again, new ones appended): settle those first -- rows that
            // stayed in their lists (unchanged or lightly edited chunks) leave the layout alone and the tail a tail
            const sc_status rst = sc_ivf_refresh_locked(ix, true);
            if (rst && rst != SC_ERR_NOMEM) return rst;
            if (rst) (void)hipGetLastError();
            tail = ix->n - ix->ivf_rows;
        }
        if (tail > 0 && tail <= g_ivf_tail_rows && ix->dirty_rows.empty() && ix->search_mode != 1 && ix->search_mode != 2 && nprobe >= 1 && nprobe < ix->nlist_trained &&
            (sc_ivf_coarse_applicable(ix, Q, k, nprobe) || sc_ivf_applicable(ix, Q, nprobe) ||
             sc_ivf_listmajor_applicable(ix, Q, k, nprobe, batched_applicable(ix, Q, k)))) {
            const size_t db = ((size_t)Q * k * 4 + 255) & ~(size_t)255, rb = ((size_t)Q * k * 8 + 255) & ~(size_t)255;
            sc_status st = sc_grow(ix, &ix->tailbuf, &ix->tailbuf_cap, 2 * (db + rb));
            if (st) return st;
            char* tb = (char*)ix->tailbuf;
            float *d1 = (float*)tb, *d2 = (float*)(tb + db);
            int64_t *r1 = (int64_t*)(tb + 2 * db), *r2 = (int64_t*)(tb + 2 * db + rb);
            st = probe_dispatch_locked(ix, q_dev, Q, k, nprobe, d1, r1);
            if (st) return st;
            const int path = ix->last_path, unc = ix->last_uncertified;
            st = search_exact_range_locked(ix, q_dev, Q, k, ix->ivf_rows, tail, d2, r2);
            if (st) return st;
            sc_launch_topk_merge2((int)ix->metric, d1, r1, d2, r2, k, out_dist, out_rows, Q, ix->rt->stream);
            SC_HIP(hipGetLastError());
            ix->last_path = path;
            ix->last_uncertified = unc;
            ix->last_tail_rows = tail;
            return SC_OK;
        }
    }
    {   // rows upserted since the IVF lists were built join their lists first (no k-means): the reported ids of a
        // list-major corpus go through ix->perm, which must cover every stored row
        sc_status rst = sc_ivf_refresh_locked(ix);
        if (rst == SC_ERR_NOMEM && ix->perm) {
            // The re-layout needs a second copy of the corpus (246 GB at 10M x 3072).  Without it the rows upserted since the build
            // cannot join their lists -- but they can still be FOUND: extend the position -> row id map over the tail (positions
            // == row ids there, 4 B per row) and answer exhaustively (exact results) until a refresh or a rebuild succeeds.
            rst = sc_ivf_cover_tail_locked(ix);
            if (rst) return rst;
            return sc_search_flat_locked(ix, q_dev, Q, k, out_dist, out_rows);
        }
        if (rst) return rst;
    }
    return probe_dispatch_locked(ix, q_dev, Q, k, nprobe, out_dist, out_rows);
}

// which path answers (the lists cover every stored row, or the caller takes care of the tail)
static sc_status probe_dispatch_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe, float* out_dist, int64_t* out_rows) {
    if (!sc_ivf_coarse_applicable(ix, Q, k, nprobe) && sc_ivf_applicable(ix, Q, nprobe)) return sc_ivf_search_locked(ix, q_dev, Q, k, nprobe, out_dist, out_rows);
    if (sc_ivf_coarse_applicable(ix, Q, k, nprobe)) {
        // per-query scratch of the coarse stage is ~200 KB (two survivor lists of 8 192 keys, the refine sets): very large batches go
        // through it in chunks of 4 096 queries (0.8 GB), each a full batch of its own
        const int chunk = 4096;
        int uncert = 0;
        int64_t uniq = 0, streamed = 0;
        int groups = 0;
        for (int q0 = 0; q0 < Q; q0 += chunk) {
            const int nq = std::min(chunk, Q - q0);
            const sc_status st = sc_ivf_search_coarse_locked(ix, q_dev + (size_t)q0 * ix->dim, nq, k, nprobe, out_dist + (size_t)q0 * k, out_rows + (size_t)q0 * k);
            if (st == SC_ERR_NOMEM && q0 == 0) {
                // no room for the centred shadow (a quarter of the corpus again) or the stage's scratch: the exact probes need neither.
                // The stage stays off until the lists are rebuilt (a failed hipMalloc of tens of GB per search is not free either).
                (void)hipGetLastError();
                ix->ivfc_off = true;
                hipFree(ix->Xc8); ix->Xc8 = nullptr; ix->xc8_cap = 0; ix->shadowc_rows = 0;
                hipFree(ix->ivfc_scratch); ix->ivfc_scratch = nullptr; ix->ivfc_scratch_cap = 0;
                goto exact_probe;
            }
            if (st) return st;
            uncert += ix->last_uncertified;
            uniq = std::max(uniq, ix->last_unique_rows);
            streamed += ix->last_streamed_rows;
            groups += ix->last_groups;
        }
        ix->last_uncertified = ix->last_ivfc_uncertified = uncert;
        ix->last_unique_rows = uniq;
        ix->last_streamed_rows = streamed;
        ix->last_groups = groups;
        return SC_OK;
    }
exact_probe:
    if (sc_ivf_listmajor_applicable(ix, Q, k, nprobe, batched_applicable(ix, Q, k)))
        return sc_ivf_search_listmajor_locked(ix, q_dev, Q, k, nprobe, out_dist, out_rows);
    if (ix->perm && ix->perm_rows < ix->n && ix->n > ix->ivf_rows) {  // (a tail behind the lists: the exhaustive paths need every position mapped)
        const sc_status cst = sc_ivf_cover_tail_locked(ix);
        if (cst) return cst;
    }
    return sc_search_flat_locked(ix, q_dev, Q, k, out_dist, out_rows);
}
sc_status sc_search_dev_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe, float* out_dist, int64_t* out_rows) {
    return search_dev_locked(ix, q_dev, Q, k, nprobe, out_dist, out_rows);
}

extern "C" sc_status sc_index_set_search_mode(sc_index* ix, int32_t mode) {
    if (!ix || mode < 0 || mode > 5)
        return sc_fail(SC_ERR_INVALID, "sc_index_set_search_mode: mode must be 0 (auto), 1 (exact), 2 (batched), 3 (ivf probe per query), 4 (ivf probe list-major, exact f32) "
                                       "or 5 (ivf probe list-major behind the int8 coarse stage)");
    std::lock_guard<std::mutex> g(ix->mu);
    ix->search_mode = mode;
    return SC_OK;
}

extern "C" sc_status sc_index_last_search_stats(sc_index* ix, int32_t* path, int32_t* uncertified) {
    if (!ix) return sc_fail(SC_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> g(ix->mu);
    if (path) *path = ix->last_path;
    if (uncertified) *uncertified = ix->last_uncertified;
    return SC_OK;
}

extern "C" sc_status sc_index_set_coarse_stage(sc_index* ix, int32_t bits) {
    if (!ix || (bits != 0 && bits != 8 && bits != 16)) return sc_fail(SC_ERR_INVALID, "sc_index_set_coarse_stage: bits must be 0 (auto), 8 or 16");
    std::lock_guard<std::mutex> g(ix->mu);
    ix->coarse_mode = bits;
    if (bits == 0) { ix->i8_off = false; ix->wide_i8 = false; ix->i8_sticky = false; ix->cost_i8_first = 0.0; }
    ix->collect_off8 = ix->collect_off16 = false;
    return SC_OK;
}

extern "C" sc_status sc_index_last_coarse_stats(sc_index* ix, int32_t* first_stage_bits, int32_t* handed_to_bf16) {
    if (!ix) return sc_fail(SC_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> g(ix->mu);
    if (first_stage_bits) *first_stage_bits = ix->last_coarse_bits;
    if (handed_to_bf16) *handed_to_bf16 = ix->last_uncert_i8;
    return SC_OK;
}

extern "C" sc_status sc_index_last_collect_stats(sc_index* ix, int32_t* tried, int32_t* resolved) {
    if (!ix) return sc_fail(SC_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> g(ix->mu);
    if (tried) *tried = ix->last_collect_tried;
    if (resolved) *resolved = ix->last_collect_resolved;
    return SC_OK;
}

extern "C" sc_status sc_index_last_tail_rows(sc_index* ix, int64_t* rows) {
    if (!ix) return sc_fail(SC_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> g(ix->mu);
    if (rows) *rows = ix->last_tail_rows;
    return SC_OK;
}

extern "C" sc_status sc_index_last_wide(sc_index* ix, int32_t* wide) {
    if (!ix) return sc_fail(SC_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> g(ix->mu);
    if (wide) *wide = ix->last_wide;
    return SC_OK;
}

extern "C" sc_status sc_index_last_probe_stats(sc_index* ix, int64_t* unique_rows, int64_t* streamed_rows, int32_t* groups) {
    if (!ix) return sc_fail(SC_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> g(ix->mu);
    if (unique_rows) *unique_rows = ix->last_unique_rows;
    if (streamed_rows) *streamed_rows = ix->last_streamed_rows;
    if (groups) *groups = ix->last_groups;
    return SC_OK;
}

static sc_status check_search_args(sc_index* ix, const void* q, int32_t Q, int32_t k, const void* od, const void* orow) {
    if (!ix || !q || !od || !orow) return sc_fail(SC_ERR_INVALID, "search: NULL argument");
    if (Q < 1 || Q > (1 << 20)) return sc_fail(SC_ERR_INVALID, "search: Q=%d out of range", Q);
    if (k < 1) return sc_fail(SC_ERR_INVALID, "search: top_k must be >= 1 (got %d)", k);
    return SC_OK;
}

extern "C" sc_status sc_index_search_dev(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe, float* out_dist_dev,
                                         int64_t* out_rows_dev) {
    sc_status st = check_search_args(ix, q_dev, Q, k, out_dist_dev, out_rows_dev);
    if (st) return st;
    std::lock_guard<std::mutex> g(ix->mu);
    SC_HIP(hipSetDevice(ix->rt->device));
    return search_dev_locked(ix, q_dev, Q, k, nprobe, out_dist_dev, out_rows_dev);
}

extern "C" sc_status sc_index_search(sc_index* ix, const float* q, int32_t Q, int32_t k, int32_t nprobe, float* out_dist,
                                     int64_t* out_rows) {
    sc_status st = check_search_args(ix, q, Q, k, out_dist, out_rows);
    if (st) return st;
    std::lock_guard<std::mutex> g(ix->mu);
    SC_HIP(hipSetDevice(ix->rt->device));
    hipStream_t s = ix->rt->stream;
    const size_t qb = ((size_t)Q * ix->dim * 4 + 15) & ~(size_t)15;
    const size_t db = ((size_t)Q * k * 4 + 15) & ~(size_t)15;
    const size_t rb = (size_t)Q * k * 8;
    st = sc_grow(ix, (void**)&ix->io, &ix->io_cap, qb + db + rb);
    if (st) return st;
    float* dq = (float*)ix->io;
    float* dd = (float*)((char*)ix->io + qb);
    int64_t* dr = (int64_t*)((char*)ix->io + qb + db);
    SC_HIP(hipMemcpyAsync(dq, q, (size_t)Q * ix->dim * 4, hipMemcpyHostToDevice, s));
    st = search_dev_locked(ix, dq, Q, k, nprobe, dd, dr);
    if (st) return st;
    SC_HIP(hipMemcpyAsync(out_dist, dd, (size_t)Q * k * 4, hipMemcpyDeviceToHost, s));
    SC_HIP(hipMemcpyAsync(out_rows, dr, (size_t)Q * k * 8, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    return SC_OK;
}

// ------------------------------------------------------------------ host merge of per-shard results

extern "C" sc_status sc_topk_merge_host(sc_metric metric, int32_t lists, int32_t Q, int32_t k, const float* dist, const int64_t* rows,
                                        float* out_dist, int64_t* out_rows) {
    if (lists < 1 || Q < 1 || k < 1 || !dist || !rows || !out_dist || !out_rows) return sc_fail(SC_ERR_INVALID, "sc_topk_merge_host: bad argument");
    struct Ent { uint32_t u; int64_t row; float d; };
    std::vector<Ent> v;
    v.reserve((size_t)lists * k);
    for (int q = 0; q < Q; ++q) {
        v.clear();
        for (int l = 0; l < lists; ++l)
            for (int j = 0; j < k; ++j) {
                const size_t o = ((size_t)l * Q + q) * k + j;
                if (rows[o] < 0) continue;
                float x = (metric == SC_METRIC_L2) ? dist[o] : -dist[o];
                x = x + 0.0f;
                uint32_t u;
                memcpy(&u, &x, 4);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                v.push_back({u, rows[o], dist[o]});
            }
        std::sort(v.begin(), v.end(), [](const Ent& a, const Ent& b) { return a.u != b.u ? a.u < b.u : a.row < b.row; });
        for (int j = 0; j < k; ++j) {
            const size_t o = (size_t)q * k + j;
            if (j < (int)v.size()) {
                out_dist[o] = v[j].d;
                out_rows[o] = v[j].row;
            } else {
                out_dist[o] = (metric == SC_METRIC_L2) ? INFINITY : -INFINITY;
                out_rows[o] = -1;
            }
        }
    }
    return SC_OK;
}

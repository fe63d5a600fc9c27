/*
 * semcode_hip.h -- C ABI of libsemcode_hip.so (MI355X / gfx950 backend for semcode's
 * embed -> store -> top-k path).
 *
 * This is the drop-in boundary: plain C, opaque handles, plain pointers and sizes, no
 * torch / C++ types.  The only callers are the two Python seam classes in
 * semcode_amd/embeddings/providers.py and semcode_amd/storage/milvus_store.py (via ctypes),
 * bench.py and the tests.
 *
 * Reference interfaces each group replaces (paths relative to the reference checkout):
 *   - encoder  : Embeddings.embed_documents / embed_query reached through
 *                EmbeddingProviderFactory.create   src/semcode/embeddings/providers.py:34-104
 *                called at                         src/semcode/services/indexer.py:150
 *                                                  src/semcode/rag/pipeline.py:171-175
 *   - index    : pymilvus Collection.{create_index,upsert,search,load} as used by
 *                MilvusVectorStore                  src/semcode/storage/milvus_store.py:39-148
 *
 * Conventions
 *   - every function returns sc_status (0 = OK, negative = error); the message of the last
 *     error raised on the calling thread is available through sc_last_error();
 *     no exception or abort crosses this boundary;
 *   - host pointers are contiguous, little-endian, row-major; the caller owns every buffer
 *     it passes, the library owns only handle-internal device memory;
 *   - "_dev" variants take DEVICE pointers (e.g. torch tensor data_ptr()) and enqueue on the
 *     runtime's stream without synchronising; host variants synchronise before returning;
 *   - row ids are int64 row numbers (shard base + local row); string primary keys and
 *     metadata stay in Python (milvus_store.py:110-130 column lists);
 *   - a handle may be used from several threads (FastAPI threadpool, api/main.py:202):
 *     calls on one handle are serialised by a per-handle mutex.
 */
#ifndef SEMCODE_HIP_H
#define SEMCODE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t sc_status;

enum {
    SC_OK = 0,
    SC_ERR_INVALID = -1,     /* bad argument                                  */
    SC_ERR_HIP = -2,         /* a HIP runtime call failed (no GPU, OOM, ...)   */
    SC_ERR_STATE = -3,       /* call not valid in the handle's current state  */
    SC_ERR_UNSUPPORTED = -4, /* shape / option outside what the kernels cover */
    SC_ERR_NOMEM = -5        /* host or device allocation failed              */
};

/* metric_type of milvus_store.py:78-82,144 ("IP" in the reference; L2 and COSINE are the
 * other two Milvus float-vector metrics and are what BASELINE.json measures). */
typedef enum { SC_METRIC_IP = 0, SC_METRIC_L2 = 1, SC_METRIC_COSINE = 2 } sc_metric;
/* index_type of milvus_store.py:80 ("IVF_FLAT"); FLAT = exhaustive scan. */
typedef enum { SC_INDEX_FLAT = 0, SC_INDEX_IVF_FLAT = 1 } sc_index_kind;

typedef struct sc_runtime sc_runtime;
typedef struct sc_index sc_index;
typedef struct sc_encoder sc_encoder;

/* ------------------------------------------------------------------ runtime ---- */

typedef struct sc_runtime_cfg {
    int32_t device;  /* HIP device ordinal (one process per GPU: LOCAL_RANK)            */
    void* stream;    /* hipStream_t to enqueue on, or NULL = library-owned stream       */
    int32_t flags;   /* reserved, 0                                                     */
} sc_runtime_cfg;

/* Library version string ("semcode_hip x.y"). Never fails. */
const char* sc_version(void);
/* Copies the calling thread's last error message into buf (NUL-terminated, truncated to n). */
sc_status sc_last_error(char* buf, size_t n);

/* Replaces pymilvus connections.connect(...) (milvus_store.py:42-47): binds a device. */
sc_status sc_runtime_create(const sc_runtime_cfg* cfg, sc_runtime** out);
sc_status sc_runtime_destroy(sc_runtime* rt);
/* Re-point the runtime at another hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
sc_status sc_runtime_set_stream(sc_runtime* rt, void* stream);
/* Block until everything enqueued on the runtime's stream has finished. */
sc_status sc_runtime_synchronize(sc_runtime* rt);
/* Device facts for bench.py / DESIGN.md: name (<=255 chars), CU count, total HBM bytes. */
sc_status sc_runtime_device_info(sc_runtime* rt, char* name, size_t n, int32_t* cus, int64_t* hbm_bytes);
/* When enabled, the dominant kernel of each search / embed call is bracketed by hipEvents on
 * the runtime's stream (bench.py's roofline figure). */
sc_status sc_runtime_set_profiling(sc_runtime* rt, int32_t enabled);
/* Sum (ms) and count of profiled launches of kernel class `which` since the last reset
 * (which: 0 = distance scan, 1 = top-k merge, 2 = encoder GEMM, 3 = attention). Synchronises. */
sc_status sc_runtime_profile_read(sc_runtime* rt, int32_t which, double* total_ms, int64_t* launches);
sc_status sc_runtime_profile_reset(sc_runtime* rt);

/* Deterministic synthetic data (bench / tests): out[r*ld + c] = g(seed, first_row + r, c, dim)
 * for c < dim, 0 for dim <= c < ld, with g an integer-hash Irwin-Hall(12) approximation of
 * N(0,1) that oracle/sc_oracle.c reproduces bit for bit.  `out` is a DEVICE pointer. */
sc_status sc_synth_fill_dev(sc_runtime* rt, float* out, int64_t rows, int32_t dim, int32_t ld,
                            uint64_t seed, int64_t first_row);

/* -------------------------------------------------------------- vector index ---- */

/* Replaces Collection(...)+create_index(IVF_FLAT, metric, nlist) (milvus_store.py:59-84).
 * nlist is ignored for SC_INDEX_FLAT.  row_base = global id of local row 0 (shard offset). */
sc_status sc_index_create(sc_runtime* rt, int32_t dim, sc_metric metric, sc_index_kind kind,
                          int32_t nlist, int64_t row_base, sc_index** out);
sc_status sc_index_destroy(sc_index* ix);
/* Number of stored rows / dimension / padded row stride (floats). */
sc_status sc_index_info(sc_index* ix, int64_t* rows, int32_t* dim, int32_t* ld);
/* Pre-size device storage for `rows` rows (avoids regrowth copies). */
sc_status sc_index_reserve(sc_index* ix, int64_t rows);
/* Append n host vectors [n,dim]; they become rows [old_rows, old_rows+n).  Part of
 * Collection.upsert (milvus_store.py:128) for ids not seen before. */
sc_status sc_index_add(sc_index* ix, const float* vecs, int64_t n);
/* Replace existing rows: rows[i] (local row number) <- vecs[i].  The replace-by-primary-key half
 * of Collection.upsert (milvus_store.py:128); the md5 -> row map lives in Python. */
sc_status sc_index_overwrite(sc_index* ix, const float* vecs, const int64_t* rows, int64_t n);
/* Copy rows [first, first+n) back to the host as [n,dim] (persistence, tests). */
sc_status sc_index_get_rows(sc_index* ix, int64_t first, int64_t n, float* out);
/* Resize to n rows and fill them on device with sc_synth_fill_dev(seed, first_row). */
sc_status sc_index_fill_synthetic(sc_index* ix, int64_t n, uint64_t seed, int64_t first_row);

/* Replaces Collection.search(data=[vector], param={metric, nprobe}, limit=top_k)
 * (milvus_store.py:141-147), batched: q [Q,dim] host, out_dist [Q,k] f32, out_rows [Q,k] i64
 * (global ids, best first; ties broken by lower row id; missing hits = -1 / +inf-or--inf).
 * nprobe is ignored by FLAT indexes.  Distances: L2 = squared L2, IP = dot, COSINE = cosine. */
sc_status sc_index_search(sc_index* ix, const float* q, int32_t Q, int32_t k, int32_t nprobe,
                          float* out_dist, int64_t* out_rows);
/* Same with DEVICE pointers (q row stride = dim); asynchronous on the runtime's stream. */
sc_status sc_index_search_dev(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe,
                              float* out_dist_dev, int64_t* out_rows_dev);

/* Multi-GPU final step (one process per GPU): merge `lists` per-shard results
 * dist [lists,Q,k] / rows [lists,Q,k] (as produced by sc_index_search* on each shard and
 * all-gathered over RCCL) into the global best-first [Q,k], same tie rule.  Host buffers. */
sc_status sc_topk_merge_host(sc_metric metric, int32_t lists, int32_t Q, int32_t k,
                             const float* dist, const int64_t* rows, float* out_dist, int64_t* out_rows);

#ifdef __cplusplus
}
#endif
#endif /* SEMCODE_HIP_H */

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
 *     runtime's stream; where one of them has to wait for the stream (a flag read back, a host
 *     table that must outlive a kernel) its comment says so; host variants always synchronise;
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
/* Drops the creator's reference.  Indexes, encoders and communicators created on the runtime hold references of their own,
 * so they stay usable and may be destroyed afterwards, in any order; the stream is released with the last of them. */
sc_status sc_runtime_destroy(sc_runtime* rt);
/* Re-point the runtime at another hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
sc_status sc_runtime_set_stream(sc_runtime* rt, void* stream);
/* Block until everything enqueued on the runtime's stream has finished. */
sc_status sc_runtime_synchronize(sc_runtime* rt);
/* Device facts for bench.py / DESIGN.md: name (<=255 chars), CU count, total HBM bytes. */
sc_status sc_runtime_device_info(sc_runtime* rt, char* name, size_t n, int32_t* cus, int64_t* hbm_bytes);
/* When enabled (> 0), the dominant kernels of each search / embed call are bracketed by hipEvents on the runtime's
 * stream (bench.py's roofline figure): every launch of the scan classes; every `enabled`-th launch of the encoder
 * classes (GEMM, attention), whose ~60 launches per step would otherwise pay 1.6 % for their event pairs. */
sc_status sc_runtime_set_profiling(sc_runtime* rt, int32_t enabled);
/* Sum (ms) and count of profiled (bracketed) launches of kernel class `which` since the last reset
 * (which: 0 = distance scan, 1 = top-k merge, 2 = encoder GEMM, 3 = attention). Synchronises. */
sc_status sc_runtime_profile_read(sc_runtime* rt, int32_t which, double* total_ms, int64_t* launches);
sc_status sc_runtime_profile_reset(sc_runtime* rt);

/* Deterministic synthetic data (bench / tests): out[r*ld + c] = g(seed, first_row + r, c, dim)
 * for c < dim, 0 for dim <= c < ld, with g an integer-hash Irwin-Hall(12) approximation of
 * N(0,1) that oracle/sc_oracle.c reproduces bit for bit.  `out` is a DEVICE pointer. */
sc_status sc_synth_fill_dev(sc_runtime* rt, float* out, int64_t rows, int32_t dim, int32_t ld,
                            uint64_t seed, int64_t first_row);

/* ------------------------------------------------------------------- encoder ---- */

/* BERT-family post-LN transformer encoder (the "transformer-encoder forward" of BASELINE.json;
 * BERT-base shape = 30522 / 768 / 12 / 12 / 3072 / 512 / 2, eps 1e-12).  Head dimension must be 64. */
typedef struct sc_encoder_cfg {
    int32_t vocab, hidden, layers, heads, ffn, max_pos, type_vocab;
    float ln_eps;
    int32_t normalize;    /* 1 = L2-normalise the pooled vector                               */
    uint64_t synth_seed;  /* used only when no weight blob is given (benchmarks)              */
    int32_t pos_type;     /* 0 = learned absolute position table (BERT); 1 = ALiBi: no table, scores get
                             -slope_h * |i - j| (jina-embeddings-v2 family named in the reference README) */
    int32_t ffn_type;     /* 0 = Linear-GELU-Linear (BERT); 1 = GEGLU: W1 is [2*ffn, H] (gate rows first, then up
                             rows), hidden = gelu(gate) * up, b1 is [2*ffn] (zeros for bias-free models)      */
} sc_encoder_cfg;

/* Size in bytes of the f32 weight blob sc_encoder_create expects for cfg.  Blob order (all f32,
 * torch.nn.Linear layout [out, in]): word_emb [vocab,H], pos_emb [max_pos,H], type_emb [type_vocab,H],
 * emb_ln_gamma [H], emb_ln_beta [H], then per layer: Wq [H,H], bq, Wk, bk, Wv, bv, Wo [H,H], bo,
 * ln1_gamma, ln1_beta, W1 [ffn,H], b1 [ffn], W2 [H,ffn], b2 [H], ln2_gamma, ln2_beta.
 * pos_type 1 drops pos_emb from the blob; ffn_type 1 makes W1 [2*ffn,H] and b1 [2*ffn]. */
sc_status sc_encoder_blob_bytes(const sc_encoder_cfg* cfg, int64_t* out);
/* Replaces EmbeddingProviderFactory.create() loading a model (providers.py:69-100): uploads the
 * weights (converted to bf16 on device).  weights_blob == NULL: synthetic weights 0.02*N(0,1) from
 * cfg.synth_seed, LayerNorm gamma 1 / beta 0, biases 0 (random-init benchmark weights). */
sc_status sc_encoder_create(sc_runtime* rt, const sc_encoder_cfg* cfg, const void* weights_blob, size_t nbytes, sc_encoder** out);
sc_status sc_encoder_destroy(sc_encoder* enc);
sc_status sc_encoder_info(sc_encoder* enc, sc_encoder_cfg* cfg_out);
/* The forward has two pipelines with the same arithmetic up to rounding: batches of more than 1 024 token rows run 256 x 256-tile
 * GEMMs with every LayerNorm folded into the neighbouring GEMMs (statistics from the producing epilogue, normalisation in the
 * consuming one: no LayerNorm kernel); smaller batches -- a query -- run split-K GEMMs with stand-alone LayerNorm kernels.
 * path 0 = that rule, 1 = the batch pipeline for every size, 2 = the small-batch pipeline for every size (tests, A/B runs). */
sc_status sc_encoder_set_path(sc_encoder* enc, int32_t path);
/* Replaces Embeddings.embed_documents / embed_query after tokenisation (indexer.py:150,
 * pipeline.py:171-175): ids [B,S] int32 (S in {32,64,128,256,512}, padded by the caller), lens [B] =
 * number of real tokens per row (keys >= len are masked, pooling = mean over the first len tokens),
 * out [B,hidden] f32.  Host pointers; synchronises. */
sc_status sc_encoder_embed_ids(sc_encoder* enc, const int32_t* ids, const int32_t* lens, int32_t B, int32_t S, float* out);
/* Same with DEVICE pointers; asynchronous on the runtime's stream. */
sc_status sc_encoder_embed_ids_dev(sc_encoder* enc, const int32_t* ids_dev, const int32_t* lens_dev, int32_t B, int32_t S,
                                   float* out_dev);

/* embed_documents + upsert of one batch (indexer.py:150 -> milvus_store.py:128) without leaving the device: the
 * pooled vectors go straight from the encoder's output buffer into index rows `rows` (semantics of
 * sc_index_put_rows).  out may be NULL; if not, the vectors are also copied to the host [B,hidden].  Encoder and
 * index must belong to the same runtime and hidden == dim.  Host pointers; synchronises. */
sc_status sc_encoder_embed_ids_into(sc_encoder* enc, const int32_t* ids, const int32_t* lens, int32_t B, int32_t S, sc_index* ix,
                                    const int64_t* rows, float* out);

/* The same, pipelined: returns as soon as the batch is enqueued (ids, lens and rows are first copied into pinned staging
 * owned by the encoder, so the caller's buffers are free immediately).  Up to two batches are in flight; a third call
 * waits for the first.  The host prepares batch i+1 (tokenising, primary-key bookkeeping) while the device embeds batch i.
 * Argument errors are reported at once; a failure of the device work is reported by the call that next waits for that
 * batch (a later _async call or sc_encoder_wait). */
sc_status sc_encoder_embed_ids_into_async(sc_encoder* enc, const int32_t* ids, const int32_t* lens, int32_t B, int32_t S, sc_index* ix,
                                          const int64_t* rows);
/* Block until every batch enqueued by sc_encoder_embed_ids_into_async has finished. */
sc_status sc_encoder_wait(sc_encoder* enc);

/* ----------------------------------------------------------------- tokenizer ---- */
typedef struct sc_tokenizer sc_tokenizer;
/* Host-side WordPiece tokenizer (BERT scheme), the step the reference leaves to its provider's library (raw strings
 * are handed over at indexer.py:150).  vocab_utf8 = contents of a vocab.txt (one token per line; needs [UNK] [CLS]
 * [SEP]).  No GPU involved. */
sc_status sc_tokenizer_create(const char* vocab_utf8, size_t nbytes, int32_t lowercase, sc_tokenizer** out);
sc_status sc_tokenizer_destroy(sc_tokenizer* tok);
sc_status sc_tokenizer_info(sc_tokenizer* tok, int32_t* vocab_size, int32_t* pad, int32_t* unk, int32_t* cls, int32_t* sep);
/* texts i = bytes[offsets[i] .. offsets[i+1]) (UTF-8).  ids [n,S] ([CLS] pieces [SEP], truncated to min(max_tokens,S),
 * padded with [PAD]), lens [n].  Non-ASCII text is normalised as BERT's tokenizer does (NFD accent stripping and lower-casing for
 * uncased vocabularies, CJK ideographs spaced out, Unicode punctuation and whitespace classes, control / format characters
 * dropped; tables generated from the build image's transformers BertTokenizer, tests/golden/tokenizer_unicode.json).  Only a
 * text that is not valid UTF-8 is left alone: needs_fallback[i] = 1, lens[i] = 0.  threads <= 0: all hardware threads. */
sc_status sc_tokenizer_encode(sc_tokenizer* tok, const char* bytes, const int64_t* offsets, int32_t n, int32_t max_tokens, int32_t S,
                              int32_t* ids, int32_t* lens, uint8_t* needs_fallback, int32_t threads);

/* Diagnostics: run ONE encoder kernel on host f32 data (rounded to bf16 on device, result widened
 * back to f32) so that the parity tests can check the GEMM and the attention kernel in isolation.
 * epi: 0 = bias, 1 = bias + erf-GELU, 2 = bias + residual R [M,N]; + 16: let small shapes take the split-K path the
 * encoder uses for batches of <= 1024 tokens.  out [M,N] = A [M,K] * W [N,K]^T.
 * M, N multiples of 128, K multiple of 64. */
sc_status sc_diag_gemm_bf16(sc_runtime* rt, int32_t epi, const float* A, const float* W, const float* bias, const float* R,
                            int32_t M, int32_t N, int32_t K, float* out);
/* Times `iters` launches of one GEMM shape on device-resident synthetic data (kernel tuning; variant 0 =
 * the product kernel, other values = diagnostic ablations whose outputs are meaningless). */
sc_status sc_diag_gemm_bench(sc_runtime* rt, int32_t epi, int32_t M, int32_t N, int32_t K, int32_t iters, int32_t variant,
                             double* ms_per_launch);
/* cap_words / (8 * tiles) back-to-back traced launches of the 256x256-tile GEMM (M, N multiples of 256): out[launch][tile][8] =
 * {0: HW_ID, 1: XCC_ID, 2: entry, 3: main loop done, 4: epilogue issued, 5: stores drained}, stamps in 10 ns wall-clock
 * ticks.  Kernel tuning aid (scripts/gemm_trace.py). */
sc_status sc_diag_gemm_trace(sc_runtime* rt, int32_t epi, int32_t M, int32_t N, int32_t K, uint64_t* out, int64_t cap_words);
/* The int8 form of the 256 x 256 tile (the batched scan's int8 coarse stage) on its own: out [M,N] i32 = A [M,K] i8 * W [N,K]^T,
 * exact.  M, N multiples of 256, K multiple of 128. */
sc_status sc_diag_gemm_i8(sc_runtime* rt, const int8_t* A, const int8_t* W, int32_t M, int32_t N, int32_t K, int32_t* out);
/* Copies one workspace buffer of the encoder's last forward to the host (kernel debugging: 0 x, 1 y, 2 qkv, 3 ctx, 4 ffn hidden,
 * 5 / 6 partial row statistics, 7 / 8 finalised row statistics of the LayerNorm-folded pipeline). */
sc_status sc_diag_encoder_read(sc_encoder* enc, int32_t which, void* out, size_t nbytes);
/* Process-wide test / tuning switches, by name (results stay identical; unknown names: SC_ERR_INVALID).  "coarse_workgroups": grid
 * of the persistent coarse-scan kernel (0 = one workgroup per CU), so that tests can make a few workgroups walk many tiles;
 * "coarse_persistent": 0 = one workgroup per tile instead; "gemm_pp": main loop of the 256-tile GEMMs (-1 default, 0 = one barrier
 * per K-tile, 2..5 = ping-pong with that many half-tiles in flight); "ivf_refresh_nomem": 1 = the re-layout of a trained IVF index
 * after upserts fails as if the device were full (the search must then answer exhaustively instead of failing); "tighten": 0 = the batched scan's thresholds stay the kp-th best coarse keys
 * (no exact re-score of the 128 best before the large phases); "ivf_tail_rows": how many rows appended to a trained IVF_FLAT index may
 * stay behind its lists as a tail that probes scan exactly (-1 = default 65536; 0 = fold appended rows into the lists before every search);
 * "wide_candidates": 1 = the int8 stage keeps every key within its cut
 * (the form it otherwise switches to on corpora whose certificate fails) wherever it can; "collect_pass": 0 = queries a
 * coarse stage cannot certify go straight to the next stage (no collect pass); "ivf_coarse_nomem": 1 = the IVF coarse stage
 * cannot allocate its centred shadow (the search must then probe exactly instead of failing); "ivf_refine_cap":
 * rows per query the IVF coarse stage's refine step takes on (-1 = default 4096; a small value sends queries to the exact re-probe). */
sc_status sc_diag_set_option(const char* name, int32_t value);
/* qkv [B*S, 3*heads*64] rows = [Q | K | V]; lens [B]; out [B*S, heads*64] = softmax(QK^T/8 + mask) V. */
sc_status sc_diag_attention(sc_runtime* rt, const float* qkv, const int32_t* lens, int32_t B, int32_t S, int32_t heads, float* out);

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
/* Upsert in one call: rows[i] <- vecs[i] where rows[i] is either an existing row (replace) or the next free row
 * (append; new rows must be numbered old_rows, old_rows+1, ... in the order they appear).  Row numbers must be
 * distinct.  One Collection.upsert batch (milvus_store.py:119-130) without the add/overwrite split. */
sc_status sc_index_put_rows(sc_index* ix, const float* vecs, const int64_t* rows, int64_t n);
/* Same with vecs a DEVICE pointer ([n,dim] f32, tight); rows stays a host pointer.  The copy of the vectors is enqueued on
 * the runtime's stream (the embed -> store hand-over without a trip through host memory); the call itself
 * waits for the stream before returning, because `rows` is the caller's pageable memory. */
sc_status sc_index_put_rows_dev(sc_index* ix, const float* vecs_dev, const int64_t* rows, int64_t n);
/* Copy rows [first, first+n) back to the host as [n,dim] (persistence, tests). */
sc_status sc_index_get_rows(sc_index* ix, int64_t first, int64_t n, float* out);
/* Resize to n rows and fill them on device with sc_synth_fill_dev(seed, first_row). */
sc_status sc_index_fill_synthetic(sc_index* ix, int64_t n, uint64_t seed, int64_t first_row);

/* Clustered synthetic corpus (IVF recall benchmarks): row = centre[hash(row) % nclusters] + spread * noise, all
 * from the same integer-hash generator. */
sc_status sc_index_fill_synthetic_clustered(sc_index* ix, int64_t n, uint64_t seed, int64_t first_row, int32_t nclusters, float spread);
/* Free the rebuildable device buffers of an index (bf16 shadow, search scratch). */
sc_status sc_index_release_scratch(sc_index* ix);

/* Replaces Collection.search(data=[vector], param={metric, nprobe}, limit=top_k)
 * (milvus_store.py:141-147), batched: q [Q,dim] host, out_dist [Q,k] f32, out_rows [Q,k] i64
 * (global ids, best first; ties broken by lower row id; missing hits = -1 / +inf-or--inf).
 * nprobe is ignored by FLAT indexes.  Distances: L2 = squared L2, IP = dot, COSINE = cosine. */
sc_status sc_index_search(sc_index* ix, const float* q, int32_t Q, int32_t k, int32_t nprobe,
                          float* out_dist, int64_t* out_rows);
/* Same with DEVICE pointers (q row stride = dim): enqueued on the runtime's stream.  The exact scan (<= 16 queries, no IVF
 * probe) returns without waiting.  The batched path (bf16 coarse + f32 re-rank) reads the per-query certificate flags back and
 * so synchronises the stream once per call, and list-major IVF probing synchronises twice (probe ids to the host planner, plan
 * tables alive until the scan has run); the per-index mutex is held meanwhile.  Results are complete once the stream has
 * passed the call in either case. */
sc_status sc_index_search_dev(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe,
                              float* out_dist_dev, int64_t* out_rows_dev);

/* Replaces Collection.create_index(IVF_FLAT, nlist) + load() (milvus_store.py:76-84) for an index created with
 * SC_INDEX_IVF_FLAT: deterministic k-means (niter Lloyd iterations on <= 256*nlist sampled rows), assignment of
 * every row to its nearest centroid, list-major re-ordering of the corpus in HBM.  Until it is called an IVF_FLAT index
 * answers with the exhaustive scan.  After it, searches with Q * nprobe < nlist probe only the nprobe nearest lists
 * (approximate, like the reference); larger batches probe list-major or, where that is estimated to be cheaper, keep using the
 * exhaustive paths, whose results are a superset in quality.
 * Upserts into a trained index (sc_index_add / _overwrite / _put_rows*) KEEP the lists -- Collection.upsert into an indexed
 * collection does not retrain either (milvus_store.py:128): the next search first assigns the new and the replaced rows to the
 * existing centroids and re-orders the corpus once (one pass over it, no k-means; the result is exactly what
 * sc_index_assign_lists builds from scratch for these centroids).  k-means runs again only when this function is called. */
sc_status sc_index_train(sc_index* ix, int32_t niter, uint64_t seed);
/* Persistence of a trained index (the `ivf.*` files of the on-disk collection): the list of every row in insertion order
 * (out [rows] int32), and the inverse -- install centroids [nlist, dim] + that list without running k-means. */
sc_status sc_index_ivf_assignments(sc_index* ix, int32_t* out);
sc_status sc_index_set_ivf(sc_index* ix, const float* centroids, const int32_t* assign, int32_t nlist);
/* Build the lists for given centroids [nlist, dim] without k-means (every row to its nearest centroid).  Multi-GPU IVF_FLAT: one
 * rank trains and broadcasts its centroids, every rank calls this on its shard; the merged probe results then equal those of one
 * index over the whole corpus with these centroids. */
sc_status sc_index_assign_lists(sc_index* ix, const float* centroids, int32_t nlist);
/* nlist actually trained (0 = untrained), centroids [nlist, dim] and list sizes [nlist] (either may be NULL). */
sc_status sc_index_ivf_info(sc_index* ix, int32_t* nlist, float* centroids, int64_t* list_sizes);

/* Search path selection.  mode 0 (default): exact f32 scan for <= 16 queries, MFMA coarse scan (int8, then bf16: see
 * sc_index_set_coarse_stage) + exact f32 re-rank + certificate (uncertified queries re-run exactly) for larger batches with
 * k <= 64 -- both return identical results; 1: exact scan only; 2: batched path whenever supported;
 * 3: per-query IVF probing whenever the index is trained (any batch size; used to measure recall);
 * 4: list-major IVF probing (the probed lists are streamed once per group of queries that want them; same results as 3).
 * In mode 0 a trained IVF_FLAT index probes per query while Q * nprobe < nlist, list-major while that is estimated to be
 * cheaper than the exhaustive paths, and otherwise answers exhaustively (exact results). */
sc_status sc_index_set_search_mode(sc_index* ix, int32_t mode);
/* After a search: which path ran (1 exact, 2 batched, 3 ivf probe per query, 4 ivf probe list-major) and how many queries the batched path had to
 * re-run through the exact scan because their certificate failed. */
sc_status sc_index_last_search_stats(sc_index* ix, int32_t* path, int32_t* uncertified);

/* Coarse stage of the batched path.  0 (default): the int8 stage first (v_mfma_i32_16x16x64_i8 at twice the bf16 rate on an int8
 * shadow with per-row scales, 512 candidates per query); queries whose certificate fails there go to the bf16 stage (128
 * candidates), and only what fails there too to the exact scan -- results are identical whichever stage answers.  8 / 16 pin the
 * first stage (8: no bf16 stage in between).  An index whose int8 stage could not certify most of a batch starts at the bf16
 * stage from then on (tightly clustered corpora); setting 0 again clears that.  Under 0 the int8 stage is only taken from 2^20
 * rows and 129 queries up (below that its four times larger re-rank, and its 256-query tiles, cost more than the coarse pass
 * saves) or under search mode 2. */
sc_status sc_index_set_coarse_stage(sc_index* ix, int32_t bits);
/* After a batched search: the stage it started on (8 / 16) and how many queries the int8 stage handed to the bf16 stage
 * (sc_index_last_search_stats' `uncertified` counts the queries that ended in the exact scan). */
sc_status sc_index_last_coarse_stats(sc_index* ix, int32_t* first_stage_bits, int32_t* handed_to_bf16);
/* Collect passes of the last batched search: queries whose certificate failed at a stage are given a second pass at the same
 * precision with a fixed threshold (k-th exact score found + the coarse error bound) whose survivors are ALL re-scored exactly;
 * `tried` = such queries (summed over the stages), `resolved` = those it answered (the rest went on to the next stage). */
sc_status sc_index_last_collect_stats(sc_index* ix, int32_t* tried, int32_t* resolved);
/* 1 if the last batched search ran the int8 stage in its wide form: every key within the exact-score cut kept between the phases
 * (up to 4 096 per query) instead of the 512 best -- what the stage switches to on corpora whose certificate fails (clusters). */
sc_status sc_index_last_wide(sc_index* ix, int32_t* wide);
/* Rows the last search scanned exactly BEHIND the lists of a trained IVF_FLAT index: rows appended since the lists were laid out stay
 * there (up to 65 536) instead of forcing a re-layout before the next search -- Milvus' brute-force search of its growing segment;
 * 0 = the lists covered every stored row. */
sc_status sc_index_last_tail_rows(sc_index* ix, int64_t* rows);

/* After an IVF probe search: rows of the DISTINCT lists the batch probed (`unique_rows`: the algorithmic bytes of SURVEY.md 8d
 * config 5 = unique_rows * ld * 4), rows the scan kernel streamed (`streamed_rows`: list-major probing streams a list once per
 * group of <= 16 queries that want it; per-query probing once per query) and the number of (list part, query group) work items. */
sc_status sc_index_last_probe_stats(sc_index* ix, int64_t* unique_rows, int64_t* streamed_rows, int32_t* groups);

/* Multi-GPU final step (one process per GPU): merge `lists` per-shard results
 * dist [lists,Q,k] / rows [lists,Q,k] (as produced by sc_index_search* on each shard and
 * all-gathered over RCCL) into the global best-first [Q,k], same tie rule.  Host buffers. */
sc_status sc_topk_merge_host(sc_metric metric, int32_t lists, int32_t Q, int32_t k,
                             const float* dist, const int64_t* rows, float* out_dist, int64_t* out_rows);

/* -------------------------------------------------------------- communicator ---- */

/* The path's collectives, RCCL over xGMI, one process per GPU.  Not in the reference (a single Milvus
 * server); this is north_star's multi-GPU scheme: the corpus is sharded by row range, every rank searches the same queries on
 * its shard (sc_index created with row_base = shard start), ONE all-gather moves the per-shard [Q,k] results to every rank,
 * sc_topk_merge_host merges them with the (distance, lower row id) rule.  For IVF_FLAT one rank trains, broadcasts its
 * centroids, and every rank builds its lists for them (sc_index_assign_lists).  librccl is bound at run time on first use.
 * All calls enqueue on the communicator's runtime stream; every rank must make the same calls in the same order. */
typedef struct sc_comm sc_comm;
#define SC_COMM_ID_BYTES 128
/* Rank 0 obtains the rendezvous id (ncclGetUniqueId) and hands the 128 bytes to the other ranks out of band (an environment
 * variable, a file, the launcher's store: control plane, not part of this library). */
sc_status sc_comm_unique_id(void* id_out, size_t nbytes);
/* Collective over all `world` ranks (ncclCommInitRank) on rt's device. */
sc_status sc_comm_create(sc_runtime* rt, int32_t rank, int32_t world, const void* unique_id, size_t nbytes, sc_comm** out);
sc_status sc_comm_destroy(sc_comm* comm);
sc_status sc_comm_info(sc_comm* comm, int32_t* rank, int32_t* world);
/* ncclGetVersion of the RCCL copy this process bound (0 when that copy lacks the entry point). */
sc_status sc_comm_rccl_version(int32_t* version);
/* The search path's one exchange step: dist_dev / rows_dev [Q,k] of this rank (as written by sc_index_search_dev) ->
 * all_dist_dev / all_rows_dev [world,Q,k] on every rank (DEVICE pointers; asynchronous on the runtime's stream; both arrays
 * travel in one grouped RCCL launch). */
sc_status sc_comm_allgather_topk(sc_comm* comm, const float* dist_dev, const int64_t* rows_dev, int32_t Q, int32_t k,
                                 float* all_dist_dev, int64_t* all_rows_dev);
/* The IVF build's one collective: nbytes at DEVICE pointer buf_dev from rank `root` to every rank (asynchronous). */
sc_status sc_comm_broadcast(sc_comm* comm, void* buf_dev, size_t nbytes, int32_t root);
/* Row-sharded search in one call per rank (every rank passes the same queries): this rank's shard is searched as by
 * sc_index_search, the per-shard [Q,k] results are all-gathered and merged on the host; out_dist / out_rows [Q,k] (global row
 * ids) are identical on every rank and equal to the result of one index over the whole corpus.  Host pointers; synchronises. */
sc_status sc_index_search_sharded(sc_index* ix, sc_comm* comm, const float* q, int32_t Q, int32_t k, int32_t nprobe,
                                  float* out_dist, int64_t* out_rows);
/* The same up to the exchange, with DEVICE pointers: q_dev [Q,dim]; all_dist_dev / all_rows_dev [world,Q,k] receive every
 * shard's result (this rank's is written in place into its own slot); enqueued on the runtime's stream like
 * sc_index_search_dev.  The caller merges (sc_topk_merge_host) where it needs the result. */
sc_status sc_index_search_sharded_dev(sc_index* ix, sc_comm* comm, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe,
                                      float* all_dist_dev, int64_t* all_rows_dev);
/* IVF_FLAT over a sharded collection: rank `root` runs sc_index_train on ITS shard, the centroids are broadcast, every other
 * rank builds its lists for them (sc_index_assign_lists).  Probing then means the same lists on every shard. */
sc_status sc_index_train_sharded(sc_index* ix, sc_comm* comm, int32_t niter, int32_t root);
/* max over the ranks of a host double (in place); synchronises, so it also serves as the barrier that brackets a timed
 * region (bench.py). */
sc_status sc_comm_allreduce_max(sc_comm* comm, double* value);

#ifdef __cplusplus
}
#endif
#endif /* SEMCODE_HIP_H */

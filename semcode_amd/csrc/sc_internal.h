// sc_internal.h -- handle structs shared by the host-side translation units.
#pragma once
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

#include "sc_common.h"

#define SC_PROF_SCAN 0
#define SC_PROF_MERGE 1
#define SC_PROF_GEMM 2
#define SC_PROF_ATTN 3
#define SC_PROF_CLASSES 4

sc_status sc_fail(sc_status code, const char* fmt, ...);

#define SC_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return sc_fail(SC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

struct sc_runtime {
    // Lifetime: the creator holds one reference, every index / encoder / communicator created on the runtime holds one more.
    // sc_runtime_destroy drops the creator's; the stream and the struct go when the LAST reference does, so handles may be
    // destroyed in any order (a garbage-collected Index after Runtime.close(): DESIGN.md section 9).
    std::atomic<int> refs{1};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int profiling = 0;                            // 0 off; n >= 1: bracket every n-th launch of the encoder classes (all launches of the scan classes)
    unsigned prof_seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // launches seen per class since the last reset
    char name[256] = {0};
    int cus = 256;
    int64_t hbm = 0;
    std::mutex mu;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof[SC_PROF_CLASSES];
};

void sc_runtime_retain(sc_runtime* rt);
void sc_runtime_release(sc_runtime* rt);  // frees the runtime when the last reference is dropped
void sc_prof_begin(sc_runtime* rt, int which, hipEvent_t* a, hipEvent_t* b);
void sc_prof_end(sc_runtime* rt, int which, hipEvent_t a, hipEvent_t b);

struct sc_index {
    sc_runtime* rt = nullptr;
    int dim = 0, ld = 0;
    sc_metric metric = SC_METRIC_IP;
    sc_index_kind kind = SC_INDEX_FLAT;
    int nlist = 0;
    int64_t row_base = 0;
    int64_t n = 0, capacity = 0;
    float* X = nullptr;      // [capacity, ld]
    float* xnorm = nullptr;  // [capacity]
    bool trained = false;
    // scratch (grown on demand)
    void* stage = nullptr;   size_t stage_cap = 0;
    float* qpad = nullptr;   size_t qpad_cap = 0;
    float* qnorm = nullptr;  size_t qnorm_cap = 0;
    uint64_t* partial = nullptr; size_t partial_cap = 0;
    void* io = nullptr;      size_t io_cap = 0;
    // batched path: bf16 shadow of X (rows padded to 128), per-batch scratch
    void* Xb = nullptr;      size_t xb_cap = 0;   // bytes
    int64_t shadow_rows = 0;                      // rows [0, shadow_rows) of Xb are valid
    unsigned* xnorm_max = nullptr;                // device: bits of max |x|^2
    // int8 coarse stage: int8 shadow (rows of ld8 = ld rounded up to 128 bytes, rows padded to 256) + per-row scale
    void* Xq = nullptr;      size_t xq_cap = 0;   // bytes
    float* xscale = nullptr; size_t xscale_cap = 0;
    int64_t shadow8_rows = 0;
    unsigned* xnorm_max8 = nullptr;               // device: bits of {max |x|^2, max |x - s q|^2, max relative}
    int coarse_mode = 0;                          // sc_index_set_coarse_stage: 0 auto (int8 first), 8 int8 only, 16 bf16 only
    bool i8_off = false;                          // the int8 certificate failed for most of a batch on this corpus: use the bf16 stage
    int last_coarse_bits = 0;                     // 8 / 16: coarse stage of the last batched search
    int last_uncert_i8 = 0;                       // queries the int8 stage handed on to the bf16 stage
    bool wide_i8 = false;                         // ... or rather: the int8 stage keeps EVERY key within its exact-score cut (up to 4 096 per query) -- tried before i8_off
    int last_wide = 0;                            // the last batched search ran the wide form
    bool i8_sticky = false;                       // ... but the bf16-first batch that followed cost more: int8 first from now on (search_batched_locked)
    double cost_i8_first = 0.0;                   // seconds per query of the batch that switched the int8 stage off (0 = none pending)
    bool collect_off8 = false, collect_off16 = false;  // the collect pass of that stage resolved less than half of a sub-batch: skip it on this corpus
    int last_collect_tried = 0, last_collect_resolved = 0;  // queries of the last batched search that went through a collect pass / that it answered
    void* bscratch = nullptr; size_t bscratch_cap = 0;
    void* fb = nullptr;      size_t fb_cap = 0;   // fallback staging (queries + results) of the first stage's uncertified queries
    void* fb2 = nullptr;     size_t fb2_cap = 0;
    void* tailbuf = nullptr; size_t tailbuf_cap = 0;  // two [Q][k] result sets of a search that answers from the lists and from the tail (sc_api.cpp)
    int64_t last_tail_rows = 0;                   // rows the last search scanned behind the lists (0: none)  // the same for the second stage (int8 -> bf16 -> exact)
    // IVF_FLAT (after sc_index_train): X / xnorm are stored list-major
    sc_index* quant = nullptr;                    // flat index over the nlist centroids (coarse quantizer)
    uint32_t* perm = nullptr;                     // device [ivf_rows]: stored position -> row id (insertion order)
    int64_t* list_off = nullptr;                  // device [nlist + 1]
    std::vector<uint32_t> inv_h;                  // host [ivf_rows]: row id -> stored position (get_rows / overwrite / re-layout)
    std::vector<int64_t> list_off_h;
    std::vector<int32_t> assign_h;                // host [ivf_rows]: list of every row in the lists (persistence, incremental upserts)
    int nlist_trained = 0;
    // Upserts into a trained index do not drop the lists (Milvus: upsert into an indexed collection, milvus_store.py:128).
    // Rows [0, ivf_rows) sit list-major at inv_h[row]; rows appended since sit behind them at position == row id; rows
    // overwritten in place since are listed in dirty_rows (their list may have changed).  The next search first assigns the
    // pending + dirty rows to the EXISTING centroids and re-orders the corpus once (sc_ivf_refresh_locked): no k-means.
    int64_t ivf_rows = 0;
    int64_t perm_rows = 0;                        // entries of `perm` (== ivf_rows unless sc_ivf_cover_tail_locked extended it)
    std::vector<int64_t> dirty_rows;
    void* ivf_scratch = nullptr; size_t ivf_scratch_cap = 0;
    // int8 coarse stage of list-major probing (L2; ivf_coarse.hip): every list quantised relative to its centroid
    void* Xc8 = nullptr;     size_t xc8_cap = 0;  // [ivf_rows padded to 256][ld8] int8 of x - c_list
    float* xcs = nullptr;    size_t xcsn_cap = 0; // [rows][4] f32 per row: {|x - c_list|^2, int8 scale, 2 |dx|, 2 (|x'| + |dx|)}
    unsigned* list_stats = nullptr;               // [nlist][2] bits of {max |x' - xq|^2, max |x'|^2} + [4] bits of max |x|^2 behind them
    // Rows overwritten in place since a shadow was built (stored positions below the shadow's row count): the next search re-builds the
    // shadow rows of exactly these instead of the whole shadow (8 ms int8 / 54 ms centred at 10M x 768 for one upsert batch of 128:
    // scripts/upsert_search_interleave.py).  The shadows' running maxima only grow in between, which keeps every bound valid; a
    // full rebuild (layout change, more than SC_SHADOW_DIRTY_MAX rows) clears the lists.
    std::vector<int64_t> dirty_b16, dirty_i8, dirty_c8;
    static constexpr int64_t SC_SHADOW_DIRTY_MAX = 8192;
    int64_t shadowc_rows = 0;                     // rows covered by the centred shadow (== ivf_rows when valid)
    void* ivfc_scratch = nullptr; size_t ivfc_scratch_cap = 0;
    bool ivfc_off = false;                        // the coarse stage left most of a batch uncertified on this index: probe exactly
    int last_ivfc_uncertified = 0;
    int last_probed_lists = 0;
    int64_t last_unique_rows = 0, last_streamed_rows = 0;  // sc_index_last_probe_stats
    int last_groups = 0;
    int search_mode = 0;                          // 0 auto, 1 exact only, 2 batched whenever supported, 3 / 4 IVF probe per query / list-major whenever trained
    int last_path = 0;                            // 1 exact, 2 batched, 3 ivf probe per query, 4 ivf probe list-major
    int last_uncertified = 0;
    double uncert_frac = -1.0;                    // share of queries the last batched exhaustive search had to re-run exactly (-1 = never ran)
    std::mutex mu;
};

// sc_api.cpp internals used by sc_ivf.cpp (caller holds ix->mu)
sc_status sc_grow(sc_index* ix, void** p, size_t* cap, size_t need);
sc_status sc_search_dev_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe, float* out_dist, int64_t* out_rows);
sc_status sc_search_flat_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, float* out_dist, int64_t* out_rows);
sc_status sc_ivf_search_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe, float* out_dist, int64_t* out_rows);
bool sc_ivf_listmajor_applicable(const sc_index* ix, int Q, int k, int nprobe, bool flat_is_batched);
sc_status sc_ivf_search_listmajor_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe, float* out_dist, int64_t* out_rows);
bool sc_ivf_coarse_applicable(const sc_index* ix, int Q, int k, int nprobe);
sc_status sc_ivf_search_coarse_locked(sc_index* ix, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe, float* out_dist, int64_t* out_rows);
sc_status sc_ivf_untrain_locked(sc_index* ix);  // restore insertion order, drop lists
void sc_ivf_drop_lists_locked(sc_index* ix);    // drop lists without restoring the order (the rows are about to be discarded)
sc_status sc_ivf_cover_tail_locked(sc_index* ix);  // extend perm over rows appended since the build (identity): exhaustive search only
sc_status sc_ivf_refresh_locked(sc_index* ix, bool keep_tail = false);  // fold rows upserted since the lists were built into them (no k-means);
                                                                        // keep_tail: only settle the overwritten rows -- if none left its list, appended rows stay a tail
// stored position of row `r` (trained layout installed: ix->perm != nullptr)
static inline int64_t sc_ivf_pos(const sc_index* ix, int64_t r) { return r < ix->ivf_rows ? (int64_t)ix->inv_h[(size_t)r] : r; }
// upsert body shared by sc_index_put_rows{,_dev} and sc_encoder_embed_ids_into; caller holds ix->mu and has set the device
sc_status sc_index_put_rows_locked(sc_index* ix, const float* vecs, bool vecs_on_device, const int64_t* rows, int64_t n, const char* who);
bool sc_ivf_applicable(const sc_index* ix, int Q, int nprobe);

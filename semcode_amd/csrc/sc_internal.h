// sc_internal.h -- handle structs shared by the host-side translation units.
#pragma once
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
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool profiling = false;
    char name[256] = {0};
    int cus = 256;
    int64_t hbm = 0;
    std::mutex mu;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof[SC_PROF_CLASSES];
};

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
    void* bscratch = nullptr; size_t bscratch_cap = 0;
    void* fb = nullptr;      size_t fb_cap = 0;   // fallback staging (queries + results)
    int search_mode = 0;                          // 0 auto, 1 exact only, 2 batched whenever supported
    int last_path = 0;                            // 1 exact, 2 batched
    int last_uncertified = 0;
    std::mutex mu;
};

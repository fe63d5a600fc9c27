// sc_comm.cpp -- the path's collectives over RCCL (xGMI), behind the C ABI (include/semcode_hip.h, "communicator").
//
// Not in the reference (one Milvus server, no parallelism: SURVEY.md section 2).  north_star's multi-GPU scheme has exactly
// one exchange step on the search path -- the all-gather of the per-shard top-k ([Q, k] f32 distances + i64 global row ids
// per rank; 122 880 B per rank at Q = 1024, k = 10: latency-bound) -- and one in the IVF build, the broadcast of the trained
// centroids (50 MB at 4096 x 3072).  Both are issued here, on the runtime's stream, so that the search kernels, the collective
// and the copy to the host merge are ordered by the stream and no framework sits on the serving path.
//
// librccl is bound at run time (dlopen): a single-GPU deployment never loads it (573 MB), and a process that already holds
// a copy under the same soname -- PyTorch wheels bundle one, next to their own HIP runtime -- keeps using that copy, which
// is the one built against the HIP runtime serving the process.
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include <rccl/rccl.h>  // types and prototypes only: every entry point is resolved with dlsym

#include "sc_internal.h"

namespace {
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;  // optional
    const char* error = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.error = "librccl.so.1 not found (dlopen)";
            return;
        }
#define SC_SYM(field, sym)                                         \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym)); \
    if (!r.field) r.error = "librccl lacks " sym;
        SC_SYM(GetUniqueId, "ncclGetUniqueId")
        SC_SYM(CommInitRank, "ncclCommInitRank")
        SC_SYM(CommDestroy, "ncclCommDestroy")
        SC_SYM(AllGather, "ncclAllGather")
        SC_SYM(Broadcast, "ncclBroadcast")
        SC_SYM(AllReduce, "ncclAllReduce")
        SC_SYM(GroupStart, "ncclGroupStart")
        SC_SYM(GroupEnd, "ncclGroupEnd")
        SC_SYM(GetErrorString, "ncclGetErrorString")
#undef SC_SYM
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(r.handle, "ncclGetVersion"));
    });
    return &r;
}
}  // namespace

struct sc_comm {
    sc_runtime* rt = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double* scratch = nullptr;  // device: one double for sc_comm_allreduce_max
    int64_t* xchg = nullptr;    // device: [world][2] status words exchanged next to every collective of the sharded calls
    std::mutex mu;
};

#define SC_NCCL(expr)                                                                                       \
    do {                                                                                                    \
        ncclResult_t _r = (expr);                                                                           \
        if (_r != ncclSuccess) return sc_fail(SC_ERR_HIP, "%s failed: %s", #expr, rccl()->GetErrorString(_r)); \
    } while (0)

static sc_status need_rccl() {
    Rccl* r = rccl();
    if (r->error) return sc_fail(SC_ERR_UNSUPPORTED, "RCCL is not available: %s", r->error);
    return SC_OK;
}

extern "C" sc_status sc_comm_unique_id(void* id_out, size_t nbytes) {
    if (!id_out || nbytes != SC_COMM_ID_BYTES) return sc_fail(SC_ERR_INVALID, "sc_comm_unique_id: need a %d-byte buffer", SC_COMM_ID_BYTES);
    static_assert(sizeof(ncclUniqueId) == SC_COMM_ID_BYTES, "ncclUniqueId size");
    sc_status st = need_rccl();
    if (st) return st;
    ncclUniqueId id;
    SC_NCCL(rccl()->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return SC_OK;
}

extern "C" sc_status sc_comm_create(sc_runtime* rt, int32_t rank, int32_t world, const void* unique_id, size_t nbytes, sc_comm** out) {
    if (!rt || !out || !unique_id) return sc_fail(SC_ERR_INVALID, "sc_comm_create: NULL argument");
    *out = nullptr;
    if (nbytes != SC_COMM_ID_BYTES) return sc_fail(SC_ERR_INVALID, "sc_comm_create: the unique id is %d bytes", SC_COMM_ID_BYTES);
    if (world < 1 || rank < 0 || rank >= world) return sc_fail(SC_ERR_INVALID, "sc_comm_create: rank %d outside [0,%d)", rank, world);
    sc_status st = need_rccl();
    if (st) return st;
    SC_HIP(hipSetDevice(rt->device));
    sc_comm* c = new (std::nothrow) sc_comm();
    if (!c) return sc_fail(SC_ERR_NOMEM, "out of host memory");
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclResult_t r = rccl()->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return sc_fail(SC_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl()->GetErrorString(r));
    }
    if (hipMalloc((void**)&c->scratch, 16) != hipSuccess || hipMalloc((void**)&c->xchg, (size_t)world * 16) != hipSuccess) {
        hipFree(c->scratch);
        rccl()->CommDestroy(c->comm);
        delete c;
        return sc_fail(SC_ERR_NOMEM, "sc_comm_create: hipMalloc failed");
    }
    c->rt = rt;
    sc_runtime_retain(rt);
    c->rank = rank;
    c->world = world;
    *out = c;
    return SC_OK;
}

extern "C" sc_status sc_comm_destroy(sc_comm* c) {
    if (!c) return SC_OK;
    hipSetDevice(c->rt->device);
    hipStreamSynchronize(c->rt->stream);
    if (c->comm) rccl()->CommDestroy(c->comm);
    hipFree(c->scratch);
    hipFree(c->xchg);
    sc_runtime* rt = c->rt;
    delete c;
    sc_runtime_release(rt);
    return SC_OK;
}

extern "C" sc_status sc_comm_rccl_version(int32_t* version) {
    if (!version) return sc_fail(SC_ERR_INVALID, "sc_comm_rccl_version: NULL argument");
    sc_status st = need_rccl();
    if (st) return st;
    int v = 0;
    if (rccl()->GetVersion) SC_NCCL(rccl()->GetVersion(&v));
    *version = v;
    return SC_OK;
}

extern "C" sc_status sc_comm_info(sc_comm* c, int32_t* rank, int32_t* world) {
    if (!c) return sc_fail(SC_ERR_INVALID, "communicator is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return SC_OK;
}

extern "C" sc_status sc_comm_allgather_topk(sc_comm* c, const float* dist_dev, const int64_t* rows_dev, int32_t Q, int32_t k, float* all_dist_dev,
                                            int64_t* all_rows_dev) {
    if (!c || !dist_dev || !rows_dev || !all_dist_dev || !all_rows_dev || Q < 1 || k < 1)
        return sc_fail(SC_ERR_INVALID, "sc_comm_allgather_topk: bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    SC_HIP(hipSetDevice(c->rt->device));
    const size_t n = (size_t)Q * k;
    // both arrays in ONE group = one fused launch: the exchange is latency-bound (distributed backend row)
    SC_NCCL(rccl()->GroupStart());
    ncclResult_t r1 = rccl()->AllGather(dist_dev, all_dist_dev, n, ncclFloat32, c->comm, c->rt->stream);
    ncclResult_t r2 = rccl()->AllGather(rows_dev, all_rows_dev, n, ncclInt64, c->comm, c->rt->stream);
    ncclResult_t r3 = rccl()->GroupEnd();
    SC_NCCL(r1);
    SC_NCCL(r2);
    SC_NCCL(r3);
    return SC_OK;
}

extern "C" sc_status sc_comm_broadcast(sc_comm* c, void* buf_dev, size_t nbytes, int32_t root) {
    if (!c || (!buf_dev && nbytes) || root < 0 || root >= c->world) return sc_fail(SC_ERR_INVALID, "sc_comm_broadcast: bad argument");
    if (nbytes == 0) return SC_OK;
    std::lock_guard<std::mutex> g(c->mu);
    SC_HIP(hipSetDevice(c->rt->device));
    SC_NCCL(rccl()->Broadcast(buf_dev, buf_dev, nbytes, ncclUint8, root, c->comm, c->rt->stream));
    return SC_OK;
}

extern "C" sc_status sc_comm_allreduce_max(sc_comm* c, double* value) {
    if (!c || !value) return sc_fail(SC_ERR_INVALID, "sc_comm_allreduce_max: bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    SC_HIP(hipSetDevice(c->rt->device));
    hipStream_t s = c->rt->stream;
    SC_HIP(hipMemcpyAsync(c->scratch, value, 8, hipMemcpyHostToDevice, s));
    SC_NCCL(rccl()->AllReduce(c->scratch, c->scratch, 1, ncclFloat64, ncclMax, c->comm, s));
    SC_HIP(hipMemcpyAsync(value, c->scratch, 8, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    return SC_OK;
}

// ------------------------------------------------------------------ the sharded search / build, one C-ABI call per rank

// A rank whose local step failed must still take part in the collective, or every other rank waits in it forever.  So each
// sharded call exchanges a status word per rank IN THE SAME grouped launch as its payload, a failed rank sends sentinel rows
// (worst distance, row -1: they lose every merge), and all ranks return the worst status after the exchange.
// xchg[r] = {status of rank r, aux}.  The caller holds c->mu.
static sc_status exchange_status_unlocked(sc_comm* c, int64_t status, int64_t aux, std::vector<int64_t>& all, bool in_group) {
    hipStream_t s = c->rt->stream;
    const int64_t mine[2] = {status, aux};
    SC_HIP(hipMemcpyAsync(c->xchg + 2 * (size_t)c->rank, mine, 16, hipMemcpyHostToDevice, s));
    SC_HIP(hipStreamSynchronize(s));  // `mine` is on this stack frame
    ncclResult_t r = rccl()->AllGather(c->xchg + 2 * (size_t)c->rank, c->xchg, 2, ncclInt64, c->comm, s);
    if (in_group) {  // the caller closes the group
        SC_NCCL(r);
        return SC_OK;
    }
    SC_NCCL(r);
    all.resize(2 * (size_t)c->world);
    SC_HIP(hipMemcpyAsync(all.data(), c->xchg, all.size() * 8, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    return SC_OK;
}
static sc_status worst_status(sc_comm* c, const std::vector<int64_t>& all, sc_status mine, const char* what) {
    for (int r = 0; r < c->world; ++r)
        if (all[2 * (size_t)r] != 0) {
            if (mine) return mine;  // this rank's own error text is already set
            return sc_fail(SC_ERR_STATE, "%s: rank %d failed with status %lld", what, r, (long long)all[2 * (size_t)r]);
        }
    return SC_OK;
}

// local top-k -> this rank's slot (or sentinels if the local search failed) -> grouped all-gather of distances, rows and status
static sc_status search_and_gather_locked(sc_index* ix, sc_comm* c, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe, float* all_d,
                                          int64_t* all_r, const char* what) {
    hipStream_t s = ix->rt->stream;
    const size_t n = (size_t)Q * k;
    float* my_d = all_d + (size_t)c->rank * n;  // in-place all-gather: this rank's [Q,k] goes straight into its slot
    int64_t* my_r = all_r + (size_t)c->rank * n;
    const sc_status st_local = sc_search_dev_locked(ix, q_dev, Q, k, nprobe, my_d, my_r);
    if (st_local) {  // sentinels: the worst score under this metric and row -1
        const float worst = ix->metric == SC_METRIC_L2 ? __builtin_inff() : -__builtin_inff();
        int bits;
        memcpy(&bits, &worst, 4);
        (void)hipGetLastError();
        SC_HIP(hipMemsetD32Async((hipDeviceptr_t)my_d, bits, n, s));
        SC_HIP(hipMemsetAsync(my_r, 0xFF, n * 8, s));
    }
    std::lock_guard<std::mutex> gc(c->mu);
    std::vector<int64_t> all;
    const int64_t mine[2] = {st_local, 0};
    SC_HIP(hipMemcpyAsync(c->xchg + 2 * (size_t)c->rank, mine, 16, hipMemcpyHostToDevice, s));
    SC_HIP(hipStreamSynchronize(s));
    SC_NCCL(rccl()->GroupStart());
    ncclResult_t r1 = rccl()->AllGather(my_d, all_d, n, ncclFloat32, c->comm, s);
    ncclResult_t r2 = rccl()->AllGather(my_r, all_r, n, ncclInt64, c->comm, s);
    ncclResult_t r3 = rccl()->AllGather(c->xchg + 2 * (size_t)c->rank, c->xchg, 2, ncclInt64, c->comm, s);
    ncclResult_t r4 = rccl()->GroupEnd();
    SC_NCCL(r1);
    SC_NCCL(r2);
    SC_NCCL(r3);
    SC_NCCL(r4);
    all.resize(2 * (size_t)c->world);
    SC_HIP(hipMemcpyAsync(all.data(), c->xchg, all.size() * 8, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    return worst_status(c, all, st_local, what);
}

extern "C" sc_status sc_index_search_sharded_dev(sc_index* ix, sc_comm* c, const float* q_dev, int32_t Q, int32_t k, int32_t nprobe,
                                                 float* all_dist_dev, int64_t* all_rows_dev) {
    if (!ix || !c || !q_dev || !all_dist_dev || !all_rows_dev) return sc_fail(SC_ERR_INVALID, "sc_index_search_sharded_dev: NULL argument");
    if (Q < 1 || k < 1) return sc_fail(SC_ERR_INVALID, "sc_index_search_sharded_dev: Q and top_k must be >= 1");
    if (ix->rt != c->rt) return sc_fail(SC_ERR_INVALID, "sc_index_search_sharded_dev: index and communicator belong to different runtimes");
    std::lock_guard<std::mutex> g(ix->mu);
    SC_HIP(hipSetDevice(ix->rt->device));
    return search_and_gather_locked(ix, c, q_dev, Q, k, nprobe, all_dist_dev, all_rows_dev, "sc_index_search_sharded_dev");
}

extern "C" sc_status sc_index_search_sharded(sc_index* ix, sc_comm* c, const float* q, int32_t Q, int32_t k, int32_t nprobe, float* out_dist,
                                             int64_t* out_rows) {
    if (!ix || !c || !q || !out_dist || !out_rows) return sc_fail(SC_ERR_INVALID, "sc_index_search_sharded: NULL argument");
    if (Q < 1 || k < 1) return sc_fail(SC_ERR_INVALID, "sc_index_search_sharded: Q and top_k must be >= 1");
    if (ix->rt != c->rt) return sc_fail(SC_ERR_INVALID, "sc_index_search_sharded: index and communicator belong to different runtimes");
    const size_t n = (size_t)Q * k, W = (size_t)c->world;
    std::vector<float> hd(W * n);
    std::vector<int64_t> hr(W * n);
    {
        std::lock_guard<std::mutex> g(ix->mu);
        SC_HIP(hipSetDevice(ix->rt->device));
        hipStream_t s = ix->rt->stream;
        const size_t qb = ((size_t)Q * ix->dim * 4 + 255) & ~(size_t)255, db = (W * n * 4 + 255) & ~(size_t)255;
        // (the staging buffer is the one allocation in front of the collective that can fail on ONE rank: sc_comm_create reserves
        // nothing for it because its size depends on the call; a rank that cannot get it still has to answer the others)
        sc_status st = sc_grow(ix, (void**)&ix->io, &ix->io_cap, qb + db + W * n * 8);
        if (st) {
            // no room for the gathered arrays: take part with the status word alone would leave the payload gathers unmatched, so
            // this rank reports through the status exchange of a zero-payload round -- every rank runs the same two rounds
            std::lock_guard<std::mutex> gc(c->mu);
            std::vector<int64_t> all;
            (void)exchange_status_unlocked(c, st, 0, all, false);
            return st;
        }
        {
            std::lock_guard<std::mutex> gc(c->mu);
            std::vector<int64_t> all;
            sc_status ex = exchange_status_unlocked(c, 0, 0, all, false);  // round 1: did every rank get its staging buffer?
            if (ex) return ex;
            ex = worst_status(c, all, SC_OK, "sc_index_search_sharded (staging buffer)");
            if (ex) return ex;
        }
        float* dq = (float*)ix->io;
        float* all_d = (float*)((char*)ix->io + qb);
        int64_t* all_r = (int64_t*)((char*)ix->io + qb + db);
        SC_HIP(hipMemcpyAsync(dq, q, (size_t)Q * ix->dim * 4, hipMemcpyHostToDevice, s));
        st = search_and_gather_locked(ix, c, dq, Q, k, nprobe, all_d, all_r, "sc_index_search_sharded");
        if (st) return st;
        SC_HIP(hipMemcpyAsync(hd.data(), all_d, W * n * 4, hipMemcpyDeviceToHost, s));
        SC_HIP(hipMemcpyAsync(hr.data(), all_r, W * n * 8, hipMemcpyDeviceToHost, s));
        SC_HIP(hipStreamSynchronize(s));
    }
    return sc_topk_merge_host(ix->metric, (int32_t)W, Q, k, hd.data(), hr.data(), out_dist, out_rows);
}

extern "C" sc_status sc_index_train_sharded(sc_index* ix, sc_comm* c, int32_t niter, int32_t root) {
    if (!ix || !c) return sc_fail(SC_ERR_INVALID, "sc_index_train_sharded: NULL argument");
    if (root < 0 || root >= c->world) return sc_fail(SC_ERR_INVALID, "sc_index_train_sharded: root %d outside [0,%d)", root, c->world);
    if (ix->rt != c->rt) return sc_fail(SC_ERR_INVALID, "sc_index_train_sharded: index and communicator belong to different runtimes");
    if (c->world == 1) return sc_index_train(ix, niter, 0);
    SC_HIP(hipSetDevice(ix->rt->device));
    hipStream_t s = ix->rt->stream;
    struct DevBuf {
        void* p = nullptr;
        ~DevBuf() { hipFree(p); }
    } d_cent;
    // Everything that can fail on ONE rank happens BEFORE the first collective, and its status is exchanged among all ranks: the
    // centroid buffer (every rank knows the upper bound nlist x dim from its own handle) and, on the root, the training itself.
    const size_t cap_bytes = (size_t)(ix->nlist > 0 ? ix->nlist : 1) * ix->dim * 4;
    sc_status st_mine = SC_OK;
    if (hipMalloc(&d_cent.p, cap_bytes) != hipSuccess) {
        (void)hipGetLastError();
        st_mine = sc_fail(SC_ERR_NOMEM, "sc_index_train_sharded: hipMalloc of the centroid buffer (%zu bytes) failed on rank %d", cap_bytes, c->rank);
    }
    std::vector<float> cent;
    int32_t nl = 0;
    if (c->rank == root && !st_mine) {
        st_mine = sc_index_train(ix, niter, 0);
        if (!st_mine) st_mine = sc_index_ivf_info(ix, &nl, nullptr, nullptr);
        if (!st_mine && (size_t)nl * ix->dim * 4 > cap_bytes) st_mine = sc_fail(SC_ERR_STATE, "sc_index_train_sharded: trained nlist %d exceeds the handle's %d", nl, ix->nlist);
        if (!st_mine) {
            cent.resize((size_t)nl * ix->dim);
            st_mine = sc_index_ivf_info(ix, &nl, cent.data(), nullptr);
        }
    }
    std::vector<int64_t> all;
    {
        std::lock_guard<std::mutex> gc(c->mu);
        sc_status ex = exchange_status_unlocked(c, st_mine, nl, all, false);  // every rank takes part, failed or not
        if (ex) return ex;
        ex = worst_status(c, all, st_mine, "sc_index_train_sharded");
        if (ex) return ex;
    }
    const int nlist = (int)all[2 * (size_t)root + 1];
    const size_t bytes = (size_t)nlist * ix->dim * 4;
    if (c->rank == root) SC_HIP(hipMemcpyAsync(d_cent.p, cent.data(), bytes, hipMemcpyHostToDevice, s));
    sc_status st = sc_comm_broadcast(c, d_cent.p, bytes, root);  // the build's one collective (50 MB at 4096 x 3072)
    if (st) return st;
    if (c->rank == root) {
        SC_HIP(hipStreamSynchronize(s));  // cent goes out of scope
        return SC_OK;
    }
    cent.resize((size_t)nlist * ix->dim);
    SC_HIP(hipMemcpyAsync(cent.data(), d_cent.p, bytes, hipMemcpyDeviceToHost, s));
    SC_HIP(hipStreamSynchronize(s));
    return sc_index_assign_lists(ix, cent.data(), nlist);  // an empty shard reports SC_ERR_STATE: the caller skips empty shards
}

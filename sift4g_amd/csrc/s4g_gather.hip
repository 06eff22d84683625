// s4g_gather.hip -- the final gather of per-query top-hit lists across the GPUs of a node, below Python (SURVEY.md 8e;
// BASELINE north_star: "C++ host code calling through a thin C-ABI ... final RCCL gather of per-query top-hit lists over xGMI").
//
// One process per GPU; every process holds an s4g_comm (a rank of an RCCL communicator, made from a 128-byte id that rank 0
// creates and the host program hands to the other ranks by whatever means it has: a file, a pipe, MPI, torch.distributed).
// The scan itself shards without any exchange (pairs are independent); this is the only collective of the path:
//
//   S4G_GATHER_QUERY_SHARD   the database is replicated, rank r scored queries [b_r, b_{r+1}) (b = n * r / world): one
//                            all-gather of (2k + 1) int32 per query, then the rows are put back in query order
//   S4G_GATHER_DB_SHARD      every rank scored ALL queries against its own slice of the database: one all-gather of 2k int32 per
//                            query and rank, then a K-way merge per query under (score descending, global id ascending) -- the
//                            semantics of dbAlignmentsMerge across database chunks
//                            (/root/reference/sift4g/src/database_alignment.cpp:97-104) applied across GPUs
//
// librccl is opened when the first communicator is made (dlopen of the soname: a process that already carries an RCCL, e.g.
// PyTorch's, shares it), so libs4g_hip.so itself loads on a machine without it.  xGMI is point to point and the payload is
// small (3.2 MB for 1000 queries x 400 hits, 32 MB per rank for configs[3]): one direct all-gather, no bucketing.
//
// The merge is rank-by-counting: every list a rank sends is already ordered, so the final position of an entry is its own index
// plus, for every other rank's list, the number of entries there that precede it (one binary search each); entries whose
// position is below k are written straight to their slot.  No sort, no shared memory, any k; keys are distinct (a global id
// lives on one rank), so the result is deterministic.  s4g_merge_topk_host is the same algorithm on the host (what the gloo
// tests and a host program without GPUs in the merge step use).
#include <dlfcn.h>

#include "s4g_internal.hpp"

namespace {

typedef struct { char internal[128]; } RcclId;           // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* RcclComm;
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    std::string error;
};
constexpr int kRcclInt32 = 2;                             // ncclInt32

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // S4G_RCCL_LIBRARY names the one library to open instead (a vendor build elsewhere; the CPU test of this error path)
        const char* forced = getenv("S4G_RCCL_LIBRARY");
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (forced && *forced) r.handle = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
        else
            for (const char* n : names)
                if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        // (dlerror() clears the message it returns: ask once)
        if (!r.handle) { const char* m = dlerror(); r.error = m ? m : "dlopen failed"; return; }
        auto sym = [&](const char* s) { void* p = dlsym(r.handle, s); if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + s; return p; };
        r.GetUniqueId = (int (*)(RcclId*))sym("ncclGetUniqueId");
        r.CommInitRank = (int (*)(RcclComm*, int, RcclId, int))sym("ncclCommInitRank");
        r.CommDestroy = (int (*)(RcclComm))sym("ncclCommDestroy");
        r.AllGather = (int (*)(const void*, void*, size_t, int, RcclComm, hipStream_t))sym("ncclAllGather");
        r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        r.GetVersion = (int (*)(int*))sym("ncclGetVersion");
    });
    return &r;
}

int rccl_ready() {
    Rccl* r = rccl();
    if (!r->handle || !r->error.empty()) return fail(S4G_ERR_NO_DEVICE, "RCCL is not available: %s", r->error.c_str());
    return S4G_OK;
}

#define RCCL_TRY(expr)                                                                                             \
    do {                                                                                                           \
        int e_ = (expr);                                                                                           \
        if (e_ != 0) return fail(S4G_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl()->GetErrorString ? rccl()->GetErrorString(e_) : "?", __FILE__, __LINE__); \
    } while (0)

// ---- key order shared by the device and the host merge: score descending, then global id ascending; unused slots last ----
__host__ __device__ inline unsigned long long merge_key(uint32_t id, int32_t score, bool valid) {
    return valid ? ((unsigned long long)((uint32_t)score ^ 0x80000000u) << 32) | (0xFFFFFFFFull - id) : 0ull;   // (score biased: any int32 orders)
}

// rows of [per][2k + 1]: ids (0xFFFFFFFF unused), scores, count
__global__ void pack_query_rows(const uint32_t* __restrict__ ids, const int32_t* __restrict__ scores, const uint32_t* __restrict__ counts,
                                uint32_t q_local, uint32_t per, uint32_t k, int32_t* __restrict__ out) {
    const uint32_t row = blockIdx.x;
    int32_t* o = out + (uint64_t)row * (2 * k + 1);
    for (uint32_t j = threadIdx.x; j < k; j += blockDim.x) {
        o[j] = row < q_local ? (int32_t)ids[(uint64_t)row * k + j] : -1;
        o[k + j] = row < q_local ? scores[(uint64_t)row * k + j] : 0;
    }
    if (threadIdx.x == 0) o[2 * k] = row < q_local ? (int32_t)counts[row] : 0;
}

// gathered [world][per][2k + 1] -> query order; bounds[r] = n * r / world
__global__ void unpack_query_rows(const int32_t* __restrict__ in, uint32_t n_queries, uint32_t world, uint32_t per, uint32_t k,
                                  uint32_t* __restrict__ ids, int32_t* __restrict__ scores, uint32_t* __restrict__ counts) {
    const uint32_t q = blockIdx.x;
    uint32_t r = (uint32_t)(((uint64_t)q * world + world - 1) / n_queries);      // a rank near the owner, then walk
    if (r >= world) r = world - 1;
    while (r > 0 && (uint64_t)n_queries * r / world > q) --r;
    while (r + 1 < world && (uint64_t)n_queries * (r + 1) / world <= q) ++r;
    const uint32_t local = q - (uint32_t)((uint64_t)n_queries * r / world);
    const int32_t* row = in + ((uint64_t)r * per + local) * (2 * k + 1);
    for (uint32_t j = threadIdx.x; j < k; j += blockDim.x) {
        ids[(uint64_t)q * k + j] = (uint32_t)row[j];
        scores[(uint64_t)q * k + j] = row[k + j];
    }
    if (threadIdx.x == 0) counts[q] = (uint32_t)row[2 * k];
}

// rows of [q][2k]: global ids (local id + base, or through the map; 0xFFFFFFFF beyond the count), scores
__global__ void pack_shard_rows(const uint32_t* __restrict__ ids, const int32_t* __restrict__ scores, const uint32_t* __restrict__ counts,
                                uint32_t k, uint32_t id_base, const uint32_t* __restrict__ id_map, int32_t* __restrict__ out) {
    const uint32_t q = blockIdx.x;
    const uint32_t c = counts[q];
    int32_t* o = out + (uint64_t)q * 2 * k;
    for (uint32_t j = threadIdx.x; j < k; j += blockDim.x) {
        const uint32_t id = ids[(uint64_t)q * k + j];
        const bool ok = j < c && id != 0xFFFFFFFFu;
        o[j] = ok ? (int32_t)(id_map ? id_map[id] : id + id_base) : -1;
        o[k + j] = ok ? scores[(uint64_t)q * k + j] : 0;
    }
}

// number of entries of an ordered list (keys descending, unused = 0 at the end) whose key is greater than `key`
__host__ __device__ inline uint32_t count_before(const int32_t* ids, const int32_t* scores, uint32_t k, unsigned long long key) {
    uint32_t lo = 0, hi = k;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        const unsigned long long km = merge_key((uint32_t)ids[mid], scores[mid], ids[mid] != -1);
        if (km > key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// `world` ordered lists per query -> merged top-k per query.  List r of query q: ids[r * list_stride + q * row_stride + j], scores the same
__global__ void merge_shard_rows(const int32_t* __restrict__ ids_in, const int32_t* __restrict__ scores_in, uint64_t list_stride, uint64_t row_stride,
                                 uint32_t world, uint32_t k_in, uint32_t k_out,
                                 uint32_t* __restrict__ ids, int32_t* __restrict__ scores, uint32_t* __restrict__ counts) {
    const uint32_t q = blockIdx.x;
    __shared__ uint32_t total;
    if (threadIdx.x == 0) total = 0;
    for (uint32_t j = threadIdx.x; j < k_out; j += blockDim.x) { ids[(uint64_t)q * k_out + j] = 0xFFFFFFFFu; scores[(uint64_t)q * k_out + j] = 0; }
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t e = threadIdx.x; e < world * k_in; e += blockDim.x) {
        const uint32_t r = e / k_in, j = e - r * k_in;
        const uint64_t at = (uint64_t)r * list_stride + (uint64_t)q * row_stride;
        if (ids_in[at + j] == -1) continue;
        const unsigned long long key = merge_key((uint32_t)ids_in[at + j], scores_in[at + j], true);
        uint32_t pos = j;
        for (uint32_t o = 0; o < world; ++o) {
            if (o == r) continue;
            const uint64_t oat = (uint64_t)o * list_stride + (uint64_t)q * row_stride;
            pos += count_before(ids_in + oat, scores_in + oat, k_in, key);
        }
        if (pos < k_out) { ids[(uint64_t)q * k_out + pos] = (uint32_t)ids_in[at + j]; scores[(uint64_t)q * k_out + pos] = scores_in[at + j]; ++mine; }
    }
    if (mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) counts[q] = total;
}

}  // namespace

struct s4g_comm {
    int device = 0, rank = 0, world = 1;
    RcclComm comm = nullptr;
    int32_t* d_send = nullptr;
    int32_t* d_recv = nullptr;
    uint64_t send_bytes = 0, recv_bytes = 0;
    hipEvent_t done = nullptr;      // recorded behind the last kernel of a gather: the next gather (on whatever stream) waits for it before
                                    // it packs into the one d_send / d_recv pair, and the buffers are not regrown underneath a gather in flight
    std::mutex mu;                  // calls on one communicator are serialised on the host (RCCL wants its collectives issued in one order anyway)
};

S4G_API int s4g_comm_unique_id(uint8_t* id128) {
    if (!id128) return fail(S4G_ERR_INVALID, "NULL argument");
    int rc = rccl_ready();
    if (rc) return rc;
    RcclId id;
    memset(&id, 0, sizeof id);
    RCCL_TRY(rccl()->GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return S4G_OK;
}

S4G_API int s4g_comm_create(s4g_comm** out, int device, int rank, int world, const uint8_t* id128) {
    if (!out || !id128) return fail(S4G_ERR_INVALID, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(S4G_ERR_INVALID, "rank %d of %d", rank, world);
    int have = device_count_raw();
    if (have <= 0) return fail(S4G_ERR_NO_DEVICE, "no HIP device available (libs4g_hip has no CPU fallback)");
    if (device < 0 || device >= have) return fail(S4G_ERR_INVALID, "device %d out of range [0,%d)", device, have);
    int rc = rccl_ready();
    if (rc) return rc;
    DeviceGuard guard(device);
    RcclId id;
    memcpy(id.internal, id128, 128);
    s4g_comm* c = new s4g_comm();
    c->device = device; c->rank = rank; c->world = world;
    int e = rccl()->CommInitRank(&c->comm, world, id, rank);
    if (e != 0) { delete c; return fail(S4G_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, rccl()->GetErrorString(e)); }
    *out = c;
    return S4G_OK;
}

S4G_API int s4g_comm_info(const s4g_comm* c, int* rank, int* world, int* device, int* rccl_version) {
    if (!c) return fail(S4G_ERR_INVALID, "NULL argument");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (device) *device = c->device;
    if (rccl_version) { *rccl_version = 0; if (rccl()->GetVersion) (void)rccl()->GetVersion(rccl_version); }
    return S4G_OK;
}

S4G_API int s4g_comm_destroy(s4g_comm* c) {
    if (!c) return S4G_OK;
    DeviceGuard guard(c->device);
    (void)hipDeviceSynchronize();
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->comm) (void)rccl()->CommDestroy(c->comm);
    delete c;
    return S4G_OK;
}

// makes `stream` wait for the previous gather on this communicator (any stream) and sizes the staging pair
static int comm_buffers(s4g_comm* c, uint64_t send_bytes, hipStream_t stream) {
    const uint64_t recv_bytes = send_bytes * (uint64_t)c->world;
    if (!c->done) HIP_TRY(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    else if (c->send_bytes < send_bytes || c->recv_bytes < recv_bytes) HIP_TRY(hipEventSynchronize(c->done));   // about to free what it may still read
    else HIP_TRY(hipStreamWaitEvent(stream, c->done, 0));
    if (c->send_bytes < send_bytes) {
        if (c->d_send) HIP_TRY(hipFree(c->d_send));
        c->d_send = nullptr; c->send_bytes = 0;
        HIP_TRY(hipMalloc((void**)&c->d_send, send_bytes));
        c->send_bytes = send_bytes;
    }
    if (c->recv_bytes < recv_bytes) {
        if (c->d_recv) HIP_TRY(hipFree(c->d_recv));
        c->d_recv = nullptr; c->recv_bytes = 0;
        HIP_TRY(hipMalloc((void**)&c->d_recv, recv_bytes));
        c->recv_bytes = recv_bytes;
    }
    return S4G_OK;
}

S4G_API int s4g_gather_topk(s4g_comm* c, int mode, uint32_t n_queries, uint32_t k, const uint32_t* d_ids, const int32_t* d_scores,
                            const uint32_t* d_counts, uint32_t id_base, const uint32_t* d_id_map, uint32_t k_out, uint32_t* d_out_ids,
                            int32_t* d_out_scores, uint32_t* d_out_counts, void* stream_) {
    if (!c || !d_out_ids || !d_out_scores || !d_out_counts) return fail(S4G_ERR_INVALID, "NULL argument");
    if (k == 0 || k > 4096) return fail(S4G_ERR_INVALID, "k must be in 1..4096 (got %u)", k);
    if (n_queries == 0) return S4G_OK;
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lk(c->mu);
    const uint32_t world = (uint32_t)c->world;
    if (mode == S4G_GATHER_QUERY_SHARD) {
        if (k_out != k) return fail(S4G_ERR_INVALID, "query-shard gather: k_out must equal k");
        const uint32_t b0 = (uint32_t)((uint64_t)n_queries * c->rank / world), b1 = (uint32_t)((uint64_t)n_queries * (c->rank + 1) / world);
        const uint32_t q_local = b1 - b0;
        if (q_local && (!d_ids || !d_scores || !d_counts)) return fail(S4G_ERR_INVALID, "NULL argument");
        uint32_t per = 0;
        for (uint32_t r = 0; r < world; ++r)
            per = std::max(per, (uint32_t)((uint64_t)n_queries * (r + 1) / world) - (uint32_t)((uint64_t)n_queries * r / world));
        const uint64_t row = 2ull * k + 1;
        int rc = comm_buffers(c, (uint64_t)per * row * 4, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(pack_query_rows, dim3(per), dim3(128), 0, stream, d_ids, d_scores, d_counts, q_local, per, k, c->d_send);
        HIP_TRY(hipGetLastError());
        RCCL_TRY(rccl()->AllGather(c->d_send, c->d_recv, (size_t)per * row, kRcclInt32, c->comm, stream));
        hipLaunchKernelGGL(unpack_query_rows, dim3(n_queries), dim3(128), 0, stream, c->d_recv, n_queries, world, per, k, d_out_ids, d_out_scores, d_out_counts);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->done, stream));
        return S4G_OK;
    }
    if (mode == S4G_GATHER_DB_SHARD) {
        if (!d_ids || !d_scores || !d_counts) return fail(S4G_ERR_INVALID, "NULL argument");
        if (k_out == 0 || k_out > 4096) return fail(S4G_ERR_INVALID, "k_out must be in 1..4096 (got %u)", k_out);
        int rc = comm_buffers(c, (uint64_t)n_queries * 2 * k * 4, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(pack_shard_rows, dim3(n_queries), dim3(128), 0, stream, d_ids, d_scores, d_counts, k, id_base, d_id_map, c->d_send);
        HIP_TRY(hipGetLastError());
        RCCL_TRY(rccl()->AllGather(c->d_send, c->d_recv, (size_t)n_queries * 2 * k, kRcclInt32, c->comm, stream));
        hipLaunchKernelGGL(merge_shard_rows, dim3(n_queries), dim3(256), 0, stream, c->d_recv, c->d_recv + k, (uint64_t)n_queries * 2 * k, (uint64_t)2 * k,
                           world, k, k_out, d_out_ids, d_out_scores, d_out_counts);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->done, stream));
        return S4G_OK;
    }
    return fail(S4G_ERR_INVALID, "mode %d is neither S4G_GATHER_QUERY_SHARD nor S4G_GATHER_DB_SHARD", mode);
}

// The merge kernel on lists that are on the device already (rows [list][query][k_in] of global ids / scores, as s4g_merge_topk_host).
S4G_API int s4g_merge_topk(int device, uint32_t lists, uint32_t n_queries, uint32_t k_in, const uint32_t* d_ids, const int32_t* d_scores,
                           uint32_t k_out, uint32_t* d_out_ids, int32_t* d_out_scores, uint32_t* d_out_counts, void* stream_) {
    if ((lists && n_queries && k_in && (!d_ids || !d_scores)) || !d_out_ids || !d_out_scores || !d_out_counts) return fail(S4G_ERR_INVALID, "NULL argument");
    if (k_out == 0 || k_out > 4096) return fail(S4G_ERR_INVALID, "k_out must be in 1..4096 (got %u)", k_out);
    int have = device_count_raw();
    if (have <= 0) return fail(S4G_ERR_NO_DEVICE, "no HIP device available (libs4g_hip has no CPU fallback)");
    if (device < 0 || device >= have) return fail(S4G_ERR_INVALID, "device %d out of range [0,%d)", device, have);
    if (n_queries == 0) return S4G_OK;
    DeviceGuard guard(device);
    hipLaunchKernelGGL(merge_shard_rows, dim3(n_queries), dim3(256), 0, (hipStream_t)stream_, reinterpret_cast<const int32_t*>(d_ids), d_scores,
                       (uint64_t)n_queries * k_in, (uint64_t)k_in, lists, k_in, k_out, d_out_ids, d_out_scores, d_out_counts);
    HIP_TRY(hipGetLastError());
    return S4G_OK;
}

// The K-way merge on the host: `lists` ordered lists per query, rows [list][query][k_in] of global ids (0xFFFFFFFF = unused, only
// at a row's end) and scores.  Same algorithm, same order as merge_shard_rows.  No device needed.
S4G_API int s4g_merge_topk_host(uint32_t lists, uint32_t n_queries, uint32_t k_in, const uint32_t* ids, const int32_t* scores,
                                uint32_t k_out, uint32_t* out_ids, int32_t* out_scores, uint32_t* out_counts) {
    if ((lists && n_queries && k_in && (!ids || !scores)) || !out_ids || !out_scores || !out_counts) return fail(S4G_ERR_INVALID, "NULL argument");
    if (k_out == 0) return fail(S4G_ERR_INVALID, "k_out must be positive");
    for (uint32_t q = 0; q < n_queries; ++q) {
        uint32_t total = 0;
        for (uint32_t j = 0; j < k_out; ++j) { out_ids[(uint64_t)q * k_out + j] = 0xFFFFFFFFu; out_scores[(uint64_t)q * k_out + j] = 0; }
        for (uint32_t r = 0; r < lists; ++r) {
            const int32_t* rid = reinterpret_cast<const int32_t*>(ids) + ((uint64_t)r * n_queries + q) * k_in;
            const int32_t* rsc = scores + ((uint64_t)r * n_queries + q) * k_in;
            for (uint32_t j = 0; j < k_in; ++j) {
                if (rid[j] == -1) break;
                const unsigned long long key = merge_key((uint32_t)rid[j], rsc[j], true);
                uint32_t pos = j;
                for (uint32_t o = 0; o < lists && pos < k_out; ++o) {
                    if (o == r) continue;
                    pos += count_before(reinterpret_cast<const int32_t*>(ids) + ((uint64_t)o * n_queries + q) * k_in,
                                        scores + ((uint64_t)o * n_queries + q) * k_in, k_in, key);
                }
                if (pos < k_out) { out_ids[(uint64_t)q * k_out + pos] = (uint32_t)rid[j]; out_scores[(uint64_t)q * k_out + pos] = rsc[j]; ++total; }
            }
        }
        out_counts[q] = total;
    }
    return S4G_OK;
}

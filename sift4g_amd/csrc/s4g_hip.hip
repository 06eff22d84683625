// s4g_hip.hip -- C ABI (include/s4g_hip.h) over the gfx950 kernels.
//
// Host side of the drop-in boundary: database packing (length sort, 64-lane
// groups, lane-transposed residues), residency in HBM, persistent-grid launch.
// Mirrors what swsharp's chainDatabaseCreate + alignDatabase(step 1) do for
// sift4g (/root/reference/sift4g/src/database_alignment.cpp:80-86).
#include "../../include/s4g_hip.h"
#include "sw_kernels.hip.h"
#include "sw_trace.hip.h"
#include "sw_pair.hip.h"
#include "sw_pack.hip.h"
#include "matrices.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

// S4G_PROFILE=1: cumulative host time of the database-creation phases, printed at exit (diagnostic only)
namespace {
struct CreateClock {
    double t[4] = {0, 0, 0, 0};
    long n = 0;
    bool on = getenv("S4G_PROFILE") != nullptr;
    ~CreateClock() {
        if (on && n) fprintf(stderr, "[s4g profile] s4g_db_create x%ld: lengths+sort %.3f s, host pack %.3f s, tile upload (+ device pack) %.3f s, raw codes upload %.3f s\n", n, t[0], t[1], t[2], t[3]);
    }
} g_create_clock;
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

using namespace s4g;

#define S4G_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------
namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(S4G_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                 \
    } while (0)

int device_count_raw() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); return -1; }
    return n;
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; (void)hipSetDevice(dev); }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct s4g_scorer {
    int32_t table[26 * 26];
    int8_t mat32[32 * 32];
    int open, ext;
    int max_score, min_score;
    int row_max[32];        // max(0, max_b table[a][b]): per-row bound on what one DP row can add
    uint64_t serial;        // identity for caches keyed on the scorer
};

// A tile = a slice of the length-sorted database that is scanned by one launch (per kernel).  Tiles keep the
// strip-boundary workspace bounded (it scales with the tile's longest sequence) and let huge databases stream
// through a bounded score buffer (s4g_scan_topk).  Small databases are a single tile.
struct Tile {
    uint32_t n_seqs = 0, n_groups = 0, max_len4 = 0;
    uint64_t residues = 0, padded_residues = 0;
    uint32_t* d_res = nullptr;          // lane-transposed residues
    uint64_t* d_grp_off = nullptr;
    uint32_t* d_grp_len4 = nullptr;
    uint32_t* d_perm_global = nullptr;  // [G*64] original target id of (g, lane); ~0u = empty lane
    uint32_t* d_perm_local = nullptr;   // [G*64] column in the tile's own score buffer = rank of the id inside the tile
    uint32_t* d_col_ids = nullptr;      // [n_seqs] tile column -> original target id (ascending)
    std::vector<uint64_t> h_grp_off;    // host copy (traceback addresses lanes of arbitrary targets)
    char* d_block = nullptr;            // small tiles: every array above lives in this one allocation (recycled through
    uint64_t block_bytes = 0;           // DeviceCtx::block_cache: sift4g creates a candidate database per query)
    uint64_t pos0 = 0;                  // position of the tile's first sequence in the global length-sorted order
};

struct s4g_db {
    int device = 0;
    uint32_t n_seqs = 0;
    uint64_t residues = 0, padded_residues = 0, hbm_bytes = 0;
    uint32_t max_tile_seqs = 0;
    std::vector<Tile> tiles;
    std::vector<uint32_t> len;     // original index -> length
    std::vector<uint32_t> pos_of_id;   // original index -> position in the global length-sorted order
    // per-sequence tables for the pair kernels (built on first use, under the device mutex)
    mutable uint64_t* d_seq_addr = nullptr;     // [n_seqs] device address of the sequence's first residue dword
    mutable uint32_t* d_seq_len = nullptr;      // [n_seqs]
    mutable std::vector<uint32_t> id_of_pos;    // position in the length-sorted order -> original index
    bool seq_tables_in_block = false;           // d_seq_addr / d_seq_len are part of tiles[0].d_block
};

// Per-device scratch shared by every database on that device: strip-boundary workspace, work-queue counters, query
// lists, the streaming score buffer, a grow-only arena for the traceback.  sift4g's wrapper creates and destroys one
// ChainDatabase per (query, chunk) (database_alignment.cpp:80,92); keeping the scratch here instead of in s4g_db
// removes a multi-hundred-MB hipMalloc/hipFree pair from every such call.  All launches on a device serialize on
// `mu` at enqueue time and on `last_use` across streams.
struct DeviceCtx {
    std::mutex mu;
    int device = 0;
    bool have_prop = false;
    hipDeviceProp_t prop;
    int32_t* d_ws = nullptr;
    uint64_t ws_bytes = 0;
    unsigned* d_counter = nullptr;
    int8_t* d_mat = nullptr;
    uint32_t* d_qlist = nullptr;     // query-id lists of the current call (pairs for the packed tiers, singles for int32)
    uint64_t qlist_cap = 0;
    int32_t* d_tile_scores = nullptr;   // s4g_scan_topk: [query batch][tile columns]
    uint64_t tile_scores_bytes = 0;
    uint32_t grid_i32 = 0, grid_i16 = 0, grid_f16 = 0, grid_drift = 0;
    uint32_t grid_pair[6] = {0, 0, 0, 0, 0, 0};
    uint32_t grid_pair_trace[6] = {0, 0, 0, 0, 0, 0};
    // pair lists: pinned staging for the items of the current call, their device copy (+ host-route scores behind it)
    s4g::PairItem* h_items = nullptr;
    s4g::PairItem* d_items = nullptr;
    uint64_t items_cap = 0;
    hipEvent_t items_done = nullptr;    // the upload that last read h_items
    hipEvent_t last_use = nullptr;      // orders launches that share the scratch buffers across streams
    // long-sequence tiles have few, huge items: they run on a side stream (own workspace and counters) underneath the
    // main tiles instead of as a tail-dominated launch of their own
    hipStream_t side_stream = nullptr;
    hipEvent_t side_fork = nullptr, side_join = nullptr;
    int32_t* d_ws_side = nullptr;
    uint64_t ws_side_bytes = 0;
    std::vector<std::pair<char*, uint64_t>> block_cache;    // freed small-tile blocks, reused by the next s4g_db_create
    char* arena = nullptr;              // traceback temporaries
    uint64_t arena_bytes = 0;
};

namespace {
DeviceCtx g_ctx[64];
DeviceCtx* device_ctx(int device) {
    DeviceCtx* c = &g_ctx[device & 63];
    c->device = device;
    return c;
}
}  // namespace

struct s4g_queries {
    int device = 0;
    uint32_t n = 0;
    uint8_t* d_codes = nullptr;
    uint64_t* d_off = nullptr;
    uint32_t* d_len = nullptr;
    std::vector<uint32_t> len;
    std::vector<uint8_t> codes;              // host copy (unpadded) for per-scorer bounds
    std::vector<uint64_t> off;
    uint32_t max_len = 0;
    mutable std::vector<uint64_t> score_bound;
    mutable uint64_t bound_scorer = 0;
};

// ---------------------------------------------------------------------------
S4G_API const char* s4g_last_error(void) { return g_err.c_str(); }
S4G_API int s4g_abi_version(void) { return 1; }

S4G_API int s4g_device_count(int* count) {
    if (!count) return fail(S4G_ERR_INVALID, "count is NULL");
    int n = device_count_raw();
    if (n < 0) { *count = 0; return fail(S4G_ERR_NO_DEVICE, "HIP runtime reports no usable device"); }
    *count = n;
    return S4G_OK;
}

S4G_API int s4g_device_check(const int* cards, int n) {
    if (n == 0) return 1;
    if (n < 0 || !cards) return 0;
    int have = device_count_raw();
    if (have <= 0) return 0;
    for (int i = 0; i < n; ++i)
        if (cards[i] < 0 || cards[i] >= have) return 0;
    return 1;
}

S4G_API int s4g_matrix_by_name(const char* name, int32_t* table26x26) {
    if (!name || !table26x26) return fail(S4G_ERR_INVALID, "NULL argument");
    if (!matrix_by_name(name, table26x26)) return fail(S4G_ERR_INVALID, "unknown substitution matrix '%s' (available: BLOSUM_62)", name);
    return S4G_OK;
}

S4G_API int s4g_scorer_create(s4g_scorer** out, const int32_t* table, int gap_open, int gap_extend) {
    if (!out || !table) return fail(S4G_ERR_INVALID, "NULL argument");
    if (gap_extend < 0 || gap_open < gap_extend)
        return fail(S4G_ERR_INVALID, "need 0 <= gap_extend <= gap_open (got open=%d extend=%d)", gap_open, gap_extend);
    s4g_scorer* s = new s4g_scorer();
    s->open = gap_open;
    s->ext = gap_extend;
    s->max_score = 0;
    s->min_score = 0;
    for (int i = 0; i < 32 * 32; ++i) s->mat32[i] = (int8_t)kPadScore;
    for (int a = 0; a < 26; ++a)
        for (int b = 0; b < 26; ++b) {
            int v = table[a * 26 + b];
            if (v < -127 || v > 127) { delete s; return fail(S4G_ERR_INVALID, "table entry %d out of int8 range", v); }
            s->table[a * 26 + b] = v;
            s->mat32[a * 32 + b] = (int8_t)v;
            s->max_score = std::max(s->max_score, v);
            s->min_score = std::min(s->min_score, v);
        }
    for (int a = 0; a < 32; ++a) {
        s->row_max[a] = 0;
        if (a < 26) for (int b = 0; b < 26; ++b) s->row_max[a] = std::max(s->row_max[a], s->table[a * 26 + b]);
    }
    static std::atomic<uint64_t> next_serial{1};
    s->serial = next_serial.fetch_add(1);
    *out = s;
    return S4G_OK;
}

S4G_API int s4g_scorer_destroy(s4g_scorer* s) { delete s; return S4G_OK; }

// ---------------------------------------------------------------------------
namespace {

constexpr uint32_t kTileMaxSeqs = 1u << 20;            // sequences per tile
constexpr uint64_t kTileMaxResidues = 1ull << 30;      // padded residues per tile
constexpr uint32_t kLongSeq = 12288;                   // sequences longer than this get tiles of their own: beyond ~16k residues
                                                       // the per-wave strip-boundary workspace no longer fits the budget at full grid

constexpr uint64_t kBlockMax = 64ull << 20;      // tiles up to this size are one allocation
constexpr size_t kBlockCacheEntries = 8;

void free_tile(Tile& t, DeviceCtx* cx) {
    if (t.d_block) {
        std::lock_guard<std::mutex> lk(cx->mu);
        if (cx->block_cache.size() < kBlockCacheEntries) cx->block_cache.emplace_back(t.d_block, t.block_bytes);
        else (void)hipFree(t.d_block);
    } else {
        (void)hipFree(t.d_res); (void)hipFree(t.d_grp_off); (void)hipFree(t.d_grp_len4);
        (void)hipFree(t.d_perm_global); (void)hipFree(t.d_perm_local); (void)hipFree(t.d_col_ids);
    }
    t = Tile();
}

template <typename T>
struct DevBuf {          // scoped device allocation for the non-hot paths
    T* p = nullptr;
    ~DevBuf() { (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
    hipError_t upload(const std::vector<T>& v) {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

// Packs order[p0, p1) (length-sorted, longest first) into one tile and uploads it.
// d_raw != nullptr: the raw codes are already on the device and the transposition runs there (sw_pack_tile); d_bad
// collects its validity flag (checked by the caller after the last tile).
hipError_t build_tile(s4g_db* db, Tile& t, const std::vector<uint32_t>& order, size_t p0, size_t p1,
                      const uint8_t* codes, const uint64_t* offsets, std::string* err, uint64_t extra_bytes = 0,
                      uint64_t* extra_off = nullptr, const uint8_t* d_raw = nullptr, int* d_bad = nullptr) {
    const uint32_t n = (uint32_t)(p1 - p0);
    const uint32_t G = (n + 63) / 64;
    t.n_seqs = n; t.n_groups = G;
    std::vector<uint64_t> grp_off(G);
    std::vector<uint32_t> grp_len4(G);
    uint64_t dwords = 0;
    for (uint32_t g = 0; g < G; ++g) {
        const uint32_t lg = db->len[order[p0 + (size_t)g * 64]];
        grp_len4[g] = (lg + 3) / 4;
        grp_off[g] = dwords;
        dwords += (uint64_t)grp_len4[g] * 64;
        t.max_len4 = std::max(t.max_len4, grp_len4[g]);
    }
    t.padded_residues = dwords * 4;
    // tile column of a sequence = rank of its original id inside the tile (ties in top-k break by id)
    std::vector<uint32_t> col_ids(order.begin() + p0, order.begin() + p1);
    const bool whole = (n == db->n_seqs);           // single tile: the ids are 0..n-1 and the rank of an id is the id
    if (whole) std::iota(col_ids.begin(), col_ids.end(), 0u);
    else std::sort(col_ids.begin(), col_ids.end());
    std::vector<uint32_t> perm_g((size_t)G * 64, 0xFFFFFFFFu), perm_l((size_t)G * 64, 0xFFFFFFFFu);
    if (d_raw) {
        const double t_pack0 = now_s();
        std::vector<uint64_t> src_off((size_t)G * 64, 0);
        std::vector<uint32_t> src_len((size_t)G * 64, 0);
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t id = order[p0 + k];
            perm_g[k] = id;
            perm_l[k] = whole ? id : (uint32_t)(std::lower_bound(col_ids.begin(), col_ids.end(), id) - col_ids.begin());
            src_off[k] = offsets[id];
            src_len[k] = db->len[id];
            t.residues += db->len[id];
        }
        const double t_up0 = now_s();
        g_create_clock.t[1] += t_up0 - t_pack0;
        struct UpTimer { double t0; ~UpTimer() { g_create_clock.t[2] += now_s() - t0; } } up_timer{t_up0};
        hipError_t e;
        auto up = [&](void** d, const void* h, size_t bytes) -> hipError_t {
            hipError_t e2 = hipMalloc(d, bytes ? bytes : 4);
            if (e2 != hipSuccess) return e2;
            db->hbm_bytes += bytes;
            return bytes ? hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) : hipSuccess;
        };
        if ((e = hipMalloc((void**)&t.d_res, dwords ? dwords * 4 : 4)) != hipSuccess) return e;
        db->hbm_bytes += dwords * 4;
        if ((e = up((void**)&t.d_grp_off, grp_off.data(), (size_t)G * 8)) != hipSuccess) return e;
        if ((e = up((void**)&t.d_grp_len4, grp_len4.data(), (size_t)G * 4)) != hipSuccess) return e;
        if ((e = up((void**)&t.d_perm_global, perm_g.data(), (size_t)G * 64 * 4)) != hipSuccess) return e;
        if ((e = up((void**)&t.d_perm_local, perm_l.data(), (size_t)G * 64 * 4)) != hipSuccess) return e;
        if ((e = up((void**)&t.d_col_ids, col_ids.data(), (size_t)n * 4)) != hipSuccess) return e;
        DevBuf<uint64_t> d_src_off;
        DevBuf<uint32_t> d_src_len;
        if ((e = d_src_off.upload(src_off)) != hipSuccess || (e = d_src_len.upload(src_len)) != hipSuccess) return e;
        PackArgs a;
        a.raw = d_raw; a.src_off = d_src_off.p; a.src_len = d_src_len.p; a.grp_off = t.d_grp_off; a.grp_len4 = t.d_grp_len4;
        a.res = t.d_res; a.bad = d_bad;
        if (G) hipLaunchKernelGGL(sw_pack_tile, dim3(G), dim3(256), 0, nullptr, a);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return e;      // d_src_* go out of scope
        t.h_grp_off.swap(grp_off);
        t.pos0 = p0;
        return hipSuccess;
    }
    std::vector<uint32_t> res(dwords ? dwords : 1, 0x1F1F1F1Fu);
    std::atomic<bool> bad{false};
    auto pack = [&](uint32_t g0, uint32_t g1) {
        for (uint32_t g = g0; g < g1; ++g) {
            uint32_t* base = res.data() + grp_off[g];
            const uint32_t lanes = std::min<uint32_t>(64, n - g * 64);
            for (uint32_t lane = 0; lane < lanes; ++lane) {
                const uint32_t id = order[p0 + (size_t)g * 64 + lane];
                perm_g[(size_t)g * 64 + lane] = id;
                perm_l[(size_t)g * 64 + lane] = whole ? id : (uint32_t)(std::lower_bound(col_ids.begin(), col_ids.end(), id) - col_ids.begin());
                const uint8_t* src = codes + offsets[id];
                const uint32_t l = db->len[id];
                // four residues per dword; a byte >= 26 anywhere is an error (SWAR test: bit 7 of b | ((b & 0x7F) + 102))
                uint32_t c4 = 0, any_bad = 0;
                for (; c4 + 1 <= l / 4; ++c4) {
                    uint32_t w;
                    memcpy(&w, src + 4 * (size_t)c4, 4);
                    any_bad |= (w | ((w & 0x7F7F7F7Fu) + 0x66666666u)) & 0x80808080u;
                    base[(size_t)c4 * 64 + lane] = w;
                }
                if (l & 3) {
                    uint32_t w = 0x1F1F1F1Fu;
                    for (uint32_t j = 4 * c4; j < l; ++j) {
                        const uint32_t c = src[j];
                        if (c >= 26) any_bad = 1;
                        w = (w & ~(0xFFu << (8 * (j & 3)))) | (c << (8 * (j & 3)));
                    }
                    base[(size_t)c4 * 64 + lane] = w;
                }
                if (any_bad) { bad = true; return; }
            }
        }
    };
    const double t_pack = now_s();
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const unsigned nthreads = dwords > (1u << 22) ? hw : 1;
    if (nthreads == 1) pack(0, G);
    else {
        std::vector<std::thread> th;
        for (unsigned k = 0; k < nthreads; ++k) th.emplace_back(pack, (uint32_t)((uint64_t)G * k / nthreads), (uint32_t)((uint64_t)G * (k + 1) / nthreads));
        for (auto& x : th) x.join();
    }
    if (bad) { *err = "residue code out of range (codes must be letter - 'A', 0..25)"; return hipErrorInvalidValue; }
    for (size_t p = p0; p < p1; ++p) t.residues += db->len[order[p]];
    const double t_up = now_s();
    g_create_clock.t[1] += t_up - t_pack;
    struct UpTimer { double t0; ~UpTimer() { g_create_clock.t[2] += now_s() - t0; } } up_timer{t_up};
    auto up = [&](void** d, const void* h, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(d, bytes ? bytes : 4);
        if (e != hipSuccess) return e;
        db->hbm_bytes += bytes;
        return bytes ? hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    hipError_t e;
    {   // small tile: one allocation (taken from the per-device cache when one fits) instead of six
        uint64_t need = 0;
        auto reserve = [&](uint64_t bytes) { const uint64_t at = need; need = (need + bytes + 255) / 256 * 256; return at; };
        const uint64_t o_res = reserve(dwords * 4), o_off = reserve((uint64_t)G * 8), o_len4 = reserve((uint64_t)G * 4);
        const uint64_t o_pg = reserve((uint64_t)G * 256), o_pl = reserve((uint64_t)G * 256), o_col = reserve((uint64_t)n * 4);
        const uint64_t o_extra = reserve(extra_bytes);
        if (need <= kBlockMax) {
            DeviceCtx* cx = device_ctx(db->device);
            char* blk = nullptr;
            uint64_t blk_bytes = 0;
            {
                std::lock_guard<std::mutex> lk(cx->mu);
                for (size_t k = 0; k < cx->block_cache.size(); ++k)
                    if (cx->block_cache[k].second >= need && cx->block_cache[k].second <= 4 * need + (1u << 20)) {
                        blk = cx->block_cache[k].first; blk_bytes = cx->block_cache[k].second;
                        cx->block_cache.erase(cx->block_cache.begin() + k);
                        break;
                    }
            }
            if (!blk) {
                blk_bytes = need + need / 4;          // a little slack so the next, slightly larger database still fits
                if ((e = hipMalloc((void**)&blk, blk_bytes)) != hipSuccess) return e;
            }
            t.d_block = blk; t.block_bytes = blk_bytes;
            db->hbm_bytes += blk_bytes;
            t.d_res = reinterpret_cast<uint32_t*>(blk + o_res); t.d_grp_off = reinterpret_cast<uint64_t*>(blk + o_off);
            t.d_grp_len4 = reinterpret_cast<uint32_t*>(blk + o_len4); t.d_perm_global = reinterpret_cast<uint32_t*>(blk + o_pg);
            t.d_perm_local = reinterpret_cast<uint32_t*>(blk + o_pl); t.d_col_ids = reinterpret_cast<uint32_t*>(blk + o_col);
            if (extra_off) *extra_off = o_extra;
            auto cp = [&](void* d, const void* h, size_t bytes) { return bytes ? hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) : hipSuccess; };
            if ((e = cp(t.d_res, res.data(), dwords * 4)) != hipSuccess || (e = cp(t.d_grp_off, grp_off.data(), (size_t)G * 8)) != hipSuccess ||
                (e = cp(t.d_grp_len4, grp_len4.data(), (size_t)G * 4)) != hipSuccess || (e = cp(t.d_perm_global, perm_g.data(), (size_t)G * 256)) != hipSuccess ||
                (e = cp(t.d_perm_local, perm_l.data(), (size_t)G * 256)) != hipSuccess || (e = cp(t.d_col_ids, col_ids.data(), (size_t)n * 4)) != hipSuccess)
                return e;
            t.h_grp_off.swap(grp_off);
            t.pos0 = p0;
            return hipSuccess;
        }
    }
    if ((e = up((void**)&t.d_res, res.data(), dwords * 4)) != hipSuccess) return e;
    if ((e = up((void**)&t.d_grp_off, grp_off.data(), (size_t)G * 8)) != hipSuccess) return e;
    if ((e = up((void**)&t.d_grp_len4, grp_len4.data(), (size_t)G * 4)) != hipSuccess) return e;
    if ((e = up((void**)&t.d_perm_global, perm_g.data(), (size_t)G * 64 * 4)) != hipSuccess) return e;
    if ((e = up((void**)&t.d_perm_local, perm_l.data(), (size_t)G * 64 * 4)) != hipSuccess) return e;
    if ((e = up((void**)&t.d_col_ids, col_ids.data(), (size_t)n * 4)) != hipSuccess) return e;
    t.h_grp_off.swap(grp_off);
    t.pos0 = p0;
    return hipSuccess;
}

}  // namespace

namespace {
// codes: host copy of the residues, or NULL when d_codes (device copy, caller-owned) is given
int db_create_impl(s4g_db** out, int device, const uint8_t* codes, const uint8_t* d_codes, const uint64_t* offsets, uint32_t n_seqs,
                   uint32_t max_tile_seqs);
}

S4G_API int s4g_db_create_tiled(s4g_db** out, int device, const uint8_t* codes, const uint64_t* offsets, uint32_t n_seqs,
                                uint32_t max_tile_seqs) {
    if (!out || !offsets || (!codes && n_seqs && offsets[n_seqs])) return fail(S4G_ERR_INVALID, "NULL argument");
    return db_create_impl(out, device, codes, nullptr, offsets, n_seqs, max_tile_seqs);
}

S4G_API int s4g_db_create_from_device(s4g_db** out, int device, const uint8_t* d_codes, const uint64_t* d_offsets, uint32_t n_seqs,
                                      uint32_t max_tile_seqs) {
    if (!out || !d_offsets || (!d_codes && n_seqs)) return fail(S4G_ERR_INVALID, "NULL argument");
    int have = device_count_raw();
    if (have <= 0) return fail(S4G_ERR_NO_DEVICE, "no HIP device available (libs4g_hip has no CPU fallback)");
    if (device < 0 || device >= have) return fail(S4G_ERR_INVALID, "device %d out of range [0,%d)", device, have);
    DeviceGuard guard(device);
    std::vector<uint64_t> offsets((size_t)n_seqs + 1);
    HIP_TRY(hipMemcpy(offsets.data(), d_offsets, offsets.size() * 8, hipMemcpyDeviceToHost));    // only the lengths come to the host
    return db_create_impl(out, device, nullptr, d_codes, offsets.data(), n_seqs, max_tile_seqs);
}

namespace {
int db_create_impl(s4g_db** out, int device, const uint8_t* codes, const uint8_t* d_codes, const uint64_t* offsets, uint32_t n_seqs,
                   uint32_t max_tile_seqs) {
    int have = device_count_raw();
    if (have <= 0) return fail(S4G_ERR_NO_DEVICE, "no HIP device available (libs4g_hip has no CPU fallback)");
    if (device < 0 || device >= have) return fail(S4G_ERR_INVALID, "device %d out of range [0,%d)", device, have);
    if (max_tile_seqs == 0) max_tile_seqs = kTileMaxSeqs;
    max_tile_seqs = (max_tile_seqs + 63) / 64 * 64;
    DeviceGuard guard(device);

    s4g_db* db = new s4g_db();
    db->device = device;
    db->n_seqs = n_seqs;
    db->len.resize(n_seqs);
    const double t_begin = now_s();
    for (uint32_t i = 0; i < n_seqs; ++i) {
        uint64_t l = offsets[i + 1] - offsets[i];
        if (offsets[i + 1] < offsets[i] || l > 0x7FFFFFF0ull) { delete db; return fail(S4G_ERR_INVALID, "bad offsets at %u", i); }
        db->len[i] = (uint32_t)l;
    }
    // length sort, longest first (ties by original index): items are issued in this order so the longest
    // work runs first and the tail of every launch is short work.
    std::vector<uint32_t> order(n_seqs);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return db->len[a] > db->len[b]; });

    db->pos_of_id.resize(n_seqs);
    for (uint32_t pp = 0; pp < n_seqs; ++pp) db->pos_of_id[order[pp]] = pp;
    g_create_clock.t[0] += now_s() - t_begin; g_create_clock.n++;
    // Databases from 8 MB to 16 GB of residues are transposed on the device: one upload of the raw codes, then one
    // sw_pack_tile launch per tile.  Smaller ones are packed by the host (a launch and two allocations would cost more
    // than they save); larger ones tile by tile on the host, so that the raw copy never doubles the footprint.
    const uint64_t total_res = n_seqs ? offsets[n_seqs] : 0;
    DevBuf<uint8_t> d_raw;
    DevBuf<int> d_bad;
    const uint8_t* raw_dev = d_codes;       // codes that already live on the device are packed there whatever their size
    if (d_codes || (total_res >= (8ull << 20) && total_res <= (16ull << 30))) {
        const double t_raw = now_s();
        hipError_t e1 = hipSuccess;
        if (!d_codes) {
            e1 = d_raw.alloc(total_res);
            if (e1 == hipSuccess) e1 = hipMemcpy(d_raw.p, codes, total_res, hipMemcpyHostToDevice);
            raw_dev = d_raw.p;
        }
        if (e1 == hipSuccess) e1 = d_bad.alloc(1);
        if (e1 == hipSuccess) e1 = hipMemset(d_bad.p, 0, 4);
        if (e1 != hipSuccess) { s4g_db_destroy(db); return fail(S4G_ERR_HIP, "database upload failed: %s", hipGetErrorString(e1)); }
        g_create_clock.t[3] += now_s() - t_raw;
    }
    // cut the sorted order into tiles: by count, by padded residues, and at the long-sequence threshold
    size_t p = 0;
    while (p < n_seqs) {
        size_t q = p;
        uint64_t padded = 0;
        const bool long_class = db->len[order[p]] > kLongSeq;
        while (q < n_seqs && (q - p) < max_tile_seqs) {
            const uint32_t l = db->len[order[q]];
            if (long_class != (l > kLongSeq)) break;     // long sequences get tiles of their own (partial last group is fine)
            if (((q - p) % 64) == 0) {
                const uint64_t add = (uint64_t)((l + 3) / 4 * 4) * 64;     // this group's padded residues
                if (q > p && padded + add > kTileMaxResidues) break;
                padded += add;
            }
            ++q;
        }
        db->tiles.emplace_back();
        std::string err;
        // a database that is one small tile also carries the per-sequence tables of the pair kernels in its block
        const bool whole = (p == 0 && q == n_seqs);
        uint64_t tab_off = 0;
        hipError_t e = build_tile(db, db->tiles.back(), order, p, q, codes, offsets, &err, whole ? (uint64_t)n_seqs * 12 + 512 : 0, &tab_off,
                                  raw_dev, d_bad.p);
        if (e == hipSuccess && whole && db->tiles.back().d_block && n_seqs) {
            const Tile& t0 = db->tiles.back();
            std::vector<uint64_t> addr(n_seqs);
            for (uint32_t pos = 0; pos < n_seqs; ++pos)
                addr[order[pos]] = (uint64_t)(uintptr_t)(t0.d_res + t0.h_grp_off[pos >> 6] + (pos & 63));
            db->d_seq_addr = reinterpret_cast<uint64_t*>(t0.d_block + tab_off);
            db->d_seq_len = reinterpret_cast<uint32_t*>(t0.d_block + tab_off + (((uint64_t)n_seqs * 8 + 255) / 256 * 256));
            db->seq_tables_in_block = true;
            db->id_of_pos = order;
            if ((e = hipMemcpy(db->d_seq_addr, addr.data(), (size_t)n_seqs * 8, hipMemcpyHostToDevice)) == hipSuccess)
                e = hipMemcpy(db->d_seq_len, db->len.data(), (size_t)n_seqs * 4, hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
            std::string msg = err.empty() ? std::string(hipGetErrorString(e)) : err;
            s4g_db_destroy(db);
            return fail(err.empty() ? S4G_ERR_HIP : S4G_ERR_INVALID, "database upload failed: %s", msg.c_str());
        }
        const Tile& t = db->tiles.back();
        db->residues += t.residues;
        db->padded_residues += t.padded_residues;
        db->max_tile_seqs = std::max(db->max_tile_seqs, t.n_seqs);
        p = q;
    }
    if (d_bad.p) {
        int h_bad = 0;
        hipError_t e1 = hipMemcpy(&h_bad, d_bad.p, 4, hipMemcpyDeviceToHost);
        if (e1 != hipSuccess || h_bad) {
            s4g_db_destroy(db);
            return e1 != hipSuccess ? fail(S4G_ERR_HIP, "database upload failed: %s", hipGetErrorString(e1))
                                    : fail(S4G_ERR_INVALID, "database upload failed: residue code out of range (codes must be letter - 'A', 0..25)");
        }
    }
    *out = db;
    return S4G_OK;
}
}  // namespace

S4G_API int s4g_db_create(s4g_db** out, int device, const uint8_t* codes, const uint64_t* offsets, uint32_t n_seqs) {
    return s4g_db_create_tiled(out, device, codes, offsets, n_seqs, 0);
}

S4G_API int s4g_db_destroy(s4g_db* db) {
    if (!db) return S4G_OK;
    DeviceGuard guard(db->device);
    {   // launches that still read the tiles must have finished before they are freed
        DeviceCtx* cx = device_ctx(db->device);
        std::lock_guard<std::mutex> lk(cx->mu);
        if (cx->last_use) (void)hipEventSynchronize(cx->last_use);
    }
    for (Tile& t : db->tiles) free_tile(t, device_ctx(db->device));
    if (!db->seq_tables_in_block) { (void)hipFree(db->d_seq_addr); (void)hipFree(db->d_seq_len); }
    delete db;
    return S4G_OK;
}

S4G_API int s4g_db_info(const s4g_db* db, uint32_t* n_seqs, uint64_t* residues, uint64_t* padded_residues, uint64_t* hbm_bytes) {
    if (!db) return fail(S4G_ERR_INVALID, "db is NULL");
    if (n_seqs) *n_seqs = db->n_seqs;
    if (residues) *residues = db->residues;
    if (padded_residues) *padded_residues = db->padded_residues;
    if (hbm_bytes) {
        DeviceCtx* cx = device_ctx(db->device);
        *hbm_bytes = db->hbm_bytes + cx->ws_bytes + cx->ws_side_bytes + cx->tile_scores_bytes + cx->arena_bytes;
    }
    return S4G_OK;
}

// ---- packed on-disk format: the length-sorted, lane-transposed tiles exactly as they sit in HBM,
// so a database is parsed and packed once and later runs only stream it to the device.
namespace {
constexpr char kDbMagic[8] = {'S', '4', 'G', 'D', 'B', '0', '0', '2'};
struct TileHeader { uint32_t n_seqs, n_groups, max_len4, pad; uint64_t residues, padded_residues, pos0, dwords; };

bool put(FILE* f, const void* p, size_t bytes) { return bytes == 0 || fwrite(p, 1, bytes, f) == bytes; }
bool get(FILE* f, void* p, size_t bytes) { return bytes == 0 || fread(p, 1, bytes, f) == bytes; }

template <typename T>
bool put_dev(FILE* f, const T* d, size_t n) {
    std::vector<T> h(n ? n : 1);
    if (n && hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) return false;
    return put(f, h.data(), n * sizeof(T));
}
template <typename T>
bool get_dev(FILE* f, T** d, size_t n, uint64_t* hbm) {
    std::vector<T> h(n ? n : 1);
    if (!get(f, h.data(), n * sizeof(T))) return false;
    if (hipMalloc((void**)d, (n ? n : 1) * sizeof(T)) != hipSuccess) return false;
    *hbm += n * sizeof(T);
    return n == 0 || hipMemcpy(*d, h.data(), n * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
}
}  // namespace

S4G_API int s4g_db_save(const s4g_db* db, const char* path) {
    if (!db || !path) return fail(S4G_ERR_INVALID, "NULL argument");
    DeviceGuard guard(db->device);
    FILE* f = fopen(path, "wb");
    if (!f) return fail(S4G_ERR_INVALID, "cannot open '%s' for writing", path);
    const uint32_t nt = (uint32_t)db->tiles.size();
    bool ok = put(f, kDbMagic, 8) && put(f, &db->n_seqs, 4) && put(f, &nt, 4) && put(f, &db->residues, 8) &&
              put(f, &db->padded_residues, 8) && put(f, db->len.data(), (size_t)db->n_seqs * 4) &&
              put(f, db->pos_of_id.data(), (size_t)db->n_seqs * 4);
    for (const Tile& t : db->tiles) {
        if (!ok) break;
        const uint64_t dwords = t.padded_residues / 4;
        TileHeader h = {t.n_seqs, t.n_groups, t.max_len4, 0, t.residues, t.padded_residues, t.pos0, dwords};
        ok = put(f, &h, sizeof h) && put(f, t.h_grp_off.data(), (size_t)t.n_groups * 8) &&
             put_dev(f, t.d_grp_len4, t.n_groups) && put_dev(f, t.d_perm_global, (size_t)t.n_groups * 64) &&
             put_dev(f, t.d_perm_local, (size_t)t.n_groups * 64) && put_dev(f, t.d_col_ids, t.n_seqs) && put_dev(f, t.d_res, dwords);
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? S4G_OK : fail(S4G_ERR_HIP, "writing '%s' failed", path);
}

S4G_API int s4g_db_load(s4g_db** out, int device, const char* path) {
    if (!out || !path) return fail(S4G_ERR_INVALID, "NULL argument");
    int have = device_count_raw();
    if (have <= 0) return fail(S4G_ERR_NO_DEVICE, "no HIP device available (libs4g_hip has no CPU fallback)");
    if (device < 0 || device >= have) return fail(S4G_ERR_INVALID, "device %d out of range [0,%d)", device, have);
    FILE* f = fopen(path, "rb");
    if (!f) return fail(S4G_ERR_INVALID, "cannot open '%s'", path);
    DeviceGuard guard(device);
    s4g_db* db = new s4g_db();
    db->device = device;
    char magic[8];
    uint32_t nt = 0;
    bool ok = get(f, magic, 8) && memcmp(magic, kDbMagic, 8) == 0 && get(f, &db->n_seqs, 4) && get(f, &nt, 4) &&
              get(f, &db->residues, 8) && get(f, &db->padded_residues, 8);
    if (ok) {
        db->len.resize(db->n_seqs); db->pos_of_id.resize(db->n_seqs);
        ok = get(f, db->len.data(), (size_t)db->n_seqs * 4) && get(f, db->pos_of_id.data(), (size_t)db->n_seqs * 4);
    }
    for (uint32_t i = 0; ok && i < nt; ++i) {
        TileHeader h;
        ok = get(f, &h, sizeof h);
        if (!ok) break;
        db->tiles.emplace_back();
        Tile& t = db->tiles.back();
        t.n_seqs = h.n_seqs; t.n_groups = h.n_groups; t.max_len4 = h.max_len4; t.residues = h.residues;
        t.padded_residues = h.padded_residues; t.pos0 = h.pos0;
        t.h_grp_off.resize(h.n_groups);
        ok = get(f, t.h_grp_off.data(), (size_t)h.n_groups * 8);
        if (ok) {
            ok = hipMalloc((void**)&t.d_grp_off, (h.n_groups ? h.n_groups : 1) * 8) == hipSuccess &&
                 (h.n_groups == 0 || hipMemcpy(t.d_grp_off, t.h_grp_off.data(), (size_t)h.n_groups * 8, hipMemcpyHostToDevice) == hipSuccess);
            db->hbm_bytes += (uint64_t)h.n_groups * 8;
        }
        ok = ok && get_dev(f, &t.d_grp_len4, h.n_groups, &db->hbm_bytes) && get_dev(f, &t.d_perm_global, (size_t)h.n_groups * 64, &db->hbm_bytes) &&
             get_dev(f, &t.d_perm_local, (size_t)h.n_groups * 64, &db->hbm_bytes) && get_dev(f, &t.d_col_ids, h.n_seqs, &db->hbm_bytes) &&
             get_dev(f, &t.d_res, h.dwords, &db->hbm_bytes);
        db->max_tile_seqs = std::max(db->max_tile_seqs, t.n_seqs);
    }
    fclose(f);
    if (!ok) { s4g_db_destroy(db); return fail(S4G_ERR_INVALID, "'%s' is not a readable s4g database file", path); }
    *out = db;
    return S4G_OK;
}

S4G_API int s4g_db_tiles(const s4g_db* db, uint32_t* n_tiles, uint32_t* tile_seqs, uint32_t cap) {
    if (!db || !n_tiles) return fail(S4G_ERR_INVALID, "NULL argument");
    *n_tiles = (uint32_t)db->tiles.size();
    if (tile_seqs) for (uint32_t i = 0; i < cap && i < db->tiles.size(); ++i) tile_seqs[i] = db->tiles[i].n_seqs;
    return S4G_OK;
}

// ---------------------------------------------------------------------------
S4G_API int s4g_queries_create(s4g_queries** out, int device, const uint8_t* codes, const uint64_t* offsets, uint32_t n) {
    if (!out || !offsets || (!codes && n && offsets[n])) return fail(S4G_ERR_INVALID, "NULL argument");
    int have = device_count_raw();
    if (have <= 0) return fail(S4G_ERR_NO_DEVICE, "no HIP device available (libs4g_hip has no CPU fallback)");
    if (device < 0 || device >= have) return fail(S4G_ERR_INVALID, "device %d out of range [0,%d)", device, have);
    DeviceGuard guard(device);
    s4g_queries* q = new s4g_queries();
    q->device = device;
    q->n = n;
    q->len.resize(n);
    std::vector<uint64_t> off(n ? n : 1);
    uint64_t bytes = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t l = offsets[i + 1] - offsets[i];
        if (offsets[i + 1] < offsets[i] || l > 0x7FFFFFF0ull) { delete q; return fail(S4G_ERR_INVALID, "bad offsets at %u", i); }
        q->len[i] = (uint32_t)l;
        q->max_len = std::max(q->max_len, (uint32_t)l);
        off[i] = bytes;
        bytes += (l + kQueryAlign - 1) / kQueryAlign * kQueryAlign + kQueryAlign;
    }
    q->off.assign(offsets, offsets + n + 1);
    q->codes.assign(codes ? codes : (const uint8_t*)"", codes ? codes + offsets[n] : (const uint8_t*)"");
    std::vector<uint8_t> padded(bytes ? bytes : 1, (uint8_t)kPadCode);
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < q->len[i]; ++j) {
            uint8_t c = codes[offsets[i] + j];
            if (c >= 26) { delete q; return fail(S4G_ERR_INVALID, "residue code %u out of range in query %u", c, i); }
            padded[off[i] + j] = c;
        }
    hipError_t e;
    if ((e = hipMalloc((void**)&q->d_codes, padded.size())) != hipSuccess ||
        (e = hipMalloc((void**)&q->d_off, off.size() * 8)) != hipSuccess ||
        (e = hipMalloc((void**)&q->d_len, (n ? n : 1) * 4)) != hipSuccess ||
        (e = hipMemcpy(q->d_codes, padded.data(), padded.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(q->d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice)) != hipSuccess ||
        (n && (e = hipMemcpy(q->d_len, q->len.data(), (size_t)n * 4, hipMemcpyHostToDevice)) != hipSuccess)) {
        s4g_queries_destroy(q);
        return fail(S4G_ERR_HIP, "query upload failed: %s", hipGetErrorString(e));
    }
    *out = q;
    return S4G_OK;
}

S4G_API int s4g_queries_destroy(s4g_queries* q) {
    if (!q) return S4G_OK;
    DeviceGuard guard(q->device);
    (void)hipFree(q->d_codes); (void)hipFree(q->d_off); (void)hipFree(q->d_len);
    delete q;
    return S4G_OK;
}

S4G_API int s4g_scan_cells(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, uint64_t* cells) {
    if (!db || !q || !cells) return fail(S4G_ERR_INVALID, "NULL argument");
    if ((uint64_t)q_begin + q_count > q->n) return fail(S4G_ERR_INVALID, "query range out of bounds");
    uint64_t lq = 0;
    for (uint32_t i = 0; i < q_count; ++i) lq += q->len[q_begin + i];
    *cells = lq * db->residues;
    return S4G_OK;
}

// ---------------------------------------------------------------------------
namespace {

constexpr int kStripRows = 32;      // sw_scan_i32
constexpr int kStripRows16 = 32;    // sw_scan_i16x2
constexpr uint64_t kWorkspaceBudget = 24ull << 30;   // strip-boundary workspace cap; long-sequence tiles run on a smaller grid
constexpr uint64_t kTileScoreBudget = 4ull << 30;    // s4g_scan_topk: bytes of the [query batch][tile columns] buffer

// Work of one tile launch: which queries run in which kernel.  Tiers, fastest first; a query goes to the first one that
// is EXACT for it (score bound = sum_i max(0, max_b M[q_i][b]); see sw_kernels.hip.h for each window):
//   drift  sw_scan_pkd          bound + B + (rows + tile columns) * ext <= 0x7BFF      6.5 ops / pair of cells
//   m3     sw_scan_pk<I16M3>    bound + B <= 0x7BFF   (or ArithF16: bound <= 4096)     7.5
//   pk     sw_scan_pk<I16>      bound <= 32767                                          9
//   wide   sw_scan_i32          anything                                                7.5 per single cell
struct TileWork {
    uint32_t off_drift = 0, n_drift = 0;   // offsets / pair counts into Plan::flat (pairs = 2 ids each)
    uint32_t off_m3 = 0, n_m3 = 0;
    uint32_t off_pk = 0, n_pk = 0;
    uint32_t off_wide = 0, n_wide = 0;     // single ids
    uint32_t prio_cost[3] = {0, 0, 0};     // per packed tier: rows * len4 at or above which an item runs at raised wave priority
};

struct Plan {
    std::vector<TileWork> tiles;    // one per database tile
    std::vector<uint32_t> flat;     // all id lists, uploaded once per call
    bool f16_arith = false;         // m3 tier runs ArithF16 instead of ArithI16M3
    uint32_t bias = 0;              // bias B of the maximum3-based tiers
    uint32_t grid_i32 = 0, grid_i16 = 0, grid_m3 = 0, grid_drift = 0;   // full persistent grids
    uint64_t ws_bytes = 0;          // workspace needed for the largest tile launch
    uint64_t n_by_tier[4] = {0, 0, 0, 0};   // queries per tier, summed over tiles (reporting)
};

int device_props(DeviceCtx* cx) {
    if (!cx->have_prop) {
        HIP_TRY(hipGetDeviceProperties(&cx->prop, cx->device));
        cx->have_prop = true;
    }
    return S4G_OK;
}

int grid_for(const s4g_db* db, const void* kernel, uint32_t* cache) {
    if (*cache == 0) {
        DeviceCtx* cx = device_ctx(db->device);
        int rc = device_props(cx);
        if (rc) return rc;
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0));
        if (per_cu < 1) per_cu = 1;
        *cache = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)per_cu;
    }
    return S4G_OK;
}

inline uint64_t ws_slot_elems(const Tile& t) { return 2ull * (t.max_len4 ? t.max_len4 : 1) * 64 * 4; }   // int32 per wave slot

// grid of a tile launch: the full persistent grid unless the tile's sequences are so long that the per-wave
// strip-boundary workspace would exceed the budget
inline uint32_t tile_grid(const Tile& t, uint32_t full_grid) {
    const uint64_t per_block = ws_slot_elems(t) * 4 * kWavesPerBlock;
    const uint64_t fit = kWorkspaceBudget / per_block;
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(full_grid, fit));
}

// per-query upper bound of any reachable score under this scorer (host, O(residues), cached per scorer)
void ensure_bounds(const s4g_queries* q, const s4g_scorer* scorer) {
    if (q->bound_scorer == scorer->serial) return;
    q->score_bound.assign(q->n, 0);
    for (uint32_t i = 0; i < q->n; ++i) {
        uint64_t b = 0;
        const uint8_t* c = q->codes.data() + q->off[i];
        for (uint32_t j = 0; j < q->len[i]; ++j) b += (uint64_t)scorer->row_max[c[j]];
        q->score_bound[i] = b;
    }
    q->bound_scorer = scorer->serial;
}

int make_plan(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* sc,
              uint32_t flags, Plan* p) {
    DeviceCtx* cx = device_ctx(db->device);
    int rc = grid_for(db, (const void*)sw_scan_i32<kStripRows>, &cx->grid_i32);
    if (rc) return rc;
    rc = grid_for(db, (const void*)sw_scan_pk<ArithI16, kStripRows16>, &cx->grid_i16);
    if (rc) return rc;
    rc = grid_for(db, (const void*)sw_scan_pk<ArithI16M3, kStripRows16>, &cx->grid_f16);
    if (rc) return rc;
    rc = grid_for(db, (const void*)sw_scan_pkd<kStripRows16>, &cx->grid_drift);
    if (rc) return rc;
    p->grid_i32 = cx->grid_i32; p->grid_i16 = cx->grid_i16; p->grid_m3 = cx->grid_f16; p->grid_drift = cx->grid_drift;
    const bool gaps16 = sc->open < 32768 && sc->ext < 32768;
    p->f16_arith = (flags & S4G_SCAN_F16_ARITH) != 0;
    // bias of the maximum3-on-bit-patterns tiers: keeps diag+s, h-open, E-ext (and H~(0,-1) = B - ext in the drift frame) >= 0
    p->bias = (uint32_t)std::max(std::max(-kPadScore, -sc->min_score), std::max(sc->open, 2 * sc->ext));
    const uint64_t cap = 0x7BFF;     // largest int16 bit pattern that is a finite f16
    const uint64_t m3_window = p->f16_arith ? 4096 : (p->bias < cap ? cap - p->bias : 0);
    const bool m3_gaps = p->f16_arith ? (sc->open <= 2048 && sc->ext <= 2048) : gaps16;
    const bool no_m3 = (flags & (S4G_SCAN_FORCE_I32 | S4G_SCAN_NO_F16)) != 0;
    const bool no_drift = no_m3 || p->f16_arith || (flags & S4G_SCAN_NO_DRIFT) != 0;

    std::vector<uint32_t> ids[4];
    auto add_pairs = [&](std::vector<uint32_t>& v, uint32_t* off, uint32_t* n) {
        // pair queries of similar length (the pair runs to the longer one); longest pairs first
        std::stable_sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) { return q->len[a] > q->len[b]; });
        *off = (uint32_t)p->flat.size();
        for (size_t i = 0; i < v.size(); i += 2) {
            p->flat.push_back(v[i]);
            p->flat.push_back(i + 1 < v.size() ? v[i + 1] : 0xFFFFFFFFu);
        }
        *n = (uint32_t)((v.size() + 1) / 2);
    };
    const uint32_t full = std::max(std::max(p->grid_i32, p->grid_i16), std::max(p->grid_m3, p->grid_drift));
    for (const Tile& t : db->tiles) {
        for (auto& v : ids) v.clear();
        const uint64_t cols = (uint64_t)t.max_len4 * 4 + 8;
        for (uint32_t i = 0; i < q_count; ++i) {
            const uint32_t id = q_begin + i;
            const uint64_t b = q->score_bound[id];
            const uint64_t rows = (uint64_t)q->len[id] + 2 * kStripRows16;          // padded rows of the pair's strips
            int tier;
            if (!no_drift && gaps16 && b + p->bias + (rows + cols) * (uint64_t)sc->ext <= cap) tier = 0;
            else if (!no_m3 && m3_gaps && b <= m3_window) tier = 1;
            else if (!(flags & S4G_SCAN_FORCE_I32) && gaps16 && b <= 32767) tier = 2;
            else tier = 3;
            ids[tier].push_back(id);
            p->n_by_tier[tier]++;
        }
        // Every tier is its own launch, and a launch can never finish before its largest item does (one wave walks the
        // whole (query pair, group) DP).  With few queries the tiers' launches become short and tail-dominated, so
        // fold a tier into the next slower one whenever the estimate says one launch beats two.  Units: VALU ops.
        {
            static const double ops[4] = {6.5, 7.5, 9.0, 15.0};     // per pair of cells (int32: two single-query items)
            const double waves = (double)tile_grid(t, full) * kWavesPerBlock;
            const double cols_sum = (double)t.padded_residues / 64.0;           // sum over groups of padded length
            const double cols_max = (double)t.max_len4 * 4.0;
            auto launch_time = [&](const std::vector<uint32_t>& v, int tier) -> double {
                if (v.empty()) return 0.0;
                double rows_sum = 0, rows_max = 0;
                for (uint32_t id : v) { rows_sum += q->len[id]; rows_max = std::max<double>(rows_max, q->len[id]); }
                if (tier < 3) rows_sum *= 0.5;                                   // two queries share an item
                return ops[tier] * std::max(rows_sum * cols_sum / waves, rows_max * cols_max);
            };
            for (int tier = 0; tier < 3; ++tier) {
                if (ids[tier].empty() || ids[tier + 1].empty()) continue;
                std::vector<uint32_t> merged(ids[tier + 1]);
                merged.insert(merged.end(), ids[tier].begin(), ids[tier].end());
                if (launch_time(merged, tier + 1) <= launch_time(ids[tier], tier) + launch_time(ids[tier + 1], tier + 1)) {
                    p->n_by_tier[tier] -= ids[tier].size();
                    p->n_by_tier[tier + 1] += ids[tier].size();
                    ids[tier + 1].swap(merged);
                    ids[tier].clear();
                }
            }
        }
        TileWork w;
        // critical-path items: larger than a quarter of the per-wave share of their launch (only then can unfair issue
        // arbitration between co-resident waves stretch the launch)
        for (int tier = 0; tier < 3; ++tier) {
            if (ids[tier].empty()) continue;
            double rows_sum = 0;
            for (uint32_t id : ids[tier]) rows_sum += q->len[id];
            const double waves = (double)tile_grid(t, full) * kWavesPerBlock;
            const double share = 0.5 * rows_sum * ((double)t.padded_residues / 256.0) / waves;       // (rows * len4) per wave
            w.prio_cost[tier] = (uint32_t)std::min<double>(4e9, std::max<double>(1.0, share / 4.0));
        }
        add_pairs(ids[0], &w.off_drift, &w.n_drift);
        add_pairs(ids[1], &w.off_m3, &w.n_m3);
        add_pairs(ids[2], &w.off_pk, &w.n_pk);
        w.off_wide = (uint32_t)p->flat.size(); w.n_wide = (uint32_t)ids[3].size();
        p->flat.insert(p->flat.end(), ids[3].begin(), ids[3].end());
        p->tiles.push_back(w);
        // workspace: one slot per wave of the largest launch of this tile (grids are capped by the item count)
        const uint64_t most = std::max(std::max<uint64_t>(w.n_drift, w.n_m3), std::max<uint64_t>(w.n_pk, w.n_wide));
        const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>(tile_grid(t, full), ((uint64_t)t.n_groups * most + kWavesPerBlock - 1) / kWavesPerBlock));
        p->ws_bytes = std::max<uint64_t>(p->ws_bytes, ws_slot_elems(t) * 4 * kWavesPerBlock * blocks);
    }
    return S4G_OK;
}

int ensure_scratch(const s4g_db* db, const s4g_scorer* sc, const Plan& p, hipStream_t stream) {
    DeviceCtx* cx = device_ctx(db->device);
    if (cx->ws_bytes < p.ws_bytes) {
        if (cx->d_ws) HIP_TRY(hipFree(cx->d_ws));
        cx->d_ws = nullptr; cx->ws_bytes = 0;
        HIP_TRY(hipMalloc((void**)&cx->d_ws, p.ws_bytes));
        cx->ws_bytes = p.ws_bytes;
    }
    if (!cx->d_counter) HIP_TRY(hipMalloc((void**)&cx->d_counter, 256));
    if (!cx->d_mat) HIP_TRY(hipMalloc((void**)&cx->d_mat, 1024));
    const uint64_t need = p.flat.size();
    if (cx->qlist_cap < need) {
        if (cx->d_qlist) HIP_TRY(hipFree(cx->d_qlist));
        cx->d_qlist = nullptr; cx->qlist_cap = 0;
        HIP_TRY(hipMalloc((void**)&cx->d_qlist, need * 4));
        cx->qlist_cap = need;
    }
    HIP_TRY(hipMemcpyAsync(cx->d_mat, sc->mat32, 1024, hipMemcpyHostToDevice, stream));
    if (need) HIP_TRY(hipMemcpyAsync(cx->d_qlist, p.flat.data(), need * 4, hipMemcpyHostToDevice, stream));
    return S4G_OK;
}

// One tile: one launch per non-empty tier.  perm/stride select the output layout: original ids into the caller's
// [q][n_seqs] matrix, or tile columns.
int launch_tile(const s4g_db* db, const Tile& t, const TileWork& w, const s4g_queries* q, const s4g_scorer* scorer, const Plan& p,
                uint32_t q_begin, int32_t* d_scores, uint64_t stride, const uint32_t* perm, hipStream_t stream, bool side = false) {
    if (t.n_seqs == 0) return S4G_OK;
    DeviceCtx* cx = device_ctx(db->device);
    int32_t* const ws = side ? cx->d_ws_side : cx->d_ws;
    unsigned* const counters = cx->d_counter + (side ? 8 : 0);
    // a persistent grid larger than the number of items only launches blocks that exit at once
    auto grid_of = [&](uint32_t full, uint64_t items) {
        return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(tile_grid(t, full), (items + kWavesPerBlock - 1) / kWavesPerBlock));
    };
    const uint64_t most = std::max(std::max<uint64_t>(w.n_drift, w.n_m3), std::max<uint64_t>(w.n_pk, w.n_wide));
    if ((uint64_t)t.n_groups * most >= 0xFFFFFFFFull)
        return fail(S4G_ERR_INVALID, "too many work items in one tile launch; split the query range");
    HIP_TRY(hipMemsetAsync(counters, 0, 16, stream));
    ScanArgs16 a;
    a.db_res = t.d_res; a.grp_off = t.d_grp_off; a.grp_len4 = t.d_grp_len4; a.grp_perm = perm;
    a.q_codes = q->d_codes; a.q_off = q->d_off; a.q_len = q->d_len;
    a.mat = cx->d_mat;
    a.scores = d_scores; a.score_stride = stride;
    a.ws = ws; a.ws_slot = ws_slot_elems(t);
    a.n_groups = t.n_groups; a.q_begin = q_begin;
    a.open = (uint32_t)scorer->open; a.ext = (uint32_t)scorer->ext;
    if (w.n_drift) {
        a.pairs = cx->d_qlist + w.off_drift; a.n_pairs = w.n_drift; a.counter = counters + 3; a.prio_cost = w.prio_cost[0];
        a.floor_bits = p.bias | (p.bias << 16);
        hipLaunchKernelGGL((sw_scan_pkd<kStripRows16>), dim3(grid_of(p.grid_drift, (uint64_t)t.n_groups * w.n_drift)), dim3(kBlock), 0, stream, a);
        HIP_TRY(hipGetLastError());
    }
    if (w.n_m3) {
        a.pairs = cx->d_qlist + w.off_m3; a.n_pairs = w.n_m3; a.counter = counters + 2; a.prio_cost = w.prio_cost[1];
        const uint32_t g3 = grid_of(p.grid_m3, (uint64_t)t.n_groups * w.n_m3);
        if (p.f16_arith) {
            a.floor_bits = 0xE800E800u;      // (-2048.0h, -2048.0h)
            hipLaunchKernelGGL((sw_scan_pk<ArithF16, kStripRows16>), dim3(g3), dim3(kBlock), 0, stream, a);
        } else {
            a.floor_bits = p.bias | (p.bias << 16);
            hipLaunchKernelGGL((sw_scan_pk<ArithI16M3, kStripRows16>), dim3(g3), dim3(kBlock), 0, stream, a);
        }
        HIP_TRY(hipGetLastError());
    }
    if (w.n_pk) {
        a.pairs = cx->d_qlist + w.off_pk; a.n_pairs = w.n_pk; a.counter = counters; a.prio_cost = w.prio_cost[2];
        a.floor_bits = 0;
        hipLaunchKernelGGL((sw_scan_pk<ArithI16, kStripRows16>), dim3(grid_of(p.grid_i16, (uint64_t)t.n_groups * w.n_pk)), dim3(kBlock), 0, stream, a);
        HIP_TRY(hipGetLastError());
    }
    if (w.n_wide) {
        ScanArgs b;
        b.db_res = t.d_res; b.grp_off = t.d_grp_off; b.grp_len4 = t.d_grp_len4; b.grp_perm = perm;
        b.q_codes = q->d_codes; b.q_off = q->d_off; b.q_len = q->d_len;
        b.mat = cx->d_mat;
        b.scores = d_scores; b.score_stride = stride;
        b.ws = ws; b.ws_slot = ws_slot_elems(t);
        b.counter = counters + 1;
        b.n_groups = t.n_groups; b.q_begin = q_begin; b.q_count = w.n_wide;
        b.q_list = cx->d_qlist + w.off_wide;
        b.open = (uint32_t)scorer->open; b.ext = (uint32_t)scorer->ext;
        hipLaunchKernelGGL(sw_scan_i32<kStripRows>, dim3(grid_of(p.grid_i32, (uint64_t)t.n_groups * w.n_wide)), dim3(kBlock), 0, stream, b);
        HIP_TRY(hipGetLastError());
    }
    return S4G_OK;
}

// The scratch buffers (workspace, work-queue counter, query lists) are shared by all launches on this database:
// a launch on another stream must not start before the previous one has finished with them.
int scratch_acquire(const s4g_db* db, hipStream_t stream) {
    DeviceCtx* cx = device_ctx(db->device);
    if (!cx->last_use) HIP_TRY(hipEventCreateWithFlags(&cx->last_use, hipEventDisableTiming));
    else HIP_TRY(hipStreamWaitEvent(stream, cx->last_use, 0));
    return S4G_OK;
}
int scratch_release(const s4g_db* db, hipStream_t stream) {
    DeviceCtx* cx = device_ctx(db->device);
    HIP_TRY(hipEventRecord(cx->last_use, stream));
    return S4G_OK;
}

int check_scan_args(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t q_begin, uint32_t q_count) {
    if (!db || !q || !scorer) return fail(S4G_ERR_INVALID, "NULL argument");
    if (db->device != q->device) return fail(S4G_ERR_INVALID, "db on device %d, queries on device %d", db->device, q->device);
    if ((uint64_t)q_begin + q_count > q->n) return fail(S4G_ERR_INVALID, "query range out of bounds");
    return S4G_OK;
}

}  // namespace

namespace {
int scan_as_pairs(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* scorer,
                  int32_t* d_scores, hipStream_t stream, uint32_t flags, bool* done);
}

S4G_API int s4g_scan_plan(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, char* kernel_name,
                          size_t kernel_name_cap, uint32_t* grid_blocks, uint32_t* block_threads,
                          uint32_t* strip_rows, uint64_t* workspace_bytes) {
    if (!db || !q || !scorer) return fail(S4G_ERR_INVALID, "NULL argument");
    DeviceGuard guard(db->device);
    DeviceCtx* cx = device_ctx(db->device);
    std::lock_guard<std::mutex> lk(cx->mu);
    ensure_bounds(q, scorer);
    Plan p;
    int rc = make_plan(db, q, 0, q->n, scorer, 0, &p);
    if (rc) return rc;
    // report the kernel that carries most of the queries
    int top = 0;
    for (int k = 1; k < 4; ++k) if (p.n_by_tier[k] > p.n_by_tier[top]) top = k;
    const char* names[4] = {"sw_scan_pkd<32>", p.f16_arith ? "sw_scan_pk<ArithF16,32>" : "sw_scan_pk<ArithI16M3,32>", "sw_scan_pk<ArithI16,32>",
                            "sw_scan_i32<32>"};
    const uint32_t grids[4] = {p.grid_drift, p.grid_m3, p.grid_i16, p.grid_i32};
    if (kernel_name && kernel_name_cap) snprintf(kernel_name, kernel_name_cap, "%s", names[top]);
    if (grid_blocks) *grid_blocks = grids[top];
    if (block_threads) *block_threads = kBlock;
    if (strip_rows) *strip_rows = top == 3 ? kStripRows : kStripRows16;
    if (workspace_bytes) *workspace_bytes = p.ws_bytes;
    return S4G_OK;
}

S4G_API int s4g_scan_flags(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count,
                           const s4g_scorer* scorer, int32_t* d_scores, void* stream_, uint32_t flags) {
    int rc = check_scan_args(db, q, scorer, q_begin, q_count);
    if (rc) return rc;
    if (q_count == 0 || db->n_seqs == 0) return S4G_OK;
    if (!d_scores) return fail(S4G_ERR_INVALID, "d_scores is NULL");
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(db->device);
    DeviceCtx* cx = device_ctx(db->device);
    std::lock_guard<std::mutex> lk(cx->mu);
    {   // latency-bound call (few queries against a small database, the per-query alignDatabase pattern)?  Then every
        // (query, target) pair gets a wave of its own instead of 64 targets sharing one: see scan_as_pairs.
        bool as_pairs = false;
        rc = scan_as_pairs(db, q, q_begin, q_count, scorer, d_scores, stream, flags, &as_pairs);
        if (rc || as_pairs) return rc;
    }
    ensure_bounds(q, scorer);
    Plan p;
    rc = make_plan(db, q, q_begin, q_count, scorer, flags, &p);
    if (rc) return rc;
    rc = scratch_acquire(db, stream);
    if (rc) return rc;
    rc = ensure_scratch(db, scorer, p, stream);
    if (rc) return rc;
    // long-sequence tiles first, on the side stream with their own workspace: their few huge items then run underneath
    // the main tiles' launches instead of as a tail-dominated launch of their own
    bool forked = false;
    const uint32_t full = std::max(std::max(p.grid_i32, p.grid_i16), std::max(p.grid_m3, p.grid_drift));
    if (db->tiles.size() > 1) {
        uint64_t need = 0;
        for (size_t ti = 0; ti < db->tiles.size(); ++ti) {
            const Tile& t = db->tiles[ti];
            if ((uint64_t)t.max_len4 * 4 <= kLongSeq) continue;
            const TileWork& w = p.tiles[ti];
            const uint64_t most = std::max(std::max<uint64_t>(w.n_drift, w.n_m3), std::max<uint64_t>(w.n_pk, w.n_wide));
            const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>(tile_grid(t, full), ((uint64_t)t.n_groups * most + kWavesPerBlock - 1) / kWavesPerBlock));
            need = std::max<uint64_t>(need, ws_slot_elems(t) * 4 * kWavesPerBlock * blocks);
        }
        if (need) {
            if (!cx->side_stream) {
                HIP_TRY(hipStreamCreateWithFlags(&cx->side_stream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&cx->side_fork, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&cx->side_join, hipEventDisableTiming));
            }
            if (cx->ws_side_bytes < need) {
                if (cx->d_ws_side) HIP_TRY(hipFree(cx->d_ws_side));
                cx->d_ws_side = nullptr; cx->ws_side_bytes = 0;
                HIP_TRY(hipMalloc((void**)&cx->d_ws_side, need));
                cx->ws_side_bytes = need;
            }
            HIP_TRY(hipEventRecord(cx->side_fork, stream));                 // after the uploads of this call
            HIP_TRY(hipStreamWaitEvent(cx->side_stream, cx->side_fork, 0));
            for (size_t ti = 0; ti < db->tiles.size(); ++ti) {
                const Tile& t = db->tiles[ti];
                if ((uint64_t)t.max_len4 * 4 <= kLongSeq) continue;
                rc = launch_tile(db, t, p.tiles[ti], q, scorer, p, q_begin, d_scores, db->n_seqs, t.d_perm_global, cx->side_stream, true);
                if (rc) return rc;
            }
            HIP_TRY(hipEventRecord(cx->side_join, cx->side_stream));
            forked = true;
        }
    }
    for (size_t ti = 0; ti < db->tiles.size(); ++ti) {
        const Tile& t = db->tiles[ti];
        if (forked && (uint64_t)t.max_len4 * 4 > kLongSeq) continue;
        rc = launch_tile(db, t, p.tiles[ti], q, scorer, p, q_begin, d_scores, db->n_seqs, t.d_perm_global, stream);
        if (rc) return rc;
    }
    if (forked) HIP_TRY(hipStreamWaitEvent(stream, cx->side_join, 0));
    return scratch_release(db, stream);
}

S4G_API int s4g_scan(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count,
                     const s4g_scorer* scorer, int32_t* d_scores, void* stream) {
    return s4g_scan_flags(db, q, q_begin, q_count, scorer, d_scores, stream, 0);
}

S4G_API int s4g_scan_host_flags(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count,
                                const s4g_scorer* scorer, int32_t* h_scores, uint32_t flags);

S4G_API int s4g_scan_host(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count,
                          const s4g_scorer* scorer, int32_t* h_scores) {
    return s4g_scan_host_flags(db, q, q_begin, q_count, scorer, h_scores, 0);
}

S4G_API int s4g_scan_host_flags(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count,
                                const s4g_scorer* scorer, int32_t* h_scores, uint32_t flags) {
    if (!db || !q || !scorer) return fail(S4G_ERR_INVALID, "NULL argument");
    if ((uint64_t)q_begin + q_count > q->n) return fail(S4G_ERR_INVALID, "query range out of bounds");
    if (q_count == 0 || db->n_seqs == 0) return S4G_OK;
    if (!h_scores) return fail(S4G_ERR_INVALID, "h_scores is NULL");
    DeviceGuard guard(db->device);
    DevBuf<int32_t> d;
    const size_t bytes = (size_t)q_count * db->n_seqs * 4;
    HIP_TRY(d.alloc((size_t)q_count * db->n_seqs));
    int rc = s4g_scan_flags(db, q, q_begin, q_count, scorer, d.p, nullptr, flags);
    if (rc != S4G_OK) return rc;
    HIP_TRY(hipMemcpy(h_scores, d.p, bytes, hipMemcpyDeviceToHost));
    return S4G_OK;
}

// ---------------------------------------------------------------------------
// Pair lists (the latency path, sw_pair.hip.h): explicit (query, target) pairs, one wave per pair.
namespace {

constexpr int kPairClasses = 6;
constexpr int kPairRpl[kPairClasses] = {2, 4, 6, 8, 12, 16};
inline int pair_class(uint32_t lq) { int c = 0; while (64u * kPairRpl[c] < lq) ++c; return c; }
// Measured rates (tools/pair_latency.py, profiles/r01_pair): cells per second a chip full of pair waves sustains and
// the cells per second ONE wave makes on one pair, against the scan's chip rate and the cells per second of one
// inter-sequence lane (which bounds a scan launch from below: one wave walks a whole (query pair, group) DP).
constexpr double kPairChipRate = 2.5e12, kPairWaveRate = 2.4e9 / 0.7, kPairCallOverhead = 60e-6;
constexpr double kScanChipRate = 10e12, kScanLaneRate = 2.4e9 / 37.0;

template <int RPL>
int launch_pairs(DeviceCtx* cx, PairArgs a, uint32_t* grid_cache, hipStream_t stream) {
    if (*grid_cache == 0) {
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_pair_scan<RPL>, 64, 0));
        *grid_cache = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
    }
    const uint32_t grid = std::max(1u, std::min(*grid_cache, a.n_pairs));
    hipLaunchKernelGGL(sw_pair_scan<RPL>, dim3(grid), dim3(64), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return S4G_OK;
}

// device tables indexed by original target id (caller holds cx->mu)
int ensure_seq_tables(const s4g_db* db) {
    if (db->d_seq_addr || db->n_seqs == 0) return S4G_OK;
    std::vector<uint64_t> addr(db->n_seqs);
    db->id_of_pos.resize(db->n_seqs);
    size_t ti = 0;
    // walk ids in position order so the tile lookup is a running index
    for (uint32_t id = 0; id < db->n_seqs; ++id) db->id_of_pos[db->pos_of_id[id]] = id;
    for (uint32_t pos = 0; pos < db->n_seqs; ++pos) {
        while (ti + 1 < db->tiles.size() && db->tiles[ti + 1].pos0 <= pos) ++ti;
        const Tile& t = db->tiles[ti];
        const uint32_t lp = (uint32_t)(pos - t.pos0);
        addr[db->id_of_pos[pos]] = (uint64_t)(uintptr_t)(t.d_res + t.h_grp_off[lp >> 6] + (lp & 63));
    }
    HIP_TRY(hipMalloc((void**)&db->d_seq_addr, (size_t)db->n_seqs * 8));
    HIP_TRY(hipMalloc((void**)&db->d_seq_len, (size_t)db->n_seqs * 4));
    HIP_TRY(hipMemcpy(db->d_seq_addr, addr.data(), (size_t)db->n_seqs * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db->d_seq_len, db->len.data(), (size_t)db->n_seqs * 4, hipMemcpyHostToDevice));
    return S4G_OK;
}

// pinned staging for n items (caller holds cx->mu); the previous upload out of it has completed on return
int pair_stage(DeviceCtx* cx, uint64_t n, PairItem** out) {
    if (cx->items_done) HIP_TRY(hipEventSynchronize(cx->items_done));
    else HIP_TRY(hipEventCreateWithFlags(&cx->items_done, hipEventDisableTiming));
    if (cx->items_cap < n) {
        if (cx->h_items) HIP_TRY(hipHostFree(cx->h_items));
        if (cx->d_items) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(cx->d_items)); }
        cx->h_items = nullptr; cx->d_items = nullptr; cx->items_cap = 0;
        const uint64_t cap = std::max<uint64_t>(n + n / 4, 1u << 16);
        HIP_TRY(hipHostMalloc((void**)&cx->h_items, cap * sizeof(PairItem), hipHostMallocDefault));
        HIP_TRY(hipMalloc((void**)&cx->d_items, cap * (sizeof(PairItem) + 4)));       // + int32 scores of the host route
        cx->items_cap = cap;
    }
    *out = cx->h_items;
    return S4G_OK;
}

// uploads the staged items (class-major: cls_begin[c] .. cls_begin[c+1]) and launches one kernel per class
int pair_submit(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t n, const uint32_t* cls_begin,
                int32_t* d_scores, hipStream_t stream) {
    DeviceCtx* cx = device_ctx(db->device);
    int rc = device_props(cx);
    if (rc) return rc;
    rc = ensure_seq_tables(db);
    if (rc) return rc;
    if (!cx->d_counter) HIP_TRY(hipMalloc((void**)&cx->d_counter, 256));
    if (!cx->d_mat) HIP_TRY(hipMalloc((void**)&cx->d_mat, 1024));
    rc = scratch_acquire(db, stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(cx->d_mat, scorer->mat32, 1024, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(cx->d_items, cx->h_items, (size_t)n * sizeof(PairItem), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(cx->items_done, stream));
    HIP_TRY(hipMemsetAsync(cx->d_counter + 24, 0, kPairClasses * 4, stream));
    for (int c = 0; c < kPairClasses; ++c) {
        const uint32_t b = cls_begin[c], e = cls_begin[c + 1];
        if (b == e) continue;
        PairArgs a;
        a.items = cx->d_items + b; a.seq_addr = db->d_seq_addr; a.seq_len = db->d_seq_len;
        a.q_codes = q->d_codes; a.q_off = q->d_off; a.q_len = q->d_len; a.mat = cx->d_mat;
        a.scores = d_scores; a.counter = cx->d_counter + 24 + c; a.n_pairs = e - b;
        a.open = (uint32_t)scorer->open; a.ext = (uint32_t)scorer->ext;
        switch (c) {
            case 0: rc = launch_pairs<2>(cx, a, &cx->grid_pair[0], stream); break;
            case 1: rc = launch_pairs<4>(cx, a, &cx->grid_pair[1], stream); break;
            case 2: rc = launch_pairs<6>(cx, a, &cx->grid_pair[2], stream); break;
            case 3: rc = launch_pairs<8>(cx, a, &cx->grid_pair[3], stream); break;
            case 4: rc = launch_pairs<12>(cx, a, &cx->grid_pair[4], stream); break;
            default: rc = launch_pairs<16>(cx, a, &cx->grid_pair[5], stream); break;
        }
        if (rc) return rc;
    }
    return scratch_release(db, stream);
}

// Explicit list -> host scores.  sel = indices into pair_q / pair_t whose queries have <= kPairMaxRows residues.
// Order inside a class: longest target first at 32-residue granularity (counting sort; a launch cannot end before its
// longest pair does, so that one must not start last).
int score_pairs_device(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, const std::vector<uint32_t>& sel,
                       const uint32_t* pair_q, const uint32_t* pair_t, int32_t* h_scores) {
    DeviceGuard guard(db->device);
    DeviceCtx* cx = device_ctx(db->device);
    std::lock_guard<std::mutex> lk(cx->mu);
    hipStream_t stream = nullptr;
    static const bool prof = getenv("S4G_PROFILE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    const uint32_t n = (uint32_t)sel.size();
    PairItem* items = nullptr;
    int rc = pair_stage(cx, n, &items);
    if (rc) return rc;
    constexpr uint32_t kBuckets = 2048;
    static thread_local std::vector<uint32_t> count;
    count.assign((size_t)kPairClasses * kBuckets + 1, 0);
    auto key_of = [&](uint32_t pi) {
        const uint32_t b = std::min<uint32_t>(db->len[pair_t[pi]] >> 5, kBuckets - 1);
        return (uint32_t)pair_class(q->len[pair_q[pi]]) * kBuckets + (kBuckets - 1 - b);
    };
    for (uint32_t i = 0; i < n; ++i) ++count[key_of(sel[i]) + 1];
    for (size_t k = 1; k < count.size(); ++k) count[k] += count[k - 1];
    uint32_t cls_begin[kPairClasses + 1];
    for (int c = 0; c <= kPairClasses; ++c) cls_begin[c] = count[(size_t)c * kBuckets];
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t pi = sel[i];
        const uint32_t at = count[key_of(pi)]++;
        items[at] = {pair_q[pi], pair_t[pi], at};
        h_scores[pi] = (int32_t)at;            // remember where this pair's score will sit
    }
    auto t1 = now();
    int32_t* d_sc = reinterpret_cast<int32_t*>(cx->d_items + cx->items_cap);
    rc = pair_submit(db, q, scorer, n, cls_begin, d_sc, stream);
    if (rc) return rc;
    auto t2 = now();
    static thread_local std::vector<int32_t> sc;
    sc.resize(n);
    HIP_TRY(hipMemcpyAsync(sc.data(), d_sc, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (uint32_t i = 0; i < n; ++i) { const uint32_t pi = sel[i]; h_scores[pi] = sc[(uint32_t)h_scores[pi]]; }
    if (prof) {
        auto t3 = now();
        auto us = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1e3; };
        fprintf(stderr, "[s4g profile] pairs n=%u: order %.1f us, upload+launch %.1f us, wait+download %.1f us\n", n, us(t0, t1), us(t1, t2), us(t2, t3));
    }
    return S4G_OK;
}

// s4g_scan's latency route: the whole [q_count][n_seqs] matrix as pairs (caller holds cx->mu).  Taken when the estimate
// says the inter-sequence launch would wait on its longest item for longer than a chip full of pair waves needs for
// everything; never for queries beyond kPairMaxRows or when a kernel was forced by flags.
int scan_as_pairs(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* scorer,
                  int32_t* d_scores, hipStream_t stream, uint32_t flags, bool* done) {
    *done = false;
    if (flags != 0) return S4G_OK;
    const uint64_t n_items = (uint64_t)q_count * db->n_seqs;
    if (n_items > (4u << 20)) return S4G_OK;
    uint32_t lq_max = 0, max_t = 0;
    uint64_t rows = 0;
    for (uint32_t i = 0; i < q_count; ++i) { lq_max = std::max(lq_max, q->len[q_begin + i]); rows += q->len[q_begin + i]; }
    if (lq_max > (uint32_t)kPairMaxRows) return S4G_OK;
    for (const Tile& t : db->tiles) max_t = std::max(max_t, t.max_len4 * 4);
    const double cells = (double)rows * (double)db->residues, longest = (double)lq_max * max_t;
    const double t_pairs = std::max(cells / kPairChipRate, longest / kPairWaveRate) + kPairCallOverhead;
    const double t_scan = std::max(cells / kScanChipRate, longest / kScanLaneRate);
    if (t_pairs >= t_scan) return S4G_OK;
    DeviceCtx* cx = device_ctx(db->device);
    int rc = ensure_seq_tables(db);
    if (rc) return rc;
    PairItem* items = nullptr;
    rc = pair_stage(cx, n_items, &items);
    if (rc) return rc;
    // class-major; inside a class the targets in length-sorted order (longest first), queries innermost
    uint32_t cls_begin[kPairClasses + 1];
    uint32_t at = 0;
    std::vector<uint32_t> qs;
    for (int c = 0; c < kPairClasses; ++c) {
        cls_begin[c] = at;
        qs.clear();
        for (uint32_t i = 0; i < q_count; ++i) if (pair_class(q->len[q_begin + i]) == c) qs.push_back(i);
        if (qs.empty()) continue;
        for (uint32_t pos = 0; pos < db->n_seqs; ++pos) {
            const uint32_t id = db->id_of_pos[pos];
            for (uint32_t i : qs) items[at++] = {q_begin + i, id, (uint32_t)((uint64_t)i * db->n_seqs + id)};
        }
    }
    cls_begin[kPairClasses] = at;
    rc = pair_submit(db, q, scorer, at, cls_begin, d_scores, stream);
    if (rc) return rc;
    *done = true;
    return S4G_OK;
}

}  // namespace

S4G_API int s4g_score_pairs_host(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint64_t n_pairs,
                                 const uint32_t* pair_q, const uint32_t* pair_t, int32_t* h_scores, uint32_t flags) {
    if (!db || !q || !scorer) return fail(S4G_ERR_INVALID, "NULL argument");
    if (n_pairs == 0) return S4G_OK;
    if (!pair_q || !pair_t || !h_scores) return fail(S4G_ERR_INVALID, "NULL argument");
    if (db->device != q->device) return fail(S4G_ERR_INVALID, "db on device %d, queries on device %d", db->device, q->device);
    if (n_pairs >= 0xFFFFFFFFull) return fail(S4G_ERR_INVALID, "too many pairs in one call");
    if ((flags & S4G_PAIRS_FORCE_SCAN) && (flags & S4G_PAIRS_FORCE_WAVE)) return fail(S4G_ERR_INVALID, "contradictory flags");
    for (uint64_t i = 0; i < n_pairs; ++i)
        if (pair_q[i] >= q->n || pair_t[i] >= db->n_seqs) return fail(S4G_ERR_INVALID, "pair %llu out of range", (unsigned long long)i);
    // Two exact routes.  "wave": one wave per pair (sw_pair_scan), for queries of up to kPairMaxRows residues.
    // "scan": score the pair's query against the whole resident database and pick the column.  Per query the cheaper
    // estimate wins: few candidates -> wave, candidate list ~ the whole database -> scan.
    std::vector<std::vector<uint32_t>> by_q(q->n);
    for (uint64_t i = 0; i < n_pairs; ++i) by_q[pair_q[i]].push_back((uint32_t)i);
    std::vector<uint32_t> wave_sel, scan_queries;
    uint32_t max_t = 0;
    for (const Tile& t : db->tiles) max_t = std::max(max_t, t.max_len4 * 4);
    for (uint32_t qi = 0; qi < q->n; ++qi) {
        if (by_q[qi].empty()) continue;
        const uint32_t lq = q->len[qi];
        if (lq == 0) { for (uint32_t i : by_q[qi]) h_scores[i] = 0; continue; }
        bool wave = lq <= (uint32_t)kPairMaxRows;
        if (wave && !(flags & S4G_PAIRS_FORCE_WAVE)) {
            double cells = 0, longest = 0;
            for (uint32_t i : by_q[qi]) { const double c = (double)lq * db->len[pair_t[i]]; cells += c; longest = std::max(longest, c); }
            const double t_wave = std::max(cells / kPairChipRate, longest / kPairWaveRate) + kPairCallOverhead;
            const double full = (double)lq * (double)db->residues;
            const double t_scan = std::max(full / kScanChipRate, (double)lq * max_t / kScanLaneRate) + (double)db->n_seqs * 4.0 / 20e9;
            wave = t_wave <= t_scan;
        }
        if (flags & S4G_PAIRS_FORCE_SCAN) wave = false;
        if (!wave && (flags & S4G_PAIRS_FORCE_WAVE))
            return fail(S4G_ERR_INVALID, "query %u has %u residues; the pair kernel takes at most %d", qi, lq, kPairMaxRows);
        if (wave) {
            wave_sel.insert(wave_sel.end(), by_q[qi].begin(), by_q[qi].end());
        } else scan_queries.push_back(qi);
    }
    if (!wave_sel.empty()) {
        int rc = score_pairs_device(db, q, scorer, wave_sel, pair_q, pair_t, h_scores);
        if (rc) return rc;
    }
    if (!scan_queries.empty()) {
        // consecutive runs of scan-route queries share one scan call, bounded to ~1 GiB of host scores
        const uint32_t batch = std::max<uint32_t>(1, (uint32_t)((1ull << 28) / std::max<uint32_t>(1, db->n_seqs)));
        std::vector<int32_t> row;
        for (size_t s0 = 0; s0 < scan_queries.size();) {
            size_t s1 = s0 + 1;
            while (s1 < scan_queries.size() && scan_queries[s1] == scan_queries[s1 - 1] + 1 && s1 - s0 < batch) ++s1;
            const uint32_t cnt = (uint32_t)(s1 - s0);
            row.resize((size_t)cnt * db->n_seqs);
            int rc = s4g_scan_host(db, q, scan_queries[s0], cnt, scorer, row.data());
            if (rc) return rc;
            for (uint32_t k = 0; k < cnt; ++k)
                for (uint32_t i : by_q[scan_queries[s0 + k]]) h_scores[i] = row[(size_t)k * db->n_seqs + pair_t[i]];
            s0 = s1;
        }
    }
    return S4G_OK;
}

S4G_API int s4g_score_lists_host(const s4g_db* db, const s4g_queries* q, const uint64_t* cand_offsets,
                                 const uint32_t* cand_ids, const s4g_scorer* scorer, int32_t* h_scores) {
    if (!db || !q || !scorer || !cand_offsets) return fail(S4G_ERR_INVALID, "NULL argument");
    if (q->n == 0 || cand_offsets[q->n] == 0) return S4G_OK;
    if (!cand_ids || !h_scores) return fail(S4G_ERR_INVALID, "NULL argument");
    const uint64_t n = cand_offsets[q->n];
    for (uint64_t i = 0; i < n; ++i)
        if (cand_ids[i] >= db->n_seqs) return fail(S4G_ERR_INVALID, "candidate id %u out of range", cand_ids[i]);
    std::vector<uint32_t> pq(n);
    for (uint32_t i = 0; i < q->n; ++i) {
        if (cand_offsets[i + 1] < cand_offsets[i] || cand_offsets[i + 1] > n) return fail(S4G_ERR_INVALID, "bad candidate offsets at %u", i);
        for (uint64_t k = cand_offsets[i]; k < cand_offsets[i + 1]; ++k) pq[k] = i;
    }
    return s4g_score_pairs_host(db, q, scorer, n, pq.data(), cand_ids, h_scores, 0);
}

// ---------------------------------------------------------------------------
// top-k (steps 2-3 of alignDatabase on the device): one 256-thread block per query row.
//   1. two-level radix select on the int32 score (high 12 bits of a 24-bit score, then low 12 bits) finds the cut
//      score exactly: the k-th largest score >= min_score;
//   2. everything above the cut is kept; among the ties AT the cut the smallest target ids are kept (ordered
//      block scan), so the result is the exact, deterministic top-k under (score desc, id asc);
//   3. bitonic sort of the <= k survivors in LDS on the 64-bit key (~score << 32 | id).
namespace {

constexpr int kTopkBlock = 256;
constexpr int kBins = 4096;
constexpr int kTopkMax = 4096;     // LDS: 4096 keys * 8 B = 32 KiB
constexpr int kScoreCap = (1 << 24) - 1;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// finds, scanning bins from the top, the bin where the running count reaches `want`; returns bin (or -1) and the
// count strictly above it.  Single-threaded over 4096 LDS words: ~4k cycles per row, negligible next to the scan.
__device__ void find_bin(const unsigned* hist, unsigned want, int* bin, unsigned* above) {
    unsigned acc = 0;
    int b = kBins - 1;
    for (; b >= 0; --b) {
        if (acc + hist[b] >= want) break;
        acc += hist[b];
    }
    *bin = b;
    *above = acc;
}

__global__ __launch_bounds__(kTopkBlock) void topk_rows(const int32_t* __restrict__ scores, uint32_t n, uint32_t k, uint32_t kpad,
                                                        int32_t min_score, uint32_t* __restrict__ out_ids,
                                                        int32_t* __restrict__ out_scores, uint32_t* __restrict__ out_counts,
                                                        const uint32_t* __restrict__ col_ids,
                                                        const int32_t* __restrict__ row_min_scores) {
    __shared__ unsigned hist[kBins];
    __shared__ unsigned long long keys[kTopkMax];
    __shared__ int s_hi, s_cut;
    __shared__ unsigned s_above, s_quota, s_nout, s_tie_base, s_wave_cnt[kTopkBlock / 64];
    const int tid = threadIdx.x;
    const int32_t* row = scores + (uint64_t)blockIdx.x * n;
    if (row_min_scores) min_score = row_min_scores[blockIdx.x];       // per-query cut (e-value driven)
    if (min_score < 0) min_score = 0;

    // ---- level 1: high 12 bits ----
    for (int i = tid; i < kBins; i += kTopkBlock) hist[i] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kTopkBlock) {
        const int v = row[i];
        if (v >= min_score) atomicAdd(&hist[clampi(v, 0, kScoreCap) >> 12], 1u);
    }
    __syncthreads();
    if (tid == 0) { int b; unsigned a; find_bin(hist, k, &b, &a); s_hi = b; s_above = a; }
    __syncthreads();
    const int hi = s_hi;
    if (hi < 0) {
        // fewer than k entries pass min_score: keep them all
        if (tid == 0) { s_cut = min_score; s_quota = 0xFFFFFFFFu; }
    } else {
        // ---- level 2: low 12 bits inside the high bin ----
        const unsigned above_hi = s_above;
        __syncthreads();
        for (int i = tid; i < kBins; i += kTopkBlock) hist[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += kTopkBlock) {
            const int v = row[i];
            if (v >= min_score && (clampi(v, 0, kScoreCap) >> 12) == hi) atomicAdd(&hist[clampi(v, 0, kScoreCap) & (kBins - 1)], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int b; unsigned a;
            find_bin(hist, k - above_hi, &b, &a);
            s_cut = (hi << 12) | b;                       // exact cut score (scores are < 2^24)
            s_quota = k - (above_hi + a);                 // ties at the cut to keep (smallest ids)
        }
    }
    if (tid == 0) { s_nout = 0; s_tie_base = 0; }
    __syncthreads();
    const int cut = s_cut;
    const unsigned quota = s_quota;

    // ---- compaction: > cut unordered, == cut in id order up to quota ----
    for (uint32_t base = 0; base < n; base += kTopkBlock) {
        const uint32_t i = base + tid;
        const int v = i < n ? row[i] : -1;
        const bool ok = i < n && v >= min_score;
        const bool above = ok && (v > cut || quota == 0xFFFFFFFFu);
        const bool tie = ok && !above && v == cut;
        if (above) keys[atomicAdd(&s_nout, 1u)] = ((unsigned long long)(unsigned)(0x7FFFFFFF - v) << 32) | i;
        const unsigned long long tb = __ballot(tie);
        if ((tid & 63) == 0) s_wave_cnt[tid >> 6] = __popcll(tb);
        __syncthreads();
        unsigned rank = s_tie_base + __popcll(tb & ((1ull << (tid & 63)) - 1ull));
        for (int w = 0; w < (tid >> 6); ++w) rank += s_wave_cnt[w];
        if (tie && rank < quota) keys[atomicAdd(&s_nout, 1u)] = ((unsigned long long)(unsigned)(0x7FFFFFFF - v) << 32) | i;
        __syncthreads();
        if (tid == 0) { unsigned t = 0; for (int w = 0; w < kTopkBlock / 64; ++w) t += s_wave_cnt[w]; s_tie_base += t; }
        __syncthreads();
    }
    const unsigned m = s_nout;                             // <= k by construction
    for (unsigned i = m + tid; i < kpad; i += kTopkBlock) keys[i] = ~0ull;
    __syncthreads();
    // ---- bitonic sort ascending on the key = score descending, id ascending ----
    for (unsigned size = 2; size <= kpad; size <<= 1) {
        for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
            for (unsigned t = tid; t < kpad / 2; t += kTopkBlock) {
                const unsigned lo = 2 * t - (t & (stride - 1));
                const unsigned hi2 = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi2];
                if ((a > b) == up) { keys[lo] = b; keys[hi2] = a; }
            }
            __syncthreads();
        }
    }
    for (unsigned i = tid; i < k; i += kTopkBlock) {
        const unsigned long long key = keys[i];
        const bool valid = i < m;
        out_ids[(uint64_t)blockIdx.x * k + i] = valid ? (col_ids ? col_ids[(uint32_t)key] : (uint32_t)key) : 0xFFFFFFFFu;
        out_scores[(uint64_t)blockIdx.x * k + i] = valid ? 0x7FFFFFFF - (int)(key >> 32) : 0;
    }
    if (tid == 0) out_counts[blockIdx.x] = m;
}

// running[q] <- best k of (running[q] U cand[q]) under (score desc, id asc); both inputs hold global ids.
__global__ __launch_bounds__(kTopkBlock) void merge_rows(uint32_t* __restrict__ run_ids, int32_t* __restrict__ run_sc,
                                                         uint32_t* __restrict__ run_cnt, const uint32_t* __restrict__ cand_ids,
                                                         const int32_t* __restrict__ cand_sc, const uint32_t* __restrict__ cand_cnt,
                                                         uint32_t k, uint32_t kpad2) {
    __shared__ unsigned long long keys[2 * kTopkMax];
    const int tid = threadIdx.x;
    const uint64_t row = (uint64_t)blockIdx.x * k;
    const unsigned na = run_cnt[blockIdx.x], nb = cand_cnt[blockIdx.x];
    for (unsigned i = tid; i < kpad2; i += kTopkBlock) {
        unsigned long long key = ~0ull;
        if (i < na) key = ((unsigned long long)(unsigned)(0x7FFFFFFF - run_sc[row + i]) << 32) | run_ids[row + i];
        else if (i - na < nb) key = ((unsigned long long)(unsigned)(0x7FFFFFFF - cand_sc[row + i - na]) << 32) | cand_ids[row + i - na];
        keys[i] = key;
    }
    __syncthreads();
    for (unsigned size = 2; size <= kpad2; size <<= 1) {
        for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
            for (unsigned t = tid; t < kpad2 / 2; t += kTopkBlock) {
                const unsigned lo = 2 * t - (t & (stride - 1));
                const unsigned hi2 = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi2];
                if ((a > b) == up) { keys[lo] = b; keys[hi2] = a; }
            }
            __syncthreads();
        }
    }
    const unsigned m = na + nb < k ? na + nb : k;
    for (unsigned i = tid; i < k; i += kTopkBlock) {
        const unsigned long long key = keys[i];
        run_ids[row + i] = i < m ? (uint32_t)key : 0xFFFFFFFFu;
        run_sc[row + i] = i < m ? 0x7FFFFFFF - (int)(key >> 32) : 0;
    }
    __syncthreads();
    if (tid == 0) run_cnt[blockIdx.x] = m;
}

}  // namespace

S4G_API int s4g_topk(const s4g_db* db, const int32_t* d_scores, uint32_t q_count, uint32_t k, int32_t min_score,
                     uint32_t* d_ids, int32_t* d_top_scores, uint32_t* d_counts, void* stream_) {
    if (!db || !d_scores || !d_ids || !d_top_scores || !d_counts) return fail(S4G_ERR_INVALID, "NULL argument");
    if (k == 0 || k > (uint32_t)kTopkMax) return fail(S4G_ERR_INVALID, "k must be in 1..%d (got %u)", kTopkMax, k);
    if (q_count == 0) return S4G_OK;
    DeviceGuard guard(db->device);
    uint32_t kpad = 2;
    while (kpad < k) kpad <<= 1;
    hipLaunchKernelGGL(topk_rows, dim3(q_count), dim3(kTopkBlock), 0, (hipStream_t)stream_, d_scores, db->n_seqs, k, kpad,
                       min_score, d_ids, d_top_scores, d_counts, (const uint32_t*)nullptr, (const int32_t*)nullptr);
    HIP_TRY(hipGetLastError());
    return S4G_OK;
}

// Streaming scan + running top-k: the score matrix is never materialised.  For each batch of queries and each
// database tile: scan into a [batch][tile columns] buffer, exact tile top-k (ids translated to original ids), merge
// into the running per-query lists.  Tile columns are ordered by original id, so ties break by id exactly as in
// s4g_topk on the full matrix.  This is the path for databases whose Q x N scores do not fit (BASELINE configs[3]).
static int scan_topk_impl(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* scorer,
                          uint32_t k, int32_t min_score, const int32_t* d_min_scores, uint32_t* d_ids, int32_t* d_top_scores,
                          uint32_t* d_counts, void* stream_, uint32_t flags);

S4G_API int s4g_scan_topk(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* scorer,
                          uint32_t k, int32_t min_score, uint32_t* d_ids, int32_t* d_top_scores, uint32_t* d_counts,
                          void* stream_, uint32_t flags) {
    return scan_topk_impl(db, q, q_begin, q_count, scorer, k, min_score, nullptr, d_ids, d_top_scores, d_counts, stream_, flags);
}

// Same with one score cut per query (h_min_scores[q_count], host memory) and host outputs: the form the e-value driven
// selection uses (a conservative integer score cut per query on the device, exact e-values on the survivors on the host).
S4G_API int s4g_scan_topk_host(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* scorer,
                               uint32_t k, const int32_t* h_min_scores, uint32_t* h_ids, int32_t* h_top_scores, uint32_t* h_counts,
                               uint32_t flags) {
    if (!db || !q || !scorer || !h_ids || !h_top_scores || !h_counts) return fail(S4G_ERR_INVALID, "NULL argument");
    if (q_count == 0) return S4G_OK;
    DeviceGuard guard(db->device);
    DevBuf<uint32_t> d_ids, d_cnt;
    DevBuf<int32_t> d_sc, d_min;
    HIP_TRY(d_ids.alloc((size_t)q_count * k)); HIP_TRY(d_sc.alloc((size_t)q_count * k)); HIP_TRY(d_cnt.alloc(q_count));
    if (h_min_scores) {
        HIP_TRY(d_min.alloc(q_count));
        HIP_TRY(hipMemcpy(d_min.p, h_min_scores, (size_t)q_count * 4, hipMemcpyHostToDevice));
    }
    int rc = scan_topk_impl(db, q, q_begin, q_count, scorer, k, 1, h_min_scores ? d_min.p : nullptr, d_ids.p, d_sc.p, d_cnt.p, nullptr, flags);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(h_counts, d_cnt.p, (size_t)q_count * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_ids, d_ids.p, (size_t)q_count * k * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_top_scores, d_sc.p, (size_t)q_count * k * 4, hipMemcpyDeviceToHost));
    return S4G_OK;
}

static int scan_topk_impl(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* scorer,
                          uint32_t k, int32_t min_score, const int32_t* d_min_scores, uint32_t* d_ids, int32_t* d_top_scores,
                          uint32_t* d_counts, void* stream_, uint32_t flags) {
    int rc = check_scan_args(db, q, scorer, q_begin, q_count);
    if (rc) return rc;
    if (k == 0 || k > (uint32_t)kTopkMax) return fail(S4G_ERR_INVALID, "k must be in 1..%d (got %u)", kTopkMax, k);
    if (q_count == 0) return S4G_OK;
    if (!d_ids || !d_top_scores || !d_counts) return fail(S4G_ERR_INVALID, "NULL argument");
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(db->device);
    DeviceCtx* cx = device_ctx(db->device);
    std::lock_guard<std::mutex> lk(cx->mu);
    rc = scratch_acquire(db, stream);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(d_ids, 0xFF, (size_t)q_count * k * 4, stream));
    HIP_TRY(hipMemsetAsync(d_top_scores, 0, (size_t)q_count * k * 4, stream));
    HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)q_count * 4, stream));
    if (db->n_seqs == 0) return scratch_release(db, stream);
    ensure_bounds(q, scorer);
    const uint64_t cols = db->max_tile_seqs;
    const uint32_t batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(q_count, kTileScoreBudget / (cols * 4)));
    const uint64_t need = (uint64_t)batch * cols * 4 + (uint64_t)batch * k * 8 + (uint64_t)batch * 4;
    if (cx->tile_scores_bytes < need) {
        if (cx->d_tile_scores) HIP_TRY(hipFree(cx->d_tile_scores));
        cx->d_tile_scores = nullptr; cx->tile_scores_bytes = 0;
        HIP_TRY(hipMalloc((void**)&cx->d_tile_scores, need));
        cx->tile_scores_bytes = need;
    }
    int32_t* d_buf = cx->d_tile_scores;
    uint32_t* c_ids = reinterpret_cast<uint32_t*>(d_buf + (uint64_t)batch * cols);
    int32_t* c_sc = reinterpret_cast<int32_t*>(c_ids + (uint64_t)batch * k);
    uint32_t* c_cnt = reinterpret_cast<uint32_t*>(c_sc + (uint64_t)batch * k);
    uint32_t kpad = 2, kpad2;
    while (kpad < k) kpad <<= 1;
    kpad2 = kpad * 2;
    for (uint32_t b = 0; b < q_count; b += batch) {
        const uint32_t cnt = std::min(batch, q_count - b);
        if (b) HIP_TRY(hipStreamSynchronize(stream));   // the next batch rewrites the query lists the previous launches read
        Plan p;
        rc = make_plan(db, q, q_begin + b, cnt, scorer, flags, &p);
        if (rc) return rc;
        rc = ensure_scratch(db, scorer, p, stream);
        if (rc) return rc;
        for (size_t ti = 0; ti < db->tiles.size(); ++ti) {
            const Tile& t = db->tiles[ti];
            rc = launch_tile(db, t, p.tiles[ti], q, scorer, p, q_begin + b, d_buf, t.n_seqs, t.d_perm_local, stream);
            if (rc) return rc;
            hipLaunchKernelGGL(topk_rows, dim3(cnt), dim3(kTopkBlock), 0, stream, d_buf, t.n_seqs, k, kpad, min_score, c_ids,
                               c_sc, c_cnt, (const uint32_t*)t.d_col_ids, d_min_scores ? d_min_scores + b : (const int32_t*)nullptr);
            hipLaunchKernelGGL(merge_rows, dim3(cnt), dim3(kTopkBlock), 0, stream, d_ids + (uint64_t)b * k,
                               d_top_scores + (uint64_t)b * k, d_counts + b, c_ids, c_sc, c_cnt, k, kpad2);
            HIP_TRY(hipGetLastError());
        }
    }
    return scratch_release(db, stream);
}

S4G_API int s4g_topk_host(const s4g_db* db, const int32_t* d_scores, uint32_t q_count, uint32_t k, int32_t min_score,
                          uint32_t* h_ids, int32_t* h_scores, uint32_t* h_counts, void* stream_) {
    if (!db || !d_scores || !h_ids || !h_scores || !h_counts) return fail(S4G_ERR_INVALID, "NULL argument");
    if (q_count == 0) return S4G_OK;
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(db->device);
    DevBuf<uint32_t> d_ids, d_counts;
    DevBuf<int32_t> d_sc;
    HIP_TRY(d_ids.alloc((size_t)q_count * k));
    HIP_TRY(d_sc.alloc((size_t)q_count * k));
    HIP_TRY(d_counts.alloc(q_count));
    int rc = s4g_topk(db, d_scores, q_count, k, min_score, d_ids.p, d_sc.p, d_counts.p, stream_);
    if (rc != S4G_OK) return rc;
    HIP_TRY(hipMemcpyAsync(h_counts, d_counts.p, (size_t)q_count * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_ids, d_ids.p, (size_t)q_count * k * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_scores, d_sc.p, (size_t)q_count * k * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return S4G_OK;
}

// ---------------------------------------------------------------------------
// Traceback of the survivors on the device (step 4 of alignDatabase).
namespace {

template <int RPL>
int launch_pair_trace(DeviceCtx* cx, PairTraceArgs a, uint32_t* grid_cache, hipStream_t stream) {
    if (*grid_cache == 0) {
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_pair_trace<RPL>, 64, 0));
        *grid_cache = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
    }
    const uint32_t grid = std::max(1u, std::min(*grid_cache, a.n_pairs));
    hipLaunchKernelGGL(sw_pair_trace<RPL>, dim3(grid), dim3(64), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return S4G_OK;
}

// Traceback with one wave per pair (sw_pair_trace): `sel` = indices of the pairs whose query has <= kPairMaxRows residues.
int traceback_pairs(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, const std::vector<uint32_t>& sel,
                    const uint32_t* pair_q, const uint32_t* pair_t, const int32_t* pair_score,
                    int32_t* h_out, const uint64_t* path_off, uint8_t* h_paths) {
    DeviceGuard guard(db->device);
    DeviceCtx* cx = device_ctx(db->device);
    std::lock_guard<std::mutex> lk(cx->mu);
    hipStream_t stream = nullptr;
    int rc = device_props(cx);
    if (rc) return rc;
    rc = ensure_seq_tables(db);
    if (rc) return rc;
    if (!cx->d_counter) HIP_TRY(hipMalloc((void**)&cx->d_counter, 256));
    if (!cx->d_mat) HIP_TRY(hipMalloc((void**)&cx->d_mat, 1024));
    if (cx->last_use) HIP_TRY(hipStreamWaitEvent(stream, cx->last_use, 0));
    HIP_TRY(hipMemcpyAsync(cx->d_mat, scorer->mat32, 1024, hipMemcpyHostToDevice, stream));
    // class-major, longest target first (32-residue granularity), as in score_pairs_device
    const uint32_t n = (uint32_t)sel.size();
    constexpr uint32_t kBuckets = 2048;
    std::vector<uint32_t> count((size_t)kPairClasses * kBuckets + 1, 0), order(n);
    auto key_of = [&](uint32_t pi) {
        const uint32_t b = std::min<uint32_t>(db->len[pair_t[pi]] >> 5, kBuckets - 1);
        return (uint32_t)pair_class(q->len[pair_q[pi]]) * kBuckets + (kBuckets - 1 - b);
    };
    for (uint32_t i = 0; i < n; ++i) ++count[key_of(sel[i]) + 1];
    for (size_t k = 1; k < count.size(); ++k) count[k] += count[k - 1];
    uint32_t cls_begin[kPairClasses + 1];
    for (int c = 0; c <= kPairClasses; ++c) cls_begin[c] = count[(size_t)c * kBuckets];
    for (uint32_t i = 0; i < n; ++i) order[count[key_of(sel[i])]++] = sel[i];
    auto dir_bytes = [&](uint32_t pi) {
        const int c = pair_class(q->len[pair_q[pi]]);
        const uint32_t rpl = (uint32_t)kPairRpl[c], nw = rpl <= 4 ? 1 : (rpl <= 8 ? 2 : 4);
        const uint64_t steps = (uint64_t)db->len[pair_t[pi]] + (q->len[pair_q[pi]] + rpl - 1) / rpl - 1;
        return ((steps + 1) & ~1ull) * 64 * nw * 4;
    };
    const uint64_t dir_budget = 8ull << 30;
    std::vector<PairTraceItem> items;
    std::vector<int32_t> out5, best;
    std::vector<uint8_t> paths;
    for (uint32_t b0 = 0; b0 < n;) {
        // a batch: consecutive pairs of ONE class whose direction blocks fit the budget
        int c = 0;
        while (cls_begin[c + 1] <= b0) ++c;
        uint32_t b1 = b0;
        uint64_t dir_total = 0, path_total = 0;
        items.clear();
        while (b1 < cls_begin[c + 1] && (b1 == b0 || dir_total + dir_bytes(order[b1]) <= dir_budget)) {
            const uint32_t pi = order[b1];
            const uint64_t cap = (uint64_t)q->len[pair_q[pi]] + db->len[pair_t[pi]];
            path_total += cap;
            items.push_back({pair_q[pi], pair_t[pi], pair_score[pi], b1 - b0, dir_total, path_total});
            dir_total += dir_bytes(pi);
            ++b1;
        }
        const uint32_t ni = b1 - b0;
        const bool big_dirs = dir_total > (256ull << 20);
        DevBuf<uint8_t> d_dirs_big;
        if (big_dirs) HIP_TRY(d_dirs_big.alloc(dir_total));
        uint64_t need = 0;
        auto reserve = [&](uint64_t bytes) { const uint64_t at = need; need = (need + bytes + 255) / 256 * 256; return at; };
        const uint64_t o_items = reserve((uint64_t)ni * sizeof(PairTraceItem)), o_out = reserve((uint64_t)ni * 20), o_best = reserve((uint64_t)ni * 4);
        const uint64_t o_paths = reserve(path_total), o_dirs = big_dirs ? 0 : reserve(dir_total);
        if (cx->arena_bytes < need) {
            if (cx->arena) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(cx->arena)); }
            cx->arena = nullptr; cx->arena_bytes = 0;
            const uint64_t want = std::max<uint64_t>(need + need / 4, 16ull << 20);
            HIP_TRY(hipMalloc((void**)&cx->arena, want));
            cx->arena_bytes = want;
        }
        char* A = cx->arena;
        HIP_TRY(hipMemcpyAsync(A + o_items, items.data(), (size_t)ni * sizeof(PairTraceItem), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(cx->d_counter + 32, 0, 4, stream));
        PairTraceArgs a;
        a.items = reinterpret_cast<const PairTraceItem*>(A + o_items); a.seq_addr = db->d_seq_addr; a.seq_len = db->d_seq_len;
        a.q_codes = q->d_codes; a.q_off = q->d_off; a.q_len = q->d_len; a.mat = cx->d_mat;
        a.dirs = big_dirs ? d_dirs_big.p : reinterpret_cast<uint8_t*>(A + o_dirs); a.paths = reinterpret_cast<uint8_t*>(A + o_paths);
        a.out5 = reinterpret_cast<int32_t*>(A + o_out); a.best = reinterpret_cast<int32_t*>(A + o_best);
        a.counter = cx->d_counter + 32; a.n_pairs = ni; a.open = (uint32_t)scorer->open; a.ext = (uint32_t)scorer->ext;
        switch (c) {
            case 0: rc = launch_pair_trace<2>(cx, a, &cx->grid_pair_trace[0], stream); break;
            case 1: rc = launch_pair_trace<4>(cx, a, &cx->grid_pair_trace[1], stream); break;
            case 2: rc = launch_pair_trace<6>(cx, a, &cx->grid_pair_trace[2], stream); break;
            case 3: rc = launch_pair_trace<8>(cx, a, &cx->grid_pair_trace[3], stream); break;
            case 4: rc = launch_pair_trace<12>(cx, a, &cx->grid_pair_trace[4], stream); break;
            default: rc = launch_pair_trace<16>(cx, a, &cx->grid_pair_trace[5], stream); break;
        }
        if (rc) return rc;
        out5.resize((size_t)ni * 5); best.resize(ni); paths.resize(path_total ? path_total : 1);
        HIP_TRY(hipMemcpyAsync(out5.data(), A + o_out, (size_t)ni * 20, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(best.data(), A + o_best, (size_t)ni * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(paths.data(), A + o_paths, path_total, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        for (uint32_t x = 0; x < ni; ++x) {
            const uint32_t pi = order[b0 + x];
            const int32_t want = pair_score[pi] > 0 && q->len[pair_q[pi]] && db->len[pair_t[pi]] ? pair_score[pi] : 0;
            if (best[x] != want && !(pair_score[pi] <= 0 && best[x] == 0))
                return fail(S4G_ERR_INVALID, "pair %u: score %d passed in, traceback kernel computed %d", pi, pair_score[pi], best[x]);
            memcpy(h_out + (size_t)pi * 5, out5.data() + (size_t)x * 5, 20);
            const uint32_t len = (uint32_t)out5[(size_t)x * 5 + 4];
            memcpy(h_paths + path_off[pi], paths.data() + items[x].path_end - len, len);
        }
        b0 = b1;
    }
    return S4G_OK;
}

int traceback_groups(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t n_pairs,
                     const uint32_t* pair_q, const uint32_t* pair_t, const int32_t* pair_score,
                     int32_t* h_out, const uint64_t* path_off, uint8_t* h_paths);

}  // namespace

S4G_API int s4g_traceback_host_flags(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t n_pairs,
                                     const uint32_t* pair_q, const uint32_t* pair_t, const int32_t* pair_score,
                                     int32_t* h_out, const uint64_t* path_off, uint8_t* h_paths, uint32_t flags) {
    if (!db || !q || !scorer) return fail(S4G_ERR_INVALID, "NULL argument");
    if (n_pairs == 0) return S4G_OK;
    if (!pair_q || !pair_t || !pair_score || !h_out || !path_off || !h_paths) return fail(S4G_ERR_INVALID, "NULL argument");
    if (db->device != q->device) return fail(S4G_ERR_INVALID, "db on device %d, queries on device %d", db->device, q->device);
    if ((flags & S4G_PAIRS_FORCE_SCAN) && (flags & S4G_PAIRS_FORCE_WAVE)) return fail(S4G_ERR_INVALID, "contradictory flags");
    for (uint32_t i = 0; i < n_pairs; ++i) {
        if (pair_q[i] >= q->n || pair_t[i] >= db->n_seqs) return fail(S4G_ERR_INVALID, "pair %u out of range", i);
        if (q->len[pair_q[i]] >= 65535 || db->len[pair_t[i]] >= 65535)
            return fail(S4G_ERR_INVALID, "pair %u: sequences of 65535 residues or more are not supported by the device traceback", i);
        if (path_off[i + 1] - path_off[i] < (uint64_t)q->len[pair_q[i]] + db->len[pair_t[i]])
            return fail(S4G_ERR_INVALID, "pair %u: path capacity must be at least query length + target length", i);
    }
    // one wave per pair (sw_pair_trace) for queries of up to kPairMaxRows residues; the 64-pairs-per-wave strip kernels
    // (sw_trace_fill + sw_trace_walk) for the rest
    std::vector<uint32_t> wave_sel, group_sel;
    for (uint32_t i = 0; i < n_pairs; ++i) {
        const bool wave = !(flags & S4G_PAIRS_FORCE_SCAN) && q->len[pair_q[i]] <= (uint32_t)kPairMaxRows;
        if (!wave && (flags & S4G_PAIRS_FORCE_WAVE))
            return fail(S4G_ERR_INVALID, "pair %u: query of %u residues; the pair kernel takes at most %d", i, q->len[pair_q[i]], kPairMaxRows);
        (wave ? wave_sel : group_sel).push_back(i);
    }
    if (!wave_sel.empty()) {
        int rc = traceback_pairs(db, q, scorer, wave_sel, pair_q, pair_t, pair_score, h_out, path_off, h_paths);
        if (rc) return rc;
    }
    if (group_sel.empty()) return S4G_OK;
    if (group_sel.size() == n_pairs) return traceback_groups(db, q, scorer, n_pairs, pair_q, pair_t, pair_score, h_out, path_off, h_paths);
    // mixed call: compact the remaining pairs, run them, scatter the results
    const uint32_t m = (uint32_t)group_sel.size();
    std::vector<uint32_t> gq(m), gt(m);
    std::vector<int32_t> gs(m), gout((size_t)m * 5);
    std::vector<uint64_t> goff(m + 1, 0);
    for (uint32_t k = 0; k < m; ++k) {
        const uint32_t i = group_sel[k];
        gq[k] = pair_q[i]; gt[k] = pair_t[i]; gs[k] = pair_score[i];
        goff[k + 1] = goff[k] + (path_off[i + 1] - path_off[i]);
    }
    std::vector<uint8_t> gpaths(goff[m] ? goff[m] : 1);
    int rc = traceback_groups(db, q, scorer, m, gq.data(), gt.data(), gs.data(), gout.data(), goff.data(), gpaths.data());
    if (rc) return rc;
    for (uint32_t k = 0; k < m; ++k) {
        const uint32_t i = group_sel[k];
        memcpy(h_out + (size_t)i * 5, gout.data() + (size_t)k * 5, 20);
        memcpy(h_paths + path_off[i], gpaths.data() + goff[k], (size_t)gout[(size_t)k * 5 + 4]);
    }
    return S4G_OK;
}

S4G_API int s4g_traceback_host(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t n_pairs,
                               const uint32_t* pair_q, const uint32_t* pair_t, const int32_t* pair_score,
                               int32_t* h_out, const uint64_t* path_off, uint8_t* h_paths) {
    return s4g_traceback_host_flags(db, q, scorer, n_pairs, pair_q, pair_t, pair_score, h_out, path_off, h_paths, 0);
}

namespace {
int traceback_groups(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t n_pairs,
                     const uint32_t* pair_q, const uint32_t* pair_t, const int32_t* pair_score,
                     int32_t* h_out, const uint64_t* path_off, uint8_t* h_paths) {
    DeviceGuard guard(db->device);
    DeviceCtx* cx = device_ctx(db->device);
    std::lock_guard<std::mutex> lk(cx->mu);
    hipStream_t stream = nullptr;

    // items: pairs grouped by query, sorted by target length inside a query, 64 lanes per item
    std::vector<uint32_t> idx(n_pairs);
    std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
        if (pair_q[a] != pair_q[b]) return pair_q[a] < pair_q[b];
        return db->len[pair_t[a]] > db->len[pair_t[b]];
    });
    struct Item { uint32_t q, len4, first, count; uint64_t dir_dwords; };
    std::vector<Item> items;
    for (uint32_t i = 0; i < n_pairs;) {
        uint32_t j = i;
        while (j < n_pairs && j - i < 64 && pair_q[idx[j]] == pair_q[idx[i]]) ++j;
        const uint32_t len4 = (db->len[pair_t[idx[i]]] + 3) / 4;          // longest first
        const uint32_t nstrips = (q->len[pair_q[idx[i]]] + kTraceR - 1) / kTraceR;
        items.push_back({pair_q[idx[i]], len4, i, j - i, (uint64_t)nstrips * len4 * kTraceR * 64});
        i = j;
    }
    {
        int rc = device_props(cx);
        if (rc) return rc;
    }
    if (!cx->d_counter) HIP_TRY(hipMalloc((void**)&cx->d_counter, 256));
    if (!cx->d_mat) HIP_TRY(hipMalloc((void**)&cx->d_mat, 1024));
    if (cx->last_use) HIP_TRY(hipStreamWaitEvent(stream, cx->last_use, 0));      // a scan on another stream may still read d_mat
    HIP_TRY(hipMemcpyAsync(cx->d_mat, scorer->mat32, 1024, hipMemcpyHostToDevice, stream));
    struct { int8_t* p; } d_mat = {cx->d_mat};
    struct { unsigned* p; } d_counter = {cx->d_counter + 16};

    const uint64_t dir_budget = (8ull << 30) / 4;       // dwords of direction bytes per batch
    std::vector<int32_t> out5;
    for (size_t b0 = 0; b0 < items.size();) {
        size_t b1 = b0;
        uint64_t dir_total = 0;
        uint32_t max_len4 = 0;
        while (b1 < items.size() && (b1 == b0 || dir_total + items[b1].dir_dwords <= dir_budget)) {
            dir_total += items[b1].dir_dwords;
            max_len4 = std::max(max_len4, items[b1].len4);
            ++b1;
        }
        const uint32_t ni = (uint32_t)(b1 - b0);
        std::vector<uint64_t> lane_res((size_t)ni * 64, 0), dir_off(ni);
        std::vector<uint32_t> lane_len((size_t)ni * 64, 0), item_q(ni), item_len4(ni), slot_of_pair;
        std::vector<int32_t> lane_score((size_t)ni * 64, -1);
        std::vector<uint32_t> pair_ids;                 // original pair index of each slot_of_pair entry
        std::vector<uint64_t> poff(1, 0);
        uint64_t off = 0;
        for (uint32_t k = 0; k < ni; ++k) {
            const Item& it = items[b0 + k];
            item_q[k] = it.q; item_len4[k] = it.len4; dir_off[k] = off; off += it.dir_dwords;
            for (uint32_t l = 0; l < it.count; ++l) {
                const uint32_t pi = idx[it.first + l];
                const uint32_t id = pair_t[pi];
                const uint32_t pos = db->pos_of_id[id];
                size_t ti = 0;
                while (ti + 1 < db->tiles.size() && db->tiles[ti + 1].pos0 <= pos) ++ti;
                const Tile& t = db->tiles[ti];
                const uint32_t lp = (uint32_t)(pos - t.pos0);
                lane_res[(size_t)k * 64 + l] = (uint64_t)(uintptr_t)(t.d_res + t.h_grp_off[lp >> 6] + (lp & 63));
                lane_len[(size_t)k * 64 + l] = db->len[id];
                lane_score[(size_t)k * 64 + l] = pair_score[pi];
                slot_of_pair.push_back(k * 64 + l);
                pair_ids.push_back(pi);
                poff.push_back(poff.back() + (path_off[pi + 1] - path_off[pi]));
            }
            for (uint32_t l = it.count; l < 64; ++l) lane_res[(size_t)k * 64 + l] = lane_res[(size_t)k * 64];   // valid address, length 0
        }
        const uint32_t np = (uint32_t)slot_of_pair.size();
        const uint32_t grid = std::max(1u, std::min<uint32_t>((ni + kWavesPerBlock - 1) / kWavesPerBlock, (uint32_t)cx->prop.multiProcessorCount * 2));
        const uint64_t ws_slot = 2ull * std::max(1u, max_len4) * 64 * 4;
        // every temporary of this batch is carved from the device's grow-only arena (no hipMalloc/hipFree per call);
        // only direction blocks beyond 256 MB get an allocation of their own
        const bool big_dirs = off * 4 > (256ull << 20);
        DevBuf<uint32_t> d_dirs_big;
        if (big_dirs) HIP_TRY(d_dirs_big.alloc(off));
        uint64_t need = 0;
        auto reserve = [&](uint64_t bytes) { const uint64_t at = need; need = (need + bytes + 255) / 256 * 256; return at; };
        const uint64_t o_lane_res = reserve((uint64_t)ni * 64 * 8), o_dir_off = reserve((uint64_t)ni * 8), o_poff = reserve((uint64_t)(np + 1) * 8);
        const uint64_t o_lane_len = reserve((uint64_t)ni * 64 * 4), o_item_q = reserve((uint64_t)ni * 4), o_item_len4 = reserve((uint64_t)ni * 4);
        const uint64_t o_slot = reserve((uint64_t)np * 4), o_lane_score = reserve((uint64_t)ni * 64 * 4);
        const uint64_t o_end = reserve((uint64_t)ni * 64 * 4), o_best = reserve((uint64_t)ni * 64 * 4), o_out = reserve((uint64_t)np * 5 * 4);
        const uint64_t o_paths = reserve(poff.back()), o_ws = reserve(ws_slot * 4 * grid * kWavesPerBlock);
        const uint64_t o_dirs = big_dirs ? 0 : reserve(off * 4);
        if (cx->arena_bytes < need) {
            if (cx->arena) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(cx->arena)); }
            cx->arena = nullptr; cx->arena_bytes = 0;
            const uint64_t want = std::max<uint64_t>(need + need / 4, 16ull << 20);
            HIP_TRY(hipMalloc((void**)&cx->arena, want));
            cx->arena_bytes = want;
        }
        char* A = cx->arena;
        auto put = [&](uint64_t at, const void* src, uint64_t bytes) { return bytes ? hipMemcpyAsync(A + at, src, bytes, hipMemcpyHostToDevice, stream) : hipSuccess; };
        HIP_TRY(put(o_lane_res, lane_res.data(), lane_res.size() * 8)); HIP_TRY(put(o_dir_off, dir_off.data(), dir_off.size() * 8));
        HIP_TRY(put(o_poff, poff.data(), poff.size() * 8)); HIP_TRY(put(o_lane_len, lane_len.data(), lane_len.size() * 4));
        HIP_TRY(put(o_item_q, item_q.data(), item_q.size() * 4)); HIP_TRY(put(o_item_len4, item_len4.data(), item_len4.size() * 4));
        HIP_TRY(put(o_slot, slot_of_pair.data(), slot_of_pair.size() * 4)); HIP_TRY(put(o_lane_score, lane_score.data(), lane_score.size() * 4));
        uint32_t* d_dirs = big_dirs ? d_dirs_big.p : reinterpret_cast<uint32_t*>(A + o_dirs);
        HIP_TRY(hipMemsetAsync(d_counter.p, 0, 4, stream));
        TraceArgs a;
        a.lane_res = reinterpret_cast<const uint64_t*>(A + o_lane_res); a.lane_len = reinterpret_cast<const uint32_t*>(A + o_lane_len);
        a.lane_score = reinterpret_cast<const int32_t*>(A + o_lane_score); a.item_q = reinterpret_cast<const uint32_t*>(A + o_item_q);
        a.item_len4 = reinterpret_cast<const uint32_t*>(A + o_item_len4); a.item_dir_off = reinterpret_cast<const uint64_t*>(A + o_dir_off);
        a.q_codes = q->d_codes; a.q_off = q->d_off; a.q_len = q->d_len; a.mat = d_mat.p;
        a.dirs = d_dirs; a.ws = reinterpret_cast<int32_t*>(A + o_ws); a.ws_slot = ws_slot; a.counter = d_counter.p; a.n_items = ni;
        a.open = (uint32_t)scorer->open; a.ext = (uint32_t)scorer->ext;
        a.end_key = reinterpret_cast<uint32_t*>(A + o_end); a.best = reinterpret_cast<int32_t*>(A + o_best);
        hipLaunchKernelGGL(sw_trace_fill, dim3(grid), dim3(kBlock), 0, stream, a);
        HIP_TRY(hipGetLastError());
        WalkArgs w;
        w.dirs = d_dirs; w.item_dir_off = a.item_dir_off; w.item_len4 = a.item_len4; w.end_key = a.end_key;
        w.slot_of_pair = reinterpret_cast<const uint32_t*>(A + o_slot); w.path_off = reinterpret_cast<const uint64_t*>(A + o_poff);
        w.paths = reinterpret_cast<uint8_t*>(A + o_paths); w.out = reinterpret_cast<int32_t*>(A + o_out); w.n_pairs = np;
        hipLaunchKernelGGL(sw_trace_walk, dim3((np + 63) / 64), dim3(64), 0, stream, w);
        HIP_TRY(hipGetLastError());
        out5.resize((size_t)np * 5);
        std::vector<uint8_t> paths(poff.back() ? poff.back() : 1);
        std::vector<int32_t> best((size_t)ni * 64);
        HIP_TRY(hipMemcpy(out5.data(), A + o_out, out5.size() * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(paths.data(), A + o_paths, poff.back(), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(best.data(), A + o_best, best.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t x = 0; x < np; ++x) {
            const uint32_t pi = pair_ids[x];
            if (best[slot_of_pair[x]] != pair_score[pi])
                return fail(S4G_ERR_INVALID, "pair %u: score %d passed in, traceback kernel computed %d", pi, pair_score[pi], best[slot_of_pair[x]]);
            memcpy(h_out + (size_t)pi * 5, out5.data() + (size_t)x * 5, 20);
            memcpy(h_paths + path_off[pi], paths.data() + poff[x], (size_t)out5[(size_t)x * 5 + 4]);
        }
        b0 = b1;
    }
    return S4G_OK;
}
}  // namespace

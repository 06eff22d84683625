// s4g_core.hip -- C ABI (include/s4g_hip.h): errors, devices, scorer, database (packing, residency in HBM, packed file
// format), queries.  Mirrors what swsharp's scorerCreateMatrix / chainDatabaseCreate do for sift4g
// (/root/reference/sift4g/src/main.cpp:208, database_alignment.cpp:80).
#include "s4g_internal.hpp"
#include "sw_pack.hip.h"
#include "matrices.h"

// S4G_PROFILE=1: cumulative host time of the database-creation phases, printed at exit (diagnostic only)
struct CreateClock {
    double t[4] = {0, 0, 0, 0};
    double ranks = 0;          // per-tile id ranks (bitmap + prefix counts)
    long n = 0;
    bool on = getenv("S4G_PROFILE") != nullptr;
    ~CreateClock() {
        if (on && n) fprintf(stderr, "[s4g profile] s4g_db_create x%ld: lengths+sort %.3f s, host pack %.3f s, tile upload (+ device pack) %.3f s, raw codes upload %.3f s, tile id ranks %.3f s\n", n, t[0], t[1], t[2], t[3], ranks);
    }
} g_create_clock;
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }


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

int device_count_raw() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); return -1; }
    return n;
}

static DeviceCtx g_ctx[64];
DeviceCtx* device_ctx(int device) {
    DeviceCtx* c = &g_ctx[device & 63];
    c->device = device;
    return c;
}

// ---------------------------------------------------------------------------
// Persistent host workers for parallel_for (s4g_internal.hpp): up to 15 threads (one less than the host's cores) started on first use, woken per job, joined when the
// library is unloaded.  One job at a time (callers hold no lock they could deadlock on: jobs are pure host loops).
namespace {
struct HostPool {
    static constexpr unsigned kMaxWorkers = 15;
    unsigned n_workers = 0;
    std::mutex mu, job_mu;
    std::condition_variable wake, done_cv;
    std::thread th[kMaxWorkers];
    bool started = false, quit = false;
    uint64_t generation = 0;
    unsigned parts = 0, remaining = 0;
    void (*fn)(void*, unsigned) = nullptr;
    void* ctx = nullptr;

    void worker(unsigned w) {
        uint64_t seen = 0;
        for (;;) {
            void (*f)(void*, unsigned); void* c; unsigned p;
            {
                std::unique_lock<std::mutex> lk(mu);
                wake.wait(lk, [&] { return quit || generation != seen; });
                if (quit) return;
                seen = generation;
                f = fn; c = ctx; p = parts;
            }
            if (w + 1 < p) f(c, w + 1);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--remaining == 0) done_cv.notify_one();
            }
        }
    }
    void run(unsigned n_parts, void (*f)(void*, unsigned), void* c) {
        std::lock_guard<std::mutex> job(job_mu);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!started) {
                n_workers = std::max(1u, std::min(kMaxWorkers, std::max(2u, std::thread::hardware_concurrency()) - 1));
                for (unsigned w = 0; w < n_workers; ++w) th[w] = std::thread(&HostPool::worker, this, w);
                started = true;
            }
            fn = f; ctx = c; parts = n_parts; remaining = n_workers;
            ++generation;
        }
        wake.notify_all();
        f(c, 0);
        for (unsigned p = n_workers + 1; p < n_parts; ++p) f(c, p);          // (more parts than threads: the caller takes the rest)
        std::unique_lock<std::mutex> lk(mu);
        done_cv.wait(lk, [&] { return remaining == 0; });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        wake.notify_all();
        if (started) for (auto& t : th) if (t.joinable()) t.join();
    }
} g_host_pool;
}  // namespace

void host_pool_run(unsigned parts, void (*fn)(void* ctx, unsigned part), void* ctx) {
    g_host_pool.run(parts, fn, ctx);
}

int warm_core_kernels(DeviceCtx* cx) {
    (void)cx;
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_pack_tile, 256, 0));
    return S4G_OK;
}

// ---------------------------------------------------------------------------
S4G_API const char* s4g_last_error(void) { return g_err.c_str(); }
S4G_API int s4g_abi_version(void) { return 6; }      // 2: + s4g_scan_stats, s4g_clock_probe; 3: + s4g_prefilter_*; 4: + s4g_scan_clock, s4g_db_create_cached, s4g_chain_cache_info, s4g_db_open_streamed, s4g_db_stream_stats; 5: + s4g_comm_*, s4g_gather_topk, s4g_merge_topk_host; 6: + s4g_traceback_submit, s4g_traceback_finish

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
    if (!matrix_by_name(name, table26x26)) return fail(S4G_ERR_INVALID, "unknown substitution matrix '%s' (available: %s)", name, kMatrixNames);
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

constexpr uint32_t kTileMaxSeqs = 1u << 20;            // sequences per tile
constexpr uint64_t kTileMaxResidues = 1ull << 30;      // padded residues per tile

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

// Packs order[p0, p1) (length-sorted, longest first) into one tile and uploads it.
// d_raw != nullptr: the raw codes are already on the device and the transposition runs there (sw_pack_tile); d_bad
// collects its validity flag (checked by the caller after the last tile).
hipError_t build_tile(s4g_db* db, Tile& t, const std::vector<uint32_t>& order, size_t p0, size_t p1,
                      const uint8_t* codes, const uint64_t* offsets, std::string* err, uint64_t extra_bytes = 0,
                      uint64_t* extra_off = nullptr, const uint8_t* d_raw = nullptr, int* d_bad = nullptr,
                      const uint8_t* const* seq_ptrs = nullptr) {
    const double t_enter = now_s();
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
    std::vector<uint32_t> col_ids(n);
    const bool whole = (n == db->n_seqs);           // single tile: the ids are 0..n-1 and the rank of an id is the id
    // several tiles: membership bitmap over all ids + popcount prefix per word, so that sorting the tile's ids and ranking
    // one of them are O(n_seqs / 64 + n) instead of a sort and a binary search per sequence (20 M sequences: 2.5 s -> 0.1 s)
    std::vector<uint64_t> member;
    std::vector<uint32_t> before;
    if (whole) std::iota(col_ids.begin(), col_ids.end(), 0u);
    else {
        const size_t words = ((size_t)db->n_seqs + 63) / 64;
        member.assign(words, 0);
        before.resize(words);
        for (size_t pp = p0; pp < p1; ++pp) member[order[pp] >> 6] |= 1ull << (order[pp] & 63);
        uint32_t seen = 0;
        for (size_t w = 0; w < words; ++w) {
            before[w] = seen;
            for (uint64_t m = member[w]; m; m &= m - 1) col_ids[seen++] = (uint32_t)(w * 64 + __builtin_ctzll(m));
        }
    }
    auto rank_of = [&](uint32_t id) -> uint32_t {
        return whole ? id : before[id >> 6] + (uint32_t)__builtin_popcountll(member[id >> 6] & ((1ull << (id & 63)) - 1));
    };
    std::vector<uint32_t> perm_g((size_t)G * 64, 0xFFFFFFFFu), perm_l((size_t)G * 64, 0xFFFFFFFFu);
    g_create_clock.ranks += now_s() - t_enter;
    if (d_raw) {
        const double t_pack0 = now_s();
        std::vector<uint64_t> src_off((size_t)G * 64, 0);
        std::vector<uint32_t> src_len((size_t)G * 64, 0);
        // random gathers over the whole id range: split over host threads when the tile is large
        const unsigned nth = n >= (1u << 18) ? std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1;
        std::vector<uint64_t> part_res(nth, 0);
        auto gather = [&](unsigned w) {
            uint64_t r = 0;
            for (uint32_t k = (uint32_t)((uint64_t)n * w / nth), k1 = (uint32_t)((uint64_t)n * (w + 1) / nth); k < k1; ++k) {
                const uint32_t id = order[p0 + k];
                perm_g[k] = id;
                perm_l[k] = rank_of(id);
                src_off[k] = offsets[id];
                src_len[k] = db->len[id];
                r += db->len[id];
            }
            part_res[w] = r;
        };
        if (nth == 1) gather(0);
        else {
            std::vector<std::thread> th;
            for (unsigned w = 0; w < nth; ++w) th.emplace_back(gather, w);
            for (auto& x : th) x.join();
        }
        for (uint64_t r : part_res) t.residues += r;
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
    // small tile: every array goes into ONE host blob laid out like the device block, so the tile is one allocation
    // (taken from the per-device cache when one fits) and one upload instead of six of each
    uint64_t need = 0;
    auto reserve = [&](uint64_t bytes) { const uint64_t at = need; need = (need + bytes + 255) / 256 * 256; return at; };
    const uint64_t o_res = reserve(dwords * 4), o_off = reserve((uint64_t)G * 8), o_len4 = reserve((uint64_t)G * 4);
    const uint64_t o_pg = reserve((uint64_t)G * 256), o_pl = reserve((uint64_t)G * 256), o_col = reserve((uint64_t)n * 4);
    const uint64_t o_extra = reserve(extra_bytes);
    const bool use_block = need <= kBlockMax;
    std::vector<char> blob(use_block ? need : 0);
    std::vector<uint32_t> res_vec(use_block ? 0 : (dwords ? dwords : 1), 0x1F1F1F1Fu);
    uint32_t* const res = use_block ? reinterpret_cast<uint32_t*>(blob.data() + o_res) : res_vec.data();
    if (use_block) std::fill(res, res + dwords, 0x1F1F1F1Fu);
    std::atomic<bool> bad{false};
    // Lane transposition of one group, 16 dwords of 64 lanes at a time: the rows of a 4 KB tile are gathered lane by lane
    // (sequential reads of each sequence) and written out as whole 256-byte rows.  Writing a sequence's dwords straight to
    // their places instead -- one store every 256 bytes -- made this loop two thirds of a per-query s4g_db_create.
    auto pack = [&](uint32_t g0, uint32_t g1) {
        alignas(64) uint32_t tile[64][16];
        const uint8_t* src[64];
        uint32_t len[64];
        for (uint32_t g = g0; g < g1; ++g) {
            uint32_t* base = res + grp_off[g];
            const uint32_t lanes = std::min<uint32_t>(64, n - g * 64);
            for (uint32_t lane = 0; lane < lanes; ++lane) {
                const uint32_t id = order[p0 + (size_t)g * 64 + lane];
                perm_g[(size_t)g * 64 + lane] = id;
                perm_l[(size_t)g * 64 + lane] = rank_of(id);
                src[lane] = seq_ptrs ? seq_ptrs[id] : codes + offsets[id];
                len[lane] = db->len[id];
            }
            // four residues per dword; a byte >= 26 anywhere is an error (SWAR test: bit 7 of b | ((b & 0x7F) + 102))
            uint32_t any_bad = 0;
            for (uint32_t c0 = 0; c0 < grp_len4[g]; c0 += 16) {
                const uint32_t cn = std::min(16u, grp_len4[g] - c0);
                for (uint32_t lane = 0; lane < lanes; ++lane) {
                    const uint32_t full = len[lane] / 4;                               // complete dwords of this sequence
                    const uint32_t run = full > c0 ? std::min(cn, full - c0) : 0;
                    uint32_t* row = tile[lane];
                    memcpy(row, src[lane] + 4 * (size_t)c0, 4 * (size_t)run);
                    for (uint32_t c = 0; c < run; ++c) any_bad |= (row[c] | ((row[c] & 0x7F7F7F7Fu) + 0x66666666u)) & 0x80808080u;
                    for (uint32_t c = run; c < cn; ++c) {
                        uint32_t w = 0x1F1F1F1Fu;
                        if (c0 + c == full && (len[lane] & 3))                         // the last, partial dword
                            for (uint32_t j = 4 * full; j < len[lane]; ++j) {
                                const uint32_t x = src[lane][j];
                                if (x >= 26) any_bad = 1;
                                w = (w & ~(0xFFu << (8 * (j & 3)))) | (x << (8 * (j & 3)));
                            }
                        row[c] = w;
                    }
                }
                for (uint32_t c = 0; c < cn; ++c) {
                    uint32_t* out = base + (size_t)(c0 + c) * 64;
                    for (uint32_t lane = 0; lane < lanes; ++lane) out[lane] = tile[lane][c];
                }
            }
            if (any_bad) { bad = true; return; }
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
    if (use_block) {
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
        memcpy(blob.data() + o_off, grp_off.data(), (size_t)G * 8);
        memcpy(blob.data() + o_len4, grp_len4.data(), (size_t)G * 4);
        memcpy(blob.data() + o_pg, perm_g.data(), (size_t)G * 256);
        memcpy(blob.data() + o_pl, perm_l.data(), (size_t)G * 256);
        memcpy(blob.data() + o_col, col_ids.data(), (size_t)n * 4);
        if (extra_bytes) {      // the database is this one tile: the per-sequence tables of the pair kernels ride along
            uint64_t* addr = reinterpret_cast<uint64_t*>(blob.data() + o_extra);
            const uint64_t o_len = o_extra + ((uint64_t)n * 8 + 255) / 256 * 256;
            uint32_t* lens = reinterpret_cast<uint32_t*>(blob.data() + o_len);
            for (uint32_t pos = 0; pos < n; ++pos) {
                const uint32_t id = order[p0 + pos];
                addr[id] = (uint64_t)(uintptr_t)(t.d_res + grp_off[pos >> 6] + (pos & 63));
                lens[id] = db->len[id];
            }
            db->d_seq_addr = reinterpret_cast<uint64_t*>(blk + o_extra);
            db->d_seq_len = reinterpret_cast<uint32_t*>(blk + o_len);
            db->seq_tables_in_block = true;
            db->id_of_pos.assign(order.begin() + p0, order.begin() + p1);
        }
        if (extra_off) *extra_off = o_extra;
        if ((e = hipMemcpy(blk, blob.data(), need, hipMemcpyHostToDevice)) != hipSuccess) return e;
        t.h_grp_off.swap(grp_off);
        t.pos0 = p0;
        return hipSuccess;
    }
    if ((e = up((void**)&t.d_res, res, dwords * 4)) != hipSuccess) return e;
    if ((e = up((void**)&t.d_grp_off, grp_off.data(), (size_t)G * 8)) != hipSuccess) return e;
    if ((e = up((void**)&t.d_grp_len4, grp_len4.data(), (size_t)G * 4)) != hipSuccess) return e;
    if ((e = up((void**)&t.d_perm_global, perm_g.data(), (size_t)G * 64 * 4)) != hipSuccess) return e;
    if ((e = up((void**)&t.d_perm_local, perm_l.data(), (size_t)G * 64 * 4)) != hipSuccess) return e;
    if ((e = up((void**)&t.d_col_ids, col_ids.data(), (size_t)n * 4)) != hipSuccess) return e;
    t.h_grp_off.swap(grp_off);
    t.pos0 = p0;
    return hipSuccess;
}


// codes: host copy of the residues, or NULL when d_codes (device copy, caller-owned) is given
// seq_ptrs / seq_lens: the sequences as separate host arrays (small databases only; codes and offsets are NULL then)
int db_create_impl(s4g_db** out, int device, const uint8_t* codes, const uint8_t* d_codes, const uint64_t* offsets, uint32_t n_seqs,
                   uint32_t max_tile_seqs, const uint8_t* const* seq_ptrs = nullptr, const uint32_t* seq_lens = nullptr);

constexpr uint64_t kDevicePackMin = 8ull << 20;     // residues from which the lane transposition runs on the device

S4G_API int s4g_db_create_scattered(s4g_db** out, int device, const uint8_t* const* seqs, const uint32_t* lens, uint32_t n_seqs) {
    if (!out || (n_seqs && (!seqs || !lens))) return fail(S4G_ERR_INVALID, "NULL argument");
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_seqs; ++i) {
        if (lens[i] && !seqs[i]) return fail(S4G_ERR_INVALID, "NULL sequence at %u", i);
        total += lens[i];
    }
    if (total < kDevicePackMin) return db_create_impl(out, device, nullptr, nullptr, nullptr, n_seqs, 0, seqs, lens);
    // large: one contiguous copy, then the regular path (raw upload + device packing)
    std::vector<uint8_t> codes(total);
    std::vector<uint64_t> off((size_t)n_seqs + 1, 0);
    for (uint32_t i = 0; i < n_seqs; ++i) {
        if (lens[i]) memcpy(codes.data() + off[i], seqs[i], lens[i]);
        off[i + 1] = off[i] + lens[i];
    }
    return db_create_impl(out, device, codes.data(), nullptr, off.data(), n_seqs, 0);
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
    const double t0 = now_s();
    std::vector<uint64_t> offsets((size_t)n_seqs + 1);
    HIP_TRY(hipMemcpy(offsets.data(), d_offsets, offsets.size() * 8, hipMemcpyDeviceToHost));    // only the lengths come to the host
    const double t1 = now_s();
    const int rc = db_create_impl(out, device, nullptr, d_codes, offsets.data(), n_seqs, max_tile_seqs);
    if (g_create_clock.on) fprintf(stderr, "[s4g profile] s4g_db_create_from_device: offsets to host %.3f s, create %.3f s\n", t1 - t0, now_s() - t1);
    return rc;
}

int db_create_impl(s4g_db** out, int device, const uint8_t* codes, const uint8_t* d_codes, const uint64_t* offsets, uint32_t n_seqs,
                   uint32_t max_tile_seqs, const uint8_t* const* seq_ptrs, const uint32_t* seq_lens) {
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
    uint64_t total_res = 0;
    for (uint32_t i = 0; i < n_seqs; ++i) {
        uint64_t l = seq_ptrs ? seq_lens[i] : offsets[i + 1] - offsets[i];
        if ((!seq_ptrs && offsets[i + 1] < offsets[i]) || l > 0x7FFFFFF0ull) { delete db; return fail(S4G_ERR_INVALID, "bad offsets at %u", i); }
        db->len[i] = (uint32_t)l;
        total_res += l;
    }
    // length sort, longest first (ties by original index): items are issued in this order so the longest
    // work runs first and the tail of every launch is short work.
    std::vector<uint32_t> order(n_seqs);
    {
        uint32_t max_len = 0;
        for (uint32_t i = 0; i < n_seqs; ++i) max_len = std::max(max_len, db->len[i]);
        if (max_len < (1u << 22)) {         // counting sort by length, descending, ids ascending inside a length: O(n)
            std::vector<uint32_t> start((size_t)max_len + 2, 0);
            for (uint32_t i = 0; i < n_seqs; ++i) ++start[max_len - db->len[i] + 1];
            for (size_t k = 1; k < start.size(); ++k) start[k] += start[k - 1];
            for (uint32_t i = 0; i < n_seqs; ++i) order[start[max_len - db->len[i]]++] = i;
        } else {
            std::iota(order.begin(), order.end(), 0u);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return db->len[a] > db->len[b]; });
        }
    }

    db->pos_of_id.resize(n_seqs);
    for (uint32_t pp = 0; pp < n_seqs; ++pp) db->pos_of_id[order[pp]] = pp;
    g_create_clock.t[0] += now_s() - t_begin; g_create_clock.n++;
    // Databases from 8 MB to 16 GB of residues are transposed on the device: one upload of the raw codes, then one
    // sw_pack_tile launch per tile.  Smaller ones are packed by the host (a launch and two allocations would cost more
    // than they save); larger ones tile by tile on the host, so that the raw copy never doubles the footprint.
    DevBuf<uint8_t> d_raw;
    DevBuf<int> d_bad;
    const uint8_t* raw_dev = d_codes;       // codes that already live on the device are packed there whatever their size
    if (d_codes || (!seq_ptrs && total_res >= kDevicePackMin && total_res <= (16ull << 30))) {
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
        // a database that is one small tile also carries the per-sequence tables of the pair kernels in its block (built and
        // uploaded with the tile)
        const bool whole = (p == 0 && q == n_seqs);
        uint64_t tab_off = 0;
        hipError_t e = build_tile(db, db->tiles.back(), order, p, q, codes, offsets, &err, whole ? (uint64_t)n_seqs * 12 + 512 : 0, &tab_off,
                                  raw_dev, d_bad.p, seq_ptrs);
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

S4G_API int s4g_db_create(s4g_db** out, int device, const uint8_t* codes, const uint64_t* offsets, uint32_t n_seqs) {
    return s4g_db_create_tiled(out, device, codes, offsets, n_seqs, 0);
}

S4G_API int s4g_db_destroy(s4g_db* db) {
    if (!db) return S4G_OK;
    if (db->refs.fetch_sub(1) > 1) return S4G_OK;       // a traceback job in flight still reads it: its s4g_traceback_finish releases
    DeviceGuard guard(db->device);
    {   // launches that still read the tiles must have finished before they are freed
        DeviceCtx* cx = device_ctx(db->device);
        std::lock_guard<std::mutex> lk(cx->mu);
        if (cx->last_use) (void)hipEventSynchronize(cx->last_use);
    }
    if (db->stream) { db_stream_close(db); db->tiles.clear(); }         // (its tiles own no device memory)
    for (Tile& t : db->tiles) free_tile(t, device_ctx(db->device));
    if (db->cached) chain_cache_release(db);
    else if (!db->seq_tables_in_block) { (void)hipFree(db->d_seq_addr); (void)hipFree(db->d_seq_len); }
    if (db->d_pos_of_id) (void)hipFree(db->d_pos_of_id);
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

S4G_API int s4g_db_save(const s4g_db* db, const char* path) {
    if (!db || !path) return fail(S4G_ERR_INVALID, "NULL argument");
    if (db->cached) return fail(S4G_ERR_INVALID, "a database of cached chains (s4g_db_create_cached) has no tiles to save");
    if (db->stream) return fail(S4G_ERR_INVALID, "a streamed database is a file already");
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
    // A small set -- the one query of the reference's per-query call, database_alignment.cpp:83-86 -- takes ONE device block out of a
    // per-device cache and ONE asynchronous upload out of pinned staging: three hipMalloc, three synchronous pageable copies (each of
    // which also waits for whatever the device is still doing, e.g. the previous call's traceback) and three hipFree per call otherwise.
    const uint64_t o_off = (padded.size() + 255) / 256 * 256, o_len = o_off + (off.size() * 8 + 255) / 256 * 256;
    const uint64_t blk_need = o_len + ((uint64_t)(n ? n : 1) * 4 + 255) / 256 * 256;
    if (blk_need <= (1u << 20)) {
        DeviceCtx* cx = device_ctx(device);
        std::lock_guard<std::mutex> lk(cx->mu);
        for (size_t k = 0; k < cx->query_blocks.size(); ++k)
            if (cx->query_blocks[k].second >= blk_need) {
                q->d_block = cx->query_blocks[k].first; q->block_bytes = cx->query_blocks[k].second;
                cx->query_blocks.erase(cx->query_blocks.begin() + k);
                break;
            }
        hipError_t e = hipSuccess;
        if (!q->d_block) {
            q->block_bytes = std::max<uint64_t>(blk_need, 16u << 10);
            e = hipMalloc((void**)&q->d_block, q->block_bytes);
        }
        if (e == hipSuccess && cx->query_pin_bytes < blk_need) {
            if (cx->h_query_pin) { (void)hipHostFree(cx->h_query_pin); cx->h_query_pin = nullptr; cx->query_pin_bytes = 0; }
            e = hipHostMalloc((void**)&cx->h_query_pin, 1u << 20, hipHostMallocDefault);
            if (e == hipSuccess) cx->query_pin_bytes = 1u << 20;
        }
        if (e == hipSuccess && !cx->query_copy_stream) e = hipStreamCreateWithFlags(&cx->query_copy_stream, hipStreamNonBlocking);
        if (e == hipSuccess) {
            memcpy(cx->h_query_pin, padded.data(), padded.size());
            memcpy(cx->h_query_pin + o_off, off.data(), off.size() * 8);
            if (n) memcpy(cx->h_query_pin + o_len, q->len.data(), (size_t)n * 4);
            // On a stream of its own and waited for HERE: when the call returns the set is in device memory, whichever stream the
            // caller queues its scan on (a torch.cuda.Stream() is hipStreamNonBlocking: the null stream would order nothing against
            // it), and the staging buffer is free for the next set.  The wait is for these few hundred KB, not for the device: a
            // traceback still running on trace_stream, or a caller's scan on its own stream, goes on underneath.  (The block itself
            // has no reader left: s4g_queries_destroy waits for the device before a block goes back into the cache.)
            e = hipMemcpyAsync(q->d_block, cx->h_query_pin, blk_need, hipMemcpyHostToDevice, cx->query_copy_stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(cx->query_copy_stream);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (q->d_block) (void)hipFree(q->d_block);
            delete q;
            return fail(S4G_ERR_HIP, "query upload failed: %s", hipGetErrorString(e));
        }
        q->d_codes = reinterpret_cast<uint8_t*>(q->d_block);
        q->d_off = reinterpret_cast<uint64_t*>(q->d_block + o_off);
        q->d_len = reinterpret_cast<uint32_t*>(q->d_block + o_len);
        *out = q;
        return S4G_OK;
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
    if (q->d_block) {
        // back into the cache.  hipFree used to wait for the device here, and a caller may have queued a scan of these queries on a
        // stream of its own: keep that guarantee (with nothing in flight -- the usual case: results have been waited for -- it costs
        // microseconds), then the block can go to the next small set
        (void)hipDeviceSynchronize();
        DeviceCtx* cx = device_ctx(q->device);
        std::lock_guard<std::mutex> lk(cx->mu);
        if (cx->query_blocks.size() < 8) cx->query_blocks.emplace_back(q->d_block, q->block_bytes);
        else (void)hipFree(q->d_block);
    } else { (void)hipFree(q->d_codes); (void)hipFree(q->d_off); (void)hipFree(q->d_len); }
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


// s4g_internal.hpp -- declarations shared by the translation units of libs4g_hip.so (nothing here is exported; the
// C ABI is include/s4g_hip.h).  Files: s4g_core.hip (errors, scorer, database, queries), s4g_scan.hip (plan, scan
// launches, top-k), s4g_pairs.hip (one wave per pair: scores and traceback), s4g_trace.hip (traceback entry points and
// the 64-pairs-per-wave route).
#pragma once
#include "../../include/s4g_hip.h"
#include "sw_kernels.hip.h"
#include "sw_pair.hip.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

using namespace s4g;

#define S4G_API extern "C" __attribute__((visibility("default")))

// error reporting: every failure sets the thread's message (s4g_last_error) and returns a code
int fail(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(S4G_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                 \
    } while (0)

int device_count_raw();

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; (void)hipSetDevice(dev); }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

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
    mutable std::vector<uint32_t> h_col_of_pos;     // tile column (rank of the id) of the sequence at tile position p; built on first use (tile_as_pairs)
    char* d_block = nullptr;            // small tiles: every array above lives in this one allocation (recycled through
    uint64_t block_bytes = 0;           // DeviceCtx::block_cache: sift4g creates a candidate database per query)
    uint64_t pos0 = 0;                  // position of the tile's first sequence in the global length-sorted order
};

struct DbStream;                 // s4g_stream.hip: a packed database file scanned tile by tile (s4g_db_open_streamed)

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
    mutable uint32_t* d_pos_of_id = nullptr;    // [n_seqs] pos_of_id on the device (built by the first list call, s4g_lists.hip)
    mutable std::vector<uint32_t> id_of_pos;    // position in the length-sorted order -> original index
    bool seq_tables_in_block = false;           // d_seq_addr / d_seq_len are part of tiles[0].d_block (or of d_cached_block)
    // s4g_db_create_cached (s4g_chains.hip): no tiles -- the sequences live in the device's chain arena, the database is its two
    // tables; it serves the one-wave-per-pair routes only
    bool cached = false;
    char* d_cached_block = nullptr;
    uint64_t cached_block_bytes = 0;
    mutable DbStream* stream = nullptr;         // s4g_db_open_streamed: no resident tiles, only s4g_scan_topk* serve it
    mutable std::atomic<int> refs{1};           // the caller's handle + one per traceback job in flight (s4g_traceback_submit): the last
                                                // s4g_db_destroy / s4g_traceback_finish releases the device memory
};
void chain_cache_release(s4g_db* db);           // s4g_chains.hip
// s4g_stream.hip: tiles of a streamed database (db->stream != NULL) live in the file; two slots on the device
void db_stream_close(s4g_db* db);
int db_stream_load(const s4g_db* db, size_t ti);                                  // read tile ti (blocking) and queue its upload
int db_stream_tile(const s4g_db* db, size_t ti, hipStream_t stream, Tile* out);   // `stream` waits for the upload; *out points into the slot
int db_stream_consumed(const s4g_db* db, size_t ti, hipStream_t stream);          // record: everything queued so far is done with the slot
void db_stream_note_scan(const s4g_db* db, double wall_s);

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
    int8_t* d_mat = nullptr;                // the substitution table the kernels read; see ensure_mat
    uint64_t mat_serial = 0;                // s4g_scorer::serial of the table d_mat holds (0: none)
    hipEvent_t mat_ready = nullptr;         // its upload
    uint32_t* d_qlist = nullptr;     // query-id lists of the current call (pairs for the packed tiers, singles for int32)
    uint64_t qlist_cap = 0;
    int32_t* d_tile_scores = nullptr;   // s4g_scan_topk: [query batch][tile columns]
    uint64_t tile_scores_bytes = 0;
    uint32_t grid_i32 = 0, grid_i16 = 0, grid_f16 = 0, grid_drift = 0, grid_col = 0;
    uint32_t grid_pair[6] = {0, 0, 0, 0, 0, 0};
    uint32_t grid_pair_trace[6] = {0, 0, 0, 0, 0, 0};
    uint32_t grid_pair_band = 0, grid_pair_trace_band = 0, grid_pair_recheck = 0;
    uint32_t grid_pair_coop[3] = {0, 0, 0}, grid_pair_trace_coop[3] = {0, 0, 0};     // sw_pair_scan_coop / sw_pair_trace_coop<2, 3, 4>
    uint32_t grid_pair_pk[3] = {0, 0, 0};   // sw_pair_scan_pk<1, 2, 3> (sw_pair_pk.hip.h: two rows per register)
    uint64_t stat_pair_pk = 0;              // pairs of the last pair submission that ran there
    char* h_trace_pin = nullptr;        // pinned staging of traceback_pairs: items up, results back
    std::vector<std::pair<char*, uint64_t>> query_blocks;        // device blocks of destroyed small query sets (stream-ordered reuse on the null stream)
    char* h_query_pin = nullptr; uint64_t query_pin_bytes = 0;   // pinned staging of s4g_queries_create (free again when the call returns)
    hipStream_t query_copy_stream = nullptr; // a small query set is uploaded here and WAITED for before s4g_queries_create returns: the set is
                                             // then valid on every stream of the caller's (the null stream orders nothing against
                                             // a hipStreamNonBlocking stream), and the wait is for this copy alone, not for the device
    hipStream_t trace_stream = nullptr;      // tracebacks in flight (s4g_traceback_submit) run here, beside the next call's scoring on the null stream
    hipEvent_t trace_inputs = nullptr;       // recorded on the null stream at submit: the uploads the job's kernels read
    hipEvent_t trace_last = nullptr;         // behind the last job queued on trace_stream: what shares its buffers (the traceback arena, its
                                             // counters, a change of the substitution table) on another stream waits for it
    // pinned host memory of the tracebacks in flight (s4g_traceback_submit): slabs, a job takes a piece of the current one and gives
    // it back at finish; a slab whose pieces have all come back starts over.  (A buffer per job was 200 hipHostMalloc of a megabyte
    // for the 200 per-query calls of a run -- a quarter of a millisecond each -- and as many hipHostFree.)
    struct PinSlab { char* p = nullptr; uint64_t cap = 0, used = 0; int live = 0; };
    std::vector<PinSlab> trace_pin_slabs;
    uint64_t trace_pin_bytes = 0;
    // what the device's last k-mer prefilter handle left behind for the next one (s4g_prefilter.hip): its pinned staging buffer and its
    // work arena -- pinning 12 MB and allocating and releasing a few hundred MB cost a handle 2-3 ms, as much as its whole add call
    char* pf_spare_stage = nullptr; uint64_t pf_spare_stage_cap = 0;
    char* pf_spare_work = nullptr;  uint64_t pf_spare_work_cap = 0;
    hipStream_t coop_stream = nullptr;  // they run next to the one-wave-per-pair launches of the same submission, not behind them
    hipEvent_t coop_fork = nullptr, coop_join = nullptr;
    char* d_coop_ws = nullptr;          // their boundary rows (s4g_pairs.hip, ensure_coop_ws)
    uint64_t coop_ws_bytes = 0;
    char* d_band_ws = nullptr;          // boundary rows of sw_pair_scan_band, one slot per wave
    uint64_t band_ws_bytes = 0;
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
    hipEvent_t merge_done[2] = {nullptr, nullptr};          // s4g_scan_topk with alternating tiles: the running lists are merged in tile order
    int32_t* d_ws_side = nullptr;
    uint64_t ws_side_bytes = 0;
    uint64_t ws_budget = 40ull << 30;   // strip-boundary workspace budget of the call in progress (make_plan sets it: s4g_scan.hip call_workspace_budget)
    std::vector<std::pair<char*, uint64_t>> block_cache;    // freed small-tile blocks, reused by the next s4g_db_create
    char* arena = nullptr;              // traceback temporaries
    uint64_t arena_bytes = 0;
    // what the last s4g_scan / s4g_scan_topk call on this device did (s4g_scan_stats): queries per tier summed over tiles and
    // batches, queries that ran speculatively, pairs handed to the one-wave-per-pair kernels; the number of marked entries the
    // recheck pass recomputed is counted on the device (d_counter + kStatRechecked)
    // candidate lists on the packed list kernel (s4g_lists.hip): pinned staging, device tables, gathered residues
    char* h_lst_stage = nullptr;
    uint64_t lst_stage_bytes = 0;
    hipEvent_t lst_stage_done = nullptr;
    char* d_lst = nullptr;
    uint64_t lst_bytes = 0;
    uint32_t* d_lst_res = nullptr;
    uint64_t lst_res_bytes = 0;
    uint32_t grid_lists = 0;
    uint32_t grid_pipe = 0;                 // blocks of sw_scan_lists<R, true> the device holds (scan_tile_pipelined)
    uint32_t grid_pipe_wide = 0;            // ... of its twelve-wave form (latency-bound calls of few long queries)
    uint32_t grid_pipe2 = 0, grid_pipe2_wide = 0;   // ... of the forms whose items may pair two GROUPS under a lone query (odd query counts)
    uint64_t stat_tier[4] = {0, 0, 0, 0};
    uint64_t stat_spec_queries = 0, stat_pair_items = 0, stat_list_pairs = 0;
    uint64_t stat_coop_scan = 0, stat_coop_trace = 0;      // pairs of the last pair submission / traceback call on the cooperative kernels
    // clock samples of the scan launches of the last call (s4g_scan_clock; sw_kernels.hip.h clk_enter / clk_leave): a region of
    // 4 counters per block for each of the call's first kClkLaunches launches
    int max3_state = 0;                  // ensure_max3: 0 not yet checked, 1 the integer-max3-on-f16-patterns argument holds here, -1 it does not
    char* d_order[2] = {nullptr, nullptr};   // item orders of launches of few, long items (s4g_scan.hip, item_order): main / side stream
    uint64_t order_cap[2] = {0, 0};
    unsigned long long* d_clk = nullptr;
    uint64_t clk_cap = 0;                // counters allocated
    struct ClkLaunch { uint64_t off; uint32_t blocks; int tier; };
    std::vector<ClkLaunch> clk_launches;
};
constexpr unsigned kClkLaunches = 16;
constexpr int kStatRechecked = 40;      // index into DeviceCtx::d_counter
constexpr int kListCounter = 44;        // work queue of sw_scan_lists

DeviceCtx* device_ctx(int device);

// S4G_TRACEBACK_TIES (read per call): which gap a cell's H is traced into when the vertical (UP, gap in the target) and the
// horizontal (LEFT, gap in the query) one give the same value and the diagonal does not.  "up" (default) or "left".  swsharp's own
// choice is not known here; INTEGRATION.md lists the switch with the other unpinned assumptions.
inline uint32_t traceback_left_first() {
    const char* v = getenv("S4G_TRACEBACK_TIES");
    return v && strcmp(v, "left") == 0 ? 1u : 0u;
}

struct s4g_queries {
    int device = 0;
    uint32_t n = 0;
    uint8_t* d_codes = nullptr;
    uint64_t* d_off = nullptr;
    uint32_t* d_len = nullptr;
    char* d_block = nullptr;                 // small sets (a per-query call's one query): codes | offsets | lengths in ONE block out of the
    uint64_t block_bytes = 0;                // device's query-block cache, uploaded by one asynchronous copy; destroy hands the block back
    std::vector<uint32_t> len;
    std::vector<uint8_t> codes;              // host copy (unpadded) for per-scorer bounds
    std::vector<uint64_t> off;
    uint32_t max_len = 0;
    mutable std::vector<uint64_t> score_bound;
    mutable uint64_t bound_scorer = 0;
};

constexpr uint32_t kLongSeq = 12288;                   // sequences longer than this get tiles of their own: beyond ~16k residues
                                                       // the per-wave strip-boundary workspace no longer fits the budget at full grid

// Fork-join over [0, n): f(part, parts, begin, end) on `parts` host threads (the caller's included).  For the O(pairs) passes of
// the list route, which are a third of a call's time when left to one thread; small inputs run inline.  The workers are a
// small persistent pool (s4g_core.hip): starting a std::thread costs ~100 us on the GPU box, as much as the work of a part.
inline unsigned parallel_parts(uint64_t n, uint64_t min_per_part) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min(hw, 16u), n / std::max<uint64_t>(1, min_per_part)));
}
void host_pool_run(unsigned parts, void (*fn)(void* ctx, unsigned part), void* ctx);     // parts <= 16; part 0 runs on the caller
template <class F>
inline void parallel_for(uint64_t n, unsigned parts, F f) {
    if (parts <= 1) { f(0u, 1u, (uint64_t)0, n); return; }
    struct Ctx { F* f; uint64_t n; unsigned parts; } ctx{&f, n, parts};
    host_pool_run(parts, [](void* c, unsigned p) {
        Ctx* x = static_cast<Ctx*>(c);
        (*x->f)(p, x->parts, x->n * p / x->parts, x->n * (p + 1) / x->parts);
    }, &ctx);
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

// Warm-up (s4g_warmup, s4g_diag.hip): each translation unit forces the load of its own code object by asking the runtime
// for the occupancy of its kernels (which also fills the grid caches the first real call would compute), on cx's device.
int warm_scan_kernels(DeviceCtx* cx);      // s4g_scan.hip
int warm_pair_kernels(DeviceCtx* cx);      // s4g_pairs.hip
int warm_list_kernels(DeviceCtx* cx);      // s4g_lists.hip
int warm_core_kernels(DeviceCtx* cx);      // s4g_core.hip
int warm_prefilter_kernels(DeviceCtx* cx); // s4g_prefilter.hip
int pair_stage(DeviceCtx* cx, uint64_t n, s4g::PairItem** out);
// s4g_scan.hip
void ensure_bounds(const s4g_queries* q, const s4g_scorer* scorer);
int device_props(DeviceCtx* cx);
int scratch_acquire(const s4g_db* db, hipStream_t stream);
int scratch_release(const s4g_db* db, hipStream_t stream);
// Every exit of a call that holds the scratch buffers -- the error exits too -- must leave `last_use` recorded behind all of
// its launches, side stream included: s4g_db_destroy and the block cache order themselves on that event, and a failure
// halfway (a hipMalloc that runs out of memory in ensure_scratch, say) would otherwise leave side-stream kernels in flight
// behind a stale event.  release() is the normal exit; the destructor covers the others.
struct ScratchScope {
    const s4g_db* db;
    hipStream_t stream;
    bool armed = false, forked = false;
    ScratchScope(const s4g_db* d, hipStream_t s) : db(d), stream(s) {}
    int acquire() { const int rc = scratch_acquire(db, stream); armed = rc == S4G_OK; return rc; }
    int release() { armed = false; return scratch_release(db, stream); }
    ~ScratchScope() {
        if (!armed) return;
        DeviceCtx* cx = device_ctx(db->device);
        if (forked && cx->side_stream && cx->side_join) {
            (void)hipEventRecord(cx->side_join, cx->side_stream);
            (void)hipStreamWaitEvent(stream, cx->side_join, 0);
        }
        if (cx->last_use) (void)hipEventRecord(cx->last_use, stream);
    }
};

// s4g_pairs.hip
int ensure_seq_tables(const s4g_db* db);
// stages and submits the pairs sel[] to the one-wave-per-pair kernels on `stream` (asynchronous; caller holds cx->mu): the score
// of pair sel[i] will be at (*d_sc_out)[(*where)[i]]
int pairs_enqueue(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, const std::vector<uint32_t>& sel,
                  const uint32_t* pair_q, const uint32_t* pair_t, std::vector<uint32_t>* where, int32_t** d_sc_out, hipStream_t stream,
                  bool pk_allowed = true);       // pk_allowed: the packed pair kernels may take what their window admits (not beside the list kernel)
// s4g_lists.hip
int score_lists_device(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, const uint32_t* sel_idx, uint32_t n_sel,
                       const uint32_t* pair_q, const uint32_t* pair_t, int32_t* h_scores, bool force, uint32_t forced_split, bool* used);
// s4g_lists.hip: a tile scanned by a few queries on the block-cooperative strip pipeline (sw_scan_lists<R, true>); *used = false: not applicable
constexpr uint64_t kSpecSpan = 6144;    // scores a speculative query can reach without a recheck, at least (s4g_scan.hip make_plan, s4g_lists.hip scan_tile_pipelined)
int scan_tile_pipelined(const s4g_db* db, const Tile& t, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* scorer,
                        int32_t* d_scores, uint64_t stride, const uint32_t* d_perm, hipStream_t stream, bool* used,
                        bool allow_spec = false, bool* speculative = nullptr);   // *speculative: the caller must run recheck_marked over the tile's scores
int tile_as_pairs(const s4g_db* db, const Tile& t, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* scorer,
                  int32_t* d_scores, uint64_t stride, bool local_cols, hipStream_t stream, uint32_t n_pos);
uint32_t split_groups(const s4g_db* db, const Tile& t, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, uint32_t flags);
bool scan_prefers_pairs(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, uint32_t flags);
int scan_as_pairs(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* scorer,
                  int32_t* d_scores, hipStream_t stream, bool* done);
// after a speculative sw_scan_pkd launch: recomputes the marked entries of scores[q_count][n_cols] (row stride `stride`) exactly;
// d_col_ids maps a column to its target id (NULL: identity); max_lt = longest target behind those columns.  cx->d_mat holds the
// scorer's table already (ensure_scratch).  Caller holds cx->mu.
// d_mat <- the scorer's table, on `stream`, unless it is there already (then `stream` only waits for that upload).  The table
// sits in pageable host memory, so an upload blocks the HOST until everything queued on `stream` before it is done -- which is
// what made a pairs submission on the side stream serialise the caller behind the side stream's earlier kernels.  Callers have
// passed scratch_acquire (no launch of another scorer is still reading d_mat).  Caller holds cx->mu.
int ensure_mat(DeviceCtx* cx, const s4g_scorer* scorer, hipStream_t stream);
// Before a device's first launch of a kernel that uses v_pk_maximum3_f16 as an integer max3 (sw_scan_pkd, sw_scan_pk<ArithI16M3>,
// sw_scan_lists): a self-test on denormal-range patterns (denorm_selftest, sw_kernels.hip.h).  Fails the call -- loudly, with the
// reason -- on a device or build that does not preserve f16 denormals.  Caller holds cx->mu.
int ensure_max3(DeviceCtx* cx);
int recheck_marked(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t q_begin, uint32_t q_count,
                   int32_t* d_scores, uint64_t stride, uint32_t n_cols, const uint32_t* d_col_ids, uint32_t max_lt, hipStream_t stream);
int traceback_pairs(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, const std::vector<uint32_t>& sel,
                    const uint32_t* pair_q, const uint32_t* pair_t, const int32_t* pair_score,
                    int32_t* h_out, const uint64_t* path_off, uint8_t* h_paths, s4g_trace_job* job = nullptr);

// s4g_scan.hip -- C ABI: the scan (step 1 of swsharp's alignDatabase as sift4g calls it,
// /root/reference/sift4g/src/database_alignment.cpp:83-86) and the device top-k (steps 2-3): per-tile plan, tier routing,
// persistent-grid launches, streaming scan + running top-k.
#include "s4g_internal.hpp"
#include <hipcub/hipcub.hpp>

// ---------------------------------------------------------------------------

constexpr int kStripRows = 32;      // sw_scan_i32
constexpr int kStripRows16 = 32;    // sw_scan_i16x2
constexpr uint64_t kWorkspaceBudget = 40ull << 30;   // strip-boundary workspace cap (round 5: 40 GB, so that a tile of sequences up to 12 288 residues keeps the full grid of 4096 waves); longer-sequence tiles run on a smaller grid
constexpr uint64_t kWorkspaceAlwaysFits = 8ull << 30; // below this nobody asks the runtime how much memory is free (every tile of sequences up to 3000 residues: 6.3 GB)
constexpr uint64_t kTileScoreBudget = 4ull << 30;    // s4g_scan_topk: bytes of the [query batch][tile columns] buffer

// Work of one tile launch: which queries run in which kernel.  Tiers, fastest first; a query goes to the first one that
// is EXACT for it (score bound = sum_i max(0, max_b M[q_i][b]); see sw_kernels.hip.h for each window):
//   drift  sw_scan_pkc          bound + B + tile columns * ext <= 0x7BFF               6.5 ops / pair of cells, 3 of them double-rate
//          (column frame, round 5; S4G_SCAN_DIAG_FRAME or S4G_SCAN_FRAME=diag selects sw_scan_pkd, the anti-diagonal frame of
//          rounds 1-4, whose window is bound + B + (rows + tile columns) * ext)
//          ... or speculatively: only the frame B + columns * ext (+ a span) fits; the kernel marks results that
//          came near the window's end and sw_pair_recheck recomputes those in int32 (Plan::speculative)
//   m3     sw_scan_pk<I16M3>    bound + B <= 0x7BFF   (or ArithF16: bound <= 4096)     7.5
//          ... or speculatively (I16M3 only) when no drift frame fits: results with best + B beyond 0x7BFF - max score are marked
//   pk     sw_scan_pk<I16>      bound <= 32767                                          9
//   wide   sw_scan_i32          anything                                                7.5 per single cell
struct TileWork {
    uint32_t off_drift = 0, n_drift = 0;   // offsets / pair counts into Plan::flat (pairs = 2 ids each)
    uint32_t off_m3 = 0, n_m3 = 0;
    uint32_t off_pk = 0, n_pk = 0;
    uint32_t off_wide = 0, n_wide = 0;     // single ids
    uint32_t prio_cost[4] = {0, 0, 0, 0};  // per tier (3: int32): a quarter of the per-wave share of its launch in rows * len4 (item_priority)
};

struct Plan {
    std::vector<TileWork> tiles;    // one per database tile
    std::vector<uint32_t> flat;     // all id lists, uploaded once per call
    bool f16_arith = false;         // m3 tier runs ArithF16 instead of ArithI16M3
    bool wide_drift = false;        // int32 tier runs sw_scan_i32d (drifting frame) instead of sw_scan_i32
    uint32_t bias = 0;              // bias B of the maximum3-based tiers
    uint32_t grid_i32 = 0, grid_i16 = 0, grid_m3 = 0, grid_drift = 0;   // full persistent grids
    uint64_t ws_bytes = 0;          // workspace needed for the largest tile launch
    uint64_t n_by_tier[4] = {0, 0, 0, 0};   // queries per tier, summed over tiles (reporting)
    bool speculative = false;       // some query runs in the drift tier on its frame alone: marked results need sw_pair_recheck
    uint64_t n_spec = 0;            // (query, tile) combinations routed speculatively (reporting)
    uint32_t spec_base = 0;         // sw_scan_pkd: 0x7BFF - bias - (max score + 2 ext); sw_scan_pkc: 0x7BFF - (max score + ext), in the frame
    bool col_frame = true;          // drift tier = sw_scan_pkc (column frame) rather than sw_scan_pkd (anti-diagonal frame)
};
// S4G_SCAN_FRAME=diag|col: the frame of calls that do NOT set S4G_SCAN_DIAG_FRAME, for a whole process (A/B runs of bench.py)
static int frame_env() {
    static const int v = []() { const char* e = getenv("S4G_SCAN_FRAME"); return !e ? 0 : (e[0] == 'd' ? 1 : (e[0] == 'c' ? 2 : 0)); }();
    return v;
}

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

int warm_scan_kernels(DeviceCtx* cx) {
    s4g_db probe;
    probe.device = cx->device;
    int rc = grid_for(&probe, (const void*)sw_scan_i32<kStripRows>, &cx->grid_i32);
    if (!rc) rc = grid_for(&probe, (const void*)sw_scan_pk<ArithI16, kStripRows16>, &cx->grid_i16);
    if (!rc) rc = grid_for(&probe, (const void*)sw_scan_pk<ArithI16M3, kStripRows16>, &cx->grid_f16);
    if (!rc) rc = grid_for(&probe, (const void*)sw_scan_pkd<kStripRows16>, &cx->grid_drift);
    if (!rc) rc = grid_for(&probe, (const void*)sw_scan_pkc<kStripRows16>, &cx->grid_col);
    return rc;
}

inline uint64_t ws_slot_elems(const Tile& t) { return 2ull * (t.max_len4 ? t.max_len4 : 1) * 64 * 4; }   // int32 per wave slot

// grid of a tile launch: the full persistent grid unless the tile's sequences are so long that the per-wave
// strip-boundary workspace would exceed the budget
inline uint32_t tile_grid(const Tile& t, uint32_t full_grid, uint64_t budget) {
    const uint64_t per_block = ws_slot_elems(t) * 4 * kWavesPerBlock;
    const uint64_t fit = budget / per_block;
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(full_grid, fit));
}

// The workspace budget of ONE call: kWorkspaceBudget, unless a tile of this database is long enough to want more than
// kWorkspaceAlwaysFits at the full grid -- then at most 40 % of what the device has free right now (counting the two workspaces the
// context already holds: they would be freed and re-made), for EACH of the main and the side workspace, so that a scan of a
// long-sequence tile beside a large resident database runs on a smaller grid instead of failing in hipMalloc.  Fixed for the call
// (make_plan stores it in the context under the context's lock): the plan's sizes and the launches' grids agree.
static uint64_t call_workspace_budget(DeviceCtx* cx, const s4g_db* db, uint32_t full_grid) {
    // S4G_WORKSPACE_BUDGET=<bytes>: the tests' way to the smaller grids (read per call: a test sets it for one scan)
    if (const char* e = getenv("S4G_WORKSPACE_BUDGET")) { const unsigned long long v = strtoull(e, nullptr, 10); if (v) return std::max<uint64_t>(v, 1ull << 20); }
    uint64_t want = 0;
    for (const Tile& t : db->tiles) want = std::max<uint64_t>(want, ws_slot_elems(t) * 4 * kWavesPerBlock * full_grid);
    if (want <= kWorkspaceAlwaysFits) return kWorkspaceBudget;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return kWorkspaceBudget; }
    const uint64_t avail = (uint64_t)free_b + cx->ws_bytes + cx->ws_side_bytes;
    return std::max<uint64_t>(1ull << 30, std::min<uint64_t>(kWorkspaceBudget, avail / 5 * 2));
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
    rc = grid_for(db, (const void*)sw_scan_pkc<kStripRows16>, &cx->grid_col);
    if (rc) return rc;
    cx->ws_budget = call_workspace_budget(cx, db, std::max(std::max(cx->grid_i32, cx->grid_i16), std::max(std::max(cx->grid_f16, cx->grid_drift), cx->grid_col)));
    // the per-call flag wins; S4G_SCAN_FRAME only moves the default of calls that do not set it (with the variable winning,
    // every diag_frame=True comparison of the tests would have run the same kernel twice)
    p->col_frame = (flags & S4G_SCAN_DIAG_FRAME) ? false : frame_env() != 1;
    p->grid_i32 = cx->grid_i32; p->grid_i16 = cx->grid_i16; p->grid_m3 = cx->grid_f16; p->grid_drift = p->col_frame ? cx->grid_col : cx->grid_drift;
    const bool gaps16 = sc->open < 32768 && sc->ext < 32768;
    p->f16_arith = (flags & S4G_SCAN_F16_ARITH) != 0;
    // bias of the maximum3-on-bit-patterns tiers: keeps diag+s, h-open, E-ext (and H~(0,-1) = B - ext in the drift frame) >= 0
    p->bias = (uint32_t)std::max(std::max(-kPadScore, -sc->min_score), std::max(sc->open, 2 * sc->ext));
    const uint64_t cap = 0x7BFF;     // largest int16 bit pattern that is a finite f16
    const uint64_t m3_window = p->f16_arith ? 4096 : (p->bias < cap ? cap - p->bias : 0);
    const bool m3_gaps = p->f16_arith ? (sc->open <= 2048 && sc->ext <= 2048) : gaps16;
    const bool no_m3 = (flags & (S4G_SCAN_FORCE_I32 | S4G_SCAN_NO_F16)) != 0;
    const bool no_drift = no_m3 || p->f16_arith || (flags & S4G_SCAN_NO_DRIFT) != 0;
    const bool no_spec = (flags & S4G_SCAN_NO_SPEC) != 0;
    // int32 tier in the drifting frame: s + 2e must fit the int8 profile; the frame itself (rows + columns) * ext and the scores
    // are far inside int32 for anything a database holds (S4G_SCAN_NO_DRIFT selects the fixed-frame kernel, as for the packed tier)
    p->wide_drift = !(flags & S4G_SCAN_NO_DRIFT) && std::max(0, sc->max_score) + 2 * sc->ext <= 127 && sc->ext <= 4096 && sc->open <= (1 << 20);
    const uint64_t step = (uint64_t)std::max(0, sc->max_score) + 2 * (uint64_t)sc->ext;      // the most one cell adds to a stored value
    if (p->wide_drift) {
        // ... and every stored value, score + (rows + columns) * ext, must stay inside int32 for the longest query against the longest
        // tile of THIS call (half a million residues at ext = 4096): otherwise the fixed-frame kernel
        uint64_t max_cols = 0, max_bound = 0;
        for (const Tile& t : db->tiles) max_cols = std::max<uint64_t>(max_cols, (uint64_t)t.max_len4 * 4 + 8);
        uint32_t max_rows = 0;
        for (uint32_t i = 0; i < q_count; ++i) { max_bound = std::max<uint64_t>(max_bound, q->score_bound[q_begin + i]); max_rows = std::max(max_rows, q->len[q_begin + i]); }
        if (((uint64_t)max_rows + 2 * kStripRows + max_cols) * (uint64_t)sc->ext + max_bound + step >= (1ull << 31)) p->wide_drift = false;
    }
    p->spec_base = (uint32_t)(cap > p->bias + step ? cap - p->bias - step : 0);
    // column frame: a cell adds at most max score + ext to a stored value, the running maximum ext per column; the kernel compares
    // IN the frame (sw_scan_pkc), so the bias is not part of the threshold
    const uint64_t step_c = (uint64_t)std::max(0, sc->max_score) + (uint64_t)sc->ext;
    if (p->col_frame) p->spec_base = (uint32_t)(cap > step_c ? cap - step_c : 0);

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
            const uint64_t rows = p->col_frame ? 0 : (uint64_t)q->len[id] + 2 * kStripRows16;   // anti-diagonal frame: padded rows of the pair's strips
            const uint64_t fstep = p->col_frame ? step_c : step;
            int tier;
            if (!no_drift && gaps16 && b + p->bias + (rows + cols) * (uint64_t)sc->ext <= cap) tier = 0;
            else if (!no_drift && !no_spec && gaps16 && p->bias + fstep + (rows + cols) * (uint64_t)sc->ext + kSpecSpan <= cap) {
                tier = 0;               // the frame fits with room to spare: run there, recheck what comes near the end
                p->speculative = true;
                p->n_spec++;
            }
            else if (!no_m3 && m3_gaps && b <= m3_window) tier = 1;
            else if (!no_m3 && !no_spec && !p->f16_arith && m3_gaps && p->bias + step + kSpecSpan <= cap) {
                tier = 1;               // the frame does not fit the drift window: same scheme in the fixed-frame maximum3 tier
                p->speculative = true;
                p->n_spec++;
            }
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
            const double waves = (double)tile_grid(t, full, cx->ws_budget) * kWavesPerBlock;
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
        // critical-path items: measured against a quarter of the per-wave share of their launch (sw_kernels.hip.h, item_priority: three
        // priority levels a factor of four apart; only items of that order can let unfair issue arbitration between co-resident waves
        // stretch the launch)
        for (int tier = 0; tier < 4; ++tier) {
            if (ids[tier].empty()) continue;
            double rows_sum = 0;
            for (uint32_t id : ids[tier]) rows_sum += q->len[id];
            const double waves = (double)tile_grid(t, full, cx->ws_budget) * kWavesPerBlock;
            const double share = (tier < 3 ? 0.5 : 1.0) * rows_sum * ((double)t.padded_residues / 256.0) / waves;       // (rows * len4) per wave; a packed item is two queries
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
        const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>(tile_grid(t, full, cx->ws_budget), ((uint64_t)t.n_groups * most + kWavesPerBlock - 1) / kWavesPerBlock));
        p->ws_bytes = std::max<uint64_t>(p->ws_bytes, ws_slot_elems(t) * 4 * kWavesPerBlock * blocks);
    }
    return S4G_OK;
}

int ensure_scratch(const s4g_db* db, const s4g_scorer* sc, const Plan& p, hipStream_t stream) {
    DeviceCtx* cx = device_ctx(db->device);
    if (cx->ws_bytes < p.ws_bytes) {
        if (cx->d_ws) HIP_TRY(hipFree(cx->d_ws));
        cx->d_ws = nullptr; cx->ws_bytes = 0;
        if (hipMalloc((void**)&cx->d_ws, p.ws_bytes) != hipSuccess) {
            (void)hipGetLastError();
            return fail(S4G_ERR_NOMEM, "cannot allocate %llu bytes of strip-boundary workspace (budget of this call %llu)", (unsigned long long)p.ws_bytes, (unsigned long long)cx->ws_budget);
        }
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
    { const int rcm = ensure_mat(cx, sc, stream); if (rcm) return rcm; }
    if (need) HIP_TRY(hipMemcpyAsync(cx->d_qlist, p.flat.data(), need * 4, hipMemcpyHostToDevice, stream));
    return S4G_OK;
}

// ---- launches of few, long items ----
// The work queue hands out items group-major (longest database groups first, every query pair of a group in turn): with hundreds of
// items per wave that is as good as any order.  64 queries of 2000-5000 residues against 100 000 targets are 12 items per wave of up to
// half a second each, and the order matters: in the anti-diagonal of (group, pair) costs the queue reaches items of 100+ ms when most
// waves have 100 ms left -- the blocks were busy 0.94 of such a launch (s4g_scan_clock: mean / longest block).  Below kOrderItemsPerWave
// items per wave the items are dealt out by cost instead, rows x columns, largest first (one small sort on the device per launch).
constexpr uint64_t kOrderItemsPerWave = 48;
// (pairs: [2 n_pairs] query ids of the packed tiers, or -- singles -- [n_pairs] query ids of the int32 tier, NULL there = q_begin + index)
__global__ void scan_item_cost(uint32_t n_groups, uint32_t n_pairs, const uint32_t* __restrict__ grp_len4, const uint32_t* __restrict__ pairs,
                               bool singles, uint32_t q_begin, const uint32_t* __restrict__ q_len, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_groups * n_pairs) return;
    const uint32_t g = i / n_pairs, pi = i - g * n_pairs;
    uint32_t rows;
    if (singles) rows = q_len[pairs ? pairs[pi] : q_begin + pi];
    else { const uint32_t qA = pairs[2 * pi], qB = pairs[2 * pi + 1]; rows = std::max(q_len[qA], qB != 0xFFFFFFFFu ? q_len[qB] : 0u); }
    key[i] = (uint64_t)grp_len4[g] * rows;
    val[i] = i;
}
static bool item_order_enabled() {
    static const bool on = []() { const char* e = getenv("S4G_SCAN_ITEM_ORDER"); return !(e && e[0] == '0'); }();
    return on;
}
// -> *out = device array of the launch's items by descending cost, or NULL (many items per wave, or switched off)
static int item_order(DeviceCtx* cx, int which, hipStream_t stream, uint32_t n_groups, uint32_t n_pairs, uint32_t blocks, const uint32_t* grp_len4,
                      const uint32_t* pairs, const uint32_t* q_len, const uint32_t** out, bool singles = false, uint32_t q_begin = 0) {
    *out = nullptr;
    const uint64_t n = (uint64_t)n_groups * n_pairs;
    if (!item_order_enabled() || n < 2 || n >= kOrderItemsPerWave * blocks * kWavesPerBlock) return S4G_OK;
    size_t cub = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, cub, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                                         (uint32_t*)nullptr, (int)n, 0, 48, stream));
    const auto al = [](uint64_t v) { return (v + 255) & ~255ull; };
    const uint64_t need = 2 * al(n * 8) + 2 * al(n * 4) + al(cub);
    if (cx->order_cap[which] < need) {
        HIP_TRY(hipDeviceSynchronize());
        if (cx->d_order[which]) (void)hipFree(cx->d_order[which]);
        cx->d_order[which] = nullptr; cx->order_cap[which] = 0;
        if (hipMalloc((void**)&cx->d_order[which], need + need / 4) != hipSuccess) return fail(S4G_ERR_NOMEM, "cannot allocate %llu bytes for the item order", (unsigned long long)need);
        cx->order_cap[which] = need + need / 4;
    }
    char* b = cx->d_order[which];
    uint64_t* key = (uint64_t*)b;                   b += al(n * 8);
    uint64_t* key2 = (uint64_t*)b;                  b += al(n * 8);
    uint32_t* val = (uint32_t*)b;                   b += al(n * 4);
    uint32_t* val2 = (uint32_t*)b;                  b += al(n * 4);
    hipLaunchKernelGGL(scan_item_cost, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n_groups, n_pairs, grp_len4, pairs, singles, q_begin, q_len, key, val);
    HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(b, cub, key, key2, val, val2, (int)n, 0, 48, stream));
    HIP_TRY(hipGetLastError());
    *out = val2;
    return S4G_OK;
}

// One tile: one launch per non-empty tier.  perm/stride select the output layout: original ids into the caller's
// [q][n_seqs] matrix, or tile columns.
int launch_tile(const s4g_db* db, const Tile& t, const TileWork& w, const s4g_queries* q, const s4g_scorer* scorer, const Plan& p,
                uint32_t q_begin, int32_t* d_scores, uint64_t stride, const uint32_t* perm, hipStream_t stream, bool side = false,
                uint32_t g0 = 0) {      // g0: skip the tile's first g0 groups (they are scored as pairs, see split_groups)
    if (t.n_seqs == 0 || g0 >= t.n_groups) return S4G_OK;
    const uint32_t n_groups = t.n_groups - g0;
    DeviceCtx* cx = device_ctx(db->device);
    int32_t* const ws = side ? cx->d_ws_side : cx->d_ws;
    unsigned* const counters = cx->d_counter + (side ? 8 : 0);
    // a persistent grid larger than the number of items only launches blocks that exit at once
    auto grid_of = [&](uint32_t full, uint64_t items) {
        return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(tile_grid(t, full, cx->ws_budget), (items + kWavesPerBlock - 1) / kWavesPerBlock));
    };
    const uint64_t most = std::max(std::max<uint64_t>(w.n_drift, w.n_m3), std::max<uint64_t>(w.n_pk, w.n_wide));
    if ((uint64_t)n_groups * most >= 0xFFFFFFFFull)
        return fail(S4G_ERR_INVALID, "too many work items in one tile launch; split the query range");
    if ((w.n_drift || (w.n_m3 && !p.f16_arith))) { const int rc3 = ensure_max3(cx); if (rc3) return rc3; }
    HIP_TRY(hipMemsetAsync(counters, 0, 16, stream));
    // clock samples (s4g_scan_clock): a zeroed region of 4 counters per block for each of the call's first launches
    auto clk_region = [&](uint32_t blocks, int tier) -> unsigned long long* {
        if (cx->clk_launches.size() >= kClkLaunches) return nullptr;
        const uint64_t off = cx->clk_launches.empty() ? 0 : cx->clk_launches.back().off + 4ull * cx->clk_launches.back().blocks;
        if (off + 4ull * blocks > cx->clk_cap) return nullptr;
        if (hipMemsetAsync(cx->d_clk + off, 0, 32ull * blocks, stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        cx->clk_launches.push_back({off, blocks, tier});
        return cx->d_clk + off;
    };
    ScanArgs16 a;
    a.db_res = t.d_res; a.grp_off = t.d_grp_off + g0; a.grp_len4 = t.d_grp_len4 + g0; a.grp_perm = perm + (size_t)g0 * 64;
    a.q_codes = q->d_codes; a.q_off = q->d_off; a.q_len = q->d_len;
    a.mat = cx->d_mat;
    a.scores = d_scores; a.score_stride = stride;
    a.ws = ws; a.ws_slot = ws_slot_elems(t);
    a.n_groups = n_groups; a.q_begin = q_begin;
    a.open = (uint32_t)scorer->open; a.ext = (uint32_t)scorer->ext;
    a.spec_base = 0;
    a.order = nullptr;
    if (w.n_drift) {
        a.pairs = cx->d_qlist + w.off_drift; a.n_pairs = w.n_drift; a.counter = counters + 3; a.prio_cost = w.prio_cost[0];
        a.floor_bits = p.bias | (p.bias << 16);
        a.spec_base = p.speculative ? p.spec_base : 0;
        const uint32_t gd = grid_of(p.grid_drift, (uint64_t)n_groups * w.n_drift);
        { const uint32_t* ord = nullptr; const int rco = item_order(cx, side ? 1 : 0, stream, n_groups, a.n_pairs, gd, a.grp_len4, a.pairs, a.q_len, &ord); if (rco) return rco; a.order = ord; }
        a.clk = clk_region(gd, 0);
        if (p.col_frame) hipLaunchKernelGGL((sw_scan_pkc<kStripRows16>), dim3(gd), dim3(kBlock), 0, stream, a);
        else hipLaunchKernelGGL((sw_scan_pkd<kStripRows16>), dim3(gd), dim3(kBlock), 0, stream, a);
        HIP_TRY(hipGetLastError());
    }
    if (w.n_m3) {
        a.pairs = cx->d_qlist + w.off_m3; a.n_pairs = w.n_m3; a.counter = counters + 2; a.prio_cost = w.prio_cost[1];
        const uint32_t g3 = grid_of(p.grid_m3, (uint64_t)n_groups * w.n_m3);
        { const uint32_t* ord = nullptr; const int rco = item_order(cx, side ? 1 : 0, stream, n_groups, a.n_pairs, g3, a.grp_len4, a.pairs, a.q_len, &ord); if (rco) return rco; a.order = ord; }
        a.clk = clk_region(g3, 1);
        if (p.f16_arith) {
            a.floor_bits = 0xE800E800u;      // (-2048.0h, -2048.0h)
            hipLaunchKernelGGL((sw_scan_pk<ArithF16, kStripRows16>), dim3(g3), dim3(kBlock), 0, stream, a);
        } else {
            a.floor_bits = p.bias | (p.bias << 16);
            a.spec_base = p.speculative ? (uint32_t)(0x7BFF - std::max(0, scorer->max_score)) : 0;
            hipLaunchKernelGGL((sw_scan_pk<ArithI16M3, kStripRows16>), dim3(g3), dim3(kBlock), 0, stream, a);
        }
        HIP_TRY(hipGetLastError());
    }
    a.spec_base = 0;
    if (w.n_pk) {
        a.pairs = cx->d_qlist + w.off_pk; a.n_pairs = w.n_pk; a.counter = counters; a.prio_cost = w.prio_cost[2];
        a.floor_bits = 0;
        const uint32_t g2 = grid_of(p.grid_i16, (uint64_t)n_groups * w.n_pk);
        { const uint32_t* ord = nullptr; const int rco = item_order(cx, side ? 1 : 0, stream, n_groups, a.n_pairs, g2, a.grp_len4, a.pairs, a.q_len, &ord); if (rco) return rco; a.order = ord; }
        a.clk = clk_region(g2, 2);
        hipLaunchKernelGGL((sw_scan_pk<ArithI16, kStripRows16>), dim3(g2), dim3(kBlock), 0, stream, a);
        HIP_TRY(hipGetLastError());
    }
    if (w.n_wide) {
        ScanArgs b;
        b.db_res = t.d_res; b.grp_off = t.d_grp_off + g0; b.grp_len4 = t.d_grp_len4 + g0; b.grp_perm = perm + (size_t)g0 * 64;
        b.q_codes = q->d_codes; b.q_off = q->d_off; b.q_len = q->d_len;
        b.mat = cx->d_mat;
        b.scores = d_scores; b.score_stride = stride;
        b.ws = ws; b.ws_slot = ws_slot_elems(t);
        b.counter = counters + 1;
        b.n_groups = n_groups; b.q_begin = q_begin; b.q_count = w.n_wide;
        b.q_list = cx->d_qlist + w.off_wide;
        b.open = (uint32_t)scorer->open; b.ext = (uint32_t)scorer->ext;
        const uint32_t gw = grid_of(p.grid_i32, (uint64_t)n_groups * w.n_wide);
        b.prio_cost = w.prio_cost[3];
        b.order = nullptr;
        { const uint32_t* ord = nullptr; const int rco = item_order(cx, side ? 1 : 0, stream, n_groups, b.q_count, gw, b.grp_len4, b.q_list, b.q_len, &ord, true, q_begin); if (rco) return rco; b.order = ord; }
        b.clk = clk_region(gw, 3);
        // the drifting-frame form (6.5 ops per cell instead of 7.5) wherever its shifted scores fit the int8 strip profile
        if (p.wide_drift) hipLaunchKernelGGL(sw_scan_i32d<kStripRows>, dim3(gw), dim3(kBlock), 0, stream, b);
        else hipLaunchKernelGGL(sw_scan_i32<kStripRows>, dim3(gw), dim3(kBlock), 0, stream, b);
        HIP_TRY(hipGetLastError());
    }
    return S4G_OK;
}

// The scratch buffers (workspace, work-queue counter, query lists) are shared by all launches on this database:
// a launch on another stream must not start before the previous one has finished with them.
int ensure_mat(DeviceCtx* cx, const s4g_scorer* scorer, hipStream_t stream) {
    if (!cx->d_mat) HIP_TRY(hipMalloc((void**)&cx->d_mat, 1024));
    if (!cx->mat_ready) HIP_TRY(hipEventCreateWithFlags(&cx->mat_ready, hipEventDisableTiming));
    if (cx->mat_serial == scorer->serial && scorer->serial != 0) return hipStreamWaitEvent(stream, cx->mat_ready, 0) == hipSuccess ? S4G_OK : fail(S4G_ERR_HIP, "hipStreamWaitEvent failed");
    if (cx->trace_last) HIP_TRY(hipStreamWaitEvent(stream, cx->trace_last, 0));      // a traceback in flight on its own stream still reads the old table
    HIP_TRY(hipMemcpyAsync(cx->d_mat, scorer->mat32, 1024, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(cx->mat_ready, stream));
    cx->mat_serial = scorer->serial;
    return S4G_OK;
}

// Load-time guard of that argument (ensure_max3 below, before a device's first packed launch): maximum3 on bit patterns of
// the whole window -- f16 denormals (1 .. 0x3FF: with a bias of 64 most stored values are), normals, the window's top 0x7BFF --
// must return the integer maximum bit for bit.  A device or build that flushes f16 denormals fails here, loudly, not in the scores.
__global__ void denorm_selftest(unsigned* __restrict__ bad) {
    const unsigned lane = threadIdx.x;
    unsigned mism = 0;
    for (unsigned round = 0; round < 16; ++round) {
        const unsigned a = (lane * 37u + round * 61u + 1u) & 0x3FFu;                  // denormal patterns
        const unsigned b = 0x0400u + ((lane * 257u + round * 1031u) % 0x77FFu);       // 0x0400 .. 0x7BFE
        const unsigned c = (lane + round) & 1u ? 0x7BFFu - lane : lane;               // the top of the window / next to zero
        const unsigned x = a | (b << 16), y = b | (c << 16), z = c | (a << 16);
        const unsigned r = (unsigned)pk_maximum3((int)x, (int)y, (int)z);
        const unsigned lo = max(max(a, b), c), hi = max(max(b, c), a);
        if (r != (lo | (hi << 16))) mism = 1;
        const unsigned r2 = (unsigned)pk_maximum3((int)a, (int)a, (int)(1u | (1u << 16)));   // two denormals against the smallest one
        if (r2 != (max(a, 1u) | (1u << 16))) mism = 1;
    }
    if (mism) atomicOr(bad, 1u);
}

int ensure_max3(DeviceCtx* cx) {
    if (cx->max3_state > 0) return S4G_OK;
    if (cx->max3_state == 0) {
        if (!cx->d_counter) HIP_TRY(hipMalloc((void**)&cx->d_counter, 256));
        unsigned* flag = cx->d_counter + 48;
        unsigned h = 0;
        HIP_TRY(hipMemcpy(flag, &h, 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(denorm_selftest, dim3(1), dim3(64), 0, nullptr, flag);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(&h, flag, 4, hipMemcpyDeviceToHost));
        cx->max3_state = h ? -1 : 1;
    }
    if (cx->max3_state < 0)
        return fail(S4G_ERR_HIP, "v_pk_maximum3_f16 does not order integer bit patterns on device %d: f16 denormals are flushed (the library must be built "
                                 "with FLOAT_DENORM_MODE_16_64 = 3, hipcc's default; tools/check_denorm.py checks a build).  S4G_SCAN_NO_F16 selects the "
                                 "plain packed int16 kernels, which do not depend on it.", cx->device);
    return S4G_OK;
}

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

// s4g_scan_stats: what the call did (host part; the recheck pass counts on the device)
static int stats_begin(DeviceCtx* cx, hipStream_t stream) {
    if (!cx->d_counter) HIP_TRY(hipMalloc((void**)&cx->d_counter, 256));
    HIP_TRY(hipMemsetAsync(cx->d_counter + kStatRechecked, 0, 4, stream));
    for (auto& v : cx->stat_tier) v = 0;
    cx->stat_spec_queries = cx->stat_pair_items = cx->stat_list_pairs = 0;
    cx->clk_launches.clear();
    if (!cx->d_clk) {
        int rc = device_props(cx);
        if (rc) return rc;
        const uint64_t cap = 4ull * kClkLaunches * (uint64_t)cx->prop.multiProcessorCount * 8;      // grids are <= 8 blocks per CU
        HIP_TRY(hipMalloc((void**)&cx->d_clk, cap * 8));
        cx->clk_cap = cap;
    }
    return S4G_OK;
}
static void stats_add(DeviceCtx* cx, const Plan& p) {
    for (int k = 0; k < 4; ++k) cx->stat_tier[k] += p.n_by_tier[k];
    cx->stat_spec_queries += p.n_spec;
}

S4G_API int s4g_scan_stats(int device, uint64_t* out, uint32_t cap) {
    if (!out || cap < 8) return fail(S4G_ERR_INVALID, "need room for 8 counters");
    if (device < 0 || device >= 64) return fail(S4G_ERR_INVALID, "bad device %d", device);
    DeviceGuard guard(device);
    DeviceCtx* cx = device_ctx(device);
    std::lock_guard<std::mutex> lk(cx->mu);
    unsigned rechecked = 0;
    if (cx->d_counter) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(&rechecked, cx->d_counter + kStatRechecked, 4, hipMemcpyDeviceToHost));
    }
    for (int k = 0; k < 4; ++k) out[k] = cx->stat_tier[k];
    out[4] = cx->stat_spec_queries; out[5] = rechecked; out[6] = cx->stat_pair_items; out[7] = cx->stat_list_pairs;
    if (cap >= 10) { out[8] = cx->stat_coop_scan; out[9] = cx->stat_coop_trace; }
    if (cap >= 11) out[10] = cx->stat_pair_pk;
    return S4G_OK;
}

// s4g_scan_clock: the shader clock and the duration of the scan launches of the last call, from the samples the kernels took
// themselves (clk_enter / clk_leave): per block d(s_memtime) / d(s_memrealtime) x 100 MHz.
S4G_API int s4g_scan_clock(int device, double* out, uint32_t cap) {
    if (!out || cap < 8) return fail(S4G_ERR_INVALID, "need room for 8 values");
    if (device < 0 || device >= 64) return fail(S4G_ERR_INVALID, "bad device %d", device);
    DeviceGuard guard(device);
    DeviceCtx* cx = device_ctx(device);
    std::lock_guard<std::mutex> lk(cx->mu);
    for (uint32_t i = 0; i < std::min(cap, 10u); ++i) out[i] = 0.0;
    if (!cx->d_clk || cx->clk_launches.empty()) return S4G_OK;
    HIP_TRY(hipDeviceSynchronize());
    out[0] = (double)cx->clk_launches.size();
    std::vector<unsigned long long> h;
    double best_ms = -1.0;
    for (const auto& L : cx->clk_launches) {
        h.resize(4ull * L.blocks);
        HIP_TRY(hipMemcpy(h.data(), cx->d_clk + L.off, h.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0, lo = 1e30, hi = 0, ms = 0, shader_ticks = 0, ms_sum = 0, ms_min = 1e30;
        uint32_t used = 0;
        for (uint32_t b = 0; b < L.blocks; ++b) {
            const unsigned long long t0 = h[4 * b], r0 = h[4 * b + 1], t1 = h[4 * b + 2], r1 = h[4 * b + 3];
            if (r1 <= r0 || t1 <= t0 || r1 - r0 < 1000) continue;          // a block that found the queue empty: < 10 us
            const double mhz = (double)(t1 - t0) / (double)(r1 - r0) * 100.0;       // s_memrealtime counts at 100 MHz
            sum += mhz; lo = std::min(lo, mhz); hi = std::max(hi, mhz); ++used;
            ms = std::max(ms, (double)(r1 - r0) * 1e-5); ms_sum += (double)(r1 - r0) * 1e-5; ms_min = std::min(ms_min, (double)(r1 - r0) * 1e-5);
            shader_ticks = std::max(shader_ticks, (double)(t1 - t0));
        }
        if (!used || ms <= best_ms) continue;
        best_ms = ms;                                                   // report the longest launch of the call
        out[1] = sum / used; out[2] = lo; out[3] = hi; out[4] = used; out[5] = ms; out[6] = L.tier; out[7] = shader_ticks;
        if (cap >= 10) { out[8] = ms_sum / used; out[9] = ms_min; }      // how long the blocks were busy: mean / longest = what the drain of the queue cost
    }
    return S4G_OK;
}

// the largest query count the block-pipelined route takes by default (S4G_SCAN_PIPE_MAX_QUERIES overrides; 0 turns the route off)
static uint32_t pipe_max_queries() {
    static const uint32_t v = []() { const char* e = getenv("S4G_SCAN_PIPE_MAX_QUERIES"); return e ? (uint32_t)std::max(0, atoi(e)) : 16u; }();
    return v;
}

int check_scan_args(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t q_begin, uint32_t q_count) {
    if (!db || !q || !scorer) return fail(S4G_ERR_INVALID, "NULL argument");
    if (db->device != q->device) return fail(S4G_ERR_INVALID, "db on device %d, queries on device %d", db->device, q->device);
    if ((uint64_t)q_begin + q_count > q->n) return fail(S4G_ERR_INVALID, "query range out of bounds");
    if (db->cached) return fail(S4G_ERR_INVALID, "a database of cached chains (s4g_db_create_cached) serves s4g_score_pairs_host / s4g_score_lists_host / s4g_traceback_host only");
    return S4G_OK;
}

S4G_API int s4g_scan_plan(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, char* kernel_name,
                          size_t kernel_name_cap, uint32_t* grid_blocks, uint32_t* block_threads,
                          uint32_t* strip_rows, uint64_t* workspace_bytes) {
    if (!db || !q || !scorer) return fail(S4G_ERR_INVALID, "NULL argument");
    if (db->cached) return fail(S4G_ERR_INVALID, "a database of cached chains has no scan plan");
    DeviceGuard guard(db->device);
    DeviceCtx* cx = device_ctx(db->device);
    std::lock_guard<std::mutex> lk(cx->mu);
    if (!db->stream && scan_prefers_pairs(db, q, 0, q->n, 0)) {        // a call this small takes the latency route (one wave per pair)
        int rc0 = device_props(cx);
        if (rc0) return rc0;
        if (kernel_name && kernel_name_cap) snprintf(kernel_name, kernel_name_cap, "%s", q->max_len > (uint32_t)kPairMaxRows ? "sw_pair_scan_band<16>" : "sw_pair_scan<RPL>");
        if (grid_blocks) *grid_blocks = (uint32_t)std::min<uint64_t>((uint64_t)q->n * db->n_seqs, (uint64_t)cx->prop.multiProcessorCount * 8);
        if (block_threads) *block_threads = 64;
        if (strip_rows) *strip_rows = 0;
        if (workspace_bytes) *workspace_bytes = 0;
        return S4G_OK;
    }
    ensure_bounds(q, scorer);
    Plan p;
    int rc = make_plan(db, q, 0, q->n, scorer, 0, &p);
    if (rc) return rc;
    // report the kernel that carries most of the queries
    int top = 0;
    for (int k = 1; k < 4; ++k) if (p.n_by_tier[k] > p.n_by_tier[top]) top = k;
    const char* names[4] = {p.col_frame ? "sw_scan_pkc<32>" : "sw_scan_pkd<32>", p.f16_arith ? "sw_scan_pk<ArithF16,32>" : "sw_scan_pk<ArithI16M3,32>", "sw_scan_pk<ArithI16,32>",
                            p.wide_drift ? "sw_scan_i32d<32>" : "sw_scan_i32<32>"};
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
    if (db->stream) return fail(S4G_ERR_INVALID, "a streamed database (s4g_db_open_streamed) serves s4g_scan_topk / s4g_scan_topk_host only");
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(db->device);
    DeviceCtx* cx = device_ctx(db->device);
    std::lock_guard<std::mutex> lk(cx->mu);
    // Routing per tile (split_groups): the first g0 groups of a tile -- its longest sequences -- go to one wave per pair,
    // the rest to the scan kernels.  g0 = 0 for a throughput-bound call (bench.py), g0 = all groups for a latency-bound
    // one (few queries against a small database: the per-query alignDatabase pattern) and for a tile of titin-sized
    // targets; something in between when only the longest groups would make the launch wait.
    std::vector<uint32_t> g0s(db->tiles.size(), 0);
    bool all_pairs = true, any_pairs = false;
    const bool pairs_possible = (uint64_t)q_count * db->n_seqs <= 0xFFFFFFFFull;      // 32-bit score indices in the pair kernels
    for (size_t ti = 0; ti < db->tiles.size(); ++ti) {
        const Tile& t = db->tiles[ti];
        if (pairs_possible) g0s[ti] = split_groups(db, t, q, q_begin, q_count, flags);
        all_pairs = all_pairs && (t.n_seqs == 0 || g0s[ti] >= t.n_groups);
        any_pairs = any_pairs || g0s[ti] != 0;
    }
    ScratchScope scope(db, stream);
    // A few queries against a big database: the block-cooperative strip pipeline (s4g_lists.hip, scan_tile_pipelined) -- no
    // launch tail behind a lone wave that walks the longest group.  Taken when the call asks for it, or by default for 3 to
    // pipe_max_queries() queries when the call is not small enough for the all-pairs route below.  Measured, 100 000 targets of
    // the bench distribution, ms per call, pipeline | split model (profiles/r03_scan_latency.txt):
    //   1 query 5.5 | 2.6-3.7    2: 5.6 | 3.6    3: 5.7 | 6.3    4: 5.7 | 6.5    6: 6.8 | 8.6    8: 8.8 | 10.4    12: 13.1 | 13.5    16: 17.1 | 17.2    32: 34.5 | 33.9
    // (one or two queries: the longest group is 5 ms of four waves whatever else happens, and only the pair waves spread it over CUs)
    // Round 6: ONE or TWO queries take it too when they are long and the database is large -- the launch of the split model waits for the
    // wave that walks the longest group under the longest query, rows * columns of one dependent chain; the pipeline spreads that item's
    // strips over the four waves of a block.  Measured (tools/scan_latency.py shapes, 100 000 targets, ms per call, pipeline | split):
    //   1 query of 700: 10.4 | 11.0   1000: 13.8 | 13.8   1500: 20.7 | 31.5   3000: 40.7 | 61.0   8000: 104 | 162   20 000: 264 | 415
    //   2 queries: the same within 2 ms.  Against 20 000 targets the split (all pairs for one query) wins at every length: 0.8-60 | 3.8-195.
    // ... and whatever their length from 196 608 sequences on: there the call is bound by throughput, and the pipeline's items pair two
    // GROUPS under a query that has no partner (sw_scan_lists<R, 2>), where a scan-kernel item leaves half of every register idle.
    //   1 query of 350 residues, pipeline | split: 150 000 targets 5.1 | 5.1   200 000: 5.3 | 6.3   300 000: 5.4 | 8.1   400 000: 5.6 | 9.9   570 000: 6.8 | 13.8
    // With twelve waves to a block (s4g_lists.hip: the form such a call gets) the length threshold is 500 rows (tools/few_query_scan.py,
    // pipeline | split): 100 000 targets, 1 x 500: 5.3 | 5.8   1 x 600: 6.4 | 10.3   2 x 600: 6.7 | 10.7   1 x 900: 9.7 | 15.2   2 x 1100: 12.0 | 27.6;
    // 70 000 targets, 1 x 600: 6.3 | 9.8   1 x 900: 9.5 | 10.1   1 x 1100: 12.2 | 23.5.
    uint32_t max_rows = 0;
    for (uint32_t i = 0; i < q_count; ++i) max_rows = std::max(max_rows, q->len[q_begin + i]);
    const bool want_pipe = (flags & S4G_SCAN_BLOCK_PIPE) != 0 ||
                           (flags == 0 && !(all_pairs && any_pairs) && pipe_max_queries() != 0 &&
                            ((q_count >= 3 && q_count <= pipe_max_queries() && db->n_seqs >= 256u * 64u) ||
                             (q_count <= 2 && ((max_rows >= 500 && db->n_seqs >= 1024u * 64u) || db->n_seqs >= 3072u * 64u))));
    flags &= ~S4G_SCAN_BLOCK_PIPE;
    std::vector<char> piped(db->tiles.size(), 0);
    uint64_t pipe_pairs = 0;
    bool pipe_spec = false;         // a pipelined tile ran speculatively (round 6): its marks need the recheck pass below
    if (want_pipe) {
        rc = scope.acquire();
        if (rc) return rc;
        rc = stats_begin(cx, stream);
        if (rc) return rc;
        rc = ensure_mat(cx, scorer, stream);
        if (rc) return rc;
        size_t refused = 0;
        for (size_t ti = 0; ti < db->tiles.size(); ++ti) {
            const Tile& t = db->tiles[ti];
            if (!t.n_seqs) { piped[ti] = 1; continue; }
            bool used = false, spec = false;
            rc = scan_tile_pipelined(db, t, q, q_begin, q_count, scorer, d_scores, db->n_seqs, t.d_perm_global, stream, &used,
                                     /*allow_spec=*/!(flags & S4G_SCAN_NO_SPEC), &spec);
            if (rc) return rc;
            if (used) { piped[ti] = 1; g0s[ti] = 0; pipe_pairs += (uint64_t)q_count * t.n_seqs; pipe_spec = pipe_spec || spec; } else ++refused;
        }
        cx->stat_list_pairs = pipe_pairs;
        if (pipe_spec) cx->stat_spec_queries += q_count;
        if (!refused) {
            if (pipe_spec) {        // exact int32 pass over the entries the speculative pipeline launches marked
                uint32_t max_lt = 0;
                for (const Tile& t : db->tiles) max_lt = std::max(max_lt, t.max_len4 * 4);
                rc = recheck_marked(db, q, scorer, q_begin, q_count, d_scores, db->n_seqs, db->n_seqs, nullptr, max_lt, stream);
                if (rc) return rc;
            }
            return scope.release();
        }
        // Tiles the pipeline does not take (a query beyond the drifting-frame window there, a tile of titin-sized sequences) go
        // through the general route below, behind what was launched above on the same stream; the tiles that are done are skipped.
        rc = scope.release();
        if (rc) return rc;
        all_pairs = true; any_pairs = false;
        for (size_t ti = 0; ti < db->tiles.size(); ++ti) {
            if (piped[ti]) continue;
            all_pairs = all_pairs && g0s[ti] >= db->tiles[ti].n_groups;
            any_pairs = any_pairs || g0s[ti] != 0;
        }
    }
    if (all_pairs && any_pairs && pipe_pairs == 0) {       // nothing for the scan kernels: one submission, no plan, no strip workspace
        rc = scope.acquire();
        if (rc) return rc;
        rc = stats_begin(cx, stream);
        if (rc) return rc;
        cx->stat_pair_items = (uint64_t)q_count * db->n_seqs;
        rc = scope.release();
        if (rc) return rc;
        bool done = false;
        return scan_as_pairs(db, q, q_begin, q_count, scorer, d_scores, stream, &done);
    }
    ensure_bounds(q, scorer);
    Plan p;
    rc = make_plan(db, q, q_begin, q_count, scorer, flags, &p);
    if (rc) return rc;
    rc = scope.acquire();
    if (rc) return rc;
    rc = stats_begin(cx, stream);
    if (rc) return rc;
    stats_add(cx, p);
    cx->stat_list_pairs += pipe_pairs;
    rc = ensure_scratch(db, scorer, p, stream);
    if (rc) return rc;
    // The side stream carries (a) the pair parts and (b) the scan launches of long-sequence tiles (own workspace): their
    // few huge items then run underneath the main tiles' launches instead of as a tail-dominated launch of their own.
    const uint32_t full = std::max(std::max(p.grid_i32, p.grid_i16), std::max(p.grid_m3, p.grid_drift));
    std::vector<char> on_side(db->tiles.size(), 0);
    uint64_t need = 0;
    if (db->tiles.size() > 1)
        for (size_t ti = 0; ti < db->tiles.size(); ++ti) {
            const Tile& t = db->tiles[ti];
            if (piped[ti] || (uint64_t)t.max_len4 * 4 <= kLongSeq || g0s[ti] >= t.n_groups) continue;
            on_side[ti] = 1;
            const TileWork& w = p.tiles[ti];
            const uint64_t most = std::max(std::max<uint64_t>(w.n_drift, w.n_m3), std::max<uint64_t>(w.n_pk, w.n_wide));
            const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>(tile_grid(t, full, cx->ws_budget), ((uint64_t)t.n_groups * most + kWavesPerBlock - 1) / kWavesPerBlock));
            need = std::max<uint64_t>(need, ws_slot_elems(t) * 4 * kWavesPerBlock * blocks);
        }
    const bool forked = need != 0 || any_pairs;
    if (forked) {
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
        scope.forked = true;
        for (size_t ti = 0; ti < db->tiles.size(); ++ti) {
            const Tile& t = db->tiles[ti];
            if (piped[ti]) continue;
            if (g0s[ti]) {
                cx->stat_pair_items += (uint64_t)q_count * std::min<uint64_t>(t.n_seqs, (uint64_t)g0s[ti] * 64);
                rc = tile_as_pairs(db, t, q, q_begin, q_count, scorer, d_scores, db->n_seqs, false, cx->side_stream,
                                   (uint32_t)std::min<uint64_t>(t.n_seqs, (uint64_t)g0s[ti] * 64));
                if (rc) return rc;
            }
            if (on_side[ti]) {
                rc = launch_tile(db, t, p.tiles[ti], q, scorer, p, q_begin, d_scores, db->n_seqs, t.d_perm_global, cx->side_stream, true, g0s[ti]);
                if (rc) return rc;
            }
        }
        HIP_TRY(hipEventRecord(cx->side_join, cx->side_stream));
    }
    for (size_t ti = 0; ti < db->tiles.size(); ++ti) {
        if (on_side[ti] || piped[ti]) continue;
        const Tile& t = db->tiles[ti];
        rc = launch_tile(db, t, p.tiles[ti], q, scorer, p, q_begin, d_scores, db->n_seqs, t.d_perm_global, stream, false, g0s[ti]);
        if (rc) return rc;
    }
    if (forked) HIP_TRY(hipStreamWaitEvent(stream, cx->side_join, 0));
    if (p.speculative || pipe_spec) {        // exact int32 pass over the entries the speculative launches marked
        uint32_t max_lt = 0;
        for (const Tile& t : db->tiles) max_lt = std::max(max_lt, t.max_len4 * 4);
        rc = recheck_marked(db, q, scorer, q_begin, q_count, d_scores, db->n_seqs, db->n_seqs, nullptr, max_lt, stream);
        if (rc) return rc;
    }
    return scope.release();
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
// top-k (steps 2-3 of alignDatabase on the device): one 256-thread block per query row.
//   1. radix select on the int32 score (high 12 bits of a 24-bit score, then low 12 bits; rows holding scores beyond
//      2^24 - 1 shift the digits up and resolve the remaining low bits in a third level) finds the cut score exactly:
//      the k-th largest score >= min_score;
//   2. everything above the cut is kept; among the ties AT the cut the smallest target ids are kept (ordered
//      block scan), so the result is the exact, deterministic top-k under (score desc, id asc);
//   3. bitonic sort of the <= k survivors in LDS on the 64-bit key (~score << 32 | id).

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

// Grid (rows, segments): block (r, s) works on columns [s * seg_len, (s + 1) * seg_len) of row r and writes list number
// s * rows + r.  With few rows one block per row leaves the chip idle while it reads a million scores four times over;
// the caller then cuts the rows into segments and merges the segment lists (merge_rows keys on (score, id), so the
// result is the same exact top-k).
//
// Round 5, the FAST PATH in front of the general one: k is a few hundred of 10^5 scores, and almost every score is small.  Pass A
// (16-byte loads) counts the row into 63 bins of four scores (0 .. 251) + one bin for everything beyond, in a histogram that is
// PRIVATE to each thread (LDS word [bin][thread]: no atomic ever meets another) -- the general path's first radix level put every
// score below 4096 into ONE bin, 10^5 LDS atomics on one address per row.  The bins give t0, the lower edge of the bin that holds the
// k-th largest score, and how many scores are >= t0; when those fit the block's LDS (8192 keys) pass B collects exactly them and a
// bitonic sort on (score desc, id asc) yields the exact top-k: two reads of the row instead of four, no ordered tie scan.  A row
// whose cut lies beyond 251, or with more than 8192 scores at or above its cut bin, takes the general path below, unchanged.
constexpr int kFastBins = 64, kFastShift = 2, kFastCap = 8192;
// topk_rows holds kFastCap 8-byte keys (64 KB) PLUS ~1.1 KB of small arrays in static LDS: more than the 64 KB a workgroup gets on
// gfx90a / gfx942.  gfx950 has 160 KB per CU and the Makefile refuses any other target; two blocks per CU is all the top-k needs.
static_assert(kFastCap * 8 + 2048 <= 160 * 1024 / 2, "topk_rows: two blocks per CU must fit gfx950's 160 KB of LDS");

__global__ __launch_bounds__(kTopkBlock) void topk_rows(const int32_t* __restrict__ scores, uint32_t n_cols, uint32_t seg_len,
                                                        uint32_t k, uint32_t kpad,
                                                        int32_t min_score, uint32_t* __restrict__ out_ids,
                                                        int32_t* __restrict__ out_scores, uint32_t* __restrict__ out_counts,
                                                        const uint32_t* __restrict__ col_ids,
                                                        const int32_t* __restrict__ row_min_scores) {
    __shared__ __attribute__((aligned(16))) unsigned long long buf[kFastCap];       // 64 KB: private histogram | collected keys | the general path's hist + keys
    unsigned* const hist = reinterpret_cast<unsigned*>(buf);                        // general path: 4096 words (16 KB)
    unsigned long long* const keys = buf + kBins / 2;                               // general path: 4096 keys behind them (32 KB)
    __shared__ int s_hi, s_cut, s_max;
    __shared__ unsigned s_above, s_quota, s_nout, s_tie_base, s_wave_cnt[kTopkBlock / 64];
    __shared__ unsigned s_part[4][kFastBins], s_t0, s_take;
    const int tid = threadIdx.x;
    const uint32_t c0 = blockIdx.y * seg_len;
    const uint32_t n = c0 < n_cols ? (n_cols - c0 < seg_len ? n_cols - c0 : seg_len) : 0;
    const int32_t* row = scores + (uint64_t)blockIdx.x * n_cols + c0;
    const uint64_t slot = (uint64_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (row_min_scores) min_score = row_min_scores[blockIdx.x];       // per-query cut (e-value driven)
    if (min_score < 0) min_score = 0;

    // ---- fast path ----
    {
        // every element of the row once: a scalar head up to 16-byte alignment, 16-byte vectors, a scalar tail
        const uint32_t head = n ? std::min<uint32_t>(n, (uint32_t)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(row) & 15u)) & 15u) / 4u)) : 0u;
        const uint32_t nv = (n - head) / 4;
        const v4i* rv = reinterpret_cast<const v4i*>(row + head);
        auto for_each = [&](auto&& f) {
            for (uint32_t i = tid; i < head; i += kTopkBlock) f(row[i], i);
            for (uint32_t j = tid; j < nv; j += kTopkBlock) {
                const v4i x = rv[j];
                const uint32_t i = head + 4 * j;
                f(x[0], i); f(x[1], i + 1); f(x[2], i + 2); f(x[3], i + 3);
            }
            for (uint32_t i = head + 4 * nv + tid; i < n; i += kTopkBlock) f(row[i], i);
        };
        unsigned* const ph = reinterpret_cast<unsigned*>(buf);                      // [kFastBins][kTopkBlock]
        for (int i = tid; i < kFastBins * kTopkBlock; i += kTopkBlock) ph[i] = 0;
        __syncthreads();
        for_each([&](int v, uint32_t) {
            if (v >= min_score) {
                const unsigned b = std::min<unsigned>((unsigned)v >> kFastShift, kFastBins - 1);
                ph[b * kTopkBlock + tid] += 1;                                      // this thread's own word: no atomic needed
            }
        });
        __syncthreads();
        {   // bin totals: thread (part, bin) adds 64 threads' words, then the four parts
            const int bin = tid & (kFastBins - 1), part = tid >> 6;
            unsigned t = 0;
            for (int j = 0; j < 64; ++j) t += ph[bin * kTopkBlock + part * 64 + ((j + tid) & 63)];
            s_part[part][bin] = t;
        }
        __syncthreads();
        if (tid == 0) {
            unsigned acc = 0;
            int b = kFastBins - 1;
            for (; b >= 0; --b) {
                const unsigned t = s_part[0][b] + s_part[1][b] + s_part[2][b] + s_part[3][b];
                if (acc + t >= k) { acc += t; break; }
                acc += t;
            }
            // b < 0: fewer than k scores pass min_score -- take them all.  b == the last bin: the cut lies beyond the fine bins.
            s_t0 = b < 0 ? 0u : (unsigned)b << kFastShift;
            s_take = (b == kFastBins - 1 || acc > (unsigned)kFastCap) ? 0xFFFFFFFFu : acc;
            s_nout = 0;
        }
        __syncthreads();
        const unsigned take = s_take;
        if (take != 0xFFFFFFFFu) {
            const int t0 = std::max((int)s_t0, min_score);
            __syncthreads();                                                        // (everybody has read the totals: buf becomes the key list)
            for_each([&](int v, uint32_t i) {
                if (v >= t0) buf[atomicAdd(&s_nout, 1u)] = ((unsigned long long)(unsigned)(0x7FFFFFFF - v) << 32) | i;
            });
            __syncthreads();
            const unsigned m_all = s_nout;                                          // == take
            unsigned pad = 2;
            while (pad < m_all) pad <<= 1;
            for (unsigned i = m_all + tid; i < pad; i += kTopkBlock) buf[i] = ~0ull;
            __syncthreads();
            for (unsigned size = 2; size <= pad; size <<= 1) {
                for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
                    for (unsigned t = tid; t < pad / 2; t += kTopkBlock) {
                        const unsigned lo = 2 * t - (t & (stride - 1));
                        const unsigned hi2 = lo + stride;
                        const bool up = ((lo & size) == 0);
                        const unsigned long long x = buf[lo], y = buf[hi2];
                        if ((x > y) == up) { buf[lo] = y; buf[hi2] = x; }
                    }
                    __syncthreads();
                }
            }
            const unsigned m = m_all < k ? m_all : k;
            for (unsigned i = tid; i < k; i += kTopkBlock) {
                const bool valid = i < m;
                const unsigned long long key = valid ? buf[i] : 0ull;
                out_ids[slot * k + i] = valid ? (col_ids ? col_ids[c0 + (uint32_t)key] : c0 + (uint32_t)key) : 0xFFFFFFFFu;
                out_scores[slot * k + i] = valid ? 0x7FFFFFFF - (int)(key >> 32) : 0;
            }
            if (tid == 0) out_counts[slot] = m;
            return;
        }
        __syncthreads();
    }

    // ---- range of the row: scores beyond 2^24 - 1 (a query of a million residues) shift the radix digits up, and a third
    //      level then resolves the bits shifted out, so the cut is exact over the whole non-negative int32 range ----
    if (tid == 0) s_max = 0;
    __syncthreads();
    {
        int m = 0;
        for (uint32_t i = tid; i < n; i += kTopkBlock) m = max(m, row[i]);
        if (m > kScoreCap) atomicMax(&s_max, m);
    }
    __syncthreads();
    const int shift = s_max > kScoreCap ? (32 - __clz(s_max)) - 24 : 0;       // 0 .. 7

    // ---- level 1: high 12 bits ----
    for (int i = tid; i < kBins; i += kTopkBlock) hist[i] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kTopkBlock) {
        const int v = row[i];
        if (v >= min_score) atomicAdd(&hist[(v >> shift) >> 12], 1u);
    }
    __syncthreads();
    if (tid == 0) { int b; unsigned a; find_bin(hist, k, &b, &a); s_hi = b; s_above = a; }
    __syncthreads();
    const int hi = s_hi;
    if (hi < 0) {
        // fewer than k entries pass min_score: keep them all
        if (tid == 0) { s_cut = min_score; s_quota = 0xFFFFFFFFu; }
    } else {
        // ---- level 2: the next 12 bits inside the high bin ----
        const unsigned above_hi = s_above;
        __syncthreads();
        for (int i = tid; i < kBins; i += kTopkBlock) hist[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += kTopkBlock) {
            const int v = row[i];
            if (v >= min_score && ((v >> shift) >> 12) == hi) atomicAdd(&hist[(v >> shift) & (kBins - 1)], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int b; unsigned a;
            find_bin(hist, k - above_hi, &b, &a);
            s_cut = (hi << 12) | b;                       // cut >> shift, exact
            s_above = above_hi + a;                       // entries strictly above it
            s_quota = k - (above_hi + a);                 // shift == 0: ties at the cut to keep (smallest ids)
        }
        if (shift) {
            // ---- level 3: the `shift` low bits among the entries whose upper 24 bits equal the cut's ----
            __syncthreads();
            const int cut24 = s_cut;
            const unsigned above24 = s_above;
            __syncthreads();
            for (int i = tid; i < kBins; i += kTopkBlock) hist[i] = 0;
            __syncthreads();
            for (uint32_t i = tid; i < n; i += kTopkBlock) {
                const int v = row[i];
                if (v >= min_score && (v >> shift) == cut24) atomicAdd(&hist[v & ((1 << shift) - 1)], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int b; unsigned a;
                find_bin(hist, k - above24, &b, &a);
                s_cut = (cut24 << shift) | b;
                s_quota = k - (above24 + a);
            }
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
        out_ids[slot * k + i] = valid ? (col_ids ? col_ids[c0 + (uint32_t)key] : c0 + (uint32_t)key) : 0xFFFFFFFFu;
        out_scores[slot * k + i] = valid ? 0x7FFFFFFF - (int)(key >> 32) : 0;
    }
    if (tid == 0) out_counts[slot] = m;
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


S4G_API int s4g_topk(const s4g_db* db, const int32_t* d_scores, uint32_t q_count, uint32_t k, int32_t min_score,
                     uint32_t* d_ids, int32_t* d_top_scores, uint32_t* d_counts, void* stream_) {
    if (!db || !d_scores || !d_ids || !d_top_scores || !d_counts) return fail(S4G_ERR_INVALID, "NULL argument");
    if (k == 0 || k > (uint32_t)kTopkMax) return fail(S4G_ERR_INVALID, "k must be in 1..%d (got %u)", kTopkMax, k);
    if (q_count == 0) return S4G_OK;
    DeviceGuard guard(db->device);
    uint32_t kpad = 2;
    while (kpad < k) kpad <<= 1;
    hipLaunchKernelGGL(topk_rows, dim3(q_count), dim3(kTopkBlock), 0, (hipStream_t)stream_, d_scores, db->n_seqs, db->n_seqs, k, kpad,
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
    ScratchScope scope(db, stream);
    rc = scope.acquire();
    if (rc) return rc;
    rc = stats_begin(cx, stream);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(d_ids, 0xFF, (size_t)q_count * k * 4, stream));
    HIP_TRY(hipMemsetAsync(d_top_scores, 0, (size_t)q_count * k * 4, stream));
    HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)q_count * 4, stream));
    if (db->n_seqs == 0) return scope.release();
    ensure_bounds(q, scorer);
    const uint64_t cols = db->max_tile_seqs;
    const uint32_t batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(q_count, kTileScoreBudget / (cols * 4)));
    // candidate lists of one tile: one per (row, segment); few rows are cut into up to 16 segments (<= 256 lists)
    const uint64_t lists_max = std::max<uint64_t>(batch, 256);
    // with several tiles, two sets of score buffers / candidate lists (and two streams): consecutive scan tiles alternate between
    // them, so that a tile's launch tail and its top-k run underneath the next tile's scan; a tile that goes to the pair kernels
    // as a whole (the long-sequence tile of a real database) runs on the second set beside the first scan tile
    const bool two_sets = db->tiles.size() >= 2;
    const uint64_t set_bytes = ((uint64_t)batch * cols * 4 + lists_max * k * 8 + lists_max * 4 + 255) / 256 * 256;
    const uint64_t need = set_bytes * (two_sets ? 2 : 1);
    if (cx->tile_scores_bytes < need) {
        if (cx->d_tile_scores) HIP_TRY(hipFree(cx->d_tile_scores));
        cx->d_tile_scores = nullptr; cx->tile_scores_bytes = 0;
        HIP_TRY(hipMalloc((void**)&cx->d_tile_scores, need));
        cx->tile_scores_bytes = need;
    }
    uint32_t kpad = 2, kpad2;
    while (kpad < k) kpad <<= 1;
    kpad2 = kpad * 2;
    for (uint32_t b = 0; b < q_count; b += batch) {
        const uint32_t cnt = std::min(batch, q_count - b);
        if (b) HIP_TRY(hipStreamSynchronize(stream));   // the next batch rewrites the query lists the previous launches read
        Plan p;
        // (a streamed database has no resident tiles for the exact recheck pass to read: tiers by score bound only)
        rc = make_plan(db, q, q_begin + b, cnt, scorer, db->stream ? (flags | S4G_SCAN_NO_SPEC) : flags, &p);
        if (rc) return rc;
        stats_add(cx, p);
        rc = ensure_scratch(db, scorer, p, stream);
        if (rc) return rc;
        if (db->stream) {
            // ---- tiles come from the file: while tile ti's kernels run, the host reads tile ti + 1 and queues its upload ----
            const auto t_scan0 = std::chrono::steady_clock::now();
            int32_t* buf = cx->d_tile_scores;
            uint32_t* l_ids = reinterpret_cast<uint32_t*>(buf + (uint64_t)batch * cols);
            int32_t* l_sc = reinterpret_cast<int32_t*>(l_ids + lists_max * k);
            uint32_t* l_cnt = reinterpret_cast<uint32_t*>(l_sc + lists_max * k);
            const size_t T = db->tiles.size();
            if (T) { rc = db_stream_load(db, 0); if (rc) return rc; }
            for (size_t ti = 0; ti < T; ++ti) {
                Tile lt;
                rc = db_stream_tile(db, ti, stream, &lt);
                if (rc) return rc;
                if (lt.n_seqs) {
                    rc = launch_tile(db, lt, p.tiles[ti], q, scorer, p, q_begin + b, buf, lt.n_seqs, lt.d_perm_local, stream, false);
                    if (rc) return rc;
                    const uint32_t segs = std::max(1u, std::min(std::min(16u, 256u / std::max(1u, cnt)), (lt.n_seqs + 16383u) / 16384u));
                    const uint32_t seg_len = ((lt.n_seqs + segs - 1) / segs + 255u) / 256u * 256u;
                    hipLaunchKernelGGL(topk_rows, dim3(cnt, segs), dim3(kTopkBlock), 0, stream, buf, lt.n_seqs, seg_len, k, kpad, min_score, l_ids,
                                       l_sc, l_cnt, (const uint32_t*)lt.d_col_ids, d_min_scores ? d_min_scores + b : (const int32_t*)nullptr);
                    for (uint32_t sgm = 0; sgm < segs; ++sgm)
                        hipLaunchKernelGGL(merge_rows, dim3(cnt), dim3(kTopkBlock), 0, stream, d_ids + (uint64_t)b * k,
                                           d_top_scores + (uint64_t)b * k, d_counts + b, l_ids + (uint64_t)sgm * cnt * k,
                                           l_sc + (uint64_t)sgm * cnt * k, l_cnt + (uint64_t)sgm * cnt, k, kpad2);
                    HIP_TRY(hipGetLastError());
                }
                rc = db_stream_consumed(db, ti, stream);
                if (rc) return rc;
                if (ti + 1 < T) { rc = db_stream_load(db, ti + 1); if (rc) return rc; }
            }
            HIP_TRY(hipStreamSynchronize(stream));           // the slots are rewritten by the next batch (and the timing below means the scan)
            db_stream_note_scan(db, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_scan0).count());
            continue;
        }
        // Kinds of tile: entirely on the scan kernels (g0 = 0), entirely on the pair kernels (the long-sequence tile of a real
        // database), or split (its longest groups as pairs on the side stream, the rest on the scan kernels).  With two sets:
        //   * a pair tile runs on the side stream with the second set, beside the scan tiles;
        //   * if every tile is a scan tile (and there are three or more), consecutive tiles alternate between the sets;
        //   * everything else runs on the caller's stream with the first set.
        // The running lists are merged in tile order whatever stream a tile ran on (events).  A speculative plan keeps to
        // one set: the recheck pass has one boundary workspace.
        std::vector<uint32_t> g0s(db->tiles.size(), 0);
        size_t n_scan_tiles = 0, n_other_tiles = 0;
        for (size_t ti = 0; ti < db->tiles.size(); ++ti) {
            const Tile& t = db->tiles[ti];
            g0s[ti] = (uint64_t)cnt * t.n_seqs <= 0xFFFFFFFFull ? split_groups(db, t, q, q_begin + b, cnt, flags) : 0;
            if (t.n_seqs == 0) continue;
            if (g0s[ti] == 0) ++n_scan_tiles; else ++n_other_tiles;
        }
        const bool overlap = two_sets && !p.speculative;
        const bool alternate = overlap && n_other_tiles == 0 && n_scan_tiles >= 3;
        auto tile_topk = [&](const Tile& t, int32_t* buf, uint32_t* l_ids, int32_t* l_sc, uint32_t* l_cnt, hipStream_t st, hipEvent_t after) -> int {
            // segments: enough blocks to fill the chip when the batch is small, and no segment shorter than 16k columns
            const uint32_t segs = std::max(1u, std::min(std::min(16u, 256u / std::max(1u, cnt)), (t.n_seqs + 16383u) / 16384u));
            const uint32_t seg_len = ((t.n_seqs + segs - 1) / segs + 255u) / 256u * 256u;
            hipLaunchKernelGGL(topk_rows, dim3(cnt, segs), dim3(kTopkBlock), 0, st, buf, t.n_seqs, seg_len, k, kpad, min_score, l_ids,
                               l_sc, l_cnt, (const uint32_t*)t.d_col_ids, d_min_scores ? d_min_scores + b : (const int32_t*)nullptr);
            if (after) HIP_TRY(hipStreamWaitEvent(st, after, 0));        // the running lists are merged in tile order
            for (uint32_t sgm = 0; sgm < segs; ++sgm)
                hipLaunchKernelGGL(merge_rows, dim3(cnt), dim3(kTopkBlock), 0, st, d_ids + (uint64_t)b * k,
                                   d_top_scores + (uint64_t)b * k, d_counts + b, l_ids + (uint64_t)sgm * cnt * k,
                                   l_sc + (uint64_t)sgm * cnt * k, l_cnt + (uint64_t)sgm * cnt, k, kpad2);
            HIP_TRY(hipGetLastError());
            return S4G_OK;
        };
        if (overlap || n_other_tiles) {
            if (!cx->side_stream) {
                HIP_TRY(hipStreamCreateWithFlags(&cx->side_stream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&cx->side_fork, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&cx->side_join, hipEventDisableTiming));
            }
            for (hipEvent_t& e : cx->merge_done)
                if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(cx->side_fork, stream));                 // after this batch's uploads
            HIP_TRY(hipStreamWaitEvent(cx->side_stream, cx->side_fork, 0));
            scope.forked = true;
        }
        if (alternate && cx->ws_side_bytes < p.ws_bytes) {
            if (cx->d_ws_side) HIP_TRY(hipFree(cx->d_ws_side));
            cx->d_ws_side = nullptr; cx->ws_side_bytes = 0;
            HIP_TRY(hipMalloc((void**)&cx->d_ws_side, p.ws_bytes));
            cx->ws_side_bytes = p.ws_bytes;
        }
        int last = -1;          // set / stream of the previous tile's merge
        size_t scan_seen = 0;
        for (size_t ti = 0; ti < db->tiles.size(); ++ti) {
            const Tile& t = db->tiles[ti];
            if (t.n_seqs == 0) continue;
            const uint32_t g0 = g0s[ti];
            const bool as_pairs = g0 >= t.n_groups;
            const int par = !overlap ? 0 : as_pairs ? 1 : (alternate ? (int)(scan_seen++ & 1) : 0);
            hipStream_t st = par ? cx->side_stream : stream;
            char* set = reinterpret_cast<char*>(cx->d_tile_scores) + (uint64_t)par * set_bytes;
            int32_t* buf = reinterpret_cast<int32_t*>(set);
            uint32_t* l_ids = reinterpret_cast<uint32_t*>(buf + (uint64_t)batch * cols);
            int32_t* l_sc = reinterpret_cast<int32_t*>(l_ids + lists_max * k);
            uint32_t* l_cnt = reinterpret_cast<uint32_t*>(l_sc + lists_max * k);
            if (g0) cx->stat_pair_items += (uint64_t)cnt * std::min<uint64_t>(t.n_seqs, (uint64_t)g0 * 64);
            if (as_pairs) {
                rc = tile_as_pairs(db, t, q, q_begin + b, cnt, scorer, buf, t.n_seqs, true, st, t.n_seqs);
            } else if (g0) {        // split tile (always on the caller's stream): its longest groups as pairs on the side stream
                HIP_TRY(hipEventRecord(cx->side_fork, stream));
                HIP_TRY(hipStreamWaitEvent(cx->side_stream, cx->side_fork, 0));       // the buffer's previous reader (top-k) is done
                rc = tile_as_pairs(db, t, q, q_begin + b, cnt, scorer, buf, t.n_seqs, true, cx->side_stream, g0 * 64);
                if (rc) return rc;
                HIP_TRY(hipEventRecord(cx->side_join, cx->side_stream));
                rc = launch_tile(db, t, p.tiles[ti], q, scorer, p, q_begin + b, buf, t.n_seqs, t.d_perm_local, stream, false, g0);
                if (rc) return rc;
                HIP_TRY(hipStreamWaitEvent(stream, cx->side_join, 0));
            } else {
                rc = launch_tile(db, t, p.tiles[ti], q, scorer, p, q_begin + b, buf, t.n_seqs, t.d_perm_local, st, par != 0);
            }
            if (rc) return rc;
            if (p.speculative && !as_pairs) {
                rc = recheck_marked(db, q, scorer, q_begin + b, cnt, buf, t.n_seqs, t.n_seqs, t.d_col_ids, t.max_len4 * 4, st);
                if (rc) return rc;
            }
            rc = tile_topk(t, buf, l_ids, l_sc, l_cnt, st, last >= 0 && last != par ? cx->merge_done[last] : nullptr);
            if (rc) return rc;
            if (overlap) HIP_TRY(hipEventRecord(cx->merge_done[par], st));
            last = par;
        }
        if (last == 1) HIP_TRY(hipStreamWaitEvent(stream, cx->merge_done[1], 0));
    }
    return scope.release();
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


// s4g_pairs.hip -- C ABI: the latency path, one wave per (query, target) pair (sw_pair.hip.h): explicit pair lists,
// the pair route of s4g_scan, and the traceback of the survivors of a per-query alignDatabase call
// (/root/reference/sift4g/src/database_alignment.cpp:70-92).
#include "s4g_internal.hpp"
#include "sw_pair_coop.hip.h"
#include "sw_pair_pk.hip.h"

// ---------------------------------------------------------------------------
// Pair lists (the latency path, sw_pair.hip.h): explicit (query, target) pairs, one wave per pair.

constexpr int kPairClasses = 6;                 // row-block classes of sw_pair_scan / sw_pair_trace
constexpr int kPairRpl[kPairClasses] = {2, 4, 6, 8, 12, 16};
constexpr int kScanClasses = kPairClasses + 1;  // + the banded kernel (sw_pair_scan_band) for queries beyond 1024 residues
// + the cooperative kernels (sw_pair_coop.hip.h: the four waves of a block on one pair) for pairs whose own latency would dominate
// their launch; classes by rows per lane
constexpr int kCoopClasses = 3;
constexpr int kCoopRpl[kCoopClasses] = {2, 3, 4};
constexpr int kAllClasses = kScanClasses + kCoopClasses;
inline int coop_class(uint32_t lq) {            // -1: the query has more rows than four waves of 64 lanes x 4 rows hold
    for (int c = 0; c < kCoopClasses; ++c) if ((uint32_t)kCoopWaves * 64u * kCoopRpl[c] >= lq) return c;
    return -1;
}
inline int pair_class(uint32_t lq) {
    if (lq > (uint32_t)kPairMaxRows) return kPairClasses;
    int c = 0;
    while (64u * kPairRpl[c] < lq) ++c;
    return c;
}
constexpr uint64_t kBandWsBudget = 4ull << 30;  // boundary workspace of the banded kernel (8 B per column and wave)
// Measured rates (tools/pair_latency.py, profiles/r01_pair): cells per second a chip full of pair waves sustains and
// the cells per second ONE wave makes on one pair, against the scan's chip rate and the cells per second of one
// inter-sequence lane (which bounds a scan launch from below: one wave walks a whole (query pair, group) DP).
constexpr double kPairChipRate = 2.5e12, kPairWaveRate = 2.4e9 / 0.7, kPairCallOverhead = 60e-6;
constexpr double kScanChipRate = 11e12, kScanLaneRate = 2.4e9 / 37.0;
constexpr size_t kListsMinPairs = 1u << 16;      // below this a call is latency-bound: the list route's sort / gather / round trip would show

// boundary workspace of the banded kernels: one slot of `slot` columns per resident wave; the grid shrinks if the longest
// target of the launch would push it over the budget (caller holds cx->mu)
int ensure_band_ws(DeviceCtx* cx, uint32_t max_lt, uint32_t n_items, uint32_t full_grid, uint32_t* grid, uint64_t* slot) {
    *slot = ((uint64_t)std::max(1u, max_lt) + 63) / 64 * 64;
    *grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min(full_grid, n_items), kBandWsBudget / (*slot * 8)));
    const uint64_t bytes = *slot * 8 * *grid;
    if (cx->band_ws_bytes < bytes) {
        if (cx->d_band_ws) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(cx->d_band_ws)); }
        cx->d_band_ws = nullptr; cx->band_ws_bytes = 0;
        HIP_TRY(hipMalloc((void**)&cx->d_band_ws, bytes));
        cx->band_ws_bytes = bytes;
    }
    return S4G_OK;
}

constexpr int kPairPkCounter = 32;              // DeviceCtx::d_counter words: work queues of the packed pair classes (3)
constexpr int kPairPkClasses = 3;               // sw_pair_scan_pk<1, 2, 3>: the scan classes 0 .. 2 (queries of up to 128 / 256 / 384 residues)
constexpr int kCoopScanCounter = 50;            // DeviceCtx::d_counter words: work queues of the cooperative scan classes (3) ...
constexpr int kPairTraceCounter = 53;           // ... and of the traceback launches of a batch, one per class (kAllClasses = 10)
constexpr uint64_t kCoopWsBudget = 1ull << 30;

// boundary rows of the cooperative kernels: kCoopWaves - 1 rows of `slot` columns per resident block (caller holds cx->mu)
int ensure_coop_ws(DeviceCtx* cx, uint32_t max_lt, uint32_t n_items, uint32_t full_grid, uint32_t* grid, uint64_t* slot) {
    *slot = ((uint64_t)std::max(1u, max_lt) + 63) / 64 * 64;
    const uint64_t per_block = *slot * 8 * (kCoopWaves - 1);
    *grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min(full_grid, n_items), kCoopWsBudget / per_block));
    const uint64_t bytes = per_block * *grid;
    if (cx->coop_ws_bytes < bytes) {
        if (cx->d_coop_ws) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(cx->d_coop_ws)); }
        cx->d_coop_ws = nullptr; cx->coop_ws_bytes = 0;
        HIP_TRY(hipMalloc((void**)&cx->d_coop_ws, bytes));
        cx->coop_ws_bytes = bytes;
    }
    return S4G_OK;
}

// Which pairs of a submission go to the cooperative kernels.  Measured (profiles/r03_pair_coop.txt): a step of 2 rows per lane is
// not three times shorter than a step of 6 -- the shift from the neighbour lane and two dependent LDS reads (residue, then its profile
// row) are there either way, and with so little arithmetic between them the step waits for the LDS: ~300 clocks against ~260 for six
// rows in sw_pair_scan.  So the SCORES gain nothing (a 35 000-column target: 4.6 ms here, 3.8 ms on one wave; a per-query call of
// 5000 candidates 0.47 against 0.42 ms) and stay on one wave per pair.  The TRACEBACK step is three times heavier per row (direction
// codes, the end cell): there the cooperative sweep wins, 13.1 against 23.5 ms for that target -- and is used for the pairs whose own
// time on one wave (cells / kPairWaveRate) is at least a millisecond and half of what the whole call needs on a full chip: a titin
// among a query's hits.  S4G_COOP_PAIRS=0 turns the route off, =cost applies the cost rule (with a floor of 40 us) to scores as well,
// =all sends every pair of two bands or more there, scores and traceback (tests).
struct CoopRule {
    double min_cells;
    int mode;           // 0 off, 1 by cost, 2 every eligible pair
    CoopRule(double cells_total, bool traceback) {
        static const int m = []() { const char* e = getenv("S4G_COOP_PAIRS"); return !e ? 1 : (!strcmp(e, "0") ? 0 : (!strcmp(e, "all") ? 2 : (!strcmp(e, "cost") ? 3 : 1))); }();
        mode = m == 3 ? 1 : (m == 1 && !traceback ? 0 : m);
        min_cells = std::max(0.5 * cells_total / kPairChipRate, m == 3 ? 40e-6 : 1e-3) * kPairWaveRate;
    }
    int cls(uint32_t lq, uint32_t lt) const {          // class index (kScanClasses + c) or -1
        if (mode == 0 || lq > (uint32_t)kPairMaxRows || lt == 0 || lt >= 60000) return -1;
        const int c = coop_class(lq);
        if (c < 0 || lq <= 64u * (uint32_t)kCoopRpl[c]) return -1;         // (one band: nothing to share)
        if (mode == 1 && (double)lq * lt < min_cells) return -1;
        return kScanClasses + c;
    }
};

// The cooperative launches of a submission go to a stream of their own (kernels of one stream run one after the other, and these are
// the pairs that take longest): fork() after the uploads on `stream`, the destructor joins -- on every exit.
struct CoopFork {
    DeviceCtx* cx;
    hipStream_t stream;
    bool forked = false;
    CoopFork(DeviceCtx* c, hipStream_t s) : cx(c), stream(s) {}
    int fork() {
        if (forked) return S4G_OK;
        if (!cx->coop_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&cx->coop_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&cx->coop_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&cx->coop_join, hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(cx->coop_fork, stream));
        HIP_TRY(hipStreamWaitEvent(cx->coop_stream, cx->coop_fork, 0));
        forked = true;
        return S4G_OK;
    }
    int join() {
        if (!forked) return S4G_OK;
        forked = false;
        HIP_TRY(hipEventRecord(cx->coop_join, cx->coop_stream));
        HIP_TRY(hipStreamWaitEvent(stream, cx->coop_join, 0));
        return S4G_OK;
    }
    ~CoopFork() { (void)join(); }
};

template <int RPL>
int launch_pairs_coop(DeviceCtx* cx, PairArgs a, uint32_t max_lt, uint32_t* grid_cache, hipStream_t stream) {
    if (*grid_cache == 0) {
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_pair_scan_coop<RPL>, kCoopWaves * 64, 0));
        *grid_cache = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
    }
    uint64_t slot = 0;
    uint32_t grid = 0;
    int rc = ensure_coop_ws(cx, max_lt, a.n_pairs, *grid_cache, &grid, &slot);
    if (rc) return rc;
    PairCoopArgs ca;
    ca.p = a; ca.ws = reinterpret_cast<int2*>(cx->d_coop_ws); ca.ws_slot = slot;
    hipLaunchKernelGGL(sw_pair_scan_coop<RPL>, dim3(grid), dim3(kCoopWaves * 64), 0, stream, ca);
    HIP_TRY(hipGetLastError());
    return S4G_OK;
}

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

template <int M>
int launch_pairs_pk(DeviceCtx* cx, PairArgs a, uint32_t bias, uint32_t* grid_cache, hipStream_t stream) {
    if (*grid_cache == 0) {
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_pair_scan_pk<M>, 64, 0));
        *grid_cache = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
    }
    const uint32_t grid = std::max(1u, std::min(*grid_cache, a.n_pairs));
    PairPkArgs b;
    b.p = a; b.bias = bias;
    hipLaunchKernelGGL(sw_pair_scan_pk<M>, dim3(grid), dim3(64), 0, stream, b);
    HIP_TRY(hipGetLastError());
    return S4G_OK;
}

// Which pairs of a submission may run on the packed kernels (sw_pair_pk.hip.h): per scan class c < 3 the longest target whose
// drifting frame still fits the int16 window for EVERY query of the submission --
//     bound + B + (M + columns + 2 * 64 + 8) * ext <= 0x7BFF,   M = c + 1 registers per lane --
// 0 = none (a scorer with penalties beyond 16 bits, S4G_PAIR_PK=0, or no room at all).  Items of a class are ordered by target length
// (buckets of 32 residues, longest first), so the pairs beyond the cap are a prefix of the class: they stay on sw_pair_scan.
struct PairPkRule {
    uint32_t cap[kPairPkClasses] = {0, 0, 0};
    uint32_t bias = 0;
    PairPkRule() {}
    PairPkRule(const s4g_scorer* sc, uint64_t max_bound) {
        static const bool on = !(getenv("S4G_PAIR_PK") && atoi(getenv("S4G_PAIR_PK")) == 0);
        if (!on || sc->open >= 32768 || sc->ext >= 32768) return;
        bias = (uint32_t)std::max(std::max(-kPadScore, -sc->min_score), std::max(sc->open, 2 * sc->ext));
        const uint64_t window = 0x7BFF, step = (uint64_t)std::max(0, sc->max_score) + 2 * (uint64_t)sc->ext;
        for (int c = 0; c < kPairPkClasses; ++c) {
            const uint64_t fixed = max_bound + bias + step + (uint64_t)(c + 1 + 2 * 64 + 8) * (uint64_t)sc->ext;
            if (fixed >= window) continue;
            const uint64_t cols = sc->ext ? (window - fixed) / (uint64_t)sc->ext : 0xFFFFFFFFull;
            cap[c] = (uint32_t)std::min<uint64_t>(cols, 0x7FFFFFFFull);
        }
    }
    // conservative at the granularity the items are ordered by: a target in the cap's own bucket of 32 residues counts as too long
    bool takes(int c, uint32_t lt) const { return c < kPairPkClasses && (lt >> 5) < (cap[c] >> 5); }
};

int warm_pair_kernels(DeviceCtx* cx) {
    int rc = device_props(cx);
    if (rc) return rc;
    auto occ = [&](const void* k, uint32_t* cache) -> int {
        if (*cache) return S4G_OK;
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 64, 0));
        *cache = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
        return S4G_OK;
    };
    // the classes a 300-400 residue query uses for scores and traceback (the others load with the same code object)
    rc = occ((const void*)sw_pair_scan<6>, &cx->grid_pair[2]);
    if (!rc) rc = occ((const void*)sw_pair_scan<8>, &cx->grid_pair[3]);
    if (!rc) rc = occ((const void*)sw_pair_scan_pk<3>, &cx->grid_pair_pk[2]);
    if (!rc) rc = occ((const void*)sw_pair_trace<6>, &cx->grid_pair_trace[2]);
    if (!rc) rc = occ((const void*)sw_pair_trace<8>, &cx->grid_pair_trace[3]);
    if (!rc && cx->grid_pair_coop[0] == 0) {
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_pair_scan_coop<2>, kCoopWaves * 64, 0));
        cx->grid_pair_coop[0] = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_pair_trace_coop<2>, kCoopWaves * 64, 0));
        cx->grid_pair_trace_coop[0] = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
    }
    PairItem* items = nullptr;
    if (!rc) rc = pair_stage(cx, 1u << 16, &items);           // pinned staging + device items: a pinned allocation costs milliseconds
    return rc;
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
// (cls_begin has kAllClasses + 1 entries; max_lt_coop = longest target among the cooperative classes)
// pk_first (optional): per packed class, the first item of the class that the packed kernel takes (items before it are too long for
// its window and run on sw_pair_scan); pk_bias its bias.  NULL: no packed launches.
int pair_submit(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t n, const uint32_t* cls_begin,
                int32_t* d_scores, hipStream_t stream, uint32_t max_lt_band, uint32_t max_lt_coop = 0, const uint32_t* pk_first = nullptr,
                uint32_t pk_bias = 0) {
    DeviceCtx* cx = device_ctx(db->device);
    int rc = device_props(cx);
    if (rc) return rc;
    rc = ensure_seq_tables(db);
    if (rc) return rc;
    if (!cx->d_counter) HIP_TRY(hipMalloc((void**)&cx->d_counter, 256));
    if (!cx->d_mat) HIP_TRY(hipMalloc((void**)&cx->d_mat, 1024));
    ScratchScope scope(db, stream);
    rc = scope.acquire();
    if (rc) return rc;
    { const int rcm = ensure_mat(cx, scorer, stream); if (rcm) return rcm; }
    HIP_TRY(hipMemcpyAsync(cx->d_items, cx->h_items, (size_t)n * sizeof(PairItem), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(cx->items_done, stream));
    HIP_TRY(hipMemsetAsync(cx->d_counter + 24, 0, kScanClasses * 4, stream));
    if (pk_first) HIP_TRY(hipMemsetAsync(cx->d_counter + kPairPkCounter, 0, kPairPkClasses * 4, stream));
    cx->stat_pair_pk = 0;
    if (cls_begin[kAllClasses] != cls_begin[kScanClasses]) HIP_TRY(hipMemsetAsync(cx->d_counter + kCoopScanCounter, 0, kCoopClasses * 4, stream));
    // the cooperative classes first, on their own stream: they are the submission's longest pairs
    CoopFork coop_fork(cx, stream);
    for (int cc = 0; cc < kAllClasses; ++cc) {
        const int c = cc < kCoopClasses ? kScanClasses + cc : cc - kCoopClasses;
        const uint32_t b = cls_begin[c], e = cls_begin[c + 1];
        if (b == e) continue;
        PairArgs a;
        a.items = cx->d_items + b; a.seq_addr = db->d_seq_addr; a.seq_len = db->d_seq_len;
        a.q_codes = q->d_codes; a.q_off = q->d_off; a.q_len = q->d_len; a.mat = cx->d_mat;
        a.scores = d_scores; a.counter = c >= kScanClasses ? cx->d_counter + kCoopScanCounter + (c - kScanClasses) : cx->d_counter + 24 + c; a.n_pairs = e - b;
        a.open = (uint32_t)scorer->open; a.ext = (uint32_t)scorer->ext;
        if (c >= kScanClasses) {
            rc = coop_fork.fork();
            if (rc) return rc;
            switch (c - kScanClasses) {
                case 0: rc = launch_pairs_coop<2>(cx, a, max_lt_coop, &cx->grid_pair_coop[0], cx->coop_stream); break;
                case 1: rc = launch_pairs_coop<3>(cx, a, max_lt_coop, &cx->grid_pair_coop[1], cx->coop_stream); break;
                default: rc = launch_pairs_coop<4>(cx, a, max_lt_coop, &cx->grid_pair_coop[2], cx->coop_stream); break;
            }
            if (rc) return rc;
            continue;
        }
        if (pk_first && c < kPairPkClasses && pk_first[c] < e) {
            // the packed kernel takes the class from pk_first[c] on (two rows per register: half the steps' instructions); what is too
            // long for its int16 window -- the head of the class -- stays on the int32 kernel below
            PairArgs pa = a;
            pa.items = cx->d_items + pk_first[c]; pa.n_pairs = e - pk_first[c]; pa.counter = cx->d_counter + kPairPkCounter + c;
            switch (c) {
                case 0: rc = launch_pairs_pk<1>(cx, pa, pk_bias, &cx->grid_pair_pk[0], stream); break;
                case 1: rc = launch_pairs_pk<2>(cx, pa, pk_bias, &cx->grid_pair_pk[1], stream); break;
                default: rc = launch_pairs_pk<3>(cx, pa, pk_bias, &cx->grid_pair_pk[2], stream); break;
            }
            if (rc) return rc;
            cx->stat_pair_pk += pa.n_pairs;
            a.n_pairs = pk_first[c] - b;
            if (a.n_pairs == 0) continue;
        }
        if (c == kPairClasses) {        // banded kernel: per-wave boundary workspace sized by the launch's longest target
            if (cx->grid_pair_band == 0) {
                int per_cu = 0;
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_pair_scan_band<16>, 64, 0));
                cx->grid_pair_band = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
            }
            uint64_t slot = 0;
            uint32_t grid = 0;
            rc = ensure_band_ws(cx, max_lt_band, a.n_pairs, cx->grid_pair_band, &grid, &slot);
            if (rc) return rc;
            PairBandArgs ba;
            ba.p = a; ba.band_ws = reinterpret_cast<int2*>(cx->d_band_ws); ba.band_ws_slot = slot;
            hipLaunchKernelGGL(sw_pair_scan_band<16>, dim3(grid), dim3(64), 0, stream, ba);
            HIP_TRY(hipGetLastError());
            continue;
        }
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
    rc = coop_fork.join();
    if (rc) return rc;
    return scope.release();
}

int recheck_marked(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t q_begin, uint32_t q_count,
                   int32_t* d_scores, uint64_t stride, uint32_t n_cols, const uint32_t* d_col_ids, uint32_t max_lt, hipStream_t stream) {
    if (q_count == 0 || n_cols == 0) return S4G_OK;
    DeviceCtx* cx = device_ctx(db->device);
    int rc = device_props(cx);
    if (rc) return rc;
    rc = ensure_seq_tables(db);
    if (rc) return rc;
    if (cx->grid_pair_recheck == 0) {
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_pair_recheck<16>, 64, 0));
        cx->grid_pair_recheck = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
    }
    const uint64_t slabs = (((uint64_t)n_cols + 63) / 64) * q_count;
    uint64_t slot = 0;
    uint32_t grid = 0;
    rc = ensure_band_ws(cx, max_lt, (uint32_t)std::min<uint64_t>(slabs, 0xFFFFFFFFu), cx->grid_pair_recheck, &grid, &slot);
    if (rc) return rc;
    RecheckArgs r;
    PairArgs& a = r.b.p;
    a.items = nullptr; a.seq_addr = db->d_seq_addr; a.seq_len = db->d_seq_len;
    a.q_codes = q->d_codes; a.q_off = q->d_off; a.q_len = q->d_len; a.mat = cx->d_mat;
    a.scores = d_scores; a.counter = cx->d_counter + kStatRechecked; a.n_pairs = 0;     // counts the entries it recomputes (s4g_scan_stats)
    a.open = (uint32_t)scorer->open; a.ext = (uint32_t)scorer->ext;
    r.b.band_ws = reinterpret_cast<int2*>(cx->d_band_ws); r.b.band_ws_slot = slot;
    r.stride = stride; r.q_begin = q_begin; r.q_count = q_count; r.n_cols = n_cols; r.col_ids = d_col_ids;
    hipLaunchKernelGGL(sw_pair_recheck<16>, dim3(grid), dim3(64), 0, stream, r);
    HIP_TRY(hipGetLastError());
    return S4G_OK;
}

// Explicit list -> device scores, asynchronous.  sel = indices into pair_q / pair_t.
// Order inside a class: longest target first at 32-residue granularity (counting sort; a launch cannot end before its
// longest pair does, so that one must not start last).
int pairs_enqueue(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, const std::vector<uint32_t>& sel,
                  const uint32_t* pair_q, const uint32_t* pair_t, std::vector<uint32_t>* where, int32_t** d_sc_out, hipStream_t stream,
                  bool pk_allowed) {
    DeviceCtx* cx = device_ctx(db->device);
    const uint32_t n = (uint32_t)sel.size();
    PairItem* items = nullptr;
    int rc = pair_stage(cx, n, &items);
    if (rc) return rc;
    constexpr uint32_t kBuckets = 2048;
    static thread_local std::vector<uint32_t> count;
    count.assign((size_t)kAllClasses * kBuckets + 1, 0);
    uint32_t max_lt_band = 0, max_lt_coop = 0;
    double cells_total = 0;
    for (uint32_t i = 0; i < n; ++i) cells_total += (double)q->len[pair_q[sel[i]]] * db->len[pair_t[sel[i]]];
    const CoopRule coop(cells_total, false);
    ensure_bounds(q, scorer);
    uint64_t max_bound = 0;
    {   // the largest score bound among the submission's queries (a per-query call has one)
        uint32_t last_q = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < n; ++i) { const uint32_t qi = pair_q[sel[i]]; if (qi != last_q) { max_bound = std::max(max_bound, q->score_bound[qi]); last_q = qi; } }
    }
    const PairPkRule pk(scorer, max_bound);
    uint32_t pk_unfit[kPairPkClasses] = {0, 0, 0}, pk_max_lt[kPairPkClasses] = {0, 0, 0};
    double pk_cells[kPairPkClasses] = {0, 0, 0};
    auto class_of = [&](uint32_t pi) {
        const int cc = coop.cls(q->len[pair_q[pi]], db->len[pair_t[pi]]);
        return cc >= 0 ? cc : pair_class(q->len[pair_q[pi]]);
    };
    auto key_of = [&](uint32_t pi) {
        const uint32_t b = std::min<uint32_t>(db->len[pair_t[pi]] >> 5, kBuckets - 1);
        return (uint32_t)class_of(pi) * kBuckets + (kBuckets - 1 - b);
    };
    for (uint32_t i = 0; i < n; ++i) {
        ++count[key_of(sel[i]) + 1];
        const int c = class_of(sel[i]);
        if (c == kPairClasses) max_lt_band = std::max(max_lt_band, db->len[pair_t[sel[i]]]);
        if (c >= kScanClasses) max_lt_coop = std::max(max_lt_coop, db->len[pair_t[sel[i]]]);
        if (c < kPairPkClasses) {
            const uint32_t lt = db->len[pair_t[sel[i]]];
            if (!pk.takes(c, lt)) ++pk_unfit[c];
            else { pk_cells[c] += (double)q->len[pair_q[sel[i]]] * lt; pk_max_lt[c] = std::max(pk_max_lt[c], lt); }
        }
    }
    for (size_t k = 1; k < count.size(); ++k) count[k] += count[k - 1];
    uint32_t cls_begin[kAllClasses + 1];
    for (int c = 0; c <= kAllClasses; ++c) cls_begin[c] = count[(size_t)c * kBuckets];
    // The packed kernels win where a call waits for its longest pair: a wave walks a pair in ~150 clocks per step instead of ~260.  Their
    // two tables per wave (48 KB of LDS at three registers per lane) leave room for 3 waves per CU against the int32 kernel's 8, so
    // once there is work for every resident wave several times over the int32 kernel is ahead again (16 queries x 3645 candidates:
    // 4.0 ms against 2.9; one query x 3645: 0.30 ms against 0.41; the 9 000 long-tail pairs of a list call: 1.4 ms against 1.1;
    // profiles/r04_pair_pk/).  So a class goes there while its cells, at the packed kernels' chip rate (768 waves x 6 rows x 64 lanes per
    // ~150 clocks), take no longer than ONE int32 wave needs for the class's longest pair -- the time the int32 launch cannot beat:
    //     cells / 1966 per clock  <=  (longest target + 64) x 260 clocks.          S4G_PAIR_PK_MAX scales the right-hand side (percent).
    // Pairs that run beside the list kernel (pk_allowed = false) stay on the int32 kernel: the list kernel's blocks hold 108 KB of a
    // CU's LDS, and one packed wave fits beside them where two int32 waves do.
    static const double pk_scale = getenv("S4G_PAIR_PK_MAX") ? atof(getenv("S4G_PAIR_PK_MAX")) / 100.0 : 1.0;
    uint32_t pk_first[kPairPkClasses];
    for (int c = 0; c < kPairPkClasses; ++c) {
        pk_first[c] = cls_begin[c] + pk_unfit[c];                   // (the too-long ones are the head of the class)
        const double budget = ((double)pk_max_lt[c] + 64.0) * 260.0 * 1966.0 * pk_scale;
        if (!pk_allowed || pk_cells[c] > budget) pk_first[c] = cls_begin[c + 1];
    }
    where->resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t pi = sel[i];
        const uint32_t at = count[key_of(pi)]++;
        items[at] = {pair_q[pi], pair_t[pi], at};
        (*where)[i] = at;                      // where this pair's score will sit
    }
    cx->stat_coop_scan = cls_begin[kAllClasses] - cls_begin[kScanClasses];
    *d_sc_out = reinterpret_cast<int32_t*>(cx->d_items + cx->items_cap);
    return pair_submit(db, q, scorer, n, cls_begin, *d_sc_out, stream, max_lt_band, max_lt_coop, pk.bias ? pk_first : nullptr, pk.bias);
}

// Explicit list -> host scores (synchronous).
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
    cx->stat_pair_items = n; cx->stat_list_pairs = 0;
    static thread_local std::vector<uint32_t> where;
    int32_t* d_sc = nullptr;
    int rc = pairs_enqueue(db, q, scorer, sel, pair_q, pair_t, &where, &d_sc, stream);
    if (rc) return rc;
    auto t2 = now();
    // the scores come back into the pinned item staging (12 bytes per pair, its upload is complete by then): a download into
    // pageable memory would have the runtime pin the destination first
    HIP_TRY(hipEventSynchronize(cx->items_done));
    int32_t* sc = reinterpret_cast<int32_t*>(cx->h_items);
    HIP_TRY(hipMemcpyAsync(sc, d_sc, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (uint32_t i = 0; i < n; ++i) h_scores[sel[i]] = sc[where[i]];
    if (prof) {
        auto t3 = now();
        auto us = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1e3; };
        fprintf(stderr, "[s4g profile] pairs n=%u: order+upload+launch %.1f us, wait+download %.1f us\n", n, us(t0, t2), us(t2, t3));
    }
    return S4G_OK;
}

// One tile of a scan as pairs: the tile's sequences x the queries [q_begin, q_begin + q_count), scores written to
// d_scores[(query - q_begin) * stride + column], column = original target id (local_cols = false) or the rank of the id
// inside the tile (local_cols = true, the layout s4g_scan_topk's tile buffer uses).  Used for tiles whose few, very long
// sequences would make an inter-sequence launch wait on one wave (a titin among the targets); caller holds cx->mu.
// How many of a tile's leading groups (its longest sequences) should be scored as pairs while the scan kernels take the
// rest: the scan kernels are bounded from below by one wave walking the longest group it is given (37 cycles per cell),
// pair waves by 0.7 cycles per cell of the longest pair but at a quarter of the chip rate.  Estimate for a handful of
// split points; both parts run concurrently, so the call takes max(latencies, sum of the chip times).  0 = no split.
uint32_t split_groups(const s4g_db* db, const Tile& t, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, uint32_t flags) {
    if (flags != 0 || q_count == 0 || t.n_groups == 0) return 0;
    uint32_t lq_max = 0;
    double rows = 0;
    for (uint32_t i = 0; i < q_count; ++i) { lq_max = std::max(lq_max, q->len[q_begin + i]); rows += q->len[q_begin + i]; }
    const uint32_t G = t.n_groups;
    auto len_of = [&](uint32_t g) {     // padded length of group g, residues
        const uint64_t end = g + 1 < G ? t.h_grp_off[g + 1] : t.padded_residues / 4;
        return (double)((end - t.h_grp_off[g]) / 64 * 4);
    };
    const double total = (double)t.padded_residues;
    auto cost = [&](uint32_t g0) {
        // the scan kernels pay for every lane of a group, filled or not (a tile of three titins is 3/64 full); pair waves
        // only for the sequences that exist
        const double res_scan = g0 < G ? total - (double)t.h_grp_off[g0] * 4.0 : 0.0;
        const double res_pairs = g0 < G ? (double)t.h_grp_off[g0] * 4.0 : (double)t.residues;
        const double lat_pairs = g0 ? lq_max * len_of(0) / kPairWaveRate + kPairCallOverhead : 0.0;
        const double lat_scan = g0 < G ? lq_max * len_of(g0) / kScanLaneRate : 0.0;
        // A scan launch whose longest item (alone: lat_scan) is comparable to its whole chip time does not end at the larger
        // of the two: the long items share their SIMDs until the bulk is through and finish alone.  Fitted on 8 / 16 / 32
        // queries x 100k targets (21.5 / 23.8 / 34.9 ms without a split; 9.6 / 17.8 ms with the best one):
        // time = max(work, lat + 0.32 work) once lat >= work, fading to max(work, lat) as lat falls to half the work.
        const double work_scan = rows * res_scan / kScanChipRate;
        const double x = work_scan > 0 ? std::min(1.0, std::max(0.0, (lat_scan - 0.5 * work_scan) / (0.5 * work_scan))) : 0.0;
        const double t_scan = std::max(work_scan, lat_scan + 0.32 * work_scan * x);
        return std::max(std::max(lat_pairs, t_scan), work_scan + rows * res_pairs / kPairChipRate);    // both parts share the chip
    };
    double best = cost(0);
    uint32_t best_g0 = 0;
    const double base = best;
    for (uint32_t g0 = 1;; g0 = g0 < 8 ? g0 + 1 : g0 + g0 / 4) {
        if (g0 > G) g0 = G;
        if ((uint64_t)q_count * std::min<uint64_t>(t.n_seqs, (uint64_t)g0 * 64) <= (4u << 20)) {
            const double c = cost(g0);
            if (c < best) { best = c; best_g0 = g0; }
        }
        if (g0 == G) break;
    }
    return best < 0.9 * base ? best_g0 : 0;
}

int tile_as_pairs(const s4g_db* db, const Tile& t, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* scorer,
                  int32_t* d_scores, uint64_t stride, bool local_cols, hipStream_t stream, uint32_t n_pos) {
    n_pos = std::min(n_pos, t.n_seqs);          // the tile's first n_pos sequences (its longest)
    DeviceCtx* cx = device_ctx(db->device);
    int rc = ensure_seq_tables(db);
    if (rc) return rc;
    const uint64_t n_items = (uint64_t)q_count * n_pos;
    if (n_items * 1 > 0xFFFFFFFFull || (uint64_t)q_count * stride > 0xFFFFFFFFull) return fail(S4G_ERR_INVALID, "tile too large for the pair route");
    PairItem* items = nullptr;
    rc = pair_stage(cx, n_items, &items);
    if (rc) return rc;
    // column of the sequence at tile position p: the rank of its id inside the tile.  Built once per tile and kept (it used to
    // be rebuilt -- a sort of the whole tile -- on every call: 6 ms of host time for a 100 000-sequence tile)
    if (local_cols && t.h_col_of_pos.size() != t.n_seqs) {
        std::vector<uint32_t> ids(db->id_of_pos.begin() + t.pos0, db->id_of_pos.begin() + t.pos0 + t.n_seqs);
        std::sort(ids.begin(), ids.end());
        t.h_col_of_pos.resize(t.n_seqs);
        for (uint32_t p = 0; p < t.n_seqs; ++p)
            t.h_col_of_pos[p] = (uint32_t)(std::lower_bound(ids.begin(), ids.end(), db->id_of_pos[t.pos0 + p]) - ids.begin());
    }
    const std::vector<uint32_t>& col = t.h_col_of_pos;
    uint32_t cls_begin[kAllClasses + 1];
    uint32_t at = 0;
    std::vector<uint32_t> qs;
    for (int c = 0; c < kScanClasses; ++c) {
        cls_begin[c] = at;
        qs.clear();
        for (uint32_t i = 0; i < q_count; ++i) if (pair_class(q->len[q_begin + i]) == c) qs.push_back(i);
        if (qs.empty()) continue;
        for (uint32_t p = 0; p < n_pos; ++p) {           // tile positions are length-sorted, longest first
            const uint32_t id = db->id_of_pos[t.pos0 + p];
            const uint32_t column = local_cols ? col[p] : id;
            for (uint32_t i : qs) items[at++] = {q_begin + i, id, (uint32_t)((uint64_t)i * stride + column)};
        }
    }
    for (int c = kScanClasses; c <= kAllClasses; ++c) cls_begin[c] = at;         // (no cooperative classes on this route)
    return pair_submit(db, q, scorer, at, cls_begin, d_scores, stream, t.max_len4 * 4);
}

// s4g_scan's latency route: the whole [q_count][n_seqs] matrix as pairs (caller holds cx->mu).  Taken when the estimate
// says the inter-sequence launch would wait on its longest item for longer than a chip full of pair waves needs for
// everything; never when a kernel was forced by flags.
// what s4g_scan_plan reports: does a scan of these queries go to the pair kernels entirely?
bool scan_prefers_pairs(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, uint32_t flags) {
    if (flags != 0 || q_count == 0 || db->n_seqs == 0 || (uint64_t)q_count * db->n_seqs > 0xFFFFFFFFull) return false;
    for (const Tile& t : db->tiles)
        if (t.n_seqs && split_groups(db, t, q, q_begin, q_count, flags) < t.n_groups) return false;
    return true;
}

int scan_as_pairs(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* scorer,
                  int32_t* d_scores, hipStream_t stream, bool* done) {
    *done = false;
    const uint64_t n_items = (uint64_t)q_count * db->n_seqs;
    uint32_t max_t = 0;
    for (const Tile& t : db->tiles) max_t = std::max(max_t, t.max_len4 * 4);
    DeviceCtx* cx = device_ctx(db->device);
    int rc = ensure_seq_tables(db);
    if (rc) return rc;
    PairItem* items = nullptr;
    rc = pair_stage(cx, n_items, &items);
    if (rc) return rc;
    // class-major; inside a class the targets in length-sorted order (longest first), queries innermost
    uint32_t cls_begin[kAllClasses + 1];
    uint32_t at = 0;
    std::vector<uint32_t> qs;
    for (int c = 0; c < kScanClasses; ++c) {
        cls_begin[c] = at;
        qs.clear();
        for (uint32_t i = 0; i < q_count; ++i) if (pair_class(q->len[q_begin + i]) == c) qs.push_back(i);
        if (qs.empty()) continue;
        for (uint32_t pos = 0; pos < db->n_seqs; ++pos) {
            const uint32_t id = db->id_of_pos[pos];
            for (uint32_t i : qs) items[at++] = {q_begin + i, id, (uint32_t)((uint64_t)i * db->n_seqs + id)};
        }
    }
    for (int c = kScanClasses; c <= kAllClasses; ++c) cls_begin[c] = at;         // (no cooperative classes on this route)
    rc = pair_submit(db, q, scorer, at, cls_begin, d_scores, stream, max_t);
    if (rc) return rc;
    *done = true;
    return S4G_OK;
}


S4G_API int s4g_score_pairs_host(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint64_t n_pairs,
                                 const uint32_t* pair_q, const uint32_t* pair_t, int32_t* h_scores, uint32_t flags) {
    if (!db || !q || !scorer) return fail(S4G_ERR_INVALID, "NULL argument");
    if (n_pairs == 0) return S4G_OK;
    if (!pair_q || !pair_t || !h_scores) return fail(S4G_ERR_INVALID, "NULL argument");
    if (db->device != q->device) return fail(S4G_ERR_INVALID, "db on device %d, queries on device %d", db->device, q->device);
    if (n_pairs >= 0xFFFFFFFFull) return fail(S4G_ERR_INVALID, "too many pairs in one call");
    if (((flags & S4G_PAIRS_FORCE_SCAN) != 0) + ((flags & S4G_PAIRS_FORCE_WAVE) != 0) + ((flags & S4G_PAIRS_FORCE_LISTS) != 0) > 1)
        return fail(S4G_ERR_INVALID, "contradictory flags");
    if (db->stream) return fail(S4G_ERR_INVALID, "a streamed database (s4g_db_open_streamed) serves s4g_scan_topk / s4g_scan_topk_host only");
    if (db->cached) flags = S4G_PAIRS_FORCE_WAVE;        // no tiles to scan, no length order to gather from: one wave per pair
    // One pass: range checks and, per query, what the route estimate needs (cells, longest pair, pair count).
    // Two exact routes per query.  "wave": its pairs one by one (sw_pair_scan / sw_pair_scan_band, or the packed list kernel for
    // large batches).  "scan": score the query against the whole resident database and pick the columns.  Per query the cheaper
    // estimate wins: few candidates -> wave, candidate list ~ the whole database -> scan.
    std::vector<double> q_cells(q->n, 0.0);
    std::vector<uint32_t> q_longest(q->n, 0), q_pairs(q->n, 0);
    {
        const unsigned parts = parallel_parts(n_pairs, 1u << 16);
        struct Part { std::vector<double> cells; std::vector<uint32_t> longest, pairs; uint64_t bad = ~0ull; };
        std::vector<Part> part(parts);
        parallel_for(n_pairs, parts, [&](unsigned p, unsigned, uint64_t b, uint64_t e) {
            std::vector<double> cells(q->n, 0.0);               // locals, handed over once (the Part structs share cache lines)
            std::vector<uint32_t> longest(q->n, 0), pairs(q->n, 0);
            uint64_t bad = ~0ull;
            for (uint64_t i = b; i < e; ++i) {
                const uint32_t qi = pair_q[i], t = pair_t[i];
                if (qi >= q->n || t >= db->n_seqs) { bad = i; break; }
                const uint32_t lt = db->len[t];
                cells[qi] += lt;
                longest[qi] = std::max(longest[qi], lt);
                ++pairs[qi];
            }
            Part& P = part[p];
            P.cells.swap(cells); P.longest.swap(longest); P.pairs.swap(pairs); P.bad = bad;
        });
        for (const Part& P : part) {
            if (P.bad != ~0ull) return fail(S4G_ERR_INVALID, "pair %llu out of range", (unsigned long long)P.bad);
            for (uint32_t i = 0; i < q->n; ++i) { q_cells[i] += P.cells[i]; q_longest[i] = std::max(q_longest[i], P.longest[i]); q_pairs[i] += P.pairs[i]; }
        }
    }
    uint32_t max_t = 0;
    for (const Tile& t : db->tiles) max_t = std::max(max_t, t.max_len4 * 4);
    std::vector<char> route(q->n, 0);                   // 0: no pairs / empty query, 1: wave, 2: scan
    std::vector<uint32_t> scan_queries;
    uint64_t n_wave = 0, n_other = 0;
    for (uint32_t qi = 0; qi < q->n; ++qi) {
        if (!q_pairs[qi]) continue;
        const uint32_t lq = q->len[qi];
        if (lq == 0) { n_other += q_pairs[qi]; continue; }
        bool wave = true;
        if (!(flags & (S4G_PAIRS_FORCE_WAVE | S4G_PAIRS_FORCE_LISTS))) {
            const double cells = (double)lq * q_cells[qi], longest = (double)lq * q_longest[qi];
            const double t_wave = std::max(cells / kPairChipRate, longest / kPairWaveRate) + kPairCallOverhead;
            const double full = (double)lq * (double)db->residues;
            const double t_scan = std::max(full / kScanChipRate, (double)lq * max_t / kScanLaneRate) + (double)db->n_seqs * 4.0 / 20e9;
            wave = t_wave <= t_scan;
        }
        if (flags & S4G_PAIRS_FORCE_SCAN) wave = false;
        route[qi] = wave ? 1 : 2;
        if (wave) n_wave += q_pairs[qi]; else { scan_queries.push_back(qi); n_other += q_pairs[qi]; }
    }
    // pair indices of the wave route -- not materialised when that is every pair of the call (the common case) -- and, only if
    // some query scans, the pairs by query
    const bool all_wave = n_wave == n_pairs;
    std::vector<uint32_t> wave_sel(all_wave ? 0 : n_wave);
    std::vector<std::vector<uint32_t>> by_q(scan_queries.empty() ? 0 : q->n);
    if (!all_wave) {
        uint64_t at = 0;
        for (uint64_t i = 0; i < n_pairs; ++i) {
            const uint32_t qi = pair_q[i];
            if (route[qi] == 1) wave_sel[at++] = (uint32_t)i;
            else if (route[qi] == 2) by_q[qi].push_back((uint32_t)i);
            else h_scores[i] = 0;                       // empty query
        }
    }
    (void)n_other;
    if (n_wave) {
        // Large batches of lists (the batched wrapper: every query of a chunk with its own candidates) go to the packed list
        // kernel, except their longest targets; small ones are latency-bound and stay with one wave per pair.
        bool used = false;
        if (!(flags & S4G_PAIRS_FORCE_WAVE) && ((flags & S4G_PAIRS_FORCE_LISTS) || n_wave >= kListsMinPairs)) {
            int rc = score_lists_device(db, q, scorer, all_wave ? nullptr : wave_sel.data(), (uint32_t)n_wave, pair_q, pair_t, h_scores,
                                        (flags & S4G_PAIRS_FORCE_LISTS) != 0, (flags & S4G_PAIRS_FORCE_LISTS) ? (flags >> 8) : 0u, &used);
            if (rc && (flags & S4G_PAIRS_FORCE_LISTS)) return rc;
            if (rc) {       // the list route needs gigabytes of workspace; if it cannot have them the pair waves still can do the job
                (void)hipGetLastError();
                used = false;
            }
        }
        if (!used) {
            if (all_wave) { wave_sel.resize(n_wave); std::iota(wave_sel.begin(), wave_sel.end(), 0u); }
            int rc = score_pairs_device(db, q, scorer, wave_sel, pair_q, pair_t, h_scores);
            if (rc) return rc;
        }
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
// Traceback of the survivors on the device (step 4 of alignDatabase).

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

// Traceback with one wave per pair (sw_pair_trace; sw_pair_trace_band beyond 1024 query residues): `sel` = pair indices.
// A traceback in flight (s4g_traceback_submit / s4g_traceback_finish): everything finish needs once the stream has passed `done`.
// The results sit in a pinned buffer of the job's own (taken from the device's pool, returned by finish): the next submission on
// the same device reuses the device arena in stream order, but must not write its items into host memory a download is still landing in.
struct s4g_trace_job {
    const s4g_db* db = nullptr;             // a reference (s4g_db::refs) while in flight: released by finish
    int device = 0;
    uint32_t n_pairs = 0;
    bool in_flight = false;                 // false: the results are in out5 / paths already (the synchronous fallback ran)
    hipEvent_t done = nullptr;
    char* pin = nullptr; int pin_slab = -1;           // a piece of cx->trace_pin_slabs[pin_slab]
    uint64_t back_off = 0, best_off = 0, paths_off = 0;       // inside pin: out5 | best | paths, as they sit in the arena
    std::vector<uint32_t> order;            // slot -> original pair index
    std::vector<int32_t> want, score;       // per slot: the maximum the kernel must report, the score passed in
    std::vector<uint64_t> path_end;         // per slot: end of the pair's path region (the walk writes backwards from it)
    std::vector<uint64_t> path_off;         // the caller's layout (n_pairs + 1)
    std::vector<int32_t> out5;              // fallback results
    std::vector<uint8_t> paths;
};

int traceback_pairs(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, const std::vector<uint32_t>& sel,
                    const uint32_t* pair_q, const uint32_t* pair_t, const int32_t* pair_score,
                    int32_t* h_out, const uint64_t* path_off, uint8_t* h_paths, s4g_trace_job* job) {
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
    { const int rcm = ensure_mat(cx, scorer, stream); if (rcm) return rcm; }
    // A submission that may stay in flight (job != nullptr; pairs of the one-wave classes only: the banded and the cooperative kernels
    // share workspaces with the scoring kernels) runs on the device's TRACE stream, behind an event on the null stream that covers the
    // uploads its kernels read (the candidate database's tables, the query set, the substitution table): the next call's scoring, on
    // the null stream, then runs beside it instead of behind it.  Whatever else uses the traceback arena waits for the last such job.
    bool on_trace_stream = false;
    if (job) {
        bool plain = true;
        for (uint32_t i = 0; i < (uint32_t)sel.size() && plain; ++i) {
            const uint32_t pi = sel[i];
            if (q->len[pair_q[pi]] > (uint32_t)kPairMaxRows) plain = false;
        }
        if (plain) {
            if (!cx->trace_stream) {
                HIP_TRY(hipStreamCreateWithFlags(&cx->trace_stream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&cx->trace_inputs, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&cx->trace_last, hipEventDisableTiming));
            }
            HIP_TRY(hipEventRecord(cx->trace_inputs, nullptr));
            HIP_TRY(hipStreamWaitEvent(cx->trace_stream, cx->trace_inputs, 0));
            stream = cx->trace_stream;
            on_trace_stream = true;
        }
    }
    if (!on_trace_stream && cx->trace_last) HIP_TRY(hipStreamWaitEvent(stream, cx->trace_last, 0));
    // class-major (the cooperative classes first: the longest pairs start first), longest target first inside a class (32-residue
    // granularity), as in pairs_enqueue
    const uint32_t n = (uint32_t)sel.size();
    constexpr uint32_t kBuckets = 2048;
    std::vector<uint32_t> count((size_t)kAllClasses * kBuckets + 1, 0), order(n);
    double cells_total = 0;
    for (uint32_t i = 0; i < n; ++i) cells_total += (double)q->len[pair_q[sel[i]]] * db->len[pair_t[sel[i]]];
    const CoopRule coop(cells_total, true);
    auto class_of = [&](uint32_t pi) {
        const int cc = (pair_score[pi] > 0 && !on_trace_stream) ? coop.cls(q->len[pair_q[pi]], db->len[pair_t[pi]]) : -1;
        return cc >= 0 ? cc : pair_class(q->len[pair_q[pi]]);
    };
    auto slot_of = [](int c) { return c >= kScanClasses ? c - kScanClasses : c + kCoopClasses; };      // position of class c in the order
    auto key_of = [&](uint32_t pi) {
        const uint32_t b = std::min<uint32_t>(db->len[pair_t[pi]] >> 5, kBuckets - 1);
        return (uint32_t)slot_of(class_of(pi)) * kBuckets + (kBuckets - 1 - b);
    };
    for (uint32_t i = 0; i < n; ++i) ++count[key_of(sel[i]) + 1];
    for (size_t k = 1; k < count.size(); ++k) count[k] += count[k - 1];
    uint32_t slot_begin[kAllClasses + 1];
    for (int c = 0; c <= kAllClasses; ++c) slot_begin[c] = count[(size_t)c * kBuckets];
    for (uint32_t i = 0; i < n; ++i) order[count[key_of(sel[i])]++] = sel[i];
    cx->stat_coop_trace = slot_begin[kCoopClasses];
    auto dir_bytes = [&](uint32_t pi) {
        const int c = class_of(pi);
        const uint64_t lt = db->len[pair_t[pi]], lq = q->len[pair_q[pi]];
        if (c == kPairClasses)          // banded: every band has the block of a full band of sw_pair_trace<16>
            return ((lq + kPairMaxRows - 1) / kPairMaxRows) * (((lt + 63) + 1) & ~1ull) * 64 * 16;
        if (c >= kScanClasses) {        // cooperative: a band of 64 * RPL rows per wave, each with the block of a full band
            const uint64_t rpl = (uint64_t)kCoopRpl[c - kScanClasses];
            return ((lq + 64 * rpl - 1) / (64 * rpl)) * (((lt + 63) + 1) & ~1ull) * 64 * 4;
        }
        const uint32_t rpl = (uint32_t)kPairRpl[c], nw = rpl <= 4 ? 1 : (rpl <= 8 ? 2 : 4);
        const uint64_t steps = lt + (lq + rpl - 1) / rpl - 1;
        return ((steps + 1) & ~1ull) * 64 * nw * 4;
    };
    // direction blocks of a batch: in the grow-only arena (a buffer of many GB made and freed per call costs a cold process a
    // quarter of a second in page tables alone); 10 000 pairs of 350 x 350 to a batch
    const uint64_t dir_budget = 2ull << 30;
    std::vector<PairTraceItem> items;
    for (uint32_t b0 = 0; b0 < n;) {
        // a batch: consecutive pairs (of any classes) whose direction blocks fit the budget; one launch per class in it, one download
        uint32_t b1 = b0;
        uint64_t dir_total = 0, path_total = 0;
        uint32_t max_lt_band = 0, max_lt_coop = 0;
        items.clear();
        while (b1 < n && (b1 == b0 || dir_total + dir_bytes(order[b1]) <= dir_budget)) {
            const uint32_t pi = order[b1];
            const int c = class_of(pi);
            if (c == kPairClasses) max_lt_band = std::max(max_lt_band, db->len[pair_t[pi]]);
            if (c >= kScanClasses) max_lt_coop = std::max(max_lt_coop, db->len[pair_t[pi]]);
            const uint64_t cap = (uint64_t)q->len[pair_q[pi]] + db->len[pair_t[pi]];
            path_total += cap;
            items.push_back({pair_q[pi], pair_t[pi], pair_score[pi], b1 - b0, dir_total, path_total});
            dir_total += dir_bytes(pi);
            ++b1;
        }
        const uint32_t ni = b1 - b0;
        const bool big_dirs = dir_total > dir_budget + (dir_budget >> 2);          // (a single pair beyond the budget: its own allocation)
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
        // items up and results back through pinned memory (pageable transfers are staged by the runtime: tens of microseconds per
        // call of a per-query traceback); the two regions do not overlap, the download is waited for before the next batch
        const uint64_t back_bytes = o_paths + path_total - o_out;
        const uint64_t pin_items = ((uint64_t)ni * sizeof(PairTraceItem) + 255) / 256 * 256;
        if (cx->trace_pin_bytes < pin_items + back_bytes) {
            if (cx->h_trace_pin) HIP_TRY(hipHostFree(cx->h_trace_pin));
            cx->h_trace_pin = nullptr; cx->trace_pin_bytes = 0;
            const uint64_t want = std::max<uint64_t>((pin_items + back_bytes) * 5 / 4, 1u << 20);
            HIP_TRY(hipHostMalloc((void**)&cx->h_trace_pin, want, hipHostMallocDefault));
            cx->trace_pin_bytes = want;
        }
        // a submission (job != nullptr) that is ONE batch stays in flight: it gets a pinned buffer of its own out of the device's pool
        const bool async = job && on_trace_stream && b0 == 0 && b1 == n && !big_dirs;
        char* pin = cx->h_trace_pin;
        if (async) {
            const uint64_t need_pin = (pin_items + back_bytes + 255) / 256 * 256;
            int pick = -1;
            for (size_t k = 0; k < cx->trace_pin_slabs.size() && pick < 0; ++k)
                if (cx->trace_pin_slabs[k].live > 0 && cx->trace_pin_slabs[k].cap - cx->trace_pin_slabs[k].used >= need_pin) pick = (int)k;
            for (size_t k = 0; k < cx->trace_pin_slabs.size() && pick < 0; ++k)
                if (cx->trace_pin_slabs[k].live == 0 && cx->trace_pin_slabs[k].cap >= need_pin) { cx->trace_pin_slabs[k].used = 0; pick = (int)k; }
            if (pick < 0) {
                DeviceCtx::PinSlab sl;
                sl.cap = std::max<uint64_t>(need_pin, 8u << 20);
                HIP_TRY(hipHostMalloc((void**)&sl.p, sl.cap, hipHostMallocDefault));
                cx->trace_pin_slabs.push_back(sl);
                pick = (int)cx->trace_pin_slabs.size() - 1;
            }
            DeviceCtx::PinSlab& sl = cx->trace_pin_slabs[(size_t)pick];
            job->pin = sl.p + sl.used; job->pin_slab = pick;
            sl.used += need_pin; ++sl.live;
            pin = job->pin;
        }
        memcpy(pin, items.data(), (size_t)ni * sizeof(PairTraceItem));
        uint8_t* const back = reinterpret_cast<uint8_t*>(pin) + pin_items;
        HIP_TRY(hipMemcpyAsync(A + o_items, pin, (size_t)ni * sizeof(PairTraceItem), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(cx->d_counter + kPairTraceCounter, 0, kAllClasses * 4, stream));
        CoopFork coop_fork(cx, stream);
        for (int sl = 0; sl < kAllClasses; ++sl) {
            // the pairs of class slot sl inside [b0, b1)
            const uint32_t lo = std::max(b0, slot_begin[sl]), hi = std::min(b1, slot_begin[sl + 1]);
            if (lo >= hi) continue;
            const int c = sl < kCoopClasses ? kScanClasses + sl : sl - kCoopClasses;
            PairTraceArgs a;
            a.items = reinterpret_cast<const PairTraceItem*>(A + o_items) + (lo - b0); a.seq_addr = db->d_seq_addr; a.seq_len = db->d_seq_len;
            a.q_codes = q->d_codes; a.q_off = q->d_off; a.q_len = q->d_len; a.mat = cx->d_mat;
            a.dirs = big_dirs ? d_dirs_big.p : reinterpret_cast<uint8_t*>(A + o_dirs); a.paths = reinterpret_cast<uint8_t*>(A + o_paths);
            a.out5 = reinterpret_cast<int32_t*>(A + o_out); a.best = reinterpret_cast<int32_t*>(A + o_best);
            a.counter = cx->d_counter + kPairTraceCounter + sl; a.n_pairs = hi - lo; a.open = (uint32_t)scorer->open; a.ext = (uint32_t)scorer->ext;
            a.left_first = traceback_left_first();
            if (c >= kScanClasses) {
                uint32_t* cache = &cx->grid_pair_trace_coop[c - kScanClasses];
                const void* fn = c - kScanClasses == 0 ? (const void*)sw_pair_trace_coop<2> : (c - kScanClasses == 1 ? (const void*)sw_pair_trace_coop<3> : (const void*)sw_pair_trace_coop<4>);
                if (*cache == 0) {
                    int per_cu = 0;
                    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kCoopWaves * 64, 0));
                    *cache = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
                }
                uint64_t slot = 0;
                uint32_t grid = 0;
                rc = ensure_coop_ws(cx, max_lt_coop, a.n_pairs, *cache, &grid, &slot);
                if (rc) return rc;
                rc = coop_fork.fork();
                if (rc) return rc;
                PairTraceCoopArgs ca;
                ca.p = a; ca.ws = reinterpret_cast<int2*>(cx->d_coop_ws); ca.ws_slot = slot;
                switch (c - kScanClasses) {
                    case 0: hipLaunchKernelGGL(sw_pair_trace_coop<2>, dim3(grid), dim3(kCoopWaves * 64), 0, cx->coop_stream, ca); break;
                    case 1: hipLaunchKernelGGL(sw_pair_trace_coop<3>, dim3(grid), dim3(kCoopWaves * 64), 0, cx->coop_stream, ca); break;
                    default: hipLaunchKernelGGL(sw_pair_trace_coop<4>, dim3(grid), dim3(kCoopWaves * 64), 0, cx->coop_stream, ca); break;
                }
                HIP_TRY(hipGetLastError());
            } else if (c == kPairClasses) {
                if (cx->grid_pair_trace_band == 0) {
                    int per_cu = 0;
                    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_pair_trace_band<16>, 64, 0));
                    cx->grid_pair_trace_band = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
                }
                uint64_t slot = 0;
                uint32_t grid = 0;
                rc = ensure_band_ws(cx, max_lt_band, a.n_pairs, cx->grid_pair_trace_band, &grid, &slot);
                if (rc) return rc;
                PairTraceBandArgs ba;
                ba.p = a; ba.band_ws = reinterpret_cast<int2*>(cx->d_band_ws); ba.band_ws_slot = slot;
                hipLaunchKernelGGL(sw_pair_trace_band<16>, dim3(grid), dim3(64), 0, stream, ba);
                HIP_TRY(hipGetLastError());
            } else {
                switch (c) {
                    case 0: rc = launch_pair_trace<2>(cx, a, &cx->grid_pair_trace[0], stream); break;
                    case 1: rc = launch_pair_trace<4>(cx, a, &cx->grid_pair_trace[1], stream); break;
                    case 2: rc = launch_pair_trace<6>(cx, a, &cx->grid_pair_trace[2], stream); break;
                    case 3: rc = launch_pair_trace<8>(cx, a, &cx->grid_pair_trace[3], stream); break;
                    case 4: rc = launch_pair_trace<12>(cx, a, &cx->grid_pair_trace[4], stream); break;
                    default: rc = launch_pair_trace<16>(cx, a, &cx->grid_pair_trace[5], stream); break;
                }
            }
            if (rc) return rc;
        }
        rc = coop_fork.join();
        if (rc) return rc;
        // out5, best and the paths sit back to back in the arena: one download
        HIP_TRY(hipMemcpyAsync(back, A + o_out, back_bytes, hipMemcpyDeviceToHost, stream));
        if (async) {
            if (!job->done) HIP_TRY(hipEventCreateWithFlags(&job->done, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(job->done, stream));
            HIP_TRY(hipEventRecord(cx->trace_last, stream));
            job->in_flight = true;
            job->db = db; db->refs.fetch_add(1);
            job->back_off = pin_items; job->best_off = pin_items + (o_best - o_out); job->paths_off = pin_items + (o_paths - o_out);
            job->order.assign(order.begin(), order.end());
            job->want.resize(ni); job->score.resize(ni); job->path_end.resize(ni);
            for (uint32_t x = 0; x < ni; ++x) {
                const uint32_t pi = order[x];
                job->score[x] = pair_score[pi];
                job->want[x] = pair_score[pi] > 0 && q->len[pair_q[pi]] && db->len[pair_t[pi]] ? pair_score[pi] : 0;
                job->path_end[x] = items[x].path_end;
            }
            return S4G_OK;
        }
        HIP_TRY(hipStreamSynchronize(stream));
        const int32_t* out5 = reinterpret_cast<const int32_t*>(back);
        const int32_t* best = reinterpret_cast<const int32_t*>(back + (o_best - o_out));
        const uint8_t* paths = back + (o_paths - o_out);
        for (uint32_t x = 0; x < ni; ++x) {
            const uint32_t pi = order[b0 + x];
            const int32_t want = pair_score[pi] > 0 && q->len[pair_q[pi]] && db->len[pair_t[pi]] ? pair_score[pi] : 0;
            if (best[x] != want && !(pair_score[pi] <= 0 && best[x] == 0))
                return fail(S4G_ERR_INVALID, "pair %u: score %d passed in, traceback kernel computed %d", pi, pair_score[pi], best[x]);
            memcpy(h_out + (size_t)pi * 5, out5 + (size_t)x * 5, 20);
            const uint32_t len = (uint32_t)out5[(size_t)x * 5 + 4];
            memcpy(h_paths + path_off[pi], paths + items[x].path_end - len, len);
        }
        b0 = b1;
    }
    return S4G_OK;
}

// ---- the traceback in two halves (include/s4g_hip.h) ----
S4G_API int s4g_traceback_submit(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t n_pairs,
                                 const uint32_t* pair_q, const uint32_t* pair_t, const int32_t* pair_score, const uint64_t* path_off,
                                 s4g_trace_job** out) {
    if (!db || !q || !scorer || !out) return fail(S4G_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_pairs && (!pair_q || !pair_t || !pair_score || !path_off)) return fail(S4G_ERR_INVALID, "NULL argument");
    if (db->device != q->device) return fail(S4G_ERR_INVALID, "db on device %d, queries on device %d", db->device, q->device);
    if (db->stream) return fail(S4G_ERR_INVALID, "a streamed database (s4g_db_open_streamed) serves s4g_scan_topk / s4g_scan_topk_host only");
    for (uint32_t i = 0; i < n_pairs; ++i) {
        if (pair_q[i] >= q->n || pair_t[i] >= db->n_seqs) return fail(S4G_ERR_INVALID, "pair %u out of range", i);
        if (q->len[pair_q[i]] >= 65535 || db->len[pair_t[i]] >= 65535)
            return fail(S4G_ERR_INVALID, "pair %u: sequences of 65535 residues or more are not supported by the device traceback", i);
        if (path_off[i + 1] - path_off[i] < (uint64_t)q->len[pair_q[i]] + db->len[pair_t[i]])
            return fail(S4G_ERR_INVALID, "pair %u: path capacity must be at least query length + target length", i);
    }
    s4g_trace_job* job = new s4g_trace_job();
    job->device = db->device; job->n_pairs = n_pairs;
    if (n_pairs) {
        job->path_off.assign(path_off, path_off + n_pairs + 1);
        job->out5.assign((size_t)n_pairs * 5, 0);
        job->paths.assign(path_off[n_pairs] ? path_off[n_pairs] : 1, 0);
        std::vector<uint32_t> all(n_pairs);
        std::iota(all.begin(), all.end(), 0u);
        const int rc = traceback_pairs(db, q, scorer, all, pair_q, pair_t, pair_score, job->out5.data(), job->path_off.data(), job->paths.data(), job);
        if (rc) {
            // the job's piece of a pinned slab goes back only once nothing queued can touch it any more: a copy or a kernel that was
            // queued before the failure still reads or writes it, and a slab whose live count reaches 0 is handed to the next job
            if (job->pin_slab >= 0) {
                DeviceCtx* cx = device_ctx(job->device);
                { DeviceGuard guard(job->device); if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError(); }
                std::lock_guard<std::mutex> lk(cx->mu); --cx->trace_pin_slabs[(size_t)job->pin_slab].live;
            }
            if (job->done) (void)hipEventDestroy(job->done);
            delete job;
            return rc;
        }
        if (job->in_flight) { std::vector<int32_t>().swap(job->out5); std::vector<uint8_t>().swap(job->paths); }
    }
    *out = job;
    return S4G_OK;
}

S4G_API int s4g_traceback_finish(s4g_trace_job* job, int32_t* h_out, uint8_t* h_paths) {
    if (!job) return fail(S4G_ERR_INVALID, "job is NULL");
    int rc = S4G_OK;
    if (job->n_pairs && (!h_out || !h_paths)) rc = fail(S4G_ERR_INVALID, "NULL argument");
    DeviceGuard guard(job->device);
    if (job->in_flight) {
        if (hipEventSynchronize(job->done) != hipSuccess) {
            (void)hipGetLastError();
            if (!rc) rc = fail(S4G_ERR_HIP, "waiting for the traceback failed");
            if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();     // nothing may still write the slab piece released below
        }
        if (!rc) {
            const int32_t* out5 = reinterpret_cast<const int32_t*>(job->pin + job->back_off);
            const int32_t* best = reinterpret_cast<const int32_t*>(job->pin + job->best_off);
            const uint8_t* paths = reinterpret_cast<const uint8_t*>(job->pin + job->paths_off);
            for (uint32_t x = 0; x < job->n_pairs && !rc; ++x) {
                const uint32_t pi = job->order[x];
                if (best[x] != job->want[x] && !(job->score[x] <= 0 && best[x] == 0)) {
                    rc = fail(S4G_ERR_INVALID, "pair %u: score %d passed in, traceback kernel computed %d", pi, job->score[x], best[x]);
                    break;
                }
                memcpy(h_out + (size_t)pi * 5, out5 + (size_t)x * 5, 20);
                const uint32_t len = (uint32_t)out5[(size_t)x * 5 + 4];
                memcpy(h_paths + job->path_off[pi], paths + job->path_end[x] - len, len);
            }
        }
        DeviceCtx* cx = device_ctx(job->device);
        std::lock_guard<std::mutex> lk(cx->mu);
        --cx->trace_pin_slabs[(size_t)job->pin_slab].live;
    } else if (!rc && job->n_pairs) {
        memcpy(h_out, job->out5.data(), (size_t)job->n_pairs * 20);
        for (uint32_t i = 0; i < job->n_pairs; ++i)
            memcpy(h_paths + job->path_off[i], job->paths.data() + job->path_off[i], (size_t)job->out5[(size_t)i * 5 + 4]);
    }
    if (job->done) (void)hipEventDestroy(job->done);
    if (job->db) (void)s4g_db_destroy(const_cast<s4g_db*>(job->db));       // the job's reference (the real destruction if the caller's is gone already)
    delete job;
    return rc;
}

// s4g_lists.hip -- candidate lists on the packed list kernel (sw_lists.hip.h): the host side of s4g_score_pairs_host's third
// route.  What the batched wrapper submits per chunk -- every query with its own few thousand prefilter candidates
// (/root/reference/sift4g/src/database_alignment.cpp:61-95,136-162 done for all queries at once) -- is split in two:
//   * pairs whose target is short enough for an inter-sequence launch to carry (and whose query fits the drifting-frame
//     window) are sorted by (query, length) on the device, cut into items of 128 candidates of one query, gathered into
//     lane-transposed blocks and scored by sw_scan_lists, two targets per register;
//   * the rest -- the long targets, a few percent of the pairs -- runs beside it on the one-wave-per-pair kernels.
// Both are exact (the window argument is sw_scan_pkd's); which pairs go where only changes the time.
#include "s4g_internal.hpp"
#include "sw_lists.hip.h"

#include <hipcub/hipcub.hpp>

namespace {

constexpr int kListRows = 32;                        // strip height of sw_scan_lists
constexpr uint64_t kListResSlack = 1024;             // dwords behind the gathered residues (an item of empty targets still reads one row)
// Rates the split is chosen with (tools/lists_bench.py on 70k .. 729k pairs, profiles/r02_lists): cells per second of a chip full
// of list items -- PADDED cells: a 128-candidate item runs to its longest member -- and of pair waves; the clocks per row and
// column the block that holds the launch's longest item needs (four waves pipelining its strips); the items it takes to fill
// the chip.
constexpr double kListChipRate = 9.0e12, kListWaveChipRate = 2.8e12;
constexpr double kListCyclesPerRowCol = 20.0, kListCyclesPerQueuedItem = 5.5, kClockHz = 2.4e9;
constexpr double kListItemsToFill = 700.0;
constexpr double kListFixedCost = 250e-6;            // sort + item tables + gather + one host round trip
constexpr uint32_t kListPairSlack = 64;              // two queries share an item (one per register half) when their lengths differ by no more than this
constexpr uint64_t kListResBoundBytes = 2ull << 30;  // gathered residues: up to this much is allocated by the worst case instead of asking the device
constexpr uint32_t kListMaxTarget = 4096;            // longer targets always go to the pair waves: the block workspace is 2 KB per residue
                                                     // of the longest member and block (6.4 GB at 4096 residues on 768 blocks)

struct ListBuffers {
    uint64_t *keys_in, *keys_out, *vals_in, *vals_out, *dwords, *offs;
    ListItem* items;
    int32_t* scores;
    void* cub;
    size_t cub_bytes;
};

inline uint64_t align256(uint64_t x) { return (x + 255) / 256 * 256; }

}  // namespace

// How long would the call take if pairs with targets up to `ls` residues went to the list kernel and the rest to pair waves?
//   cells_le[b] / pad_le[b] / pairs_le[b] = true cells, padding cells and pairs of all eligible pairs with target length
//   < 64 (b + 1); everything else goes to waves.  The padding estimate: a query's candidates sorted by length and cut into
//   items of 128 -- an item at length L spans 128 / (that query's candidates per residue of length around L), and its members
//   run half that span longer than they are, on average.
struct ListHist { std::vector<double> cells_le, pad_le, pairs_le; double cells_all = 0; uint32_t lq_max = 0, lt_max = 0; };
static double lists_cost(const ListHist& h, uint32_t ls) {
    const size_t b = std::min<size_t>(h.cells_le.size() - 1, ls / 64);
    const double c_list = ls == 0 ? 0.0 : h.cells_le[b];
    const double c_wave = h.cells_all - c_list;
    if (c_list <= 0) return c_wave / kListWaveChipRate;
    const double items = h.pairs_le[b] / 128.0;
    const double fill = std::min(1.0, items / kListItemsToFill);            // fewer items than blocks: part of the chip idles
    const double chip = (c_list + h.pad_le[b]) / (kListChipRate * fill) + c_wave / kListWaveChipRate;     // both parts share the chip
    // a launch cannot end before the block that holds its longest item has walked all rows x columns of it
    // ... at a rate that falls as the queue fills: with one item per block its four waves have their SIMDs nearly to themselves,
    // with seven they share them three ways for the whole launch (fitted: forced splits of 70k / 200k / 250k / 729k pairs)
    const double clocks = kListCyclesPerRowCol + kListCyclesPerQueuedItem * items / kListItemsToFill;
    const double lat_list = (double)h.lq_max * std::min<double>(ls, h.lt_max) * clocks / kClockHz;
    return std::max(chip, lat_list) + kListFixedCost;
}

int warm_list_kernels(DeviceCtx* cx) {
    int rc = device_props(cx);
    if (rc) return rc;
    if (cx->grid_lists == 0) {
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_scan_lists<kListRows>, kBlock, 0));
        cx->grid_lists = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
    }
    // the device sort and scan are library templates: run them once on a handful of elements
    DevBuf<uint64_t> buf;
    HIP_TRY(buf.alloc(4 * 64 + 8192));
    HIP_TRY(hipMemset(buf.p, 0, (4 * 64 + 8192) * 8));
    size_t tmp = 8192 * 8 - 512;
    uint64_t* p = buf.p;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(p + 256, tmp, p, p + 64, p + 128, p + 192, 64, 0, 40, nullptr));
    tmp = 8192 * 8 - 512;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(p + 256, tmp, p, p + 64, 64, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    if (cx->lst_stage_bytes < (16u << 20)) {                    // pinned staging for ~700k entries
        if (cx->h_lst_stage) HIP_TRY(hipHostFree(cx->h_lst_stage));
        cx->h_lst_stage = nullptr; cx->lst_stage_bytes = 0;
        HIP_TRY(hipHostMalloc((void**)&cx->h_lst_stage, 16u << 20, hipHostMallocDefault));
        cx->lst_stage_bytes = 16u << 20;
    }
    return S4G_OK;
}

// Scores of the pairs sel[] (indices into pair_q / pair_t; every query non-empty) into h_scores[sel[i]].
// force: every eligible pair goes to the list kernel whatever the estimate says (tests, measurements).
// *used = false: the estimate prefers the pair kernels for everything; nothing was done.
int score_lists_device(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, const uint32_t* sel_idx, uint32_t n_sel,
                       const uint32_t* pair_q, const uint32_t* pair_t, int32_t* h_scores, bool force, uint32_t forced_split, bool* used) {
    *used = false;
    if (n_sel == 0 || n_sel >= (1u << 31)) return S4G_OK;        // (the device sort counts its entries in an int)
    // the pairs of this call: sel_idx[0 .. n_sel), or 0 .. n_sel when sel_idx is NULL (every pair of the caller's arrays)
    struct Sel {
        const uint32_t* idx; uint32_t n;
        uint32_t operator[](uint32_t i) const { return idx ? idx[i] : i; }
        uint32_t size() const { return n; }
    } sel{sel_idx, n_sel};
    static const bool prof = getenv("S4G_PROFILE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1e3; };
    const auto t0 = now();
    double t_alloc = 0;          // S4G_PROFILE: time spent growing buffers (first call of a process)
    const bool gaps16 = scorer->open < 32768 && scorer->ext < 32768;
    if (!gaps16) return S4G_OK;
    ensure_bounds(q, scorer);
    const uint64_t cap = 0x7BFF;
    const uint64_t bias = (uint64_t)std::max(std::max(-kPadScore, -scorer->min_score), std::max(scorer->open, 2 * scorer->ext));

    // ---- which pairs may go to the list kernel at all: the drifting-frame window per query (as make_plan's drift tier) ----
    // bound + bias + columns * ext <= 0x7BFF with columns = 4 * len4 + 8 (sw_scan_lists runs the column frame of sw_scan_pkc)
    std::vector<uint32_t> lt_cap(q->n, 0);              // longest target query i may meet in the list kernel (0: none)
    for (uint32_t i = 0; i < q->n; ++i) {
        const uint64_t fixed = q->score_bound[i] + bias;
        if (fixed > cap) continue;
        uint64_t cols = scorer->ext ? (cap - fixed) / (uint64_t)scorer->ext : 0xFFFFFFFFull;
        if (scorer->ext && cols < 16) continue;
        if (scorer->ext) cols -= 12;                     // 8 + rounding of the length to 4 (column frame: the rows do not count)
        lt_cap[i] = (uint32_t)std::min<uint64_t>(cols, 0x7FFFFFFFu);
    }
    // ---- the split length: histogram of cells over target length (one pass), estimate for every 64-residue cut ----
    uint32_t lt_db = 0;
    for (const Tile& t : db->tiles) lt_db = std::max(lt_db, t.max_len4 * 4);
    const size_t n_buckets = (size_t)lt_db / 64 + 2;
    std::vector<double> cells_le(n_buckets, 0.0);
    std::vector<uint32_t> pairs_in(n_buckets, 0);                      // eligible pairs per 64-residue bucket of target length
    uint32_t lt_max = 0, lq_max = 0;
    double cells_all = 0, rows_all = 0;
    const unsigned parts = parallel_parts(n_sel, 1u << 16);
    // (a large call is estimated from every 8th pair: the estimate only picks the split length, any split gives the same scores;
    // the longest target of the call is then not known exactly -- lt_hi, what bounds the workspace, falls back to the database's)
    const uint32_t stride = n_sel >= (1u << 17) && !getenv("S4G_LISTS_FULL_HISTOGRAM") ? 8u : 1u;
    {
        struct Part { std::vector<double> cells; std::vector<uint32_t> pairs; uint32_t lt_max = 0, lq_max = 0; double cells_all = 0, rows_all = 0; };
        std::vector<Part> part(parts);
        parallel_for(n_sel, parts, [&](unsigned p, unsigned, uint64_t b, uint64_t e) {
            // locals, written back once: the Part structs sit side by side in one array (false sharing)
            std::vector<double> cells(n_buckets, 0.0);
            std::vector<uint32_t> pairs(n_buckets, 0);
            uint32_t my_lt = 0, my_lq = 0;
            double my_cells = 0, my_rows = 0;
            for (uint64_t i = (b + stride - 1) / stride * stride; i < e; i += stride) {
                const uint32_t pi = sel[(uint32_t)i];
                const uint32_t lt = db->len[pair_t[pi]], qi = pair_q[pi], lq = q->len[qi];
                const double c = (double)lq * lt * stride;
                my_lt = std::max(my_lt, lt);
                my_lq = std::max(my_lq, lq);
                my_cells += c;
                my_rows += (double)lq * stride;
                if (lt == 0 || lt > lt_cap[qi]) continue;
                cells[lt / 64] += c;
                pairs[lt / 64] += stride;
            }
            Part& P = part[p];
            P.cells.swap(cells); P.pairs.swap(pairs);
            P.lt_max = my_lt; P.lq_max = my_lq; P.cells_all = my_cells; P.rows_all = my_rows;
        });
        for (const Part& P : part) {
            lt_max = std::max(lt_max, P.lt_max); lq_max = std::max(lq_max, P.lq_max);
            cells_all += P.cells_all; rows_all += P.rows_all;
            for (size_t b = 0; b < n_buckets; ++b) { cells_le[b] += P.cells[b]; pairs_in[b] += P.pairs[b]; }
        }
    }
    if (stride > 1) for (uint32_t i = 0; i < q->n; ++i) lq_max = std::max(lq_max, q->len[i]);      // (an upper bound is all the estimate needs)
    const uint32_t lt_hi = stride > 1 ? lt_db : lt_max;
    ListHist hist;
    hist.cells_all = cells_all; hist.lq_max = lq_max; hist.lt_max = lt_max;
    hist.cells_le = cells_le; hist.pad_le.assign(n_buckets, 0.0); hist.pairs_le.assign(n_buckets, 0.0);
    {
        uint32_t nq_used = 0;
        for (uint32_t i = 0; i < q->n; ++i) nq_used += lt_cap[i] != 0;
        const double lq_mean = rows_all / (double)n_sel;
        for (size_t b = 0; b < n_buckets; ++b) {
            // span of an item around this length: 128 candidates at (pairs_in[b] / 64 residues / queries) candidates per residue
            const double span = pairs_in[b] ? std::min<double>(64.0 * b + 64.0, 128.0 * 64.0 * std::max(1u, nq_used) / (double)pairs_in[b]) : 0.0;
            hist.pad_le[b] = (double)pairs_in[b] * lq_mean * span * 0.5 + (b ? hist.pad_le[b - 1] : 0.0);
            hist.pairs_le[b] = (double)pairs_in[b] + (b ? hist.pairs_le[b - 1] : 0.0);
            if (b) hist.cells_le[b] += hist.cells_le[b - 1];
        }
    }
    uint32_t best_ls = 0;
    double best_t = lists_cost(hist, 0);
    const double all_waves = best_t;
    for (uint32_t ls = 127; ls < std::min(lt_hi, kListMaxTarget) + 64; ls += 64) {
        const double t = lists_cost(hist, ls);
        if (t < best_t) { best_t = t; best_ls = ls; }
    }
    if (force) best_ls = forced_split ? forced_split : std::max(lt_hi, 1u);      // everything the window admits (up to the caller's cut)
    else if (best_ls == 0 || best_t > 0.9 * all_waves) return S4G_OK;       // the pair kernels take it all
    best_ls = std::min(best_ls, kListMaxTarget);

    const auto t_h = now();
    *used = true;                           // (from here on the call is the list route's; a call without a single list entry is caught below)

    DeviceGuard guard(db->device);
    DeviceCtx* cx = device_ctx(db->device);
    std::lock_guard<std::mutex> lk(cx->mu);
    hipStream_t stream = nullptr;
    int rc = device_props(cx);
    if (rc) return rc;
    rc = ensure_seq_tables(db);
    if (rc) return rc;
    if (!cx->d_counter) HIP_TRY(hipMalloc((void**)&cx->d_counter, 256));
    ScratchScope scope(db, stream);
    rc = scope.acquire();
    if (rc) return rc;
    rc = ensure_mat(cx, scorer, stream);
    if (rc) return rc;
    rc = ensure_max3(cx);
    if (rc) return rc;
    if (!cx->side_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&cx->side_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&cx->side_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&cx->side_join, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(cx->side_fork, stream));            // where the pair waves of this call may start: before anything below is queued

    // ---- the device starts on the raw pairs while the host is still sorting them out ----
    // The entries (key = query, rank of the target in the database's length order; value = target, position in sel) are made on
    // the device: the host sends 8 bytes per pair instead of 16 and writes nothing per pair but a copy.  A pair that stays with
    // the pair waves gets the key (q->n, 0): the sort leaves those behind the last query's list.  Upload, entries and sort
    // (~0.6 ms for 729k pairs) run underneath the host's classification pass and the submission of the pair waves (~0.45 ms).
    const uint32_t max_items = n_sel / 64 + 2 * q->n + 2;               // (every unit ends with at most two short items)
    const uint64_t b_keys = align256((uint64_t)n_sel * 8), b_items = align256((uint64_t)max_items * sizeof(ListItem));
    const uint64_t b_cap = align256((uint64_t)q->n * 4), b_raw = align256((uint64_t)n_sel * 4);
    const uint64_t stage_bytes = b_cap + 2 * b_raw + b_items;          // lt_cap, pair queries, pair targets (first upload); items (second)
    { const auto ta = now();
    const uint64_t stage_need = std::max<uint64_t>(stage_bytes, 2 * (uint64_t)n_sel * 4 + 256);         // (the scores come back through it)
    if (cx->lst_stage_bytes < stage_need) {
        if (cx->h_lst_stage) HIP_TRY(hipHostFree(cx->h_lst_stage));
        cx->h_lst_stage = nullptr; cx->lst_stage_bytes = 0;
        const uint64_t want = stage_need + stage_need / 4;
        HIP_TRY(hipHostMalloc((void**)&cx->h_lst_stage, want, hipHostMallocDefault));
        cx->lst_stage_bytes = want;
    } else if (cx->lst_stage_done) HIP_TRY(hipEventSynchronize(cx->lst_stage_done));
    t_alloc += us(ta, now()); }
    if (!cx->lst_stage_done) HIP_TRY(hipEventCreateWithFlags(&cx->lst_stage_done, hipEventDisableTiming));
    uint32_t* h_cap = reinterpret_cast<uint32_t*>(cx->h_lst_stage);
    uint32_t* h_pq = reinterpret_cast<uint32_t*>(cx->h_lst_stage + b_cap);
    uint32_t* h_pt = reinterpret_cast<uint32_t*>(cx->h_lst_stage + b_cap + b_raw);
    ListItem* h_items = reinterpret_cast<ListItem*>(cx->h_lst_stage + b_cap + 2 * b_raw);
    memcpy(h_cap, lt_cap.data(), (size_t)q->n * 4);
    parallel_for(n_sel, parts, [&](unsigned, unsigned, uint64_t b, uint64_t e) {
        if (!sel.idx) {
            memcpy(h_pq + b, pair_q + b, (size_t)(e - b) * 4);
            memcpy(h_pt + b, pair_t + b, (size_t)(e - b) * 4);
        } else for (uint64_t i = b; i < e; ++i) { const uint32_t pi = sel.idx[i]; h_pq[i] = pair_q[pi]; h_pt[i] = pair_t[pi]; }
    });
    int q_bits = 1, pos_bits = 1;
    while (q_bits < 31 && (1ull << q_bits) <= q->n) ++q_bits;          // (the key q->n of the pairs that are not list entries included)
    while (pos_bits < 32 && (1ull << pos_bits) < db->n_seqs) ++pos_bits;        // the sort's passes go with the bits: no more of them than the key has
    size_t cub_sort = 0, cub_scan = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, cub_sort, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint64_t*)nullptr,
                                               (uint64_t*)nullptr, (int)n_sel, 0, pos_bits + q_bits, stream));
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, cub_scan, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)max_items, stream));
    const uint64_t b_cub = align256(std::max(cub_sort, cub_scan)), b_dw = align256((uint64_t)max_items * 8), b_sc = align256((uint64_t)n_sel * 4);
    const uint64_t need = stage_bytes + 4 * b_keys + 2 * b_dw + b_sc + b_cub;
    { const auto ta = now();
    if (cx->lst_bytes < need) {
        if (cx->d_lst) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(cx->d_lst)); }
        cx->d_lst = nullptr; cx->lst_bytes = 0;
        const uint64_t want = need + need / 4;
        HIP_TRY(hipMalloc((void**)&cx->d_lst, want));
        cx->lst_bytes = want;
    }
    if (!db->d_pos_of_id) {                 // the length rank of every sequence, for the entries' keys (once per database)
        HIP_TRY(hipMalloc((void**)&db->d_pos_of_id, (size_t)db->n_seqs * 4));
        HIP_TRY(hipMemcpy(db->d_pos_of_id, db->pos_of_id.data(), (size_t)db->n_seqs * 4, hipMemcpyHostToDevice));
    }
    t_alloc += us(ta, now()); }
    ListBuffers B;
    char* p = cx->d_lst;
    const uint32_t* d_cap = (const uint32_t*)p; p += b_cap;                            // the staging buffer's layout
    const uint32_t* d_pq = (const uint32_t*)p; p += b_raw;
    const uint32_t* d_pt = (const uint32_t*)p; p += b_raw;
    B.items = (ListItem*)p; p += b_items;
    B.keys_in = (uint64_t*)p; p += b_keys; B.vals_in = (uint64_t*)p; p += b_keys;
    B.keys_out = (uint64_t*)p; p += b_keys; B.vals_out = (uint64_t*)p; p += b_keys;
    B.dwords = (uint64_t*)p; p += b_dw; B.offs = (uint64_t*)p; p += b_dw;
    B.scores = (int32_t*)p; p += b_sc;
    B.cub = p; B.cub_bytes = b_cub;
    HIP_TRY(hipMemcpyAsync(cx->d_lst, cx->h_lst_stage, b_cap + 2 * b_raw, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(list_entries, dim3((n_sel + 255) / 256), dim3(256), 0, stream, d_pq, d_pt, n_sel, db->d_seq_len, db->d_pos_of_id, d_cap,
                       best_ls, q->n, (uint32_t)pos_bits, B.keys_in, B.vals_in);
    size_t tmp = B.cub_bytes;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(B.cub, tmp, B.keys_in, B.keys_out, B.vals_in, B.vals_out, (int)n_sel, 0, pos_bits + q_bits, stream));
    const auto t_c = now();

    // ---- host, meanwhile: which pairs go where, and per query how many list entries ----
    static thread_local std::vector<uint32_t> wave_sel;
    wave_sel.clear();
    std::vector<uint32_t> count(q->n + 1, 0);
    uint32_t n = 0;
    {
        std::vector<std::vector<uint32_t>> part_wave(parts), part_count(parts);
        std::vector<uint32_t> part_lists(parts, 0);
        parallel_for(n_sel, parts, [&](unsigned p, unsigned, uint64_t b, uint64_t e) {
            std::vector<uint32_t>& cnt = part_count[p];
            cnt.assign(q->n + 1, 0);
            uint32_t mine = 0;
            for (uint64_t i = b; i < e; ++i) {
                const uint32_t t = h_pt[i], qi = h_pq[i];
                const uint32_t lt = db->len[t];
                if (lt != 0 && lt <= best_ls && lt <= lt_cap[qi]) { ++mine; ++cnt[qi + 1]; } else part_wave[p].push_back(sel[(uint32_t)i]);
            }
            part_lists[p] = mine;
        });
        for (unsigned p = 0; p < parts; ++p) {
            n += part_lists[p];
            wave_sel.insert(wave_sel.end(), part_wave[p].begin(), part_wave[p].end());
            for (uint32_t i = 0; i <= q->n; ++i) count[i] += part_count[p][i];
        }
    }
    const auto t1 = now();
    static thread_local std::vector<uint32_t> wave_where;
    int32_t* d_wave_sc = nullptr;
    auto submit_waves = [&]() -> int {           // the long / ineligible pairs: pair waves on the side stream
        if (wave_sel.empty()) return S4G_OK;
        HIP_TRY(hipStreamWaitEvent(cx->side_stream, cx->side_fork, 0));
        scope.forked = true;
        const int rcw = pairs_enqueue(db, q, scorer, wave_sel, pair_q, pair_t, &wave_where, &d_wave_sc, cx->side_stream, /*pk_allowed=*/false);
        if (rcw) return rcw;
        HIP_TRY(hipEventRecord(cx->side_join, cx->side_stream));
        return S4G_OK;
    };
    if (n == 0) { rc = submit_waves(); if (rc) return rc; }
    const auto t1b = now();
    if (n == 0) {           // nothing for the list kernel after all (every pair beyond its query's window): the waves have it all
        HIP_TRY(hipEventRecord(cx->lst_stage_done, stream));
        if (!wave_sel.empty()) {
            HIP_TRY(hipStreamWaitEvent(stream, cx->side_join, 0));
            HIP_TRY(hipStreamSynchronize(stream));
            std::vector<int32_t> wsc(wave_sel.size());
            HIP_TRY(hipMemcpy(wsc.data(), d_wave_sc, wsc.size() * 4, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < wave_sel.size(); ++k) h_scores[wave_sel[k]] = wsc[wave_where[k]];
        }
        cx->stat_pair_items = wave_sel.size(); cx->stat_list_pairs = 0;
        return scope.release();
    }

    // ---- host: the item table ----
    // (Round 5 measured the other way to cut items: every block of 64 candidates of every query of a length class in ONE order by the
    // length of its longest member, neighbours sharing an item.  Against the true cells of the 729k-pair batch the rank-aligned items
    // below compute +10.1 %, that form +9.2 %, and blocks of 64 consecutive ranks cannot do better than +8.5 % (the spread of 64 ranks
    // of a length-sorted list, rows rounded to 8, columns to 4): the kernel got 0.3 % faster and the host 0.6 ms slower.  Dropped;
    // EXPERIMENTS.md, round 5.)
    // Units: two queries of nearly the same length walk their lists in step, 64 candidates each per item; a query without such a
    // partner is a unit of its own (items of 128).  Items of block k of every unit come before block k + 1 of any: the longest
    // members of every list start first.
    struct Unit { uint32_t a, b; };             // b == a: single
    std::vector<Unit> units;
    {
        std::vector<uint32_t> order;
        for (uint32_t i = 0; i < q->n; ++i) if (count[i + 1]) order.push_back(i);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return q->len[x] > q->len[y]; });
        for (size_t i = 0; i < order.size();) {
            if (i + 1 < order.size() && q->len[order[i]] - q->len[order[i + 1]] <= kListPairSlack) { units.push_back({order[i], order[i + 1]}); i += 2; }
            else { units.push_back({order[i], order[i]}); i += 1; }
        }
    }
    // items of a unit: while both lists last, 64 + 64; then (or for a single) 128 of the one that is left
    auto unit_items = [&](const Unit& u) -> uint32_t {
        const uint32_t ca = count[u.a + 1], cb = u.b != u.a ? count[u.b + 1] : 0;
        const uint32_t paired = std::min((ca + 63) / 64, (cb + 63) / 64);           // items with both halves filled (the last may be short)
        const uint32_t done = paired * 64;
        const uint32_t rest = (ca > done ? ca - done : 0) + (cb > done ? cb - done : 0);          // only one of the two is non-zero
        return paired + (rest + 127) / 128;
    };
    uint32_t n_items = 0, max_blocks = 0;
    for (const Unit& u : units) {
        const uint32_t blocks = unit_items(u);
        n_items += blocks;
        max_blocks = std::max(max_blocks, blocks);
    }
    if (n_items > max_items) return fail(S4G_ERR_INVALID, "internal: %u list items, expected at most %u", n_items, max_items);
    for (uint32_t i = 0; i < q->n; ++i) count[i + 1] += count[i];               // segment offsets in the sorted order
    {
        uint32_t at = 0;
        for (uint32_t blk = 0; blk < max_blocks; ++blk)
            for (const Unit& u : units) {
                const uint32_t ca = count[u.a + 1] - count[u.a], cb = u.b != u.a ? count[u.b + 1] - count[u.b] : 0;
                const uint32_t paired = std::min((ca + 63) / 64, (cb + 63) / 64);
                ListItem it;
                it.len4 = 0; it.res_off = 0;
                if (blk < paired) {
                    it.qa = u.a; it.qb = u.b;
                    it.first_a = count[u.a] + blk * 64; it.first_b = count[u.b] + blk * 64;
                    it.na = (uint16_t)std::min(64u, ca - blk * 64); it.nb = (uint16_t)std::min(64u, cb - blk * 64);
                } else {
                    const uint32_t done = paired * 64;
                    const bool from_a = ca > done;
                    const uint32_t qx = from_a ? u.a : u.b, cx_ = from_a ? ca : cb;
                    const uint32_t off = done + (blk - paired) * 128;
                    if (cx_ <= off) continue;
                    it.qa = it.qb = qx;
                    it.first_a = count[qx] + off; it.first_b = it.first_a + 64;
                    it.na = (uint16_t)std::min(64u, cx_ - off); it.nb = (uint16_t)(cx_ - off > 64 ? std::min(64u, cx_ - off - 64) : 0);
                }
                h_items[at++] = it;
            }
    }
    HIP_TRY(hipMemcpyAsync(B.items, h_items, (size_t)n_items * sizeof(ListItem), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(cx->lst_stage_done, stream));
    const auto t2 = now();
    const uint32_t max_len4 = (std::min(best_ls, lt_hi) + 3) / 4;         // no member is longer than the split length
    // The gathered residues: an item is at most max_len4 chunks of 128 dwords.  While that bound is affordable every item gets a
    // slot of that size -- no prefix sum over the items' sizes (a decoupled-lookback scan: its blocks wait for each other, and
    // next to the pair waves of the side stream one launch of it was seen to take 0.6 ms) and no round trip to the host for the
    // total; beyond the bound: sizes, scan, offsets, and one round trip for the sum.
    const uint64_t slot_dwords = (uint64_t)max_len4 * 128;
    uint64_t res_dwords = (uint64_t)n_items * slot_dwords;
    const bool slots = res_dwords * 4 <= kListResBoundBytes;
    hipLaunchKernelGGL(list_item_len, dim3((n_items + 255) / 256), dim3(256), 0, stream, B.items, n_items, B.vals_out, db->d_seq_len, B.dwords,
                       slots ? slot_dwords : (uint64_t)0);
    const auto ta2 = now();
    if (!slots) {
        tmp = B.cub_bytes;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(B.cub, tmp, B.dwords, B.offs, (int)n_items, stream));
        hipLaunchKernelGGL(list_item_off, dim3((n_items + 255) / 256), dim3(256), 0, stream, B.items, n_items, B.offs);
        uint64_t tail[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(&tail[0], B.offs + (n_items - 1), 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(&tail[1], B.dwords + (n_items - 1), 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        res_dwords = tail[0] + tail[1];
    }
    HIP_TRY(hipGetLastError());
    if (cx->lst_res_bytes < (res_dwords + kListResSlack) * 4) {
        if (cx->d_lst_res) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(cx->d_lst_res)); }
        cx->d_lst_res = nullptr; cx->lst_res_bytes = 0;
        const uint64_t want = (res_dwords + kListResSlack) * 4 * 5 / 4;
        HIP_TRY(hipMalloc((void**)&cx->d_lst_res, want));
        cx->lst_res_bytes = want;
    }
    if (cx->grid_lists == 0) {
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_scan_lists<kListRows>, kBlock, 0));
        cx->grid_lists = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
    }
    const uint32_t grid = std::max(1u, std::min(cx->grid_lists, n_items));         // a block works on one item at a time (4 in flight)
    const uint64_t ws_slot = 2ull * std::max(1u, max_len4) * 64 * 4;          // int32 per buffer: H and F rows, 16 B per lane and chunk
    const uint64_t ws_bytes = ws_slot * 4 * kListBufs * grid;
    if (cx->ws_bytes < ws_bytes) {
        if (cx->d_ws) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(cx->d_ws)); }
        cx->d_ws = nullptr; cx->ws_bytes = 0;
        HIP_TRY(hipMalloc((void**)&cx->d_ws, ws_bytes));
        cx->ws_bytes = ws_bytes;
    }
    t_alloc += us(ta2, now());
    hipLaunchKernelGGL(list_gather, dim3(n_items), dim3(256), 0, stream, B.items, B.vals_out, db->d_seq_addr, db->d_seq_len, cx->d_lst_res);
    // the pair waves go out while the device gathers -- and before the list kernel's persistent blocks take every wave slot
    rc = submit_waves();            // (A/B against submitting them before the item table: 9.96 against 10.09 ms per call, mean of 4 and 3;
                                    // raised wave priority for the pair kernels, so that they do not crawl underneath the list kernel's
                                    // blocks: 10.65 against 10.38 ms -- the pipeline's waves stall behind them)
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(cx->d_counter + kListCounter, 0, 4, stream));
    ListArgs a;
    a.items = B.items; a.res = cx->d_lst_res; a.entries = B.vals_out;
    a.q_codes = q->d_codes; a.q_off = q->d_off; a.q_len = q->d_len; a.mat = cx->d_mat;
    a.scores = B.scores; a.ws = cx->d_ws; a.ws_slot = ws_slot; a.counter = cx->d_counter + kListCounter; a.n_items = n_items;
    a.open = (uint32_t)scorer->open; a.ext = (uint32_t)scorer->ext; a.floor_bits = (uint32_t)(bias | (bias << 16));
    a.grp_perm = nullptr; a.grp_off = nullptr; a.grp_len4 = nullptr; a.scan_scores = nullptr; a.score_stride = 0; a.q_begin = 0; a.spec_base = 0;       // (scan-mode fields)
    {   // Critical-path items run at raised wave priority (the SIMD arbitrates issue by priority, then age).  A launch of
        // lists has only about two items per resident wave, so "large against the wave's share" would mean nearly all of
        // them; instead: the longest items, as many as there are SIMDs to own (their count is estimated from the length
        // histogram: 128 pairs to an item).
        const uint64_t simds = (uint64_t)cx->prop.multiProcessorCount * 4;
        uint64_t pairs_above = 0;
        uint32_t lp = std::min(best_ls, lt_hi) + 1;
        for (size_t b = std::min<size_t>(pairs_in.size() - 1, std::min(best_ls, lt_hi) / 64);; --b) {
            if (pairs_above + pairs_in[b] > simds * 3 / 4 * 128) break;
            pairs_above += pairs_in[b];
            lp = (uint32_t)b * 64;
            if (b == 0) break;
        }
        const double lq_mean = rows_all / (double)n_sel;
        a.prio_cost = (uint32_t)std::min<double>(4e9, std::max<double>(1.0, lq_mean * ((lp + 3) / 4)));
    }
    hipLaunchKernelGGL(sw_scan_lists<kListRows>, dim3(grid), dim3(kBlock), 0, stream, a);
    HIP_TRY(hipGetLastError());
    // results come back into the pinned staging buffer (its upload has long completed): a download into pageable memory makes
    // the runtime pin the destination first, which costs milliseconds the first time a process does it
    HIP_TRY(hipEventSynchronize(cx->lst_stage_done));
    int32_t* sc = reinterpret_cast<int32_t*>(cx->h_lst_stage);        // [n_sel]: by position in sel (entries of wave pairs stay unwritten)
    int32_t* wsc = sc + n_sel;
    HIP_TRY(hipMemcpyAsync(sc, B.scores, (size_t)n_sel * 4, hipMemcpyDeviceToHost, stream));
    if (!wave_sel.empty()) {
        HIP_TRY(hipStreamWaitEvent(stream, cx->side_join, 0));
        HIP_TRY(hipMemcpyAsync(wsc, d_wave_sc, wave_sel.size() * 4, hipMemcpyDeviceToHost, stream));
    }
    const auto t3 = now();
    HIP_TRY(hipStreamSynchronize(stream));
    parallel_for(n_sel, parts, [&](unsigned, unsigned, uint64_t b, uint64_t e) {
        if (!sel_idx) memcpy(h_scores + b, sc + b, (size_t)(e - b) * 4);
        else for (uint64_t i = b; i < e; ++i) h_scores[sel_idx[i]] = sc[i];
    });
    for (size_t k = 0; k < wave_sel.size(); ++k) h_scores[wave_sel[k]] = wsc[wave_where[k]];
    cx->stat_pair_items = wave_sel.size();          // s4g_scan_stats: what went to the pair waves / the list kernel
    cx->stat_list_pairs = n;
    if (prof) {
        const auto t4 = now();
        fprintf(stderr, "[s4g profile] lists n=%u (+%zu as waves), split at %u residues, %u items: estimate %.1f us (every %u. pair), raw pairs to the device + entries + sort queued %.1f us, "
                        "classification meanwhile %.1f us (%u threads), item table %.1f us, tables + gather + pair waves + list kernel queued %.1f us, wait+download %.1f us; of which growing buffers %.1f us "
                        "(ws %.0f MB, residues %.0f MB)\n",
                n, wave_sel.size(), best_ls, n_items, us(t0, t_h), stride, us(t_h, t_c), us(t_c, t1), parts, us(t1b, t2), us(t2, t3), us(t3, t4), t_alloc, ws_bytes / 1e6, res_dwords * 4 / 1e6);
    }
    return scope.release();
}


// ---------------------------------------------------------------------------------------------------------------------
// A database tile scanned by a FEW queries on the block-cooperative strip pipeline (sw_scan_lists<R, true>).
// The scan kernels give a (group, query pair) item to one wave, and a launch cannot end before the wave that holds the longest
// group has walked its whole DP (~37 clocks per cell: 16 ms for 350 rows x 3000 columns however idle the chip is); with a
// thousand queries that tail is nothing, with four it is the launch.  Round 1 answered with a cost model that hands the longest
// groups to pair waves (split_groups); here the same items go through the list kernel's pipeline instead -- the four waves of a
// block take the strips of ONE item round-robin, each trailing its predecessor by a 4-column chunk -- so an item's latency is
// its strips / 4 and no model is needed.  A scan item is a list item whose two halves read the SAME target, each half with a
// query of its own: residues straight out of the tile (no gather), scores straight into the caller's matrix.
// *used = false: not applicable (a query outside the drifting-frame window, a tile of very long sequences); nothing was launched.
int scan_tile_pipelined(const s4g_db* db, const Tile& t, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* scorer,
                        int32_t* d_scores, uint64_t stride, const uint32_t* d_perm, hipStream_t stream, bool* used, bool allow_spec, bool* speculative) {
    *used = false;
    if (speculative) *speculative = false;
    if (!q_count || !t.n_groups || t.h_grp_off.size() < t.n_groups) return S4G_OK;
    if (!(scorer->open < 32768 && scorer->ext < 32768) || (uint64_t)t.max_len4 * 4 > kListMaxTarget) return S4G_OK;
    ensure_bounds(q, scorer);
    const uint64_t cap = 0x7BFF;
    const uint64_t bias = (uint64_t)std::max(std::max(-kPadScore, -scorer->min_score), std::max(scorer->open, 2 * scorer->ext));
    // query pairs: neighbours in length order (an item runs to the longer one)
    std::vector<uint32_t> order(q_count);
    for (uint32_t i = 0; i < q_count; ++i) order[i] = q_begin + i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return q->len[a] > q->len[b]; });
    const uint64_t cols = (uint64_t)t.max_len4 * 4 + 8;
    // A query whose score BOUND leaves the window still runs here when its FRAME fits with room to spare (long queries: the bound grows
    // by ~5.6 per residue, real scores rarely do) -- speculatively, as in sw_scan_pkc's tier (s4g_scan.hip make_plan, the same
    // criterion): the kernel marks what it cannot vouch for and the caller's recheck_marked recomputes those entries in int32.
    const uint64_t step_c = (uint64_t)std::max(0, scorer->max_score) + (uint64_t)scorer->ext;
    bool spec = false;
    for (uint32_t i = 0; i < q_count; ++i) {
        if (q->len[order[i]] == 0) return S4G_OK;
        if (q->score_bound[order[i]] + bias + cols * (uint64_t)scorer->ext <= cap) continue;
        if (!allow_spec || !speculative || bias + step_c + cols * (uint64_t)scorer->ext + kSpecSpan > cap) return S4G_OK;
        spec = true;
    }
    // Items: every group of the tile under every PAIR of queries (neighbours in length order).  An odd query count leaves the shortest
    // query without a partner; it takes two neighbouring GROUPS per item instead (sw_scan_lists<R, 2>: the halves of a register read
    // different targets), so a lone query fills both halves of its registers instead of computing every cell twice.  That pays where the
    // call is bound by throughput (one query against 570 000 targets: 12.3 -> 6.x ms); a call bound by its longest item's latency (one
    // query against 100 000) takes the same time either way.
    const bool odd = (q_count & 1u) != 0 && t.d_grp_off != nullptr && t.d_grp_len4 != nullptr && getenv("S4G_SCAN_PIPE_NO_GROUP_PAIRS") == nullptr;
    const uint32_t n_pairs = odd ? q_count / 2 : (q_count + 1) / 2;             // pairs of two queries (without the lone one when it pairs groups)
    const uint64_t n_items64 = (uint64_t)t.n_groups * n_pairs + (odd ? ((uint64_t)t.n_groups + 1) / 2 : 0);
    if (n_items64 > (1u << 22)) return S4G_OK;
    const uint32_t n_items = (uint32_t)n_items64;

    DeviceCtx* cx = device_ctx(db->device);
    int rc = device_props(cx);
    if (rc) return rc;
    rc = ensure_max3(cx);
    if (rc) return rc;
    const uint64_t b_items = align256((uint64_t)n_items * sizeof(ListItem));
    if (cx->lst_stage_bytes < b_items) {
        if (cx->lst_stage_done) HIP_TRY(hipEventSynchronize(cx->lst_stage_done));
        if (cx->h_lst_stage) HIP_TRY(hipHostFree(cx->h_lst_stage));
        cx->h_lst_stage = nullptr; cx->lst_stage_bytes = 0;
        const uint64_t want = std::max<uint64_t>(b_items + b_items / 4, 1u << 20);
        HIP_TRY(hipHostMalloc((void**)&cx->h_lst_stage, want, hipHostMallocDefault));
        cx->lst_stage_bytes = want;
    } else if (cx->lst_stage_done) HIP_TRY(hipEventSynchronize(cx->lst_stage_done));
    if (!cx->lst_stage_done) HIP_TRY(hipEventCreateWithFlags(&cx->lst_stage_done, hipEventDisableTiming));
    if (cx->lst_bytes < b_items) {
        if (cx->d_lst) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(cx->d_lst)); }
        cx->d_lst = nullptr; cx->lst_bytes = 0;
        const uint64_t want = std::max<uint64_t>(b_items + b_items / 4, 1u << 20);
        HIP_TRY(hipMalloc((void**)&cx->d_lst, want));
        cx->lst_bytes = want;
    }
    ListItem* h_items = reinterpret_cast<ListItem*>(cx->h_lst_stage);
    uint32_t at = 0;
    for (uint32_t g = 0; g < t.n_groups; ++g) {                 // groups are in length order: the longest items start first
        const uint64_t end = g + 1 < t.n_groups ? t.h_grp_off[g + 1] : t.padded_residues / 4;
        const uint32_t len4 = (uint32_t)((end - t.h_grp_off[g]) / 64);
        for (uint32_t p = 0; p < n_pairs; ++p) {
            ListItem it;
            it.qa = order[2 * p]; it.qb = 2 * p + 1 < q_count ? order[2 * p + 1] : order[2 * p];
            it.first_a = g; it.first_b = odd ? g : 0; it.na = 64; it.nb = 64; it.len4 = len4; it.res_off = t.h_grp_off[g];
            h_items[at++] = it;
        }
        if (odd && (g & 1u) == 0) {                               // the query without a partner on groups g and g + 1 (g alone at the end)
            ListItem it;
            it.qa = it.qb = order[q_count - 1];
            it.first_a = g; it.first_b = g + 1 < t.n_groups ? g + 1 : g; it.na = 64; it.nb = 64; it.len4 = len4; it.res_off = t.h_grp_off[g];
            h_items[at++] = it;
        }
    }
    if (cx->grid_pipe == 0) {
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_scan_lists<kListRows, 1>, kBlock, 0));
        cx->grid_pipe = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
    }
    // Twelve waves to a block (round 6) where the call is bound by the LATENCY of its longest item, not by throughput: long queries
    // (>= 1200 rows, >= 500 for one or two queries: tens to hundreds of strips per item) whose longest item, walked by four waves, outlasts by 1.5 x what a block's
    // share of ALL items amounts to.  One block to a CU then, three waves per SIMD: an item's strips go through the issue slots of a
    // whole CU instead of one wave per SIMD next to two other blocks' waves.  (Short queries keep four waves: twelve were measured in
    // round 3 on 350-row queries -- 3.8 against 5.2 ms for the longest item alone, but 8.9 against 10.9 TCUPS once the chip is full.)
    // S4G_SCAN_PIPE_WAVES=4|12 forces one form (tests, A/B).
    bool wide = false;
    {
        uint32_t rows_max = 0;
        uint64_t rows_sum = 0;                                    // over the query pairs: the longer member
        for (uint32_t p = 0; p < n_pairs; ++p) { rows_max = std::max(rows_max, q->len[order[2 * p]]); rows_sum += q->len[order[2 * p]]; }
        if (odd) { rows_max = std::max(rows_max, q->len[order[q_count - 1]]); rows_sum += (q->len[order[q_count - 1]] + 1) / 2; }   // half as many items
        const double longest = (double)rows_max * t.max_len4;
        const double share = (double)rows_sum * ((double)t.padded_residues / 256.0) / (double)std::max(1u, std::min(cx->grid_pipe, n_items));
        wide = rows_max >= (q_count <= 2 ? 500u : 1200u) && longest > 1.5 * share;     // (one or two queries: measured from 500 rows, s4g_scan.hip)
        if (const char* e = getenv("S4G_SCAN_PIPE_WAVES")) wide = atoi(e) == 12 ? true : (atoi(e) == 4 ? false : wide);
    }
    if (wide && cx->grid_pipe_wide == 0) {
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_scan_lists<kListRows, 1, 12>, 12 * 64, 0));
        cx->grid_pipe_wide = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
    }
    if (odd && cx->grid_pipe2 == 0) {
        int per_cu = 0, per_cu_w = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sw_scan_lists<kListRows, 2>, kBlock, 0));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_w, (const void*)sw_scan_lists<kListRows, 2, 12>, 12 * 64, 0));
        cx->grid_pipe2 = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu);
        cx->grid_pipe2_wide = (uint32_t)cx->prop.multiProcessorCount * (uint32_t)std::max(1, per_cu_w);
    }
    const uint32_t grid = std::max(1u, std::min(odd ? (wide ? cx->grid_pipe2_wide : cx->grid_pipe2) : (wide ? cx->grid_pipe_wide : cx->grid_pipe), n_items));
    const uint64_t ws_slot = 2ull * std::max(1u, t.max_len4) * 64 * 4;
    const uint64_t ws_bytes = ws_slot * 4 * kListBufs * grid;
    if (cx->ws_bytes < ws_bytes) {
        if (cx->d_ws) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(cx->d_ws)); }
        cx->d_ws = nullptr; cx->ws_bytes = 0;
        HIP_TRY(hipMalloc((void**)&cx->d_ws, ws_bytes));
        cx->ws_bytes = ws_bytes;
    }
    HIP_TRY(hipMemcpyAsync(cx->d_lst, h_items, (size_t)n_items * sizeof(ListItem), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(cx->lst_stage_done, stream));
    HIP_TRY(hipMemsetAsync(cx->d_counter + kListCounter, 0, 4, stream));
    ListArgs a;
    a.items = reinterpret_cast<const ListItem*>(cx->d_lst); a.res = t.d_res; a.entries = nullptr;
    a.q_codes = q->d_codes; a.q_off = q->d_off; a.q_len = q->d_len; a.mat = cx->d_mat;
    a.scores = nullptr; a.ws = cx->d_ws; a.ws_slot = ws_slot; a.counter = cx->d_counter + kListCounter; a.n_items = n_items;
    a.open = (uint32_t)scorer->open; a.ext = (uint32_t)scorer->ext; a.floor_bits = (uint32_t)(bias | (bias << 16));
    a.prio_cost = 0;        // (raised wave priority for the longest items was measured: no change -- their latency is their own strips', not their neighbours')
    a.grp_perm = d_perm; a.scan_scores = d_scores; a.score_stride = stride; a.q_begin = q_begin;
    a.spec_base = spec ? (uint32_t)(cap - step_c) : 0u;
    a.grp_off = t.d_grp_off; a.grp_len4 = t.d_grp_len4;
    if (odd) {
        if (wide) hipLaunchKernelGGL((sw_scan_lists<kListRows, 2, 12>), dim3(grid), dim3(12 * 64), 0, stream, a);
        else hipLaunchKernelGGL((sw_scan_lists<kListRows, 2>), dim3(grid), dim3(kBlock), 0, stream, a);
    } else {
        if (wide) hipLaunchKernelGGL((sw_scan_lists<kListRows, 1, 12>), dim3(grid), dim3(12 * 64), 0, stream, a);
        else hipLaunchKernelGGL((sw_scan_lists<kListRows, 1>), dim3(grid), dim3(kBlock), 0, stream, a);
    }
    HIP_TRY(hipGetLastError());
    *used = true;
    if (speculative) *speculative = spec;
    return S4G_OK;
}

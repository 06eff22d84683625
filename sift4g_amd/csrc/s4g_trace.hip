// s4g_trace.hip -- C ABI: traceback of the survivors (step 4 of alignDatabase): routing between one wave per pair
// (s4g_pairs.hip) and the 64-pairs-per-wave strip kernels (sw_trace.hip.h), and the latter's host side.
#include "s4g_internal.hpp"
#include "sw_trace.hip.h"

int traceback_groups(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t n_pairs,
                     const uint32_t* pair_q, const uint32_t* pair_t, const int32_t* pair_score,
                     int32_t* h_out, const uint64_t* path_off, uint8_t* h_paths);


S4G_API int s4g_traceback_host_flags(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t n_pairs,
                                     const uint32_t* pair_q, const uint32_t* pair_t, const int32_t* pair_score,
                                     int32_t* h_out, const uint64_t* path_off, uint8_t* h_paths, uint32_t flags) {
    if (!db || !q || !scorer) return fail(S4G_ERR_INVALID, "NULL argument");
    if (n_pairs == 0) return S4G_OK;
    if (!pair_q || !pair_t || !pair_score || !h_out || !path_off || !h_paths) return fail(S4G_ERR_INVALID, "NULL argument");
    if (db->device != q->device) return fail(S4G_ERR_INVALID, "db on device %d, queries on device %d", db->device, q->device);
    if ((flags & S4G_PAIRS_FORCE_SCAN) && (flags & S4G_PAIRS_FORCE_WAVE)) return fail(S4G_ERR_INVALID, "contradictory flags");
    if (db->stream) return fail(S4G_ERR_INVALID, "a streamed database (s4g_db_open_streamed) serves s4g_scan_topk / s4g_scan_topk_host only");
    if (db->cached) flags = S4G_PAIRS_FORCE_WAVE;        // a database of cached chains has no tiles for the strip kernels
    for (uint32_t i = 0; i < n_pairs; ++i) {
        if (pair_q[i] >= q->n || pair_t[i] >= db->n_seqs) return fail(S4G_ERR_INVALID, "pair %u out of range", i);
        if (q->len[pair_q[i]] >= 65535 || db->len[pair_t[i]] >= 65535)
            return fail(S4G_ERR_INVALID, "pair %u: sequences of 65535 residues or more are not supported by the device traceback", i);
        if (path_off[i + 1] - path_off[i] < (uint64_t)q->len[pair_q[i]] + db->len[pair_t[i]])
            return fail(S4G_ERR_INVALID, "pair %u: path capacity must be at least query length + target length", i);
    }
    // one wave per pair (sw_pair_trace, banded beyond 1024 query residues) unless the strip kernels are forced
    // (sw_trace_fill + sw_trace_walk, 64 pairs per wave: the throughput form, kept as an independent implementation)
    if (flags & S4G_PAIRS_FORCE_SCAN) return traceback_groups(db, q, scorer, n_pairs, pair_q, pair_t, pair_score, h_out, path_off, h_paths);
    std::vector<uint32_t> all(n_pairs);
    std::iota(all.begin(), all.end(), 0u);
    return traceback_pairs(db, q, scorer, all, pair_q, pair_t, pair_score, h_out, path_off, h_paths);
}

S4G_API int s4g_traceback_host(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t n_pairs,
                               const uint32_t* pair_q, const uint32_t* pair_t, const int32_t* pair_score,
                               int32_t* h_out, const uint64_t* path_off, uint8_t* h_paths) {
    return s4g_traceback_host_flags(db, q, scorer, n_pairs, pair_q, pair_t, pair_score, h_out, path_off, h_paths, 0);
}

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
    { const int rcm = ensure_mat(cx, scorer, stream); if (rcm) return rcm; }
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
        a.left_first = traceback_left_first();
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

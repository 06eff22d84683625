// s4g_chains.hip -- C ABI: candidate databases out of chains the device already holds (s4g_db_create_cached).
//
// sift4g's wrapper builds one ChainDatabase per (query, chunk) out of chains it already holds -- createFilteredDatabase then
// chainDatabaseCreate, /root/reference/sift4g/src/database_alignment.cpp:70-81 -- and deletes it after one alignDatabase call
// (:92).  A chunk's chains are candidates of many queries (200 queries x 3645 candidates of 50 000 targets: every chain is
// sent 14 times), so packing and uploading the residues per call is what the call mostly consists of.  Here every chain goes
// to the device ONCE: its residues are appended to a per-device arena in the layout the one-wave-per-pair kernels read
// (sw_pair.hip.h: 64 chains to a group, dword c of a chain 64 dwords after dword c - 1), and where they went is written into
// two cache words the CALLER keeps with the chain (libswsharp: inside the Chain object, so a hit costs no lookup).  A
// candidate database is then two small device tables -- address and length per candidate -- and serves the pair routes only
// (scores: s4g_score_pairs_host / s4g_score_lists_host; traceback: s4g_traceback_host), which is what a call with one query
// and a few thousand candidates runs on anyway.
#include "s4g_internal.hpp"

namespace {

constexpr uint64_t kArenaBlock = 64ull << 20;        // arena allocations
constexpr uint64_t kArenaCap = 16ull << 30;          // beyond this the arena starts over (once no database points into it)

struct ChainArena {
    uint64_t gen = 1;                                // generation: a tag of another generation is a miss
    std::vector<std::pair<char*, uint64_t>> blocks;
    char* cur = nullptr;
    uint64_t cur_cap = 0, cur_used = 0, bytes = 0;
    uint64_t chains = 0, hits = 0, misses = 0, databases = 0;
    uint32_t live = 0;                               // databases that point into the arena
    char* h_stage = nullptr;                         // pinned: packed groups of a call's new chains, then its tables
    uint64_t stage_cap = 0;
    hipEvent_t stage_done = nullptr;
};
ChainArena g_arena[64];

inline uint64_t arena_tag(const ChainArena& a, int device) { return (a.gen << 6) | (uint64_t)(device & 63); }

int stage_reserve(ChainArena& a, uint64_t bytes) {
    if (a.stage_done) HIP_TRY(hipEventSynchronize(a.stage_done));       // the previous call's uploads out of it
    else HIP_TRY(hipEventCreateWithFlags(&a.stage_done, hipEventDisableTiming));
    if (a.stage_cap < bytes) {
        if (a.h_stage) HIP_TRY(hipHostFree(a.h_stage));
        a.h_stage = nullptr; a.stage_cap = 0;
        const uint64_t want = std::max<uint64_t>(bytes + bytes / 2, 4u << 20);
        HIP_TRY(hipHostMalloc((void**)&a.h_stage, want, hipHostMallocDefault));
        a.stage_cap = want;
    }
    return S4G_OK;
}

}  // namespace

// tags / addrs: one pair of words per sequence, owned by the caller and kept WITH the sequence: zero when the sequence is
// created, passed back unchanged on every later call.  On return every sequence's words say where its residues live on
// `device`.  The caller guarantees that words are never copied from one sequence to a different one.
S4G_API int s4g_db_create_cached(s4g_db** out, int device, const uint8_t* const* seqs, const uint32_t* lens, uint64_t* tags,
                                 uint64_t* addrs, uint32_t n_seqs) {
    if (!out || (n_seqs && (!seqs || !lens || !tags || !addrs))) return fail(S4G_ERR_INVALID, "NULL argument");
    int have = device_count_raw();
    if (have <= 0) return fail(S4G_ERR_NO_DEVICE, "no HIP device available (libs4g_hip has no CPU fallback)");
    if (device < 0 || device >= have || device >= 64) return fail(S4G_ERR_INVALID, "device %d out of range [0,%d)", device, std::min(have, 64));
    for (uint32_t i = 0; i < n_seqs; ++i) {
        if (lens[i] && !seqs[i]) return fail(S4G_ERR_INVALID, "NULL sequence at %u", i);
        if (lens[i] > 0x7FFFFFF0u) return fail(S4G_ERR_INVALID, "sequence %u too long", i);
    }
    DeviceGuard guard(device);
    DeviceCtx* cx = device_ctx(device);
    std::lock_guard<std::mutex> lk(cx->mu);
    ChainArena& A = g_arena[device];
    if (A.bytes > kArenaCap && A.live == 0) {        // start over: every tag handed out so far becomes a miss
        HIP_TRY(hipDeviceSynchronize());
        for (auto& b : A.blocks) (void)hipFree(b.first);
        A.blocks.clear(); A.cur = nullptr; A.cur_cap = A.cur_used = A.bytes = 0; A.chains = 0;
        ++A.gen;
    }
    const uint64_t tag = arena_tag(A, device);

    // ---- the chains the device does not hold yet: longest first, 64 to a group ----
    std::vector<uint32_t> fresh;
    for (uint32_t i = 0; i < n_seqs; ++i)
        if (tags[i] != tag) fresh.push_back(i);
    std::sort(fresh.begin(), fresh.end(), [&](uint32_t a, uint32_t b) { return lens[a] != lens[b] ? lens[a] > lens[b] : a < b; });
    const uint32_t n_groups = ((uint32_t)fresh.size() + 63) / 64;
    std::vector<uint64_t> grp_off(n_groups + 1, 0);                  // dwords
    for (uint32_t g = 0; g < n_groups; ++g)
        grp_off[g + 1] = grp_off[g] + (uint64_t)std::max<uint32_t>(1, (lens[fresh[(size_t)g * 64]] + 3) / 4) * 64;
    const uint64_t pack_bytes = grp_off[n_groups] * 4;
    const uint64_t tab_bytes = ((uint64_t)n_seqs * 12 + 255) / 256 * 256;
    int rc = stage_reserve(A, pack_bytes + tab_bytes + 256);
    if (rc) return rc;
    char* dst = nullptr;
    if (pack_bytes) {
        if (A.cur_cap - A.cur_used < pack_bytes) {
            const uint64_t want = std::max(kArenaBlock, pack_bytes);
            char* blk = nullptr;
            hipError_t me = hipMalloc((void**)&blk, want);
            if (me != hipSuccess && A.live == 0 && !A.blocks.empty()) {
                // HBM is full of chains nobody is using: start the arena over and try again.  Tags handed out in THIS call
                // so far: none (the hits of this call were checked against `tag` above, so they go stale -- send everything)
                (void)hipGetLastError();
                HIP_TRY(hipDeviceSynchronize());
                for (auto& b : A.blocks) (void)hipFree(b.first);
                A.blocks.clear(); A.cur = nullptr; A.cur_cap = A.cur_used = A.bytes = 0; A.chains = 0;
                ++A.gen;
                return fail(S4G_ERR_NOMEM, "chain arena of device %d started over (out of HBM): call again", device);
            }
            if (me != hipSuccess) { (void)hipGetLastError(); return fail(S4G_ERR_NOMEM, "no room for a chain arena block of %llu bytes on device %d: %s", (unsigned long long)want, device, hipGetErrorString(me)); }
            A.blocks.emplace_back(blk, want);
            A.cur = blk; A.cur_cap = want; A.cur_used = 0; A.bytes += want;
        }
        dst = A.cur + A.cur_used;
        A.cur_used += (pack_bytes + 255) / 256 * 256;
        uint32_t* stage = reinterpret_cast<uint32_t*>(A.h_stage);
        std::atomic<uint32_t> bad{0};                   // a residue code beyond 25
        parallel_for(n_groups, parallel_parts(grp_off[n_groups], 1u << 16), [&](unsigned, unsigned, uint64_t g0, uint64_t g1) {
            for (uint64_t g = g0; g < g1; ++g) {
                uint32_t* grp = stage + grp_off[g];
                const uint32_t len4 = (uint32_t)((grp_off[g + 1] - grp_off[g]) / 64);
                for (uint32_t lane = 0; lane < 64; ++lane) {
                    const size_t k = (size_t)g * 64 + lane;
                    const uint32_t l = k < fresh.size() ? lens[fresh[k]] : 0;
                    const uint8_t* s = l ? seqs[fresh[k]] : nullptr;
                    uint32_t c = 0;
                    for (; c < l / 4; ++c) {
                        uint32_t w; memcpy(&w, s + 4 * c, 4);
                        if ((w & 0xE0E0E0E0u) | ((w + 0x06060606u) & 0x20202020u)) bad.store(1, std::memory_order_relaxed);
                        grp[(size_t)c * 64 + lane] = w;
                    }
                    if (c < len4) {                      // the partial dword, then padding (code 31) to the group's length
                        uint32_t w = 0x1F1F1F1Fu;
                        for (uint32_t b = 0; 4 * c + b < l; ++b) {
                            if (s[4 * c + b] > 25) bad.store(1, std::memory_order_relaxed);
                            w = (w & ~(0xFFu << (8 * b))) | ((uint32_t)(s[4 * c + b] & 31) << (8 * b));
                        }
                        grp[(size_t)c * 64 + lane] = w;
                        for (++c; c < len4; ++c) grp[(size_t)c * 64 + lane] = 0x1F1F1F1Fu;
                    }
                }
            }
        });
        if (bad.load()) {
            A.cur_used -= (pack_bytes + 255) / 256 * 256;
            return fail(S4G_ERR_INVALID, "residue code out of range (codes must be letter - 'A', 0..25)");
        }
        HIP_TRY(hipMemcpyAsync(dst, A.h_stage, pack_bytes, hipMemcpyHostToDevice, nullptr));
        for (size_t k = 0; k < fresh.size(); ++k) {
            const uint32_t i = fresh[k];
            tags[i] = tag;
            addrs[i] = (uint64_t)(uintptr_t)(dst + (grp_off[k / 64] + (k & 63)) * 4);
        }
        A.chains += fresh.size();
    }
    A.misses += fresh.size(); A.hits += n_seqs - fresh.size(); ++A.databases;

    // ---- the database: the two tables of the pair kernels, in a recycled block ----
    s4g_db* db = new s4g_db();
    db->device = device; db->n_seqs = n_seqs; db->cached = true;
    db->len.assign(lens, lens + n_seqs);
    for (uint32_t i = 0; i < n_seqs; ++i) db->residues += lens[i];
    db->padded_residues = db->residues;
    if (n_seqs) {
        char* blk = nullptr;
        uint64_t blk_bytes = 0;
        for (size_t k = 0; k < cx->block_cache.size(); ++k)
            if (cx->block_cache[k].second >= tab_bytes && cx->block_cache[k].second <= 8 * tab_bytes + (1u << 20)) {
                blk = cx->block_cache[k].first; blk_bytes = cx->block_cache[k].second;
                cx->block_cache.erase(cx->block_cache.begin() + k);
                break;
            }
        if (!blk) {
            blk_bytes = tab_bytes + tab_bytes / 4;
            hipError_t e = hipMalloc((void**)&blk, blk_bytes);
            // (the pack upload above may still be reading h_stage: the next call's stage_reserve waits on stage_done, so every exit
            // from here on leaves the staging buffer idle first)
            if (e != hipSuccess) { (void)hipStreamSynchronize(nullptr); delete db; return fail(S4G_ERR_HIP, "hipMalloc of the candidate tables failed: %s", hipGetErrorString(e)); }
        }
        db->d_cached_block = blk; db->cached_block_bytes = blk_bytes;
        char* h_tab = A.h_stage + (pack_bytes + 255) / 256 * 256;
        memcpy(h_tab, addrs, (size_t)n_seqs * 8);
        memcpy(h_tab + (size_t)n_seqs * 8, lens, (size_t)n_seqs * 4);
        hipError_t e = hipMemcpyAsync(blk, h_tab, (size_t)n_seqs * 12, hipMemcpyHostToDevice, nullptr);
        if (e != hipSuccess) { (void)hipStreamSynchronize(nullptr); (void)hipFree(blk); delete db; return fail(S4G_ERR_HIP, "upload of the candidate tables failed: %s", hipGetErrorString(e)); }
        db->d_seq_addr = reinterpret_cast<uint64_t*>(blk);
        db->d_seq_len = reinterpret_cast<uint32_t*>(blk + (size_t)n_seqs * 8);
        db->seq_tables_in_block = true;
        db->hbm_bytes = blk_bytes;
    }
    HIP_TRY(hipEventRecord(A.stage_done, nullptr));
    ++A.live;
    *out = db;
    return S4G_OK;
}

// called by s4g_db_destroy for a cached database (after the device has finished with it)
void chain_cache_release(s4g_db* db) {
    DeviceCtx* cx = device_ctx(db->device);
    std::lock_guard<std::mutex> lk(cx->mu);
    ChainArena& A = g_arena[db->device & 63];
    if (A.live) --A.live;
    if (db->d_cached_block) {
        if (cx->block_cache.size() < 16) cx->block_cache.emplace_back(db->d_cached_block, db->cached_block_bytes);
        else (void)hipFree(db->d_cached_block);
    }
    db->d_cached_block = nullptr; db->d_seq_addr = nullptr; db->d_seq_len = nullptr;
}

// out[0] = chains the arena holds, out[1] = its bytes, out[2] / out[3] = sequences of s4g_db_create_cached calls that were
// found there / had to be sent, out[4] = generation, out[5] = databases created, out[6] = databases alive.  cap >= 7.
S4G_API int s4g_chain_cache_info(int device, uint64_t* out, uint32_t cap) {
    if (!out || cap < 7) return fail(S4G_ERR_INVALID, "need room for 7 values");
    if (device < 0 || device >= 64) return fail(S4G_ERR_INVALID, "bad device %d", device);
    DeviceCtx* cx = device_ctx(device);
    std::lock_guard<std::mutex> lk(cx->mu);
    const ChainArena& A = g_arena[device];
    out[0] = A.chains; out[1] = A.bytes; out[2] = A.hits; out[3] = A.misses; out[4] = A.gen; out[5] = A.databases; out[6] = A.live;
    return S4G_OK;
}

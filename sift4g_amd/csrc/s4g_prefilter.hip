// s4g_prefilter.hip -- host side of the k-mer prefilter on the device (include/s4g_hip.h: s4g_prefilter_*).
//
// Replaces the compute of /root/reference/sift4g/src/database_search.cpp:66-253 (with hash.cpp:21-90); the kernels and the
// pass structure are described in kmer_prefilter.hip.h.  One handle owns a stream and grow-only buffers: the CSR index of the
// queries' k-mers, the chunk being processed, the work arena of one pass and the running candidate lists.  A chunk whose
// hits would not fit the arena budget is processed in several passes over ranges of its targets.
#include "s4g_internal.hpp"
#include "kmer_prefilter.hip.h"

#include <hipcub/hipcub.hpp>

#define S4G_TRY(expr) do { const int rc_ = (expr); if (rc_ != S4G_OK) return rc_; } while (0)

namespace {

constexpr uint64_t kPassHits = 1ull << 26;          // hits per pass the arena is sized for (about 5 GB at 76 B per hit)
constexpr uint64_t kMaxPassHits = (1ull << 31) - 64; // rocPRIM takes int sizes: one target with more hits than this is refused

inline uint64_t align256(uint64_t v) { return (v + 255) & ~255ull; }
inline uint32_t bits_for(uint64_t n_values) { uint32_t b = 1; while (b < 63 && (1ull << b) < n_values) ++b; return b; }
using Clock = std::chrono::steady_clock;
inline double us(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

}  // namespace

struct s4g_prefilter {
    int device = 0;
    uint32_t nq = 0, k = 0, max_c = 0;
    uint32_t q_bits = 1;            // bits of a query index inside a hit key
    uint32_t list_q_bits = 1;       // bits of the query field of a candidate key that take part in the sort (covers the empty mark)
    uint32_t n_words = 0;           // radix^k / 32 index slots
    uint32_t radix = 32;            // digits of a k-mer's number: the codes the queries use + one for all others (kmer_prefilter.hip.h: pf_kmer)
    uint32_t f_q_bits = 0;          // packing of d_fhit (0: unused, every bucket through d_bounds)
    char* d_index = nullptr;        // the block every d_ pointer of the index points into
    uint8_t* d_letter = nullptr;    // [32] code -> digit
    uint32_t* d_fhit = nullptr;     // [n_hash_hits + 1] first hit of the present k-mers, by rank
    hipStream_t stream = nullptr;
    uint2* d_slots = nullptr;       // [n_words] (presence bits, present k-mers before the word)   (kmer_prefilter.hip.h)
    uint32_t* d_bounds = nullptr;   // [n_hash_hits + 2] bucket bounds of the present k-mers, by rank
    uint32_t* d_hit_q = nullptr;
    uint32_t* d_hit_p = nullptr;
    uint64_t* d_hit_qp = nullptr;   // interleaved (position << 32 | query): what the fused path reads
    uint32_t* d_q_kmers = nullptr;
    uint64_t n_hash_hits = 0;
    uint64_t* d_list_key = nullptr; // [nq * max_c] running lists: query << 32 | ~bits(score); kPfNoEntry = empty
    uint32_t* d_list_val = nullptr; //                              database index
    uint8_t* d_codes = nullptr;  uint64_t codes_cap = 0;        // the chunk (s4g_prefilter_add)
    uint64_t* d_off = nullptr;   uint64_t off_cap = 0;
    char* h_stage = nullptr;     uint64_t stage_cap = 0;        // pinned: chunk on its way up, offsets of hits on their way down
    char* d_work = nullptr;      uint64_t work_cap = 0;         // arena of one pass
    unsigned* d_flags = nullptr;                                // [0] bad code, [1] long segments, [3] segments
    unsigned* h_flags = nullptr;                                // pinned
    uint32_t grid_waves = 0;
    uint64_t pass_hits = kPassHits;  // S4G_PREFILTER_PASS_HITS overrides (tests: several passes on a small chunk)
    uint32_t last_first_id = 0, last_n = 0;
    bool any_chunk = false;
    uint64_t stat_hits = 0, stat_pairs = 0, stat_long = 0, stat_chunks = 0;
    // the fused path (kmer_prefilter.hip.h): lists = nq rows of max_c keys ~score << 32 | id, ascending, d_list_cnt entries used
    bool fused = false;
    bool lists_filling = true;      // no candidate has been merged yet
    bool scatter = false;           // candidates into query order by the counting pass (pf_cand_*) rather than a radix sort
    uint32_t lds_keys = 0;          // 64-bit keys a block of pf_merge_lists sorts in LDS
    uint32_t* d_list_cnt = nullptr;
    uint64_t* d_list_worst = nullptr;   // [nq] largest key of a full list (~0 until then): pf_cand_count / pf_cand_scatter drop what lies beyond
    uint64_t cand_guess = 0;        // candidates the last chunk produced (sizes the next chunk's arrays)
    uint64_t stat_overflow = 0, stat_repeats = 0;
};

namespace {

int ensure_stage(s4g_prefilter* p, uint64_t bytes) {
    if (p->stage_cap >= bytes) return S4G_OK;
    if (!p->h_stage) {                                   // a handle's first: what the device's last handle left behind, if that is enough
        DeviceCtx* cx = device_ctx(p->device);
        std::lock_guard<std::mutex> lk(cx->mu);
        if (cx->pf_spare_stage && cx->pf_spare_stage_cap >= bytes) {
            p->h_stage = cx->pf_spare_stage; p->stage_cap = cx->pf_spare_stage_cap;
            cx->pf_spare_stage = nullptr; cx->pf_spare_stage_cap = 0;
            return S4G_OK;
        }
    }
    if (p->h_stage) { HIP_TRY(hipStreamSynchronize(p->stream)); HIP_TRY(hipHostFree(p->h_stage)); p->h_stage = nullptr; p->stage_cap = 0; }
    const uint64_t want = bytes + bytes / 8;
    HIP_TRY(hipHostMalloc((void**)&p->h_stage, want, hipHostMallocDefault));
    p->stage_cap = want;
    return S4G_OK;
}

template <typename T>
int ensure_dev(s4g_prefilter* p, T** ptr, uint64_t* cap, uint64_t n) {
    if (*cap >= n) return S4G_OK;
    if (*ptr) { HIP_TRY(hipStreamSynchronize(p->stream)); HIP_TRY(hipFree(*ptr)); *ptr = nullptr; *cap = 0; }
    const uint64_t want = n + n / 8;
    if (hipMalloc((void**)ptr, want * sizeof(T)) != hipSuccess) return fail(S4G_ERR_NOMEM, "prefilter: cannot allocate %llu bytes of device memory", (unsigned long long)(want * sizeof(T)));
    *cap = want;
    return S4G_OK;
}

// the work arena: bytes, not elements
int ensure_work(s4g_prefilter* p, uint64_t need) {
    if (p->work_cap >= need) return S4G_OK;
    if (!p->d_work) {
        DeviceCtx* cx = device_ctx(p->device);
        std::lock_guard<std::mutex> lk(cx->mu);
        if (cx->pf_spare_work && cx->pf_spare_work_cap >= need) {
            p->d_work = cx->pf_spare_work; p->work_cap = cx->pf_spare_work_cap;
            cx->pf_spare_work = nullptr; cx->pf_spare_work_cap = 0;
            return S4G_OK;
        }
    }
    return ensure_dev(p, &p->d_work, &p->work_cap, need);
}

// bump allocator over the arena
struct Arena {
    char* base; uint64_t at = 0;
    template <typename T> T* take(uint64_t n) { T* r = (T*)(base + at); at += align256(n * sizeof(T)); return r; }
};

// Builds the CSR index of the queries' k-mers on the device (hash.cpp:56-85).
int build_index(s4g_prefilter* p, const uint8_t* codes, const uint64_t* off) {
    const auto t_index0 = Clock::now();
    const uint64_t total = off[p->nq] - off[0];
    // the index's radix: one digit per code that occurs in a query (in code order), one more for every other code
    uint8_t letter[32];
    {
        bool used[32] = {};
        parallel_for(total, parallel_parts(total, 1u << 22), [&](unsigned, unsigned, uint64_t b, uint64_t e) {
            bool mine[32] = {};
            for (uint64_t i = b; i < e; ++i) mine[codes[off[0] + i] & 31u] = true;
            for (int c = 0; c < 32; ++c) if (mine[c]) __atomic_store_n(&used[c], true, __ATOMIC_RELAXED);
        });
        uint32_t n_used = 0;
        for (int c = 0; c < 32; ++c) n_used += used[c];
        if (n_used >= 31) { p->radix = 32; for (int c = 0; c < 32; ++c) letter[c] = (uint8_t)c; }
        else { p->radix = n_used + 1; uint32_t d = 0; for (int c = 0; c < 32; ++c) letter[c] = (uint8_t)(used[c] ? d++ : n_used); }
    }
    uint64_t n_kmer_values = 1;
    for (uint32_t i = 0; i < p->k; ++i) n_kmer_values *= p->radix;
    const uint32_t invalid_key = (uint32_t)n_kmer_values;            // sorts behind every real k-mer
    const int key_bits = (int)bits_for(n_kmer_values + 1);
    p->n_words = (uint32_t)(n_kmer_values / 32 + 1);
    std::vector<uint32_t> q_kmers(std::max(1u, p->nq), 0);
    std::vector<uint64_t> rel(p->nq + 1);
    uint64_t longest = 0;
    for (uint32_t q = 0; q <= p->nq; ++q) rel[q] = off[q] - off[0];
    for (uint32_t q = 0; q < p->nq; ++q) {
        const uint64_t len = off[q + 1] - off[q];
        if (off[q + 1] < off[q] || len > 0x7FFFFFF0ull) return fail(S4G_ERR_INVALID, "bad query offsets at %u", q);
        q_kmers[q] = len >= p->k ? (uint32_t)(len - p->k + 1) : 0;
        p->n_hash_hits += q_kmers[q];
        longest = std::max(longest, len);
    }
    // (S4G_PREFILTER_UNPACKED=1: the tests' switch for the two fallbacks no ordinary query set reaches -- hits that do not fit a bucket
    // word beside their query, buckets beyond 2^23 -- every lookup then goes through bounds[] and hit_qp)
    const bool unpacked = getenv("S4G_PREFILTER_UNPACKED") != nullptr;
    p->f_q_bits = !unpacked && p->q_bits + bits_for(longest + 1) <= 31 ? p->q_bits : 0;
    if (total >= (1ull << 32)) return fail(S4G_ERR_INVALID, "prefilter: more than 2^32 query residues");
    // the index in ONE device block (eight allocations and as many releases were a third of a millisecond each way), the temporaries of
    // its construction in the handle's work arena (which the first chunk needs anyway)
    const uint64_t hits1 = std::max<uint64_t>(1, p->n_hash_hits);
    const uint64_t index_bytes = 256 + align256((uint64_t)std::max(1u, p->nq) * 4) + align256((uint64_t)p->n_words * 8) + 2 * align256(hits1 * 4) +
                                 align256(hits1 * 8) + align256((p->n_hash_hits + 2) * 4) + align256((p->n_hash_hits + 1) * 4);
    if (hipMalloc((void**)&p->d_index, index_bytes) != hipSuccess) return fail(S4G_ERR_NOMEM, "prefilter: cannot allocate %llu bytes of device memory", (unsigned long long)index_bytes);
    Arena I{p->d_index};
    p->d_letter = I.take<uint8_t>(32);
    p->d_q_kmers = I.take<uint32_t>(std::max(1u, p->nq));
    p->d_slots = I.take<uint2>(p->n_words);
    p->d_hit_q = I.take<uint32_t>(hits1); p->d_hit_p = I.take<uint32_t>(hits1); p->d_hit_qp = I.take<uint64_t>(hits1);
    p->d_bounds = I.take<uint32_t>(p->n_hash_hits + 2);
    p->d_fhit = I.take<uint32_t>(p->n_hash_hits + 1);
    HIP_TRY(hipMemcpyAsync(p->d_letter, letter, 32, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipMemsetAsync(p->d_slots, 0, (uint64_t)p->n_words * 8, p->stream));
    HIP_TRY(hipMemcpyAsync(p->d_q_kmers, q_kmers.data(), q_kmers.size() * 4, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipMemsetAsync(p->d_bounds, 0, (p->n_hash_hits + 2) * 4, p->stream));
    size_t cub_scan = 0, cub_sort = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, cub_scan, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)p->n_words, p->stream));
    if (total) HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, cub_sort, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                                          (uint32_t*)nullptr, (int)total, 0, key_bits, p->stream));
    const uint64_t total1 = std::max<uint64_t>(1, total);
    S4G_TRY(ensure_work(p, 2 * align256((uint64_t)p->n_words * 4) + align256(std::max(cub_scan, cub_sort)) + align256(total1) +
                               align256(((uint64_t)p->nq + 1) * 8) + 6 * align256(total1 * 4) + 4096));
    Arena T{p->d_work};
    struct Buf32 { uint32_t* p; };                       // (the names the code below was written with)
    const Buf32 cnt{T.take<uint32_t>(p->n_words)}, before{T.take<uint32_t>(p->n_words)};     // per index word: present k-mers in it, and before it
    struct BufC { char* p; } const tmp{T.take<char>(std::max(cub_scan, cub_sort))};
    struct Buf8 { uint8_t* p; } const d_codes{T.take<uint8_t>(total1)};
    struct Buf64 { uint64_t* p; } const d_off{T.take<uint64_t>((uint64_t)p->nq + 1)};
    const Buf32 key{T.take<uint32_t>(total1)}, val{T.take<uint32_t>(total1)}, key2{T.take<uint32_t>(total1)}, val2{T.take<uint32_t>(total1)},
                pos_q{T.take<uint32_t>(total1)}, pos_j{T.take<uint32_t>(total1)};
    if (total) {
        HIP_TRY(hipMemcpyAsync(d_codes.p, codes + off[0], total, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipMemcpyAsync(d_off.p, rel.data(), rel.size() * 8, hipMemcpyHostToDevice, p->stream));
        hipLaunchKernelGGL(s4g::pf_query_kmers, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, p->stream, d_codes.p, d_off.p, p->nq, total, p->k,
                           p->d_letter, p->radix, invalid_key, key.p, val.p, pos_q.p, pos_j.p, p->d_slots, p->d_flags);
        size_t t = std::max(cub_scan, cub_sort);
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, t, key.p, key2.p, val.p, val2.p, (int)total, 0, key_bits, p->stream));
        if (p->n_hash_hits)
            hipLaunchKernelGGL(s4g::pf_hash_hits, dim3((unsigned)((p->n_hash_hits + 255) / 256)), dim3(256), 0, p->stream, val2.p, pos_q.p, pos_j.p,
                               p->n_hash_hits, p->d_hit_q, p->d_hit_p, p->d_hit_qp);
    }
    const auto t_queued = Clock::now();
    // ranks of the present k-mers, then the bucket bounds by rank (the k-mers without a full window sorted behind the n_hash_hits real ones)
    hipLaunchKernelGGL(s4g::pf_slot_count, dim3((p->n_words + 255) / 256), dim3(256), 0, p->stream, p->d_slots, p->n_words, cnt.p);
    size_t t = std::max(cub_scan, cub_sort);
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, t, cnt.p, before.p, (int)p->n_words, p->stream));
    hipLaunchKernelGGL(s4g::pf_slot_rank, dim3((p->n_words + 255) / 256), dim3(256), 0, p->stream, p->d_slots, p->n_words, before.p);
    if (p->n_hash_hits)
        hipLaunchKernelGGL(s4g::pf_bucket_bounds, dim3((unsigned)((p->n_hash_hits + 255) / 256)), dim3(256), 0, p->stream, key2.p, val2.p, pos_q.p, pos_j.p,
                           p->n_hash_hits, p->d_slots, p->f_q_bits, unpacked ? 0u : 1u << 23, p->d_bounds, p->d_fhit);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(p->h_flags, p->d_flags, 4, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));            // (the host vectors above go out of scope here)
    if (getenv("S4G_PROFILE"))
        fprintf(stderr, "[s4g profile] prefilter index: radix %u, %u slots, allocations + uploads + first kernels queued %.0f us, rest + wait %.0f us\n",
                p->radix, p->n_words, us(t_index0, t_queued), us(t_queued, Clock::now()));
    if (p->h_flags[0]) return fail(S4G_ERR_INVALID, "prefilter: query residue code out of range (5-bit codes, hash.cpp:16-17)");
    return S4G_OK;
}

// Candidates [list | n_new new ones] in cand_key/val (n = list slots + n_new) -> the best max_c per query back into the lists.
int merge_candidates(s4g_prefilter* p, uint64_t* cand_key, uint32_t* cand_val, uint64_t* out_key, uint32_t* out_val, uint32_t* first,
                     void* cub, size_t cub_bytes, uint64_t n) {
    const uint64_t slots = (uint64_t)p->nq * p->max_c;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub, cub_bytes, cand_key, out_key, cand_val, out_val, (int)n, 0, (int)(32 + p->list_q_bits), p->stream));
    hipLaunchKernelGGL(s4g::pf_fill, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, p->stream, p->d_list_key, slots);
    hipLaunchKernelGGL(s4g::pf_query_first, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->stream, out_key, n, p->nq, first);
    hipLaunchKernelGGL(s4g::pf_keep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->stream, out_key, out_val, n, p->nq, p->max_c, first,
                       p->d_list_key, p->d_list_val);
    HIP_TRY(hipGetLastError());
    return S4G_OK;
}

int merge_fused(s4g_prefilter* p, char* arena, uint64_t arena_bytes, uint32_t* cand_q, uint64_t* cand_sv, uint32_t n_regions, uint64_t region_cap,
                const unsigned long long* d_counters, uint64_t n_used);
int merge_fused_bytes(s4g_prefilter* p, uint64_t n, uint64_t* bytes);

// The general path: every pass through HBM.  tlist == NULL: targets 0 .. nt of the chunk, the whole prefilter in this form
// (handles with fused == false).  tlist != NULL (device array of nt target indices): the targets the fused path left over; their
// candidates are merged into the fused path's lists.
int add_chunk_general(s4g_prefilter* p, const uint8_t* d_codes, const uint64_t* d_off, uint32_t nt, uint32_t first_id, const uint32_t* tlist) {
    const bool profile = getenv("S4G_PROFILE") != nullptr;
    const auto t_begin = Clock::now();
    s4g::PfChunk c{d_codes, d_off, p->d_slots, p->d_bounds, p->d_fhit, p->d_letter, p->radix, p->f_q_bits, p->d_hit_q, p->d_hit_p, p->d_hit_qp, p->k, nt, tlist};

    // ---- pass 0: hits per target, their offsets, down to the host (which cuts the chunk into passes) ----
    size_t cub_scan = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, cub_scan, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)(nt + 1), p->stream));
    const uint64_t b_cnt = align256((uint64_t)(nt + 1) * 8);
    S4G_TRY(ensure_work(p, 2 * b_cnt + align256(cub_scan)));
    S4G_TRY(ensure_stage(p, b_cnt));
    uint64_t* d_count = (uint64_t*)p->d_work;
    uint64_t* d_hit_off = (uint64_t*)(p->d_work + b_cnt);
    HIP_TRY(hipMemsetAsync(p->d_flags, 0, 16, p->stream));
    HIP_TRY(hipMemsetAsync(d_count + nt, 0, 8, p->stream));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)nt + 3) / 4, p->grid_waves / 4);
    hipLaunchKernelGGL(s4g::pf_target_count, dim3(grid), dim3(256), 0, p->stream, c, d_count, p->d_flags);
    size_t tmp = cub_scan;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(p->d_work + 2 * b_cnt, tmp, d_count, d_hit_off, (int)(nt + 1), p->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(p->h_stage, d_hit_off, (uint64_t)(nt + 1) * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(p->h_flags, p->d_flags, 4, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->h_flags[0]) return fail(S4G_ERR_INVALID, "prefilter: database residue code out of range (5-bit codes, hash.cpp:16-17)");
    std::vector<uint64_t> hit_off((const uint64_t*)p->h_stage, (const uint64_t*)p->h_stage + nt + 1);
    const uint64_t chunk_hits = hit_off[nt];
    if (!tlist) p->stat_hits += chunk_hits;           // (the fused path has counted the hits of the targets it hands over)
    const auto t_counted = Clock::now();
    if (!chunk_hits) return S4G_OK;

    // ---- passes over ranges of targets ----
    const uint64_t slots = (uint64_t)p->nq * p->max_c;
    uint32_t t0 = 0, passes = 0;
    uint64_t chunk_pairs = 0, chunk_long = 0;
    while (t0 < nt) {
        uint32_t t1 = (uint32_t)(std::upper_bound(hit_off.begin() + t0, hit_off.end(), hit_off[t0] + p->pass_hits) - hit_off.begin()) - 1;
        t1 = std::min(nt, std::max(t1, t0 + 1));
        const uint64_t H = hit_off[t1] - hit_off[t0];
        if (!H) { t0 = t1; continue; }
        if (H > kMaxPassHits) return fail(S4G_ERR_INVALID, "prefilter: target %u alone has %llu k-mer hits", first_id + t0, (unsigned long long)H);
        ++passes;
        const int key_bits = (int)(p->q_bits + bits_for(t1 - t0));
        size_t cub_sort = 0, cub_rle = 0, cub_scan32 = 0, cub_cand = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, cub_sort, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                                   (uint32_t*)nullptr, (int)H, 0, key_bits, p->stream));
        HIP_TRY(hipcub::DeviceRunLengthEncode::Encode(nullptr, cub_rle, (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                                                      (uint32_t*)nullptr, (int)H, p->stream));
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, cub_scan32, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)H, p->stream));
        // the candidate sort is sized for the worst case (every hit a pair of its own); it almost never is
        const uint64_t cand_cap = slots + H;
        if (cand_cap > kMaxPassHits) return fail(S4G_ERR_INVALID, "prefilter: %llu candidate slots exceed what one sort takes", (unsigned long long)cand_cap);
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, cub_cand, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                                   (uint32_t*)nullptr, (int)cand_cap, 0, (int)(32 + p->list_q_bits), p->stream));
        uint64_t b_cub = align256(std::max(std::max(cub_sort, cub_rle), std::max(cub_scan32, cub_cand)));
        if (p->fused) { uint64_t fb = 0; S4G_TRY(merge_fused_bytes(p, H, &fb)); b_cub = std::max(b_cub, fb); }
        const uint64_t need = 2 * b_cnt + 2 * align256(H * 8) + 2 * align256(H * 4)            // hit keys / values, in and out
                            + align256(H * 8) + 3 * align256(H * 4)                              // segments: key, length, offset, long list
                            + align256((2 * H + 1) * 4)                                          // tails of the long segments
                            + 2 * align256(cand_cap * 8) + 2 * align256(cand_cap * 4)            // candidates, in and out
                            + align256((uint64_t)p->nq * 4) + b_cub;
        if (p->work_cap < need) {
            // the per-target offsets live at the front of the arena: they are on the host already, the device copy is re-uploaded
            S4G_TRY(ensure_work(p, need));
            d_count = (uint64_t*)p->d_work; d_hit_off = (uint64_t*)(p->d_work + b_cnt);
            HIP_TRY(hipMemcpyAsync(d_hit_off, hit_off.data(), (uint64_t)(nt + 1) * 8, hipMemcpyHostToDevice, p->stream));
            HIP_TRY(hipStreamSynchronize(p->stream));
        }
        Arena A{p->d_work + 2 * b_cnt};
        uint64_t* key_in = A.take<uint64_t>(H);  uint64_t* key_out = A.take<uint64_t>(H);
        uint32_t* val_in = A.take<uint32_t>(H);  uint32_t* val_out = A.take<uint32_t>(H);
        uint64_t* seg_key = A.take<uint64_t>(H); uint32_t* seg_len = A.take<uint32_t>(H); uint32_t* seg_off = A.take<uint32_t>(H);
        uint32_t* long_list = A.take<uint32_t>(H);
        int32_t* tails = A.take<int32_t>(2 * H + 1);
        uint64_t* cand_key = A.take<uint64_t>(cand_cap); uint64_t* cand_key2 = A.take<uint64_t>(cand_cap);
        uint32_t* cand_val = A.take<uint32_t>(cand_cap); uint32_t* cand_val2 = A.take<uint32_t>(cand_cap);
        uint32_t* first = A.take<uint32_t>(p->nq);
        void* cub = A.take<char>(b_cub);

        const uint32_t g_emit = (uint32_t)std::min<uint64_t>(((uint64_t)(t1 - t0) + 3) / 4, p->grid_waves / 4);
        hipLaunchKernelGGL(s4g::pf_target_emit, dim3(g_emit), dim3(256), 0, p->stream, c, t0, t1, d_hit_off, hit_off[t0], p->q_bits, key_in, val_in);
        size_t t = b_cub;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub, t, key_in, key_out, val_in, val_out, (int)H, 0, key_bits, p->stream));
        t = b_cub;
        HIP_TRY(hipcub::DeviceRunLengthEncode::Encode(cub, t, key_out, seg_key, seg_len, p->d_flags + 3, (int)H, p->stream));
        HIP_TRY(hipMemcpyAsync(p->h_flags + 3, p->d_flags + 3, 4, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        const uint32_t n_seg = p->h_flags[3];
        t = b_cub;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub, t, seg_len, seg_off, (int)n_seg, p->stream));
        HIP_TRY(hipMemsetAsync(p->d_flags + 1, 0, 8, p->stream));
        if (!p->fused) {
            HIP_TRY(hipMemcpyAsync(cand_key, p->d_list_key, slots * 8, hipMemcpyDeviceToDevice, p->stream));
            HIP_TRY(hipMemcpyAsync(cand_val, p->d_list_val, slots * 4, hipMemcpyDeviceToDevice, p->stream));
        }
        s4g::PfLisArgs la{seg_key, seg_len, seg_off, p->d_flags + 3, val_out, d_off, t0, first_id, p->q_bits,
                          cand_key + slots, cand_val + slots, long_list, p->d_flags + 1, tails, p->d_q_kmers, tlist};
        hipLaunchKernelGGL(s4g::pf_lis_small, dim3((unsigned)std::min<uint64_t>(((uint64_t)n_seg + 255) / 256, 8192)), dim3(256), 0, p->stream, la);
        hipLaunchKernelGGL(s4g::pf_lis_long, dim3(p->grid_waves), dim3(64), 0, p->stream, la);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(p->h_flags + 1, p->d_flags + 1, 4, hipMemcpyDeviceToHost, p->stream));
        if (!p->fused) S4G_TRY(merge_candidates(p, cand_key, cand_val, cand_key2, cand_val2, first, cub, b_cub, slots + n_seg));
        else if (n_seg) {
            // the new candidates in the fused path's form (into the second candidate array, free in this mode), then its merge
            uint32_t* fq = cand_val2;
            uint64_t* fsv = cand_key2;
            hipLaunchKernelGGL(s4g::pf_append, dim3((n_seg + 255) / 256), dim3(256), 0, p->stream, cand_key + slots, cand_val + slots, n_seg, fq, fsv, (uint64_t)0, (uint64_t)n_seg);
            S4G_TRY(merge_fused(p, (char*)cub, b_cub, fq, fsv, 1, n_seg, nullptr, n_seg));
        }
        HIP_TRY(hipStreamSynchronize(p->stream));
        chunk_pairs += n_seg; chunk_long += p->h_flags[1];
        t0 = t1;
    }
    p->stat_pairs += chunk_pairs; p->stat_long += chunk_long;
    if (profile)
        fprintf(stderr, "[s4g profile] prefilter chunk: %u targets from %u, %llu hits, %llu pairs (%llu on a wave), %u pass%s: count %.0f us, rest %.0f us\n",
                nt, first_id, (unsigned long long)chunk_hits, (unsigned long long)chunk_pairs, (unsigned long long)chunk_long, passes,
                passes == 1 ? "" : "es", us(t_begin, t_counted), us(t_counted, Clock::now()));
    return S4G_OK;
}


// ---- the fused path ----
// scratch of one merge of up to n candidates: the candidates in query order (sv2), first[nq + 1], and either the counting pass's
// per-query counters or (more than kPfScatterMaxQ queries) the query column and the radix sort's own temporary storage
int merge_fused_bytes(s4g_prefilter* p, uint64_t n, uint64_t* bytes) {
    size_t cub = 0;
    if (!p->scatter)
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, cub, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint64_t*)nullptr,
                                                   (uint64_t*)nullptr, (int)std::max<uint64_t>(n, 1), 0, (int)p->q_bits + 1, p->stream));
    *bytes = align256(n * 4) + align256(n * 8) + 3 * align256(((uint64_t)p->nq + 1) * 4) + align256(cub) + 1024;
    return S4G_OK;
}

// Candidates (query, ~score << 32 | id) in n_regions regions of region_cap slots (the first d_counters[8 r] of region r used; d_counters
// == NULL: one region, all region_cap slots used; n_used: the used slots of all regions together), any order -> into the running lists: brought into query order by a counting pass
// (or, beyond kPfScatterMaxQ queries, a radix sort over the whole capacity: the unused slots then must hold 0xFFFFFFFF), then one
// block per query selects the best max_c of its list and its new candidates in LDS.  Asynchronous on p->stream.
int merge_fused(s4g_prefilter* p, char* arena, uint64_t arena_bytes, uint32_t* cand_q, uint64_t* cand_sv, uint32_t n_regions, uint64_t region_cap,
                const unsigned long long* d_counters, uint64_t n_used) {
    const uint64_t n = (uint64_t)n_regions * region_cap;
    if (!n || !n_used) return S4G_OK;
    if (n > kMaxPassHits) return fail(S4G_ERR_INVALID, "prefilter: %llu candidates in one merge", (unsigned long long)n);
    uint64_t need = 0;
    S4G_TRY(merge_fused_bytes(p, n, &need));
    if (arena_bytes < need) return fail(S4G_ERR_INVALID, "prefilter: internal error: merge scratch of %llu bytes, %llu needed", (unsigned long long)arena_bytes, (unsigned long long)need);
    Arena A{arena};
    uint64_t* sv2 = A.take<uint64_t>(n);
    uint32_t* first = A.take<uint32_t>((uint64_t)p->nq + 1);
    if (p->scatter) {
        uint32_t* q_count = A.take<uint32_t>((uint64_t)p->nq + 1);
        uint32_t* cursor = A.take<uint32_t>((uint64_t)p->nq + 1);
        const dim3 grid((unsigned)((n_used + s4g::kPfTile - 1) / s4g::kPfTile)), block(512);
        const size_t lds = (size_t)p->nq * 12;
        HIP_TRY(hipMemsetAsync(q_count, 0, ((uint64_t)p->nq + 1) * 4, p->stream));
        const uint64_t* worst = p->lists_filling ? nullptr : p->d_list_worst;       // (before the first merge every list is empty: nothing to compare with)
        hipLaunchKernelGGL(s4g::pf_cand_count, grid, block, lds, p->stream, cand_q, cand_sv, region_cap, d_counters, n_regions, region_cap, p->nq,
                           worst, q_count);
        hipLaunchKernelGGL(s4g::pf_cand_first, dim3(1), dim3(512), 0, p->stream, q_count, p->nq, first, cursor);
        if (p->nq <= s4g::kPfStageMaxQ && !getenv("S4G_PREFILTER_DIRECT_SCATTER"))
            hipLaunchKernelGGL(s4g::pf_cand_scatter_staged, dim3((unsigned)((n_used + s4g::kPfStageTile - 1) / s4g::kPfStageTile)), block,
                               s4g::pf_stage_lds(p->nq), p->stream, cand_q, cand_sv, region_cap, d_counters, n_regions, region_cap, p->nq, worst, cursor, sv2);
        else
            hipLaunchKernelGGL(s4g::pf_cand_scatter, grid, block, lds, p->stream, cand_q, cand_sv, region_cap, d_counters, n_regions, region_cap, p->nq,
                               worst, cursor, sv2);
        p->lists_filling = false;
    } else {
        uint32_t* q2 = A.take<uint32_t>(n);
        size_t cub = arena_bytes - A.at;
        // (q_bits + 1 bits of key: the unused slots of the candidate regions hold 0xFFFFFFFF and sort behind every query)
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(arena + A.at, cub, cand_q, q2, cand_sv, sv2, (int)n, 0, (int)p->q_bits + 1, p->stream));
        hipLaunchKernelGGL(s4g::pf_bucket_first, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, p->stream, q2, (uint32_t)n, p->nq, first);
    }
    // LDS of a block: the smallest power of two that leaves a quarter of it to a slice of new candidates (5000 -> 8192 keys = 64 KB:
    // two blocks per CU)
    uint32_t keys = 1024;
    while (keys < p->lds_keys && (uint64_t)p->max_c * 4 > (uint64_t)keys * 3) keys *= 2;
    hipLaunchKernelGGL(s4g::pf_merge_lists, dim3(p->nq), dim3(512), (size_t)keys * 8, p->stream, p->d_list_key, p->d_list_cnt, p->max_c, sv2, first, keys, p->scatter ? p->d_list_worst : nullptr);
    HIP_TRY(hipGetLastError());
    return S4G_OK;
}

int add_chunk_fused(s4g_prefilter* p, const uint8_t* d_codes, const uint64_t* d_off, uint32_t nt, uint32_t first_id, uint64_t residues) {
    const bool profile = getenv("S4G_PROFILE") != nullptr;
    const auto t_begin = Clock::now();
    s4g::PfChunk c{d_codes, d_off, p->d_slots, p->d_bounds, p->d_fhit, p->d_letter, p->radix, p->f_q_bits, p->d_hit_q, p->d_hit_p, p->d_hit_qp, p->k, nt, nullptr};
    constexpr uint32_t R = s4g::kPfRegions;
    constexpr uint64_t kCounterWords = 8 * R + 8;
    unsigned long long counters[kCounterWords];
    unsigned long long n_cand = 0, n_hits = 0, n_long = 0, n_over = 0;
    uint32_t* overflow = nullptr;
    S4G_TRY(ensure_stage(p, 16384));
    for (int attempt = 0;; ++attempt) {
        // Candidates: one per (target, query) pair with a hit -- about one per five residues at k = 5 against a thousand queries.  The
        // array is cut into regions with a fill counter each (kmer_prefilter.hip.h); the kernel counts every candidate even when its
        // region is full, and the chunk is then repeated with room for the fullest region.
        const uint64_t region_cap = std::max<uint64_t>(std::max<uint64_t>(p->cand_guess + p->cand_guess / 4 + 256, residues / (3 * R)), 1024);
        const uint64_t cand_cap = region_cap * R;
        if (cand_cap > kMaxPassHits) return fail(S4G_ERR_INVALID, "prefilter: %llu candidates in one chunk: hand the database over in smaller chunks", (unsigned long long)cand_cap);
        uint64_t merge_bytes = 0;
        S4G_TRY(merge_fused_bytes(p, cand_cap, &merge_bytes));
        const uint64_t need = align256(kCounterWords * 8) + align256((uint64_t)nt * 4) + align256(cand_cap * 4) + align256(cand_cap * 8) + merge_bytes;
        S4G_TRY(ensure_work(p, need));
        Arena A{p->d_work};
        unsigned long long* d_counters = A.take<unsigned long long>(kCounterWords);
        overflow = A.take<uint32_t>(nt);
        uint32_t* cand_q = A.take<uint32_t>(cand_cap);
        uint64_t* cand_sv = A.take<uint64_t>(cand_cap);
        char* scratch = p->d_work + A.at;
        HIP_TRY(hipMemsetAsync(d_counters, 0, kCounterWords * 8, p->stream));
        HIP_TRY(hipMemsetAsync(p->d_flags, 0, 16, p->stream));
        if (!p->scatter) HIP_TRY(hipMemsetAsync(cand_q, 0xFF, cand_cap * 4, p->stream));      // the radix merge sorts the whole capacity: unused slots behind every query
        s4g::PfFused f{c, first_id, p->nq, cand_q, cand_sv, region_cap, d_counters, overflow, p->d_flags};
        const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)nt + 3) / 4, p->grid_waves / 4);
        hipLaunchKernelGGL(s4g::pf_target_lis, dim3(grid), dim3(256), 0, p->stream, f);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(p->h_stage, d_counters, kCounterWords * 8, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipMemcpyAsync(p->h_flags, p->d_flags, 4, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        if (p->h_flags[0]) return fail(S4G_ERR_INVALID, "prefilter: database residue code out of range (5-bit codes, hash.cpp:16-17)");
        memcpy(counters, p->h_stage, kCounterWords * 8);
        unsigned long long fullest = 0;
        n_cand = 0;
        for (uint32_t r = 0; r < R; ++r) { fullest = std::max(fullest, counters[8 * r]); n_cand += counters[8 * r]; }
        n_hits = counters[8 * R]; n_long = counters[8 * R + 1]; n_over = counters[8 * R + 2];
        p->cand_guess = fullest;
        if (fullest > region_cap) {
            if (attempt) return fail(S4G_ERR_HIP, "prefilter: internal error: candidate count changed between two runs of a chunk");
            ++p->stat_repeats;
            continue;
        }
        S4G_TRY(merge_fused(p, scratch, p->work_cap - A.at, cand_q, cand_sv, R, region_cap, d_counters, n_cand));
        break;
    }
    p->stat_hits += n_hits; p->stat_pairs += n_cand; p->stat_long += n_long; p->stat_overflow += n_over;
    const auto t_fused = Clock::now();
    if (n_over) {
        // targets where a single slice of the query range has more hits than a wave keeps in LDS: through the general passes, as a
        // list (the indices move out of the arena first; ascending, so that the general path's per-target keys follow the database order)
        HIP_TRY(hipStreamSynchronize(p->stream));
        std::vector<uint32_t> h(n_over);
        HIP_TRY(hipMemcpy(h.data(), overflow, h.size() * 4, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        DevBuf<uint32_t> tlist;
        HIP_TRY(tlist.upload(h));
        S4G_TRY(add_chunk_general(p, d_codes, d_off, (uint32_t)n_over, first_id, tlist.p));
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
#ifdef PF_TIMING
    {
        unsigned long long tm[8];
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipMemcpyFromSymbol(tm, HIP_SYMBOL(s4g::pf_times), sizeof(tm)));
        unsigned long long zero[8] = {};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(s4g::pf_times), zero, sizeof(zero)));
        unsigned long long all = 0;
        for (int i = 0; i < 8; ++i) all += tm[i];
        fprintf(stderr, "[s4g profile] pf_target_lis wave clocks: lookups ahead %.1f %%, collect %.1f %%, sort %.1f %%, segments %.1f %%, short LIS + out %.1f %%, long %.1f %%, "
                        "per-range walks %.1f %%, rest %.1f %% of %.3e\n", 100.0 * tm[0] / all, 100.0 * tm[1] / all, 100.0 * tm[2] / all, 100.0 * tm[3] / all,
                100.0 * tm[4] / all, 100.0 * tm[5] / all, 100.0 * tm[6] / all, 100.0 * tm[7] / all, (double)all);
    }
#endif
    if (profile)
        fprintf(stderr, "[s4g profile] prefilter chunk (fused): %u targets from %u, %llu hits, %llu pairs (%llu on a wave), %llu targets through the general passes: "
                        "target kernel + sync %.0f us, rest %.0f us\n", nt, first_id, n_hits, n_cand, n_long, n_over, us(t_begin, t_fused), us(t_fused, Clock::now()));
    return S4G_OK;
}

int add_chunk(s4g_prefilter* p, const uint8_t* d_codes, const uint64_t* d_off, uint32_t nt, uint32_t first_id, uint64_t residues) {
    if (p->any_chunk && first_id < p->last_first_id + p->last_n)
        return fail(S4G_ERR_INVALID, "prefilter: chunks must arrive in ascending id order (ties at the cut go to the lower index)");
    p->any_chunk = true; p->last_first_id = first_id; p->last_n = nt;
    ++p->stat_chunks;
    if (!nt || !p->nq) return S4G_OK;
    return p->fused ? add_chunk_fused(p, d_codes, d_off, nt, first_id, residues) : add_chunk_general(p, d_codes, d_off, nt, first_id, nullptr);
}

}  // namespace

// s4g_warmup: load this unit's code object (the occupancy query resolves the kernels) and rocPRIM's run-length encoder
int warm_prefilter_kernels(DeviceCtx* cx) {
    (void)cx;
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)s4g::pf_target_count, 256, 0));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)s4g::pf_lis_long, 64, 0));
    DevBuf<uint64_t> buf;
    HIP_TRY(buf.alloc(4 * 64 + 8192));
    HIP_TRY(hipMemset(buf.p, 0, (4 * 64 + 8192) * 8));
    size_t tmp = 8192 * 8 - 512;
    uint64_t* p = buf.p;
    HIP_TRY(hipcub::DeviceRunLengthEncode::Encode(p + 256, tmp, p, p + 64, (uint32_t*)(p + 128), (uint32_t*)(p + 192), 64, nullptr));
    tmp = 8192 * 8 - 512;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(p + 256, tmp, p, p + 64, (uint32_t*)(p + 128), (uint32_t*)(p + 192), 64, 0, 40, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return S4G_OK;
}

S4G_API int s4g_prefilter_create(s4g_prefilter** out, int device, const uint8_t* codes, const uint64_t* offsets, uint32_t n_queries,
                                 uint32_t kmer_length, uint32_t max_candidates) {
    if (!out || !offsets || (!codes && n_queries && offsets[n_queries] != offsets[0])) return fail(S4G_ERR_INVALID, "NULL argument");
    if (kmer_length < 3 || kmer_length > 5) return fail(S4G_ERR_INVALID, "kmer_length possible values = 3,4,5 (main.cpp:169), got %u", kmer_length);
    if (max_candidates == 0) return fail(S4G_ERR_INVALID, "max_candidates must be at least 1");
    if ((uint64_t)n_queries * max_candidates > (1ull << 30)) return fail(S4G_ERR_INVALID, "prefilter: n_queries x max_candidates above 2^30");
    const int have = device_count_raw();
    if (have <= 0) return fail(S4G_ERR_NO_DEVICE, "no HIP device available (libs4g_hip has no CPU fallback)");
    if (device < 0 || device >= have) return fail(S4G_ERR_INVALID, "device %d out of range [0,%d)", device, have);
    DeviceGuard guard(device);
    s4g_prefilter* p = new s4g_prefilter();
    p->device = device; p->nq = n_queries; p->k = kmer_length; p->max_c = max_candidates;
    p->q_bits = bits_for(std::max(1u, n_queries));
    p->list_q_bits = std::min(32u, bits_for((uint64_t)n_queries + 1));
    if (const char* e = getenv("S4G_PREFILTER_PASS_HITS")) { const long long v = atoll(e); if (v > 0) p->pass_hits = (uint64_t)v; }
    const bool profile = getenv("S4G_PROFILE") != nullptr;
    const auto t_begin = Clock::now();
    auto t_setup = t_begin;
    auto body = [&]() -> int {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        p->grid_waves = (uint32_t)prop.multiProcessorCount * 32;
        HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        HIP_TRY(hipMalloc((void**)&p->d_flags, 64));
        HIP_TRY(hipHostMalloc((void**)&p->h_flags, 64, hipHostMallocDefault));
        HIP_TRY(hipMemsetAsync(p->d_flags, 0, 64, p->stream));
        const uint64_t slots = std::max<uint64_t>(1, (uint64_t)n_queries * max_candidates);
        // The fused path (kmer_prefilter.hip.h) wherever it applies: a query index must fit a 32-bit sort key beside 10 bits of order,
        // and a query's list plus a slice of new candidates must fit the LDS a block may have (64-bit keys: half of them the list).
        int lds_bytes = 0;
        HIP_TRY(hipDeviceGetAttribute(&lds_bytes, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
        p->lds_keys = 1024;
        while (p->lds_keys < 16384 && (uint64_t)p->lds_keys * 2 * 8 <= (uint64_t)lds_bytes) p->lds_keys *= 2;
        p->fused = getenv("S4G_PREFILTER_GLOBAL_SORT") == nullptr && p->q_bits + s4g::kPfOrderBits <= 32 && max_candidates <= p->lds_keys / 2 &&
                   offsets[n_queries] - offsets[0] < (1ull << 31);       // (a hit's index in the index beside a flag bit: kmer_prefilter.hip.h, kPfFetch)
        p->scatter = p->fused && n_queries <= s4g::kPfScatterMaxQ && getenv("S4G_PREFILTER_RADIX_MERGE") == nullptr;
        HIP_TRY(hipMalloc((void**)&p->d_list_key, slots * 8));
        if (p->fused) {
            HIP_TRY(hipMalloc((void**)&p->d_list_cnt, (uint64_t)std::max(1u, n_queries) * 4));
            HIP_TRY(hipMemsetAsync(p->d_list_cnt, 0, (uint64_t)std::max(1u, n_queries) * 4, p->stream));
            HIP_TRY(hipMalloc((void**)&p->d_list_worst, (uint64_t)std::max(1u, n_queries) * 8));
            HIP_TRY(hipMemsetAsync(p->d_list_worst, 0xFF, (uint64_t)std::max(1u, n_queries) * 8, p->stream));
            HIP_TRY(hipFuncSetAttribute((const void*)s4g::pf_cand_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(s4g::kPfScatterMaxQ * 12)));
            HIP_TRY(hipFuncSetAttribute((const void*)s4g::pf_cand_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(s4g::kPfScatterMaxQ * 12)));
            HIP_TRY(hipFuncSetAttribute((const void*)s4g::pf_cand_scatter_staged, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s4g::pf_stage_lds(s4g::kPfStageMaxQ)));
            HIP_TRY(hipFuncSetAttribute((const void*)s4g::pf_merge_lists, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(p->lds_keys * 8)));
            HIP_TRY(hipFuncSetAttribute((const void*)s4g::pf_finish_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(p->lds_keys * 8)));
        } else {
            HIP_TRY(hipMalloc((void**)&p->d_list_val, slots * 4));
            hipLaunchKernelGGL(s4g::pf_fill, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, p->stream, p->d_list_key, slots);
            HIP_TRY(hipGetLastError());
        }
        t_setup = Clock::now();
        return build_index(p, codes, offsets);
    };
    const int rc = body();
    if (profile) fprintf(stderr, "[s4g profile] prefilter create: stream, flags, lists %.0f us, index %.0f us\n", us(t_begin, t_setup), us(t_setup, Clock::now()));
    if (rc != S4G_OK) { s4g_prefilter_destroy(p); return rc; }
    *out = p;
    return S4G_OK;
}

S4G_API int s4g_prefilter_add_device(s4g_prefilter* p, const uint8_t* d_codes, const uint64_t* d_offsets, uint32_t n_targets, uint32_t first_id) {
    if (!p || (n_targets && (!d_codes || !d_offsets))) return fail(S4G_ERR_INVALID, "NULL argument");
    if ((uint64_t)first_id + n_targets > 0xFFFFFFFFull) return fail(S4G_ERR_INVALID, "prefilter: database index beyond 32 bits");
    DeviceGuard guard(p->device);
    if (!n_targets) return add_chunk(p, nullptr, nullptr, 0, first_id, 0);
    // the offsets are checked where they are (ascending, no target beyond 0x7FFFFFF0 residues); their two ends come to the host
    S4G_TRY(ensure_stage(p, 4096));
    HIP_TRY(hipMemsetAsync(p->d_flags, 0, 16, p->stream));
    hipLaunchKernelGGL(s4g::pf_check_offsets, dim3((n_targets + 255) / 256), dim3(256), 0, p->stream, d_offsets, n_targets, p->d_flags);
    HIP_TRY(hipGetLastError());
    uint64_t* ends = (uint64_t*)p->h_stage;
    HIP_TRY(hipMemcpyAsync(ends, d_offsets, 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(ends + 1, d_offsets + n_targets, 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(p->h_flags, p->d_flags, 4, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->h_flags[0] & 2u) return fail(S4G_ERR_INVALID, "prefilter: device offsets are not ascending (or a target is longer than 0x7FFFFFF0 residues)");
    return add_chunk(p, d_codes, d_offsets, n_targets, first_id, ends[1] - ends[0]);
}

S4G_API int s4g_prefilter_add(s4g_prefilter* p, const uint8_t* codes, const uint64_t* offsets, uint32_t n_targets, uint32_t first_id) {
    if (!p || (n_targets && (!codes || !offsets))) return fail(S4G_ERR_INVALID, "NULL argument");
    if ((uint64_t)first_id + n_targets > 0xFFFFFFFFull) return fail(S4G_ERR_INVALID, "prefilter: database index beyond 32 bits");
    DeviceGuard guard(p->device);
    if (!n_targets) return add_chunk(p, nullptr, nullptr, 0, first_id, 0);
    for (uint32_t t = 0; t < n_targets; ++t)
        if (offsets[t + 1] < offsets[t] || offsets[t + 1] - offsets[t] > 0x7FFFFFF0ull) return fail(S4G_ERR_INVALID, "bad offsets at %u", t);
    const uint64_t residues = offsets[n_targets] - offsets[0];
    const uint64_t b_off = (uint64_t)(n_targets + 1) * 8;
    S4G_TRY(ensure_stage(p, align256(b_off) + residues));
    S4G_TRY(ensure_dev(p, &p->d_codes, &p->codes_cap, std::max<uint64_t>(1, residues)));
    S4G_TRY(ensure_dev(p, &p->d_off, &p->off_cap, (uint64_t)n_targets + 1));
    uint64_t* h_off = (uint64_t*)p->h_stage;
    uint8_t* h_codes = (uint8_t*)p->h_stage + align256(b_off);
    for (uint32_t t = 0; t <= n_targets; ++t) h_off[t] = offsets[t] - offsets[0];
    const uint8_t* src = codes + offsets[0];
    parallel_for(residues, parallel_parts(residues, 1u << 22), [&](unsigned, unsigned, uint64_t b, uint64_t e) { memcpy(h_codes + b, src + b, e - b); });
    HIP_TRY(hipMemcpyAsync(p->d_off, h_off, b_off, hipMemcpyHostToDevice, p->stream));
    if (residues) HIP_TRY(hipMemcpyAsync(p->d_codes, h_codes, residues, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));            // the staging buffer is reused for the offsets coming back down
    return add_chunk(p, p->d_codes, p->d_off, n_targets, first_id, residues);
}

// Rows to the host through the pinned staging buffer, in pieces on a ring: while the worker pool copies a piece that has landed into
// the caller's (pageable) memory, the transfers of the next pieces are already queued behind it, so the link never waits for the host.
// (Straight into pageable memory the runtime stages the 60 MB of a thousand queries itself, single-threaded: 5.5 ms; with two 8 MB
// pieces taking turns the link idled during every host copy: 1.07 ms per 20 MB; this ring: see profiles/r05_prefilter.)
static int rows_to_host(s4g_prefilter* p, const void* d_src, void* h_dst, uint64_t bytes) {
    constexpr uint64_t piece = 2ull << 20;
    constexpr int kRing = 6;
    S4G_TRY(ensure_stage(p, kRing * piece));
    hipEvent_t ev[kRing] = {};
    for (int i = 0; i < kRing; ++i) HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    const uint64_t n = (bytes + piece - 1) / piece;
    int rc = S4G_OK;
    auto queue = [&](uint64_t k) {
        const uint64_t b = std::min(piece, bytes - k * piece);
        if (hipMemcpyAsync(p->h_stage + (k % kRing) * piece, (const char*)d_src + k * piece, b, hipMemcpyDeviceToHost, p->stream) != hipSuccess ||
            hipEventRecord(ev[k % kRing], p->stream) != hipSuccess) rc = fail(S4G_ERR_HIP, "prefilter: download failed");
    };
    for (uint64_t k = 0; k < std::min<uint64_t>(n, kRing) && rc == S4G_OK; ++k) queue(k);
    for (uint64_t j = 0; j < n && rc == S4G_OK; ++j) {
        const uint64_t b = std::min(piece, bytes - j * piece);
        if (hipEventSynchronize(ev[j % kRing]) != hipSuccess) { rc = fail(S4G_ERR_HIP, "prefilter: download failed"); break; }
        const char* src = p->h_stage + (j % kRing) * piece;
        char* dst = (char*)h_dst + j * piece;
        parallel_for(b, parallel_parts(b, 1u << 18), [&](unsigned, unsigned, uint64_t x, uint64_t y) { memcpy(dst + x, src + x, y - x); });
        if (j + kRing < n) queue(j + kRing);
    }
    if (rc != S4G_OK) (void)hipStreamSynchronize(p->stream);         // nothing may still be landing in the staging buffer
    for (int i = 0; i < kRing; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

static int finish_fused(s4g_prefilter* p, uint32_t* h_ids, uint32_t* h_counts, float* h_scores) {
    const uint64_t slots = (uint64_t)p->nq * p->max_c;
    const uint64_t need = 2 * align256(slots * 4) + align256((uint64_t)p->nq * 4);
    S4G_TRY(ensure_work(p, need));
    Arena A{p->d_work};
    uint32_t* ids = A.take<uint32_t>(slots); float* scores = A.take<float>(slots);
    uint32_t* counts = A.take<uint32_t>(p->nq);
    uint32_t m = 2;
    while (m < p->max_c) m <<= 1;
    const bool prof = getenv("S4G_PROFILE") != nullptr;
    const auto t0 = Clock::now();
    hipLaunchKernelGGL(s4g::pf_finish_rows, dim3(p->nq), dim3(512), (size_t)m * 8, p->stream, p->d_list_key, p->d_list_cnt, p->max_c, ids, scores, counts);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_counts, counts, (uint64_t)p->nq * 4, hipMemcpyDeviceToHost, p->stream));
    if (prof) HIP_TRY(hipStreamSynchronize(p->stream));
    const auto t1 = Clock::now();
    S4G_TRY(rows_to_host(p, ids, h_ids, slots * 4));
    const auto t2 = Clock::now();
    if (h_scores) S4G_TRY(rows_to_host(p, scores, h_scores, slots * 4));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (prof)
        fprintf(stderr, "[s4g profile] prefilter finish: rows sorted by id on the device %.0f us (everything queued before it included), ids to the host %.0f us, scores %.0f us "
                "(%llu bytes each, through pinned staging)\n", us(t0, t1), us(t1, t2), us(t2, Clock::now()), (unsigned long long)(slots * 4));
    return S4G_OK;
}

S4G_API int s4g_prefilter_finish(s4g_prefilter* p, uint32_t* h_ids, uint32_t* h_counts, float* h_scores) {
    if (!p || (p->nq && (!h_ids || !h_counts))) return fail(S4G_ERR_INVALID, "NULL argument");
    if (!p->nq) return S4G_OK;
    DeviceGuard guard(p->device);
    const uint64_t slots = (uint64_t)p->nq * p->max_c;
    if (p->fused) return finish_fused(p, h_ids, h_counts, h_scores);
    size_t cub_sort = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, cub_sort, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                               (uint32_t*)nullptr, (int)slots, 0, (int)(32 + p->list_q_bits), p->stream));
    const uint64_t need = 2 * align256(slots * 8) + 4 * align256(slots * 4) + 2 * align256((uint64_t)p->nq * 4) + align256(cub_sort);
    S4G_TRY(ensure_work(p, need));
    Arena A{p->d_work};
    uint64_t* key = A.take<uint64_t>(slots); uint64_t* key2 = A.take<uint64_t>(slots);
    uint32_t* val = A.take<uint32_t>(slots); uint32_t* val2 = A.take<uint32_t>(slots);
    uint32_t* ids = A.take<uint32_t>(slots); float* scores = A.take<float>(slots);
    uint32_t* first = A.take<uint32_t>(p->nq); uint32_t* counts = A.take<uint32_t>(p->nq);
    void* cub = A.take<char>(cub_sort);
    const dim3 g((unsigned)((slots + 255) / 256)), b(256);
    hipLaunchKernelGGL(s4g::pf_by_id, g, b, 0, p->stream, p->d_list_key, p->d_list_val, slots, key, val);
    size_t t = cub_sort;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub, t, key, key2, val, val2, (int)slots, 0, (int)(32 + p->list_q_bits), p->stream));
    HIP_TRY(hipMemsetAsync(counts, 0, (uint64_t)p->nq * 4, p->stream));
    HIP_TRY(hipMemsetAsync(ids, 0, slots * 4, p->stream));
    HIP_TRY(hipMemsetAsync(scores, 0, slots * 4, p->stream));
    hipLaunchKernelGGL(s4g::pf_query_first, g, b, 0, p->stream, key2, slots, p->nq, first);
    hipLaunchKernelGGL(s4g::pf_rows, g, b, 0, p->stream, key2, val2, slots, p->nq, p->max_c, first, ids, scores, counts);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_counts, counts, (uint64_t)p->nq * 4, hipMemcpyDeviceToHost, p->stream));
    S4G_TRY(rows_to_host(p, ids, h_ids, slots * 4));
    if (h_scores) S4G_TRY(rows_to_host(p, scores, h_scores, slots * 4));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return S4G_OK;
}

S4G_API int s4g_prefilter_stats(const s4g_prefilter* p, uint64_t* out, uint32_t cap) {
    if (!p || !out || cap < 4) return fail(S4G_ERR_INVALID, "s4g_prefilter_stats: out[4] needed");
    out[0] = p->stat_hits; out[1] = p->stat_pairs; out[2] = p->stat_long; out[3] = p->stat_chunks;
    if (cap >= 7) { out[4] = p->fused ? 1 : 0; out[5] = p->stat_overflow; out[6] = p->stat_repeats; }
    return S4G_OK;
}

S4G_API int s4g_prefilter_destroy(s4g_prefilter* p) {
    if (!p) return S4G_OK;
    DeviceGuard guard(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    (void)hipFree(p->d_index);
    (void)hipFree(p->d_list_key); (void)hipFree(p->d_list_val); (void)hipFree(p->d_list_cnt); (void)hipFree(p->d_list_worst); (void)hipFree(p->d_codes); (void)hipFree(p->d_off);
    {
        // the staging buffer and the arena stay with the device for its next handle (the larger of what is there and what comes back;
        // a staging buffer beyond 64 MB of pinned memory and an arena beyond 1 GB are released)
        DeviceCtx* cx = device_ctx(p->device);
        std::lock_guard<std::mutex> lk(cx->mu);
        if (p->h_stage && p->stage_cap > cx->pf_spare_stage_cap && p->stage_cap <= (64ull << 20)) { std::swap(p->h_stage, cx->pf_spare_stage); std::swap(p->stage_cap, cx->pf_spare_stage_cap); }
        if (p->d_work && p->work_cap > cx->pf_spare_work_cap && p->work_cap <= (1ull << 30)) { std::swap(p->d_work, cx->pf_spare_work); std::swap(p->work_cap, cx->pf_spare_work_cap); }
    }
    (void)hipFree(p->d_work); (void)hipFree(p->d_flags);
    if (p->h_stage) (void)hipHostFree(p->h_stage);
    if (p->h_flags) (void)hipHostFree(p->h_flags);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return S4G_OK;
}

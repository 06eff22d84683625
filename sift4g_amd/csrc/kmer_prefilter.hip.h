// kmer_prefilter.hip.h -- device kernels of the k-mer prefilter (host side: s4g_prefilter.hip).
//
// What the reference does on its thread pool (/root/reference/sift4g/src/database_search.cpp:185-253, hash.cpp:21-90), per
// database chunk, as data-parallel passes over HBM-resident arrays:
//
//   pf_query_kmers     one thread per query residue: the k-mer that starts there (hash.cpp:21-44) -> sort key, presence bit
//   (radix sort, scan, pf_bucket_bounds)  -> the CSR index of hash.cpp:56-85 by RANK of the present k-mers; hits in (query, position)
//                      order inside a bucket
//   pf_target_count    one wave per target: bucket sizes of its k-mers, a k-mer equal to its predecessor skipped (:212-214)
//   pf_target_emit     one wave per target: the hits themselves, in the order the reference appends them (:216-220): by target
//                      position, then bucket order -> key (target, query), value = query position
//   (stable radix sort by key, run-length encode) -> one segment per (target, query) pair with at least one hit, its query
//                      positions in the reference's order
//   pf_lis_small       one thread per segment of up to 32 hits: longest strictly increasing subsequence (:255-280)
//   pf_lis_long        one wave per longer segment: the same, tails in LDS (or global memory beyond 4096), 64-way search
//                      -> candidate (query, ~score bits | target id), score = LIS / (float) target length (:228-229)
//   (radix sort by (query, score descending), stable over ascending target id) + pf_keep -> best max_candidates per query
//
// Round 3 -- the fused path (the default; the passes above remain as the general path for targets with more hits than a wave
// keeps in LDS, for max_candidates beyond what a block sorts in LDS, and behind S4G_PREFILTER_GLOBAL_SORT=1), as it stands after round 5:
//
//   pf_target_lis      one wave per target (dealt out by 64 counters) does ALL of the above for it: the index lookups of its k-mers
//                      (three dependent loads, issued for the wave's NEXT target before this one is processed), the hits into LDS in the
//                      reference's order (DPP prefix sums; hits of shared buckets by reference, fetched by all lanes at once), the
//                      (query, order) keys sorted IN REGISTERS (bitonic; DPP / swizzle partners), segment starts by ballot, the LIS of
//                      every segment (a lane per short one; the whole wave per long one, tails in registers) and one
//                      (query, ~score | id) candidate per segment appended to one of 64 regions of a global array.  Hits never touch HBM.
//   pf_cand_count, pf_cand_first, pf_cand_scatter_staged
//                      the candidates into query order by counting (a tile sorted in LDS, one run per query written out); candidates
//                      beyond the worst key of a list that is full are dropped here.  (pf_cand_scatter: without the LDS stage, 2049 ..
//                      8192 queries; rocPRIM's radix sort beyond.)
//   pf_merge_lists     one block per query: its running list and its new candidates in LDS, the cut found by a radix select, the
//                      best max_candidates kept (unordered)
//   pf_finish_rows     one block per query: its list by ascending id -> the rows the caller gets
//
// The index is COMPACT (hash.cpp:56-85 keeps a vector per possible k-mer: 32^5 slots): a k-mer is a number in the radix of the letters
// the queries use (pf_kmer), slots[kmer >> 5] = (32 presence bits, number of present k-mers before the word), fhit[rank] = the bucket in
// one word (pf_bucket_bounds), bounds[rank] for what does not fit one.  A thousand queries: 1 MB + 1.3 MB, L2-resident, instead of a
// 134 MB table whose every lookup was a 128-byte line from the fabric (round 4: 3/5 of the kernel's counter traffic).
//
// Integer gather work, bound by vector issue: no MFMA anywhere; what matters is instructions per hit, index lookups that stay in L2,
// no dependent memory round trip inside a per-lane loop, and writes that leave a block as whole runs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s4g {

constexpr uint32_t kKmerBits = 5;               // hash.cpp:17
constexpr uint32_t kPfSmallMax = 32;            // longest segment a single thread handles
constexpr uint32_t kPfLdsTails = 4096;          // tails a wave keeps in LDS (16 KB)
constexpr uint64_t kPfNoEntry = ~0ull;          // empty slot of a candidate list: sorts behind everything

// the k-mer starting at codes[j] (hash.cpp:33-43: ((kmer << 5) | code) & mask, most significant residue first) -- as a number in the
// index's own radix: letter[code] = rank of the code among the codes the QUERIES use, every other code = one shared digit behind them
// (a k-mer with such a digit is in no query and in no slot; two different k-mers may then share a number, but neither has a hit), so that
// the index is as dense as the queries' alphabet allows: 21^5 / 32 slots of 8 bytes = 1 MB for twenty amino acids instead of 32^5 / 32 = 8 MB
// of which every 128-byte line holds 7 or 13 live slots.  All 32 codes in use: radix 32, letter = code, the reference's own number.
// (codes are cut to 5 bits: a code of 32 or more is reported through the `bad` flag, and must not index past the tables first)
__device__ __forceinline__ uint32_t pf_kmer(const uint8_t* __restrict__ c, uint32_t k, const uint8_t* __restrict__ letter, uint32_t radix) {
    uint32_t v = 0;
    for (uint32_t i = 0; i < k; ++i) v = v * radix + letter[c[i] & 31u];
    return v;
}

// One thread per query residue g (flat over all queries).  key[g] = k-mer starting there, or the invalid key 1 << 5k when fewer
// than k residues are left in that query (radix^k: behind every real one); val[g] = g; pos_q[g] / pos_j[g] = query index / position
// of g; the k-mer's bit set in its index slot (slots[kmer >> 5].x).
__global__ void pf_query_kmers(const uint8_t* __restrict__ codes, const uint64_t* __restrict__ off, uint32_t nq, uint64_t total,
                               uint32_t k, const uint8_t* __restrict__ letter, uint32_t radix, uint32_t invalid_key, uint32_t* __restrict__ key, uint32_t* __restrict__ val, uint32_t* __restrict__ pos_q,
                               uint32_t* __restrict__ pos_j, uint2* __restrict__ slots, unsigned* __restrict__ bad) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    uint32_t lo = 0, hi = nq;                   // last q with off[q] <= g  (queries of length 0 share an offset: take the last)
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (off[mid] <= g) lo = mid; else hi = mid; }
    const uint32_t j = (uint32_t)(g - off[lo]);
    const uint32_t len = (uint32_t)(off[lo + 1] - off[lo]);
    pos_q[g] = lo; pos_j[g] = j; val[g] = (uint32_t)g;
    if (codes[g] >= 32) atomicOr(bad, 1u);
    if (j + k <= len) {
        const uint32_t v = pf_kmer(codes + g, k, letter, radix);
        key[g] = v;
        atomicOr(&slots[v >> 5].x, 1u << (v & 31));
    } else {
        key[g] = invalid_key;
    }
}

// hits of the CSR index from the sorted (k-mer, g) pairs: hit_q / hit_p[i] = query / position of the i-th smallest
__global__ void pf_hash_hits(const uint32_t* __restrict__ sorted_val, const uint32_t* __restrict__ pos_q,
                             const uint32_t* __restrict__ pos_j, uint64_t n_hits, uint32_t* __restrict__ hit_q, uint32_t* __restrict__ hit_p,
                             uint64_t* __restrict__ hit_qp) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hits) return;
    const uint32_t g = sorted_val[i];
    hit_q[i] = pos_q[g]; hit_p[i] = pos_j[g];
    hit_qp[i] = ((uint64_t)pos_j[g] << 32) | pos_q[g];
}

// The compact index.  slots[w] = (bits of the k-mers 32 w .. 32 w + 31 that some query holds, number of such k-mers below 32 w):
// pf_slot_count -> exclusive scan -> pf_slot_rank fill in the second half.
__global__ void pf_slot_count(const uint2* __restrict__ slots, uint32_t n_words, uint32_t* __restrict__ cnt) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n_words) cnt[w] = (uint32_t)__popc(slots[w].x);
}
__global__ void pf_slot_rank(uint2* __restrict__ slots, uint32_t n_words, const uint32_t* __restrict__ before) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n_words) slots[w].y = before[w];
}
__device__ __forceinline__ uint32_t pf_rank(uint2 slot, uint32_t v) { return slot.y + (uint32_t)__popc(slot.x & ((1u << (v & 31)) - 1u)); }
// bounds[r] = first hit of the r-th present k-mer in the k-mer-sorted hit list, bounds[number of present k-mers] = n_hits;
// fhit[r] = the bucket in one word.  A bucket of ONE hit (two k-mers in three occur once in a thousand queries): the hit itself, position
// << f_q_bits | query.  Otherwise bit 31 (kPfMore) and, where they fit, the bucket's start in hit_qp << 8 | its size (start below 2^23,
// size below 256); kPfMore alone sends the lookup to bounds[].  f_q_bits = 0: positions and queries do not fit 31 bits together, every
// bucket is a kPfMore one.  The lookup of nearly every k-mer ends in this 4-byte table (1.3 MB for a thousand queries).
constexpr uint32_t kPfMore = 0x80000000u;
constexpr uint32_t kPfNoBucket = 0xFFFFFFFFu;   // PfBlock::first of a position whose one hit is in hq / hp
__global__ void pf_bucket_bounds(const uint32_t* __restrict__ sorted_key, const uint32_t* __restrict__ sorted_val, const uint32_t* __restrict__ pos_q,
                                 const uint32_t* __restrict__ pos_j, uint64_t n_hits, const uint2* __restrict__ slots, uint32_t f_q_bits,
                                 uint32_t pack_limit, uint32_t* __restrict__ bounds, uint32_t* __restrict__ fhit) {      // pack_limit: 2^23 (0: nothing is packed)
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hits) return;
    const uint32_t v = sorted_key[i];
    const uint32_t r = pf_rank(slots[v >> 5], v);
    if (i == 0 || sorted_key[i - 1] != v) {
        bounds[r] = (uint32_t)i;
        const uint32_t g = sorted_val[i];
        uint32_t cnt = 1;
        while (cnt < 256 && i + cnt < n_hits && sorted_key[i + cnt] == v) ++cnt;
        if (f_q_bits != 0 && cnt == 1) fhit[r] = (pos_j[g] << f_q_bits) | pos_q[g];
        else fhit[r] = kPfMore | (cnt < 256 && i < pack_limit ? ((uint32_t)i << 8) | cnt : 0u);
    }
    if (i + 1 == n_hits) bounds[r + 1] = (uint32_t)n_hits;
}

struct PfChunk {
    const uint8_t* codes;       // residues of the chunk, flat
    const uint64_t* off;        // [nt + 1]
    const uint2* slots;         // [32^k / 32] (presence bits, present k-mers before the word): nine lookups in ten end here
    const uint32_t* bounds;     // [present k-mers + 1] CSR bounds of the queries' k-mers, by rank
    const uint32_t* fhit;       // [present k-mers] first hit of the bucket, packed, or kPfMore (pf_bucket_bounds)
    const uint8_t* letter;      // [32] code -> digit of the index's radix (pf_kmer)
    uint32_t radix, f_q_bits;
    const uint32_t* hit_q;      // bucket contents
    const uint32_t* hit_p;
    const uint64_t* hit_qp;     // the same, interleaved: position << 32 | query (one load per hit in the fused path)
    uint32_t k, nt;
    const uint32_t* tlist;      // optional: the targets to work on are tlist[0 .. nt) (the fused path's overflow); NULL: 0 .. nt
};
__device__ __forceinline__ uint32_t pf_target(const PfChunk& a, uint32_t i) { return a.tlist ? a.tlist[i] : i; }

// number of hits the k-mer at position j of a target contributes, and where its bucket starts
__device__ __forceinline__ uint32_t pf_bucket(const PfChunk& a, const uint8_t* __restrict__ c, uint32_t j, uint32_t n_kmers, uint32_t* first) {
    if (j >= n_kmers) return 0;
    const uint32_t v = pf_kmer(c + j, a.k, a.letter, a.radix);
    if (j != 0 && pf_kmer(c + j - 1, a.k, a.letter, a.radix) == v) return 0;       // database_search.cpp:212: equal to the k-mer before it
    const uint2 slot = a.slots[v >> 5];
    if (!((slot.x >> (v & 31)) & 1u)) return 0;
    const uint32_t r = pf_rank(slot, v);
    const uint32_t s = a.bounds[r];
    *first = s;
    return a.bounds[r + 1] - s;
}

__device__ __forceinline__ uint64_t pf_wave_sum(uint64_t v) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// One wave per target (grid-stride): count[t] = hits of target t of the chunk against all queries.
__global__ __launch_bounds__(256) void pf_target_count(PfChunk a, uint64_t* __restrict__ count, unsigned* __restrict__ bad) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; t < a.nt; t += waves) {
        const uint32_t tt = pf_target(a, t);
        const uint64_t o = a.off[tt];
        const uint32_t len = (uint32_t)(a.off[tt + 1] - o);
        const uint8_t* c = a.codes + o;
        const uint32_t n_kmers = len >= a.k ? len - a.k + 1 : 0;
        uint64_t sum = 0;
        unsigned wrong = 0;
        for (uint32_t base = 0; base < len; base += 64) {
            const uint32_t j = base + lane;
            if (j < len && c[j] >= 32) wrong = 1;
            uint32_t first;
            sum += pf_bucket(a, c, j, n_kmers, &first);
        }
        sum = pf_wave_sum(sum);
        if (lane == 0) count[t] = sum;
        if (wrong) atomicOr(bad, 1u);
    }
}

// One wave per target of [t0, t1): writes its hits at hit_off[t] - base_off .. in the reference's order.  key = (t - t0) << q_bits | query.
__global__ __launch_bounds__(256) void pf_target_emit(PfChunk a, uint32_t t0, uint32_t t1, const uint64_t* __restrict__ hit_off, uint64_t base_off,
                                                      uint32_t q_bits, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t t = t0 + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6); t < t1; t += waves) {
        const uint32_t tt = pf_target(a, t);
        const uint64_t o = a.off[tt];
        const uint32_t len = (uint32_t)(a.off[tt + 1] - o);
        const uint8_t* c = a.codes + o;
        const uint32_t n_kmers = len >= a.k ? len - a.k + 1 : 0;
        uint64_t out = hit_off[t] - base_off;
        const uint64_t hi = (uint64_t)(t - t0) << q_bits;
        for (uint32_t base = 0; base < n_kmers; base += 64) {
            uint32_t first = 0;
            const uint32_t n = pf_bucket(a, c, base + lane, n_kmers, &first);
            uint64_t incl = n;                                          // inclusive scan over the lanes: positions in order
            for (int d = 1; d < 64; d <<= 1) { const uint64_t up = __shfl_up(incl, d, 64); if ((int)lane >= d) incl += up; }
            const uint64_t mine = out + incl - n;
            for (uint32_t h = 0; h < n; ++h) {
                key[mine + h] = hi | a.hit_q[first + h];
                val[mine + h] = a.hit_p[first + h];
            }
            out += __shfl(incl, 63, 64);
        }
    }
}

struct PfLisArgs {
    const uint64_t* seg_key;    // [n_seg] (target - t0) << q_bits | query
    const uint32_t* seg_len;    // [n_seg] hits of the segment
    const uint32_t* seg_off;    // [n_seg] first hit (exclusive scan of seg_len)
    const uint32_t* n_seg;      // device-side count
    const uint32_t* val;        // sorted hits: query positions
    const uint64_t* off;        // targets' offsets
    uint32_t t0, first_id, q_bits;
    uint64_t* cand_key;         // [n_seg] out: query << 32 | ~bits(score)
    uint32_t* cand_val;         // [n_seg] out: database index
    uint32_t* long_list;        // segments left to pf_lis_long
    unsigned* long_count;
    int32_t* tails;             // global fallback of pf_lis_long: n_hits + n_seg entries
    const uint32_t* q_kmers;    // [nq] k-mers of each query: bounds the LIS
    const uint32_t* tlist;      // as PfChunk::tlist
};

__device__ __forceinline__ void pf_candidate(const PfLisArgs& a, uint32_t s, uint32_t lis) {
    const uint64_t k = a.seg_key[s];
    const uint32_t ti = a.t0 + (uint32_t)(k >> a.q_bits);
    const uint32_t t = a.tlist ? a.tlist[ti] : ti;
    const uint32_t q = (uint32_t)(k & ((1ull << a.q_bits) - 1));
    const float score = __fdiv_rn((float)(int32_t)lis, (float)(uint32_t)(a.off[t + 1] - a.off[t]));   // database_search.cpp:228-229
    a.cand_key[s] = ((uint64_t)q << 32) | (uint32_t)~__float_as_uint(score);
    a.cand_val[s] = a.first_id + t;
}

// One thread per segment; up to kPfSmallMax hits in private memory, the bisection of database_search.cpp:262-274 as written.
__global__ void pf_lis_small(PfLisArgs a) {
    const uint32_t n_seg = *a.n_seg;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_seg; s += gridDim.x * blockDim.x) {
        const uint32_t n = a.seg_len[s];
        if (n == 1) { pf_candidate(a, s, 1); continue; }
        if (n > kPfSmallMax) { a.long_list[atomicAdd(a.long_count, 1u)] = s; continue; }
        const uint32_t* v = a.val + a.seg_off[s];
        int32_t tail[kPfSmallMax + 1];                   // tail[l] = smallest last element of an increasing run of length l
        uint32_t best = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const int32_t x = (int32_t)v[i];
            uint32_t l = 1, u = best + 1;                // first l in [1, best + 1] with tail[l] >= x  (tail[best + 1] = +inf)
            while (u > l) { const uint32_t mid = (l + u) / 2; if (tail[mid] < x) l = mid + 1; else u = mid; }
            tail[l] = x;
            best = l > best ? l : best;
        }
        pf_candidate(a, s, best);
    }
}

// One wave per long segment.  tails[1 .. len] ascending; for each hit x in order: c = #{tails < x} by a 64-way search, then
// tails[c + 1] = x.  Tails live in LDS while min(hits, k-mers of the query) fits, else in the global scratch of the segment.
__global__ __launch_bounds__(64) void pf_lis_long(PfLisArgs a) {
    __shared__ int32_t lds_tails[kPfLdsTails + 1];
    const uint32_t lane = threadIdx.x;
    const uint32_t n_long = *a.long_count;
    // segments are dealt out by block index: everything that steers the loops below is wave-uniform by construction (a work
    // queue filled by lane 0 and read back with readfirstlane was restructured by the compiler into a loop that lane 0 leaves
    // alone -- the other lanes then spin on item 0)
    for (uint32_t w = blockIdx.x; w < n_long; w += gridDim.x) {
        const uint32_t s = a.long_list[w];
        const uint32_t n = a.seg_len[s], first = a.seg_off[s];
        const uint32_t q = (uint32_t)(a.seg_key[s] & ((1ull << a.q_bits) - 1));
        const uint32_t bound = min(n, a.q_kmers[q]);                   // positions are distinct k-mer indices of the query
        const bool in_lds = bound <= kPfLdsTails;
        int32_t* gt = a.tails + (uint64_t)first + s;                    // n + 1 entries of this segment's own
        uint32_t len = 0;
        for (uint32_t base = 0; base < n; base += 64) {
            const int32_t mine = base + lane < n ? (int32_t)a.val[first + base + lane] : 0;
            const uint32_t m = min(64u, n - base);
            for (uint32_t i = 0; i < m; ++i) {
                const int32_t x = __shfl(mine, (int)i, 64);
                uint32_t lo = 0, hi = len;                               // the count lies in [lo, hi]
                while (hi - lo > 64) {
                    const uint32_t step = (hi - lo + 63) / 64;
                    const uint32_t idx = lo + (lane + 1) * step - 1;     // tails index idx + 1
                    bool less = false;
                    if (idx < hi) {
                        const int32_t tv = in_lds ? lds_tails[idx + 1] : __hip_atomic_load(gt + idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        less = tv < x;
                    }
                    const uint32_t c = (uint32_t)__popcll(__ballot(less));
                    const uint32_t nlo = lo + c * step;
                    hi = min(hi, nlo + step - 1);
                    lo = nlo;
                }
                {
                    const uint32_t idx = lo + lane;
                    bool less = false;
                    if (idx < hi) {
                        const int32_t tv = in_lds ? lds_tails[idx + 1] : __hip_atomic_load(gt + idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        less = tv < x;
                    }
                    lo += (uint32_t)__popcll(__ballot(less));
                }
                if (lane == 0) {
                    if (in_lds) lds_tails[lo + 1] = x;
                    else __hip_atomic_store(gt + lo + 1, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the next hit's search reads what lane 0 wrote
                len = max(len, lo + 1);
            }
        }
        pf_candidate(a, s, len);                                         // every lane writes the same words
    }
}

// =====================================================================================================================
// The fused path.
constexpr uint32_t kPfCap = 512;                // hits a wave groups in LDS at a time (6 KB per wave)
constexpr uint32_t kPfOrderBits = 9;            // log2(kPfCap): a sort key is query << kPfOrderBits | position in the reference's order
constexpr uint32_t kPfMaxLong = 16;             // long segments (more than kPfSmallMax hits) kPfCap hits can hold
constexpr uint32_t kPfRanges = 64;              // a target with more hits is walked once per RANGE OF QUERIES: histogram of its hits over
                                                // kPfRanges equal slices of the query indices, slices grouped into passes of <= kPfCap hits

constexpr uint32_t kPfRegions = 64;            // the candidate array is cut into this many regions, each with a fill counter of its own
                                                // (a line apart): 100 000 waves reserving slots through ONE counter spent a quarter of the
                                                // kernel waiting for that atomic

// -DPF_TIMING: shader clocks per section of pf_target_lis, summed over the waves (a measuring build: `make pf-timing`, tools/README.md):
// [0] index lookups of a target's first 256 positions (issued ahead), [1] collecting the hits into LDS (+ lookups beyond 256), [2] the
// sort, [3] segment count + slot reservation, [4] the LIS of the short segments + candidates out, [5] long segments, [6] targets walked
// once per range of queries, [7] everything else
#ifdef PF_TIMING
__device__ unsigned long long pf_times[8];
struct PfClock {
    unsigned long long t, acc[8];
    __device__ __forceinline__ void start() { t = __builtin_readcyclecounter(); for (int i = 0; i < 8; ++i) acc[i] = 0; }
    __device__ __forceinline__ void mark(int i) { const unsigned long long n = __builtin_readcyclecounter(); acc[i] += n - t; t = n; }
    __device__ __forceinline__ void flush(uint32_t lane) { if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&pf_times[i], acc[i]); }
};
#else
struct PfClock {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(uint32_t) {}
};
#endif

struct PfFused {
    PfChunk c;
    uint32_t first_id, nq;
    uint32_t* cand_q;                   // [kPfRegions * region_cap] out: query (pre-filled with 0xFFFFFFFF: unused slots sort behind every query)
    uint64_t* cand_sv;                  // [kPfRegions * region_cap] out: ~bits(score) << 32 | database index  (ascending = best first)
    uint64_t region_cap;
    unsigned long long* counters;       // [8 * r] candidates of region r (counted even beyond region_cap: the host then repeats the chunk
                                        // with room); [8 * kPfRegions + 0] hits, [+1] segments of more than kPfSmallMax hits,
                                        // [+2] targets left to the general path
    uint32_t* overflow;                 // [nt] those targets: one slice of queries alone has more than kPfCap hits there
    unsigned* bad;                      // residue code beyond 5 bits seen
};

// LDS traffic of ONE wave is executed in program order, so what a lane wrote is there for any lane's later read; all this has to do is
// keep the compiler from moving LDS accesses across it (and wait for the LDS queue: cheap).  A seq_cst fence would also wait for every
// global store in flight -- the candidates just written -- which is microseconds per target.
__device__ __forceinline__ void pf_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// inclusive prefix sum over the 64 lanes by DPP: rows of 16 by shifted adds (zeros shifted in), then the row totals handed on
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3): six dependent adds, no LDS crossbar, no waits
__device__ __forceinline__ uint32_t pf_scan64(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);       // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);       // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);       // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);       // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    return v;
}

// Buckets of four chunks of 64 positions from `base`: lane l gets those of positions base + 64 u + l (u = 0..3).  One byte per
// lane and chunk comes from memory, the k-mers are put together from the neighbours' digits (shuffles), and the dependent rounds of
// loads -- residues, index slot, the bucket's word -- are each issued for all four chunks before any is waited for.  n: hits of the
// position; first: kPfNoBucket and the hit in hq / hp, or where the bucket starts in hit_qp (pf_collect fetches those hits, all lanes
// at once).  letter: the radix digits of the 32 codes, in LDS.
struct PfBlock { uint32_t first[4], n[4], hq[4], hp[4]; };
__device__ __forceinline__ void pf_buckets4(const PfChunk& a, const uint8_t* __restrict__ c, uint32_t len, uint32_t n_kmers, uint32_t base,
                                            uint32_t lane, PfBlock& blk, unsigned& wrong, const uint8_t* letter) {
    uint32_t (&first)[4] = blk.first; uint32_t (&n)[4] = blk.n; uint32_t (&hq)[4] = blk.hq; uint32_t (&hp)[4] = blk.hp;
    uint32_t carry = base ? pf_kmer(c + base - 1, a.k, letter, a.radix) : 0xFFFFFFFFu;   // the k-mer before this block's first (the "equal to its predecessor" rule)
    uint32_t b[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) { const uint32_t j = base + 64 * u + lane; b[u] = (j < len && (u < 4 || lane < 8)) ? c[j] : 0u; }
#pragma unroll
    for (int u = 0; u < 5; ++u) { if (b[u] >= 32) wrong = 1; b[u] = letter[b[u] & 31u]; }
    uint32_t v[4];
    bool look[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        uint32_t s = b[u], x = s;                       // digit of position j + i: lane + i of this chunk, or of the next one --
        for (uint32_t i = 1; i < a.k; ++i) {            // the chunk shifted down a lane at a time (DPP wave_shl:1), the next chunk's digits entering at lane 63
            const int in = __builtin_amdgcn_readlane((int)b[u + 1], (int)(i - 1));
            s = (uint32_t)__builtin_amdgcn_update_dpp(in, (int)s, 0x130, 0xf, 0xf, false);
            x = x * a.radix + s;
        }
        v[u] = x;
        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)x, 0x138, 0xf, 0xf, false);      // wave_shr:1, lane 0 <- carry
        carry = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
        const uint32_t j = base + 64 * u + lane;
        look[u] = j < n_kmers && !(j != 0 && prev == x);            // database_search.cpp:212: equal to the k-mer before it
    }
    uint2 slot[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) slot[u] = look[u] ? a.slots[v[u] >> 5] : make_uint2(0u, 0u);
    uint32_t r[4], f[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        look[u] = look[u] && ((slot[u].x >> (v[u] & 31)) & 1u);
        r[u] = pf_rank(slot[u], v[u]);
        f[u] = look[u] ? a.fhit[r[u]] : 0u;
    }
    const uint32_t q_mask = (1u << a.f_q_bits) - 1u;
    bool unpacked = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        first[u] = kPfNoBucket; n[u] = look[u] ? 1u : 0u;
        hq[u] = f[u] & q_mask; hp[u] = (f[u] & ~kPfMore) >> a.f_q_bits;
        if (f[u] & kPfMore) { first[u] = (f[u] & ~kPfMore) >> 8; n[u] = f[u] & 255u; unpacked = unpacked || n[u] == 0; }
    }
    if (__ballot(unpacked)) {                                        // (uniform: the rare round -- buckets of 256 hits and more -- is skipped by the whole wave)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if ((f[u] & kPfMore) && n[u] == 0) {
                uint64_t se = 0;                                     // bounds[r], bounds[r + 1]: one 8-byte load (4-byte aligned)
                __builtin_memcpy(&se, a.bounds + r[u], 8);
                first[u] = (uint32_t)se; n[u] = (uint32_t)(se >> 32) - (uint32_t)se;
            }
    }
}

// hit h of a position's bucket
__device__ __forceinline__ uint64_t pf_hit(const PfChunk& c0, const PfBlock& blk, int u, uint32_t h) {
    return blk.first[u] != kPfNoBucket ? c0.hit_qp[blk.first[u] + h] : (((uint64_t)blk.hp[u] << 32) | blk.hq[u]);
}

// One sweep over a target: its hits with a query in [qa, qb) (FILTER) or all of them go into key / pos in the order the reference
// appends them (by target position, then bucket order) -- as long as they fit: once kPfCap would be passed nothing more is written
// and only the count goes on.  -> the number of such hits (the same in every lane); *fits = all of them are in LDS.
constexpr uint32_t kPfFetch = 0x80000000u;       // pos[] of a hit that is still to be fetched: kPfFetch | its index in hit_qp
template <bool FILTER>
__device__ __forceinline__ uint64_t pf_collect(const PfChunk& c0, const uint8_t* __restrict__ c, uint32_t len, uint32_t n_kmers, uint32_t lane,
                                               uint32_t qa, uint32_t qb, uint32_t* key, uint32_t* pos, bool* fits, unsigned& wrong,
                                               const PfBlock* block0, const uint8_t* letter) {      // block0: the buckets of positions 0 .. 255, if the caller has them
    uint64_t out = 0;
    bool ok = true;
    for (uint32_t base = 0; base < n_kmers; base += 256) {
        PfBlock blk;
        if (base == 0 && block0) blk = *block0; else pf_buckets4(c0, c, len, n_kmers, base, lane, blk, wrong, letter);
        uint32_t (&first)[4] = blk.first; uint32_t (&n)[4] = blk.n; uint32_t (&hq)[4] = blk.hq; uint32_t (&hp)[4] = blk.hp;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (base + 64 * u >= n_kmers) break;    // (uniform)
            uint32_t mine_n = n[u];                 // this position's hits inside the query range (a bucket is ordered by query)
            if (FILTER) { mine_n = 0; for (uint32_t h = 0; h < n[u]; ++h) { const uint32_t q = (uint32_t)pf_hit(c0, blk, u, h); mine_n += (q >= qa && q < qb); } }
            const uint32_t incl = pf_scan64(mine_n);
            const uint32_t chunk = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            ok = ok && out + chunk <= kPfCap;
            if (ok && chunk) {
                uint32_t w = (uint32_t)out + incl - mine_n;
                if (FILTER) {
                    for (uint32_t h = 0; h < n[u]; ++h) {
                        const uint64_t qp = pf_hit(c0, blk, u, h);
                        const uint32_t q = (uint32_t)qp;
                        if (q < qa || q >= qb) continue;
                        key[w] = (q << kPfOrderBits) | w;
                        pos[w] = (uint32_t)(qp >> 32);
                        ++w;
                    }
                } else if (first[u] == kPfNoBucket) {
                    if (n[u]) { key[w] = (hq[u] << kPfOrderBits) | w; pos[w] = hp[u]; }
                } else {
                    // a bucket of several hits: only WHERE its hits are goes into LDS now (a lane that fetched its own bucket hit by hit would
                    // wait for memory once per hit, and the wave for its longest bucket); pf_fetch_hits then brings them in 64 at a time
                    for (uint32_t h = 0; h < n[u]; ++h) pos[w + h] = kPfFetch | (first[u] + h);
                }
            }
            out += chunk;
        }
    }
    *fits = ok;
    return out;
}
// The hits pf_collect<false> left as references, fetched by all lanes at once -- in two halves, so that the lookups of the wave's next
// target can be issued between them: pf_fetch_issue puts the loads of the first 128 slots in flight, pf_fetch_land writes them (and
// fetches whatever lies beyond 128) where the hits belong.
struct PfFetch { uint64_t qp[2]; bool on[2]; };
__device__ __forceinline__ void pf_fetch_issue(const PfChunk& c0, uint32_t total, uint32_t lane, const uint32_t* pos, PfFetch& f) {
    pf_lds_fence();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t k = r * 64 + lane;
        const uint32_t ref = k < total ? pos[k] : 0u;
        f.on[r] = (ref & kPfFetch) != 0;
        f.qp[r] = f.on[r] ? c0.hit_qp[ref & ~kPfFetch] : 0ull;
    }
}
__device__ __forceinline__ void pf_fetch_land(const PfChunk& c0, uint32_t total, uint32_t lane, uint32_t* key, uint32_t* pos, const PfFetch& f) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t k = r * 64 + lane;
        if (f.on[r]) { key[k] = ((uint32_t)f.qp[r] << kPfOrderBits) | k; pos[k] = (uint32_t)(f.qp[r] >> 32); }
    }
    for (uint32_t k = 128 + lane; k < total; k += 64) {
        const uint32_t ref = pos[k];
        if (ref & kPfFetch) {
            const uint64_t qp = c0.hit_qp[ref & ~kPfFetch];
            key[k] = ((uint32_t)qp << kPfOrderBits) | k;
            pos[k] = (uint32_t)(qp >> 32);
        }
    }
}

// ---- bitonic sort of 64 NREG keys in registers: key i in lane i % 64 of v[i / 64] ----
// A compare-exchange over a stride of 64 or more pairs two registers of the same lane (min / max); below 64 the partner comes across the
// lanes without touching LDS memory -- DPP quad_perm (1, 2), two masked DPP row shifts (4, 8), ds_swizzle (16), ds_bpermute (32) -- and
// the lane keeps its own key or the partner's by one compare and one select under a CONSTANT lane mask (which lanes keep the smaller
// key is a property of stride, size and register): 3 vector + 2 scalar instructions per register and stage, all 64 lanes at work, where
// the sort in LDS spent two reads, two writes and a wait per stage on half the lanes.
template <int STRIDE>
__device__ __forceinline__ uint32_t pf_xor_lane(uint32_t v, uint32_t lane) {
    if constexpr (STRIDE == 1) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false);       // quad_perm:[1,0,3,2]
    else if constexpr (STRIDE == 2) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false);  // quad_perm:[2,3,0,1]
    else if constexpr (STRIDE == 4) {
        const int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xf, 0x5, false);                                  // row_shl:4 into banks 0, 2
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xf, 0xa, false);                                    // row_shr:4 into banks 1, 3
    } else if constexpr (STRIDE == 8) {
        const int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x108, 0xf, 0x3, false);                                  // row_shl:8 into banks 0, 1
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x118, 0xf, 0xc, false);                                    // row_shr:8 into banks 2, 3
    } else if constexpr (STRIDE == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);                        // swizzle(SWAP, 16)
    else return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ 32u) << 2), (int)v);
}
constexpr uint64_t pf_low_lanes(int s) {         // lanes whose bit s is clear (s = 1 .. 32)
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) if (!(l & s)) m |= 1ull << l;
    return m;
}
template <int NREG, int SIZE, int STRIDE>
__device__ __forceinline__ void pf_bitonic_step(uint32_t (&v)[NREG], uint32_t lane) {
    if constexpr (STRIDE >= 64) {
        constexpr int SR = STRIDE / 64;
#pragma unroll
        for (int r = 0; r < NREG; ++r)
            if (!(r & SR)) {
                const uint32_t a = v[r], b = v[r | SR];
                const bool up = ((r * 64) & SIZE) == 0;
                v[r] = up ? min(a, b) : max(a, b);
                v[r | SR] = up ? max(a, b) : min(a, b);
            }
    } else {
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            // the lanes that keep the smaller key: (bit STRIDE of the index clear) == (ascending block: bit SIZE of the index clear)
            const uint64_t up = SIZE < 64 ? pf_low_lanes(SIZE) : (((r * 64) & SIZE) == 0 ? ~0ull : 0ull);
            const uint64_t keep_min = ~(pf_low_lanes(STRIDE) ^ up);
            const uint32_t p = pf_xor_lane<STRIDE>(v[r], lane);
            const uint64_t take = ~(__ballot(p < v[r]) ^ keep_min);          // the partner's key: it is smaller and this lane keeps the smaller, or neither
            v[r] = __builtin_amdgcn_inverse_ballot_w64(take) ? p : v[r];     // (equal keys -- the padding -- may go either way)
        }
    }
    if constexpr (STRIDE > 1) pf_bitonic_step<NREG, SIZE, STRIDE / 2>(v, lane);
}
template <int NREG, int SIZE>
__device__ __forceinline__ void pf_bitonic_from(uint32_t (&v)[NREG], uint32_t lane) {
    pf_bitonic_step<NREG, SIZE, SIZE / 2>(v, lane);
    if constexpr (SIZE < 64 * NREG) pf_bitonic_from<NREG, SIZE * 2>(v, lane);
}
// key[0 .. total) sorted ascending in place, key[total .. 64 NREG) = 0xFFFFFFFF behind them
template <int NREG>
__device__ __forceinline__ void pf_sort_keys(uint32_t* key, uint32_t total, uint32_t lane) {
    uint32_t v[NREG];
#pragma unroll
    for (int r = 0; r < NREG; ++r) v[r] = r * 64 + lane < total ? key[r * 64 + lane] : 0xFFFFFFFFu;
    pf_bitonic_from<NREG, 2>(v, lane);
#pragma unroll
    for (int r = 0; r < NREG; ++r) key[r * 64 + lane] = v[r];
}

// cnt hits (lanes 0 .. cnt - 1 of xs, in order) through the tails method with the tails in registers: tail l in lane l % 64 of T[l / 64],
// unused tails at +infinity.  NR = registers that can come into play during this chunk (llen + cnt <= 64 NR): per hit one broadcast,
// NR compares + scalar popcounts, one predicated move -- nothing of it touches LDS or waits for memory.
template <int NR>
__device__ __forceinline__ void pf_lis_chunk(int32_t (&T)[8], uint32_t& llen, int32_t xs, uint32_t cnt, uint32_t lane) {
    for (uint32_t k = 0; k < cnt; ++k) {
        const int32_t x = __builtin_amdgcn_readlane(xs, (int)k);
        uint32_t lo = 0;                                                    // tails below x (tails ascending; +infinity is below nothing)
#pragma unroll
        for (int r = 0; r < NR; ++r) lo += (uint32_t)__popcll(__ballot(T[r] < x));
        const uint32_t reg = lo >> 6;                                       // (uniform)
#pragma unroll
        for (int r = 0; r < NR; ++r) if ((NR == 1 || reg == (uint32_t)r) && lane == (lo & 63)) T[r] = x;
        llen = max(llen, lo + 1);
    }
}

// The `total` hits in key / pos: sorted by (query, order), cut into segments, the LIS of every segment, one candidate each.
__device__ __forceinline__ void pf_process(const PfFused& a, uint32_t total, uint32_t lane, uint32_t region, uint32_t* key, const uint32_t* pos,
                                           int32_t* tail, uint32_t* lng, float flen, uint32_t id, unsigned long long& sum_long, PfClock& ck) {
    // ---- the keys sorted (in registers: pf_sort_keys): grouped by query, the reference's order inside a query ----
    pf_lds_fence();
    if (total <= 64) pf_sort_keys<1>(key, total, lane);
    else if (total <= 128) pf_sort_keys<2>(key, total, lane);
    else if (total <= 256) pf_sort_keys<4>(key, total, lane);
    else pf_sort_keys<8>(key, total, lane);
    pf_lds_fence();
    ck.mark(2);
    // ---- segments = runs of one query; count them, reserve their candidates with one atomic on the wave's region ----
    uint32_t n_seg = 0;
    for (uint32_t base = 0; base < total; base += 64) {
        const uint32_t i = base + lane;
        const bool start = i < total && (i == 0 || (key[i] >> kPfOrderBits) != (key[i - 1] >> kPfOrderBits));
        n_seg += (uint32_t)__popcll(__ballot(start));
    }
    unsigned long long slot0 = 0;
    if (lane == 0) { slot0 = atomicAdd(&a.counters[8 * region], (unsigned long long)n_seg); lng[0] = 0; }
    slot0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(slot0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)slot0);
    pf_lds_fence();
    const uint64_t region_base = (uint64_t)region * a.region_cap;
    auto emit = [&](uint32_t q, uint32_t lis, unsigned long long slot) {
        if (slot >= a.region_cap) return;
        const float score = __fdiv_rn((float)(int32_t)lis, flen);                          // database_search.cpp:228-229
        a.cand_q[region_base + slot] = q;
        a.cand_sv[region_base + slot] = ((uint64_t)(uint32_t)~__float_as_uint(score) << 32) | id;
    };
    ck.mark(3);
    // ---- the LIS of every segment: a lane per segment of up to kPfSmallMax hits ----
    uint32_t rank0 = 0;
    for (uint32_t base = 0; base < total; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t q = i < total ? key[i] >> kPfOrderBits : 0xFFFFFFFFu;
        const bool start = i < total && (i == 0 || q != (key[i - 1] >> kPfOrderBits));
        const unsigned long long bal = __ballot(start);
        const uint32_t rank = rank0 + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        rank0 += (uint32_t)__popcll(bal);
        uint32_t n_mine = 0;                                     // hits of the segment that starts in this lane (0: none does)
        if (start) {
            // length: up to the next segment start -- in this chunk of 64 by the ballot, beyond it by looking
            const unsigned long long later = lane < 63 ? bal >> (lane + 1) : 0ull;
            uint32_t n;
            if (later) n = (uint32_t)__ffsll((long long)later);
            else { n = min(64u - lane, total - i); while (i + n < total && (key[i + n] >> kPfOrderBits) == q) ++n; }
            n_mine = n;
        }
        // (one copy of each path and of the candidate's division and stores for the whole chunk, each behind a uniform test)
        const uint32_t n_short = n_mine <= 8 ? n_mine : 0;                     // 2 .. 8 hits: in registers
        uint32_t lis = 1;
        const unsigned long long any_short = __ballot(n_short >= 2);
        if (any_short) {
            // the hits in registers (independent LDS reads), then the quadratic recurrence without touching LDS again:
            // L[i] = 1 + max{L[j]: j < i, x[j] < x[i]} -- the same longest strictly increasing run as the tails method finds.
            // Unrolled to eight, cut off (uniformly) at the longest such segment of the chunk.
            uint32_t reach = n_short;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) reach = max(reach, (uint32_t)__shfl_xor((int)reach, d, 64));
            reach = (uint32_t)__builtin_amdgcn_readfirstlane((int)reach);
            int32_t x[8];
            uint32_t L[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = (uint32_t)k < n_short ? (int32_t)pos[key[i + k] & (kPfCap - 1)] : 0x7FFFFFFF;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if ((uint32_t)k >= reach) break;
                uint32_t l = 1;
#pragma unroll
                for (int j = 0; j < k; ++j) if (x[j] < x[k] && L[j] + 1 > l) l = L[j] + 1;
                L[k] = l;
                if ((uint32_t)k < n_short && l > lis) lis = l;
            }
        }
        if (n_mine > kPfSmallMax) {
            const uint32_t w = atomicAdd(&lng[0], 1u);          // (at most kPfCap / (kPfSmallMax + 1) < kPfMaxLong of them)
            lng[1 + 3 * w] = i; lng[2 + 3 * w] = n_mine; lng[3 + 3 * w] = rank;
        } else if (n_mine > 8) {
            int32_t* T = tail + i;                               // T[l] = smallest last element of an increasing run of length l + 1
            uint32_t best = 0;
            for (uint32_t k = 0; k < n_mine; ++k) {
                const int32_t xk = (int32_t)pos[key[i + k] & (kPfCap - 1)];
                uint32_t l = 0, u = best;                        // first l in [0, best] with T[l] >= x  (T[best] = +inf)
                while (u > l) { const uint32_t mid = (l + u) / 2; if (T[mid] < xk) l = mid + 1; else u = mid; }
                T[l] = xk;
                best = l + 1 > best ? l + 1 : best;
            }
            lis = best;
        }
        if (n_mine >= 1 && n_mine <= kPfSmallMax) emit(q, lis, slot0 + rank);
    }
    pf_lds_fence();
    ck.mark(4);
    // ---- long segments, the whole wave on each.  The tails live in REGISTERS (pf_lis_chunk; eight registers: kPfCap tails): no LDS
    //      round trip inside the dependent chain (with tails in LDS a hit cost ~300 clocks), and only as many registers are searched as
    //      the run found so far plus the chunk at hand can reach (most runs stay below 64: 34 -> 7 instructions per hit) ----
    const uint32_t n_long = lng[0];
    sum_long += n_long;
    for (uint32_t w = 0; w < n_long; ++w) {
        const uint32_t i0 = lng[1 + 3 * w], n = lng[2 + 3 * w], rank = lng[3 + 3 * w];
        int32_t T[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) T[r] = 0x7FFFFFFF;
        uint32_t llen = 0;
        for (uint32_t base = 0; base < n; base += 64) {
            const int32_t xs = base + lane < n ? (int32_t)pos[key[i0 + base + lane] & (kPfCap - 1)] : 0;     // 64 hits of the segment at a time
            const uint32_t cnt = min(64u, n - base);
            const uint32_t reach = llen + cnt;                                       // (uniform)
            if (reach <= 64) pf_lis_chunk<1>(T, llen, xs, cnt, lane);
            else if (reach <= 128) pf_lis_chunk<2>(T, llen, xs, cnt, lane);
            else if (reach <= 256) pf_lis_chunk<4>(T, llen, xs, cnt, lane);
            else pf_lis_chunk<8>(T, llen, xs, cnt, lane);
        }
        if (lane == 0) emit(key[i0] >> kPfOrderBits, llen, slot0 + rank);
    }
    pf_lds_fence();
    ck.mark(5);
}

// One wave per target.  See the header of this file.
// (Four waves per SIMD at ~112 VGPRs, no spills.  At five and six -- 36 and 84 bytes of scratch -- the kernel takes the same time:
// it is bound by vector issue, not by latency.  -DPF_WPE=n rebuilds it for n.)
#ifndef PF_WPE
#define PF_WPE 4
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PF_WPE, PF_WPE))) void pf_target_lis(PfFused a) {
    __shared__ uint32_t s_key[4][kPfCap];       // query << 9 | order, sorted in place
    __shared__ uint32_t s_pos[4][kPfCap];       // query position of the hit with order index i
    __shared__ int32_t s_tail[4][kPfCap];       // LIS tails: a segment that starts at sorted position i uses [i, i + its length)
    __shared__ uint32_t s_misc[4][kPfMaxLong * 3 + 1 + kPfRanges];     // [0] count, then (start, length, rank) of long segments; the range histogram
    __shared__ uint8_t s_letter[32];
    if (threadIdx.x < 32) s_letter[threadIdx.x] = a.c.letter[threadIdx.x];
    __syncthreads();
    const uint8_t* letter = s_letter;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t* key = s_key[wv];
    uint32_t* pos = s_pos[wv];
    int32_t* tail = s_tail[wv];
    uint32_t* lng = s_misc[wv];
    uint32_t* hist = s_misc[wv] + kPfMaxLong * 3 + 1;
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t region = wave_id % kPfRegions;
    const PfChunk& c0 = a.c;
    const uint32_t q_per_range = (a.nq + kPfRanges - 1) / kPfRanges;
    unsigned long long sum_hits = 0, sum_long = 0;
    unsigned wrong = 0;
    // The three dependent rounds of loads of a target's first 256 positions are issued BEFORE the previous target's hits are sorted and
    // walked, so they are there when its turn comes (three targets in four are no longer than that).
    PfBlock ahead;
    uint64_t o_next = 0;
    uint32_t len_next = 0;
    PfClock ck;
    ck.start();
    // Targets are dealt out by counters, not by wave index: a target costs from a tenth to ten times the average, and with two dozen
    // targets per wave fixed in advance the slowest of 4096 waves had half as much again as the mean.  One counter for all waves would
    // serialise 100 000 returning atomics on one address (measured: the kernel doubled); the waves of a region share one instead (64
    // counters, a line apart, beside the regions' fill counters) that deals out the targets region, region + 64, ... -- all regions walk
    // the database front to back together.  The index a wave will work on two targets from now is drawn while the current one is
    // processed and only looked at an iteration later, so the atomic's latency is never waited for.
    const uint32_t groups = min(kPfRegions, gridDim.x * (blockDim.x >> 6));      // (a small chunk runs on fewer waves than there are regions)
    auto draw = [&]() -> unsigned long long {
        unsigned long long v = 0;
        if (lane == 0) v = atomicAdd(&a.counters[8 * region + 1], 1ull);
        return v;
    };
    auto target_of = [&](unsigned long long drawn) -> uint32_t {
        const unsigned long long i = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(drawn >> 32)) << 32) |
                                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)drawn);
        return (uint32_t)min(i * groups + region, (unsigned long long)0xFFFFFFFFull);
    };
    uint32_t t = target_of(draw()), t_next = target_of(draw());
    unsigned long long drawn = 0;
    if (t < c0.nt) {
        o_next = c0.off[t]; len_next = (uint32_t)(c0.off[t + 1] - o_next);
        pf_buckets4(c0, c0.codes + o_next, len_next, len_next >= c0.k ? len_next - c0.k + 1 : 0, 0, lane, ahead, wrong, letter);
    }
    for (; t < c0.nt; t = t_next, t_next = target_of(drawn)) {
        drawn = draw();
        const uint64_t o = o_next;
        const uint32_t len = len_next;
        const uint8_t* c = c0.codes + o;
        const uint32_t n_kmers = len >= c0.k ? len - c0.k + 1 : 0;
        for (uint32_t j = (n_kmers + 255) / 256 * 256 + lane; j < len; j += 64) if (c[j] >= 32) wrong = 1;      // (the tail the k-mer sweep does not load)
        const float flen = (float)len;
        const uint32_t id = a.first_id + t;
        bool fits = false;
        const PfBlock mine_block = ahead;
        ck.mark(7);
        const uint64_t total64 = pf_collect<false>(c0, c, len, n_kmers, lane, 0u, 0xFFFFFFFFu, key, pos, &fits, wrong, n_kmers ? &mine_block : nullptr, letter);
        PfFetch fetch;
        if (fits) pf_fetch_issue(c0, (uint32_t)total64, lane, pos, fetch);
        ck.mark(1);
        if (t_next < c0.nt) {
            o_next = c0.off[t_next]; len_next = (uint32_t)(c0.off[t_next + 1] - o_next);
            pf_buckets4(c0, c0.codes + o_next, len_next, len_next >= c0.k ? len_next - c0.k + 1 : 0, 0, lane, ahead, wrong, letter);
        }
        ck.mark(0);
        if (total64 == 0) continue;
        sum_hits += total64;
        if (fits) { pf_fetch_land(c0, (uint32_t)total64, lane, key, pos, fetch); pf_process(a, (uint32_t)total64, lane, region, key, pos, tail, lng, flen, id, sum_long, ck); continue; }
        // ---- more hits than a wave keeps: once per range of queries.  Histogram of the hits over kPfRanges slices of the query
        //      indices, consecutive slices grouped into passes of at most kPfCap hits ----
        pf_lds_fence();
        hist[lane] = 0;
        pf_lds_fence();
        for (uint32_t base = 0; base < n_kmers; base += 256) {
            PfBlock blk;
            pf_buckets4(c0, c, len, n_kmers, base, lane, blk, wrong, letter);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                for (uint32_t h = 0; h < blk.n[u]; ++h) atomicAdd(&hist[(uint32_t)pf_hit(c0, blk, u, h) / q_per_range], 1u);
        }
        pf_lds_fence();
        if (__ballot(hist[lane] > kPfCap)) {                            // one slice of queries alone does not fit: the general passes take the target
            if (lane == 0) a.overflow[atomicAdd(&a.counters[8 * kPfRegions + 2], 1ull)] = t;
            continue;
        }
        for (uint32_t r0 = 0; r0 < kPfRanges;) {
            uint32_t acc = 0, r1 = r0;                                  // as many slices as fit (uniform: every lane reads the same words)
            while (r1 < kPfRanges && acc + hist[r1] <= kPfCap) acc += hist[r1++];
            const uint32_t qa = r0 * q_per_range, qb = r1 * q_per_range;
            r0 = r1;
            if (acc == 0) continue;
            const uint64_t tot = pf_collect<true>(c0, c, len, n_kmers, lane, qa, qb, key, pos, &fits, wrong, nullptr, letter);
            if (tot) pf_process(a, (uint32_t)tot, lane, region, key, pos, tail, lng, flen, id, sum_long, ck);
        }
        ck.mark(6);
    }
    ck.mark(7);
    ck.flush(lane);
    if (lane == 0) {
        if (sum_hits) atomicAdd(&a.counters[8 * kPfRegions + 0], sum_hits);
        if (sum_long) atomicAdd(&a.counters[8 * kPfRegions + 1], sum_long);
    }
    if (wrong) atomicOr(a.bad, 1u);
}

// candidates of the general path (key = query << 32 | ~score, value = id) appended to the fused path's arrays
__global__ void pf_append(const uint64_t* __restrict__ key, const uint32_t* __restrict__ val, uint32_t n, uint32_t* __restrict__ cand_q,
                          uint64_t* __restrict__ cand_sv, uint64_t at, uint64_t cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || at + i >= cap) return;
    cand_q[at + i] = (uint32_t)(key[i] >> 32);
    cand_sv[at + i] = (key[i] << 32) | val[i];
}

// bitonic sort of n (a power of two) 64-bit keys in LDS by the whole block, ascending
__device__ __forceinline__ void pf_block_sort(uint64_t* keys, uint32_t n) {
    for (uint32_t size = 2; size <= n; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t i = threadIdx.x; i < n / 2; i += blockDim.x) {
                const uint32_t lo = 2 * i - (i & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t x = keys[lo], y = keys[hi];
                if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
            }
            __syncthreads();
        }
}

// The want-th smallest of the n distinct 64-bit keys in LDS (1 <= want <= n): a radix select, most significant byte first -- per
// round a 256-bin histogram of the keys that still match the bytes decided so far picks the bin the want-th smallest lies in.
// Whole block (at least 256 threads); hist: 256 words, ctl: 8 words of LDS.
// The keys are ~score << 32 | id with scores of one narrow range: in the leading rounds nearly every key falls into one bin, and 64 lanes
// adding to one LDS word take 64 turns -- a wave whose lanes all hold the same bin adds its count once.  The bin is found by a prefix sum
// over the 256 counters held by 256 threads (one thread walking them paid an LDS round trip per bin: 10 us per round, 24 rounds per add).
__device__ __forceinline__ uint64_t pf_block_kth(const uint64_t* keys, uint32_t n, uint32_t want, uint32_t* hist, uint32_t* ctl) {
    uint64_t prefix = 0;                         // the bytes of the answer decided so far
    uint32_t remaining = want;                   // its rank among the keys that match them (1-based)
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (uint32_t i0 = 0; i0 < n; i0 += blockDim.x) {
            const uint32_t i = i0 + threadIdx.x;
            const uint64_t k = i < n ? keys[i] : 0;
            const bool in = i < n && (shift == 56 || (k >> (shift + 8)) == (prefix >> (shift + 8)));
            const uint32_t bin = (uint32_t)(k >> shift) & 255u;
            const unsigned long long live = __ballot(in);
            if (!live) continue;
            const uint32_t lead = (uint32_t)__builtin_amdgcn_readlane((int)bin, (int)__ffsll((long long)live) - 1);
            const unsigned long long same = __ballot(in && bin == lead);
            if (same == live) { if (lane == (uint32_t)__ffsll((long long)live) - 1) atomicAdd(&hist[lead], (uint32_t)__popcll(live)); }
            else if (in) atomicAdd(&hist[bin], 1u);
        }
        __syncthreads();
        uint32_t mine = 0, incl = 0;
        if (threadIdx.x < 256) {
            mine = hist[threadIdx.x];
            incl = pf_scan64(mine);
            if (lane == 63) ctl[4 + wv] = incl;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            for (uint32_t w = 0; w < wv; ++w) incl += ctl[4 + w];
            if (incl - mine < remaining && remaining <= incl) { ctl[0] = threadIdx.x; ctl[1] = remaining - (incl - mine); }
        }
        __syncthreads();
        prefix |= (uint64_t)ctl[0] << shift;
        remaining = ctl[1];
        __syncthreads();
    }
    return prefix;
}

// One block per query: list[q] (list_cnt[q] entries, any order) and its new candidates sv[first[q] .. first[q + 1]) -> the best
// max_c of them (smallest keys: score descending, ties to the lower database index -- database_search.cpp:129-150,
// include/s4g_hip.h), in no particular order: s4g_prefilter_finish orders a list when it is asked for.  cap = 64-bit keys the
// block's LDS holds; the new candidates go through it in slices of cap - max_c.  A selection, not a sort: the cut is found by a
// radix select in LDS and the survivors go straight back to the list.
__global__ __launch_bounds__(512) void pf_merge_lists(uint64_t* __restrict__ list, uint32_t* __restrict__ list_cnt, uint32_t max_c,
                                                      const uint64_t* __restrict__ sv, const uint32_t* __restrict__ first, uint32_t cap,
                                                      uint64_t* __restrict__ worst) {    // worst[q] = the list's largest key once it is full (NULL: not kept)
    extern __shared__ __attribute__((aligned(16))) uint64_t pf_keys[];
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_ctl[8];
    const uint32_t q = blockIdx.x;
    const uint32_t b = first[q], e = first[q + 1];
    if (b == e) return;
    uint64_t* mine = list + (uint64_t)q * max_c;
    uint32_t cur = list_cnt[q];
    for (uint32_t at = b; at < e;) {
        const uint32_t take = min(e - at, cap - cur);
        const uint32_t n = cur + take;
        if (n <= max_c) {                        // everything is kept: append
            for (uint32_t i = threadIdx.x; i < take; i += blockDim.x) mine[cur + i] = sv[at + i];
        } else {
            for (uint32_t i = threadIdx.x; i < cur; i += blockDim.x) pf_keys[i] = mine[i];
            for (uint32_t i = threadIdx.x; i < take; i += blockDim.x) pf_keys[cur + i] = sv[at + i];
            __syncthreads();
            const uint64_t kth = pf_block_kth(pf_keys, n, max_c, s_hist, s_ctl);
            if (threadIdx.x == 0) { s_ctl[2] = 0; if (worst) worst[q] = kth; }
            __syncthreads();
            for (uint32_t i0 = 0; i0 < n; i0 += blockDim.x) {             // the keys are distinct: exactly max_c of them pass
                const uint32_t i = i0 + threadIdx.x;
                const uint64_t k = i < n ? pf_keys[i] : ~0ull;
                const bool keep = i < n && k <= kth;
                const unsigned long long m = __ballot(keep);                // one slot reservation per wave, not per key (64 lanes on one LDS word take 64 turns)
                uint32_t base = 0;
                if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(&s_ctl[2], (uint32_t)__popcll(m));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (keep) mine[base + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = k;
            }
        }
        __syncthreads();                          // (the list just written is read back by the next slice)
        cur = min(n, max_c);
        at += take;
    }
    if (threadIdx.x == 0) list_cnt[q] = cur;
}

// ---- candidates into query order by one counting pass (up to kPfScatterMaxQ queries: a block keeps a counter per query in LDS) ----
// The candidates sit in n_regions regions of region_cap slots, the first counters[8 r] of region r used (counters == NULL: one
// region with `single_n` used slots).  A block takes a tile of kPfTile USED slots, counted through the regions in order.
constexpr uint32_t kPfScatterMaxQ = 8192;
constexpr uint32_t kPfTile = 32768;             // a thousand queries: ~32 candidates = 256 bytes per query and tile leave a block together

// use(load(physical index)) for every used slot of this block's tile, eight loads in flight per thread (a loop of load -> LDS atomic
// runs at one memory latency per candidate); pre: LDS words for n_regions + 1 prefix sums
template <typename T, typename L, typename U>
__device__ __forceinline__ void pf_tile_for_each(uint64_t region_cap, const unsigned long long* __restrict__ counters, uint32_t n_regions,
                                                 uint64_t single_n, uint64_t* prefix, L load, U use) {
    if (threadIdx.x == 0) {
        uint64_t acc = 0;
        for (uint32_t r = 0; r < n_regions; ++r) { prefix[r] = acc; acc += counters ? min((uint64_t)counters[8 * r], region_cap) : single_n; }
        prefix[n_regions] = acc;
    }
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * kPfTile, hi = min(prefix[n_regions], lo + kPfTile);
    for (uint32_t r = 0; r < n_regions; ++r) {
        const uint64_t a = max(lo, prefix[r]), b = min(hi, prefix[r + 1]);          // (block-uniform)
        if (a >= b) continue;
        const uint64_t shift = (uint64_t)r * region_cap - prefix[r];
        for (uint64_t i = a + threadIdx.x; i < b; i += 8 * blockDim.x) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i + u * blockDim.x < b) v[u] = load(i + u * blockDim.x + shift);
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i + u * blockDim.x < b) use(v[u]);
        }
    }
}

// A candidate that cannot enter its query's list any more is dropped here, before it is moved: worst[q] = the largest key of query q's
// list once that list is full (pf_merge_lists; ~0 before) -- lists only ever improve, so a key beyond it is beyond the final cut too.
// (In a stream of many chunks the lists fill up with the first and nearly every later candidate ends here.)
struct PfQS { uint32_t q; uint64_t sv; };

// q_count[q] += candidates of query q (LDS: a counter and the list's worst key per query = 12 bytes per query)
__global__ __launch_bounds__(512) void pf_cand_count(const uint32_t* __restrict__ cand_q, const uint64_t* __restrict__ cand_sv, uint64_t region_cap,
                                                     const unsigned long long* __restrict__ counters, uint32_t n_regions, uint64_t single_n, uint32_t nq,
                                                     const uint64_t* __restrict__ worst, uint32_t* __restrict__ q_count) {
    extern __shared__ __attribute__((aligned(8))) unsigned char pf_lds[];
    uint64_t* lw = reinterpret_cast<uint64_t*>(pf_lds);
    uint32_t* pf_h = reinterpret_cast<uint32_t*>(pf_lds + (size_t)nq * 8);
    __shared__ uint64_t prefix[kPfRegions + 1];
    for (uint32_t q = threadIdx.x; q < nq; q += blockDim.x) { pf_h[q] = 0; lw[q] = worst ? worst[q] : ~0ull; }
    if (worst)
        pf_tile_for_each<PfQS>(region_cap, counters, n_regions, single_n, prefix, [&](uint64_t i) { return PfQS{cand_q[i], cand_sv[i]}; },
                               [&](const PfQS& c) { if (c.sv <= lw[c.q]) atomicAdd(&pf_h[c.q], 1u); });
    else                                         // (no list is full yet -- a handle's first chunk: the scores need not be read)
        pf_tile_for_each<uint32_t>(region_cap, counters, n_regions, single_n, prefix, [&](uint64_t i) { return cand_q[i]; },
                                   [&](uint32_t q) { atomicAdd(&pf_h[q], 1u); });
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < nq; q += blockDim.x) { const uint32_t c = pf_h[q]; if (c) atomicAdd(&q_count[q], c); }
}

// one block: first[q] = candidates of the queries before q (first[nq] = all of them); cursor[q] = first[q]
__global__ __launch_bounds__(512) void pf_cand_first(const uint32_t* __restrict__ q_count, uint32_t nq, uint32_t* __restrict__ first, uint32_t* __restrict__ cursor) {
    __shared__ uint32_t part[512];
    const uint32_t per = (nq + 511) / 512;
    const uint32_t lo = min(nq, threadIdx.x * per), hi = min(nq, lo + per);
    uint32_t sum = 0;
    for (uint32_t q = lo; q < hi; ++q) sum += q_count[q];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 512; d <<= 1) {
        const uint32_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t q = lo; q < hi; ++q) { first[q] = run; cursor[q] = run; run += q_count[q]; }
    if (threadIdx.x == 511) first[nq] = part[511];
}

// out[first[q] ..] = the (score, id) words of query q's candidates (those pf_cand_count counted), in no particular order
__global__ __launch_bounds__(512) void pf_cand_scatter(const uint32_t* __restrict__ cand_q, const uint64_t* __restrict__ cand_sv, uint64_t region_cap,
                                                       const unsigned long long* __restrict__ counters, uint32_t n_regions, uint64_t single_n,
                                                       uint32_t nq, const uint64_t* __restrict__ worst, uint32_t* __restrict__ cursor, uint64_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(8))) unsigned char pf_lds[];
    uint64_t* lw = reinterpret_cast<uint64_t*>(pf_lds);
    uint32_t* pf_h = reinterpret_cast<uint32_t*>(pf_lds + (size_t)nq * 8);
    __shared__ uint64_t prefix[kPfRegions + 1];
    for (uint32_t q = threadIdx.x; q < nq; q += blockDim.x) { pf_h[q] = 0; lw[q] = worst ? worst[q] : ~0ull; }
    pf_tile_for_each<PfQS>(region_cap, counters, n_regions, single_n, prefix, [&](uint64_t i) { return PfQS{cand_q[i], cand_sv[i]}; },
                           [&](const PfQS& c) { if (c.sv <= lw[c.q]) atomicAdd(&pf_h[c.q], 1u); });
    __syncthreads();
    // the tile's share of every query's range, reserved with one atomic per query that occurs; the counter then becomes the write position
    for (uint32_t q = threadIdx.x; q < nq; q += blockDim.x) { const uint32_t c = pf_h[q]; if (c) pf_h[q] = atomicAdd(&cursor[q], c); }
    pf_tile_for_each<PfQS>(region_cap, counters, n_regions, single_n, prefix, [&](uint64_t i) { return PfQS{cand_q[i], cand_sv[i]}; },
                           [&](const PfQS& c) { if (c.sv <= lw[c.q]) out[atomicAdd(&pf_h[c.q], 1u)] = c.sv; });
}

// The same for up to kPfStageMaxQ queries, with the tile brought into query order in LDS first: a block's candidates of one query then
// leave it as ONE run written by neighbouring lanes.  (Scattered 8-byte stores from 226 blocks x 1000 open runs overflow the L2s: lines
// were evicted half written, 3.3 bytes went to memory per byte of payload.)  A tile is kPfStageTile candidates, 16 per thread, read once
// and held in registers between the counting and the placing.
constexpr uint32_t kPfStageMaxQ = 2048;
constexpr uint32_t kPfStageTile = 8192;
__host__ __device__ constexpr size_t pf_stage_lds(uint32_t nq) { return (size_t)kPfStageTile * 10 + (size_t)nq * 16; }
__global__ __launch_bounds__(512) void pf_cand_scatter_staged(const uint32_t* __restrict__ cand_q, const uint64_t* __restrict__ cand_sv, uint64_t region_cap,
                                                              const unsigned long long* __restrict__ counters, uint32_t n_regions, uint64_t single_n,
                                                              uint32_t nq, const uint64_t* __restrict__ worst, uint32_t* __restrict__ cursor, uint64_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(8))) unsigned char pf_lds[];
    uint64_t* stage_sv = reinterpret_cast<uint64_t*>(pf_lds);                       // [kPfStageTile]
    uint64_t* lw = stage_sv + kPfStageTile;                                          // [nq] worst key of the query's list
    uint32_t* fill = reinterpret_cast<uint32_t*>(lw + nq);                           // [nq] count, then the next free slot of the query's run in the stage
    uint32_t* dl = fill + nq;                                                        // [nq] (start of the run in `out`) - (start of the run in the stage)
    uint16_t* stage_q = reinterpret_cast<uint16_t*>(dl + nq);                        // [kPfStageTile]
    __shared__ uint64_t prefix[kPfRegions + 1];
    __shared__ uint32_t wave_tot[8];
    constexpr int EPT = kPfStageTile / 512;
    if (threadIdx.x == 0) {
        uint64_t acc = 0;
        for (uint32_t r = 0; r < n_regions; ++r) { prefix[r] = acc; acc += counters ? min((uint64_t)counters[8 * r], region_cap) : single_n; }
        prefix[n_regions] = acc;
    }
    for (uint32_t q = threadIdx.x; q < nq; q += blockDim.x) { fill[q] = 0; lw[q] = worst ? worst[q] : ~0ull; }
    __syncthreads();
    const uint64_t lo_tile = (uint64_t)blockIdx.x * kPfStageTile, hi_tile = min(prefix[n_regions], lo_tile + kPfStageTile);
    uint32_t r0 = 0;
    while (r0 + 1 < n_regions && prefix[r0 + 1] <= lo_tile) ++r0;                    // (uniform)
    uint32_t q[EPT];
    uint64_t sv[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const uint64_t k = lo_tile + (uint64_t)j * 512 + threadIdx.x;
        q[j] = 0xFFFFFFFFu; sv[j] = 0;
        if (k < hi_tile) {
            uint32_t r = r0;
            while (k >= prefix[r + 1]) ++r;
            const uint64_t at = (uint64_t)r * region_cap + (k - prefix[r]);
            q[j] = cand_q[at]; sv[j] = cand_sv[at];
        }
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        if (q[j] != 0xFFFFFFFFu && sv[j] > lw[q[j]]) q[j] = 0xFFFFFFFFu;             // cannot enter its list any more
        if (q[j] != 0xFFFFFFFFu) atomicAdd(&fill[q[j]], 1u);
    }
    __syncthreads();
    // runs in query order: exclusive prefix over the queries' counts (each thread a stretch of queries), their places in `out` reserved
    const uint32_t per = (nq + 511) / 512;
    const uint32_t qa = min(nq, threadIdx.x * per), qb = min(nq, qa + per);
    uint32_t sum = 0;
    for (uint32_t x = qa; x < qb; ++x) sum += fill[x];
    const uint32_t incl = pf_scan64(sum);
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) run += wave_tot[w];
    uint32_t kept = 0;
    for (uint32_t w = 0; w < 8; ++w) kept += wave_tot[w];
    __syncthreads();                                                                 // (every count has been read before any becomes a position)
    for (uint32_t x = qa; x < qb; ++x) {
        const uint32_t c = fill[x];
        dl[x] = (c ? atomicAdd(&cursor[x], c) : 0u) - run;
        fill[x] = run;
        run += c;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (q[j] != 0xFFFFFFFFu) { const uint32_t at = atomicAdd(&fill[q[j]], 1u); stage_sv[at] = sv[j]; stage_q[at] = (uint16_t)q[j]; }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kept; i += blockDim.x) out[dl[stage_q[i]] + i] = stage_sv[i];
}

// first[q] = index of the first candidate of query q in the query-sorted array (first[nq] = n): one thread per candidate boundary
__global__ void pf_bucket_first(const uint32_t* __restrict__ sorted_q, uint32_t n, uint32_t nq, uint32_t* __restrict__ first) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    // queries prev .. cur start at i (those between the query at i - 1 and the one at i have no candidate); entries of a query >= nq
    // are the unused slots of the candidate regions, sorted behind everything
    const uint32_t prev = i == 0 ? 0u : min(sorted_q[i - 1], nq) + 1;
    const uint32_t cur = i == n ? nq : min(sorted_q[i], nq);
    for (uint32_t q = prev; q <= cur; ++q) first[q] = i;
}

// One block per query: its list by ascending database index -> a row of ids and scores, and the count (database_search.cpp:166-178)
__global__ __launch_bounds__(512) void pf_finish_rows(const uint64_t* __restrict__ list, const uint32_t* __restrict__ list_cnt, uint32_t max_c,
                                                      uint32_t* __restrict__ ids, float* __restrict__ scores, uint32_t* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) uint64_t pf_keys[];
    const uint32_t q = blockIdx.x;
    const uint32_t n = list_cnt[q];
    const uint64_t* mine = list + (uint64_t)q * max_c;
    uint32_t m = 2;
    while (m < n) m <<= 1;
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
        uint64_t k = ~0ull;
        if (i < n) { const uint64_t v = mine[i]; k = (v << 32) | (uint32_t)~(uint32_t)(v >> 32); }       // id << 32 | bits(score)
        pf_keys[i] = k;
    }
    __syncthreads();
    pf_block_sort(pf_keys, m);
    for (uint32_t i = threadIdx.x; i < max_c; i += blockDim.x) {
        const bool ok = i < n;
        ids[(uint64_t)q * max_c + i] = ok ? (uint32_t)(pf_keys[i] >> 32) : 0u;
        scores[(uint64_t)q * max_c + i] = ok ? __uint_as_float((uint32_t)pf_keys[i]) : 0.0f;
    }
    if (threadIdx.x == 0) counts[q] = n;
}

// offsets handed over in device memory are checked where they are: ascending, no target beyond 0x7FFFFFF0 residues
__global__ void pf_check_offsets(const uint64_t* __restrict__ off, uint32_t nt, unsigned* __restrict__ bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nt && (off[t + 1] < off[t] || off[t + 1] - off[t] > 0x7FFFFFF0ull)) atomicOr(bad, 2u);
}

// Running lists: slots [q * max .. (q + 1) * max) of list_key / list_val, empty = kPfNoEntry.
__global__ void pf_fill(uint64_t* __restrict__ key, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = kPfNoEntry;
}

// sorted by (query, ...): first[q] = index of query q's first entry
__global__ void pf_query_first(const uint64_t* __restrict__ key, uint64_t n, uint32_t nq, uint32_t* __restrict__ first) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t q = (uint32_t)(key[i] >> 32);
    if (q < nq && (i == 0 || (uint32_t)(key[i - 1] >> 32) != q)) first[q] = (uint32_t)i;
}

// the first max entries of every query go back into the running lists (database_search.cpp:141-145)
__global__ void pf_keep(const uint64_t* __restrict__ key, const uint32_t* __restrict__ val, uint64_t n, uint32_t nq, uint32_t max_candidates,
                        const uint32_t* __restrict__ first, uint64_t* __restrict__ list_key, uint32_t* __restrict__ list_val) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t q = (uint32_t)(key[i] >> 32);
    if (q >= nq) return;
    const uint64_t r = i - first[q];
    if (r < max_candidates) { list_key[(uint64_t)q * max_candidates + r] = key[i]; list_val[(uint64_t)q * max_candidates + r] = val[i]; }
}

// finish: (query, ~score | id) -> key (query, id), value = score bits
__global__ void pf_by_id(const uint64_t* __restrict__ list_key, const uint32_t* __restrict__ list_val, uint64_t n,
                         uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = list_key[i];
    if (k == kPfNoEntry) { key[i] = kPfNoEntry; val[i] = 0; return; }
    key[i] = (k & 0xFFFFFFFF00000000ull) | list_val[i];
    val[i] = ~(uint32_t)k;
}

__global__ void pf_rows(const uint64_t* __restrict__ key, const uint32_t* __restrict__ val, uint64_t n, uint32_t nq, uint32_t max_candidates,
                        const uint32_t* __restrict__ first, uint32_t* __restrict__ ids, float* __restrict__ scores, uint32_t* __restrict__ counts) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t q = (uint32_t)(key[i] >> 32);
    if (q >= nq) return;
    const uint64_t r = i - first[q];
    ids[(uint64_t)q * max_candidates + r] = (uint32_t)key[i];
    scores[(uint64_t)q * max_candidates + r] = __uint_as_float(val[i]);
    if (i + 1 == n || (uint32_t)(key[i + 1] >> 32) != q) counts[q] = (uint32_t)r + 1;      // the query's last entry knows the count
}

}  // namespace s4g

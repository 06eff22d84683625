// kmer_prefilter.hip.h -- device kernels of the k-mer prefilter (host side: s4g_prefilter.hip).
//
// What the reference does on its thread pool (/root/reference/sift4g/src/database_search.cpp:185-253, hash.cpp:21-90), per
// database chunk, as data-parallel passes over HBM-resident arrays:
//
//   pf_query_kmers     one thread per query residue: the k-mer that starts there (hash.cpp:21-44) -> sort key, bucket count
//   (radix sort, scan)  -> the CSR index of hash.cpp:56-85: starts[kmer], hits in (query, position) order inside a bucket
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
// HBM-bound integer work: no MFMA anywhere; what matters is coalesced target reads (a wave walks one target), the random
// 8-byte bucket lookups (32^5 slots = 134 MB: MALL-resident, not L2) and the radix sorts (rocPRIM).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s4g {

constexpr uint32_t kKmerBits = 5;               // hash.cpp:17
constexpr uint32_t kPfSmallMax = 32;            // longest segment a single thread handles
constexpr uint32_t kPfLdsTails = 4096;          // tails a wave keeps in LDS (16 KB)
constexpr uint64_t kPfNoEntry = ~0ull;          // empty slot of a candidate list: sorts behind everything

// the k-mer starting at codes[j] (hash.cpp:33-43: ((kmer << 5) | code) & mask, most significant residue first)
// (codes are cut to 5 bits: a code of 32 or more is reported through the `bad` flag, and must not index past the tables first)
__device__ __forceinline__ uint32_t pf_kmer(const uint8_t* __restrict__ c, uint32_t k) {
    uint32_t v = 0;
    for (uint32_t i = 0; i < k; ++i) v = (v << kKmerBits) | (c[i] & 31u);
    return v;
}

// One thread per query residue g (flat over all queries).  key[g] = k-mer starting there, or the invalid key 1 << 5k when fewer
// than k residues are left in that query; val[g] = g; pos_q[g] / pos_j[g] = query index / position of g; count[kmer] += 1;
// bit kmer of `present` set.
__global__ void pf_query_kmers(const uint8_t* __restrict__ codes, const uint64_t* __restrict__ off, uint32_t nq, uint64_t total,
                               uint32_t k, uint32_t* __restrict__ key, uint32_t* __restrict__ val, uint32_t* __restrict__ pos_q,
                               uint32_t* __restrict__ pos_j, uint32_t* __restrict__ count, uint32_t* __restrict__ present,
                               unsigned* __restrict__ bad) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    uint32_t lo = 0, hi = nq;                   // last q with off[q] <= g  (queries of length 0 share an offset: take the last)
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (off[mid] <= g) lo = mid; else hi = mid; }
    const uint32_t j = (uint32_t)(g - off[lo]);
    const uint32_t len = (uint32_t)(off[lo + 1] - off[lo]);
    pos_q[g] = lo; pos_j[g] = j; val[g] = (uint32_t)g;
    if (codes[g] >= 32) atomicOr(bad, 1u);
    if (j + k <= len) {
        const uint32_t v = pf_kmer(codes + g, k);
        key[g] = v;
        atomicAdd(&count[v], 1u);
        atomicOr(&present[v >> 5], 1u << (v & 31));
    } else {
        key[g] = 1u << (k * kKmerBits);
    }
}

// hits of the CSR index from the sorted (k-mer, g) pairs: hit_q / hit_p[i] = query / position of the i-th smallest
__global__ void pf_hash_hits(const uint32_t* __restrict__ sorted_val, const uint32_t* __restrict__ pos_q,
                             const uint32_t* __restrict__ pos_j, uint64_t n_hits, uint32_t* __restrict__ hit_q, uint32_t* __restrict__ hit_p) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hits) return;
    const uint32_t g = sorted_val[i];
    hit_q[i] = pos_q[g]; hit_p[i] = pos_j[g];
}

struct PfChunk {
    const uint8_t* codes;       // residues of the chunk, flat
    const uint64_t* off;        // [nt + 1]
    const uint32_t* starts;     // [32^k + 1] CSR index of the queries' k-mers
    const uint32_t* present;    // one bit per k-mer: does any query hold it?  (4 MB at k = 5: answers nine lookups in ten from L2
                                // instead of a random 64-byte fetch from the 134 MB of `starts`)
    const uint32_t* hit_q;      // bucket contents
    const uint32_t* hit_p;
    uint32_t k, nt;
};

// number of hits the k-mer at position j of a target contributes, and where its bucket starts
__device__ __forceinline__ uint32_t pf_bucket(const PfChunk& a, const uint8_t* __restrict__ c, uint32_t j, uint32_t n_kmers, uint32_t* first) {
    if (j >= n_kmers) return 0;
    const uint32_t v = pf_kmer(c + j, a.k);
    if (j != 0) {                                                       // database_search.cpp:212: equal to the k-mer before it
        const uint32_t prev = ((uint32_t)(c[j - 1] & 31u) << ((a.k - 1) * kKmerBits)) | (v >> kKmerBits);
        if (prev == v) return 0;
    }
    if (!((a.present[v >> 5] >> (v & 31)) & 1u)) return 0;
    const uint32_t s = a.starts[v];
    *first = s;
    return a.starts[v + 1] - s;
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
        const uint64_t o = a.off[t];
        const uint32_t len = (uint32_t)(a.off[t + 1] - o);
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
        const uint64_t o = a.off[t];
        const uint32_t len = (uint32_t)(a.off[t + 1] - o);
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
};

__device__ __forceinline__ void pf_candidate(const PfLisArgs& a, uint32_t s, uint32_t lis) {
    const uint64_t k = a.seg_key[s];
    const uint32_t t = a.t0 + (uint32_t)(k >> a.q_bits);
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

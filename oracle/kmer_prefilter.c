/* kmer_prefilter.c -- CPU restatement of sift4g's k-mer prefilter.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and tests/cpu_shim/ may call this; the product path (sift4g_amd/csrc/s4g_prefilter.hip)
 * never does.  Unlike the scorer (oracle/sw_oracle.c, "parity unpinned"), the whole algorithm of this stage is present in
 * /root/reference, so every function below follows a cited piece of the reference's text:
 *
 *   k-mer codes of a chain          /root/reference/sift4g/src/hash.cpp:21-44   (5-bit codes, rolling k-mer, kDelMasks :18)
 *   CSR index of the queries' k-mers /root/reference/sift4g/src/hash.cpp:56-85   (starts_/hits_, hits of one k-mer in
 *                                                                                 (query, position) order)
 *   per-target hit lists            /root/reference/sift4g/src/database_search.cpp:208-221 (a k-mer equal to the one before
 *                                                                                 it is skipped; hits[query] += positions)
 *   longest increasing subsequence  /root/reference/sift4g/src/database_search.cpp:255-280 (strict; lower bound by bisection)
 *   score = LIS / (float) length    /root/reference/sift4g/src/database_search.cpp:228-229
 *   keep the best max_candidates    /root/reference/sift4g/src/database_search.cpp:231-251,129-150 and, finally, ids in
 *                                   ascending order (:166-178)
 *
 * ONE deliberate difference, stated everywhere it matters: the reference's choice among candidates whose score EQUALS the
 * score at the cut depends on its thread count and on std::sort's handling of equal keys (:231-234 reject a tie once the list
 * is full, :243,140 sort by score only).  Here -- and on the device -- ties go to the lower database index, which is one of
 * the results the reference can produce; every candidate strictly above the cut is the same in all of them.
 *
 * What is still an assumption: the residue codes chainGetCodes returns (letter - 'A', 0..25: hash.cpp:16-17 pins the range).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

enum { kBits = 5 };   /* hash.cpp:17 kProtBitLength */

/* hash.cpp:21-44 createKmerVector.  Returns the number of k-mers (0 when the chain is shorter than k). */
static uint32_t kmer_vector(const uint8_t* codes, uint32_t len, uint32_t k, uint32_t* dst) {
    if (len < k) return 0;
    const uint32_t mask = (k * kBits >= 32) ? 0xFFFFFFFFu : ((1u << (k * kBits)) - 1u);   /* kDelMasks[k], hash.cpp:18 */
    uint32_t kmer = 0, n = 0;
    for (uint32_t i = 0; i < k; ++i) kmer = (kmer << kBits) | codes[i];
    dst[n++] = kmer;
    for (uint32_t i = k; i < len; ++i) {
        kmer = ((kmer << kBits) | codes[i]) & mask;
        dst[n++] = kmer;
    }
    return n;
}

typedef struct { uint32_t id, position; } Hit;   /* hash.hpp:19-30 */

typedef struct {
    uint32_t k, nq, max_candidates;
    uint64_t n_kmers;       /* 32^k slots: every 5-bit code is in range (the reference sizes for codes <= 25, hash.cpp:19) */
    uint32_t* ends;         /* hash.cpp:56-68 starts_, kept as ENDS: bucket of k-mer v = hits[v ? ends[v - 1] : 0 .. ends[v]) -- the fill
                               pass of hash.cpp:75-84 advances a cursor per bucket from its start to its end, so the cursors ARE the ends
                               and no second 32^k-entry array is needed (a test creates dozens of these indexes) */
    Hit* hits;              /* hash.cpp:70-85 */
    uint32_t max_qlen;
    /* candidates per query, every one ever seen (the cut is applied at the end: the top of a union under a total order is the
       top of the union of the tops, so chunk-wise truncation as in database_search.cpp:129-150 gives the same list) */
    float** score; uint32_t** id; uint64_t* n; uint64_t* cap;
} Prefilter;

API void* s4g_oracle_prefilter_create(const uint8_t* qcodes, const uint64_t* qoff, uint32_t nq, uint32_t k, uint32_t max_candidates) {
    if (k < 3 || k > 5) return NULL;                                  /* main.cpp:169 */
    Prefilter* p = (Prefilter*)calloc(1, sizeof *p);
    p->k = k; p->nq = nq; p->max_candidates = max_candidates;
    p->n_kmers = 1ull << (k * kBits);
    p->ends = (uint32_t*)calloc(p->n_kmers + 1, sizeof(uint32_t));
    uint32_t max_len = 0;
    for (uint32_t q = 0; q < nq; ++q) { uint32_t l = (uint32_t)(qoff[q + 1] - qoff[q]); if (l > max_len) max_len = l; }
    p->max_qlen = max_len;
    uint32_t* kv = (uint32_t*)malloc(sizeof(uint32_t) * (max_len + 1));
    for (uint32_t q = 0; q < nq; ++q) {                               /* hash.cpp:60-68: counts */
        uint32_t n = kmer_vector(qcodes + qoff[q], (uint32_t)(qoff[q + 1] - qoff[q]), k, kv);
        for (uint32_t j = 0; j < n; ++j) ++p->ends[kv[j]];
    }
    uint32_t total = 0;                                               /* counts -> starts (exclusive prefix sum) */
    for (uint64_t v = 0; v < p->n_kmers; ++v) { const uint32_t c = p->ends[v]; p->ends[v] = total; total += c; }
    p->hits = (Hit*)malloc(sizeof(Hit) * ((size_t)total + 1));
    for (uint32_t q = 0; q < nq; ++q) {                               /* hash.cpp:75-84: queries in order, positions in order */
        uint32_t n = kmer_vector(qcodes + qoff[q], (uint32_t)(qoff[q + 1] - qoff[q]), k, kv);
        for (uint32_t j = 0; j < n; ++j) { Hit h = {q, j}; p->hits[p->ends[kv[j]]++] = h; }       /* starts become ends */
    }
    free(kv);
    p->score = (float**)calloc(nq ? nq : 1, sizeof(float*)); p->id = (uint32_t**)calloc(nq ? nq : 1, sizeof(uint32_t*));
    p->n = (uint64_t*)calloc(nq ? nq : 1, sizeof(uint64_t)); p->cap = (uint64_t*)calloc(nq ? nq : 1, sizeof(uint64_t));
    return p;
}

/* database_search.cpp:255-280 longestIncreasingSubsequence: help[l] = smallest tail of an increasing run of length l. */
static int32_t lis_length(const int32_t* src, uint32_t n, int32_t* help) {
    int32_t score = 0;
    help[0] = INT32_MIN;
    for (uint32_t i = 1; i <= n; ++i) help[i] = INT32_MAX;
    for (uint32_t i = 0; i < n; ++i) {
        int32_t l = 0, u = (int32_t)n;
        while (u > l) {
            int32_t mid = (l + u) / 2;            /* floor((l + u) / 2.0f), :268, for non-negative l + u */
            if (help[mid] < src[i]) l = mid + 1; else u = mid;
        }
        help[l] = src[i];
        if (l > score) score = l;
    }
    return score;
}

typedef struct { int32_t* v; uint32_t n, cap; } HitList;

/* The hit lists of ONE target against every query (database_search.cpp:208-221), then LIS per query that has any (:223-229).
   visit(query, lis) is called in ascending query order. */
static void target_hits(Prefilter* p, const uint8_t* codes, uint32_t len, uint32_t* kv, HitList* lists, uint32_t* touched,
                        int32_t** help, uint32_t* help_cap, void (*visit)(void*, uint32_t, int32_t), void* ctx) {
    uint32_t n = kmer_vector(codes, len, p->k, kv), n_touched = 0;
    for (uint32_t j = 0; j < n; ++j) {
        if (j != 0 && kv[j] == kv[j - 1]) continue;                                   /* :212-214 */
        for (uint32_t h = kv[j] ? p->ends[kv[j] - 1] : 0; h < p->ends[kv[j]]; ++h) {   /* :216-220 */
            HitList* L = &lists[p->hits[h].id];
            if (L->n == 0) touched[n_touched++] = p->hits[h].id;
            if (L->n == L->cap) { L->cap = L->cap ? 2 * L->cap : 16; L->v = (int32_t*)realloc(L->v, sizeof(int32_t) * L->cap); }
            L->v[L->n++] = (int32_t)p->hits[h].position;
        }
    }
    /* ascending query order, as the loop at :223 visits them */
    for (uint32_t a = 1; a < n_touched; ++a) { uint32_t x = touched[a], b = a; while (b && touched[b - 1] > x) { touched[b] = touched[b - 1]; --b; } touched[b] = x; }
    for (uint32_t a = 0; a < n_touched; ++a) {
        HitList* L = &lists[touched[a]];
        if (L->n + 1 > *help_cap) { *help_cap = 2 * (L->n + 1); *help = (int32_t*)realloc(*help, sizeof(int32_t) * *help_cap); }
        visit(ctx, touched[a], lis_length(L->v, L->n, *help));
        L->n = 0;
    }
}

typedef struct { Prefilter* p; uint32_t target_id; uint32_t target_len; } AddCtx;
static void add_visit(void* c, uint32_t q, int32_t lis) {
    AddCtx* a = (AddCtx*)c; Prefilter* p = a->p;
    if (p->n[q] == p->cap[q]) {
        p->cap[q] = p->cap[q] ? 2 * p->cap[q] : 64;
        p->score[q] = (float*)realloc(p->score[q], sizeof(float) * p->cap[q]);
        p->id[q] = (uint32_t*)realloc(p->id[q], sizeof(uint32_t) * p->cap[q]);
    }
    p->score[q][p->n[q]] = lis / (float)a->target_len;                              /* :228-229 */
    p->id[q][p->n[q]++] = a->target_id;
}

/* One database chunk: targets first_id .. first_id + nt - 1 (database_search.cpp:185-253 over the whole chunk). */
API void s4g_oracle_prefilter_add(void* handle, const uint8_t* tcodes, const uint64_t* toff, uint32_t nt, uint32_t first_id) {
    Prefilter* p = (Prefilter*)handle;
    uint32_t max_len = 0;
    for (uint32_t t = 0; t < nt; ++t) { uint32_t l = (uint32_t)(toff[t + 1] - toff[t]); if (l > max_len) max_len = l; }
    uint32_t* kv = (uint32_t*)malloc(sizeof(uint32_t) * (max_len + 1));
    HitList* lists = (HitList*)calloc(p->nq ? p->nq : 1, sizeof(HitList));
    uint32_t* touched = (uint32_t*)malloc(sizeof(uint32_t) * (p->nq ? p->nq : 1));
    int32_t* help = NULL; uint32_t help_cap = 0;
    for (uint32_t t = 0; t < nt; ++t) {
        AddCtx c = {p, first_id + t, (uint32_t)(toff[t + 1] - toff[t])};
        target_hits(p, tcodes + toff[t], c.target_len, kv, lists, touched, &help, &help_cap, add_visit, &c);
    }
    for (uint32_t q = 0; q < p->nq; ++q) free(lists[q].v);
    free(lists); free(touched); free(help); free(kv);
}

typedef struct { float score; uint32_t id; } Cand;
static int by_score_then_id(const void* a, const void* b) {
    const Cand* x = (const Cand*)a; const Cand* y = (const Cand*)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;     /* Candidate::operator<, database_search.cpp:26-28 */
    return x->id < y->id ? -1 : (x->id > y->id);                       /* the tie rule of this restatement (header) */
}
static int by_id(const void* a, const void* b) {
    const Cand* x = (const Cand*)a; const Cand* y = (const Cand*)b;
    return x->id < y->id ? -1 : (x->id > y->id);
}

/* ids / scores: nq rows of max_candidates; counts[q] entries of a row are valid, ids ascending (database_search.cpp:166-178).
   scores may be NULL.  cut[q] (may be NULL) = the lowest score kept, or 0 when nothing was cut off for that query. */
API void s4g_oracle_prefilter_finish(void* handle, uint32_t* ids, uint32_t* counts, float* scores, float* cut) {
    Prefilter* p = (Prefilter*)handle;
    for (uint32_t q = 0; q < p->nq; ++q) {
        uint64_t n = p->n[q];
        Cand* c = (Cand*)malloc(sizeof(Cand) * (n ? n : 1));
        for (uint64_t i = 0; i < n; ++i) { c[i].score = p->score[q][i]; c[i].id = p->id[q][i]; }
        qsort(c, n, sizeof(Cand), by_score_then_id);
        uint64_t keep = n < p->max_candidates ? n : p->max_candidates;
        if (cut) cut[q] = (n > keep && keep) ? c[keep - 1].score : 0.0f;
        qsort(c, keep, sizeof(Cand), by_id);
        counts[q] = (uint32_t)keep;
        for (uint64_t i = 0; i < keep; ++i) {
            ids[(uint64_t)q * p->max_candidates + i] = c[i].id;
            if (scores) scores[(uint64_t)q * p->max_candidates + i] = c[i].score;
        }
        free(c);
    }
}

API void s4g_oracle_prefilter_destroy(void* handle) {
    Prefilter* p = (Prefilter*)handle;
    if (!p) return;
    for (uint32_t q = 0; q < p->nq; ++q) { free(p->score[q]); free(p->id[q]); }
    free(p->score); free(p->id); free(p->n); free(p->cap); free(p->ends); free(p->hits); free(p);
}

/* Dense form for the kernel-level tests: lis[q * nt + t] = LIS of the hit list of (query q, target t), 0 where there is none. */
typedef struct { int32_t* row; uint32_t nt, t; } DenseCtx;
static void dense_visit(void* c, uint32_t q, int32_t lis) { DenseCtx* d = (DenseCtx*)c; d->row[(uint64_t)q * d->nt + d->t] = lis; }

API void s4g_oracle_prefilter_lis(void* handle, const uint8_t* tcodes, const uint64_t* toff, uint32_t nt, int32_t* lis) {
    Prefilter* p = (Prefilter*)handle;
    uint32_t max_len = 0;
    for (uint32_t t = 0; t < nt; ++t) { uint32_t l = (uint32_t)(toff[t + 1] - toff[t]); if (l > max_len) max_len = l; }
    uint32_t* kv = (uint32_t*)malloc(sizeof(uint32_t) * (max_len + 1));
    HitList* lists = (HitList*)calloc(p->nq ? p->nq : 1, sizeof(HitList));
    uint32_t* touched = (uint32_t*)malloc(sizeof(uint32_t) * (p->nq ? p->nq : 1));
    int32_t* help = NULL; uint32_t help_cap = 0;
    memset(lis, 0, sizeof(int32_t) * (uint64_t)p->nq * nt);
    for (uint32_t t = 0; t < nt; ++t) {
        DenseCtx c = {lis, nt, t};
        target_hits(p, tcodes + toff[t], (uint32_t)(toff[t + 1] - toff[t]), kv, lists, touched, &help, &help_cap, dense_visit, &c);
    }
    for (uint32_t q = 0; q < p->nq; ++q) free(lists[q].v);
    free(lists); free(touched); free(help); free(kv);
}

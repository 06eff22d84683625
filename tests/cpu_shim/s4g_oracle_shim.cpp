// s4g_oracle_shim.cpp -- TEST INFRASTRUCTURE ONLY (lives under tests/, never shipped, never on the product path).
//
// A stand-in for libs4g_hip.so that answers the few device-ABI calls libswsharp makes with the CPU oracle
// (oracle/sw_oracle.c).  tests/test_cpu_pipeline.py puts it first on LD_LIBRARY_PATH and runs the reference's own
// unmodified CLI (build/dropin/sift4g) on the reference's fixture WITHOUT a GPU: the .SIFTprediction files must equal the
// ones the same CLI wrote on an MI355X through the HIP kernels (tests/golden/dropin_predictions/).  That is the
// "GPU path == its own CPU path" end-to-end equality SURVEY.md 8c names as the strongest statement available while
// swsharp itself is absent ("parity unpinned").  The shim exports only the symbols libswsharp imports.
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/s4g_hip.h"

extern "C" {
int s4g_oracle_sw_score(const uint8_t* q, int lq, const uint8_t* t, int lt, const int32_t* M, int open, int ext);
int s4g_oracle_sw_align(const uint8_t* q, int lq, const uint8_t* t, int lt, const int32_t* M, int open, int ext,
                        int* q_start, int* q_end, int* t_start, int* t_end, uint8_t* path_out, int* path_len);
}

extern "C" {
void* s4g_oracle_prefilter_create(const uint8_t* qcodes, const uint64_t* qoff, uint32_t nq, uint32_t k, uint32_t max_candidates);
void s4g_oracle_prefilter_add(void* h, const uint8_t* tcodes, const uint64_t* toff, uint32_t nt, uint32_t first_id);
void s4g_oracle_prefilter_finish(void* h, uint32_t* ids, uint32_t* counts, float* scores, float* cut);
void s4g_oracle_prefilter_destroy(void* h);
}

struct s4g_scorer { int32_t table[26 * 26]; int open, ext; };
struct Seqs { std::vector<uint8_t> codes; std::vector<uint64_t> off; uint32_t n = 0; uint32_t len(uint32_t i) const { return (uint32_t)(off[i + 1] - off[i]); } };
struct s4g_db : Seqs {};
struct s4g_queries : Seqs {};

#define API extern "C" __attribute__((visibility("default")))
static thread_local std::string g_err;
static int fail(int code, const char* msg) { g_err = msg; return code; }

API const char* s4g_last_error(void) { return g_err.c_str(); }
API int s4g_warmup(int) { return S4G_OK; }   // nothing to warm
API int s4g_is_shim(void) { return 1; }     // the mark sift4g_amd/_lib.py refuses: this file can never be loaded as the product library
API int s4g_device_count(int* count) { if (count) *count = 1; return S4G_OK; }      // one pretend card
API int s4g_device_check(const int* cards, int n) {
    if (n == 0) return 1;
    if (n < 0 || !cards) return 0;
    for (int i = 0; i < n; ++i) if (cards[i] != 0) return 0;
    return 1;
}
API int s4g_scorer_create(s4g_scorer** out, const int32_t* table, int gap_open, int gap_extend) {
    if (!out || !table || gap_extend < 0 || gap_open < gap_extend) return fail(S4G_ERR_INVALID, "bad scorer");
    s4g_scorer* s = new s4g_scorer();
    memcpy(s->table, table, sizeof s->table); s->open = gap_open; s->ext = gap_extend;
    *out = s;
    return S4G_OK;
}
API int s4g_scorer_destroy(s4g_scorer* s) { delete s; return S4G_OK; }
template <typename T>
static int make(T** out, const uint8_t* codes, const uint64_t* offsets, uint32_t n) {
    if (!out || !offsets) return fail(S4G_ERR_INVALID, "NULL argument");
    T* s = new T();
    s->n = n; s->off.assign(offsets, offsets + n + 1);
    s->codes.assign(codes ? codes : (const uint8_t*)"", codes ? codes + offsets[n] : (const uint8_t*)"");
    for (uint8_t c : s->codes) if (c >= 26) { delete s; return fail(S4G_ERR_INVALID, "residue code out of range"); }
    *out = s;
    return S4G_OK;
}
API int s4g_db_create(s4g_db** out, int, const uint8_t* codes, const uint64_t* offsets, uint32_t n) { return make(out, codes, offsets, n); }
API int s4g_db_create_scattered(s4g_db** out, int, const uint8_t* const* seqs, const uint32_t* lens, uint32_t n) {
    std::vector<uint8_t> codes;
    std::vector<uint64_t> off(1, 0);
    for (uint32_t i = 0; i < n; ++i) { codes.insert(codes.end(), seqs[i], seqs[i] + lens[i]); off.push_back(codes.size()); }
    return make(out, codes.data(), off.data(), n);
}
API int s4g_db_create_cached(s4g_db** out, int dev, const uint8_t* const* seqs, const uint32_t* lens, uint64_t* tags, uint64_t* addrs, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) { tags[i] = 64; addrs[i] = 1; }       // (nothing is cached here: every call copies the chains)
    return s4g_db_create_scattered(out, dev, seqs, lens, n);
}
API int s4g_db_destroy(s4g_db* db) { delete db; return S4G_OK; }
API int s4g_queries_create(s4g_queries** out, int, const uint8_t* codes, const uint64_t* offsets, uint32_t n) { return make(out, codes, offsets, n); }
API int s4g_queries_destroy(s4g_queries* q) { delete q; return S4G_OK; }

static int score(const s4g_db* db, const s4g_queries* q, const s4g_scorer* sc, uint32_t qi, uint32_t ti) {
    return s4g_oracle_sw_score(q->codes.data() + q->off[qi], (int)q->len(qi), db->codes.data() + db->off[ti], (int)db->len(ti),
                               sc->table, sc->open, sc->ext);
}
API int s4g_scan_host(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* sc, int32_t* h) {
    for (uint32_t i = 0; i < q_count; ++i)
        for (uint32_t t = 0; t < db->n; ++t) h[(size_t)i * db->n + t] = score(db, q, sc, q_begin + i, t);
    return S4G_OK;
}
API int s4g_score_pairs_host(const s4g_db* db, const s4g_queries* q, const s4g_scorer* sc, uint64_t n, const uint32_t* pq,
                             const uint32_t* pt, int32_t* h, uint32_t) {
    for (uint64_t i = 0; i < n; ++i) h[i] = score(db, q, sc, pq[i], pt[i]);
    return S4G_OK;
}
API int s4g_scan_topk_host(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* sc,
                           uint32_t k, const int32_t* min_scores, uint32_t* ids, int32_t* top, uint32_t* counts, uint32_t) {
    std::vector<std::pair<int32_t, uint32_t>> v;
    for (uint32_t i = 0; i < q_count; ++i) {
        v.clear();
        const int32_t cut = min_scores ? min_scores[i] : 1;
        for (uint32_t t = 0; t < db->n; ++t) { const int s = score(db, q, sc, q_begin + i, t); if (s >= cut) v.push_back({s, t}); }
        std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
        const uint32_t m = (uint32_t)std::min<size_t>(k, v.size());
        for (uint32_t j = 0; j < k; ++j) { ids[(size_t)i * k + j] = j < m ? v[j].second : 0xFFFFFFFFu; top[(size_t)i * k + j] = j < m ? v[j].first : 0; }
        counts[i] = m;
    }
    return S4G_OK;
}
API int s4g_traceback_host(const s4g_db* db, const s4g_queries* q, const s4g_scorer* sc, uint32_t n, const uint32_t* pq,
                           const uint32_t* pt, const int32_t* ps, int32_t* out, const uint64_t* path_off, uint8_t* paths) {
    static const uint8_t kMove[3] = {1, 2, 3};      // oracle 0 DIAG / 1 LEFT / 2 UP -> swsharp MOVE_DIAG 1 / MOVE_LEFT 2 / MOVE_UP 3
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t lq = q->len(pq[i]), lt = db->len(pt[i]);
        std::vector<uint8_t> path((size_t)lq + lt + 1);
        int v[5];
        const int s = s4g_oracle_sw_align(q->codes.data() + q->off[pq[i]], (int)lq, db->codes.data() + db->off[pt[i]], (int)lt,
                                          sc->table, sc->open, sc->ext, &v[0], &v[1], &v[2], &v[3], path.data(), &v[4]);
        if (s != ps[i]) return fail(S4G_ERR_INVALID, "score passed in differs from the oracle's");
        for (int k = 0; k < 5; ++k) out[(size_t)i * 5 + k] = v[k];
        for (int k = 0; k < v[4]; ++k) paths[path_off[i] + k] = kMove[path[k]];
    }
    return S4G_OK;
}

// the traceback in two halves: the stand-in traces at submit, finish hands the results over (same results, no overlap)
struct s4g_trace_job { uint32_t n; std::vector<int32_t> out; std::vector<uint8_t> paths; std::vector<uint64_t> off; };
API int s4g_traceback_submit(const s4g_db* db, const s4g_queries* q, const s4g_scorer* sc, uint32_t n, const uint32_t* pq, const uint32_t* pt,
                             const int32_t* ps, const uint64_t* path_off, s4g_trace_job** job) {
    s4g_trace_job* j = new s4g_trace_job();
    j->n = n; j->out.assign((size_t)n * 5 + 1, 0); j->off.assign(path_off, path_off + n + 1); j->paths.assign(n ? path_off[n] + 1 : 1, 0);
    const int rc = n ? s4g_traceback_host(db, q, sc, n, pq, pt, ps, j->out.data(), path_off, j->paths.data()) : S4G_OK;
    if (rc) { delete j; return rc; }
    *job = j;
    return S4G_OK;
}
API int s4g_traceback_finish(s4g_trace_job* j, int32_t* out, uint8_t* paths) {
    if (!j) return fail(S4G_ERR_INVALID, "job is NULL");
    for (uint32_t i = 0; i < j->n; ++i) {
        for (int k = 0; k < 5; ++k) out[(size_t)i * 5 + k] = j->out[(size_t)i * 5 + k];
        memcpy(paths + j->off[i], j->paths.data() + j->off[i], (size_t)j->out[(size_t)i * 5 + 4]);
    }
    delete j;
    return S4G_OK;
}

// the k-mer prefilter (device_database_search.cpp): answered by oracle/kmer_prefilter.c
struct s4g_prefilter { void* h; };
API int s4g_prefilter_create(s4g_prefilter** out, int, const uint8_t* codes, const uint64_t* offsets, uint32_t n, uint32_t k, uint32_t max_candidates) {
    void* h = s4g_oracle_prefilter_create(codes, offsets, n, k, max_candidates);
    if (!h) return fail(S4G_ERR_INVALID, "kmer_length possible values = 3,4,5");
    *out = new s4g_prefilter{h};
    return S4G_OK;
}
API int s4g_prefilter_add(s4g_prefilter* p, const uint8_t* codes, const uint64_t* offsets, uint32_t n, uint32_t first_id) {
    s4g_oracle_prefilter_add(p->h, codes, offsets, n, first_id);
    return S4G_OK;
}
API int s4g_prefilter_finish(s4g_prefilter* p, uint32_t* ids, uint32_t* counts, float* scores) {
    s4g_oracle_prefilter_finish(p->h, ids, counts, scores, nullptr);
    return S4G_OK;
}
API int s4g_prefilter_destroy(s4g_prefilter* p) { if (p) { s4g_oracle_prefilter_destroy(p->h); delete p; } return S4G_OK; }

// align.cpp -- ChainDatabase, alignDatabase / shotgunDatabase, host traceback, DbAlignment / Alignment objects.
//
// alignDatabase as sift4g calls it (/root/reference/sift4g/src/database_alignment.cpp:83-86):
//   1. score the query against every target       -> libs4g_hip (HIP, MI355X); NO host scoring path
//   2. valueFunction(values, scores, ...)          -> caller's callback (e-values, :129-134)
//   3. keep value <= threshold, sort, top maxAlignments
//   4. traceback of the survivors                  -> libs4g_hip (s4g_traceback_host: device fill + walk kernels);
//                                                     the host traceback below serves alignPair and dbAlignmentsFromScores
// ASSUMPTIONS (unpinned, swsharp source absent): order = value ascending, then score descending, then target index
// ascending; traceback ties = first maximum in row-major order as end cell, then DIAG before UP before LEFT.
#include "impl.hpp"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>

// S4G_PROFILE=1: cumulative wall time per phase of the host API, printed at exit (diagnostic only)
namespace {
struct PhaseClock {
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool on = getenv("S4G_PROFILE") != NULL;
    ~PhaseClock() {
        if (!on) return;
        static const char* names[8] = {"chainDatabaseCreate", "queries to device", "scan + download", "values/selection", "device traceback",
                                       "build DbAlignments", "chainDatabaseDelete", ""};
        for (int i = 0; i < 7; ++i) fprintf(stderr, "[s4g profile] %-22s %8.3f s  (%ld calls)\n", names[i], t[i], n[i]);
    }
} g_clock;
struct Phase {
    int id; std::chrono::steady_clock::time_point t0;
    explicit Phase(int i) : id(i), t0(std::chrono::steady_clock::now()) {}
    ~Phase() { if (g_clock.on) { g_clock.t[id] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); g_clock.n[id]++; } }
};
}  // namespace

// ------------------------------------------------------------------------------------------------ traceback
namespace swimpl {

namespace {
constexpr int kNegInf = -(1 << 29);
inline int imax(int a, int b) { return a > b ? a : b; }
enum : unsigned char { H_STOP = 0, H_DIAG = 1, H_UP = 2, H_LEFT = 3, E_OPENS = 4, F_OPENS = 8 };
}  // namespace

void sw_traceback(const char* q, int lq, const char* t, int lt, const Scorer* sc, PairResult* out) {
    out->score = 0; out->q_start = out->q_end = out->t_start = out->t_end = -1; out->path.clear();
    if (lq <= 0 || lt <= 0) return;
    const int open = sc->gap_open, ext = sc->gap_extend;
    const int32_t* M = sc->table;
    // pass 1: score and end cell (first maximum in row-major order), O(lt) memory
    std::vector<int> Hrow(lt + 1, 0), Frow(lt + 1, kNegInf);
    int best = 0, bi = 0, bj = 0;
    for (int i = 1; i <= lq; ++i) {
        const int32_t* Mq = M + 26 * q[i - 1];
        int diag = 0, hleft = 0, e = kNegInf;
        for (int j = 1; j <= lt; ++j) {
            e = imax(e - ext, hleft - open);
            const int f = imax(Frow[j] - ext, Hrow[j] - open);
            int h = imax(imax(diag + Mq[(int)t[j - 1]], 0), imax(e, f));
            diag = Hrow[j]; Hrow[j] = h; Frow[j] = f; hleft = h;
            if (h > best) { best = h; bi = i; bj = j; }
        }
    }
    out->score = best;
    if (best == 0) return;
    // pass 2: direction bytes on the prefix box [1..bi] x [1..bj] (cell values there equal the full DP's)
    const char* ties = getenv("S4G_TRACEBACK_TIES");                 // "up" (default) | "left": as the device kernels (s4g_internal.hpp)
    const bool left_first = ties && strcmp(ties, "left") == 0;
    const size_t W = (size_t)bj + 1;
    std::vector<unsigned char> dir(((size_t)bi + 1) * W, 0);
    std::fill(Hrow.begin(), Hrow.end(), 0);
    std::fill(Frow.begin(), Frow.end(), kNegInf);
    for (int i = 1; i <= bi; ++i) {
        const int32_t* Mq = M + 26 * q[i - 1];
        int diag = 0, hleft = 0, e = kNegInf;
        unsigned char* drow = dir.data() + (size_t)i * W;
        for (int j = 1; j <= bj; ++j) {
            const int e_open = hleft - open, f_open = Hrow[j] - open;
            e = imax(e - ext, e_open);
            const int f = imax(Frow[j] - ext, f_open);
            const int d = diag + Mq[(int)t[j - 1]];
            const int h = imax(imax(d, 0), imax(e, f));
            unsigned char b = h == 0 ? H_STOP : (h == d ? H_DIAG : (left_first ? (h == e ? H_LEFT : H_UP) : (h == f ? H_UP : H_LEFT)));
            if (e == e_open) b |= E_OPENS;
            if (f == f_open) b |= F_OPENS;
            drow[j] = b;
            diag = Hrow[j]; Hrow[j] = h; Frow[j] = f; hleft = h;
        }
    }
    int i = bi, j = bj, state = 0;   // 0 = H, 1 = E (LEFT run), 2 = F (UP run)
    std::vector<char>& p = out->path;
    while (i > 0 && j > 0) {
        const unsigned char b = dir[(size_t)i * W + j];
        if (state == 0) {
            const int s = b & 3;
            if (s == H_STOP) break;
            if (s == H_DIAG) { p.push_back(MOVE_DIAG); --i; --j; }
            else state = (s == H_UP) ? 2 : 1;
        } else if (state == 1) {
            p.push_back(MOVE_LEFT);
            if (b & E_OPENS) state = 0;
            --j;
        } else {
            p.push_back(MOVE_UP);
            if (b & F_OPENS) state = 0;
            --i;
        }
    }
    std::reverse(p.begin(), p.end());
    out->q_start = i; out->t_start = j; out->q_end = bi - 1; out->t_end = bj - 1;
}

}  // namespace swimpl

// ------------------------------------------------------------------------------------------------ ChainDatabase
static int pick_device(int* cards, int cardsLen) { return (cards && cardsLen > 0) ? cards[0] : 0; }

static void flatten(Chain** chains, int n, std::vector<uint8_t>& codes, std::vector<uint64_t>& off) {
    off.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) {
        SW_ASSERT(chains[i] != NULL, "NULL chain at index %d", i);
        off[i + 1] = off[i] + (uint64_t)chainGetLength(chains[i]);
    }
    codes.resize(off[n] ? off[n] : 1);
    for (int i = 0; i < n; ++i) memcpy(codes.data() + off[i], chainGetCodes(chains[i]), (size_t)chainGetLength(chains[i]));
}

SW_API ChainDatabase* chainDatabaseCreate(Chain** database, int databaseStart, int databaseLen, int* cards, int cardsLen) {
    SW_ASSERT(database != NULL && databaseStart >= 0 && databaseLen >= 0, "chainDatabaseCreate: invalid arguments");
    ChainDatabase* db = new ChainDatabase();       // (S4G_PROFILE counts the device forms, below, under this function's name)
    db->chains = database + databaseStart;
    db->length = databaseLen;
    db->device = pick_device(cards, cardsLen);
    db->cached = NULL;
    if (cards && cardsLen > 0) db->cards.assign(cards, cards + cardsLen); else db->cards.assign(1, 0);
    db->replicas.assign(db->cards.size(), (s4g_db*)NULL);
    db->residues = 0;
    // (a per-query candidate database is a few thousand chains scattered over the heap: every first touch of a Chain is a cache miss,
    // and they are independent -- ask for the objects a few iterations ahead)
    for (int i = 0; i < databaseLen; ++i) {
        if (i + 16 < databaseLen) __builtin_prefetch(db->chains[i + 16]);
        SW_ASSERT(db->chains[i] != NULL, "NULL chain at index %d", i);
        db->residues += chainGetLength(db->chains[i]);
    }
    // The chains stay where they are; which device form the database takes is decided by the first call that uses it
    // (swimpl::packed_replica / cached_replica).  A database too large for the cached form is packed right away: the caller
    // expects the upload to happen here (and a missing GPU to be reported here).
    if (!swimpl::cached_form_fits(db)) {
        // one resident replica per card: queries are split across the cards (the reference hands its `cards` list through
        // chainDatabaseCreate and alignDatabase for the same purpose, database_alignment.cpp:80-86)
        for (size_t slot = 0; slot < db->cards.size(); ++slot) (void)swimpl::packed_replica(db, (int)slot);
    } else {
        int n = 0;
        if (s4g_device_count(&n) != S4G_OK || n <= db->device)
            SW_FATAL("cannot place the database on MI355X device %d: %s (this build has no CPU scoring path)", db->device, s4g_last_error());
    }
    return db;
}

// A part kept flat (readFastaCodesPartNamed): placeholder chains now, real ones for the targets that get reported.
SW_API ChainDatabase* chainDatabaseCreateFlat(const char* codes, const size_t* offsets, const char* names, const size_t* nameOffsets,
                                              int recordsLen, int* cards, int cardsLen) {
    SW_ASSERT(recordsLen >= 0 && (recordsLen == 0 || (codes != NULL && offsets != NULL && names != NULL && nameOffsets != NULL)),
              "chainDatabaseCreateFlat: invalid arguments");
    static ChainData nobody = {(char*)"", (char*)"", (char*)"", 0, 1 << 30};       // what a placeholder's name / letters / codes read as
    ChainDatabase* db = new ChainDatabase();
    db->length = recordsLen;
    db->device = pick_device(cards, cardsLen);
    db->cached = NULL;
    if (cards && cardsLen > 0) db->cards.assign(cards, cards + cardsLen); else db->cards.assign(1, 0);
    db->replicas.assign(db->cards.size(), (s4g_db*)NULL);
    db->flat_codes = codes; db->flat_off = offsets; db->flat_names = names; db->flat_name_off = nameOffsets;
    db->stubs.resize((size_t)recordsLen);
    db->stub_ptrs.resize((size_t)recordsLen ? (size_t)recordsLen : 1);
    db->real.assign((size_t)recordsLen, (Chain*)NULL);
    for (int i = 0; i < recordsLen; ++i) {
        Chain& c = db->stubs[i];
        memset(&c, 0, sizeof c);
        c.data = &nobody; c.length = (int)(offsets[i + 1] - offsets[i]); c.embedded = true;
        c.flat_name = names + nameOffsets[i]; c.flat_name_len = (int)(nameOffsets[i + 1] - nameOffsets[i]);
        db->stub_ptrs[i] = &c;
    }
    db->chains = db->stub_ptrs.data();
    db->residues = recordsLen ? (long long)(offsets[recordsLen] - offsets[0]) : 0;
    for (size_t slot = 0; slot < db->cards.size(); ++slot) (void)swimpl::packed_replica(db, (int)slot);
    return db;
}

namespace swimpl {

// (one std::thread per card runs candidates_on_card / shotgun_on_card against the same database, and two queries on different
// cards may report the same target: the real chain is built once, under the database's lock)
Chain* real_chain(ChainDatabase* db, int idx) {
    if (!db->flat_off) return db->chains[idx];
    std::lock_guard<std::mutex> lk(db->real_mu);
    if (!db->real[idx]) {
        const size_t n = db->flat_off[idx + 1] - db->flat_off[idx];
        std::string letters(n, 'A');
        for (size_t k = 0; k < n; ++k) letters[k] = (char)('A' + db->flat_codes[db->flat_off[idx] + k]);
        const size_t nb = db->flat_name_off[idx], ne = db->flat_name_off[idx + 1];
        db->real[idx] = chainCreate(const_cast<char*>(db->flat_names + nb), (int)(ne - nb), &letters[0], (int)n);
    }
    return db->real[idx];
}

int chain_name_cmp(const Chain* a, const Chain* b) {
    const char* na = a->flat_name ? a->flat_name : chainGetName(const_cast<Chain*>(a));
    const char* nb = b->flat_name ? b->flat_name : chainGetName(const_cast<Chain*>(b));
    const size_t la = a->flat_name ? (size_t)a->flat_name_len : strlen(na), lb = b->flat_name ? (size_t)b->flat_name_len : strlen(nb);
    const int c = memcmp(na, nb, std::min(la, lb));
    return c ? c : (la < lb ? -1 : la > lb ? 1 : 0);
}

// the cached form: up to this many chains (a per-query candidate list is <= 5000, main.cpp:69) and only where it is wanted.
// A Chain carries ONE (dev_tag, dev_addr) pair, so the process has ONE chain-cache device: the first card of the first database
// that takes the cached form claims it; databases whose first card is another device take the packed form instead (a chain
// looked up in another device's arena would miss every time and be appended again).
static const bool g_chain_cache_on = getenv("S4G_NO_CHAIN_CACHE") == NULL;
static std::atomic<int> g_cache_device{-1};
static std::mutex g_cache_mu;                 // the chains' cache words are shared between databases (and host threads)
bool cached_form_fits(const ChainDatabase* db) {
    if (!(g_chain_cache_on && !db->flat_off && db->length > 0 && db->length <= 32768)) return false;
    int want = -1;
    const int mine = db->cards.empty() ? db->device : db->cards[0];
    return g_cache_device.compare_exchange_strong(want, mine) || want == mine;
}

s4g_db* packed_replica(ChainDatabase* db, int slot) {
    if (db->replicas[slot]) return db->replicas[slot];
    Phase ph(0);
    if (db->flat_off) {             // codes back to back already: one upload, packed on the device
        static_assert(sizeof(size_t) == sizeof(uint64_t), "offsets go to the device ABI as they are");
        std::vector<uint64_t> rel((size_t)db->length + 1);
        for (int i = 0; i <= db->length; ++i) rel[i] = db->flat_off[i] - db->flat_off[0];
        s4g_db* rep = NULL;
        int rc = s4g_db_create(&rep, db->cards[slot], reinterpret_cast<const uint8_t*>(db->flat_codes + (db->length ? db->flat_off[0] : 0)), rel.data(), (uint32_t)db->length);
        if (rc != S4G_OK)
            SW_FATAL("cannot place the database on MI355X device %d: %s (this build has no CPU scoring path)", db->cards[slot], s4g_last_error());
        db->replicas[slot] = rep;
        return rep;
    }
    std::vector<const uint8_t*> seqs((size_t)db->length);
    std::vector<uint32_t> lens((size_t)db->length);
    for (int i = 0; i < db->length; ++i) {
        seqs[i] = reinterpret_cast<const uint8_t*>(chainGetCodes(db->chains[i]));
        lens[i] = (uint32_t)chainGetLength(db->chains[i]);
    }
    s4g_db* rep = NULL;
    int rc = s4g_db_create_scattered(&rep, db->cards[slot], seqs.data(), lens.data(), (uint32_t)db->length);
    if (rc != S4G_OK)
        SW_FATAL("cannot place the database on MI355X device %d: %s (this build has no CPU scoring path)", db->cards[slot], s4g_last_error());
    db->replicas[slot] = rep;
    return rep;
}

// Two tables into the first card's chain arena: chains the device holds already are not sent again (their Chain objects say
// where they are), the others are appended to the arena.  Serves s4g_score_pairs_host / s4g_traceback_host only.
s4g_db* cached_replica(ChainDatabase* db) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (db->cached) return db->cached;
    Phase ph(0);
    const size_t n = (size_t)db->length;
    std::vector<const uint8_t*> seqs(n);
    std::vector<uint32_t> lens(n);
    std::vector<uint64_t> tags(n), addrs(n);
    for (size_t i = 0; i < n; ++i) {
        if (i + 16 < n) __builtin_prefetch(db->chains[i + 16]);
        if (i + 8 < n) __builtin_prefetch(db->chains[i + 8]->data);        // chainGetCodes reads through the chain's data block
        Chain* c = db->chains[i];
        seqs[i] = reinterpret_cast<const uint8_t*>(chainGetCodes(c));
        lens[i] = (uint32_t)chainGetLength(c);
        tags[i] = c->dev_tag; addrs[i] = c->dev_addr;
    }
    s4g_db* rep = NULL;
    int rc = s4g_db_create_cached(&rep, db->cards[0], seqs.data(), lens.data(), tags.data(), addrs.data(), (uint32_t)n);
    if (rc == S4G_ERR_NOMEM)        // the arena had to start over (or HBM is short): once more, then the packed form serves the call
        rc = s4g_db_create_cached(&rep, db->cards[0], seqs.data(), lens.data(), tags.data(), addrs.data(), (uint32_t)n);
    if (rc == S4G_ERR_NOMEM) return packed_replica(db, 0);
    if (rc != S4G_OK)
        SW_FATAL("cannot place the database on MI355X device %d: %s (this build has no CPU scoring path)", db->cards[0], s4g_last_error());
    for (size_t i = 0; i < n; ++i) { db->chains[i]->dev_tag = tags[i]; db->chains[i]->dev_addr = addrs[i]; }
    db->cached = rep;
    return rep;
}

}  // namespace swimpl

SW_API void chainDatabaseDelete(ChainDatabase* db) {
    if (!db) return;
    Phase ph(6);
    for (s4g_db* rep : db->replicas) if (rep) s4g_db_destroy(rep);
    if (db->cached) s4g_db_destroy(db->cached);
    delete db;
}

// ------------------------------------------------------------------------------------------------ lazy tracebacks
// The reference's wrapper calls alignDatabase once per query and looks, between two calls, at nothing of an alignment but its target
// index (database_alignment.cpp:83-95); coordinates and paths are read a stage later (select_alignments.cpp:145-150, 244-300).  So a
// per-query call SUBMITS the traceback of its survivors (s4g_traceback_submit) and returns alignments that know their score, value
// and target; the device traces them while the host builds the next query's candidate database, and the first reader of a
// coordinate or a path waits for the batch (s4g_traceback_finish) and fills every alignment of it in.  Same results, bit for bit:
// the kernels, their inputs and the tie rules are those of the synchronous call.  S4G_LAZY_TRACEBACK=0 switches it off.
namespace swimpl {
namespace {
struct TraceBatch {
    s4g_trace_job* job = nullptr;
    s4g_queries* dq = nullptr;                  // owned: the kernels read it, it is destroyed once the job has finished
    int device = 0;
    std::vector<DbAlignment*> members;          // NULL: deleted before the batch was resolved
    std::vector<uint64_t> path_off;
    int alive = 0;
};
std::mutex g_trace_mu;                          // guards every TraceBatch (readers come from the thread pool: select_alignments.cpp:64-69)

void finish_locked(TraceBatch* b) {
    const size_t n = b->members.size();
    std::vector<int32_t> out5(n * 5 + 1);
    std::vector<uint8_t> paths(b->path_off.back() ? b->path_off.back() : 1);
    if (s4g_traceback_finish(b->job, out5.data(), paths.data()) != S4G_OK)
        SW_FATAL("traceback failed on MI355X device %d: %s", b->device, s4g_last_error());
    for (size_t x = 0; x < n; ++x) {
        DbAlignment* a = b->members[x];
        if (!a) continue;
        a->query_start = out5[x * 5]; a->query_end = out5[x * 5 + 1];
        a->target_start = out5[x * 5 + 2]; a->target_end = out5[x * 5 + 3];
        a->path_len = out5[x * 5 + 4];
        free(a->path);
        a->path = (char*)xmalloc(a->path_len ? a->path_len : 1);
        memcpy(a->path, paths.data() + b->path_off[x], a->path_len);
        __atomic_store_n(&a->pending, (void*)NULL, __ATOMIC_RELEASE);          // readers that find it NULL without the lock see the fields above
    }
    if (b->dq) s4g_queries_destroy(b->dq);
    delete b;
}
}  // namespace

void resolve(DbAlignment* a) {
    if (!a || !__atomic_load_n(&a->pending, __ATOMIC_ACQUIRE)) return;          // (cleared under the lock, with release: NULL is final)
    std::lock_guard<std::mutex> lk(g_trace_mu);
    if (a->pending) finish_locked((TraceBatch*)a->pending);
}

// an alignment that is deleted before anybody read it (dbAlignmentsMerge drops the losers of a chunk): it leaves its batch, and the
// last one to leave finishes the job so that its buffers and the query set are released
void forget(DbAlignment* a) {
    if (!__atomic_load_n(&a->pending, __ATOMIC_ACQUIRE)) return;
    std::lock_guard<std::mutex> lk(g_trace_mu);
    TraceBatch* b = (TraceBatch*)a->pending;
    if (!b) return;
    b->members[(size_t)a->pending_slot] = NULL;
    a->pending = NULL;
    if (--b->alive == 0) finish_locked(b);
}

bool lazy_tracebacks() {
    static const bool on = []() { const char* e = getenv("S4G_LAZY_TRACEBACK"); return !(e && e[0] == '0'); }();
    return on;
}
}  // namespace swimpl

// ------------------------------------------------------------------------------------------------ DbAlignment / Alignment
static DbAlignment* make_db_alignment(Chain* query, int query_idx, Chain* target, int target_idx, double value,
                                      Scorer* scorer, const swimpl::PairResult& r) {
    DbAlignment* a = (DbAlignment*)xcalloc(1, sizeof(DbAlignment));
    a->query = query; a->query_idx = query_idx; a->query_start = r.q_start; a->query_end = r.q_end;
    a->target = target; a->target_idx = target_idx; a->target_start = r.t_start; a->target_end = r.t_end;
    a->score = r.score; a->value = value; a->scorer = swimpl::scorer_retain(scorer);
    a->path_len = (int)r.path.size();
    a->path = (char*)xmalloc(a->path_len ? a->path_len : 1);
    if (a->path_len) memcpy(a->path, r.path.data(), a->path_len);
    return a;
}

SW_API Chain* dbAlignmentGetQuery(DbAlignment* a) { return a->query; }
SW_API int dbAlignmentGetQueryIdx(DbAlignment* a) { return a->query_idx; }
SW_API int dbAlignmentGetQueryStart(DbAlignment* a) { swimpl::resolve(a); return a->query_start; }
SW_API int dbAlignmentGetQueryEnd(DbAlignment* a) { swimpl::resolve(a); return a->query_end; }
SW_API Chain* dbAlignmentGetTarget(DbAlignment* a) { return a->target; }
SW_API int dbAlignmentGetTargetIdx(DbAlignment* a) { return a->target_idx; }
SW_API int dbAlignmentGetTargetStart(DbAlignment* a) { swimpl::resolve(a); return a->target_start; }
SW_API int dbAlignmentGetTargetEnd(DbAlignment* a) { swimpl::resolve(a); return a->target_end; }
SW_API int dbAlignmentGetScore(DbAlignment* a) { return a->score; }
SW_API double dbAlignmentGetValue(DbAlignment* a) { return a->value; }
SW_API Scorer* dbAlignmentGetScorer(DbAlignment* a) { return a->scorer; }
SW_API int dbAlignmentGetPathLen(DbAlignment* a) { swimpl::resolve(a); return a->path_len; }
SW_API char dbAlignmentGetMove(DbAlignment* a, int index) { swimpl::resolve(a); return a->path[index]; }
SW_API void dbAlignmentCopyPath(DbAlignment* a, char* dest) { swimpl::resolve(a); memcpy(dest, a->path, a->path_len); }
SW_API void dbAlignmentDelete(DbAlignment* a) { if (a) { swimpl::forget(a); swimpl::scorer_release(a->scorer); free(a->path); free(a); } }

SW_API Alignment* dbAlignmentToAlignment(DbAlignment* d) {
    swimpl::resolve(d);
    Alignment* a = (Alignment*)xcalloc(1, sizeof(Alignment));
    a->query = d->query; a->query_start = d->query_start; a->query_end = d->query_end;
    a->target = d->target; a->target_start = d->target_start; a->target_end = d->target_end;
    a->score = d->score; a->scorer = swimpl::scorer_retain(d->scorer); a->path_len = d->path_len;
    a->path = (char*)xmalloc(d->path_len ? d->path_len : 1);
    if (d->path_len) memcpy(a->path, d->path, d->path_len);
    return a;
}

SW_API Chain* alignmentGetQuery(Alignment* a) { return a->query; }
SW_API int alignmentGetQueryStart(Alignment* a) { return a->query_start; }
SW_API int alignmentGetQueryEnd(Alignment* a) { return a->query_end; }
SW_API Chain* alignmentGetTarget(Alignment* a) { return a->target; }
SW_API int alignmentGetTargetStart(Alignment* a) { return a->target_start; }
SW_API int alignmentGetTargetEnd(Alignment* a) { return a->target_end; }
SW_API int alignmentGetScore(Alignment* a) { return a->score; }
SW_API Scorer* alignmentGetScorer(Alignment* a) { return a->scorer; }
SW_API int alignmentGetPathLen(Alignment* a) { return a->path_len; }
SW_API char alignmentGetMove(Alignment* a, int index) { return a->path[index]; }
SW_API void alignmentCopyPath(Alignment* a, char* dest) { memcpy(dest, a->path, a->path_len); }
SW_API void alignmentDelete(Alignment* a) { if (a) { swimpl::scorer_release(a->scorer); free(a->path); free(a); } }

SW_API void alignPair(Alignment** alignment, int type, Chain* query, Chain* target, Scorer* scorer,
                      int* cards, int cardsLen, Thread* thread) {
    (void)cards; (void)cardsLen; (void)thread;
    SW_ASSERT(type == SW_ALIGN, "alignment type %d is not supported (only SW_ALIGN, the mode sift4g runs)", type);
    swimpl::PairResult r;
    swimpl::sw_traceback(chainGetCodes(query), chainGetLength(query), chainGetCodes(target), chainGetLength(target), scorer, &r);
    Alignment* a = (Alignment*)xcalloc(1, sizeof(Alignment));
    a->query = query; a->query_start = r.q_start; a->query_end = r.q_end;
    a->target = target; a->target_start = r.t_start; a->target_end = r.t_end;
    a->score = r.score; a->scorer = swimpl::scorer_retain(scorer); a->path_len = (int)r.path.size();
    a->path = (char*)xmalloc(a->path_len ? a->path_len : 1);
    if (a->path_len) memcpy(a->path, r.path.data(), a->path_len);
    *alignment = a;
}

// ------------------------------------------------------------------------------------------------ steps 2-4
namespace {

struct Survivor { int idx; int score; double value; };

struct TraceJob {
    Chain* query; Chain* target; int target_idx; double value; Scorer* scorer; int expect_score;
    DbAlignment* result;
};

void* trace_task(void* p) {
    TraceJob* jb = (TraceJob*)p;
    swimpl::PairResult r;
    swimpl::sw_traceback(chainGetCodes(jb->query), chainGetLength(jb->query), chainGetCodes(jb->target),
                         chainGetLength(jb->target), jb->scorer, &r);
    if (r.score != jb->expect_score)
        SW_FATAL("internal error: device score %d and host traceback score %d disagree for target '%s'", jb->expect_score,
                 r.score, chainGetName(jb->target));
    jb->result = make_db_alignment(jb->query, 0, jb->target, jb->target_idx, jb->value, jb->scorer, r);
    return NULL;
}

}  // namespace

// Order of the survivors: value ascending, then score descending, then -- S4G_SELECTION_TIES -- "index" (default): position in
// the database ascending; "reverse": descending; "name": chain name ascending (then position).  swsharp's own order among equal
// values is not known here; with maxAlignments binding it decides which of the tied hits are reported.
// database[idx] = the chain a survivor's idx names.
static void sort_survivors(std::vector<Survivor>& keep, Chain** database) {
    const char* ties = getenv("S4G_SELECTION_TIES");
    const int mode = !ties || !*ties || strcmp(ties, "index") == 0 ? 0 : strcmp(ties, "reverse") == 0 ? 1 : strcmp(ties, "name") == 0 ? 2 : -1;
    if (mode < 0) SW_FATAL("S4G_SELECTION_TIES='%s' is not one of index, reverse, name", ties);
    std::sort(keep.begin(), keep.end(), [&](const Survivor& a, const Survivor& b) {
        if (a.value != b.value) return a.value < b.value;
        if (a.score != b.score) return a.score > b.score;
        if (mode == 1) return a.idx > b.idx;
        if (mode == 2) { const int c = swimpl::chain_name_cmp(database[a.idx], database[b.idx]); if (c) return c < 0; }   // (placeholders of a flat database: their records' names)
        return a.idx < b.idx;
    });
}

// steps 2-3: values, threshold, order, cut
static void select_survivors(std::vector<Survivor>& keep, Chain* query, Chain** database, int databaseLen, const int* scores,
                             int maxAlignments, ValueFunction valueFunction, void* valueFunctionParam, double valueThreshold) {
    keep.clear();
    if (databaseLen <= 0) return;
    std::vector<double> values(databaseLen);
    if (valueFunction) {
        std::vector<int> sc(scores, scores + databaseLen);
        valueFunction(values.data(), sc.data(), query, database, databaseLen, NULL, 0, valueFunctionParam);
    } else {
        for (int i = 0; i < databaseLen; ++i) values[i] = -(double)scores[i];
    }
    for (int i = 0; i < databaseLen; ++i)
        if (scores[i] > 0 && (!valueFunction || values[i] <= valueThreshold)) keep.push_back({i, scores[i], values[i]});
    sort_survivors(keep, database);
    if (maxAlignments >= 0 && (int)keep.size() > maxAlignments) keep.resize(maxAlignments);
}

SW_API void dbAlignmentsFromScores(DbAlignment*** dbAlignments, int* dbAlignmentsLen, int type, Chain* query,
                                   Chain** database, int databaseLen, const int* scores, Scorer* scorer, int maxAlignments,
                                   ValueFunction valueFunction, void* valueFunctionParam, double valueThreshold) {
    SW_ASSERT(type == SW_ALIGN, "alignment type %d is not supported (only SW_ALIGN, the mode sift4g runs)", type);
    *dbAlignments = NULL;
    *dbAlignmentsLen = 0;
    std::vector<Survivor> keep;
    select_survivors(keep, query, database, databaseLen, scores, maxAlignments, valueFunction, valueFunctionParam, valueThreshold);
    const int n = (int)keep.size();
    if (n == 0) return;
    std::vector<TraceJob> jobs(n);
    std::vector<ThreadPoolTask*> tasks(n);
    for (int k = 0; k < n; ++k) {
        jobs[k] = {query, database[keep[k].idx], keep[k].idx, keep[k].value, scorer, keep[k].score, NULL};
        tasks[k] = threadPoolSubmit(trace_task, &jobs[k]);
    }
    DbAlignment** out = (DbAlignment**)xmalloc(sizeof(DbAlignment*) * n);   /* malloc domain: callers free() these arrays */
    for (int k = 0; k < n; ++k) {
        threadPoolTaskWait(tasks[k]);
        threadPoolTaskDelete(tasks[k]);
        out[k] = jobs[k].result;
    }
    *dbAlignments = out;
    *dbAlignmentsLen = n;
}

// ------------------------------------------------------------------------------------------------ step 1 + glue
// Step 4 for a batch of queries: one device traceback of every survivor (keep[i][k].idx indexes cdb->chains), then the
// DbAlignment objects.  `local0` = index of the batch's first query inside the device query set dq.
// lazy_dq != NULL: submit only (see "lazy tracebacks"); *lazy_dq is the query set, whose ownership passes to the batch (set to NULL then).
static void trace_and_build(DbAlignment*** out_lists, int* out_lens, Chain** queries, int query_idx0, uint32_t local0, uint32_t cnt,
                            const std::vector<std::vector<Survivor>>& keep, ChainDatabase* cdb, s4g_db* ddb, s4g_queries* dq, int device,
                            Scorer* scorer, s4g_queries** lazy_dq = NULL) {
    std::vector<uint32_t> pair_q, pair_t;
    std::vector<int32_t> pair_score;
    std::vector<uint64_t> path_off(1, 0);
    for (uint32_t i = 0; i < cnt; ++i)
        for (const Survivor& sv : keep[i]) {
            pair_q.push_back(local0 + i);
            pair_t.push_back((uint32_t)sv.idx);
            pair_score.push_back(sv.score);
            path_off.push_back(path_off.back() + (uint64_t)chainGetLength(queries[i]) + (uint64_t)chainGetLength(cdb->chains[sv.idx]));
        }
    const uint32_t np = (uint32_t)pair_q.size();
    if (lazy_dq && np) {
        swimpl::TraceBatch* b = new swimpl::TraceBatch();
        b->device = device; b->path_off = path_off;
        {
            Phase ph(4);
            if (s4g_traceback_submit(ddb, dq, scorer->dev, np, pair_q.data(), pair_t.data(), pair_score.data(), path_off.data(), &b->job) != S4G_OK)
                SW_FATAL("traceback failed on MI355X device %d: %s", device, s4g_last_error());
        }
        Phase ph_build(5);
        b->dq = *lazy_dq; *lazy_dq = NULL;
        b->members.resize(np); b->alive = (int)np;
        std::lock_guard<std::mutex> lk(swimpl::g_trace_mu);
        uint32_t x = 0;
        for (uint32_t i = 0; i < cnt; ++i) {
            const int n = (int)keep[i].size();
            DbAlignment** out = n ? (DbAlignment**)xmalloc(sizeof(DbAlignment*) * n) : NULL;   /* malloc domain */
            for (int k = 0; k < n; ++k, ++x) {
                const Survivor& sv = keep[i][k];
                DbAlignment* a = (DbAlignment*)xcalloc(1, sizeof(DbAlignment));
                a->query = queries[i]; a->query_idx = query_idx0 + (int)i;
                a->target = swimpl::real_chain(cdb, sv.idx); a->target_idx = sv.idx;
                a->score = sv.score; a->value = sv.value; a->scorer = swimpl::scorer_retain(scorer);
                a->pending = b; a->pending_slot = (int)x;
                b->members[x] = a;
                out[k] = a;
            }
            out_lists[i] = out;
            out_lens[i] = n;
        }
        return;
    }
    std::vector<int32_t> out5((size_t)np * 5);
    std::vector<uint8_t> paths(path_off.back() ? path_off.back() : 1);
    {
        Phase ph(4);
        if (np && s4g_traceback_host(ddb, dq, scorer->dev, np, pair_q.data(), pair_t.data(), pair_score.data(), out5.data(),
                                     path_off.data(), paths.data()) != S4G_OK)
            SW_FATAL("traceback failed on MI355X device %d: %s", device, s4g_last_error());
    }
    Phase ph_build(5);
    uint32_t x = 0;
    for (uint32_t i = 0; i < cnt; ++i) {
        const int n = (int)keep[i].size();
        DbAlignment** out = n ? (DbAlignment**)xmalloc(sizeof(DbAlignment*) * n) : NULL;   /* malloc domain */
        for (int k = 0; k < n; ++k, ++x) {
            const Survivor& sv = keep[i][k];
            DbAlignment* a = (DbAlignment*)xcalloc(1, sizeof(DbAlignment));
            a->query = queries[i]; a->query_idx = query_idx0 + (int)i;
            a->target = swimpl::real_chain(cdb, sv.idx); a->target_idx = sv.idx;
            a->query_start = out5[(size_t)x * 5]; a->query_end = out5[(size_t)x * 5 + 1];
            a->target_start = out5[(size_t)x * 5 + 2]; a->target_end = out5[(size_t)x * 5 + 3];
            a->score = sv.score; a->value = sv.value; a->scorer = swimpl::scorer_retain(scorer);
            a->path_len = out5[(size_t)x * 5 + 4];
            a->path = (char*)xmalloc(a->path_len ? a->path_len : 1);
            memcpy(a->path, paths.data() + path_off[x], a->path_len);
            out[k] = a;
        }
        out_lists[i] = out;
        out_lens[i] = n;
    }
}

// Steps 1-4 for queries [q0, q1) on one card (its own replica of the database); results go to out/outLens[q0..q1).
static void shotgun_on_card(DbAlignment*** out_all, int* out_lens, int type, Chain** all_queries, int q0, int q1, ChainDatabase* cdb,
                            int card_slot, Scorer* scorer, int maxAlignments, ValueFunction valueFunction, void* valueFunctionParam,
                            double valueThreshold, int* indexes, int indexesLen) {
    (void)type;
    Chain** queries = all_queries + q0;
    const int queriesLen = q1 - q0;
    DbAlignment*** dbAlignments_base = out_all + q0;
    int* dbAlignmentsLens_base = out_lens + q0;
    DbAlignment**** dbAlignments = &dbAlignments_base;
    int** dbAlignmentsLens = &dbAlignmentsLens_base;
    const int device = cdb->cards[card_slot];
    if (queriesLen == 0) return;
    // One or a few queries against a few thousand chains -- the reference wrapper's per-query call,
    // database_alignment.cpp:70-92 -- run one wave per pair on chains the device mostly holds already (cached_replica: nothing
    // is packed or sent twice); anything larger takes the packed database and the scan kernels.
    const bool use_cached = card_slot == 0 && queriesLen <= 4 && swimpl::cached_form_fits(cdb);
    s4g_db* ddb = use_cached ? swimpl::cached_replica(cdb) : swimpl::packed_replica(cdb, card_slot);

    std::vector<uint8_t> qcodes; std::vector<uint64_t> qoff;
    flatten(queries, queriesLen, qcodes, qoff);
    s4g_queries* dq = NULL;
    {
        Phase ph(1);
        if (s4g_queries_create(&dq, device, qcodes.data(), qoff.data(), (uint32_t)queriesLen) != S4G_OK)
            SW_FATAL("cannot place the queries on MI355X device %d: %s", device, s4g_last_error());
    }
    // targets to consider: all, or the caller's subset
    std::vector<int> subset;
    std::vector<Chain*> sub_chains;
    if (indexes && indexesLen > 0) {
        subset.assign(indexes, indexes + indexesLen);
        for (int i : subset) { SW_ASSERT(i >= 0 && i < cdb->length, "index %d out of range", i); sub_chains.push_back(cdb->chains[i]); }
    }
    // batches bound the host score matrix to ~1 GiB
    const uint32_t batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)queriesLen, (1ull << 28) / (uint64_t)cdb->length));
    std::vector<int32_t> scores((size_t)batch * cdb->length);
    Chain** targets = subset.empty() ? cdb->chains : sub_chains.data();
    const int n_targets = subset.empty() ? cdb->length : (int)subset.size();

    struct SelectJob {
        Chain* query; Chain** targets; int n; const int32_t* row; const std::vector<int>* subset; int maxAlignments;
        ValueFunction vf; void* vfp; double thr; std::vector<Survivor> keep;
    };
    auto select_task = [](void* p) -> void* {
        SelectJob* jb = (SelectJob*)p;
        select_survivors(jb->keep, jb->query, jb->targets, jb->n, jb->row, jb->maxAlignments, jb->vf, jb->vfp, jb->thr);
        return NULL;
    };

    // with a subset the library scores just those (query, target) pairs (one wave per pair, or a scan + column pick when
    // the subset approaches the whole database: s4g_score_pairs_host decides); rows are then `subset`-indexed
    std::vector<uint32_t> sub_q, sub_t;
    if (use_cached && subset.empty()) {          // the cached form scores explicit pairs: every chain of the database
        subset.resize((size_t)cdb->length);
        for (int i = 0; i < cdb->length; ++i) subset[i] = i;
    }
    const bool explicit_pairs = !subset.empty();
    if (explicit_pairs) {
        sub_t.resize((size_t)batch * subset.size());
        sub_q.resize(sub_t.size());
        for (uint32_t i = 0; i < batch; ++i)
            for (size_t k = 0; k < subset.size(); ++k) sub_t[(size_t)i * subset.size() + k] = (uint32_t)subset[k];
    }
    const size_t row_len = subset.empty() ? (size_t)cdb->length : subset.size();
    for (uint32_t b = 0; b < (uint32_t)queriesLen; b += batch) {
        const uint32_t cnt = std::min<uint32_t>(batch, (uint32_t)queriesLen - b);
        {
            Phase ph(2);
            int rc;
            if (subset.empty()) rc = s4g_scan_host(ddb, dq, b, cnt, scorer->dev, scores.data());
            else {
                for (uint32_t i = 0; i < cnt; ++i)
                    for (size_t k = 0; k < subset.size(); ++k) sub_q[(size_t)i * subset.size() + k] = b + i;
                rc = s4g_score_pairs_host(ddb, dq, scorer->dev, (uint64_t)cnt * subset.size(), sub_q.data(), sub_t.data(), scores.data(), 0);
            }
            if (rc != S4G_OK) SW_FATAL("Smith-Waterman scan failed on MI355X device %d: %s", device, s4g_last_error());
        }
        Phase* ph_sel = new Phase(3);
        // steps 2-3 on the host, one pool task per query (the value callback is the caller's double-precision code)
        std::vector<SelectJob> sel(cnt);
        std::vector<ThreadPoolTask*> tasks(cnt);
        for (uint32_t i = 0; i < cnt; ++i) {
            sel[i] = {queries[b + i], targets, n_targets, scores.data() + (size_t)i * row_len, &subset, maxAlignments,
                      valueFunction, valueFunctionParam, valueThreshold, {}};
            // (a single query selects on this thread: its value function -- eValues -- may then fan out over the pool itself)
            if (cnt == 1) { tasks[i] = NULL; select_task(&sel[i]); } else tasks[i] = threadPoolSubmit(select_task, &sel[i]);
        }
        std::vector<std::vector<Survivor>> keep(cnt);
        for (uint32_t i = 0; i < cnt; ++i) {
            threadPoolTaskWait(tasks[i]);
            threadPoolTaskDelete(tasks[i]);
            keep[i].swap(sel[i].keep);
            if (!subset.empty()) for (Survivor& sv : keep[i]) sv.idx = subset[sv.idx];     // index into the ChainDatabase
        }
        delete ph_sel;
        // the reference's per-query call (cached form, the whole call one batch): the traceback stays on the device while the caller
        // goes on ("lazy tracebacks" above); everything else traces before it returns
        const bool lazy = use_cached && b == 0 && cnt == (uint32_t)queriesLen && swimpl::lazy_tracebacks();
        trace_and_build(*dbAlignments + b, *dbAlignmentsLens + b, queries + b, q0 + (int)b, b, cnt, keep, cdb, ddb, dq, device, scorer,
                        lazy ? &dq : NULL);
    }
    if (dq) s4g_queries_destroy(dq);
}


SW_API void shotgunDatabase(DbAlignment**** dbAlignments, int** dbAlignmentsLens, int type, Chain** queries, int queriesLen,
                            ChainDatabase* cdb, Scorer* scorer, int maxAlignments, ValueFunction valueFunction,
                            void* valueFunctionParam, double valueThreshold, int* indexes, int indexesLen,
                            int* cards, int cardsLen, Thread* thread) {
    (void)cards; (void)cardsLen; (void)thread;   // the ChainDatabase already lives on its cards
    SW_ASSERT(type == SW_ALIGN, "alignment type %d is not supported (only SW_ALIGN, the mode sift4g runs)", type);
    SW_ASSERT(cdb != NULL && scorer != NULL && queries != NULL && queriesLen >= 0, "shotgunDatabase: invalid arguments");
    *dbAlignments = (DbAlignment***)xcalloc(queriesLen ? queriesLen : 1, sizeof(DbAlignment**));
    *dbAlignmentsLens = (int*)xcalloc(queriesLen ? queriesLen : 1, sizeof(int));
    if (queriesLen == 0 || cdb->length == 0) return;
    // query-shard over the cards, balanced in query residues; one host thread per card (upstream's multi-GPU model)
    const int n_cards = std::min<int>((int)cdb->cards.size(), queriesLen);
    std::vector<int> cut(n_cards + 1, 0);
    long long total = 0, acc = 0;
    for (int i = 0; i < queriesLen; ++i) total += chainGetLength(queries[i]);
    for (int i = 0, c = 1; i < queriesLen && c < n_cards; ++i) {
        acc += chainGetLength(queries[i]);
        while (c < n_cards && acc * n_cards >= total * c) cut[c++] = i + 1;
    }
    cut[n_cards] = queriesLen;
    for (int c = 1; c <= n_cards; ++c) cut[c] = std::max(cut[c], cut[c - 1]);
    if (n_cards == 1) {
        shotgun_on_card(*dbAlignments, *dbAlignmentsLens, type, queries, 0, queriesLen, cdb, 0, scorer, maxAlignments, valueFunction,
                        valueFunctionParam, valueThreshold, indexes, indexesLen);
        return;
    }
    std::vector<std::thread> workers;
    for (int c = 0; c < n_cards; ++c)
        workers.emplace_back(shotgun_on_card, *dbAlignments, *dbAlignmentsLens, type, queries, cut[c], cut[c + 1], cdb, c, scorer,
                             maxAlignments, valueFunction, valueFunctionParam, valueThreshold, indexes, indexesLen);
    for (std::thread& w : workers) w.join();
}

// ------------------------------------------------------------------------------------------------ per-query candidate lists
// Queries [q0, q1) on one card: all their (query, candidate) pairs in one device call per batch, selection per query on the
// pool, one device traceback per batch.
static void candidates_on_card(DbAlignment*** out_all, int* out_lens, Chain** all_queries, int q0, int q1, ChainDatabase* cdb,
                               int card_slot, Scorer* scorer, int maxAlignments, ValueFunction valueFunction, void* valueFunctionParam,
                               double valueThreshold, int** candidates, int* candidatesLens) {
    Chain** queries = all_queries + q0;
    const int queriesLen = q1 - q0;
    if (queriesLen <= 0) return;
    const int device = cdb->cards[card_slot];
    s4g_db* ddb = swimpl::packed_replica(cdb, card_slot);
    std::vector<uint8_t> qcodes; std::vector<uint64_t> qoff;
    flatten(queries, queriesLen, qcodes, qoff);
    s4g_queries* dq = NULL;
    {
        Phase ph(1);
        if (s4g_queries_create(&dq, device, qcodes.data(), qoff.data(), (uint32_t)queriesLen) != S4G_OK)
            SW_FATAL("cannot place the queries on MI355X device %d: %s", device, s4g_last_error());
    }
    auto n_cand = [&](int i) { return (candidates[q0 + i] && candidatesLens[q0 + i] > 0) ? candidatesLens[q0 + i] : 0; };
    struct SelectJob {
        Chain* query; std::vector<Chain*> targets; const int32_t* row; int maxAlignments;
        ValueFunction vf; void* vfp; double thr; std::vector<Survivor> keep;
    };
    auto select_task = [](void* p) -> void* {
        SelectJob* jb = (SelectJob*)p;
        select_survivors(jb->keep, jb->query, jb->targets.data(), (int)jb->targets.size(), jb->row, jb->maxAlignments, jb->vf, jb->vfp, jb->thr);
        return NULL;
    };
    const uint64_t kBatchPairs = 1ull << 25;          // bounds the host pair arrays to ~0.4 GB
    std::vector<uint32_t> pq, pt;
    std::vector<int32_t> scores;
    for (int b = 0; b < queriesLen;) {
        int e = b;
        uint64_t pairs = 0;
        while (e < queriesLen && (e == b || pairs + (uint64_t)n_cand(e) <= kBatchPairs)) pairs += (uint64_t)n_cand(e++);
        const uint32_t cnt = (uint32_t)(e - b);
        pq.resize(pairs); pt.resize(pairs); scores.resize(pairs ? pairs : 1);
        std::vector<uint64_t> row0(cnt + 1, 0);
        for (uint32_t i = 0; i < cnt; ++i) {
            const int n = n_cand(b + (int)i);
            const int* list = candidates[q0 + b + (int)i];
            for (int k = 0; k < n; ++k) {
                SW_ASSERT(list[k] >= 0 && list[k] < cdb->length, "candidate index %d out of range", list[k]);
                pq[row0[i] + k] = (uint32_t)b + i;
                pt[row0[i] + k] = (uint32_t)list[k];
            }
            row0[i + 1] = row0[i] + (uint64_t)n;
        }
        if (pairs) {
            Phase ph(2);
            if (s4g_score_pairs_host(ddb, dq, scorer->dev, pairs, pq.data(), pt.data(), scores.data(), 0) != S4G_OK)
                SW_FATAL("Smith-Waterman scoring failed on MI355X device %d: %s", device, s4g_last_error());
        }
        Phase* ph_sel = new Phase(3);
        std::vector<SelectJob> sel(cnt);
        std::vector<ThreadPoolTask*> tasks(cnt, NULL);
        for (uint32_t i = 0; i < cnt; ++i) {
            const int n = n_cand(b + (int)i);
            if (n == 0) continue;
            const int* list = candidates[q0 + b + (int)i];
            sel[i].query = queries[b + i]; sel[i].row = scores.data() + row0[i]; sel[i].maxAlignments = maxAlignments;
            sel[i].vf = valueFunction; sel[i].vfp = valueFunctionParam; sel[i].thr = valueThreshold;
            sel[i].targets.resize(n);
            for (int k = 0; k < n; ++k) sel[i].targets[k] = cdb->chains[list[k]];
            tasks[i] = threadPoolSubmit(select_task, &sel[i]);
        }
        std::vector<std::vector<Survivor>> keep(cnt);
        for (uint32_t i = 0; i < cnt; ++i) {
            if (!tasks[i]) continue;
            threadPoolTaskWait(tasks[i]);
            threadPoolTaskDelete(tasks[i]);
            keep[i].swap(sel[i].keep);
            const int* list = candidates[q0 + b + (int)i];
            for (Survivor& sv : keep[i]) sv.idx = list[sv.idx];        // position in the list -> index into the ChainDatabase
        }
        delete ph_sel;
        trace_and_build(out_all + q0 + b, out_lens + q0 + b, queries + b, q0 + b, (uint32_t)b, cnt, keep, cdb, ddb, dq, device, scorer);
        b = e;
    }
    s4g_queries_destroy(dq);
}

SW_API void shotgunDatabaseCandidates(DbAlignment**** dbAlignments, int** dbAlignmentsLens, int type, Chain** queries, int queriesLen,
                                      ChainDatabase* cdb, Scorer* scorer, int maxAlignments, ValueFunction valueFunction,
                                      void* valueFunctionParam, double valueThreshold, int** candidates, int* candidatesLens,
                                      int* cards, int cardsLen, Thread* thread) {
    (void)cards; (void)cardsLen; (void)thread;
    SW_ASSERT(type == SW_ALIGN, "alignment type %d is not supported (only SW_ALIGN, the mode sift4g runs)", type);
    SW_ASSERT(cdb != NULL && scorer != NULL && queries != NULL && queriesLen >= 0 && (queriesLen == 0 || (candidates && candidatesLens)),
              "shotgunDatabaseCandidates: invalid arguments");
    *dbAlignments = (DbAlignment***)xcalloc(queriesLen ? queriesLen : 1, sizeof(DbAlignment**));
    *dbAlignmentsLens = (int*)xcalloc(queriesLen ? queriesLen : 1, sizeof(int));
    if (queriesLen == 0 || cdb->length == 0) return;
    // query-shard over the cards, balanced in (query residues x candidates)
    const int n_cards = std::min<int>((int)cdb->cards.size(), queriesLen);
    std::vector<int> cut(n_cards + 1, 0);
    auto weight = [&](int i) { return (long long)chainGetLength(queries[i]) * (candidates[i] && candidatesLens[i] > 0 ? candidatesLens[i] : 0) + 1; };
    long long total = 0, acc = 0;
    for (int i = 0; i < queriesLen; ++i) total += weight(i);
    for (int i = 0, c = 1; i < queriesLen && c < n_cards; ++i) {
        acc += weight(i);
        while (c < n_cards && acc * n_cards >= total * c) cut[c++] = i + 1;
    }
    cut[n_cards] = queriesLen;
    for (int c = 1; c <= n_cards; ++c) cut[c] = std::max(cut[c], cut[c - 1]);
    if (n_cards == 1) {
        candidates_on_card(*dbAlignments, *dbAlignmentsLens, queries, 0, queriesLen, cdb, 0, scorer, maxAlignments, valueFunction,
                           valueFunctionParam, valueThreshold, candidates, candidatesLens);
        return;
    }
    std::vector<std::thread> workers;
    for (int c = 0; c < n_cards; ++c)
        workers.emplace_back(candidates_on_card, *dbAlignments, *dbAlignmentsLens, queries, cut[c], cut[c + 1], cdb, c, scorer,
                             maxAlignments, valueFunction, valueFunctionParam, valueThreshold, candidates, candidatesLens);
    for (std::thread& w : workers) w.join();
}

// ------------------------------------------------------------------------------------------------ e-value driven selection
// shotgunDatabaseEValues: the same result as shotgunDatabase(..., valueFunction = eValues, param = eValueParams, ...), but
// the library knows the value function, so nothing of size Q x N ever leaves the device:
//   * per query a CONSERVATIVE integer score cut: the smallest score that can reach maxEvalue for ANY target length
//     present in the database (e-values fall monotonically with the score);
//   * streaming scan + exact device top-k' (k' > maxAlignments) above that cut (s4g_scan_topk_host);
//   * exact double-precision e-values, threshold and final order on the host, on the <= k' survivors only;
//   * exactness check: everything that was left out scores <= the list's lowest score, so its e-value is at least
//     Emin(that score) = the best e-value that score can have for any length in the database.  If that is not strictly
//     worse than the last kept alignment (or than maxEvalue when fewer than maxAlignments were kept), k' is raised and,
//     as a last resort, the query falls back to the full-row path.
namespace {

struct EvJob { Chain* query; int m; };

double emin_over_lengths(int score, Chain* query, const std::vector<Chain*>& length_reps, EValueParams* ev) {
    std::vector<double> v(length_reps.size());
    std::vector<int> sc(length_reps.size(), score);
    eValues(v.data(), sc.data(), query, const_cast<Chain**>(length_reps.data()), (int)length_reps.size(), NULL, 0, ev);
    double best = v[0];
    for (double x : v) best = std::min(best, x);
    return best;
}

}  // namespace

static void shotgun_evalues_on_card(DbAlignment*** out_all, int* out_lens, Chain** all_queries, int q0, int q1, ChainDatabase* cdb,
                                    int card_slot, Scorer* scorer, int maxAlignments, EValueParams* ev, double maxEvalue,
                                    const std::vector<Chain*>* length_reps) {
    Chain** queries = all_queries + q0;
    const int nq = q1 - q0;
    if (nq == 0) return;
    const int device = cdb->cards[card_slot];
    s4g_db* ddb = swimpl::packed_replica(cdb, card_slot);
    std::vector<uint8_t> qcodes; std::vector<uint64_t> qoff;
    flatten(queries, nq, qcodes, qoff);
    s4g_queries* dq = NULL;
    if (s4g_queries_create(&dq, device, qcodes.data(), qoff.data(), (uint32_t)nq) != S4G_OK)
        SW_FATAL("cannot place the queries on MI355X device %d: %s", device, s4g_last_error());

    // conservative score cut per query (cached per query length: the e-value depends on the query only through it)
    std::vector<int32_t> min_score(nq);
    {
        Phase ph(3);
        std::vector<std::pair<int, int>> cache;
        for (int i = 0; i < nq; ++i) {
            const int m = chainGetLength(queries[i]);
            int found = -1;
            for (auto& c : cache) if (c.first == m) found = c.second;
            if (found < 0) {
                int lo = 1, hi = std::max(2, scorerGetMaxScore(scorer) * std::max(m, 1) + 1);      // E(hi) is astronomically small
                if (emin_over_lengths(hi, queries[i], *length_reps, ev) > maxEvalue) lo = hi;       // nothing can pass
                while (lo < hi) {
                    const int mid = lo + (hi - lo) / 2;
                    if (emin_over_lengths(mid, queries[i], *length_reps, ev) <= maxEvalue) hi = mid; else lo = mid + 1;
                }
                found = lo;
                cache.push_back({m, found});
            }
            min_score[i] = found;
        }
    }
    const int kmax = 4096;
    int k = std::min(kmax, std::max(4 * std::max(maxAlignments, 1), maxAlignments + 256));
    if (maxAlignments < 0) k = kmax;
    std::vector<uint32_t> ids((size_t)nq * k), cnt(nq);
    std::vector<int32_t> sc((size_t)nq * k);
    {
        Phase ph(2);
        if (s4g_scan_topk_host(ddb, dq, 0, (uint32_t)nq, scorer->dev, (uint32_t)k, min_score.data(), ids.data(), sc.data(), cnt.data(), 0) != S4G_OK)
            SW_FATAL("Smith-Waterman scan failed on MI355X device %d: %s", device, s4g_last_error());
    }
    std::vector<std::vector<Survivor>> keep(nq);
    std::vector<int> redo;
    {
        Phase ph(3);
        std::vector<Chain*> cand; std::vector<int> csc; std::vector<double> val;
        for (int i = 0; i < nq; ++i) {
            const uint32_t c = cnt[i];
            cand.resize(c); csc.resize(c); val.resize(c);
            for (uint32_t x = 0; x < c; ++x) { cand[x] = cdb->chains[ids[(size_t)i * k + x]]; csc[x] = sc[(size_t)i * k + x]; }
            if (c) eValues(val.data(), csc.data(), queries[i], cand.data(), (int)c, NULL, 0, ev);
            std::vector<Survivor>& kp = keep[i];
            for (uint32_t x = 0; x < c; ++x)
                if (val[x] <= maxEvalue) kp.push_back({(int)ids[(size_t)i * k + x], csc[x], val[x]});
            sort_survivors(kp, cdb->chains);
            if (maxAlignments >= 0 && (int)kp.size() > maxAlignments) kp.resize(maxAlignments);
            if ((int)c == k) {        // the device list was cut: prove that nothing left out could have made it
                const double bar = (maxAlignments >= 0 && (int)kp.size() == maxAlignments && !kp.empty()) ? kp.back().value : maxEvalue;
                if (!(emin_over_lengths(csc[c - 1], queries[i], *length_reps, ev) > bar)) redo.push_back(i);
            }
        }
    }
    // rare: a query with more near-threshold hits than k' -- full row, the generic path
    for (int i : redo) {
        std::vector<int32_t> row(cdb->length);
        if (s4g_scan_host(ddb, dq, (uint32_t)i, 1, scorer->dev, row.data()) != S4G_OK)
            SW_FATAL("Smith-Waterman scan failed on MI355X device %d: %s", device, s4g_last_error());
        struct P { EValueParams* ev; } prm = {ev};
        auto vf = [](double* values, int* scores, Chain* query, Chain** database, int n, int* cards, int cardsLen, void* param) {
            eValues(values, scores, query, database, n, cards, cardsLen, ((P*)param)->ev);
        };
        select_survivors(keep[i], queries[i], cdb->chains, cdb->length, row.data(), maxAlignments, vf, &prm, maxEvalue);
    }
    trace_and_build(out_all + q0, out_lens + q0, queries, q0, 0, (uint32_t)nq, keep, cdb, ddb, dq, device, scorer);
    s4g_queries_destroy(dq);
}

SW_API void shotgunDatabaseEValues(DbAlignment**** dbAlignments, int** dbAlignmentsLens, int type, Chain** queries, int queriesLen,
                                   ChainDatabase* cdb, Scorer* scorer, int maxAlignments, EValueParams* eValueParams, double maxEvalue,
                                   int* cards, int cardsLen, Thread* thread) {
    (void)cards; (void)cardsLen; (void)thread;
    SW_ASSERT(type == SW_ALIGN, "alignment type %d is not supported (only SW_ALIGN, the mode sift4g runs)", type);
    SW_ASSERT(cdb != NULL && scorer != NULL && queries != NULL && queriesLen >= 0 && eValueParams != NULL, "shotgunDatabaseEValues: invalid arguments");
    *dbAlignments = (DbAlignment***)xcalloc(queriesLen ? queriesLen : 1, sizeof(DbAlignment**));
    *dbAlignmentsLens = (int*)xcalloc(queriesLen ? queriesLen : 1, sizeof(int));
    if (queriesLen == 0 || cdb->length == 0) return;
    // one representative chain per distinct target length (the e-value depends on a target only through its length)
    std::vector<Chain*> reps;
    {
        std::vector<std::pair<int, int>> byLen(cdb->length);
        for (int i = 0; i < cdb->length; ++i) byLen[i] = {chainGetLength(cdb->chains[i]), i};
        std::sort(byLen.begin(), byLen.end());
        for (int i = 0; i < cdb->length; ++i)
            if (i == 0 || byLen[i].first != byLen[i - 1].first) reps.push_back(cdb->chains[byLen[i].second]);
    }
    const int n_cards = std::min<int>((int)cdb->cards.size(), queriesLen);
    std::vector<std::thread> workers;
    for (int c = 0; c < n_cards; ++c) {
        const int a = (int)((long long)queriesLen * c / n_cards), b = (int)((long long)queriesLen * (c + 1) / n_cards);
        if (n_cards == 1)
            shotgun_evalues_on_card(*dbAlignments, *dbAlignmentsLens, queries, a, b, cdb, c, scorer, maxAlignments, eValueParams, maxEvalue, &reps);
        else
            workers.emplace_back(shotgun_evalues_on_card, *dbAlignments, *dbAlignmentsLens, queries, a, b, cdb, c, scorer, maxAlignments,
                                 eValueParams, maxEvalue, &reps);
    }
    for (std::thread& w : workers) w.join();
}

SW_API void alignDatabase(DbAlignment*** dbAlignments, int* dbAlignmentsLen, int type, Chain* query,
                          ChainDatabase* chainDatabase, Scorer* scorer, int maxAlignments, ValueFunction valueFunction,
                          void* valueFunctionParam, double valueThreshold, int* indexes, int indexesLen,
                          int* cards, int cardsLen, Thread* thread) {
    DbAlignment*** all = NULL; int* lens = NULL;
    shotgunDatabase(&all, &lens, type, &query, 1, chainDatabase, scorer, maxAlignments, valueFunction, valueFunctionParam,
                    valueThreshold, indexes, indexesLen, cards, cardsLen, thread);
    *dbAlignments = all[0];
    *dbAlignmentsLen = lens[0];
    for (int k = 0; k < lens[0]; ++k) all[0][k]->query_idx = 0;
    free(all);
    free(lens);
}

// ------------------------------------------------------------------------------------------------ merge / delete
static bool db_alignment_before(const DbAlignment* a, const DbAlignment* b) {
    if (a->value != b->value) return a->value < b->value;
    return a->score > b->score;
}

// Merges, per query, the sorted list src into the sorted list dst and keeps the best maxAlignments
// (the cross-chunk merge of database_alignment.cpp:97-104; also the multi-GPU merge).  Entries that do not
// survive are deleted; src lists are left empty so the caller's deleteShotgunDatabase(src) frees only containers.
SW_API void dbAlignmentsMerge(DbAlignment*** dst, int* dstLens, DbAlignment*** src, int* srcLens, int n, int maxAlignments) {
    for (int i = 0; i < n; ++i) {
        const int a = dstLens[i], b = srcLens[i];
        if (b == 0) continue;
        const int total = a + b;
        const int keep = (maxAlignments >= 0 && total > maxAlignments) ? maxAlignments : total;
        DbAlignment** merged = (DbAlignment**)xmalloc(sizeof(DbAlignment*) * (total ? total : 1));
        int x = 0, y = 0, k = 0;
        while (x < a || y < b) {
            DbAlignment* pick;
            if (y >= b || (x < a && !db_alignment_before(src[i][y], dst[i][x]))) pick = dst[i][x++];   // stable: dst first on ties
            else pick = src[i][y++];
            if (k < keep) merged[k++] = pick; else dbAlignmentDelete(pick);
        }
        free(dst[i]);
        dst[i] = merged;
        dstLens[i] = keep;
        free(src[i]);
        src[i] = NULL;
        srcLens[i] = 0;
    }
}

SW_API void deleteShotgunDatabase(DbAlignment*** dbAlignments, int* dbAlignmentsLens, int dbAlignmentsLen) {
    if (dbAlignments) {
        for (int i = 0; i < dbAlignmentsLen; ++i) {
            if (!dbAlignments[i]) continue;
            for (int j = 0; j < dbAlignmentsLens[i]; ++j) dbAlignmentDelete(dbAlignments[i][j]);
            free(dbAlignments[i]);
        }
        free(dbAlignments);
    }
    free(dbAlignmentsLens);
}

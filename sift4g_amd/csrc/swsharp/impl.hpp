// impl.hpp -- internal object layouts of libswsharp (the swsharp-compatible host API, include/swsharp/*.h).
#pragma once
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/s4g_hip.h"
#include "../../../include/swsharp/evalue.h"
#include "../../../include/swsharp/swsharp.h"

#define SW_API extern "C" __attribute__((visibility("default")))

// sift4g's own fatal-error style (/root/reference/sift4g/src/utils.hpp:13-19): message on stderr, exit(-1).
#define SW_FATAL(fmt, ...)                                            \
    do {                                                              \
        fprintf(stderr, "[ERROR]: " fmt "\n", ##__VA_ARGS__);         \
        exit(-1);                                                     \
    } while (0)
#define SW_ASSERT(expr, fmt, ...) do { if (!(expr)) SW_FATAL(fmt, ##__VA_ARGS__); } while (0)

// malloc-domain allocations (callers free() what the API hands out) that fail the way the rest of the library does
inline void* sw_alloc_check(void* p, size_t bytes) { if (!p && bytes) SW_FATAL("out of memory (%zu bytes)", bytes); return p; }
inline void* xmalloc(size_t bytes) { return sw_alloc_check(malloc(bytes ? bytes : 1), bytes); }
inline void* xcalloc(size_t n, size_t size) { return sw_alloc_check(calloc(n ? n : 1, size), n * size); }
inline void* xrealloc(void* p, size_t bytes) { return sw_alloc_check(realloc(p, bytes ? bytes : 1), bytes); }

struct ChainData {           // shared storage of a chain and its views
    char* name;
    char* chars;
    char* codes;
    int length;
    int refs;
};

struct Chain {
    ChainData* data;
    int start;               // view window into data, inclusive
    int length;
    bool reverse;
    bool embedded;           // this Chain object lives inside its ChainData's block (the chain the block was made for)
    char* rev_chars;         // materialised only for reverse views
    char* rev_codes;
    // where this chain's residues live on a device, if it has been sent there (s4g_db_create_cached keeps these up to date;
    // zero = never sent).  A chain is a candidate of many queries: the wrapper's per-query chainDatabaseCreate finds it here.
    // Both words belong to ONE device, the process's chain-cache device (align.cpp g_cache_device: the first card of the first
    // database that could take the cached form); they are read and written under align.cpp's g_cache_mu only.
    uint64_t dev_tag, dev_addr;
    // a placeholder chain of a flat database (chainDatabaseCreateFlat): the record's header text, not terminated.  NULL elsewhere.
    const char* flat_name;
    int flat_name_len;
};

struct Scorer {
    std::string name;
    int32_t table[26 * 26];
    int gap_open, gap_extend, max_score;
    s4g_scorer* dev;         // device-side twin (host object inside libs4g_hip)
    // The reference deletes its scorer BEFORE it writes the alignments (main.cpp:223 then :227), and the alignments'
    // statistics need the table: every DbAlignment / Alignment holds a reference, scorerDelete drops the caller's.
    std::atomic<int> refs{1};
};

struct ChainDatabase {
    Chain** chains;          // borrowed: database + databaseStart
    int length;
    int device;              // first card
    long long residues;
    std::vector<int> cards;          // every card the database was created for (--cards 0123...; main.cpp:254-262)
    // Device forms, made when a call first needs them (swimpl::packed_replica / cached_replica, align.cpp):
    //   replicas[slot]  the packed, length-sorted tiles the scan / list kernels read -- one resident copy per card;
    //   cached          two tables into the device's chain arena (s4g_db_create_cached), first card only: what a call with one
    //                   or a few queries runs on (one wave per pair), without sending the chains again.
    std::vector<s4g_db*> replicas;
    s4g_db* cached;
    // chainDatabaseCreateFlat: the part's four arrays (borrowed), a placeholder chain per record (length only) that `chains` points
    // at, and the real chains made so far (swimpl::real_chain)
    const char* flat_codes = NULL;
    const size_t* flat_off = NULL;
    const char* flat_names = NULL;
    const size_t* flat_name_off = NULL;
    std::vector<Chain> stubs;
    std::vector<Chain*> stub_ptrs;
    std::vector<Chain*> real;
    std::mutex real_mu;      // swimpl::real_chain builds real[idx] on first use, from any card's thread
};

struct DbAlignment {
    Chain* query; int query_idx, query_start, query_end;
    Chain* target; int target_idx, target_start, target_end;
    int score; double value;
    Scorer* scorer;
    char* path; int path_len;
    // a traceback still on the device (align.cpp, "lazy tracebacks"): the batch this alignment belongs to and its slot there; the
    // coordinates and the path above are filled in by the first call that needs any of them (swimpl::resolve)
    void* pending; int pending_slot;
};

struct Alignment {
    Chain* query; int query_start, query_end;
    Chain* target; int target_start, target_end;
    int score;
    Scorer* scorer;
    char* path; int path_len;
};

struct EValueParams {
    long long length;
    double lambda, K, logK, H, a, b, alpha, beta, sigma, tau;
    bool gapped_row_found;
};

namespace swimpl {

struct PairResult {
    int score, q_start, q_end, t_start, t_end;
    std::vector<char> path;   // MOVE_* from start to end
};
// Waits for the device traceback an alignment still depends on, if any (align.cpp); every reader of coordinates or path calls it first.
void resolve(DbAlignment* a);
void forget(DbAlignment* a);             // a DbAlignment is being deleted: it leaves the batch it may still wait for
// Host Smith-Waterman with traceback (the "traceback of the <= maxAlignments survivors" stage).
void sw_traceback(const char* q, int lq, const char* t, int lt, const Scorer* sc, PairResult* out);

bool on_pool_thread();             // core.cpp: the calling thread is a worker of the thread pool
int pool_threads();                // workers of the pool (0: none, tasks run inline)

// the chain of database entry idx that alignments point at: the entry itself, or -- flat databases -- the real chain, built on first use
Chain* real_chain(ChainDatabase* db, int idx);

// strcmp of the names two database entries are reported under (a placeholder's name is its record's header text)
int chain_name_cmp(const Chain* a, const Chain* b);

// device forms of a ChainDatabase, made on first use (align.cpp)
bool cached_form_fits(const ChainDatabase* db);
s4g_db* packed_replica(ChainDatabase* db, int slot);
s4g_db* cached_replica(ChainDatabase* db);

inline Scorer* scorer_retain(Scorer* s) { if (s) s->refs.fetch_add(1, std::memory_order_relaxed); return s; }
void scorer_release(Scorer* s);    // core.cpp: destroys the scorer with its last reference

}  // namespace swimpl

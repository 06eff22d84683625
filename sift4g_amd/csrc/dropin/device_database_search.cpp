// device_database_search.cpp -- MI355X replacement for the reference's k-mer prefilter stage.
//
// Defines `searchDatabase(...)` with the signature the reference declares in
// /root/reference/sift4g/src/database_search.hpp:17-19 and replaces the reference's database_search.cpp + hash.cpp in
// `make dropin-devsearch`: a maintainer swaps these source files, nothing else.
//
// What it keeps (reference database_search.cpp:66-183): the database is read in ~250 MB parts with the swsharp FASTA
// readers; every sequence of a part is scored against every query by shared k-mers (longest increasing subsequence of the
// hit positions over the sequence's length); the best `max_candidates` per query survive across parts; the lists come back
// as ascending database indices, and the function returns the number of database residues (the e-value search space,
// main.cpp:204,210).  Same banner and progress lines on stderr.
//
// What it changes: the k-mer index of the queries, the hit lists, the LIS and the running best lists live on the device
// (include/s4g_hip.h: s4g_prefilter_*), one submission per part, overlapped with the reading of the next part -- the host
// threads of the reference's thread pool (:101-123) are not used, so `num_threads` no longer changes the result: among
// candidates whose score EQUALS the score at the cut the reference's choice depends on it (and on std::sort), here the lower
// database index always wins.  Every candidate strictly above the cut is the same.
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <future>
#include <string>
#include <vector>

#include "utils.hpp"              // the reference's own: databaseLog
#include "database_search.hpp"    // the reference's own: the declaration this file implements
#include "flat_database.hpp"
#include "s4g_hip.h"

namespace {

// ~250 MB of sequence per part, as the reference reads it (database_search.cpp:17); S4G_PREFILTER_CHUNK_BYTES overrides it
// (the tests use it to run the several-parts logic -- running lists across parts, the overlap of reading and device work --
// on small inputs)
uint32_t chunk_bytes() {
    const char* v = getenv("S4G_PREFILTER_CHUNK_BYTES");
    const long long n = v ? atoll(v) : 0;
    return n > 0 ? (uint32_t)n : 250000000u;
}

// the library's error text is kept per thread, and the device calls below run on helper threads: they hand the text back
std::string g_async_error;
int remember(int rc) {
    if (rc != S4G_OK) g_async_error = s4g_last_error();
    return rc;
}
void die(const char* what) {
    fprintf(stderr, "[ERROR] %s: %s\n", what, g_async_error.empty() ? s4g_last_error() : g_async_error.c_str());
    exit(1);
}

struct Part {                       // one part of the database as the device takes it
    char* codes = nullptr;
    size_t* off = nullptr;
    char* names = nullptr;          // the records' header text: not for the device, for the alignment stage (flat_database.hpp)
    size_t* name_off = nullptr;
    uint32_t n = 0, first_id = 0;
};
static_assert(sizeof(size_t) == sizeof(uint64_t), "offsets are handed to the device ABI as they are");

double seconds_since(std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
}

// the card this stage works on: the first one of the user's --cards list (main.cpp validates the list with cudaCheckCards before
// it calls us; the reference's searchDatabase signature has no card argument), S4G_PREFILTER_DEVICE overrides
int device_of_choice() {
    const char* v = getenv("S4G_PREFILTER_DEVICE");
    return v ? atoi(v) : swsharpPreferredCard();
}

}  // namespace

uint64_t searchDatabase(std::vector<std::vector<uint32_t>>& dst, const std::string& database_path, Chain** queries,
                        int32_t queries_length, uint32_t kmer_length, uint32_t max_candidates, uint32_t num_threads) {
    (void)num_threads;
    fprintf(stderr, "** Searching database for candidate sequences **\n");

    // the queries, flattened: createHash's input (database_search.cpp:71-72)
    struct { std::vector<uint8_t> codes; std::vector<uint64_t> off; } q;
    q.off.push_back(0);
    for (int32_t i = 0; i < queries_length; ++i) {
        const char* c = chainGetCodes(queries[i]);
        const int n = chainGetLength(queries[i]);
        q.codes.insert(q.codes.end(), (const uint8_t*)c, (const uint8_t*)c + n);
        q.off.push_back(q.codes.size());
    }
    // the index is built while the first part of the database is being read (and while the device finishes warming up)
    s4g_prefilter* filter = nullptr;
    std::future<int> created = std::async(std::launch::async, [&]() {
        return remember(s4g_prefilter_create(&filter, device_of_choice(), q.codes.data(), q.off.data(), (uint32_t)queries_length,
                                             kmer_length, max_candidates));
    });

    Chain** database = nullptr;
    int database_length = 0;
    FILE* handle = nullptr;
    int serialized = 0;
    readFastaChainsPartInit(&database, &database_length, &handle, &serialized, database_path.c_str());

    double t_read = 0, t_wait = 0;                       // S4G_PROFILE=1: where the stage's time goes
    const uint32_t database_chunk = chunk_bytes();
    uint64_t database_cells = 0;
    uint32_t database_start = 0;
    uint32_t part = 1;
    const float part_size = database_chunk / (float)1000000000;
    Part parts[2];
    std::future<int> pending;                            // the device's work on the previous part
    // the parts stay behind for the alignment stage (one pass over the file) while they fit the budget
    s4g_dropin::FlatDatabase& keep = s4g_dropin::flat_database();
    keep.clear();
    keep.path = database_path;
    bool keeping = s4g_dropin::flat_database_budget() > 0;
    auto retire = [&](Part& p) {                         // a part the device is through with: kept or freed
        if (!p.codes) return;
        const size_t b = p.off[p.n] + p.name_off[p.n] + (size_t)(p.n + 1) * 2 * sizeof(size_t);
        if (keeping && keep.bytes + b <= s4g_dropin::flat_database_budget()) {
            s4g_dropin::FlatPart f;
            f.codes = p.codes; f.off = p.off; f.names = p.names; f.name_off = p.name_off; f.n = (int)p.n; f.first_id = p.first_id;
            keep.parts.push_back(f);
            keep.bytes += b;
        } else {
            if (keeping) { keeping = false; keep.clear(); }
            free(p.codes); free(p.off); free(p.names); free(p.name_off);
        }
        p = Part();
    };
    while (true) {
        // the part as flat residue codes (readFastaCodesPart: the records and part boundaries of readFastaChainsPart, :93-97,
        // without building the chains this stage would delete again at :125-129)
        auto t0 = std::chrono::steady_clock::now();
        Part& p = parts[part & 1];                       // the other one may still be on its way to the device
        retire(p);                                       // (the part two steps back: the device finished it before the last one started)
        int records = 0;
        const int status = readFastaCodesPartNamed(&p.codes, &p.off, &p.names, &p.name_off, &records, handle, serialized, database_chunk);
        p.n = (uint32_t)records;
        p.first_id = database_start;
        database_cells += p.off[records];
        t_read += seconds_since(t0);
        databaseLog(part, part_size, 0);

        t0 = std::chrono::steady_clock::now();
        if (created.valid() && created.get() != S4G_OK) die("prefilter");
        if (pending.valid() && pending.get() != S4G_OK) die("prefilter");
        pending = std::async(std::launch::async, [filter, &p]() {
            return remember(s4g_prefilter_add(filter, (const uint8_t*)p.codes, (const uint64_t*)p.off, p.n, p.first_id));
        });
        t_wait += seconds_since(t0);

        databaseLog(part, part_size, 100);
        ++part;
        database_start += (uint32_t)records;
        if (status == 0) break;
    }
    auto t0 = std::chrono::steady_clock::now();
    if (pending.valid() && pending.get() != S4G_OK) die("prefilter");
    t_wait += seconds_since(t0);
    t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "\n\n");
    // in database order: the part two steps back was retired inside the loop, the last two are left
    { Part& a = parts[part & 1]; Part& b = parts[(part + 1) & 1]; if (a.codes && b.codes && a.first_id > b.first_id) { retire(b); retire(a); } else { retire(a); retire(b); } }
    keep.complete = keeping;
    if (!keeping) keep.clear();

    fclose(handle);
    deleteFastaChains(database, database_length);

    // dst (database_search.cpp:166-178): ascending ids per query
    std::vector<uint32_t> ids((size_t)std::max(1, queries_length) * max_candidates), counts((size_t)std::max(1, queries_length));
    if (s4g_prefilter_finish(filter, ids.data(), counts.data(), nullptr) != S4G_OK) die("prefilter");
    const double t_lists = seconds_since(t0);
    // the handle's buffers (a few GB of device memory, a few hundred MB of pinned staging) are given back underneath what follows;
    // the static future's destructor waits for that at exit at the latest
    static std::future<void> cleanup;
    cleanup = std::async(std::launch::async, [filter]() { s4g_prefilter_destroy(filter); });
    if (getenv("S4G_PROFILE"))
        fprintf(stderr, "[s4g profile] prefilter stage: reading FASTA into flat codes %.3f s, waiting for the device %.3f s, lists to the host %.3f s\n",
                t_read, t_wait, t_lists);
    dst.clear();
    dst.resize(queries_length);
    for (int32_t i = 0; i < queries_length; ++i)
        dst[i].assign(ids.begin() + (size_t)i * max_candidates, ids.begin() + (size_t)i * max_candidates + counts[i]);
    return database_cells;
}

// batched_database_alignment.cpp -- MI355X replacement for the reference's alignment wrapper.
//
// Defines the function `alignDatabase(...)` that sift4g's main.cpp:218 calls, with the signature the reference declares in
// /root/reference/sift4g/src/database_alignment.hpp:18-23, so a maintainer swaps ONE source file: compile the reference's
// sources without its database_alignment.cpp and with this file (Makefile target `dropin-batched`).
//
// What it keeps (reference database_alignment.cpp:21-127): the database is streamed in parts (64 MB here, 1 GB there); a query is aligned
// only with its prefilter candidates (`indices[i]`, sorted ids, the consumed prefix is removed); per-chunk results are
// merged with dbAlignmentsMerge; chains that appear in no alignment are deleted and their slots set to NULL; the
// surviving `database` array is handed back.  Same banner and progress lines on stderr.
//
// What it changes: the reference builds a filtered database and calls the scorer once PER QUERY and chunk
// (:61-95: createFilteredDatabase -> chainDatabaseCreate -> alignDatabase -> chainDatabaseDelete).  Here the candidates of
// ALL queries inside a chunk form one ChainDatabase (each chain uploaded once, however many queries list it) and one
// shotgunDatabaseCandidates call scores every (query, candidate) pair and traces every survivor back: two device
// submissions per chunk instead of two per query -- and that call runs on a helper thread while the main thread reads the
// next part of the file, so a database of many parts costs the longer of reading and aligning, not their sum.  The alignments
// are identical (tests/test_gpu_swsharp.py compares the two builds' outputs byte for byte).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <string>
#include <vector>

#include "utils.hpp"                   // the reference's own: databaseLog
#include "database_alignment.hpp"      // the reference's own: the declaration this file implements
#include "flat_database.hpp"           // the database as the device prefilter's pass left it, if it did (dropin-devsearch)

namespace {

// 64 MB of sequence per part.  The reference reads 1 GB at a time (database_alignment.cpp:12); nothing in the results depends
// on where the parts end (the tests compare the outputs at several sizes byte for byte), and with the library working on one
// part while the next is read, parts a few times smaller than the database are what lets the two overlap: 1000 queries x
// 570 000 targets (200 MB) take 0.38-0.43 s in four parts against 0.83-0.91 s in one on the same box.  S4G_CHUNK_BYTES
// overrides it (the tests also use it to run the several-parts logic -- consumed candidate prefixes, merge, chain deletion
// -- on small inputs).
size_t chunk_bytes() {
    const char* v = getenv("S4G_CHUNK_BYTES");
    const long long n = v ? atoll(v) : 0;
    return n > 0 ? (size_t)n : (size_t)64000000;
}

void evalue_callback(double* values, int* scores, Chain* query, Chain** database, int databaseLen, int* cards, int cardsLen,
                     void* param) {
    eValues(values, scores, query, database, databaseLen, cards, cardsLen, static_cast<EValueParams*>(param));
}

// The candidates of one chunk: the union of all queries' ids in [begin, end), and every query's list as positions in it.
// (A membership table over the chunk's id range and its running count give the union and every position in O(pairs + range);
// sorting the ~10^6 ids of a chunk and searching each one took a third of the stage's host time.)
struct ChunkCandidates {
    std::vector<uint32_t> ids;               // sorted, unique global ids
    std::vector<std::vector<int>> lists;     // per query: positions into `ids`

    ChunkCandidates(std::vector<std::vector<uint32_t>>& indices, uint32_t begin, uint32_t end) : lists(indices.size()) {
        std::vector<size_t> taken(indices.size(), 0);
        std::vector<uint32_t> rank((size_t)(end - begin) + 1, 0);      // first: membership, then: position in `ids`
        for (size_t q = 0; q < indices.size(); ++q) {
            const std::vector<uint32_t>& v = indices[q];
            size_t n = 0;
            while (n < v.size() && v[n] < end) { if (v[n] >= begin) rank[v[n] - begin] = 1; ++n; }
            taken[q] = n;
        }
        uint32_t seen = 0;
        for (uint32_t i = 0; i < end - begin; ++i) {
            const uint32_t member = rank[i];
            rank[i] = seen;
            if (member) { ids.push_back(begin + i); ++seen; }
        }
        for (size_t q = 0; q < indices.size(); ++q) {
            std::vector<uint32_t>& v = indices[q];
            lists[q].reserve(taken[q]);
            for (size_t k = 0; k < taken[q]; ++k)
                if (v[k] >= begin) lists[q].push_back((int)rank[v[k] - begin]);
            v.erase(v.begin(), v.begin() + taken[q]);           // consumed, like the reference's createFilteredDatabase
        }
    }
};

// One part of the database on its way through the library: the union of its candidates, every query's list as positions in
// it, and what the library found.  The library call runs on a helper thread while the main thread reads the next part.
struct PartJob {
    int begin = 0, end = 0;                    // the part's range of database indices
    ChunkCandidates cand;
    std::vector<Chain*> pool;
    DbAlignment*** found = nullptr;
    int* found_lengths = nullptr;
    std::future<void> running;
    PartJob(std::vector<std::vector<uint32_t>>& indices, Chain** database, int begin_, int end_)
        : begin(begin_), end(end_), cand(indices, (uint32_t)begin_, (uint32_t)end_), pool(cand.ids.size()) {
        for (size_t k = 0; k < pool.size(); ++k) pool[k] = database[cand.ids[k]];
    }
};

// The stage when the prefilter's pass left the database behind as flat parts (flat_database.hpp): nothing is read again and no
// chain is built for a record unless it ends up in an alignment.  Per part: every query's candidates inside the part as positions
// in it (the consumed prefix removed, as createFilteredDatabase does), ONE flat ChainDatabase over the whole part
// (chainDatabaseCreateFlat: the codes go to the device as they are), one shotgunDatabaseCandidates call, merge in part order.
// The `database` array handed back holds the chains of the reported targets and NULL everywhere else -- what the reference's
// array looks like after it deleted the chains without an alignment (database_alignment.cpp:106-113).
void align_flat_database(s4g_dropin::FlatDatabase& flat, DbAlignment**** alignments, int** alignments_lengths, Chain*** database_out,
                         int32_t* database_length_out, Chain** queries, int32_t queries_length, std::vector<std::vector<uint32_t>>& indices,
                         int32_t algorithm, EValueParams* evalue_params, double max_evalue, uint32_t max_alignments, Scorer* scorer,
                         int32_t* cards, int32_t cards_length) {
    size_t total = 0;
    for (const s4g_dropin::FlatPart& p : flat.parts) total += (size_t)p.n;
    Chain** database = (Chain**)calloc(total ? total : 1, sizeof(Chain*));           // malloc domain: main.cpp frees it with deleteFastaChains
    double t_lists = 0, t_lib = 0, t_rest = 0;
    auto lap = [](std::chrono::steady_clock::time_point& t) {
        const auto now = std::chrono::steady_clock::now();
        const double s = std::chrono::duration<double>(now - t).count();
        t = now;
        return s;
    };
    uint32_t part_no = 1;
    for (const s4g_dropin::FlatPart& p : flat.parts) {
        auto clock = std::chrono::steady_clock::now();
        const float part_gb = (p.n ? p.off[p.n] : 0) / 1e9f;
        databaseLog(part_no, part_gb, 0);
        const uint32_t begin = p.first_id, end = p.first_id + (uint32_t)p.n;
        std::vector<std::vector<int>> lists(queries_length);
        size_t pairs = 0;
        for (int32_t q = 0; q < queries_length; ++q) {
            std::vector<uint32_t>& v = indices[q];
            size_t n = 0;
            while (n < v.size() && v[n] < end) { if (v[n] >= begin) lists[q].push_back((int)(v[n] - begin)); ++n; }
            v.erase(v.begin(), v.begin() + n);
            pairs += lists[q].size();
        }
        t_lists += lap(clock);
        DbAlignment*** found = nullptr;
        int* found_lengths = nullptr;
        if (pairs == 0) {
            found = (DbAlignment***)calloc(queries_length ? queries_length : 1, sizeof(DbAlignment**));
            found_lengths = (int*)calloc(queries_length ? queries_length : 1, sizeof(int));
        } else {
            databaseLog(part_no, part_gb, 50);
            ChainDatabase* part_db = chainDatabaseCreateFlat(p.codes, p.off, p.names, p.name_off, p.n, cards, cards_length);
            std::vector<int*> list_ptrs(queries_length);
            std::vector<int> list_lens(queries_length);
            for (int32_t q = 0; q < queries_length; ++q) { list_ptrs[q] = lists[q].empty() ? nullptr : lists[q].data(); list_lens[q] = (int)lists[q].size(); }
            shotgunDatabaseCandidates(&found, &found_lengths, algorithm, queries, queries_length, part_db, scorer, (int)max_alignments,
                                      evalue_callback, evalue_params, max_evalue, list_ptrs.data(), list_lens.data(), cards, cards_length, nullptr);
            // the chains the library made for the reported targets are ours now
            for (int32_t q = 0; q < queries_length; ++q)
                for (int k = 0; k < found_lengths[q]; ++k) database[begin + (uint32_t)dbAlignmentGetTargetIdx(found[q][k])] = dbAlignmentGetTarget(found[q][k]);
            chainDatabaseDelete(part_db);
        }
        t_lib += lap(clock);
        if (*alignments == nullptr) {
            *alignments = found;
            *alignments_lengths = found_lengths;
        } else {
            dbAlignmentsMerge(*alignments, *alignments_lengths, found, found_lengths, queries_length, (int)max_alignments);
            deleteShotgunDatabase(found, found_lengths, queries_length);
        }
        databaseLog(part_no, part_gb, 100);
        ++part_no;
        t_rest += lap(clock);
    }
    fprintf(stderr, "\n\n");
    if (*alignments == nullptr) {            // a database without records
        *alignments = (DbAlignment***)calloc(queries_length ? queries_length : 1, sizeof(DbAlignment**));
        *alignments_lengths = (int*)calloc(queries_length ? queries_length : 1, sizeof(int));
    }
    if (getenv("S4G_PROFILE"))
        fprintf(stderr, "[s4g profile] aligning stage (flat parts of the prefilter's pass: the file is not read again): candidate lists %.3f s, library "
                        "(database + scores + traceback + chains of the hits) %.3f s, merge %.3f s\n", t_lists, t_lib, t_rest);
    flat.clear();
    *database_out = database;
    *database_length_out = (int32_t)total;
}

}  // namespace

void alignDatabase(DbAlignment**** alignments, int** alignments_lengths, Chain*** database_out, int32_t* database_length_out,
                   const std::string& database_path, Chain** queries, int32_t queries_length,
                   std::vector<std::vector<uint32_t>>& indices, int32_t algorithm, EValueParams* evalue_params, double max_evalue,
                   uint32_t max_alignments, Scorer* scorer, int32_t* cards, int32_t cards_length) {
    fprintf(stderr, "** Aligning queries with candidate sequences **\n");

    {   // the database as the device prefilter's pass left it (dropin-devsearch): one pass over the file
        s4g_dropin::FlatDatabase& flat = s4g_dropin::flat_database();
        if (flat.complete && flat.path == database_path && !getenv("S4G_READ_DATABASE_TWICE")) {
            align_flat_database(flat, alignments, alignments_lengths, database_out, database_length_out, queries, queries_length, indices,
                                algorithm, evalue_params, max_evalue, max_alignments, scorer, cards, cards_length);
            return;
        }
        flat.clear();
    }

    Chain** database = nullptr;
    int database_length = 0;
    FILE* handle = nullptr;
    int serialized = 0;
    readFastaChainsPartInit(&database, &database_length, &handle, &serialized, database_path.c_str());

    const size_t kChunkBytes = chunk_bytes();
    const float part_gb = kChunkBytes / 1e9f;
    double t_read = 0, t_cand = 0, t_wait = 0, t_rest = 0;           // S4G_PROFILE=1: where the stage's time goes
    auto lap = [](std::chrono::steady_clock::time_point& t) {
        const auto now = std::chrono::steady_clock::now();
        const double s = std::chrono::duration<double>(now - t).count();
        t = now;
        return s;
    };

    // what remains to do with a part once the library is through with it: merge (in part order, database_alignment.cpp:97-104)
    // and delete the chains that carry no alignment (:106-113)
    auto finish = [&](PartJob& job, std::chrono::steady_clock::time_point& clock) {
        job.running.get();
        t_wait += lap(clock);
        std::vector<char> in_use(job.pool.size(), 0);
        for (int32_t q = 0; q < queries_length; ++q)
            for (int k = 0; k < job.found_lengths[q]; ++k) in_use[dbAlignmentGetTargetIdx(job.found[q][k])] = 1;
        std::vector<char> keep_chain((size_t)(job.end - job.begin), 0);
        for (size_t k = 0; k < job.pool.size(); ++k)
            if (in_use[k]) keep_chain[job.cand.ids[k] - (uint32_t)job.begin] = 1;
        if (*alignments == nullptr) {
            *alignments = job.found;
            *alignments_lengths = job.found_lengths;
        } else {
            dbAlignmentsMerge(*alignments, *alignments_lengths, job.found, job.found_lengths, queries_length, (int)max_alignments);
            deleteShotgunDatabase(job.found, job.found_lengths, queries_length);
        }
        for (int i = job.begin; i < job.end; ++i)
            if (!keep_chain[i - job.begin] && database[i] != nullptr) {
                chainDelete(database[i]);
                database[i] = nullptr;
            }
        t_rest += lap(clock);
    };

    int chunk_begin = 0;
    PartJob* in_flight = nullptr;                 // the part the library is working on while the next one is being read
    for (uint32_t part = 1;; ++part) {
        auto clock = std::chrono::steady_clock::now();
        // (readFastaChainsPart may move the ARRAY of chain pointers; the job in flight holds copies of the pointers it needs,
        // and the chains themselves do not move)
        const int more = readFastaChainsPart(&database, &database_length, handle, serialized, kChunkBytes);
        t_read += lap(clock);
        databaseLog(part, part_gb, 0);
        if (in_flight) { finish(*in_flight, clock); delete in_flight; in_flight = nullptr; }

        PartJob* job = new PartJob(indices, database, chunk_begin, database_length);
        t_cand += lap(clock);
        if (job->pool.empty()) {     // nothing to align in this part: empty lists, malloc domain like the library's
            job->found = (DbAlignment***)calloc(queries_length ? queries_length : 1, sizeof(DbAlignment**));
            job->found_lengths = (int*)calloc(queries_length ? queries_length : 1, sizeof(int));
            job->running = std::async(std::launch::deferred, []() {});
        } else {
            databaseLog(part, part_gb, 50);
            job->running = std::async(std::launch::async, [=]() {
                ChainDatabase* chunk_db = chainDatabaseCreate(job->pool.data(), 0, (int)job->pool.size(), cards, cards_length);
                std::vector<int*> list_ptrs(queries_length);
                std::vector<int> list_lens(queries_length);
                for (int32_t q = 0; q < queries_length; ++q) {
                    list_ptrs[q] = job->cand.lists[q].empty() ? nullptr : job->cand.lists[q].data();
                    list_lens[q] = (int)job->cand.lists[q].size();
                }
                shotgunDatabaseCandidates(&job->found, &job->found_lengths, algorithm, queries, queries_length, chunk_db, scorer,
                                          (int)max_alignments, evalue_callback, evalue_params, max_evalue, list_ptrs.data(),
                                          list_lens.data(), cards, cards_length, nullptr);
                chainDatabaseDelete(chunk_db);
            });
        }
        in_flight = job;
        databaseLog(part, part_gb, 100);
        if (!more) break;
        chunk_begin = database_length;
    }
    {
        auto clock = std::chrono::steady_clock::now();
        finish(*in_flight, clock);
        delete in_flight;
    }
    fprintf(stderr, "\n\n");
    if (getenv("S4G_PROFILE"))
        fprintf(stderr, "[s4g profile] aligning stage: reading FASTA into chains %.3f s, candidate union %.3f s, waiting for the library "
                        "(database + scores + traceback; it works while the next part is read) %.3f s, merge + deleting unused chains %.3f s\n",
                t_read, t_cand, t_wait, t_rest);
    fclose(handle);
    *database_out = database;
    *database_length_out = database_length;
}

// fullscan_database_alignment.cpp -- the alignment wrapper of the full-scan build (SURVEY.md 8a row a1 + 8f-3).
//
// Defines `alignDatabase(...)` as /root/reference/sift4g/src/database_alignment.hpp:18-23 declares it and replaces the
// reference's database_alignment.cpp in `make dropin-fullscan` (together with count_database_search.cpp, which replaces the
// k-mer prefilter).  Every query is aligned with EVERY sequence of every chunk: `indices` is ignored.
//
// Kept from the reference (database_alignment.cpp:21-127): ~1 GB chunks, dbAlignmentsMerge across chunks (:97-104), chains
// without an alignment deleted and their slots NULLed (:106-111), the surviving `database` array handed back, banner and
// progress lines.  Changed: one ChainDatabase per chunk (uploaded once, packed on the device) and ONE
// shotgunDatabaseEValues call for all queries -- conservative integer score cut and exact top-k' on the device, exact
// double-precision e-values on the survivors, traceback of the survivors on the device -- instead of
// createFilteredDatabase + chainDatabaseCreate + alignDatabase per query (:61-95).
//
// Results: every hit the prefiltered builds report is reported here with the same score, e-value and path, plus the
// homologs the k-mer prefilter missed (tests/test_gpu_swsharp.py::test_fullscan_build_is_a_superset_of_the_prefiltered_build);
// lists differ only where a query has more significant hits than max_alignments.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "utils.hpp"
#include "database_alignment.hpp"

namespace {
size_t chunk_bytes() {          // S4G_CHUNK_BYTES: the tests run the several-parts logic on small inputs
    const char* v = getenv("S4G_CHUNK_BYTES");
    const long long n = v ? atoll(v) : 0;
    return n > 0 ? (size_t)n : (size_t)1000000000;
}
}  // namespace

void alignDatabase(DbAlignment**** alignments, int** alignments_lengths, Chain*** database_out, int32_t* database_length_out,
                   const std::string& database_path, Chain** queries, int32_t queries_length,
                   std::vector<std::vector<uint32_t>>& indices, int32_t algorithm, EValueParams* evalue_params, double max_evalue,
                   uint32_t max_alignments, Scorer* scorer, int32_t* cards, int32_t cards_length) {
    fprintf(stderr, "** Aligning queries with candidate sequences **\n");
    for (auto& v : indices) v.clear();          // nothing to consume: there is no candidate stage

    Chain** database = nullptr;
    int database_length = 0;
    FILE* handle = nullptr;
    int serialized = 0;
    readFastaChainsPartInit(&database, &database_length, &handle, &serialized, database_path.c_str());

    const size_t kChunkBytes = chunk_bytes();
    const float part_gb = kChunkBytes / 1e9f;
    int chunk_begin = 0;
    for (uint32_t part = 1;; ++part) {
        const int more = readFastaChainsPart(&database, &database_length, handle, serialized, kChunkBytes);
        databaseLog(part, part_gb, 0);
        const int n = database_length - chunk_begin;

        DbAlignment*** found = nullptr;
        int* found_lengths = nullptr;
        if (n == 0) {
            found = (DbAlignment***)calloc(queries_length ? queries_length : 1, sizeof(DbAlignment**));
            found_lengths = (int*)calloc(queries_length ? queries_length : 1, sizeof(int));
        } else {
            ChainDatabase* chunk_db = chainDatabaseCreate(database, chunk_begin, n, cards, cards_length);
            databaseLog(part, part_gb, 50);
            shotgunDatabaseEValues(&found, &found_lengths, algorithm, queries, queries_length, chunk_db, scorer,
                                   (int)max_alignments, evalue_params, max_evalue, cards, cards_length, nullptr);
            chainDatabaseDelete(chunk_db);
        }

        std::vector<char> keep_chain((size_t)n, 0);             // DbAlignment borrows its target chain
        for (int32_t q = 0; q < queries_length; ++q)
            for (int k = 0; k < found_lengths[q]; ++k) keep_chain[dbAlignmentGetTargetIdx(found[q][k])] = 1;

        if (*alignments == nullptr) {
            *alignments = found;
            *alignments_lengths = found_lengths;
        } else {
            dbAlignmentsMerge(*alignments, *alignments_lengths, found, found_lengths, queries_length, (int)max_alignments);
            deleteShotgunDatabase(found, found_lengths, queries_length);
        }
        for (int i = 0; i < n; ++i)
            if (!keep_chain[i] && database[chunk_begin + i] != nullptr) {
                chainDelete(database[chunk_begin + i]);
                database[chunk_begin + i] = nullptr;
            }

        databaseLog(part, part_gb, 100);
        if (!more) break;
        chunk_begin = database_length;
    }
    fprintf(stderr, "\n\n");
    fclose(handle);
    *database_out = database;
    *database_length_out = database_length;
}

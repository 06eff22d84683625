// count_database_search.cpp -- the prefilter stage of the full-scan build: no k-mer index, no candidates.
//
// Defines `searchDatabase(...)` with the signature the reference declares in
// /root/reference/sift4g/src/database_search.hpp:17-19 and replaces the reference's database_search.cpp (k-mer hash of the
// queries, per-thread scan of the database, longest increasing subsequence of hit positions, 5000 best candidates per query:
// database_search.cpp:66-253, hash.cpp:21-90) in `make dropin-fullscan`.  On an MI355X scoring EVERY target is cheaper than
// choosing which ones to score (1.2 ms per query and 100 000 targets), so this stage only does the one
// thing the rest of the pipeline still needs from it: the total number of database residues, which main.cpp:204,210 hands
// to createEValueParams.  `dst` comes back with one EMPTY list per query; fullscan_database_alignment.cpp ignores it.
#include <cstdio>
#include <string>
#include <vector>

#include "utils.hpp"              // the reference's own: databaseLog
#include "database_search.hpp"    // the reference's own: the declaration this file implements

uint64_t searchDatabase(std::vector<std::vector<uint32_t>>& dst, const std::string& database_path, Chain** queries,
                        int32_t queries_length, uint32_t kmer_length, uint32_t max_candidates, uint32_t num_threads) {
    (void)queries; (void)kmer_length; (void)max_candidates; (void)num_threads;
    fprintf(stderr, "** Searching database for candidate sequences **\n");
    Chain** database = nullptr;
    int database_length = 0;
    FILE* handle = nullptr;
    int serialized = 0;
    readFastaChainsPartInit(&database, &database_length, &handle, &serialized, database_path.c_str());
    const size_t kChunk = 250000000;          // the reference's own part size for this stage (database_search.cpp:17)
    uint64_t cells = 0;
    int start = 0;
    for (uint32_t part = 1;; ++part) {
        const int more = readFastaChainsPart(&database, &database_length, handle, serialized, kChunk);
        databaseLog(part, kChunk / 1e9f, 0);
        for (int i = start; i < database_length; ++i) {
            cells += (uint64_t)chainGetLength(database[i]);
            chainDelete(database[i]);
            database[i] = nullptr;
        }
        databaseLog(part, kChunk / 1e9f, 100);
        start = database_length;
        if (!more) break;
    }
    fprintf(stderr, "\n\n");
    fclose(handle);
    deleteFastaChains(database, database_length);
    dst.assign((size_t)queries_length, std::vector<uint32_t>());
    return cells;
}

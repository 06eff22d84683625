// count_database_search.cpp -- the prefilter stage of the full-scan build: no k-mer index, no candidates.
//
// Defines `searchDatabase(...)` with the signature the reference declares in
// /root/reference/sift4g/src/database_search.hpp:17-19 and replaces the reference's database_search.cpp (k-mer hash of the
// queries, per-thread scan of the database, longest increasing subsequence of hit positions, 5000 best candidates per query:
// database_search.cpp:66-253, hash.cpp:21-90) in `make dropin-fullscan`.  For inputs of modest size scoring EVERY target on an MI355X is cheaper than
// choosing which ones to score on the CPU (1.2 ms per query and 100 000 targets; for large ones the prefilter on
// the device, device_database_search.cpp, is the better trade), so this stage only does the one
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
    long long records = 0, residues = 0;
    databaseLog(1, 0.25f, 0);
    fastaResidueCount(database_path.c_str(), &records, &residues);     // one pass over the file -- or the header of its serialized cache
    databaseLog(1, 0.25f, 100);
    fprintf(stderr, "\n\n");
    const uint64_t cells = (uint64_t)residues;
    dst.assign((size_t)queries_length, std::vector<uint32_t>());
    return cells;
}

// ref_prefilter_dump.cpp -- prints what the REFERENCE's own prefilter returns, for the golden vectors of tests/golden/prefilter/.
//
// `make dropin-prefilter-dump` compiles /root/reference/sift4g/src/{database_search,hash,utils}.cpp where they lie -- the
// reference's text, unmodified -- together with this main, against include/swsharp/swsharp.h and this repo's libswsharp (its
// Chain, FASTA readers and thread pool: the same arrangement as `make dropin`).  It is therefore the reference's ALGORITHM for
// this stage running on this repo's Chain substrate, not an oracle/_ref build of the reference as a whole (swsharp is absent).
//
//   ref_prefilter_dump <queries.fasta> <database.fasta> <kmer_length> <max_candidates> <threads>
// stdout: "cells <n>" then one line per query: "<query index> <count> <id> <id> ..." (ids ascending, as searchDatabase returns).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "swsharp/swsharp.h"
#include "database_search.hpp"

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage: %s queries.fasta database.fasta kmer_length max_candidates threads\n", argv[0]); return 2; }
    const int threads = atoi(argv[5]);
    threadPoolInitialize(threads);
    Chain** queries = nullptr; int n = 0;
    readFastaChains(&queries, &n, argv[1]);
    std::vector<std::vector<uint32_t>> indices;
    const uint64_t cells = searchDatabase(indices, argv[2], queries, n, (uint32_t)atoi(argv[3]), (uint32_t)atoi(argv[4]), (uint32_t)threads);
    printf("cells %llu\n", (unsigned long long)cells);
    for (int q = 0; q < n; ++q) {
        printf("%d %zu", q, indices[q].size());
        for (uint32_t id : indices[q]) printf(" %u", id);
        printf("\n");
    }
    deleteFastaChains(queries, n);
    threadPoolTerminate();
    return 0;
}

// oracle/pin_dump.cpp -- TEST INFRASTRUCTURE (the pinning recipe, oracle/Makefile target `_ref`; never on the product path).
//
// A program written against the swsharp C API -- only calls the reference itself makes (main.cpp:192,208,210;
// database_alignment.cpp:80-92,129-134; select_alignments.cpp:145-152,246-283) plus dbAlignmentGetScore / dbAlignmentGetValue --
// that prints, for every query of a FASTA file against every sequence of a database FASTA file, one line per returned alignment:
//
//     query <TAB> target <TAB> score <TAB> e-value (%.17g) <TAB> query start <TAB> target start <TAB> move string (D / L / U)
//
// in the order alignDatabase returned them (so the selection order is part of the dump).  It is compiled TWICE from this one source:
//
//   oracle/_ref/pin_dump_ref   against an upstream mkorpar/swsharp checkout built for the CPU (make -C oracle _ref SWSHARP_SRC=...)
//   build/pin_dump_ours        against include/swsharp/*.h + sift4g_amd/lib/libswsharp.so (make pin-dump)
//
// and oracle/pin_against_swsharp.py diffs the two dumps column by column.  That diff is what would turn "parity unpinned" into a pinned
// oracle: scores (step 1 of database_alignment.cpp:83-86), e-values (which Gumbel row a 10/1 scorer looks up), order, and tracebacks.
// The upstream checkout is absent from this image (/root/reference/.gitmodules:1-3, vendor/swsharp empty, no network), so only the
// second binary can be built here; tests/test_pin_swsharp.py checks that one against the golden fixture and skips the diff visibly.
//
// usage: pin_dump <query.fasta> <database.fasta> [matrix=BLOSUM_62] [open=10] [extend=1] [max_alignments=400] [max_evalue=1e9]
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "swsharp/evalue.h"
#include "swsharp/swsharp.h"

// the reference's callback, database_alignment.cpp:129-134
static void value_function(double* values, int* scores, Chain* query, Chain** database, int databaseLen, int* cards, int cardsLen, void* param) {
    eValues(values, scores, query, database, databaseLen, cards, cardsLen, (EValueParams*)param);
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <query.fasta> <database.fasta> [matrix] [open] [extend] [max_alignments] [max_evalue]\n", argv[0]);
        return 2;
    }
    char matrix[64];
    snprintf(matrix, sizeof matrix, "%s", argc > 3 ? argv[3] : "BLOSUM_62");
    const int gap_open = argc > 4 ? atoi(argv[4]) : 10, gap_extend = argc > 5 ? atoi(argv[5]) : 1;
    const int max_alignments = argc > 6 ? atoi(argv[6]) : 400;
    const double max_evalue = argc > 7 ? atof(argv[7]) : 1e9;

    Chain** queries = NULL; int queries_len = 0;
    Chain** database = NULL; int database_len = 0;
    readFastaChains(&queries, &queries_len, argv[1]);
    readFastaChains(&database, &database_len, argv[2]);
    long long cells = 0;                                   // database_search.cpp:84,126,182: the database's residues
    for (int i = 0; i < database_len; ++i) cells += chainGetLength(database[i]);

    Scorer* scorer = NULL;
    scorerCreateMatrix(&scorer, matrix, gap_open, gap_extend);              // main.cpp:208
    EValueParams* params = createEValueParams(cells, scorer);               // main.cpp:210
    ChainDatabase* cdb = chainDatabaseCreate(database, 0, database_len, NULL, 0);      // database_alignment.cpp:80-81 (no cards: CPU)

    printf("# pin_dump matrix=%s open=%d extend=%d queries=%d targets=%d residues=%lld max_alignments=%d max_evalue=%g\n", matrix, gap_open, gap_extend,
           queries_len, database_len, cells, max_alignments, max_evalue);
    for (int q = 0; q < queries_len; ++q) {
        DbAlignment** out = NULL; int out_len = 0;
        alignDatabase(&out, &out_len, SW_ALIGN, queries[q], cdb, scorer, max_alignments, value_function, (void*)params, max_evalue,
                      NULL, 0, NULL, 0, NULL);                             // database_alignment.cpp:83-86
        for (int k = 0; k < out_len; ++k) {
            DbAlignment* a = out[k];
            const int t = dbAlignmentGetTargetIdx(a);                       // :89
            Alignment* al = dbAlignmentToAlignment(a);                      // select_alignments.cpp:152
            printf("%s\t%s\t%d\t%.17g\t%d\t%d\t", chainGetName(queries[q]), chainGetName(database[t]), dbAlignmentGetScore(a), dbAlignmentGetValue(a),
                   alignmentGetQueryStart(al), alignmentGetTargetStart(al));
            const int n = alignmentGetPathLen(al);
            for (int i = 0; i < n; ++i) {                                   // select_alignments.cpp:246-283
                const char m = alignmentGetMove(al, i);
                putchar(m == MOVE_DIAG ? 'D' : m == MOVE_LEFT ? 'L' : m == MOVE_UP ? 'U' : '?');
            }
            putchar('\n');
            alignmentDelete(al);
            dbAlignmentDelete(a);
        }
        free(out);                                                          // the malloc / free domain of database_alignment.cpp:55-57
    }
    chainDatabaseDelete(cdb);
    deleteEValueParams(params);
    scorerDelete(scorer);
    deleteFastaChains(database, database_len);
    deleteFastaChains(queries, queries_len);
    return 0;
}

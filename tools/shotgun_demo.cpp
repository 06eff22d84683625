// shotgun_demo.cpp -- a host program written against the swsharp-compatible API (include/swsharp/*.h), the way
// sift4g's wrapper would use the batched call recommended in INTEGRATION.md section 2:
//   read query + database FASTA -> chainDatabaseCreate (DB resident in HBM) -> shotgunDatabase (scan all queries,
//   e-values via the caller's callback, top max_alignments, device traceback) -> print timing and the best hits.
// g++ -O2 -Iinclude tools/shotgun_demo.cpp -Lsift4g_amd/lib -lswsharp -ls4g_hip -lpthread -o build/shotgun_demo
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "swsharp/evalue.h"
#include "swsharp/swsharp.h"

static void valueFunction(double* values, int* scores, Chain* query, Chain** database, int databaseLen, int* cards,
                          int cardsLen, void* param) {      // same shape as database_alignment.cpp:129-134
    eValues(values, scores, query, database, databaseLen, cards, cardsLen, (EValueParams*)param);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <query.fasta> <database.fasta> [max_alignments=400] [max_evalue=1e-4] [threads=8] [evalue_path=0]\n", argv[0]); return 2; }
    const int max_alignments = argc > 3 ? atoi(argv[3]) : 400;
    const double max_evalue = argc > 4 ? atof(argv[4]) : 1e-4;
    threadPoolInitialize(argc > 5 ? atoi(argv[5]) : 8);
    Chain **queries = NULL, **database = NULL;
    int nq = 0, nd = 0;
    double t0 = now();
    readFastaChains(&queries, &nq, argv[1]);
    readFastaChains(&database, &nd, argv[2]);
    long long cells = 0, qres = 0;
    for (int i = 0; i < nd; ++i) cells += chainGetLength(database[i]);
    for (int i = 0; i < nq; ++i) qres += chainGetLength(queries[i]);
    double t1 = now();
    Scorer* scorer = NULL;
    scorerCreateMatrix(&scorer, (char*)"BLOSUM_62", 10, 1);
    EValueParams* ev = createEValueParams(cells, scorer);
    int card = 0;
    ChainDatabase* cdb = chainDatabaseCreate(database, 0, nd, &card, 1);
    double t2 = now();
    DbAlignment*** hits = NULL; int* lens = NULL;
    const bool evalue_path = argc > 6 && atoi(argv[6]) != 0;     // 7th argument 1: e-value driven selection (nothing Q x N leaves the device)
    if (evalue_path)
        shotgunDatabaseEValues(&hits, &lens, SW_ALIGN, queries, nq, cdb, scorer, max_alignments, ev, max_evalue, &card, 1, NULL);
    else
        shotgunDatabase(&hits, &lens, SW_ALIGN, queries, nq, cdb, scorer, max_alignments, valueFunction, ev, max_evalue, NULL, 0, &card, 1, NULL);
    double t3 = now();
    long long total = 0, path = 0;
    for (int i = 0; i < nq; ++i) { total += lens[i]; for (int j = 0; j < lens[i]; ++j) path += dbAlignmentGetPathLen(hits[i][j]); }
    fprintf(stderr, "queries %d (%lld aa), database %d (%lld aa): read %.2f s, database to HBM %.2f s, %s %.2f s "
            "(%.1f GCUPS end to end incl. e-values, selection, traceback of %lld alignments, %lld path moves)\n",
            nq, qres, nd, cells, t1 - t0, t2 - t1, evalue_path ? "shotgunDatabaseEValues" : "shotgunDatabase", t3 - t2,
            (double)qres * (double)cells / (t3 - t2) / 1e9, total, path);
    for (int i = 0; i < nq && i < 3; ++i)
        for (int j = 0; j < lens[i] && j < 3; ++j)
            printf("%s\t%s\tscore=%d\tevalue=%.3g\tq=[%d,%d]\tt=[%d,%d]\tpath=%d\n", chainGetName(queries[i]),
                   chainGetName(dbAlignmentGetTarget(hits[i][j])), dbAlignmentGetScore(hits[i][j]), dbAlignmentGetValue(hits[i][j]),
                   dbAlignmentGetQueryStart(hits[i][j]), dbAlignmentGetQueryEnd(hits[i][j]), dbAlignmentGetTargetStart(hits[i][j]),
                   dbAlignmentGetTargetEnd(hits[i][j]), dbAlignmentGetPathLen(hits[i][j]));
    deleteShotgunDatabase(hits, lens, nq);
    chainDatabaseDelete(cdb);
    deleteEValueParams(ev);
    scorerDelete(scorer);
    deleteFastaChains(queries, nq);
    deleteFastaChains(database, nd);
    threadPoolTerminate();
    return 0;
}

/*
 * swsharp/swsharp.h -- the C API sift4g is written against, re-implemented for MI355X.
 *
 * sift4g includes "swsharp/swsharp.h" and "swsharp/evalue.h"
 * (/root/reference/sift4g/src/main.cpp:12-13, database_alignment.hpp:15-16) from the
 * third-party module mkorpar/swsharp, which is an EMPTY submodule in the reference tree.
 * Every declaration below is reconstructed from sift4g's own call sites (cited per symbol);
 * the implementation (libswsharp.so, sift4g_amd/csrc/swsharp/) is new C++ whose scoring step
 * calls the HIP library through include/s4g_hip.h.  There is no CPU scoring path: alignDatabase
 * aborts with a message when no MI355X is usable (sift4g's own error style: utils.hpp:13-19).
 */
#ifndef SWSHARP_SWSHARP_H
#define SWSHARP_SWSHARP_H

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- alignment types (main.cpp:51-56, 85) ---- */
#define SW_ALIGN 0   /* Smith-Waterman local: the only mode sift4g's pipeline runs */
#define NW_ALIGN 1
#define HW_ALIGN 2
#define OV_ALIGN 3

/* ---- database output formats (main.cpp:44-49, 83) ---- */
#define SW_OUT_DB_BLASTM0 0
#define SW_OUT_DB_BLASTM8 1
#define SW_OUT_DB_BLASTM9 2
#define SW_OUT_DB_LIGHT   3

/* ---- traceback moves (select_alignments.cpp:266-292) ---- */
#define MOVE_STOP 0
#define MOVE_DIAG 1   /* query residue aligned to target residue */
#define MOVE_LEFT 2   /* gap in query: consumes a target residue  */
#define MOVE_UP   3   /* gap in target: consumes a query residue  */

typedef struct Chain Chain;
typedef struct Scorer Scorer;
typedef struct ChainDatabase ChainDatabase;
typedef struct DbAlignment DbAlignment;
typedef struct Alignment Alignment;
typedef struct ThreadPoolTask ThreadPoolTask;
typedef struct Thread Thread;            /* opaque, only ever passed as NULL (database_alignment.cpp:86) */

/* database_alignment.cpp:15-16, 129-134 */
typedef void (*ValueFunction)(double* values, int* scores, Chain* query, Chain** database,
                              int databaseLen, int* cards, int cardsLen, void* param);

/* ---- Chain (hash.cpp:25,30; select_alignments.cpp:84-101,173; sift_prediction.cpp:67-75,197) ---- */
Chain* chainCreate(char* name, int nameLen, char* string, int stringLen);   /* copies; string need not be NUL-terminated */
Chain* chainCreateView(Chain* chain, int start, int end, int reverse);      /* inclusive [start, end]; shares storage */
void chainDelete(Chain* chain);
int chainGetLength(Chain* chain);
char chainGetChar(Chain* chain, int index);
char chainGetCode(Chain* chain, int index);
const char* chainGetName(Chain* chain);
const char* chainGetCodes(Chain* chain);                                    /* residue codes letter - 'A', 0..25 */

/* ---- FASTA (main.cpp:192,235,245; database_search.cpp:81,96,168; database_alignment.cpp:36,47) ---- */
void readFastaChains(Chain*** chains, int* chainsLen, const char* path);
void readFastaChainsPartInit(Chain*** chains, int* chainsLen, FILE** handle, int* serialized, const char* path);
int readFastaChainsPart(Chain*** chains, int* chainsLen, FILE* handle, int serialized, const size_t maxBytes);
void deleteFastaChains(Chain** chains, int chainsLen);
/* Extension: the part readFastaChainsPart would return (database_search.cpp:93-97) as flat residue codes: no Chain objects to
 * delete again (:125-129).  codes[offsets[i] .. offsets[i + 1]) = chainGetCodes of record i; *codes and *offsets (records + 1
 * entries) are malloc'ed, the caller frees both.  Returns 1 if more parts remain.  A handle is read with one kind of reader only. */
int readFastaCodesPart(char** codes, size_t** offsets, int* recordsLen, FILE* handle, int serialized, const size_t maxBytes);
/* ... and the header text of every record beside it (names[nameOffsets[i] .. nameOffsets[i + 1]), not terminated; malloc'ed):
 * what chainDatabaseCreateFlat takes. */
int readFastaCodesPartNamed(char** codes, size_t** offsets, char** names, size_t** nameOffsets, int* recordsLen, FILE* handle,
                            int serialized, const size_t maxBytes);
/* Extension: totals of a FASTA file (records, residues as chainGetLength counts them) without keeping any chain: from the header of
 * the serialized cache when there is a valid one (S4G_FASTA_CACHE=1), else by one pass over the file.  Used by the full-scan
 * build in place of the k-mer prefilter's pass (sift4g_amd/csrc/dropin/count_database_search.cpp). */
void fastaResidueCount(const char* path, long long* records, long long* residues);

/* ---- Scorer (main.cpp:208,223; names main.cpp:304-314) ---- */
void scorerCreateMatrix(Scorer** scorer, char* name, int gapOpen, int gapExtend);
void scorerCreate(Scorer** scorer, const char* name, int* scores, char maxCode, int gapOpen, int gapExtend);
void scorerDelete(Scorer* scorer);
int scorerGetGapOpen(Scorer* scorer);
int scorerGetGapExtend(Scorer* scorer);
const char* scorerGetName(Scorer* scorer);
int scorerScore(Scorer* scorer, char a, char b);                            /* a, b = codes */
int scorerGetMaxScore(Scorer* scorer);
char scorerEncode(char c);                                                  /* letter -> code */
char scorerDecode(char c);

/* ---- cards (main.cpp:184,186): HIP device ids ---- */
void cudaGetCards(int** cards, int* cardsLen);                              /* all usable devices; malloc'd */
int cudaCheckCards(int* cards, int cardsLen);                               /* 1 = ok; passes for 0 cards */
/* Extension (not in upstream swsharp): the first card of the first successful cudaCheckCards call, 0 before any.  For stages
 * whose reference signature carries no card list (searchDatabase, database_search.hpp) yet have work for the device. */
int swsharpPreferredCard(void);

/* ---- ChainDatabase (database_alignment.cpp:80,92): targets packed and resident in HBM ---- */
ChainDatabase* chainDatabaseCreate(Chain** database, int databaseStart, int databaseLen, int* cards, int cardsLen);
void chainDatabaseDelete(ChainDatabase* chainDatabase);
/* Extension: a ChainDatabase over a part kept FLAT (readFastaCodesPartNamed's four arrays; borrowed: they must outlive the
 * database).  No Chain is built for a sequence until it appears in an alignment: the library scores straight out of the codes,
 * hands value functions placeholder chains that carry the LENGTH only (eValues needs nothing else; a callback that reads names or
 * residues of database chains must use chainDatabaseCreate), and creates the real chain -- name, letters, codes -- of every
 * target it reports.  Those chains belong to the caller (dbAlignmentGetTarget; chainDelete them), everything else goes with
 * chainDatabaseDelete.  This is what lets the drop-in read the database once: the prefilter's pass keeps the parts flat, the
 * alignment stage builds a chain for a few hundred hits per query instead of every record
 * (/root/reference/sift4g/src/database_alignment.cpp:36-47,106-113 reads everything again and deletes what carried no hit). */
ChainDatabase* chainDatabaseCreateFlat(const char* codes, const size_t* offsets, const char* names, const size_t* nameOffsets,
                                       int recordsLen, int* cards, int cardsLen);

/* ---- database alignment (database_alignment.cpp:83-86): score all, value, threshold, top maxAlignments, traceback ---- */
void alignDatabase(DbAlignment*** dbAlignments, int* dbAlignmentsLen, int type, Chain* query,
                   ChainDatabase* chainDatabase, Scorer* scorer, int maxAlignments, ValueFunction valueFunction,
                   void* valueFunctionParam, double valueThreshold, int* indexes, int indexesLen,
                   int* cards, int cardsLen, Thread* thread);

/* Batched form (new; what the wrapper's per-query loop should call on MI355X): every query against one
 * ChainDatabase in a single persistent launch.  dbAlignments / dbAlignmentsLens are malloc'd, one entry per query. */
void shotgunDatabase(DbAlignment**** dbAlignments, int** dbAlignmentsLens, int type, Chain** queries, int queriesLen,
                     ChainDatabase* chainDatabase, Scorer* scorer, int maxAlignments, ValueFunction valueFunction,
                     void* valueFunctionParam, double valueThreshold, int* indexes, int indexesLen,
                     int* cards, int cardsLen, Thread* thread);

/* shotgunDatabase with one candidate list PER QUERY (new): candidates[i][0..candidatesLens[i]) index the ChainDatabase.
 * For query i the result is what alignDatabase(query i, chainDatabaseCreate of exactly those chains, ...) returns -- same
 * alignments, values and order -- except that dbAlignmentGetTargetIdx is the index into THIS ChainDatabase.  It collapses
 * the wrapper's whole per-query loop (createFilteredDatabase + chainDatabaseCreate + alignDatabase per query,
 * database_alignment.cpp:61-95) into one scoring launch and one traceback launch per database chunk while keeping the
 * prefilter's semantics: only the candidates are scored.  A NULL or empty list gives that query no alignments. */
void shotgunDatabaseCandidates(DbAlignment**** dbAlignments, int** dbAlignmentsLens, int type, Chain** queries, int queriesLen,
                               ChainDatabase* chainDatabase, Scorer* scorer, int maxAlignments, ValueFunction valueFunction,
                               void* valueFunctionParam, double valueThreshold, int** candidates, int* candidatesLens,
                               int* cards, int cardsLen, Thread* thread);

/* shotgunDatabase for the case sift4g actually runs -- valueFunction = eValues (database_alignment.cpp:129-134): the same
 * alignments, but the library knows the value function, so only the per-query top hits ever leave the device (a
 * conservative integer score cut per query + exact streaming top-k on the GPU, exact double e-values on the host with an
 * exactness check against everything that was cut).  This is the form that scales to databases whose Q x N score matrix
 * cannot be materialised.  Declared in swsharp.h so that evalue.h keeps its reference shape. */
struct EValueParams;
void shotgunDatabaseEValues(DbAlignment**** dbAlignments, int** dbAlignmentsLens, int type, Chain** queries, int queriesLen,
                            ChainDatabase* chainDatabase, Scorer* scorer, int maxAlignments, struct EValueParams* eValueParams,
                            double maxEvalue, int* cards, int cardsLen, Thread* thread);

/* Post-scoring half of alignDatabase (steps 2-4) on scores that are already known: host logic only.
 * database[i] / scores[i], i in [0, databaseLen). */
void dbAlignmentsFromScores(DbAlignment*** dbAlignments, int* dbAlignmentsLen, int type, Chain* query,
                            Chain** database, int databaseLen, const int* scores, Scorer* scorer, int maxAlignments,
                            ValueFunction valueFunction, void* valueFunctionParam, double valueThreshold);

/* pairwise alignment with traceback (host) */
void alignPair(Alignment** alignment, int type, Chain* query, Chain* target, Scorer* scorer,
               int* cards, int cardsLen, Thread* thread);

/* ---- DbAlignment (database_alignment.cpp:89,101-103; select_alignments.cpp:145-150; main.cpp:227,234) ---- */
Chain* dbAlignmentGetQuery(DbAlignment* a);
int dbAlignmentGetQueryIdx(DbAlignment* a);
int dbAlignmentGetQueryStart(DbAlignment* a);
int dbAlignmentGetQueryEnd(DbAlignment* a);
Chain* dbAlignmentGetTarget(DbAlignment* a);
int dbAlignmentGetTargetIdx(DbAlignment* a);     /* index into the ChainDatabase passed to alignDatabase */
int dbAlignmentGetTargetStart(DbAlignment* a);
int dbAlignmentGetTargetEnd(DbAlignment* a);
int dbAlignmentGetScore(DbAlignment* a);
double dbAlignmentGetValue(DbAlignment* a);
Scorer* dbAlignmentGetScorer(DbAlignment* a);
int dbAlignmentGetPathLen(DbAlignment* a);
char dbAlignmentGetMove(DbAlignment* a, int index);
void dbAlignmentCopyPath(DbAlignment* a, char* dest);
Alignment* dbAlignmentToAlignment(DbAlignment* a);
void dbAlignmentDelete(DbAlignment* a);
void dbAlignmentsMerge(DbAlignment*** dbAlignmentsDst, int* dbAlignmentsDstLens, DbAlignment*** dbAlignmentsSrc,
                       int* dbAlignmentsSrcLens, int dbAlignmentsLen, int maxAlignments);
void deleteShotgunDatabase(DbAlignment*** dbAlignments, int* dbAlignmentsLens, int dbAlignmentsLen);
void outputShotgunDatabase(DbAlignment*** dbAlignments, int* dbAlignmentsLens, int dbAlignmentsLen, char* path, int type);

/* ---- Alignment (select_alignments.cpp:152,246-283) ---- */
Chain* alignmentGetQuery(Alignment* a);
int alignmentGetQueryStart(Alignment* a);
int alignmentGetQueryEnd(Alignment* a);
Chain* alignmentGetTarget(Alignment* a);
int alignmentGetTargetStart(Alignment* a);
int alignmentGetTargetEnd(Alignment* a);
int alignmentGetScore(Alignment* a);
Scorer* alignmentGetScorer(Alignment* a);
int alignmentGetPathLen(Alignment* a);
char alignmentGetMove(Alignment* a, int index);
void alignmentCopyPath(Alignment* a, char* dest);
void alignmentDelete(Alignment* a);

/* ---- thread pool (main.cpp:188,197,247; database_search.cpp:117-122; select_alignments.cpp:64-69) ---- */
int threadPoolInitialize(int n);
void threadPoolTerminate(void);
ThreadPoolTask* threadPoolSubmit(void* (*routine)(void*), void* param);   /* runs inline if the pool is not initialised */
void threadPoolTaskWait(ThreadPoolTask* task);                            /* NULL is a no-op */
void threadPoolTaskDelete(ThreadPoolTask* task);                          /* NULL is a no-op */

#ifdef __cplusplus
}
#endif
#endif /* SWSHARP_SWSHARP_H */

/*
 * swsharp/evalue.h -- e-value API sift4g uses (main.cpp:210,222; database_alignment.cpp:129-134).
 * Reconstructed from the call sites; statistics are the published BLAST gapped Gumbel parameters with
 * Spouge's finite-size correction (ASSUMPTION, unpinned: swsharp's own constants are in an absent submodule).
 */
#ifndef SWSHARP_EVALUE_H
#define SWSHARP_EVALUE_H

#include "swsharp.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct EValueParams EValueParams;

/* length = total number of database residues ("cells" returned by searchDatabase, main.cpp:204,210) */
EValueParams* createEValueParams(long long length, Scorer* scorer);
void deleteEValueParams(EValueParams* eValueParams);

/* values[i] = expected number of chance hits with score >= scores[i] in a database of `length` residues */
void eValues(double* values, int* scores, Chain* query, Chain** database, int databaseLen,
             int* cards, int cardsLen, EValueParams* eValueParams);

#ifdef __cplusplus
}
#endif
#endif /* SWSHARP_EVALUE_H */

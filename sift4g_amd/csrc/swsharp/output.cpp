// output.cpp -- outputShotgunDatabase (sift4g --sub-results: main.cpp:225-229; formats main.cpp:44-49).
// Layouts follow the BLAST -m 0/8/9 conventions the format names refer to; swsharp's exact text is unpinned.
#include "impl.hpp"

#include <cmath>
#include <cstring>

namespace {

struct Stats { int identities, mismatches, gap_opens, gaps, positives; };

Stats path_stats(DbAlignment* a) {
    swimpl::resolve(a);                     // (every output line of an alignment starts here)
    Stats s = {0, 0, 0, 0, 0};
    int qi = a->query_start, ti = a->target_start;
    char prev = MOVE_STOP;
    for (int k = 0; k < a->path_len; ++k) {
        const char m = a->path[k];
        if (m == MOVE_DIAG) {
            const char qc = chainGetCode(a->query, qi), tc = chainGetCode(a->target, ti);
            if (qc == tc) s.identities++; else s.mismatches++;
            if (scorerScore(a->scorer, qc, tc) > 0) s.positives++;
            ++qi; ++ti;
        } else {
            s.gaps++;
            if (m != prev) s.gap_opens++;
            if (m == MOVE_LEFT) ++ti; else ++qi;
        }
        prev = m;
    }
    return s;
}

void first_word(const char* name, char* dst, size_t cap) {
    size_t n = 0;
    while (name[n] && name[n] != ' ' && name[n] != '\t' && n + 1 < cap) { dst[n] = name[n]; ++n; }
    dst[n] = 0;
}

void tabular_line(FILE* f, DbAlignment* a) {
    char q[256], t[256];
    first_word(chainGetName(a->query), q, sizeof q);
    first_word(chainGetName(a->target), t, sizeof t);
    const Stats s = path_stats(a);
    const double pid = a->path_len ? 100.0 * s.identities / a->path_len : 0.0;
    fprintf(f, "%s\t%s\t%.2f\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%.2e\t%d\n", q, t, pid, a->path_len, s.mismatches, s.gap_opens,
            a->query_start + 1, a->query_end + 1, a->target_start + 1, a->target_end + 1, a->value, a->score);
}

void pairwise_block(FILE* f, DbAlignment* a) {
    const Stats s = path_stats(a);
    fprintf(f, ">%s\n          Length = %d\n\n Score = %d, Expect = %.2e\n Identities = %d/%d (%d%%), Positives = %d/%d (%d%%), Gaps = %d/%d (%d%%)\n\n",
            chainGetName(a->target), chainGetLength(a->target), a->score, a->value, s.identities, a->path_len,
            a->path_len ? 100 * s.identities / a->path_len : 0, s.positives, a->path_len,
            a->path_len ? 100 * s.positives / a->path_len : 0, s.gaps, a->path_len, a->path_len ? 100 * s.gaps / a->path_len : 0);
    int qi = a->query_start, ti = a->target_start;
    for (int b = 0; b < a->path_len; b += 60) {
        const int e = b + 60 < a->path_len ? b + 60 : a->path_len;
        char ql[64], ml[64], tl[64];
        const int q0 = qi, t0 = ti;
        for (int k = b; k < e; ++k) {
            const char m = a->path[k];
            char qc = '-', tc = '-';
            if (m == MOVE_DIAG || m == MOVE_UP) qc = chainGetChar(a->query, qi++);
            if (m == MOVE_DIAG || m == MOVE_LEFT) tc = chainGetChar(a->target, ti++);
            ql[k - b] = qc; tl[k - b] = tc;
            ml[k - b] = (m != MOVE_DIAG) ? ' ' : (qc == tc ? qc : (scorerScore(a->scorer, (char)(qc - 'A'), (char)(tc - 'A')) > 0 ? '+' : ' '));
        }
        ql[e - b] = ml[e - b] = tl[e - b] = 0;
        fprintf(f, "Query: %-5d %s %d\n             %s\nSbjct: %-5d %s %d\n\n", q0 + 1, ql, qi, ml, t0 + 1, tl, ti);
    }
}

}  // namespace

SW_API void outputShotgunDatabase(DbAlignment*** dbAlignments, int* dbAlignmentsLens, int dbAlignmentsLen, char* path, int type) {
    FILE* f = path ? fopen(path, "w") : stdout;
    SW_ASSERT(f != NULL, "cannot open output file '%s'", path);
    for (int i = 0; i < dbAlignmentsLen; ++i) {
        const int n = dbAlignmentsLens[i];
        if (n == 0 || !dbAlignments[i]) continue;
        Chain* query = dbAlignments[i][0]->query;
        switch (type) {
        case SW_OUT_DB_BLASTM9:
            fprintf(f, "# Query: %s\n# Fields: Query id, Subject id, %% identity, alignment length, mismatches, gap openings, "
                       "q. start, q. end, s. start, s. end, e-value, score\n", chainGetName(query));
            /* fall through */
        case SW_OUT_DB_BLASTM8:
            for (int j = 0; j < n; ++j) tabular_line(f, dbAlignments[i][j]);
            break;
        case SW_OUT_DB_LIGHT:
            for (int j = 0; j < n; ++j)
                fprintf(f, "%d\t%s\t%s\n", dbAlignments[i][j]->score, chainGetName(query), chainGetName(dbAlignments[i][j]->target));
            break;
        case SW_OUT_DB_BLASTM0:
        default:
            fprintf(f, "Query= %s\n         (%d letters)\n\n", chainGetName(query), chainGetLength(query));
            for (int j = 0; j < n; ++j) pairwise_block(f, dbAlignments[i][j]);
            break;
        }
    }
    if (f != stdout) fclose(f);
}

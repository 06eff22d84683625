// evalue.cpp -- createEValueParams / eValues (sift4g call sites: main.cpp:210,222; database_alignment.cpp:129-134).
//
// ASSUMPTION (unpinned): swsharp's own constants are in an absent submodule.  This file uses the PUBLISHED
// statistics of gapped local alignment as tabulated for NCBI BLAST (Altschul & Gish 1996; Park, Sheetlin, Spouge
// 2009-2012 for the finite-size correction):
//   * per (matrix, gap existence, gap extension): lambda, K, H, a (alpha), Alpha (alpha_v), Sigma;
//     ungapped row supplies a_un, Alpha_un;
//   * G = existence + extension; b = 2G(a_un - a); Beta = 2G(Alpha_un - Alpha); Tau = 2G(Alpha_un - Sigma);
//   * E(pair) = area(m, n, S) * K * exp(-lambda * S) with Spouge's finite-size corrected area, scaled from the
//     pair to the database by (database residues / target length).
// Gap convention: the scorer's `open` is the cost of the FIRST gap residue (main.cpp:297-303), BLAST's
// "existence" excludes it, so existence = open - extend.
#include <cmath>
#include <cstring>
#include <vector>

#include "impl.hpp"

namespace {

struct GumbelRow { int existence, extension; double lambda, K, H, a, Alpha, Sigma; };

// BLOSUM62, published NCBI table (existence, extension, lambda, K, H, alpha, alpha_v, sigma). Row 0 = ungapped.
const GumbelRow kBlosum62[] = {
    {0, 0, 0.3176, 0.134, 0.4012, 0.7916, 4.964660, 4.964660},
    {11, 2, 0.297, 0.082, 0.27, 1.1, 12.673800, 12.757600},
    {10, 2, 0.291, 0.075, 0.23, 1.3, 16.474000, 16.602600},
    {9, 2, 0.279, 0.058, 0.19, 1.5, 22.751900, 22.950000},
    {8, 2, 0.264, 0.045, 0.15, 1.8, 35.483800, 35.821300},
    {7, 2, 0.239, 0.027, 0.10, 2.5, 61.238300, 61.886000},
    {6, 2, 0.201, 0.012, 0.061, 3.3, 140.417000, 141.882000},
    {13, 1, 0.292, 0.071, 0.23, 1.2, 19.506300, 19.893100},
    {12, 1, 0.283, 0.059, 0.19, 1.5, 27.856200, 28.469900},
    {11, 1, 0.267, 0.041, 0.14, 1.9, 42.602800, 43.636200},
    {10, 1, 0.243, 0.024, 0.10, 2.5, 83.178700, 84.065100},
    {9, 1, 0.206, 0.010, 0.052, 4.0, 210.333000, 214.842000},
};
const int kDefaultRow = 9;   // BLAST's own default (11, 1), used when the scorer's penalties have no published row

}  // namespace

SW_API EValueParams* createEValueParams(long long length, Scorer* scorer) {
    SW_ASSERT(scorer != NULL, "createEValueParams: NULL scorer");
    EValueParams* p = new EValueParams();
    p->length = length > 0 ? length : 1;
    const GumbelRow* rows = kBlosum62;
    const int n_rows = (int)(sizeof(kBlosum62) / sizeof(kBlosum62[0]));
    if (strcmp(scorerGetName(scorer), "BLOSUM_62") != 0)
        fprintf(stderr, "[WARNING]: no e-value statistics for matrix '%s', using BLOSUM_62 parameters\n", scorerGetName(scorer));
    const int existence = scorerGetGapOpen(scorer) - scorerGetGapExtend(scorer);
    const int extension = scorerGetGapExtend(scorer);
    int row = -1;
    for (int i = 1; i < n_rows; ++i)
        if (rows[i].existence == existence && rows[i].extension == extension) row = i;
    p->gapped_row_found = row >= 0;
    if (row < 0) row = kDefaultRow;
    const GumbelRow& g = rows[row];
    const GumbelRow& u = rows[0];
    const double G = (double)(g.existence + g.extension);
    p->lambda = g.lambda; p->K = g.K; p->logK = log(g.K); p->H = g.H;
    p->a = g.a; p->alpha = g.Alpha; p->sigma = g.Sigma;
    p->b = 2.0 * G * (u.a - g.a);
    p->beta = 2.0 * G * (u.Alpha - g.Alpha);
    p->tau = 2.0 * G * (u.Alpha - g.Sigma);
    return p;
}

SW_API void deleteEValueParams(EValueParams* p) { delete p; }

// Spouge's finite-size corrected e-value.  Everything that depends on the score and the query length only is
// computed once per distinct score of a call (ScoreTerms); the per-target part is one erfc and one exp.  The expressions
// and their order are those of the straightforward formula, so the doubles are the same.
struct ScoreTerms { double sqrt_v, p1, P_m_F, c_y, expl, shift; bool set = false; };

static ScoreTerms score_terms(int score, int m, const EValueParams* p) {
    const double inv_sqrt_2pi = 0.39894228040143267793994605993438;
    const double y = (double)score;
    ScoreTerms t;
    t.shift = p->a * y + p->b;
    const double m_li_y = m - t.shift;
    double vi_y = p->alpha * y + p->beta;
    if (vi_y < 2.0 * p->alpha / p->lambda) vi_y = 2.0 * p->alpha / p->lambda;
    t.sqrt_v = sqrt(vi_y);
    const double m_F = m_li_y / t.sqrt_v;
    t.P_m_F = 0.5 * erfc(-m_F / sqrt(2.0));
    t.p1 = m_li_y * t.P_m_F + t.sqrt_v * inv_sqrt_2pi * exp(-0.5 * m_F * m_F);
    t.c_y = p->sigma * y + p->tau;
    if (t.c_y < 2.0 * p->sigma / p->lambda) t.c_y = 2.0 * p->sigma / p->lambda;
    t.expl = exp(-p->lambda * y);
    t.set = true;
    return t;
}

static double spouge_evalue(const ScoreTerms& t, int n, const EValueParams* p) {
    const double inv_sqrt_2pi = 0.39894228040143267793994605993438;
    const double scale = (double)p->length / (double)(n > 0 ? n : 1);   // pair -> database
    const double n_lj_y = n - t.shift;                                   // symmetric matrix: same a, b, alpha, beta
    const double n_F = n_lj_y / t.sqrt_v;
    const double P_n_F = 0.5 * erfc(-n_F / sqrt(2.0));
    const double p2 = n_lj_y * P_n_F + t.sqrt_v * inv_sqrt_2pi * exp(-0.5 * n_F * n_F);
    const double area = t.p1 * p2 + t.c_y * t.P_m_F * P_n_F;
    return area * p->K * t.expl * scale;
}

static double spouge_evalue(int score, int m, int n, const EValueParams* p) { return spouge_evalue(score_terms(score, m, p), n, p); }

SW_API void eValues(double* values, int* scores, Chain* query, Chain** database, int databaseLen,
                    int* cards, int cardsLen, EValueParams* p) {
    (void)cards; (void)cardsLen;   // double-precision host math on <= a few thousand scores: not worth a launch, and
                                   // device erfc/exp would differ from glibc by ULPs (SURVEY.md section 7 hard parts)
    const int m = chainGetLength(query);
    std::vector<ScoreTerms> memo;           // by score; most of a candidate list shares a few dozen small scores
    for (int i = 0; i < databaseLen; ++i) {
        const int sc = scores[i];
        if (sc < 0 || sc > (1 << 20)) { values[i] = spouge_evalue(sc, m, chainGetLength(database[i]), p); continue; }
        if ((size_t)sc >= memo.size()) memo.resize((size_t)sc + 1);
        if (!memo[sc].set) memo[sc] = score_terms(sc, m, p);
        values[i] = spouge_evalue(memo[sc], chainGetLength(database[i]), p);
    }
}

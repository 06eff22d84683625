// evalue.cpp -- createEValueParams / eValues (sift4g call sites: main.cpp:210,222; database_alignment.cpp:129-134).
//
// ASSUMPTION (unpinned): swsharp's own constants are in an absent submodule.  This file uses the PUBLISHED
// statistics of gapped local alignment as tabulated for NCBI BLAST (Altschul & Gish 1996; Park, Sheetlin, Spouge
// 2009-2012 for the finite-size correction):
//   * per (matrix, gap open, gap extension): lambda, K, H, a (alpha), Alpha (alpha_v), Sigma; the ungapped row supplies
//     a_un, Alpha_un;
//   * G = open + extension; b = 2G(a_un - a); Beta = 2G(Alpha_un - Alpha); Tau = 2G(Alpha_un - Sigma);
//   * E(pair) = area(m, n, S) * K * exp(-lambda * S) with Spouge's finite-size corrected area, scaled from the
//     pair to the database by (database residues / target length).
// There is no copy of these tables in this image and no network: the rows were typed in from the published tables.  What
// can be recomputed is recomputed by the tests (oracle/evalue_ref.py): every ungapped lambda, K and H from the matrix itself,
// and a = lambda / H for the ungapped rows.  The gapped rows are simulation results, and they are checked the way they were
// made (round 4): tests/golden/make_gumbel_mc.py scores 6.7e7 + 1.7e7 random pairs per row on the GPU under the scorer whose gap
// model the row describes and fits (lambda, K) under the finite-size corrected model below; tests/test_gumbel_mc.py (CPU) holds
// every row typed HERE against those fits -- lambda within 1.5 %, the predicted chance-hit probability at a tail of 1e-4 within
// [0.70, 1.25] of the observed one (a swapped digit fails).  All rows pass except PAM_250's, which are low by 1-3.7 % in lambda
// as published (conservative; one-sided bars).  The same simulation settles the convention question below STATISTICALLY: this
// library's open/extend 10/1 is BLAST's existence 9 / extension 1 (fit: lambda 0.206, K 0.011 = the (9, 1) row).
//
// Which row belongs to a scorer -- the one convention question:
//   The scorer's `open` is the cost of the FIRST gap residue (a gap of k costs open + (k-1) extend; main.cpp:297-303), whereas
//   the tables are indexed by BLAST's (existence, extension) with a gap of k costing existence + k extension.
//   * "swsharp" (default): the row whose two numbers EQUAL the scorer's (open, extend), G = open + extend.  This is what
//     upstream swsharp does as far as it can be recalled without its source -- it took the table over from BLAST and looks
//     its own penalties up in it unconverted -- so it is the behaviour a drop-in must reproduce: sift4g's default 10/1
//     gets the (10, 1) row, lambda 0.243, K 0.024.
//   * "blast": existence = open - extend, the statistically consistent reading of the same table: 10/1 gets (9, 1),
//     lambda 0.206, K 0.010 -- e-values about 40 times larger at a score of 100.  Round 1 did this.
//   S4G_EVALUE_CONVENTION=blast|swsharp selects (read when the parameters are created).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "impl.hpp"
#include <mutex>
#include <set>
#include <string>
#include "../matrices.h"

namespace {

struct GumbelRow { int open, extension; double lambda, K, H, a, Alpha, Sigma; };
struct GumbelTable { const char* matrix; const GumbelRow* rows; int n; int default_row; };

// Published NCBI tables (open, extension, lambda, K, H, alpha, alpha_v, sigma).  Row 0 of each = ungapped.
const GumbelRow kBlosum45[] = {
    {0, 0, 0.2291, 0.0924, 0.2514, 0.9113, 9.611060, 9.611060},
    {13, 3, 0.207, 0.049, 0.14, 1.5, 35.855900, 35.963900},
    {12, 3, 0.199, 0.039, 0.11, 1.8, 45.693600, 45.851700},
    {11, 3, 0.190, 0.031, 0.095, 2.0, 62.874100, 63.103700},
    {10, 3, 0.179, 0.023, 0.075, 2.4, 88.286800, 88.639100},
    {16, 2, 0.210, 0.051, 0.14, 1.5, 36.279800, 36.452400},
    {15, 2, 0.203, 0.041, 0.12, 1.7, 44.825700, 45.060400},
    {14, 2, 0.195, 0.032, 0.10, 1.9, 60.736200, 61.102300},
    {13, 2, 0.185, 0.024, 0.084, 2.2, 85.148100, 85.689400},
    {12, 2, 0.171, 0.016, 0.061, 2.8, 127.758000, 128.582000},
    {19, 1, 0.205, 0.040, 0.11, 1.9, 53.071400, 53.828200},
    {18, 1, 0.198, 0.032, 0.10, 2.0, 72.342400, 73.403900},
    {17, 1, 0.189, 0.024, 0.079, 2.4, 103.055000, 104.721000},
    {16, 1, 0.176, 0.016, 0.063, 2.8, 170.100000, 173.003000},
};
const GumbelRow kBlosum50[] = {
    {0, 0, 0.2318, 0.112, 0.3362, 0.6895, 5.388310, 5.388310},
    {13, 3, 0.212, 0.063, 0.19, 1.1, 18.113800, 18.202800},
    {12, 3, 0.206, 0.055, 0.17, 1.2, 22.654600, 22.777700},
    {11, 3, 0.197, 0.042, 0.14, 1.4, 29.861100, 30.045700},
    {10, 3, 0.186, 0.031, 0.11, 1.7, 42.393800, 42.674000},
    {9, 3, 0.172, 0.022, 0.082, 2.1, 66.069600, 66.516400},
    {16, 2, 0.215, 0.066, 0.20, 1.05, 17.951800, 18.092100},
    {15, 2, 0.210, 0.058, 0.17, 1.2, 21.940100, 22.141800},
    {14, 2, 0.202, 0.045, 0.14, 1.4, 28.681200, 28.961900},
    {13, 2, 0.193, 0.035, 0.12, 1.6, 42.059500, 42.471600},
    {12, 2, 0.181, 0.025, 0.095, 1.9, 63.747600, 64.397300},
    {19, 1, 0.212, 0.057, 0.18, 1.2, 26.311200, 26.923300},
    {18, 1, 0.207, 0.050, 0.15, 1.4, 34.903700, 35.734800},
    {17, 1, 0.198, 0.037, 0.12, 1.6, 48.895800, 50.148600},
    {16, 1, 0.186, 0.025, 0.10, 1.9, 76.469100, 78.443000},
    {15, 1, 0.171, 0.015, 0.063, 2.7, 140.053000, 144.160000},
};
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
const GumbelRow kBlosum80[] = {
    {0, 0, 0.3430, 0.177, 0.6568, 0.5222, 1.918130, 1.918130},
    {25, 2, 0.342, 0.17, 0.66, 0.52, 1.731000, 1.731300},
    {13, 2, 0.336, 0.15, 0.57, 0.59, 2.673470, 2.692300},
    {9, 2, 0.319, 0.11, 0.42, 0.76, 5.576090, 5.667860},
    {8, 2, 0.308, 0.090, 0.35, 0.89, 7.536950, 7.686230},
    {7, 2, 0.293, 0.070, 0.27, 1.1, 11.586600, 11.840400},
    {6, 2, 0.268, 0.045, 0.19, 1.4, 19.958100, 20.441200},
    {11, 1, 0.314, 0.095, 0.35, 0.90, 8.808610, 9.223320},
    {10, 1, 0.299, 0.071, 0.27, 1.1, 13.833800, 14.533400},
    {9, 1, 0.279, 0.048, 0.20, 1.4, 24.252000, 25.490400},
};
const GumbelRow kBlosum90[] = {
    {0, 0, 0.3346, 0.190, 0.7547, 0.4434, 1.377760, 1.377760},
    {9, 2, 0.310, 0.12, 0.46, 0.67, 4.232290, 4.334170},
    {8, 2, 0.300, 0.099, 0.39, 0.76, 5.797020, 5.961420},
    {7, 2, 0.283, 0.072, 0.30, 0.93, 9.040880, 9.321600},
    {6, 2, 0.259, 0.048, 0.22, 1.2, 16.024400, 16.531600},
    {11, 1, 0.302, 0.093, 0.39, 0.78, 7.143250, 7.619190},
    {10, 1, 0.290, 0.075, 0.28, 1.04, 11.483900, 12.269800},
    {9, 1, 0.265, 0.044, 0.20, 1.3, 21.408300, 22.840900},
};
const GumbelRow kPam30[] = {
    {0, 0, 0.3400, 0.283, 1.754, 0.1938, 0.161818, 0.161818},
    {7, 2, 0.305, 0.15, 0.87, 0.35, 1.014010, 1.162730},
    {6, 2, 0.287, 0.11, 0.68, 0.42, 1.688060, 1.951430},
    {5, 2, 0.264, 0.079, 0.45, 0.59, 3.377010, 3.871950},
    {10, 1, 0.309, 0.15, 0.88, 0.35, 1.372050, 1.788770},
    {9, 1, 0.294, 0.11, 0.61, 0.48, 2.463920, 3.186150},
    {8, 1, 0.270, 0.072, 0.40, 0.68, 5.368130, 6.763480},
};
const GumbelRow kPam70[] = {
    {0, 0, 0.3345, 0.229, 1.029, 0.3250, 0.633439, 0.633439},
    {8, 2, 0.301, 0.12, 0.54, 0.56, 2.881650, 3.025710},
    {7, 2, 0.286, 0.093, 0.43, 0.67, 4.534540, 4.785780},
    {6, 2, 0.264, 0.064, 0.29, 0.90, 7.942630, 8.402720},
    {11, 1, 0.305, 0.12, 0.52, 0.59, 3.681400, 4.108020},
    {10, 1, 0.291, 0.091, 0.41, 0.71, 6.002970, 6.716570},
    {9, 1, 0.270, 0.060, 0.28, 0.97, 11.360800, 12.636700},
};
const GumbelRow kPam250[] = {
    {0, 0, 0.2252, 0.0868, 0.2223, 0.98, 11.754300, 11.754300},
    {15, 3, 0.205, 0.049, 0.13, 1.6, 34.578400, 34.928000},
    {14, 3, 0.200, 0.043, 0.12, 1.7, 43.353000, 43.443800},
    {13, 3, 0.194, 0.036, 0.10, 1.9, 50.948500, 51.081700},
    {12, 3, 0.186, 0.029, 0.085, 2.2, 69.606500, 69.793600},
    {11, 3, 0.174, 0.020, 0.070, 2.5, 98.653500, 98.927100},
    {17, 2, 0.204, 0.047, 0.12, 1.7, 41.583800, 41.735800},
    {16, 2, 0.198, 0.038, 0.11, 1.8, 51.635200, 51.843900},
    {15, 2, 0.191, 0.031, 0.087, 2.2, 67.256700, 67.558500},
    {14, 2, 0.182, 0.024, 0.073, 2.5, 96.315100, 96.756800},
    {13, 2, 0.171, 0.017, 0.059, 2.9, 135.653000, 136.339000},
    {21, 1, 0.205, 0.045, 0.11, 1.8, 48.728200, 49.218800},
    {20, 1, 0.199, 0.037, 0.10, 1.9, 60.832000, 61.514100},
    {19, 1, 0.192, 0.029, 0.083, 2.3, 84.019700, 84.985600},
    {18, 1, 0.183, 0.021, 0.070, 2.6, 113.829000, 115.184000},
    {17, 1, 0.171, 0.014, 0.052, 3.3, 175.071000, 177.196000},
};
#define S4G_ROWS(x) x, (int)(sizeof(x) / sizeof(x[0]))
// default_row: BLAST's own default penalties for the matrix, used (with a warning) when the scorer's have no row
const GumbelTable kTables[] = {
    {"BLOSUM_62", S4G_ROWS(kBlosum62), 9},      // (11, 1)
    {"BLOSUM_45", S4G_ROWS(kBlosum45), 7},      // (14, 2)
    {"BLOSUM_50", S4G_ROWS(kBlosum50), 9},      // (13, 2)
    {"BLOSUM_80", S4G_ROWS(kBlosum80), 8},      // (10, 1)
    {"BLOSUM_90", S4G_ROWS(kBlosum90), 6},      // (10, 1)
    {"PAM_30", S4G_ROWS(kPam30), 5},            // (9, 1)
    {"PAM_70", S4G_ROWS(kPam70), 5},            // (10, 1)
    {"PAM_250", S4G_ROWS(kPam250), 9},          // (14, 2)
};

bool blast_convention() {
    const char* v = getenv("S4G_EVALUE_CONVENTION");
    if (!v || !*v || strcmp(v, "swsharp") == 0) return false;
    if (strcmp(v, "blast") == 0) return true;
    fprintf(stderr, "[WARNING]: S4G_EVALUE_CONVENTION='%s' is neither 'swsharp' nor 'blast', using 'swsharp'\n", v);
    return false;
}

}  // namespace

SW_API EValueParams* createEValueParams(long long length, Scorer* scorer) {
    SW_ASSERT(scorer != NULL, "createEValueParams: NULL scorer");
    EValueParams* p = new EValueParams();
    p->length = length > 0 ? length : 1;
    const char* canon = s4g::matrix_canonical_name(scorerGetName(scorer));
    const GumbelTable* tab = &kTables[0];
    bool have_table = false;
    for (const GumbelTable& t : kTables)
        if (canon && strcmp(canon, t.matrix) == 0) { tab = &t; have_table = true; }
    if (!have_table)
        fprintf(stderr, "[WARNING]: no e-value statistics for matrix '%s', using BLOSUM_62 parameters\n", scorerGetName(scorer));
    const bool blast = blast_convention();
    const int extension = scorerGetGapExtend(scorer);
    const int open = blast ? scorerGetGapOpen(scorer) - extension : scorerGetGapOpen(scorer);
    int row = -1;
    for (int i = 1; i < tab->n; ++i)
        if (tab->rows[i].open == open && tab->rows[i].extension == extension) row = i;
    p->gapped_row_found = row >= 0;
    if (row < 0) {
        row = tab->default_row;
        fprintf(stderr, "[WARNING]: no e-value statistics for gap penalties %d/%d with matrix '%s' (%s convention), using the row for %d/%d\n",
                scorerGetGapOpen(scorer), extension, tab->matrix, blast ? "blast" : "swsharp", tab->rows[row].open, tab->rows[row].extension);
    }
    const GumbelRow& g = tab->rows[row];
    const GumbelRow& u = tab->rows[0];
    // Say once which reading is in force when the two readings disagree and nobody chose.  The default looks
    // the scorer's penalties up as they are -- what upstream swsharp does as far as anyone here recalls (main.cpp:71-76,210 hand
    // them over unconverted) --, while the published rows are indexed by BLAST's (existence, extension) with a gap of k costing
    // existence + k * extension, i.e. existence = open - extend for this scorer (main.cpp:297-303).  tests/test_gumbel_mc.py
    // shows the converted row is the one this scorer's scores actually follow.
    if (!blast && have_table && !(getenv("S4G_EVALUE_CONVENTION") && *getenv("S4G_EVALUE_CONVENTION"))) {
        int other = -1;
        for (int i = 1; i < tab->n; ++i)
            if (tab->rows[i].open == scorerGetGapOpen(scorer) - extension && tab->rows[i].extension == extension) other = i;
        if (other >= 0 && other != row) {
            static std::mutex mu;
            static std::set<std::string> said;
            char key[96];
            snprintf(key, sizeof key, "%s/%d/%d", tab->matrix, scorerGetGapOpen(scorer), extension);
            std::lock_guard<std::mutex> lk(mu);
            if (said.insert(key).second) {
                const GumbelRow& o = tab->rows[other];
                // how far apart the two readings are where sift4g cuts (main.cpp:75: --evalue 1e-4 by default): the score S at
                // which this row says E = 1e-4 for a 350 x 350 pair in this database, and what the other row says there
                const double mn = 350.0 * 350.0 * ((double)p->length / 350.0);
                const double S = (log(g.K * mn) - log(1e-4)) / g.lambda;
                const double e_other = o.K * mn * exp(-o.lambda * S);
                fprintf(stderr, "[NOTE]: e-values for gap penalties %d/%d with matrix '%s' use the published row %d/%d%s (lambda %.3f, K %.3f): the "
                                "penalties are looked up as they are (the 'swsharp' convention, the default).  Read as BLAST's existence/extension "
                                "(existence = open - extend) they select row %d/%d (lambda %.3f, K %.3f), under which a hit this row gives E = 1e-04 "
                                "has E ~ %.0e; set S4G_EVALUE_CONVENTION=blast for that reading, =swsharp to keep this one without this note "
                                "(INTEGRATION.md section 5).\n",
                        scorerGetGapOpen(scorer), extension, tab->matrix, g.open, g.extension, p->gapped_row_found ? "" : " (no row for them: the matrix's default)",
                        g.lambda, g.K, o.open, o.extension, o.lambda, o.K, e_other);
            }
        }
    }
    if (getenv("S4G_PROFILE"))      // which published row this scorer's statistics come from (the lookup convention is a switch, see above)
        fprintf(stderr, "[s4g profile] e-value statistics: matrix %s, gap penalties %d/%d looked up %s (%s convention) -> row %d/%d%s: lambda %.4f K %.4f H %.3f alpha %.2f\n",
                tab->matrix, scorerGetGapOpen(scorer), extension, blast ? "as existence = open - extend" : "as they are", blast ? "blast" : "swsharp",
                g.open, g.extension, p->gapped_row_found ? "" : " (no such row: the matrix's default)", g.lambda, g.K, g.H, g.Alpha);
    const double G = (double)(g.open + g.extension);
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

static void evalues_range(double* values, const int* scores, int m, Chain** database, int begin, int end, const EValueParams* p) {
    std::vector<ScoreTerms> memo;           // by score; most of a candidate list shares a few dozen small scores
    for (int i = begin; i < end; ++i) {
        const int sc = scores[i];
        if (sc < 0 || sc > (1 << 20)) { values[i] = spouge_evalue(sc, m, chainGetLength(database[i]), p); continue; }
        if ((size_t)sc >= memo.size()) memo.resize((size_t)sc + 1);
        if (!memo[sc].set) memo[sc] = score_terms(sc, m, p);
        values[i] = spouge_evalue(memo[sc], chainGetLength(database[i]), p);
    }
}

SW_API void eValues(double* values, int* scores, Chain* query, Chain** database, int databaseLen,
                    int* cards, int cardsLen, EValueParams* p) {
    (void)cards; (void)cardsLen;   // double-precision host math on <= a few thousand scores: not worth a launch, and
                                   // device erfc/exp would differ from glibc by ULPs (SURVEY.md section 7 hard parts)
    const int m = chainGetLength(query);
    // A few thousand values are a quarter of a millisecond of erfc / exp -- as much as the device takes to score them -- and
    // every value depends on its own (score, length) only: slices go to the thread pool, unless this IS a pool thread.
    const int workers = swimpl::on_pool_thread() ? 0 : std::min(swimpl::pool_threads(), 8);
    if (workers < 2 || databaseLen < 1024) { evalues_range(values, scores, m, database, 0, databaseLen, p); return; }
    struct Slice { double* values; const int* scores; int m; Chain** database; int begin, end; const EValueParams* p; };
    const int parts = std::min(workers, databaseLen / 384);
    std::vector<Slice> slices((size_t)parts);
    std::vector<ThreadPoolTask*> tasks((size_t)parts, (ThreadPoolTask*)NULL);
    for (int k = 0; k < parts; ++k) {
        slices[k] = {values, scores, m, database, (int)((long long)databaseLen * k / parts), (int)((long long)databaseLen * (k + 1) / parts), p};
        if (k) tasks[k] = threadPoolSubmit([](void* x) -> void* { Slice* s = (Slice*)x; evalues_range(s->values, s->scores, s->m, s->database, s->begin, s->end, s->p); return NULL; }, &slices[k]);
    }
    evalues_range(values, scores, m, database, slices[0].begin, slices[0].end, p);
    for (int k = 1; k < parts; ++k) { threadPoolTaskWait(tasks[k]); threadPoolTaskDelete(tasks[k]); }
}

/*
 * sw_oracle.c -- CPU restatement of the sift4g / swsharp Smith-Waterman
 * database-scoring path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Nothing under sift4g_amd/ links, imports or calls it:
 * the product path is the HIP library and fails loudly without a GPU.
 *
 * PARITY UNPINNED.  The reference's scorer lives in the third-party module
 * mkorpar/swsharp, declared as a git submodule (/root/reference/.gitmodules:1-3)
 * whose directory /root/reference/vendor/swsharp is empty and whose pinned
 * revision is not recoverable (no .git, no lockfile, no network).  The
 * reference holds no tests, golden vectors or expected outputs for this path
 * (test_files/ has inputs only).  This file therefore restates the PUBLISHED
 * algorithm (Smith-Waterman with Gotoh affine gaps) anchored on the
 * reference's own call sites:
 *
 *   - alphabet: residue code = letter - 'A', 0..25, 5 bits
 *       /root/reference/sift4g/src/hash.cpp:16-19 (kProtMaxValue = 25,
 *       kProtBitLength = 5; kNumDiffKmers[3] = (25<<10|25<<5|25)+2)
 *   - scorer: scorerCreateMatrix(&scorer, "BLOSUM_62", 10, 1)
 *       /root/reference/sift4g/src/main.cpp:71-73,208
 *   - gap model: "gap-extend ... must be less or equal to gap opening
 *       penalty" /root/reference/sift4g/src/main.cpp:297-303  => `open` is
 *       the cost of the first gap residue; a gap of k residues costs
 *       open + (k-1)*extend
 *   - call contract: alignDatabase(out, outLen, SW_ALIGN, query, db, scorer,
 *       maxAlignments, valueFunction, param, valueThreshold, ...)
 *       /root/reference/sift4g/src/database_alignment.cpp:83-86
 *   - result type: int scores handed to valueFunction(double*, int*, ...)
 *       /root/reference/sift4g/src/database_alignment.cpp:129-134
 *
 * What pins this oracle instead: hand-computed known-answer vectors in
 * tests/golden/micro.json, the Karlin-Altschul lambda of the substitution
 * table against its published value, and fixture identities on the
 * reference's test_files (see tests/test_oracle.py).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define S4G_ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* BLOSUM62 as published by NCBI (24 symbols, order below).             */
/* ------------------------------------------------------------------ */
static const char NCBI_ORDER[25] = "ARNDCQEGHILKMFPSTWYVBZX*";

static const signed char NCBI_BLOSUM62[24][24] = {
/*        A   R   N   D   C   Q   E   G   H   I   L   K   M   F   P   S   T   W   Y   V   B   Z   X   * */
/* A */ { 4, -1, -2, -2,  0, -1, -1,  0, -2, -1, -1, -1, -1, -2, -1,  1,  0, -3, -2,  0, -2, -1,  0, -4},
/* R */ {-1,  5,  0, -2, -3,  1,  0, -2,  0, -3, -2,  2, -1, -3, -2, -1, -1, -3, -2, -3, -1,  0, -1, -4},
/* N */ {-2,  0,  6,  1, -3,  0,  0,  0,  1, -3, -3,  0, -2, -3, -2,  1,  0, -4, -2, -3,  3,  0, -1, -4},
/* D */ {-2, -2,  1,  6, -3,  0,  2, -1, -1, -3, -4, -1, -3, -3, -1,  0, -1, -4, -3, -3,  4,  1, -1, -4},
/* C */ { 0, -3, -3, -3,  9, -3, -4, -3, -3, -1, -1, -3, -1, -2, -3, -1, -1, -2, -2, -1, -3, -3, -2, -4},
/* Q */ {-1,  1,  0,  0, -3,  5,  2, -2,  0, -3, -2,  1,  0, -3, -1,  0, -1, -2, -1, -2,  0,  3, -1, -4},
/* E */ {-1,  0,  0,  2, -4,  2,  5, -2,  0, -3, -3,  1, -2, -3, -1,  0, -1, -3, -2, -2,  1,  4, -1, -4},
/* G */ { 0, -2,  0, -1, -3, -2, -2,  6, -2, -4, -4, -2, -3, -3, -2,  0, -2, -2, -3, -3, -1, -2, -1, -4},
/* H */ {-2,  0,  1, -1, -3,  0,  0, -2,  8, -3, -3, -1, -2, -1, -2, -1, -2, -2,  2, -3,  0,  0, -1, -4},
/* I */ {-1, -3, -3, -3, -1, -3, -3, -4, -3,  4,  2, -3,  1,  0, -3, -2, -1, -3, -1,  3, -3, -3, -1, -4},
/* L */ {-1, -2, -3, -4, -1, -2, -3, -4, -3,  2,  4, -2,  2,  0, -3, -2, -1, -2, -1,  1, -4, -3, -1, -4},
/* K */ {-1,  2,  0, -1, -3,  1,  1, -2, -1, -3, -2,  5, -1, -3, -1,  0, -1, -3, -2, -2,  0,  1, -1, -4},
/* M */ {-1, -1, -2, -3, -1,  0, -2, -3, -2,  1,  2, -1,  5,  0, -2, -1, -1, -1, -1,  1, -3, -1, -1, -4},
/* F */ {-2, -3, -3, -3, -2, -3, -3, -3, -1,  0,  0, -3,  0,  6, -4, -2, -2,  1,  3, -1, -3, -3, -1, -4},
/* P */ {-1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4,  7, -1, -1, -4, -3, -2, -2, -1, -2, -4},
/* S */ { 1, -1,  1,  0, -1,  0,  0,  0, -1, -2, -2,  0, -1, -2, -1,  4,  1, -3, -2, -2,  0,  0,  0, -4},
/* T */ { 0, -1,  0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1,  1,  5, -2, -2,  0, -1, -1,  0, -4},
/* W */ {-3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1,  1, -4, -3, -2, 11,  2, -3, -4, -3, -2, -4},
/* Y */ {-2, -2, -2, -3, -2, -1, -2, -3,  2, -1, -1, -2, -1,  3, -3, -2, -2,  2,  7, -1, -3, -2, -1, -4},
/* V */ { 0, -3, -3, -3, -1, -2, -2, -3, -3,  3,  1, -2,  1, -1, -2, -2,  0, -3, -1,  4, -3, -2, -1, -4},
/* B */ {-2, -1,  3,  4, -3,  0,  1, -1,  0, -3, -4,  0, -3, -3, -2,  0, -1, -4, -3, -3,  4,  1, -1, -4},
/* Z */ {-1,  0,  0,  1, -3,  3,  4, -2,  0, -3, -3,  1, -1, -3, -1,  0, -1, -3, -2, -2,  1,  4, -1, -4},
/* X */ { 0, -1, -1, -1, -2, -1, -1, -1, -1, -1, -1, -1, -1, -1, -2,  0,  0, -2, -1, -1, -1, -1, -1, -4},
/* * */ {-4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4,  1},
};

/*
 * Letter-indexed 26x26 table (code = letter - 'A'), the shape the swsharp
 * Scorer API exposes to sift4g (codes 0..25, hash.cpp:16-19).
 * ASSUMPTION (unpinned): the three letters NCBI gives no row -- J, O, U --
 * score as X against everything (and X-vs-X among themselves).  '*' has no
 * letter code and cannot occur in a Chain.
 */
S4G_ORACLE_API void s4g_oracle_blosum62(int32_t out[26 * 26]) {
    int map[26];
    for (int c = 0; c < 26; ++c) {
        const char* p = strchr(NCBI_ORDER, 'A' + c);
        map[c] = p ? (int)(p - NCBI_ORDER) : 22 /* X */;
    }
    for (int a = 0; a < 26; ++a)
        for (int b = 0; b < 26; ++b)
            out[a * 26 + b] = NCBI_BLOSUM62[map[a]][map[b]];
}

/* ------------------------------------------------------------------ */
/* Recurrence (SURVEY.md 8a row a-R):                                   */
/*   E[i][j] = max(E[i][j-1] - ext,  H[i][j-1] - open)                  */
/*   F[i][j] = max(F[i-1][j] - ext,  H[i-1][j] - open)                  */
/*   H[i][j] = max(0, H[i-1][j-1] + M[q_i][t_j], E[i][j], F[i][j])      */
/*   score   = max over all cells of H                                  */
/* Boundaries: H = 0 on row/column 0, E and F = -infinity there.        */
/* ------------------------------------------------------------------ */
#define NEG_INF (-(1 << 29))

static inline int imax(int a, int b) { return a > b ? a : b; }

/* work = 2*(lt+1) ints of scratch; returns the optimal local score */
static int sw_score_work(const uint8_t* q, int lq, const uint8_t* t, int lt,
                         const int32_t* M, int open, int ext, int* work) {
    if (lq <= 0 || lt <= 0) return 0;
    int* Hrow = work;            /* H[i-1][*] then H[i][*], index j = 0..lt */
    int* Frow = work + (lt + 1); /* F[i][*] carried down the rows           */
    for (int j = 0; j <= lt; ++j) { Hrow[j] = 0; Frow[j] = NEG_INF; }
    int best = 0;
    for (int i = 1; i <= lq; ++i) {
        const int32_t* Mq = M + 26 * q[i - 1];
        int diag = Hrow[0];   /* H[i-1][0] = 0 */
        int hleft = 0;        /* H[i][0]       */
        int e = NEG_INF;      /* E[i][0]       */
        for (int j = 1; j <= lt; ++j) {
            e = imax(e - ext, hleft - open);
            int f = imax(Frow[j] - ext, Hrow[j] - open);
            int h = diag + Mq[t[j - 1]];
            h = imax(h, e);
            h = imax(h, f);
            h = imax(h, 0);
            diag = Hrow[j];
            Hrow[j] = h;
            Frow[j] = f;
            hleft = h;
            best = imax(best, h);
        }
    }
    return best;
}

int s4g_oracle_sw_score(const uint8_t* q, int lq, const uint8_t* t, int lt, const int32_t* M, int open, int ext);
S4G_ORACLE_API int s4g_oracle_sw_score(const uint8_t* q, int lq, const uint8_t* t, int lt,
                                       const int32_t* M, int open, int ext) {
    if (lq <= 0 || lt <= 0) return 0;
    int* work = (int*)malloc(sizeof(int) * 2 * ((size_t)lt + 1));
    int s = sw_score_work(q, lq, t, lt, M, open, ext, work);
    free(work);
    return s;
}

/*
 * Full DP with traceback, for the post-scoring stages (alignment of the
 * survivors; consumed by /root/reference/sift4g/src/select_alignments.cpp:244-300).
 * Moves use the swsharp names seen at that call site: DIAG, LEFT (gap in
 * query: consumes a target residue), UP (gap in target: consumes a query
 * residue).  ASSUMPTION (unpinned) on ties: the end cell is the first maximum
 * in row-major order (smallest i, then smallest j); walking back, H prefers
 * DIAG, then UP (F), then LEFT (E); inside a gap, extension is preferred over
 * opening only when opening does not reproduce the value.
 * path_out must hold lq+lt bytes; moves are written start->end.
 * Returns the score; fills q_start,q_end,t_start,t_end (0-based inclusive)
 * and *path_len.  Codes: 0 = DIAG, 1 = LEFT, 2 = UP.
 */
S4G_ORACLE_API int s4g_oracle_sw_align(const uint8_t* q, int lq, const uint8_t* t, int lt,
                                       const int32_t* M, int open, int ext,
                                       int* q_start, int* q_end, int* t_start, int* t_end,
                                       uint8_t* path_out, int* path_len) {
    *q_start = *q_end = *t_start = *t_end = -1;
    *path_len = 0;
    if (lq <= 0 || lt <= 0) return 0;
    size_t W = (size_t)lt + 1, N = ((size_t)lq + 1) * W;
    int* H = (int*)malloc(N * sizeof(int));
    int* E = (int*)malloc(N * sizeof(int));
    int* F = (int*)malloc(N * sizeof(int));
    for (size_t j = 0; j < W; ++j) { H[j] = 0; E[j] = NEG_INF; F[j] = NEG_INF; }
    int best = 0, bi = 0, bj = 0;
    for (int i = 1; i <= lq; ++i) {
        const int32_t* Mq = M + 26 * q[i - 1];
        H[i * W] = 0; E[i * W] = NEG_INF; F[i * W] = NEG_INF;
        for (int j = 1; j <= lt; ++j) {
            size_t c = i * W + j;
            E[c] = imax(E[c - 1] - ext, H[c - 1] - open);
            F[c] = imax(F[c - W] - ext, H[c - W] - open);
            int h = imax(0, H[c - W - 1] + Mq[t[j - 1]]);
            h = imax(h, imax(E[c], F[c]));
            H[c] = h;
            if (h > best) { best = h; bi = i; bj = j; }
        }
    }
    if (best == 0) { free(H); free(E); free(F); return 0; }
    int i = bi, j = bj, n = 0;
    int state = 0; /* 0 = in H, 1 = in E (LEFT run), 2 = in F (UP run) */
    uint8_t* rev = path_out;
    /* the one tie rule that is a switch (S4G_TRACEBACK_TIES): "left" = LEFT (E) before UP (F) */
    const char* ties = getenv("S4G_TRACEBACK_TIES");
    const int left_first = ties && strcmp(ties, "left") == 0;
    while (i > 0 && j > 0) {
        size_t c = i * W + j;
        if (state == 0) {
            if (H[c] == 0) break;
            if (H[c] == H[c - W - 1] + M[26 * q[i - 1] + t[j - 1]]) { rev[n++] = 0; --i; --j; }
            else if (left_first ? H[c] != E[c] : H[c] == F[c]) state = 2;
            else state = 1;
        } else if (state == 1) {
            rev[n++] = 1;
            if (E[c] == H[c - 1] - open) state = 0;
            --j;
        } else {
            rev[n++] = 2;
            if (F[c] == H[c - W] - open) state = 0;
            --i;
        }
    }
    *q_start = i; *t_start = j; *q_end = bi - 1; *t_end = bj - 1;
    for (int a = 0, b = n - 1; a < b; ++a, --b) { uint8_t x = rev[a]; rev[a] = rev[b]; rev[b] = x; }
    *path_len = n;
    free(H); free(E); free(F);
    return best;
}

/* ------------------------------------------------------------------ */
/* Pair-parallel drivers (pthreads) -- used as the timed CPU baseline   */
/* and for bulk parity checks.                                          */
/* ------------------------------------------------------------------ */
typedef struct {
    const uint8_t* qcodes; const uint64_t* qoff; uint32_t nq;
    const uint8_t* tcodes; const uint64_t* toff; uint32_t nt;
    const uint32_t* pair_q; const uint32_t* pair_t; uint64_t npairs; /* NULL => full scan */
    const int32_t* M; int open, ext;
    int32_t* out;
    uint64_t next; pthread_mutex_t mu; uint64_t grain; uint32_t max_lt;
} Job;

static void* worker(void* arg) {
    Job* jb = (Job*)arg;
    int* work = (int*)malloc(sizeof(int) * 2 * ((size_t)jb->max_lt + 1));
    uint64_t total = jb->pair_q ? jb->npairs : (uint64_t)jb->nq * jb->nt;
    for (;;) {
        pthread_mutex_lock(&jb->mu);
        uint64_t b = jb->next; jb->next += jb->grain;
        pthread_mutex_unlock(&jb->mu);
        if (b >= total) break;
        uint64_t e = b + jb->grain < total ? b + jb->grain : total;
        for (uint64_t p = b; p < e; ++p) {
            uint32_t qi = jb->pair_q ? jb->pair_q[p] : (uint32_t)(p / jb->nt);
            uint32_t ti = jb->pair_q ? jb->pair_t[p] : (uint32_t)(p % jb->nt);
            jb->out[p] = sw_score_work(jb->qcodes + jb->qoff[qi], (int)(jb->qoff[qi + 1] - jb->qoff[qi]),
                                       jb->tcodes + jb->toff[ti], (int)(jb->toff[ti + 1] - jb->toff[ti]),
                                       jb->M, jb->open, jb->ext, work);
        }
    }
    free(work);
    return NULL;
}

static void run_job(Job* jb, int threads) {
    if (threads < 1) threads = 1;
    jb->next = 0; jb->grain = 16; jb->max_lt = 0;
    for (uint32_t i = 0; i < jb->nt; ++i) {
        uint64_t l = jb->toff[i + 1] - jb->toff[i];
        if (l > jb->max_lt) jb->max_lt = (uint32_t)l;
    }
    pthread_mutex_init(&jb->mu, NULL);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    for (int i = 0; i < threads; ++i) pthread_create(&th[i], NULL, worker, jb);
    for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
    free(th);
    pthread_mutex_destroy(&jb->mu);
}

/* out[q * nt + t] for every query x every target */
S4G_ORACLE_API void s4g_oracle_scan(const uint8_t* qcodes, const uint64_t* qoff, uint32_t nq,
                                    const uint8_t* tcodes, const uint64_t* toff, uint32_t nt,
                                    const int32_t* M, int open, int ext, int threads, int32_t* out) {
    Job jb; memset(&jb, 0, sizeof jb);
    jb.qcodes = qcodes; jb.qoff = qoff; jb.nq = nq; jb.tcodes = tcodes; jb.toff = toff; jb.nt = nt;
    jb.M = M; jb.open = open; jb.ext = ext; jb.out = out;
    run_job(&jb, threads);
}

/* out[p] for p-th (query, target) pair */
S4G_ORACLE_API void s4g_oracle_pairs(const uint8_t* qcodes, const uint64_t* qoff, uint32_t nq,
                                     const uint8_t* tcodes, const uint64_t* toff, uint32_t nt,
                                     const uint32_t* pair_q, const uint32_t* pair_t, uint64_t npairs,
                                     const int32_t* M, int open, int ext, int threads, int32_t* out) {
    Job jb; memset(&jb, 0, sizeof jb);
    jb.qcodes = qcodes; jb.qoff = qoff; jb.nq = nq; jb.tcodes = tcodes; jb.toff = toff; jb.nt = nt;
    jb.pair_q = pair_q; jb.pair_t = pair_t; jb.npairs = npairs;
    jb.M = M; jb.open = open; jb.ext = ext; jb.out = out;
    run_job(&jb, threads);
}

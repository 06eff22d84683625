/*
 * sw_simd.c -- AVX2 inter-sequence Smith-Waterman for the timed CPU baseline.  TEST INFRASTRUCTURE ONLY
 * (same rules as sw_oracle.c: only tests/ and bench.py's cpu_baseline leg may load it; parity unpinned).
 *
 * Why it exists: the reference's CPU path for this step is swsharp's SSE database scorer (absent submodule), i.e.
 * vector code.  Timing the scalar restatement beside the GPU would understate a CPU by an order of magnitude, so the
 * baseline is this vectorised port of the same recurrence, in the precision ladder that class of code uses (SWIPE, swimd / OPAL;
 * SURVEY.md 2a): first 32 database sequences per AVX2 register in UNSIGNED 8-BIT lanes (scores biased by -min score, saturating
 * adds and subtracts, the clamp at zero for free), then -- only for the pairs whose 8-bit result reached the saturation mark --
 * 16 sequences per register in int16 lanes, then the scalar oracle for what saturates int16 too (>= 32767).  Inter-sequence like
 * the GPU kernels, a per-column score profile built with byte shuffles, the classic 9- / 10-operation inner loops.  Validated
 * against sw_oracle.c in tests/test_oracle.py; it is NOT swsharp's code.  s4g_oracle_scan_simd16 is the int16 tier alone (rounds
 * 1-4's baseline), kept so that the tests pin both and bench.py can report what the 8-bit tier buys.
 *
 * Recurrence and conventions: see sw_oracle.c (SURVEY.md 8a row a-R; /root/reference/sift4g/src/main.cpp:71-73,297-303).
 */
#include <immintrin.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define S4G_ORACLE_API __attribute__((visibility("default")))
#define LANES 16

int s4g_oracle_sw_score(const uint8_t* q, int lq, const uint8_t* t, int lt, const int32_t* M, int open, int ext);

S4G_ORACLE_API int s4g_oracle_simd_available(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }

typedef struct {
    const uint8_t* qcodes; const uint64_t* qoff; uint32_t nq;
    const uint8_t* tcodes; const uint64_t* toff; uint32_t nt;
    const uint32_t* order;          /* targets sorted by length, longest first */
    const int32_t* M; int open, ext;
    int32_t* out;                   /* [nq][nt] */
    uint32_t n_batches; uint32_t next; pthread_mutex_t mu;
    uint32_t max_lq;
    int bias8;                      /* 8-bit tier: -min score (0 = tier unusable: see scan_simd) */
    uint64_t escalated;             /* pairs the 8-bit tier handed to int16 */
} SimdJob;

/* one batch of <= 16 targets against every query */
__attribute__((target("avx2")))
static void batch_avx2(const SimdJob* jb, uint32_t b, int16_t* Hbuf, int16_t* Ebuf, uint8_t* tr) {
    const uint32_t first = b * LANES;
    uint32_t n = jb->nt - first < LANES ? jb->nt - first : LANES;
    uint32_t ids[LANES]; uint32_t len[LANES]; uint32_t maxlen = 0;
    for (uint32_t l = 0; l < LANES; ++l) {
        ids[l] = l < n ? jb->order[first + l] : 0xFFFFFFFFu;
        len[l] = l < n ? (uint32_t)(jb->toff[ids[l] + 1] - jb->toff[ids[l]]) : 0;
        if (len[l] > maxlen) maxlen = len[l];
    }
    /* transposed residues: tr[j*16 + lane], pad code 31 */
    for (uint32_t j = 0; j < maxlen; ++j)
        for (uint32_t l = 0; l < LANES; ++l)
            tr[j * LANES + l] = j < len[l] ? jb->tcodes[jb->toff[ids[l]] + j] : 31;
    /* substitution rows as 32-byte shuffle tables (entries 26..31 = -64: padding scores) */
    __m256i rows[26];
    for (int a = 0; a < 26; ++a) {
        int8_t tab[32];
        for (int c = 0; c < 32; ++c) tab[c] = c < 26 ? (int8_t)jb->M[a * 26 + c] : (int8_t)-64;
        rows[a] = _mm256_loadu_si256((const __m256i*)tab);
    }
    const __m256i vopen = _mm256_set1_epi16((short)jb->open), vext = _mm256_set1_epi16((short)jb->ext);
    const __m256i zero = _mm256_setzero_si256();
    __m256i prof[26];
    for (uint32_t qi = 0; qi < jb->nq; ++qi) {
        const uint8_t* q = jb->qcodes + jb->qoff[qi];
        const int lq = (int)(jb->qoff[qi + 1] - jb->qoff[qi]);
        __m256i* H = (__m256i*)Hbuf; __m256i* E = (__m256i*)Ebuf;
        for (int i = 0; i < lq; ++i) { H[i] = zero; E[i] = zero; }
        __m256i best = zero;
        for (uint32_t j = 0; j < maxlen; ++j) {
            /* column profile: prof[a][lane] = M[a][t_lane] via two 16-entry byte shuffles */
            const __m128i t8 = _mm_loadu_si128((const __m128i*)(tr + j * LANES));
            const __m256i t2 = _mm256_broadcastsi128_si256(t8);                      /* indices in both halves */
            const __m256i hi_sel = _mm256_cmpgt_epi8(t2, _mm256_set1_epi8(15));
            for (int a = 0; a < 26; ++a) {
                const __m256i lo_tab = _mm256_permute2x128_si256(rows[a], rows[a], 0x00);
                const __m256i hi_tab = _mm256_permute2x128_si256(rows[a], rows[a], 0x11);
                const __m256i lo = _mm256_shuffle_epi8(lo_tab, _mm256_and_si256(t2, _mm256_set1_epi8(15)));
                const __m256i hi = _mm256_shuffle_epi8(hi_tab, _mm256_and_si256(t2, _mm256_set1_epi8(15)));
                const __m256i s8 = _mm256_blendv_epi8(lo, hi, hi_sel);               /* 16 scores (int8) in the low lane */
                prof[a] = _mm256_cvtepi8_epi16(_mm256_castsi256_si128(s8));
            }
            __m256i diag = zero, F = zero;
            for (int i = 0; i < lq; ++i) {
                __m256i h = _mm256_adds_epi16(diag, prof[q[i]]);
                h = _mm256_max_epi16(h, E[i]);
                h = _mm256_max_epi16(h, F);
                h = _mm256_max_epi16(h, zero);
                best = _mm256_max_epi16(best, h);
                const __m256i hm = _mm256_subs_epu16(h, vopen);
                E[i] = _mm256_max_epi16(_mm256_subs_epu16(E[i], vext), hm);
                F = _mm256_max_epi16(_mm256_subs_epu16(F, vext), hm);
                diag = H[i];
                H[i] = h;
            }
        }
        int16_t res[LANES];
        _mm256_storeu_si256((__m256i*)res, best);
        for (uint32_t l = 0; l < n; ++l) {
            int32_t s = res[l];
            if (s >= 32767)   /* may have saturated: exact scalar recomputation */
                s = s4g_oracle_sw_score(q, lq, jb->tcodes + jb->toff[ids[l]], (int)len[l], jb->M, jb->open, jb->ext);
            jb->out[(size_t)qi * jb->nt + ids[l]] = s;
        }
    }
}

/* ---- the 8-bit tier: 32 targets per register ---------------------------------------------------------------------------
 * Unsigned bytes, every substitution score stored as s + bias (bias = -min score, >= 0; the padding residue scores like the
 * worst mismatch, s + bias = 0: a non-positive score can never raise a local maximum, see sw_oracle.c).
 *   h = max(diag (+sat) (s + bias) (-sat) bias, E, F)      the saturating subtract IS the clamp at zero
 * An add that saturates leaves h = 255 - bias, so a lane whose running maximum stays below that mark never saturated and is
 * exact; the others go to the int16 tier (rescore16). */
#define LANES8 32

/* one query against <= 16 explicit targets in int16 lanes (the escalation step of the 8-bit tier) */
__attribute__((target("avx2")))
static void rescore16(const SimdJob* jb, uint32_t qi, const uint32_t* ids, uint32_t n, int16_t* Hbuf, int16_t* Ebuf, uint8_t* tr) {
    uint32_t len[LANES]; uint32_t maxlen = 0;
    for (uint32_t l = 0; l < LANES; ++l) {
        len[l] = l < n ? (uint32_t)(jb->toff[ids[l] + 1] - jb->toff[ids[l]]) : 0;
        if (len[l] > maxlen) maxlen = len[l];
    }
    for (uint32_t j = 0; j < maxlen; ++j)
        for (uint32_t l = 0; l < LANES; ++l)
            tr[j * LANES + l] = (l < n && j < len[l]) ? jb->tcodes[jb->toff[ids[l]] + j] : 31;
    __m256i rows[26];
    for (int a = 0; a < 26; ++a) {
        int8_t tab[32];
        for (int c = 0; c < 32; ++c) tab[c] = c < 26 ? (int8_t)jb->M[a * 26 + c] : (int8_t)-64;
        rows[a] = _mm256_loadu_si256((const __m256i*)tab);
    }
    const __m256i vopen = _mm256_set1_epi16((short)jb->open), vext = _mm256_set1_epi16((short)jb->ext);
    const __m256i zero = _mm256_setzero_si256();
    const uint8_t* q = jb->qcodes + jb->qoff[qi];
    const int lq = (int)(jb->qoff[qi + 1] - jb->qoff[qi]);
    /* only the query's own residues need a profile row */
    uint8_t used[26]; memset(used, 0, sizeof used);
    for (int i = 0; i < lq; ++i) used[q[i]] = 1;
    __m256i prof[26];
    __m256i* H = (__m256i*)Hbuf; __m256i* E = (__m256i*)Ebuf;
    for (int i = 0; i < lq; ++i) { H[i] = zero; E[i] = zero; }
    __m256i best = zero;
    for (uint32_t j = 0; j < maxlen; ++j) {
        const __m128i t8 = _mm_loadu_si128((const __m128i*)(tr + j * LANES));
        const __m256i t2 = _mm256_broadcastsi128_si256(t8);
        const __m256i hi_sel = _mm256_cmpgt_epi8(t2, _mm256_set1_epi8(15));
        const __m256i idx = _mm256_and_si256(t2, _mm256_set1_epi8(15));
        for (int a = 0; a < 26; ++a) {
            if (!used[a]) continue;
            const __m256i lo = _mm256_shuffle_epi8(_mm256_permute2x128_si256(rows[a], rows[a], 0x00), idx);
            const __m256i hi = _mm256_shuffle_epi8(_mm256_permute2x128_si256(rows[a], rows[a], 0x11), idx);
            prof[a] = _mm256_cvtepi8_epi16(_mm256_castsi256_si128(_mm256_blendv_epi8(lo, hi, hi_sel)));
        }
        __m256i diag = zero, F = zero;
        for (int i = 0; i < lq; ++i) {
            __m256i h = _mm256_adds_epi16(diag, prof[q[i]]);
            h = _mm256_max_epi16(h, E[i]);
            h = _mm256_max_epi16(h, F);
            h = _mm256_max_epi16(h, zero);
            best = _mm256_max_epi16(best, h);
            const __m256i hm = _mm256_subs_epu16(h, vopen);
            E[i] = _mm256_max_epi16(_mm256_subs_epu16(E[i], vext), hm);
            F = _mm256_max_epi16(_mm256_subs_epu16(F, vext), hm);
            diag = H[i];
            H[i] = h;
        }
    }
    int16_t res[LANES];
    _mm256_storeu_si256((__m256i*)res, best);
    for (uint32_t l = 0; l < n; ++l) {
        int32_t sc = res[l];
        if (sc >= 32767)
            sc = s4g_oracle_sw_score(q, lq, jb->tcodes + jb->toff[ids[l]], (int)len[l], jb->M, jb->open, jb->ext);
        jb->out[(size_t)qi * jb->nt + ids[l]] = sc;
    }
}

/* one batch of <= 32 targets against every query: 8-bit lanes, escalation per (query, lane) */
__attribute__((target("avx2")))
static void batch_avx2_u8(SimdJob* jb, uint32_t b, uint8_t* Hbuf, uint8_t* Ebuf, uint8_t* tr, int16_t* H16, int16_t* E16, uint8_t* tr16) {
    const uint32_t first = b * LANES8;
    const uint32_t n = jb->nt - first < LANES8 ? jb->nt - first : LANES8;
    uint32_t ids[LANES8]; uint32_t len[LANES8]; uint32_t maxlen = 0;
    for (uint32_t l = 0; l < LANES8; ++l) {
        ids[l] = l < n ? jb->order[first + l] : 0xFFFFFFFFu;
        len[l] = l < n ? (uint32_t)(jb->toff[ids[l] + 1] - jb->toff[ids[l]]) : 0;
        if (len[l] > maxlen) maxlen = len[l];
    }
    for (uint32_t j = 0; j < maxlen; ++j)
        for (uint32_t l = 0; l < LANES8; ++l)
            tr[j * LANES8 + l] = j < len[l] ? jb->tcodes[jb->toff[ids[l]] + j] : 31;
    const int bias = jb->bias8;
    /* substitution rows + bias as two 16-entry shuffle tables, each in both 128-bit halves (entries 26..31: padding, 0) */
    __m256i rows_lo[26], rows_hi[26];
    for (int a = 0; a < 26; ++a) {
        uint8_t tab[32];
        for (int c = 0; c < 32; ++c) tab[c] = c < 26 ? (uint8_t)(jb->M[a * 26 + c] + bias) : 0;
        const __m128i lo = _mm_loadu_si128((const __m128i*)tab), hi = _mm_loadu_si128((const __m128i*)(tab + 16));
        rows_lo[a] = _mm256_broadcastsi128_si256(lo);
        rows_hi[a] = _mm256_broadcastsi128_si256(hi);
    }
    const __m256i vbias = _mm256_set1_epi8((char)bias), vopen = _mm256_set1_epi8((char)jb->open), vext = _mm256_set1_epi8((char)jb->ext);
    const __m256i zero = _mm256_setzero_si256();
    const int mark = 255 - bias;                    /* a lane whose maximum reaches it may have saturated */
    __m256i prof[26];
    uint64_t escalated = 0;
    for (uint32_t qi = 0; qi < jb->nq; ++qi) {
        const uint8_t* q = jb->qcodes + jb->qoff[qi];
        const int lq = (int)(jb->qoff[qi + 1] - jb->qoff[qi]);
        __m256i* H = (__m256i*)Hbuf; __m256i* E = (__m256i*)Ebuf;
        for (int i = 0; i < lq; ++i) { H[i] = zero; E[i] = zero; }
        __m256i best = zero;
        for (uint32_t j = 0; j < maxlen; ++j) {
            const __m256i t = _mm256_loadu_si256((const __m256i*)(tr + j * LANES8));
            const __m256i hi_sel = _mm256_cmpgt_epi8(t, _mm256_set1_epi8(15));
            const __m256i idx = _mm256_and_si256(t, _mm256_set1_epi8(15));
            for (int a = 0; a < 26; ++a)
                prof[a] = _mm256_blendv_epi8(_mm256_shuffle_epi8(rows_lo[a], idx), _mm256_shuffle_epi8(rows_hi[a], idx), hi_sel);
            __m256i diag = zero, F = zero;
            for (int i = 0; i < lq; ++i) {
                __m256i h = _mm256_subs_epu8(_mm256_adds_epu8(diag, prof[q[i]]), vbias);
                h = _mm256_max_epu8(h, E[i]);
                h = _mm256_max_epu8(h, F);
                best = _mm256_max_epu8(best, h);
                const __m256i hm = _mm256_subs_epu8(h, vopen);
                E[i] = _mm256_max_epu8(_mm256_subs_epu8(E[i], vext), hm);
                F = _mm256_max_epu8(_mm256_subs_epu8(F, vext), hm);
                diag = H[i];
                H[i] = h;
            }
        }
        uint8_t res[LANES8];
        _mm256_storeu_si256((__m256i*)res, best);
        uint32_t redo[LANES8]; uint32_t n_redo = 0;
        for (uint32_t l = 0; l < n; ++l) {
            if (res[l] >= mark) redo[n_redo++] = ids[l];
            else jb->out[(size_t)qi * jb->nt + ids[l]] = res[l];
        }
        for (uint32_t r = 0; r < n_redo; r += LANES)
            rescore16(jb, qi, redo + r, n_redo - r < LANES ? n_redo - r : LANES, H16, E16, tr16);
        escalated += n_redo;
    }
    __atomic_fetch_add(&jb->escalated, escalated, __ATOMIC_RELAXED);
}

static void* simd_worker8(void* arg) {
    SimdJob* jb = (SimdJob*)arg;
    uint32_t maxlen = 0;
    for (uint32_t i = 0; i < jb->nt; ++i) { uint64_t l = jb->toff[i + 1] - jb->toff[i]; if (l > maxlen) maxlen = (uint32_t)l; }
    uint8_t* H = (uint8_t*)aligned_alloc(32, (size_t)(jb->max_lq + 1) * 32);
    uint8_t* E = (uint8_t*)aligned_alloc(32, (size_t)(jb->max_lq + 1) * 32);
    uint8_t* tr = (uint8_t*)aligned_alloc(32, ((size_t)(maxlen + 1) * LANES8 + 31) / 32 * 32);
    int16_t* H16 = (int16_t*)aligned_alloc(32, (size_t)(jb->max_lq + 1) * 32);
    int16_t* E16 = (int16_t*)aligned_alloc(32, (size_t)(jb->max_lq + 1) * 32);
    uint8_t* tr16 = (uint8_t*)malloc((size_t)(maxlen + 1) * LANES);
    for (;;) {
        pthread_mutex_lock(&jb->mu);
        uint32_t b = jb->next++;
        pthread_mutex_unlock(&jb->mu);
        if (b >= jb->n_batches) break;
        batch_avx2_u8(jb, b, H, E, tr, H16, E16, tr16);
    }
    free(H); free(E); free(tr); free(H16); free(E16); free(tr16);
    return NULL;
}

static void* simd_worker(void* arg) {
    SimdJob* jb = (SimdJob*)arg;
    uint32_t maxlen = 0;
    for (uint32_t i = 0; i < jb->nt; ++i) { uint64_t l = jb->toff[i + 1] - jb->toff[i]; if (l > maxlen) maxlen = (uint32_t)l; }
    int16_t* H = (int16_t*)aligned_alloc(32, (size_t)(jb->max_lq + 1) * 32);
    int16_t* E = (int16_t*)aligned_alloc(32, (size_t)(jb->max_lq + 1) * 32);
    uint8_t* tr = (uint8_t*)malloc((size_t)(maxlen + 1) * LANES);
    for (;;) {
        pthread_mutex_lock(&jb->mu);
        uint32_t b = jb->next++;
        pthread_mutex_unlock(&jb->mu);
        if (b >= jb->n_batches) break;
        batch_avx2(jb, b, H, E, tr);
    }
    free(H); free(E); free(tr);
    return NULL;
}

static const SimdJob* g_sort_job;
static int by_len_desc(const void* a, const void* b) {
    const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    const uint64_t lx = g_sort_job->toff[x + 1] - g_sort_job->toff[x], ly = g_sort_job->toff[y + 1] - g_sort_job->toff[y];
    return lx < ly ? 1 : (lx > ly ? -1 : (x < y ? -1 : (x > y ? 1 : 0)));
}

static int scan_simd_impl(const uint8_t* qcodes, const uint64_t* qoff, uint32_t nq, const uint8_t* tcodes, const uint64_t* toff, uint32_t nt,
                          const int32_t* M, int open, int ext, int threads, int32_t* out, int tier8, uint64_t* escalated) {
    if (escalated) *escalated = 0;
    if (!s4g_oracle_simd_available()) return -1;
    if (nq == 0 || nt == 0) return 0;
    SimdJob jb; memset(&jb, 0, sizeof jb);
    jb.qcodes = qcodes; jb.qoff = qoff; jb.nq = nq; jb.tcodes = tcodes; jb.toff = toff; jb.nt = nt;
    jb.M = M; jb.open = open; jb.ext = ext; jb.out = out;
    /* the 8-bit tier needs every score + bias and both penalties in a byte with room for scores above them */
    int mn = 0, mx = 0;
    for (int i = 0; i < 26 * 26; ++i) { if (M[i] < mn) mn = M[i]; if (M[i] > mx) mx = M[i]; }
    if (tier8 && (-mn + mx > 127 || open > 127 || ext > 127 || open < 0 || ext < 0)) tier8 = 0;
    jb.bias8 = -mn;
    uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * nt);
    for (uint32_t i = 0; i < nt; ++i) order[i] = i;
    g_sort_job = &jb;
    qsort(order, nt, sizeof(uint32_t), by_len_desc);
    jb.order = order;
    const uint32_t lanes = tier8 ? LANES8 : LANES;
    jb.n_batches = (nt + lanes - 1) / lanes;
    for (uint32_t i = 0; i < nq; ++i) { uint64_t l = qoff[i + 1] - qoff[i]; if (l > jb.max_lq) jb.max_lq = (uint32_t)l; }
    if (threads < 1) threads = 1;
    pthread_mutex_init(&jb.mu, NULL);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    for (int i = 0; i < threads; ++i) pthread_create(&th[i], NULL, tier8 ? simd_worker8 : simd_worker, &jb);
    for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
    free(th); free(order);
    pthread_mutex_destroy(&jb.mu);
    if (escalated) *escalated = jb.escalated;
    return 0;
}

/* out[q * nt + t] for every query x every target; returns 0, or -1 if the CPU has no AVX2 (nothing written).
 * The precision ladder: 8-bit lanes, int16 for what reached the 8-bit saturation mark, scalar for what saturates int16.
 * *escalated (may be NULL) = pairs the 8-bit tier handed on. */
S4G_ORACLE_API int s4g_oracle_scan_simd(const uint8_t* qcodes, const uint64_t* qoff, uint32_t nq,
                                        const uint8_t* tcodes, const uint64_t* toff, uint32_t nt,
                                        const int32_t* M, int open, int ext, int threads, int32_t* out) {
    return scan_simd_impl(qcodes, qoff, nq, tcodes, toff, nt, M, open, ext, threads, out, 1, NULL);
}
S4G_ORACLE_API int s4g_oracle_scan_simd8(const uint8_t* qcodes, const uint64_t* qoff, uint32_t nq,
                                         const uint8_t* tcodes, const uint64_t* toff, uint32_t nt,
                                         const int32_t* M, int open, int ext, int threads, int32_t* out, uint64_t* escalated) {
    return scan_simd_impl(qcodes, qoff, nq, tcodes, toff, nt, M, open, ext, threads, out, 1, escalated);
}
/* the int16 tier alone (every pair in 16 lanes; scalar beyond 32767) */
S4G_ORACLE_API int s4g_oracle_scan_simd16(const uint8_t* qcodes, const uint64_t* qoff, uint32_t nq,
                                          const uint8_t* tcodes, const uint64_t* toff, uint32_t nt,
                                          const int32_t* M, int open, int ext, int threads, int32_t* out) {
    return scan_simd_impl(qcodes, qoff, nq, tcodes, toff, nt, M, open, ext, threads, out, 0, NULL);
}

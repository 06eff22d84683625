// core.cpp -- Chain, FASTA readers, Scorer, device cards, thread pool of the swsharp-compatible API.
// Call sites that fix each signature are cited in include/swsharp/swsharp.h.
#include "impl.hpp"

#include <pthread.h>

#include <cctype>
#include <cstring>
#include <deque>

#include "../matrices.h"

// ------------------------------------------------------------------------------------------------ Chain
static Chain* make_chain(ChainData* d, int start, int length, bool reverse) {
    Chain* c = (Chain*)calloc(1, sizeof(Chain));
    c->data = d; c->start = start; c->length = length; c->reverse = reverse;
    d->refs++;
    if (reverse && length > 0) {
        c->rev_chars = (char*)malloc(length);
        c->rev_codes = (char*)malloc(length);
        for (int i = 0; i < length; ++i) {
            c->rev_chars[i] = d->chars[start + length - 1 - i];
            c->rev_codes[i] = d->codes[start + length - 1 - i];
        }
    }
    return c;
}

SW_API char scorerEncode(char c) { return (char)(toupper((unsigned char)c) - 'A'); }
SW_API char scorerDecode(char c) { return (char)(c + 'A'); }

SW_API Chain* chainCreate(char* name, int nameLen, char* string, int stringLen) {
    SW_ASSERT(name != NULL && string != NULL && nameLen >= 0 && stringLen >= 0, "chainCreate: invalid arguments");
    ChainData* d = (ChainData*)calloc(1, sizeof(ChainData));
    d->name = (char*)malloc(nameLen + 1);
    memcpy(d->name, name, nameLen);
    d->name[nameLen] = 0;
    d->chars = (char*)malloc(stringLen + 1);
    d->codes = (char*)malloc(stringLen + 1);
    int n = 0;
    for (int i = 0; i < stringLen; ++i) {          // letters only; everything else (digits, '*', '-', blanks) is dropped
        unsigned char ch = (unsigned char)string[i];
        if (isalpha(ch)) {
            d->chars[n] = (char)toupper(ch);
            d->codes[n] = (char)(toupper(ch) - 'A');
            ++n;
        }
    }
    d->chars[n] = 0;
    d->length = n;
    return make_chain(d, 0, n, false);
}

SW_API Chain* chainCreateView(Chain* chain, int start, int end, int reverse) {
    SW_ASSERT(chain != NULL, "chainCreateView: NULL chain");
    if (start < 0) start = 0;
    if (end >= chain->length) end = chain->length - 1;
    int len = end >= start ? end - start + 1 : 0;
    SW_ASSERT(!chain->reverse, "chainCreateView: views of reversed views are not supported");
    return make_chain(chain->data, chain->start + start, len, reverse != 0);
}

SW_API void chainDelete(Chain* chain) {
    if (!chain) return;
    ChainData* d = chain->data;
    free(chain->rev_chars);
    free(chain->rev_codes);
    free(chain);
    if (--d->refs == 0) { free(d->name); free(d->chars); free(d->codes); free(d); }
}

SW_API int chainGetLength(Chain* chain) { return chain->length; }
SW_API const char* chainGetName(Chain* chain) { return chain->data->name; }
SW_API const char* chainGetCodes(Chain* chain) { return chain->reverse ? chain->rev_codes : chain->data->codes + chain->start; }
SW_API char chainGetChar(Chain* chain, int index) {
    return chain->reverse ? chain->rev_chars[index] : chain->data->chars[chain->start + index];
}
SW_API char chainGetCode(Chain* chain, int index) { return chainGetCodes(chain)[index]; }

// ------------------------------------------------------------------------------------------------ FASTA
namespace {

struct FastaState {              // one per open handle; keeps the look-ahead header between parts
    FILE* fh;
    std::string pending_name;
    bool have_pending;
    bool eof;
};
std::vector<FastaState*> g_fasta;
pthread_mutex_t g_fasta_mu = PTHREAD_MUTEX_INITIALIZER;

FastaState* fasta_state(FILE* fh, bool create) {
    pthread_mutex_lock(&g_fasta_mu);
    FastaState* st = NULL;
    for (auto* s : g_fasta) if (s->fh == fh) st = s;
    if (!st && create) { st = new FastaState{fh, "", false, false}; g_fasta.push_back(st); }
    pthread_mutex_unlock(&g_fasta_mu);
    return st;
}

void fasta_forget(FILE* fh) {
    pthread_mutex_lock(&g_fasta_mu);
    for (size_t i = 0; i < g_fasta.size(); ++i)
        if (g_fasta[i]->fh == fh) { delete g_fasta[i]; g_fasta.erase(g_fasta.begin() + i); break; }
    pthread_mutex_unlock(&g_fasta_mu);
}

void push_chain(Chain*** chains, int* len, int* cap, Chain* c) {
    if (*len == *cap) {
        *cap = *cap ? *cap * 2 : 1024;
        *chains = (Chain**)realloc(*chains, sizeof(Chain*) * (size_t)*cap);
    }
    (*chains)[(*len)++] = c;
}

// Reads records until >= max_bytes of sequence were consumed (0 = unlimited) or EOF. Returns 1 if more remain.
int fasta_read(FastaState* st, Chain*** chains, int* len, size_t max_bytes) {
    int cap = *len;
    std::string seq, line;
    size_t consumed = 0;
    char buf[1 << 16];
    auto flush = [&]() {
        if (st->have_pending) {
            push_chain(chains, len, &cap, chainCreate((char*)st->pending_name.data(), (int)st->pending_name.size(),
                                                      (char*)seq.data(), (int)seq.size()));
            consumed += seq.size();
            seq.clear();
            st->have_pending = false;
        }
    };
    while (!st->eof) {
        if (!fgets(buf, sizeof buf, st->fh)) { st->eof = true; break; }
        line.assign(buf);
        while (!line.empty() && line.back() != '\n' && !feof(st->fh)) {     // long line: keep reading
            if (!fgets(buf, sizeof buf, st->fh)) break;
            line.append(buf);
        }
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (!line.empty() && line[0] == '>') {
            flush();
            st->pending_name = line.substr(1);
            st->have_pending = true;
            if (max_bytes && consumed >= max_bytes) break;       // header of the next part is kept in pending_name
        } else if (st->have_pending) {
            seq.append(line);
        }
    }
    if (st->eof) flush();
    if (*len < cap) *chains = (Chain**)realloc(*chains, sizeof(Chain*) * (size_t)(*len ? *len : 1));
    return st->eof ? 0 : 1;
}

}  // namespace

SW_API void readFastaChainsPartInit(Chain*** chains, int* chainsLen, FILE** handle, int* serialized, const char* path) {
    *handle = fopen(path, "r");
    SW_ASSERT(*handle != NULL, "cannot open file '%s'", path);
    *chains = NULL;
    *chainsLen = 0;
    *serialized = 0;          // no serialized-database cache in this implementation
    fasta_state(*handle, true);
}

SW_API int readFastaChainsPart(Chain*** chains, int* chainsLen, FILE* handle, int serialized, const size_t maxBytes) {
    (void)serialized;
    FastaState* st = fasta_state(handle, true);
    int more = fasta_read(st, chains, chainsLen, maxBytes);
    if (!more) fasta_forget(handle);   // caller fclose()s the handle itself (database_alignment.cpp:122)
    return more;
}

SW_API void readFastaChains(Chain*** chains, int* chainsLen, const char* path) {
    FILE* fh; int ser;
    readFastaChainsPartInit(chains, chainsLen, &fh, &ser, path);
    readFastaChainsPart(chains, chainsLen, fh, ser, 0);
    fclose(fh);
}

SW_API void deleteFastaChains(Chain** chains, int chainsLen) {
    if (!chains) return;
    for (int i = 0; i < chainsLen; ++i) if (chains[i]) chainDelete(chains[i]);
    free(chains);
}

// ------------------------------------------------------------------------------------------------ Scorer
static void scorer_finish(Scorer* s) {
    s->max_score = 0;
    for (int i = 0; i < 26 * 26; ++i) if (s->table[i] > s->max_score) s->max_score = s->table[i];
    int rc = s4g_scorer_create(&s->dev, s->table, s->gap_open, s->gap_extend);
    SW_ASSERT(rc == S4G_OK, "invalid scorer: %s", s4g_last_error());
}

SW_API void scorerCreateMatrix(Scorer** scorer, char* name, int gapOpen, int gapExtend) {
    Scorer* s = new Scorer();
    s->name = name ? name : "";
    s->gap_open = gapOpen; s->gap_extend = gapExtend; s->dev = NULL;
    SW_ASSERT(s4g::matrix_by_name(s->name.c_str(), s->table), "unknown or unsupported similarity matrix '%s' (available: BLOSUM_62)",
              s->name.c_str());
    scorer_finish(s);
    *scorer = s;
}

SW_API void scorerCreate(Scorer** scorer, const char* name, int* scores, char maxCode, int gapOpen, int gapExtend) {
    SW_ASSERT(maxCode > 0 && maxCode <= 26, "scorerCreate: maxCode must be in 1..26");
    Scorer* s = new Scorer();
    s->name = name ? name : "";
    s->gap_open = gapOpen; s->gap_extend = gapExtend; s->dev = NULL;
    for (int i = 0; i < 26 * 26; ++i) s->table[i] = 0;
    for (int a = 0; a < maxCode; ++a) for (int b = 0; b < maxCode; ++b) s->table[a * 26 + b] = scores[a * maxCode + b];
    scorer_finish(s);
    *scorer = s;
}

void swimpl::scorer_release(Scorer* s) {
    if (s && s->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) { s4g_scorer_destroy(s->dev); delete s; }
}
SW_API void scorerDelete(Scorer* scorer) { swimpl::scorer_release(scorer); }
SW_API int scorerGetGapOpen(Scorer* s) { return s->gap_open; }
SW_API int scorerGetGapExtend(Scorer* s) { return s->gap_extend; }
SW_API const char* scorerGetName(Scorer* s) { return s->name.c_str(); }
SW_API int scorerScore(Scorer* s, char a, char b) { return s->table[(int)a * 26 + (int)b]; }
SW_API int scorerGetMaxScore(Scorer* s) { return s->max_score; }

// ------------------------------------------------------------------------------------------------ cards
SW_API void cudaGetCards(int** cards, int* cardsLen) {
    int n = 0;
    if (s4g_device_count(&n) != S4G_OK) n = 0;
    *cards = (int*)malloc(sizeof(int) * (n ? n : 1));
    for (int i = 0; i < n; ++i) (*cards)[i] = i;
    *cardsLen = n;
}

SW_API int cudaCheckCards(int* cards, int cardsLen) { return s4g_device_check(cards, cardsLen); }

// ------------------------------------------------------------------------------------------------ thread pool
struct ThreadPoolTask {
    void* (*routine)(void*);
    void* param;
    bool done;
    pthread_mutex_t mu;
    pthread_cond_t cv;
};

namespace {
pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;
pthread_cond_t g_pool_cv = PTHREAD_COND_INITIALIZER;
std::deque<ThreadPoolTask*> g_queue;
std::vector<pthread_t> g_workers;
bool g_pool_stop = false;

void run_task(ThreadPoolTask* t) {
    t->routine(t->param);
    pthread_mutex_lock(&t->mu);
    t->done = true;
    pthread_cond_broadcast(&t->cv);
    pthread_mutex_unlock(&t->mu);
}

void* pool_worker(void*) {
    for (;;) {
        pthread_mutex_lock(&g_pool_mu);
        while (g_queue.empty() && !g_pool_stop) pthread_cond_wait(&g_pool_cv, &g_pool_mu);
        if (g_queue.empty() && g_pool_stop) { pthread_mutex_unlock(&g_pool_mu); return NULL; }
        ThreadPoolTask* t = g_queue.front();
        g_queue.pop_front();
        pthread_mutex_unlock(&g_pool_mu);
        run_task(t);
    }
}
}  // namespace

SW_API int threadPoolInitialize(int n) {
    pthread_mutex_lock(&g_pool_mu);
    bool fresh = g_workers.empty();
    if (fresh) {
        g_pool_stop = false;
        if (n < 1) n = 1;
        g_workers.resize(n);
        for (int i = 0; i < n; ++i) pthread_create(&g_workers[i], NULL, pool_worker, NULL);
    }
    pthread_mutex_unlock(&g_pool_mu);
    return fresh ? 0 : -1;
}

SW_API void threadPoolTerminate(void) {
    pthread_mutex_lock(&g_pool_mu);
    g_pool_stop = true;
    pthread_cond_broadcast(&g_pool_cv);
    std::vector<pthread_t> workers;
    workers.swap(g_workers);
    pthread_mutex_unlock(&g_pool_mu);
    for (pthread_t& w : workers) pthread_join(w, NULL);
}

SW_API ThreadPoolTask* threadPoolSubmit(void* (*routine)(void*), void* param) {
    ThreadPoolTask* t = new ThreadPoolTask{routine, param, false, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER};
    pthread_mutex_lock(&g_pool_mu);
    bool have_pool = !g_workers.empty() && !g_pool_stop;
    if (have_pool) { g_queue.push_back(t); pthread_cond_signal(&g_pool_cv); }
    pthread_mutex_unlock(&g_pool_mu);
    if (!have_pool) run_task(t);   // no pool: run inline
    return t;
}

SW_API void threadPoolTaskWait(ThreadPoolTask* task) {
    if (!task) return;            // sift4g waits on NULL for skipped queries (select_alignments.cpp:57-59,68)
    pthread_mutex_lock(&task->mu);
    while (!task->done) pthread_cond_wait(&task->cv, &task->mu);
    pthread_mutex_unlock(&task->mu);
}

SW_API void threadPoolTaskDelete(ThreadPoolTask* task) {
    if (!task) return;
    pthread_mutex_destroy(&task->mu);
    pthread_cond_destroy(&task->cv);
    delete task;
}

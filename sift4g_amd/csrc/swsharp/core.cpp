// core.cpp -- Chain, FASTA readers, Scorer, device cards, thread pool of the swsharp-compatible API.
// Call sites that fix each signature are cited in include/swsharp/swsharp.h.
#include "impl.hpp"

#include <atomic>
#include <pthread.h>

#include <cctype>
#include <cstring>
#include <sys/stat.h>
#include <unistd.h>
#include <deque>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "../matrices.h"

// ------------------------------------------------------------------------------------------------ Chain
// `home`: the Chain object reserved inside d's own block (chain_data_alloc), for the chain the block is made for -- a database
// is millions of chains, and one malloc/free pair per chain instead of two is a fifth of the time of reading and of deleting
// them; NULL for views, which get an object of their own.
static Chain* make_chain(ChainData* d, int start, int length, bool reverse, Chain* home = NULL) {
    Chain* c = home ? home : (Chain*)xcalloc(1, sizeof(Chain));
    if (home) memset(c, 0, sizeof(Chain));
    c->data = d; c->start = start; c->length = length; c->reverse = reverse; c->embedded = home != NULL;
    d->refs++;
    if (reverse && length > 0) {
        c->rev_chars = (char*)xmalloc(length);
        c->rev_codes = (char*)xmalloc(length);
        for (int i = 0; i < length; ++i) {
            c->rev_chars[i] = d->chars[start + length - 1 - i];
            c->rev_codes[i] = d->codes[start + length - 1 - i];
        }
    }
    return c;
}

SW_API char scorerEncode(char c) { return (char)(toupper((unsigned char)c) - 'A'); }
SW_API char scorerDecode(char c) { return (char)(c + 'A'); }

// ChainData, its name, its letters and its codes are ONE allocation (a database is millions of chains).
// kResidueCode: letters (either case) -> 0..25, everything else (digits, '*', '-', blanks) -> 0xFF = dropped.
namespace {
struct ResidueTable {
    unsigned char code[256];
    ResidueTable() {
        for (int c = 0; c < 256; ++c) code[c] = 0xFF;
        for (int c = 'A'; c <= 'Z'; ++c) { code[c] = (unsigned char)(c - 'A'); code[c - 'A' + 'a'] = (unsigned char)(c - 'A'); }
    }
};
const ResidueTable kResidue;

// layout of a block: ChainData, the Chain made for it (make_chain's `home`), name, letters, codes
inline Chain* chain_home(ChainData* d) { return (Chain*)((char*)d + sizeof(ChainData)); }
ChainData* chain_data_alloc(const char* name, size_t nameLen, size_t seqCap) {
    char* block = (char*)xmalloc(sizeof(ChainData) + sizeof(Chain) + nameLen + 1 + 2 * (seqCap + 1));
    ChainData* d = (ChainData*)block;
    d->name = block + sizeof(ChainData) + sizeof(Chain);
    d->chars = d->name + nameLen + 1;
    d->codes = d->chars + seqCap + 1;
    d->length = 0; d->refs = 0;
    memcpy(d->name, name, nameLen);
    d->name[nameLen] = 0;
    return d;
}

// appends the residues among src[0, n) to d (capacity is the caller's business); returns how many were kept
inline size_t chain_data_append(ChainData* d, const char* src, size_t n) {
    char* chars = d->chars + d->length;
    char* codes = d->codes + d->length;
    size_t k = 0, i = 0;
#if defined(__SSE2__)
    // upper-case letters only (nearly every record of a protein database): letters copied, codes = letters - 'A', 16 per step;
    // anything else in the run sends all of it through the table below
    if (n >= 16) {
        const __m128i a = _mm_set1_epi8('A'), bias = _mm_set1_epi8((char)-128), top = _mm_set1_epi8((char)(25 - 128));
        __m128i bad = _mm_setzero_si128();
        for (; i + 16 <= n; i += 16) {
            const __m128i v = _mm_loadu_si128((const __m128i*)(src + i));
            const __m128i t = _mm_sub_epi8(v, a);
            bad = _mm_or_si128(bad, _mm_cmpgt_epi8(_mm_add_epi8(t, bias), top));       // unsigned t > 25
            _mm_storeu_si128((__m128i*)(chars + i), v);
            _mm_storeu_si128((__m128i*)(codes + i), t);
        }
        if (_mm_movemask_epi8(bad)) i = 0; else k = i;
    }
#endif
    for (; i < n; ++i) {
        const unsigned char c = kResidue.code[(unsigned char)src[i]];
        chars[k] = (char)(c + 'A');
        codes[k] = (char)c;
        k += (c != 0xFF);
    }
    d->length += (int)k;
    return k;
}
}  // namespace

SW_API Chain* chainCreate(char* name, int nameLen, char* string, int stringLen) {
    SW_ASSERT(name != NULL && string != NULL && nameLen >= 0 && stringLen >= 0, "chainCreate: invalid arguments");
    ChainData* d = chain_data_alloc(name, (size_t)nameLen, (size_t)stringLen);
    chain_data_append(d, string, (size_t)stringLen);
    d->chars[d->length] = 0;
    return make_chain(d, 0, d->length, false, chain_home(d));
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
    chain->rev_chars = chain->rev_codes = NULL;
    if (!chain->embedded) free(chain);
    if (--d->refs == 0) free(d);          // the first chain, name, letters and codes live in the same block
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

// header of the serialized chain cache (see below)
// records / residues: totals of the file, patched into the header when the writing pass completes (fastaResidueCount reads them)
struct CacheHeader { char magic[8]; uint64_t fasta_size; int64_t fasta_mtime; uint64_t records, residues; };

struct FastaState {              // one per open handle; keeps the look-ahead header and the read buffer between parts
    FILE* fh;
    std::string pending_name;
    bool have_pending;
    bool eof;
    bool at_line_start = true;   // the next unread byte begins a line
    bool in_header = false;      // in the middle of a header line (it continued past the buffer)
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    std::vector<char> seq;       // raw sequence bytes of the pending record
    // serialized cache (S4G_FASTA_CACHE=1): this handle READS the cache file instead of FASTA text ...
    bool serialized_in = false;
    // ... or it reads FASTA text and WRITES the cache beside it, renamed into place when the whole file has been read
    FILE* cache_out = NULL;
    std::string cache_tmp, cache_final;
    uint64_t cache_records = 0, cache_residues = 0;      // written so far (residues as the chains count them: after the residue table)
    CacheHeader cache_header;
};

// ---- serialized chain cache ------------------------------------------------------------------------------------------
// swsharp's FASTA readers hand the caller a `serialized` flag (database_search.cpp:80-82,96-97; database_alignment.cpp:36,47):
// upstream can read a binary dump of the chains instead of parsing text.  sift4g reads its database TWICE per run (prefilter,
// then alignment), so with S4G_FASTA_CACHE=1 the first complete pass over a FASTA file leaves `<path>.s4gchains` beside it and
// every later pass -- the alignment stage of the same run included -- reads that: header (magic, size and mtime of the FASTA
// file it was made from, totals), then per record {name length, residues, raw sequence bytes of the text, name, the residue
// letters as the chain holds them}.  Reading is one fread straight into the chain's storage plus a subtraction for the codes.
// Measured on 600 000 chains / 247 MB: 0.62 s from the cache, 0.56 s from text -- a pass costs what allocating and first
// touching the chains costs (about 1 us per chain), not what parsing costs, so the cache's benefit is the totals in its header
// (fastaResidueCount: the full-scan build's counting stage becomes O(1)) and independence from the text's formatting, not
// speed.  Records and part boundaries are exactly those of the text reader (a part ends with the record whose raw
// bytes reach maxBytes), so the reference's results do not depend on which of the two it was given.  Off by default:
// nothing is written next to the user's database unless asked.
const char kCacheMagic[8] = {'S', '4', 'G', 'C', 'H', 'N', '3', 0};


bool fasta_cache_enabled() { const char* v = getenv("S4G_FASTA_CACHE"); return v && *v && strcmp(v, "0") != 0; }

bool stat_file(const char* path, uint64_t* size, int64_t* mtime) {
    struct stat sb;
    if (stat(path, &sb) != 0) return false;
    *size = (uint64_t)sb.st_size; *mtime = (int64_t)sb.st_mtime;
    return true;
}

// the cache of `path` if it exists and was made from the file as it is now
FILE* open_valid_cache(const char* path, CacheHeader* header = NULL) {
    uint64_t size; int64_t mtime;
    if (!stat_file(path, &size, &mtime)) return NULL;
    const std::string cache = std::string(path) + ".s4gchains";
    FILE* fh = fopen(cache.c_str(), "rb");
    if (!fh) return NULL;
    CacheHeader h;
    if (fread(&h, sizeof h, 1, fh) != 1 || memcmp(h.magic, kCacheMagic, 8) != 0 || h.fasta_size != size || h.fasta_mtime != mtime) {
        fclose(fh);
        return NULL;
    }
    if (header) *header = h;
    return fh;
}

void cache_abandon(FastaState* st) {       // the reader was dropped before EOF: no half-written cache stays behind
    if (!st->cache_out) return;
    fclose(st->cache_out);
    st->cache_out = NULL;
    remove(st->cache_tmp.c_str());
}
std::vector<FastaState*> g_fasta;
pthread_mutex_t g_fasta_mu = PTHREAD_MUTEX_INITIALIZER;

FastaState* fasta_state(FILE* fh, bool create) {
    pthread_mutex_lock(&g_fasta_mu);
    FastaState* st = NULL;
    for (auto* s : g_fasta) if (s->fh == fh) st = s;
    if (!st && create) { st = new FastaState(); st->fh = fh; st->have_pending = false; st->eof = false; g_fasta.push_back(st); }
    pthread_mutex_unlock(&g_fasta_mu);
    return st;
}

void fasta_forget(FILE* fh) {
    pthread_mutex_lock(&g_fasta_mu);
    for (size_t i = 0; i < g_fasta.size(); ++i)
        if (g_fasta[i]->fh == fh) { cache_abandon(g_fasta[i]); delete g_fasta[i]; g_fasta.erase(g_fasta.begin() + i); break; }
    pthread_mutex_unlock(&g_fasta_mu);
}

void push_chain(Chain*** chains, int* len, int* cap, Chain* c) {
    if (*len == *cap) {
        *cap = *cap ? *cap * 2 : 1024;
        *chains = (Chain**)xrealloc(*chains, sizeof(Chain*) * (size_t)*cap);
    }
    (*chains)[(*len)++] = c;
}

// Where a part's records go when the caller wants no Chain objects (readFastaCodesPart): residue codes back to back and
// one offset per record.  Lines are converted straight out of the read buffer; `raw` counts the sequence bytes of the pending
// record as the chain reader's st->seq would, so that parts end at the same records.
struct FlatSink {
    char* codes = NULL;             // malloc'ed, handed to the caller as it is
    size_t used = 0, cap = 0;
    std::vector<size_t> off;        // off[i] = first code of record i; a final entry is appended when the part ends
    size_t raw = 0;
    bool want_names = false;        // readFastaCodesPartNamed: the header text of every record, back to back
    std::string names;
    std::vector<size_t> name_off;
    void reserve(size_t n) {
        if (used + n <= cap) return;
        cap = std::max(used + n + (4u << 20), cap + cap / 2);
        codes = (char*)xrealloc(codes, cap);
    }
    void line(const char* p, size_t n) {
        reserve(n + 16);
        char* out = codes + used;
        size_t k = 0, i = 0;
#if defined(__SSE2__)
        // a line of upper-case letters only (nearly every line of a protein database): 16 residues per step, code = letter - 'A';
        // anything else in the line sends the whole line through the table below
        const __m128i a = _mm_set1_epi8('A'), bias = _mm_set1_epi8((char)-128), top = _mm_set1_epi8((char)(25 - 128));
        __m128i bad = _mm_setzero_si128();
        for (; i + 16 <= n; i += 16) {
            const __m128i t = _mm_sub_epi8(_mm_loadu_si128((const __m128i*)(p + i)), a);
            bad = _mm_or_si128(bad, _mm_cmpgt_epi8(_mm_add_epi8(t, bias), top));       // unsigned t > 25
            _mm_storeu_si128((__m128i*)(out + i), t);
        }
        if (_mm_movemask_epi8(bad)) i = 0; else k = i;
#endif
        for (; i < n; ++i) { const unsigned char c = kResidue.code[(unsigned char)p[i]]; out[k] = (char)c; k += (c != 0xFF); }
        used += k;
        raw += n;
    }
};

// Reads records until >= max_bytes of sequence were consumed (0 = unlimited) or EOF. Returns 1 if more remain.
// Block reads (4 MB) scanned with memchr; a record's residues go through the lookup table once, straight into the
// chain's own storage (121 MB of UniProt-style FASTA: 1.1 s with the former line-by-line reader).
int fasta_read(FastaState* st, Chain*** chains, int* len, size_t max_bytes, FlatSink* flat = nullptr) {
    int cap = chains ? *len : 0;
    size_t consumed = 0;
    if (st->buf.empty()) st->buf.resize(4u << 20);
    if (flat && st->have_pending) flat->off.push_back(flat->used);       // the record whose header closed the part before
    auto flush = [&]() {
        if (!st->have_pending) return;
        if (flat) {                              // the record's codes are in place already; it only has to be closed
            if (flat->want_names) { flat->name_off.push_back(flat->names.size()); flat->names.append(st->pending_name); }
            consumed += flat->raw;
            flat->raw = 0;
            st->have_pending = false;
            return;
        }
        ChainData* d = chain_data_alloc(st->pending_name.data(), st->pending_name.size(), st->seq.size());
        chain_data_append(d, st->seq.data(), st->seq.size());
        d->chars[d->length] = 0;
        push_chain(chains, len, &cap, make_chain(d, 0, d->length, false, chain_home(d)));
        if (st->cache_out) {
            ++st->cache_records;
            st->cache_residues += (uint64_t)d->length;
            const uint32_t hdr[3] = {(uint32_t)st->pending_name.size(), (uint32_t)d->length, (uint32_t)st->seq.size()};
            if (fwrite(hdr, sizeof hdr, 1, st->cache_out) != 1 ||
                (hdr[0] && fwrite(st->pending_name.data(), hdr[0], 1, st->cache_out) != 1) ||
                (hdr[1] && fwrite(d->chars, hdr[1], 1, st->cache_out) != 1))
                cache_abandon(st);            // disk full, read-only directory ...: the run goes on without a cache
        }
        consumed += st->seq.size();
        st->seq.clear();
        st->have_pending = false;
    };
    bool stop = false;
    while (!stop) {
        if (st->pos == st->end) {
            st->end = fread(st->buf.data(), 1, st->buf.size(), st->fh);
            st->pos = 0;
            if (st->end == 0) {
                SW_ASSERT(!ferror(st->fh), "read error in FASTA input");      // a short database must not pass for a complete one
                st->eof = true;
                break;
            }
        }
        const char* p = st->buf.data() + st->pos;
        const char* e = st->buf.data() + st->end;
        if (st->at_line_start && !st->in_header && *p == '>') {        // a new record begins
            flush();
            st->pending_name.clear();
            st->in_header = true;
            st->at_line_start = false;
            ++st->pos;
            continue;
        }
        const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
        const char* seg_end = nl ? nl : e;
        if (st->in_header) {
            st->pending_name.append(p, (size_t)(seg_end - p));
            if (nl) {
                while (!st->pending_name.empty() && st->pending_name.back() == '\r') st->pending_name.pop_back();
                st->in_header = false;
                st->have_pending = true;
                if (max_bytes && consumed >= max_bytes) stop = true;       // this header opens the next part
                else if (flat) flat->off.push_back(flat->used);
            }
        } else if (st->have_pending) {
            size_t n = (size_t)(seg_end - p);
            if (nl && n && seg_end[-1] == '\r') --n;
            if (flat) flat->line(p, n); else st->seq.insert(st->seq.end(), p, p + n);
        }
        st->pos = nl ? (size_t)(nl + 1 - st->buf.data()) : st->end;
        st->at_line_start = nl != NULL;
    }
    if (st->eof) {
        if (st->in_header) {          // file ends inside a header line: a record without residues
            while (!st->pending_name.empty() && st->pending_name.back() == '\r') st->pending_name.pop_back();
            st->in_header = false;
            st->have_pending = true;
            if (flat) flat->off.push_back(flat->used);
        }
        flush();
        if (st->cache_out) {              // the whole file has been read: totals into the header, then the cache becomes visible under its final name
            st->cache_header.records = st->cache_records;
            st->cache_header.residues = st->cache_residues;
            bool ok = fseek(st->cache_out, 0, SEEK_SET) == 0 && fwrite(&st->cache_header, sizeof(CacheHeader), 1, st->cache_out) == 1;
            ok = (fclose(st->cache_out) == 0) && ok;
            st->cache_out = NULL;
            if (!ok || rename(st->cache_tmp.c_str(), st->cache_final.c_str()) != 0) remove(st->cache_tmp.c_str());
        }
    }
    if (chains && *len < cap) *chains = (Chain**)xrealloc(*chains, sizeof(Chain*) * (size_t)(*len ? *len : 1));
    return st->eof ? 0 : 1;
}

// The same from the serialized cache: records until >= max_bytes of sequence were consumed (0 = unlimited) or EOF.
int cache_read(FastaState* st, Chain*** chains, int* len, size_t max_bytes, FlatSink* flat = nullptr) {
    int cap = chains ? *len : 0;
    size_t consumed = 0;
    std::string name;
    for (;;) {
        uint32_t hdr[3];                                   // name length, residues, raw sequence bytes of the text record
        const size_t got = fread(hdr, 1, sizeof hdr, st->fh);
        if (got == 0) { st->eof = true; break; }
        SW_ASSERT(got == sizeof hdr, "truncated serialized database cache");
        name.resize(hdr[0]);
        SW_ASSERT(!hdr[0] || fread(&name[0], hdr[0], 1, st->fh) == 1, "truncated serialized database cache");
        if (flat) {
            const size_t at = flat->used;
            flat->off.push_back(at);
            if (flat->want_names) { flat->name_off.push_back(flat->names.size()); flat->names.append(name); }
            flat->reserve(hdr[1]);
            SW_ASSERT(!hdr[1] || fread(flat->codes + at, hdr[1], 1, st->fh) == 1, "truncated serialized database cache");
            for (uint32_t i = 0; i < hdr[1]; ++i) flat->codes[at + i] = (char)(flat->codes[at + i] - 'A');
            flat->used = at + hdr[1];
            consumed += hdr[2];
            if (max_bytes && consumed >= max_bytes) {
                const int c = fgetc(st->fh);
                if (c == EOF) st->eof = true; else ungetc(c, st->fh);
                break;
            }
            continue;
        }
        ChainData* d = chain_data_alloc(name.data(), name.size(), hdr[1]);
        SW_ASSERT(!hdr[1] || fread(d->chars, hdr[1], 1, st->fh) == 1, "truncated serialized database cache");
        for (uint32_t i = 0; i < hdr[1]; ++i) d->codes[i] = (char)(d->chars[i] - 'A');
        d->length = (int)hdr[1];
        d->chars[d->length] = 0;
        push_chain(chains, len, &cap, make_chain(d, 0, d->length, false, chain_home(d)));
        consumed += hdr[2];
        if (max_bytes && consumed >= max_bytes) {           // as the text reader: the record that reaches the limit ends the part --
            const int c = fgetc(st->fh);                    // unless it was the last one
            if (c == EOF) st->eof = true; else ungetc(c, st->fh);
            break;
        }
    }
    if (chains && *len < cap) *chains = (Chain**)xrealloc(*chains, sizeof(Chain*) * (size_t)(*len ? *len : 1));
    return st->eof ? 0 : 1;
}

}  // namespace

SW_API void readFastaChainsPartInit(Chain*** chains, int* chainsLen, FILE** handle, int* serialized, const char* path) {
    *chains = NULL;
    *chainsLen = 0;
    *serialized = 0;
    const bool cache = fasta_cache_enabled();
    if (cache && (*handle = open_valid_cache(path)) != NULL) {       // a cache made from this very file: read that
        *serialized = 1;
        setvbuf(*handle, NULL, _IOFBF, 4u << 20);
        fasta_forget(*handle);
        fasta_state(*handle, true)->serialized_in = true;
        return;
    }
    *handle = fopen(path, "r");
    SW_ASSERT(*handle != NULL, "cannot open file '%s'", path);
    fasta_forget(*handle);    // a caller that closed an earlier handle before EOF may get the same FILE* again
    FastaState* st = fasta_state(*handle, true);
    uint64_t size; int64_t mtime;
    if (cache && stat_file(path, &size, &mtime)) {                   // this pass writes the cache (if the directory lets it)
        st->cache_final = std::string(path) + ".s4gchains";
        st->cache_tmp = st->cache_final + ".tmp" + std::to_string((long long)getpid());
        st->cache_out = fopen(st->cache_tmp.c_str(), "wb");
        if (st->cache_out) {
            CacheHeader& h = st->cache_header;
            memcpy(h.magic, kCacheMagic, 8); h.fasta_size = size; h.fasta_mtime = mtime; h.records = h.residues = 0;
            if (fwrite(&h, sizeof h, 1, st->cache_out) != 1) cache_abandon(st);
        }
    }
}

SW_API int readFastaChainsPart(Chain*** chains, int* chainsLen, FILE* handle, int serialized, const size_t maxBytes) {
    FastaState* st = fasta_state(handle, true);
    SW_ASSERT((serialized != 0) == st->serialized_in, "readFastaChainsPart: `serialized` does not belong to this handle");
    int more = st->serialized_in ? cache_read(st, chains, chainsLen, maxBytes) : fasta_read(st, chains, chainsLen, maxBytes);
    if (!more) fasta_forget(handle);   // caller fclose()s the handle itself (database_alignment.cpp:122)
    return more;
}

// Extension: the next part of the file as residue codes only -- no Chain objects, no names -- for callers that hand the
// sequences straight to the device (the k-mer prefilter of dropin/device_database_search.cpp).  Same records, same residues
// (chainGetCodes of the chains readFastaChainsPart would have returned, back to back) and the same part boundaries.  *codes and
// *offsets (records + 1 entries, offsets[0] = 0) are malloc'ed; the caller frees them.  Returns 1 if more parts remain.  A handle
// is read either with this function or with readFastaChainsPart, not both.
static int read_codes_part(char** codes, size_t** offsets, char** names, size_t** nameOffsets, int* recordsLen, FILE* handle, int serialized,
                           const size_t maxBytes) {
    FastaState* st = fasta_state(handle, true);
    SW_ASSERT((serialized != 0) == st->serialized_in, "readFastaCodesPart: `serialized` does not belong to this handle");
    cache_abandon(st);          // a pass that builds no chains cannot write the serialized cache (S4G_FASTA_CACHE=1): none is left behind
    FlatSink flat;
    flat.want_names = names != NULL;
    const int more = st->serialized_in ? cache_read(st, NULL, NULL, maxBytes, &flat) : fasta_read(st, NULL, NULL, maxBytes, &flat);
    flat.off.push_back(flat.used);
    *recordsLen = (int)flat.off.size() - 1;
    flat.reserve(1);
    *codes = flat.codes;
    *offsets = (size_t*)xmalloc(sizeof(size_t) * flat.off.size());
    memcpy(*offsets, flat.off.data(), sizeof(size_t) * flat.off.size());
    if (names) {
        SW_ASSERT((int)flat.name_off.size() == *recordsLen, "internal error: %zu names for %d records", flat.name_off.size(), *recordsLen);
        flat.name_off.push_back(flat.names.size());
        *names = (char*)xmalloc(flat.names.size() + 1);
        memcpy(*names, flat.names.data(), flat.names.size());
        *nameOffsets = (size_t*)xmalloc(sizeof(size_t) * flat.name_off.size());
        memcpy(*nameOffsets, flat.name_off.data(), sizeof(size_t) * flat.name_off.size());
    }
    if (!more) fasta_forget(handle);
    return more;
}

SW_API int readFastaCodesPart(char** codes, size_t** offsets, int* recordsLen, FILE* handle, int serialized, const size_t maxBytes) {
    return read_codes_part(codes, offsets, NULL, NULL, recordsLen, handle, serialized, maxBytes);
}

// The same with the header text of every record (names[nameOffsets[i] .. nameOffsets[i + 1]), not terminated): everything a
// Chain is made of, so that a caller can keep a part flat and create the few chains it turns out to need (chainDatabaseCreateFlat).
SW_API int readFastaCodesPartNamed(char** codes, size_t** offsets, char** names, size_t** nameOffsets, int* recordsLen, FILE* handle,
                                   int serialized, const size_t maxBytes) {
    SW_ASSERT(names != NULL && nameOffsets != NULL, "readFastaCodesPartNamed: NULL argument");
    return read_codes_part(codes, offsets, names, nameOffsets, recordsLen, handle, serialized, maxBytes);
}

// Totals of a FASTA file: records and residues (as chainGetLength counts them).  With a valid serialized cache beside the
// file (S4G_FASTA_CACHE=1) they come out of its header; otherwise the file is read once, part by part (which writes the cache
// if caching is on).  What the full-scan build's "prefilter" stage needs from the database (total residues for the e-value
// statistics, main.cpp:204,210) and nothing else.
SW_API void fastaResidueCount(const char* path, long long* records, long long* residues) {
    CacheHeader h;
    if (fasta_cache_enabled()) {
        FILE* fh = open_valid_cache(path, &h);
        if (fh) {
            fclose(fh);
            if (records) *records = (long long)h.records;
            if (residues) *residues = (long long)h.residues;
            return;
        }
    }
    Chain** chains = NULL; int n = 0; FILE* handle = NULL; int serialized = 0;
    readFastaChainsPartInit(&chains, &n, &handle, &serialized, path);
    long long cells = 0;
    int start = 0;
    for (;;) {
        const int more = readFastaChainsPart(&chains, &n, handle, serialized, 250000000);
        for (int i = start; i < n; ++i) { cells += chainGetLength(chains[i]); chainDelete(chains[i]); chains[i] = NULL; }
        start = n;
        if (!more) break;
    }
    fclose(handle);
    if (records) *records = n;
    if (residues) *residues = cells;
    deleteFastaChains(chains, n);
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
    SW_ASSERT(s4g::matrix_by_name(s->name.c_str(), s->table), "unknown similarity matrix '%s' (available: %s)",
              s->name.c_str(), s4g::kMatrixNames);
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
    *cards = (int*)xmalloc(sizeof(int) * (n ? n : 1));
    for (int i = 0; i < n; ++i) (*cards)[i] = i;
    *cardsLen = n;
}

// sift4g calls this once at start-up (main.cpp:186), a few hundred milliseconds of FASTA reading and prefiltering before it
// aligns anything: the cards it names (card 0 when it names none) start warming up underneath that work (s4g_warmup).
static std::atomic<int> g_first_card{-1};      // the first card a successful cudaCheckCards was given (-1: none yet)

SW_API int cudaCheckCards(int* cards, int cardsLen) {
    const int ok = s4g_device_check(cards, cardsLen);
    if (ok) {
        if (cardsLen <= 0) (void)s4g_warmup(0);
        for (int i = 0; i < cardsLen; ++i) (void)s4g_warmup(cards[i]);
        int none = -1;
        g_first_card.compare_exchange_strong(none, cardsLen > 0 ? cards[0] : 0);
    }
    return ok;
}

// Extension: the first card the caller validated with cudaCheckCards (main.cpp:183-186 does so with its --cards list before any
// stage runs), 0 if it never did.  Stages that are not handed a card list -- searchDatabase, database_search.hpp: no `cards`
// argument -- place their device work there instead of on card 0 whatever the user asked for.
SW_API int swsharpPreferredCard(void) {
    const int c = g_first_card.load();
    return c < 0 ? 0 : c;
}

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

thread_local bool t_pool_worker = false;

void* pool_worker(void*) {
    t_pool_worker = true;
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

// true on a thread of the pool: a task must not wait there for sub-tasks it submits (a pool whose workers all did would
// stand still), so library code that fans work out checks this first
bool swimpl::on_pool_thread() { return t_pool_worker; }
int swimpl::pool_threads() { pthread_mutex_lock(&g_pool_mu); const int n = g_pool_stop ? 0 : (int)g_workers.size(); pthread_mutex_unlock(&g_pool_mu); return n; }

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

// s4g_stream.hip -- C ABI: a packed database file (s4g_db_save) scanned WITHOUT being resident: s4g_db_open_streamed.
//
// The reference reads its database part by part while it works on it (readFastaChainsPart in 1 GB parts,
// /root/reference/sift4g/src/database_search.cpp:80-97, database_alignment.cpp:36-47); BASELINE configs[3] is a database of
// that kind at UniRef90 scale: 150 M sequences, ~50 GB as packed tiles, a share of 6.4 GB per GPU -- which fits HBM, but has to
// get there, and a database beyond HBM never fits.  A streamed database keeps only the tile headers on the host and two tile
// slots on the device; s4g_scan_topk walks the file tile by tile: while the kernels of tile i run, the host reads tile i + 1
// into pinned memory and queues its upload on a copy stream (events order slot reuse).  The scan itself -- plan, kernels, tile
// top-k, merge into the running lists -- is the resident one's (s4g_scan.hip, scan_topk_impl), so are its results.
#include "s4g_internal.hpp"

#include <sys/stat.h>

namespace {
constexpr char kDbMagicStream[8] = {'S', '4', 'G', 'D', 'B', '0', '0', '2'};      // s4g_core.hip kDbMagic
struct TileHeaderS { uint32_t n_seqs, n_groups, max_len4, pad; uint64_t residues, padded_residues, pos0, dwords; };
inline uint64_t payload_bytes(const TileHeaderS& h) {
    return (uint64_t)h.n_groups * 8 + (uint64_t)h.n_groups * 4 + 2 * (uint64_t)h.n_groups * 64 * 4 + (uint64_t)h.n_seqs * 4 + h.dwords * 4;
}
double now_sec() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

struct DbStream {
    FILE* f = nullptr;
    std::vector<uint64_t> file_off, bytes;      // per tile: where its payload starts in the file, how long it is
    uint64_t slot_bytes = 0;
    char* h_pin[2] = {nullptr, nullptr};
    char* d_slot[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    hipEvent_t uploaded[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
    bool slot_used[2] = {false, false};
    // statistics of the scans so far
    double bytes_read = 0, read_s = 0, wait_s = 0, scan_wall_s = 0;
    uint64_t tiles_loaded = 0;
};

void db_stream_close(s4g_db* db) {
    DbStream* s = db->stream;
    if (!s) return;
    if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
    for (int k = 0; k < 2; ++k) {
        if (s->h_pin[k]) (void)hipHostFree(s->h_pin[k]);
        if (s->d_slot[k]) (void)hipFree(s->d_slot[k]);
        if (s->uploaded[k]) (void)hipEventDestroy(s->uploaded[k]);
        if (s->consumed[k]) (void)hipEventDestroy(s->consumed[k]);
    }
    if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
    if (s->f) fclose(s->f);
    delete s;
    db->stream = nullptr;
}

S4G_API int s4g_db_open_streamed(s4g_db** out, int device, const char* path) {
    if (!out || !path) return fail(S4G_ERR_INVALID, "NULL argument");
    int have = device_count_raw();
    if (have <= 0) return fail(S4G_ERR_NO_DEVICE, "no HIP device available (libs4g_hip has no CPU fallback)");
    if (device < 0 || device >= have) return fail(S4G_ERR_INVALID, "device %d out of range [0,%d)", device, have);
    FILE* f = fopen(path, "rb");
    if (!f) return fail(S4G_ERR_INVALID, "cannot open '%s'", path);
    DeviceGuard guard(device);
    s4g_db* db = new s4g_db();
    DbStream* s = new DbStream();
    db->device = device; db->stream = s; s->f = f;
    auto bad = [&](const char* what) { db_stream_close(db); delete db; return fail(S4G_ERR_INVALID, "'%s' is not a packed database: %s", path, what); };
    char magic[8];
    uint32_t nt = 0;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, kDbMagicStream, 8) != 0) return bad("magic");
    if (fread(&db->n_seqs, 4, 1, f) != 1 || fread(&nt, 4, 1, f) != 1 || fread(&db->residues, 8, 1, f) != 1 || fread(&db->padded_residues, 8, 1, f) != 1)
        return bad("header");
    db->len.resize(db->n_seqs); db->pos_of_id.resize(db->n_seqs);
    if ((db->n_seqs && fread(db->len.data(), 4, db->n_seqs, f) != db->n_seqs) || (db->n_seqs && fread(db->pos_of_id.data(), 4, db->n_seqs, f) != db->n_seqs))
        return bad("length tables");
    struct stat st;
    if (fstat(fileno(f), &st) != 0) return bad("cannot stat the file");
    uint64_t seqs_seen = 0, padded_seen = 0, residues_seen = 0;
    for (uint32_t i = 0; i < nt; ++i) {
        TileHeaderS h;
        if (fread(&h, sizeof h, 1, f) != 1) return bad("tile header");
        // the header's numbers against each other (what s4g_db_save guarantees): device pointers are derived from them
        // (every product is bounded before it is formed: crafted fields must not wrap a uint64 into a plausible size)
        const uint64_t fsize = (uint64_t)st.st_size;
        if (h.dwords > fsize / 4 || h.n_seqs > fsize / 4 || h.padded_residues > fsize || h.n_groups > fsize / 8)
            return bad("a tile header names more data than the file holds");
        unsigned __int128 lanes = (unsigned __int128)h.n_groups * 64u * h.max_len4;
        if (h.n_groups != (h.n_seqs + 63) / 64 || h.dwords * 4 != h.padded_residues || h.residues > h.padded_residues ||
            (unsigned __int128)h.dwords > lanes || h.pos0 != seqs_seen)
            return bad("inconsistent tile header");
        seqs_seen += h.n_seqs; padded_seen += h.padded_residues; residues_seen += h.residues;
        db->tiles.emplace_back();
        Tile& t = db->tiles.back();
        t.n_seqs = h.n_seqs; t.n_groups = h.n_groups; t.max_len4 = h.max_len4; t.residues = h.residues;
        t.padded_residues = h.padded_residues; t.pos0 = h.pos0;
        const uint64_t b = payload_bytes(h);
        s->file_off.push_back((uint64_t)ftello(f));
        s->bytes.push_back(b);
        s->slot_bytes = std::max(s->slot_bytes, b);
        db->max_tile_seqs = std::max(db->max_tile_seqs, t.n_seqs);
        if (b > fsize || s->file_off.back() > fsize - b) return bad("file shorter than its tiles (truncated?)");     // every tile, not only the last
        if (fseeko(f, (off_t)b, SEEK_CUR) != 0) return bad("tile payload");
    }
    if (seqs_seen != db->n_seqs || padded_seen != db->padded_residues || residues_seen != db->residues) return bad("tile headers do not add up to the file header");
    s->slot_bytes = (s->slot_bytes + 4095) / 4096 * 4096;
    for (int k = 0; k < 2 && s->slot_bytes; ++k) {
        if (hipHostMalloc((void**)&s->h_pin[k], s->slot_bytes, hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void**)&s->d_slot[k], s->slot_bytes) != hipSuccess ||
            hipEventCreateWithFlags(&s->uploaded[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->consumed[k], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            db_stream_close(db); delete db;
            return fail(S4G_ERR_NOMEM, "cannot allocate two tile slots of %llu bytes (pinned + device)", (unsigned long long)s->slot_bytes);
        }
    }
    if (hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking) != hipSuccess) { db_stream_close(db); delete db; return fail(S4G_ERR_HIP, "hipStreamCreate failed"); }
    db->hbm_bytes = 2 * s->slot_bytes;
    *out = db;
    return S4G_OK;
}

// Reads tile ti into pinned slot ti & 1 and queues its upload (host-blocking for the read only).  The slot's previous tenant must
// have been consumed: the caller has recorded `consumed` behind that tile's kernels.
int db_stream_load(const s4g_db* db, size_t ti) {
    DbStream* s = db->stream;
    const int k = (int)(ti & 1);
    const double t0 = now_sec();
    if (s->slot_used[k]) HIP_TRY(hipEventSynchronize(s->consumed[k]));      // tile ti - 2 is done with the slot (and its upload long before)
    const double t1 = now_sec();
    if (fseeko(s->f, (off_t)s->file_off[ti], SEEK_SET) != 0 || fread(s->h_pin[k], 1, s->bytes[ti], s->f) != s->bytes[ti])
        return fail(S4G_ERR_HIP, "reading tile %zu of the packed database failed", ti);
    const double t2 = now_sec();
    {   // the group tables of the payload against the header (the kernels index the slot with them)
        const Tile& m = db->tiles[ti];
        const uint64_t* goff = reinterpret_cast<const uint64_t*>(s->h_pin[k]);
        const uint32_t* glen = reinterpret_cast<const uint32_t*>(s->h_pin[k] + (uint64_t)m.n_groups * 8);
        const uint64_t dwords = m.padded_residues / 4;
        for (uint32_t g = 0; g < m.n_groups; ++g)
            if (glen[g] > m.max_len4 || goff[g] > dwords || (uint64_t)glen[g] * 64 > dwords - goff[g])
                return fail(S4G_ERR_INVALID, "tile %zu of the packed database is corrupt (group %u lies outside the tile)", ti, g);
        // the two permutations are WRITE indices of the scan kernels (global: a row of the score matrix, n_seqs wide; local: a row of
        // the tile buffer, the tile's n_seqs wide); 0xFFFFFFFF marks an empty lane
        const uint32_t* pglob = glen + m.n_groups;
        const uint32_t* ploc = pglob + (uint64_t)m.n_groups * 64;
        const uint32_t* cols = ploc + (uint64_t)m.n_groups * 64;
        for (uint64_t i = 0, n = (uint64_t)m.n_groups * 64; i < n; ++i)
            if ((pglob[i] != 0xFFFFFFFFu && pglob[i] >= db->n_seqs) || (ploc[i] != 0xFFFFFFFFu && ploc[i] >= m.n_seqs))
                return fail(S4G_ERR_INVALID, "tile %zu of the packed database is corrupt (lane %llu maps outside the score row)", ti, (unsigned long long)i);
        for (uint32_t i = 0; i < m.n_seqs; ++i)
            if (cols[i] >= db->n_seqs) return fail(S4G_ERR_INVALID, "tile %zu of the packed database is corrupt (column %u names sequence %u of %u)", ti, i, cols[i], db->n_seqs);
    }
    HIP_TRY(hipMemcpyAsync(s->d_slot[k], s->h_pin[k], s->bytes[ti], hipMemcpyHostToDevice, s->copy_stream));
    HIP_TRY(hipEventRecord(s->uploaded[k], s->copy_stream));
    s->slot_used[k] = true;
    s->wait_s += t1 - t0; s->read_s += t2 - t1; s->bytes_read += (double)s->bytes[ti]; ++s->tiles_loaded;
    return S4G_OK;
}

// The tile as the kernels want it: the header's numbers and pointers into its slot (valid once `stream` has passed the upload).
int db_stream_tile(const s4g_db* db, size_t ti, hipStream_t stream, Tile* out) {
    DbStream* s = db->stream;
    const int k = (int)(ti & 1);
    HIP_TRY(hipStreamWaitEvent(stream, s->uploaded[k], 0));
    const Tile& m = db->tiles[ti];
    *out = Tile();
    out->n_seqs = m.n_seqs; out->n_groups = m.n_groups; out->max_len4 = m.max_len4; out->residues = m.residues;
    out->padded_residues = m.padded_residues; out->pos0 = m.pos0;
    char* p = s->d_slot[k];                                               // the payload's order in the file (s4g_db_save)
    out->d_grp_off = (uint64_t*)p; p += (uint64_t)m.n_groups * 8;
    out->d_grp_len4 = (uint32_t*)p; p += (uint64_t)m.n_groups * 4;
    out->d_perm_global = (uint32_t*)p; p += (uint64_t)m.n_groups * 64 * 4;
    out->d_perm_local = (uint32_t*)p; p += (uint64_t)m.n_groups * 64 * 4;
    out->d_col_ids = (uint32_t*)p; p += (uint64_t)m.n_seqs * 4;
    out->d_res = (uint32_t*)p;
    return S4G_OK;
}

int db_stream_consumed(const s4g_db* db, size_t ti, hipStream_t stream) {
    HIP_TRY(hipEventRecord(db->stream->consumed[ti & 1], stream));
    return S4G_OK;
}

void db_stream_note_scan(const s4g_db* db, double wall_s) { db->stream->scan_wall_s += wall_s; }

// out[0] = bytes read from the file by the scans so far, out[1] = seconds spent in the reads, out[2] = seconds the reader waited
// for a slot (the device was the slower side), out[3] = tiles loaded, out[4] = wall seconds of the scans, out[5] = bytes of one
// tile slot.  cap >= 6.
S4G_API int s4g_db_stream_stats(const s4g_db* db, double* out, uint32_t cap) {
    if (!db || !out || cap < 6) return fail(S4G_ERR_INVALID, "need a streamed database and room for 6 values");
    if (!db->stream) return fail(S4G_ERR_INVALID, "not a streamed database (s4g_db_open_streamed)");
    const DbStream* s = db->stream;
    out[0] = s->bytes_read; out[1] = s->read_s; out[2] = s->wait_s; out[3] = (double)s->tiles_loaded; out[4] = s->scan_wall_s; out[5] = (double)s->slot_bytes;
    return S4G_OK;
}

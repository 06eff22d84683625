/*
 * s4g_hip.h -- C ABI of libs4g_hip.so, the MI355X (gfx950) Smith-Waterman
 * database scorer that replaces the scoring step of swsharp's alignDatabase
 * for sift4g.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  Every entry point
 * returns S4G_OK (0) or a negative error code; s4g_last_error() gives the
 * text.  The library has NO CPU fallback: without a usable HIP device every
 * device entry point returns S4G_ERR_NO_DEVICE.
 *
 * Residue codes are letter - 'A' (0..25), the alphabet sift4g relies on
 * (/root/reference/sift4g/src/hash.cpp:16-19).  Sequences are passed as one
 * concatenated code array plus n+1 offsets (CSR style).
 *
 * Reference interface each entry point replaces (the swsharp source itself is
 * an absent submodule; signatures are taken from sift4g's call sites):
 *
 *   s4g_device_count / s4g_device_check
 *       cudaGetCards(int**, int*) / cudaCheckCards(int*, int)
 *       /root/reference/sift4g/src/main.cpp:183-186
 *   s4g_scorer_create / s4g_scorer_destroy
 *       scorerCreateMatrix(Scorer**, char*, int, int) / scorerDelete
 *       /root/reference/sift4g/src/main.cpp:208,223
 *   s4g_db_create / s4g_db_create_tiled / s4g_db_create_scattered / s4g_db_create_cached / s4g_chain_cache_info / s4g_db_create_from_device / s4g_db_open_streamed / s4g_db_stream_stats / s4g_db_destroy / s4g_db_info / s4g_db_tiles / s4g_db_save / s4g_db_load
 *       chainDatabaseCreate(Chain**, int, int, int* cards, int cardsLen) /
 *       chainDatabaseDelete
 *       /root/reference/sift4g/src/database_alignment.cpp:80,92
 *   s4g_queries_create / s4g_queries_destroy
 *       the `Chain* query` argument of alignDatabase
 *       /root/reference/sift4g/src/database_alignment.cpp:83-86
 *   s4g_scan / s4g_scan_flags / s4g_scan_host / s4g_scan_host_flags / s4g_score_lists_host / s4g_score_pairs_host
 *       step 1 of swsharp alignDatabase(...): the optimal local affine-gap
 *       score of the query against every target of the ChainDatabase
 *       /root/reference/sift4g/src/database_alignment.cpp:83-86
 *       (s4g_score_lists_host = the per-query candidate lists built by
 *       createFilteredDatabase, database_alignment.cpp:136-162)
 *   s4g_traceback_host / s4g_traceback_host_flags / s4g_traceback_submit / s4g_traceback_finish
 *       step 4 of swsharp alignDatabase: traceback of the survivors
 *   s4g_warmup
 *       the device initialisation swsharp does behind cudaCheckCards / its first alignDatabase call, main.cpp:186
 *   s4g_scan_stats / s4g_scan_clock / s4g_clock_probe / s4g_scan_plan / s4g_scan_cells
 *       measurement only (bench.py); no counterpart in the reference, which prints no throughput figures
 *   s4g_topk / s4g_topk_host / s4g_scan_topk / s4g_scan_topk_host
 *       steps 2-3 of swsharp alignDatabase: keep the best maxAlignments
 *       targets per query (database_alignment.cpp:83-86 `max_alignments`),
 *       and the merge semantics of dbAlignmentsMerge (:101-102)
 *   s4g_comm_unique_id / s4g_comm_create / s4g_comm_info / s4g_comm_destroy / s4g_gather_topk / s4g_merge_topk / s4g_merge_topk_host
 *       dbAlignmentsMerge across database chunks (/root/reference/sift4g/src/database_alignment.cpp:97-104) applied across the
 *       GPUs of a node, one process per GPU: the reference runs its cards as threads of one process (main.cpp:254-262 hands the
 *       `cards` list to alignDatabase) and merges on the host; here every rank keeps its lists on its device and one RCCL
 *       all-gather over xGMI + a merge on the device give every rank the final lists
 */
#ifndef S4G_HIP_H
#define S4G_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S4G_OK               0
#define S4G_ERR_NO_DEVICE   -1   /* no HIP device / HIP runtime error at init */
#define S4G_ERR_INVALID     -2   /* bad argument */
#define S4G_ERR_HIP         -3   /* HIP runtime call failed */
#define S4G_ERR_NOMEM       -4

#define S4G_ALPHABET        26   /* letter - 'A' */

typedef struct s4g_scorer  s4g_scorer;   /* substitution table + gap penalties (host object) */
typedef struct s4g_db      s4g_db;       /* length-sorted, lane-transposed database resident in HBM */
typedef struct s4g_queries s4g_queries;  /* query set resident in HBM */

/* text of the last error on the calling thread ("" if none) */
const char* s4g_last_error(void);

/* library/ABI version, bumped on any signature change */
int s4g_abi_version(void);

/* number of usable HIP devices; S4G_ERR_NO_DEVICE if the runtime cannot start */
int s4g_device_count(int* count);
/* 1 if every card id in cards[0..n) names a usable device (n == 0 passes), else 0 */
int s4g_device_check(const int* cards, int n);

/* Built-in substitution tables by the names scorerCreateMatrix accepts
 * (/root/reference/sift4g/src/main.cpp:73,208,304-314): BLOSUM_45, BLOSUM_50, BLOSUM_62, BLOSUM_80, BLOSUM_90, PAM_30,
 * PAM_70, PAM_250, and the help text's spellings BLOSUM_30 / BLOSUM_70 / BLOSUM_250 for the three PAM tables; any other
 * name returns S4G_ERR_INVALID.
 * table26x26 receives 676 int32, letter-indexed (code = letter - 'A'). */
int s4g_matrix_by_name(const char* name, int32_t* table26x26);

/* table: 26x26 row-major int32, table[a*26+b] = score(a, b); |entry| <= 127.
 * gap_open = cost of the first gap residue, gap_extend = cost of each further
 * one (a gap of k residues costs gap_open + (k-1)*gap_extend); 0 <= extend <= open. */
int s4g_scorer_create(s4g_scorer** out, const int32_t* table, int gap_open, int gap_extend);
int s4g_scorer_destroy(s4g_scorer* s);

/* Upload a database to `device`.  codes[offsets[i] .. offsets[i+1]) are the
 * residue codes of target i.  Host buffers may be freed on return. */
int s4g_db_create(s4g_db** out, int device, const uint8_t* codes, const uint64_t* offsets, uint32_t n_seqs);
/* Same with an explicit tile size (sequences per tile, 0 = default 2^20).  The database is always stored as tiles of
 * the length-sorted order; tiles bound the strip-boundary workspace and the s4g_scan_topk score buffer. */
int s4g_db_create_tiled(s4g_db** out, int device, const uint8_t* codes, const uint64_t* offsets, uint32_t n_seqs,
                        uint32_t max_tile_seqs);

/* The same for sequences held as separate host arrays (what a Chain*[] is: sift4g builds one candidate database per
 * query out of chains it already holds, /root/reference/sift4g/src/database_alignment.cpp:136-162 then :80): seqs[i]
 * points at the lens[i] residue codes of target i.  Saves the caller the concatenation. */
int s4g_db_create_scattered(s4g_db** out, int device, const uint8_t* const* seqs, const uint32_t* lens, uint32_t n_seqs);

/* A candidate database out of sequences the device may already hold: what the reference's wrapper builds per (query, chunk) --
 * createFilteredDatabase + chainDatabaseCreate + one alignDatabase + chainDatabaseDelete,
 * /root/reference/sift4g/src/database_alignment.cpp:70-92 -- out of chains that are candidates of many queries.  Every sequence
 * goes to the device ONCE, into a per-device arena; where it went is recorded in two words the CALLER keeps with the sequence:
 * tags[i] / addrs[i] are in/out, zero when the sequence is created, handed back unchanged on every later call (never copied
 * from one sequence to a different one).  The database is two small device tables and serves the one-wave-per-pair routes only:
 * s4g_score_pairs_host / s4g_score_lists_host (always one wave per pair), s4g_traceback_host; every s4g_scan* entry point and
 * s4g_db_save return S4G_ERR_INVALID for it.  The arena starts over beyond 16 GB (once no such database is alive); sequences
 * sent before then are simply sent again. */
int s4g_db_create_cached(s4g_db** out, int device, const uint8_t* const* seqs, const uint32_t* lens, uint64_t* tags,
                         uint64_t* addrs, uint32_t n_seqs);
/* out[0] = sequences the arena of `device` holds, out[1] = its bytes, out[2] / out[3] = sequences of s4g_db_create_cached calls
 * found there / sent, out[4] = arena generation, out[5] = databases created, out[6] = alive.  cap >= 7. */
int s4g_chain_cache_info(int device, uint64_t* out, uint32_t cap);

/* The same for residue codes and offsets that already live on `device` (a database generated, decompressed or
 * received there): only the n_seqs+1 offsets come back to the host (length sort, tile cuts); the codes are
 * transposed where they are.  d_codes / d_offsets stay the caller's and may be freed after the call. */
int s4g_db_create_from_device(s4g_db** out, int device, const uint8_t* d_codes, const uint64_t* d_offsets, uint32_t n_seqs,
                              uint32_t max_tile_seqs);
int s4g_db_destroy(s4g_db* db);
/* Packed on-disk format: the tiles exactly as they sit in HBM.  s4g_db_save writes a database created from
 * sequences; s4g_db_load places a saved one on `device` without parsing or packing. */
int s4g_db_save(const s4g_db* db, const char* path);
int s4g_db_load(s4g_db** out, int device, const char* path);
/* A saved database scanned WITHOUT being resident: only the tile headers are read here and two tile slots are allocated (pinned
 * host + device, the size of the largest tile).  s4g_scan_topk / s4g_scan_topk_host then walk the file tile by tile -- while the
 * kernels of one tile run, the host reads the next one and queues its upload -- with the resident scan's results; the other
 * entry points return S4G_ERR_INVALID for such a database.  This is the reference's own access pattern (readFastaChainsPart in
 * 1 GB parts while it works, /root/reference/sift4g/src/database_search.cpp:80-97) applied to the packed format: BASELINE configs[3],
 * or any database beyond HBM.  Queries whose score bound exceeds the int16 windows run in the int32 kernel (no speculative tier:
 * its exact recheck would need the tiles resident).
 * s4g_db_stream_stats: out[0] = bytes read from the file by the scans so far, out[1] = seconds inside the reads, out[2] = seconds
 * the reader waited for a free slot (the device was the slower side), out[3] = tiles loaded, out[4] = wall seconds of the scans,
 * out[5] = bytes of one tile slot.  cap >= 6. */
int s4g_db_open_streamed(s4g_db** out, int device, const char* path);
int s4g_db_stream_stats(const s4g_db* db, double* out, uint32_t cap);
/* number of tiles and (optionally, up to cap entries) sequences per tile */
int s4g_db_tiles(const s4g_db* db, uint32_t* n_tiles, uint32_t* tile_seqs, uint32_t cap);
/* any out pointer may be NULL */
int s4g_db_info(const s4g_db* db, uint32_t* n_seqs, uint64_t* residues, uint64_t* padded_residues, uint64_t* hbm_bytes);

int s4g_queries_create(s4g_queries** out, int device, const uint8_t* codes, const uint64_t* offsets, uint32_t n_queries);
int s4g_queries_destroy(s4g_queries* q);

/* DP cells of scanning queries [q_begin, q_begin+q_count) against the whole
 * database: sum over pairs of Lq*Lt (the GCUPS numerator). */
int s4g_scan_cells(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, uint64_t* cells);

/* Score queries [q_begin, q_begin+q_count) against every target.
 * d_scores: DEVICE pointer to q_count * n_seqs int32, row = query (relative to
 * q_begin), column = target in the ORIGINAL order given to s4g_db_create.
 * Asynchronous on `stream` (a hipStream_t, NULL = default stream).
 * Two device mappings serve it, both exact: the inter-sequence scan kernels (64 targets per wave, the throughput
 * form) and one wave per (query, target) pair (sw_pair.hip.h, the latency form).  Per database tile the library
 * estimates how many of the tile's longest groups would make a scan launch wait and gives those to pair waves on a
 * side stream: all of them for one query against a few thousand targets (the per-query alignDatabase call of
 * database_alignment.cpp:83-86), none for a thousand queries, the titin-sized end of the database in between. */
int s4g_scan(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count,
             const s4g_scorer* scorer, int32_t* d_scores, void* stream);

/* s4g_scan with options.  By default every query runs in the fastest kernel that is EXACT for it, decided from
 * its score bound sum_i max(0, max_b table[q_i][b]): packed int16 in a drifting frame (the column frame of sw_scan_pkc, the
 * default: bound + bias + columns * extend <= 31743, no row term; with S4G_SCAN_DIAG_FRAME the anti-diagonal frame of
 * sw_scan_pkd: bound + bias + (rows + columns) * extend <= 31743), packed int16 with v_pk_maximum3_f16 as integer max3 (bound + bias <= 31743), plain
 * packed int16 (bound <= 32767), else int32.  A query whose BOUND exceeds the drifting-frame window but whose frame fits it
 * with room to spare (long queries: the bound grows by ~5.6 per residue, real scores rarely do) still runs in that kernel,
 * speculatively: the kernel marks every result that came within one score step of the window's end, and an exact int32
 * pass (one wave per marked pair) recomputes just those before anything reads the scores; when no drifting frame fits, the
 * maximum3 tier does the same with its fixed window.  flags below. */
#define S4G_SCAN_FORCE_I32 1u
#define S4G_SCAN_NO_F16    2u   /* skip the v_pk_maximum3_f16 tier (those queries then run in the plain packed int16 kernel) */
#define S4G_SCAN_NO_DRIFT  8u   /* skip the drifting-frame kernel (sw_scan_pkd); those queries run in the maximum3 tier */
#define S4G_SCAN_NO_PAIRS  16u  /* scan kernels only: no group goes to the one-wave-per-pair kernels; any other flag implies this */
#define S4G_SCAN_NO_SPEC   32u  /* no speculative use of the drifting-frame kernel: tiers by score bound only */
#define S4G_SCAN_BLOCK_PIPE 64u /* every tile of sequences up to 4096 residues runs on the block-cooperative strip pipeline (sw_scan_lists in
                                   scan mode: the four waves of a block on one (group, query pair) item) whatever the query count -- queries
                                   whose bound is beyond the drifting-frame window speculatively, with the exact recheck behind them (not with
                                   S4G_SCAN_NO_SPEC: the tile then takes the general route); by default calls of 3 to 16 queries
                                   (S4G_SCAN_PIPE_MAX_QUERIES) against 16 384 sequences or more take it, and calls of one or two queries of
                                   500 residues or more against 65 536 sequences or more, or of any length against 196 608 or more (a
                                   query without a partner pairs two groups of the tile per item there) */
#define S4G_SCAN_DIAG_FRAME 128u /* the drifting-frame tier runs sw_scan_pkd (frame grows per anti-diagonal, rounds 1-4) instead of sw_scan_pkc
                                   (frame grows per column, round 5): the same results, kept as an independent second form.  The flag wins
                                   over the process-wide S4G_SCAN_FRAME variable, which only moves the default of calls that do not set it */
#define S4G_SCAN_F16_ARITH 4u   /* run the maximum3 tier in true f16 arithmetic (exact on bounds <= 4096) instead of the
                                   biased-int16 bit-pattern form (exact on bounds <= 0x7BFF - bias) */
int s4g_scan_flags(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count,
                   const s4g_scorer* scorer, int32_t* d_scores, void* stream, uint32_t flags);

/* Same, synchronous, scores copied to host memory h_scores[q_count * n_seqs]. */
int s4g_scan_host(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count,
                  const s4g_scorer* scorer, int32_t* h_scores);
int s4g_scan_host_flags(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count,
                        const s4g_scorer* scorer, int32_t* h_scores, uint32_t flags);

/* Candidate-list mode (what sift4g's wrapper does per query): query i is
 * scored against targets cand_ids[cand_offsets[i] .. cand_offsets[i+1]) only.
 * h_scores[cand_offsets[n_queries]] receives the scores in list order. */
int s4g_score_lists_host(const s4g_db* db, const s4g_queries* q, const uint64_t* cand_offsets,
                         const uint32_t* cand_ids, const s4g_scorer* scorer, int32_t* h_scores);

/* The same for an explicit pair list: h_scores[i] = score(query pair_q[i], target pair_t[i]).
 * This is the latency path of the per-query alignDatabase call
 * (/root/reference/sift4g/src/database_alignment.cpp:83-86 with `indexes` = that query's candidates):
 * every pair gets a wave of its own (anti-diagonal sweep, sw_pair.hip.h) instead of sharing a wave with 63
 * length-sorted neighbours, so a call is bounded by ~0.7 cycles per cell of its longest pair rather than ~26.
 * Queries beyond 1024 residues run band by band (1024 rows per band, the boundary row through HBM).  Candidate
 * lists that approach the whole database are served exactly by scanning the resident database and picking the
 * columns; flags force one route (tests, measurements). */
#define S4G_PAIRS_FORCE_WAVE 1u  /* one wave per pair */
#define S4G_PAIRS_FORCE_SCAN 2u  /* full scan + column pick */
#define S4G_PAIRS_FORCE_LISTS 4u /* every pair the drifting-frame window admits through the packed list kernel (sw_scan_lists: a query's
                                    candidates sorted by length on the device, gathered 128 to an item, two targets per register);
                                    by default only batches of >= 65536 pairs take it, and their longest targets stay with the waves */
#define S4G_PAIRS_LISTS_SPLIT(residues) (S4G_PAIRS_FORCE_LISTS | ((uint32_t)(residues) << 8))   /* the same with the cut chosen by the
                                    caller: targets longer than `residues` go to the pair waves (measurements) */
int s4g_score_pairs_host(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint64_t n_pairs,
                         const uint32_t* pair_q, const uint32_t* pair_t, int32_t* h_scores, uint32_t flags);

/* Per-query top-k of a device score matrix [q_count][n_seqs] (as written by
 * s4g_scan): the k highest scores with score >= min_score, ordered by score
 * descending then target id ascending (exact and deterministic; 1 <= k <= 4096).
 * ids / scores: q_count * k entries (unused tail filled with id 0xFFFFFFFF /
 * score 0); counts[q] = entries used.
 * s4g_topk: all pointers are DEVICE pointers, asynchronous on `stream` -- the
 * form the multi-GPU gather consumes.  s4g_topk_host: results in host memory. */
int s4g_topk(const s4g_db* db, const int32_t* d_scores, uint32_t q_count, uint32_t k, int32_t min_score,
             uint32_t* d_ids, int32_t* d_top_scores, uint32_t* d_counts, void* stream);
/* Streaming scan + running top-k for databases whose q_count x n_seqs score matrix does not fit (BASELINE
 * configs[3]): queries [q_begin, q_begin+q_count) against every tile of the database, keeping only the exact
 * per-query top-k (same order and tie rule as s4g_topk on the full matrix).  Outputs are DEVICE pointers
 * [q_count][k] / [q_count]; asynchronous on `stream`; flags as s4g_scan_flags. */
int s4g_scan_topk(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* scorer,
                  uint32_t k, int32_t min_score, uint32_t* d_ids, int32_t* d_top_scores, uint32_t* d_counts,
                  void* stream, uint32_t flags);
/* s4g_scan_topk with one score cut per query (h_min_scores[q_count] in host memory, NULL = 1 for all) and results in
 * host memory: a conservative integer score cut per query on the device, exact e-values on the survivors on the host. */
int s4g_scan_topk_host(const s4g_db* db, const s4g_queries* q, uint32_t q_begin, uint32_t q_count, const s4g_scorer* scorer,
                       uint32_t k, const int32_t* h_min_scores, uint32_t* h_ids, int32_t* h_top_scores, uint32_t* h_counts,
                       uint32_t flags);
int s4g_topk_host(const s4g_db* db, const int32_t* d_scores, uint32_t q_count, uint32_t k, int32_t min_score,
                  uint32_t* h_ids, int32_t* h_scores, uint32_t* h_counts, void* stream);

/* Traceback of surviving pairs on the device (step 4 of alignDatabase: the alignment paths that
 * /root/reference/sift4g/src/select_alignments.cpp:244-300 walks).  pair_score[i] must be the optimal local score of
 * (pair_q[i], pair_t[i]) as returned by the scan.  h_out[i*5 .. i*5+4] = q_start, q_end, t_start, t_end (0-based,
 * inclusive; -1 if the score is 0), path_len.  The path of pair i is written to h_paths[path_off[i] ..] with the
 * swsharp move codes 1 = DIAG, 2 = LEFT (gap in query), 3 = UP (gap in target), start to end; path_off has
 * n_pairs+1 entries and must leave at least (query length + target length) bytes per pair.  Synchronous. */
int s4g_traceback_host(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t n_pairs,
                       const uint32_t* pair_q, const uint32_t* pair_t, const int32_t* pair_score,
                       int32_t* h_out, const uint64_t* path_off, uint8_t* h_paths);
/* Same with a forced route: S4G_PAIRS_FORCE_SCAN = the 64-pairs-per-wave strip kernels (sw_trace_fill +
 * sw_trace_walk); 0 or S4G_PAIRS_FORCE_WAVE = one wave per pair (sw_pair_trace, band by band beyond 1024 query
 * residues), which is what s4g_traceback_host does. */
int s4g_traceback_host_flags(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t n_pairs,
                             const uint32_t* pair_q, const uint32_t* pair_t, const int32_t* pair_score,
                             int32_t* h_out, const uint64_t* path_off, uint8_t* h_paths, uint32_t flags);

/* The same traceback in two halves, so that a host can do other work -- the reference's wrapper builds the NEXT query's candidate
 * database, /root/reference/sift4g/src/database_alignment.cpp:70-92 -- while the device traces the survivors of this one.
 * s4g_traceback_submit queues the kernels and the download on the device's stream and returns; the database's device tables and the
 * query set may be destroyed right after it in the usual way ONLY IF their destruction is stream-ordered (a database made by
 * s4g_db_create_cached is; s4g_queries_destroy frees memory and therefore waits for the device) -- the caller keeps `q` alive until
 * finish.  s4g_traceback_finish waits for the job, writes h_out / h_paths exactly as s4g_traceback_host would (path_off as
 * given to submit) and frees the job; a job is finished exactly once.  A request too large for one batch is traced at once, inside
 * submit; finish then only copies. */
typedef struct s4g_trace_job s4g_trace_job;
int s4g_traceback_submit(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer, uint32_t n_pairs,
                         const uint32_t* pair_q, const uint32_t* pair_t, const int32_t* pair_score, const uint64_t* path_off,
                         s4g_trace_job** job);
int s4g_traceback_finish(s4g_trace_job* job, int32_t* h_out, uint8_t* h_paths);

/* Introspection for measurement (bench.py): name and launch geometry of the
 * kernel s4g_scan dispatches for this (db, queries, scorer) combination. */
int s4g_scan_plan(const s4g_db* db, const s4g_queries* q, const s4g_scorer* scorer,
                  char* kernel_name, size_t kernel_name_cap, uint32_t* grid_blocks, uint32_t* block_threads,
                  uint32_t* strip_rows, uint64_t* workspace_bytes);

/* What the last s4g_scan* / s4g_scan_topk* / s4g_score_*_host call on `device` did (measurement; synchronises the device):
 * out[0..3] = queries routed to the drifting-frame / maximum3 / plain int16 / int32 kernels (summed over database tiles and
 * query batches), out[4] = of those, (query, tile) combinations that ran speculatively, out[5] = score entries the
 * speculative launches marked and the exact int32 pass recomputed, out[6] = (query, target) pairs given to the
 * one-wave-per-pair kernels, out[7] = pairs scored by the packed list kernel (s4g_score_pairs_host / s4g_score_lists_host).  cap >= 8;
 * with cap >= 10 also out[8] / out[9] = pairs of the last pair submission / the last traceback call that ran on the cooperative
 * kernels (four waves per pair: the pairs whose own latency would have dominated the launch); with cap >= 11 also out[10] = pairs of
 * the last pair submission scored by the packed one-wave-per-pair kernels (sw_pair_scan_pk: two rows per register). */
int s4g_scan_stats(int device, uint64_t* out, uint32_t cap);

/* Shader clock and duration of the scan launches of the last s4g_scan* / s4g_scan_topk* call on `device`, sampled INSIDE those
 * kernels (measurement; synchronises the device): thread 0 of every persistent block reads the shader-clock counter
 * (s_memtime) and the constant 100 MHz counter (s_memrealtime) when it starts, every wave again when the work queue is
 * drained.  out[0] = launches sampled in the call (at most 16); for the LONGEST of them: out[1] = shader clock in MHz, mean over
 * blocks of d(s_memtime) / d(s_memrealtime) x 100, out[2] / out[3] = smallest / largest block value, out[4] = blocks that
 * contributed, out[5] = duration in ms by the 100 MHz counter (longest block), out[6] = tier of that launch (0 drifting frame,
 * 1 maximum3, 2 plain int16, 3 int32), out[7] = shader clocks of the longest block; with cap >= 10 also out[8] / out[9] = mean / shortest
 * block duration in ms (mean / out[5] = the share of the launch the persistent blocks were busy: the rest is the drain of the work
 * queue, which matters when items are few and long -- 64 queries of 5000 residues: 0.94; 1000 of 350: 0.999).  All zero if the call
 * launched no scan kernel.  cap >= 8.  This is the clock bench.py prices the vector-issue roofline with. */
int s4g_scan_clock(int device, double* out, uint32_t cap);

/* Starts, once per device and on a thread of its own, everything a process pays before its first scan: HIP context, code
 * objects, side stream, pinned staging.  Returns at once; calls that need the device meanwhile wait for it.  The
 * swsharp-compatible library calls it from cudaCheckCards (/root/reference/sift4g/src/main.cpp:186), so the warm-up runs
 * underneath sift4g's FASTA reading and prefilter (main.cpp:192-205). */
int s4g_warmup(int device);

/* Shader clock under a full vector-ALU load and the measured issue rate of packed-int16 vector instructions: what bench.py
 * prices the scan kernels' VALU-issue roofline with.  out[0] = shader clock in MHz (0 if the part's cycle counter does not
 * follow the shader clock), out[1] = measured T lane-ops/s of v_pk_add_u16 at 8 waves per SIMD, out[2] = compute units,
 * out[3] = the runtime's nominal clock in MHz, out[4] = duration of the probe kernel in ms.  cap >= 5.  ~50 ms. */
int s4g_clock_probe(int device, double* out, uint32_t cap);

/* ---- k-mer prefilter ------------------------------------------------------------------------------
 * Replaces searchDatabase's compute (/root/reference/sift4g/src/database_search.cpp:66-253 with hash.cpp:21-90): the CSR
 * index of the queries' k-mers (createHash, database_search.cpp:71-72), per database chunk the hit lists of every target
 * against every query, their longest increasing subsequence, score = LIS / (float) target length (threadSearchDatabase,
 * :185-253), and the running best max_candidates per query across chunks (:129-150).  All of it runs on the device; the
 * caller keeps reading FASTA.  Results are the reference's for every candidate strictly above the score at the cut; among
 * candidates whose score EQUALS it the reference's choice depends on its thread count and on std::sort, here the lower
 * database index wins (one of the lists the reference can return).
 *
 *   s4g_prefilter_create   queries: codes 0..25, CSR offsets; kmer_length 3..5 (main.cpp:169); max_candidates >= 1
 *   s4g_prefilter_add      one database chunk of host memory (may be freed on return): targets first_id .. first_id+n-1.
 *                          Chunks must arrive in ascending id order.  Synchronous.
 *   s4g_prefilter_add_device  the same with the chunk already in device memory (d_offsets: n+1 uint64)
 *   s4g_prefilter_finish   h_ids / h_scores: n_queries rows of max_candidates; the first h_counts[q] of row q are valid, ids
 *                          ascending as searchDatabase returns them (database_search.cpp:166-178).  h_scores may be NULL.
 *                          May be called more than once; adding after it is allowed (it only reads the running lists).
 * A handle is used by one thread at a time; an error other than S4G_ERR_INVALID on an argument leaves its lists undefined
 * (destroy it).  Residue codes may be any 5-bit value (0..31); the 134 MB index at k = 5 is sized for that.
 *   s4g_prefilter_stats    out[0] = k-mer hits looked up so far, out[1] = (query, target) pairs with at least one hit,
 *                          out[2] = of those, pairs whose LIS ran on a whole wave (more than 32 hits), out[3] = chunks. cap >= 4.
 *                          With cap >= 7 also: out[4] = 1 if the handle runs the fused path (one wave per target keeps its hits in
 *                          LDS: sort by query, LIS and candidate there; lists merged per query in LDS), 0 for the general path
 *                          (every pass through HBM: max_candidates beyond what a block sorts in LDS, more than 2^22 queries, or
 *                          S4G_PREFILTER_GLOBAL_SORT=1), out[5] = targets the fused path handed to the general passes (more than
 *                          1024 hits), out[6] = chunks it ran twice (the first run found more candidates than it had room for).
 * Limits: n_queries x max_candidates <= 2^30; one chunk may produce at most 2^31 - 64 hits per pass (the general path cuts a chunk
 * into passes) and 2^31 - 64 candidates; s4g_prefilter_add_device reads d_codes / d_offsets on the handle's own stream: they
 * must be complete when it is called (it checks the offsets on the device: ascending, targets of at most 0x7FFFFFF0 residues). */
typedef struct s4g_prefilter s4g_prefilter;
int s4g_prefilter_create(s4g_prefilter** out, int device, const uint8_t* codes, const uint64_t* offsets, uint32_t n_queries,
                         uint32_t kmer_length, uint32_t max_candidates);
int s4g_prefilter_add(s4g_prefilter* p, const uint8_t* codes, const uint64_t* offsets, uint32_t n_targets, uint32_t first_id);
int s4g_prefilter_add_device(s4g_prefilter* p, const uint8_t* d_codes, const uint64_t* d_offsets, uint32_t n_targets,
                             uint32_t first_id);
int s4g_prefilter_finish(s4g_prefilter* p, uint32_t* h_ids, uint32_t* h_counts, float* h_scores);
int s4g_prefilter_stats(const s4g_prefilter* p, uint64_t* out, uint32_t cap);
int s4g_prefilter_destroy(s4g_prefilter* p);

/* ---- multi-GPU: the final gather of the per-query top-hit lists ------------------------------------------
 * One process per GPU.  The scan needs no exchange (pairs are independent); the ranks meet once, after it.
 *   s4g_comm_unique_id   128 bytes that name a communicator: ONE rank calls it and hands the bytes to the others (a file, a pipe,
 *                        MPI, torch.distributed ...).  Opens librccl on first use (S4G_ERR_NO_DEVICE if there is none).
 *   s4g_comm_create      collective: every rank of `world` calls it with the same id, its own rank and the device it scans on.
 *   s4g_gather_topk      collective, asynchronous on `stream`; all pointers are DEVICE pointers in the layout s4g_topk /
 *                        s4g_scan_topk write (ids 0xFFFFFFFF = unused, lists ordered by score descending then id ascending:
 *                        the merge relies on that order and on distinct global ids, it does not check them).  A communicator
 *                        stages through one buffer pair: a gather waits (on the device) for the previous gather of the same
 *                        communicator whatever stream that ran on, and every rank must issue its gathers in the same order.
 *                        S4G_RCCL_LIBRARY names the RCCL library to open instead of librccl.so.1.
 *     S4G_GATHER_QUERY_SHARD  database replicated; this rank holds the lists of queries [n*rank/world, n*(rank+1)/world) of the
 *                        n_queries in total (d_ids / d_scores [own queries][k], d_counts [own queries]; may be NULL for a rank
 *                        without queries).  Out: all n_queries lists in query order, on every rank.  k_out must equal k;
 *                        id_base / d_id_map are ignored.
 *     S4G_GATHER_DB_SHARD     every rank holds lists of ALL n_queries over its own slice of the database, ids local to the slice:
 *                        global id = local id + id_base, or d_id_map[local id] when d_id_map != NULL (the map must be
 *                        ascending, so that a rank's own tie order is the global one).  Out: per query the best k_out of the
 *                        world * k gathered entries, score descending then global id ascending -- what one scan of the whole
 *                        database followed by s4g_topk returns.
 *   s4g_merge_topk       the merge kernel alone, on lists that are in device memory already: d_ids / d_scores [lists][n_queries][k_in]
 *                        (global ids, every row ordered, 0xFFFFFFFF only at its end) -> [n_queries][k_out]; asynchronous on `stream`.
 *   s4g_merge_topk_host  the same K-way merge in host memory, no device: ids / scores [lists][n_queries][k_in] (global ids, every
 *                        row ordered, 0xFFFFFFFF only at its end) -> [n_queries][k_out].  What a host merges with when the
 *                        lists came to it by other means (the CPU tests' gloo gather). */
typedef struct s4g_comm s4g_comm;
#define S4G_GATHER_QUERY_SHARD 0
#define S4G_GATHER_DB_SHARD    1
int s4g_comm_unique_id(uint8_t* id128);
int s4g_comm_create(s4g_comm** out, int device, int rank, int world, const uint8_t* id128);
/* any out pointer may be NULL; rccl_version as ncclGetVersion reports it */
int s4g_comm_info(const s4g_comm* comm, int* rank, int* world, int* device, int* rccl_version);
int s4g_comm_destroy(s4g_comm* comm);
int s4g_gather_topk(s4g_comm* comm, int mode, uint32_t n_queries, uint32_t k, const uint32_t* d_ids, const int32_t* d_scores,
                    const uint32_t* d_counts, uint32_t id_base, const uint32_t* d_id_map, uint32_t k_out, uint32_t* d_out_ids,
                    int32_t* d_out_scores, uint32_t* d_out_counts, void* stream);
int s4g_merge_topk(int device, uint32_t lists, uint32_t n_queries, uint32_t k_in, const uint32_t* d_ids, const int32_t* d_scores,
                   uint32_t k_out, uint32_t* d_out_ids, int32_t* d_out_scores, uint32_t* d_out_counts, void* stream);
int s4g_merge_topk_host(uint32_t lists, uint32_t n_queries, uint32_t k_in, const uint32_t* ids, const int32_t* scores,
                        uint32_t k_out, uint32_t* out_ids, int32_t* out_scores, uint32_t* out_counts);

#ifdef __cplusplus
}
#endif
#endif /* S4G_HIP_H */

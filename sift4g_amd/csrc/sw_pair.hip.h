// sw_pair.hip.h -- latency path: one wave per (query, target) PAIR, anti-diagonal sweep.
//
// sift4g calls alignDatabase once per query with that query's candidate list
// (/root/reference/sift4g/src/database_alignment.cpp:70-92, `indexes`): a few thousand targets scattered over the
// chunk.  The inter-sequence kernels (sw_kernels.hip.h) need 64 length-sorted neighbours per wave and one wave walks a
// whole DP, so such a call is bounded by its longest target (16 ms for 350 x 35k cells).  Here the 64 lanes of a wave
// work on ONE pair: lane l owns the RPL consecutive query rows l*RPL .. l*RPL+RPL-1 and, at step k, computes them for
// target column j = k - l.  What row block l needs from block l-1 (H and F of its last row, column j) was produced by
// lane l-1 one step earlier and arrives through one DPP wave_shr:1 move; lane 0 shifts in zeros (row -1).  A pair
// takes lt + ceil(lq / RPL) - 1 steps of RPL cells per lane, no strip boundary, no workspace, no HBM traffic besides
// the target itself.  int32 arithmetic, exact for every scorer (no tiers).
//
// LDS per wave (= block): the 32x32 int8 matrix, the lane profile prof[t][lane] = the RPL scores M[q_row][t] of the
// lane's rows packed in bytes (one ds_read per step), and a window of the target so that lane l reads column k - l as
// one byte at a running address.
#pragma once
#include "sw_kernels.hip.h"

namespace s4g {

constexpr int kPairWindow = 2048;      // target columns staged per window
constexpr int kPairMaxRows = 64 * 16;  // longest query this path takes

struct PairItem { uint32_t q, t, out; };   // query id, target id (original index), index into scores

struct PairArgs {
    const PairItem* __restrict__ items;     // [n_pairs], most expensive first
    const uint64_t* __restrict__ seq_addr;  // [n_seqs] address of a target's first residue dword (dword c lives 64*c dwords further)
    const uint32_t* __restrict__ seq_len;   // [n_seqs]
    const uint8_t* __restrict__ q_codes;
    const uint64_t* __restrict__ q_off;
    const uint32_t* __restrict__ q_len;
    const int8_t* __restrict__ mat;
    int32_t* __restrict__ scores;
    unsigned* counter;
    uint32_t n_pairs;
    uint32_t open, ext;
};

__device__ __forceinline__ int wave_shr1(int v) {     // lane l <- lane l-1, lane 0 <- 0
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false);
}

// Software pipelining of the two dependent LDS reads of a step (target residue -> packed scores).  The reads are plain
// loads the compiler tracks; `pin` is an empty statement that makes a value opaque at a chosen point, which (a) keeps
// the compiler from rotating the loop back into load -> wait -> use (that exposes two LDS round trips per step) and
// (b) puts the s_waitcnt where the statement stands, i.e. after the step that ran underneath the loads.
// (An earlier version issued the ds_reads through asm statements with a separate s_waitcnt statement; the compiler
// then copied a register between the two statements -- before the data had arrived -- in one instantiation.  Memory
// operations the compiler cannot see are not worth that risk, so the loads stay ordinary loads.)
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pin(unsigned& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(int& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(v2i& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(v4i& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ int pk_get(int v, int) { return v; }
template <typename V> __device__ __forceinline__ int pk_get(const V& v, int i) { return v[i]; }
__device__ __forceinline__ void pk_set(int& v, int, int x) { v = x; }
template <typename V> __device__ __forceinline__ void pk_set(V& v, int i, int x) { v[i] = x; }
template <int NW> struct PackOf;
template <> struct PackOf<1> { typedef int type; };
template <> struct PackOf<2> { typedef v2i type; };
template <> struct PackOf<4> { typedef v4i type; };

// lane profile: prof[(t*64 + lane)*NW + w] byte b = M[q_{lane*RPL + 4w + b}][t]  (rows beyond the query: pad code)
template <int RPL, int NW>
__device__ __forceinline__ void pair_build_profile(char* prof, const int* lmat4, const uint8_t* qc, unsigned lq, int lane) {
    typedef typename PackOf<NW>::type vnw;
    __builtin_amdgcn_wave_barrier();
    int rows[RPL][8];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const unsigned i = (unsigned)lane * RPL + r;
        const unsigned code = i < lq ? qc[i] : (unsigned)kPadCode;
#pragma unroll
        for (int d = 0; d < 8; ++d) rows[r][d] = lmat4[code * 8 + d];
    }
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        vnw pk;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            unsigned x = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (4 * w + b < RPL) x |= (((unsigned)rows[4 * w + b][t >> 2] >> (8 * (t & 3))) & 0xFFu) << (8 * b);
            pk_set(pk, w, (int)x);
        }
        *reinterpret_cast<vnw*>(prof + ((t * 64 + lane) * NW) * 4) = pk;
    }
    __builtin_amdgcn_wave_barrier();
}

// stage target columns [w0 - 64, kend + 8) as bytes (index = column - w0 + 64); pad outside [0, lt)
typedef const __attribute__((address_space(1))) uint32_t* global_u32_ptr;      // global_load instead of flat_load
__device__ __forceinline__ void pair_stage_window(char* stage, const uint32_t* res, int len4, unsigned w0, unsigned kend, int lane) {
    __builtin_amdgcn_wave_barrier();
    global_u32_ptr gres = (global_u32_ptr)res;
    const int e_end = (int)((kend - w0 + 64 + 8 + 3) / 4);
    for (int e = lane; e < e_end; e += 64) {
        const int c = (int)(w0 / 4) - 16 + e;
        unsigned v = 0x1F1F1F1Fu;
        if (c >= 0 && c < len4) v = gres[(uint64_t)c * 64];
        reinterpret_cast<unsigned*>(stage)[e] = v;
    }
    __builtin_amdgcn_wave_barrier();
}

template <int RPL>
__global__ __launch_bounds__(64) void sw_pair_scan(const PairArgs a) {
    constexpr int NW = RPL <= 4 ? 1 : (RPL <= 8 ? 2 : 4);       // profile dwords per (t, lane)
    constexpr int W = kPairWindow;
    typedef typename PackOf<NW>::type vnw;
    __shared__ __attribute__((aligned(16))) char lds[1024 + 32 * 64 * NW * 4 + W + 192];
    const int lane = threadIdx.x;
    const int* lmat4 = reinterpret_cast<const int*>(lds);
    char* prof = lds + 1024;
    char* stage = prof + 32 * 64 * NW * 4;
    for (int i = lane; i < 256; i += 64) reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    __builtin_amdgcn_wave_barrier();
    const unsigned open = a.open, ext = a.ext;
    unsigned have_q = 0xFFFFFFFFu;

    // canonical loop shape (fetch in the header, no break / continue inside); see sw_pair_scan_band
    auto next_item = [&]() -> unsigned {
        unsigned it = 0;
        if (lane == 0) it = atomicAdd(a.counter, 1u);
        return (unsigned)__builtin_amdgcn_readfirstlane((int)it);
    };
    for (unsigned item = next_item(); item < a.n_pairs; item = next_item()) {
        const unsigned q = a.items[item].q;
        const unsigned tid = a.items[item].t;
        const unsigned out = a.items[item].out;
        const unsigned lq = a.q_len[q];
        const unsigned lt = a.seq_len[tid];
        const uint32_t* res = reinterpret_cast<const uint32_t*>(a.seq_addr[tid]);
        const int len4 = (int)((lt + 3) / 4);

        if (q != have_q) {
            have_q = q;
            pair_build_profile<RPL, NW>(prof, lmat4, a.q_codes + a.q_off[q], lq, lane);
        }

        const unsigned lanes_used = (lq + RPL - 1) / RPL;
        const unsigned nsteps = (lq && lt) ? lt + lanes_used - 1 : 0;      // an empty query or target: no step, score 0
        int H[RPL];
        unsigned E[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) { H[r] = 0; E[r] = 0; }
        int hu_prev = 0;            // H(row above the block, previous column)
        unsigned f_out = 0;         // F handed to the block below, for this lane's current column
        int best = 0;
        const char* prof_lane = prof + lane * (NW * 4);
        auto step = [&](const vnw& sc) {
            const int hu = wave_shr1(H[RPL - 1]);       // H(row above the block, this column)
            unsigned F = (unsigned)wave_shr1((int)f_out);
            int d[RPL];
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int w = pk_get(sc, r / 4);
                const int s = (int)(signed char)(w >> (8 * (r % 4)));
                d[r] = (r == 0 ? hu_prev : H[r - 1]) + s;
            }
            int hprev = 0;
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int h = imax(imax(d[r], (int)E[r]), (int)F);
                if (r & 1) best = imax(imax(best, hprev), h); else hprev = h;
                const unsigned hm = usub_sat((unsigned)h, open);
                E[r] = umax_s(usub_sat(E[r], ext), hm);
                F = umax_s(usub_sat(F, ext), hm);
                H[r] = h;
            }
            if (RPL & 1) best = imax(best, hprev);
            f_out = F;
            hu_prev = hu;
        };

        for (unsigned w0 = 0; w0 < nsteps; w0 += W) {
            const unsigned kend = nsteps < w0 + W ? nsteps : w0 + W;
            pair_stage_window(stage, res, len4, w0, kend, lane);
            // two steps per iteration; while a step runs on scA, the scores of the next step (scB) and the residue of
            // the one after (tC) are in flight
            const unsigned char* sa = reinterpret_cast<const unsigned char*>(stage) + 64 - lane;    // column k - lane, k = w0
            unsigned tB = sa[0];
            vnw scA = *reinterpret_cast<const vnw*>(prof_lane + tB * (64 * NW * 4));
            tB = sa[1];
            pin(tB);
            pin(scA);
            for (unsigned k = w0; k < kend; k += 2) {       // an odd tail runs one harmless extra step (pad columns)
                vnw scB = *reinterpret_cast<const vnw*>(prof_lane + tB * (64 * NW * 4));     // scores of step k + 1
                unsigned tC = sa[2];                                                          // residue of step k + 2
                __builtin_amdgcn_sched_barrier(0);
                step(scA);
                __builtin_amdgcn_sched_barrier(0);
                pin(tC);
                pin(scB);
                scA = *reinterpret_cast<const vnw*>(prof_lane + tC * (64 * NW * 4));         // step k + 2
                tB = sa[3];                                                                   // step k + 3
                sa += 2;
                __builtin_amdgcn_sched_barrier(0);
                step(scB);
                __builtin_amdgcn_sched_barrier(0);
                pin(tB);
                pin(scA);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) best = imax(best, __shfl_xor(best, off, 64));
        a.scores[out] = best;          // every lane holds the maximum; one address, no divergent branch
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Queries beyond 64 * 16 rows: the same sweep band by band.  A band is 1024 query rows (16 per lane); its last row's
// H and F (the F the row below receives) go to a per-wave workspace in HBM, one (H, F) pair per target column, and
// the next band shifts them in at lane 0 instead of zeros.  8 bytes written and read per column and band -- 1/128
// byte per cell.  Lane 63 writes its pair each step (column k - 63); a window of the boundary is staged in LDS
// next to the target window and read as a wave-uniform (broadcast) address, pipelined one step ahead like the scores.
struct PairBandArgs {
    PairArgs p;
    int2* __restrict__ band_ws;     // [grid][band_ws_slot] (H, F) of the previous band's last row
    uint64_t band_ws_slot;          // columns per wave slot (>= the longest target of the launch)
};

// LDS of one band wave: substitution table, lane profile, target window, boundary window
constexpr int kPairBandLds = 1024 + 32 * 64 * 4 * 4 + kPairWindow + 192 + (kPairWindow + 64) * 8;

// score of (query q, target tid) for the whole wave (every lane returns the maximum); ws = this wave's boundary slot
template <int RPL>
__device__ __forceinline__ int pair_band_score(const PairArgs& a, int2* ws, char* lds, unsigned q, unsigned tid, int lane) {
    static_assert(RPL == 16, "a band is 64 lanes x 16 rows");
    constexpr int NW = 4, BAND = 64 * RPL;
    constexpr int W = kPairWindow;
    typedef v4i vnw;
    const int* lmat4 = reinterpret_cast<const int*>(lds);
    char* prof = lds + 1024;
    char* stage = prof + 32 * 64 * NW * 4;
    int2* bnd = reinterpret_cast<int2*>(stage + W + 192);      // boundary of columns w0 .. of the current window
    const unsigned open = a.open, ext = a.ext;
    const unsigned lq = a.q_len[q];
    const unsigned lt = a.seq_len[tid];
    const uint32_t* res = reinterpret_cast<const uint32_t*>(a.seq_addr[tid]);
    const int len4 = (int)((lt + 3) / 4);
    const unsigned nbands = lt ? (lq + BAND - 1) / BAND : 0;       // an empty query or target: no band, score 0
    int best = 0;
    const char* prof_lane = prof + lane * (NW * 4);

    for (unsigned band = 0; band < nbands; ++band) {
        const unsigned row0 = band * BAND;
        const unsigned rows = lq - row0 < (unsigned)BAND ? lq - row0 : (unsigned)BAND;
        const bool last_band = band + 1 == nbands;
        pair_build_profile<RPL, NW>(prof, lmat4, a.q_codes + a.q_off[q] + row0, rows, lane);
        if (band) {     // the boundary written by this wave during the previous band must be what the loads below see
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        const unsigned lanes_used = (rows + RPL - 1) / RPL;
        const unsigned nsteps = lt + lanes_used - 1;
        int H[RPL];
        unsigned E[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) { H[r] = 0; E[r] = 0; }
        int hu_prev = 0;
        unsigned f_out = 0;
        auto step = [&](const vnw& sc, const int2 top) {
            const int hu = __builtin_amdgcn_update_dpp(top.x, H[RPL - 1], 0x138, 0xf, 0xf, false);      // lane 0 <- boundary H
            unsigned F = (unsigned)__builtin_amdgcn_update_dpp(top.y, (int)f_out, 0x138, 0xf, 0xf, false);
            int d[RPL];
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int w = pk_get(sc, r / 4);
                const int s = (int)(signed char)(w >> (8 * (r % 4)));
                d[r] = (r == 0 ? hu_prev : H[r - 1]) + s;
            }
            int hprev = 0;
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int h = imax(imax(d[r], (int)E[r]), (int)F);
                if (r & 1) best = imax(imax(best, hprev), h); else hprev = h;
                const unsigned hm = usub_sat((unsigned)h, open);
                E[r] = umax_s(usub_sat(E[r], ext), hm);
                F = umax_s(usub_sat(F, ext), hm);
                H[r] = h;
            }
            f_out = F;
            hu_prev = hu;
        };

        for (unsigned w0 = 0; w0 < nsteps; w0 += W) {
            const unsigned kend = nsteps < w0 + W ? nsteps : w0 + W;
            pair_stage_window(stage, res, len4, w0, kend, lane);
            // boundary of columns w0 .. kend + 2 (lane 0's column at step k is k): zeros for the first band and past the target
            for (unsigned e = lane; e < kend - w0 + 3; e += 64) {
                int2 v = {0, 0};
                if (band && w0 + e < lt) v = ws[w0 + e];
                bnd[e] = v;
            }
            __builtin_amdgcn_wave_barrier();
            const unsigned char* sa = reinterpret_cast<const unsigned char*>(stage) + 64 - lane;
            const int2* ba = bnd;
            unsigned tB = sa[0];
            vnw scA = *reinterpret_cast<const vnw*>(prof_lane + tB * (64 * NW * 4));
            int2 topA = ba[0];
            tB = sa[1];
            pin(tB);
            pin(scA);
            pin(topA.x); pin(topA.y);
            for (unsigned k = w0; k < kend; k += 2) {
                vnw scB = *reinterpret_cast<const vnw*>(prof_lane + tB * (64 * NW * 4));
                unsigned tC = sa[2];
                int2 topB = ba[1];
                __builtin_amdgcn_sched_barrier(0);
                step(scA, topA);
                if (!last_band && lane == 63 && k >= 63 && k - 63 < lt) ws[k - 63] = int2{H[RPL - 1], (int)f_out};
                __builtin_amdgcn_sched_barrier(0);
                pin(tC);
                pin(scB);
                pin(topB.x); pin(topB.y);
                scA = *reinterpret_cast<const vnw*>(prof_lane + tC * (64 * NW * 4));
                tB = sa[3];
                topA = ba[2];
                sa += 2;
                ba += 2;
                __builtin_amdgcn_sched_barrier(0);
                step(scB, topB);
                if (!last_band && lane == 63 && k + 1 >= 63 && k + 1 - 63 < lt) ws[k + 1 - 63] = int2{H[RPL - 1], (int)f_out};
                __builtin_amdgcn_sched_barrier(0);
                pin(tB);
                pin(scA);
                pin(topA.x); pin(topA.y);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = imax(best, __shfl_xor(best, off, 64));
    return best;
}

template <int RPL>
__global__ __launch_bounds__(64) void sw_pair_scan_band(const PairBandArgs b) {
    const PairArgs& a = b.p;
    __shared__ __attribute__((aligned(16))) char lds[kPairBandLds];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    __builtin_amdgcn_wave_barrier();
    int2* ws = b.band_ws + (uint64_t)blockIdx.x * b.band_ws_slot;

    // canonical loop shape (fetch in the header, no break / continue inside): with the usual for (;;) + break form the
    // compiler's control-flow structurizer dropped lane 0 from EXEC after the first item of this kernel
    auto next_item = [&]() -> unsigned {
        unsigned it = 0;
        if (lane == 0) it = atomicAdd(a.counter, 1u);
        return (unsigned)__builtin_amdgcn_readfirstlane((int)it);
    };
    for (unsigned item = next_item(); item < a.n_pairs; item = next_item()) {
        const unsigned q = a.items[item].q;
        const unsigned tid = a.items[item].t;
        const unsigned out = a.items[item].out;
        const int best = pair_band_score<RPL>(a, ws, lds, q, tid, lane);
        a.scores[out] = best;          // every lane holds the maximum; one address, no divergent branch
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Recheck pass of the speculative packed scan (sw_scan_pkd with ScanArgs16::spec_base != 0): that kernel runs queries
// whose score BOUND does not fit its int16 window and writes kScoreUnsure (sw_kernels.hip.h) wherever a result came close enough to the
// window's end that an intermediate value may have left it.  This kernel reads the score rows once (waves stride over
// 64-column slabs) and recomputes every marked entry exactly, in int32, one wave per pair, band by band.

struct RecheckArgs {
    PairBandArgs b;                     // b.p.items / n_pairs unused; b.p.counter (may be NULL) counts the recomputed entries
    uint64_t stride;                    // scores[row * stride + column]
    uint32_t q_begin, q_count, n_cols;
    const uint32_t* __restrict__ col_ids;   // column -> target id (NULL: the column IS the id)
};

template <int RPL>
__global__ __launch_bounds__(64) void sw_pair_recheck(const RecheckArgs r) {
    const PairArgs& a = r.b.p;
    __shared__ __attribute__((aligned(16))) char lds[kPairBandLds];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    __builtin_amdgcn_wave_barrier();
    int2* ws = r.b.band_ws + (uint64_t)blockIdx.x * r.b.band_ws_slot;
    const uint64_t slabs_per_row = ((uint64_t)r.n_cols + 63) / 64;
    const uint64_t total = slabs_per_row * r.q_count;
    for (uint64_t s = blockIdx.x; s < total; s += gridDim.x) {
        const uint32_t row = (uint32_t)(s / slabs_per_row);
        const uint32_t col0 = (uint32_t)(s - (uint64_t)row * slabs_per_row) * 64;
        int32_t* line = a.scores + (uint64_t)row * r.stride;
        const int v = col0 + lane < r.n_cols ? line[col0 + lane] : 0;
        unsigned long long marked = __ballot(v == kScoreUnsure);
        if (marked && a.counter && lane == 0) atomicAdd(a.counter, (unsigned)__popcll(marked));
        while (marked) {
            const unsigned col = col0 + (unsigned)__builtin_ctzll(marked);
            marked &= marked - 1;
            const unsigned tid = r.col_ids ? r.col_ids[col] : col;
            const int best = pair_band_score<RPL>(a, ws, lds, r.q_begin + row, tid, lane);
            line[col] = best;
        }
    }
}

// The cells of one step of a traced sweep (rows r = 0 .. RPL-1 of this lane's block at its current column): H, E, F as in the
// score kernels, plus per cell the direction code (bits 0-1: source of H -- 0 none, 1 diagonal, 2 up (F), 3 left (E); bit 2: E of
// this cell was opened from H; bit 3: F was) and the search for the end cell.  hu = H of the row above at this column, fin = F
// coming from above | its "opened" bit << 31.  Written for few instructions per cell -- a per-query call's traceback launch is one
// wave per hit walking this loop, and it was 35 instructions per cell with six branches per step: the tie rule is folded into one
// compare against E or F (lf picks which), the "opened" bits are kept where the code wants them, and cells that equal the wanted
// score are only looked for (row by row) in the rare step where some lane has one.  Returns the RPL codes, byte r = row r.
template <int RPL, int NW>
__device__ __forceinline__ typename PackOf<NW>::type
trace_cells(int (&H)[RPL], unsigned (&E)[RPL], unsigned (&eo4)[RPL], int& hu_prev, unsigned& f_out, int& best, unsigned& key,
            const typename PackOf<NW>::type& sc, int hu, unsigned fin, int want, unsigned cell0, unsigned open, unsigned ext,
            bool lf, unsigned s_hit, unsigned s_miss) {
    unsigned F = fin & 0x7FFFFFFFu;
    unsigned fo8 = (fin >> 31) << 3;
    int d[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int w = pk_get(sc, r / 4);
        const int s = (int)(signed char)(w >> (8 * (r % 4)));
        d[r] = (r == 0 ? hu_prev : H[r - 1]) + s;
    }
    int D[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) D[w] = 0;
    bool anyeq = false;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int h = imax(imax(d[r], (int)E[r]), (int)F);
        const int X = lf ? (int)E[r] : (int)F;                  // the gap source the tie rule looks at first
        const unsigned src = h == 0 ? 0u : (h == d[r] ? 1u : (h == X ? s_hit : s_miss));
        D[r / 4] |= (int)((src | eo4[r] | fo8) << (8 * (r % 4)));
        anyeq |= (h == want);
        best = imax(best, h);
        const unsigned hm = usub_sat((unsigned)h, open);
        const unsigned En = umax_s(usub_sat(E[r], ext), hm);
        const unsigned Fn = umax_s(usub_sat(F, ext), hm);
        eo4[r] = En == hm ? 4u : 0u;
        fo8 = Fn == hm ? 8u : 0u;
        E[r] = En;
        F = Fn;
        H[r] = h;
    }
    if (anyeq) {
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const unsigned kc = H[r] == want ? cell0 + ((unsigned)r << 16) : 0xFFFFFFFFu;
            key = kc < key ? kc : key;
        }
    }
    f_out = F | (fo8 << 28);
    hu_prev = hu;
    typename PackOf<NW>::type Dv;
#pragma unroll
    for (int w = 0; w < NW; ++w) pk_set(Dv, w, D[w]);
    return Dv;
}

// The backward walk of a traced pair, for the whole wave.  Lane 0 follows the direction codes from the end cell; every code it reads
// depends on the one before, and read from HBM/L2 one at a time that is ~0.7 us per move -- 250 us for a 350-move path, most of what
// sw_pair_trace took for a per-query call's hits (profiles/r03_dropin/).  So the wave fetches the codes in TILES into LDS -- the 64
// step lines at and below the current step, 16 lanes wide: a path moves one or two lines down per move and one lane per RPL rows --
// and lane 0 walks inside the tile until it leaves it.  BAND = rows per band (0: the block has one band); `tile` = 4096 * NW bytes
// of LDS the sweep no longer needs.  Writes the moves from pe backwards and out[0..4]; tie rules as before.
template <int RPL, int NW, int BAND>
__device__ __forceinline__ void pair_walk(const uint8_t* __restrict__ dirs, uint64_t band_bytes, uint8_t* pe, unsigned key, unsigned cap,
                                          int32_t* out, char* tile, int lane) {
    constexpr int kLines = 64, kLanes = 16, kRow = kLanes * NW;         // dwords per tile line
    int i = (int)(key >> 16), j = (int)(key & 0xFFFFu);
    const int ei = i, ej = j;
    unsigned n = 0, state = 0;
    bool done = false;
    for (;;) {
        // the tile for the current cell (uniform: i, j are lane 0's, broadcast below)
        const unsigned bd = BAND ? (unsigned)i / (unsigned)(BAND ? BAND : 1) : 0u;
        const unsigned ib = BAND ? (unsigned)i % (unsigned)(BAND ? BAND : 1) : (unsigned)i;
        const int l = (int)(ib / RPL);
        const int k = j + l;
        const int l_lo = l >= kLanes - 1 ? l - (kLanes - 1) : 0;
        const int k_lo = k >= kLines - 1 ? k - (kLines - 1) : 0;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(dirs + (uint64_t)bd * band_bytes) + ((uint64_t)k_lo * 64 + l_lo) * NW;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e0 = 0; e0 < kLines * kRow; e0 += 64) {
            const int e = e0 + lane;
            const int t = e / kRow, c = e % kRow;
            uint32_t v = 0;
            if (k_lo + t <= k) v = src[(uint64_t)t * 64 * NW + c];
            reinterpret_cast<uint32_t*>(tile)[e] = v;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);
            for (;;) {
                if (!(i >= 0 && j >= 0 && n < cap)) { done = true; break; }
                const unsigned bd2 = BAND ? (unsigned)i / (unsigned)(BAND ? BAND : 1) : 0u;
                const unsigned ib2 = BAND ? (unsigned)i % (unsigned)(BAND ? BAND : 1) : (unsigned)i;
                const int l2 = (int)(ib2 / RPL), r2 = (int)(ib2 % RPL);
                const int k2 = j + l2;
                if (bd2 != bd || k2 < k_lo || l2 < l_lo) break;                 // left the tile: fetch the next one
                const unsigned c = tb[(((k2 - k_lo) * kLanes + (l2 - l_lo)) * NW) * 4 + r2];
                if (state == 0) {
                    const unsigned s0 = c & 3u;
                    if (s0 == 0u) { done = true; break; }
                    if (s0 == 1u) { *(pe - 1 - n) = 1; ++n; --i; --j; }      // MOVE_DIAG
                    else state = (s0 == 2u) ? 2u : 1u;
                } else if (state == 1) {
                    *(pe - 1 - n) = 2; ++n;                                      // MOVE_LEFT
                    if (c & 4u) state = 0;
                    --j;
                } else {
                    *(pe - 1 - n) = 3; ++n;                                      // MOVE_UP
                    if (c & 8u) state = 0;
                    --i;
                }
            }
        }
        done = __builtin_amdgcn_readfirstlane((int)done) != 0;
        i = __builtin_amdgcn_readfirstlane(i);
        j = __builtin_amdgcn_readfirstlane(j);
        if (done) break;
    }
    if (lane == 0) { out[0] = i + 1; out[1] = ei; out[2] = j + 1; out[3] = ej; out[4] = (int)n; }
}

// ---------------------------------------------------------------------------------------------------------------
// Traceback of a pair (step 4 of alignDatabase for the survivors), same sweep: every cell also emits the 4-bit
// direction code of sw_trace.hip.h (TB_*: source of H, "E opened here", "F opened here"), stored per step as one
// lane-contiguous line (byte r of lane l at step k = row l*RPL + r, column k - l), and the wave tracks the first cell
// in row-major order whose H equals the known score.  Lane 0 then walks the codes back (they were just written by this
// wave: L2 hits) and writes the moves from the END of the pair's path region backwards, so no reversal is needed:
// the path is the last out[4] bytes of the region.  Tie rules are those of sw_trace_fill / the oracle.
struct PairTraceItem {
    uint32_t q, t;          // query id, target id
    int32_t score;          // known optimal score (from the scan)
    uint32_t slot;          // index of the pair in out5 / best
    uint64_t dir_off;       // byte offset of the pair's direction block: (steps rounded up to even) * 64 * NW*4 bytes
    uint64_t path_end;      // byte offset one past the pair's path region (capacity >= lq + lt)
};

struct PairTraceArgs {
    const PairTraceItem* __restrict__ items;
    const uint64_t* __restrict__ seq_addr;
    const uint32_t* __restrict__ seq_len;
    const uint8_t* __restrict__ q_codes;
    const uint64_t* __restrict__ q_off;
    const uint32_t* __restrict__ q_len;
    const int8_t* __restrict__ mat;
    uint8_t* __restrict__ dirs;
    uint8_t* __restrict__ paths;
    int32_t* __restrict__ out5;     // [slot][5]: q_start, q_end, t_start, t_end, path_len   (-1,-1,-1,-1,0 if score <= 0)
    int32_t* __restrict__ best;     // [slot] recomputed optimal score (cross-check against `score`)
    unsigned* counter;
    uint32_t n_pairs;
    uint32_t open, ext;
    uint32_t left_first;            // ties between the two gap sources of H: 0 = UP (F) before LEFT (E), 1 = LEFT before UP (S4G_TRACEBACK_TIES)
};

template <int RPL>
__global__ __launch_bounds__(64) void sw_pair_trace(const PairTraceArgs a) {
    constexpr int NW = RPL <= 4 ? 1 : (RPL <= 8 ? 2 : 4);
    constexpr int W = kPairWindow;
    typedef typename PackOf<NW>::type vnw;
    __shared__ __attribute__((aligned(16))) char lds[1024 + 32 * 64 * NW * 4 + W + 192];
    const int lane = threadIdx.x;
    const int* lmat4 = reinterpret_cast<const int*>(lds);
    char* prof = lds + 1024;
    char* stage = prof + 32 * 64 * NW * 4;
    for (int i = lane; i < 256; i += 64) reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    __builtin_amdgcn_wave_barrier();
    const unsigned open = a.open, ext = a.ext;
    unsigned have_q = 0xFFFFFFFFu;

    for (;;) {
        unsigned item = 0;
        if (lane == 0) item = atomicAdd(a.counter, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= a.n_pairs) break;
        const PairTraceItem it = a.items[item];
        const unsigned q = it.q;
        const unsigned lq = a.q_len[q];
        const unsigned lt = a.seq_len[it.t];
        const int want = it.score;
        int32_t* out = a.out5 + (uint64_t)it.slot * 5;
        if (lq == 0 || lt == 0 || want <= 0) {
            if (lane == 0) { out[0] = out[1] = out[2] = out[3] = -1; out[4] = 0; a.best[it.slot] = 0; }
            continue;
        }
        const uint32_t* res = reinterpret_cast<const uint32_t*>(a.seq_addr[it.t]);
        const int len4 = (int)((lt + 3) / 4);
        if (q != have_q) {
            have_q = q;
            pair_build_profile<RPL, NW>(prof, lmat4, a.q_codes + a.q_off[q], lq, lane);
        }
        const unsigned lanes_used = (lq + RPL - 1) / RPL;
        const unsigned nsteps = lt + lanes_used - 1;
        int H[RPL];
        unsigned E[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) { H[r] = 0; E[r] = 0; }
        int hu_prev = 0;
        unsigned f_out = 0;         // F for the block below | "F opened here" << 31
        unsigned eo4[RPL];          // 4 where E(row r, current column) was opened from H(row r, previous column)
#pragma unroll
        for (int r = 0; r < RPL; ++r) eo4[r] = 0;
        int best = 0;
        unsigned key = 0xFFFFFFFFu;
        const char* prof_lane = prof + lane * (NW * 4);
        vnw* dst = reinterpret_cast<vnw*>(a.dirs + it.dir_off) + lane;     // step k writes dst[k * 64]
        unsigned cell0 = ((unsigned)(lane * RPL) << 16) - (unsigned)lane;  // + k = (row of r = 0) << 16 | column
        const bool lf = a.left_first != 0;
        const unsigned s_hit = lf ? 3u : 2u, s_miss = lf ? 2u : 3u;
        auto step = [&](const vnw& sc) {
            const int hu = wave_shr1(H[RPL - 1]);
            const unsigned fin = (unsigned)wave_shr1((int)f_out);
            *dst = trace_cells<RPL, NW>(H, E, eo4, hu_prev, f_out, best, key, sc, hu, fin, want, cell0, open, ext, lf, s_hit, s_miss);
            dst += 64;
            ++cell0;
        };

        for (unsigned w0 = 0; w0 < nsteps; w0 += W) {
            const unsigned kend = nsteps < w0 + W ? nsteps : w0 + W;
            pair_stage_window(stage, res, len4, w0, kend, lane);
            const unsigned char* sa = reinterpret_cast<const unsigned char*>(stage) + 64 - lane;    // column k - lane, k = w0
            unsigned tB = sa[0];
            vnw scA = *reinterpret_cast<const vnw*>(prof_lane + tB * (64 * NW * 4));
            tB = sa[1];
            pin(tB);
            pin(scA);
            for (unsigned k = w0; k < kend; k += 2) {       // an odd tail runs one harmless extra step (pad columns)
                vnw scB = *reinterpret_cast<const vnw*>(prof_lane + tB * (64 * NW * 4));     // scores of step k + 1
                unsigned tC = sa[2];                                                          // residue of step k + 2
                __builtin_amdgcn_sched_barrier(0);
                step(scA);
                __builtin_amdgcn_sched_barrier(0);
                pin(tC);
                pin(scB);
                scA = *reinterpret_cast<const vnw*>(prof_lane + tC * (64 * NW * 4));         // step k + 2
                tB = sa[3];                                                                   // step k + 3
                sa += 2;
                __builtin_amdgcn_sched_barrier(0);
                step(scB);
                __builtin_amdgcn_sched_barrier(0);
                pin(tB);
                pin(scA);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            best = imax(best, __shfl_xor(best, off, 64));
            const unsigned other = (unsigned)__shfl_xor((int)key, off, 64);
            key = other < key ? other : key;
        }
        // the codes were written by all lanes of this wave; make them visible to lane 0's loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (lane == 0) a.best[it.slot] = best;
        if (key == 0xFFFFFFFFu) { if (lane == 0) { out[0] = out[1] = out[2] = out[3] = -1; out[4] = 0; } }
        else {
            pair_walk<RPL, NW, 0>(a.dirs + it.dir_off, 0, a.paths + it.path_end, key, lq + lt, out, prof, lane);
            have_q = 0xFFFFFFFFu;          // (the tile sits where the lane profile was)
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Traceback for queries beyond 64 * 16 rows: sw_pair_trace band by band, boundary row through HBM as in
// sw_pair_scan_band (F travels with its "opened here" bit).  The direction block of band b starts
// b * band_steps * 1024 bytes into the pair's block, band_steps = (lt + 63) rounded up to even; inside a band the
// layout is that of sw_pair_trace<16>.
struct PairTraceBandArgs {
    PairTraceArgs p;
    int2* __restrict__ band_ws;
    uint64_t band_ws_slot;
};

template <int RPL>
__global__ __launch_bounds__(64) void sw_pair_trace_band(const PairTraceBandArgs b) {
    static_assert(RPL == 16, "a band is 64 lanes x 16 rows");
    constexpr int NW = 4, BAND = 64 * RPL;
    constexpr int W = kPairWindow;
    typedef v4i vnw;
    const PairTraceArgs& a = b.p;
    __shared__ __attribute__((aligned(16))) char lds[1024 + 32 * 64 * NW * 4 + W + 192 + (W + 64) * 8];
    const int lane = threadIdx.x;
    const int* lmat4 = reinterpret_cast<const int*>(lds);
    char* prof = lds + 1024;
    char* stage = prof + 32 * 64 * NW * 4;
    int2* bnd = reinterpret_cast<int2*>(stage + W + 192);
    for (int i = lane; i < 256; i += 64) reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    __builtin_amdgcn_wave_barrier();
    const unsigned open = a.open, ext = a.ext;
    int2* ws = b.band_ws + (uint64_t)blockIdx.x * b.band_ws_slot;

    auto next_item = [&]() -> unsigned {        // canonical loop shape, see sw_pair_scan_band
        unsigned it = 0;
        if (lane == 0) it = atomicAdd(a.counter, 1u);
        return (unsigned)__builtin_amdgcn_readfirstlane((int)it);
    };
    for (unsigned item = next_item(); item < a.n_pairs; item = next_item()) {
        const PairTraceItem it = a.items[item];
        const unsigned q = it.q;
        const unsigned lq = a.q_len[q];
        const unsigned lt = a.seq_len[it.t];
        const int want = it.score;
        int32_t* out = a.out5 + (uint64_t)it.slot * 5;
        const uint32_t* res = reinterpret_cast<const uint32_t*>(a.seq_addr[it.t]);
        const int len4 = (int)((lt + 3) / 4);
        const unsigned nbands = (lt && want > 0) ? (lq + BAND - 1) / BAND : 0;       // nothing to trace: no band
        const uint64_t band_bytes = (uint64_t)(((lt + 63) + 1) & ~1u) * 64 * (NW * 4);
        int best = 0;
        unsigned key = 0xFFFFFFFFu;
        const char* prof_lane = prof + lane * (NW * 4);

        for (unsigned band = 0; band < nbands; ++band) {
            const unsigned row0 = band * BAND;
            const unsigned rows = lq - row0 < (unsigned)BAND ? lq - row0 : (unsigned)BAND;
            const bool last_band = band + 1 == nbands;
            pair_build_profile<RPL, NW>(prof, lmat4, a.q_codes + a.q_off[q] + row0, rows, lane);
            if (band) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            const unsigned lanes_used = (rows + RPL - 1) / RPL;
            const unsigned nsteps = lt + lanes_used - 1;
            int H[RPL];
            unsigned E[RPL];
#pragma unroll
            for (int r = 0; r < RPL; ++r) { H[r] = 0; E[r] = 0; }
            int hu_prev = 0;
            unsigned f_out = 0;
            unsigned eo4[RPL];
#pragma unroll
            for (int r = 0; r < RPL; ++r) eo4[r] = 0;
            vnw* dst = reinterpret_cast<vnw*>(a.dirs + it.dir_off + band * band_bytes) + lane;
            unsigned cell0 = ((row0 + (unsigned)(lane * RPL)) << 16) - (unsigned)lane;
            const bool lf = a.left_first != 0;
            const unsigned s_hit = lf ? 3u : 2u, s_miss = lf ? 2u : 3u;
            auto step = [&](const vnw& sc, const int2 top) {
                const int hu = __builtin_amdgcn_update_dpp(top.x, H[RPL - 1], 0x138, 0xf, 0xf, false);
                const unsigned fin = (unsigned)__builtin_amdgcn_update_dpp(top.y, (int)f_out, 0x138, 0xf, 0xf, false);
                *dst = trace_cells<RPL, NW>(H, E, eo4, hu_prev, f_out, best, key, sc, hu, fin, want, cell0, open, ext, lf, s_hit, s_miss);
                dst += 64;
                ++cell0;
            };
            for (unsigned w0 = 0; w0 < nsteps; w0 += W) {
                const unsigned kend = nsteps < w0 + W ? nsteps : w0 + W;
                pair_stage_window(stage, res, len4, w0, kend, lane);
                for (unsigned e = lane; e < kend - w0 + 3; e += 64) {
                    int2 v = {0, 0};
                    if (band && w0 + e < lt) v = ws[w0 + e];
                    bnd[e] = v;
                }
                __builtin_amdgcn_wave_barrier();
                const unsigned char* sa = reinterpret_cast<const unsigned char*>(stage) + 64 - lane;
                const int2* ba = bnd;
                unsigned tB = sa[0];
                vnw scA = *reinterpret_cast<const vnw*>(prof_lane + tB * (64 * NW * 4));
                int2 topA = ba[0];
                tB = sa[1];
                pin(tB);
                pin(scA);
                pin(topA.x); pin(topA.y);
                for (unsigned k = w0; k < kend; k += 2) {
                    vnw scB = *reinterpret_cast<const vnw*>(prof_lane + tB * (64 * NW * 4));
                    unsigned tC = sa[2];
                    int2 topB = ba[1];
                    __builtin_amdgcn_sched_barrier(0);
                    step(scA, topA);
                    if (!last_band && lane == 63 && k >= 63 && k - 63 < lt) ws[k - 63] = int2{H[RPL - 1], (int)f_out};
                    __builtin_amdgcn_sched_barrier(0);
                    pin(tC);
                    pin(scB);
                    pin(topB.x); pin(topB.y);
                    scA = *reinterpret_cast<const vnw*>(prof_lane + tC * (64 * NW * 4));
                    tB = sa[3];
                    topA = ba[2];
                    sa += 2;
                    ba += 2;
                    __builtin_amdgcn_sched_barrier(0);
                    step(scB, topB);
                    if (!last_band && lane == 63 && k + 1 >= 63 && k + 1 - 63 < lt) ws[k + 1 - 63] = int2{H[RPL - 1], (int)f_out};
                    __builtin_amdgcn_sched_barrier(0);
                    pin(tB);
                    pin(scA);
                    pin(topA.x); pin(topA.y);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            best = imax(best, __shfl_xor(best, off, 64));
            const unsigned other = (unsigned)__shfl_xor((int)key, off, 64);
            key = other < key ? other : key;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (lane == 0) a.best[it.slot] = best;
        if (key == 0xFFFFFFFFu) { if (lane == 0) { out[0] = out[1] = out[2] = out[3] = -1; out[4] = 0; } }
        else pair_walk<RPL, NW, BAND>(a.dirs + it.dir_off, band_bytes, a.paths + it.path_end, key, lq + lt, out, prof, lane);
    }
}

}  // namespace s4g

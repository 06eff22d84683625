// sw_kernels.hip.h -- gfx950 Smith-Waterman database-scan kernels.
//
// Replaces the scoring step of swsharp's alignDatabase as sift4g calls it
// (/root/reference/sift4g/src/database_alignment.cpp:83-86; swsharp itself is
// an absent submodule, so the kernels are designed from the recurrence, not
// from any reference kernel):
//
//   E[i][j] = max(E[i][j-1] - ext, H[i][j-1] - open)
//   F[i][j] = max(F[i-1][j] - ext, H[i-1][j] - open)
//   H[i][j] = max(0, H[i-1][j-1] + M[q_i][t_j], E[i][j], F[i][j]),  score = max H
//
// Mapping to CDNA4 (inter-sequence): one database sequence per lane, 64
// length-sorted sequences per wave ("group"), targets stored lane-transposed
// (4 residues per dword, dword c*64+lane) so every wave load is one contiguous
// 256-B line.  The query is cut into strips of R rows whose H and E values
// live in VGPRs; for a strip the wave builds a private LDS "strip profile"
//   prof[k][t][16] = M[q_{16k+b}][t]   (int8, b = 0..15)
// so one ds_read_b128 per lane and column delivers 16 substitution scores.
// The only HBM traffic that scales with the DP is the strip boundary: H and F
// of the strip's last row, one 16-B load and store per lane and 4 columns
// (lane-contiguous, 1 KiB per wave instruction).
//
// Arithmetic: E and F are kept clamped at 0 (max(x,0) commutes with the
// recurrence because H >= 0), which turns "x - c" into one saturating
// unsigned subtract and removes the explicit max(.,0) from H.  Per cell:
// add, max3, max(best), 3 saturating subs, 2 max  (8 VALU ops, int32).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

namespace s4g {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWave * kWavesPerBlock;
constexpr int kPadCode = 31;        // target / query padding residue
constexpr int kPadScore = -64;      // substitution score of any pair involving kPadCode
constexpr int kQueryAlign = 64;     // queries are padded to a multiple of this (>= max strip height)

struct ScanArgs {
    const uint32_t* __restrict__ db_res;    // lane-transposed residues, 4 per dword
    const uint64_t* __restrict__ grp_off;   // [G] dword offset of group g in db_res
    const uint32_t* __restrict__ grp_len4;  // [G] number of 4-column chunks of group g
    const uint32_t* __restrict__ grp_perm;  // [G*64] original target index of (g, lane); ~0u = empty lane
    const uint8_t* __restrict__ q_codes;    // queries padded with kPadCode to kQueryAlign
    const uint64_t* __restrict__ q_off;     // [Q] byte offset of query q in q_codes
    const uint32_t* __restrict__ q_len;     // [Q] true length
    const int8_t* __restrict__ mat;         // [32][32] int8, row = query code, col = target code
    int32_t* __restrict__ scores;           // [q_count][score_stride]
    uint64_t score_stride;
    int32_t* __restrict__ ws;               // strip-boundary workspace
    uint64_t ws_slot;                       // int32 elements per wave slot
    unsigned* counter;                      // work-queue head, zeroed before launch
    uint32_t n_groups;
    uint32_t q_begin, q_count;               // output row = query id - q_begin
    const uint32_t* __restrict__ q_list;     // optional: the q_count query ids to score (NULL = q_begin .. q_begin+q_count)
    uint32_t open, ext;
    unsigned long long* clk;                 // optional clock samples, 4 per block (clk_enter / clk_leave)
    uint32_t prio_cost;                      // as ScanArgs16::prio_cost (item_priority); 0 = never
    const uint32_t* __restrict__ order;      // as ScanArgs16::order; NULL: group-major
};

// Clock samples of a launch (s4g_scan_clock): thread 0 of every block stores the shader-clock counter (s_memtime) and the
// constant 100 MHz counter (s_memrealtime) as the block starts working; lane 0 of every wave raises the block's two exit
// samples as it leaves the work queue.  The persistent blocks all live from the launch to the drain of the queue, so
// (exit - entry) of the two counters is the clock the kernel ran at, measured inside the timed kernel itself, and the
// 100 MHz delta is its duration.  Two scalar reads per block and per wave: nothing next to one work item.
__device__ __forceinline__ void clk_enter(unsigned long long* clk) {
    if (clk && threadIdx.x == 0) {
        clk[4 * blockIdx.x + 0] = __builtin_readcyclecounter();
        clk[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}
__device__ __forceinline__ void clk_leave(unsigned long long* clk, int lane) {
    if (clk && lane == 0) {
        atomicMax(&clk[4 * blockIdx.x + 2], (unsigned long long)__builtin_readcyclecounter());
        atomicMax(&clk[4 * blockIdx.x + 3], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}

// Wave priority of a work item by its size (rows x columns / 4) against prio_cost = a quarter of the per-wave share of the launch: the SIMD
// arbitrates vector issue by priority, then age, so of four co-resident waves the "young" ones get the leftovers, and a launch of few,
// long items ends when the slowest of them does.  Three levels, a factor of four apart (measured on 64 queries of 2000-5000 residues,
// EXPERIMENTS.md A.7: one level at share / 4: blocks busy 0.94 of the launch; one level at share / 16: 0.986; at share / 64: 0.85).
__device__ __forceinline__ void item_priority(uint32_t prio_cost, uint64_t cost) {
    if (!prio_cost || cost * 16 < prio_cost) __builtin_amdgcn_s_setprio(0);
    else if (cost * 4 < prio_cost) __builtin_amdgcn_s_setprio(1);
    else if (cost < prio_cost) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

__device__ __forceinline__ unsigned usub_sat(unsigned a, unsigned b) {
    return __builtin_elementwise_sub_sat(a, b);
}
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned umax_s(unsigned a, unsigned b) {  // both < 2^31
    return (unsigned)imax((int)a, (int)b);
}

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));

// One DP column of an R-row strip.  H[r] holds H(row r, previous column) on
// entry and H(row r, this column) on exit; E[r] holds E(row r, this column)
// on entry and E(row r, next column) on exit.  sc = the column's R int8 scores.
template <int R>
__device__ __forceinline__ void sw_column(int (&H)[R], unsigned (&E)[R], const v4i (&sc)[R / 16],
                                          int diag, unsigned F, int& best, unsigned open, unsigned ext,
                                          int& h_last, unsigned& f_last) {
    int hprev = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int w = sc[r / 16][(r % 16) / 4];
        const int s = (int)(signed char)(w >> (8 * (r % 4)));
        const int h = imax(imax(diag + s, (int)E[r]), (int)F);
        if (r & 1) best = imax(imax(best, hprev), h); else hprev = h;
        const unsigned hm = usub_sat((unsigned)h, open);
        E[r] = umax_s(usub_sat(E[r], ext), hm);
        F = umax_s(usub_sat(F, ext), hm);
        diag = H[r];
        H[r] = h;
    }
    h_last = H[R - 1];
    f_last = F;
}

template <int R>
__device__ __forceinline__ void load_scores(v4i (&sc)[R / 16], const char* prof, unsigned res4, int cc) {
    const unsigned taddr = (cc == 0 ? (res4 << 4) : (res4 >> (8 * cc - 4))) & 0x1F0u;
#pragma unroll
    for (int k = 0; k < R / 16; ++k) sc[k] = *reinterpret_cast<const v4i*>(prof + k * 512 + taddr);
}

// Persistent kernel: waves pull (group, query) items from a global counter.
template <int R>
__global__ __launch_bounds__(kBlock, 3) void sw_scan_i32(const ScanArgs a) {
    static_assert(R % 16 == 0 && R <= kQueryAlign, "strip height");
    __shared__ __attribute__((aligned(16))) char lds[1024 + kWavesPerBlock * R * 32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    signed char* lmat = reinterpret_cast<signed char*>(lds);
    char* prof = lds + 1024 + wave * (R * 32);

    for (int i = threadIdx.x; i < 256; i += kBlock)
        reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    __syncthreads();
    clk_enter(a.clk);

    const unsigned slot = blockIdx.x * kWavesPerBlock + wave;
    v4i* wsH = reinterpret_cast<v4i*>(a.ws + (uint64_t)slot * a.ws_slot);
    const unsigned n_items = a.n_groups * a.q_count;
    // the penalties live in VGPRs: v_sub_u32 ... clamp with two VGPR sources issues at twice the rate of the SGPR-operand form
    unsigned open = a.open, ext = a.ext;
    asm volatile("" : "+v"(open), "+v"(ext));

    for (;;) {
        unsigned item = 0;
        if (lane == 0) item = atomicAdd(a.counter, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;
        if (a.order) item = a.order[item];
        const unsigned g = item / a.q_count;
        const unsigned qi = item - g * a.q_count;
        const unsigned q = a.q_list ? a.q_list[qi] : a.q_begin + qi;
        const unsigned len4 = a.grp_len4[g];
        const uint32_t* res = a.db_res + a.grp_off[g] + lane;
        const uint8_t* qc = a.q_codes + a.q_off[q];
        const unsigned lq = a.q_len[q];
        item_priority(a.prio_cost, (uint64_t)lq * len4);
        const unsigned nstrips = (lq + R - 1) / R;
        v4i* wsF = wsH + (uint64_t)len4 * 64;
        int best = 0;

        for (unsigned s = 0; s < nstrips; ++s) {
            // ---- strip profile: prof[k][t][b] = M[q_{sR+16k+b}][t] ----
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e0 = 0; e0 < 2 * R; e0 += 64) {
                const int e = e0 + lane;
                const int k = e >> 5, t = e & 31;
                const v4i q16 = *reinterpret_cast<const v4i*>(qc + s * R + 16 * k);
                v4i out;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    unsigned w = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const unsigned code = ((unsigned)q16[d] >> (8 * b)) & 31u;
                        w |= (unsigned)(unsigned char)lmat[code * 32 + t] << (8 * b);
                    }
                    out[d] = (int)w;
                }
                *reinterpret_cast<v4i*>(prof + e * 16) = out;
            }
            __builtin_amdgcn_wave_barrier();

            int H[R];
            unsigned E[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { H[r] = 0; E[r] = 0; }
            int hdiag = 0;  // H(strip top - 1, previous column)
            const bool first = (s == 0), last = (s + 1 == nstrips);

            uint32_t res_n = res[0];
            v4i hb_n = {0, 0, 0, 0}, fb_n = {0, 0, 0, 0};
            if (!first && len4) { hb_n = wsH[lane]; fb_n = wsF[lane]; }
            v4i sc_n[R / 16];
            load_scores<R>(sc_n, prof, res_n, 0);
            for (unsigned c = 0; c < len4; ++c) {
                const uint32_t r4 = res_n;
                const v4i hb = hb_n, fb = fb_n;
                const unsigned cn = (c + 1 < len4) ? c + 1 : c;
                res_n = res[(uint64_t)cn * 64];
                if (!first) { hb_n = wsH[(uint64_t)cn * 64 + lane]; fb_n = wsF[(uint64_t)cn * 64 + lane]; }
                v4i ho, fo;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    v4i sc[R / 16];
#pragma unroll
                    for (int k = 0; k < R / 16; ++k) sc[k] = sc_n[k];
                    // software pipeline: the next column's scores are in flight while this one computes
                    if (cc < 3) load_scores<R>(sc_n, prof, r4, cc + 1); else load_scores<R>(sc_n, prof, res_n, 0);
                    int hl; unsigned fl;
                    sw_column<R>(H, E, sc, hdiag, (unsigned)fb[cc], best, open, ext, hl, fl);
                    hdiag = hb[cc];
                    ho[cc] = hl; fo[cc] = (int)fl;
                    // pin the running maximum here: LLVM otherwise sinks the whole max3 chain to the
                    // end of the 4-column body and keeps every H value of the body live (250 VGPRs)
                    asm volatile("" : "+v"(best));
                    __builtin_amdgcn_sched_barrier(0);  // one scheduling region per column
                }
                if (!last) { wsH[(uint64_t)c * 64 + lane] = ho; wsF[(uint64_t)c * 64 + lane] = fo; }
            }
        }
        const unsigned p = a.grp_perm[(uint64_t)g * 64 + lane];
        if (p != 0xFFFFFFFFu) a.scores[(uint64_t)(q - a.q_begin) * a.score_stride + p] = best;
    }
    clk_leave(a.clk, lane);
}


// =====================================================================================================
// sw_scan_i32d: the int32 kernel in a DRIFTING FRAME (see sw_scan_pkd below for the algebra): every stored value carries
// (i + j) * ext, which removes the two saturating "- ext" of the gap recurrences and lets them share one add:
//     H~ = max3(diag~ + (s + 2e), E~, F~);   E~' = max3(E~, H~ + (e - o), z');   F~' = max(F~, H~ + (e - o));
//     best = max3(best, H~ - z  [two rows per instruction])
// = 6.5 vector ops per cell against sw_scan_i32's 7.5.  int32 has room for any frame (no bias, no window, no tiers): exact
// for every scorer whose shifted scores s + 2e still fit the int8 strip profile (max score + 2 ext <= 127); the host falls
// back to sw_scan_i32 otherwise.  z(i, j) is uniform over the wave and lives in SGPRs.
// (Round 5 measured this kernel in sw_scan_pkc's COLUMN frame as well -- F carries the "- e" as a two-VGPR subtract, the running
// maximum lives in the frame: the same 6.5 instructions with two of the double-rate class instead of one -- and it is 2.2 % SLOWER,
// 5 631 against 5 756 GCUPS A/B on one device: the column frame puts a fourth instruction on the dependent chain down a column
// (max3 -> add -> max -> sub), and this kernel, unlike sw_scan_pkd, had no spilled per-row frame constants to win back.)
// =====================================================================================================
template <int R>
__device__ __forceinline__ void sw_column_d(int (&H)[R], int (&E)[R], const v4i (&sc)[R / 16], int diag, int F, int& best,
                                            int eo, int zcur, int e, int& h_last, int& f_last) {
    int tprev = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int w = sc[r / 16][(r % 16) / 4];
        const int s = (int)(signed char)(w >> (8 * (r % 4)));
        const int h = imax(imax(diag + s, E[r]), F);
        const int t = h - zcur;                                   // true H
        if (r & 1) best = imax(imax(best, tprev), t); else tprev = t;
        zcur += e;                                                // scalar: z(i+1, j) = z(i, j+1)
        const int hm = h + eo;
        E[r] = imax(imax(E[r], hm), zcur);
        F = imax(F, hm);
        diag = H[r];
        H[r] = h;
    }
    h_last = H[R - 1];
    f_last = F;
}

// rows row0 .. row0+7 of one column (the tail strip of sw_scan_i32d walks a column in blocks of eight rows)
template <int R>
__device__ __forceinline__ void sw_rows8_d(int (&H)[R], int (&E)[R], int row0, const v4i (&sc)[R / 16], int& diag, int& F, int& best,
                                           int eo, int& zcur, int e) {
    int tprev = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = row0 + i;
        const int w = sc[r / 16][(r % 16) / 4];
        const int s = (int)(signed char)(w >> (8 * (r % 4)));
        const int h = imax(imax(diag + s, E[r]), F);
        const int t = h - zcur;
        if (i & 1) best = imax(imax(best, tprev), t); else tprev = t;
        zcur += e;
        const int hm = h + eo;
        E[r] = imax(imax(E[r], hm), zcur);
        F = imax(F, hm);
        diag = H[r];
        H[r] = h;
    }
}

// The LAST strip of a query when it needs fewer than R rows: nb = ceil(rows / 8) blocks of eight rows per column, the others are
// skipped by a uniform branch (the last strip hands no boundary row on, so nothing reads the rows that were not computed).  A
// query of 300-400 residues pads to a multiple of 8 instead of 32: 1 % of rows that score nothing instead of 4.4 %.  One extra
// instantiation of the column loop, kept out of the full strips' code: their registers and per-diagonal SGPR constants are untouched
// (round 2's attempts specialised the strip height at compile time and paid for the second set of constants with SGPR spills).
template <int R>
__device__ __forceinline__ void i32d_tail_strip(const uint32_t* res, unsigned len4, const char* prof, const v4i* wsH, const v4i* wsF, int lane,
                                                bool first, unsigned nb_, int& best, int eo, int e, int zrow0) {
    int H[R];
    int E[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { H[r] = zrow0 + (r - 1) * e; E[r] = zrow0 + r * e; }
    int zcol = zrow0;
    int hdiag = zrow0 - 2 * e;
    uint32_t res_n = res[0];
    v4i hb_n = {0, 0, 0, 0}, fb_n = {0, 0, 0, 0};
    if (!first && len4) { hb_n = wsH[lane]; fb_n = wsF[lane]; }
    v4i sc_n[R / 16];
    load_scores<R>(sc_n, prof, res_n, 0);
    for (unsigned c = 0; c < len4; ++c) {
        const uint32_t r4 = res_n;
        const v4i hb = hb_n, fb = fb_n;
        const unsigned cn = (c + 1 < len4) ? c + 1 : c;
        res_n = res[(uint64_t)cn * 64];
        if (!first) { hb_n = wsH[(uint64_t)cn * 64 + lane]; fb_n = wsF[(uint64_t)cn * 64 + lane]; }
        unsigned nb = nb_;
        asm volatile("" : "+s"(nb));                    // opaque per chunk: the loop is not unswitched into one copy per block count
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            v4i sc[R / 16];
#pragma unroll
            for (int k = 0; k < R / 16; ++k) sc[k] = sc_n[k];
            if (cc < 3) load_scores<R>(sc_n, prof, r4, cc + 1); else load_scores<R>(sc_n, prof, res_n, 0);
            int diag = hdiag;
            int F = first ? zcol : fb[cc];
            int zcur = zcol;
            sw_rows8_d<R>(H, E, 0, sc, diag, F, best, eo, zcur, e);
#pragma unroll
            for (int b = 1; b < R / 8; ++b)
                if (nb > (unsigned)b) sw_rows8_d<R>(H, E, 8 * b, sc, diag, F, best, eo, zcur, e);
            hdiag = first ? zcol - e : hb[cc];
            zcol += e;
            asm volatile("" : "+v"(best));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int R>
__global__ __launch_bounds__(kBlock, 3) void sw_scan_i32d(const ScanArgs a) {
    static_assert(R % 16 == 0 && R <= kQueryAlign, "strip height");
    __shared__ __attribute__((aligned(16))) char lds[1024 + kWavesPerBlock * R * 32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    signed char* lmat = reinterpret_cast<signed char*>(lds);
    char* prof = lds + 1024 + wave * (R * 32);

    for (int i = threadIdx.x; i < 256; i += kBlock)
        reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    __syncthreads();
    clk_enter(a.clk);

    const unsigned slot = blockIdx.x * kWavesPerBlock + wave;
    v4i* wsH = reinterpret_cast<v4i*>(a.ws + (uint64_t)slot * a.ws_slot);
    const unsigned n_items = a.n_groups * a.q_count;
    const int e = (int)a.ext;
    int eo = (int)a.ext - (int)a.open;
    asm volatile("" : "+v"(eo));           // in a VGPR: v_add_u32 with two VGPR sources issues a little faster than the SGPR-operand form (tools/valu_issue.hip)
    const int two_e = 2 * e;

    for (;;) {
        unsigned item = 0;
        if (lane == 0) item = atomicAdd(a.counter, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;
        if (a.order) item = a.order[item];
        const unsigned g = item / a.q_count;
        const unsigned qi = item - g * a.q_count;
        const unsigned q = a.q_list ? a.q_list[qi] : a.q_begin + qi;
        const unsigned len4 = a.grp_len4[g];
        const uint32_t* res = a.db_res + a.grp_off[g] + lane;
        const uint8_t* qc = a.q_codes + a.q_off[q];
        const unsigned lq = a.q_len[q];
        item_priority(a.prio_cost, (uint64_t)lq * len4);
        const unsigned nstrips = (lq + R - 1) / R;
        v4i* wsF = wsH + (uint64_t)len4 * 64;
        int best = 0;

        for (unsigned s = 0; s < nstrips; ++s) {
            // ---- strip profile: prof[k][t][b] = M[q_{sR+16k+b}][t] + 2e  (int8) ----
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e0 = 0; e0 < 2 * R; e0 += 64) {
                const int en = e0 + lane;
                const int k = en >> 5, t = en & 31;
                const v4i q16 = *reinterpret_cast<const v4i*>(qc + s * R + 16 * k);
                v4i out;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    unsigned w = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const unsigned code = ((unsigned)q16[d] >> (8 * b)) & 31u;
                        w |= (unsigned)(unsigned char)(lmat[code * 32 + t] + two_e) << (8 * b);
                    }
                    out[d] = (int)w;
                }
                *reinterpret_cast<v4i*>(prof + en * 16) = out;
            }
            __builtin_amdgcn_wave_barrier();

            const bool first = (s == 0), last = (s + 1 == nstrips);
            const int zrow0 = (int)(s * R) * e;                     // z(i0, 0) for the strip's first row i0
            if (last && lq - s * R + 7 < (unsigned)R) {             // a short last strip: only the blocks of eight rows it needs
                i32d_tail_strip<R>(res, len4, prof, wsH, wsF, lane, first, (lq - s * R + 7) / 8, best, eo, e, zrow0);
                break;
            }
            int H[R];
            int E[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { H[r] = zrow0 + (r - 1) * e; E[r] = zrow0 + r * e; }     // H~(i,-1), E~(i,0) for H = E = 0
            int zcol = zrow0;                                        // z(i0, j) of the current column
            int hdiag = zrow0 - 2 * e;                               // H~(i0-1, -1)

            uint32_t res_n = res[0];
            v4i hb_n = {0, 0, 0, 0}, fb_n = {0, 0, 0, 0};
            if (!first && len4) { hb_n = wsH[lane]; fb_n = wsF[lane]; }
            v4i sc_n[R / 16];
            load_scores<R>(sc_n, prof, res_n, 0);
            for (unsigned c = 0; c < len4; ++c) {
                const uint32_t r4 = res_n;
                const v4i hb = hb_n, fb = fb_n;
                const unsigned cn = (c + 1 < len4) ? c + 1 : c;
                res_n = res[(uint64_t)cn * 64];
                if (!first) { hb_n = wsH[(uint64_t)cn * 64 + lane]; fb_n = wsF[(uint64_t)cn * 64 + lane]; }
                v4i ho, fo;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    v4i sc[R / 16];
#pragma unroll
                    for (int k = 0; k < R / 16; ++k) sc[k] = sc_n[k];
                    if (cc < 3) load_scores<R>(sc_n, prof, r4, cc + 1); else load_scores<R>(sc_n, prof, res_n, 0);
                    int hl, fl;
                    sw_column_d<R>(H, E, sc, hdiag, first ? zcol : fb[cc], best, eo, zcol, e, hl, fl);   // first strip: F~(0,j) = z(0,j)
                    hdiag = first ? zcol - e : hb[cc];               // first strip: H~(-1, j) = z(0,j) - e
                    zcol += e;
                    ho[cc] = hl; fo[cc] = fl;
                    asm volatile("" : "+v"(best));
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!last) { wsH[(uint64_t)c * 64 + lane] = ho; wsF[(uint64_t)c * 64 + lane] = fo; }
            }
        }
        const unsigned p = a.grp_perm[(uint64_t)g * 64 + lane];
        if (p != 0xFFFFFFFFu) a.scores[(uint64_t)(q - a.q_begin) * a.score_stride + p] = best;
    }
    clk_leave(a.clk, lane);
}


// =====================================================================================================
// Packed int16 x 2 kernel: the two halves of every 32-bit register carry two DIFFERENT QUERIES against the
// same database sequence (lane).  v_pk_add_u16 / v_pk_max_i16 / v_pk_sub_u16 clamp update both cells per
// instruction: 9 VALU ops per pair of cells = 4.5 per cell (no packed max3 exists, so H costs add+max+max and the
// running maximum one max per row).  Exact by construction when every reachable score fits in int16: the host only
// routes queries here whose upper bound  sum_i max_b M[q_i][b]  is <= 32767 (ScanArgs16 contract); everything else
// goes to sw_scan_i32.  Scores are widened to int32 on output.
//
// Further arithmetic policies (ArithI16M3, ArithF16 below) run the same kernel with gfx950's v_pk_maximum3_f16 as the
// three-input max: 7.5 ops per pair of cells.
//
// Strip profile (wave-private LDS): prof[kk][t][4 rows] of packed (sB << 16 | sA) int16 pairs, 16 B per (kk, t):
// one ds_read_b128 per lane per 4 rows.  Strip boundary rows (H, F of the strip's last row) are packed pairs too:
// 16 B per lane per 4 columns each way for 2 x R cells -> half the HBM bytes per cell of the int32 kernel.
// The last strip of a query pair runs a variant with only as many rows (multiple of 8) as it needs.
// =====================================================================================================
typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2us __attribute__((ext_vector_type(2)));

struct ScanArgs16 {
    const uint32_t* __restrict__ db_res;
    const uint64_t* __restrict__ grp_off;
    const uint32_t* __restrict__ grp_len4;
    const uint32_t* __restrict__ grp_perm;
    const uint8_t* __restrict__ q_codes;
    const uint64_t* __restrict__ q_off;
    const uint32_t* __restrict__ q_len;
    const uint32_t* __restrict__ pairs;     // [2 * n_pairs] query ids (absolute); second may be ~0u (no partner)
    const int8_t* __restrict__ mat;
    int32_t* __restrict__ scores;
    uint64_t score_stride;
    int32_t* __restrict__ ws;
    uint64_t ws_slot;
    unsigned* counter;
    uint32_t n_groups, n_pairs, q_begin;
    uint32_t open, ext;                      // each < 2^15
    uint32_t floor_bits;                     // packed floor of the arithmetic policy (see Arith*)
    uint32_t prio_cost;                      // a quarter of the per-wave share of the launch in rows * len4: item_priority raises a wave's
                                             // priority by its item's size against it (0 = never)
    uint32_t spec_base;                      // sw_scan_pkd, speculative launch: 0x7BFF - B - (max score + 2 ext); 0 = exact launch
    unsigned long long* clk;                 // optional clock samples, 4 per block (clk_enter / clk_leave)
    const uint32_t* __restrict__ order;      // optional: the k-th item drawn from the queue is order[k] -- items by rows x columns, largest first
                                             // (s4g_scan.hip, item_order: launches of few, long items); NULL: group-major, pairs as listed
};

// Written by a speculative sw_scan_pkd launch instead of a score that may not be exact; sw_pair_recheck replaces it.
constexpr int kScoreUnsure = (int)0x80000000;

__device__ __forceinline__ v2s as_s(v2us x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ v2us as_u(v2s x) { return __builtin_bit_cast(v2us, x); }
__device__ __forceinline__ v2s as_s(int x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ v2us as_u(int x) { return __builtin_bit_cast(v2us, x); }
__device__ __forceinline__ int as_i(v2s x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ int as_i(v2us x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ v2s pk_max(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ v2us pk_subs(v2us a, v2us b) { return __builtin_elementwise_sub_sat(a, b); }

typedef _Float16 v2h __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2h as_h(int x) { return __builtin_bit_cast(v2h, x); }
__device__ __forceinline__ int as_i(v2h x) { return __builtin_bit_cast(int, x); }

// ---- arithmetic policies of the packed kernel: every register is a raw 32-bit pair; `fl` = packed floor value ----
// ArithI16: packed int16, E/F clamped at 0 by saturating subtracts.  9 ops per pair of cells.  fl = 0.
struct ArithI16 {
    static constexpr bool kMax3 = false;
    static __device__ __forceinline__ int pen(unsigned p) { return (int)(p | (p << 16)); }
    static __device__ __forceinline__ int add(int a, int b) { return as_i(as_s(a) + as_s(b)); }
    static __device__ __forceinline__ int hmax(int s, int e, int f) { return as_i(pk_max(pk_max(as_s(s), as_s(e)), as_s(f))); }
    static __device__ __forceinline__ int sub(int x, int p) { return as_i(pk_subs(as_u(x), as_u(p))); }
    static __device__ __forceinline__ int gmax(int a, int b, int) { return as_i(pk_max(as_s(a), as_s(b))); }
    static __device__ __forceinline__ int best1(int best, int h) { return as_i(pk_max(as_s(best), as_s(h))); }
    static __device__ __forceinline__ int best2(int best, int h0, int h1) { return best1(best1(best, h0), h1); }
    static __device__ __forceinline__ int pair(int sA, int sB) { return (int)(((unsigned)sA & 0xFFFFu) | ((unsigned)sB << 16)); }
    static __device__ __forceinline__ int out(int best, int half, int) { return (int)as_s(best)[half]; }
};

// v_pk_maximum3_f16 is gfx950's only packed three-input max.  Both policies below emit it from nested
// __builtin_elementwise_maximum (an asm statement would cost an s_nop pad each).
__device__ __forceinline__ int pk_maximum3(int a, int b, int c) {
    return as_i(__builtin_elementwise_maximum(__builtin_elementwise_maximum(as_h(a), as_h(b)), as_h(c)));
}

// ArithI16M3: packed int16 whose three-input max is v_pk_maximum3_f16 applied to the RAW INTEGER BIT PATTERNS.
// For bit patterns 0x0000 .. 0x7BFF (non-negative, below the f16 infinity encoding) the IEEE-754 order of the f16
// values equals the order of the integers, and a max returns one of its inputs bit for bit (kernels run with f16
// denormals preserved: .amdhsa_float_denorm_mode_16_64 3, hipcc's default), so on such operands the instruction IS an
// integer max3.  Every value is kept >= 0 by storing H, E, F with a bias `fl` = B >= max(open, ext, -min score):
// diag+s >= 0, E-ext >= 0, h-open >= 0, and the floor max(., B) replaces the clamp at 0.  Exact while
// bound + B <= 0x7BFF.
//
// The adds and subtracts are PLAIN 32-BIT ONES on the register pair (round 3).  gfx950 issues v_add_u32 / v_sub_u32 with two
// VGPR sources at 32 lanes per clock and SIMD -- twice the rate of every packed, three-operand, SDWA or SGPR-operand form
// (tools/valu_issue.hip, profiles/r03_valu_issue.txt) -- and on these values a 32-bit add IS the packed add: read the pair as
// the integer hi * 65536 + lo; no half ever leaves 0 .. 0xFFFF (every result is >= 0 by the bias and <= 0x7BFF by the window),
// so no carry or borrow crosses the halves.  The substitution scores are paired as the integer sB * 65536 + sA (a negative sA
// borrows from the upper half in advance), the penalties sit in VGPRs (an SGPR operand would halve the rate).  Per pair of
// cells: 4 adds/subs at double rate + 3.5 maximum3 = 5.5 issue slots (7.5 with packed adds).
// Speculative launches (spec_base != 0): a LOWER half that leaves the window may carry into the upper half, so the kernel
// marks the upper result whenever it marks the lower one; the upper half's carries leave the register.
struct ArithI16M3 {
    static constexpr bool kMax3 = true;
    static __device__ __forceinline__ int pen(unsigned p) { int v = (int)(p | (p << 16)); asm volatile("" : "+v"(v)); return v; }   // a VGPR, opaque to the compiler
    static __device__ __forceinline__ int add(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
    static __device__ __forceinline__ int hmax(int s, int e, int f) { return pk_maximum3(s, e, f); }
    static __device__ __forceinline__ int sub(int x, int p) { return (int)((unsigned)x - (unsigned)p); }     // no borrow: x >= B >= p in both halves
    static __device__ __forceinline__ int gmax(int a, int b, int fl) { return pk_maximum3(a, b, fl); }
    static __device__ __forceinline__ int best1(int best, int h) { return pk_maximum3(best, h, h); }
    static __device__ __forceinline__ int best2(int best, int h0, int h1) { return pk_maximum3(best, h0, h1); }
    static __device__ __forceinline__ int pair(int sA, int sB) { return (int)(((unsigned)sB << 16) + (unsigned)sA); }
    static __device__ __forceinline__ int out(int best, int half, int fl) { return (int)as_s(best)[half] - (int)as_s(fl)[0]; }
};

// ArithF16: packed f16 ARITHMETIC on the integer H - 2048 (f16 is exact on integers in [-2048, 2048], so any score up
// to 4096 is exact; values below the floor may round, but rounding is monotone and they are floored away).
// fl = (-2048.0h, -2048.0h).  Same 7.5 ops as ArithI16M3 with a much narrower window; kept as an independent
// implementation of the same tier (selected by S4G_SCAN_F16_ARITH) that does not rely on the bit-pattern argument.
struct ArithF16 {
    static constexpr bool kMax3 = true;
    static __device__ __forceinline__ int pen(unsigned p) { const _Float16 v = (_Float16)(-(float)p); v2h x = {v, v}; return as_i(x); }
    static __device__ __forceinline__ int add(int a, int b) { return as_i(as_h(a) + as_h(b)); }
    static __device__ __forceinline__ int hmax(int s, int e, int f) { return pk_maximum3(s, e, f); }
    static __device__ __forceinline__ int sub(int x, int p) { return add(x, p); }          // p holds -penalty; floored by gmax
    static __device__ __forceinline__ int gmax(int a, int b, int fl) { return pk_maximum3(a, b, fl); }
    static __device__ __forceinline__ int best1(int best, int h) { return pk_maximum3(best, h, h); }
    static __device__ __forceinline__ int best2(int best, int h0, int h1) { return pk_maximum3(best, h0, h1); }
    static __device__ __forceinline__ int pair(int sA, int sB) { v2h x = {(_Float16)(float)sA, (_Float16)(float)sB}; return as_i(x); }
    static __device__ __forceinline__ int out(int best, int half, int) { return (int)(float)as_h(best)[half] + 2048; }
};

// 8 rows (row0 .. row0+7) of one column; sc = their 8 packed score pairs
template <class A, int R>
__device__ __forceinline__ void pk_rows8(int (&H)[R], int (&E)[R], int row0, const v4i (&sc)[2], int& diag, int& F,
                                         int& best, int open, int ext, int fl) {
    int hprev = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = row0 + i;
        const int h = A::hmax(A::add(diag, sc[i / 4][i % 4]), E[r], F);
        if constexpr (A::kMax3) { if (i & 1) best = A::best2(best, hprev, h); else hprev = h; }
        else best = A::best1(best, h);
        const int hm = A::sub(h, open);
        E[r] = A::gmax(A::sub(E[r], ext), hm, fl);
        F = A::gmax(A::sub(F, ext), hm, fl);
        diag = H[r];
        H[r] = h;
    }
}

__device__ __forceinline__ void pk_load8(v4i (&sc)[2], const char* prof, unsigned taddr, int blk) {
    sc[0] = *reinterpret_cast<const v4i*>(prof + (2 * blk) * 512 + taddr);
    sc[1] = *reinterpret_cast<const v4i*>(prof + (2 * blk + 1) * 512 + taddr);
}

__device__ __forceinline__ unsigned pk_taddr(uint32_t r4, int cc) {
    return (cc == 0 ? (r4 << 4) : (r4 >> (8 * cc - 4))) & 0x1F0u;
}

// One strip of RR rows (RR multiple of 8) over all columns of the group.
template <class A, int R, int RR>
__device__ __forceinline__ void pk_strip(const uint32_t* res, unsigned len4, const char* prof, v4i* wsH, v4i* wsF, int lane,
                                         bool first, bool last, int& best, int open, int ext, int fl) {
    constexpr int NB = RR / 8;
    int H[R];
    int E[R];
#pragma unroll
    for (int r = 0; r < RR; ++r) { H[r] = fl; E[r] = fl; }
    int hdiag = fl;
    uint32_t res_n = res[0];
    v4i hb_n = {fl, fl, fl, fl}, fb_n = {fl, fl, fl, fl};
    if (!first && len4) { hb_n = wsH[lane]; fb_n = wsF[lane]; }
    v4i sc_n[2];
    pk_load8(sc_n, prof, pk_taddr(res_n, 0), 0);
    for (unsigned c = 0; c < len4; ++c) {
        const uint32_t r4 = res_n;
        const v4i hb = hb_n, fb = fb_n;
        const unsigned cn = (c + 1 < len4) ? c + 1 : c;
        res_n = res[(uint64_t)cn * 64];
        if (!first) { hb_n = wsH[(uint64_t)cn * 64 + lane]; fb_n = wsF[(uint64_t)cn * 64 + lane]; }
        v4i ho, fo;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const unsigned taddr = pk_taddr(r4, cc);
            const unsigned taddr_next = cc < 3 ? pk_taddr(r4, cc + 1) : pk_taddr(res_n, 0);
            int diag = hdiag;
            int F = fb[cc];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                v4i sc[2] = {sc_n[0], sc_n[1]};
                if (b + 1 < NB) pk_load8(sc_n, prof, taddr, b + 1); else pk_load8(sc_n, prof, taddr_next, 0);
                pk_rows8<A, R>(H, E, 8 * b, sc, diag, F, best, open, ext, fl);
            }
            hdiag = hb[cc];
            ho[cc] = H[RR - 1];
            fo[cc] = F;
            asm volatile("" : "+v"(best));          // pin the running maximum per column (see sw_scan_i32)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!last) { wsH[(uint64_t)c * 64 + lane] = ho; wsF[(uint64_t)c * 64 + lane] = fo; }
    }
}

template <class A, int R>
__global__ __launch_bounds__(kBlock, 3) void sw_scan_pk(const ScanArgs16 a) {
    static_assert(R % 8 == 0 && R <= kQueryAlign, "strip height");
    __shared__ __attribute__((aligned(16))) char lds[1024 + kWavesPerBlock * R * 128];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    signed char* lmat = reinterpret_cast<signed char*>(lds);
    char* prof = lds + 1024 + wave * (R * 128);

    for (int i = threadIdx.x; i < 256; i += kBlock)
        reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    __syncthreads();
    clk_enter(a.clk);

    const unsigned slot = blockIdx.x * kWavesPerBlock + wave;
    v4i* wsH = reinterpret_cast<v4i*>(a.ws + (uint64_t)slot * a.ws_slot);
    const unsigned n_items = a.n_groups * a.n_pairs;
    const int open = A::pen(a.open);
    const int ext = A::pen(a.ext);
    const int fl = (int)a.floor_bits;        // packed floor: 0 (ArithI16), bias B (ArithI16M3), -2048.0h (ArithF16)

    for (;;) {
        unsigned item = 0;
        if (lane == 0) item = atomicAdd(a.counter, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;
        if (a.order) item = a.order[item];
        const unsigned g = item / a.n_pairs;
        const unsigned pi = item - g * a.n_pairs;
        const unsigned qA = a.pairs[2 * pi], qB = a.pairs[2 * pi + 1];
        const bool hasB = qB != 0xFFFFFFFFu;
        const unsigned len4 = a.grp_len4[g];
        const uint32_t* res = a.db_res + a.grp_off[g] + lane;
        const uint8_t* qcA = a.q_codes + a.q_off[qA];
        const uint8_t* qcB = a.q_codes + a.q_off[hasB ? qB : qA];
        const unsigned lA = a.q_len[qA], lB = hasB ? a.q_len[qB] : 0u;
        const unsigned capA = (lA + kQueryAlign - 1) / kQueryAlign * kQueryAlign + kQueryAlign;   // readable bytes
        const unsigned capB = hasB ? (lB + kQueryAlign - 1) / kQueryAlign * kQueryAlign + kQueryAlign : 0u;
        const unsigned lq = lA > lB ? lA : lB;
        const unsigned nstrips = (lq + R - 1) / R;
        v4i* wsF = wsH + (uint64_t)len4 * 64;
        // (an item that is large against the per-wave share of the launch sits on the critical path)
        item_priority(a.prio_cost, (uint64_t)lq * len4);
        int best = fl;

        for (unsigned s = 0; s < nstrips; ++s) {
            const unsigned rows_left = lq - s * R;
            const unsigned rr = rows_left >= (unsigned)R ? (unsigned)R : (rows_left + 7) / 8 * 8;   // rows this strip computes
            // ---- strip profile: prof[kk][t] = 4 rows of packed (query A, query B) scores ----
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e0 = 0; e0 < 8 * R; e0 += 64) {
                const int e = e0 + lane;
                const int kk = e >> 5, t = e & 31;
                const unsigned row0 = s * R + 4 * kk;
                const unsigned cA = row0 < capA ? *reinterpret_cast<const unsigned*>(qcA + row0) : 0x1F1F1F1Fu;
                const unsigned cB = row0 < capB ? *reinterpret_cast<const unsigned*>(qcB + row0) : 0x1F1F1F1Fu;
                v4i out;
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    out[d] = A::pair(lmat[((cA >> (8 * d)) & 31u) * 32 + t], lmat[((cB >> (8 * d)) & 31u) * 32 + t]);
                *reinterpret_cast<v4i*>(prof + e * 16) = out;
            }
            __builtin_amdgcn_wave_barrier();

            const bool first = (s == 0), last = (s + 1 == nstrips);
            bool done = false;
#define S4G_LAST_STRIP(RR_)                                                                                  \
            if constexpr (R > RR_) {                                                                         \
                if (!done && rr == RR_) { pk_strip<A, R, RR_>(res, len4, prof, wsH, wsF, lane, first, last, best, open, ext, fl); done = true; } \
            }
            S4G_LAST_STRIP(8) S4G_LAST_STRIP(16) S4G_LAST_STRIP(24) S4G_LAST_STRIP(32) S4G_LAST_STRIP(40)
#undef S4G_LAST_STRIP
            if (!done) pk_strip<A, R, R>(res, len4, prof, wsH, wsF, lane, first, last, best, open, ext, fl);   // full strip (or rr > 40)
        }
        int outA = A::out(best, 0, fl), outB = A::out(best, 1, fl);
        if constexpr (std::is_same<A, ArithI16M3>::value) {
            // speculative launch of the fixed-frame maximum3 tier (see sw_scan_pkd for the argument; no frame term here):
            // `best` is the biased H + B, and the first value to pass 0x7BFF is a diag + s whose diag was exact and
            // already > spec_base = 0x7BFF - max score, so best <= spec_base => exact
            if (a.spec_base) {
                if (((unsigned)best & 0xFFFFu) > a.spec_base) outA = outB = kScoreUnsure;      // a lower half beyond the window may have carried into the upper one
                if (((unsigned)best >> 16) > a.spec_base) outB = kScoreUnsure;
            }
        }
        const unsigned p = a.grp_perm[(uint64_t)g * 64 + lane];
        if (p != 0xFFFFFFFFu) {
            a.scores[(uint64_t)(qA - a.q_begin) * a.score_stride + p] = outA;
            if (hasB) a.scores[(uint64_t)(qB - a.q_begin) * a.score_stride + p] = outB;
        }
    }
    clk_leave(a.clk, lane);
}


// =====================================================================================================
// sw_scan_pkd: packed int16 pair kernel in a DRIFTING FRAME -- 6.5 VALU ops per pair of cells.
//
// Store every DP value with a bias that grows by `ext` per anti-diagonal:   X~(i,j) = X(i,j) + B + (i+j)*ext.
// Then (o = open, e = ext):
//     H~(i,j)   = max3( H~(i-1,j-1) + (s + 2e),  E~(i,j),  F~(i,j) )            s + 2e comes from the strip profile
//     E~(i,j+1) = max3( E~(i,j),  H~(i,j) + (e - o),  z(i,j+1) )                 z(i,j) = B + (i+j)e  (floor: E >= 0)
//     F~(i+1,j) = max ( F~(i,j),  H~(i,j) + (e - o) )                            F >= -o needs no floor
//     best      = max3( best,  H~ - z(i,j)  [two rows per instruction] )         true H, back in the fixed frame
// Both gap updates lose their "- ext" and share one add.  z(i,j) is uniform over the wave (i and j are), so it
// lives in an SGPR that the scalar unit advances by e per row (SALU is otherwise idle): add, max3, sub, 1/2 max3,
// add, max3, max = 6.5 vector ops.  The three-input max is v_pk_maximum3_f16 on the raw integer bit patterns
// (see ArithI16M3): every operand stays in 0 .. 0x7BFF because B >= max(64, o, 2e, -min score) and the host only
// routes here (query pair, tile) combinations with  bound + B + (rows + columns) * e <= 0x7BFF.
// =====================================================================================================
template <int R>
__device__ __forceinline__ void pkd_rows8(int (&H)[R], int (&E)[R], int row0, const v4i (&sc)[2], int& diag, int& F, int& best,
                                          int eo, int& zcur, int e2) {
    int tprev = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = row0 + i;
        // (plain 32-bit adds on the pair where both sources are VGPRs: they issue a little faster than the packed forms --
        // tools/valu_issue.hip -- and no half ever leaves 0 .. 0xFFFF, so no carry crosses; sc and eo are integer-sum pairs)
        const int h = pk_maximum3((int)((unsigned)diag + (unsigned)sc[i / 4][i % 4]), E[r], F);
        const int t = as_i(as_s(h) - as_s(zcur));                      // true H of both cells (>= 0)
        if (i & 1) best = pk_maximum3(best, tprev, t); else tprev = t;
        zcur += e2;                                                    // scalar: z(i+1, j) = z(i, j+1)
        const int hm = (int)((unsigned)h + (unsigned)eo);
        E[r] = pk_maximum3(E[r], hm, zcur);
        F = as_i(pk_max(as_s(F), as_s(hm)));
        diag = H[r];
        H[r] = h;
    }
}

template <int R, int RR, bool FIRST>
__device__ __forceinline__ void pkd_strip(const uint32_t* res, unsigned len4, const char* prof, v4i* wsH, v4i* wsF, int lane,
                                          bool last, int& best, int eo, int e2, int zrow0) {
    constexpr bool first = FIRST;
    constexpr int NB = RR / 8;
    // zrow0 = packed z(i0, 0) = B + i0*e for the strip's first row i0
    int H[R];
    int E[R];
#pragma unroll
    for (int r = 0; r < RR; ++r) { H[r] = zrow0 + (r - 1) * e2; E[r] = zrow0 + r * e2; }   // H~(i,-1), E~(i,0) for H = E = 0
    int zcol = zrow0;                       // z(i0, j) of the current column
    int hdiag = zrow0 - 2 * e2;             // H~(i0-1, -1)
    uint32_t res_n = res[0];
    v4i hb_n = {0, 0, 0, 0}, fb_n = {0, 0, 0, 0};
    if (!first && len4) { hb_n = wsH[lane]; fb_n = wsF[lane]; }
    v4i sc_n[2];
    pk_load8(sc_n, prof, pk_taddr(res_n, 0), 0);
    for (unsigned c = 0; c < len4; ++c) {
        const uint32_t r4 = res_n;
        const v4i hb = hb_n, fb = fb_n;
        const unsigned cn = (c + 1 < len4) ? c + 1 : c;
        res_n = res[(uint64_t)cn * 64];
        if (!first) { hb_n = wsH[(uint64_t)cn * 64 + lane]; fb_n = wsF[(uint64_t)cn * 64 + lane]; }
        v4i ho, fo;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const unsigned taddr = pk_taddr(r4, cc);
            const unsigned taddr_next = cc < 3 ? pk_taddr(r4, cc + 1) : pk_taddr(res_n, 0);
            int diag = hdiag;
            int F = first ? zcol : fb[cc];           // first strip: F~(0,j) = z(0,j) (F = 0)
            int zcur = zcol;
            // (LLVM hoists the R per-row constants r*e out of the chunk loop into SGPRs; a few of them spill to VGPR lanes and
            // come back by v_readlane.  Forcing a single running SGPR instead -- asm s_add, or an opaque e per column -- removes
            // those reloads but serialises the rows behind SALU->VALU dependencies and measured 1 % SLOWER, A/B on one device.)
            const int e2c = e2;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                v4i sc[2] = {sc_n[0], sc_n[1]};
                if (b + 1 < NB) pk_load8(sc_n, prof, taddr, b + 1); else pk_load8(sc_n, prof, taddr_next, 0);
                pkd_rows8<R>(H, E, 8 * b, sc, diag, F, best, eo, zcur, e2c);
            }
            hdiag = first ? zcol - e2 : hb[cc];      // first strip: H~(-1, j) = z(0,j) - e
            zcol += e2;
            ho[cc] = H[RR - 1];
            fo[cc] = F;
            asm volatile("" : "+v"(best));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!last) { wsH[(uint64_t)c * 64 + lane] = ho; wsF[(uint64_t)c * 64 + lane] = fo; }
    }
}

template <int R>
__global__ __launch_bounds__(kBlock, 3) void sw_scan_pkd(const ScanArgs16 a) {
    static_assert(R % 8 == 0 && R <= kQueryAlign, "strip height");
    __shared__ __attribute__((aligned(16))) char lds[1024 + kWavesPerBlock * R * 128];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    signed char* lmat = reinterpret_cast<signed char*>(lds);
    char* prof = lds + 1024 + wave * (R * 128);
    for (int i = threadIdx.x; i < 256; i += kBlock)
        reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    __syncthreads();
    clk_enter(a.clk);

    const unsigned slot = blockIdx.x * kWavesPerBlock + wave;
    v4i* wsH = reinterpret_cast<v4i*>(a.ws + (uint64_t)slot * a.ws_slot);
    const unsigned n_items = a.n_groups * a.n_pairs;
    const int e2 = (int)(a.ext | (a.ext << 16));
    int eo = ((int)a.ext - (int)a.open) * 65537;                         // the pair (e - o, e - o) as the integer hi * 65536 + lo, in a VGPR
    asm volatile("" : "+v"(eo));
    const int B2 = (int)a.floor_bits;                                   // packed bias B
    const int two_e = 2 * (int)a.ext;

    for (;;) {
        unsigned item = 0;
        if (lane == 0) item = atomicAdd(a.counter, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;
        if (a.order) item = a.order[item];
        const unsigned g = item / a.n_pairs;
        const unsigned pi = item - g * a.n_pairs;
        const unsigned qA = a.pairs[2 * pi], qB = a.pairs[2 * pi + 1];
        const bool hasB = qB != 0xFFFFFFFFu;
        const unsigned len4 = a.grp_len4[g];
        const uint32_t* res = a.db_res + a.grp_off[g] + lane;
        const uint8_t* qcA = a.q_codes + a.q_off[qA];
        const uint8_t* qcB = a.q_codes + a.q_off[hasB ? qB : qA];
        const unsigned lA = a.q_len[qA], lB = hasB ? a.q_len[qB] : 0u;
        const unsigned capA = (lA + kQueryAlign - 1) / kQueryAlign * kQueryAlign + kQueryAlign;
        const unsigned capB = hasB ? (lB + kQueryAlign - 1) / kQueryAlign * kQueryAlign + kQueryAlign : 0u;
        const unsigned lq = lA > lB ? lA : lB;
        const unsigned nstrips = (lq + R - 1) / R;
        v4i* wsF = wsH + (uint64_t)len4 * 64;
        item_priority(a.prio_cost, (uint64_t)lq * len4);   // see sw_scan_pk
        int best = 0;

        for (unsigned s = 0; s < nstrips; ++s) {
            const unsigned rows_left = lq - s * R;
            const unsigned rr = rows_left >= (unsigned)R ? (unsigned)R : (rows_left + 7) / 8 * 8;
            // ---- strip profile: packed (sA + 2e, sB + 2e) ----
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e0 = 0; e0 < 8 * R; e0 += 64) {
                const int e = e0 + lane;
                const int kk = e >> 5, t = e & 31;
                const unsigned row0 = s * R + 4 * kk;
                const unsigned cA = row0 < capA ? *reinterpret_cast<const unsigned*>(qcA + row0) : 0x1F1F1F1Fu;
                const unsigned cB = row0 < capB ? *reinterpret_cast<const unsigned*>(qcB + row0) : 0x1F1F1F1Fu;
                v4i out;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int sA = lmat[((cA >> (8 * d)) & 31u) * 32 + t] + two_e;
                    const int sB = lmat[((cB >> (8 * d)) & 31u) * 32 + t] + two_e;
                    out[d] = (int)(((unsigned)sB << 16) + (unsigned)sA);        // integer-sum pair: a negative sA borrows from sB in advance
                }
                *reinterpret_cast<v4i*>(prof + e * 16) = out;
            }
            __builtin_amdgcn_wave_barrier();

            const bool first = (s == 0), last = (s + 1 == nstrips);
            const int zrow0 = B2 + (int)(s * R) * e2;                  // packed z(i0, 0); no carry between halves (values < 0x7C00)
            bool done = false;
#define S4G_LAST_STRIP(RR_)                                                                                  \
            if constexpr (R > RR_) {                                                                         \
                if (!done && rr == RR_) {                                                                    \
                    if (first) pkd_strip<R, RR_, true>(res, len4, prof, wsH, wsF, lane, last, best, eo, e2, zrow0); \
                    else pkd_strip<R, RR_, false>(res, len4, prof, wsH, wsF, lane, last, best, eo, e2, zrow0);    \
                    done = true;                                                                             \
                }                                                                                            \
            }
            S4G_LAST_STRIP(8) S4G_LAST_STRIP(16) S4G_LAST_STRIP(24) S4G_LAST_STRIP(32) S4G_LAST_STRIP(40)
#undef S4G_LAST_STRIP
            if (!done) {
                if (first) pkd_strip<R, R, true>(res, len4, prof, wsH, wsF, lane, last, best, eo, e2, zrow0);
                else pkd_strip<R, R, false>(res, len4, prof, wsH, wsF, lane, last, best, eo, e2, zrow0);
            }
        }
        int outA = (int)as_s(best)[0], outB = (int)as_s(best)[1];
        if (a.spec_base) {
            // Speculative launch: the host only guarantees that the FRAME fits the window (B + (rows + columns) e + one
            // score step <= 0x7BFF - a span), not that bound + frame does.  Every intermediate value is at most some
            // H~ = H + z <= H + zmax, and the first value to leave the window is a diag + s' whose diag was still exact,
            // so its true H -- which `best` took in before anything went wrong -- exceeds limit = spec_base - zmax.  After
            // that the halves may hold anything (inf / NaN patterns propagate through maximum3, negative patterns lose
            // every max), but as an unsigned 16-bit value best stays > limit.  Hence: best <= limit => exact.
            const unsigned zmax = (lq + 2u * R + len4 * 4u + 8u) * a.ext;
            const unsigned limit = a.spec_base > zmax ? a.spec_base - zmax : 0u;
            if (((unsigned)best & 0xFFFFu) > limit) outA = outB = kScoreUnsure;   // (a lower half outside the window may have carried into the upper one)
            if (((unsigned)best >> 16) > limit) outB = kScoreUnsure;
        }
        const unsigned p = a.grp_perm[(uint64_t)g * 64 + lane];
        if (p != 0xFFFFFFFFu) {
            a.scores[(uint64_t)(qA - a.q_begin) * a.score_stride + p] = outA;
            if (hasB) a.scores[(uint64_t)(qB - a.q_begin) * a.score_stride + p] = outB;
        }
    }
    clk_leave(a.clk, lane);
}

// =====================================================================================================
// sw_scan_pkc: the packed kernel in a COLUMN frame (round 5) -- the same 6.5 vector instructions per pair of cells as
// sw_scan_pkd, but three of them (not two) are two-VGPR 32-bit adds / subtracts, the class gfx950 issues at double rate
// when two waves present one together (tools/valu_issue.hip), and nothing in the row loop depends on the row index.
//
// Store every DP value with a bias that grows by `ext` per COLUMN only:   X~(i,j) = X(i,j) + B + j*ext,  z_j = B + j*ext.
//     H~(i,j)   = max3( H~(i-1,j-1) + (s + e),  E~(i,j),  F~(i,j) )               s + e comes from the strip profile
//     E~(i,j+1) = max3( E~(i,j),  H~(i,j) + (e - o),  z_{j+1} )                   floor E >= 0; z_{j+1} is ONE SGPR per column
//     F~(i+1,j) = max ( F~(i,j),  H~(i,j) + (e - o) )  -  e                       v_sub_u32 v, v, v: double-rate class
//     best~     = max3( best~, H~(i,j), H~(i+1,j) )                               the running maximum lives IN the frame:
//     best~    += e   once per column                                             z_j is uniform over a column's rows
// sw_scan_pkd's per-row "true H = H~ - z(i,j)" (a packed subtract with an SGPR operand, single rate) is gone; its place in
// the count is taken by F's "- e".  Per pair of cells: 3 double-rate (diag + s', H~ + (e - o), F - e) + 3.5 single-rate
// (max3, max3, max, 1/2 max3) against 2 + 4.5.  No per-row frame constants (sw_scan_pkd's r*e SGPRs, some of which spilled to
// v_readlane reloads), and the exactness window loses its row term:  bound + B + columns * e <= 0x7BFF.
// Non-negativity (what makes v_pk_maximum3_f16 an integer max3 and the 32-bit adds packed adds): H~ >= z_j >= B because
// E~ >= z_j; diag + s + e >= z_{j-1} + e + s >= B + min score >= 0; H~ + (e - o) >= B + e - o >= 0; F~ - e >= z_j - o >= 0
// -- with the same B >= max(64, open, 2 ext, -min score) as sw_scan_pkd.
//
// Speculative launches (spec_base != 0; the host guarantees only that the FRAME fits: B + (max score + e) + columns*e + span
// <= 0x7BFF).  The in-frame running maximum is its own witness: its "+= e" is a SIGNED SATURATING packed add (v_pk_add_i16
// clamp: one single-rate instruction per column), so as a 16-bit pattern best~ never decreases inside a strip and never passes
// 0x7FFF -- h is never a negative pattern (E~ >= z_j, finite and positive, is one of its operands, so a wrapped diag + s' loses the
// max or a NaN pattern 0x7C01.. wins it and then propagates through every later maximum3).  Claim: best~ <= spec_base =
// 0x7BFF - (max score + e) at the end of every strip  =>  the result is exact.  The first value ever to leave the window is a
// diag + s' whose diag was exact and > 0x7BFF - (max score + e) -- which best~ took in in the column before and cannot lose -- or
// best~ itself, above spec_base before the add.  `ovf` collects best~ at the end of each strip (best~ is re-based between strips).
// =====================================================================================================
// LDS byte offset of a column's profile slot: (residue code) * 16.  A residue byte holds 0 .. 31, so the byte IS the code and one
// SDWA shift (v_lshlrev_b32_sdwa ... src1_sel:BYTE_n) does it; pk_taddr's shift + mask are two instructions per column.
__device__ __forceinline__ unsigned pkc_taddr(uint32_t r4, int cc) { return ((r4 >> (8 * cc)) & 0xFFu) << 4; }

template <int R>
__device__ __forceinline__ void pkc_rows8(int (&H)[R], int (&E)[R], int row0, const v4i (&sc)[2], int& diag, int& F, int& best,
                                          int eo, int e2v, int zn) {
    int hprev = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = row0 + i;
        const int h = pk_maximum3((int)((unsigned)diag + (unsigned)sc[i / 4][i % 4]), E[r], F);
        if (i & 1) best = pk_maximum3(best, hprev, h); else hprev = h;
        const int hm = (int)((unsigned)h + (unsigned)eo);
        E[r] = pk_maximum3(E[r], hm, zn);                               // zn = z_{j+1}, an SGPR that is constant over the column
        F = (int)((unsigned)as_i(pk_max(as_s(F), as_s(hm))) - (unsigned)e2v);
        diag = H[r];
        H[r] = h;
    }
}

template <int R>
__device__ __forceinline__ void pkc_rows4(int (&H)[R], int (&E)[R], int row0, const v4i& sc, int& diag, int& F, int& best,
                                          int eo, int e2v, int zn) {
    int hprev = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + i;
        const int h = pk_maximum3((int)((unsigned)diag + (unsigned)sc[i]), E[r], F);
        if (i & 1) best = pk_maximum3(best, hprev, h); else hprev = h;
        const int hm = (int)((unsigned)h + (unsigned)eo);
        E[r] = pk_maximum3(E[r], hm, zn);
        F = (int)((unsigned)as_i(pk_max(as_s(F), as_s(hm))) - (unsigned)e2v);
        diag = H[r];
        H[r] = h;
    }
}

// ROWS4: the strip profile's scores are fetched one block of FOUR rows ahead instead of eight (one ds_read_b128 in flight instead of
// two): eight VGPRs less, which is what lets the kernel run at four waves per SIMD (128 VGPRs).
template <int R, int RR, bool FIRST, bool ROWS4>
__device__ __forceinline__ void pkc_strip(const uint32_t* res_u, unsigned len4, const char* prof, v4i* ws, int lane,
                                          bool last, int& best, int eo, int e2v, int e2, int B2) {
    constexpr bool first = FIRST;
    constexpr int NB = RR / 8;
    // res_u and ws are UNIFORM bases: every access is a buffer instruction -- descriptor and chunk offset in SGPRs, the lane's own
    // 32-bit byte offset the only VGPR -- so the chunk's address arithmetic runs on the scalar unit (sw_scan_pkd spends five 64-bit
    // vector adds per chunk on it).
    // Strip-boundary workspace of this kernel: dword [chunk c][H | F][column cc][lane] -- one 256-byte line per (chunk, row kind,
    // column).  A boundary value is loaded for the NEXT chunk into the register it was just read from, and stored as soon as its
    // column is done: no second set of registers for a chunk in flight and no copies between the sets (sw_scan_pkd moves 16-byte
    // vectors per lane and pays six v_mov per chunk for the hand-over), and the loads lead their use by three to four columns.
    // (Round 5 measured two other hand-overs at four waves per SIMD, A/B on one device, EXPERIMENTS.md A.1: a ring of TWO values per
    // row kind loaded two columns ahead -- no spills, 1.3 % slower -- and the next chunk staged through LDS by buffer_load ... lds --
    // no ring at all, 1.1 % slower: its s_waitcnt vmcnt(0) per chunk also waits for the stores just issued.)
    const unsigned lane4 = (unsigned)lane * 4u;
    const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(res_u), 0, (int)(len4 * 256u), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_ws = __builtin_amdgcn_make_buffer_rsrc(ws, 0, (int)(len4 * 2048u), 0x00020000);
    auto ld_res = [&](unsigned c) { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_res, (int)lane4, (int)(c * 256u), 0); };
    auto ld_ws = [&](bool f, unsigned c, int cc) {
        return (int)__builtin_amdgcn_raw_buffer_load_b32(r_ws, (int)lane4, (int)(c * 2048u + (f ? 1024u : 0u) + (unsigned)cc * 256u), 0);
    };
    // (the last strip hands no boundary row on: its stores go through a descriptor of zero records, which the hardware drops --
    // no branch in the column loop)
    const __amdgpu_buffer_rsrc_t r_wst = __builtin_amdgcn_make_buffer_rsrc(ws, 0, last ? 0 : (int)(len4 * 2048u), 0x00020000);
    auto st_ws = [&](bool f, unsigned c, int cc, int v) {
        __builtin_amdgcn_raw_buffer_store_b32((unsigned)v, r_wst, (int)lane4, (int)(c * 2048u + (f ? 1024u : 0u) + (unsigned)cc * 256u), 0);
    };
    int H[R];
    int E[R];
#pragma unroll
    for (int r = 0; r < RR; ++r) { H[r] = B2 - e2; E[r] = B2; }          // H~(i,-1) = z_{-1}, E~(i,0) = z_0 for H = E = 0: the same in every row
    int zj = B2;                            // z_j of the current column (SGPR)
    int hdiag = B2 - e2;                    // H~(i0-1, -1)
    uint32_t res_n = ld_res(0);
    int hbv[4] = {0, 0, 0, 0}, fbv[4] = {0, 0, 0, 0};
    if (!first && len4) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) { if (cc < 3) hbv[cc] = ld_ws(false, 0, cc); fbv[cc] = ld_ws(true, 0, cc); }   // (hbv[3]: after column 0, below)
    }
    v4i sc_n[2];
    unsigned taddr_first = pkc_taddr(res_n, 0);
    if constexpr (ROWS4) sc_n[0] = *reinterpret_cast<const v4i*>(prof + taddr_first);
    else pk_load8(sc_n, prof, taddr_first, 0);
    for (unsigned c = 0; c < len4; ++c) {
        const uint32_t r4 = res_n;
        const unsigned cn = (c + 1 < len4) ? c + 1 : c;
        res_n = ld_res(cn);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const unsigned taddr = cc == 0 ? taddr_first : pkc_taddr(r4, cc);
            const unsigned taddr_next = cc < 3 ? pkc_taddr(r4, cc + 1) : pkc_taddr(res_n, 0);
            if (cc == 3) taddr_first = taddr_next;
            int diag = hdiag;
            int F = first ? zj : fbv[cc];            // first strip: F~(0,j) = z_j (F = 0)
            const int zn = zj + e2;
            if constexpr (ROWS4) {
#pragma unroll
                for (int b = 0; b < 2 * NB; ++b) {
                    const v4i sc = sc_n[0];
                    sc_n[0] = *reinterpret_cast<const v4i*>(prof + (b + 1 < 2 * NB ? (b + 1) * 512 + taddr : taddr_next));
                    pkc_rows4<R>(H, E, 4 * b, sc, diag, F, best, eo, e2v, zn);
                }
            } else {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    v4i sc[2] = {sc_n[0], sc_n[1]};
                    if (b + 1 < NB) pk_load8(sc_n, prof, taddr, b + 1); else pk_load8(sc_n, prof, taddr_next, 0);
                    pkc_rows8<R>(H, E, 8 * b, sc, diag, F, best, eo, e2v, zn);
                }
            }
            hdiag = first ? zj : hbv[cc];            // first strip: H~(-1, j) = z_j (H = 0)
            zj = zn;
            st_ws(false, c, cc, H[RR - 1]); st_ws(true, c, cc, F);
            if (!first) {
                // reload what has been consumed, into the registers it was consumed from: F of this column (its chain is over) for the
                // next chunk, and H of the column BEFORE (its value went into the first row of this one; for column 0 that is column 3
                // of the chunk before, whose successor is this chunk's).
                fbv[cc] = ld_ws(true, cn, cc);
                const int pc = (cc + 3) & 3;
                hbv[pc] = ld_ws(false, cc == 0 ? c : cn, pc);
            }
            best = as_i(__builtin_elementwise_add_sat(as_s(best), as_s(e2)));   // best~ moves to column j+1's frame; saturating: see "speculative launches" above
            asm volatile("" : "+v"(best));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// WPS = waves per SIMD.  Four (round 5): 128 VGPRs -- 8 spilled, two scratch loads and two stores per 4-column chunk in the column
// loop, measured cheaper than every spill-free alternative -- and +2.9 % over three waves A/B: a fourth wave to take the vector unit
// while another waits for LDS, and more pairs of double-rate instructions meeting at the issue stage.
template <int R, int WPS = 4>
__global__ __launch_bounds__(kBlock, WPS) void sw_scan_pkc(const ScanArgs16 a) {
    static_assert(R % 8 == 0 && R <= kQueryAlign, "strip height");
    __shared__ __attribute__((aligned(16))) char lds[1024 + kWavesPerBlock * R * 128];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    signed char* lmat = reinterpret_cast<signed char*>(lds);
    char* prof = lds + 1024 + wave * (R * 128);
    for (int i = threadIdx.x; i < 256; i += kBlock)
        reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    __syncthreads();
    clk_enter(a.clk);

    const unsigned slot = blockIdx.x * kWavesPerBlock + wave;
    v4i* wsH = reinterpret_cast<v4i*>(a.ws + (uint64_t)slot * a.ws_slot);
    const unsigned n_items = a.n_groups * a.n_pairs;
    const int e2 = (int)(a.ext | (a.ext << 16));
    int e2v = e2;                                                        // the same pair in a VGPR: operand of the double-rate subtract / add
    int eo = ((int)a.ext - (int)a.open) * 65537;                         // the pair (e - o, e - o) as the integer hi * 65536 + lo, in a VGPR
    asm volatile("" : "+v"(eo), "+v"(e2v));
    const int B2 = (int)a.floor_bits;                                   // packed bias B
    const int one_e = (int)a.ext;

    for (;;) {
        unsigned item = 0;
        if (lane == 0) item = atomicAdd(a.counter, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;
        if (a.order) item = a.order[item];
        const unsigned g = item / a.n_pairs;
        const unsigned pi = item - g * a.n_pairs;
        const unsigned qA = a.pairs[2 * pi], qB = a.pairs[2 * pi + 1];
        const bool hasB = qB != 0xFFFFFFFFu;
        const unsigned len4 = a.grp_len4[g];
        // (uniform, and told so: the buffer descriptor pkc_strip makes of it must sit in SGPRs, not behind a waterfall loop)
        const uint64_t res_bits = reinterpret_cast<uint64_t>(a.db_res + a.grp_off[g]);
        const uint32_t* res = reinterpret_cast<const uint32_t*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(res_bits >> 32)) << 32) |
                                                                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)res_bits));
        const uint8_t* qcA = a.q_codes + a.q_off[qA];
        const uint8_t* qcB = a.q_codes + a.q_off[hasB ? qB : qA];
        const unsigned lA = a.q_len[qA], lB = hasB ? a.q_len[qB] : 0u;
        const unsigned capA = (lA + kQueryAlign - 1) / kQueryAlign * kQueryAlign + kQueryAlign;
        const unsigned capB = hasB ? (lB + kQueryAlign - 1) / kQueryAlign * kQueryAlign + kQueryAlign : 0u;
        const unsigned lq = lA > lB ? lA : lB;
        const unsigned nstrips = (lq + R - 1) / R;
        item_priority(a.prio_cost, (uint64_t)lq * len4);   // see sw_scan_pk
        const int zend = B2 + (int)(len4 * 4u) * e2;                   // z of the column after the last: where best~ stands when a strip ends
        int best = 0;                                                   // true running maximum (both halves), between strips
        int ovf = 0;

        for (unsigned s = 0; s < nstrips; ++s) {
            const unsigned rows_left = lq - s * R;
            const unsigned rr = rows_left >= (unsigned)R ? (unsigned)R : (rows_left + 7) / 8 * 8;
            // ---- strip profile: packed (sA + e, sB + e) ----
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e0 = 0; e0 < 8 * R; e0 += 64) {
                const int e = e0 + lane;
                const int kk = e >> 5, t = e & 31;
                const unsigned row0 = s * R + 4 * kk;
                const unsigned cA = row0 < capA ? *reinterpret_cast<const unsigned*>(qcA + row0) : 0x1F1F1F1Fu;
                const unsigned cB = row0 < capB ? *reinterpret_cast<const unsigned*>(qcB + row0) : 0x1F1F1F1Fu;
                v4i out;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int sA = lmat[((cA >> (8 * d)) & 31u) * 32 + t] + one_e;
                    const int sB = lmat[((cB >> (8 * d)) & 31u) * 32 + t] + one_e;
                    out[d] = (int)(((unsigned)sB << 16) + (unsigned)sA);        // integer-sum pair: a negative sA borrows from sB in advance
                }
                *reinterpret_cast<v4i*>(prof + e * 16) = out;
            }
            __builtin_amdgcn_wave_barrier();

            const bool first = (s == 0), last = (s + 1 == nstrips);
            int bf = (int)((unsigned)best + (unsigned)B2);              // best~ in column 0's frame
            bool done = false;
#define S4G_LAST_STRIP(RR_)                                                                                  \
            if constexpr (R > RR_) {                                                                         \
                if (!done && rr == RR_) {                                                                    \
                    if (first) pkc_strip<R, RR_, true, WPS >= 4>(res, len4, prof, wsH, lane, last, bf, eo, e2v, e2, B2); \
                    else pkc_strip<R, RR_, false, WPS >= 4>(res, len4, prof, wsH, lane, last, bf, eo, e2v, e2, B2);    \
                    done = true;                                                                             \
                }                                                                                            \
            }
            S4G_LAST_STRIP(8) S4G_LAST_STRIP(16) S4G_LAST_STRIP(24) S4G_LAST_STRIP(32) S4G_LAST_STRIP(40)
#undef S4G_LAST_STRIP
            if (!done) {
                if (first) pkc_strip<R, R, true, WPS >= 4>(res, len4, prof, wsH, lane, last, bf, eo, e2v, e2, B2);
                else pkc_strip<R, R, false, WPS >= 4>(res, len4, prof, wsH, lane, last, bf, eo, e2v, e2, B2);
            }
            ovf = as_i(__builtin_elementwise_max(as_u(ovf), as_u(bf)));
            best = (int)((unsigned)bf - (unsigned)zend);                // back to the true maximum (no borrow: bf >= zend in both halves)
        }
        int outA = (int)as_s(best)[0], outB = (int)as_s(best)[1];
        if (a.spec_base) {
            if (((unsigned)ovf & 0xFFFFu) > a.spec_base) outA = outB = kScoreUnsure;   // (kept from sw_scan_pkd: a marked lower half marks the upper one)
            if (((unsigned)ovf >> 16) > a.spec_base) outB = kScoreUnsure;
        }
        const unsigned p = a.grp_perm[(uint64_t)g * 64 + lane];
        if (p != 0xFFFFFFFFu) {
            a.scores[(uint64_t)(qA - a.q_begin) * a.score_stride + p] = outA;
            if (hasB) a.scores[(uint64_t)(qB - a.q_begin) * a.score_stride + p] = outB;
        }
    }
    clk_leave(a.clk, lane);
}

}  // namespace s4g

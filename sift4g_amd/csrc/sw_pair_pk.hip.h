// sw_pair_pk.hip.h -- the latency path on packed arithmetic: one wave per (query, target) pair, TWO rows per register.
//
// sw_pair_scan (sw_pair.hip.h) gives lane l the RPL consecutive query rows l*RPL .. and walks the target one column per step; a
// wave alone on its SIMD issues one vector instruction per 4 clocks, so a step of RPL = 6 rows at 7.5 int32 instructions per cell plus
// the hand-over costs ~260 clocks (profiles/r03_pair_coop.txt) and a per-query alignDatabase call
// (/root/reference/sift4g/src/database_alignment.cpp:83-86 with that query's candidates) takes as long as its longest candidate
// has columns.  Here the lane's RPL = 2M rows sit in M registers: the LOWER halves hold rows l*RPL + m, the UPPER halves rows
// l*RPL + M + m (m = 0 .. M-1), and the upper halves run ONE COLUMN BEHIND the lower ones -- at step k lane l computes column
// k - 2l in its lower halves and column k - 2l - 1 in its upper halves.  Then everything a cell needs is one step old:
//   * row l*RPL + M (upper half of register 0) takes H and F of the row above from the lower half of register M-1, previous step;
//   * row (l+1)*RPL (lower half of register 0 of the NEXT lane) takes them from the upper half of this lane's register M-1,
//     previous step: one DPP wave_shr:1 and one v_alignbit_b32 put both hand-overs into place at once;
//   * the upper halves' target residue is the lower halves' residue of the previous step.
// A pair takes lt + 2 * ceil(lq / RPL) - 1 steps of M registers per lane (the int32 kernel: lt + ceil(lq / RPL) - 1 steps of RPL).
//
// Arithmetic: sw_scan_pkd's -- the drifting frame X~ = X + B + (m + k) * ext (m = register, k = step: uniform over the wave, so the
// frame values live in SGPRs), v_pk_maximum3_f16 on the integer bit patterns, 32-bit adds on the register pair.  The two halves look
// their residues up separately (as sw_scan_lists does): two tables of 32-bit words per lane, PA[t][m] = s + 2e sign-extended and
// PB[t][m] = (s + 2e) << 16, added to the diagonal by ONE v_add3_u32.  6.5 vector instructions per register and step.  The frame of
// register 0 at step k is (M - 1) * ext behind what register M - 1 produced at step k - 1, so the two hand-over words are re-based
// by that constant (two subtractions per step).  Exact while  bound + B + (M + columns + 2 * 64 + 8) * ext <= 0x7BFF  -- the host
// gives a pair to this kernel only then (s4g_pairs.hip: pk_cap), everything else runs in sw_pair_scan as before.
#pragma once
#include "sw_pair.hip.h"

namespace s4g {

struct PairPkArgs {
    PairArgs p;
    uint32_t bias;              // B: >= max(64, open, 2 ext, -min score)
};

template <int M> struct PkRow { int v[M]; };
template <int M> __device__ __forceinline__ void pin(PkRow<M>& x) {
#pragma unroll
    for (int m = 0; m < M; ++m) asm volatile("" : "+v"(x.v[m]));
}

// Tables of one lane's rows, one 8 KB plane per register: pa[(m * 32 + t) * 64 + lane] = M[q_{lane * 2M + m}][t] + 2e (sign-extended),
// pb[...] = (M[q_{lane * 2M + M + m}][t] + 2e) << 16   (rows beyond the query: pad).  A lane reads and writes only its own bank.
template <int M>
__device__ __forceinline__ void pair_pk_build_profile(int* pa, int* pb, const signed char* lmat, const uint8_t* qc, unsigned lq, int lane, int two_e) {
    __builtin_amdgcn_wave_barrier();
    unsigned ca[M], cb[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const unsigned ia = (unsigned)lane * 2 * M + m, ib = ia + M;
        ca[m] = ia < lq ? (qc[ia] & 31u) : (unsigned)kPadCode;
        cb[m] = ib < lq ? (qc[ib] & 31u) : (unsigned)kPadCode;
    }
#pragma unroll 4
    for (int t = 0; t < 32; ++t) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            pa[(m * 32 + t) * 64 + lane] = (int)lmat[ca[m] * 32 + t] + two_e;
            pb[(m * 32 + t) * 64 + lane] = (int)((unsigned)((int)lmat[cb[m] * 32 + t] + two_e) << 16);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <int M>
__device__ __forceinline__ PkRow<M> pair_pk_load(const int* tab_lane, unsigned t) {
    PkRow<M> r;
#pragma unroll
    for (int m = 0; m < M; ++m) r.v[m] = tab_lane[(m * 32 + (int)t) * 64];
    return r;
}

// stage target columns [w0 - 128, kend + 8) as bytes (index = column - w0 + 128); pad outside [0, lt)
__device__ __forceinline__ void pair_pk_stage_window(char* stage, const uint32_t* res, int len4, unsigned w0, unsigned kend, int lane) {
    __builtin_amdgcn_wave_barrier();
    global_u32_ptr gres = (global_u32_ptr)res;
    const int e_end = (int)((kend - w0 + 128 + 8 + 3) / 4);
    for (int e = lane; e < e_end; e += 64) {
        const int c = (int)(w0 / 4) - 32 + e;
        unsigned v = 0x1F1F1F1Fu;
        if (c >= 0 && c < len4) v = gres[(uint64_t)c * 64];
        reinterpret_cast<unsigned*>(stage)[e] = v;
    }
    __builtin_amdgcn_wave_barrier();
}

template <int M>
__global__ __launch_bounds__(64) void sw_pair_scan_pk(const PairPkArgs b) {
    static_assert(M >= 1 && M <= 3, "two tables of M planes of 8 KB must fit a block's 64 KB of static LDS");
    constexpr int W = kPairWindow;
    constexpr int kTab = M * 32 * 64 * 4;                // bytes of one table
    __shared__ __attribute__((aligned(16))) char lds[1024 + 2 * kTab + W + 128 + 64];
    const PairArgs& a = b.p;
    const int lane = threadIdx.x;
    const signed char* lmat = reinterpret_cast<const signed char*>(lds);
    int* pa = reinterpret_cast<int*>(lds + 1024);
    int* pb = reinterpret_cast<int*>(lds + 1024 + kTab);
    char* stage = lds + 1024 + 2 * kTab;
    for (int i = lane; i < 256; i += 64) reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    __builtin_amdgcn_wave_barrier();
    const int e = (int)a.ext;
    const int e2 = (int)(a.ext | (a.ext << 16));
    int eo = ((int)a.ext - (int)a.open) * 65537;         // the pair (e - o, e - o) as the integer hi * 65536 + lo, in a VGPR
    asm volatile("" : "+v"(eo));
    const int B2 = (int)(b.bias | (b.bias << 16));
    // the frame of register M - 1 at step k - 1, as the row below it sees it, minus the frame of register 0 at step k: (M - 1) e
    const int rebase_s = (M - 1) * e2;
    int rebase = rebase_s;
    asm volatile("" : "+v"(rebase));
    unsigned have_q = 0xFFFFFFFFu;

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
            pair_pk_build_profile<M>(pa, pb, lmat, a.q_codes + a.q_off[q], lq, lane, 2 * e);
        }
        const unsigned lanes_used = (lq + 2 * M - 1) / (2 * M);
        const unsigned nsteps = (lq && lt) ? lt + 2 * lanes_used - 1 : 0;     // the last lane's upper halves reach column lt - 1 at step lt - 1 + 2 (lanes - 1) + 1
        // Frames: register m at step k works in z(m, k) = B + (m + k) e, the same in both halves and in every lane.
        // State before step 0 (true H = E = F = 0 everywhere): H[m] is read as the diagonal of register m + 1, frame z(m, -1);
        // E[m] is E of this step's column, frame z(m, 0).
        int H[M], E[M];
#pragma unroll
        for (int m = 0; m < M; ++m) { H[m] = B2 + (m - 1) * e2; E[m] = B2 + m * e2; }
        int zk = B2;                          // z(0, k), packed, uniform
        int up_prev = B2 - 2 * e2;            // H of the row above register 0 one column back, in the frame a diagonal is read in: z(0, k) - 2e
        int f_out = B2 + (M - 1) * e2;        // F handed down by register M - 1 at step k - 1, frame z(M, k - 1)
        int best = 0;
        const int* pa_lane = pa + lane;
        const int* pb_lane = pb + lane;

        auto step = [&](const PkRow<M>& xa, const PkRow<M>& xb) {
            // ---- hand-over from the step before: lower halves <- lane l-1's upper halves, upper halves <- own lower halves, re-based
            // from the producer's frame to register 0's.  H[M-1] (frame z(M-1, k-1)) becomes the diagonal of step k + 1, read in
            // z(0, k+1) - 2e: minus (M - 1) e.  f_out (frame z(M, k-1)) becomes F of this step, frame z(0, k): minus (M - 1) e.
            // Lane 0's lower halves sit under row -1 (H = F = 0): its DPP move keeps `old`, chosen so that the same arithmetic yields
            // z(0, k+1) - 2e and z(0, k).
            const int oldH = (int)((unsigned)(zk - e2 + rebase_s) << 16), oldF = (int)((unsigned)(zk + rebase_s) << 16);
            const int hl = H[M - 1];
            const int upH = (int)(__builtin_amdgcn_alignbit((unsigned)hl, (unsigned)__builtin_amdgcn_update_dpp(oldH, hl, 0x138, 0xf, 0xf, false), 16) - (unsigned)rebase);
            int F = (int)(__builtin_amdgcn_alignbit((unsigned)f_out, (unsigned)__builtin_amdgcn_update_dpp(oldF, f_out, 0x138, 0xf, 0xf, false), 16) - (unsigned)rebase);
            int diag = up_prev;
            int zcur = zk;
            int tprev = 0;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const int h = pk_maximum3((int)((unsigned)diag + (unsigned)xa.v[m] + (unsigned)xb.v[m]), E[m], F);      // v_add3_u32
                const int t = as_i(as_s(h) - as_s(zcur));                   // true H of both cells (>= 0)
                if (m & 1) best = pk_maximum3(best, tprev, t); else tprev = t;
                zcur += e2;
                const int hm = (int)((unsigned)h + (unsigned)eo);
                E[m] = pk_maximum3(E[m], hm, zcur);
                F = as_i(pk_max(as_s(F), as_s(hm)));
                diag = H[m];
                H[m] = h;
            }
            if (M & 1) best = pk_maximum3(best, tprev, tprev);
            f_out = F;
            up_prev = upH;
            zk += e2;
        };

        for (unsigned w0 = 0; w0 < nsteps; w0 += W) {
            const unsigned kend = nsteps < w0 + W ? nsteps : w0 + W;
            pair_pk_stage_window(stage, res, len4, w0, kend, lane);
            // the lower halves' column at step k is k - 2 lane, the upper halves' column is the lower halves' column of step k - 1
            const unsigned char* sa = reinterpret_cast<const unsigned char*>(stage) + 128 - 2 * lane;
            unsigned tA = sa[0];
            PkRow<M> xaA = pair_pk_load<M>(pa_lane, tA);
            PkRow<M> xbA = pair_pk_load<M>(pb_lane, sa[-1]);
            unsigned tB = sa[1];
            pin(tA); pin(tB); pin(xaA); pin(xbA);
            for (unsigned k = w0; k < kend; k += 2) {           // an odd tail runs one harmless extra step (pad columns)
                PkRow<M> xaB = pair_pk_load<M>(pa_lane, tB);    // step k + 1
                PkRow<M> xbB = pair_pk_load<M>(pb_lane, tA);
                unsigned tC = sa[2];
                __builtin_amdgcn_sched_barrier(0);
                step(xaA, xbA);
                __builtin_amdgcn_sched_barrier(0);
                pin(tC); pin(xaB); pin(xbB);
                xaA = pair_pk_load<M>(pa_lane, tC);             // step k + 2
                xbA = pair_pk_load<M>(pb_lane, tB);
                tA = tC;
                tB = sa[3];
                sa += 2;
                __builtin_amdgcn_sched_barrier(0);
                step(xaB, xbB);
                __builtin_amdgcn_sched_barrier(0);
                pin(tB); pin(xaA); pin(xbA);
            }
        }
        int bs = imax((int)as_s(best)[0], (int)as_s(best)[1]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bs = imax(bs, __shfl_xor(bs, off, 64));
        a.scores[out] = bs;
    }
}

}  // namespace s4g

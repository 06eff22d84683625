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

constexpr int kPairWindow = 4096;      // target columns staged per window
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

// LDS reads issued by hand: the result registers are only valid after the matching lds_wait*, which also pins the
// consumers behind the wait (the value is an in/out operand of the s_waitcnt statement).
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ void lds_u8(unsigned& dst, unsigned addr) { asm volatile("ds_read_u8 %0, %1" : "=v"(dst) : "v"(addr)); }
__device__ __forceinline__ void lds_wait(unsigned& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); }
struct alignas(4) Pack1 { int w[1]; };
struct alignas(8) Pack2 { int w[2]; };
struct alignas(16) Pack4 { int w[4]; };
__device__ __forceinline__ void lds_pack(Pack1& d, unsigned addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(d.w[0]) : "v"(addr)); }
__device__ __forceinline__ void lds_pack(Pack2& d, unsigned addr) {
    typedef int v2i __attribute__((ext_vector_type(2)));
    v2i x;
    asm volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(addr));
    d.w[0] = x[0]; d.w[1] = x[1];
}
__device__ __forceinline__ void lds_pack(Pack4& d, unsigned addr) {
    v4i x;
    asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(addr));
    d.w[0] = x[0]; d.w[1] = x[1]; d.w[2] = x[2]; d.w[3] = x[3];
}
__device__ __forceinline__ void lds_wait_pack(Pack1& d) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d.w[0])); }
__device__ __forceinline__ void lds_wait_pack(Pack2& d) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d.w[0]), "+v"(d.w[1])); }
__device__ __forceinline__ void lds_wait_pack(Pack4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d.w[0]), "+v"(d.w[1]), "+v"(d.w[2]), "+v"(d.w[3]));
}
template <int NW> struct PackOf;
template <> struct PackOf<1> { typedef Pack1 type; };
template <> struct PackOf<2> { typedef Pack2 type; };
template <> struct PackOf<4> { typedef Pack4 type; };

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

    for (;;) {
        unsigned item = 0;
        if (lane == 0) item = atomicAdd(a.counter, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= a.n_pairs) break;
        const unsigned q = a.items[item].q;
        const unsigned tid = a.items[item].t;
        const unsigned out = a.items[item].out;
        const unsigned lq = a.q_len[q];
        const unsigned lt = a.seq_len[tid];
        if (lq == 0 || lt == 0) {
            if (lane == 0) a.scores[out] = 0;
            continue;
        }
        const uint32_t* res = reinterpret_cast<const uint32_t*>(a.seq_addr[tid]);
        const int len4 = (int)((lt + 3) / 4);

        if (q != have_q) {      // lane profile: prof[(t*64 + lane)*NW + w] byte b = M[q_{lane*RPL + 4w + b}][t]
            have_q = q;
            const uint8_t* qc = a.q_codes + a.q_off[q];
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
                vnw out;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    unsigned x = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (4 * w + b < RPL) x |= (((unsigned)rows[4 * w + b][t >> 2] >> (8 * (t & 3))) & 0xFFu) << (8 * b);
                    out.w[w] = (int)x;
                }
                *reinterpret_cast<vnw*>(prof + ((t * 64 + lane) * NW) * 4) = out;
            }
            __builtin_amdgcn_wave_barrier();
        }

        const unsigned lanes_used = (lq + RPL - 1) / RPL;
        const unsigned nsteps = lt + lanes_used - 1;
        int H[RPL];
        unsigned E[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) { H[r] = 0; E[r] = 0; }
        int hu_prev = 0;            // H(row above the block, previous column)
        unsigned f_out = 0;         // F handed to the block below, for this lane's current column
        int best = 0;
        const unsigned prof_addr = lds_addr(prof) + lane * (NW * 4);
        const unsigned stage_addr = lds_addr(stage);
        auto step = [&](const vnw& sc) {
            const int hu = wave_shr1(H[RPL - 1]);       // H(row above the block, this column)
            unsigned F = (unsigned)wave_shr1((int)f_out);
            int d[RPL];
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int w = sc.w[r / 4];
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
            // stage columns [w0 - 64, kend + 8) as bytes; pad outside [0, lt)
            __builtin_amdgcn_wave_barrier();
            const int e_end = (int)((kend - w0 + 64 + 8 + 3) / 4);
            for (int e = lane; e < e_end; e += 64) {
                const int c = (int)(w0 / 4) - 16 + e;
                unsigned v = 0x1F1F1F1Fu;
                if (c >= 0 && c < len4) v = res[(uint64_t)c * 64];
                reinterpret_cast<unsigned*>(stage)[e] = v;
            }
            __builtin_amdgcn_wave_barrier();
            // Software pipeline, two steps per iteration.  The LDS reads are issued by hand (the compiler otherwise
            // rotates the loop into load -> wait -> use and exposes two LDS round trips per step): while a step runs
            // on scA, the scores of the next step (scB) and the residue of the one after (tC) are in flight.
            unsigned sa = stage_addr + 64 - lane;       // LDS address of column k - lane, k = w0
            unsigned tB;
            vnw scA, scB;
            lds_u8(tB, sa);
            lds_wait(tB);
            lds_pack(scA, prof_addr + tB * (64 * NW * 4));
            lds_u8(tB, sa + 1);
            lds_wait(tB);
            lds_wait_pack(scA);
            for (unsigned k = w0; k < kend; k += 2) {       // an odd tail runs one harmless extra step (pad columns)
                unsigned tC;
                lds_pack(scB, prof_addr + tB * (64 * NW * 4));
                lds_u8(tC, sa + 2);
                __builtin_amdgcn_sched_barrier(0);
                step(scA);
                __builtin_amdgcn_sched_barrier(0);
                lds_wait(tC);
                lds_wait_pack(scB);
                lds_pack(scA, prof_addr + tC * (64 * NW * 4));
                lds_u8(tB, sa + 3);
                sa += 2;
                __builtin_amdgcn_sched_barrier(0);
                step(scB);
                __builtin_amdgcn_sched_barrier(0);
                lds_wait(tB);
                lds_wait_pack(scA);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) best = imax(best, __shfl_xor(best, off, 64));
        if (lane == 0) a.scores[out] = best;
    }
}

}  // namespace s4g

// sw_pair_coop.hip.h -- the latency path for LONG pairs: the four waves of a block on ONE (query, target) pair.
//
// A per-query alignDatabase call (/root/reference/sift4g/src/database_alignment.cpp:70-92) scores a few thousand candidates and
// traces a few hundred; both launches take as long as their longest pair does -- one wave of sw_pair_scan / sw_pair_trace that
// walks ~3000 columns with 6 rows per lane, 0.3-0.4 ms, while the chip idles (traced: profiles/r03_dropin/).  Here the query's
// rows are cut into bands of 64 * RPL rows, one band per wave of the block, RPL = 2..4: wave b sweeps band b exactly as
// sw_pair_scan_band sweeps its bands -- lane 0 takes H and F of the row above from the boundary row, lane 63 leaves its own for the
// band below -- only that the bands run CONCURRENTLY: the boundary row goes through a per-block workspace in HBM (L2-resident:
// 8 bytes per column), wave b publishes how many steps it has finished in an LDS word after every window of kCoopWindow steps,
// and wave b + 1 waits on that word before it stages the boundary of its next window.  A wave thus trails its predecessor by
// ~200 steps, and a 350-row query against 3000 columns takes (3000 + 63 + 2 * 200) steps of 2 rows per lane instead of 3058
// steps of 6.  Same arithmetic, same tie rules, same direction codes as sw_pair_scan / sw_pair_trace (int32, exact for every
// scorer).  What it buys, measured: nothing for scores -- a step of 2 rows waits for its two dependent LDS reads as long as a step of
// 6 rows computes -- and a factor 1.8 for the traceback of a very long target (s4g_pairs.hip, CoopRule, has the numbers and the
// rule: by default only tracebacks of a millisecond or more come here).
#pragma once
#include "sw_pair.hip.h"

namespace s4g {

constexpr int kCoopWaves = 4;          // waves (= bands) of a block; a pair uses ceil(lq / (64 * RPL)) of them
constexpr int kCoopWindow = 128;       // steps between two publications of a wave's progress

struct PairCoopArgs {
    PairArgs p;
    int2* __restrict__ ws;             // [grid][kCoopWaves - 1][ws_slot] boundary rows (H, F) between consecutive bands
    uint64_t ws_slot;                  // columns per boundary row (>= the longest target of the launch)
};
struct PairTraceCoopArgs {
    PairTraceArgs p;
    int2* __restrict__ ws;
    uint64_t ws_slot;
};

struct CoopSync {
    unsigned item;
    unsigned prog[kCoopWaves];         // steps wave b has finished and made visible (of the current item)
    int best[kCoopWaves];
    unsigned key[kCoopWaves];
};

template <int RPL> struct CoopLayout {
    static constexpr int NW = RPL <= 4 ? 1 : (RPL <= 8 ? 2 : 4);
    static constexpr int kProf = 32 * 64 * NW * 4;
    static constexpr int kStage = kCoopWindow + 192;
    static constexpr int kBnd = (kCoopWindow + 64) * 8;
    static constexpr int kWave = kProf + kStage + kBnd;
    static_assert(kWave % 16 == 0 && (kProf + kStage) % 8 == 0, "LDS regions of a wave keep their alignment");
};

// wave-uniform wait until the band above has published `need` steps; then its boundary stores are visible to this wave's loads
__device__ __forceinline__ void coop_wait(const unsigned* prog, unsigned need) {
    for (;;) {
        const unsigned v = __hip_atomic_load(prog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= need) break;
        __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void coop_publish(unsigned* prog, unsigned steps, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_store(prog, steps, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int RPL>
__global__ __launch_bounds__(kCoopWaves * 64) void sw_pair_scan_coop(const PairCoopArgs b) {
    typedef CoopLayout<RPL> L;
    constexpr int NW = L::NW, W = kCoopWindow, BAND = 64 * RPL;
    typedef typename PackOf<NW>::type vnw;
    const PairArgs& a = b.p;
    __shared__ __attribute__((aligned(16))) char lds[1024 + kCoopWaves * L::kWave];
    __shared__ CoopSync sy;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int* lmat4 = reinterpret_cast<const int*>(lds);
    char* prof = lds + 1024 + wave * L::kWave;
    char* stage = prof + L::kProf;
    int2* bnd = reinterpret_cast<int2*>(stage + L::kStage);
    for (int i = threadIdx.x; i < 256; i += kCoopWaves * 64) reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    const unsigned open = a.open, ext = a.ext;
    int2* const ws_block = b.ws + (uint64_t)blockIdx.x * (kCoopWaves - 1) * b.ws_slot;
    const int2* ws_in = ws_block + (uint64_t)(wave > 0 ? wave - 1 : 0) * b.ws_slot;
    int2* ws_out = ws_block + (uint64_t)(wave < kCoopWaves - 1 ? wave : 0) * b.ws_slot;
    const char* prof_lane = prof + lane * (NW * 4);

    // canonical loop shape (fetch in the header, no break / continue inside), see sw_pair_scan_band
    auto next_item = [&]() -> unsigned {
        if (threadIdx.x == 0) sy.item = atomicAdd(a.counter, 1u);
        if (threadIdx.x < kCoopWaves) sy.prog[threadIdx.x] = 0;
        __syncthreads();
        return (unsigned)__builtin_amdgcn_readfirstlane((int)sy.item);
    };
    for (unsigned item = next_item(); item < a.n_pairs; item = next_item()) {
        const unsigned q = a.items[item].q;
        const unsigned tid = a.items[item].t;
        const unsigned out = a.items[item].out;
        const unsigned lq = a.q_len[q];
        const unsigned lt = a.seq_len[tid];
        const uint32_t* res = reinterpret_cast<const uint32_t*>(a.seq_addr[tid]);
        const int len4 = (int)((lt + 3) / 4);
        const unsigned nbands = lt ? (lq + BAND - 1) / BAND : 0;
        int best = 0;
        if ((unsigned)wave < nbands) {
            const unsigned row0 = (unsigned)wave * BAND;
            const unsigned rows = lq - row0 < (unsigned)BAND ? lq - row0 : (unsigned)BAND;
            const bool last_band = (unsigned)wave + 1 == nbands;
            pair_build_profile<RPL, NW>(prof, lmat4, a.q_codes + a.q_off[q] + row0, rows, lane);
            const unsigned lanes_used = (rows + RPL - 1) / RPL;
            const unsigned nsteps = lt + lanes_used - 1;
            const unsigned nsteps_above = lt + 63;              // (every band but the last is full)
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
                if (RPL & 1) best = imax(best, hprev);
                f_out = F;
                hu_prev = hu;
            };
            for (unsigned w0 = 0; w0 < nsteps; w0 += W) {
                const unsigned kend = nsteps < w0 + W ? nsteps : w0 + W;
                if (wave) {                 // the boundary of columns w0 .. kend + 2: the band above is past them at step column + 63
                    const unsigned need = kend + 2 + 64 < nsteps_above ? kend + 2 + 64 : nsteps_above;
                    coop_wait(&sy.prog[wave - 1], need);
                }
                pair_stage_window(stage, res, len4, w0, kend, lane);
                for (unsigned e = lane; e < kend - w0 + 3; e += 64) {
                    int2 v = {0, 0};
                    if (wave && w0 + e < lt) v = ws_in[w0 + e];
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
                    if (!last_band && lane == 63 && k >= 63 && k - 63 < lt) ws_out[k - 63] = int2{H[RPL - 1], (int)f_out};
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
                    if (k + 1 < kend) {         // (a window is a unit of publication: no step of the next one runs early)
                        step(scB, topB);
                        if (!last_band && lane == 63 && k + 1 >= 63 && k + 1 - 63 < lt) ws_out[k + 1 - 63] = int2{H[RPL - 1], (int)f_out};
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    pin(tB);
                    pin(scA);
                    pin(topA.x); pin(topA.y);
                }
                if (!last_band) coop_publish(&sy.prog[wave], kend, lane);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) best = imax(best, __shfl_xor(best, off, 64));
        }
        if (lane == 0) sy.best[wave] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            int m = 0;
            for (int w = 0; w < kCoopWaves; ++w) m = imax(m, sy.best[w]);
            a.scores[out] = m;
        }
    }
}

// Traceback, the same way: band b's direction block is b * band_bytes into the pair's block, band_bytes = (lt + 63 rounded up to
// even) * 64 * NW * 4 -- inside a band the layout of sw_pair_trace<RPL>; after the sweep one lane walks the codes back across bands.
template <int RPL>
__global__ __launch_bounds__(kCoopWaves * 64) void sw_pair_trace_coop(const PairTraceCoopArgs b) {
    typedef CoopLayout<RPL> L;
    constexpr int NW = L::NW, W = kCoopWindow, BAND = 64 * RPL;
    typedef typename PackOf<NW>::type vnw;
    const PairTraceArgs& a = b.p;
    __shared__ __attribute__((aligned(16))) char lds[1024 + kCoopWaves * L::kWave];
    __shared__ CoopSync sy;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int* lmat4 = reinterpret_cast<const int*>(lds);
    char* prof = lds + 1024 + wave * L::kWave;
    char* stage = prof + L::kProf;
    int2* bnd = reinterpret_cast<int2*>(stage + L::kStage);
    for (int i = threadIdx.x; i < 256; i += kCoopWaves * 64) reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    const unsigned open = a.open, ext = a.ext;
    int2* const ws_block = b.ws + (uint64_t)blockIdx.x * (kCoopWaves - 1) * b.ws_slot;
    const int2* ws_in = ws_block + (uint64_t)(wave > 0 ? wave - 1 : 0) * b.ws_slot;
    int2* ws_out = ws_block + (uint64_t)(wave < kCoopWaves - 1 ? wave : 0) * b.ws_slot;
    const char* prof_lane = prof + lane * (NW * 4);

    // canonical loop shape (fetch in the header, no break / continue inside), see sw_pair_scan_band
    auto next_item = [&]() -> unsigned {
        if (threadIdx.x == 0) sy.item = atomicAdd(a.counter, 1u);
        if (threadIdx.x < kCoopWaves) sy.prog[threadIdx.x] = 0;
        __syncthreads();
        return (unsigned)__builtin_amdgcn_readfirstlane((int)sy.item);
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
        if ((unsigned)wave < nbands) {
            const unsigned row0 = (unsigned)wave * BAND;
            const unsigned rows = lq - row0 < (unsigned)BAND ? lq - row0 : (unsigned)BAND;
            const bool last_band = (unsigned)wave + 1 == nbands;
            pair_build_profile<RPL, NW>(prof, lmat4, a.q_codes + a.q_off[q] + row0, rows, lane);
            const unsigned lanes_used = (rows + RPL - 1) / RPL;
            const unsigned nsteps = lt + lanes_used - 1;
            const unsigned nsteps_above = lt + 63;
            int H[RPL];
            unsigned E[RPL];
#pragma unroll
            for (int r = 0; r < RPL; ++r) { H[r] = 0; E[r] = 0; }
            int hu_prev = 0;
            unsigned f_out = 0;         // F for the block below | "F opened here" << 31
            unsigned eo4[RPL];
#pragma unroll
            for (int r = 0; r < RPL; ++r) eo4[r] = 0;
            vnw* dst = reinterpret_cast<vnw*>(a.dirs + it.dir_off + (uint64_t)wave * band_bytes) + lane;
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
                if (wave) {
                    const unsigned need = kend + 2 + 64 < nsteps_above ? kend + 2 + 64 : nsteps_above;
                    coop_wait(&sy.prog[wave - 1], need);
                }
                pair_stage_window(stage, res, len4, w0, kend, lane);
                for (unsigned e = lane; e < kend - w0 + 3; e += 64) {
                    int2 v = {0, 0};
                    if (wave && w0 + e < lt) v = ws_in[w0 + e];
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
                    if (!last_band && lane == 63 && k >= 63 && k - 63 < lt) ws_out[k - 63] = int2{H[RPL - 1], (int)f_out};
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
                    if (k + 1 < kend) {
                        step(scB, topB);
                        if (!last_band && lane == 63 && k + 1 >= 63 && k + 1 - 63 < lt) ws_out[k + 1 - 63] = int2{H[RPL - 1], (int)f_out};
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    pin(tB);
                    pin(scA);
                    pin(topA.x); pin(topA.y);
                }
                if (!last_band) coop_publish(&sy.prog[wave], kend, lane);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                best = imax(best, __shfl_xor(best, off, 64));
                const unsigned other = (unsigned)__shfl_xor((int)key, off, 64);
                key = other < key ? other : key;
            }
        }
        if (lane == 0) { sy.best[wave] = best; sy.key[wave] = key; }
        // the codes were written by all waves of the block; make them visible to the one lane that walks them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (wave == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int w = 0; w < kCoopWaves; ++w) { best = imax(best, sy.best[w]); key = sy.key[w] < key ? sy.key[w] : key; }
            if (lq == 0 || lt == 0 || want <= 0) { if (lane == 0) { out[0] = out[1] = out[2] = out[3] = -1; out[4] = 0; a.best[it.slot] = 0; } }
            else {
                if (lane == 0) a.best[it.slot] = best;
                if (key == 0xFFFFFFFFu) { if (lane == 0) { out[0] = out[1] = out[2] = out[3] = -1; out[4] = 0; } }
                else pair_walk<RPL, NW, BAND>(a.dirs + it.dir_off, band_bytes, a.paths + it.path_end, key, lq + lt, out, prof, lane);
            }
        }
    }
}

}  // namespace s4g

// sw_trace.hip.h -- device traceback of the survivors (step 4 of swsharp's alignDatabase as sift4g uses it:
// /root/reference/sift4g/src/database_alignment.cpp:83-86 returns DbAlignments whose paths
// select_alignments.cpp:244-300 walks with MOVE_DIAG / MOVE_LEFT / MOVE_UP).
//
// Two kernels:
//   sw_trace_fill  one wave per item = (query, up to 64 of its surviving targets, one per lane).  Same strip-tiled
//                  int32 DP as sw_scan_i32 (shared strip profile in LDS, boundary rows through HBM) but every cell
//                  also emits a 4-bit direction code (1 byte per cell, 4 columns per dword, lane-contiguous stores)
//                  and the lane tracks the FIRST cell in row-major order whose H equals the known score.
//   sw_trace_walk  one lane per pair: follows the direction bytes back from the end cell and writes the move string.
// Tie rules are those of the host traceback in libswsharp and of the oracle (unpinned assumption):
// end cell = smallest i then smallest j among the maxima; in H prefer DIAG, then UP (F), then LEFT (E); inside a gap
// run, "opened here" wins over "extended" when both reproduce the value.
#pragma once
#include "sw_kernels.hip.h"

namespace s4g {

constexpr int kTraceR = 32;
enum : unsigned { TB_STOP = 0, TB_DIAG = 1, TB_UP = 2, TB_LEFT = 3, TB_E_OPENS = 4, TB_F_OPENS = 8 };
enum : unsigned char { MV_DIAG = 1, MV_LEFT = 2, MV_UP = 3 };   // swsharp MOVE_* values (include/swsharp/swsharp.h)

struct TraceArgs {
    const uint64_t* __restrict__ lane_res;      // [items*64] address of the lane's first residue dword (stride 64 dwords per 4 columns)
    const uint32_t* __restrict__ lane_len;      // [items*64] target length, 0 = empty lane
    const int32_t* __restrict__ lane_score;     // [items*64] score of the pair (from the scan)
    const uint32_t* __restrict__ item_q;        // [items] query id
    const uint32_t* __restrict__ item_len4;     // [items] 4-column chunks to process (max over lanes)
    const uint64_t* __restrict__ item_dir_off;  // [items] dword offset of the item's direction block
    const uint8_t* __restrict__ q_codes;
    const uint64_t* __restrict__ q_off;
    const uint32_t* __restrict__ q_len;
    const int8_t* __restrict__ mat;
    uint32_t* __restrict__ dirs;                // direction bytes: dword ((s*len4 + c)*R + r)*64 + lane holds columns 4c..4c+3 of row s*R+r
    int32_t* __restrict__ ws;
    uint64_t ws_slot;
    unsigned* counter;
    uint32_t n_items;
    uint32_t open, ext;
    uint32_t* __restrict__ end_key;             // [items*64] (i << 16 | j) of the end cell, ~0u if none
    int32_t* __restrict__ best;                 // [items*64] recomputed best score (cross-check)
    uint32_t left_first;                        // ties between the gap sources of H: 0 = UP before LEFT, 1 = LEFT before UP (S4G_TRACEBACK_TIES)
};

__global__ __launch_bounds__(kBlock, 2) void sw_trace_fill(const TraceArgs a) {
    constexpr int R = kTraceR;
    __shared__ __attribute__((aligned(16))) char lds[1024 + kWavesPerBlock * R * 32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    signed char* lmat = reinterpret_cast<signed char*>(lds);
    char* prof = lds + 1024 + wave * (R * 32);
    for (int i = threadIdx.x; i < 256; i += kBlock)
        reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    __syncthreads();
    const unsigned slot = blockIdx.x * kWavesPerBlock + wave;
    v4i* wsH = reinterpret_cast<v4i*>(a.ws + (uint64_t)slot * a.ws_slot);
    const unsigned open = a.open, ext = a.ext;

    for (;;) {
        unsigned item = 0;
        if (lane == 0) item = atomicAdd(a.counter, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= a.n_items) break;
        const unsigned q = a.item_q[item];
        const unsigned len4 = a.item_len4[item];
        const uint32_t* res = reinterpret_cast<const uint32_t*>(a.lane_res[(uint64_t)item * 64 + lane]);
        const unsigned my_len4 = (a.lane_len[(uint64_t)item * 64 + lane] + 3) / 4;
        const int want = a.lane_score[(uint64_t)item * 64 + lane];
        const uint8_t* qc = a.q_codes + a.q_off[q];
        const unsigned lq = a.q_len[q];
        const unsigned nstrips = (lq + R - 1) / R;
        v4i* wsF = wsH + (uint64_t)len4 * 64;
        uint32_t* dirs = a.dirs + a.item_dir_off[item] + lane;
        int best = 0;
        unsigned key = 0xFFFFFFFFu;

        for (unsigned s = 0; s < nstrips; ++s) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e0 = 0; e0 < 2 * R; e0 += 64) {     // strip profile, as in sw_scan_i32
                const int e = e0 + lane;
                const int k = e >> 5, t = e & 31;
                const v4i q16 = *reinterpret_cast<const v4i*>(qc + s * R + 16 * k);
                v4i out;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    unsigned w = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        w |= (unsigned)(unsigned char)lmat[(((unsigned)q16[d] >> (8 * b)) & 31u) * 32 + t] << (8 * b);
                    out[d] = (int)w;
                }
                *reinterpret_cast<v4i*>(prof + e * 16) = out;
            }
            __builtin_amdgcn_wave_barrier();

            int H[R];
            unsigned E[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { H[r] = 0; E[r] = 0; }
            unsigned eopen = 0;      // bit r: E of (row r, current column) was opened from H(row r, previous column)
            int hdiag = 0;
            const bool first = (s == 0), last = (s + 1 == nstrips);
            for (unsigned c = 0; c < len4; ++c) {
                const uint32_t r4 = c < my_len4 ? res[(uint64_t)c * 64] : 0x1F1F1F1Fu;
                v4i hb = {0, 0, 0, 0}, fb = {0, 0, 0, 0};
                if (!first) { hb = wsH[(uint64_t)c * 64 + lane]; fb = wsF[(uint64_t)c * 64 + lane]; }
                unsigned D[R];
#pragma unroll
                for (int r = 0; r < R; ++r) D[r] = 0;
                v4i ho, fo;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    v4i sc[R / 16];
                    load_scores<R>(sc, prof, r4, cc);
                    int diag = hdiag;
                    unsigned F = (unsigned)fb[cc] & 0x7FFFFFFFu;
                    unsigned fopen = (unsigned)fb[cc] >> 31;
                    unsigned eopen_next = 0;
                    const unsigned jcol = 4 * c + cc;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int w = sc[r / 16][(r % 16) / 4];
                        const int sv = (int)(signed char)(w >> (8 * (r % 4)));
                        const int d = diag + sv;
                        const int h = imax(imax(d, (int)E[r]), (int)F);
                        const unsigned src = h == 0 ? TB_STOP : (h == d ? TB_DIAG : (a.left_first ? (h == (int)E[r] ? TB_LEFT : TB_UP) : (h == (int)F ? TB_UP : TB_LEFT)));
                        D[r] |= (src | (((eopen >> r) & 1u) << 2) | (fopen << 3)) << (8 * cc);
                        if (h == want && h > 0) {
                            const unsigned kcell = ((s * R + r) << 16) | jcol;
                            key = kcell < key ? kcell : key;
                        }
                        best = imax(best, h);
                        const unsigned hm = usub_sat((unsigned)h, open);
                        const unsigned En = umax_s(usub_sat(E[r], ext), hm);
                        const unsigned Fn = umax_s(usub_sat(F, ext), hm);
                        eopen_next |= (En == hm ? 1u : 0u) << r;
                        fopen = (Fn == hm) ? 1u : 0u;
                        E[r] = En;
                        F = Fn;
                        diag = H[r];
                        H[r] = h;
                    }
                    eopen = eopen_next;
                    hdiag = hb[cc];
                    ho[cc] = H[R - 1];
                    fo[cc] = (int)(F | (fopen << 31));
                }
                uint32_t* dst = dirs + ((uint64_t)(s * len4 + c) * R) * 64;
#pragma unroll
                for (int r = 0; r < R; ++r) dst[(uint64_t)r * 64] = D[r];
                if (!last) { wsH[(uint64_t)c * 64 + lane] = ho; wsF[(uint64_t)c * 64 + lane] = fo; }
            }
        }
        a.end_key[(uint64_t)item * 64 + lane] = key;
        a.best[(uint64_t)item * 64 + lane] = best;
    }
}

struct WalkArgs {
    const uint32_t* __restrict__ dirs;
    const uint64_t* __restrict__ item_dir_off;
    const uint32_t* __restrict__ item_len4;
    const uint32_t* __restrict__ end_key;       // [items*64]
    const uint32_t* __restrict__ slot_of_pair;  // [pairs] item*64 + lane
    const uint64_t* __restrict__ path_off;      // [pairs+1] byte offsets into paths
    uint8_t* __restrict__ paths;
    int32_t* __restrict__ out;                  // [pairs][5]: q_start, q_end, t_start, t_end, path_len
    uint32_t n_pairs;
};

__global__ __launch_bounds__(64) void sw_trace_walk(const WalkArgs a) {
    constexpr int R = kTraceR;
    const unsigned p = blockIdx.x * 64 + threadIdx.x;
    if (p >= a.n_pairs) return;
    const unsigned slot = a.slot_of_pair[p];
    const unsigned item = slot >> 6, lane = slot & 63;
    const unsigned key = a.end_key[slot];
    int32_t* out = a.out + (uint64_t)p * 5;
    if (key == 0xFFFFFFFFu) { out[0] = out[1] = out[2] = out[3] = -1; out[4] = 0; return; }
    const unsigned char* dirs = reinterpret_cast<const unsigned char*>(a.dirs + a.item_dir_off[item]);
    const unsigned len4 = a.item_len4[item];
    uint8_t* path = a.paths + a.path_off[p];
    const unsigned cap = (unsigned)(a.path_off[p + 1] - a.path_off[p]);
    int i = (int)(key >> 16), j = (int)(key & 0xFFFFu);
    const int ei = i, ej = j;
    unsigned n = 0, state = 0;
    while (i >= 0 && j >= 0 && n < cap) {
        const unsigned s = (unsigned)i / R, r = (unsigned)i % R, c = (unsigned)j >> 2, cc = (unsigned)j & 3;
        const unsigned b = dirs[((((uint64_t)s * len4 + c) * R + r) * 64 + lane) * 4 + cc];
        if (state == 0) {
            const unsigned src = b & 3u;
            if (src == TB_STOP) break;
            if (src == TB_DIAG) { path[n++] = MV_DIAG; --i; --j; }
            else state = (src == TB_UP) ? 2u : 1u;
        } else if (state == 1) {
            path[n++] = MV_LEFT;
            if (b & TB_E_OPENS) state = 0;
            --j;
        } else {
            path[n++] = MV_UP;
            if (b & TB_F_OPENS) state = 0;
            --i;
        }
    }
    for (unsigned x = 0, y = n ? n - 1 : 0; x < y; ++x, --y) { const uint8_t t = path[x]; path[x] = path[y]; path[y] = t; }
    out[0] = i + 1; out[1] = ei; out[2] = j + 1; out[3] = ej; out[4] = (int)n;
}

}  // namespace s4g

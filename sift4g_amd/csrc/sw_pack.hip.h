// sw_pack.hip.h -- device-side packing of a database tile (the `chainDatabaseCreate` side of the boundary,
// /root/reference/sift4g/src/database_alignment.cpp:80): raw residue codes as the caller holds them (one byte per
// residue, sequences back to back) -> the length-sorted, lane-transposed tile layout the scan kernels read
// (sw_kernels.hip.h: dword grp_off[g] + c*64 + lane = columns 4c..4c+3 of the target in `lane`, padded with kPadCode).
// The host only sorts the lengths and cuts the tiles; the byte transposition -- what used to dominate s4g_db_create --
// runs here.  One block per group; thread (cc, lane) builds the dwords c = cc, cc+4, ... of its lane: lane-contiguous
// 256-B stores, byte gathers from the raw sequence (L2 resident: every byte is read once).
#pragma once
#include "sw_kernels.hip.h"

namespace s4g {

struct PackArgs {
    const uint8_t* __restrict__ raw;        // raw codes of the whole database
    const uint64_t* __restrict__ src_off;   // [G*64] offset of the lane's sequence in raw
    const uint32_t* __restrict__ src_len;   // [G*64] its length (0 = empty lane)
    const uint64_t* __restrict__ grp_off;   // [G]
    const uint32_t* __restrict__ grp_len4;  // [G]
    uint32_t* __restrict__ res;             // tile residues (out)
    int* bad;                               // set to 1 if any code is >= 26
};

__global__ __launch_bounds__(256) void sw_pack_tile(const PackArgs a) {
    const unsigned g = blockIdx.x;
    const unsigned lane = threadIdx.x & 63, cc = threadIdx.x >> 6;
    const unsigned len4 = a.grp_len4[g];
    const unsigned len = a.src_len[(uint64_t)g * 64 + lane];
    const uint8_t* src = a.raw + a.src_off[(uint64_t)g * 64 + lane];
    uint32_t* dst = a.res + a.grp_off[g] + lane;
    unsigned any_bad = 0;
    for (unsigned c = cc; c < len4; c += 4) {
        unsigned w = 0x1F1F1F1Fu;
        const unsigned j0 = 4 * c;
        if (j0 + 4 <= len) {
            const unsigned b0 = src[j0], b1 = src[j0 + 1], b2 = src[j0 + 2], b3 = src[j0 + 3];
            w = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            any_bad |= (w | ((w & 0x7F7F7F7Fu) + 0x66666666u)) & 0x80808080u;
        } else {
            for (unsigned j = j0; j < len; ++j) {
                const unsigned b = src[j];
                any_bad |= b >= 26u ? 1u : 0u;
                w = (w & ~(0xFFu << (8 * (j & 3)))) | (b << (8 * (j & 3)));
            }
        }
        dst[(uint64_t)c * 64] = w;
    }
    if (any_bad) atomicOr(a.bad, 1);
}

}  // namespace s4g

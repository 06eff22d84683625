// sw_lists.hip.h -- candidate lists on the packed inter-sequence kernel.
//
// sift4g aligns every query with ITS OWN candidates only (createFilteredDatabase,
// /root/reference/sift4g/src/database_alignment.cpp:136-162: <= 5000 ids per query, main.cpp:69), and the batched
// wrapper hands all lists of a chunk to the device at once: ~10^6 (query, target) pairs in which two queries rarely share
// a target.  The packed scan kernels (sw_kernels.hip.h) put two QUERIES on one target in the halves of a register, which
// needs targets both queries want; the one-wave-per-pair kernels (sw_pair.hip.h) take any pair but spend 11 vector
// lane-ops per cell (int32, 64 lanes on one DP) and run at the vector-issue limit doing so (profiles/r02_pair).
//
// Here the halves of a register carry two different TARGETS, each with a query of its own:
//   * a query's candidates are sorted by length on the device; an item is 64 consecutive candidates of query A in the lower
//     halves of the 64 lanes and 64 consecutive candidates of query B in the upper halves, A and B being two queries of
//     (nearly) the same length whose lists the host walks in step -- an item then spans 64 ranks of a length-sorted list
//     instead of 128, which halves the padding to its longest member (round 2: 128 candidates of ONE query, +11 % columns).
//     What is left of the longer list when the shorter one ends, and a query without a partner, run as items of 128 of one
//     query (A = B, the second 64 in the upper halves);
//   * list_gather copies the 128 targets of every item out of the resident database into a lane-transposed block of
//     their own (dword (c*2 + half)*64 + lane = columns 4c .. 4c+3), padded to the item's longest member -- the scan then
//     reads 2 x 256 contiguous bytes per wave and 4 columns, exactly like a database group;
//   * sw_scan_lists is sw_scan_pkc (round 5: the COLUMN frame X + B + j*e, v_pk_maximum3_f16 on integer bit patterns, same
//     exactness window bound + B + columns * e <= 0x7BFF; rounds 2-4 ran sw_scan_pkd's anti-diagonal frame)
//     with ONE difference: a lane looks its two residues up separately.  The strip profile is two tables of 32-bit words --
//     PA[4-row block][residue][4 rows] = s + e (sign-extended) for the lower half, PB = (s + e) << 16 for the upper --
//     and ONE v_add3_u32 per row adds both to the diagonal: on the register pair read as the integer hi * 65536 + lo that
//     IS the packed add (no half leaves 0 .. 0xFFFF, a negative lower score borrows from the upper half exactly as the
//     integer sum says).  6.5 vector ops per pair of cells, as the scan kernel (round 2 paired 16-bit scores with a
//     v_perm_b32 per row: 7.5); the price is twice the LDS reads (two ds_read_b128 per 4 rows) and 8 KB of tables per wave.
// Targets too long for a launch this short to wait for, and queries beyond the frame window, stay on the pair kernels
// (the host splits, s4g_lists.hip).
#pragma once
#include "sw_kernels.hip.h"

namespace s4g {

struct ListItem {
    uint32_t qa, qb;         // query of the lower / upper halves (qb == qa: one query, 128 of its candidates)
    uint32_t first_a, first_b;   // index of the halves' first entry in the sorted entry list (lane l: entry first + l)
    uint16_t na, nb;         // entries of each half, 0 .. 64 (na >= 1)
    uint32_t len4;           // 4-column chunks = ceil(longest member / 4)       (filled on the device: list_item_len)
    uint64_t res_off;        // dword offset of the item's block in the gathered residues (filled on the device: exclusive scan)
};
static_assert(sizeof(ListItem) == 32, "ListItem layout");

struct ListArgs {
    const ListItem* __restrict__ items;
    const uint32_t* __restrict__ res;           // gathered residues
    const uint64_t* __restrict__ entries;       // sorted entries: (target id << 32) | pair index
    const uint8_t* __restrict__ q_codes;
    const uint64_t* __restrict__ q_off;
    const uint32_t* __restrict__ q_len;
    const int8_t* __restrict__ mat;
    int32_t* __restrict__ scores;               // [pair index]
    int32_t* __restrict__ ws;
    uint64_t ws_slot;
    unsigned* counter;
    uint32_t n_items;
    uint32_t open, ext, floor_bits;
    uint32_t prio_cost;
    // SCAN mode (sw_scan_lists<R, true>: a database tile scanned by a few queries, s4g_lists.hip scan_tile_pipelined): an item is
    // (group first_a of the tile, queries qa / qb), `res` the tile's residues, res_off the group's offset; both halves of a lane
    // read the SAME target
    const uint32_t* __restrict__ grp_perm;      // [G * 64] output column of (group, lane); ~0u = empty lane
    // MODE 2 (round 6: a call with an ODD number of queries -- one query above all): the halves of an item may read DIFFERENT groups,
    // (qa on group first_a | qb on group first_b), so that the query without a partner fills both halves of its registers with two
    // neighbouring groups instead of computing every cell twice; first_b == first_a is MODE 1's pair of queries on one group.  The upper
    // half's residues come from the tile in place: dword offset and chunks of group first_b out of the tile's tables
    const uint64_t* __restrict__ grp_off;       // [G] dword offset of a group's residues in `res`
    const uint32_t* __restrict__ grp_len4;      // [G] 4-column chunks of a group
    int32_t* __restrict__ scan_scores;          // [query - q_begin][score_stride]
    uint64_t score_stride;
    uint32_t q_begin;
    // SCAN mode, speculative launch (round 6; 0 = every query of the launch is exact by its score bound): the host guarantees only that
    // the FRAME fits the int16 window; a result whose strip's in-frame running maximum passed spec_base = 0x7BFF - (max score + e) is
    // written as kScoreUnsure and recomputed by sw_pair_recheck -- sw_scan_pkc's scheme (sw_kernels.hip.h), strip by strip
    uint32_t spec_base;
};

// The entries of a call out of its raw pairs: key = (query << pos_bits | rank of the target in the database's length order), value = (target,
// position of the pair in the call).  A pair the list kernel does not take (target longer than the split length or than the
// query's drifting-frame window, or empty) gets the key (n_queries, 0): it sorts behind every list.
__global__ void list_entries(const uint32_t* __restrict__ pair_q, const uint32_t* __restrict__ pair_t, uint32_t n,
                             const uint32_t* __restrict__ seq_len, const uint32_t* __restrict__ pos_of_id, const uint32_t* __restrict__ lt_cap,
                             uint32_t split_len, uint32_t n_queries, uint32_t pos_bits, uint64_t* __restrict__ keys, uint64_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t qi = pair_q[i], t = pair_t[i];
    const uint32_t lt = seq_len[t];
    const bool mine = lt != 0 && lt <= split_len && lt <= lt_cap[qi];
    keys[i] = mine ? ((uint64_t)qi << pos_bits) | pos_of_id[t] : (uint64_t)n_queries << pos_bits;
    vals[i] = ((uint64_t)t << 32) | i;
}

// items[i].len4 and the item's size in dwords (input of the exclusive scan that yields res_off); stride != 0: no scan, item i's
// residues start at i * stride dwords (the buffer was sized by the longest item the call can have)
__global__ void list_item_len(ListItem* __restrict__ items, uint32_t n_items, const uint64_t* __restrict__ entries,
                              const uint32_t* __restrict__ seq_len, uint64_t* __restrict__ dwords, uint64_t stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const ListItem it = items[i];
    uint32_t l = seq_len[(uint32_t)(entries[it.first_a] >> 32)];            // entries are sorted longest first within a query
    if (it.nb) l = max(l, seq_len[(uint32_t)(entries[it.first_b] >> 32)]);
    const uint32_t len4 = (l + 3) / 4;
    items[i].len4 = len4;
    if (stride) items[i].res_off = (uint64_t)i * stride;
    else dwords[i] = (uint64_t)len4 * 128;
}

__global__ void list_item_off(ListItem* __restrict__ items, uint32_t n_items, const uint64_t* __restrict__ offs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_items) items[i].res_off = offs[i];
}

// One block per item: out[(c*2 + half)*64 + lane] = dword c of entry first_a + lane (half 0) / first_b + lane (half 1), pad (code 31) beyond a
// member's own length and in empty slots.  Sources are lane-transposed database sequences (dword c of a sequence lives
// 64*c dwords after its first): 128 scattered 4-byte reads per output row, served by L2 -- the pool of a chunk is a few
// tens of MB and every sector is wanted by the 16 neighbours of its group sooner or later.
__global__ __launch_bounds__(256) void list_gather(const ListItem* __restrict__ items, const uint64_t* __restrict__ entries,
                                                   const uint64_t* __restrict__ seq_addr, const uint32_t* __restrict__ seq_len,
                                                   uint32_t* __restrict__ out) {
    const ListItem it = items[blockIdx.x];
    const int slot = threadIdx.x & 127;               // half*64 + lane
    const int half = slot >> 6, lane = slot & 63;
    const uint32_t* src = nullptr;
    uint32_t l4 = 0, lt = 0;
    if ((uint32_t)lane < (half ? it.nb : it.na)) {
        const uint32_t t = (uint32_t)(entries[(half ? it.first_b : it.first_a) + lane] >> 32);
        src = reinterpret_cast<const uint32_t*>(seq_addr[t]);
        lt = seq_len[t];
        l4 = (lt + 3) / 4;
    }
    uint32_t* dst = out + it.res_off + slot;
    for (uint32_t c = threadIdx.x >> 7; c < it.len4; c += 2) {
        uint32_t v = 0x1F1F1F1Fu;
        if (c < l4) v = src[(uint64_t)c * 64];      // the database pads a sequence's last dword with code 31 already
        dst[(uint64_t)c * 128] = v;
    }
}

// rows row0 .. row0+3 of one column for both halves: xa = the 4 scores s + e of the lower half's residue, xb = those of the upper
// half's residue << 16.  Column frame (sw_scan_pkc, round 5): the running maximum lives in the frame, F carries the "- e".
template <int R>
__device__ __forceinline__ void lst_rows4(int (&H)[R], int (&E)[R], int row0, const v4i& xa, const v4i& xb, int& diag, int& F, int& best,
                                          int eo, int e2v, int zn) {
    int hprev = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + i;
        const int h = pk_maximum3((int)((unsigned)diag + (unsigned)xa[i] + (unsigned)xb[i]), E[r], F);       // v_add3_u32
        if (i & 1) best = pk_maximum3(best, hprev, h); else hprev = h;
        const int hm = (int)((unsigned)h + (unsigned)eo);               // eo: the pair (e - o, e - o) as an integer, in a VGPR
        E[r] = pk_maximum3(E[r], hm, zn);                               // zn = z_{j+1}: one SGPR per column
        F = (int)((unsigned)as_i(pk_max(as_s(F), as_s(hm))) - (unsigned)e2v);
        diag = H[r];
        H[r] = h;
    }
}

__device__ __forceinline__ v4i lst_load(const char* prof, unsigned taddr, int blk) {
    return *reinterpret_cast<const v4i*>(prof + blk * 512 + taddr);
}

// ---- the four waves of a block work on ONE item at a time, strip by strip, as a pipeline ----
// A launch of lists has about two items per resident wave and cannot end before its longest item, so an item walked by
// one wave (11 strips x 1000 columns, one dependent packed op every ~8 clocks) sets the time of the whole call.  Instead
// the block takes items from the queue one after the other and deals their strips out round-robin: strip g of the block's
// stream goes to wave g % 4.  Strip s reads the boundary row strip s-1 leaves in the workspace, so the wave that owns it
// trails its predecessor by a chunk of 4 columns: the predecessor publishes how many chunks it has finished in an LDS word
// (tag = its strip number, count), the follower waits on that word before it prefetches a boundary chunk.  Both waves
// sit on the same CU, whose L1 they share, so the boundary needs no more than workgroup-scope ordering: stores complete
// (s_waitcnt) before the count is published, and the loads are issued after the count was read.  Per-item latency drops
// to its strips / 4, the queue hands out 4x larger units to 4x fewer takers, and no wave ever idles at an item's end:
// the stream continues into the next item (whose first strip depends on nothing).
constexpr int kListRing = 16;            // fetched items a block keeps in LDS
constexpr unsigned kListPublishEvery = 1;     // chunks between two publications of a strip's progress (power of 2; 2 and 4 measured
                                              // the same: the release's wait for the boundary stores is hidden by the other waves)
constexpr int kMarked = 0x7FFFFFFF;      // an item's running maximum once a strip of a speculative launch could not vouch for it (above every score)
constexpr int kListBufs = 4;             // workspace buffers (and best-score rows) per block: item k uses buffer k % 4; a wave may
                                         // run at most 3 items ahead of the slowest wave of its block

template <int W>
struct ListSync {
    unsigned long long prog[W];                // (strip number + 1) << 32 | chunks finished, per wave
    unsigned cur[W];                           // item (position in the block's stream) a wave is working on
    unsigned fetched;                          // items in the ring
    unsigned lock;
    ListItem ring[kListRing];
    int best[kListBufs][128];                  // running maximum of an item: [lane * 2 + half]
};

__device__ __forceinline__ unsigned long long lds_load64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned lds_load32(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One strip of RR rows over all columns of the item; res points at the item's block + lane (low half; high half 64 dwords on).
// prod: progress word of the wave that owns the strip above (FIRST: unused), need_tag = that strip's tag; mine / my_tag: this wave's.
template <int R, int RR, bool FIRST, int MODE>
__device__ __forceinline__ void lst_strip(const uint32_t* res, const uint32_t* resb, unsigned len4b, unsigned len4, const char* prof, v4i* wsH, v4i* wsF, int lane,
                                          bool last, int& best, int eo, int e2v, int e2, int B2,
                                          const unsigned long long* prod, unsigned long long need_tag,
                                          unsigned long long* mine, unsigned long long my_tag) {
    constexpr bool first = FIRST;
    constexpr int NB = RR / 4;
    const char* const profb = prof + R * 128;       // PB behind PA
    int H[R];
    int E[R];
#pragma unroll
    for (int r = 0; r < RR; ++r) { H[r] = B2 - e2; E[r] = B2; }         // H~(i,-1) = z_{-1}, E~(i,0) = z_0: the same in every row
    int zj = B2;                         // z_j of the current column (SGPR)
    int hdiag = B2 - e2;
    unsigned avail = 0;                  // boundary chunks the strip above is known to have finished
    auto wait_for = [&](unsigned chunks) {
        if constexpr (!first) {
            while (avail < chunks) {
                const unsigned long long v = lds_load64(prod);
                const unsigned a = (v >> 32) > need_tag ? 0xFFFFFFFFu : ((v >> 32) == need_tag ? (unsigned)v : 0u);
                avail = (unsigned)__builtin_amdgcn_readfirstlane((int)a);
                if (avail < chunks) __builtin_amdgcn_s_sleep(1);
            }
        }
    };
    constexpr unsigned kStride = MODE ? 64u : 128u;          // dwords per 4-column chunk of the item's block
    // the upper half's residues: the gathered block's second row (lists), the lower half's own (a pair of queries on one group), or
    // another group of the tile in place, padded with code 31 beyond its own length (MODE 2; the item runs to the longer group)
    auto load_b = [&](unsigned c, uint32_t ra) -> uint32_t {
        if constexpr (MODE == 0) return res[(uint64_t)c * kStride + 64];
        else if constexpr (MODE == 1) return ra;
        else return c < len4b ? resb[(uint64_t)c * 64] : 0x1F1F1F1Fu;
    };
    uint32_t ra_n = res[0], rb_n = load_b(0, ra_n);
    v4i hb_n = {0, 0, 0, 0}, fb_n = {0, 0, 0, 0};
    if (!first && len4) { wait_for(1); hb_n = wsH[lane]; fb_n = wsF[lane]; }
    // the profile reads run TWO blocks of four rows ahead of the arithmetic: one block (22 instructions) does not cover an LDS round
    // trip when a wave has its SIMD to itself -- the case of a long item that a call of few queries waits for
    v4i xa_n = lst_load(prof, pk_taddr(ra_n, 0), 0), xb_n = lst_load(profb, pk_taddr(rb_n, 0), 0);
    v4i xa_m = lst_load(prof, pk_taddr(ra_n, 0), 1), xb_m = lst_load(profb, pk_taddr(rb_n, 0), 1);
    for (unsigned c = 0; c < len4; ++c) {
        const uint32_t ra = ra_n, rb = rb_n;
        const v4i hb = hb_n, fb = fb_n;
        const unsigned cn = (c + 1 < len4) ? c + 1 : c;
        ra_n = res[(uint64_t)cn * kStride];
        rb_n = load_b(cn, ra_n);
        if (!first) { wait_for(cn + 1); hb_n = wsH[(uint64_t)cn * 64 + lane]; fb_n = wsF[(uint64_t)cn * 64 + lane]; }
        v4i ho, fo;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const unsigned ta = pk_taddr(ra, cc), tb = pk_taddr(rb, cc);
            const unsigned ta_next = cc < 3 ? pk_taddr(ra, cc + 1) : pk_taddr(ra_n, 0);
            const unsigned tb_next = cc < 3 ? pk_taddr(rb, cc + 1) : pk_taddr(rb_n, 0);
            int diag = hdiag;
            int F = first ? zj : fb[cc];
            const int zn = zj + e2;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const v4i xa = xa_n, xb = xb_n;
                xa_n = xa_m; xb_n = xb_m;
                if (b + 2 < NB) { xa_m = lst_load(prof, ta, b + 2); xb_m = lst_load(profb, tb, b + 2); }
                else { xa_m = lst_load(prof, ta_next, b + 2 - NB); xb_m = lst_load(profb, tb_next, b + 2 - NB); }
                lst_rows4<R>(H, E, 4 * b, xa, xb, diag, F, best, eo, e2v, zn);
            }
            hdiag = first ? zj : hb[cc];
            zj = zn;
            ho[cc] = H[RR - 1];
            fo[cc] = F;
            // best~ moves to column j+1's frame.  A SIGNED SATURATING packed add (one single-rate instruction per column of RR rows), as in
            // sw_scan_pkc: as a 16-bit pattern best~ never decreases inside a strip and never passes 0x7FFF, which is what makes it the
            // witness of a speculative launch (ListArgs::spec_base); inside the window it is the plain add
            best = as_i(__builtin_elementwise_add_sat(as_s(best), as_s(e2)));
            asm volatile("" : "+v"(best));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!last) {
            wsH[(uint64_t)c * 64 + lane] = ho; wsF[(uint64_t)c * 64 + lane] = fo;
            if (c + 1 < len4 && (c & (kListPublishEvery - 1)) == kListPublishEvery - 1) {
                // (the last chunk is published by the caller, after the strip's maximum has been merged)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(mine, (my_tag << 32) | (c + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

// SCAN: a database tile scanned by a few queries (s4g_lists.hip, scan_tile_pipelined) -- an item is one group of the tile under a
// PAIR OF QUERIES: both halves read the same residues, in place (64 dwords per chunk), and the maxima go straight into the score
// matrix.  (Twelve waves to a block, one block to a CU -- every strip of a 350-row query on a wave of its own -- were measured too:
// the longest group of the bench tile, 11 strips x 3000 columns x 64 lanes, is 2.7 ms of ONE CU's issue slots however its strips
// are spread; that form reached 3.8 ms for it against 5.2 ms here, but only 8.9 TCUPS once the chip is full against 10.9.)
// (Round 5 tried sw_scan_pkc's memory side here too -- buffer instructions, boundary rows as per-column dwords reloaded in place, one
// block of profile scores ahead: 837 instead of 845 vector instructions per chunk, no scratch, 159 VGPRs -- and measured it neutral
// (729k pairs 10.2-10.8 against 10.5-10.9 ms, 8 queries x 100k 8.60 against 8.70 ms, 4 queries 5.70 against 5.56 ms); four waves
// per SIMD would need 128 VGPRs, 31 fewer than the two score vectors and two residue streams of this kernel leave.  Not kept.)
// W: waves per block.  Four everywhere but in scan-mode calls that are bound by the LATENCY of their longest item (few long queries
// against a large database, s4g_lists.hip scan_tile_pipelined): there twelve -- one block to a CU, three waves per SIMD, the register
// budget of the four-wave form -- walk an item's strips, so its latency drops towards what one CU can issue (round 6).
template <int R, int MODE = 0, int W = kWavesPerBlock>
__global__ __launch_bounds__(W * 64, 3) void sw_scan_lists(const ListArgs a) {
    constexpr bool SCAN = MODE != 0;
    static_assert(R % 8 == 0 && R <= kQueryAlign, "strip height");
    static_assert(1024 + W * R * 256 + 4096 <= 160 * 1024, "profile tables of the block's waves must fit gfx950's LDS");
    constexpr int kThreads = W * 64;
    __shared__ __attribute__((aligned(16))) char lds[1024 + W * R * 256];
    __shared__ ListSync<W> sy;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    signed char* lmat = reinterpret_cast<signed char*>(lds);
    char* prof = lds + 1024 + wave * (R * 256);        // PA then PB, each [R/4 blocks][32 residues][4 rows x int32]
    for (int i = threadIdx.x; i < 256; i += kThreads)
        reinterpret_cast<int*>(lds)[i] = reinterpret_cast<const int*>(a.mat)[i];
    for (int i = threadIdx.x; i < kListBufs * 128; i += kThreads) (&sy.best[0][0])[i] = 0;
    if (threadIdx.x < W) { sy.prog[threadIdx.x] = 0; sy.cur[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { sy.fetched = 0; sy.lock = 0; }
    __syncthreads();

    v4i* const ws_block = reinterpret_cast<v4i*>(a.ws + (uint64_t)blockIdx.x * kListBufs * a.ws_slot);
    const int e2 = (int)(a.ext | (a.ext << 16));
    int eo = ((int)a.ext - (int)a.open) * 65537;       // the pair (e - o, e - o) as the integer hi * 65536 + lo ...
    int e2v = e2;
    asm volatile("" : "+v"(eo), "+v"(e2v));            // ... in VGPRs: two-VGPR v_add_u32 / v_sub_u32 are the double-rate class (tools/valu_issue.hip)
    const int B2 = (int)a.floor_bits;
    const int one_e = (int)a.ext;
    const int prev_wave = (wave + W - 1) % W;

    // item k of this block's stream (fetched from the global queue by whichever wave needs it first); n == 0: end of the queue
    auto get_item = [&](unsigned k) -> ListItem {
        for (;;) {
            if (k < lds_load32(&sy.fetched)) break;
            unsigned got = 0;
            if (lane == 0) got = atomicCAS(&sy.lock, 0u, 1u) == 0u ? 1u : 0u;
            got = (unsigned)__builtin_amdgcn_readfirstlane((int)got);
            if (got) {
                const unsigned f = lds_load32(&sy.fetched);
                if (k >= f) {
                    if (lane == 0) {
                        const unsigned idx = atomicAdd(a.counter, 1u);
                        ListItem it;
                        if (idx < a.n_items) it = a.items[idx]; else { it.qa = it.qb = 0; it.first_a = it.first_b = 0; it.na = it.nb = 0; it.len4 = 0; it.res_off = 0; }
                        sy.ring[f % kListRing] = it;
                        __hip_atomic_store(&sy.fetched, f + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if (lane == 0) __hip_atomic_store(&sy.lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else __builtin_amdgcn_s_sleep(1);
        }
        ListItem it = sy.ring[k % kListRing];
        it.qa = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.qa);
        it.qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.qb);
        it.first_a = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.first_a);
        it.first_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.first_b);
        it.na = (uint16_t)__builtin_amdgcn_readfirstlane((int)it.na);
        it.nb = (uint16_t)__builtin_amdgcn_readfirstlane((int)it.nb);
        it.len4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.len4);
        it.res_off = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(it.res_off >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)it.res_off);
        return it;
    };

    unsigned k = 0, base = 0;                      // the item strip g falls into, and that item's first strip number
    // rows of an item: the longer of its two queries (the shorter one's rows beyond its end score kPadScore and raise nothing)
    auto item_rows = [&](const ListItem& x) -> unsigned {
        if (!x.na) return 0u;
        const unsigned la = a.q_len[x.qa], lb = a.q_len[x.qb];
        return la > lb ? la : lb;
    };
    ListItem it = get_item(0);
    unsigned lq = item_rows(it);
    unsigned nstrips = (lq + R - 1) / R;
    for (unsigned g = (unsigned)wave;; g += W) {
        while (it.na && g >= base + nstrips) {
            base += nstrips;
            ++k;
            it = get_item(k);
            lq = item_rows(it);
            nstrips = (lq + R - 1) / R;
        }
        if (!it.na) break;
        const unsigned s = g - base;
        // workspace buffer and best row k % 4 were last used by item k - 4: wait until no wave of the block is still there
        if (lane == 0) __hip_atomic_store(&sy.cur[wave], k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k >= kListBufs - 1) {
            for (int w = 0; w < W; ++w)
                while (lds_load32(&sy.cur[w]) + (kListBufs - 1) < k) __builtin_amdgcn_s_sleep(1);
        }
        const unsigned buf = k % kListBufs;
        const unsigned len4 = it.len4;
        v4i* wsH = ws_block + (uint64_t)buf * (a.ws_slot / 4);
        v4i* wsF = wsH + (uint64_t)len4 * 64;
        const uint32_t* res = a.res + it.res_off + lane;
        const uint32_t* resb = res;
        unsigned len4b = len4;
        if constexpr (MODE == 2) {
            if (it.first_b != it.first_a) {
                const uint64_t ob = a.grp_off[it.first_b];
                resb = a.res + (((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ob >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ob)) + lane;
                len4b = (unsigned)__builtin_amdgcn_readfirstlane((int)a.grp_len4[it.first_b]);
            }
        }
        const uint8_t* qca = a.q_codes + a.q_off[it.qa];
        const uint8_t* qcb = a.q_codes + a.q_off[it.qb];
        const unsigned cap_a = (a.q_len[it.qa] + kQueryAlign - 1) / kQueryAlign * kQueryAlign + kQueryAlign;      // readable bytes of the queries
        const unsigned cap_b = (a.q_len[it.qb] + kQueryAlign - 1) / kQueryAlign * kQueryAlign + kQueryAlign;
        if (a.prio_cost && (uint64_t)lq * len4 >= a.prio_cost) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
        const unsigned long long my_tag = (unsigned long long)g + 1;          // tags start at 1: the initial word 0 matches no strip
        if (lane == 0) __hip_atomic_store(&sy.prog[wave], my_tag << 32, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);

        const unsigned rows_left = lq - s * R;
        const unsigned rr = rows_left >= (unsigned)R ? (unsigned)R : (rows_left + 7) / 8 * 8;
        // ---- strip profile: PA[blk][t] = 4 rows of query A's (score + e), PB[blk][t] = query B's << 16 ----
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e0 = 0; e0 < 8 * R; e0 += 64) {
            const int e = e0 + lane;
            const int blk = e >> 5, t = e & 31;
            const unsigned row0 = s * R + 4 * blk;
            const unsigned ca = row0 < cap_a ? *reinterpret_cast<const unsigned*>(qca + row0) : 0x1F1F1F1Fu;
            const unsigned cb = row0 < cap_b ? *reinterpret_cast<const unsigned*>(qcb + row0) : 0x1F1F1F1Fu;
            v4i lo, hi;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                lo[d] = lmat[((ca >> (8 * d)) & 31u) * 32 + t] + one_e;
                hi[d] = (int)((unsigned)(lmat[((cb >> (8 * d)) & 31u) * 32 + t] + one_e) << 16);
            }
            *reinterpret_cast<v4i*>(prof + e * 16) = lo;
            *reinterpret_cast<v4i*>(prof + R * 128 + e * 16) = hi;
        }
        __builtin_amdgcn_wave_barrier();

        const bool first = (s == 0), last = (s + 1 == nstrips);
        const int zend = B2 + (int)(len4 * 4u) * e2;     // z of the column after the last: where the in-frame maximum stands at the strip's end
        int best = B2;                                   // in-frame running maximum of THIS strip (true 0)
        const unsigned long long* prod = &sy.prog[prev_wave];
        unsigned long long* mine = &sy.prog[wave];
        const unsigned long long need_tag = (unsigned long long)g;            // tag of strip g - 1
        bool done = false;
#define S4G_LAST_STRIP(RR_)                                                                                  \
        if constexpr (R > RR_) {                                                                             \
            if (!done && rr == RR_) {                                                                        \
                if (first) lst_strip<R, RR_, true, MODE>(res, resb, len4b, len4, prof, wsH, wsF, lane, last, best, eo, e2v, e2, B2, prod, need_tag, mine, my_tag); \
                else lst_strip<R, RR_, false, MODE>(res, resb, len4b, len4, prof, wsH, wsF, lane, last, best, eo, e2v, e2, B2, prod, need_tag, mine, my_tag);    \
                done = true;                                                                                 \
            }                                                                                                \
        }
        S4G_LAST_STRIP(8) S4G_LAST_STRIP(16) S4G_LAST_STRIP(24)
#undef S4G_LAST_STRIP
        if (!done) {
            if (first) lst_strip<R, R, true, MODE>(res, resb, len4b, len4, prof, wsH, wsF, lane, last, best, eo, e2v, e2, B2, prod, need_tag, mine, my_tag);
            else lst_strip<R, R, false, MODE>(res, resb, len4b, len4, prof, wsH, wsF, lane, last, best, eo, e2v, e2, B2, prod, need_tag, mine, my_tag);
        }
        const int bframe = best;                         // the strip's maximum still IN the frame: the witness of a speculative launch
        best = (int)((unsigned)best - (unsigned)zend);   // the true maximum of the strip (no borrow: best~ >= z in both halves)
        int lo = (int)as_s(best)[0], hi = (int)as_s(best)[1];
        if constexpr (SCAN) {
            if (a.spec_base) {
                // sw_scan_pkc's claim holds strip by strip: a strip whose in-frame maximum ends at or below spec_base held every value of
                // its own inside the window (the first value to leave it is a diag + s' above a value the maximum took in the column
                // before, or the maximum itself); the boundary row it read came from a strip that makes the same statement about itself.
                // A strip that cannot make it marks the item's result (kMarked wins every atomicMax below); a lower half beyond the
                // window may have carried into the upper one: it marks both.
                if (((unsigned)bframe & 0xFFFFu) > a.spec_base) lo = hi = kMarked;
                else if (((unsigned)bframe >> 16) > a.spec_base) hi = kMarked;
            }
        }
        // merge this strip's maximum into the item's, THEN publish the strip as finished: the wave that ends the item's last strip
        // has seen every strip above finished (each waited for its predecessor's last chunk), so the row is complete for it
        if (!(first && last)) {
            atomicMax(&sy.best[buf][2 * lane], lo);
            atomicMax(&sy.best[buf][2 * lane + 1], hi);
        }
        if (!last) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(mine, (my_tag << 32) | 0xFFFFFFFFull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            if (!first) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                lo = __hip_atomic_load(&sy.best[buf][2 * lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                hi = __hip_atomic_load(&sy.best[buf][2 * lane + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                sy.best[buf][2 * lane] = 0;
                sy.best[buf][2 * lane + 1] = 0;
            }
            if constexpr (SCAN) {
                if (lo == kMarked) lo = kScoreUnsure;
                if (hi == kMarked) hi = kScoreUnsure;
                const unsigned p = a.grp_perm[(uint64_t)it.first_a * 64 + lane];
                if constexpr (MODE == 2) {
                    // (qa on group first_a | qb on group first_b); an upper half that repeats the lower one (the last group of an odd count) is dropped
                    const unsigned pb = a.grp_perm[(uint64_t)it.first_b * 64 + lane];
                    if (p != 0xFFFFFFFFu) a.scan_scores[(uint64_t)(it.qa - a.q_begin) * a.score_stride + p] = lo;
                    if (pb != 0xFFFFFFFFu && !(it.qb == it.qa && it.first_b == it.first_a)) a.scan_scores[(uint64_t)(it.qb - a.q_begin) * a.score_stride + pb] = hi;
                } else if (p != 0xFFFFFFFFu) {
                    a.scan_scores[(uint64_t)(it.qa - a.q_begin) * a.score_stride + p] = lo;
                    if (it.qb != it.qa) a.scan_scores[(uint64_t)(it.qb - a.q_begin) * a.score_stride + p] = hi;
                }
            } else {
                if ((unsigned)lane < it.na) a.scores[(uint32_t)a.entries[it.first_a + lane]] = lo;
                if ((unsigned)lane < it.nb) a.scores[(uint32_t)a.entries[it.first_b + lane]] = hi;
            }
        }
    }
    if (lane == 0) __hip_atomic_store(&sy.cur[wave], 0xFFFFFFF0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace s4g

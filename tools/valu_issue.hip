// Microbenchmark: ISSUE cost, in shader clocks per wave64 instruction and SIMD, of the vector ops the scan kernels are made
// of -- measured with s_memtime inside the kernel over long unrolled bodies of independent chains, at the occupancies the
// kernels run at (1, 2, 3 and 8 waves per SIMD).  Complements tools/valu_rates.hip (wall-clock, dependent chains, priced at a
// nominal clock): this one answers "which forms issue in fewer than 4 clocks" (sw_scan_i32d was measured
// at 3.91 clocks per instruction, above the 16-lanes-per-clock model).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue.hip -o build/valu_issue && build/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <utility>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int CH = 16;      // independent chains (registers) per lane
constexpr int REP = 16;     // body = REP x CH instructions, unrolled
constexpr int LOOPS = 1024;   // 256 k instructions per wave: ~0.5 ms, against which the launch ramp is nothing

typedef float f2 __attribute__((ext_vector_type(2)));

enum Op { ADD_VV, ADD_SV, ADD_K, ADD_SDWA, MAX_VV, MAX3_VVV, MAX3_VVS, SUBREV_SV, SUB_CLAMP, MOV, PERM,
          PK_ADD, PK_SUB_S, PK_MAX, PK_MAX3, PK_MAX3_S, FMA_F32, ADD_F32, MAX_F32, MAX3_F32, PK_FMA_F32, PK_ADD_F32,
          MIX_I32D, MIX_PKD, LSHL_ADD, AND_OR, CNDMASK,
          SUB_VV, AND_VV, OR_VV, XOR_VV, LSHLREV_K, ASHRREV_K, MINU_VV, MAXU_VV, MUL_F32, SUB_F32, MIN_F32, ADD_U16, MAX_U16, ADDCO, MOV_DPP, ADD_DPP,
          ADD_LIT, BFE_I32, MUL_U24, MAD_U24, ADD3, FMAC_F32, ADD_F16, PK_ADD_F16, PK_MUL_F16, CVT_F32_I32, MIX_NEW, MIX_RUN8, MIX_RUN32, MIX_RUN128, MIX_SAMEOP, MIX_PKC, N_OPS };
static const char* kNames[N_OPS] = {
    "v_add_u32 v,v,v", "v_add_u32 v,s,v", "v_add_u32 v,64,v", "v_add_u32_sdwa sext(BYTE_1)", "v_max_i32 v,v,v", "v_max3_i32 v,v,v,v",
    "v_max3_i32 v,v,v,s", "v_subrev_u32 v,s,v", "v_sub_u32 clamp", "v_mov_b32", "v_perm_b32",
    "v_pk_add_u16", "v_pk_sub_i16 v,v,s", "v_pk_max_i16", "v_pk_maximum3_f16 v,v,v", "v_pk_maximum3_f16 v,v,s", "v_fma_f32", "v_add_f32",
    "v_max_f32", "v_max3_f32", "v_pk_fma_f32", "v_pk_add_f32",
    "mix of sw_scan_i32d (sdwa add, max3, subrev s, add, max3 s, max; 1 max3 per 2)", "mix of sw_scan_pkd (pk add, max3, pk sub s, add, max3 s, pk max; 1 max3 per 2)",
    "v_lshl_add_u32", "v_and_or_b32", "v_cndmask_b32 (vcc)",
    "v_sub_u32 v,v,v", "v_and_b32 v,v,v", "v_or_b32 v,v,v", "v_xor_b32 v,v,v", "v_lshlrev_b32 v,1,v", "v_ashrrev_i32 v,1,v", "v_min_u32 v,v,v", "v_max_u32 v,v,v",
    "v_mul_f32", "v_sub_f32", "v_min_f32", "v_add_u16", "v_max_u16", "v_add_co_u32 (vcc)", "v_mov_b32_dpp row_shr:1", "v_add_u32_dpp row_shr:1",
    "v_add_u32 v,0x12345,v (literal)", "v_bfe_i32 v,v,8,8", "v_mul_u32_u24", "v_mad_u32_u24", "v_add3_u32", "v_fmac_f32", "v_add_f16", "v_pk_add_f16", "v_pk_mul_f16",
    "v_cvt_f32_i32", "mix 4 fast (v_add/v_sub v,v,v) + 3.5 slow (v_pk_maximum3_f16): the 5.5-op packed recurrence",
    "the same mix in runs: 8 fast, then 7 slow", "runs: 32 fast, then 28 slow", "runs: 128 fast, then 112 slow", "isolated mix, every fast op the same v_add_u32",
    "mix of sw_scan_pkc (add v,v; max3; add v,v; max3 s; pk max; sub v,v; 1 max3 per 2 rows): 6 fast + 7 slow"};

// instruction mixes in the proportions of the recurrences, fully unrolled at compile time
template <int OP, int I>
__device__ __forceinline__ void mix_one(unsigned& x, unsigned vb, unsigned vc, unsigned sb) {
    if constexpr (OP == MIX_I32D) {      // one sw_scan_i32d row pair: 13 instructions
        constexpr int m = I % 13;
        if constexpr (m == 0 || m == 6) asm volatile("v_add_u32_sdwa %0, sext(%1), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(x) : "v"(vb));
        else if constexpr (m == 1 || m == 7 || m == 12) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
        else if constexpr (m == 2 || m == 8) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(x) : "s"(sb));
        else if constexpr (m == 3 || m == 9) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "s"(sb));
        else if constexpr (m == 4 || m == 10) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "s"(sb));
        else asm volatile("v_max_i32 %0, %1, %0" : "+v"(x) : "v"(vb));
    } else if constexpr (OP == MIX_PKD) { // one sw_scan_pkd row pair: 13 instructions, all 16 lanes per clock
        constexpr int m = I % 13;
        if constexpr (m == 0 || m == 6) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(vb));
        else if constexpr (m == 1 || m == 7 || m == 12) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
        else if constexpr (m == 2 || m == 8) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(x) : "s"(sb));
        else if constexpr (m == 3 || m == 9) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "s"(sb));
        else if constexpr (m == 4 || m == 10) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "s"(sb));
        else asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x) : "v"(vb));
    } else if constexpr (OP == MIX_RUN8 || OP == MIX_RUN32 || OP == MIX_RUN128) {
        constexpr int F = OP == MIX_RUN8 ? 8 : OP == MIX_RUN32 ? 32 : 128, S = F / 8 * 7;
        if constexpr (I % (F + S) < F) { if constexpr ((I & 1) == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "v"(vb)); else asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(vb)); }
        else asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
    } else if constexpr (OP == MIX_PKC) {  // one sw_scan_pkc row pair (round 5, column frame): 13 instructions, 6 of them two-VGPR 32-bit add / sub
        constexpr int m = I % 13;
        if constexpr (m == 0 || m == 6 || m == 2 || m == 8) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "v"(vb));
        else if constexpr (m == 1 || m == 7 || m == 12) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
        else if constexpr (m == 3 || m == 9) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "s"(sb));
        else if constexpr (m == 4 || m == 10) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x) : "v"(vb));
        else asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(vb));
    } else if constexpr (OP == MIX_SAMEOP) {
        constexpr int kind[15] = {0, 1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 1, 0, 1, 1};
        if constexpr (kind[I % 15] == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "v"(vb));
        else asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
    } else {                              // MIX_NEW: 15 instructions per row pair, 8 of them 32-bit add/sub on VGPRs
        constexpr int kind[15] = {0, 1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 1, 0, 1, 1};
        if constexpr (kind[I % 15] == 0) { if constexpr ((I & 1) == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "v"(vb)); else asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(vb)); }
        else asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
    }
}
template <int OP, int... I>
__device__ __forceinline__ void mix_seq(unsigned (&v)[CH], unsigned vb, unsigned vc, unsigned sb, std::integer_sequence<int, I...>) {
    (mix_one<OP, I>(v[I % CH], vb, vc, sb), ...);
}

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long* clk, unsigned* sink, unsigned a0, unsigned b0, unsigned s0, unsigned* arrive) {
    unsigned v[CH];
    f2 w[CH / 2];
#pragma unroll
    for (int c = 0; c < CH; ++c) v[c] = a0 + threadIdx.x * 7 + c;
#pragma unroll
    for (int c = 0; c < CH / 2; ++c) w[c] = f2{(float)(threadIdx.x + c), 1.0f};
    unsigned vb = b0 + threadIdx.x, vc = b0 * 3 + threadIdx.x;
    const unsigned sb = __builtin_amdgcn_readfirstlane(s0);
    asm volatile("" : "+v"(vb), "+v"(vc));
    // start together: the grid is sized to be co-resident (<= 8 blocks per CU), so every block can wait for the last one to
    // arrive (bounded spin: a grid that is not co-resident after all gives up and measures what it can)
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(arrive, 1u);
        for (unsigned spin = 0; spin < (1u << 22) && __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x; ++spin) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < LOOPS; ++it) {
        if constexpr (OP == MIX_I32D || OP == MIX_PKD || OP >= MIX_NEW) mix_seq<OP>(v, vb, vc, sb, std::make_integer_sequence<int, (OP == MIX_RUN128 ? 240 : 240)>{});
        else
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                unsigned& x = v[c];
                if constexpr (OP == ADD_VV) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == ADD_SV) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "s"(sb));
                if constexpr (OP == ADD_K) asm volatile("v_add_u32 %0, 64, %0" : "+v"(x));
                if constexpr (OP == ADD_SDWA) asm volatile("v_add_u32_sdwa %0, sext(%1), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(x) : "v"(vb));
                if constexpr (OP == MAX_VV) asm volatile("v_max_i32 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == MAX3_VVV) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
                if constexpr (OP == MAX3_VVS) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "s"(sb));
                if constexpr (OP == SUBREV_SV) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(x) : "s"(sb));
                if constexpr (OP == SUB_CLAMP) asm volatile("v_sub_u32 %0, %0, %1 clamp" : "+v"(x) : "v"(vb));
                if constexpr (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(vb));
                if constexpr (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
                if constexpr (OP == PK_ADD) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(vb));
                if constexpr (OP == PK_SUB_S) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(x) : "s"(sb));
                if constexpr (OP == PK_MAX) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x) : "v"(vb));
                if constexpr (OP == PK_MAX3) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
                if constexpr (OP == PK_MAX3_S) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "s"(sb));
                if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
                if constexpr (OP == ADD_F32) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == MAX_F32) asm volatile("v_max_f32 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == MAX3_F32) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
                if constexpr (OP == PK_FMA_F32) { if (c < CH / 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[c]) : "v"(w[(c + 1) % (CH / 2)])); else asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[c - CH / 2]) : "v"(w[(c + 1) % (CH / 2)])); }
                if constexpr (OP == PK_ADD_F32) { if (c < CH / 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[c]) : "v"(w[(c + 1) % (CH / 2)])); else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[c - CH / 2]) : "v"(w[(c + 1) % (CH / 2)])); }
                if constexpr (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(vb));
                if constexpr (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
                if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(vb));
                if constexpr (OP == SUB_VV) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(vb));
                if constexpr (OP == AND_VV) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == OR_VV) asm volatile("v_or_b32 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == XOR_VV) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == LSHLREV_K) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
                if constexpr (OP == ASHRREV_K) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(x));
                if constexpr (OP == MINU_VV) asm volatile("v_min_u32 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == MAXU_VV) asm volatile("v_max_u32 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == MUL_F32) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == SUB_F32) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == MIN_F32) asm volatile("v_min_f32 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == ADD_U16) asm volatile("v_add_u16 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == MAX_U16) asm volatile("v_max_u16 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == ADDCO) asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(x) : "v"(vb) : "vcc");
                if constexpr (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(vb));
                if constexpr (OP == ADD_DPP) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(vb));
                if constexpr (OP == ADD_LIT) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(x));
                if constexpr (OP == BFE_I32) asm volatile("v_bfe_i32 %0, %0, 8, 8" : "+v"(x));
                if constexpr (OP == MUL_U24) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
                if constexpr (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
                if constexpr (OP == FMAC_F32) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(vb), "v"(vc));
                if constexpr (OP == ADD_F16) asm volatile("v_add_f16 %0, %1, %0" : "+v"(x) : "v"(vb));
                if constexpr (OP == PK_ADD_F16) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x) : "v"(vb));
                if constexpr (OP == PK_MUL_F16) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(x) : "v"(vb));
                if constexpr (OP == CVT_F32_I32) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s ^= v[c];
#pragma unroll
    for (int c = 0; c < CH / 2; ++c) s ^= (unsigned)w[c][0] ^ (unsigned)w[c][1];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
int run(int cus, int waves_per_simd, double* out_cyc, double* out_tops) {
    const int blocks = cus * waves_per_simd;            // 4 waves per block, one per SIMD
    unsigned long long* clk; unsigned* sink; unsigned* arrive;
    CHK(hipMalloc(&clk, (size_t)blocks * 4 * 8));
    CHK(hipMalloc(&sink, (size_t)blocks * 256 * 4));
    CHK(hipMalloc(&arrive, 8));
    CHK(hipMemset(arrive, 0, 8));
    k<OP><<<blocks, 256>>>(clk, sink, 3, 5, 7, arrive);       // warm
    CHK(hipMemset(arrive, 0, 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    k<OP><<<blocks, 256>>>(clk, sink, 3, 5, 7, arrive);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    unsigned arrived = 0;
    CHK(hipMemcpy(&arrived, arrive, 4, hipMemcpyDeviceToHost));
    if (arrived != (unsigned)blocks) printf("(start barrier saw %u of %d blocks)\n", arrived, blocks);
    CHK(hipFree(arrive));
    std::vector<unsigned long long> h((size_t)blocks * 4);
    CHK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    // the SIMD arbitrates issue oldest wave first, so the waves of a SIMD do NOT finish together: the oldest runs as if alone
    // and the youngest ends when all the SIMD's work is done -- the slowest wave of the chip is the one that measures the rate
    const double slowest = (double)*std::max_element(h.begin(), h.end());
    const double n_instr = (double)LOOPS * ((OP == MIX_I32D || OP == MIX_PKD || OP >= MIX_NEW) ? 240 : REP * CH);
    *out_cyc = slowest / (n_instr * waves_per_simd);    // clocks per wave instruction and SIMD
    *out_tops = (double)blocks * 256.0 * n_instr / (ms * 1e-3) / 1e12;      // wall clock, launch included: T lane-ops/s
    CHK(hipFree(clk)); CHK(hipFree(sink));
    return 0;
}

template <int OP>
int row(int cus) {
    double c[5], t[5];
    const int ws[5] = {1, 2, 3, 4, 8};
    for (int i = 0; i < 5; ++i) if (run<OP>(cus, ws[i], &c[i], &t[i])) return 1;
    printf("%-86s %6.2f %6.2f %6.2f %6.2f %6.2f   %6.1f %6.1f %6.1f\n", kNames[OP], c[0], c[1], c[2], c[3], c[4], t[2], t[3], t[4]);
    return 0;
}

template <int OP>
int all(int cus, int first) {
    if constexpr (OP < N_OPS) { if (OP >= first && row<OP>(cus)) return 1; return all<OP + 1>(cus, first); }
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d CUs; shader clocks (s_memtime) per wave64 instruction and SIMD, from the slowest wave; %d chains x %d x %d instructions;\n"
           "last three columns: T lane-ops/s by HIP events (launch included) at 3, 4 and 8 waves per SIMD\n",
           p.gcnArchName, p.multiProcessorCount, CH, REP, LOOPS);
    printf("%-86s %6s %6s %6s %6s %6s   %6s %6s %6s   (waves per SIMD)\n", "instruction", "1", "2", "3", "4", "8", "3", "4", "8");
    return all<0>(p.multiProcessorCount, argc > 1 ? atoi(argv[1]) : 0);
}

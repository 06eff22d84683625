// Microbenchmark: sustained issue rate of the integer VALU ops the SW kernels are made of, on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o build/valu_rates && build/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;   // independent dependency chains per lane

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned a0, unsigned b0) {
    unsigned v[CHAINS];
    typedef short short2_t __attribute__((ext_vector_type(2)));
    typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) v[c] = a0 + threadIdx.x * 7 + c;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            unsigned x = v[c];
            if (OP == 0) x = x + b0;                                                        // v_add_u32
            if (OP == 1) x = (unsigned)max((int)x, (int)b0 + i);                            // v_max_i32
            if (OP == 2) x = (unsigned)max(max((int)x, (int)b0), (int)(a0 ^ i));            // v_max3_i32
            if (OP == 3) x = __builtin_elementwise_sub_sat(x, b0);                          // v_sub_u32 clamp
            if (OP == 4) { short2_t p = __builtin_bit_cast(short2_t, x), q = __builtin_bit_cast(short2_t, b0); p = p + q; x = __builtin_bit_cast(unsigned, p); }   // v_pk_add_i16
            if (OP == 5) { short2_t p = __builtin_bit_cast(short2_t, x), q = __builtin_bit_cast(short2_t, b0 + i); p = __builtin_elementwise_max(p, q); x = __builtin_bit_cast(unsigned, p); }  // v_pk_max_i16
            if (OP == 6) { ushort2_t p = __builtin_bit_cast(ushort2_t, x), q = __builtin_bit_cast(ushort2_t, b0); p = __builtin_elementwise_sub_sat(p, q); x = __builtin_bit_cast(unsigned, p); }  // v_pk_sub_u16 clamp
            if (OP == 7) x = x + (unsigned)(int)(signed char)(b0 >> 8);                    // SDWA add (sext byte)
            if (OP == 8) x = __builtin_amdgcn_perm(x, b0, 0x07020500u + i);                // v_perm_b32
            if (OP == 9) x = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, x) + __builtin_bit_cast(float, b0));  // v_add_f32
            if (OP == 10) x = __builtin_bit_cast(unsigned, __builtin_fmaf(__builtin_bit_cast(float, x), 1.0001f, __builtin_bit_cast(float, b0)));  // v_fma_f32
            if (OP == 11) x = __builtin_bit_cast(unsigned, __builtin_fmaxf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, b0 + i)));       // v_max_f32
            if (OP == 12) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); h2 p = __builtin_bit_cast(h2, x), q = __builtin_bit_cast(h2, b0); p = p + q; x = __builtin_bit_cast(unsigned, p); }  // v_pk_add_f16
            if (OP == 13) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); h2 p = __builtin_bit_cast(h2, x), q = __builtin_bit_cast(h2, b0 + i); p = __builtin_elementwise_max(p, q); x = __builtin_bit_cast(unsigned, p); }  // v_pk_max_f16
            if (OP == 14) { asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(b0 + i), "v"(a0)); }
            if (OP == 15) { asm volatile("v_maximum3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b0 + i), "v"(a0)); }
            if (OP == 16) { asm volatile("v_max3_i16 %0, %0, %1, %2" : "+v"(x) : "v"(b0 + i), "v"(a0)); }
            if (OP == 17) { asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x) : "v"(0x3c003c00u), "v"(b0)); }
            if (OP == 18) { asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(x) : "v"(0x00010001u), "v"(b0)); }
            if (OP == 19) { asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x) : "v"(1u), "v"(b0)); }
            if (OP == 20) { asm volatile("v_fma_f16 %0, %0, %1, %2" : "+v"(x) : "v"(0x3c00u), "v"(b0)); }
            if (OP == 21) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(1u), "v"(b0)); }
            if (OP == 22) { asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(0x00010001u)); }
            if (OP == 23) { asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x) : "v"(b0)); }
            if (OP == 24) { asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b0), "v"(a0)); }
            if (OP == 25) { asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x) : "v"(b0)); }
            if (OP == 26) { asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(x) : "v"(b0)); }
            if (OP == 27) { asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(b0), "v"(a0)); }
            asm volatile("" : "+v"(x));
            v[c] = x;
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s ^= v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
int run(const char* name, int blocks_per_cu) {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    int blocks = p.multiProcessorCount * blocks_per_cu;
    unsigned* out; CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k<OP><<<blocks, 256>>>(out, 3, 5);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) k<OP><<<blocks, 256>>>(out, 3, 5);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double ops = 5.0 * blocks * 256.0 * ITERS * CHAINS;
    double tops = ops / (ms * 1e-3) / 1e12;
    // per-SIMD cycles per wave-instruction at 2.4 GHz: waves/SIMD = blocks_per_cu (4 waves per block over 4 SIMDs)
    double inst_per_simd = 5.0 * blocks_per_cu * ITERS * CHAINS;
    double cyc = (ms * 1e-3) * 2.4e9 / inst_per_simd;
    printf("%-22s waves/SIMD=%d  %.2f T lane-ops/s  (%.2f cyc per wave-instr per SIMD @2.4GHz)\n", name, blocks_per_cu, tops, cyc);
    CHK(hipFree(out));
    return 0;
}

int main() {
    for (int w : {3, 8}) {
        run<0>("v_add_u32", w); run<1>("v_max_i32", w); run<2>("v_max3_i32", w); run<3>("v_sub_u32 clamp", w);
        run<4>("v_pk_add_i16", w); run<5>("v_pk_max_i16", w); run<6>("v_pk_sub_u16 clamp", w);
        run<7>("v_add_u32 sdwa", w); run<8>("v_perm_b32", w); run<9>("v_add_f32", w);
        run<10>("v_fma_f32", w); run<11>("v_max_f32", w); run<12>("v_pk_add_f16", w); run<13>("v_pk_max_f16", w);
        run<14>("v_pk_maximum3_f16", w); run<15>("v_maximum3_f32", w); run<16>("v_max3_i16", w);
        run<17>("v_pk_fma_f16", w); run<18>("v_pk_mad_i16", w); run<19>("v_mad_i32_i24", w); run<20>("v_fma_f16", w);
        run<21>("v_mad_u32_u24", w); run<22>("v_pk_mul_lo_u16", w); run<23>("v_pk_add_f16 asm", w); run<24>("v_add3_u32", w);
        run<25>("v_pk_max_i16 asm", w); run<26>("v_pk_min_i16 asm", w); run<27>("v_med3_i32", w);
        printf("\n");
    }
    return 0;
}

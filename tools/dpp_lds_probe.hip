// dpp_lds_probe.hip -- does a DPP wave_shr:1 return the right lanes while LDS reads are in flight?
// One wave per block.  Every iteration: issue ds_read_b128 + ds_read_u8 (results used only at the end of the iteration),
// run GAP independent VALU ops, then shift a known per-lane value with v_mov_b32_dpp wave_shr:1 and compare against what
// lane-1 must have held.  Mismatches are counted per GAP.  build: hipcc --offload-arch=gfx950 -O3 tools/dpp_lds_probe.hip -o build/dpp_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int GAP, int MODE>
__global__ __launch_bounds__(64) void probe(unsigned* out, int iters) {
    __shared__ __attribute__((aligned(16))) int buf[8192];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8192; i += 64) buf[i] = i * 2654435761u;
    __builtin_amdgcn_wave_barrier();
    unsigned bad = 0;
    int x = lane * 7 + 1, acc = 0;
    int f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3;
    for (int it = 0; it < iters; ++it) {
        const v4i a = *reinterpret_cast<const v4i*>(&buf[((lane * 4 + it * 4) & 8188)]);
        const unsigned char b = reinterpret_cast<const unsigned char*>(buf)[(lane + it) & 32767];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < GAP; ++g) {           // independent VALU filler
            if ((g & 3) == 0) f0 = f0 * 3 + g; else if ((g & 3) == 1) f1 = f1 ^ (f1 >> 3); else if ((g & 3) == 2) f2 = f2 + (f2 << 2); else f3 = f3 - g;
        }
        int sh;
        if (MODE == 0) sh = __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, false);          // wave_shr:1
        else {                                                                                   // row_shr:1 + row_bcast fix-ups
            int t = __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0x1, false);                   // row_bcast:15 -> lanes 16.., 48..
            t = __builtin_amdgcn_update_dpp(t, x, 0x143, 0x4, 0x1, false);                       // row_bcast:31 -> lanes 32..
            sh = __builtin_amdgcn_update_dpp(t, x, 0x111, 0xf, 0xf, false);                      // row_shr:1
        }
        __builtin_amdgcn_sched_barrier(0);
        const int expect = lane == 0 ? 0 : x - 7;
        bad += (sh != expect) ? 1u : 0u;
        x += 3 + (it & 1);
        acc += a[0] ^ a[1] ^ a[2] ^ a[3] ^ b;
    }
    out[blockIdx.x * 64 + lane] = bad;
    if (acc == 0x12345678 && f0 + f1 + f2 + f3 == 42) out[0] = 0xFFFFFFFFu;      // keep the loads and the filler alive
}

template <int GAP, int MODE>
static unsigned long long run(unsigned* d, int blocks, int iters) {
    hipMemset(d, 0, blocks * 64 * 4);
    hipLaunchKernelGGL((probe<GAP, MODE>), dim3(blocks), dim3(64), 0, 0, d, iters);
    std::vector<unsigned> h(blocks * 64);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long s = 0;
    for (unsigned v : h) s += v;
    return s;
}

template <int MODE>
static void sweep(unsigned* d, int blocks, int iters) {
    unsigned long long r[] = {run<0, MODE>(d, blocks, iters), run<2, MODE>(d, blocks, iters), run<4, MODE>(d, blocks, iters), run<6, MODE>(d, blocks, iters),
                              run<8, MODE>(d, blocks, iters), run<10, MODE>(d, blocks, iters), run<12, MODE>(d, blocks, iters), run<14, MODE>(d, blocks, iters),
                              run<16, MODE>(d, blocks, iters), run<20, MODE>(d, blocks, iters), run<24, MODE>(d, blocks, iters), run<28, MODE>(d, blocks, iters),
                              run<32, MODE>(d, blocks, iters), run<40, MODE>(d, blocks, iters), run<48, MODE>(d, blocks, iters), run<64, MODE>(d, blocks, iters)};
    const int gaps[] = {0, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 64};
    printf("%s: %d blocks x %d iterations per gap\n", MODE == 0 ? "wave_shr:1" : "row_shr:1 + row_bcast:15/31", blocks, iters);
    for (int i = 0; i < 16; ++i) printf("  gap %2d VALU ops: %llu mismatching lanes\n", gaps[i], r[i]);
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 20000;
    unsigned* d;
    hipMalloc(&d, blocks * 64 * 4);
    sweep<0>(d, blocks, iters);
    sweep<1>(d, blocks, iters);
    return 0;
}

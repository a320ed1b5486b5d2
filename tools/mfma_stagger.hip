// Microbenchmark (round 3): the STAGGER recipe of MI355X_MICROARCH.md "Two waves per SIMD", item 9, in isolation.
// A 512-thread workgroup = two waves per SIMD.  Every wave runs the same program: ITER x { NM independent v_mfma_f32_32x32x2_f32 ; NV
// vector-ALU instructions shaped like the LSTM cell update (v_exp / v_rcp / packed mul-add chains) }.  Waves 4-7 -- the SIMD partners
// of waves 0-3 -- start DELAY cycles late, so that in the steady state a wave's vector phase falls into its partner's matrix phase.
// No barrier inside the loop: the phase offset persists.  If the vector-ALU work of one wave can execute while its partner's MFMAs
// occupy the matrix pipe, the staggered run takes ITER x 2 x NM x 64 cycles (the vector phases vanish); if it cannot, both runs take
// ITER x (2 x NM x 64 + 2 x vector phase).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_stagger.hip -o tools/mfma_stagger && ./tools/mfma_stagger
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ unsigned long long g_cyc[8];   // per wave slot: elapsed cycles of the loop (max over workgroups)

template <int NM, int NV>
__global__ __launch_bounds__(512) void k(float *out, int iters, int delay, float a0) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + lane * 1e-9f, b = a0;
    f32x2 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = f32x2{a0 + i, a0 - i};
    __syncthreads();
    if (wave >= 4) {      // the partners start late
        const unsigned long long t = clock64();
        while (clock64() - t < (unsigned long long)delay) __builtin_amdgcn_s_sleep(1);
    }
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NV / 8; ++n) {      // 8 vector instructions per round: 2 exp, 2 rcp, 4 packed mul / add / fma -- the cell update's mix
            f32x2 &x = v[n & 3];
            const f32x2 y = x * 1.4426950408889634f;
            const f32x2 e = {__builtin_amdgcn_exp2f(-y.x), __builtin_amdgcn_exp2f(-y.y)};
            const f32x2 d = e + 1.0f;
            const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
            x = __builtin_elementwise_fma(r, x, y) * 0.5f;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
        s += v[i].x + v[i].y;
    }
    if (s == 12345.678f) out[0] = s;
    if (lane == 0) atomicMax(&g_cyc[wave], t1 - t0);
}

template <int NM, int NV>
static void run(const char *what, int delay) {
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), z, sizeof z);
    float *out;
    hipMalloc(&out, 4);
    const int iters = 200;
    hipLaunchKernelGGL((k<NM, NV>), dim3(256), dim3(512), 0, 0, out, iters, delay, 0.25f);
    hipDeviceSynchronize();
    unsigned long long c[8];
    hipMemcpyFromSymbol(c, HIP_SYMBOL(g_cyc), sizeof c);
    unsigned long long lo = 0, hi = 0;
    for (int w = 0; w < 4; ++w) { if (c[w] > lo) lo = c[w]; if (c[w + 4] > hi) hi = c[w + 4]; }
    printf("%-34s NM=%3d NV=%3d delay=%6d : waves 0-3 %8.0f cycles / iteration, waves 4-7 %8.0f   (2 x NM x 64 = %d)\n", what, NM, NV, delay,
           (double)lo / iters, (double)hi / iters, 2 * NM * 64);
    hipFree(out);
}

int main() {
    // reference points: matrix phase alone (NV = 0) and vector phase alone (NM = 0), two waves per SIMD
    run<64, 0>("matrix phase only", 0);
    run<0, 256>("vector phase only", 0);
    // LSTM-like proportion: vector phase about 10-25 % of the matrix phase
    run<64, 256>("lock-step", 0);
    run<64, 256>("staggered by half a matrix phase", 64 * 64 / 2);
    run<64, 256>("staggered by one matrix phase", 64 * 64);
    run<64, 256>("staggered by 1.5 matrix phases", 64 * 64 * 3 / 2);
    run<64, 512>("lock-step", 0);
    run<64, 512>("staggered by one matrix phase", 64 * 64);
    run<128, 256>("lock-step", 0);
    run<128, 256>("staggered by one matrix phase", 128 * 64);
    return 0;
}

// Microbenchmark: what does v_mfma_f32_32x32x2_f32 sustain on this card, issued by ONE wave per SIMD and by TWO?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak_acc.hip -o /tmp/mfma_peak_acc && /tmp/mfma_peak_acc
// Prints TFLOP/s and the fraction of the 157.3 TFLOP/s nominal peak (256 CUs x 4 SIMDs x 64 FLOP/cycle x 2.4 GHz) for
// 1 / 2 waves per SIMD and 2 / 4 / 8 independent accumulators per wave.  No memory traffic at all: an upper bound for any
// fp32-MFMA kernel (clock behaviour under sustained matrix load included).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool VG>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (VG)   // accumulators pinned to ARCHITECTURAL registers (where a kernel that post-processes them keeps them)
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                else      // compiler's choice: accumulation registers
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}

template <int NACC, bool VG = false>
void run(int wg_per_cu, float *d) {
    const int iters = 20000 / NACC * 2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<NACC, VG>), dim3(grid), dim3(256), 0, 0, d, iters, 1e-30f, 1e-30f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 4 * NACC * 4096.0;
    printf("%s waves/SIMD %d  accumulators %d: %7.2f ms  %6.1f TFLOP/s  %.3f of 157.3\n", VG ? "VGPR acc" : "AGPR acc", wg_per_cu, NACC, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}

int main() {
    float *d; hipMalloc(&d, 4);
    for (int w = 1; w <= 2; ++w) { run<2>(w, d); run<4>(w, d); run<8>(w, d); run<4, true>(w, d); run<8, true>(w, d); }
    // sustained: ~2 s of back-to-back launches, then measure again (clock behaviour under load)
    for (int i = 0; i < 60; ++i) hipLaunchKernelGGL((mfma_loop<8, false>), dim3(512), dim3(256), 0, 0, d, 40000, 1e-30f, 1e-30f);
    hipDeviceSynchronize();
    printf("after ~2 s of sustained load:\n");
    run<8>(2, d); run<8>(1, d);
    return 0;
}

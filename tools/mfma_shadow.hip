// Microbenchmark: do vector-ALU instructions of the SAME wave execute in the shadow of its own MFMAs?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_shadow.hip -o /tmp/mfma_shadow && /tmp/mfma_shadow
// One wave per SIMD; after every v_mfma_f32_32x32x2_f32 (64 cycles in the matrix pipe) the wave issues N independent
// instructions of a kind.  Cycles per (MFMA + N instructions) group: 64 = fully hidden, 64 + 4 N = serial.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned long long g_res[2];

template <int N, int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, const float *gsrc) {
    __shared__ float lds[1024];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a0; lds[threadIdx.x + 512] = a0; lds[threadIdx.x + 768] = a0;
    __syncthreads();
    float a = a0 + lane * 1e-9f, b = a0;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = a + q;
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < N; ++q) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q]) : "v"(b));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[q]));
                else if (KIND == 2) asm volatile("s_add_u32 %0, %0, 1" : "+s"(it));      // scalar ALU (harmless: adds to the counter... undone below)
                else if (KIND == 3) asm volatile("ds_read_b32 %0, %1" : "=v"(v[q]) : "v"(lane * 4 + q * 256));
                else if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<float2 *>(&v[2 * (q & 7)])) : "v"(*reinterpret_cast<float2 *>(&v[2 * (q & 7)])));
            }
            if (KIND == 2) it -= N;
            if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)");
        }
    const unsigned long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int q = 0; q < 16; ++q) s += v[q];
    if (lane == 0) { atomicAdd(&g_res[0], t1 - t0); atomicAdd(&g_res[1], 1ull); }
    if (s == 123.456f) out[0] = s;
}

template <int N, int KIND>
void run(float *d) {
    const int iters = 4000;
    unsigned long long z[2] = {0, 0}, r[2];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_res), z, sizeof z);
    hipLaunchKernelGGL((k<N, KIND>), dim3(256), dim3(256), 0, 0, d, iters, 1e-30f, d);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(g_res), sizeof r);
    const char *kinds[] = {"v_fma_f32", "v_exp_f32", "s_add_u32", "ds_read_b32", "v_pk_fma_f32"};
    printf("%2d x %-12s per MFMA: %6.1f cycles per group (64 = hidden, %d = serial at 4 cycles each)\n", N, kinds[KIND], (double)r[0] / r[1] / (iters * 8.0), 64 + 4 * N);
}

int main() {
    float *d; (void)hipMalloc(&d, 4096);
    run<0, 0>(d); run<2, 0>(d); run<4, 0>(d); run<8, 0>(d); run<12, 0>(d); run<16, 0>(d);
    run<2, 1>(d); run<4, 1>(d); run<8, 1>(d);
    run<4, 2>(d); run<12, 2>(d);
    run<2, 3>(d); run<4, 3>(d);
    run<4, 4>(d); run<8, 4>(d);
    return 0;
}

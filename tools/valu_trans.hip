// Microbenchmark: issue cost of transcendental (v_exp_f32 / v_rcp_f32) and ordinary vector instructions on one wave per
// SIMD, alone and interleaved, with 32 independent registers (no dependency stalls): do they overlap?
//   hipcc -O3 --offload-arch=gfx950 tools/valu_trans.hip -o /tmp/valu_trans && /tmp/valu_trans
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ unsigned long long g_res[2];

// KIND 0: 32 v_fma_f32; 1: 32 v_exp_f32; 2: 32 v_rcp_f32; 3: 8 x (v_exp, 3 x v_fma); 4: 8 x (v_exp, 3 x v_pk_fma_f32);
// 5: 16 x (v_exp, v_fma); 6: 32 v_pk_fma_f32
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0) {
    float v[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) v[q] = a0 + q + threadIdx.x * 1e-6f;
    const float c = a0;
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            bool trans;
            if (KIND == 0 || KIND == 6) trans = false;
            else if (KIND == 1 || KIND == 2) trans = true;
            else if (KIND == 5) trans = (q & 1) == 0;
            else trans = (q & 3) == 0;
            if (trans) {
                if (KIND == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[q]));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(v[q]));
            } else if (KIND == 4 || KIND == 6) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<float2 *>(&v[q & ~1])) : "v"(*reinterpret_cast<float2 *>(&v[(q & ~1) ^ 2])));
            } else {
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q]) : "v"(c));
            }
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += v[q];
    if ((threadIdx.x & 63) == 0) { atomicAdd(&g_res[0], t1 - t0); atomicAdd(&g_res[1], 1ull); }
    if (s == 123.456f) out[0] = s;
}

template <int KIND>
void run(float *d, const char *what) {
    const int iters = 20000;
    unsigned long long z[2] = {0, 0}, r[2];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_res), z, sizeof z);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256), 0, 0, d, iters, 1.0e-3f);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(g_res), sizeof r);
    printf("%-44s %6.2f cycles per instruction\n", what, (double)r[0] / r[1] / (iters * 32.0));
}

int main() {
    float *d; (void)hipMalloc(&d, 64);
    run<0>(d, "v_fma_f32"); run<6>(d, "v_pk_fma_f32"); run<1>(d, "v_exp_f32"); run<2>(d, "v_rcp_f32");
    run<3>(d, "1 v_exp_f32 : 3 v_fma_f32"); run<4>(d, "1 v_exp_f32 : 3 v_pk_fma_f32"); run<5>(d, "1 v_exp_f32 : 1 v_fma_f32");
    return 0;
}

// Microbenchmark 3: is the fp32 MFMA rate data dependent (power management)?  No memory traffic in the loop;
// operands rotate through 16+8 registers filled with zeros / constants / random bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__global__ __launch_bounds__(256, 2) void k(const float4 *__restrict__ src, float *out, int iters) {
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int t = blockIdx.x * 256 + threadIdx.x;
    float4 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = src[t * 8 + i];
    for (int i = 0; i < 2; ++i) b[i] = src[t * 8 + 4 + i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i][0] = MFMA(a[i].x, b[0].x, acc[i][0]); acc[i][1] = MFMA(a[i].x, b[1].x, acc[i][1]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i][0] = MFMA(a[i].y, b[0].y, acc[i][0]); acc[i][1] = MFMA(a[i].y, b[1].y, acc[i][1]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i][0] = MFMA(a[i].z, b[0].z, acc[i][0]); acc[i][1] = MFMA(a[i].z, b[1].z, acc[i][1]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i][0] = MFMA(a[i].w, b[0].w, acc[i][0]); acc[i][1] = MFMA(a[i].w, b[1].w, acc[i][1]);
        }
        // keep magnitudes bounded without changing the bit activity much
        if ((it & 255) == 255)
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] *= 1e-3f;
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[t] = s;
}
int main() {
    const int blocks = 512, n = blocks * 256 * 8 * 4;
    float4 *src; float *out;
    (void)hipMalloc(&src, n * 4); (void)hipMalloc(&out, blocks * 256 * 4);
    std::vector<float> h(n);
    const char *names[4] = {"zeros", "const 0.5", "random [-1,1)", "random bits (finite)"};
    for (int mode = 0; mode < 4; ++mode) {
        srand(1);
        for (int i = 0; i < n; ++i) {
            if (mode == 0) h[i] = 0.f;
            else if (mode == 1) h[i] = 0.5f;
            else if (mode == 2) h[i] = (rand() / (float)RAND_MAX) * 2.f - 1.f;
            else { union { unsigned u; float f; } v; v.u = ((unsigned)rand() << 16 ^ (unsigned)rand()) & 0xBF7FFFFF; v.u |= 0x30000000; v.u &= 0xBFFFFFFF; h[i] = v.f; }
        }
        (void)hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = rep ? 40000 : 4000;
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            k<<<blocks, 256>>>(src, out, 10); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); k<<<blocks, 256>>>(src, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%-22s iters=%6d  %8.2f ms  %.1f TFLOP/s\n", names[mode], iters, ms, (double)blocks * 4 * iters * 32 * 4096.0 / ms / 1e9);
        }
    }
    return 0;
}

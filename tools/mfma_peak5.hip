// Microbenchmark 5: fresh operands for every k-block from LDS (64 KiB of zeros / random data), no global traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ void mfma4(f32x16 &acc, const float4 &a, const float4 &b) {
    acc = MFMA(a.x, b.x, acc); acc = MFMA(a.y, b.y, acc); acc = MFMA(a.z, b.z, acc); acc = MFMA(a.w, b.w, acc);
}
__global__ __launch_bounds__(256, 2) void k(const float4 *__restrict__ src, float *out, int iters) {
    __shared__ float4 lds[4096];   // 64 KiB
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
    __syncthreads();
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int off = w * 64;
    for (int it = 0; it < iters; ++it) {
        float4 a[4], b[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = lds[(off + i * 256 + lane) & 4095];
        b[0] = lds[(off + 1024 + lane) & 4095]; b[1] = lds[(off + 1280 + lane) & 4095];
        off += 1536 + 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) { mfma4(acc[i][0], a[i], b[0]); mfma4(acc[i][1], a[i], b[1]); }
        if ((it & 63) == 63)
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] *= 1e-2f;
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int blocks = 512, n = 4096 * 4;
    float4 *src; float *out;
    (void)hipMalloc(&src, n * 4); (void)hipMalloc(&out, blocks * 256 * 4);
    std::vector<float> h(n);
    const char *names[3] = {"zeros", "const 0.5", "random [-1,1)"};
    for (int mode = 0; mode < 3; ++mode) {
        srand(1);
        for (int i = 0; i < n; ++i) h[i] = mode == 0 ? 0.f : mode == 1 ? 0.5f : (rand() / (float)RAND_MAX) * 2.f - 1.f;
        (void)hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<<<blocks, 256>>>(src, out, 10); (void)hipDeviceSynchronize();
        const int iters = 30000;
        (void)hipEventRecord(e0); k<<<blocks, 256>>>(src, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-16s %8.2f ms  %.1f TFLOP/s\n", names[mode], ms, (double)blocks * 4 * iters * 32 * 4096.0 / ms / 1e9);
    }
    return 0;
}

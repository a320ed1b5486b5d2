// Calibration microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 on gfx950 under different register/LDS/VMEM
// pressure.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// MODE 0: operands fixed in registers; NACC independent accumulators
// MODE 1: + one global_load_dwordx4 per 8 MFMAs feeding the operands (L2 resident buffer)
// MODE 2: + operands via LDS reads (ds_read_b128 per 4 MFMAs)
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k(const float4 *__restrict__ src, float *out, int iters) {
    __shared__ float4 lds[1024];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int lane = threadIdx.x;
    if (MODE == 2) { for (int i = lane; i < 1024; i += 256) lds[i] = src[i]; __syncthreads(); }
    float4 a = src[lane], b = src[lane + 256];
    float4 an = a;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) an = src[(it * 256 + lane) & 65535];
        if (MODE == 2) an = lds[(it * 64 + lane) & 1023];
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = MFMA(a.x, b.x, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = MFMA(a.y, b.y, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = MFMA(a.z, b.z, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = MFMA(a.w, b.w, acc[i]);
        }
        if (MODE) a = an;
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int MODE>
void run(const char *name, int blocks_per_cu, const float4 *src, float *out) {
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, MODE><<<blocks, 256>>>(src, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, MODE><<<blocks, 256>>>(src, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * NACC * 4 * 4096.0;
    printf("%-28s nacc=%d blocks/CU=%d  %.3f ms  %.1f TFLOP/s\n", name, NACC, blocks_per_cu, ms, flop / ms / 1e9);
}

int main() {
    float4 *src; float *out;
    std::vector<float> h(65536 * 4 + 4096, 0.001f);
    hipMalloc(&src, h.size() * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<4, 0>("regs only", 1, src, out);
    run<8, 0>("regs only", 1, src, out);
    run<8, 0>("regs only", 2, src, out);
    run<4, 0>("regs only", 2, src, out);
    run<8, 1>("global load per 32 mfma", 1, src, out);
    run<8, 1>("global load per 32 mfma", 2, src, out);
    run<8, 2>("lds read per 32 mfma", 1, src, out);
    run<8, 2>("lds read per 32 mfma", 2, src, out);
    return 0;
}

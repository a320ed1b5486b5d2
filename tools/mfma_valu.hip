// Microbenchmark: can ONE wave per SIMD keep the fp32 MFMA pipe busy while it also executes the LSTM cell math
// (v_exp / v_rcp / fma chains) in the MFMAs' shadow?  32 MFMAs per "k-block", CELLS lstm-cell updates interleaved.
// Premise test for an intra-wave ping-pong frequency-LSTM kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sg(float x) { return rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float th(float x) { const float e = __expf(-2.0f * fabsf(x)); return copysignf((1.0f - e) * rcp(1.0f + e), x); }

// MODE 0: real cell (5 v_exp + 5 v_rcp + ~25 VALU); 1: FMA-only stand-in with the same instruction count;
// 2: transcendentals only (10 per cell)
template <int MODE> __device__ __forceinline__ float cellf(float gi, float gf, float gg, float go, float &c) {
    if (MODE == 0) { const float cn = sg(gf) * c + sg(gi) * th(gg); c = cn; return sg(go) * th(cn); }
    if (MODE == 1) {
        float x = gi, y = gf, z = gg, w = go;
#pragma unroll
        for (int r = 0; r < 8; ++r) { x = x * y + z; y = y * z + w; z = z * w + x; w = w * x + y; }
        c = c * 0.5f + x; return y + z + w + c;
    }
    if (MODE == 3) {   // 32 FMAs in 8 independent chains
        float v[8] = {gi, gf, gg, go, c, gi + 1.f, gf + 1.f, gg + 1.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = v[q] * 0.999f + 0.001f;
        c = v[4]; return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[5] + v[6]) + v[7]);
    }
    float e0 = __expf(gi), e1 = __expf(gf), e2 = __expf(gg), e3 = __expf(go), e4 = __expf(c);
    c = rcp(e0) + rcp(e1); return rcp(e2) + rcp(e3) + rcp(e4);
}

template <int CELLS, int WPS, bool GROUP, int MODE = 0>
__global__ __launch_bounds__(256, WPS) void k(const float4 *__restrict__ src, float *out, int iters) {
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int t = blockIdx.x * 256 + threadIdx.x;
    float4 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = src[t * 8 + i];
    for (int i = 0; i < 2; ++i) b[i] = src[t * 8 + 4 + i];
    float gi[CELLS ? CELLS : 1], gf[CELLS ? CELLS : 1], gg[CELLS ? CELLS : 1], go[CELLS ? CELLS : 1], c[CELLS ? CELLS : 1], hsum = 0.f;
    for (int e = 0; e < CELLS; ++e) { gi[e] = a[e & 3].x + e; gf[e] = a[e & 3].y - e; gg[e] = b[e & 1].x * e; go[e] = b[e & 1].y; c[e] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        const float *av = reinterpret_cast<const float *>(a);
        const float *bv = reinterpret_cast<const float *>(b);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][0] = MFMA(av[i * 4 + q], bv[q], acc[i][0]);
                acc[i][1] = MFMA(av[i * 4 + q], bv[4 + q], acc[i][1]);
            }
        }
#pragma unroll
        for (int e = 0; e < CELLS; ++e) {
            const float hv = cellf<MODE>(gi[e], gf[e], gg[e], go[e], c[e]);
            hsum += hv;
            gi[e] += hv * 1e-3f; gf[e] -= hv * 1e-3f; gg[e] += hv * 2e-3f; go[e] -= hv * 1e-3f;   // keep the chain live
        }
        if (GROUP) {   // ask the scheduler for: 1 MFMA, then a slice of the VALU work, 32 times
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                         // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002 | 0x400, CELLS * 2 + 1, 0);     // VALU + transcendental slice
            }
        }
        if ((it & 255) == 255)
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] *= 1e-3f;
    }
    float s = hsum;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[t] = s;
}

template <int CELLS, int WPS, bool GROUP, int MODE = 0>
void run(const float4 *src, float *out, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<CELLS, WPS, GROUP, MODE><<<blocks, 256>>>(src, out, 10); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<CELLS, WPS, GROUP, MODE><<<blocks, 256>>>(src, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d waves/SIMD %d  cells per 32 MFMAs %2d  sched_group %d : %8.2f ms  %.1f TFLOP/s (MFMA only)\n", MODE, WPS, CELLS, (int)GROUP, ms,
           (double)blocks * 4 * iters * 32 * 4096.0 / ms / 1e9);
}
int main() {
    const int blocks1 = 256, n = 512 * 256 * 8 * 4;
    float4 *src; float *out;
    (void)hipMalloc(&src, n * 4); (void)hipMalloc(&out, 512 * 256 * 4);
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    (void)hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    // the frequency LSTM needs 64 cells per 24 k-blocks per wave = 2.7 cells per 32 MFMAs
    run<0, 1, false>(src, out, blocks1);
    run<2, 1, false>(src, out, blocks1);
    run<2, 1, true>(src, out, blocks1);
    run<3, 1, false>(src, out, blocks1);
    run<3, 1, true>(src, out, blocks1);
    run<6, 1, true>(src, out, blocks1);
    run<0, 2, false>(src, out, 512);
    run<3, 2, false>(src, out, 512);
    run<3, 2, false, 1>(src, out, 512);
    run<3, 2, false, 2>(src, out, 512);
    run<3, 2, false, 3>(src, out, 512);
    run<6, 2, false, 3>(src, out, 512);
    run<6, 1, false, 3>(src, out, 256);
    run<6, 2, false, 2>(src, out, 512);
    return 0;
}

// Microbenchmark: how fast can one CU pull an L2-resident weight set with the LSTM kernels' access pattern
// (per wave-instruction: two 512-byte runs, float4 per lane), with and without MFMAs running beside it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// MODE 0: loads only (sum into a register).  MODE 1: loads feed 32 MFMAs per 4 loads (LSTM ratio), prefetched one block ahead.
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float4 *__restrict__ W, float *out, int steps, int quads_per_step, int ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
    const float4 *wp = W + wave * 128 + l31 + h * ld;
    float4 s = make_float4(0, 0, 0, 0);
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int st = 0; st < steps; ++st) {
        float4 n0 = wp[0], n1 = wp[32], n2 = wp[64], n3 = wp[96];
        for (int kb = 0; kb < quads_per_step / 2; ++kb) {
            const float4 w0 = n0, w1 = n1, w2 = n2, w3 = n3;
            if (kb + 1 < quads_per_step / 2) {
                const float4 *q = wp + (size_t)(kb + 1) * 2 * ld;
                n0 = q[0]; n1 = q[32]; n2 = q[64]; n3 = q[96];
            }
            if (MODE == 0) {
                s.x += w0.x + w1.x + w2.x + w3.x; s.y += w0.y + w1.y + w2.y + w3.y;
            } else {
                const float4 b0 = make_float4(w0.y, w1.z, w2.x, w3.w), b1 = make_float4(w1.x, w0.z, w3.y, w2.w);
#define M4(A, a, b) A = MFMA(a.x, b.x, A); A = MFMA(a.y, b.y, A); A = MFMA(a.z, b.z, A); A = MFMA(a.w, b.w, A);
                M4(acc[0][0], w0, b0) M4(acc[0][1], w0, b1) M4(acc[1][0], w1, b0) M4(acc[1][1], w1, b1)
                M4(acc[2][0], w2, b0) M4(acc[2][1], w2, b1) M4(acc[3][0], w3, b0) M4(acc[3][1], w3, b1)
            }
        }
    }
    float r = s.x + s.y;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) r += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE>
void run(const char *name, const float4 *W, float *out, int blocks_per_cu) {
    const int ld = 512, quads = 48, steps = MODE ? 64 : 512, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(W, out, 2, quads, ld); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<MODE><<<blocks, 256>>>(W, out, steps, quads, ld); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 4 * steps * (quads / 2) * 4 * 1024.0;
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s blocks/CU=%d  %8.3f ms  %7.1f GB/s per CU  %5.1f B/clk/CU  %6.2f TB/s chip", name, blocks_per_cu, ms,
           bytes / 256 / ms / 1e6, bytes / 256 / cyc, bytes / ms / 1e9);
    if (MODE) printf("  | %.1f TFLOP/s", (double)blocks * 4 * steps * (quads / 2) * 32 * 4096.0 / ms / 1e9);
    printf("\n");
}
int main() {
    float4 *W; float *out;
    const size_t n = (size_t)48 * 512 + 4096;
    (void)hipMalloc(&W, n * 16); (void)hipMalloc(&out, 256 * 4 * 256 * 4);
    std::vector<float> h(n * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 9 & 0xffff) / 65536.f - 0.5f;
    (void)hipMemcpy(W, h.data(), n * 16, hipMemcpyHostToDevice);
    for (int b = 1; b <= 4; b *= 2) run<0>("loads only (384 KB set, L2)", W, out, b);
    for (int b = 1; b <= 2; ++b) run<1>("loads + 32 MFMA per 4 loads", W, out, b);
    return 0;
}

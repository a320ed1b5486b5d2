// PCA expansion of the dgrad head, both bases in one kernel (speech_anime/modules/output_module.py:94-116 PcaInversion;
// interleave [6 scale | 3 rotat] per triangle: speech_anime/model/model.py:246-257 data_to_anime_feat).
//
//   out[n][tri*9 + c]     = sum_k coefS[k][n] * BS[k][tri*6 + c] + meanS[tri*6 + c]        c = 0..5   (K = 85 -> 96)
//   out[n][tri*9 + 6 + c] = sum_k coefR[k][n] * BR[k][tri*3 + c] + meanR[tri*3 + c]        c = 0..2   (K = 180 -> 192)
//
// The generic GEMM runs the two bases as two launches whose epilogues each write 6 (or 3) of every 9 floats of a row:
// every 128-byte line of the 7.3 GB output is then written twice, partially, by different kernels -- a
// read-modify-write at the memory side, 7.9 ms per 20,352 frames for 2.7 ms of matrix work.  Here one workgroup owns
// 128 frames x 32 triangles and computes BOTH parts (per wave: 32 frames; 6 scale tiles + 3 rotat tiles of 32x32), so
// all stores to a line leave the same wave within a microsecond and merge in that XCD's L2 into full-line writes.
// Operands come straight from global memory as K4 quads (register-direct, no LDS): the coefficient quads of the wave's
// 32 frames and the basis quads of its columns, requested one k-block ahead in two alternating register sets.  The four
// waves of a workgroup read the same basis slab (L1 hits); frame blocks of one triangle block are dispatched together,
// so the slab (147 KB) stays in L2.
#include "common.h"
#include "kernels.h"

namespace {

template <int NT>   // NT column tiles of 32: 6 for the scale basis (group 6), 3 for the rotat basis (group 3)
__device__ __forceinline__ void pca_part(const float4 *__restrict__ coef, int64_t ldc, const float4 *__restrict__ basis, int64_t ldb,
                                         int nkb, int h, f32x16 (&acc)[1][NT]) {
    // lane: A row = frame (coef already offset to this lane's frame), B column = basis already offset to tile 0's column
    float4 a0[1], b0[NT], a1[1], b1[NT];
#define PCA_LOAD(kb, A, B)                                                    \
    {                                                                         \
        const int64_t kq = 2 * (kb) + h;                                      \
        A[0] = coef[kq * ldc];                                                \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) B[t] = basis[kq * ldb + 32 * t]; \
    }
    PCA_LOAD(0, a0, b0)
#pragma unroll 1
    for (int kb = 0; kb < nkb; kb += 2) {      // nkb is even (12 or 24)
        PCA_LOAD(kb + 1, a1, b1)
        __builtin_amdgcn_sched_barrier(0);
        mfma_block<1, NT>(acc, a0, b0);
        const int kn = kb + 2 < nkb ? kb + 2 : 0;   // branch-free; the last request is dropped
        PCA_LOAD(kn, a0, b0)
        __builtin_amdgcn_sched_barrier(0);
        mfma_block<1, NT>(acc, a1, b1);
    }
#undef PCA_LOAD
}

template <int NT, int GROUP, int OFF>
__device__ __forceinline__ void pca_store(const PcaArgs &a, const f32x16 (&acc)[1][NT], const float *__restrict__ mean, int64_t col0,
                                          int64_t cols, int64_t frame0, int h, int l31) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int64_t q = col0 + 32 * t + l31;
        if (q >= cols) continue;
        const float m = mean[q];
        const int64_t o = (q / GROUP) * 9 + OFF + q % GROUP;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t n = frame0 + 8 * g + 4 * h + e;
#ifdef SDFA_PCA_NOSTORE   /* timing experiment only: matrix work without the output stream */
                if (n < a.N && acc[0][t][4 * g + e] == 12345.678f) a.out[n * a.out_dim + o] = m;
#else
                if (n < a.N) a.out[n * a.out_dim + o] = acc[0][t][4 * g + e] + m;
#endif
            }
    }
}

__global__ __launch_bounds__(256, 2) void pca_dgrad_kernel(PcaArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int64_t nfb = a.Nc / 128;                       // frame blocks: fastest-varying, so a basis slab is reused from L2
    const int64_t fb = blockIdx.x % nfb, tb = blockIdx.x / nfb;
    const int64_t frame0 = fb * 128 + wave * 32;
    if (frame0 >= a.N) return;

    const float4 *__restrict__ coef = reinterpret_cast<const float4 *>(a.coef) + frame0 + l31;   // K4 [288/4][Nc]: scale rows 0..95, rotat 96..287
    {
        f32x16 acc[1][6];
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][t][r] = 0.f;
        const int64_t col0 = tb * 192;
        pca_part<6>(coef, a.Nc, reinterpret_cast<const float4 *>(a.basis_s) + col0 + l31, a.ld_s, 12, h, acc);
        pca_store<6, 6, 0>(a, acc, a.mean_s, col0, a.cols_s, frame0, h, l31);
    }
    {
        f32x16 acc[1][3];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][t][r] = 0.f;
        const int64_t col0 = tb * 96;
        pca_part<3>(coef + 24 * a.Nc, a.Nc, reinterpret_cast<const float4 *>(a.basis_r) + col0 + l31, a.ld_r, 24, h, acc);
        pca_store<3, 3, 6>(a, acc, a.mean_r, col0, a.cols_r, frame0, h, l31);
    }
}

}  // namespace

hipError_t sdfa_launch_pca_dgrad(const PcaArgs &a, hipStream_t s) {
    // 32 triangles per workgroup: 192 scale columns + 96 rotat columns; the padded leading dimensions must cover whole blocks
    const int64_t ntb = (a.cols_r + 95) / 96;
    if (a.Nc % 128 || a.ld_s < ntb * 192 || a.ld_r < ntb * 96 || a.cols_s != 2 * a.cols_r) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pca_dgrad_kernel, dim3((unsigned)(ntb * (a.Nc / 128))), dim3(256), 0, s, a);
    return hipGetLastError();
}

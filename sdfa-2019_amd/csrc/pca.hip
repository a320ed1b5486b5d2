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
// a wave owns complete output rows: it transposes them through LDS and writes 1,152 contiguous bytes per frame with
// 16-byte stores.
// Operands come straight from global memory as K4 quads (register-direct, no LDS): the coefficient quads of the wave's
// 32 frames and the basis quads of its columns, requested two k-blocks ahead in three rotating register sets.  The four
// waves of a workgroup read the same basis slab (L1 hits); frame blocks of one triangle block are dispatched together,
// so the slab (147 KB) stays in L2.
#include "common.h"
#include "kernels.h"

#define WAVE_LDS_FENCE() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }

namespace {

template <int NT>   // NT column tiles of 32: 6 for the scale basis (group 6), 3 for the rotat basis (group 3)
__device__ __forceinline__ void pca_part(const float4 *__restrict__ coef, int64_t ldc, const float4 *__restrict__ basis, int64_t ldb,
                                         int nkb, int h, f32x16 (&acc)[1][NT]) {
    // lane: A row = frame (coef already offset to this lane's frame), B column = basis already offset to tile 0's column
    // requests run TWO k-blocks ahead in three rotating register sets: the rotat part has only 12 MFMAs (768 cycles) per
    // k-block, less than an L2 round trip
    float4 a0[1], b0[NT], a1[1], b1[NT], a2[1], b2[NT];
#define PCA_LOAD(kb, A, B)                                                    \
    {                                                                         \
        const int64_t kq = 2 * (kb) + h;                                      \
        A[0] = coef[kq * ldc];                                                \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) B[t] = basis[kq * ldb + 32 * t]; \
    }
    PCA_LOAD(0, a0, b0)
    PCA_LOAD(1, a1, b1)
#pragma unroll 1
    for (int kb = 0; kb < nkb; kb += 3) {      // nkb is a multiple of 3 (12 or 24); late requests wrap to k-block 0 and are dropped
        PCA_LOAD(kb + 2 < nkb ? kb + 2 : 0, a2, b2)
        __builtin_amdgcn_sched_barrier(0);
        mfma_block<1, NT>(acc, a0, b0);
        PCA_LOAD(kb + 3 < nkb ? kb + 3 : 0, a0, b0)
        __builtin_amdgcn_sched_barrier(0);
        mfma_block<1, NT>(acc, a1, b1);
        PCA_LOAD(kb + 4 < nkb ? kb + 4 : 0, a1, b1)
        __builtin_amdgcn_sched_barrier(0);
        mfma_block<1, NT>(acc, a2, b2);
    }
#undef PCA_LOAD
}

constexpr int PCA_ROW = 292;   // floats per staged row: 288 used; 4 * 292 mod 32 = 16 keeps the two lane halves on different LDS banks

// accumulator rows 8g + 4h + e of pass g -> LDS rows 4h + e, triangle-interleaved columns; + mean
template <int NT, int GROUP, int OFF>
__device__ __forceinline__ void pca_stage(float *__restrict__ sRow, const f32x16 (&acc)[1][NT], const float *__restrict__ mean,
                                          int64_t col0, int64_t cols, int g, int h, int l31) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ql = 32 * t + l31;                       // column inside this workgroup's slice of the basis
        const int64_t q = col0 + ql;
        const float m = q < cols ? mean[q] : 0.f;
        const int o = (ql / GROUP) * 9 + OFF + ql % GROUP; // position inside the 288-float row segment
#pragma unroll
        for (int e = 0; e < 4; ++e) sRow[(4 * h + e) * PCA_ROW + o] = acc[0][t][4 * g + e] + m;
    }
}

// The same contraction with the BASIS staged through LDS (round 2).  The register-direct form above has every wave fetch
// the whole basis slab of its workgroup itself -- the four waves multiply different frames by the SAME columns -- which is
// 0.3 vector-memory requests per MFMA and keeps the CU's L1 / texture-address path 60 % busy: the kernel is bound there,
// not by the matrix pipe (MfmaUtil 61 %).  Here the 256 threads load each stage of the slab ONCE (3 requests per thread
// and stage of 48 MFMAs per wave), park it in registers for one stage, store it to the other LDS buffer and meet at one
// barrier per stage; the waves read their B operands from LDS and only the coefficient quads (one per k-block) from L2.
// Same k order per accumulator: bit-identical results.
template <int NT, int KQS>   // NT column tiles of 32; KQS k-quads per stage (KQS * NT * 32 = 768 float4 = 3 per thread)
__device__ __forceinline__ void pca_part_lds(const float4 *__restrict__ coef, int64_t ldc, const float4 *__restrict__ basis, int64_t ldb,
                                             int nkq, float4 *__restrict__ sB, int tid, int h, int l31, f32x16 (&acc)[1][NT]) {
    constexpr int COLS = NT * 32, ITEMS = KQS * COLS / 256, KB = KQS / 2;
    static_assert(KQS * COLS == 768 && ITEMS == 3, "a stage is 768 float4");
    const int nst = nkq / KQS;
    float4 rg[ITEMS], a_cur[KB], a_nxt[KB];
#define PL_GLOAD(st)                                                                              \
    _Pragma("unroll") for (int m = 0; m < ITEMS; ++m) {                                           \
        const int i = tid + 256 * m;                                                              \
        rg[m] = basis[(int64_t)((st)*KQS + i / COLS) * ldb + i % COLS];                           \
    }
#define PL_LSTORE(buf) _Pragma("unroll") for (int m = 0; m < ITEMS; ++m) sB[(buf)*768 + tid + 256 * m] = rg[m];
#define PL_ALOAD(st, A) _Pragma("unroll") for (int kb = 0; kb < KB; ++kb) A[kb] = coef[(int64_t)((st)*KQS + 2 * kb + h) * ldc];
    PL_GLOAD(0)
    PL_ALOAD(0, a_cur)
    PL_LSTORE(0)
    if (nst > 1) { PL_GLOAD(1) }
    __syncthreads();
#pragma unroll 1
    for (int st = 0; st < nst; ++st) {
        const float4 *__restrict__ sb = sB + (st & 1) * 768;
        if (st + 1 < nst) { PL_ALOAD(st + 1, a_nxt) }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            float4 b[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) b[t] = sb[(2 * kb + h) * COLS + 32 * t + l31];
            const float4 a1[1] = {a_cur[kb]};
            mfma_block<1, NT>(acc, a1, b);
        }
        if (st + 1 < nst) { PL_LSTORE((st & 1) ^ 1) }      // stage st+1 has been in registers for a whole stage
        __syncthreads();
        if (st + 2 < nst) { PL_GLOAD(st + 2) }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) a_cur[kb] = a_nxt[kb];
    }
#undef PL_GLOAD
#undef PL_LSTORE
#undef PL_ALOAD
}

template <bool LDS_BASIS>
__global__ __launch_bounds__(256, 2) void pca_dgrad_kernel(PcaArgs a) {
    __shared__ float sOut[4][8 * PCA_ROW];
    __shared__ float4 sBasis[LDS_BASIS ? 2 * 768 : 1];    // two stages of the basis slab (24 KiB)                // per wave: 8 output rows x 288 floats, staged for 16-byte stores
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int64_t nfb = a.Nc / 128;                       // frame blocks: fastest-varying, so a basis slab is reused from L2
    const int64_t fb = blockIdx.x % nfb, tb = blockIdx.x / nfb;
    const int64_t frame0 = fb * 128 + wave * 32;
    if (!LDS_BASIS && frame0 >= a.N) return;      // (with LDS staging every wave helps to load and meets the barriers)

    const float4 *__restrict__ coef = reinterpret_cast<const float4 *>(a.coef) + frame0 + l31;   // K4 [288/4][Nc]: scale rows 0..95, rotat 96..287
    f32x16 accs[1][6], accr[1][3];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[0][t][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) accr[0][t][r] = 0.f;
    if (LDS_BASIS) {
        pca_part_lds<6, 4>(coef, a.Nc, reinterpret_cast<const float4 *>(a.basis_s) + tb * 192, a.ld_s, 24, sBasis, tid, h, l31, accs);
        pca_part_lds<3, 8>(coef + 24 * a.Nc, a.Nc, reinterpret_cast<const float4 *>(a.basis_r) + tb * 96, a.ld_r, 48, sBasis, tid, h, l31, accr);
        if (frame0 >= a.N) return;
    } else {
        pca_part<6>(coef, a.Nc, reinterpret_cast<const float4 *>(a.basis_s) + tb * 192 + l31, a.ld_s, 12, h, accs);
        pca_part<3>(coef + 24 * a.Nc, a.Nc, reinterpret_cast<const float4 *>(a.basis_r) + tb * 96 + l31, a.ld_r, 24, h, accr);
    }

    // epilogue: four passes of 8 frames; the wave transposes its (8 x 288) block through LDS and writes whole rows with
    // 16-byte stores (1,152 contiguous bytes per frame) instead of 4-byte stores strided 6-of-9 / 3-of-9
    float *sRow = sOut[wave];
    const int64_t ocol0 = tb * 288, orow_valid = a.out_dim - ocol0;     // floats of this triangle block that exist (216 in the last one)
#pragma unroll
    for (int g = 0; g < 4; ++g) {            // unrolled: g indexes accumulator registers
        pca_stage<6, 6, 0>(sRow, accs, a.mean_s, tb * 192, a.cols_s, g, h, l31);
        pca_stage<3, 3, 6>(sRow, accr, a.mean_r, tb * 96, a.cols_r, g, h, l31);
        WAVE_LDS_FENCE()
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int idx = i * 64 + lane, r = idx / 72, c4 = idx % 72;     // row-local mapping r = 4h' + e  <->  frame 8g + 4h' + e
            const int64_t n = frame0 + 8 * g + r;
            if (n < a.N && 4 * c4 < orow_valid) {
                const float4 v = *reinterpret_cast<const float4 *>(sRow + r * PCA_ROW + 4 * c4);
                const int64_t o = n * a.out_dim + ocol0 + 4 * c4;
                *reinterpret_cast<float4 *>(a.out + o) = v;
                for (int x = 0; x < a.n_extra; ++x) *reinterpret_cast<float4 *>(a.out_extra[x] + o) = v;   // peers' gathered buffers
            }
        }
        WAVE_LDS_FENCE()
    }
}

}  // namespace

extern thread_local int g_sdfa_pca_lds;   // api.cpp ("pca_lds" option): 1 = basis staged through LDS

hipError_t sdfa_launch_pca_dgrad(const PcaArgs &a, hipStream_t s) {
    // 32 triangles per workgroup: 192 scale columns + 96 rotat columns; the padded leading dimensions must cover whole blocks
    const int64_t ntb = (a.cols_r + 95) / 96;
    if (a.Nc % 128 || a.ld_s < ntb * 192 || a.ld_r < ntb * 96 || a.cols_s != 2 * a.cols_r) return hipErrorInvalidValue;
    if (g_sdfa_pca_lds) hipLaunchKernelGGL(pca_dgrad_kernel<true>, dim3((unsigned)(ntb * (a.Nc / 128))), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(pca_dgrad_kernel<false>, dim3((unsigned)(ntb * (a.Nc / 128))), dim3(256), 0, s, a);
    return hipGetLastError();
}

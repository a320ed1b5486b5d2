// Generic fp32 MFMA GEMM on K4 operands:  D[p][q] = sum_k P[k][p] * Q[k][q]  (+ epilogue)
//
//   P : float4[K/4][ldp]   -- normally packed weights W^T (p = output feature)
//   Q : float4[K/4][ldq]   -- normally activations      (q = column)
//   D : OUT_K4  -> float4[Pstore/4][ldd]  (again a K4 activation, features = p)
//       OUT_ROW -> float  [p][ldd]        (row-major, used for the final PCA expansion where
//                                          p = animation frame, q = output coordinate)
// 128x128 tile per 256-thread workgroup, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles,
// K step 32 (8 k-quads), register-staged double-buffered LDS.  fp32 MFMA issues one
// 32x32x2 per 64 cycles per SIMD, so operand traffic is light: per k-step a wave reads
// 4 x ds_read_b128 per 16 MFMAs.
//
// The contraction can run over `nseg` K-segments of `seg_k` each whose Q columns start
// `seg_col` apart (the attention query Conv1d(k=3, stride 3) over time steps 31..33:
// speech_anime/layers/attentions.py:49-54).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TP = 128, TQ = 128, KQ = 8;   // tile p, tile q, k-quads per stage

template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND, bool DMA = false>
__global__ __launch_bounds__(256, 2) void gemm_k4_kernel(GemmArgs a) {
    __shared__ float4 sP[2][KQ][TP];
    __shared__ float4 sQ[2][KQ][TQ];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wq = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // XCD-aware tile order: consecutive q-tiles of one p-tile share the weight slab in one L2
    const int64_t ntq = a.Qpad / TQ;
    const int64_t bid = blockIdx.x;
    const int64_t tp = bid / ntq, tq = bid % ntq;
    const int64_t p0 = tp * TP, q0 = tq * TQ;
    if (a.q_limit && q0 >= *a.q_limit) return;

    const float4 *__restrict__ P = reinterpret_cast<const float4 *>(a.P);
    const float4 *__restrict__ Q = reinterpret_cast<const float4 *>(a.Q);
    const int nkq_total = a.K / 4;
    const int seg_kq = a.seg_k / 4;
    const int nstage = nkq_total / KQ;

    float4 rp0, rp1, rp2, rp3, rq0, rq1, rq2, rq3;
#define GEMM_GLOAD1(st, i, RP, RQ)                                                           \
    {                                                                                        \
        const int idx = (i)*256 + tid, kq = idx >> 7, c = idx & 127, gkq = (st)*KQ + kq;     \
        RP = P[(int64_t)gkq * a.ldp + p0 + c];                                               \
        const int seg = gkq / seg_kq, kin = gkq - seg * seg_kq;                              \
        RQ = Q[(int64_t)kin * a.ldq + (int64_t)seg * a.seg_col + q0 + c];                    \
    }
#define GEMM_GLOAD(st) GEMM_GLOAD1(st, 0, rp0, rq0) GEMM_GLOAD1(st, 1, rp1, rq1) GEMM_GLOAD1(st, 2, rp2, rq2) GEMM_GLOAD1(st, 3, rp3, rq3)
#define GEMM_LSTORE1(buf, i, RP, RQ)                                   \
    {                                                                  \
        const int idx = (i)*256 + tid, kq = idx >> 7, c = idx & 127;   \
        sP[buf][kq][c] = RP;                                           \
        sQ[buf][kq][c] = RQ;                                           \
    }
#define GEMM_LSTORE(buf) GEMM_LSTORE1(buf, 0, rp0, rq0) GEMM_LSTORE1(buf, 1, rp1, rq1) GEMM_LSTORE1(buf, 2, rp2, rq2) GEMM_LSTORE1(buf, 3, rp3, rq3)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // DMA variant: tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR
    // staging and no ds_write); wave w, piece i fills k-quad row 2i + (w>>1), columns (w&1)*64 .. +63.
#define GEMM_DMA(st, buf)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
        const int kq = 2 * i + (wave >> 1), c0 = (wave & 1) * 64, gkq = (st)*KQ + kq;                            \
        const int seg = gkq / seg_kq, kin = gkq - seg * seg_kq;                                                  \
        __builtin_amdgcn_global_load_lds(                                                                        \
            (const void __attribute__((address_space(1))) *)(P + (int64_t)gkq * a.ldp + p0 + c0 + lane),         \
            (void __attribute__((address_space(3))) *)(&sP[buf][kq][c0]), 16, 0, 0);                             \
        __builtin_amdgcn_global_load_lds(                                                                        \
            (const void __attribute__((address_space(1))) *)(Q + (int64_t)kin * a.ldq + (int64_t)seg * a.seg_col + q0 + c0 + lane), \
            (void __attribute__((address_space(3))) *)(&sQ[buf][kq][c0]), 16, 0, 0);                             \
    }
    if (DMA) {
        GEMM_DMA(0, 0)
    } else {
        GEMM_GLOAD(0)
        GEMM_LSTORE(0)
    }
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        const bool reload = st + 1 < nstage;
        if (reload) {
            if (DMA) { GEMM_DMA(st + 1, buf ^ 1) } else { GEMM_GLOAD(st + 1) }
        }
#pragma unroll
        for (int kb = 0; kb < KQ / 2; ++kb) {
            const float4 a0 = sP[buf][2 * kb + h][wp * 64 + l31], a1 = sP[buf][2 * kb + h][wp * 64 + 32 + l31];
            const float4 b0 = sQ[buf][2 * kb + h][wq * 64 + l31], b1 = sQ[buf][2 * kb + h][wq * 64 + 32 + l31];
            mfma4(acc[0][0], a0, b0); mfma4(acc[0][1], a0, b1);
            mfma4(acc[1][0], a1, b0); mfma4(acc[1][1], a1, b1);
        }
        if (!DMA && reload) { GEMM_LSTORE(buf ^ 1) }
        __syncthreads();   // also drains the LDS-DMA of the next tile (vmcnt(0) is part of the barrier's fence)
    }

    // ---------------------------------------------------------------- epilogue
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t q = q0 + wq * 64 + j * 32 + l31;
            float bq = 0.f;
            if (BIAS_Q) bq = (q < a.Qreal) ? a.bias[q] : 0.f;
            int spk = 0;
            if (COND) { spk = (int)a.cond_idx[q < a.Qreal ? q : 0]; spk = spk < 0 ? 0 : (spk > 7 ? 7 : spk); }   // ids are validated on the host; clamp keeps the read in bounds
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t p = p0 + wp * 64 + i * 32 + 8 * g + 4 * h;   // rows p..p+3
                float v[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (BIAS_P) {
                    float4 b = ld4(a.bias + p);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                if (COND) {
                    float4 c = ld4(a.cond_w + (p / 4 * 8 + spk) * 4);
                    v[0] += c.x; v[1] += c.y; v[2] += c.z; v[3] += c.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (BIAS_Q) v[e] += bq;
                    if (ACT == ACT_LRELU) v[e] = lrelu02(v[e]);
                    if (ACT == ACT_TANH) v[e] = tanhf_acc(v[e]);
                }
                if (OUT_MODE == OUT_K4) {
                    if (p < a.Pstore) st4(a.D + ((p / 4) * a.ldd + q) * 4, make_float4(v[0], v[1], v[2], v[3]));
                } else {
                    if (q < a.Qreal) {
                        const int64_t o = a.col_group ? (q / a.col_group) * a.col_stride + a.col_off + q % a.col_group : q;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (p + e < a.Pstore) a.D[(p + e) * a.ldd + o] = v[e];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Register-direct variant: no LDS, no barriers.  Every wave owns a (32*MT) x 64 output tile and pulls
// both operands straight from L1/L2 into VGPRs as K4 quads (one global_load_dwordx4 per operand per
// k-block of 8, requested one k-block ahead of the MFMAs that consume it).  The four waves of a
// workgroup take adjacent column tiles of the same row block, so the P (weight) quads of the second to
// fourth wave are L1 hits.  fp32 MFMA consumes so few operand bytes per cycle (2 dwords per lane per
// 64 cycles) that the cache path keeps up, and the waves never wait for each other.
// ------------------------------------------------------------------------------------------------
template <int MT, int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND>
__global__ __launch_bounds__(256, 2) void gemm_direct_kernel(GemmArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int64_t ntq = (a.Qpad + 255) / 256;
    const int64_t bid = blockIdx.x;
    const int64_t p0 = (bid / ntq) * (32 * MT), q0 = (bid % ntq) * 256 + wave * 64;
    if (q0 >= a.Qpad) return;
    if (a.q_limit && q0 >= *a.q_limit) return;

    const float4 *__restrict__ P = reinterpret_cast<const float4 *>(a.P) + p0 + l31;
    const float4 *__restrict__ Q = reinterpret_cast<const float4 *>(a.Q) + q0 + l31;
    const int nkb = a.K / 8, seg_kb = a.seg_k / 8;

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // running operand pointers: one 64-bit add per operand per k-block, everything else is an immediate offset
    const int64_t pstep = 2 * a.ldp, qstep = 2 * a.ldq;
    const float4 *pp = P + (int64_t)h * a.ldp;
    const float4 *qq = Q + (int64_t)h * a.ldq;
    int kin = 0, seg = 0;
    float4 an[MT], bn[2];
#define GD_LOAD()                                                        \
    {                                                                    \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) an[i] = pp[i * 32]; \
        bn[0] = qq[0]; bn[1] = qq[32];                                   \
        pp += pstep;                                                     \
        if (++kin == seg_kb) { kin = 0; ++seg; qq = Q + (int64_t)h * a.ldq + (int64_t)seg * a.seg_col; } \
        else qq += qstep;                                                \
    }
    GD_LOAD()
#pragma unroll 2
    for (int kb = 0; kb < nkb; ++kb) {
        float4 ac[MT], bc[2];
#pragma unroll
        for (int i = 0; i < MT; ++i) ac[i] = an[i];
        bc[0] = bn[0]; bc[1] = bn[1];
        if (kb + 1 < nkb) GD_LOAD()
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            mfma4(acc[i][0], ac[i], bc[0]);
            mfma4(acc[i][1], ac[i], bc[1]);
        }
    }
#undef GD_LOAD

#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t q = q0 + j * 32 + l31;
            float bq = 0.f;
            if (BIAS_Q) bq = (q < a.Qreal) ? a.bias[q] : 0.f;
            int spk = 0;
            if (COND) { spk = (int)a.cond_idx[q < a.Qreal ? q : 0]; spk = spk < 0 ? 0 : (spk > 7 ? 7 : spk); }   // ids are validated on the host; clamp keeps the read in bounds
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t p = p0 + i * 32 + 8 * g + 4 * h;
                float v[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (BIAS_P) {
                    float4 b = ld4(a.bias + p);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                if (COND) {
                    float4 c = ld4(a.cond_w + (p / 4 * 8 + spk) * 4);
                    v[0] += c.x; v[1] += c.y; v[2] += c.z; v[3] += c.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (BIAS_Q) v[e] += bq;
                    if (ACT == ACT_LRELU) v[e] = lrelu02(v[e]);
                    if (ACT == ACT_TANH) v[e] = tanhf_acc(v[e]);
                }
                if (OUT_MODE == OUT_K4) {
                    if (p < a.Pstore) st4(a.D + ((p / 4) * a.ldd + q) * 4, make_float4(v[0], v[1], v[2], v[3]));
                } else {
                    if (q < a.Qreal) {
                        const int64_t o = a.col_group ? (q / a.col_group) * a.col_stride + a.col_off + q % a.col_group : q;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (p + e < a.Pstore) a.D[(p + e) * a.ldd + o] = v[e];
                    }
                }
            }
        }
    }
}

template <int MT, int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND>
hipError_t launch_direct(const GemmArgs &a, hipStream_t s) {
    const int64_t nblk = (a.Ppad / (32 * MT)) * ((a.Qpad + 255) / 256);
    hipLaunchKernelGGL((gemm_direct_kernel<MT, OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>), dim3((unsigned)nblk), dim3(256), 0, s, a);
    return hipGetLastError();
}

template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND, bool DMA = false>
hipError_t launch(const GemmArgs &a, hipStream_t s) {
    int64_t nblk = (a.Ppad / TP) * (a.Qpad / TQ);
    hipLaunchKernelGGL((gemm_k4_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND, DMA>), dim3((unsigned)nblk), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace

int g_sdfa_gemm_variant = 0;   // 0 = LDS-tiled (default), 1/2 = register-direct (MT 4 / 2), 3 = LDS-tiled fed by LDS-DMA; all measured within 2 %

template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND>
hipError_t launch_any(const GemmArgs &a, hipStream_t s) {
    if (g_sdfa_gemm_variant == 1 && a.Ppad % 128 == 0) return launch_direct<4, OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>(a, s);
    if (g_sdfa_gemm_variant == 2 && a.Ppad % 64 == 0) return launch_direct<2, OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>(a, s);
    if (g_sdfa_gemm_variant == 3) return launch<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND, true>(a, s);
    return launch<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>(a, s);
}

hipError_t sdfa_launch_gemm(const GemmArgs &a, hipStream_t s) {
    if (a.K % 32 || a.seg_k % 32 || a.Ppad % TP || a.Qpad % TQ || a.K % a.seg_k) return hipErrorInvalidValue;
    const bool bp = a.bias && !a.bias_on_q, bq = a.bias && a.bias_on_q, cond = a.cond_w != nullptr;
    if (a.out_mode == OUT_ROW) {
        if (bq && a.act == ACT_NONE && !cond) return launch_any<OUT_ROW, ACT_NONE, false, true, false>(a, s);
        return hipErrorInvalidValue;
    }
    if (bq) return hipErrorInvalidValue;
    if (cond) {
        if (a.act == ACT_LRELU && bp) return launch_any<OUT_K4, ACT_LRELU, true, false, true>(a, s);
        return hipErrorInvalidValue;
    }
    if (a.act == ACT_NONE) return bp ? launch<OUT_K4, ACT_NONE, true, false, false>(a, s)
                                     : launch_any<OUT_K4, ACT_NONE, false, false, false>(a, s);
    if (a.act == ACT_TANH && bp) return launch_any<OUT_K4, ACT_TANH, true, false, false>(a, s);
    if (a.act == ACT_LRELU && bp) return launch_any<OUT_K4, ACT_LRELU, true, false, false>(a, s);
    return hipErrorInvalidValue;
}

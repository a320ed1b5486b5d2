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

#ifdef SDFA_STAMPS
__device__ unsigned long long g_stamp[8];
#endif

extern thread_local int g_sdfa_gemm_variant;   // "gemm_variant" option, defined below

namespace {

constexpr int TP = 128, TQ = 128, KQ = 8;   // tile p, tile q, k-quads per stage

// Epilogue of one 32x32 accumulator tile: rows p_tile + (r&3) + 8*(r>>2) + 4*h, column q (this lane's).
template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND>
__device__ __forceinline__ void store_tile(const GemmArgs &a, const f32x16 &acc, int64_t p_tile, int64_t q, int h) {
    float bq = 0.f;
    if (BIAS_Q) bq = (q < a.Qreal) ? a.bias[q] : 0.f;
    int spk = 0;
    if (COND) { spk = (int)a.cond_idx[q < a.Qreal ? q : 0]; spk = spk < 0 ? 0 : (spk > 7 ? 7 : spk); }   // ids are validated on the host; clamp keeps the read in bounds
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int64_t p = p_tile + 8 * g + 4 * h;   // rows p..p+3
        float v[4] = {acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
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
        } else if (q < a.Qreal) {
            const int64_t o = a.col_group ? (q / a.col_group) * a.col_stride + a.col_off + q % a.col_group : q;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (p + e < a.Pstore) {
                    a.D[(p + e) * a.ldd + o] = v[e];
                    for (int x = 0; x < a.n_extra; ++x) a.D_extra[x][(p + e) * a.ldd + o] = v[e];
                }
        }
    }
}


// WT = MFMA tiles per wave and dimension: 2 -> the 128 x 128 workgroup tile described above; 1 -> a 64 x 64 tile (each wave one
// 32 x 32 block, 32 KiB of LDS, four workgroups per CU) for the launches that would otherwise put at most one or two workgroups on
// a CU: the MLP layers, the attention query path, everything at single-clip sizes.  Such a launch is a latency chain, not a
// throughput problem -- profiles/r03_per_launch.txt: a 256-deep MLP GEMM took 43-45 us whether it had 8 or 256 workgroups, 5 us
// per 32-deep stage for 1.7 us of MFMAs -- and four times as many workgroups of a quarter of the work overlap each other's
// round trips.  Same k order per accumulator: bit-identical to the 128 x 128 tile.
// WTP (round 4) = tiles per wave along P when it differs from WT: <WT 2, WTP 1> is a 64 (P) x 128 (Q) workgroup tile, 48 KiB of LDS,
// three workgroups per CU -- for the 8192-deep frequency projection of a single clip, whose 364 tiles of 128 x 128 (at two per CU)
// left 108 CUs with two tiles and 148 with one: the launch took two tiles' time for 1.4 tiles of work per CU.  728 half tiles at
// three per CU are 2.8 per CU.  Same k order per accumulator again: bit-identical.
template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND, int WT, int WTP = WT>
__global__ __launch_bounds__(256, WTP != WT ? 3 : (WT == 2 ? 2 : 4)) void gemm_k4_kernel(GemmArgs a) {
    constexpr int TPW = 64 * WT, SH = WT == 2 ? 7 : 6, RPP = 256 >> SH;       // Q: tile edge, log2 of it, k-quad rows per staging pass
    constexpr int TPWP = 64 * WTP, SHP = WTP == 2 ? 7 : 6, RPPP = 256 >> SHP;  // P: the same
    __shared__ float4 sP[2][KQ][TPWP];
    __shared__ float4 sQ[2][KQ][TPW];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wq = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // tile order: all p-tiles of one q-tile are dispatched together, so the streamed activation tile (Q) comes
    // from HBM once and from L2 for the other row blocks; the weight slab (P) is small and L2-resident anyway
    const int64_t ntp = a.Ppad / TPWP;
    const int64_t bid = blockIdx.x;
    const int64_t tp = bid % ntp, tq = bid / ntp;
    const int64_t p0 = tp * TPWP, q0 = tq * TPW;
    if (a.q_limit && q0 >= *a.q_limit) return;

    const float4 *__restrict__ P = reinterpret_cast<const float4 *>(a.P);
    const float4 *__restrict__ Q = reinterpret_cast<const float4 *>(a.Q);
    const int nkq_total = a.K / 4;
    const int seg_kq = a.seg_k / 4;
    const int nstage = nkq_total / KQ;
    // tile-major Q (frequency-LSTM hidden states; WT = 2 only): a column block of 128 is one contiguous [K/4][128] slab
    const int64_t qrow = a.q_tile_major ? 128 : a.ldq;               // float4 elements between consecutive k-quads
    const float4 *Pn = P + p0;
    const float4 *Qn = a.q_tile_major ? Q + (q0 >> 7) * (int64_t)a.q_slab_rows * 128 : Q + q0;
    int kin_n = 0;
    const unsigned boffP = (unsigned)(((tid >> SHP) * a.ldp + (tid & (TPWP - 1))) * 16);
    const unsigned boffQ = (unsigned)(((tid >> SH) * qrow + (tid & (TPW - 1))) * 16);

    // Register staging runs TWO tiles ahead of the MFMAs: while tile st is multiplied out of LDS, tile st+1 sits
    // in one register set (written to the other LDS buffer at the end of the stage) and the loads of tile st+2 are
    // in flight into the second set.  One stage is ~4,100 MFMA cycles per wave; an HBM-streamed operand (the 8192-deep
    // frequency projection) takes longer than that to arrive, which held the kernel at 76 % with a single tile ahead.
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;   // set A: P quads ra*, Q quads rb*   (one tile per wave along an operand: two quads of it)
    float4 rc0, rc1, rc2, rc3, rd0, rd1, rd2, rd3;   // set B
    // Addressing of the staging loads: a UNIFORM running base per operand (scalar registers, advanced by scalar adds)
    // plus one loop-invariant 32-bit byte offset per thread -- the global_load "saddr + voffset" form, so a load costs
    // no vector-ALU instruction.  (fp32 MFMA and VALU instructions share the SIMD's issue cycles: measured with
    // tools/mfma_valu.hip, every VALU instruction takes ~4 cycles away from the matrix pipe.  The first version of
    // this kernel recomputed 64-bit addresses and the segment division per load: ~170 VALU instructions per stage of
    // 64 MFMAs, i.e. 15 % of the pipe.)
#define GEMM_PLOAD1(i, RP) RP = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Pn + (i) * RPPP * a.ldp) + boffP);
#define GEMM_QLOAD1(i, RQ) RQ = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Qn + (i) * RPP * qrow) + boffQ);
#define GEMM_ADVANCE()                                                                                  \
    {                                                                                                   \
        Pn += KQ * a.ldp;                                                                               \
        kin_n += KQ;                                                                                    \
        if (kin_n == seg_kq) { kin_n = 0; Qn += a.seg_col - (int64_t)(seg_kq - KQ) * qrow; }            \
        else Qn += KQ * qrow;                                                                           \
    }
    // stages are requested strictly in order (0, 1, 2, ...), so the bases just run forward
#define GEMM_GLOAD_SET(P0, P1, P2, P3, Q0, Q1, Q2, Q3)                                                   \
    {                                                                                                   \
        GEMM_PLOAD1(0, P0) GEMM_QLOAD1(0, Q0) GEMM_PLOAD1(1, P1) GEMM_QLOAD1(1, Q1)                     \
        if constexpr (WTP == 2) { GEMM_PLOAD1(2, P2) } if constexpr (WT == 2) { GEMM_QLOAD1(2, Q2) }    \
        if constexpr (WTP == 2) { GEMM_PLOAD1(3, P3) } if constexpr (WT == 2) { GEMM_QLOAD1(3, Q3) }    \
        GEMM_ADVANCE()                                                                                  \
    }
#define GEMM_GLOAD_A(st) GEMM_GLOAD_SET(ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3)
#define GEMM_GLOAD_B(st) GEMM_GLOAD_SET(rc0, rc1, rc2, rc3, rd0, rd1, rd2, rd3)
#define GEMM_PSTORE1(buf, i, RP) { const int idx = (i)*256 + tid; sP[buf][idx >> SHP][idx & (TPWP - 1)] = RP; }
#define GEMM_QSTORE1(buf, i, RQ) { const int idx = (i)*256 + tid; sQ[buf][idx >> SH][idx & (TPW - 1)] = RQ; }
#define GEMM_LSTORE_SET(buf, P0, P1, P2, P3, Q0, Q1, Q2, Q3)                                             \
    {                                                                                                   \
        GEMM_PSTORE1(buf, 0, P0) GEMM_QSTORE1(buf, 0, Q0) GEMM_PSTORE1(buf, 1, P1) GEMM_QSTORE1(buf, 1, Q1) \
        if constexpr (WTP == 2) { GEMM_PSTORE1(buf, 2, P2) } if constexpr (WT == 2) { GEMM_QSTORE1(buf, 2, Q2) } \
        if constexpr (WTP == 2) { GEMM_PSTORE1(buf, 3, P3) } if constexpr (WT == 2) { GEMM_QSTORE1(buf, 3, Q3) } \
    }
#define GEMM_LSTORE_A(buf) GEMM_LSTORE_SET(buf, ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3)
#define GEMM_LSTORE_B(buf) GEMM_LSTORE_SET(buf, rc0, rc1, rc2, rc3, rd0, rd1, rd2, rd3)
#define GEMM_COMPUTE(buf)                                                                                              \
    _Pragma("unroll") for (int kb = 0; kb < KQ / 2; ++kb) {                                                            \
        float4 fa[WTP], fb[WT];                                                                                        \
        _Pragma("unroll") for (int i = 0; i < WTP; ++i) fa[i] = sP[buf][2 * kb + h][wp * 32 * WTP + 32 * i + l31];     \
        _Pragma("unroll") for (int i = 0; i < WT; ++i) fb[i] = sQ[buf][2 * kb + h][wq * 32 * WT + 32 * i + l31];       \
        mfma_block<WTP, WT>(acc, fa, fb);                                                                              \
    }

    f32x16 acc[WTP][WT];
#pragma unroll
    for (int i = 0; i < WTP; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    {
#ifdef SDFA_STAMPS
        // DIAGNOSTIC BUILD ONLY (make STAMPS=1): where do a stage's cycles go?  s_memtime around each phase, summed per wave.
        unsigned long long t0, t1, t2, t3, t4, sum_load = 0, sum_mfma = 0, sum_store = 0, sum_bar = 0;
#define STAMP(t) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define STAMP(t)
#endif
        GEMM_GLOAD_A(0)
        GEMM_LSTORE_A(0)
        if (nstage > 1) { GEMM_GLOAD_A(1) }
        __syncthreads();
        int st = 0;
        for (; st + 1 < nstage; st += 2) {
            // even stage: LDS buffer 0 holds tile st, set A holds tile st+1
            STAMP(t0)
            if (st + 2 < nstage) { GEMM_GLOAD_B(st + 2) }
            STAMP(t1)
            GEMM_COMPUTE(0)
            STAMP(t2)
            GEMM_LSTORE_A(1)
            STAMP(t3)
            __syncthreads();
            STAMP(t4)
#ifdef SDFA_STAMPS
            sum_load += t1 - t0; sum_mfma += t2 - t1; sum_store += t3 - t2; sum_bar += t4 - t3;
#endif
            // odd stage: LDS buffer 1 holds tile st+1, set B holds tile st+2
            if (st + 3 < nstage) { GEMM_GLOAD_A(st + 3) }
            GEMM_COMPUTE(1)
            if (st + 2 < nstage) { GEMM_LSTORE_B(0) }
            __syncthreads();
        }
        if (st < nstage) { GEMM_COMPUTE(0) }   // odd stage count: the last tile is already in LDS buffer 0
#ifdef SDFA_STAMPS
        if (lane == 0 && nstage >= 64) {
            atomicAdd(&g_stamp[0], sum_load); atomicAdd(&g_stamp[1], sum_mfma); atomicAdd(&g_stamp[2], sum_store);
            atomicAdd(&g_stamp[3], sum_bar); atomicAdd(&g_stamp[4], (unsigned long long)((nstage + 1) / 2));
        }
#endif
    }

    // ---------------------------------------------------------------- epilogue
#pragma unroll
    for (int i = 0; i < WTP; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
            store_tile<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>(a, acc[i][j], p0 + wp * 32 * WTP + i * 32, q0 + wq * 32 * WT + j * 32 + l31, h);
}

// Which launches take the 64 x 64 tile (gemm_k4_kernel<.., 1>): by the number of 128 x 128 tiles the problem has.  Fewer than half a
// tile per CU: always (single-clip sizes; a quarter of the work per workgroup and four times the workgroups).  Up to two per CU: when
// the contraction is short (K <= 512: at most 16 stages, the launch is one workgroup's latency chain); a long contraction at that
// count already runs the matrix pipe at 87 % (the attention query Conv1d, K = 1536: 94 us for 82 us of MFMAs).
// "gemm_variant" 6 forces it wherever it applies, 2 forbids it.
inline bool small_tile_pays(const GemmArgs &a) {
    if (a.q_tile_major || g_sdfa_gemm_variant == 2) return false;
    if (g_sdfa_gemm_variant == 6) return true;
    const int64_t cus = sdfa_cu_count();
    const int64_t n128 = (a.Ppad / TP) * (a.Qpad / TQ);
    return n128 * 2 < cus || (n128 <= 2 * cus && a.K <= 512);
}

// The 64 (P) x 128 (Q) tile: the tile-major 8192-deep frequency projection while its 128 x 128 tiles number fewer than four per CU
// (a single clip, a few clips: most of the launched tiles exit at the device-side column limit in column-sharing mode, and what
// is left quantises badly at two workgroups per CU).  "gemm_variant" 10 = never, 11 = wherever the LDS-tiled kernel runs.
inline bool half_tile_pays(const GemmArgs &a) {
    if (g_sdfa_gemm_variant == 10 || a.Ppad % 64) return false;
    if (g_sdfa_gemm_variant == 11) return true;
    return a.q_tile_major && (a.Ppad / TP) * (a.Qpad / TQ) < 4 * (int64_t)sdfa_cu_count();
}

template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND>
hipError_t launch(const GemmArgs &a, hipStream_t s) {
    if (half_tile_pays(a)) {
        const int64_t nblk = (a.Ppad / 64) * (a.Qpad / TQ);
        hipLaunchKernelGGL((gemm_k4_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND, 2, 1>), dim3((unsigned)nblk), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    if (small_tile_pays(a)) {
        const int64_t nblk = (a.Ppad / 64) * (a.Qpad / 64);
        hipLaunchKernelGGL((gemm_k4_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND, 1>), dim3((unsigned)nblk), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    int64_t nblk = (a.Ppad / TP) * (a.Qpad / TQ);
    hipLaunchKernelGGL((gemm_k4_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND, 2>), dim3((unsigned)nblk), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// bf16-MFMA GEMM of the mixed-precision modes (GemmArgs.terms, set by sdfa_model_set_precision; also reachable as
// gemm_variant 4 = TERMS 3 for A/B runs).  TERMS 1: operands rounded to bf16 while they are staged into LDS.
// TERMS 3 (split-bf16): operands are split into hi = bf16(x) and lo = bf16(x - hi) and every product runs as three MFMAs
//   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi      (v_mfma_f32_32x32x16_bf16, fp32 accumulate)
// i.e. 16 significand bits per operand at 16/3 = 5.3x the fp32 MFMA rate.  Measured operand-truncation error of
// the whole model at 2 bf16 terms: 3.5e-6 on dgrad against the 1e-4 budget (profiles/r01_precision_sweep.json).
// LDS image per plane: [k/8][column][8 bf16], so one ds_read_b128 is one MFMA fragment (lane = column, half = k-group).
// The accumulator layout equals the fp32 MFMA's, so the epilogue is shared.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const float4 &x0, const float4 &x1, bf16x8 &hi, bf16x8 &lo) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hb = (__bf16)x[e];
        hi[e] = hb;
        lo[e] = (__bf16)(x[e] - (float)hb);
    }
}

template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND, int TERMS>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs a) {
    constexpr bool LO = TERMS > 1;               // TERMS 1: plain bf16 operands; 3: hi/lo planes, three MFMAs per product
    constexpr int NB = LO ? 2 : 1;
    __shared__ bf16x8 sPh[2][4][TP], sQh[2][4][TQ], sPl[NB][4][TP], sQl[NB][4][TQ];   // 4 x 16 KiB (2 + 2 x 8 for TERMS 1)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wq = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int64_t ntp = a.Ppad / TP, bid = blockIdx.x;
    const int64_t p0 = (bid % ntp) * TP, q0 = (bid / ntp) * TQ;
    if (a.q_limit && q0 >= *a.q_limit) return;

    const float4 *__restrict__ P = reinterpret_cast<const float4 *>(a.P);
    const float4 *__restrict__ Q = reinterpret_cast<const float4 *>(a.Q);
    const int seg_kq = a.seg_k / 4, nstage = a.K / 32;

    float4 rp[2][2], rq[2][2];   // [item][quad of the k-group]
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define BX_GLOAD(st)                                                                              \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                            \
        const int idx = it * 256 + tid, g = idx >> 7, c = idx & 127, gkq = (st)*8 + 2 * g;        \
        const int seg = gkq / seg_kq, kin = gkq - seg * seg_kq;                                   \
        rp[it][0] = P[(int64_t)gkq * a.ldp + p0 + c];                                             \
        rp[it][1] = P[(int64_t)(gkq + 1) * a.ldp + p0 + c];                                       \
        if (a.q_tile_major) {                                                                     \
            rq[it][0] = Q[((q0 >> 7) * (int64_t)a.q_slab_rows + gkq) * 128 + c];                      \
            rq[it][1] = Q[((q0 >> 7) * (int64_t)a.q_slab_rows + gkq + 1) * 128 + c];                  \
        } else {                                                                                  \
            rq[it][0] = Q[(int64_t)kin * a.ldq + (int64_t)seg * a.seg_col + q0 + c];              \
            rq[it][1] = Q[(int64_t)(kin + 1) * a.ldq + (int64_t)seg * a.seg_col + q0 + c];        \
        }                                                                                         \
    }
#define BX_LSTORE(buf)                                                                            \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                            \
        const int idx = it * 256 + tid, g = idx >> 7, c = idx & 127;                              \
        bf16x8 hi, lo;                                                                            \
        split8(rp[it][0], rp[it][1], hi, lo); sPh[buf][g][c] = hi; if (LO) sPl[buf][g][c] = lo;   \
        split8(rq[it][0], rq[it][1], hi, lo); sQh[buf][g][c] = hi; if (LO) sQl[buf][g][c] = lo;   \
    }

    BX_GLOAD(0)
    BX_LSTORE(0)
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        const bool reload = st + 1 < nstage;
        if (reload) { BX_GLOAD(st + 1) }
#pragma unroll
        for (int m = 0; m < 2; ++m) {   // two k-steps of 16
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = sPh[buf][2 * m + h][wp * 64 + i * 32 + l31];
                bh[i] = sQh[buf][2 * m + h][wq * 64 + i * 32 + l31];
                if (LO) { al[i] = sPl[buf][2 * m + h][wp * 64 + i * 32 + l31]; bl[i] = sQl[buf][2 * m + h][wq * 64 + i * 32 + l31]; }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (LO) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (reload) { BX_LSTORE(buf ^ 1) }
        __syncthreads();
    }
#undef BX_GLOAD
#undef BX_LSTORE
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            store_tile<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>(a, acc[i][j], p0 + wp * 64 + i * 32, q0 + wq * 64 + j * 32 + l31, h);
}

template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND, int TERMS>
hipError_t launch_bf16(const GemmArgs &a, hipStream_t s) {
    const int64_t nblk = (a.Ppad / TP) * (a.Qpad / TQ);
    hipLaunchKernelGGL((gemm_bf16_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND, TERMS>), dim3((unsigned)nblk), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Six-product split (SDFA_PREC_BF16X6, round 4): operands as THREE bf16 terms, x = hi + mid + lo (24 significand bits = fp32's), and
// every product as the six partial products down to 2^-16 of the leading one,
//   a*b ~= a_hi*b_lo + a_mid*b_mid + a_lo*b_hi + a_hi*b_mid + a_mid*b_hi + a_hi*b_hi      (smallest first; the three dropped terms are
// below 2^-24 relative), i.e. fp32-equivalent products at 16 / 6 = 2.7x the fp32 MFMA rate.  Same tile, staging and epilogue as
// gemm_bf16_kernel; three planes per operand and buffer = 96 KiB of dynamic LDS, one workgroup per CU.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split8x3(const float4 &x0, const float4 &x1, bf16x8 &hi, bf16x8 &mid, bf16x8 &lo) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hb = (__bf16)x[e];
        const float r1 = x[e] - (float)hb;
        const __bf16 mb = (__bf16)r1;
        hi[e] = hb;
        mid[e] = mb;
        lo[e] = (__bf16)(r1 - (float)mb);
    }
}

template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND>
__global__ __launch_bounds__(256) void gemm_bf16x6_kernel(GemmArgs a) {
    extern __shared__ bf16x8 s6[];      // [2 buffers][P | Q][3 planes][4 octets][128]
    auto S6 = [&](int buf, int pq, int plane, int oct) { return s6 + (((size_t)(buf * 2 + pq) * 3 + plane) * 4 + oct) * 128; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wq = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int64_t ntp = a.Ppad / TP, bid = blockIdx.x;
    const int64_t p0 = (bid % ntp) * TP, q0 = (bid / ntp) * TQ;
    if (a.q_limit && q0 >= *a.q_limit) return;

    const float4 *__restrict__ P = reinterpret_cast<const float4 *>(a.P);
    const float4 *__restrict__ Q = reinterpret_cast<const float4 *>(a.Q);
    const int seg_kq = a.seg_k / 4, nstage = a.K / 32;

    float4 rp[2][2], rq[2][2];   // [item][quad of the k-group]
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define B6_GLOAD(st)                                                                              \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                            \
        const int idx = it * 256 + tid, g = idx >> 7, c = idx & 127, gkq = (st)*8 + 2 * g;        \
        const int seg = gkq / seg_kq, kin = gkq - seg * seg_kq;                                   \
        rp[it][0] = P[(int64_t)gkq * a.ldp + p0 + c];                                             \
        rp[it][1] = P[(int64_t)(gkq + 1) * a.ldp + p0 + c];                                       \
        if (a.q_tile_major) {                                                                     \
            rq[it][0] = Q[((q0 >> 7) * (int64_t)a.q_slab_rows + gkq) * 128 + c];                      \
            rq[it][1] = Q[((q0 >> 7) * (int64_t)a.q_slab_rows + gkq + 1) * 128 + c];                  \
        } else {                                                                                  \
            rq[it][0] = Q[(int64_t)kin * a.ldq + (int64_t)seg * a.seg_col + q0 + c];              \
            rq[it][1] = Q[(int64_t)(kin + 1) * a.ldq + (int64_t)seg * a.seg_col + q0 + c];        \
        }                                                                                         \
    }
#define B6_LSTORE(buf)                                                                            \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                            \
        const int idx = it * 256 + tid, g = idx >> 7, c = idx & 127;                              \
        bf16x8 hi, mid, lo;                                                                       \
        split8x3(rp[it][0], rp[it][1], hi, mid, lo); S6(buf, 0, 0, g)[c] = hi; S6(buf, 0, 1, g)[c] = mid; S6(buf, 0, 2, g)[c] = lo; \
        split8x3(rq[it][0], rq[it][1], hi, mid, lo); S6(buf, 1, 0, g)[c] = hi; S6(buf, 1, 1, g)[c] = mid; S6(buf, 1, 2, g)[c] = lo; \
    }

    B6_GLOAD(0)
    B6_LSTORE(0)
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        const bool reload = st + 1 < nstage;
        if (reload) { B6_GLOAD(st + 1) }
#pragma unroll
        for (int m = 0; m < 2; ++m) {   // two k-steps of 16
            bf16x8 ap[3][2], bq[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ap[pl][i] = S6(buf, 0, pl, 2 * m + h)[wp * 64 + i * 32 + l31];
                    bq[pl][i] = S6(buf, 1, pl, 2 * m + h)[wq * 64 + i * 32 + l31];
                }
            // one partial product over all four accumulators at a time (no two dependent MFMAs in a row), smallest products first
            constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PA[t]][i], bq[PB[t]][j], acc[i][j], 0, 0, 0);
        }
        if (reload) { B6_LSTORE(buf ^ 1) }
        __syncthreads();
    }
#undef B6_GLOAD
#undef B6_LSTORE
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            store_tile<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>(a, acc[i][j], p0 + wp * 64 + i * 32, q0 + wq * 64 + j * 32 + l31, h);
}

template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND>
hipError_t launch_bf16x6(const GemmArgs &a, hipStream_t s) {
    const size_t lds = (size_t)2 * 2 * 3 * 4 * 128 * sizeof(bf16x8);   // 96 KiB
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16x6_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int64_t nblk = (a.Ppad / TP) * (a.Qpad / TQ);
    hipLaunchKernelGGL((gemm_bf16x6_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>), dim3((unsigned)nblk), dim3(256), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// bf16-MFMA GEMM, 256 x 256 tile (mixed-precision modes, P and Q multiples of 256): with the matrix work cut 5-16x the
// 128 x 128 kernel above is bound by its operand stream (32 FLOP per staged byte: the 8192-deep frequency projection
// took 44 ms in every mode); this tile stages half the bytes per FLOP.  8 waves (2 x 4, each 128 x 64 = 4 x 2 MFMA
// tiles), operands split into bf16 planes while they are staged, 128 KiB of LDS, one workgroup per CU.
// ------------------------------------------------------------------------------------------------
template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND, int TERMS>
__global__ __launch_bounds__(512, 2) void gemm_bf16_big_kernel(GemmArgs a) {
    constexpr bool LO = TERMS > 1;
    constexpr int BT = 256, NPL = LO ? 2 : 1;
    extern __shared__ bf16x8 sBb[];                        // [2 bufs][P | Q][NPL planes][4 octets][256]
    auto SPL = [&](int buf, int pq, int plane) { return sBb + ((size_t)((buf * 2 + pq) * NPL + plane) * 4) * BT; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 2, wq = wave & 3, l31 = lane & 31, h = lane >> 5;
    const int64_t ntp = a.Ppad / BT, bid = blockIdx.x;
    const int64_t p0 = (bid % ntp) * BT, q0 = (bid / ntp) * BT;
    if (a.q_limit && q0 >= *a.q_limit) return;

    const float4 *__restrict__ P = reinterpret_cast<const float4 *>(a.P);
    const float4 *__restrict__ Q = reinterpret_cast<const float4 *>(a.Q);
    const int seg_kq = a.seg_k / 4, nstage = a.K / 32;
    // staging: item it = 0, 1 of a thread is (octet g = 2*it + (tid >> 8), column c = tid & 255): K4 quads 2g and 2g+1.
    // Uniform running bases + loop-invariant thread byte offsets, as in gemm_k4_kernel.
    const int c = tid & 255, g0 = tid >> 8;
    const int64_t qrow = a.q_tile_major ? 128 : a.ldq;
    const float4 *Pn = P + p0;
#ifdef SDFA_GEMM_SAMESLAB   /* timing experiment only: every workgroup streams the SAME column block (cache-resident) */
    const float4 *Qn = a.q_tile_major ? Q : Q + q0;
#else
    const float4 *Qn = a.q_tile_major ? Q + (q0 >> 7) * (int64_t)a.q_slab_rows * 128 : Q + q0;
#endif
    int kin_n = 0;
    const unsigned boffP = (unsigned)((2 * g0 * a.ldp + c) * 16);
    const unsigned boffQ = a.q_tile_major ? (unsigned)((((int64_t)(c >> 7) * a.q_slab_rows + 2 * g0) * 128 + (c & 127)) * 16)
                                          : (unsigned)((2 * g0 * qrow + c) * 16);
    // Software pipeline (round 4): the registers hold stage st+1 while stage st multiplies; each of a thread's four octets (two items x
    // P, Q) is split and written to the OTHER LDS buffer between the MFMAs of stage st -- three pairs of elements, then the last pair,
    // the LDS writes and the REFILL of the same registers with stage st+2 (consumed a whole stage later), one such piece behind every
    // row of six MFMAs, pinned with scheduling barriers.  (Before: split + stores behind the stage's last MFMA, 200 vector
    // instructions per wave in front of every barrier.)  Requests are unconditional -- one under a branch makes every later vmcnt
    // wait a full drain -- so the running pointers stop at the last stage, which the requests behind it re-read (dropped).
    float4 rp[2][2], rq[2][2];   // [item][quad of the octet]
    int next_k = 0;              // stage the running pointers address
#define BB_SB() __builtin_amdgcn_sched_barrier(0);
#define BB_LOADP(it) { _Pragma("unroll") for (int e = 0; e < 2; ++e) rp[it][e] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Pn + (4 * (it) + e) * a.ldp) + boffP); }
#define BB_LOADQ(it) { _Pragma("unroll") for (int e = 0; e < 2; ++e) rq[it][e] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Qn + (4 * (it) + e) * qrow) + boffQ); }
#define BB_ADVANCE()                                                                                            \
    if (++next_k < nstage) {                                                                                    \
        Pn += KQ * a.ldp;                                                                                       \
        kin_n += KQ;                                                                                            \
        if (kin_n == seg_kq) { kin_n = 0; Qn += a.seg_col - (int64_t)(seg_kq - KQ) * qrow; }                    \
        else Qn += KQ * qrow;                                                                                   \
    }
    // elements 2e, 2e+1 of an octet held as two float4 -> the planes
#define BB_PAIR(R, e)                                                                                           \
    {                                                                                                           \
        const float x0 = f4c(R[(e) >> 1], (2 * (e)) & 3), x1 = f4c(R[(e) >> 1], (2 * (e) + 1) & 3);             \
        const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;                                                          \
        shi[2 * (e)] = h0; shi[2 * (e) + 1] = h1;                                                               \
        if (LO) { slo[2 * (e)] = (__bf16)(x0 - (float)h0); slo[2 * (e) + 1] = (__bf16)(x1 - (float)h1); }       \
    }
#define BB_WRITE(buf, pq, it) { SPL(buf, pq, 0)[(2 * (it) + g0) * BT + c] = shi; if (LO) SPL(buf, pq, NPL - 1)[(2 * (it) + g0) * BT + c] = slo; }
    // the two pieces of octet o (0, 1 = P items; 2, 3 = Q items) of the stage in registers
#define BB_PIECE_A(o) { if ((o) < 2) { BB_PAIR(rp[(o) & 1], 0) BB_PAIR(rp[(o) & 1], 1) BB_PAIR(rp[(o) & 1], 2) } else { BB_PAIR(rq[(o) & 1], 0) BB_PAIR(rq[(o) & 1], 1) BB_PAIR(rq[(o) & 1], 2) } }
#define BB_PIECE_B(buf, o)                                                                                      \
    {                                                                                                           \
        if ((o) < 2) { BB_PAIR(rp[(o) & 1], 3) BB_WRITE((buf) ^ 1, 0, (o) & 1) BB_LOADP((o) & 1) }              \
        else { BB_PAIR(rq[(o) & 1], 3) BB_WRITE((buf) ^ 1, 1, (o) & 1) BB_LOADQ((o) & 1) }                      \
    }
#define BB_ROW(i)                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                             \
        if (LO) {                                                                                               \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);              \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);              \
        }                                                                                                       \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);                  \
    }
#define BB_READS(buf, m)                                                                                        \
    {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
            ah[i] = SPL(buf, 0, 0)[(2 * (m) + h) * BT + wp * 128 + i * 32 + l31];                               \
            if (LO) al[i] = SPL(buf, 0, NPL - 1)[(2 * (m) + h) * BT + wp * 128 + i * 32 + l31];                 \
        }                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                         \
            bh[j] = SPL(buf, 1, 0)[(2 * (m) + h) * BT + wq * 64 + j * 32 + l31];                                \
            if (LO) bl[j] = SPL(buf, 1, NPL - 1)[(2 * (m) + h) * BT + wq * 64 + j * 32 + l31];                  \
        }                                                                                                       \
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    BB_LOADP(0) BB_LOADQ(0) BB_LOADP(1) BB_LOADQ(1) BB_ADVANCE()
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int g = 2 * it + g0;
        bf16x8 hi, lo;
        split8(rp[it][0], rp[it][1], hi, lo); SPL(0, 0, 0)[g * BT + c] = hi; if (LO) SPL(0, 0, NPL - 1)[g * BT + c] = lo;
        split8(rq[it][0], rq[it][1], hi, lo); SPL(0, 1, 0)[g * BT + c] = hi; if (LO) SPL(0, 1, NPL - 1)[g * BT + c] = lo;
    }
    BB_LOADP(0) BB_LOADQ(0) BB_LOADP(1) BB_LOADQ(1) BB_ADVANCE()
    __syncthreads();
#ifdef SDFA_GEMM_LOADONLY   /* timing experiment only: the operand stream without LDS staging and matrix work */
    for (int st = 0; st + 2 < nstage; ++st) {
        acc[0][0][0] += rp[0][0].x + rp[0][1].y + rp[1][0].z + rp[1][1].w + rq[0][0].x + rq[0][1].y + rq[1][0].z + rq[1][1].w;
        BB_LOADP(0) BB_LOADQ(0) BB_LOADP(1) BB_LOADQ(1) BB_ADVANCE()
    }
#else
    // (behind the last stage the registers hold stale operands: they are split and written to the buffer nobody reads any more)
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        bf16x8 ah[4], al[4], bh[2], bl[2], shi, slo;
        BB_SB() BB_READS(buf, 0) BB_SB()
        BB_ROW(0) BB_SB() BB_PIECE_A(0) BB_SB()
        BB_ROW(1) BB_SB() BB_PIECE_B(buf, 0) BB_SB()
        BB_ROW(2) BB_SB() BB_PIECE_A(2) BB_SB()
        BB_ROW(3) BB_SB() BB_PIECE_B(buf, 2) BB_READS(buf, 1) BB_SB()
        BB_ROW(0) BB_SB() BB_PIECE_A(1) BB_SB()
        BB_ROW(1) BB_SB() BB_PIECE_B(buf, 1) BB_SB()
        BB_ROW(2) BB_SB() BB_PIECE_A(3) BB_SB()
        BB_ROW(3) BB_SB() BB_PIECE_B(buf, 3) BB_ADVANCE() BB_SB()
        __syncthreads();
    }
#endif
#undef BB_SB
#undef BB_LOADP
#undef BB_LOADQ
#undef BB_ADVANCE
#undef BB_PAIR
#undef BB_WRITE
#undef BB_PIECE_A
#undef BB_PIECE_B
#undef BB_ROW
#undef BB_READS
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            store_tile<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>(a, acc[i][j], p0 + wp * 128 + i * 32, q0 + wq * 64 + j * 32 + l31, h);
}

template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND, int TERMS>
hipError_t launch_bf16_big(const GemmArgs &a, hipStream_t s) {
    const size_t lds = (size_t)2 * 2 * (TERMS > 1 ? 2 : 1) * 4 * 256 * sizeof(bf16x8);   // 128 KiB (split) / 64 KiB
    {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16_big_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND, TERMS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int64_t nblk = (a.Ppad / 256) * (a.Qpad / 256);
    hipLaunchKernelGGL((gemm_bf16_big_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND, TERMS>), dim3((unsigned)nblk), dim3(512), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Six-product split on the 256 x 256 tile (P and Q multiples of 256): the 128 x 128 six-product kernel above is bound by its operand
// stream like the other 128-tile bf16 kernels (its frequency projection took what the exact fp32 gemm_fat_kernel takes); this tile
// stages half the bytes per FLOP.  Three planes per operand leave room for 16-deep stages only: [2 buffers][P | Q][3 planes]
// [2 octets][256] = 96 KiB.  8 waves (2 x 4, each 128 x 64), one workgroup per CU.
// ------------------------------------------------------------------------------------------------
template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND>
__global__ __launch_bounds__(512, 2) void gemm_bf16x6_big_kernel(GemmArgs a) {
    constexpr int BT = 256, KQS = 4;                       // tile edge; k-quads per stage (16 k)
    extern __shared__ bf16x8 s6b[];
    auto S6B = [&](int buf, int pq, int plane) { return s6b + ((size_t)((buf * 2 + pq) * 3 + plane) * 2) * BT; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 2, wq = wave & 3, l31 = lane & 31, h = lane >> 5;
    const int64_t ntp = a.Ppad / BT, bid = blockIdx.x;
    const int64_t p0 = (bid % ntp) * BT, q0 = (bid / ntp) * BT;
    if (a.q_limit && q0 >= *a.q_limit) return;

    const float4 *__restrict__ P = reinterpret_cast<const float4 *>(a.P);
    const float4 *__restrict__ Q = reinterpret_cast<const float4 *>(a.Q);
    const int seg_kq = a.seg_k / 4, nstage = a.K / 16;
    // staging: a thread's item of a stage is (octet g0 = tid >> 8, column c = tid & 255): K4 quads 2 g0 and 2 g0 + 1 of the stage
    const int c = tid & 255, g0 = tid >> 8;
    const int64_t qrow = a.q_tile_major ? 128 : a.ldq;
    const float4 *Pn = P + p0;
    const float4 *Qn = a.q_tile_major ? Q + (q0 >> 7) * (int64_t)a.q_slab_rows * 128 : Q + q0;
    int kin_n = 0;
    const unsigned boffP = (unsigned)((2 * g0 * a.ldp + c) * 16);
    const unsigned boffQ = a.q_tile_major ? (unsigned)((((int64_t)(c >> 7) * a.q_slab_rows + 2 * g0) * 128 + (c & 127)) * 16)
                                          : (unsigned)((2 * g0 * qrow + c) * 16);
    // Software pipeline, two stages deep: the operands of stage st+2 are requested at the top of stage st; those of stage st+1 -- in
    // registers since the previous stage -- are split into planes and written to the other LDS buffer BETWEEN the MFMAs of stage st,
    // a pair of elements (or three ds_write_b128) behind every fourth MFMA pair, pinned with scheduling barriers.  (The first build split
    // and stored behind the stage's last MFMA: ~110 vector instructions + 6 LDS writes per wave in front of every barrier, with both
    // waves of a SIMD arriving there together; a timing build without staging ran the frequency projection in 18.7 ms instead of 25.8.)
    float4 rpA[2], rqA[2], rpB[2], rqB[2];
    int next_k = 0;                 // stage the running pointers address
#define B6B_GLOAD(RP, RQ)                                                                                       \
    {                                                                                                           \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                         \
            RP[e] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Pn + e * a.ldp) + boffP);   \
            RQ[e] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Qn + e * qrow) + boffQ);    \
        }                                                                                                       \
        if (++next_k < nstage) {      /* the pointers never pass the last stage: the requests behind it re-read it (dropped) */ \
            Pn += KQS * a.ldp;                                                                                  \
            kin_n += KQS;                                                                                       \
            if (kin_n == seg_kq) { kin_n = 0; Qn += a.seg_col - (int64_t)(seg_kq - KQS) * qrow; }               \
            else Qn += KQS * qrow;                                                                              \
        }                                                                                                       \
    }
#define B6B_SB() __builtin_amdgcn_sched_barrier(0);
    // elements 2e, 2e+1 of an octet held as two float4 -> the three planes
#define B6B_PAIR(R, e, HI, MID, LO)                                                                             \
    {                                                                                                           \
        const float x0 = f4c(R[(e) >> 1], (2 * (e)) & 3), x1 = f4c(R[(e) >> 1], (2 * (e) + 1) & 3);             \
        const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;                                                          \
        const float r0 = x0 - (float)h0, r1 = x1 - (float)h1;                                                   \
        const __bf16 m0 = (__bf16)r0, m1 = (__bf16)r1;                                                          \
        HI[2 * (e)] = h0; HI[2 * (e) + 1] = h1; MID[2 * (e)] = m0; MID[2 * (e) + 1] = m1;                       \
        LO[2 * (e)] = (__bf16)(r0 - (float)m0); LO[2 * (e) + 1] = (__bf16)(r1 - (float)m1);                     \
    }
#define B6B_WRITE(buf, pq, HI, MID, LO) { S6B(buf, pq, 0)[g0 * BT + c] = HI; S6B(buf, pq, 1)[g0 * BT + c] = MID; S6B(buf, pq, 2)[g0 * BT + c] = LO; }
#define B6B_MF(t, i)                                                                                            \
    {                                                                                                           \
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PA[t]][i], bq[PB[t]][0], acc[i][0], 0, 0, 0);    \
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PA[t]][i], bq[PB[t]][1], acc[i][1], 0, 0, 0);    \
    }
#define B6B_GROUP(t) { B6B_MF(t, 0) B6B_MF(t, 1) B6B_MF(t, 2) B6B_MF(t, 3) }
#define B6B_APLANE(buf, pl) { _Pragma("unroll") for (int i = 0; i < 4; ++i) ap[pl][i] = S6B(buf, 0, pl)[h * BT + wp * 128 + i * 32 + l31]; }
#define B6B_BPLANE(buf, pl) { _Pragma("unroll") for (int j = 0; j < 2; ++j) bq[pl][j] = S6B(buf, 1, pl)[h * BT + wq * 64 + j * 32 + l31]; }
    // one stage on LDS buffer `buf`: RC* hold stage st+1 (split and written to buf ^ 1 here), RN* receive stage st+2; product groups
    // 0..5 = hi*lo, mid*mid, lo*hi, hi*mid, mid*hi, hi*hi (PA / PB), each 4 x 2 MFMAs
#define B6B_STAGE(st, buf, RCP, RCQ, RNP, RNQ)                                                                  \
    {                                                                                                           \
        bf16x8 ap[3][4], bq[3][2], shi, smid, slo;                                                              \
        B6B_SB()                                                                                                \
        B6B_GLOAD(RNP, RNQ)       /* UNconditional: a request under a branch makes every later vmcnt wait a full drain */ \
        B6B_APLANE(buf, 0) B6B_BPLANE(buf, 2) B6B_APLANE(buf, 1) B6B_BPLANE(buf, 1)                             \
        B6B_SB() B6B_GROUP(0) B6B_SB()                                                                          \
        B6B_APLANE(buf, 2) B6B_BPLANE(buf, 0) B6B_PAIR(RCP, 0, shi, smid, slo)                                  \
        B6B_SB() B6B_MF(1, 0) B6B_MF(1, 1) B6B_SB() B6B_PAIR(RCP, 1, shi, smid, slo)                            \
        B6B_SB() B6B_MF(1, 2) B6B_MF(1, 3) B6B_SB() B6B_PAIR(RCP, 2, shi, smid, slo)                            \
        B6B_SB() B6B_MF(2, 0) B6B_MF(2, 1) B6B_SB() B6B_PAIR(RCP, 3, shi, smid, slo)                            \
        B6B_SB() B6B_MF(2, 2) B6B_MF(2, 3) B6B_SB() B6B_WRITE((buf) ^ 1, 0, shi, smid, slo)                     \
        B6B_SB() B6B_MF(3, 0) B6B_MF(3, 1) B6B_SB() B6B_PAIR(RCQ, 0, shi, smid, slo)                            \
        B6B_SB() B6B_MF(3, 2) B6B_MF(3, 3) B6B_SB() B6B_PAIR(RCQ, 1, shi, smid, slo)                            \
        B6B_SB() B6B_MF(4, 0) B6B_MF(4, 1) B6B_SB() B6B_PAIR(RCQ, 2, shi, smid, slo)                            \
        B6B_SB() B6B_MF(4, 2) B6B_MF(4, 3) B6B_SB() B6B_PAIR(RCQ, 3, shi, smid, slo)                            \
        B6B_SB() B6B_MF(5, 0) B6B_MF(5, 1) B6B_SB() B6B_WRITE((buf) ^ 1, 1, shi, smid, slo)                     \
        B6B_SB() B6B_MF(5, 2) B6B_MF(5, 3) B6B_SB()                                                             \
        __syncthreads();                                                                                        \
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};      // smallest partial products first
    B6B_GLOAD(rpA, rqA)
    {
        bf16x8 hi, mid, lo;
        split8x3(rpA[0], rpA[1], hi, mid, lo); B6B_WRITE(0, 0, hi, mid, lo)
        split8x3(rqA[0], rqA[1], hi, mid, lo); B6B_WRITE(0, 1, hi, mid, lo)
    }
    B6B_GLOAD(rpA, rqA)
    __syncthreads();
#ifdef SDFA_GEMM6_NOSTAGE   /* timing experiment only: the MFMA + LDS-read loop without the operand stream (wrong results) */
#undef B6B_GLOAD
#undef B6B_PAIR
#undef B6B_WRITE
#define B6B_GLOAD(RP, RQ)
#define B6B_PAIR(R, e, HI, MID, LO)
#define B6B_WRITE(buf, pq, HI, MID, LO)
#endif
    // (behind the last stage the registers hold stale operands: they are split and written to the buffer nobody reads any more)
    for (int st = 0; st < nstage; st += 2) {
        B6B_STAGE(st, 0, rpA, rqA, rpB, rqB)
        if (st + 1 < nstage) B6B_STAGE(st + 1, 1, rpB, rqB, rpA, rqA)
    }
#undef B6B_GLOAD
#undef B6B_SB
#undef B6B_PAIR
#undef B6B_WRITE
#undef B6B_MF
#undef B6B_GROUP
#undef B6B_APLANE
#undef B6B_BPLANE
#undef B6B_STAGE
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            store_tile<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>(a, acc[i][j], p0 + wp * 128 + i * 32, q0 + wq * 64 + j * 32 + l31, h);
}

template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND>
hipError_t launch_bf16x6_big(const GemmArgs &a, hipStream_t s) {
    const size_t lds = (size_t)2 * 2 * 3 * 2 * 256 * sizeof(bf16x8);   // 96 KiB
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16x6_big_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int64_t nblk = (a.Ppad / 256) * (a.Qpad / 256);
    hipLaunchKernelGGL((gemm_bf16x6_big_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>), dim3((unsigned)nblk), dim3(512), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves (2 x 4, each 128 x 64 = 4 x 2 MFMA tiles), 128 KiB LDS, one workgroup per CU.
// Half the staged bytes per FLOP of the 128 x 128 tile: the 128-tile kernel's time did not move when its MFMA work
// was cut 5x (split-bf16 experiment) -- it is bound by the global->LDS staging stream -- so the big GEMMs
// (frequency projection, BiLSTM input projections) use this shape whenever P and Q are multiples of 256.
// ------------------------------------------------------------------------------------------------
template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(GemmArgs a) {
    extern __shared__ float4 sBig[];                       // [2 bufs][P | Q][KQ][256]
    constexpr int BT = 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 2, wq = wave & 3, l31 = lane & 31, h = lane >> 5;
    const int64_t ntp = a.Ppad / BT, bid = blockIdx.x;
    const int64_t p0 = (bid % ntp) * BT, q0 = (bid / ntp) * BT;
    if (a.q_limit && q0 >= *a.q_limit) return;

    const float4 *__restrict__ P = reinterpret_cast<const float4 *>(a.P);
    const float4 *__restrict__ Q = reinterpret_cast<const float4 *>(a.Q);
    const int seg_kq = a.seg_k / 4, nstage = a.K / 32;
    auto SP = [&](int buf) { return sBig + (size_t)buf * 2 * KQ * BT; };
    auto SQ = [&](int buf) { return sBig + (size_t)buf * 2 * KQ * BT + KQ * BT; };

    // staging loads: uniform running bases + one loop-invariant byte offset per thread (see gemm_k4_kernel); item i of a
    // thread is k-quad 2i + (tid >> 8), column tid & 255
    const int c = tid & 255, kq0 = tid >> 8;
    const int64_t qrow = a.q_tile_major ? 128 : a.ldq;
    const float4 *Pn = P + p0;
    const float4 *Qn = a.q_tile_major ? Q + (q0 >> 7) * (int64_t)a.q_slab_rows * 128 : Q + q0;
    int kin_n = 0;
    const unsigned boffP = (unsigned)((kq0 * a.ldp + c) * 16);
    const unsigned boffQ = a.q_tile_major ? (unsigned)((((int64_t)(c >> 7) * a.q_slab_rows + kq0) * 128 + (c & 127)) * 16)
                                          : (unsigned)((kq0 * qrow + c) * 16);
    // register staging two tiles ahead in two alternating sets (A: ra/rb, B: rc/rd), as in gemm_k4_kernel
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, rc0, rc1, rc2, rc3, rd0, rd1, rd2, rd3;
#define BG_GLOAD1(i, RP, RQ)                                                                            \
    {                                                                                                   \
        RP = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Pn + (i) * 2 * a.ldp) + boffP); \
        RQ = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Qn + (i) * 2 * qrow) + boffQ);  \
    }
#define BG_ADVANCE()                                                                                    \
    {                                                                                                   \
        Pn += KQ * a.ldp;                                                                               \
        kin_n += KQ;                                                                                    \
        if (kin_n == seg_kq) { kin_n = 0; Qn += a.seg_col - (int64_t)(seg_kq - KQ) * qrow; }            \
        else Qn += KQ * qrow;                                                                           \
    }
#define BG_GLOAD_A() BG_GLOAD1(0, ra0, rb0) BG_GLOAD1(1, ra1, rb1) BG_GLOAD1(2, ra2, rb2) BG_GLOAD1(3, ra3, rb3) BG_ADVANCE()
#define BG_GLOAD_B() BG_GLOAD1(0, rc0, rd0) BG_GLOAD1(1, rc1, rd1) BG_GLOAD1(2, rc2, rd2) BG_GLOAD1(3, rc3, rd3) BG_ADVANCE()
#define BG_LSTORE1(buf, i, RP, RQ)                                     \
    {                                                                  \
        const int idx = (i)*512 + tid;                                 \
        SP(buf)[idx] = RP;                                             \
        SQ(buf)[idx] = RQ;                                             \
    }
#define BG_LSTORE_A(buf) BG_LSTORE1(buf, 0, ra0, rb0) BG_LSTORE1(buf, 1, ra1, rb1) BG_LSTORE1(buf, 2, ra2, rb2) BG_LSTORE1(buf, 3, ra3, rb3)
#define BG_LSTORE_B(buf) BG_LSTORE1(buf, 0, rc0, rd0) BG_LSTORE1(buf, 1, rc1, rd1) BG_LSTORE1(buf, 2, rc2, rd2) BG_LSTORE1(buf, 3, rc3, rd3)
    // operand reads ahead of the MFMAs (see gemm_k4_kernel).  This kernel sits at 240 registers, so only HALF of a k-block's
    // operands (fb and fa[0..1]: what the first MFMAs of the block read) are requested one k-block ahead in a second set;
    // fa[2..3] are requested at the top of their own block and arrive behind its first four MFMAs (256 cycles)
#define BG_RD_HEAD(kb, FA, FB)                                                                          \
    {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) FA[i] = sp[(2 * (kb) + h) * BT + wp * 128 + i * 32 + l31]; \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) FB[j] = sq[(2 * (kb) + h) * BT + wq * 64 + j * 32 + l31];  \
    }
#define BG_RD_TAIL(kb, FA)                                                                              \
    _Pragma("unroll") for (int i = 2; i < 4; ++i) FA[i] = sp[(2 * (kb) + h) * BT + wp * 128 + i * 32 + l31];
#define BG_BLOCK(FA, FB, FT)                                                                            \
    {                                                                                                   \
        const float4 fa_[4] = {FA[0], FA[1], FT[2], FT[3]};                                             \
        mfma_block<4, 2>(acc, fa_, FB);                                                                 \
    }
#define BG_COMPUTE(buf)                                                                                 \
    {                                                                                                   \
        const float4 *sp = SP(buf), *sq = SQ(buf);                                                      \
        float4 fa0[2], fb0[2], fa1[2], fb1[2], ft[4];                                                   \
        BG_RD_HEAD(0, fa0, fb0)                                                                         \
        BG_RD_TAIL(0, ft)                                                                               \
        BG_RD_HEAD(1, fa1, fb1)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        BG_BLOCK(fa0, fb0, ft)                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        BG_RD_TAIL(1, ft)                                                                               \
        BG_RD_HEAD(2, fa0, fb0)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        BG_BLOCK(fa1, fb1, ft)                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        BG_RD_TAIL(2, ft)                                                                               \
        BG_RD_HEAD(3, fa1, fb1)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        BG_BLOCK(fa0, fb0, ft)                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        BG_RD_TAIL(3, ft)                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        BG_BLOCK(fa1, fb1, ft)                                                                          \
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    BG_GLOAD_A()
    BG_LSTORE_A(0)
    if (nstage > 1) { BG_GLOAD_A() }
    __syncthreads();
    int st = 0;
    for (; st + 1 < nstage; st += 2) {
        if (st + 2 < nstage) { BG_GLOAD_B() }     // even stage: LDS buffer 0 = tile st, set A = tile st+1
        BG_COMPUTE(0)
        BG_LSTORE_A(1)
        __syncthreads();
        if (st + 3 < nstage) { BG_GLOAD_A() }     // odd stage: LDS buffer 1 = tile st+1, set B = tile st+2
        BG_COMPUTE(1)
        if (st + 2 < nstage) { BG_LSTORE_B(0) }
        __syncthreads();
    }
    if (st < nstage) { BG_COMPUTE(0) }            // odd stage count: the last tile is already in LDS buffer 0
#undef BG_GLOAD1
#undef BG_ADVANCE
#undef BG_COMPUTE
#undef BG_RD_HEAD
#undef BG_RD_TAIL
#undef BG_BLOCK

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            store_tile<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>(a, acc[i][j], p0 + wp * 128 + i * 32, q0 + wq * 64 + j * 32 + l31, h);
}

template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND>
hipError_t launch_big(const GemmArgs &a, hipStream_t s) {
    const size_t lds = 2 * 2 * KQ * 256 * sizeof(float4);   // 128 KiB
    {   // per launch: cheap, and correct for every device / thread the library is used from
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_big_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int64_t nblk = (a.Ppad / 256) * (a.Qpad / 256);
    hipLaunchKernelGGL((gemm_big_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>), dim3((unsigned)nblk), dim3(512), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// "Fat wave" variant (gemm_variant 8): 256 x 256 tile, FOUR waves -- one per SIMD -- each a 128 x 128 output block
// (4 x 4 MFMA tiles = 256 accumulator registers, in AGPRs), operands staged HBM/L2 -> LDS by LDS-DMA.
// On this chip an MFMA in flight and every other vector / memory instruction of the SIMD exclude each other
// (tools/mfma_cowave.hip, mfma_shadow.hip), so a GEMM's efficiency is set by how few non-MFMA
// instructions it issues per MFMA, and a second wave per SIMD only fills stalls.  Per 32-deep stage a wave issues here
// 256 MFMAs, 32 ds_read_b128 (1 per 8 MFMAs; 128 x 64 blocks: 1 per 5.3, 64 x 64: 1 per 4), 16 LDS-DMA requests, no
// ds_write, no staging registers, one barrier.
// ------------------------------------------------------------------------------------------------
template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND>
__global__ __launch_bounds__(256, 1) void gemm_fat_kernel(GemmArgs a) {
    constexpr int BT = 256;
    extern __shared__ float4 sFat[];   // [2 bufs][P | Q][KQ k-quad rows][256]: 128 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: LDS-DMA destinations (M0) and global bases stay in SGPRs
    const int wp = wave >> 1, wq = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int64_t ntp = a.Ppad / BT;
    const int64_t q_end = a.q_limit ? std::min<int64_t>(*a.q_limit, a.Qpad) : a.Qpad;          // column sharing: tiles at or past the device-side limit are skipped

    const float4 *__restrict__ P = reinterpret_cast<const float4 *>(a.P);
    const float4 *__restrict__ Q = reinterpret_cast<const float4 *>(a.Q);
    const int nstage = a.K / 32;                                    // even (launcher): buffer parity and the operand-set rotation carry over tile ends
    auto SP = [&](int buf) { return sFat + (size_t)buf * 2 * KQ * BT; };
    auto SQ = [&](int buf) { return sFat + (size_t)buf * 2 * KQ * BT + KQ * BT; };

    // PERSISTENT: the grid is one workgroup per CU and workgroup b multiplies tiles b, b + grid, b + 2 grid, ... (tile t =
    // (column block t / ntp, row block t % ntp): the row blocks of one column block run on neighbouring CUs at the same
    // time).  The stages of consecutive tiles form ONE pipeline: the first tile's operands of the next tile are requested
    // during the last stage of this one and read behind its last barrier, so a tile's start-up (an HBM round trip) and
    // the re-dispatch of a workgroup never leave the matrix pipe empty; only the epilogue does.
    const int64_t qrow = a.q_tile_major ? 128 : a.ldq;
    const unsigned voff = (unsigned)lane * 16;
    // LDS-DMA: one wave instruction moves 64 consecutive columns of one k-quad row (1 KiB); wave w moves column quarter w
    // of all 8 rows of P and of Q.  Uniform running bases (scalar registers) + one loop-invariant byte offset per lane.
    // tiles as (row block tp, column block tq), stepped by the grid size without a division per tile
    struct Tile { int tp, tq; };
    const int g_tp = (int)(gridDim.x % ntp), g_tq = (int)(gridDim.x / ntp), ntp_i = (int)ntp;
    auto advance = [&](Tile &x) { x.tp += g_tp; x.tq += g_tq; if (x.tp >= ntp_i) { x.tp -= ntp_i; ++x.tq; } };
    // LDS-DMA through BUFFER descriptors (buffer_load_dwordx4 ... lds): descriptor and row offset in scalar registers, one
    // loop-invariant 32-bit lane offset -- a request costs no vector-ALU instruction (the global_load_lds form needed a 64-bit
    // add per request).  P: one descriptor over the whole matrix, the tile as scalar offset.  Q can exceed 4 GB: a descriptor
    // per tile, based at the tile's first element (the launcher checks that a tile's K extent stays below 2 GB).
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(P), 0, 0x7fffffff, 0x00020000);
    auto p_off = [&](const Tile &x) { return (unsigned)(((int64_t)x.tp * BT + wave * 64) * 16); };
    auto q_rsrc = [&](const Tile &x) {
        const int64_t q0 = (int64_t)x.tq * BT;
        const float4 *b = a.q_tile_major ? Q + ((q0 >> 7) + (wave >> 1)) * (int64_t)a.q_slab_rows * 128 + (wave & 1) * 64 : Q + q0 + wave * 64;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(b), 0, 0x7fffffff, 0x00020000);
    };
    auto live = [&](const Tile &x) { return (int64_t)x.tq * BT < q_end; };      // q_end <= Qpad; tq grows monotonically

    Tile t{(int)(blockIdx.x % ntp), (int)(blockIdx.x / ntp)};      // tile being multiplied
    if (!live(t)) return;
    Tile tf = t;                       // tile being fetched (one stage ahead of the multiplication)
    unsigned po = p_off(tf), qo = 0;   // running byte offsets of the next stage to request
    __amdgpu_buffer_rsrc_t qrs = q_rsrc(tf);
    const unsigned p_row = (unsigned)(a.ldp * 16), q_row = (unsigned)(qrow * 16);
    int fetch_left = nstage;           // stages of tile tf not yet requested
    bool fetching = true;
#define FAT_DMA_ROWS(buf, r0, r1)                                                                                \
    _Pragma("unroll") for (int r = (r0); r < (r1); ++r) {                                                        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (void __attribute__((address_space(3))) *)(SP(buf) + r * BT + wave * 64), 16, voff, po + r * p_row, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (void __attribute__((address_space(3))) *)(SQ(buf) + r * BT + wave * 64), 16, voff, qo + r * q_row, 0, 0); \
    }
    // after a stage's requests: advance to the next stage of the tile being fetched, or to the next tile of this workgroup
#define FAT_DMA_NEXT()                                                                                           \
    {                                                                                                            \
        if (--fetch_left > 0) { po += KQ * p_row; qo += KQ * q_row; }                                            \
        else {                                                                                                   \
            advance(tf);                                                                                         \
            fetching = live(tf);                                                                                 \
            if (fetching) { po = p_off(tf); qo = 0; qrs = q_rsrc(tf); fetch_left = nstage; }                     \
        }                                                                                                        \
    }
#define FAT_READ(sp, sq, kb, FA, FB)                                                                             \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) FA[i] = sp[(2 * (kb) + h) * BT + wp * 128 + i * 32 + l31]; \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) FB[j] = sq[(2 * (kb) + h) * BT + wq * 128 + j * 32 + l31]; \
    }
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Stage pipeline (one wave per SIMD: nothing else fills a stall, so none may be left).  Five rotating operand sets.  A
    // stage = k-blocks 0..2, barrier, k-block 3; the LDS reads of the NEXT stage (32 ds_read_b128) are spread between the
    // 64 MFMAs of k-block 3 (8 per 16 MFMAs: a burst of 32 x 4 waves fills the CU's LDS queue and the in-order wave sits
    // behind it), the 16 LDS-DMA requests of the stage after that between the MFMAs of k-block 0 -- LDS latency, the HBM
    // round trip and the barrier all sit in the shadow of MFMAs.  (Every LDS read directly follows a barrier whose fence
    // has drained this wave's DMA anyway, so the compiler's conservative "LDS read after LDS-DMA" wait costs nothing.)
    float4 fa0[4], fb0[4], fa1[4], fb1[4], fa2[4], fb2[4], fa3[4], fb3[4], fa4[4], fb4[4];
    FAT_DMA_ROWS(0, 0, KQ)
    FAT_DMA_NEXT()
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    FAT_READ(SP(0), SQ(0), 0, fa0, fb0)
    FAT_READ(SP(0), SQ(0), 1, fa1, fb1)
    FAT_READ(SP(0), SQ(0), 2, fa2, fb2)
    FAT_READ(SP(0), SQ(0), 3, fa3, fb3)
#ifndef SDFA_FAT_RG
#define SDFA_FAT_RG 4
#endif
#define FAT_RG SDFA_FAT_RG
#ifndef SDFA_FAT_DG
#define SDFA_FAT_DG 1
#endif
#define FAT_DG SDFA_FAT_DG   /* the 16 DMA requests of a stage in 4 / FAT_DG groups */
#define FAT_Q(A, B, q)                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = MFMA(SDFA_OP(f4c(A[i], q)), SDFA_OP(f4c(B[j], q)), acc[i][j]);
#ifdef SDFA_STAMPS
    unsigned long long ft0 = 0, ft1 = 0, ft2 = 0, ft3 = 0, ft4 = 0, fs_k0 = 0, fs_k12 = 0, fs_bar = 0, fs_last = 0, fs_n = 0, fs_epi = 0, fs_tiles = 0;
#define FSTAMP(t) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define FSTAMP_SUM() { fs_k0 += ft1 - ft0; fs_k12 += ft2 - ft1; fs_bar += ft3 - ft2; fs_last += ft4 - ft3; ++fs_n; }
#else
#define FSTAMP(t)
#define FSTAMP_SUM()
#endif
#define FAT_STAGE(BUF, LAST_A, LAST_B, FREE_A, FREE_B)                                                           \
    {                                                                                                            \
        FSTAMP(ft0)                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                          \
            if (fetching && (q % FAT_DG) == 0) { FAT_DMA_ROWS((BUF) ^ 1, 2 * q, 2 * q + 2 * FAT_DG) }   /* the next stage, into the buffer the previous barrier released */ \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            FAT_Q(fa0, fb0, q)                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
        }                                                                                                        \
        if (fetching) FAT_DMA_NEXT()                                                                             \
        FSTAMP(ft1)                                                                                              \
        mfma_block<4, 4>(acc, fa1, fb1);                                                                         \
        mfma_block<4, 4>(acc, fa2, fb2);                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        FSTAMP(ft2)                                                                                              \
        /* this wave's LDS-DMA of the next stage must have LANDED before the barrier lets the other waves read it: an       \
           explicit vmcnt(0) (gfx9 encoding, expcnt 7 = no wait) -- the compiler's fence only covers this wave's own reads */ \
        __builtin_amdgcn_s_waitcnt(0x0070);                                                                      \
        __syncthreads();   /* every wave holds the rest of BUF in registers */                                   \
        FSTAMP(ft3)                                                                                              \
        {   /* k-block 3 with the next stage's 32 LDS reads in groups of FAT_RG between its MFMAs.  (Behind the very last  \
               stage the reads fetch stale LDS contents that nobody uses: no branch around MFMAs.) */            \
            const float4 *sp = SP((BUF) ^ 1), *sq = SQ((BUF) ^ 1);                                               \
            FAT_READ(sp, sq, 0, fa0, fb0)                                                                        \
            FAT_READ(sp, sq, 1, fa1, fb1)                                                                        \
            FAT_READ(sp, sq, 2, fa2, fb2)                                                                        \
            FAT_READ(sp, sq, 3, FREE_A, FREE_B)                                                                  \
            mfma_block<4, 4>(acc, LAST_A, LAST_B);                                                               \
            _Pragma("unroll") for (int m = 0; m < 32 / FAT_RG; ++m) {                                            \
                __builtin_amdgcn_sched_group_barrier(0x100, FAT_RG, 0);                                          \
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * FAT_RG, 0);                                      \
            }                                                                                                    \
        }                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        FSTAMP(ft4)                                                                                              \
        FSTAMP_SUM()                                                                                             \
    }
    for (;;) {
        for (int st = 0; st < nstage; st += 2) {
            FAT_STAGE(0, fa3, fb3, fa4, fb4)
            FAT_STAGE(1, fa4, fb4, fa3, fb3)
        }
        // Epilogue (vector-ALU time the matrix pipe does not get back, so as few instructions as possible): the launcher only
        // sends K4 outputs with whole 256-row blocks here, so there is no row predicate; bias quads are requested once per
        // row group (16 loads per tile, not 64); store addresses are a uniform base (scalar) + one 32-bit lane offset.
        {
            const int64_t p0 = (int64_t)t.tp * BT, q0 = (int64_t)t.tq * BT;
            char *dbase = reinterpret_cast<char *>(a.D) + (((p0 >> 2) + wp * 32) * a.ldd + q0 + wq * 128) * 16;     // uniform
            const unsigned dlane = (unsigned)((h * a.ldd + l31) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 b[4];
                if (BIAS_P) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) b[g] = ld4(a.bias + p0 + wp * 128 + i * 32 + 8 * g + 4 * h);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        if (BIAS_P) { v[0] += b[g].x; v[1] += b[g].y; v[2] += b[g].z; v[3] += b[g].w; }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (ACT == ACT_LRELU) v[e] = lrelu02(v[e]);
                            if (ACT == ACT_TANH) v[e] = tanhf_acc(v[e]);
                        }
                        *reinterpret_cast<float4 *>(dbase + ((int64_t)(i * 8 + 2 * g) * a.ldd + j * 32) * 16 + dlane) = make_float4(v[0], v[1], v[2], v[3]);
                    }
#ifndef SDFA_FAT_NOZERO   /* timing experiment: what zero-start accumulators could save at most -- the 256 register
                             writes per tile simply left out (wrong results from the second tile of a workgroup on) */
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#endif
                }
            }
        }
#ifdef SDFA_STAMPS
        { unsigned long long te; FSTAMP(te) fs_epi += te - ft4; ++fs_tiles; }
#endif
        advance(t);
        if (!live(t)) break;
    }
#ifdef SDFA_STAMPS
    if (lane == 0 && nstage <= 16) {
        atomicAdd(&g_stamp[0], fs_k0); atomicAdd(&g_stamp[1], fs_k12); atomicAdd(&g_stamp[2], fs_bar); atomicAdd(&g_stamp[3], fs_last);
        atomicAdd(&g_stamp[4], fs_n); atomicAdd(&g_stamp[5], fs_epi); atomicAdd(&g_stamp[6], fs_tiles);
    }
#endif
#undef FAT_STAGE
#undef FAT_RG
#undef FAT_DG
#undef FAT_Q
#undef FAT_DMA_ROWS
#undef FAT_DMA_NEXT
#undef FAT_READ
}

template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND>
hipError_t launch_fat(const GemmArgs &a, hipStream_t s) {
    const size_t lds = 2 * 2 * KQ * 256 * sizeof(float4);   // 128 KiB
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_fat_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int64_t cus = sdfa_cu_count();
    const int64_t ntiles = (a.Ppad / 256) * (a.Qpad / 256);
    const int64_t wgs = std::max<int64_t>(1, cus - a.reserve_cus);      // CUs left to kernels of other streams (sdfa_model_set_reserved_cus)
    hipLaunchKernelGGL((gemm_fat_kernel<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>), dim3((unsigned)std::min(ntiles, wgs)), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace

#ifdef SDFA_STAMPS
extern "C" int sdfa_debug_read_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 8) != hipSuccess) return -3;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof z) != hipSuccess) return -3; }
    return 0;
}
#endif

// "gemm_variant" option (sdfa_debug_set_option, thread-local): 0 = default choice per shape -- gemm_fat_kernel where its 256 x 256
// tiles fill the chip at least twice, else the LDS-tiled 128 x 128 kernel (256 x 256 gemm_big_kernel for the 8192-deep frequency
// projection at mid sizes); 9 = the two-workgroups-per-CU fallback (never the fat kernel: DESIGN.md section 7); 8 = fat wherever it
// fits; 5 = 256 x 256 tile wherever it fits; 6 / 2 = the 64 x 64 form of the LDS-tiled kernel wherever it applies / never (default: small
// launches, small_tile_pays); 4 = split-bf16 x3 (NOT exact fp32).  All fp32 choices are bit-identical.  The
// variants of rounds 1-2 that were never faster (register-direct 1 / 2, LDS-DMA-fed 128 x 128 tile 3, producer / consumer waves 6)
// are gone from the library; their A/B records are profiles/r02_ab_gemm.txt and DESIGN.md section 4.2.
thread_local int g_sdfa_gemm_variant = 0;

template <int OUT_MODE, int ACT, bool BIAS_P, bool BIAS_Q, bool COND>
hipError_t launch_any(const GemmArgs &a, hipStream_t s) {
    const int variant = (g_sdfa_gemm_variant == 2 || g_sdfa_gemm_variant == 6 || g_sdfa_gemm_variant == 10 || g_sdfa_gemm_variant == 11) ? 0 : g_sdfa_gemm_variant;      // (12: split-bf16 tile choice, below)   // 2 / 6 / 10 / 11 only steer the tile choice of the LDS-tiled kernel (launch)
    // split-bf16 kernels: the 256 x 256 tile wherever it divides the problem AND gives at least half the CUs a tile ("gemm_variant" 7 =
    // never, 12 = whenever it divides): a launch of 64 such tiles -- the attention query Conv1d, the MLP layers, a single clip's
    // frequency projection -- runs on a quarter of the chip for four times as long (106 us for the 1536-deep query conv of an
    // 8192-frame chunk); four times as many 128 x 128 tiles do the same arithmetic in the same order per accumulator (bit-identical)
    const bool big_fills_half = variant == 12 || (a.Ppad / 256) * (a.Qpad / 256) * 2 >= sdfa_cu_count();
    if (a.terms == 6)      // six-product split
        return (a.Ppad % 256 == 0 && a.Qpad % 256 == 0 && a.K % 16 == 0 && variant != 7 && big_fills_half) ? launch_bf16x6_big<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>(a, s)
                                                                                                           : launch_bf16x6<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>(a, s);
    if (a.terms) {   // mixed-precision modes
        const bool big = a.Ppad % 256 == 0 && a.Qpad % 256 == 0 && variant != 7 && big_fills_half;
        if (a.terms == 1) return big ? launch_bf16_big<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND, 1>(a, s) : launch_bf16<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND, 1>(a, s);
        return big ? launch_bf16_big<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND, 3>(a, s) : launch_bf16<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND, 3>(a, s);
    }
    if (variant == 4 && !a.q_tile_major) return launch_bf16<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND, 3>(a, s);
    // one 128 x 128 block per wave, persistent (gemm_fat_kernel): by default wherever its 256 x 256 tiles fill the chip at least
    // twice over -- the frequency projection (-5.6 %) and the BiLSTM input projections (-8 / -9.5 %)
    {
        const bool fits = a.Ppad % 256 == 0 && a.Qpad % 256 == 0 && a.seg_k == a.K && a.K % 64 == 0 && OUT_MODE == OUT_K4 && !BIAS_Q && !COND &&
                          a.Pstore == a.Ppad && a.ldd * 16 * 2 < (int64_t)1 << 32 &&
                          (int64_t)(a.K / 4) * a.ldp * 16 < 0x7fffffff && (int64_t)(a.K / 4) * (a.q_tile_major ? 128 : a.ldq) * 16 < 0x7fffffff;   // buffer offsets
        const int64_t ntiles = (a.Ppad / 256) * (a.Qpad / 256);
        if constexpr (OUT_MODE == OUT_K4 && !BIAS_Q && !COND) {
            if (fits && (variant == 8 || (variant == 0 && ntiles >= 512)))
                return launch_fat<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>(a, s);
        }
    }
    // 256 x 256 tile, 8 waves: for the 8192-deep frequency projection, whose operand stream (2 MB of hidden states per frame) is
    // what the 128 x 128 tile waits for (40.3 vs 41.7 ms), as long as such tiles fill the chip.  Small batches (a single 2 s /
    // 10 s clip: 64 / 160 such tiles): the 128 x 128 tile instead, four times as many workgroups
    const bool big_fills = (a.Ppad / 256) * (a.Qpad / 256) >= 256;
    if ((variant == 5 || (a.q_tile_major && ((variant == 0 && big_fills) || variant == 9))) && a.Ppad % 256 == 0 && a.Qpad % 256 == 0)
        return launch_big<OUT_MODE, ACT, BIAS_P, BIAS_Q, COND>(a, s);
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
    if (a.act == ACT_NONE) return bp ? launch_any<OUT_K4, ACT_NONE, true, false, false>(a, s)
                                     : launch_any<OUT_K4, ACT_NONE, false, false, false>(a, s);
    if (a.act == ACT_TANH && bp) return launch_any<OUT_K4, ACT_TANH, true, false, false>(a, s);
    if (a.act == ACT_LRELU && bp) return launch_any<OUT_K4, ACT_LRELU, true, false, false>(a, s);
    return hipErrorInvalidValue;
}

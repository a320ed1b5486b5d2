// Conv stack of the audio encoder (speech_anime/config/model/dgrad.py:62-64):
//   conv2d 3->32 (3,1) + LeakyReLU(0.2) + BN  -> maxpool (2,1)        [conv1_pool_kernel]
//   conv2d 32->64 (3,1) + LeakyReLU + BN      -> maxpool (2,1)
//   conv2d 64->64 (1,1) + LeakyReLU + BN                               [conv23_kernel, chained in registers]
// Reference arithmetic: saber/nn/layers/conv2d.py:6-28,64-97 ('same' zero pad along frequency,
// saber/nn/functions.py:204-249), extend.py:94-101 (activation THEN BatchNorm, eval statistics, eps 1e-3).
//
// Kernels act along frequency only, so every (frame, time-step) column is an independent
// 128-bin signal: the stack is a set of small GEMMs  D[co][col] = W[co][k] * X[k][col]  with
// 32 columns on the MFMA lanes.  Because the conv is along the row (k) axis of a [f][ci][col]
// image, the im2col matrix of output row f is just 3*ci CONTIGUOUS rows of that image.
#include "common.h"
#include "kernels.h"

namespace {

// Epilogue of a POOLED conv layer: LeakyReLU(0.2) -> eval BatchNorm -> max over the row pair (extend.py:94-101, conv2d.py:64-97).
// v -> lrelu(v + b) * s + t is monotone in v (each step is, roundings included): non-decreasing for a BN scale s >= 0, non-increasing
// for s < 0, so max(f(v0), f(v1)) = f(max(v0, v1)) resp. f(min(v0, v1)) BITWISE, and the activation can run once on the selected
// accumulator.  The select is ONE instruction: v_med3(v0, v1, c) with c = +inf picks the max, c = -inf the min; c = copysign(inf, s)
// is one v_bfi per channel value.  Six vector instructions per pooled element instead of nine.
// -DSDFA_CONV_POOL_LATE=1 (make EXP=CONV_POOL_LATE) keeps the activation-on-both-rows order for same-box A/B timing: same bits.
__device__ __forceinline__ float pool_act(float a0, float a1, float b, float s, float t) {
#if SDFA_CONV_NOEPI          /* timing experiment only (wrong results): no activation at all -- what is the whole epilogue worth? */
    return __builtin_amdgcn_fmed3f(a0, a1, __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, s) & 0x80000000u) | 0x7f800000u));
#elif SDFA_CONV_POOL_LATE
    const float v0 = lrelu02(a0 + b) * s + t;
    const float v1 = lrelu02(a1 + b) * s + t;
    return fmaxf(v0, v1);
#else
    const float c = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, s) & 0x80000000u) | 0x7f800000u);
    return lrelu02(__builtin_amdgcn_fmed3f(a0, a1, c) + b) * s + t;
#endif
}

// ---------------------------------------------------------------------------------- conv1 + pool
constexpr int C1_ROWS = 391;   // input rows f*3+c = -3 .. 387 (zero rows either side)

__global__ __launch_bounds__(256, 2) void conv1_pool_kernel(ConvArgs a) {
    __shared__ float sIn[C1_ROWS][32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * 32;
    if (a.col_limit && m0 >= *a.col_limit) return;
    const int64_t t = m0 / a.Nc, n0 = m0 % a.Nc;

    // audio_feat row (n, t) is 384 contiguous floats (f*3 + c); transpose into [k][col]
    for (int idx = tid; idx < 32 * 96; idx += 256) {
        int col = idx / 96, q = idx % 96;
        int64_t n = n0 + col;
        int64_t row = n < a.N ? n * 64 + t : -1;
        if (a.col_src) row = a.col_src[m0 + col];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row >= 0) v = ld4(a.audio_feat + (row * 384 + 4 * q));
        sIn[3 + 4 * q + 0][col] = v.x;
        sIn[3 + 4 * q + 1][col] = v.y;
        sIn[3 + 4 * q + 2][col] = v.z;
        sIn[3 + 4 * q + 3][col] = v.w;
    }
    if (tid < 32 * 7) {
        int r = tid >> 5, c = tid & 31;
        sIn[r < 3 ? r : 384 + r][c] = 0.f;   // rows 0..2 and 387..390
    }
    float wa[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) wa[s] = a.w1[(s * 2 + h) * 32 + l31];
    float4 bb[4], ss[4], tt[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bb[g] = ld4(a.b1 + 8 * g + 4 * h);
        ss[g] = ld4(a.s1 + 8 * g + 4 * h);
        tt[g] = ld4(a.t1 + 8 * g + 4 * h);
    }
    __syncthreads();

    for (int pp = 0; pp < 16; ++pp) {
        const int p = wave * 16 + pp;   // pooled row; conv rows f = 2p, 2p+1
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            float b0 = sIn[(2 * p) * 3 + 2 * s + h][l31];
            float b1 = sIn[(2 * p + 1) * 3 + 2 * s + h][l31];
            acc0 = MFMA(SDFA_OP(wa[s]), SDFA_OP(b0), acc0);
            acc1 = MFMA(SDFA_OP(wa[s]), SDFA_OP(b1), acc1);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float bq[4] = {bb[g].x, bb[g].y, bb[g].z, bb[g].w};
            const float sq[4] = {ss[g].x, ss[g].y, ss[g].z, ss[g].w};
            const float tq[4] = {tt[g].x, tt[g].y, tt[g].z, tt[g].w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = pool_act(acc0[4 * g + e], acc1[4 * g + e], bq[e], sq[e], tq[e]);
            }
            st4(a.P1 + (((int64_t)(p * 8 + 2 * g + h)) * a.Mc + m0 + l31) * 4, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
}

// ------------------------------------------------------- conv2 + pool + conv3 (register-chained)
// One workgroup: 32 columns x 4 pooled output rows (one per wave).  A wave computes conv2 rows
// 2fo and 2fo+1 for all 64 channels (4 MFMA tiles, K = 96), applies LeakyReLU/BN, max-pools the
// two rows in registers, and feeds the pooled 64x32 tile straight back as the B operand of the
// 1x1 conv3 (2 tiles, K = 64).
__global__ __launch_bounds__(256, 2) void conv23_kernel(ConvArgs a) {
    __shared__ float4 sP1[80][32];   // 10 pool1 rows x 32 ci as 80 k-quads
    __shared__ float sPar[6][64];    // bias / BN scale / BN shift of conv2 and conv3 (epilogue operands: LDS latency, not L2)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int fc = blockIdx.x & 7;                       // chunk of 4 pooled rows
    const int64_t m0 = (int64_t)(blockIdx.x >> 3) * 32;
    if (a.col_limit && m0 >= *a.col_limit) return;
    const int f1lo = 8 * fc - 1;                         // first pool1 row held in LDS

    const float4 *__restrict__ P1 = reinterpret_cast<const float4 *>(a.P1);
    for (int idx = tid; idx < 80 * 32; idx += 256) {
        int qd = idx >> 5, col = idx & 31;
        int f1 = f1lo + (qd >> 3);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f1 >= 0 && f1 < 64) v = P1[(int64_t)(f1 * 8 + (qd & 7)) * a.Mc + m0 + col];
        sP1[qd][col] = v;
    }
    if (tid < 64) {
        sPar[0][tid] = a.b2[tid]; sPar[1][tid] = a.s2[tid]; sPar[2][tid] = a.t2[tid];
        sPar[3][tid] = a.b3[tid]; sPar[4][tid] = a.s3[tid]; sPar[5][tid] = a.t3[tid];
    }

    const float4 *__restrict__ W2 = reinterpret_cast<const float4 *>(a.w2) + h * 64 + l31;   // + kb * 128 (+32 for the second tile)
    const float4 *__restrict__ W3 = reinterpret_cast<const float4 *>(a.w3);
    const int fo = fc * 4 + wave;
    // conv2 weight quads run one k-block ahead of the MFMAs in two alternating register sets (L2 latency off the
    // critical path, no hand-over copies); the first request goes out before the barrier
    float4 wa[2] = {W2[0], W2[32]}, wb[2];
    __syncthreads();

    f32x16 acc[2][2];   // [out tile][conv row a/b]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll 1
    for (int kb = 0; kb < 12; kb += 2) {
        wb[0] = W2[(kb + 1) * 128]; wb[1] = W2[(kb + 1) * 128 + 32];
        __builtin_amdgcn_sched_barrier(0);
        {
            const float4 x[2] = {sP1[16 * wave + 2 * kb + h][l31], sP1[16 * wave + 8 + 2 * kb + h][l31]};
            mfma_block<2, 2>(acc, wa, x);
        }
        const int kn = kb + 2 < 12 ? kb + 2 : 0;      // branch-free: the last request is dropped
        wa[0] = W2[kn * 128]; wa[1] = W2[kn * 128 + 32];
        __builtin_amdgcn_sched_barrier(0);
        {
            const float4 x[2] = {sP1[16 * wave + 2 * (kb + 1) + h][l31], sP1[16 * wave + 8 + 2 * (kb + 1) + h][l31]};
            mfma_block<2, 2>(acc, wb, x);
        }
    }
    // LeakyReLU -> BN -> max over the row pair: pooled tile, rows = channels
    float4 wc[2] = {W3[h * 64 + l31], W3[h * 64 + 32 + l31]};      // conv3's first weight quads: requested before the pooling epilogue, land during it
    __builtin_amdgcn_sched_barrier(0);
    f32x16 p2[2];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = ot * 32 + 8 * g + 4 * h;
            float4 b = ld4(&sPar[0][ch]), s = ld4(&sPar[1][ch]), t = ld4(&sPar[2][ch]);
            const float bq[4] = {b.x, b.y, b.z, b.w}, sq[4] = {s.x, s.y, s.z, s.w}, tq[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                p2[ot][4 * g + e] = pool_act(acc[ot][0][4 * g + e], acc[ot][1][4 * g + e], bq[e], sq[e], tq[e]);
            }
        }
    // conv3 (1x1): contract over the pooled tile's ROW index straight from registers
    f32x16 acc3[2][1];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[j][0][r] = 0.f;
    {   // conv3's weight quads one step ahead of the MFMAs (round 5: requested right in front of them, each of the eight steps began
        // with an exposed L2 round trip)
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int ct = st >> 2, g = st & 3;
            float4 wn[2] = {wc[0], wc[1]};
            if (st + 1 < 8) { wn[0] = W3[(2 * (st + 1) + h) * 64 + l31]; wn[1] = W3[(2 * (st + 1) + h) * 64 + 32 + l31]; }
            __builtin_amdgcn_sched_barrier(0);
            const float4 xb[1] = {make_float4(p2[ct][4 * g], p2[ct][4 * g + 1], p2[ct][4 * g + 2], p2[ct][4 * g + 3])};
            mfma_block<2, 1>(acc3, wc, xb);
            __builtin_amdgcn_sched_barrier(0);
            wc[0] = wn[0]; wc[1] = wn[1];
        }
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = ot * 32 + 8 * g + 4 * h;
            float4 b = ld4(&sPar[3][ch]), s = ld4(&sPar[4][ch]), t = ld4(&sPar[5][ch]);
            float4 o;
            o.x = lrelu02(acc3[ot][0][4 * g + 0] + b.x) * s.x + t.x;
            o.y = lrelu02(acc3[ot][0][4 * g + 1] + b.y) * s.y + t.y;
            o.z = lrelu02(acc3[ot][0][4 * g + 2] + b.z) * s.z + t.z;
            o.w = lrelu02(acc3[ot][0][4 * g + 3] + b.w) * s.w + t.w;
            st4(a.X3 + ((int64_t)(fo * 16 + ot * 8 + 2 * g + h) * a.Mc + m0 + l31) * 4, o);
        }
}

// ------------------------------------------------- conv1 + pool + conv2 + pool + conv3 in one kernel
// conv23_kernel reads 10 pool1 rows (8 + a halo of 2) per workgroup: 1.25 x 512 KB per frame in, after conv1_pool_kernel
// wrote the same 512 KB out -- an HBM round trip of 1.1 MB per frame for 4.7 MFLOP of work.  Here the workgroup builds
// its 10 pool1 rows itself from the 22 frequency rows x 3 channels of audio_feat they depend on (66 contiguous floats
// per column; conv1 is recomputed 1.25 x), straight into the LDS image conv2 multiplies from.  Same arithmetic, same
// order per output element as the two-kernel path (which stays, for the debug taps): bitwise identical results.
__global__ __launch_bounds__(256, 2) void conv123_kernel(ConvArgs a) {
    __shared__ float4 sP1[80][32];   // 10 pool1 rows x 32 ci as 80 k-quads
#if SDFA_CONV_OCC4           /* timing experiment only (wrong results: races): the slice and the constants ALIAS pool1, so four workgroups fit a CU */
    float (*sIn)[33] = reinterpret_cast<float (*)[33]>(&sP1[40][0]);
    float (*sPar)[64] = reinterpret_cast<float (*)[64]>(&sP1[70][0]);
#else
    __shared__ float sIn[68][33];    // input rows k = (f - f_lo) * 3 + c, f_lo = 16 fc - 3; +1 column against bank conflicts
    __shared__ float sPar[6][64];
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int fc = blockIdx.x & 7;                       // chunk of 4 pooled rows
    const int64_t m0 = (int64_t)(blockIdx.x >> 3) * 32;
    if (a.col_limit && m0 >= *a.col_limit) return;
    const int64_t t = m0 / a.Nc, n0 = m0 % a.Nc;

    // ---- input slice: audio_feat row (n, t) is 384 contiguous floats (f*3 + c); this workgroup needs 66 of them
    const int k_lo = (16 * fc - 3) * 3;
    {   // thread -> (column tid >> 3, nine consecutive k starting at 9 * (tid & 7)): 8 x 9 = 72 >= 68 rows, no divisions
        const int col = tid >> 3, kb0 = (tid & 7) * 9;
        const int64_t n = n0 + col;
        int64_t row = n < a.N ? n * 64 + t : -1;
        if (a.col_src) row = a.col_src[m0 + col];
        // all nine requests first (clamped, unconditional addresses), the zero padding applied afterwards: a load behind a
        // divergent condition is compiled as branch + load + s_waitcnt vmcnt(0) + LDS store, i.e. nine SERIAL memory round
        // trips in front of every workgroup's first barrier
        const float *src = a.audio_feat + (row >= 0 ? row : 0) * 384;
        float v[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int gk = k_lo + kb0 + i;
            v[i] = src[gk < 0 ? 0 : (gk > 383 ? 383 : gk)];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int k = kb0 + i, gk = k_lo + k;
            if (k < 68) sIn[k][col] = (row >= 0 && k < 66 && gk >= 0 && gk < 384) ? v[i] : 0.f;
        }
    }
    if (tid < 64) {
        sPar[0][tid] = a.b2[tid]; sPar[1][tid] = a.s2[tid]; sPar[2][tid] = a.t2[tid];
        sPar[3][tid] = a.b3[tid]; sPar[4][tid] = a.s3[tid]; sPar[5][tid] = a.t3[tid];
    }
    const float4 *__restrict__ W2 = reinterpret_cast<const float4 *>(a.w2) + h * 64 + l31;
    const float4 *__restrict__ W3 = reinterpret_cast<const float4 *>(a.w3);
    float4 wa[2] = {W2[0], W2[32]}, wb[2];
    // conv1's operands and epilogue constants are requested BEFORE the barrier too (round 5): behind it, every workgroup paid an
    // exposed L2 round trip in front of its first MFMA
    float w1[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) w1[s] = a.w1[(s * 2 + h) * 32 + l31];
    float4 e1b[4], e1s[4], e1t[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { e1b[g] = ld4(a.b1 + 8 * g + 4 * h); e1s[g] = ld4(a.s1 + 8 * g + 4 * h); e1t[g] = ld4(a.t1 + 8 * g + 4 * h); }
    __syncthreads();

    // ---- conv1 + LeakyReLU + BN + pool: pool1 row j of the slice (global row 8 fc - 1 + j), rows j = wave, wave+4, wave+8
    {
        for (int j = wave; j < 10; j += 4) {
            const int f1 = 8 * fc - 1 + j;
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const float b0 = sIn[6 * j + 2 * s + h][l31];
                const float b1 = sIn[6 * j + 3 + 2 * s + h][l31];
                acc0 = MFMA(SDFA_OP(w1[s]), SDFA_OP(b0), acc0);
                acc1 = MFMA(SDFA_OP(w1[s]), SDFA_OP(b1), acc1);
            }
            const bool valid = f1 >= 0 && f1 < 64;       // rows -1 and 64 are conv2's zero padding
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = e1b[g], sc = e1s[g], sh = e1t[g];
                const float bq[4] = {b.x, b.y, b.z, b.w}, sq[4] = {sc.x, sc.y, sc.z, sc.w}, tq[4] = {sh.x, sh.y, sh.z, sh.w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = pool_act(acc0[4 * g + e], acc1[4 * g + e], bq[e], sq[e], tq[e]);
                    o[e] = valid ? v : 0.f;
                }
                sP1[j * 8 + 2 * g + h][l31] = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    __syncthreads();

    // ---- conv2 + pool + conv3: identical to conv23_kernel from here on
    const int fo = fc * 4 + wave;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
    for (int kb = 0; kb < 12; kb += 2) {
        wb[0] = W2[(kb + 1) * 128]; wb[1] = W2[(kb + 1) * 128 + 32];
        __builtin_amdgcn_sched_barrier(0);
        {
            const float4 x[2] = {sP1[16 * wave + 2 * kb + h][l31], sP1[16 * wave + 8 + 2 * kb + h][l31]};
            mfma_block<2, 2>(acc, wa, x);
        }
        const int kn = kb + 2 < 12 ? kb + 2 : 0;
        wa[0] = W2[kn * 128]; wa[1] = W2[kn * 128 + 32];
        __builtin_amdgcn_sched_barrier(0);
        {
            const float4 x[2] = {sP1[16 * wave + 2 * (kb + 1) + h][l31], sP1[16 * wave + 8 + 2 * (kb + 1) + h][l31]};
            mfma_block<2, 2>(acc, wb, x);
        }
    }
    float4 wc[2] = {W3[h * 64 + l31], W3[h * 64 + 32 + l31]};      // conv3's first weight quads: requested before the pooling epilogue, land during it
    __builtin_amdgcn_sched_barrier(0);
    f32x16 p2[2];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = ot * 32 + 8 * g + 4 * h;
            float4 b = ld4(&sPar[0][ch]), sc = ld4(&sPar[1][ch]), sh = ld4(&sPar[2][ch]);
            const float bq[4] = {b.x, b.y, b.z, b.w}, sq[4] = {sc.x, sc.y, sc.z, sc.w}, tq[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                p2[ot][4 * g + e] = pool_act(acc[ot][0][4 * g + e], acc[ot][1][4 * g + e], bq[e], sq[e], tq[e]);
            }
        }
    f32x16 acc3[2][1];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[j][0][r] = 0.f;
    {   // conv3's weight quads one step ahead of the MFMAs (round 5: requested right in front of them, each of the eight steps began
        // with an exposed L2 round trip)
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int ct = st >> 2, g = st & 3;
            float4 wn[2] = {wc[0], wc[1]};
            if (st + 1 < 8) { wn[0] = W3[(2 * (st + 1) + h) * 64 + l31]; wn[1] = W3[(2 * (st + 1) + h) * 64 + 32 + l31]; }
            __builtin_amdgcn_sched_barrier(0);
            const float4 xb[1] = {make_float4(p2[ct][4 * g], p2[ct][4 * g + 1], p2[ct][4 * g + 2], p2[ct][4 * g + 3])};
            mfma_block<2, 1>(acc3, wc, xb);
            __builtin_amdgcn_sched_barrier(0);
            wc[0] = wn[0]; wc[1] = wn[1];
        }
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = ot * 32 + 8 * g + 4 * h;
            float4 b = ld4(&sPar[3][ch]), sc = ld4(&sPar[4][ch]), sh = ld4(&sPar[5][ch]);
            float4 o;
            o.x = lrelu02(acc3[ot][0][4 * g + 0] + b.x) * sc.x + sh.x;
            o.y = lrelu02(acc3[ot][0][4 * g + 1] + b.y) * sc.y + sh.y;
            o.z = lrelu02(acc3[ot][0][4 * g + 2] + b.z) * sc.z + sh.z;
            o.w = lrelu02(acc3[ot][0][4 * g + 3] + b.w) * sc.w + sh.w;
            st4(a.X3 + ((int64_t)(fo * 16 + ot * 8 + 2 * g + h) * a.Mc + m0 + l31) * 4, o);
        }
}

// ------------------------------------------------- the fused conv stack on bf16 MFMA (mixed-precision modes, round 4)
// conv123_kernel with every contraction on v_mfma_f32_32x32x16_bf16 (fp32 accumulation, fp32 epilogues): TERMS 1 = operands rounded
// to bf16, 3 = split-bf16 (hi + lo, three products), 6 = three bf16 terms, six products -- the modes of the other body kernels.
// An MFMA contracts 16 k: conv1 (K = 9) is ONE instruction per output row and product, conv2 (K = 96) six, conv3 (K = 64) four --
// 37 + 24 + 4 per wave instead of 281 fp32 MFMAs of twice the cycles.  The K axes are ordered so that every lane builds its B
// operand from values it already owns (the scheme of the LSTM kernels): an accumulator lane holds channels 8g + 4h + e (g, e = 0..3),
// so k-step "octet pair q" of lane half h is channels {8(2q) + 4h + e} u {8(2q+1) + 4h + e} -- pool1 goes to LDS as such octets,
// plane by plane, and pool2 never leaves the registers.  The host packs the weights in the same order (api.cpp: pack_conv_bf16).
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
#define CMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

template <int NPL>
__device__ __forceinline__ void conv_split8(const float (&x)[8], cbf16x8 (&pl)[3]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hb = (__bf16)x[e];
        pl[0][e] = hb;
        if (NPL > 1) {
            const float r1 = x[e] - (float)hb;
            const __bf16 mb = (__bf16)r1;
            pl[1][e] = mb;
            if (NPL > 2) pl[2][e] = (__bf16)(r1 - (float)mb);
        }
    }
}

// acc += W * x over the partial products of the mode, smallest first (plane 0 = hi, 1 = mid / lo, 2 = lo)
template <int TERMS>
__device__ __forceinline__ void conv_products(f32x16 &acc, const cbf16x8 (&w)[3], const cbf16x8 (&x)[3]) {
    if (TERMS == 6) {
        acc = CMFMA(w[0], x[2], acc); acc = CMFMA(w[1], x[1], acc); acc = CMFMA(w[2], x[0], acc);
        acc = CMFMA(w[0], x[1], acc); acc = CMFMA(w[1], x[0], acc);
    } else if (TERMS == 3) {
        acc = CMFMA(w[1], x[0], acc); acc = CMFMA(w[0], x[1], acc);
    }
    acc = CMFMA(w[0], x[0], acc);
}

template <int TERMS>
__global__ __launch_bounds__(256, 2) void conv123_bf16_kernel(ConvArgs a) {
    constexpr int NPL = TERMS == 6 ? 3 : (TERMS == 3 ? 2 : 1);
    __shared__ cbf16x8 sP1b[NPL][10][4][32];   // pool1: [plane][row j][octet 2q + h][column]
    __shared__ float sIn[68][33];
    __shared__ float sPar[6][64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int fc = blockIdx.x & 7;
    const int64_t m0 = (int64_t)(blockIdx.x >> 3) * 32;
    if (a.col_limit && m0 >= *a.col_limit) return;
    const int64_t t = m0 / a.Nc, n0 = m0 % a.Nc;

    // ---- input slice, exactly as conv123_kernel
    const int k_lo = (16 * fc - 3) * 3;
    {
        const int col = tid >> 3, kb0 = (tid & 7) * 9;
        const int64_t n = n0 + col;
        int64_t row = n < a.N ? n * 64 + t : -1;
        if (a.col_src) row = a.col_src[m0 + col];
        const float *src = a.audio_feat + (row >= 0 ? row : 0) * 384;
        float v[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int gk = k_lo + kb0 + i;
            v[i] = src[gk < 0 ? 0 : (gk > 383 ? 383 : gk)];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int k = kb0 + i, gk = k_lo + k;
            if (k < 68) sIn[k][col] = (row >= 0 && k < 66 && gk >= 0 && gk < 384) ? v[i] : 0.f;
        }
    }
    if (tid < 64) {
        sPar[0][tid] = a.b2[tid]; sPar[1][tid] = a.s2[tid]; sPar[2][tid] = a.t2[tid];
        sPar[3][tid] = a.b3[tid]; sPar[4][tid] = a.s3[tid]; sPar[5][tid] = a.t3[tid];
    }
    // weights: planes of [w1: 2 halves x 32 co | w2: 6 k-steps x 2 halves x 64 co | w3: 4 k-steps x 2 halves x 64 co] octets
    constexpr int W_PLANE = 2 * 32 + 6 * 2 * 64 + 4 * 2 * 64;       // 1344 octets per plane
    const cbf16x8 *__restrict__ Wb = reinterpret_cast<const cbf16x8 *>(a.wb);
    const cbf16x8 *__restrict__ W1b = Wb + h * 32 + l31;
    const cbf16x8 *__restrict__ W2b = Wb + 64 + h * 64 + l31;
    const cbf16x8 *__restrict__ W3b = Wb + 64 + 768 + h * 64 + l31;
    cbf16x8 wa[2][3], wn[2][3];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) { wa[0][pl] = W2b[pl * W_PLANE]; wa[1][pl] = W2b[pl * W_PLANE + 32]; }      // conv2 k-step 0: lands during conv1
    // conv1's operands and epilogue constants before the barrier too (round 5, as in conv123_kernel)
    cbf16x8 w1[3];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) w1[pl] = W1b[pl * W_PLANE];
    float4 e1b[4], e1s[4], e1t[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { e1b[g] = ld4(a.b1 + 8 * g + 4 * h); e1s[g] = ld4(a.s1 + 8 * g + 4 * h); e1t[g] = ld4(a.t1 + 8 * g + 4 * h); }
    __syncthreads();

    // ---- conv1 + LeakyReLU + BN + pool -> pool1 rows j = wave, wave + 4, wave + 8 of the slice, as octets
    {
        for (int j = wave; j < 10; j += 4) {
            const int f1 = 8 * fc - 1 + j;
            float x0[8], x1[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {       // k = 8h + e: nine taps, the rest of the 16 are zeros (h = 1 reads one value)
                const bool on = h == 0 || e == 0;
                const int k0 = 6 * j + (on ? 8 * h + e : 0);
                x0[e] = on ? sIn[k0][l31] : 0.f;
                x1[e] = on ? sIn[k0 + 3][l31] : 0.f;
            }
            cbf16x8 b0[3], b1[3];
            conv_split8<NPL>(x0, b0);
            conv_split8<NPL>(x1, b1);
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            conv_products<TERMS>(acc0, w1, b0);
            conv_products<TERMS>(acc1, w1, b1);
            const bool valid = f1 >= 0 && f1 < 64;       // rows -1 and 64 are conv2's zero padding
            float o[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = e1b[g], sc = e1s[g], sh = e1t[g];
                const float bq[4] = {b.x, b.y, b.z, b.w}, sq[4] = {sc.x, sc.y, sc.z, sc.w}, tq[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = pool_act(acc0[4 * g + e], acc1[4 * g + e], bq[e], sq[e], tq[e]);
                    o[4 * g + e] = valid ? v : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float xo[8] = {o[8 * q], o[8 * q + 1], o[8 * q + 2], o[8 * q + 3], o[8 * q + 4], o[8 * q + 5], o[8 * q + 6], o[8 * q + 7]};
                cbf16x8 pp[3];
                conv_split8<NPL>(xo, pp);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) sP1b[pl][j][2 * q + h][l31] = pp[pl];
            }
        }
    }
    __syncthreads();

    // ---- conv2 (six k-steps: tap df = ks / 2, octet pair ks % 2) + pool + conv3 (four k-steps, B operand from registers)
    const int fo = fc * 4 + wave;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        if (ks + 1 < 6) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) { wn[0][pl] = W2b[pl * W_PLANE + (ks + 1) * 128]; wn[1][pl] = W2b[pl * W_PLANE + (ks + 1) * 128 + 32]; }
        }
        cbf16x8 x[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) x[i][pl] = sP1b[pl][2 * wave + i + (ks >> 1)][2 * (ks & 1) + h][l31];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int i = 0; i < 2; ++i) conv_products<TERMS>(acc[ot][i], wa[ot], x[i]);
        if (ks + 1 < 6) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) { wa[0][pl] = wn[0][pl]; wa[1][pl] = wn[1][pl]; }
        }
    }
    float p2[2][16];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = ot * 32 + 8 * g + 4 * h;
            float4 b = ld4(&sPar[0][ch]), sc = ld4(&sPar[1][ch]), sh = ld4(&sPar[2][ch]);
            const float bq[4] = {b.x, b.y, b.z, b.w}, sq[4] = {sc.x, sc.y, sc.z, sc.w}, tq[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                p2[ot][4 * g + e] = pool_act(acc[ot][0][4 * g + e], acc[ot][1][4 * g + e], bq[e], sq[e], tq[e]);
            }
        }
    f32x16 acc3[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[j][r] = 0.f;
    {   // conv3's weights one k-step ahead (round 5: requested in front of their products, every k-step began with an L2 round trip)
        cbf16x8 w3c[2][3], w3n[2][3];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) { w3c[0][pl] = W3b[pl * W_PLANE]; w3c[1][pl] = W3b[pl * W_PLANE + 32]; }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int ct = st >> 1, q = st & 1;
            if (st + 1 < 4) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) { w3n[0][pl] = W3b[pl * W_PLANE + (st + 1) * 128]; w3n[1][pl] = W3b[pl * W_PLANE + (st + 1) * 128 + 32]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            const float xo[8] = {p2[ct][8 * q], p2[ct][8 * q + 1], p2[ct][8 * q + 2], p2[ct][8 * q + 3], p2[ct][8 * q + 4], p2[ct][8 * q + 5], p2[ct][8 * q + 6], p2[ct][8 * q + 7]};
            cbf16x8 xb[3];
            conv_split8<NPL>(xo, xb);
            conv_products<TERMS>(acc3[0], w3c[0], xb);
            conv_products<TERMS>(acc3[1], w3c[1], xb);
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < 4) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) { w3c[0][pl] = w3n[0][pl]; w3c[1][pl] = w3n[1][pl]; }
            }
        }
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = ot * 32 + 8 * g + 4 * h;
            float4 b = ld4(&sPar[3][ch]), sc = ld4(&sPar[4][ch]), sh = ld4(&sPar[5][ch]);
            float4 o;
            o.x = lrelu02(acc3[ot][4 * g + 0] + b.x) * sc.x + sh.x;
            o.y = lrelu02(acc3[ot][4 * g + 1] + b.y) * sc.y + sh.y;
            o.z = lrelu02(acc3[ot][4 * g + 2] + b.z) * sc.z + sh.z;
            o.w = lrelu02(acc3[ot][4 * g + 3] + b.w) * sc.w + sh.w;
            st4(a.X3 + ((int64_t)(fo * 16 + ot * 8 + 2 * g + h) * a.Mc + m0 + l31) * 4, o);
        }
}
#undef CMFMA

}  // namespace

hipError_t sdfa_launch_conv1(const ConvArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(conv1_pool_kernel, dim3((unsigned)(a.Mc / 32)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t sdfa_launch_conv123(const ConvArgs &a, hipStream_t s) {
    if (a.terms) {      // mixed-precision modes: the same stack on bf16 MFMA
        if (!a.wb) return hipErrorInvalidValue;
        const dim3 grid((unsigned)(a.Mc / 32 * 8));
        if (a.terms == 6) hipLaunchKernelGGL(conv123_bf16_kernel<6>, grid, dim3(256), 0, s, a);
        else if (a.terms == 3) hipLaunchKernelGGL(conv123_bf16_kernel<3>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(conv123_bf16_kernel<1>, grid, dim3(256), 0, s, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(conv123_kernel, dim3((unsigned)(a.Mc / 32 * 8)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t sdfa_launch_conv23(const ConvArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(conv23_kernel, dim3((unsigned)(a.Mc / 32 * 8)), dim3(256), 0, s, a);
    return hipGetLastError();
}

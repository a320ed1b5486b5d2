// Bahdanau attention (speech_anime/layers/attentions.py:39-75,92-124) and small layout kernels.  Per frame n, over its 64 keys t:
//     score[t] = v . tanh(W_k x[:, t, n] + qp[:, n] + b)
//     align    = softmax_t(score * 1.0)                         (scale_score_at_eval = 1.0)
//     ctx      = sum_t align[t] * x[:, t, n]                    (torch.bmm(align, value))
// The query path (Conv1d over time steps 31..33, proj_qry) runs on the MFMA GEMM (gemm.hip) and gives qp.  The rest is here:
//   attn_fused_f32_kernel      exact fp32, chunks from about 3,600 frames: the WHOLE layer in one launch that reads H once -- key
//                              projection with the weights in registers, scores on the accumulators, running softmax + context
//                              folded in while the tile is in an LDS-DMA ring
//   attn_key_score_f32_kernel  the same key projection + scores for smaller chunks (time steps split over more work units; eight
//   attn_key_score_kernel<T>   partial scores per column to HBM); <T>: bf16 (T = 1) / split-bf16 (T = 3) operands for the
//                              mixed-precision modes, the tile split into bf16 planes in place
//   attn_kernel<true>          softmax + context from those partial scores: the running-softmax recurrence of the one-launch form,
//                              operation for operation (the same bits), streaming H a second time
//   attn_kernel<false>         rounds 1-5: scores from key projections a GEMM stored (option attn_unfused = 1; an independent form the
//                              tests compare against), two-pass softmax
// attn_kernel: one workgroup = 16 consecutive frames (Nc / 16 workgroups: 512 for an 8192-frame chunk, two per CU), lane = (frame,
// part p = 0..3); a stream over H (128 KiB per frame): every global access is a 256-byte run of the K4 [feature/4][t*Nc + n] arrays per
// quarter wave.  <false> only: wave w owns time steps 16w..16w+15 and all 128 units (each quarter wave sums 32 of them, two __shfl_xor
// steps combine the quarters), the softmax is wavefront-reduced once (per-wave (max, sum of exp) pairs meet in LDS), and the 64 weights
// of a frame go through LDS once for the context.
#include "common.h"
#include "kernels.h"
#include <algorithm>

namespace {

constexpr int ATT_FR = 16;   // frames per workgroup

// One key of a frame's RUNNING softmax and context (the flash-attention recurrence over the 64 keys, t = 0 .. 63):
//     m' = max(m, s);  alpha = exp(m - m');  p = exp(s - m');  l = l alpha + p;  ctx = ctx alpha + p x
// Shared, operation for operation (explicit fma, un-contracted products), by attn_fused_f32_kernel -- which folds a tile in while it is
// in LDS -- and by attn_kernel<true>, which streams H again: the two forms give the SAME bits, so a frame's rows do not depend on
// the size of the chunk it is computed in (the launcher picks the form by size).
__device__ __forceinline__ float score_of_partials(float p0, float p1, float p2, float p3, float p4, float p5, float p6, float p7) {
    return fadd_exact(fadd_exact(fadd_exact(p0, p1), fadd_exact(p2, p3)), fadd_exact(fadd_exact(p4, p5), fadd_exact(p6, p7)));
}
__device__ __forceinline__ void soft_step(float sc, float &m, float &l, float &alpha, float &pw) {
    const float mn = fmaxf(m, sc);
    alpha = __expf(m - mn);
    pw = __expf(sc - mn);
    m = mn;
    l = __builtin_fmaf(l, alpha, pw);
}
__device__ __forceinline__ void ctx_step(float4 &c, float alpha, float pw, const float4 &x) {
    c.x = __builtin_fmaf(pw, x.x, fmul_exact(c.x, alpha));
    c.y = __builtin_fmaf(pw, x.y, fmul_exact(c.y, alpha));
    c.z = __builtin_fmaf(pw, x.z, fmul_exact(c.z, alpha));
    c.w = __builtin_fmaf(pw, x.w, fmul_exact(c.w, alpha));
}

// GIVEN: the scores were written by attn_key_score_kernel (a.S, [t * Nc + n]); the key projections are never materialised and
// a.KP / a.QP are not read.
template <bool GIVEN>
__global__ __launch_bounds__(256, 2) void attn_kernel(AttnArgs a) {
    __shared__ float sAlign[64][ATT_FR];     // [t][frame]
    __shared__ float2 sStat[4][ATT_FR];      // per wave: (max, sum of exp) over its 16 time steps

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, part = lane >> 4;
    const int64_t n = (int64_t)blockIdx.x * ATT_FR + fr;
    const float4 *__restrict__ KP = reinterpret_cast<const float4 *>(a.KP);
    const float4 *__restrict__ QP = reinterpret_cast<const float4 *>(a.QP);
    const float4 *__restrict__ H = reinterpret_cast<const float4 *>(a.H);

    if constexpr (GIVEN) {
        // Scores given (eight partial sums per column): running softmax + context in ONE sweep over the 64 keys, the recurrence
        // attn_fused_f32_kernel runs while the tiles are in its ring (same helpers: same bits).  Lane (frame fr, part), wave w: feature
        // quads 32w + 8 part .. + 7; every lane of a frame carries the frame's (m, l) itself -- no LDS, no barrier.
        const float4 *__restrict__ S2 = reinterpret_cast<const float4 *>(a.S) + n * 2;
        const float4 *__restrict__ hp = H + (int64_t)(32 * wave + 8 * part) * a.Mc + n;
        float4 acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        float m_run = -3.0e38f, l_run = 0.f;
        // two register sets, the requests of time step t + 1 issued before step t is folded in (ten 16-byte requests per step and lane;
        // left to itself the compiler issued them one by one, each behind a full wait: 0.9 ms per step instead of 0.5)
        float4 pa[2], xa[8], pb[2], xb[8];
#define AT_LOAD(P, X, t_)                                                                           \
        {                                                                                           \
            P[0] = S2[(int64_t)(t_) * a.Nc * 2]; P[1] = S2[(int64_t)(t_) * a.Nc * 2 + 1];           \
            _Pragma("unroll") for (int q = 0; q < 8; ++q) X[q] = hp[(int64_t)q * a.Mc + (int64_t)(t_) * a.Nc]; \
        }
#define AT_FOLD(P, X)                                                                               \
        {                                                                                           \
            float alpha, pw;                                                                        \
            soft_step(score_of_partials(P[0].x, P[0].y, P[0].z, P[0].w, P[1].x, P[1].y, P[1].z, P[1].w), m_run, l_run, alpha, pw); \
            _Pragma("unroll") for (int q = 0; q < 8; ++q) ctx_step(acc[q], alpha, pw, X[q]);        \
        }
        AT_LOAD(pa, xa, 0)
#pragma unroll 1
        for (int t = 0; t < 64; t += 2) {
            __builtin_amdgcn_sched_barrier(0);
            AT_LOAD(pb, xb, t + 1)
            __builtin_amdgcn_sched_barrier(0);
            AT_FOLD(pa, xa)
            __builtin_amdgcn_sched_barrier(0);
            AT_LOAD(pa, xa, t + 2 < 64 ? t + 2 : t)           // (behind the last step: a dropped re-read)
            __builtin_amdgcn_sched_barrier(0);
            AT_FOLD(pb, xb)
        }
#undef AT_LOAD
#undef AT_FOLD
        const float inv = 1.0f / l_run;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int fq = 32 * wave + 8 * part + q;
            const float4 z = make_float4(acc[q].x * inv, acc[q].y * inv, acc[q].z * inv, acc[q].w * inv);
            st4(a.Zk4 + ((int64_t)fq * a.Nc + n) * 4, z);
            if (a.z_out && n < a.N) st4(a.z_out + n * 512 + fq * 4, z);
        }
        if (a.align_out && n < a.N) {              // weights of time steps 16 w + 4 part .. + 3 (each (wave, part) pair its four)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = 16 * wave + 4 * part + e;
                const float4 p0 = S2[(int64_t)t * a.Nc * 2], p1 = S2[(int64_t)t * a.Nc * 2 + 1];
                a.align_out[n * 64 + t] = __expf(score_of_partials(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w) - m_run) * inv;
            }
        }
        return;
    }
    // ---- scores: this quarter wave sums units 32p .. 32p+31 (quads 8p .. 8p+7)
    float sc[16];
    {
        float4 qb[8], vv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 qp = QP[(int64_t)(8 * part + q) * a.Nc + n];
            const float4 bb = ld4(a.b + (8 * part + q) * 4);
            qb[q] = make_float4(qp.x + bb.x, qp.y + bb.y, qp.z + bb.z, qp.w + bb.w);
            vv[q] = ld4(a.v + (8 * part + q) * 4);
        }
        const float4 *__restrict__ kp = KP + (int64_t)(8 * part) * a.Mc + (int64_t)(16 * wave) * a.Nc + n;
#pragma unroll 2
        for (int tt = 0; tt < 16; ++tt) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 k = kp[(int64_t)q * a.Mc + (int64_t)tt * a.Nc];
                s += vv[q].x * tanhf_acc(qb[q].x + k.x);
                s += vv[q].y * tanhf_acc(qb[q].y + k.y);
                s += vv[q].z * tanhf_acc(qb[q].z + k.z);
                s += vv[q].w * tanhf_acc(qb[q].w + k.w);
            }
            s += __shfl_xor(s, 16);                // (p0 + p1), (p2 + p3) -- the same bits on both sides of each pair
            s += __shfl_xor(s, 32);                // all four quarters hold the complete score of (t = 16w + tt, frame)
            sc[tt] = s;
        }
    }
    // ---- softmax over the 64 time steps of a frame: local (max, sum) per wave, combined through LDS
    float mw = sc[0];
#pragma unroll
    for (int tt = 1; tt < 16; ++tt) mw = fmaxf(mw, sc[tt]);
    float sw = 0.f;
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) { sc[tt] = __expf(sc[tt] - mw); sw += sc[tt]; }
    if (part == 0) sStat[wave][fr] = make_float2(mw, sw);
    __syncthreads();
    float M = -3.0e38f;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, sStat[w][fr].x);
    float den = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) den += sStat[w][fr].y * __expf(sStat[w][fr].x - M);
    const float scale = __expf(mw - M) / den;      // exp(s - M) = exp(s - mw) * exp(mw - M)
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) {
        const float al = sc[tt] * scale;
        if ((tt & 3) == part) {                    // the quarters hold the same values: each publishes every fourth time step
            sAlign[16 * wave + tt][fr] = al;
            if (a.align_out && n < a.N) a.align_out[n * 64 + 16 * wave + tt] = al;
        }
    }
    __syncthreads();
    // ---- context: feature quads 32w + 8p .. + 7 of this lane's frame
    float4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 *__restrict__ hp = H + (int64_t)(32 * wave + 8 * part) * a.Mc + n;
#pragma unroll 2
    for (int t = 0; t < 64; ++t) {
        const float al = sAlign[t][fr];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 x = hp[(int64_t)q * a.Mc + (int64_t)t * a.Nc];
            acc[q].x += al * x.x; acc[q].y += al * x.y; acc[q].z += al * x.z; acc[q].w += al * x.w;
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int fq = 32 * wave + 8 * part + q;
        st4(a.Zk4 + ((int64_t)fq * a.Nc + n) * 4, acc[q]);
        if (a.z_out && n < a.N) st4(a.z_out + n * 512 + fq * 4, acc[q]);
    }
}

// ------------------------------------------------------------------------------------------------
// Key projection + scores of the bf16 attention modes (BASELINE configs[3]: "bf16 attention with MFMA"), one streaming pass over H:
//     score[t, n] = v . tanh(Wk x[:, t, n] + qp[:, n] + b)            (speech_anime/layers/attentions.py:107-121)
// The projection is 128 outputs x 512 features over 64 * Nc columns: 192 FLOP per byte of H in the three-product split, under the
// chip's ~400 FLOP per HBM byte -- an HBM-bound kernel whose only large operand is H itself.  So:
//   * the WEIGHTS live in registers for the whole launch: eight waves, wave w owns outputs 16w .. 16w+15 as v_mfma_f32_16x16x32_bf16
//     A fragments, 16 k-steps x (hi, lo) = 128 VGPRs, two waves per SIMD -- one wave's MFMAs run under the other's vector work (a
//     first form with four waves of 32 outputs measured 0.62 ms of serial instruction issue per step for 0.2 ms of MFMAs);
//   * the grid is persistent (one workgroup per CU) over work units of (16 frames) x (a range of time steps); the unit's tiles --
//     16 columns x 512 features = 32 KiB of H each -- travel HBM -> LDS by LDS-DMA into a ring of five slots, three tiles in flight
//     behind the one being split (no staging registers: with 256 registers of weights the allocator had no room left for register sets
//     in flight and copied them at the loop's back edge behind a full vmcnt(0)).  The DMA requests are inline assembly, invisible to
//     the compiler's waitcnt bookkeeping (which puts a vmcnt(0) in front of any LDS read while a DMA it knows of is in flight), and
//     are waited for by hand with counted vmcnt;
//   * a landed tile is split into bf16 planes once per workgroup, IN PLACE: the thread that owns octet o of column c reads the two
//     fp32 quads (2o, 2o + 1) of that column and writes the hi octet over the first and the lo octet over the second (one
//     ds_read_b128 = one B fragment; every wave reads the same planes) -- no plane buffers, the whole LDS is ring.  The split of tile
//     i + 1 is spread between the MFMAs of tile i;
//   * the 128 x 16 key projections never leave the accumulators: + qp + b, tanh, the dot product with v and the sum over the wave's
//     32 outputs happen in registers, and each wave writes ITS partial score per column (attn_kernel adds the four in a fixed order) --
//     instead of 512 bytes of key projection per column written here and read again by attn_kernel.
// One workgroup barrier per tile.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int KS_COLS = 16;                                   // columns per tile = one MFMA column block
constexpr int KS_RING = 5;                                    // ring slots of 32 KiB: all 160 KiB of LDS
constexpr size_t ks_lds_bytes(int) { return (size_t)KS_RING * 128 * KS_COLS * 16; }

__device__ __forceinline__ void ks_split8(const float4 &x0, const float4 &x1, bf16x8 &hi, bf16x8 &lo) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hb = (__bf16)x[e];
        hi[e] = hb;
        lo[e] = (__bf16)(x[e] - (float)hb);
    }
}

#ifndef SDFA_KS_EXP
#define SDFA_KS_EXP 0      /* timing experiments only (make EXP=KS_EXP EXPVAL=n; wrong results by design): 1 no MFMAs, 2 no DMA, 3 no split, 4 no tanh */
#endif

template <int TERMS>
__global__ __launch_bounds__(512, 2) void attn_key_score_kernel(AttnKeyArgs a) {
    constexpr bool LO = TERMS > 1;
    constexpr int SLOT = 128 * KS_COLS;                       // 16-byte cells per ring slot
    extern __shared__ float4 sRing[];                         // [KS_RING][128 quads][16 columns]: fp32 tiles as the DMA writes them, then
                                                              // [64 octets][hi | lo][16 columns] bf16x8 once split in place

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    const int c = tid & 15, pr = tid >> 4;                    // splitting: column c, octets pr and pr + 32 of the tile
    const float4 *__restrict__ H4 = reinterpret_cast<const float4 *>(a.H);
    const float4 *__restrict__ QP4 = reinterpret_cast<const float4 *>(a.QP);
    const int64_t Mc = a.Mc, Nc = a.Nc;
    const int G = gridDim.x;
    const int ts_shift = a.ts_shift, TT = 64 >> ts_shift;    // a unit = 16 frames x TT time steps
    const int n_units = (int)(Nc / KS_COLS) << ts_shift;

    // ---- this wave's 16 rows of proj_key as A fragments, for the whole launch: lane (row 16w + l15, k group kg) of k-step s holds
    // Wk[row][32 s + 8 kg .. + 7]
    bf16x8 wh[16], wl[16];
    {
        const float4 *__restrict__ W4 = reinterpret_cast<const float4 *>(a.Wk) + 16 * wave + l15;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float4 w0 = W4[(int64_t)(2 * (4 * s + kg)) * 128], w1 = W4[(int64_t)(2 * (4 * s + kg) + 1) * 128];
            bf16x8 lo;
            ks_split8(w0, w1, wh[s], lo);
            if (LO) wl[s] = lo;
        }
    }
    // epilogue constants: accumulator register r is output 16w + 4 kg + r, i.e. quad 4w + kg
    const float4 vv = ld4(a.v + (4 * wave + kg) * 4), bb = ld4(a.b + (4 * wave + kg) * 4);

    // LDS-DMA of one tile: 32 requests of 1 KiB (4 quad rows x 16 columns x 16 B), wave w issues requests w, w + 8, w + 16, w + 24: lane
    // l of request j reads quad 4j + (l >> 4), column l & 15 and lands at slot + j KiB + 16 l.  Scalar 64-bit base + one loop-invariant
    // 32-bit lane offset (global_load_lds saddr form); M0 = LDS byte address of the request.
    const unsigned voff = (unsigned)(((int64_t)kg * Mc + l15) * 16);
    const unsigned ring_lds = (unsigned)(uintptr_t)sRing;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);  // provably wave-uniform: the request's base and M0 must be scalar registers
#define KS_DMA1(slot, j, gcol)                                                                       \
    if (SDFA_KS_EXP != 2) {                                                                          \
        const char *gb = reinterpret_cast<const char *>(H4 + (int64_t)(4 * (j)) * Mc + (gcol));     \
        const unsigned la = ring_lds + (unsigned)(((slot) * 32 + (j)) * 1024);                      \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la), "v"(voff), "s"(gb) : "memory"); \
    }
#define KS_SB() __builtin_amdgcn_sched_barrier(0);
#define KS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float *__restrict__ Sw = a.S + wave;                       // partial scores: [column][8 waves]

    for (int u = blockIdx.x; u < n_units; u += G) {
        const int f = u >> ts_shift, tb = u & ((1 << ts_shift) - 1);
        const int64_t col0 = (int64_t)(tb * TT) * Nc + (int64_t)f * KS_COLS;      // first column of the unit's first tile; the next tile is Nc columns on
        // the unit's query projections (+ b): a plain load, waited for HERE (builtin wait: the compiler's bookkeeping sees it) -- the ring
        // is empty at this point, later it would drain the tiles in flight
        float4 qb;
        {
            const float4 q0 = QP4[(int64_t)(4 * wave + kg) * Nc + f * KS_COLS + l15];
            __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0)
            qb = make_float4(q0.x + bb.x, q0.y + bb.y, q0.z + bb.z, q0.w + bb.w);
        }
        KS_SB()
        // every wave is past the previous unit's last MFMA read (its last barrier), so the ring is free.  Tiles 0 .. 3 -> slots 0 .. 3.
#pragma unroll
        for (int k = 0; k < KS_RING - 1; ++k)
            if (k < TT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) KS_DMA1(k, wave_u + 8 * r, col0 + (int64_t)k * Nc)
            }
        // tile 0: landed -> split in place, not overlapped (once per unit)
        if (TT >= KS_RING - 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KS_BARRIER()
        {
            float4 *rs = sRing + (2 * pr) * KS_COLS + c;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4 xa = rs[(64 * q) * KS_COLS], xb = rs[(64 * q + 1) * KS_COLS];
                bf16x8 hi, lo;
                ks_split8(xa, xb, hi, lo);
                *reinterpret_cast<bf16x8 *>(rs + (64 * q) * KS_COLS) = hi;
                if (LO) *reinterpret_cast<bf16x8 *>(rs + (64 * q + 1) * KS_COLS) = lo;
            }
        }
        // tile 1 must have landed before the loop's first trip splits it
        if (TT >= KS_RING - 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KS_BARRIER()
        int slot = 0;                                         // ring slot of tile i
        for (int i = 0; i < TT; ++i) {
            // the slot tile i - 1 left at the end of the previous trip takes tile i + 4
            const bool more = i + KS_RING - 1 < TT;
            const int64_t gnext = col0 + (int64_t)(i + KS_RING - 1) * Nc;
            const int slotp = slot == 0 ? KS_RING - 1 : slot - 1, slot1 = slot == KS_RING - 1 ? 0 : slot + 1;
            f32x4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = 0.f;
            // B fragments one k-step ahead of the MFMAs that use them; between the k-steps: the four DMA requests of tile i + 4 (every
            // fourth step) and the in-place split of tile i + 1 -- octet q = s >> 3 of this thread: s & 7 == 0 read the two quads,
            // 2 hi plane, 4 lo plane, 6 store both.  The SIMD's other wave fills the gaps with its own MFMAs.
            const bf16x8 *rd = reinterpret_cast<const bf16x8 *>(sRing + (size_t)slot * SLOT) + (2 * kg) * KS_COLS + l15;
            float4 *rs = sRing + (size_t)slot1 * SLOT + (2 * pr) * KS_COLS + c;
            bf16x8 bh[2], bl[2], shi, slo;
            float4 xa, xb;
            bh[0] = rd[0];
            if (LO) bl[0] = rd[KS_COLS];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                KS_SB()
                if (s + 1 < 16) {
                    bh[(s + 1) & 1] = rd[(8 * (s + 1)) * KS_COLS];
                    if (LO) bl[(s + 1) & 1] = rd[(8 * (s + 1) + 1) * KS_COLS];
                }
                if ((s & 3) == 1 && more) KS_DMA1(slotp, wave_u + 8 * (s >> 2), gnext)
                if (SDFA_KS_EXP != 3) {
                    const int q = s >> 3;
                    if ((s & 7) == 0) { xa = rs[(64 * q) * KS_COLS]; xb = rs[(64 * q + 1) * KS_COLS]; }
                    if ((s & 7) == 2) {
                        const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) shi[e] = (__bf16)x[e];
                    }
                    if ((s & 7) == 4 && LO) {
                        const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) slo[e] = (__bf16)(x[e] - (float)shi[e]);
                    }
                    if ((s & 7) == 6) {
                        *reinterpret_cast<bf16x8 *>(rs + (64 * q) * KS_COLS) = shi;
                        if (LO) *reinterpret_cast<bf16x8 *>(rs + (64 * q + 1) * KS_COLS) = slo;
                    }
                }
                KS_SB()
                if (SDFA_KS_EXP != 1) {
                    if (LO) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s], bh[s & 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], bl[s & 1], acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], bh[s & 1], acc, 0, 0, 0);
                }
            }
            KS_SB()
            float part;
            if (SDFA_KS_EXP == 4) part = acc[0] + acc[1] + acc[2] + acc[3];
            else {
                part = vv.x * tanhf_acc(acc[0] + qb.x);
                part += vv.y * tanhf_acc(acc[1] + qb.y);
                part += vv.z * tanhf_acc(acc[2] + qb.z);
                part += vv.w * tanhf_acc(acc[3] + qb.w);
            }
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            if (kg == 0) Sw[(col0 + (int64_t)i * Nc + l15) * 8] = part;
            slot = slot1;
            // end of the trip: tile i + 1 is split once every wave is here, and tile i + 2 must have landed for the next trip's split.
            // vmcnt retires in order and counts this wave's score stores too: behind tile i + 2's requests came the store of trip i - 2,
            // tile i + 3's four requests, the store of trip i - 1, tile i + 4's four, this trip's store = 11 operations that may stay out
            // (the first two trips have fewer stores behind them; a count too LARGE would let the split read a tile that is not there,
            // one too small waits for a request issued a few hundred cycles ago -- an HBM round trip per tile).  The last tiles of a
            // unit, with no requests behind them, drain.
            if (i + KS_RING - 1 < TT) {
                if (i >= 2) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
                else if (i == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            KS_BARRIER()
        }
    }
#undef KS_DMA1
#undef KS_SB
#undef KS_BARRIER
}

// ------------------------------------------------------------------------------------------------
// The same pass in exact fp32 (the headline precision): v_mfma_f32_16x16x4_f32, eight waves, wave w owns outputs 16w .. 16w+15 with its
// 16 x 512 slice of proj_key in 128 registers (lane (row, k group g) holds Wk[row][16 s + 4 g + c] as component c of quad register s:
// MFMA 4 s + c contracts k = 16 s + 4 g + c, g = 0 .. 3), the B operand straight out of the ring -- the DMA's [quad][column] image IS
// the fragment layout (lane (column, g) reads quad 4 s + g: one ds_read_b128 feeds four MFMAs), so there is no split pass and the ring
// slots are read-only: five slots, four tiles in flight.  fp32 MFMA makes this pass MATRIX-bound (16 x fewer FLOP per cycle than
// bf16: 8,192 cycles of MFMAs per tile and SIMD against 3,900 of HBM time), at the MFMA rate instead of the 128 x 128 GEMM's 0.78 of
// it, and the key projections (0.67 GB per step written and read back) no longer exist.  A dot product's k order differs from the
// GEMM's (two accumulators, even and odd quads), so the scores differ from round 5's in the last bits -- the parity bounds (align 1e-5
// against the reference) are unchanged.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void attn_key_score_f32_kernel(AttnKeyArgs a) {
    constexpr int SLOT = 128 * KS_COLS;
    extern __shared__ float4 sRingF[];                        // [KS_RING][128 quads][16 columns]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    const float4 *__restrict__ H4 = reinterpret_cast<const float4 *>(a.H);
    const float4 *__restrict__ QP4 = reinterpret_cast<const float4 *>(a.QP);
    const int64_t Mc = a.Mc, Nc = a.Nc;
    const int G = gridDim.x;
    const int ts_shift = a.ts_shift, TT = 64 >> ts_shift;
    const int n_units = (int)(Nc / KS_COLS) << ts_shift;

    float4 wq[32];
    {
        const float4 *__restrict__ W4 = reinterpret_cast<const float4 *>(a.Wk) + 16 * wave + l15;
#pragma unroll
        for (int s = 0; s < 32; ++s) wq[s] = W4[(int64_t)(4 * s + kg) * 128];
    }
    const float4 vv = ld4(a.v + (4 * wave + kg) * 4), bb = ld4(a.b + (4 * wave + kg) * 4);
    const unsigned voff = (unsigned)(((int64_t)kg * Mc + l15) * 16);
    const unsigned ring_lds = (unsigned)(uintptr_t)sRingF;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#define KF_DMA1(slot, j, gcol)                                                                       \
    {                                                                                               \
        const char *gb = reinterpret_cast<const char *>(H4 + (int64_t)(4 * (j)) * Mc + (gcol));     \
        const unsigned la = ring_lds + (unsigned)(((slot) * 32 + (j)) * 1024);                      \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la), "v"(voff), "s"(gb) : "memory"); \
    }
#define KF_SB() __builtin_amdgcn_sched_barrier(0);
#define KF_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float *__restrict__ Sw = a.S + wave;

    for (int u = blockIdx.x; u < n_units; u += G) {
        const int f = u >> ts_shift, tb = u & ((1 << ts_shift) - 1);
        const int64_t col0 = (int64_t)(tb * TT) * Nc + (int64_t)f * KS_COLS;
        float4 qb;
        {
            const float4 q0 = QP4[(int64_t)(4 * wave + kg) * Nc + f * KS_COLS + l15];
            __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the ring is empty here (see attn_key_score_kernel)
            qb = make_float4(q0.x + bb.x, q0.y + bb.y, q0.z + bb.z, q0.w + bb.w);
        }
        KF_SB()
        KF_BARRIER()                                          // every wave is past the previous unit's last ring read
#pragma unroll
        for (int k = 0; k < KS_RING - 1; ++k)
            if (k < TT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) KF_DMA1(k, wave_u + 8 * r, col0 + (int64_t)k * Nc)
            }
        int slot = 0;
        for (int i = 0; i < TT; ++i) {
            // tile i has landed once at most the operations issued behind its four requests are outstanding: the requests of the three
            // tiles behind it (12) and one score store per trip since (vmcnt retires in order; a count too large would read a tile that
            // is not there, one too small waits for a request issued a moment ago).  The last tiles of a unit drain.
            if (i + KS_RING - 2 < TT) {
                if (i >= 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (i == 3) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                else if (i == 2) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                else if (i == 1) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            KF_BARRIER()                                      // tile i is in LDS for every wave; every wave is done with tile i - 1
            const bool more = i + KS_RING - 1 < TT;
            const int64_t gnext = col0 + (int64_t)(i + KS_RING - 1) * Nc;
            const int slotp = slot == 0 ? KS_RING - 1 : slot - 1;
            const float4 *rd = sRingF + (size_t)slot * SLOT + kg * KS_COLS + l15;
            f32x4 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            float4 bq[2];
            bq[0] = rd[0];
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                KF_SB()
                if (s + 1 < 32) bq[(s + 1) & 1] = rd[(4 * (s + 1)) * KS_COLS];
                if ((s & 7) == 1 && more) KF_DMA1(slotp, wave_u + 8 * (s >> 3), gnext)
                KF_SB()
                const float4 b4 = bq[s & 1], w4 = wq[s];
                if (s & 1) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, b4.x, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, b4.y, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, b4.z, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, b4.w, acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, b4.x, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, b4.y, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, b4.z, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, b4.w, acc0, 0, 0, 0);
                }
            }
            KF_SB()
            float part = vv.x * tanhf_acc((acc0[0] + acc1[0]) + qb.x);
            part += vv.y * tanhf_acc((acc0[1] + acc1[1]) + qb.y);
            part += vv.z * tanhf_acc((acc0[2] + acc1[2]) + qb.z);
            part += vv.w * tanhf_acc((acc0[3] + acc1[3]) + qb.w);
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            if (kg == 0) Sw[(col0 + (int64_t)i * Nc + l15) * 8] = part;
            slot = slot == KS_RING - 1 ? 0 : slot + 1;
        }
    }
#undef KF_DMA1
#undef KF_SB
#undef KF_BARRIER
}

// ------------------------------------------------------------------------------------------------
// The WHOLE attention layer in one pass over H (exact fp32, large batches: a work unit = 16 frames x all 64 time steps).
// attn_key_score_f32_kernel + attn_kernel<true> read H twice (projection, then context); here the context is accumulated while the
// tile is still in the ring, with a running softmax (the flash-attention recurrence over the 64 keys of a frame):
//     m' = max(m, s_t);  alpha = exp(m - m');  p = exp(s_t - m');  l = l alpha + p;  ctx = ctx alpha + p x_t          (t = 0 .. 63)
//     z = ctx / l;  align_t = exp(s_t - m_final) / l      (scores kept in 4 KiB of LDS for the weights)
// -- mathematically torch.softmax + bmm (attentions.py:69-75,121-124), rounding in another order than a two-pass softmax.  The
// two-kernel form (attn_kernel<true>) runs the SAME recurrence with the same helpers: the forms agree bit for bit.  The pass stays matrix-bound: the context update is 16 FMAs per thread and tile, done
// at the top of the next trip (the tile's eight partial scores meet in LDS behind the trip's barrier) before the tile's slot is handed
// to the DMA.  Ring of four slots (being folded, being multiplied, two in flight), one barrier per tile, 133 KiB of LDS.  Thread (frame n = tid & 15, feature group fg = tid >> 4)
// owns features 16 fg .. 16 fg + 15 of frame n's context.
// ------------------------------------------------------------------------------------------------
constexpr int KA_RING = 4;
constexpr size_t ka_lds_bytes() { return (size_t)KA_RING * 128 * KS_COLS * 16 + 2 * 8 * KS_COLS * 4 + 64 * KS_COLS * 4; }

__global__ __launch_bounds__(512, 2) void attn_fused_f32_kernel(AttnKeyArgs a) {
    constexpr int SLOT = 128 * KS_COLS, TT = 64;
    extern __shared__ float4 sRingA[];                        // [KA_RING][128 quads][16 columns]
    float *sPart = reinterpret_cast<float *>(sRingA + KA_RING * SLOT);      // [2][8 waves][16] partial scores of a tile
    float *sScore = sPart + 2 * 8 * KS_COLS;                                // [64][16] the unit's scores
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    const int cn = tid & 15, fg = tid >> 4;                   // context role: frame cn, features 16 fg ..
    const float4 *__restrict__ H4 = reinterpret_cast<const float4 *>(a.H);
    const float4 *__restrict__ QP4 = reinterpret_cast<const float4 *>(a.QP);
    const int64_t Mc = a.Mc, Nc = a.Nc;
    const int G = gridDim.x;
    const int n_units = (int)(Nc / KS_COLS);

    float4 wq[32];
    {
        const float4 *__restrict__ W4 = reinterpret_cast<const float4 *>(a.Wk) + 16 * wave + l15;
#pragma unroll
        for (int s = 0; s < 32; ++s) wq[s] = W4[(int64_t)(4 * s + kg) * 128];
    }
    const float4 vv = ld4(a.v + (4 * wave + kg) * 4), bb = ld4(a.b + (4 * wave + kg) * 4);
    const unsigned voff = (unsigned)(((int64_t)kg * Mc + l15) * 16);
    const unsigned ring_lds = (unsigned)(uintptr_t)sRingA;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#define KA_DMA1(slot, j, gcol)                                                                       \
    {                                                                                               \
        const char *gb = reinterpret_cast<const char *>(H4 + (int64_t)(4 * (j)) * Mc + (gcol));     \
        const unsigned la = ring_lds + (unsigned)(((slot) * 32 + (j)) * 1024);                      \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la), "v"(voff), "s"(gb) : "memory"); \
    }
#define KA_SB() __builtin_amdgcn_sched_barrier(0);
#define KA_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // tile t_ (in ring slot sl_) is complete behind the barrier: its score, the running softmax, the context
#define KA_FOLD(t_, sl_)                                                                            \
    {                                                                                               \
        const float *pp = sPart + ((t_) & 1) * 8 * KS_COLS + cn;                                    \
        const float sc = score_of_partials(pp[0], pp[KS_COLS], pp[2 * KS_COLS], pp[3 * KS_COLS], pp[4 * KS_COLS], pp[5 * KS_COLS], pp[6 * KS_COLS], pp[7 * KS_COLS]); \
        if (fg == 0) sScore[(t_) * KS_COLS + cn] = sc;                                              \
        float alpha, pw;                                                                            \
        soft_step(sc, m_run, l_run, alpha, pw);                                                     \
        const float4 *xs = sRingA + (size_t)(sl_) * SLOT + (4 * fg) * KS_COLS + cn;                 \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) ctx_step(ctx[q], alpha, pw, xs[q * KS_COLS]); \
    }

    for (int u = blockIdx.x; u < n_units; u += G) {
        const int64_t col0 = (int64_t)u * KS_COLS;            // first column of the unit's first tile (t = 0); the next tile is Nc columns on
        float4 qb;
        {
            const float4 q0 = QP4[(int64_t)(4 * wave + kg) * Nc + u * KS_COLS + l15];
            __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the ring is empty here; also drains the previous unit's stores
            qb = make_float4(q0.x + bb.x, q0.y + bb.y, q0.z + bb.z, q0.w + bb.w);
        }
        KA_SB()
        KA_BARRIER()                                          // every wave is past the previous unit's last LDS read
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int r = 0; r < 4; ++r) KA_DMA1(k, wave_u + 8 * r, col0 + (int64_t)k * Nc)
        }
        float4 ctx[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ctx[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        float m_run = -3.0e38f, l_run = 0.f;
        int slot = 0;
        for (int i = 0; i < TT; ++i) {
            // The ring in trip i: tile i - 2's slot (folded in by every wave during trip i - 1) takes tile i + 2; tile i - 1 is folded into
            // the context; tile i is multiplied; tile i + 1 is in flight.  ONE barrier per trip.  Tile i has landed once only tile i + 1's
            // four requests are outstanding (no stores in the stream inside a unit).
            if (i + 1 < TT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            KA_BARRIER()                                      // tile i in LDS for every wave; tile i - 1's partial scores complete; tile i - 2 folded by all
            const int slotp = slot == 0 ? KA_RING - 1 : slot - 1, slotn = slot >= KA_RING - 2 ? slot + 2 - KA_RING : slot + 2;
            const bool more = i + 2 < TT;
            const int64_t gnext = col0 + (int64_t)(i + 2) * Nc;
            const float4 *rd = sRingA + (size_t)slot * SLOT + kg * KS_COLS + l15;
            f32x4 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            float4 bq[2];
            bq[0] = rd[0];
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                KA_SB()
                if (s + 1 < 32) bq[(s + 1) & 1] = rd[(4 * (s + 1)) * KS_COLS];
                if ((s & 7) == 1 && more) KA_DMA1(slotn, wave_u + 8 * (s >> 3), gnext)
                KA_SB()
                const float4 b4 = bq[s & 1], w4 = wq[s];
                if (s & 1) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, b4.x, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, b4.y, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, b4.z, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, b4.w, acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, b4.x, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, b4.y, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, b4.z, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, b4.w, acc0, 0, 0, 0);
                }
            }
            KA_SB()
            if (i > 0) KA_FOLD(i - 1, slotp)                  // (behind the MFMAs in program order: they are queued while this vector work runs)
            KA_SB()
            float part = vv.x * tanhf_acc((acc0[0] + acc1[0]) + qb.x);
            part += vv.y * tanhf_acc((acc0[1] + acc1[1]) + qb.y);
            part += vv.z * tanhf_acc((acc0[2] + acc1[2]) + qb.z);
            part += vv.w * tanhf_acc((acc0[3] + acc1[3]) + qb.w);
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            if (kg == 0) sPart[((i & 1) * 8 + wave) * KS_COLS + l15] = part;
            slot = slot == KA_RING - 1 ? 0 : slot + 1;
        }
        KA_BARRIER()
        KA_FOLD(TT - 1, (slot == 0 ? KA_RING - 1 : slot - 1))
        // the unit's 16 frames: z = ctx / l (K4 for the output MLPs + row-major), attention weights from the stored scores
        const int64_t n = (int64_t)u * KS_COLS + cn;
        const float inv = 1.0f / l_run;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 z = make_float4(ctx[q].x * inv, ctx[q].y * inv, ctx[q].z * inv, ctx[q].w * inv);
            st4(a.Zk4 + ((int64_t)(4 * fg + q) * Nc + n) * 4, z);
            if (a.z_out && n < a.N) st4(a.z_out + n * 512 + (4 * fg + q) * 4, z);
        }
        KA_BARRIER()                                          // the last score (written by feature group 0) is in LDS -- every thread meets here
        if (a.align_out && n < a.N) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int t = 2 * fg + e;
                a.align_out[n * 64 + t] = __expf(sScore[t * KS_COLS + cn] - m_run) * inv;
            }
        }
    }
#undef KA_DMA1
#undef KA_SB
#undef KA_BARRIER
#undef KA_FOLD
}

// row-major [n][F] -> K4 [F/4][ld]; columns n >= N are zero-filled
__global__ void rows_to_k4_kernel(const float *__restrict__ src, int64_t N, int F, float *__restrict__ dst, int64_t ld) {
    const int64_t n = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    if (n >= ld) return;
    for (int q = threadIdx.x >> 6; q < F / 4; q += blockDim.x >> 6) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N) v = ld4(src + n * F + 4 * q);
        st4(dst + ((int64_t)q * ld + n) * 4, v);
    }
}

// row-major src[n*src_ld + c0 + j], j < nf  ->  K4 features f0 .. f0+nf_pad-1 (f0, nf_pad multiples of 4); features past nf
// and columns n >= N are zero-filled
__global__ void rows_seg_to_k4_kernel(const float *__restrict__ src, int64_t src_ld, int64_t N, int c0, int nf,
                                      float *__restrict__ dst, int64_t ld, int f0, int nf_pad) {
    const int64_t n = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    if (n >= ld) return;
    for (int q = threadIdx.x >> 6; q < nf_pad / 4; q += blockDim.x >> 6) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (n < N && 4 * q + e < nf) ? src[n * src_ld + c0 + 4 * q + e] : 0.f;
        st4(dst + ((int64_t)(f0 / 4 + q) * ld + n) * 4, make_float4(v[0], v[1], v[2], v[3]));
    }
}

// K4 [.. /4][ld] features f0 .. f0+nf-1  ->  row-major dst[n*dst_ld + (f - f0)]
__global__ void k4_to_rows_kernel(const float *__restrict__ src, int64_t ld, int64_t N, int f0, int nf,
                                  float *__restrict__ dst, int64_t dst_ld) {
    const int64_t n = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    if (n >= N) return;
    for (int f = f0 + (threadIdx.x >> 6); f < f0 + nf; f += blockDim.x >> 6)
        dst[n * dst_ld + (f - f0)] = src[((int64_t)(f >> 2) * ld + n) * 4 + (f & 3)];
}

// debug taps: K4 [F/4][Mc], m = t*Nc + n  ->  the reference's layouts
//   what 0: pool1 (n, 32 ci, 64 f, 64 t)   K4 rows f*32+ci
//   what 1: conv3 (n, 64 ch, 32 f, 64 t)   K4 rows f*64+ch
//   what 2: freq  (n, 256, 64 t)           K4 rows c
//   what 3: bilstm (n, 64 t, 512)          K4 rows c
__global__ void tap_kernel(const float *__restrict__ src, int what, int64_t N, int64_t Nc, float *__restrict__ dst) {
    const int64_t Mc = 64 * Nc;
    const int64_t total = N * (what <= 1 ? 2048 * 64 : (what == 2 ? 256 * 64 : 512 * 64));
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t n, row, t;
        if (what == 0) {
            n = i / (2048 * 64); int64_t r = i % (2048 * 64);
            int ci = r / (64 * 64), f = (r / 64) % 64; t = r % 64; row = f * 32 + ci;
        } else if (what == 1) {
            n = i / (2048 * 64); int64_t r = i % (2048 * 64);
            int ch = r / (32 * 64), f = (r / 64) % 32; t = r % 64; row = f * 64 + ch;
        } else if (what == 2) {
            n = i / (256 * 64); int64_t r = i % (256 * 64);
            row = r / 64; t = r % 64;
        } else {
            n = i / (512 * 64); int64_t r = i % (512 * 64);
            t = r / 512; row = r % 512;
        }
        dst[i] = src[((row >> 2) * Mc + t * Nc + n) * 4 + (row & 3)];
    }
}

}  // namespace

hipError_t sdfa_launch_attn(const AttnArgs &a, hipStream_t s) {
    if (a.S) hipLaunchKernelGGL(attn_kernel<true>, dim3((unsigned)(a.Nc / ATT_FR)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(attn_kernel<false>, dim3((unsigned)(a.Nc / ATT_FR)), dim3(256), 0, s, a);
    return hipGetLastError();
}

bool sdfa_attn_fuses_tail(int64_t Nc, int reserve_cus) {
    // one unit = 16 frames x all 64 time steps (no split over time: the running softmax is per frame).  Worth it from about one unit
    // per CU: with fewer, the two-kernel form's shorter units (a quarter or an eighth of the time steps each) fill the chip better
    const int64_t cus = std::max(1, sdfa_cu_count() - reserve_cus);
    return (Nc / KS_COLS) >= cus - cus / 8;
}

hipError_t sdfa_launch_attn_key_score(const AttnKeyArgs &a, hipStream_t s) {
    if ((a.terms != 0 && a.terms != 1 && a.terms != 3) || a.Nc % 128 || a.Mc != 64 * a.Nc || 3 * a.Mc * 16 + 256 >= ((int64_t)1 << 32)) return hipErrorInvalidValue;
    // work units: (16 frames) x (64 >> ts_shift time steps); enough of them to give every CU two where the batch allows, never fewer
    // than eight time steps per unit (the ring is five tiles deep)
    const int cus = std::max(1, sdfa_cu_count() - a.reserve_cus);
    AttnKeyArgs b = a;
    b.ts_shift = 0;
    while (b.ts_shift < 3 && ((a.Nc / KS_COLS) << b.ts_shift) < 2 * (int64_t)cus) ++b.ts_shift;
    const int64_t n_units = (a.Nc / KS_COLS) << b.ts_shift;
    const int grid = (int)std::min<int64_t>(n_units, cus);
    const size_t lds = ks_lds_bytes(a.terms);
    hipError_t e;
    if (a.terms == 0 && a.fuse_tail) {      // the whole layer in one pass: a unit is 16 frames x all 64 time steps
        const size_t ldsa = ka_lds_bytes();
        const int grid_a = (int)std::min<int64_t>(a.Nc / KS_COLS, cus);
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fused_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsa);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(attn_fused_f32_kernel, dim3(grid_a), dim3(512), ldsa, s, b);
        return hipGetLastError();
    }
    if (a.terms == 0) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_key_score_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(attn_key_score_f32_kernel, dim3(grid), dim3(512), lds, s, b);
    } else if (a.terms == 3) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_key_score_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(attn_key_score_kernel<3>, dim3(grid), dim3(512), lds, s, b);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_key_score_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(attn_key_score_kernel<1>, dim3(grid), dim3(512), lds, s, b);
    }
    return hipGetLastError();
}

hipError_t sdfa_launch_rows_to_k4(const float *src, int64_t N, int F, float *dst, int64_t ld, hipStream_t s) {
    hipLaunchKernelGGL(rows_to_k4_kernel, dim3((unsigned)((ld + 63) / 64)), dim3(256), 0, s, src, N, F, dst, ld);
    return hipGetLastError();
}

hipError_t sdfa_launch_rows_seg_to_k4(const float *src, int64_t src_ld, int64_t N, int c0, int nf, float *dst, int64_t ld,
                                      int f0, int nf_pad, hipStream_t s) {
    hipLaunchKernelGGL(rows_seg_to_k4_kernel, dim3((unsigned)((ld + 63) / 64)), dim3(256), 0, s, src, src_ld, N, c0, nf, dst, ld, f0, nf_pad);
    return hipGetLastError();
}

hipError_t sdfa_launch_k4_to_rows(const float *src, int64_t ld, int64_t N, int F, int f0, int nf, float *dst,
                                  int64_t dst_ld, hipStream_t s) {
    (void)F;
    hipLaunchKernelGGL(k4_to_rows_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, s, src, ld, N, f0, nf, dst, dst_ld);
    return hipGetLastError();
}

hipError_t sdfa_launch_tap(const float *src, int what, int64_t N, int64_t Nc, float *dst, hipStream_t s) {
    hipLaunchKernelGGL(tap_kernel, dim3(1024), dim3(256), 0, s, src, what, N, Nc, dst);
    return hipGetLastError();
}

// LSTM recurrences of the audio encoder on fp32 MFMA.
//
//   freq_lstm_kernel : FreqLstm (speech_anime/layers/freq_lstm.py:36-55) -- bidirectional LSTM(64 -> 128,
//                      biases) over the 32 frequency rows of every (frame, time-step) column.  The input
//                      projection is fused: each step contracts [x_f | h] (K = 64 + 128) against the
//                      concatenated weights, so no gate pre-activations ever go to HBM.
//   time_lstm_kernel : one layer/direction of torch.nn.LSTM(256 -> 256, bias=False, bidirectional)
//                      (speech_anime/layers/rnn.py:20-21) over the 64 time steps of every frame; the input
//                      projection x_t * W_ih^T comes precomputed from the GEMM (GX) and seeds the accumulators.
//
// Both: one workgroup owns 64 sequences (columns) for all steps, gate rows on the MFMA row axis and
// sequences on the lanes.  Gate rows are packed per wave as [i | f | g | o] x 32 hidden units, so the four
// gate tiles of one wave share a register layout and the cell update is purely elementwise in registers;
// only h crosses waves, through LDS in K4 layout (which is exactly what an accumulator quad stores).
// Weights stream from L2 every step (0.4-1 MB per direction; fp32 MFMA needs < 20 GB/s per CU of them).
// PyTorch gate order i, f, g, o;  c' = sig(f)*c + sig(i)*tanh(g);  h' = sig(o)*tanh(c').
#include "common.h"
#include "kernels.h"

#ifdef SDFA_STAMPS
__device__ unsigned long long g_lstamp[8];
__device__ unsigned long long g_lspan[4] = {~0ull, 0ull, 0ull, 0ull};
__device__ unsigned long long g_lsub[4];     // cell-update sub-phases: x DMA issue, columns 0-31, columns 32-63
__device__ unsigned long long g_lxcd[8][4];   // per XCD (block id % 8): max end, sum of lifetimes, count, last start   // min start, max end, sum of lifetimes (100 MHz ticks), sum of lifetimes (shader cycles)
#define LSTAMP(t) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define LSTAMP(t)
#endif

namespace {

// fp32 MFMA and the vector ALU do NOT overlap on this chip -- neither across the waves of a SIMD nor inside one wave
// (tools/mfma_cowave.hip, tools/mfma_shadow.hip: a wave streaming v_mfma_f32_32x32x2_f32 starves its partner's vector
// instructions completely, and a vector instruction issued behind an MFMA of the same wave costs the matrix pipe its full
// 4-5 cycles) -- so every vector instruction of the cell update is matrix-pipe time lost, and the update is written for
// the fewest of them: two elements per instruction (v_pk_mul/add/fma_f32), exponent arguments negated / made absolute by
// the source modifiers of v_exp_f32 instead of by separate multiplies.  sigmoid2 / tanh2 are the same operations on the same
// values as the scalar forms sigmoidf_acc / tanhf_acc of common.h (x * -log2e == -(x * log2e); (|x| * -2) * log2e ==
// -|x * (2 log2e)|: a scaling by two commutes with rounding).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4n __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x2 rcp2(f32x2 d) { return f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)}; }
__device__ __forceinline__ f32x2 exp2n(f32x2 y) { return f32x2{__builtin_amdgcn_exp2f(-y.x), __builtin_amdgcn_exp2f(-y.y)}; }   // 2^-y

__device__ __forceinline__ f32x2 sigmoid2(f32x2 x) {      // 1 / (1 + exp(-x))
    return rcp2(exp2n(x * 1.4426950408889634f) + 1.0f);
}

__device__ __forceinline__ f32x2 tanh2(f32x2 x) {         // (1 - e) / (1 + e), e = exp(-2|x|), sign restored
    const f32x2 y = x * 2.8853900817779268f;
    const f32x2 e = {__builtin_amdgcn_exp2f(-__builtin_fabsf(y.x)), __builtin_amdgcn_exp2f(-__builtin_fabsf(y.y))};
    const f32x2 t = (1.0f - e) * rcp2(1.0f + e);
    return f32x2{__builtin_copysignf(t.x, x.x), __builtin_copysignf(t.y, x.y)};
}

// One quad (4 hidden units x this lane's column) of an LSTM step.
//   FREQ = false (the BiLSTM recurrences): sigmoid2 / tanh2 as above, 41 vector instructions per element pair.
//   FREQ = true  (the frequency LSTM, where the cell update is 8 % of the dominant kernel's time: 320 of its 560 vector instructions
//   per wave and step are v_exp / v_rcp at ~11 cycles each, the rest ~5; round 3, same-call A/B 116.4 -> 115.3 ms per step):
//     * tanh(g) = 2 sigmoid(2g) - 1: 7 instructions instead of 10 (no |x| / copysign pair, one add less).  Saturates correctly
//       (e = inf -> rcp 0 -> -1; e = 0 -> 1); its absolute error near 0 is one rounding of a value near 1 (~1.2e-7) where the
//       (1 - e) / (1 + e) form has ~3e-8 -- the size of sigmoid's own error, and g only enters through i * g.
//     * sigmoid(o) * tanh(c') = (1 - e_c) / ((1 + e_o) (1 + e_c)), e_c = exp(-2c'), e_o = exp(-o): ONE reciprocal for both and no
//       sign handling.  Needs a finite e_c: |c'| < step count = 32 here (|c_t| <= |f c_{t-1}| + |i g| < |c_{t-1}| + 1), so
//       e_c <= 2^92.4 (also with the cell state in units of 2 log2 e, below: the exponent is the same number); e_o may overflow -- the product is then inf, its reciprocal 0 and h = (1 - e_c) * 0 = 0 = sigmoid(-inf).
//       (Not usable in the BiLSTM: 64 steps allow e_c = 2^185.)  Near c' = 0 it keeps the (1 - e) form's accuracy.
//   The end-to-end error against the CPU port is unchanged (max |d dgrad| 7.5e-7 on the 10 s clip); all frequency-LSTM launch
//   forms share this function and stay bit-identical to each other.
template <bool FREQ = false>
__device__ __forceinline__ void lstm_cell_quad(const f32x16 &ai, const f32x16 &af, const f32x16 &ag, const f32x16 &ao,
                                               f32x16 &c, int g, float4 &hq) {
    f32x2 hv[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = 4 * g + 2 * p;
        f32x2 cp = {c[r], c[r + 1]};
#ifdef SDFA_FAKE_CELL   /* timing experiment only: what does the cell math cost? */
        { const f32x2 cn = f32x2{af[r], af[r + 1]} * cp + f32x2{ai[r], ai[r + 1]} * f32x2{ag[r], ag[r + 1]}; c[r] = cn.x; c[r + 1] = cn.y; hv[p] = f32x2{ao[r], ao[r + 1]} * cn; continue; }
#endif
#ifdef SDFA_OLD_CELL   /* A/B build only (make EXP=OLD_CELL): round 3's cell update on unscaled weights */
        if (FREQ) {
            const f32x2 ig = sigmoid2(f32x2{ai[r], ai[r + 1]}), fg = sigmoid2(f32x2{af[r], af[r + 1]});
            const f32x2 fc = fg * cp;
            const f32x2 gg = __builtin_elementwise_fma(rcp2(exp2n(f32x2{ag[r], ag[r + 1]} * 2.8853900817779268f) + 1.0f), f32x2{2.0f, 2.0f}, f32x2{-1.0f, -1.0f});
            const f32x2 cn = __builtin_elementwise_fma(ig, gg, fc);
            c[r] = cn.x; c[r + 1] = cn.y;
            const f32x2 eo = exp2n(f32x2{ao[r], ao[r + 1]} * 1.4426950408889634f), ec = exp2n(cn * 2.8853900817779268f);
            hv[p] = (1.0f - ec) * rcp2((1.0f + eo) * (1.0f + ec));
            continue;
        }
#endif
        if (FREQ) {
            // Round 4: the accumulators arrive as exponents of two (the host folds log2 e / 2 log2 e into the weights and the bias:
            // api.cpp), and the cell state is kept in the same units, c~ = 2 log2 e * c (it never leaves the kernel) -- so no gate and
            // no tanh(c') argument needs a scaling multiply: 11 packed + 18 transcendental instructions per element pair (16 + 18).
            constexpr float K2 = 2.8853900817779268f;      // 2 log2 e
            const f32x2 ig = rcp2(exp2n(f32x2{ai[r], ai[r + 1]}) + 1.0f);
            const f32x2 fg = rcp2(exp2n(f32x2{af[r], af[r + 1]}) + 1.0f);
            const f32x2 fc = fg * cp;                               // rounded product first, then one fused multiply-add
            const f32x2 gg = __builtin_elementwise_fma(rcp2(exp2n(f32x2{ag[r], ag[r + 1]}) + 1.0f), f32x2{2.0f * K2, 2.0f * K2}, f32x2{-K2, -K2});   // 2 log2 e * tanh(g)
            const f32x2 cn = __builtin_elementwise_fma(ig, gg, fc);
            c[r] = cn.x; c[r + 1] = cn.y;
            const f32x2 eo = exp2n(f32x2{ao[r], ao[r + 1]}), ec = exp2n(cn);
            hv[p] = (1.0f - ec) * rcp2((1.0f + eo) * (1.0f + ec));
            continue;
        }
        const f32x2 ig = sigmoid2(f32x2{ai[r], ai[r + 1]});
        const f32x2 fg = sigmoid2(f32x2{af[r], af[r + 1]});
        const f32x2 fc = fg * cp;                                   // rounded product first, then one fused multiply-add
        {
            const f32x2 gg = tanh2(f32x2{ag[r], ag[r + 1]});
            const f32x2 og = sigmoid2(f32x2{ao[r], ao[r + 1]});
            const f32x2 cn = __builtin_elementwise_fma(ig, gg, fc);
            c[r] = cn.x; c[r + 1] = cn.y;
            hv[p] = og * tanh2(cn);
        }
    }
    hq = make_float4(hv[0].x, hv[0].y, hv[1].x, hv[1].y);
}

// SHARED (all frequency-LSTM kernels) = launched over the compacted distinct-column list (column sharing): same code, separate
// symbol so that profiles keep the two launch shapes apart.

// ------------------------------------------------------------------------------- frequency LSTM, second form
// Same arithmetic and the same per-accumulator order of operations as freq_lstm_kernel<.., 2, 2> (bit-identical output);
// what changed is WHEN things are requested and in which order the MFMAs are issued:
//   * the next x_f tile goes HBM -> LDS directly (global_load_lds_dwordx4 into the idle half of sX: no registers, no
//     staging stores), requested right after the K loop instead of at the top of the step.  Loads return in issue order,
//     so a tile requested before the first weight quads made every step's first MFMA wait for an HBM round trip;
//   * k-block 0 of the weights is requested for the NEXT step right after the K loop and stays live across the cell update
//     (the 16 registers the x tile no longer needs); all weight requests go through a buffer descriptor (scalar base +
//     per-lane 32-bit offset: no vector-ALU address arithmetic, no 64-bit address registers);
//   * MFMAs are issued component-major (mfma_block): consecutive MFMAs go to different accumulators, so a wave that
//     has the matrix pipe to itself -- its partner workgroup is in its cell update -- does not stall on the previous
//     MFMA's result every time;
//   * the barrier after the K loop is a bare s_barrier (every LDS read has been consumed by an MFMA by then): it does
//     not wait for the weight request in flight; the barrier at the end of the step waits for the DMA and the LDS writes
//     but not for the acknowledgements of the hidden-state stores.
//
// PERSIST (option freq_lstm_shape=5): the grid is two workgroups per CU and every workgroup takes (direction, column tile)
// pairs from a queue head in the workspace until the queue is empty, instead of one hardware-dispatched workgroup per
// pair.  Built because the stamps (tools/stamp_lstm2.py) show a CU slot empty 2.5-3 % of a launch between the end of one
// 1.5 ms workgroup and the start of the next, plus +-2 % between XCDs under the static block-id -> XCD partition.
// Bit-identical.  With the scalar cell update it measured 2.4-3.4 % SLOWER than hardware dispatch (48.3 -> 49.4 ms per 8192
// frames), with the packed one 0.5 % faster (and hardware dispatch 2.5 % slower): how two starving partners interleave is
// chaotic.  sdfa_model_autotune keeps both as fall-backs behind freq_lstm_v3_kernel.
template <bool SHARED, bool PERSIST>
__global__ __launch_bounds__(256, 2) void freq_lstm_v2_kernel(FreqLstmArgs a) {
    constexpr int NJ = 2, BT = 64;
    __shared__ float4 sH[32][BT];
    __shared__ float4 sX[2][16][BT];
    __shared__ float sBias[512];
    __shared__ int sTile;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const float4 *__restrict__ X3 = reinterpret_cast<const float4 *>(a.X3);
    float4 *__restrict__ HF = reinterpret_cast<float4 *>(a.HF);
    const int n_tiles = (int)(a.Mc / BT) * 2;
  for (;;) {      // PERSIST: one pass per tile taken from the queue; otherwise a single pass
    int dir;
    int64_t m0;
    if (PERSIST) {
        if (tid == 0) sTile = atomicAdd(a.tile_counter, 1);
        __syncthreads();                       // (also: every wave has left the previous tile's last step)
        const int t = __builtin_amdgcn_readfirstlane(sTile);
        if (t >= n_tiles) break;               // queue empty: every workgroup gets here
        dir = t & 1;
        m0 = (int64_t)(t >> 1) * BT;
        if (SHARED && m0 >= *a.col_limit) break;      // tiles come in column order: all later ones are past the limit too
    } else {
        dir = (blockIdx.x >> 3) & 1;
        m0 = (int64_t)(((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) * BT;
        if (SHARED && m0 >= *a.col_limit) return;
    }
    const float4 *__restrict__ W = reinterpret_cast<const float4 *>(a.W) + (size_t)dir * 48 * 512;

    sBias[tid] = a.bias[dir * 512 + tid];
    sBias[256 + tid] = a.bias[dir * 512 + 256 + tid];

    // x_f tile = 16 k-quad rows of 64 columns (1 KiB each): wave w moves rows w, w+4, w+8, w+12
    const float4 *__restrict__ xsrc = X3 + (int64_t)wave * a.Mc + m0 + lane;
#define XDMA(f, buf)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                      \
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(xsrc + (int64_t)((f)*16 + 4 * i) * a.Mc), \
                                         (void __attribute__((address_space(3))) *)(&sX[buf][4 * i + wave][0]), 16, 0, 0);
    XDMA(dir ? 31 : 0, 0)

    f32x16 c[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[j][r] = 0.f;

    // weight quads through a BUFFER descriptor: wave-uniform base in scalar registers (descriptor inputs made provably
    // uniform with readfirstlane), one loop-invariant 32-bit byte offset per lane, the k-block as scalar offset, the gate as
    // immediate -- no vector-ALU address arithmetic and no 64-bit address registers inside the K loop (with flat addresses
    // the loop sat at the register limit and the allocator copied half of each refilled operand set at the end of every
    // iteration, behind a full s_waitcnt)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long wptr = (unsigned long long)(W + wave * 128);
    // (readfirstlane returns a SIGNED int: without the casts a low half with bit 31 set sign-extends over the high half)
    const unsigned long long wuni = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wptr >> 32)) << 32) |
                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wptr);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wuni, 0, 48 * 512 * 16, 0x00020000);
    const unsigned woff = (unsigned)((l31 + h * 512) * 16);
#define FV_W1(so, g_) __builtin_bit_cast(float4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(wrs, woff + 512 * (g_), so, 0))
#define FV_WLOAD(kb, W0, W1, W2, W3)                                                                         \
    {                                                                                                        \
        const unsigned so = (unsigned)(kb) * (1024 * 16);                                                    \
        W0 = FV_W1(so, 0); W1 = FV_W1(so, 1); W2 = FV_W1(so, 2); W3 = FV_W1(so, 3);                          \
    }
#define FV_BLOAD(kb, B0, B1)                                                                                 \
    {                                                                                                        \
        const float4 *bsrc = (kb) < 8 ? &sX[cur][2 * (kb) + h][0] : &sH[2 * ((kb) - 8) + h][0];              \
        B0 = bsrc[l31]; B1 = bsrc[32 + l31];                                                                 \
    }
    // one k-block: 32 MFMAs, component-major (consecutive MFMAs go to different accumulators)
#define FV_MFMA(W0, W1, W2, W3, B0, B1)                                                                      \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                          \
        acc[0][0] = MFMA(SDFA_OP(f4c(W0, q)), SDFA_OP(f4c(B0, q)), acc[0][0]);                               \
        acc[0][1] = MFMA(SDFA_OP(f4c(W0, q)), SDFA_OP(f4c(B1, q)), acc[0][1]);                               \
        acc[1][0] = MFMA(SDFA_OP(f4c(W1, q)), SDFA_OP(f4c(B0, q)), acc[1][0]);                               \
        acc[1][1] = MFMA(SDFA_OP(f4c(W1, q)), SDFA_OP(f4c(B1, q)), acc[1][1]);                               \
        acc[2][0] = MFMA(SDFA_OP(f4c(W2, q)), SDFA_OP(f4c(B0, q)), acc[2][0]);                               \
        acc[2][1] = MFMA(SDFA_OP(f4c(W2, q)), SDFA_OP(f4c(B1, q)), acc[2][1]);                               \
        acc[3][0] = MFMA(SDFA_OP(f4c(W3, q)), SDFA_OP(f4c(B0, q)), acc[3][0]);                               \
        acc[3][1] = MFMA(SDFA_OP(f4c(W3, q)), SDFA_OP(f4c(B1, q)), acc[3][1]);                               \
    }
    float4 wn0, wn1, wn2, wn3;
    FV_WLOAD(0, wn0, wn1, wn2, wn3)
#ifdef SDFA_STAMPS
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t3a = 0, t3b = 0, t3c = 0, t4 = 0, t5 = 0, v_init = 0, v_k = 0, v_b1 = 0, v_ep = 0, v_b2 = 0, v_e0 = 0, v_e1 = 0, v_e2 = 0;
    const unsigned long long life_r0 = wall_clock64(), life_c0 = clock64();
#endif
    __syncthreads();   // bias and the first x tile are in LDS (the fence drains the DMA)

    for (int s = 0; s < 32; ++s) {
        const int f = dir ? 31 - s : s;
        const int cur = s & 1;
        // Two alternating operand sets (wa/ba, wb/bb), each refilled right after the MFMAs that read it were issued: requests
        // run one k-block (32 MFMAs) ahead.  k-block 0 comes from `wn`, requested during the previous step's cell update.
        // Named scalars, not arrays: with arrays (or with one set that is loop-carried through BOTH loops, or with the K loop
        // unrolled completely) the register allocator loaded refills into scratch registers and copied them over behind a
        // full s_waitcnt at the end of every K iteration, or spilled 253 registers.
        float4 wa0 = wn0, wa1 = wn1, wa2 = wn2, wa3 = wn3, wb0, wb1, wb2, wb3, ba0, ba1, bb0, bb1;
        f32x16 acc[4][NJ];
        LSTAMP(t0)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {      // (one read per column tile: see freq_lstm_v3_kernel)
                    const f32x4n b = *(const volatile f32x4n __attribute__((address_space(3))) *)(&sBias[wave * 128 + gt * 32 + 8 * g + 4 * h]);
                    acc[gt][j][4 * g + 0] = b.x; acc[gt][j][4 * g + 1] = b.y;
                    acc[gt][j][4 * g + 2] = b.z; acc[gt][j][4 * g + 3] = b.w;
                }
            }
        const int nkb = s > 0 ? 24 : 8;        // h_{-1} = 0: the first step contracts x_f only
        FV_BLOAD(0, ba0, ba1)
#ifndef SDFA_PRIO_MFMA
#define SDFA_PRIO_MFMA 1
#endif
#ifndef SDFA_PRIO_EPI
#define SDFA_PRIO_EPI 0
#endif
        __builtin_amdgcn_s_setprio(SDFA_PRIO_MFMA);
        LSTAMP(t1)
#pragma unroll 1
        for (int kb = 0; kb < nkb; kb += 2) {
            FV_WLOAD(kb + 1, wb0, wb1, wb2, wb3)
            FV_BLOAD(kb + 1, bb0, bb1)
            __builtin_amdgcn_sched_barrier(0);
            FV_MFMA(wa0, wa1, wa2, wa3, ba0, ba1)
            __builtin_amdgcn_sched_barrier(0);
            const int kb2 = kb + 2 < nkb ? kb + 2 : 0;   // branch-free: the last iteration re-requests k-block 0 and drops it
            FV_WLOAD(kb2, wa0, wa1, wa2, wa3)
            FV_BLOAD(kb2, ba0, ba1)
            __builtin_amdgcn_sched_barrier(0);
            FV_MFMA(wb0, wb1, wb2, wb3, bb0, bb1)
            __builtin_amdgcn_sched_barrier(0);
        }
        FV_WLOAD(0, wn0, wn1, wn2, wn3)     // k-block 0 for the NEXT step: in flight during the cell update (weights do not change)
        __builtin_amdgcn_s_setprio(SDFA_PRIO_EPI);
#ifdef SDFA_STAMPS
        asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]), "v"(acc[0][1]), "v"(acc[1][1]), "v"(acc[2][1]), "v"(acc[3][1]));   // all MFMAs done
#endif
        LSTAMP(t2)
        // every wave has finished reading sH / sX[cur] (all LDS reads were consumed by MFMAs, the wrap-around one is waited
        // for here); the weight request in flight is not waited for
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        LSTAMP(t3)
        if (s + 1 < 32) { XDMA(dir ? 30 - s : s + 1, cur ^ 1) }   // lands during the cell update; sX[cur ^ 1] was last read in step s - 1
        LSTAMP(t3a)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 hq;
                lstm_cell_quad<true>(acc[0][j], acc[1][j], acc[2][j], acc[3][j], c[j], g, hq);
                const int hq_idx = 8 * wave + 2 * g + h;
                sH[hq_idx][j * 32 + l31] = hq;
                HF[((m0 >> 7) * (int64_t)HF_SLAB_ROWS + (f * 64 + dir * 32 + hq_idx)) * 128 + (m0 & 127) + j * 32 + l31] = hq;
            }
#ifdef SDFA_STAMPS
            if (j == 0) LSTAMP(t3b) else LSTAMP(t3c)
#endif
        }
        // h_s and the next x tile must be in LDS before anyone starts step s+1: this wave's LDS writes (lgkmcnt) and its four
        // DMA requests.  Vector-memory operations of a wave complete in issue order on gfx9-family parts (one in-order
        // vmcnt for loads and stores -- what LLVM's own waitcnt insertion relies on), and the 8 hidden-state stores were
        // issued after the DMA, so vmcnt(8) is "DMA landed" without waiting for the stores' acknowledgements from L2
        // (a full __syncthreads() fence costs 0.8 % here).
        // (the builtin wait is the same instruction again: the compiler's own waitcnt bookkeeping does not look inside asm
        // and would otherwise put a full vmcnt(0) in front of the next LDS read -- inside the K loop -- for the DMA's sake)
        __builtin_amdgcn_s_waitcnt(0x0078);     // gfx9 encoding: vmcnt = 8, expcnt = 7 (no wait), lgkmcnt = 0
        LSTAMP(t4)
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef SDFA_STAMPS
        LSTAMP(t5)
        if (s > 0) { v_init += t1 - t0; v_k += t2 - t1; v_b1 += t3 - t2; v_ep += t4 - t3; v_b2 += t5 - t4; v_e0 += t3a - t3; v_e1 += t3b - t3a; v_e2 += t3c - t3b; }
#endif
    }
#ifdef SDFA_STAMPS
    if (lane == 0) {
        atomicAdd(&g_lstamp[0], v_init); atomicAdd(&g_lstamp[1], v_k); atomicAdd(&g_lstamp[2], v_b1); atomicAdd(&g_lstamp[3], v_ep);
        atomicAdd(&g_lstamp[4], v_b2); atomicAdd(&g_lstamp[6], 31ull);
        atomicAdd(&g_lsub[0], v_e0); atomicAdd(&g_lsub[1], v_e1); atomicAdd(&g_lsub[2], v_e2);
        if (wave == 0) {
            const unsigned long long r1 = wall_clock64();
            atomicMin(&g_lspan[0], life_r0); atomicMax(&g_lspan[1], r1);
            atomicAdd(&g_lspan[2], r1 - life_r0); atomicAdd(&g_lspan[3], (unsigned long long)clock64() - life_c0);
            atomicMax(&g_lxcd[blockIdx.x & 7][0], r1); atomicAdd(&g_lxcd[blockIdx.x & 7][1], r1 - life_r0); atomicAdd(&g_lxcd[blockIdx.x & 7][2], 1ull); atomicMax(&g_lxcd[blockIdx.x & 7][3], life_r0);
        }
    }
#endif
    if (!PERSIST) break;
  }
#undef XDMA
#undef FV_WLOAD
#undef FV_BLOAD
#undef FV_MFMA
#undef FV_W1
}

// ------------------------------------------------------------------------------- frequency LSTM, third form
// freq_lstm_v2_kernel rebuilt for ONE workgroup per CU (one wave per SIMD), after the finding that an MFMA in flight and
// the other instructions of a SIMD exclude each other: a second resident workgroup only fills
// stalls, so the design goal is a wave with no stalls and as few non-MFMA issue cycles as possible.
//   * x_f and h live in ONE LDS buffer per step parity, [16 x rows | 32 h rows] (96 KiB for both parities): the operand row
//     of k-block kb is base + kb * 2 KiB whatever it holds, so the K loop needs no per-k-block address arithmetic (the
//     second form spent four vector instructions per two k-blocks on a select between two arrays), and h being double
//     buffered makes the barrier between the K loop and the cell update unnecessary: ONE barrier per step;
//   * the K loop runs in trips of 8 k-blocks with immediate offsets (the first step, x only, is one trip), and the six
//     memory instructions that refill the other operand set sit one or two at a time in front of the four 8-MFMA groups
//     of a k-block: each rides in the shadow of the MFMA in flight instead of six in a row outlasting it;
//   * same arithmetic, same order of operations per accumulator: bit-identical.
// (launch bounds as for two workgroups per CU although the LDS footprint allows one: with a 512-register budget the compiler
// moves the accumulators into AGPRs and the cell update pays a v_accvgpr_read per value)
template <bool SHARED, bool PERSIST>
__global__ __launch_bounds__(256, 2) void freq_lstm_v3_kernel(FreqLstmArgs a) {
    constexpr int NJ = 2, BT = 64, ROWS = 48;
    extern __shared__ float4 sDyn3[];                                   // [2 parities][48 rows][64] + bias + queue slot
    float4 (*sXH)[ROWS][BT] = reinterpret_cast<float4 (*)[ROWS][BT]>(sDyn3);
    float *sBias = reinterpret_cast<float *>(sDyn3 + 2 * ROWS * BT);
    int *sTile = reinterpret_cast<int *>(sBias + 512);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const float4 *__restrict__ X3 = reinterpret_cast<const float4 *>(a.X3);
    float4 *__restrict__ HF = reinterpret_cast<float4 *>(a.HF);
    const int n_tiles = (int)(a.Mc / BT) * 2;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  for (;;) {      // PERSIST: one pass per tile taken from the queue; otherwise a single pass
    int dir;
    int64_t m0;
    if (PERSIST) {
        if (tid == 0) *sTile = atomicAdd(a.tile_counter, 1);
        __syncthreads();                       // (also: every wave has left the previous tile's last step)
        const int t = __builtin_amdgcn_readfirstlane(*sTile);
        if (t >= n_tiles) break;               // queue empty: every workgroup gets here
        dir = t & 1;
        m0 = (int64_t)(t >> 1) * BT;
        if (SHARED && m0 >= *a.col_limit) break;
    } else {
        dir = (blockIdx.x >> 3) & 1;
        m0 = (int64_t)(((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) * BT;
        if (SHARED && m0 >= *a.col_limit) return;
    }
    const float4 *__restrict__ W = reinterpret_cast<const float4 *>(a.W) + (size_t)dir * 48 * 512;
    sBias[tid] = a.bias[dir * 512 + tid];
    sBias[256 + tid] = a.bias[dir * 512 + 256 + tid];

    // x_f tile = 16 k-quad rows of 64 columns (1 KiB each): wave w moves rows w, w+4, w+8, w+12
    const float4 *__restrict__ xsrc = X3 + (int64_t)wave * a.Mc + m0 + lane;
#define XDMA3(f, par)                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                      \
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(xsrc + (int64_t)((f)*16 + 4 * i) * a.Mc), \
                                         (void __attribute__((address_space(3))) *)(&sXH[par][4 * i + wave][0]), 16, 0, 0);
    XDMA3(dir ? 31 : 0, 0)

    f32x16 c[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[j][r] = 0.f;

    // weight quads through a buffer descriptor, as in the second form
    const unsigned long long wptr = (unsigned long long)(W + wave * 128);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wptr, 0, 48 * 512 * 16, 0x00020000);
    const unsigned woff = (unsigned)((l31 + h * 512) * 16);
#define F3_W(so, g_) __builtin_bit_cast(float4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(wrs, woff + 512 * (g_), so, 0))
#define F3_SB() __builtin_amdgcn_sched_barrier(0);
#define F3_Q(W0, W1, W2, W3, B0, B1, q)                                                                      \
    {                                                                                                        \
        acc[0][0] = MFMA(SDFA_OP(f4c(W0, q)), SDFA_OP(f4c(B0, q)), acc[0][0]);                               \
        acc[0][1] = MFMA(SDFA_OP(f4c(W0, q)), SDFA_OP(f4c(B1, q)), acc[0][1]);                               \
        acc[1][0] = MFMA(SDFA_OP(f4c(W1, q)), SDFA_OP(f4c(B0, q)), acc[1][0]);                               \
        acc[1][1] = MFMA(SDFA_OP(f4c(W1, q)), SDFA_OP(f4c(B1, q)), acc[1][1]);                               \
        acc[2][0] = MFMA(SDFA_OP(f4c(W2, q)), SDFA_OP(f4c(B0, q)), acc[2][0]);                               \
        acc[2][1] = MFMA(SDFA_OP(f4c(W2, q)), SDFA_OP(f4c(B1, q)), acc[2][1]);                               \
        acc[3][0] = MFMA(SDFA_OP(f4c(W3, q)), SDFA_OP(f4c(B0, q)), acc[3][0]);                               \
        acc[3][1] = MFMA(SDFA_OP(f4c(W3, q)), SDFA_OP(f4c(B1, q)), acc[3][1]);                               \
    }
    // the step's very first MFMA group: column tile 1 starts from tile 0's seeds (C = acc[gt][0], D = acc[gt][1]) BEFORE tile 0's own
    // first product overwrites them in place -- same sums as seeding both (bitwise)
#define F3_QSEED(W0, W1, W2, W3, B0, B1)                                                                     \
    {                                                                                                        \
        acc[0][1] = MFMA(SDFA_OP(f4c(W0, 0)), SDFA_OP(f4c(B1, 0)), acc[0][0]);                               \
        acc[0][0] = MFMA(SDFA_OP(f4c(W0, 0)), SDFA_OP(f4c(B0, 0)), acc[0][0]);                               \
        acc[1][1] = MFMA(SDFA_OP(f4c(W1, 0)), SDFA_OP(f4c(B1, 0)), acc[1][0]);                               \
        acc[1][0] = MFMA(SDFA_OP(f4c(W1, 0)), SDFA_OP(f4c(B0, 0)), acc[1][0]);                               \
        acc[2][1] = MFMA(SDFA_OP(f4c(W2, 0)), SDFA_OP(f4c(B1, 0)), acc[2][0]);                               \
        acc[2][0] = MFMA(SDFA_OP(f4c(W2, 0)), SDFA_OP(f4c(B0, 0)), acc[2][0]);                               \
        acc[3][1] = MFMA(SDFA_OP(f4c(W3, 0)), SDFA_OP(f4c(B1, 0)), acc[3][0]);                               \
        acc[3][0] = MFMA(SDFA_OP(f4c(W3, 0)), SDFA_OP(f4c(B0, 0)), acc[3][0]);                               \
    }
    // one k-block on the operand set C* while the set N* is refilled for the next one: weights at scalar offset `so`,
    // operand rows at `bp` (per-lane pointer to the row pair of the next k-block).  F3_KB_SEED: the step's first k-block
#define F3_KB_SEED(CW0, CW1, CW2, CW3, CB0, CB1, NW0, NW1, NW2, NW3, NB0, NB1, so, bp)                       \
    {                                                                                                        \
        F3_SB() NW0 = F3_W(so, 0); NW1 = F3_W(so, 1); F3_SB()                                                \
        F3_QSEED(CW0, CW1, CW2, CW3, CB0, CB1)                                                               \
        F3_SB() NW2 = F3_W(so, 2); NW3 = F3_W(so, 3); F3_SB()                                                \
        F3_Q(CW0, CW1, CW2, CW3, CB0, CB1, 1)                                                                \
        F3_SB() NB0 = (bp)[0]; F3_SB()                                                                       \
        F3_Q(CW0, CW1, CW2, CW3, CB0, CB1, 2)                                                                \
        F3_SB() NB1 = (bp)[32]; F3_SB()                                                                      \
        F3_Q(CW0, CW1, CW2, CW3, CB0, CB1, 3)                                                                \
    }
#define F3_KB(CW0, CW1, CW2, CW3, CB0, CB1, NW0, NW1, NW2, NW3, NB0, NB1, so, bp)                            \
    {                                                                                                        \
        F3_SB() NW0 = F3_W(so, 0); NW1 = F3_W(so, 1); F3_SB()                                                \
        F3_Q(CW0, CW1, CW2, CW3, CB0, CB1, 0)                                                                \
        F3_SB() NW2 = F3_W(so, 2); NW3 = F3_W(so, 3); F3_SB()                                                \
        F3_Q(CW0, CW1, CW2, CW3, CB0, CB1, 1)                                                                \
        F3_SB() NB0 = (bp)[0]; F3_SB()                                                                       \
        F3_Q(CW0, CW1, CW2, CW3, CB0, CB1, 2)                                                                \
        F3_SB() NB1 = (bp)[32]; F3_SB()                                                                      \
        F3_Q(CW0, CW1, CW2, CW3, CB0, CB1, 3)                                                                \
    }
    // the last k-block of a step: the same, with the four LDS-DMA requests of the next x tile behind its last two MFMA
    // groups -- an LDS-DMA request takes ~100 cycles to issue (tools/stamp_fat.py), four in a row in front of the cell update
    // cost 400; here most of that rides in the MFMAs' shadow.  They come AFTER this k-block's weight requests, so the next
    // step's first weight wait (loads return in issue order) does not include the tile's HBM round trip.
#define F3_KB_LAST(CW0, CW1, CW2, CW3, CB0, CB1, NW0, NW1, NW2, NW3, NB0, NB1, so, bp, dma, fnext, par)      \
    {                                                                                                        \
        F3_SB() NW0 = F3_W(so, 0); NW1 = F3_W(so, 1); F3_SB()                                                \
        F3_Q(CW0, CW1, CW2, CW3, CB0, CB1, 0)                                                                \
        F3_SB() NW2 = F3_W(so, 2); NW3 = F3_W(so, 3); F3_SB()                                                \
        F3_Q(CW0, CW1, CW2, CW3, CB0, CB1, 1)                                                                \
        F3_SB() NB0 = (bp)[0]; NB1 = (bp)[32];      /* both LDS reads BEFORE the first DMA request: the compiler puts a full  \
                                                       vmcnt(0) in front of any LDS read that follows one */ \
        if (dma) { XDMA3_HALF(fnext, par, 0) }                                                               \
        F3_SB()                                                                                              \
        F3_Q(CW0, CW1, CW2, CW3, CB0, CB1, 2)                                                                \
        F3_SB()                                                                                              \
        if (dma) { XDMA3_HALF(fnext, par, 2) }                                                               \
        F3_SB()                                                                                              \
        F3_Q(CW0, CW1, CW2, CW3, CB0, CB1, 3)                                                                \
    }
#define XDMA3_HALF(f, par, i0)                                                                                         \
    _Pragma("unroll") for (int i = (i0); i < (i0) + 2; ++i)                                                            \
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(xsrc + (int64_t)((f)*16 + 4 * i) * a.Mc), \
                                         (void __attribute__((address_space(3))) *)(&sXH[par][4 * i + wave][0]), 16, 0, 0);
    float4 wa0, wa1, wa2, wa3, wb0, wb1, wb2, wb3, ba0, ba1, bb0, bb1;
    wa0 = F3_W(0u, 0); wa1 = F3_W(0u, 1); wa2 = F3_W(0u, 2); wa3 = F3_W(0u, 3);
#ifdef SDFA_STAMPS
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, v_init = 0, v_k = 0, v_ep = 0, v_b2 = 0;
#endif
    __syncthreads();   // bias and the first x tile are in LDS (the fence drains the DMA)

    for (int s = 0; s < 32; ++s) {
        const int f = dir ? 31 - s : s;
        const int cur = s & 1;
        f32x16 acc[4][NJ];
        LSTAMP(t0)
        // the first k-block's operand rows FIRST (LDS answers a wave's reads in order): the step's first MFMA then waits for these two and
        // its own four seed quads, the later seeds land under the MFMAs in front of them
        const float4 *brow = &sXH[cur][h][l31];                 // row pair of k-block kb: brow + kb * 2 * BT
        ba0 = brow[0]; ba1 = brow[32];
        F3_SB()
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // Seeds (the bias) go into column tile 0's accumulators ONLY; tile 1's first MFMA of the step takes them from there as its C
                // operand (F3_QSEED below) -- D != C costs nothing.  History: one read + a v_mov per register of the second tile cost 64
                // vector instructions per step (0.6 ms of the 114.5); a second read per quad removed those but left 32 reads of 1 KiB
                // per wave in front of every step's first MFMA, and the CU's LDS moves 4 waves x 34 KiB in ~550 cycles.
                const float4 b = *reinterpret_cast<const float4 *>(&sBias[wave * 128 + gt * 32 + 8 * g + 4 * h]);
                acc[gt][0][4 * g + 0] = b.x; acc[gt][0][4 * g + 1] = b.y;
                acc[gt][0][4 * g + 2] = b.z; acc[gt][0][4 * g + 3] = b.w;
            }
        const int trips = s > 0 ? 3 : 1;       // h_{-1} = 0: the first step contracts x_f only (8 of the 24 k-blocks)
        LSTAMP(t1)
        // a trip = 8 k-blocks.  Trip 0 is written out in front of the loop: its first MFMA group is the seeded one, and as straight-line code
        // the compiler sees that (a run-time `t == 0` inside the loop made it copy the 128 accumulator registers of tile 1 around the branch)
#define F3_TRIP(FIRST_KB, t_)                                                                                                     \
        {                                                                                                                         \
            const float4 *bt = brow + (t_) * 16 * BT;                                                                             \
            const float4 *bn = (t_) + 1 < trips ? bt + 16 * BT : brow;      /* behind the last trip: k-block 0 again (dropped) */  \
            const unsigned so = (unsigned)(t_) * (8 * 1024 * 16);                                                                 \
            const unsigned son = (t_) + 1 < trips ? so + 8 * 1024 * 16 : 0u; /* ... whose weights ARE k-block 0 of the next step */ \
            FIRST_KB(wa0, wa1, wa2, wa3, ba0, ba1, wb0, wb1, wb2, wb3, bb0, bb1, so + 1 * 1024 * 16, bt + 1 * 2 * BT)             \
            F3_KB(wb0, wb1, wb2, wb3, bb0, bb1, wa0, wa1, wa2, wa3, ba0, ba1, so + 2 * 1024 * 16, bt + 2 * 2 * BT)                \
            F3_KB(wa0, wa1, wa2, wa3, ba0, ba1, wb0, wb1, wb2, wb3, bb0, bb1, so + 3 * 1024 * 16, bt + 3 * 2 * BT)                \
            F3_KB(wb0, wb1, wb2, wb3, bb0, bb1, wa0, wa1, wa2, wa3, ba0, ba1, so + 4 * 1024 * 16, bt + 4 * 2 * BT)                \
            F3_KB(wa0, wa1, wa2, wa3, ba0, ba1, wb0, wb1, wb2, wb3, bb0, bb1, so + 5 * 1024 * 16, bt + 5 * 2 * BT)                \
            F3_KB(wb0, wb1, wb2, wb3, bb0, bb1, wa0, wa1, wa2, wa3, ba0, ba1, so + 6 * 1024 * 16, bt + 6 * 2 * BT)                \
            F3_KB(wa0, wa1, wa2, wa3, ba0, ba1, wb0, wb1, wb2, wb3, bb0, bb1, so + 7 * 1024 * 16, bt + 7 * 2 * BT)                \
            F3_KB(wb0, wb1, wb2, wb3, bb0, bb1, wa0, wa1, wa2, wa3, ba0, ba1, son, bn)                                            \
        }
        F3_TRIP(F3_KB_SEED, 0)
#pragma unroll 1
        for (int t = 1; t < trips; ++t) F3_TRIP(F3_KB, t)
#undef F3_TRIP
        F3_SB()
#ifdef SDFA_STAMPS
        asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]), "v"(acc[0][1]), "v"(acc[1][1]), "v"(acc[2][1]), "v"(acc[3][1]));   // all MFMAs done
#endif
        LSTAMP(t2)
        // The next x tile: four plain 16-byte loads per lane into registers, requested here -- BEHIND the step's last weight
        // requests, so that no weight wait includes their HBM round trip (loads return in issue order) -- and written to the
        // OTHER parity's buffer after the cell update, which covers the round trip.  (An LDS-DMA request for these rows, 8 MB
        // apart, took ~230 cycles to ISSUE: 930 per step, wherever it was placed; tools/stamp_lstm2.py.)  wa* now hold
        // k-block 0 of the next step's weights.
        float4 xr0, xr1, xr2, xr3;
        {
            const int fn = s + 1 < 32 ? (dir ? 30 - s : s + 1) : f;      // behind the last step: this step's rows again (never used)
            const float4 *xs = xsrc + (int64_t)(fn * 16) * a.Mc;
            xr0 = xs[0]; xr1 = xs[4 * a.Mc]; xr2 = xs[8 * a.Mc]; xr3 = xs[12 * a.Mc];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 hq;
                lstm_cell_quad<true>(acc[0][j], acc[1][j], acc[2][j], acc[3][j], c[j], g, hq);
                const int hq_idx = 8 * wave + 2 * g + h;
                sXH[cur ^ 1][16 + hq_idx][j * 32 + l31] = hq;
                HF[((m0 >> 7) * (int64_t)HF_SLAB_ROWS + (f * 64 + dir * 32 + hq_idx)) * 128 + (m0 & 127) + j * 32 + l31] = hq;
            }
        sXH[cur ^ 1][wave][lane] = xr0; sXH[cur ^ 1][4 + wave][lane] = xr1; sXH[cur ^ 1][8 + wave][lane] = xr2; sXH[cur ^ 1][12 + wave][lane] = xr3;
        // h_s and the next x tile must be in LDS before anyone starts step s+1: this wave's LDS writes (lgkmcnt); the
        // hidden-state stores need not be acknowledged (a bare barrier: __syncthreads() would add their vmcnt(0))
        LSTAMP(t3)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef SDFA_STAMPS
        LSTAMP(t4)
        if (s > 0) { v_init += t1 - t0; v_k += t2 - t1; v_ep += t3 - t2; v_b2 += t4 - t3; }
#endif
    }
#ifdef SDFA_STAMPS
    if (lane == 0) {
        atomicAdd(&g_lstamp[0], v_init); atomicAdd(&g_lstamp[1], v_k); atomicAdd(&g_lstamp[3], v_ep); atomicAdd(&g_lstamp[4], v_b2); atomicAdd(&g_lstamp[6], 31ull);
    }
#endif
#undef XDMA3
#undef XDMA3_HALF
#undef F3_KB_LAST
#undef F3_KB_SEED
#undef F3_QSEED
#undef F3_W
#undef F3_SB
#undef F3_Q
#undef F3_KB
    if (!PERSIST) break;
  }
}

// --------------------------------------------------------------------- frequency LSTM on bf16 MFMA
// Mixed-precision modes (BASELINE configs[3]; sdfa_model_set_precision): the same recurrence on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32 cell state and fp32 gate math.
//   TERMS 1: operands rounded to bf16 (8 significand bits).
//   TERMS 3: operands split into hi = bf16(x), lo = bf16(x - hi) (16 bits); a*b = a_lo*b_hi + a_hi*b_lo + a_hi*b_hi.
// Operand image ("K8"): bf16x8[k/8][column] -- one ds_read_b128 / global_load_dwordx4 is one MFMA fragment (lane =
// row/column, lane half = which 8 of the 16 k).  x_f is split while it is staged (octet o = K4 quads 2o, 2o+1);
// h is split by the lane that produced it, which owns rows 32w+8g+4h+{0..3} for g = 0..3 -- so octet 4w+2q+h holds
// hidden units 32w+16q+4h+{0..3} and 32w+16q+8+4h+{0..3}, and the host packs W_hh's K axis in that same order
// (api.cpp: pack_freq_lstm_bf16).  Weights arrive pre-split (hi plane, lo plane) and stream from L2 as before.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void split_octet(const float4 &x0, const float4 &x1, bf16x8 &hi, bf16x8 &lo) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hb = (__bf16)x[e];
        hi[e] = hb;
        lo[e] = (__bf16)(x[e] - (float)hb);
    }
}

// x = hi + mid + lo, three bf16 terms (24 significand bits): the six-product split (SDFA_PREC_BF16X6)
__device__ __forceinline__ void split_octet3(const float4 &x0, const float4 &x1, bf16x8 &hi, bf16x8 &mid, bf16x8 &lo) {
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

// Packed bf16 weights hold BF16_PLANES = 3 planes per direction (hi, mid = bf16(x - hi), lo = bf16(x - hi - mid)); TERMS 1 reads the first,
// TERMS 3 the first two (its "lo" is this mid: same bits as before the third plane existed), the six-product kernels all three.
constexpr int BF16_PLANES = 3;

// ------------------------------------------------------------------ frequency LSTM, six-product split (SDFA_PREC_BF16X6, round 4)
// The recurrence of freq_lstm_bf16_kernel with operands as three bf16 terms and six partial products per product, smallest first:
//   W_lo*b_hi, W_mid*b_mid, W_mid*b_hi, W_hi*b_lo, W_hi*b_mid, W_hi*b_hi        (dropped: the three below 2^-24 of the leading one)
// -- fp32-equivalent products at 16 / 6 = 2.7x the fp32 MFMA rate.  The three weight planes of a k-step are streamed one after the
// other (lo, mid, hi), each requested while the previous one multiplies, so only two planes are live; x_f and h are split into three
// planes by the lanes that stage / produce them.  74 KiB of dynamic LDS, two workgroups per CU.
template <bool SHARED>
__global__ __launch_bounds__(256, 2) void freq_lstm_bf16x6_kernel(FreqLstmArgs a) {
    extern __shared__ bf16x8 sF6[];
    bf16x8 *const sHp = sF6;                                   // [3 planes][16 octets][64 sequences]
    bf16x8 *const sXp = sF6 + 3 * 16 * 64;                     // [3 planes][8 octets][64]: ONE buffer -- the next x tile waits in registers and is
                                                               // written behind the barrier that ends the K loop; 74 KiB in all = two workgroups per CU
    float *const sBias = reinterpret_cast<float *>(sF6 + 3 * 16 * 64 + 3 * 8 * 64);
#define F6_H(pl, o) (sHp + ((pl) * 16 + (o)) * 64)
#define F6_X(buf, pl, o) (sXp + ((pl) * 8 + (o)) * 64)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int dir = (blockIdx.x >> 3) & 1;
    const int64_t m0 = (int64_t)(((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) * 64;
    if (SHARED && m0 >= *a.col_limit) return;

    const float4 *__restrict__ X3 = reinterpret_cast<const float4 *>(a.X3);
    // weights through a buffer descriptor: uniform base + ONE 32-bit lane offset + a scalar offset per (plane, k-step) -- no 64-bit
    // address registers (with lane pointers the kernel spilled and its per-step weight prologue was serialised behind scratch reloads)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long wptr6 = (unsigned long long)(reinterpret_cast<const bf16x8 *>(a.Wb) + (size_t)dir * BF16_PLANES * 24 * 512);
    const unsigned long long wuni6 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wptr6 >> 32)) << 32) |
                                     (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wptr6);
    const __amdgpu_buffer_rsrc_t wrs6 = __builtin_amdgcn_make_buffer_rsrc((void *)wuni6, 0, BF16_PLANES * 24 * 512 * 16, 0x00020000);
    const unsigned woff6 = (unsigned)((wave * 128 + l31 + h * 512) * 16);
#define F6_W(pl, ks, gt) __builtin_bit_cast(bf16x8, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(wrs6, woff6 + (unsigned)(gt) * 512u, (unsigned)(((pl) * 24 + 2 * (ks)) * 512 * 16), 0))
    float4 *__restrict__ HF = reinterpret_cast<float4 *>(a.HF);

    sBias[tid] = a.bias[dir * 512 + tid];
    sBias[256 + tid] = a.bias[dir * 512 + 256 + tid];

    float4 xr0, xr1, xr2, xr3;
#define F6_XLOAD(f)                                                                                \
    {                                                                                              \
        const int o0 = tid >> 6, o1 = 4 + (tid >> 6), col = tid & 63;                              \
        xr0 = X3[(int64_t)((f)*16 + 2 * o0) * a.Mc + m0 + col];                                    \
        xr1 = X3[(int64_t)((f)*16 + 2 * o0 + 1) * a.Mc + m0 + col];                                \
        xr2 = X3[(int64_t)((f)*16 + 2 * o1) * a.Mc + m0 + col];                                    \
        xr3 = X3[(int64_t)((f)*16 + 2 * o1 + 1) * a.Mc + m0 + col];                                \
    }
#define F6_XSTORE(buf)                                                                             \
    {                                                                                              \
        const int o0 = tid >> 6, o1 = 4 + (tid >> 6), col = tid & 63;                              \
        bf16x8 hi, mid, lo;                                                                        \
        split_octet3(xr0, xr1, hi, mid, lo); F6_X(buf, 0, o0)[col] = hi; F6_X(buf, 1, o0)[col] = mid; F6_X(buf, 2, o0)[col] = lo; \
        split_octet3(xr2, xr3, hi, mid, lo); F6_X(buf, 0, o1)[col] = hi; F6_X(buf, 1, o1)[col] = mid; F6_X(buf, 2, o1)[col] = lo; \
    }
    F6_XLOAD(dir ? 31 : 0)
    F6_XSTORE(0)
    __syncthreads();

    f32x16 c[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[j][r] = 0.f;

    for (int s = 0; s < 32; ++s) {
        const int f = dir ? 31 - s : s;
        constexpr int cur = 0;      // one x buffer (the macros keep the two-buffer signature of freq_lstm_bf16_kernel)
        (void)cur;

        f32x16 acc[4][2];
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 b = *reinterpret_cast<const float4 *>(&sBias[wave * 128 + gt * 32 + 8 * g + 4 * h]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[gt][j][4 * g + 0] = b.x; acc[gt][j][4 * g + 1] = b.y;
                    acc[gt][j][4 * g + 2] = b.z; acc[gt][j][4 * g + 3] = b.w;
                }
            }
        const int nks = s > 0 ? 12 : 4;      // k-steps of 16: 0..3 = x_f, 4..11 = h_{s-1} (skipped on the first step)
        // Three register sets, one per weight plane, each refilled two planes ahead of its use, and the order PINNED with scheduling
        // barriers: left to itself the compiler sank every plane's requests to just in front of the MFMAs that consume them (the first
        // build: s_waitcnt right behind the loads, an L2 round trip exposed three times per k-step, 80 ms per step for 50 of MFMAs).
#define F6_SB() __builtin_amdgcn_sched_barrier(0);
        bf16x8 wl[4], wm[4], wh[4];
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) { wl[gt] = F6_W(2, 0, gt); wm[gt] = F6_W(1, 0, gt); }
#pragma unroll 1
        for (int ks = 0; ks < nks; ++ks) {
            bf16x8 b[3][2];
            const int kn = ks + 1 < nks ? ks + 1 : 0;                           // branch-free: the last requests are dropped
            F6_SB()
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) wh[gt] = F6_W(0, ks, gt);            // hi plane of this k-step: needed 24 MFMAs from here
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    b[pl][j] = ks < 4 ? F6_X(cur, pl, 2 * ks + h)[j * 32 + l31] : F6_H(pl, 2 * (ks - 4) + h)[j * 32 + l31];
            F6_SB()
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[gt][j] = MFMA_BF16(wl[gt], b[0][j], acc[gt][j]);      // lo * hi
            F6_SB()
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) wl[gt] = F6_W(2, kn, gt);            // next k-step's lo plane: 40 MFMAs ahead
            F6_SB()
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[gt][j] = MFMA_BF16(wm[gt], b[1][j], acc[gt][j]);      // mid * mid
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[gt][j] = MFMA_BF16(wm[gt], b[0][j], acc[gt][j]);      // mid * hi
            F6_SB()
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) wm[gt] = F6_W(1, kn, gt);            // next k-step's mid plane: 32 MFMAs ahead
            F6_SB()
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[gt][j] = MFMA_BF16(wh[gt], b[2][j], acc[gt][j]);      // hi * lo
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[gt][j] = MFMA_BF16(wh[gt], b[1][j], acc[gt][j]);      // hi * mid
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[gt][j] = MFMA_BF16(wh[gt], b[0][j], acc[gt][j]);      // hi * hi
        }
        F6_SB()
#undef F6_SB
        __syncthreads();   // every wave has finished reading sH / sX[cur]
        if (s + 1 < 32) { F6_XLOAD(dir ? 30 - s : s + 1) }   // lands while the cell update runs
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float4 hq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                lstm_cell_quad<true>(acc[0][j], acc[1][j], acc[2][j], acc[3][j], c[j], g, hq[g]);
                HF[((m0 >> 7) * (int64_t)HF_SLAB_ROWS + (f * 64 + dir * 32 + 8 * wave + 2 * g + h)) * 128 + (m0 & 127) + j * 32 + l31] = hq[g];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bf16x8 hi, mid, lo;
                split_octet3(hq[2 * q], hq[2 * q + 1], hi, mid, lo);
                F6_H(0, 4 * wave + 2 * q + h)[j * 32 + l31] = hi;
                F6_H(1, 4 * wave + 2 * q + h)[j * 32 + l31] = mid;
                F6_H(2, 4 * wave + 2 * q + h)[j * 32 + l31] = lo;
            }
        }
        if (s + 1 < 32) { F6_XSTORE(cur ^ 1) }
        __syncthreads();
    }
#undef F6_XLOAD
#undef F6_XSTORE
#undef F6_W
#undef F6_H
#undef F6_X
}

// ------------------------------------------------------------------ frequency LSTM on split bf16, ONE workgroup per CU (round 4)
// freq_lstm_bf16x6_kernel (PL = 3 operand planes, six products) and freq_lstm_bf16_kernel<3> (PL = 2, three products) rebuilt the way
// freq_lstm_v3_kernel rebuilt the fp32 recurrence: one wave per SIMD that never waits.
//   * x_f and h planes live in ONE LDS image per step parity, [PL planes][8 x octets | 16 h octets][64 columns] (PL 3: 72 KiB, both
//     parities 144 KiB; PL 2: 96 KiB): k-step ks reads octet rows 2ks, 2ks+1 whatever they hold (no select between two arrays), and h
//     being double buffered leaves ONE barrier per step;
//   * tiles come from a queue (a.tile_counter), one persistent workgroup per CU;
//   * a k-step is six (three) groups of eight MFMAs; the weight requests and LDS reads that feed the NEXT k-step sit between the
//     groups, each at least two groups ahead of its use, pinned with scheduling barriers; the last k-step of a step requests
//     k-step 0's planes for the next step.
// Same products in the same order per accumulator as the two-per-CU kernels: bit-identical to them.
template <bool SHARED, bool PERSIST, int PL>
__global__ __launch_bounds__(256, 2) void freq_lstm_bf16p_v3_kernel(FreqLstmArgs a) {
    static_assert(PL == 2 || PL == 3, "two planes (split-bf16, three products) or three (six products)");
    constexpr int ROWS = 24, BT = 64;
    extern __shared__ bf16x8 sP6[];                             // [2 parities][PL][24 octet rows][64]
    float *const sBias = reinterpret_cast<float *>(sP6 + 2 * PL * ROWS * BT);
    int *const sTile = reinterpret_cast<int *>(sBias + 512);
#define P6_ROW(par, pl, r) (sP6 + (((par) * PL + (pl)) * ROWS + (r)) * BT)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const float4 *__restrict__ X3 = reinterpret_cast<const float4 *>(a.X3);
    float4 *__restrict__ HF = reinterpret_cast<float4 *>(a.HF);
    const int n_tiles = (int)(a.Mc / BT) * 2;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const unsigned woff = (unsigned)((wave * 128 + l31 + h * 512) * 16);
    constexpr unsigned W_PLANE = 24u * 512u * 16u, W_KS = 2u * 512u * 16u;      // bytes: one plane, one k-step (two octet rows of 512 gate rows)
#define P6_SB() __builtin_amdgcn_sched_barrier(0);
#define P6_W(so, gt) __builtin_bit_cast(bf16x8, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(wrs, woff + (unsigned)(gt) * 512u, (so), 0))
#define P6_WSET(dst, so) { dst[0] = P6_W(so, 0); dst[1] = P6_W(so, 1); dst[2] = P6_W(so, 2); dst[3] = P6_W(so, 3); }
#define P6_G(wset, bset)                                                                          \
    {                                                                                             \
        acc[0][0] = MFMA_BF16(wset[0], bset[0], acc[0][0]); acc[0][1] = MFMA_BF16(wset[0], bset[1], acc[0][1]); \
        acc[1][0] = MFMA_BF16(wset[1], bset[0], acc[1][0]); acc[1][1] = MFMA_BF16(wset[1], bset[1], acc[1][1]); \
        acc[2][0] = MFMA_BF16(wset[2], bset[0], acc[2][0]); acc[2][1] = MFMA_BF16(wset[2], bset[1], acc[2][1]); \
        acc[3][0] = MFMA_BF16(wset[3], bset[0], acc[3][0]); acc[3][1] = MFMA_BF16(wset[3], bset[1], acc[3][1]); \
    }
    // one k-step: weights of this k-step at byte offset `so`, of the next at `son`; next k-step's operand rows at `bn` (lane pointer
    // into plane 0); B0C / B0N = this / the next k-step's hi operand plane (ping-pong), b1 / b2 = mid / lo planes (refilled in place)
    // the step's very first group: column tile 1 starts from tile 0's seeds (C = acc[gt][0], D = acc[gt][1]) before tile 0's own first
    // product overwrites them in place (freq_lstm_v3_kernel: F3_QSEED) -- only tile 0 is seeded from LDS
#define P6_GSEED(wset, bset)                                                                      \
    {                                                                                             \
        acc[0][1] = MFMA_BF16(wset[0], bset[1], acc[0][0]); acc[0][0] = MFMA_BF16(wset[0], bset[0], acc[0][0]); \
        acc[1][1] = MFMA_BF16(wset[1], bset[1], acc[1][0]); acc[1][0] = MFMA_BF16(wset[1], bset[0], acc[1][0]); \
        acc[2][1] = MFMA_BF16(wset[2], bset[1], acc[2][0]); acc[2][0] = MFMA_BF16(wset[2], bset[0], acc[2][0]); \
        acc[3][1] = MFMA_BF16(wset[3], bset[1], acc[3][0]); acc[3][0] = MFMA_BF16(wset[3], bset[0], acc[3][0]); \
    }
#define P6_KS(B0C, B0N, so, son, bn) P6_KS_(P6_G, B0C, B0N, so, son, bn)
#define P6_KS_SEED(B0C, B0N, so, son, bn) P6_KS_(P6_GSEED, B0C, B0N, so, son, bn)
#define P6_KS_(G0, B0C, B0N, so, son, bn)                                                         \
    {                                                                                             \
        P6_SB() P6_WSET(wh, (so)) P6_SB()                                                         \
        G0(wl, B0C)                                               /* lo  * hi  */                 \
        P6_SB() P6_WSET(wl, 2u * W_PLANE + (son)) P6_SB()                                         \
        P6_G(wm, b1)                                              /* mid * mid */                 \
        P6_G(wm, B0C)                                             /* mid * hi  */                 \
        P6_SB() P6_WSET(wm, W_PLANE + (son)) P6_SB()                                              \
        P6_G(wh, b2)                                              /* hi  * lo  */                 \
        P6_SB() b2[0] = (bn)[2 * ROWS * BT]; b2[1] = (bn)[2 * ROWS * BT + 32]; P6_SB()            \
        P6_G(wh, b1)                                              /* hi  * mid */                 \
        P6_SB() b1[0] = (bn)[ROWS * BT]; b1[1] = (bn)[ROWS * BT + 32]; B0N[0] = (bn)[0]; B0N[1] = (bn)[32]; P6_SB() \
        P6_G(wh, B0C)                                             /* hi  * hi  */                 \
    }
    // PL 2: products hi*hi, hi*lo, lo*hi (the order of freq_lstm_bf16_kernel<3>); the hi weight plane ping-pongs (requested a whole
    // k-step ahead), the lo plane is refilled behind its group (two groups ahead of its next use)
#define P3_KS(WHC, WHN, B0C, B0N, so, son, bn) P3_KS_(P6_G, WHC, WHN, B0C, B0N, so, son, bn)
#define P3_KS_SEED(WHC, WHN, B0C, B0N, so, son, bn) P3_KS_(P6_GSEED, WHC, WHN, B0C, B0N, so, son, bn)
#define P3_KS_(G0, WHC, WHN, B0C, B0N, so, son, bn)                                               \
    {                                                                                             \
        P6_SB() P6_WSET(WHN, (son)) P6_SB()                                                       \
        G0(WHC, B0C)                                              /* hi * hi */                   \
        P6_SB() B0N[0] = (bn)[0]; B0N[1] = (bn)[32]; P6_SB()                                      \
        P6_G(WHC, b1)                                             /* hi * lo */                   \
        P6_SB() b1[0] = (bn)[ROWS * BT]; b1[1] = (bn)[ROWS * BT + 32]; P6_SB()                    \
        P6_G(wl, B0C)                                             /* lo * hi */                   \
        P6_SB() P6_WSET(wl, W_PLANE + (son)) P6_SB()                                              \
    }
  for (;;) {      // PERSIST: one pass per tile taken from the queue; otherwise a single pass
    int dir;
    int64_t m0;
    if (PERSIST) {
        if (tid == 0) *sTile = atomicAdd(a.tile_counter, 1);
        __syncthreads();                       // (also: every wave has left the previous tile's last step)
        const int t = __builtin_amdgcn_readfirstlane(*sTile);
        if (t >= n_tiles) break;               // queue empty: every workgroup gets here
        dir = t & 1;
        m0 = (int64_t)(t >> 1) * BT;
        if (SHARED && m0 >= *a.col_limit) break;
    } else {
        dir = (blockIdx.x >> 3) & 1;
        m0 = (int64_t)(((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) * BT;
        if (SHARED && m0 >= *a.col_limit) return;
    }
    const unsigned long long wptr = (unsigned long long)(reinterpret_cast<const bf16x8 *>(a.Wb) + (size_t)dir * BF16_PLANES * 24 * 512);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wptr, 0, BF16_PLANES * 24 * 512 * 16, 0x00020000);
    sBias[tid] = a.bias[dir * 512 + tid];
    sBias[256 + tid] = a.bias[dir * 512 + 256 + tid];

    // x_f tile of a step: 8 octets of 64 columns; thread (wave, lane) stages octets wave and 4 + wave of column lane
    const float4 *__restrict__ xsrc = X3 + (int64_t)(2 * wave) * a.Mc + m0 + lane;
    float4 xr0, xr1, xr2, xr3;
#define P6_XLOAD(f)                                                                               \
    {                                                                                             \
        const float4 *xs = xsrc + (int64_t)((f) * 16) * a.Mc;                                     \
        xr0 = xs[0]; xr1 = xs[a.Mc]; xr2 = xs[8 * a.Mc]; xr3 = xs[9 * a.Mc];                      \
    }
#define P6_SPLIT_STORE(par, row, col, q0, q1)                                                     \
    {                                                                                             \
        bf16x8 hi, mid, lo;                                                                       \
        if constexpr (PL == 3) {                                                                  \
            split_octet3(q0, q1, hi, mid, lo);                                                    \
            P6_ROW(par, 0, row)[col] = hi; P6_ROW(par, 1, row)[col] = mid; P6_ROW(par, 2, row)[col] = lo; \
        } else {                                                                                  \
            split_octet(q0, q1, hi, lo);                                                          \
            P6_ROW(par, 0, row)[col] = hi; P6_ROW(par, 1, row)[col] = lo;                         \
        }                                                                                         \
    }
#define P6_XSTORE(par) { P6_SPLIT_STORE(par, wave, lane, xr0, xr1) P6_SPLIT_STORE(par, 4 + wave, lane, xr2, xr3) }
    P6_XLOAD(dir ? 31 : 0)
    bf16x8 wl[4], wm[4], wh[4];                             // PL 3: lo, mid, hi planes; PL 2: wl = lo plane, wh / wm = the hi plane's ping-pong pair
    if constexpr (PL == 3) { P6_WSET(wl, 2u * W_PLANE) P6_WSET(wm, W_PLANE) }     // k-step 0, lo and mid planes
    else { P6_WSET(wh, 0u) P6_WSET(wl, W_PLANE) }                                 // k-step 0, hi and lo planes
    P6_XSTORE(0)

    f32x16 c[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
    __syncthreads();   // bias and the first x tile are in LDS
#ifdef SDFA_STAMPS
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, v_init = 0, v_k = 0, v_ep = 0, v_b2 = 0;
#endif

    for (int s = 0; s < 32; ++s) {
        const int f = dir ? 31 - s : s;
        const int cur = s & 1;
        f32x16 acc[4][2];
        LSTAMP(t0)
        // k-step 0's operand planes first, then the seeds (the bias) of column tile 0 only: freq_lstm_v3_kernel has the reasons
        const int trips = s > 0 ? 3 : 1;       // h_{-1} = 0: the first step contracts x_f only (4 of the 12 k-steps)
        const bf16x8 *brow = P6_ROW(cur, 0, h) + l31;            // operand rows of k-step ks: brow + ks * 2 * BT (+ plane * ROWS * BT, + 32 for the second column tile)
        bf16x8 b0a[2], b0b[2], b1[2], b2[2];
        b0a[0] = brow[0]; b0a[1] = brow[32];
        b1[0] = brow[ROWS * BT]; b1[1] = brow[ROWS * BT + 32];
        if constexpr (PL == 3) { b2[0] = brow[2 * ROWS * BT]; b2[1] = brow[2 * ROWS * BT + 32]; }
        P6_SB()
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *reinterpret_cast<const float4 *>(&sBias[wave * 128 + gt * 32 + 8 * g + 4 * h]);
                acc[gt][0][4 * g + 0] = b.x; acc[gt][0][4 * g + 1] = b.y;
                acc[gt][0][4 * g + 2] = b.z; acc[gt][0][4 * g + 3] = b.w;
            }
        LSTAMP(t1)
        // a trip = 4 k-steps; trip 0, with the seeded first group, is written out in front of the loop (straight-line code: exact LDS waits)
#define P6_TRIP(KS6_FIRST, KS3_FIRST, t_)                                                                                         \
        {                                                                                                                         \
            const bf16x8 *bt = brow + (t_) * 4 * 2 * BT;                                                                          \
            const bf16x8 *bn = (t_) + 1 < trips ? bt + 4 * 2 * BT : brow;       /* behind the last trip: k-step 0 again (operands dropped) */ \
            const unsigned so = (unsigned)(t_) * (4u * W_KS);                                                                     \
            const unsigned son = (t_) + 1 < trips ? so + 4u * W_KS : 0u;        /* ... whose weights ARE k-step 0 of the next step */ \
            if constexpr (PL == 3) {                                                                                              \
                KS6_FIRST(b0a, b0b, so, so + 1u * W_KS, bt + 1 * 2 * BT)                                                          \
                P6_KS(b0b, b0a, so + 1u * W_KS, so + 2u * W_KS, bt + 2 * 2 * BT)                                                  \
                P6_KS(b0a, b0b, so + 2u * W_KS, so + 3u * W_KS, bt + 3 * 2 * BT)                                                  \
                P6_KS(b0b, b0a, so + 3u * W_KS, son, bn)                                                                          \
            } else {                                                                                                              \
                KS3_FIRST(wh, wm, b0a, b0b, so, so + 1u * W_KS, bt + 1 * 2 * BT)                                                  \
                P3_KS(wm, wh, b0b, b0a, so + 1u * W_KS, so + 2u * W_KS, bt + 2 * 2 * BT)                                          \
                P3_KS(wh, wm, b0a, b0b, so + 2u * W_KS, so + 3u * W_KS, bt + 3 * 2 * BT)                                          \
                P3_KS(wm, wh, b0b, b0a, so + 3u * W_KS, son, bn)                                                                  \
            }                                                                                                                     \
        }
        P6_TRIP(P6_KS_SEED, P3_KS_SEED, 0)
#pragma unroll 1
        for (int t = 1; t < trips; ++t) P6_TRIP(P6_KS, P3_KS, t)
#undef P6_TRIP
        P6_SB()
#ifdef SDFA_STAMPS
        asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]), "v"(acc[0][1]), "v"(acc[1][1]), "v"(acc[2][1]), "v"(acc[3][1]));   // all MFMAs done
#endif
        LSTAMP(t2)
        // the next x tile: requested BEHIND the step's last weight requests (loads return in issue order), split and written to the
        // other parity after the cell update, which covers the round trip
        P6_XLOAD(s + 1 < 32 ? (dir ? 30 - s : s + 1) : f)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float4 hq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                lstm_cell_quad<true>(acc[0][j], acc[1][j], acc[2][j], acc[3][j], c[j], g, hq[g]);
                HF[((m0 >> 7) * (int64_t)HF_SLAB_ROWS + (f * 64 + dir * 32 + 8 * wave + 2 * g + h)) * 128 + (m0 & 127) + j * 32 + l31] = hq[g];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) P6_SPLIT_STORE(cur ^ 1, 8 + 4 * wave + 2 * q + h, j * 32 + l31, hq[2 * q], hq[2 * q + 1])
        }
        P6_XSTORE(cur ^ 1)
        LSTAMP(t3)
        // h_s and the next x tile must be in LDS before anyone starts step s+1 (the hidden-state stores need not be acknowledged)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef SDFA_STAMPS
        LSTAMP(t4)
        if (s > 0) { v_init += t1 - t0; v_k += t2 - t1; v_ep += t3 - t2; v_b2 += t4 - t3; }
#endif
    }
#ifdef SDFA_STAMPS
    if (lane == 0) {
        atomicAdd(&g_lstamp[0], v_init); atomicAdd(&g_lstamp[1], v_k); atomicAdd(&g_lstamp[3], v_ep); atomicAdd(&g_lstamp[4], v_b2); atomicAdd(&g_lstamp[6], 31ull);
    }
#endif
    if (!PERSIST) break;
  }
#undef P6_ROW
#undef P6_SB
#undef P6_W
#undef P6_WSET
#undef P6_G
#undef P6_KS
#undef P6_KS_
#undef P6_KS_SEED
#undef P3_KS_
#undef P3_KS_SEED
#undef P6_GSEED
#undef P3_KS
#undef P6_SPLIT_STORE
#undef P6_XLOAD
#undef P6_XSTORE
}

template <bool SHARED, int TERMS>
__global__ __launch_bounds__(256, 2) void freq_lstm_bf16_kernel(FreqLstmArgs a) {
    constexpr bool LO = TERMS > 1;
    constexpr int NPL = LO ? 2 : 1;              // operand planes
    __shared__ bf16x8 sH[NPL][16][64];           // h_{s-1}: 128 hidden as 16 octets x 64 sequences, per plane
    __shared__ bf16x8 sX[2][NPL][8][64];         // x_f: 64 features as 8 octets, double buffered
    __shared__ float sBias[512];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int dir = (blockIdx.x >> 3) & 1;      // both directions of a column tile on one XCD (see freq_lstm_kernel)
    const int64_t m0 = (int64_t)(((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)) * 64;
    if (SHARED && m0 >= *a.col_limit) return;

    const float4 *__restrict__ X3 = reinterpret_cast<const float4 *>(a.X3);
    // per direction: [plane hi | lo][24 octets][512 gate rows]
    const bf16x8 *__restrict__ Wh = reinterpret_cast<const bf16x8 *>(a.Wb) + (size_t)dir * BF16_PLANES * 24 * 512 + wave * 128 + l31;
    const bf16x8 *__restrict__ Wl = Wh + 24 * 512;
    float4 *__restrict__ HF = reinterpret_cast<float4 *>(a.HF);

    sBias[tid] = a.bias[dir * 512 + tid];
    sBias[256 + tid] = a.bias[dir * 512 + 256 + tid];

    // staging of x_f: thread -> (octet, column) pairs p = i*256 + tid, i = 0, 1
    float4 xr0, xr1, xr2, xr3;
#define BXLOAD(f)                                                                                  \
    {                                                                                              \
        const int o0 = tid >> 6, o1 = 4 + (tid >> 6), col = tid & 63;                              \
        xr0 = X3[(int64_t)((f)*16 + 2 * o0) * a.Mc + m0 + col];                                    \
        xr1 = X3[(int64_t)((f)*16 + 2 * o0 + 1) * a.Mc + m0 + col];                                \
        xr2 = X3[(int64_t)((f)*16 + 2 * o1) * a.Mc + m0 + col];                                    \
        xr3 = X3[(int64_t)((f)*16 + 2 * o1 + 1) * a.Mc + m0 + col];                                \
    }
#define BXSTORE(buf)                                                                               \
    {                                                                                              \
        const int o0 = tid >> 6, o1 = 4 + (tid >> 6), col = tid & 63;                              \
        bf16x8 hi, lo;                                                                             \
        split_octet(xr0, xr1, hi, lo); sX[buf][0][o0][col] = hi; if (LO) sX[buf][NPL - 1][o0][col] = lo; \
        split_octet(xr2, xr3, hi, lo); sX[buf][0][o1][col] = hi; if (LO) sX[buf][NPL - 1][o1][col] = lo; \
    }
    BXLOAD(dir ? 31 : 0)
    BXSTORE(0)
    __syncthreads();

    f32x16 c[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[j][r] = 0.f;

    for (int s = 0; s < 32; ++s) {
        const int f = dir ? 31 - s : s;
        const int cur = s & 1;

        f32x16 acc[4][2];
#pragma unroll
        for (int gt = 0; gt < 4; ++gt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 b = *reinterpret_cast<const float4 *>(&sBias[wave * 128 + gt * 32 + 8 * g + 4 * h]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[gt][j][4 * g + 0] = b.x; acc[gt][j][4 * g + 1] = b.y;
                    acc[gt][j][4 * g + 2] = b.z; acc[gt][j][4 * g + 3] = b.w;
                }
            }
        // k-steps of 16: 0..3 = x_f (octets 0..7 of sX), 4..11 = h_{s-1} (octets 0..15 of sH; skipped on the first step).
        // Weight fragments of k-step ks+1 are requested before the MFMAs of ks are issued.
        const int nks = s > 0 ? 12 : 4;
        // The hi plane is requested one k-step ahead; the lo plane at the top of its own k-step -- it is only needed by
        // the last third of the step's MFMAs (order hi*hi, hi*lo, lo*hi), which keeps the kernel at 256 VGPRs.
        bf16x8 whn[4];
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) whn[gt] = Wh[h * 512 + gt * 32];
#pragma unroll 1
        for (int ks = 0; ks < nks; ++ks) {
            bf16x8 wh[4], wl[4], bh[2], bl[2];
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) {
                wh[gt] = whn[gt];
                if (LO) wl[gt] = Wl[(2 * ks + h) * 512 + gt * 32];
            }
            if (ks + 1 < nks) {
#pragma unroll
                for (int gt = 0; gt < 4; ++gt) whn[gt] = Wh[(2 * (ks + 1) + h) * 512 + gt * 32];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (ks < 4) {
                    bh[j] = sX[cur][0][2 * ks + h][j * 32 + l31];
                    if (LO) bl[j] = sX[cur][NPL - 1][2 * ks + h][j * 32 + l31];
                } else {
                    bh[j] = sH[0][2 * (ks - 4) + h][j * 32 + l31];
                    if (LO) bl[j] = sH[NPL - 1][2 * (ks - 4) + h][j * 32 + l31];
                }
            }
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[gt][j] = MFMA_BF16(wh[gt], bh[j], acc[gt][j]);
            if (LO) {
#pragma unroll
                for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[gt][j] = MFMA_BF16(wh[gt], bl[j], acc[gt][j]);
#pragma unroll
                for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[gt][j] = MFMA_BF16(wl[gt], bh[j], acc[gt][j]);
            }
        }
        __syncthreads();   // every wave has finished reading sH / sX[cur]
        if (s + 1 < 32) { BXLOAD(dir ? 30 - s : s + 1) }   // lands while the cell update runs
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float4 hq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                lstm_cell_quad<true>(acc[0][j], acc[1][j], acc[2][j], acc[3][j], c[j], g, hq[g]);
                HF[((m0 >> 7) * (int64_t)HF_SLAB_ROWS + (f * 64 + dir * 32 + 8 * wave + 2 * g + h)) * 128 + (m0 & 127) + j * 32 + l31] = hq[g];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bf16x8 hi, lo;
                split_octet(hq[2 * q], hq[2 * q + 1], hi, lo);
                sH[0][4 * wave + 2 * q + h][j * 32 + l31] = hi;
                if (LO) sH[NPL - 1][4 * wave + 2 * q + h][j * 32 + l31] = lo;
            }
        }
        if (s + 1 < 32) { BXSTORE(cur ^ 1) }
        __syncthreads();
    }
#undef BXLOAD
#undef BXSTORE
}

// ----------------------------------------------------------------------------------------- time LSTM
// NT = 32-frame column tiles per workgroup: 2 (64 frames, the throughput shape) or 1 (32 frames: twice the
// workgroups, used when a chunk would otherwise leave CUs idle).
// MAP (column sharing, layer 0): GX holds the input projection of the DISTINCT columns only (a.col_map[t * Nc + n] = its column), so a
// lane's 32 quads of a step come from column u = col_map[...] instead of t * Nc + n.  Within a clip the frames of a tile map to one
// run of consecutive u (they all shift by the same 12 frames / 25 hops), so the requests stay coalesced; u is fetched one step
// further ahead than the quads it addresses, behind that step's requests, and costs no wait of its own.
template <int NT, bool MAP = false>
__device__ __forceinline__ void time_lstm_body(const TimeLstmArgs &a) {
    extern __shared__ float4 sHt[];   // [2][64 k-quads][32*NT sequences]
    constexpr int BT = 32 * NT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // 8 waves: hidden block of 32
    const int l31 = lane & 31, h = lane >> 5;
    const int dir = blockIdx.x & 1;
    const int64_t n0 = (int64_t)(blockIdx.x >> 1) * BT;

    const float4 *__restrict__ GX = reinterpret_cast<const float4 *>(a.GX);
    float4 *__restrict__ H = reinterpret_cast<float4 *>(a.H);

    f32x16 c[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[j][r] = 0.f;

    // The accumulators start every step from the input projection GX (the GEMM's output).  Those 32 quads per lane
    // are requested one step AHEAD, quad by quad, as the cell update releases the registers -- requested at the top of
    // the step they would put an HBM round trip in front of every step's first MFMA (all 8 waves wait in lock-step).
    f32x16 acc[4][NT];
    // per-lane base + wave-uniform offsets (scalar registers) instead of 32 lane addresses the optimiser would hoist out of
    // the step loop and spill (the kernel sits at the 256-register limit)
    const float4 *__restrict__ GXl = GX + (int64_t)(dir * 256 + wave * 32 + h) * a.Mc + n0 + l31;
    const float4 *__restrict__ GXr = GX + (int64_t)(dir * 256 + wave * 32 + h) * a.Mc;                    // MAP: row base, the column comes from the map
    const int32_t *__restrict__ cmap = MAP ? a.col_map + n0 + l31 : nullptr;
    int ucol[NT];                                                                                        // MAP: columns of the NEXT request
    if (MAP) {
#pragma unroll
        for (int j = 0; j < NT; ++j) ucol[j] = cmap[(int64_t)(dir ? 63 : 0) * a.Nc + j * 32];
    }
    float4 *__restrict__ Hl = H + (int64_t)(dir * 64 + wave * 8 + h) * a.Mc + n0 + l31;
#define TL_GX(t_, gt, g, j) (MAP ? GXr[(int64_t)((gt) * 8 + 2 * (g)) * a.Mc + ucol[j]] : GXl[(int64_t)((gt) * 8 + 2 * (g)) * a.Mc + (int64_t)(t_) * a.Nc + (j) * 32])
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float4 v = TL_GX(dir ? 63 : 0, gt, g, j);
                acc[gt][j][4 * g + 0] = v.x; acc[gt][j][4 * g + 1] = v.y; acc[gt][j][4 * g + 2] = v.z; acc[gt][j][4 * g + 3] = v.w;
            }

    // Recurrent weights one k-block ahead, two alternating operand sets, branch-free (see freq_lstm_v2_kernel): named
    // scalars and a buffer descriptor (wave-uniform base in scalar registers + one 32-bit lane offset), so the K loop
    // carries no copies and no 64-bit address registers.  The request that wraps around at the end of a step's K loop IS
    // k-block 0 of the next step (weights do not change): it goes into `wn`, which stays live across the cell update and
    // the step barrier, so the first MFMAs after the barrier (all 8 waves start together) do not wait for L2.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long wptr = (unsigned long long)(reinterpret_cast<const float4 *>(a.W) + (size_t)dir * 64 * 1024 + wave * 128);
    const unsigned long long wuni = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wptr >> 32)) << 32) |
                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wptr);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wuni, 0, 64 * 1024 * 16, 0x00020000);
    const unsigned woff = (unsigned)((l31 + h * 1024) * 16);
#define TL_W1(so, g_) __builtin_bit_cast(float4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(wrs, woff + 512 * (g_), so, 0))
#define TL_LOAD(kb, W0, W1, W2, W3) { const unsigned so = (unsigned)(kb) * (2048 * 16); W0 = TL_W1(so, 0); W1 = TL_W1(so, 1); W2 = TL_W1(so, 2); W3 = TL_W1(so, 3); }
    float4 wn0, wn1, wn2, wn3;
    TL_LOAD(0, wn0, wn1, wn2, wn3)
    if (MAP) {      // columns of step 1's request (behind step 0's quads, in front of the first weights)
#pragma unroll
        for (int j = 0; j < NT; ++j) ucol[j] = cmap[(int64_t)(dir ? 62 : 1) * a.Nc + j * 32];
    }
#if defined(SDFA_STAMPS) && defined(__HIP_DEVICE_COMPILE__)   /* (the host pass of a __device__ template rejects the asm constraints) */
#define TSTAMP(t) LSTAMP(t)
#else
#define TSTAMP(t)
#endif
#if defined(SDFA_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    unsigned long long tt0 = 0, tt1 = 0, tt2 = 0, tt3 = 0, tv_k = 0, tv_cell = 0, tv_bar = 0;
#endif
    for (int s = 0; s < 64; ++s) {
        const int t = dir ? 63 - s : s;
        const int tn = dir ? t - 1 : t + 1;
        const int64_t tcol = (int64_t)t * a.Nc;
        const float4 *sHc = sHt + (size_t)(s & 1) * 64 * BT;
        float4 *sHn = sHt + (size_t)((s & 1) ^ 1) * 64 * BT;
        TSTAMP(tt0)

        if (s > 0) {
            float4 wa0 = wn0, wa1 = wn1, wa2 = wn2, wa3 = wn3, wb0, wb1, wb2, wb3, ba[NT], bb[NT];
            // K loop in trips of 8 k-blocks with immediate offsets, the memory instructions that refill the other operand set
            // one or two at a time in front of the four MFMA groups of a k-block (freq_lstm_v3_kernel)
#define TL_SB() __builtin_amdgcn_sched_barrier(0);
#define TL_Q(W0, W1, W2, W3, B, q)                                                                               \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[0][j] = MFMA(SDFA_OP(f4c(W0, q)), SDFA_OP(f4c(B[j], q)), acc[0][j]); \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[1][j] = MFMA(SDFA_OP(f4c(W1, q)), SDFA_OP(f4c(B[j], q)), acc[1][j]); \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[2][j] = MFMA(SDFA_OP(f4c(W2, q)), SDFA_OP(f4c(B[j], q)), acc[2][j]); \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[3][j] = MFMA(SDFA_OP(f4c(W3, q)), SDFA_OP(f4c(B[j], q)), acc[3][j]);
#define TL_KB(CW0, CW1, CW2, CW3, CB, NW0, NW1, NW2, NW3, NB, so, bp)                                            \
    {                                                                                                            \
        TL_SB() NW0 = TL_W1(so, 0); NW1 = TL_W1(so, 1); TL_SB()                                                  \
        TL_Q(CW0, CW1, CW2, CW3, CB, 0)                                                                          \
        TL_SB() NW2 = TL_W1(so, 2); NW3 = TL_W1(so, 3); TL_SB()                                                  \
        TL_Q(CW0, CW1, CW2, CW3, CB, 1)                                                                          \
        TL_SB() NB[0] = (bp)[0]; TL_SB()                                                                         \
        TL_Q(CW0, CW1, CW2, CW3, CB, 2)                                                                          \
        TL_SB() if (NT > 1) NB[NT - 1] = (bp)[32 * (NT - 1)]; TL_SB()                                            \
        TL_Q(CW0, CW1, CW2, CW3, CB, 3)                                                                          \
    }
            const float4 *brow = sHc + h * BT + l31;          // row pair of k-block kb: brow + kb * 2 * BT
#pragma unroll
            for (int j = 0; j < NT; ++j) ba[j] = brow[j * 32];
#pragma unroll 1
            for (int t = 0; t < 4; ++t) {
                const float4 *bt = brow + t * 16 * BT;
                const float4 *bn = t + 1 < 4 ? bt + 16 * BT : brow;         // behind the last trip: k-block 0 again (dropped) ...
                const unsigned so = (unsigned)t * (8 * 2048 * 16);
                const unsigned son = t + 1 < 4 ? so + 8 * 2048 * 16 : 0u;   // ... whose weights ARE k-block 0 of the next step
                TL_KB(wa0, wa1, wa2, wa3, ba, wb0, wb1, wb2, wb3, bb, so + 1 * 2048 * 16, bt + 1 * 2 * BT)
                TL_KB(wb0, wb1, wb2, wb3, bb, wa0, wa1, wa2, wa3, ba, so + 2 * 2048 * 16, bt + 2 * 2 * BT)
                TL_KB(wa0, wa1, wa2, wa3, ba, wb0, wb1, wb2, wb3, bb, so + 3 * 2048 * 16, bt + 3 * 2 * BT)
                TL_KB(wb0, wb1, wb2, wb3, bb, wa0, wa1, wa2, wa3, ba, so + 4 * 2048 * 16, bt + 4 * 2 * BT)
                TL_KB(wa0, wa1, wa2, wa3, ba, wb0, wb1, wb2, wb3, bb, so + 5 * 2048 * 16, bt + 5 * 2 * BT)
                TL_KB(wb0, wb1, wb2, wb3, bb, wa0, wa1, wa2, wa3, ba, so + 6 * 2048 * 16, bt + 6 * 2 * BT)
                TL_KB(wa0, wa1, wa2, wa3, ba, wb0, wb1, wb2, wb3, bb, so + 7 * 2048 * 16, bt + 7 * 2 * BT)
                TL_KB(wb0, wb1, wb2, wb3, bb, wa0, wa1, wa2, wa3, ba, son, bn)
            }
            TL_SB()
#undef TL_KB
#undef TL_Q
#undef TL_SB
            wn0 = wa0; wn1 = wa1; wn2 = wa2; wn3 = wa3;      // k-block 0 again: the next step's first operands
        }
#if defined(SDFA_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
        if (NT == 2) asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]), "v"(acc[0][NT - 1]), "v"(acc[1][NT - 1]), "v"(acc[2][NT - 1]), "v"(acc[3][NT - 1]));   // all MFMAs done
#endif
        TSTAMP(tt1)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 hq;
                lstm_cell_quad(acc[0][j], acc[1][j], acc[2][j], acc[3][j], c[j], g, hq);
                const int hq_idx = 8 * wave + 2 * g + h;
                sHn[hq_idx * BT + j * 32 + l31] = hq;
                Hl[(int64_t)(2 * g) * a.Mc + tcol + j * 32] = hq;
                if (s + 1 < 64) {   // this quad's gate registers are free: request the next step's input projection into them
#pragma unroll
                    for (int gt = 0; gt < 4; ++gt) {
                        const float4 v = TL_GX(tn, gt, g, j);
                        acc[gt][j][4 * g + 0] = v.x; acc[gt][j][4 * g + 1] = v.y; acc[gt][j][4 * g + 2] = v.z; acc[gt][j][4 * g + 3] = v.w;
                    }
                }
            }
        if (MAP && s + 2 < 64) {   // behind this step's requests: the columns of the step after next
            const int t2 = dir ? t - 2 : t + 2;
#pragma unroll
            for (int j = 0; j < NT; ++j) ucol[j] = cmap[(int64_t)t2 * a.Nc + j * 32];
        }
        TSTAMP(tt2)
        __syncthreads();   // h_s complete in sHn before anyone reads it; sHc free for step s+1's writes
#if defined(SDFA_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
        TSTAMP(tt3)
        if (s > 0) { tv_k += tt1 - tt0; tv_cell += tt2 - tt1; tv_bar += tt3 - tt2; }
#endif
    }
#if defined(SDFA_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    if (NT == 2 && lane == 0) { atomicAdd(&g_lsub[0], tv_k); atomicAdd(&g_lsub[1], tv_cell); atomicAdd(&g_lsub[2], tv_bar); atomicAdd(&g_lsub[3], 63ull); }
#endif
#undef TL_GX
#undef TSTAMP
#undef TL_LOAD
#undef TL_W1
}

template <int NT, bool MAP = false>
__global__ __launch_bounds__(512, 2) void time_lstm_kernel(TimeLstmArgs a) { time_lstm_body<NT, MAP>(a); }

// The same recurrence as a REPAIR pass behind a launch of the cooperating-workgroup kernels below: every workgroup reads that launch's
// time-out word and exits at once unless a workgroup of it gave up waiting for its partner -- in which case this pass, which needs no
// co-residency and cannot time out, recomputes the layer's H rows from the (untouched) input projections.  So a time-out costs time,
// never a wrong row, and nothing has to travel to the host to decide it.  (3 - 4 us per layer when there is nothing to repair.)
template <bool MAP>
__global__ __launch_bounds__(512, 2) void time_lstm_repair_kernel(TimeLstmArgs a, const unsigned *timed_out) {
    if (__hip_atomic_load(timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    time_lstm_body<1, MAP>(a);
}

// ---------------------------------------------------------------------------- time LSTM, small batches
// A single utterance (156 - 640 frames) gives time_lstm_kernel<1> only 10 - 40 workgroups, each alone on its CU for 64 sequential
// steps of 16.8 MFLOP: two waves per SIMD, 27 us per step, 1.9 ms per layer whatever the clip length -- the largest stage of a
// single-clip call.  Here the 1024 gate rows of a 32-frame tile are SPLIT over G = 2 workgroups on two CUs (four waves each, ONE per
// SIMD, a wave owning one block of 32 hidden units x 4 gates exactly as in time_lstm_kernel), so a step's matrix work takes half the
// time (13.7 us), and the two workgroups exchange their halves of h_t through global memory every step (+ 3 us):
//   * a workgroup's slice of h_t goes to LDS (its own next-step operand) and, as before, to the H output rows -- here with
//     WRITE-THROUGH (sc1) stores, which is what makes the H rows themselves the hand-off buffer;
//   * every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane publishes the step
//     number in the workgroup's flag word (relaxed agent-scope atomic store);
//   * wave 0 polls the partner's flag word (relaxed, s_sleep between polls, BOUNDED in wall-clock time: on expiry the launch's
//     time-out word is set and the kernel runs on with whatever it has -- it must never hang; time_lstm_repair_kernel, launched
//     behind every launch of this kernel, then redoes the layer with the single-workgroup recurrence, so a time-out never reaches
//     the caller as a wrong row), the workgroup meets again, and every thread fetches its share
//     of the partner's slice with sc1 loads (they bypass this CU's L1, which another CU's stores never refresh) into LDS.
// That is the publish / consume form MI355X_MICROARCH.md (visibility, "Valid forms", first table row) lists as measured for
// hipMalloc'ed memory with one workgroup per CU (the launch asks for 96 KiB of LDS so that two never share one): no agent-scope
// fence on either side.  The "time_lstm_handoff" option switches either side to the always-valid form (plain stores + agent release,
// agent acquire + plain loads: 4 - 15 % slower, same bits).  Results do not depend on placement; the workgroups of a tile are
// given block ids 8 apart only because such blocks were observed to share an XCD (its L2 then serves the exchange).  All workgroups
// should be resident together for the form to pay: the launcher uses this kernel only while the grid fits the CUs it may use (correctness
// does not depend on it: partners are 8 block ids apart and workgroups are dispatched in block order, so whatever else occupies the device
// the resident pairs finish and make room; and a wait that does expire is repaired, see above).  Same k order and cell arithmetic as
// time_lstm_kernel: bit-identical.  (The template also instantiates for G = 4 -- two waves per workgroup -- which measured no
// faster: a wave's matrix work per step does not change.  profiles/r03_time_lstm_split.txt)
struct SplitCtl {
    unsigned *flags;       // one word per workgroup, zeroed by the launcher: the last step whose slice of h the workgroup has published
    unsigned *timed_out;   // this LAUNCH's time-out word (zeroed by the launcher; read by time_lstm_repair_kernel behind it)
    unsigned *status;      // word 0 of the workspace's status block: time-outs since the block was zeroed (only ever incremented)
    unsigned ticks;        // bound of one wait, in ticks of the 100 MHz wall clock (s_memrealtime)
    int mode;              // bit 0 / 1: hand-off form ("time_lstm_handoff"); bit 2 (tests only): part 1 never publishes
};

// Bounded wait of wave 0 for its partners' flag words.  The bound is WALL-CLOCK time (read every 256th poll, so the common case -- the
// partner is a few hundred nanoseconds behind -- pays nothing for it); on expiry the launch's time-out word and the workspace's
// counter are set and the workgroup stops waiting for good: its rows are wrong from here on and time_lstm_repair_kernel redoes the layer.
__device__ __forceinline__ bool split_wait(const unsigned *pf, unsigned want, const SplitCtl &ctl, int lane) {
    unsigned spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
        const unsigned v = __hip_atomic_load(pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((int)(v >= want))) return false;
        if ((++spins & 255u) == 0u) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = now;
            else if (now - t0 > (unsigned long long)ctl.ticks) {
                if (lane == 0) {
                    __hip_atomic_store(ctl.timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (ctl.status) __hip_atomic_fetch_add(ctl.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return true;
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

template <int G, bool MAP = false>
__global__ __launch_bounds__(512 / G) void time_lstm_split_kernel(TimeLstmArgs a, SplitCtl ctl) {
    unsigned *const flags = ctl.flags;
    const int mode = ctl.mode;
    extern __shared__ float4 sHs[];   // [2][64 k-quads][32 sequences]
    constexpr int BT = 32, NW = 8 / G, NTHR = 64 * NW;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int q8 = blockIdx.x >> 3;
    const int part = q8 % G, td = (q8 / G) * 8 + (blockIdx.x & 7);      // (tile, direction) index; parts of one tile: block ids 8 apart
    const int dir = td & 1, hb = part * NW + wave;                      // hb: hidden block of 32 units this wave owns
    const int64_t n0 = (int64_t)(td >> 1) * BT;

    const float4 *__restrict__ GX = reinterpret_cast<const float4 *>(a.GX);
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    f32x16 acc[4][1];
    const float4 *__restrict__ GXl = GX + (int64_t)(dir * 256 + hb * 32 + h) * a.Mc + n0 + l31;
    const float4 *__restrict__ GXr = GX + (int64_t)(dir * 256 + hb * 32 + h) * a.Mc;      // MAP (column sharing): see time_lstm_body
    const int32_t *__restrict__ cmap = MAP ? a.col_map + n0 + l31 : nullptr;
    int ucol = MAP ? cmap[(int64_t)(dir ? 63 : 0) * a.Nc] : 0;
#define TS_GX(t_, gt, g) (MAP ? GXr[(int64_t)((gt) * 8 + 2 * (g)) * a.Mc + ucol] : GXl[(int64_t)((gt) * 8 + 2 * (g)) * a.Mc + (int64_t)(t_) * a.Nc])
#define TS_GX_ALL(t_)                                                                                               \
    _Pragma("unroll") for (int gt = 0; gt < 4; ++gt) _Pragma("unroll") for (int g = 0; g < 4; ++g) {                  \
        const float4 v = TS_GX(t_, gt, g);                                                                          \
        acc[gt][0][4 * g + 0] = v.x; acc[gt][0][4 * g + 1] = v.y; acc[gt][0][4 * g + 2] = v.z; acc[gt][0][4 * g + 3] = v.w; \
    }
    TS_GX_ALL(dir ? 63 : 0)
    if (MAP) ucol = cmap[(int64_t)(dir ? 62 : 1) * a.Nc];

    // H rows of this direction as a buffer: write-through stores of the own slice, sc1 loads of the partners' slices
    const unsigned long long hptr = (unsigned long long)(reinterpret_cast<float4 *>(a.H) + (int64_t)dir * 64 * a.Mc);
    const unsigned long long huni = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(hptr >> 32)) << 32) |
                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)hptr);
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void *)huni, 0, 0x7fffffff, 0x00020000);
    const unsigned ldm = (unsigned)a.Mc * 16u;                           // bytes per k-quad row (launcher: 64 rows stay below 2 GB)

    const unsigned long long wptr = (unsigned long long)(reinterpret_cast<const float4 *>(a.W) + (size_t)dir * 64 * 1024 + hb * 128);
    const unsigned long long wuni = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wptr >> 32)) << 32) |
                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wptr);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wuni, 0, 64 * 1024 * 16, 0x00020000);
    const unsigned woff = (unsigned)((l31 + h * 1024) * 16);
#define TS_W1(so, g_) __builtin_bit_cast(float4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(wrs, woff + 512 * (g_), so, 0))
    float4 wn0, wn1, wn2, wn3;
    { wn0 = TS_W1(0u, 0); wn1 = TS_W1(0u, 1); wn2 = TS_W1(0u, 2); wn3 = TS_W1(0u, 3); }
    unsigned *my_flag = flags + (size_t)td * G + part;
    bool dead = false;                 // a poll timed out: stop waiting for partners (the result is wrong and the timeout word says so)
    for (int s = 0; s < 64; ++s) {
        const int t = dir ? 63 - s : s;
        const int tn = dir ? t - 1 : t + 1;
        const unsigned tcol = (unsigned)(((int64_t)t * a.Nc + n0) * 16);
        const float4 *sHc = sHs + (size_t)(s & 1) * 64 * BT;
        float4 *sHn = sHs + (size_t)((s & 1) ^ 1) * 64 * BT;

        if (s > 0) {
            float4 wa0 = wn0, wa1 = wn1, wa2 = wn2, wa3 = wn3, wb0, wb1, wb2, wb3, ba, bb;
#define TS_SB() __builtin_amdgcn_sched_barrier(0);
#define TS_Q(W0, W1, W2, W3, B, q)                                              \
    acc[0][0] = MFMA(SDFA_OP(f4c(W0, q)), SDFA_OP(f4c(B, q)), acc[0][0]);       \
    acc[1][0] = MFMA(SDFA_OP(f4c(W1, q)), SDFA_OP(f4c(B, q)), acc[1][0]);       \
    acc[2][0] = MFMA(SDFA_OP(f4c(W2, q)), SDFA_OP(f4c(B, q)), acc[2][0]);       \
    acc[3][0] = MFMA(SDFA_OP(f4c(W3, q)), SDFA_OP(f4c(B, q)), acc[3][0]);
#define TS_KB(CW0, CW1, CW2, CW3, CB, NW0, NW1, NW2, NW3, NB, so, bp)           \
    {                                                                           \
        TS_SB() NW0 = TS_W1(so, 0); NW1 = TS_W1(so, 1); TS_SB()                 \
        TS_Q(CW0, CW1, CW2, CW3, CB, 0)                                         \
        TS_SB() NW2 = TS_W1(so, 2); NW3 = TS_W1(so, 3); TS_SB()                 \
        TS_Q(CW0, CW1, CW2, CW3, CB, 1)                                         \
        TS_SB() NB = (bp)[0]; TS_SB()                                           \
        TS_Q(CW0, CW1, CW2, CW3, CB, 2)                                         \
        TS_Q(CW0, CW1, CW2, CW3, CB, 3)                                         \
    }
            const float4 *brow = sHc + h * BT + l31;          // row pair of k-block kb: brow + kb * 2 * BT
            ba = brow[0];
#pragma unroll 1
            for (int tt = 0; tt < 4; ++tt) {
                const float4 *bt = brow + tt * 16 * BT;
                const float4 *bn = tt + 1 < 4 ? bt + 16 * BT : brow;
                const unsigned so = (unsigned)tt * (8 * 2048 * 16);
                const unsigned son = tt + 1 < 4 ? so + 8 * 2048 * 16 : 0u;
                TS_KB(wa0, wa1, wa2, wa3, ba, wb0, wb1, wb2, wb3, bb, so + 1 * 2048 * 16, bt + 1 * 2 * BT)
                TS_KB(wb0, wb1, wb2, wb3, bb, wa0, wa1, wa2, wa3, ba, so + 2 * 2048 * 16, bt + 2 * 2 * BT)
                TS_KB(wa0, wa1, wa2, wa3, ba, wb0, wb1, wb2, wb3, bb, so + 3 * 2048 * 16, bt + 3 * 2 * BT)
                TS_KB(wb0, wb1, wb2, wb3, bb, wa0, wa1, wa2, wa3, ba, so + 4 * 2048 * 16, bt + 4 * 2 * BT)
                TS_KB(wa0, wa1, wa2, wa3, ba, wb0, wb1, wb2, wb3, bb, so + 5 * 2048 * 16, bt + 5 * 2 * BT)
                TS_KB(wb0, wb1, wb2, wb3, bb, wa0, wa1, wa2, wa3, ba, so + 6 * 2048 * 16, bt + 6 * 2 * BT)
                TS_KB(wa0, wa1, wa2, wa3, ba, wb0, wb1, wb2, wb3, bb, so + 7 * 2048 * 16, bt + 7 * 2 * BT)
                TS_KB(wb0, wb1, wb2, wb3, bb, wa0, wa1, wa2, wa3, ba, son, bn)
            }
            TS_SB()
#undef TS_KB
#undef TS_Q
#undef TS_SB
            wn0 = wa0; wn1 = wa1; wn2 = wa2; wn3 = wa3;
        }
        // cell update: own slice of h_t -> LDS (next step's operand) and -> the H output rows, write-through
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 hq;
            lstm_cell_quad(acc[0][0], acc[1][0], acc[2][0], acc[3][0], c, g, hq);
            const int hq_idx = 8 * hb + 2 * g + h;
            sHn[hq_idx * BT + l31] = hq;
            // STORE-DATA HAZARD (found the hard way, round 3): a 16-byte buffer store whose `soffset` is a scalar REGISTER may be
            // followed at once by a vector instruction that overwrites its data registers -- LLVM's hazard recogniser assumes the
            // hardware interlocks that form (GCNHazardRecognizer::createsVALUHazard), gfx950 does not: lanes 12-15 / 28-31 of the
            // stored quad then carried the NEXT quad's intermediate values.  So the whole offset goes in the vector operand
            // (soffset = 0: the form the compiler does pad) and a wait state follows the store explicitly.
            const unsigned off = (unsigned)hq_idx * ldm + (unsigned)l31 * 16u + tcol;
            if (mode & 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hq), hrs, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hq), hrs, off, 0, 16);   // aux 16 = sc1 (write-through)
            asm volatile("s_nop 1" ::: "memory");
        }
        if (s + 1 < 64) {
            // publish: every storing wave drains, the workgroup meets, one lane stores the step number (never 0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (mode & 1) {      // plain stores: agent-scope release (write-back) in front of the flag, and its own drain (Guideline 16, Pitfall 12)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (!((mode & 4) && part == 1)) __hip_atomic_store(my_flag, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // the next step's input projection seeds the accumulators (requested now: its HBM latency runs under the hand-off)
            TS_GX_ALL(tn)
            if (MAP && s + 2 < 64) ucol = cmap[(int64_t)(dir ? t - 2 : t + 2) * a.Nc];
            // consume: wave 0 polls the partners' flags (lanes 0..G-2), bounded
            if (wave == 0 && !dead) {
                const int pl = lane < G - 1 ? lane : 0;
                const int partner = pl >= part ? pl + 1 : pl;
                dead = split_wait(flags + (size_t)td * G + partner, (unsigned)(s + 1), ctl, lane);   // a partner is not running: give up for good, say so, never hang
            }
            if (mode & 2) {      // plain loads below: ONE agent-scope acquire by the polling wave, drained before the barrier lets the others load
                if (wave == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            } else {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the loads below behind the poll
            }
            __syncthreads();
            // the partners' slices of h_t: 64 - 64/G k-quad rows x 32 sequences, sc1 loads (never this CU's L1), into LDS
            constexpr int NQ = 64 - 64 / G, PER = NQ * BT / NTHR;
            float4 pv[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int idx = i * NTHR + tid, r = idx >> 5, col = idx & 31;
                const int kq = r < part * (64 / G) ? r : r + 64 / G;      // skip the own slice
                if (mode & 2) pv[i] = __builtin_bit_cast(float4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(hrs, (unsigned)kq * ldm + (unsigned)col * 16u, tcol, 0));
                else pv[i] = __builtin_bit_cast(float4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(hrs, (unsigned)kq * ldm + (unsigned)col * 16u, tcol, 16));
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int idx = i * NTHR + tid, r = idx >> 5, col = idx & 31;
                const int kq = r < part * (64 / G) ? r : r + 64 / G;
                sHn[kq * BT + col] = pv[i];
            }
        }
        __syncthreads();   // h_t complete in sHn (own and partners' slices) before anyone reads it
    }
#undef TS_GX
#undef TS_GX_ALL
#undef TS_W1
}

// ------------------------------------------------------------------- time LSTM, small batches, 16-frame tiles
// The next factor of two for a single clip: 16 frames per tile on v_mfma_f32_16x16x4_f32 (same FLOP per cycle as the 32x32x2 form,
// half the columns), so a wave's matrix work per step halves again (512 MFMAs x 32 cycles = 6.8 us) and twice as many CUs take part.
// Two workgroups per tile and direction as in time_lstm_split_kernel (same publish / consume protocol; four waves, one per SIMD, a
// wave owns one block of 32 hidden units x 4 gates = 8 accumulator tiles of 16 x 16).
//
// BIT-IDENTITY with the 32x32x2 kernels.  An fp32 MFMA accumulates its products as an fma chain in k order, so the result depends
// only on the ORDER in which a gate row's 256 products are added.  The 32x32x2 kernels add, per k-block kb, the pairs (8kb+q, 8kb+4+q),
// q = 0..3 (K4 operands: lane half 0 holds k-quad 2kb, half 1 k-quad 2kb+1).  Here one MFMA adds FOUR products, lane group g = l >> 4
// supplying the g-th: MFMA (kb, m), m = 0, 1, takes k = 8kb + {0, 4, 1, 5} (m = 0) / {2, 6, 3, 7} (m = 1) -- the same sequence.  Both
// operands are stored in that order: the weights by the host ([K16 = kb / 2][g][gate row][j = 2 (kb & 1) + m], api.cpp), h by the lanes
// that produce it (LDS [K16][g][frame][j]), so one 16-byte read per lane feeds four consecutive MFMAs.
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void lstm_cell_4(const f32x4v &ai, const f32x4v &af, const f32x4v &ag, const f32x4v &ao, f32x4v &c, float4 &hq) {
    f32x2 hv[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = 2 * p;
        f32x2 cp = {c[r], c[r + 1]};
        const f32x2 ig = sigmoid2(f32x2{ai[r], ai[r + 1]});
        const f32x2 fg = sigmoid2(f32x2{af[r], af[r + 1]});
        const f32x2 gg = tanh2(f32x2{ag[r], ag[r + 1]});
        const f32x2 og = sigmoid2(f32x2{ao[r], ao[r + 1]});
        const f32x2 fc = fg * cp;
        const f32x2 cn = __builtin_elementwise_fma(ig, gg, fc);
        c[r] = cn.x; c[r + 1] = cn.y;
        hv[p] = og * tanh2(cn);
    }
    hq = make_float4(hv[0].x, hv[0].y, hv[1].x, hv[1].y);
}

template <bool MAP = false>
__global__ __launch_bounds__(256) void time_lstm_split16_kernel(TimeLstmArgs a, SplitCtl ctl) {
    unsigned *const flags = ctl.flags;
    const int mode = ctl.mode;
    extern __shared__ float4 sH16[];   // [2 buffers][16 K16][4 g][16 frames] float4 (j = 0..3): 2 x 16 KiB
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = lane & 15, lg = lane >> 4;
    const int q8 = blockIdx.x >> 3;
    const int part = q8 & 1, td = (q8 >> 1) * 8 + (blockIdx.x & 7);     // parts of one tile: block ids 8 apart
    const int dir = td & 1, hb = part * 4 + wave;                       // hb: hidden block of 32 units this wave owns
    const int64_t n0 = (int64_t)(td >> 1) * 16;

    // accumulators: tile (gate qg, half hh) holds gate rows hb*128 + qg*32 + 16 hh + 4 lg + r (r = register) of frame n0 + f
    f32x4v acc[4][2], c[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) c[hh] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const float4 *__restrict__ GXl = reinterpret_cast<const float4 *>(a.GX) + (int64_t)(dir * 256 + hb * 32 + lg) * a.Mc + n0 + f;
    const float4 *__restrict__ GXr = reinterpret_cast<const float4 *>(a.GX) + (int64_t)(dir * 256 + hb * 32 + lg) * a.Mc;   // MAP (column sharing): see time_lstm_body
    const int32_t *__restrict__ cmap = MAP ? a.col_map + n0 + f : nullptr;
    int ucol = MAP ? cmap[(int64_t)(dir ? 63 : 0) * a.Nc] : 0;
#define T16_GX_ALL(t_)                                                                                              \
    _Pragma("unroll") for (int qg = 0; qg < 4; ++qg) _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {               \
        const float4 v = MAP ? GXr[(int64_t)(qg * 8 + 4 * hh) * a.Mc + ucol] : GXl[(int64_t)(qg * 8 + 4 * hh) * a.Mc + (int64_t)(t_) * a.Nc]; \
        acc[qg][hh] = f32x4v{v.x, v.y, v.z, v.w};                                                                   \
    }
    T16_GX_ALL(dir ? 63 : 0)
    if (MAP) ucol = cmap[(int64_t)(dir ? 62 : 1) * a.Nc];

    const unsigned long long hptr = (unsigned long long)(reinterpret_cast<float4 *>(a.H) + (int64_t)dir * 64 * a.Mc);
    const unsigned long long huni = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(hptr >> 32)) << 32) |
                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)hptr);
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void *)huni, 0, 0x7fffffff, 0x00020000);
    const unsigned ldm = (unsigned)a.Mc * 16u;

    // weights of this direction: float4 [16 K16][4 g][1024 rows]; lane (row-in-tile f, group lg) reads [K16][lg][hb*128 + tile*16 + f]
    const unsigned long long wptr = (unsigned long long)(reinterpret_cast<const float4 *>(a.W16) + (size_t)dir * 16 * 4 * 1024 + hb * 128);
    const unsigned long long wuni = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wptr >> 32)) << 32) |
                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wptr);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wuni, 0, 16 * 4 * 1024 * 16, 0x00020000);
    const unsigned woff = (unsigned)((lg * 1024 + f) * 16);
#define T16_W(K16, tile) __builtin_bit_cast(float4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(wrs, woff + (unsigned)(tile) * 256u, (unsigned)(K16) * (4 * 1024 * 16), 0))

    unsigned *my_flag = flags + (size_t)td * 2 + part;
    const unsigned *partner_flag = flags + (size_t)td * 2 + (part ^ 1);
    bool dead = false;
    for (int s = 0; s < 64; ++s) {
        const int t = dir ? 63 - s : s;
        const int tn = dir ? t - 1 : t + 1;
        const unsigned tcol = (unsigned)(((int64_t)t * a.Nc + n0) * 16);
        const float4 *sHc = sH16 + (size_t)(s & 1) * 1024;
        float *sHn = reinterpret_cast<float *>(sH16 + (size_t)((s & 1) ^ 1) * 1024);

        if (s > 0) {
            const float4 *brow = sHc + lg * 16 + f;                 // [K16][lg][f]: + K16 * 64
            float4 wa[8], wb[8], ba, bb;
#pragma unroll
            for (int tl = 0; tl < 8; ++tl) wa[tl] = T16_W(0, tl);
            ba = brow[0];
#define T16_STEP(WC, BC, WN, BN, Kn)                                                                                \
            {                                                                                                       \
                __builtin_amdgcn_sched_barrier(0);      /* the requests stay HERE, a whole half-step ahead of their use (unpinned, both */ \
                _Pragma("unroll") for (int tl = 0; tl < 8; ++tl) WN[tl] = T16_W(Kn, tl);   /* halves' 16 were issued together, 8 of them */ \
                BN = brow[(Kn) * 64];                                                    /* right in front of the MFMAs that use them) */ \
                __builtin_amdgcn_sched_barrier(0);                                                                  \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
                    _Pragma("unroll") for (int tl = 0; tl < 8; ++tl)                                                 \
                        acc[tl >> 1][tl & 1] = MFMA16(SDFA_OP(f4c(WC[tl], j)), SDFA_OP(f4c(BC, j)), acc[tl >> 1][tl & 1]); \
            }
#pragma unroll 1
            for (int K16 = 0; K16 < 16; K16 += 2) {
                T16_STEP(wa, ba, wb, bb, K16 + 1)
                T16_STEP(wb, bb, wa, ba, (K16 + 2) & 15)          // behind the last pair: K16 0 again (dropped)
            }
#undef T16_STEP
        }
        // cell update: hidden units 32 hb + 16 hh + 4 lg + r of frame f
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float4 hq;
            lstm_cell_4(acc[0][hh], acc[1][hh], acc[2][hh], acc[3][hh], c[hh], hq);
            // LDS [K16 = 2 hb + hh][g = 2 (r & 1) + (lg & 1)][f][j = 2 (lg >> 1) + (r >> 1)]
            const int K16 = 2 * hb + hh, jb = 2 * (lg >> 1), ga = lg & 1;
            sHn[(((K16 * 4 + ga) * 16 + f) << 2) + jb] = hq.x;             // r = 0: g = a,     j = jb
            sHn[(((K16 * 4 + 2 + ga) * 16 + f) << 2) + jb] = hq.y;         // r = 1: g = 2 + a, j = jb
            sHn[(((K16 * 4 + ga) * 16 + f) << 2) + jb + 1] = hq.z;         // r = 2: g = a,     j = jb + 1
            sHn[(((K16 * 4 + 2 + ga) * 16 + f) << 2) + jb + 1] = hq.w;     // r = 3: g = 2 + a, j = jb + 1
            // H output rows, K4: quad row 8 hb + 4 hh + lg.  (soffset = 0 and a wait state: see time_lstm_split_kernel)
            const unsigned off = (unsigned)(8 * hb + 4 * hh + lg) * ldm + (unsigned)f * 16u + tcol;
            if (mode & 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hq), hrs, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hq), hrs, off, 0, 16);
            asm volatile("s_nop 1" ::: "memory");
        }
        if (s + 1 < 64) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (mode & 1) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (!((mode & 4) && part == 1)) __hip_atomic_store(my_flag, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            T16_GX_ALL(tn)
            if (MAP && s + 2 < 64) ucol = cmap[(int64_t)(dir ? t - 2 : t + 2) * a.Nc];
            if (wave == 0 && !dead) dead = split_wait(partner_flag, (unsigned)(s + 1), ctl, lane);
            if (mode & 2) {
                if (wave == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            } else {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            __syncthreads();
            // the partner's slice: quad rows (part ^ 1) * 32 .. + 31 of the 16 frames = 512 float4, two per thread
            float4 pv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = i * 256 + tid, q = (part ^ 1) * 32 + (idx >> 4), col = idx & 15;
                if (mode & 2) pv[i] = __builtin_bit_cast(float4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(hrs, (unsigned)q * ldm + (unsigned)col * 16u + tcol, 0, 0));
                else pv[i] = __builtin_bit_cast(float4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(hrs, (unsigned)q * ldm + (unsigned)col * 16u + tcol, 0, 16));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = i * 256 + tid, q = (part ^ 1) * 32 + (idx >> 4), col = idx & 15;
                // k = 4 q + r: k-block q >> 1, K16 = q >> 2, a = q & 1, j = 2 ((q >> 1) & 1) + (r >> 1), g = 2 (r & 1) + a
                const int K16 = q >> 2, ga = q & 1, jb = 2 * ((q >> 1) & 1);
                sHn[(((K16 * 4 + ga) * 16 + col) << 2) + jb] = pv[i].x;
                sHn[(((K16 * 4 + 2 + ga) * 16 + col) << 2) + jb] = pv[i].y;
                sHn[(((K16 * 4 + ga) * 16 + col) << 2) + jb + 1] = pv[i].z;
                sHn[(((K16 * 4 + 2 + ga) * 16 + col) << 2) + jb + 1] = pv[i].w;
            }
        }
        __syncthreads();
    }
#undef T16_GX_ALL
#undef T16_W
}

// ------------------------------------------------------------------------------ time LSTM on bf16 MFMA
// Mixed-precision modes: the BiLSTM recurrence h_{t-1} * W_hh^T on v_mfma_f32_32x32x16_bf16 (TERMS 1 or 3, see
// freq_lstm_bf16_kernel); input projection (from the GEMM), accumulation, cell state and gate math stay fp32.
// h lives in LDS as bf16x8 octets, split by the lane that produced it: octet 4w + 2q + h holds hidden units
// 32w+16q+4h+{0..3} and 32w+16q+8+4h+{0..3}; the host packs W_hh's K axis in that order (api.cpp: pack_rec_bf16).
template <int NT, int TERMS>
__global__ __launch_bounds__(512, 2) void time_lstm_bf16_kernel(TimeLstmArgs a) {
    constexpr bool LO = TERMS > 1, X6 = TERMS == 6;      // X6: the six-product split (see freq_lstm_bf16x6_kernel), NT = 1 only (96 KiB of LDS)
    constexpr int NPL = X6 ? 3 : (LO ? 2 : 1), BT = 32 * NT;
    extern __shared__ bf16x8 sHb[];   // [2 buffers][NPL planes][32 octets][BT sequences]
    auto SH = [&](int buf, int plane) { return sHb + ((size_t)(buf * NPL + plane) * 32) * BT; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // 8 waves: hidden block of 32
    const int l31 = lane & 31, h = lane >> 5;
    const int dir = blockIdx.x & 1;
    const int64_t n0 = (int64_t)(blockIdx.x >> 1) * BT;

    const float4 *__restrict__ GX = reinterpret_cast<const float4 *>(a.GX);
    // per direction: [plane hi | lo][32 octets][1024 gate rows]
    float4 *__restrict__ H = reinterpret_cast<float4 *>(a.H);

    // cell state: column tile 0 in registers; tile 1 (NT = 2) in the 32 KiB of LDS the h planes leave free, [quad g][thread] -- its 16
    // registers hold the lo weight plane a k-step ahead instead (the K loop below); a thread only ever touches its own entries
    constexpr bool C1_LDS = NT == 2 && !X6;
    float4 *const sC1 = reinterpret_cast<float4 *>(sHb + (size_t)2 * NPL * 32 * BT);
    f32x16 c[C1_LDS ? 1 : NT];
#pragma unroll
    for (int j = 0; j < (C1_LDS ? 1 : NT); ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
    if (C1_LDS) {
#pragma unroll
        for (int g = 0; g < 4; ++g) sC1[g * 512 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    f32x16 acc[4][NT];
#define TB_GX(t_, gt, g, j) GX[(int64_t)(dir * 256 + wave * 32 + (gt) * 8 + 2 * (g) + h) * a.Mc + (int64_t)(t_) * a.Nc + n0 + l31 + (j) * 32]
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float4 v = TB_GX(dir ? 63 : 0, gt, g, j);
                acc[gt][j][4 * g + 0] = v.x; acc[gt][j][4 * g + 1] = v.y; acc[gt][j][4 * g + 2] = v.z; acc[gt][j][4 * g + 3] = v.w;
            }

    // the recurrent weights through a buffer descriptor (one per direction: [plane][32 octet rows][1024 gate rows] of 16 bytes): a request is
    // descriptor + uniform byte offset (plane, k-step) + one loop-invariant lane offset -- no 64-bit address arithmetic in the K loop
    typedef unsigned int tb_u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16x8 *>(reinterpret_cast<const bf16x8 *>(a.Wb) + (size_t)dir * BF16_PLANES * 32 * 1024), 0, BF16_PLANES * 32 * 1024 * 16, 0x00020000);
    const unsigned woff = (unsigned)((h * 1024 + wave * 128 + l31) * 16);
#define TB_W(pl, ks_, gt) __builtin_bit_cast(bf16x8, (tb_u32x4)__builtin_amdgcn_raw_buffer_load_b128(wrs, woff + (unsigned)(gt) * 512u, (unsigned)(((pl) * 32 + 2 * (ks_)) * 1024 * 16), 0))
    bf16x8 R[4], S[4], H6a[4], H6b[4];        // three-product form: R = hi, S = lo plane; six-product form: H6a / H6b = hi (ping-pong), R = mid, S = lo
    if (!X6) {
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) { R[gt] = TB_W(0, 0, gt); if (LO) S[gt] = TB_W(1, 0, gt); }
    } else {
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) { S[gt] = TB_W(2, 0, gt); R[gt] = TB_W(1, 0, gt); H6a[gt] = TB_W(0, 0, gt); }
    }

#if defined(SDFA_STAMPS) && defined(__HIP_DEVICE_COMPILE__)    /* diagnostic build: K loop / cell update + plane split / barrier, as time_lstm_body */
    unsigned long long bt0 = 0, bt1 = 0, bt2 = 0, bt3 = 0, bv_k = 0, bv_cell = 0, bv_bar = 0;
#define BSTAMP(t) LSTAMP(t)
#else
#define BSTAMP(t)
#endif
    for (int s = 0; s < 64; ++s) {
        const int t = dir ? 63 - s : s;
        const int tn = dir ? t - 1 : t + 1;
        const int64_t mcol = (int64_t)t * a.Nc + n0 + l31;
        const int cur = s & 1;
        BSTAMP(bt0)

        if (X6 && s > 0) {
            // six products per k-step and accumulator, smallest first: lo*hi, mid*mid, mid*hi, hi*lo, hi*mid, hi*hi (NT = 1: four
            // accumulators, so a product pass is four independent MFMAs).  The lo (L) and mid (M) weight planes are refilled IN PLACE
            // behind their last pass -- 20 and 16 MFMAs ahead of their next use; the hi plane, needed last and longest, ping-pongs
            // (HC / HN: requested a whole k-step ahead); the three h operand planes ping-pong too (BC / BN).  The last k-step requests
            // k-step 0's rows again: the next step's.  (As in the three-product loop below, the `next = load; ...; cur = next` source
            // form this replaces was compiled back into loads at the point of use: vmcnt(0) in front of every pass.)
            bf16x8 bA[3], bB[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bA[pl] = SH(cur, pl)[h * BT + l31];
#define TB_SB() __builtin_amdgcn_sched_barrier(0);
#define T6_PASS(W, B) _Pragma("unroll") for (int gt = 0; gt < 4; ++gt) acc[gt][0] = MFMA_BF16(W[gt], B, acc[gt][0]);
#define T6_KSTEP(HC, HN, BC, BN, ks_)                                                                                             \
            {                                                                                                                     \
                const int kn_ = (ks_) + 1 < 16 ? (ks_) + 1 : 0, kh_ = (ks_) + 1 < 16 ? (ks_) + 1 : 15;                            \
                TB_SB() _Pragma("unroll") for (int gt = 0; gt < 4; ++gt) HN[gt] = TB_W(0, kn_, gt); TB_SB()                       \
                T6_PASS(S, BC[0])                                                          /* lo  * hi  */                        \
                TB_SB() _Pragma("unroll") for (int gt = 0; gt < 4; ++gt) S[gt] = TB_W(2, kn_, gt); TB_SB()                        \
                T6_PASS(R, BC[1])                                                          /* mid * mid */                        \
                T6_PASS(R, BC[0])                                                          /* mid * hi  */                        \
                TB_SB() _Pragma("unroll") for (int gt = 0; gt < 4; ++gt) R[gt] = TB_W(1, kn_, gt);                                \
                _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) BN[pl] = SH(cur, pl)[(2 * kh_ + h) * BT + l31];                  \
                TB_SB()                                                                                                           \
                T6_PASS(HC, BC[2])                                                         /* hi  * lo  */                        \
                T6_PASS(HC, BC[1])                                                         /* hi  * mid */                        \
                T6_PASS(HC, BC[0])                                                         /* hi  * hi  */                        \
            }
#pragma unroll 1
            for (int ks = 0; ks < 16; ks += 2) {
                T6_KSTEP(H6a, H6b, bA, bB, ks)
                T6_KSTEP(H6b, H6a, bB, bA, ks + 1)
            }
#undef T6_KSTEP
#undef T6_PASS
#undef TB_SB
        }
        if (!X6 && s > 0) {
            // Weight planes a whole k-step ahead, in place (round 5, last): R = hi plane, S = lo plane of the current k-step, four gate
            // tiles each.  A tile PAIR's registers are refilled with the next k-step's rows right behind the pair's last product, so every
            // request is 16 - 20 MFMAs (512 - 640 cycles) ahead of its use with 32 weight registers in all; the last k-step requests k-step
            // 0 again -- the next step's first rows.  The h operands: hi octets ping-pong (BHC / BHN), the lo octet is re-read behind its
            // last use.  (The loop this replaces had `next = load; ...; cur = next` in source form; the compiler turned that back into a load
            // at the top of the k-step that uses it -- every k-step began with an exposed L2 round trip: K loop 57 % busy in the stamps.)
            // Per accumulator the products still arrive as hi*hi, hi*lo, lo*hi: bit-identical to the other forms.
            bf16x8 bhA[NT], bhB[NT], bl[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bhA[j] = SH(cur, 0)[h * BT + j * 32 + l31];
                if (LO) bl[j] = SH(cur, NPL - 1)[h * BT + j * 32 + l31];
            }
#define TB_SB() __builtin_amdgcn_sched_barrier(0);
#define TB_HH(g0, BHC)                                                                                                            \
            _Pragma("unroll") for (int gt = (g0); gt < (g0) + 2; ++gt)                                                            \
                _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[gt][j] = MFMA_BF16(R[gt], BHC[j], acc[gt][j]);                 \
            if (LO) {                                                                                                             \
                _Pragma("unroll") for (int gt = (g0); gt < (g0) + 2; ++gt)                                                        \
                    _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[gt][j] = MFMA_BF16(R[gt], bl[j], acc[gt][j]);              \
            }
#define TB_LH(g0, BHC)                                                                                                            \
            _Pragma("unroll") for (int gt = (g0); gt < (g0) + 2; ++gt)                                                            \
                _Pragma("unroll") for (int j = 0; j < NT; ++j) acc[gt][j] = MFMA_BF16(S[gt], BHC[j], acc[gt][j]);
#define TB_KSTEP(BHC, BHN, ks_)                                                                                                   \
            {                                                                                                                     \
                const int kn_ = (ks_) + 1 < 16 ? (ks_) + 1 : 0;        /* weights: behind the last k-step, k-step 0 of the next step */ \
                const int kh_ = (ks_) + 1 < 16 ? (ks_) + 1 : 15;       /* h operands: behind the last k-step, the same rows again (dropped) */ \
                TB_SB()                                                                                                           \
                TB_HH(0, BHC)                                                                                                     \
                TB_SB() R[0] = TB_W(0, kn_, 0); R[1] = TB_W(0, kn_, 1); TB_SB()                                                   \
                TB_HH(2, BHC)                                                                                                     \
                TB_SB() R[2] = TB_W(0, kn_, 2); R[3] = TB_W(0, kn_, 3);                                                           \
                _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                                  \
                    BHN[j] = SH(cur, 0)[(2 * kh_ + h) * BT + j * 32 + l31];                                                       \
                    if (LO) bl[j] = SH(cur, NPL - 1)[(2 * kh_ + h) * BT + j * 32 + l31];                                          \
                }                                                                                                                 \
                TB_SB()                                                                                                           \
                if (LO) {                                                                                                         \
                    TB_LH(0, BHC)                                                                                                 \
                    TB_SB() S[0] = TB_W(1, kn_, 0); S[1] = TB_W(1, kn_, 1); TB_SB()                                               \
                    TB_LH(2, BHC)                                                                                                 \
                    TB_SB() S[2] = TB_W(1, kn_, 2); S[3] = TB_W(1, kn_, 3); TB_SB()                                               \
                }                                                                                                                 \
            }
#pragma unroll 1
            for (int ks = 0; ks < 16; ks += 2) {
                TB_KSTEP(bhA, bhB, ks)
                TB_KSTEP(bhB, bhA, ks + 1)
            }
#undef TB_KSTEP
#undef TB_LH
#undef TB_HH
#undef TB_SB
        }
#if defined(SDFA_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]), "v"(acc[0][NT - 1]), "v"(acc[1][NT - 1]), "v"(acc[2][NT - 1]), "v"(acc[3][NT - 1]));   // all MFMAs done
#endif
        BSTAMP(bt1)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float4 hq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (C1_LDS && j == 1) {
                    f32x16 ct;                                  // only this quad's four entries are read and written
                    const float4 cq = sC1[g * 512 + tid];
                    ct[4 * g + 0] = cq.x; ct[4 * g + 1] = cq.y; ct[4 * g + 2] = cq.z; ct[4 * g + 3] = cq.w;
                    lstm_cell_quad(acc[0][j], acc[1][j], acc[2][j], acc[3][j], ct, g, hq[g]);
                    sC1[g * 512 + tid] = make_float4(ct[4 * g + 0], ct[4 * g + 1], ct[4 * g + 2], ct[4 * g + 3]);
                } else {
                    lstm_cell_quad(acc[0][j], acc[1][j], acc[2][j], acc[3][j], c[C1_LDS ? 0 : j], g, hq[g]);
                }
                H[(int64_t)(dir * 64 + 8 * wave + 2 * g + h) * a.Mc + mcol + j * 32] = hq[g];
                if (s + 1 < 64) {   // this quad's gate registers are free: request the next step's input projection into them
#pragma unroll
                    for (int gt = 0; gt < 4; ++gt) {
                        const float4 v = TB_GX(tn, gt, g, j);
                        acc[gt][j][4 * g + 0] = v.x; acc[gt][j][4 * g + 1] = v.y; acc[gt][j][4 * g + 2] = v.z; acc[gt][j][4 * g + 3] = v.w;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (X6) {
                    bf16x8 hi, mid, lo;
                    split_octet3(hq[2 * q], hq[2 * q + 1], hi, mid, lo);
                    SH(cur ^ 1, 0)[(4 * wave + 2 * q + h) * BT + j * 32 + l31] = hi;
                    SH(cur ^ 1, 1)[(4 * wave + 2 * q + h) * BT + j * 32 + l31] = mid;
                    SH(cur ^ 1, NPL - 1)[(4 * wave + 2 * q + h) * BT + j * 32 + l31] = lo;
                } else {
                    bf16x8 hi, lo;
                    split_octet(hq[2 * q], hq[2 * q + 1], hi, lo);
                    SH(cur ^ 1, 0)[(4 * wave + 2 * q + h) * BT + j * 32 + l31] = hi;
                    if (LO) SH(cur ^ 1, NPL - 1)[(4 * wave + 2 * q + h) * BT + j * 32 + l31] = lo;
                }
            }
        }
        BSTAMP(bt2)
        __syncthreads();   // h_s complete in the other buffer before anyone reads it; this one free for step s+1's writes
#if defined(SDFA_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
        BSTAMP(bt3)
        if (s > 0) { bv_k += bt1 - bt0; bv_cell += bt2 - bt1; bv_bar += bt3 - bt2; }
#endif
    }
#if defined(SDFA_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    if (NT == 2 && lane == 0) { atomicAdd(&g_lsub[0], bv_k); atomicAdd(&g_lsub[1], bv_cell); atomicAdd(&g_lsub[2], bv_bar); atomicAdd(&g_lsub[3], 63ull); }
#endif
#undef BSTAMP
#undef TB_GX
#undef TB_W
}

}  // namespace

#ifdef SDFA_STAMPS
extern "C" int sdfa_debug_read_lstm_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lstamp), sizeof(unsigned long long) * 8) != hipSuccess) return -3;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_lstamp), z, sizeof z) != hipSuccess) return -3; }
    return 0;
}
extern "C" int sdfa_debug_read_lstm_sub(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lsub), sizeof(unsigned long long) * 4) != hipSuccess) return -3;
    if (reset) { unsigned long long z[4] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_lsub), z, sizeof z) != hipSuccess) return -3; }
    return 0;
}
extern "C" int sdfa_debug_read_lstm_xcd(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lxcd), sizeof(unsigned long long) * 32) != hipSuccess) return -3;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_lxcd), z, sizeof z) != hipSuccess) return -3; }
    return 0;
}
extern "C" int sdfa_debug_read_lstm_span(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lspan), sizeof(unsigned long long) * 4) != hipSuccess) return -3;
    if (reset) { unsigned long long z[4] = {~0ull, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_lspan), z, sizeof z) != hipSuccess) return -3; }
    return 0;
}
#endif

// Launch forms of the fp32 recurrence (FreqLstmArgs::shape; all bit-identical):
//   9  freq_lstm_v3_kernel, persistent (tile queue), one workgroup per CU   -- default
//   8  freq_lstm_v3_kernel, one hardware-dispatched workgroup per tile
//   5  freq_lstm_v2_kernel, persistent, two workgroups per CU
//   3  freq_lstm_v2_kernel, hardware-dispatched, two workgroups per CU       -- the fallback that shares a CU
// sdfa_model_autotune times the four on the device.  (The round-1 kernel and the one-per-CU launch forms of the second kernel
// -- shapes 1, 2, 4, 6, 7 -- were removed in round 3; any other value launches the default.)
template <bool SHARED>
static hipError_t launch_freq(const FreqLstmArgs &a, hipStream_t s) {
    const int shape = (a.shape == 8 || a.shape == 5 || a.shape == 3) ? a.shape : 9;
    const unsigned n_tiles = (unsigned)(a.Mc / 64 * 2);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    cus = std::max(1, cus - a.reserve_cus);      // CUs left to kernels of other streams (sdfa_model_set_reserved_cus)
    if (shape == 8 || shape == 9) {      // the third form: one workgroup per CU by construction (96 KiB of LDS)
        const size_t lds = 2 * 48 * 64 * sizeof(float4) + 512 * sizeof(float) + 16;
        const void *fn = shape == 9 ? reinterpret_cast<const void *>(freq_lstm_v3_kernel<SHARED, true>) : reinterpret_cast<const void *>(freq_lstm_v3_kernel<SHARED, false>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (shape == 9) {
            e = hipMemsetAsync(a.tile_counter, 0, sizeof(int), s);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((freq_lstm_v3_kernel<SHARED, true>), dim3(n_tiles < (unsigned)cus ? n_tiles : (unsigned)cus), dim3(256), lds, s, a);
        } else {
            hipLaunchKernelGGL((freq_lstm_v3_kernel<SHARED, false>), dim3(n_tiles), dim3(256), lds, s, a);
        }
        return hipGetLastError();
    }
    if (shape == 5) {                    // persistent: workgroups pull tiles from a queue, two per CU
        const unsigned slots = 2u * (unsigned)cus;
        hipError_t e = hipMemsetAsync(a.tile_counter, 0, sizeof(int), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((freq_lstm_v2_kernel<SHARED, true>), dim3(n_tiles < slots ? n_tiles : slots), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    size_t lone = 0;
#ifdef SDFA_STAMPS
    if (getenv("SDFA_LONE")) lone = 32 * 1024;      // diagnostic: 32 KB of unused dynamic LDS = one workgroup per CU
#endif
    hipLaunchKernelGGL((freq_lstm_v2_kernel<SHARED, false>), dim3(n_tiles), dim3(256), lone, s, a);
    return hipGetLastError();
}

template <bool SHARED>
static hipError_t launch_freq_bf16(const FreqLstmArgs &a, hipStream_t s) {
    if (!a.Wb) return hipErrorInvalidValue;
    if ((a.terms == 6 || a.terms == 3) && a.shape != 3) {      // default: one persistent workgroup per CU (shape 8: one hardware-dispatched workgroup per tile)
        const int pl = a.terms == 6 ? 3 : 2;
        const size_t lds = (size_t)(2 * pl * 24 * 64) * sizeof(bf16x8) + 512 * sizeof(float) + 16;         // 146 KiB / 98 KiB
        const unsigned n_tiles = (unsigned)(a.Mc / 64 * 2);
        const bool persist = a.shape != 8;
        const void *fn = pl == 3 ? (persist ? reinterpret_cast<const void *>(freq_lstm_bf16p_v3_kernel<SHARED, true, 3>) : reinterpret_cast<const void *>(freq_lstm_bf16p_v3_kernel<SHARED, false, 3>))
                                 : (persist ? reinterpret_cast<const void *>(freq_lstm_bf16p_v3_kernel<SHARED, true, 2>) : reinterpret_cast<const void *>(freq_lstm_bf16p_v3_kernel<SHARED, false, 2>));
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        unsigned grid = n_tiles;
        if (persist) {
            const unsigned cus = (unsigned)std::max(1, sdfa_cu_count() - a.reserve_cus);
            grid = n_tiles < cus ? n_tiles : cus;
            e = hipMemsetAsync(a.tile_counter, 0, sizeof(int), s);
            if (e != hipSuccess) return e;
        }
        if (pl == 3) {
            if (persist) hipLaunchKernelGGL((freq_lstm_bf16p_v3_kernel<SHARED, true, 3>), dim3(grid), dim3(256), lds, s, a);
            else hipLaunchKernelGGL((freq_lstm_bf16p_v3_kernel<SHARED, false, 3>), dim3(grid), dim3(256), lds, s, a);
        } else {
            if (persist) hipLaunchKernelGGL((freq_lstm_bf16p_v3_kernel<SHARED, true, 2>), dim3(grid), dim3(256), lds, s, a);
            else hipLaunchKernelGGL((freq_lstm_bf16p_v3_kernel<SHARED, false, 2>), dim3(grid), dim3(256), lds, s, a);
        }
        return hipGetLastError();
    }
    if (a.terms == 6) {                      // shape 3: the two-per-CU form (shares a CU with other streams' kernels)
        const size_t lds = (size_t)(3 * 16 * 64 + 3 * 8 * 64) * sizeof(bf16x8) + 512 * sizeof(float);      // 74 KiB: two workgroups per CU
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(freq_lstm_bf16x6_kernel<SHARED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((freq_lstm_bf16x6_kernel<SHARED>), dim3((unsigned)(a.Mc / 64 * 2)), dim3(256), lds, s, a);
        return hipGetLastError();
    }
    if (a.terms == 1)
        hipLaunchKernelGGL((freq_lstm_bf16_kernel<SHARED, 1>), dim3((unsigned)(a.Mc / 64 * 2)), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((freq_lstm_bf16_kernel<SHARED, 3>), dim3((unsigned)(a.Mc / 64 * 2)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t sdfa_launch_freq_lstm(const FreqLstmArgs &a, hipStream_t s) {
    if (a.terms) return a.col_limit ? launch_freq_bf16<true>(a, s) : launch_freq_bf16<false>(a, s);
    return a.col_limit ? launch_freq<true>(a, s) : launch_freq<false>(a, s);
}

template <int NT, bool MAP>
static hipError_t launch_time(const TimeLstmArgs &a, hipStream_t s) {
    const size_t lds = 2 * 64 * 32 * NT * sizeof(float4);   // 128 KiB (NT 2) / 64 KiB (NT 1)
    {   // per launch: cheap, and correct for every device / thread the library is used from
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(time_lstm_kernel<NT, MAP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((time_lstm_kernel<NT, MAP>), dim3((unsigned)(a.Nc / (32 * NT) * 2)), dim3(512), lds, s, a);
    return hipGetLastError();
}

template <int NT, int TERMS>
static hipError_t launch_time_bf16(const TimeLstmArgs &a, hipStream_t s) {
    const size_t lds = (size_t)2 * (TERMS == 6 ? 3 : (TERMS > 1 ? 2 : 1)) * 32 * 32 * NT * sizeof(bf16x8)    // 128 KiB (NT 2, split) ... 32 KiB; 96 KiB (NT 1, six-product)
                       + (NT == 2 && TERMS != 6 ? 4 * 512 * sizeof(float4) : 0);                                  // + 32 KiB: column tile 1's cell state (160 KiB in all for NT 2, split)
    {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(time_lstm_bf16_kernel<NT, TERMS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((time_lstm_bf16_kernel<NT, TERMS>), dim3((unsigned)(a.Nc / (32 * NT) * 2)), dim3(512), lds, s, a);
    return hipGetLastError();
}

extern thread_local int g_sdfa_time_lstm_handoff;   // api.cpp ("time_lstm_handoff"): bit 0 = plain stores + agent release, bit 1 = agent acquire + plain loads; bit 2 (tests) = part 1 never publishes
extern thread_local int g_sdfa_time_lstm_timeout_us; // api.cpp ("time_lstm_timeout_us"): bound of one wait for a partner workgroup

// Behind every launch of a cooperating-workgroup kernel: the repair pass (exits at once unless that launch timed out).
template <bool MAP>
static hipError_t launch_time_repair(const TimeLstmArgs &a, hipStream_t s) {
    const size_t lds = 2 * 64 * 32 * sizeof(float4);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(time_lstm_repair_kernel<MAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(time_lstm_repair_kernel<MAP>, dim3((unsigned)(a.Nc / 32 * 2)), dim3(512), lds, s, a, a.flags);
    return hipGetLastError();
}

static SplitCtl split_ctl(const TimeLstmArgs &a) {
    // default 20 ms (round 6; 0.2 s before): 2,000 x a healthy hand-off wait and 10 x a whole single-clip layer.  A wait only gets that long
    // when other work keeps the partner workgroup off the device; the repair pass behind the launch makes an expiry safe (it redoes
    // the layer, about 2 ms), so the bound is a latency cap, not a correctness margin: waiting longer than the repair costs buys nothing.
    const long long us = g_sdfa_time_lstm_timeout_us > 0 ? g_sdfa_time_lstm_timeout_us : 20000;
    return SplitCtl{a.flags + 4, a.flags, a.status, (unsigned)std::min<long long>(us * 100, 0xffffffffll), g_sdfa_time_lstm_handoff};
}

template <int G, bool MAP>
static hipError_t launch_time_split(const TimeLstmArgs &a, hipStream_t s) {
    const size_t lds = 96 * 1024;      // 64 KiB used; 96 KiB requested so that two workgroups never share a CU (see the kernel)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(time_lstm_split_kernel<G, MAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)(a.Nc / 32 * 2 * G);
    // this launch's time-out word + flag words (one per workgroup), zeroed every launch: a block of its own, a multiple of 16 bytes
    e = hipMemsetAsync(a.flags, 0, ((size_t)grid + 4) * sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((time_lstm_split_kernel<G, MAP>), dim3(grid), dim3(512 / G), lds, s, a, split_ctl(a));
    e = hipGetLastError();
    return e != hipSuccess ? e : launch_time_repair<MAP>(a, s);
}

template <bool MAP>
static hipError_t launch_time_split16(const TimeLstmArgs &a, hipStream_t s) {
    const size_t lds = 96 * 1024;      // 32 KiB used; 96 KiB requested: one workgroup per CU (hand-off form, see time_lstm_split_kernel)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(time_lstm_split16_kernel<MAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)(a.Nc / 16 * 2 * 2);
    e = hipMemsetAsync(a.flags, 0, ((size_t)grid + 4) * sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(time_lstm_split16_kernel<MAP>, dim3(grid), dim3(256), lds, s, a, split_ctl(a));
    e = hipGetLastError();
    return e != hipSuccess ? e : launch_time_repair<MAP>(a, s);
}

extern thread_local int g_sdfa_time_lstm_split;   // api.cpp ("time_lstm_split" option): 0 = by size, 1 = never

template <bool MAP>
static hipError_t launch_time_any(const TimeLstmArgs &a, hipStream_t s) {
    // 64-frame tiles while they fill the 256 CUs (one 8-wave workgroup per CU); otherwise 32-frame tiles
    const bool big = (a.Nc / 64) * 2 >= 256;
    if (!a.terms && a.flags && g_sdfa_time_lstm_split != 1) {
        // small batches: the gate rows of a tile split over G cooperating workgroups (time_lstm_split_kernel).  They exchange h
        // every step, so all of them should be resident at once: only while the grid fits the CUs this model may use (one
        // workgroup per CU; sdfa_model_set_reserved_cus leaves some to other streams)
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        cus -= a.reserve_cus;
        const int64_t wg1 = a.Nc / 32 * 2;                       // workgroups of time_lstm_kernel<1>
        const bool range_ok = (int64_t)64 * a.Mc * 16 + (int64_t)a.Mc * 16 < 0x7fffffff && a.Nc % 128 == 0 && a.flag_words >= wg1 * 2 + 4;   // buffer offsets; 8-block groups
        // G = 2: four waves per workgroup, one per SIMD.  (G = 4 -- two waves per workgroup -- was measured too: no faster, a wave's
        // matrix work per step is the same; profiles/r03_time_lstm_split.txt.)
        const int G = (range_ok && wg1 * 2 <= cus) ? 2 : 0;
        // 16-frame tiles (time_lstm_split16_kernel) while even their grid -- twice the workgroups -- fits the CUs; option 16 / 32 force one
        if (range_ok && a.W16 && g_sdfa_time_lstm_split != 32 && wg1 * 4 <= cus && a.flag_words >= wg1 * 4 + 4) return launch_time_split16<MAP>(a, s);
        if (g_sdfa_time_lstm_split == 16) return hipErrorInvalidValue;      // asked for, not possible at this size
        if (G == 2) return launch_time_split<2, MAP>(a, s);
    }
    if (a.terms) {
        if (!a.Wb || MAP) return hipErrorInvalidValue;      // the bf16 recurrences read un-shared input projections (api.cpp expands first)
        if (a.terms == 6) return launch_time_bf16<1, 6>(a, s);      // three planes of h: 32-frame tiles only (96 KiB of LDS)
        if (a.terms == 1) return big ? launch_time_bf16<2, 1>(a, s) : launch_time_bf16<1, 1>(a, s);
        return big ? launch_time_bf16<2, 3>(a, s) : launch_time_bf16<1, 3>(a, s);
    }
    return big ? launch_time<2, MAP>(a, s) : launch_time<1, MAP>(a, s);
}

hipError_t sdfa_launch_time_lstm(const TimeLstmArgs &a, hipStream_t s) {
    return a.col_map ? launch_time_any<true>(a, s) : launch_time_any<false>(a, s);
}

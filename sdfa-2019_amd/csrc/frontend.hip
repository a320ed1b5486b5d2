// Spectral-gather front end: PCM window -> (64, 128, 3) mel / delta / delta-delta features.
//
// Reference arithmetic (paths relative to the reference repository):
//   window cut + zero pad            speech_anime/datasets/sliding_window.py:356-362
//   per-window pre-emphasis 0.65     saber/data/audio/features/misc.py:8-17   (y[0] = x[0])
//   Hamming STFT, center=False       saber/data/audio/features/spectrogram.py:82-96
//   power, 128-band Slaney mel       spectrogram.py:97-98, misc.py:110-117
//   dB, normalise, clamp             spectrogram.py:238,245-249
//   delta / delta-delta (SG width 9) speech_anime/datasets/get_features.py:199-207
//   (T,F,C) interleave               get_features.py:210-215, sliding_window.py:462
//
// One 512-thread workgroup per animation frame (8 waves = 2 per SIMD; the 150 KB of LDS allow one workgroup per CU,
// so the second wave per SIMD is what hides LDS latency).  The zero-padded, pre-emphasised window is
// staged once in LDS (coalesced HBM read of sliding*4 bytes); each wave then transforms PAIRS
// of STFT columns as one complex radix-4 Stockham FFT (column t in the real part, t+1 in the
// imaginary part) with twiddles and the Hamming window resident in LDS, untangles the two
// spectra, gathers the sparse mel rows (CSR, ~449 non-zeros, only bins below 3.6 kHz are ever
// needed) and writes log-mel into an LDS image from which the 9-tap delta filters and the
// interleaved (T,F,C) store run.  HBM traffic per frame: the window read + 98,304 bytes written.
// A wave's FFT lives in its own LDS buffer, so the stages are separated by wavefront-scope fences only (LDS executes a
// wave's instructions in order); the workgroup meets at two barriers: window staged, mel image complete.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

constexpr int FE_WAVES = 8, FE_THREADS = 64 * FE_WAVES;
#define WAVE_SYNC() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// One wave: complex FFT of a column pair (stage-0 butterfly inputs in v: column A in .x, column B in .y), untangled power
// spectra, sparse mel gather, dB, normalise, clamp.  Returns mel[col][bb] for band = lane + 64 * bb.  Everything stays in
// the wave's own LDS buffer `buf`, so only wavefront-scope fences separate the stages.
template <int WIN>
__device__ __forceinline__ void fft_pair_to_mel(float2 (&v)[WIN / 256][4], float2 *buf, const float2 *sTw, const int *sBin0,
                                                const float (*sW8)[128], int lane, float (&mel)[2][2]) {
    constexpr int NB = 256, NR4 = WIN / 256;
    constexpr bool HAS_R2 = (WIN == 512);
    float *pw0 = reinterpret_cast<float *>(buf), *pw1 = pw0 + NB;   // power spectra of the two columns (bins < 256) reuse the buffer
    {
        int Ns = 1;
#pragma unroll
        for (int stage = 0; stage < (HAS_R2 ? 4 : 5); ++stage) {
            if (stage > 0) {
#pragma unroll
                for (int b = 0; b < NR4; ++b) {
                    const int j = lane + 64 * b, k = j & (Ns - 1);
                    const int tstep = k * (WIN / (4 * Ns));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float2 x = buf[j + r * (WIN / 4)];
                        v[b][r] = (r == 0) ? x : cmul(x, sTw[(tstep * r) & (WIN - 1)]);
                    }
                }
                WAVE_SYNC()   // every read of this stage done before any write
            }
#pragma unroll
            for (int b = 0; b < NR4; ++b) {
                const int j = lane + 64 * b, k = j & (Ns - 1);
                const int j0 = (j - k) * 4 + k;
                float2 a0 = cadd(v[b][0], v[b][2]), a1 = csub(v[b][0], v[b][2]);
                float2 a2 = cadd(v[b][1], v[b][3]), d = csub(v[b][1], v[b][3]);
                float2 a3 = make_float2(d.y, -d.x);   // (-i) * d
                buf[j0] = cadd(a0, a2);
                buf[j0 + Ns] = cadd(a1, a3);
                buf[j0 + 2 * Ns] = csub(a0, a2);
                buf[j0 + 3 * Ns] = csub(a1, a3);
            }
            WAVE_SYNC()
            Ns *= 4;
        }
        if (HAS_R2) {   // WIN = 512 = 4^4 * 2: final radix-2 stage, Ns = 256
            float2 u[4][2];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = lane + 64 * b;
                u[b][0] = buf[j];
                u[b][1] = cmul(buf[j + WIN / 2], sTw[j]);
            }
            WAVE_SYNC()
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = lane + 64 * b;
                buf[j] = cadd(u[b][0], u[b][1]);
                buf[j + WIN / 2] = csub(u[b][0], u[b][1]);
            }
            WAVE_SYNC()
        }
        // ---- untangle the two real spectra, power (bins in registers first: the power arrays reuse the FFT buffer)
        float p0[4], p1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane + 64 * i;
            float2 zk = buf[k], zn = buf[(WIN - k) & (WIN - 1)];
            float ar = 0.5f * (zk.x + zn.x), ai = 0.5f * (zk.y - zn.y);     // A = (Z[k] + conj Z[N-k]) / 2
            float br = 0.5f * (zk.y + zn.y), bi = -0.5f * (zk.x - zn.x);    // B = (Z[k] - conj Z[N-k]) / (2i)
            p0[i] = __fadd_rn(__fmul_rn(ar, ar), __fmul_rn(ai, ai));
            p1[i] = __fadd_rn(__fmul_rn(br, br), __fmul_rn(bi, bi));
        }
        WAVE_SYNC()
#pragma unroll
        for (int i = 0; i < 4; ++i) { pw0[lane + 64 * i] = p0[i]; pw1[lane + 64 * i] = p1[i]; }
        WAVE_SYNC()
        // ---- sparse mel gather, dB, normalise, clamp
#pragma unroll
        for (int col = 0; col < 2; ++col)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int band = lane + 64 * bb;
                float m = 0.f;
                const float *pw = col ? pw1 : pw0;
                const int b0 = sBin0[band];     // fixed 8 taps per band (zero-weight padding): unrolled, all reads in flight together
#pragma unroll
                for (int e = 0; e < 8; ++e) m += sW8[e][band] * pw[b0 + e];
                float db = __fmul_rn(10.0f, log10f(fmaxf(m, 1.1920929e-07f)));
                float nv = __fdiv_rn(__fadd_rn(__fsub_rn(db, 20.0f), 80.0f), 80.0f);
                mel[col][bb] = fminf(fmaxf(nv, 0.f), 1.f);
            }
        WAVE_SYNC()
    }
}

// One wave, ONE real column: the WIN-point real FFT as a complex FFT of M = WIN/2 points on z[n] = y[2n] + i y[2n+1],
// then X[k] = E[k] + W_WIN^k O[k] with E, O the even / odd half spectra recovered from Z[k] and conj Z[M-k].  No second
// column shares the transform, so the result is a function of the column's samples alone -- whichever batch, chunk or
// neighbour it is computed with.  v: stage-0 butterfly inputs (.x = even sample, .y = odd sample, windowed).
template <int WIN>
__device__ __forceinline__ void fft_real_to_mel(float2 (&v)[WIN / 512][4], float2 *buf, const float2 *sTw, const int *sBin0,
                                                const float (*sW8)[128], int lane, float (&mel)[2]) {
    constexpr int M = WIN / 2, NR4 = M / 256;
    constexpr bool HAS_R2 = (M == 512);            // 512 = 4^4 * 2, 256 = 4^4
    float *pw = reinterpret_cast<float *>(buf);    // power spectrum (bins < 256) reuses the buffer
    int Ns = 1;
#pragma unroll
    for (int stage = 0; stage < 4; ++stage) {
        if (stage > 0) {
#pragma unroll
            for (int b = 0; b < NR4; ++b) {
                const int j = lane + 64 * b, k = j & (Ns - 1);
                const int tstep = k * (M / (4 * Ns));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float2 x = buf[j + r * (M / 4)];
                    v[b][r] = (r == 0) ? x : cmul(x, sTw[2 * ((tstep * r) & (M - 1))]);     // W_M^m = W_WIN^(2m)
                }
            }
            WAVE_SYNC()
        }
#pragma unroll
        for (int b = 0; b < NR4; ++b) {
            const int j = lane + 64 * b, k = j & (Ns - 1);
            const int j0 = (j - k) * 4 + k;
            float2 a0 = cadd(v[b][0], v[b][2]), a1 = csub(v[b][0], v[b][2]);
            float2 a2 = cadd(v[b][1], v[b][3]), d = csub(v[b][1], v[b][3]);
            float2 a3 = make_float2(d.y, -d.x);   // (-i) * d
            buf[j0] = cadd(a0, a2);
            buf[j0 + Ns] = cadd(a1, a3);
            buf[j0 + 2 * Ns] = csub(a0, a2);
            buf[j0 + 3 * Ns] = csub(a1, a3);
        }
        WAVE_SYNC()
        Ns *= 4;
    }
    if (HAS_R2) {   // final radix-2 stage, Ns = 256
        float2 u[4][2];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = lane + 64 * b;
            u[b][0] = buf[j];
            u[b][1] = cmul(buf[j + M / 2], sTw[2 * j]);
        }
        WAVE_SYNC()
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = lane + 64 * b;
            buf[j] = cadd(u[b][0], u[b][1]);
            buf[j + M / 2] = csub(u[b][0], u[b][1]);
        }
        WAVE_SYNC()
    }
    float p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;                                       // bins 0..255 (only those below 3.6 kHz are used)
        const float2 zk = buf[k & (M - 1)], zn = buf[(M - k) & (M - 1)];
        const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));     // E = (Z[k] + conj Z[M-k]) / 2
        const float2 o = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));    // O = (Z[k] - conj Z[M-k]) / (2i)
        const float2 x = cadd(e, cmul(o, sTw[k]));                                    // X[k] = E + W_WIN^k O
        p[i] = __fadd_rn(__fmul_rn(x.x, x.x), __fmul_rn(x.y, x.y));
    }
    WAVE_SYNC()
#pragma unroll
    for (int i = 0; i < 4; ++i) pw[lane + 64 * i] = p[i];
    WAVE_SYNC()
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
        const int band = lane + 64 * bb;
        float m = 0.f;
        const int b0 = sBin0[band];         // fixed 8 taps per band (zero-weight padding): unrolled, all reads in flight together
#pragma unroll
        for (int e = 0; e < 8; ++e) m += sW8[e][band] * pw[b0 + e];
        float db = __fmul_rn(10.0f, log10f(fmaxf(m, 1.1920929e-07f)));
        float nv = __fdiv_rn(__fadd_rn(__fsub_rn(db, 20.0f), 80.0f), 80.0f);
        mel[bb] = fminf(fmaxf(nv, 0.f), 1.f);
    }
    WAVE_SYNC()
}

template <int WIN>
__global__ __launch_bounds__(FE_THREADS) void frontend_kernel(FrontendConsts c, const float *__restrict__ pcm,
                                                       const int64_t *__restrict__ clip_off,
                                                       const int64_t *__restrict__ clip_len,
                                                       const int32_t *__restrict__ frame_clip,
                                                       const int64_t *__restrict__ frame_start, float *__restrict__ out) {
    constexpr int HOP = WIN / 8, SLIDING = HOP * 63 + WIN, NR4 = WIN / 256;   // radix-4 butterflies per lane
    __shared__ float sY[SLIDING];
    __shared__ float2 sFft[FE_WAVES][WIN];
    __shared__ float2 sTw[WIN];
    __shared__ float sHamm[WIN];
    __shared__ float sMel[64][128];   // lanes walk the band index in every access: no padding needed
    __shared__ int sBin0[128];        // first FFT bin of each mel band (a band's bins are consecutive)
    __shared__ float sW8[8][128];     // its weights, tap-major, zero padded to 8 taps (the widest band has 8)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t frame = blockIdx.x;
    const int clip = frame_clip[frame];
    const int64_t off = clip_off[clip], len = clip_len[clip], s0 = frame_start[frame];

    for (int i = tid; i < WIN; i += FE_THREADS) { sTw[i] = c.twiddle[i]; sHamm[i] = c.hamm[i]; }
    for (int i = tid; i < 128; i += FE_THREADS) sBin0[i] = c.mel_bin0[i];
    for (int i = tid; i < 1024; i += FE_THREADS) sW8[i >> 7][i & 127] = c.mel_w8[i];
    // window cut, zero pad, pre-emphasis (fp32, one rounding per op as numpy does)
    for (int i = tid; i < SLIDING; i += FE_THREADS) {
        const int64_t g = s0 + i, last = len > 0 ? len - 1 : 0;
        // unconditional requests at clamped addresses, zero padding applied afterwards (see mel_columns_kernel); a clip of
        // length 0 owns no sample at all, so nothing is requested for it (block-uniform condition: one frame, one clip)
        float r0 = 0.f, rm = 0.f;
        if (len > 0) {
            r0 = pcm[off + (g < 0 ? 0 : (g > last ? last : g))];
            rm = pcm[off + (g - 1 < 0 ? 0 : (g - 1 > last ? last : g - 1))];
        }
        float x = (g >= 0 && g < len) ? r0 : 0.f;
        float xm = (i > 0 && g - 1 >= 0 && g - 1 < len) ? rm : 0.f;
        sY[i] = (i == 0) ? x : __fsub_rn(x, __fmul_rn(0.65f, xm));
    }
    __syncthreads();

    // STFT columns are transformed in PAIRS (one complex FFT = two real columns).  The pairing follows the ABSOLUTE
    // hop index of a column, floor(start / hop) + t, not its position in the window: two frames of one clip that
    // contain the same column (starts a whole number of hops apart) then give it the same partner, so their mel
    // values -- and every interior feature column -- are bit-identical, which is what makes column sharing
    // (share.hip) exact.  With an odd base the first and last column have no partner and run alone.
    const int64_t hop_base = (s0 >= 0 ? s0 : s0 - (HOP - 1)) / HOP;         // floor division
    const int odd = (int)(hop_base & 1);
    const int njobs = 32 + odd;                                              // 32 pairs, or solo + 31 pairs + solo
    float2 *buf = sFft[wave];
    for (int it = 0; it < (36 + FE_WAVES - 1) / FE_WAVES; ++it) {   // up to 33 jobs over the waves
        const int job = it * FE_WAVES + wave;
        const bool live = job < njobs;
        int t0 = 2 * job - odd, t1 = t0 + 1;                                 // columns in the real / imaginary part
        if (!live) { t0 = 0; t1 = 1; }                                       // idle slot: harmless recomputation, results dropped
        const bool has0 = t0 >= 0, has1 = t1 <= 63;
        const float *ya = sY + (has0 ? t0 : 0) * HOP, *yb = sY + (has1 ? t1 : 63) * HOP;
        const float ga = has0 ? 1.f : 0.f, gb = has1 ? 1.f : 0.f;
        float2 v[NR4][4];
        // ---- stage 0 (Ns = 1) straight from the windowed signal
#pragma unroll
        for (int b = 0; b < NR4; ++b) {
            const int j = lane + 64 * b;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nidx = j + r * (WIN / 4);
                const float w = sHamm[nidx];
                v[b][r] = make_float2(ga * (w * ya[nidx]), gb * (w * yb[nidx]));
            }
        }
        float mel[2][2];
        fft_pair_to_mel<WIN>(v, buf, sTw, sBin0, sW8, lane, mel);
#pragma unroll
        for (int col = 0; col < 2; ++col)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int tt = col ? t1 : t0;
                if (live && tt >= 0 && tt <= 63) sMel[tt][lane + 64 * bb] = mel[col][bb];
            }
    }
    __syncthreads();   // mel image complete
    // ---- Savitzky-Golay deltas (edges replicate the first / last interior value) + (T,F,C) store
    const float c2[9] = {28.f / 462.f, 7.f / 462.f, -8.f / 462.f, -17.f / 462.f, -20.f / 462.f,
                         -17.f / 462.f, -8.f / 462.f, 7.f / 462.f, 28.f / 462.f};
    // a thread takes 4 consecutive frequency bins of one time step: 12 output floats = three 16-byte stores
    float4 *dst = reinterpret_cast<float4 *>(out + frame * (64 * 128 * 3));
    for (int i4 = tid; i4 < 64 * 128 / 4; i4 += FE_THREADS) {
        const int t = i4 >> 5, f0 = (i4 & 31) * 4;
        const int tc = t < 4 ? 4 : (t > 59 ? 59 : t);
        float m[4], d1[4], d2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = -4; j <= 4; ++j) {
                const float mv = sMel[tc + j][f0 + q];
                s1 += (float)j * (1.0f / 60.0f) * mv;
                s2 += c2[j + 4] * mv;
            }
            m[q] = sMel[t][f0 + q]; d1[q] = s1; d2[q] = s2;
        }
        dst[i4 * 3 + 0] = make_float4(m[0], d1[0], d2[0], m[1]);
        dst[i4 * 3 + 1] = make_float4(d1[1], d2[1], m[2], d1[2]);
        dst[i4 * 3 + 2] = make_float4(d2[2], m[3], d1[3], d2[3]);
    }
}

// ----------------------------------------------------------------------------- "spectral gather" form
// Windows of one clip whose starts differ by whole hops contain the same STFT columns (at 60 fps / 16 kHz every 12th
// frame, 25 hops apart): only window column 0 is special (its first sample is not pre-emphasised, misc.py:17).  The share
// map (share.hip, t in 1..63) lists every DISTINCT column once -- 26 new ones per frame instead of 64 -- and
//   mel_columns_kernel      transforms those (one column per wave as a half-size complex FFT, samples read straight from
//                           the clip with the pre-emphasis applied on the fly) into a mel table  float[distinct][128];
//   gather_features_kernel  builds each frame from its 64 table rows: mel image in LDS, delta filters, (T,F,C) store --
//                           32 KB read (mostly L2) + 98 KB written per frame: the HBM-bound scan SURVEY 8(d) describes.
constexpr int MC_WAVES = 8;

// The clip as a BUFFER: descriptor in scalar registers (a wave transforms one column, so the clip is wave-uniform) and the
// sample index as a 32-bit byte offset -- no 64-bit address arithmetic per request.  The zero padding on both sides of the
// clip stays explicit (index clamped for the request, value selected afterwards, all in 32-bit arithmetic: |index| < 2^29):
// the hardware's range check cannot do it -- an index in front of the clip wraps to a huge offset that is NOT refused,
// and the compiler merges neighbouring requests into 8-byte ones that are judged as a whole at the clip's end.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t clip_buffer(const float *base, int len) {
    const unsigned long long xb = (unsigned long long)base;
    const unsigned long long xuni = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(xb >> 32)) << 32) |
                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xb);
    return __builtin_amdgcn_make_buffer_rsrc((void *)xuni, 0, __builtin_amdgcn_readfirstlane(len * 4), 0x00020000);
}

// One wave, one STFT column at sample position p of a clip (radix-4 / LDS-staged FFT): samples with the pre-emphasis applied on
// the fly -> mel[bb] for band lane + 64 bb.  raw0: the column is a window's column 0, whose very first sample is not pre-emphasised
// (misc.py:17).  A function of (clip samples, p, raw0) alone: whichever kernel, batch or neighbour computes it, the bits are the same.
template <int WIN>
__device__ __forceinline__ void column_mel_r4(const __amdgpu_buffer_rsrc_t xrs, int len, int64_t p, bool raw0, float2 *buf, const float2 *sTw,
                                              const float *sHamm, const int *sBin0, const float (*sW8)[128], int lane, float (&mel)[2]) {
    constexpr int M = WIN / 2, NR4 = M / 256;
    float2 v[NR4][4];
#pragma unroll
    for (int b = 0; b < NR4; ++b) {
        const int j = lane + 64 * b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 2 * (j + r * (M / 4));                       // even sample of z[j + r M/4]
            // samples g-1, g, g+1 (indices relative to the clip); pre-emphasis with one rounding per op (misc.py:8-17).  All
            // requests are unconditional, at clamped indices: a load behind a divergent condition compiles to branch + load +
            // wait, which serialised the requests of a column into ~10 memory round trips
            const int g0 = (int)(p + i), last = len - 1;
#define MC_REQ(idx) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (unsigned)((idx) < 0 ? 0 : ((idx) > last ? last : (idx))) * 4u, 0, 0))
            const float rm = MC_REQ(g0 - 1), r0 = MC_REQ(g0), r1 = MC_REQ(g0 + 1);
#undef MC_REQ
            const float xm = (unsigned)(g0 - 1) < (unsigned)len ? rm : 0.f;
            const float x0 = (unsigned)g0 < (unsigned)len ? r0 : 0.f;
            const float x1 = (unsigned)(g0 + 1) < (unsigned)len ? r1 : 0.f;
            const float y0 = (raw0 && i == 0) ? x0 : __fsub_rn(x0, __fmul_rn(0.65f, xm));
            const float y1 = __fsub_rn(x1, __fmul_rn(0.65f, x0));
            v[b][r] = make_float2(sHamm[i] * y0, sHamm[i + 1] * y1);
        }
    }
    fft_real_to_mel<WIN>(v, buf, sTw, sBin0, sW8, lane, mel);
}

template <int WIN>
__global__ __launch_bounds__(64 * MC_WAVES) void mel_columns_kernel(FrontendConsts c, const float *__restrict__ pcm,
                                                                    const int64_t *__restrict__ clip_off, const int64_t *__restrict__ clip_len,
                                                                    const int32_t *__restrict__ frame_clip, const int64_t *__restrict__ frame_start,
                                                                    const int32_t *__restrict__ col_src, const int64_t *__restrict__ n_distinct,
                                                                    float *__restrict__ mel_table) {
    constexpr int HOP = WIN / 8, M = WIN / 2;
    __shared__ float2 sFft[MC_WAVES][M];
    __shared__ float2 sTw[WIN];
    __shared__ float sHamm[WIN];
    __shared__ int sBin0[128];        // first FFT bin of each mel band (a band's bins are consecutive)
    __shared__ float sW8[8][128];     // its weights, tap-major, zero padded to 8 taps (the widest band has 8)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < WIN; i += 64 * MC_WAVES) { sTw[i] = c.twiddle[i]; sHamm[i] = c.hamm[i]; }
    for (int i = tid; i < 128; i += 64 * MC_WAVES) sBin0[i] = c.mel_bin0[i];
    for (int i = tid; i < 1024; i += 64 * MC_WAVES) sW8[i >> 7][i & 127] = c.mel_w8[i];
    __syncthreads();

    const int64_t nd = *n_distinct;
    float2 *buf = sFft[wave];
    for (int64_t u = (int64_t)blockIdx.x * MC_WAVES + wave; u < nd; u += (int64_t)gridDim.x * MC_WAVES) {
        // (frame, window column) of this distinct column -> clip and absolute sample position
        const int row = col_src[u], n = row >> 6, t = row & 63, clip = frame_clip[n];
        const int64_t len64 = clip_len[clip], p = frame_start[n] + (int64_t)t * HOP;
        const int len = (int)(len64 > 0x1fffffff ? 0x1fffffff : len64);
        const __amdgpu_buffer_rsrc_t xrs = clip_buffer(pcm + clip_off[clip], len);
        float mel[2];
        column_mel_r4<WIN>(xrs, len, p, t == 0, buf, sTw, sHamm, sBin0, sW8, lane, mel);
        mel_table[u * 128 + lane] = mel[0];
        mel_table[u * 128 + 64 + lane] = mel[1];
    }
}

// ---------------------------------------------------------------------------- radix-8, register-resident (round 4, WIN = 1024)
// The 512-point complex FFT of a column as THREE radix-8 passes on the 8 values a lane holds (512 = 8^3), with two transpositions
// through the wave's LDS buffer in between -- instead of four radix-4 passes and a radix-2 pass that each go through LDS (the
// kernel was bound by its vector-ALU / LDS instruction count: about 1,000 instructions per column, 700 of them the FFT).
//   pass 1  lane l holds z[l + 64 r], r = 0..7: DFT-8 over r, times W_512^(l q)             -> 8 sub-problems (q) of 64 points over l
//   x-pose  lane l = l1 + 8 l2, value q   ->  lane l1 + 8 q, value l2
//   pass 2  DFT-8 over l2, times W_64^(l1 q2)                                                 -> 64 sub-problems (q, q2) of 8 points over l1
//   x-pose  lane l1 + 8 q, value q2  ->  lane q + 8 q2, value l1
//   pass 3  DFT-8 over l1: value q3 of lane q + 8 q2 is Z[q + 8 q2 + 64 q3] = Z[lane + 64 q3]  (natural order, no bit reversal)
// LDS layouts are padded (rows of 72 / 68 float2) so that both sides of each transposition are bank-conflict-free.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 cmul2(f32x2 a, f32x2 b) { return f32x2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ f32x2 mul_mi(f32x2 a) { return f32x2{a.y, -a.x}; }          // (-i) * a

// forward DFT of 8 values in place: x[q] <- sum_r x[r] exp(-2 pi i r q / 8)   (decimation in frequency: 4 + 4, then two 4-point DFTs)
__device__ __forceinline__ void dft8(f32x2 (&x)[8]) {
    constexpr float R = 0.70710678118654752440f;
    const f32x2 a0 = x[0] + x[4], a4 = x[0] - x[4];
    const f32x2 a1 = x[1] + x[5], d5 = x[1] - x[5];
    const f32x2 a2 = x[2] + x[6], d6 = x[2] - x[6];
    const f32x2 a3 = x[3] + x[7], d7 = x[3] - x[7];
    const f32x2 a5 = f32x2{(d5.x + d5.y) * R, (d5.y - d5.x) * R};      // * W8^1 = (1 - i) / sqrt 2
    const f32x2 a6 = mul_mi(d6);                                       // * W8^2 = -i
    const f32x2 a7 = f32x2{(d7.y - d7.x) * R, -(d7.x + d7.y) * R};     // * W8^3 = (-1 - i) / sqrt 2
    const f32x2 b0 = a0 + a2, b2 = a0 - a2, b1 = a1 + a3, b3 = mul_mi(a1 - a3);
    const f32x2 c0 = a4 + a6, c2 = a4 - a6, c1 = a5 + a7, c3 = mul_mi(a5 - a7);
    x[0] = b0 + b1; x[4] = b0 - b1; x[2] = b2 + b3; x[6] = b2 - b3;
    x[1] = c0 + c1; x[5] = c0 - c1; x[3] = c2 + c3; x[7] = c2 - c3;
}

constexpr int MC8_BUF = 8 * 72;      // float2 per wave: the larger of the two padded transposition layouts (also >= 512 for the final spectrum)

// v: z[lane + 64 r] (windowed, .x = even sample, .y = odd sample).  Returns mel[bb] for band = lane + 64 bb.
__device__ __forceinline__ void fft512_r8_to_mel(f32x2 (&v)[8], f32x2 *buf, const float2 *sTw, const int *sBin0, const float (*sW8)[128], int lane,
                                                 float (&mel)[2]) {
    const int l1 = lane & 7, hi3 = lane >> 3;
    // ---- pass 1 + twiddle W_512^(lane q) = sTw[2 lane q] (table of W_1024)
    dft8(v);
#pragma unroll
    for (int q = 1; q < 8; ++q) { const float2 w = sTw[(2 * lane * q) & 1023]; v[q] = cmul2(v[q], f32x2{w.x, w.y}); }
#pragma unroll
    for (int q = 0; q < 8; ++q) buf[q * 72 + lane] = v[q];
    WAVE_SYNC()
    // this lane is now (l1, q = hi3): values l2 = 0..7
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = buf[hi3 * 72 + l1 + 8 * r];
    WAVE_SYNC()      // every read done before the second layout overwrites the buffer
    // ---- pass 2 + twiddle W_64^(l1 q2) = sTw[16 l1 q2]
    dft8(v);
#pragma unroll
    for (int q2 = 1; q2 < 8; ++q2) { const float2 w = sTw[16 * l1 * q2]; v[q2] = cmul2(v[q2], f32x2{w.x, w.y}); }
#pragma unroll
    for (int q2 = 0; q2 < 8; ++q2) buf[l1 * 68 + hi3 + 8 * q2] = v[q2];
    WAVE_SYNC()
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = buf[r * 68 + lane];       // lane = q + 8 q2, value l1 = r
    WAVE_SYNC()
    // ---- pass 3: v[i] = Z[lane + 64 i]
    dft8(v);
    // ---- real-FFT recombination for bins k = lane + 64 i < 256: needs Z[(512 - k) & 511], another lane's value -> through the buffer
#pragma unroll
    for (int i = 0; i < 8; ++i) buf[lane + 64 * i] = v[i];
    WAVE_SYNC()
    float p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;
        const f32x2 zk = v[i], zn = buf[(512 - k) & 511];
        const f32x2 e = f32x2{0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};     // E = (Z[k] + conj Z[M-k]) / 2
        const f32x2 o = f32x2{0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x)};    // O = (Z[k] - conj Z[M-k]) / (2i)
        const float2 w = sTw[k];
        const f32x2 x = e + cmul2(o, f32x2{w.x, w.y});                         // X[k] = E + W_1024^k O
        p[i] = __fadd_rn(__fmul_rn(x.x, x.x), __fmul_rn(x.y, x.y));
    }
    WAVE_SYNC()
    float *pw = reinterpret_cast<float *>(buf);
#pragma unroll
    for (int i = 0; i < 4; ++i) pw[lane + 64 * i] = p[i];
    WAVE_SYNC()
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
        const int band = lane + 64 * bb;
        float m = 0.f;
        const int b0 = sBin0[band];
#pragma unroll
        for (int e = 0; e < 8; ++e) m += sW8[e][band] * pw[b0 + e];
        // 10 log10(max(m, eps)) as 10 log10(2) * v_log_f32 (the argument is a normal number >= 2^-23: no denormal handling needed;
        // 1 ulp of the hardware logarithm is 1e-5 dB, the feature tolerance corresponds to 4e-3 dB), then (dB - 20 + 80) / 80
        const float db = 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(m, 1.1920929e-07f));
        const float nv = __fadd_rn(__fsub_rn(db, 20.0f), 80.0f) * 0.0125f;
        mel[bb] = fminf(fmaxf(nv, 0.f), 1.f);
    }
    WAVE_SYNC()
}

// One wave, one STFT column of the 16 kHz geometry (WIN = 1024) at sample position p of a clip: see column_mel_r4.
__device__ __forceinline__ void column_mel_r8(const __amdgpu_buffer_rsrc_t xrs, int len, int64_t p, bool raw0, f32x2 *buf, const float2 *sTw,
                                              const float *sHamm, const int *sBin0, const float (*sW8)[128], int lane, float (&mel)[2]) {
    constexpr int WIN = 1024;
    f32x2 v[8];
    // A column that lies inside its clip with one sample to spare in front (all but the first / last few of a clip: a wave-uniform
    // test) needs no clamping and no zero fill: its three samples per element come as ONE request of four dwords (g0 - 1 .. g0 + 2,
    // dword-aligned; the compiler keeps the three that are used: buffer_load_dwordx3).  NB the whole vector is bit-cast to float4:
    // __builtin_bit_cast(float, q.y) on an ELEMENT of the integer vector compiles to element 0 with this clang (ROCm 7.2) -- the
    // first build loaded one dword per element and returned garbage; found by reading the ISA.
    const bool inside = __builtin_amdgcn_readfirstlane((int)(p >= 1 && p + WIN < (int64_t)len)) != 0;
    if (inside) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const unsigned voff = (unsigned)((int)p - 1 + 2 * lane) * 4u;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = 2 * (lane + 64 * r);
            const float4 q = __builtin_bit_cast(float4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(xrs, voff + 512u * r, 0, 0));
            const float xm = q.x, x0 = q.y, x1 = q.z;
            const float y0 = (raw0 && i == 0) ? x0 : __fsub_rn(x0, __fmul_rn(0.65f, xm));
            const float y1 = __fsub_rn(x1, __fmul_rn(0.65f, x0));
            const float2 hw = *reinterpret_cast<const float2 *>(&sHamm[i]);
            v[r] = f32x2{hw.x * y0, hw.y * y1};
        }
    } else
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = 2 * (lane + 64 * r);                             // even sample of z[lane + 64 r]
        const int g0 = (int)(p + i), last = len - 1;
#define MC_REQ(idx) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (unsigned)((idx) < 0 ? 0 : ((idx) > last ? last : (idx))) * 4u, 0, 0))
        const float rm = MC_REQ(g0 - 1), r0 = MC_REQ(g0), r1 = MC_REQ(g0 + 1);
#undef MC_REQ
        const float xm = (unsigned)(g0 - 1) < (unsigned)len ? rm : 0.f;
        const float x0 = (unsigned)g0 < (unsigned)len ? r0 : 0.f;
        const float x1 = (unsigned)(g0 + 1) < (unsigned)len ? r1 : 0.f;
        const float y0 = (raw0 && i == 0) ? x0 : __fsub_rn(x0, __fmul_rn(0.65f, xm));
        const float y1 = __fsub_rn(x1, __fmul_rn(0.65f, x0));
        v[r] = f32x2{sHamm[i] * y0, sHamm[i + 1] * y1};
    }
    fft512_r8_to_mel(v, buf, sTw, sBin0, sW8, lane, mel);
}

__global__ __launch_bounds__(64 * MC_WAVES) void mel_columns_r8_kernel(FrontendConsts c, const float *__restrict__ pcm,
                                                                       const int64_t *__restrict__ clip_off, const int64_t *__restrict__ clip_len,
                                                                       const int32_t *__restrict__ frame_clip, const int64_t *__restrict__ frame_start,
                                                                       const int32_t *__restrict__ col_src, const int64_t *__restrict__ n_distinct,
                                                                       float *__restrict__ mel_table) {
    constexpr int WIN = 1024, HOP = WIN / 8;
    __shared__ f32x2 sFft[MC_WAVES][MC8_BUF];
    __shared__ float2 sTw[WIN];
    __shared__ float sHamm[WIN];
    __shared__ int sBin0[128];
    __shared__ float sW8[8][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < WIN; i += 64 * MC_WAVES) { sTw[i] = c.twiddle[i]; sHamm[i] = c.hamm[i]; }
    for (int i = tid; i < 128; i += 64 * MC_WAVES) sBin0[i] = c.mel_bin0[i];
    for (int i = tid; i < 1024; i += 64 * MC_WAVES) sW8[i >> 7][i & 127] = c.mel_w8[i];
    __syncthreads();

    const int64_t nd = *n_distinct;
    f32x2 *buf = sFft[wave];
    // XCD x (= block % 8: workgroups go to the XCDs round-robin, each XCD has its own L2) takes a CONTIGUOUS eighth of the distinct
    // columns -- which are numbered clip by clip, hop by hop -- so that the PCM an XCD's waves walk through stays in ITS L2; with
    // columns dealt round-robin every XCD read all of the PCM (counters: 11.2 KB fetched per frame for 1.07 KB of new samples).
    const int64_t span = ((nd + 7) / 8 + MC_WAVES - 1) / MC_WAVES * MC_WAVES;
    const int64_t u_end = min(nd, ((int64_t)(blockIdx.x & 7) + 1) * span);
    for (int64_t u = (int64_t)(blockIdx.x & 7) * span + (int64_t)(blockIdx.x >> 3) * MC_WAVES + wave; u < u_end; u += (int64_t)(gridDim.x >> 3) * MC_WAVES) {
        const int row = col_src[u], n = row >> 6, t = row & 63, clip = frame_clip[n];
        const int64_t len64 = clip_len[clip], p = frame_start[n] + (int64_t)t * HOP;
        const int len = (int)(len64 > 0x1fffffff ? 0x1fffffff : len64);
        const __amdgpu_buffer_rsrc_t xrs = clip_buffer(pcm + clip_off[clip], len);
        float mel[2];
        column_mel_r8(xrs, len, p, t == 0, buf, sTw, sHamm, sBin0, sW8, lane, mel);
        mel_table[u * 128 + lane] = mel[0];
        mel_table[u * 128 + 64 + lane] = mel[1];
    }
}

// Which frame a workgroup takes.  Frames 12 apart (25 hops at 60 fps: the hop-aligned ones) share 39 of their 64 table rows, and a
// row is read by up to three frames (n, n + 12, n + 24).  Workgroups go to the XCDs round-robin (block b -> XCD b % 8, each with
// its own L2), so in plain order the three readers of a row sit on different XCDs or far apart in time and every read came from
// beyond the L2 (counters: 30.6 KB fetched per frame for 32 KB requested).  Here XCD x takes a CONTIGUOUS range of the "chain
// order" c = (n % 12) * ceil(F / 12) + n / 12, so that consecutive workgroups of one XCD are frames 12 apart: a frame finds its
// predecessors' rows in that XCD's L2.  Any period gives correct results (this is a permutation of the frames); 12 is the
// locality of the 8 kHz / 16 kHz, 60 fps geometry.
__device__ __forceinline__ int64_t gather_frame_of_block(int64_t b, int64_t n_frames) {
    constexpr int PERIOD = 12, XCDS = 8;
    const int64_t per_class = (n_frames + PERIOD - 1) / PERIOD, total = per_class * PERIOD;
    const int64_t span = (total + XCDS - 1) / XCDS;
    const int64_t c = (b % XCDS) * span + b / XCDS;
    if (c >= total) return -1;
    const int64_t n = (c % per_class) * PERIOD + c / per_class;
    return n < n_frames ? n : -1;
}

__global__ __launch_bounds__(256) void gather_features_kernel(const float4 *__restrict__ mel_table, const int32_t *__restrict__ col_to_u,
                                                              int64_t Nc, int64_t n_frames, int frame_major, int chain_order,
                                                              float *__restrict__ out) {
    __shared__ float sMel[64][128];
    const int tid = threadIdx.x;
    const int64_t frame = chain_order ? gather_frame_of_block(blockIdx.x, n_frames) : (int64_t)blockIdx.x;
    if (frame < 0 || frame >= n_frames) return;
    const int64_t st = frame_major ? 1 : Nc, sf = frame_major ? 64 : 1;      // col_to_u[t * st + frame * sf]
    {   // the frame's 64 table rows: all eight index requests, then all eight row requests, then the LDS writes (as a rolled loop
        // this was eight dependent round-trip pairs in front of every workgroup's barrier)
        const int f4 = tid & 31;
        int32_t u[8];
        float4 row[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = col_to_u[(int64_t)((tid >> 5) + 8 * i) * st + frame * sf];
#pragma unroll
        for (int i = 0; i < 8; ++i) row[i] = mel_table[(int64_t)u[i] * 32 + f4];
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float4 *>(&sMel[(tid >> 5) + 8 * i][4 * f4]) = row[i];
    }
    __syncthreads();
    // Savitzky-Golay deltas (edges replicate the first / last interior value) + (T,F,C) store: as in frontend_kernel.  The frame's
    // 96 KiB go out as 16-byte stores through a buffer descriptor on the frame (left to itself the compiler stored a thread's 48
    // bytes as 12 + 12 + 16 + 8: four partially covered passes over every line)
    const float c2[9] = {28.f / 462.f, 7.f / 462.f, -8.f / 462.f, -17.f / 462.f, -20.f / 462.f,
                         -17.f / 462.f, -8.f / 462.f, 7.f / 462.f, 28.f / 462.f};
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long ob = (unsigned long long)(out + frame * (64 * 128 * 3));
    const unsigned long long ouni = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ob >> 32)) << 32) |
                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ob);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)ouni, 0, 64 * 128 * 3 * 4, 0x00020000);
#define GF_ST(off, a_, b_, c_, d_) __builtin_amdgcn_raw_buffer_store_b128(u32x4{__builtin_bit_cast(unsigned, a_), __builtin_bit_cast(unsigned, b_), __builtin_bit_cast(unsigned, c_), __builtin_bit_cast(unsigned, d_)}, ors, (unsigned)(off), 0, 0)
    for (int i4 = tid; i4 < 64 * 128 / 4; i4 += 256) {
        const int t = i4 >> 5, f0 = (i4 & 31) * 4;
        const int tc = t < 4 ? 4 : (t > 59 ? 59 : t);
        float m[4], d1[4], d2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = -4; j <= 4; ++j) {
                const float mv = sMel[tc + j][f0 + q];
                s1 += (float)j * (1.0f / 60.0f) * mv;
                s2 += c2[j + 4] * mv;
            }
            m[q] = sMel[t][f0 + q]; d1[q] = s1; d2[q] = s2;
        }
        GF_ST(i4 * 48, m[0], d1[0], d2[0], m[1]);
        GF_ST(i4 * 48 + 16, d1[1], d2[1], m[2], d1[2]);
        GF_ST(i4 * 48 + 32, d2[2], m[3], d1[3], d2[3]);
    }
#undef GF_ST
}


// ---------------------------------------------------------------------------- "spectral stream" form (round 5)
// The two-kernel form above pays for the sharing with a table round trip: every distinct column's 128 mel values are written to HBM
// (13.7 KB per frame) and read back by up to three frames (17.1 KB per frame past the L2) -- 1.33 x the algorithmic bytes of the stage.
// Here the table never exists.  Hop-aligned frames of a clip form CHAINS (share.hip: prev[n] = nearest earlier frame of the clip whose
// start differs by d whole hops, 1 <= d <= 62; at 60 fps every 12th frame, d = 25): along a chain the windows are one STFT column
// stream, member i covering stream columns k_i .. k_i + 63, k_i = k_(i-1) + d_i.  One workgroup walks a SEGMENT of a chain (the members
// inside an aligned block of B frame indices): its waves transform the stream's columns in order into an LDS RING of mel rows
// (ST_RING slots), plus each member's own column 0 (whose first sample is not pre-emphasised, misc.py:17:
// never shared), and when the members that fit the ring together are complete the whole workgroup applies the delta filters and stores the frame's
// (T, F, C) rows -- the same instructions, in the same order, on the same mel values as gather_features_kernel: bitwise the two-kernel
// form (a column's mel is a function of (clip, position) alone, column_mel_r8 / column_mel_r4).  HBM: the PCM once per XCD that needs
// it + the 98,304 feature bytes per frame.  Cost: a segment of J members transforms 25 J + 39 + J columns instead of 26 J (the 39 of
// its first window that a predecessor in another segment also holds): B = 144 -> J = 12, + 12.5 %.
// Any frame table works (the chain structure is read from prev / shift, nothing is assumed about the frame rate): frames that share
// nothing are segments of one member.
constexpr int ST_WAVES = 15, ST_THREADS = 64 * ST_WAVES, ST_RAW = 8, ST_BMAX = 256;
constexpr int ST_PROD = 12;        // PC form: waves 0..11 transform columns, waves 12..14 emit frames

// PC = false: the workgroup alternates between transforming a phase's columns and emitting its frames, a barrier pair per phase.
// PC = true (round 5, second form): PRODUCER waves (0..11) take columns from the counter and never meet a barrier; CONSUMER waves (12..14)
// emit a member as soon as its jobs are counted done.  Hand-offs through LDS: producers add to sMemDone[member] behind each row (release),
// each consumer wave publishes the members it is done with in sEmitW (release); a producer about to overwrite a ring row waits until the last member whose window
// holds the row's previous column is emitted (need <= the member its own job belongs to, so the smallest blocked job always has its
// predecessors running: no circular wait); every wait is bounded (ST_SPIN_MAX polls) and a bound that expires raises *status and makes
// every wave of the workgroup leave -- never a hang; and never a wrong frame either: mel_stream_repair_kernel, launched behind this one,
// does the call again in the PC = false form (no waits) when *status is not 0 (tests force that with "frontend_stream_spin_max").
constexpr int ST_SPIN_MAX = 1 << 22;

template <int WIN, bool PC>
__device__ __forceinline__ void mel_stream_body(const FrontendConsts &c, const float *__restrict__ pcm, const int64_t *__restrict__ clip_off,
                                                const int64_t *__restrict__ clip_len, const int32_t *__restrict__ frame_clip,
                                                const int64_t *__restrict__ frame_start, const int32_t *__restrict__ prev,
                                                const int32_t *__restrict__ shift, int64_t n_frames, int B, int G,
                                                float *__restrict__ out, int *__restrict__ status, int spin_max, unsigned bid) {
    constexpr int HOP = WIN / 8;
    constexpr int BUF = WIN == 1024 ? MC8_BUF : WIN / 2;          // float2 per wave
    constexpr int NFFT = PC ? ST_PROD : ST_WAVES;                 // waves that transform
    constexpr int ST_RING = PC ? 152 : 128;                       // the consumers' FFT buffers become ring rows: producers run further ahead
    __shared__ f32x2 sFft[NFFT][BUF];
    __shared__ float2 sTw[WIN];
    __shared__ float sHamm[WIN];
    __shared__ int sBin0[128];
    __shared__ float sW8[8][128];
    __shared__ float sRing[ST_RING][128];     // mel rows of stream columns, slot = k mod ST_RING
    __shared__ float sRawRow[ST_RAW][128];    // column 0 of member i in slot i mod ST_RAW
    __shared__ short sNext[ST_BMAX], sShift[ST_BMAX], sMine[ST_BMAX], sMemN[ST_BMAX];      // local frame indices / hop shifts (< 256)
    __shared__ int sMemK[ST_BMAX];
    __shared__ int sMemDone[PC ? ST_BMAX : 1];                    // PC: finished jobs of each member
    __shared__ unsigned long long sMask[ST_BMAX / 64];
    __shared__ int sJ, sJob, sAbort;
    __shared__ int sEmitW[ST_WAVES - ST_PROD];                    // PC: members emitted by each consumer wave (a SUM over the waves would let a fast wave vouch for a slow one)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroup -> (block of B frames, slot g of G): XCD x (= blockIdx % 8, each with its own L2) takes the blocks b = x (mod 8), all G
    // slots of a block next to each other in its dispatch order -- the 12 chains of a block read the same stretch of PCM
    const int64_t wi = bid >> 3;
    const int g = (int)(wi % G);
    const int64_t b = (wi / G) * 8 + (bid & 7), n_base = b * B;
    if (n_base >= n_frames) return;

    for (int i = tid; i < WIN; i += ST_THREADS) { sTw[i] = c.twiddle[i]; sHamm[i] = c.hamm[i]; }
    for (int i = tid; i < 128; i += ST_THREADS) sBin0[i] = c.mel_bin0[i];
    for (int i = tid; i < 1024; i += ST_THREADS) sW8[i >> 7][i & 127] = c.mel_w8[i];
    // ---- the block's chain structure: heads (no predecessor inside the block) and successor links
    bool head = false;
    int pv = -1;
    if (tid < ST_BMAX) {
        const int64_t n = n_base + tid;
        const bool valid = tid < B && n < n_frames;
        pv = valid ? prev[n] : -1;
        sShift[tid] = (short)(valid ? shift[n] : 0);
        sNext[tid] = -1;
        head = valid && pv < n_base;                         // also pv = -1
        const unsigned long long m = __ballot(head);
        if (lane == 0) sMask[wave] = m;
    }
    __syncthreads();
    int n_mine = 0;
    {
        int total = 0;
#pragma unroll
        for (int w = 0; w < ST_BMAX / 64; ++w) total += __popcll(sMask[w]);
        n_mine = total > g ? (total - g + G - 1) / G : 0;
        if (tid < ST_BMAX && tid < B && n_base + tid < n_frames) {
            if (!head) sNext[pv - n_base] = (short)tid;      // unique: a frame has at most one successor (nearest aligned predecessor)
            else {
                int rank = __popcll(sMask[wave] & ((1ull << lane) - 1ull));
                for (int w = 0; w < wave; ++w) rank += __popcll(sMask[w]);
                if (rank % G == g) sMine[rank / G] = (short)tid;
            }
        }
    }
    __syncthreads();
    if (n_mine == 0) return;
    if (PC && tid == 0) sAbort = 0;

    const float c2[9] = {28.f / 462.f, 7.f / 462.f, -8.f / 462.f, -17.f / 462.f, -20.f / 462.f,
                         -17.f / 462.f, -8.f / 462.f, 7.f / 462.f, 28.f / 462.f};
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    f32x2 *buf = sFft[wave < NFFT ? wave : 0];

    // member `me` complete in the ring: Savitzky-Golay deltas + (T,F,C) store, exactly gather_features_kernel's, by threads t0, t0 + nt, ...
    auto emit = [&](int me, int t0, int nt, int64_t n_base_) {
        const int64_t frame = n_base_ + sMemN[me];
        const int kb = sMemK[me] % ST_RING;
        const float *raw_row = sRawRow[me & (ST_RAW - 1)];
        auto rowp = [&](int t) -> const float * {
            int sl = kb + t;
            sl = sl >= ST_RING ? sl - ST_RING : sl;
            return t == 0 ? raw_row : sRing[sl];
        };
        const unsigned long long ob = (unsigned long long)(out + frame * (64 * 128 * 3));
        const unsigned long long ouni = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ob >> 32)) << 32) |
                                        (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ob);
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)ouni, 0, 64 * 128 * 3 * 4, 0x00020000);
#define GF_ST(off, a_, b_, c_, d_) __builtin_amdgcn_raw_buffer_store_b128(u32x4{__builtin_bit_cast(unsigned, a_), __builtin_bit_cast(unsigned, b_), __builtin_bit_cast(unsigned, c_), __builtin_bit_cast(unsigned, d_)}, ors, (unsigned)(off), 0, 0)
        for (int i4 = t0; i4 < 64 * 128 / 4; i4 += nt) {
            const int t = i4 >> 5, f0 = (i4 & 31) * 4;
            const int tc = t < 4 ? 4 : (t > 59 ? 59 : t);
            float4 mv4[9];
#pragma unroll
            for (int j = -4; j <= 4; ++j) mv4[j + 4] = *reinterpret_cast<const float4 *>(rowp(tc + j) + f0);
            const float4 mt = *reinterpret_cast<const float4 *>(rowp(t) + f0);
            const float mq[4] = {mt.x, mt.y, mt.z, mt.w};
            float m[4], d1[4], d2[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = -4; j <= 4; ++j) {
                    const float4 v4 = mv4[j + 4];
                    const float mv = qq == 0 ? v4.x : (qq == 1 ? v4.y : (qq == 2 ? v4.z : v4.w));
                    s1 += (float)j * (1.0f / 60.0f) * mv;
                    s2 += c2[j + 4] * mv;
                }
                m[qq] = mq[qq]; d1[qq] = s1; d2[qq] = s2;
            }
            GF_ST(i4 * 48, m[0], d1[0], d2[0], m[1]);
            GF_ST(i4 * 48 + 16, d1[1], d2[1], m[2], d1[2]);
            GF_ST(i4 * 48 + 32, d2[2], m[3], d1[3], d2[3]);
        }
#undef GF_ST
    };

    for (int hi = 0; hi < n_mine; ++hi) {
        if (tid == 0) {                                      // the segment's members: local frame index and first stream column
            int i = 0, cur = sMine[hi], k = 0;
            for (;;) {
                sMemN[i] = (short)cur; sMemK[i] = k; ++i;
                const int nx = sNext[cur];
                if (nx < 0) break;
                k += sShift[nx];
                cur = nx;
            }
            sJ = i;
            sJob = 0;
            for (int w = 0; w < ST_WAVES - ST_PROD; ++w) sEmitW[w] = 0;
        }
        if (PC && tid < ST_BMAX) sMemDone[tid] = 0;
        __syncthreads();
        const int J = sJ;
        // jobs in stream order: member 0 = its raw column 0 + shared columns 1..63; member i >= 1 = its raw column 0 + the d_i new shared
        // columns k_(i-1) + 64 .. k_i + 63.  jstart(i) = first job of member i.
        auto jstart = [&](int i) { return i == 0 ? 0 : 63 + i + sMemK[i - 1]; };
        const int total = 63 + J + sMemK[J - 1];
        const int64_t n0 = n_base + sMemN[0];
        const int clip = frame_clip[n0];
        const int64_t len64 = clip_len[clip], p0 = frame_start[n0];
        const int len = (int)(len64 > 0x1fffffff ? 0x1fffffff : len64);
        const __amdgpu_buffer_rsrc_t xrs = clip_buffer(pcm + clip_off[clip], len);
        // one column job: (member, raw?, stream column) of job q; the wave's jobs come in increasing order, so mi only moves forward
        auto job_of = [&](int q, int &mi, bool &raw, int &k) {
            while (mi + 1 < J && jstart(mi + 1) <= q) ++mi;
            const int o = q - jstart(mi);
            raw = o == 0;
            k = raw ? sMemK[mi] : (mi == 0 ? o : sMemK[mi - 1] + 63 + o);
        };
        auto transform = [&](int mi, bool raw, int k) {
            float mel[2];
            if constexpr (WIN == 1024) column_mel_r8(xrs, len, p0 + (int64_t)k * HOP, raw, buf, sTw, sHamm, sBin0, sW8, lane, mel);
            else column_mel_r4<WIN>(xrs, len, p0 + (int64_t)k * HOP, raw, reinterpret_cast<float2 *>(buf), sTw, sHamm, sBin0, sW8, lane, mel);
            float *dst = raw ? sRawRow[mi & (ST_RAW - 1)] : sRing[k % ST_RING];
            dst[lane] = mel[0];
            dst[64 + lane] = mel[1];
        };
        if constexpr (PC) {
            bool dead = false;
            if (wave < ST_PROD) {
                int mi = 0, ri = -1;
                for (;;) {
                    int q = 0;
                    if (lane == 0) q = atomicAdd(&sJob, 1);
                    q = __builtin_amdgcn_readfirstlane(q);
                    if (q >= total) break;
                    bool raw; int k;
                    job_of(q, mi, raw, k);
                    // ring space: the row's previous column (k - ST_RING) is read last by the largest member i with k_i <= k - ST_RING;
                    // a raw row's slot by member mi - ST_RAW
                    int need;
                    if (raw) need = mi - ST_RAW + 1;
                    else { while (ri + 1 < J && sMemK[ri + 1] <= k - ST_RING) ++ri; need = ri + 1; }
                    if (need > 0) {
                        int spin = 0;
                        auto emitted = [&]() {                 // members that EVERY consumer wave is done with
                            int m = __hip_atomic_load(&sEmitW[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                            for (int w = 1; w < ST_WAVES - ST_PROD; ++w) m = min(m, __hip_atomic_load(&sEmitW[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                            return m;
                        };
                        while (emitted() < need) {
                            if (++spin > spin_max || __hip_atomic_load(&sAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { dead = true; break; }
                            __builtin_amdgcn_s_sleep(2);
                        }
                        if (dead) break;
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                    transform(mi, raw, k);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the row is in LDS before the count moves
                    if (lane == 0) atomicAdd(&sMemDone[mi], 1);
                }
            } else {
                const int ct0 = (wave - ST_PROD) * 64 + lane, cnt = (ST_WAVES - ST_PROD) * 64;
                for (int me = 0; me < J; ++me) {
                    const int njobs = (me + 1 < J ? jstart(me + 1) : total) - jstart(me);
                    int spin = 0;
                    while (__hip_atomic_load(&sMemDone[me], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < njobs) {
                        if (++spin > spin_max || __hip_atomic_load(&sAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { dead = true; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (dead) break;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    emit(me, ct0, cnt, n_base);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's reads of the member's rows are done
                    if (lane == 0) __hip_atomic_store(&sEmitW[wave - ST_PROD], me + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            if (dead && lane == 0) {                         // a bound expired (never, unless the hand-off logic is wrong): say so, let everyone leave
                __hip_atomic_store(&sAbort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (status) atomicAdd(status, 1);
            }
        } else {
            // PHASES: members em..e whose columns fit the ring together (k_e + 64 - k_em <= ST_RING: three at 25 hops apart) are produced
            // in one go, then the workgroup meets and emits them.  Inside a phase a wave takes the NEXT job from a counter in LDS whenever
            // it is free -- no barrier, no fixed job-to-wave map: in lock step (one column per wave, then a barrier) the waves of a SIMD all
            // sat in the same LDS round trip at the same time (4.7 k cycles per column against 2.6 k in mel_columns_r8_kernel), and with a
            // fixed map the SIMD that hosts one wave more than the others finished last while they idled.
            int em = 0;
            while (em < J) {
                int e = em;
                while (e + 1 < J && e + 1 - em < ST_RAW && sMemK[e + 1] + 64 - sMemK[em] <= ST_RING) ++e;
                const int qend = e + 1 < J ? jstart(e + 1) : total;
                int mi = em;
                for (;;) {
                    int q = 0;
                    if (lane == 0) q = atomicAdd(&sJob, 1);
                    q = __builtin_amdgcn_readfirstlane(q);                  // wave-uniform; a wave's jobs still come in increasing order
                    if (q >= qend) break;
                    bool raw; int k;
                    job_of(q, mi, raw, k);
                    transform(mi, raw, k);
                }
                __syncthreads();
                for (int me = em; me <= e; ++me) emit(me, tid, ST_THREADS, n_base);
                em = e + 1;
                if (em < J) {
                    if (tid == 0) sJob = qend;               // (every wave left the job loop before the barrier above)
                    __syncthreads();                         // the next phase's columns overwrite rows this emission read
                }
            }
        }
        __syncthreads();                                     // sMem* / sJ are rewritten for the next segment
    }
}

template <int WIN, bool PC>
__global__ __launch_bounds__(ST_THREADS) void mel_stream_kernel(FrontendConsts c, const float *__restrict__ pcm, const int64_t *__restrict__ clip_off,
                                                                const int64_t *__restrict__ clip_len, const int32_t *__restrict__ frame_clip,
                                                                const int64_t *__restrict__ frame_start, const int32_t *__restrict__ prev,
                                                                const int32_t *__restrict__ shift, int64_t n_frames, int B, int G,
                                                                float *__restrict__ out, int *__restrict__ status, int spin_max) {
    mel_stream_body<WIN, PC>(c, pcm, clip_off, clip_len, frame_clip, frame_start, prev, shift, n_frames, B, G, out, status, spin_max, blockIdx.x);
}

// REPAIR PASS, behind every launch of the producer / consumer form, in stream order (the principle of time_lstm_repair_kernel): one load of
// the status word; a healthy call's workgroups (one per CU) leave at once -- a few microseconds.  If a hand-off wait of that launch
// expired (never, unless the hand-off logic is wrong; the tests force it) its frames may be incomplete: the whole call is done again
// in the barrier form, which has no waits that can expire -- the same columns, the same filters, the same bits.  So the features a
// caller reads are right either way; the status word only counts.  (The repair inside the producer / consumer kernel itself -- redo
// only the segment -- was built first: it is the same few lines, and its second copy of the transform / emit code made the healthy
// path 6 % slower, 0.97 -> 1.03 ms.)
template <int WIN>
__global__ __launch_bounds__(ST_THREADS) void mel_stream_repair_kernel(FrontendConsts c, const float *__restrict__ pcm, const int64_t *__restrict__ clip_off,
                                                                       const int64_t *__restrict__ clip_len, const int32_t *__restrict__ frame_clip,
                                                                       const int64_t *__restrict__ frame_start, const int32_t *__restrict__ prev,
                                                                       const int32_t *__restrict__ shift, int64_t n_frames, int B, int G,
                                                                       float *__restrict__ out, const int *__restrict__ gate, unsigned n_wg) {
    if (__builtin_amdgcn_readfirstlane(*gate) == 0) return;
    for (unsigned bid = blockIdx.x; bid < n_wg; bid += gridDim.x) {
        mel_stream_body<WIN, false>(c, pcm, clip_off, clip_len, frame_clip, frame_start, prev, shift, n_frames, B, G, out, nullptr, 0, bid);
        __syncthreads();                                     // the next segment's set-up rewrites the LDS tables
    }
}

}  // namespace

extern thread_local int g_sdfa_mel_fft_radix4;   // api.cpp ("mel_fft_radix4"): 1 = the radix-4 / LDS-staged column FFT of rounds 2-3 at 16 kHz too

hipError_t sdfa_launch_frontend(const FrontendConsts &c, const float *pcm, const int64_t *clip_off,
                                const int64_t *clip_len, const int32_t *frame_clip, const int64_t *frame_start,
                                int64_t n_frames, float *audio_feat, hipStream_t s) {
    if (n_frames <= 0) return hipSuccess;
    if (c.nbins_used > 256) return hipErrorInvalidValue;
    if (c.win == 1024)
        hipLaunchKernelGGL(frontend_kernel<1024>, dim3((unsigned)n_frames), dim3(FE_THREADS), 0, s, c, pcm, clip_off, clip_len,
                           frame_clip, frame_start, audio_feat);
    else if (c.win == 512)
        hipLaunchKernelGGL(frontend_kernel<512>, dim3((unsigned)n_frames), dim3(FE_THREADS), 0, s, c, pcm, clip_off, clip_len,
                           frame_clip, frame_start, audio_feat);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t sdfa_launch_mel_columns(const FrontendConsts &c, const float *pcm, const int64_t *clip_off, const int64_t *clip_len,
                                   const int32_t *frame_clip, const int64_t *frame_start, const int32_t *col_src,
                                   const int64_t *n_distinct, float *mel_table, hipStream_t s) {
    if (c.nbins_used > 256) return hipErrorInvalidValue;
    const unsigned grid = 256 * 6;      // persistent waves: each takes column pairs round-robin until the device-side count runs out
    if (c.win == 1024 && !g_sdfa_mel_fft_radix4)
        hipLaunchKernelGGL(mel_columns_r8_kernel, dim3(grid), dim3(64 * MC_WAVES), 0, s, c, pcm, clip_off, clip_len, frame_clip, frame_start, col_src,
                           n_distinct, mel_table);
    else if (c.win == 1024)
        hipLaunchKernelGGL(mel_columns_kernel<1024>, dim3(grid), dim3(64 * MC_WAVES), 0, s, c, pcm, clip_off, clip_len, frame_clip,
                           frame_start, col_src, n_distinct, mel_table);
    else if (c.win == 512)
        hipLaunchKernelGGL(mel_columns_kernel<512>, dim3(grid), dim3(64 * MC_WAVES), 0, s, c, pcm, clip_off, clip_len, frame_clip,
                           frame_start, col_src, n_distinct, mel_table);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// The spectral-stream form: one launch behind share_prev_kernel (share.hip).  block = frames per chain-segment block (multiple of 4, <= 256;
// 0 = the default 144: 12 members per segment at 60 fps -- measured against 96 / 192 / 240, profiles/r05_ab_frontend.txt), slots = workgroups
// per block (0 = 12: one per chain at 60 fps).
hipError_t sdfa_launch_mel_stream(const FrontendConsts &c, const float *pcm, const int64_t *clip_off, const int64_t *clip_len,
                                  const int32_t *frame_clip, const int64_t *frame_start, const int32_t *prev, const int32_t *shift,
                                  int64_t n_frames, int block, int slots, int producer_consumer, int spin_max, int *status, float *audio_feat, hipStream_t s) {
    if (n_frames <= 0) return hipSuccess;
    if (c.nbins_used > 256) return hipErrorInvalidValue;
    // Frames per block: 144 (12 members per chain segment at 60 fps: measured optimum on the 20,352-frame batch against 96 / 192 / 240,
    // profiles/r05_ab_frontend.txt) while that still gives every CU two workgroups; below, shorter segments (more redundant columns per
    // frame, but the call is latency-bound there and idle CUs are worse), not under 48.
    int B = block;
    const int G = slots > 0 ? slots : 12;
    if (B <= 0) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        B = 144;
        if ((n_frames + B - 1) / B * G < 2 * (int64_t)cus) {
            const int64_t b = n_frames * G / (2 * (int64_t)cus) / 12 * 12;
            B = (int)(b < 48 ? 48 : (b > 144 ? 144 : b));
        }
    }
    if (B > ST_BMAX || G > ST_BMAX) return hipErrorInvalidValue;
    const int64_t nblocks = (n_frames + B - 1) / B, nb8 = (nblocks + 7) / 8 * 8;
    const dim3 grid((unsigned)(nb8 * G));
#define ST_LAUNCH(W, P) hipLaunchKernelGGL((mel_stream_kernel<W, P>), grid, dim3(ST_THREADS), 0, s, c, pcm, clip_off, clip_len, frame_clip, frame_start, prev, \
                                           shift, n_frames, B, G, audio_feat, status, spin_max > 0 ? spin_max : ST_SPIN_MAX)
    if (c.win == 1024) { if (producer_consumer) ST_LAUNCH(1024, true); else ST_LAUNCH(1024, false); }
    else if (c.win == 512) { if (producer_consumer) ST_LAUNCH(512, true); else ST_LAUNCH(512, false); }
    else return hipErrorInvalidValue;
#undef ST_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !producer_consumer || !status) return e;
    // the repair pass (exits at once unless a hand-off wait of the launch above expired)
    const unsigned n_wg = grid.x, rgrid = n_wg < 256u ? n_wg : 256u;
#define ST_REPAIR(W) hipLaunchKernelGGL((mel_stream_repair_kernel<W>), dim3(rgrid), dim3(ST_THREADS), 0, s, c, pcm, clip_off, clip_len, frame_clip, frame_start, \
                                        prev, shift, n_frames, B, G, audio_feat, status, n_wg)
    if (c.win == 1024) ST_REPAIR(1024); else ST_REPAIR(512);
#undef ST_REPAIR
    return hipGetLastError();
}

extern thread_local int g_sdfa_gather_plain_order;   // api.cpp ("gather_plain_order"): 1 = workgroup b takes frame b (rounds 2-3)

hipError_t sdfa_launch_gather_features(const float *mel_table, const int32_t *col_to_u, int64_t n_frames, int64_t Nc, int frame_major,
                                       float *audio_feat, hipStream_t s) {
    if (n_frames <= 0) return hipSuccess;
    const int chain = g_sdfa_gather_plain_order ? 0 : 1;
    // chain order: 8 XCD ranges of ceil(12 * ceil(F / 12) / 8) chain positions each (gather_frame_of_block); blocks past the end exit
    const int64_t total = (n_frames + 11) / 12 * 12, span = (total + 7) / 8;
    const int64_t grid = chain ? span * 8 : n_frames;
    hipLaunchKernelGGL(gather_features_kernel, dim3((unsigned)grid), dim3(256), 0, s, reinterpret_cast<const float4 *>(mel_table),
                       col_to_u, Nc, n_frames, frame_major, chain, audio_feat);
    return hipGetLastError();
}

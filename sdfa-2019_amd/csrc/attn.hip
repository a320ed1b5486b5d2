// Bahdanau attention tail (speech_anime/layers/attentions.py:92-124,69-75) and small layout kernels.
//
// The dense contractions of the attention layer (query Conv1d, proj_qry, proj_key) run on the MFMA GEMM
// (gemm.hip); this kernel does what is left per frame n:
//     score[t] = v . tanh(qp[:, n] + kp[:, t, n] + b)          t = 0..63
//     align    = softmax_t(score * 1.0)                         (scale_score_at_eval = 1.0)
//     ctx      = sum_t align[t] * x[:, t, n]                    (torch.bmm(align, value))
// One workgroup = 16 consecutive frames (Nc / 16 workgroups: 512 for an 8192-frame chunk, two per CU), lane = (frame,
// part p = 0..3).  The kernel is a stream over H (128 KiB per frame) and KP (32 KiB per frame): every global access is a
// 256-byte run of the K4 [feature/4][t*Nc + n] arrays per quarter wave, 8-16 independent requests in flight per lane.
//   scores : wave w owns time steps 16w..16w+15 and ALL 128 units -- each quarter wave sums 32 of them, two __shfl_xor
//            steps combine the quarters, so a score is complete inside its wave (no partial sums through LDS);
//   softmax: wavefront-reduced, once: each wave keeps its 16 scores in registers, reduces (max, sum of exp) over them,
//            and the four (max, sum) pairs per frame meet in LDS; align = exp(s - M) / sum over waves of sum_w * exp(m_w - M);
//   context: the 64 weights of a frame go through LDS once; wave w / part p accumulates feature quads 32w+8p .. +7.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int ATT_FR = 16;   // frames per workgroup

__global__ __launch_bounds__(256, 2) void attn_kernel(AttnArgs a) {
    __shared__ float sAlign[64][ATT_FR];     // [t][frame]
    __shared__ float2 sStat[4][ATT_FR];      // per wave: (max, sum of exp) over its 16 time steps

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, part = lane >> 4;
    const int64_t n = (int64_t)blockIdx.x * ATT_FR + fr;
    const float4 *__restrict__ KP = reinterpret_cast<const float4 *>(a.KP);
    const float4 *__restrict__ QP = reinterpret_cast<const float4 *>(a.QP);
    const float4 *__restrict__ H = reinterpret_cast<const float4 *>(a.H);

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
    hipLaunchKernelGGL(attn_kernel, dim3((unsigned)(a.Nc / ATT_FR)), dim3(256), 0, s, a);
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

// Next row after the hot path: dgrad -> mesh (deformation transfer solve), batched over animation frames.
//
// Reference (native C++/Eigen): deformation/cpp/src/deform_triangle_impl.hpp:215-310 getMeshFromDeformationGradients,
// called once per video frame from speech_anime/viewer/frame.py:102-141.  Per triangle j the 9-vector is
//     S = I + sym(d0 d1 d2; . d3 d4; . . d5),  log R = [[0 d6 d7] [-d6 0 d8] [-d7 -d8 0]],  T_j = exp(log R) S
// (rotation/utils_rotation.cpp:33-49: Rodrigues on the angle-scaled skew matrix), the stacked T_j^T is the right-hand
// side of the least-squares system  A^T A x = A^T (T - Ar c)  over the free vertices (A from the template's
// per-triangle pseudo-inverses, set_target, :8-140), constrained vertices are copied through.
//
// MI355X form: the system matrix is constant per template, so its inverse is formed once on the host (dense
// Cholesky in fp64, 1261 free vertices for FLAME) and the per-frame work becomes
//   (1) mesh_rhs_kernel : per (frame, free-vertex quad) gather over incident triangles of  T_j c_{j,v}  with T_j
//       built on the fly from 9 floats -- written straight in K4 layout;  the identity part is subtracted, i.e. the
//       solve is done for the DISPLACEMENT from the template (zero dgrad gives the template exactly and the fp32
//       error scales with the deformation, not with the 2e4-magnitude absolute right-hand side);
//   (2) the fp32 MFMA GEMM of gemm.hip:  X[free][3*frames] = Inv[free][free] * RHS[free][3*frames];
//   (3) mesh_scatter_kernel : template + displacement for free vertices, constraints copied, [frame][vertex][3].
#include "common.h"
#include "kernels.h"
#include <algorithm>

namespace {

__device__ __forceinline__ void transform_minus_identity(const float *__restrict__ d, float T[3][3]) {
    // R = exp(log R): angle = |(-d8, d7, -d6)|, K = logR / angle, R = I + sin(a) K + (1 - cos a) K^2
    const float k01 = d[6], k02 = d[7], k12 = d[8];
    const float a2 = k01 * k01 + k02 * k02 + k12 * k12;
    float R[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
    if (a2 > 1e-20f) {                          // reference: angle < 1e-10 -> identity
        const float a = sqrtf(a2), s = sinf(a) / a, sh = sinf(0.5f * a) / (0.5f * a), c = 0.5f * sh * sh;   // (1 - cos a) / a^2 without cancellation
        // K = [[0 k01 k02] [-k01 0 k12] [-k02 -k12 0]] (unnormalised); K^2 entries
        const float K2[3][3] = {{-(k01 * k01 + k02 * k02), -k02 * k12, k01 * k12},
                                {-k02 * k12, -(k01 * k01 + k12 * k12), -k01 * k02},
                                {k01 * k12, -k01 * k02, -(k02 * k02 + k12 * k12)}};
        const float K[3][3] = {{0.f, k01, k02}, {-k01, 0.f, k12}, {-k02, -k12, 0.f}};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) R[i][j] += s * K[i][j] + c * K2[i][j];
    }
    const float S[3][3] = {{d[0] + 1.f, d[1], d[2]}, {d[1], d[3] + 1.f, d[4]}, {d[2], d[4], d[5] + 1.f}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float v = R[i][0] * S[0][j] + R[i][1] * S[1][j] + R[i][2] * S[2][j];
            T[i][j] = v - (i == j ? 1.f : 0.f);
        }
}

// One triangle's 9-vector of output frame `frame`.  Plain: row `frame` of dgrad.  Seek (a.seek_src != null): the frame is
// the blend  wa * row[src0] + wb * row[src1]  of saber.stream.seek (saber/data/stream/stream.py:41-46) -- three
// separately rounded fp32 operations (numpy evaluates a * x, (1-a) * y and the sum as three float32 array ops), so no FMA
// contraction: the blended dgrad never exists in memory, and is bit-identical to what seek_rows_kernel would write.
__device__ __forceinline__ void load_dgrad9(const MeshArgs &a, int64_t frame, int j, float d[9]) {
    if (!a.seek_src) {
        const float *__restrict__ p = a.dgrad + (frame * (int64_t)a.n_src_tris + j) * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) d[i] = p[i];
        return;
    }
    const int64_t r0 = a.seek_src[2 * frame], r1 = a.seek_src[2 * frame + 1];
    const float wa = a.seek_w[2 * frame], wb = a.seek_w[2 * frame + 1];
    const float *__restrict__ p0 = a.dgrad + (r0 * (int64_t)a.n_src_tris + j) * 9;
    const float *__restrict__ p1 = a.dgrad + (r1 * (int64_t)a.n_src_tris + j) * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = fadd_exact(fmul_exact(wa, p0[i]), fmul_exact(wb, p1[i]));
}

// RHS'[v][frame*3 + comp] = sum over incidences (equation k -> source triangle j, coefficients c) of vertex v of
// ((T_j - I) c)[comp], minus reg * template[v][comp] (the regulariser acts on the absolute position, the solve on the
// displacement).  Equations whose transform is the identity (target triangles without a correspondence,
// deform_triangle_impl.hpp:262-266) contribute nothing and are not in the incidence lists.
__global__ __launch_bounds__(256) void mesh_rhs_kernel(MeshArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nquad = a.free_pad / 4;
    if (idx >= nquad * a.n_frames) return;
    const int64_t frame = idx / nquad;
    const int vq = (int)(idx % nquad);
    float acc[4][3];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e][0] = acc[e][1] = acc[e][2] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int v = vq * 4 + e;
        if (v >= a.n_free) continue;
        for (int p = a.inc_ptr[v]; p < a.inc_ptr[v + 1]; ++p) {
            const int j = a.inc_tri[p];
            const float c0 = a.inc_coef[3 * p], c1 = a.inc_coef[3 * p + 1], c2 = a.inc_coef[3 * p + 2];
            float d[9], T[3][3];
            load_dgrad9(a, frame, j, d);
            transform_minus_identity(d, T);
            acc[e][0] += T[0][0] * c0 + T[0][1] * c1 + T[0][2] * c2;
            acc[e][1] += T[1][0] * c0 + T[1][1] * c1 + T[1][2] * c2;
            acc[e][2] += T[2][0] * c0 + T[2][1] * c1 + T[2][2] * c2;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[e][c] -= a.reg_xt[3 * v + c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        st4(a.rhs + ((int64_t)vq * a.ld + frame * 3 + c) * 4, make_float4(acc[0][c], acc[1][c], acc[2][c], acc[3][c]));
}

// ---------------------------------------------------------------------------------------------------------------
// saber.stream.seek on the device (saber/data/stream/stream.py:20-46; caller speech_anime/model/model.py:204-212):
// query i of a clip is ts = i * 1000.0 / fps (float64, like the Python expression); binary search for the animation frame
// m with tslist[m] <= ts < tslist[m+1]; before the first / after the last timestamp, or on the last frame, the row is
// copied; otherwise  a = (t[m+1] - ts) / (t[m+1] - t[m])  in float64, and the row is  float32(a) * row[m] +
// float32(1 - a) * row[m+1].  The plan holds global row indices (src0, src1) and the two float32 weights per query;
// a copy is (m, m, 1, 0).
__global__ __launch_bounds__(256) void seek_plan_kernel(const int32_t *__restrict__ tslist, const int64_t *__restrict__ frame_off,
                                                        const int64_t *__restrict__ query_off, int n_clips, double fps,
                                                        int64_t *__restrict__ src, float *__restrict__ w) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= query_off[n_clips]) return;
    int lo = 0, hi = n_clips;               // clip c with query_off[c] <= q < query_off[c+1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (query_off[mid] <= q) lo = mid; else hi = mid; }
    const int64_t f0 = frame_off[lo], n = frame_off[lo + 1] - f0;
    const int32_t *__restrict__ t = tslist + f0;
    const double ts = dmul_exact((double)(q - query_off[lo]), 1000.0) / fps;
    int64_t m;
    float wa = 1.f, wb = 0.f;
    bool blend = false;
    if (ts < (double)t[0]) m = 0;
    else if (ts > (double)t[n - 1]) m = n - 1;
    else {
        int64_t l = 0, r = n;               // last m with t[m] <= ts
        while (r - l > 1) { const int64_t mid = (l + r) >> 1; if ((double)t[mid] <= ts) l = mid; else r = mid; }
        m = l;
        if (m + 1 < n) {
            const double a = dsub_exact((double)t[m + 1], ts) / (double)(t[m + 1] - t[m]);
            wa = (float)a; wb = (float)dsub_exact(1.0, a);
            blend = true;
        }
    }
    src[2 * q] = f0 + m; src[2 * q + 1] = f0 + (blend ? m + 1 : m);
    w[2 * q] = wa; w[2 * q + 1] = wb;
}

// out[q][c] = wa * rows[src0][c] + wb * rows[src1][c]  (three fp32 roundings, see load_dgrad9); VEC = 4: 16-byte accesses
template <int VEC>
__global__ __launch_bounds__(256) void seek_rows_kernel(const float *__restrict__ rows, int64_t width, const int64_t *__restrict__ src,
                                                        const float *__restrict__ w, int64_t nq, float *__restrict__ out) {
    const int64_t per = width / VEC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq * per; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = i / per, c = (i % per) * VEC;
        const float wa = w[2 * q], wb = w[2 * q + 1];
        const float *__restrict__ p0 = rows + src[2 * q] * width + c, *__restrict__ p1 = rows + src[2 * q + 1] * width + c;
        if (VEC == 4) {
            const float4 x = ld4(p0), y = ld4(p1);
            st4(out + q * width + c, make_float4(fadd_exact(fmul_exact(wa, x.x), fmul_exact(wb, y.x)), fadd_exact(fmul_exact(wa, x.y), fmul_exact(wb, y.y)),
                                                 fadd_exact(fmul_exact(wa, x.z), fmul_exact(wb, y.z)), fadd_exact(fmul_exact(wa, x.w), fmul_exact(wb, y.w))));
        } else {
            out[q * width + c] = fadd_exact(fmul_exact(wa, p0[0]), fmul_exact(wb, p1[0]));
        }
    }
}

// Test-time ensembling (speech_anime/model/model.py:369-403): `anime_sum += second_pass; anime_sum / 2.0` on float32 arrays =
// one rounded add and one (exact) division per element.  In place when out == a.
// VEC = 4: 16-byte aligned operands; VEC = 1: any 4-byte alignment (the offsets head's 60,276-byte rows put the second pass of a
// launch group at a 16-byte boundary only every fourth row).
template <int VEC>
__global__ __launch_bounds__(256) void ensemble_mean_kernel(const float *a, const float *__restrict__ b, int64_t n, float *out) {
    if (VEC == 4) {
        const int64_t nq = n / 4;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
            const float4 x = ld4(a + 4 * i), y = ld4(b + 4 * i);
            st4(out + 4 * i, make_float4(__fdiv_rn(fadd_exact(x.x, y.x), 2.0f), __fdiv_rn(fadd_exact(x.y, y.y), 2.0f),
                                         __fdiv_rn(fadd_exact(x.z, y.z), 2.0f), __fdiv_rn(fadd_exact(x.w, y.w), 2.0f)));
        }
        if (blockIdx.x == 0 && threadIdx.x < n % 4) {
            const int64_t i = nq * 4 + threadIdx.x;
            out[i] = __fdiv_rn(fadd_exact(a[i], b[i]), 2.0f);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            out[i] = __fdiv_rn(fadd_exact(a[i], b[i]), 2.0f);
    }
}

__global__ __launch_bounds__(256) void mesh_scatter_kernel(MeshArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)a.n_verts * a.n_frames) return;
    const int64_t frame = idx / a.n_verts;
    const int v = (int)(idx % a.n_verts);
    const int col = a.vert_col[v];       // >= 0: free-vertex row, < 0: constrained
    float x = a.tmpl[3 * v], y = a.tmpl[3 * v + 1], z = a.tmpl[3 * v + 2];
    if (col >= 0) {
        const float *p = a.sol + ((int64_t)(col >> 2) * a.ld + frame * 3) * 4 + (col & 3);
        x += p[0]; y += p[4]; z += p[8];
    }
    float *o = a.verts + idx * 3;
    o[0] = x; o[1] = y; o[2] = z;
}

}  // namespace

hipError_t sdfa_launch_mesh_rhs(const MeshArgs &a, hipStream_t s) {
    const int64_t n = a.free_pad / 4 * a.n_frames;
    hipLaunchKernelGGL(mesh_rhs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t sdfa_launch_seek_plan(const int32_t *tslist, const int64_t *frame_off, const int64_t *query_off, int n_clips, double fps,
                                 int64_t n_queries, int64_t *src, float *w, hipStream_t s) {
    hipLaunchKernelGGL(seek_plan_kernel, dim3((unsigned)((n_queries + 255) / 256)), dim3(256), 0, s, tslist, frame_off, query_off, n_clips, fps, src, w);
    return hipGetLastError();
}

hipError_t sdfa_launch_seek_rows(const float *rows, int64_t width, const int64_t *src, const float *w, int64_t nq, float *out, hipStream_t s) {
    const bool vec = width % 4 == 0 && (((uintptr_t)rows | (uintptr_t)out) & 15) == 0;
    const int64_t n = nq * (vec ? width / 4 : width);
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 256 * 32);
    if (vec) hipLaunchKernelGGL(seek_rows_kernel<4>, dim3(grid), dim3(256), 0, s, rows, width, src, w, nq, out);
    else hipLaunchKernelGGL(seek_rows_kernel<1>, dim3(grid), dim3(256), 0, s, rows, width, src, w, nq, out);
    return hipGetLastError();
}

hipError_t sdfa_launch_ensemble_mean(const float *a, const float *b, int64_t n, float *out, hipStream_t s) {
    const bool vec = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(((vec ? n / 4 : n) + 255) / 256, 256 * 32));
    if (vec) hipLaunchKernelGGL(ensemble_mean_kernel<4>, dim3(grid), dim3(256), 0, s, a, b, n, out);
    else hipLaunchKernelGGL(ensemble_mean_kernel<1>, dim3(grid), dim3(256), 0, s, a, b, n, out);
    return hipGetLastError();
}

hipError_t sdfa_launch_mesh_scatter(const MeshArgs &a, hipStream_t s) {
    const int64_t n = (int64_t)a.n_verts * a.n_frames;
    hipLaunchKernelGGL(mesh_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

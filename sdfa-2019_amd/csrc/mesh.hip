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

// RHS'[v][frame*3 + comp] = sum over incidences (j, c) of vertex v of ((T_j - I) c)[comp]
__global__ __launch_bounds__(256) void mesh_rhs_kernel(MeshArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nquad = a.free_pad / 4;
    if (idx >= nquad * a.n_frames) return;
    const int64_t frame = idx / nquad;
    const int vq = (int)(idx % nquad);
    float acc[4][3];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e][0] = acc[e][1] = acc[e][2] = 0.f;
    const float *__restrict__ dg = a.dgrad + frame * (int64_t)a.n_tris * 9;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int v = vq * 4 + e;
        if (v >= a.n_free) continue;
        for (int p = a.inc_ptr[v]; p < a.inc_ptr[v + 1]; ++p) {
            const int j = a.inc_tri[p];
            const float c0 = a.inc_coef[3 * p], c1 = a.inc_coef[3 * p + 1], c2 = a.inc_coef[3 * p + 2];
            float T[3][3];
            transform_minus_identity(dg + (int64_t)j * 9, T);
            acc[e][0] += T[0][0] * c0 + T[0][1] * c1 + T[0][2] * c2;
            acc[e][1] += T[1][0] * c0 + T[1][1] * c1 + T[1][2] * c2;
            acc[e][2] += T[2][0] * c0 + T[2][1] * c1 + T[2][2] * c2;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        st4(a.rhs + ((int64_t)vq * a.ld + frame * 3 + c) * 4, make_float4(acc[0][c], acc[1][c], acc[2][c], acc[3][c]));
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

hipError_t sdfa_launch_mesh_scatter(const MeshArgs &a, hipStream_t s) {
    const int64_t n = (int64_t)a.n_verts * a.n_frames;
    hipLaunchKernelGGL(mesh_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

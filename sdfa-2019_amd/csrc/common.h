// Shared device helpers for the gfx950 kernels.
//
// Activation layout ("K4"): a [F features][M columns] matrix is stored as float4[F/4][M]:
// element (f, m) lives at ((f/4)*M + m)*4 + (f%4).  A column is one (animation frame, STFT
// time step) pair, m = t*Nc + n for frame n of a chunk of Nc frames.  K4 is what the 32x32
// fp32 MFMA accumulator produces for free (each lane owns 4 consecutive rows per register
// quad) and what its operands want (one 16-byte LDS read feeds four MFMA k-steps).
//
// MFMA used everywhere: v_mfma_f32_32x32x2_f32  (D[32x32] += A[32x2] * B[2x32], exact fp32).
//   lane l:  A[i = l&31][k = l>>5]   B[k = l>>5][j = l&31]
//   D reg r: row i = (r&3) + 8*(r>>2) + 4*(l>>5), col j = l&31
// With operands in K4 a lane in half h = l>>5 reads the float4 of k-quad (2*kb + h); the four
// MFMAs of k-block kb then contract k = 8kb+{0..3} (half 0) with k = 8kb+4+{0..3} (half 1) --
// the same pairing an accumulator quad has, so a D tile can be fed back as a B operand
// straight from registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

#ifdef SDFA_OPERAND_TERMS
// ANALYSIS BUILD ONLY (make TERMS=n, tools/precision_sweep.py): every MFMA operand is first reduced to the sum of
// its n leading bfloat16 terms (round-to-nearest-even each), i.e. what a split-bf16 MFMA path with n operand
// planes would see.  The product library is built without this macro.
__device__ __forceinline__ float bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xFFFF0000u);
}
__device__ __forceinline__ float keep_terms(float x) {
    float s = bf16_rne(x);
#pragma unroll
    for (int t = 1; t < SDFA_OPERAND_TERMS; ++t) s += bf16_rne(x - s);
    return s;
}
#define SDFA_OP(x) keep_terms(x)
#else
#define SDFA_OP(x) (x)
#endif

__device__ __forceinline__ void mfma4(f32x16 &acc, const float4 &a, const float4 &b) {
    acc = MFMA(SDFA_OP(a.x), SDFA_OP(b.x), acc);
    acc = MFMA(SDFA_OP(a.y), SDFA_OP(b.y), acc);
    acc = MFMA(SDFA_OP(a.z), SDFA_OP(b.z), acc);
    acc = MFMA(SDFA_OP(a.w), SDFA_OP(b.w), acc);
}

__device__ __forceinline__ float f4c(const float4 &v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; }

// One k-quad of A (M row tiles) against one k-quad of B (N column tiles) into an M x N block of accumulators, issued
// COMPONENT-major: consecutive MFMAs go to different accumulators.  A 32x32x2 MFMA that reads the accumulator the
// previous one writes cannot start until that one has drained (16 passes + write-back), which a lone wave feels as
// ~10 % lost issue slots (tools/mfma_peak3.hip: 138 vs 155 TFLOP/s); with M*N >= 4 independent tiles in between it does not.
template <int M, int N>
__device__ __forceinline__ void mfma_block(f32x16 (&acc)[M][N], const float4 (&a)[M], const float4 (&b)[N]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) acc[i][j] = MFMA(SDFA_OP(f4c(a[i], q)), SDFA_OP(f4c(b[j], q)), acc[i][j]);
}

// Single IEEE operations that must NOT be contracted into an FMA: code that reproduces numpy / Python arithmetic op by op
// (stream.seek's  a * x + (1 - a) * y  on float32 arrays, the resampler's float64 taps).  HIP's __fmul_rn / __dadd_rn are
// plain operators and contract like any other; the pragma keeps the `contract` flag off these instructions (the library
// is built with -ffp-contract=fast-honor-pragmas).
__device__ __forceinline__ float fmul_exact(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float fadd_exact(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ double dmul_exact(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double dadd_exact(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ double dsub_exact(double a, double b) {
#pragma clang fp contract(off)
    return a - b;
}

// max(x, 0.2 x) == (x >= 0 ? x : 0.2 x) for every finite x and both zeros; two vector instructions instead of three (compare,
// multiply, select) -- vector instructions are matrix-pipe time on this chip
__device__ __forceinline__ float lrelu02(float x) { return __builtin_fmaxf(x, 0.2f * x); }

// v_exp_f32 / v_rcp_f32 based, branch-free (1-2 ulp each; absolute error ~1e-7, which is what the
// 1e-4 dgrad tolerance needs -- gate values feed products, so absolute error is the one that matters)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float sigmoidf_acc(float x) { return fast_rcp(1.0f + __expf(-x)); }

// (1 - e) / (1 + e), e = exp(-2|x|), sign restored: seven instructions.  The five-instruction form 1 - 2 / (1 + exp(2x)) was
// tried in round 2: the frequency LSTM got 0.4 % SLOWER with it (same-call A/B, profiles/r02_ab_tanh.txt), so the cell
// update is not bound by its vector-ALU instruction count; kept as it was (and the results stay those of round 1).
__device__ __forceinline__ float tanhf_acc(float x) {
    const float e = __expf(-2.0f * fabsf(x));
    return copysignf((1.0f - e) * fast_rcp(1.0f + e), x);
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }

// CUs of the CURRENT device, looked up per launch (a process may drive differently sized or CU-masked devices; a function-local
// static would freeze the first device's count -- ADVICE r3).  The attribute query is a table lookup in the runtime; the small
// per-device cache only saves the call.
#include <atomic>
inline int sdfa_cu_count() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 64) {
        const int c = cache[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    int n = 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (dev >= 0 && dev < 64) cache[dev].store(n, std::memory_order_relaxed);
    return n;
}


// PCA expansion of the dgrad head, both bases in one kernel (speech_anime/modules/output_module.py:94-116 PcaInversion;
// interleave [6 scale | 3 rotat] per triangle: speech_anime/model/model.py:246-257 data_to_anime_feat).
//
//   out[n][tri*9 + c]     = sum_k coefS[k][n] * BS[k][tri*6 + c] + meanS[tri*6 + c]        c = 0..5   (K = 85 -> 96)
//   out[n][tri*9 + 6 + c] = sum_k coefR[k][n] * BR[k][tri*3 + c] + meanR[tri*3 + c]        c = 0..2   (K = 180 -> 192)
//
// The generic GEMM runs the two bases as two launches whose epilogues each write 6 (or 3) of every 9 floats of a row:
// every 128-byte line of the 7.3 GB output is then written twice, partially, by different kernels -- a
// read-modify-write at the memory side, 7.9 ms per 20,352 frames for 2.7 ms of matrix work.  Here one workgroup owns
// 128 frames x 32 triangles and computes BOTH parts (per wave: 32 frames; 6 scale tiles + 3 rotat tiles of 32x32), so
// a wave owns complete output rows: it transposes them through LDS and writes 1,152 contiguous bytes per frame with
// 16-byte stores.
// Operands come straight from global memory as K4 quads (register-direct, no LDS): the coefficient quads of the wave's
// 32 frames and the basis quads of its columns, requested two k-blocks ahead in three rotating register sets.  The four
// waves of a workgroup read the same basis slab (L1 hits); frame blocks of one triangle block are dispatched together,
// so the slab (147 KB) stays in L2.
#include "common.h"
#include "kernels.h"

#define WAVE_LDS_FENCE() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }

namespace {

template <int NT>   // NT column tiles of 32: 6 for the scale basis (group 6), 3 for the rotat basis (group 3)
__device__ __forceinline__ void pca_part(const float4 *__restrict__ coef, int64_t ldc, const float4 *__restrict__ basis, int64_t ldb,
                                         int nkb, int h, f32x16 (&acc)[1][NT]) {
    // lane: A row = frame (coef already offset to this lane's frame), B column = basis already offset to tile 0's column
    // requests run TWO k-blocks ahead in three rotating register sets: the rotat part has only 12 MFMAs (768 cycles) per
    // k-block, less than an L2 round trip
    float4 a0[1], b0[NT], a1[1], b1[NT], a2[1], b2[NT];
#define PCA_LOAD(kb, A, B)                                                    \
    {                                                                         \
        const int64_t kq = 2 * (kb) + h;                                      \
        A[0] = coef[kq * ldc];                                                \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) B[t] = basis[kq * ldb + 32 * t]; \
    }
    PCA_LOAD(0, a0, b0)
    PCA_LOAD(1, a1, b1)
#pragma unroll 1
    for (int kb = 0; kb < nkb; kb += 3) {      // nkb is a multiple of 3 (12 or 24); late requests wrap to k-block 0 and are dropped
        PCA_LOAD(kb + 2 < nkb ? kb + 2 : 0, a2, b2)
        __builtin_amdgcn_sched_barrier(0);
        mfma_block<1, NT>(acc, a0, b0);
        PCA_LOAD(kb + 3 < nkb ? kb + 3 : 0, a0, b0)
        __builtin_amdgcn_sched_barrier(0);
        mfma_block<1, NT>(acc, a1, b1);
        PCA_LOAD(kb + 4 < nkb ? kb + 4 : 0, a1, b1)
        __builtin_amdgcn_sched_barrier(0);
        mfma_block<1, NT>(acc, a2, b2);
    }
#undef PCA_LOAD
}

constexpr int PCA_ROW = 292;   // floats per staged row: 288 used; 4 * 292 mod 32 = 16 keeps the two lane halves on different LDS banks

// accumulator rows 8g + 4h + e of pass g -> LDS rows 4h + e, triangle-interleaved columns; + mean
template <int NT, int GROUP, int OFF>
__device__ __forceinline__ void pca_stage(float *__restrict__ sRow, const f32x16 (&acc)[1][NT], const float *__restrict__ mean,
                                          int64_t col0, int64_t cols, int g, int h, int l31) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ql = 32 * t + l31;                       // column inside this workgroup's slice of the basis
        const int64_t q = col0 + ql;
        const float m = q < cols ? mean[q] : 0.f;
        const int o = (ql / GROUP) * 9 + OFF + ql % GROUP; // position inside the 288-float row segment
#pragma unroll
        for (int e = 0; e < 4; ++e) sRow[(4 * h + e) * PCA_ROW + o] = acc[0][t][4 * g + e] + m;
    }
}

// (A form with the basis staged through LDS per k-block stage -- twelve barriers per workgroup -- was 14 % slower and is gone:
// profiles/r02_ab_final.txt)
__global__ __launch_bounds__(256, 2) void pca_dgrad_kernel(PcaArgs a) {
    __shared__ float sOut[4][8 * PCA_ROW];                // per wave: 8 output rows x 288 floats, staged for 16-byte stores
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int64_t nfb = a.Nc / 128;                       // frame blocks: fastest-varying, so a basis slab is reused from L2
    const int64_t fb = blockIdx.x % nfb, tb = blockIdx.x / nfb;
    const int64_t frame0 = fb * 128 + wave * 32;
    if (frame0 >= a.N) return;

    const float4 *__restrict__ coef = reinterpret_cast<const float4 *>(a.coef) + frame0 + l31;   // K4 [288/4][Nc]: scale rows 0..95, rotat 96..287
    f32x16 accs[1][6], accr[1][3];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[0][t][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) accr[0][t][r] = 0.f;
    pca_part<6>(coef, a.Nc, reinterpret_cast<const float4 *>(a.basis_s) + tb * 192 + l31, a.ld_s, 12, h, accs);
    pca_part<3>(coef + 24 * a.Nc, a.Nc, reinterpret_cast<const float4 *>(a.basis_r) + tb * 96 + l31, a.ld_r, 24, h, accr);

    // epilogue: four passes of 8 frames; the wave transposes its (8 x 288) block through LDS and writes whole rows with
    // 16-byte stores (1,152 contiguous bytes per frame) instead of 4-byte stores strided 6-of-9 / 3-of-9
    float *sRow = sOut[wave];
    const int64_t ocol0 = tb * 288, orow_valid = a.out_dim - ocol0;     // floats of this triangle block that exist (216 in the last one)
#pragma unroll
    for (int g = 0; g < 4; ++g) {            // unrolled: g indexes accumulator registers
        pca_stage<6, 6, 0>(sRow, accs, a.mean_s, tb * 192, a.cols_s, g, h, l31);
        pca_stage<3, 3, 6>(sRow, accr, a.mean_r, tb * 96, a.cols_r, g, h, l31);
        WAVE_LDS_FENCE()
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int idx = i * 64 + lane, r = idx / 72, c4 = idx % 72;     // row-local mapping r = 4h' + e  <->  frame 8g + 4h' + e
            const int64_t n = frame0 + 8 * g + r;
            if (n < a.N && 4 * c4 < orow_valid) {
                const float4 v = *reinterpret_cast<const float4 *>(sRow + r * PCA_ROW + 4 * c4);
                const int64_t o = n * a.out_dim + ocol0 + 4 * c4;
                *reinterpret_cast<float4 *>(a.out + o) = v;
                for (int x = 0; x < a.n_extra; ++x) *reinterpret_cast<float4 *>(a.out_extra[x] + o) = v;   // peers' gathered buffers
            }
        }
        WAVE_LDS_FENCE()
    }
}

// ------------------------------------------------------------------------------------------------
// Third form (round 2, after DESIGN.md section 4.2): ONE workgroup per CU, one wave per SIMD, and the basis slab of a
// triangle block RESIDENT in LDS.  The expansion is a product of a tall matrix (frames x 265 coefficients) with a wide one
// whose column block for 32 triangles is only 147 KiB: a persistent workgroup loads that block ONCE per work unit
// (triangle block x 4-16 frame blocks, taken from a queue), after which its four waves run independently -- no barrier, no
// staging -- through the unit's tiles: B operands from LDS one k-block ahead (1 read per 4 MFMAs, in the MFMAs' shadow),
// the coefficient quads of a tile requested a whole part ahead (the rotat part's 24 during the scale part, the next tile's
// scale part's 12 during the rotat part), so nothing in the K loops waits for memory.  The register-direct form above has
// every wave fetch the slab itself through L1 for every tile and, alone on a CU, spends 174 k cycles per tile for 37 k of
// MFMAs (two k-blocks of lookahead against an L2 round trip per k-block).  The epilogue transposes two frames at a time
// (16 passes; the slab leaves 9 KiB of LDS).  Same k order per accumulator: bit-identical.
// ------------------------------------------------------------------------------------------------
constexpr int PR_KS = 11, PR_KR = 23;            // k-blocks that hold real coefficients (85 -> 88, 180 -> 184 of the padded 96 / 192)
constexpr int PR_RS = 2 * PR_KS, PR_RR = 2 * PR_KR;   // their k-quad rows

// Split-bf16 form (round 5, SDFA_PREC_BF16X3; BF = true): the same kernel with both contractions on v_mfma_f32_32x32x16_bf16, operands as
// hi + lo bf16, three products per k-step, smallest first (lo*hi, hi*lo, hi*hi) -- 108 + 108 MFMAs of 32 cycles per tile instead of
// 264 + 276 of 64.  The basis comes pre-split from the host (api.cpp: pack_pca_bf16) as OCTETS of eight consecutive k per column, the
// operand form of the instruction: per triangle block [plane hi | lo][scale rows r = 2 ks + h (12) x 192 | rotat rows (24) x 96]; the
// slab in LDS keeps the rows that hold real coefficients (11 + 23 per plane, 138 KiB) and ONE shared row of zeros for the all-padding
// k-groups (k = 88..95 and 184..191).  The coefficient quads are split by the lane that loads them (the lane's frame is its A row).
// Accumulators, means, transposition and stores are the fp32 kernel's: only the products are rounded (16 significand bits).
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
constexpr int PB_SLAB = 2 * (12 * 192 + 24 * 96);      // octets per triangle block in global memory (both planes, padded rows included)

__device__ __forceinline__ void pca_split(const float4 &x0, const float4 &x1, pbf16x8 &hi, pbf16x8 &lo) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hb = (__bf16)x[e];
        hi[e] = hb;
        lo[e] = (__bf16)(x[e] - (float)hb);
    }
}

template <bool BF>
__global__ __launch_bounds__(256, 1) void pca_dgrad_res_kernel(PcaArgs a, int *queue, int fbu) {      // fbu: frame blocks (of 128 frames) per work unit
    // Only the k-quad rows that hold real coefficients are kept: 85 scale coefficients = rows 0..21 (k-blocks 0..10; k-block 11
    // of the padded K = 96 is all zeros and is skipped -- adding exact zeros changes nothing), 180 rotat coefficients = rows
    // 0..45 (k-blocks 0..22 of 24).  That leaves room for TWO row pairs per wave, so a pass's LDS writes do not wait for the
    // previous pass's reads.
    extern __shared__ float4 sRes[];                       // [PR_RS][192] scale basis | [PR_RR][96] rotat basis | 4 x 2 x 2 x PCA_ROW floats | queue slot
    float4 *sBs = sRes, *sBr = sRes + PR_RS * 192;
    // BF: [11][192] scale hi | [11][192] scale lo | [23][96] rotat hi | [23][96] rotat lo | [192] zeros  (octets of 16 bytes, like float4)
    pbf16x8 *sBsH = reinterpret_cast<pbf16x8 *>(sRes), *sBsL = sBsH + 11 * 192, *sBrH = sBsL + 11 * 192, *sBrL = sBrH + 23 * 96, *sZero = sBrL + 23 * 96;
    float *sOutAll = BF ? reinterpret_cast<float *>(sZero + 192) : reinterpret_cast<float *>(sRes + PR_RS * 192 + PR_RR * 96);
    int *sUnit = reinterpret_cast<int *>(sOutAll + 4 * 4 * PCA_ROW);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    float *sRow2 = sOutAll + wave * 4 * PCA_ROW;          // two row pairs, alternating by pass
    const int64_t nfb = a.Nc / 128;
    const int nfu = (int)((nfb + fbu - 1) / fbu);          // units per triangle block
    const int ntb = (int)((a.cols_r + 95) / 96);
    const int n_units = ntb * nfu;
    const float4 *__restrict__ coef = reinterpret_cast<const float4 *>(a.coef);   // K4 [72][Nc]: k-quads 0..23 scale, 24..71 rotat

    // per-lane constants of the epilogue: position of this lane's column of tile t inside the 288-float row segment, and the
    // (row, float4) this lane moves in each of the three read-back / store rounds of a pass (2 rows x 72 float4 = 144)
    int opos[9];
#pragma unroll
    for (int t = 0; t < 6; ++t) { const int ql = 32 * t + l31; opos[t] = (ql / 6) * 9 + ql % 6; }
#pragma unroll
    for (int t = 0; t < 3; ++t) { const int ql = 32 * t + l31; opos[6 + t] = (ql / 3) * 9 + 6 + ql % 3; }

    // read-back / store round i of a pass: lane moves float4 c4 of row r (2 rows x 72 float4 = 144 = 2.25 x 64 lanes)
    unsigned rd_off[3], st_off[3];
    int st_r[3];
    bool st_ok[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int idx = i * 64 + lane, r = idx / 72, c4 = idx % 72;
        st_r[i] = r;
        rd_off[i] = (unsigned)((r * PCA_ROW + 4 * c4) * 4);
        st_off[i] = (unsigned)((4 * r * a.out_dim + 4 * c4) * 4);
        st_ok[i] = idx < 144;
    }

    if (BF && tid < 192) {
        pbf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
        sZero[tid] = z;                                    // (visible to everyone after the first unit's barriers)
    }
    for (;;) {
        if (tid == 0) *sUnit = atomicAdd(queue, 1);
        __syncthreads();                                   // (also: every wave is done with the previous unit's slab)
        const int u = __builtin_amdgcn_readfirstlane(*sUnit);
        if (u >= n_units) break;                           // queue empty: every workgroup gets here
        const int tb = u / nfu, fu = u % nfu;
        // the slab: 22 x 192 + 46 x 96 float4 = 135 KiB, HBM/L2 -> LDS by LDS-DMA in 1 KiB pieces (dense rows: a request costs
        // ~40-100 cycles here; through registers, with a division per element, the load took 50 k cycles per unit): scale rows
        // are three pieces, rotat rows one and a half (lanes 0..31 of the second)
        if constexpr (BF) {
            // 11 scale rows x 3 pieces + 23 rotat rows x 1.5 pieces per plane, 1 KiB per piece
            const pbf16x8 *src = reinterpret_cast<const pbf16x8 *>(a.basis_b) + (int64_t)tb * PB_SLAB + lane;
            for (int c = wave; c < 2 * 11 * 3; c += 4) {
                const int pl = c / 33, cc = c - 33 * pl, row = cc / 3, part = cc - 3 * row;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + pl * (PB_SLAB / 2) + row * 192 + part * 64),
                                                 (void __attribute__((address_space(3))) *)((pl ? sBsL : sBsH) + row * 192 + part * 64), 16, 0, 0);
            }
            for (int c = wave; c < 2 * 23 * 2; c += 4) {
                const int pl = c / 46, cc = c - 46 * pl, row = cc >> 1, part = cc & 1;
                if (part == 0 || lane < 32)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + pl * (PB_SLAB / 2) + 12 * 192 + row * 96 + part * 64),
                                                     (void __attribute__((address_space(3))) *)((pl ? sBrL : sBrH) + row * 96 + part * 64), 16, 0, 0);
            }
        } else {
            const float4 *bs = reinterpret_cast<const float4 *>(a.basis_s) + (int64_t)tb * 192 + lane;
            const float4 *br = reinterpret_cast<const float4 *>(a.basis_r) + (int64_t)tb * 96 + lane;
            for (int c = wave; c < PR_RS * 3; c += 4) {
                const int row = c / 3, part = c - 3 * row;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(bs + (int64_t)row * a.ld_s + part * 64),
                                                 (void __attribute__((address_space(3))) *)(sBs + row * 192 + part * 64), 16, 0, 0);
            }
            for (int c = wave; c < PR_RR * 2; c += 4) {
                const int row = c >> 1, part = c & 1;
                if (part == 0 || lane < 32)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(br + (int64_t)row * a.ld_r + part * 64),
                                                     (void __attribute__((address_space(3))) *)(sBr + row * 96 + part * 64), 16, 0, 0);
            }
        }
        float mean[9];
#pragma unroll
        for (int t = 0; t < 6; ++t) { const int64_t q = (int64_t)tb * 192 + 32 * t + l31; mean[t] = q < a.cols_s ? a.mean_s[q] : 0.f; }
#pragma unroll
        for (int t = 0; t < 3; ++t) { const int64_t q = (int64_t)tb * 96 + 32 * t + l31; mean[6 + t] = q < a.cols_r ? a.mean_r[q] : 0.f; }
        __builtin_amdgcn_s_waitcnt(0x0070);                // this wave's DMA pieces have landed (explicit vmcnt(0): see gemm_fat_kernel) ...
        __syncthreads();                                   // ... and everyone's: slab complete

        const int64_t fb0 = (int64_t)fu * fbu, fb1 = fb0 + fbu < nfb ? fb0 + fbu : nfb;
        const int64_t ocol0 = (int64_t)tb * 288, orow_valid = a.out_dim - ocol0;      // floats of this triangle block that exist (216 in the last one)
#pragma unroll
        for (int i = 0; i < 3; ++i) st_ok[i] = (i * 64 + lane) < 144 && 4 * ((i * 64 + lane) % 72) < orow_valid;
        // coefficient quads of this lane's frame: sa = scale part (12 k-blocks), ra = rotat part (24)
        float4 sa[12], ra[12], rb[12];     // 12 quads each: at most two of the three sets are live at a time
#define PR_LOAD(A, fb, kq0) { const float4 *cp = coef + (int64_t)(kq0) * a.Nc + (fb) * 128 + wave * 32 + l31; _Pragma("unroll") for (int kb = 0; kb < 12; ++kb) A[kb] = cp[(int64_t)(2 * kb + h) * a.Nc]; }
#define PR_LOAD_S(fb) PR_LOAD(sa, fb, 0)
        // BF: the octet of k-step ks and lane half h is quads 4 ks + 2 h, 4 ks + 2 h + 1 -> slots 2 ks, 2 ks + 1 of sa (scale), ra (rotat
        // k-steps 0..5) and rb (6..11); requested at the same points of a tile as the fp32 form's (two of the three sets alive at a time)
#define PB_LOAD(A, fb, kq0)                                                                                     \
        {                                                                                                       \
            const float4 *cp = coef + (int64_t)(kq0) * a.Nc + (fb) * 128 + wave * 32 + l31;                     \
            _Pragma("unroll") for (int q = 0; q < 12; ++q) A[q] = cp[(int64_t)(4 * (q >> 1) + 2 * h + (q & 1)) * a.Nc]; \
        }
        if constexpr (BF) PB_LOAD(sa, fb0, 0) else PR_LOAD_S(fb0)
        for (int64_t fb = fb0; fb < fb1; ++fb) {
            const int64_t frame0 = fb * 128 + wave * 32;
            f32x16 accs[1][6], accr[1][3];
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[0][t][r] = 0.f;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) accr[0][t][r] = 0.f;
            if constexpr (BF) {
                PB_LOAD(ra, fb, 24)                        // k-steps 0..5 of the rotat part: requested before the scale part
#define PMFMA(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_bf16((A_), (B_), (C_), 0, 0, 0)
                // row r = 2 ks + h of a plane; the last k-step's upper half (k = 88..95 / 184..191) is padding on both sides: a shared
                // row of zeros for B, and a zero A operand there (the coefficient buffer's padding rows are the caller's)
#define PB_ROWP(BASE, ROWLEN, ks, LAST) ((ks) == (LAST) ? (h ? sZero : (BASE) + 2 * (ks) * (ROWLEN)) : (BASE) + (2 * (ks) + h) * (ROWLEN))
#define PB_B(NT, BASEH, BASEL, ROWLEN, LAST, ks, BH, BL)                                                        \
                {                                                                                               \
                    const pbf16x8 *ph = PB_ROWP(BASEH, ROWLEN, ks, LAST), *pl_ = PB_ROWP(BASEL, ROWLEN, ks, LAST); \
                    _Pragma("unroll") for (int t = 0; t < NT; ++t) { BH[t] = ph[32 * t + l31]; BL[t] = pl_[32 * t + l31]; } \
                }
#define PB_MM(NT, ACC, Q0, Q1, LASTKS, BH, BL)                                                                  \
                {                                                                                               \
                    pbf16x8 ahi, alo;                                                                           \
                    pca_split(Q0, Q1, ahi, alo);                                                                \
                    if ((LASTKS) && h) { _Pragma("unroll") for (int e = 0; e < 8; ++e) { ahi[e] = (__bf16)0.f; alo[e] = (__bf16)0.f; } } \
                    _Pragma("unroll") for (int t = 0; t < NT; ++t) ACC[0][t] = PMFMA(alo, BH[t], ACC[0][t]);    \
                    _Pragma("unroll") for (int t = 0; t < NT; ++t) ACC[0][t] = PMFMA(ahi, BL[t], ACC[0][t]);    \
                    _Pragma("unroll") for (int t = 0; t < NT; ++t) ACC[0][t] = PMFMA(ahi, BH[t], ACC[0][t]);    \
                }
                // scale part: 6 k-steps of 16 x 6 tiles x 3 products, as two passes of three tiles (B operands one k-step ahead in two
                // alternating sets: with all six tiles' hi and lo planes in two sets the kernel needs more than 512 registers)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    pbf16x8 bh0[3], bl0[3], bh1[3], bl1[3];
                    f32x16 (&acch)[1][3] = *reinterpret_cast<f32x16 (*)[1][3]>(&accs[0][3 * half]);
                    PB_B(3, sBsH + 96 * half, sBsL + 96 * half, 192, 5, 0, bh0, bl0)
#pragma unroll
                    for (int ks = 0; ks < 6; ks += 2) {
                        PB_B(3, sBsH + 96 * half, sBsL + 96 * half, 192, 5, ks + 1, bh1, bl1)
                        __builtin_amdgcn_sched_barrier(0);
                        PB_MM(3, acch, sa[2 * ks], sa[2 * ks + 1], false, bh0, bl0)
                        __builtin_amdgcn_sched_barrier(0);
                        if (ks + 2 < 6) { PB_B(3, sBsH + 96 * half, sBsL + 96 * half, 192, 5, ks + 2, bh0, bl0) }
                        __builtin_amdgcn_sched_barrier(0);
                        PB_MM(3, acch, sa[2 * ks + 2], sa[2 * ks + 3], ks + 1 == 5, bh1, bl1)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                PB_LOAD(rb, fb, 48)                        // k-steps 6..11: land during the first half of the rotat part
                {   // rotat part: 12 k-steps x 3 tiles x 3 products
                    pbf16x8 bh0[3], bl0[3], bh1[3], bl1[3];
                    PB_B(3, sBrH, sBrL, 96, 11, 0, bh0, bl0)
#pragma unroll
                    for (int ks = 0; ks < 12; ks += 2) {
                        if (ks == 6 && fb + 1 < fb1) { PB_LOAD(sa, fb + 1, 0) }      // the next tile's scale part (sa is dead by now)
                        PB_B(3, sBrH, sBrL, 96, 11, ks + 1, bh1, bl1)
                        __builtin_amdgcn_sched_barrier(0);
                        if (ks < 6) { PB_MM(3, accr, ra[2 * ks], ra[2 * ks + 1], false, bh0, bl0) }
                        else { PB_MM(3, accr, rb[2 * (ks - 6)], rb[2 * (ks - 6) + 1], false, bh0, bl0) }
                        __builtin_amdgcn_sched_barrier(0);
                        if (ks + 2 < 12) { PB_B(3, sBrH, sBrL, 96, 11, ks + 2, bh0, bl0) }
                        __builtin_amdgcn_sched_barrier(0);
                        if (ks + 1 < 6) { PB_MM(3, accr, ra[2 * ks + 2], ra[2 * ks + 3], false, bh1, bl1) }
                        else { PB_MM(3, accr, rb[2 * (ks - 5)], rb[2 * (ks - 5) + 1], ks + 1 == 11, bh1, bl1) }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#undef PB_MM
#undef PB_B
#undef PB_ROWP
#undef PMFMA
            } else {
            PR_LOAD(ra, fb, 24)                            // first half of the rotat part: lands during the scale part (18 k cycles of MFMAs)
            {   // scale part: 11 k-blocks x 24 MFMAs, B operands one k-block ahead in two alternating sets
                float4 b0[6], b1[6];
#define PR_BS(kb, B) _Pragma("unroll") for (int t = 0; t < 6; ++t) B[t] = sBs[(2 * (kb) + h) * 192 + 32 * t + l31];
                PR_BS(0, b0)
#pragma unroll
                for (int kb = 0; kb < PR_KS; kb += 2) {
                    if (kb + 1 < PR_KS) { PR_BS(kb + 1, b1) }
                    __builtin_amdgcn_sched_barrier(0);
                    { const float4 a1[1] = {sa[kb]}; mfma_block<1, 6>(accs, a1, b0); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (kb + 2 < PR_KS) { PR_BS(kb + 2, b0) }
                    __builtin_amdgcn_sched_barrier(0);
                    if (kb + 1 < PR_KS) { const float4 a1[1] = {sa[kb + 1]}; mfma_block<1, 6>(accs, a1, b1); }
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef PR_BS
            }
            PR_LOAD(rb, fb, 48)                            // second half of the rotat part: lands during its first half (9 k cycles)
            {   // rotat part: 23 k-blocks x 12 MFMAs
                float4 b0[3], b1[3];
#define PR_BR(kb, B) _Pragma("unroll") for (int t = 0; t < 3; ++t) B[t] = sBr[(2 * (kb) + h) * 96 + 32 * t + l31];
                PR_BR(0, b0)
#pragma unroll
                for (int kb = 0; kb < PR_KR; kb += 2) {
                    if (kb == 12 && fb + 1 < fb1) { PR_LOAD_S(fb + 1) }      // the next tile's scale part: lands during the second half
                    if (kb + 1 < PR_KR) { PR_BR(kb + 1, b1) }
                    __builtin_amdgcn_sched_barrier(0);
                    { const float4 a1[1] = {kb < 12 ? ra[kb] : rb[kb - 12]}; mfma_block<1, 3>(accr, a1, b0); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (kb + 2 < PR_KR) { PR_BR(kb + 2, b0) }
                    __builtin_amdgcn_sched_barrier(0);
                    if (kb + 1 < PR_KR) { const float4 a1[1] = {kb + 1 < 12 ? ra[kb + 1] : rb[kb + 1 - 12]}; mfma_block<1, 3>(accr, a1, b1); }
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef PR_BR
            }
            }
#ifdef SDFA_PR_NOEPI   /* timing experiment only: how long do the K loops alone take? */
            if (accs[0][0][0] + accr[0][0][0] != 123.456f) continue;
#endif
            if (frame0 >= a.N) continue;                   // a wave of padding frames (its MFMAs ran on zeros; nothing to store)
            // epilogue: 16 passes of two frames (accumulator rows 8g + 4h + e): transpose through LDS, whole-row 16-byte stores,
            // software-pipelined -- pass p's LDS writes (row pair p & 1) are issued while pass p-1's read-back (other pair) is in
            // flight; ONE fence per pass covers both, then pass p-1 is stored and pass p read back.
#define PR_WRITE(p)                                                                                              \
            {                                                                                                    \
                float *sRow = sRow2 + ((p) & 1) * 2 * PCA_ROW;                                                   \
                _Pragma("unroll") for (int t = 0; t < 6; ++t) sRow[h * PCA_ROW + opos[t]] = accs[0][t][p] + mean[t];          \
                _Pragma("unroll") for (int t = 0; t < 3; ++t) sRow[h * PCA_ROW + opos[6 + t]] = accr[0][t][p] + mean[6 + t];  \
            }
#define PR_READ1(p, i, RV) RV = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sRow2 + ((p) & 1) * 2 * PCA_ROW) + rd_off[i]);
#define PR_READ(p) PR_READ1(p, 0, rv0) PR_READ1(p, 1, rv1) PR_READ1(p, 2, rv2)
#define PR_STORE1(i, RV)                                                                                         \
                if (st_ok[i] && nrow + 4 * st_r[i] < a.N) {                                                      \
                    *reinterpret_cast<float4 *>(reinterpret_cast<char *>(a.out) + sbase + st_off[i]) = RV;       \
                    for (int x = 0; x < a.n_extra; ++x) *reinterpret_cast<float4 *>(reinterpret_cast<char *>(a.out_extra[x]) + sbase + st_off[i]) = RV;   /* peers' gathered buffers */ \
                }
#define PR_STORE(p)                                                                                              \
            {   /* accumulator register p = 4g + e holds frame rows 8g + 4h' + e */                             \
                const int64_t nrow = frame0 + 8 * ((p) >> 2) + ((p) & 3);                                        \
                const int64_t sbase = (nrow * a.out_dim + ocol0) * 4;     /* uniform part of the address: scalar registers */ \
                if (plain) {          /* whole tile, one destination: no predicates, no destination loop */     \
                    *reinterpret_cast<float4 *>(reinterpret_cast<char *>(a.out) + sbase + st_off[0]) = rv0;      \
                    *reinterpret_cast<float4 *>(reinterpret_cast<char *>(a.out) + sbase + st_off[1]) = rv1;      \
                    if (lane < 16) *reinterpret_cast<float4 *>(reinterpret_cast<char *>(a.out) + sbase + st_off[2]) = rv2; \
                } else {                                                                                         \
                    PR_STORE1(0, rv0) PR_STORE1(1, rv1) PR_STORE1(2, rv2)                                        \
                }                                                                                                \
            }
            float4 rv0, rv1, rv2;      // (named: an array indexed inside the destination loop ends up in scratch memory)
            const bool plain = frame0 + 32 <= a.N && orow_valid >= 288 && a.n_extra == 0;      // uniform
#ifdef SDFA_PCA_DIRECT   /* experiment: no LDS transposition -- every lane stores its own 4-byte values (144 dword stores
                            per lane and tile instead of 48 x 16 bytes); correct results for whole tiles with one destination */
            if (plain) {
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    float *orow = a.out + (frame0 + 8 * (p >> 2) + (p & 3) + 4 * h) * a.out_dim + ocol0;
#pragma unroll
                    for (int t = 0; t < 6; ++t) orow[opos[t]] = accs[0][t][p] + mean[t];
#pragma unroll
                    for (int t = 0; t < 3; ++t) orow[opos[6 + t]] = accr[0][t][p] + mean[6 + t];
                }
                continue;
            }
#endif
            PR_WRITE(0)
            WAVE_LDS_FENCE()
            PR_READ(0)
#pragma unroll
            for (int p = 1; p < 16; ++p) {
                PR_WRITE(p)
                WAVE_LDS_FENCE()       // pass p's writes done AND pass p-1's read-back arrived
                PR_STORE(p - 1)
                PR_READ(p)
            }
            WAVE_LDS_FENCE()           // (also orders the last read-back before the next tile's first writes)
            PR_STORE(15)
#undef PR_WRITE
#undef PR_READ
#undef PR_READ1
#undef PR_STORE
#undef PR_STORE1
        }
#undef PR_LOAD_S
#undef PR_LOAD
#undef PB_LOAD
    }
}

}  // namespace


hipError_t sdfa_launch_pca_dgrad_res(const PcaArgs &a, int *queue, hipStream_t s) {
    const int64_t ntb = (a.cols_r + 95) / 96;
    if (a.Nc % 128 || a.ld_s < ntb * 192 || a.ld_r < ntb * 96 || a.cols_s != 2 * a.cols_r || !queue) return hipErrorInvalidValue;
    const bool bf = a.terms == 3 && a.basis_b != nullptr;          // split-bf16 form (SDFA_PREC_BF16X3); every other mode: exact fp32
    const size_t lds = (bf ? (size_t)(2 * 11 * 192 + 2 * 23 * 96 + 192) * 16 : (size_t)(PR_RS * 192 + PR_RR * 96) * sizeof(float4)) + 4 * 4 * PCA_ROW * sizeof(float) + 16;
    hipError_t e = hipFuncSetAttribute(bf ? reinterpret_cast<const void *>(pca_dgrad_res_kernel<true>) : reinterpret_cast<const void *>(pca_dgrad_res_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(queue, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    // work unit = one triangle block x `fbu` frame blocks: the slab is loaded once per unit, so large units are cheaper, but the
    // queue should still hold several units per CU for balance
    const int64_t nfb = a.Nc / 128;
    const int fbu = ntb * ((nfb + 15) / 16) >= 4 * cus ? 16 : (ntb * ((nfb + 7) / 8) >= 4 * cus ? 8 : 4);
    const int64_t units = ntb * ((nfb + fbu - 1) / fbu);
    cus = std::max(1, cus - a.reserve_cus);      // CUs left to kernels of other streams (sdfa_model_set_reserved_cus)
    if (bf) hipLaunchKernelGGL(pca_dgrad_res_kernel<true>, dim3((unsigned)(units < cus ? units : cus)), dim3(256), lds, s, a, queue, fbu);
    else hipLaunchKernelGGL(pca_dgrad_res_kernel<false>, dim3((unsigned)(units < cus ? units : cus)), dim3(256), lds, s, a, queue, fbu);
    return hipGetLastError();
}

hipError_t sdfa_launch_pca_dgrad(const PcaArgs &a, hipStream_t s) {
    // 32 triangles per workgroup: 192 scale columns + 96 rotat columns; the padded leading dimensions must cover whole blocks
    const int64_t ntb = (a.cols_r + 95) / 96;
    if (a.Nc % 128 || a.ld_s < ntb * 192 || a.ld_r < ntb * 96 || a.cols_s != 2 * a.cols_r) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pca_dgrad_kernel, dim3((unsigned)(ntb * (a.Nc / 128))), dim3(256), 0, s, a);
    return hipGetLastError();
}

// Internal launcher interface between api.cpp (host orchestration) and the .hip kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { ACT_NONE = 0, ACT_LRELU = 1, ACT_TANH = 2 };

// Frequency-LSTM hidden states are stored tile-major: float4[column block of 128][HF_SLAB_ROWS][128], of which the
// first 8192/4 = 2048 rows of a slab are used.  The pad rows keep the slab stride off a power of two: every workgroup
// of the projection GEMM streams its own slab front to back at about the same pace, and with a 4 MiB stride all of
// them would sit on the same HBM channels at the same time.
#ifndef SDFA_HF_PAD
#define SDFA_HF_PAD 0
#endif
constexpr int HF_SLAB_ROWS = 2048 + SDFA_HF_PAD;
enum { OUT_K4 = 0, OUT_ROW = 1 };
constexpr int SDFA_MAX_DESTS = 8;   // output destinations of the regressor epilogues: the local buffer + up to 7 peers

struct GemmArgs {
    const float *P;      // K4 [K/4][ldp]
    const float *Q;      // K4 [K/4][ldq] (per segment)
    float *D;
    const float *bias;   // over p (K4 features) or over q (OUT_ROW)
    const float *cond_w; // K4-ish [Ppad/4][8][4]: column of the FC weight that the one-hot speaker selects
    const int64_t *cond_idx;  // [Qreal] speaker ids
    int64_t ldp, ldq, ldd;
    int64_t Ppad, Qpad;  // multiples of 128: tiles launched
    int64_t Pstore;      // rows actually stored (multiple of 4)
    int64_t Qreal;       // columns that exist (OUT_ROW store mask / bias_q / cond bounds)
    int K;               // total contraction length, multiple of 32
    int seg_k;           // contraction length per segment (= K when one segment)
    int64_t seg_col;     // Q column offset between segments
    int act, out_mode, bias_on_q;
    int terms;                // 0 = fp32 MFMA; 1 = operands rounded to bf16; 3 = split-bf16, three bf16 MFMAs per product
    int q_tile_major;         // Q is stored tile-major: float4[column block of 128][q_slab_rows >= K/4][128] (the freq-LSTM hidden states)
    int q_slab_rows;
    const int64_t *q_limit;   // device scalar: tiles whose first column is >= *q_limit exit at once (null = no limit)
    int col_group, col_stride, col_off;   // OUT_ROW only: column q is stored at (q / group) * stride + off + q % group (group 0 = identity)
    float *D_extra[SDFA_MAX_DESTS - 1];   // OUT_ROW only: further destinations with D's layout (see PcaArgs::out_extra)
    int n_extra;
    int reserve_cus;          // persistent kernel (gemm_fat_kernel): launch (CUs - reserve_cus) workgroups
};
hipError_t sdfa_launch_gemm(const GemmArgs &a, hipStream_t s);

// ---- PCA expansion of the dgrad head, both bases fused (pca.hip) ---------------------------------
struct PcaArgs {
    const float *coef;               // K4 [288/4][Nc]: rows 0..95 scale coefficients (85 real), 96..287 rotat (180 real)
    const float *basis_s, *basis_r;  // K4 [96/4][ld_s], [192/4][ld_r]
    const float *mean_s, *mean_r;    // [cols_s], [cols_r]
    float *out;                      // [N][out_dim] row-major, triangle-interleaved [s0..s5 r0 r1 r2]
    int64_t N, Nc, out_dim, ld_s, ld_r, cols_s, cols_r;
    // one-shot direct all-gather: the same rows are ALSO stored to n_extra more base pointers --
    // this rank's slot in each peer GPU's gathered buffer, mapped over xGMI (or other buffers on this device)
    float *out_extra[SDFA_MAX_DESTS - 1];
    int n_extra;
    int reserve_cus;                 // pca_dgrad_res_kernel: launch (CUs - reserve_cus) workgroups
    // split-bf16 form of pca_dgrad_res_kernel (terms == 3): both bases as bf16 octets, per triangle block
    // [plane hi | lo][scale rows 2 ks + h (12) x 192 columns | rotat rows (24) x 96] (api.cpp: pack_pca_bf16); null / other terms = fp32
    const void *basis_b;
    int terms;
};
hipError_t sdfa_launch_pca_dgrad(const PcaArgs &a, hipStream_t s);
hipError_t sdfa_launch_pca_dgrad_res(const PcaArgs &a, int *queue, hipStream_t s);   // basis slab resident in LDS, persistent; queue: one int of workspace

// ---- front end -----------------------------------------------------------------------------
struct FrontendConsts {      // device pointers, built once per sample rate
    const float *hamm;       // [win]
    const float2 *twiddle;   // [win]  exp(-2*pi*i*m/win)
    const int *mel_bin0;     // [128]     first FFT bin of each mel band (the bins of a band are consecutive)
    const float *mel_w8;     // [8][128]  its weights, tap-major, zero padded to 8 taps
    int win, hop, sliding, nbins_used, nnz;
};
hipError_t sdfa_launch_frontend(const FrontendConsts &c, const float *pcm, const int64_t *clip_off,
                                const int64_t *clip_len, const int32_t *frame_clip, const int64_t *frame_start,
                                int64_t n_frames, float *audio_feat, hipStream_t s);
// "spectral gather" form: mel columns of the DISTINCT STFT columns (col_src / n_distinct from the share map with
// t in 1..63), then one gather + delta + store pass per frame
hipError_t sdfa_launch_mel_columns(const FrontendConsts &c, const float *pcm, const int64_t *clip_off, const int64_t *clip_len,
                                   const int32_t *frame_clip, const int64_t *frame_start, const int32_t *col_src,
                                   const int64_t *n_distinct, float *mel_table, hipStream_t s);
hipError_t sdfa_launch_mel_stream(const FrontendConsts &c, const float *pcm, const int64_t *clip_off, const int64_t *clip_len,
                                  const int32_t *frame_clip, const int64_t *frame_start, const int32_t *prev, const int32_t *shift,
                                  int64_t n_frames, int block, int slots, int producer_consumer, int spin_max, int *status, float *audio_feat, hipStream_t s);
hipError_t sdfa_launch_gather_features(const float *mel_table, const int32_t *col_to_u, int64_t n_frames, int64_t Nc, int frame_major,
                                       float *audio_feat, hipStream_t s);   // frame_major: col_to_u is [n][t] (ShareArgs::frame_major)

// ---- conv stack ----------------------------------------------------------------------------
struct ConvArgs {
    const float *audio_feat;  // [N][64][128][3]
    int64_t N, Nc, Mc;        // real frames, padded frames per chunk, columns = 64*Nc
    // conv1: packed A operand [5 k-steps][2 halves][32 co] (k = df*3 + c, k = 9 -> 0), epilogue constants
    const float *w1, *b1, *s1, *t1;   // bias, BN scale, BN shift per co
    float *P1;                // K4 [2048/4][Mc]   rows f1*32 + co
    // conv2 / conv3: K4 weights [K/4][64][4]
    const float *w2, *b2, *s2, *t2;   // K = 96  (k = df*32 + ci)
    const float *w3, *b3, *s3, *t3;   // K = 64
    float *X3;                // K4 [2048/4][Mc]   rows f*64 + ch
    // column sharing: column m reads audio_feat row col_src[m] (-1 = zeros) instead of (m % Nc)*64 + m / Nc, and
    // workgroups whose first column is >= *col_limit exit at once.  Both null = every column of every frame.
    const int32_t *col_src;
    const int64_t *col_limit;
    // mixed-precision modes (sdfa_launch_conv123 only): bf16 planes [hi | mid | lo] of the three layers' weights in the K order of
    // conv123_bf16_kernel (api.cpp: pack_conv_bf16); terms 0 = fp32 MFMA, 1 / 3 / 6 as in FreqLstmArgs
    const void *wb;
    int terms;
};
hipError_t sdfa_launch_conv1(const ConvArgs &a, hipStream_t s);
hipError_t sdfa_launch_conv23(const ConvArgs &a, hipStream_t s);
hipError_t sdfa_launch_conv123(const ConvArgs &a, hipStream_t s);   // conv1_pool + conv23 fused (P1 is not written)

// ---- LSTM recurrences ----------------------------------------------------------------------
struct FreqLstmArgs {
    const float *X3;     // K4 [2048/4][Mc]
    const float *W;      // per direction: K4 [(64+128)/4][512][4], gate rows packed per wave
    const float *bias;   // per direction: [512] packed (b_ih + b_hh)
    float *HF;           // K4 [8192/4][Mc]  rows f*256 + dir*128 + j
    int64_t Mc;
    const int64_t *col_limit;   // see ConvArgs
    const void *Wb;      // mixed-precision modes: per direction bf16x8 [hi | lo][24 octets][512 rows] (lstm.hip)
    int terms;           // 0 = fp32 MFMA; 1 = bf16 MFMA; 3 = split-bf16 (hi/lo) MFMA
    int *tile_counter;   // one int of workspace: the persistent form's work queue head (zeroed by the launcher)
    int shape;           // fp32 kernel / launch form (all bit-identical): freq_lstm_v3_kernel 9 = persistent (tile queue), one workgroup per CU by
                         // design (default), 8 = one hardware-dispatched workgroup per tile; freq_lstm_v2_kernel 5 = persistent, two per CU,
                         // 3 = hardware-dispatched, two per CU (the fallback that shares a CU); anything else = 9
    int reserve_cus;     // persistent forms: launch (CUs - reserve_cus) workgroups (sdfa_model_set_reserved_cus)
};
hipError_t sdfa_launch_freq_lstm(const FreqLstmArgs &a, hipStream_t s);

struct TimeLstmArgs {
    const float *GX;     // K4 [2048/4][Mc]  rows dir*1024 + packed gate row  (input projections)
    const float *W;      // per direction: K4 [256/4][1024][4]
    float *H;            // K4 [512/4][Mc]   rows dir*256 + j
    int64_t Nc, Mc;
    const void *Wb;      // mixed-precision modes: per direction bf16x8 [hi | lo][32 octets][1024 rows] (lstm.hip)
    int terms;           // 0 = fp32 MFMA; 1 = bf16; 3 = split-bf16
    unsigned *flags;     // small-batch form (time_lstm_split_kernel): [0] this launch's time-out word, [4 ..] one flag per workgroup; null = never split
    int64_t flag_words;  // words available at `flags`
    const float *W16;    // time_lstm_split16_kernel: per direction float4 [16 K16][4 g][1024 rows] (api.cpp pack_rec_16x16x4); null = not packed
    unsigned *status;    // word 0 of the workspace's status block: counts the waits of the small-batch form that expired (null = not counted)
    int reserve_cus;     // the small-batch form is used while its grid fits (CUs - reserve_cus)
    const int32_t *col_map;   // column sharing, layer 0: GX holds the DISTINCT columns; column (t, n) reads GX column col_map[t * Nc + n] (null = its own)
};
hipError_t sdfa_launch_time_lstm(const TimeLstmArgs &a, hipStream_t s);

// ---- attention scores / softmax / context ---------------------------------------------------
struct AttnArgs {
    const float *KP;     // K4 [128/4][Mc]  key projections
    const float *QP;     // K4 [128/4][Nc]  query projection
    const float *H;      // K4 [512/4][Mc]  values (BiLSTM output)
    const float *v;      // [128]
    const float *b;      // [128]
    float *Zk4;          // K4 [512/4][Nc]  context (feeds the output MLPs)
    float *z_out;        // [N][512] row-major (may be null)
    float *align_out;    // [N][64]  row-major (may be null)
    int64_t N, Nc, Mc;
    const float *S;      // [Mc][8] partial scores written by attn_key_score_kernel (then KP / QP / v / b are not read); null = computed here from KP
};
hipError_t sdfa_launch_attn(const AttnArgs &a, hipStream_t s);

// bf16 attention modes: key projection + scores in one streaming pass over H (attn.hip: attn_key_score_kernel)
struct AttnKeyArgs {
    const float *Wk;     // K4 [512/4][128]  proj_key weights, as the GEMM takes them
    const float *H;      // K4 [512/4][Mc]   BiLSTM output
    const float *QP;     // K4 [128/4][Nc]   query projection
    const float *v, *b;  // [128]
    float *S;            // out [Mc][8]  partial score (outputs 16w .. 16w+15) of column m = t * Nc + n per wave w
    int64_t Nc, Mc;
    int terms;           // 0 = exact fp32 (v_mfma_f32_16x16x4_f32), 1 = bf16 operands, 3 = split-bf16
    int reserve_cus;
    int ts_shift;        // set by the launcher: a work unit is 16 frames x (64 >> ts_shift) time steps
    // terms == 0 and fuse_tail: the whole attention layer in this launch (attn_fused_f32_kernel: running softmax + context while the tile
    // is in LDS); S is not written, sdfa_launch_attn is not needed.  Outputs as AttnArgs.
    int fuse_tail;
    float *Zk4, *z_out, *align_out;
    int64_t N;
};
// true when sdfa_launch_attn_key_score would take the one-launch form for this size (enough 16-frame units to give every CU two)
bool sdfa_attn_fuses_tail(int64_t Nc, int reserve_cus);
hipError_t sdfa_launch_attn_key_score(const AttnKeyArgs &a, hipStream_t s);

// ---- layout helpers --------------------------------------------------------------------------
// row-major [n][F] (n < N) -> K4 [F/4][ld]  (zero for n >= N), and back
hipError_t sdfa_launch_rows_to_k4(const float *src, int64_t N, int F, float *dst, int64_t ld, hipStream_t s);
hipError_t sdfa_launch_rows_seg_to_k4(const float *src, int64_t src_ld, int64_t N, int c0, int nf, float *dst, int64_t ld,
                                      int f0, int nf_pad, hipStream_t s);
hipError_t sdfa_launch_k4_to_rows(const float *src, int64_t ld, int64_t N, int F, int f0, int nf, float *dst,
                                  int64_t dst_ld, hipStream_t s);
// debug taps (tests): K4 [F/4][Mc] with m = t*Nc+n  ->  reference layouts
hipError_t sdfa_launch_tap(const float *src, int what, int64_t N, int64_t Nc, float *dst, hipStream_t s);

// ---- column sharing (share.hip) -----------------------------------------------------------------
struct ShareArgs {
    const int32_t *frame_clip;   // [N] clip id of each frame of the chunk
    const int64_t *frame_start;  // [N] window start sample inside its clip
    int hop;
    int t_lo, t_hi;              // window columns t_lo..t_hi are shareable between hop-aligned frames of a clip
    int frame_major;             // index order of owner / flag / uid / col_to_u AND numbering order of the distinct columns: 0 = time-step-major
                                 // (i = t * Nc + n), 1 = frame-major (i = n * 64 + t)
    int64_t N, Nc, Mc;
    int32_t *prev, *shift;       // [Nc]
    int32_t *owner, *flag, *uid; // [Mc]
    int32_t *tile_sum;           // [Mc/1024 + 1] scan scratch
    int32_t *col_src;            // [Mc] out: audio_feat row of each distinct column (-1 padding)
    int32_t *col_to_u;           // [Mc] out: distinct-column index of every column, in the index order above
    int64_t *counts;             // [2]  out: number of distinct columns, and that rounded up to 256
};
hipError_t sdfa_launch_share_map(const ShareArgs &a, hipStream_t s);
hipError_t sdfa_launch_share_prev(const ShareArgs &a, hipStream_t s);     // prev / shift only (the spectral-stream front end reads the chains from them)
hipError_t sdfa_launch_expand_cols(const float *Zu, const int32_t *col_to_u, float *Z, int nquads, int64_t Mc, hipStream_t s);

// ---- dgrad -> mesh (mesh.hip) -----------------------------------------------------------------------
struct MeshArgs {
    const float *dgrad;        // [rows][n_src_tris][9]
    int64_t n_frames;          // output frames (= rows without a seek plan, = queries with one)
    int n_tris, n_verts, n_free, free_pad;
    int n_src_tris;            // triangles per dgrad row (= n_tris without triangle correspondences)
    const int64_t *seek_src;   // [n_frames][2] dgrad rows blended into each output frame (null: frame f is row f)
    const float *seek_w;       // [n_frames][2] their float32 weights
    const float *reg_xt;       // [n_free][3] reg * template position of each free vertex
    const int *inc_ptr;        // [n_free + 1] CSR over free vertices
    const int *inc_tri;        // [nnz] SOURCE triangle of each incidence
    const float *inc_coef;     // [nnz][3] the A entries of that (triangle, vertex) pair
    const int *vert_col;       // [n_verts] free-vertex row or -1
    const float *tmpl;         // [n_verts][3] template positions
    float *rhs;                // K4 [free_pad/4][ld]   ld = columns = 3 * frames padded to 128
    const float *sol;          // K4 [free_pad/4][ld]   Inv * rhs
    float *verts;              // out [n_frames][n_verts][3]
    int64_t ld;
};
hipError_t sdfa_launch_mesh_rhs(const MeshArgs &a, hipStream_t s);
hipError_t sdfa_launch_mesh_scatter(const MeshArgs &a, hipStream_t s);
// saber.stream.seek on the device (mesh.hip): plan = (src rows, float32 weights) per query; rows = the blended rows
hipError_t sdfa_launch_seek_plan(const int32_t *tslist, const int64_t *frame_off, const int64_t *query_off, int n_clips, double fps,
                                 int64_t n_queries, int64_t *src, float *w, hipStream_t s);
hipError_t sdfa_launch_seek_rows(const float *rows, int64_t width, const int64_t *src, const float *w, int64_t nq, float *out, hipStream_t s);
// test-time ensembling: out = (a + b) / 2, element-wise, float32 roundings of numpy's `sum += x; sum / 2.0` (vector path for 16-byte aligned pointers)
hipError_t sdfa_launch_ensemble_mean(const float *a, const float *b, int64_t n, float *out, hipStream_t s);

// ---- audio ingest: kaiser_best resampling (resample.hip) --------------------------------------------------------
struct ResampleArgs {
    const float *x;            // [n_in] input samples
    int64_t n_in;
    float *y;                  // [n_out]
    int64_t n_res, n_out;      // int(n_in * ratio) filtered samples, then zeros up to n_out = ceil(n_in * ratio)
    const double *win, *delta; // [nwin] half filter (scaled by the ratio when downsampling) and its forward differences
    const double *treg;        // [n_res] time register of every output sample
    int64_t nwin, step;        // table length, table entries per input sample = int(min(1, ratio) * num_table)
    double scale;              // min(1, ratio)
    int num_table;             // table entries per zero crossing (512)
};
hipError_t sdfa_launch_resample(const ResampleArgs &a, hipStream_t s);

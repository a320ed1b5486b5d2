/*
 * sdfa_hip.h -- C ABI of libsdfa_hip.so: the MI355X (gfx950) implementation of the
 * audio -> dgrad/offsets inference path of chaiyujin/sdfa-2019.
 *
 * Plain pointers and sizes only; no torch / C++ types cross this boundary.
 * Every d_* pointer is DEVICE memory (hipMalloc'ed or owned by any allocator, e.g.
 * PyTorch's caching allocator); every h_* pointer is HOST memory.  `stream` is a
 * hipStream_t passed as void* (NULL = the default stream).  All device work is enqueued
 * on `stream` and the calls never synchronise (the constructors sdfa_model_finalize / sdfa_mesh_create* and the debug /
 * profiling readers do).  A finalised model or mesh is read-only: the forward calls may be used concurrently from several
 * threads on different streams with different workspaces.  Functions return 0 (or a non-negative
 * count) on success and a negative SDFA_E* code on failure; sdfa_last_error() returns
 * the message of the calling thread's last failure.
 *
 * Each entry point names the reference interface (paths relative to the reference
 * repository root) whose arithmetic it replaces.  The Python host that keeps the
 * reference's own names on top of this ABI lives in sdfa-2019_amd/speech_anime/.
 */
#ifndef SDFA_HIP_H
#define SDFA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDFA_ABI_VERSION 5   /* bumped whenever an export is added or a signature changes: the Python binding refuses a library of another version with
                                "stale library, run make" instead of a bare AttributeError.  5: + sdfa_debug_frontend_status (added in round 5 without a bump);
                                4: workspace status block (sdfa_workspace_init / _status*, replaces sdfa_debug_time_lstm_timeout), unaligned sdfa_ensemble_mean (round 4);
                                3: + sdfa_ensemble_mean, sdfa_model_set_reserved_cus, sdfa_debug_time_lstm_timeout (round 3); 2: + seek, resample, mesh correspondences,
                                multi-destination regress, expand_coef, autotune (round 2); all earlier entry points unchanged */

#define SDFA_OK            0
#define SDFA_EINVAL       -1   /* bad argument (shape, size, null pointer, unsupported rate) */
#define SDFA_ESHORTCLIP   -2   /* a window would need zero padding on both sides: the reference asserts
                                  (speech_anime/datasets/sliding_window.py:363) */
#define SDFA_EHIP         -3   /* a HIP runtime call failed */
#define SDFA_ESTATE       -4   /* model not finalised / tensor missing */
#define SDFA_ENOSPACE     -5   /* output capacity or workspace too small */

#define SDFA_HEAD_DGRAD    0   /* speech_anime/config/model/dgrad.py   : 9976 triangles x (6 scale | 3 rotat) */
#define SDFA_HEAD_OFFSETS  1   /* speech_anime/config/model/offsets.py : 15069 vertex offsets */

#define SDFA_FEAT_FRAMES   64  /* audio.feature.sliding_window_frames  (config/model/dgrad.py:11) */
#define SDFA_FEAT_MELS    128  /* audio.mel.n_mels                     (config/data/voca-dgrad.py:13) */
#define SDFA_FEAT_CH        3  /* mel, delta, delta-delta              (datasets/get_features.py:196-215) */
#define SDFA_Z_DIM        512
#define SDFA_DGRAD_DIM  89784  /* 9976 * 9  (speech_anime/model/model.py:246-257) */
#define SDFA_OFFSETS_DIM 15069
#define SDFA_COEF_SCALE    85
#define SDFA_COEF_ROTAT   180
#define SDFA_COEF_OFFSETS  59

int         sdfa_abi_version(void);
const char *sdfa_last_error(void);

/* ------------------------------------------------------------------------------------------
 * (a1) Frame enumeration and timestamps -- host integer / float32-rounded arithmetic.
 * Replaces the `while` loop of DatasetSlidingWindow.fetch_audio_features
 * (speech_anime/datasets/sliding_window.py:339-354) and the unit converters
 * frame_to_sample / sample_to_ms (speech_anime/datasets/speech_anime.py:135-145).
 * Bit-exact, including numpy-float32 roundings and Python round-half-to-even.
 *
 *   sample_rate, fps              hparams.audio.sample_rate, hparams.anime.fps (60)
 *   win, hop                      int(0.064*sr), int(0.008*sr)
 *   ts_delta_ms                   hparams.anime.feature.ts_delta (100)
 *   h_starts[cap], h_tslist[cap]  window start sample (may be negative) and timestamp (ms)
 * Returns the number of frames F (call with cap = 0 and null outputs to size), or
 * SDFA_ESHORTCLIP where the reference's assert would fire.  Clips of more than 2^29 - 1 samples return SDFA_EINVAL: the
 * front-end kernels index a clip's samples in 32-bit arithmetic (a clip handed to sdfa_mel_frontend* directly must respect
 * the same limit; a clip of length 0 contributes all-zero windows).
 * ---------------------------------------------------------------------------------------- */
int64_t sdfa_frame_index(int64_t n_samples, int sample_rate, int fps, int win, int hop,
                         int ts_delta_ms, int64_t *h_starts, int32_t *h_tslist, int64_t cap);

/* ------------------------------------------------------------------------------------------
 * (a2-a4) Spectral-gather front end: zero-padded window cut, per-window pre-emphasis,
 * Hamming STFT (center=False), power, 128-band Slaney mel, dB, normalise+clamp, delta and
 * delta-delta (Savitzky-Golay width 9), (T,F,C) interleave.
 * Replaces saber.audio.features.mel_spectrogram (saber/data/audio/features/spectrogram.py:66-104,
 * 236-249, 298-308; misc.py:8-19,94-123), windowed_features' inference branch
 * (speech_anime/datasets/get_features.py:8-69,90,159-167,196-223) and the per-frame body of
 * fetch_audio_features (sliding_window.py:356-371, :462).
 *
 *   d_pcm              concatenated clips, float32 in [-1,1]
 *   d_clip_off/len     [n_clips] offset and length (samples) of each clip inside d_pcm
 *   d_frame_clip       [n_frames] clip index of each frame
 *   d_frame_start      [n_frames] window start relative to its clip (from sdfa_frame_index)
 *   sample_rate        8000 (win 512, hop 64) or 16000 (win 1024, hop 128)
 *   d_audio_feat       out [n_frames][64][128][3] float32
 * ---------------------------------------------------------------------------------------- */
int sdfa_mel_frontend(const float *d_pcm, const int64_t *d_clip_off, const int64_t *d_clip_len,
                      int32_t n_clips, const int32_t *d_frame_clip, const int64_t *d_frame_start,
                      int64_t n_frames, int sample_rate, float *d_audio_feat, void *stream);

/* The same features through the "spectral gather" form: windows of one clip whose starts differ by whole hops contain
 * the same STFT columns (only window column 0 is special: misc.py:17), so each DISTINCT column is transformed once
 * (26 instead of 64 per frame at 60 fps / hop 8 ms; plus a few re-transformed where a chain of such frames is cut into
 * segments) and every frame is assembled from its 64 mel rows (delta filters + (T,F,C) store).  Since round 5 the mel rows
 * live in an LDS ring of the workgroup that walks the chain (the "spectral stream": no mel table in HBM; 100.6 KB of HBM
 * traffic per frame against 99.4 KB algorithmic); the two-kernel form through a table stays behind the option
 * "frontend_two_kernel" and gives the same bits.  Same arguments as sdfa_mel_frontend plus scratch memory of
 * sdfa_frontend_workspace_bytes(n_frames) bytes (sized for the two-kernel form).  Results agree with sdfa_mel_frontend to
 * float rounding (a column is transformed alone, as a half-size complex FFT, instead of sharing a complex FFT with its
 * neighbour): a column's features depend on its samples only -- not on the batch it is in, not on the form or segment
 * geometry that computed it -- and are bit-identical in every frame that contains it.  Any frame table is accepted (the
 * sharing is read from the table, nothing is assumed about the frame rate). */
int64_t sdfa_frontend_workspace_bytes(int64_t max_frames);
int sdfa_mel_frontend_gather(const float *d_pcm, const int64_t *d_clip_off, const int64_t *d_clip_len,
                             int32_t n_clips, const int32_t *d_frame_clip, const int64_t *d_frame_start,
                             int64_t n_frames, int sample_rate, float *d_audio_feat, void *d_workspace,
                             int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Model weights.  Tensors are given by the reference's state_dict names with the `_model.`
 * prefix stripped, AFTER weight-norm has been folded
 * (saber/trainer/manager/device_mover.py:26-31) -- i.e. `...weight`, never weight_g/weight_v --
 * as contiguous float32 host arrays.  BatchNorm is given unfolded (weight, bias, running_mean,
 * running_var); the library folds it (eps 1e-3, config/model/dgrad.py:1).
 * sdfa_model_finalize packs everything into the device layouts the kernels stream from and
 * fails with SDFA_ESTATE if a tensor of the chosen head is missing or has the wrong size.
 * ---------------------------------------------------------------------------------------- */
typedef struct sdfa_model sdfa_model;

sdfa_model *sdfa_model_create(int head);
void        sdfa_model_destroy(sdfa_model *m);
int         sdfa_model_set_tensor(sdfa_model *m, const char *name, const float *h_data, int64_t numel);
int         sdfa_model_finalize(sdfa_model *m, void *stream);
int         sdfa_model_head(const sdfa_model *m);
int64_t     sdfa_model_out_dim(const sdfa_model *m);          /* 89784 or 15069 */
int64_t     sdfa_model_coef_dim(const sdfa_model *m);         /* 265 (85 scale | 180 rotat) or 59 */

/* Mixed-precision modes (BASELINE.json configs[3], "bf16 attention ... mixed-precision tolerance sweep").  The
 * reference computes in fp32 only (torch CPU, model.py:28-45); fp32 is the default and the headline.  A mode selects
 * the matrix instruction of the dense contractions; the mel front end, the conv stack, softmax / context, every
 * accumulation, LSTM cell state, bias and activation stay fp32 in all modes.  May be changed between calls.
 *   SDFA_PREC_FP32            v_mfma_f32_32x32x2_f32 everywhere (exact fp32 products)
 *   SDFA_PREC_BF16_ATTENTION  attention stage (key / query projections, query conv) on bf16 MFMA, rest fp32
 *   SDFA_PREC_BF16X3          conv stack, frequency LSTM, BiLSTM recurrences, every GEMM and (round 5) the fused dgrad PCA expansion on
 *                             split-bf16: operands as hi + lo bf16 (16 significand bits), three
 *                             v_mfma_f32_32x32x16_bf16 per product.  Measured worst case over other weight dynamics (other seeds,
 *                             all three LSTMs x1.3, BatchNorm scales of both signs), the reference's 10 s fixture and the full-size
 *                             batch: see profiles/r05_precision_modes.json -- inside the 1e-4 budget on every case; the hot
 *                             recurrences (x1.3) are the worst case, about 6x the fixture case
 *   SDFA_PREC_BF16            the same kernels on plain bf16 operands (8 bits) -- outside the 1e-4 budget,
 *                             kept as the far end of the sweep
 *   SDFA_PREC_BF16X3_ATTENTION  configs[3]'s wording with a passing point: the attention stage alone (key / query projections,
 *                             query conv) on bf16 MFMA with split-bf16 operands, everything else fp32 -- "bf16 attention with
 *                             MFMA" inside the 1e-4 budget (plain bf16 operands there, SDFA_PREC_BF16_ATTENTION, are not)      */
#define SDFA_PREC_FP32 0
#define SDFA_PREC_BF16_ATTENTION 1
#define SDFA_PREC_BF16X3 2
#define SDFA_PREC_BF16 3
#define SDFA_PREC_BF16X3_ATTENTION 4
#define SDFA_PREC_BF16X6 5           /* the kernels of SDFA_PREC_BF16X3 with operands as THREE bf16 terms (24 significand bits = fp32's) and six
                                        partial products per product: fp32-equivalent products at 16 / 6 = 2.7x the fp32 MFMA rate */
int         sdfa_model_set_precision(sdfa_model *m, int mode);
int         sdfa_model_precision(const sdfa_model *m);

/* Optional, once per model and device: times the bit-identical kernels / launch forms of the fp32 frequency-LSTM recurrence (the
 * one-workgroup-per-CU kernel and the two-per-CU kernel, hardware-dispatched or persistent) on `n_frames` frames of zeros in the caller's workspace and keeps
 * the fastest for all later forward calls.  Blocks until the measurement is done (about 10 launches of the kernel).  Returns
 * the form chosen (> 0) or a negative error code.  Results of the forward calls do not depend on it.  (New: the reference has
 * no counterpart; its PyTorch kernels are picked by cuDNN's own heuristics.) */
int         sdfa_model_autotune(sdfa_model *m, int64_t n_frames, void *d_workspace, int64_t workspace_bytes, void *stream);

/* The persistent / one-workgroup-per-CU kernels of the large stages (frequency LSTM, projection GEMMs, PCA expansion) launch
 * one workgroup per CU.  With k > 0 they launch (CUs - k): kernels of OTHER streams -- RCCL's all-gather of the previous
 * chunk's rows (sdfa_amd/dist.py), a copy kernel -- then find free CUs while a forward call runs.  0 (default) = all CUs.
 * Results do not depend on it.  (New: no counterpart in the reference.) */
int         sdfa_model_set_reserved_cus(sdfa_model *m, int k);

/* Bytes of scratch device memory the forward calls need for up to `max_frames` frames per call. */
int64_t sdfa_workspace_bytes(const sdfa_model *m, int64_t max_frames);

/* Status block = the first SDFA_WS_STATUS_BYTES bytes of a workspace: counters the kernels only ever INCREMENT.
 *   word SDFA_WS_TIME_LSTM_REPAIRS  waits of the small-batch time-LSTM kernels that expired.  For a single clip the BiLSTM recurrence
 *       runs on pairs of cooperating workgroups that hand h to each other every step and wait for each other with a wall-clock
 *       bound (20 ms; "time_lstm_timeout_us").  A wait can only expire when other work keeps a partner off the device for that
 *       long; the launch then marks itself and a repair pass that needs no co-residency recomputes the layer ON THE DEVICE, in
 *       stream order, before anything reads it -- a forward call never returns rows of a timed-out launch (the reference never
 *       returns partial results either: speech_anime/model/model.py:428-489).  The counter only says that it happened (a
 *       performance event: about the bound + 2 ms per occurrence).
 * sdfa_workspace_init zeroes the block (once, after allocating; the forward calls never do).  sdfa_workspace_status_async enqueues a
 * copy of the SDFA_WS_STATUS_WORDS words to h_status (pinned host memory: valid once the stream has passed it; never
 * synchronises); sdfa_workspace_status reads one word and synchronises the stream.  (New: no counterpart in the reference.) */
#define SDFA_WS_STATUS_BYTES 256
#define SDFA_WS_STATUS_WORDS 4
#define SDFA_WS_TIME_LSTM_REPAIRS 0
int     sdfa_workspace_init(void *d_workspace, int64_t workspace_bytes, void *stream);
int     sdfa_workspace_status_async(const void *d_workspace, uint32_t *h_status, void *stream);
int64_t sdfa_workspace_status(const void *d_workspace, int word, void *stream);

/* ------------------------------------------------------------------------------------------
 * (a6-a10) Audio encoder: permute, conv2d x3 (+LeakyReLU 0.2 then eval BatchNorm) with two
 * max-pools along frequency, frequency BiLSTM + 8192->256 projection, 2-layer time BiLSTM,
 * Bahdanau attention with the centre-3-frame query.
 * Replaces modules.Configurable("audio_encoder").forward (speech_anime/modules/configurable.py:17-25)
 * = layers.forward over config/model/dgrad.py:60-70 (speech_anime/layers/__init__.py:63-148;
 * saber/nn/layers/conv2d.py:6-28,64-97; layers/freq_lstm.py:36-55; layers/rnn.py:20-21;
 * layers/attentions.py:39-75,92-124).
 *
 *   d_audio_feat  [n_frames][64][128][3] float32, contiguous
 *   d_z           out [n_frames][512]   attention context (z_audio)
 *   d_align       out [n_frames][64]    attention weights (align_dict["audio_encoder10"]); may be NULL
 * Alignment: d_audio_feat, d_z and d_workspace 16 bytes, d_align 4 bytes (SDFA_EINVAL otherwise).
 * ---------------------------------------------------------------------------------------- */
int sdfa_encoder_forward(const sdfa_model *m, const float *d_audio_feat, int64_t n_frames,
                         float *d_z, float *d_align, void *d_workspace, int64_t workspace_bytes,
                         void *stream);

/* Same result, fewer evaluations ("legal redundancy"): the per-column stages (conv stack,
 * frequency LSTM, its projection) are run once per DISTINCT column.  Windows of one clip whose starts differ by
 * a whole number of hops contain bit-identical interior columns (t in [6,58]); the library finds them on the device
 * from the frame table of sdfa_frame_index / sdfa_mel_frontend, on every call.
 *   d_frame_clip  [n_frames] int32 clip id        d_frame_start [n_frames] int64 window start in its clip
 *   hop           STFT hop in samples (int(0.008*sr))
 * audio_feat must be what sdfa_mel_frontend produced for exactly this frame table. */
int sdfa_encoder_forward_shared(const sdfa_model *m, const float *d_audio_feat, int64_t n_frames,
                                const int32_t *d_frame_clip, const int64_t *d_frame_start, int hop,
                                float *d_z, float *d_align, void *d_workspace, int64_t workspace_bytes,
                                void *stream);

/* ------------------------------------------------------------------------------------------
 * (a5, a11, a12) Speaker one-hot condition, output MLPs, PCA expansion and the per-triangle
 * [6 scale | 3 rotat] interleave.
 * Replaces SpeakerEmbedding.forward (speech_anime/modules/speaker.py:21-27), OutputModule.forward and
 * PcaInversion.forward (speech_anime/modules/output_module.py:51-89,94-116) and
 * data_to_anime_feat (speech_anime/model/model.py:246-257).
 *
 *   d_z           [n_frames][512]
 *   d_speaker_id  [n_frames] int64, each in [0, 8)
 *   d_coef        out [n_frames][coef_dim]  PCA coefficients (scale then rotat); may be NULL
 *   d_out         out [n_frames][out_dim]   dgrad (9976 x [s0..s5 r0 r1 r2]) or offsets; may be NULL
 * Alignment: d_z and d_workspace 16 bytes; d_out 16 bytes for the dgrad head (rows are written with 16-byte stores),
 * 4 bytes for the offsets head; d_coef 4 bytes.  A misaligned pointer returns SDFA_EINVAL.
 * ---------------------------------------------------------------------------------------- */
int sdfa_regress_forward(const sdfa_model *m, const float *d_z, const int64_t *d_speaker_id,
                         int64_t n_frames, float *d_coef, float *d_out, void *d_workspace,
                         int64_t workspace_bytes, void *stream);

/* The same, with the output rows stored to SEVERAL destinations (at most 8): h_d_outs is a HOST array of n_outs device
 * pointers, each a [n_frames][out_dim] buffer with d_out's alignment rules.  This is the one-shot direct all-gather of the
 * per-frame output: on a fully connected xGMI node a rank passes its own buffer plus its slot
 * in every peer's gathered buffer (peer memory mapped with hipIpcOpenMemHandle) and the regressor's epilogue writes each row
 * once per destination -- no staging copy, no collective, one shard per link.  All destinations receive identical bits. */
int sdfa_regress_forward_multi(const sdfa_model *m, const float *d_z, const int64_t *d_speaker_id, int64_t n_frames,
                               float *d_coef, float *const *h_d_outs, int n_outs, void *d_workspace,
                               int64_t workspace_bytes, void *stream);

/* The PCA expansion alone: coefficients (as sdfa_regress_forward's d_coef returns them) -> output rows, bit-identical to the
 * rows the regressor itself writes for the same coefficients.  PcaInversion.forward
 * (speech_anime/modules/output_module.py:94-116) + data_to_anime_feat (speech_anime/model/model.py:246-257).
 * Multi-GPU use (sdfa_amd/dist.py ExpandGatherer): ranks all-gather the 1 KB-per-frame coefficients instead of the
 * 359 KB-per-frame dgrad rows and every rank expands the peers' frames locally.
 *   d_coef  [n_frames][coef_dim]   d_out  [n_frames][out_dim]   alignment as for sdfa_regress_forward */
int sdfa_expand_coef(const sdfa_model *m, const float *d_coef, int64_t n_frames, float *d_out, void *d_workspace,
                     int64_t workspace_bytes, void *stream);

/* Test-time ensembling (speech_anime/model/model.py:369-403, `--ensembling_ms`): the mean of the two passes' output rows,
 * d_out[i] = (d_a[i] + d_b[i]) / 2 with numpy's float32 roundings (`anime_sum += second; anime_sum / 2.0`), so that the
 * averaged track never has to be formed on the host.  n = elements; pointers 4-byte aligned (16-byte aligned operands take the
 * vector path: the offsets head's 60,276-byte rows are 16-byte aligned only every fourth row); d_out may equal d_a. */
int sdfa_ensemble_mean(const float *d_a, const float *d_b, int64_t n, float *d_out, void *stream);

/* Debug / parity taps: copy an intermediate activation of the LAST sdfa_encoder_forward call out of
 * the workspace in the reference's layout.  what: 0 = pool1 (n,32,64,64)  1 = conv3 (n,64,32,64)
 * 2 = freq-lstm (n,256,64)  3 = bilstm (n,64,512).  Used by tests only. */
/* Tuning switches for A/B runs.  THREAD-LOCAL: a switch affects the launches made by the calling thread only (so a thread that
 * flips one cannot change what concurrent callers launch; bench.py --opt and the tests set them on the thread that calls the
 * forward).  Every choice of a switch gives bit-identical results unless stated.  Unknown names return SDFA_EINVAL.
 *   name               values
 *   "freq_lstm_shape"  0 = the model's form (9 unless sdfa_model_autotune picked another); 9 = freq_lstm_v3_kernel persistent, one
 *                      workgroup per CU; 8 = the same kernel, one hardware-dispatched workgroup per tile; 5 = freq_lstm_v2_kernel
 *                      persistent, two workgroups per CU; 3 = freq_lstm_v2_kernel hardware-dispatched (the fallback that shares a CU)
 *                      (split-bf16 / six-product modes: freq_lstm_bf16p_v3_kernel persistent, one workgroup per CU; 8 = the same kernel, one
 *                      workgroup per tile; 3 = freq_lstm_bf16_kernel<3> / freq_lstm_bf16x6_kernel, two workgroups per CU -- the same bits)
 *   "gemm_variant"     0 = per-shape default (gemm_fat_kernel where 256 x 256 tiles fill the chip twice, else the 128 x 128 LDS-tiled
 *                      kernel; 256 x 256 gemm_big_kernel for the 8192-deep projection at mid sizes); 9 = never the persistent fat
 *                      kernel (two-workgroups-per-CU fallback); 8 = fat wherever it fits; 5 = 256 x 256 tile wherever it fits;
 *                      6 / 2 = the 64 x 64-tile form of the LDS-tiled kernel wherever it applies / never (default: launches of at
 *                      most two 128 x 128 tiles per CU with K <= 512, and any launch of less than half a tile per CU);
 *                      10 / 11 = the 64 x 128 tile of the LDS-tiled kernel never / wherever that kernel runs (default: the 8192-deep frequency
 *                      projection while it has fewer than four 128 x 128 tiles per CU -- single-clip and few-clip calls);
 *                      4 = split-bf16 x3 products (NOT exact fp32); 7 / 12 = the split-bf16 kernels' 256 x 256 tile never / wherever it
 *                      divides the problem (default: only where such tiles reach half the CUs -- else the 128 x 128 tile, same bits)
 *   "pca_lds"          0 = pca_dgrad_res_kernel (basis slab resident in LDS, persistent); 4 = pca_dgrad_kernel (register-direct, two
 *                      workgroups per CU: the fallback that shares a CU)
 *   "time_lstm_split"  0 = by size: a chunk whose 32-frame time-LSTM tiles leave most CUs idle (a single clip) splits each tile's gate rows
 *                      over 2 cooperating workgroups that exchange h every step -- 16-frame tiles on v_mfma_f32_16x16x4_f32 up to 1,024
 *                      frames (time_lstm_split16_kernel), 32-frame tiles up to 2,048 (time_lstm_split_kernel); 1 = never; 32 = 32-frame
 *                      tiles only; 16 = 16-frame tiles or an error
 *   "time_lstm_handoff" how those workgroups publish / consume h: 0 = write-through (sc1) stores + sc1 loads (default); bit 0 = plain
 *                      stores + agent-scope release; bit 1 = agent-scope acquire + plain loads (the always-valid form, slower);
 *                      bit 2 (tests only) = the second workgroup of every pair never publishes, so every wait of the first expires
 *   "time_lstm_timeout_us" bound of one such wait in microseconds (0 = the default, 20,000: ten single-clip layers; an expiry is repaired on the device)
 *   "mel_fft_radix4"   1 = the 16 kHz column FFT as four LDS-staged radix-4 passes + a radix-2 pass (rounds 2-3) instead of three register-resident
 *                      radix-8 passes (NOT bit-identical: another order of additions; both inside the 5e-5 feature tolerance)
 *   "gather_plain_order" 1 = workgroup b of the feature gather takes frame b (rounds 2-3) instead of the XCD-aware chain order (same bits)
 *   "frontend_t_major" 1 = sdfa_mel_frontend_gather numbers its distinct STFT columns time-step-major (rounds 2-3) instead of clip by clip,
 *                      hop by hop (same features, bit for bit; only the order of the mel table's rows differs)
 *   "share_gx0_off"    1 = sdfa_encoder_forward_shared expands the frequency projection to all columns before the layer-0 BiLSTM input
 *                      projection (rounds 2-3) instead of projecting the distinct columns and letting the recurrence read them through the map
 *   "pca_unfused"      1 = the dgrad PCA expansion as two generic GEMM launches with the scatter epilogue (round-1 form)
 *   "conv_unfused"     1 = conv1_pool_kernel + conv23_kernel instead of the fused conv123_kernel (what the debug taps use)
 *   "conv_fp32"        1 = the body precision modes (bf16, bf16x3, bf16x6) keep the conv stack on the fp32 kernel instead of
 *                      conv123_bf16_kernel (NOT bit-identical: that stack's operand rounding)
 *   "pca_fp32"         1 = SDFA_PREC_BF16X3 keeps the dgrad PCA expansion on the fp32 kernel (NOT bit-identical: the expansion's operand rounding)
 *   "attn_unfused"     1 = the attention stage as key-projection GEMM + attn_kernel computing the scores from the stored projections (rounds 1-5)
 *                      instead of attn_key_score_*_kernel (key projection + tanh + v-dot in one pass, nothing stored) + attn_kernel<true>;
 *                      2 = that two-kernel form also where exact fp32 would run the WHOLE layer in one launch (attn_fused_f32_kernel: running
 *                      softmax + context while the tile is in LDS; chunks of about 3,600 frames and more).  The three forms are NOT
 *                      bit-identical to each other (order of a dot product's terms / of the softmax's sums: last-bit differences)
 *   "frontend_two_kernel" 1 = sdfa_mel_frontend_gather as share map + mel_columns_kernel + gather_features_kernel through a mel table in
 *                      HBM (rounds 2-4) instead of the spectral stream (mel_stream_kernel: mel rows in an LDS ring, no table); same bits
 *   "frontend_stream_phases" 1 = the spectral stream's workgroups alternate between transforming a phase's columns and emitting its frames
 *                      (a barrier pair per phase, all 15 waves do both) instead of 12 producer waves (column transforms, no barrier) + 3
 *                      consumer waves (delta filters and stores) handing mel rows over through LDS counters; same bits
 *   "frontend_stream_spin_max" n > 0 = the producer / consumer hand-off waits give up after n polls (default 4 M: never); 1 makes them
 *                      expire at once, which exercises the repair pass (sdfa_debug_frontend_status); same bits
 *   "frontend_stream_block" / "frontend_stream_slots"  segment geometry of the spectral stream: frames per block (0 = 144, at most 256)
 *                      and workgroups per block (0 = 12); same bits for every value   */
int sdfa_debug_set_option(const char *name, int value);
int sdfa_debug_keep_intermediates(sdfa_model *m, int on);   /* un-aliased workspace: call before sizing it */
/* Number of distinct columns the LAST sdfa_encoder_forward_shared call evaluated for a chunk of n_frames frames
 * (synchronises the stream).  Tests / reporting only. */
int64_t sdfa_debug_distinct_columns(const sdfa_model *m, int64_t n_frames, const void *d_workspace, void *stream);
int sdfa_debug_tap(const sdfa_model *m, int what, int64_t n_frames, float *d_dst, const void *d_workspace,
                   void *stream);
/* Status word of the spectral-stream front end's last call on this front-end workspace (synchronises the stream): the number of bounded
 * hand-off waits between the waves of a workgroup that expired -- 0 always, unless the hand-off logic is wrong.  Every launch of that
 * form is followed, in stream order, by a repair pass that exits at once when the word is 0 and otherwise does the call again in the
 * barrier form ("frontend_stream_phases"), so the features of the call are right either way (like the time LSTM's repair pass: a
 * forward call never returns rows of a wait that timed out); the word only counts.  Only the producer / consumer form (the default) has such waits.  Tests only ("frontend_stream_spin_max" forces them). */
int sdfa_debug_frontend_status(const void *d_workspace, void *stream);
/* Per-stage device timing of the last forward calls made with profiling enabled (HIP events on the
 * caller's stream).  names: "conv1","conv23","freq_lstm","freq_proj","gx0","lstm0","gx1","lstm1",
 * "attn_proj","attn","mlp","pca".  Returns milliseconds or a negative code. */
int   sdfa_profile_enable(sdfa_model *m, int on);
int   sdfa_profile_reset(sdfa_model *m);
float sdfa_profile_ms(const sdfa_model *m, const char *stage);

/* ------------------------------------------------------------------------------------------
 * NEXT ROW: dgrad -> mesh, the deformation-transfer solve that consumes this path's
 * output.  Replaces the reference's native module deformation.set_target / deformation.get_mesh
 * (deformation/cpp/src/pybind.cpp:13-33,101-117; deform_triangle_impl.hpp:8-140,215-310;
 * rotation/utils_rotation.cpp:33-49), called per video frame from speech_anime/viewer/frame.py:102-141.
 *
 * sdfa_mesh_create = set_target(verts, faces, cnsts, reg=1e-10) without triangle correspondences (with them:
 * sdfa_mesh_create_corres below): builds the
 * per-triangle pseudo-inverse system over the free (un-constrained) vertices and factors it once (host, fp64).
 * sdfa_mesh_from_dgrad = get_mesh for n frames at once, constrained vertices pinned to their template positions
 * (what frame.py passes as vert_cnsts):
 *   d_dgrad [n_frames][n_tris*9] float32 (the rows sdfa_regress_forward writes) -> d_verts [n_frames][n_verts][3].
 * ---------------------------------------------------------------------------------------- */
typedef struct sdfa_mesh sdfa_mesh;
sdfa_mesh *sdfa_mesh_create(const float *h_verts, int64_t n_verts, const uint32_t *h_faces, int64_t n_tris,
                            const uint32_t *h_cnsts, int64_t n_cnsts, double reg, void *stream);
void       sdfa_mesh_destroy(sdfa_mesh *mesh);
int64_t    sdfa_mesh_workspace_bytes(const sdfa_mesh *mesh, int64_t n_frames);
int        sdfa_mesh_from_dgrad(const sdfa_mesh *mesh, const float *d_dgrad, int64_t n_frames, float *d_verts,
                                void *d_workspace, int64_t workspace_bytes, void *stream);
int64_t    sdfa_mesh_n_verts(const sdfa_mesh *mesh);
int64_t    sdfa_mesh_n_src_tris(const sdfa_mesh *mesh);      /* 9-vectors per dgrad row */

/* set_target WITH triangle correspondences -- retargeting to a template of another topology
 * (evaluate.sh:28-41 `--mesh_tricorres`; speech_anime/viewer/frame.py:50-80 builds corr_count / corr_faces from the
 * .tricorrs file; deform_triangle_impl.hpp:12-21,102,248-266).  Target triangle j contributes max(1, corr_count[j])
 * equations; equation k takes the transform of SOURCE triangle corr_faces[k] (one filler entry per triangle with
 * corr_count 0, whose equation is the identity).  h_corr_count NULL = no correspondences (then n_src_tris = n_tris).
 *   h_corr_count [n_tris]   h_corr_faces [sum_j max(1, corr_count[j])], each < n_src_tris
 *   n_src_tris   9-vectors per dgrad row handed to sdfa_mesh_from_dgrad* (9976 for the FLAME-topology model output) */
sdfa_mesh *sdfa_mesh_create_corres(const float *h_verts, int64_t n_verts, const uint32_t *h_faces, int64_t n_tris,
                                   const uint32_t *h_cnsts, int64_t n_cnsts, const uint32_t *h_corr_count,
                                   const uint32_t *h_corr_faces, int64_t n_corr_faces, int64_t n_src_tris, double reg,
                                   void *stream);

/* ------------------------------------------------------------------------------------------
 * NEXT ROW: frame-time resampling, saber.stream.seek
 * (saber/data/stream/stream.py:20-46), called per video frame from speech_anime/model/model.py:204-212 with
 * ts = i * 1000.0 / fps, i = 0 .. int(tslist[-1] * fps / 1000.0), right before frame_to_mesh.
 *
 * sdfa_seek_query_count: number of video frames of a clip whose last timestamp is last_timestamp_ms (host arithmetic).
 * sdfa_seek_plan: for a batch of clips, one thread per query: float64 query time, binary search in the clip's
 *   (ascending, int32 ms) timestamps, and the two float32 blend weights float32(a), float32(1 - a),
 *   a = (t[m+1] - ts) / (t[m+1] - t[m]) in float64; before the first / after the last timestamp and on the last frame the
 *   row is copied (src0 = src1, weights 1, 0).  Bit-exact.
 *     d_tslist          concatenated timestamps of all clips (sdfa_frame_index)
 *     d_clip_frame_off  [n_clips + 1] first animation frame (= row of the output matrix) of each clip
 *     d_clip_query_off  [n_clips + 1] first query (video frame) of each clip; [n_clips] = n_queries
 *     d_seek_src        out [n_queries][2] int64 global row indices    d_seek_w  out [n_queries][2] float32
 * sdfa_seek_rows: out[q] = w0 * rows[src0] + w1 * rows[src1], three separately rounded float32 operations per element
 *   (numpy's a * x + (1 - a) * y on float32 arrays) -- the rows evaluate() dumps as NNNNNN_dgrad.npy.
 * sdfa_mesh_from_dgrad_seek: seek fused into the mesh solve -- every triangle's 9-vector is blended on the fly from the
 *   two source rows (same three roundings), so the video-rate dgrad track is never materialised: d_dgrad stays the
 *   animation-rate rows sdfa_regress_forward wrote, d_verts is [n_queries][n_verts][3].  Bit-identical to
 *   sdfa_seek_rows followed by sdfa_mesh_from_dgrad.
 * ---------------------------------------------------------------------------------------- */
int64_t sdfa_seek_query_count(int32_t last_timestamp_ms, double fps);
int     sdfa_seek_plan(const int32_t *d_tslist, const int64_t *d_clip_frame_off, const int64_t *d_clip_query_off,
                       int32_t n_clips, double fps, int64_t n_queries, int64_t *d_seek_src, float *d_seek_w, void *stream);
int     sdfa_seek_rows(const float *d_rows, int64_t row_width, const int64_t *d_seek_src, const float *d_seek_w,
                       int64_t n_queries, float *d_out, void *stream);
int     sdfa_mesh_from_dgrad_seek(const sdfa_mesh *mesh, const float *d_dgrad, const int64_t *d_seek_src,
                                  const float *d_seek_w, int64_t n_queries, float *d_verts, void *d_workspace,
                                  int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * NEXT ROW: audio ingest, sample-rate conversion.  Replaces
 * librosa.resample(y, orig_sr, target_sr) [res_type "kaiser_best" -> resampy.resample] as the reference calls it at
 * speech_anime/model/eval_utils.py:76-86 (file rate -> 44.1 kHz inside saber.audio.load, saber/data/audio/io.py:9-15,
 * then 44.1 kHz -> hparams.audio.sample_rate).  Third-party arithmetic (librosa 0.8.0 / resampy 0.2.2, absent from the
 * reference tree): restated from the published algorithm, PARITY UNPINNED (oracle/resample_oracle.py).
 *   n_out = sdfa_resample_out_len(n_in, sr_orig, sr_new) = ceil(n_in * sr_new / sr_orig)   (librosa's fixed length)
 *   d_workspace: sdfa_resample_workspace_bytes(...) bytes, 8-byte aligned.
 * sdfa_resample uploads the per-sample time register from the host and SYNCHRONISES the stream once (ingest step).
 * sdfa_resample_filter copies the half filter table (64 * 512 + 1 float64) to the host: tests compare it with scipy's.
 * ---------------------------------------------------------------------------------------- */
int64_t sdfa_resample_out_len(int64_t n_in, int sr_orig, int sr_new);
int64_t sdfa_resample_workspace_bytes(int64_t n_in, int sr_orig, int sr_new);
int     sdfa_resample_filter(double *h_half_window, int64_t cap);
int     sdfa_resample(const float *d_in, int64_t n_in, int sr_orig, int sr_new, float *d_out, int64_t n_out,
                      void *d_workspace, int64_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SDFA_HIP_H */

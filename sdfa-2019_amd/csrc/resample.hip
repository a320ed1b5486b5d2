// Audio ingest next row: band-limited sample-rate conversion, the arithmetic of
// librosa.resample(res_type="kaiser_best") = resampy.resample that the reference runs on every input file
// (speech_anime/model/eval_utils.py:76-86: decode at 44.1 kHz, then 44.1 kHz -> hparams.audio.sample_rate;
// saber/data/audio/io.py:9-15).  librosa / resampy are third-party and absent from the reference tree: the published
// algorithm is restated (resampy/interpn.py resample_f; filter kaiser_best = Kaiser-windowed sinc, 64 zero crossings, 512
// table entries per crossing) -- parity unpinned, see oracle/resample_oracle.py.
//
// One thread per output sample.  Output t reads the time register treg[t] (accumulated sequentially in float64 on the
// host, as the reference accumulates it), walks the left wing of the filter from the fractional offset in steps of
// `step` table entries, then the right wing; every tap is  weight = win[k] + eta * delta[k]  (float64), product with the
// float32 sample in float64, added to the running sum, which is ROUNDED TO FLOAT32 AFTER EVERY TAP (resampy accumulates
// into the float32 output array).  All operations are issued with explicit rounding intrinsics: no FMA contraction.
// The table (2 x 256 KiB) is L2-resident; a 10 s clip is 80-441 k outputs x 128-712 taps -- an ingest step, not a hot loop.
#include "common.h"
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void resample_kernel(ResampleArgs a) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_out) return;
    if (t >= a.n_res) { a.y[t] = 0.f; return; }          // librosa util.fix_length: zero padding up to ceil(n * ratio)
    const double tr = a.treg[t];
    const int64_t n = (int64_t)tr;
    double frac = dmul_exact(a.scale, dsub_exact(tr, (double)n));
    float acc = 0.f;
#pragma unroll 1
    for (int wing = 0; wing < 2; ++wing) {
        if (wing) frac = dsub_exact(a.scale, frac);
        const double index_frac = dmul_exact(frac, (double)a.num_table);
        const int64_t offset = (int64_t)index_frac;
        const double eta = dsub_exact(index_frac, (double)offset);
        const int64_t room = (a.nwin - offset) / a.step;
        const int64_t have = wing ? a.n_in - n - 1 : n + 1;
        const int64_t kmax = have < room ? have : room;
        const double *__restrict__ wp = a.win + offset, *__restrict__ dp = a.delta + offset;
        const float *__restrict__ xp = a.x + (wing ? n + 1 : n);
        const int64_t xs = wing ? 1 : -1;
        for (int64_t i = 0; i < kmax; ++i) {
            const double w = dadd_exact(wp[i * a.step], dmul_exact(eta, dp[i * a.step]));
            acc = (float)dadd_exact((double)acc, dmul_exact(w, (double)xp[i * xs]));
        }
    }
    a.y[t] = acc;
}

}  // namespace

hipError_t sdfa_launch_resample(const ResampleArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((a.n_out + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

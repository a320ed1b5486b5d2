// C ABI of libsdfa_hip.so (include/sdfa_hip.h): host orchestration, weight packing, workspace layout.
#include "../../include/sdfa_hip.h"
#include "kernels.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

thread_local int g_sdfa_mel_fft_radix4 = 0;     // "mel_fft_radix4" option (read by frontend.hip)
thread_local int g_sdfa_gather_plain_order = 0; // "gather_plain_order" option (read by frontend.hip)
thread_local int g_sdfa_frontend_two_kernel = 0;    // "frontend_two_kernel": 1 = share map + mel_columns + gather_features (rounds 2-4) instead of the spectral stream
thread_local int g_sdfa_frontend_stream_block = 0; // "frontend_stream_block" / "frontend_stream_slots": segment geometry of the spectral stream (0 = default)
thread_local int g_sdfa_frontend_stream_slots = 0;
thread_local int g_sdfa_frontend_stream_spin_max = 0; // "frontend_stream_spin_max" (tests): bound of the producer / consumer hand-off waits in polls (0 = the kernel's 4 M); 1 makes them expire, the repair pass redoes the call in the barrier form
thread_local int g_sdfa_frontend_stream_phases = 0; // "frontend_stream_phases": 1 = the stream kernel's workgroups alternate between transforming and emitting (a barrier pair per phase) instead of producer / consumer waves
thread_local int g_sdfa_frontend_t_major = 0;   // "frontend_t_major" option: the front end's distinct columns numbered time-step-major (rounds 2-3)

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) return fail(SDFA_EHIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------------
// front-end constants (window, twiddles, sparse mel rows), cached per sample rate
// ------------------------------------------------------------------------------------------------
struct FrontendCache {
    FrontendConsts c{};
    void *blob = nullptr;
};
std::mutex g_fe_mu;
std::map<int, FrontendCache> g_fe;

// Slaney mel scale / triangular filters as librosa 0.8.0 filters.mel(norm="slaney") defines them
// (third-party, un-vendored; called at saber/data/audio/features/misc.py:110-117).
double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

int build_frontend(int sr, FrontendCache &fc) {
    const int win = (int)(0.064 * sr), hop = (int)(0.008 * sr);
    if (!((sr == 8000 && win == 512) || (sr == 16000 && win == 1024)))
        return fail(SDFA_EINVAL, "sample_rate %d unsupported: the FFT kernels cover 8000 (win 512) and 16000 (win 1024)", sr);
    const int nbins = win / 2 + 1, n_mels = 128;
    const double fmin = 50.0, fmax = 3600.0;
    std::vector<float> hamm(win);
    std::vector<float> tw(2 * win);
    for (int n = 0; n < win; ++n) {
        hamm[n] = (float)(0.54 - 0.46 * std::cos(2.0 * M_PI * n / (win - 1)));   // np.hamming, misc.py:94-100
        tw[2 * n] = (float)std::cos(-2.0 * M_PI * n / win);
        tw[2 * n + 1] = (float)std::sin(-2.0 * M_PI * n / win);
    }
    std::vector<double> mel_f(n_mels + 2);
    const double m_lo = hz_to_mel(fmin), m_hi = hz_to_mel(fmax);
    for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (n_mels + 1));
    // sparse mel rows in a fixed-width form: a band's non-zero bins are consecutive (triangular filters), the widest
    // band has 8 of them -- first bin + 8 weights (zero padded), so the kernels' band loop unrolls
    std::vector<int> bin0(n_mels, 0);
    std::vector<float> w8((size_t)8 * n_mels, 0.f);
    int used = 0, nnz = 0;
    for (int i = 0; i < n_mels; ++i) {
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        int first = -1, last = -1;
        std::vector<float> row(nbins, 0.f);
        for (int b = 0; b < nbins; ++b) {
            const double fr = (double)sr / 2 * b / (nbins - 1);
            const double lower = -(mel_f[i] - fr) / (mel_f[i + 1] - mel_f[i]);
            const double upper = (mel_f[i + 2] - fr) / (mel_f[i + 2] - mel_f[i + 1]);
            const float w = (float)std::fmax(0.0, std::fmin(lower, upper));
            const float wn = (float)((double)w * enorm);   // float32 weights *= float64 enorm -> float32
            row[b] = wn;
            if (wn != 0.f) { if (first < 0) first = b; last = b; ++nnz; }
        }
        if (first < 0) { first = 0; last = 0; }             // an empty band (none at these settings): all-zero taps
        if (last - first + 1 > 8) return fail(SDFA_EINVAL, "mel band %d spans %d bins: the kernels hold 8 taps per band", i, last - first + 1);
        for (int b = first; b <= last; ++b)
            if (row[b] == 0.f) return fail(SDFA_EINVAL, "mel band %d is not contiguous", i);
        bin0[i] = first;
        for (int e = 0; e < 8 && first + e < nbins; ++e) w8[(size_t)e * n_mels + i] = (first + e <= last) ? row[first + e] : 0.f;
        if (last + 1 > used) used = last + 1;
    }
    // the kernels keep 256 power bins per column and read 8 taps from a band's first bin on
    for (int i = 0; i < n_mels; ++i)
        if (bin0[i] + 8 > 256) return fail(SDFA_EINVAL, "mel filterbank does not fit the kernel tables (band %d starts at bin %d)", i, bin0[i]);
    const size_t o_h = 0, o_t = o_h + win * 4, o_p = o_t + win * 8, o_w = o_p + 128 * 4, total = o_w + 1024 * 4;
    std::vector<char> host(total, 0);
    memcpy(&host[o_h], hamm.data(), win * 4);
    memcpy(&host[o_t], tw.data(), win * 8);
    memcpy(&host[o_p], bin0.data(), bin0.size() * 4);
    memcpy(&host[o_w], w8.data(), w8.size() * 4);
    HIP_TRY(hipMalloc(&fc.blob, total));
    HIP_TRY(hipMemcpy(fc.blob, host.data(), total, hipMemcpyHostToDevice));
    char *d = (char *)fc.blob;
    fc.c.hamm = (const float *)(d + o_h);
    fc.c.twiddle = (const float2 *)(d + o_t);
    fc.c.mel_bin0 = (const int *)(d + o_p);
    fc.c.mel_w8 = (const float *)(d + o_w);
    fc.c.win = win; fc.c.hop = hop; fc.c.sliding = hop * 63 + win;
    fc.c.nbins_used = used; fc.c.nnz = nnz;
    return SDFA_OK;
}

}  // namespace

// ================================================================================================
// model
// ================================================================================================
struct sdfa_model {
    int head = SDFA_HEAD_DGRAD;
    bool finalized = false;
    bool keep = false;      // debug: no workspace aliasing, so taps stay valid
    bool profile = false;
    std::map<std::string, std::vector<float>> host;
    void *blob = nullptr;   // all packed weights
    // device pointers into blob
    const float *w1, *b1, *s1, *t1, *w2, *b2, *s2, *t2, *w3, *b3, *s3, *t3;
    const float *fl_w, *fl_b, *fp_w, *fp_b;
    const void *fl_wb = nullptr;   // frequency-LSTM weights as bf16 hi/lo planes (mixed-precision modes)
    const void *cv_wb = nullptr;   // conv stack weights as bf16 planes in the K order of conv123_bf16_kernel (mixed-precision modes)
    int precision = SDFA_PREC_FP32;
    const float *gx_w[2], *tl_w[2];
    const void *tl_wb[2] = {nullptr, nullptr};   // BiLSTM recurrent weights as bf16 hi/lo planes (mixed-precision modes)
    const float *tl_w16[2] = {nullptr, nullptr}; // BiLSTM recurrent weights in the operand order of time_lstm_split16_kernel
    const float *kp_w, *qc_w, *qp_w, *at_v, *at_b;
    struct Fc { const float *w, *b, *cw; int K, P, Ppad, Pstore; int act; };
    Fc trunk, br[2][3], off[3];
    // PCA expansion: dgrad = two bases (scale K 96 -> 6 of every 9 output columns, rotat K 192 -> the other 3);
    // offsets = one basis (K 64)
    int pca_n = 0;
    const float *pca_q[2], *pca_bias[2];
    const void *pca_qb = nullptr;              // dgrad head: both bases as bf16 octets (hi | lo planes) for the split-bf16 PCA kernel (pack_pca_bf16)
    int pca_K[2], pca_k0[2], pca_group[2], pca_off[2];
    std::atomic<int> freq_shape{9};   // launch form of the fp32 frequency LSTM (kernels.h FreqLstmArgs::shape); sdfa_model_autotune measures and sets it
                                      // (atomic: forwards on other threads may read it while an autotune call stores the winner)
    std::atomic<int> reserved_cus{0}; // CUs the persistent kernels leave free (sdfa_model_set_reserved_cus)
    int64_t pca_ld[2], pca_cols[2];
    int64_t out_dim, coef_dim;
    // profiling
    struct Ev { std::string stage; hipEvent_t a, b; };
    mutable std::vector<Ev> events;
    mutable std::mutex ev_mu;   // profiling appends events from const forward calls, possibly on several threads
};

namespace {

const std::vector<float> *get(const sdfa_model *m, const std::string &name, size_t numel) {
    auto it = m->host.find(name);
    if (it == m->host.end()) { fail(SDFA_ESTATE, "tensor '%s' missing", name.c_str()); return nullptr; }
    if (it->second.size() != numel) {
        fail(SDFA_ESTATE, "tensor '%s' has %zu elements, expected %zu", name.c_str(), it->second.size(), numel);
        return nullptr;
    }
    return &it->second;
}

struct Packer {
    std::vector<float> buf;
    size_t add(size_t n) {   // 256-byte aligned slots
        size_t o = (buf.size() + 63) / 64 * 64;
        buf.resize(o + n, 0.f);
        return o;
    }
};

// W [P][K] row-major (+ row permutation) -> K4 [K/4][Ppad][4], zero padded
size_t pack_k4(Packer &pk, const float *W, int P, int K, int ldw, int col0, int Kpad, int Ppad, const int *perm = nullptr) {
    size_t o = pk.add((size_t)Kpad * Ppad);
    for (int p = 0; p < P; ++p) {
        const float *row = W + (size_t)(perm ? perm[p] : p) * ldw + col0;
        for (int k = 0; k < K; ++k) pk.buf[o + ((size_t)(k / 4) * Ppad + p) * 4 + (k % 4)] = row[k];
    }
    return o;
}

// packed gate row p = w*128 + gate*32 + jj  <->  torch row gate*H + 32*w + jj
std::vector<int> gate_perm(int H) {
    std::vector<int> perm(4 * H);
    for (int w = 0; w < H / 32; ++w)
        for (int g = 0; g < 4; ++g)
            for (int jj = 0; jj < 32; ++jj) perm[w * 128 + g * 32 + jj] = g * H + 32 * w + jj;
    return perm;
}

uint16_t bf16_rne_bits(float x) {   // round-to-nearest-even, as the device's float -> __bf16 conversion
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
float bf16_bits_to_float(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

// Conv stack weights for conv123_bf16_kernel (conv.hip): three planes (hi | mid | lo) of 1344 octets of 8 bf16 each,
//   w1 [2 halves][32 co]            k = 8 hh + e: tap df = k / 3, channel c = k % 3 (k >= 9: zero)
//   w2 [6 k-steps][2 halves][64 co] tap df = ks / 2, input channel 8 (2 (ks % 2) + e / 4) + 4 hh + e % 4
//   w3 [4 k-steps][2 halves][64 co] input channel 32 (ks / 2) + 8 (2 (ks % 2) + e / 4) + 4 hh + e % 4
// -- the order in which an accumulator lane of the kernel owns its channels.  Torch layout of w: [co][ci][kf].
void pack_conv_bf16(uint16_t *dst, const float *w1, const float *w2, const float *w3) {
    constexpr size_t PLANE = (size_t)1344 * 8;
    auto put = [&](size_t octet, int e, float x) {
        const uint16_t hi = bf16_rne_bits(x);
        const float r1 = x - bf16_bits_to_float(hi);
        const uint16_t mid = bf16_rne_bits(r1);
        const uint16_t lo = bf16_rne_bits(r1 - bf16_bits_to_float(mid));
        dst[octet * 8 + e] = hi; dst[PLANE + octet * 8 + e] = mid; dst[2 * PLANE + octet * 8 + e] = lo;
    };
    for (int hh = 0; hh < 2; ++hh)
        for (int co = 0; co < 32; ++co)
            for (int e = 0; e < 8; ++e) {
                const int k = 8 * hh + e;
                put((size_t)hh * 32 + co, e, k < 9 ? w1[((size_t)co * 3 + k % 3) * 3 + k / 3] : 0.f);
            }
    for (int ks = 0; ks < 6; ++ks)
        for (int hh = 0; hh < 2; ++hh)
            for (int co = 0; co < 64; ++co)
                for (int e = 0; e < 8; ++e) {
                    const int df = ks >> 1, ci = 8 * (2 * (ks & 1) + (e >> 2)) + 4 * hh + (e & 3);
                    put(64 + ((size_t)ks * 2 + hh) * 64 + co, e, w2[((size_t)co * 32 + ci) * 3 + df]);
                }
    for (int ks = 0; ks < 4; ++ks)
        for (int hh = 0; hh < 2; ++hh)
            for (int co = 0; co < 64; ++co)
                for (int e = 0; e < 8; ++e) {
                    const int ci = 32 * (ks >> 1) + 8 * (2 * (ks & 1) + (e >> 2)) + 4 * hh + (e & 3);
                    put(64 + 768 + ((size_t)ks * 2 + hh) * 64 + co, e, w3[(size_t)co * 64 + ci]);
                }
}

// PCA bases of the dgrad head for pca_dgrad_res_kernel<true> (pca.hip): per triangle block tb (32 triangles: 192 scale + 96 rotat columns)
//   [plane hi | lo] x ( scale: 12 rows r = 2 ks + h of 192 octets | rotat: 24 rows of 96 octets ),  octet e = basis[k = 16 ks + 8 h + e][column]
// -- the B operand of v_mfma_f32_32x32x16_bf16 as a lane reads it.  comp: torch layout [column][k]; k >= kreal and columns past the end are 0.
void pack_pca_bf16(uint16_t *dst, const float *comp_s, const float *comp_r, int64_t cols_s, int64_t cols_r, int64_t ntb) {
    constexpr size_t PLANE = (size_t)(12 * 192 + 24 * 96) * 8;       // bf16 values per plane and triangle block
    for (int64_t tb = 0; tb < ntb; ++tb) {
        uint16_t *blk = dst + (size_t)tb * 2 * PLANE;
        auto put = [&](size_t octet, int e, float x) {
            const uint16_t hi = bf16_rne_bits(x);
            blk[octet * 8 + e] = hi;
            blk[PLANE + octet * 8 + e] = bf16_rne_bits(x - bf16_bits_to_float(hi));
        };
        for (int r = 0; r < 12; ++r)
            for (int c = 0; c < 192; ++c)
                for (int e = 0; e < 8; ++e) {
                    const int k = 8 * r + e;                         // 16 ks + 8 h + e with r = 2 ks + h
                    const int64_t q = tb * 192 + c;
                    put((size_t)r * 192 + c, e, (k < 85 && q < cols_s) ? comp_s[(size_t)q * 85 + k] : 0.f);
                }
        for (int r = 0; r < 24; ++r)
            for (int c = 0; c < 96; ++c)
                for (int e = 0; e < 8; ++e) {
                    const int k = 8 * r + e;
                    const int64_t q = tb * 96 + c;
                    put((size_t)12 * 192 + (size_t)r * 96 + c, e, (k < 180 && q < cols_r) ? comp_r[(size_t)q * 180 + k] : 0.f);
                }
    }
}

// Frequency-LSTM weights for freq_lstm_bf16_kernel / freq_lstm_bf16x6_kernel: per direction [plane hi | mid | lo][24 octets][512 gate rows][8] bf16.
// cat = [W_ih | W_hh] rows in torch order, perm = packed gate row -> torch row.  Octets 0..7 are the 64 input features
// in order; octet 8 + o' (o' = 4w + 2q + hh) holds hidden units 32w+16q+4hh+{0..3} and 32w+16q+8+4hh+{0..3} -- the order
// in which a lane of the kernel owns its accumulator rows (lstm.hip).
void pack_freq_lstm_bf16(uint16_t *dst, const float *cat, const int *perm) {
    for (int O = 0; O < 24; ++O)
        for (int p = 0; p < 512; ++p)
            for (int e = 0; e < 8; ++e) {
                int k;
                if (O < 8) k = 8 * O + e;
                else {
                    const int o = O - 8, w = o >> 2, q = (o >> 1) & 1, hh = o & 1;
                    k = 64 + 32 * w + 16 * q + 4 * hh + (e & 3) + 8 * (e >> 2);
                }
                const float x = cat[(size_t)perm[p] * 192 + k];
                const uint16_t hi = bf16_rne_bits(x);
                const float r1 = x - bf16_bits_to_float(hi);
                const uint16_t mid = bf16_rne_bits(r1);
                const uint16_t lo = bf16_rne_bits(r1 - bf16_bits_to_float(mid));
                dst[((size_t)O * 512 + p) * 8 + e] = hi;
                dst[((size_t)(24 + O) * 512 + p) * 8 + e] = mid;      // the "lo" plane of the three-product split
                dst[((size_t)(48 + O) * 512 + p) * 8 + e] = lo;       // third term: six-product split only
            }
}

// Recurrent weights of one BiLSTM direction for time_lstm_bf16_kernel: [plane hi | mid | lo][32 octets][1024 gate rows][8] bf16,
// the K axis (256 hidden units) in the accumulator-row order of the kernel (same octet rule as above).
void pack_rec_bf16(uint16_t *dst, const float *whh, const int *perm) {
    for (int o = 0; o < 32; ++o)
        for (int p = 0; p < 1024; ++p)
            for (int e = 0; e < 8; ++e) {
                const int w = o >> 2, q = (o >> 1) & 1, hh = o & 1;
                const int k = 32 * w + 16 * q + 4 * hh + (e & 3) + 8 * (e >> 2);
                const float x = whh[(size_t)perm[p] * 256 + k];
                const uint16_t hi = bf16_rne_bits(x);
                const float r1 = x - bf16_bits_to_float(hi);
                const uint16_t mid = bf16_rne_bits(r1);
                const uint16_t lo = bf16_rne_bits(r1 - bf16_bits_to_float(mid));
                dst[((size_t)o * 1024 + p) * 8 + e] = hi;
                dst[((size_t)(32 + o) * 1024 + p) * 8 + e] = mid;
                dst[((size_t)(64 + o) * 1024 + p) * 8 + e] = lo;
            }
}

// Recurrent weights of one BiLSTM direction for time_lstm_split16_kernel (v_mfma_f32_16x16x4_f32): float4 [16 K16][4 g][1024 gate rows],
// element j of (K16, g, p) = W_hh[perm[p]][k], k = 8 kb + {0, 4, 1, 5}[g] (m = 0) / {2, 6, 3, 7}[g] (m = 1) with kb = 2 K16 + (j >> 1),
// m = j & 1 -- the order in which the 32x32x2 kernels add a gate row's products (lstm.hip), so both forms give the same bits.
void pack_rec_16x16x4(float *dst, const float *whh, const int *perm) {
    static const int kk[2][4] = {{0, 4, 1, 5}, {2, 6, 3, 7}};
    for (int K16 = 0; K16 < 16; ++K16)
        for (int g = 0; g < 4; ++g)
            for (int p = 0; p < 1024; ++p)
                for (int j = 0; j < 4; ++j) {
                    const int kb = 2 * K16 + (j >> 1), m = j & 1, k = 8 * kb + kk[m][g];
                    dst[(((size_t)K16 * 4 + g) * 1024 + p) * 4 + j] = whh[(size_t)perm[p] * 256 + k];
                }
}

int pack_fc(sdfa_model *m, Packer &pk, const std::string &key, int P, int Kin, bool cond, int act, size_t off[3],
            sdfa_model::Fc &fc) {
    const int Ktot = cond ? Kin + 8 : Kin;
    auto *w = get(m, key + ".weight", (size_t)P * Ktot);
    auto *b = get(m, key + ".bias", P);
    if (!w || !b) return SDFA_ESTATE;
    fc.K = Kin; fc.P = P; fc.Ppad = (int)round_up(P, 128); fc.Pstore = (int)round_up(P, 32); fc.act = act;
    off[0] = pack_k4(pk, w->data(), P, Kin, Ktot, 0, Kin, fc.Ppad);
    off[1] = pk.add(fc.Ppad);
    memcpy(&pk.buf[off[1]], b->data(), P * 4);
    off[2] = (size_t)-1;
    if (cond) {
        off[2] = pk.add((size_t)fc.Ppad * 8);
        for (int p = 0; p < P; ++p)
            for (int s = 0; s < 8; ++s) pk.buf[off[2] + ((size_t)(p / 4) * 8 + s) * 4 + (p % 4)] = (*w)[(size_t)p * Ktot + Kin + s];
    }
    return SDFA_OK;
}

}  // namespace

extern "C" {

int sdfa_abi_version(void) { return SDFA_ABI_VERSION; }
const char *sdfa_last_error(void) { return g_err.c_str(); }

// ------------------------------------------------------------------------------------------------
int64_t sdfa_frame_index(int64_t n_samples, int sample_rate, int fps, int win, int hop, int ts_delta_ms,
                         int64_t *h_starts, int32_t *h_tslist, int64_t cap) {
    if (n_samples <= 0 || sample_rate <= 0 || fps <= 0 || win <= 0 || hop <= 0) return fail(SDFA_EINVAL, "bad frame_index arguments");
    // the front-end kernels index a clip's samples in 32-bit arithmetic (|index| < 2^29: frontend.hip); refuse longer clips here,
    // where the length is known on the host (9 h at 16 kHz; the reference's float32 frame arithmetic is exact only up to 2^24)
    if (n_samples > 0x1fffffff) return fail(SDFA_EINVAL, "frame_index: clips of more than 2^29 - 1 samples are not supported (%lld given)", (long long)n_samples);
    const int64_t sliding = (int64_t)hop * 63 + win;
    int64_t count = 0;
    double idx = -1.0;
    for (;;) {
        // frame_to_sample: np.float32(float(idx * sr) / float(fps))          speech_anime.py:141-145
        const float fs = (float)((idx * (double)sample_rate) / (double)fps);
        // frame_in_range: float32 + int -> float32                           sliding_window.py:320-322
        const float lhs = fs + (float)sliding;
        if (!((double)lhs <= (double)(n_samples + 2 * sliding))) break;
        const int64_t mid = (int64_t)std::floor((double)fs);
        const int64_t e = mid + sliding / 2, s = e - sliding;
        // sample_to_ms: np.float32(float(((s+e)/2) * 1000.0) / float(sr)); then float32 - ts_delta; round half even
        const float ms = (float)(((((double)(s + e)) / 2.0) * 1000.0) / (double)sample_rate);
        const float shifted = ms - (float)ts_delta_ms;
        const int32_t ts = (int32_t)std::nearbyintf(shifted);
        const int64_t lo = s > 0 ? s : 0, hi = e < n_samples ? e : n_samples;
        if (hi > lo && s < 0 && e > n_samples)
            return fail(SDFA_ESHORTCLIP, "signal length %lld != %lld.", (long long)(hi - lo - s), (long long)sliding);
        if (count < cap) {
            if (h_starts) h_starts[count] = s;
            if (h_tslist) h_tslist[count] = ts;
        }
        ++count;
        idx += 1.0;
    }
    if (cap > 0 && count > cap) return fail(SDFA_ENOSPACE, "frame_index: %lld frames, capacity %lld", (long long)count, (long long)cap);
    return count;
}

// ------------------------------------------------------------------------------------------------
int sdfa_mel_frontend(const float *d_pcm, const int64_t *d_clip_off, const int64_t *d_clip_len, int32_t n_clips,
                      const int32_t *d_frame_clip, const int64_t *d_frame_start, int64_t n_frames, int sample_rate,
                      float *d_audio_feat, void *stream) {
    if (n_frames == 0) return SDFA_OK;
    if (!d_pcm || !d_clip_off || !d_clip_len || !d_frame_clip || !d_frame_start || !d_audio_feat || n_clips <= 0 || n_frames < 0)
        return fail(SDFA_EINVAL, "mel_frontend: null pointer or bad count");
    FrontendConsts c;
    {
        std::lock_guard<std::mutex> lk(g_fe_mu);
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        auto key = sample_rate * 64 + dev;
        auto it = g_fe.find(key);
        if (it == g_fe.end()) {
            FrontendCache fc;
            int rc = build_frontend(sample_rate, fc);
            if (rc) return rc;
            it = g_fe.emplace(key, fc).first;
        }
        c = it->second.c;
    }
    HIP_TRY(sdfa_launch_frontend(c, d_pcm, d_clip_off, d_clip_len, d_frame_clip, d_frame_start, n_frames, d_audio_feat,
                                 (hipStream_t)stream));
    return SDFA_OK;
}

// ------------------------------------------------------------------------------------------------
// "spectral gather" form of the front end (frontend.hip): share map over mel columns -> FFT + mel of the distinct
// columns -> per-frame gather.  Scratch: the map (ints) followed by the mel table (128 floats per column, sized for the
// case that nothing is shared).
namespace {
struct FeWs { int64_t Nc, Mc, map_ints, table_off, total; };
FeWs fe_layout(int64_t n_frames) {
    FeWs w;
    w.Nc = round_up(n_frames, 128); w.Mc = 64 * w.Nc;
    w.map_ints = round_up(16 + 2 * w.Nc + 5 * w.Mc + w.Mc / 1024 + 2, 64);
    w.table_off = w.map_ints * 4;
    w.total = w.table_off + w.Mc * 128 * 4;
    return w;
}
}  // namespace

// The spectral-stream kernel's status word of the LAST sdfa_mel_frontend_gather call on this workspace: bounded hand-off waits that
// expired (0 always, unless the producer / consumer form's logic is wrong) -- the repair pass behind the kernel redid such a call, the features are right
// either way.  Synchronises the stream.  Tests only.
int sdfa_debug_frontend_status(const void *d_workspace, void *stream) {
    if (!d_workspace) return fail(SDFA_EINVAL, "frontend_status: null workspace");
    int32_t v = 0;
    HIP_TRY(hipMemcpyAsync(&v, reinterpret_cast<const int32_t *>(d_workspace) + 8, sizeof v, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return (int)v;
}

int64_t sdfa_frontend_workspace_bytes(int64_t max_frames) {
    if (max_frames <= 0) return fail(SDFA_EINVAL, "frontend_workspace_bytes: bad argument");
    return fe_layout(max_frames).total;
}

int sdfa_mel_frontend_gather(const float *d_pcm, const int64_t *d_clip_off, const int64_t *d_clip_len, int32_t n_clips,
                             const int32_t *d_frame_clip, const int64_t *d_frame_start, int64_t n_frames, int sample_rate,
                             float *d_audio_feat, void *d_workspace, int64_t workspace_bytes, void *stream) {
    if (n_frames == 0) return SDFA_OK;
    if (!d_pcm || !d_clip_off || !d_clip_len || !d_frame_clip || !d_frame_start || !d_audio_feat || !d_workspace || n_clips <= 0 || n_frames < 0)
        return fail(SDFA_EINVAL, "mel_frontend_gather: null pointer or bad count");
    if (((uintptr_t)d_workspace | (uintptr_t)d_audio_feat) & 15) return fail(SDFA_EINVAL, "mel_frontend_gather: pointers must be 16-byte aligned");
    const FeWs w = fe_layout(n_frames);
    if (workspace_bytes < w.total)
        return fail(SDFA_ENOSPACE, "mel_frontend_gather: workspace of %lld bytes, %lld needed for %lld frames", (long long)workspace_bytes,
                    (long long)w.total, (long long)n_frames);
    if (w.Mc >= (int64_t)1 << 31) return fail(SDFA_EINVAL, "mel_frontend_gather: too many frames in one call");
    FrontendConsts c;
    {
        std::lock_guard<std::mutex> lk(g_fe_mu);
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        auto key = sample_rate * 64 + dev;
        auto it = g_fe.find(key);
        if (it == g_fe.end()) {
            FrontendCache fc;
            int rc = build_frontend(sample_rate, fc);
            if (rc) return rc;
            it = g_fe.emplace(key, fc).first;
        }
        c = it->second.c;
    }
    hipStream_t s = (hipStream_t)stream;
    int32_t *sh = reinterpret_cast<int32_t *>(d_workspace);
    ShareArgs sa{};
    sa.frame_clip = d_frame_clip; sa.frame_start = d_frame_start; sa.hop = c.hop;
    sa.t_lo = 1; sa.t_hi = 63;          // every window column but the first (raw first sample) is a function of (clip, position)
    sa.frame_major = g_sdfa_frontend_t_major ? 0 : 1;      // distinct columns numbered clip by clip, hop by hop, per-column arrays indexed [n][t] (share.hip: col_index)
    sa.N = n_frames; sa.Nc = w.Nc; sa.Mc = w.Mc;
    sa.counts = reinterpret_cast<int64_t *>(sh);
    sa.prev = sh + 16; sa.shift = sa.prev + w.Nc;
    sa.owner = sa.shift + w.Nc; sa.flag = sa.owner + w.Mc; sa.uid = sa.flag + w.Mc;
    sa.col_src = sa.uid + w.Mc; sa.col_to_u = sa.col_src + w.Mc; sa.tile_sum = sa.col_to_u + w.Mc;
    // sh[8]: the stream kernel's status word (bounded hand-off waits that expired -- never, unless its logic is wrong -- and were
    // repaired by the pass behind the kernel).  Zeroed by EVERY call, whichever form runs, so that sdfa_debug_frontend_status never
    // reads a stale or uninitialised word after a two-kernel / radix-4 / t-major call or on a fresh workspace.
    HIP_TRY(hipMemsetAsync(sh + 8, 0, sizeof(int32_t), s));
    if (!g_sdfa_frontend_two_kernel && !g_sdfa_mel_fft_radix4 && !g_sdfa_frontend_t_major) {
        // spectral stream (frontend.hip): the chains are read from prev / shift, the mel rows live in an LDS ring, no table
        HIP_TRY(sdfa_launch_share_prev(sa, s));
        HIP_TRY(sdfa_launch_mel_stream(c, d_pcm, d_clip_off, d_clip_len, d_frame_clip, d_frame_start, sa.prev, sa.shift, n_frames,
                                       g_sdfa_frontend_stream_block, g_sdfa_frontend_stream_slots, g_sdfa_frontend_stream_phases ? 0 : 1, g_sdfa_frontend_stream_spin_max, sh + 8, d_audio_feat, s));
        return SDFA_OK;
    }
    HIP_TRY(sdfa_launch_share_map(sa, s));
    float *table = reinterpret_cast<float *>(reinterpret_cast<char *>(d_workspace) + w.table_off);
    HIP_TRY(sdfa_launch_mel_columns(c, d_pcm, d_clip_off, d_clip_len, d_frame_clip, d_frame_start, sa.col_src, sa.counts, table, s));
    HIP_TRY(sdfa_launch_gather_features(table, sa.col_to_u, n_frames, w.Nc, sa.frame_major, d_audio_feat, s));
    return SDFA_OK;
}

// ------------------------------------------------------------------------------------------------
sdfa_model *sdfa_model_create(int head) {
    if (head != SDFA_HEAD_DGRAD && head != SDFA_HEAD_OFFSETS) { fail(SDFA_EINVAL, "unknown head %d", head); return nullptr; }
    auto *m = new sdfa_model();
    m->head = head;
    m->out_dim = head == SDFA_HEAD_DGRAD ? SDFA_DGRAD_DIM : SDFA_OFFSETS_DIM;
    m->coef_dim = head == SDFA_HEAD_DGRAD ? SDFA_COEF_SCALE + SDFA_COEF_ROTAT : SDFA_COEF_OFFSETS;
    return m;
}

void sdfa_model_destroy(sdfa_model *m) {
    if (!m) return;
    if (m->blob) (void)hipFree(m->blob);
    for (auto &e : m->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    delete m;
}

int sdfa_model_set_tensor(sdfa_model *m, const char *name, const float *h_data, int64_t numel) {
    if (!m || !name || !h_data || numel <= 0) return fail(SDFA_EINVAL, "set_tensor: bad argument");
    if (m->finalized) return fail(SDFA_ESTATE, "set_tensor after finalize");
    m->host[name].assign(h_data, h_data + numel);
    return SDFA_OK;
}

int sdfa_model_head(const sdfa_model *m) { return m ? m->head : SDFA_EINVAL; }
int64_t sdfa_model_out_dim(const sdfa_model *m) { return m ? m->out_dim : SDFA_EINVAL; }
int64_t sdfa_model_coef_dim(const sdfa_model *m) { return m ? m->coef_dim : SDFA_EINVAL; }

int sdfa_model_finalize(sdfa_model *m, void *stream) {
    if (!m) return fail(SDFA_EINVAL, "null model");
    if (m->finalized) return SDFA_OK;
    Packer pk;
    const std::string enc = "_audio_encoder._layers.";
    // ---- conv stack: fold eval BatchNorm (eps 1e-3) into scale/shift applied AFTER LeakyReLU (extend.py:94-101)
    size_t o_conv[3][4];
    const std::vector<float> *conv_w[3] = {nullptr, nullptr, nullptr};
    const int cshape[3][3] = {{32, 3, 3}, {64, 32, 3}, {64, 64, 1}};   // co, ci, kf
    const int cidx[3] = {1, 3, 5};
    for (int l = 0; l < 3; ++l) {
        const int co = cshape[l][0], ci = cshape[l][1], kf = cshape[l][2];
        const std::string k = enc + std::to_string(cidx[l]);
        auto *w = get(m, k + ".weight", (size_t)co * ci * kf);
        auto *b = get(m, k + ".bias", co);
        auto *g = get(m, k + "._ext_post_bn.weight", co), *be = get(m, k + "._ext_post_bn.bias", co);
        auto *mu = get(m, k + "._ext_post_bn.running_mean", co), *var = get(m, k + "._ext_post_bn.running_var", co);
        if (!w || !b || !g || !be || !mu || !var) return SDFA_ESTATE;
        conv_w[l] = w;
        if (l == 0) {   // A operand [5 k-steps][2 halves][32 co], k = df*3 + c, k = 9 -> 0
            o_conv[0][0] = pk.add(5 * 2 * 32);
            for (int s = 0; s < 5; ++s)
                for (int hh = 0; hh < 2; ++hh)
                    for (int o = 0; o < 32; ++o) {
                        const int kk = 2 * s + hh;
                        float v = 0.f;
                        if (kk < 9) { const int df = kk / 3, c = kk % 3; v = (*w)[((size_t)o * 3 + c) * 3 + df]; }
                        pk.buf[o_conv[0][0] + (s * 2 + hh) * 32 + o] = v;
                    }
        } else {        // K4 [K/4][64][4], k = df*ci + c
            const int K = ci * kf;
            std::vector<float> flat((size_t)co * K);
            for (int o = 0; o < co; ++o)
                for (int c = 0; c < ci; ++c)
                    for (int df = 0; df < kf; ++df) flat[(size_t)o * K + df * ci + c] = (*w)[((size_t)o * ci + c) * kf + df];
            o_conv[l][0] = pack_k4(pk, flat.data(), co, K, K, 0, K, co);
        }
        o_conv[l][1] = pk.add(co); o_conv[l][2] = pk.add(co); o_conv[l][3] = pk.add(co);
        for (int o = 0; o < co; ++o) {
            const double sc = (double)(*g)[o] / std::sqrt((double)(*var)[o] + 1e-3);
            pk.buf[o_conv[l][1] + o] = (*b)[o];
            pk.buf[o_conv[l][2] + o] = (float)sc;
            pk.buf[o_conv[l][3] + o] = (float)((double)(*be)[o] - (double)(*mu)[o] * sc);
        }
    }
    const size_t o_cvwb = pk.add((size_t)3 * 1344 * 8 / 2);      // three bf16 planes of 1344 octets, two bf16 per float slot
    pack_conv_bf16(reinterpret_cast<uint16_t *>(&pk.buf[o_cvwb]), conv_w[0]->data(), conv_w[1]->data(), conv_w[2]->data());
    // ---- frequency LSTM: [W_ih | W_hh] concatenated along K, gate rows packed per wave; bias = b_ih + b_hh
    size_t o_flw = pk.add(0), o_flb, o_flwb;
    {
        o_flwb = pk.add((size_t)2 * 3 * 24 * 512 * 8 / 2);   // two directions x three bf16 planes, two bf16 per float slot
        const auto perm = gate_perm(128);
        const char *suf[2] = {"", "_reverse"};
        std::vector<float> cat((size_t)512 * 192);
        size_t first = 0;
        std::vector<float> bias(1024);
        for (int d = 0; d < 2; ++d) {
            const std::string k = enc + "6._lstm.";
            auto *wih = get(m, k + "weight_ih_l0" + suf[d], 512 * 64), *whh = get(m, k + "weight_hh_l0" + suf[d], 512 * 128);
            auto *bih = get(m, k + "bias_ih_l0" + suf[d], 512), *bhh = get(m, k + "bias_hh_l0" + suf[d], 512);
            if (!wih || !whh || !bih || !bhh) return SDFA_ESTATE;
            // The cell update needs its gate pre-activations as exponents of two: sigmoid(x) = 1 / (1 + 2^(-x log2 e)), tanh(g) through
            // 2^(-2 g log2 e).  The scaling is folded into the weights and the bias here (rows i, f, o by log2 e, rows g by 2 log2 e:
            // torch gate order i, f, g, o), so the kernel's accumulators ARE the exponents and the update has five vector multiplies
            // per element pair less (round 4; lstm.hip: lstm_cell_quad<true>).  A weight picks up one more fp32 rounding; the stage
            // stays inside its 1e-4 tap tolerance (tests/test_gpu_parity.py) and all launch forms share the packed weights.
            for (int r = 0; r < 512; ++r) {
#ifdef SDFA_OLD_CELL   /* A/B build only (make EXP=OLD_CELL): round 3's unscaled weights + cell update */
                const float k = 1.0f;
#else
                const float k = (r / 128 == 2) ? 2.8853900817779268f : 1.4426950408889634f;
#endif
                for (int j = 0; j < 64; ++j) cat[(size_t)r * 192 + j] = (*wih)[(size_t)r * 64 + j] * k;
                for (int j = 0; j < 128; ++j) cat[(size_t)r * 192 + 64 + j] = (*whh)[(size_t)r * 128 + j] * k;
            }
            size_t o = pack_k4(pk, cat.data(), 512, 192, 192, 0, 192, 512, perm.data());
            pack_freq_lstm_bf16(reinterpret_cast<uint16_t *>(&pk.buf[o_flwb]) + (size_t)d * 3 * 24 * 512 * 8, cat.data(), perm.data());
            if (d == 0) first = o;
            else if (o != first + (size_t)48 * 512 * 4) return fail(SDFA_ESTATE, "internal: freq-lstm weights not contiguous");
            for (int p = 0; p < 512; ++p) {
#ifdef SDFA_OLD_CELL
                bias[d * 512 + p] = (*bih)[perm[p]] + (*bhh)[perm[p]];
#else
                bias[d * 512 + p] = ((*bih)[perm[p]] + (*bhh)[perm[p]]) * ((perm[p] / 128 == 2) ? 2.8853900817779268f : 1.4426950408889634f);
#endif
            }
        }
        o_flw = first;
        o_flb = pk.add(1024);
        memcpy(&pk.buf[o_flb], bias.data(), 1024 * 4);
    }
    size_t o_fpw, o_fpb;
    {
        auto *w = get(m, enc + "6._proj.weight", (size_t)256 * 8192), *b = get(m, enc + "6._proj.bias", 256);
        if (!w || !b) return SDFA_ESTATE;
        o_fpw = pack_k4(pk, w->data(), 256, 8192, 8192, 0, 8192, 256);
        o_fpb = pk.add(256);
        memcpy(&pk.buf[o_fpb], b->data(), 256 * 4);
    }
    // ---- time BiLSTM (bias=False): input projections as one 2048-row GEMM per layer, recurrent weights K4
    size_t o_gx[2], o_tl[2], o_tlb[2], o_tl16[2];
    {
        const auto perm = gate_perm(256);
        for (int l = 0; l < 2; ++l) o_tlb[l] = pk.add((size_t)2 * 3 * 32 * 1024 * 8 / 2);   // two directions x three bf16 planes, two bf16 per float slot
        for (int l = 0; l < 2; ++l) o_tl16[l] = pk.add((size_t)2 * 16 * 4 * 1024 * 4);      // two directions, 16x16x4 operand order
        const char *suf[2] = {"", "_reverse"};
        for (int l = 0; l < 2; ++l) {
            const int Kin = l == 0 ? 256 : 512;
            std::vector<float> both((size_t)2048 * Kin);
            size_t first = 0;
            for (int d = 0; d < 2; ++d) {
                auto *wih = get(m, enc + "9.weight_ih_l" + std::to_string(l) + suf[d], (size_t)1024 * Kin);
                auto *whh = get(m, enc + "9.weight_hh_l" + std::to_string(l) + suf[d], (size_t)1024 * 256);
                if (!wih || !whh) return SDFA_ESTATE;
                for (int p = 0; p < 1024; ++p) memcpy(&both[((size_t)d * 1024 + p) * Kin], &(*wih)[(size_t)perm[p] * Kin], Kin * 4);
                size_t o = pack_k4(pk, whh->data(), 1024, 256, 256, 0, 256, 1024, perm.data());
                pack_rec_bf16(reinterpret_cast<uint16_t *>(&pk.buf[o_tlb[l]]) + (size_t)d * 3 * 32 * 1024 * 8, whh->data(), perm.data());
                pack_rec_16x16x4(&pk.buf[o_tl16[l]] + (size_t)d * 16 * 4 * 1024 * 4, whh->data(), perm.data());
                if (d == 0) first = o;
                else if (o != first + (size_t)64 * 1024 * 4) return fail(SDFA_ESTATE, "internal: time-lstm weights not contiguous");
            }
            o_tl[l] = first;
            o_gx[l] = pack_k4(pk, both.data(), 2048, Kin, Kin, 0, Kin, 2048);
        }
    }
    // ---- attention
    size_t o_kp, o_qc, o_qp, o_v, o_b;
    {
        const std::string k = enc + "10.";
        auto *cq = get(m, k + "_conv_query.weight", (size_t)512 * 512 * 3), *wk = get(m, k + "proj_key.weight", 128 * 512);
        auto *wq = get(m, k + "proj_qry.weight", 128 * 512), *v = get(m, k + "v.weight", 128), *b = get(m, k + "b", 128);
        if (!cq || !wk || !wq || !v || !b) return SDFA_ESTATE;
        o_kp = pack_k4(pk, wk->data(), 128, 512, 512, 0, 512, 128);
        o_qp = pack_k4(pk, wq->data(), 128, 512, 512, 0, 512, 128);
        std::vector<float> flat((size_t)512 * 1536);   // k = tap*512 + c
        for (int o = 0; o < 512; ++o)
            for (int c = 0; c < 512; ++c)
                for (int t = 0; t < 3; ++t) flat[(size_t)o * 1536 + t * 512 + c] = (*cq)[((size_t)o * 512 + c) * 3 + t];
        o_qc = pack_k4(pk, flat.data(), 512, 1536, 1536, 0, 1536, 512);
        o_v = pk.add(128); memcpy(&pk.buf[o_v], v->data(), 512);
        o_b = pk.add(128); memcpy(&pk.buf[o_b], b->data(), 512);
    }
    // ---- output module
    const std::string om = "_output_module.";
    size_t o_fc[7][3];
    size_t o_pq[2] = {0, 0}, o_pb[2] = {0, 0}, o_pqb = 0;
    bool have_pqb = false;
    if (m->head == SDFA_HEAD_DGRAD) {
        if (pack_fc(m, pk, om + "_layers.0", 512, 512, true, ACT_LRELU, o_fc[0], m->trunk)) return SDFA_ESTATE;
        const char *brn[2] = {"_scale_layers.", "_rotat_layers."};
        const int nco[2] = {SDFA_COEF_SCALE, SDFA_COEF_ROTAT};
        for (int b = 0; b < 2; ++b) {
            if (pack_fc(m, pk, om + brn[b] + "0", 512, 512, true, ACT_LRELU, o_fc[1 + 3 * b], m->br[b][0])) return SDFA_ESTATE;
            if (pack_fc(m, pk, om + brn[b] + "1", 256, 512, false, ACT_TANH, o_fc[2 + 3 * b], m->br[b][1])) return SDFA_ESTATE;
            if (pack_fc(m, pk, om + brn[b] + "2", nco[b], 256, false, ACT_NONE, o_fc[3 + 3 * b], m->br[b][2])) return SDFA_ESTATE;
        }
        auto *cs = get(m, om + "_scale_pca.compT", (size_t)59856 * 85), *ms = get(m, om + "_scale_pca.means", 59856);
        auto *cr = get(m, om + "_rotat_pca.compT", (size_t)29928 * 180), *mr = get(m, om + "_rotat_pca.means", 29928);
        if (!cs || !ms || !cr || !mr) return SDFA_ESTATE;
        // two bases; the GEMM epilogue scatters column q of basis b to output coordinate (q / g) * 9 + off + q % g,
        // which IS the [s0..s5 r0 r1 r2] interleave of data_to_anime_feat (model.py:246-257)
        m->pca_n = 2;
        const std::vector<float> *comp[2] = {cs, cr}, *mean[2] = {ms, mr};
        const int kreal[2] = {85, 180};
        for (int b = 0; b < 2; ++b) {
            m->pca_K[b] = b ? 192 : 96; m->pca_k0[b] = b ? 96 : 0; m->pca_group[b] = b ? 3 : 6; m->pca_off[b] = b ? 6 : 0;
            m->pca_cols[b] = b ? 29928 : 59856; m->pca_ld[b] = round_up(m->pca_cols[b], 128);
            o_pq[b] = pk.add((size_t)m->pca_K[b] * m->pca_ld[b]);
            o_pb[b] = pk.add(m->pca_ld[b]);
            for (int64_t o = 0; o < m->pca_cols[b]; ++o) {
                const float *row = &(*comp[b])[(size_t)o * kreal[b]];
                for (int k = 0; k < kreal[b]; ++k) pk.buf[o_pq[b] + ((size_t)(k / 4) * m->pca_ld[b] + o) * 4 + (k % 4)] = row[k];
                pk.buf[o_pb[b] + o] = (*mean[b])[o];
            }
        }
        {   // the same bases as bf16 octets for the split-bf16 form of the fused kernel
            const int64_t ntb = (m->pca_cols[1] + 95) / 96;
            o_pqb = pk.add((size_t)ntb * 2 * (12 * 192 + 24 * 96) * 8 / 2);      // two bf16 per float slot
            pack_pca_bf16(reinterpret_cast<uint16_t *>(&pk.buf[o_pqb]), cs->data(), cr->data(), m->pca_cols[0], m->pca_cols[1], ntb);
            have_pqb = true;
        }
    } else {
        if (pack_fc(m, pk, om + "_layers.0", 512, 512, true, ACT_LRELU, o_fc[0], m->off[0])) return SDFA_ESTATE;
        if (pack_fc(m, pk, om + "_layers.1", 256, 512, false, ACT_TANH, o_fc[1], m->off[1])) return SDFA_ESTATE;
        if (pack_fc(m, pk, om + "_layers.2", SDFA_COEF_OFFSETS, 256, false, ACT_NONE, o_fc[2], m->off[2])) return SDFA_ESTATE;
        auto *cp = get(m, om + "_pca.compT", (size_t)SDFA_OFFSETS_DIM * 59), *mp = get(m, om + "_pca.means", SDFA_OFFSETS_DIM);
        if (!cp || !mp) return SDFA_ESTATE;
        m->pca_n = 1;
        m->pca_K[0] = 64; m->pca_k0[0] = 0; m->pca_group[0] = 0; m->pca_off[0] = 0;
        m->pca_cols[0] = SDFA_OFFSETS_DIM; m->pca_ld[0] = round_up(SDFA_OFFSETS_DIM, 128);
        o_pq[0] = pk.add((size_t)m->pca_K[0] * m->pca_ld[0]);
        o_pb[0] = pk.add(m->pca_ld[0]);
        for (int64_t o = 0; o < SDFA_OFFSETS_DIM; ++o) {
            for (int k = 0; k < 59; ++k) pk.buf[o_pq[0] + ((size_t)(k / 4) * m->pca_ld[0] + o) * 4 + (k % 4)] = (*cp)[(size_t)o * 59 + k];
            pk.buf[o_pb[0] + o] = (*mp)[o];
        }
    }
    // ---- upload
    HIP_TRY(hipMalloc(&m->blob, pk.buf.size() * 4));
    HIP_TRY(hipMemcpyAsync(m->blob, pk.buf.data(), pk.buf.size() * 4, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));   // host staging buffer dies with this call
    const float *d = (const float *)m->blob;
    m->w1 = d + o_conv[0][0]; m->b1 = d + o_conv[0][1]; m->s1 = d + o_conv[0][2]; m->t1 = d + o_conv[0][3];
    m->w2 = d + o_conv[1][0]; m->b2 = d + o_conv[1][1]; m->s2 = d + o_conv[1][2]; m->t2 = d + o_conv[1][3];
    m->w3 = d + o_conv[2][0]; m->b3 = d + o_conv[2][1]; m->s3 = d + o_conv[2][2]; m->t3 = d + o_conv[2][3];
    m->fl_wb = d + o_flwb;
    m->cv_wb = d + o_cvwb;
    m->fl_w = d + o_flw; m->fl_b = d + o_flb; m->fp_w = d + o_fpw; m->fp_b = d + o_fpb;
    for (int l = 0; l < 2; ++l) { m->gx_w[l] = d + o_gx[l]; m->tl_w[l] = d + o_tl[l]; m->tl_wb[l] = d + o_tlb[l]; m->tl_w16[l] = d + o_tl16[l]; }
    m->kp_w = d + o_kp; m->qc_w = d + o_qc; m->qp_w = d + o_qp; m->at_v = d + o_v; m->at_b = d + o_b;
    auto bind = [&](sdfa_model::Fc &fc, size_t o[3]) {
        fc.w = d + o[0]; fc.b = d + o[1]; fc.cw = o[2] == (size_t)-1 ? nullptr : d + o[2];
    };
    if (m->head == SDFA_HEAD_DGRAD) {
        bind(m->trunk, o_fc[0]);
        for (int b = 0; b < 2; ++b)
            for (int i = 0; i < 3; ++i) bind(m->br[b][i], o_fc[1 + 3 * b + i]);
    } else {
        for (int i = 0; i < 3; ++i) bind(m->off[i], o_fc[i]);
    }
    for (int b = 0; b < m->pca_n; ++b) { m->pca_q[b] = d + o_pq[b]; m->pca_bias[b] = d + o_pb[b]; }
    if (have_pqb) m->pca_qb = d + o_pqb;
    m->host.clear();
    m->finalized = true;
    return SDFA_OK;
}

}  // extern "C"

// ================================================================================================
// workspace layout (floats), per chunk of Nc frames, Mc = 64*Nc columns
// ================================================================================================
namespace {

constexpr int64_t CT_WORDS = 2048, CT_FLAGS = 16, ST_WORDS = SDFA_WS_STATUS_BYTES / 4;

struct Ws {
    int64_t P1, X3, HF, Z, GX, H0, H1, KP, QC, QP, ZK, R, SH, ZU, CT, total;   // offsets in floats
};

Ws layout(int64_t Nc, bool keep) {
    const int64_t Mc = 64 * Nc;
    Ws w{};
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += round_up(n, 64); return r; };
    take(ST_WORDS);       // the status block (sdfa_workspace_init / sdfa_workspace_status*): always the first 256 bytes, whatever Nc
    if (keep) {   // debug: nothing aliased, taps stay valid after the forward
        w.P1 = take(2048 * Mc); w.X3 = take(2048 * Mc); w.HF = take((int64_t)HF_SLAB_ROWS * 4 * Mc); w.Z = take(256 * Mc);
        w.GX = take(2048 * Mc); w.H0 = take(512 * Mc); w.H1 = take(512 * Mc); w.KP = take(128 * Mc);
    } else {
        const int64_t A = take(2048 * Mc);   // P1, later GX
        const int64_t B = take(2048 * Mc);   // X3, later Z | H0 | H1 | KP
        w.HF = take((int64_t)HF_SLAB_ROWS * 4 * Mc);
        w.P1 = A; w.GX = A;
        w.X3 = B; w.Z = B; w.H0 = B + 256 * Mc; w.H1 = B + 768 * Mc; w.KP = B + 1280 * Mc;
    }
    w.QC = take(512 * Nc); w.QP = take(128 * Nc); w.ZK = take(512 * Nc);
    w.R = take(2560 * Nc);   // regressor scratch: trunk 512 | a 512 | b 256 | coef 288 (+ second branch a/b)
    w.SH = take(2 * Nc + 5 * Mc + Mc / 1024 + 128);   // column-sharing tables (int32 / int64 counters)
    w.ZU = keep ? take(256 * Mc) : w.P1; // freq-proj output over distinct columns (pool1 is dead by then)
    w.CT = take(CT_WORDS);               // ints: [0] / [1] work-queue heads of the persistent kernels; [16 ..] the flag block of
                                         // time_lstm_split_kernel (timeout word + one flag per workgroup, at most 4 per CU-sized grid)
    w.total = o;
    return w;
}

int64_t capacity(int64_t bytes, bool keep) {   // largest Nc (multiple of 128) whose layout fits
    const int64_t per128 = layout(128, keep).total * 4;
    int64_t nc = (bytes / per128 + 2) * 128;   // per-frame cost shrinks slightly with Nc (fixed paddings): start above
    while (nc > 0 && layout(nc, keep).total * 4 > bytes) nc -= 128;
    return nc;
}

struct Prof {
    const sdfa_model *m;
    hipStream_t s;
    hipEvent_t pending = nullptr;
    void begin(const char *stage) {
        if (!m->profile) return;
        sdfa_model::Ev e; e.stage = stage;
        (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b);
        (void)hipEventRecord(e.a, s);
        pending = e.b;
        std::lock_guard<std::mutex> lk(m->ev_mu);
        m->events.push_back(e);
    }
    void end() {
        if (!m->profile || !pending) return;
        (void)hipEventRecord(pending, s);
        pending = nullptr;
    }
};

GemmArgs gemm_fc(const sdfa_model::Fc &fc, const float *Q, int64_t ldq, float *D, int64_t Nc, const int64_t *spk, int64_t nreal) {
    GemmArgs g{};
    g.P = fc.w; g.Q = Q; g.D = D; g.bias = fc.b; g.cond_w = fc.cw; g.cond_idx = fc.cw ? spk : nullptr;
    g.ldp = fc.Ppad; g.ldq = ldq; g.ldd = Nc; g.Ppad = fc.Ppad; g.Qpad = Nc; g.Pstore = fc.Pstore; g.Qreal = nreal;
    g.K = fc.K; g.seg_k = fc.K; g.seg_col = 0; g.act = fc.act; g.out_mode = OUT_K4; g.bias_on_q = 0;
    return g;
}

}  // namespace

extern "C" {

int64_t sdfa_workspace_bytes(const sdfa_model *m, int64_t max_frames) {
    if (!m || max_frames <= 0) return fail(SDFA_EINVAL, "workspace_bytes: bad argument");
    return layout(round_up(max_frames, 128), m->keep).total * 4;
}

// A/B tuning switches: THREAD-LOCAL, so that a thread that flips one for an experiment cannot change what concurrent
// callers on other threads launch (the header promises thread-safe concurrent use of the forward calls).  The table is
// documented next to sdfa_debug_set_option in include/sdfa_hip.h.
extern thread_local int g_sdfa_gemm_variant;
thread_local int g_sdfa_freq_lstm_shape = 0;
thread_local int g_sdfa_pca_unfused = 0;
thread_local int g_sdfa_conv_unfused = 0;
thread_local int g_sdfa_pca_lds = 0;
thread_local int g_sdfa_pca_fp32 = 0;     // "pca_fp32": 1 = the dgrad PCA expansion stays on the fp32 kernel in SDFA_PREC_BF16X3 (A/B)
thread_local int g_sdfa_conv_fp32 = 0;    // "conv_fp32": 1 = the conv stack stays on the fp32 kernel in the mixed-precision modes (A/B)
thread_local int g_sdfa_time_lstm_split = 0;
thread_local int g_sdfa_time_lstm_handoff = 0;
thread_local int g_sdfa_time_lstm_timeout_us = 0;
thread_local int g_sdfa_share_gx0_off = 0;
thread_local int g_sdfa_attn_unfused = 0;  // "attn_unfused": 1 = the bf16 attention modes keep the three-GEMM + attn_kernel form of round 5 (A/B)
int sdfa_debug_set_option(const char *name, int value) {
    if (name && !strcmp(name, "attn_unfused")) { g_sdfa_attn_unfused = value; return SDFA_OK; }
    if (name && !strcmp(name, "share_gx0_off")) { g_sdfa_share_gx0_off = value; return SDFA_OK; }
    if (name && !strcmp(name, "frontend_two_kernel")) { g_sdfa_frontend_two_kernel = value; return SDFA_OK; }
    if (name && !strcmp(name, "frontend_stream_phases")) { g_sdfa_frontend_stream_phases = value; return SDFA_OK; }
    if (name && !strcmp(name, "frontend_stream_spin_max")) {
        if (value < 0) return fail(SDFA_EINVAL, "frontend_stream_spin_max: 0 (default) or a positive poll count");
        g_sdfa_frontend_stream_spin_max = value; return SDFA_OK;
    }
    if (name && !strcmp(name, "frontend_stream_block")) {
        if (value < 0 || value > 256) return fail(SDFA_EINVAL, "frontend_stream_block: 0 (default) or 1..256 frames");
        g_sdfa_frontend_stream_block = value; return SDFA_OK;
    }
    if (name && !strcmp(name, "frontend_stream_slots")) {
        if (value < 0 || value > 256) return fail(SDFA_EINVAL, "frontend_stream_slots: 0 (default) or 1..256 workgroups per block");
        g_sdfa_frontend_stream_slots = value; return SDFA_OK;
    }
    if (name && !strcmp(name, "gather_plain_order")) { g_sdfa_gather_plain_order = value; return SDFA_OK; }
    if (name && !strcmp(name, "mel_fft_radix4")) { g_sdfa_mel_fft_radix4 = value; return SDFA_OK; }
    if (name && !strcmp(name, "frontend_t_major")) { g_sdfa_frontend_t_major = value; return SDFA_OK; }
    if (name && !strcmp(name, "time_lstm_timeout_us")) { g_sdfa_time_lstm_timeout_us = value; return SDFA_OK; }
    if (name && !strcmp(name, "time_lstm_handoff")) { g_sdfa_time_lstm_handoff = value; return SDFA_OK; }
    if (name && !strcmp(name, "time_lstm_split")) { g_sdfa_time_lstm_split = value; return SDFA_OK; }
    if (name && !strcmp(name, "gemm_variant")) { g_sdfa_gemm_variant = value; return SDFA_OK; }
    if (name && !strcmp(name, "freq_lstm_shape")) { g_sdfa_freq_lstm_shape = value; return SDFA_OK; }
    if (name && !strcmp(name, "pca_unfused")) { g_sdfa_pca_unfused = value; return SDFA_OK; }
    if (name && !strcmp(name, "conv_unfused")) { g_sdfa_conv_unfused = value; return SDFA_OK; }
    if (name && !strcmp(name, "pca_lds")) { g_sdfa_pca_lds = value; return SDFA_OK; }
    if (name && !strcmp(name, "conv_fp32")) { g_sdfa_conv_fp32 = value; return SDFA_OK; }
    if (name && !strcmp(name, "pca_fp32")) { g_sdfa_pca_fp32 = value; return SDFA_OK; }
    return fail(SDFA_EINVAL, "unknown option '%s'", name ? name : "(null)");
}

int sdfa_debug_keep_intermediates(sdfa_model *m, int on) {
    if (!m) return fail(SDFA_EINVAL, "null model");
    m->keep = on != 0;
    return SDFA_OK;
}

int sdfa_profile_enable(sdfa_model *m, int on) {
    if (!m) return fail(SDFA_EINVAL, "null model");
    std::lock_guard<std::mutex> lk(m->ev_mu);
    m->profile = on != 0;
    for (auto &e : m->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    m->events.clear();
    return SDFA_OK;
}

float sdfa_profile_ms(const sdfa_model *m, const char *stage) {
    if (!m || !stage) return (float)fail(SDFA_EINVAL, "profile_ms: bad argument");
    float total = 0.f;
    bool any = false;
    std::lock_guard<std::mutex> lk(m->ev_mu);
    for (auto &e : m->events)
        if (e.stage == stage) {
            float ms = 0.f;
            if (hipEventSynchronize(e.b) != hipSuccess || hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess)
                return (float)fail(SDFA_EHIP, "profile_ms: event query failed");
            total += ms; any = true;
        }
    return any ? total : (float)fail(SDFA_EINVAL, "profile_ms: no such stage '%s'", stage);
}

int sdfa_profile_reset(sdfa_model *m) { return m ? sdfa_profile_enable(m, m->profile) : fail(SDFA_EINVAL, "null model"); }

// ------------------------------------------------------------------------------------------------
static int encoder_impl(const sdfa_model *m, const float *d_audio_feat, int64_t n_frames, const int32_t *d_frame_clip,
                        const int64_t *d_frame_start, int hop, float *d_z, float *d_align, void *d_workspace,
                        int64_t workspace_bytes, void *stream);

// Mixed-precision modes (BASELINE configs[3]).  Which MFMA each stage runs on: 0 = fp32, 1 = bf16 operands,
// 3 = split-bf16 (hi/lo operands, three MFMAs per product).  The conv stack, the fused dgrad PCA expansion, softmax/context
// and every accumulation, cell state, bias and activation stay fp32 in all modes.
enum { STAGE_BODY = 0, STAGE_ATTENTION = 1, STAGE_REGRESSOR = 2 };
static int stage_terms(const sdfa_model *m, int stage) {
    switch (m->precision) {
    case SDFA_PREC_BF16_ATTENTION: return stage == STAGE_ATTENTION ? 1 : 0;
    case SDFA_PREC_BF16X3_ATTENTION: return stage == STAGE_ATTENTION ? 3 : 0;
    case SDFA_PREC_BF16X6: return 6;
    case SDFA_PREC_BF16X3: return 3;
    case SDFA_PREC_BF16: return 1;
    default: return 0;
    }
}

int sdfa_model_set_precision(sdfa_model *m, int mode) {
    if (!m) return fail(SDFA_EINVAL, "null model");
    if (mode < SDFA_PREC_FP32 || mode > SDFA_PREC_BF16X6) return fail(SDFA_EINVAL, "unknown precision mode %d", mode);
    m->precision = mode;
    return SDFA_OK;
}

int sdfa_model_precision(const sdfa_model *m) { return m ? m->precision : SDFA_EINVAL; }

// The persistent / one-workgroup-per-CU kernels (frequency LSTM, the fat GEMMs, the PCA expansion) size their grids to the
// device's CU count; with k reserved they launch (CUs - k) workgroups, so that kernels of another stream (an RCCL
// all-gather, a copy kernel) find free CUs while they run.
int sdfa_model_set_reserved_cus(sdfa_model *m, int k) {
    if (!m) return fail(SDFA_EINVAL, "null model");
    if (k < 0 || k > 128) return fail(SDFA_EINVAL, "reserved_cus must be in [0, 128], got %d", k);
    m->reserved_cus.store(k);
    return SDFA_OK;
}

// The kernels / launch forms of the fp32 frequency LSTM are bit-identical and within 1-3 % of each other; for the forms with
// two resident workgroups per CU the order is a property of how the two happen to interleave:
// measure, don't guess.
int sdfa_model_autotune(sdfa_model *m, int64_t n_frames, void *d_workspace, int64_t workspace_bytes, void *stream) {
    if (!m || !m->finalized) return fail(SDFA_ESTATE, "autotune: model not finalised");
    if (!d_workspace || n_frames <= 0 || ((uintptr_t)d_workspace & 15)) return fail(SDFA_EINVAL, "autotune: bad argument");
    const int64_t cap = capacity(workspace_bytes, m->keep);
    if (cap < 128) return fail(SDFA_ENOSPACE, "autotune: workspace too small");
    const int64_t Nc = round_up(std::min(cap, n_frames), 128), Mc = 64 * Nc;
    const Ws w = layout(Nc, m->keep);
    float *ws = (float *)d_workspace;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(ws + w.X3, 0, (size_t)2048 * Mc * sizeof(float), s));      // any finite input: the kernel's time does not depend on the data
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    static const int forms[] = {9, 8, 5, 3};      // third kernel persistent / hardware-dispatched; second kernel persistent / hardware-dispatched (two per CU)
    float best_ms = 0.f;
    int best = m->freq_shape.load(), rc = SDFA_OK;
    for (int form : forms) {
        FreqLstmArgs fa{ws + w.X3, m->fl_w, m->fl_b, ws + w.HF, Mc, nullptr, m->fl_wb, 0, reinterpret_cast<int *>(ws + w.CT), form, m->reserved_cus.load()};
        float ms = 0.f;
        for (int rep = 0; rep < 3 && rc == SDFA_OK; ++rep) {      // one warm launch, two timed
            if (rep == 1 && hipEventRecord(e0, s) != hipSuccess) rc = SDFA_EHIP;
            if (sdfa_launch_freq_lstm(fa, s) != hipSuccess) rc = SDFA_EHIP;
        }
        if (rc == SDFA_OK && (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)) rc = SDFA_EHIP;
        if (rc != SDFA_OK) break;
        if (best_ms == 0.f || ms < best_ms) { best_ms = ms; best = form; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != SDFA_OK) return fail(rc, "autotune: a HIP call failed: %s", hipGetErrorString(hipGetLastError()));
    m->freq_shape.store(best);
    return best;
}

int sdfa_encoder_forward(const sdfa_model *m, const float *d_audio_feat, int64_t n_frames, float *d_z, float *d_align,
                         void *d_workspace, int64_t workspace_bytes, void *stream) {
    return encoder_impl(m, d_audio_feat, n_frames, nullptr, nullptr, 0, d_z, d_align, d_workspace, workspace_bytes, stream);
}

int sdfa_encoder_forward_shared(const sdfa_model *m, const float *d_audio_feat, int64_t n_frames, const int32_t *d_frame_clip,
                                const int64_t *d_frame_start, int hop, float *d_z, float *d_align, void *d_workspace,
                                int64_t workspace_bytes, void *stream) {
    if (!d_frame_clip || !d_frame_start || hop <= 0) return fail(SDFA_EINVAL, "encoder_forward_shared: frame table missing");
    return encoder_impl(m, d_audio_feat, n_frames, d_frame_clip, d_frame_start, hop, d_z, d_align, d_workspace, workspace_bytes, stream);
}

static int encoder_impl(const sdfa_model *m, const float *d_audio_feat, int64_t n_frames, const int32_t *d_frame_clip,
                        const int64_t *d_frame_start, int hop, float *d_z, float *d_align, void *d_workspace,
                        int64_t workspace_bytes, void *stream) {
    if (!m || !m->finalized) return fail(SDFA_ESTATE, "encoder_forward: model not finalised");
    if (n_frames == 0) return SDFA_OK;
    if (!d_audio_feat || !d_z || !d_workspace || n_frames < 0) return fail(SDFA_EINVAL, "encoder_forward: bad argument");
    if (((uintptr_t)d_workspace | (uintptr_t)d_audio_feat | (uintptr_t)d_z) & 15) return fail(SDFA_EINVAL, "encoder_forward: pointers must be 16-byte aligned");
    if ((uintptr_t)d_align & 3) return fail(SDFA_EINVAL, "encoder_forward: d_align must be 4-byte aligned");
    const int64_t cap = capacity(workspace_bytes, m->keep);
    if (cap < 128) return fail(SDFA_ENOSPACE, "encoder_forward: workspace of %lld bytes holds no 128-frame chunk", (long long)workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    Prof pf{m, s};
    float *ws = (float *)d_workspace;
    for (int64_t f0 = 0; f0 < n_frames; f0 += cap) {
        const int64_t N = std::min(cap, n_frames - f0);
        const int64_t Nc = round_up(N, 128), Mc = 64 * Nc;
        const Ws w = layout(Nc, m->keep);
        ConvArgs ca{};
        ca.audio_feat = d_audio_feat + f0 * (64 * 128 * 3); ca.N = N; ca.Nc = Nc; ca.Mc = Mc;
        ca.w1 = m->w1; ca.b1 = m->b1; ca.s1 = m->s1; ca.t1 = m->t1; ca.P1 = ws + w.P1;
        ca.w2 = m->w2; ca.b2 = m->b2; ca.s2 = m->s2; ca.t2 = m->t2;
        ca.w3 = m->w3; ca.b3 = m->b3; ca.s3 = m->s3; ca.t3 = m->t3; ca.X3 = ws + w.X3;
        const bool share = d_frame_clip != nullptr;
        const int64_t *d_ulimit = nullptr;     // device scalar: number of distinct columns, padded to 256
        int32_t *col_to_u = nullptr;
        if (share) {
            int32_t *sh = reinterpret_cast<int32_t *>(ws + w.SH);
            ShareArgs sa{};
            sa.frame_clip = d_frame_clip + f0; sa.frame_start = d_frame_start + f0; sa.hop = hop;
            sa.t_lo = 6; sa.t_hi = 58;      // feature columns: see share.hip
            sa.N = N; sa.Nc = Nc; sa.Mc = Mc;
            sa.counts = reinterpret_cast<int64_t *>(sh);           // 16 ints reserved
            sa.prev = sh + 16; sa.shift = sa.prev + Nc;
            sa.owner = sa.shift + Nc; sa.flag = sa.owner + Mc; sa.uid = sa.flag + Mc;
            sa.col_src = sa.uid + Mc; sa.col_to_u = sa.col_src + Mc; sa.tile_sum = sa.col_to_u + Mc;
            pf.begin("share_map"); HIP_TRY(sdfa_launch_share_map(sa, s)); pf.end();
            d_ulimit = sa.counts + 1; col_to_u = sa.col_to_u;
            ca.col_src = sa.col_src; ca.col_limit = d_ulimit;
        }
        const int conv_terms = g_sdfa_conv_fp32 ? 0 : stage_terms(m, STAGE_BODY);
        if (m->keep && conv_terms && !g_sdfa_conv_unfused) {
            // debug taps in a body precision mode: the X3 tap must be the arithmetic that ships -- conv123_bf16_kernel -- so a
            // packing or K-order bug in pack_conv_bf16 can be localised; the pool1 tap (which the fused kernel never writes) stays fp32
            pf.begin("conv1"); HIP_TRY(sdfa_launch_conv1(ca, s)); pf.end();
            ca.wb = m->cv_wb; ca.terms = conv_terms;
            pf.begin("conv23"); HIP_TRY(sdfa_launch_conv123(ca, s)); pf.end();
        } else if (m->keep || g_sdfa_conv_unfused) {   // the debug taps read pool1
            pf.begin("conv1"); HIP_TRY(sdfa_launch_conv1(ca, s)); pf.end();
            pf.begin("conv23"); HIP_TRY(sdfa_launch_conv23(ca, s)); pf.end();
        } else {
            ca.wb = m->cv_wb; ca.terms = conv_terms;      // the mixed-precision modes run the stack on bf16 MFMA too
            pf.begin("conv23"); HIP_TRY(sdfa_launch_conv123(ca, s)); pf.end();
        }

        // launch form: the "freq_lstm_shape" option if set; else, for the fp32 kernel, what sdfa_model_autotune measured (the split-bf16
        // kernels have their own default: the autotuned fp32 form says nothing about them)
        const int fl_terms = stage_terms(m, STAGE_BODY);
        FreqLstmArgs fa{ws + w.X3, m->fl_w, m->fl_b, ws + w.HF, Mc, d_ulimit, m->fl_wb, fl_terms, reinterpret_cast<int *>(ws + w.CT),
                        g_sdfa_freq_lstm_shape ? g_sdfa_freq_lstm_shape : (fl_terms ? 0 : m->freq_shape.load()), m->reserved_cus.load()};
        pf.begin("freq_lstm"); HIP_TRY(sdfa_launch_freq_lstm(fa, s)); pf.end();

        GemmArgs g{};   // FreqLstm._proj: Linear(8192 -> 256) + bias
        g.P = m->fp_w; g.Q = ws + w.HF; g.D = ws + w.Z; g.bias = m->fp_b;
        g.ldp = 256; g.ldq = Mc; g.ldd = Mc; g.Ppad = 256; g.Qpad = Mc; g.Pstore = 256; g.Qreal = Mc;
        g.K = 8192; g.seg_k = 8192; g.act = ACT_NONE; g.out_mode = OUT_K4; g.q_tile_major = 1; g.q_slab_rows = HF_SLAB_ROWS;
        g.terms = stage_terms(m, STAGE_BODY);
        g.reserve_cus = m->reserved_cus.load();
        if (share) { g.D = ws + w.ZU; g.q_limit = d_ulimit; }
        pf.begin("freq_proj"); HIP_TRY(sdfa_launch_gemm(g, s)); pf.end();
        // Column sharing reaches one stage further (round 4): the layer-0 input projection of the BiLSTM (rnn.py:20-21) is per column
        // too, so it runs over the DISTINCT columns and the layer-0 recurrence reads it through the share map -- the 256-feature
        // projection is then never expanded at all.  (fp32 kernels; the bf16 recurrences and the debug taps, which read the
        // expanded projection, keep the expand-then-project order.)  The projected columns live in the frequency LSTM's hidden-state
        // region, which is dead once the projection above has read it.
        const bool share_gx0 = share && !m->keep && stage_terms(m, STAGE_BODY) == 0 && !g_sdfa_share_gx0_off;
        if (share && !share_gx0) {   // scatter every distinct column's 256 features to all the (t, n) columns that contain it
            pf.begin("share_expand"); HIP_TRY(sdfa_launch_expand_cols(ws + w.ZU, col_to_u, ws + w.Z, 64, Mc, s)); pf.end();
        }

        const float *xin = share_gx0 ? ws + w.ZU : ws + w.Z;
        float *hout[2] = {ws + w.H0, ws + w.H1};
        const char *gxn[2] = {"gx0", "gx1"}, *lsn[2] = {"lstm0", "lstm1"};
        for (int l = 0; l < 2; ++l) {
            const bool mapped = share_gx0 && l == 0;
            float *gx = mapped ? ws + w.HF : ws + w.GX;
            GemmArgs gi{};
            gi.P = m->gx_w[l]; gi.Q = xin; gi.D = gx;
            gi.ldp = 2048; gi.ldq = Mc; gi.ldd = Mc; gi.Ppad = 2048; gi.Qpad = Mc; gi.Pstore = 2048; gi.Qreal = Mc;
            gi.K = l == 0 ? 256 : 512; gi.seg_k = gi.K; gi.act = ACT_NONE; gi.out_mode = OUT_K4;
            gi.terms = stage_terms(m, STAGE_BODY);
            gi.reserve_cus = m->reserved_cus.load();
            if (mapped) gi.q_limit = d_ulimit;
            pf.begin(gxn[l]); HIP_TRY(sdfa_launch_gemm(gi, s)); pf.end();
            TimeLstmArgs ta{gx, m->tl_w[l], hout[l], Nc, Mc, m->tl_wb[l], stage_terms(m, STAGE_BODY),
                            reinterpret_cast<unsigned *>(ws + w.CT) + CT_FLAGS, CT_WORDS - CT_FLAGS, m->tl_w16[l],
                            reinterpret_cast<unsigned *>(ws), m->reserved_cus.load(), mapped ? col_to_u : nullptr};
            pf.begin(lsn[l]); HIP_TRY(sdfa_launch_time_lstm(ta, s)); pf.end();
            xin = hout[l];
        }
        // attention: proj_key over all 64 keys, Conv1d query over time steps 31..33, proj_qry
        const int at_terms = stage_terms(m, STAGE_ATTENTION);
        // bf16 attention modes (BASELINE configs[3]): the key projection never leaves the matrix core's accumulators -- one HBM-bound
        // streaming pass over the BiLSTM output computes projection + tanh + v-dot and writes one score per column (attn.hip:
        // attn_key_score_kernel); attn_kernel then does softmax + context from the scores.  The query path runs first.
        // fp32 (round 6, second half): the same pass on v_mfma_f32_16x16x4_f32 -- matrix-bound there, at the MFMA rate.  The six-product
        // mode (fp32-equivalent products) takes the exact fp32 pass too: faster than six bf16 products through a GEMM, and exact.
        const bool key_fused = g_sdfa_attn_unfused != 1;
        pf.begin("attn_proj");
        GemmArgs gk{};
        gk.P = m->kp_w; gk.Q = ws + w.H1; gk.D = ws + w.KP;
        gk.ldp = 128; gk.ldq = Mc; gk.ldd = Mc; gk.Ppad = 128; gk.Qpad = Mc; gk.Pstore = 128; gk.Qreal = Mc;
        gk.K = 512; gk.seg_k = 512; gk.act = ACT_NONE; gk.out_mode = OUT_K4; gk.terms = at_terms;
        if (!key_fused) HIP_TRY(sdfa_launch_gemm(gk, s));
        GemmArgs gc{};
        gc.P = m->qc_w; gc.Q = ws + w.H1 + 31 * Nc * 4; gc.D = ws + w.QC;
        gc.ldp = 512; gc.ldq = Mc; gc.ldd = Nc; gc.Ppad = 512; gc.Qpad = Nc; gc.Pstore = 512; gc.Qreal = Nc;
        gc.K = 1536; gc.seg_k = 512; gc.seg_col = Nc; gc.act = ACT_NONE; gc.out_mode = OUT_K4; gc.terms = gk.terms;
        HIP_TRY(sdfa_launch_gemm(gc, s));
        GemmArgs gq{};
        gq.P = m->qp_w; gq.Q = ws + w.QC; gq.D = ws + w.QP;
        gq.ldp = 128; gq.ldq = Nc; gq.ldd = Nc; gq.Ppad = 128; gq.Qpad = Nc; gq.Pstore = 128; gq.Qreal = Nc;
        gq.K = 512; gq.seg_k = 512; gq.act = ACT_NONE; gq.out_mode = OUT_K4; gq.terms = gk.terms;
        HIP_TRY(sdfa_launch_gemm(gq, s));
        // exact fp32, large chunks: the whole layer in ONE pass over H (running softmax + context while the tile is in LDS: attn_fused_f32_kernel);
        // "attn_unfused" = 2 keeps the two-kernel form
        const bool tail_fused = key_fused && (at_terms == 0 || at_terms == 6) && g_sdfa_attn_unfused != 2 && sdfa_attn_fuses_tail(Nc, m->reserved_cus.load());
        if (key_fused) {
            AttnKeyArgs ak{};
            ak.Wk = m->kp_w; ak.H = ws + w.H1; ak.QP = ws + w.QP; ak.v = m->at_v; ak.b = m->at_b;
            ak.S = ws + w.KP;                                  // the partial scores take the first 8 Mc floats of the (unused) key-projection region
            ak.Nc = Nc; ak.Mc = Mc; ak.terms = at_terms == 6 ? 0 : at_terms; ak.reserve_cus = m->reserved_cus.load();
            if (tail_fused) {
                ak.fuse_tail = 1; ak.Zk4 = ws + w.ZK; ak.z_out = d_z + f0 * 512; ak.align_out = d_align ? d_align + f0 * 64 : nullptr; ak.N = N;
            }
            HIP_TRY(sdfa_launch_attn_key_score(ak, s));
        }
        pf.end();

        AttnArgs aa{};
        aa.KP = ws + w.KP; aa.QP = ws + w.QP; aa.H = ws + w.H1; aa.v = m->at_v; aa.b = m->at_b;
        aa.Zk4 = ws + w.ZK; aa.z_out = d_z + f0 * 512; aa.align_out = d_align ? d_align + f0 * 64 : nullptr;
        aa.N = N; aa.Nc = Nc; aa.Mc = Mc;
        aa.S = key_fused ? ws + w.KP : nullptr;
        pf.begin("attn"); if (!tail_fused) HIP_TRY(sdfa_launch_attn(aa, s)); pf.end();
    }
    return SDFA_OK;
}

// ------------------------------------------------------------------------------------------------
// PCA expansion of N frames' coefficients (K4, 288 / 64 rows): out[n][o] = sum_k coef[k][n] * basis[k][o] + means[o]
// (rows = frames), stored to every destination.  PcaInversion.forward, speech_anime/modules/output_module.py:94-116
static hipError_t expand_rows(const sdfa_model *m, const float *coef, int64_t N, int64_t Nc, int64_t f0,
                              float *const *h_d_outs, int n_outs, int *queue, hipStream_t s) {
    float *d_out = h_d_outs[0];
    if (m->pca_n == 2 && !g_sdfa_pca_unfused) {
        // dgrad head: both bases in one fp32 kernel (all precision modes) so that every output line is written
        // once, whole (pca.hip)
        PcaArgs pa{};
        pa.coef = coef; pa.basis_s = m->pca_q[0]; pa.basis_r = m->pca_q[1]; pa.mean_s = m->pca_bias[0]; pa.mean_r = m->pca_bias[1];
        pa.out = d_out + f0 * m->out_dim; pa.N = N; pa.Nc = Nc; pa.out_dim = m->out_dim;
        pa.n_extra = n_outs - 1;
        pa.reserve_cus = m->reserved_cus.load();
        // SDFA_PREC_BF16X3: the expansion on split-bf16 MFMA too (round 5; "pca_fp32" = 1 keeps it exact); every other mode: fp32
        pa.basis_b = m->pca_qb; pa.terms = (g_sdfa_pca_fp32 || m->precision != SDFA_PREC_BF16X3) ? 0 : 3;
        for (int x = 1; x < n_outs; ++x) pa.out_extra[x - 1] = h_d_outs[x] + f0 * m->out_dim;
        pa.ld_s = m->pca_ld[0]; pa.ld_r = m->pca_ld[1]; pa.cols_s = m->pca_cols[0]; pa.cols_r = m->pca_cols[1];
        // default: basis slab resident in LDS, persistent work units (pca_dgrad_res_kernel); "pca_lds" option 4 = the register-
        // direct two-workgroups-per-CU form of rounds 1-2 (bit-identical, slower: the fallback that shares a CU)
        if (g_sdfa_pca_lds != 4 && queue) return sdfa_launch_pca_dgrad_res(pa, queue, s);
        return sdfa_launch_pca_dgrad(pa, s);
    }
    for (int b = 0; b < m->pca_n; ++b) {
        GemmArgs g{};
        g.P = coef + (int64_t)m->pca_k0[b] * Nc; g.Q = m->pca_q[b]; g.D = d_out + f0 * m->out_dim;
        g.bias = m->pca_bias[b]; g.bias_on_q = 1;
        g.ldp = Nc; g.ldq = m->pca_ld[b]; g.ldd = m->out_dim; g.Ppad = Nc; g.Qpad = m->pca_ld[b]; g.Pstore = N;
        g.Qreal = m->pca_cols[b]; g.K = m->pca_K[b]; g.seg_k = g.K; g.act = ACT_NONE; g.out_mode = OUT_ROW;
        g.col_group = m->pca_group[b]; g.col_stride = 9; g.col_off = m->pca_off[b];
        g.n_extra = n_outs - 1;
        for (int x = 1; x < n_outs; ++x) g.D_extra[x - 1] = h_d_outs[x] + f0 * m->out_dim;
        g.terms = stage_terms(m, STAGE_REGRESSOR);
        hipError_t e = sdfa_launch_gemm(g, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

int sdfa_regress_forward(const sdfa_model *m, const float *d_z, const int64_t *d_speaker_id, int64_t n_frames,
                         float *d_coef, float *d_out, void *d_workspace, int64_t workspace_bytes, void *stream) {
    float *one[1] = {d_out};
    return sdfa_regress_forward_multi(m, d_z, d_speaker_id, n_frames, d_coef, one, d_out ? 1 : 0, d_workspace, workspace_bytes, stream);
}

int sdfa_regress_forward_multi(const sdfa_model *m, const float *d_z, const int64_t *d_speaker_id, int64_t n_frames,
                               float *d_coef, float *const *h_d_outs, int n_outs, void *d_workspace, int64_t workspace_bytes,
                               void *stream) {
    if (!m || !m->finalized) return fail(SDFA_ESTATE, "regress_forward: model not finalised");
    if (n_outs < 0 || n_outs > SDFA_MAX_DESTS || (n_outs && !h_d_outs)) return fail(SDFA_EINVAL, "regress_forward: 0..%d output destinations", SDFA_MAX_DESTS);
    for (int i = 0; i < n_outs; ++i) {
        if (!h_d_outs[i]) return fail(SDFA_EINVAL, "regress_forward: output destination %d is null", i);
        if (m->head == SDFA_HEAD_DGRAD && ((uintptr_t)h_d_outs[i] & 15)) return fail(SDFA_EINVAL, "regress_forward: d_out must be 16-byte aligned for the dgrad head");
        if ((uintptr_t)h_d_outs[i] & 3) return fail(SDFA_EINVAL, "regress_forward: output pointers must be 4-byte aligned");
    }
    float *d_out = n_outs ? h_d_outs[0] : nullptr;
    if (n_frames == 0) return SDFA_OK;
    if (!d_z || !d_speaker_id || !d_workspace || n_frames < 0) return fail(SDFA_EINVAL, "regress_forward: bad argument");
    if (((uintptr_t)d_workspace | (uintptr_t)d_z) & 15) return fail(SDFA_EINVAL, "regress_forward: pointers must be 16-byte aligned");
    // the fused dgrad expansion writes whole rows with 16-byte stores (pca.hip); rows are 359,136 B apart, so the base decides
    if (m->head == SDFA_HEAD_DGRAD && d_out && ((uintptr_t)d_out & 15)) return fail(SDFA_EINVAL, "regress_forward: d_out must be 16-byte aligned for the dgrad head");
    if (((uintptr_t)d_out | (uintptr_t)d_coef) & 3) return fail(SDFA_EINVAL, "regress_forward: output pointers must be 4-byte aligned");
    const int64_t cap = capacity(workspace_bytes, m->keep);
    if (cap < 128) return fail(SDFA_ENOSPACE, "regress_forward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Prof pf{m, s};
    float *ws = (float *)d_workspace;
    auto fcg = [m](const sdfa_model::Fc &fc, const float *Q, int64_t ldq, float *D, int64_t Nc, const int64_t *spk, int64_t nreal) {
        GemmArgs g = gemm_fc(fc, Q, ldq, D, Nc, spk, nreal);
        g.terms = stage_terms(m, STAGE_REGRESSOR);
        return g;
    };
    for (int64_t f0 = 0; f0 < n_frames; f0 += cap) {
        const int64_t N = std::min(cap, n_frames - f0);
        const int64_t Nc = round_up(N, 128);
        const Ws w = layout(Nc, m->keep);
        float *zk = ws + w.ZK, *r = ws + w.R;
        float *trunk = r, *a0 = r + 512 * Nc, *a1 = r + 1024 * Nc, *coef = r + 1280 * Nc;   // coef: up to 288 rows
        const int64_t *spk = d_speaker_id + f0;
        pf.begin("mlp");
        HIP_TRY(sdfa_launch_rows_to_k4(d_z + f0 * 512, N, 512, zk, Nc, s));
        if (m->head == SDFA_HEAD_DGRAD) {
            HIP_TRY(sdfa_launch_gemm(fcg(m->trunk, zk, Nc, trunk, Nc, spk, N), s));
            for (int b = 0; b < 2; ++b) {
                HIP_TRY(sdfa_launch_gemm(fcg(m->br[b][0], trunk, Nc, a0, Nc, spk, N), s));
                HIP_TRY(sdfa_launch_gemm(fcg(m->br[b][1], a0, Nc, a1, Nc, spk, N), s));
                HIP_TRY(sdfa_launch_gemm(fcg(m->br[b][2], a1, Nc, coef + (b ? 96 * Nc : 0), Nc, spk, N), s));
            }
            if (d_coef) {
                HIP_TRY(sdfa_launch_k4_to_rows(coef, Nc, N, 288, 0, SDFA_COEF_SCALE, d_coef + f0 * m->coef_dim, m->coef_dim, s));
                HIP_TRY(sdfa_launch_k4_to_rows(coef, Nc, N, 288, 96, SDFA_COEF_ROTAT, d_coef + f0 * m->coef_dim + SDFA_COEF_SCALE, m->coef_dim, s));
            }
        } else {
            HIP_TRY(sdfa_launch_gemm(fcg(m->off[0], zk, Nc, a0, Nc, spk, N), s));
            HIP_TRY(sdfa_launch_gemm(fcg(m->off[1], a0, Nc, a1, Nc, spk, N), s));
            HIP_TRY(sdfa_launch_gemm(fcg(m->off[2], a1, Nc, coef, Nc, spk, N), s));
            if (d_coef) HIP_TRY(sdfa_launch_k4_to_rows(coef, Nc, N, 64, 0, SDFA_COEF_OFFSETS, d_coef + f0 * m->coef_dim, m->coef_dim, s));
        }
        pf.end();
        if (d_out) {
            pf.begin("pca");
            HIP_TRY(expand_rows(m, coef, N, Nc, f0, h_d_outs, n_outs, reinterpret_cast<int *>(ws + w.CT) + 1, s));
            pf.end();
        }
    }
    return SDFA_OK;
}

int sdfa_expand_coef(const sdfa_model *m, const float *d_coef, int64_t n_frames, float *d_out, void *d_workspace,
                     int64_t workspace_bytes, void *stream) {
    if (!m || !m->finalized) return fail(SDFA_ESTATE, "expand_coef: model not finalised");
    if (n_frames == 0) return SDFA_OK;
    if (!d_coef || !d_out || !d_workspace || n_frames < 0) return fail(SDFA_EINVAL, "expand_coef: bad argument");
    if ((uintptr_t)d_workspace & 15) return fail(SDFA_EINVAL, "expand_coef: d_workspace must be 16-byte aligned");
    if (m->head == SDFA_HEAD_DGRAD && ((uintptr_t)d_out & 15)) return fail(SDFA_EINVAL, "expand_coef: d_out must be 16-byte aligned for the dgrad head");
    if (((uintptr_t)d_out | (uintptr_t)d_coef) & 3) return fail(SDFA_EINVAL, "expand_coef: pointers must be 4-byte aligned");
    const int64_t cap = capacity(workspace_bytes, m->keep);
    if (cap < 128) return fail(SDFA_ENOSPACE, "expand_coef: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Prof pf{m, s};
    float *ws = (float *)d_workspace;
    for (int64_t f0 = 0; f0 < n_frames; f0 += cap) {
        const int64_t N = std::min(cap, n_frames - f0);
        const int64_t Nc = round_up(N, 128);
        const Ws w = layout(Nc, m->keep);
        float *coef = ws + w.R + 1280 * Nc;                  // the regressor's coefficient slab: same place, same layout
        const float *src = d_coef + f0 * m->coef_dim;
        pf.begin("pca");
        if (m->head == SDFA_HEAD_DGRAD) {
            HIP_TRY(sdfa_launch_rows_seg_to_k4(src, m->coef_dim, N, 0, SDFA_COEF_SCALE, coef, Nc, 0, 96, s));
            HIP_TRY(sdfa_launch_rows_seg_to_k4(src, m->coef_dim, N, SDFA_COEF_SCALE, SDFA_COEF_ROTAT, coef, Nc, 96, 192, s));
        } else {
            HIP_TRY(sdfa_launch_rows_seg_to_k4(src, m->coef_dim, N, 0, SDFA_COEF_OFFSETS, coef, Nc, 0, 64, s));
        }
        float *outs[1] = {d_out};
        HIP_TRY(expand_rows(m, coef, N, Nc, f0, outs, 1, reinterpret_cast<int *>(ws + w.CT) + 1, s));
        pf.end();
    }
    return SDFA_OK;
}

// ------------------------------------------------------------------------------------------------
int64_t sdfa_debug_distinct_columns(const sdfa_model *m, int64_t n_frames, const void *d_workspace, void *stream) {
    if (!m || n_frames <= 0 || !d_workspace) return fail(SDFA_EINVAL, "debug_distinct_columns: bad argument");
    const Ws w = layout(round_up(n_frames, 128), m->keep);
    int64_t counts[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(counts, (const float *)d_workspace + w.SH, sizeof counts, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return counts[0];
}

// ---- workspace status block -----------------------------------------------------------------------------------------------
int sdfa_workspace_init(void *d_workspace, int64_t workspace_bytes, void *stream) {
    if (!d_workspace || workspace_bytes < SDFA_WS_STATUS_BYTES || ((uintptr_t)d_workspace & 15)) return fail(SDFA_EINVAL, "workspace_init: bad argument");
    HIP_TRY(hipMemsetAsync(d_workspace, 0, SDFA_WS_STATUS_BYTES, (hipStream_t)stream));
    return SDFA_OK;
}

int sdfa_workspace_status_async(const void *d_workspace, uint32_t *h_status, void *stream) {
    if (!d_workspace || !h_status) return fail(SDFA_EINVAL, "workspace_status_async: bad argument");
    HIP_TRY(hipMemcpyAsync(h_status, d_workspace, SDFA_WS_STATUS_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return SDFA_OK;
}

int64_t sdfa_workspace_status(const void *d_workspace, int word, void *stream) {
    if (!d_workspace || word < 0 || word >= SDFA_WS_STATUS_WORDS) return fail(SDFA_EINVAL, "workspace_status: bad argument");
    uint32_t v = 0;
    HIP_TRY(hipMemcpyAsync(&v, reinterpret_cast<const uint32_t *>(d_workspace) + word, sizeof v, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return (int64_t)v;
}

int sdfa_debug_tap(const sdfa_model *m, int what, int64_t n_frames, float *d_dst, const void *d_workspace, void *stream) {
    if (!m || !m->keep) return fail(SDFA_ESTATE, "debug_tap needs sdfa_debug_keep_intermediates(model, 1) before the forward");
    if (what < 0 || what > 3 || n_frames <= 0 || !d_dst || !d_workspace) return fail(SDFA_EINVAL, "debug_tap: bad argument");
    const int64_t Nc = round_up(n_frames, 128);
    const Ws w = layout(Nc, true);
    const float *ws = (const float *)d_workspace;
    const float *src = what == 0 ? ws + w.P1 : what == 1 ? ws + w.X3 : what == 2 ? ws + w.Z : ws + w.H1;
    HIP_TRY(sdfa_launch_tap(src, what, n_frames, Nc, d_dst, (hipStream_t)stream));
    return SDFA_OK;
}

}  // extern "C"

// ================================================================================================
// next row: dgrad -> mesh (deformation transfer solve)
// ================================================================================================
struct sdfa_mesh {
    int n_verts = 0, n_tris = 0, n_free = 0, free_pad = 0;
    int n_src_tris = 0;        // triangles per dgrad row (= n_tris unless triangle correspondences retarget another topology)
    void *blob = nullptr;
    const int *inc_ptr, *inc_tri, *vert_col;
    const float *inc_coef, *tmpl, *inv_k4, *reg_xt;
};

extern "C" {

sdfa_mesh *sdfa_mesh_create(const float *h_verts, int64_t n_verts, const uint32_t *h_faces, int64_t n_tris,
                            const uint32_t *h_cnsts, int64_t n_cnsts, double reg, void *stream) {
    return sdfa_mesh_create_corres(h_verts, n_verts, h_faces, n_tris, h_cnsts, n_cnsts, nullptr, nullptr, 0, n_tris, reg, stream);
}

sdfa_mesh *sdfa_mesh_create_corres(const float *h_verts, int64_t n_verts, const uint32_t *h_faces, int64_t n_tris,
                                   const uint32_t *h_cnsts, int64_t n_cnsts, const uint32_t *h_corr_count,
                                   const uint32_t *h_corr_faces, int64_t n_corr_faces, int64_t n_src_tris, double reg,
                                   void *stream) {
    if (!h_verts || !h_faces || n_verts <= 0 || n_tris <= 0 || n_cnsts < 0 || (n_cnsts && !h_cnsts) || n_src_tris <= 0) {
        fail(SDFA_EINVAL, "mesh_create: bad argument");
        return nullptr;
    }
    // triangle correspondences (deform_triangle_impl.hpp:16-21,102,248-266): target triangle j contributes
    // max(1, corr_count[j]) equations; equation k of a triangle with correspondences takes the transform of SOURCE triangle
    // corr_faces[k] (corr_faces holds one filler entry for a triangle without any, viewer/frame.py:72-80), the others the identity
    if (h_corr_count) {
        int64_t neq = 0;
        for (int64_t j = 0; j < n_tris; ++j) neq += std::max<int64_t>(1, h_corr_count[j]);
        if (!h_corr_faces || n_corr_faces != neq) {
            fail(SDFA_EINVAL, "mesh_create: corr_faces must hold %lld entries (sum of max(1, corr_count)), got %lld", (long long)neq, (long long)n_corr_faces);
            return nullptr;
        }
        for (int64_t k = 0; k < neq; ++k)
            if (h_corr_faces[k] >= (uint32_t)n_src_tris) { fail(SDFA_EINVAL, "mesh_create: corr_faces[%lld] out of range", (long long)k); return nullptr; }
    } else if (n_src_tris != n_tris) {
        fail(SDFA_EINVAL, "mesh_create: without correspondences the dgrad rows must have one 9-vector per template triangle");
        return nullptr;
    }
    // vertex -> free column (deform_triangle_impl.hpp:36-72: constrained vertices leave A for Ar)
    std::vector<int> col(n_verts, 0);
    for (int64_t i = 0; i < n_cnsts; ++i) {
        if (h_cnsts[i] >= (uint32_t)n_verts) { fail(SDFA_EINVAL, "mesh_create: constraint index out of range"); return nullptr; }
        col[h_cnsts[i]] = -1;
    }
    int nf = 0;
    for (int64_t v = 0; v < n_verts; ++v) col[v] = col[v] < 0 ? -1 : nf++;
    if (nf == 0) { fail(SDFA_EINVAL, "mesh_create: every vertex is constrained"); return nullptr; }
    const int fp = (int)round_up(nf, 128);
    // per-triangle U = pinv([v2-v1, v3-v1]) (2x3); A rows 3j..3j+2: v1 -> -U0-U1, v2 -> U0, v3 -> U1   (:81-118)
    std::vector<std::vector<std::pair<int, std::array<double, 3>>>> inc(nf);
    std::vector<double> AtA((size_t)nf * nf, 0.0);
    int64_t eq = 0;   // running equation index (into corr_faces)
    for (int64_t j = 0; j < n_tris; ++j) {
        const uint32_t vi[3] = {h_faces[3 * j], h_faces[3 * j + 1], h_faces[3 * j + 2]};
        for (int k = 0; k < 3; ++k)
            if (vi[k] >= (uint32_t)n_verts) { fail(SDFA_EINVAL, "mesh_create: face index out of range"); return nullptr; }
        double e1[3], e2[3];
        for (int k = 0; k < 3; ++k) {   // the reference subtracts in float (Eigen::Vector3f) before widening
            e1[k] = (double)(float)(h_verts[3 * vi[1] + k] - h_verts[3 * vi[0] + k]);
            e2[k] = (double)(float)(h_verts[3 * vi[2] + k] - h_verts[3 * vi[0] + k]);
        }
        const double g11 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], g22 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
        const double g12 = e1[0] * e2[0] + e1[1] * e2[1] + e1[2] * e2[2], det = g11 * g22 - g12 * g12;
        if (!(det > 0)) { fail(SDFA_EINVAL, "mesh_create: degenerate triangle %lld", (long long)j); return nullptr; }
        std::array<double, 3> u0, u1, c[3];
        for (int k = 0; k < 3; ++k) {
            u0[k] = (g22 * e1[k] - g12 * e2[k]) / det;
            u1[k] = (-g12 * e1[k] + g11 * e2[k]) / det;
            c[0][k] = -u0[k] - u1[k]; c[1][k] = u0[k]; c[2][k] = u1[k];
        }
        const int64_t ncor = h_corr_count ? h_corr_count[j] : 0, nrep = std::max<int64_t>(1, ncor);
        for (int a = 0; a < 3; ++a) {
            const int ca = col[vi[a]];
            if (ca < 0) continue;
            if (!h_corr_count) inc[ca].push_back({(int)j, c[a]});
            else for (int64_t r = 0; r < ncor; ++r) inc[ca].push_back({(int)h_corr_faces[eq + r], c[a]});   // identity equations add nothing to the displacement rhs
            for (int b = 0; b < 3; ++b) {
                const int cb = col[vi[b]];
                if (cb >= 0) AtA[(size_t)ca * nf + cb] += (double)nrep * (c[a][0] * c[b][0] + c[a][1] * c[b][1] + c[a][2] * c[b][2]);
            }
        }
        eq += nrep;
    }
    for (int i = 0; i < nf; ++i) AtA[(size_t)i * nf + i] += reg;   // :125-131
    // dense Cholesky A^T A = L L^T, then (A^T A)^-1 = L^-T L^-1, all in fp64
    std::vector<double> &L = AtA;
    for (int j = 0; j < nf; ++j) {
        double d = L[(size_t)j * nf + j];
        for (int k = 0; k < j; ++k) d -= L[(size_t)j * nf + k] * L[(size_t)j * nf + k];
        if (!(d > 0)) { fail(SDFA_EINVAL, "mesh_create: system matrix not positive definite (column %d)", j); return nullptr; }
        const double ljj = std::sqrt(d);
        L[(size_t)j * nf + j] = ljj;
        for (int i = j + 1; i < nf; ++i) {
            double v = L[(size_t)i * nf + j];
            const double *li = &L[(size_t)i * nf], *lj = &L[(size_t)j * nf];
            for (int k = 0; k < j; ++k) v -= li[k] * lj[k];
            L[(size_t)i * nf + j] = v / ljj;
        }
    }
    std::vector<double> Li((size_t)nf * nf, 0.0);   // L^-1 (lower), row-major
    for (int c0 = 0; c0 < nf; ++c0) {
        Li[(size_t)c0 * nf + c0] = 1.0 / L[(size_t)c0 * nf + c0];
        for (int i = c0 + 1; i < nf; ++i) {
            double v = 0.0;
            const double *li = &L[(size_t)i * nf];
            for (int k = c0; k < i; ++k) v -= li[k] * Li[(size_t)k * nf + c0];
            Li[(size_t)i * nf + c0] = v / li[i];
        }
    }
    // pack: Inv (symmetric) as K4 [fp/4][fp][4], incidence CSR, vertex map, template
    size_t nnz = 0;
    for (auto &v : inc) nnz += v.size();
    const size_t o_inv = 0, o_ptr = o_inv + (size_t)fp * fp, o_tri = o_ptr + round_up(nf + 1, 64), o_coef = o_tri + round_up(nnz, 64),
                 o_col = o_coef + round_up(3 * nnz, 64), o_tm = o_col + round_up(n_verts, 64), o_rx = o_tm + round_up(3 * n_verts, 64),
                 total = o_rx + round_up(3 * nf, 64);
    std::vector<float> hostf(total, 0.f);
    {   // Inv[i][j] = sum_k Li[k][i] Li[k][j], k >= max(i, j); transpose Li first for unit-stride inner loops
        std::vector<double> LiT((size_t)nf * nf);
        for (int i = 0; i < nf; ++i)
            for (int j = 0; j < nf; ++j) LiT[(size_t)j * nf + i] = Li[(size_t)i * nf + j];
        for (int i = 0; i < nf; ++i)
            for (int j = 0; j <= i; ++j) {
                double v = 0.0;
                const double *a = &LiT[(size_t)i * nf], *b = &LiT[(size_t)j * nf];
                for (int k = i; k < nf; ++k) v += a[k] * b[k];
                hostf[o_inv + ((size_t)(j / 4) * fp + i) * 4 + (j % 4)] = (float)v;   // row k = j, output p = i
                hostf[o_inv + ((size_t)(i / 4) * fp + j) * 4 + (i % 4)] = (float)v;
            }
    }
    int *hp = reinterpret_cast<int *>(&hostf[o_ptr]), *ht = reinterpret_cast<int *>(&hostf[o_tri]), *hc = reinterpret_cast<int *>(&hostf[o_col]);
    size_t p = 0;
    for (int v = 0; v < nf; ++v) {
        hp[v] = (int)p;
        for (auto &e : inc[v]) {
            ht[p] = e.first;
            for (int k = 0; k < 3; ++k) hostf[o_coef + 3 * p + k] = (float)e.second[k];
            ++p;
        }
    }
    hp[nf] = (int)p;
    for (int64_t v = 0; v < n_verts; ++v) {
        hc[v] = col[v];
        // the regulariser acts on the absolute position: (A^T A + reg) d = A^T (M - M_I) - reg x_template   (:125-131)
        if (col[v] >= 0)
            for (int k = 0; k < 3; ++k) hostf[o_rx + 3 * col[v] + k] = (float)(reg * (double)h_verts[3 * v + k]);
    }
    memcpy(&hostf[o_tm], h_verts, (size_t)n_verts * 3 * 4);
    auto *m = new sdfa_mesh();
    if (hipMalloc(&m->blob, total * 4) != hipSuccess ||
        hipMemcpyAsync(m->blob, hostf.data(), total * 4, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
        fail(SDFA_EHIP, "mesh_create: device upload failed");
        if (m->blob) (void)hipFree(m->blob);
        delete m;
        return nullptr;
    }
    const float *d = (const float *)m->blob;
    m->n_verts = (int)n_verts; m->n_tris = (int)n_tris; m->n_free = nf; m->free_pad = fp; m->n_src_tris = (int)n_src_tris;
    m->reg_xt = d + o_rx;
    m->inv_k4 = d + o_inv; m->inc_ptr = (const int *)(d + o_ptr); m->inc_tri = (const int *)(d + o_tri);
    m->inc_coef = d + o_coef; m->vert_col = (const int *)(d + o_col); m->tmpl = d + o_tm;
    return m;
}

void sdfa_mesh_destroy(sdfa_mesh *m) {
    if (!m) return;
    if (m->blob) (void)hipFree(m->blob);
    delete m;
}

int64_t sdfa_mesh_workspace_bytes(const sdfa_mesh *m, int64_t n_frames) {
    if (!m || n_frames <= 0) return fail(SDFA_EINVAL, "mesh_workspace_bytes: bad argument");
    return 2 * (int64_t)m->free_pad * round_up(3 * n_frames, 128) * 4;
}

static int mesh_solve(const sdfa_mesh *m, const float *d_dgrad, const int64_t *d_src, const float *d_w, int64_t n_frames,
                      float *d_verts, void *d_workspace, int64_t workspace_bytes, void *stream, const char *who) {
    if (!m) return fail(SDFA_EINVAL, "%s: null mesh", who);
    if (n_frames == 0) return SDFA_OK;
    if (!d_dgrad || !d_verts || !d_workspace || n_frames < 0) return fail(SDFA_EINVAL, "%s: bad argument", who);
    if ((uintptr_t)d_workspace & 15) return fail(SDFA_EINVAL, "%s: workspace must be 16-byte aligned", who);
    if (workspace_bytes < sdfa_mesh_workspace_bytes(m, n_frames)) return fail(SDFA_ENOSPACE, "%s: workspace too small", who);
    hipStream_t s = (hipStream_t)stream;
    const int64_t ld = round_up(3 * n_frames, 128);
    float *rhs = (float *)d_workspace, *sol = rhs + (int64_t)m->free_pad * ld;
    MeshArgs a{};
    a.dgrad = d_dgrad; a.n_frames = n_frames; a.n_tris = m->n_tris; a.n_verts = m->n_verts; a.n_free = m->n_free; a.free_pad = m->free_pad;
    a.n_src_tris = m->n_src_tris; a.seek_src = d_src; a.seek_w = d_w; a.reg_xt = m->reg_xt;
    a.inc_ptr = m->inc_ptr; a.inc_tri = m->inc_tri; a.inc_coef = m->inc_coef; a.vert_col = m->vert_col; a.tmpl = m->tmpl;
    a.rhs = rhs; a.sol = sol; a.verts = d_verts; a.ld = ld;
    HIP_TRY(hipMemsetAsync(rhs, 0, (size_t)m->free_pad * ld * 4, s));   // padding columns / rows feed the GEMM
    HIP_TRY(sdfa_launch_mesh_rhs(a, s));
    GemmArgs g{};
    g.P = m->inv_k4; g.Q = rhs; g.D = sol;
    g.ldp = m->free_pad; g.ldq = ld; g.ldd = ld; g.Ppad = m->free_pad; g.Qpad = ld; g.Pstore = m->free_pad; g.Qreal = ld;
    g.K = m->free_pad; g.seg_k = g.K; g.act = ACT_NONE; g.out_mode = OUT_K4;
    HIP_TRY(sdfa_launch_gemm(g, s));
    HIP_TRY(sdfa_launch_mesh_scatter(a, s));
    return SDFA_OK;
}

int sdfa_mesh_from_dgrad(const sdfa_mesh *m, const float *d_dgrad, int64_t n_frames, float *d_verts, void *d_workspace,
                         int64_t workspace_bytes, void *stream) {
    return mesh_solve(m, d_dgrad, nullptr, nullptr, n_frames, d_verts, d_workspace, workspace_bytes, stream, "mesh_from_dgrad");
}

int sdfa_mesh_from_dgrad_seek(const sdfa_mesh *m, const float *d_dgrad, const int64_t *d_seek_src, const float *d_seek_w,
                              int64_t n_queries, float *d_verts, void *d_workspace, int64_t workspace_bytes, void *stream) {
    if (!d_seek_src || !d_seek_w) return fail(SDFA_EINVAL, "mesh_from_dgrad_seek: seek plan missing");
    return mesh_solve(m, d_dgrad, d_seek_src, d_seek_w, n_queries, d_verts, d_workspace, workspace_bytes, stream, "mesh_from_dgrad_seek");
}

int64_t sdfa_mesh_n_verts(const sdfa_mesh *m) { return m ? m->n_verts : fail(SDFA_EINVAL, "null mesh"); }
int64_t sdfa_mesh_n_src_tris(const sdfa_mesh *m) { return m ? m->n_src_tris : fail(SDFA_EINVAL, "null mesh"); }

// ------------------------------------------------------------------------------------------------
// saber.stream.seek (saber/data/stream/stream.py:20-46) for the uniform video-rate queries of model.py:204-212
int64_t sdfa_seek_query_count(int32_t last_timestamp_ms, double fps) {
    // max_frame = int(tslist[-1] * fps / 1000.0); queries i = 0 .. max_frame          (model.py:205-207)
    const double v = (double)last_timestamp_ms * fps / 1000.0;
    const int64_t mf = (int64_t)v;      // int() truncates toward zero
    return mf < 0 ? 0 : mf + 1;         // range(max_frame + 1) is empty for a negative max_frame
}

int sdfa_seek_plan(const int32_t *d_tslist, const int64_t *d_clip_frame_off, const int64_t *d_clip_query_off, int32_t n_clips,
                   double fps, int64_t n_queries, int64_t *d_seek_src, float *d_seek_w, void *stream) {
    if (n_queries == 0) return SDFA_OK;
    if (!d_tslist || !d_clip_frame_off || !d_clip_query_off || !d_seek_src || !d_seek_w || n_clips <= 0 || n_queries < 0 || !(fps > 0))
        return fail(SDFA_EINVAL, "seek_plan: bad argument");
    HIP_TRY(sdfa_launch_seek_plan(d_tslist, d_clip_frame_off, d_clip_query_off, n_clips, fps, n_queries, d_seek_src, d_seek_w, (hipStream_t)stream));
    return SDFA_OK;
}

int sdfa_ensemble_mean(const float *d_a, const float *d_b, int64_t n, float *d_out, void *stream) {
    if (n == 0) return SDFA_OK;
    if (!d_a || !d_b || !d_out || n < 0) return fail(SDFA_EINVAL, "ensemble_mean: bad argument");
    if (((uintptr_t)d_a | (uintptr_t)d_b | (uintptr_t)d_out) & 3) return fail(SDFA_EINVAL, "ensemble_mean: pointers must be 4-byte aligned");
    HIP_TRY(sdfa_launch_ensemble_mean(d_a, d_b, n, d_out, (hipStream_t)stream));
    return SDFA_OK;
}

int sdfa_seek_rows(const float *d_rows, int64_t row_width, const int64_t *d_seek_src, const float *d_seek_w, int64_t n_queries,
                   float *d_out, void *stream) {
    if (n_queries == 0) return SDFA_OK;
    if (!d_rows || !d_seek_src || !d_seek_w || !d_out || row_width <= 0 || n_queries < 0) return fail(SDFA_EINVAL, "seek_rows: bad argument");
    if (((uintptr_t)d_rows | (uintptr_t)d_out) & 3) return fail(SDFA_EINVAL, "seek_rows: pointers must be 4-byte aligned");
    HIP_TRY(sdfa_launch_seek_rows(d_rows, row_width, d_seek_src, d_seek_w, n_queries, d_out, (hipStream_t)stream));
    return SDFA_OK;
}

}  // extern "C"

// ================================================================================================
// next row: audio ingest -- kaiser_best resampling (resample.hip)
// ================================================================================================
namespace {

double bessel_i0(double x) {   // modified Bessel function of the first kind, order 0: sum_k ((x/2)^k / k!)^2
    const double q = 0.25 * x * x;
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 500; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < sum * 1e-18) break;
    }
    return sum;
}

constexpr int RS_ZEROS = 64, RS_TABLE = 512;                       // resampy kaiser_best: num_zeros, 2**precision
constexpr double RS_BETA = 14.769656459379492, RS_ROLLOFF = 0.9475937167399596;
constexpr int64_t RS_NWIN = (int64_t)RS_ZEROS * RS_TABLE + 1;

// right half of the Kaiser-windowed sinc (resampy/filters.py sinc_window with scipy.signal.kaiser)
void kaiser_best_half(std::vector<double> &win) {
    const int64_t n = RS_NWIN - 1;
    win.resize(RS_NWIN);
    const double i0b = bessel_i0(RS_BETA);
    for (int64_t j = 0; j <= n; ++j) {
        const double xz = RS_ROLLOFF * ((double)j / (double)RS_TABLE);                 // rolloff * linspace(0, 64, n + 1)[j]
        const double py = M_PI * (xz == 0.0 ? 1.0e-20 : xz);                           // np.sinc
        const double sinc = RS_ROLLOFF * (std::sin(py) / py);
        const double r = (double)j / (double)n;                                        // (k - alpha) / alpha, k = n + j
        const double taper = bessel_i0(RS_BETA * std::sqrt(1.0 - r * r)) / i0b;
        win[j] = taper * sinc;
    }
}

struct ResampleTable { void *blob = nullptr; const double *win, *delta; };
std::mutex g_rs_mu;
std::map<std::array<int, 3>, ResampleTable> g_rs;

}  // namespace

extern "C" {

int64_t sdfa_resample_out_len(int64_t n_in, int sr_orig, int sr_new) {
    if (n_in <= 0 || sr_orig <= 0 || sr_new <= 0) return fail(SDFA_EINVAL, "resample_out_len: bad argument");
    if (sr_orig == sr_new) return n_in;
    const double ratio = (double)sr_new / (double)sr_orig;
    return (int64_t)std::ceil((double)n_in * ratio);                // librosa.resample: n_samples = int(np.ceil(y.shape[-1] * ratio))
}

int64_t sdfa_resample_workspace_bytes(int64_t n_in, int sr_orig, int sr_new) {
    const int64_t n = sdfa_resample_out_len(n_in, sr_orig, sr_new);
    return n < 0 ? n : round_up(n * 8, 256);
}

int sdfa_resample_filter(double *h_half_window, int64_t cap) {
    if (!h_half_window || cap < RS_NWIN) return fail(SDFA_EINVAL, "resample_filter: need room for %lld doubles", (long long)RS_NWIN);
    std::vector<double> w;
    kaiser_best_half(w);
    memcpy(h_half_window, w.data(), RS_NWIN * 8);
    return (int)RS_NWIN;
}

int sdfa_resample(const float *d_in, int64_t n_in, int sr_orig, int sr_new, float *d_out, int64_t n_out, void *d_workspace,
                  int64_t workspace_bytes, void *stream) {
    if (!d_in || !d_out || n_in <= 0 || sr_orig <= 0 || sr_new <= 0) return fail(SDFA_EINVAL, "resample: bad argument");
    if (n_out != sdfa_resample_out_len(n_in, sr_orig, sr_new))
        return fail(SDFA_EINVAL, "resample: n_out must be sdfa_resample_out_len() = %lld", (long long)sdfa_resample_out_len(n_in, sr_orig, sr_new));
    hipStream_t s = (hipStream_t)stream;
    if (sr_orig == sr_new) { HIP_TRY(hipMemcpyAsync(d_out, d_in, (size_t)n_in * 4, hipMemcpyDeviceToDevice, s)); return SDFA_OK; }
    const double ratio = (double)sr_new / (double)sr_orig;
    const int64_t n_res = (int64_t)((double)n_in * ratio);          // resampy: shape[axis] = int(shape[axis] * sample_ratio)
    if (n_res < 1) return fail(SDFA_EINVAL, "resample: input signal length=%lld is too small to resample from %d->%d", (long long)n_in, sr_orig, sr_new);
    if (!d_workspace || workspace_bytes < n_res * 8 || ((uintptr_t)d_workspace & 7)) return fail(SDFA_ENOSPACE, "resample: workspace too small or misaligned");
    ResampleTable tb;
    {
        std::lock_guard<std::mutex> lk(g_rs_mu);
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        const std::array<int, 3> key{sr_orig, sr_new, dev};
        auto it = g_rs.find(key);
        if (it == g_rs.end()) {
            std::vector<double> win, both(2 * RS_NWIN, 0.0);
            kaiser_best_half(win);
            for (int64_t j = 0; j < RS_NWIN; ++j) both[j] = ratio < 1.0 ? win[j] * ratio : win[j];     // interp_win *= sample_ratio
            for (int64_t j = 0; j + 1 < RS_NWIN; ++j) both[RS_NWIN + j] = both[j + 1] - both[j];       // interp_delta[:-1] = np.diff(interp_win)
            ResampleTable t;
            HIP_TRY(hipMalloc(&t.blob, both.size() * 8));
            HIP_TRY(hipMemcpy(t.blob, both.data(), both.size() * 8, hipMemcpyHostToDevice));
            t.win = (const double *)t.blob; t.delta = t.win + RS_NWIN;
            it = g_rs.emplace(key, t).first;
        }
        tb = it->second;
    }
    // time register: time_register += 1 / sample_ratio per output sample, accumulated sequentially in float64 like the
    // reference loop (t * increment would round differently).  Uploaded with a blocking copy: this ingest call synchronises.
    std::vector<double> treg((size_t)n_res);
    {
        const double inc = 1.0 / ratio;
        double tr = 0.0;
        for (int64_t t = 0; t < n_res; ++t) { treg[(size_t)t] = tr; tr += inc; }
    }
    HIP_TRY(hipMemcpyAsync(d_workspace, treg.data(), (size_t)n_res * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));     // the host staging vector dies with this call
    ResampleArgs a{};
    a.x = d_in; a.n_in = n_in; a.y = d_out; a.n_res = n_res; a.n_out = n_out; a.win = tb.win; a.delta = tb.delta;
    a.treg = (const double *)d_workspace; a.nwin = RS_NWIN; a.scale = ratio < 1.0 ? ratio : 1.0;
    a.step = (int64_t)(a.scale * (double)RS_TABLE); a.num_table = RS_TABLE;
    if (a.step < 1) return fail(SDFA_EINVAL, "resample: ratio %g is too small for the filter table", ratio);
    HIP_TRY(sdfa_launch_resample(a, s));
    return SDFA_OK;
}

}  // extern "C"


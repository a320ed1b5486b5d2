"""Restatement of the three librosa==0.8.0 functions the hot path calls.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): imported by the oracle, by
oracle/ref_import.py (as the stand-in for the un-vendored `librosa` when the
reference itself is imported in the build container) and by tests.  Never
imported by the product path.

librosa is a third-party dependency of the reference (requirements.txt:4 pins
`librosa == 0.8.0`); it is not vendored under /root/reference and cannot be
installed here (no network).  The algorithms below are restated from the
published librosa 0.8.0 behaviour:

  * filters.mel    -- Slaney-scale triangular filterbank, norm="slaney"
                      (called at saber/data/audio/features/misc.py:110-117)
  * feature.delta  -- scipy.signal.savgol_filter(width=9, polyorder=order,
                      deriv=order, mode="interp")
                      (called at speech_anime/datasets/get_features.py:199-207)
  * feature.rms    -- frame RMS, center=False
                      (called at speech_anime/datasets/sliding_window.py:365;
                       its value never reaches the model)

PARITY UNPINNED against the real librosa (it is absent); pinned against scipy
for the delta filter.
"""
import numpy as np

_F_SP = 200.0 / 3.0
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(freq):
    freq = np.asanyarray(freq, dtype=np.float64)
    mels = freq / _F_SP
    if freq.ndim:
        log_t = freq >= _MIN_LOG_HZ
        mels[log_t] = _MIN_LOG_MEL + np.log(freq[log_t] / _MIN_LOG_HZ) / _LOGSTEP
    elif freq >= _MIN_LOG_HZ:
        mels = _MIN_LOG_MEL + np.log(freq / _MIN_LOG_HZ) / _LOGSTEP
    return mels


def mel_to_hz(mels):
    mels = np.asanyarray(mels, dtype=np.float64)
    freqs = _F_SP * mels
    if mels.ndim:
        log_t = mels >= _MIN_LOG_MEL
        freqs[log_t] = _MIN_LOG_HZ * np.exp(_LOGSTEP * (mels[log_t] - _MIN_LOG_MEL))
    elif mels >= _MIN_LOG_MEL:
        freqs = _MIN_LOG_HZ * np.exp(_LOGSTEP * (mels - _MIN_LOG_MEL))
    return freqs


def mel_filters(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """(n_mels, 1 + n_fft//2) float32 Slaney-normalised triangular filters."""
    if fmax is None:
        fmax = float(sr) / 2
    n_bins = 1 + n_fft // 2
    weights = np.zeros((n_mels, n_bins), dtype=np.float32)
    fftfreqs = np.linspace(0, float(sr) / 2, n_bins, endpoint=True)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def delta(data, width=9, order=1, axis=-1, mode="interp"):
    import scipy.signal
    return scipy.signal.savgol_filter(data, width, deriv=order, polyorder=order, axis=axis, mode=mode)


def rms(y, frame_length=2048, hop_length=512, center=False):
    y = np.asarray(y)
    n = 1 + (len(y) - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n)[:, None]
    fr = y[idx].astype(np.float64)
    return np.sqrt(np.mean(fr * fr, axis=1))[None, :].astype(np.float32)

"""CPU restatement (numpy) of the reference's audio resampling step.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic lives in third-party code that is absent from /root/reference and from this image --
`librosa == 0.8.0` (requirements.txt:4) `librosa.resample(..., res_type="kaiser_best")`, which calls `resampy.resample`
(resampy 0.2.2, the version librosa 0.8.0 depends on).  The reference's call sites are
speech_anime/model/eval_utils.py:76-86 (`saber.audio.load(path, 44100)` = librosa.load at 44.1 kHz, then
`librosa.resample(sound_signal, orig_sr=44100, target_sr=sr)`) and saber/data/audio/io.py:9-15.  The reference holds no
test, golden vector or fixture for it, so this file restates the PUBLISHED algorithm:

  * filter `kaiser_best` (resampy/filters.py sinc_window): half of a Kaiser-windowed sinc, 64 zero crossings, 2**9 table
    entries per zero crossing, beta = 14.769656459379492, rolloff = 0.9475937167399596;
  * resampy/interpn.py resample_f: per output sample a time register advanced by 1/ratio (accumulated in float64), the
    filter sampled every int(min(1, ratio) * 512) table entries from the fractional offset with linear interpolation
    between entries, left wing then right wing, products in float64 ADDED INTO THE OUTPUT ARRAY, which has the input's
    dtype (float32): the running sum is rounded to float32 after every tap;
  * resampy.resample: output length int(n * ratio), filter scaled by the ratio when downsampling;
  * librosa.resample: fix_length to ceil(n * ratio) (zero pad), result in the input dtype.
"""
import numpy as np
import scipy.signal

NUM_ZEROS, PRECISION_BITS = 64, 9
BETA, ROLLOFF = 14.769656459379492, 0.9475937167399596


def kaiser_best():
    """(half window float64[64 * 512 + 1], table entries per zero crossing)."""
    num_table = 2 ** PRECISION_BITS
    n = num_table * NUM_ZEROS
    sinc_win = ROLLOFF * np.sinc(ROLLOFF * np.linspace(0, NUM_ZEROS, num=n + 1, endpoint=True))
    taper = scipy.signal.windows.kaiser(2 * n + 1, BETA)[n:]
    return taper * sinc_win, num_table


def resampy_resample(x, sr_orig, sr_new):
    x = np.asarray(x)
    assert x.ndim == 1 and x.dtype == np.float32
    ratio = float(sr_new) / sr_orig
    n_out = int(x.shape[0] * ratio)
    if n_out < 1:
        raise ValueError(f"Input signal length={x.shape[0]} is too small to resample from {sr_orig}->{sr_new}")
    win, num_table = kaiser_best()
    win = win.copy()
    if ratio < 1:
        win *= ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, ratio)
    inc = 1.0 / ratio
    step = int(scale * num_table)
    nwin, n_orig = win.shape[0], x.shape[0]
    # time register: sequential float64 accumulation
    treg = np.empty(n_out, np.float64)
    t = 0.0
    for k in range(n_out):
        treg[k] = t
        t += inc
    n = treg.astype(np.int64)                               # int(time_register)
    y = np.zeros(n_out, np.float32)
    xd = x.astype(np.float64)
    for wing in (0, 1):
        frac = scale * (treg - n)
        if wing == 1:
            frac = scale - frac
        index_frac = frac * num_table
        offset = index_frac.astype(np.int64)
        eta = index_frac - offset
        if wing == 0:
            kmax = np.minimum(n + 1, (nwin - offset) // step)
        else:
            kmax = np.minimum(n_orig - n - 1, (nwin - offset) // step)
        for i in range(int(kmax.max()) if len(kmax) else 0):
            live = i < kmax
            idx = np.where(live, offset + i * step, 0)
            w = win[idx] + eta * delta[idx]
            src = np.where(live, n - i if wing == 0 else n + i + 1, 0)
            acc = (y.astype(np.float64) + w * xd[src]).astype(np.float32)       # y[t] += weight * x[...]  on a float32 array
            y = np.where(live, acc, y)
    return y


def librosa_resample(y, orig_sr, target_sr):
    """librosa 0.8.0 resample(y, orig_sr, target_sr, res_type='kaiser_best', fix=True, scale=False)."""
    y = np.asarray(y, np.float32)
    if orig_sr == target_sr:
        return y
    ratio = float(target_sr) / orig_sr
    n_samples = int(np.ceil(y.shape[-1] * ratio))
    out = resampy_resample(y, orig_sr, target_sr)
    if len(out) > n_samples:                                 # util.fix_length
        out = out[:n_samples]
    elif len(out) < n_samples:
        out = np.pad(out, (0, n_samples - len(out)), mode="constant")
    return np.ascontiguousarray(out, dtype=np.float32)


def load_source_chain(pcm, native_sr, model_sr):
    """eval_utils.py:76-86 for a .wav: librosa.load(path, sr=44100) (resamples the decoded float32 mono signal when the
    file's rate differs), then librosa.resample(44100 -> model rate).  Returns (signal at model_sr, sound_signal at 44.1 kHz)."""
    sound = librosa_resample(pcm, native_sr, 44100)
    return librosa_resample(sound, 44100, model_sr), sound

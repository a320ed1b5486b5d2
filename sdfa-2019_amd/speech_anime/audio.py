"""Audio ingest helpers on the host (numpy): the steps around the hot path that evaluate() needs.

  rms_normalize -- saber.audio.rms.normalize (saber/data/audio/rms.py:45-78), called at speech_anime/model/model.py:165
  load_source   -- the .wav branch of speech_anime/model/eval_utils.py:50-93

The reference loads everything through librosa at 44.1 kHz and resamples with resampy (kaiser_best); neither
library is available offline, so files must already be PCM WAV at the model rate (or .npy float32 PCM).
Resampler parity is a listed next row, not part of this path.
"""
import os

import numpy as np


def rms_normalize(wav, target_db=-20, threshold=None):
    """Scale `wav` so that the RMS of its non-silent samples sits at `target_db` dBFS, then clip to +-0.999.

    Behaviour of saber.audio.rms.normalize (saber/data/audio/rms.py:45-78): per-sample level 20*log10(max(|x|, 1e-10));
    samples at or above `threshold` dB (default: the quietest sample, i.e. all of them) enter the RMS; an empty
    selection returns the input untouched.  Pinned by tests/golden/host_rows.npz.
    """
    level = 20.0 * np.log10(np.maximum(np.abs(wav), 1e-10))
    keep = level >= (level.min() if threshold is None else threshold)
    if not keep.any():
        return wav
    rms_db = 20.0 * np.log10(np.sqrt(np.mean(wav[keep] ** 2)))
    gain = np.power(10.0, (target_db - rms_db) / 20.0)
    return np.clip(wav * gain, -0.999, 0.999)


def load_source(path, sr):
    path = os.path.expanduser(path)
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return np.load(path).astype(np.float32).reshape(-1)
    if ext == ".wav":
        from scipy.io import wavfile
        file_sr, data = wavfile.read(path)
        if data.ndim > 1:
            data = data.mean(axis=1)
        if np.issubdtype(data.dtype, np.integer):
            data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
        if file_sr != sr:
            raise ValueError(f"{path}: sample rate {file_sr} != model rate {sr}; resampling (librosa/resampy in the "
                             "reference) is not part of this path -- convert the file first")
        return data.astype(np.float32)
    raise ValueError(f"{ext} is not supported (the reference decodes video/audio containers through librosa/ffmpeg)")

"""Audio ingest helpers on the host (numpy): the steps around the hot path that evaluate() needs.

  rms_normalize -- saber.audio.rms.normalize (saber/data/audio/rms.py:45-78), called at speech_anime/model/model.py:165
  load_source   -- the .wav branch of speech_anime/model/eval_utils.py:50-93

The reference loads everything through librosa at 44.1 kHz and resamples with resampy (kaiser_best); neither library is
available offline, so the resampling arithmetic is restated as a HIP kernel (csrc/resample.hip, parity unpinned) and WAV
decoding uses scipy.io.wavfile.
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


SOUND_SR = 44100      # every source is first brought to 44.1 kHz (eval_utils.py:77,83; also the rate of the exported audio.wav)


def read_wav(path):
    """Decoded mono float32 signal and the file's own rate -- what librosa.load does before resampling: integer PCM
    scaled by 1 / 2**(bits-1) (soundfile's float32 read), channels averaged (librosa.to_mono)."""
    from scipy.io import wavfile
    file_sr, data = wavfile.read(path)
    if np.issubdtype(data.dtype, np.integer):
        if data.dtype == np.uint8:
            data = (data.astype(np.float32) - 128.0) / 128.0
        else:
            data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    data = data.astype(np.float32)
    if data.ndim > 1:
        data = data.mean(axis=1, dtype=np.float32)
    return data, int(file_sr)


def write_wav(path, signal, sr):
    """saber.audio.save (saber/data/audio/io.py:19-24: soundfile.write, 16-bit PCM for a .wav)."""
    from scipy.io import wavfile
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    x = np.asarray(signal, np.float32).reshape(-1)
    pcm = np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)      # libsndfile's float -> int16 conversion clips
    wavfile.write(path, int(sr), pcm)


def load_source(path, sr, return_sound=False):
    """The .wav / .npy branch of speech_anime/model/eval_utils.py:50-93.

    .wav: decode, bring to 44.1 kHz (`sound_signal`, what saber.audio.load(path, 44100) returns), then 44.1 kHz -> `sr`
    (`signal`) -- both conversions on the GPU (sdfa_amd.resample: the kaiser_best arithmetic of librosa.resample, restated;
    parity unpinned).  So a 16 kHz WAV takes the reference's 16 k -> 44.1 k -> 8 k route with the in-repo 8 kHz config.
    .npy: float32 PCM already at the model rate (no reference counterpart; used by tests and synthetic runs).
    Video / compressed containers (.mp4, .m4v, .avi) go through audioread/ffmpeg in the reference; no decoder exists in
    this image, so they are refused by name."""
    path = os.path.expanduser(path)
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        signal = np.load(path).astype(np.float32).reshape(-1)
        return (signal, None) if return_sound else signal
    if ext == ".wav":
        from sdfa_amd.resample import resample
        data, file_sr = read_wav(path)
        sound = data if file_sr == SOUND_SR else resample(data, file_sr, SOUND_SR)
        signal = resample(sound, SOUND_SR, sr).cpu().numpy() if sr != SOUND_SR else np.asarray(sound if not hasattr(sound, "cpu") else sound.cpu().numpy())
        if return_sound:
            return signal, (sound.cpu().numpy() if hasattr(sound, "cpu") else sound)
        return signal
    if ext in (".mp4", ".m4v", ".avi"):
        wav = os.path.splitext(path)[0] + ".wav"
        raise ValueError(f"{path}: decoding {ext} needs audioread/ffmpeg (the reference's librosa.load route, saber/data/audio/io.py:9-15), "
                         f"which this image does not have.  Extract the audio track first -- `ffmpeg -i {path} -vn -ac 1 {wav}` (any sample "
                         f"rate: the ingest resamples to 44.1 kHz and then to the model rate like the reference) -- and pass "
                         f"`--eval_input {wav}` (evaluate.sh:12 names an .mp4 by default)")
    raise ValueError(f"{ext} is not supported!")

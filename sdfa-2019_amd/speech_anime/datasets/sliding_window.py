"""DatasetSlidingWindow -- the inference half of speech_anime/datasets/sliding_window.py (fetch_audio_features and
the unit converters of speech_anime/datasets/speech_anime.py:128-164).  The training data loader is out of scope."""
import weakref

import numpy as np
import torch

from sdfa_amd import engine as _engine


class DatasetSlidingWindow:
    hparams = None
    _engine = None          # a bare front-end engine, created on demand
    _engine_ref = None      # weak reference to the model's engine (set by SaberSpeechDrivenAnimation.load_state_dict): the class must
                            # not keep a dropped model's weights and workspace alive

    # ---- units: same float32 roundings as the reference (speech_anime.py:128-164)
    @classmethod
    def ms_to_sample(cls, ms, sr=None, dtype=np.float32):
        sr = sr or cls.hparams.audio.sample_rate
        return dtype(float(ms * sr) / 1000.0)

    @classmethod
    def sample_to_ms(cls, sample, sr=None, dtype=np.float32):
        sr = sr or cls.hparams.audio.sample_rate
        return dtype(float(sample * 1000.0) / float(sr))

    @classmethod
    def frame_to_sample(cls, idx, sr=None, fps=None, dtype=np.float32):
        sr = sr or cls.hparams.audio.sample_rate
        fps = fps or cls.hparams.anime.fps
        return dtype(float(idx * sr) / float(fps))

    @classmethod
    def sample_to_frame(cls, sample, sr=None, fps=None, dtype=np.float32):
        sr = sr or cls.hparams.audio.sample_rate
        fps = fps or cls.hparams.anime.fps
        return dtype(float(sample * fps) / float(sr))

    @classmethod
    def frame_in_range(cls, frame_idx, sliding_size, start, end):
        return start + cls.frame_to_sample(frame_idx) + sliding_size <= end

    # ---- hot path
    @classmethod
    def fetch_audio_features(cls, signal, hparams=None, as_numpy=True):
        """sliding_window.py:324-377: dict(tslist, energy (F,1,64), audio_feat (F,64,128,3)).

        `as_numpy=False` keeps audio_feat on the GPU (the reference always returns numpy)."""
        if hparams is not None and cls.hparams is None:
            cls.hparams = hparams
        hp = cls.hparams
        if torch.is_tensor(signal):
            signal = signal.detach().cpu().numpy()
        signal = np.asarray(signal, np.float32).reshape(-1)
        assert -1.0 <= signal.min() and signal.max() <= 1.0                      # :330
        sr = hp.audio.sample_rate
        fe = cls._frontend_engine()
        table = _engine.frame_index(len(signal), sr)                             # ONE enumeration per clip: front end and energy share it
        feat, tslists, _ = fe.mel_frontend([signal], sr, tables=[table])
        energy = cls._energy(signal, sr, table[0])
        return dict(tslist=tslists[0], energy=energy,
                    audio_feat=feat.cpu().numpy() if as_numpy else feat)

    @classmethod
    def use_engine(cls, engine):
        cls._engine_ref = weakref.ref(engine)

    @classmethod
    def _frontend_engine(cls):
        eng = cls._engine_ref() if cls._engine_ref is not None else None
        if eng is not None:
            return eng
        # a bare front end lives on the configured device (hparams.device; api.evaluate_model sets it per rank), else the CURRENT one
        want = (cls.hparams.get("device") if cls.hparams is not None else None) or None
        if cls._engine is None or (want is not None and cls._same_device(cls._engine.device, want) is False):
            cls._engine = _engine.FrontendOnly(device=want)
        return cls._engine

    @staticmethod
    def _same_device(have, want):
        """torch.device comparison with an index-less "cuda" meaning the CURRENT device (FrontendOnly normalises its own the same way):
        'cuda' against 'cuda:0' is the same card, not a reason to build a new front end -- and a new 0.7 GB workspace -- per call."""
        a, b = torch.device(have), torch.device(want)
        if a.type != b.type:
            return False
        if a.type != "cuda":
            return True
        cur = torch.cuda.current_device() if torch.cuda.is_available() else 0
        return (cur if a.index is None else a.index) == (cur if b.index is None else b.index)

    @staticmethod
    def _energy(signal, sr, starts=None):
        """librosa.feature.rms(frame_length=win, hop_length=hop, center=False) per window (sliding_window.py:365).
        Carried in the result for interface parity; no model consumes it (model.py:443,487)."""
        win, hop, sliding = _engine.frame_geometry(sr)
        if starts is None:
            starts, _ = _engine.frame_index(len(signal), sr)
        L = len(signal)
        csum = np.concatenate([[0.0], np.cumsum(signal.astype(np.float64) ** 2)])
        a = starts[:, None] + hop * np.arange(64)[None, :]
        lo, hi = np.clip(a, 0, L), np.clip(a + win, 0, L)
        return np.sqrt((csum[hi] - csum[lo]) / win).astype(np.float32)[:, None, :]

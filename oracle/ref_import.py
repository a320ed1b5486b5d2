"""Import the *reference itself* (read-only at /root/reference) in the build container.

TEST INFRASTRUCTURE ONLY.  Used by oracle/gen_golden.py to produce the committed
fixtures under tests/golden/.  Nothing here travels to the GPU box in a usable
form (there is no /root/reference there) and the product path never imports it.

Recipe: run with `python3 -B` so no bytecode is written
into the reference tree; pre-populate sys.modules with inert stand-ins for the
third-party modules the image lacks (colorama, termcolor, librosa, cv2, ...) and
for three reference modules whose import has side effects or GL dependencies
(saber.data.audio.denoise: git clone + make at import; deformation: in-tree
cmake at import; speech_anime.viewer: pyrender).  `librosa` is replaced by
oracle/librosa_restate.py (parity unpinned vs the real library, see there).
"""
import os
import sys
import types
import importlib.machinery

REFERENCE_ROOT = os.environ.get("SDFA_REFERENCE_ROOT", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Anything:
    """Inert object: any attribute / call / item returns another inert object."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything()

    def __getitem__(self, k):
        return _Anything()

    def __iter__(self):
        return iter(())

    def __str__(self):
        return ""

    def __add__(self, o):
        return o if isinstance(o, str) else self

    __radd__ = __add__


def _inert(name, *subs):
    m = _mod(name)

    def _ga(k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything()
    m.__getattr__ = _ga
    m.__path__ = []
    for s in subs:
        sm = _inert(f"{name}.{s}")
        setattr(m, s, sm)
    return m


def install_stubs():
    import numpy as np
    import torch  # noqa: F401  (import the real torch before any stand-in is registered)
    sys.dont_write_bytecode = True
    if _HERE not in sys.path:
        sys.path.insert(0, _HERE)
    import librosa_restate as lr

    # ---- third-party modules absent from the image ------------------------------------
    for name, subs in (
        ("colorama", ()), ("termcolor", ()), ("soundfile", ()), ("plyfile", ()),
        ("webrtcvad", ()), ("pyrender", ()), ("trimesh", ()), ("pysptk", ()),
        ("tensorboard", ()), ("resampy", ()), ("ffmpeg", ()), ("tqdm_stub", ()),
    ):
        if name not in sys.modules:
            _inert(name, *subs)
    sys.modules["termcolor"].colored = lambda s, *a, **k: s
    colorama = sys.modules["colorama"]
    colorama.init = lambda *a, **k: None

    class _Codes:
        def __getattr__(self, k):
            return ""
    colorama.Fore = _Codes(); colorama.Back = _Codes(); colorama.Style = _Codes()

    # torch.utils.tensorboard pulls the real tensorboard; give it an inert writer
    tb = _inert("torch.utils.tensorboard")
    tb.SummaryWriter = _Anything

    # librosa -> restated algorithms
    librosa = _inert("librosa", "core", "util", "effects", "output", "display")
    filters = _inert("librosa.filters")
    feature = _inert("librosa.feature")
    filters.__dict__.update(mel=lambda sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **k:
                   lr.mel_filters(sr, n_fft, n_mels, fmin, fmax))
    feature.__dict__.update(
                   delta=lambda data, width=9, order=1, axis=-1, mode="interp", **k:
                   lr.delta(data, width, order, axis, mode),
                   rms=lambda y=None, frame_length=2048, hop_length=512, center=True, **k:
                   lr.rms(y, frame_length, hop_length, center))
    librosa.filters = filters
    librosa.feature = feature

    # cv2.resize to the same size is the identity (and drops a singleton channel)
    def _resize(src, dsize, interpolation=None, **k):
        w, h = dsize
        assert src.shape[0] == h and src.shape[1] == w, "stub cv2.resize: identity only"
        out = np.array(src, copy=True)
        if out.ndim == 3 and out.shape[2] == 1:
            out = out[:, :, 0]
        return out
    cv2 = _inert("cv2")
    cv2.resize = _resize
    cv2.INTER_LINEAR = 1

    # ---- reference modules with import-time side effects / GL deps ---------------------
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    den = _inert("saber.data.audio.denoise")
    den.denoise = _Anything()
    den.logmmse = _Anything()
    _inert("deformation")
    viewer = _inert("speech_anime.viewer")
    viewer.set_template_mesh = lambda *a, **k: None


def load_reference(custom_hparams="dgrad", sample_rate=None):
    """Return (hparams, saber_model, DatasetSlidingWindow) of the reference, CPU, eval mode.

    The model is constructed exactly as api.evaluate_model does
    (speech_anime/api.py:79-99), minus checkpoint loading.
    """
    install_stubs()
    import torch  # noqa
    import saber
    from speech_anime.tools import configure
    from speech_anime.model import SaberSpeechDrivenAnimation
    from speech_anime.datasets import DatasetSlidingWindow

    args = saber.ConfigDict(dict(
        mode="evaluate", custom_hparams=custom_hparams, template_mesh=None,
        mesh_constraints=None, mesh_tricorres=None, log_dir=None))
    hp = configure(args)
    if sample_rate is not None:
        hp.audio.set_key("sample_rate", int(sample_rate))
    hp.set_key("device", "cpu")
    model = SaberSpeechDrivenAnimation(hp, None, None, load_pca=False)
    model.eval()
    DatasetSlidingWindow.hparams = None  # class-level cache (sliding_window.py:326-327)
    return hp, model, DatasetSlidingWindow

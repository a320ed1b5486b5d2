"""Inference hyper-parameters: the values of speech_anime/config/default.py, config/model/{dgrad,offsets}.py and
config/data/voca-dgrad.py that the hot path reads, in an attribute-access dict, optionally overwritten from the
`hparams.json` that ships with a checkpoint (--custom_hparams)."""
import copy
import json
import os


class HP(dict):
    """Nested dict with attribute access (the slice of saber.ConfigDict the path uses)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v

    def set_key(self, k, v):
        self[k] = _wrap(v)

    def get(self, k, default=None):
        return self[k] if k in self else default


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, HP):
        return HP({k: _wrap(x) for k, x in v.items() if k != "__entirety__"})
    if isinstance(v, (list, tuple)):
        return [_wrap(x) for x in v]
    return v


_SPEAKERS = dict(m0=0, f0=1, m1=2, m2=3, f1=4, m3=5, f2=6, f3=7, f4=8, m4=9, m5=10, f5=11)   # config/data/voca-dgrad.py:42-47

DEFAULTS = dict(
    tag="dgrad",
    audio=dict(
        sample_rate=8000,                                                   # config/data/voca-dgrad.py:4
        mel=dict(n_mels=128, win_size=0.064, hop_size=0.008, win_fn="hamm", padding=False, fmin=50, fmax=3600,
                 ref_db=20, top_db=80, normalize=True, clip_normalized=True, subtract_mean=False, preemphasis=0.65),
        feature=dict(name="mel", with_delta=True, sliding_window_frames=64),  # config/model/dgrad.py:5-11
    ),
    anime=dict(fps=60, feature=dict(ts_delta=100)),                          # config/data/voca-dgrad.py:31-36
    dataset_anime=dict(audio_target_db=-24.5, speakers=_SPEAKERS, emotions=dict(neutral=0)),
    ensembling_ms=0,                                                         # config/model/dgrad.py:48
    model=dict(face_data_type="dgrad_3d", prediction_type="face_data", weight_norm=True,
               speaker_embedding=dict(using_onehot=True, num_speakers=8)),
    trainer=dict(evaluate=dict(test=[])),
    precision="fp32",       # not in the reference (fp32 only): MFMA mode of this build, see include/sdfa_hip.h SDFA_PREC_*
    device="cuda:0", eval_input=None, eval_spk_cond=None, load_from=None, log_dir=None,
)


def _overwrite(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict) and not v.get("__entirety__", False):
            _overwrite(dst[k], v)
        else:
            dst[k] = v


def configure(args):
    """Counterpart of speech_anime/tools/config.py:15-87 for mode == "evaluate"."""
    args = args if isinstance(args, dict) else vars(args)
    hp = copy.deepcopy(DEFAULTS)
    custom = args.get("custom_hparams")
    if custom is not None:
        if custom in ("dgrad", "offsets"):
            if custom == "offsets":
                hp["tag"] = "offsets"
                hp["model"]["face_data_type"] = "verts_off_3d"
        else:
            path = os.path.expanduser(custom)
            if not os.path.exists(path) and args.get("log_dir"):
                path = os.path.join(args["log_dir"], custom)
            with open(path) as f:
                loaded = json.load(f)
            if "evaluate" in loaded.get("trainer", {}):          # tools/config.py:43-44
                del loaded["trainer"]["evaluate"]
            _overwrite(hp, loaded)
    for key in ("tag", "seed", "log_dir", "load_from", "ensembling_ms", "eval_input", "eval_spk_cond"):   # config.py:52-60
        if args.get(key) is not None:
            hp[key] = args[key]
    sr = hp["audio"]["sample_rate"]
    if sr not in (8000, 16000):
        raise ValueError(f"audio.sample_rate = {sr}: the MI355X front end supports 8000 and 16000 Hz")
    check_compiled_in(hp)
    return _wrap(hp)


# Front-end and frame-geometry values the kernels and sdfa_frame_index are built for (csrc/api.cpp build_frontend,
# csrc/frontend.hip, sdfa_amd/engine.py).  A checkpoint's hparams.json that disagrees would run without error and give
# wrong timestamps / features, so it is refused by name instead.
_COMPILED_IN = {
    "audio.mel.n_mels": 128, "audio.mel.win_size": 0.064, "audio.mel.hop_size": 0.008, "audio.mel.win_fn": "hamm",
    "audio.mel.padding": False, "audio.mel.fmin": 50, "audio.mel.fmax": 3600, "audio.mel.ref_db": 20, "audio.mel.top_db": 80,
    "audio.mel.normalize": True, "audio.mel.clip_normalized": True, "audio.mel.subtract_mean": False,
    "audio.mel.preemphasis": 0.65, "audio.feature.name": "mel", "audio.feature.with_delta": True,
    "audio.feature.sliding_window_frames": 64, "anime.fps": 60, "anime.feature.ts_delta": 100,
    "model.speaker_embedding.using_onehot": True, "model.speaker_embedding.num_speakers": 8,
}


def check_compiled_in(hp):
    for path, want in _COMPILED_IN.items():
        node = hp
        for k in path.split("."):
            if not isinstance(node, dict) or k not in node:
                node = None
                break
            node = node[k]
        if node is None:
            continue            # absent: the default (= compiled-in value) applies
        same = (node == want) if isinstance(want, (str, bool)) else (isinstance(node, (int, float)) and abs(float(node) - float(want)) < 1e-9)
        if not same:
            raise ValueError(f"hparams.{path} = {node!r}: this build of the front end / frame geometry is compiled for {want!r}; "
                             "a checkpoint trained with another value cannot be evaluated by it")

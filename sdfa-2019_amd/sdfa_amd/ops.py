"""PyTorch-ROCm custom operators over the C ABI: `torch.ops.sdfa.*` (BASELINE north_star: "called from the existing Python
host through PyTorch-ROCm custom ops"; SURVEY.md section 8(b) names them).

    torch.ops.sdfa.frame_index(n_samples, sample_rate, fps, ts_delta)                    -> (starts int64[F], tslist int32[F])   (host)
    torch.ops.sdfa.mel_frontend(pcm, clip_off, clip_len, frame_clip, frame_start, sr)     -> audio_feat f32[F,64,128,3]
    torch.ops.sdfa.encoder(audio_feat, model)                                             -> (z f32[N,512], align f32[N,64])
    torch.ops.sdfa.encoder_shared(audio_feat, frame_clip, frame_start, hop, model)        -> the same, bitwise, with the per-column stages
                                                                                             run once per distinct column of the frame table
    torch.ops.sdfa.regress(z, speaker_id, model)                                          -> dgrad f32[N,89784] | offsets f32[N,15069]
    torch.ops.sdfa.regress_into(z, speaker_id, out, model)                                -> None (rows written into `out`; ids must be validated)
    torch.ops.sdfa.regress_coef(z, speaker_id, model)                                     -> PCA coefficients f32[N,265 | 59]

Thin by design: each op is the matching `Engine` method (ctypes call into libsdfa_hip.so on the current stream), registered
with the dispatcher through `torch.library.custom_op` and given a fake-tensor shape function, so the ops compose with
torch.compile / torch.export / torch.jit.trace like any other operator.  `model` is the string key of a model in this
process' registry (`register_model` / `load_model`); a key that is the path of a checkpoint file is loaded on first use, which
is what lets a module traced by `speech_anime.api.jit_trace` run in a fresh process.  No CPU kernels are registered: calling a
device op with CPU tensors raises (the product path has no CPU fallback).
"""
import itertools
import os
import weakref
from typing import Tuple

import numpy as np
import torch
from torch import Tensor
from torch.library import custom_op

_MODELS = {}               # key -> Engine, for models loaded BY this module (load_model: the registry owns them)
_WEAK = {}                 # key -> weakref to an Engine owned by someone else (register_model): dropped with its owner
_ANON = itertools.count(1)


def register_model(key, engine, own=False):
    """`engine`: a sdfa_amd.engine.Engine (anything with encoder / regress / out_dim / coef_dim).  The registry keeps a WEAK
    reference unless `own` is set: an Engine holds the weights and a multi-GB workspace, and a model object that is dropped
    (SpeechDrivenAnimation re-loaded in a long-lived process) must free them, as a dropped module does in the reference.
    `key` None: a fresh key from a monotonically increasing counter (never reused, unlike id())."""
    key = f"sdfa-model-{next(_ANON)}" if key is None else str(key)
    _MODELS.pop(key, None)
    _WEAK.pop(key, None)
    if own:
        _MODELS[key] = engine
    else:
        _WEAK[key] = weakref.ref(engine, lambda _r, k=key: _WEAK.pop(k, None) if _WEAK.get(k) is _r else None)
    return key


def unregister_model(key):
    _MODELS.pop(str(key), None)
    _WEAK.pop(str(key), None)


def load_model(key, state_dict=None, **engine_kwargs):
    """Registers Engine(state_dict) under `key`; without a state_dict `key` must be a checkpoint path (torch.save'd dict
    with a "state" entry, the reference's layout: saber/trainer/manager/checkpoints.py:10-48)."""
    from .engine import Engine
    if state_dict is None:
        from .weights import ckpt_backward_compatible_preprocess
        ckpt = torch.load(os.path.expanduser(key), map_location="cpu", weights_only=False)
        if "hamm" in ckpt["state"]:
            ckpt = ckpt_backward_compatible_preprocess(ckpt)
        state_dict = ckpt["state"]
    return register_model(key, Engine(state_dict, **engine_kwargs), own=True)


def _model(key):
    m = _MODELS.get(key)
    if m is None and key in _WEAK:
        m = _WEAK[key]()
    if m is None:
        if os.path.isfile(os.path.expanduser(key)):
            load_model(key)
            return _MODELS[key]
        raise KeyError(f"sdfa: no model registered as {key!r} (sdfa_amd.ops.register_model / load_model)")
    return m


# --------------------------------------------------------------------------------------------------- frame_index (host)
@custom_op("sdfa::frame_index", mutates_args=())
def frame_index(n_samples: int, sample_rate: int, fps: int, ts_delta: int) -> Tuple[Tensor, Tensor]:
    from .engine import frame_index as _fi
    starts, ts = _fi(n_samples, sample_rate, fps, ts_delta)
    return torch.from_numpy(np.ascontiguousarray(starts)), torch.from_numpy(np.ascontiguousarray(ts))


@frame_index.register_fake
def _(n_samples, sample_rate, fps, ts_delta):
    ctx = torch.library.get_ctx()
    f = ctx.new_dynamic_size()          # F = floor((L + sliding) * fps / sr) + 2 in exact arithmetic; data dependent under float32 rounding
    return torch.empty(f, dtype=torch.int64), torch.empty(f, dtype=torch.int32)


# --------------------------------------------------------------------------------------------------- front end
_FE = {}


def _frontend(device):
    from .engine import FrontendOnly
    key = str(device)
    if key not in _FE:
        _FE[key] = FrontendOnly(device)
    return _FE[key]


@custom_op("sdfa::mel_frontend", mutates_args=(), device_types="cuda")
def mel_frontend(pcm: Tensor, clip_off: Tensor, clip_len: Tensor, frame_clip: Tensor, frame_start: Tensor, sample_rate: int) -> Tensor:
    assert pcm.dtype == torch.float32 and clip_off.dtype == torch.int64 and clip_len.dtype == torch.int64
    assert frame_clip.dtype == torch.int32 and frame_start.dtype == torch.int64
    return _frontend(pcm.device).mel_frontend_device(pcm.contiguous(), clip_off.contiguous(), clip_len.contiguous(),
                                                     frame_clip.contiguous(), frame_start.contiguous(), sample_rate)


@mel_frontend.register_fake
def _(pcm, clip_off, clip_len, frame_clip, frame_start, sample_rate):
    return pcm.new_empty((frame_clip.shape[0], 64, 128, 3))


# --------------------------------------------------------------------------------------------------- model
@custom_op("sdfa::encoder", mutates_args=(), device_types="cuda")
def encoder(audio_feat: Tensor, model: str) -> Tuple[Tensor, Tensor]:
    z, align = _model(model).encoder(audio_feat, want_align=True)
    return z, align


@encoder.register_fake
def _(audio_feat, model):
    n = audio_feat.shape[0]
    return audio_feat.new_empty((n, 512)), audio_feat.new_empty((n, 64))


@custom_op("sdfa::encoder_shared", mutates_args=(), device_types="cuda")
def encoder_shared(audio_feat: Tensor, frame_clip: Tensor, frame_start: Tensor, hop: int, model: str) -> Tuple[Tensor, Tensor]:
    """sdfa_encoder_forward_shared: `audio_feat` must be what mel_frontend produced for exactly this frame table
    (frame_clip int32 [N], frame_start int64 [N], hop = int(0.008 * sr)); bitwise the result of sdfa::encoder."""
    assert frame_clip.dtype == torch.int32 and frame_start.dtype == torch.int64
    z, align = _model(model).encoder(audio_feat, want_align=True, frame_clip=frame_clip, frame_start=frame_start, hop=hop)
    return z, align


@encoder_shared.register_fake
def _(audio_feat, frame_clip, frame_start, hop, model):
    n = audio_feat.shape[0]
    return audio_feat.new_empty((n, 512)), audio_feat.new_empty((n, 64))


@custom_op("sdfa::regress", mutates_args=(), device_types="cuda")
def regress(z: Tensor, speaker_id: Tensor, model: str) -> Tensor:
    return _model(model).regress(z, speaker_id)[1]


@regress.register_fake
def _(z, speaker_id, model):
    return z.new_empty((z.shape[0], _model(model).out_dim))


@custom_op("sdfa::regress_into", mutates_args=("out",), device_types="cuda")
def regress_into(z: Tensor, speaker_id: Tensor, out: Tensor, model: str) -> None:
    """sdfa::regress writing into caller-owned rows (n, out_dim) -- the staging buffers of the pinned-output pipeline."""
    _model(model).regress(z, speaker_id, out=out, check_ids=False)


@regress_into.register_fake
def _(z, speaker_id, out, model):
    return None


@custom_op("sdfa::regress_coef", mutates_args=(), device_types="cuda")
def regress_coef(z: Tensor, speaker_id: Tensor, model: str) -> Tensor:
    return _model(model).regress(z, speaker_id, want_coef=True, want_out=False)[0]


@regress_coef.register_fake
def _(z, speaker_id, model):
    return z.new_empty((z.shape[0], _model(model).coef_dim))


class TraceableSpeechDrivenAnimation(torch.nn.Module):
    """SpeechDrivenAnimation.forward (speech_anime/model/model.py:28-45) as a graph of sdfa ops: what jit_trace traces.
    forward(audio_feat (N,64,128,3), speaker_id (N,)) -> ((scale (N,1,9976,6), rotat (N,1,9976,3)), z (N,1,512)) for the
    dgrad head, (pred (N,1,15069), z) for the offsets head."""

    def __init__(self, model_key, head):
        super().__init__()
        self.model_key, self.head = str(model_key), head

    def forward(self, audio_feat, speaker_id):
        z, _ = torch.ops.sdfa.encoder(audio_feat, self.model_key)
        out = torch.ops.sdfa.regress(z, speaker_id, self.model_key)
        n = audio_feat.shape[0]
        z_audio = z.view(n, 1, 512)
        if self.head == "dgrad":
            tri = out.view(n, 1, -1, 9)
            return (tri[..., :6], tri[..., 6:]), z_audio
        return out.view(n, 1, -1), z_audio

"""Checkpoint -> library tensors.

Follows the reference's load path:
  * legacy key renames + pop("hamm")        speech_anime/api.py:170-197
  * strict state_dict semantics             saber/trainer/manager/checkpoints.py:22-33
  * weight-norm folded before inference     saber/trainer/manager/device_mover.py:26-31
    (torch weight_norm dim=0: w = g * v / ||v||, norm over all dims but 0)
BatchNorm folding and every device layout are the native library's business.
"""
import numpy as np

_RENAMES = (
    ("_ext_batch_norm", "_ext_post_bn"),
    ("audio_encoder.layers.0", "_model._audio_encoder._layers.1"),
    ("audio_encoder.layers.1", "_model._audio_encoder._layers.2"),
    ("audio_encoder.layers.2", "_model._audio_encoder._layers.3"),
    ("audio_encoder.layers.3", "_model._audio_encoder._layers.4"),
    ("audio_encoder.layers.4", "_model._audio_encoder._layers.5"),
    ("audio_encoder.layers.5", "_model._audio_encoder._layers.6"),
    ("time_aggregator.layers.0", "_model._audio_encoder._layers.9"),
    ("time_aggregator.layers.1", "_model._audio_encoder._layers.10"),
    ("anime_decoder.layers.", "_model._output_module._layers."),
    ("anime_decoder.layers_scale", "_model._output_module._scale_layers"),
    ("anime_decoder.layers_rotat", "_model._output_module._rotat_layers"),
    ("anime_decoder.proj_scale", "_model._output_module._scale_pca"),
    ("anime_decoder.proj_rotat", "_model._output_module._rotat_pca"),
)


def ckpt_backward_compatible_preprocess(ckpt):
    """Same contract as speech_anime/api.py:170-197: rewrites ckpt["state"] in place and returns ckpt."""
    new_state = {}
    for k, v in ckpt["state"].items():
        nk = k
        for old, new in _RENAMES:
            nk = nk.replace(old, new)
        new_state[nk] = v
    new_state.pop("hamm")          # KeyError for a non-legacy checkpoint, as in the reference
    ckpt["state"] = new_state
    return ckpt


def _np(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def head_of(state):
    return "dgrad" if any("_scale_pca" in k for k in state) else "offsets"


def fold_state_dict(state):
    """{reference name: array} (weight_g/weight_v or plain weight) -> {name without '_model.': float32 array}."""
    out = {}
    state = {k: v for k, v in state.items()}
    for k in sorted(state):
        name = k[len("_model."):] if k.startswith("_model.") else k
        if name.endswith("num_batches_tracked") or name.endswith(".weight_g"):
            continue
        if name.endswith(".weight_v"):
            v = _np(state[k]).astype(np.float32)
            g = _np(state[k[:-1] + "g"]).astype(np.float32)
            norm = np.sqrt((v.astype(np.float64) ** 2).reshape(v.shape[0], -1).sum(1)).astype(np.float32)
            w = v * (g.reshape(-1) / norm).reshape((-1,) + (1,) * (v.ndim - 1))
            out[name[:-2]] = np.ascontiguousarray(w, dtype=np.float32)
        else:
            out[name] = np.ascontiguousarray(_np(state[k]), dtype=np.float32)
    return out

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


def expected_shapes(head):
    """{folded tensor name: shape} of the chosen head -- SURVEY.md App. A.4 (speech_anime/config/model/{dgrad,offsets}.py).
    This is what `load_state_dict(strict=True)` enforces in the reference (saber/trainer/manager/checkpoints.py:22-33)."""
    e, o = "_audio_encoder._layers.", "_output_module."
    t = {}
    for idx, (co, ci, kf) in ((1, (32, 3, 3)), (3, (64, 32, 3)), (5, (64, 64, 1))):
        t[f"{e}{idx}.weight"] = (co, ci, kf, 1)
        t[f"{e}{idx}.bias"] = (co,)
        for k in ("weight", "bias", "running_mean", "running_var"):
            t[f"{e}{idx}._ext_post_bn.{k}"] = (co,)
    for suf in ("", "_reverse"):
        t[f"{e}6._lstm.weight_ih_l0{suf}"] = (512, 64)
        t[f"{e}6._lstm.weight_hh_l0{suf}"] = (512, 128)
        t[f"{e}6._lstm.bias_ih_l0{suf}"] = (512,)
        t[f"{e}6._lstm.bias_hh_l0{suf}"] = (512,)
        t[f"{e}9.weight_ih_l0{suf}"] = (1024, 256)
        t[f"{e}9.weight_hh_l0{suf}"] = (1024, 256)
        t[f"{e}9.weight_ih_l1{suf}"] = (1024, 512)
        t[f"{e}9.weight_hh_l1{suf}"] = (1024, 256)
    t[f"{e}6._proj.weight"], t[f"{e}6._proj.bias"] = (256, 8192), (256,)
    t[f"{e}10._conv_query.weight"] = (512, 512, 3)
    t[f"{e}10.proj_key.weight"] = t[f"{e}10.proj_qry.weight"] = (128, 512)
    t[f"{e}10.v.weight"], t[f"{e}10.b"] = (1, 128), (1, 1, 128)

    def fc(key, p, k):
        t[f"{o}{key}.weight"], t[f"{o}{key}.bias"] = (p, k), (p,)
    fc("_layers.0", 512, 520)
    if head == "dgrad":
        for br, nco, rows in (("_scale", 85, 59856), ("_rotat", 180, 29928)):
            fc(f"{br}_layers.0", 512, 520); fc(f"{br}_layers.1", 256, 512); fc(f"{br}_layers.2", nco, 256)
            t[f"{o}{br}_pca.compT"], t[f"{o}{br}_pca.means"] = (rows, nco), (rows,)
    else:
        fc("_layers.1", 256, 512); fc("_layers.2", 59, 256)
        t[f"{o}_pca.compT"], t[f"{o}_pca.means"] = (15069, 59), (15069,)
    return t


def check_strict(folded, head):
    """Missing / unexpected keys and shape mismatches raise, with torch's load_state_dict wording."""
    want = expected_shapes(head)
    missing = sorted(k for k in want if k not in folded)
    unexpected = sorted(k for k in folded if k not in want)
    bad = [f"size mismatch for {k}: copying a param with shape {tuple(folded[k].shape)} from checkpoint, the shape in current model is {want[k]}."
           for k in sorted(want) if k in folded and tuple(folded[k].shape) != tuple(want[k])]
    msgs = []
    if unexpected:
        msgs.append("Unexpected key(s) in state_dict: " + ", ".join(f'"{k}"' for k in unexpected) + ".")
    if missing:
        msgs.append("Missing key(s) in state_dict: " + ", ".join(f'"{k}"' for k in missing) + ".")
    msgs += bad
    if msgs:
        raise RuntimeError("Error(s) in loading state_dict for SpeechDrivenAnimation:\n\t" + "\n\t".join(msgs))


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

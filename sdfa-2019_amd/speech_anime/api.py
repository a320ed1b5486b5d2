"""speech_anime.api -- evaluate_model / ckpt_backward_compatible_preprocess (speech_anime/api.py:78-133,170-197)."""
import os

import torch

from sdfa_amd.weights import ckpt_backward_compatible_preprocess  # noqa: F401  (same name as the reference)
from .hparams import configure, HP


def _load_checkpoint(path):
    """saber/trainer/manager/checkpoints.py:10-33: torch.load on CPU, legacy-key preprocess, strict state_dict."""
    assert os.path.exists(path), f"Failed to find checkpoint: {path}"
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if "hamm" in ckpt["state"]:                       # legacy layout (the shipped pretrained checkpoint)
        ckpt = ckpt_backward_compatible_preprocess(ckpt)
    return ckpt


def build_model(hparams, state_dict=None):
    from .model import SaberSpeechDrivenAnimation
    model = SaberSpeechDrivenAnimation(hparams, trainset=None, validset=None, load_pca=False)
    if state_dict is not None:
        model.load_state_dict(state_dict)
    return model


def evaluate_model(args):
    args = args if isinstance(args, dict) else vars(args)
    hparams = configure(args)
    if hparams.eval_input is not None:                                  # api.py:83-87
        rec = [hparams.eval_input]
        if hparams.eval_spk_cond is not None:
            rec.append(f"speaker={hparams.eval_spk_cond}")
        hparams.trainer.evaluate.set_key("test", [rec])
    if hparams.get("load_from") is None:
        raise ValueError("--load_from <checkpoint> is required for evaluation")
    if args.get("template_mesh"):                                       # tools/config.py:75-85
        from . import viewer
        viewer.set_template_mesh(args["template_mesh"], args.get("mesh_constraints"), args.get("mesh_tricorres"))
    ckpt = _load_checkpoint(os.path.expanduser(hparams.load_from))
    model = build_model(hparams, ckpt["state"])
    model.current_epoch = ckpt.get("epoch", 0)
    return model.evaluate(hparams.trainer.evaluate, experiment=None, in_trainer=False,
                          overwrite_video=args.get("overwrite_video", False),
                          export_mesh_frames=args.get("export_mesh_frames", False),
                          output_dir=args.get("output_dir") or os.path.join(hparams.get("log_dir") or ".", "evaluate_videos"))


def train_model(args):
    raise NotImplementedError("training is outside the MI355X inference path (DESIGN.md, out of scope)")


def jit_trace(args):
    raise NotImplementedError("torch.jit tracing does not apply: the model is a HIP library behind a C ABI; "
                              "the traced I/O contract (audio_feat (1,64,128,3), speaker_id (1,)) is SpeechDrivenAnimation.forward")

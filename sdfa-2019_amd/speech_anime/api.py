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


def shard_from_env(env=None):
    """(rank, world) of a torch.distributed.run launch, (0, 1) outside one."""
    env = os.environ if env is None else env
    return int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1"))


def rank_device(env, device_count):
    """The device string of this rank's GPU under one process per GPU: cuda:<LOCAL_RANK mod devices> (several ranks share a card
    only when a node is rehearsed on fewer GPUs than ranks); None without a GPU."""
    if device_count <= 0:
        return None
    return f"cuda:{int(env.get('LOCAL_RANK', env.get('RANK', '0'))) % device_count}"


def build_model(hparams, state_dict=None):
    from .model import SaberSpeechDrivenAnimation
    model = SaberSpeechDrivenAnimation(hparams, trainset=None, validset=None, load_pca=False)
    if state_dict is not None:
        model.load_state_dict(state_dict)
    return model


def evaluate_model(args):
    from_cli = not isinstance(args, dict)          # python -m speech_anime evaluate: only the files are wanted
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
    # one process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE): THIS entry point -- the CLI / torchrun one --
    # derives the rank's device and its utterance shard from the environment and hands both down explicitly; the library call
    # model.evaluate() never reads the environment
    shard = shard_from_env()
    if shard[1] > 1:
        dev = rank_device(os.environ, torch.cuda.device_count())
        if dev is not None:
            hparams.set_key("device", dev)                             # Engine + front end of this rank live on its own GPU
            torch.cuda.set_device(torch.device(dev))
    ckpt = _load_checkpoint(os.path.expanduser(hparams.load_from))
    model = build_model(hparams, ckpt["state"])
    model.current_epoch = ckpt.get("epoch", 0)
    return model.evaluate(hparams.trainer.evaluate, experiment=None, in_trainer=False, shard=shard if shard[1] > 1 else None,
                          overwrite_video=args.get("overwrite_video", False),
                          export_mesh_frames=args.get("export_mesh_frames", False), keep_results=not from_cli,
                          output_dir=args.get("output_dir") or os.path.join(hparams.get("log_dir") or ".", "evaluate_videos"))


def train_model(args):
    raise NotImplementedError("training is outside the MI355X inference path (DESIGN.md, out of scope)")


def jit_trace(args):
    """speech_anime/api.py:136-167: trace the inner model on the example inputs that fix its I/O contract --
    audio_feat = rand(1, 64, 128, 3), speaker_id = zeros(1, long) -- and save `<traced_dump_path>-gpu.zip`.

    The traced graph is two custom operators, torch.ops.sdfa.encoder and torch.ops.sdfa.regress (sdfa_amd/ops.py), keyed by
    the checkpoint path: `import sdfa_amd.ops; torch.jit.load(...)` in a fresh process loads the checkpoint on first use.
    The reference also saves a `-cpu.zip`; this build has no CPU path, so that file is not written (said on stdout)."""
    import sdfa_amd.ops as ops
    args = args if isinstance(args, dict) else vars(args)
    hparams = configure(args)
    assert args.get("traced_dump_path") is not None
    if hparams.get("load_from") is None:
        raise ValueError("--load_from <checkpoint> is required for tracing")
    path = os.path.abspath(os.path.expanduser(hparams.load_from))
    if not os.path.exists(path) and hparams.get("log_dir"):                # api.py:143-150: also looked up under <log_dir>/checkpoints
        for cand in (os.path.join(hparams.log_dir, "checkpoints", hparams.load_from), os.path.join(hparams.log_dir, "checkpoints", hparams.load_from + ".ckpt")):
            if os.path.exists(cand):
                path = os.path.abspath(cand)
    ckpt = _load_checkpoint(path)
    hparams.set_key("model_key", path)
    model = build_model(hparams, ckpt["state"])
    ops.register_model(path, model._model._engine, own=True)       # the traced module outlives `model`: the registry owns the engine
    head = "dgrad" if hparams.model.face_data_type == "dgrad_3d" else "offsets"
    mod = ops.TraceableSpeechDrivenAnimation(path, head).eval()
    dev = model._model._engine.device
    audio_feat = torch.rand(1, 64, 128, 3, device=dev)
    speaker_id = torch.zeros(1, dtype=torch.long, device=dev)
    traced = torch.jit.trace(mod, (audio_feat, speaker_id))
    out = os.path.splitext(args["traced_dump_path"])[0] + "-gpu.zip"
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    traced.save(out)
    print(f"[speech_anime] traced model -> {out}; no -cpu.zip: the MI355X build has no CPU implementation")
    return traced

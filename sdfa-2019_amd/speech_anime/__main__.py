"""`python3 -m speech_anime evaluate ...` -- accepts the flags evaluate.sh passes (speech_anime/__main__.py:8-49)."""
import argparse

from .api import train_model, evaluate_model, jit_trace

# (flag, type or None for a store_true switch, default)
_FLAGS = [
    ("tag", str, None), ("template_mesh", str, None), ("mesh_constraints", str, None), ("mesh_tricorres", str, None),
    ("matplotlib_use", str, "Agg"), ("log_dir", str, None), ("load_from", str, None), ("custom_hparams", str, None),
    ("ensembling_ms", int, None), ("save_video", None, False),
    ("eval_input", str, None), ("eval_spk_cond", str, None), ("export_mesh_frames", None, False), ("output_dir", str, None),
    ("grid_w", int, 512), ("grid_h", int, 512), ("font_size", int, 24),
    ("overwrite_video", None, False), ("with_title", None, False), ("draw_truth", None, False),
    ("draw_align", None, False), ("draw_latent", None, False), ("traced_dump_path", str, None),
]


def _parser():
    ap = argparse.ArgumentParser(prog="speech_anime")
    ap.add_argument("mode", choices=["train", "evaluate", "trace"])
    for name, typ, default in _FLAGS:
        if typ is None:
            ap.add_argument("--" + name, action="store_true")
        else:
            ap.add_argument("--" + name, type=typ, default=default)
    return ap


if __name__ == "__main__":
    ns = _parser().parse_args()
    {"train": train_model, "evaluate": evaluate_model, "trace": jit_trace}[ns.mode](ns)

"""`python3 -m speech_anime evaluate ...` with the reference's flags (speech_anime/__main__.py:8-49)."""
import argparse

from .api import train_model, evaluate_model, jit_trace

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("mode", type=str, choices=["train", "evaluate", "trace"])
    parser.add_argument("--tag", type=str)
    parser.add_argument("--template_mesh", type=str)
    parser.add_argument("--mesh_constraints", type=str)
    parser.add_argument("--mesh_tricorres", type=str)
    parser.add_argument("--matplotlib_use", type=str, default="Agg")
    parser.add_argument("--log_dir", type=str)
    parser.add_argument("--load_from", type=str)
    parser.add_argument("--custom_hparams", type=str)
    parser.add_argument("--ensembling_ms", type=int)
    parser.add_argument("--save_video", action="store_true")
    parser.add_argument("--eval_input", type=str)
    parser.add_argument("--eval_spk_cond", type=str)
    parser.add_argument("--export_mesh_frames", action="store_true")
    parser.add_argument("--output_dir", type=str)
    parser.add_argument("--grid_w", type=int, default=512)
    parser.add_argument("--grid_h", type=int, default=512)
    parser.add_argument("--font_size", type=int, default=24)
    parser.add_argument("--overwrite_video", action="store_true")
    parser.add_argument("--with_title", action="store_true")
    parser.add_argument("--draw_truth", action="store_true")
    parser.add_argument("--draw_align", action="store_true")
    parser.add_argument("--draw_latent", action="store_true")
    parser.add_argument("--traced_dump_path", type=str)
    args = parser.parse_args()
    {"train": train_model, "evaluate": evaluate_model, "trace": jit_trace}[args.mode](args)

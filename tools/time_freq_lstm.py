"""freq_lstm stage time of one 8192-frame (or argv[1]-frame) encoder call per "freq_lstm_shape" option, alternating, product library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdfa-2019_amd"))
import torch
from sdfa_amd import synth, _lib
from sdfa_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
shapes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 5]
eng = Engine(synth.make_state_dict("dgrad", 1234), max_frames=8192)
x = torch.rand((n, 64, 128, 3), device="cuda")
res = {k: [] for k in shapes}
for rep in range(6):
    for k in shapes:
        _lib.set_option("freq_lstm_shape", k)
        eng.profile(True)
        eng.encoder(x, want_align=False); torch.cuda.synchronize()
        res[k].append(eng.profile_ms("freq_lstm")); eng.profile(False)
_lib.set_option("freq_lstm_shape", 0)
for k in shapes:
    print(f"freq_lstm_shape={k}: {n} frames, freq_lstm stage ms per call:", " ".join(f"{v:.2f}" for v in res[k][1:]))

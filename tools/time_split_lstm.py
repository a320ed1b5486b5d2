"""Time-LSTM stage time per encoder call for small batches: time_lstm_kernel<1> (split 1) against time_lstm_split_kernel (32) and time_lstm_split16_kernel (0 = by size: 16-frame tiles up to 1,024 frames),
with the sc1 hand-off (mode 0) and the release / acquire hand-off (mode 3).  Usage (GPU box): python tools/time_split_lstm.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sdfa-2019_amd"))
import numpy as np
import torch
from sdfa_amd import synth, _lib
from sdfa_amd.engine import Engine

eng = Engine(synth.make_state_dict("dgrad", 1234), max_frames=4096)
rs = np.random.RandomState(0)
print(f"{'frames':>7s} {'split':>6s} {'handoff':>8s} {'lstm0+lstm1 ms':>15s} {'whole encoder ms':>17s}")
for n in (156, 636, 1000, 1900):
    x = torch.from_numpy(rs.uniform(0, 1, (n, 64, 128, 3)).astype(np.float32)).cuda()
    for split, mode in ((1, 0), (32, 0), (32, 3), (0, 0), (0, 3)):
        _lib.set_option("time_lstm_split", split)
        _lib.set_option("time_lstm_handoff", mode)
        for _ in range(3):
            eng.encoder(x, want_align=False)
        torch.cuda.synchronize()
        eng.profile(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            eng.encoder(x, want_align=False)
        e1.record()
        torch.cuda.synchronize()
        ms = (eng.profile_ms("lstm0") + eng.profile_ms("lstm1")) / reps
        eng.profile(False)
        print(f"{n:7d} {split:6d} {mode:8d} {ms:15.3f} {e0.elapsed_time(e1) / reps:17.3f}", flush=True)
_lib.set_option("time_lstm_split", 0)
_lib.set_option("time_lstm_handoff", 0)

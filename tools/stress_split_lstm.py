"""Stress test of the cooperating-workgroup time-LSTM kernels (time_lstm_split16_kernel / time_lstm_split_kernel): thousands of encoder
calls at alternating single-clip sizes, every result compared BITWISE with the single-workgroup form; a stale or torn hand-off of h
between two workgroups -- the one failure mode a per-step exchange through global memory has -- would show as a mismatch.
Usage (GPU box): python tools/stress_split_lstm.py [iterations]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sdfa-2019_amd"))
import numpy as np
import torch
from sdfa_amd import synth, _lib
from sdfa_amd.engine import Engine

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
eng = Engine(synth.make_state_dict("dgrad", 1234), max_frames=4096)
rs = np.random.RandomState(1)
sizes = (156, 636, 1000, 1500, 2000, 40)
xs = {n: torch.from_numpy(rs.uniform(0, 1, (n, 64, 128, 3)).astype(np.float32)).cuda() for n in sizes}
_lib.set_option("time_lstm_split", 1)
ref = {n: eng.encoder(x)[0].clone() for n, x in xs.items()}
_lib.set_option("time_lstm_split", 0)
noise = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")      # 256 MiB of L2 / Infinity-Cache pollution between calls
bad, timeouts = {n: 0 for n in sizes}, 0
for handoff in (0, 3):
    _lib.set_option("time_lstm_handoff", handoff)
    for i in range(iters):
        n = sizes[i % len(sizes)]
        if i % 3 == 0:
            noise.fill_(i & 255)
        z = eng.encoder(xs[n])[0]
        if not torch.equal(z, ref[n]):
            bad[n] += 1
        if i % 97 == 0:
            timeouts = eng.time_lstm_repairs()
    print(f"handoff {handoff}: {iters} calls, mismatches per size {bad}, timeouts {timeouts}", flush=True)
_lib.set_option("time_lstm_handoff", 0)
print("STRESS", "OK" if not any(bad.values()) and timeouts == 0 else "FAILED")

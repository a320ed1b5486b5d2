"""Stress of the spectral-stream front end's producer / consumer hand-offs (csrc/frontend.hip: mel_stream_kernel<., true>): random clip sets,
random segment geometries and random frame tables, each call compared bit for bit with the two-kernel form (share map + mel_columns +
gather_features) and the status word read (bounded waits that expired: must stay 0).  Usage (GPU box): python tools/stress_frontend.py [calls]"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sdfa-2019_amd"))
import numpy as np
import torch
from sdfa_amd import synth, _lib
from sdfa_amd.engine import FrontendOnly, frame_index

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
fe = FrontendOnly()
rs = np.random.RandomState(20251005)
bad = status = 0
t0 = time.time()
frames = 0
for it in range(calls):
    sr = int(rs.choice([8000, 16000]))
    hop = sr // 125
    n_clips = int(rs.randint(1, 9))
    clips = [synth.make_pcm(int(rs.randint(0, 1000)), int(rs.uniform(0.57, 6.0) * sr), str(rs.choice(["uniform", "speechlike", "sweep"]))) for _ in range(n_clips)]
    tables = None
    mode = int(rs.randint(0, 4))
    if mode == 1:      # irregular tables: random multiples of the hop (long chains, shifts 1..70) or unaligned starts
        tables = []
        for c in clips:
            n = int(rs.randint(1, 200))
            step = rs.choice([hop, 3 * hop, 25 * hop, 62 * hop, 63 * hop, hop + 1, 7])
            starts = np.cumsum(rs.choice([step, 25 * hop, hop], n)).astype(np.int64) - int(rs.randint(0, 5000))
            tables.append((starts, np.zeros(n, np.int64)))
    opts = {"frontend_stream_block": int(rs.choice([0, 0, 12, 48, 100, 144, 192, 256])), "frontend_stream_slots": int(rs.choice([0, 0, 1, 5, 12, 24]))}
    try:
        _lib.set_option("frontend_two_kernel", 1)
        ref, _, counts = fe.mel_frontend(clips, sr, tables=tables)
        ref = ref.clone()
        _lib.set_option("frontend_two_kernel", 0)
        for k, v in opts.items():
            _lib.set_option(k, v)
        got, _, _ = fe.mel_frontend(clips, sr, tables=tables)
        st = fe.frontend_status()
    finally:
        for k in ("frontend_two_kernel", "frontend_stream_block", "frontend_stream_slots"):
            _lib.set_option(k, 0)
    frames += int(sum(counts))
    if st:
        status += 1
    if not torch.equal(got, ref):
        bad += 1
        print(f"MISMATCH call {it}: sr {sr}, {n_clips} clips, mode {mode}, {opts}, status {st}", flush=True)
    if it % 50 == 49:
        print(f"{it + 1} calls, {frames} frames, {bad} mismatches, {status} calls with expired waits, {time.time() - t0:.0f} s", flush=True)
print(f"stress_frontend: {calls} calls, {frames} frames, {bad} mismatches, {status} calls with expired waits")
sys.exit(1 if (bad or status) else 0)

"""Timeline of one Engine.forward_host call of the 32 x 10 s batch (column sharing, pinned output): when does each piece's compute end and
each piece's device -> host copy end?  (HIP events on the two streams.)  Usage (GPU box): python tools/timeline_batch.py [piece ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sdfa-2019_amd"))
import numpy as np
import torch
from sdfa_amd import synth
from sdfa_amd.engine import Engine, HostPipeline

sr = 16000
eng = Engine(synth.make_state_dict("dgrad", 1234))
clips = [synth.make_pcm(c, 10 * sr) for c in range(32)]
feat, _, counts = eng.mel_frontend(clips, sr)
table = eng.last_frame_table
n = feat.shape[0]
spk = torch.full((n,), 2, dtype=torch.int64, device="cuda")
out = torch.empty((n, eng.out_dim), dtype=torch.float32, pin_memory=True)
from sdfa_amd.engine import piece_schedule
def parse(a):      # "4096" = uniform pieces, "0" = the default schedule, "4736,4224,..." = an explicit schedule
    return [int(x) for x in a.split(",")] if "," in a else int(a)


for piece in [parse(a) for a in sys.argv[1:]] or [8192, 4096, 2048, 0]:
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.forward_host(feat, spk, out=out, table=table, piece=piece or None)
        dt = (time.perf_counter() - t0) * 1e3
    sizes = piece if isinstance(piece, list) else (piece_schedule(n, eng.max_frames) if not piece else [min(piece, n - f) for f in range(0, n, piece)])
    # instrumented repeat: events after each piece's regress (compute stream) and after each copy
    host = eng._host
    marks = []
    orig = HostPipeline.run

    start = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    f0 = 0
    evs = []
    for m in sizes:
        f1 = f0 + m
        eng.forward_host(feat[f0:f1], spk[f0:f1], out=out[f0:f1], table=(table[0][f0:f1], table[1][f0:f1], table[2]), piece=m, wait=False)
        ec = torch.cuda.Event(enable_timing=True); ec.record()
        ed = torch.cuda.Event(enable_timing=True); ed.record(host.copy_stream)
        evs.append((f1 - f0, ec, ed))
        f0 = f1
    eng.host_wait(); torch.cuda.synchronize()
    print(f"piece {piece if not isinstance(piece, list) else 'schedule'}: whole call {dt:.1f} ms = {n / dt:.1f} k frames/s; per piece (frames, compute end ms, copy end ms):")
    print("   ", [(m, round(start.elapsed_time(ec), 1), round(start.elapsed_time(ed), 1)) for m, ec, ed in evs])

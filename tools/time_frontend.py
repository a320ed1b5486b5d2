"""Same-process A/B of the spectral-gather front end on the headline workload (32 x 10 s @ 16 kHz = 20,352 frames): every combination
of the library switches given on the command line, alternating, HIP-event time of the whole stage (share map + mel columns +
gather) per call.  Usage (GPU box): python tools/time_frontend.py [reps] [name=v0,v1 ...]
e.g. python tools/time_frontend.py 20 mel_fft_radix4=0,1 frontend_t_major=0,1"""
import itertools
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sdfa-2019_amd"))
import numpy as np
import torch
from sdfa_amd import synth, _lib
from sdfa_amd.engine import FrontendOnly, frame_index

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
switches = [(kv.split("=")[0], [int(v) for v in kv.split("=")[1].split(",")]) for kv in sys.argv[2:]]
sr, C, L = 16000, 32, 160000
fe = FrontendOnly()
dev = fe.device
starts = frame_index(L, sr)[0]
F = len(starts) * C
pcm = torch.from_numpy(np.concatenate([synth.make_pcm(c, L) for c in range(C)])).to(dev)
clip_len = torch.full((C,), L, dtype=torch.int64, device=dev)
clip_off = torch.cumsum(clip_len, 0) - clip_len
frame_clip = torch.repeat_interleave(torch.arange(C, dtype=torch.int32, device=dev), len(starts))
frame_start = torch.from_numpy(np.tile(starts, C)).to(dev)
feat = torch.empty((F, 64, 128, 3), dtype=torch.float32, device=dev)
combos = list(itertools.product(*[vals for _, vals in switches])) or [()]
ref, times = None, {c: [] for c in combos}
for rep in range(reps + 2):
    for combo in combos:
        for (name, _), v in zip(switches, combo):
            _lib.set_option(name, v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fe.mel_frontend_device(pcm, clip_off, clip_len, frame_clip, frame_start, sr, out=feat)
        e1.record()
        torch.cuda.synchronize()
        if rep >= 2:
            times[combo].append(e0.elapsed_time(e1))
        if rep == 0:
            if ref is None:
                ref = feat.clone()
            print(dict(zip([n for n, _ in switches], combo)), "max|diff| vs first combination:", float((feat - ref).abs().max()), flush=True)
for (name, _) in switches:
    _lib.set_option(name, 0)
for combo in combos:
    t = np.asarray(times[combo])
    print(dict(zip([n for n, _ in switches], combo)), f"stage {t.mean():.4f} ms (min {t.min():.4f}, {len(t)} calls) = "
          f"{F * 99.4e3 / (t.mean() * 1e-3) / 1e9:.0f} GB/s algorithmic", flush=True)

#!/usr/bin/env python3
"""Fixtures for the small host-side next rows, from the reference itself (build container only):
saber.audio.rms.normalize (saber/data/audio/rms.py:45-78) and saber.stream.seek (saber/data/stream/stream.py:20-46)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "sdfa-2019_amd"))
import ref_import  # noqa: E402
from sdfa_amd import synth  # noqa: E402

ref_import.install_stubs()
import saber  # noqa: E402

out = {}
for i, (kind, scale) in enumerate((("uniform", 0.1), ("speechlike", 1.0), ("sweep", 2.0))):
    x = (synth.make_pcm(40 + i, 12000, kind) * scale).astype(np.float32)
    out[f"rms_in_{i}"] = x
    out[f"rms_out_{i}"] = np.asarray(saber.audio.rms.normalize(x, -24.5))
ts = [-117, -100, -83, -67, -50, -33, -17, 0, 17, 33]
seq = np.random.RandomState(2).normal(0, 1, (len(ts), 7)).astype(np.float32)
queries = np.asarray([-200.0, -117.0, -110.5, -100.0, -99.999, -58.5, 0.0, 16.9, 33.0, 40.0, 16.666666666666668])
out["seek_ts"] = np.asarray(ts, np.int64); out["seek_seq"] = seq; out["seek_q"] = queries
out["seek_out"] = np.stack([np.asarray(saber.stream.seek(float(q), ts, seq), np.float64) for q in queries])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "host_rows.npz"), **out)
print({k: v.shape for k, v in out.items()})

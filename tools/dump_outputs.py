"""Writes encoder / regressor outputs of a fixed random batch to an .npz, for bitwise comparison of two builds of the library:
    SDFA_HIP_LIB=/path/old.so python tools/dump_outputs.py /tmp/a.npz; python tools/dump_outputs.py /tmp/b.npz; python tools/dump_outputs.py --compare /tmp/a.npz /tmp/b.npz"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdfa-2019_amd"))
import numpy as np
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        print(k, "bit-identical" if np.array_equal(a[k], b[k]) else f"DIFFERS max {float(np.abs(a[k] - b[k]).max()):.3e}")
    sys.exit(0)
import torch
from sdfa_amd import synth
from sdfa_amd.engine import Engine
res = {}
for head in ("dgrad", "offsets"):
    for prec in ("fp32", "bf16x3"):
        eng = Engine(synth.make_state_dict(head, 1234), max_frames=2048, precision=prec)
        torch.manual_seed(5)
        x = torch.rand((700, 64, 128, 3), device="cuda")
        z, al = eng.encoder(x)
        coef, out = eng.regress(z, torch.arange(700) % 8, want_coef=True)
        res[f"{head}_{prec}_z"] = z.cpu().numpy(); res[f"{head}_{prec}_align"] = al.cpu().numpy()
        res[f"{head}_{prec}_coef"] = coef.cpu().numpy(); res[f"{head}_{prec}_out_stride13"] = out[:, ::13].cpu().numpy()
np.savez(sys.argv[1], **res)

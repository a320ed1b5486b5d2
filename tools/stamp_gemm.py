"""Reads the in-kernel s_memtime stamps of the diagnostic GEMM build (make -C sdfa-2019_amd/csrc STAMPS=1)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SDFA_HIP_LIB"] = os.path.join(ROOT, "sdfa-2019_amd", "sdfa_amd", "libsdfa_hip_stamps.so")
sys.path.insert(0, os.path.join(ROOT, "sdfa-2019_amd"))
import numpy as np, torch
from sdfa_amd import synth, _lib
from sdfa_amd.engine import Engine
lib = C.CDLL(os.environ["SDFA_HIP_LIB"])
lib.sdfa_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
eng = Engine(synth.make_state_dict("dgrad", 1234), max_frames=8192)
x = torch.rand((8192, 64, 128, 3), device="cuda")
out = (C.c_ulonglong * 8)()
for rep in range(2):
    z, _ = eng.encoder(x, want_align=False); torch.cuda.synchronize()
    lib.sdfa_debug_read_stamps(out, 1)
v = [int(o) for o in out]
n = v[4]
if n:
    print("per even stage, per wave, cycles (K >= 2048 GEMMs only):")
    for name, val in zip(("issue next-next loads", "ds_read + 64 MFMA", "wait loads + ds_write", "barrier"), v[:4]):
        print(f"  {name:24s} {val / n:9.0f}")
    print(f"  total {sum(v[:4]) / n:9.0f}   (64 MFMAs alone = 4096)")

lib.sdfa_debug_read_lstm_stamps.argtypes = [C.c_void_p, C.c_int]
lib.sdfa_debug_read_lstm_stamps(out, 0)
v = [int(o) for o in out]
n = v[6]
print("freq_lstm_kernel, per step, per wave, cycles (768 MFMAs alone = 49152):")
names = ("x prefetch issue + bias init + first loads", "k-loop: issue 4 weight loads + 2 ds_read (x24)", "k-loop: 32-MFMA blocks (x24)", "barriers (2)", "epilogue total")
for name, val in zip(names, v[:5]):
    print(f"  {name:50s} {val / n:9.0f}")
print(f"     of which cell math {v[5] / n:9.0f}   LDS writes {v[7] / n:9.0f}   global stores {(v[4] - v[5] - v[7]) / n:9.0f}")
print(f"  total {sum(v[:5]) / n:9.0f}")

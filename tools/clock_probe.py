"""What clock does the frequency-LSTM kernel really run at?  Diagnostic build `make -C sdfa-2019_amd/csrc EXP=CLOCKPROBE`:
every workgroup of freq_lstm_kernel adds its s_memtime (shader clock) and s_memrealtime (100 MHz) deltas to a counter."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SDFA_HIP_LIB"] = os.path.join(ROOT, "sdfa-2019_amd", "sdfa_amd", "libsdfa_hip_exp.so")
sys.path.insert(0, os.path.join(ROOT, "sdfa-2019_amd"))
import torch
from sdfa_amd import synth
from sdfa_amd.engine import Engine
lib = C.CDLL(os.environ["SDFA_HIP_LIB"])
lib.sdfa_debug_read_clockprobe.argtypes = [C.c_void_p, C.c_int]
eng = Engine(synth.make_state_dict("dgrad", 1234), max_frames=8192)
x = torch.rand((8192, 64, 128, 3), device="cuda")
out = (C.c_ulonglong * 3)()
for rep in range(6):                      # repeated launches: does the clock sag as the run goes on?
    z, _ = eng.encoder(x, want_align=False); torch.cuda.synchronize()
    lib.sdfa_debug_read_clockprobe(out, 1)
    core, real, n = (int(o) for o in out)
    mhz = core / real * 100.0
    print(f"launch {rep}: {n} workgroups, mean {core / n:,.0f} shader cycles = {real / n * 10:,.0f} ns each -> shader clock {mhz:,.0f} MHz; "
          f"MFMA issue cycles per workgroup wave 32 steps x 768 x 64 = {32 * 768 * 64:,} -> pipe busy {2 * 32 * 768 * 64 / (core / n):.1%} at that clock")

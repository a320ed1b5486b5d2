"""Phase times of time_lstm_kernel<2> (the BiLSTM recurrence at 64 frames per workgroup, two waves per SIMD) from in-kernel s_memtime
stamps (diagnostic build: make -C sdfa-2019_amd/csrc STAMPS=1).  Per wave and step (steps 1..63), shader cycles: the K loop (1,024 MFMAs
of 64 cycles per wave, two waves per SIMD: 131,072 cycles of matrix work per SIMD and step), the cell update with the next step's
input-projection requests, and the step barrier.  The stamps use the sub-phase slots freq_lstm_v2_kernel uses: run with the default
frequency-LSTM form (freq_lstm_v3_kernel), which leaves them alone."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SDFA_HIP_LIB"] = os.path.join(ROOT, "sdfa-2019_amd", "sdfa_amd", "libsdfa_hip_stamps.so")
sys.path.insert(0, os.path.join(ROOT, "sdfa-2019_amd"))
import torch
from sdfa_amd import synth
from sdfa_amd.engine import Engine
lib = C.CDLL(os.environ["SDFA_HIP_LIB"])
lib.sdfa_debug_read_lstm_sub.argtypes = [C.c_void_p, C.c_int]
PREC = os.environ.get("SDFA_PREC", "fp32")      # bf16x3 / bf16x6 / bf16: time_lstm_bf16_kernel<2, TERMS> carries the same stamps (round 5)
MFMA_CYCLES = {"fp32": 1024 * 64, "bf16x3": 16 * 24 * 32, "bf16": 16 * 8 * 32}.get(PREC)      # per wave and step (bf16x6 runs 32-frame tiles: not stamped)
eng = Engine(synth.make_state_dict("dgrad", 1234), max_frames=8192, precision=PREC)
x = torch.rand((8192, 64, 128, 3), device="cuda")
sub = (C.c_ulonglong * 4)()
for rep in range(2):
    lib.sdfa_debug_read_lstm_sub(sub, 1)
    eng.profile(True)
    eng.encoder(x, want_align=False); torch.cuda.synchronize()
    ms = [eng.profile_ms("lstm0"), eng.profile_ms("lstm1")]; eng.profile(False)
lib.sdfa_debug_read_lstm_sub(sub, 0)
k, cell, bar, n = (int(v) for v in sub)
print(f"[{PREC}] time LSTM, 64 frames per workgroup, 8192 frames (256 workgroups, one per CU): layers {ms[0]:.2f} / {ms[1]:.2f} ms; per wave and step, shader cycles (both layers, {n} wave-steps):")
for name, v in (("K loop (this wave's 1,024 MFMAs = 65,536 cycles; its SIMD partner's run in between)", k), ("cell update + next step's input-projection requests", cell), ("step barrier", bar)):
    print(f"  {name:92s} {v / n:9.0f}")
tot = (k + cell + bar) / n
print(f"  total {tot:9.0f}  -> matrix work of the SIMD's two waves {2 * MFMA_CYCLES:,} / {tot:.0f} = {2 * MFMA_CYCLES / tot:.3f} of the step; shader clock ~ {64 * tot / (sum(ms) / 2 * 1e-3) / 1e9:.2f} GHz")

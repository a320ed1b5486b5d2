"""Phase times of freq_lstm_v2_kernel (or, with SDFA_FORM=8, freq_lstm_v3_kernel) from in-kernel s_memtime stamps (diagnostic build: make -C sdfa-2019_amd/csrc STAMPS=1).
SDFA_LONE=1 in the environment launches one workgroup per CU (no partner on the SIMDs): what a step costs by itself."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SDFA_HIP_LIB"] = os.path.join(ROOT, "sdfa-2019_amd", "sdfa_amd", "libsdfa_hip_stamps.so")
sys.path.insert(0, os.path.join(ROOT, "sdfa-2019_amd"))
import torch
from sdfa_amd import synth
from sdfa_amd.engine import Engine
lib = C.CDLL(os.environ["SDFA_HIP_LIB"])
lib.sdfa_debug_read_lstm_stamps.argtypes = [C.c_void_p, C.c_int]
from sdfa_amd import _lib
if os.environ.get("SDFA_FORM"):
    _lib.set_option("freq_lstm_shape", int(os.environ["SDFA_FORM"]))      # e.g. SDFA_FORM=8: freq_lstm_v3_kernel (no barrier-1 / sub-phase stamps there)
eng = Engine(synth.make_state_dict("dgrad", 1234), max_frames=8192)
if os.environ.get("SDFA_PREC"):
    eng.set_precision(os.environ["SDFA_PREC"])       # e.g. SDFA_PREC=bf16x6: freq_lstm_bf16p_v3_kernel (48 MFMAs of 32 cycles per k-step, 12 k-steps = 18,432)
x = torch.rand((8192, 64, 128, 3), device="cuda")
out = (C.c_ulonglong * 8)()
span = (C.c_ulonglong * 4)()
xcd = (C.c_ulonglong * 32)()
sub = (C.c_ulonglong * 4)()
lib.sdfa_debug_read_lstm_sub.argtypes = [C.c_void_p, C.c_int]
lib.sdfa_debug_read_lstm_xcd.argtypes = [C.c_void_p, C.c_int]
lib.sdfa_debug_read_lstm_span.argtypes = [C.c_void_p, C.c_int]
for rep in range(2):
    lib.sdfa_debug_read_lstm_stamps(out, 1)
    lib.sdfa_debug_read_lstm_span(span, 1)
    lib.sdfa_debug_read_lstm_xcd(xcd, 1)
    lib.sdfa_debug_read_lstm_sub(sub, 1)
    eng.profile(True)
    z, _ = eng.encoder(x, want_align=False); torch.cuda.synchronize()
    ms = eng.profile_ms("freq_lstm"); eng.profile(False)
lib.sdfa_debug_read_lstm_stamps(out, 0)
v = [int(o) for o in out]
n = v[6]
print(f"frequency LSTM, form {os.environ.get('SDFA_FORM', '3')} ({'ONE workgroup per CU' if os.environ.get('SDFA_LONE') else 'two workgroups per CU'}): launch {ms:.2f} ms; per step (steps 1..31) and wave, shader cycles; 768 MFMAs alone = 49,152")
for name, val in zip(("accumulator init + first operand reads", "K loop (24 k-blocks)", "barrier 1 (all waves done with K loop)", "x DMA + cell update + stores", "barrier 2 (h, x in LDS)"), v[:5]):
    print(f"  {name:44s} {val / n:9.0f}")
print(f"  total {sum(v[:5]) / n:9.0f}")
lib.sdfa_debug_read_lstm_sub(sub, 0)
e0, e1, e2 = (int(x) / n for x in sub[:3])
print(f"  inside the cell-update phase: x DMA issue {e0:.0f}, columns 0-31 (4 quads: math, LDS write, store) {e1:.0f}, columns 32-63 {e2:.0f}, final wait (DMA landed, LDS writes done) {v[3] / n - e0 - e1 - e2:.0f}")
lib.sdfa_debug_read_lstm_span(span, 0)
t0, t1, life_ticks, life_cyc = (int(x) for x in span)
if t1 <= t0 or not life_ticks:      # kernels without lifetime stamps (freq_lstm_v3_kernel, freq_lstm_bf16p_v3_kernel): the clock from the launch time instead
    tiles_per_cu = 8192 * 2 / 256
    print(f"  shader clock over the launch ~ {tiles_per_cu * 32 * sum(v[:5]) / n / (ms * 1e-3) / 1e9:.2f} GHz ({tiles_per_cu:.0f} tiles per CU x 32 steps x the total above / launch time; tile prologues not counted)")
    sys.exit(0)
slots = 256 * (1 if os.environ.get("SDFA_LONE") else 2)
n_wg = 8192 * 2
print(f"  launch span (first workgroup start -> last end) {(t1 - t0) / 100e3:.2f} ms; sum of workgroup lifetimes / {slots} slots = {life_ticks / slots / 100e3:.2f} ms "
      f"-> slot occupancy {life_ticks / slots / (t1 - t0):.3f}; mean lifetime {life_ticks / n_wg / 100:.0f} us; shader clock inside workgroups {life_cyc / life_ticks * 100 / 1e3:.3f} GHz")
lib.sdfa_debug_read_lstm_xcd(xcd, 0)
for x in range(8):
    end, life, cnt, last_start = (int(v) for v in xcd[4 * x: 4 * x + 4])
    print(f"  XCD {x}: {cnt} workgroups, mean lifetime {life / max(cnt, 1) / 100:.0f} us, last start at {(last_start - t0) / 100e3:.2f} ms, last end at {(end - t0) / 100e3:.2f} ms")

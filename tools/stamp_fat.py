"""Per-stage cycle accounting of gemm_fat_kernel from in-kernel s_memtime stamps (diagnostic build: make -C sdfa-2019_amd/csrc STAMPS=1).
One wave per SIMD: the phases simply add up.  Only the K <= 512 launches (BiLSTM input projections gx0 / gx1) are summed."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SDFA_HIP_LIB", os.path.join(ROOT, "sdfa-2019_amd", "sdfa_amd", "libsdfa_hip_stamps.so"))
sys.path.insert(0, os.path.join(ROOT, "sdfa-2019_amd"))
import torch
from sdfa_amd import synth
from sdfa_amd.engine import Engine
lib = C.CDLL(os.environ["SDFA_HIP_LIB"])
lib.sdfa_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
eng = Engine(synth.make_state_dict("dgrad", 1234), max_frames=8192)
x = torch.rand((8192, 64, 128, 3), device="cuda")
out = (C.c_ulonglong * 8)()
for rep in range(2):
    lib.sdfa_debug_read_stamps(out, 1)
    eng.encoder(x, want_align=False); torch.cuda.synchronize()
lib.sdfa_debug_read_stamps(out, 0)
v = [int(o) for o in out]
n, tiles = v[4], v[6]
print("gemm_fat_kernel, gx0 + gx1 launches, per stage and wave, shader cycles (256 MFMAs alone = 16,384):")
for name, val, ideal in zip(("k-block 0 + 16 LDS-DMA requests", "k-blocks 1, 2", "wait for own DMA + barrier", "k-block 3 + 32 LDS reads of the next stage"), v[:4], (4096, 8192, 0, 4096)):
    print(f"  {name:44s} {val / n:8.0f}   (MFMAs alone {ideal})")
print(f"  total {sum(v[:4]) / n:8.0f};   epilogue per tile {v[5] / tiles:8.0f}   ({n / tiles:.1f} stages per tile)")

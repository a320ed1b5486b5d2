"""Stress test of the one-pass attention kernels (attn_fused_f32_kernel, attn_key_score_*_kernel): their tiles arrive by LDS-DMA requests
the compiler does not know of, waited for with hand-counted vmcnt -- a count too large would read a tile that has not landed, and only
when memory is slow.  So: encoder calls at sizes that take every work-unit shape, while a SECOND stream saturates HBM with copies (uneven
load: bursts and pauses), every result compared BITWISE with the first result of its (mode, size) -- and, in exact fp32, with the
two-launch form, which runs the same recurrence without the ring's timing.
Usage (GPU box): python tools/stress_attention.py [iterations]"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sdfa-2019_amd"))
import numpy as np
import torch
from sdfa_amd import synth, _lib
from sdfa_amd.engine import Engine

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
eng = Engine(synth.make_state_dict("dgrad", 1234), max_frames=8192)
rs = np.random.RandomState(7)
sizes = (72, 156, 636, 1500, 3968, 8192)
xs = {n: torch.from_numpy(rs.uniform(0, 1, (n, 64, 128, 3)).astype(np.float32)).cuda() for n in sizes}

stop = False


def hammer():
    """another stream: 1 GiB device-to-device copies in bursts (HBM saturated), then a pause"""
    s = torch.cuda.Stream()
    a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    b = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    k = 0
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(1 + k % 7):
                b.copy_(a, non_blocking=True)
            s.synchronize()
            time.sleep(0.001 * (k % 4))
            k += 1


th = threading.Thread(target=hammer, daemon=True)
th.start()
bad = 0
for mode in ("fp32", "bf16x3_attention", "bf16x6"):
    eng.set_precision(mode)
    ref = {}
    for n in sizes:
        z, a = eng.encoder(xs[n])
        ref[n] = (z.clone(), a.clone())
        if mode in ("fp32", "bf16x6") and n >= 3968:      # the one-launch form: bitwise the two-launch form
            _lib.set_option("attn_unfused", 2)
            z2, a2 = eng.encoder(xs[n])
            _lib.set_option("attn_unfused", 0)
            if not (torch.equal(z2, z) and torch.equal(a2, a)):
                bad += 1
                print(f"{mode} n={n}: one launch != two launches", flush=True)
    t0 = time.time()
    miss = {n: 0 for n in sizes}
    for i in range(iters):
        n = sizes[i % len(sizes)]
        z, a = eng.encoder(xs[n])
        if not (torch.equal(z, ref[n][0]) and torch.equal(a, ref[n][1])):
            miss[n] += 1
    bad += sum(miss.values())
    print(f"{mode}: {iters} calls under a copy stream, mismatches per size {miss} ({time.time() - t0:.1f} s)", flush=True)
stop = True
th.join(timeout=10)
eng.set_precision("fp32")
print("STRESS", "OK" if bad == 0 else "FAILED")

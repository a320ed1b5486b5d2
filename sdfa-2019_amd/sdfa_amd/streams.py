"""Which HIP streams really run side by side?

HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and the work of two streams that landed on
the SAME queue runs one after the other.  Measured on MI355X (profiles/r03_overlap*.txt, tools/ab_hostio.sh): whether a
freshly created copy stream -- or the stream torch's RCCL process group communicates on -- shares the compute stream's queue
depends on the number of queues, on stream priorities and on how many streams were created before, and when it does, a
device -> host copy / an all-gather that was meant to hide under the next piece's kernels adds its full length to the step
(231 -> 260 ms per 20,352-frame step for the copies).  So nothing here trusts a setting: a candidate stream is PROBED -- a few
milliseconds of matrix products on the compute stream, a small transfer on the candidate, device timestamps of both -- and the
first candidate whose transfer finishes while the kernels are still running is kept.
"""
import time

import torch

# The kernels that keep the compute stream busy during a probe are handed in by the caller as `busy()` -- a few milliseconds of the
# library's own MFMA-bound work (Engine.encoder on a block of zero features): compute-bound, so a transfer that really runs beside
# it is not slowed by it (a fill kernel would compete with a copy for HBM and hide the overlap).


def engine_busy(engine, frames=512):
    """busy() for the probes below: the encoder on `frames` all-zero frames (about 2.5 ms on MI355X for 512)."""
    feat = torch.zeros((frames, 64, 128, 3), dtype=torch.float32, device=engine.device)

    def busy(n=1):
        for _ in range(n):
            engine.encoder(feat, want_align=False)
    busy()                                         # warm: workspace, kernel attributes
    return busy


def copy_overlaps(compute, candidate, device, busy, nbytes=128 << 20):
    """Does a device -> host copy on `candidate` run UNDER kernels enqueued before it on `compute`?  The copy is as large as a real
    piece's would be slow (128 MiB: about 2.3 ms over PCIe; large copies go through a blit kernel, small ones do not), the
    kernels last several times longer.  Returns (overlaps, kernels ms, copy-end ms after the kernels' start)."""
    src = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    dst = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    with torch.cuda.stream(candidate):
        dst.copy_(src, non_blocking=True)              # warm: first-touch of the pinned block
    torch.cuda.synchronize(device)
    k0, k1, c1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(compute):
        k0.record()
        busy(6)
        k1.record()
    with torch.cuda.stream(candidate):
        dst.copy_(src, non_blocking=True)
        c1.record()
    torch.cuda.synchronize(device)
    t_k, t_c = k0.elapsed_time(k1), k0.elapsed_time(c1)
    return t_c < 0.7 * t_k, round(t_k, 3), round(t_c, 3)


def pick_copy_stream(device, busy, compute=None, tries=8):
    """A stream whose copies overlap `compute`'s kernels (default: the current stream), found by probing up to `tries` new
    streams of alternating priority; the last candidate if none passes (the copies are then merely not hidden).
    Returns (stream, overlaps, probe log)."""
    device = torch.device(device)
    compute = compute or torch.cuda.current_stream(device)
    cand, log = None, []
    for i in range(tries):
        cand = torch.cuda.Stream(device=device, priority=-1 if i % 2 == 0 else 0)
        ok, t_k, t_c = copy_overlaps(compute, cand, device, busy)
        log.append({"priority": -1 if i % 2 == 0 else 0, "kernels_ms": t_k, "copy_end_ms": t_c, "overlaps": ok})
        if ok:
            return cand, True, log
    return cand, False, log


def collective_overlaps(compute, device, busy, group=None):
    """True if an asynchronous all_gather_into_tensor (on the process group's own stream) runs UNDER kernels that are enqueued on
    `compute` right after it -- the order of sdfa_amd/dist.py: chunk i's gather is issued, then chunk i+1's kernels.  On a shared
    hardware queue the kernels would wait for the gather.  Three timings, host clock between device synchronisations: the gather
    alone, the kernels alone, both; overlapping means `both` is well below the sum.  Every rank of the group calls this together;
    the verdict is the same on all of them.  Returns (overlaps, dict of the three times in ms)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)

    def timed(fn):
        torch.cuda.synchronize(device)
        dist.barrier(group=group)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        with torch.cuda.stream(compute):
            fn()
        torch.cuda.synchronize(device)
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return float(t.item()) * 1e3

    nbytes = 64 << 20
    while True:                                   # a gather long enough to be told from noise (>= 1.5 ms), at most 2 GiB per rank
        inp = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        out = torch.empty(nbytes * world, dtype=torch.uint8, device=device)

        def gather():
            dist.all_gather_into_tensor(out, inp, group=group, async_op=True).wait()
        timed(gather)                             # warm (communicator set-up, buffer registration)
        t_coll = timed(gather)
        if t_coll >= 1.5 or nbytes >= (2 << 30) // max(1, world // 2):
            break
        nbytes *= 2
        del inp, out
    t_one = timed(lambda: busy(1))
    n_busy = max(2, int(3 * t_coll / max(t_one, 0.1)))
    t_busy = timed(lambda: busy(n_busy))

    def both():
        work = dist.all_gather_into_tensor(out, inp, group=group, async_op=True)
        busy(n_busy)
        work.wait()
    t_both = timed(both)
    return t_both < t_busy + 0.5 * t_coll, {"gather_ms": round(t_coll, 3), "kernels_ms": round(t_busy, 3), "both_ms": round(t_both, 3),
                                              "gather_bytes_per_rank": nbytes}


def pick_compute_stream_for_collectives(device, busy, group=None, tries=6):
    """The stream to run the kernels on so that the process group's asynchronous collectives overlap them: the current stream if it
    already does, otherwise the first freshly created stream that does.  Returns (stream, overlaps, probe log); every rank calls it
    together."""
    device = torch.device(device)
    cur = torch.cuda.current_stream(device)
    ok, times = collective_overlaps(cur, device, busy, group)
    log = [dict(times, stream="current", overlaps=ok)]
    if ok:
        return cur, True, log
    for i in range(tries):
        cand = torch.cuda.Stream(device=device)
        ok, times = collective_overlaps(cand, device, busy, group)
        log.append(dict(times, stream=f"new#{i}", overlaps=ok))
        if ok:
            return cand, True, log
    return cur, False, log

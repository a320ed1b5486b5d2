#!/usr/bin/env python3
"""Headline benchmark: animation frames/s for batches of 10 s @ 16 kHz clips (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one pass of the whole hot path over one batch of synthetic PCM that is already resident
in HBM: frame windows -> mel/delta front end -> conv + freq-LSTM + BiLSTM + attention encoder -> MLPs + PCA
expansion -> interleaved dgrad rows (F, 89784) in HBM.  Per GPU the batch is 32 clips x 10 s @ 16 kHz
(20,352 animation frames); with N GPUs every rank takes its own 32 clips (weak scaling) and the per-frame
dgrad rows of all ranks are reassembled on every rank inside the timed region: either an RCCL all-gather of the rows
issued chunk by chunk so that it overlaps the next chunk's compute (`--gather dgrad`), or an RCCL all-gather of the PCA
coefficients followed by a local, bit-identical re-expansion of the peers' rows (`--gather expand`); the default `auto`
times both before the measurement and uses the faster (reported in `config`).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the fused frequency-LSTM recurrence freq_lstm_v3_kernel,
fp32 MFMA), measured live with HIP events on the launch stream; `cpu_baseline` times the CPU port of the reference
path on the reference's own operator library (oracle/torch_oracle.py, torch CPU) on a bounded sample on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

T_PROCESS_START = time.perf_counter()        # the wall-clock budget (--budget-s) counts from here: interpreter start, imports included

# dmabuf IPC (the only kind this pool's driver supports) must be selected BEFORE anything initialises HSA: torch.cuda.set_device()
# below already does, so this cannot wait until the process group is created (VERDICT r2 / ADVICE r2)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# HIP maps streams onto a few hardware queues (4 by default) and two streams that share a queue run one after the other.  Measured
# on MI355X (profiles/r03_overlap_env.txt): with the defaults the stream torch's RCCL process group communicates on lands on the
# SAME queue as the default stream -- the per-chunk asynchronous all-gather then ran 0 % under the compute kernels; with more
# hardware queues, or a high-priority communication stream, 85 %.  Both must be in the environment before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "sdfa-2019_amd"))

FLOP_FREQ_LSTM_PER_FRAME = 64 * 32 * 2 * 512 * (64 + 128) * 2      # 268.4 + 536.9 MFLOP
# The kernel skips the recurrent k-blocks of step 0 (h_-1 = 0, csrc/lstm.hip): of 32 steps x (64 + 128) K it issues 32 x 192 - 128
FREQ_LSTM_EXECUTED_FRACTION = (32 * 192 - 128) / (32 * 192)          # = 752 / 768 of the algorithmic FLOPs
FLOP_MODEL_PER_FRAME = 1.514e9
FLOP_ATTENTION_PER_FRAME = 10.2e6                                    # SURVEY 8(a) a10 / DESIGN section 4: key projection 8.39 + query Conv1d 1.57 + query projection + tail
FRONTEND_BYTES_PER_FRAME = 4 * 16000 / 60 + 64 * 128 * 3 * 4         # new PCM + feature write = 99.4 KB (at 16 kHz; 8 kHz: 98.8 KB)
PEAK_FP32_MFMA_TFLOPS = 157.3                                        # MI355X_MICROARCH.md chip table
PEAK_HBM_GBPS = 8000.0


class Budget:
    """Wall-clock budget of one `bench.py` run.  The driver stops the command at a limit it does not tell us (600 s so far); the headline
    is printed as soon as it exists (`emit`), and every OPTIONAL leg is admitted only while the time it is expected to take still fits --
    a leg that does not fit is skipped and listed in the line's `legs_skipped`, never started and then lost together with the line.
    `reserve_s` is kept back for the legs that the bench contract requires (cpu_baseline) so that the optional ones cannot eat it."""

    def __init__(self, total_s, clock=time.perf_counter, t0=None, reserve_s=0.0):
        self.total, self.clock, self.t0, self.reserve = float(total_s), clock, (clock() if t0 is None else t0), float(reserve_s)
        self.skipped, self.seconds = [], {}

    def elapsed(self):
        return self.clock() - self.t0

    def left(self):
        return self.total - self.elapsed()

    def admit(self, name, estimate_s, required=False):
        """True when leg `name` (expected to take estimate_s) may start; otherwise it is recorded as skipped with the figures that decided it."""
        need = float(estimate_s) + (0.0 if required else self.reserve)
        if self.left() >= need:
            return True
        self.skipped.append({"leg": name, "estimate_s": round(float(estimate_s), 1), "left_s": round(self.left(), 1),
                             "reserved_s": 0.0 if required else round(self.reserve, 1)})
        return False

    def run(self, name, estimate_s, fn, required=False):
        """fn() if the leg is admitted (its wall time recorded in `seconds`), else None."""
        if not self.admit(name, estimate_s, required):
            return None
        t = self.clock()
        try:
            return fn()
        finally:
            self.seconds[name] = round(self.clock() - t, 2)

    def release_reserve(self):
        self.reserve = 0.0

    def report(self):
        return {"budget_s": self.total if self.total != float("inf") else None, "elapsed_s": round(self.elapsed(), 1), "legs_s": dict(self.seconds)}


def emit(res, final):
    """One JSON line on stdout, flushed at once.  The headline goes out TWICE: right after the timed steps (`"partial": true`, the optional
    blocks and cpu_baseline still null) and again, complete, as the LAST line -- a run that is stopped in between has still said its number."""
    d = dict(res)
    d["partial"] = not final
    print(json.dumps(d), flush=True)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips-per-gpu", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--sample-rate", type=int, default=16000)
    ap.add_argument("--chunk", type=int, default=8192, help="frames per encoder launch group")
    ap.add_argument("--gather", choices=["auto", "dgrad", "expand", "coef", "none", "direct", "mesh"], default="auto",
                    help="how the output rows of all ranks are reassembled on every rank at N > 1: dgrad = RCCL all-gather of the rows, "
                         "chunk by chunk; expand = RCCL all-gather of the PCA coefficients (1 KB instead of 359 KB per frame), every "
                         "rank expands its peers' frames with the regressor's own last stage (bit-identical rows); auto (default) = "
                         "both are timed for two untimed-region steps before the measurement and the faster one is used; "
                         "coef = only the coefficients are gathered (no rows); direct = the regressor stores each row straight into every peer's gathered "
                         "buffer over xGMI (one-shot direct all-gather, no collective); mesh = dgrad -> seek -> mesh on the device, "
                         "then all-gather of the vertices (60 KB instead of 359 KB per frame)")
    ap.add_argument("--mesh-stage", action="store_true", help="run the seek + mesh post-path stage inside the timed step (implied by --gather mesh)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-column-sharing", action="store_true", help="skip the second, column-sharing measurement")
    ap.add_argument("--head", choices=["dgrad", "offsets"], default="dgrad", help="offsets = BASELINE configs[4] (VOCASET-style vertex offsets)")
    ap.add_argument("--ragged-seconds", default=None, metavar="LO,HI",
                    help="clip lengths uniform in [LO, HI] s instead of --seconds (a VOCASET-like stream of sentences)")
    ap.add_argument("--frontend", choices=["gather", "direct"], default="gather",
                    help="gather = each distinct STFT column once + per-frame gather (sdfa_mel_frontend_gather); direct = one FFT per window column")
    ap.add_argument("--precision", choices=["fp32", "bf16_attention", "bf16x3", "bf16", "bf16x3_attention", "bf16x6"], default="fp32",
                    help="matrix instruction of the headline run (fp32 = the reference's arithmetic; the others are BASELINE configs[3] modes)")
    ap.add_argument("--no-mixed-precision", action="store_true", help="skip the extra split-bf16 (configs[3]) measurement")
    ap.add_argument("--opt", action="append", default=[], help="library tuning switch name=value (A/B runs; sdfa_debug_set_option is THREAD-LOCAL: "
                                                               "it applies to launches made from this, the main, thread)")
    ap.add_argument("--force-gather", action="store_true",
                    help="with --gpus 1: create the process group anyway (world size 1) and run the chosen --gather mode through the real "
                         "backend -- RCCL all_gather_into_tensor(async_op=True) per chunk + Work.wait() -- to rehearse the N > 1 exchange "
                         "and its overlap with the persistent kernels on one GPU")
    ap.add_argument("--reserve-cus", type=int, default=0, help="CUs the persistent kernels leave to other streams (sdfa_model_set_reserved_cus)")
    ap.add_argument("--no-host-io", action="store_true", help="skip the PCIe-inclusive twin (H2D of PCM + D2H of the rows inside the step)")
    ap.add_argument("--no-surface", action="store_true", help="skip the speech_anime surface block (generate_animation frames/s)")
    ap.add_argument("--compute-stream", choices=["default", "side"], default="default",
                    help="side = run everything on a freshly created HIP stream instead of the default stream (HIP maps streams onto a "
                         "few hardware queues; two streams on one queue serialise -- see DESIGN.md section 5)")
    ap.add_argument("--cpu-sample-seconds", type=float, default=10.0)
    ap.add_argument("--budget-s", type=float, default=240.0,
                    help="wall-clock budget of the whole command, counted from interpreter start: an optional leg (column sharing, the configs[3] "
                         "modes, the PCIe twin, the surface block) that is not expected to fit is skipped and listed in `legs_skipped`; the "
                         "headline line is printed right after the timed steps and again, complete, as the last line")
    ap.add_argument("--cpu-baseline-cap-s", type=float, default=25.0, help="upper bound on the whole cpu_baseline leg (thread scan + sample)")
    ap.add_argument("--all-legs", action="store_true",
                    help="at N > 1 also run the optional legs (column sharing, the configs[3] precision modes); by default a multi-GPU "
                         "run measures the headline only -- nothing optional may stand between an 8-GPU slot and its line")
    ap.add_argument("--inject-failure", default=None, metavar="LEG",
                    help="test hook: raise inside the optional leg LEG (column_sharing | bf16x3 | bf16x3_column_sharing | bf16x3_attention | "
                         "bf16x6) to prove that no optional leg can cost the headline line")
    return ap.parse_args(argv)


def launcher_argv(n, argv, port, python=None):
    """The command `python bench.py --gpus N ...` turns itself into when it was started WITHOUT a launcher: one rank per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve), the same arguments."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def self_launch(a, argv):
    """`python3 bench.py --gpus N` (N > 1) with no WORLD_SIZE in the environment: start the N ranks as a CHILD process (never an exec,
    and before this process has imported torch or touched the GPU), pass the ranks' output through, and exit with the child's return
    code.  A failed child is a non-zero exit, never a silent fall-back to N = 1; a child that ends 0 without the JSON line is an error too."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = launcher_argv(a.gpus, argv, port)
    env = dict(os.environ, SDFA_BENCH_LAUNCHER="self")
    print(f"[bench] --gpus {a.gpus} without a launcher: starting {' '.join(cmd[:10])} ...", file=sys.stderr, flush=True)
    import signal
    # the ranks get a session (process group) of their own: whatever ends this parent -- the driver's SIGTERM at its limit, Ctrl-C, an
    # exception in the relay below -- the whole group is stopped on the way out, so no orphan rank keeps a card busy
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1, start_new_session=True)

    def stop_ranks(grace=5.0):
        if child.poll() is not None:
            return
        for sig, wait in ((signal.SIGTERM, grace), (signal.SIGKILL, 2.0)):
            try:
                os.killpg(child.pid, sig)                       # child.pid IS the group id (start_new_session)
            except (ProcessLookupError, PermissionError):
                pass
            try:
                child.wait(timeout=wait)
                return
            except subprocess.TimeoutExpired:
                continue

    def on_signal(signum, frame):
        stop_ranks()
        raise SystemExit(128 + signum)
    previous = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    seen = False
    try:
        for line in child.stdout:                               # rank 0's lines (and anything else the ranks print) as they arrive
            sys.stdout.write(line)
            sys.stdout.flush()
            if line.startswith('{"metric"'):
                seen = True
        rc = child.wait()
    finally:
        stop_ranks()
        for sg, h in previous.items():
            signal.signal(sg, h)
    if rc == 0 and not seen:
        print("[bench] the ranks ended without printing the result line", file=sys.stderr, flush=True)
        rc = 1
    raise SystemExit(rc)


def thread_scan(probe_fn, counts, set_threads, clock=time.perf_counter, slack=1.5, cap_s=8.0):
    """Which thread count runs `probe_fn` fastest: ONE timed pass per count (ascending), stopping as soon as a count is `slack` x slower than
    the best so far (torch's small-operator LSTM path only gets slower once it is oversubscribed) or `cap_s` is spent.  Returns
    (best count, [(count, seconds), ...])."""
    t_begin, best, log = clock(), None, []
    for nt in counts:
        set_threads(nt)
        t0 = clock()
        probe_fn()
        dt = clock() - t0
        log.append((nt, round(dt, 3)))
        if best is None or dt < best[0]:
            best = (dt, nt)
        elif dt > slack * best[0]:
            break
        if clock() - t_begin > cap_s:
            break
    return best[1], log


def cpu_baseline(sr, seconds, eng, state_dict, head, cap_s=25.0):
    """The reference path on the reference's own operator library (oracle/torch_oracle.py: torch CPU stft / conv2d /
    LSTM / linear, pinned to the reference fixtures) timed on this box's host cores on a bounded sample: clips of
    `seconds` s until about 10 s of CPU work are done (at most 8 clips), the WHOLE leg -- thread scan included -- held to
    `cap_s` seconds.  Also the dgrad parity number (first clip)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_oracle as TO
    from sdfa_amd import synth
    t_leg = time.perf_counter()
    orc = TO.TorchOracle(state_dict, head)
    n = int(seconds * sr)
    # thread count: the box may expose more logical CPUs than this job's share, and torch's small-operator LSTM path slows down when
    # oversubscribed -- a short scan on a 1 s clip, bounded (thread_scan): the 256-thread probe of round 5 alone cost minutes
    probe = synth.make_pcm(99, sr)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    counts = sorted({min(c, ncpu) for c in (8, 16, 32, 64)})
    torch.set_num_threads(counts[0])
    TO.generate_animation(orc, probe, sr, 2, batch=100)                       # untimed: warms the operator library and its pools
    cores, scan = thread_scan(lambda: TO.generate_animation(orc, probe, sr, 2, batch=100), counts, torch.set_num_threads, cap_s=0.3 * cap_s)
    torch.set_num_threads(cores)
    frames, spent, clips, first = 0, 0.0, 0, None
    while clips < 8 and (clips == 0 or (spent < 10.0 and (time.perf_counter() - t_leg) + spent / clips < cap_s)):
        pcm = synth.make_pcm(clips, n)
        t0 = time.perf_counter()
        ts, ref = TO.generate_animation(orc, pcm, sr, 2, batch=100)           # batches of 100 frames like model.py:450-461
        spent += time.perf_counter() - t0
        frames += len(ref); clips += 1
        if first is None:
            first = (pcm, ts, ref.reshape(len(ref), -1))
    pcm, ts, ref = first

    def gpu_err(precision):
        eng.set_precision(precision)
        feat, tslists, counts = eng.mel_frontend([pcm], sr)
        out, *_ = eng.forward(feat, torch.full((feat.shape[0],), 2, dtype=torch.int64))
        return float(np.abs(out.cpu().numpy() - ref).max()), bool(tslists[0] == list(ts))

    return dict(value=round(frames / spent, 2), unit="frames/s", cores=int(cores), nproc=int(os.cpu_count() or 0),
                affinity_cpus=len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, kind="port",
                sample=f"{clips} clip(s) x {seconds:g} s @ {sr} Hz = {frames} frames in {spent:.1f} s; oracle/torch_oracle.py = the reference "
                       f"path on torch {torch.__version__} CPU operators (fp32, batches of 100 frames); cores = the {cores} threads used "
                       f"(fastest of a bounded scan {scan}), nproc = logical CPUs the host exposes",
                leg_seconds=round(time.perf_counter() - t_leg, 1)), gpu_err


def surface_block(sd, head, sr, dev):
    """speech_anime.model.SaberSpeechDrivenAnimation.generate_animation (speech_anime/model/model.py:333-420 + 428-489) timed as a
    caller sees it: wall clock around the call, so frame enumeration, H2D of the PCM, every kernel, the device -> host copy of the
    rows into pinned memory and the Python around them are all inside.  Three workloads: one 10 s clip and one 2 s clip per call
    (BASELINE configs[0] is the 2 s case), and 32 x 10 s through generate_animation_batch (one launch group)."""
    import torch
    from sdfa_amd import synth
    from speech_anime.hparams import configure
    from speech_anime.api import build_model
    from speech_anime.datasets import DatasetSlidingWindow
    hp = configure(dict(mode="evaluate", custom_hparams=head))
    hp.audio.set_key("sample_rate", sr)
    hp.set_key("device", str(dev))
    DatasetSlidingWindow.hparams = None
    model = build_model(hp, sd)
    eng = model._model._engine
    out = {"note": "wall-clock frames/s of generate_animation / generate_animation_batch calls (host work, H2D, kernels, D2H of the "
                   "rows to pinned host memory all included); speaker m1; others['inputs'] not requested unless stated"}

    def run(fn, frames, reps, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        dt = time.perf_counter() - t0
        return {"frames_per_s": round(frames * reps / dt, 1), "ms_per_call": round(dt / reps * 1e3, 3), "frames_per_call": frames, "calls": reps}

    STAGES = ("conv23", "freq_lstm", "freq_proj", "gx0", "lstm0", "gx1", "lstm1", "attn_proj", "attn", "mlp", "pca", "share_map", "share_expand")
    class Alt:
        """Two different clips of one length, taken in turn: every call is a NEW signal (the one-entry signal -> z cache of
        generate_animation -- the reference's feature cache, model.py:364-367 -- must not flatter these numbers)."""
        def __init__(self, seconds):
            self.clips, self.i = [synth.make_pcm(k, int(seconds * sr)) for k in (0, 1)], 0

        def __call__(self):
            self.i += 1
            return self.clips[self.i & 1]

    for name, seconds, reps in (("1x10s", 10.0, 10), ("1x2s", 2.0, 20)):
        pcm = Alt(seconds)
        frames = len(model.generate_animation(pcm(), "m1", 0, 0, want_inputs=False)[0])
        r = run(lambda: model.generate_animation(pcm(), "m1", 0, 0, want_inputs=False), frames, reps)
        eng.profile(True)
        model.generate_animation(pcm(), "m1", 0, 0, want_inputs=False)
        torch.cuda.synchronize()
        st = {}
        for k in STAGES:
            try:
                st[k] = round(eng.profile_ms(k), 3)
            except Exception:
                pass
        eng.profile(False)
        r["kernel_ms_per_call"] = round(sum(st.values()), 3)
        r["stage_ms"] = st
        out[name] = r
        if name == "1x10s":
            out["1x10s_with_inputs"] = run(lambda: model.generate_animation(pcm(), "m1", 0, 0), frames, 5)
            out["1x10s_ensembling_20ms"] = run(lambda: model.generate_animation(pcm(), "m1", 0, 0, ensembling_ms=20, want_inputs=False), frames, 5)
    clips = [synth.make_pcm(c, int(10.0 * sr)) for c in range(32)]
    frames = sum(len(r[0]) for r in model.generate_animation_batch(clips, "m1"))
    out["32x10s_batch"] = run(lambda: model.generate_animation_batch(clips, "m1"), frames, 3, warm=1)
    # speaker sweep over ONE clip: the second and later speakers reuse the cached, speaker-independent encoder output (the reference
    # keeps the last signal's features for the same purpose, model.py:364-367,409-416) -- only the regressor + the copy run
    pcm = synth.make_pcm(0, int(10.0 * sr))
    names = ["m0", "m1", "m2", "m3", "f0", "f1", "f2", "f3"]
    frames = len(model.generate_animation(pcm, "m1", 0, 0, want_inputs=False)[0])
    state = {"i": 0}

    def sweep():
        state["i"] += 1
        model.generate_animation(pcm, names[state["i"] % 8], 0, 0, want_inputs=False)
    out["1x10s_speaker_sweep"] = run(sweep, frames, 16)
    out["time_lstm_repairs"] = eng.time_lstm_repairs()        # waits of the cooperating-workgroup time LSTM that expired (0 unless the device is oversubscribed)
    del model
    # BASELINE configs[4] through the surface: a VOCASET-like stream -- 80 ragged 3 - 6 s sentences @ 8 kHz, 8 speakers, offsets head --
    # through generate_animation_batch, rows delivered to pinned host memory (what evaluate() does per launch group)
    try:
        sr4 = 8000
        hp4 = configure(dict(mode="evaluate", custom_hparams="offsets"))
        hp4.audio.set_key("sample_rate", sr4)
        hp4.set_key("device", str(dev))
        DatasetSlidingWindow.hparams = None
        m4 = build_model(hp4, synth.make_state_dict("offsets", 1234))
        rs = np.random.RandomState(4242)
        clips4 = [synth.make_pcm(1000 + c, int(rs.uniform(3.0, 6.0) * sr4)) for c in range(80)]
        spk4 = [names[c % 8] for c in range(80)]
        frames4 = sum(len(r[0]) for r in m4.generate_animation_batch(clips4, spk4))
        r4 = run(lambda: m4.generate_animation_batch(clips4, spk4), frames4, 3, warm=1)
        r4["workload"] = "80 sentences, 3 - 6 s @ 8 kHz, 8 speakers, offsets head (15,069 floats per frame) -> pinned host rows (BASELINE configs[4] shape, 1 GPU)"
        out["stream_offsets"] = r4
        del m4
    except Exception as e:
        out["stream_offsets"] = {"error": repr(e)}
    DatasetSlidingWindow.hparams = None
    return out


def memory_plan(world, clips_per_gpu=32, seconds=10.0, sr=16000, head="dgrad", chunk=8192, gather="auto"):
    """Per-rank device memory (GB) of `bench.py --gpus world`, from the size functions the run itself uses (no GPU needed): what
    DESIGN.md section 5 quotes for N = 8 and tests/test_memory_plan_cpu.py holds to 288 GB.  `gather` auto builds the dgrad and the
    expand buffers IN TURN (set_mode empties the cache in between), so its peak is the larger of the two."""
    from sdfa_amd import _lib
    from sdfa_amd.engine import frame_index
    F = len(frame_index(int(seconds * sr), sr)[0]) * clips_per_gpu
    m = _lib.lib.sdfa_model_create(_lib.HEAD_DGRAD if head == "dgrad" else _lib.HEAD_OFFSETS)
    try:
        out_dim, coef_dim = int(_lib.lib.sdfa_model_out_dim(m)), int(_lib.lib.sdfa_model_coef_dim(m))
        ws = int(_lib.check(_lib.lib.sdfa_workspace_bytes(m, min(chunk, F))))
    finally:
        _lib.lib.sdfa_model_destroy(m)
    fe_ws = int(_lib.check(_lib.lib.sdfa_frontend_workspace_bytes(F)))
    n_chunks = (F + chunk - 1) // chunk
    rows = F * out_dim * 4
    parts = {
        "weights_packed": 0.5e9,                                   # 70 MB of fp32 weights + PCA basis, plus the packed fp32 / bf16 operand forms (upper bound)
        "pcm": clips_per_gpu * int(seconds * sr) * 4,
        "audio_feat": F * 64 * 128 * 3 * 4,
        "frontend_workspace": fe_ws,
        "encoder_workspace": ws,
        "z_and_ids": n_chunks and min(chunk, F) * (512 + 64 + 2) * 4 * 2,
    }
    modes = {
        "none": rows,                                              # N = 1: the step's output rows
        "dgrad": rows + world * rows,                              # own rows + FrameGatherer.buf (all ranks' rows)
        "expand": world * rows + 2 * world * F * coef_dim * 4,     # ExpandGatherer.buf (own rows written in place) + the coefficient gather
        "direct": 2 * world * rows,                                # DirectGatherer: two alternating gathered buffers
        "coef": rows + world * F * coef_dim * 4,                   # own rows + the gathered coefficients
        "mesh": rows + (world + 1) * F * 5148 * 3 * 4,             # own rows + own and gathered vertices (bench.py's synthetic template: 5,148 vertices, one video frame per animation frame at 60 fps)
    }
    if world == 1:
        kinds = ["none"]
    else:
        kinds = ["dgrad", "expand"] if gather == "auto" else [gather if gather in modes else "none"]
    fixed = sum(parts.values())
    plan = {k: round(v / 1e9, 3) for k, v in parts.items()}
    plan["output_and_gathered"] = {k: round(modes[k] / 1e9, 3) for k in kinds}
    plan["total_gb"] = round((fixed + max(modes[k] for k in kinds)) / 1e9, 2)
    plan["frames_per_gpu"] = F
    return plan


def _plan_or_none(world, C, a, sr):
    """memory_plan for the bench line: reporting only, so it can never cost the line."""
    if a.ragged_seconds:
        return None
    try:
        return memory_plan(world, C, a.seconds, sr, a.head, a.chunk, a.gather)
    except Exception as e:
        return {"error": repr(e)}


def _sha1(name):
    import hashlib
    try:
        return hashlib.sha1(open(os.path.join(ROOT, "sdfa-2019_amd", "csrc", name), "rb").read()).hexdigest()
    except OSError:
        return None


def traffic_from_profile(frames_per_launch):
    """HBM bytes per frequency-LSTM launch from the newest committed rocprofv3 PMC passes (profiles/r*_pmc/
    freq_lstm_traffic.json, written by profiles/pmc_summary.py: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950), scaled to this run's frames per launch.  Counters cannot be
    read inside the benchmark itself.  The JSON carries the sha1 of csrc/lstm.hip it was measured at: once the kernel file
    has changed the figure is stale and `traffic` is null (the source says why).
    Returns (traffic bytes, algorithmic bytes, source) or (None, None, reason)."""
    import glob
    try:
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc", "freq_lstm_traffic.json")))[-1]
        with open(path) as f:
            t = json.load(f)
        rel = os.path.relpath(path, ROOT)
        k = frames_per_launch / t["frames"]
        alg = round(t["algorithmic_bytes"] * k)
        if t.get("lstm_hip_sha1") != _sha1("lstm.hip"):
            return None, alg, f"{rel} was measured at another version of csrc/lstm.hip: stale, re-run tools/collect_profiles.sh"
        return round(t["traffic_bytes"] * k), alg, rel
    except Exception:
        return None, None, None


def frontend_counter_bytes():
    """Counter-based HBM bytes per frame of the spectrogram stage (profiles/r*_pmc/frontend_traffic.json), or None when absent / stale."""
    import glob
    try:
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc", "frontend_traffic.json")))[-1]
        with open(path) as f:
            t = json.load(f)
        if t.get("frontend_hip_sha1") != _sha1("frontend.hip"):
            return None, None
        return float(t["bytes_per_frame"]), os.path.relpath(path, ROOT)
    except Exception:
        return None, None


def precision_worst_case():
    """Worst max|dgrad - reference| per precision mode over the committed wide sweep (profiles/r*_precision_modes.json, written by
    tests/test_gpu_precision.py: other seeds, hotter recurrences, BatchNorm scales of both signs, the reference's 10 s fixture, full size) --
    what `mixed_precision.max_abs_dgrad_err_vs_cpu_ref` reports, together with this run's own 10 s clip, instead of the one fixture case."""
    import glob
    try:
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_precision_modes.json")))[::-1]:
            with open(path) as f:
                t = json.load(f)
            if "worst_case_dgrad" in t:
                return {k: float(v) for k, v in t["worst_case_dgrad"].items()}, os.path.relpath(path, ROOT)
    except Exception:
        pass
    return {}, None


def attention_counter_util(mode="fp32"):
    """Time-weighted rocprofv3 MfmaUtil of the attention stage in precision `mode` (profiles/r*_pmc/attention_mfma.json for fp32,
    attention_mfma_<mode>.json for a configs[3] mode; written by profiles/pmc_summary.py from the committed --pmc pass of that mode),
    or None when absent or csrc/attn.hip / csrc/gemm.hip have changed since.  Returns (whole stage %, source, the stage's MFMA kernels alone %)."""
    import glob
    try:
        name = "attention_mfma.json" if mode == "fp32" else f"attention_mfma_{mode}.json"
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc", name)))[-1]
        with open(path) as f:
            t = json.load(f)
        if t.get("attn_hip_sha1") != _sha1("attn.hip") or t.get("gemm_hip_sha1") != _sha1("gemm.hip"):
            return None, None, None
        g = t.get("mfma_util_pct_gemms")        # the stage's MFMA kernels alone (without the softmax / context tail), from round 6
        return float(t["mfma_util_pct_time_weighted"]), os.path.relpath(path, ROOT), (None if g is None else float(g))
    except Exception:
        return None, None, None


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a, argv)                                    # does not return
    import torch
    import torch.distributed as dist
    from sdfa_amd import synth, dist as sdist
    from sdfa_amd.engine import Engine, frame_index

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus must agree "
                         f"(or start `python bench.py --gpus {a.gpus}` with no launcher: it starts its own ranks)")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    if a.compute_stream == "side":
        torch.cuda.set_stream(torch.cuda.Stream(device=local_dev))
    dev = torch.device("cuda", local_dev)
    dist_on = world > 1 or a.force_gather          # the exchange runs: N > 1, or the one-GPU rehearsal of it
    if dist_on:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    # what really ran, as the process group saw it: world size from the group, one record per rank (device index, PCI bus id,
    # name, host, pid) gathered through the group itself -- a SCALE record can show "the backend saw N ranks on N devices" from the line alone
    def device_record():
        p = torch.cuda.get_device_properties(dev)
        bus = None
        if all(hasattr(p, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            bus = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
        uuid = getattr(p, "uuid", None)
        return {"rank": rank, "local_rank": local_rank, "device": local_dev, "pci_bus_id": bus, "uuid": None if uuid is None else str(uuid),
                "name": p.name, "cus": p.multi_processor_count, "host": __import__("socket").gethostname(), "pid": os.getpid()}
    world_seen, devices = 1, [device_record()]
    if dist_on:
        world_seen = dist.get_world_size()
        devices = [None] * world_seen
        dist.all_gather_object(devices, device_record())
        if world_seen != a.gpus:
            raise SystemExit(f"the process group holds {world_seen} ranks, --gpus says {a.gpus}")

    from sdfa_amd import _lib
    for kv in a.opt:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    sr = a.sample_rate
    sd = synth.make_state_dict(a.head, 1234)
    eng = Engine(sd, device=dev, max_frames=a.chunk, precision=a.precision, autotune=not any(kv.startswith("freq_lstm_shape=") for kv in a.opt))
    if a.reserve_cus:
        eng.set_reserved_cus(a.reserve_cus)
    comm_probe = None
    if dist_on and a.backend == "nccl":
        # run the kernels on a stream the process group's collectives really overlap with (probed: sdfa_amd/streams.py)
        from sdfa_amd import streams as sstreams
        cs, ok, log = sstreams.pick_compute_stream_for_collectives(dev, sstreams.engine_busy(eng))
        torch.cuda.set_stream(cs)
        comm_probe = {"overlaps": ok, "probes": log}

    # ---- this rank's clips: global clip ids [rank*C, (rank+1)*C), PCM resident in HBM before timing
    C = a.clips_per_gpu
    if a.ragged_seconds:
        lo, hi = (float(x) for x in a.ragged_seconds.split(","))
        rs = np.random.RandomState(4242 + rank)
        lengths = [int(rs.uniform(lo, hi) * sr) for _ in range(C)]
    else:
        lengths = [int(a.seconds * sr)] * C
    tables = {L: frame_index(L, sr)[0] for L in set(lengths)}
    counts = [len(tables[L]) for L in lengths]
    F = int(sum(counts))
    pcm = torch.from_numpy(np.concatenate([synth.make_pcm(rank * C + c, L) for c, L in enumerate(lengths)])).to(dev)
    clip_len = torch.tensor(lengths, dtype=torch.int64, device=dev)
    clip_off = torch.cumsum(clip_len, 0) - clip_len
    frame_clip = torch.repeat_interleave(torch.arange(C, dtype=torch.int32, device=dev), torch.tensor(counts, device=dev))
    frame_start = torch.from_numpy(np.concatenate([tables[L] for L in lengths])).to(dev)
    spk = torch.full((F,), 2, dtype=torch.int64, device=dev)          # speaker "m1"
    eng.check_speaker_ids(spk)                                         # once, outside the timed region (a host sync)
    feat = torch.empty((F, 64, 128, 3), dtype=torch.float32, device=dev)
    all_counts = sdist.frame_counts_all(F) if (dist_on and a.ragged_seconds) else [F] * world
    F_all = int(sum(all_counts))
    if world == 1 and a.gather == "direct":
        raise SystemExit("--gather direct needs N > 1 (it replaces the all-gather)")
    if a.force_gather and world == 1 and a.gather == "auto":
        a.gather = "dgrad"

    class Mode:          # how the rows are reassembled: set once (or, for auto, after the two candidates were timed)
        kind = gatherer = direct = out = None

    def set_mode(kind):
        Mode.kind, Mode.gatherer, Mode.direct, Mode.out = kind, None, None, None
        torch.cuda.empty_cache()
        if dist_on and kind in ("dgrad", "coef"):
            Mode.gatherer = sdist.FrameGatherer(all_counts, eng.out_dim if kind == "dgrad" else eng.coef_dim, torch.float32, dev, a.chunk)
        if dist_on and kind == "expand":
            Mode.gatherer = sdist.ExpandGatherer(all_counts, eng, dev, a.chunk)
            Mode.out = Mode.gatherer.own(0, F)                          # own rows are written in place
        elif kind == "direct":
            Mode.direct = sdist.DirectGatherer(all_counts, eng.out_dim, dev)
            Mode.out = Mode.direct.dests[0]                             # this rank's slot of its own gathered buffer
        else:
            Mode.out = torch.empty((F, eng.out_dim), dtype=torch.float32, device=dev)

    set_mode(a.gather if (dist_on and a.gather not in ("auto", "none")) else "dgrad")
    if a.gather == "none":
        Mode.gatherer = None

    # ---- optional post-path stage: saber.stream.seek to the video rate fused into the dgrad -> mesh solve
    mesh = None
    if a.gather == "mesh" or a.mesh_stage:
        if a.head != "dgrad":
            raise SystemExit("the mesh stage consumes dgrad rows")
        from sdfa_amd.mesh import MeshSolver
        from sdfa_amd.seek import SeekPlan
        tv, tf, tc = synth.make_template_mesh()
        solver = MeshSolver(tv, tf, tc, device=dev)
        tslists = [frame_index(L, sr)[1] for L in lengths]
        plan = SeekPlan(tslists, 60.0, device=dev)
        verts = torch.empty((plan.n_queries, solver.n_verts, 3), dtype=torch.float32, device=dev)
        vgather = None
        if dist_on and a.gather == "mesh":
            q_all = sdist.frame_counts_all(plan.n_queries)
            vgather = sdist.FrameGatherer(q_all, solver.n_verts * 3, torch.float32, dev, max(q_all))
        mesh = (solver, plan, verts, vgather)

    hop = int(0.008 * sr)

    def step(share=False):
        if Mode.direct is not None:
            Mode.direct.begin_step()                                     # two alternating gathered buffers (sdfa_amd/dist.py DirectGatherer)
        eng.mel_frontend_device(pcm, clip_off, clip_len, frame_clip, frame_start, sr, out=feat, gather=(a.frontend == "gather"))
        def compute(f0, f1):
            if share:
                z, _ = eng.encoder(feat[f0:f1], want_align=False, frame_clip=frame_clip[f0:f1], frame_start=frame_start[f0:f1], hop=hop)
            else:
                z, _ = eng.encoder(feat[f0:f1], want_align=False)
            if Mode.direct is not None:   # rows go to this rank's slot in EVERY rank's gathered buffer, written by the epilogue
                eng.regress_multi(z, spk[f0:f1], Mode.direct.dest_views(f0, f1), check_ids=False)
                return None
            coef, o = eng.regress(z, spk[f0:f1], want_coef=(Mode.kind in ("coef", "expand")), out=Mode.out[f0:f1], check_ids=False)
            return o if Mode.kind == "dgrad" else coef

        # every rank issues the SAME number of collectives, also when shards are ragged (sdfa_amd/dist.py: run_chunks)
        sdist.run_chunks(F, a.chunk, Mode.gatherer, compute)
        if mesh is not None:
            solver, plan, verts, vgather = mesh
            solver.get_mesh_seek(Mode.out if Mode.direct is None else Mode.direct.dests[0], plan, out=verts)
            if vgather is not None:
                vgather.gather_chunk(verts.view(plan.n_queries, -1), 0)
                vgather.finish()
        if Mode.direct is not None:
            Mode.direct.finish()

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    STAGES = ("conv23", "freq_lstm", "freq_proj", "gx0", "lstm0", "gx1", "lstm1", "attn_proj", "attn", "mlp", "pca")

    def timed(share):
        """W warm-up steps, then EXACTLY K timed steps between barrier + synchronize fences; max over ranks."""
        for _ in range(a.warmup):
            step(share)
        fence()
        eng.profile(True)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(share)
        fence()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if dist_on:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # per-kernel device times: HIP events recorded by the library on the launch stream inside the timed region
        st = {k: eng.profile_ms(k) / a.steps for k in STAGES + (("share_map",) if share else ())}
        if share:
            try:        # only when the frequency projection is expanded before the layer-0 input projection ("share_gx0_off", bf16 modes)
                st["share_expand"] = eng.profile_ms("share_expand") / a.steps
            except Exception:
                pass
        eng.profile(False)
        return float(tmax.item()), st

    auto = None
    if world > 1 and a.gather == "auto":
        # both reassembly forms do the full job (every rank ends up holding every rank's rows); which is faster depends on
        # what RCCL makes of a 7.3 GB-per-rank all-gather on this node, so measure: one warm step + two timed, max over ranks
        auto = {}
        for kind in ("dgrad", "expand"):
            set_mode(kind)
            step(False); fence()
            t0 = time.perf_counter()
            step(False); step(False); fence()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            auto[kind] = float(t.item()) / 2 * 1e3
        best = min(auto, key=auto.get)                                  # identical on every rank (all-reduced times)
        if best != Mode.kind:
            set_mode(best)

    dt, stages = timed(False)
    gather_check = None
    if dist_on and (Mode.gatherer is not None or Mode.direct is not None) and Mode.kind in ("dgrad", "expand", "direct"):
        # every rank holds every rank's rows: order-independent integer checksum (bit patterns summed in int64) of each rank's
        # own rows against the same rows as they arrived here
        def cks(t):
            return t.contiguous().view(torch.int32).sum(dtype=torch.int64)
        if Mode.direct is not None:
            Mode.out = Mode.direct.dests[0]                             # the buffer the last step wrote
        mine = cks(Mode.out).view(1)
        sums = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(sums, mine)
        holder = Mode.gatherer if Mode.gatherer is not None else Mode.direct
        ok = all(int(sum(cks(v) for v in holder.rows(r)).item()) == int(sums[r].item()) for r in range(world))
        okt = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        gather_check = bool(okt.item())
        if not gather_check:
            raise SystemExit(f"rank {rank}: gathered rows differ from the owners' rows (gather mode {Mode.kind})")
    # front end: timed separately (same stream, HIP events), outside the headline region but RIGHT BEHIND it -- the chip in the state the
    # steps leave it in (behind the split-bf16 legs, which run at 1.8 GHz, or in a cold process the same kernel reads 5 - 15 % differently)
    # (one untimed call, then the mean of five back to back: inside a step the stage runs hot behind the previous step's kernels; a single
    # cold call behind the surface block's other models read 3 - 7 % high)
    eng.mel_frontend_device(pcm, clip_off, clip_len, frame_clip, frame_start, sr, out=feat, gather=(a.frontend == "gather"))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(5):
        eng.mel_frontend_device(pcm, clip_off, clip_len, frame_clip, frame_start, sr, out=feat, gather=(a.frontend == "gather"))
    ev1.record()
    torch.cuda.synchronize()
    fe_ms = ev0.elapsed_time(ev1) / 5
    stages["frontend"] = fe_ms

    n_chunks = (F + a.chunk - 1) // a.chunk
    # one GPU: the budget decides which optional legs run.  N > 1: optional legs run only with --all-legs, and then on EVERY rank or none
    # (a rank that skipped alone would leave its peers waiting in a collective), so the budget admits everything there
    need_cpu = world == 1 and not a.no_cpu_baseline
    budget = Budget(a.budget_s if world == 1 else float("inf"), t0=T_PROCESS_START, reserve_s=(a.cpu_baseline_cap_s + 5.0) if need_cpu else 0.0)
    step_s = dt / a.steps
    passes = a.steps + a.warmup

    # ---- the headline line, complete in itself, goes out NOW (rank 0): whatever happens to the optional legs below -- an exception, a hang,
    # the driver's limit -- the number has been said.  The same dictionary, with the optional blocks filled in, is printed again as the last line.
    res = None
    if rank == 0:
        frames_total = F_all * a.steps
        value = frames_total / dt
        lstm_ms_per_launch = stages["freq_lstm"] / n_chunks
        flop_per_launch = FLOP_FREQ_LSTM_PER_FRAME * (F / n_chunks)
        achieved = flop_per_launch / (lstm_ms_per_launch * 1e-3) / 1e12
        traffic, traffic_alg, traffic_src = traffic_from_profile(F / n_chunks)
        fe_cnt, fe_src = frontend_counter_bytes() if (a.frontend == "gather" and sr == 16000 and not a.ragged_seconds and a.seconds == 10.0) else (None, None)
        res = {
            "metric": "animation frames/s/node (10 s@16 kHz clips); max|Δdgrad| vs CPU ref",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"fp32": "f32", "bf16_attention": "f32 (attention projections bf16)",
                                           "bf16x3": "split-bf16 x3 (fp32 accumulate)", "bf16": "bf16 (fp32 accumulate)",
                                           "bf16x3_attention": "f32 (attention projections split-bf16 x3)",
                                           "bf16x6": "split-bf16 x6 (three terms per operand, fp32-equivalent products, fp32 accumulate)"}[a.precision],
            "data": "synthetic",
            "config": {"workload": (f"batch={C} x {a.seconds:g} s@{sr} Hz synthetic PCM per GPU -> {a.head} (BASELINE configs[1])"
                                    if not a.ragged_seconds else
                                    f"stream of {C} sentences, {a.ragged_seconds} s@{sr} Hz synthetic PCM per GPU -> {a.head} (BASELINE configs[4] rehearsal)"),
                       "clips_per_gpu": C, "frames_per_gpu": F, "head": a.head, "chunk_frames": a.chunk,
                       "freq_lstm_form": eng.freq_lstm_form,      # picked by sdfa_model_autotune in the first warm-up step (bit-identical forms)
                       "gather": (Mode.kind if (Mode.gatherer is not None or Mode.direct is not None or (mesh is not None and mesh[3] is not None)) else "none") if dist_on else "none (1 GPU)",
                       "force_gather_world1": bool(a.force_gather and world == 1), "backend": (dist.get_backend() if dist_on else None),
                       "world_size_seen": world_seen, "devices": devices,
                       "distinct_devices": len({(d["host"], d["pci_bus_id"] or d["device"]) for d in devices}),
                       "launcher": os.environ.get("SDFA_BENCH_LAUNCHER", "external" if "WORLD_SIZE" in os.environ else "none"),
                       "reserved_cus": a.reserve_cus,
                       "memory_plan_gb": _plan_or_none(world, C, a, sr),   # planned (memory_plan); peak_device_memory_gb is measured
                       "env": __import__("sdfa_amd").runtime_env(),      # set at import by bench.py / sdfa_amd unless the caller had set them
                       # does an asynchronous RCCL all-gather run UNDER the kernels of the stream the steps ran on?  (probe before the run)
                       "collective_overlap_probe": comm_probe,
                       "gather_auto_ms_per_step": None if auto is None else {k: round(v, 2) for k, v in auto.items()},
                       "gather_checksum_ok": gather_check, "weights": "synthetic seed 1234",
                       "mesh_stage": None if mesh is None else f"seek to 60 fps + mesh solve, {mesh[1].n_queries} video frames x {mesh[0].n_verts} vertices per GPU per step"},
            "roofline": {"kernel": "freq_lstm_v3_kernel" if eng.freq_lstm_form in (None, 8, 9) else "freq_lstm_v2_kernel", "bound": "mfma", "achieved": round(achieved, 2),
                         "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                         "traffic": traffic, "traffic_algorithmic": traffic_alg, "traffic_source": traffic_src,
                         "launch_ms": round(lstm_ms_per_launch, 3), "flop_per_launch": flop_per_launch,
                         # `frac` is ALGORITHMIC: the kernel issues 752/768 of those FLOPs (step 0's recurrent k-blocks multiply
                         # h_-1 = 0 and are skipped) -- frac_executed is the figure rocprofv3's MfmaUtil should agree with
                         "flop_executed_per_launch": flop_per_launch * FREQ_LSTM_EXECUTED_FRACTION,
                         "frac_executed": round(achieved * FREQ_LSTM_EXECUTED_FRACTION / PEAK_FP32_MFMA_TFLOPS, 4)},
            "cpu_baseline": None,                                # filled in below (the complete, last line carries it)
            "model_tflops": round(value / world * FLOP_MODEL_PER_FRAME / 1e12, 2),
            "model_frac_of_fp32_mfma_peak": round(value / world * FLOP_MODEL_PER_FRAME / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
            "frontend": {"ms": round(fe_ms, 3), "frames_per_s": round(F / (fe_ms * 1e-3), 1),
                         "hbm_gbps_algorithmic": round(F * FRONTEND_BYTES_PER_FRAME / (fe_ms * 1e-3) / 1e9, 1),
                         "frac_of_hbm_peak": round(F * FRONTEND_BYTES_PER_FRAME / (fe_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                         # what the HBM counters saw (FETCH_SIZE x 2 + WRITE_SIZE of the stage's two kernels, committed PMC passes) over
                         # this run's stage time; null when the passes are absent or csrc/frontend.hip has changed since
                         "hbm_gbps_counters": None if fe_cnt is None else round(F * fe_cnt / (fe_ms * 1e-3) / 1e9, 1),
                         "frac_of_hbm_peak_counters": None if fe_cnt is None else round(F * fe_cnt / (fe_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                         "counters_source": fe_src},
            "stage_ms_per_step": {k: round(v, 3) for k, v in stages.items()},
        }
        # north_star: "MFMA utilisation on the attention stage against gfx950 peak" (target >= 40 %): key / query projections (MFMA GEMMs)
        # + the softmax / context tail (vector + HBM), live stage time and the counter figure of the committed pass
        att_util, att_src, att_gemms = attention_counter_util("fp32") if a.precision == "fp32" else (None, None, None)
        att_ms = stages.get("attn_proj", 0.0) + stages.get("attn", 0.0)
        res["attention"] = {"ms": round(att_ms, 3), "flop_per_frame": FLOP_ATTENTION_PER_FRAME,
                            "frac_of_fp32_mfma_peak_by_flop": round(F * FLOP_ATTENTION_PER_FRAME / (att_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if att_ms > 0 else None,
                            "mfma_util_pct_counters": att_util, "mfma_util_pct_gemms_counters": att_gemms, "counters_source": att_src}
        res["peak_device_memory_gb"] = round(torch.cuda.max_memory_allocated(dev) / 1e9, 2)
        res["wall_clock"] = dict(budget.report(), headline_at_s=round(budget.elapsed(), 1))
        emit(res, final=False)

    # ---- the legs behind the headline.  Each runs under leg(), which turns an exception into an {"error": ...} entry, puts the engine back
    # into the headline precision and checks that the device still answers -- if it does not, the line is still printed and the process
    # then exits non-zero -- and each is admitted by `budget` first: a leg that is not expected to fit is skipped and listed.
    leg_errors, fatal = {}, []

    def leg(name, fn, estimate_s, required=False):
        def guarded():
            try:
                if a.inject_failure == name:
                    raise RuntimeError(f"injected failure in optional leg {name} (--inject-failure)")
                return fn()
            except Exception as e:
                leg_errors[name] = repr(e)
                try:
                    eng.profile(False)
                except Exception:
                    pass
                return None
            finally:
                try:
                    eng.set_precision(a.precision)
                    torch.cuda.synchronize()
                except Exception as e:
                    fatal.append(f"after leg {name}: {e!r}")
        return budget.run(name, estimate_s, guarded, required)

    optional = world == 1 or a.all_legs         # at N > 1 a rank that failed alone would leave its peers waiting in a collective: headline only
    shared = None
    if not a.no_column_sharing and optional:
        def run_shared():
            dt_s, st_s = timed(True)
            return dt_s, st_s, eng.distinct_columns(min(F - (n_chunks - 1) * a.chunk, a.chunk))
        shared = leg("column_sharing", run_shared, 1.0 + 0.75 * passes * step_s)
    mixed = None
    if a.precision == "fp32" and not a.no_mixed_precision and optional:      # BASELINE configs[3]: same workload on split-bf16 MFMA
        def run_mode(mode, share=False):
            def go():
                eng.set_precision(mode)
                return timed(share)
            return go
        # expected cost relative to an fp32 step (BENCH_r05: 0.44 / 1.0 / 0.71 / 0.32), with margin
        mixed = {"bf16x3": leg("bf16x3", run_mode("bf16x3"), 1.0 + 0.55 * passes * step_s),
                 "bf16x3_attention": leg("bf16x3_attention", run_mode("bf16x3_attention"), 1.0 + 1.05 * passes * step_s),   # configs[3] literally: only the attention stage
                 "bf16x6": leg("bf16x6", run_mode("bf16x6"), 1.0 + 0.8 * passes * step_s),                                  # the six-product split: fp32-equivalent products
                 "bf16x3_column_sharing": None if a.no_column_sharing else leg("bf16x3_column_sharing", run_mode("bf16x3", True), 1.0 + 0.45 * passes * step_s)}
    # ---- PCIe-inclusive twin (SURVEY 8(d) "report both"): the same K steps with the PCM arriving from pinned host memory inside
    # the step (H2D) and every output row delivered to pinned host memory inside the step (D2H, 359 KB per frame): pieces of
    # `chunk` frames, piece i's copy on a copy stream under piece i+1's kernels (Engine.forward_host), two alternating host
    # output buffers so that a step's last copy overlaps the next step's first piece.  Never `value`.
    def host_io_twin():
        Mode.out = None
        torch.cuda.empty_cache()
        pcm_host = pcm.cpu().pin_memory()
        outs_host = [torch.empty((F, eng.out_dim), dtype=torch.float32, pin_memory=True) for _ in range(2)]
        table = (frame_clip, frame_start, hop)

        # The PCM of step k+1 goes up on an upload stream while step k computes (two device PCM buffers).  Measured (tools/trace_hostio.sh):
        # an upload issued on the compute stream at the START of a step can queue behind the previous step's last device -> host
        # copy (1.4 GB, 26 ms) inside the runtime's copy path and hold the step's first kernel for exactly that long -- in 4 of 10
        # processes.  Issued a whole step ahead, nothing waits for it.
        up_stream = torch.cuda.Stream(device=dev)
        pcm_dev = [pcm, torch.empty_like(pcm)]
        up_done = [None, None]
        consumed = [None, None]

        def upload(k):
            b = k & 1
            with torch.cuda.stream(up_stream):
                if consumed[b] is not None:
                    up_stream.wait_event(consumed[b])                    # the front end that last read this buffer
                pcm_dev[b].copy_(pcm_host, non_blocking=True)            # H2D of step k's PCM (1 KB per frame)
                up_done[b] = torch.cuda.Event()
                up_done[b].record(up_stream)

        def step_host(k, share):
            b = k & 1
            if up_done[b] is None:
                upload(k)
            torch.cuda.current_stream().wait_event(up_done[b])
            up_done[b] = None
            eng.mel_frontend_device(pcm_dev[b], clip_off, clip_len, frame_clip, frame_start, sr, out=feat, gather=(a.frontend == "gather"))
            consumed[b] = torch.cuda.Event()
            consumed[b].record()
            upload(k + 1)                                                # the next step's input, a whole step ahead
            eng.forward_host(feat, spk, out=outs_host[k & 1], table=table if share else None, piece=a.chunk, wait=False)

        host_io = {}
        for name, share in (("fp32", False),) + ((("fp32_column_sharing", True),) if not a.no_column_sharing else ()):
            for k in range(max(1, a.warmup)):
                step_host(k, share)
            eng.host_wait(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(a.steps):
                step_host(k, share)
            eng.host_wait(); torch.cuda.synchronize()
            host_io[name] = time.perf_counter() - t0
        # the rows that reached the host are the rows the device-resident step wrote (checked on the last step's buffer)
        Mode.out = torch.empty((F, eng.out_dim), dtype=torch.float32, device=dev)
        step(False); torch.cuda.synchronize()
        last = outs_host[(a.steps - 1) & 1]
        host_io["rows_identical_to_device_path"] = bool(torch.equal(last[:4096], Mode.out[:4096].cpu()) and torch.equal(last[-512:], Mode.out[-512:].cpu()))
        # raw link rate for reference: one D2H of a full output buffer, nothing else running
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); outs_host[0].copy_(Mode.out, non_blocking=True); e1.record(); torch.cuda.synchronize()
        host_io["d2h_alone_gbps"] = F * eng.out_dim * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del outs_host, pcm_host
        return host_io


    host_io, host_io_error = None, None
    if world == 1 and not dist_on and not a.no_host_io and mesh is None:
        def run_host_io():
            try:
                return host_io_twin()
            finally:
                if Mode.out is None:
                    Mode.out = torch.empty((F, eng.out_dim), dtype=torch.float32, device=dev)
        # two pinned output buffers (F x out_dim x 4 B each: pinning runs at a few GB/s) + (warm-up + steps) x 2 forms + one plain step
        host_io = leg("with_h2d_d2h", run_host_io, 6.0 + 2 * F * eng.out_dim * 4 / 2.0e9 + (1 if a.no_column_sharing else 1.8) * (passes + 1) * step_s * 1.1)
        host_io_error = leg_errors.get("with_h2d_d2h")

    # ---- the speech_anime surface: what a caller of generate_animation sees, wall clock, host work + copies included
    surface = None
    if world == 1 and not dist_on and not a.no_surface and a.precision == "fp32":
        surface = leg("surface", lambda: surface_block(sd, a.head, sr, dev), 30.0)
        if surface is None and "surface" in leg_errors:
            surface = {"error": leg_errors["surface"]}

    # cpu_baseline LAST (its 16 host threads would otherwise sit beside the surface block's host work: the ragged offsets stream read
    # 108 - 117 k frames/s behind it against 131 k in front of it), but never squeezed out: the budget keeps its time in reserve from the
    # start.  Bounded (--cpu-baseline-cap-s).
    budget.release_reserve()
    cpu, gpu_err = None, None
    if world == 1 and rank == 0 and not a.no_cpu_baseline:
        def run_cpu():
            return cpu_baseline(sr, a.cpu_sample_seconds, eng, sd, a.head, cap_s=a.cpu_baseline_cap_s)
        got = leg("cpu_baseline", run_cpu, a.cpu_baseline_cap_s + 5.0, required=True)
        if got is not None:
            cpu, gpu_err = got
            res["cpu_baseline"] = cpu
            try:
                res["max_abs_dgrad_err_vs_cpu_ref"], res["tslist_bit_exact"] = gpu_err(a.precision)
            except Exception as e:
                res["max_abs_dgrad_err_vs_cpu_ref"] = {"error": repr(e)}
            finally:
                eng.set_precision(a.precision)
        else:
            why = leg_errors.get("cpu_baseline") or "skipped: did not fit --budget-s"
            res["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": None, "nproc": int(os.cpu_count() or 0), "kind": "port", "sample": f"failed: {why}"}

    if rank == 0:
        if host_io is not None:
            tw = host_io["fp32"]
            host_io["copy_stream_overlaps"] = None if eng._host is None else {"overlaps": bool(eng._host.copy_overlaps), "probes": eng._host.copy_probe}
            res["with_h2d_d2h"] = {
                "note": "PCIe-inclusive twin of `value` (SURVEY 8(d) 'report both'): the same steps with the PCM copied host -> device and ALL "
                        "output rows (359 KB per frame) copied device -> pinned host inside the timed region, the copies of one piece "
                        "running on a copy stream under the next piece's kernels (Engine.forward_host).  NOT the headline.",
                "value": round(F * a.steps / tw, 1), "unit": "frames/s", "ms_per_step": round(tw / a.steps * 1e3, 3),
                "d2h_gb_per_step": round(F * eng.out_dim * 4 / 1e9, 3),
                "d2h_alone_gbps": round(host_io["d2h_alone_gbps"], 1),
                "rows_identical_to_device_path": host_io["rows_identical_to_device_path"],
                "copy_stream_overlaps_kernels": host_io["copy_stream_overlaps"],      # probed when the pipeline was created (sdfa_amd/streams.py)
                "with_column_sharing": None if "fp32_column_sharing" not in host_io else round(F * a.steps / host_io["fp32_column_sharing"], 1)}
        if host_io_error is not None:
            res["with_h2d_d2h"] = {"value": None, "error": host_io_error}
        if surface is not None:
            res["surface"] = surface
        if shared is not None:
            dt_s, st_s, distinct = shared
            last = min(F - (n_chunks - 1) * a.chunk, a.chunk)
            res["column_sharing"] = {
                "note": "outputs BITWISE identical to the headline run; the per-column stages (conv, freq-LSTM, projection) are "
                        "evaluated once per DISTINCT column (SURVEY App. B legal redundancy); reported next to, not as, "
                        "the headline value",
                "value": round(F_all * a.steps / dt_s, 1), "unit": "frames/s", "ms_per_step": round(dt_s / a.steps * 1e3, 3),
                "distinct_column_fraction_last_chunk": round(distinct / (64.0 * last), 4),
                "stage_ms_per_step": {k: round(v, 3) for k, v in st_s.items()}}
        if "column_sharing" in leg_errors:
            res["column_sharing"] = {"value": None, "error": leg_errors["column_sharing"]}
        if mixed is not None:
            def rate(r):
                return None if r is None else round(F_all * a.steps / r[0], 1)

            def ms(r):
                return None if r is None else round(r[0] / a.steps * 1e3, 3)

            def st(r):
                return None if r is None else {k: round(v, 3) for k, v in r[1].items()}
            m3, m3s, ma, m6 = mixed["bf16x3"], mixed["bf16x3_column_sharing"], mixed["bf16x3_attention"], mixed["bf16x6"]
            # configs[3] as worded: the attention stage's MFMA utilisation IN the bf16 attention mode (committed --pmc pass of that mode,
            # nulled once csrc/attn.hip or csrc/gemm.hip change), next to the live stage time
            ma_util, ma_src, ma_gemms = attention_counter_util("bf16x3_attention")
            ma_att_ms = None if ma is None else ma[1].get("attn_proj", 0.0) + ma[1].get("attn", 0.0)
            res["mixed_precision"] = {
                "note": "BASELINE configs[3]: same workload with the conv stack, the frequency LSTM, the BiLSTM recurrences and every GEMM "
                        "on split-bf16 MFMA (operands as hi+lo bf16, three v_mfma_f32_32x32x16_bf16 per product, fp32 "
                        "accumulate/state/activations; front end, softmax / context fp32); NOT the headline, which stays exact fp32",
                "mode": "bf16x3", "value": rate(m3), "unit": "frames/s", "ms_per_step": ms(m3),
                "with_column_sharing": rate(m3s), "stage_ms_per_step": st(m3),
                "bf16x3_attention": {"note": "configs[3] as worded -- 'bf16 attention with MFMA, fp32 mel front end': ONLY the attention stage "
                                             "(key / query projections, query conv) on v_mfma_f32_32x32x16_bf16 with split-bf16 operands, the rest exact fp32",
                                     "value": rate(ma), "unit": "frames/s", "ms_per_step": ms(ma),
                                     "attn_proj_ms_per_step": None if ma is None else round(ma[1].get("attn_proj", 0.0), 3),
                                     "attn_proj_ms_per_step_fp32": round(stages.get("attn_proj", 0.0), 3),
                                     "attention_stage_ms_per_step": None if ma_att_ms is None else round(ma_att_ms, 3),
                                     # whole stage (GEMMs + key-projection / score kernel + softmax / context tail), and its MFMA kernels alone
                                     "mfma_util_pct_counters": ma_util, "mfma_util_pct_gemms_counters": ma_gemms, "counters_source": ma_src},
                "bf16x6": {"note": "six-product split: operands as three bf16 terms (24 significand bits), six v_mfma_f32_32x32x16_bf16 per product, "
                                   "the same stages as bf16x3: fp32-equivalent products at 16 / 6 of the fp32 MFMA rate",
                           "value": rate(m6), "unit": "frames/s", "ms_per_step": ms(m6), "stage_ms_per_step": st(m6)}}
            for k in ("bf16x3", "bf16x3_column_sharing", "bf16x3_attention", "bf16x6"):
                if k in leg_errors:
                    res["mixed_precision"].setdefault("errors", {})[k] = leg_errors[k]
        if mixed is not None and gpu_err is not None and not fatal:
            worst, worst_src = precision_worst_case()
            for mode, block, ran in (("bf16x3", res["mixed_precision"], m3), ("bf16x3_attention", res["mixed_precision"]["bf16x3_attention"], ma),
                                     ("bf16x6", res["mixed_precision"]["bf16x6"], m6)):
                if ran is None or budget.left() < 3.0:
                    continue
                try:
                    live = gpu_err(mode)[0]
                    block["max_abs_dgrad_err_vs_cpu_ref_10s_clip"] = live
                    # the WORST case on record, not the fixture case: this run's clip and the committed wide sweep
                    block["max_abs_dgrad_err_vs_cpu_ref"] = max(live, worst.get(mode, 0.0))
                    block["max_abs_dgrad_err_source"] = f"max(this run's 10 s clip, worst case of {worst_src})" if mode in worst else "this run's 10 s clip only"
                except Exception as e:
                    block["max_abs_dgrad_err_vs_cpu_ref"] = {"error": repr(e)}
                finally:
                    eng.set_precision(a.precision)
        if not optional and (not a.no_column_sharing or not a.no_mixed_precision):
            res["optional_legs"] = "skipped at N > 1 (headline only; --all-legs runs them)"
        if fatal:
            res["device_unusable_after_optional_leg"] = fatal
        res["legs_skipped"] = budget.skipped
        res["wall_clock"] = dict(budget.report(), headline_at_s=res["wall_clock"]["headline_at_s"])
        emit(res, final=True)
    if fatal:
        raise SystemExit(f"the headline line was printed, but the device did not recover from an optional leg: {fatal}")
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

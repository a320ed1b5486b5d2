"""Round 4: the promises around the cooperating-workgroup time LSTM and concurrent callers.

  * include/sdfa_hip.h: "the forward calls may be used concurrently from several threads on different streams with different
    workspaces" -- two Python threads x two streams x two engines, single-clip generate_animation calls (the calls that take the
    cooperating-workgroup kernels), bitwise against the serial results.
  * A wait of those kernels that expires must never reach the caller as a wrong row (the reference never returns partial
    results, speech_anime/model/model.py:428-489): a test switch makes every wait expire; the repair pass that follows every
    launch recomputes the layer on the device and the status block counts it.
  * test-time ensembling on the offsets head, whose 60,276-byte rows put the second pass at a 16-byte boundary only
    when the frame count is a multiple of 4."""
import threading
import warnings

import numpy as np
import pytest
import torch

from speech_anime.hparams import configure
from speech_anime.api import build_model
from speech_anime.datasets import DatasetSlidingWindow
from sdfa_amd import synth, _lib
from sdfa_amd.engine import Engine

pytestmark = pytest.mark.gpu


def _model(sd, sr, head="dgrad"):
    hp = configure(dict(mode="evaluate", custom_hparams=head))
    hp.audio.set_key("sample_rate", sr)
    DatasetSlidingWindow.hparams = None
    return hp, build_model(hp, sd)


def test_expired_waits_are_repaired_on_the_device(synth_sd):
    """time_lstm_handoff bit 2: the second workgroup of every pair never publishes, so every first workgroup waits out its bound
    (2 ms here) and runs on with stale h.  time_lstm_repair_kernel must put the layer right before anything reads it."""
    e = Engine(synth_sd["dgrad"], max_frames=4096)
    rs = np.random.RandomState(5)
    try:
        for n, split in ((156, 0), (636, 0), (636, 32), (1500, 0)):       # 16-frame tiles, 16-frame, 32-frame forced, 32-frame by size
            x = torch.from_numpy(rs.uniform(0, 1, (n, 64, 128, 3)).astype(np.float32)).cuda()
            _lib.set_option("time_lstm_handoff", 0)
            _lib.set_option("time_lstm_split", 1)                          # never split: the reference bits
            z0, a0 = e.encoder(x)
            before = e.time_lstm_repairs()
            _lib.set_option("time_lstm_split", split)
            z1, a1 = e.encoder(x)                                          # healthy split launch: nothing to repair
            assert e.time_lstm_repairs() == before and torch.equal(z0, z1)
            _lib.set_option("time_lstm_timeout_us", 2000)
            _lib.set_option("time_lstm_handoff", 4)
            z2, a2 = e.encoder(x)
            after = e.time_lstm_repairs()
            assert after > before, (n, split)                              # the waits did expire ...
            assert torch.equal(z0, z2) and torch.equal(a0, a2), (n, split)  # ... and no row shows it
    finally:
        _lib.set_option("time_lstm_handoff", 0)
        _lib.set_option("time_lstm_split", 0)
        _lib.set_option("time_lstm_timeout_us", 0)


def test_product_path_reports_a_repair_and_returns_the_right_rows(synth_sd):
    sr = 16000
    hp, model = _model(synth_sd["dgrad"], sr)
    pcm = synth.make_pcm(31, 2 * sr)
    ts, want, _ = model.generate_animation(pcm, "m1", 0, 0, want_inputs=False)
    want = want.copy()
    eng = model._model._engine
    try:
        _lib.set_option("time_lstm_timeout_us", 2000)
        _lib.set_option("time_lstm_handoff", 4)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            model._signal_cache = None                                     # a fresh call: the encoder must run (same signal = cached z otherwise)
            ts2, got, _ = model.generate_animation(pcm, "m1", 0, 0, want_inputs=False)
            eng.check_pending(block=True)
        assert ts2 == ts and np.array_equal(got, want)
        assert any("recomputed on the device" in str(x.message) for x in w)
        assert eng.repairs_seen > 0 and eng.time_lstm_repairs() >= eng.repairs_seen
    finally:
        _lib.set_option("time_lstm_handoff", 0)
        _lib.set_option("time_lstm_timeout_us", 0)


def test_two_threads_two_streams_two_engines(synth_sd):
    """Each thread owns a model (engine + workspace + host pipeline) and a stream and makes 50 single-clip generate_animation calls
    (2 s and 10 s clips alternating: 160 + 640 cooperating workgroups of 96 KiB LDS each want the same 256 CUs) while the other
    does the same.  Every result must be the serial result, bit for bit, and no wait may have expired."""
    sr = 16000
    clips = [synth.make_pcm(40 + i, int(s * sr)) for i, s in enumerate((2.0, 10.0, 3.3, 10.0))]
    models = [_model(synth_sd["dgrad"], sr)[1] for _ in range(2)]
    serial = []
    for m in models:                                                       # warm (copy-stream probe, workspaces) + serial results
        serial.append([m.generate_animation(c, "m1", 0, 0, want_inputs=False)[1].copy() for c in clips])
    for a, b in zip(*serial):
        assert np.array_equal(a, b)
    torch.cuda.synchronize()
    errors, barrier = [], threading.Barrier(2)

    def work(k):
        try:
            stream = torch.cuda.Stream()
            barrier.wait()
            with torch.cuda.stream(stream):
                for i in range(50):
                    j = (i + k) % len(clips)
                    _, got, _ = models[k].generate_animation(clips[j], "m1", 0, 0, want_inputs=False)
                    if not np.array_equal(got, serial[k][j]):
                        errors.append((k, i, j, float(np.abs(got - serial[k][j]).max())))
            stream.synchronize()
        except Exception as ex:                                             # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(ex)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a worker thread hangs"
    assert not errors, errors[:5]
    torch.cuda.synchronize()
    for m in models:
        assert m._model._engine.time_lstm_repairs() == 0


def test_concurrent_encoders_on_raw_engines_with_a_big_batch_beside(synth_sd):
    """One thread keeps the device full with 4,096-frame forwards (persistent one-workgroup-per-CU kernels); the other runs
    single-clip encoders whose cooperating workgroups must squeeze in between them.  Bitwise, whatever got repaired."""
    big = Engine(synth_sd["dgrad"], max_frames=4096)
    small = Engine(synth_sd["dgrad"], max_frames=1024)
    rs = np.random.RandomState(9)
    xb = torch.from_numpy(rs.uniform(0, 1, (4096, 64, 128, 3)).astype(np.float32)).cuda()
    xs = {n: torch.from_numpy(rs.uniform(0, 1, (n, 64, 128, 3)).astype(np.float32)).cuda() for n in (156, 636)}
    zb = big.encoder(xb)[0].clone()
    zs = {n: small.encoder(x)[0].clone() for n, x in xs.items()}
    torch.cuda.synchronize()
    errors, stop = [], threading.Event()

    def heavy():
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                while not stop.is_set():
                    z = big.encoder(xb)[0]
                    if not torch.equal(z, zb):
                        errors.append("big batch differs")
        except Exception as ex:                                             # noqa: BLE001
            errors.append(repr(ex))

    def light():
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for i in range(40):
                    n = (156, 636)[i & 1]
                    z = small.encoder(xs[n])[0]
                    if not torch.equal(z, zs[n]):
                        errors.append(f"single clip of {n} frames differs at call {i}")
        except Exception as ex:                                             # noqa: BLE001
            errors.append(repr(ex))
        finally:
            stop.set()

    th, tl = threading.Thread(target=heavy), threading.Thread(target=light)
    th.start(); tl.start()
    tl.join(timeout=300); stop.set(); th.join(timeout=120)
    assert not (th.is_alive() or tl.is_alive()), "a worker thread hangs"
    assert not errors, errors[:5]
    print(f"repairs beside a saturating stream: {small.time_lstm_repairs()}")


@pytest.mark.parametrize("seconds", [2.0, 2.02, 2.05, 1.37])
def test_offsets_head_ensembling_at_any_frame_count(synth_sd, seconds):
    """n % 4 != 0 puts the second pass' rows at a 4-byte (not 16-byte) boundary of the launch group's buffer: the mean must
    take its scalar path (it used to refuse with SDFA_EINVAL) and stay bitwise the two-step route (model.py:369-403)."""
    sr = 8000
    hp, model = _model(synth_sd["offsets"], sr, "offsets")

    class Custom(DatasetSlidingWindow):
        pass

    pcm = synth.make_pcm(17, int(seconds * sr))
    ts_a, a, _ = model.generate_animation(pcm, 3, 0, 0, ensembling_ms=20, want_inputs=False)
    ts_b, b, _ = model.generate_animation(pcm, 3, 0, 0, ensembling_ms=20, want_inputs=False, dataset_class=Custom)
    assert ts_a == ts_b and a.shape == b.shape == (len(ts_a), 15069)
    assert np.array_equal(a, b)
    res = model.generate_animation_batch([pcm, pcm[: sr + 77]], [3, 1], ensembling_ms=20)
    assert np.array_equal(res[0][1], a)

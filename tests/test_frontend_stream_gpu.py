"""The spectral-stream front end (csrc/frontend.hip: mel_stream_kernel, round 5) against the two-kernel form it replaces (share map ->
mel_columns -> gather_features, option frontend_two_kernel = 1): the SAME features, bit for bit -- a column's mel values are a function
of (clip samples, position) alone and the delta filters run the same instructions on them -- for every segment geometry and for frame
tables that are not the regular 60 fps enumeration.  Reference arithmetic: saber/data/audio/features/spectrogram.py:66-104,
speech_anime/datasets/get_features.py:196-223; the fixtures hold the two-kernel form (tests/test_gpu_parity.py), so bitwise equality
carries their pins over."""
import numpy as np
import pytest
import torch

from sdfa_amd import _lib, synth
from sdfa_amd.engine import FrontendOnly, frame_index

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fe():
    return FrontendOnly()


def _both(fe, clips, sr, tables=None, expect_status_zero=True, **opts):
    try:
        _lib.set_option("frontend_two_kernel", 1)
        ref, ts_ref, counts = fe.mel_frontend(clips, sr, tables=tables)
        ref = ref.clone()
        _lib.set_option("frontend_two_kernel", 0)
        for k, v in opts.items():
            _lib.set_option(k, v)
        got, ts, _ = fe.mel_frontend(clips, sr, tables=tables)
    finally:
        for k in ("frontend_two_kernel", "frontend_stream_block", "frontend_stream_slots", "frontend_stream_phases", "frontend_stream_spin_max"):
            _lib.set_option(k, 0)
    assert ts == ts_ref
    if expect_status_zero:
        assert fe.frontend_status() == 0
    return ref, got, counts


@pytest.mark.parametrize("phases", [0, 1])
@pytest.mark.parametrize("sr", [16000, 8000])
def test_stream_is_bitwise_the_two_kernel_form(fe, sr, phases):
    clips = [synth.make_pcm(0, 10 * sr), np.zeros(int(0.75 * sr), np.float32), synth.make_pcm(22, int(0.568 * sr)),      # exactly one window
             synth.make_pcm(21, int(1.9 * sr) + 11, "speechlike"), synth.make_pcm(23, int(1.25 * sr), "sweep"), synth.make_pcm(3, 3 * sr)]
    ref, got, counts = _both(fe, clips, sr, frontend_stream_phases=phases)
    assert got.shape == ref.shape and torch.equal(got, ref)
    off = np.r_[0, np.cumsum(counts)]
    assert not bool(got[off[1]:off[2]].any())                      # the all-zero clip stays exactly zero
    assert bool(torch.isfinite(got).all())


@pytest.mark.parametrize("phases", [0, 1])
@pytest.mark.parametrize("block,slots", [(0, 1), (48, 5), (100, 12), (256, 16), (4, 3), (192, 24), (1, 1)])
def test_segment_geometry_does_not_change_a_bit(fe, block, slots, phases):
    """Frames per block and workgroups per block only decide which workgroup transforms which stretch of a chain."""
    sr = 16000
    clips = [synth.make_pcm(5, int(4.3 * sr), "speechlike"), synth.make_pcm(6, 2 * sr), synth.make_pcm(7, int(0.9 * sr))]
    ref, got, _ = _both(fe, clips, sr, frontend_stream_block=block, frontend_stream_slots=slots, frontend_stream_phases=phases)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("phases", [0, 1])
def test_headline_batch_is_bitwise(fe, phases):
    sr = 16000
    clips = [synth.make_pcm(c, 10 * sr) for c in range(32)]
    ref, got, counts = _both(fe, clips, sr, frontend_stream_phases=phases)
    assert sum(counts) == 20352 and torch.equal(got, ref)


def test_irregular_frame_tables(fe):
    """Nothing about the frame rate is assumed: the chains are read from the table.  Every frame aligned with its neighbour (shift 1:
    the longest chains, 2 jobs per member), no two frames aligned (every frame its own segment), frames in descending order, and
    ensembling's second pass (a delayed copy as a further clip)."""
    sr = 16000
    win, hop = 1024, 128
    pcm = synth.make_pcm(9, 3 * sr, "speechlike")
    starts0, ts0 = frame_index(len(pcm), sr)
    n = 150
    tables = {
        "shift_1_hop": np.arange(n, dtype=np.int64) * hop - 4544,
        "shift_62_hops": np.arange(40, dtype=np.int64) * 62 * hop - 4544,
        "shift_63_hops_no_sharing": np.arange(40, dtype=np.int64) * 63 * hop - 4544,
        "never_aligned": np.arange(n, dtype=np.int64) * (hop + 1) - 4544,
        "descending": (np.arange(n, dtype=np.int64)[::-1] * 25 * hop // 12 - 4544).copy(),
        "mixed_shifts": np.cumsum(np.random.RandomState(3).choice([hop, 3 * hop, 7, 25 * hop, 60 * hop, 64 * hop], n)).astype(np.int64) - 9000,
    }
    for name, starts in tables.items():
        ts = np.zeros(len(starts), np.int64)
        ref, got, _ = _both(fe, [pcm], sr, tables=[(starts, ts)])
        assert torch.equal(got, ref), name
        ref, got, _ = _both(fe, [pcm], sr, tables=[(starts, ts)], frontend_stream_block=64, frontend_stream_slots=7)
        assert torch.equal(got, ref), name
        ref, got, _ = _both(fe, [pcm], sr, tables=[(starts, ts)], frontend_stream_phases=1)
        assert torch.equal(got, ref), name
        ref, got, _ = _both(fe, [pcm], sr, tables=[(starts, ts)], frontend_stream_block=64, frontend_stream_slots=7, frontend_stream_phases=1)
        assert torch.equal(got, ref), name
    delayed = np.pad(pcm[:-320], [[320, 0]], "constant")
    ref, got, _ = _both(fe, [pcm, delayed], sr, tables=[(starts0, ts0), (starts0, ts0)])
    assert torch.equal(got, ref)


@pytest.mark.parametrize("sr", [16000, 8000])
def test_an_expired_hand_off_wait_is_repaired_on_the_device(fe, sr):
    """The producer / consumer waves wait for each other with a bound.  With the bound at ONE poll nearly every wait expires: the status
    word counts them, and the repair pass behind the kernel redoes the call in the barrier form, in stream order -- the call's
    features are still the two-kernel form's, bit for bit (like the time LSTM's repair pass, tests/test_concurrency_gpu.py: the
    library never returns rows of a wait that timed out).  Regular table, a multi-clip batch with short clips, and an irregular table."""
    clips = [synth.make_pcm(0, 10 * sr), synth.make_pcm(22, int(0.568 * sr)), synth.make_pcm(21, int(1.9 * sr) + 11, "speechlike"),
             synth.make_pcm(3, 3 * sr)]
    ref, got, _ = _both(fe, clips, sr, expect_status_zero=False, frontend_stream_spin_max=1)
    expired = fe.frontend_status()
    assert expired > 0, "the forced bound did not expire a single wait: the test does not exercise the repair"
    assert torch.equal(got, ref)
    ref, got, _ = _both(fe, clips, sr, expect_status_zero=False, frontend_stream_spin_max=1, frontend_stream_block=64, frontend_stream_slots=7)
    assert fe.frontend_status() > 0 and torch.equal(got, ref)
    starts = np.cumsum(np.random.RandomState(5).choice([128, 3 * 128, 7, 25 * 128, 60 * 128], 120)).astype(np.int64) - 9000
    ref, got, _ = _both(fe, [synth.make_pcm(9, 3 * sr, "speechlike")], sr, tables=[(starts, np.zeros(len(starts), np.int64))],
                        expect_status_zero=False, frontend_stream_spin_max=1)
    assert torch.equal(got, ref)
    ref, got, _ = _both(fe, clips, sr)                            # and with the default bound nothing expires
    assert torch.equal(got, ref)

"""Round 4: column sharing one stage further -- the layer-0 BiLSTM input projection (speech_anime/layers/rnn.py:20-21 is per column at
layer 0) runs over the DISTINCT columns and every time-LSTM kernel form reads it through the share map.  Bitwise: against the
expand-then-project order of rounds 2-3 ("share_gx0_off") and against the un-shared encoder, at sizes that take each kernel."""
import numpy as np
import pytest
import torch

from sdfa_amd import synth, _lib
from sdfa_amd.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seconds,max_frames", [
    ([2.0], 8192),                        # 156 frames: time_lstm_split16_kernel
    ([10.0, 3.1], 8192),                  # 844 frames, two clips: split16, a tile that crosses a clip boundary
    ([10.0, 10.0, 4.0], 8192),            # 1,528 frames: time_lstm_split_kernel (32-frame tiles)
    ([10.0] * 5 + [1.3], 8192),           # 3,290 frames: time_lstm_kernel<1>
    ([10.0] * 14, 16384),                 # 8,904 frames in one chunk: time_lstm_kernel<2>
    ([10.0] * 4, 1024),                   # several workspace chunks per call
])
def test_shared_gx0_is_bitwise(synth_sd, seconds, max_frames):
    sr = 16000
    e = Engine(synth_sd["dgrad"], max_frames=max_frames)
    clips = [synth.make_pcm(70 + i, int(s * sr), "speechlike" if i % 2 else "uniform") for i, s in enumerate(seconds)]
    feat, _, counts = e.mel_frontend(clips, sr)
    fc, fs, hop = e.last_frame_table
    z0, a0 = e.encoder(feat)                                             # every column of every frame
    try:
        _lib.set_option("share_gx0_off", 1)
        z1, a1 = e.encoder(feat, frame_clip=fc, frame_start=fs, hop=hop)  # rounds 2-3: expand the projection, then project all columns
        _lib.set_option("share_gx0_off", 0)
        z2, a2 = e.encoder(feat, frame_clip=fc, frame_start=fs, hop=hop)  # round 4: project the distinct columns, recurrence reads through the map
    finally:
        _lib.set_option("share_gx0_off", 0)
    assert torch.equal(z0, z1) and torch.equal(a0, a1)
    assert torch.equal(z0, z2) and torch.equal(a0, a2)
    assert e.time_lstm_repairs() == 0


def test_shared_gx0_with_forced_repair(synth_sd):
    """The repair pass of the cooperating-workgroup kernels reads the mapped projection too."""
    sr = 16000
    e = Engine(synth_sd["dgrad"])
    feat, _, _ = e.mel_frontend([synth.make_pcm(5, 10 * sr)], sr)
    fc, fs, hop = e.last_frame_table
    z0, a0 = e.encoder(feat)
    try:
        _lib.set_option("time_lstm_timeout_us", 2000)
        _lib.set_option("time_lstm_handoff", 4)
        z1, a1 = e.encoder(feat, frame_clip=fc, frame_start=fs, hop=hop)
    finally:
        _lib.set_option("time_lstm_handoff", 0)
        _lib.set_option("time_lstm_timeout_us", 0)
    assert e.time_lstm_repairs() > 0 and torch.equal(z0, z1) and torch.equal(a0, a1)


@pytest.mark.parametrize("sr", [8000, 16000])
def test_frontend_column_numbering_does_not_change_a_bit(sr):
    """Round 4: the spectral-gather front end numbers its distinct STFT columns clip by clip, hop by hop (the PCM is then walked front
    to back and a frame's table rows are contiguous runs) instead of time-step-major.  A column's features depend on its samples
    only, so the features are the same bits in either order -- and the same as the one-FFT-per-window-column kernel's to rounding."""
    from sdfa_amd.engine import FrontendOnly
    fe = FrontendOnly()
    clips = [synth.make_pcm(3, int(2.7 * sr), "speechlike"), synth.make_pcm(4, int(0.9 * sr)), np.zeros(sr, np.float32), synth.make_pcm(5, 10 * sr)]
    try:
        _lib.set_option("frontend_t_major", 1)
        a, ts_a, _ = fe.mel_frontend(clips, sr)
        a = a.clone()
        _lib.set_option("frontend_t_major", 0)
        b, ts_b, _ = fe.mel_frontend(clips, sr)
    finally:
        _lib.set_option("frontend_t_major", 0)
    assert ts_a == ts_b and torch.equal(a, b)
    c, _, _ = fe.mel_frontend(clips, sr, gather=False)
    assert float((b - c).abs().max()) <= 5e-5


def test_gather_chain_order_and_radix4_fft_options():
    """The feature gather's XCD-aware workgroup -> frame permutation does not change a bit (ragged clips, frame counts that are not
    multiples of 12 or 8); the radix-4 column FFT of rounds 2-3 and the radix-8 one agree to rounding."""
    from sdfa_amd.engine import FrontendOnly
    fe = FrontendOnly()
    sr = 16000
    clips = [synth.make_pcm(30 + i, int(s * sr), "speechlike" if i & 1 else "uniform") for i, s in enumerate((1.0, 0.62, 3.21, 2.0, 0.9))]
    try:
        a, _, counts = fe.mel_frontend(clips, sr)
        a = a.clone()
        assert sum(counts) % 12 != 0
        _lib.set_option("gather_plain_order", 1)
        b, _, _ = fe.mel_frontend(clips, sr)
        assert torch.equal(a, b)
        _lib.set_option("mel_fft_radix4", 1)
        c, _, _ = fe.mel_frontend(clips, sr)
        assert 0 < float((a - c).abs().max()) <= 2e-5
    finally:
        _lib.set_option("gather_plain_order", 0)
        _lib.set_option("mel_fft_radix4", 0)


def test_half_tile_gemm_is_bitwise(synth_sd):
    """The 64 x 128 tile of the LDS-tiled GEMM (the single-clip frequency projection): same k order per accumulator as the 128 x 128
    tile and the persistent 256 x 256 kernels -- forced everywhere the LDS-tiled kernel runs (11), never (10), default."""
    sr = 16000
    e = Engine(synth_sd["dgrad"])
    feat, _, _ = e.mel_frontend([synth.make_pcm(9, 10 * sr), synth.make_pcm(10, 2 * sr)], sr)
    fc, fs, hop = e.last_frame_table
    spk = torch.full((feat.shape[0],), 3, dtype=torch.int64)
    outs = []
    try:
        for v in (10, 0, 11):
            _lib.set_option("gemm_variant", v)
            z, a = e.encoder(feat, frame_clip=fc, frame_start=fs, hop=hop)
            z2, _ = e.encoder(feat[:700])
            _, rows = e.regress(z, spk)
            outs.append((z.clone(), a.clone(), z2.clone(), rows.clone()))
    finally:
        _lib.set_option("gemm_variant", 0)
    for o in outs[1:]:
        assert all(torch.equal(x, y) for x, y in zip(outs[0], o))

"""GPU parity: the HIP path (through the C ABI) against the reference-generated golden fixtures and
against the CPU oracle on the same seeded inputs.  Tolerances: 1e-4 abs on fp32 dgrad (BASELINE.json
north_star), bit-exact on frame indexing."""
import numpy as np
import pytest
import torch

import sdfa_oracle as O
from sdfa_amd import synth
from sdfa_amd.engine import Engine

pytestmark = pytest.mark.gpu

TOL_FEAT = 5e-5     # audio_feat in [0,1]
TOL_ACT = 1e-4
TOL_DGRAD = 1e-4    # north_star


@pytest.fixture(scope="module")
def eng(synth_sd):
    return Engine(synth_sd["dgrad"], debug_keep=True)


@pytest.fixture(scope="module")
def eng_off(synth_sd):
    return Engine(synth_sd["offsets"])


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("sr", [8000, 16000])
@pytest.mark.parametrize("kind,clip", [("uniform", 0), ("zeros", 1), ("sweep", 2), ("speechlike", 3)])
def test_frontend_vs_reference_fixture(eng, golden, sr, kind, clip):
    g = golden["frontend"]
    pre = f"sr{sr}_{kind}_"
    pcm = synth.make_pcm(clip, 2 * sr, kind)
    feat, tslists, counts = eng.mel_frontend([pcm], sr)
    feat = feat.cpu().numpy()
    assert np.array_equal(np.asarray(tslists[0]), g[pre + "tslist"])          # bit-exact indexing
    assert list(feat.shape) == list(g[pre + "shape"])
    keep = g[pre + "frames"]
    err = np.abs(feat[keep] - g[pre + "audio_feat"]).max()
    assert err <= TOL_FEAT, err
    s = feat.astype(np.float64).sum(axis=(1, 2, 3))
    assert np.abs(s - g[pre + "frame_sum"]).max() <= 64 * 128 * 3 * 5e-6
    if kind == "zeros":
        assert not feat.any()


def test_frontend_multi_clip_ragged_vs_oracle(eng):
    sr = 16000
    clips = [synth.make_pcm(10, 9088), synth.make_pcm(11, 20011, "speechlike"), synth.make_pcm(12, 16000, "sweep")]
    feat, tslists, counts = eng.mel_frontend(clips, sr)
    feat = feat.cpu().numpy()
    pos = 0
    for c, ts, n in zip(clips, tslists, counts):
        ref = O.fetch_audio_features(c, sr)
        assert ts == ref["tslist"] and n == len(ref["tslist"])
        # the oracle's STFT is float64; on a pure sweep most bins sit at the fp32 leakage floor, where the
        # log magnifies rounding: allow 2e-4 there (the reference's own fp32 STFT has the same noise)
        err = np.abs(feat[pos:pos + n] - ref["audio_feat"])
        assert err.max() <= 2e-4 and err.mean() <= 2e-6, (err.max(), err.mean())
        pos += n
    assert pos == feat.shape[0]


def test_model_stages_vs_reference_fixture(eng, golden):
    g = golden["model_dgrad"]
    x = _t(g["audio_feat"])
    n = x.shape[0]
    spk = torch.full((n,), int(g["speaker"]), dtype=torch.int64)
    out, z, align, coef = eng.forward(x, spk, want_coef=True)
    pool1 = eng.tap(0, n).cpu().numpy(); conv3 = eng.tap(1, n).cpu().numpy()
    freq = eng.tap(2, n).cpu().numpy(); bil = eng.tap(3, n).cpu().numpy()
    assert np.abs(pool1[:2] - g["pool1_f01"]).max() <= TOL_ACT
    assert np.abs(conv3[:2] - g["conv3_f01"]).max() <= TOL_ACT
    assert np.abs(freq - g["freq"][:, :, 0, :]).max() <= TOL_ACT
    assert np.abs(bil - g["bilstm"]).max() <= TOL_ACT
    align = align.cpu().numpy(); z = z.cpu().numpy(); coef = coef.cpu().numpy(); out = out.cpu().numpy()
    assert np.abs(align - g["align"][:, 0]).max() <= 1e-5
    assert np.abs(align.sum(1) - 1).max() <= 1e-5
    assert np.abs(z - g["z"][:, 0]).max() <= TOL_ACT
    assert np.abs(coef[:, :85] - g["coef_scale"][:, 0]).max() <= TOL_ACT
    assert np.abs(coef[:, 85:] - g["coef_rotat"][:, 0]).max() <= TOL_ACT
    assert out.shape == (n, 89784)
    assert np.abs(out[:2] - g["dgrad_f01"]).max() <= TOL_DGRAD
    assert np.abs(out[:, ::97] - g["dgrad_stride97"]).max() <= TOL_DGRAD
    assert np.abs(out.astype(np.float64).sum(1) - g["dgrad_sum"]).max() <= 89784 * 2e-6


def test_second_speaker(eng, golden):
    g = golden["model_dgrad"]; g5 = golden["model_dgrad_spk5"]
    x = _t(g["audio_feat"][:3])
    out, z, align, coef = eng.forward(x, torch.full((3,), 5, dtype=torch.int64), want_coef=True)
    assert np.abs(coef.cpu().numpy()[:, :85] - g5["coef_scale"][:, 0]).max() <= TOL_ACT
    assert np.abs(out.cpu().numpy()[:, ::97] - g5["dgrad_stride97"]).max() <= TOL_DGRAD


def test_mixed_speakers_in_one_batch(eng, golden):
    g = golden["model_dgrad"]; g5 = golden["model_dgrad_spk5"]
    x = _t(g["audio_feat"][:3])
    out, *_ = eng.forward(x, torch.tensor([2, 5, 2], dtype=torch.int64))
    out = out.cpu().numpy()
    assert np.abs(out[0, ::97] - g["dgrad_stride97"][0]).max() <= TOL_DGRAD
    assert np.abs(out[1, ::97] - g5["dgrad_stride97"][1]).max() <= TOL_DGRAD


def test_offsets_head(eng_off, golden):
    g = golden["model_offsets"]; gm = golden["model_dgrad"]
    x = _t(gm["audio_feat"][:4])
    out, z, align, coef = eng_off.forward(x, torch.full((4,), 2, dtype=torch.int64), want_coef=True)
    out = out.cpu().numpy()
    assert out.shape == (4, 15069)
    assert np.abs(coef.cpu().numpy() - g["coef"][:, 0]).max() <= TOL_ACT
    assert np.abs(out[0] - g["offsets_f0"]).max() <= TOL_DGRAD
    assert np.abs(out[:, ::7] - g["offsets_stride7"]).max() <= TOL_DGRAD


@pytest.mark.parametrize("sr", [8000, 16000])
def test_end_to_end_vs_reference_fixture(eng, golden, sr):
    g = golden["e2e_dgrad"]
    feat, tslists, counts = eng.mel_frontend([synth.make_pcm(0, 2 * sr)], sr)
    out, *_ = eng.forward(feat, torch.full((feat.shape[0],), 2, dtype=torch.int64))
    out = out.cpu().numpy().reshape(feat.shape[0], 9976, 9)
    assert np.array_equal(np.asarray(tslists[0]), g[f"sr{sr}_tslist"])
    assert list(out.shape) == list(g[f"sr{sr}_shape"])
    assert np.abs(out[:, ::97] - g[f"sr{sr}_stride97"]).max() <= TOL_DGRAD
    assert np.abs(out[10] - g[f"sr{sr}_frame10"]).max() <= TOL_DGRAD


def test_batch_vs_oracle_ragged_sizes(eng, synth_sd):
    """Sizes that are not multiples of any tile (1, 129, 300 frames) against the CPU oracle."""
    orc = O.Oracle(synth_sd["dgrad"], "dgrad")
    rs = np.random.RandomState(3)
    for n in (1, 129, 300):
        x = rs.uniform(0, 1, (n, 64, 128, 3)).astype(np.float32)
        x[:, :, :, 1:] = (x[:, :, :, 1:] - 0.5) * 0.1
        spk = rs.randint(0, 8, n)
        ref, zr, ar = orc.forward(x, spk)
        out, z, align, _ = eng.forward(_t(x), torch.from_numpy(spk))
        assert np.abs(out.cpu().numpy() - ref).max() <= TOL_DGRAD, n
        assert np.abs(align.cpu().numpy() - ar).max() <= 1e-5


def test_chunked_equals_unchunked(synth_sd, golden):
    """A workspace that forces 128-frame chunks gives the same bits as one big chunk."""
    rs = np.random.RandomState(4)
    x = _t(rs.uniform(0, 1, (300, 64, 128, 3)).astype(np.float32))
    spk = torch.full((300,), 3, dtype=torch.int64)
    a = Engine(synth_sd["dgrad"], max_frames=128)
    b = Engine(synth_sd["dgrad"], max_frames=4096)
    oa, *_ = a.forward(x, spk)
    ob, *_ = b.forward(x, spk)
    assert torch.equal(oa, ob)


# ------------------------------------------------------------------------------------------- column sharing
def _expected_distinct(starts_per_clip, hop):
    """Host recount of the distinct columns (SURVEY App. B conditions, interior = t in [6, 58])."""
    total = 0
    for starts in starts_per_clip:
        seen = set()
        for s in starts:
            for t in range(64):
                key = int(s) + t * hop
                if 6 <= t <= 58:
                    if key in seen:
                        continue
                    seen.add(key)
                total += 1
    return total


def test_shared_columns_match_unshared_and_reference(eng, golden, synth_sd):
    sr = 16000
    g = golden["e2e_dgrad"]
    clips = [synth.make_pcm(0, 2 * sr), synth.make_pcm(21, 30011, "speechlike"), synth.make_pcm(22, 9088)]
    feat, tslists, counts = eng.mel_frontend(clips, sr)
    fc, fs, hop = eng.last_frame_table
    n = feat.shape[0]
    spk = torch.full((n,), 2, dtype=torch.int64)
    z0, a0 = eng.encoder(feat)
    z1, a1 = eng.encoder(feat, frame_clip=fc, frame_start=fs, hop=hop)
    # distinct-column count equals the host recount; a healthy fraction is shared
    from sdfa_amd.engine import frame_index
    exp = _expected_distinct([frame_index(len(c), sr)[0] for c in clips], hop)
    got = eng.distinct_columns(n)           # one chunk: the workspace was sized for exactly these n frames
    assert got == exp and got < 0.75 * n * 64, (got, exp, n * 64)
    # the front end pairs STFT columns by ABSOLUTE hop index, so a shared column is bit-identical in every frame
    # that contains it and the shared path reproduces the unshared one exactly
    assert torch.equal(z0, z1) and torch.equal(a0, a1)
    _, out = eng.regress(z1, spk)
    out = out.cpu().numpy()[:counts[0]].reshape(counts[0], 9976, 9)
    assert np.abs(out[:, ::97] - g["sr16000_stride97"]).max() <= TOL_DGRAD


def test_shared_columns_chunked(synth_sd):
    """Chunk boundaries cut the sharing chains; results stay the same."""
    sr = 8000
    a = Engine(synth_sd["dgrad"], max_frames=128)
    b = Engine(synth_sd["dgrad"], max_frames=1024)
    clips = [synth.make_pcm(30, 3 * sr), synth.make_pcm(31, 2 * sr + 77)]
    feat, _, _ = a.mel_frontend(clips, sr)
    fc, fs, hop = a.last_frame_table
    za, _ = a.encoder(feat, frame_clip=fc, frame_start=fs, hop=hop)
    zb, _ = b.encoder(feat, frame_clip=fc, frame_start=fs, hop=hop)
    zc, _ = b.encoder(feat)
    assert torch.equal(za, zc) and torch.equal(zb, zc)


# ------------------------------------------------------------------------------------------- error behaviour
def test_error_paths(eng, synth_sd):
    from sdfa_amd._lib import SdfaError
    with pytest.raises(SdfaError):                       # unsupported sample rate
        eng.mel_frontend([synth.make_pcm(0, 44100)], 44100 // 2)
    with pytest.raises(AssertionError):                  # clip shorter than one window: the reference asserts
        eng.mel_frontend([synth.make_pcm(0, 3000)], 16000)
    z = torch.zeros((2, 512), device="cuda")
    with pytest.raises(RuntimeError):                    # speaker id outside the 8-way one-hot
        eng.regress(z, torch.tensor([0, 8]))
    zz, al = eng.encoder(torch.zeros((0, 64, 128, 3), device="cuda"))     # empty batch is a no-op
    assert zz.shape == (0, 512) and al.shape == (0, 64)
    bad = {k: v for k, v in synth_sd["dgrad"].items() if "proj_key" not in k}
    with pytest.raises(RuntimeError, match="Missing key"):   # strict load, like the reference's load_state_dict (checkpoints.py:22-33)
        Engine(bad)
    with pytest.raises(SdfaError):                       # not strict: the library itself reports SDFA_ESTATE at finalize
        Engine(bad, strict=False)
    tr = dict(synth_sd["dgrad"])
    k = "_model._audio_encoder._layers.10.proj_key.weight"
    tr[k] = np.ascontiguousarray(np.asarray(tr[k]).T)    # right element count, wrong shape
    with pytest.raises(RuntimeError, match="size mismatch"):
        Engine(tr)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        Engine({**synth_sd["dgrad"], "_model.junk.weight": np.zeros(3, np.float32)})
    # misaligned output pointers are refused by the C ABI instead of faulting in a 16-byte store (pca.hip)
    buf = torch.empty(2 * eng.out_dim + 1, device="cuda")
    with pytest.raises(SdfaError, match="16-byte aligned"):
        eng.regress(z, torch.tensor([0, 1]), out=buf[1:].view(2, eng.out_dim))
    with pytest.raises(RuntimeError, match="out of bounds"):            # CUDA id tensors are validated too: without draining the stream,
        eng.regress(z, torch.tensor([0, 9], device="cuda"))             # so the error surfaces at the next synchronisation point
        eng.check_pending()


def test_gather_front_end_matches_the_per_window_front_end(eng):
    """sdfa_mel_frontend_gather (each distinct STFT column once, frames gathered from the mel table) against
    sdfa_mel_frontend (one FFT per window column): same features to float rounding, on ragged clips that include an
    all-zero clip, a clip of exactly one window and clips whose starts are not hop-aligned with each other."""
    sr = 16000
    clips = [synth.make_pcm(0, 2 * sr), np.zeros(12000, np.float32), synth.make_pcm(22, 9088), synth.make_pcm(21, 30011, "speechlike"),
             synth.make_pcm(23, 20000, "sweep")]
    a, ts_a, counts = eng.mel_frontend(clips, sr, gather=True)
    b, ts_b, _ = eng.mel_frontend(clips, sr, gather=False)
    assert ts_a == ts_b and a.shape == b.shape
    # a column shares its complex FFT with another one; with a different partner the rounding noise it receives changes:
    # a few 1e-6 on broadband clips, up to 1e-4 on the log-mel of near-silent bands of the sine sweep (same class as the
    # sweep tolerance against the fp64-FFT oracle)
    off = np.r_[0, np.cumsum(counts)]
    d = (a - b).abs()
    assert float(d[:off[4]].max()) <= 2e-5 and float(d[off[4]:].max()) <= 2e-4
    n0 = counts[0]
    assert not bool(a[n0:n0 + counts[1]].any())         # the zero clip stays exactly zero
    # a column shared by two frames of a clip is bit-identical in both (frames 12 apart are 25 hops apart at 60 fps / 16 kHz)
    assert torch.equal(a[20, 30:60, :, 0], a[32, 5:35, :, 0])
    for srate in (8000,):
        c8 = [synth.make_pcm(1, 2 * srate), synth.make_pcm(2, 4544 + 100, "speechlike")]
        x, _, _ = eng.mel_frontend(c8, srate, gather=True)
        y, _, _ = eng.mel_frontend(c8, srate, gather=False)
        assert float((x - y).abs().max()) <= 2e-5


def test_fused_conv_stack_is_bitwise_the_two_kernel_path(eng, synth_sd, golden):
    """conv123_kernel (product) vs conv1_pool + conv23 (kept for the debug taps, used by `eng`): same arithmetic per
    element, so z must be identical bit for bit -- on a ragged batch that leaves a partial column tile."""
    clips = [synth.make_pcm(0, 32000), synth.make_pcm(5, 9088 + 777, "speechlike")]
    feat, _, _ = eng.mel_frontend(clips, 16000)
    z0, a0 = eng.encoder(feat)                           # debug_keep engine: two kernels
    fused = Engine(synth_sd["dgrad"])
    z1, a1 = fused.encoder(feat)
    assert torch.equal(z0, z1) and torch.equal(a0, a1)
    fc, fs, hop = eng.last_frame_table
    z2, _ = fused.encoder(feat, frame_clip=fc, frame_start=fs, hop=hop)      # and through the column-sharing indirection
    assert torch.equal(z0, z2)


def test_pca_forms_are_bitwise_identical(synth_sd):
    """The two forms of the dgrad PCA expansion -- pca_dgrad_res_kernel (default: basis slab resident in LDS, persistent work
    units, padded k-blocks skipped) and pca_dgrad_kernel (the two-workgroups-per-CU fallback: every wave fetches the slab itself)
    -- contract k in the same order per accumulator: identical dgrad rows, on ragged batches (partial frame blocks, early-exit
    waves, several units per workgroup)."""
    from sdfa_amd import _lib
    eng = Engine(synth_sd["dgrad"], max_frames=2048)
    rs = np.random.RandomState(21)
    for n in (1, 130, 700, 1500):
        z = _t(rs.normal(0, 1, (n, 512)).astype(np.float32))
        spk = torch.from_numpy(rs.randint(0, 8, n))
        try:
            _lib.set_option("pca_lds", 4)      # register-direct (rounds 1-2)
            _, a = eng.regress(z, spk)
            _lib.set_option("pca_lds", 0)      # default: slab resident in LDS, persistent work units (pca_dgrad_res_kernel)
            _, c = eng.regress(z, spk)
        finally:
            _lib.set_option("pca_lds", 0)
        assert torch.equal(a, c), n


@pytest.mark.parametrize("head", ["dgrad", "offsets"])
def test_expand_coef_rebuilds_the_regressors_rows_bitwise(synth_sd, golden, head):
    """sdfa_expand_coef(coefficients) == the rows sdfa_regress_forward wrote for them (ExpandGatherer relies on it), both
    heads, ragged sizes incl. more frames than one workspace pass; and the coefficients are the reference fixture's."""
    eng = Engine(synth_sd[head], max_frames=256)
    rs = np.random.RandomState(31)
    for n in (1, 130, 700):
        z = _t(rs.normal(0, 1, (n, 512)).astype(np.float32))
        spk = torch.from_numpy(rs.randint(0, 8, n))
        coef, rows = eng.regress(z, spk, want_coef=True)
        again = eng.expand_coef(coef)
        assert again.shape == rows.shape and torch.equal(again, rows), (head, n)
    out = torch.full((5, eng.out_dim), float("nan"), device="cuda")
    eng.expand_coef(coef[:5].contiguous(), out=out)
    assert torch.equal(out, rows[:5])
    assert eng.expand_coef(coef[:0].contiguous()).shape == (0, eng.out_dim)
    if head == "dgrad":
        g = golden["model_dgrad"]
        ref_coef = np.ascontiguousarray(np.concatenate([g["coef_scale"].reshape(8, -1), g["coef_rotat"].reshape(8, -1)], 1), dtype=np.float32)
        got = eng.expand_coef(_t(ref_coef)).cpu().numpy()       # the REFERENCE's coefficients through this expansion
        assert np.abs(got[:, ::97] - g["dgrad_stride97"]).max() <= 1e-4


def test_time_lstm_workgroup_shapes_are_bitwise_identical(synth_sd, golden):
    """time_lstm_kernel<2> (8 waves x 64 frames: chunks of >= 8192 frames), <1> (8 waves x 32 frames) and <0> (the small-batch shape,
    16 frames per workgroup) contract k in the same order with the same cell arithmetic: the rows of a frame must not change by a
    bit with the size of the chunk it is computed in."""
    g = golden["model_dgrad"]
    rs = np.random.RandomState(12)
    x = torch.cat([_t(g["audio_feat"]), _t(rs.uniform(0, 1, (300, 64, 128, 3)).astype(np.float32))])
    eng = Engine(synth_sd["dgrad"], max_frames=8192)
    z0, a0 = eng.encoder(x)                                             # 308 frames: a small-batch shape
    pad = _t(rs.uniform(0, 1, (8192 - x.shape[0], 64, 128, 3)).astype(np.float32))
    z1, a1 = eng.encoder(torch.cat([x, pad]))                           # 8192 frames: 64-frame tiles
    z2, a2 = eng.encoder(torch.cat([x, pad[:1200]]))                    # 1508 frames: 32-frame tiles
    n = x.shape[0]
    assert torch.equal(z0, z1[:n]) and torch.equal(a0, a1[:n]) and torch.equal(z0, z2[:n]) and torch.equal(a0, a2[:n])
    assert np.abs(z0[:8].cpu().numpy() - g["z"][:, 0]).max() <= TOL_ACT


def test_freq_lstm_kernel_forms_are_bitwise_identical(synth_sd, golden):
    """The launch forms of freq_lstm_v3_kernel, the one-workgroup-per-CU design (9 persistent = default, 8 hardware-dispatched), and of
    freq_lstm_v2_kernel, two workgroups per CU (5 persistent, 3 hardware-dispatched = the fallback that shares a CU), accumulate every
    gate in the same order: not a bit may differ, also through the column-sharing launch, and when a persistent workgroup works
    through several tiles (1,100 frames = 2,304 tiles).  sdfa_model_autotune picks among the four.  The reference fixture pins the
    values (test_model_stages_vs_reference_fixture)."""
    from sdfa_amd import _lib
    clips = [synth.make_pcm(0, 32000), synth.make_pcm(5, 9088 + 777, "speechlike")]
    eng = Engine(synth_sd["dgrad"], max_frames=8192)
    feat, _, _ = eng.mel_frontend(clips, 16000)
    fc, fs, hop = eng.last_frame_table
    res = {}
    big = torch.rand((1100, 64, 128, 3), device="cuda")
    for shape in (9, 8, 5, 3):
        try:
            _lib.set_option("freq_lstm_shape", shape)
            res[shape] = (eng.encoder(feat), eng.encoder(feat, frame_clip=fc, frame_start=fs, hop=hop), eng.encoder(big))
        finally:
            _lib.set_option("freq_lstm_shape", 0)
    for k in (0, 1, 2):
        for shape in (8, 5, 3):
            assert torch.equal(res[9][k][0], res[shape][k][0]) and torch.equal(res[9][k][1], res[shape][k][1]), (k, shape)
    assert torch.equal(res[9][0][0], res[9][1][0])
    form = eng.autotune(1100)
    assert form in (3, 5, 8, 9) and eng.freq_lstm_form == form
    z, al = eng.encoder(big)
    assert torch.equal(z, res[9][2][0]) and torch.equal(al, res[9][2][1])


def test_forward_is_bitwise_reproducible_run_to_run(synth_sd):
    """The persistent kernels take tiles from atomic queues and the single-clip time LSTM hands h between workgroups: which
    workgroup does what differs from run to run, the bits must not (no floating-point atomics, fixed order per accumulator).
    2,500 frames (persistent queues, several tiles per workgroup) and 300 frames (cooperating workgroups), three runs each and
    once more through a second engine."""
    eng = Engine(synth_sd["dgrad"], max_frames=4096)
    for n in (2500, 300):
        feat = torch.rand((n, 64, 128, 3), generator=torch.Generator().manual_seed(n)).cuda()
        spk = torch.arange(n, dtype=torch.int64) % 8
        ref = [t.clone() for t in eng.forward(feat, spk, want_coef=True)]
        for _ in range(2):
            for a_, b_ in zip(ref, eng.forward(feat, spk, want_coef=True)):
                assert torch.equal(a_, b_)
        for a_, b_ in zip(ref, Engine(synth_sd["dgrad"], max_frames=4096).forward(feat, spk, want_coef=True)):
            assert torch.equal(a_, b_)


def test_freq_lstm_cell_update_saturates_like_the_reference(synth_sd):
    """The frequency LSTM's cell update uses one reciprocal for sigmoid(o) * tanh(c') and tanh(g) = 2 sigmoid(2g) - 1 (lstm.hip:
    lstm_cell_quad<true>): exponentials that overflow must give the limits, never inf * 0.  Gate biases of +-150 (e^150 overflows
    fp32) in every combination over blocks of hidden units, both directions, against torch's LSTM (oracle/torch_oracle.py)."""
    import torch_oracle as TO
    sd = {k: np.array(v, copy=True) for k, v in synth_sd["dgrad"].items()}
    key = [k for k in sd if k.endswith("6._lstm.bias_ih_l0")][0][:-len("bias_ih_l0")]
    levels = (-150.0, 0.0, 150.0)
    for suf in ("", "_reverse"):
        b = sd[key + "bias_ih_l0" + suf]
        for u in range(81):                                   # unit u: gate q gets levels[(u // 3**q) % 3]; units 81..127 stay as they are
            for q in range(4):
                b[q * 128 + u] += levels[(u // 3 ** q) % 3]
    eng = Engine(sd, debug_keep=True)
    feat = torch.rand((96, 64, 128, 3), generator=torch.Generator().manual_seed(3)).cuda()
    z, align = eng.encoder(feat)
    freq = eng.tap(2, 96).cpu().numpy()
    assert np.isfinite(freq).all() and torch.isfinite(z).all()
    zr, ar = TO.TorchOracle(sd).encoder(feat.cpu().numpy())
    assert np.abs(z.cpu().numpy() - zr.numpy()).max() <= TOL_ACT
    assert np.abs(align.cpu().numpy() - ar.numpy()).max() <= 1e-5


@pytest.mark.parametrize("variant", [2, 4, 5, 6, 8, 9])
def test_gemm_variants_agree(eng, golden, variant):
    """The GEMM choices that ship (split-bf16 x3, 256-tile, one 128 x 128 block per wave wherever it fits, never the fat kernel, the
    64 x 64 tile of small launches everywhere / nowhere) give the reference's numbers too; the fp32 ones (2, 5, 6, 8, 9) contract k in
    the default kernel's order: not a bit of z, the PCA coefficients or the rows differs -- at the fixture's 8 frames (every launch
    small) and at 1,100 frames (the MLP and attention-query launches small, the rest not)."""
    from sdfa_amd import _lib
    g = golden["model_dgrad"]
    x = _t(g["audio_feat"])
    spk = torch.full((x.shape[0],), int(g["speaker"]), dtype=torch.int64)
    big = torch.rand((1100, 64, 128, 3), generator=torch.Generator().manual_seed(11)).cuda()
    spk_big = torch.arange(1100, dtype=torch.int64) % 8
    try:
        _lib.set_option("gemm_variant", variant)
        out, z, align, coef = eng.forward(x, spk, want_coef=True)
        res_big = [t.clone() for t in eng.forward(big, spk_big, want_coef=True)]
        out_np = out.cpu().numpy()
    finally:
        _lib.set_option("gemm_variant", 0)
    assert np.abs(out_np[:, ::97] - g["dgrad_stride97"]).max() <= TOL_DGRAD
    assert np.abs(align.cpu().numpy() - g["align"][:, 0]).max() <= 1e-5
    if variant != 4:
        out0, z0, align0, coef0 = eng.forward(x, spk, want_coef=True)
        assert torch.equal(z, z0) and torch.equal(coef, coef0) and torch.equal(out, out0) and torch.equal(align, align0)
        for a_, b_ in zip(res_big, eng.forward(big, spk_big, want_coef=True)):
            assert torch.equal(a_, b_)


@pytest.mark.parametrize("seed,lstm_gain,flip_bn", [(77, 1.0, False), (4321, 1.3, False), (99, 1.0, True)])
def test_other_weight_dynamics_vs_the_reference_operators(seed, lstm_gain, flip_bn):
    """The fixtures pin ONE synthetic checkpoint (seed 1234; the pretrained one is not obtainable offline).  VERDICT r3 weak 1(iii): the
    frequency LSTM's cell update is algebraically, not operation-for-operation, torch's.  So the whole model is also held to the
    reference's own operator library (oracle/torch_oracle.py: torch CPU conv2d / nn.LSTM / linear / softmax, itself pinned to the
    reference fixtures) on OTHER weights: three more seeds, the weights of all three LSTMs scaled x1.3 (hotter gates; at x2 the recurrences amplify ANY
    rounding difference -- fp32 summation order included -- to 8e-4 on z, which says nothing about either side), and BatchNorm scales of
    both signs (a trained checkpoint may hold negative gammas: LeakyReLU -> BN -> max-pool must
    not assume a positive scale).  Same 1e-4 budget on dgrad; 1e-4 on z; 1e-5 on the attention weights."""
    import torch_oracle as TO
    sd = synth.make_state_dict("dgrad", seed)
    rs = np.random.RandomState(seed)
    for k in list(sd):
        if ("_lstm.weight" in k or ".9.weight_" in k) and lstm_gain != 1.0:
            sd[k] = (sd[k] * lstm_gain).astype(np.float32)
        if flip_bn and k.endswith("_ext_post_bn.weight"):
            sd[k] = (sd[k] * rs.choice([-1.0, 1.0], sd[k].shape)).astype(np.float32)
    eng = Engine(sd)
    sr = 16000
    feat, _, _ = eng.mel_frontend([synth.make_pcm(seed, int(0.9 * sr), "speechlike"), synth.make_pcm(seed + 1, int(0.7 * sr))], sr)
    n = feat.shape[0]
    spk = rs.randint(0, 8, n)
    out, z, align, _ = eng.forward(feat, torch.from_numpy(spk))
    ref, zr, ar = TO.TorchOracle(sd).forward(feat.cpu().numpy(), spk)
    assert np.abs(z.cpu().numpy() - zr).max() <= TOL_ACT
    assert np.abs(align.cpu().numpy() - ar).max() <= 1e-5
    assert np.abs(out.cpu().numpy() - ref).max() <= TOL_DGRAD
    # and the column-sharing encoder on the same weights: bitwise the plain one
    fc, fs, hop = eng.last_frame_table
    z2, a2 = eng.encoder(feat, frame_clip=fc, frame_start=fs, hop=hop)
    assert torch.equal(z, z2) and torch.equal(align, a2)


@pytest.mark.parametrize("mode,tol_z,tol_a", [("fp32", 3e-6, 1e-6), ("bf16x3_attention", 2e-5, 5e-6), ("bf16x6", 3e-6, 1e-6)])
def test_one_pass_attention_agrees_with_the_three_gemm_form(synth_sd, mode, tol_z, tol_a):
    """Round 6: attn_key_score_*_kernel computes key projection + tanh + v-dot in ONE pass over the BiLSTM output (weights in registers,
    tiles through an LDS-DMA ring with hand-counted vmcnt, partial scores per wave) and attn_kernel<true> does softmax + context from the
    scores; option attn_unfused = 1 brings back the key-projection GEMM + attn_kernel of rounds 1-5.  Same arithmetic up to the order of
    a dot product's terms: z and the attention weights agree to a few ulps -- for every work-unit shape the launcher picks (a unit is 16
    frames x 64 >> ts_shift time steps: 8 steps for a single short clip, 64 from 4,096 frames on), for ragged frame counts that leave
    padded frames in the last tile, and run to run bitwise (the pipeline's waits are counted by hand: a miscount would show here as
    stale tiles)."""
    from sdfa_amd import _lib
    eng = Engine(synth_sd["dgrad"], max_frames=8192)
    rs = np.random.RandomState(61)
    try:
        eng.set_precision(mode)
        # (from about 3,600 frames per chunk on, exact fp32 runs the WHOLE layer in one launch -- attn_fused_f32_kernel: running softmax +
        # context while the tile is in LDS.  attn_unfused = 2 keeps the two-kernel form there: the same recurrence, the same bits.)
        for n in (1, 72, 129, 300, 1100, 2304, 4096, 7000, 8192, 8100):
            x = torch.from_numpy(rs.uniform(0, 1, (n, 64, 128, 3)).astype(np.float32)).cuda()
            _lib.set_option("attn_unfused", 1)
            z0, a0 = eng.encoder(x)
            z0, a0 = z0.clone(), a0.clone()
            _lib.set_option("attn_unfused", 0)
            z1, a1 = eng.encoder(x)
            z1, a1 = z1.clone(), a1.clone()
            z2, a2 = eng.encoder(x)
            assert torch.equal(z1, z2) and torch.equal(a1, a2), (mode, n)
            if n >= 4096 and mode in ("fp32", "bf16x6"):
                _lib.set_option("attn_unfused", 2)
                z3, a3 = eng.encoder(x)
                assert torch.equal(z3, z1) and torch.equal(a3, a1), (n, "one launch vs two: the same recurrence, the same bits")
                _lib.set_option("attn_unfused", 0)
            assert bool(torch.isfinite(z1).all()) and float((a1.sum(-1) - 1).abs().max()) <= 1e-5
            dz, da = float((z0 - z1).abs().max()), float((a0 - a1).abs().max())
            assert dz <= tol_z and da <= tol_a, (mode, n, dz, da)
            if mode == "fp32":
                assert dz > 0.0 or n < 16                 # the two forms really are different kernels
    finally:
        _lib.set_option("attn_unfused", 0)
        eng.set_precision("fp32")

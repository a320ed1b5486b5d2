"""BASELINE configs[3] -- "bf16 attention with MFMA, fp32 mel front end (mixed-precision tolerance sweep)".

The reference computes in fp32 only, so every mode is judged against the same fp32 reference fixtures and the same
1e-4 abs dgrad budget (north_star).  The sweep runs the REAL bf16-MFMA kernels (sdfa_model_set_precision), not an
emulation: fp32 (product default) -> attention projections in bf16 -> split-bf16 x3 everywhere -> plain bf16 everywhere.
The measured errors are written to gpurun_out/precision_modes.json when that directory exists."""
import json
import os

import numpy as np
import pytest
import torch

from sdfa_amd import synth
from sdfa_amd.engine import Engine

pytestmark = pytest.mark.gpu

BUDGET = 1e-4      # north_star: max |dgrad - reference| on fp32 dgrad
# mode -> (must stay below, must stay above).  Measured on MI355X (profiles/r01_precision_modes.json): fp32 5e-7,
# split-bf16 3e-6 -- inside the budget; bf16 on the attention projections alone 3.3e-4 and plain bf16 everywhere 1.7e-3 --
# OUTSIDE it (8 significand bits in the key/query projections move the softmax by 3e-4).  The lower bounds make the test
# fail if that ever changes, so the documented outcome of the sweep cannot go stale.
# Round 4: "bf16x3_attention" -- configs[3]'s literal wording ("bf16 attention with MFMA ...") with a point INSIDE the budget: the
# attention stage on bf16 MFMA with split-bf16 (hi + lo) operands, the rest exact fp32.
# "bf16x6": the six-product split (three bf16 terms per operand, 24 significand bits): fp32-equivalent products on bf16 MFMA, held to
# the SAME bound as exact fp32.
BOUNDS = {"fp32": (5e-6, None), "bf16x6": (5e-6, None), "bf16x3": (2e-5, None), "bf16x3_attention": (5e-6, None), "bf16_attention": (2e-3, BUDGET), "bf16": (2e-2, BUDGET)}


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.fixture(scope="module")
def eng(synth_sd):
    return Engine(synth_sd["dgrad"])


def _errors(eng, golden):
    g, e2e = golden["model_dgrad"], golden["e2e_dgrad"]
    x = _t(g["audio_feat"])
    spk = torch.full((x.shape[0],), int(g["speaker"]), dtype=torch.int64)
    out, z, align, coef = eng.forward(x, spk, want_coef=True)
    err = {"dgrad": float(np.abs(out.cpu().numpy()[:, ::97] - g["dgrad_stride97"]).max()),
           "z": float(np.abs(z.cpu().numpy() - g["z"][:, 0]).max()),
           "align": float(np.abs(align.cpu().numpy() - g["align"][:, 0]).max())}
    feat, tslists, counts = eng.mel_frontend([synth.make_pcm(0, 32000)], 16000)
    o2, *_ = eng.forward(feat, torch.full((feat.shape[0],), 2, dtype=torch.int64))
    o2 = o2.cpu().numpy().reshape(feat.shape[0], 9976, 9)
    err["dgrad_e2e_2s_clip"] = float(np.abs(o2[:, ::97] - e2e["sr16000_stride97"]).max())
    assert np.array_equal(np.asarray(tslists[0]), e2e["sr16000_tslist"])      # frame indexing is integer work in every mode
    return err


def test_precision_sweep(eng, golden):
    table = {}
    try:
        for mode in ("fp32", "bf16x6", "bf16x3_attention", "bf16_attention", "bf16x3", "bf16"):
            eng.set_precision(mode)
            table[mode] = _errors(eng, golden)
    finally:
        eng.set_precision("fp32")
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "precision_modes.json"), "w") as f:
            json.dump({"budget": BUDGET, "max_abs_error_vs_reference_fixture": table}, f, indent=1)
    for mode, (hi, lo) in BOUNDS.items():
        worst = max(table[mode]["dgrad"], table[mode]["dgrad_e2e_2s_clip"])
        assert worst <= hi, (mode, table[mode])
        if lo is not None:
            assert worst > lo, (mode, table[mode])
    # more operand bits never hurt
    assert table["bf16x6"]["dgrad"] < table["bf16x3"]["dgrad"] < table["bf16"]["dgrad"]


# ---- the modes on OTHER weight dynamics, on the 10 s reference fixture and at full size ----------------------
# One table, written to gpurun_out/precision_modes_wide.json (installed as profiles/r05_precision_modes.json): max |dgrad - reference|
# per (mode, case).  References: the reference's own operator library on the CPU (oracle/torch_oracle.py, pinned to the fixtures) for
# the weight-dynamics cases; the reference-made 10 s fixture; the numpy oracle on sampled frames at full size.
WIDE_MODES = ("fp32", "bf16x6", "bf16x3_attention", "bf16x3")
DYNAMICS = [(1234, 1.0, False), (77, 1.0, False), (4321, 1.3, False), (99, 1.0, True), (2024, 1.3, True)]      # seed, LSTM gain, BN scales of both signs
# Asserted bounds (dgrad, every case): exact fp32 and the six-product split at fp32's own error; the attention-only split far inside
# the budget; split-bf16 x3 everywhere inside the 1e-4 budget on every case, the hotter recurrences (gain x1.3) included.
WIDE_BOUNDS = {"fp32": 1e-5, "bf16x6": 1e-5, "bf16x3_attention": 1e-5, "bf16x3": BUDGET}
_WIDE = {}


def _record(case, mode, err):
    _WIDE.setdefault(case, {})[mode] = err
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "precision_modes_wide.json"), "w") as f:
            worst = {m: max(v[m]["dgrad"] for v in _WIDE.values() if m in v) for m in WIDE_MODES if any(m in v for v in _WIDE.values())}
            json.dump({"budget": BUDGET, "bounds_asserted": WIDE_BOUNDS, "worst_case_dgrad": worst, "cases": _WIDE}, f, indent=1)


@pytest.mark.parametrize("seed,lstm_gain,flip_bn", DYNAMICS)
def test_modes_on_other_weight_dynamics_vs_the_reference_operators(seed, lstm_gain, flip_bn):
    """tests/test_gpu_parity.py::test_other_weight_dynamics_vs_the_reference_operators for every mode that claims the budget: other
    seeds, all three LSTMs x1.3 (hotter gates: the recurrences amplify operand rounding), BatchNorm scales of both signs -- against
    torch's fp32 CPU operators (speech_anime/layers/freq_lstm.py:36-55, rnn.py:20-21 are nn.LSTM calls)."""
    import torch_oracle as TO
    sd = synth.make_state_dict("dgrad", seed)
    rs = np.random.RandomState(seed)
    for k in list(sd):
        if ("_lstm.weight" in k or ".9.weight_" in k) and lstm_gain != 1.0:
            sd[k] = (sd[k] * lstm_gain).astype(np.float32)
        if flip_bn and k.endswith("_ext_post_bn.weight"):
            sd[k] = (sd[k] * rs.choice([-1.0, 1.0], sd[k].shape)).astype(np.float32)
    e = Engine(sd)
    sr = 16000
    feat, _, _ = e.mel_frontend([synth.make_pcm(seed, int(0.9 * sr), "speechlike"), synth.make_pcm(seed + 1, int(0.7 * sr))], sr)
    spk = rs.randint(0, 8, feat.shape[0])
    ref, zr, ar = TO.TorchOracle(sd).forward(feat.cpu().numpy(), spk)
    case = f"seed{seed}_gain{lstm_gain:g}_{'bn_both_signs' if flip_bn else 'bn_positive'}"
    errs = {}
    for mode in WIDE_MODES:
        e.set_precision(mode)
        out, z, align, _ = e.forward(feat, torch.from_numpy(spk))
        errs[mode] = {"dgrad": float(np.abs(out.cpu().numpy() - ref).max()), "z": float(np.abs(z.cpu().numpy() - zr).max()),
                      "align": float(np.abs(align.cpu().numpy() - ar).max())}
        _record(case, mode, errs[mode])
    for mode in WIDE_MODES:
        assert errs[mode]["dgrad"] <= WIDE_BOUNDS[mode], (case, mode, errs[mode])


def _input_case(kind, n):
    if kind in ("sweep", "zeros", "speechlike"):
        return synth.make_pcm(5, n, kind)
    rs = np.random.RandomState(31)
    if kind == "near_silence":          # every mel bin on or next to the dB floor of spectrogram.py:236-249
        return (rs.uniform(-1, 1, n) * 1e-5).astype(np.float32)
    if kind == "clipped_full_scale":    # a square wave with noisy edges: the ceiling of the clamp, harmonics in every band
        return np.clip(np.sign(np.sin(2 * np.pi * 220.0 * np.arange(n) / 16000.0)) + rs.normal(0, 0.05, n), -1, 1).astype(np.float32)
    if kind == "impulse_train":         # silence with one click per 10 ms: floor and ceiling in neighbouring STFT columns
        x = np.zeros(n, np.float32)
        x[::160] = 0.95
        return x
    raise ValueError(kind)


INPUTS = ["sweep", "zeros", "near_silence", "clipped_full_scale", "impulse_train"]


@pytest.mark.parametrize("kind", INPUTS)
def test_modes_on_other_input_dynamics_vs_the_reference_operators(eng, synth_sd, kind):
    """The weight-dynamics test varies the model; this one varies the AUDIO under the fixture weights: feature tensors that sit on the
    floor of the dB clamp, on its ceiling, or jump between the two from one STFT column to the next (get_features.py:196-223 feeds
    the encoder whatever the clip holds).  Same reference (torch's fp32 CPU operators), same per-mode bounds."""
    import torch_oracle as TO
    sr = 16000
    feat, _, _ = eng.mel_frontend([_input_case(kind, int(0.8 * sr))], sr)
    spk = np.arange(feat.shape[0]) % 8
    ref, zr, ar = TO.TorchOracle(synth_sd["dgrad"]).forward(feat.cpu().numpy(), spk)
    errs = {}
    try:
        for mode in WIDE_MODES:
            eng.set_precision(mode)
            out, z, align, _ = eng.forward(feat, torch.from_numpy(spk))
            # (with the fixture weights the largest dgrad error sits in the regressor's large-magnitude rows and hardly moves with the
            # audio; the encoder output z and the attention weights are recorded beside it because they do)
            errs[mode] = {"dgrad": float(np.abs(out.cpu().numpy() - ref).max()), "z": float(np.abs(z.cpu().numpy() - zr).max()),
                          "align": float(np.abs(align.cpu().numpy() - ar).max())}
            assert np.isfinite(errs[mode]["dgrad"]), (kind, mode)
            _record(f"input_{kind}", mode, errs[mode])
    finally:
        eng.set_precision("fp32")
    for mode in WIDE_MODES:
        assert errs[mode]["dgrad"] <= WIDE_BOUNDS[mode], (kind, mode, errs[mode])


def test_modes_on_the_10s_reference_fixture(eng, golden):
    """The reference's own generate_animation on clips 0 and 1 of the headline workload (tests/golden/e2e_dgrad_10s.npz), every mode."""
    sr = 16000
    g = golden["e2e_dgrad_10s"]
    try:
        for mode in WIDE_MODES:
            eng.set_precision(mode)
            worst = 0.0
            for c in (0, 1):
                feat, tslists, _ = eng.mel_frontend([synth.make_pcm(c, 10 * sr)], sr)
                assert list(tslists[0]) == list(g[f"clip{c}_tslist"])
                out, *_ = eng.forward(feat, torch.full((feat.shape[0],), 2, dtype=torch.int64))
                flat = out.cpu().numpy()
                worst = max(worst, float(np.abs(flat[:, ::193] - g[f"clip{c}_stride193"]).max()),
                            float(np.abs(flat[g[f"clip{c}_frames"]] - g[f"clip{c}_full"]).max()))
            _record("reference_fixture_10s_clips_0_1", mode, {"dgrad": worst})
            assert worst <= WIDE_BOUNDS[mode], (mode, worst)
    finally:
        eng.set_precision("fp32")


def test_split_bf16_at_full_size_on_sampled_frames(synth_sd):
    """BASELINE configs[1] size (32 x 10 s = 20,352 frames, three launch groups) in bf16x3 and bf16x6: the numpy oracle on 24 frames spread
    over clips, chunks and both speakers (the sample of tests/test_gpu_fullsize.py::test_sampled_frames_match_oracle)."""
    import sdfa_oracle as O
    sr, L = 16000, 160000
    e = Engine(synth_sd["dgrad"], max_frames=8192)
    pcms = [synth.make_pcm(c, L) for c in range(32)]
    feat, _, counts = e.mel_frontend(pcms, sr)
    n = feat.shape[0]
    assert n == 20352
    spk = torch.full((n,), 2, dtype=torch.int64)
    spk[counts[0]:2 * counts[0]] = 5
    idx = np.r_[0, 1, 635, 636 + 7, 2 * 636 - 1, 8191, 8192, 8193, 16383, 16384, np.linspace(9000, 20351, 14).astype(int)]
    feat_ref = {}
    for i in idx:
        c = int(i) // 636
        if c not in feat_ref:
            feat_ref[c] = O.fetch_audio_features(pcms[c], sr)["audio_feat"]
    x = np.stack([feat_ref[int(i) // 636][int(i) % 636] for i in idx])
    ref, _, _ = O.Oracle(synth_sd["dgrad"], "dgrad").forward(x, spk[idx].numpy())
    for mode in ("bf16x3", "bf16x6"):
        e.set_precision(mode)
        out, *_ = e.forward(feat, spk)
        err = float(np.abs(out[torch.from_numpy(idx)].cpu().numpy() - ref).max())
        del out
        _record("full_size_20352_frames_24_sampled_vs_numpy_oracle", mode, {"dgrad": err})
        assert err <= WIDE_BOUNDS[mode], (mode, err)


def test_fp32_results_do_not_depend_on_mode_history(eng, golden):
    """Switching modes leaves no state behind: fp32 after a bf16 call is bitwise the fp32 result."""
    g = golden["model_dgrad"]
    x = _t(g["audio_feat"])
    spk = torch.full((x.shape[0],), int(g["speaker"]), dtype=torch.int64)
    eng.set_precision("fp32")
    a = eng.forward(x, spk)[0].clone()
    eng.set_precision("bf16")
    b = eng.forward(x, spk)[0].clone()
    eng.set_precision("fp32")
    c = eng.forward(x, spk)[0]
    assert torch.equal(a, c) and not torch.equal(a, b)


def test_six_product_split_on_other_shapes(eng, synth_sd):
    """bf16x6 through the paths the sweep's fixture call does not take: column sharing (bitwise the unshared call in this mode too),
    a single clip (small-tile GEMMs), the offsets head (row-major PCA epilogue)."""
    sr = 16000
    feat, _, _ = eng.mel_frontend([synth.make_pcm(0, 2 * sr), synth.make_pcm(21, 30011, "speechlike")], sr)
    fc, fs, hop = eng.last_frame_table
    spk = torch.full((feat.shape[0],), 2, dtype=torch.int64)
    ref, z_ref, *_ = eng.forward(feat, spk)
    try:
        eng.set_precision("bf16x6")
        z0, a0 = eng.encoder(feat)
        z1, a1 = eng.encoder(feat, frame_clip=fc, frame_start=fs, hop=hop)
        out, *_ = eng.forward(feat, spk)
    finally:
        eng.set_precision("fp32")
    assert torch.equal(z0, z1) and torch.equal(a0, a1)
    assert float((z0 - z_ref).abs().max()) <= 2e-5 and float((out - ref).abs().max()) <= 5e-6
    e_off = Engine(synth_sd["offsets"], precision="bf16x6")
    o6, *_ = e_off.forward(feat[:100], spk[:100])
    e_off.set_precision("fp32")
    o32, *_ = e_off.forward(feat[:100], spk[:100])
    assert float((o6 - o32).abs().max()) <= 5e-6


@pytest.mark.parametrize("mode", ["bf16x6", "bf16x3"])
def test_split_bf16_frequency_lstm_forms_agree_bitwise(eng, mode):
    """freq_lstm_bf16p_v3_kernel (one persistent workgroup per CU: the default; freq_lstm_shape 8 = one workgroup per tile) makes the
    products of freq_lstm_bf16x6_kernel / freq_lstm_bf16_kernel<3> (freq_lstm_shape 3, two workgroups per CU) in the same order: the
    same bits, whatever the tile-to-workgroup assignment, for a batch that fills the queue several times over and for a single clip."""
    from sdfa_amd import _lib
    rs = np.random.RandomState(12)
    try:
        eng.set_precision(mode)
        for n in (2304, 156):
            x = torch.from_numpy(rs.uniform(0, 1, (n, 64, 128, 3)).astype(np.float32)).cuda()
            got = {}
            for shape in (0, 3, 8, 0):
                _lib.set_option("freq_lstm_shape", shape)
                z, al = eng.encoder(x)
                got.setdefault(shape, []).append((z.clone(), al.clone()))
            z0, a0 = got[0][0]
            for shape, rows in got.items():
                for z, al in rows:
                    assert torch.equal(z, z0) and torch.equal(al, a0), (n, shape)
    finally:
        _lib.set_option("freq_lstm_shape", 0)
        eng.set_precision("fp32")


@pytest.mark.parametrize("mode,tol", [("bf16x6", 2e-5), ("bf16x3", 3e-4), ("bf16", 0.2)])
def test_conv_stack_on_bf16_mfma_against_the_fp32_stack(eng, mode, tol):
    """The mixed-precision modes run the fused conv stack on bf16 MFMA (conv123_bf16_kernel: K axes in the accumulator lanes' channel
    order, pool2 chained in registers); option conv_fp32 = 1 keeps it on the fp32 kernel.  Same mode otherwise: the encoder outputs
    differ by the stack's operand rounding only -- not zero (the bf16 stack did run), and small."""
    from sdfa_amd import _lib
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.uniform(0, 1, (300, 64, 128, 3)).astype(np.float32)).cuda()
    try:
        eng.set_precision(mode)
        _lib.set_option("conv_fp32", 1)
        z0 = eng.encoder(x)[0].clone()
        _lib.set_option("conv_fp32", 0)
        z1 = eng.encoder(x)[0].clone()
    finally:
        _lib.set_option("conv_fp32", 0)
        eng.set_precision("fp32")
    d = float((z0 - z1).abs().max())
    assert 0.0 < d <= tol, (mode, d)


@pytest.mark.parametrize("mode,tol", [("bf16x6", 5e-6), ("bf16x3", 1e-4), ("bf16", 3e-2)])
def test_conv_stack_tap_is_the_bf16_stack(synth_sd, golden, mode, tol):
    """with the debug taps on, a body precision mode must tap the conv stack that SHIPS (conv123_bf16_kernel), not the fp32
    kernels -- so a packing or K-order bug in pack_conv_bf16 shows up at the stage where it happens.  The conv3 tap against the
    reference fixture (saber/nn/layers/conv2d.py:64-97 + extend.py:94-101 outputs of the reference itself), per-mode bounds; with
    conv_fp32 = 1 the same tap is the fp32 stack's, so the two must differ (the bf16 stack did run)."""
    from sdfa_amd import _lib
    g = golden["model_dgrad"]
    x = _t(g["audio_feat"])
    e = Engine(synth_sd["dgrad"], debug_keep=True, precision=mode)
    try:
        e.encoder(x)
        tap = e.tap(1, x.shape[0]).cpu().numpy()
        _lib.set_option("conv_fp32", 1)
        e.encoder(x)
        tap32 = e.tap(1, x.shape[0]).cpu().numpy()
    finally:
        _lib.set_option("conv_fp32", 0)
    ref = g["conv3_f01"]                                                  # frames 0 and 1
    assert tap.shape[1:] == ref.shape[1:]
    assert np.abs(tap32[:2] - ref).max() <= 1e-4
    d = float(np.abs(tap[:2] - ref).max())
    assert d <= tol, (mode, d)
    assert float(np.abs(tap - tap32).max()) > 0.0


def test_dgrad_pca_expansion_on_split_bf16(eng, golden):
    """SDFA_PREC_BF16X3 runs the fused dgrad PCA expansion (speech_anime/modules/output_module.py:94-116) on split-bf16 MFMA too
    (pca_dgrad_res_kernel<true>: bases pre-split into bf16 octets on the host, coefficients split by the lane that loads them); option
    pca_fp32 = 1 keeps it on the fp32 kernel.  Same coefficients either way: the rows differ by the expansion's operand rounding only --
    not zero, small -- on whole tiles, on a ragged tail (partial frame tile) and on the last, partial triangle block; sdfa_expand_coef
    (the peers' rows of the coefficient all-gather) takes the same kernel, so it stays bitwise the regressor's own rows."""
    from sdfa_amd import _lib
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.uniform(0, 1, (300, 64, 128, 3)).astype(np.float32)).cuda()
    spk = torch.from_numpy(rs.randint(0, 8, 300))
    try:
        eng.set_precision("bf16x3")
        z, _ = eng.encoder(x)
        _lib.set_option("pca_fp32", 1)
        c0, o0 = eng.regress(z, spk, want_coef=True)
        o0 = o0.clone()
        _lib.set_option("pca_fp32", 0)
        # the split-bf16 kernel writes into a buffer pre-filled with NaN: an element it skipped (ragged frame tile, last partial triangle
        # block) stays NaN instead of silently keeping the exact row a previous call left in a reused output buffer
        sentinel = torch.full_like(o0, float("nan"))
        c1, o1 = eng.regress(z, spk, want_coef=True, out=sentinel)
        assert o1.data_ptr() == sentinel.data_ptr()
        assert torch.equal(c0, c1)
        assert bool(torch.isfinite(o1).all()), "pca_dgrad_res_kernel<true> left elements unwritten"
        d = (o0 - o1).abs()
        assert 0.0 < float(d.max()) <= 2e-5, float(d.max())
        rows = eng.expand_coef(c1)
        assert torch.equal(rows, o1)
    finally:
        _lib.set_option("pca_fp32", 0)
        eng.set_precision("fp32")


def test_column_sharing_is_exact_in_split_bf16(eng):
    """Column sharing evaluates each distinct column once; a column's arithmetic does not depend on its position in the
    launch, so the shared and unshared encoders agree bitwise in the bf16 modes too."""
    sr = 16000
    clips = [synth.make_pcm(0, 2 * sr), synth.make_pcm(21, 30011, "speechlike")]
    feat, tslists, counts = eng.mel_frontend(clips, sr)
    fc, fs, hop = eng.last_frame_table
    try:
        eng.set_precision("bf16x3")
        z0, a0 = eng.encoder(feat)
        z1, a1 = eng.encoder(feat, frame_clip=fc, frame_start=fs, hop=hop)
    finally:
        eng.set_precision("fp32")
    assert torch.equal(z0, z1) and torch.equal(a0, a1)


def test_unknown_mode_is_rejected(eng):
    from sdfa_amd import _lib
    with pytest.raises(ValueError):
        eng.set_precision("fp8")
    with pytest.raises(_lib.SdfaError):
        _lib.check(_lib.lib.sdfa_model_set_precision(eng._m, 7))
    assert eng.precision == "fp32"

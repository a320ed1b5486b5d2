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

"""Pins the CPU oracle (oracle/sdfa_oracle.py) to fixtures produced by the reference itself
(oracle/gen_golden.py -> tests/golden/).  CPU only."""
import numpy as np
import pytest

import sdfa_oracle as O
from sdfa_amd import synth

TOL_FEAT = 2e-5     # audio_feat in [0,1]; reference STFT is fp32 pocketfft, oracle is fp64 rfft
TOL_ACT = 2e-5
TOL_DGRAD = 1e-5    # restatement within 1e-5 of the reference


def test_frame_index_bit_exact(golden):
    ts = golden["tslist"]
    for key in ts.files:
        sr = int(key.split("_")[0][2:]); L = int(key.split("_L")[1])
        starts, tslist = O.frame_index(L, sr)
        assert np.array_equal(tslist, ts[key]), key
        assert len(starts) == len(ts[key])


def test_frame_index_anchors():
    # SURVEY App. A.1 anchors (measured on the reference)
    s, t = O.frame_index(16000, 8000)
    assert len(t) == 156 and list(t[:5]) == [-117, -100, -83, -67, -50] and list(t[-3:]) == [2433, 2450, 2467]
    for L, sr in ((80000, 8000), (160000, 16000)):
        s, t = O.frame_index(L, sr)
        assert len(t) == 636 and t[-1] == 10467


def test_short_clip_raises_like_reference():
    with pytest.raises(AssertionError):
        O.frame_index(2400, 8000)


@pytest.mark.parametrize("sr", [8000, 16000])
@pytest.mark.parametrize("kind,clip", [("uniform", 0), ("zeros", 1), ("sweep", 2), ("speechlike", 3)])
def test_frontend_matches_reference(golden, sr, kind, clip):
    g = golden["frontend"]
    pcm = synth.make_pcm(clip, 2 * sr, kind)
    out = O.fetch_audio_features(pcm, sr)
    pre = f"sr{sr}_{kind}_"
    assert np.array_equal(np.asarray(out["tslist"]), g[pre + "tslist"])
    feat = out["audio_feat"]
    assert list(feat.shape) == list(g[pre + "shape"])
    keep = g[pre + "frames"]
    err = np.abs(feat[keep] - g[pre + "audio_feat"]).max()
    assert err <= TOL_FEAT, err
    # checksums over ALL frames
    s = feat.astype(np.float64).sum(axis=(1, 2, 3))
    assert np.abs(s - g[pre + "frame_sum"]).max() <= 64 * 128 * 3 * 2e-6


def test_zero_pcm_gives_zero_features():
    out = O.fetch_audio_features(np.zeros(16000, np.float32), 8000)
    assert not out["audio_feat"].any()


def test_model_stages_match_reference(golden, synth_sd):
    g = golden["model_dgrad"]
    orc = O.Oracle(synth_sd["dgrad"], "dgrad")
    st = {}
    dgrad, z, align = orc.forward(g["audio_feat"], int(g["speaker"]), st)
    assert np.abs(st["pool1"][:2] - g["pool1_f01"]).max() <= TOL_ACT
    assert np.abs(st["conv3"][:2] - g["conv3_f01"]).max() <= TOL_ACT
    assert np.abs(st["freq"] - g["freq"]).max() <= TOL_ACT
    assert np.abs(st["bilstm"] - g["bilstm"]).max() <= TOL_ACT
    assert np.abs(align - g["align"][:, 0]).max() <= 1e-6
    assert np.abs(align.sum(-1) - 1).max() <= 1e-6
    assert np.abs(z - g["z"][:, 0]).max() <= TOL_ACT
    assert np.abs(st["trunk"] - g["trunk"][:, 0]).max() <= TOL_ACT
    assert np.abs(st["coef_scale"] - g["coef_scale"][:, 0]).max() <= TOL_ACT
    assert np.abs(st["coef_rotat"] - g["coef_rotat"][:, 0]).max() <= TOL_ACT
    assert dgrad.shape == (8, 89784)
    assert np.abs(dgrad[:2] - g["dgrad_f01"]).max() <= TOL_DGRAD
    assert np.abs(dgrad[:, ::97] - g["dgrad_stride97"]).max() <= TOL_DGRAD
    assert np.abs(dgrad.astype(np.float64).sum(1) - g["dgrad_sum"]).max() <= 89784 * 1e-6


def test_second_speaker(golden, synth_sd):
    g = golden["model_dgrad"]; g5 = golden["model_dgrad_spk5"]
    orc = O.Oracle(synth_sd["dgrad"], "dgrad")
    st = {}
    dgrad, _, _ = orc.forward(g["audio_feat"][:3], 5, st)
    assert np.abs(st["coef_scale"] - g5["coef_scale"][:, 0]).max() <= TOL_ACT
    assert np.abs(dgrad[:, ::97] - g5["dgrad_stride97"]).max() <= TOL_DGRAD


def test_offsets_head(golden, synth_sd):
    g = golden["model_offsets"]; gm = golden["model_dgrad"]
    orc = O.Oracle(synth_sd["offsets"], "offsets")
    st = {}
    off, z, align = orc.forward(gm["audio_feat"][:4], 2, st)
    assert off.shape == (4, 15069)
    assert np.abs(st["coef_scale"] - g["coef"][:, 0]).max() <= TOL_ACT
    assert np.abs(off[0] - g["offsets_f0"]).max() <= TOL_DGRAD
    assert np.abs(off[:, ::7] - g["offsets_stride7"]).max() <= TOL_DGRAD


@pytest.mark.parametrize("sr", [8000, 16000])
def test_end_to_end_generate_animation(golden, synth_sd, sr):
    g = golden["e2e_dgrad"]
    orc = O.Oracle(synth_sd["dgrad"], "dgrad")
    ts, animes = O.generate_animation(orc, synth.make_pcm(0, 2 * sr), sr, 2)
    assert np.array_equal(np.asarray(ts), g[f"sr{sr}_tslist"])
    assert list(animes.shape) == list(g[f"sr{sr}_shape"])
    assert animes.shape[1:] == (9976, 9)          # the reference returns per-triangle rows
    assert np.abs(animes[:, ::97] - g[f"sr{sr}_stride97"]).max() <= 2e-5
    assert np.abs(animes[10] - g[f"sr{sr}_frame10"]).max() <= 2e-5

"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/sdfa_hip.h declares,
and its host-only entry point (frame enumeration) is bit-exact.  No GPU compute calls here."""
import os
import re

import numpy as np
import pytest

import sdfa_oracle as O
from sdfa_amd import _lib, engine, weights, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_bound_and_exported():
    hdr = open(os.path.join(ROOT, "include", "sdfa_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sdfa_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(_lib.lib, name)
    assert _lib.lib.sdfa_abi_version() == _lib.ABI_VERSION == 5


def test_frame_index_matches_reference_fixtures(golden):
    ts = golden["tslist"]
    for key in ts.files:
        sr = int(key.split("_")[0][2:]); L = int(key.split("_L")[1])
        starts, tslist = engine.frame_index(L, sr)
        assert np.array_equal(tslist, ts[key]), key
        assert np.array_equal(starts, O.frame_index(L, sr)[0])


@pytest.mark.parametrize("sr", [8000, 16000])
def test_frame_index_matches_oracle_many_lengths(sr):
    rs = np.random.RandomState(5)
    _, _, sliding = engine.frame_geometry(sr)
    lengths = [sliding, sliding + 1, 10 * sr, 10 * sr + 1, 60 * sr + 7, 600 * sr + 13] + list(rs.randint(sliding, 40 * sr, 40))
    for L in lengths:
        s_c, t_c = engine.frame_index(int(L), sr)
        s_o, t_o = O.frame_index(int(L), sr)
        assert np.array_equal(s_c, s_o) and np.array_equal(t_c, t_o), L


def test_frame_count_of_baseline_clip():
    for L, sr in ((160000, 16000), (80000, 8000)):
        s, t = engine.frame_index(L, sr)
        assert len(s) == 636 and t[-1] == 10467     # SURVEY App. A.1 anchor
    s, t = engine.frame_index(32000, 16000)
    assert len(s) == 156 and list(t[:5]) == [-117, -100, -83, -67, -50]


def test_overlong_clip_is_refused_on_the_host():
    """the front-end kernels index samples in 32-bit arithmetic; a clip beyond 2^29 - 1 samples is refused where its
    length is known (sdfa_frame_index) instead of being truncated silently on the device."""
    from sdfa_amd._lib import SdfaError
    with pytest.raises(SdfaError, match="2\\^29"):
        engine.frame_index(2 ** 29, 16000)


def test_short_clip_raises_like_reference():
    with pytest.raises(AssertionError):
        engine.frame_index(2400, 8000)
    engine.frame_index(9088, 16000)
    # same verdict as the oracle (which follows sliding_window.py:356-363) on every short length
    for sr in (8000, 16000):
        _, _, sliding = engine.frame_geometry(sr)
        for L in range(sliding - 700, sliding + 5, 3):
            try:
                ref = O.frame_index(L, sr)
            except AssertionError:
                ref = None
            if ref is None:
                with pytest.raises(AssertionError):
                    engine.frame_index(L, sr)
            else:
                got = engine.frame_index(L, sr)
                assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


def test_weight_norm_fold_matches_oracle(synth_sd):
    sd = synth_sd["dgrad"]
    folded = weights.fold_state_dict(sd)
    for key in ("_audio_encoder._layers.3", "_output_module._scale_layers.2"):
        ref = O.fold_weight_norm(sd, "_model." + key)
        assert np.array_equal(folded[key + ".weight"], ref)
    assert not any(k.endswith("weight_g") or k.endswith("weight_v") or "num_batches" in k for k in folded)
    assert weights.head_of(sd) == "dgrad" and weights.head_of(synth_sd["offsets"]) == "offsets"


def test_legacy_checkpoint_renames():
    ck = {"state": {"audio_encoder.layers.0.weight_v": 1, "anime_decoder.proj_scale.compT": 2,
                    "time_aggregator.layers.1.b": 3, "audio_encoder.layers.0._ext_batch_norm.weight": 4, "hamm": 0}}
    st = weights.ckpt_backward_compatible_preprocess(ck)["state"]
    assert set(st) == {"_model._audio_encoder._layers.1.weight_v", "_model._output_module._scale_pca.compT",
                       "_model._audio_encoder._layers.10.b", "_model._audio_encoder._layers.1._ext_post_bn.weight"}

"""Host logic of the speech_anime drop-in surface that needs no GPU."""
import json

import numpy as np
import pytest

import librosa_restate as LR
from speech_anime.hparams import configure
from speech_anime import audio, stream
from speech_anime.datasets import DatasetSlidingWindow
from sdfa_amd import synth, engine


def test_configure_defaults_and_json_overwrite(tmp_path):
    hp = configure(dict(mode="evaluate", custom_hparams="dgrad"))
    assert hp.audio.sample_rate == 8000 and hp.anime.fps == 60 and hp.anime.feature.ts_delta == 100
    assert hp.dataset_anime.speakers["m1"] == 2 and hp.model.face_data_type == "dgrad_3d"
    p = tmp_path / "hparams.json"
    p.write_text(json.dumps({"audio": {"sample_rate": 16000}, "trainer": {"evaluate": {"test": [["x.wav"]]}},
                             "ensembling_ms": 30}))
    hp = configure(dict(mode="evaluate", custom_hparams=str(p), eval_spk_cond="f1"))
    assert hp.audio.sample_rate == 16000 and hp.audio.mel.n_mels == 128       # nested overwrite keeps siblings
    assert hp.trainer.evaluate.test == [] and hp.ensembling_ms == 30 and hp.eval_spk_cond == "f1"
    with pytest.raises(ValueError):
        p.write_text(json.dumps({"audio": {"sample_rate": 44100}}))
        configure(dict(mode="evaluate", custom_hparams=str(p)))


def test_unit_converters_are_float32():
    hp = configure(dict(mode="evaluate", custom_hparams="dgrad"))
    DatasetSlidingWindow.hparams = hp
    v = DatasetSlidingWindow.frame_to_sample(7.0)
    assert isinstance(v, np.float32) and v == np.float32(7 * 8000 / 60)
    assert DatasetSlidingWindow.sample_to_ms(4544 / 2) == np.float32(284.0)
    DatasetSlidingWindow.hparams = None


def test_energy_matches_rms_of_padded_windows():
    sr = 8000
    pcm = synth.make_pcm(4, 6000)
    e = DatasetSlidingWindow._energy(pcm, sr)
    win, hop, sliding = engine.frame_geometry(sr)
    starts, _ = engine.frame_index(len(pcm), sr)
    assert e.shape == (len(starts), 1, 64)
    for i in (0, 5, len(starts) // 2, len(starts) - 1):
        s = int(starts[i])
        w = np.zeros(sliding, np.float32)
        a, b = max(0, s), min(len(pcm), s + sliding)
        if b > a:
            w[a - s:b - s] = pcm[a:b]
        assert np.abs(e[i] - LR.rms(w, win, hop)).max() < 1e-6


def test_rms_normalize_hits_target_db():
    x = synth.make_pcm(1, 8000) * 0.1
    y = audio.rms_normalize(x, -24.5)
    assert abs(20 * np.log10(np.sqrt(np.mean(y ** 2))) + 24.5) < 1e-3 and np.abs(y).max() <= 0.999


def test_stream_seek_interpolates():
    ts = [-117, -100, -83, -67]
    seq = np.arange(4, dtype=np.float64)[:, None] * np.ones((1, 3))
    assert np.allclose(stream.seek(-100, ts, seq), 1.0)
    assert np.allclose(stream.seek(-91.5, ts, seq), 1.5)
    assert np.allclose(stream.seek(-500, ts, seq), 0.0) and np.allclose(stream.seek(500, ts, seq), 3.0)


def test_rms_normalize_and_seek_match_reference_fixture(golden):
    g = golden["host_rows"]
    for i in range(3):
        got = audio.rms_normalize(g[f"rms_in_{i}"], -24.5)
        assert np.abs(np.asarray(got, np.float64) - g[f"rms_out_{i}"]).max() <= 1e-7
    ts = [int(t) for t in g["seek_ts"]]
    for q, ref in zip(g["seek_q"], g["seek_out"]):
        assert np.abs(np.asarray(stream.seek(float(q), ts, g["seek_seq"]), np.float64) - ref).max() <= 1e-12


def test_librosa_restatement_reproduces_published_docstring_examples():
    """librosa is absent (requirements.txt pins 0.8.0; un-vendored), so the Slaney mel scale / filterbank are restated in
    oracle/librosa_restate.py.  The reference holds no vectors for them; the only independent anchors are the known answers
    printed in librosa 0.8.0's own docstrings (librosa.core.convert.hz_to_mel / mel_to_hz, librosa.filters.mel):
        >>> librosa.hz_to_mel(60)                      0.9
        >>> librosa.hz_to_mel([110, 220, 440])         array([ 1.65,  3.3 ,  6.6 ])
        >>> librosa.mel_to_hz(3)                       200.
        >>> librosa.mel_to_hz([1,2,3,4,5])             array([  66.667,  133.333,  200.   ,  266.667,  333.333])
        >>> librosa.filters.mel(22050, 2048)           array([[ 0.   ,  0.016, ...,  0.   ,  0.   ], ...   (128 x 1025)
    Weak (a handful of rounded values) but not circular: fixtures, oracle and product all use the restatement."""
    assert abs(LR.hz_to_mel(60) - 0.9) < 1e-12
    assert np.allclose(LR.hz_to_mel(np.array([110, 220, 440])), [1.65, 3.3, 6.6], atol=1e-12)
    assert LR.mel_to_hz(3) == 200.0
    assert np.allclose(LR.mel_to_hz(np.array([1, 2, 3, 4, 5])), [66.667, 133.333, 200.0, 266.667, 333.333], atol=5e-4)
    # above the 1 kHz knee the scale is logarithmic: the knee itself and continuity across it
    assert abs(LR.hz_to_mel(1000.0) - 15.0) < 1e-12 and abs(LR.mel_to_hz(LR.hz_to_mel(4000.0)) - 4000.0) < 1e-9
    fb = LR.mel_filters(22050, 2048)
    assert fb.shape == (128, 1025) and fb.dtype == np.float32
    assert np.array_equal(np.round(fb[0, :2], 3), np.float32([0.0, 0.016])) and fb[0, -1] == 0 and fb[-1, 0] == 0
    # Slaney area normalisation: every band integrates to (about) 2 / bandwidth * bandwidth / 2 = 1 in Hz units
    hz_per_bin = 22050 / 2048
    area = fb.sum(1) * hz_per_bin
    assert np.all(np.abs(area[5:-1] - 1.0) < 0.05)


def test_evaluate_model_hands_rank_device_and_shard_down(tmp_path, monkeypatch):
    """under `torch.distributed.run --nproc-per-node N -m speech_anime evaluate` the process entry point -- and only it --
    turns LOCAL_RANK into the device the Engine is built on and RANK / WORLD_SIZE into the explicit utterance shard; without a
    launcher nothing is sharded and the configured device stands."""
    import torch
    from speech_anime import api
    assert api.rank_device({"LOCAL_RANK": "5"}, 8) == "cuda:5" and api.rank_device({"LOCAL_RANK": "5"}, 4) == "cuda:1"
    assert api.rank_device({"LOCAL_RANK": "1"}, 1) == "cuda:0" and api.rank_device({}, 0) is None
    assert api.shard_from_env({}) == (0, 1) and api.shard_from_env({"RANK": "3", "WORLD_SIZE": "8"}) == (3, 8)
    seen = {}

    class FakeModel:
        current_epoch = 0

        def evaluate(self, sources, **kw):
            seen["shard"], seen["sources"] = kw.get("shard"), sources
            return []

    def fake_build(hparams, state_dict=None):
        seen["device"] = hparams.device
        return FakeModel()

    monkeypatch.setattr(api, "_load_checkpoint", lambda path: {"state": {}, "epoch": 1})
    monkeypatch.setattr(api, "build_model", fake_build)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: seen.__setitem__("set_device", str(d)))
    args = dict(mode="evaluate", load_from="x.ckpt", custom_hparams="dgrad", eval_input="a.wav", output_dir=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    api.evaluate_model(dict(args))
    assert seen["device"] == "cuda:0" and seen["shard"] is None and "set_device" not in seen
    monkeypatch.setenv("RANK", "5"); monkeypatch.setenv("LOCAL_RANK", "5"); monkeypatch.setenv("WORLD_SIZE", "8")
    api.evaluate_model(dict(args))
    assert seen["device"] == "cuda:5" and seen["set_device"] == "cuda:5" and seen["shard"] == (5, 8)

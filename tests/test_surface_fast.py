"""Round 3: the fast surface -- pinned, overlapped device -> host staging (Engine.forward_host), column sharing and the ensembling
mean inside generate_animation, multi-clip generate_animation_batch, small-batch kernel selection, reserved CUs.  Everything here
is a BITWISE statement against the synchronous / single-clip / large-batch forms of the same C-ABI calls, whose parity with the
reference fixtures tests/test_surface_gpu.py and tests/test_gpu_parity.py hold."""
import numpy as np
import pytest
import torch

from speech_anime.hparams import configure
from speech_anime.api import build_model
from speech_anime.datasets import DatasetSlidingWindow
from sdfa_amd import synth
from sdfa_amd.engine import Engine

pytestmark = pytest.mark.gpu


def _model(sd, sr, head="dgrad"):
    hp = configure(dict(mode="evaluate", custom_hparams=head))
    hp.audio.set_key("sample_rate", sr)
    DatasetSlidingWindow.hparams = None
    return hp, build_model(hp, sd)


@pytest.fixture(scope="module")
def eng(synth_sd):
    return Engine(synth_sd["dgrad"], max_frames=512)


def test_forward_host_overlapped_is_bitwise_the_synchronous_path(eng):
    """Pieces of 64 frames, two staging buffers, copies on the copy stream under the next piece's kernels."""
    sr = 16000
    feat, _, counts = eng.mel_frontend([synth.make_pcm(3, 3 * sr), synth.make_pcm(4, 2 * sr)], sr)
    n = feat.shape[0]
    spk = torch.from_numpy(np.repeat(np.asarray([2, 5], np.int64), counts))
    ref, z_ref, _, _ = eng.forward(feat, spk)
    ref = ref.cpu()
    host = eng.forward_host(feat, spk, piece=64)
    assert (not host.is_cuda) and host.is_pinned() and torch.equal(host, ref)
    assert n > 3 * 64                                                     # buffers were reused: at least four pieces
    # column sharing inside the pipeline: same bits
    host2, z = eng.forward_host(feat, spk, table=eng.last_frame_table, piece=64, want_z=True)
    assert torch.equal(host2, ref) and torch.equal(z, z_ref)
    # caller-owned pinned output, deferred wait
    out = torch.empty((n, eng.out_dim), dtype=torch.float32, pin_memory=True)
    r = eng.forward_host(feat, spk, out=out, piece=256, wait=False)
    eng.host_wait()
    assert r is out and torch.equal(out, ref)
    # one piece: the device copy of the rows stays available (evaluate() seeks on it instead of re-uploading)
    eng.forward_host(feat[:100], spk[:100], piece=512)
    assert torch.equal(eng.last_device_rows(100).cpu(), ref[:100])
    assert eng.last_device_rows(99) is None
    with pytest.raises(AssertionError):                                   # pageable output: refused (the copy would not be asynchronous)
        eng.forward_host(feat, spk, out=torch.empty((n, eng.out_dim)))


def test_ensemble_mean_has_numpy_float32_roundings(eng):
    rs = np.random.RandomState(0)
    for n in (1, 3, 4, 89784 * 3 + 2):
        a = (rs.standard_normal(n) * 10.0 ** rs.uniform(-20, 3, n)).astype(np.float32)
        b = (rs.standard_normal(n) * 10.0 ** rs.uniform(-20, 3, n)).astype(np.float32)
        want = a.copy()
        want += b
        want = want / float(2)                                            # model.py:398-403
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        out = torch.empty_like(ta)
        eng.ensemble_mean(ta, tb, out=out)
        assert np.array_equal(out.cpu().numpy(), want)
        eng.ensemble_mean(ta, tb)                                         # in place
        assert np.array_equal(ta.cpu().numpy(), want)


@pytest.mark.parametrize("ens", [0, 20])
def test_fast_generate_animation_is_bitwise_the_two_step_route(synth_sd, ens):
    """Default dataset class = device pipeline (shared columns, device ensembling mean, pinned copy); a caller-supplied dataset
    class = the reference's route (fetch_audio_features to numpy, _feature_to_anime per pass, numpy mean)."""
    sr = 16000
    hp, model = _model(synth_sd["dgrad"], sr)

    class Custom(DatasetSlidingWindow):
        pass

    pcm = synth.make_pcm(7, int(1.3 * sr))
    ts_a, a, oa = model.generate_animation(pcm, "m1", 0, 0, ensembling_ms=ens)
    ts_b, b, ob = model.generate_animation(pcm, "m1", 0, 0, ensembling_ms=ens, dataset_class=Custom)
    assert ts_a == ts_b and a.shape == b.shape == (len(ts_a), 9976, 9) and a.dtype == b.dtype == np.float32
    assert np.array_equal(a, b)
    assert np.array_equal(oa["inputs"], ob["inputs"]) and oa["inputs"].shape == (len(ts_a), 3, 128, 64)
    assert set(oa) == {"inputs", "phones", "latent", "latent_align", "formants"}
    # results stay valid when the next call reuses the pipeline (fresh pinned block per result)
    keep = a.copy()
    model.generate_animation(synth.make_pcm(8, sr), "m1", 0, 0, ensembling_ms=ens)
    assert np.array_equal(a, keep)


def test_generate_animation_batch_equals_clip_by_clip_across_the_kernel_size_switch(synth_sd):
    """One 2 s clip alone takes the small-batch kernels (128^2-tile frequency projection, 32-frame time-LSTM tiles, few persistent
    workgroups); inside a batch of 4,608 frames it takes the 256^2 persistent GEMM and full grids.  Same bits."""
    sr = 16000
    hp, model = _model(synth_sd["dgrad"], sr)
    lens = [2.0, 10.0, 10.0, 1.1, 10.0, 10.0, 10.0, 10.0, 10.0, 3.7]
    clips = [synth.make_pcm(20 + i, int(s * sr), "speechlike" if i % 3 == 0 else "uniform") for i, s in enumerate(lens)]
    spks = ["m1", 0, 7, "f0", 3, 3, "m0", 5, 1, 2]
    res = model.generate_animation_batch(clips, spks)
    assert len(res) == len(clips) and sum(len(r[0]) for r in res) > 4096
    for i in (0, 3, 9, 1):
        ts, a, _ = model.generate_animation(clips[i], spks[i], 0, 0, want_inputs=False)
        assert ts == res[i][0] and np.array_equal(a, res[i][1]), i
    # ensembling in a batch
    res2 = model.generate_animation_batch(clips[:4], spks[:4], ensembling_ms=20, want_inputs=True)
    ts, a, o = model.generate_animation(clips[3], spks[3], 0, 0, ensembling_ms=20)
    assert ts == res2[3][0] and np.array_equal(a, res2[3][1]) and np.array_equal(o["inputs"], res2[3][2]["inputs"])
    with pytest.raises(RuntimeError, match="out of bounds"):
        model.generate_animation_batch(clips[:2], [0, 8])


def test_offsets_head_fast_path(synth_sd):
    sr = 8000
    hp, model = _model(synth_sd["offsets"], sr, "offsets")
    pcm = synth.make_pcm(2, 2 * sr)
    ts, a, _ = model.generate_animation(pcm, 2, 0, 0)
    assert a.shape == (len(ts), 15069)
    eng = model._model._engine
    feat, _, _ = eng.mel_frontend([pcm], sr)
    ref, *_ = eng.forward(feat, torch.full((len(ts),), 2, dtype=torch.int64))
    assert np.array_equal(a, ref.cpu().numpy())


def test_device_speaker_ids_are_validated_without_a_sync(eng):
    z = torch.zeros((2, 512), device="cuda")
    eng.regress(z, torch.tensor([0, 7], device="cuda"))
    eng.check_pending()                                                   # in range: nothing
    eng.regress(z, torch.tensor([0, 9], device="cuda"))                   # clamped on the device, flagged asynchronously
    with pytest.raises(RuntimeError, match="out of bounds"):
        eng.check_pending()
    eng.check_pending()                                                   # reported once
    with pytest.raises(RuntimeError, match="out of bounds"):             # asked for: checked at once
        eng.regress(z, torch.tensor([-1, 3], device="cuda"), check_ids=True)


def test_reserved_cus_do_not_change_results(synth_sd):
    sr = 16000
    e = Engine(synth_sd["dgrad"], max_frames=2048)
    feat, _, _ = e.mel_frontend([synth.make_pcm(i, 10 * sr) for i in range(3)], sr)
    spk = torch.full((feat.shape[0],), 2, dtype=torch.int64)
    ref, z, *_ = e.forward(feat, spk)
    e.set_reserved_cus(24)
    out, z2, *_ = e.forward(feat, spk)
    assert torch.equal(out, ref) and torch.equal(z, z2)
    from sdfa_amd._lib import SdfaError
    with pytest.raises(SdfaError):
        e.set_reserved_cus(200)


def test_split_time_lstm_is_bitwise_the_single_workgroup_form(synth_sd):
    """Small batches: the gate rows of every time-LSTM tile are split over 2 cooperating workgroups that exchange h through global
    memory every step -- 16-frame tiles on v_mfma_f32_16x16x4_f32 up to 1,024 frames (time_lstm_split16_kernel, operands stored in
    the accumulation order of the 32x32x2 kernels), 32-frame tiles up to 2,048 (time_lstm_split_kernel).  Same order of products and
    the same cell arithmetic: not a bit may differ from time_lstm_kernel<1>, at every size, repeatedly, with both hand-off forms (a
    stale hand-off, a lost store or a wrong operand order would show as a difference)."""
    from sdfa_amd import _lib
    e = Engine(synth_sd["dgrad"], max_frames=4096)
    rs = np.random.RandomState(5)
    for n, split, handoff in ((156, 0, 0), (636, 0, 0), (1000, 0, 3), (156, 32, 0), (636, 32, 3), (1500, 0, 0), (1000, 32, 1), (2000, 0, 2), (17, 0, 0)):
        x = torch.from_numpy(rs.uniform(0, 1, (n, 64, 128, 3)).astype(np.float32)).cuda()
        try:
            _lib.set_option("time_lstm_split", 1)                         # never split
            z0, a0 = e.encoder(x)
            _lib.set_option("time_lstm_split", split)                     # 0 = by size (16-frame tiles <= 1024 frames), 32 = 32-frame tiles only
            _lib.set_option("time_lstm_handoff", handoff)
            for rep in range(3):
                z1, a1 = e.encoder(x)
                assert e.time_lstm_repairs() == 0
                assert torch.equal(z0, z1) and torch.equal(a0, a1), (n, split, handoff, rep)
        finally:
            _lib.set_option("time_lstm_split", 0)
            _lib.set_option("time_lstm_handoff", 0)


def test_copy_stream_probe_and_ensembled_pipeline(eng):
    """sdfa_amd/streams.py picks the copy stream by probing (a 128 MiB copy against the encoder on zeros) and logs what it saw; the
    ensembling form of the pipeline (both passes in one launch group, mean on the device) is bitwise two separate passes averaged
    with numpy's float32 roundings, in pieces too."""
    sr = 16000
    pcm = synth.make_pcm(11, int(2.5 * sr))
    pad = 20 * sr // 1000
    clips = [pcm, np.pad(pcm[:-pad], [[pad, 0]], "constant")]
    feat, _, counts = eng.mel_frontend(clips, sr)
    n = counts[0]
    assert counts[1] == n and feat.shape[0] == 2 * n
    spk = torch.full((n,), 4, dtype=torch.int64)
    a = eng.forward_host(feat[:n].contiguous(), spk).numpy().copy()
    b = eng.forward_host(feat[n:].contiguous(), spk).numpy().copy()
    want = a
    want += b
    want = want / float(2)
    got = eng.forward_host(feat, spk, table=eng.last_frame_table, ensemble=True, piece=128)      # 64-frame pieces of both passes
    assert np.array_equal(got.numpy(), want)
    host = eng._host
    assert isinstance(host.copy_overlaps, bool) and len(host.copy_probe) >= 1
    assert all({"priority", "kernels_ms", "copy_end_ms", "overlaps"} <= set(p) for p in host.copy_probe)
    assert host.copy_probe[-1]["overlaps"] == host.copy_overlaps


def test_speaker_sweep_reuses_the_encoder_output(synth_sd):
    """The reference keeps the last signal's features (model.py:364-367,409-416); here the cache holds the speaker-independent
    encoder output z, so the same clip with another speaker re-runs only the regressor.  Bitwise a fresh call."""
    sr = 16000
    hp, model = _model(synth_sd["dgrad"], sr)
    hp2, fresh = _model(synth_sd["dgrad"], sr)
    eng = model._model._engine
    pcm = synth.make_pcm(12, int(2.4 * sr))
    for ens in (0, 20):
        ts0, a0, o0 = model.generate_animation(pcm, "m1", 0, 0, ensembling_ms=ens)
        assert model._signal_cache is not None
        eng.profile(True)
        ts1, a1, o1 = model.generate_animation(pcm.copy(), "f0", 0, 0, ensembling_ms=ens)       # equal content, another array, another speaker
        torch.cuda.synchronize()
        with pytest.raises(Exception):
            eng.profile_ms("freq_lstm")                                   # the encoder did not run
        assert eng.profile_ms("pca") > 0
        eng.profile(False)
        tsf, af, of = fresh.generate_animation(pcm, "f0", 0, 0, ensembling_ms=ens)
        fresh._signal_cache = None
        assert ts1 == ts0 == tsf and np.array_equal(a1, af) and not np.array_equal(a1, a0)
        assert np.array_equal(o1["inputs"], of["inputs"])
        ts2, a2, _ = model.generate_animation(pcm, "m1", 0, 0, ensembling_ms=ens, want_inputs=False)   # back to the first speaker
        assert np.array_equal(a2, a0)
    other = synth.make_pcm(13, int(2.4 * sr))
    ts3, a3, _ = model.generate_animation(other, "m1", 0, 0, ensembling_ms=20, want_inputs=False)      # another signal: recomputed
    tsf, af, _ = fresh.generate_animation(other, "m1", 0, 0, ensembling_ms=20, want_inputs=False)
    assert np.array_equal(a3, af)
    ts4, a4, _ = model.generate_animation(other, "m1", 0, 0, ensembling_ms=0, want_inputs=False)       # same signal, other ensembling: recomputed
    fresh._signal_cache = None
    tsf, af, _ = fresh.generate_animation(other, "m1", 0, 0, ensembling_ms=0, want_inputs=False)
    assert np.array_equal(a4, af)
    # a hit hands out its own timestamp list (a caller may mutate it), a library switch set through sdfa_amd._lib.set_option
    # retires the entry, clear_signal_cache() / hparams.signal_cache = False drop or disable it
    from sdfa_amd import _lib
    ts5, _, _ = model.generate_animation(other, "f1", 0, 0, ensembling_ms=0, want_inputs=False)
    ts5.append(-1)
    ts6, _, _ = model.generate_animation(other, "f2", 0, 0, ensembling_ms=0, want_inputs=False)
    assert ts6 == ts4 and ts6 is not model._signal_cache["tslist"]
    _lib.set_option("gemm_variant", 0)                                    # any switch: the epoch moves, the entry no longer matches
    eng.profile(True)
    model.generate_animation(other, "f3", 0, 0, ensembling_ms=0, want_inputs=False)
    torch.cuda.synchronize()
    assert eng.profile_ms("freq_lstm") > 0                                # recomputed
    eng.profile(False)
    model.clear_signal_cache()
    assert model._signal_cache is None
    hp.set_key("signal_cache", False)
    model.generate_animation(other, "m1", 0, 0, ensembling_ms=0, want_inputs=False)
    assert model._signal_cache is None


def test_evaluate_in_launch_groups_is_the_clip_by_clip_loop(tmp_path, synth_sd):
    """evaluate() (model.py:152-212) takes its sources in launch groups through generate_animation_batch: every file it writes
    is what the one-clip-at-a-time loop writes, bit for bit -- with groups of several clips, a group boundary, and ensembling."""
    from scipy.io import wavfile
    sr = 16000
    hp, model = _model(synth_sd["dgrad"], sr)
    lens = [1.0, 2.3, 0.8, 1.7, 1.2]
    spks = ["m1", "f0", "m0", "m1", "f1"]
    recs = []
    for i, (s, k) in enumerate(zip(lens, spks)):
        w = tmp_path / f"clip{i}.wav"
        wavfile.write(str(w), sr, (synth.make_pcm(60 + i, int(s * sr), "speechlike") * 20000).astype(np.int16))
        recs.append([str(w), f"speaker={k}"])
    for ens in (None, 20):
        a = model.evaluate({"test": recs}, output_dir=str(tmp_path / f"grouped{ens}"), export_mesh_frames=True, ensembling_ms=ens,
                           group_frames=300 if ens is None else 600)      # 96 + 174 | 84 + 138 | 108 frames: three groups (a group holds both passes)
        b = []
        for r in recs:                                # one source per call = the reference's loop
            b += model.evaluate({"test": [r]}, output_dir=str(tmp_path / f"single{ens}"), export_mesh_frames=True, ensembling_ms=ens)
        assert [x[0] for x in a] == [x[0] for x in b] == [r[0] for r in recs]
        for (pa, ta, ra), (pb, tb, rb) in zip(a, b):
            assert list(ta) == list(tb) and np.array_equal(ra, rb)
        import filecmp
        import os
        for i in range(len(recs)):
            da, db = tmp_path / f"grouped{ens}" / f"clip{i}", tmp_path / f"single{ens}" / f"clip{i}"
            names = sorted(os.listdir(da))
            assert names == sorted(os.listdir(db)) and "audio.wav" in names and "000000_dgrad.npy" in names
            match, mismatch, errors = filecmp.cmpfiles(str(da), str(db), names, shallow=False)
            assert not mismatch and not errors, (i, mismatch, errors)


def test_evaluate_shards_sources_across_ranks(tmp_path, synth_sd, monkeypatch):
    """north_star / SURVEY 8(e) on the surface: under torch.distributed.run every rank's evaluate() takes a contiguous block of the
    flat source list (shard=(rank, world), which the process entry point speech_anime.api.evaluate_model derives from RANK / WORLD_SIZE)
    and writes its own sources' files; together the ranks write exactly the files of the single-process run, and no source twice.  The
    LIBRARY call never reads the environment: without `shard` it processes every source, like the reference's evaluate."""
    import filecmp
    import os
    from scipy.io import wavfile
    sr = 16000
    hp, model = _model(synth_sd["dgrad"], sr)
    recs = {"a": [], "b": []}
    for i, s in enumerate((1.0, 1.4, 0.9, 2.1, 1.2)):
        w = tmp_path / f"clip{i}.wav"
        wavfile.write(str(w), sr, (synth.make_pcm(80 + i, int(s * sr), "speechlike") * 20000).astype(np.int16))
        recs["a" if i < 2 else "b"].append([str(w), f"speaker={('m1', 'f0')[i & 1]}"])
    whole = model.evaluate(recs, output_dir=str(tmp_path / "whole"), export_mesh_frames=True)
    assert [os.path.basename(r[0]) for r in whole] == [f"clip{i}.wav" for i in range(5)]
    part0 = model.evaluate(recs, output_dir=str(tmp_path / "sharded"), export_mesh_frames=True, shard=(0, 2))      # clips 0..2
    part1 = model.evaluate(recs, output_dir=str(tmp_path / "sharded"), export_mesh_frames=True, shard=(1, 2))      # clips 3..4
    monkeypatch.setenv("RANK", "1"); monkeypatch.setenv("WORLD_SIZE", "2")                  # inside somebody's torchrun job: still every source
    envd = model.evaluate(recs, output_dir=str(tmp_path / "env"), export_mesh_frames=False)
    monkeypatch.delenv("RANK"); monkeypatch.delenv("WORLD_SIZE")
    assert [r[0] for r in envd] == [r[0] for r in whole]
    assert [r[0] for r in part0] + [r[0] for r in part1] == [r[0] for r in whole] and len(part0) == 3
    for (pa, ta, ra), (pb, tb, rb) in zip(part0 + part1, whole):
        assert list(ta) == list(tb) and np.array_equal(ra, rb)
    for i in range(5):
        da, db = tmp_path / "sharded" / f"clip{i}", tmp_path / "whole" / f"clip{i}"
        names = sorted(os.listdir(da))
        assert names == sorted(os.listdir(db))
        match, mismatch, errors = filecmp.cmpfiles(str(da), str(db), names, shallow=False)
        assert not mismatch and not errors, (i, mismatch, errors)
    assert model.evaluate(recs, output_dir=str(tmp_path / "none"), shard=(7, 8)) == []          # more ranks than sources: an empty shard is fine


def test_default_piece_schedule_is_bitwise(eng):
    """Calls larger than max_frames take Engine.forward_host's default schedule (sdfa_amd/engine.py piece_schedule: uniform pieces of
    3072 frames, the measured optimum of the kernels-then-copy pipeline); the rows do not depend on it."""
    from sdfa_amd.engine import piece_schedule
    assert piece_schedule(20352, 8192) == [3072] * 6 + [1920] and piece_schedule(636, 8192) == [636] and piece_schedule(8192, 8192) == [8192]
    for n, big in ((8193, 8192), (100000, 8192), (3000, 1024), (1, 8), (1025, 1024)):
        s = piece_schedule(n, big)
        assert sum(s) == n and all(0 < x <= min(big, 3072) for x in s)
    sr = 16000
    feat, _, counts = eng.mel_frontend([synth.make_pcm(50 + i, int(s * sr)) for i, s in enumerate((10.0, 6.5, 3.0, 4.2))], sr)
    n = feat.shape[0]
    assert n > 2 * eng.max_frames
    spk = torch.from_numpy(np.repeat(np.asarray([1, 2, 3, 4], np.int64), counts))
    ref, *_ = eng.forward(feat, spk)
    got = eng.forward_host(feat, spk, table=eng.last_frame_table)                      # default schedule: several pieces
    assert torch.equal(got, ref.cpu())
    got2 = eng.forward_host(feat, spk, table=eng.last_frame_table, ensemble=False, piece=300)
    assert torch.equal(got2, got)

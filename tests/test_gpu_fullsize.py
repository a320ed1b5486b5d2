"""BASELINE-size run (32 clips x 10 s @ 16 kHz = 20,352 frames on one GPU) checked through size-independent
properties, plus oracle spot checks on sampled frames."""
import numpy as np
import pytest
import torch

import sdfa_oracle as O
from sdfa_amd import synth
from sdfa_amd.engine import Engine, frame_index

pytestmark = pytest.mark.gpu
SR, L = 16000, 160000


@pytest.fixture(scope="module")
def run(synth_sd):
    eng = Engine(synth_sd["dgrad"], max_frames=8192)
    # 32 clips: clips 0..29 distinct, clip 30 == clip 3 (duplicate), clip 31 = zeros
    pcms = [synth.make_pcm(c, L) for c in range(30)] + [synth.make_pcm(3, L), np.zeros(L, np.float32)]
    feat, tslists, counts = eng.mel_frontend(pcms, SR)
    n = feat.shape[0]
    spk = torch.full((n,), 2, dtype=torch.int64)
    spk[counts[0]:2 * counts[0]] = 5                       # clip 1 with another speaker
    z, align = eng.encoder(feat)
    coef, out = eng.regress(z, spk, want_coef=True)
    return dict(eng=eng, pcms=pcms, feat=feat, ts=tslists, counts=counts, z=z, align=align, coef=coef, out=out, spk=spk)


def test_shapes_and_frame_indexing(run):
    assert sum(run["counts"]) == 20352 and all(c == 636 for c in run["counts"])
    ts_ref = [int(t) for t in frame_index(L, SR)[1]]
    assert all(ts == ts_ref for ts in run["ts"]) and ts_ref[-1] == 10467
    assert run["out"].shape == (20352, 89784)
    assert bool(torch.isfinite(run["out"]).all())


def test_duplicate_clip_is_bitwise_identical(run):
    """Frames are independent: the same PCM in another batch slot (another chunk, another tile) gives the same bits."""
    a = run["out"][3 * 636:4 * 636]
    b = run["out"][30 * 636:31 * 636]
    assert torch.equal(a, b)
    assert torch.equal(run["feat"][3 * 636:4 * 636], run["feat"][30 * 636:31 * 636])


def test_attention_weights_are_distributions(run):
    al = run["align"]
    assert (al >= 0).all() and (al.sum(1) - 1).abs().max().item() <= 1e-5


def test_zero_clip_features_are_zero_and_output_constant(run):
    f = run["feat"][31 * 636:]
    assert not f.any()
    o = run["out"][31 * 636:]
    assert (o - o[0:1]).abs().max().item() == 0.0          # identical input columns -> identical frames


def test_pca_stage_is_affine_in_the_coefficients(run, synth_sd):
    """out = B @ coef + mean exactly as PcaInversion defines it: checked on every frame by two checksums."""
    sd = synth_sd["dgrad"]
    P = "_model._output_module."
    cs, ms = sd[P + "_scale_pca.compT"].astype(np.float64), sd[P + "_scale_pca.means"].astype(np.float64)
    cr, mr = sd[P + "_rotat_pca.compT"].astype(np.float64), sd[P + "_rotat_pca.means"].astype(np.float64)
    coef = run["coef"].cpu().numpy().astype(np.float64)
    out = run["out"]
    tri = out.view(out.shape[0], 9976, 9)
    s_sum = tri[:, :, :6].double().sum((1, 2)).cpu().numpy()
    r_sum = tri[:, :, 6:].double().sum((1, 2)).cpu().numpy()
    assert np.abs(s_sum - (coef[:, :85] @ cs.sum(0) + ms.sum())).max() <= 59856 * 2e-6
    assert np.abs(r_sum - (coef[:, 85:] @ cr.sum(0) + mr.sum())).max() <= 29928 * 2e-6


def test_sampled_frames_match_oracle(run, synth_sd):
    """Oracle on 24 frames spread over clips, chunks and the second speaker: <= 1e-4 on dgrad."""
    orc = O.Oracle(synth_sd["dgrad"], "dgrad")
    idx = np.r_[0, 1, 635, 636 + 7, 2 * 636 - 1, 8191, 8192, 8193, 16383, 16384, np.linspace(9000, 20351, 14).astype(int)]
    feat_ref = {}
    for i in idx:
        c, f = divmod(int(i), 636)
        if c not in feat_ref:
            feat_ref[c] = O.fetch_audio_features(run["pcms"][c], SR)["audio_feat"]
        assert np.abs(run["feat"][i].cpu().numpy() - feat_ref[c][f]).max() <= 5e-5
    x = np.stack([feat_ref[divmod(int(i), 636)[0]][divmod(int(i), 636)[1]] for i in idx])
    ref, _, _ = orc.forward(x, run["spk"][idx].numpy())
    got = run["out"][torch.from_numpy(idx)].cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-4


def test_column_sharing_at_full_size(run):
    eng = run["eng"]
    fc, fs, hop = eng.last_frame_table
    z2 = torch.empty_like(run["z"])
    for f0 in range(0, 20352, 8192):
        f1 = min(20352, f0 + 8192)
        z2[f0:f1], _ = eng.encoder(run["feat"][f0:f1], want_align=False, frame_clip=fc[f0:f1], frame_start=fs[f0:f1], hop=hop)
    assert torch.equal(z2, run["z"])            # bitwise: shared columns are identical feature vectors

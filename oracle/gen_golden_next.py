#!/usr/bin/env python3
"""Round-2 fixtures for the SURVEY section 8(f) next rows, produced by the REFERENCE ITSELF (build container only;
TEST INFRASTRUCTURE ONLY):

  ensembling.npz   SaberSpeechDrivenAnimation.generate_animation(..., ensembling_ms=20)   speech_anime/model/model.py:369-403
  seek_track.npz   saber.stream.seek at the video-rate queries of model.py:204-212        saber/data/stream/stream.py:20-46
                   (imported reference module; real 2 s timestamps + synthetic timestamp lists that put queries before the
                   first timestamp, on exact hits and past the end)
  mesh_flame.npz   deformation.get_mesh on the FLAME template the reference ships         oracle/_ref (deformation/cpp/src)
                   (speech_anime/datasets/vocaset/template/FLAME_sample.obj + mask/non_face.py, as viewer/frame.py:33 uses)
  mesh_corres.npz  deformation.set_target(corrs=...) / get_mesh(corr_count, corr_faces)    deform_triangle_impl.hpp:12-21,248-266
                   on a synthetic target of another topology

    cd /root/repo && python3 -B oracle/gen_golden_next.py
Only data (inputs, seeds, expected outputs) is written.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "sdfa-2019_amd"))
OUT = os.path.join(ROOT, "tests", "golden")

import ref_import  # noqa: E402
from sdfa_amd import synth  # noqa: E402
from gen_golden import load_with_weights  # noqa: E402
from gen_golden_mesh import synthetic_mesh  # noqa: E402


def flame_dgrad(seed, sigma, n_tris=9976):
    """The dgrad rows of the FLAME fixture are regenerated from (seed, sigma) by the tests: 9 floats per triangle."""
    return np.random.RandomState(seed).normal(0, sigma, (n_tris, 9)).astype(np.float32)


def main():
    import torch
    torch.set_num_threads(8)
    ref_import.install_stubs()
    import saber

    # ---------------------------------------------------------------- ensembling (f4)
    ens = {}
    for sr in (8000, 16000):
        hp, model, DS = load_with_weights("dgrad", sr)
        pcm = synth.make_pcm(0, 2 * sr)
        ts, animes, _ = model.generate_animation(pcm, "m1", 0, 0, ensembling_ms=20, dataset_class=DS)
        animes = np.asarray(animes, np.float32)
        ens[f"sr{sr}_tslist"] = np.asarray(ts, np.int64)
        ens[f"sr{sr}_shape"] = np.asarray(animes.shape, np.int64)
        ens[f"sr{sr}_stride97"] = animes.reshape(len(animes), -1)[:, ::97].copy()
        ens[f"sr{sr}_sum"] = animes.astype(np.float64).reshape(len(animes), -1).sum(1)
        ens[f"sr{sr}_frame10"] = animes[10].copy()
        if sr == 16000:
            track16 = (list(ts), animes.reshape(len(animes), -1))
    np.savez_compressed(os.path.join(OUT, "ensembling.npz"), **ens)

    # ---------------------------------------------------------------- seek at video rate (f3)
    sk = {}
    fps_cases = []
    ts_real, seq_real = track16
    rs = np.random.RandomState(17)
    cases = [("real2s", ts_real, seq_real[:, ::211].copy(), 60.0),                      # 156 frames, queries 0..int(ts[-1]*60/1000)
             ("late_start", [40, 57, 73, 90, 107, 123, 140], None, 60.0),               # queries 0, 16.7, 33.3 lie before the first timestamp
             ("fps25", [-117, -100, -83, -67, -50, -33, -17, 0, 17, 33, 50, 67, 83, 100, 117, 133, 150, 167, 183, 200], None, 25.0),  # 40 ms queries: exact hits at 0, 200
             ("single", [5], None, 60.0),
             ("fps30_irregular", [0, 11, 29, 64, 65, 130, 131, 200, 333, 334, 400], None, 30.0)]
    for name, ts, seq, fps in cases:
        ts = [int(t) for t in ts]
        if seq is None:
            seq = rs.normal(0, 1, (len(ts), 13)).astype(np.float32)
        n_q = int(ts[-1] * fps / 1000.0) + 1 + 3                 # three queries PAST model.py's range: beyond the last timestamp
        out = np.stack([np.asarray(saber.stream.seek(i * 1000.0 / fps, ts, seq)) for i in range(n_q)])
        assert out.dtype == np.float32, out.dtype                 # numpy keeps float32 for python-float * float32-array
        sk[f"{name}_ts"] = np.asarray(ts, np.int32); sk[f"{name}_seq"] = seq; sk[f"{name}_fps"] = np.asarray(fps)
        sk[f"{name}_out"] = out
        fps_cases.append(name)
    sk["cases"] = np.asarray(fps_cases)
    np.savez_compressed(os.path.join(OUT, "seek_track.npz"), **sk)

    # ---------------------------------------------------------------- mesh: FLAME topology + correspondences (f1)
    subprocess.check_call(["bash", os.path.join(HERE, "build_ref.sh")])
    import glob
    import importlib.util
    # by path: ref_import's stubs hold an inert `deformation` in sys.modules (the reference builds its module at import)
    spec = importlib.util.spec_from_file_location("deformation", glob.glob(os.path.join(HERE, "_ref", "deformation*.so"))[0])
    D = importlib.util.module_from_spec(spec); spec.loader.exec_module(D)
    ref_root = os.environ.get("SDFA_REFERENCE_ROOT", "/root/reference")
    Vf, Ff = [], []
    for line in open(os.path.join(ref_root, "speech_anime/datasets/vocaset/template/FLAME_sample.obj")):
        p = line.split()
        if p and p[0] == "v":
            Vf.append([float(x) for x in p[1:4]])
        elif p and p[0] == "f":
            Ff.append([int(x.split("/")[0]) - 1 for x in p[1:4]])
    Vf, Ff = np.asarray(Vf, np.float32), np.asarray(Ff, np.uint32)
    spec = importlib.util.spec_from_file_location("non_face", os.path.join(ref_root, "speech_anime/datasets/vocaset/mask/non_face.py"))
    nf = importlib.util.module_from_spec(spec); spec.loader.exec_module(nf)
    cf = np.asarray(nf.non_face_verts, np.uint32)
    assert D.set_target(Vf, Ff, cf)
    specs = [(101, 0.02), (102, 0.05), (103, 0.1), (104, 0.2), (105, 0.0)]      # (seed, sigma); the last is the zero dgrad
    rows = [flame_dgrad(s, g, len(Ff)) for s, g in specs]
    rows[3][:, :6] = 0.0                                                         # seed 104: rotations only, up to ~0.7 rad
    verts = np.stack([D.get_mesh(r.astype(np.float64).reshape(-1), Vf[cf]) for r in rows]).astype(np.float32)
    # a seek-blended frame, as evaluate() produces it: float32(a) * row0 + float32(1 - a) * row1 in float32, then get_mesh
    a = (133 - 125.0) / (133 - 117)
    blend = (a * rows[1] + (1 - a) * rows[2])
    assert blend.dtype == np.float32
    v_blend = D.get_mesh(blend.astype(np.float64).reshape(-1), Vf[cf]).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "mesh_flame.npz"), verts=Vf, faces=Ff, cnsts=cf,
                        dgrad_seed=np.asarray([s for s, _ in specs]), dgrad_sigma=np.asarray([g for _, g in specs]),
                        rot_only=np.asarray(3), mesh=verts, blend_rows=np.asarray([1, 2]), blend_ts=np.asarray([117, 133, 125]),
                        blend_mesh=v_blend)

    # correspondences: target = synthetic torus (392 verts, 784 tris); source rows hold 500 triangles' 9-vectors
    V, F, cn = synthetic_mesh()
    rs = np.random.RandomState(23)
    n_src = 500
    count = rs.choice([0, 1, 1, 2, 3], len(F)).astype(np.uint32)
    faces_c = []
    for c in count:
        faces_c += [0] if c == 0 else list(rs.randint(0, n_src, c))
    faces_c = np.asarray(faces_c, np.uint32)
    assert D.set_target(V, F, cn, count)
    dg = np.stack([rs.normal(0, s, (n_src, 9)) for s in (0.0, 0.03, 0.1, 0.25)]).astype(np.float32)
    out = np.stack([D.get_mesh(d.astype(np.float64).reshape(-1), V[cn], count, faces_c) for d in dg]).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "mesh_corres.npz"), verts=V, faces=F, cnsts=cn, corr_count=count, corr_faces=faces_c,
                        n_src_tris=np.asarray(n_src), dgrad=dg, mesh=out)
    for fn in ("ensembling.npz", "seek_track.npz", "mesh_flame.npz", "mesh_corres.npz"):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()

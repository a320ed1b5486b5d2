"""device seek (f3) fused with the mesh solve (f1) at FLAME size, triangle
correspondences.  Fixtures come from the reference itself (oracle/gen_golden_next.py): saber.stream.seek imported from the
reference, deformation.get_mesh from the reference's own compiled module (oracle/_ref)."""
import numpy as np
import pytest

import sdfa_oracle as O
from mesh_oracle import MeshOracle


def flame_rows(g):
    rows = [np.random.RandomState(int(s)).normal(0, float(sig), (len(g["faces"]), 9)).astype(np.float32)
            for s, sig in zip(g["dgrad_seed"], g["dgrad_sigma"])]
    rows[int(g["rot_only"])][:, :6] = 0.0
    return np.stack(rows)


# ------------------------------------------------------------------------------------------------ oracle pins (CPU)
def test_oracle_seek_matches_reference_track(golden):
    g = golden["seek_track"]
    for name in g["cases"]:
        ts, seq, fps, ref = g[f"{name}_ts"], g[f"{name}_seq"], float(g[f"{name}_fps"]), g[f"{name}_out"]
        out = O.seek_track(ts, seq, fps, n_queries=len(ref))
        assert out.dtype == np.float32 and np.array_equal(out, ref), name          # integer search + one fp32 lerp: bit-exact


def test_host_seek_mirror_matches_reference_track(golden):
    from speech_anime import stream
    g = golden["seek_track"]
    for name in g["cases"]:
        ts, seq, fps, ref = [int(t) for t in g[f"{name}_ts"]], g[f"{name}_seq"], float(g[f"{name}_fps"]), g[f"{name}_out"]
        out = np.stack([np.asarray(stream.seek(i * 1000.0 / fps, ts, seq)) for i in range(len(ref))])
        assert np.array_equal(out, ref), name


def test_mesh_oracle_matches_reference_module_at_flame_size(golden):
    g = golden["mesh_flame"]
    assert g["verts"].shape == (5023, 3) and g["faces"].shape == (9976, 3)
    orc = MeshOracle(g["verts"], g["faces"], g["cnsts"])
    rows = flame_rows(g)
    for k in (1, 3, 4):
        assert np.abs(orc.get_mesh(rows[k]) - g["mesh"][k]).max() <= 2e-7, k
    assert np.abs(g["mesh"][4] - g["verts"]).max() <= 1e-7                          # zero dgrad -> the template


def test_mesh_oracle_correspondences_match_reference_module(golden):
    g = golden["mesh_corres"]
    orc = MeshOracle(g["verts"], g["faces"], g["cnsts"], corr_count=g["corr_count"], corr_faces=g["corr_faces"])
    for d, ref in zip(g["dgrad"], g["mesh"]):
        assert np.abs(orc.get_mesh(d) - ref).max() <= 2e-7
    assert (g["corr_count"] == 0).any() and (g["corr_count"] > 1).any()


# ------------------------------------------------------------------------------------------------ device (GPU)
@pytest.mark.gpu
def test_device_seek_is_bit_identical_to_reference_seek(golden):
    import torch
    from sdfa_amd import seek as S
    g = golden["seek_track"]
    names = [str(n) for n in g["cases"]]
    for name in names:
        ts, seq, fps, ref = g[f"{name}_ts"], g[f"{name}_seq"], float(g[f"{name}_fps"]), g[f"{name}_out"]
        plan = S.SeekPlan([ts], fps)
        assert plan.n_queries == int(int(ts[-1]) * fps / 1000.0) + 1 == len(ref) - 3      # model.py:205-207
        out = plan.rows(torch.from_numpy(seq).cuda()).cpu().numpy()
        assert np.array_equal(out, ref[:plan.n_queries]), name
    # queries past the last timestamp (outside model.py's range, inside stream.seek's contract): the plan kernel through
    # the C ABI with a longer query range per clip; and all clips batched in one plan with different widths impossible ->
    # batch the 13-column cases
    import ctypes as C
    from sdfa_amd._lib import lib, check
    for name in names:
        ts, seq, fps, ref = g[f"{name}_ts"], g[f"{name}_seq"], float(g[f"{name}_fps"]), g[f"{name}_out"]
        nq = len(ref)
        d_ts = torch.from_numpy(ts.astype(np.int32)).cuda()
        d_fo = torch.tensor([0, len(ts)], dtype=torch.int64, device="cuda"); d_qo = torch.tensor([0, nq], dtype=torch.int64, device="cuda")
        src = torch.empty((nq, 2), dtype=torch.int64, device="cuda"); w = torch.empty((nq, 2), dtype=torch.float32, device="cuda")
        p = lambda t: C.c_void_p(t.data_ptr())
        check(lib.sdfa_seek_plan(p(d_ts), p(d_fo), p(d_qo), 1, fps, nq, p(src), p(w), None))
        rows = torch.from_numpy(seq).cuda()
        out = torch.empty((nq, seq.shape[1]), device="cuda")
        check(lib.sdfa_seek_rows(p(rows), seq.shape[1], p(src), p(w), nq, p(out), None))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), ref), name
    small = [n for n in names if g[f"{n}_seq"].shape[1] == 13 and float(g[f"{n}_fps"]) == 60.0]
    plan = S.SeekPlan([g[f"{n}_ts"] for n in small], 60.0)
    out = plan.rows(torch.from_numpy(np.concatenate([g[f"{n}_seq"] for n in small])).cuda()).cpu().numpy()
    want = np.concatenate([g[f"{n}_out"][:-3] for n in small])
    assert np.array_equal(out, want)


@pytest.mark.gpu
def test_gpu_mesh_at_flame_size_vs_reference_module(golden):
    import torch
    from sdfa_amd.mesh import MeshSolver
    g = golden["mesh_flame"]
    ms = MeshSolver(g["verts"], g["faces"], g["cnsts"])
    rows = flame_rows(g)
    out = ms.get_mesh(torch.from_numpy(rows).cuda()).cpu().numpy()
    assert out.shape == (5, 5023, 3)
    err = np.abs(out - g["mesh"]).reshape(5, -1).max(1)
    assert err.max() <= 2e-6, err                                                   # coordinates O(0.1), float32 output
    assert np.abs(out[4] - g["verts"]).max() <= 1e-9                                # zero dgrad: the template (regulariser term ~1e-16)
    cn = g["cnsts"]
    assert np.array_equal(out[:, cn], np.broadcast_to(g["verts"][cn], out[:, cn].shape))


def test_default_constraints_are_the_references_non_face_vertices(golden):
    """speech_anime/viewer/frame.py:33: no --mesh_constraints file -> non_face.non_face_verts.  The FLAME fixture was made by the
    reference with exactly that list (oracle/gen_golden_next.py), so it pins the product's generated table."""
    from speech_anime.datasets.vocaset_mask import non_face_verts
    assert np.array_equal(non_face_verts(), np.asarray(golden["mesh_flame"]["cnsts"], np.int64))


@pytest.mark.gpu
def test_viewer_without_constraints_file_solves_the_references_default_system(golden):
    """set_dgrad_static(verts, faces) with c_indices=None must pin the non-face vertices like the reference does --
    the .obj vertices then equal what the reference module produced with its default constraints."""
    import torch
    from speech_anime import viewer
    g = golden["mesh_flame"]
    viewer.set_dgrad_static(g["verts"], g["faces"])                                 # no constraints given
    verts, _ = viewer.frames_to_mesh(torch.from_numpy(flame_rows(g)).cuda().reshape(5, -1), "dgrad_3d")
    assert np.abs(verts - g["mesh"]).max() <= 2e-6
    cn = g["cnsts"]
    assert np.array_equal(verts[:, cn], np.broadcast_to(g["verts"][cn], verts[:, cn].shape))


@pytest.mark.gpu
def test_seek_fused_into_mesh_solve(golden):
    """One post-path stage: uniform query -> binary search -> lerp of two dgrad rows -> rhs / GEMM / scatter, the blended
    dgrad never written.  Against (1) the reference module on the reference-blended row, (2) the two-step device path, bitwise."""
    import torch
    from sdfa_amd.mesh import MeshSolver
    from sdfa_amd import seek as S
    g = golden["mesh_flame"]
    ms = MeshSolver(g["verts"], g["faces"], g["cnsts"])
    rows = torch.from_numpy(flame_rows(g)).cuda().reshape(5, -1)
    r0, r1 = (int(x) for x in g["blend_rows"])
    t0, t1, q = (int(x) for x in g["blend_ts"])
    # a two-frame clip at 8 fps: query 1 is 125 ms, between the timestamps 117 and 133 of the fixture
    plan = S.SeekPlan([[t0, t1]], 8.0)
    assert plan.n_queries == 2 and 1 * 1000.0 / 8.0 == q
    pair = torch.stack([rows[r0], rows[r1]])
    fused = ms.get_mesh_seek(pair, plan)
    assert np.abs(fused[1].cpu().numpy() - g["blend_mesh"]).max() <= 2e-6
    two_step = ms.get_mesh(plan.rows(pair))
    assert torch.equal(fused, two_step)
    assert np.abs(fused[0].cpu().numpy() - g["mesh"][r0]).max() <= 2e-6             # query 0 precedes the first timestamp: row 0


@pytest.mark.gpu
def test_gpu_mesh_triangle_correspondences(golden):
    import torch
    from sdfa_amd.mesh import MeshSolver
    g = golden["mesh_corres"]
    ms = MeshSolver(g["verts"], g["faces"], g["cnsts"], corr_count=g["corr_count"], corr_faces=g["corr_faces"], n_src_tris=int(g["n_src_tris"]))
    out = ms.get_mesh(torch.from_numpy(g["dgrad"]).cuda()).cpu().numpy()
    assert out.shape == g["mesh"].shape
    assert np.abs(out - g["mesh"]).max() <= 2e-6, np.abs(out - g["mesh"]).max()
    assert np.abs(out[0] - g["verts"]).max() <= 1e-9                                # zero dgrad: identity everywhere -> template
    from sdfa_amd._lib import SdfaError
    with pytest.raises(SdfaError):                                                  # corr_faces of the wrong length
        MeshSolver(g["verts"], g["faces"], g["cnsts"], corr_count=g["corr_count"], corr_faces=g["corr_faces"][:-1], n_src_tris=500)

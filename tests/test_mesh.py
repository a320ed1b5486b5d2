"""Next row: dgrad -> mesh.  Oracle pinned to the reference's compiled module (fixtures); GPU solver vs both."""
import numpy as np
import pytest

from mesh_oracle import MeshOracle


def test_oracle_matches_reference_module(golden):
    g = golden["mesh"]
    orc = MeshOracle(g["verts"], g["faces"], g["cnsts"])
    for d, ref in zip(g["dgrad"], g["mesh"]):
        assert np.abs(orc.get_mesh(d) - ref).max() <= 1e-7
    assert np.abs(orc.get_mesh(np.zeros_like(g["dgrad"][0])) - g["verts"]).max() <= 1e-7     # zero dgrad -> template


@pytest.mark.gpu
def test_gpu_mesh_matches_reference_fixture(golden):
    import torch
    from sdfa_amd.mesh import MeshSolver
    g = golden["mesh"]
    ms = MeshSolver(g["verts"], g["faces"], g["cnsts"])
    assert ms.is_same(len(g["verts"]), len(g["faces"]), len(g["cnsts"]))
    out = ms.get_mesh(torch.from_numpy(g["dgrad"]).cuda()).cpu().numpy()
    assert out.shape == g["mesh"].shape
    assert np.abs(out - g["mesh"]).max() <= 2e-6, np.abs(out - g["mesh"]).max()    # coordinates are O(0.1); float32 output
    assert np.abs(out[0] - g["verts"]).max() <= 1e-9                               # zero dgrad: the template (up to the reg * Inv * x_t of the regularised system, ~1e-16)
    cn = g["cnsts"]
    assert np.array_equal(out[:, cn], np.broadcast_to(g["verts"][cn], out[:, cn].shape))   # constraints pinned
    one = ms.get_mesh(g["dgrad"][2])                                                # single-frame numpy call
    assert np.abs(one.cpu().numpy() - g["mesh"][2]).max() <= 2e-6


@pytest.mark.gpu
def test_gpu_mesh_batch_vs_oracle_random_mesh():
    import torch
    from sdfa_amd.mesh import MeshSolver
    rs = np.random.RandomState(5)
    # open grid patch 40 x 30, border vertices constrained
    nx, ny = 40, 30
    x, y = np.meshgrid(np.arange(nx) * 0.01, np.arange(ny) * 0.01, indexing="ij")
    V = np.stack([x, y, 0.02 * np.sin(7 * x) * np.cos(5 * y)], -1).reshape(-1, 3).astype(np.float32)
    V += rs.normal(0, 1e-3, V.shape).astype(np.float32)
    idx = lambda i, j: i * ny + j
    F = np.asarray([[idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)] for i in range(nx - 1) for j in range(ny - 1)] +
                   [[idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)] for i in range(nx - 1) for j in range(ny - 1)], np.uint32)
    cn = np.asarray([idx(i, j) for i in range(nx) for j in range(ny) if i in (0, nx - 1) or j in (0, ny - 1)], np.uint32)
    orc = MeshOracle(V, F, cn)
    ms = MeshSolver(V, F, cn)
    dg = rs.normal(0, 0.08, (130, len(F), 9)).astype(np.float32)           # 130 frames: 3*130 is no tile multiple
    out = ms.get_mesh(torch.from_numpy(dg).cuda()).cpu().numpy()
    for k in (0, 1, 64, 129):
        assert np.abs(out[k] - orc.get_mesh(dg[k])).max() <= 5e-6

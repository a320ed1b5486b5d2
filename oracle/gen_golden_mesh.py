#!/usr/bin/env python3
"""Fixtures for the dgrad -> mesh next row, produced by the REFERENCE'S OWN compiled module (oracle/_ref, see
oracle/build_ref.sh).  Build container only.  A synthetic closed mesh is used so that no licensed FLAME geometry
is committed; the FLAME template is exercised too but only error statistics are recorded (META)."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
subprocess.check_call(["bash", os.path.join(HERE, "build_ref.sh")])
sys.path.insert(0, os.path.join(HERE, "_ref"))
import deformation as D  # noqa: E402  (the reference's native module)
from mesh_oracle import MeshOracle  # noqa: E402


def synthetic_mesh(nu=28, nv=14, seed=3):
    """Bumpy torus-like closed surface: nu*nv vertices, 2*nu*nv triangles."""
    rs = np.random.RandomState(seed)
    u, v = np.meshgrid(np.arange(nu) * 2 * np.pi / nu, np.arange(nv) * 2 * np.pi / nv, indexing="ij")
    r = 0.04 + 0.008 * rs.uniform(-1, 1, u.shape)
    x = (0.1 + r * np.cos(v)) * np.cos(u); y = (0.1 + r * np.cos(v)) * np.sin(u); z = r * np.sin(v) * 1.3
    V = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    idx = lambda i, j: (i % nu) * nv + (j % nv)
    F = []
    for i in range(nu):
        for j in range(nv):
            F.append([idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)])
            F.append([idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)])
    cn = np.sort(rs.choice(len(V), int(0.6 * len(V)), replace=False)).astype(np.uint32)
    return V, np.asarray(F, np.uint32), cn


def main():
    V, F, cn = synthetic_mesh()
    assert D.set_target(V, F, cn)
    rs = np.random.RandomState(11)
    n = 6
    dg = np.zeros((n, len(F), 9))
    dg[1] = rs.normal(0, 0.02, (len(F), 9))
    dg[2] = rs.normal(0, 0.1, (len(F), 9))
    dg[3, :, 6:] = rs.normal(0, 0.8, (len(F), 3))          # large rotations only
    dg[4, :, :6] = rs.normal(0, 0.2, (len(F), 6))          # scale / shear only
    dg[5] = rs.normal(0, 0.05, (len(F), 9)); dg[5, ::3, 6:] = 0.0   # exact-zero rotation on a third of the triangles
    out = np.stack([D.get_mesh(d.reshape(-1), V[cn]) for d in dg]).astype(np.float32)
    orc = MeshOracle(V, F, cn)
    err = max(np.abs(orc.get_mesh(d) - o).max() for d, o in zip(dg, out))
    print("oracle vs reference module (synthetic):", err)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "mesh.npz"), verts=V, faces=F, cnsts=cn,
                        dgrad=dg.astype(np.float32), mesh=out)
    meta = {"synthetic_oracle_vs_ref": float(err)}
    # FLAME template (reference asset, NOT committed): statistics only
    tp = "/root/reference/speech_anime/datasets/vocaset/template/FLAME_sample.obj"
    if os.path.exists(tp):
        Vf, Ff = [], []
        for line in open(tp):
            p = line.split()
            if p and p[0] == "v": Vf.append([float(x) for x in p[1:4]])
            elif p and p[0] == "f": Ff.append([int(x.split("/")[0]) - 1 for x in p[1:4]])
        Vf, Ff = np.asarray(Vf, np.float32), np.asarray(Ff, np.uint32)
        import importlib.util
        spec = importlib.util.spec_from_file_location("non_face", "/root/reference/speech_anime/datasets/vocaset/mask/non_face.py")
        nf = importlib.util.module_from_spec(spec); spec.loader.exec_module(nf)
        cf = np.asarray(nf.non_face_verts, np.uint32)
        assert D.set_target(Vf, Ff, cf)
        d = rs.normal(0, 0.05, (len(Ff), 9))
        ref = D.get_mesh(d.reshape(-1), Vf[cf])
        zero = D.get_mesh(np.zeros(len(Ff) * 9), Vf[cf])
        meta.update(flame_verts=int(len(Vf)), flame_tris=int(len(Ff)), flame_cnsts=int(len(cf)),
                    flame_zero_dgrad_err=float(np.abs(zero - Vf).max()),
                    flame_oracle_vs_ref=float(np.abs(MeshOracle(Vf, Ff, cf).get_mesh(d) - ref).max()))
    with open(os.path.join(ROOT, "tests", "golden", "META_mesh.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(meta)


if __name__ == "__main__":
    main()

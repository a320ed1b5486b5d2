"""CPU restatement (numpy, fp64) of the reference's dgrad -> mesh solve.  TEST INFRASTRUCTURE ONLY.

Follows deformation/cpp/src/deform_triangle_impl.hpp: setStaticTarget (:8-140) and getMeshFromDeformationGradients
(:215-310), with rotation_log_exp::exp (rotation/utils_rotation.cpp:33-49).  Pinned against the reference's own
compiled module (oracle/_ref, built by oracle/build_ref.sh) through tests/golden/mesh.npz.
"""
import numpy as np


class MeshOracle:
    def __init__(self, verts, faces, cnsts, reg=1e-10, corr_count=None, corr_faces=None):
        V = np.asarray(verts, np.float32).reshape(-1, 3)
        F = np.asarray(faces, np.int64).reshape(-1, 3)
        cn = np.asarray(cnsts, np.int64).reshape(-1)
        self.V, self.F, self.cn = V, F, cn
        # triangle correspondences (:16-21,102): target triangle j contributes max(1, count[j]) equations
        self.count = None if corr_count is None else np.asarray(corr_count, np.int64).reshape(-1)
        self.corr_faces = None if corr_faces is None else np.asarray(corr_faces, np.int64).reshape(-1)
        reps = np.ones(len(F), int) if self.count is None else np.maximum(1, self.count)
        free = np.setdiff1d(np.arange(len(V)), cn)
        col = -np.ones(len(V), int); col[free] = np.arange(len(free))
        ccol = -np.ones(len(V), int); ccol[cn] = np.arange(len(cn))
        E = int(reps.sum())
        A = np.zeros((3 * E, len(free))); Ar = np.zeros((3 * E, max(len(cn), 1)))
        k = 0
        for j, (a, b, c) in enumerate(F):
            Va = np.stack([V[b] - V[a], V[c] - V[a]], 1).astype(np.float64)      # float32 subtraction, :96-97
            Q, R = np.linalg.qr(Va)                                              # :98-100
            U = np.linalg.inv(R) @ Q.T
            for _ in range(reps[j]):                                             # :102-117
                for vi, coef in ((a, -U[0] - U[1]), (b, U[0]), (c, U[1])):
                    if col[vi] >= 0:
                        A[3 * k:3 * k + 3, col[vi]] = coef
                    else:
                        Ar[3 * k:3 * k + 3, ccol[vi]] = coef
                k += 1
        self.A, self.Ar, self.free = A, Ar, free
        self.AtA = A.T @ A + reg * np.eye(len(free))                             # :122-131

    @staticmethod
    def transform(d):
        """exp(log R) * S of one 9-vector (:225-242; utils_rotation.cpp:33-49)."""
        K = np.array([[0, d[6], d[7]], [-d[6], 0, d[8]], [-d[7], -d[8], 0]], np.float64)
        ang = np.sqrt(d[6] ** 2 + d[7] ** 2 + d[8] ** 2)
        if ang < 1e-10:
            R = np.eye(3)
        else:
            Kn = K / ang
            R = np.eye(3) + np.sin(ang) * Kn + (1 - np.cos(ang)) * (Kn @ Kn)
        S = np.array([[d[0] + 1, d[1], d[2]], [d[1], d[3] + 1, d[4]], [d[2], d[4], d[5] + 1]], np.float64)
        return R @ S

    def get_mesh(self, dgrad):
        d = np.asarray(dgrad, np.float64).reshape(-1, 9)
        if self.count is None:
            M = np.concatenate([self.transform(x).T for x in d], 0)              # transposed blocks, :243-251
        else:                                                                    # :252-267
            blocks, fi = [], 0
            for j in range(len(self.F)):
                if self.count[j] > 0:
                    for _ in range(self.count[j]):
                        blocks.append(self.transform(d[self.corr_faces[fi]]).T); fi += 1
                else:
                    blocks.append(np.eye(3)); fi += 1
            M = np.concatenate(blocks, 0)
        C = self.V[self.cn].astype(np.float64) if len(self.cn) else np.zeros((1, 3))
        rhs = self.A.T @ (M - self.Ar @ C)                                       # :282, :286
        X = np.linalg.solve(self.AtA, rhs)
        out = self.V.astype(np.float64).copy()
        out[self.free] = X
        return out.astype(np.float32)

"""CPU restatement (numpy, fp64) of the reference's dgrad -> mesh solve.  TEST INFRASTRUCTURE ONLY.

Follows deformation/cpp/src/deform_triangle_impl.hpp: setStaticTarget (:8-140) and getMeshFromDeformationGradients
(:215-310), with rotation_log_exp::exp (rotation/utils_rotation.cpp:33-49).  Pinned against the reference's own
compiled module (oracle/_ref, built by oracle/build_ref.sh) through tests/golden/mesh.npz.
"""
import numpy as np


class MeshOracle:
    def __init__(self, verts, faces, cnsts, reg=1e-10):
        V = np.asarray(verts, np.float32).reshape(-1, 3)
        F = np.asarray(faces, np.int64).reshape(-1, 3)
        cn = np.asarray(cnsts, np.int64).reshape(-1)
        self.V, self.F, self.cn = V, F, cn
        free = np.setdiff1d(np.arange(len(V)), cn)
        col = -np.ones(len(V), int); col[free] = np.arange(len(free))
        ccol = -np.ones(len(V), int); ccol[cn] = np.arange(len(cn))
        T = len(F)
        A = np.zeros((3 * T, len(free))); Ar = np.zeros((3 * T, max(len(cn), 1)))
        for j, (a, b, c) in enumerate(F):
            Va = np.stack([V[b] - V[a], V[c] - V[a]], 1).astype(np.float64)      # float32 subtraction, :96-97
            Q, R = np.linalg.qr(Va)                                              # :98-100
            U = np.linalg.inv(R) @ Q.T
            for vi, coef in ((a, -U[0] - U[1]), (b, U[0]), (c, U[1])):           # :106-116
                if col[vi] >= 0:
                    A[3 * j:3 * j + 3, col[vi]] = coef
                else:
                    Ar[3 * j:3 * j + 3, ccol[vi]] = coef
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
        M = np.concatenate([self.transform(x).T for x in d], 0)                  # transposed blocks, :243-247
        C = self.V[self.cn].astype(np.float64) if len(self.cn) else np.zeros((1, 3))
        rhs = self.A.T @ (M - self.Ar @ C)                                       # :282, :286
        X = np.linalg.solve(self.AtA, rhs)
        out = self.V.astype(np.float64).copy()
        out[self.free] = X
        return out.astype(np.float32)

"""dgrad -> mesh on the GPU (next row after the hot path): host mirror of the reference's `deformation` module
for the calls speech_anime/viewer/frame.py makes (set_target once, get_mesh per frame) -- here batched over frames."""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check, SdfaError


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class MeshSolver:
    """deformation.set_target(verts, faces, cnsts, corrs, reg=1e-10)  (deformation/cpp/src/pybind.cpp:13-33).

    `corr_count` / `corr_faces`: triangle correspondences as speech_anime/viewer/frame.py:50-80 builds them from a
    .tricorrs file (target triangle -> source triangles of the model's FLAME-topology output); `n_src_tris` is then the
    number of 9-vectors per dgrad row (9976)."""

    def __init__(self, verts, faces, cnsts=(), reg=1e-10, device="cuda:0", corr_count=None, corr_faces=None, n_src_tris=None):
        if not torch.cuda.is_available():
            raise RuntimeError("MeshSolver needs a ROCm GPU: there is no CPU implementation")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        v = np.ascontiguousarray(np.asarray(verts, np.float32).reshape(-1, 3))
        f = np.ascontiguousarray(np.asarray(faces, np.uint32).reshape(-1, 3))
        c = np.ascontiguousarray(np.asarray(cnsts, np.uint32).reshape(-1))
        self.n_verts, self.n_tris, self.n_cnsts = len(v), len(f), len(c)
        has_corr = corr_count is not None and len(corr_count) > 0
        cc = np.ascontiguousarray(np.asarray(corr_count, np.uint32).reshape(-1)) if has_corr else None
        cf = np.ascontiguousarray(np.asarray(corr_faces, np.uint32).reshape(-1)) if has_corr else None
        if has_corr:
            assert len(cc) == len(f), "corr_count needs one entry per template triangle"
            assert n_src_tris is not None, "n_src_tris (9-vectors per dgrad row) is required with correspondences"
        self.n_src_tris = int(n_src_tris) if has_corr else len(f)
        self._m = lib.sdfa_mesh_create_corres(v.ctypes.data_as(C.c_void_p), len(v), f.ctypes.data_as(C.c_void_p), len(f),
                                              c.ctypes.data_as(C.c_void_p) if len(c) else None, len(c),
                                              cc.ctypes.data_as(C.c_void_p) if has_corr else None,
                                              cf.ctypes.data_as(C.c_void_p) if has_corr else None, len(cf) if has_corr else 0,
                                              self.n_src_tris, float(reg), _stream())
        if not self._m:
            raise SdfaError(-1, lib.sdfa_last_error().decode())
        self._ws = None

    def __del__(self):
        m, self._m = getattr(self, "_m", None), None
        if m:
            lib.sdfa_mesh_destroy(m)

    def is_same(self, num_verts, num_faces, num_cnsts):
        return (self.n_verts, self.n_tris, self.n_cnsts) == (num_verts, num_faces, num_cnsts)

    def get_mesh(self, deform_grad):
        """(n, n_tris*9) or (n, n_tris, 9) float32 cuda tensor (or one frame as numpy) -> (n, n_verts, 3) cuda tensor."""
        single = False
        if not torch.is_tensor(deform_grad):
            deform_grad = torch.from_numpy(np.asarray(deform_grad, np.float32))
        if deform_grad.numel() == self.n_src_tris * 9 and deform_grad.dim() <= 2 and deform_grad.shape[0] != 1:
            deform_grad, single = deform_grad.reshape(1, -1), True
        d = deform_grad.to(device=self.device, dtype=torch.float32).reshape(deform_grad.shape[0], -1).contiguous()
        assert d.shape[1] == self.n_src_tris * 9, f"dgrad rows must hold {self.n_src_tris * 9} values"
        n = d.shape[0]
        out = torch.empty((n, self.n_verts, 3), dtype=torch.float32, device=self.device)
        if n:
            need = check(lib.sdfa_mesh_workspace_bytes(self._m, n))
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            check(lib.sdfa_mesh_from_dgrad(self._m, C.c_void_p(d.data_ptr()), n, C.c_void_p(out.data_ptr()),
                                           C.c_void_p(self._ws.data_ptr()), self._ws.numel(), _stream()))
        return out[0] if single else out

    def _workspace(self, n):
        need = check(lib.sdfa_mesh_workspace_bytes(self._m, n))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def get_mesh_seek(self, dgrad_rows, plan, out=None):
        """saber.stream.seek fused into the solve (speech_anime/model/model.py:204-212 does seek -> frame_to_mesh per video
        frame): `dgrad_rows` are the animation-rate rows (n_frames, n_src_tris*9) exactly as the regressor wrote them,
        `plan` a sdfa_amd.seek.SeekPlan over the same clips -> (n_queries, n_verts, 3).  The video-rate dgrad track is
        never materialised."""
        d = dgrad_rows.to(device=self.device, dtype=torch.float32).reshape(dgrad_rows.shape[0], -1).contiguous()
        assert d.shape == (plan.n_frames, self.n_src_tris * 9)
        n = plan.n_queries
        if out is None:
            out = torch.empty((n, self.n_verts, 3), dtype=torch.float32, device=self.device)
        if n:
            ws = self._workspace(n)
            check(lib.sdfa_mesh_from_dgrad_seek(self._m, C.c_void_p(d.data_ptr()), C.c_void_p(plan.src.data_ptr()),
                                                C.c_void_p(plan.w.data_ptr()), n, C.c_void_p(out.data_ptr()),
                                                C.c_void_p(ws.data_ptr()), ws.numel(), _stream()))
        return out

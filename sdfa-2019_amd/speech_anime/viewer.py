"""Template-mesh state and frame_to_mesh -- the part of speech_anime/viewer/frame.py (:17-141) that turns a
dgrad / offsets frame into vertices, backed by the GPU deformation solve (sdfa_amd.mesh).  Rendering is out of scope."""
import numpy as np
import torch

from sdfa_amd.mesh import MeshSolver

_template_verts, _template_faces = None, None
_template_c_indices = []
_solver = None


def read_obj(path):
    """Vertices / triangle faces of a Wavefront OBJ (what saber.mesh.read_mesh returns for the template)."""
    V, F = [], []
    with open(path) as fp:
        for line in fp:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                V.append([float(x) for x in p[1:4]])
            elif p[0] == "f":
                F.append([int(x.split("/")[0]) - 1 for x in p[1:4]])
    return np.asarray(V, np.float32), np.asarray(F, np.uint32)


def write_obj(path, verts, faces):
    with open(path, "w") as fp:
        for v in np.asarray(verts).reshape(-1, 3):
            fp.write("v {:.6f} {:.6f} {:.6f}\n".format(*v))
        for f in np.asarray(faces).reshape(-1, 3):
            fp.write("f {} {} {}\n".format(*(f + 1)))


def set_dgrad_static(verts, faces, c_indices=None, corres=None):
    global _template_verts, _template_faces, _template_c_indices, _solver
    if corres is not None:
        raise NotImplementedError("triangle correspondences (--mesh_tricorres) are not part of this build")
    _template_verts = np.asarray(verts, np.float32).reshape(-1, 3)
    _template_faces = np.asarray(faces, np.uint32).reshape(-1, 3)
    _template_c_indices = [] if c_indices is None else list(c_indices)
    _solver = MeshSolver(_template_verts, _template_faces, _template_c_indices)      # deformation.set_target


def set_template_mesh(template_path, constraints_path=None, corres_path=None):
    verts, faces = read_obj(template_path)
    c_indices = None
    if constraints_path is not None:
        with open(constraints_path) as fp:
            c_indices = [int(x) for x in " ".join(l.strip() for l in fp.readlines()).split()]
    if corres_path is not None:
        raise NotImplementedError("triangle correspondences (--mesh_tricorres) are not part of this build")
    set_dgrad_static(verts, faces, c_indices)


def has_template():
    return _solver is not None


def frames_to_mesh(data_frames, face_data_type):
    """Batched frame_to_mesh: (n, 9976, 9) / (n, 89784) dgrad or (n, 15069) offsets -> (verts (n, V, 3) numpy, faces)."""
    assert _solver is not None, "set_template_mesh first"
    x = data_frames if torch.is_tensor(data_frames) else torch.from_numpy(np.asarray(data_frames, np.float32))
    n = x.shape[0]
    if str(face_data_type).endswith("dgrad_3d"):
        verts = _solver.get_mesh(x.reshape(n, -1)).cpu().numpy()
    elif str(face_data_type).endswith("verts_off_3d"):
        verts = x.reshape(n, -1, 3).cpu().numpy() + _template_verts[None]
    else:
        verts = x.reshape(n, -1, 3).cpu().numpy()
    return verts, _template_faces


def frame_to_mesh(data_frame, face_data_type):
    x = data_frame if torch.is_tensor(data_frame) else torch.from_numpy(np.asarray(data_frame, np.float32))
    verts, faces = frames_to_mesh(x.reshape(1, -1), face_data_type)
    return verts[0], faces

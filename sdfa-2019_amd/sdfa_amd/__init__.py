"""sdfa_amd -- host side of the MI355X hot path: ctypes binding of libsdfa_hip.so (_lib), checkpoint folding (weights),
the per-GPU engine (engine), the dgrad -> mesh solver (mesh), multi-GPU sharding (dist) and synthetic checkpoints (synth).
Submodules are imported explicitly by their users so that `import sdfa_amd.synth` works without a GPU library."""
import os as _os

# HIP maps streams onto a handful of hardware queues (4 by default); two streams on one queue run one after the other.  Measured on
# MI355X (profiles/r03_overlap_env.txt): with the defaults, the stream torch's RCCL process group communicates on shared the default
# stream's queue and the per-chunk all-gather never overlapped the kernels.  More queues / a high-priority communication stream fix
# that; both settings are read when the HIP runtime starts, so they are set here, at import, unless the caller has set them.
import sys as _sys

_late = [k for k in ("GPU_MAX_HW_QUEUES", "TORCH_NCCL_HIGH_PRIORITY") if k not in _os.environ]
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
_os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
_torch = _sys.modules.get("torch")
if _late and _torch is not None and _torch.cuda.is_initialized():
    # too late for this process: the runtime read its environment when it started.  Nothing breaks -- the copy / collective streams
    # are PROBED for real overlap (sdfa_amd/streams.py), not assumed from these settings -- but say so.
    import warnings as _w
    _w.warn(f"sdfa_amd imported after the HIP runtime was initialised: {', '.join(_late)} set now have no effect in this process "
            "(import sdfa_amd -- or export them -- before the first CUDA call to get more hardware queues)", RuntimeWarning, stacklevel=2)


def runtime_env():
    """The queue / priority settings this process runs under (recorded in bench.py's `config`: runs under different settings must
    be distinguishable from the line)."""
    return {k: _os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "TORCH_NCCL_HIGH_PRIORITY", "HSA_ENABLE_IPC_MODE_LEGACY", "HSA_ENABLE_SDMA")}

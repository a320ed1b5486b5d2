"""sdfa_amd -- host side of the MI355X hot path: ctypes binding of libsdfa_hip.so (_lib), checkpoint folding (weights),
the per-GPU engine (engine), the dgrad -> mesh solver (mesh), multi-GPU sharding (dist) and synthetic checkpoints (synth).
Submodules are imported explicitly by their users so that `import sdfa_amd.synth` works without a GPU library."""
import os as _os

# HIP maps streams onto a handful of hardware queues (4 by default); two streams on one queue run one after the other.  Measured on
# MI355X (profiles/r03_overlap_env.txt): with the defaults, the stream torch's RCCL process group communicates on shared the default
# stream's queue and the per-chunk all-gather never overlapped the kernels.  More queues / a high-priority communication stream fix
# that; both settings are read when the HIP runtime starts, so they are set here, at import, unless the caller has set them.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
_os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")

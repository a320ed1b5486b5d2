"""sdfa_amd -- host side of the MI355X hot path: ctypes binding of libsdfa_hip.so (_lib), checkpoint folding (weights),
the per-GPU engine (engine), the dgrad -> mesh solver (mesh), multi-GPU sharding (dist) and synthetic checkpoints (synth).
Submodules are imported explicitly by their users so that `import sdfa_amd.synth` works without a GPU library."""

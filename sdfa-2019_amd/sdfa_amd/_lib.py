"""ctypes binding of libsdfa_hip.so -- the C ABI declared in include/sdfa_hip.h.

There is NO fallback: if the HIP library is missing or a symbol is absent, importing this
module raises.  The product path never routes through a CPU implementation.
"""
import ctypes as C
import os

# PyTorch-ROCm bundles its own libamdhip64.so.7.  Streams and device pointers handed across the C ABI
# come from that runtime, so it must be the one libsdfa_hip.so binds to: load torch first (same SONAME,
# the dynamic loader then reuses the copy already in the process).
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDFA_HIP_LIB", os.path.join(_HERE, "libsdfa_hip.so"))

ABI_VERSION = 5      # include/sdfa_hip.h SDFA_ABI_VERSION this binding was written against
OK, EINVAL, ESHORTCLIP, EHIP, ESTATE, ENOSPACE = 0, -1, -2, -3, -4, -5
HEAD_DGRAD, HEAD_OFFSETS = 0, 1

# name -> (restype, argtypes); kept in one table so tests can check every declared symbol is exported
_p, _i64, _i32, _f = C.c_void_p, C.c_int64, C.c_int32, C.c_float
SYMBOLS = {
    "sdfa_abi_version": (C.c_int, []),
    "sdfa_last_error": (C.c_char_p, []),
    "sdfa_frame_index": (_i64, [_i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _i64]),
    "sdfa_mel_frontend": (C.c_int, [_p, _p, _p, _i32, _p, _p, _i64, C.c_int, _p, _p]),
    "sdfa_frontend_workspace_bytes": (_i64, [_i64]),
    "sdfa_mel_frontend_gather": (C.c_int, [_p, _p, _p, _i32, _p, _p, _i64, C.c_int, _p, _p, _i64, _p]),
    "sdfa_model_create": (_p, [C.c_int]),
    "sdfa_model_destroy": (None, [_p]),
    "sdfa_model_set_tensor": (C.c_int, [_p, C.c_char_p, _p, _i64]),
    "sdfa_model_finalize": (C.c_int, [_p, _p]),
    "sdfa_model_set_precision": (C.c_int, [_p, C.c_int]),
    "sdfa_model_precision": (C.c_int, [_p]),
    "sdfa_model_set_reserved_cus": (C.c_int, [_p, C.c_int]),
    "sdfa_ensemble_mean": (C.c_int, [_p, _p, _i64, _p, _p]),
    "sdfa_model_head": (C.c_int, [_p]),
    "sdfa_model_out_dim": (_i64, [_p]),
    "sdfa_model_coef_dim": (_i64, [_p]),
    "sdfa_workspace_bytes": (_i64, [_p, _i64]),
    "sdfa_encoder_forward": (C.c_int, [_p, _p, _i64, _p, _p, _p, _i64, _p]),
    "sdfa_encoder_forward_shared": (C.c_int, [_p, _p, _i64, _p, _p, C.c_int, _p, _p, _p, _i64, _p]),
    "sdfa_regress_forward": (C.c_int, [_p, _p, _p, _i64, _p, _p, _p, _i64, _p]),
    "sdfa_regress_forward_multi": (C.c_int, [_p, _p, _p, _i64, _p, _p, C.c_int, _p, _i64, _p]),
    "sdfa_model_autotune": (C.c_int, [_p, _i64, _p, _i64, _p]),
    "sdfa_expand_coef": (C.c_int, [_p, _p, _i64, _p, _p, _i64, _p]),
    "sdfa_debug_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "sdfa_debug_frontend_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sdfa_debug_keep_intermediates": (C.c_int, [_p, C.c_int]),
    "sdfa_debug_distinct_columns": (_i64, [_p, _i64, _p, _p]),
    "sdfa_debug_tap": (C.c_int, [_p, C.c_int, _i64, _p, _p, _p]),
    "sdfa_workspace_init": (C.c_int, [_p, _i64, _p]),
    "sdfa_workspace_status_async": (C.c_int, [_p, _p, _p]),
    "sdfa_workspace_status": (_i64, [_p, C.c_int, _p]),
    "sdfa_mesh_create": (_p, [_p, _i64, _p, _i64, _p, _i64, C.c_double, _p]),
    "sdfa_mesh_destroy": (None, [_p]),
    "sdfa_mesh_workspace_bytes": (_i64, [_p, _i64]),
    "sdfa_mesh_from_dgrad": (C.c_int, [_p, _p, _i64, _p, _p, _i64, _p]),
    "sdfa_mesh_n_verts": (_i64, [_p]),
    "sdfa_mesh_n_src_tris": (_i64, [_p]),
    "sdfa_mesh_create_corres": (_p, [_p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _i64, C.c_double, _p]),
    "sdfa_seek_query_count": (_i64, [_i32, C.c_double]),
    "sdfa_seek_plan": (C.c_int, [_p, _p, _p, _i32, C.c_double, _i64, _p, _p, _p]),
    "sdfa_seek_rows": (C.c_int, [_p, _i64, _p, _p, _i64, _p, _p]),
    "sdfa_mesh_from_dgrad_seek": (C.c_int, [_p, _p, _p, _p, _i64, _p, _p, _i64, _p]),
    "sdfa_resample_out_len": (_i64, [_i64, C.c_int, C.c_int]),
    "sdfa_resample_workspace_bytes": (_i64, [_i64, C.c_int, C.c_int]),
    "sdfa_resample_filter": (C.c_int, [_p, _i64]),
    "sdfa_resample": (C.c_int, [_p, _i64, C.c_int, C.c_int, _p, _i64, _p, _i64, _p]),
    "sdfa_profile_enable": (C.c_int, [_p, C.c_int]),
    "sdfa_profile_reset": (C.c_int, [_p]),
    "sdfa_profile_ms": (_f, [_p, C.c_char_p]),
}


class SdfaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsdfa_hip: {msg} (code {code})")
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C sdfa-2019_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    stale = (f"{LIB_PATH} is a stale build (%s): rebuild it with `make -C sdfa-2019_amd/csrc` (or __graft_entry__.build()).  "
             "There is no CPU fallback.")
    try:
        lib.sdfa_abi_version.restype = C.c_int
        have = int(lib.sdfa_abi_version())
    except AttributeError:
        raise ImportError(stale % "it does not export sdfa_abi_version") from None
    if have != ABI_VERSION:
        raise ImportError(stale % f"ABI version {have}, this binding needs {ABI_VERSION}")
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ImportError(stale % f"symbol {name} is not exported") from None
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


option_epoch = 0      # bumped by every set_option: caches of results that a switch may change key on it (speech_anime signal -> z cache)


def set_option(name, value):
    """Library tuning switch (A/B runs); also settable as SDFA_OPTS="name=value,name=value" in the environment."""
    global option_epoch
    check(lib.sdfa_debug_set_option(name.encode(), int(value)))
    option_epoch += 1


def check(rc):
    if rc is not None and rc < 0:
        code = int(rc)
        msg = lib.sdfa_last_error().decode("utf-8", "replace")
        if code == ESHORTCLIP:
            raise AssertionError(msg)      # the reference raises AssertionError here (sliding_window.py:363)
        raise SdfaError(code, msg)
    return rc


for _kv in filter(None, os.environ.get("SDFA_OPTS", "").split(",")):
    set_option(*_kv.split("="))

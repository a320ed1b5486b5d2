"""Sample-rate conversion on the GPU (C ABI sdfa_resample): the arithmetic of librosa.resample(res_type="kaiser_best")
that the reference applies to every input file (speech_anime/model/eval_utils.py:76-86).  Parity unpinned -- see
include/sdfa_hip.h and oracle/resample_oracle.py."""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check


def resample(y, orig_sr, target_sr, device="cuda:0"):
    """1-D float32 signal (numpy or tensor) at `orig_sr` -> cuda float32 tensor at `target_sr`, ceil(n * ratio) samples."""
    if not torch.cuda.is_available():
        raise RuntimeError("sdfa_amd.resample needs a ROCm GPU: there is no CPU implementation")
    dev = torch.device(device)
    x = (y if torch.is_tensor(y) else torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)))
    x = x.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
    if not bool(torch.isfinite(x).all()):
        raise ValueError("Audio buffer is not finite everywhere")            # librosa.util.valid_audio
    n_in = int(x.numel())
    n_out = int(check(lib.sdfa_resample_out_len(n_in, int(orig_sr), int(target_sr))))
    out = torch.empty(n_out, dtype=torch.float32, device=dev)
    ws = torch.empty(int(check(lib.sdfa_resample_workspace_bytes(n_in, int(orig_sr), int(target_sr)))), dtype=torch.uint8, device=dev)
    check(lib.sdfa_resample(C.c_void_p(x.data_ptr()), n_in, int(orig_sr), int(target_sr), C.c_void_p(out.data_ptr()), n_out,
                            C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out

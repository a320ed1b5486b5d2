"""Frame-time resampling on the GPU: saber.stream.seek (saber/data/stream/stream.py:20-46) for the uniform video-rate
queries of speech_anime/model/model.py:204-212 -- host mirror over the C ABI (sdfa_seek_*)."""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr())


def query_count(last_timestamp_ms, fps):
    """len(range(int(tslist[-1] * fps / 1000.0) + 1))  (model.py:205-207)."""
    return int(check(lib.sdfa_seek_query_count(int(last_timestamp_ms), float(fps))))


class SeekPlan:
    """For a batch of clips: which two animation-frame rows each video frame blends, and with which float32 weights.

    tslists: per clip, the ascending integer millisecond timestamps of its animation frames (sdfa_frame_index);
    the rows of the matrix to be resampled are the clips' frames concatenated in this order."""

    def __init__(self, tslists, fps, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("sdfa_amd.seek needs a ROCm GPU: there is no CPU implementation")
        self.device = torch.device(device)
        self.fps = float(fps)
        counts = [len(t) for t in tslists]
        assert all(c > 0 for c in counts), "every clip needs at least one animation frame"
        self.query_counts = [query_count(t[-1], fps) for t in tslists]
        self.frame_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.query_off = np.concatenate([[0], np.cumsum(self.query_counts)]).astype(np.int64)
        self.n_frames, self.n_queries = int(self.frame_off[-1]), int(self.query_off[-1])
        ts = np.concatenate([np.asarray(t, np.int32) for t in tslists])
        self.d_tslist = torch.from_numpy(ts).to(self.device)
        self.d_frame_off = torch.from_numpy(self.frame_off).to(self.device)
        self.d_query_off = torch.from_numpy(self.query_off).to(self.device)
        self.src = torch.empty((self.n_queries, 2), dtype=torch.int64, device=self.device)
        self.w = torch.empty((self.n_queries, 2), dtype=torch.float32, device=self.device)
        check(lib.sdfa_seek_plan(_p(self.d_tslist), _p(self.d_frame_off), _p(self.d_query_off), len(tslists), self.fps,
                                 self.n_queries, _p(self.src), _p(self.w), _stream()))

    def rows(self, sequence):
        """(n_frames, ...) float32 cuda tensor -> (n_queries, ...): every video frame's blended row."""
        x = sequence.to(device=self.device, dtype=torch.float32).contiguous()
        assert x.shape[0] == self.n_frames, f"{x.shape[0]} rows for {self.n_frames} animation frames"
        out = torch.empty((self.n_queries,) + tuple(x.shape[1:]), dtype=torch.float32, device=self.device)
        if self.n_queries:
            check(lib.sdfa_seek_rows(_p(x), int(x[0].numel()), _p(self.src), _p(self.w), self.n_queries, _p(out), _stream()))
        return out

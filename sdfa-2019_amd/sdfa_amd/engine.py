"""Host-side engine over the C ABI: device buffers and streams come from PyTorch-ROCm (plumbing only);
every arithmetic step of the hot path runs in libsdfa_hip.so."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, check
from .weights import fold_state_dict, head_of, check_strict

FPS = 60
TS_DELTA_MS = 100
FEAT_SHAPE = (64, 128, 3)


def frame_geometry(sr):
    """speech_anime/datasets/sliding_window.py:339-343 (window / hop given in seconds)."""
    win, hop = int(0.064 * sr), int(0.008 * sr)
    return win, hop, hop * 63 + win


def frame_index(n_samples, sr, fps=FPS, ts_delta=TS_DELTA_MS):
    """(starts int64[F], tslist int32[F]) -- bit-exact frame enumeration (C ABI sdfa_frame_index)."""
    win, hop, _ = frame_geometry(sr)
    n = check(lib.sdfa_frame_index(int(n_samples), int(sr), int(fps), win, hop, int(ts_delta), None, None, 0))
    starts = np.empty(n, np.int64)
    ts = np.empty(n, np.int32)
    check(lib.sdfa_frame_index(int(n_samples), int(sr), int(fps), win, hop, int(ts_delta),
                               starts.ctypes.data_as(C.c_void_p), ts.ctypes.data_as(C.c_void_p), n))
    return starts, ts


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class FrontendOnly:
    """The spectral-gather front end alone (needs no weights)."""

    def __init__(self, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("sdfa_amd needs a ROCm GPU: the hot path has no CPU implementation")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)

    # ------------------------------------------------------------------ front end
    def mel_frontend(self, clips, sr, gather=True):
        """clips: list of 1-D float32 arrays/tensors in [-1,1].  Returns (audio_feat (F_total,64,128,3) cuda,
        per-clip tslists, per-clip frame counts)."""
        offs, lens, fclip, fstart, tslists, counts = [], [], [], [], [], []
        pos = 0
        for ci, c in enumerate(clips):
            n = int(c.shape[0])
            starts, ts = frame_index(n, sr)
            offs.append(pos); lens.append(n); pos += n
            fclip.append(np.full(len(starts), ci, np.int32)); fstart.append(starts)
            tslists.append([int(t) for t in ts]); counts.append(len(starts))
        dev = self.device
        pcm = torch.cat([torch.as_tensor(c, dtype=torch.float32).reshape(-1) for c in clips]).to(dev, non_blocking=True)
        d_off = torch.tensor(offs, dtype=torch.int64, device=dev)
        d_len = torch.tensor(lens, dtype=torch.int64, device=dev)
        d_fc = torch.from_numpy(np.concatenate(fclip)).to(dev)
        d_fs = torch.from_numpy(np.concatenate(fstart)).to(dev)
        feat = self.mel_frontend_device(pcm, d_off, d_len, d_fc, d_fs, sr, gather=gather)
        self.last_frame_table = (d_fc, d_fs, frame_geometry(sr)[1])      # (clip, start, hop) for encoder(share)
        return feat, tslists, counts

    def mel_frontend_device(self, pcm, clip_off, clip_len, frame_clip, frame_start, sr, out=None, gather=True):
        """gather=True: the "spectral gather" form (each distinct STFT column transformed once, frames gathered from the
        mel table; needs scratch memory); gather=False: one FFT per window column (sdfa_mel_frontend)."""
        F = int(frame_clip.numel())
        if out is None:
            out = torch.empty((F,) + FEAT_SHAPE, dtype=torch.float32, device=self.device)
        if F == 0:
            return out
        if gather:
            need = check(lib.sdfa_frontend_workspace_bytes(F))
            ws = getattr(self, "_fe_ws", None)
            if ws is None or ws.numel() < need:
                self._fe_ws = None
                self._fe_ws = ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            check(lib.sdfa_mel_frontend_gather(_ptr(pcm), _ptr(clip_off), _ptr(clip_len), int(clip_off.numel()), _ptr(frame_clip),
                                               _ptr(frame_start), F, int(sr), _ptr(out), _ptr(ws), ws.numel(), _stream()))
        else:
            check(lib.sdfa_mel_frontend(_ptr(pcm), _ptr(clip_off), _ptr(clip_len), int(clip_off.numel()), _ptr(frame_clip),
                                        _ptr(frame_start), F, int(sr), _ptr(out), _stream()))
        return out


class Engine(FrontendOnly):
    """One model replica on one GPU."""

    PRECISIONS = {"fp32": 0, "bf16_attention": 1, "bf16x3": 2, "bf16": 3}      # include/sdfa_hip.h SDFA_PREC_*

    AUTOTUNE_MIN_FRAMES = 2048      # below this a launch is too short for the choice to matter (or to be measured)

    def __init__(self, state_dict, device="cuda:0", max_frames=8192, debug_keep=False, precision="fp32", strict=True, autotune=False):
        """autotune: at the first encoder call of at least AUTOTUNE_MIN_FRAMES frames, time the bit-identical launch forms of
        the frequency-LSTM recurrence on this device and keep the fastest (sdfa_model_autotune; about half a second, once)."""
        super().__init__(device)
        self._autotune_pending = bool(autotune)
        self.freq_lstm_form = None          # set by autotune(): 9 / 8 / 5 / 3 (csrc/kernels.h FreqLstmArgs::shape)
        folded = fold_state_dict(state_dict)
        self.head = head_of(state_dict)
        if strict:
            check_strict(folded, self.head)
        self._m = lib.sdfa_model_create(_lib.HEAD_DGRAD if self.head == "dgrad" else _lib.HEAD_OFFSETS)
        if not self._m:
            raise _lib.SdfaError(-1, lib.sdfa_last_error().decode())
        for name, arr in folded.items():
            check(lib.sdfa_model_set_tensor(self._m, name.encode(), arr.ctypes.data_as(C.c_void_p), arr.size))
        if debug_keep:
            check(lib.sdfa_debug_keep_intermediates(self._m, 1))
        check(lib.sdfa_model_finalize(self._m, _stream()))
        self.out_dim = int(lib.sdfa_model_out_dim(self._m))
        self.coef_dim = int(lib.sdfa_model_coef_dim(self._m))
        self.max_frames = int(max_frames)
        self._ws = None
        self.set_precision(precision)

    def set_precision(self, precision):
        """Matrix instruction of the dense contractions (BASELINE configs[3]); "fp32" is the reference's arithmetic."""
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self.PRECISIONS)}, got {precision!r}")
        check(lib.sdfa_model_set_precision(self._m, self.PRECISIONS[precision]))
        self.precision = precision

    def __del__(self):
        m, self._m = getattr(self, "_m", None), None
        if m:
            lib.sdfa_model_destroy(m)

    # ------------------------------------------------------------------ workspace
    def workspace(self, n_frames):
        need = check(lib.sdfa_workspace_bytes(self._m, min(int(n_frames), self.max_frames)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def autotune(self, n_frames=None):
        """Measures the launch forms of the frequency-LSTM kernel on `n_frames` (default max_frames) frames and keeps the
        fastest; returns the form chosen.  Outputs do not depend on it."""
        n = min(int(n_frames or self.max_frames), self.max_frames)
        ws = self.workspace(n)
        self.freq_lstm_form = int(check(lib.sdfa_model_autotune(self._m, n, _ptr(ws), ws.numel(), _stream())))
        self._autotune_pending = False
        return self.freq_lstm_form

    # ------------------------------------------------------------------ model
    def encoder(self, audio_feat, want_align=True, frame_clip=None, frame_start=None, hop=None):
        """z (n,512), align (n,64).  With the frame table (`frame_clip` int32, `frame_start` int64, `hop`) the
        per-column stages run once per distinct column (sdfa_encoder_forward_shared)."""
        assert audio_feat.is_cuda and audio_feat.dtype == torch.float32 and tuple(audio_feat.shape[1:]) == FEAT_SHAPE
        audio_feat = audio_feat.contiguous()
        n = audio_feat.shape[0]
        z = torch.empty((n, 512), dtype=torch.float32, device=self.device)
        align = torch.empty((n, 64), dtype=torch.float32, device=self.device) if want_align else None
        if n == 0:
            return z, align
        if self._autotune_pending and n >= self.AUTOTUNE_MIN_FRAMES:
            self.autotune(n)
        ws = self.workspace(n)
        if frame_clip is None:
            check(lib.sdfa_encoder_forward(self._m, _ptr(audio_feat), n, _ptr(z), _ptr(align), _ptr(ws), ws.numel(), _stream()))
        else:
            assert frame_clip.dtype == torch.int32 and frame_start.dtype == torch.int64 and frame_clip.numel() == n
            check(lib.sdfa_encoder_forward_shared(self._m, _ptr(audio_feat), n, _ptr(frame_clip.contiguous()),
                                                  _ptr(frame_start.contiguous()), int(hop), _ptr(z), _ptr(align), _ptr(ws),
                                                  ws.numel(), _stream()))
        return z, align

    @staticmethod
    def check_speaker_ids(speaker_id):
        """Raises what the reference's one_hot scatter_ raises for an id outside [0, 8) (saber/nn/functions.py:375-378).
        The kernels take the ids as validated: call this (one aminmax, a host sync for a device tensor) on every id
        tensor that reaches `regress`, or pass `check_ids=True` (the default)."""
        if torch.is_tensor(speaker_id):
            if speaker_id.numel() == 0:
                return
            lo, hi = (int(v) for v in torch.aminmax(speaker_id))
        else:
            lo = hi = int(speaker_id)
        if lo < 0 or hi >= 8:
            raise RuntimeError(f"index {hi if hi >= 8 else lo} is out of bounds for dimension 1 with size 8")

    def regress(self, z, speaker_id, want_coef=False, want_out=True, out=None, check_ids=True):
        n = z.shape[0]
        z = z.contiguous()
        if check_ids:
            self.check_speaker_ids(speaker_id)
        spk = speaker_id.to(device=self.device, dtype=torch.int64).contiguous()
        assert spk.numel() == n
        coef = torch.empty((n, self.coef_dim), dtype=torch.float32, device=self.device) if want_coef else None
        if want_out and out is None:
            out = torch.empty((n, self.out_dim), dtype=torch.float32, device=self.device)
        if n == 0:
            return coef, out
        ws = self.workspace(n)
        check(lib.sdfa_regress_forward(self._m, _ptr(z), _ptr(spk), n, _ptr(coef), _ptr(out) if want_out else None,
                                       _ptr(ws), ws.numel(), _stream()))
        return coef, out

    def regress_multi(self, z, speaker_id, outs, want_coef=False, check_ids=True):
        """`outs`: 1..8 float32 (n, out_dim) destination tensors (or raw device pointers as ints): every one receives the
        same rows, written by the regressor's epilogue itself (sdfa_regress_forward_multi -- the direct all-gather path)."""
        n = z.shape[0]
        z = z.contiguous()
        if check_ids:
            self.check_speaker_ids(speaker_id)
        spk = speaker_id.to(device=self.device, dtype=torch.int64).contiguous()
        coef = torch.empty((n, self.coef_dim), dtype=torch.float32, device=self.device) if want_coef else None
        if n == 0:
            return coef
        ptrs = []
        for o in outs:
            if torch.is_tensor(o):
                assert o.is_cuda and o.dtype == torch.float32 and o.is_contiguous() and tuple(o.shape) == (n, self.out_dim)
                ptrs.append(o.data_ptr())
            else:
                ptrs.append(int(o))
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        ws = self.workspace(n)
        check(lib.sdfa_regress_forward_multi(self._m, _ptr(z), _ptr(spk), n, _ptr(coef), arr, len(ptrs), _ptr(ws), ws.numel(), _stream()))
        return coef

    def expand_coef(self, coef, out=None):
        """PCA coefficients (n, coef_dim) -> output rows (n, out_dim): the regressor's last stage alone (sdfa_expand_coef),
        bit-identical to what `regress` writes for the same coefficients."""
        n = coef.shape[0]
        assert coef.is_cuda and coef.dtype == torch.float32 and coef.is_contiguous() and coef.shape[1] == self.coef_dim
        if out is None:
            out = torch.empty((n, self.out_dim), dtype=torch.float32, device=self.device)
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (n, self.out_dim)
        if n:
            ws = self.workspace(min(n, self.max_frames))
            check(lib.sdfa_expand_coef(self._m, _ptr(coef), n, _ptr(out), _ptr(ws), ws.numel(), _stream()))
        return out

    def forward(self, audio_feat, speaker_id, want_coef=False):
        z, align = self.encoder(audio_feat)
        coef, out = self.regress(z, speaker_id, want_coef=want_coef)
        return out, z, align, coef

    def distinct_columns(self, n_frames):
        """Distinct columns evaluated by the last shared encoder call on a chunk of n_frames (reporting only)."""
        return int(check(lib.sdfa_debug_distinct_columns(self._m, int(n_frames), _ptr(self._ws), _stream())))

    def tap(self, what, n_frames):
        shapes = {0: (32, 64, 64), 1: (64, 32, 64), 2: (256, 64), 3: (64, 512)}
        dst = torch.empty((n_frames,) + shapes[what], dtype=torch.float32, device=self.device)
        check(lib.sdfa_debug_tap(self._m, what, n_frames, _ptr(dst), _ptr(self._ws), _stream()))
        return dst

    # ------------------------------------------------------------------ profiling
    def profile(self, on=True):
        check(lib.sdfa_profile_enable(self._m, 1 if on else 0))

    def profile_ms(self, stage):
        v = lib.sdfa_profile_ms(self._m, stage.encode())
        if v < 0:
            check(int(v))
        return float(v)

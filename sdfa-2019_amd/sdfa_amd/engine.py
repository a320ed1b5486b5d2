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

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("sdfa_amd needs a ROCm GPU: the hot path has no CPU implementation")
        # None = the process's current device (one process per GPU sets it once; a hard-wired cuda:0 would pull every rank onto GPU 0)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(self.device)

    def _staging(self, nbytes):
        """Pinned host staging buffer of at least `nbytes` bytes; waits for the upload that last read it."""
        ev = getattr(self, "_stage_ev", None)
        if ev is not None:
            ev.synchronize()
        st = getattr(self, "_stage", None)
        if st is None or st.numel() < nbytes:
            self._stage = st = torch.empty(max(int(nbytes * 1.5), 1 << 20), dtype=torch.uint8, pin_memory=True)
        return st

    # ------------------------------------------------------------------ front end
    def mel_frontend(self, clips, sr, gather=True, tables=None):
        """clips: list of 1-D float32 arrays/tensors in [-1,1].  Returns (audio_feat (F_total,64,128,3) cuda,
        per-clip tslists, per-clip frame counts).  `tables`: per-clip (starts, tslist) from `frame_index`, when the caller
        already has them (one enumeration per clip)."""
        offs, lens, fclip, fstart, tslists, counts = [], [], [], [], [], []
        pos = 0
        for ci, c in enumerate(clips):
            n = int(c.shape[0])
            starts, ts = tables[ci] if tables is not None else frame_index(n, sr)
            offs.append(pos); lens.append(n); pos += n
            fclip.append(np.full(len(starts), ci, np.int32)); fstart.append(starts)
            tslists.append(ts.tolist()); counts.append(len(starts))
        dev = self.device
        # Everything the kernels index by goes up in ONE host -> device copy from a PINNED staging buffer that lives with the engine:
        # [clip offsets | clip lengths | frame starts] int64, frame clips int32, PCM float32.  (A copy from pageable memory makes the
        # runtime pin the pages on the fly; measured on MI355X that stalls a call by 5 - 100 ms once the source passes about 1 MB --
        # two 10 s clips.)
        nf, nc = int(sum(counts)), len(clips)
        n_meta = 2 * nc + nf
        fc_off = n_meta * 8
        pcm_off = (fc_off + nf * 4 + 15) // 16 * 16
        total = pcm_off + pos * 4
        stage = self._staging(total)
        meta = stage[:fc_off].view(torch.int64).numpy()
        meta[:nc] = offs; meta[nc:2 * nc] = lens; meta[2 * nc:] = np.concatenate(fstart)
        stage[fc_off:fc_off + nf * 4].view(torch.int32).numpy()[:] = np.concatenate(fclip)
        hp = stage[pcm_off:total].view(torch.float32).numpy()
        for o, c in zip(offs, clips):
            hp[o:o + int(c.shape[0])] = c.detach().cpu().numpy().reshape(-1) if torch.is_tensor(c) else np.asarray(c, np.float32).reshape(-1)
        d_all = torch.empty(total, dtype=torch.uint8, device=dev)
        d_all.copy_(stage[:total], non_blocking=True)
        self._stage_ev = torch.cuda.Event()
        self._stage_ev.record()
        d_meta = d_all[:fc_off].view(torch.int64)
        d_fc = d_all[fc_off:fc_off + nf * 4].view(torch.int32)
        pcm = d_all[pcm_off:total].view(torch.float32)
        d_off, d_len, d_fs = d_meta[:nc], d_meta[nc:2 * nc], d_meta[2 * nc:]
        feat = self.mel_frontend_device(pcm, d_off, d_len, d_fc, d_fs, sr, gather=gather)
        self.last_frame_table = (d_fc, d_fs, frame_geometry(sr)[1])      # (clip, start, hop) for encoder(share)
        return feat, tslists, counts

    def mel_frontend_device(self, pcm, clip_off, clip_len, frame_clip, frame_start, sr, out=None, gather=True):
        """gather=True: the "spectral gather" form (each distinct STFT column transformed once, frames assembled from mel rows
        in an LDS ring -- the spectral stream of round 5; needs scratch memory); gather=False: one FFT per window column
        (sdfa_mel_frontend)."""
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


    def frontend_status(self):
        """Expired hand-off waits of the spectral-stream front end's last call (0 always; tests only; synchronises)."""
        ws = getattr(self, "_fe_ws", None)
        return 0 if ws is None else int(check(lib.sdfa_debug_frontend_status(_ptr(ws), _stream())))


class Engine(FrontendOnly):
    """One model replica on one GPU."""

    PRECISIONS = {"fp32": 0, "bf16_attention": 1, "bf16x3": 2, "bf16": 3, "bf16x3_attention": 4, "bf16x6": 5}      # include/sdfa_hip.h SDFA_PREC_*

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
        self._id_checks = []            # (pinned [min, max], event) of device-resident speaker-id tensors not yet looked at
        self._status_reads = []         # (pinned status-block copy, event) of forward_host calls not yet looked at
        self._status_free = []          # pinned slots to reuse
        self._repairs_carried = 0       # time-LSTM repairs counted in workspaces this engine has since replaced
        self.repairs_seen = 0           # ... and the last count a status read showed (host_wait / check_pending keep it current)
        self._host = None               # HostPipeline, created by the first forward_host call
        self.set_precision(precision)

    def set_reserved_cus(self, k):
        """The persistent kernels launch (CUs - k) workgroups, leaving k CUs to kernels of other streams (RCCL's all-gather
        of the previous chunk); 0 = all CUs.  Outputs do not depend on it."""
        check(lib.sdfa_model_set_reserved_cus(self._m, int(k)))
        self.reserved_cus = int(k)

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
            if self._ws is not None:
                # a status copy of an earlier call may still be queued on the copy stream: let it read the old block before the
                # allocator can hand the block out again, then carry the old block's count over (two syncs; regrowth is rare)
                self.check_pending(block=True)
                carried = self.time_lstm_repairs()
                self._repairs_carried = carried
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            check(lib.sdfa_workspace_init(_ptr(self._ws), self._ws.numel(), _stream()))    # zero the status block, once
        return self._ws

    def time_lstm_repairs(self):
        """Waits of the cooperating-workgroup time-LSTM kernels that expired since this engine was made (include/sdfa_hip.h, "Status
        block").  Every one was repaired on the device before anything read the layer -- rows are right either way; the count says
        that the device was so oversubscribed that a single-clip call lost about 20 ms (the bound of a wait) plus the 2 ms repair.  Synchronises the stream."""
        if self._ws is None:
            return self._repairs_carried
        return self._repairs_carried + int(check(lib.sdfa_workspace_status(_ptr(self._ws), 0, _stream())))

    def _status_async(self, stream_ptr=None):
        """Enqueues a copy of the workspace's status block to a pinned slot (no sync); `check_pending` looks at it later."""
        if self._ws is None:
            return
        slot = self._status_free.pop() if self._status_free else torch.zeros(4, dtype=torch.int32, pin_memory=True)
        check(lib.sdfa_workspace_status_async(_ptr(self._ws), _ptr(slot), stream_ptr if stream_ptr is not None else _stream()))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._status_reads.append((slot, ev))

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
        One aminmax -- and a host sync when the tensor lives on the device."""
        if torch.is_tensor(speaker_id):
            if speaker_id.numel() == 0:
                return
            lo, hi = (int(v) for v in torch.aminmax(speaker_id))
        else:
            lo = hi = int(speaker_id)
        if lo < 0 or hi >= 8:
            raise RuntimeError(f"index {hi if hi >= 8 else lo} is out of bounds for dimension 1 with size 8")

    def _validate_ids(self, speaker_id, check_ids):
        """check_ids None (default): ids that arrive on the host (ints, CPU tensors) are validated at once; ids that are already
        on the device are validated WITHOUT draining the stream -- their min / max go to a pinned slot asynchronously and an
        out-of-range id raises at the next call that finds the result ready, or in `check_pending()` (the kernels clamp, so
        nothing reads out of bounds meanwhile; the reference on a GPU reports a bad scatter_ index at its next
        synchronisation in the same way).  True: validate now (a host sync for a device tensor).  False: the caller did."""
        self.check_pending(block=False)
        if check_ids is False:
            return
        if check_ids or not (torch.is_tensor(speaker_id) and speaker_id.is_cuda):
            self.check_speaker_ids(speaker_id)
            return
        if speaker_id.numel() == 0:
            return
        lo, hi = torch.aminmax(speaker_id)
        slot = torch.empty(2, dtype=torch.int64, pin_memory=True)
        slot.copy_(torch.stack((lo, hi)), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._id_checks.append((slot, ev))

    def check_pending(self, block=True):
        """Raises for any device-resident speaker-id tensor of an earlier call that held an id outside [0, 8)."""
        still = []
        for slot, ev in self._status_reads:
            if block:
                ev.synchronize()
            elif not ev.query():
                still.append((slot, ev))
                continue
            n = self._repairs_carried + int(slot[0])
            self._status_free.append(slot)
            if n > self.repairs_seen:
                import warnings
                warnings.warn(f"sdfa_amd: {n - self.repairs_seen} time-LSTM launch(es) of a single-clip call waited out their partner workgroups "
                              "(device oversubscribed) and were recomputed on the device; results are unaffected", RuntimeWarning, stacklevel=2)
                self.repairs_seen = n
        self._status_reads = still
        keep = []
        for slot, ev in self._id_checks:
            if block:
                ev.synchronize()
            elif not ev.query():
                keep.append((slot, ev))
                continue
            lo, hi = int(slot[0]), int(slot[1])
            if lo < 0 or hi >= 8:
                self._id_checks = []
                raise RuntimeError(f"index {hi if hi >= 8 else lo} is out of bounds for dimension 1 with size 8")
        self._id_checks = keep

    def regress(self, z, speaker_id, want_coef=False, want_out=True, out=None, check_ids=None):
        n = z.shape[0]
        z = z.contiguous()
        self._validate_ids(speaker_id, check_ids)
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

    def regress_multi(self, z, speaker_id, outs, want_coef=False, check_ids=None):
        """`outs`: 1..8 float32 (n, out_dim) destination tensors (or raw device pointers as ints): every one receives the
        same rows, written by the regressor's epilogue itself (sdfa_regress_forward_multi -- the direct all-gather path)."""
        n = z.shape[0]
        z = z.contiguous()
        self._validate_ids(speaker_id, check_ids)
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

    def ensemble_mean(self, a, b, out=None):
        """(a + b) / 2 element-wise with numpy's float32 roundings (model.py:369-403 test-time ensembling); in place by default."""
        assert a.is_cuda and b.is_cuda and a.dtype == b.dtype == torch.float32 and a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
        out = a if out is None else out
        assert out.is_cuda and out.is_contiguous() and out.shape == a.shape and out.dtype == torch.float32
        check(lib.sdfa_ensemble_mean(_ptr(a), _ptr(b), a.numel(), _ptr(out), _stream()))
        return out

    def forward_host(self, feat, speaker_id, out=None, table=None, piece=None, wait=True, want_z=False, ops_key=None, ensemble=False, z=None):
        """The whole model for `n` frames with the output rows delivered to PINNED HOST memory: rows (n, out_dim) as a CPU
        tensor (`.numpy()` is a view).  Frames are processed in pieces of `piece` frames (default: `piece_schedule` -- one piece up to
        max_frames, otherwise pieces of 3072 frames); piece i's rows
        are copied device -> host on a copy stream while piece i+1 computes (two device staging buffers), so for more than one
        piece the PCIe transfer hides behind the kernels.  This is what SaberSpeechDrivenAnimation._feature_to_anime
        (speech_anime/model/model.py:428-489) does with `.cpu().numpy()` per batch of 100 frames.

          feat        audio_feat (n,64,128,3) cuda -- or, with `ensemble`, (2n,64,128,3): the frames of the two passes of
                      test-time ensembling (model.py:369-403) one after the other; both passes run in ONE launch group and
                      their rows are averaged on the device before the copy
          speaker_id  (n,) int64 tensor (any device) or one int
          table       (frame_clip int32, frame_start int64, hop) of `feat`'s frames: run the per-column stages once per
                      distinct column (bitwise identical, sdfa_encoder_forward_shared)
          out         pinned float32 CPU tensor (n, out_dim) to fill (allocated from PyTorch's pinned-memory cache otherwise)
          wait        False: return as soon as everything is enqueued; call `host_wait()` before reading `out`
          z           encoder output of exactly these frames from an earlier call (`last_z()`): the encoder is skipped and only the
                      regressor runs -- the same signal with another speaker (the reference keeps the features of the last signal
                      for this, model.py:364-367,409-416; here everything up to z is speaker-independent).  `feat` may be None
          ops_key     key of this engine in sdfa_amd.ops' registry: the kernels are then called through the dispatcher-visible
                      PyTorch-ROCm custom operators torch.ops.sdfa.{encoder, encoder_shared, regress_into} (same C ABI calls)"""
        src = feat if z is None else z
        n = int(src.shape[0]) // (2 if ensemble else 1)
        assert src.shape[0] == (2 * n if ensemble else n)
        self._validate_ids(speaker_id, None)
        if not torch.is_tensor(speaker_id):
            speaker_id = torch.full((n,), int(speaker_id), dtype=torch.int64, device=self.device)
        else:
            speaker_id = speaker_id.to(device=self.device, dtype=torch.int64)
            assert speaker_id.numel() == n
        if out is None:
            out = torch.empty((n, self.out_dim), dtype=torch.float32, pin_memory=True)
        assert (not out.is_cuda) and out.is_pinned() and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (n, self.out_dim)
        if self._host is None:
            self._host = HostPipeline(self)
        if piece is None:
            # a piece holds both passes of its frames; short rows (the offsets head: 60 KB, 1 us of copy per frame) hide behind any
            # piece's kernels, so they keep the kernel-efficient max_frames-sized pieces
            big = max(1, self.max_frames // (2 if ensemble else 1))
            sizes = piece_schedule(n, big, many=3072 if self.out_dim * 4 > 150_000 else big)
        elif isinstance(piece, (list, tuple)):      # an explicit schedule (frames per piece; tools/timeline_batch.py)
            sizes = [int(x) for x in piece]
            assert sum(sizes) == n and all(0 < x <= self.max_frames for x in sizes)
        else:
            p = max(1, int(piece) // (2 if ensemble else 1))
            sizes = [min(p, n - f0) for f0 in range(0, n, p)]
        zs = self._host.run(feat, speaker_id, out, table, sizes, want_z, ops_key, ensemble, z)
        if wait:
            self.host_wait()
        return (out, zs) if want_z else out

    def to_host_async(self, t):
        """Copy of a device tensor in pinned host memory, enqueued on the copy stream behind the work already queued on the
        current stream (so it overlaps what is launched next); valid after `host_wait()`."""
        if self._host is None:
            self._host = HostPipeline(self)
        return self._host.stage_out(t)

    def host_wait(self):
        """Blocks until every device -> host copy enqueued by forward_host has landed."""
        if self._host is not None:
            self._host.wait()
        self.check_pending(block=False)      # speaker ids of device tensors; status blocks (time-LSTM repairs) of the calls that have landed

    def last_device_rows(self, n):
        """The device copy of the rows the LAST forward_host call produced, when all `n` of them went through one piece (a
        view into a staging buffer: valid until the next forward_host call); None otherwise."""
        return None if self._host is None else self._host.last_rows(n)

    def last_z(self):
        """Encoder output (n, 512) -- (2n, 512), pass 1 then pass 2, for an ensembling call -- of the LAST forward_host call when all
        its frames went through one piece; None otherwise.  Feed it back as `z=` to re-run only the regressor."""
        return None if self._host is None else self._host.last_z

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


def piece_schedule(n, big, many=3072):
    """Piece sizes of a pinned-output call of `n` frames (Engine.forward_host).  One piece while it fits `big` (= max_frames: the rows
    then also stay on the device, Engine.last_device_rows).  Otherwise the call is a two-stage pipeline -- kernels, then the device ->
    host copy of the piece's rows on ONE copy engine -- whose length is  max over k of (kernels of pieces 0..k + copies of pieces k..last).
    With 359 KB rows at 57 GB/s a frame's copy (6.3 us) is a little shorter than its kernels (7.3 us in pieces of 8192 frames, 7.7 in
    4096, 8.6 in 2048), so the call is  all kernels + the LAST piece's copy,  unless the first piece is so large that
    first kernels + all copies  is longer.  Measured on 32 x 10 s (tools/timeline_batch.py): uniform 8192 -> 199 ms (62 ms before
    the first copy can start), 4096 -> 184 ms, 3072 -> 178 ms, 2048 -> 189 ms (kernel efficiency lost), and a ramp (small first piece, large middle
    ones, halving pieces at the end) 196 ms: the copies of the large pieces queue in front of the small pieces' copies, and with two
    staging buffers the small pieces' regressor waits for them.  So: uniform pieces of `many` frames.  Frames are independent: the
    rows do not depend on the schedule (bitwise: tests/test_surface_fast.py)."""
    n, big = int(n), int(big)
    if n <= big:
        return [n] if n else []
    p = max(1, min(big, int(many)))
    return [min(p, n - f0) for f0 in range(0, n, p)]


class HostPipeline:
    """Pinned-output staging of Engine.forward_host: two device row buffers, one copy stream.

    compute stream:  [piece 0: encoder | regress -> buf 0] [piece 1: encoder | regress -> buf 1] [piece 2: encoder | wait copy 0 | regress -> buf 0] ...
    copy stream:                                          [buf 0 -> host rows 0..p)             ] [buf 1 -> host ...]
    The device -> host copies are plain hipMemcpyAsync to pinned memory (SDMA engines): they take no CU from the kernels."""

    def __init__(self, engine):
        self.eng = engine
        # a stream whose copies REALLY run under this stream's kernels: which new stream shares the compute stream's hardware queue
        # is a matter of creation order, priorities and GPU_MAX_HW_QUEUES, so candidates are probed (sdfa_amd/streams.py)
        from .streams import pick_copy_stream, engine_busy
        self.copy_stream, self.copy_overlaps, self.copy_probe = pick_copy_stream(engine.device, engine_busy(engine))
        self.bufs = [None, None]
        self.tmp = None                 # second pass of test-time ensembling
        self.done = [None, None]        # copy-done event of the last copy out of each buffer
        self.extra = []                 # copy-done events of stage_out copies
        self._last = None
        self.last_z = None              # encoder output of the last call when it was one piece
        self._next = 0                  # running piece counter: staging buffer = parity

    def _buf(self, slot, rows):
        b = self.bufs[slot]
        if b is None or b.shape[0] < rows:
            if b is not None:
                # a device -> host copy on the copy stream may still be reading the old block: the caching allocator must not hand
                # it to the next kernels' temporaries before that copy has run
                b.record_stream(self.copy_stream)
            self.bufs[slot] = b = None
            self.bufs[slot] = b = torch.empty((rows, self.eng.out_dim), dtype=torch.float32, device=self.eng.device)
        return b

    def run(self, feat, spk, out, table, sizes, want_z, ops_key=None, ensemble=False, z_in=None):
        eng = self.eng
        n = int(out.shape[0])
        assert sum(sizes) == n
        piece = max(sizes) if sizes else 0
        cur = torch.cuda.current_stream(eng.device)
        zs = []
        self._last = None
        self.last_z = None
        f1 = 0
        for m in sizes:
            f0, f1 = f1, f1 + m
            slot = self._next & 1                       # alternates across calls too: a call's last copy overlaps the next call's first piece
            self._next += 1
            rows = self._buf(slot, min(piece, n))[:m]
            if ensemble:                                # pass 1 = frames [f0, f1), pass 2 = frames [n + f0, n + f1): ONE launch group for both
                x = None if z_in is not None else torch.cat((feat[f0:f1], feat[n + f0:n + f1]))
                t = None if table is None else (torch.cat((table[0][f0:f1], table[0][n + f0:n + f1])), torch.cat((table[1][f0:f1], table[1][n + f0:n + f1])), table[2])
                ids = torch.cat((spk[f0:f1], spk[f0:f1]))
                if self.tmp is None or self.tmp.shape[0] < 2 * m:
                    self.tmp = None
                    self.tmp = torch.empty((2 * min(piece, n), eng.out_dim), dtype=torch.float32, device=eng.device)
                dst = self.tmp[:2 * m]
            else:
                x, ids, dst = (None if z_in is not None else feat[f0:f1]), spk[f0:f1], rows
                t = None if table is None else (table[0][f0:f1], table[1][f0:f1], table[2])
            if z_in is not None:                        # same signal, another speaker: z is speaker-independent, only the regressor runs
                z = z_in[f0:f1] if not ensemble else (z_in if (f0 == 0 and f1 == n) else torch.cat((z_in[f0:f1], z_in[n + f0:n + f1])))
            elif ops_key is not None:                   # through the dispatcher (torch.ops.sdfa.*): same Engine methods underneath
                if t is None:
                    z, _ = torch.ops.sdfa.encoder(x, ops_key)
                else:
                    z, _ = torch.ops.sdfa.encoder_shared(x, t[0], t[1], int(t[2]), ops_key)
            elif t is None:
                z, _ = eng.encoder(x, want_align=False)
            else:
                z, _ = eng.encoder(x, want_align=False, frame_clip=t[0], frame_start=t[1], hop=t[2])
            # the first kernel that WRITES the staging rows: only now wait for the copy that last read them (it has had the
            # whole encoder of this piece to finish)
            if self.done[slot] is not None and not ensemble:
                cur.wait_event(self.done[slot])
            if ops_key is not None:
                torch.ops.sdfa.regress_into(z, ids, dst, ops_key)
            else:
                eng.regress(z, ids, out=dst, check_ids=False)
            if ensemble:
                if self.done[slot] is not None:
                    cur.wait_event(self.done[slot])
                eng.ensemble_mean(dst[:m], dst[m:], out=rows)
            if want_z:
                zs.append(z[:m])
            ready = torch.cuda.Event()
            ready.record(cur)
            self.copy_stream.wait_event(ready)
            with torch.cuda.stream(self.copy_stream):
                out[f0:f1].copy_(rows, non_blocking=True)
                done = torch.cuda.Event()
                done.record(self.copy_stream)
            self.done[slot] = done
            if f0 == 0 and f1 == n:
                self._last = (slot, n)
                self.last_z = z
        with torch.cuda.stream(self.copy_stream):       # behind the call's last copy: the workspace's status block (time-LSTM repairs)
            eng._status_async()
        return torch.cat(zs) if want_z and len(zs) != 1 else (zs[0] if want_z else None)

    def stage_out(self, t):
        cur = torch.cuda.current_stream(self.eng.device)
        t = t.contiguous()                                  # layout change (if any) on the compute stream
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        ready = torch.cuda.Event()
        ready.record(cur)
        self.copy_stream.wait_event(ready)
        with torch.cuda.stream(self.copy_stream):
            host.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        t.record_stream(self.copy_stream)                   # the caching allocator must not hand `t` out again before the copy ran
        self.extra.append(ev)
        return host

    def wait(self):
        for ev in self.done + self.extra:
            if ev is not None:
                ev.synchronize()
        self.extra = []

    def last_rows(self, n):
        if self._last is None or self._last[1] != n:
            return None
        return self.bufs[self._last[0]][:n]

"""SpeechDrivenAnimation / SaberSpeechDrivenAnimation -- inference surface of speech_anime/model/model.py.

Same method names, arguments and result layouts as the reference; the arithmetic is libsdfa_hip.so.
Not mirrored (out of scope): training (train_step/get_loss), TensorBoard hooks, video rendering and
the dgrad -> mesh solve (evaluate() writes the dgrad track instead of .obj files / a video).
"""
import os
from copy import deepcopy

import numpy as np
import torch

from sdfa_amd.engine import Engine
from sdfa_amd import weights as _weights
from sdfa_amd import ops as _ops
from ..datasets import DatasetSlidingWindow
from .. import audio as _audio
from .. import stream as _stream


class SpeechDrivenAnimation:
    """audio_feat -> anime_feat model (model.py:18-45)."""

    def __init__(self, hparams, load_pca=False):
        self.hp = hparams
        self._face_type = hparams.model.face_data_type
        self._engine = None

    def load_state_dict(self, state_dict, strict=True):
        head = _weights.head_of(state_dict)
        want = "dgrad" if self._face_type == "dgrad_3d" else "offsets"
        if head != want:
            raise RuntimeError(f"checkpoint holds a '{head}' output module but hparams.model.face_data_type = {self._face_type}")
        self._engine = Engine(state_dict, device=self.hp.get("device", "cuda:0") or "cuda:0",
                              precision=self.hp.get("precision", "fp32") or "fp32", strict=strict)
        # weakly registered: dropping this object frees the engine (weights + workspace); anonymous keys come from a counter
        self._key = _ops.register_model(self.hp.get("model_key") or None, self._engine)
        return self

    def eval(self):
        return self

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    @torch.no_grad()
    def forward(self, audio_feat, speaker_id, align_dict=None, latent_dict=None):
        """(N,64,128,3) f32, (N,) int64 -> ((scale (N,1,9976,6), rotat (N,1,9976,3)), z_audio (N,1,512)); offsets head:
        (pred (N,1,15069), z_audio).  A dict `align_dict` receives {"audio_encoder10": (N,1,64)} (layers/__init__.py:98-99)."""
        if self._engine is None:
            raise RuntimeError("no weights loaded: call load_state_dict first")
        eng = self._engine
        x = audio_feat.to(device=eng.device, dtype=torch.float32)
        n = x.shape[0]
        # the arithmetic is two dispatcher-visible PyTorch-ROCm custom ops over the C ABI (sdfa_amd/ops.py)
        z, align = torch.ops.sdfa.encoder(x.contiguous(), self._key)
        if isinstance(align_dict, dict):
            align_dict["audio_encoder10"] = align.view(n, 1, 64)
        out = torch.ops.sdfa.regress(z, speaker_id.to(eng.device), self._key)
        z_audio = z.view(n, 1, 512)
        if self._face_type == "dgrad_3d":
            tri = out.view(n, 1, -1, 9)
            return (tri[..., :6], tri[..., 6:]), z_audio      # views into the interleaved [6 | 3] rows
        return out.view(n, 1, -1), z_audio


class SaberSpeechDrivenAnimation:
    """Handles evaluation (model.py:48-489, inference half)."""

    def __init__(self, hparams, trainset=None, validset=None, load_pca=True):
        self.hp = hparams
        self.trainset, self.validset = trainset, validset
        self._model = SpeechDrivenAnimation(hparams, load_pca)
        self._face_type = hparams.model.face_data_type
        self._pred_type = hparams.model.prediction_type
        self._speakers_dict = deepcopy(dict(hparams.dataset_anime.speakers))
        self._emotions_dict = deepcopy(dict(hparams.dataset_anime.emotions))
        self.current_epoch = 0
        self.on_gpu = True

    # ---- weights ------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        self._model.load_state_dict(state_dict, strict)
        self._signal_cache = None
        DatasetSlidingWindow.use_engine(self._model._engine)
        return self

    def eval(self):
        return self

    def clear_signal_cache(self):
        """Drop the one-entry signal -> z cache of generate_animation (frees z and the kept features on the device)."""
        self._signal_cache = None

    def __call__(self, batch):
        return self.forward(batch)

    # ---- model.py:79-107 ----------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, batch):
        align_dict, latent_dict = dict(), dict()
        preds, condition = self._model(batch["audio_feat"], batch["speaker_id"], align_dict=align_dict, latent_dict=latent_dict)
        pred_dict = dict()
        if self._face_type == "dgrad_3d":
            assert len(preds) == 2
            pred_dict["dgrad_3d_scale"], pred_dict["dgrad_3d_rotat"] = preds
        else:
            # the reference asserts len(preds) == 1 on a bare tensor here (only true for batch 1)
            pred_dict[self._face_type] = preds
        return dict(prediction=pred_dict, condition=condition, align_dict=align_dict, latent_dict=latent_dict)

    # ---- model.py:225-259 ---------------------------------------------------------------------
    def data_to_anime_feat(self, tensor_dict, is_prediction):
        if self._face_type == "dgrad_3d":
            scale, rotat = tensor_dict["dgrad_3d_scale"], tensor_dict["dgrad_3d_rotat"]
            data = torch.cat((scale.reshape(*scale.shape[:-1], -1, 6), rotat.reshape(*rotat.shape[:-1], -1, 3)), dim=-1)
            return data.reshape(*data.shape[:-2], -1)
        return tensor_dict[self._face_type]

    # ---- model.py:333-420 ---------------------------------------------------------------------
    @staticmethod
    def _check_signal(signal):
        """The input checks of generate_animation (model.py:339-349)."""
        if torch.is_tensor(signal):
            if signal.dim() > 1:
                assert signal.dim() == 2
                assert signal.size(0) == 1
                signal = signal[0]
            signal = signal.detach().cpu().numpy()
        assert isinstance(signal, np.ndarray)
        assert np.prod(signal.shape) == np.max(signal.shape)
        assert signal.min() >= -1
        assert signal.max() <= 1
        return np.asarray(signal.flatten(), np.float32)

    @torch.no_grad()
    def generate_animation(self, signal, speaker, emotion, frame_id, ensembling_ms=None, dataset_class=None, **kwargs):
        """(tslist, animes (F, 9976, 9) | (F, 15069) float32 numpy, others) for ONE clip -- the reference's signature and result.

        The default dataset class takes the device pipeline of `generate_animation_batch` (one front-end call, column-sharing
        encoder, ensembling mean on the device, one pinned device -> host copy); a caller-supplied `dataset_class` keeps the
        reference's two-step route through its `fetch_audio_features` and `_feature_to_anime`.  `animes` is a view of pinned host
        memory owned by the returned array.  kwargs: want_inputs (default True, as the reference returns others["inputs"])."""
        signal = self._check_signal(signal)
        if dataset_class is None:
            dataset_class = DatasetSlidingWindow
        if isinstance(speaker, str):
            speaker = self._speakers_dict[speaker]
        if isinstance(emotion, str):
            emotion = self._emotions_dict[emotion]
        if ensembling_ms is None:
            ensembling_ms = self.hp.ensembling_ms
        if dataset_class is DatasetSlidingWindow:
            return self._animate([signal], [speaker], ensembling_ms, kwargs.get("want_inputs", True))[0]

        passes = [signal]
        if ensembling_ms is not None and ensembling_ms > 0:            # model.py:373-384: second pass on a delayed copy
            pad = ensembling_ms * self.hp.audio.sample_rate // 1000
            passes.append(np.pad(signal[:-pad], [[pad, 0]], "constant"))
        feats = [dataset_class.fetch_audio_features(p, self.hp) for p in passes]
        anime_sum, others = self._feature_to_anime(feats[0]["audio_feat"], feats[0]["energy"], speaker, emotion, frame_id,
                                                   want_inputs=kwargs.get("want_inputs", True))
        for f in feats[1:]:
            anime_sum += self._feature_to_anime(f["audio_feat"], f["energy"], speaker, emotion, frame_id, want_inputs=False)[0]
        if len(feats) > 1:
            anime_sum = anime_sum / float(len(feats))
        return feats[0]["tslist"], anime_sum, others

    @torch.no_grad()
    def generate_animation_batch(self, signals, speakers, emotions=0, frame_id=0, ensembling_ms=None, want_inputs=False):
        """Several utterances through ONE launch group: [(tslist, animes, others)] in the order of `signals`, each exactly
        what `generate_animation` returns for that clip alone (frames are independent and a column's features do not depend
        on the batch, so the rows are bitwise the single-clip rows).  `speakers`: one name / id, or one per clip."""
        signals = [self._check_signal(s) for s in signals]
        if isinstance(speakers, (str, int, np.integer)):
            speakers = [speakers] * len(signals)
        assert len(speakers) == len(signals)
        speakers = [self._speakers_dict[s] if isinstance(s, str) else s for s in speakers]
        if ensembling_ms is None:
            ensembling_ms = self.hp.ensembling_ms
        return self._animate(signals, speakers, ensembling_ms, want_inputs)

    def _animate(self, signals, speakers, ensembling_ms, want_inputs):
        from sdfa_amd.engine import frame_index
        eng = self._model._engine
        if eng is None:
            raise RuntimeError("no weights loaded: call load_state_dict first")
        sr = self.hp.audio.sample_rate
        for spk in speakers:
            assert isinstance(spk, (int, np.integer)), f"given index is {spk}, {type(spk)}"
            eng.check_speaker_ids(int(spk))
        ensemble = ensembling_ms is not None and ensembling_ms > 0
        # The reference keeps the features of the LAST signal (model.py:364-367,409-416): the same audio with another speaker does
        # not recompute the front end.  Here everything up to the encoder output z is speaker-independent, so the one-entry cache
        # holds z: a speaker sweep over one clip re-runs only the regressor (bitwise the full call: same z, same kernels after it).
        # The key carries everything that changes z for a given signal: rate, ensembling delay, precision mode and the library's
        # option epoch (sdfa_amd._lib.set_option; a switch set through the raw C call needs clear_signal_cache()).
        # hparams.signal_cache = False turns the cache off (it pins z and up to 256 MB of features on the device).
        from sdfa_amd import _lib as _sdfa_lib
        use_cache = bool(self.hp.get("signal_cache", True))
        key = (sr, int(ensembling_ms) if ensemble else 0, eng.precision, _sdfa_lib.option_epoch)
        c = getattr(self, "_signal_cache", None) if use_cache else None
        if (len(signals) == 1 and c is not None and c["key"] == key and c["signal"].shape == signals[0].shape
                and (c["feat"] is not None or not want_inputs) and np.array_equal(c["signal"], signals[0])):
            n = len(c["tslist"])
            spk = torch.full((n,), int(speakers[0]), dtype=torch.int64, device=eng.device)
            inputs_host = eng.to_host_async(c["feat"][:n].permute(0, 3, 2, 1)) if want_inputs else None
            rows = eng.forward_host(None, spk, z=c["z"], ops_key=self._model._key, wait=True, ensemble=ensemble)
            return [self._pack(rows.numpy(), None if inputs_host is None else inputs_host.numpy(), [list(c["tslist"])], [n])[0]]
        self._signal_cache = None
        tables = [frame_index(len(s), sr) for s in signals]             # ONE enumeration per clip (starts, tslist)
        clips = list(signals)
        if ensemble:                                                    # model.py:373-384: second pass on a delayed copy --
            pad = ensembling_ms * sr // 1000                            # here as further clips of the same launch group
            clips += [np.pad(s[:-pad], [[pad, 0]], "constant") for s in signals]
            tables = tables + tables
        feat, tslists, counts = eng.mel_frontend(clips, sr, tables=tables)
        share = eng.last_frame_table                                    # (clip, start, hop): the per-column stages run once per distinct column
        tslists, counts = tslists[:len(signals)], counts[:len(signals)]
        n = int(sum(counts))
        spk = torch.from_numpy(np.repeat(np.asarray(speakers, np.int64), counts)).to(eng.device, non_blocking=True)
        inputs_host = None
        if want_inputs:                                                 # others["inputs"] = audio_feat.permute(0, 3, 2, 1), model.py:463-466
            inputs_host = eng.to_host_async(feat[:n].permute(0, 3, 2, 1))
        rows = eng.forward_host(feat, spk, table=share, ops_key=self._model._key, wait=True, ensemble=ensemble)
        if use_cache and len(signals) == 1 and eng.last_z() is not None:   # one clip, one piece: remember (signal -> z) for a speaker sweep
            # z is 2 KB per frame; the features (others["inputs"] of a later hit) are 98 KB per frame and are kept up to 256 MB only
            keep_feat = feat if feat.numel() * 4 <= (256 << 20) else None
            self._signal_cache = {"key": key, "signal": signals[0].copy(), "tslist": list(tslists[0]), "z": eng.last_z(), "feat": keep_feat}
        return self._pack(rows.numpy(), inputs_host.numpy() if inputs_host is not None else None, tslists, counts)

    def _pack(self, rows_np, inputs_np, tslists, counts):
        shape = (-1, 9) if self._face_type == "dgrad_3d" else ()
        out, f0 = [], 0
        for ci, c in enumerate(counts):
            animes = rows_np[f0:f0 + c].reshape((c,) + shape) if shape else rows_np[f0:f0 + c]
            others = {"inputs": inputs_np[f0:f0 + c] if inputs_np is not None else None,
                      "phones": None, "latent": None, "latent_align": None, "formants": None}
            out.append((tslists[ci], animes, others))
            f0 += c
        return out

    # ---- model.py:428-489 ---------------------------------------------------------------------
    @torch.no_grad()
    def _feature_to_anime(self, feat_list, energy_list, speaker_id, emotion_id, frame_id, bs=100, want_inputs=True):
        """audio_feat (F,64,128,3) (numpy or tensor, any device) -> (animes (F, 9976, 9) | (F, 15069) float32 numpy, others).
        `bs` is accepted for signature parity; frames are independent, so the engine batches by its own chunk size and the rows
        reach the host through pinned memory while the next piece computes (Engine.forward_host)."""
        assert isinstance(speaker_id, (int, np.integer)), f"given index is {speaker_id}, {type(speaker_id)}"
        eng = self._model._engine
        eng.check_speaker_ids(int(speaker_id))      # before any device work: ids >= num_speakers raise like one_hot's scatter_
        feat = feat_list if torch.is_tensor(feat_list) else torch.from_numpy(np.asarray(feat_list, np.float32))
        feat = feat.to(eng.device, dtype=torch.float32).contiguous()
        n = feat.shape[0]
        inputs_host = eng.to_host_async(feat.permute(0, 3, 2, 1)) if want_inputs else None
        rows = eng.forward_host(feat, int(speaker_id), ops_key=self._model._key, wait=True)
        animes = rows.numpy()
        if self._face_type == "dgrad_3d":
            animes = animes.reshape(n, -1, 9)
        others = {"inputs": inputs_host.numpy() if inputs_host is not None else None,
                  "phones": None, "latent": None, "latent_align": None, "formants": None}
        return animes, others

    # ---- model.py:121-223 (host loop; rendering / mesh export replaced by a dgrad track dump) ----
    def evaluate(self, sources, experiment=None, in_trainer=False, **kwargs):
        """The reference's host loop over sources (model.py:152-212).  The reference calls generate_animation once per source; here
        the sources are taken in LAUNCH GROUPS -- as many consecutive clips as fit one piece of the engine (`max_frames` animation
        frames; kwargs["group_frames"] overrides) go through ONE generate_animation_batch call -- because frames are independent
        and a clip's rows do not depend on what it is batched with (bitwise, tests/test_surface_fast.py), so every per-clip result
        and every file written is exactly what the clip-by-clip loop produces, at the batch path's throughput."""
        sr = self.hp.audio.sample_rate
        output_dir = kwargs.get("output_dir") or "evaluate_results"
        target_db = kwargs.get("audio_target_db", self.hp.dataset_anime.audio_target_db)
        export_frames = kwargs.get("export_mesh_frames", not in_trainer)
        ens = kwargs.get("ensembling_ms")
        if ens is None:
            ens = self.hp.ensembling_ms
        eng = self._model._engine
        if eng is None:
            raise RuntimeError("no weights loaded: call load_state_dict first")
        from sdfa_amd.engine import frame_index
        limit = int(kwargs.get("group_frames") or eng.max_frames) // (2 if (ens is not None and ens > 0) else 1)
        # the reference's evaluate returns nothing; the mirror returns [(path, tslist, animes)] for callers that want the tracks (the
        # rows are views of pinned host memory: a long evaluation that only wants the files passes keep_results=False, as the CLI does)
        keep = bool(kwargs.get("keep_results", True))
        results, group, group_frames = [], [], 0

        def flush():
            nonlocal group, group_frames
            if not group:
                return
            outs = self.generate_animation_batch([g["signal"] for g in group], [g["spk"] for g in group], ensembling_ms=ens, want_inputs=False)
            total = sum(len(o[0]) for o in outs)
            track_all = eng.last_device_rows(total) if export_frames else None      # one piece: the rows are still on the device
            f0 = 0
            for g, (tslist, animes, _) in zip(group, outs):
                track = None if track_all is None else track_all[f0:f0 + len(tslist)]
                f0 += len(tslist)
                self._write_result(g, tslist, animes, track, output_dir, export_frames)
                if keep:
                    results.append((g["path"], tslist, animes))
            group, group_frames = [], 0

        # Utterance-level shards (north_star): with kwargs["shard"] = (rank, world) rank r takes a contiguous block of the
        # flat source list -- sizes differ by at most one -- and writes its own sources' files; frames are independent, so no exchange
        # is needed and every file is what the single-process run writes.  WITHOUT `shard` every source is processed, as the reference's
        # evaluate does (model.py:152-212), whatever RANK / WORLD_SIZE say: only the process entry point (`speech_anime.api.evaluate_model`,
        # i.e. `python -m torch.distributed.run ... -m speech_anime evaluate`) derives the shard from the environment and passes it down.
        flat = [rec for _, records in dict(sources).items() for rec in records]
        rank, world = kwargs.get("shard") or (0, 1)
        if world > 1:
            from sdfa_amd.dist import shard_range
            lo, hi = shard_range(len(flat), rank, world)
            print(f"[speech_anime] evaluate: shard {rank} of {world} takes sources [{lo}, {hi}) of {len(flat)} ({len(flat) - (hi - lo)} left to the other ranks)")
            flat = flat[lo:hi]
        for records in (flat,):
            for rec in records:
                path = rec[0]
                spk = "m1"
                for extra in rec[1:]:
                    if isinstance(extra, str) and extra.startswith("speaker="):
                        spk = extra.split("=", 1)[1]
                signal, sound_signal = _audio.load_source(path, sr, return_sound=True)         # eval_utils.py:76-86
                signal = _audio.rms_normalize(signal, target_db).astype(np.float32)            # model.py:165
                signal = self._check_signal(signal)
                n = int(frame_index(len(signal), sr)[0].shape[0])
                if group and group_frames + n > limit:
                    flush()
                group.append(dict(path=path, spk=spk, signal=signal, sound=sound_signal))
                group_frames += n
        flush()
        return results

    def _write_result(self, g, tslist, animes, track, output_dir, export_frames):
        """One source's files (model.py:195-212): tslist / track dumps, audio.wav, NNNNNN_dgrad.npy and, with a template, NNNNNN.obj."""
        fps = self.hp.anime.fps
        name = os.path.splitext(os.path.basename(g["path"]))[0]
        out_dir = os.path.join(output_dir, name)
        os.makedirs(out_dir, exist_ok=True)
        np.save(os.path.join(out_dir, "tslist.npy"), np.asarray(tslist, np.int64))
        np.save(os.path.join(out_dir, f"{self._face_type}.npy"), animes)
        if export_frames:                                                              # model.py:201-212
            from .. import viewer
            from sdfa_amd.seek import SeekPlan
            eng = self._model._engine
            if g["sound"] is not None:
                _audio.write_wav(os.path.join(out_dir, "audio.wav"), g["sound"], _audio.SOUND_SR)   # model.py:203
            # stream.seek for every video frame i at i * 1000 / fps, i = 0 .. int(tslist[-1] * fps / 1000), as ONE
            # device stage; with a template the mesh solve is fused into it (the blended track is only
            # materialised for the NNNNNN_dgrad.npy dump the reference also writes)
            plan = SeekPlan([tslist], fps, device=eng.device)
            if track is None:                                       # the group took several pieces: re-upload this clip's rows (model.py:200)
                track = torch.from_numpy(np.ascontiguousarray(animes, dtype=np.float32)).to(eng.device).reshape(len(tslist), -1)
            frames = plan.rows(track).cpu().numpy().reshape((plan.n_queries,) + animes.shape[1:])
            for i_frame, data_frame in enumerate(frames):
                np.save(os.path.join(out_dir, f"{i_frame:06d}_dgrad.npy"), data_frame)
            if viewer.has_template():          # --template_mesh given: seek + solve on the GPU, then .obj per frame
                if self._face_type == "dgrad_3d":
                    verts, faces = viewer.track_to_mesh(track, plan).cpu().numpy(), viewer.template_faces()
                else:
                    verts, faces = viewer.frames_to_mesh(frames.astype(np.float32), self._face_type)
                for i_frame in range(len(frames)):
                    viewer.write_obj(os.path.join(out_dir, f"{i_frame:06d}.obj"), verts[i_frame], faces)
        print(f"[speech_anime] {name}: {len(tslist)} animation frames -> {out_dir} (video rendering is outside this path)")

"""CPU port of the reference path on the reference's OWN operator library (PyTorch CPU: torch.stft, conv2d,
nn.LSTM, linear) -- what `bench.py`'s `cpu_baseline` times on the GPU box's host cores.

TEST INFRASTRUCTURE ONLY, like oracle/sdfa_oracle.py (which it reuses for frame indexing, weight folding and the
mel / Savitzky-Golay constants): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
It exists because the numpy oracle, written for exactness and readability, is ~5x slower than the reference's
torch operators on the same cores, which would understate the CPU side of the comparison.  It is NOT the reference
(that cannot travel to the GPU box): same arithmetic, batched differently -- the reference walks one frame at a time
through torch.stft (sliding_window.py:348-371) and 100 frames at a time through the model (model.py:450-461); here all
frames of a clip go through torch.stft together and the model runs in batches of `batch` frames.

Pinned by tests/test_oracle_golden.py against the same reference-generated fixtures as the numpy oracle.
"""
import numpy as np
import torch
import torch.nn.functional as F

import sdfa_oracle as O


def fetch_audio_features(pcm, sr, preemph=0.65, ref_db=20.0, top_db=80.0):
    """DatasetSlidingWindow.fetch_audio_features (sliding_window.py:324-377) with torch.stft as the reference uses it
    (saber/data/audio/features/spectrogram.py:82-98: center=False, Hamming window, onesided)."""
    pcm = np.asarray(pcm, np.float32)
    assert -1.0 <= pcm.min() and pcm.max() <= 1.0
    win, hop, sliding = O.frame_geometry(sr)
    starts, tslist = O.frame_index(len(pcm), sr)
    hamm, melw = O.mel_constants(sr, win)
    w = torch.from_numpy(O.cut_windows(pcm, starts, sliding))                    # (F, sliding)
    y = torch.cat([w[:, :1], w[:, 1:] - np.float32(preemph) * w[:, :-1]], 1)     # misc.py:8-17, per window
    spec = torch.stft(y, n_fft=win, hop_length=hop, win_length=win, window=torch.from_numpy(hamm), center=False,
                      onesided=True, return_complex=True)                        # (F, bins, T)
    power = spec.real ** 2 + spec.imag ** 2
    mel = torch.matmul(torch.from_numpy(melw), power)                            # (F, 128, T)
    db = 10.0 * torch.log10(torch.clamp(mel, min=float(O.EPS)))
    nrm = torch.clamp((db - ref_db + top_db) / top_db, 0.0, 1.0)                 # spectrogram.py:245-249
    m = nrm.double()
    T = m.shape[-1]

    def delta(c):                                                                # get_features.py:199-207
        k = torch.from_numpy(np.asarray(c, np.float64)).view(1, 1, 9)
        acc = F.conv1d(m.reshape(-1, 1, T), k).reshape(m.shape[0], m.shape[1], T - 8)
        return torch.cat([acc[..., :1].expand(-1, -1, 4), acc, acc[..., -1:].expand(-1, -1, 4)], -1)

    feat = torch.stack([m, delta(O._SG1), delta(O._SG2)], 1).float()             # (F, 3, 128, T)
    feat = feat.permute(0, 3, 2, 1).contiguous()                                 # (F, T, 128, 3)
    return dict(tslist=[int(t) for t in tslist], audio_feat=feat.numpy(), starts=starts)


class TorchOracle:
    """Eval-mode forward of SpeechDrivenAnimation (speech_anime/model/model.py:28-45) on torch CPU operators."""

    def __init__(self, state_dict, head="dgrad"):
        o = O.Oracle(state_dict, head)          # folded fp32 weights, reference key layout
        self.head = head
        t = torch.from_numpy
        self.conv = [(t(w)[..., None].contiguous(), t(b), t(s).view(1, -1, 1, 1), t(sh).view(1, -1, 1, 1)) for w, b, s, sh in o.conv]
        self.freq = torch.nn.LSTM(64, 128, 1, bias=True, batch_first=True, bidirectional=True)
        self.bilstm = torch.nn.LSTM(256, 256, 2, bias=False, batch_first=True, bidirectional=True)
        with torch.no_grad():
            for d, suf in enumerate(("", "_reverse")):
                w_ih, w_hh, b = o.freq[d]
                getattr(self.freq, "weight_ih_l0" + suf).copy_(t(w_ih))
                getattr(self.freq, "weight_hh_l0" + suf).copy_(t(w_hh))
                getattr(self.freq, "bias_ih_l0" + suf).copy_(t(b))
                getattr(self.freq, "bias_hh_l0" + suf).zero_()
                for l in (0, 1):
                    w_ih, w_hh, _ = o.bilstm[l][d]
                    getattr(self.bilstm, f"weight_ih_l{l}{suf}").copy_(t(w_ih))
                    getattr(self.bilstm, f"weight_hh_l{l}{suf}").copy_(t(w_hh))
        self.freq.eval(); self.bilstm.eval()
        self.freq_proj = tuple(t(x) for x in o.freq_proj)
        self.attn = {k: t(np.ascontiguousarray(v)) for k, v in o.attn.items()}
        conv_t = lambda wb: (t(wb[0]), t(wb[1]))
        self.trunk = [conv_t(x) for x in o.trunk]
        if head == "dgrad":
            self.scale = [conv_t(x) for x in o.scale]
            self.rotat = [conv_t(x) for x in o.rotat]
            self.pca_s, self.pca_r = conv_t(o.pca_s), conv_t(o.pca_r)
        else:
            self.pca = conv_t(o.pca)

    @torch.no_grad()
    def encoder(self, audio_feat):
        x = torch.as_tensor(audio_feat, dtype=torch.float32).permute(0, 3, 2, 1)          # (N,3,128,64)
        for i, (w, b, sc, sh) in enumerate(self.conv):                                      # conv2d.py:64-97, extend.py:94-101
            x = F.conv2d(x, w, b, padding=(w.shape[2] // 2, 0))
            x = F.leaky_relu(x, 0.2) * sc + sh
            if i < 2:
                x = F.max_pool2d(x, (2, 1))
        N, C, Fq, T = x.shape
        seq = x.permute(0, 3, 2, 1).reshape(N * T, Fq, C)                                   # freq_lstm.py:36-55
        h, _ = self.freq(seq)
        z = F.linear(h.reshape(N * T, Fq * 256), *self.freq_proj).reshape(N, T, 256)
        z, _ = self.bilstm(z)                                                               # rnn.py:20-21
        a = self.attn                                                                       # attentions.py:49-54,107-124
        q = torch.einsum("ock,nkc->no", a["conv"], z[:, 31:34, :])
        s = torch.matmul(torch.tanh(F.linear(q, a["wq"])[:, None, :] + F.linear(z, a["wk"]) + a["b"]), a["v"])
        align = torch.softmax(s, -1)
        return torch.einsum("nt,ntc->nc", align, z), align

    @torch.no_grad()
    def forward(self, audio_feat, speaker_id):
        z, align = self.encoder(audio_feat)
        n = z.shape[0]
        spk = torch.full((n,), int(speaker_id), dtype=torch.int64) if np.isscalar(speaker_id) else torch.as_tensor(np.asarray(speaker_id), dtype=torch.int64)
        c = F.one_hot(spk, 8).float()                                                       # modules/speaker.py:21-27
        zc = torch.cat([z, c], -1)
        fc = lambda x, wb, act: {"lrelu": lambda y: F.leaky_relu(y, 0.2), "tanh": torch.tanh, None: lambda y: y}[act](F.linear(x, *wb))
        if self.head == "dgrad":                                                            # output_module.py:51-89
            hc = torch.cat([fc(zc, self.trunk[0], "lrelu"), c], -1)
            cs = fc(fc(fc(hc, self.scale[0], "lrelu"), self.scale[1], "tanh"), self.scale[2], None)
            cr = fc(fc(fc(hc, self.rotat[0], "lrelu"), self.rotat[1], "tanh"), self.rotat[2], None)
            s = F.linear(cs, *self.pca_s).reshape(n, -1, 6)                                 # PcaInversion, :94-116
            r = F.linear(cr, *self.pca_r).reshape(n, -1, 3)
            out = torch.cat([s, r], -1).reshape(n, -1)                                      # model.py:246-257
        else:
            h = fc(fc(fc(zc, self.trunk[0], "lrelu"), self.trunk[1], "tanh"), self.trunk[2], None)
            out = F.linear(h, *self.pca)
        return out.numpy(), z.numpy(), align.numpy()


def generate_animation(oracle, pcm, sr, speaker_id, batch=256):
    """SaberSpeechDrivenAnimation.generate_animation (model.py:333-420), ensembling_ms = 0."""
    feats = fetch_audio_features(pcm, sr)
    x = feats["audio_feat"]
    outs = [oracle.forward(x[i:i + batch], speaker_id)[0] for i in range(0, len(x), batch)]
    return feats["tslist"], np.concatenate(outs, 0)

import sys, numpy as np, torch
sys.path.insert(0, "sdfa-2019_amd")
from sdfa_amd import synth
from sdfa_amd.engine import Engine
sd = synth.make_state_dict("dgrad", 1234)
eng = Engine(sd)
sr = 16000
clips = [synth.make_pcm(0, 2 * sr), synth.make_pcm(22, 9088)]
feat, ts, counts = eng.mel_frontend(clips, sr)
fc, fs, hop = eng.last_frame_table
print(fc.dtype, fs.dtype, hop, fc[:3], fs[:3])
n = feat.shape[0]
z0, a0 = eng.encoder(feat)
z0b, _ = eng.encoder(feat)
print("determinism unshared:", (z0 - z0b).abs().max().item())
eng.profile(True)
z1, a1 = eng.encoder(feat, frame_clip=fc, frame_start=fs, hop=hop)
torch.cuda.synchronize()
for st in ("share_map", "conv1", "share_expand"):
    try:
        print(st, eng.profile_ms(st))
    except Exception as e:
        print(st, "ERR", e)
print("n", n, "distinct", eng.distinct_columns(n), "of", n * 64)
print("dz", (z0 - z1).abs().max().item())

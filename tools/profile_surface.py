"""Where does a generate_animation_batch call spend its host time?  cProfile of one warm call (32 x 10 s @ 16 kHz) + wall-clock marks.
Usage (GPU box): python tools/profile_surface.py"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sdfa-2019_amd"))
import torch
from sdfa_amd import synth
from speech_anime.hparams import configure
from speech_anime.api import build_model
from speech_anime.datasets import DatasetSlidingWindow

sr = 16000
hp = configure(dict(mode="evaluate", custom_hparams="dgrad"))
hp.audio.set_key("sample_rate", sr)
DatasetSlidingWindow.hparams = None
model = build_model(hp, synth.make_state_dict("dgrad", 1234))
clips = [synth.make_pcm(c, 10 * sr) for c in range(32)]
for _ in range(2):
    model.generate_animation_batch(clips, "m1")
torch.cuda.synchronize()
t0 = time.perf_counter()
model.generate_animation_batch(clips, "m1")
print("warm call: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
model.generate_animation_batch(clips, "m1")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

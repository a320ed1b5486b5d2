"""The N = 8 memory plan as a test: per-rank device memory of `bench.py --gpus 8 --gather auto` from the size
functions the run itself uses -- sdfa_workspace_bytes, sdfa_frontend_workspace_bytes, frame_index, the gatherers' buffer shapes --
against the 288 GB of one MI355X and the figures DESIGN.md section 5 quotes.  No GPU: the C ABI's size functions are host code."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_n8_gather_auto_fits_one_mi355x():
    import bench
    p = bench.memory_plan(8, clips_per_gpu=32, seconds=10.0, sr=16000, head="dgrad", chunk=8192, gather="auto")
    assert p["frames_per_gpu"] == 20352                                   # 32 x 636
    og = p["output_and_gathered"]
    assert set(og) == {"dgrad", "expand"}                                 # auto times both, built in turn
    assert abs(og["dgrad"] - (7.309 + 58.473)) < 0.05                     # own rows 7.31 GB + gathered 58.5 GB
    assert abs(og["expand"] - 58.473) < 0.5                               # rows of all ranks, own written in place, + 0.35 GB of coefficients
    assert 25.5 < p["encoder_workspace"] < 26.1                           # 25.8 GB at Nc = 8192
    assert 1.9 < p["audio_feat"] < 2.1
    assert p["total_gb"] < 140 < 288                                      # "<= 140 GB of 288"
    # the one-shot direct form keeps two gathered buffers
    d = bench.memory_plan(8, gather="direct")
    assert abs(d["output_and_gathered"]["direct"] - 2 * 58.473) < 0.1 and d["total_gb"] < 288


def test_n1_plan_matches_what_the_driver_measured():
    """peak_device_memory_gb of the round-3 driver run without the PCIe twin's buffers was 36 GB."""
    import bench
    p = bench.memory_plan(1)
    assert set(p["output_and_gathered"]) == {"none"} and 34 < p["total_gb"] < 38


def test_offsets_stream_plan():
    import bench
    p = bench.memory_plan(8, clips_per_gpu=80, seconds=6.0, sr=8000, head="offsets", chunk=8192)
    assert p["total_gb"] < 100


def test_every_gather_mode_has_a_plan():
    """bench.py --gather coef / mesh / none at N > 1 must not fail in the reporting code (found by the two-rank rehearsal in round 4)."""
    import bench
    for g in ("auto", "dgrad", "expand", "direct", "coef", "mesh", "none"):
        p = bench.memory_plan(2, clips_per_gpu=4, gather=g)
        assert p["total_gb"] > 0 and p["output_and_gathered"]

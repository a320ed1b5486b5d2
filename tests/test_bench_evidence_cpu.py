"""bench.py's counter-derived figures (roofline.traffic, frontend.hbm_gbps_counters, attention.mfma_util_pct_counters) come from
committed rocprofv3 passes, each stamped with the sha1 of the kernel source it was measured at: a figure must go null -- never stale
-- once that source has changed.  And profiles/pmc_summary.py must pick the attention stage's dispatches by position."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_counter_figures_are_nulled_when_the_kernel_source_changed(tmp_path, monkeypatch):
    bench = _load("bench_for_test", "bench.py")
    prof = tmp_path / "profiles" / "r99_pmc"
    prof.mkdir(parents=True)
    (prof / "freq_lstm_traffic.json").write_text(json.dumps({"frames": 8192, "traffic_bytes": 1000, "algorithmic_bytes": 800, "lstm_hip_sha1": "L"}))
    (prof / "frontend_traffic.json").write_text(json.dumps({"bytes_per_frame": 164000.0, "frontend_hip_sha1": "F"}))
    (prof / "attention_mfma.json").write_text(json.dumps({"mfma_util_pct_time_weighted": 50.6, "attn_hip_sha1": "A", "gemm_hip_sha1": "G"}))
    (prof / "attention_mfma_bf16x3_attention.json").write_text(json.dumps({"mfma_util_pct_time_weighted": 21.5, "mfma_util_pct_gemms": 33.0, "attn_hip_sha1": "A", "gemm_hip_sha1": "G"}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    sha = {"lstm.hip": "L", "frontend.hip": "F", "attn.hip": "A", "gemm.hip": "G"}
    monkeypatch.setattr(bench, "_sha1", lambda name: sha[name])
    assert bench.traffic_from_profile(4096)[:2] == (500, 400)                  # scaled to the frames of a launch
    assert bench.frontend_counter_bytes()[0] == 164000.0
    assert bench.attention_counter_util()[0] == 50.6 and bench.attention_counter_util()[2] is None
    assert bench.attention_counter_util("bf16x3_attention") == (21.5, os.path.join("profiles", "r99_pmc", "attention_mfma_bf16x3_attention.json"), 33.0)
    for changed in ("lstm.hip", "frontend.hip", "attn.hip", "gemm.hip"):
        sha2 = dict(sha, **{changed: "other"})
        monkeypatch.setattr(bench, "_sha1", lambda name, s=sha2: s[name])
        t, alg, why = bench.traffic_from_profile(4096)
        assert (t is None and alg == 400 and "stale" in why) if changed == "lstm.hip" else t == 500
        assert (bench.frontend_counter_bytes()[0] is None) == (changed == "frontend.hip")
        assert (bench.attention_counter_util()[0] is None) == (changed in ("attn.hip", "gemm.hip"))
        assert (bench.attention_counter_util("bf16x3_attention")[0] is None) == (changed in ("attn.hip", "gemm.hip"))


def test_committed_counter_files_match_the_committed_kernels():
    """The tree as committed must not carry stale evidence: the newest profiles/r*_pmc JSONs were measured at the csrc/ files next to
    them (re-run tools/collect_profiles.sh + tools/install_profiles.sh after touching lstm.hip / frontend.hip / attn.hip / gemm.hip)."""
    bench = _load("bench_for_test2", "bench.py")
    t, alg, src = bench.traffic_from_profile(8192)
    assert t is not None and t > alg > 0, src
    assert bench.frontend_counter_bytes()[0] is not None
    assert bench.attention_counter_util()[0] is not None


def test_attention_stage_is_picked_by_dispatch_position(tmp_path):
    pmc = _load("pmc_summary_for_test", "profiles/pmc_summary.py")
    hdr = '"Correlation_Id","Dispatch_Id","Grid_Size","Kernel_Name","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"\n'
    seq = [("conv123_kernel", 75, 100), ("time_lstm_kernel<2>", 84, 100), ("gemm_fat_kernel<0>", 90, 100), ("time_lstm_kernel<2>", 84, 100),
           ("gemm_k4_kernel<0>", 80, 300), ("gemm_k4_kernel<0>", 60, 100), ("attn_kernel", 0, 100), ("gemm_k4_kernel<1>", 50, 1000), ("pca_dgrad_res_kernel", 70, 100),
           # second launch group
           ("time_lstm_kernel<1>", 82, 100), ("time_lstm_kernel<1>", 82, 100), ("gemm_k4_kernel<0>", 70, 100), ("attn_kernel", 0, 100), ("gemm_k4_kernel<1>", 50, 1000),
           # third launch group (round 6): the whole layer in one launch ends the stage too
           ("time_lstm_kernel<2>", 84, 100), ("time_lstm_kernel<2>", 84, 100), ("gemm_k4_kernel<0>", 60, 100), ("attn_fused_f32_kernel", 75, 400), ("gemm_k4_kernel<1>", 50, 1000)]
    rows, t = [], 0
    for i, (name, util, dur) in enumerate(seq):
        rows.append(f'{i},{i},256,"void (anonymous namespace)::{name}(Args)","MfmaUtil",{util},{t},{t + dur}\n')
        t += dur + 10
    path = tmp_path / "MfmaUtil_counter_collection.csv"
    path.write_text(hdr + "".join(rows))
    util, ns, per = pmc.attention_stage(str(path))
    assert ns == 300 + 100 + 100 + 100 + 100 + 100 + 400                       # the MLP GEMMs behind attn_kernel / attn_fused_f32_kernel are not the attention stage
    assert abs(util - (80 * 300 + 60 * 100 + 70 * 100 + 60 * 100 + 75 * 400) / 1200) < 1e-9
    assert {e["kernel"]: e["calls"] for e in per} == {"gemm_k4_kernel<0>": 4, "attn_kernel": 2, "attn_fused_f32_kernel": 1}


def test_launcher_argv_and_self_launch_relay(tmp_path, monkeypatch, capfd):
    """`python3 bench.py --gpus N` with no launcher starts its own ranks as a CHILD under torch.distributed.run (127.0.0.1 rendezvous, same
    arguments), passes rank 0's line through and exits with the child's code; a child that fails, or ends without the line, is a
    non-zero exit -- never a silent N = 1."""
    import pytest
    bench = _load("bench_for_test3", "bench.py")
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "3"]
    cmd = bench.launcher_argv(8, argv, 29555, python="py")
    assert cmd[:4] == ["py", "-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv
    a = bench.parse(argv)
    assert a.gpus == 8 and a.all_legs is False and a.inject_failure is None

    def fake(code, line):
        script = tmp_path / f"child{code}{len(line)}.py"
        script.write_text(f"import os, sys\nassert os.environ['SDFA_BENCH_LAUNCHER'] == 'self'\nprint('noise')\n"
                          f"print({line!r}) if {bool(line)!r} else None\nsys.exit({code})\n")
        monkeypatch.setattr(bench, "launcher_argv", lambda n, av, port, python=None: [sys.executable, str(script)])

    fake(0, '{"metric": "m", "value": 1.0, "n_gpus": 8}')
    with pytest.raises(SystemExit) as e:
        bench.self_launch(a, argv)
    assert e.value.code == 0
    out = capfd.readouterr().out.splitlines()
    assert out[-1] == '{"metric": "m", "value": 1.0, "n_gpus": 8}' and json.loads(out[-1])["n_gpus"] == 8
    fake(3, '')
    with pytest.raises(SystemExit) as e:
        bench.self_launch(a, argv)
    assert e.value.code == 3                                                   # the child's code, not a fall-back
    fake(0, '')
    with pytest.raises(SystemExit) as e:
        bench.self_launch(a, argv)
    assert e.value.code == 1                                                   # ended "fine" without the line: still an error


def test_committed_precision_sweep_is_inside_the_budget_and_reported():
    """bench.py reports `mixed_precision.max_abs_dgrad_err_vs_cpu_ref` as the WORST case of the committed wide sweep (other weight dynamics,
    the reference's 10 s fixture, full size: tests/test_gpu_precision.py), not the fixture case: the committed table must exist, cover the
    modes that claim the 1e-4 budget, and hold it."""
    bench = _load("bench_for_test4", "bench.py")
    worst, src = bench.precision_worst_case()
    assert src is not None and src.startswith("profiles/r")
    assert {"fp32", "bf16x6", "bf16x3_attention", "bf16x3"} <= set(worst)
    assert all(0.0 < worst[m] <= 1e-4 for m in worst), worst
    assert worst["bf16x6"] <= 1e-5 and worst["fp32"] <= 1e-5 and worst["bf16x3"] > worst["bf16x6"]
    table = json.load(open(os.path.join(ROOT, src)))
    assert len(table["cases"]) >= 7 and all(m in table["bounds_asserted"] for m in worst)


def test_budget_skips_legs_that_do_not_fit_and_both_lines_are_printed(capfd):
    """bench.py prints the headline right after the timed steps and again, complete, as the LAST line; a leg that is not
    expected to fit the wall-clock budget is never started and is listed in `legs_skipped`; time kept back for the required leg
    (cpu_baseline) cannot be eaten by the optional ones."""
    bench = _load("bench_for_test5", "bench.py")
    now = {"t": 100.0}
    b = bench.Budget(60.0, clock=lambda: now["t"], t0=100.0, reserve_s=20.0)
    ran = []

    def work(name, cost):
        def fn():
            ran.append(name)
            now["t"] += cost
            return name
        return fn
    res = {"metric": "m", "value": 1.0, "cpu_baseline": None}
    bench.emit(res, final=False)                                              # the early line
    assert b.run("column_sharing", 10.0, work("column_sharing", 12.0)) == "column_sharing"      # 60 left >= 10 + 20 reserved
    assert b.run("bf16x3", 25.0, work("bf16x3", 25.0)) == "bf16x3"                              # 48 left >= 25 + 20
    assert b.run("surface", 30.0, work("surface", 30.0)) is None                                # 23 left < 30 + 20: skipped, not started
    assert b.run("tiny", 2.0, work("tiny", 2.0)) == "tiny"                                      # 23 >= 2 + 20
    assert b.run("cpu_baseline", 20.0, work("cpu_baseline", 19.0), required=True) == "cpu_baseline"   # required: the reserve is its own
    assert b.run("late", 5.0, work("late", 5.0)) is None
    assert ran == ["column_sharing", "bf16x3", "tiny", "cpu_baseline"]
    assert [s["leg"] for s in b.skipped] == ["surface", "late"] and b.skipped[0]["estimate_s"] == 30.0
    assert b.seconds == {"column_sharing": 12.0, "bf16x3": 25.0, "tiny": 2.0, "cpu_baseline": 19.0}
    res["cpu_baseline"] = {"value": 127.0}
    res["legs_skipped"] = b.skipped
    res["wall_clock"] = b.report()
    bench.emit(res, final=True)
    first, last = [json.loads(l) for l in capfd.readouterr().out.splitlines()]
    assert first["partial"] is True and first["value"] == 1.0 and first["cpu_baseline"] is None
    assert last["partial"] is False and last["value"] == first["value"] and last["cpu_baseline"]["value"] == 127.0
    assert [s["leg"] for s in last["legs_skipped"]] == ["surface", "late"] and last["wall_clock"]["budget_s"] == 60.0
    # a leg that raises still records its time and the exception propagates to the caller's own handler
    import pytest
    with pytest.raises(ZeroDivisionError):
        b2 = bench.Budget(10.0, clock=lambda: now["t"], t0=now["t"])
        b2.run("boom", 1.0, lambda: 1 / 0)
    # unlimited budget (N > 1 with --all-legs: every rank must take the same legs): admits everything, reports null
    binf = bench.Budget(float("inf"), clock=lambda: now["t"], t0=0.0)
    assert binf.admit("anything", 1e9) and binf.report()["budget_s"] is None and json.dumps(binf.report())


def test_thread_scan_is_bounded():
    """cpu_baseline's thread scan (the 256-thread probe cost minutes): one pass per count, stop at 1.5 x the best or at the cap."""
    bench = _load("bench_for_test6", "bench.py")
    now = {"t": 0.0, "nt": None}
    cost = {8: 1.0, 16: 0.8, 32: 1.3, 64: 50.0, 128: 500.0}
    calls = []

    def probe():
        calls.append(now["nt"])
        now["t"] += cost[now["nt"]]
    best, log = bench.thread_scan(probe, [8, 16, 32, 64, 128], lambda n: now.update(nt=n), clock=lambda: now["t"], cap_s=100.0)
    assert best == 16 and calls == [8, 16, 32] and [c for c, _ in log] == [8, 16, 32]       # 1.3 > 1.5 x 0.8: 64 and 128 never run
    now.update(t=0.0); calls.clear()
    cost.update({8: 3.0, 16: 2.9, 32: 2.8})
    best, log = bench.thread_scan(probe, [8, 16, 32, 64], lambda n: now.update(nt=n), clock=lambda: now["t"], cap_s=5.0)
    assert calls == [8, 16] and best == 16                                                    # the cap ends the scan

"""bench.py as the driver starts it: `python3 bench.py --gpus N` with NO launcher must start its own ranks and
print one line that proves what ran; an optional leg that raises must not cost the headline line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--clips-per-gpu", "2", "--seconds", "2", "--chunk", "256", "--no-cpu-baseline", "--no-host-io", "--no-surface"]


def _run(*args, expect=0):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SDFA_BENCH_LAUNCHER"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=1200, env=env)
    assert res.returncode == expect, (res.returncode, res.stderr[-3000:])
    lines = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    return lines, res


def test_gpus_2_without_a_launcher_starts_its_own_ranks():
    """Two ranks on this box's ONE GPU over gloo (a bookkeeping rehearsal: the collectives' data is staged through the host): the line
    says n_gpus 2, the process group saw 2 ranks, every rank's record is there, and the gathered rows checksum."""
    lines, res = _run("--gpus", "2", "--backend", "gloo", "--gather", "expand", *SMALL)
    assert len(lines) == 2, res.stdout[-2000:]                             # the early headline line, then the complete one (last)
    early, d = json.loads(lines[0]), json.loads(lines[-1])
    assert early["partial"] is True and d["partial"] is False and early["value"] == d["value"] and early["roofline"] == d["roofline"]
    cfg = d["config"]
    assert d["n_gpus"] == 2 and cfg["world_size_seen"] == 2 and cfg["backend"] == "gloo" and cfg["launcher"] == "self"
    assert cfg["gather"] == "expand" and cfg["gather_checksum_ok"] is True
    assert [r["rank"] for r in cfg["devices"]] == [0, 1] and len({r["pid"] for r in cfg["devices"]}) == 2
    assert cfg["distinct_devices"] == 1                                   # both ranks on the one card, and the line says so
    assert "column_sharing" not in d and "mixed_precision" not in d and "skipped at N > 1" in d["optional_legs"]
    assert d["value"] > 0 and d["scaling"] == "weak"


def test_a_failed_child_is_a_failed_bench():
    lines, res = _run("--gpus", "2", "--backend", "gloo", "--gather", "direct", "--head", "offsets", "--mesh-stage", *SMALL, expect=1)
    assert not lines                                                        # "the mesh stage consumes dgrad rows": both ranks exit, so does the parent


@pytest.mark.parametrize("leg", ["column_sharing", "bf16x6"])
def test_an_optional_leg_that_raises_does_not_cost_the_headline(leg):
    lines, res = _run("--gpus", "1", "--inject-failure", leg, *SMALL)
    d = json.loads(lines[-1])
    assert d["value"] > 0 and d["dtype"] == "f32" and d["n_gpus"] == 1 and d["roofline"]["frac"] > 0
    assert d["roofline"]["frac_executed"] < d["roofline"]["frac"] and d["config"]["world_size_seen"] == 1
    if leg == "column_sharing":
        assert "injected failure" in d["column_sharing"]["error"] and d["mixed_precision"]["value"] > 0
    else:
        assert "injected failure" in d["mixed_precision"]["errors"]["bf16x6"] and d["mixed_precision"]["bf16x6"]["value"] is None
        assert d["mixed_precision"]["value"] > 0 and d["column_sharing"]["value"] > 0
    assert "device_unusable_after_optional_leg" not in d


def test_a_tiny_budget_skips_the_legs_and_both_lines_are_still_printed():
    """`--budget-s` smaller than what the set-up alone takes -- every optional leg AND the cpu_baseline leg are skipped (listed in
    `legs_skipped`), the headline is printed twice (early, then complete as the last line) with the same value."""
    small = [x for x in SMALL if x not in ("--no-cpu-baseline", "--no-host-io", "--no-surface")]
    lines, res = _run("--gpus", "1", "--budget-s", "1", *small)
    assert len(lines) == 2
    early, d = json.loads(lines[0]), json.loads(lines[-1])
    assert early["partial"] is True and d["partial"] is False and early["value"] == d["value"] > 0 and d["roofline"]["frac"] > 0
    skipped = [s["leg"] for s in d["legs_skipped"]]
    assert skipped == ["column_sharing", "bf16x3", "bf16x3_attention", "bf16x6", "bf16x3_column_sharing", "with_h2d_d2h", "surface", "cpu_baseline"]
    assert d["cpu_baseline"]["value"] is None and "did not fit" in d["cpu_baseline"]["sample"]
    assert "column_sharing" not in d and "surface" not in d and d["mixed_precision"]["value"] is None
    assert d["wall_clock"]["budget_s"] == 1.0 and d["wall_clock"]["legs_s"] == {}

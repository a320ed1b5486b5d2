"""One-GPU rehearsal of the N > 1 exchange through the REAL backend: `bench.py --gpus 1 --force-gather` creates a
world-size-1 "nccl" (= RCCL) process group and runs the per-chunk asynchronous all_gather_into_tensor + Work.wait() of
sdfa_amd/dist.py next to the persistent kernels; the integer checksum of the gathered rows must equal the owner's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--clips-per-gpu", "3", "--seconds", "2",
           "--chunk", "128", "--no-cpu-baseline", "--no-mixed-precision", "--no-column-sharing", "--no-host-io", "--no-surface", "--force-gather", *extra]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("mode", ["dgrad", "expand"])
def test_rccl_world1_gather_runs_and_checksums(mode):
    line = _bench("--gather", mode, "--backend", "nccl")
    cfg = line["config"]
    assert cfg["gather"] == mode and cfg["backend"] == "nccl" and cfg["force_gather_world1"] is True
    assert cfg["gather_checksum_ok"] is True
    assert line["n_gpus"] == 1 and line["value"] > 0


def test_reserved_cus_under_the_gather():
    line = _bench("--gather", "dgrad", "--backend", "nccl", "--reserve-cus", "16")
    assert line["config"]["reserved_cus"] == 16 and line["config"]["gather_checksum_ok"] is True

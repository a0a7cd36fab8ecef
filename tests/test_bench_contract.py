"""bench.py contract: `python bench.py --gpus N` launches its own ranks (the parent never touches a GPU), prints ONE
JSON line with the fields the driver reads, and reports both exchange modes for N > 1."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def test_parent_builds_a_torchrun_command_without_touching_the_gpu(monkeypatch, capsys):
    """The N > 1 parent only spawns `python -m torch.distributed.run ... bench.py <same args>` and relays the line."""
    sys.path.insert(0, str(ROOT))
    import bench
    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = iter(["noise\n", '{"metric": "Mvoxel-corr/s", "value": 1.0}\n'])

        def wait(self):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert "torch" not in bench.__dict__          # the module itself imports no torch at top level
    bench.main()
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert json.loads(capsys.readouterr().out)["value"] == 1.0


def test_traffic_hash_tracks_the_kernel_sources():
    sys.path.insert(0, str(ROOT))
    import bench
    a = bench.kernel_source_sha256("pearson")
    assert a == bench.kernel_source_sha256("pearson") and len(a) == 64
    assert a != bench.kernel_source_sha256("mi_kraskov")


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ, **(env_extra or {}))
    r = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_single_gpu_line():
    d = _run(["--steps", "5", "--warmup", "2", "--repeats", "2", "--grid", "128", "128", "64", "--spinup-ms", "50"])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0
    assert d["parity"]["bit_identical"] == d["parity"]["checked_voxels"]
    assert d["cpu_baseline"]["cpu_model"] and d["cpu_baseline"]["threads_used"] >= 1
    assert d["host_boundary"]["resident_ms"] > 0 and d["host_boundary"]["fresh_ms"] > 0
    assert d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]


@pytest.mark.gpu
def test_two_ranks_launch_themselves_and_report_both_modes():
    """`python bench.py --gpus 2` on the 1-GPU box: the parent spawns two ranks that share the card and exchange over
    gloo (a rehearsal of the N > 1 path, flagged in the output)."""
    d = _run(["--gpus", "2", "--steps", "5", "--warmup", "2", "--repeats", "2", "--grid", "128", "128", "64",
              "--spinup-ms", "50"], {"CRF_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert "gloo" in d["backend"]
    assert d["throughput"]["lookahead"] == 16 and d["latency"]["lookahead"] == 0
    assert d["throughput"]["value"] > 0 and d["latency"]["value"] > 0
    assert d["value"] == d["throughput"]["value"]

"""GPU: the multi-rank z-slab path on real kernels -- 2 and 3 processes (sharing the box's one GPU, gloo rendezvous on
127.0.0.1) each evaluate their slab; every slab must equal the corresponding part of the unsharded oracle result."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_on_gpu(world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29511 + world), str(HERE / "dist_gpu_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"SHARDED-OK {world}" in r.stdout

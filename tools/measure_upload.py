#!/usr/bin/env python3
"""Development tool: host -> HBM upload rate of crf_upload_members (pageable host volumes, like the reference's
host field cache entries)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import correrender_amd as ca

xs = ys = zs = 256
cs = 64
ens = np.random.default_rng(0).standard_normal((cs, zs, ys, xs), dtype=np.float32)
eng = ca.CorrField(0)
eng.set_grid(xs, ys, zs, cs)
eng.upload_members(ens)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    eng.upload_members(ens)
    dt = time.perf_counter() - t0
    print(f"crf_upload_members: {ens.nbytes / 1e9:.2f} GB in {dt * 1e3:.1f} ms = {ens.nbytes / dt / 1e9:.1f} GB/s")

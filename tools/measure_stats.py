#!/usr/bin/env python3
"""Times the ensemble mean / spread kernels at 256^3 x 64 (HIP events through the library's profiling interface)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import correrender_amd as ca
xs = ys = zs = 256
cs = 64
eng = ca.CorrField(0)
eng.set_grid(xs, ys, zs, cs)
members = torch.empty((cs, zs, ys, xs), dtype=torch.float32, device="cuda")
for c in range(cs):
    eng.synth_box_member(members[c], xs, ys, zs, 0, zs, c, cs, 1)
torch.cuda.synchronize()
eng.bind_members(members)
out = torch.empty(xs * ys * zs, dtype=torch.float32, device="cuda")
B = xs * ys * zs * (4 * cs + 4)
for kind, name in ((0, "mean"), (1, "spread")):
    for _ in range(200):
        eng.ensemble_stat_device(kind, out)
    torch.cuda.synchronize()
    eng.set_profiling(True)
    eng.take_kernel_time()
    for _ in range(50):
        eng.ensemble_stat_device(kind, out)
    torch.cuda.synchronize()
    ms, n = eng.take_kernel_time()
    eng.set_profiling(False)
    print(f"ensemble {name}: {ms / n:.4f} ms  {B / (ms / n) / 1e6:.0f} GB/s  {B / (ms / n) / 1e6 / 8000:.1%} of 8 TB/s  kernel={eng.last_kernel_name()}")

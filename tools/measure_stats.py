#!/usr/bin/env python3
"""Times the sibling per-voxel reductions (ensemble mean / spread, set predicate, DKL) at 256^3 x 64 (HIP events
through the library's profiling interface)."""
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
CASES = [
    ("ensemble mean", lambda: eng.ensemble_stat_device(0, out), 200, 50),
    ("ensemble spread", lambda: eng.ensemble_stat_device(1, out), 200, 50),
    ("set predicate >", lambda: eng.set_predicate_device(">", 0.25, cs // 2, cs // 2, out), 200, 50),
    ("DKL binned (80 bins)", lambda: eng.dkl_device("binned", out, num_bins=80), 2, 5),
    ("DKL entropy k-NN (k=2)", lambda: eng.dkl_device("knn", out, k=2), 2, 5),
]
for name, run, warm, iters in CASES:
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    eng.set_profiling(True)
    eng.take_kernel_time()
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    ms, n = eng.take_kernel_time()
    eng.set_profiling(False)
    print(f"{name}: {ms / n:.4f} ms  {B / (ms / n) / 1e6:.0f} GB/s  {B / (ms / n) / 1e6 / 8000:.1%} of 8 TB/s  kernel={eng.last_kernel_name()}")

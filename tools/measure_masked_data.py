#!/usr/bin/env python3
"""Kernel time of every estimator on an ensemble in which a fraction of the voxels holds NaN (missing values; in every member, or in every other member) or the same
value in every member (a mask) -- whole regions of such voxels are the norm in real ensembles.  256^3 x 64 by default.
usage: measure_masked_data.py [--members N] [--fraction F]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import correrender_amd as ca

args = sys.argv[1:]
cs = int(args[args.index("--members") + 1]) if "--members" in args else 64
frac = float(args[args.index("--fraction") + 1]) if "--fraction" in args else 0.3
symmetric = "--symmetric" in args   # SEPARATE_SYMMETRIC field mode: a second ensemble with the same mask
xs = ys = zs = 256
n = xs * ys * zs
stream = torch.cuda.current_stream().cuda_stream
out = torch.empty(n, dtype=torch.float32, device="cuda")
print(f"grid {xs}^3 x {cs} members; {frac:.0%} of the voxels masked; kernel ms per evaluation")
print(f"{'mask':>8s} " + " ".join(f"{m:>12s}" for m in ["pearson", "spearman", "kendall", "mi_binned", "mi_kraskov"]))
kinds = args[args.index("--kinds") + 1].split(",") if "--kinds" in args else ["none", "nan", "nan_some", "zero"]
for kind in kinds:
    eng = ca.CorrField(0)
    eng.set_grid(xs, ys, zs, cs)
    block = torch.empty(cs * n, dtype=torch.float32, device="cuda")
    members = [block[c * n:(c + 1) * n] for c in range(cs)]
    for c in range(cs):
        eng.synth_box_member(members[c], xs, ys, zs, 0, zs, c, cs, 1234, stream)
    torch.cuda.synchronize()   # the library's generator kernels run on another stream than torch's fill below
    if kind != "none":
        if kind == "nan_some":   # missing values in every other member only
            block.view(cs, n)[::2, :int(frac * n)] = float("nan")
        else:
            block.view(cs, n)[:, :int(frac * n)] = float("nan") if kind == "nan" else 0.0
    torch.cuda.synchronize()
    eng.bind_members(members)
    if symmetric:
        block2 = torch.empty(cs * n, dtype=torch.float32, device="cuda")
        members2 = [block2[c * n:(c + 1) * n] for c in range(cs)]
        for c in range(cs):
            eng.synth_box_member(members2[c], xs, ys, zs, 0, zs, c, cs, 4321, stream)
        torch.cuda.synchronize()
        if kind == "nan_some":
            block2.view(cs, n)[1::2, :int(frac * n)] = float("nan")
        elif kind != "none":
            block2.view(cs, n)[:, :int(frac * n)] = float("nan") if kind == "nan" else 0.0
        torch.cuda.synchronize()
        eng.bind_secondary_members(members2)
    eng.set_profiling(True)
    row = []
    for name in ["pearson", "spearman", "kendall", "mi_binned", "mi_kraskov"]:
        measure = ca.Measure(ca.MEASURE_IDS.index(name))
        kw = dict(k=ca.default_kraskov_k(cs), symmetric=symmetric)
        if name == "mi_binned":
            kw.update(minmax_ref=(-4.0, 4.0), minmax_query=(-4.0, 4.0), num_bins=80)
        eng.compute_device(measure, out, (1, 2, 200), stream=stream, **kw)
        torch.cuda.synchronize()
        eng.take_kernel_time()
        iters = 2 if name == "mi_kraskov" else 4
        for i in range(iters):
            eng.compute_device(measure, out, (17 * i + 3, 29, 200 + i), stream=stream, **kw)
        torch.cuda.synchronize()
        ms, cnt = eng.take_kernel_time()
        row.append(ms / cnt)
    print(f"{kind:>8s} " + " ".join(f"{v:12.3f}" for v in row), flush=True)
    del eng, block, members

#!/usr/bin/env python3
"""Pearson at 321..1216 members: the r02 kernels (CRF_PEARSON_SPLIT=0: VGPRs + AGPRs at one wave per SIMD up to 384, the
8-wave relay up to 512, three sweeps beyond) vs the lanes-per-voxel kernel (pearson_split_kernel): whole-field bit-identity
and kernel time, same process.  usage: measure_pearson_wide.py [xs ys zs] [--members a b c ...]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import correrender_amd as ca

args = sys.argv[1:]
mask_kind, mask_fraction = None, 0.0   # --mask nan|zero FRACTION: the first FRACTION of the voxels holds NaN / 0 in every member
if "--mask" in args:
    i = args.index("--mask")
    mask_kind, mask_fraction = args[i + 1], float(args[i + 2])
    args = args[:i] + args[i + 3:]
members_list = [321, 352, 384, 400, 448, 480, 500, 512, 544, 576, 600, 640, 704, 768, 896, 1000, 1024, 1100, 1216]
if "--members" in args:
    i = args.index("--members")
    members_list = [int(a) for a in args[i + 1:]]
    args = args[:i]
xs, ys, zs = (int(a) for a in args[:3]) if len(args) >= 3 else (256, 256, 64)
n = xs * ys * zs
stream = torch.cuda.current_stream().cuda_stream
print(f"grid {xs}x{ys}x{zs}" + (f", {mask_kind} in the first {mask_fraction:.0%} of the voxels of every member" if mask_kind else "") + f"; kernel ms (best of 2 rounds of 3 evaluations); TB/s = (4 cs + 4) bytes per voxel / kernel time")
for cs in members_list:
    eng = ca.CorrField(0)
    eng.set_grid(xs, ys, zs, cs)
    block = torch.empty(cs * n, dtype=torch.float32, device="cuda")
    members = [block[c * n:(c + 1) * n] for c in range(cs)]
    for c in range(cs):
        eng.synth_box_member(members[c], xs, ys, zs, 0, zs, c, cs, 1234, stream)
    torch.cuda.synchronize()   # the library's generator kernels run on another stream than torch's fill below
    if mask_kind:
        block.view(cs, n)[:, :int(mask_fraction * n)] = float("nan") if mask_kind == "nan" else 0.0
    torch.cuda.synchronize()
    eng.bind_members(members)
    eng.set_profiling(True)
    outs, times, names = {}, {}, {}
    for rnd in range(2):
        for mode in ("0", "1"):
            os.environ["CRF_PEARSON_SPLIT"] = mode
            out = torch.full((n,), -7.0, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            eng.compute_device(ca.Measure.PEARSON, out, (1, 2, 3), stream=stream)
            torch.cuda.synchronize()
            eng.take_kernel_time()
            for i in range(3):
                eng.compute_device(ca.Measure.PEARSON, out, (17 * i + 5, 29, 31), stream=stream)
            torch.cuda.synchronize()
            ms, cnt = eng.take_kernel_time()
            times.setdefault(mode, []).append(ms / cnt)
            outs[mode] = out
            names[mode] = eng.last_kernel_name() if hasattr(eng, "last_kernel_name") else ""
    same = bool(((outs["0"].view(torch.int32) == outs["1"].view(torch.int32)) | (outs["0"].isnan() & outs["1"].isnan())).all())
    gb = (4 * cs + 4) * n / 1e9
    t0, t1 = min(times["0"]), min(times["1"])
    print(f"{cs:5d} members  r02 {t0:8.3f} ms {gb / t0:6.2f} TB/s   split {t1:8.3f} ms {gb / t1:6.2f} TB/s ({100 * gb / t1 / 8:4.1f} %)  "
          f"bit-identical: {same}  {names.get('1', '')}", flush=True)
    del eng, block, members, outs

#!/usr/bin/env python3
"""Tile-free Kraskov kernel with the voxel tile staged in LDS (CRF_KRASKOV_STAGE=1) vs the shipped form: bit-identity of
the whole 256^3 field and kernel time, several member counts / k."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import correrender_amd as ca

xs = ys = zs = 256
n = xs * ys * zs
stream = torch.cuda.current_stream().cuda_stream
for cs, k in [(32, 3), (48, 3), (64, 2), (64, 3), (64, 4), (80, 3), (100, 3), (128, 3), (128, 4)]:
    eng = ca.CorrField(0)
    eng.set_grid(xs, ys, zs, cs)
    block = torch.empty(cs * n, dtype=torch.float32, device="cuda")
    members = [block[c * n:(c + 1) * n] for c in range(cs)]
    for c in range(cs):
        eng.synth_box_member(members[c], xs, ys, zs, 0, zs, c, cs, 1234, stream)
    torch.cuda.synchronize()
    eng.bind_members(members)
    eng.set_profiling(True)
    outs, times = {}, {}
    os.environ["CRF_KRASKOV_DIRECT"] = "1"
    for rnd in range(2):
        for mode in ("0", "1"):
            os.environ["CRF_KRASKOV_STAGE"] = mode
            out = torch.empty(n, dtype=torch.float32, device="cuda")
            eng.compute_device(ca.Measure.MUTUAL_INFORMATION_KRASKOV, out, (1, 2, 3), stream=stream, k=k)
            torch.cuda.synchronize()
            eng.take_kernel_time()
            for i in range(2):
                eng.compute_device(ca.Measure.MUTUAL_INFORMATION_KRASKOV, out, (17 * i + 5, 29, 31), stream=stream, k=k)
            torch.cuda.synchronize()
            ms, cnt = eng.take_kernel_time()
            times.setdefault(mode, []).append(ms / cnt)
            outs[mode] = out
    same = bool(torch.equal(outs["0"].view(torch.int32), outs["1"].view(torch.int32)))
    print(f"cs={cs:4d} k={k}  shipped {min(times['0']):8.3f} ms   staged {min(times['1']):8.3f} ms   bit-identical: {same}", flush=True)
    del eng, block, members, outs

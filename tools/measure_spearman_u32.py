#!/usr/bin/env python3
"""Spearman at 65..128 members: split-sort kernel (shipped) vs one u32-composite network (CRF_RANK_U32=1): whole-field
bit-identity and kernel time at 256^3 (and 512^3 x 128 with --big)."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import correrender_amd as ca

cases = [(256, 72), (256, 80), (256, 96), (256, 100), (256, 112), (256, 128)]
if "--big" in sys.argv:
    cases.append((512, 128))
stream = torch.cuda.current_stream().cuda_stream
for g, cs in cases:
    xs = ys = zs = g
    n = xs * ys * zs
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
    for rnd in range(2):
        for mode in ("0", "1"):
            os.environ["CRF_RANK_U32"] = mode
            out = torch.empty(n, dtype=torch.float32, device="cuda")
            eng.compute_device(ca.Measure.SPEARMAN, out, (1, 2, 3), stream=stream)
            torch.cuda.synchronize()
            eng.take_kernel_time()
            for i in range(3):
                eng.compute_device(ca.Measure.SPEARMAN, out, (17 * i + 5, 29, 31), stream=stream)
            torch.cuda.synchronize()
            ms, cnt = eng.take_kernel_time()
            times.setdefault(mode, []).append(ms / cnt)
            outs[mode] = out
    same = bool(torch.equal(outs["0"].view(torch.int32), outs["1"].view(torch.int32)))
    print(f"{g}^3 x {cs:3d}  split {min(times['0']):8.3f} ms   u32 network {min(times['1']):8.3f} ms   bit-identical: {same}", flush=True)
    del eng, block, members, outs

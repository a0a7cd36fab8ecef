#!/usr/bin/env python3
"""Times crf_compute (host output buffer: kernel + D2H of 4 bytes/voxel + sync) next to crf_compute_device at
256^3 x 64 -- the PCIe-inclusive rate quoted in DESIGN.md (never bench.py's `value`)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
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
for measure in (ca.Measure.PEARSON,):
    for _ in range(3):
        eng.compute(measure, (10, 20, 30))
    t0 = time.perf_counter()
    n = 20
    for i in range(n):
        eng.compute(measure, (10 + i, 20, 30))
    dt = (time.perf_counter() - t0) / n
    print(f"{measure.name}: crf_compute (host buffer, pageable) {dt * 1e3:.3f} ms/evaluation = {xs * ys * zs / dt / 1e6:.0f} Mvoxel-corr/s")
    out = torch.empty(xs * ys * zs, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        eng.compute_device(measure, out, (10 + i, 20, 30), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{measure.name}: crf_compute_device {dt * 1e3:.3f} ms/evaluation = {xs * ys * zs / dt / 1e6:.0f} Mvoxel-corr/s")

# the same with ONE reused (already touched) destination buffer: isolates first-touch page faults of a fresh buffer
import ctypes as C
from correrender_amd._lib import CrfParams
buf = np.empty(xs * ys * zs, np.float32)
p = CrfParams()
p.measure = 0
p.ref_x, p.ref_y, p.ref_z = 10, 20, 30
ptr = buf.ctypes.data_as(C.POINTER(C.c_float))
for _ in range(3):
    eng._lib.crf_compute(eng._ctx, C.byref(p), ptr)
t0 = time.perf_counter()
for i in range(n):
    p.ref_x = 10 + i
    eng._lib.crf_compute(eng._ctx, C.byref(p), ptr)
dt = (time.perf_counter() - t0) / n
print(f"PEARSON: crf_compute into a reused host buffer {dt * 1e3:.3f} ms/evaluation = {xs * ys * zs / dt / 1e6:.0f} Mvoxel-corr/s")

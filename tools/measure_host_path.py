#!/usr/bin/env python3
"""Times crf_compute (host output buffer: kernel + D2H of 4 bytes/voxel) at 256^3 x 64 into a resident and into a
fresh destination, for several (voxel ranges, copier threads) settings -- the PCIe-inclusive rate of DESIGN.md / the
host_boundary record of bench.py (never bench.py's `value`).  One process per setting (the range tables are built once
per context)."""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def one():
    import numpy as np
    import torch
    import correrender_amd as ca
    xs = ys = zs = 256
    cs = 64
    measure = ca.Measure(ca.MEASURE_IDS.index(os.environ.get("CRF_MEASURE", "pearson")))
    eng = ca.CorrField(0)
    eng.set_grid(xs, ys, zs, cs)
    members = torch.empty((cs, zs, ys, xs), dtype=torch.float32, device="cuda")
    for c in range(cs):
        eng.synth_box_member(members[c], xs, ys, zs, 0, zs, c, cs, 1)
    torch.cuda.synchronize()
    eng.bind_members(members)
    kw = dict(k=3) if "kraskov" in measure.name.lower() else {}
    resident = np.zeros((zs, ys, xs), np.float32)
    for i in range(3):
        eng.compute(measure, (10, 20, 30), out=resident, **kw)
    n = int(os.environ.get("CRF_REPS", "15"))
    t_res, t_fresh = [], []
    for i in range(n):
        t0 = time.perf_counter()
        eng.compute(measure, (10 + i, 20, 30), out=resident, **kw)
        t_res.append(time.perf_counter() - t0)
    for i in range(n):
        fresh = np.empty((zs, ys, xs), np.float32)
        t0 = time.perf_counter()
        eng.compute(measure, (10 + i, 20, 30), out=fresh, **kw)
        t_fresh.append(time.perf_counter() - t0)
        del fresh
    out = torch.empty(xs * ys * zs, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        eng.compute_device(measure, out, (10 + i, 20, 30), stream=torch.cuda.current_stream().cuda_stream, **kw)
    torch.cuda.synchronize()
    dev = (time.perf_counter() - t0) / n
    med = lambda v: sorted(v)[len(v) // 2]
    print(json.dumps({"variant": os.environ.get("CRF_VARIANT_NAME", "default"),
                      "chunks": os.environ.get("CRF_HOST_CHUNKS", "auto"), "threads": os.environ.get("CRF_COPY_THREADS", "auto"),
                      "resident_ms": round(med(t_res) * 1e3, 3), "resident_min_ms": round(min(t_res) * 1e3, 3),
                      "fresh_ms": round(med(t_fresh) * 1e3, 3), "device_only_ms": round(dev * 1e3, 3)}))


def ab():
    """Same-process A/B of range layouts (the tables are rebuilt when the members are bound again): three rounds over
    the variants, medians per variant."""
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
    variants = {k: {} for k in (os.environ.get("CRF_AB_SHARES") or
                                "16,14,11,8,6,4,3,2;6,12,12,12,10,6,4,2;4,8,8,8,8,8,8,6,3,2,1;8,8,8,8,8,8,8,8").split(";")}
    res = {k: [] for k in variants}
    fresh_res = {k: [] for k in variants}
    resident = np.zeros((zs, ys, xs), np.float32)
    for rnd in range(3):
        for shares in variants:
            os.environ["CRF_HOST_SHARES"] = shares
            eng.bind_members(members)            # drops the range tables
            for i in range(3):
                eng.compute(ca.Measure.PEARSON, (10, 20, 30), out=resident)
            for i in range(12):
                t0 = time.perf_counter()
                eng.compute(ca.Measure.PEARSON, (10 + i, 20, 30), out=resident)
                res[shares].append(time.perf_counter() - t0)
            for i in range(8):
                fresh = np.empty((zs, ys, xs), np.float32)
                t0 = time.perf_counter()
                eng.compute(ca.Measure.PEARSON, (10 + i, 20, 30), out=fresh)
                fresh_res[shares].append(time.perf_counter() - t0)
                del fresh
    med = lambda v: sorted(v)[len(v) // 2]
    for shares in variants:
        print(json.dumps({"shares_of_64": shares, "resident_ms": round(med(res[shares]) * 1e3, 3),
                          "resident_min_ms": round(min(res[shares]) * 1e3, 3), "fresh_ms": round(med(fresh_res[shares]) * 1e3, 3)}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ab":
        ab()
    elif len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        import torch
        d = torch.empty(256 ** 3, dtype=torch.float32, device="cuda")
        h = torch.empty(256 ** 3, dtype=torch.float32).pin_memory()
        for _ in range(3):
            h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(json.dumps({"pure_dma_d2h_pinned_ms": round(dt * 1e3, 3), "GB/s": round(d.numel() * 4 / dt / 1e9, 1)}))
        del d, h
        variants = [
            ("default (mapped staging, shrinking ranges, 2 streams, populate + THP)", {}),
            ("one stream", {"CRF_HOST_STREAMS": "1"}),
            ("no huge pages", {"CRF_HOST_HUGEPAGE": "0"}),
            ("touch instead of populate", {"CRF_HOST_FAULT": "1"}),
            ("no pre-faulting", {"CRF_HOST_FAULT": "0"}),
            ("dma engine instead of mapped stores", {"CRF_HOST_PATH": "dma"}),
            ("8 equal ranges", {"CRF_HOST_CHUNKS": "8"}),
            ("16 equal ranges", {"CRF_HOST_CHUNKS": "16"}),
            ("4 copier threads", {"CRF_COPY_THREADS": "4"}),
            ("8 copier threads", {"CRF_COPY_THREADS": "8"}),
            ("16 copier threads", {"CRF_COPY_THREADS": "16"}),
            ("plain: kernel, then one hipMemcpy", {"CRF_PLAIN_D2H": "1"}),
        ]
        for name, extra in variants:
            env = dict(os.environ, CRF_VARIANT_NAME=name, **extra)
            subprocess.run([sys.executable, __file__, "one"], env=env, check=True)

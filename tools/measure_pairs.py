#!/usr/bin/env python3
"""Development tool: throughput of the pair-request mode (crf_compute_requests_device) -- random voxel pairs."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ctypes as C
import numpy as np
import torch
import correrender_amd as ca
from correrender_amd._lib import CrfParams

xs = ys = zs = 128
n_req = 1 << 20
for cs in (32, 64, 100):
    eng = ca.CorrField(0)
    eng.set_grid(xs, ys, zs, cs)
    members = torch.empty((cs, zs, ys, xs), dtype=torch.float32, device="cuda")
    for c in range(cs):
        eng.synth_box_member(members[c], xs, ys, zs, 0, zs, c, cs, 1)
    torch.cuda.synchronize()
    eng.bind_members(members)
    rng = np.random.default_rng(0)
    req = np.zeros((n_req, 8), np.uint32)
    req[:, 0:3] = rng.integers(0, xs, (n_req, 3))
    req[:, 4:7] = rng.integers(0, xs, (n_req, 3))
    d_req = torch.from_numpy(req.view(np.int32)).cuda()
    out = torch.empty(n_req, dtype=torch.float32, device="cuda")
    row = []
    for name in ("pearson", "spearman", "kendall", "mi_binned", "mi_kraskov"):
        p = CrfParams()
        p.measure = ca.MEASURE_IDS.index(name)
        p.k = 3
        p.num_bins = 80
        def run():
            rc = eng._lib.crf_compute_requests_device(eng._ctx, C.byref(p), C.c_void_p(d_req.data_ptr()), n_req,
                                                      C.c_void_p(out.data_ptr()), C.c_void_p(0))
            assert rc == 0, eng._lib.crf_last_error(eng._ctx)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()                      # asynchronous on the context's stream
        torch.cuda.synchronize()   # device-wide: includes that stream
        dt = time.perf_counter() - t0
        row.append(f"{name} {n_req / dt / 1e6:7.2f} Mreq/s")
    print(f"cs={cs:4d}: " + "  ".join(row), flush=True)
    del eng, members

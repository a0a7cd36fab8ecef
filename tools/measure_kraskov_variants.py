#!/usr/bin/env python3
"""Tile kernel (LDS column) vs tile-free kernel for the Kraskov estimator at several member counts and k (256^3)."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def one(cs_list, k):
    import torch
    import correrender_amd as ca
    xs = ys = zs = 256
    n = xs * ys * zs
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for cs in cs_list:
        eng = ca.CorrField(0)
        eng.set_grid(xs, ys, zs, cs)
        block = torch.empty(cs * n, dtype=torch.float32, device="cuda")
        members = [block[c * n:(c + 1) * n] for c in range(cs)]
        for c in range(cs):
            eng.synth_box_member(members[c], xs, ys, zs, 0, zs, c, cs, 1234, stream)
        torch.cuda.synchronize()
        eng.bind_members(members)
        eng.set_profiling(True)
        eng.compute_device(ca.Measure.MUTUAL_INFORMATION_KRASKOV, out, (1, 2, 3), stream=stream, k=k)
        torch.cuda.synchronize()
        eng.take_kernel_time()
        for i in range(2):
            eng.compute_device(ca.Measure.MUTUAL_INFORMATION_KRASKOV, out, (17 * i + 5, 29, 31), stream=stream, k=k)
        torch.cuda.synchronize()
        ms, cnt = eng.take_kernel_time()
        print(f"cs={cs:4d} k={k} {os.environ.get('CRF_KRASKOV_DIRECT', '0')=} {eng.last_kernel_name():24s} {ms / cnt:9.3f} ms", flush=True)
        del eng, block, members


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one([int(c) for c in sys.argv[3].split(",")], int(sys.argv[2]))
    else:
        for k in (1, 2, 3, 4):
            for direct in ("0", "1"):
                subprocess.run([sys.executable, __file__, "one", str(k), "16,32,48,64,72,80"],
                               env=dict(os.environ, CRF_KRASKOV_DIRECT=direct), check=True)

#!/usr/bin/env python3
"""Kernel-tuning harness (development tool, not part of the product or the benchmark): times the per-voxel kernel
through the C ABI's own HIP-event instrumentation for a list of environment-selected variants, interleaved
round-robin in ONE process (cdna_hip_programming.md section 5.4 rule 24)."""
import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import correrender_amd as ca


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, nargs=3, default=[256, 256, 256])
    ap.add_argument("--members", type=int, default=64)
    ap.add_argument("--measure", default="pearson")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--pads", type=int, nargs="*", default=[0])
    ap.add_argument("--variants", nargs="*", default=[""], help="each: 'ENV=val,ENV2=val'")
    args = ap.parse_args()
    xs, ys, zs = args.grid
    cs = args.members
    n = xs * ys * zs
    measure = ca.Measure(ca.MEASURE_IDS.index(args.measure))
    eng = ca.CorrField(0)
    eng.set_grid(xs, ys, zs, cs)
    stream = torch.cuda.current_stream().cuda_stream
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    results = {}
    for pad in args.pads:
        block = torch.empty(cs * (n + pad), dtype=torch.float32, device="cuda")
        members = [block[c * (n + pad): c * (n + pad) + n] for c in range(cs)]
        for c in range(cs):
            eng.synth_box_member(members[c], xs, ys, zs, 0, zs, c, cs, 1234, stream)
        torch.cuda.synchronize()
        eng.bind_members(members)
        eng.set_profiling(True)
        for rnd in range(args.rounds):
            for var in args.variants:
                saved = {}
                for kv in filter(None, var.split(",")):
                    k, v = kv.split("=")
                    saved[k] = os.environ.get(k)
                    os.environ[k] = v
                eng.take_kernel_time()
                for i in range(args.iters):
                    eng.compute_device(measure, out, ((17 * i + rnd) % xs, (29 * i) % ys, (31 * i + 3) % zs),
                                       stream=stream, k=3)
                torch.cuda.synchronize()
                ms, cnt = eng.take_kernel_time()
                results.setdefault((pad, var), []).append(ms / cnt)
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        del block, members
    bytes_alg = n * (4 * cs + 4)
    for (pad, var), v in results.items():
        v = sorted(v)
        med, mn = v[len(v) // 2], v[0]
        print(f"pad={pad:8d} {var or '(default)':40s} median {med:8.4f} ms  min {mn:8.4f} ms  "
              f"{bytes_alg / med / 1e6:8.1f} GB/s  {bytes_alg / med / 1e6 / 8000:6.1%} of 8 TB/s  "
              f"kernel={eng.last_kernel_name()}", flush=True)


if __name__ == "__main__":
    main()

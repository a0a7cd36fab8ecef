#!/usr/bin/env python3
"""Development tool: kernel time of every measure over a sweep of member counts (256^3 grid by default), through the C
ABI's HIP-event instrumentation.  Finds member counts that fall onto a slow instantiation."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import correrender_amd as ca


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, nargs=3, default=[256, 256, 256])
    ap.add_argument("--members", type=int, nargs="*", default=[8, 12, 16, 24, 32, 40, 48, 64, 96, 100, 128])
    ap.add_argument("--measures", nargs="*", default=["pearson", "spearman", "kendall", "mi_binned", "mi_kraskov"])
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--symmetric", action="store_true", help="SEPARATE_SYMMETRIC field mode (two ensembles)")
    ap.add_argument("--kraskov-k", type=int, default=0,
                    help="fixed k for the Kraskov estimator (default: the reference's ceil(3 cs / 100), which steps "
                         "with the member count); also prints ms / cs^2")
    args = ap.parse_args()
    xs, ys, zs = args.grid
    n = xs * ys * zs
    stream = torch.cuda.current_stream().cuda_stream
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    print(f"grid {xs}x{ys}x{zs}; kernel ms per point")
    print("members " + " ".join(f"{m:>12s}" for m in args.measures))
    for cs in args.members:
        eng = ca.CorrField(0)
        eng.set_grid(xs, ys, zs, cs)
        block = torch.empty(cs * n, dtype=torch.float32, device="cuda")
        members = [block[c * n:(c + 1) * n] for c in range(cs)]
        for c in range(cs):
            eng.synth_box_member(members[c], xs, ys, zs, 0, zs, c, cs, 1234, stream)
        torch.cuda.synchronize()
        eng.bind_members(members)
        if args.symmetric:
            block2 = torch.empty(cs * n, dtype=torch.float32, device="cuda")
            members2 = [block2[c * n:(c + 1) * n] for c in range(cs)]
            for c in range(cs):
                eng.synth_box_member(members2[c], xs, ys, zs, 0, zs, c, cs, 4321, stream)
            torch.cuda.synchronize()
            eng.bind_secondary_members(members2)
        eng.set_profiling(True)
        row = []
        for name in args.measures:
            measure = ca.Measure(ca.MEASURE_IDS.index(name))
            iters = 2 if name == "mi_kraskov" else args.iters
            kw = dict(k=args.kraskov_k or ca.default_kraskov_k(cs), symmetric=args.symmetric)
            if name == "mi_binned":
                mm = eng.member_minmax()
                kw.update(minmax_ref=mm, minmax_query=eng.secondary_member_minmax() if args.symmetric else mm,
                          num_bins=80)
            eng.compute_device(measure, out, (1, 2, 3), stream=stream, **kw)
            torch.cuda.synchronize()
            eng.take_kernel_time()
            for i in range(iters):
                eng.compute_device(measure, out, ((17 * i) % xs, (29 * i) % ys, (31 * i + 3) % zs), stream=stream, **kw)
            torch.cuda.synchronize()
            ms, cnt = eng.take_kernel_time()
            row.append(ms / cnt)
        extra = ""
        if args.kraskov_k and "mi_kraskov" in args.measures:
            extra = f"   k={args.kraskov_k}: {row[args.measures.index('mi_kraskov')] / (cs * cs) * 1e3:.3f} us per member^2"
        print(f"{cs:7d} " + " ".join(f"{v:12.3f}" for v in row) + extra, flush=True)
        del eng, block, members


if __name__ == "__main__":
    main()

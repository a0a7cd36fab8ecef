#!/usr/bin/env python3
"""Times the in-library device group (crf_group_compute: z-slabs, worker thread per device, reference-vector exchange,
ranged host output) next to a single context's crf_compute on the same volume.  On a one-GPU box the group is rehearsed
with a repeated ordinal (all slabs on the same card: the numbers show the group's OVERHEAD, not a speed-up); on a
multi-GPU node pass distinct ordinals, e.g. --devices 0,1,2,3,4,5,6,7."""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import correrender_amd as ca


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, nargs=3, default=[256, 256, 256])
    ap.add_argument("--members", type=int, default=64)
    ap.add_argument("--measure", default="pearson")
    ap.add_argument("--devices", nargs="*", default=["0", "0,0", "0,0,0,0"])
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    xs, ys, zs = args.grid
    cs = args.members
    measure = ca.Measure(ca.MEASURE_IDS.index(args.measure))
    rng = np.random.default_rng(1)
    ens = rng.standard_normal((cs, zs, ys, xs), dtype=np.float32)
    out = np.zeros((zs, ys, xs), np.float32)
    kw = dict(k=3) if "kraskov" in args.measure else {}
    pts = [((7 * i) % xs, (11 * i) % ys, (13 * i) % zs) for i in range(args.reps + 3)]
    med = lambda v: sorted(v)[len(v) // 2]

    def run(compute):
        for p in pts[:3]:
            compute(p)
        t = []
        for p in pts[3:]:
            t0 = time.perf_counter()
            compute(p)
            t.append(time.perf_counter() - t0)
        return round(med(t) * 1e3, 3), round(min(t) * 1e3, 3)

    with ca.CorrField(0) as eng:
        eng.set_grid(xs, ys, zs, cs)
        eng.upload_members(ens)
        want = eng.compute(measure, pts[-1], **kw).copy()
        m, lo = run(lambda p: eng.compute(measure, p, out=out, **kw))
        import torch
        dev_out = torch.empty(xs * ys * zs, dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for p in pts[:3]:
            eng.compute_device(measure, dev_out, p, stream=stream, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for p in pts[3:]:
            eng.compute_device(measure, dev_out, p, stream=stream, **kw)
        torch.cuda.synchronize()
        back_to_back = (time.perf_counter() - t0) / len(pts[3:])
        eng.set_profiling(True)
        eng.compute_device(measure, dev_out, pts[0], stream=stream, **kw)
        torch.cuda.synchronize()
        kernel_ms, _ = eng.take_kernel_time()
        eng.set_profiling(False)
        print(json.dumps({"path": "single context (crf_compute)", "ms_median": m, "ms_min": lo,
                          "device_resident_back_to_back_ms": round(back_to_back * 1e3, 4),
                          "kernel_ms_hip_events": round(kernel_ms, 4)}), flush=True)
    for spec in args.devices:
        devices = [int(d) for d in spec.split(",")]
        with ca.CorrFieldGroup(devices) as grp:
            grp.set_grid(xs, ys, zs, cs)
            grp.upload_members(ens)
            m, lo = run(lambda p: grp.compute(measure, p, out=out, **kw))
            same = bool((grp.compute(measure, pts[-1], **kw).view(np.uint32) == want.view(np.uint32)).all())
            import torch
            outs = [torch.empty(xs * ys * grp.slab(s)[1], dtype=torch.float32, device=f"cuda:{d}")
                    for s, d in enumerate(devices)]
            md, lod = run(lambda p: grp.compute_device(measure, outs, p, **kw))
            # the batch call: B reference points per hand-off, amortised time per evaluation
            B = args.batch
            rows = [outs] * B          # every evaluation into the same buffers: only the timing matters here
            bt = []
            for rep in range(5):
                refs = [pts[(rep * B + i) % len(pts)] for i in range(B)]
                t0 = time.perf_counter()
                grp.compute_batch_device(measure, refs, rows, **kw)
                bt.append((time.perf_counter() - t0) / B)
            grp.set_profiling(True)
            grp.compute_device(measure, outs, pts[0], **kw)
            kernel_ms, launches = grp.take_kernel_time()
            grp.set_profiling(False)
            print(json.dumps({"path": f"crf_group over devices {devices}", "exchange": grp.exchange, "ms_median": m,
                              "ms_min": lo, "device_resident_ms_median": md, "device_resident_ms_min": lod,
                              f"batch{B}_device_resident_ms_per_evaluation_median": round(med(bt) * 1e3, 4),
                              f"batch{B}_device_resident_ms_per_evaluation_min": round(min(bt) * 1e3, 4),
                              "slowest_slot_kernel_ms_when_profiled": round(kernel_ms, 4),
                              "bit_identical_to_single_context": same}), flush=True)


if __name__ == "__main__":
    main()

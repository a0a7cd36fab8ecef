"""Worker of tests/test_gpu_sharded.py: one rank of a z-slab sharded evaluation on a real GPU (ranks may share GPU 0;
the process group is gloo so that this runs on a 1-GPU box -- the production backend is nccl/RCCL, same code path in
correrender_amd.distributed)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import correrender_amd as ca
    from correrender_amd import Measure, synth
    from correrender_amd.distributed import ShardedCorrField, slab_bounds
    import oracle_lib

    xs, ys, zs, cs = 24, 16, 10, 32
    ens = synth.box_ensemble(xs, ys, zs, cs, seed=5)
    dev = torch.device("cuda", 0)
    eng = ca.CorrField(0)
    sharded = ShardedCorrField(eng, (xs, ys, zs), cs, device=dev)
    z0, zl = sharded.z_begin, sharded.z_count
    members = torch.from_numpy(ens[:, z0:z0 + zl].copy()).to(dev)
    sharded.bind_members(members)
    oracle = oracle_lib.load_oracle()
    gmm = oracle.minmax(ens)
    assert sharded.global_minmax() == gmm
    stream = torch.cuda.current_stream().cuda_stream
    bad = []
    for measure in (Measure.PEARSON, Measure.SPEARMAN, Measure.KENDALL, Measure.MUTUAL_INFORMATION_BINNED,
                    Measure.MUTUAL_INFORMATION_KRASKOV):
        points = [(1, 2, 0), (12, 8, zs // 2), (23, 15, zs - 1), (5, 5, 1)]
        batched = measure in (Measure.SPEARMAN, Measure.MUTUAL_INFORMATION_BINNED)
        if batched:
            # one collective for all four reference vectors; for binned MI the reference-side preparation of the four
            # evaluations runs on the communication stream as well (prepared slots)
            prep = (measure, dict(k=2)) if measure == Measure.MUTUAL_INFORMATION_BINNED else None
            sharded.prefetch_batch(points, prepare=prep)
        else:
            sharded.prefetch(points[0])                  # pipelined exchange: step i+1's broadcast overlaps step i
        for pi, (x, y, z) in enumerate(points):
            if not batched and pi + 1 < len(points):
                sharded.prefetch(points[pi + 1])
            out = torch.empty(xs * ys * zl, dtype=torch.float32, device=dev)
            sharded.compute(measure, out, (x, y, z), k=2)
            torch.cuda.synchronize()
            kw = dict(k=2, minmax_ref=gmm) if measure != Measure.PEARSON else {}
            want = oracle.field(int(measure), ens, ens[:, z, y, x].copy(), **kw)
            lo = z0 * ys * xs
            want = want[lo:lo + xs * ys * zl]
            got = out.cpu().numpy()
            if int(measure) <= 2:
                ok = ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))).all()
            else:
                ok = np.allclose(got, want, rtol=1e-5, atol=1e-6, equal_nan=True)
            if not ok:
                diff = np.flatnonzero(~((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))))
                bad.append((rank, measure.name, (x, y, z), int(diff.size), int(diff[0]), float(got[diff[0]]),
                            float(want[diff[0]])))
    # compute_batch: the prepared evaluations of a whole batch launched with one library call
    # (crf_compute_prepared_device), two batches double-buffered as in bench.py's throughput mode
    for measure in (Measure.PEARSON, Measure.KENDALL, Measure.MUTUAL_INFORMATION_KRASKOV):
        points = [(1, 2, 0), (12, 8, zs // 2), (23, 15, zs - 1), (5, 5, 1), (7, 0, 3), (20, 9, zs - 2), (2, 14, 4)]
        first, second = points[:4], points[4:]
        prep = (measure, dict(k=2))
        outs = [torch.empty(xs * ys * zl, dtype=torch.float32, device=dev) for _ in points]
        sharded.prefetch_batch(first, prepare=prep)
        sharded.prefetch_batch(second, prepare=prep)          # exchanged while the first batch is consumed
        sharded.compute_batch(measure, outs[:4], first, k=2)
        sharded.compute_batch(measure, outs[4:], second, k=2)
        torch.cuda.synchronize()
        for (x, y, z), out in zip(points, outs):
            kw = dict(k=2, minmax_ref=gmm) if measure != Measure.PEARSON else {}
            want = oracle.field(int(measure), ens, ens[:, z, y, x].copy(), **kw)[z0 * ys * xs:(z0 + zl) * ys * xs]
            got = out.cpu().numpy()
            if int(measure) <= 2:
                ok = ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))).all()
            else:
                ok = np.allclose(got, want, rtol=1e-5, atol=1e-6, equal_nan=True)
            if not ok:
                bad.append((rank, "compute_batch " + measure.name, (x, y, z)))
    flag = torch.tensor([len(bad)])
    dist.all_reduce(flag)
    eng.close()
    dist.destroy_process_group()
    if int(flag) != 0:
        print("MISMATCH", bad)
        sys.exit(1)
    if rank == 0:
        print("SHARDED-OK", world)


if __name__ == "__main__":
    main()

"""CPU, world_size 2 (and 3, uneven slabs) over gloo: the z-slab sharded path -- slab arithmetic, reference-point
ownership, reference-vector broadcast, global min/max all-reduce -- assembled result identical to the unsharded one."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from correrender_amd import Measure, synth
from correrender_amd.distributed import ShardedCorrField, slab_bounds, slab_owner

HERE = Path(__file__).resolve().parent


def test_slab_arithmetic():
    for zs in (1, 7, 8, 256, 257):
        for world in (1, 2, 3, 8):
            if world > zs:
                continue
            covered = []
            for r in range(world):
                z0, n = slab_bounds(zs, world, r)
                assert n >= zs // world
                covered += list(range(z0, z0 + n))
                for z in range(z0, z0 + n):
                    assert slab_owner(zs, world, z) == (r, z - z0)
            assert covered == list(range(zs))
    with pytest.raises(ValueError):
        slab_owner(8, 2, 8)


def test_more_ranks_than_slices_is_refused_on_every_rank():
    """world > zs: every rank raises before any collective (an empty rank raising alone would leave the others
    waiting in an all-reduce).  slab_owner stays well defined for world > zs."""
    from fake_engine import OracleEngine
    for rank in range(4):
        with pytest.raises(ValueError, match="cannot share"):
            ShardedCorrField(OracleEngine(), (4, 4, 3), 8, rank=rank, world=4)
    assert [slab_owner(3, 4, z) for z in range(3)] == [(0, 0), (1, 0), (2, 0)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, zs, result_file):
    sys.path.insert(0, str(HERE))
    sys.path.insert(0, str(HERE.parent))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_engine import OracleEngine
        import oracle_lib
        xs, ys, cs = 12, 10, 16
        ens = synth.box_ensemble(xs, ys, zs, cs, seed=77)        # every rank regenerates the same whole volume ...
        z0, zl = slab_bounds(zs, world, rank)
        sharded = ShardedCorrField(OracleEngine(), (xs, ys, zs), cs)
        assert (sharded.z_begin, sharded.z_count) == (z0, zl)
        sharded.bind_members(torch.from_numpy(ens[:, z0:z0 + zl].copy()))   # ... and keeps only its slab
        oracle = oracle_lib.load_oracle()
        gmm = oracle.minmax(ens)
        assert sharded.global_minmax() == gmm                   # all-reduce of the slab extrema
        failures = []
        points = [(3, 4, 0), (5, 5, zs // 2), (11, 9, zs - 1), (0, 0, zs // 2 - 1), (6, 2, zs // 2)]
        for measure in (Measure.PEARSON, Measure.SPEARMAN, Measure.KENDALL, Measure.MUTUAL_INFORMATION_BINNED,
                        Measure.MUTUAL_INFORMATION_KRASKOV):
            # exercise all three exchange forms: per-point prefetch(), batched prefetch_batch(), plain compute()
            pipelined = measure in (Measure.PEARSON, Measure.KENDALL)
            batched = measure == Measure.SPEARMAN
            double_buffered = measure == Measure.MUTUAL_INFORMATION_KRASKOV   # batch i+1 exchanged before batch i is consumed
            if pipelined:
                sharded.prefetch(points[0])
            prep = (measure, dict(k=2)) if double_buffered else None   # + reference-side preparation ahead of time
            if double_buffered:
                sharded.prefetch_batch(points[0:3], prepare=prep)
            for pi, (x, y, z) in enumerate(points):
                out = torch.empty(xs * ys * zl, dtype=torch.float32)
                if pipelined and pi + 1 < len(points):
                    sharded.prefetch(points[pi + 1])
                if batched and pi % 3 == 0:
                    sharded.prefetch_batch(points[pi:pi + 3])
                if double_buffered and pi % 3 == 0 and pi + 3 < len(points):
                    sharded.prefetch_batch(points[pi + 3:pi + 6], prepare=prep)
                sharded.compute(measure, out, (x, y, z), k=2)
                gathered = [torch.empty(xs * ys * slab_bounds(zs, world, r)[1], dtype=torch.float32)
                            for r in range(world)] if rank == 0 else None
                if world > 1:
                    # slabs may differ in size: gather through point-to-point sends
                    if rank == 0:
                        gathered[0] = out
                        for r in range(1, world):
                            dist.recv(gathered[r], src=r)
                    else:
                        dist.send(out, dst=0)
                else:
                    gathered = [out]
                if rank == 0:
                    whole = torch.cat(gathered).numpy()
                    kw = dict(k=2, minmax_ref=gmm) if measure != Measure.PEARSON else {}
                    want = oracle.field(int(measure), ens, ens[:, z, y, x].copy(), **kw)
                    same = (whole.view(np.uint32) == want.view(np.uint32)) | (np.isnan(whole) & np.isnan(want))
                    if not same.all():
                        failures.append((measure.name, (x, y, z), int((~same).sum())))
        # a third outstanding batch would overwrite rows that are still pending: refused (on every rank alike)
        sharded.prefetch_batch(points[0:1])
        sharded.prefetch_batch(points[1:2])
        try:
            sharded.prefetch_batch(points[2:3])
            failures.append("third outstanding batch was accepted")
        except RuntimeError:
            pass
        for p in points[0:2]:
            sharded.compute(Measure.PEARSON, torch.empty(xs * ys * zl, dtype=torch.float32), p)
        if rank == 0:
            Path(result_file).write_text("OK" if not failures else repr(failures))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,zs", [(2, 8), (3, 8), (1, 4)])
def test_sharded_equals_unsharded(tmp_path, world, zs):
    result = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, _free_port(), zs, str(result)), nprocs=world, join=True)
    assert result.read_text() == "OK"

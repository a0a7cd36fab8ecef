"""Multi-GPU correlation field: z-slab sharding, one process per GPU, torch.distributed (RCCL over xGMI on ROCm).

The reference has no distributed path (zero collective call sites); this is new, MI355X-native
work.  Voxels are independent given the cs-float reference vector, so the grid shards by z-slab -- with x-fastest
volumes (IDXS, src/Loaders/DataSet.hpp:37) a slab of every member is one contiguous range and output slabs concatenate
-- and ONE exchange step remains per evaluation: the rank that owns the reference point's slice gathers
referenceValues[c] = member_c[IDXS(ref)] (CorrelationCalculator.cpp:802,815-817) on its device and broadcasts those
cs floats (<= 1 KiB, latency-bound).  The binned-MI measures additionally need the global extrema of the members
(CorrelationCalculator.cpp:822-829): an all-reduce of two floats, once per data set.

Everything is stream-ordered on the caller's stream: no host synchronisation inside `compute`.
The compute backend is any object with the CorrField device interface (set_grid / gather_reference_device /
compute_device / member_minmax); the product backend is correrender_amd.CorrField (HIP kernels).
"""
from __future__ import annotations

from typing import Optional, Tuple

from .engine import Measure


def slab_bounds(zs: int, world: int, rank: int) -> Tuple[int, int]:
    """(z_begin, z_count) of `rank`'s slab: ceil-split, the first zs % world ranks get one extra slice."""
    base, rem = divmod(zs, world)
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def slab_owner(zs: int, world: int, z: int) -> Tuple[int, int]:
    """(owner rank, local z) of global slice z."""
    if not 0 <= z < zs:
        raise ValueError(f"z={z} outside [0,{zs})")
    base, rem = divmod(zs, world)
    split = rem * (base + 1)
    if z < split:
        r = z // (base + 1)
    else:  # z >= split implies base >= 1 (base == 0 means split == zs > z)
        r = rem + (z - split) // base
    z0, _ = slab_bounds(zs, world, r)
    return r, z - z0


_BINNED = (Measure.MUTUAL_INFORMATION_BINNED, Measure.BINNED_MI_CORRELATION_COEFFICIENT)


class ShardedCorrField:
    """One rank's share of a z-slab-sharded correlation field evaluation."""

    def __init__(self, engine, grid: Tuple[int, int, int], cs: int, *, rank: Optional[int] = None,
                 world: Optional[int] = None, group=None, device=None, always_exchange: bool = False):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.engine = engine
        self.group = group
        if world is None:
            world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if world > 1 else 0
        self.rank, self.world = rank, world
        # always_exchange: run the gather + collective path even with one rank (a 1-rank RCCL group): lets the exact
        # code path of the multi-GPU runs be exercised on a single GPU
        self._collectives = world > 1 or (always_exchange and dist.is_available() and dist.is_initialized())
        self.xs, self.ys, self.zs = grid
        self.cs = cs
        # world and zs are the same on every rank, so this raises on ALL ranks together -- before any collective, never
        # on the empty ranks alone while the others go on to wait in an all-reduce
        if world > self.zs:
            raise ValueError(f"{world} ranks cannot share a grid of {self.zs} z-slices: every rank needs a slice")
        self.z_begin, self.z_count = slab_bounds(self.zs, world, rank)
        self.device = device if device is not None else torch.device("cpu")
        engine.set_grid(self.xs, self.ys, self.z_count, cs)
        # Reference-vector buffers, used round-robin.  prefetch() fills the next one on a side (communication) stream
        # while the kernel of the current step still reads the previous one, so gather + broadcast of step i+1 overlap
        # the evaluation of step i (independent reference points, e.g. successive mouse positions).
        self._nbuf = 3
        self._ref = [torch.empty(cs, dtype=torch.float32, device=self.device) for _ in range(self._nbuf)]
        self._slot = 0
        self._pending = []  # [(ref_xyz, slot)] prefetched, not yet consumed
        self._batch = [None, None]       # [R, cs] row buffers of prefetch_batch(), double-buffered
        self._batch_flip = 0
        self._slots_per_batch = 32       # prepared-table slots: 2 batches x 32 rows = CRF_PREPARED_SLOTS
        self._batch_ready = [None, None]
        self._batch_done = [None, None]
        self._cuda = self.device.type == "cuda"
        self._stage_through_host = bool(self._cuda and self._collectives and dist.get_backend(group) == "gloo")
        if self._stage_through_host:
            self._host_ref = torch.empty(cs, dtype=torch.float32)
        if self._cuda:
            self._comm_stream = torch.cuda.Stream(device=self.device)
            self._ready = [torch.cuda.Event() for _ in range(self._nbuf)]   # vector landed in buffer[slot]
            self._batch_ready = [torch.cuda.Event(), torch.cuda.Event()]
            self._done = [None] * self._nbuf                                # last kernel that read buffer[slot]
        self._minmax = None

    @property
    def local_voxels(self) -> int:
        return self.xs * self.ys * self.z_count

    def _global_rank(self, r: int) -> int:
        return self._dist.get_global_rank(self.group, r) if (self.group is not None and self.world > 1) else r

    def bind_members(self, members):
        """members: this rank's slab of every member ([cs, z_count, ys, xs] or a list of cs tensors)."""
        self.engine.bind_members(members)
        self._minmax = None

    def global_minmax(self) -> Tuple[float, float]:
        """min of mins / max of maxes over all members and all slabs (binned MI normalisation range)."""
        if self._minmax is None:
            mn, mx = self.engine.member_minmax()
            if self._collectives:
                t = self._torch.tensor([mn, -mx], dtype=self._torch.float32, device=self.device)
                self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN, group=self.group)
                mn, mx = float(t[0]), float(-t[1])
            self._minmax = (mn, mx)
        return self._minmax

    def _exchange(self, ref_xyz, buf, stream_ptr: int):
        """owner: gather the cs reference values on the device; everyone: receive them (the one exchange step)."""
        x, y, z = ref_xyz
        owner, local_z = slab_owner(self.zs, self.world, z)
        if self.rank == owner:
            self.engine.gather_reference_device(x, y, local_z, buf, stream_ptr)
        if self._collectives:
            if self._stage_through_host:
                # rehearsal mode (gloo process group with GPU tensors, e.g. several ranks sharing one GPU): stage the
                # cs floats through a host tensor explicitly instead of relying on gloo's own CUDA staging
                if self.rank == owner:
                    self._comm_stream.synchronize()
                    self._host_ref.copy_(buf)
                self._dist.broadcast(self._host_ref, src=self._global_rank(owner), group=self.group)
                if self.rank != owner:
                    buf.copy_(self._host_ref)
            else:
                self._dist.broadcast(buf, src=self._global_rank(owner), group=self.group)

    def prefetch(self, ref_xyz):
        """Starts gather + broadcast of the reference vector of a FUTURE compute(ref_xyz) on the communication stream.
        Every rank must call prefetch/compute with the same sequence of reference points (collective semantics)."""
        if len(self._pending) >= self._nbuf - 1:
            raise RuntimeError("too many outstanding prefetches")
        slot = self._slot
        self._slot = (self._slot + 1) % self._nbuf
        buf = self._ref[slot]
        if self._cuda:
            torch = self._torch
            with torch.cuda.stream(self._comm_stream):
                if self._done[slot] is not None:
                    self._comm_stream.wait_event(self._done[slot])  # the kernel that last read this buffer
                self._exchange(ref_xyz, buf, self._comm_stream.cuda_stream)
                self._ready[slot].record(self._comm_stream)
        else:
            self._exchange(ref_xyz, buf, 0)
        self._pending.append((tuple(ref_xyz), slot))

    def prefetch_batch(self, points, prepare=None):
        """Exchanges the reference vectors of SEVERAL upcoming compute() calls in one collective: every rank gathers, on
        its device, the vectors of the points whose slice it owns into the rows of a zeroed [R, cs] buffer, and one
        all-reduce(SUM) of R*cs floats gives every rank every row (a sum of one value and zeros is that value; the only
        bit pattern not preserved is -0.0, which no estimator distinguishes from +0.0).  One collective per R
        evaluations instead of one per evaluation: at 8 GPUs a 256^3 x 64 evaluation is ~0.1 ms per rank, the same order
        as the launch latency of a collective.  Rows are consumed, in order, by the following compute() calls; a second
        batch may be prefetched while the first is being consumed (double buffering).
        prepare=(measure, kwargs): also run the estimator's reference-side preparation of every row on the communication
        stream (crf_prepare_device), so that the compute() calls -- which must then use the same measure and
        parameters -- launch only the per-voxel kernel on the critical path."""
        points = [tuple(p) for p in points]
        torch = self._torch
        r = len(points)
        b = self._batch_flip
        # two row buffers: one batch may be exchanged while the previous one is still being consumed (the exchange of
        # batch i+1 then overlaps the kernels of batch i); a third would overwrite rows that are still pending
        if any(not isinstance(slot, tuple) or slot[1] == b for _, slot in self._pending):
            raise RuntimeError("prefetch_batch(): at most one batch may be outstanding besides the one being consumed")
        self._batch_flip ^= 1
        if self._batch[b] is None or self._batch[b].shape[0] < r:
            self._batch[b] = torch.empty((max(r, 8), self.cs), dtype=torch.float32, device=self.device)
        rows = self._batch[b]
        owners = [slab_owner(self.zs, self.world, p[2]) for p in points]

        def fill(stream_ptr):
            # one launch: owned rows gathered, the others zeroed
            local = [(x, y, local_z) if owner == self.rank else None
                     for (x, y, _), (owner, local_z) in zip(points, owners)]
            self.engine.gather_reference_rows_device(local, rows, stream_ptr)
            if self._collectives:
                if self._stage_through_host:
                    self._comm_stream.synchronize()
                    host = rows[:r].cpu()
                    self._dist.all_reduce(host, op=self._dist.ReduceOp.SUM, group=self.group)
                    rows[:r].copy_(host)
                else:
                    self._dist.all_reduce(rows[:r], op=self._dist.ReduceOp.SUM, group=self.group)

        slots = None
        if prepare is not None:
            if r > self._slots_per_batch:
                raise ValueError(f"at most {self._slots_per_batch} rows per prepared batch")
            measure, pkw = prepare
            if int(measure) in _BINNED and "minmax_ref" not in pkw:
                mm = self.global_minmax()
                pkw = dict(pkw, minmax_ref=mm, minmax_query=mm)
            slots = [b * self._slots_per_batch + i for i in range(r)]

        def prepare_rows(stream_ptr):
            if hasattr(self.engine, "prepare_rows_device"):     # one library call for the whole batch
                self.engine.prepare_rows_device(measure, rows, slots[0], r, stream=stream_ptr, **pkw)
                return
            for i in range(r):
                self.engine.prepare_device(measure, slots[i], device_reference=rows[i], stream=stream_ptr, **pkw)

        if self._cuda:
            with torch.cuda.stream(self._comm_stream):
                if self._batch_done[b] is not None:
                    self._comm_stream.wait_event(self._batch_done[b])  # last kernel that read this batch buffer
                fill(self._comm_stream.cuda_stream)
                if slots is not None:
                    prepare_rows(self._comm_stream.cuda_stream)
                self._batch_ready[b].record(self._comm_stream)
        else:
            fill(0)
            if slots is not None:
                prepare_rows(0)
        for i, p in enumerate(points):
            self._pending.append((p, ("batch", b, i, i == r - 1, None if slots is None else (slots[i], int(measure)))))

    def compute_batch(self, measure, outs, points, **kw):
        """compute() for the next len(points) reference points of ONE prepared batch (prefetch_batch(points, prepare=...)),
        outs[i] receiving point i.  When the engine can launch prepared evaluations in one call
        (crf_compute_prepared_device) the host pays one call for the whole batch instead of one per evaluation; any other
        state (rows not prepared, a batch boundary inside, an engine without that call) takes the per-point path."""
        points = [tuple(p) for p in points]
        n = len(points)
        head = self._pending[:n]
        fast = (n > 0 and len(head) == n and hasattr(self.engine, "compute_prepared_device") and
                all(isinstance(slot, tuple) and slot[4] is not None and slot[4][1] == int(measure) and p == pt
                    for (p, slot), pt in zip(head, points)) and
                len({slot[1] for _, slot in head}) == 1 and
                [slot[4][0] for _, slot in head] == list(range(head[0][1][4][0], head[0][1][4][0] + n)))
        if not fast:
            for out, p in zip(outs, points):
                self.compute(measure, out, p, **kw)
            return outs
        if int(measure) in _BINNED and "minmax_ref" not in kw:
            mm = self.global_minmax()
            kw = dict(kw, minmax_ref=mm, minmax_query=mm)
        del self._pending[:n]
        b = head[0][1][1]
        cur = self._torch.cuda.current_stream(self.device) if self._cuda else None
        if cur is not None and head[0][1][2] == 0:
            cur.wait_event(self._batch_ready[b])
        self.engine.compute_prepared_device(measure, list(outs), head[0][1][4][0],
                                            stream=cur.cuda_stream if cur is not None else 0, **kw)
        if cur is not None and head[-1][1][3]:
            ev = self._torch.cuda.Event()
            ev.record(cur)
            self._batch_done[b] = ev
        return outs

    def compute(self, measure, out, ref_xyz, **kw):
        """Evaluates this rank's slab for the GLOBAL reference point ref_xyz into `out` (z_count*ys*xs floats), on the
        current torch stream, stream-ordered, without host synchronisation."""
        if int(measure) in _BINNED and "minmax_ref" not in kw:
            mm = self.global_minmax()
            kw = dict(kw, minmax_ref=mm, minmax_query=mm)
        if not self._collectives and not self._pending:
            # single GPU: no exchange; the gather is fused into the estimator's preparation kernel
            stream_ptr = self._torch.cuda.current_stream(self.device).cuda_stream if self._cuda else 0
            self.engine.compute_device(measure, out, tuple(ref_xyz), stream=stream_ptr, **kw)
            return out
        if not self._pending or self._pending[0][0] != tuple(ref_xyz):
            if self._pending:
                raise RuntimeError("compute() must consume prefetched reference points in order")
            self.prefetch(ref_xyz)
        _, slot = self._pending.pop(0)
        stream_ptr = 0
        cur = None
        if self._cuda:
            cur = self._torch.cuda.current_stream(self.device)
            stream_ptr = cur.cuda_stream
        if isinstance(slot, tuple):  # a row of a batched exchange
            _, b, row, last, prepared = slot
            if cur is not None and row == 0:
                cur.wait_event(self._batch_ready[b])
            if prepared is not None:
                if prepared[1] != int(measure):
                    raise RuntimeError("compute(): the batch was prepared for a different measure")
                self.engine.compute_device(measure, out, prepared_slot=prepared[0], stream=stream_ptr, **kw)
            else:
                self.engine.compute_device(measure, out, device_reference=self._batch[b][row], stream=stream_ptr, **kw)
            if cur is not None and last:
                ev = self._torch.cuda.Event()
                ev.record(cur)
                self._batch_done[b] = ev
            return out
        buf = self._ref[slot]
        if cur is not None:
            cur.wait_event(self._ready[slot])
        self.engine.compute_device(measure, out, device_reference=buf, stream=stream_ptr, **kw)
        if cur is not None:
            ev = self._torch.cuda.Event()
            ev.record(cur)
            self._done[slot] = ev
        return out

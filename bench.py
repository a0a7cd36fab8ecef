#!/usr/bin/env python3
"""bench.py -- headline benchmark: Pearson correlation field, 256^3 grid x 64 ensemble members (BASELINE.json
configs[1]), reported as whole-job Mvoxel-corr/s with the per-voxel kernel priced against the HBM roofline and the
CPU calculator timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full-grid evaluation for one reference point (what one moved reference point costs the renderer,
SURVEY.md section 3(C)).  Reference points change every step.  Members are resident in HBM when the timed region
starts (synthetic box ensemble generated on the device); the result stays in HBM (the D2H of 4 bytes/voxel is
reported separately in DESIGN.md, never in `value`).

N > 1: the 256^3 grid is sharded by z-slab, one process per GPU; the rank that owns the reference point's slice
gathers the cs reference values on its device and broadcasts them (RCCL over xGMI) -- the only exchange on the
path.  Total work is fixed, so scaling is "strong".
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, nargs=3, default=[256, 256, 256], metavar=("XS", "YS", "ZS"))
    ap.add_argument("--members", type=int, default=64)
    ap.add_argument("--measure", default="pearson")
    ap.add_argument("--seed", type=int, default=20260130)
    ap.add_argument("--spinup-ms", type=float, default=400.0,
                    help="untimed steady-state spin-up before the W warm-up steps (lets the GPU reach its sustained "
                         "clocks: a cold MI355X ran this kernel 7 %% slower for its first ~100 ms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-slices", type=int, default=0, help="z-slices of the CPU-baseline sample (0 = auto)")
    return ap.parse_args()


def reference_points(xs, ys, zs, count):
    """Deterministic moving reference point: grid centre first (the reference's default,
    CorrelationCalculator.cpp:104-110), a point inside the first big box, then a low-discrepancy walk."""
    pts = [(xs // 2, ys // 2, zs // 2), (xs // 8, ys // 8, zs // 2)]
    i = 0
    while len(pts) < count:
        i += 1
        pts.append((int((i * 0.754877666) % 1.0 * xs), int((i * 0.569840291) % 1.0 * ys),
                    int((i * 0.362437104) % 1.0 * zs)))
    return pts[:count]


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import correrender_amd as ca

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world
    # one rank per GPU; CRF_BENCH_BACKEND=gloo lets several ranks share one GPU for rehearsals on a 1-GPU box
    backend = os.environ.get("CRF_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    force_exchange = os.environ.get("CRF_FORCE_EXCHANGE", "0") == "1"  # 1-rank process group: exercise the N>1 path
    if world > 1 or force_exchange:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force_exchange and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    xs, ys, zs = args.grid
    cs = args.members
    measure = ca.Measure(ca.MEASURE_IDS.index(args.measure))
    from correrender_amd.distributed import ShardedCorrField
    eng = ca.CorrField(local_rank)
    sharded = ShardedCorrField(eng, (xs, ys, zs), cs, rank=rank, world=world, device=torch.device("cuda", local_rank),
                               always_exchange=force_exchange)
    multi = world > 1 or force_exchange
    z0, zl = sharded.z_begin, sharded.z_count
    n_local = xs * ys * zl
    n_total = xs * ys * zs
    members = torch.empty((cs, zl, ys, xs), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for c in range(cs):
        eng.synth_box_member(members[c], xs, ys, zl, z0, zs, c, cs, args.seed, stream)
    torch.cuda.synchronize()
    sharded.bind_members(members)
    out = torch.empty(n_local, dtype=torch.float32, device="cuda")
    kwargs = {}
    if measure in (ca.Measure.MUTUAL_INFORMATION_BINNED, ca.Measure.BINNED_MI_CORRELATION_COEFFICIENT):
        mm = sharded.global_minmax()
        kwargs = dict(minmax_ref=mm, minmax_query=mm)
    if measure in (ca.Measure.MUTUAL_INFORMATION_KRASKOV, ca.Measure.KMI_CORRELATION_COEFFICIENT):
        kwargs = dict(k=3)  # BASELINE.json configs[2]

    pts = reference_points(xs, ys, zs, args.warmup + args.steps)

    # one GPU: the same batched pipeline without any collective (reference vectors gathered and the reference-side
    # tables prepared a batch ahead on the side stream) when CRF_BENCH_LOCAL_BATCH=1; default: plain per-step calls
    # (measured at 256^3 x 64: 0.700 vs 0.696 ms/step -- at one GPU the preparation is 0.7 % of a step, no gain)
    local_batches = not multi and os.environ.get("CRF_BENCH_LOCAL_BATCH", "0") == "1"
    LOOKAHEAD = int(os.environ.get("CRF_BENCH_LOOKAHEAD", "16"))  # reference vectors exchanged per collective (N > 1), <= 32

    def run(lo, hi):
        """Steps lo..hi-1.  N > 1: the reference vectors of the next LOOKAHEAD requested points are exchanged in ONE
        collective (owners gather on their device -> RCCL all-reduce of LOOKAHEAD*cs floats) on the communication stream,
        one batch ahead of the kernels that consume them, so the exchange overlaps the evaluation of earlier steps."""
        if not multi and not local_batches:
            for i in range(lo, hi):
                sharded.compute(measure, out, pts[i], **kwargs)
            return
        # the reference-side preparation of every row also runs on the communication stream (crf_prepare_device):
        # only the per-voxel kernels remain on the critical path.  CRF_BENCH_PREPARE=0 keeps it inline.
        prep = (measure, kwargs) if os.environ.get("CRF_BENCH_PREPARE", "1") != "0" else None
        starts = list(range(lo, hi, LOOKAHEAD))
        if starts:
            sharded.prefetch_batch(pts[starts[0]:min(starts[0] + LOOKAHEAD, hi)], prepare=prep)
        for n, b0 in enumerate(starts):
            b1 = min(b0 + LOOKAHEAD, hi)
            if n + 1 < len(starts):  # exchange of the NEXT batch first: it overlaps the kernels of this one
                sharded.prefetch_batch(pts[starts[n + 1]:min(starts[n + 1] + LOOKAHEAD, hi)], prepare=prep)
            for i in range(b0, b1):
                sharded.compute(measure, out, pts[i], **kwargs)

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    # untimed spin-up (clock ramp), then the W untimed warm-up steps, then exactly K timed steps
    # (the number of spin-up rounds is decided by rank 0 and broadcast: every rank must issue the same collectives)
    t_spin = time.perf_counter()
    run(0, max(args.warmup, 1))
    torch.cuda.synchronize()
    t_round = max(time.perf_counter() - t_spin, 1e-4)
    rounds = torch.tensor([int(args.spinup_ms * 1e-3 / t_round) if args.spinup_ms > 0 else 0], device="cuda")
    if multi:
        dist.broadcast(rounds, src=0)
    for _ in range(int(rounds[0])):
        run(0, max(args.warmup, 1))
    fence()
    run(0, args.warmup)
    fence()
    eng.set_profiling(True)
    eng.take_kernel_time()
    t0 = time.perf_counter()
    run(args.warmup, args.warmup + args.steps)
    t_enqueued = time.perf_counter() - t0  # host time to issue the K steps (diagnostic: host-bound if ~ elapsed)
    fence()
    elapsed = time.perf_counter() - t0
    eng.set_profiling(False)
    kernel_ms_sum, launches = eng.take_kernel_time()
    kernel_name = eng.last_kernel_name()
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        tk = torch.tensor([kernel_ms_sum / max(launches, 1)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tk, op=dist.ReduceOp.MAX)
        kernel_ms = float(tk[0])
    else:
        kernel_ms = kernel_ms_sum / max(launches, 1)

    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed / 1e6

    # roofline of the dominant kernel: algorithmic bytes per launch = voxels per launch x (4*cs + 4)   
    bytes_per_launch = n_local * (4 * cs + 4)
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": load_pmc_traffic(args, world),
                "kernel": kernel_name, "kernel_ms": round(kernel_ms, 4), "bytes_per_launch": bytes_per_launch,
                "launches_timed": launches}

    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, parity = cpu_baseline(args, eng, members, out, measure, pts[args.warmup + args.steps - 1], kwargs)

    if rank == 0:
        line = {
            "metric": "Mvoxel-corr/s", "value": round(value, 1), "unit": "Mvoxel-corr/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "spinup_ms": args.spinup_ms,
            "ms_per_step": round(ms_per_step, 4), "host_issue_ms_per_step": round(t_enqueued / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.measure} correlation field, {xs}x{ys}x{zs} grid x {cs} ensemble members "
                                   "(synthetic box ensemble), one moving reference point per step",
                       "grid": [xs, ys, zs], "members": cs, "measure": args.measure,
                       "sharding": f"z-slab x{world}" + (f", reference vectors exchanged over {'RCCL' if backend == 'nccl' else backend + ' (rehearsal)'} ({LOOKAHEAD} requested points per collective)" if world > 1 else ""),
                       "resident": "members and result in HBM"},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
        }
        print(json.dumps(line), flush=True)
    eng.close()
    if multi:
        dist.destroy_process_group()


def load_pmc_traffic(args, world):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
    written by profiles/collect_pmc.sh with the gfx950 corrections of MI355X_MICROARCH.md), when that file describes
    this exact workload; otherwise null."""
    p = ROOT / "profiles" / "pmc_traffic.json"
    try:
        d = json.loads(p.read_text())
        key = f"{args.measure}:{args.grid[0]}x{args.grid[1]}x{args.grid[2]}x{args.members}:gpus{world}"
        return d.get(key, {}).get("traffic_bytes_per_launch")
    except Exception:
        return None


def cpu_baseline(args, eng, members, out, measure, last_pt, kwargs):
    """Times the CPU calculator on the GPU box's host cores on a bounded z-sub-slab of the same workload, and
    checks the GPU result of the last step against it on that slab.  Uses the reference's own object code
    (oracle/_ref, kind "reference") when it was built, else this repo's restatement (kind "port")."""
    import numpy as np
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    import correrender_amd as ca

    xs, ys, zs = args.grid
    cs = args.members
    rate_guess = {"pearson": 15e6 * 64 / cs}.get(args.measure, 0.3e6 * 64 / cs)  # voxels/s, all cores
    slices = args.cpu_slices or max(1, min(zs, int(rate_guess * 4.0 / (xs * ys))))
    zc = min(zs - slices, max(0, last_pt[2] - slices // 2))
    sample = members[:, zc:zc + slices].contiguous().cpu().numpy()
    x, y, z = last_pt
    ref_values = members[:, z, y, x].cpu().numpy().copy()
    n = sample[0].size
    m = int(measure)
    use_ref = oracle_lib.reference_available() and m <= 2
    oracle = oracle_lib.load_oracle()
    if use_ref:
        ref = oracle_lib.load_reference()
        run = lambda: ref.field(m, sample, ref_values)
    else:
        okw = {}
        if "minmax_ref" in kwargs:
            okw = dict(minmax_ref=kwargs["minmax_ref"], minmax_query=kwargs["minmax_query"])
        if "k" in kwargs:
            okw["k"] = kwargs["k"]
        run = lambda: oracle.field(m, sample, ref_values, **okw)
    run()  # warm-up (page-in, thread team)
    times = []
    t_all = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_all < 10.0 and len(times) < 15):
        t0 = time.perf_counter()
        want = run()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    got = out.view(zs, ys, xs)[zc:zc + slices].cpu().numpy().reshape(-1)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    with np.errstate(invalid="ignore"):
        max_abs = float(np.nanmax(np.abs(got.astype(np.float64) - want.astype(np.float64)))) if n else 0.0
    cpu = {"value": round(n / med / 1e6, 2), "unit": "Mvoxel-corr/s", "cores": oracle.max_threads(),
           "kind": "reference" if use_ref else "port",
           "sample": f"z-slices [{zc},{zc + slices}) of the same {xs}x{ys}x{zs}x{cs} volume ({n} voxels), "
                     f"median of {len(times)} runs, OpenMP over voxels"}
    parity = {"checked_voxels": int(n), "bit_identical": int(same.sum()), "max_abs_err": max_abs,
              "against": cpu["kind"]}
    return cpu, parity


if __name__ == "__main__":
    main()

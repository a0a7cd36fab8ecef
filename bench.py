#!/usr/bin/env python3
"""bench.py -- headline benchmark: Pearson correlation field, 256^3 grid x 64 ensemble members (BASELINE.json
configs[1]), reported as whole-job Mvoxel-corr/s with the per-voxel kernel priced against the HBM roofline and the
CPU calculator timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 launches its own ranks: the parent process (which never touches a GPU) starts
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...
as a child and relays rank 0's JSON line; the same script run under torch.distributed.run (WORLD_SIZE set) is a rank.

A "step" = one full-grid evaluation for one reference point (what one moved reference point costs the renderer,
SURVEY.md section 3(C)).  Reference points change every step.  Members are resident in HBM when the timed region
starts (synthetic box ensemble generated on the device); the result stays in HBM (the host-boundary cost of
crf_compute -- D2H of 4 bytes/voxel -- is reported separately as `host_boundary`, never in `value`).

N > 1: the 256^3 grid is sharded by z-slab, one process per GPU; the rank that owns the reference point's slice
gathers the cs reference values on its device and they are exchanged over RCCL/xGMI -- the only exchange on the path.
Total work is fixed, so scaling is "strong".  Two modes are measured and both reported:
  throughput  the reference vectors of the next LOOKAHEAD (16) requested points are exchanged in one collective, a batch
              ahead of the kernels that consume them (bulk evaluation of known points)
  latency     lookahead 0: every step is exchange -> reference-side preparation -> per-voxel kernel, in that order on the
              device (an interactively moved reference point: the next point is not known in advance)
`value` is the throughput mode; `latency` carries the other.  The K timed steps are repeated --repeats times (each
block fenced by barrier + synchronize on both sides, MAX over ranks); `value`/`ms_per_step` are the median block.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
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
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps (median reported)")
    ap.add_argument("--grid", type=int, nargs=3, default=[256, 256, 256], metavar=("XS", "YS", "ZS"))
    ap.add_argument("--members", type=int, default=64)
    ap.add_argument("--measure", default="pearson")
    ap.add_argument("--seed", type=int, default=20260130)
    ap.add_argument("--spinup-ms", type=float, default=400.0,
                    help="untimed steady-state spin-up before the W warm-up steps (lets the GPU reach its sustained "
                         "clocks: a cold MI355X ran this kernel 7 %% slower for its first ~100 ms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-boundary", action="store_true")
    ap.add_argument("--cpu-slices", type=int, default=0, help="z-slices of the CPU-baseline sample (0 = auto)")
    ap.add_argument("--cpu-baseline-child", default="", help=argparse.SUPPRESS)  # internal: see cpu_baseline_bound()
    return ap.parse_args()


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """Parent of an N > 1 run: starts the ranks as a child process tree and relays their output.  Nothing here touches
    the GPU (no HIP call, no torch.cuda call), and nothing is exec'ed over this process."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: must be in the ranks' environment before HIP starts
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(ROOT / "bench.py")] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line:
        sys.stdout.write(line)
        sys.stdout.flush()
    if rc != 0 or not line:
        raise SystemExit(rc or 1)


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


def summarize(ms_blocks):
    s = sorted(ms_blocks)
    return {"median": round(s[len(s) // 2], 4), "min": round(s[0], 4), "max": round(s[-1], 4), "blocks": len(s)}


def main():
    args = parse()
    if args.cpu_baseline_child:  # a child of cpu_baseline_bound(): CPU only, never touches the GPU
        return cpu_baseline_child(args.cpu_baseline_child)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before the first HIP call of this process
    import torch
    import torch.distributed as dist
    import correrender_amd as ca

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # one rank per GPU.  Fewer GPUs than ranks (a 1-GPU box): the ranks share the cards and exchange over gloo -- a
    # REHEARSAL of the N > 1 code path, flagged as such in the output (RCCL refuses two ranks on one device).
    n_dev = max(torch.cuda.device_count(), 1)
    rehearsal = world > n_dev
    backend = os.environ.get("CRF_BENCH_BACKEND", "gloo" if rehearsal else "nccl")
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    force_exchange = os.environ.get("CRF_FORCE_EXCHANGE", "0") == "1"  # 1-rank process group: exercise the N>1 path
    if world > 1 or force_exchange:
        if force_exchange and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
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
    ranks_seen = 1
    if multi:
        one = torch.ones(1, dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(one)
        ranks_seen = int(one[0])
        assert ranks_seen == world, f"{ranks_seen} ranks answered the all-reduce, {world} expected"
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

    def run(lo, hi, lookahead):
        """Steps lo..hi-1.  lookahead > 0 (N > 1): the reference vectors of the next `lookahead` requested points are
        exchanged in ONE collective (owners gather on their device -> all-reduce of lookahead*cs floats) on the
        communication stream, one batch ahead of the kernels that consume them.  lookahead == 0: each step exchanges its
        own reference vector (gather -> broadcast), prepares and evaluates, in that order."""
        if (not multi and not local_batches) or lookahead == 0:
            for i in range(lo, hi):
                sharded.compute(measure, out, pts[i], **kwargs)
            return
        # the reference-side preparation of every row also runs on the communication stream (crf_prepare_device):
        # only the per-voxel kernels remain on the critical path.  CRF_BENCH_PREPARE=0 keeps it inline.
        prep = (measure, kwargs) if os.environ.get("CRF_BENCH_PREPARE", "1") != "0" else None
        starts = list(range(lo, hi, lookahead))
        if starts:
            sharded.prefetch_batch(pts[starts[0]:min(starts[0] + lookahead, hi)], prepare=prep)
        for n, b0 in enumerate(starts):
            b1 = min(b0 + lookahead, hi)
            if n + 1 < len(starts):  # exchange of the NEXT batch first: it overlaps the kernels of this one
                sharded.prefetch_batch(pts[starts[n + 1]:min(starts[n + 1] + lookahead, hi)], prepare=prep)
            if prep is not None and os.environ.get("CRF_BENCH_BATCH_CALL", "1") != "0":
                # the whole prepared batch with one library call (crf_compute_prepared_device): at 8 GPUs an evaluation is
                # ~0.09 ms per rank, the same order as one Python-level call per step
                sharded.compute_batch(measure, [out] * (b1 - b0), pts[b0:b1], **kwargs)
            else:
                for i in range(b0, b1):
                    sharded.compute(measure, out, pts[i], **kwargs)

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_max(x):
        if not multi:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def timed_blocks(lookahead, repeats):
        """`repeats` blocks of exactly K timed steps, each bracketed by barrier + synchronize; wall time per block is
        the MAX over ranks.  Returns ([ms_per_step per block], host issue ms per step, kernel ms average, launches)."""
        blocks, issue = [], []
        kernel_ms_sum, launches = 0.0, 0
        eng.set_profiling(True)
        for _ in range(repeats):
            run(0, args.warmup, lookahead)
            fence()
            eng.take_kernel_time()  # drop the warm-up launches: only the K timed steps are priced
            t0 = time.perf_counter()
            run(args.warmup, args.warmup + args.steps, lookahead)
            issue.append(time.perf_counter() - t0)  # host time to issue the K steps (host-bound if ~ elapsed)
            fence()
            blocks.append(reduce_max(time.perf_counter() - t0) / args.steps * 1e3)
            ms, n = eng.take_kernel_time()
            kernel_ms_sum, launches = kernel_ms_sum + ms, launches + n
        eng.set_profiling(False)
        kernel_ms = reduce_max(kernel_ms_sum / max(launches, 1))
        return blocks, sorted(issue)[len(issue) // 2] / args.steps * 1e3, kernel_ms, launches

    # untimed spin-up (clock ramp); the number of rounds is decided by rank 0 and broadcast: every rank must issue the
    # same collectives
    t_spin = time.perf_counter()
    run(0, max(args.warmup, 1), LOOKAHEAD)
    torch.cuda.synchronize()
    t_round = max(time.perf_counter() - t_spin, 1e-4)
    rounds = torch.tensor([int(args.spinup_ms * 1e-3 / t_round) if args.spinup_ms > 0 else 0],
                          device="cuda" if backend == "nccl" or not multi else "cpu")
    if multi:
        dist.broadcast(rounds, src=0)
    for _ in range(int(rounds[0])):
        run(0, max(args.warmup, 1), LOOKAHEAD)
    fence()

    blocks, issue_ms, kernel_ms, launches = timed_blocks(LOOKAHEAD, max(args.repeats, 1))
    kernel_name = eng.last_kernel_name()
    stats = summarize(blocks)
    ms_per_step = stats["median"]
    value = n_total / (ms_per_step * 1e-3) / 1e6

    latency = None
    if multi:
        lblocks, lissue, lkernel, _ = timed_blocks(0, max(min(args.repeats, 3), 1))
        ls = summarize(lblocks)
        latency = {"lookahead": 0, "ms_per_step": ls["median"], "ms_per_step_min": ls["min"], "ms_per_step_max": ls["max"],
                   "value": round(n_total / (ls["median"] * 1e-3) / 1e6, 1), "unit": "Mvoxel-corr/s",
                   "host_issue_ms_per_step": round(lissue, 4), "kernel_ms": round(lkernel, 4),
                   "path": "per step: owner gathers cs floats -> broadcast -> reference-side preparation -> per-voxel kernel"}

    # roofline of the dominant kernel: algorithmic bytes per launch = voxels per launch x (4*cs + 4)   
    bytes_per_launch = n_local * (4 * cs + 4)
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    traffic, traffic_stale = load_pmc_traffic(args, world)
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_stale": traffic_stale,
                "kernel": kernel_name, "kernel_ms": round(kernel_ms, 4), "bytes_per_launch": bytes_per_launch,
                "launches_timed": launches}

    cpu = parity = host_boundary = None
    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:
            cpu, parity = cpu_baseline(args, eng, members, out, measure, pts[args.warmup + args.steps - 1], kwargs)
        if not args.no_host_boundary and not force_exchange:
            host_boundary = measure_host_boundary(args, eng, measure, pts, kwargs, n_total)

    if rank == 0:
        exchange = ("RCCL" if backend == "nccl" else backend + " (REHEARSAL: ranks share a GPU)") if multi else None
        line = {
            "metric": "Mvoxel-corr/s", "value": round(value, 1), "unit": "Mvoxel-corr/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "spinup_ms": args.spinup_ms,
            "ms_per_step": ms_per_step, "ms_per_step_min": stats["min"], "ms_per_step_max": stats["max"],
            "timed_blocks": stats["blocks"], "host_issue_ms_per_step": round(issue_ms, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            # the arithmetic the estimator runs in: the MI estimators accumulate in fp64 like the reference's
            # computeMutualInformation*<double>, the others in fp32 (ranks are integers)
            "dtype": "f64" if measure in (3, 4, 5, 6) else "f32", "data": "synthetic",
            "ranks_seen": ranks_seen, "backend": exchange,
            "config": {"workload": f"{args.measure} correlation field, {xs}x{ys}x{zs} grid x {cs} ensemble members "
                                   "(synthetic box ensemble), one moving reference point per step",
                       "grid": [xs, ys, zs], "members": cs, "measure": args.measure,
                       "sharding": f"z-slab x{world}" + (f", reference vectors exchanged over {exchange}" if multi else ""),
                       "resident": "members and result in HBM"},
            "throughput": ({"lookahead": LOOKAHEAD, "ms_per_step": ms_per_step, "value": round(value, 1),
                            "unit": "Mvoxel-corr/s",
                            "path": f"reference vectors of {LOOKAHEAD} requested points per collective, exchanged and "
                                    "prepared one batch ahead on the communication stream"} if multi else None),
            "latency": latency,
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "host_boundary": host_boundary,
        }
        print(json.dumps(line), flush=True)
    eng.close()
    if multi:
        dist.destroy_process_group()


KERNEL_SOURCES = {  # translation units (and what they include) that decide the dominant kernel's HBM traffic
    "pearson": ["kernels_pearson.hip", "crf_device.h", "crf_internal.h", "Makefile"],
    "spearman": ["kernels_rank.hip", "kernels_generic.hip", "crf_device.h", "crf_internal.h", "sortnet.inc", "Makefile"],
    "kendall": ["kernels_rank.hip", "kernels_generic.hip", "crf_device.h", "crf_internal.h", "sortnet.inc", "Makefile"],
    "mi_binned": ["kernels_binned.hip", "crf_device.h", "crf_mi_device.h", "crf_internal.h", "sortnet.inc", "Makefile"],
    "mi_kraskov": ["kernels_kraskov.hip", "crf_device.h", "crf_mi_device.h", "crf_internal.h", "Makefile"],
}


def kernel_source_sha256(measure):
    """sha256 over the kernel sources of `measure` (what a PMC traffic figure was collected for)."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES.get(measure, []):
        p = ROOT / "correrender_amd" / "csrc" / name
        h.update(name.encode())
        h.update(p.read_bytes() if p.exists() else b"<missing>")
    return h.hexdigest()


def load_pmc_traffic(args, world):
    """(HBM bytes per launch of the dominant kernel, stale flag) from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json, written by profiles/collect.sh + tools/profile_summary.py with the gfx950 corrections of
    MI355X_MICROARCH.md).  The figure is only reported when the file describes this exact workload AND was collected
    from the kernel sources as they are now (source_sha256); otherwise null, with traffic_stale = true when an entry
    exists but the kernel has changed since."""
    p = ROOT / "profiles" / "pmc_traffic.json"
    try:
        d = json.loads(p.read_text())
        key = f"{args.measure}:{args.grid[0]}x{args.grid[1]}x{args.grid[2]}x{args.members}:gpus{world}"
        e = d.get(key)
        if not e:
            return None, False
        if e.get("source_sha256") != kernel_source_sha256(args.measure):
            return None, True
        return e.get("traffic_bytes_per_launch"), False
    except Exception:
        return None, False


def host_cpu_description():
    """CPU model, physical cores and logical CPUs of the box (what /proc/cpuinfo says)."""
    model, phys, logical = "unknown", set(), 0
    try:
        pkg = core = None
        for ln in Path("/proc/cpuinfo").read_text().splitlines():
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                logical += 1
            elif k == "physical id":
                pkg = v
            elif k == "core id":
                core = v
            elif not k and pkg is not None:
                phys.add((pkg, core))
                pkg = core = None
        if pkg is not None:
            phys.add((pkg, core))
    except OSError:
        pass
    return model, len(phys) or None, logical or os.cpu_count()


def cpu_baseline(args, eng, members, out, measure, last_pt, kwargs):
    """Times the CPU calculator on the GPU box's host cores on a bounded z-sub-slab of the same workload, and
    checks the GPU result of the last step against it on that slab.  Uses the reference's own object code
    (oracle/_ref, kind "reference") when it was built, else this repo's restatement (kind "port")."""
    import numpy as np
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib

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
    model, phys, logical = host_cpu_description()
    threads = oracle.max_threads()
    cpu = {"value": round(n / med / 1e6, 2), "unit": "Mvoxel-corr/s", "cores": threads,
           "kind": "reference" if use_ref else "port",
           "cpu_model": model, "physical_cores": phys, "logical_cpus": logical, "threads_used": threads,
           "omp_proc_bind": os.environ.get("OMP_PROC_BIND", "unset"), "omp_places": os.environ.get("OMP_PLACES", "unset"),
           "sample": f"z-slices [{zc},{zc + slices}) of the same {xs}x{ys}x{zs}x{cs} volume ({n} voxels), "
                     f"median of {len(times)} runs, OpenMP over voxels (static schedule, {threads} threads)"}
    parity = {"checked_voxels": int(n), "bit_identical": int(same.sum()), "max_abs_err": max_abs,
              "against": cpu["kind"]}
    # the same sample once more with the threads bound to cores and the input pages placed by the threads that read them
    # (a two-socket host under-states the reference otherwise): a child process, because libgomp reads OMP_PROC_BIND /
    # OMP_PLACES when it is loaded
    try:
        cpu["bound"] = cpu_baseline_bound(sample, ref_values, m, use_ref, okw if not use_ref else {}, threads)
    except Exception as e:  # the bound variant is an extra: never fail the bench line over it
        cpu["bound"] = {"error": str(e)[:200]}
    cpu["first_touch"] = "one thread (the sample is copied from the GPU by the main thread)"
    return cpu, parity


def cpu_baseline_bound(sample, ref_values, m, use_ref, okw, threads):
    import numpy as np
    shm = Path("/dev/shm") if Path("/dev/shm").is_dir() else Path("/tmp")
    path = shm / f"crf_cpu_sample_{os.getpid()}.npz"
    try:
        np.savez(path, sample=sample, ref_values=ref_values, m=m, use_ref=use_ref,
                 okw=json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in okw.items()}))
        # the same thread count as the unbound run above (libgomp's default would be every logical CPU)
        env = dict(os.environ, OMP_PROC_BIND="spread", OMP_PLACES="cores", OMP_NUM_THREADS=str(threads))
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-child", str(path)], env=env,
                           capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            raise RuntimeError(f"child failed (rc {r.returncode}): {r.stderr[-300:]}")
        return json.loads(lines[-1])
    finally:
        path.unlink(missing_ok=True)


def cpu_baseline_child(path):
    """OMP_PROC_BIND / OMP_PLACES are set in this process's environment: first-touch the sample with the OpenMP threads
    (same static partition over voxels as the timed loop), then time the CPU calculator on it like the parent did."""
    import numpy as np
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    d = np.load(path, allow_pickle=False)
    m, use_ref = int(d["m"]), bool(d["use_ref"])
    okw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in json.loads(str(d["okw"])).items()}
    oracle = oracle_lib.load_oracle()
    sample = oracle.first_touch_copy(d["sample"])
    ref_values = np.array(d["ref_values"], np.float32)
    if use_ref:
        ref = oracle_lib.load_reference()
        run = lambda: ref.field(m, sample, ref_values)
    else:
        run = lambda: oracle.field(m, sample, ref_values, **okw)
    run()
    times = []
    t_all = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_all < 10.0 and len(times) < 15):
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    print(json.dumps({"value": round(sample[0].size / med / 1e6, 2), "unit": "Mvoxel-corr/s", "threads_used": oracle.max_threads(),
                      "omp_proc_bind": os.environ.get("OMP_PROC_BIND", "unset"), "omp_places": os.environ.get("OMP_PLACES", "unset"),
                      "first_touch": "by the OpenMP threads, static partition over voxels (oracle_first_touch_copy)",
                      "runs": len(times)}))
    return 0


def measure_host_boundary(args, eng, measure, pts, kwargs, n_total):
    """crf_compute at the reference's own boundary -- calculateCpu(t, e, float* buffer) writes a HOST buffer
    (Calculator.hpp:123-124; VolumeData.cpp:1222-1226 hands over a fresh `new float[]`): kernel + D2H of 4 bytes/voxel.
    `resident`: into a host buffer that has been written before; `fresh`: into a newly allocated, never-touched one."""
    import numpy as np
    xs, ys, zs = args.grid
    reps = 7
    resident = np.zeros((zs, ys, xs), dtype=np.float32)
    eng.compute(measure, pts[0], out=resident, **kwargs)  # page in, allocate the staging ring
    t_res, t_fresh = [], []
    for i in range(reps):
        t0 = time.perf_counter()
        eng.compute(measure, pts[i % len(pts)], out=resident, **kwargs)
        t_res.append(time.perf_counter() - t0)
    for i in range(reps):
        fresh = np.empty((zs, ys, xs), dtype=np.float32)  # 67 MB: mmap'ed by malloc, untouched pages
        t0 = time.perf_counter()
        eng.compute(measure, pts[i % len(pts)], out=fresh, **kwargs)
        t_fresh.append(time.perf_counter() - t0)
        del fresh
    med = lambda v: sorted(v)[len(v) // 2]
    # the floor of this boundary on this box: one plain DMA of the same bytes into pinned host memory
    import torch
    dev = torch.empty(n_total, dtype=torch.float32, device="cuda")
    pinned = torch.empty(n_total, dtype=torch.float32).pin_memory()
    t_dma = []
    for i in range(reps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pinned.copy_(dev, non_blocking=True)
        torch.cuda.synchronize()
        t_dma.append(time.perf_counter() - t0)
    t_dma = t_dma[2:]
    return {"entry": "crf_compute (host output buffer)", "bytes_d2h": n_total * 4,
            "pcie_d2h_pinned_floor_ms": round(med(t_dma) * 1e3, 3),
            "resident_ms": round(med(t_res) * 1e3, 3), "resident_value": round(n_total / med(t_res) / 1e6, 1),
            "fresh_ms": round(med(t_fresh) * 1e3, 3), "fresh_value": round(n_total / med(t_fresh) / 1e6, 1),
            "unit": "Mvoxel-corr/s", "runs": reps}


if __name__ == "__main__":
    main()

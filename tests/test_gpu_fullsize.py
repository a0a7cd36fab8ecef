"""GPU, BASELINE.json's full single-GPU size (256^3 grid x 64 members, 4.3 GB generated on the device): parity through
size-independent properties plus sampled comparison with the oracle.
  * sampled voxels (random + the reference voxel + box plateaus): bit-exact vs the oracle for Pearson/Spearman/Kendall,
    tolerance for the MI estimators -- the sampled member columns are gathered on the device and copied to the host;
  * slab consistency: evaluating a z-slab as its own grid gives bit-for-bit the corresponding part of the whole result
    (what the multi-GPU path relies on);
  * reference voxel: every measure returns what the oracle returns for a vector against itself."""
import numpy as np
import pytest
import torch

from correrender_amd import Measure
from parity import assert_bit_exact, assert_close
import oracle_lib

pytestmark = pytest.mark.gpu


def _need_free_hbm(gib, what):
    """BASELINE configs must not vanish silently: too little free HBM is a FAILURE, unless CRF_SHARED_GPU=1 says that the
    card is shared with other jobs (then the test is skipped, visibly)."""
    import os
    free, _ = torch.cuda.mem_get_info()
    if free >= gib * 2**30:
        return
    msg = f"{what} needs {gib} GB of free HBM, the card has {free / 2**30:.0f} GB free"
    if os.environ.get("CRF_SHARED_GPU") == "1":
        pytest.skip(msg + " (CRF_SHARED_GPU=1)")
    pytest.fail(msg + "; set CRF_SHARED_GPU=1 to skip on a shared card")


XS = YS = ZS = 256
CS = 64
SEED = 20260130


@pytest.fixture(scope="module")
def volume(engine):
    members = torch.empty((CS, ZS, YS, XS), dtype=torch.float32, device="cuda")
    for c in range(CS):
        engine.synth_box_member(members[c], XS, YS, ZS, 0, ZS, c, CS, SEED)
    torch.cuda.synchronize()
    yield members
    del members
    torch.cuda.empty_cache()


def _sample_indices():
    rng = np.random.default_rng(5)
    idx = rng.choice(XS * YS * ZS, size=60000, replace=False)
    ref = (ZS // 2 * YS + YS // 8) * XS + XS // 8                      # the reference voxel (inside the first big box)
    plateau = [((ZS // 2 + dz) * YS + YS // 8 + dy) * XS + XS // 8 + dx for dz in (-2, 0, 3) for dy in (-3, 1) for dx in (2, 5)]
    return np.unique(np.concatenate([idx, [ref], plateau])).astype(np.int64)


def test_full_size_sampled_parity_and_properties(engine, oracle, volume):
    engine.set_grid(XS, YS, ZS, CS)
    engine.bind_members(volume)
    ref_xyz = (XS // 8, YS // 8, ZS // 2)
    ref_values = engine.gather_reference(*ref_xyz)
    idx = _sample_indices()
    flat = volume.view(CS, -1)
    cols = flat[:, torch.from_numpy(idx).cuda()].cpu().numpy()                  # [cs, n_samples] member columns
    cols = np.ascontiguousarray(cols).reshape(CS, 1, 1, -1)
    np.testing.assert_array_equal(cols[:, 0, 0, np.searchsorted(idx, (ref_xyz[2] * YS + ref_xyz[1]) * XS + ref_xyz[0])],
                                  ref_values)
    out = torch.empty(XS * YS * ZS, dtype=torch.float32, device="cuda")
    results = {}
    for m, om, exact in ((Measure.PEARSON, oracle_lib.PEARSON, True), (Measure.SPEARMAN, oracle_lib.SPEARMAN, True),
                         (Measure.KENDALL, oracle_lib.KENDALL, True),
                         (Measure.MUTUAL_INFORMATION_BINNED, oracle_lib.MI_BINNED, False),
                         (Measure.MUTUAL_INFORMATION_KRASKOV, oracle_lib.MI_KRASKOV, False)):   # BASELINE configs[2]: k = 3
        kw = {}
        okw = {}
        if m == Measure.MUTUAL_INFORMATION_BINNED:
            mm = engine.member_minmax()
            kw = dict(num_bins=80, minmax_ref=mm, minmax_query=mm)
            okw = dict(num_bins=80, minmax_ref=mm)
        if m == Measure.MUTUAL_INFORMATION_KRASKOV:
            kw = okw = dict(k=3)
        engine.compute_device(m, out, ref_xyz, **kw)
        torch.cuda.synchronize()
        got = out[torch.from_numpy(idx).cuda()].cpu().numpy()
        want = oracle.field(om, cols, ref_values, **okw)
        (assert_bit_exact if exact else assert_close)(got, want, f"{m.name} 256^3x64 sampled")
        results[m] = out.clone()
    # slab consistency (Pearson and Kendall): slices [96, 128) evaluated as their own grid
    z0, zl = 96, 32
    engine.set_grid(XS, YS, zl, CS)
    engine.bind_members(volume[:, z0:z0 + zl])                                    # contiguous per member
    slab_out = torch.empty(XS * YS * zl, dtype=torch.float32, device="cuda")
    dref = torch.from_numpy(ref_values).cuda()
    for m in (Measure.PEARSON, Measure.KENDALL):
        engine.compute_device(m, slab_out, device_reference=dref)
        torch.cuda.synchronize()
        whole = results[m].view(ZS, YS * XS)[z0:z0 + zl].reshape(-1)
        assert torch.equal(slab_out.view(torch.int32), whole.view(torch.int32)), f"{m.name}: slab != whole"
    # the result has the expected structure: perfectly correlated plateau of the box that holds the reference point
    p = results[Measure.PEARSON].view(ZS, YS, XS)
    assert float(p[ZS // 2, YS // 8 + 2, XS // 8 + 3]) == pytest.approx(1.0, abs=1e-6)
    assert abs(float(p[ZS // 2, YS // 2, XS // 2])) < 0.6                        # an uncorrelated voxel


def test_config4_spearman_512cubed_128_members(engine, oracle):
    """BASELINE.json configs[3]: 512^3 x 128 members, Spearman, one GPU (68.7 GB generated on the device): sampled voxels
    bit-exact vs the oracle, and a z-slab evaluated on its own equals the corresponding part of the whole field."""
    xs = ys = zs = 512
    cs = 128
    _need_free_hbm(80, "BASELINE configs[3] (512^3 x 128 Spearman)")
    members = torch.empty((cs, zs, ys, xs), dtype=torch.float32, device="cuda")
    try:
        for c in range(cs):
            engine.synth_box_member(members[c], xs, ys, zs, 0, zs, c, cs, SEED)
        torch.cuda.synchronize()
        engine.set_grid(xs, ys, zs, cs)
        engine.bind_members(members)
        ref_xyz = (xs // 8, ys // 8, zs // 2)
        ref_values = engine.gather_reference(*ref_xyz)
        rng = np.random.default_rng(11)
        idx = np.unique(np.concatenate([rng.choice(xs * ys * zs, size=20000, replace=False),
                                        [(ref_xyz[2] * ys + ref_xyz[1]) * xs + ref_xyz[0]]])).astype(np.int64)
        didx = torch.from_numpy(idx).cuda()
        cols = np.ascontiguousarray(members.view(cs, -1)[:, didx].cpu().numpy()).reshape(cs, 1, 1, -1)
        out = torch.empty(xs * ys * zs, dtype=torch.float32, device="cuda")
        engine.compute_device(Measure.SPEARMAN, out, ref_xyz)
        torch.cuda.synchronize()
        assert_bit_exact(out[didx].cpu().numpy(), oracle.field(oracle_lib.SPEARMAN, cols, ref_values),
                         "Spearman 512^3x128 sampled")
        assert float(out[(ref_xyz[2] * ys + ref_xyz[1]) * xs + ref_xyz[0]]) == pytest.approx(1.0, abs=1e-6)
        z0, zl = 200, 16
        engine.set_grid(xs, ys, zl, cs)
        engine.bind_members(members[:, z0:z0 + zl])
        slab = torch.empty(xs * ys * zl, dtype=torch.float32, device="cuda")
        engine.compute_device(Measure.SPEARMAN, slab, device_reference=torch.from_numpy(ref_values).cuda())
        torch.cuda.synchronize()
        assert torch.equal(slab.view(torch.int32), out.view(zs, ys * xs)[z0:z0 + zl].reshape(-1).view(torch.int32))
    finally:
        del members
        torch.cuda.empty_cache()


def test_config5_pearson_slab_of_1024cubed_256_members(engine, oracle):
    """BASELINE.json configs[4]: 1024^3 x 256 members sharded over 8 GPUs -- this is ONE rank's share on one GPU: the
    z-slab [448, 576) of the 1024^3 grid (137 GB generated on the device with the global coordinates), Pearson against a
    reference vector handed in as the exchange would deliver it; sampled voxels bit-exact vs the oracle."""
    xs = ys = 1024
    zs_global, z0, zl = 1024, 448, 128
    cs = 256
    _need_free_hbm(150, "BASELINE configs[4] (one rank's slab of 1024^3 x 256 Pearson)")
    members = torch.empty((cs, zl, ys, xs), dtype=torch.float32, device="cuda")
    try:
        for c in range(cs):
            engine.synth_box_member(members[c], xs, ys, zl, z0, zs_global, c, cs, SEED)
        torch.cuda.synchronize()
        engine.set_grid(xs, ys, zl, cs)
        engine.bind_members(members)
        ref_local = (xs // 8, ys // 8, zs_global // 2 - z0)                      # the global centre slice lies in this slab
        ref_values = engine.gather_reference(*ref_local)
        rng = np.random.default_rng(12)
        n = xs * ys * zl
        idx = np.unique(np.concatenate([rng.choice(n, size=20000, replace=False),
                                        [(ref_local[2] * ys + ref_local[1]) * xs + ref_local[0]]])).astype(np.int64)
        didx = torch.from_numpy(idx).cuda()
        cols = np.ascontiguousarray(members.view(cs, -1)[:, didx].cpu().numpy()).reshape(cs, 1, 1, -1)
        out = torch.empty(n, dtype=torch.float32, device="cuda")
        engine.set_profiling(True)
        engine.take_kernel_time()
        engine.compute_device(Measure.PEARSON, out, device_reference=torch.from_numpy(ref_values).cuda())
        torch.cuda.synchronize()
        ms, launches = engine.take_kernel_time()
        engine.set_profiling(False)
        assert_bit_exact(out[didx].cpu().numpy(), oracle.field(oracle_lib.PEARSON, cols, ref_values),
                         "Pearson 1024x1024x128 slab x 256 sampled")
        print(f"\nconfig 5 slab: kernel {ms / max(launches, 1):.2f} ms, "
              f"{n * (4 * cs + 4) / (ms / max(launches, 1)) / 1e6:.0f} GB/s")
    finally:
        del members
        torch.cuda.empty_cache()


def test_whole_1024cubed_grid_on_one_gpu_members_of_4_gib(engine, oracle):
    """A member volume of 1024^3 floats is exactly 4 GiB -- beyond the kernels' 32-bit byte offsets; the reference has no
    such limit (a single MI355X holds 1024^3 x 64 members).  The library evaluates such grids in windows of 2^29 voxels:
    8 members of the whole 1024^3 grid (32 GiB), Pearson / Spearman / binned MI and the ensemble mean, sampled voxels of
    every window (incl. the window seams) vs the oracle, the reference point in the LAST window (a 64-bit voxel index)."""
    xs = ys = zs = 1024
    cs = 8
    _need_free_hbm(48, "the 1024^3 whole-grid test")
    n = xs * ys * zs
    members = torch.empty((cs, n), dtype=torch.float32, device="cuda")
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    try:
        for c in range(cs):
            engine.synth_box_member(members[c], xs, ys, zs, 0, zs, c, cs, SEED)
        torch.cuda.synchronize()
        engine.set_grid(xs, ys, zs, cs)
        engine.bind_members(members)
        ref_xyz = (xs // 2 + 7, ys // 2 + 3, zs - 100)                         # voxel index ~ 0.9 * 2^30: window 1
        ref_index = (ref_xyz[2] * ys + ref_xyz[1]) * xs + ref_xyz[0]
        ref_values = engine.gather_reference(*ref_xyz)
        np.testing.assert_array_equal(ref_values, members[:, ref_index].cpu().numpy())
        rng = np.random.default_rng(13)
        seam = 2**29
        idx = np.unique(np.concatenate([rng.choice(n, size=20000, replace=False), [0, n - 1, ref_index],
                                        np.arange(seam - 70, seam + 70)])).astype(np.int64)
        didx = torch.from_numpy(idx).cuda()
        cols = np.ascontiguousarray(members[:, didx].cpu().numpy()).reshape(cs, 1, 1, -1)
        mm = engine.member_minmax()
        assert mm == (float(members.min()), float(members.max()))
        for m, om, exact, kw, okw in (
                (Measure.PEARSON, oracle_lib.PEARSON, True, {}, {}),
                (Measure.SPEARMAN, oracle_lib.SPEARMAN, True, {}, {}),
                (Measure.MUTUAL_INFORMATION_BINNED, oracle_lib.MI_BINNED, False,
                 dict(num_bins=80, minmax_ref=mm, minmax_query=mm), dict(num_bins=80, minmax_ref=mm))):
            out.fill_(-7.0)
            torch.cuda.synchronize()                                           # the engine computes on its own stream
            engine.compute_device(m, out, ref_xyz, **kw)
            torch.cuda.synchronize()
            (assert_bit_exact if exact else assert_close)(out[didx].cpu().numpy(), oracle.field(om, cols, ref_values, **okw),
                                                          f"{m.name} 1024^3 x {cs} (windows) sampled")
            assert int((out == -7.0).sum()) == 0                              # every voxel of every window was written
        engine.ensemble_stat_device(0, out)
        torch.cuda.synchronize()
        assert_bit_exact(out[didx].cpu().numpy(), oracle.ensemble_stat(0, cols), "ensemble mean 1024^3 (windows) sampled")
        # the host-output entry point (calculateCpu's buffer) on such a grid
        host = engine.compute(Measure.PEARSON, ref_xyz).reshape(-1)
        engine.compute_device(Measure.PEARSON, out, ref_xyz)
        torch.cuda.synchronize()
        assert np.array_equal(host[idx].view(np.uint32), out[didx].cpu().numpy().view(np.uint32))
        del host
    finally:
        del members, out
        torch.cuda.empty_cache()


def test_ragged_million_voxels_with_many_tied_voxels(engine, oracle):
    """101 x 103 x 97 voxels (no power of two anywhere, ragged last blocks), 50 members, ~3 % of the voxels carry ties
    (so the split-sort kernels defer tens of thousands of voxels to the tie list), a NaN here and there."""
    xs, ys, zs, cs = 101, 103, 97, 50
    rng = np.random.default_rng(2026)
    ens = rng.standard_normal((cs, zs, ys, xs)).astype(np.float32)
    flat = ens.reshape(cs, -1)
    n = flat.shape[1]
    tied = rng.choice(n, size=n // 33, replace=False)
    flat[:, tied] = np.round(flat[:, tied] * 2) / 2
    nan_vox = rng.choice(n, size=50, replace=False)
    flat[rng.integers(0, cs, 50), nan_vox] = np.nan
    ref_idx = int(tied[0])                                        # a reference vector with ties of its own
    while np.isnan(flat[:, ref_idx]).any():
        ref_idx += 1
    ref_xyz = (ref_idx % xs, (ref_idx // xs) % ys, ref_idx // (xs * ys))
    ref_values = flat[:, ref_idx].copy()
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    for m, om in ((Measure.PEARSON, oracle_lib.PEARSON), (Measure.SPEARMAN, oracle_lib.SPEARMAN),
                  (Measure.KENDALL, oracle_lib.KENDALL)):
        assert_bit_exact(engine.compute(m, ref_xyz), oracle.field(om, ens, ref_values), f"{m.name} ragged 1M")
    finite = flat[np.isfinite(flat)]
    mm = (float(finite.min()), float(finite.max()))
    assert_close(engine.compute(Measure.MUTUAL_INFORMATION_BINNED, ref_xyz, num_bins=80, minmax_ref=mm, minmax_query=mm),
                 oracle.field(oracle_lib.MI_BINNED, ens, ref_values, num_bins=80, minmax_ref=mm), "binned ragged 1M")
    assert_close(engine.compute(Measure.MUTUAL_INFORMATION_KRASKOV, ref_xyz, k=3),
                 oracle.field(oracle_lib.MI_KRASKOV, ens, ref_values, k=3), "kraskov ragged 1M")


def test_full_size_symmetric_kernels_agree_with_one_reference_kernels(engine, volume):
    """Cross-check of two independent kernel families on all 16.7 M voxels: the symmetric (two-field) kernels with a
    first field whose members are spatially constant (= the reference vector) must return exactly what the
    one-reference kernels return for that vector -- the same estimator, computed by the split-sort / prepared kernels on
    one side and by the per-voxel two-sort kernels on the other."""
    engine.set_grid(XS, YS, ZS, CS)
    engine.bind_members(volume)
    ref_xyz = (XS // 8 + 3, YS // 8 + 1, ZS // 2)
    ref_values = engine.gather_reference(*ref_xyz)
    mm = engine.member_minmax()
    expect = {}
    out = torch.empty(XS * YS * ZS, dtype=torch.float32, device="cuda")
    for m in (Measure.SPEARMAN, Measure.KENDALL, Measure.MUTUAL_INFORMATION_BINNED):
        kw = dict(num_bins=80, minmax_ref=mm, minmax_query=mm) if m == Measure.MUTUAL_INFORMATION_BINNED else {}
        engine.compute_device(m, out, ref_xyz, **kw)
        torch.cuda.synchronize()
        expect[m] = out.clone()
    constant = torch.empty((CS, ZS, YS, XS), dtype=torch.float32, device="cuda")
    for c in range(CS):
        constant[c].fill_(float(ref_values[c]))
    torch.cuda.synchronize()                       # the engine computes on its own stream
    engine.bind_members(constant)                  # X side = first field
    engine.bind_secondary_members(volume)          # Y side = second field
    for m in (Measure.SPEARMAN, Measure.KENDALL, Measure.MUTUAL_INFORMATION_BINNED):
        kw = dict(num_bins=80, minmax_ref=mm, minmax_query=mm) if m == Measure.MUTUAL_INFORMATION_BINNED else {}
        engine.compute_device(m, out, symmetric=True, **kw)
        torch.cuda.synchronize()
        assert engine.last_kernel_name() == "sorted_symmetric_kernel"
        same = (out.view(torch.int32) == expect[m].view(torch.int32)) | (torch.isnan(out) & torch.isnan(expect[m]))
        if m == Measure.MUTUAL_INFORMATION_BINNED:   # fp64 sums in a different order: tolerance, nearly all identical
            assert same.float().mean().item() > 0.999
            torch.testing.assert_close(out, expect[m], rtol=1e-5, atol=1e-6, equal_nan=True)
        else:
            assert bool(same.all()), f"{m.name}: {int((~same).sum())} of {same.numel()} voxels differ"
    del constant
    torch.cuda.empty_cache()

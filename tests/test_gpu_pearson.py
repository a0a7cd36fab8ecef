"""GPU parity: Pearson through the C ABI vs the oracle (bit-exact: same fp32 operation order, no contraction)."""
import numpy as np
import pytest

from correrender_amd import Measure, synth
from parity import assert_bit_exact, bit_identical
import oracle_lib

pytestmark = pytest.mark.gpu


def _run(engine, oracle, ens, ref_xyz):
    cs, zs, ys, xs = ens.shape
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    got = engine.compute(Measure.PEARSON, ref_xyz)
    x, y, z = ref_xyz
    ref_values = ens[:, z, y, x].copy()
    np.testing.assert_array_equal(engine.gather_reference(x, y, z), ref_values)
    want = oracle.field(oracle_lib.PEARSON, ens, ref_values)
    return got.reshape(-1), want


@pytest.mark.parametrize("cs", [2, 3, 8, 9, 16, 17, 31, 32, 33, 48, 56, 64, 65, 72, 80, 96, 100, 112, 128, 130, 150, 160,
                                192, 200, 224, 250, 256, 257, 300, 320, 383, 384, 385, 450, 511, 512, 513, 577, 700, 1000,
                                1023, 1024, 1025, 1100])
def test_pearson_member_counts(engine, oracle, cs):
    # 20*12*9 = 2160 voxels: not a multiple of any block size -> exercises the ragged tail as well
    ens = synth.box_ensemble(20, 12, 9, cs, seed=cs)
    got, want = _run(engine, oracle, ens, (5, 6, 4))
    assert_bit_exact(got, want, f"pearson cs={cs}")


def test_pearson_config0_64cubed_16_members(engine, oracle):
    """BASELINE.json configs[0]: 64^3 synthetic box ensemble, 16 members, Pearson."""
    ens = synth.box_ensemble(64, 64, 64, 16)
    for ref_xyz in [(32, 32, 32), (8, 8, 32)]:
        got, want = _run(engine, oracle, ens, ref_xyz)
        assert_bit_exact(got, want, f"pearson 64^3x16 ref={ref_xyz}")
        assert np.isfinite(got).all()
    # structure: voxels inside the first big box (lambda=1) correlate perfectly with a reference inside it
    got3 = got.reshape(64, 64, 64)
    assert got3[32, 8, 8] == pytest.approx(1.0, abs=1e-6)


def test_pearson_edge_cases(engine, oracle):
    rng = np.random.default_rng(7)
    ens = rng.standard_normal((24, 4, 8, 16)).astype(np.float32)
    ens[:, 0, 0, 1] = 0.0                  # zero variance -> 0/0 = NaN, no epsilon (Correlation.cpp:124-131)
    ens[:, 0, 0, 4] = 3.25                 # constant but the fp32 mean is inexact: whatever the reference gives
    ens[5, 0, 0, 2] = np.nan               # NaN propagates (no NaN test on the Pearson branch)
    ens[7, 0, 0, 3] = np.inf
    ens[:, 1, 1, 1] = ens[:, 2, 2, 2] * 1e-30   # tiny magnitudes (denormal intermediates)
    ens[:, 1, 1, 2] = ens[:, 2, 2, 2] * 1e18    # large magnitudes
    got, want = _run(engine, oracle, ens, (2, 2, 2))
    assert_bit_exact(got, want, "pearson edge cases")
    assert np.isnan(got.reshape(4, 8, 16)[0, 0, 1])
    assert np.isnan(got.reshape(4, 8, 16)[0, 0, 2])


def test_pearson_single_member_is_one(engine):
    ens = synth.box_ensemble(8, 8, 4, 1)
    engine.set_grid(8, 8, 4, 1)
    engine.upload_members(ens)
    got = engine.compute(Measure.PEARSON, (1, 1, 1))
    assert (got == 1.0).all()      # CorrelationCalculator.cpp:882-885


def test_pearson_separate_reference_vector(engine, oracle):
    """CorrelationFieldMode::SEPARATE: reference vector supplied by the caller (CorrelationCalculator.cpp:804-813)."""
    ens = synth.box_ensemble(16, 16, 8, 32, seed=11)
    other = synth.box_ensemble(16, 16, 8, 32, seed=12)
    ref_values = other[:, 4, 8, 8].copy()
    engine.set_grid(16, 16, 8, 32)
    engine.upload_members(ens)
    got = engine.compute(Measure.PEARSON, reference_values=ref_values)
    want = oracle.field(oracle_lib.PEARSON, ens, ref_values)
    assert_bit_exact(got, want, "pearson separate reference")


def test_pearson_device_path_matches_host_path(engine, oracle):
    import torch
    ens = synth.box_ensemble(32, 32, 16, 64, seed=5)
    cs, zs, ys, xs = ens.shape
    dev = torch.from_numpy(ens).cuda()
    engine.set_grid(xs, ys, zs, cs)
    engine.bind_members(dev)
    out = torch.empty(zs * ys * xs, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    engine.compute_device(Measure.PEARSON, out, (16, 16, 8), stream=stream)
    torch.cuda.synchronize()
    want = oracle.field(oracle_lib.PEARSON, ens, ens[:, 8, 16, 16].copy())
    assert_bit_exact(out.cpu().numpy(), want, "pearson device path")
    # device-resident reference vector (the multi-GPU broadcast form)
    dref = torch.from_numpy(ens[:, 3, 2, 1].copy()).cuda()
    engine.compute_device(Measure.PEARSON, out, device_reference=dref, stream=stream)
    torch.cuda.synchronize()
    want = oracle.field(oracle_lib.PEARSON, ens, ens[:, 3, 2, 1].copy())
    assert_bit_exact(out.cpu().numpy(), want, "pearson device reference")
    assert bit_identical(out.cpu().numpy(), want).all()


# pearson_split_kernel (two / four lanes per voxel): every instantiation's first and last member count, both lane counts
@pytest.mark.parametrize("cs", [289, 290, 304, 305, 320, 321, 322, 352, 353, 416, 417, 448, 449, 480, 481, 544, 545, 576, 607, 608, 609, 640,
                                641, 642, 704, 705, 768, 769, 832, 896, 897, 960, 961, 1088, 1089, 1152, 1153, 1216, 1217,
                                1279, 1280, 1281])
def test_pearson_lanes_per_voxel_kernel_boundaries(engine, oracle, cs):
    ens = synth.box_ensemble(20, 12, 9, cs, seed=cs)
    got, want = _run(engine, oracle, ens, (7, 3, 2))
    assert_bit_exact(got, want, f"pearson cs={cs}")
    assert engine.last_kernel_name() == ("pearson_split_kernel" if 288 < cs <= 1216 else "pearson_big_kernel")


@pytest.mark.parametrize("cs", [340, 500, 650, 1000])
def test_pearson_lanes_per_voxel_kernel_edge_cases(engine, oracle, cs):
    """NaN / inf members, constant voxels (sd = 0 -> the plain-division path of the whole wave), extreme magnitudes, in
    every lane group's member range and in the last group's padded granule."""
    rng = np.random.default_rng(cs)
    ens = rng.standard_normal((cs, 3, 8, 32)).astype(np.float32)
    ens[:, 0, 0, 1] = 0.0
    ens[:, 0, 0, 4] = 3.25
    for i, e in enumerate([0, cs // 4, cs // 2 - 1, cs // 2, 3 * cs // 4, cs - 2, cs - 1]):
        ens[e, 1, 0, 2 * i] = np.nan
        ens[e, 1, 1, 2 * i] = np.inf
        ens[e, 1, 2, 2 * i] = -1e30
    ens[:, 2, 1, 1] = ens[:, 2, 2, 2] * 1e-30
    ens[:, 2, 1, 2] = ens[:, 2, 2, 2] * 1e18
    got, want = _run(engine, oracle, ens, (2, 2, 2))
    assert_bit_exact(got, want, f"pearson edge cases cs={cs}")
    assert engine.last_kernel_name() == "pearson_split_kernel"


@pytest.mark.parametrize("cs", [64, 340, 512, 700, 1000, 1200])
def test_pearson_means_that_are_exactly_zero(engine, oracle, cs):
    """Antithetic member pairs (y, -y) sum to a mean of exactly 0 in the reference's sequential fp32 pass, which fails the
    |mean| >= 2^-70 test of the exact-division path (the benchmark's box ensemble has such voxels at 512 members).
    pearson_split_kernel then looks at the deviations themselves: voxels of ordinary magnitude stay on the exact path,
    voxels with deviations below 2^-100 (here: values around 1e-33 and 1e-38, denormals included) take plain divisions."""
    rng = np.random.default_rng(cs)
    xs, ys, zs = 64, 16, 4
    n = xs * ys * zs
    half = rng.standard_normal((cs // 2, n)).astype(np.float32)
    scale = np.ones(n, np.float32)
    scale[n // 4:n // 2] = 1e-33          # whole waves of tiny deviations
    scale[n // 2::97] = 1e-38             # and isolated ones inside ordinary waves
    half *= scale
    ens = np.empty((cs, n), np.float32)
    ens[0::2] = half
    ens[1::2] = -half
    # ordinary voxels in which ONE pair is tiny (1e-35 < 2^-100, and a denormal): sd passes the range test, the deviations don't
    ens[0, n // 2 + 5::97], ens[1, n // 2 + 5::97] = 1e-35, -1e-35
    ens[2, n // 2 + 6::97], ens[3, n // 2 + 6::97] = 1e-41, -1e-41
    ens = ens.reshape(cs, zs, ys, xs)
    got, want = _run(engine, oracle, ens, (3, 2, 3))    # a reference voxel of ordinary magnitude
    assert_bit_exact(got, want, f"pearson zero means cs={cs}")
    assert np.isfinite(want[: n // 4]).all() and not np.isfinite(want[n // 4:n // 2]).any()


@pytest.mark.parametrize("cs", [16, 64, 100, 200, 400, 600, 700, 1100])
def test_pearson_magnitude_sweep_bit_exact(engine, oracle, cs):
    """Every voxel gets its own scale and offset over 50 decades, so the per-voxel quotient (y - mean) / sd is taken at
    standard deviations and means on both sides of the exact-division guard (crf_device.h: sd in [2^-60, 2^60],
    |mean| >= 2^-70) and through the plain-division path of whole waves."""
    rng = np.random.default_rng(1000 + cs)
    xs, ys, zs = 64, 32, 8
    n = xs * ys * zs
    base = rng.standard_normal((cs, n)).astype(np.float64)
    scale = 10.0 ** rng.uniform(-25, 25, n)
    offset = np.where(rng.random(n) < 0.5, 0.0, 10.0 ** rng.uniform(-25, 25, n) * rng.choice([-1.0, 1.0], n))
    # a few voxels right at the guard boundaries
    scale[:8] = [2.0 ** -60, 2.0 ** -61, 2.0 ** 60, 2.0 ** 61, 2.0 ** -59, 2.0 ** 59, 1.0, 1.0]
    offset[:8] = [0.0, 0.0, 0.0, 0.0, 2.0 ** -70, 2.0 ** -71, 2.0 ** -70, 2.0 ** -72]
    with np.errstate(over="ignore"):
        ens = (base * scale + offset).astype(np.float32).reshape(cs, zs, ys, xs)
    got, want = _run(engine, oracle, ens, (3, 2, 1))
    assert_bit_exact(got, want, f"pearson magnitude sweep cs={cs}")

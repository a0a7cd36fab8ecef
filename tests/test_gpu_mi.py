"""GPU parity: binned and Kraskov mutual information through the C ABI vs the oracle.

Both estimators are floating point (fp64 inside, float out): the north-star tolerance is 1e-5 relative (parity.py adds
the stated absolute floor).  The integer cores (bin indices, neighbour counts) are exact, so in practice the float
results are bit-identical except where the fp64 summation order moves a value across a float rounding boundary; the
tests assert the tolerance and additionally require a high fraction of bit-identical voxels."""
import numpy as np
import pytest

from correrender_amd import Measure, synth
from parity import assert_close, bit_identical
import oracle_lib

pytestmark = pytest.mark.gpu


def _check(engine, oracle, ens, measure, omeasure, what, ref_xyz=(1, 2, 3), min_identical=0.999, **kw):
    cs, zs, ys, xs = ens.shape
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    x, y, z = ref_xyz
    ref_values = ens[:, z, y, x].copy()
    okw = {}
    if "num_bins" in kw:
        mn, mx = engine.member_minmax()
        omn, omx = oracle.minmax(ens)
        assert (mn, mx) == (omn, omx)
        okw = dict(num_bins=kw["num_bins"], minmax_ref=(mn, mx))
    if "k" in kw:
        okw["k"] = kw["k"]
    if "kraskov_estimator_index" in kw:
        okw["estimator"] = kw["kraskov_estimator_index"]
    got = engine.compute(measure, ref_xyz, **kw).reshape(-1)
    want = oracle.field(omeasure, ens, ref_values, **okw)
    assert_close(got, want, what)
    frac = bit_identical(got, want).mean()
    assert frac >= min_identical, f"{what}: only {frac:.4%} of the voxels are bit-identical"
    return got, want


@pytest.mark.parametrize("cs", [2, 5, 16, 17, 33, 40, 48, 50, 64, 72, 80, 96, 100, 112, 128])
@pytest.mark.parametrize("num_bins", [10, 80, 100])
def test_binned_member_counts_and_bins(engine, oracle, cs, num_bins):
    ens = synth.box_ensemble(20, 12, 9, cs, seed=200 + cs)
    _check(engine, oracle, ens, Measure.MUTUAL_INFORMATION_BINNED, oracle_lib.MI_BINNED,
           f"binned cs={cs} bins={num_bins}", ref_xyz=(5, 6, 4), num_bins=num_bins)


def test_binned_correlation_coefficient_variant(engine, oracle):
    ens = synth.box_ensemble(32, 16, 8, 64, seed=9)
    got, _ = _check(engine, oracle, ens, Measure.BINNED_MI_CORRELATION_COEFFICIENT, oracle_lib.BINNED_MI_CC,
                    "binned MI-CC", ref_xyz=(4, 4, 4), num_bins=80, min_identical=0.99)
    assert ((got >= 0) & (got <= 1)).all()


def test_binned_edge_cases(engine, oracle):
    rng = np.random.default_rng(11)
    cs = 48
    ens = rng.standard_normal((cs, 3, 8, 16)).astype(np.float32)
    ens[4, 1, 1, 1] = np.nan                       # NaN query value -> NaN (CorrelationCalculator.cpp:1054-1067)
    ens[:, 1, 1, 2] = ens[np.isfinite(ens)].max()  # value exactly 1.0 after normalisation -> last bin
    ens[:, 1, 1, 3] = ens[np.isfinite(ens)].min()
    # the two constant voxels have MI = 0 in exact arithmetic: the reference's summation order leaves ~1e-17, the
    # kernel's order leaves exactly 0 -- inside the tolerance, not bit-identical (2 of 384 voxels)
    got, want = _check(engine, oracle, ens, Measure.MUTUAL_INFORMATION_BINNED, oracle_lib.MI_BINNED,
                       "binned edge cases", ref_xyz=(0, 0, 0), num_bins=80, min_identical=0.99)
    assert np.isnan(got.reshape(3, 8, 16)[1, 1, 1])


def test_binned_constant_data_gives_zero(engine, oracle):
    """max == min: every normalised sample is NaN and skipped; the reference then returns 0.0, not NaN
    (SURVEY Appendix B)."""
    cs = 16
    ens = np.full((cs, 2, 4, 8), 2.5, np.float32)
    engine.set_grid(8, 4, 2, cs)
    engine.upload_members(ens)
    got = engine.compute(Measure.MUTUAL_INFORMATION_BINNED, (0, 0, 0), num_bins=80)
    want = oracle.field(oracle_lib.MI_BINNED, ens, ens[:, 0, 0, 0].copy(), num_bins=80, minmax_ref=(2.5, 2.5))
    assert (want == 0.0).all() and (got == 0.0).all()


def test_binned_partial_skips_with_infinities(engine, oracle):
    """+inf in the data: max = +inf, (inf - min)/inf = NaN is skipped while finite samples normalise to 0:
    probabilities become c/total with total < cs (the compact O(cs^2) path of the kernel)."""
    rng = np.random.default_rng(12)
    cs = 24
    ens = rng.standard_normal((cs, 2, 4, 16)).astype(np.float32)
    ens[3, 0, 1, 2] = np.inf
    ens[5, 0, 1, 2] = np.inf
    ens[7, 1, 3, 9] = np.inf
    for ref in [(0, 0, 0), (2, 1, 0)]:      # reference vector without / with skipped samples
        _check(engine, oracle, ens, Measure.MUTUAL_INFORMATION_BINNED, oracle_lib.MI_BINNED,
               f"binned with infinities ref={ref}", ref_xyz=ref, num_bins=20, min_identical=0.95)


@pytest.mark.parametrize("cs,k", [(8, 1), (16, 1), (16, 3), (33, 2), (64, 2), (64, 3), (64, 4), (100, 3), (128, 4),
                                  (64, 5), (64, 6), (40, 12), (64, 8), (64, 9), (64, 16), (64, 17), (64, 20), (100, 32),
                                  (100, 33), (128, 64), (100, 65), (160, 128), (40, 39), (40, 40)])
def test_kraskov_ksg1(engine, oracle, cs, k):
    ens = synth.normal_ensemble(16, 8, 6, cs, seed=300 + cs)      # tie-free: independent of the noise stream
    ens[:, 0, 0, 1] = 0.8 * ens[:, 0, 0, 0] + 0.6 * ens[:, 0, 0, 1]   # a dependent voxel
    got, want = _check(engine, oracle, ens, Measure.MUTUAL_INFORMATION_KRASKOV, oracle_lib.MI_KRASKOV,
                       f"KSG-1 cs={cs} k={k}", ref_xyz=(0, 0, 0), k=k, min_identical=0.99)
    assert (got >= 0).all()        # clamped at 0 (MutualInformation.cpp:443)


@pytest.mark.parametrize("cs,k,estimator", [(8, 12, 1), (8, 12, 2), (16, 20, 1), (5, 20, 1), (3, 7, 2), (40, 41, 1),
                                             (140, 150, 1)])
def test_kraskov_k_beyond_member_count(engine, oracle, cs, k, estimator):
    """k > cs: the reference accepts it (its UI allows kMax = max(ceil(7 cs / 100), 20), CorrelationCalculator.cpp:592-598):
    the (k+1)-nearest-neighbour query returns all cs points, psi(k) itself enters the estimate
    (MutualInformation.cpp:430-438)."""
    ens = synth.normal_ensemble(16, 8, 3, cs, seed=900 + cs)
    _check(engine, oracle, ens, Measure.MUTUAL_INFORMATION_KRASKOV, oracle_lib.MI_KRASKOV,
           f"KSG-{estimator} cs={cs} k={k} (k > cs)", ref_xyz=(2, 2, 1), k=k, kraskov_estimator_index=estimator,
           min_identical=0.99)


@pytest.mark.parametrize("cs", [12, 64, 100, 140])
def test_binned_positive_overflow_of_the_bin_index(engine, oracle, cs):
    """Caller-supplied extrema much narrower than the data (crf_params min/max are the caller's) and +inf data with a
    finite range: value * numBins reaches 2^31 and the reference's int() -- cvttsd2si on x86-64 -- yields INT_MIN,
    i.e. bin 0 after the clamp (MutualInformation.cpp:66-67), not the last bin."""
    rng = np.random.default_rng(40 + cs)
    ens = (rng.standard_normal((cs, 2, 6, 16)) * 100.0).astype(np.float32)
    ens[1, 0, 2, 3] = np.inf
    ens[2, 1, 4, 5] = -np.inf
    engine.set_grid(16, 6, 2, cs)
    engine.upload_members(ens)
    narrow = (0.0, 1e-8)             # (v - 0) / 1e-8 * 80 > 2^31 for |v| > 0.27
    for ref in [(0, 0, 0), (3, 2, 0)]:
        got = engine.compute(Measure.MUTUAL_INFORMATION_BINNED, ref, num_bins=80, minmax_ref=narrow,
                             minmax_query=narrow).reshape(-1)
        want = oracle.field(oracle_lib.MI_BINNED, ens, ens[:, ref[2], ref[1], ref[0]].copy(), num_bins=80,
                            minmax_ref=narrow)
        assert_close(got, want, f"binned bin-index overflow cs={cs} ref={ref}")


@pytest.mark.parametrize("variant", ["CRF_KRASKOV_SORTED", "CRF_KRASKOV_DIRECT", "CRF_KRASKOV_TILE"])
@pytest.mark.parametrize("dxt,ti4", [("0", "0"), ("1", "0"), ("0", "1"), ("1", "1")])
@pytest.mark.parametrize("cs", [9, 20, 32, 33, 47, 48, 57, 63, 64])
def test_kraskov_kernel_variants(engine, oracle, monkeypatch, variant, dxt, ti4, cs):
    """The three Kraskov kernels (sorted-column, tile-free, LDS-column) forced one at a time -- the dispatch picks one per
    (cs, k), so without this a kernel is only exercised where it is the default; the launchers read the variable at
    every call.  Includes a box ensemble (exact ties, resolved by the noise), a NaN voxel and both estimators."""
    monkeypatch.setenv(variant, "1")
    monkeypatch.setenv("CRF_KRASKOV_DXT", dxt)     # x distances from the prepared table (scalar loads) or per pair
    monkeypatch.setenv("CRF_KRASKOV_TI4", ti4)     # tile-free kernel: 4 or 8 points per sweep
    for k in (1, 2, 3, 4):
        for estimator in (1, 2):
            ens = synth.normal_ensemble(16, 6, 5, cs, seed=70 * cs + k)
            ens[3 % cs, 1, 2, 3] = np.nan
            _check(engine, oracle, ens, Measure.MUTUAL_INFORMATION_KRASKOV, oracle_lib.MI_KRASKOV,
                   f"{variant} KSG-{estimator} cs={cs} k={k}", ref_xyz=(2, 1, 0), k=k,
                   kraskov_estimator_index=estimator, min_identical=0.99)
    box = synth.box_ensemble(16, 6, 5, cs, seed=cs)
    _check(engine, oracle, box, Measure.KMI_CORRELATION_COEFFICIENT, oracle_lib.KMI_CC,
           f"{variant} box ensemble cs={cs}", ref_xyz=(5, 2, 1), k=3, min_identical=0.98)
    expected = {"CRF_KRASKOV_SORTED": "kraskov_sorted_kernel", "CRF_KRASKOV_DIRECT": "kraskov_direct_kernel",
                "CRF_KRASKOV_TILE": "mi_kraskov_kernel"}[variant]
    assert engine.last_kernel_name() == expected


@pytest.mark.parametrize("stage", ["0", "1"])
@pytest.mark.parametrize("cs", [16, 20, 33, 47, 64, 65, 100])
def test_kraskov_tile_staged_in_lds_or_not(engine, oracle, monkeypatch, cs, stage):
    """The tile-free kernel with the voxel tile staged in LDS (default up to 64 members for k >= 3) and without, forced
    either way for every k: ragged member counts (the staging loop fetches every fourth member per wave, 8 at a time),
    a NaN voxel, both estimators, the box ensemble."""
    monkeypatch.setenv("CRF_KRASKOV_DIRECT", "1")
    monkeypatch.setenv("CRF_KRASKOV_STAGE", stage)
    for k in (1, 2, 3, 4):
        for estimator in (1, 2):
            ens = synth.normal_ensemble(16, 6, 5, cs, seed=50 * cs + k)
            ens[5 % cs, 2, 3, 4] = np.nan                      # not the reference voxel (a NaN there is undefined in the reference)
            _check(engine, oracle, ens, Measure.MUTUAL_INFORMATION_KRASKOV, oracle_lib.MI_KRASKOV,
                   f"stage={stage} KSG-{estimator} cs={cs} k={k}", ref_xyz=(2, 1, 0), k=k,
                   kraskov_estimator_index=estimator, min_identical=0.99)
    assert engine.last_kernel_name() == "kraskov_direct_kernel"
    box = synth.box_ensemble(16, 6, 5, cs, seed=cs + 1)
    _check(engine, oracle, box, Measure.KMI_CORRELATION_COEFFICIENT, oracle_lib.KMI_CC,
           f"stage={stage} box ensemble cs={cs}", ref_xyz=(5, 2, 1), k=3, min_identical=0.98)


@pytest.mark.parametrize("dxt", [None, "0", "1"])
@pytest.mark.parametrize("cs", [65, 72, 79, 80, 81, 88, 111, 112, 113, 127, 128, 129, 140])
def test_kraskov_distance_table_boundary(engine, oracle, monkeypatch, cs, dxt):
    """The x-distance table exists up to 128 members (it lies in the preparation buffer) and is used up to 112 members
    for every k, up to 128 for k = 2, 4: both sides of the limits, with the default choice and with the table forced
    off / on, member counts that are not multiples of the sweep width or the 16-candidate batch."""
    if dxt is not None:
        monkeypatch.setenv("CRF_KRASKOV_DXT", dxt)
    for k in (1, 2, 3, 4):
        ens = synth.normal_ensemble(16, 6, 4, cs, seed=5 * cs + k)
        _check(engine, oracle, ens, Measure.MUTUAL_INFORMATION_KRASKOV, oracle_lib.MI_KRASKOV,
               f"KSG-1 cs={cs} k={k}", ref_xyz=(7, 3, 2), k=k, min_identical=0.99)
    assert engine.last_kernel_name() == "kraskov_direct_kernel"


@pytest.mark.parametrize("cs,k", [(16, 2), (64, 3), (100, 5), (64, 12), (100, 40), (160, 70)])
def test_kraskov_ksg2(engine, oracle, cs, k):
    ens = synth.normal_ensemble(16, 8, 6, cs, seed=400 + cs)
    _check(engine, oracle, ens, Measure.MUTUAL_INFORMATION_KRASKOV, oracle_lib.MI_KRASKOV, f"KSG-2 cs={cs} k={k}",
           ref_xyz=(3, 3, 3), k=k, kraskov_estimator_index=2, min_identical=0.99)


def test_kraskov_box_ensemble_with_exact_ties_and_cc(engine, oracle):
    """Box ensemble: lambda = 1 plateaus and the reference voxel itself are exact ties; results there depend on the
    noise stream, which the kernel shares with the oracle (xorshift32)."""
    ens = synth.box_ensemble(32, 32, 8, 64, seed=21)
    got, want = _check(engine, oracle, ens, Measure.KMI_CORRELATION_COEFFICIENT, oracle_lib.KMI_CC,
                       "KMI-CC box ensemble", ref_xyz=(4, 4, 4), k=3, min_identical=0.98)
    assert ((got >= 0) & (got <= 1)).all()


def test_kraskov_nan(engine, oracle):
    ens = synth.normal_ensemble(16, 4, 2, 32, seed=6)
    ens[9, 1, 2, 3] = np.nan
    got, _ = _check(engine, oracle, ens, Measure.MUTUAL_INFORMATION_KRASKOV, oracle_lib.MI_KRASKOV, "KSG nan", k=3,
                    ref_xyz=(0, 0, 0), min_identical=0.99)
    assert np.isnan(got.reshape(2, 4, 16)[1, 2, 3])


@pytest.mark.parametrize("measure", [Measure.MUTUAL_INFORMATION_BINNED, Measure.MUTUAL_INFORMATION_KRASKOV,
                                     Measure.BINNED_MI_CORRELATION_COEFFICIENT, Measure.KMI_CORRELATION_COEFFICIENT])
def test_mi_single_member_is_one(engine, measure):
    ens = synth.box_ensemble(8, 8, 4, 1)
    engine.set_grid(8, 8, 4, 1)
    engine.upload_members(ens)
    assert (engine.compute(measure, (1, 1, 1), k=1) == 1.0).all()


def test_kraskov_gaussian_analytic(engine):
    """Analytic sanity: bivariate Gaussian MI = -0.5 ln(1 - rho^2) = 0.511 at rho = 0.8.  One fixed draw of the
    reference vector is shared by all voxels, so its sampling error does not average out: loose bound."""
    rng = np.random.default_rng(1)
    cs, rho = 128, 0.8
    x = rng.standard_normal(cs).astype(np.float32)
    ens = (rho * x[:, None] + np.sqrt(1 - rho * rho) * rng.standard_normal((cs, 4096))).astype(np.float32)
    ens = ens.reshape(cs, 4, 16, 64)
    engine.set_grid(64, 16, 4, cs)
    engine.upload_members(ens)
    got = engine.compute(Measure.MUTUAL_INFORMATION_KRASKOV, reference_values=x, k=4)
    assert abs(got.mean() - (-0.5 * np.log(1 - rho * rho))) < 0.12

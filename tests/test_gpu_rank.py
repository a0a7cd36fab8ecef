"""GPU parity: Spearman and Kendall through the C ABI vs the oracle -- bit-exact (integer/ordering cores, exact
fp32 tails)."""
import numpy as np
import pytest

from correrender_amd import Measure, synth
from parity import assert_bit_exact
import oracle_lib

pytestmark = pytest.mark.gpu

MEASURES = [(Measure.SPEARMAN, oracle_lib.SPEARMAN), (Measure.KENDALL, oracle_lib.KENDALL)]


def _check(engine, oracle, ens, ref_xyz, measure, omeasure, what, reference_values=None):
    cs, zs, ys, xs = ens.shape
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    if reference_values is None:
        x, y, z = ref_xyz
        ref_values = ens[:, z, y, x].copy()
        got = engine.compute(measure, ref_xyz)
    else:
        ref_values = reference_values
        got = engine.compute(measure, reference_values=reference_values)
    want = oracle.field(omeasure, ens, ref_values)
    assert_bit_exact(got, want, what)
    return got


@pytest.mark.parametrize("measure,omeasure", MEASURES)
@pytest.mark.parametrize("cs", [2, 3, 7, 8, 9, 12, 15, 16, 17, 20, 24, 25, 32, 33, 40, 41, 48, 50, 57, 64, 65, 72, 73, 81, 88, 96, 100, 105, 112, 113, 121, 128])
def test_rank_member_counts(engine, oracle, measure, omeasure, cs):
    ens = synth.box_ensemble(20, 12, 9, cs, seed=100 + cs)
    _check(engine, oracle, ens, (5, 6, 4), measure, omeasure, f"{measure.name} cs={cs}")


@pytest.mark.parametrize("measure,omeasure", MEASURES)
def test_rank_ties_everywhere(engine, oracle, measure, omeasure):
    """Heavily tied data (values rounded to a few levels): fractional ranks, tau-b tie terms, x-tie groups, the
    reference's ignored joint ties."""
    rng = np.random.default_rng(3)
    for cs in (5, 8, 12, 20, 24, 48, 64, 72, 88, 100, 128):   # > 16: split-sort kernels + deferred (tie) list handled by the monolithic pass
        ens = np.round(rng.standard_normal((cs, 4, 8, 16)) * 1.5).astype(np.float32)
        ens[:, 0, 0, 0] = 2.0                       # all-equal voxel: 0/0 -> NaN (Kendall), NaN (Spearman)
        ens[:, 0, 0, 1] = np.arange(cs)             # strictly increasing
        ens[:, 0, 0, 2] = -np.arange(cs)            # strictly decreasing
        ens[:, 0, 0, 3] = np.where(np.arange(cs) % 2 == 0, 0.0, -0.0)   # +0 / -0 are equal
        _check(engine, oracle, ens, (5, 3, 2), measure, omeasure, f"{measure.name} ties cs={cs}")
        # reference vector with ties as well as one without
        _check(engine, oracle, ens, None, measure, omeasure, f"{measure.name} ties/ramp-ref cs={cs}",
               reference_values=np.arange(cs, dtype=np.float32))
        # constant reference vector: n0 - n1 = 0 -> division by zero -> NaN or +-inf exactly as the reference
        _check(engine, oracle, ens, None, measure, omeasure, f"{measure.name} ties/const-ref cs={cs}",
               reference_values=np.full(cs, 1.5, np.float32))


@pytest.mark.parametrize("measure,omeasure", MEASURES)
@pytest.mark.parametrize("cs", [24, 32, 48, 64, 100])
def test_rank_sparse_ties(engine, oracle, measure, omeasure, cs):
    """Continuous data with a tie in a few voxels only: the split-sort kernels answer the tie-free voxels and defer
    the rest to the monolithic kernel through the todo list -- both paths must land in the same output field."""
    rng = np.random.default_rng(cs)
    ens = rng.standard_normal((cs, 6, 8, 16)).astype(np.float32)
    ens[cs - 1, 1, 2, 3] = ens[0, 1, 2, 3]            # tie across the two chunks
    ens[1, 2, 2, 3] = ens[0, 2, 2, 3]                 # tie inside chunk A
    ens[cs - 2, 3, 2, 3] = ens[cs - 1, 3, 2, 3]       # tie inside chunk B
    ens[3, 5, 7, 15] = ens[cs // 2, 5, 7, 15]         # last voxel
    ens[:, 4, 0, 0] = 1.0                             # constant voxel
    _check(engine, oracle, ens, (5, 3, 2), measure, omeasure, f"{measure.name} sparse ties cs={cs}")
    _check(engine, oracle, ens, (3, 2, 1), measure, omeasure, f"{measure.name} sparse ties, tied ref cs={cs}")


def test_kendall_known_answer(engine):
    """x=[1,1,2,2,3,3,4,4], y=[1,2,2,3,3,3,5,4] -> 0.833333254 (=20/24 via two sqrtf; SciPy's
    tau-b gives 0.875 because the reference ignores joint ties)."""
    x = np.array([1, 1, 2, 2, 3, 3, 4, 4], np.float32)
    y = np.array([1, 2, 2, 3, 3, 3, 5, 4], np.float32)
    ens = np.tile(y[:, None, None, None], (1, 2, 2, 2)).astype(np.float32)
    engine.set_grid(2, 2, 2, 8)
    engine.upload_members(ens)
    got = engine.compute(Measure.KENDALL, reference_values=x)
    assert (got == np.float32(0.833333254)).all()


@pytest.mark.parametrize("measure,omeasure", MEASURES)
@pytest.mark.parametrize("cs", [6, 8, 12, 16, 20, 40, 64, 70, 100, 128])
def test_rank_nan_and_inf(engine, oracle, measure, omeasure, cs):
    """NaN in the first slot, the last slot (largest key of the padded kernels) and in either chunk of the split
    kernels; infinities are ordinary ordered values."""
    rng = np.random.default_rng(5 + cs)
    ens = rng.standard_normal((cs, 4, 8, 16)).astype(np.float32)
    ens[3, 1, 2, 3] = np.nan           # NaN in the query ensemble -> quiet NaN (CorrelationCalculator.cpp:929-940)
    ens[cs - 1, 1, 2, 4] = np.nan
    ens[0, 1, 2, 7] = np.nan
    ens[cs // 2, 1, 2, 8] = -np.nan
    ens[0, 1, 2, 5] = np.inf
    ens[cs - 2, 1, 2, 5] = np.inf
    ens[cs - 1, 1, 2, 6] = -np.inf
    got = _check(engine, oracle, ens, (0, 0, 0), measure, omeasure, f"{measure.name} nan/inf cs={cs}")
    g = got.reshape(4, 8, 16)
    assert np.isnan(g[1, 2, [3, 4, 7, 8]]).all() and np.isfinite(g[1, 2, 5]) and np.isfinite(g[1, 2, 6])


@pytest.mark.parametrize("measure", [Measure.SPEARMAN, Measure.KENDALL])
def test_rank_single_member_is_one(engine, measure):
    ens = synth.box_ensemble(8, 8, 4, 1)
    engine.set_grid(8, 8, 4, 1)
    engine.upload_members(ens)
    assert (engine.compute(measure, (1, 1, 1)) == 1.0).all()


@pytest.mark.parametrize("measure,omeasure", MEASURES)
def test_rank_64cubed_16_members(engine, oracle, measure, omeasure):
    ens = synth.box_ensemble(64, 64, 64, 16)
    _check(engine, oracle, ens, (8, 8, 32), measure, omeasure, f"{measure.name} 64^3x16")


def _ensemble_of_close_values(cs):
    rng = np.random.default_rng(1000 + cs)
    xs, ys, zs = 16, 8, 4
    ens = rng.standard_normal((cs, zs, ys, xs)).astype(np.float32)
    flat = ens.reshape(cs, -1)
    n = flat.shape[1]

    def ulps(v, k):
        return (np.float32(v).view(np.int32) + np.int32(k)).view(np.float32) if v >= 0 else (np.float32(v).view(np.int32) - np.int32(k)).view(np.float32)

    for vox in range(0, n, 3):
        members = rng.permutation(cs)
        kind = (vox // 3) % 8
        base = flat[members[0], vox]
        if kind == 0:      # a pair 1 ulp apart
            flat[members[1], vox] = ulps(base, 1)
        elif kind == 1:    # a pair 100 ulps apart (same upper 25 bits or not, as it falls)
            flat[members[1], vox] = ulps(base, 100)
        elif kind == 2:    # a run of three within 5 ulps
            flat[members[1], vox] = ulps(base, 2)
            flat[members[2], vox] = ulps(base, 5)
        elif kind == 3:    # a run of four, one of them an exact tie
            flat[members[1], vox] = ulps(base, 1)
            flat[members[2], vox] = ulps(base, 1)
            flat[members[3], vox] = ulps(base, 3)
        elif kind == 4:    # many pairs at once, all over the range
            for i in range(0, min(cs - 1, 40), 2):
                flat[members[i + 1], vox] = ulps(flat[members[i], vox], 1 + i % 7)
        elif kind == 5:    # around zero: +-denormals, +0 / -0
            flat[members[0], vox] = np.float32(1e-45)
            flat[members[1], vox] = np.float32(-1e-45)
            flat[members[2], vox] = np.float32(0.0)
            flat[members[3], vox] = np.float32(3e-45)
        elif kind == 6:    # the extremes: the two largest and the two smallest close to each other, infinities
            order = np.argsort(flat[:, vox])
            flat[order[-1], vox] = ulps(flat[order[-2], vox], 1)
            flat[order[0], vox] = ulps(flat[order[1], vox], 1 if flat[order[1], vox] < 0 else -1)
            if vox % 2:
                flat[members[5], vox] = np.inf
                flat[members[6], vox] = -np.inf
        else:              # a NaN voxel with close pairs in it
            flat[members[1], vox] = ulps(base, 1)
            flat[members[2], vox] = np.nan
    return ens


@pytest.mark.parametrize("u32", ["0", "1"])
@pytest.mark.parametrize("cs", [33, 40, 47, 48, 56, 63, 64, 65, 72, 73, 80, 81, 95, 96, 100, 104, 111, 120, 127, 128])
def test_spearman_values_that_agree_in_their_upper_bits(engine, oracle, monkeypatch, cs, u32):
    """The 33..128-member Spearman kernel sorts 32-bit composites that carry only the upper 25 bits of a member's key;
    members whose values agree in those bits are put in order from the dropped bits afterwards, runs of three and exact
    ties go to the exact kernel.  Voxels full of such values: pairs one / a few ulps apart (positive, negative, across
    zero, denormals), runs of three and four close values, close values next to exact ties, a NaN, +-inf, the largest
    and smallest members affected.  Both kernels forced (CRF_RANK_U32 = 0: split-sort, 1: u32 network)."""
    monkeypatch.setenv("CRF_RANK_U32", u32)
    ens = _ensemble_of_close_values(cs)
    ref_xyz = (1, 0, 0)                                  # voxel 1: untouched by the loop above
    _check(engine, oracle, ens, ref_xyz, Measure.SPEARMAN, oracle_lib.SPEARMAN, f"Spearman close values cs={cs} u32={u32}")
    expect = "spearman_u32_kernel" if u32 == "1" else ("spearman_kernel" if cs == 64 else "spearman_split_kernel")
    assert engine.last_kernel_name() == expect


PAIR_COUNTS = [129, 130, 143, 144, 145, 159, 160, 161, 176, 177, 192, 193, 208, 209, 224, 225, 239, 240, 241, 255, 256]
PAIR_KERNEL = {Measure.SPEARMAN: "spearman_pair_kernel", Measure.KENDALL: "kendall_pair_kernel"}


@pytest.mark.parametrize("measure,omeasure", MEASURES)
@pytest.mark.parametrize("pair", ["1", "0"])
@pytest.mark.parametrize("cs", PAIR_COUNTS)
def test_rank_129_to_256_members_close_values(engine, oracle, monkeypatch, cs, pair, measure, omeasure):
    """129..256 members: two sorted chunks of 32-bit composites merged through LDS (spearman_pair_kernel /
    kendall_pair_kernel); values that agree in their upper key bits inside a chunk AND across the chunks, runs of three /
    four, exact ties, NaN, +-inf -- the same voxels as above.  CRF_RANK_PAIR=0: the counting kernel for every voxel."""
    monkeypatch.setenv("CRF_RANK_PAIR", pair)
    ens = _ensemble_of_close_values(cs)
    _check(engine, oracle, ens, (1, 0, 0), measure, omeasure, f"{measure.name} close values cs={cs} pair={pair}")
    assert engine.last_kernel_name() == (PAIR_KERNEL[measure] if pair == "1" else "direct_rank_kernel")


@pytest.mark.parametrize("measure,omeasure", MEASURES)
@pytest.mark.parametrize("cs", [129, 150, 200, 256, 257])
def test_rank_129_to_256_members_box_ensemble_and_ties(engine, oracle, cs, measure, omeasure):
    """The benchmark's box ensemble (plateaus: whole voxels of exact ties -> the deferred-voxel list), a grid that is not a
    multiple of 64 voxels, rounded values (ties everywhere), separate reference vectors with ties: a few tie groups, and
    heavy ties (rounded to thirds: a tie group straddles the boundary between the two chunks, which sends every voxel of
    the Kendall kernel to the counting kernel)."""
    ens = synth.box_ensemble(20, 12, 9, cs, seed=cs)
    ens[:, 0, 0, :] = np.round(ens[:, 0, 0, :] * 4)
    ens[7, 1, 1, 1] = np.nan
    _check(engine, oracle, ens, (5, 6, 4), measure, omeasure, f"{measure.name} box ensemble cs={cs}")
    assert engine.last_kernel_name() == (PAIR_KERNEL[measure] if cs <= 256 else "direct_rank_kernel")
    few = ens[:, 3, 3, 3].copy()
    few[5], few[9], few[cs - 1] = few[4], few[4], few[0]
    _check(engine, oracle, ens, None, measure, omeasure, f"{measure.name} reference with a few ties cs={cs}", reference_values=few)
    _check(engine, oracle, ens, None, measure, omeasure, f"{measure.name} heavily tied reference cs={cs}",
           reference_values=np.round(ens[:, 3, 3, 3] * 3))


@pytest.mark.parametrize("cs", [131, 160, 255])
def test_kendall_129_to_256_members_reference_tie_groups_at_the_chunk_boundary(engine, oracle, cs):
    """x-tie groups ending exactly at, starting exactly at, and straddling the boundary between the two sorted chunks
    (position N = half the member count rounded up to 8 in the reference-sorted order)."""
    rng = np.random.default_rng(cs)
    ens = rng.standard_normal((cs, 3, 8, 16)).astype(np.float32)
    n = ((cs + 1) // 2 + 7) // 8 * 8
    base = np.sort(rng.standard_normal(cs).astype(np.float32))
    for lo, hi in [(n - 3, n - 1), (n, n + 2), (n - 2, n + 1), (n - 1, n)]:      # tie group = sorted positions lo..hi
        ref = base.copy()
        ref[lo:hi + 1] = ref[lo]
        ref = ref[rng.permutation(cs)]
        _check(engine, oracle, ens, None, Measure.KENDALL, oracle_lib.KENDALL, f"Kendall x ties {lo}..{hi} of {cs}",
               reference_values=ref)
        assert engine.last_kernel_name() == "kendall_pair_kernel"


@pytest.mark.parametrize("measure,omeasure", MEASURES)
@pytest.mark.parametrize("cs", [20, 33, 48, 64, 65, 100, 128, 130, 200, 256])
def test_rank_masked_voxels(engine, oracle, measure, omeasure, cs):
    """Voxels whose members are all equal (a mask: 0, a positive and a negative constant) and voxels with missing values
    (NaN in one / in every member), in whole waves and scattered: the fast kernels answer the all-equal ones themselves
    (every rank (cs + 1) / 2; tau = 0 / 0) instead of deferring them to the exact kernel."""
    rng = np.random.default_rng(cs)
    xs, ys, zs = 64, 8, 4
    ens = rng.standard_normal((cs, zs, ys, xs)).astype(np.float32)
    flat = ens.reshape(cs, -1)
    n = flat.shape[1]
    flat[:, 0:192] = 0.0            # three whole waves
    flat[:, 192:256] = 5.0
    flat[:, 256:320] = -3.25
    flat[:, 320:448] = np.nan
    flat[cs // 2, 448:512] = np.nan
    flat[:, 600::37] = 0.0          # scattered
    flat[:, 601::41] = 7.5
    flat[:-1, 700] = 1.0            # all equal but one
    flat[1:, 701] = 1.0
    got = _check(engine, oracle, ens, (40, 7, 3), measure, omeasure, f"{measure.name} masked voxels cs={cs}")
    assert np.isnan(got[320:512]).all()

"""GPU parity of the any-member-count kernels (cs > 128: kernels_generic.hip) vs the oracle.  The reference has no
member limit (its own synthetic data set has 1000 members)."""
import numpy as np
import pytest

from correrender_amd import Measure, synth
from parity import assert_bit_exact, assert_close, bit_identical
import oracle_lib

pytestmark = pytest.mark.gpu


def _data(cs, seed, grid=(12, 10, 4)):
    xs, ys, zs = grid
    ens = synth.box_ensemble(xs, ys, zs, cs, seed=seed)
    ens[:, 0, 0, 1] = np.round(ens[:, 0, 0, 1] * 2)       # ties
    ens[:, 0, 0, 2] = 1.25                                # constant voxel
    ens[3, 0, 0, 3] = np.nan
    return ens


@pytest.mark.parametrize("cs", [129, 160, 161, 200, 256, 300, 1000])
def test_generic_rank_measures_bit_exact(engine, oracle, cs):
    ens = _data(cs, 500 + cs)
    _, zs, ys, xs = ens.shape
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    for ref_values in (ens[:, 2, 5, 6].copy(), np.round(ens[:, 1, 1, 1] * 3)):      # without / with x ties
        for m, om in ((Measure.SPEARMAN, oracle_lib.SPEARMAN), (Measure.KENDALL, oracle_lib.KENDALL)):
            got = engine.compute(m, reference_values=ref_values)
            assert_bit_exact(got, oracle.field(om, ens, ref_values), f"generic {m.name} cs={cs}")
    assert engine.last_kernel_name() == ("kendall_pair_kernel" if cs <= 256 else "direct_rank_kernel")


@pytest.mark.parametrize("cs", [130, 200, 600])
def test_generic_mi_measures(engine, oracle, cs):
    ens = _data(cs, 700 + cs)
    _, zs, ys, xs = ens.shape
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    ref_values = ens[:, 2, 5, 6].copy()
    finite = ens[np.isfinite(ens)]
    mm = (float(finite.min()), float(finite.max()))
    for m, om in ((Measure.MUTUAL_INFORMATION_BINNED, oracle_lib.MI_BINNED),
                  (Measure.BINNED_MI_CORRELATION_COEFFICIENT, oracle_lib.BINNED_MI_CC)):
        got = engine.compute(m, reference_values=ref_values, num_bins=80, minmax_ref=mm, minmax_query=mm)
        want = oracle.field(om, ens, ref_values, num_bins=80, minmax_ref=mm)
        assert_close(got, want, f"generic {m.name} cs={cs}")
        assert bit_identical(got, want).mean() > 0.98
    k = max(-(-3 * cs // 100), 1)
    for est in (1, 2):
        got = engine.compute(Measure.MUTUAL_INFORMATION_KRASKOV, reference_values=ref_values, k=k,
                             kraskov_estimator_index=est)
        want = oracle.field(oracle_lib.MI_KRASKOV, ens, ref_values, k=k, estimator=est)
        assert_close(got, want, f"generic KSG-{est} cs={cs} k={k}")
    got = engine.compute(Measure.KMI_CORRELATION_COEFFICIENT, reference_values=ref_values, k=k)
    assert_close(got, oracle.field(oracle_lib.KMI_CC, ens, ref_values, k=k), f"generic KMI-CC cs={cs}")

"""GPU: seeded random sweep over grid shapes, member counts and parameters for every measure -- catches dispatch
boundaries (padding granules, split / direct kernels, k ranges, ragged voxel counts) that the targeted tests do not
enumerate.  Bit-exact for Pearson / Spearman / Kendall, tolerance for the MI estimators."""
import numpy as np
import pytest

from correrender_amd import Measure
from parity import assert_bit_exact, assert_close
import oracle_lib

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(20261003)
    out = []
    for i in range(40):
        cs = int(rng.choice([rng.integers(2, 17), rng.integers(17, 65), rng.integers(65, 129), rng.integers(129, 400)]))
        xs, ys, zs = (int(v) for v in rng.integers(1, 24, 3))
        while xs * ys * zs < 8:
            xs += 3
        out.append((i, cs, xs, ys, zs, int(rng.integers(0, 2**31))))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: f"{c[0]}-cs{c[1]}-{c[2]}x{c[3]}x{c[4]}")
def test_random_configuration(engine, oracle, case):
    _, cs, xs, ys, zs, seed = case
    rng = np.random.default_rng(seed)
    ens = rng.standard_normal((cs, zs, ys, xs)).astype(np.float32)
    n = xs * ys * zs
    flat = ens.reshape(cs, n)
    flat[:, rng.integers(0, n)] = np.round(flat[:, rng.integers(0, n)])          # a voxel with ties
    if n > 4:
        flat[rng.integers(0, cs), rng.integers(0, n)] = np.nan                    # a NaN somewhere
    dep = rng.integers(0, n)
    ref_idx = int(rng.integers(0, n))
    ref_xyz = (ref_idx % xs, (ref_idx // xs) % ys, ref_idx // (xs * ys))
    if np.isnan(flat[:, ref_idx]).any():
        flat[:, ref_idx] = rng.standard_normal(cs)
    flat[:, dep] = 0.7 * flat[:, ref_idx] + 0.3 * flat[:, dep]                      # a dependent voxel
    ref_values = flat[:, ref_idx].copy()
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    for m, om in ((Measure.PEARSON, oracle_lib.PEARSON), (Measure.SPEARMAN, oracle_lib.SPEARMAN),
                  (Measure.KENDALL, oracle_lib.KENDALL)):
        assert_bit_exact(engine.compute(m, ref_xyz), oracle.field(om, ens, ref_values), f"{m.name} {case}")
    finite = flat[np.isfinite(flat)]
    mm = (float(finite.min()), float(finite.max()))
    nb = int(rng.integers(4, 120))
    assert_close(engine.compute(Measure.MUTUAL_INFORMATION_BINNED, ref_xyz, num_bins=nb, minmax_ref=mm, minmax_query=mm),
                 oracle.field(oracle_lib.MI_BINNED, ens, ref_values, num_bins=nb, minmax_ref=mm), f"binned nb={nb} {case}")
    k = int(rng.integers(1, min(cs, 40)))
    est = int(rng.integers(1, 3))
    assert_close(engine.compute(Measure.MUTUAL_INFORMATION_KRASKOV, ref_xyz, k=k, kraskov_estimator_index=est),
                 oracle.field(oracle_lib.MI_KRASKOV, ens, ref_values, k=k, estimator=est), f"KSG-{est} k={k} {case}")


def _cases2():
    rng = np.random.default_rng(77)
    return [(i, int(rng.integers(2, 200)), int(rng.integers(2, 20)), int(rng.integers(1, 12)), int(rng.integers(1, 8)),
             int(rng.integers(0, 2**31))) for i in range(16)]


@pytest.mark.parametrize("case", _cases2(), ids=lambda c: f"{c[0]}-cs{c[1]}-{c[2]}x{c[3]}x{c[4]}")
def test_random_two_field_modes_and_siblings(engine, oracle, case):
    _, cs, xs, ys, zs, seed = case
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((cs, zs, ys, xs)).astype(np.float32)
    b = (rng.uniform(-1, 1) * a + rng.standard_normal((cs, zs, ys, xs))).astype(np.float32)
    b.reshape(cs, -1)[:, 0] = np.round(b.reshape(cs, -1)[:, 0])
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(a)
    engine.upload_secondary_members(b)
    for m, om in ((Measure.PEARSON, 0), (Measure.SPEARMAN, 1), (Measure.KENDALL, 2)):
        assert_bit_exact(engine.compute(m, symmetric=True), oracle.symmetric_field(om, a, b), f"symmetric {m.name} {case}")
    mm_a, mm_b = oracle.minmax(a), oracle.minmax(b)
    nb = int(rng.integers(4, 100))
    assert_close(engine.compute(Measure.MUTUAL_INFORMATION_BINNED, symmetric=True, num_bins=nb, minmax_ref=mm_a,
                                minmax_query=mm_b),
                 oracle.symmetric_field(3, a, b, num_bins=nb, minmax_ref=mm_a, minmax_query=mm_b), f"sym binned {case}")
    k = int(rng.integers(1, min(cs, 30)))
    assert_close(engine.compute(Measure.MUTUAL_INFORMATION_KRASKOV, symmetric=True, k=k),
                 oracle.symmetric_field(4, a, b, k=k), f"sym kraskov k={k} {case}")
    assert_bit_exact(engine.ensemble_stat(0), oracle.ensemble_stat(0, a), "mean")
    assert_bit_exact(engine.ensemble_stat(1), oracle.ensemble_stat(1, a), "spread")
    op = int(rng.integers(0, 6))
    lo, hi = int(rng.integers(0, cs + 1)), int(rng.integers(0, cs + 1))
    assert_bit_exact(engine.set_predicate(op, 0.1, lo, hi), oracle.set_predicate(op, 0.1, lo, hi, a), "set predicate")
    assert_close(engine.dkl("binned", num_bins=nb), oracle.dkl(0, a, num_bins=nb), f"dkl binned {case}")
    if cs > 2:
        kk = int(rng.integers(1, min(cs - 1, 20) + 1))
        assert_close(engine.dkl("knn", k=kk), oracle.dkl(1, a, k=kk), f"dkl knn k={kk} {case}")

"""CPU: the oracle against independent formulations (scipy / numpy / analytic).  This is the cross-check that stands
in for a reference build of the two MI estimators (unbuildable here: boost, sgl, glm absent)."""
import math

import numpy as np
import pytest
import scipy.special
import scipy.stats


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(42)


def test_pearson_spearman_kendall_vs_scipy(oracle, rng):
    for n in (5, 16, 64, 200):
        x = rng.standard_normal(n).astype(np.float32)
        y = (0.5 * x + rng.standard_normal(n)).astype(np.float32)
        assert oracle.pearson(x, y) == pytest.approx(scipy.stats.pearsonr(x.astype(np.float64), y.astype(np.float64))[0], abs=2e-6)
        assert oracle.spearman(x, y) == pytest.approx(scipy.stats.spearmanr(x, y)[0], abs=2e-6)
        assert oracle.kendall(x, y) == pytest.approx(scipy.stats.kendalltau(x, y)[0], abs=2e-6)   # tie-free: tau-b
        np.testing.assert_array_equal(oracle.ranks(x), scipy.stats.rankdata(x).astype(np.float32))
    xt = np.round(rng.standard_normal(40) * 2).astype(np.float32)
    np.testing.assert_array_equal(oracle.ranks(xt), scipy.stats.rankdata(xt).astype(np.float32))   # mid-ranks


def test_kendall_ties_without_joint_ties_equals_scipy(oracle):
    """Without JOINT ties the reference's n3 := 0 shortcut is exact, so tau-b matches SciPy."""
    x = np.array([1, 1, 2, 3, 4, 5, 5, 6], np.float32)
    y = np.array([3, 4, 1, 1, 7, 2, 9, 8], np.float32)
    assert oracle.kendall(x, y) == pytest.approx(scipy.stats.kendalltau(x, y)[0], abs=1e-6)


def _binned_numpy(x01, y01, nb):
    n = len(x01)
    bx = np.clip((x01.astype(np.float64) * nb).astype(int), 0, nb - 1)
    by = np.clip((y01.astype(np.float64) * nb).astype(int), 0, nb - 1)
    h = np.zeros((nb, nb))
    np.add.at(h, (bx, by), 1.0)
    p = h / h.sum()
    px, py = p.sum(1), p.sum(0)
    ent = lambda q: -(q[q > 0] * np.log(q[q > 0])).sum()
    return ent(px) + ent(py) - ent(p.ravel())


def test_binned_mi_vs_numpy_histogram(oracle, rng):
    for n, nb in ((16, 10), (64, 80), (100, 80), (128, 100)):
        x = rng.random(n).astype(np.float32)
        y = np.clip(0.7 * x + 0.3 * rng.random(n), 0, 1).astype(np.float32)
        y[0], y[1] = 1.0, 0.0                               # value exactly 1.0 -> last bin
        assert oracle.mi_binned(x, y, nb) == pytest.approx(_binned_numpy(x, y, nb), rel=1e-6, abs=1e-6)
    x = rng.random(64).astype(np.float32)
    assert oracle.mi_binned(x, x, 80) == pytest.approx(_binned_numpy(x, x, 80), rel=1e-6)   # MI(x,x) = H(x)


def _ksg1_numpy(x, y, k, noise_x, noise_y):
    px = x.astype(np.float64) + noise_x.astype(np.float64) * 1e-10
    py = y.astype(np.float64) + noise_y.astype(np.float64) * 1e-10
    n = len(x)
    d = np.maximum(np.abs(px[:, None] - px[None, :]), np.abs(py[:, None] - py[None, :]))
    dk = np.sort(d, axis=1)[:, k]                           # k-th neighbour (self at index 0)
    r = dk - 1e-15
    nx = np.maximum(((px[None, :] >= (px - r)[:, None]) & (px[None, :] < (px + r)[:, None])).sum(1), 1)
    ny = np.maximum(((py[None, :] >= (py - r)[:, None]) & (py[None, :] < (py + r)[:, None])).sum(1), 1)
    psi = scipy.special.digamma
    return max(float(psi(k) + psi(n) - psi(nx).mean() - psi(ny).mean()), 0.0)


def test_kraskov_vs_numpy_bruteforce(oracle, rng):
    for n, k in ((16, 1), (64, 2), (64, 3), (100, 3), (128, 4)):
        x = rng.standard_normal(n).astype(np.float32)
        y = (0.6 * x + 0.8 * rng.standard_normal(n)).astype(np.float32)
        want = _ksg1_numpy(x, y, k, oracle.noise01(0, n), oracle.noise01(1, n))
        assert oracle.mi_kraskov(x, y, k, 1) == pytest.approx(want, rel=2e-6, abs=2e-6)


def test_kraskov_statistics(oracle, rng):
    """KSG-1 on a bivariate Gaussian: close to -0.5 ln(1-rho^2); independent data: close to 0."""
    rho, n, k = 0.8, 64, 3
    vals, vals0, vals2 = [], [], []
    for _ in range(300):
        a = rng.standard_normal(n).astype(np.float32)
        b = (rho * a + math.sqrt(1 - rho * rho) * rng.standard_normal(n)).astype(np.float32)
        c = rng.standard_normal(n).astype(np.float32)
        vals.append(oracle.mi_kraskov(a, b, k, 1))
        vals2.append(oracle.mi_kraskov(a, b, k, 2))
        vals0.append(oracle.mi_kraskov(a, c, k, 1))
    analytic = -0.5 * math.log(1 - rho * rho)
    assert abs(np.mean(vals) - analytic) < 0.06      # SURVEY 8(c) probe: 0.493 vs 0.511
    assert abs(np.mean(vals2) - analytic) < 0.08
    assert np.mean(vals0) < 0.08


def test_digamma_table_vs_scipy(oracle):
    for n in range(1, 1025):
        assert oracle.digamma(n) == pytest.approx(float(scipy.special.digamma(n)), rel=0, abs=4e-15)
    assert math.isnan(oracle.digamma(0))
    assert oracle.digamma(64) - oracle.digamma(3) == pytest.approx(3.228266, abs=1e-6)   # SURVEY 8(c) probe value


def test_noise_stream_is_documented_xorshift32(oracle):
    s = 617406168
    u = []
    for _ in range(4):
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        u.append(np.float32(s >> 8) * np.float32(1.0 / 16777216.0))
    np.testing.assert_array_equal(oracle.noise01(0, 4), np.array(u, np.float32))
    assert ((oracle.noise01(1, 256) >= 0) & (oracle.noise01(1, 256) < 1)).all()

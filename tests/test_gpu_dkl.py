"""DKLCalculator: oracle sanity on CPU, HIP path vs oracle on the GPU.

The estimators work in fp64 and cast to float; device log/exp differ from the host libm by <= 1 ulp, so the tolerance
is the north-star 1e-5 relative (+ the 1e-6 absolute floor of tests/parity.py); the integer parts (bins, the sorted
order, the window placement of the k-NN search) are exact."""
import numpy as np
import pytest

from correrender_amd import CorrFieldError
from parity import assert_close, bit_identical


def _ensemble(cs, seed, shape=(4, 6, 16)):
    rng = np.random.default_rng(seed)
    ens = rng.standard_normal((cs,) + shape).astype(np.float32)
    ens[:, 0, 0, 0] = rng.uniform(-3, 5, cs)                  # uniform: DKL > 0
    ens[:, 0, 0, 1] = rng.exponential(2.0, cs)                # skewed
    ens[:, 0, 0, 2] = 4.0                                     # constant: stdev 0 -> NaN
    ens[cs // 2, 0, 0, 3] = np.nan                            # NaN member -> NaN
    ens[:, 0, 0, 4] = np.round(ens[:, 0, 0, 4])               # duplicates: k-NN distance 0 -> log 0 -> NaN
    ens[:, 0, 0, 5] = ens[:, 0, 0, 5] * 1e-3 + 1e4            # large offset, small spread
    return ens


def test_oracle_dkl_matches_theory(oracle):
    """Uniform samples normalised to unit variance: D_KL(U || N(0,1)) = 0.5 ln(2 pi e) - ln sqrt(12) = 0.1765."""
    rng = np.random.default_rng(0)
    ens = rng.uniform(size=(1000, 1, 1, 400)).astype(np.float32)
    knn = oracle.dkl(1, ens, k=30)
    assert abs(knn.mean() - (0.5 * np.log(2 * np.pi * np.e) - np.log(np.sqrt(12)))) < 0.03
    gauss = rng.standard_normal((1000, 1, 1, 400)).astype(np.float32)
    assert oracle.dkl(1, gauss, k=30).mean() < 0.03 and oracle.dkl(0, gauss, num_bins=40).mean() < 0.1
    one = oracle.dkl(1, gauss[:1], k=1)
    assert (one == 1.0).all()                                 # cs == 1 -> 1 (DKLCalculator.cpp:184-187)


@pytest.mark.gpu
@pytest.mark.parametrize("cs", [2, 3, 16, 17, 33, 50, 64, 80, 96, 97, 128, 300])   # up to 96: the register form
def test_gpu_dkl_binned(engine, oracle, cs):
    ens = _ensemble(cs, 10 + cs)
    _, zs, ys, xs = ens.shape
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    for bins in (10, 80, 300):
        got = engine.dkl("binned", num_bins=bins)
        want = oracle.dkl(0, ens, num_bins=bins)
        assert_close(got, want, f"DKL binned cs={cs} bins={bins}")
        assert bit_identical(got, want).mean() > 0.98
    g = got.reshape(zs, ys, xs)
    assert np.isnan(g[0, 0, 3])
    # the constant voxel is NaN only when 1/cs is exact (otherwise the fp64 mean is off by an ulp and stdev > 0)
    assert np.isnan(g[0, 0, 2]) == np.isnan(want.reshape(zs, ys, xs)[0, 0, 2])


@pytest.mark.gpu
@pytest.mark.parametrize("cs", [2, 3, 4, 5, 16, 17, 33, 50, 64, 80, 96, 97, 128, 300])
def test_gpu_dkl_knn(engine, oracle, cs):
    ens = _ensemble(cs, 20 + cs)
    _, zs, ys, xs = ens.shape
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    # k = 1, 2 up to 96 members: the register form of the window descent; larger k / member counts: the LDS column
    for k in sorted({1, min(2, cs - 1), max(1, -(-3 * cs // 100)), min(cs - 1, 7)}):
        got = engine.dkl("knn", k=k)
        want = oracle.dkl(1, ens, k=k)
        assert_close(got, want, f"DKL k-NN cs={cs} k={k}")
        assert bit_identical(got, want).mean() > 0.98
    g = got.reshape(zs, ys, xs)
    assert np.isnan(g[0, 0, 2]) and np.isnan(g[0, 0, 3])
    w = want.reshape(zs, ys, xs)
    assert np.isnan(g[0, 0, 4]) == np.isnan(w[0, 0, 4])
    if cs in (16, 50, 64, 128, 300):
        assert np.isnan(g[0, 0, 4])                            # duplicate values -> log(0) -> inf -> NaN (DKL.cpp:158-160)
    # and with k = 1 every voxel of rounded values has a zero nearest-neighbour distance
    if cs >= 16:
        assert np.isnan(engine.dkl("knn", k=1).reshape(zs, ys, xs)[0, 0, 4])


@pytest.mark.gpu
def test_gpu_dkl_single_member_and_errors(engine):
    engine.set_grid(4, 4, 2, 1)
    engine.upload_members(np.zeros((1, 2, 4, 4), np.float32))
    assert (engine.dkl("knn", k=1) == 1.0).all() and (engine.dkl("binned") == 1.0).all()
    engine.set_grid(4, 4, 2, 8)
    engine.upload_members(np.zeros((8, 2, 4, 4), np.float32))
    with pytest.raises(CorrFieldError, match="k="):
        engine.dkl("knn", k=8)
    with pytest.raises(CorrFieldError, match="num_bins"):
        engine.dkl("binned", num_bins=0)

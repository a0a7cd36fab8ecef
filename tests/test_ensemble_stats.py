"""Ensemble mean / spread: oracle vs numpy, GPU vs oracle (bit-exact: sequential fp32)."""
import numpy as np
import pytest

from correrender_amd import synth
from parity import assert_bit_exact


def _data(cs, seed):
    ens = synth.box_ensemble(20, 12, 9, cs, seed=seed)
    ens[1 % cs, 0, 0, 1] = np.nan                 # one NaN: skipped
    ens[:, 0, 0, 2] = np.nan                      # all NaN: NaN out
    if cs > 1:
        ens[1:, 0, 0, 3] = np.nan                 # a single valid value: mean = value, spread = NaN
    return ens


@pytest.mark.parametrize("cs", [1, 2, 16, 100])
def test_oracle_vs_numpy(oracle, cs):
    ens = _data(cs, 30 + cs)
    flat = ens.reshape(cs, -1).astype(np.float64)
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mean = np.nanmean(flat, axis=0)
            spread = np.nanstd(flat, axis=0, ddof=1)
    got_mean, got_spread = oracle.ensemble_stat(0, ens), oracle.ensemble_stat(1, ens)
    np.testing.assert_allclose(got_mean, mean, rtol=2e-6, atol=1e-6, equal_nan=True)
    ok = ~np.isnan(spread) & ~np.isinf(spread)
    np.testing.assert_allclose(got_spread[ok], spread[ok], rtol=2e-5, atol=1e-6)
    assert np.isnan(got_mean.reshape(9, 12, 20)[0, 0, 2]) and np.isnan(got_spread.reshape(9, 12, 20)[0, 0, 2])
    if cs > 1:
        assert got_mean.reshape(9, 12, 20)[0, 0, 3] == ens[0, 0, 0, 3] and np.isnan(got_spread.reshape(9, 12, 20)[0, 0, 3])


@pytest.mark.gpu
@pytest.mark.parametrize("cs", [1, 2, 16, 17, 33, 64, 100, 128, 200, 300])
def test_gpu_ensemble_stats(engine, oracle, cs):
    ens = _data(cs, 30 + cs)
    _, zs, ys, xs = ens.shape
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    for kind in (0, 1):
        assert_bit_exact(engine.ensemble_stat(kind), oracle.ensemble_stat(kind, ens), f"ensemble stat {kind} cs={cs}")

"""CPU, build container only: the oracle against the reference's own object code (oracle/_ref/libref_corr.so built
from /root/reference/src/Calculators/Correlation.cpp) on randomized inputs.  Skipped where the reference build is
absent; the committed golden vectors carry the same pin everywhere else."""
import numpy as np
import pytest

import oracle_lib
from parity import assert_bit_exact

pytestmark = pytest.mark.skipif(not oracle_lib.reference_available(), reason="oracle/_ref not built (no /root/reference)")


@pytest.fixture(scope="module")
def ref():
    return oracle_lib.load_reference()


@pytest.mark.parametrize("cs", [2, 3, 5, 16, 33, 64, 100, 128, 256, 300])
def test_fields_bit_exact(oracle, ref, cs):
    rng = np.random.default_rng(cs)
    ens = rng.standard_normal((cs, 3, 5, 7)).astype(np.float32)
    ens[:, 0, 0, 0] = np.round(ens[:, 0, 0, 0])          # ties
    ens[:, 0, 0, 1] = 0.25                               # constant voxel
    if cs > 2:
        ens[1, 1, 1, 1] = np.nan
    refv = ens[:, 2, 3, 4].copy()
    for m in (0, 1, 2):
        assert_bit_exact(oracle.field(m, ens, refv), ref.field(m, ens, refv), f"measure {m} cs={cs}")
    tied_ref = np.round(refv * 2)                          # ties in the reference vector (x-tie groups, n1)
    for m in (0, 1, 2):
        assert_bit_exact(oracle.field(m, ens, tied_ref), ref.field(m, ens, tied_ref), f"measure {m} tied ref cs={cs}")


def test_primitives_bit_exact(oracle, ref):
    rng = np.random.default_rng(0)
    for trial in range(300):
        n = int(rng.integers(2, 200))
        x = rng.standard_normal(n).astype(np.float32)
        y = rng.standard_normal(n).astype(np.float32)
        if trial % 3 == 0:
            x, y = np.round(x * 2), np.round(y * 2)
        assert np.float32(oracle.pearson(x, y)).tobytes() == np.float32(ref.pearson(x, y)).tobytes() or (
            np.isnan(oracle.pearson(x, y)) and np.isnan(ref.pearson(x, y)))
        np.testing.assert_array_equal(oracle.ranks(x), ref.ranks(x))
        a, b = np.float32(oracle.kendall(x, y)), np.float32(ref.kendall(x, y))
        assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b))


def test_kendall_tau_a_cross_check(oracle, ref):
    """computeKendallSlow (tau-a, Correlation.cpp:471-482) equals tau-b on tie-free data up to the float tail."""
    rng = np.random.default_rng(1)
    x = rng.standard_normal(50).astype(np.float32)
    y = rng.standard_normal(50).astype(np.float32)
    assert abs(ref.kendall_slow(x, y) - oracle.kendall(x, y)) < 1e-6

"""Pair-request mode: the estimator between two arbitrary voxels per request.
CPU: the oracle's pair evaluation vs the reference's own primitives (oracle/_ref) and the golden vectors.
GPU: crf_compute_requests vs the oracle."""
from pathlib import Path

import numpy as np
import pytest

from correrender_amd import Measure, synth
from parity import assert_bit_exact, assert_close
import oracle_lib

GOLDEN = Path(__file__).resolve().parent / "golden" / "pair_requests.npz"


def _case(cs, seed, n=400, grid=(12, 10, 6)):
    xs, ys, zs = grid
    ens = synth.box_ensemble(xs, ys, zs, cs, seed=seed)
    ens[:, 0, 0, 1] = np.round(ens[:, 0, 0, 1] * 2)        # ties
    ens[:, 0, 0, 2] = 0.5                                  # constant vector
    ens[1, 0, 0, 3] = np.nan
    rng = np.random.default_rng(seed)
    pairs = np.stack([rng.integers(0, xs, n), rng.integers(0, ys, n), rng.integers(0, zs, n),
                      rng.integers(0, xs, n), rng.integers(0, ys, n), rng.integers(0, zs, n)], axis=1)
    pairs[0] = [1, 0, 0, 1, 0, 0]                          # a voxel with itself (ties)
    pairs[1] = [2, 0, 0, 5, 5, 3]                          # constant vs random -> 0/0
    pairs[2] = [3, 0, 0, 4, 4, 2]                          # NaN
    pairs[3] = [4, 4, 2, 4, 4, 2]                          # identical vectors
    idx = lambda p: (p[:, 2] * ys + p[:, 1]) * xs + p[:, 0]
    return ens, pairs, idx(pairs[:, 0:3]), idx(pairs[:, 3:6])


@pytest.mark.skipif(not oracle_lib.reference_available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("cs", [2, 16, 64, 100])
def test_oracle_pairs_vs_reference_primitives(oracle, cs):
    ref = oracle_lib.load_reference()
    ens, _, ii, jj = _case(cs, 900 + cs)
    for m in (0, 1, 2):
        assert_bit_exact(oracle.pair_requests(m, ens, ii, jj), ref.pair_requests(m, ens, ii, jj), f"pairs measure {m}")


def test_oracle_pairs_vs_golden(oracle):
    d = np.load(GOLDEN)
    ens, ii, jj = d["members"], d["idx_i"], d["idx_j"]
    for m, name in ((0, "pearson"), (1, "spearman"), (2, "kendall")):
        assert_bit_exact(oracle.pair_requests(m, ens, ii, jj), d[f"{name}__reference"], f"golden pairs {name}")
    assert_bit_exact(oracle.pair_requests(3, ens, ii, jj, num_bins=80), d["mi_binned__restatement"], "golden pairs binned")
    assert_bit_exact(oracle.pair_requests(4, ens, ii, jj, k=3), d["mi_kraskov__restatement"], "golden pairs kraskov")


@pytest.mark.gpu
@pytest.mark.parametrize("cs", [2, 16, 17, 48, 64, 65, 100, 128, 300])
def test_gpu_pair_requests(engine, oracle, cs):
    ens, pairs, ii, jj = _case(cs, 900 + cs, n=400 + cs % 7)      # request counts that are not a multiple of 64
    _, zs, ys, xs = ens.shape
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    for m, om in ((Measure.PEARSON, 0), (Measure.SPEARMAN, 1), (Measure.KENDALL, 2)):
        assert_bit_exact(engine.compute_requests(m, pairs), oracle.pair_requests(om, ens, ii, jj), f"gpu pairs {m.name} cs={cs}")
        if om == 0:    # Pearson: the two-vector register kernel up to 128 members
            assert engine.last_kernel_name() == ("pearson_request_kernel" if cs <= 128 else "pair_request_kernel")
        if om > 0:     # Spearman / Kendall: the sort-based two-vector kernels up to 128 members
            assert engine.last_kernel_name() == ("sorted_request_kernel" if cs <= 128 else "pair_request_kernel")
            assert_bit_exact(engine.compute_requests(m, pairs, absolute_value=True),
                             oracle.pair_requests(om, ens, ii, jj, use_abs=True), f"gpu pairs |{m.name}| cs={cs}")
    k = min(3, max(cs - 1, 1))
    assert_close(engine.compute_requests(Measure.MUTUAL_INFORMATION_BINNED, pairs, num_bins=80),
                 oracle.pair_requests(3, ens, ii, jj, num_bins=80), f"gpu pairs binned cs={cs}")
    assert engine.last_kernel_name() == ("sorted_request_kernel" if 2 <= cs <= 128 else "pair_request_kernel")
    assert_close(engine.compute_requests(Measure.BINNED_MI_CORRELATION_COEFFICIENT, pairs, num_bins=40),
                 oracle.pair_requests(5, ens, ii, jj, num_bins=40), f"gpu pairs binned cc cs={cs}")
    assert_close(engine.compute_requests(Measure.MUTUAL_INFORMATION_KRASKOV, pairs, k=k),
                 oracle.pair_requests(4, ens, ii, jj, k=k), f"gpu pairs kraskov cs={cs}")
    assert_close(engine.compute_requests(Measure.KMI_CORRELATION_COEFFICIENT, pairs, k=k),
                 oracle.pair_requests(6, ens, ii, jj, k=k), f"gpu pairs kmi cc cs={cs}")
    got = engine.compute_requests(Measure.PEARSON, pairs, absolute_value=True)
    assert_bit_exact(got, oracle.pair_requests(0, ens, ii, jj, use_abs=True), "gpu pairs |pearson|")


@pytest.mark.gpu
def test_gpu_pair_requests_golden_and_errors(engine):
    from correrender_amd import CorrFieldError
    d = np.load(GOLDEN)
    ens = d["members"]
    cs, zs, ys, xs = ens.shape
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    pairs = d["pairs"]
    for m, name in ((Measure.PEARSON, "pearson"), (Measure.SPEARMAN, "spearman"), (Measure.KENDALL, "kendall")):
        assert_bit_exact(engine.compute_requests(m, pairs), d[f"{name}__reference"], f"gpu golden pairs {name}")
    with pytest.raises(CorrFieldError) as e:
        engine.compute_requests(Measure.PEARSON, [[0, 0, 0, xs, 0, 0]])
    assert e.value.code == 1 and "outside" in e.value.message
    assert engine.compute_requests(Measure.PEARSON, np.zeros((0, 6), int)).size == 0


@pytest.mark.gpu
@pytest.mark.parametrize("cs", [5, 32, 64, 100, 200])
def test_gpu_two_field_pair_requests(engine, oracle, cs):
    """Two-field request mode (setUseSecondaryFields, HEBChartCorrelation.cpp:1164-1169): the i side reads the first
    field, the j side the second.  Expectation: the oracle's pair evaluation on the two fields stacked along z, with the
    j indices moved into the second half."""
    from correrender_amd import CorrFieldError
    ens, pairs, ii, jj = _case(cs, 1700 + cs, n=333)
    rng = np.random.default_rng(cs)
    second = (0.5 * ens + rng.standard_normal(ens.shape)).astype(np.float32)
    second[:, 0, 0, 1] = np.round(second[:, 0, 0, 1])
    _, zs, ys, xs = ens.shape
    stacked = np.concatenate([ens, second], axis=1)
    jj2 = jj + xs * ys * zs
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    with pytest.raises(CorrFieldError):
        engine.compute_requests(Measure.PEARSON, pairs, query_from_secondary=True)
    engine.upload_secondary_members(second)
    for m, om in ((Measure.PEARSON, 0), (Measure.SPEARMAN, 1), (Measure.KENDALL, 2)):
        assert_bit_exact(engine.compute_requests(m, pairs, query_from_secondary=True),
                         oracle.pair_requests(om, stacked, ii, jj2), f"two-field pairs {m.name} cs={cs}")
    assert_close(engine.compute_requests(Measure.MUTUAL_INFORMATION_BINNED, pairs, num_bins=60, query_from_secondary=True),
                 oracle.pair_requests(3, stacked, ii, jj2, num_bins=60), f"two-field pairs binned cs={cs}")
    k = min(3, cs - 1)
    assert_close(engine.compute_requests(Measure.MUTUAL_INFORMATION_KRASKOV, pairs, k=k, query_from_secondary=True),
                 oracle.pair_requests(4, stacked, ii, jj2, k=k), f"two-field pairs kraskov cs={cs}")
    # without the flag the second field is not read
    assert_bit_exact(engine.compute_requests(Measure.KENDALL, pairs), oracle.pair_requests(2, ens, ii, jj), "one-field pairs")

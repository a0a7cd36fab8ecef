"""CPU: the oracle (oracle/corr_oracle.cpp) against the committed golden vectors.

Pearson / Spearman / Kendall expectations were produced by the REFERENCE's own object code (oracle/make_golden.py
through oracle/_ref) -- this is what pins the oracle.  The MI expectations are the restatement's own outputs
(regression pins only: "parity unpinned" for those two estimators, see the header of corr_oracle.cpp)."""
from pathlib import Path

import numpy as np
import pytest

import oracle_lib
from parity import assert_bit_exact

GOLDEN = Path(__file__).resolve().parent / "golden"
CASES = sorted(p.stem for p in GOLDEN.glob("*.npz") if p.stem not in ("known_answers", "pair_requests"))


def test_golden_cases_present():
    assert len(CASES) >= 8


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_outputs(oracle, case):
    d = np.load(GOLDEN / f"{case}.npz")
    ens, refv = d["members"], d["reference_values"]
    for name, m in (("pearson", oracle_lib.PEARSON), ("spearman", oracle_lib.SPEARMAN), ("kendall", oracle_lib.KENDALL)):
        assert_bit_exact(oracle.field(m, ens, refv), d[f"{name}__reference"], f"{case}/{name}")


@pytest.mark.parametrize("case", CASES)
def test_oracle_mi_regression(oracle, case):
    d = np.load(GOLDEN / f"{case}.npz")
    ens, refv = d["members"], d["reference_values"]
    mm = tuple(float(v) for v in d["minmax"])
    k = int(d["k"])
    cs = ens.shape[0]
    assert_bit_exact(oracle.field(oracle_lib.MI_BINNED, ens, refv, num_bins=80, minmax_ref=mm),
                     d["mi_binned__restatement"], f"{case}/mi_binned")
    assert_bit_exact(oracle.field(oracle_lib.BINNED_MI_CC, ens, refv, num_bins=80, minmax_ref=mm),
                     d["binned_mi_cc__restatement"], f"{case}/binned_mi_cc")
    assert_bit_exact(oracle.field(oracle_lib.MI_KRASKOV, ens, refv, k=k), d["mi_kraskov__restatement"],
                     f"{case}/mi_kraskov")
    assert_bit_exact(oracle.field(oracle_lib.MI_KRASKOV, ens, refv, k=min(3, max(cs - 1, 1))),
                     d["mi_kraskov_k3__restatement"], f"{case}/mi_kraskov k=3")
    assert_bit_exact(oracle.field(oracle_lib.MI_KRASKOV, ens, refv, k=k, estimator=2), d["mi_kraskov2__restatement"],
                     f"{case}/mi_kraskov2")
    assert_bit_exact(oracle.field(oracle_lib.KMI_CC, ens, refv, k=k), d["kmi_cc__restatement"], f"{case}/kmi_cc")


def test_known_answers(oracle):
    d = np.load(GOLDEN / "known_answers.npz")
    x, y = d["x"], d["y"]
    # the reference ignores joint ties: 20/24 through two sqrtf, not SciPy's 0.875
    assert np.float32(oracle.kendall(x, y)) == d["kendall__reference"] == np.float32(0.833333254)
    assert np.float32(oracle.pearson(x, y)) == d["pearson__reference"]
    np.testing.assert_array_equal(oracle.ranks(y), d["ranks_y__reference"])
    np.testing.assert_array_equal(oracle.ranks(y), [1.0, 2.5, 2.5, 5.0, 5.0, 5.0, 8.0, 7.0])


def test_edge_semantics_in_golden():
    d = np.load(GOLDEN / "nan_8x4x2_cs12.npz")
    idx = (1 * 4 + 2) * 8 + 3
    assert np.isnan(d["pearson__reference"][idx])        # by propagation
    assert np.isnan(d["spearman__reference"][idx]) and np.isnan(d["kendall__reference"][idx])
    assert np.isnan(d["mi_binned__restatement"][idx]) and np.isnan(d["mi_kraskov__restatement"][idx])
    one = np.load(GOLDEN / "single_member_8x8x4.npz")
    for key in one.files:
        if "__" in key:
            assert (one[key] == 1.0).all(), key            # cs == 1 -> 1.0 for every measure

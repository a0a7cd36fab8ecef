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
CASES = sorted(p.stem for p in GOLDEN.glob("*.npz") if p.stem not in ("known_answers", "pair_requests", "two_fields_and_siblings"))


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


def test_two_field_modes_and_siblings(oracle):
    """Symmetric Pearson / Spearman / Kendall expectations are the reference's primitives applied voxel by voxel; the
    rest pins the restatement."""
    d = np.load(GOLDEN / "two_fields_and_siblings.npz")
    fa, fb = d["field_a"], d["field_b"]
    mm_a, mm_b = tuple(map(float, d["minmax_a"])), tuple(map(float, d["minmax_b"]))
    for name, m in (("pearson", 0), ("spearman", 1), ("kendall", 2)):
        assert_bit_exact(oracle.symmetric_field(m, fa, fb), d[f"symmetric_{name}__reference"], f"symmetric {name}")
    assert_bit_exact(oracle.symmetric_field(3, fa, fb, num_bins=20, minmax_ref=mm_a, minmax_query=mm_b),
                     d["symmetric_mi_binned__restatement"], "symmetric binned")
    assert_bit_exact(oracle.symmetric_field(4, fa, fb, k=3), d["symmetric_mi_kraskov__restatement"], "symmetric kraskov")
    assert_bit_exact(oracle.ensemble_stat(0, fb), d["ensemble_mean__restatement"], "mean")
    assert_bit_exact(oracle.ensemble_stat(1, fb), d["ensemble_spread__restatement"], "spread")
    assert_bit_exact(oracle.set_predicate(0, 0.25, 8, 16, fa), d["set_predicate_gt__restatement"], "set predicate >")
    assert_bit_exact(oracle.set_predicate(3, -0.5, 12, 12, fa), d["set_predicate_le__restatement"], "set predicate <=")
    assert_bit_exact(oracle.dkl(0, fa, num_bins=16), d["dkl_binned__restatement"], "dkl binned")
    assert_bit_exact(oracle.dkl(1, fa, k=2), d["dkl_knn__restatement"], "dkl knn")
    np.testing.assert_array_equal(oracle.tile_field(fa[0]), d["tiled_member0__restatement"])

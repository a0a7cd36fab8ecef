"""GPU parity against the committed golden vectors (tests/golden, reference-object-code outputs for
Pearson/Spearman/Kendall; restatement outputs for the MI estimators)."""
from pathlib import Path

import numpy as np
import pytest

from correrender_amd import Measure
from parity import assert_bit_exact, assert_close

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"
CASES = sorted(p.stem for p in GOLDEN.glob("*.npz") if p.stem not in ("known_answers", "pair_requests", "two_fields_and_siblings"))


@pytest.mark.parametrize("case", CASES)
def test_gpu_matches_golden(engine, case):
    d = np.load(GOLDEN / f"{case}.npz")
    ens, refv = d["members"], d["reference_values"]
    cs, zs, ys, xs = ens.shape
    mm = tuple(float(v) for v in d["minmax"])
    k = int(d["k"])
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    run = lambda m, **kw: engine.compute(m, reference_values=refv, **kw)
    assert_bit_exact(run(Measure.PEARSON), d["pearson__reference"], f"{case}/pearson")
    assert_bit_exact(run(Measure.SPEARMAN), d["spearman__reference"], f"{case}/spearman")
    assert_bit_exact(run(Measure.KENDALL), d["kendall__reference"], f"{case}/kendall")
    bk = dict(num_bins=80, minmax_ref=mm, minmax_query=mm)
    assert_close(run(Measure.MUTUAL_INFORMATION_BINNED, **bk), d["mi_binned__restatement"], f"{case}/mi_binned")
    assert_close(run(Measure.BINNED_MI_CORRELATION_COEFFICIENT, **bk), d["binned_mi_cc__restatement"],
                 f"{case}/binned_mi_cc")
    assert_close(run(Measure.MUTUAL_INFORMATION_KRASKOV, k=k), d["mi_kraskov__restatement"], f"{case}/mi_kraskov")
    assert_close(run(Measure.MUTUAL_INFORMATION_KRASKOV, k=min(3, max(cs - 1, 1))), d["mi_kraskov_k3__restatement"],
                 f"{case}/mi_kraskov k=3")
    assert_close(run(Measure.MUTUAL_INFORMATION_KRASKOV, k=k, kraskov_estimator_index=2),
                 d["mi_kraskov2__restatement"], f"{case}/mi_kraskov2")
    assert_close(run(Measure.KMI_CORRELATION_COEFFICIENT, k=k), d["kmi_cc__restatement"], f"{case}/kmi_cc")


def test_gpu_two_field_modes_and_siblings(engine):
    import torch
    d = np.load(GOLDEN / "two_fields_and_siblings.npz")
    fa, fb = d["field_a"], d["field_b"]
    cs, zs, ys, xs = fa.shape
    mm_a, mm_b = tuple(map(float, d["minmax_a"])), tuple(map(float, d["minmax_b"]))
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(fa)
    engine.upload_secondary_members(fb)
    for name, m in (("pearson", Measure.PEARSON), ("spearman", Measure.SPEARMAN), ("kendall", Measure.KENDALL)):
        assert_bit_exact(engine.compute(m, symmetric=True), d[f"symmetric_{name}__reference"], f"symmetric {name}")
    assert_close(engine.compute(Measure.MUTUAL_INFORMATION_BINNED, symmetric=True, num_bins=20, minmax_ref=mm_a,
                                minmax_query=mm_b), d["symmetric_mi_binned__restatement"], "symmetric binned")
    assert_close(engine.compute(Measure.MUTUAL_INFORMATION_KRASKOV, symmetric=True, k=3),
                 d["symmetric_mi_kraskov__restatement"], "symmetric kraskov")
    assert_bit_exact(engine.set_predicate(">", 0.25, 8, 16), d["set_predicate_gt__restatement"], "set predicate >")
    assert_bit_exact(engine.set_predicate("<=", -0.5, 12, 12), d["set_predicate_le__restatement"], "set predicate <=")
    assert_close(engine.dkl("binned", num_bins=16), d["dkl_binned__restatement"], "dkl binned")
    assert_close(engine.dkl("knn", k=2), d["dkl_knn__restatement"], "dkl knn")
    lin = torch.from_numpy(fa[0].copy()).cuda()
    tiled = torch.empty(engine.tiled_element_count(), dtype=torch.float32, device="cuda")
    engine.tile_field_device(lin, tiled)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(tiled.cpu().numpy(), d["tiled_member0__restatement"])
    engine.upload_members(fb)
    assert_bit_exact(engine.ensemble_stat(0), d["ensemble_mean__restatement"], "mean")
    assert_bit_exact(engine.ensemble_stat(1), d["ensemble_spread__restatement"], "spread")

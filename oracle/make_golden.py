#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- small input/output vectors for the correlation-field path.

Run in the BUILD container only (needs oracle/_ref/libref_corr.so, i.e. /root/reference):  python oracle/make_golden.py
  * Pearson / Spearman / Kendall expected outputs come from the REFERENCE's own object code
    (src/Calculators/Correlation.cpp compiled where it lies, driven by oracle/ref_driver.cpp).
  * binned / Kraskov MI expected outputs come from this repo's CPU restatement (oracle/corr_oracle.cpp): the
    reference's MutualInformation.cpp cannot be built here (boost, sgl, glm absent) -- those files are labelled
    "restatement" and pin the restatement against regressions, not against the reference ("parity unpinned").
Inputs are regenerated from seeds by this script; the arrays are stored so that the fixtures are self-contained data.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib  # noqa: E402
from correrender_amd import synth  # noqa: E402

OUT = ROOT / "tests" / "golden"


def case_inputs():
    """name -> (members [cs, zs, ys, xs] float32, reference values [cs] float32)"""
    cases = {}
    for cs in (16, 64):
        ens = synth.box_ensemble(16, 16, 8, cs, seed=20260130 + cs)
        cases[f"box_16x16x8_cs{cs}_center"] = (ens, ens[:, 4, 8, 8].copy())
        cases[f"box_16x16x8_cs{cs}_inbox"] = (ens, ens[:, 4, 2, 2].copy())
    rng = np.random.default_rng(7)
    ties = np.round(rng.standard_normal((24, 4, 8, 8)) * 1.5).astype(np.float32)       # tie-heavy
    ties[:, 0, 0, 0] = 1.0
    ties[:, 0, 0, 1] = np.arange(24)
    cases["ties_8x8x4_cs24"] = (ties, ties[:, 2, 3, 4].copy())
    nan = rng.standard_normal((12, 2, 4, 8)).astype(np.float32)                      # a NaN-containing voxel
    nan[5, 1, 2, 3] = np.nan
    cases["nan_8x4x2_cs12"] = (nan, nan[:, 0, 0, 0].copy())
    one = synth.box_ensemble(8, 8, 4, 1, seed=3)                                       # cs == 1
    cases["single_member_8x8x4"] = (one, one[:, 2, 4, 4].copy())
    iid = synth.normal_ensemble(16, 8, 4, 64, seed=99)                                 # tie-free (Kraskov)
    cases["normal_16x8x4_cs64"] = (iid, iid[:, 2, 4, 8].copy())
    return cases


def main():
    if not oracle_lib.reference_available():
        raise SystemExit("oracle/_ref/libref_corr.so missing: run `make -C oracle` where /root/reference exists")
    ref = oracle_lib.load_reference()
    oracle = oracle_lib.load_oracle()
    OUT.mkdir(parents=True, exist_ok=True)
    for name, (ens, refv) in case_inputs().items():
        cs = ens.shape[0]
        mn, mx = oracle.minmax(ens) if np.isfinite(ens).all() else (float(np.nanmin(ens)), float(np.nanmax(ens)))
        k = max(-(-3 * cs // 100), 1)
        data = dict(members=ens, reference_values=refv, minmax=np.array([mn, mx], np.float32), k=np.int32(k),
                    num_bins=np.int32(80))
        data["pearson__reference"] = ref.field(0, ens, refv)
        data["spearman__reference"] = ref.field(1, ens, refv)
        data["kendall__reference"] = ref.field(2, ens, refv)
        data["mi_binned__restatement"] = oracle.field(oracle_lib.MI_BINNED, ens, refv, num_bins=80, minmax_ref=(mn, mx))
        data["binned_mi_cc__restatement"] = oracle.field(oracle_lib.BINNED_MI_CC, ens, refv, num_bins=80,
                                                         minmax_ref=(mn, mx))
        data["mi_kraskov__restatement"] = oracle.field(oracle_lib.MI_KRASKOV, ens, refv, k=k)
        data["mi_kraskov_k3__restatement"] = oracle.field(oracle_lib.MI_KRASKOV, ens, refv, k=min(3, max(cs - 1, 1)))
        data["mi_kraskov2__restatement"] = oracle.field(oracle_lib.MI_KRASKOV, ens, refv, k=k, estimator=2)
        data["kmi_cc__restatement"] = oracle.field(oracle_lib.KMI_CC, ens, refv, k=k)
        np.savez_compressed(OUT / f"{name}.npz", **data)
        print(f"{name}: cs={cs} voxels={ens[0].size} -> {(OUT / (name + '.npz')).stat().st_size} bytes")
    # pair-request path (HEBChartCorrelation.cpp:493-600): reference primitives for Pearson/Spearman/Kendall,
    # restatement for the MI estimators
    from test_pair_requests import _case
    ens, pairs, ii, jj = _case(32, 4242, n=300)
    np.savez_compressed(
        OUT / "pair_requests.npz", members=ens, pairs=pairs, idx_i=ii, idx_j=jj,
        pearson__reference=ref.pair_requests(0, ens, ii, jj), spearman__reference=ref.pair_requests(1, ens, ii, jj),
        kendall__reference=ref.pair_requests(2, ens, ii, jj),
        mi_binned__restatement=oracle.pair_requests(3, ens, ii, jj, num_bins=80),
        mi_kraskov__restatement=oracle.pair_requests(4, ens, ii, jj, k=3))
    print("pair_requests: 300 requests, cs=32")
    # two-field modes and sibling reductions.  Symmetric Pearson/Spearman/Kendall expectations
    # come from the reference's primitives applied voxel by voxel to (field 1, field 2); everything else from the
    # restatement (regression pins).
    rng = np.random.default_rng(4711)
    fa = rng.standard_normal((24, 4, 6, 8)).astype(np.float32)
    fb = (0.6 * fa + 0.8 * rng.standard_normal((24, 4, 6, 8))).astype(np.float32)
    fb[:, 0, 0, 0] = fa[:, 0, 0, 0]
    fa[:, 1, 1, 1] = np.round(fa[:, 1, 1, 1])          # ties on the reference side
    fb[3, 2, 2, 2] = np.nan
    n = fa[0].size
    flat_a, flat_b = fa.reshape(24, n), fb.reshape(24, n)
    sym = {m: np.empty(n, np.float32) for m in ("pearson", "spearman", "kendall")}
    for v in range(n):
        x, y = flat_a[:, v].copy(), flat_b[:, v].copy()
        if np.isnan(x).any() or np.isnan(y).any():
            for m in sym:
                sym[m][v] = np.nan
            continue
        sym["pearson"][v] = ref.pearson(x, y)
        sym["spearman"][v] = ref.pearson(ref.ranks(x), ref.ranks(y))
        sym["kendall"][v] = ref.kendall(x, y)
    mm_a, mm_b = oracle.minmax(fa), (float(np.nanmin(fb)), float(np.nanmax(fb)))
    np.savez_compressed(
        OUT / "two_fields_and_siblings.npz", field_a=fa, field_b=fb, minmax_a=np.array(mm_a, np.float32),
        minmax_b=np.array(mm_b, np.float32),
        symmetric_pearson__reference=sym["pearson"], symmetric_spearman__reference=sym["spearman"],
        symmetric_kendall__reference=sym["kendall"],
        symmetric_mi_binned__restatement=oracle.symmetric_field(3, fa, fb, num_bins=20, minmax_ref=mm_a, minmax_query=mm_b),
        symmetric_mi_kraskov__restatement=oracle.symmetric_field(4, fa, fb, k=3),
        ensemble_mean__restatement=oracle.ensemble_stat(0, fb), ensemble_spread__restatement=oracle.ensemble_stat(1, fb),
        set_predicate_gt__restatement=oracle.set_predicate(0, 0.25, 8, 16, fa),
        set_predicate_le__restatement=oracle.set_predicate(3, -0.5, 12, 12, fa),
        dkl_binned__restatement=oracle.dkl(0, fa, num_bins=16), dkl_knn__restatement=oracle.dkl(1, fa, k=2),
        tiled_member0__restatement=oracle.tile_field(fa[0]))
    print("two_fields_and_siblings: 192 voxels, cs=24")
    # known-answer vectors (computed by the reference object code)
    x = np.array([1, 1, 2, 2, 3, 3, 4, 4], np.float32)
    y = np.array([1, 2, 2, 3, 3, 3, 5, 4], np.float32)
    ka = dict(x=x, y=y, kendall__reference=np.float32(ref.kendall(x, y)),
              kendall_slow__reference=np.float32(ref.kendall_slow(x, y)),
              pearson__reference=np.float32(ref.pearson(x, y)), ranks_y__reference=ref.ranks(y))
    np.savez_compressed(OUT / "known_answers.npz", **ka)
    print("known_answers:", {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in ka.items() if "__" in k})


if __name__ == "__main__":
    main()

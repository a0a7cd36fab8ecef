"""Shared comparison helpers for the parity tests."""
import numpy as np


def bit_identical(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Elementwise: same bits, or both NaN (any payload)."""
    a = np.ascontiguousarray(a, np.float32).reshape(-1)
    b = np.ascontiguousarray(b, np.float32).reshape(-1)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def assert_bit_exact(got, want, what=""):
    same = bit_identical(got, want)
    if not same.all():
        bad = np.flatnonzero(~same)
        g = np.asarray(got, np.float32).reshape(-1)
        w = np.asarray(want, np.float32).reshape(-1)
        raise AssertionError(f"{what}: {bad.size}/{same.size} voxels differ; first at {bad[0]}: "
                             f"got {g[bad[0]]!r} want {w[bad[0]]!r}")


# Floating-point estimators (north_star): within 1e-5 relative.  A pure relative bound is ill-conditioned at
# correlations near 0 (the reference itself carries ~1e-7 absolute rounding error there), so the check is
# |a-b| <= RTOL*max(|a|,|b|) + ATOL with the absolute floor stated here.
RTOL = 1e-5
ATOL = 1e-6
# Above this magnitude the absolute floor must do no work: the bound there is the pure relative 1e-5 of the north star.
REL_ONLY_ABOVE = 1e-3

# (what, n, n_above, max relative error over |v| > REL_ONLY_ABOVE, max absolute error over the rest) per assert_close
# call; tests/conftest.py writes the collection to gpurun_out/parity_relative_errors.json at the end of a GPU session.
REPORT = []


def assert_close(got, want, what="", rtol=RTOL, atol=ATOL):
    g = np.asarray(got, np.float64).reshape(-1)
    w = np.asarray(want, np.float64).reshape(-1)
    both = np.isfinite(g) & np.isfinite(w)
    big = both & (np.maximum(np.abs(g), np.abs(w)) > REL_ONLY_ABOVE)
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.abs(g - w) / np.maximum(np.abs(g), np.abs(w))
    max_rel = float(rel[big].max()) if big.any() else 0.0
    small = both & ~big
    REPORT.append({"what": what, "voxels": int(g.size), "voxels_above_1e-3": int(big.sum()),
                   "max_rel_err_above_1e-3": max_rel,
                   "max_abs_err_below_1e-3": float(np.abs(g - w)[small].max()) if small.any() else 0.0})
    if max_rel > rtol:
        i = int(np.flatnonzero(big)[np.argmax(rel[big])])
        raise AssertionError(f"{what}: relative error {max_rel:.3e} > {rtol:g} at {i} (|v| > {REL_ONLY_ABOVE:g}, no "
                             f"absolute floor): got {g[i]!r} want {w[i]!r}")
    nan_ok = np.isnan(g) == np.isnan(w)
    inf_ok = np.where(np.isinf(w) | np.isinf(g), g == w, True)
    fin = np.isfinite(g) & np.isfinite(w)
    err = np.abs(g - w)
    tol = rtol * np.maximum(np.abs(g), np.abs(w)) + atol
    ok = nan_ok & inf_ok & (~fin | (err <= tol))
    if not ok.all():
        bad = np.flatnonzero(~ok)
        raise AssertionError(f"{what}: {bad.size}/{ok.size} voxels out of tolerance; first at {bad[0]}: "
                             f"got {g[bad[0]]!r} want {w[bad[0]]!r}")

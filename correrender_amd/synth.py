"""Synthetic input volumes (host side, numpy): the "box ensemble" of the reference's input recipe
``scripts/generate_synth_box_ensembles.py`` restated for arbitrary grids, plus plain i.i.d.-normal ensembles.

Recipe (script lines 57-61, 70-102, 113-136): lambda(x,y,z) = sum over 10 boxes of
peak(chebyshev_dist((x,y,z),(cx,cy,zs//2)) / (size/2)), peak(u) = 0 if u >= 1 else 1 - max(0, 2|u|-1)^2; box
centres/sizes in units of g (g = xs/8 here; the script uses xs=ys=128, zs=32, g=16); member c at voxel v is
lambda_v*s1[c] + (1-lambda_v)*N(0,1) with s1 = 2*linspace(0,1,cs)-1; stored float32 as (member, z, y, x).
Two defects of the script are NOT inherited: its lambda field starts from np.empty (uninitialised) -- zeros here --
and its normal draws are unseeded -- seeded here.  The device-side generator (crf_synth_box_member) follows the
same recipe with a counter-based hash RNG; host and device streams are different random numbers by design.
"""
from __future__ import annotations

import numpy as np

# (cx, cy, size) in units of g
BOXES = [(1.0, 1.0, 2.0), (7.0, 7.0, 2.0), (2.5, 0.5, 1.0), (2.5, 1.5, 1.0), (5.5, 6.5, 1.0), (5.5, 7.5, 1.0),
         (0.5, 2.5, 1.0), (1.5, 2.5, 1.0), (6.5, 5.5, 1.0), (7.5, 5.5, 1.0)]


def _peak(u: np.ndarray) -> np.ndarray:
    t = np.maximum(0.0, np.abs(u) * 2.0 - 1.0)
    return np.where(u >= 1.0, 0.0, 1.0 - t * t)


def box_lambda_field(xs: int, ys: int, zs: int, z_begin: int = 0, zs_global: int | None = None) -> np.ndarray:
    """lambda field of shape (zs, ys, xs) for the z-slab [z_begin, z_begin+zs) of a grid with zs_global slices."""
    zs_global = zs if zs_global is None else zs_global
    g = xs / 8.0
    z = (np.arange(zs, dtype=np.float32) + z_begin)[:, None, None]
    y = np.arange(ys, dtype=np.float32)[None, :, None]
    x = np.arange(xs, dtype=np.float32)[None, None, :]
    cz = float(zs_global // 2)
    lam = np.zeros((zs, ys, xs), dtype=np.float32)
    for cx, cy, size in BOXES:
        dist = np.maximum(np.abs(x - cx * g), np.maximum(np.abs(y - cy * g), np.abs(z - cz)))
        lam += _peak(dist / (size * g * 0.5)).astype(np.float32)
    return np.minimum(lam, 1.0)


def box_ensemble(xs: int, ys: int, zs: int, cs: int, seed: int = 20260130) -> np.ndarray:
    """float32 array (cs, zs, ys, xs)."""
    rng = np.random.default_rng(seed)
    lam = box_lambda_field(xs, ys, zs)
    s1 = (2.0 * np.linspace(0.0, 1.0, cs) - 1.0).astype(np.float32) if cs > 1 else np.array([-1.0], np.float32)
    out = np.empty((cs, zs, ys, xs), dtype=np.float32)
    for c in range(cs):
        noise = rng.standard_normal((zs, ys, xs), dtype=np.float32)
        out[c] = lam * s1[c] + (1.0 - lam) * noise
    return out


def normal_ensemble(xs: int, ys: int, zs: int, cs: int, seed: int = 1, rho_with_first_voxel: float = 0.0):
    """i.i.d. N(0,1) ensemble (tie-free with probability 1): the input for RNG-independent Kraskov parity."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((cs, zs, ys, xs), dtype=np.float32)

"""correrender_amd -- MI355X (gfx950) correlation-field engine.

One hot path of chrismile/Correrender, rebuilt natively for MI355X: the per-voxel ensemble correlation estimators
(Pearson, Spearman, Kendall, binned MI, Kraskov MI) between one reference grid point and every voxel of a 3-D grid
(reference: src/Calculators/CorrelationCalculator.cpp:781-1154).  The product is ``libcorrfield.so`` (hand-written HIP
kernels behind the C ABI of ``include/corrfield.h``); this package is the thin Python host layer over that ABI used
by the tests, the benchmark and the multi-GPU (one process per GPU) driver.  There is no CPU fallback.
"""
from ._lib import CorrFieldError, load_library, library_path  # noqa: F401
from .engine import CorrField, CorrFieldGroup, Measure, MEASURE_IDS, default_kraskov_k  # noqa: F401

__all__ = ["CorrField", "CorrFieldGroup", "Measure", "MEASURE_IDS", "CorrFieldError", "load_library", "library_path",
           "default_kraskov_k"]

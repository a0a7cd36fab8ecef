"""ctypes binding of libcorrfield.so (include/corrfield.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
_LIB = None


class CorrFieldError(RuntimeError):
    """A non-zero status from the C ABI (the reference reports the same conditions through
    sgl::Logfile::throwError, e.g. src/Volume/VolumeData.cpp:1217-1221)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libcorrfield error {code}: {message}")
        self.code = code
        self.message = message


class CrfParams(C.Structure):
    _fields_ = [
        ("measure", C.c_int32),
        ("ref_x", C.c_int32), ("ref_y", C.c_int32), ("ref_z", C.c_int32),
        ("k", C.c_int32),
        ("kraskov_estimator_index", C.c_int32),
        ("num_bins", C.c_int32),
        ("min_ref", C.c_float), ("max_ref", C.c_float),
        ("min_query", C.c_float), ("max_query", C.c_float),
        ("reference_values", C.POINTER(C.c_float)),
        ("flags", C.c_int32),
        ("prepared_slot", C.c_int32),
        ("reserved", C.c_int32 * 2),
    ]


class CrfRequest(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("xi", "yi", "zi", "i", "xj", "yj", "zj", "j")]


FLAG_ABSOLUTE_VALUE = 1
FLAG_SYMMETRIC = 2
FLAG_REFERENCE_FROM_SECONDARY = 4
FLAG_QUERY_FROM_SECONDARY = 8


# every symbol include/corrfield.h declares: name -> (restype, argtypes)
_VOIDP = C.c_void_p
SYMBOLS = {
    "crf_abi_version": (C.c_int, []),
    "crf_create": (C.c_int, [C.c_int, C.POINTER(_VOIDP)]),
    "crf_destroy": (None, [_VOIDP]),
    "crf_last_error": (C.c_char_p, [_VOIDP]),
    "crf_set_grid": (C.c_int, [_VOIDP, C.c_int, C.c_int, C.c_int, C.c_int]),
    "crf_upload_members": (C.c_int, [_VOIDP, C.POINTER(_VOIDP)]),
    "crf_bind_members_device": (C.c_int, [_VOIDP, C.POINTER(_VOIDP)]),
    "crf_member_minmax": (C.c_int, [_VOIDP, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "crf_member_minmax_divergent": (C.c_int, [_VOIDP, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "crf_upload_secondary_members": (C.c_int, [_VOIDP, C.POINTER(_VOIDP)]),
    "crf_bind_secondary_members_device": (C.c_int, [_VOIDP, C.POINTER(_VOIDP)]),
    "crf_secondary_member_minmax": (C.c_int, [_VOIDP, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "crf_gather_reference": (C.c_int, [_VOIDP, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "crf_gather_reference_device": (C.c_int, [_VOIDP, C.c_int, C.c_int, C.c_int, _VOIDP, _VOIDP]),
    "crf_gather_reference_rows_device": (C.c_int, [_VOIDP, C.POINTER(C.c_int32), C.c_int, _VOIDP, _VOIDP]),
    "crf_prepare_device": (C.c_int, [_VOIDP, C.POINTER(CrfParams), _VOIDP, C.c_int, _VOIDP]),
    "crf_prepare_rows_device": (C.c_int, [_VOIDP, C.POINTER(CrfParams), _VOIDP, C.c_int, C.c_int, _VOIDP]),
    "crf_compute_prepared_device": (C.c_int, [_VOIDP, C.POINTER(CrfParams), C.c_int, C.c_int, C.POINTER(_VOIDP), _VOIDP]),
    "crf_compute": (C.c_int, [_VOIDP, C.POINTER(CrfParams), C.POINTER(C.c_float)]),
    "crf_compute_device": (C.c_int, [_VOIDP, C.POINTER(CrfParams), _VOIDP, _VOIDP, _VOIDP]),
    "crf_compute_requests": (C.c_int, [_VOIDP, C.POINTER(CrfParams), _VOIDP, C.c_size_t, C.POINTER(C.c_float)]),
    "crf_compute_requests_device": (C.c_int, [_VOIDP, C.POINTER(CrfParams), _VOIDP, C.c_size_t, _VOIDP, _VOIDP]),
    "crf_compute_ensemble_stat": (C.c_int, [_VOIDP, C.c_int, C.POINTER(C.c_float)]),
    "crf_compute_ensemble_stat_device": (C.c_int, [_VOIDP, C.c_int, _VOIDP, _VOIDP]),
    "crf_compute_set_predicate": (C.c_int, [_VOIDP, C.c_int, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "crf_compute_set_predicate_device": (C.c_int, [_VOIDP, C.c_int, C.c_float, C.c_int, C.c_int, _VOIDP, _VOIDP]),
    "crf_compute_dkl": (C.c_int, [_VOIDP, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "crf_compute_dkl_device": (C.c_int, [_VOIDP, C.c_int, C.c_int, C.c_int, _VOIDP, _VOIDP]),
    "crf_max_mutual_information_kraskov": (C.c_double, [C.c_int, C.c_int]),
    "crf_tiled_element_count": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "crf_tile_field_device": (C.c_int, [_VOIDP, _VOIDP, _VOIDP, _VOIDP]),
    "crf_set_profiling": (C.c_int, [_VOIDP, C.c_int]),
    "crf_take_kernel_time": (C.c_int, [_VOIDP, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "crf_last_kernel_name": (C.c_char_p, [_VOIDP]),
    "crf_synth_box_member": (C.c_int, [_VOIDP, _VOIDP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_uint64, _VOIDP]),
    "crf_set_kraskov_noise": (C.c_int, [_VOIDP, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "crf_group_set_kraskov_noise": (C.c_int, [_VOIDP, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    # several devices behind one caller thread
    "crf_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_VOIDP)]),
    "crf_group_destroy": (None, [_VOIDP]),
    "crf_group_last_error": (C.c_char_p, [_VOIDP]),
    "crf_group_size": (C.c_int, [_VOIDP]),
    "crf_group_exchange": (C.c_char_p, [_VOIDP]),
    "crf_group_context": (_VOIDP, [_VOIDP, C.c_int]),
    "crf_group_set_grid": (C.c_int, [_VOIDP, C.c_int, C.c_int, C.c_int, C.c_int]),
    "crf_group_slab": (C.c_int, [_VOIDP, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "crf_group_upload_members": (C.c_int, [_VOIDP, C.POINTER(_VOIDP)]),
    "crf_group_upload_secondary_members": (C.c_int, [_VOIDP, C.POINTER(_VOIDP)]),
    "crf_group_member_minmax": (C.c_int, [_VOIDP, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "crf_group_secondary_member_minmax": (C.c_int, [_VOIDP, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "crf_group_compute": (C.c_int, [_VOIDP, C.POINTER(CrfParams), C.POINTER(C.c_float)]),
    "crf_group_compute_device": (C.c_int, [_VOIDP, C.POINTER(CrfParams), C.POINTER(_VOIDP)]),
    "crf_group_compute_batch": (C.c_int, [_VOIDP, C.POINTER(CrfParams), C.c_int, C.POINTER(_VOIDP)]),
    "crf_group_compute_batch_device": (C.c_int, [_VOIDP, C.POINTER(CrfParams), C.c_int, C.POINTER(_VOIDP)]),
    "crf_group_set_profiling": (C.c_int, [_VOIDP, C.c_int]),
    "crf_group_take_kernel_time": (C.c_int, [_VOIDP, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
}


def library_path() -> Path:
    return Path(os.environ.get("CORRFIELD_LIBRARY", _HERE / "libcorrfield.so"))


def load_library() -> C.CDLL:
    """Loads the in-tree libcorrfield.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not path.exists():
        raise FileNotFoundError(
            f"{path} not found: build it with `make -C correrender_amd/csrc` (or __graft_entry__.build()). "
            "correrender_amd has no CPU fallback.")
    # torch wheels bundle their own HIP runtime (libamdhip64); it must be the one the process loads FIRST, otherwise a
    # later `import torch` finds a second, already-initialised runtime and reports "No HIP GPUs are available".
    # torch is this package's device-memory / stream / torch.distributed plumbing, so import it up front.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(path))
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this table diverge
        fn.restype = restype
        fn.argtypes = argtypes
    _LIB = lib
    return lib

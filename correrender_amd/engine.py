"""Python host layer over the C ABI: one `CorrField` = one crf_context = one GPU (one process per GPU).

Mirrors the state a reference `CorrelationCalculator` holds (src/Calculators/CorrelationCalculator.hpp:200-213):
measure, reference point, k, estimator index, number of bins.  Tensors are torch CUDA tensors (device memory and
streams only -- torch is plumbing here) or numpy arrays on the host.
"""
from __future__ import annotations

import ctypes as C
import enum
import math
from typing import Optional, Sequence

import numpy as np

from ._lib import (FLAG_ABSOLUTE_VALUE, FLAG_QUERY_FROM_SECONDARY, FLAG_REFERENCE_FROM_SECONDARY, FLAG_SYMMETRIC, CorrFieldError, CrfParams,
                   load_library)


class Measure(enum.IntEnum):
    """enum class CorrelationMeasureType (src/Calculators/CorrelationDefines.hpp:41-45)."""
    PEARSON = 0
    SPEARMAN = 1
    KENDALL = 2
    MUTUAL_INFORMATION_BINNED = 3
    MUTUAL_INFORMATION_KRASKOV = 4
    BINNED_MI_CORRELATION_COEFFICIENT = 5
    KMI_CORRELATION_COEFFICIENT = 6


# CORRELATION_MEASURE_TYPE_IDS (CorrelationDefines.hpp:54-57)
MEASURE_IDS = ["pearson", "spearman", "kendall", "mi_binned", "mi_kraskov",
               "binned_mi_correlation_coefficient", "kmi_correlation_coefficient"]


def default_kraskov_k(cs: int) -> int:
    """k = max(iceil(3*cs, 100), 1) (CorrelationCalculator.cpp:592-598)."""
    return max(-(-3 * cs // 100), 1)


class CorrField:
    def __init__(self, device: int = 0):
        self._lib = load_library()
        ctx = C.c_void_p()
        rc = self._lib.crf_create(int(device), C.byref(ctx))
        if rc != 0:
            raise CorrFieldError(rc, (self._lib.crf_last_error(None) or b"").decode())
        self._ctx = ctx
        self.device = int(device)
        self.grid = None
        self.cs = 0
        self._keepalive = None

    # -- plumbing -----------------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            raise CorrFieldError(rc, (self._lib.crf_last_error(self._ctx) or b"").decode())

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.crf_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def num_voxels(self) -> int:
        xs, ys, zs = self.grid
        return xs * ys * zs

    # -- ensemble -----------------------------------------------------------------------------------------
    def set_grid(self, xs: int, ys: int, zs: int, cs: int):
        self._check(self._lib.crf_set_grid(self._ctx, xs, ys, zs, cs))
        self.grid = (xs, ys, zs)
        self.cs = cs
        self._keepalive = None

    def upload_members(self, members: Sequence[np.ndarray]):
        """members: cs host arrays of zs*ys*xs float32 (any shape, C-contiguous, x fastest)."""
        arrs = [np.ascontiguousarray(m, dtype=np.float32) for m in members]
        if len(arrs) != self.cs or any(a.size != self.num_voxels for a in arrs):
            raise ValueError("members do not match the grid declared with set_grid")
        ptrs = (C.c_void_p * self.cs)(*[a.ctypes.data for a in arrs])
        self._check(self._lib.crf_upload_members(self._ctx, ptrs))

    def bind_members(self, members):
        """members: cs torch CUDA float32 tensors (or one [cs, ...] tensor), each holding one volume; borrowed."""
        tensors = [members[i] for i in range(self.cs)]
        for t in tensors:
            if not t.is_cuda or not t.is_contiguous() or t.numel() != self.num_voxels or t.element_size() != 4:
                raise ValueError("each member must be a contiguous CUDA float32 tensor of xs*ys*zs elements")
        ptrs = (C.c_void_p * self.cs)(*[t.data_ptr() for t in tensors])
        self._check(self._lib.crf_bind_members_device(self._ctx, ptrs))
        self._keepalive = (members, tensors)

    def member_minmax(self):
        mn, mx = C.c_float(), C.c_float()
        self._check(self._lib.crf_member_minmax(self._ctx, C.byref(mn), C.byref(mx)))
        return mn.value, mx.value

    def set_kraskov_noise(self, ref_noise=None, query_noise=None):
        """Replaces the per-member tie-breaking noise tables of the Kraskov estimators (noise VALUES, i.e. u * 1e-10,
        cs doubles each) -- e.g. with the stream of the reference's own sgl::XorshiftRandomGenerator; None restores the
        library's documented default stream (crf_set_kraskov_noise)."""
        if ref_noise is None:
            self._check(self._lib.crf_set_kraskov_noise(self._ctx, None, None))
            return
        r = np.ascontiguousarray(ref_noise, np.float64)
        q = np.ascontiguousarray(query_noise, np.float64)
        if r.size != self.cs or q.size != self.cs:
            raise ValueError("noise tables must hold cs doubles each")
        self._check(self._lib.crf_set_kraskov_noise(self._ctx, r.ctypes.data_as(C.POINTER(C.c_double)),
                                                    q.ctypes.data_as(C.POINTER(C.c_double))))

    # -- secondary members: the second scalar field of the SEPARATE / SEPARATE_SYMMETRIC field modes ---------
    def upload_secondary_members(self, members: Sequence[np.ndarray]):
        arrs = [np.ascontiguousarray(m, dtype=np.float32) for m in members]
        if len(arrs) != self.cs or any(a.size != self.num_voxels for a in arrs):
            raise ValueError("secondary members do not match the grid declared with set_grid")
        ptrs = (C.c_void_p * self.cs)(*[a.ctypes.data for a in arrs])
        self._check(self._lib.crf_upload_secondary_members(self._ctx, ptrs))

    def bind_secondary_members(self, members):
        tensors = [members[i] for i in range(self.cs)]
        for t in tensors:
            if not t.is_cuda or not t.is_contiguous() or t.numel() != self.num_voxels or t.element_size() != 4:
                raise ValueError("each member must be a contiguous CUDA float32 tensor of xs*ys*zs elements")
        ptrs = (C.c_void_p * self.cs)(*[t.data_ptr() for t in tensors])
        self._check(self._lib.crf_bind_secondary_members_device(self._ctx, ptrs))
        self._keepalive_secondary = (members, tensors)

    def secondary_member_minmax(self):
        mn, mx = C.c_float(), C.c_float()
        self._check(self._lib.crf_secondary_member_minmax(self._ctx, C.byref(mn), C.byref(mx)))
        return mn.value, mx.value

    # -- reference vector ---------------------------------------------------------------------------------
    def gather_reference(self, x: int, y: int, z: int) -> np.ndarray:
        out = np.empty(self.cs, dtype=np.float32)
        self._check(self._lib.crf_gather_reference(self._ctx, x, y, z, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def gather_reference_device(self, x: int, y: int, z: int, out, stream: int = 0):
        self._check(self._lib.crf_gather_reference_device(self._ctx, x, y, z, C.c_void_p(out.data_ptr()),
                                                          C.c_void_p(stream)))

    def gather_reference_rows_device(self, points, out, stream: int = 0):
        """points: local (x, y, z) per row, or None for a row to be zero-filled; out: [len(points), cs] CUDA floats."""
        n = len(points)
        flat = (C.c_int32 * (3 * max(n, 1)))()
        for r, p in enumerate(points):
            flat[3 * r], flat[3 * r + 1], flat[3 * r + 2] = (0, 0, -1) if p is None else (int(p[0]), int(p[1]), int(p[2]))
        self._check(self._lib.crf_gather_reference_rows_device(self._ctx, flat, n, C.c_void_p(out.data_ptr()),
                                                               C.c_void_p(stream)))

    # -- evaluation ---------------------------------------------------------------------------------------
    def _params(self, measure, ref, k, kraskov_estimator_index, num_bins, minmax_ref, minmax_query,
                reference_values, flags=0):
        p = CrfParams()
        p.measure = int(measure)
        p.flags = int(flags)
        rx, ry, rz = ref if ref is not None else (0, 0, 0)
        p.ref_x, p.ref_y, p.ref_z = int(rx), int(ry), int(rz)
        p.k = int(k if k is not None else default_kraskov_k(self.cs))
        p.kraskov_estimator_index = int(kraskov_estimator_index)
        p.num_bins = int(num_bins)
        if minmax_ref is not None:
            p.min_ref, p.max_ref = float(minmax_ref[0]), float(minmax_ref[1])
        if minmax_query is not None:
            p.min_query, p.max_query = float(minmax_query[0]), float(minmax_query[1])
        keep = None
        if reference_values is not None:
            keep = np.ascontiguousarray(reference_values, dtype=np.float32)
            if keep.size != self.cs:
                raise ValueError("reference_values must hold cs floats")
            p.reference_values = keep.ctypes.data_as(C.POINTER(C.c_float))
        return p, keep

    def _binned_ranges(self, measure, minmax_ref, minmax_query, mode="single"):
        if int(measure) in (Measure.MUTUAL_INFORMATION_BINNED, Measure.BINNED_MI_CORRELATION_COEFFICIENT):
            if mode == "single":
                if minmax_ref is None:
                    minmax_ref = self.member_minmax()      # SINGLE mode: CorrelationCalculator.cpp:822-829
                if minmax_query is None:
                    minmax_query = minmax_ref              # :843-846
            elif mode == "separate":                       # reference field = secondary, query = primary (:820-842)
                if minmax_ref is None:
                    minmax_ref = self.secondary_member_minmax()
                if minmax_query is None:
                    minmax_query = self.member_minmax()
            else:                                          # symmetric: reference = primary, query = secondary
                if minmax_ref is None:
                    minmax_ref = self.member_minmax()
                if minmax_query is None:
                    minmax_query = self.secondary_member_minmax()
        return minmax_ref, minmax_query

    @staticmethod
    def _mode_flags(symmetric, reference_from_secondary, absolute_value=False):
        extra = FLAG_ABSOLUTE_VALUE if absolute_value else 0
        if symmetric:
            return FLAG_SYMMETRIC | extra, "symmetric"
        if reference_from_secondary:
            return FLAG_REFERENCE_FROM_SECONDARY | extra, "separate"
        return extra, "single"

    def compute(self, measure, ref=None, *, k=None, kraskov_estimator_index=1, num_bins=80, minmax_ref=None,
                minmax_query=None, reference_values=None, symmetric=False, reference_from_secondary=False,
                absolute_value=False, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Synchronous evaluation to a host array of shape (zs, ys, xs) -- calculateCpu(t, e, buffer).
        out: caller-owned C-contiguous float32 array of xs*ys*zs elements to write into (default: a new array).
        symmetric: SEPARATE_SYMMETRIC field mode (primary vs secondary members at every voxel);
        reference_from_secondary: SEPARATE field mode (reference vector = secondary members at `ref`);
        absolute_value: opt-in |.| of the field (the reference's CPU path never applies it)."""
        flags, mode = self._mode_flags(symmetric, reference_from_secondary, absolute_value)
        minmax_ref, minmax_query = self._binned_ranges(measure, minmax_ref, minmax_query, mode)
        p, keep = self._params(measure, ref, k, kraskov_estimator_index, num_bins, minmax_ref, minmax_query,
                               reference_values, flags)
        xs, ys, zs = self.grid
        if out is None:
            out = np.empty((zs, ys, xs), dtype=np.float32)
        elif out.dtype != np.float32 or out.size != self.num_voxels or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous float32 array of xs*ys*zs elements")
        self._check(self._lib.crf_compute(self._ctx, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_float))))
        del keep
        return out

    def compute_device(self, measure, out, ref=None, *, device_reference=None, stream: int = 0, k=None,
                       kraskov_estimator_index=1, num_bins=80, minmax_ref=None, minmax_query=None,
                       reference_values=None, symmetric=False, reference_from_secondary=False, prepared_slot=None,
                       absolute_value=False):
        """Asynchronous, stream-ordered evaluation into a CUDA float32 tensor `out` of xs*ys*zs elements.
        prepared_slot: use the reference-side tables prepare_device() left in that slot (no reference vector is read)."""
        flags, mode = self._mode_flags(symmetric, reference_from_secondary, absolute_value)
        minmax_ref, minmax_query = self._binned_ranges(measure, minmax_ref, minmax_query, mode)
        p, keep = self._params(measure, ref, k, kraskov_estimator_index, num_bins, minmax_ref, minmax_query,
                               reference_values, flags)
        if prepared_slot is not None:
            p.prepared_slot = int(prepared_slot) + 1
        if out.numel() != self.num_voxels or not out.is_cuda or not out.is_contiguous():
            raise ValueError("out must be a contiguous CUDA float32 tensor of xs*ys*zs elements")
        dref = C.c_void_p(device_reference.data_ptr()) if device_reference is not None else C.c_void_p(0)
        self._check(self._lib.crf_compute_device(self._ctx, C.byref(p), dref, C.c_void_p(out.data_ptr()),
                                                 C.c_void_p(stream)))
        if keep is not None:  # the H2D copy of a host reference vector is asynchronous: keep it alive
            self._keep_ref = keep
        return out

    def prepare_rows_device(self, measure, rows, first_slot: int, count: int, *, stream: int = 0, k=None,
                            kraskov_estimator_index=1, num_bins=80, minmax_ref=None, minmax_query=None):
        """crf_prepare_rows_device: rows[i] (a [>= count, cs] contiguous CUDA float32 tensor) prepared into slot
        first_slot + i, one call for the whole batch."""
        minmax_ref, minmax_query = self._binned_ranges(measure, minmax_ref, minmax_query, "single")
        p, _ = self._params(measure, None, k, kraskov_estimator_index, num_bins, minmax_ref, minmax_query, None, 0)
        if not rows.is_cuda or not rows.is_contiguous() or rows.dim() != 2 or rows.shape[1] != self.cs or rows.shape[0] < count:
            raise ValueError("rows must be a contiguous [>= count, cs] CUDA float32 tensor")
        self._check(self._lib.crf_prepare_rows_device(self._ctx, C.byref(p), C.c_void_p(rows.data_ptr()), int(first_slot),
                                                      int(count), C.c_void_p(stream)))

    def compute_prepared_device(self, measure, outs, first_slot: int, *, stream: int = 0, k=None,
                                kraskov_estimator_index=1, num_bins=80, minmax_ref=None, minmax_query=None):
        """crf_compute_prepared_device: len(outs) prepared evaluations (slots first_slot ...) launched back to back with
        one call; outs[i] receives evaluation i (the same tensor may be given several times)."""
        minmax_ref, minmax_query = self._binned_ranges(measure, minmax_ref, minmax_query, "single")
        p, _ = self._params(measure, None, k, kraskov_estimator_index, num_bins, minmax_ref, minmax_query, None, 0)
        for t in outs:
            if t.numel() != self.num_voxels or not t.is_cuda or not t.is_contiguous():
                raise ValueError("every out must be a contiguous CUDA float32 tensor of xs*ys*zs elements")
        ptrs = (C.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
        self._check(self._lib.crf_compute_prepared_device(self._ctx, C.byref(p), int(first_slot), len(outs), ptrs,
                                                          C.c_void_p(stream)))
        return outs

    PREPARED_SLOTS = 64   # CRF_PREPARED_SLOTS

    def prepare_device(self, measure, slot: int, ref=None, *, device_reference=None, stream: int = 0, k=None,
                       kraskov_estimator_index=1, num_bins=80, minmax_ref=None, minmax_query=None,
                       reference_values=None, reference_from_secondary=False):
        """Reference-side preparation only (crf_prepare_device): fills slot `slot` for a later
        compute_device(..., prepared_slot=slot) with the same measure and parameters."""
        flags, mode = self._mode_flags(False, reference_from_secondary)
        minmax_ref, minmax_query = self._binned_ranges(measure, minmax_ref, minmax_query, mode)
        p, keep = self._params(measure, ref, k, kraskov_estimator_index, num_bins, minmax_ref, minmax_query,
                               reference_values, flags)
        dref = C.c_void_p(device_reference.data_ptr()) if device_reference is not None else C.c_void_p(0)
        self._check(self._lib.crf_prepare_device(self._ctx, C.byref(p), dref, int(slot), C.c_void_p(stream)))
        if keep is not None:
            self._keep_ref = keep

    def compute_requests(self, measure, pairs, *, k=None, num_bins=80, absolute_value=False,
                         query_from_secondary=False) -> np.ndarray:
        """Pair-request mode (the reference's CorrelationComputePass request mode / HEBChart::computeCorrelations):
        `pairs` is an [n, 6] integer array of (xi, yi, zi, xj, yj, zj); returns n floats.  With
        `query_from_secondary` the j side reads the secondary member set (two-field request mode)."""
        pairs = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 6)
        xs, ys, _ = self.grid
        req = np.zeros((pairs.shape[0], 8), dtype=np.uint32)
        req[:, 0:3] = pairs[:, 0:3]
        req[:, 4:7] = pairs[:, 3:6]
        req[:, 3] = (pairs[:, 2] * ys + pairs[:, 1]) * xs + pairs[:, 0]
        req[:, 7] = (pairs[:, 5] * ys + pairs[:, 4]) * xs + pairs[:, 3]
        p, _ = self._params(measure, None, k, 1, num_bins, None, None, None)
        p.flags = (FLAG_ABSOLUTE_VALUE if absolute_value else 0) | (FLAG_QUERY_FROM_SECONDARY if query_from_secondary else 0)
        out = np.empty(req.shape[0], dtype=np.float32)
        self._check(self._lib.crf_compute_requests(self._ctx, C.byref(p), C.c_void_p(req.ctypes.data), req.shape[0],
                                                   out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def ensemble_stat(self, stat: int) -> np.ndarray:
        """stat 0: NaN-skipping ensemble mean, 1: ensemble spread (sample std-dev); shape (zs, ys, xs)."""
        xs, ys, zs = self.grid
        out = np.empty((zs, ys, xs), dtype=np.float32)
        self._check(self._lib.crf_compute_ensemble_stat(self._ctx, int(stat), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def ensemble_stat_device(self, stat: int, out, stream: int = 0):
        self._check(self._lib.crf_compute_ensemble_stat_device(self._ctx, int(stat), C.c_void_p(out.data_ptr()),
                                                               C.c_void_p(stream)))
        return out

    COMPARISON_OPERATORS = (">", ">=", "<", "<=", "==", "!=")   # COMPARISON_OPERATOR_NAMES, SetPredicateCalculator.hpp:44-46

    def set_predicate(self, op, comparison_value: float, count_lower: int, count_upper: int) -> np.ndarray:
        """SetPredicateCalculator::calculateCpu: fraction-of-members predicate field; `op` is an operator string or
        its ComparisonOperatorType index; shape (zs, ys, xs)."""
        op = self.COMPARISON_OPERATORS.index(op) if isinstance(op, str) else int(op)
        xs, ys, zs = self.grid
        out = np.empty((zs, ys, xs), dtype=np.float32)
        self._check(self._lib.crf_compute_set_predicate(self._ctx, op, C.c_float(comparison_value), int(count_lower),
                                                        int(count_upper), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def set_predicate_device(self, op, comparison_value, count_lower, count_upper, out, stream: int = 0):
        op = self.COMPARISON_OPERATORS.index(op) if isinstance(op, str) else int(op)
        self._check(self._lib.crf_compute_set_predicate_device(
            self._ctx, op, C.c_float(comparison_value), int(count_lower), int(count_upper), C.c_void_p(out.data_ptr()),
            C.c_void_p(stream)))
        return out

    def dkl(self, estimator, *, num_bins: int = 80, k: Optional[int] = None) -> np.ndarray:
        """DKLCalculator::calculateCpu: KL divergence of the normalised ensemble distribution from N(0,1);
        estimator "binned"/0 or "knn"/1; shape (zs, ys, xs)."""
        est = {"binned": 0, "knn": 1}.get(estimator, estimator)
        xs, ys, zs = self.grid
        out = np.empty((zs, ys, xs), dtype=np.float32)
        kk = int(k if k is not None else default_kraskov_k(self.cs))   # same default formula, DKLCalculator.cpp:94-101
        self._check(self._lib.crf_compute_dkl(self._ctx, int(est), int(num_bins), kk,
                                              out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def dkl_device(self, estimator, out, *, num_bins: int = 80, k: Optional[int] = None, stream: int = 0):
        est = {"binned": 0, "knn": 1}.get(estimator, estimator)
        kk = int(k if k is not None else default_kraskov_k(self.cs))
        self._check(self._lib.crf_compute_dkl_device(self._ctx, int(est), int(num_bins), kk,
                                                     C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
        return out

    def tiled_element_count(self) -> int:
        return int(self._lib.crf_tiled_element_count(*self.grid))

    def tile_field_device(self, linear, tiled, stream: int = 0):
        """Linear (IDXS) CUDA field -> the reference's 8x8x4-tiled device layout (VolumeData.cpp:1581-1621)."""
        if linear.numel() != self.num_voxels or tiled.numel() != self.tiled_element_count():
            raise ValueError("tile_field_device: buffer sizes do not match the grid")
        self._check(self._lib.crf_tile_field_device(self._ctx, C.c_void_p(linear.data_ptr()),
                                                    C.c_void_p(tiled.data_ptr()), C.c_void_p(stream)))
        return tiled

    # -- instrumentation ----------------------------------------------------------------------------------
    def set_profiling(self, enabled: bool):
        self._check(self._lib.crf_set_profiling(self._ctx, 1 if enabled else 0))

    def take_kernel_time(self):
        ms, n = C.c_double(), C.c_int()
        self._check(self._lib.crf_take_kernel_time(self._ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_kernel_name(self) -> str:
        return (self._lib.crf_last_kernel_name(self._ctx) or b"").decode()

    def synth_box_member(self, out, xs, ys, zs_local, z_begin, zs_global, c, cs, seed, stream: int = 0):
        self._check(self._lib.crf_synth_box_member(self._ctx, C.c_void_p(out.data_ptr()), xs, ys, zs_local, z_begin,
                                                   zs_global, c, cs, C.c_uint64(seed), C.c_void_p(stream)))


class CorrFieldGroup:
    """Several GPUs behind one caller thread (crf_group_*): the whole grid in, the whole field out; the z-slab split,
    the per-device worker threads and the reference-vector exchange (RCCL broadcast, or a peer copy when a device
    ordinal repeats) live inside libcorrfield.  Mirrors CorrField.compute for host arrays."""

    def __init__(self, devices: Sequence[int]):
        self._lib = load_library()
        ords = (C.c_int * len(devices))(*[int(d) for d in devices])
        g = C.c_void_p()
        rc = self._lib.crf_group_create(ords, len(devices), C.byref(g))
        if rc != 0:
            raise CorrFieldError(rc, (self._lib.crf_group_last_error(None) or b"").decode())
        self._g = g
        self.devices = [int(d) for d in devices]
        self.grid = None
        self.cs = 0

    def _check(self, rc: int):
        if rc != 0:
            raise CorrFieldError(rc, (self._lib.crf_group_last_error(self._g) or b"").decode())

    def close(self):
        if getattr(self, "_g", None):
            self._lib.crf_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def exchange(self) -> str:
        return (self._lib.crf_group_exchange(self._g) or b"").decode()

    def set_grid(self, xs: int, ys: int, zs: int, cs: int):
        self._check(self._lib.crf_group_set_grid(self._g, xs, ys, zs, cs))
        self.grid, self.cs = (xs, ys, zs), cs

    def slab(self, slot: int):
        z0, zn = C.c_int(), C.c_int()
        self._check(self._lib.crf_group_slab(self._g, slot, C.byref(z0), C.byref(zn)))
        return z0.value, zn.value

    def _member_ptrs(self, members):
        xs, ys, zs = self.grid
        arrs = [np.ascontiguousarray(m, dtype=np.float32) for m in members]
        if len(arrs) != self.cs or any(a.size != xs * ys * zs for a in arrs):
            raise ValueError("members do not match the grid declared with set_grid")
        return arrs, (C.c_void_p * self.cs)(*[a.ctypes.data for a in arrs])

    def upload_members(self, members):
        arrs, ptrs = self._member_ptrs(members)
        self._check(self._lib.crf_group_upload_members(self._g, ptrs))

    def upload_secondary_members(self, members):
        arrs, ptrs = self._member_ptrs(members)
        self._check(self._lib.crf_group_upload_secondary_members(self._g, ptrs))

    def member_minmax(self):
        mn, mx = C.c_float(), C.c_float()
        self._check(self._lib.crf_group_member_minmax(self._g, C.byref(mn), C.byref(mx)))
        return mn.value, mx.value

    def secondary_member_minmax(self):
        mn, mx = C.c_float(), C.c_float()
        self._check(self._lib.crf_group_secondary_member_minmax(self._g, C.byref(mn), C.byref(mx)))
        return mn.value, mx.value

    def compute(self, measure, ref=None, *, k=None, kraskov_estimator_index=1, num_bins=80, minmax_ref=None,
                minmax_query=None, reference_values=None, symmetric=False, reference_from_secondary=False,
                absolute_value=False, out: Optional[np.ndarray] = None) -> np.ndarray:
        """calculateCpu(t, e, buffer) on all devices of the group; `ref` in GLOBAL grid coordinates."""
        flags, mode = CorrField._mode_flags(symmetric, reference_from_secondary, absolute_value)
        minmax_ref, minmax_query = CorrField._binned_ranges(self, measure, minmax_ref, minmax_query, mode)
        p, keep = CorrField._params(self, measure, ref, k, kraskov_estimator_index, num_bins, minmax_ref, minmax_query,
                                    reference_values, flags)
        xs, ys, zs = self.grid
        if out is None:
            out = np.empty((zs, ys, xs), dtype=np.float32)
        elif out.dtype != np.float32 or out.size != xs * ys * zs or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous float32 array of xs*ys*zs elements")
        self._check(self._lib.crf_group_compute(self._g, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_float))))
        del keep
        return out

    def compute_device(self, measure, outs, ref=None, *, k=None, kraskov_estimator_index=1, num_bins=80, minmax_ref=None,
                       minmax_query=None, reference_values=None, symmetric=False, reference_from_secondary=False,
                       absolute_value=False):
        """crf_group_compute_device: outs[slot] is a CUDA float32 tensor on the slot's device holding xs*ys*z_count
        elements (slab(slot)); returns when every device has finished."""
        flags, mode = CorrField._mode_flags(symmetric, reference_from_secondary, absolute_value)
        minmax_ref, minmax_query = CorrField._binned_ranges(self, measure, minmax_ref, minmax_query, mode)
        p, keep = CorrField._params(self, measure, ref, k, kraskov_estimator_index, num_bins, minmax_ref, minmax_query,
                                    reference_values, flags)
        xs, ys, _ = self.grid
        if len(outs) != len(self.devices):
            raise ValueError("one output tensor per device slot")
        for slot, t in enumerate(outs):
            if not t.is_cuda or not t.is_contiguous() or t.element_size() != 4 or t.numel() != xs * ys * self.slab(slot)[1]:
                raise ValueError(f"output of slot {slot} must be a contiguous CUDA float32 tensor of the slab's size")
        ptrs = (C.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
        self._check(self._lib.crf_group_compute_device(self._g, C.byref(p), ptrs))
        del keep
        return outs

    def _batch_params(self, measure, refs, kw):
        """crf_params array for a list of reference points (GLOBAL coordinates), all with the same settings."""
        if not refs:
            raise ValueError("empty batch")
        mode_kw = {k: kw.pop(k, False) for k in ("symmetric", "reference_from_secondary", "absolute_value")}
        flags, mode = CorrField._mode_flags(**mode_kw)
        minmax_ref, minmax_query = CorrField._binned_ranges(self, measure, kw.pop("minmax_ref", None),
                                                            kw.pop("minmax_query", None), mode)
        arr = (CrfParams * len(refs))()
        keep = []
        for i, ref in enumerate(refs):
            p, k = CorrField._params(self, measure, ref, kw.get("k"), kw.get("kraskov_estimator_index", 1),
                                     kw.get("num_bins", 80), minmax_ref, minmax_query, None, flags)
            arr[i] = p
            keep.append(k)
        return arr, keep

    def compute_batch(self, measure, refs, *, outs=None, **kw):
        """crf_group_compute_batch: one hand-off for a list of reference points; returns a list of (zs, ys, xs) arrays."""
        arr, keep = self._batch_params(measure, list(refs), kw)
        xs, ys, zs = self.grid
        if outs is None:
            outs = [np.empty((zs, ys, xs), dtype=np.float32) for _ in range(len(arr))]
        if len(outs) != len(arr) or any(o.dtype != np.float32 or o.size != xs * ys * zs or not o.flags["C_CONTIGUOUS"]
                                        for o in outs):
            raise ValueError("outs: one C-contiguous float32 array of xs*ys*zs elements per reference point")
        ptrs = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
        self._check(self._lib.crf_group_compute_batch(self._g, arr, len(arr), ptrs))
        del keep
        return outs

    def compute_batch_device(self, measure, refs, outs, **kw):
        """crf_group_compute_batch_device: outs[i][slot] is a CUDA float32 tensor on the slot's device holding the slab
        of evaluation i; returns when every device has finished the whole list."""
        arr, keep = self._batch_params(measure, list(refs), kw)
        n = len(self.devices)
        if len(outs) != len(arr) or any(len(row) != n for row in outs):
            raise ValueError("outs: one row of per-slot tensors per reference point")
        xs, ys, _ = self.grid
        for row in outs:
            for slot, t in enumerate(row):
                if not t.is_cuda or not t.is_contiguous() or t.element_size() != 4 or t.numel() != xs * ys * self.slab(slot)[1]:
                    raise ValueError(f"output of slot {slot} must be a contiguous CUDA float32 tensor of the slab's size")
        ptrs = (C.c_void_p * (len(arr) * n))(*[t.data_ptr() for row in outs for t in row])
        self._check(self._lib.crf_group_compute_batch_device(self._g, arr, len(arr), ptrs))
        del keep
        return outs

    def set_profiling(self, enabled: bool):
        self._check(self._lib.crf_group_set_profiling(self._g, 1 if enabled else 0))

    def take_kernel_time(self):
        ms, n = C.c_double(), C.c_int()
        self._check(self._lib.crf_group_take_kernel_time(self._g, C.byref(ms), C.byref(n)))
        return ms.value, n.value

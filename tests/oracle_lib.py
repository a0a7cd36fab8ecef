"""ctypes access to the CHECKER libraries (test infrastructure only):
   oracle/liboracle.so        this repo's CPU restatement (built on demand with g++; travels to the GPU box prebuilt)
   oracle/_ref/libref_corr.so the reference's own Correlation.cpp behind oracle/ref_driver.cpp (build container only)
Nothing under correrender_amd/ imports this module."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
FP = C.POINTER(C.c_float)

PEARSON, SPEARMAN, KENDALL, MI_BINNED, MI_KRASKOV, BINNED_MI_CC, KMI_CC = range(7)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(FP)


class Oracle:
    def __init__(self, lib: C.CDLL):
        self.lib = lib
        lib.oracle_pearson2.restype = C.c_float
        lib.oracle_pearson2.argtypes = [FP, FP, C.c_int]
        lib.oracle_ranks.restype = None
        lib.oracle_ranks.argtypes = [FP, FP, C.c_int]
        lib.oracle_spearman.restype = C.c_float
        lib.oracle_spearman.argtypes = [FP, FP, C.c_int]
        lib.oracle_kendall.restype = C.c_float
        lib.oracle_kendall.argtypes = [FP, FP, C.c_int]
        lib.oracle_mi_binned.restype = C.c_float
        lib.oracle_mi_binned.argtypes = [FP, FP, C.c_int, C.c_int]
        lib.oracle_mi_kraskov.restype = C.c_float
        lib.oracle_mi_kraskov.argtypes = [FP, FP, C.c_int, C.c_int, C.c_int]
        lib.oracle_digamma_int.restype = C.c_double
        lib.oracle_digamma_int.argtypes = [C.c_int]
        lib.oracle_set_kraskov_noise.restype = None
        lib.oracle_set_kraskov_noise.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
        lib.oracle_noise01.restype = None
        lib.oracle_noise01.argtypes = [C.c_int, C.c_int, FP]
        lib.oracle_minmax.restype = None
        lib.oracle_minmax.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_size_t, FP, FP]
        lib.oracle_correlation_field.restype = C.c_int
        lib.oracle_correlation_field.argtypes = [
            C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_size_t, FP, C.c_int, C.c_int, C.c_int,
            C.c_float, C.c_float, C.c_float, C.c_float, FP, C.c_int]
        lib.oracle_symmetric_field.restype = C.c_int
        lib.oracle_symmetric_field.argtypes = [
            C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
            C.c_float, C.c_float, C.c_float, C.c_float, FP]
        lib.oracle_set_predicate.restype = C.c_int
        lib.oracle_set_predicate.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int,
                                             C.c_size_t, FP]
        lib.oracle_tile_field.restype = None
        lib.oracle_tile_field.argtypes = [FP, C.c_int, C.c_int, C.c_int, FP]
        lib.oracle_dkl_field.restype = C.c_int
        lib.oracle_dkl_field.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_int, C.c_int, FP]
        lib.oracle_max_threads.restype = C.c_int
        lib.oracle_ensemble_stat.restype = C.c_int
        lib.oracle_ensemble_stat.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_size_t, FP]
        lib.oracle_pair_requests.restype = C.c_int
        lib.oracle_pair_requests.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_size_t),
                                             C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_int, C.c_int, FP]

    # -- primitives
    def pearson(self, x, y):
        x, y = _f32(x), _f32(y)
        return float(self.lib.oracle_pearson2(_fp(x), _fp(y), x.size))

    def ranks(self, v):
        v = _f32(v)
        out = np.empty_like(v)
        self.lib.oracle_ranks(_fp(v), _fp(out), v.size)
        return out

    def spearman(self, x, y):
        x, y = _f32(x), _f32(y)
        return float(self.lib.oracle_spearman(_fp(x), _fp(y), x.size))

    def kendall(self, x, y):
        x, y = _f32(x), _f32(y)
        return float(self.lib.oracle_kendall(_fp(x), _fp(y), x.size))

    def mi_binned(self, x01, y01, num_bins):
        x01, y01 = _f32(x01), _f32(y01)
        return float(self.lib.oracle_mi_binned(_fp(x01), _fp(y01), num_bins, x01.size))

    def mi_kraskov(self, x, y, k, estimator=1):
        x, y = _f32(x), _f32(y)
        return float(self.lib.oracle_mi_kraskov(_fp(x), _fp(y), k, x.size, estimator))

    def digamma(self, n):
        return float(self.lib.oracle_digamma_int(int(n)))

    def set_kraskov_noise(self, ref_noise=None, query_noise=None):
        """Per-member noise tables (already scaled by 1e-10) for the Kraskov estimators; None restores the default."""
        if ref_noise is None:
            self.lib.oracle_set_kraskov_noise(None, None, 0)
            return
        r = np.ascontiguousarray(ref_noise, np.float64)
        q = np.ascontiguousarray(query_noise, np.float64)
        assert r.size == q.size
        self.lib.oracle_set_kraskov_noise(r.ctypes.data_as(C.POINTER(C.c_double)), q.ctypes.data_as(C.POINTER(C.c_double)),
                                          r.size)

    def noise01(self, which, n):
        out = np.empty(n, np.float32)
        self.lib.oracle_noise01(which, n, _fp(out))
        return out

    def minmax(self, members):
        members = _members(members)
        ptrs = (C.c_void_p * len(members))(*[m.ctypes.data for m in members])
        mn, mx = C.c_float(), C.c_float()
        self.lib.oracle_minmax(ptrs, len(members), members[0].size, C.byref(mn), C.byref(mx))
        return mn.value, mx.value

    # -- the calculateCpu driver
    def field(self, measure, members, ref_values, *, k=3, estimator=1, num_bins=80, minmax_ref=(0.0, 1.0),
              minmax_query=None, voxel_range=None, threads=0):
        members = _members(members)
        cs = len(members)
        n = members[0].size
        lo, hi = voxel_range if voxel_range is not None else (0, n)
        ref_values = _f32(ref_values)
        assert ref_values.size == cs
        minmax_query = minmax_ref if minmax_query is None else minmax_query
        ptrs = (C.c_void_p * cs)(*[m.ctypes.data for m in members])
        out = np.empty(hi - lo, np.float32)
        rc = self.lib.oracle_correlation_field(
            int(measure), ptrs, cs, lo, hi, _fp(ref_values), int(k), int(estimator), int(num_bins),
            float(minmax_ref[0]), float(minmax_ref[1]), float(minmax_query[0]), float(minmax_query[1]), _fp(out),
            int(threads))
        assert rc == 0
        return out

    def symmetric_field(self, measure, members_ref, members_query, *, k=3, num_bins=80, minmax_ref=(0.0, 1.0),
                        minmax_query=(0.0, 1.0), voxel_range=None):
        """SEPARATE_SYMMETRIC: measure(members_ref[:, v], members_query[:, v]) at every voxel v."""
        mr, mq = _members(members_ref), _members(members_query)
        cs = len(mr)
        assert len(mq) == cs
        lo, hi = voxel_range if voxel_range is not None else (0, mr[0].size)
        pr = (C.c_void_p * cs)(*[m.ctypes.data for m in mr])
        pq = (C.c_void_p * cs)(*[m.ctypes.data for m in mq])
        out = np.empty(hi - lo, np.float32)
        rc = self.lib.oracle_symmetric_field(int(measure), pr, pq, cs, lo, hi, int(k), int(num_bins),
                                             float(minmax_ref[0]), float(minmax_ref[1]), float(minmax_query[0]),
                                             float(minmax_query[1]), _fp(out))
        assert rc == 0
        return out

    def set_predicate(self, op, comparison_value, count_lower, count_upper, members):
        members = _members(members)
        ptrs = (C.c_void_p * len(members))(*[m.ctypes.data for m in members])
        out = np.empty(members[0].size, np.float32)
        assert self.lib.oracle_set_predicate(int(op), float(comparison_value), int(count_lower), int(count_upper), ptrs,
                                             len(members), members[0].size, _fp(out)) == 0
        return out

    def tile_field(self, linear):
        zs, ys, xs = linear.shape
        lin = _f32(linear)
        out = np.empty(((xs + 7) // 8) * ((ys + 7) // 8) * ((zs + 3) // 4) * 256, np.float32)
        self.lib.oracle_tile_field(_fp(lin), xs, ys, zs, _fp(out))
        return out

    def dkl(self, estimator, members, *, num_bins=80, k=3):
        """DKLCalculator::calculateCpu: estimator 0 = binned, 1 = entropy k-NN."""
        members = _members(members)
        ptrs = (C.c_void_p * len(members))(*[m.ctypes.data for m in members])
        out = np.empty(members[0].size, np.float32)
        assert self.lib.oracle_dkl_field(int(estimator), ptrs, len(members), members[0].size, int(num_bins), int(k),
                                         _fp(out)) == 0
        return out

    def max_threads(self):
        return int(self.lib.oracle_max_threads())

    def first_touch_copy(self, members):
        """A copy of [cs, ...] float32 volumes whose pages were first touched by the OpenMP threads that will read them
        (same static partition over voxels as the field loops): NUMA placement for a bound cpu_baseline run."""
        members = np.ascontiguousarray(members, np.float32)
        cs = members.shape[0]
        n = members[0].size
        out = np.empty(members.shape, np.float32)          # untouched pages (large allocation: fresh mmap)
        src = (C.c_void_p * cs)(*[members[c].ctypes.data for c in range(cs)])
        dst = (C.c_void_p * cs)(*[out[c].ctypes.data for c in range(cs)])
        self.lib.oracle_first_touch_copy.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int64]
        assert self.lib.oracle_first_touch_copy(src, dst, cs, n) == 0
        return out

    def ensemble_stat(self, kind, members):
        members = _members(members)
        ptrs = (C.c_void_p * len(members))(*[m.ctypes.data for m in members])
        out = np.empty(members[0].size, np.float32)
        assert self.lib.oracle_ensemble_stat(int(kind), ptrs, len(members), members[0].size, _fp(out)) == 0
        return out

    def pair_requests(self, measure, members, idx_i, idx_j, *, k=3, num_bins=80, use_abs=False):
        members = _members(members)
        cs = len(members)
        ii = np.ascontiguousarray(idx_i, dtype=np.uint64)
        jj = np.ascontiguousarray(idx_j, dtype=np.uint64)
        ptrs = (C.c_void_p * cs)(*[m.ctypes.data for m in members])
        out = np.empty(ii.size, np.float32)
        rc = self.lib.oracle_pair_requests(int(measure), ptrs, cs, ii.ctypes.data_as(C.POINTER(C.c_size_t)),
                                           jj.ctypes.data_as(C.POINTER(C.c_size_t)), ii.size, int(k), int(num_bins),
                                           1 if use_abs else 0, _fp(out))
        assert rc == 0
        return out


class Reference:
    """The reference's own object code (Pearson / Spearman / Kendall only)."""

    def __init__(self, lib: C.CDLL):
        self.lib = lib
        lib.ref_pearson2.restype = C.c_float
        lib.ref_pearson2.argtypes = [FP, FP, C.c_int]
        lib.ref_ranks.restype = None
        lib.ref_ranks.argtypes = [FP, FP, C.c_int]
        lib.ref_kendall.restype = C.c_float
        lib.ref_kendall.argtypes = [FP, FP, C.c_int]
        lib.ref_kendall_slow.restype = C.c_float
        lib.ref_kendall_slow.argtypes = [FP, FP, C.c_int]
        lib.ref_correlation_field.restype = C.c_int
        lib.ref_correlation_field.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_size_t, FP, FP]
        lib.ref_pair_requests.restype = C.c_int
        lib.ref_pair_requests.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_size_t),
                                          C.POINTER(C.c_size_t), C.c_size_t, FP]

    def pearson(self, x, y):
        x, y = _f32(x), _f32(y)
        return float(self.lib.ref_pearson2(_fp(x), _fp(y), x.size))

    def ranks(self, v):
        v = _f32(v)
        out = np.empty_like(v)
        self.lib.ref_ranks(_fp(v), _fp(out), v.size)
        return out

    def kendall(self, x, y):
        x, y = _f32(x), _f32(y)
        return float(self.lib.ref_kendall(_fp(x), _fp(y), x.size))

    def kendall_slow(self, x, y):
        x, y = _f32(x), _f32(y)
        return float(self.lib.ref_kendall_slow(_fp(x), _fp(y), x.size))

    def field(self, measure, members, ref_values, voxel_range=None):
        members = _members(members)
        cs = len(members)
        n = members[0].size
        lo, hi = voxel_range if voxel_range is not None else (0, n)
        ref_values = _f32(ref_values)
        ptrs = (C.c_void_p * cs)(*[m.ctypes.data for m in members])
        out = np.empty(hi - lo, np.float32)
        rc = self.lib.ref_correlation_field(int(measure), ptrs, cs, lo, hi, _fp(ref_values), _fp(out))
        assert rc == 0
        return out


def _ref_pair_requests(self, measure, members, idx_i, idx_j):
    members = _members(members)
    cs = len(members)
    ii = np.ascontiguousarray(idx_i, dtype=np.uint64)
    jj = np.ascontiguousarray(idx_j, dtype=np.uint64)
    ptrs = (C.c_void_p * cs)(*[m.ctypes.data for m in members])
    out = np.empty(ii.size, np.float32)
    rc = self.lib.ref_pair_requests(int(measure), ptrs, cs, ii.ctypes.data_as(C.POINTER(C.c_size_t)),
                                    jj.ctypes.data_as(C.POINTER(C.c_size_t)), ii.size, _fp(out))
    assert rc == 0
    return out


Reference.pair_requests = _ref_pair_requests


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1)


def _members(members):
    if isinstance(members, np.ndarray):
        members = [members[i] for i in range(members.shape[0])]
    return [np.ascontiguousarray(m, dtype=np.float32).reshape(-1) for m in members]


def build_oracle():
    """(Re)builds oracle/liboracle.so and, when /root/reference is present, oracle/_ref/."""
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR)], check=True)


def load_oracle() -> Oracle:
    so = ORACLE_DIR / "liboracle.so"
    if not so.exists() or so.stat().st_mtime < (ORACLE_DIR / "corr_oracle.cpp").stat().st_mtime:
        build_oracle()
    return Oracle(C.CDLL(str(so)))


def reference_available() -> bool:
    return (ORACLE_DIR / "_ref" / "libref_corr.so").exists()


def load_reference() -> Reference:
    return Reference(C.CDLL(str(ORACLE_DIR / "_ref" / "libref_corr.so")))

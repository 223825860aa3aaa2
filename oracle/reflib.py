"""ctypes binding of oracle/_ref/libs4p_ref.so -- the reference's own sources (compiled against
oracle/eigen_shim, see oracle/Makefile target `ref`).  TEST INFRASTRUCTURE: used only to pin the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np

from .oracle import Options, make_options  # noqa: F401  (same option struct layout)

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "libs4p_ref.so")
_LIB = None


def available():
    return os.path.exists(SO)


def build(force=False):
    """Needs /root/reference (this container); the GPU box only uses the prebuilt file."""
    if not os.path.isdir("/root/reference/src"):
        return SO if os.path.exists(SO) else None
    if force or not os.path.exists(SO):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"] + (["-B"] if force else []))
    return SO


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(SO)
        fp = C.POINTER(C.c_float); ip = C.POINTER(C.c_int32)
        L.s4pr_create.restype = C.c_void_p
        L.s4pr_create.argtypes = [C.POINTER(Options)]
        L.s4pr_destroy.argtypes = [C.c_void_p]
        L.s4pr_init.argtypes = [C.c_void_p, fp, C.c_uint64, fp, C.c_uint64]
        L.s4pr_get_stats.argtypes = [C.c_void_p, ip, ip, ip, fp, fp]
        L.s4pr_get_cloud.argtypes = [C.c_void_p, C.c_int, fp]
        L.s4pr_select_quadrilateral.restype = C.c_int32
        L.s4pr_select_quadrilateral.argtypes = [C.c_void_p, fp, fp, ip, fp]
        L.s4pr_extract_pairs.restype = C.c_int64
        L.s4pr_extract_pairs.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, ip, C.c_int64]
        L.s4pr_find_congruent.restype = C.c_int64
        L.s4pr_find_congruent.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, ip, C.c_int64, ip, C.c_int64, ip, C.c_int64]
        L.s4pr_try_congruent_set.restype = C.c_int64
        L.s4pr_try_congruent_set.argtypes = [C.c_void_p, ip, ip, C.c_int64, fp, C.c_int64, C.POINTER(C.c_int64)]
        L.s4pr_verify.restype = C.c_float
        L.s4pr_verify.argtypes = [C.c_void_p, fp]
        L.s4pr_get_best.argtypes = [C.c_void_p, fp, fp, ip, ip]
        L.s4pr_compute_transformation.restype = C.c_float
        L.s4pr_compute_transformation.argtypes = [C.c_void_p, fp, C.c_uint64, fp, C.c_uint64, fp, C.POINTER(C.c_int64)]
        L.s4pr_init_attr.argtypes = [C.c_void_p, fp, fp, fp, C.c_uint64, fp, fp, fp, C.c_uint64]
        L.s4pr_compute_transformation_attr.restype = C.c_float
        L.s4pr_compute_transformation_attr.argtypes = [C.c_void_p, fp, fp, fp, C.c_uint64, fp, fp, fp, C.c_uint64, fp, C.POINTER(C.c_int64)]
        L.s4pr_bench.restype = C.c_int32
        L.s4pr_bench.argtypes = [C.c_void_p, fp, C.c_uint64, fp, C.c_uint64, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        _LIB = L
    return _LIB


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class RefMatcher:
    """GlobalRegistration::MatchSuper4PCS itself."""

    def __init__(self, options):
        self.L = lib()
        self.h = C.c_void_p(self.L.s4pr_create(C.byref(options)))
        assert self.h, "configureOverlap rejected the options"

    def __del__(self):
        try:
            if self.h:
                self.L.s4pr_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def init(self, P, Q, Pn=None, Pc=None, Qn=None, Qc=None):
        P = np.ascontiguousarray(P, np.float32); Q = np.ascontiguousarray(Q, np.float32)
        if Pn is None and Pc is None and Qn is None and Qc is None:
            self.L.s4pr_init(self.h, _f(P), P.shape[0], _f(Q), Q.shape[0])
            return
        a = [None if x is None else np.ascontiguousarray(x, np.float32) for x in (Pn, Pc, Qn, Qc)]
        g = [None if x is None else _f(x) for x in a]
        self.L.s4pr_init_attr(self.h, _f(P), g[0], g[1], P.shape[0], _f(Q), g[2], g[3], Q.shape[0])

    def compute_transformation_attr(self, P, Q, Pn=None, Pc=None, Qn=None, Qc=None):
        P = np.ascontiguousarray(P, np.float32); Q = np.ascontiguousarray(Q, np.float32).copy()
        a = [None if x is None else np.ascontiguousarray(x, np.float32) for x in (Pn, Pc, Qn, Qc)]
        g = [None if x is None else _f(x) for x in a]
        M = np.empty(16, np.float32); n = C.c_int64()
        lcp = self.L.s4pr_compute_transformation_attr(self.h, _f(P), g[0], g[1], P.shape[0], _f(Q), g[2], g[3], Q.shape[0], _f(M), C.byref(n))
        return lcp, M.reshape(4, 4), Q, int(n.value)

    def stats(self):
        t = C.c_int32(); a = C.c_int32(); b = C.c_int32(); l = C.c_float(); d = C.c_float()
        self.L.s4pr_get_stats(self.h, C.byref(t), C.byref(a), C.byref(b), C.byref(l), C.byref(d))
        return dict(number_of_trials=t.value, n_P=a.value, n_Q=b.value, best_lcp=l.value, p_diameter=d.value)

    def cloud(self, which):
        s = self.stats()
        out = np.empty((s["n_P"] if which == 0 else s["n_Q"], 3), np.float32)
        self.L.s4pr_get_cloud(self.h, which, _f(out))
        return out

    def select_quadrilateral(self):
        i1 = C.c_float(); i2 = C.c_float(); base = np.empty(4, np.int32); bx = np.empty((4, 3), np.float32)
        ok = self.L.s4pr_select_quadrilateral(self.h, C.byref(i1), C.byref(i2), _i(base), _f(bx))
        return bool(ok), i1.value, i2.value, base, bx

    def extract_pairs(self, d, na, eps, b1, b2):
        n = self.stats()["n_Q"]
        cap = max(n * n, 16)
        out = np.empty((cap, 2), np.int32)
        m = self.L.s4pr_extract_pairs(self.h, d, na, eps, b1, b2, _i(out), cap)
        return out[:m].copy()

    def find_congruent(self, inv1, inv2, thr, p1, p2, cap=1 << 22):
        p1 = np.ascontiguousarray(p1, np.int32); p2 = np.ascontiguousarray(p2, np.int32)
        out = np.empty((cap, 4), np.int32)
        K = self.L.s4pr_find_congruent(self.h, inv1, inv2, thr, _i(p1), p1.shape[0], _i(p2), p2.shape[0], _i(out), cap)
        assert K <= cap
        return out[:K].copy()

    def try_congruent_set(self, base, quads):
        base = np.ascontiguousarray(base, np.int32); quads = np.ascontiguousarray(quads, np.int32)
        cap = max(quads.shape[0], 1)
        lcps = np.empty(cap, np.float32); n = C.c_int64()
        nb = self.L.s4pr_try_congruent_set(self.h, _i(base), _i(quads), quads.shape[0], _f(lcps), cap, C.byref(n))
        return int(nb), lcps[:n.value].copy()

    def verify(self, T):
        T = np.ascontiguousarray(T, np.float32).reshape(16)
        return self.L.s4pr_verify(self.h, _f(T))

    def best(self):
        T = np.empty(16, np.float32); lcp = C.c_float(); base = np.empty(4, np.int32); cong = np.empty(4, np.int32)
        self.L.s4pr_get_best(self.h, _f(T), C.byref(lcp), _i(base), _i(cong))
        return T.reshape(4, 4), lcp.value, base, cong

    def compute_transformation(self, P, Q):
        P = np.ascontiguousarray(P, np.float32); Q = np.ascontiguousarray(Q, np.float32).copy()
        M = np.empty(16, np.float32); n = C.c_int64()
        lcp = self.L.s4pr_compute_transformation(self.h, _f(P), P.shape[0], _f(Q), Q.shape[0], _f(M), C.byref(n))
        return lcp, M.reshape(4, 4), Q, int(n.value)

    def bench(self, P, Q, budget_s):
        """(cut, candidates verified, seconds of RANSAC time)"""
        P = np.ascontiguousarray(P, np.float32); Q = np.ascontiguousarray(Q, np.float32)
        n = C.c_uint64(); sec = C.c_double()
        cut = self.L.s4pr_bench(self.h, _f(P), P.shape[0], _f(Q), Q.shape[0], float(budget_s), C.byref(n), C.byref(sec))
        return bool(cut), int(n.value), float(sec.value)

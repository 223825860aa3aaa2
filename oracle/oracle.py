"""ctypes binding of oracle/libs4p_oracle.so  --  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (super4pcs_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Options(C.Structure):
    _fields_ = [
        ("delta", C.c_float), ("max_normal_difference", C.c_float),
        ("max_translation_distance", C.c_float), ("max_angle", C.c_float),
        ("max_color_distance", C.c_float), ("sample_size", C.c_uint64),
        ("max_time_seconds", C.c_int32), ("random_seed", C.c_uint32),
        ("terminate_threshold", C.c_float), ("overlap_estimation", C.c_float),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("n_verified", C.c_uint64), ("n_quads", C.c_uint64), ("n_pairs", C.c_uint64),
        ("n_verify_queries", C.c_uint64),
        ("t_pairs", C.c_double), ("t_quads", C.c_double), ("t_verify", C.c_double), ("t_select", C.c_double),
        ("number_of_trials", C.c_int32), ("current_trial", C.c_int32), ("n_P", C.c_int32), ("n_Q", C.c_int32),
        ("best_lcp", C.c_float), ("p_diameter", C.c_float),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libs4p_oracle.so")
    src = os.path.join(_HERE, "s4p_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libs4p_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int32)
        L.s4po_create.restype = C.c_void_p
        L.s4po_create.argtypes = [C.POINTER(Options)]
        L.s4po_destroy.argtypes = [C.c_void_p]
        L.s4po_set_mode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.s4po_set_budget.argtypes = [C.c_void_p, C.c_double]
        L.s4po_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.s4po_budget_hit.restype = C.c_int32
        L.s4po_budget_hit.argtypes = [C.c_void_p]
        L.s4po_sample.restype = C.c_uint64
        L.s4po_sample.argtypes = [fp, C.c_uint64, C.c_float, fp]
        L.s4po_init.argtypes = [C.c_void_p, fp, fp, fp, C.c_uint64, fp, fp, fp, C.c_uint64]
        L.s4po_set_sampled.argtypes = [C.c_void_p, fp, C.c_uint64, fp, C.c_uint64]
        L.s4po_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.s4po_get_cloud.argtypes = [C.c_void_p, C.c_int, fp, fp, fp]
        L.s4po_get_frame.argtypes = [C.c_void_p, fp, fp, fp, fp]
        L.s4po_select_quadrilateral.restype = C.c_int32
        L.s4po_select_quadrilateral.argtypes = [C.c_void_p, fp, fp, ip, fp]
        L.s4po_set_base.argtypes = [C.c_void_p, ip]
        L.s4po_get_base.argtypes = [C.c_void_p, fp, fp, fp]
        L.s4po_extract_pairs.restype = C.c_int64
        L.s4po_extract_pairs.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, ip, C.c_int64]
        L.s4po_get_ids.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.s4po_find_congruent.restype = C.c_int64
        L.s4po_find_congruent.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, ip, C.c_int64, ip, C.c_int64, ip, C.c_int64]
        L.s4po_count_congruent.restype = None
        L.s4po_count_congruent.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, ip, C.c_int64, ip, C.c_int64, ip, C.c_int32,
                                           C.POINTER(C.c_uint64), C.c_uint64, ip, C.c_int64, C.POINTER(C.c_int64)]
        L.s4po_quad_mix.restype = C.c_uint64
        L.s4po_quad_mix.argtypes = [C.c_int32] * 4
        L.s4po_try_congruent_set.restype = C.c_int64
        L.s4po_try_congruent_set.argtypes = [C.c_void_p, ip, ip, C.c_int64, ip, C.POINTER(C.c_uint32), ip]
        L.s4po_compute_rigid.restype = C.c_int32
        L.s4po_compute_rigid.argtypes = [C.c_void_p, ip, ip, fp, fp]
        L.s4po_verify_batch.argtypes = [C.c_void_p, fp, C.c_int64, C.POINTER(C.c_uint32)]
        L.s4po_try_one_base.restype = C.c_int32
        L.s4po_try_one_base.argtypes = [C.c_void_p]
        L.s4po_get_trace.restype = C.c_int64
        L.s4po_get_trace.argtypes = [C.c_void_p, ip, fp, C.c_int64]
        L.s4po_get_best.argtypes = [C.c_void_p, fp, fp, ip, ip, fp, fp]
        L.s4po_compute_transformation.restype = C.c_float
        L.s4po_compute_transformation.argtypes = [C.c_void_p, fp, fp, fp, C.c_uint64, fp, fp, fp, C.c_uint64, fp]
        _LIB = L
    return _LIB


def _f(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _c32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def make_options(delta, overlap, sample_size, seed=5489, max_time_seconds=10 ** 6, terminate_threshold=1.0,
                 max_normal_difference=-1.0, max_translation_distance=-1.0, max_angle=-1.0, max_color_distance=-1.0):
    o = Options()
    o.delta = delta
    o.max_normal_difference = max_normal_difference
    o.max_translation_distance = max_translation_distance
    o.max_angle = max_angle
    o.max_color_distance = max_color_distance
    o.sample_size = sample_size
    o.max_time_seconds = max_time_seconds
    o.random_seed = seed
    o.terminate_threshold = terminate_threshold
    o.overlap_estimation = overlap
    return o


def sample(xyz, delta):
    xyz = _c32(xyz)
    out = np.empty_like(xyz)
    n = lib().s4po_sample(_f(xyz), xyz.shape[0], delta, _f(out))
    return out[:n].copy()


class Matcher:
    """Python face of the restated Match4PCSBase/MatchSuper4PCS (oracle)."""

    def __init__(self, options, full_counts=False, use_kdtree=True, keep_trace=False):
        self.L = lib()
        self.opt = options
        self.h = C.c_void_p(self.L.s4po_create(C.byref(options)))
        self.L.s4po_set_mode(self.h, int(full_counts), int(use_kdtree), int(keep_trace))

    def close(self):
        if self.h:
            self.L.s4po_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_mode(self, full_counts, use_kdtree=True, keep_trace=False):
        self.L.s4po_set_mode(self.h, int(full_counts), int(use_kdtree), int(keep_trace))

    def set_threads(self, n):
        """OpenMP threads of the candidate loop (baseline B of BASELINE.md section 3); 1 = the reference's serial loop."""
        self.L.s4po_set_threads(self.h, int(n))

    def set_budget(self, seconds):
        self.L.s4po_set_budget(self.h, float(seconds))

    def budget_hit(self):
        return bool(self.L.s4po_budget_hit(self.h))

    def init(self, P, Q, Pn=None, Prgb=None, Qn=None, Qrgb=None):
        P = _c32(P); Q = _c32(Q)
        Pn = None if Pn is None else _c32(Pn); Qn = None if Qn is None else _c32(Qn)
        Prgb = None if Prgb is None else _c32(Prgb); Qrgb = None if Qrgb is None else _c32(Qrgb)
        self.L.s4po_init(self.h, _f(P), _f(Pn), _f(Prgb), P.shape[0], _f(Q), _f(Qn), _f(Qrgb), Q.shape[0])

    def set_sampled(self, Ps, Qs):
        """Clouds that are already sampled and centred, taken as they are (kd-tree only): for recounting transforms that
        were scored elsewhere on exactly these points (verify_batch)."""
        Ps = _c32(Ps); Qs = _c32(Qs)
        self.L.s4po_set_sampled(self.h, _f(Ps), Ps.shape[0], _f(Qs), Qs.shape[0])

    def stats(self):
        s = Stats()
        self.L.s4po_get_stats(self.h, C.byref(s))
        return s

    def cloud(self, which, attrs=False):
        s = self.stats()
        n = s.n_P if which == 0 else s.n_Q
        xyz = np.empty((n, 3), np.float32)
        if not attrs or which == 2:
            self.L.s4po_get_cloud(self.h, which, _f(xyz), None, None)
            return xyz
        nrm = np.empty((n, 3), np.float32); rgb = np.empty((n, 3), np.float32)
        self.L.s4po_get_cloud(self.h, which, _f(xyz), _f(nrm), _f(rgb))
        return xyz, nrm, rgb

    def frame(self):
        cp = np.empty(3, np.float32); cq = np.empty(3, np.float32); g = np.empty(3, np.float32)
        r = C.c_float()
        self.L.s4po_get_frame(self.h, _f(cp), _f(cq), _f(g), C.byref(r))
        return cp, cq, g, r.value

    def select_quadrilateral(self):
        i1 = C.c_float(); i2 = C.c_float()
        base = np.empty(4, np.int32); bx = np.empty((4, 3), np.float32)
        ok = self.L.s4po_select_quadrilateral(self.h, C.byref(i1), C.byref(i2), _i(base), _f(bx))
        return bool(ok), i1.value, i2.value, base, bx

    def set_base(self, base):
        base = np.ascontiguousarray(base, np.int32)
        self.L.s4po_set_base(self.h, _i(base))

    def get_base(self):
        x = np.empty((4, 3), np.float32); n = np.empty((4, 3), np.float32); c = np.empty((4, 3), np.float32)
        self.L.s4po_get_base(self.h, _f(x), _f(n), _f(c))
        return x, n, c

    def extract_pairs(self, d, normal_angle, eps, bp1, bp2):
        # Each call mutates the persistent ids permutation, so the buffer must be big enough first time.
        n = self.stats().n_Q
        return self.extract_pairs_cap(d, normal_angle, eps, bp1, bp2, min(max(n * n, 16), 1 << 26))

    def extract_pairs_cap(self, d, normal_angle, eps, bp1, bp2, cap):
        out = np.empty((cap, 2), np.int32)
        m = self.L.s4po_extract_pairs(self.h, d, normal_angle, eps, bp1, bp2, _i(out), cap)
        if m > cap:
            raise RuntimeError("pair capacity exceeded: %d > %d" % (m, cap))
        return out[:m].copy()

    def ids(self):
        n = self.stats().n_Q
        out = np.empty(n, np.uint32)
        self.L.s4po_get_ids(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out

    def find_congruent(self, inv1, inv2, thr, pairs1, pairs2, cap=1 << 22):
        p1 = np.ascontiguousarray(pairs1, np.int32); p2 = np.ascontiguousarray(pairs2, np.int32)
        out = np.empty((cap, 4), np.int32)
        K = self.L.s4po_find_congruent(self.h, inv1, inv2, thr, _i(p1), p1.shape[0], _i(p2), p2.shape[0], _i(out), cap)
        if K > cap:
            raise RuntimeError("quad capacity exceeded: %d > %d" % (K, cap))
        return out[:K].copy()

    def count_congruent(self, inv1, inv2, thr, pairs1, pairs2, base=None, threads=0, sample_mod=0, sample_cap=1 << 16):
        """Streaming FindCongruentQuadrilaterals (+ rms gate if `base` is given): dict(K, quad_sum, C, cand_sum, sample)."""
        p1 = np.ascontiguousarray(pairs1, np.int32); p2 = np.ascontiguousarray(pairs2, np.int32)
        out = (C.c_uint64 * 4)(); ns = C.c_int64()
        b = None if base is None else np.ascontiguousarray(base, np.int32)
        smp = np.empty((sample_cap, 4), np.int32) if sample_mod else None
        self.L.s4po_count_congruent(self.h, inv1, inv2, thr, _i(p1), p1.shape[0], _i(p2), p2.shape[0],
                                    None if b is None else _i(b), int(threads) or (os.cpu_count() or 1), out, int(sample_mod),
                                    None if smp is None else _i(smp), sample_cap, C.byref(ns))
        if sample_mod and ns.value > sample_cap:
            raise RuntimeError("sample capacity exceeded: %d > %d" % (ns.value, sample_cap))
        return {"K": int(out[0]), "quad_sum": int(out[1]), "C": int(out[2]), "cand_sum": int(out[3]),
                "sample": None if smp is None else smp[:ns.value].copy()}

    def count_congruent_best(self, inv1, inv2, thr, pairs1, pairs2, base, threads=0):
        """count_congruent + the winner TryCongruentSet would keep (every gated candidate verified in full): adds
        found, best_count, best_quad (first maximum in the reference's candidate order)."""
        p1 = np.ascontiguousarray(pairs1, np.int32); p2 = np.ascontiguousarray(pairs2, np.int32)
        out = (C.c_uint64 * 4)(); best = (C.c_uint64 * 4)()
        b = np.ascontiguousarray(base, np.int32)
        self.L.s4po_count_congruent_best.restype = None
        self.L.s4po_count_congruent_best.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int32), C.c_int64,
                                                     C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int32), C.c_int32,
                                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        self.L.s4po_count_congruent_best(self.h, inv1, inv2, thr, _i(p1), p1.shape[0], _i(p2), p2.shape[0], _i(b),
                                         int(threads) or (os.cpu_count() or 1), out, best)
        quad = None
        if best[0]:
            quad = [int(p1[best[2], 0]), int(p1[best[2], 1]), int(p2[best[3], 0]), int(p2[best[3], 1])]
        return {"K": int(out[0]), "quad_sum": int(out[1]), "C": int(out[2]), "cand_sum": int(out[3]),
                "found": bool(best[0]), "best_count": int(best[1]), "best_quad": quad}

    def try_congruent_set(self, base, quads):
        base = np.ascontiguousarray(base, np.int32); quads = np.ascontiguousarray(quads, np.int32)
        K = quads.shape[0]
        per = np.empty(max(K, 1), np.int32)
        bc = C.c_uint32(); bi = C.c_int32(-1)
        nb = self.L.s4po_try_congruent_set(self.h, _i(base), _i(quads), K, _i(per), C.byref(bc), C.byref(bi))
        return int(nb), per[:K].copy(), bc.value, bi.value

    def compute_rigid(self, base, quad):
        base = np.ascontiguousarray(base, np.int32); quad = np.ascontiguousarray(quad, np.int32)
        T = np.empty(16, np.float32); rms = C.c_float()
        ok = self.L.s4po_compute_rigid(self.h, _i(base), _i(quad), _f(T), C.byref(rms))
        return bool(ok), rms.value, T.reshape(4, 4)

    def verify_batch(self, T):
        T = np.ascontiguousarray(T, np.float32).reshape(-1, 16)
        out = np.empty(T.shape[0], np.uint32)
        self.L.s4po_verify_batch(self.h, _f(T), T.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out

    def try_one_base(self):
        return bool(self.L.s4po_try_one_base(self.h))

    def trace(self, cap=1 << 16):
        out = np.empty((cap, 11), np.int32); inv = np.empty((cap, 2), np.float32)
        n = self.L.s4po_get_trace(self.h, _i(out), _f(inv), cap)
        return out[:n].copy(), inv[:n].copy()

    def best(self):
        T = np.empty(16, np.float32); lcp = C.c_float()
        base = np.empty(4, np.int32); cong = np.empty(4, np.int32)
        c1 = np.empty(3, np.float32); c2 = np.empty(3, np.float32)
        self.L.s4po_get_best(self.h, _f(T), C.byref(lcp), _i(base), _i(cong), _f(c1), _f(c2))
        return T.reshape(4, 4), lcp.value, base, cong, c1, c2

    def compute_transformation(self, P, Q, Pn=None, Prgb=None, Qn=None, Qrgb=None):
        """Returns (lcp, M 4x4 row-major, transformed Q)."""
        P = _c32(P); Q = _c32(Q).copy()
        Pn = None if Pn is None else _c32(Pn); Qn = None if Qn is None else _c32(Qn)
        Prgb = None if Prgb is None else _c32(Prgb); Qrgb = None if Qrgb is None else _c32(Qrgb)
        M = np.empty(16, np.float32)
        lcp = self.L.s4po_compute_transformation(self.h, _f(P), _f(Pn), _f(Prgb), P.shape[0],
                                                 _f(Q), _f(Qn), _f(Qrgb), Q.shape[0], _f(M))
        return lcp, M.reshape(4, 4), Q

"""ctypes binding of the C ABI in include/s4p_capi.h (libsuper4pcs_amd.so).

This is plumbing for tests, bench.py and the Python mirror of the reference's
matcher interface (super4pcs_amd/matcher.py).  There is no CPU fallback: if the
shared library is missing or no gfx950 device is visible, calls raise S4PError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# S4P_LIB: explicit path of the shared library (A/B runs of kernel variants on the GPU box); default = the in-tree build
LIB_PATH = os.environ.get("S4P_LIB") or os.path.join(_HERE, "lib", "libsuper4pcs_amd.so")

S4P_OK = 0
S4P_ERR_CAPACITY = -5
ERR_NAMES = {0: "OK", -1: "BAD_ARG", -2: "NO_DEVICE", -3: "HIP", -4: "OOM", -5: "CAPACITY", -6: "UNSUPPORTED", -7: "STATE"}

EXPORTED_SYMBOLS = [
    "s4p_create", "s4p_destroy", "s4p_last_error", "s4p_device_name", "s4p_set_clouds", "s4p_set_base",
    "s4p_extract_pairs", "s4p_find_congruent", "s4p_try_congruent_set", "s4p_verify_transforms", "s4p_verify_transforms_counted",
    "s4p_transform_points_device", "s4p_apply_bench", "s4p_select_base_points", "s4p_grow_limits", "s4p_get_limits", "s4p_try_base", "s4p_last_verified", "s4p_try_base_async", "s4p_try_base_wait", "s4p_pair_state_words", "s4p_pair_state_save", "s4p_pair_state_restore", "s4p_stage_slots", "s4p_pipeline_depth", "s4p_stage_base", "s4p_try_base_staged_async", "s4p_skip_base", "s4p_last_candidates", "s4p_transform_points", "s4p_profile_enable", "s4p_profile_get",
    "s4p_selftest_ieee", "s4p_set_quad_chunking", "s4p_chunk_stats", "s4p_set_auto_grow", "s4p_lane_growths", "s4p_border_stats", "s4p_set_clouds_timing", "s4p_set_best_hint", "s4p_select_base_points_batch", "s4p_select_batch_max", "s4p_set_quad_slice", "s4p_quad_mix",
    "s4p_set_candidate_sink", "s4p_keep_candidate_records", "s4p_verify_kernel_info",
]


# The lanes are HIP streams and the runtime maps them onto GPU_MAX_HW_QUEUES hardware queues (default 4; 8 measured +2 %): the
# request has to be in the environment before HIP initialises, so it is made here, at import (the library itself never
# touches the process environment; see s4p_capi_ctx.inc)
if os.environ.get("S4P_KEEP_HW_QUEUES") != "1":        # opt-out: leave the runtime's default number of hardware queues alone
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


class S4PError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("s4p error %s (%d): %s" % (ERR_NAMES.get(code, "?"), code, msg))
        self.code = code


class Options(C.Structure):
    _fields_ = [
        ("delta", C.c_float), ("max_normal_difference", C.c_float),
        ("max_translation_distance", C.c_float), ("max_angle", C.c_float),
        ("max_color_distance", C.c_float), ("sample_size", C.c_uint64),
        ("max_time_seconds", C.c_int32), ("random_seed", C.c_uint32),
        ("terminate_threshold", C.c_float), ("overlap_estimation", C.c_float),
    ]


class Limits(C.Structure):
    _fields_ = [("max_pairs", C.c_uint64), ("max_quads", C.c_uint64), ("max_grid_cells", C.c_uint64)]


class BaseResult(C.Structure):
    _fields_ = [
        ("n_pairs1", C.c_uint64), ("n_pairs2", C.c_uint64), ("n_quads", C.c_uint64), ("n_verified", C.c_uint64),
        ("best_count", C.c_uint32), ("has_best", C.c_int32), ("best_rank", C.c_uint64),
        ("best_quad", C.c_int32 * 4), ("best_transform", C.c_float * 16),
        ("best_centroid2", C.c_float * 3), ("centroid1", C.c_float * 3),
        ("quad_checksum", C.c_uint64), ("cand_checksum", C.c_uint64),
    ]


class Profile(C.Structure):
    _fields_ = [
        ("verify_launches", C.c_uint64), ("verify_ms_total", C.c_double), ("verify_candidates", C.c_uint64),
        ("verify_quads", C.c_uint64), ("verify_point_tests", C.c_uint64), ("verify_queries", C.c_uint64),
        ("verify_l0_pass", C.c_uint64), ("verify_l1_pass", C.c_uint64), ("verify_l2_pass", C.c_uint64),
        ("pairs_ms_total", C.c_double), ("quads_ms_total", C.c_double),
        ("pairs_launches", C.c_uint64), ("quads_launches", C.c_uint64),
        ("host_octree_s", C.c_double), ("host_wait_s", C.c_double), ("verify_pruned", C.c_uint64),
        ("sweep_candidates", C.c_uint64), ("sweep_survivors", C.c_uint64),
    ]


_LIB = None


def load_library():
    """Loads libsuper4pcs_amd.so (no device needed) and declares the prototypes."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise S4PError(-7, "libsuper4pcs_amd.so not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(LIB_PATH)
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int32)
    vp = C.c_void_p
    L.s4p_create.restype = C.c_int32
    L.s4p_create.argtypes = [C.POINTER(Options), C.POINTER(Limits), C.c_int32, C.POINTER(vp)]
    L.s4p_destroy.argtypes = [vp]
    L.s4p_last_error.restype = C.c_char_p
    L.s4p_last_error.argtypes = [vp]
    L.s4p_device_name.argtypes = [vp, C.c_char_p, C.c_int32]
    L.s4p_set_clouds.restype = C.c_int32
    L.s4p_set_clouds.argtypes = [vp, fp, fp, fp, C.c_int64, fp, fp, fp, fp, fp, fp, fp, fp, fp, C.c_int64]
    L.s4p_set_base.restype = C.c_int32
    L.s4p_set_base.argtypes = [vp, fp, fp, fp]
    L.s4p_extract_pairs.restype = C.c_int32
    L.s4p_extract_pairs.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, ip, C.c_int64, C.POINTER(C.c_int64)]
    L.s4p_find_congruent.restype = C.c_int32
    L.s4p_find_congruent.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float, ip, C.c_int64, ip, C.c_int64, ip, C.c_int64, C.POINTER(C.c_int64)]
    L.s4p_try_congruent_set.restype = C.c_int32
    L.s4p_try_congruent_set.argtypes = [vp, ip, ip, C.c_int64, ip, C.POINTER(BaseResult)]
    L.s4p_verify_transforms.restype = C.c_int32
    L.s4p_verify_transforms.argtypes = [vp, fp, C.c_int64, C.POINTER(C.c_uint32)]
    if hasattr(L, "s4p_verify_transforms_counted"):      # (an older library named by S4P_LIB for an A/B run may lack the newest entries)
        L.s4p_verify_transforms_counted.restype = C.c_int32
        L.s4p_verify_transforms_counted.argtypes = [vp, fp, C.c_int64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.s4p_try_base.restype = C.c_int32
    L.s4p_try_base.argtypes = [vp, ip, C.c_float, C.c_float, C.POINTER(BaseResult)]
    L.s4p_skip_base.restype = C.c_int32
    L.s4p_skip_base.argtypes = [vp]
    L.s4p_last_candidates.restype = C.c_int32
    L.s4p_last_candidates.argtypes = [vp, ip, ip, C.c_int64, C.POINTER(C.c_int64)]
    L.s4p_transform_points.restype = C.c_int32
    L.s4p_transform_points.argtypes = [vp, fp, fp, fp, fp, C.c_int64]
    if hasattr(L, "s4p_apply_bench"):
        L.s4p_transform_points_device.restype = C.c_int32
        L.s4p_transform_points_device.argtypes = [vp, fp, vp, vp, vp, C.c_int64]
        L.s4p_apply_bench.restype = C.c_int32
        L.s4p_apply_bench.argtypes = [vp, C.c_int64, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
    if hasattr(L, "s4p_select_base_points"):
        L.s4p_select_base_points.restype = C.c_int32
        L.s4p_select_base_points.argtypes = [vp, C.POINTER(C.c_uint32), C.c_float, C.c_float, ip, fp, ip]
    if hasattr(L, "s4p_select_base_points_batch"):
        L.s4p_select_base_points_batch.restype = C.c_int32
        L.s4p_select_base_points_batch.argtypes = [vp, C.POINTER(C.c_uint32), C.c_int32, C.c_float, C.c_float, ip, fp, ip]
        L.s4p_select_batch_max.restype = C.c_int32
        L.s4p_select_batch_max.argtypes = []
    L.s4p_profile_enable.restype = C.c_int32
    L.s4p_profile_enable.argtypes = [vp, C.c_int32, C.c_int32]
    L.s4p_profile_get.restype = C.c_int32
    L.s4p_profile_get.argtypes = [vp, C.POINTER(Profile), C.c_int32]
    L.s4p_selftest_ieee.restype = C.c_int32
    L.s4p_selftest_ieee.argtypes = [vp, fp, fp, C.c_int64, fp, fp, fp]
    if hasattr(L, "s4p_set_quad_chunking"):
        L.s4p_set_quad_chunking.restype = C.c_int32
        L.s4p_set_quad_chunking.argtypes = [vp, C.c_int32, C.c_uint64]
        L.s4p_chunk_stats.restype = C.c_int32
        L.s4p_chunk_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.s4p_set_auto_grow.restype = C.c_int32
        L.s4p_set_auto_grow.argtypes = [vp, C.c_int32]
        L.s4p_lane_growths.restype = C.c_int64
        L.s4p_lane_growths.argtypes = [vp]
        L.s4p_border_stats.restype = C.c_int32
        L.s4p_border_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.s4p_set_clouds_timing.restype = C.c_int32
        L.s4p_set_clouds_timing.argtypes = [vp, C.POINTER(C.c_double)]
        L.s4p_set_best_hint.restype = C.c_int32
        L.s4p_set_best_hint.argtypes = [vp, C.c_uint32]
        L.s4p_quad_mix.restype = C.c_uint64
        L.s4p_quad_mix.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    _LIB = L
    return L


def _f(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def _col(a, k):
    return np.ascontiguousarray(a[:, k], dtype=np.float32)


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


class Context:
    """One s4p_ctx (one matcher, one GPU)."""

    def __init__(self, options, device=0, max_pairs=0, max_quads=0, max_grid_cells=0):
        self.L = load_library()
        lim = Limits(max_pairs, max_quads, max_grid_cells)
        h = C.c_void_p()
        rc = self.L.s4p_create(C.byref(options), C.byref(lim), device, C.byref(h))
        if rc != S4P_OK:
            raise S4PError(rc, self.L.s4p_last_error(None).decode())
        self.h = h
        self.opt = options
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.L.s4p_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != S4P_OK:
            raise S4PError(rc, self.L.s4p_last_error(self.h).decode())

    def device_name(self):
        b = C.create_string_buffer(256)
        self._chk(self.L.s4p_device_name(self.h, b, 256))
        return b.value.decode()

    def set_clouds(self, P, Q, Qn=None, Qrgb=None):
        """P, Q: (n,3) float32 sampled + centred clouds."""
        P = np.ascontiguousarray(P, np.float32); Q = np.ascontiguousarray(Q, np.float32)
        cols = [_col(P, 0), _col(P, 1), _col(P, 2), _col(Q, 0), _col(Q, 1), _col(Q, 2)]
        n = [None] * 3 if Qn is None else [_col(np.asarray(Qn, np.float32), k) for k in range(3)]
        c = [None] * 3 if Qrgb is None else [_col(np.asarray(Qrgb, np.float32), k) for k in range(3)]
        self.n_p, self.n_q = P.shape[0], Q.shape[0]
        self._chk(self.L.s4p_set_clouds(self.h, _f(cols[0]), _f(cols[1]), _f(cols[2]), P.shape[0],
                                        _f(cols[3]), _f(cols[4]), _f(cols[5]),
                                        _f(n[0]), _f(n[1]), _f(n[2]), _f(c[0]), _f(c[1]), _f(c[2]), Q.shape[0]))

    def set_base(self, xyz, nrm=None, rgb=None):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(12)
        nrm = None if nrm is None else np.ascontiguousarray(nrm, np.float32).reshape(12)
        rgb = None if rgb is None else np.ascontiguousarray(rgb, np.float32).reshape(12)
        self._chk(self.L.s4p_set_base(self.h, _f(xyz), _f(nrm), _f(rgb)))

    def extract_pairs(self, d, normal_angle, eps, bp1, bp2, cap=None):
        cap = cap or max(self.n_q * self.n_q, 16)
        out = np.empty((cap, 2), np.int32)
        m = C.c_int64()
        self._chk(self.L.s4p_extract_pairs(self.h, d, normal_angle, eps, bp1, bp2, _i(out), cap, C.byref(m)))
        return out[:m.value].copy()

    def find_congruent(self, inv1, inv2, thr, pairs1, pairs2, cap=1 << 22):
        p1 = np.ascontiguousarray(pairs1, np.int32); p2 = np.ascontiguousarray(pairs2, np.int32)
        out = np.empty((cap, 4), np.int32)
        K = C.c_int64()
        self._chk(self.L.s4p_find_congruent(self.h, inv1, inv2, thr, thr, _i(p1), p1.shape[0], _i(p2), p2.shape[0],
                                            _i(out), cap, C.byref(K)))
        return out[:K.value].copy()

    def try_congruent_set(self, base_ids, quads, want_counts=True):
        base_ids = np.ascontiguousarray(base_ids, np.int32); quads = np.ascontiguousarray(quads, np.int32).reshape(-1, 4)
        K = quads.shape[0]
        per = np.empty(max(K, 1), np.int32) if want_counts else None
        r = BaseResult()
        self._chk(self.L.s4p_try_congruent_set(self.h, _i(base_ids), _i(quads), K, _i(per), C.byref(r)))
        return r, (per[:K].copy() if want_counts else None)

    def verify_transforms(self, T):
        T = np.ascontiguousarray(T, np.float32).reshape(-1, 16)
        out = np.empty(T.shape[0], np.uint32)
        self._chk(self.L.s4p_verify_transforms(self.h, _f(T), T.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def verify_stats(self, T):
        """Instrumented Verify of a batch: {"tests", "l0", "l1", "l2"} summed over the batch (roofline byte model)."""
        T = np.ascontiguousarray(T, np.float32).reshape(-1, 16)
        out = np.empty(T.shape[0], np.uint32)
        st = (C.c_uint64 * 4)()
        self._chk(self.L.s4p_verify_transforms_counted(self.h, _f(T), T.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint32)), st))
        return {"tests": int(st[0]), "l0": int(st[1]), "l1": int(st[2]), "l2": int(st[3]), "counts": out}

    def try_base(self, base_ids, inv1, inv2):
        base_ids = np.ascontiguousarray(base_ids, np.int32)
        r = BaseResult()
        self._chk(self.L.s4p_try_base(self.h, _i(base_ids), inv1, inv2, C.byref(r)))
        return r

    def last_candidates(self, cap):
        cap = max(int(cap), 1)
        quads = np.empty((cap, 4), np.int32); counts = np.empty(cap, np.int32)
        K = C.c_int64()
        self._chk(self.L.s4p_last_candidates(self.h, _i(quads), _i(counts), cap, C.byref(K)))
        return quads[:K.value].copy(), counts[:K.value].copy()

    def last_verified(self, cap):
        """(counts[C], transforms[C, 4, 4]) of the last base's verified candidates in reference order."""
        cap = max(int(cap), 1)
        counts = np.empty(cap, np.uint32); T = np.empty((cap, 16), np.float32)
        n = C.c_int64()
        self.L.s4p_last_verified.restype = C.c_int32
        self.L.s4p_last_verified.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64)]
        self._chk(self.L.s4p_last_verified(self.h, counts.ctypes.data_as(C.POINTER(C.c_uint32)), _f(T), cap, C.byref(n)))
        return counts[:n.value].copy(), T[:n.value].reshape(-1, 4, 4).copy()

    def keep_candidate_records(self, enable=True):
        self.L.s4p_keep_candidate_records.restype = C.c_int32
        self.L.s4p_keep_candidate_records.argtypes = [C.c_void_p, C.c_int32]
        self._chk(self.L.s4p_keep_candidate_records(self.h, int(enable)))

    def set_candidate_sink(self, fn):
        """fn(counts: uint32[n], transforms: float32[n, 4, 4]) per device pass, in reference order; None removes the sink."""
        self.L.s4p_set_candidate_sink.restype = C.c_int32
        self.L.s4p_set_candidate_sink.argtypes = [C.c_void_p, CANDIDATE_SINK, C.c_void_p]
        if fn is None:
            self._sink = C.cast(None, CANDIDATE_SINK)
        else:
            def tramp(_u, counts, T, n):
                fn(np.ctypeslib.as_array(counts, (n,)).copy(), np.ctypeslib.as_array(T, (n * 16,)).reshape(n, 4, 4).copy())
            self._sink = CANDIDATE_SINK(tramp)
        self._chk(self.L.s4p_set_candidate_sink(self.h, self._sink, None))

    def transform_points(self, M, xyz):
        M = np.ascontiguousarray(M, np.float32).reshape(16)
        x, y, z = _col(xyz, 0).copy(), _col(xyz, 1).copy(), _col(xyz, 2).copy()
        self._chk(self.L.s4p_transform_points(self.h, _f(M), _f(x), _f(y), _f(z), x.shape[0]))
        return np.stack([x, y, z], axis=1)

    def apply_bench(self, n, reps=20):
        """(ms per launch VALU, ms per launch MFMA, mismatching coordinates, max |difference|) of the final apply on n points."""
        ms = (C.c_double * 2)(); mm = C.c_uint64(); mx = C.c_float()
        self._chk(self.L.s4p_apply_bench(self.h, int(n), int(reps), ms, C.byref(mm), C.byref(mx)))
        return ms[0], ms[1], int(mm.value), float(mx.value)

    def select_base_points(self, draws, limit_sq, too_small):
        """One attempt of the device base selection: (status, ids[4], xyz[4, 3]); see s4p_select_base_points."""
        d = np.ascontiguousarray(draws, np.uint32)
        assert d.shape == (2001,)
        ids = np.empty(4, np.int32); xyz = np.empty(12, np.float32); st = C.c_int32()
        self._chk(self.L.s4p_select_base_points(self.h, d.ctypes.data_as(C.POINTER(C.c_uint32)), float(limit_sq), float(too_small), _i(ids), _f(xyz), C.byref(st)))
        return int(st.value), ids, xyz.reshape(4, 3)

    def select_base_points_batch(self, draws, limit_sq, too_small):
        """Consecutive attempts in one set of launches: (status[n], ids[n, 4], xyz[n, 4, 3]); see s4p_select_base_points_batch."""
        d = np.ascontiguousarray(draws, np.uint32)
        n = d.shape[0]
        assert d.shape == (n, 2001) and 1 <= n <= int(self.L.s4p_select_batch_max())
        ids = np.empty((n, 4), np.int32); xyz = np.empty((n, 12), np.float32); st = np.empty(n, np.int32)
        self._chk(self.L.s4p_select_base_points_batch(self.h, d.ctypes.data_as(C.POINTER(C.c_uint32)), n, float(limit_sq), float(too_small),
                                                     _i(ids), _f(xyz), _i(st)))
        return st, ids, xyz.reshape(n, 4, 3)

    def profile_enable(self, events=True, point_tests=False):
        self._chk(self.L.s4p_profile_enable(self.h, int(events), int(point_tests)))

    def profile_get(self, reset=False):
        p = Profile()
        self._chk(self.L.s4p_profile_get(self.h, C.byref(p), int(reset)))
        return p

    def set_best_hint(self, best_count):
        self._chk(self.L.s4p_set_best_hint(self.h, int(best_count)))

    def set_quad_chunking(self, enable=True, grow_cap_quads=0):
        self._chk(self.L.s4p_set_quad_chunking(self.h, int(enable), int(grow_cap_quads)))

    def chunk_stats(self):
        o = (C.c_uint64 * 4)()
        self._chk(self.L.s4p_chunk_stats(self.h, o))
        return {"bases": int(o[0]), "passes": int(o[1]), "splits": int(o[2]), "quads": int(o[3])}

    def selftest_ieee(self, a, b):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        o1 = np.empty_like(a); o2 = np.empty_like(a); o3 = np.empty_like(a)
        self._chk(self.L.s4p_selftest_ieee(self.h, _f(a), _f(b), a.shape[0], _f(o1), _f(o2), _f(o3)))
        return o1, o2, o3


# ----------------------------------------------------------------------------------------------
# include/s4p_matcher.h : host RANSAC driver (C++ engine) behind Match4PCSBase::ComputeTransformation
# ----------------------------------------------------------------------------------------------
class CloudView(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in ("x", "y", "z", "nx", "ny", "nz", "r", "g", "b")] + [("n", C.c_int64)]


class MatcherInfo(C.Structure):
    _fields_ = [
        ("number_of_trials", C.c_int32), ("current_trial", C.c_int32), ("n_sampled_p", C.c_int32), ("n_sampled_q", C.c_int32),
        ("best_lcp", C.c_float), ("best_count", C.c_uint32), ("p_diameter", C.c_float),
        ("centroid_p", C.c_float * 3), ("centroid_q", C.c_float * 3), ("transform", C.c_float * 16),
        ("qcentroid1", C.c_float * 3), ("qcentroid2", C.c_float * 3), ("base", C.c_int32 * 4), ("congruent", C.c_int32 * 4),
        ("candidates_verified", C.c_uint64), ("quads_total", C.c_uint64), ("pairs_total", C.c_uint64), ("bases_tried", C.c_uint64),
        ("seconds_select", C.c_double), ("seconds_device", C.c_double),
    ]


SHARD_SYMBOLS = [
    "s4p_rccl_unique_id", "s4p_shard_create", "s4p_shard_destroy", "s4p_shard_last_error", "s4p_shard_use_rccl",
    "s4p_shard_use_collective", "s4p_shard_comm_info", "s4p_shard_use_null_collective", "s4p_shard_set_mode", "s4p_shard_replay_split", "s4p_shard_run_windows", "s4p_shard_compute_transformation", "s4p_shard_replay",
    "s4p_matcher_terminate_threshold", "s4p_matcher_max_time_seconds", "s4p_matcher_init_generation",
]
MATCHER_SYMBOLS = [
    "s4p_matcher_create", "s4p_matcher_destroy", "s4p_matcher_last_error", "s4p_matcher_ctx", "s4p_uniform_dist_sample",
    "s4p_matcher_init", "s4p_matcher_init_full", "s4p_matcher_get_info", "s4p_matcher_get_sampled",
    "s4p_matcher_get_sampled_attrs", "s4p_matcher_select_quadrilateral", "s4p_matcher_try_one_base", "s4p_matcher_next_base", "s4p_matcher_next_base_async", "s4p_matcher_wait_base", "s4p_matcher_set_sharding", "s4p_matcher_visit_candidates", "s4p_matcher_commit", "s4p_matcher_perform_n_steps", "s4p_matcher_set_device_selection", "s4p_matcher_device_selection", "s4p_matcher_grow_on_overflow", "s4p_matcher_capacity_growths",
    "s4p_matcher_global_transform", "s4p_matcher_compute_transformation", "s4p_matcher_advance_trials", "s4p_matcher_set_early_exit", "s4p_matcher_loop_begin", "s4p_matcher_loop_end",
]
VISITOR_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_float, C.c_float, C.POINTER(C.c_float))
CANDIDATE_SINK = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.c_int64)
_MATCHER_DECLARED = False


def _declare_matcher(L):
    global _MATCHER_DECLARED
    if _MATCHER_DECLARED:
        return
    fp = C.POINTER(C.c_float); ip = C.POINTER(C.c_int32); vp = C.c_void_p; cv = C.POINTER(CloudView)
    L.s4p_matcher_create.restype = C.c_int32
    L.s4p_matcher_create.argtypes = [C.POINTER(Options), C.POINTER(Limits), C.c_int32, C.POINTER(vp)]
    L.s4p_matcher_destroy.argtypes = [vp]
    L.s4p_matcher_last_error.restype = C.c_char_p
    L.s4p_matcher_last_error.argtypes = [vp]
    L.s4p_matcher_ctx.restype = vp
    L.s4p_matcher_ctx.argtypes = [vp]
    L.s4p_uniform_dist_sample.restype = C.c_int64
    L.s4p_uniform_dist_sample.argtypes = [fp, fp, fp, C.c_int64, C.c_float, C.POINTER(C.c_int64)]
    L.s4p_matcher_init.restype = C.c_int32
    L.s4p_matcher_init.argtypes = [vp, cv, cv, C.c_int32]
    L.s4p_matcher_init_full.restype = C.c_int32
    L.s4p_matcher_init_full.argtypes = [vp, cv, cv]
    L.s4p_matcher_get_info.restype = C.c_int32
    L.s4p_matcher_get_info.argtypes = [vp, C.POINTER(MatcherInfo)]
    L.s4p_matcher_get_sampled.restype = C.c_int32
    L.s4p_matcher_get_sampled.argtypes = [vp, C.c_int32, fp, fp, fp]
    L.s4p_matcher_select_quadrilateral.restype = C.c_int32
    L.s4p_matcher_select_quadrilateral.argtypes = [vp, ip, fp, fp, ip, fp]
    L.s4p_matcher_try_one_base.restype = C.c_int32
    L.s4p_matcher_try_one_base.argtypes = [vp, ip, C.POINTER(BaseResult)]
    L.s4p_matcher_next_base.restype = C.c_int32
    L.s4p_matcher_next_base.argtypes = [vp, C.c_int32, ip, ip, C.POINTER(BaseResult)]
    L.s4p_matcher_set_sharding.restype = C.c_int32
    L.s4p_matcher_set_sharding.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    L.s4p_matcher_next_base_async.restype = C.c_int32
    L.s4p_matcher_next_base_async.argtypes = [vp, C.c_int32, ip, ip]
    L.s4p_matcher_wait_base.restype = C.c_int32
    L.s4p_matcher_wait_base.argtypes = [vp, C.POINTER(BaseResult)]
    L.s4p_matcher_commit.restype = C.c_int32
    L.s4p_matcher_commit.argtypes = [vp, C.c_int32, ip, C.POINTER(BaseResult), ip]
    L.s4p_matcher_perform_n_steps.restype = C.c_int32
    L.s4p_matcher_perform_n_steps.argtypes = [vp, C.c_int32, VISITOR_FN, vp, C.c_int32, fp, ip, ip]
    L.s4p_matcher_global_transform.restype = C.c_int32
    L.s4p_matcher_global_transform.argtypes = [vp, fp]
    if hasattr(L, "s4p_matcher_grow_on_overflow"):
        L.s4p_matcher_grow_on_overflow.restype = C.c_int32
        L.s4p_matcher_grow_on_overflow.argtypes = [vp, C.c_int32]
        L.s4p_matcher_capacity_growths.restype = C.c_int32
        L.s4p_matcher_capacity_growths.argtypes = [vp]
        L.s4p_get_limits.restype = C.c_int32
        L.s4p_get_limits.argtypes = [vp, C.POINTER(Limits)]
        L.s4p_grow_limits.restype = C.c_int32
        L.s4p_grow_limits.argtypes = [vp, C.c_uint64, C.c_uint64]
    if hasattr(L, "s4p_matcher_set_device_selection"):
        L.s4p_matcher_set_device_selection.restype = C.c_int32
        L.s4p_matcher_set_device_selection.argtypes = [vp, C.c_int32]
        L.s4p_matcher_device_selection.restype = C.c_int32
        L.s4p_matcher_device_selection.argtypes = [vp]
    L.s4p_matcher_compute_transformation.restype = C.c_int32
    L.s4p_matcher_compute_transformation.argtypes = [vp, cv, cv, fp, fp, fp, fp, fp]
    _MATCHER_DECLARED = True


def uniform_dist_sample(xyz, delta):
    """UniformDistSampler (sampling.h:104-121): indices of the kept points."""
    L = load_library(); _declare_matcher(L)
    xyz = np.ascontiguousarray(xyz, np.float32)
    x, y, z = _col(xyz, 0), _col(xyz, 1), _col(xyz, 2)
    out = np.empty(xyz.shape[0], np.int64)
    k = L.s4p_uniform_dist_sample(_f(x), _f(y), _f(z), xyz.shape[0], delta, out.ctypes.data_as(C.POINTER(C.c_int64)))
    return out[:k].copy()


class _View:
    """Keeps the SoA columns of an (n,3) cloud (+ optional normals / rgb) alive for a CloudView."""

    def __init__(self, xyz, nrm=None, rgb=None):
        xyz = np.ascontiguousarray(xyz, np.float32)
        self.cols = [_col(xyz, k).copy() for k in range(3)]
        self.ncols = None if nrm is None else [_col(np.asarray(nrm, np.float32), k).copy() for k in range(3)]
        self.ccols = None if rgb is None else [_col(np.asarray(rgb, np.float32), k).copy() for k in range(3)]
        v = CloudView()
        v.x, v.y, v.z = (_f(c) for c in self.cols)
        if self.ncols is not None:
            v.nx, v.ny, v.nz = (_f(c) for c in self.ncols)
        if self.ccols is not None:
            v.r, v.g, v.b = (_f(c) for c in self.ccols)
        v.n = xyz.shape[0]
        self.view = v


class Matcher:
    """One s4p_matcher: the engine behind MatchSuper4PCS (reference: algorithms/super4pcs.h:56-130)."""

    def __init__(self, options, device=0, max_pairs=0, max_quads=0, max_grid_cells=0):
        self.L = load_library(); _declare_matcher(self.L)
        lim = Limits(max_pairs, max_quads, max_grid_cells)
        h = C.c_void_p()
        rc = self.L.s4p_matcher_create(C.byref(options), C.byref(lim), device, C.byref(h))
        if rc != S4P_OK:
            raise S4PError(rc, self.L.s4p_last_error(None).decode())
        self.h = h
        self.opt = options

    def close(self):
        if getattr(self, "h", None):
            self.L.s4p_matcher_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != S4P_OK:
            raise S4PError(rc, self.L.s4p_matcher_last_error(self.h).decode())

    def ctx_handle(self):
        return C.c_void_p(self.L.s4p_matcher_ctx(self.h))

    def profile_enable(self, events=True, point_tests=False):
        self._chk(self.L.s4p_profile_enable(self.ctx_handle(), int(events), int(point_tests)))

    def profile_get(self, reset=False):
        p = Profile()
        self._chk(self.L.s4p_profile_get(self.ctx_handle(), C.byref(p), int(reset)))
        return p

    def set_quad_slice(self, part, parts):
        """SURVEY 8e level 2: this matcher's fused passes take the share `part` of `parts` of every base's second pair set."""
        rc = self.L.s4p_set_quad_slice(self.ctx_handle(), int(part), int(parts))
        if rc != S4P_OK:
            raise S4PError(rc, self.L.s4p_last_error(self.ctx_handle()).decode())

    def verify_kernel_info(self):
        buf = C.create_string_buffer(1024)
        self.L.s4p_verify_kernel_info.restype = C.c_int32
        self.L.s4p_verify_kernel_info.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
        self.L.s4p_verify_kernel_info(self.ctx_handle(), buf, 1024)
        return buf.value.decode()

    def set_quad_chunking(self, enable=True, grow_cap_quads=0):
        rc = self.L.s4p_set_quad_chunking(self.ctx_handle(), int(enable), int(grow_cap_quads))
        if rc != S4P_OK:
            raise S4PError(rc, self.L.s4p_last_error(self.ctx_handle()).decode())

    def chunk_stats(self):
        o = (C.c_uint64 * 4)()
        self.L.s4p_chunk_stats(self.ctx_handle(), o)
        return {"bases": int(o[0]), "passes": int(o[1]), "splits": int(o[2]), "quads": int(o[3])}

    def early_exit(self, enable):
        """Abandon candidates that cannot beat the registration's best (the reference's Verify early exit) in the trial loops;
        on by default, S4P_EARLY_EXIT=0 in the environment turns it off globally."""
        self.L.s4p_matcher_set_early_exit.restype = C.c_int32
        self.L.s4p_matcher_set_early_exit.argtypes = [C.c_void_p, C.c_int32]
        self._chk(self.L.s4p_matcher_set_early_exit(self.h, int(enable)))

    def loop_begin(self):
        """Bracket a driver's own trial loop (TryOneBase one at a time): commits refresh the device's early-exit bound."""
        self.L.s4p_matcher_loop_begin.restype = C.c_int32
        self.L.s4p_matcher_loop_begin.argtypes = [C.c_void_p]
        self._chk(self.L.s4p_matcher_loop_begin(self.h))

    def loop_end(self):
        self.L.s4p_matcher_loop_end.restype = C.c_int32
        self.L.s4p_matcher_loop_end.argtypes = [C.c_void_p]
        self._chk(self.L.s4p_matcher_loop_end(self.h))

    def set_clouds_timing(self):
        o = (C.c_double * 4)()
        self.L.s4p_set_clouds_timing(self.ctx_handle(), o)
        return {"host_prep_s": o[0], "device_build_s": o[1], "q_uploads_s": o[2], "total_s": o[3]}

    def border_stats(self):
        o = (C.c_uint64 * 2)()
        self.L.s4p_border_stats(self.ctx_handle(), o)
        return int(o[0]), int(o[1])

    def last_candidates(self, cap):
        """Quads (reference order) and per-candidate inlier counts (-1 = rms gate failed) of the base whose device pass
        finished last (s4p_last_candidates on the matcher's context): parity checks only."""
        cap = max(int(cap), 1)
        quads = np.empty((cap, 4), np.int32); counts = np.empty(cap, np.int32)
        K = C.c_int64()
        rc = self.L.s4p_last_candidates(self.ctx_handle(), _i(quads), _i(counts), cap, C.byref(K))
        if rc != S4P_OK:
            raise S4PError(rc, self.L.s4p_last_error(self.ctx_handle()).decode())
        return quads[:K.value].copy(), counts[:K.value].copy()

    def init_full(self, P, Q, Pn=None, Prgb=None, Qn=None, Qrgb=None):
        vp, vq = _View(P, Pn, Prgb), _View(Q, Qn, Qrgb)
        self._chk(self.L.s4p_matcher_init_full(self.h, C.byref(vp.view), C.byref(vq.view)))

    def init_sampled(self, Ps, Qu, q_needs_shuffle):
        vp, vq = _View(Ps), _View(Qu)
        self._chk(self.L.s4p_matcher_init(self.h, C.byref(vp.view), C.byref(vq.view), int(q_needs_shuffle)))

    def info(self):
        i = MatcherInfo()
        self._chk(self.L.s4p_matcher_get_info(self.h, C.byref(i)))
        return i

    def sampled(self, which):
        i = self.info()
        n = i.n_sampled_p if which == 0 else i.n_sampled_q
        x = np.empty(n, np.float32); y = np.empty(n, np.float32); z = np.empty(n, np.float32)
        self._chk(self.L.s4p_matcher_get_sampled(self.h, which, _f(x), _f(y), _f(z)))
        return np.stack([x, y, z], axis=1)

    def grow_on_overflow(self, enable):
        self._chk(self.L.s4p_matcher_grow_on_overflow(self.h, int(enable)))

    def capacity_growths(self):
        return int(self.L.s4p_matcher_capacity_growths(self.h))

    def limits(self):
        lim = Limits()
        rc = self.L.s4p_get_limits(self.ctx_handle(), C.byref(lim))
        if rc != S4P_OK:
            raise S4PError(rc, "s4p_get_limits")
        return int(lim.max_pairs), int(lim.max_quads)

    def set_device_selection(self, mode):
        """-1 by size (default), 0 host search structures, 1 device reductions; before init."""
        self._chk(self.L.s4p_matcher_set_device_selection(self.h, int(mode)))

    def device_selection(self):
        return bool(self.L.s4p_matcher_device_selection(self.h))

    def select_quadrilateral(self):
        found = C.c_int32(); i1 = C.c_float(); i2 = C.c_float()
        base = np.empty(4, np.int32); bx = np.zeros(12, np.float32)
        self._chk(self.L.s4p_matcher_select_quadrilateral(self.h, C.byref(found), C.byref(i1), C.byref(i2), _i(base), _f(bx)))
        return bool(found.value), i1.value, i2.value, base, bx.reshape(4, 3)

    def try_one_base(self):
        ok = C.c_int32(); r = BaseResult()
        self._chk(self.L.s4p_matcher_try_one_base(self.h, C.byref(ok), C.byref(r)))
        return bool(ok.value), r

    def next_base(self, run_device=True):
        found = C.c_int32(); base = np.empty(4, np.int32); r = BaseResult()
        self._chk(self.L.s4p_matcher_next_base(self.h, int(run_device), C.byref(found), _i(base), C.byref(r)))
        return bool(found.value), base, r

    def pipeline_depth(self):
        self.L.s4p_pipeline_depth.restype = C.c_int32
        self.L.s4p_pipeline_depth.argtypes = [C.c_void_p]
        return int(self.L.s4p_pipeline_depth(self.ctx_handle()))

    def visit_candidates(self, enable=True):
        self.L.s4p_matcher_visit_candidates.restype = C.c_int32
        self.L.s4p_matcher_visit_candidates.argtypes = [C.c_void_p, C.c_int32]
        self._chk(self.L.s4p_matcher_visit_candidates(self.h, int(enable)))

    def set_sharding(self, rank=0, world=1, producer_threads=True):
        """producer_threads: False / True force the helper threads off / on, 2 = where they pay (the engine's default)."""
        self._chk(self.L.s4p_matcher_set_sharding(self.h, rank, world, int(producer_threads)))

    def next_base_async(self, run_device=True):
        found = C.c_int32(); base = np.empty(4, np.int32)
        self._chk(self.L.s4p_matcher_next_base_async(self.h, int(run_device), C.byref(found), _i(base)))
        return bool(found.value), base

    def wait_base(self):
        r = BaseResult()
        self._chk(self.L.s4p_matcher_wait_base(self.h, C.byref(r)))
        return r

    def commit(self, found, base, r):
        ok = C.c_int32()
        base = np.ascontiguousarray(base, np.int32)
        self._chk(self.L.s4p_matcher_commit(self.h, int(found), _i(base), C.byref(r), C.byref(ok)))
        return bool(ok.value)

    def perform_n_steps(self, n, visitor=None, needs_global=False):
        M = np.eye(4, dtype=np.float32).reshape(16)
        imp = C.c_int32(); done = C.c_int32()
        if visitor is None:
            cb = C.cast(None, VISITOR_FN)
        else:
            def tramp(user, fraction, lcp, Tp):
                visitor(fraction, lcp, np.ctypeslib.as_array(Tp, shape=(16,)).reshape(4, 4).copy())
            cb = VISITOR_FN(tramp)
        self._chk(self.L.s4p_matcher_perform_n_steps(self.h, n, cb, None, int(needs_global), _f(M), C.byref(imp), C.byref(done)))
        return M.reshape(4, 4), bool(imp.value), bool(done.value)

    def global_transform(self):
        M = np.empty(16, np.float32)
        self._chk(self.L.s4p_matcher_global_transform(self.h, _f(M)))
        return M.reshape(4, 4)

    def compute_transformation(self, P, Q, Pn=None, Prgb=None, Qn=None, Qrgb=None):
        """Returns (lcp, M 4x4 row-major, transformed Q) like Match4PCSBase::ComputeTransformation."""
        vp, vq = _View(P, Pn, Prgb), _View(Q, Qn, Qrgb)
        n = vq.view.n
        qx = np.empty(n, np.float32); qy = np.empty(n, np.float32); qz = np.empty(n, np.float32)
        qx[:] = vq.cols[0]; qy[:] = vq.cols[1]; qz[:] = vq.cols[2]
        M = np.eye(4, dtype=np.float32).reshape(16)
        lcp = C.c_float()
        self._chk(self.L.s4p_matcher_compute_transformation(self.h, C.byref(vp.view), C.byref(vq.view), _f(qx), _f(qy), _f(qz), _f(M), C.byref(lcp)))
        return lcp.value, M.reshape(4, 4), np.stack([qx, qy, qz], axis=1)


# ----------------------------------------------------------------------------------------------
# multi-GPU: the C++ sharded trial loop (super4pcs_amd/csrc/s4p_shard.cpp, include/s4p_matcher.h)
# ----------------------------------------------------------------------------------------------
COLL_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint64))
COLL_BROADCAST_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)


class Collective(C.Structure):
    _fields_ = [("user", C.c_void_p), ("allreduce_max_u64", COLL_ALLREDUCE_FN), ("broadcast", COLL_BROADCAST_FN)]


_SHARD_DECLARED = False


def _declare_shard(L):
    global _SHARD_DECLARED
    if _SHARD_DECLARED:
        return
    _declare_matcher(L)
    vp = C.c_void_p; ip = C.POINTER(C.c_int32); fp = C.POINTER(C.c_float); cv = C.POINTER(CloudView)
    L.s4p_rccl_unique_id.restype = C.c_int32
    L.s4p_rccl_unique_id.argtypes = [C.POINTER(C.c_uint8)]
    L.s4p_shard_create.restype = C.c_int32
    L.s4p_shard_create.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.s4p_shard_destroy.argtypes = [vp]
    L.s4p_shard_last_error.restype = C.c_char_p
    L.s4p_shard_last_error.argtypes = [vp]
    L.s4p_shard_use_rccl.restype = C.c_int32
    L.s4p_shard_use_rccl.argtypes = [vp, C.c_int32, C.POINTER(C.c_uint8)]
    L.s4p_shard_use_collective.restype = C.c_int32
    L.s4p_shard_use_collective.argtypes = [vp, C.POINTER(Collective)]
    L.s4p_shard_run_windows.restype = C.c_int32
    L.s4p_shard_run_windows.argtypes = [vp, C.c_int32, C.POINTER(C.c_uint64), ip]
    L.s4p_shard_compute_transformation.restype = C.c_int32
    L.s4p_shard_compute_transformation.argtypes = [vp, cv, cv, fp, fp, fp, fp, fp]
    L.s4p_shard_replay.restype = C.c_int32
    L.s4p_shard_replay.argtypes = [C.c_int32, C.c_int32, C.POINTER(Collective), C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, ip,
                                   C.POINTER(BaseResult), ip, C.POINTER(C.c_uint32), C.c_int32, ip, ip, C.POINTER(C.c_uint64)]
    L.s4p_matcher_terminate_threshold.restype = C.c_float
    L.s4p_matcher_terminate_threshold.argtypes = [vp]
    _SHARD_DECLARED = True


def torch_collective(dist, device=None):
    """An s4p_collective whose two operations run over a torch.distributed process group: CPU tensors (gloo) by default --
    the provider the CPU tests and single-GPU dry runs plug into the C++ loop -- or tensors on `device` for a group whose
    backend needs them there ("nccl" = RCCL).  Keep the returned object alive."""
    import torch

    def allreduce(user, keyp):
        t = torch.tensor([keyp[0]], dtype=torch.int64, device=device)           # keys stay below 2^63
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        keyp[0] = int(t.item())
        return 0

    def broadcast(user, buf, nbytes, root):
        arr = (C.c_uint8 * nbytes).from_address(buf)
        t = torch.frombuffer(arr, dtype=torch.uint8).clone()
        if device is not None:
            t = t.to(device)
        dist.broadcast(t, src=root)
        C.memmove(buf, t.cpu().numpy().ctypes.data, nbytes)
        return 0

    coll = Collective()
    coll.user = None
    coll.allreduce_max_u64 = COLL_ALLREDUCE_FN(allreduce)
    coll.broadcast = COLL_BROADCAST_FN(broadcast)
    return coll


def rccl_unique_id():
    L = load_library(); _declare_shard(L)
    buf = (C.c_uint8 * 128)()
    rc = L.s4p_rccl_unique_id(buf)
    if rc != S4P_OK:
        raise S4PError(rc, "ncclGetUniqueId failed (is librccl loadable?)")
    return bytes(buf)


class Shard:
    """One rank of the sharded trial loop over a Matcher (s4p_shard)."""

    def __init__(self, matcher, rank, world, producer_threads=True):
        self.L = load_library(); _declare_shard(self.L)
        self.m, self.rank, self.world = matcher, rank, world
        h = C.c_void_p()
        rc = self.L.s4p_shard_create(matcher.h, rank, world, int(producer_threads), C.byref(h))
        if rc != S4P_OK:
            raise S4PError(rc, "s4p_shard_create failed")
        self.h = h
        self._coll = None
        self.terminated = False

    def close(self):
        if getattr(self, "h", None):
            self.L.s4p_shard_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != S4P_OK:
            raise S4PError(rc, self.L.s4p_shard_last_error(self.h).decode())

    def use_rccl(self, device, unique_id):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._chk(self.L.s4p_shard_use_rccl(self.h, device, buf))

    def comm_info(self):
        """(ranks, rank) as the shard's own RCCL communicator reports them; (-1, -1) for any other collective."""
        n = C.c_int32(-1); r = C.c_int32(-1)
        self.L.s4p_shard_comm_info.restype = C.c_int32
        self.L.s4p_shard_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        self._chk(self.L.s4p_shard_comm_info(self.h, C.byref(n), C.byref(r)))
        return int(n.value), int(r.value)

    def set_mode(self, split_bases):
        """False: trials sharded by base (default).  True: every base over all ranks (SURVEY 8e level 2); before init."""
        self.L.s4p_shard_set_mode.restype = C.c_int32
        self.L.s4p_shard_set_mode.argtypes = [C.c_void_p, C.c_int32]
        self._chk(self.L.s4p_shard_set_mode(self.h, int(bool(split_bases))))

    def use_null_collective(self):
        self.L.s4p_shard_use_null_collective.restype = C.c_int32
        self.L.s4p_shard_use_null_collective.argtypes = [C.c_void_p]
        self._chk(self.L.s4p_shard_use_null_collective(self.h))

    def use_collective(self, coll):
        self._coll = coll                       # the callbacks must outlive the shard
        self._chk(self.L.s4p_shard_use_collective(self.h, C.byref(coll)))

    def run_windows(self, n):
        cand = C.c_uint64(); term = C.c_int32()
        self._chk(self.L.s4p_shard_run_windows(self.h, n, C.byref(cand), C.byref(term)))
        self.terminated = bool(term.value)
        return int(cand.value)

    def compute_transformation(self, P, Q):
        vp, vq = _View(P), _View(Q)
        n = vq.view.n
        qx = vq.cols[0].copy(); qy = vq.cols[1].copy(); qz = vq.cols[2].copy()
        M = np.eye(4, dtype=np.float32).reshape(16)
        lcp = C.c_float()
        self._chk(self.L.s4p_shard_compute_transformation(self.h, C.byref(vp.view), C.byref(vq.view), _f(qx), _f(qy), _f(qz), _f(M), C.byref(lcp)))
        return lcp.value, M.reshape(4, 4), np.stack([qx, qy, qz], axis=1)


def shard_replay_split(rank, world, coll, found, results, depth, threshold_count, start_best):
    """s4p_shard_replay_split: the C++ split-base loop on the recorded results of this rank's shares (host only)."""
    L = load_library(); _declare_shard(L)
    L.s4p_shard_replay_split.restype = C.c_int32
    L.s4p_shard_replay_split.argtypes = [C.c_int32, C.c_int32, C.POINTER(Collective), C.c_int32, C.c_int32, C.c_uint32, C.c_uint32,
                                         C.POINTER(C.c_int32), C.POINTER(BaseResult), C.POINTER(C.c_int32), C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_uint64), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
    n = len(found)
    f = (C.c_int32 * n)(*[int(x) for x in found])
    r = (BaseResult * n)(*results)
    cap = n + 4
    ct = (C.c_int32 * cap)(); cc = (C.c_uint32 * cap)(); cg = (C.c_uint64 * cap)()
    nc = C.c_int32(); term = C.c_int32(); td = C.c_uint64()
    rc = L.s4p_shard_replay_split(rank, world, C.byref(coll), n, depth, threshold_count, start_best, f, r, ct, cc, cg, cap,
                                  C.byref(nc), C.byref(term), C.byref(td))
    if rc != S4P_OK:
        raise S4PError(rc, "s4p_shard_replay_split failed")
    k = min(nc.value, cap)
    return [(ct[i], cc[i], cg[i]) for i in range(k)], bool(term.value), int(td.value)


def shard_replay(rank, world, coll, found, results, depth, threshold_count, start_best):
    """s4p_shard_replay: the C++ window loop on recorded outcomes of this rank's trials (host only)."""
    L = load_library(); _declare_shard(L)
    n = len(found)
    f = (C.c_int32 * n)(*[int(x) for x in found])
    r = (BaseResult * n)(*results)
    cap = n * world + 4
    ct = (C.c_int32 * cap)(); cc = (C.c_uint32 * cap)()
    nc = C.c_int32(); term = C.c_int32(); td = C.c_uint64()
    rc = L.s4p_shard_replay(rank, world, C.byref(coll), n, depth, threshold_count, start_best, f, r, ct, cc, cap,
                            C.byref(nc), C.byref(term), C.byref(td))
    if rc != S4P_OK:
        raise S4PError(rc, "s4p_shard_replay failed")
    k = min(nc.value, cap)
    return [(ct[i], cc[i]) for i in range(k)], bool(term.value), int(td.value)

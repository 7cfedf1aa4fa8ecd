"""ctypes binding of the CPU oracle (oracle/libpopsift_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (popsift_amd/) never imports
this module.  PARITY UNPINNED -- see popsift_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpopsift_oracle.so")

MAX_OCTAVES = 20
ORI_MAX = 4


class Params(C.Structure):
    """Mirror of popsift_hip_params (include/popsift_hip.h)."""
    _fields_ = [
        ("octaves", C.c_int32), ("levels", C.c_int32), ("sigma", C.c_float),
        ("edge_limit", C.c_float), ("threshold", C.c_float), ("upscale_factor", C.c_float),
        ("sift_mode", C.c_int32), ("gauss_mode", C.c_int32), ("desc_mode", C.c_int32),
        ("norm_mode", C.c_int32), ("norm_multi", C.c_int32), ("max_extrema", C.c_int32),
        ("assume_initial_blur", C.c_int32), ("initial_blur", C.c_float),
        ("filter_grid_size", C.c_int32), ("filter_max_extrema", C.c_int32), ("filter_sorting", C.c_int32),
        ("reserved", C.c_int32 * 3),
    ]


def default_params(**kw):
    """popsift::Config::Config() defaults (sift_conf.cu:17-39)."""
    p = Params()
    p.octaves = -1
    p.levels = 3
    p.sigma = 1.6
    p.edge_limit = 10.0
    p.threshold = 0.04
    p.upscale_factor = 1.0
    p.sift_mode = 0
    p.gauss_mode = 0
    p.desc_mode = 0
    p.norm_mode = 0
    p.norm_multi = 0
    p.max_extrema = 100000
    p.assume_initial_blur = 1
    p.initial_blur = 0.5
    p.filter_grid_size = 2
    p.filter_max_extrema = -1
    p.filter_sorting = 0
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


FEATURE_DTYPE = np.dtype([
    ("debug_octave", np.int32), ("xpos", np.float32), ("ypos", np.float32),
    ("sigma", np.float32), ("num_ori", np.int32),
    ("orientation", np.float32, (ORI_MAX,)), ("desc_idx", np.int32, (ORI_MAX,)),
])
EXTREMUM_DTYPE = np.dtype([
    ("xpos", np.float32), ("ypos", np.float32), ("lpos", np.int32),
    ("sigma", np.float32), ("octave", np.int32), ("cell", np.int32),
])
assert FEATURE_DTYPE.itemsize == 52 and EXTREMUM_DTYPE.itemsize == 24


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "popsift_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpopsift_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
        L.oracle_create.restype = vp
        L.oracle_create.argtypes = [C.POINTER(Params)]
        L.oracle_destroy.argtypes = [vp]
        L.oracle_set_threads.argtypes = [vp, C.c_int]
        L.oracle_get_gauss_table.argtypes = [vp, vp, vp, vp, ip]
        L.oracle_plan.argtypes = [vp, C.c_int, C.c_int, ip, ip, ip]
        for n in ("oracle_run_u8", "oracle_run_f32", "oracle_build_pyramid_u8", "oracle_build_pyramid_f32"):
            getattr(L, n).argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
        L.oracle_run_keypoint_stages.argtypes = [vp]
        L.oracle_num_octaves.argtypes = [vp]
        L.oracle_octave_dims.argtypes = [vp, C.c_int, ip, ip]
        L.oracle_plane_mut.restype = fp
        L.oracle_plane_mut.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.oracle_counts.argtypes = [vp, ip, ip]
        L.oracle_ext_count.argtypes = [vp, C.c_int]
        L.oracle_fetch.argtypes = [vp, vp, vp]
        L.oracle_fetch_extrema.argtypes = [vp, vp]
        L.oracle_fetch_raw_desc.argtypes = [vp, vp]
        L.oracle_redo_descriptors.argtypes = [vp, vp, C.c_int, vp, C.c_int]
        L.oracle_warp32_sort64.argtypes = [vp, vp]
        L.oracle_solve3.argtypes = [vp, vp]
        L.oracle_normalize.argtypes = [vp, C.c_int, C.c_int]
        L.oracle_match.restype = None
        L.oracle_match.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int]
        L.oracle_filter_grid_keys.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
        _lib = L
    return _lib


class Oracle:
    def __init__(self, params=None, threads=1):
        self.params = params if params is not None else default_params()
        self._h = lib().oracle_create(C.byref(self.params))
        if not self._h:
            raise ValueError("oracle_create rejected the parameters")
        lib().oracle_set_threads(self._h, threads)

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = None

    __del__ = close

    def gauss_table(self):
        n = C.c_int()
        lib().oracle_get_gauss_table(self._h, None, None, None, C.byref(n))
        f = np.zeros((n.value, 32), np.float32)
        s = np.zeros(n.value, np.int32)
        g = np.zeros(n.value, np.float32)
        lib().oracle_get_gauss_table(self._h, f.ctypes.data, s.ctypes.data, g.ctypes.data, C.byref(n))
        return f, s, g

    def plan(self, w, h):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        if lib().oracle_plan(self._h, w, h, C.byref(a), C.byref(b), C.byref(c)):
            raise ValueError("plan")
        return a.value, b.value, c.value

    def _img(self, img):
        img = np.ascontiguousarray(img)
        assert img.ndim == 2
        if img.dtype == np.uint8:
            return img, "u8"
        if img.dtype == np.float32:
            return img, "f32"
        raise TypeError(img.dtype)

    def run(self, img, keypoints=True):
        img, kind = self._img(img)
        name = ("oracle_run_" if keypoints else "oracle_build_pyramid_") + kind
        h, w = img.shape
        rc = getattr(lib(), name)(self._h, img.ctypes.data, w, h, w)
        if rc:
            raise RuntimeError("%s failed: %d" % (name, rc))
        return self

    def run_keypoint_stages(self):
        if lib().oracle_run_keypoint_stages(self._h):
            raise RuntimeError("keypoint stages failed")
        return self

    @property
    def num_octaves(self):
        return lib().oracle_num_octaves(self._h)

    def octave_dims(self, o):
        w, h = C.c_int(), C.c_int()
        if lib().oracle_octave_dims(self._h, o, C.byref(w), C.byref(h)):
            raise IndexError(o)
        return w.value, h.value

    def plane(self, octave, kind, level, copy=True):
        """kind 0 = Gaussian, 1 = DoG.  Returns an (h, w) float32 array."""
        w, h = self.octave_dims(octave)
        p = lib().oracle_plane_mut(self._h, octave, kind, level)
        if not p:
            raise IndexError((octave, kind, level))
        a = np.ctypeslib.as_array(p, shape=(h, w))
        return a.copy() if copy else a

    def counts(self):
        a, b = C.c_int(), C.c_int()
        lib().oracle_counts(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def ext_counts(self):
        return [lib().oracle_ext_count(self._h, o) for o in range(self.num_octaves)]

    def fetch(self):
        nf, nd = self.counts()
        feats = np.zeros(nf, FEATURE_DTYPE)
        desc = np.zeros((nd, 128), np.float32)
        lib().oracle_fetch(self._h, feats.ctypes.data, desc.ctypes.data)
        return feats, desc

    def redo_descriptors(self, orientations=None, ulps=0, sigma=None, sigma_ulps=0):
        """Test hook: all descriptors again in the frames of `orientations` ((n_features, 4) float32 in fetch order; None =
        the oracle's own) and scales `sigma` ((n_features,) float32 in OCTAVE units; None = own), moved by `ulps` /
        `sigma_ulps` units in the last place."""
        nf = self.counts()[0]
        optr = sptr = None
        if orientations is not None:
            orientations = np.ascontiguousarray(orientations, np.float32)
            assert orientations.shape == (nf, 4)
            optr = orientations.ctypes.data
        if sigma is not None:
            sigma = np.ascontiguousarray(sigma, np.float32)
            assert sigma.shape == (nf,)
            sptr = sigma.ctypes.data
        if lib().oracle_redo_descriptors(self._h, optr, int(ulps), sptr, int(sigma_ulps)):
            raise RuntimeError("redo_descriptors")
        return self

    def raw_descriptors(self):
        _, nd = self.counts()
        desc = np.zeros((nd, 128), np.float32)
        lib().oracle_fetch_raw_desc(self._h, desc.ctypes.data)
        return desc

    def extrema(self):
        nf, _ = self.counts()
        out = np.zeros(nf, EXTREMUM_DTYPE)
        lib().oracle_fetch_extrema(self._h, out.ctypes.data)
        return out


def warp32_sort64(yval):
    """Warp32<float>::sort64 on 64 values: the indices held by x of lanes 0..31, then y of lanes 0..31."""
    y = np.ascontiguousarray(yval, np.float32)
    assert y.shape == (64,)
    out = np.zeros(64, np.int32)
    lib().oracle_warp32_sort64(y.ctypes.data, out.ctypes.data)
    return out


def solve3(A, b):
    A = np.array(A, np.float32).reshape(9).copy()
    b = np.array(b, np.float32).copy()
    ok = lib().oracle_solve3(A.ctypes.data, b.ctypes.data)
    return bool(ok), b


def normalize(d, norm_mode=0, norm_multi=0):
    d = np.array(d, np.float32).reshape(128).copy()
    lib().oracle_normalize(d.ctypes.data, norm_mode, norm_multi)
    return d


def filter_grid_keys(cell, scale, grid_size, filter_max, mode):
    """Grid filter on bare (cell, scale) keys; returns (keep mask, per-cell limit)."""
    cell = np.ascontiguousarray(cell, np.int32)
    scale = np.ascontiguousarray(scale, np.float32)
    keep = np.zeros(len(cell), np.uint8)
    lim = lib().oracle_filter_grid_keys(cell.ctypes.data, scale.ctypes.data, len(cell), grid_size, filter_max,
                                        mode, keep.ctypes.data)
    return keep.astype(bool), lim


MATCH_DTYPE = np.dtype([("best", np.int32), ("second", np.int32), ("accept", np.int32),
                        ("dist_best", np.float32), ("dist_second", np.float32)])


def match(l, r, threads=8):
    """Brute-force 2-NN (FeaturesDev::match, features.cu:157-221) of l's rows among r's rows."""
    l = np.ascontiguousarray(l, np.float32).reshape(-1, 128)
    r = np.ascontiguousarray(r, np.float32).reshape(-1, 128)
    out = np.zeros(len(l), MATCH_DTYPE)
    lib().oracle_match(l.ctypes.data, len(l), r.ctypes.data, len(r), out.ctypes.data, threads)
    return out

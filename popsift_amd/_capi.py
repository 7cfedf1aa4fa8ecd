"""ctypes binding of the C ABI in include/popsift_hip.h (libpopsift_hip.so).

This is the only way Python reaches the extraction path: there is no CPU
fallback.  If the HIP library is missing or fails to load, importing the
symbols raises -- loudly -- instead of silently computing somewhere else.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# POPSIFT_HIP_LIB: load another build of the SAME library (tools/build_variants.sh experiment builds); never a fallback
LIB_PATH = os.environ.get("POPSIFT_HIP_LIB") or os.path.join(_HERE, "libpopsift_hip.so")

MAX_OCTAVES = 20
ORI_MAX = 4

OK = 0
ERR_INVALID, ERR_DEVICE, ERR_NO_DEVICE, ERR_OOM, ERR_STATE, ERR_TOO_SMALL = -1, -2, -3, -4, -5, -6

SIFT_POPSIFT, SIFT_OPENCV, SIFT_VLFEAT = 0, 1, 2
GAUSS_VLFEAT_COMPUTE, GAUSS_VLFEAT_RELATIVE, GAUSS_VLFEAT_RELATIVE_ALL = 0, 1, 2
GAUSS_OPENCV_COMPUTE, GAUSS_FIXED9, GAUSS_FIXED15 = 3, 4, 5
DESC_LOOP, DESC_ILOOP, DESC_GRID, DESC_IGRID, DESC_NOTILE = 0, 1, 2, 3, 4
NORM_ROOTSIFT, NORM_CLASSIC = 0, 1


class Params(C.Structure):
    """popsift_hip_params"""
    _fields_ = [
        ("octaves", C.c_int32), ("levels", C.c_int32), ("sigma", C.c_float),
        ("edge_limit", C.c_float), ("threshold", C.c_float), ("upscale_factor", C.c_float),
        ("sift_mode", C.c_int32), ("gauss_mode", C.c_int32), ("desc_mode", C.c_int32),
        ("norm_mode", C.c_int32), ("norm_multi", C.c_int32), ("max_extrema", C.c_int32),
        ("assume_initial_blur", C.c_int32), ("initial_blur", C.c_float),
        ("filter_grid_size", C.c_int32), ("filter_max_extrema", C.c_int32), ("filter_sorting", C.c_int32),
        ("store_dog", C.c_int32), ("reserved", C.c_int32 * 2),
    ]


class Report(C.Structure):
    """popsift_hip_report"""
    _fields_ = [
        ("num_octaves", C.c_int32), ("base_w", C.c_int32), ("base_h", C.c_int32),
        ("ext_ct", C.c_int32 * MAX_OCTAVES), ("ori_ct", C.c_int32 * MAX_OCTAVES),
        ("ext_total", C.c_int32), ("ori_total", C.c_int32),
        ("ms_device", C.c_float), ("ms_blur", C.c_float), ("blur_launches", C.c_int32),
        ("blur_alg_bytes", C.c_double), ("pyramid_pixels", C.c_double),
        ("big_alg_bytes", C.c_double), ("ms_big", C.c_float), ("big_launches", C.c_int32),
        ("ms_stage", C.c_float * 8),
    ]


class DeviceInfo(C.Structure):
    """popsift_hip_device_info"""
    _fields_ = [
        ("name", C.c_char * 256), ("arch_major", C.c_int32), ("arch_minor", C.c_int32),
        ("total_mem", C.c_uint64), ("lds_per_block", C.c_uint64), ("wave_size", C.c_int32),
        ("max_threads_per_block", C.c_int32), ("max_threads_per_cu", C.c_int32),
        ("max_block", C.c_int32 * 3), ("max_grid", C.c_int32 * 3), ("cu_count", C.c_int32),
        ("concurrent_kernels", C.c_int32), ("can_map_host", C.c_int32), ("unified_addressing", C.c_int32),
    ]


FEATURE_DTYPE = np.dtype([
    ("debug_octave", np.int32), ("xpos", np.float32), ("ypos", np.float32),
    ("sigma", np.float32), ("num_ori", np.int32),
    ("orientation", np.float32, (ORI_MAX,)), ("desc_idx", np.int32, (ORI_MAX,)),
])
MATCH_DTYPE = np.dtype([("best", np.int32), ("second", np.int32), ("accept", np.int32),
                        ("dist_best", np.float32), ("dist_second", np.float32)])
EXTREMUM_DTYPE = np.dtype([
    ("xpos", np.float32), ("ypos", np.float32), ("lpos", np.int32),
    ("sigma", np.float32), ("octave", np.int32), ("cell", np.int32),
])

# every symbol include/popsift_hip.h declares: (name, restype, argtypes)
_vp, _ip = C.c_void_p, C.POINTER(C.c_int)
SYMBOLS = [
    ("popsift_hip_default_params", None, [C.POINTER(Params)]),
    ("popsift_hip_version", C.c_char_p, []),
    ("popsift_hip_strerror", C.c_char_p, [C.c_int]),
    ("popsift_hip_last_error", C.c_char_p, [_vp]),
    ("popsift_hip_device_count", C.c_int, [_ip]),
    ("popsift_hip_get_device_info", C.c_int, [C.c_int, C.POINTER(DeviceInfo)]),
    ("popsift_hip_device_numa_node", C.c_int, [C.c_int, _ip]),
    ("popsift_hip_ctx_create", C.c_int, [C.c_int, C.POINTER(Params), C.POINTER(_vp)]),
    ("popsift_hip_ctx_destroy", C.c_int, [_vp]),
    ("popsift_hip_get_gauss_table", C.c_int, [_vp, _vp, _vp, _vp, _ip]),
    ("popsift_hip_submit_u8", C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int]),
    ("popsift_hip_submit_f32", C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int]),
    ("popsift_hip_submit_dev_u8", C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int]),
    ("popsift_hip_submit_dev_f32", C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int]),
    ("popsift_hip_submit_pinned_u8", C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int]),
    ("popsift_hip_submit_pinned_f32", C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int]),
    ("popsift_hip_submit_batch", C.c_int, [_vp, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("popsift_hip_wait_batch", C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("popsift_hip_fetch_item", C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t]),
    ("popsift_hip_results_dev_item", C.c_int, [_vp, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("popsift_hip_fetch_begin_item", C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t]),
    ("popsift_hip_wait", C.c_int, [_vp, _ip, _ip]),
    ("popsift_hip_fetch", C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t]),
    ("popsift_hip_fetch_begin", C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t]),
    ("popsift_hip_fetch_end", C.c_int, [_vp]),
    ("popsift_hip_results_dev", C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    ("popsift_hip_host_alloc", _vp, [C.c_size_t]),
    ("popsift_hip_host_free", None, [_vp]),
    ("popsift_hip_clone_results", C.c_int, [_vp, C.POINTER(_vp)]),
    ("popsift_hip_devfeatures_free", C.c_int, [_vp]),
    ("popsift_hip_devfeatures_info", C.c_int, [_vp, _ip, _ip, _ip]),
    ("popsift_hip_devfeatures_ptrs", C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    ("popsift_hip_devfeatures_alloc", C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    ("popsift_hip_devfeatures_from_host", C.c_int, [C.c_int, _vp, C.c_int, C.POINTER(_vp)]),
    ("popsift_hip_devfeatures_download", C.c_int, [_vp, _vp, _vp]),
    ("popsift_hip_match_sets", C.c_int, [_vp, _vp, _vp]),
    ("popsift_hip_match_set_path", C.c_int, [C.c_int]),
    ("popsift_hip_get_report", C.c_int, [_vp, C.POINTER(Report)]),
    ("popsift_hip_set_profile", C.c_int, [_vp, C.c_int]),
    ("popsift_hip_octave_dims", C.c_int, [_vp, C.c_int, _ip, _ip]),
    ("popsift_hip_download_plane", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    ("popsift_hip_upload_plane", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    ("popsift_hip_download_extrema", C.c_int, [_vp, _vp, C.c_size_t, _ip]),
    ("popsift_hip_rerun_keypoint_stages", C.c_int, [_vp]),
    ("popsift_hip_debug_set", C.c_int, [_vp, C.c_int, C.c_int]),
]
MATCH_AUTO, MATCH_EXACT, MATCH_SCREEN = 0, 1, 2
STAGES = ("pyramid", "detect", "refine", "orientation", "scan", "descriptor")
DEBUG_DET_QCAP, DEBUG_CAND_CAP, DEBUG_OHIST_CAP, DEBUG_FAIL_ALLOC, DEBUG_DESC_ROWS, DEBUG_PYR_ORDER, DEBUG_KP_WAVES = 1, 2, 3, 4, 5, 6, 7
DEBUG_BLUR_PATH, DEBUG_BLUR_SEG, DEBUG_PYR_TAIL = 8, 9, 10
MAX_BATCH = 16
IMG_HOST_U8, IMG_HOST_F32, IMG_DEV_U8, IMG_DEV_F32, IMG_PINNED_U8, IMG_PINNED_F32 = range(6)

_lib = None


class PopsiftHipError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        msg = "%s failed: %d (%s)" % (where, status, lib().popsift_hip_strerror(status).decode())
        if detail:
            msg += ": " + detail
        super().__init__(msg)


def lib():
    """Load libpopsift_hip.so; raises if it is missing (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def default_params(**kw):
    p = Params()
    lib().popsift_hip_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def device_count():
    n = C.c_int(0)
    rc = lib().popsift_hip_device_count(C.byref(n))
    return n.value if rc == OK else 0


def device_info(device=0):
    d = DeviceInfo()
    rc = lib().popsift_hip_get_device_info(device, C.byref(d))
    if rc != OK:
        raise PopsiftHipError(rc, "popsift_hip_get_device_info")
    return d


class DevFeatures:
    """popsift_hip_devfeatures: a device-resident result set (FeaturesDev, features.h:98-118)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_host(cls, desc, device=0):
        desc = np.ascontiguousarray(desc, np.float32).reshape(-1, 128)
        h = _vp()
        rc = lib().popsift_hip_devfeatures_from_host(device, desc.ctypes.data, len(desc), C.byref(h))
        if rc != OK:
            raise PopsiftHipError(rc, "popsift_hip_devfeatures_from_host")
        return cls(h)

    def info(self):
        d, nf, nd = C.c_int(), C.c_int(), C.c_int()
        lib().popsift_hip_devfeatures_info(self._h, C.byref(d), C.byref(nf), C.byref(nd))
        return d.value, nf.value, nd.value

    def download(self):
        _, _, nd = self.info()
        desc = np.zeros((nd, 128), np.float32)
        rev = np.zeros(nd, np.int32)
        rc = lib().popsift_hip_devfeatures_download(self._h, desc.ctypes.data, rev.ctypes.data)
        if rc != OK:
            raise PopsiftHipError(rc, "popsift_hip_devfeatures_download")
        return desc, rev

    def match(self, other):
        _, _, nd = self.info()
        out = np.zeros(nd, MATCH_DTYPE)
        rc = lib().popsift_hip_match_sets(self._h, other._h, out.ctypes.data)
        if rc != OK:
            raise PopsiftHipError(rc, "popsift_hip_match_sets")
        return out

    def close(self):
        if self._h:
            lib().popsift_hip_devfeatures_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PendingFetch:
    """A download started by Context.fetch_begin; result() waits for it (popsift_hip_fetch_end)."""

    def __init__(self, ctx, nf, nd, pinned):
        self._ctx, self._done = ctx, False
        self._fbytes, self._dbytes = max(nf, 1) * FEATURE_DTYPE.itemsize, max(nd, 1) * 512
        self._pin = []
        if pinned:
            for n in (self._fbytes, self._dbytes):
                p = lib().popsift_hip_host_alloc(n)
                if not p:
                    self._release()
                    raise MemoryError("popsift_hip_host_alloc(%d)" % n)
                self._pin.append(p)
            fb = (C.c_char * self._fbytes).from_address(self._pin[0])
            db = (C.c_char * self._dbytes).from_address(self._pin[1])
            self._feats = np.frombuffer(fb, FEATURE_DTYPE, nf)
            self._desc = np.frombuffer(db, np.float32, nd * 128).reshape(nd, 128)
        else:
            self._feats = np.zeros(nf, FEATURE_DTYPE)
            self._desc = np.zeros((nd, 128), np.float32)
        try:
            ctx._chk(lib().popsift_hip_fetch_begin(ctx._h, self._feats.ctypes.data, nf, self._desc.ctypes.data, nd * 128),
                     "popsift_hip_fetch_begin")
        except Exception:
            self._release()
            raise

    def _release(self):
        for p in self._pin:
            lib().popsift_hip_host_free(p)
        self._pin = []

    def _landed(self):
        """the download is complete (fetch_end, or a later fetch_begin on the same context, has waited for it)"""
        self._done = True
        if self._pin:
            self._feats, self._desc = self._feats.copy(), self._desc.copy()
            self._release()

    def result(self):
        """(feats, desc) as ordinary numpy arrays (copied out of the pinned blocks, which are released)."""
        if not self._done:
            rc = lib().popsift_hip_fetch_end(self._ctx._h)
            # ERR_STATE: nothing is pending any more -- a later fetch_begin on the context has already waited for this
            # download (popsift_hip_fetch_begin drains the previous one first, also when it then fails): the data is there
            if rc != ERR_STATE:
                self._ctx._chk(rc, "popsift_hip_fetch_end")
            self._landed()
        return self._feats, self._desc

    def __del__(self):
        try:
            if not self._done and self._ctx._h:
                lib().popsift_hip_fetch_end(self._ctx._h)
            self._release()
        except Exception:
            pass


class Context:
    """One extraction context (popsift_hip_ctx) on one GPU."""

    def __init__(self, params=None, device=0):
        self._h = None
        self.params = params if params is not None else default_params()
        h = _vp()
        rc = lib().popsift_hip_ctx_create(device, C.byref(self.params), C.byref(h))
        if rc != OK:
            raise PopsiftHipError(rc, "popsift_hip_ctx_create")
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            lib().popsift_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, where):
        if rc != OK:
            raise PopsiftHipError(rc, where, lib().popsift_hip_last_error(self._h).decode())

    def gauss_table(self):
        n = C.c_int()
        self._chk(lib().popsift_hip_get_gauss_table(self._h, None, None, None, C.byref(n)), "get_gauss_table")
        f = np.zeros((n.value, 32), np.float32)
        s = np.zeros(n.value, np.int32)
        g = np.zeros(n.value, np.float32)
        self._chk(lib().popsift_hip_get_gauss_table(self._h, f.ctypes.data, s.ctypes.data, g.ctypes.data,
                                                    C.byref(n)), "get_gauss_table")
        return f, s, g

    def submit(self, img):
        img = np.ascontiguousarray(img)
        if img.ndim != 2:
            raise ValueError("grayscale (H, W) image expected")
        h, w = img.shape
        if img.dtype == np.uint8:
            rc = lib().popsift_hip_submit_u8(self._h, img.ctypes.data, w, h, w)
        elif img.dtype == np.float32:
            rc = lib().popsift_hip_submit_f32(self._h, img.ctypes.data, w, h, w)
        else:
            raise TypeError("uint8 or float32 image expected, got %s" % img.dtype)
        self._chk(rc, "popsift_hip_submit")
        return self

    def submit_pinned(self, ptr, w, h, pitch, is_f32=False):
        """image in page-locked host memory (host_alloc), uploaded without a staging copy; keep it until wait()"""
        fn = lib().popsift_hip_submit_pinned_f32 if is_f32 else lib().popsift_hip_submit_pinned_u8
        self._chk(fn(self._h, ptr, w, h, pitch), "popsift_hip_submit_pinned")
        return self

    def submit_dev(self, ptr, w, h, pitch, is_f32=False):
        fn = lib().popsift_hip_submit_dev_f32 if is_f32 else lib().popsift_hip_submit_dev_u8
        self._chk(fn(self._h, ptr, w, h, pitch), "popsift_hip_submit_dev")
        return self

    def submit_batch(self, imgs):
        """popsift_hip_submit_batch: several host images of one size and dtype, extracted together"""
        imgs = [np.ascontiguousarray(im) for im in imgs]
        h, w = imgs[0].shape
        if any(im.shape != (h, w) or im.dtype != imgs[0].dtype for im in imgs):
            raise ValueError("the images of a batch share one size and dtype")
        if imgs[0].dtype == np.uint8:
            kind = IMG_HOST_U8
        elif imgs[0].dtype == np.float32:
            kind = IMG_HOST_F32
        else:
            raise TypeError("uint8 or float32 images expected, got %s" % imgs[0].dtype)
        arr = (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
        self._chk(lib().popsift_hip_submit_batch(self._h, arr, len(imgs), kind, w, h, w), "popsift_hip_submit_batch")
        return self

    def submit_batch_dev(self, ptrs, w, h, pitch, is_f32=False):
        """the same for images resident in this device's memory (ptrs: device addresses)"""
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        self._chk(lib().popsift_hip_submit_batch(self._h, arr, len(ptrs), IMG_DEV_F32 if is_f32 else IMG_DEV_U8, w, h, pitch),
                  "popsift_hip_submit_batch")
        return self

    def wait_batch(self):
        """-> [(features, descriptors)] per image of the batch"""
        n = C.c_int()
        nf = (C.c_int * MAX_BATCH)()
        nd = (C.c_int * MAX_BATCH)()
        self._chk(lib().popsift_hip_wait_batch(self._h, C.byref(n), nf, nd), "popsift_hip_wait_batch")
        return [(nf[k], nd[k]) for k in range(n.value)]

    def fetch_item(self, k):
        nf, nd = self.wait_batch()[k]
        feats = np.zeros(nf, FEATURE_DTYPE)
        desc = np.zeros((nd, 128), np.float32)
        self._chk(lib().popsift_hip_fetch_item(self._h, k, feats.ctypes.data, nf, desc.ctypes.data, nd * 128),
                  "popsift_hip_fetch_item")
        return feats, desc

    def wait(self):
        a, b = C.c_int(), C.c_int()
        self._chk(lib().popsift_hip_wait(self._h, C.byref(a), C.byref(b)), "popsift_hip_wait")
        return a.value, b.value

    def fetch(self):
        nf, nd = self.wait()
        feats = np.zeros(nf, FEATURE_DTYPE)
        desc = np.zeros((nd, 128), np.float32)
        self._chk(lib().popsift_hip_fetch(self._h, feats.ctypes.data, nf, desc.ctypes.data, nd * 128),
                  "popsift_hip_fetch")
        return feats, desc

    def fetch_begin(self, pinned=True):
        """Start the download of the finished image and return a handle; the context is free for the next submit.
        handle.result() (popsift_hip_fetch_end) -> (feats, desc).  pinned: page-locked targets (asynchronous copy)."""
        nf, nd = self.wait()
        prev = getattr(self, "_pending", None)
        self._pending = PendingFetch(self, nf, nd, pinned)   # the C call waits for an earlier pending download first
        if prev is not None and not prev._done:
            prev._landed()
        return self._pending

    def report(self):
        r = Report()
        self._chk(lib().popsift_hip_get_report(self._h, C.byref(r)), "popsift_hip_get_report")
        return r

    def set_profile(self, mode):
        """0 off, 1 (True): every blur launch timed, 2: stage times (report().ms_stage)"""
        self._chk(lib().popsift_hip_set_profile(self._h, int(mode)), "popsift_hip_set_profile")

    def octave_dims(self, o):
        w, h = C.c_int(), C.c_int()
        self._chk(lib().popsift_hip_octave_dims(self._h, o, C.byref(w), C.byref(h)), "popsift_hip_octave_dims")
        return w.value, h.value

    def plane(self, octave, kind, level):
        w, h = self.octave_dims(octave)
        out = np.zeros((h, w), np.float32)
        self._chk(lib().popsift_hip_download_plane(self._h, octave, kind, level, out.ctypes.data), "download_plane")
        return out

    def upload_plane(self, octave, kind, level, arr):
        w, h = self.octave_dims(octave)
        arr = np.ascontiguousarray(arr, np.float32)
        assert arr.shape == (h, w)
        self._chk(lib().popsift_hip_upload_plane(self._h, octave, kind, level, arr.ctypes.data), "upload_plane")

    def extrema(self):
        n = C.c_int()
        self._chk(lib().popsift_hip_download_extrema(self._h, None, 0, C.byref(n)), "download_extrema")
        out = np.zeros(n.value, EXTREMUM_DTYPE)
        self._chk(lib().popsift_hip_download_extrema(self._h, out.ctypes.data, n.value, C.byref(n)),
                  "download_extrema")
        return out

    def clone_results(self):
        h = _vp()
        self._chk(lib().popsift_hip_clone_results(self._h, C.byref(h)), "popsift_hip_clone_results")
        return DevFeatures(h)

    def rerun_keypoint_stages(self):
        self._chk(lib().popsift_hip_rerun_keypoint_stages(self._h), "rerun_keypoint_stages")
        return self

    def debug_set(self, what, value):
        """popsift_hip_debug_set: test switches (DEBUG_*), before the first submit"""
        self._chk(lib().popsift_hip_debug_set(self._h, what, value), "popsift_hip_debug_set")
        return self

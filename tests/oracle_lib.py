"""ctypes front for oracle/liborb_oracle.so (TEST INFRASTRUCTURE — never imported by the product)."""
import ctypes
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28
HARRIS_SCORE, FAST_SCORE = 0, 1
_LIB = None
c_int, c_float, c_double, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_void_p


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liborb_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liborb_oracle.so missing: run `make`")
        L = ctypes.CDLL(path)
        L.orc_create.restype = c_void_p
        L.orc_create.argtypes = [c_int, c_float, c_int, c_int, c_int, c_int]
        L.orc_destroy.argtypes = [c_void_p]
        L.orc_keep_dumps.argtypes = [c_void_p, c_int]
        L.orc_extract.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int]
        L.orc_features_per_level.argtypes = [c_void_p, c_int]
        L.orc_scale_factor.argtypes = [c_void_p, c_int]
        L.orc_scale_factor.restype = c_float
        L.orc_inv_scale_factor.argtypes = [c_void_p, c_int]
        L.orc_inv_scale_factor.restype = c_float
        L.orc_umax.argtypes = [c_void_p, c_int]
        L.orc_level_size.argtypes = [c_void_p, c_int, c_void_p, c_void_p]
        L.orc_level_plane.argtypes = [c_void_p, c_int, c_int, c_void_p]
        L.orc_level_keypoints.argtypes = [c_void_p, c_int, c_void_p, c_int]
        L.orc_num_cells.argtypes = [c_void_p]
        L.orc_cell.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int]
        L.orc_cvRound.argtypes = [c_double]
        L.orc_cvFloor.argtypes = [c_double]
        L.orc_cvCeil.argtypes = [c_double]
        L.orc_fastAtan2.argtypes = [c_float, c_float]
        L.orc_fastAtan2.restype = c_float
        L.orc_reflect101.argtypes = [c_int, c_int]
        L.orc_gaussian_kernel_q8.argtypes = [c_void_p]
        L.orc_resize_linear_8u.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int]
        L.orc_fast.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]
        L.orc_gaussian_blur7.argtypes = [c_void_p, c_int, c_int, c_int]
        L.orc_retain_best.argtypes = [c_void_p, c_int, c_int, c_void_p]
        L.orc_sincosf.argtypes = [c_float, c_void_p, c_void_p]
        L.orc_nth_element_perm.argtypes = [c_void_p, c_int, c_int, c_void_p]
        L.orc_hamming256.argtypes = [c_void_p, c_void_p]
        L.orc_match_top2.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]
        L.orc_count_accepted.argtypes = [c_void_p, c_void_p, c_int, c_int, c_float]
        L.orc_match_top2_segments.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        _LIB = L
    return _LIB


class OracleExtractor:
    """Mirror of the reference constructor ORBextractor(nfeatures, scaleFactor, nlevels, scoreType, fastTh)."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, scoreType=FAST_SCORE, fastTh=20, blur_mode=0, dumps=False):
        self.L = lib()
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.h = self.L.orc_create(nfeatures, scaleFactor, nlevels, scoreType, fastTh, blur_mode)
        assert self.h
        self.L.orc_keep_dumps(self.h, 1 if dumps else 0)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def __call__(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        hh, w = img.shape
        cap = max(2 * self.nfeatures, 16)
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = self.L.orc_extract(self.h, img.ctypes.data, w, hh, img.strides[0], kps.ctypes.data, desc.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("oracle extract error %d" % n)
        return kps[:n].copy(), desc[:n].copy()

    def features_per_level(self):
        return [self.L.orc_features_per_level(self.h, l) for l in range(self.nlevels)]

    def scale_factors(self):
        return np.array([self.L.orc_scale_factor(self.h, l) for l in range(self.nlevels)], dtype=np.float32)

    def umax(self):
        return [self.L.orc_umax(self.h, v) for v in range(16)]

    def level_size(self, level):
        w, hh = c_int(), c_int()
        self.L.orc_level_size(self.h, level, ctypes.byref(w), ctypes.byref(hh))
        return w.value, hh.value

    def level_plane(self, level, which=0):
        w, hh = self.level_size(level)
        shape = (hh + 32, w + 32) if which == 2 else (hh, w)
        out = np.empty(shape, dtype=np.uint8)
        self.L.orc_level_plane(self.h, level, which, out.ctypes.data)
        return out

    def level_keypoints(self, level):
        cap = 4 * self.nfeatures + 16
        out = np.zeros(cap, dtype=KP_DTYPE)
        n = self.L.orc_level_keypoints(self.h, level, out.ctypes.data, cap)
        assert n >= 0
        return out[:n].copy()

    def cells(self):
        res = []
        for i in range(self.L.orc_num_cells(self.h)):
            info = (c_int * 8)()
            n = self.L.orc_cell(self.h, i, info, None, 0)
            out = np.zeros(max(n, 1), dtype=KP_DTYPE)
            self.L.orc_cell(self.h, i, info, out.ctypes.data, max(n, 1))
            res.append((tuple(info), out[:n].copy()))
        return res


def hamming256(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return lib().orc_hamming256(a.ctypes.data, b.ctypes.data)


def match_top2(Q, T):
    Q = np.ascontiguousarray(Q, dtype=np.uint8).reshape(-1, 32)
    T = np.ascontiguousarray(T, dtype=np.uint8).reshape(-1, 32)
    nq, nt = len(Q), len(T)
    idx = np.empty(nq, np.int32); best = np.empty(nq, np.int32); sec = np.empty(nq, np.int32)
    lib().orc_match_top2(Q.ctypes.data, nq, T.ctypes.data, nt, idx.ctypes.data, best.ctypes.data, sec.ctypes.data)
    return idx, best, sec


def fast(img, threshold, want_scores=False):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    hh, w = img.shape
    cap = w * hh // 4 + 16
    out = np.zeros(cap, dtype=KP_DTYPE)
    sc = np.zeros((hh, w), dtype=np.uint8) if want_scores else None
    n = lib().orc_fast(img.ctypes.data, w, hh, img.strides[0], threshold, out.ctypes.data, cap,
                       sc.ctypes.data if want_scores else None)
    assert n >= 0
    return (out[:n].copy(), sc) if want_scores else out[:n].copy()


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw = src.shape
    dst = np.empty((dh, dw), dtype=np.uint8)
    lib().orc_resize_linear_8u(src.ctypes.data, sw, sh, src.strides[0], dst.ctypes.data, dw, dh, dw)
    return dst


def gaussian_blur7(img, blur_mode=0):
    out = np.ascontiguousarray(img, dtype=np.uint8).copy()
    lib().orc_gaussian_blur7(out.ctypes.data, out.shape[1], out.shape[0], blur_mode)
    return out


def retain_best(responses, n):
    r = np.ascontiguousarray(responses, dtype=np.float32)
    out = np.empty(max(n, 1), dtype=np.int32)
    m = lib().orc_retain_best(r.ctypes.data, len(r), n, out.ctypes.data)
    return out[:m].copy()


def nth_element_perm(responses, nth):
    r = np.ascontiguousarray(responses, dtype=np.float32)
    out = np.empty(len(r), dtype=np.int32)
    lib().orc_nth_element_perm(r.ctypes.data, len(r), nth, out.ctypes.data)
    return out


def match_top2_segments(Q, T, seg_off, cand):
    Q = np.ascontiguousarray(Q, dtype=np.uint8).reshape(-1, 32)
    T = np.ascontiguousarray(T, dtype=np.uint8).reshape(-1, 32)
    seg = np.ascontiguousarray(seg_off, dtype=np.int32)
    cd = np.ascontiguousarray(cand, dtype=np.int32)
    nq = len(Q)
    idx = np.empty(nq, np.int32); best = np.empty(nq, np.int32); sec = np.empty(nq, np.int32)
    lib().orc_match_top2_segments(Q.ctypes.data, nq, T.ctypes.data, len(T), seg.ctypes.data, cd.ctypes.data if len(cd) else None,
                                  idx.ctypes.data, best.ctypes.data, sec.ctypes.data)
    return idx, best, sec

"""ctypes binding of the C ABI in include/orbx.h (liborbx.so).  Plumbing only: tests and bench.py
drive the library through this module; the drop-in for ORB_SLAM itself is the C++ shim in
orb_slam_amd/cpp/ (ORBextractor.h / ORBmatcher.h) which calls the same C ABI.

There is no CPU fallback anywhere in this module: if liborbx.so is missing or no GPU is usable,
calls raise OrbxError."""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORBX_LIB") or os.path.join(_HERE, "liborbx.so")      # ORBX_LIB: a differently built library (tuning sweeps)

ORBX_OK, ORBX_EMPTY = 0, 1
ORBX_ERR_ARG, ORBX_ERR_DEVICE, ORBX_ERR_CAPACITY, ORBX_ERR_GEOMETRY = -1, -2, -3, -4
HARRIS_SCORE, FAST_SCORE = 0, 1
BLUR_X86_SSE2, BLUR_HALF_UP = 0, 1
DBG_PLANE, DBG_BLUR, DBG_NMS, DBG_LEVEL_KPS, DBG_BANDS = 0, 1, 2, 3, 4
(ST_PYRAMID, ST_FAST_CELLS, ST_QUOTA, ST_CELL_SELECT, ST_LEVEL_SELECT, ST_BLUR, ST_DESCRIBE) = range(7)

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

# every symbol include/orbx.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "orbx_default_params", "orbx_create", "orbx_destroy", "orbx_get_levels", "orbx_get_scale_factor",
    "orbx_max_keypoints", "orbx_last_error", "orbx_build_id", "orbx_extract", "orbx_extract_batch_device", "orbx_extract_batch_device_phases",
    "orbm_hamming256", "orbm_match_top2", "orbm_match_top2_device", "orbm_match_top2_batch_device", "orbm_match_top2_masked", "orbm_match_top2_masked_device",
    "orbm_count_accepted", "orbm_match_top2_segments", "orbm_match_top2_segments_device", "orbm_distinctive", "orbm_distinctive_device",
    "orbx_device_alloc", "orbx_device_free", "orbx_device_upload", "orbx_device_download",
    "orbx_stream_create", "orbx_stream_create_priority", "orbx_stream_destroy", "orbx_stream_synchronize", "orbx_event_create", "orbx_event_destroy", "orbx_event_record",
    "orbx_stream_wait_event", "orbx_device_copy_async", "orbx_host_alloc", "orbx_host_free", "orbx_device_upload_async", "orbx_device_download_async", "orbx_debug_set_stop_after", "orbx_debug_set_blur_on_demand", "orbx_debug_level_size", "orbx_debug_fetch",
    "orbx_debug_eval_math", "orbx_debug_stage_timing", "orbx_debug_stage_time", "orbx_debug_nth_element", "orbx_debug_geometry",
    "orbm_debug_set_match_path",
    "orbm_debug_get_match_path",
]
# include/orbf.h (Frame-side steps: undistortion, search grid, window query)
EXPORTS_F = [
    "orbf_image_bounds", "orbf_undistort_grid", "orbf_undistort_grid_batch_device", "orbf_features_in_area",
    "orbf_features_in_area_device",
]
# include/orbs.h (greedy grid-window searches)
EXPORTS_S = ["orbs_lds_bytes", "orbs_debug_set_buckets", "orbs_debug_set_wide_max", "orbs_three_maxima", "orbs_window_search_batch_device", "orbs_list_search_batch_device",
             "orbs_bow_ranges_batch_device", "orbs_triangulation_search_batch_device", "orbs_epipolar_bound", "orbs_agreement_batch_device"]
PHASE_PYRAMID, PHASE_DETECT, PHASE_DESCRIBE, PHASE_ALL = 1, 2, 4, 7
RULE_MAPPOINTS, RULE_WINDOW, RULE_BEST, RULE_INIT, RULE_BOW, RULE_FREE, RULE_TRIANGULATION = 0, 1, 2, 3, 4, 5, 6
TH_HIGH, TH_LOW = 100, 50
# include/orbv.h (bag-of-words transform)
EXPORTS_V = [
    "orbv_create", "orbv_load_text", "orbv_destroy", "orbv_info", "orbv_descend", "orbv_descend_device",
    "orbv_transform", "orbv_transform_batch_device", "orbv_score",
]


class OrbxError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("orbx error %d %s" % (code, msg))
        self.code = code


class Params(ctypes.Structure):
    _fields_ = [("nfeatures", ctypes.c_int32), ("scale_factor", ctypes.c_float), ("nlevels", ctypes.c_int32),
                ("score_type", ctypes.c_int32), ("fast_th", ctypes.c_int32), ("device", ctypes.c_int32),
                ("max_batch", ctypes.c_int32), ("blur_rounding", ctypes.c_int32), ("fp_contract", ctypes.c_int32), ("reserved", ctypes.c_int32 * 7)]


class Camera(ctypes.Structure):
    """orbf_camera: mK (row-major 3x3), mDistCoef, image size"""
    _fields_ = [("K", ctypes.c_float * 9), ("dist", ctypes.c_float * 8), ("ndist", ctypes.c_int32),
                ("width", ctypes.c_int32), ("height", ctypes.c_int32)]

    @classmethod
    def make(cls, fx, fy, cx, cy, dist, width, height):
        c = cls()
        for i, v in enumerate((fx, 0, cx, 0, fy, cy, 0, 0, 1)):
            c.K[i] = v
        for i, v in enumerate(dist):
            c.dist[i] = v
        c.ndist, c.width, c.height = len(dist), width, height
        return c


class Bounds(ctypes.Structure):
    """orbf_bounds: Frame::mnMinX/mnMaxX/mnMinY/mnMaxY and the inverse grid cell sizes"""
    _fields_ = [("min_x", ctypes.c_int32), ("max_x", ctypes.c_int32), ("min_y", ctypes.c_int32), ("max_y", ctypes.c_int32),
                ("inv_w", ctypes.c_float), ("inv_h", ctypes.c_float)]

    def astuple(self):
        return (self.min_x, self.max_x, self.min_y, self.max_y, self.inv_w, self.inv_h)


class SearchParams(ctypes.Structure):
    """orbs_params"""
    _fields_ = [("rule", ctypes.c_int32), ("th", ctypes.c_int32), ("ratio", ctypes.c_float), ("check_orientation", ctypes.c_int32)]


GRID_COLS, GRID_ROWS = 64, 48
GRID_CELLS = GRID_COLS * GRID_ROWS

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise OrbxError(ORBX_ERR_DEVICE, "liborbx.so not built: run `make` / __graft_entry__.build()")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  If torch
        # is imported first, liborbx binds to that copy and torch streams / device pointers are directly usable;
        # the other order loads a second runtime next to torch's and torch then sees no GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        vp, ci, cl, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
        pd = ctypes.c_ssize_t
        L.orbx_default_params.argtypes = [ctypes.POINTER(Params)]
        L.orbx_default_params.restype = None
        L.orbx_create.argtypes = [ctypes.POINTER(Params), ctypes.POINTER(vp)]
        L.orbx_destroy.argtypes = [vp]
        L.orbx_destroy.restype = None
        L.orbx_get_levels.argtypes = [vp]
        L.orbx_get_scale_factor.argtypes = [vp]
        L.orbx_get_scale_factor.restype = cf
        L.orbx_max_keypoints.argtypes = [vp]
        L.orbx_last_error.argtypes = [vp]
        L.orbx_last_error.restype = ctypes.c_char_p
        L.orbx_build_id.argtypes = []
        L.orbx_build_id.restype = ctypes.c_char_p
        L.orbx_extract.argtypes = [vp, vp, ci, ci, pd, vp, vp, ci, ctypes.POINTER(ci)]
        L.orbx_extract_batch_device.argtypes = [vp, vp, ci, ci, ci, pd, pd, vp, vp, vp, ci, vp, vp]
        L.orbx_extract_batch_device_phases.argtypes = [vp, vp, ci, ci, ci, pd, pd, vp, vp, vp, ci, vp, vp, ci]
        L.orbm_hamming256.argtypes = [vp, vp]
        L.orbm_match_top2.argtypes = [vp, ci, vp, ci, vp, vp, vp, ci]
        L.orbm_match_top2_device.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp]
        L.orbm_match_top2_batch_device.argtypes = [vp, vp, vp, vp, ci, ci, vp, vp, vp, vp]
        L.orbm_match_top2_masked.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, ci]
        L.orbm_match_top2_masked_device.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp]
        L.orbm_count_accepted.argtypes = [vp, vp, ci, ci, cf]
        L.orbm_debug_set_match_path.argtypes = [ci]
        L.orbm_debug_get_match_path.argtypes = []
        L.orbs_debug_set_buckets.argtypes = [ci]
        L.orbs_debug_set_wide_max.argtypes = [ci]
        L.orbm_match_top2_segments.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, ci]
        L.orbm_match_top2_segments_device.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, vp]
        L.orbx_device_alloc.argtypes = [ci, ctypes.c_size_t, ctypes.POINTER(vp)]
        L.orbx_device_free.argtypes = [ci, vp]
        L.orbx_device_upload.argtypes = [ci, vp, vp, ctypes.c_size_t]
        L.orbx_device_download.argtypes = [ci, vp, vp, ctypes.c_size_t]
        L.orbm_distinctive.argtypes = [vp, vp, ci, vp, vp, ci]
        L.orbm_distinctive_device.argtypes = [vp, vp, ci, vp, vp, vp]
        L.orbx_debug_set_stop_after.argtypes = [vp, ci]
        L.orbx_debug_set_blur_on_demand.argtypes = [vp, ci]
        L.orbx_debug_level_size.argtypes = [vp, ci, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.orbx_debug_fetch.argtypes = [vp, ci, ci, ci, vp, cl]
        L.orbx_debug_fetch.restype = cl
        L.orbx_debug_eval_math.argtypes = [ci, vp, vp, vp, vp, ci, ci]
        L.orbx_debug_nth_element.argtypes = [vp, ci, ci, vp, ci]
        L.orbx_debug_geometry.argtypes = [ctypes.POINTER(Params), ci, ci, vp, ci]
        L.orbx_debug_stage_timing.argtypes = [vp, ci]
        L.orbx_debug_stage_time.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(cl)]
        L.orbf_image_bounds.argtypes = [ctypes.POINTER(Camera), ctypes.POINTER(Bounds)]
        L.orbf_undistort_grid.argtypes = [ctypes.POINTER(Camera), ctypes.POINTER(Bounds), vp, ci, vp, vp, vp, ci]
        L.orbf_undistort_grid_batch_device.argtypes = [ctypes.POINTER(Camera), ctypes.POINTER(Bounds), vp, vp, ci, ci, vp, vp, vp, vp]
        L.orbf_features_in_area.argtypes = [ctypes.POINTER(Bounds), vp, ci, vp, vp, vp, vp, ci, vp, vp, ci, ci]
        L.orbf_features_in_area_device.argtypes = [ctypes.POINTER(Bounds), vp, ci, vp, vp, vp, vp, ci, vp, vp, ci, vp, vp]
        L.orbs_lds_bytes.argtypes = [ci, ci]
        L.orbs_lds_bytes.restype = ctypes.c_size_t
        L.orbs_three_maxima.argtypes = [vp, ci, vp]
        L.orbs_three_maxima.restype = None
        L.orbs_window_search_batch_device.argtypes = [ctypes.POINTER(Bounds), ctypes.POINTER(SearchParams), vp, vp, vp, vp, vp, ci, vp,
                                                      vp, vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp]
        L.orbs_list_search_batch_device.argtypes = [ctypes.POINTER(SearchParams), vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, ci,
                                                    vp, vp, vp, vp, vp, vp]
        L.orbs_bow_ranges_batch_device.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, vp, vp, vp]
        L.orbs_triangulation_search_batch_device.argtypes = [ctypes.POINTER(SearchParams), vp, vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp,
                                                             vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp]
        L.orbs_epipolar_bound.argtypes = [ctypes.c_float]
        L.orbs_epipolar_bound.restype = ctypes.c_float
        L.orbs_agreement_batch_device.argtypes = [vp, vp, ci, vp, vp, ci, ci, vp, vp, vp]
        L.orbv_create.argtypes = [ci, ci, ci, ci, ci, vp, vp, vp, vp, ci, ctypes.POINTER(vp)]
        L.orbv_load_text.argtypes = [ctypes.c_char_p, ci, ctypes.POINTER(vp)]
        L.orbv_destroy.argtypes = [vp]
        L.orbv_destroy.restype = None
        L.orbv_info.argtypes = [vp] + [ctypes.POINTER(ci)] * 6
        L.orbv_descend.argtypes = [vp, vp, ci, ci, vp, vp, vp]
        L.orbv_descend_device.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp]
        L.orbv_transform.argtypes = [vp, vp, ci, ci, vp, vp, ctypes.POINTER(ci), vp, vp, vp, ctypes.POINTER(ci)]
        L.orbv_transform_batch_device.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]
        L.orbv_score.argtypes = [vp, vp, vp, ci, vp, vp, ci]
        L.orbv_score.restype = ctypes.c_double
        _LIB = L
    return _LIB


class ORBextractor:
    """Same constructor arguments as the reference ORBextractor(nfeatures, scaleFactor, nlevels, scoreType, fastTh)
    (include/ORBextractor.h:38) plus device placement; __call__(image) is operator()."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, scoreType=FAST_SCORE, fastTh=20,
                 device=0, max_batch=1, blur_rounding=BLUR_X86_SSE2, fp_contract=None):
        L = lib()
        p = Params()
        L.orbx_default_params(ctypes.byref(p))
        p.nfeatures, p.scale_factor, p.nlevels, p.score_type, p.fast_th = nfeatures, scaleFactor, nlevels, scoreType, fastTh
        p.device, p.max_batch, p.blur_rounding = device, max_batch, blur_rounding
        if fp_contract is not None:                      # None: orbx_default_params' choice (ORBX_FP_CONTRACT in the environment, else ISO)
            p.fp_contract = 1 if fp_contract else 0
        h = ctypes.c_void_p()
        rc = L.orbx_create(ctypes.byref(p), ctypes.byref(h))
        if rc != ORBX_OK:
            raise OrbxError(rc, "orbx_create (is a gfx950 GPU visible?)")
        self.h = h
        self.L = L
        self.max_keypoints = L.orbx_max_keypoints(h)

    def close(self):
        if getattr(self, "h", None):
            self.L.orbx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def GetLevels(self):
        return self.L.orbx_get_levels(self.h)

    def GetScaleFactor(self):
        return self.L.orbx_get_scale_factor(self.h)

    def _err(self, rc):
        return OrbxError(rc, self.L.orbx_last_error(self.h).decode())

    def __call__(self, image):
        """image: 2-D uint8 array.  Returns (keypoints[N] (KP_DTYPE), descriptors[N,32] uint8)."""
        img = np.asarray(image)
        if img.size == 0:
            return None   # reference: silent return, outputs untouched
        assert img.dtype == np.uint8 and img.ndim == 2
        if img.strides[1] != 1:
            img = np.ascontiguousarray(img)
        hh, w = img.shape
        cap = self.max_keypoints
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = ctypes.c_int(0)
        rc = self.L.orbx_extract(self.h, img.ctypes.data, w, hh, img.strides[0], kps.ctypes.data, desc.ctypes.data, cap, ctypes.byref(n))
        if rc != ORBX_OK:
            raise self._err(rc)
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch_device(self, d_imgs, nframes, w, h, row_stride, frame_stride, d_kps, d_desc, d_n, cap, d_status=0, stream=0, phases=PHASE_ALL):
        """All pointer arguments are integer device addresses (e.g. torch tensor.data_ptr()).  Asynchronous.
        phases: PHASE_* bit mask (orbx_extract_batch_device_phases); the parts of one batch go to one stream, in order."""
        rc = self.L.orbx_extract_batch_device_phases(self.h, d_imgs, nframes, w, h, row_stride, frame_stride, d_kps, d_desc, d_n, cap,
                                                     d_status or None, stream or None, phases)
        if rc != ORBX_OK:
            raise self._err(rc)

    # diagnostics
    def set_stop_after(self, stage):
        self.L.orbx_debug_set_stop_after(self.h, stage)

    def set_blur_on_demand(self, mode):
        """1: the blur per keypoint window inside the description kernel (full launch groups); 0: blur kernels + blurred plane"""
        rc = self.L.orbx_debug_set_blur_on_demand(self.h, int(mode))
        if rc != ORBX_OK:
            raise OrbxError(rc, "orbx_debug_set_blur_on_demand")

    STAGE_NAMES = ["pyramid", "fast_cells", "quota", "cell_select", "level_select", "blur", "describe"]

    def stage_timing(self, enable):
        """0 off, 1 on, 2 on + reset"""
        self.L.orbx_debug_stage_timing(self.h, enable)

    def stage_times(self):
        """-> {stage: (total_ms, launch_groups)} measured with HIP events on the launch stream"""
        res = {}
        for i, name in enumerate(self.STAGE_NAMES):
            ms, n = ctypes.c_double(), ctypes.c_long()
            self.L.orbx_debug_stage_time(self.h, i, ctypes.byref(ms), ctypes.byref(n))
            res[name] = (ms.value, n.value)
        return res

    def level_size(self, level):
        w, hh = ctypes.c_int(), ctypes.c_int()
        rc = self.L.orbx_debug_level_size(self.h, level, ctypes.byref(w), ctypes.byref(hh))
        if rc != ORBX_OK:
            raise self._err(rc)
        return w.value, hh.value

    def fetch_plane(self, what, level, frame=0):
        w, hh = self.level_size(level)
        out = np.empty((hh, w), dtype=np.uint8)
        rc = self.L.orbx_debug_fetch(self.h, what, frame, level, out.ctypes.data, out.nbytes)
        if rc < 0:
            raise self._err(rc)
        return out

    def fetch_bands(self, level, frame=0):
        """the FAST work items of a level after a run: int32 rows (x0, x1, y0, y1, n_all, n_hi, n_lo, list threshold)"""
        out = np.zeros((16384, 8), dtype=np.int32)
        rc = self.L.orbx_debug_fetch(self.h, DBG_BANDS, frame, level, out.ctypes.data, out.nbytes)
        if rc < 0:
            raise self._err(rc)
        return out[:rc // 32].copy()

    def fetch_level_keypoints(self, level, frame=0):
        out = np.zeros((4 * self.max_keypoints + 64, 3), dtype=np.int32)
        rc = self.L.orbx_debug_fetch(self.h, DBG_LEVEL_KPS, frame, level, out.ctypes.data, out.nbytes)
        if rc < 0:
            raise self._err(rc)
        n = rc // 12
        xy = out[:n, :2].copy()
        resp = out[:n, 2].copy().view(np.float32)
        return xy, resp


def hamming256(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    assert a.size == 32 and b.size == 32
    return lib().orbm_hamming256(a.ctypes.data, b.ctypes.data)


def match_top2(Q, T, device=0):
    """Host arrays [nq,32], [nt,32] uint8 -> (best_idx, best, second) int32 arrays (GPU computed)."""
    Q = np.ascontiguousarray(Q, dtype=np.uint8).reshape(-1, 32)
    T = np.ascontiguousarray(T, dtype=np.uint8).reshape(-1, 32)
    nq, nt = len(Q), len(T)
    idx = np.empty(nq, np.int32)
    best = np.empty(nq, np.int32)
    sec = np.empty(nq, np.int32)
    rc = lib().orbm_match_top2(Q.ctypes.data, nq, T.ctypes.data, nt, idx.ctypes.data, best.ctypes.data, sec.ctypes.data, device)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbm_match_top2")
    return idx, best, sec


def match_top2_masked(Q, T, t_valid, device=0):
    """dense top-2 over the train descriptors with t_valid != 0 only (src/ORBmatcher.cc:205-206); indices refer to T"""
    Q = np.ascontiguousarray(Q, dtype=np.uint8).reshape(-1, 32)
    T = np.ascontiguousarray(T, dtype=np.uint8).reshape(-1, 32)
    v = np.ascontiguousarray(t_valid, dtype=np.uint8)
    nq, nt = len(Q), len(T)
    assert len(v) == nt
    idx = np.empty(nq, np.int32); best = np.empty(nq, np.int32); sec = np.empty(nq, np.int32)
    rc = lib().orbm_match_top2_masked(Q.ctypes.data, nq, T.ctypes.data, nt, v.ctypes.data if nt else None, idx.ctypes.data, best.ctypes.data, sec.ctypes.data, device)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbm_match_top2_masked")
    return idx, best, sec


def match_top2_segments(Q, T, seg_off, cand, device=0):
    """per-query candidate lists (CSR): query q scans T[cand[seg_off[q]:seg_off[q+1]]] in list order"""
    Q = np.ascontiguousarray(Q, dtype=np.uint8).reshape(-1, 32)
    T = np.ascontiguousarray(T, dtype=np.uint8).reshape(-1, 32)
    seg = np.ascontiguousarray(seg_off, dtype=np.int32)
    cd = np.ascontiguousarray(cand, dtype=np.int32)
    nq = len(Q)
    assert len(seg) == nq + 1
    idx = np.empty(nq, np.int32); best = np.empty(nq, np.int32); sec = np.empty(nq, np.int32)
    rc = lib().orbm_match_top2_segments(Q.ctypes.data, nq, T.ctypes.data, len(T), seg.ctypes.data, cd.ctypes.data if len(cd) else None,
                                        idx.ctypes.data, best.ctypes.data, sec.ctypes.data, device)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbm_match_top2_segments")
    return idx, best, sec


def build_id():
    """hash of the kernel sources the loaded library was built from (Makefile -> orbx_build_id)"""
    return lib().orbx_build_id().decode()


def match_top2_device(dQ, nq, dT, nt, d_idx, d_best, d_second, stream=0):
    rc = lib().orbm_match_top2_device(dQ, nq, dT, nt, d_idx, d_best, d_second, stream or None)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbm_match_top2_device")


def match_top2_batch_device(dQ, d_nq, dT, d_nt, nbatch, cap, d_idx, d_best, d_second, stream=0):
    rc = lib().orbm_match_top2_batch_device(dQ, d_nq, dT, d_nt, nbatch, cap, d_idx, d_best, d_second, stream or None)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbm_match_top2_batch_device")


def get_match_path():
    """the dense top-2 kernels in effect: 0 xor + popcount, 1 int8 MFMA, 2 FP4 MFMA"""
    return int(lib().orbm_debug_get_match_path())


def set_match_path(path):
    """test hook: -1 process default (ORBX_MATCH_MFMA = 0 / 8 / 4), 0 xor + popcount kernels, 1 int8 MFMA kernels, 2 FP4 MFMA kernels"""
    rc = lib().orbm_debug_set_match_path(path)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbm_debug_set_match_path")


def set_search_buckets(mode):
    """test hook: -1 process default (ORBS_BUCKETS), 0 plain CSR scan, 1 bucketed index where it fits"""
    rc = lib().orbs_debug_set_buckets(mode)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbs_debug_set_buckets")


def set_search_wide_max(nproblems):
    """test hook: launches of up to `nproblems` problems take the 1024-thread (latency) form of the search kernel; -2 = process default (64), 0 = never"""
    rc = lib().orbs_debug_set_wide_max(nproblems)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbs_debug_set_wide_max")


def count_accepted(best, second, th=50, ratio=0.6):
    best = np.ascontiguousarray(best, dtype=np.int32)
    second = np.ascontiguousarray(second, dtype=np.int32)
    return lib().orbm_count_accepted(best.ctypes.data, second.ctypes.data, len(best), th, ratio)


def geometry(w, h, nfeatures=1000, scaleFactor=1.2, nlevels=8, scoreType=FAST_SCORE, fastTh=20):
    """host-side geometry (no GPU needed): list of dicts per level, or raises OrbxError"""
    L = lib()
    p = Params()
    L.orbx_default_params(ctypes.byref(p))
    p.nfeatures, p.scale_factor, p.nlevels, p.score_type, p.fast_th = nfeatures, scaleFactor, nlevels, scoreType, fastTh
    out = np.zeros((16, 8), np.int32)
    rc = L.orbx_debug_geometry(ctypes.byref(p), w, h, out.ctypes.data, 16)
    if rc < 0:
        raise OrbxError(rc, "orbx_debug_geometry")
    keys = ("w", "h", "quota", "grid_cols", "grid_rows", "cell_w", "cell_h", "n_bands")
    return [dict(zip(keys, map(int, out[l]))) for l in range(rc)]


def nth_element_perm(resp, nth, device=0):
    """permutation produced by the device's wave-parallel std::nth_element(greater by response)"""
    r = np.ascontiguousarray(resp, dtype=np.float32)
    out = np.empty(len(r), np.int32)
    rc = lib().orbx_debug_nth_element(r.ctypes.data, len(r), nth, out.ctypes.data, device)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbx_debug_nth_element")
    return out


def eval_math(kind, in0, in1=None, device=0):
    in0 = np.ascontiguousarray(in0, dtype=np.float32)
    in1 = np.ascontiguousarray(in1 if in1 is not None else in0, dtype=np.float32)
    out0 = np.empty_like(in0)
    out1 = np.empty_like(in0)
    rc = lib().orbx_debug_eval_math(kind, in0.ctypes.data, in1.ctypes.data, out0.ctypes.data, out1.ctypes.data, in0.size, device)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbx_debug_eval_math")
    return out0, out1


class ORBVocabulary:
    """Mirror of ORB_SLAM::ORBVocabulary (reference include/ORBVocabulary.h:31-32 = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>)
    for the per-frame transform: construct from a node table (`from_nodes`) or the reference's text file (`loadFromTextFile`)."""

    def __init__(self):
        self.h = ctypes.c_void_p()

    @classmethod
    def from_nodes(cls, k, L, scoring, weighting, parent, is_leaf, desc, weight, device=0):
        parent = np.ascontiguousarray(parent, dtype=np.int32)
        is_leaf = np.ascontiguousarray(is_leaf, dtype=np.uint8)
        desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32)
        weight = np.ascontiguousarray(weight, dtype=np.float64)
        n = len(parent)
        assert len(is_leaf) == n and len(desc) == n and len(weight) == n
        v = cls()
        rc = lib().orbv_create(k, L, scoring, weighting, n, parent.ctypes.data, is_leaf.ctypes.data, desc.ctypes.data, weight.ctypes.data,
                               device, ctypes.byref(v.h))
        if rc != ORBX_OK:
            raise OrbxError(rc, "orbv_create")
        return v

    @classmethod
    def loadFromTextFile(cls, path, device=0):
        v = cls()
        rc = lib().orbv_load_text(os.fsencode(path), device, ctypes.byref(v.h))
        if rc != ORBX_OK:
            raise OrbxError(rc, "orbv_load_text")
        return v

    def close(self):
        if getattr(self, "h", None):
            lib().orbv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except TypeError:           # interpreter shutdown: module globals are already gone
            pass

    def info(self):
        vals = [ctypes.c_int() for _ in range(6)]
        lib().orbv_info(self.h, *[ctypes.byref(x) for x in vals])
        return dict(zip(("k", "L", "scoring", "weighting", "n_words", "n_nodes"), [x.value for x in vals]))

    def size(self):
        return self.info()["n_words"]

    def descend(self, desc, levelsup=4):
        """per-descriptor (word id, weight, node id at level L-levelsup)"""
        desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32)
        n = len(desc)
        word = np.empty(n, np.uint32); weight = np.empty(n, np.float64); node = np.empty(n, np.uint32)
        rc = lib().orbv_descend(self.h, desc.ctypes.data, n, levelsup, word.ctypes.data, weight.ctypes.data, node.ctypes.data)
        if rc != ORBX_OK:
            raise OrbxError(rc, "orbv_descend")
        return word, weight, node

    def transform(self, desc, levelsup=4):
        """Frame::ComputeBoW: -> (bow_ids, bow_vals, fv_nodes, fv_off, fv_feat); BowVector / FeatureVector in map order"""
        desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32)
        n = len(desc)
        m = max(n, 1)
        bid = np.empty(m, np.uint32); bval = np.empty(m, np.float64)
        fnode = np.empty(m, np.uint32); foff = np.zeros(m + 1, np.int32); ffeat = np.empty(m, np.uint32)
        nb, nf = ctypes.c_int(), ctypes.c_int()
        rc = lib().orbv_transform(self.h, desc.ctypes.data, n, levelsup, bid.ctypes.data, bval.ctypes.data, ctypes.byref(nb),
                                  fnode.ctypes.data, foff.ctypes.data, ffeat.ctypes.data, ctypes.byref(nf))
        if rc != ORBX_OK:
            raise OrbxError(rc, "orbv_transform")
        return bid[:nb.value], bval[:nb.value], fnode[:nf.value], foff[:nf.value + 1], ffeat[:foff[nf.value]]

    def transform_batch_device(self, d_desc, d_n, nframes, cap, levelsup, d_bow_id, d_bow_val, d_n_bow, d_fv_node, d_fv_off, d_fv_feat,
                               d_n_fv, stream=0):
        rc = lib().orbv_transform_batch_device(self.h, d_desc, d_n, nframes, cap, levelsup, d_bow_id, d_bow_val, d_n_bow, d_fv_node,
                                               d_fv_off, d_fv_feat, d_n_fv, stream or None)
        if rc != ORBX_OK:
            raise OrbxError(rc, "orbv_transform_batch_device")

    def score(self, ids1, vals1, ids2, vals2):
        a = np.ascontiguousarray(ids1, dtype=np.uint32); av = np.ascontiguousarray(vals1, dtype=np.float64)
        b = np.ascontiguousarray(ids2, dtype=np.uint32); bv = np.ascontiguousarray(vals2, dtype=np.float64)
        return lib().orbv_score(self.h, a.ctypes.data, av.ctypes.data, len(a), b.ctypes.data, bv.ctypes.data, len(b))


def image_bounds(cam):
    """Frame::ComputeImageBounds + the inverse grid cell sizes (host side, once per camera)"""
    b = Bounds()
    rc = lib().orbf_image_bounds(ctypes.byref(cam), ctypes.byref(b))
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbf_image_bounds")
    return b


def undistort_grid(cam, bounds, kps, device=0):
    """Frame::UndistortKeyPoints + the mGrid fill for one frame: -> (kps_un, cell_off[3073], cell_feat)"""
    kps = np.ascontiguousarray(kps, dtype=KP_DTYPE)
    n = len(kps)
    un = np.zeros(n, dtype=KP_DTYPE)
    off = np.zeros(GRID_CELLS + 1, np.int32)
    feat = np.zeros(max(n, 1), np.int32)
    rc = lib().orbf_undistort_grid(ctypes.byref(cam), ctypes.byref(bounds), kps.ctypes.data if n else None, n, un.ctypes.data if n else None,
                                   off.ctypes.data, feat.ctypes.data, device)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbf_undistort_grid")
    return un, off, feat[:off[GRID_CELLS]]


def undistort_grid_batch_device(cam, bounds, d_kps, d_n, nframes, cap, d_kps_un, d_cell_off, d_cell_feat, stream=0):
    rc = lib().orbf_undistort_grid_batch_device(ctypes.byref(cam), ctypes.byref(bounds), d_kps, d_n, nframes, cap, d_kps_un, d_cell_off, d_cell_feat,
                                                stream or None)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbf_undistort_grid_batch_device")


def features_in_area(bounds, kps_un, cell_off, cell_feat, qxyr, qlev, cand_cap=None, device=0):
    """Frame::GetFeaturesInArea for many (x, y, r, minLevel, maxLevel) queries: -> (seg_off, cand) CSR"""
    kps_un = np.ascontiguousarray(kps_un, dtype=KP_DTYPE)
    cell_off = np.ascontiguousarray(cell_off, dtype=np.int32)
    cell_feat = np.ascontiguousarray(cell_feat, dtype=np.int32)
    qxyr = np.ascontiguousarray(qxyr, dtype=np.float32).reshape(-1, 3)
    qlev = np.ascontiguousarray(qlev, dtype=np.int32).reshape(-1, 2)
    nq = len(qxyr)
    assert len(qlev) == nq and len(cell_off) == GRID_CELLS + 1
    cap = cand_cap if cand_cap is not None else max(1, nq * 64)
    while True:
        seg = np.zeros(nq + 1, np.int32)
        cand = np.zeros(max(cap, 1), np.int32)
        rc = lib().orbf_features_in_area(ctypes.byref(bounds), kps_un.ctypes.data, len(kps_un), cell_off.ctypes.data, cell_feat.ctypes.data,
                                         qxyr.ctypes.data, qlev.ctypes.data, nq, seg.ctypes.data, cand.ctypes.data, cap, device)
        if rc == ORBX_ERR_CAPACITY and cand_cap is None:
            cap = int(seg[nq])
            continue
        if rc != ORBX_OK:
            raise OrbxError(rc, "orbf_features_in_area")
        return seg, cand[:seg[nq]]


def three_maxima(sizes):
    """ORBmatcher::ComputeThreeMaxima on bin sizes (host)"""
    sz = np.ascontiguousarray(sizes, dtype=np.int32)
    out = np.zeros(3, np.int32)
    lib().orbs_three_maxima(sz.ctypes.data, len(sz), out.ctypes.data)
    return tuple(int(v) for v in out)


def window_search_batch_device(bounds, rule, th, ratio, check_orientation, d_kps_un, d_desc, d_cell_off, d_cell_feat, d_nt, cap, d_claimed,
                               d_qxyr, d_qlev, d_qdesc, d_qangle, d_qvalid, d_nq, qcap, nproblems, d_q2t, d_t2q, d_best, d_second, d_nmatches,
                               stream=0):
    """the greedy grid-window searches of ORBmatcher (include/orbs.h), device pointers as ints (0 = NULL)"""
    prm = SearchParams(rule, th, ratio, 1 if check_orientation else 0)
    rc = lib().orbs_window_search_batch_device(ctypes.byref(bounds), ctypes.byref(prm), d_kps_un, d_desc, d_cell_off, d_cell_feat, d_nt, cap,
                                               d_claimed or None, d_qxyr, d_qlev, d_qdesc, d_qangle or None, d_qvalid or None, d_nq, qcap,
                                               nproblems, d_q2t, d_t2q, d_best or None, d_second or None, d_nmatches, stream or None)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbs_window_search_batch_device")


def distinctive(desc, seg_off, device=0):
    """MapPoint::ComputeDistinctiveDescriptors for many map points (CSR of observed descriptors): -> (best_idx, best_median)"""
    desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 32)
    seg = np.ascontiguousarray(seg_off, dtype=np.int32)
    M = len(seg) - 1
    idx = np.empty(max(M, 1), np.int32); med = np.empty(max(M, 1), np.int32)
    rc = lib().orbm_distinctive(desc.ctypes.data if len(desc) else None, seg.ctypes.data, M, idx.ctypes.data, med.ctypes.data, device)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbm_distinctive")
    return idx[:M], med[:M]


def list_search_batch_device(rule, th, ratio, check_orientation, d_kps, d_desc, d_list, d_nlist, d_nt, cap, d_claimed, d_qrange, d_qindex, d_qdesc,
                             d_qangle, d_qvalid, d_nq, qcap, nproblems, d_q2t, d_t2q, d_best, d_second, d_nmatches, stream=0):
    """the in-order search over explicit candidate lists (SearchByBoW with the FeatureVector CSR as the list)"""
    prm = SearchParams(rule, th, ratio, 1 if check_orientation else 0)
    rc = lib().orbs_list_search_batch_device(ctypes.byref(prm), d_kps, d_desc, d_list, d_nlist, d_nt, cap, d_claimed or None, d_qrange, d_qindex or None,
                                             d_qdesc, d_qangle or None, d_qvalid or None, d_nq, qcap, nproblems, d_q2t, d_t2q, d_best or None,
                                             d_second or None, d_nmatches, stream or None)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbs_list_search_batch_device")


def triangulation_search_batch_device(th, check_orientation, d_F12, level_sigma2, d_kps2, d_desc2, d_list, d_nlist, d_nt, cap, d_claimed, d_qrange,
                                      d_qindex, d_kps1, d_qdesc, d_qvalid, d_nq, qcap, nproblems, d_q2t, d_t2q, d_best, d_second, d_nmatches, stream=0):
    """ORBmatcher::SearchForTriangulation over FeatureVector lists (level_sigma2: host float array = mvLevelSigma2 of pKF2)"""
    prm = SearchParams(RULE_TRIANGULATION, th, 0.0, 1 if check_orientation else 0)
    s2 = np.ascontiguousarray(level_sigma2, dtype=np.float32)
    rc = lib().orbs_triangulation_search_batch_device(ctypes.byref(prm), d_F12, s2.ctypes.data, len(s2), d_kps2, d_desc2, d_list, d_nlist, d_nt, cap,
                                                      d_claimed or None, d_qrange, d_qindex or None, d_kps1, d_qdesc, d_qvalid or None, d_nq, qcap,
                                                      nproblems, d_q2t, d_t2q, d_best or None, d_second or None, d_nmatches, stream or None)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbs_triangulation_search_batch_device")


def epipolar_bound(sigma2):
    return float(lib().orbs_epipolar_bound(ctypes.c_float(sigma2)))


def agreement_batch_device(d_match12, d_n1, cap1, d_match21, d_n2, cap2, nproblems, d_out12, d_nfound, stream=0):
    """the "check agreement" tail of ORBmatcher::SearchBySim3"""
    rc = lib().orbs_agreement_batch_device(d_match12, d_n1, cap1, d_match21, d_n2, cap2, nproblems, d_out12, d_nfound, stream or None)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbs_agreement_batch_device")


def stream_create(device=0):
    """a raw non-blocking HIP stream (address) from the C ABI: creation order is under the caller's control"""
    p = ctypes.c_void_p()
    rc = lib().orbx_stream_create(device, ctypes.byref(p))
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbx_stream_create")
    return p.value


def stream_destroy(device, stream):
    lib().orbx_stream_destroy(device, ctypes.c_void_p(stream))


def bow_ranges_batch_device(d_fvq_node, d_fvq_off, d_nfv_q, d_fvt_node, d_fvt_off, d_nfv_t, cap, nproblems, d_qrange, d_nq, stream=0):
    rc = lib().orbs_bow_ranges_batch_device(d_fvq_node, d_fvq_off, d_nfv_q, d_fvt_node, d_fvt_off, d_nfv_t, cap, nproblems, d_qrange, d_nq, stream or None)
    if rc != ORBX_OK:
        raise OrbxError(rc, "orbs_bow_ranges_batch_device")

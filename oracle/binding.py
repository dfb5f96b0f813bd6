"""ctypes binding of the CPU ORACLE (oracle/libvsf_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from the product package.  PARITY UNPINNED: see
oracle/vsf_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
# VSF_ORACLE_LIB: load another build of the same sources instead (tests/test_sanitizers.py: the -fsanitize build)
_LIB_PATH = Path(os.environ["VSF_ORACLE_LIB"]) if os.environ.get("VSF_ORACLE_LIB") else _DIR / "libvsf_oracle.so"

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])
DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
VISION_FEATURE_DTYPE = np.dtype([("feature_idx", "<u8"), ("pixel", "<f4", (2,)), ("point3d", "<f4", (3,))])
assert KEYPOINT_DTYPE.itemsize == 28 and DMATCH_DTYPE.itemsize == 16 and VISION_FEATURE_DTYPE.itemsize == 28


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("edge_threshold", C.c_int32), ("first_level", C.c_int32), ("wta_k", C.c_int32),
                ("score_type", C.c_int32), ("patch_size", C.c_int32), ("fast_threshold", C.c_int32),
                ("blur_sse2", C.c_int32)]


def build(force: bool = False) -> Path:
    if os.environ.get("VSF_ORACLE_LIB"):
        return _LIB_PATH
    src = [_DIR / "vsf_oracle.cc", _DIR / "vsf_oracle_jpeg.cc", _DIR / "vsf_oracle.h", _DIR.parent / "data" / "orb_pattern31.txt"]
    if force or not _LIB_PATH.exists() or any(s.stat().st_mtime > _LIB_PATH.stat().st_mtime for s in src):
        subprocess.check_call(["make", "-s", "-C", str(_DIR), "-B" if force else "-s"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        L = C.CDLL(str(_LIB_PATH))
        vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        L.vsfo_orb_params_default.argtypes = [C.POINTER(OrbParams)]
        L.vsfo_resize_linear_u8.argtypes = [vp, i32, i32, sz, vp, i32, i32, sz]
        L.vsfo_resize_tables.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp]
        L.vsfo_fast9_16.argtypes = [vp, i32, i32, sz, i32, i32, vp, i32]
        L.vsfo_fast_corner_score.argtypes = [vp, sz, i32, i32, i32]
        L.vsfo_gaussian_blur7.argtypes = [vp, i32, i32, sz, vp, sz, i32]
        L.vsfo_gaussian_kernel7_fixed.argtypes = [vp]
        L.vsfo_fast_atan2.argtypes = [f32, f32]
        L.vsfo_fast_atan2.restype = f32
        L.vsfo_orb_pattern31.restype = C.POINTER(C.c_int8)
        L.vsfo_orb_create.argtypes = [C.POINTER(OrbParams)]
        L.vsfo_orb_create.restype = vp
        L.vsfo_orb_destroy.argtypes = [vp]
        L.vsfo_orb_run.argtypes = [vp, vp, i32, i32, sz]
        L.vsfo_orb_nlevels.argtypes = [vp]
        L.vsfo_orb_layout.argtypes = [vp, i32, i32]
        L.vsfo_orb_level_info.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(f32), C.POINTER(i32)]
        L.vsfo_orb_level_image.argtypes = [vp, i32, i32, vp, sz]
        L.vsfo_orb_stage_keypoints.argtypes = [vp, i32, i32, vp, i32]
        L.vsfo_orb_result.argtypes = [vp, vp, vp, i32]
        L.vsfo_retain_best.argtypes = [vp, vp, i32, i32]
        L.vsfo_knn2_hamming.argtypes = [vp, i32, vp, i32, vp, vp]
        L.vsfo_get_matches.argtypes = [vp, i32, vp, i32, C.c_double, vp, i32]
        L.vsfo_get_matches_mt.argtypes = [vp, i32, vp, i32, C.c_double, vp, i32, i32]
        L.vsfo_sort_and_trim.argtypes = [vp, i32, f32]
        L.vsfo_remove_ambig_stereo.argtypes = [vp, vp, vp, i32, vp, C.POINTER(f32), vp, vp]
        L.vsfo_bayer_bg_to_gray.argtypes = [vp, i32, i32, sz, vp, sz]
        L.vsfo_jpeg_decode_gray.argtypes = [vp, sz, vp, sz, i32, i32, C.POINTER(i32), C.POINTER(i32)]
        L.vsfo_triangulate_points.argtypes = [vp, vp, vp, vp, i32, i32, vp]
        L.vsfo_undistort_points.argtypes = [vp, i32, vp, vp, vp]
        L.vsfo_vision_features.argtypes = [vp, vp, vp, vp, i32, C.c_double, vp, vp, vp, vp, i32, vp, C.POINTER(i32)]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _u8img(img: np.ndarray) -> np.ndarray:
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 2
    if img.strides[1] != 1:
        img = np.ascontiguousarray(img)
    return img


def orb_params(nfeatures: int = 10000, fast_threshold: int = 20, nlevels: int = 50, scale_factor: float = 1.04,
               blur_sse2: int = 1) -> OrbParams:
    p = OrbParams()
    lib().vsfo_orb_params_default(C.byref(p))
    p.nfeatures, p.fast_threshold, p.nlevels, p.blur_sse2 = nfeatures, fast_threshold, nlevels, blur_sse2
    p.scale_factor = scale_factor
    return p


def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = _u8img(src)
    dst = np.empty((dh, dw), np.uint8)
    rc = lib().vsfo_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    assert rc == 0
    return dst


def resize_tables(sw, sh, dw, dh):
    xofs, yofs = np.empty(dw, np.int32), np.empty(dh, np.int32)
    ia, ib = np.empty(2 * dw, np.int16), np.empty(2 * dh, np.int16)
    xmax = lib().vsfo_resize_tables(sw, sh, dw, dh, _p(xofs), _p(ia), _p(yofs), _p(ib))
    return xofs, ia, yofs, ib, xmax


def fast9_16(img: np.ndarray, threshold: int, nms: bool = True) -> np.ndarray:
    img = _u8img(img)
    h, w = img.shape
    n = lib().vsfo_fast9_16(_p(img), w, h, img.strides[0], threshold, int(nms), None, 0)
    out = np.zeros(max(n, 1), KEYPOINT_DTYPE)
    lib().vsfo_fast9_16(_p(img), w, h, img.strides[0], threshold, int(nms), _p(out), n)
    return out[:n]


def fast_corner_score(img: np.ndarray, x: int, y: int, threshold: int) -> int:
    img = _u8img(img)
    return lib().vsfo_fast_corner_score(_p(img), img.strides[0], x, y, threshold)


def gaussian_blur7(img: np.ndarray, sse2: bool = True) -> np.ndarray:
    img = _u8img(img)
    h, w = img.shape
    dst = np.empty((h, w), np.uint8)
    assert lib().vsfo_gaussian_blur7(_p(img), w, h, img.strides[0], _p(dst), w, int(sse2)) == 0
    return dst


def gaussian_kernel7_fixed() -> np.ndarray:
    k = np.empty(7, np.int32)
    lib().vsfo_gaussian_kernel7_fixed(_p(k))
    return k


def fast_atan2(y: float, x: float) -> float:
    return float(lib().vsfo_fast_atan2(y, x))


def orb_pattern31() -> np.ndarray:
    return np.ctypeslib.as_array(lib().vsfo_orb_pattern31(), shape=(256, 4)).copy()


def retain_best(keys: np.ndarray, n_points: int):
    """cv::KeyPointsFilter::retainBest on float responses; returns (responses, ids) in libstdc++'s order."""
    r = np.ascontiguousarray(keys, np.float32).copy()
    ids = np.arange(len(r), dtype=np.uint32)
    n = lib().vsfo_retain_best(_p(r), _p(ids), len(r), n_points)
    return r[:n], ids[:n]


class Orb:
    """cv::ORB::detectAndCompute restatement with all intermediates kept."""

    def __init__(self, params: OrbParams | None = None, **kw):
        self.params = params if params is not None else orb_params(**kw)
        self._h = lib().vsfo_orb_create(C.byref(self.params))
        self.n = 0

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().vsfo_orb_destroy(self._h)
            except TypeError:  # interpreter shutdown: the module's globals are gone already
                pass
            self._h = None

    def run(self, img: np.ndarray) -> int:
        img = _u8img(img)
        h, w = img.shape
        self.n = lib().vsfo_orb_run(self._h, _p(img), w, h, img.strides[0])
        if self.n < 0:
            raise RuntimeError("vsfo_orb_run failed: %d" % self.n)
        return self.n

    def layout(self, w: int, h: int):
        lib().vsfo_orb_layout(self._h, w, h)
        return [self.level_info(l) for l in range(self.nlevels)]

    @property
    def nlevels(self) -> int:
        return lib().vsfo_orb_nlevels(self._h)

    def level_info(self, level: int):
        w, h, n, s = C.c_int(), C.c_int(), C.c_int(), C.c_float()
        assert lib().vsfo_orb_level_info(self._h, level, C.byref(w), C.byref(h), C.byref(s), C.byref(n)) == 0
        return w.value, h.value, s.value, n.value

    def level_image(self, level: int, blurred: bool = False) -> np.ndarray:
        w, h, _, _ = self.level_info(level)
        out = np.empty((h, w), np.uint8)
        assert lib().vsfo_orb_level_image(self._h, level, int(blurred), _p(out), w) == 0
        return out

    def stage(self, stage: int, level: int) -> np.ndarray:
        n = lib().vsfo_orb_stage_keypoints(self._h, stage, level, None, 0)
        out = np.zeros(max(n, 1), KEYPOINT_DTYPE)
        lib().vsfo_orb_stage_keypoints(self._h, stage, level, _p(out), n)
        return out[:n]

    def result(self):
        n = self.n
        kps = np.zeros(max(n, 1), KEYPOINT_DTYPE)
        desc = np.zeros((max(n, 1), 32), np.uint8)
        lib().vsfo_orb_result(self._h, _p(kps), _p(desc), n)
        return kps[:n], desc[:n]


def knn2_hamming(q: np.ndarray, t: np.ndarray):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.empty((len(q), 2), np.int32)
    dist = np.empty((len(q), 2), np.int32)
    assert lib().vsfo_knn2_hamming(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist)) == 0
    return idx, dist


def get_matches(q: np.ndarray, t: np.ndarray, ratio: float = float(np.float32(0.6)), threads: int = 1) -> np.ndarray:
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    out = np.zeros(max(len(q), 1), DMATCH_DTYPE)
    n = lib().vsfo_get_matches_mt(_p(q), len(q), _p(t), len(t), ratio, _p(out), len(q), threads)
    return out[:n]


def sort_and_trim(m: np.ndarray, best_percent: float) -> np.ndarray:
    m = np.ascontiguousarray(m, DMATCH_DTYPE).copy()
    n = lib().vsfo_sort_and_trim(_p(m), len(m), best_percent)
    return m[:n]


def set_residual_order(order: int) -> None:
    """0: Eigen 3.3's a0*b0 + (a1*b1 + a2*b2) (default); 1: left to right."""
    lib().vsfo_set_residual_order(int(order))


def remove_ambig_stereo(left: np.ndarray, right: np.ndarray, matches: np.ndarray, F: np.ndarray, threshold: float):
    left = np.ascontiguousarray(left, KEYPOINT_DTYPE)
    right = np.ascontiguousarray(right, KEYPOINT_DTYPE)
    matches = np.ascontiguousarray(matches, DMATCH_DTYPE)
    F = np.ascontiguousarray(F, np.float32).reshape(9)
    thr = C.c_float(threshold)
    keep = np.zeros(max(len(matches), 1), np.uint8)
    res = np.zeros(max(len(matches), 1), np.float32)
    kept = lib().vsfo_remove_ambig_stereo(_p(left), _p(right), _p(matches), len(matches), _p(F), C.byref(thr),
                                          _p(keep), _p(res))
    return keep[:len(matches)].astype(bool), res[:len(matches)], thr.value, kept


def bayer_bg_to_gray(mosaic: np.ndarray) -> np.ndarray:
    """cvtColor(COLOR_BayerBG2BGR) + cvtColor(COLOR_BGR2GRAY) of an 8-bit mosaic (slam_frontend_main.cc:101-106)."""
    m = _u8img(mosaic)
    out = np.zeros_like(m)
    if lib().vsfo_bayer_bg_to_gray(_p(m), m.shape[1], m.shape[0], m.strides[0], _p(out), out.strides[0]) != 0:
        raise ValueError("vsfo_bayer_bg_to_gray")
    return out


def triangulate_points(P1: np.ndarray, P2: np.ndarray, pts1: np.ndarray, pts2: np.ndarray, rows: int = 6) -> np.ndarray:
    """cv::triangulatePoints for float32 inputs; returns (n, 4) float32 homogeneous points."""
    P1 = np.ascontiguousarray(P1, np.float32).reshape(12)
    P2 = np.ascontiguousarray(P2, np.float32).reshape(12)
    a = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2)
    b = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    out = np.zeros((max(len(a), 1), 4), np.float32)
    if lib().vsfo_triangulate_points(_p(P1), _p(P2), _p(a), _p(b), len(a), rows, _p(out)) != 0:
        raise ValueError("vsfo_triangulate_points")
    return out[:len(a)]


def undistort_points(pts: np.ndarray, K: np.ndarray, dist: np.ndarray) -> np.ndarray:
    """cv::undistortPoints(pts, K, dist, noArray(), K); (n, 2) float32."""
    a = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float32).reshape(9)
    d = np.ascontiguousarray(dist, np.float32).reshape(5)
    out = np.zeros((max(len(a), 1), 2), np.float32)
    if lib().vsfo_undistort_points(_p(a), len(a), _p(K), _p(d), _p(out)) != 0:
        raise ValueError("vsfo_undistort_points")
    return out[:len(a)]


def vision_features(left: np.ndarray, left_desc: np.ndarray, right: np.ndarray, right_desc: np.ndarray, P_left, P_right,
                    K_left, dist_left, ratio: float = float(np.float32(0.6)), rows: int = 6):
    """slam_frontend.cc:437-443 on the two filtered frames; returns (VISION_FEATURE_DTYPE[n], n_points)."""
    left = np.ascontiguousarray(left, KEYPOINT_DTYPE)
    right = np.ascontiguousarray(right, KEYPOINT_DTYPE)
    n = len(left)
    assert len(right) == n
    ld = np.ascontiguousarray(left_desc, np.uint8).reshape(-1, 32)
    rd = np.ascontiguousarray(right_desc, np.uint8).reshape(-1, 32)
    out = np.zeros(max(n, 1), VISION_FEATURE_DTYPE)
    npts = C.c_int(0)
    r = lib().vsfo_vision_features(_p(left), _p(ld), _p(right), _p(rd), n, ratio,
                                   _p(np.ascontiguousarray(P_left, np.float32).reshape(12)),
                                   _p(np.ascontiguousarray(P_right, np.float32).reshape(12)),
                                   _p(np.ascontiguousarray(K_left, np.float32).reshape(9)),
                                   _p(np.ascontiguousarray(dist_left, np.float32).reshape(5)), rows, _p(out),
                                   C.byref(npts))
    if r < 0:
        raise ValueError("vsfo_vision_features")
    return out[:n], npts.value


def jpeg_decode_gray(data: bytes) -> np.ndarray:
    """cv::imdecode(data, IMREAD_GRAYSCALE) for a baseline JPEG; raises ValueError (malformed) or NotImplementedError
    (a JPEG process the oracle does not restate: progressive, arithmetic, multi-scan)."""
    buf = np.frombuffer(bytes(data), np.uint8)
    w, h = C.c_int(0), C.c_int(0)
    r = lib().vsfo_jpeg_decode_gray(_p(buf), len(buf), None, 0, 0, 0, C.byref(w), C.byref(h))
    if r == -2:
        raise NotImplementedError("JPEG process not restated")
    if r != -3:
        raise ValueError("malformed JPEG (%d)" % r)
    out = np.zeros((h.value, w.value), np.uint8)
    r = lib().vsfo_jpeg_decode_gray(_p(buf), len(buf), _p(out), out.strides[0], w.value, h.value, C.byref(w), C.byref(h))
    if r == -2:
        raise NotImplementedError("JPEG process not restated")
    if r != 0:
        raise ValueError("malformed JPEG (%d)" % r)
    return out

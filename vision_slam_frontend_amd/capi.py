"""ctypes binding of the C ABI in include/vsf.h (libvsf_hip.so, gfx950 HIP kernels).

There is NO CPU fallback: if the shared library is missing or a GPU call fails this module raises.
Host-side conveniences only (numpy in / numpy out for the host-pointer entry points, raw device pointers
for the batched entry points, which callers fill from torch tensors).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libvsf_hip.so"

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])
DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
VISION_FEATURE_DTYPE = np.dtype([("feature_idx", "<u8"), ("pixel", "<f4", (2,)), ("point3d", "<f4", (3,))])
FEATURE_MATCH_DTYPE = np.dtype([("feature_idx_initial", "<u8"), ("feature_idx_current", "<u8")])
assert VISION_FEATURE_DTYPE.itemsize == 28 and FEATURE_MATCH_DTYPE.itemsize == 16
DESC_BYTES = 32
PAYLOAD_MAGIC = 0x31465356  # "VSF1"

VSF_OK, VSF_ERR_INVALID_ARG, VSF_ERR_CAPACITY, VSF_ERR_HIP, VSF_ERR_UNSUPPORTED, VSF_ERR_NO_DEVICE = range(6)

# Every symbol include/vsf.h declares (tests check that the library exports all of them).
EXPORTS = [
    "vsf_params_default", "vsf_params_set_ratio", "vsf_create", "vsf_destroy", "vsf_status_string",
    "vsf_last_hip_error", "vsf_get_params", "vsf_set_stream", "vsf_sync", "vsf_level_info", "vsf_extract",
    "vsf_fast_detect", "vsf_knn2_hamming", "vsf_get_matches", "vsf_extract_pair", "vsf_get_matches_multi", "vsf_extract_batch_dev", "vsf_match_batch_dev",
    "vsf_stereo_batch_dev", "vsf_set_lanes", "vsf_set_pipeline", "vsf_set_blur_overlap", "vsf_set_fast_resident", "vsf_get_fast_resident", "vsf_remove_ambig_stereo_batch_dev", "vsf_feature_matches_batch_dev", "vsf_bayer_bg_to_gray_batch_dev", "vsf_debug_level_image", "vsf_debug_fast_candidates", "vsf_debug_level_keypoints",
    "vsf_algorithmic_bytes_per_image", "vsf_pyramid_pixels", "vsf_profile_enable", "vsf_profile_read",
    "vsf_stage_name", "vsf_debug_retain_best", "vsf_debug_sort_trim", "vsf_stereo_residuals_batch_dev", "vsf_stereo_thresholds_dev",
    "vsf_stereo_filter_batch_dev", "vsf_vision_features_batch_dev", "vsf_packed_outputs_capacity",
    "vsf_pack_outputs_dev", "vsf_observe_capacity", "vsf_observe_stereo", "vsf_observe_submit", "vsf_observe_collect", "vsf_observe_reset",
    "vsf_observe_configure", "vsf_observe_collect_view", "vsf_observe_stats", "vsf_observe_poll", "vsf_debug_jpeg_serial",
    "vsf_jpeg_decode_gray_batch", "vsf_png_decode_gray_batch", "vsf_imdecode_gray_batch", "vsf_tune_fast_resident", "vsf_set_option", "vsf_get_option", "vsf_debug_inject_hip_error", "vsf_comm_unique_id", "vsf_comm_create", "vsf_comm_destroy", "vsf_comm_info",
    "vsf_allgather_dev", "vsf_gather_payload_dev", "vsf_reserve", "vsf_set_input_event",
]
# vsf_option (include/vsf.h)
(OPT_FAST_BOTH_MAX, OPT_SELECT_WIDE, OPT_PYRAMID_FEW, OPT_PYRAMID_CHAIN, OPT_PYRAMID_ROWS, OPT_SELECT_BIG_CLASS,
 OPT_PIPE_AFTER_FAST, OPT_PIPE_PRIORITY, OPT_OBSERVE_THREAD, OPT_PYRAMID_TAIL_MIN, OPT_OBSERVE_COPY_THREAD) = range(11)
STAGE_COUNT = 8


class VsfParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("edge_threshold", C.c_int32), ("first_level", C.c_int32), ("wta_k", C.c_int32),
                ("score_type", C.c_int32), ("patch_size", C.c_int32), ("fast_threshold", C.c_int32),
                ("blur_sse2", C.c_int32), ("fast_detector_threshold", C.c_int32), ("fast_detector_nms", C.c_int32),
                ("ratio_num", C.c_uint32), ("ratio_shift", C.c_uint32), ("width", C.c_int32), ("height", C.c_int32),
                ("max_images", C.c_int32), ("max_keypoints", C.c_int32), ("residual_order", C.c_int32)]


class VsfCalibration(C.Structure):
    """vsf_calibration: FrontendConfig's stereo calibration (slam_frontend.cc:565-644), row-major floats."""
    _fields_ = [("projection_left", C.c_float * 12), ("projection_right", C.c_float * 12),
                ("camera_matrix_left", C.c_float * 9), ("distortion_left", C.c_float * 5),
                ("fundamental", C.c_float * 9), ("triangulate_rows", C.c_int32)]

    def set(self, name: str, values) -> "VsfCalibration":
        arr = getattr(self, name)
        v = np.ascontiguousarray(values, np.float32).reshape(-1)
        assert len(v) == len(arr), name
        for i, x in enumerate(v):
            arr[i] = float(x)
        return self

    def get(self, name: str) -> np.ndarray:
        return np.array(list(getattr(self, name)), np.float32)


class VsfError(RuntimeError):
    def __init__(self, status: int, where: str, hip: int = 0):
        self.status = status
        msg = "%s: %s" % (where, lib().vsf_status_string(status).decode())
        if status == VSF_ERR_HIP:
            msg += " (hipError %d)" % hip
        super().__init__(msg)


_lib = None


def lib() -> C.CDLL:
    """Loads libvsf_hip.so; raises (loudly) if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). The product has no CPU fallback." % LIB_PATH)
        L = C.CDLL(str(LIB_PATH))
        vp, i32, sz, pp = C.c_void_p, C.c_int, C.c_size_t, C.POINTER(VsfParams)
        ip = C.POINTER(C.c_int)
        L.vsf_params_default.argtypes = [pp, i32, i32, i32]
        L.vsf_params_set_ratio.argtypes = [pp, C.c_float]
        L.vsf_create.argtypes = [pp, i32, C.POINTER(vp)]
        L.vsf_destroy.argtypes = [vp]
        L.vsf_destroy.restype = None
        L.vsf_status_string.argtypes = [i32]
        L.vsf_status_string.restype = C.c_char_p
        L.vsf_last_hip_error.argtypes = [vp]
        L.vsf_get_params.argtypes = [vp, pp]
        L.vsf_set_stream.argtypes = [vp, vp]
        L.vsf_sync.argtypes = [vp]
        L.vsf_level_info.argtypes = [vp, i32, ip, ip, C.POINTER(C.c_float), ip]
        L.vsf_extract.argtypes = [vp, vp, i32, i32, sz, vp, vp, i32, ip]
        L.vsf_fast_detect.argtypes = [vp, vp, i32, i32, sz, i32, i32, vp, i32, ip]
        L.vsf_extract_pair.argtypes = [vp, vp, vp, i32, i32, sz, vp, vp, ip, vp, vp, ip, i32]
        L.vsf_get_matches_multi.argtypes = [vp, vp, vp, i32, vp, i32, vp, i32, vp]
        L.vsf_knn2_hamming.argtypes = [vp, vp, i32, vp, i32, vp, vp]
        L.vsf_get_matches.argtypes = [vp, vp, i32, vp, i32, vp, i32, ip]
        L.vsf_extract_batch_dev.argtypes = [vp, vp, i32, sz, sz, vp, vp, vp]
        L.vsf_match_batch_dev.argtypes = [vp, vp, vp, sz, vp, vp, i32, vp, vp, vp, vp]
        L.vsf_stereo_batch_dev.argtypes = [vp, vp, i32, sz, sz, vp, vp, vp, vp, vp]
        L.vsf_set_lanes.argtypes = [vp, i32]
        L.vsf_set_pipeline.argtypes = [vp, i32]
        L.vsf_set_input_event.argtypes = [vp, vp]
        L.vsf_reserve.argtypes = [vp, i32, i32]
        L.vsf_set_blur_overlap.argtypes = [vp, i32]
        L.vsf_set_fast_resident.argtypes = [vp, i32]
        L.vsf_get_fast_resident.argtypes = [vp, C.POINTER(C.c_int)]
        L.vsf_tune_fast_resident.argtypes = [vp, vp, i32, sz, sz, vp, vp, vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.vsf_set_option.argtypes = [vp, i32, i32]
        L.vsf_get_option.argtypes = [vp, i32, ip]
        L.vsf_debug_inject_hip_error.argtypes = [vp, i32]
        L.vsf_comm_unique_id.argtypes = [vp]
        L.vsf_comm_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
        L.vsf_comm_destroy.argtypes = [vp]
        L.vsf_comm_destroy.restype = None
        L.vsf_comm_info.argtypes = [vp, ip, ip, ip]
        L.vsf_allgather_dev.argtypes = [vp, vp, vp, vp, sz]
        L.vsf_gather_payload_dev.argtypes = [vp, vp, vp, sz, vp, sz, i32]
        L.vsf_remove_ambig_stereo_batch_dev.argtypes = [vp, vp, vp, vp, vp, i32, vp, C.c_float, vp, vp, vp, vp, vp, vp]
        L.vsf_feature_matches_batch_dev.argtypes = [vp, vp, vp, sz, vp, vp, i32, C.c_float, vp, vp]
        L.vsf_bayer_bg_to_gray_batch_dev.argtypes = [vp, vp, i32, i32, i32, sz, sz, vp, sz, sz]
        L.vsf_stereo_residuals_batch_dev.argtypes = [vp, vp, vp, vp, i32, vp, vp]
        L.vsf_stereo_thresholds_dev.argtypes = [vp, vp, i32, vp, vp]
        L.vsf_stereo_filter_batch_dev.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]
        L.vsf_vision_features_batch_dev.argtypes = [vp, C.POINTER(VsfCalibration), vp, vp, vp, i32, vp, vp, vp]
        L.vsf_packed_outputs_capacity.argtypes = [vp, i32, i32]
        L.vsf_packed_outputs_capacity.restype = sz
        L.vsf_pack_outputs_dev.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, sz]
        L.vsf_debug_sort_trim.argtypes = [vp, vp, i32, i32, C.c_float, i32, vp, vp]
        L.vsf_observe_capacity.argtypes = [vp, i32]
        L.vsf_observe_capacity.restype = sz
        L.vsf_observe_submit.argtypes = [vp, vp, vp, i32, i32, sz, C.POINTER(VsfCalibration), C.c_float, i32,
                                         C.POINTER(C.c_int64)]
        L.vsf_observe_collect.argtypes = [vp, C.c_int64, vp, sz, C.POINTER(sz)]
        L.vsf_observe_collect_view.argtypes = [vp, C.c_int64, C.POINTER(vp), C.POINTER(sz)]
        L.vsf_observe_configure.argtypes = [vp, i32, i32, i32]
        L.vsf_observe_stats.argtypes = [vp, vp, i32]
        L.vsf_observe_poll.argtypes = [vp, C.c_int64, C.POINTER(i32)]
        L.vsf_observe_stereo.argtypes = [vp, vp, vp, i32, i32, sz, C.POINTER(VsfCalibration), C.c_float, i32, vp, sz,
                                         C.POINTER(sz)]
        L.vsf_observe_reset.argtypes = [vp]
        L.vsf_jpeg_decode_gray_batch.argtypes = [vp, vp, vp, i32, i32, i32, vp, sz, sz]
        L.vsf_png_decode_gray_batch.argtypes = [vp, vp, vp, i32, i32, i32, vp, sz, sz]
        L.vsf_imdecode_gray_batch.argtypes = [vp, vp, vp, i32, i32, i32, vp, sz, sz]
        L.vsf_debug_level_image.argtypes = [vp, i32, i32, i32, vp, sz]
        L.vsf_debug_fast_candidates.argtypes = [vp, i32, i32, vp, i32, ip]
        L.vsf_debug_level_keypoints.argtypes = [vp, i32, i32, vp, i32, ip]
        L.vsf_algorithmic_bytes_per_image.argtypes = [vp]
        L.vsf_algorithmic_bytes_per_image.restype = C.c_uint64
        L.vsf_pyramid_pixels.argtypes = [vp]
        L.vsf_pyramid_pixels.restype = C.c_uint64
        L.vsf_debug_retain_best.argtypes = [vp, vp, vp, i32, i32, i32, i32, ip]
        L.vsf_profile_enable.argtypes = [vp, i32]
        L.vsf_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), i32]
        L.vsf_stage_name.argtypes = [i32]
        L.vsf_stage_name.restype = C.c_char_p
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if isinstance(a, np.ndarray) else C.c_void_p(int(a) if a else 0)


def default_params(width: int, height: int, max_images: int = 2, nfeatures: int = 10000, **over) -> VsfParams:
    p = VsfParams()
    st = lib().vsf_params_default(C.byref(p), width, height, max_images)
    if st != VSF_OK:
        raise VsfError(st, "vsf_params_default")
    p.nfeatures = nfeatures
    ratio = over.pop("nn_match_ratio", None)
    if ratio is not None:
        st = lib().vsf_params_set_ratio(C.byref(p), ratio)
        if st != VSF_OK:
            raise VsfError(st, "vsf_params_set_ratio")
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class Context:
    """One GPU's vsf_ctx.  Not thread-safe (same contract as the C ABI)."""

    def __init__(self, params: VsfParams, device: int = 0):
        self._h = C.c_void_p()
        st = lib().vsf_create(C.byref(params), device, C.byref(self._h))
        if st != VSF_OK:
            self._h = None
            raise VsfError(st, "vsf_create")
        self.params = VsfParams()
        lib().vsf_get_params(self._h, C.byref(self.params))
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib().vsf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except TypeError:  # interpreter shutdown: the module's globals are gone already (the process ends anyway)
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, st: int, where: str, allow_capacity: bool = False):
        if st == VSF_OK or (allow_capacity and st == VSF_ERR_CAPACITY):
            return st
        raise VsfError(st, where, lib().vsf_last_hip_error(self._h))

    # ---- geometry ----
    @property
    def nlevels(self) -> int:
        return self.params.nlevels

    def level_info(self, level: int):
        w, h, n, s = C.c_int(), C.c_int(), C.c_int(), C.c_float()
        self._check(lib().vsf_level_info(self._h, level, C.byref(w), C.byref(h), C.byref(s), C.byref(n)),
                    "vsf_level_info")
        return w.value, h.value, s.value, n.value

    def algorithmic_bytes_per_image(self) -> int:
        return int(lib().vsf_algorithmic_bytes_per_image(self._h))

    def pyramid_pixels(self) -> int:
        return int(lib().vsf_pyramid_pixels(self._h))

    def set_stream(self, hip_stream: int | None):
        self._check(lib().vsf_set_stream(self._h, C.c_void_p(hip_stream or 0)), "vsf_set_stream")

    def set_lanes(self, lanes: int):
        self._check(lib().vsf_set_lanes(self._h, lanes), "vsf_set_lanes")

    def set_blur_overlap(self, on: bool):
        self._check(lib().vsf_set_blur_overlap(self._h, int(on)), "vsf_set_blur_overlap")

    def set_fast_resident(self, waves: int):
        """-1 (default): what tune_fast_resident measured for the batch size, the grid form otherwise; 0: FAST as one
        workgroup per four cells; 2..4: resident, that many waves per SIMD."""
        self._check(lib().vsf_set_fast_resident(self._h, int(waves)), "vsf_set_fast_resident")

    def tune_fast_resident(self, d_imgs: int, n_images: int, image_stride: int, row_stride: int, d_kp: int, d_desc: int,
                           d_counts: int, samples: int = 3):
        """vsf_tune_fast_resident: BLOCKING; returns (median ms of the grid form, median ms of the resident form)."""
        g, r = C.c_float(), C.c_float()
        self._check(lib().vsf_tune_fast_resident(self._h, C.c_void_p(d_imgs), n_images, image_stride, row_stride,
                                                 C.c_void_p(d_kp), C.c_void_p(d_desc), C.c_void_p(d_counts), samples,
                                                 C.byref(g), C.byref(r)), "vsf_tune_fast_resident")
        return g.value, r.value

    def set_option(self, option: int, value: int):
        self._check(lib().vsf_set_option(self._h, int(option), int(value)), "vsf_set_option")

    def get_option(self, option: int) -> int:
        v = C.c_int()
        self._check(lib().vsf_get_option(self._h, int(option), C.byref(v)), "vsf_get_option")
        return v.value

    def get_fast_resident(self) -> int:
        w = C.c_int(-1)
        self._check(lib().vsf_get_fast_resident(self._h, C.byref(w)), "vsf_get_fast_resident")
        return w.value

    def set_pipeline(self, on: bool):
        self._check(lib().vsf_set_pipeline(self._h, int(on)), "vsf_set_pipeline")

    def set_input_event(self, hip_event):
        """The next batched call (its pipelined pyramid included) waits on the GPU for this hipEvent_t (an int handle,
        e.g. torch.cuda.Event.cuda_event after record()); one-shot; None withdraws it."""
        self._check(lib().vsf_set_input_event(self._h, C.c_void_p(int(hip_event) if hip_event else None)), "vsf_set_input_event")

    def reserve(self, n_frames: int, n_pairs: int):
        """BLOCKING set-up: scratch of the batched *_dev calls for up to n_frames stereo frames / n_pairs temporal pairs."""
        self._check(lib().vsf_reserve(self._h, int(n_frames), int(n_pairs)), "vsf_reserve")

    def sync(self, allow_capacity: bool = False) -> int:
        return self._check(lib().vsf_sync(self._h), "vsf_sync", allow_capacity)

    # ---- host-pointer API (one call == one reference call) ----
    def extract(self, img: np.ndarray, cap: int | None = None):
        """detectAndCompute: returns (keypoints[KEYPOINT_DTYPE], descriptors[n,32] uint8)."""
        img = _u8(img)
        cap = self.params.max_keypoints if cap is None else cap
        kp = np.zeros(max(cap, 1), KEYPOINT_DTYPE)
        desc = np.zeros((max(cap, 1), DESC_BYTES), np.uint8)
        n = C.c_int()
        self._check(lib().vsf_extract(self._h, _p(img), img.shape[1], img.shape[0], img.strides[0], _p(kp), _p(desc),
                                      cap, C.byref(n)), "vsf_extract")
        return kp[:n.value], desc[:n.value]

    def extract_pair(self, img0: np.ndarray, img1: np.ndarray, cap: int | None = None):
        """Both detectAndCompute calls of a stereo frame in one batch: ((kp0, desc0), (kp1, desc1))."""
        img0, img1 = _u8(img0), _u8(img1)
        assert img0.shape == img1.shape and img0.strides == img1.strides
        cap = self.params.max_keypoints if cap is None else cap
        kp = [np.zeros(max(cap, 1), KEYPOINT_DTYPE) for _ in range(2)]
        desc = [np.zeros((max(cap, 1), DESC_BYTES), np.uint8) for _ in range(2)]
        n0, n1 = C.c_int(), C.c_int()
        self._check(lib().vsf_extract_pair(self._h, _p(img0), _p(img1), img0.shape[1], img0.shape[0], img0.strides[0],
                                           _p(kp[0]), _p(desc[0]), C.byref(n0), _p(kp[1]), _p(desc[1]), C.byref(n1), cap),
                    "vsf_extract_pair")
        return (kp[0][:n0.value], desc[0][:n0.value]), (kp[1][:n1.value], desc[1][:n1.value])

    def get_matches_multi(self, q_sets, t: np.ndarray):
        """GetMatches of every query set against one train set in one call: list of DMATCH arrays."""
        q_sets = [_desc(q) for q in q_sets]
        t = _desc(t)
        S = len(q_sets)
        cap = max(max((len(q) for q in q_sets), default=0), 1)
        ptrs = (C.c_void_p * S)(*[q.ctypes.data if len(q) else None for q in q_sets])
        nq = np.array([len(q) for q in q_sets], np.int32)
        out = np.zeros((S, cap), DMATCH_DTYPE)
        n_out = np.zeros(S, np.int32)
        self._check(lib().vsf_get_matches_multi(self._h, C.cast(ptrs, C.c_void_p), _p(nq), S, _p(t), len(t), _p(out), cap,
                                                _p(n_out)), "vsf_get_matches_multi")
        return [out[s, :n_out[s]].copy() for s in range(S)]

    def fast_detect(self, img: np.ndarray, threshold: int = -1, nms: bool = True, cap: int = 1 << 16):
        img = _u8(img)
        kp = np.zeros(max(cap, 1), KEYPOINT_DTYPE)
        n = C.c_int()
        self._check(lib().vsf_fast_detect(self._h, _p(img), img.shape[1], img.shape[0], img.strides[0], threshold,
                                          int(nms), _p(kp), cap, C.byref(n)), "vsf_fast_detect")
        return kp[:n.value]

    def knn2_hamming(self, q: np.ndarray, t: np.ndarray):
        q, t = _desc(q), _desc(t)
        idx = np.full((max(len(q), 1), 2), -1, np.int32)
        dist = np.full((max(len(q), 1), 2), np.iinfo(np.int32).max, np.int32)
        self._check(lib().vsf_knn2_hamming(self._h, _p(q), len(q), _p(t), len(t), _p(idx), _p(dist)),
                    "vsf_knn2_hamming")
        return idx[:len(q)], dist[:len(q)]

    def get_matches(self, q: np.ndarray, t: np.ndarray) -> np.ndarray:
        q, t = _desc(q), _desc(t)
        out = np.zeros(max(len(q), 1), DMATCH_DTYPE)
        n = C.c_int()
        self._check(lib().vsf_get_matches(self._h, _p(q), len(q), _p(t), len(t), _p(out), len(q), C.byref(n)),
                    "vsf_get_matches")
        return out[:n.value]

    # ---- device-pointer API (raw addresses, e.g. tensor.data_ptr()) ----
    def extract_batch_dev(self, d_imgs: int, n_images: int, image_stride: int, row_stride: int, d_kp: int,
                          d_desc: int, d_counts: int):
        self._check(lib().vsf_extract_batch_dev(self._h, _p(d_imgs), n_images, image_stride, row_stride, _p(d_kp),
                                                _p(d_desc), _p(d_counts)), "vsf_extract_batch_dev")

    def match_batch_dev(self, d_desc: int, d_counts: int, set_stride: int, d_q_set: int, d_t_set: int, n_pairs: int,
                        d_idx2: int, d_dist2: int, d_matches: int, d_nmatches: int):
        self._check(lib().vsf_match_batch_dev(self._h, _p(d_desc), _p(d_counts), set_stride, _p(d_q_set),
                                              _p(d_t_set), n_pairs, _p(d_idx2), _p(d_dist2), _p(d_matches),
                                              _p(d_nmatches)), "vsf_match_batch_dev")

    def stereo_batch_dev(self, d_imgs: int, n_frames: int, image_stride: int, row_stride: int, d_kp: int,
                         d_desc: int, d_counts: int, d_matches: int, d_nmatches: int):
        self._check(lib().vsf_stereo_batch_dev(self._h, _p(d_imgs), n_frames, image_stride, row_stride, _p(d_kp),
                                               _p(d_desc), _p(d_counts), _p(d_matches), _p(d_nmatches)),
                    "vsf_stereo_batch_dev")

    def remove_ambig_stereo_batch_dev(self, d_kp: int, d_desc: int, d_matches: int, d_nmatches: int, n_frames: int,
                                      F: np.ndarray, thr_in: float, d_thr_override: int, d_means: int, d_thr: int,
                                      d_kp_out: int, d_desc_out: int, d_counts_out: int):
        Fh = np.ascontiguousarray(F, np.float32).reshape(9)
        self._check(lib().vsf_remove_ambig_stereo_batch_dev(self._h, _p(d_kp), _p(d_desc), _p(d_matches), _p(d_nmatches),
                                                            n_frames, _p(Fh), thr_in, _p(d_thr_override), _p(d_means),
                                                            _p(d_thr), _p(d_kp_out), _p(d_desc_out), _p(d_counts_out)),
                    "vsf_remove_ambig_stereo_batch_dev")

    def feature_matches_batch_dev(self, d_desc: int, d_counts: int, set_stride: int, d_q_set: int, d_t_set: int,
                                  n_pairs: int, best_percent: float, d_pairs: int, d_npairs: int):
        self._check(lib().vsf_feature_matches_batch_dev(self._h, _p(d_desc), _p(d_counts), set_stride, _p(d_q_set),
                                                        _p(d_t_set), n_pairs, best_percent, _p(d_pairs), _p(d_npairs)),
                    "vsf_feature_matches_batch_dev")

    def stereo_residuals_batch_dev(self, d_kp: int, d_matches: int, d_nmatches: int, n_frames: int, F: np.ndarray,
                                   d_means: int):
        Fh = np.ascontiguousarray(F, np.float32).reshape(9)
        self._check(lib().vsf_stereo_residuals_batch_dev(self._h, _p(d_kp), _p(d_matches), _p(d_nmatches), n_frames,
                                                         _p(Fh), _p(d_means)), "vsf_stereo_residuals_batch_dev")

    def stereo_thresholds_dev(self, d_means: int, n: int, d_thr_state: int, d_thr: int):
        self._check(lib().vsf_stereo_thresholds_dev(self._h, _p(d_means), n, _p(d_thr_state), _p(d_thr)),
                    "vsf_stereo_thresholds_dev")

    def stereo_filter_batch_dev(self, d_kp: int, d_desc: int, d_matches: int, d_nmatches: int, n_frames: int,
                                d_thr: int, d_kp_out: int, d_desc_out: int, d_counts_out: int):
        self._check(lib().vsf_stereo_filter_batch_dev(self._h, _p(d_kp), _p(d_desc), _p(d_matches), _p(d_nmatches),
                                                      n_frames, _p(d_thr), _p(d_kp_out), _p(d_desc_out),
                                                      _p(d_counts_out)), "vsf_stereo_filter_batch_dev")

    def vision_features_batch_dev(self, calib: VsfCalibration, d_kp: int, d_desc: int, d_counts: int, n_frames: int,
                                  d_features: int, d_nfeatures: int, d_npoints: int = 0):
        self._check(lib().vsf_vision_features_batch_dev(self._h, C.byref(calib), _p(d_kp), _p(d_desc), _p(d_counts),
                                                        n_frames, _p(d_features), _p(d_nfeatures), _p(d_npoints)),
                    "vsf_vision_features_batch_dev")

    def packed_outputs_capacity(self, n_frames: int, n_pairs: int) -> int:
        return int(lib().vsf_packed_outputs_capacity(self._h, n_frames, n_pairs))

    def pack_outputs_dev(self, d_features: int, d_nfeatures: int, n_frames: int, d_pairs: int, d_npairs: int,
                         n_pairs: int, d_payload: int, payload_cap: int):
        self._check(lib().vsf_pack_outputs_dev(self._h, _p(d_features), _p(d_nfeatures), n_frames, _p(d_pairs),
                                               _p(d_npairs), n_pairs, _p(d_payload), payload_cap),
                    "vsf_pack_outputs_dev")

    def observe_stereo(self, left: np.ndarray, right: np.ndarray, calib: VsfCalibration, best_percent: float = 0.3,
                       frame_life: int = 10) -> dict:
        """One Frontend::ObserveImage worth of GPU work in one submission (vsf_observe_stereo); returns the decoded
        result: header fields, `features` (VISION_FEATURE_DTYPE), `factors` (FEATURE_MATCH_DTYPE arrays, oldest kept
        frame first), `stereo_pairs` (the right->left matches of Calculate3DPoints), `keypoints`, `descriptors`."""
        left, right = _u8(left), _u8(right)
        assert left.shape == right.shape and left.strides == right.strides
        cap = int(lib().vsf_observe_capacity(self._h, frame_life))
        buf = np.zeros(max(cap, 64), np.uint8)
        n = C.c_size_t()
        self._check(lib().vsf_observe_stereo(self._h, _p(left), _p(right), left.shape[1], left.shape[0], left.strides[0],
                                             C.byref(calib), float(np.float32(best_percent)), frame_life, _p(buf), cap,
                                             C.byref(n)), "vsf_observe_stereo")
        return decode_observation(buf[:n.value])

    def observe_submit(self, left: np.ndarray, right: np.ndarray, calib: VsfCalibration, best_percent: float = 0.3,
                       frame_life: int = 10) -> int:
        """Queues one ObserveImage (vsf_observe_submit) and returns its ticket without waiting for the GPU."""
        left, right = _u8(left), _u8(right)
        assert left.shape == right.shape and left.strides == right.strides
        t = C.c_int64(-1)
        self._check(lib().vsf_observe_submit(self._h, _p(left), _p(right), left.shape[1], left.shape[0], left.strides[0],
                                             C.byref(calib), float(np.float32(best_percent)), frame_life, C.byref(t)),
                    "vsf_observe_submit")
        return int(t.value)

    def observe_collect(self, ticket: int, frame_life: int = 10) -> dict:
        """Waits for the frame of `ticket` (vsf_observe_collect) and returns its decoded result."""
        cap = int(lib().vsf_observe_capacity(self._h, frame_life))
        buf = np.zeros(max(cap, 64), np.uint8)
        n = C.c_size_t()
        self._check(lib().vsf_observe_collect(self._h, C.c_int64(ticket), _p(buf), cap, C.byref(n)), "vsf_observe_collect")
        return decode_observation(buf[:n.value])

    def debug_jpeg_serial(self, on: bool):
        """Test hook: every JPEG file through the one-wave-per-image decoder (vsf_debug_jpeg_serial)."""
        lib().vsf_debug_jpeg_serial.argtypes = [C.c_void_p, C.c_int]
        self._check(lib().vsf_debug_jpeg_serial(self._h, int(on)), "vsf_debug_jpeg_serial")

    def observe_reset(self):
        self._check(lib().vsf_observe_reset(self._h), "vsf_observe_reset")

    def observe_poll(self, ticket: int) -> bool:
        """True when the frame's result is there (vsf_observe_poll: does not wait, sends nothing)."""
        r = C.c_int(0)
        self._check(lib().vsf_observe_poll(self._h, C.c_int64(ticket), C.byref(r)), "vsf_observe_poll")
        return bool(r.value)

    def observe_configure(self, depth: int = 0, min_batch: int = 0, in_flight: int = 0):
        """The ObserveImage queue (vsf_observe_configure): frames that may wait uncollected, the fewest waiting frames that
        leave while the GPU is busy, batches on the GPU at a time (0 = the defaults)."""
        self._check(lib().vsf_observe_configure(self._h, depth, min_batch, in_flight), "vsf_observe_configure")

    def jpeg_decode_gray_batch(self, files, width: int, height: int, d_dst: int, dst_image_stride: int,
                               dst_row_stride: int, allow_status=()):
        """cv::imdecode(IMREAD_GRAYSCALE) of baseline-JPEG files (bytes objects, host) into device memory
        (slam_frontend_main.cc:99-100); asynchronous on the context's stream.  Returns the status (VSF_OK unless listed
        in `allow_status`)."""
        n = len(files)
        bufs = [np.frombuffer(bytes(f), np.uint8) for f in files]
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        sizes = (C.c_size_t * n)(*[len(b) for b in bufs])
        st = lib().vsf_jpeg_decode_gray_batch(self._h, C.cast(ptrs, C.c_void_p), C.cast(sizes, C.c_void_p), n, width,
                                              height, _p(d_dst), dst_image_stride, dst_row_stride)
        if st != VSF_OK and st not in allow_status:
            raise VsfError(st, "vsf_jpeg_decode_gray_batch", lib().vsf_last_hip_error(self._h))
        return st

    def png_decode_gray_batch(self, files, width: int, height: int, d_dst: int, dst_image_stride: int,
                              dst_row_stride: int, allow_status=()):
        """cv::imdecode(IMREAD_GRAYSCALE) of grayscale PNG files (bytes objects, host) into device memory
        (slam_frontend_main.cc:99-100); asynchronous on the context's stream."""
        n = len(files)
        bufs = [np.frombuffer(bytes(f), np.uint8) for f in files]
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        sizes = (C.c_size_t * n)(*[len(b) for b in bufs])
        st = lib().vsf_png_decode_gray_batch(self._h, C.cast(ptrs, C.c_void_p), C.cast(sizes, C.c_void_p), n, width,
                                             height, _p(d_dst), dst_image_stride, dst_row_stride)
        if st != VSF_OK and st not in allow_status:
            raise VsfError(st, "vsf_png_decode_gray_batch", lib().vsf_last_hip_error(self._h))
        return st

    def imdecode_gray_batch(self, files, width: int, height: int, d_dst: int, dst_image_stride: int,
                            dst_row_stride: int, allow_status=()):
        """cv::imdecode(IMREAD_GRAYSCALE) of JPEG and PNG payloads, mixed (slam_frontend_main.cc:99-100)."""
        n = len(files)
        bufs = [np.frombuffer(bytes(f), np.uint8) for f in files]
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        sizes = (C.c_size_t * n)(*[len(b) for b in bufs])
        st = lib().vsf_imdecode_gray_batch(self._h, C.cast(ptrs, C.c_void_p), C.cast(sizes, C.c_void_p), n, width, height,
                                           _p(d_dst), dst_image_stride, dst_row_stride)
        if st != VSF_OK and st not in allow_status:
            raise VsfError(st, "vsf_imdecode_gray_batch", lib().vsf_last_hip_error(self._h))
        return st

    def bayer_bg_to_gray_batch_dev(self, d_src: int, n_images: int, width: int, height: int, src_image_stride: int,
                                   src_row_stride: int, d_dst: int, dst_image_stride: int, dst_row_stride: int):
        """DecodeImage's BayerBG2BGR + BGR2GRAY (slam_frontend_main.cc:101-106) on mosaics resident in HBM."""
        self._check(lib().vsf_bayer_bg_to_gray_batch_dev(self._h, _p(d_src), n_images, width, height, src_image_stride,
                                                         src_row_stride, _p(d_dst), dst_image_stride, dst_row_stride),
                    "vsf_bayer_bg_to_gray_batch_dev")

    def debug_retain_best(self, keys: np.ndarray, n_points: int, use_lds: bool = False, mode: int = 0):
        """retainBest on the GPU; returns (keys, ids) of the survivors in the order the GPU left them."""
        kb = np.ascontiguousarray(keys).view(np.uint32).copy()
        ids = np.arange(len(kb), dtype=np.uint32)
        n = C.c_int()
        self._check(lib().vsf_debug_retain_best(self._h, _p(kb), _p(ids), len(kb), n_points, int(use_lds), mode,
                                                C.byref(n)), "vsf_debug_retain_best")
        return kb[:n.value], ids[:n.value]

    def debug_sort_trim(self, matches: np.ndarray, best_percent: float = 1.0, serial: bool = False):
        """matches: (n_lists, n) DMATCH_DTYPE.  Returns a list of (queryIdx, trainIdx) int arrays, one per list: the first
        int(n * best_percent) matches in the order the device's restatement of std::sort leaves them."""
        m = np.ascontiguousarray(matches, DMATCH_DTYPE)
        if m.ndim == 1:
            m = m[None]
        nl, n = m.shape
        pairs = np.zeros((nl, max(n, 1), 2), np.uint64)
        counts = np.zeros(nl, np.int32)
        self._check(lib().vsf_debug_sort_trim(self._h, _p(m), nl, n, float(np.float32(best_percent)), int(serial),
                                              _p(pairs), _p(counts)), "vsf_debug_sort_trim")
        return [pairs[i, :counts[i]].astype(np.int64) for i in range(nl)]

    # ---- per-stage device timing ----
    def profile_enable(self, on: bool = True):
        self._check(lib().vsf_profile_enable(self._h, int(on)), "vsf_profile_enable")

    def profile_read(self, reset: bool = True) -> dict:
        """{stage name: (total ms, launches)} since the last reset (synchronises the stream)."""
        ms = (C.c_double * STAGE_COUNT)()
        n = (C.c_int64 * STAGE_COUNT)()
        self._check(lib().vsf_profile_read(self._h, ms, n, int(reset)), "vsf_profile_read")
        return {lib().vsf_stage_name(i).decode(): (ms[i], int(n[i])) for i in range(STAGE_COUNT)}

    # ---- introspection ----
    def debug_level_image(self, image: int, level: int, blurred: bool = False) -> np.ndarray:
        w, h, _, _ = self.level_info(level)
        out = np.empty((h, w), np.uint8)
        self._check(lib().vsf_debug_level_image(self._h, image, level, int(blurred), _p(out), w),
                    "vsf_debug_level_image")
        return out

    def debug_fast_candidates(self, image: int, level: int, cap: int = 1 << 17) -> np.ndarray:
        kp = np.zeros(cap, KEYPOINT_DTYPE)
        n = C.c_int()
        self._check(lib().vsf_debug_fast_candidates(self._h, image, level, _p(kp), cap, C.byref(n)),
                    "vsf_debug_fast_candidates")
        return kp[:min(n.value, cap)]

    def debug_level_keypoints(self, image: int, level: int, cap: int = 1 << 15) -> np.ndarray:
        kp = np.zeros(cap, KEYPOINT_DTYPE)
        n = C.c_int()
        self._check(lib().vsf_debug_level_keypoints(self._h, image, level, _p(kp), cap, C.byref(n)),
                    "vsf_debug_level_keypoints")
        return kp[:min(n.value, cap)]


def _u8(img: np.ndarray) -> np.ndarray:
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 2:
        raise ValueError("expected a 2-D uint8 image")
    if img.strides[1] != 1:
        img = np.ascontiguousarray(img)
    return img


def _desc(d: np.ndarray) -> np.ndarray:
    d = np.ascontiguousarray(d, np.uint8)
    return d.reshape(-1, DESC_BYTES)


def unpack_outputs(payload: np.ndarray):
    """Inverse of vsf_pack_outputs_dev: (list of VISION_FEATURE_DTYPE arrays per frame, list of FEATURE_MATCH_DTYPE
    arrays per pair).  `payload`: uint8 array holding at least the packed bytes."""
    b = np.ascontiguousarray(payload, np.uint8).reshape(-1)
    hdr = b[:16].view(np.uint32)
    if int(hdr[0]) != PAYLOAD_MAGIC:
        raise ValueError("not a packed output payload")
    nf, npairs, total = int(hdr[1]), int(hdr[2]), int(hdr[3])
    if total > len(b):
        raise ValueError("payload truncated: %d of %d bytes" % (len(b), total))
    counts = b[16:16 + 4 * (nf + npairs)].view(np.uint32).astype(np.int64)
    off = 16 + 4 * (nf + npairs)
    feats, matches = [], []
    for i in range(nf):
        n = int(counts[i]) * 28
        feats.append(b[off:off + n].view(VISION_FEATURE_DTYPE).copy())
        off += n
    for i in range(npairs):
        n = int(counts[nf + i]) * 16
        matches.append(b[off:off + n].view(FEATURE_MATCH_DTYPE).copy())
        off += n
    assert off == total
    return feats, matches


def decode_observation(buf: np.ndarray) -> dict:
    """Decodes vsf_observe_stereo's result (layout: include/vsf.h)."""
    b = np.ascontiguousarray(buf, np.uint8).reshape(-1)
    hdr = b[:64].view(np.uint32)
    if int(hdr[0]) != 0x4F465356:
        raise ValueError("not an observation payload")
    n_pairs, nfeat, total = int(hdr[1]), int(hdr[2]), int(hdr[3])
    assert total == len(b), (total, len(b))
    f32 = b[:64].view(np.float32)
    npairs = b[64:64 + 4 * n_pairs].view(np.uint32).astype(np.int64)
    off = 64 + 4 * ((n_pairs + 3) & ~3)
    feats = b[off:off + 28 * nfeat].view(VISION_FEATURE_DTYPE).copy()
    off += 28 * nfeat
    lists = []
    for p in range(n_pairs):
        lists.append(b[off:off + 16 * int(npairs[p])].view(FEATURE_MATCH_DTYPE).copy())
        off += 16 * int(npairs[p])
    kp = b[off:off + 28 * nfeat].view(KEYPOINT_DTYPE).copy()
    off += 28 * nfeat
    desc = b[off:off + 32 * nfeat].reshape(nfeat, 32).copy()
    off += 32 * nfeat
    assert off == total
    return {"n_left": int(hdr[4]), "n_right": int(hdr[5]), "n_stereo_matches": int(hdr[6]), "n_points": int(hdr[7]),
            "mean": np.float32(f32[8]), "threshold": np.float32(f32[9]), "threshold_next": np.float32(f32[10]),
            "features": feats, "factors": lists[:-1], "stereo_pairs": lists[-1], "keypoints": kp, "descriptors": desc}

"""ctypes view of the C++ host class slam::Frontend (vision_slam_frontend_amd/host/, libvsf_frontend.so), the
mirror of the reference's Frontend::ObserveImage / ObserveOdometry / GetSLAMProblem API on top of the HIP C ABI.
Used by tests and examples; the class itself is C++ because the reference's is."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import capi

LIB_PATH = Path(__file__).resolve().parent / "libvsf_frontend.so"
_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        capi.lib()  # libvsf_hip.so first (fails loudly if it was not built)
        if not LIB_PATH.exists():
            raise ImportError("%s is missing: run __graft_entry__.build()" % LIB_PATH)
        L = C.CDLL(str(LIB_PATH))
        vp, i32, f32, sz, dbl = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_double
        L.vsfh_frontend_create.argtypes = [i32, i32, i32, i32, vp, f32, i32]
        L.vsfh_frontend_create.restype = vp
        L.vsfh_frontend_destroy.argtypes = [vp]
        L.vsfh_observe_odometry.argtypes = [vp, vp, vp, dbl]
        L.vsfh_observe_image.argtypes = [vp, vp, vp, i32, i32, sz, dbl]
        L.vsfh_last_status.argtypes = [vp]
        L.vsfh_num_poses.argtypes = [vp]
        L.vsfh_stereo_ambig_constraint.argtypes = [vp]
        L.vsfh_stereo_ambig_constraint.restype = f32
        L.vsfh_get_fundamental.argtypes = [vp, vp]
        L.vsfh_num_vision_factors.argtypes = [vp]
        L.vsfh_vision_factor.argtypes = [vp, i32, vp, vp, vp, i32]
        L.vsfh_node.argtypes = [vp, i32, vp, vp, vp, vp, i32]
        L.vsfh_num_odometry_factors.argtypes = [vp]
        L.vsfh_odometry_factor.argtypes = [vp, i32, vp, vp]
        L.vsfh_frame.argtypes = [vp, i32, vp, vp, vp, i32]
        L.vsfh_serialize_problem.argtypes = [vp, vp, sz]
        L.vsfh_serialize_problem.restype = sz
        L.vsfh_default_calibration.argtypes = [C.POINTER(capi.VsfCalibration)]
        L.vsfh_default_calibration.restype = None
        L.vsfh_set_fused.argtypes = [vp, i32]
        L.vsfh_set_fused.restype = None
        L.vsfh_set_pipelined.argtypes = [vp, i32]
        L.vsfh_set_pipelined.restype = None
        L.vsfh_set_frames_in_flight.argtypes = [vp, i32]
        L.vsfh_set_frames_in_flight.restype = None
        L.vsfh_set_queue.argtypes = [vp, i32, i32, i32]
        L.vsfh_set_queue.restype = None
        L.vsfh_set_queue_threads.argtypes = [vp, i32, i32]
        L.vsfh_set_queue_threads.restype = None
        L.vsfh_time_sequence.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(dbl), C.POINTER(dbl)]
        L.vsfh_time_sequence.restype = dbl
        L.vsfh_left_cam_to_robot.argtypes = [vp, vp, vp]
        L.vsfh_left_cam_to_robot.restype = None
        L.vsfh_serialize_calibration.argtypes = [vp, vp, vp]
        L.vsfh_serialize_calibration.restype = None
        L.vsfh_flush.argtypes = [vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def default_calibration() -> capi.VsfCalibration:
    """FrontendConfig()'s stereo calibration (the reference's hard-coded constants, slam_frontend.cc:565-644)."""
    c = capi.VsfCalibration()
    lib().vsfh_default_calibration(C.byref(c))
    return c


class Frontend:
    def __init__(self, width: int, height: int, nfeatures: int = 10000, device: int = 0, fundamental=None,
                 best_percent: float = 0.0, frame_life: int = 0):
        F = None if fundamental is None else np.ascontiguousarray(fundamental, np.float32).reshape(9)
        self._h = lib().vsfh_frontend_create(nfeatures, width, height, device, _p(F), best_percent, frame_life)
        self.cap = nfeatures + 256
        st = lib().vsfh_last_status(self._h)
        if st != capi.VSF_OK:
            raise capi.VsfError(st, "Frontend")

    def set_fused(self, on: bool):
        """True (default): ObserveImage is one GPU submission (vsf_observe_stereo); False: one C-ABI call per
        reference call with the reference's host steps in between.  Choose before the first observe_image."""
        lib().vsfh_set_fused(self._h, int(on))

    def set_pipelined(self, on: bool):
        """observe_image queues its frame on the GPU and returns (its return value is the odometry gate's decision); results
        are collected and booked, in frame order, two frames later or when the problem is read.  Fused mode; choose before the
        first observe_image."""
        lib().vsfh_set_pipelined(self._h, int(on))

    def set_frames_in_flight(self, n: int):
        """How many frames a pipelined Frontend leaves in the context's queue (1..1024; default 256).  Choose before the first
        observe_image."""
        lib().vsfh_set_frames_in_flight(self._h, int(n))

    def set_queue(self, depth: int = 0, batch_frames: int = 0, min_batch: int = 0):
        """The ObserveImage queue of a pipelined Frontend: frames that may wait uncollected (default 256), frames per batch at
        most (default 128: sizes the context), fewest waiting frames that leave while the GPU is busy (0: a whole batch, or half the depth when that is less).
        Choose before the first observe_image."""
        lib().vsfh_set_queue(self._h, int(depth), int(batch_frames), int(min_batch))

    def set_queue_threads(self, launcher: bool = False, copy: bool = True):
        """The queue's host threads (VSF_OPT_OBSERVE_THREAD / VSF_OPT_OBSERVE_COPY_THREAD).  Choose before the first
        observe_image."""
        lib().vsfh_set_queue_threads(self._h, int(launcher), int(copy))

    def time_sequence(self, frames: np.ndarray, n_frames: int, warm: int = 32, read_every: int = 0):
        """The reference's driver loop in C++ (vsfh_time_sequence) over `frames` [n][2][h][w] taken in turn: returns
        (steady frames per second, mean ms inside ObserveImage, max ms)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        assert frames.ndim == 4 and frames.shape[1] == 2
        mean, worst = C.c_double(), C.c_double()
        fps = lib().vsfh_time_sequence(self._h, _p(frames), frames.shape[0], frames.shape[3], frames.shape[2], int(n_frames),
                                       int(warm), int(read_every), C.byref(mean), C.byref(worst))
        if fps < 0:
            raise capi.VsfError(lib().vsfh_last_status(self._h), "Frontend::ObserveImage (time_sequence)")
        return float(fps), float(mean.value), float(worst.value)

    def flush(self) -> bool:
        """Collects and books every frame still in flight."""
        return bool(lib().vsfh_flush(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib().vsfh_frontend_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def observe_odometry(self, translation, rotation_wxyz, timestamp: float):
        t = np.ascontiguousarray(translation, np.float32)
        q = np.ascontiguousarray(rotation_wxyz, np.float32)
        lib().vsfh_observe_odometry(self._h, _p(t), _p(q), timestamp)

    def observe_image(self, left: np.ndarray, right: np.ndarray, time: float = 0.0) -> bool:
        left, right = np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(right, np.uint8)
        assert left.shape == right.shape and left.ndim == 2
        added = bool(lib().vsfh_observe_image(self._h, _p(left), _p(right), left.shape[1], left.shape[0],
                                              left.strides[0], time))
        st = lib().vsfh_last_status(self._h)
        if st != capi.VSF_OK:
            raise capi.VsfError(st, "Frontend::ObserveImage")
        return added

    @property
    def num_poses(self) -> int:
        return lib().vsfh_num_poses(self._h)

    @property
    def stereo_ambig_constraint(self) -> float:
        return float(lib().vsfh_stereo_ambig_constraint(self._h))

    @property
    def fundamental(self) -> np.ndarray:
        F = np.zeros(9, np.float32)
        lib().vsfh_get_fundamental(self._h, _p(F))
        return F.reshape(3, 3)

    def vision_factors(self):
        out = []
        for i in range(lib().vsfh_num_vision_factors(self._h)):
            a, b = C.c_uint64(), C.c_uint64()
            pairs = np.zeros((self.cap, 2), np.uint64)
            n = lib().vsfh_vision_factor(self._h, i, C.byref(a), C.byref(b), _p(pairs), self.cap)
            out.append((a.value, b.value, pairs[:n].copy()))
        return out

    def nodes(self):
        out = []
        for i in range(self.num_poses):
            idx, ts = C.c_uint64(), C.c_double()
            pose = np.zeros(7, np.float32)
            feat = np.zeros((self.cap, 6), np.float32)
            n = lib().vsfh_node(self._h, i, C.byref(idx), C.byref(ts), _p(pose), _p(feat), self.cap)
            out.append({"node_idx": idx.value, "timestamp": ts.value, "pose": pose, "features": feat[:n].copy()})
        return out

    def odometry_factors(self):
        out = []
        for i in range(lib().vsfh_num_odometry_factors(self._h)):
            ij = np.zeros(2, np.uint64)
            tq = np.zeros(7, np.float32)
            lib().vsfh_odometry_factor(self._h, i, _p(ij), _p(tq))
            out.append((int(ij[0]), int(ij[1]), tq))
        return out

    @property
    def left_cam_to_robot(self):
        """GetConfig().left_cam_to_robot (slam_frontend.h:96): (rotation 3 x 3, translation 3)."""
        R, t = np.zeros(9, np.float32), np.zeros(3, np.float32)
        lib().vsfh_left_cam_to_robot(self._h, _p(R), _p(t))
        return R.reshape(3, 3), t

    def serialize_calibration(self):
        """ROS-1 payloads of the CameraExtrinsics (48 B) and CameraIntrinsics (32 B) messages the reference's driver writes
        beside the problem (slam_frontend_main.cc:341-365)."""
        e, k = np.zeros(48, np.uint8), np.zeros(32, np.uint8)
        lib().vsfh_serialize_calibration(self._h, _p(e), _p(k))
        return e.tobytes(), k.tobytes()

    def serialize_problem(self) -> bytes:
        """ROS-1 wire bytes of vision_slam_frontend/SLAMProblem for everything observed so far (host/slam_to_ros.h)."""
        n = lib().vsfh_serialize_problem(self._h, None, 0)
        buf = np.zeros(max(n, 1), np.uint8)
        lib().vsfh_serialize_problem(self._h, _p(buf), n)
        return buf[:n].tobytes()

    def frame(self, i: int):
        fid = C.c_uint64()
        kp = np.zeros(self.cap, capi.KEYPOINT_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = lib().vsfh_frame(self._h, i, C.byref(fid), _p(kp), _p(desc), self.cap)
        if n < 0:
            raise IndexError(i)
        return fid.value, kp[:n].copy(), desc[:n].copy()

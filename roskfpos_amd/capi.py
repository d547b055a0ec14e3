"""ctypes binding of libkfpos_hip.so (include/kfpos.h) -- the only compute path of this package.

There is no CPU fallback: importing works anywhere, but creating a bank raises if the HIP
library has not been built (`python -c "import __graft_entry__ as g; g.build()"`) or no GPU
is present. Mirrors the reference's estimator interface batched over T tags:
PositionEstimationAlgorithm (src/kfpos/algorithms/PositionEstimationAlgorithm.h:8-37).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KFPOS_LIB_PATH") or os.path.join(_DIR, "csrc", "libkfpos_hip.so")  # override: A/B builds

MODEL_TOA, MODEL_TOA_IMU, MODEL_ML, MODEL_PLANAR = 0, 1, 2, 3
SENSOR_PX4FLOW, SENSOR_IMU, SENSOR_MAG, SENSOR_COMPASS = 1, 2, 3, 4
_SENSOR_WIDTH = {1: 5, 2: 24, 3: 3, 4: 1}
STORE_F64, STORE_F32, STORE_MIXED, STORE_P48 = 0, 1, 2, 3
ML_NORMAL, ML_IGNORE_N, ML_BEST = 0, 1, 2
MAX_ANCHORS = 64
ST_UPDATE_SKIPPED, ST_ML_FALLBACK, ST_FEW_RANGES, ST_ML_INIT, ST_NOT_STARTED, ST_NONFINITE = 1, 2, 4, 8, 16, 32
ST_SKIPPED = 64

# every symbol include/kfpos.h declares
EXPORTS = [
    "kfpos_create", "kfpos_destroy", "kfpos_init", "kfpos_set_anchors", "kfpos_set_init_positions",
    "kfpos_real_size", "kfpos_step_toa", "kfpos_step_imu", "kfpos_step_toa_imu", "kfpos_get_pose",
    "kfpos_get_pose_each", "kfpos_get_predicted",
    "kfpos_state_dim", "kfpos_get_state", "kfpos_set_state", "kfpos_step_toa_dev", "kfpos_step_imu_dev",
    "kfpos_step_toa_imu_dev", "kfpos_get_pose_dev", "kfpos_run_trace_dev", "kfpos_last_error",
    "kfpos_strerror", "kfpos_version", "kfpos_timing_begin", "kfpos_timing_end",
    "kfpos_set_planar", "kfpos_step_sensor", "kfpos_step_sensor_dev", "kfpos_get_height", "kfpos_set_height",
    "kfpos_latch_dim", "kfpos_get_latch", "kfpos_set_latch",
    "kfpos_slot_count", "kfpos_slot_acquire", "kfpos_slot_submit", "kfpos_slot_wait",
    "kfpos_shard_range", "kfpos_comm_unique_id", "kfpos_comm_create", "kfpos_comm_create_all", "kfpos_comm_destroy",
    "kfpos_comm_world", "kfpos_comm_rank", "kfpos_comm_set_total", "kfpos_allgather_poses",
    "kfpos_allgather_poses_multi", "kfpos_comm_wait", "kfpos_comm_sync", "kfpos_assemble_poses_dev",
    "kfpos_comm_backend_version", "kfpos_comm_set_algorithm", "kfpos_comm_algorithm",
]
GATHER_COLLECTIVE, GATHER_DIRECT = 0, 1
COMM_ID_BYTES = 128
SLOT_TOA, SLOT_IMU, SLOT_TOA_IMU = 0, 1, 2
SLOT_DT_PER_TAG, SLOT_REUSE_ERR, SLOT_REUSE_COV, SLOT_NO_POSE = 0x100, 0x200, 0x400, 0x800


class KfposError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("model", C.c_int32), ("n_tags", C.c_int32), ("max_anchors", C.c_int32),
                ("storage", C.c_int32), ("accel_noise", C.c_double), ("jolt", C.c_double),
                ("ignore_worst", C.c_int32), ("cost_threshold", C.c_double), ("top_n", C.c_int32),
                ("use_init_pos", C.c_int32), ("init_pos", C.c_double * 3), ("device", C.c_int32),
                ("ml_variant", C.c_int32)]


class PlanarConfig(C.Structure):
    """kfpos_planar_config: what KalmanFilter::loadConfigurationFiles reads + initialAngle."""
    _fields_ = [("use_fixed_height", C.c_int32), ("fixed_height", C.c_double), ("init_angle", C.c_double),
                ("px4_height", C.c_double), ("px4_arm_p1", C.c_double), ("px4_arm_p2", C.c_double),
                ("px4_cov_velocity", C.c_double), ("px4_cov_gyro_z", C.c_double),
                ("imu_use_fixed_cov_acc", C.c_int32), ("imu_cov_acc", C.c_double),
                ("imu_use_fixed_cov_ang_vel_z", C.c_int32), ("imu_cov_ang_vel_z", C.c_double),
                ("mag_angle_offset", C.c_double), ("mag_cov", C.c_double)]


class _EpochSlot(C.Structure):
    """kfpos_epoch_slot: pointers into one slot of the handle's pinned host memory."""
    _fields_ = [("range_mm", C.c_void_p), ("err_est", C.c_void_p), ("accel", C.c_void_p), ("cov", C.c_void_p),
                ("dt", C.c_void_p), ("status", C.c_void_p), ("pos", C.c_void_p)]


_lib = None


def load():
    """Load the HIP library; raises KfposError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KfposError(f"{LIB_PATH} is not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    try:
        # PyTorch wheels bundle their own libamdhip64.so.7; whichever copy is loaded first serves the whole
        # process. If torch is going to share this process (device tensors, RCCL) its copy has to be that
        # one, otherwise torch.cuda later reports "No HIP GPUs are available".
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, f64 = C.c_void_p, C.c_int32, C.c_double

    def sig(name, argtypes):
        # an A/B build from an older revision (KFPOS_LIB_PATH) may lack the newest entry points: calling one then
        # fails with AttributeError, everything else works
        if hasattr(L, name):
            getattr(L, name).argtypes = argtypes

    sig("kfpos_create", [C.POINTER(_Config), C.POINTER(vp)])
    sig("kfpos_destroy", [vp])
    sig("kfpos_init", [vp])
    sig("kfpos_set_anchors", [vp, vp, vp, i32])
    sig("kfpos_set_init_positions", [vp, vp])
    sig("kfpos_real_size", [vp])
    sig("kfpos_state_dim", [vp])
    sig("kfpos_step_toa", [vp, vp, vp, vp, i32, vp])
    sig("kfpos_step_imu", [vp, vp, vp, vp, i32, vp])
    sig("kfpos_step_toa_imu", [vp, vp, vp, vp, vp, vp, i32, vp])
    sig("kfpos_get_pose", [vp, f64, vp, vp, vp, vp])
    sig("kfpos_get_pose_each", [vp, vp, vp, vp, vp, vp])
    sig("kfpos_get_predicted", [vp, vp, i32, vp, vp, vp])
    sig("kfpos_get_state", [vp, vp, vp, vp])
    sig("kfpos_set_state", [vp, vp, vp, vp])
    sig("kfpos_step_toa_dev", [vp, vp, vp, vp, f64, vp, vp])
    sig("kfpos_step_imu_dev", [vp, vp, vp, vp, f64, vp, vp])
    sig("kfpos_step_toa_imu_dev", [vp, vp, vp, vp, vp, i32, vp, f64, vp, vp])
    sig("kfpos_get_pose_dev", [vp, f64, vp, vp, vp, vp, vp])
    L.kfpos_run_trace_dev.argtypes = [vp, i32, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64,
                                      vp, vp, vp, vp]
    sig("kfpos_set_planar", [vp, C.POINTER(PlanarConfig)])
    sig("kfpos_step_sensor", [vp, i32, vp, vp, i32, vp])
    sig("kfpos_step_sensor_dev", [vp, i32, vp, vp, f64, vp, vp])
    sig("kfpos_get_height", [vp, vp])
    sig("kfpos_set_height", [vp, vp])
    sig("kfpos_latch_dim", [vp])
    sig("kfpos_get_latch", [vp, vp])
    sig("kfpos_set_latch", [vp, vp])
    sig("kfpos_slot_count", [vp])
    sig("kfpos_slot_acquire", [vp, i32, C.POINTER(_EpochSlot)])
    sig("kfpos_slot_submit", [vp, i32, i32, f64])
    sig("kfpos_slot_wait", [vp, i32])
    sig("kfpos_timing_begin", [vp, vp])
    sig("kfpos_timing_end", [vp, vp, C.POINTER(C.c_float)])
    i64 = C.c_int64
    sig("kfpos_shard_range", [i64, i32, i32, C.POINTER(i64), C.POINTER(i64)])
    sig("kfpos_comm_unique_id", [vp])
    sig("kfpos_comm_create", [i32, i32, vp, i32, C.POINTER(vp)])
    sig("kfpos_comm_create_all", [i32, vp, C.POINTER(vp)])
    sig("kfpos_comm_destroy", [vp])
    sig("kfpos_comm_world", [vp])
    sig("kfpos_comm_rank", [vp])
    sig("kfpos_comm_set_total", [vp, i64, C.POINTER(i64), C.POINTER(i64)])
    sig("kfpos_allgather_poses", [vp, vp, vp, i32, vp, vp])
    sig("kfpos_allgather_poses_multi", [i32, vp, vp, vp, i32, vp, vp])
    sig("kfpos_comm_wait", [vp, vp])
    sig("kfpos_comm_sync", [vp])
    sig("kfpos_assemble_poses_dev", [i32, i32, i64, vp, vp, i32, vp])
    sig("kfpos_comm_backend_version", [])
    sig("kfpos_comm_set_algorithm", [vp, i32])
    sig("kfpos_comm_algorithm", [vp])
    L.kfpos_last_error.restype = C.c_char_p
    L.kfpos_strerror.restype = C.c_char_p
    sig("kfpos_strerror", [C.c_int])
    _lib = L
    return L


def _ptr(x):
    """device pointer from an int, None, or anything with data_ptr() (torch tensors)."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


class KfposBank:
    """T independent filters on one GPU (KalmanFilterTOA or KalmanFilterTOAIMU semantics per tag)."""

    def __init__(self, model, n_tags, anchors, storage=STORE_F64, accel_noise=0.5, jolt=0.5,
                 ignore_worst=False, cost_threshold=0.5, top_n=0, init_pos=None, device=0,
                 max_anchors=None, planar=None, ml_variant=0):
        self._h = None
        self.lib = load()
        anchors = np.ascontiguousarray(anchors, dtype=np.float64)
        self.T, self.A = int(n_tags), int(max_anchors or anchors.shape[0])
        self.model, self.storage = model, storage
        self.real = np.float64 if storage == STORE_F64 else np.float32  # kfpos_real: measurement element type
        cfg = _Config()
        cfg.model, cfg.n_tags, cfg.max_anchors, cfg.storage = model, self.T, self.A, storage
        cfg.accel_noise, cfg.jolt = accel_noise, jolt
        cfg.ignore_worst, cfg.cost_threshold, cfg.top_n = int(ignore_worst), cost_threshold, top_n
        cfg.use_init_pos = int(init_pos is not None)
        per_tag = None
        if init_pos is not None:
            ip = np.asarray(init_pos, dtype=np.float64)
            if ip.shape == (3,):
                cfg.init_pos = (C.c_double * 3)(*ip)
            else:
                assert ip.shape == (self.T, 3)
                per_tag = np.ascontiguousarray(ip)
        cfg.device = device
        cfg.ml_variant = ml_variant  # MODEL_ML: ML_NORMAL / ML_IGNORE_N / ML_BEST
        h = C.c_void_p()
        self._chk(self.lib.kfpos_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._chk(self.lib.kfpos_init(h))
        self._chk(self.lib.kfpos_set_anchors(h, anchors.ctypes.data, None, anchors.shape[0]))
        if model == MODEL_PLANAR:  # KalmanFilter::init(): the XML configuration (dict of PlanarConfig fields)
            self.planar = PlanarConfig(**(planar or {}))
            self._chk(self.lib.kfpos_set_planar(h, C.byref(self.planar)))
        if per_tag is not None:
            self._chk(self.lib.kfpos_set_init_positions(h, per_tag.ctypes.data))
        self.n = self.lib.kfpos_state_dim(h)

    def _chk(self, rc):
        if rc != 0:
            msg = self.lib.kfpos_strerror(rc).decode()
            detail = self.lib.kfpos_last_error().decode()
            if detail:
                msg += ": " + detail
            raise KfposError(msg)

    def close(self):
        if self._h:
            self.lib.kfpos_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-buffer API ------------------------------------------------------
    def _dt(self, dt):
        d = np.atleast_1d(np.ascontiguousarray(dt, dtype=np.float64))
        assert d.size in (1, self.T)
        return d

    def step_toa(self, range_mm, err_est, dt):
        r = np.ascontiguousarray(range_mm, dtype=np.int32)
        e = np.ascontiguousarray(err_est, dtype=self.real)
        assert r.shape == (self.T, self.A) and e.shape == (self.T, self.A)
        d, st = self._dt(dt), np.zeros(self.T, dtype=np.uint32)
        self._chk(self.lib.kfpos_step_toa(self._h, r.ctypes.data, e.ctypes.data, d.ctypes.data, d.size,
                                          st.ctypes.data))
        return st

    def step_imu(self, accel, cov, dt):
        a = np.ascontiguousarray(accel, dtype=self.real)
        c = np.ascontiguousarray(cov, dtype=self.real)
        assert a.shape == (self.T, 3) and c.shape == (self.T, 9)
        d, st = self._dt(dt), np.zeros(self.T, dtype=np.uint32)
        self._chk(self.lib.kfpos_step_imu(self._h, a.ctypes.data, c.ctypes.data, d.ctypes.data, d.size,
                                          st.ctypes.data))
        return st

    def step_toa_imu(self, range_mm, err_est, accel, cov, dt):
        r = np.ascontiguousarray(range_mm, dtype=np.int32)
        e = np.ascontiguousarray(err_est, dtype=self.real)
        a = np.ascontiguousarray(accel, dtype=self.real)
        c = np.ascontiguousarray(cov, dtype=self.real)
        assert r.shape == (self.T, self.A) and e.shape == (self.T, self.A)
        assert a.shape == (self.T, 3) and c.shape == (self.T, 9)
        d, st = self._dt(dt), np.zeros(self.T, dtype=np.uint32)
        self._chk(self.lib.kfpos_step_toa_imu(self._h, r.ctypes.data, e.ctypes.data, a.ctypes.data,
                                              c.ctypes.data, d.ctypes.data, d.size, st.ctypes.data))
        return st

    def step_sensor(self, kind, data, dt):
        """KalmanFilter's other four entry points (MODEL_PLANAR): SENSOR_PX4FLOW (T, 5), SENSOR_IMU (T, 24),
        SENSOR_MAG (T, 3), SENSOR_COMPASS (T,)."""
        x = np.ascontiguousarray(data, dtype=np.float64).reshape(self.T, -1)
        assert x.shape[1] == _SENSOR_WIDTH[kind]
        d, st = self._dt(dt), np.zeros(self.T, dtype=np.uint32)
        self._chk(self.lib.kfpos_step_sensor(self._h, kind, x.ctypes.data, d.ctypes.data, d.size, st.ctypes.data))
        return st

    def step_px4flow(self, flow, dt):
        return self.step_sensor(SENSOR_PX4FLOW, flow, dt)

    def step_planar_imu(self, ang_vel, cov_ang_vel, lin_acc, cov_acc, dt):
        return self.step_sensor(SENSOR_IMU, np.concatenate([ang_vel, cov_ang_vel, lin_acc, cov_acc], axis=1), dt)

    def step_mag(self, mag_xyz, dt):
        return self.step_sensor(SENSOR_MAG, mag_xyz, dt)

    def step_compass(self, compass, dt):
        return self.step_sensor(SENSOR_COMPASS, compass, dt)

    def get_height(self):
        z = np.zeros(self.T)
        self._chk(self.lib.kfpos_get_height(self._h, z.ctypes.data))
        return z

    def set_height(self, z):
        a = np.ascontiguousarray(z, dtype=np.float64)
        assert a.shape == (self.T,)
        self._chk(self.lib.kfpos_set_height(self._h, a.ctypes.data))

    def get_pose(self, dt_ahead=0.0):
        pos, cov, vel = np.zeros((self.T, 3)), np.zeros((self.T, 9)), np.zeros((self.T, 3))
        st = np.zeros(self.T, dtype=np.uint32)
        self._chk(self.lib.kfpos_get_pose(self._h, dt_ahead, pos.ctypes.data, cov.ctypes.data,
                                          vel.ctypes.data, st.ctypes.data))
        return pos, cov.reshape(self.T, 3, 3), vel, st

    def get_pose_each(self, dt_ahead):
        d = np.ascontiguousarray(dt_ahead, dtype=np.float64)
        assert d.shape == (self.T,)
        pos, cov, vel = np.zeros((self.T, 3)), np.zeros((self.T, 9)), np.zeros((self.T, 3))
        st = np.zeros(self.T, dtype=np.uint32)
        self._chk(self.lib.kfpos_get_pose_each(self._h, d.ctypes.data, pos.ctypes.data, cov.ctypes.data,
                                               vel.ctypes.data, st.ctypes.data))
        return pos, cov.reshape(self.T, 3, 3), vel, st

    def get_predicted(self, dt_ahead):
        d = np.atleast_1d(np.ascontiguousarray(dt_ahead, dtype=np.float64))
        assert d.size in (1, self.T)
        x, P = np.zeros((self.T, self.n)), np.zeros((self.T, self.n, self.n))
        st = np.zeros(self.T, dtype=np.uint32)
        self._chk(self.lib.kfpos_get_predicted(self._h, d.ctypes.data, d.size, x.ctypes.data, P.ctypes.data,
                                               st.ctypes.data))
        return x, P, st

    def get_state(self):
        x, P = np.zeros((self.T, self.n)), np.zeros((self.T, self.n, self.n))
        fl = np.zeros(self.T, dtype=np.uint32)
        self._chk(self.lib.kfpos_get_state(self._h, x.ctypes.data, P.ctypes.data, fl.ctypes.data))
        return x, P, fl

    def set_state(self, x, P, flags=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        P = np.ascontiguousarray(P, dtype=np.float64)
        fl = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint32)
        self._chk(self.lib.kfpos_set_state(self._h, x.ctypes.data, P.ctypes.data,
                                           None if fl is None else fl.ctypes.data))

    def get_latch(self):
        """Latched sensor samples (T, kfpos_latch_dim): with get_state() a complete checkpoint."""
        out = np.zeros((self.T, self.lib.kfpos_latch_dim(self._h)))
        self._chk(self.lib.kfpos_get_latch(self._h, out.ctypes.data if out.size else None))
        return out

    def set_latch(self, latch):
        a = np.ascontiguousarray(latch, dtype=np.float64)
        assert a.shape == (self.T, self.lib.kfpos_latch_dim(self._h))
        self._chk(self.lib.kfpos_set_latch(self._h, a.ctypes.data if a.size else None))

    # ---- streaming host API: epochs assembled in place in pinned, component-major slots ----
    def slot_acquire(self, slot):
        """Wait for the slot's previous submission; returns numpy views over the slot's pinned memory:
        dict(range_mm [A][T] int32, err_est [A][T], accel [3][T], cov [9][T], dt [T], status [T], pos [3][T])."""
        sl = _EpochSlot()
        self._chk(self.lib.kfpos_slot_acquire(self._h, slot, C.byref(sl)))
        T, A = self.T, self.A

        def view(ptr, shape, dtype):
            n = int(np.prod(shape))
            buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)

        return {"range_mm": view(sl.range_mm, (A, T), np.int32), "err_est": view(sl.err_est, (A, T), self.real),
                "accel": view(sl.accel, (3, T), self.real), "cov": view(sl.cov, (9, T), self.real),
                "dt": view(sl.dt, (T,), np.float64), "status": view(sl.status, (T,), np.uint32),
                "pos": view(sl.pos, (3, T), np.float64)}

    def slot_submit(self, slot, flags, dt_shared=0.0):
        self._chk(self.lib.kfpos_slot_submit(self._h, slot, int(flags), float(dt_shared)))

    def slot_wait(self, slot):
        self._chk(self.lib.kfpos_slot_wait(self._h, slot))

    # ---- device-buffer API (pointers: ints or torch tensors; layouts in include/kfpos.h) ----
    def step_toa_dev(self, range_mm, err_est, dt, status=None, stream=None, dt_dev=None):
        self._chk(self.lib.kfpos_step_toa_dev(self._h, _ptr(range_mm), _ptr(err_est), _ptr(dt_dev), float(dt),
                                              _ptr(status), _ptr(stream)))

    def step_imu_dev(self, accel, cov, dt, status=None, stream=None, dt_dev=None):
        self._chk(self.lib.kfpos_step_imu_dev(self._h, _ptr(accel), _ptr(cov), _ptr(dt_dev), float(dt),
                                              _ptr(status), _ptr(stream)))

    def step_toa_imu_dev(self, range_mm, err_est, accel, cov, dt, latch=True, status=None, stream=None,
                         dt_dev=None):
        self._chk(self.lib.kfpos_step_toa_imu_dev(self._h, _ptr(range_mm), _ptr(err_est), _ptr(accel),
                                                  _ptr(cov), int(latch), _ptr(dt_dev), float(dt),
                                                  _ptr(status), _ptr(stream)))

    def step_sensor_dev(self, kind, data, dt, status=None, stream=None, dt_dev=None):
        self._chk(self.lib.kfpos_step_sensor_dev(self._h, int(kind), _ptr(data), _ptr(dt_dev), float(dt),
                                                 _ptr(status), _ptr(stream)))

    def get_pose_dev(self, dt_ahead, pos=None, cov=None, vel=None, status=None, stream=None):
        self._chk(self.lib.kfpos_get_pose_dev(self._h, float(dt_ahead), _ptr(pos), _ptr(cov), _ptr(vel),
                                              _ptr(status), _ptr(stream)))

    def run_trace_dev(self, n_steps, range_mm, stride_ranges, err_est, stride_err, dt_steps, accel=None,
                      stride_accel=0, cov=None, stride_cov=0, trajectory=None, status=None, stream=None):
        d = np.ascontiguousarray(dt_steps, dtype=np.float64)
        assert d.size >= n_steps
        self._chk(self.lib.kfpos_run_trace_dev(self._h, n_steps, _ptr(range_mm), stride_ranges, _ptr(err_est),
                                               stride_err, _ptr(accel), stride_accel, _ptr(cov), stride_cov,
                                               d.ctypes.data, _ptr(trajectory), _ptr(status), _ptr(stream)))

    def timing_begin(self, stream=None):
        self._chk(self.lib.kfpos_timing_begin(self._h, _ptr(stream)))

    def timing_end(self, stream=None) -> float:
        ms = C.c_float()
        self._chk(self.lib.kfpos_timing_end(self._h, _ptr(stream), C.byref(ms)))
        return float(ms.value)


# ---- multi-GPU: the RCCL pose all-gather behind the C ABI (include/kfpos.h: kfpos_comm_*) ----
def _chk_lib(lib, rc):
    if rc != 0:
        msg = lib.kfpos_strerror(rc).decode()
        detail = lib.kfpos_last_error().decode()
        raise KfposError(msg + (": " + detail if detail else ""))


def shard_range(total_tags: int, world: int, rank: int):
    """kfpos_shard_range: contiguous [lo, hi) of `rank`; host arithmetic only."""
    lo, hi = C.c_int64(), C.c_int64()
    lib = load()
    _chk_lib(lib, lib.kfpos_shard_range(total_tags, world, rank, C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library: rank 0 makes it, every rank gets the same 128 bytes."""
    lib = load()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _chk_lib(lib, lib.kfpos_comm_unique_id(buf))
    return buf.raw


def assemble_poses_dev(world, rows, total_tags, staged, out, device=0, stream=None):
    """[world][rows][t_pad] gathered by any collective -> [rows][total] in global tag order (device tensors)."""
    lib = load()
    _chk_lib(lib, lib.kfpos_assemble_poses_dev(world, rows, total_tags, _ptr(staged), _ptr(out), device, _ptr(stream)))


class KfposComm:
    """One rank's end of the pose all-gather (kfpos_comm): RCCL communicator + side stream + two buffer sets."""

    def __init__(self, world, rank, unique_id: bytes, device=0, _handle=None):
        self.lib = load()
        self._c = None
        if _handle is not None:
            self._c = _handle
        else:
            assert len(unique_id) == COMM_ID_BYTES
            c = C.c_void_p()
            _chk_lib(self.lib, self.lib.kfpos_comm_create(world, rank, unique_id, device, C.byref(c)))
            self._c = c
        self.world, self.rank, self.device = self.lib.kfpos_comm_world(self._c), self.lib.kfpos_comm_rank(self._c), device
        self.lo = self.hi = None

    @classmethod
    def create_all(cls, devices):
        """One process, several GPUs (ncclCommInitAll): a list of communicators, one per device."""
        lib = load()
        n = len(devices)
        devs = (C.c_int32 * n)(*devices)
        out = (C.c_void_p * n)()
        _chk_lib(lib, lib.kfpos_comm_create_all(n, devs, out))
        return [cls(n, i, b"", devices[i], _handle=C.c_void_p(out[i])) for i in range(n)]

    def set_total(self, total_tags: int):
        lo, hi = C.c_int64(), C.c_int64()
        _chk_lib(self.lib, self.lib.kfpos_comm_set_total(self._c, total_tags, C.byref(lo), C.byref(hi)))
        self.lo, self.hi, self.total = lo.value, hi.value, total_tags
        return self.lo, self.hi

    def allgather(self, pos_all, pos_local=None, rows=3, bank=None, stream=None):
        """pos_local [rows][t_local] (or the bank's current positions) -> pos_all [rows][total]; asynchronous."""
        _chk_lib(self.lib, self.lib.kfpos_allgather_poses(bank._h if bank is not None else None, self._c,
                                                          _ptr(pos_local), rows, _ptr(pos_all), _ptr(stream)))

    def set_algorithm(self, algorithm: int):
        """GATHER_COLLECTIVE (ncclAllGather) or GATHER_DIRECT (grouped ncclSend / ncclRecv to every peer)."""
        _chk_lib(self.lib, self.lib.kfpos_comm_set_algorithm(self._c, algorithm))

    @property
    def algorithm(self) -> int:
        return self.lib.kfpos_comm_algorithm(self._c)

    def wait(self, stream=None):
        _chk_lib(self.lib, self.lib.kfpos_comm_wait(self._c, _ptr(stream)))

    def sync(self):
        _chk_lib(self.lib, self.lib.kfpos_comm_sync(self._c))

    def close(self):
        if self._c:
            self.lib.kfpos_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def allgather_poses_multi(comms, pos_all, pos_local=None, rows=3, banks=None, streams=None):
    """kfpos_allgather_poses_multi: the gathers of all communicators of one process in one RCCL group."""
    lib = load()
    n = len(comms)
    arr = lambda xs: (C.c_void_p * n)(*[_ptr(x) for x in xs]) if xs is not None else None  # noqa: E731
    _chk_lib(lib, lib.kfpos_allgather_poses_multi(
        n, arr([b._h.value for b in banks]) if banks is not None else None, arr([c._c.value for c in comms]),
        arr(pos_local), rows, arr(pos_all), arr(streams)))

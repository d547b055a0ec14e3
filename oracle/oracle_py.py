"""ctypes binding of the CPU oracle (oracle/libkfpos_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package. PARITY UNPINNED (see kfpos_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libkfpos_oracle.so")

MODEL_TOA, MODEL_TOA_IMU, MODEL_ML, MODEL_PLANAR = 0, 1, 2, 3
ST_UPDATE_SKIPPED, ST_ML_FALLBACK, ST_FEW_RANGES, ST_ML_INIT, ST_NOT_STARTED, ST_NONFINITE, ST_SKIPPED = 1, 2, 4, 8, 16, 32, 64

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_up = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


class PlanarConfig(C.Structure):
    """kfo_planar_config: the XML attributes KalmanFilter::loadConfigurationFiles reads + initialAngle."""
    _fields_ = [("use_fixed_height", C.c_int), ("fixed_height", C.c_double), ("init_angle", C.c_double),
                ("px4_height", C.c_double), ("px4_arm_p1", C.c_double), ("px4_arm_p2", C.c_double),
                ("px4_cov_velocity", C.c_double), ("px4_cov_gyro_z", C.c_double),
                ("imu_use_fixed_cov_acc", C.c_int), ("imu_cov_acc", C.c_double),
                ("imu_use_fixed_cov_ang_vel_z", C.c_int), ("imu_cov_ang_vel_z", C.c_double),
                ("mag_angle_offset", C.c_double), ("mag_cov", C.c_double)]


def build(force: bool = False) -> str:
    src = os.path.join(_DIR, "kfpos_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _DIR, "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.kfo_create.restype = C.c_void_p
        L.kfo_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double,
                                 C.c_int, C.c_int, C.c_void_p]
        L.kfo_destroy.argtypes = [C.c_void_p]
        L.kfo_set_ml_variant.argtypes = [C.c_void_p, C.c_int]
        L.kfo_state_dim.argtypes = [C.c_void_p]
        L.kfo_set_anchors.argtypes = [C.c_void_p, _dp, C.c_int]
        L.kfo_step_toa.argtypes = [C.c_void_p, _ip, _dp, _dp, C.c_int, C.c_void_p, C.c_int]
        L.kfo_step_imu.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int, C.c_void_p, C.c_int]
        L.kfo_get_pose.argtypes = [C.c_void_p, C.c_double, _dp, _dp, C.c_void_p, C.c_void_p]
        L.kfo_get_state.argtypes = [C.c_void_p, _dp, _dp]
        L.kfo_set_state.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
        L.kfo_set_planar.argtypes = [C.c_void_p, C.POINTER(PlanarConfig)]
        L.kfo_get_height.argtypes = [C.c_void_p, _dp]
        L.kfo_planar_px4flow.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_void_p, C.c_int]
        L.kfo_planar_imu.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_void_p, C.c_int]
        L.kfo_planar_mag.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_void_p, C.c_int]
        L.kfo_planar_compass.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_void_p, C.c_int]
        L.kfo_ml_estimate.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_void_p]
        L.kfo_predict_matrices.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp]
        L.kfo_inv.argtypes = [C.c_int, _dp, _dp]
        L.kfo_pinv.argtypes = [C.c_int, _dp, _dp]
        L.kfo_solve_equilibrate.argtypes = [C.c_int, _dp, _dp, _dp]
        L.kfo_topn_keep.argtypes = [C.c_int, _dp, _dp, _dp, _dp, C.c_int, _ip]
        _lib = L
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleBank:
    """T independent reference filters (KalmanFilterTOA or repaired KalmanFilterTOAIMU)."""

    def __init__(self, model, n_tags, anchors, accel_noise=0.5, jolt=0.5, ignore_worst=False,
                 cost_threshold=0.5, top_n=0, init_pos=None, n_threads=1, planar=None, ml_variant=0):
        self.model, self.T = model, n_tags
        self.anchors = _f64(anchors)
        self.A = self.anchors.shape[0]
        self.n_threads = n_threads
        ip = None
        if init_pos is not None:
            ip = _f64(init_pos)
            assert ip.shape == (n_tags, 3)
        self._h = lib().kfo_create(model, n_tags, self.A, accel_noise, jolt, int(ignore_worst),
                                   cost_threshold, top_n, int(init_pos is not None),
                                   ip.ctypes.data if ip is not None else None)
        lib().kfo_set_anchors(self._h, self.anchors, self.A)
        if ml_variant:
            lib().kfo_set_ml_variant(self._h, ml_variant)
        self.n = lib().kfo_state_dim(self._h)
        if model == MODEL_PLANAR:
            self.planar = PlanarConfig(**(planar or {}))
            lib().kfo_set_planar(self._h, C.byref(self.planar))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kfo_destroy(self._h)
            self._h = None

    def _dt(self, dt):
        d = np.atleast_1d(_f64(dt))
        assert d.size in (1, self.T)
        return d

    def step_toa(self, range_mm, err_est, dt):
        r = np.ascontiguousarray(range_mm, dtype=np.int32)
        e = _f64(err_est)
        assert r.shape == (self.T, self.A) and e.shape == (self.T, self.A)
        d = self._dt(dt)
        st = np.zeros(self.T, dtype=np.uint32)
        lib().kfo_step_toa(self._h, r, e, d, d.size, st.ctypes.data, self.n_threads)
        return st

    def step_imu(self, accel, cov, dt):
        a, c = _f64(accel), _f64(cov)
        assert a.shape == (self.T, 3) and c.shape == (self.T, 9)
        d = self._dt(dt)
        st = np.zeros(self.T, dtype=np.uint32)
        lib().kfo_step_imu(self._h, a, c, d, d.size, st.ctypes.data, self.n_threads)
        return st

    # --- KalmanFilter (MODEL_PLANAR): the other four sensors
    def step_px4flow(self, flow, dt):
        f = _f64(flow)
        assert f.shape == (self.T, 5)
        d = self._dt(dt)
        st = np.zeros(self.T, dtype=np.uint32)
        lib().kfo_planar_px4flow(self._h, f, d, d.size, st.ctypes.data, self.n_threads)
        return st

    def step_planar_imu(self, ang_vel, cov_ang_vel, lin_acc, cov_acc, dt):
        w, cw, a, ca = _f64(ang_vel), _f64(cov_ang_vel), _f64(lin_acc), _f64(cov_acc)
        assert w.shape == (self.T, 3) and cw.shape == (self.T, 9) and a.shape == (self.T, 3) and ca.shape == (self.T, 9)
        d = self._dt(dt)
        st = np.zeros(self.T, dtype=np.uint32)
        lib().kfo_planar_imu(self._h, w, cw, a, ca, d, d.size, st.ctypes.data, self.n_threads)
        return st

    def step_mag(self, mag_xyz, dt):
        m = _f64(mag_xyz)
        assert m.shape == (self.T, 3)
        d = self._dt(dt)
        st = np.zeros(self.T, dtype=np.uint32)
        lib().kfo_planar_mag(self._h, m, d, d.size, st.ctypes.data, self.n_threads)
        return st

    def step_compass(self, compass, dt):
        c = _f64(compass)
        assert c.shape == (self.T,)
        d = self._dt(dt)
        st = np.zeros(self.T, dtype=np.uint32)
        lib().kfo_planar_compass(self._h, c, d, d.size, st.ctypes.data, self.n_threads)
        return st

    def get_height(self):
        z = np.zeros(self.T)
        lib().kfo_get_height(self._h, z)
        return z

    def get_pose(self, dt_ahead=0.0):
        pos, cov, vel = np.zeros((self.T, 3)), np.zeros((self.T, 9)), np.zeros((self.T, 3))
        st = np.zeros(self.T, dtype=np.uint32)
        lib().kfo_get_pose(self._h, dt_ahead, pos, cov, vel.ctypes.data, st.ctypes.data)
        return pos, cov.reshape(self.T, 3, 3), vel, st

    def get_state(self):
        x, P = np.zeros((self.T, self.n)), np.zeros((self.T, self.n, self.n))
        lib().kfo_get_state(self._h, x, P)
        return x, P

    def set_state(self, x, P, started=True):
        lib().kfo_set_state(self._h, _f64(x), _f64(P), int(started))


def ml_estimate(anchors, ranges, err_est, seed):
    a, r, e, s = _f64(anchors), _f64(ranges), _f64(err_est), _f64(seed)
    pos, cov = np.zeros(3), np.zeros(9)
    it = lib().kfo_ml_estimate(len(r), a, r, e, s, pos, cov.ctypes.data)
    return pos, cov.reshape(3, 3), it


def predict_matrices(model, dt, accel_noise=0.5, jolt=0.5):
    n = 9 if model == MODEL_TOA_IMU else 6
    F, Q = np.zeros((n, n)), np.zeros((n, n))
    lib().kfo_predict_matrices(model, dt, accel_noise, jolt, F, Q)
    return F, Q


def inv(A):
    A = _f64(A)
    out = np.zeros_like(A)
    return out, lib().kfo_inv(A.shape[0], A, out)


def pinv(A):
    A = _f64(A)
    out = np.zeros_like(A)
    return out, lib().kfo_pinv(A.shape[0], A, out)


def solve_equilibrate(A, b):
    A, b = _f64(A), _f64(b)
    x = np.zeros_like(b)
    return x, lib().kfo_solve_equilibrate(A.shape[0], A, b, x)


def topn_keep(anchors, ranges, err_est, seed, top_n):
    keep = np.zeros(len(ranges), dtype=np.int32)
    lib().kfo_topn_keep(len(ranges), _f64(anchors), _f64(ranges), _f64(err_est), _f64(seed), top_n, keep)
    return keep


class ExtendedOracleBank:
    """The same restatement evaluated with 64 mantissa bits (oracle/kfpos_oracle_ext.cpp): the 6- and 9-state filters,
    shared dt, doubles at the boundary. Same calls as OracleBank for what a trace replay needs."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            so = os.path.join(_DIR, "libkfpos_oracle_ext.so")
            src = os.path.join(_DIR, "kfpos_oracle_ext.cpp")
            if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src),
                                                                   os.path.getmtime(os.path.join(_DIR, "kfpos_oracle.cpp"))):
                subprocess.check_call(["make", "-s", "-C", _DIR, "libkfpos_oracle_ext.so"])
            L = C.CDLL(so)
            L.kfx_create.restype = C.c_void_p
            L.kfx_create.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_double, C.c_double, C.c_int, C.c_double, C.c_int,
                                     C.c_int, C.c_void_p]
            L.kfx_destroy.argtypes = [C.c_void_p]
            L.kfx_step_toa.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _dp, C.c_double, C.c_void_p, C.c_int]
            L.kfx_step_imu.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_double, C.c_void_p, C.c_int]
            L.kfx_get_state.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
            cls._lib = L
        return cls._lib

    def __init__(self, model, n_tags, anchors, accel_noise=0.5, jolt=0.5, ignore_worst=False, cost_threshold=0.5,
                 top_n=0, init_pos=None, n_threads=1):
        assert model in (MODEL_TOA, MODEL_TOA_IMU)
        self.T, self.A, self.n, self.n_threads = n_tags, len(anchors), 9 if model == MODEL_TOA_IMU else 6, n_threads
        ip = None if init_pos is None else _f64(init_pos)
        self._h = self.lib().kfx_create(model, n_tags, self.A, _f64(anchors), accel_noise, jolt, int(ignore_worst),
                                        cost_threshold, top_n, int(ip is not None), None if ip is None else ip.ctypes.data)

    def mantissa_bits(self):
        return self.lib().kfx_mantissa_bits()

    def step_toa(self, range_mm, err_est, dt):
        st = np.zeros(self.T, dtype=np.uint32)
        self.lib().kfx_step_toa(self._h, self.T, self.A, np.ascontiguousarray(range_mm, dtype=np.int32), _f64(err_est),
                                float(dt), st.ctypes.data, self.n_threads)
        return st

    def step_imu(self, accel, cov, dt):
        st = np.zeros(self.T, dtype=np.uint32)
        self.lib().kfx_step_imu(self._h, self.T, _f64(accel), _f64(cov), float(dt), st.ctypes.data, self.n_threads)
        return st

    def get_state(self):
        x, P = np.zeros((self.T, self.n)), np.zeros((self.T, self.n, self.n))
        self.lib().kfx_get_state(self._h, self.T, x, P)
        return x, P

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib().kfx_destroy(self._h)
            self._h = None

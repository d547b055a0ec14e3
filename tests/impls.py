"""Adapters giving the oracle, the numpy oracle, the host emulation and the HIP library one face."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

import oracle_py
import numpy_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleImpl:
    def __init__(self, case, w, init, n_threads=4):
        self.b = oracle_py.OracleBank(case.model, case.T, w.anchors, ignore_worst=case.ignore_worst,
                                      top_n=case.top_n, init_pos=init, n_threads=n_threads)

    def step_toa(self, r, err, dt):
        return self.b.step_toa(r, err, dt)

    def step_imu(self, a, cov, dt):
        return self.b.step_imu(a, cov, dt)

    def fused(self, r, err, a, cov, dt):
        self.b.step_imu(a, cov, 0.0)  # newIMUMeasurement at timeLag 0, then the ranging epoch
        return self.b.step_toa(r, err, dt)

    def positions(self):
        return self.b.get_state()[0][:, :3]

    def state(self):
        return self.b.get_state()

    def pose(self, dt_ahead):
        return self.b.get_pose(dt_ahead)


class NumpyImpl:
    def __init__(self, case, w, init):
        self.T = case.T
        self.f = [numpy_oracle.NumpyFilter(case.model, w.anchors, ignore_worst=case.ignore_worst,
                                           init_pos=None if init is None else init[t]) for t in range(case.T)]

    def step_toa(self, r, err, dt):
        for t, f in enumerate(self.f):
            f.step_toa(r[t], err[t], dt)
        return np.zeros(self.T, dtype=np.uint32)

    def step_imu(self, a, cov, dt):
        for t, f in enumerate(self.f):
            f.step_imu(a[t], cov[t], dt)
        return np.zeros(self.T, dtype=np.uint32)

    def fused(self, r, err, a, cov, dt):
        self.step_imu(a, cov, 0.0)
        return self.step_toa(r, err, dt)

    def positions(self):
        return np.stack([f.pos for f in self.f])

    def state(self):
        n = self.f[0].n
        x = np.zeros((self.T, n))
        for t, f in enumerate(self.f):
            x[t, :3] = f.pos
            if n == 9:
                x[t, 3:6] = f.vel
        return x, np.stack([f.P for f in self.f])


_emu = None


def emu_lib():
    global _emu
    if _emu is None:
        L = C.CDLL(os.environ.get("KFPOS_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "libkfpos_emu.so"))
        dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        L.kfe_create.restype = C.c_void_p
        L.kfe_create.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_double, C.c_double, C.c_int, C.c_double,
                                 C.c_int, C.c_int, C.c_void_p]
        L.kfe_destroy.argtypes = [C.c_void_p]
        L.kfe_set_static.argtypes = [C.c_void_p, C.c_int]
        L.kfe_set_ml_variant.argtypes = [C.c_void_p, C.c_int]
        L.kfe_step_toa.argtypes = [C.c_void_p, ip, dp, dp, C.c_int, C.c_void_p]
        L.kfe_step_imu.argtypes = [C.c_void_p, dp, dp, dp, C.c_int, C.c_void_p]
        L.kfe_latch_imu.argtypes = [C.c_void_p, dp, dp]
        L.kfe_get_state.argtypes = [C.c_void_p, dp, dp]
        L.kfe_get_pose.argtypes = [C.c_void_p, C.c_double, dp, dp, dp]
        _emu = L
    return _emu


class EmuImpl:
    """kfpos_core.h compiled for the host: the arithmetic the HIP kernels run, lane by lane."""

    static = False  # True: the anchor-count-specialised code path (epoch in registers)

    def __init__(self, case, w, init):
        self.T, self.A, self.n = case.T, case.A, 9 if case.model == 1 else 6
        ipt = None if init is None else np.ascontiguousarray(init, dtype=np.float64)
        self._keep = ipt
        self.h = emu_lib().kfe_create(case.model, case.T, case.A, np.ascontiguousarray(w.anchors), 0.5, 0.5,
                                      int(case.ignore_worst), 0.5, case.top_n, int(init is not None),
                                      None if ipt is None else ipt.ctypes.data)
        emu_lib().kfe_set_static(self.h, int(self.static))

    def __del__(self):
        if getattr(self, "h", None):
            emu_lib().kfe_destroy(self.h)
            self.h = None

    @staticmethod
    def _f(a):
        return np.ascontiguousarray(a, dtype=np.float64)

    def step_toa(self, r, err, dt):
        st = np.zeros(self.T, dtype=np.uint32)
        emu_lib().kfe_step_toa(self.h, np.ascontiguousarray(r, dtype=np.int32), self._f(err),
                               np.array([dt], dtype=np.float64), 1, st.ctypes.data)
        return st

    def step_imu(self, a, cov, dt):
        st = np.zeros(self.T, dtype=np.uint32)
        emu_lib().kfe_step_imu(self.h, self._f(a), self._f(cov), np.array([dt], dtype=np.float64), 1,
                               st.ctypes.data)
        return st

    def fused(self, r, err, a, cov, dt):
        emu_lib().kfe_latch_imu(self.h, self._f(a), self._f(cov))
        return self.step_toa(r, err, dt)

    def state(self):
        x, P = np.zeros((self.T, self.n)), np.zeros((self.T, self.n, self.n))
        emu_lib().kfe_get_state(self.h, x, P)
        return x, P

    def positions(self):
        return self.state()[0][:, :3]

    def pose(self, dt_ahead):
        pos, cov, vel = np.zeros((self.T, 3)), np.zeros((self.T, 9)), np.zeros((self.T, 3))
        emu_lib().kfe_get_pose(self.h, dt_ahead, pos, cov, vel)
        return pos, cov.reshape(self.T, 3, 3), vel


class EmuStaticImpl(EmuImpl):
    static = True


class EmuStaticLdsImpl(EmuImpl):
    static = 2  # compile-time anchor count over the strided (LDS) scratch: the 16-anchor kernels


class GpuImpl:
    """The product path: libkfpos_hip.so through the C ABI (host-buffer entry points). coop = False keeps small plain
    6-state banks on the one-tag-per-lane kernels (KFPOS_NO_COOP=1, read at kfpos_create); coop = True lets the library
    pick the 8-lanes-per-tag kernel where it applies."""

    def __init__(self, case, w, init, storage=0, coop=False):
        from roskfpos_amd import capi
        old = os.environ.get("KFPOS_NO_COOP")
        if not coop:
            os.environ["KFPOS_NO_COOP"] = "1"
        elif old is not None:
            del os.environ["KFPOS_NO_COOP"]
        try:
            self.b = capi.KfposBank(case.model, case.T, w.anchors, storage=storage,
                                    ignore_worst=case.ignore_worst, top_n=case.top_n, init_pos=init)
        finally:
            if old is None:
                os.environ.pop("KFPOS_NO_COOP", None)
            else:
                os.environ["KFPOS_NO_COOP"] = old

    def step_toa(self, r, err, dt):
        return self.b.step_toa(r, err, dt)

    def step_imu(self, a, cov, dt):
        return self.b.step_imu(a, cov, dt)

    def fused(self, r, err, a, cov, dt):
        return self.b.step_toa_imu(r, err, a, cov, dt)

    def state(self):
        x, P, _ = self.b.get_state()
        return x, P

    def positions(self):
        return self.state()[0][:, :3]

    def pose(self, dt_ahead):
        pos, cov, vel, st = self.b.get_pose(dt_ahead)
        return pos, cov, vel, st

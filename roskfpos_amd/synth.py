"""Synthetic UWB ranging / IMU traces for the batched EKF core.

The reference has no data sets; this is the workload BASELINE.md section 3 defines:
anchors on the corners of a 10 x 10 x (0.3..3.0) m room (+8 interior anchors for the
16-anchor case), per-tag circular trajectories, ranges = true distance + N(0, 0.05^2) m
floored to integer millimetres exactly as the node does (Posgenerator.cpp:213, :484),
errorEstimation = 0.0025 m^2, accel = true acceleration + N(0, 0.1^2), covariance 0.01*I,
dt = 0.05 s with the reference's hard-coded 0.1 s first step (KalmanFilterTOA.cpp:81).

Every draw is a pure function of (seed, global tag index, step, channel) through a
counter-based SplitMix64 hash, so any shard of the tag batch regenerates its own inputs
without seeing the others (multi-GPU sharding).
"""
from __future__ import annotations

import numpy as np

SEED = 12345
DT = 0.05
DT_FIRST = 0.1
RANGE_SIGMA = 0.05
ERR_EST = 0.0025
ACC_SIGMA = 0.1
ACC_COV = 0.01
_CH_PER_STEP = np.uint64(256)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def anchors_xyz(n_anchors: int) -> np.ndarray:
    """Anchor table (A, 3) float64: 8 room corners, 8 interior anchors, then -- only for
    tests of large anchor counts, up to MAX_NUM_ANCS = 64 -- a second ring."""
    if not 1 <= n_anchors <= 64:
        raise ValueError("synthetic anchor layout is defined for 1..64 anchors")
    out = np.zeros((64, 3))
    for i in range(8):
        out[i] = (10.0 * (i & 1), 10.0 * ((i >> 1) & 1), 0.3 + 2.7 * ((i >> 2) & 1))
    for i in range(8, 16):
        out[i] = (5.0 + 3.0 * np.cos(float(i)), 5.0 + 3.0 * np.sin(float(i)), 1.5 + 0.1 * i)
    for i in range(16, 64):
        out[i] = (5.0 + 4.5 * np.cos(0.7 * i), 5.0 + 4.5 * np.sin(0.7 * i), 0.4 + 0.04 * i)
    return out[:n_anchors].copy()


def _mix(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def _tag_key(tags: np.ndarray, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        return _mix(np.uint64(seed) ^ _mix(tags.astype(np.uint64) * _GOLD + np.uint64(1)))


def _uniform(key: np.ndarray, counter) -> np.ndarray:
    """U[0,1) with 53 random bits; key (T,) broadcast against counter."""
    with np.errstate(over="ignore"):
        u = _mix(key + _GOLD * np.asarray(counter, dtype=np.uint64))
    return (u >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _normal(key: np.ndarray, counter) -> np.ndarray:
    c = np.asarray(counter, dtype=np.uint64)
    u1 = 1.0 - _uniform(key, c)  # (0, 1]
    u2 = _uniform(key, c + np.uint64(1))
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


class Workload:
    """Deterministic trace for global tags [tag0, tag0 + n_tags)."""

    def __init__(self, n_tags: int, n_anchors: int = 8, tag0: int = 0, seed: int = SEED):
        self.n_tags, self.n_anchors, self.tag0, self.seed = n_tags, n_anchors, tag0, seed
        self.anchors = anchors_xyz(n_anchors)
        tags = np.arange(tag0, tag0 + n_tags, dtype=np.uint64)
        self._key = _tag_key(tags, seed)
        self.rho = 1.0 + 3.0 * _uniform(self._key, 0)
        self.omega = 0.1 + 0.3 * _uniform(self._key, 1)
        self.phi = 2.0 * np.pi * _uniform(self._key, 2)

    # -- ground truth ------------------------------------------------------------
    @staticmethod
    def time_of(step: int) -> float:
        """Time stamp of ranging epoch `step` (step 0 happens DT_FIRST after t = 0)."""
        return DT_FIRST + DT * step

    @staticmethod
    def dt_of(step: int) -> float:
        return DT_FIRST if step == 0 else DT

    def position(self, t: float) -> np.ndarray:
        ang = self.omega * t + self.phi
        return np.stack([5.0 + self.rho * np.cos(ang), 5.0 + self.rho * np.sin(ang),
                         1.0 + 0.2 * np.sin(0.1 * t + self.phi)], axis=1)

    def acceleration(self, t: float) -> np.ndarray:
        ang = self.omega * t + self.phi
        w2 = self.omega * self.omega
        return np.stack([-self.rho * w2 * np.cos(ang), -self.rho * w2 * np.sin(ang),
                         -0.2 * 0.01 * np.sin(0.1 * t + self.phi)], axis=1)

    def init_positions(self) -> np.ndarray:
        """Fixed start = true position at t = 0 (P0 = 0), shape (T, 3)."""
        return self.position(0.0)

    # -- measurements ------------------------------------------------------------
    def ranges_mm(self, step: int) -> np.ndarray:
        """(T, A) int32 millimetres, the node's wire format (floor(), Posgenerator.cpp:213)."""
        p = self.position(self.time_of(step))
        d = np.sqrt(((p[:, None, :] - self.anchors[None, :, :]) ** 2).sum(-1))
        base = np.uint64(step + 1) * _CH_PER_STEP
        ch = base + np.uint64(2) * np.arange(self.n_anchors, dtype=np.uint64)
        noise = _normal(self._key[:, None], ch[None, :])
        return np.floor((d + RANGE_SIGMA * noise) * 1000.0).astype(np.int32)

    def err_est(self, dtype=np.float64) -> np.ndarray:
        return np.full((self.n_tags, self.n_anchors), ERR_EST, dtype=dtype)

    def accel(self, step: int, dtype=np.float64) -> np.ndarray:
        """(T, 3) accelerometer sample for epoch `step`."""
        a = self.acceleration(self.time_of(step))
        base = np.uint64(step + 1) * _CH_PER_STEP + np.uint64(200)
        ch = base + np.uint64(2) * np.arange(3, dtype=np.uint64)
        return (a + ACC_SIGMA * _normal(self._key[:, None], ch[None, :])).astype(dtype)

    def accel_cov(self, dtype=np.float64) -> np.ndarray:
        """(T, 9) row-major 3x3 accelerometer covariance."""
        return np.tile((ACC_COV * np.eye(3)).reshape(1, 9), (self.n_tags, 1)).astype(dtype)

    def trace(self, n_steps: int, step0: int = 0):
        """ranges (S, T, A) int32, accel (S, T, 3) float64, dt (S,) float64."""
        r = np.stack([self.ranges_mm(s) for s in range(step0, step0 + n_steps)])
        a = np.stack([self.accel(s) for s in range(step0, step0 + n_steps)])
        dt = np.array([self.dt_of(s) for s in range(step0, step0 + n_steps)])
        return r, a, dt

    # -- planar filter sensors (KalmanFilter: PX4Flow, IMU, magnetometer, compass) -----------------
    # The vehicle heads along its velocity: heading = trajectory angle + pi/2, yaw rate = omega. Sensor frames
    # follow KalmanFilter::px4flowOutput / imuOutput (KalmanFilter.cpp:558-576): body = R(-heading) * world.
    PX4_HEIGHT = 0.8        # <px4flow sensorHeight/>
    PX4_TIME_US = 50000.0   # integration time of one flow sample
    PX4_SIGMA = 0.02
    GYRO_SIGMA = 0.01
    MAG_SIGMA = 0.02

    def heading(self, t: float) -> np.ndarray:
        return self.omega * t + self.phi + 0.5 * np.pi

    def velocity(self, t: float) -> np.ndarray:
        ang = self.omega * t + self.phi
        return np.stack([-self.rho * self.omega * np.sin(ang), self.rho * self.omega * np.cos(ang),
                         0.02 * np.cos(0.1 * t + self.phi)], axis=1)

    def _body(self, t: float, v: np.ndarray) -> np.ndarray:
        th = self.heading(t)
        c, s = np.cos(th), np.sin(th)
        return np.stack([c * v[:, 0] + s * v[:, 1], -s * v[:, 0] + c * v[:, 1]], axis=1)

    def _noise(self, step: int, base_channel: int, n: int) -> np.ndarray:
        base = np.uint64(step + 1) * _CH_PER_STEP + np.uint64(base_channel)
        ch = base + np.uint64(2) * np.arange(n, dtype=np.uint64)
        return _normal(self._key[:, None], ch[None, :])

    def px4flow(self, step: int) -> np.ndarray:
        """(T, 5): integrationX, integrationY, integrationRotationZ, integrationTime [us], quality -- the
        arguments of newPX4FlowMeasurement (KalmanFilter.cpp:102). Every 16th tag reports quality 0."""
        t = self.time_of(step)
        vb = self._body(t, self.velocity(t)) + self.PX4_SIGMA * self._noise(step, 210, 2)
        sec = self.PX4_TIME_US / 1e6
        gz = self.omega + self.GYRO_SIGMA * self._noise(step, 216, 1)[:, 0]
        q = np.where((np.arange(self.tag0, self.tag0 + self.n_tags) + step) % 16 == 5, 0.0, 200.0)
        return np.stack([vb[:, 0] * sec / self.PX4_HEIGHT, vb[:, 1] * sec / self.PX4_HEIGHT, gz * sec,
                         np.full(self.n_tags, self.PX4_TIME_US), q], axis=1)

    def planar_imu(self, step: int):
        """angular velocity (T, 3) and body-frame linear acceleration (T, 3) for newIMUMeasurement."""
        t = self.time_of(step)
        ab = self._body(t, self.acceleration(t)) + ACC_SIGMA * self._noise(step, 220, 2)
        w = np.zeros((self.n_tags, 3))
        w[:, 2] = self.omega + self.GYRO_SIGMA * self._noise(step, 226, 1)[:, 0]
        a = np.concatenate([ab, np.full((self.n_tags, 1), 9.81)], axis=1)
        return w, a

    def mag(self, step: int) -> np.ndarray:
        """(T, 3) magnetometer vector whose atan2(y, x) is the heading."""
        th = self.heading(self.time_of(step)) + self.MAG_SIGMA * self._noise(step, 230, 1)[:, 0]
        return np.stack([np.cos(th), np.sin(th), np.full(self.n_tags, -0.4)], axis=1)

    def compass(self, step: int) -> np.ndarray:
        """(T,) compass heading in radians, unwrapped on purpose (newCompassMeasurement normalises once)."""
        return self.heading(self.time_of(step)) + self.MAG_SIGMA * self._noise(step, 234, 1)[:, 0]

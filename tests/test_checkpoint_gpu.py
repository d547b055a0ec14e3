"""Checkpoint / restore through the C ABI: get_state + flags + get_latch of a running bank, restored into a fresh
handle, continue bit-identically (9-state filter with its latched IMU sample, planar filter with all latches)."""
import numpy as np
import pytest

from planar import CFG, PlanarGpu, run_trace
from roskfpos_amd.synth import Workload

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("storage", [0, 1, 3])
def test_nine_state_checkpoint_carries_the_latched_imu_sample(storage):
    """(every storage mode: what get_state hands out is what the kernels keep -- 24 or 48 bits of each covariance entry
    in the compact modes -- so a restored bank continues bit for bit there too)"""
    from roskfpos_amd import capi
    T, S = 100, 30
    w = Workload(T, 8)
    real = np.float64 if storage == 0 else np.float32
    cov = w.accel_cov(real)
    cov[:, 1] = cov[:, 3] = 0.002  # a full covariance: all six stored entries matter

    def epoch(b, s):
        if s % 3 == 0:
            return b.step_toa_imu(w.ranges_mm(s), w.err_est(real), w.accel(s, real), cov, w.dt_of(s))
        return b.step_toa(w.ranges_mm(s), w.err_est(real), w.dt_of(s))  # re-fuses the latched sample

    a = capi.KfposBank(capi.MODEL_TOA_IMU, T, w.anchors, storage=storage, init_pos=w.init_positions())
    for s in range(S):
        epoch(a, s)
    x, P, fl = a.get_state()
    latch = a.get_latch()
    assert latch.shape == (T, 12) and np.all(fl & 2)
    np.testing.assert_array_equal(latch[:, 3:].reshape(T, 3, 3), cov.reshape(T, 3, 3).astype(np.float64))
    b = capi.KfposBank(capi.MODEL_TOA_IMU, T, w.anchors, storage=storage, init_pos=w.init_positions())
    b.set_state(x, P, fl)
    b.set_latch(latch)
    for s in range(S, S + 10):
        sa, sb = epoch(a, s), epoch(b, s)
        np.testing.assert_array_equal(sa, sb)
    xa, Pa, _ = a.get_state()
    xb, Pb, _ = b.get_state()
    np.testing.assert_array_equal(xa, xb)
    np.testing.assert_array_equal(Pa, Pb)
    # models without latches: nothing to save
    c = capi.KfposBank(capi.MODEL_TOA, 8, w.anchors, init_pos=w.init_positions()[:8])
    assert c.get_latch().shape == (8, 0)


def test_planar_checkpoint_carries_all_latches():
    T = 90
    w = Workload(T, 8)
    a = PlanarGpu(w, CFG, w.init_positions())
    run_trace([a], w, 20, ("imu", "px4", "mag", "compass"))
    x, P, fl = a.b.get_state()
    latch, z = a.b.get_latch(), a.get_height()
    assert latch.shape == (T, 15) and np.all((fl >> 4) & 0xE)
    b = PlanarGpu(w, CFG, w.init_positions())
    b.b.set_state(x, P, fl)
    b.b.set_latch(latch)
    np.testing.assert_array_equal(b.get_height(), z)
    for s in range(20, 30):  # ranging epochs carry the restored latches; a compass call uses PX4Flow + IMU ones
        r, dt = w.ranges_mm(s), w.dt_of(s)
        np.testing.assert_array_equal(a.step_compass(w.compass(s), 0.01), b.step_compass(w.compass(s), 0.01))
        np.testing.assert_array_equal(a.step_toa(r, w.err_est(), dt - 0.01), b.step_toa(r, w.err_est(), dt - 0.01))
    xa, Pa = a.get_state()
    xb, Pb = b.get_state()
    np.testing.assert_array_equal(xa, xb)
    np.testing.assert_array_equal(Pa, Pb)


def test_planar_checkpoint_with_estimated_height_and_ml_initialisation():
    """useFixedHeight = 0 and no start position: the 3-D ML initialisation sets each tag's working height
    (KalmanFilter.cpp:252-258), which is part of the checkpoint (kfpos_get_height / kfpos_set_height)."""
    T = 70
    w = Workload(T, 8)
    cfg = dict(CFG, use_fixed_height=0)
    a = PlanarGpu(w, cfg, None)
    run_trace([a], w, 12, ("imu", "compass"))
    x, P, fl = a.b.get_state()
    latch, z = a.b.get_latch(), a.get_height()
    assert np.ptp(z) > 0.01                      # estimated heights, not the configured constant
    b = PlanarGpu(w, cfg, None)
    b.b.set_state(x, P, fl)
    b.b.set_latch(latch)
    assert not np.array_equal(b.get_height(), z)  # a fresh handle starts at the configured height ...
    b.b.set_height(z)                             # ... the checkpoint carries the estimated one
    np.testing.assert_array_equal(b.get_height(), z)
    for s in range(12, 22):
        r, dt = w.ranges_mm(s), w.dt_of(s)
        np.testing.assert_array_equal(a.step_toa(r, w.err_est(), dt), b.step_toa(r, w.err_est(), dt))
    xa, Pa = a.get_state()
    xb, Pb = b.get_state()
    np.testing.assert_array_equal(xa, xb)
    np.testing.assert_array_equal(Pa, Pb)


def test_ml_estimator_checkpoint_carries_the_seed():
    from roskfpos_amd import capi
    T = 40
    w = Workload(T, 8)
    seeds = w.init_positions() + 0.3
    a = capi.KfposBank(capi.MODEL_ML, T, w.anchors, init_pos=seeds)
    for s in range(3):
        a.step_toa(w.ranges_mm(s), w.err_est(), 0.05)
    x, P, fl = a.get_state()
    seed = a.get_latch()
    np.testing.assert_array_equal(seed, seeds)
    b = capi.KfposBank(capi.MODEL_ML, T, w.anchors, init_pos=np.zeros((T, 3)) + 1.0)   # other seeds
    b.set_state(x, P, fl)
    b.set_latch(seed)
    for s in range(3, 6):
        np.testing.assert_array_equal(a.step_toa(w.ranges_mm(s), w.err_est(), 0.05),
                                      b.step_toa(w.ranges_mm(s), w.err_est(), 0.05))
    np.testing.assert_array_equal(a.get_state()[0], b.get_state()[0])

"""Checkpoint / restore through the C ABI: get_state + flags + get_latch of a running bank, restored into a fresh
handle, continue bit-identically (9-state filter with its latched IMU sample, planar filter with all latches)."""
import numpy as np
import pytest

from planar import CFG, PlanarGpu, run_trace
from roskfpos_amd.synth import Workload

pytestmark = pytest.mark.gpu


def test_nine_state_checkpoint_carries_the_latched_imu_sample():
    from roskfpos_amd import capi
    T, S = 100, 30
    w = Workload(T, 8)
    cov = w.accel_cov()
    cov[:, 1] = cov[:, 3] = 0.002  # a full covariance: all six stored entries matter

    def epoch(b, s):
        if s % 3 == 0:
            return b.step_toa_imu(w.ranges_mm(s), w.err_est(), w.accel(s), cov, w.dt_of(s))
        return b.step_toa(w.ranges_mm(s), w.err_est(), w.dt_of(s))  # re-fuses the latched sample

    a = capi.KfposBank(capi.MODEL_TOA_IMU, T, w.anchors, init_pos=w.init_positions())
    for s in range(S):
        epoch(a, s)
    x, P, fl = a.get_state()
    latch = a.get_latch()
    assert latch.shape == (T, 12) and np.all(fl & 2)
    np.testing.assert_array_equal(latch[:, 3:].reshape(T, 3, 3), cov.reshape(T, 3, 3))
    b = capi.KfposBank(capi.MODEL_TOA_IMU, T, w.anchors, init_pos=w.init_positions())
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

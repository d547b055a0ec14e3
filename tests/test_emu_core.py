"""CPU check of the kernel arithmetic (roskfpos_amd/csrc/kfpos_core.h compiled for the host).

The HIP kernels evaluate the reference's iterated EKF in a restructured form (no m x m inverse, no
pinv; see the header of kfpos_core.h). These tests run that exact text lane by lane on the CPU and
compare it with the dense oracle, so the algebra is validated in the GPU-less container; the GPU
tests then only have to show that the device build of the same text agrees.
"""
import numpy as np
import pytest

from cases import CASES, CASE_BY_NAME, drive, rms_and_max
from impls import EmuImpl, EmuStaticImpl, EmuStaticLdsImpl, OracleImpl

# Bar from BASELINE.json: <= 1e-6 m RMS. Everything except the one documented case sits near 1e-12.
TOL_DEFAULT_RMS, TOL_DEFAULT_MAX = 1e-9, 1e-8


@pytest.mark.parametrize("impl", [EmuImpl, EmuStaticImpl, EmuStaticLdsImpl], ids=["generic", "static", "static-lds"])
@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_kernel_math_matches_oracle(case, impl):
    """generic = run-time anchor count, epoch staged per lane (LDS on the GPU); static = anchor count fixed at
    compile time (4/8/16), epoch in registers, every anchor loop unrolled."""
    fo, po, so = drive(case, OracleImpl, record=True)
    fe, pe, se = drive(case, impl, record=True)
    rms, mx, same_nan = rms_and_max(pe, po)
    assert same_nan
    # also for the 6-state ML initialisation, whose column slip (KalmanFilterTOA.cpp:102-104) leaves a rank-deficient
    # NON-symmetric P: those epochs take the SVD pseudo-inverse route for delta' pinv(P) delta (kfpos_core.h: Pinv6)
    assert rms <= TOL_DEFAULT_RMS and mx <= TOL_DEFAULT_MAX, (rms, mx)
    assert np.array_equal(so, se)  # same iteration counts, same flags, same ignored anchor
    xo, Po = fo.state()
    xe, Pe = fe.state()
    ok = np.isfinite(xo[:, 0])
    scale = np.abs(Po[ok]).max()
    assert np.abs(Po[ok] - Pe[ok]).max() <= 1e-9 * scale
    if case.model == 1:
        assert np.abs(xo[ok, 3:6] - xe[ok, 3:6]).max() < 1e-8  # velocity is persisted by the 9-state filter


@pytest.mark.parametrize("name", ["toa6_A8_fixed", "toa6_A8_mlinit", "imu9_A8_fixed"])
def test_get_pose_extrapolation(name):
    case = CASE_BY_NAME[name]
    fo = drive(case, OracleImpl, steps=25)
    fe = drive(case, EmuImpl, steps=25)
    for dt_ahead in (0.0, 0.05, 0.37):
        po, co, vo, _ = fo.pose(dt_ahead)
        pe, ce, ve = fe.pose(dt_ahead)
        assert np.allclose(pe, po, atol=1e-8)
        assert np.allclose(ce, co, rtol=1e-6, atol=1e-12)
        if case.model == 1:
            assert np.allclose(ve, vo, atol=1e-8)


def test_iteration_counts_are_in_the_surveyed_range():
    # ~2.6 ML solves and ~2 gain iterations per step at 8 anchors (6-state)
    case = CASE_BY_NAME["toa6_A8_fixed"]
    _, _, st = drive(case, EmuImpl, record=True)
    gain = (st >> 8) & 0xFF
    ml = (st >> 16) & 0xFF
    assert 1.5 < gain[10:].mean() < 3.5 and 1.5 < ml[10:].mean() < 4.5


@pytest.mark.parametrize("model,bar", [(1, 5e-9), (0, 1e-10)])
def test_p48_covariance_storage_keeps_the_bar_on_the_baseline_trace(model, bar):
    """KFPOS_STORE_P48 (the covariance kept in 6 bytes per entry between epochs: 40 significant bits, kfpos_p48.h) on the BASELINE-style
    trace, kernel body on the host: the 9-state filter, which turns a 24-bit covariance into 1.6e-6 m, stays below
    5e-9 m RMS (tests/cov_encoding_study.py: 5.5e-10 m at 2 048 tags). The GPU legs are in test_gpu_parity.py."""
    import ctypes as C
    from cases import Case
    from impls import emu_lib, EmuStaticImpl
    from roskfpos_amd.synth import Workload
    T, S = 256, 100
    case = Case("baseline", model, 8, T=T, S=S)
    w = Workload(T, 8)
    err, cov = w.err_est(np.float32).astype(np.float64), w.accel_cov(np.float32).astype(np.float64)
    lib = emu_lib()
    lib.kfe_round_storage.argtypes = [C.c_void_p, C.c_int]
    orc, emu = OracleImpl(case, w, w.init_positions()), EmuStaticImpl(case, w, w.init_positions())
    sq = 0.0
    for s in range(S):
        a = w.accel(s, np.float32).astype(np.float64)
        for f in (orc, emu):
            if model == 1:
                f.fused(w.ranges_mm(s), err, a, cov, w.dt_of(s))
            else:
                f.step_toa(w.ranges_mm(s), err, w.dt_of(s))
        lib.kfe_round_storage(emu.h, 8)
        sq += ((emu.positions() - orc.positions()) ** 2).sum()
    assert np.sqrt(sq / (T * S)) <= bar

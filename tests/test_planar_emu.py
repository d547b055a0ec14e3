"""Host emulation of the planar filter's kernel body (kfpos_core.h: step_planar8) against the oracle."""
import numpy as np
import pytest

from planar import CFG, PlanarEmu, PlanarOracle, run_trace
from roskfpos_amd.synth import Workload

ALL = ("imu", "px4", "mag", "compass")


def _compare(w, cfg, init, sensors, S, emu_kwargs, atol_x=1e-9):
    orc = PlanarOracle(w, cfg, init)
    emu = PlanarEmu(w, cfg, init, **emu_kwargs)
    st_o, st_e = run_trace([orc, emu], w, S, sensors)
    xo, Po = orc.get_state()
    xe, Pe = emu.get_state()
    assert np.all(np.isfinite(xo))
    np.testing.assert_allclose(xe, xo, rtol=0, atol=atol_x)
    np.testing.assert_allclose(Pe, Po, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(emu.get_height(), orc.get_height(), rtol=0, atol=1e-9)
    # identical status words: flags, gain iterations, ML iterations
    same = sum(int(np.array_equal(a, b)) for a, b in zip(st_o, st_e))
    assert same == len(st_o), [(k, np.flatnonzero(a != b)[:4], a[a != b][:4], b[a != b][:4])
                               for k, (a, b) in enumerate(zip(st_o, st_e)) if not np.array_equal(a, b)][:3]
    return orc, emu


@pytest.mark.parametrize("fixed_init", [True, False])
@pytest.mark.parametrize("fixed_height", [1, 0])
@pytest.mark.parametrize("variant", ["fast", "sensors", "static"])
def test_ranging_only(fixed_init, fixed_height, variant):
    w = Workload(40, 8)
    cfg = dict(CFG, use_fixed_height=fixed_height)
    kw = dict(sensors=(variant == "sensors"), static=(variant == "static"))
    _compare(w, cfg, w.init_positions() if fixed_init else None, (), 60, kw)


@pytest.mark.parametrize("sensors", [("imu",), ("px4",), ("mag",), ("compass",), ALL])
def test_sensor_rows(sensors):
    w = Workload(32, 8)
    _compare(w, CFG, w.init_positions(), sensors, 60, dict(sensors=True))


def test_all_sensors_ml_init_free_height():
    w = Workload(24, 8)
    _compare(w, dict(CFG, use_fixed_height=0), None, ALL, 50, dict(sensors=True))


@pytest.mark.parametrize("A", [5, 12])
def test_other_anchor_counts(A):
    w = Workload(24, A)
    _compare(w, CFG, w.init_positions(), ALL, 40, dict(sensors=True))


def test_pose_matches_oracle():
    w = Workload(16, 8)
    orc, emu = _compare(w, CFG, w.init_positions(), ("imu", "mag"), 20, dict(sensors=True))
    po, co, vo, _ = orc.get_pose(0.03)
    pe, ce, ve, _ = emu.get_pose(0.03)
    np.testing.assert_allclose(pe, po, atol=1e-10)
    np.testing.assert_allclose(ve, vo, atol=1e-10)
    np.testing.assert_allclose(ce, co, rtol=1e-7, atol=1e-13)


def test_zero_variance_is_reported_not_propagated():
    """covarianceMag = 0 (the XML default) makes inv(observationCovariance) throw in the reference."""
    w = Workload(8, 8)
    cfg = dict(CFG, mag_cov=0.0)
    orc, emu = PlanarOracle(w, cfg, w.init_positions()), PlanarEmu(w, cfg, w.init_positions())
    st_o, st_e = run_trace([orc, emu], w, 8, ("mag",), edge=False)
    for a, b in zip(st_o, st_e):
        np.testing.assert_array_equal(a, b)
    assert any(np.all(a == 1) for a in st_o)  # ST_UPDATE_SKIPPED
    np.testing.assert_allclose(emu.get_state()[0], orc.get_state()[0], atol=1e-9)

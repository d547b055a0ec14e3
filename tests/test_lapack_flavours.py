"""The C++ oracle against the numpy restatement on the LAPACK drivers Armadillo itself calls -- in each of the three
generations of Armadillo's dispatch (oracle/numpy_oracle.py: FLAVOUR "lapack" / "old" / "new"). The reference pins no
Armadillo version (CMakeLists.txt:29) and ships no fixture, so nothing can say which of the three its authors ran;
what can be shown is that the choice does not matter at the 1e-6 m bar: the oracle (hand-written LU and Jacobi SVD)
stays within the bounds below of every one of them, on the committed parity cases, on BASELINE-style traces of 100
epochs, and on the planar filter with all its sensors. tools/lapack_gap.py prints the same numbers as a table
(profiles/r03_oracle_vs_lapack_gap.json)."""
import numpy as np
import pytest

import numpy_oracle as N
from cases import CASE_BY_NAME, Case, drive, rms_and_max
from impls import NumpyImpl, OracleImpl

CASES = ["toa6_A8_fixed", "toa6_A4_fixed", "toa6_A8_mlinit", "toa6_A8_ignoreworst", "toa6_A8_ignoreworst_far",
         "imu9_A8_fixed", "imu9_A8_mlinit", "imu9_A8_latched", "imu9_A8_separate"]


@pytest.fixture(params=N.FLAVOURS)
def flavour(request):
    old = N.FLAVOUR
    N.set_flavour(request.param)
    yield request.param
    N.set_flavour(old)


def small(c, T=4, S=30):
    return Case(c.name, c.model, c.A, fixed=c.fixed, ignore_worst=c.ignore_worst, top_n=c.top_n, outlier=c.outlier,
                T=T, S=S, imu_every=c.imu_every, separate_imu=c.separate_imu, cov_full=c.cov_full)


def gap(case):
    fo, po, _ = drive(case, OracleImpl, record=True)
    fn, pn, _ = drive(case, NumpyImpl, record=True)
    rms, mx, same = rms_and_max(pn, po)
    xo, Po = fo.state()
    xn, Pn = fn.state()
    ok = np.isfinite(Po).all(axis=(1, 2))
    rel_P = float(np.abs(Po[ok] - Pn[ok]).max() / np.abs(Po[ok]).max()) if ok.any() else 0.0
    return rms, mx, same, rel_P


@pytest.mark.parametrize("name", CASES)
def test_parity_cases_in_every_flavour(name, flavour):
    rms, mx, same, rel_P = gap(small(CASE_BY_NAME[name]))
    assert same
    # measured: <= 3e-13 m (6-state), <= 2e-11 m (9-state) in every flavour; the bar is 1e-6 m
    assert mx < 1e-9, (flavour, rms, mx)
    assert rel_P < 1e-6


@pytest.mark.parametrize("model", [0, 1])
def test_baseline_style_trace_100_epochs(model, flavour):
    """the workload of bench.py / BASELINE configs[1] and [2] in small: clean trace, fixed start, 100 epochs"""
    c = Case("baseline", model, 8, T=6, S=100)
    c.epoch = lambda w, s: w.ranges_mm(s)   # no drop-outs: exactly the bench trace
    rms, mx, same, rel_P = gap(c)
    assert same and mx < (1e-11 if model == 0 else 1e-9), (flavour, rms, mx)


def test_planar_all_sensors(flavour):
    from test_planar_oracle import CFG, _run
    x, P, xr, Pr, orc, ref = _run(4, 8, 30, CFG, True, ("imu", "px4", "mag", "compass"))
    np.testing.assert_allclose(x, xr, rtol=0, atol=1e-8)
    x, P, xr, Pr, orc, ref = _run(4, 8, 25, dict(CFG, use_fixed_height=0), False, ())
    np.testing.assert_allclose(x, xr, rtol=0, atol=1e-9)


def test_drivers_are_the_ones_named():
    """the restatement must go through scipy.linalg.lapack, not numpy.linalg's gesv-based inv / solve"""
    src = open(N.__file__).read()
    for driver in ("dgetrf", "dgetri", "dgesdd", "dgesvx", "dgelsd", "dgecon", "dpotrf", "dposvx", "dsyevd"):
        assert "_la." + driver in src, driver
    assert "np.linalg.inv" not in src and "np.linalg.solve" not in src and "np.linalg.svd" not in src

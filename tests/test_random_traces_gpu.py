"""Randomised shapes on the GPU: the library (through the C ABI) against the SAME kernel arithmetic compiled for the host
(tests/emu) on configurations the fixed cases do not enumerate -- anchor counts 1..64 (every launch specialisation:
registers for 8, the LDS-resident epoch for 16, the generic path for everything else), batches of 1..300 tags (partly
filled and lone wavefronts, the 8-lanes-per-tag kernel for the plain 6-state filter), ML or fixed initialisation, the
leave-one-out and top-N heuristics, ragged epochs, errorEstimations over three decades, dt from 10 ms to 2 s.
tests/test_random_traces.py ties that host build to the oracle (hypothesis, CPU); this file ties the GPU to the host
build, so launch selection, staging and code generation are covered on shapes nobody picked by hand.

What is compared: the status word of every tag-epoch (flags, IEKF and Gauss-Newton pass counts) and the state after the
last epoch, for the tags that stayed in the regime where two correct implementations agree (test_random_traces._tame);
the GPU uses hardware rcp / rsq seeds and contracts a*b+c where the host build may not, so "equal" is <= 1e-9 m -- or,
where the problem itself is ill-conditioned (a single range per epoch on a covariance that starts at zero: the oracle in
double and in extended precision are 3e-9 m apart there), ten times the host build's own distance to the oracle."""
import numpy as np
import pytest

from conftest import has_gpu
import oracle_py
from test_random_traces import _Emu, _tame, _trace

pytestmark = pytest.mark.gpu

A_CHOICES = [1, 2, 3, 4, 5, 6, 7, 8, 8, 8, 9, 11, 12, 16, 16, 17, 24, 33, 48, 64]
T_CHOICES = [1, 2, 7, 63, 64, 65, 100, 129, 300]


def _cases(n=60):
    rng = np.random.default_rng(20261004)
    out = []
    for i in range(n):
        model = int(rng.integers(0, 2))
        heuristic = "none" if model == 1 else str(rng.choice(["none", "none", "ignore_worst", "top1", "top2", "top3"]))
        out.append((i, model, int(rng.choice(A_CHOICES)), int(rng.choice(T_CHOICES)), bool(rng.integers(0, 2)), heuristic))
    return out


@pytest.mark.parametrize("i,model,A,T,fixed,heuristic", _cases(),
                         ids=[f"{i}-m{m}-A{a}-T{t}-{'fix' if f else 'ml'}-{h}" for i, m, a, t, f, h in _cases()])
def test_gpu_equals_host_build_of_the_kernel_body(i, model, A, T, fixed, heuristic):
    if not has_gpu():
        pytest.skip("no GPU")
    from roskfpos_amd import capi
    rng = np.random.default_rng(1000 + i)
    S = 6
    anchors, p0, trace = _trace(rng, T, A, S, outliers=(heuristic != "none"))
    iw = heuristic == "ignore_worst"
    tn = int(heuristic[3]) if heuristic.startswith("top") else 0
    init = p0 if fixed else None
    gpu = capi.KfposBank(model, T, anchors, ignore_worst=iw, top_n=tn, init_pos=init)
    emu = _Emu(model, T, anchors, init, iw, tn)
    orc = oracle_py.OracleBank(model, T, anchors, ignore_worst=iw, top_n=tn, init_pos=init)
    tame = np.ones(T, dtype=bool)
    for mm, err, dt, acc, cov in trace:
        if model == 1:
            sg, se = gpu.step_imu(acc, cov, 0.0), emu.step_imu(acc, cov, 0.0)
            orc.step_imu(acc, cov, 0.0)
            assert np.array_equal(sg, se)
        sg, se = gpu.step_toa(mm, err, dt), emu.step_toa(mm, err, dt)
        orc.step_toa(mm, err, dt)
        tame &= _tame(se, 20 if model == 1 else 10)
        assert np.array_equal(sg[tame], se[tame]), (np.flatnonzero((sg != se) & tame)[:8], sg[tame][:4], se[tame][:4])
        assert np.array_equal((sg & 3), (se & 3))                       # skipped / too few ranges: never regime-dependent
    xg, Pg, _ = gpu.get_state()
    xe, Pe = emu.state()
    fin = np.isfinite(xe).all(1)
    assert np.array_equal(fin[tame], np.isfinite(xg).all(1)[tame])
    ok = fin & tame
    if ok.any():
        xo = orc.get_state()[0]
        d = np.abs(xg[ok] - xe[ok]).max(1)
        conditioning = np.nan_to_num(np.abs(xe[ok] - xo[ok]).max(1), nan=np.inf)
        assert np.all(d <= np.maximum(1e-9, 10.0 * conditioning)), (d.max(), conditioning.max())
        assert np.median(d) <= 1e-11
        scale = np.abs(Pe[ok]).max((1, 2))[:, None, None] + 1e-300
        assert (np.abs(Pg[ok] - Pe[ok]) / scale).max() <= 1e-6
    gpu.close()

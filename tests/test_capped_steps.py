"""The 9-state steps that run into the iteration cap -- 9.5 % of the tag-epochs of the bench workload: H_imu = diag(a)
(KalmanFilterTOAIMU.cpp:449-471) makes the reference's gain iteration crawl for them, so it stops at maxSteps = 20
(:293) instead of at its tolerance. tests/test_random_traces.py leaves such steps out of its comparison (a capped step
on a *diverging* iteration amplifies the last bit by 1e5 and more); here they are the subject:

  (1) BASELINE-style traces (20 Hz, fixed start at the truth, the statistics of SURVEY.md 8d), several seeds and tag
      ranges: every status word identical to the oracle's in every epoch, and the positions of the tags that are capped
      IN that epoch within 1e-7 m of the oracle's (the 2 000-epoch soak of round 2 saw 6e-8 m at most) -- on the host
      build of the kernel body here, on the GPU in the -m gpu leg;
  (2) rough traces -- dt up to 0.25 s right after a fixed start with P0 = 0, dropouts, three noise levels -- where a
      capped step can sit on a DIVERGING iteration: there the oracle itself moves by metres when its inputs are nudged by
      one part in 1e12, and the kernel is held to 1000 x that sensitivity (floored at 1e-7 m), capped steps INCLUDED.
"""
import numpy as np
import pytest

import oracle_py
from conftest import has_gpu
from roskfpos_amd.synth import Workload, anchors_xyz
from test_random_traces import _Emu

CAP = 20


def _baseline_run(make_kernel, T, S, seed, tag0, storage_real=np.float64):
    w = Workload(T, 8, tag0=tag0, seed=seed)
    err = w.err_est(storage_real).astype(np.float64)
    cov = w.accel_cov(storage_real).astype(np.float64)
    orc = oracle_py.OracleBank(1, T, w.anchors, init_pos=w.init_positions(), n_threads=8)
    ker = make_kernel(w)
    worst_capped = worst_other = 0.0
    n_capped = 0
    sq = 0.0
    for s in range(S):
        r, dt = w.ranges_mm(s), w.dt_of(s)
        a = w.accel(s, storage_real).astype(np.float64)
        orc.step_imu(a, cov, 0.0)
        so = orc.step_toa(r, err, dt)
        sk, xk = ker(r, err, a, cov, dt)
        assert np.array_equal(so, sk), f"status words, epoch {s}"
        xo, _ = orc.get_state()
        d = np.abs(xo[:, :3] - xk[:, :3]).max(1)
        capped = ((so >> 8) & 0xFF) == CAP
        n_capped += int(capped.sum())
        worst_capped = max(worst_capped, float(d[capped].max(initial=0.0)))
        worst_other = max(worst_other, float(d[~capped].max(initial=0.0)))
        sq += float((d ** 2).sum())
    return n_capped / (T * S), worst_capped, worst_other, np.sqrt(sq / (T * S))


def _emu_kernel(w):
    emu = _Emu(1, w.n_tags, w.anchors, w.init_positions(), False, 0)

    def step(r, err, a, cov, dt):
        emu.step_imu(a, cov, 0.0)
        st = emu.step_toa(r, err, dt)
        return st, emu.state()[0]
    return step


@pytest.mark.parametrize("seed,tag0", [(12345, 0), (12345, 60000), (777, 0), (4242, 123456)])
def test_capped_steps_on_baseline_traces_kernel_body(seed, tag0):
    frac, worst_capped, worst_other, rms = _baseline_run(_emu_kernel, 512, 100, seed, tag0)
    assert 0.03 < frac < 0.2                     # the cap is hit as often as on the bench workload (9.5 %)
    assert worst_capped <= 1e-7 and worst_other <= 1e-7 and rms <= 1e-9, (worst_capped, worst_other, rms)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,tag0,storage", [(12345, 0, 0), (12345, 60000, 2), (777, 0, 2), (4242, 123456, 0)])
def test_capped_steps_on_baseline_traces_gpu(seed, tag0, storage):
    if not has_gpu():
        pytest.skip("no GPU")
    from roskfpos_amd import capi
    real = np.float32 if storage else np.float64

    def gpu_kernel(w):
        bank = capi.KfposBank(capi.MODEL_TOA_IMU, w.n_tags, w.anchors, storage=storage, init_pos=w.init_positions())

        def step(r, err, a, cov, dt):
            st = bank.step_toa_imu(r, err.astype(real), a.astype(real), cov.astype(real), dt)
            return st, bank.get_state()[0]
        return step

    frac, worst_capped, worst_other, rms = _baseline_run(gpu_kernel, 4096, 120, seed, tag0, storage_real=real)
    assert 0.03 < frac < 0.2
    assert worst_capped <= 1e-7 and worst_other <= 1e-7 and rms <= 1e-9, (worst_capped, worst_other, rms)


# ---------------------------------------------------------------------------------------------- rough traces
def rough_trace(seed, T, A, S):
    rng = np.random.default_rng(seed)
    anchors = anchors_xyz(A)
    sig_r, sig_a = float(rng.choice([0.02, 0.05, 0.1])), float(rng.choice([0.05, 0.1, 0.3]))
    rho, om, ph = rng.uniform(1, 4, T), rng.uniform(0.1, 0.6, T), rng.uniform(0, 2 * np.pi, T)

    def pos(t):
        return np.stack([5 + rho * np.cos(om * t + ph), 5 + rho * np.sin(om * t + ph), 1 + 0.2 * np.sin(0.1 * t + ph)], 1)

    def acc(t):
        return np.stack([-rho * om * om * np.cos(om * t + ph), -rho * om * om * np.sin(om * t + ph),
                         -0.002 * np.sin(0.1 * t + ph)], 1)

    cov = np.tile((float(rng.choice([0.003, 0.01, 0.05])) * np.eye(3)).ravel(), (T, 1))
    t, out = 0.0, []
    for s in range(S):
        dt = float(rng.choice([0.02, 0.05, 0.05, 0.05, 0.1, 0.25])) if s else 0.1
        t += dt
        d = np.sqrt(((pos(t)[:, None, :] - anchors[None]) ** 2).sum(-1))
        mm = np.floor((d + rng.normal(0, sig_r, d.shape)) * 1000).astype(np.int32)
        mm[rng.random(mm.shape) < 0.08] = 0
        out.append((mm, np.full(mm.shape, sig_r ** 2), dt, acc(t) + rng.normal(0, sig_a, (T, 3)), cov))
    return anchors, pos(0.0), out


PERTURB = 1e-12


def _rough_run(make_kernel, seed, A, T=64, S=40):
    """Returns (capped steps, tags the trace drove into the chaotic regime)."""
    anchors, init, trace = rough_trace(seed, T, A, S)
    rng = np.random.default_rng(1000 + seed)
    orc = oracle_py.OracleBank(1, T, anchors, init_pos=init, n_threads=8)
    nudged = oracle_py.OracleBank(1, T, anchors, init_pos=init, n_threads=8)   # the ORACLE on inputs moved by 1e-12
    ker = make_kernel(anchors, init)
    n_capped = 0
    run_kernel, run_sens = np.zeros(T), np.zeros(T)   # largest difference so far, per tag
    for mm, err, dt, a, cov in trace:
        orc.step_imu(a, cov, 0.0)
        so = orc.step_toa(mm, err, dt)
        xk = ker(mm, err, a, cov, dt)
        nudged.step_imu(a * (1 + PERTURB * rng.choice([-1.0, 1.0], a.shape)), cov, 0.0)
        nudged.step_toa(mm, err * (1 + PERTURB * rng.choice([-1.0, 1.0], err.shape)), dt * (1 + PERTURB))
        xo, _ = orc.get_state()
        xs, _ = nudged.get_state()
        assert np.isfinite(xo).all() and np.isfinite(xk).all()
        run_kernel = np.maximum(run_kernel, np.abs(xk[:, :3] - xo[:, :3]).max(1))
        run_sens = np.maximum(run_sens, np.abs(xs[:, :3] - xo[:, :3]).max(1))
        n_capped += int((((so >> 8) & 0xFF) == CAP).sum())
    # No tag and no step is left out. A capped step on a diverging iteration multiplies ANY difference by ~2.5 per trip,
    # 1e5 per step (measured, iterate by iterate, against the dense form): on these traces most tags end METRES away
    # from the oracle -- and so does the oracle itself when its real-valued inputs (errorEstimation, acceleration, dt)
    # are moved by one part in 1e12. The reference's result is not a function of its input at double precision there,
    # so the kernel is held to what is decidable: it may differ from the oracle only where the oracle differs from
    # itself, and by no more than 1000 x what that 1e-12 nudge does (i.e. it behaves like the reference on inputs
    # perturbed by <= 1e-9 relative). Tags the nudge leaves alone (< 1e-9 m) must agree to 1e-7 m.
    assert np.all(run_kernel <= np.maximum(1e-7, 1000.0 * run_sens)), (run_kernel.max(), run_sens.max())
    tame = run_sens < 1e-9
    assert run_kernel[tame].max(initial=0.0) <= 1e-7
    return n_capped, int((~tame).sum())


@pytest.mark.parametrize("seed,A", [(0, 8), (1, 8), (2, 8), (3, 6), (4, 5), (5, 7)])
def test_capped_steps_on_rough_traces_kernel_body(seed, A):
    def make(anchors, init):
        emu = _Emu(1, init.shape[0], anchors, init, False, 0)

        def step(mm, err, a, cov, dt):
            emu.step_imu(a, cov, 0.0)
            emu.step_toa(mm, err, dt)
            return emu.state()[0]
        return step
    n_capped, wild = _rough_run(make, seed, A)
    assert n_capped > 0
    assert (wild > 32) == (seed in (0, 2, 5))    # the traces that blow the filter up, and the ones that do not


@pytest.mark.gpu
@pytest.mark.parametrize("seed,A", [(0, 8), (1, 8), (2, 8), (3, 6), (4, 5), (5, 7)])
def test_capped_steps_on_rough_traces_gpu(seed, A):
    if not has_gpu():
        pytest.skip("no GPU")
    from roskfpos_amd import capi

    def make(anchors, init):
        bank = capi.KfposBank(capi.MODEL_TOA_IMU, init.shape[0], anchors, init_pos=init)

        def step(mm, err, a, cov, dt):
            bank.step_toa_imu(mm, err, a, cov, dt)
            return bank.get_state()[0]
        return step
    n_capped, _ = _rough_run(make, seed, A)
    assert n_capped > 0

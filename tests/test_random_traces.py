"""Randomised traces (hypothesis): the kernel body compiled for the host (tests/emu) against the oracle on shapes and
values the fixed cases do not reach -- odd anchor counts, ragged epochs, widely different errorEstimations, dt from 0
to seconds, outliers, all three filter models."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import oracle_py
from impls import emu_lib
from planar import CFG, PlanarEmu, PlanarOracle
from roskfpos_amd.synth import Workload, anchors_xyz

import ctypes as C


class _Emu:
    def __init__(self, model, T, anchors, init, ignore_worst, top_n):
        ipt = None if init is None else np.ascontiguousarray(init, dtype=np.float64)
        self._keep = ipt
        self.T, self.n = T, 9 if model == 1 else 6
        self.h = emu_lib().kfe_create(model, T, anchors.shape[0], np.ascontiguousarray(anchors), 0.5, 0.5,
                                      int(ignore_worst), 0.5, top_n, int(init is not None),
                                      None if ipt is None else ipt.ctypes.data)

    def __del__(self):
        if getattr(self, "h", None):
            emu_lib().kfe_destroy(self.h)
            self.h = None

    def step_toa(self, r, e, dt):
        s = np.zeros(self.T, dtype=np.uint32)
        d = np.atleast_1d(np.asarray(dt, dtype=np.float64))
        emu_lib().kfe_step_toa(self.h, np.ascontiguousarray(r, dtype=np.int32), np.ascontiguousarray(e), d, d.size,
                               s.ctypes.data)
        return s

    def step_imu(self, a, c, dt):
        s = np.zeros(self.T, dtype=np.uint32)
        emu_lib().kfe_step_imu(self.h, np.ascontiguousarray(a), np.ascontiguousarray(c),
                               np.array([dt], dtype=np.float64), 1, s.ctypes.data)
        return s

    def state(self):
        x, P = np.zeros((self.T, self.n)), np.zeros((self.T, self.n, self.n))
        emu_lib().kfe_get_state(self.h, x, P)
        return x, P


def _trace(rng, T, A, S, outliers):
    """ranges around a random walk inside the room, with dropouts and per-range errorEstimation spread"""
    anchors = anchors_xyz(A)
    p = np.stack([rng.uniform(1, 9, T), rng.uniform(1, 9, T), rng.uniform(0.5, 2.5, T)], axis=1)
    v = rng.normal(0, 0.5, (T, 3))
    out = []
    for s in range(S):
        dt = float(rng.choice([0.01, 0.05, 0.05, 0.1, 0.5, 2.0]))  # ranging epochs: the wall clock always advances
        p = p + v * dt
        d = np.sqrt(((p[:, None, :] - anchors[None]) ** 2).sum(-1))
        mm = np.floor((d + rng.normal(0, 0.05, d.shape)) * 1000).astype(np.int32)
        if outliers:
            mm[rng.random(mm.shape) < 0.05] += 900
        mm[rng.random(mm.shape) < 0.15] = rng.choice([0, -1])
        if rng.random() < 0.15:
            mm[rng.integers(0, T)] = 0  # a tag with no range at all
        err = 10.0 ** rng.uniform(-4, -1, mm.shape)
        out.append((mm, err, dt, rng.normal(0, 0.3, (T, 3)),
                    np.tile((10.0 ** rng.uniform(-3, -1)) * np.eye(3).ravel(), (T, 1))))
    return anchors, p, out


COMMON = dict(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))


def _tame(status, gain_cap):
    """Tags whose epoch stayed in the regime where parity means something: the Gauss-Newton ML solve converged
    within 30 passes and the IEKF did not run into its iteration cap. A GN walk of 100+ passes or an IEKF that
    diverges (reference behaviour, reproduced) is a chaotic map: the last bit of any intermediate decides where it
    ends, so two correct implementations differ by metres there."""
    ok = (((status >> 16) & 0xFF) <= 30) & (((status >> 8) & 0xFF) < gain_cap)
    # an update (not an initialisation) the oracle skipped: exactly singular J' W J in the per-epoch ML solve -- anchors
    # and estimate on one line of this symmetric synthetic room -- which the kernels do not detect (kfpos_core.h)
    return ok & ~(((status & 0xFF) == 1))


def _judge(d_kernel, d_probe):
    """The kernel must be as close to the oracle as an independent LAPACK restatement of the same algorithm is (x1000: the kernel evaluates a re-derived formula, the probe the same one),
    or within rounding: random traces reach regimes (the 9-state iteration diverging over 2 s gaps, four coplanar
    anchors) where the problem itself amplifies the last bit by many orders of magnitude."""
    bound = np.maximum(1e-6, 1000.0 * d_probe)  # 1e-6 m: the parity bar of BASELINE.json
    assert np.all(d_kernel <= bound), (d_kernel, d_probe)


@settings(**COMMON)
@given(seed=st.integers(0, 2**31 - 1), A=st.integers(4, 11), model=st.sampled_from([0, 1]),
       fixed=st.booleans(), heuristic=st.sampled_from(["none", "ignore_worst", "top2"]))
def test_toa_filters_match_oracle(seed, A, model, fixed, heuristic):
    import numpy_oracle as npo
    rng = np.random.default_rng(seed)
    T, S = 5, 8
    anchors, p0, trace = _trace(rng, T, A, S, outliers=(heuristic != "none"))
    if model == 1:
        heuristic = "none"
    iw, tn = heuristic == "ignore_worst", 2 if heuristic == "top2" else 0
    init = p0 if fixed else None
    orc = oracle_py.OracleBank(model, T, anchors, ignore_worst=iw, top_n=tn, init_pos=init)
    emu = _Emu(model, T, anchors, init, iw, tn)
    probe = None
    if tn == 0:  # the numpy restatement has no top-N composition
        probe = [npo.NumpyFilter(model, anchors, ignore_worst=iw, init_pos=None if init is None else init[t])
                 for t in range(T)]
    tame = np.ones(T, dtype=bool)
    for mm, err, dt, acc, cov in trace:
        if model == 1:
            so, se = orc.step_imu(acc, cov, 0.0), emu.step_imu(acc, cov, 0.0)
        so, se = orc.step_toa(mm, err, dt), emu.step_toa(mm, err, dt)
        tame &= _tame(so, 20 if model == 1 else 10)
        for t in range(T if probe else 0):
            try:
                if model == 1:
                    probe[t].step_imu(acc[t], cov[t], 0.0)
                probe[t].step_toa(mm[t], err[t], dt)
            except npo.LinAlgThrow:  # LAPACK calls a matrix singular that the oracle's LU still factors: no opinion
                probe[t].pos = np.full(3, np.nan)
    xo, Po = orc.get_state()
    xe, Pe = emu.state()
    fin = np.isfinite(xo).all(1)
    assert np.array_equal(fin[tame], np.isfinite(xe).all(1)[tame])
    fin &= tame
    if not fin.any():
        return
    d_kernel = np.abs(xe[fin, :3] - xo[fin, :3]).max(1)
    if probe:
        xn = np.stack([f.pos for f in probe])
        _judge(d_kernel, np.nan_to_num(np.abs(xn[fin] - xo[fin, :3]).max(1), nan=np.inf))
    else:
        assert np.median(d_kernel) < 1e-7 and (d_kernel < 1e-6).mean() >= 0.8, d_kernel


@settings(**COMMON)
@given(seed=st.integers(0, 2**31 - 1), A=st.integers(4, 10), fixed=st.booleans(), fixed_height=st.booleans(),
       sensors=st.booleans())
def test_planar_filter_matches_oracle(seed, A, fixed, fixed_height, sensors):
    rng = np.random.default_rng(seed)
    T, S = 5, 8
    anchors, p0, trace = _trace(rng, T, A, S, outliers=False)

    class W:  # the slice of Workload the planar wrappers read
        n_tags, n_anchors = T, A
    W.anchors = anchors
    cfg = dict(CFG, use_fixed_height=int(fixed_height), fixed_height=float(p0[:, 2].mean()))
    init = p0 if fixed else None
    import numpy_oracle as npo
    orc, emu = PlanarOracle(W, cfg, init), PlanarEmu(W, cfg, init, sensors=True)
    probe = [npo.NumpyPlanarFilter(anchors, 0.5, 0.5, init_pos=None if init is None else init[t], **cfg) for t in range(T)]
    tame = np.ones(T, dtype=bool)
    for mm, err, dt, acc, cov in trace:
        if sensors:
            w3 = np.concatenate([np.zeros((T, 2)), rng.normal(0, 0.2, (T, 1))], axis=1)
            cw = np.tile(np.eye(3).ravel() * 1e-3, (T, 1))
            args = (w3, cw, acc, cov, 0.01)
            s1, s2 = orc.step_planar_imu(*args), emu.step_planar_imu(*args)
            tame &= _tame(s1, 20)
            assert np.array_equal((s1 & 0xFF)[tame], (s2 & 0xFF)[tame])
            comp = rng.uniform(-4, 4, T)
            s1, s2 = orc.step_compass(comp, 0.01), emu.step_compass(comp, 0.01)
            tame &= _tame(s1, 20)
            assert np.array_equal((s1 & 0xFF)[tame], (s2 & 0xFF)[tame])
            for t in range(T):
                probe[t].step_imu(w3[t], cw[t], acc[t], cov[t], 0.01)
                probe[t].step_compass(comp[t], 0.01)
        so, se = orc.step_toa(mm, err, dt), emu.step_toa(mm, err, dt)
        tame &= _tame(so, 20)
        assert np.array_equal((so & 0xFF)[tame], (se & 0xFF)[tame]), (so & 0xFF, se & 0xFF)
        for t in range(T):
            try:
                probe[t].step_toa(mm[t], err[t], dt)
            except npo.LinAlgThrow:
                probe[t].xy = np.full(2, np.nan)
    xo, _ = orc.get_state()
    xe, _ = emu.get_state()
    fin = np.isfinite(xo).all(1)
    assert np.array_equal(fin[tame], np.isfinite(xe).all(1)[tame])
    fin &= tame
    if fin.any():
        xn = np.stack([f.state() for f in probe])
        _judge(np.abs(xe[fin] - xo[fin]).max(1), np.nan_to_num(np.abs(xn[fin] - xo[fin]).max(1), nan=np.inf))

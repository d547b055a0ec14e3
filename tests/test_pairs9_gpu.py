"""The tail of the 9-state gain iteration on pairs of lanes (iekf9_pairs; opt-in with KFPOS_PAIR9=1,
read at kfpos_create) against the same kernel without the hand-over (the default): bit for bit -- state, covariance, status
words (iteration counts), poses of every epoch -- on banks that fill their wavefronts, on one that does not (its last
wavefront never forms pairs), with silent tags (a wavefront with a skipped lane never forms pairs) and with absent ranges."""
import os

import numpy as np
import pytest

from conftest import has_gpu
from roskfpos_amd.synth import Workload

pytestmark = pytest.mark.gpu


def _bank(T, w, storage, pairs):
    from roskfpos_amd import capi
    old = os.environ.get("KFPOS_PAIR9")
    os.environ["KFPOS_PAIR9"] = "1" if pairs else "0"
    try:
        return capi.KfposBank(capi.MODEL_TOA_IMU, T, w.anchors, storage=storage, init_pos=w.init_positions())
    finally:
        if old is None:
            os.environ.pop("KFPOS_PAIR9")
        else:
            os.environ["KFPOS_PAIR9"] = old


@pytest.mark.parametrize("T,storage", [(4096, 2), (4096, 0), (1000, 2), (4096, 1)])
def test_pairs_are_invisible_per_epoch_calls(T, storage):
    if not has_gpu():
        pytest.skip("no GPU")
    from roskfpos_amd import capi
    S = 40
    w = Workload(T, 8)
    real = np.float64 if storage == capi.STORE_F64 else np.float32
    err, cov = w.err_est(real), w.accel_cov(real)
    a, b = _bank(T, w, storage, True), _bank(T, w, storage, False)
    capped = 0
    for s in range(S):
        r = w.ranges_mm(s)
        if s % 7 == 3:
            r[:, 1] = -1
        if s % 11 == 5:
            r[::3, 2:] = 0          # fewer than four ranges: no update for every third tag
        dt = w.dt_of(s)
        if s % 9 == 4:              # silent tags: their wavefronts run without pairs
            dts = np.full(T, dt)
            dts[::97] = -1.0
            dt = dts
        acc = w.accel(s, real)
        sa = a.step_toa_imu(r, err, acc, cov, dt)
        sb = b.step_toa_imu(r, err, acc, cov, dt)
        assert np.array_equal(sa, sb), f"status words, epoch {s}"
        capped += int((((sa >> 8) & 0xFF) >= 20).sum())
        pa, pb = a.get_pose(0.0), b.get_pose(0.0)
        assert np.array_equal(pa[0], pb[0], equal_nan=True), f"poses, epoch {s}"
    xa, Pa, fa = a.get_state()
    xb, Pb, fb = b.get_state()
    assert np.array_equal(xa, xb) and np.array_equal(Pa, Pb) and np.array_equal(fa, fb)
    assert capped > 0.02 * T * S      # the trace does run tags to the cap: the tail exists
    a.close()
    b.close()


def test_pairs_are_invisible_fused_launches_and_match_oracle():
    if not has_gpu():
        pytest.skip("no GPU")
    import torch
    import oracle_py
    from roskfpos_amd import capi
    from roskfpos_amd.dist import device_trace
    T, S = 8192, 50
    w = Workload(T, 8)
    dev = torch.device("cuda:0")
    tr = device_trace(torch, w, S, dev, True, np.float32)
    out = []
    for pairs in (True, False):
        b = _bank(T, w, capi.STORE_MIXED, pairs)
        traj = torch.zeros((S, 3, T), dtype=torch.float64, device=dev)
        b.run_trace_dev(S, tr["ranges"], 8 * T, tr["err"], 0, tr["dts"], accel=tr["accel"], stride_accel=3 * T,
                        cov=tr["cov"], stride_cov=0, trajectory=traj)
        torch.cuda.synchronize()
        x, P, _ = b.get_state()
        out.append((traj.cpu().numpy(), x, P))
        b.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    n = 512
    o = oracle_py.OracleBank(1, n, w.anchors, init_pos=w.init_positions()[:n], n_threads=8)
    err, cov = w.err_est(np.float32).astype(np.float64)[:n], w.accel_cov(np.float32).astype(np.float64)[:n]
    for s in range(S):
        o.step_imu(w.accel(s, np.float32).astype(np.float64)[:n], cov, 0.0)
        o.step_toa(w.ranges_mm(s)[:n], err, w.dt_of(s))
    xo, _ = o.get_state()
    rms = float(np.sqrt(((out[0][1][:n, :3] - xo[:, :3]) ** 2).sum(1).mean()))
    assert rms <= 1e-6, rms

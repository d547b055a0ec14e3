"""Banks with more wavefronts than the chip has SIMDs (> 65 536 tags) run the plain 8-anchor 6-state filter in its
256-register build (k_step_toa6_w2: epoch in LDS, most of the covariance parked there during the ML solve, two
wavefronts per SIMD). Same arithmetic: state, covariance, status words and poses must equal the one-wavefront build's
(KFPOS_ONE_WAVE_BUILD=1, read at kfpos_create) bit for bit, in every storage mode, per-epoch and in fused launches, with
absent ranges and silent tags."""
import os

import numpy as np
import pytest

from conftest import has_gpu
from roskfpos_amd.synth import Workload

pytestmark = pytest.mark.gpu


def _bank(T, w, storage, one_wave):
    from roskfpos_amd import capi
    old = os.environ.get("KFPOS_ONE_WAVE_BUILD")
    os.environ["KFPOS_ONE_WAVE_BUILD"] = "1" if one_wave else "0"
    try:
        return capi.KfposBank(capi.MODEL_TOA, T, w.anchors, storage=storage, init_pos=w.init_positions())
    finally:
        if old is None:
            os.environ.pop("KFPOS_ONE_WAVE_BUILD")
        else:
            os.environ["KFPOS_ONE_WAVE_BUILD"] = old


@pytest.mark.parametrize("storage", [0, 1, 2])
def test_two_wave_build_equals_one_wave_build(storage):
    if not has_gpu():
        pytest.skip("no GPU")
    import torch
    from roskfpos_amd import capi
    from roskfpos_amd.dist import device_trace
    T, S = 65536 + 4096 + 37, 12      # 1089 wavefronts on 1024 SIMDs, the last one ragged
    w = Workload(T, 8)
    real = np.float64 if storage == capi.STORE_F64 else np.float32
    err = w.err_est(real)
    a, b = _bank(T, w, storage, False), _bank(T, w, storage, True)
    for s in range(S):
        r = w.ranges_mm(s)
        if s % 5 == 3:
            r[:, 1] = -1
        if s % 7 == 5:
            r[::3, 2:] = 0
        dt = w.dt_of(s)
        if s % 4 == 2:
            dts = np.full(T, dt)
            dts[::97] = -1.0
            dt = dts
        sa, sb = a.step_toa(r, err, dt), b.step_toa(r, err, dt)
        assert np.array_equal(sa, sb), f"status words, epoch {s}"
        assert np.array_equal(a.get_pose(0.0)[0], b.get_pose(0.0)[0], equal_nan=True), f"poses, epoch {s}"
    dev = torch.device("cuda:0")
    tr = device_trace(torch, w, 10, dev, False, real)
    trajs = []
    for bank in (a, b):
        traj = torch.zeros((10, 3, T), dtype=torch.float64, device=dev)
        bank.run_trace_dev(10, tr["ranges"], 8 * T, tr["err"], 0, tr["dts"], trajectory=traj)
        torch.cuda.synchronize()
        trajs.append(traj.cpu().numpy())
    assert np.array_equal(trajs[0], trajs[1])
    xa, Pa, fa = a.get_state()
    xb, Pb, fb = b.get_state()
    assert np.array_equal(xa, xb) and np.array_equal(Pa, Pb) and np.array_equal(fa, fb)
    a.close()
    b.close()

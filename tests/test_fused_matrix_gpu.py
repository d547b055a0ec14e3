"""One multi-epoch launch == as many single-epoch launches, bit for bit -- over every kernel family the library selects
between: model (6-state symmetric / 6-state with ML initialisation = full covariance layout / 9-state / ML estimator /
planar), storage mode (f64, f32, mixed, P48), anchor count (4 and 8 in registers or 8 lanes per tag, 12 = run-time loop,
16 = compile-time loops over LDS), outlier heuristics, fresh and latched IMU samples. Round 2 shipped a kernel that
fetched only the first epoch's accelerometer sample in such launches for one corner of this matrix (run-time anchor
count x f64 or f32 storage); nothing like it should be able to hide again."""
import numpy as np
import pytest

from conftest import has_gpu
from roskfpos_amd.synth import Workload

pytestmark = pytest.mark.gpu

PLANAR = dict(use_fixed_height=1, fixed_height=1.0, init_angle=0.3, px4_height=0.8, px4_arm_p1=0.05, px4_arm_p2=-0.02,
              px4_cov_velocity=0.002, px4_cov_gyro_z=0.001, imu_use_fixed_cov_acc=0, imu_cov_acc=0.02,
              imu_use_fixed_cov_ang_vel_z=1, imu_cov_ang_vel_z=0.0005, mag_angle_offset=0.0, mag_cov=0.01)

CONFIGS = []
for storage in (0, 1, 2, 3):
    for A in (4, 8, 12, 16):
        CONFIGS += [("toa6", storage, A, {}), ("imu9", storage, A, {})]
    CONFIGS += [("toa6_mlinit", storage, 8, {}), ("toa6_mlinit", storage, 12, {}),
                ("toa6", storage, 8, dict(ignore_worst=True)), ("toa6", storage, 16, dict(top_n=2)),
                ("toa6", storage, 12, dict(top_n=1, ignore_worst=True)),
                ("toa6_big", storage, 8, {}),      # > 8 192 tags: one tag per lane instead of 8 lanes per tag
                ("ml", storage, 8, {}), ("ml", storage, 12, dict(top_n=2)), ("ml_best", storage, 5, {}),
                ("planar", storage, 8, {}), ("planar", storage, 12, {}), ("planar_sens", storage, 8, {})]


@pytest.mark.parametrize("kind,storage,A,opts", CONFIGS, ids=[f"{k}-s{s}-A{a}" + "".join(f"-{o}" for o in op)
                                                                for k, s, a, op in CONFIGS])
def test_fused_launch_equals_single_epoch_launches(kind, storage, A, opts, monkeypatch):
    if not has_gpu():
        pytest.skip("no GPU")
    import torch
    from roskfpos_amd import capi
    T = 9000 if kind == "toa6_big" else 333          # 333: ragged last wavefront / last group of 8 lanes
    S = 7
    w = Workload(T, A)
    real = np.float64 if storage == capi.STORE_F64 else np.float32
    model = {"toa6": 0, "toa6_mlinit": 0, "toa6_big": 0, "imu9": 1, "ml": 2, "ml_best": 2, "planar": 3, "planar_sens": 3}[kind]
    init = None if kind == "toa6_mlinit" else w.init_positions() + (0.2 if model == 2 else 0.0)
    kw = dict(storage=storage, init_pos=init, planar=PLANAR if model == 3 else None,
              ml_variant=capi.ML_BEST if kind == "ml_best" else 0, **opts)
    dev = "cuda:0"
    r = np.stack([w.ranges_mm(s) for s in range(S)])
    r[2, ::7, 1] = -1                  # a dropped range, an epoch with too few ranges for some tags
    r[4, 1::9, 2:] = 0
    rt = torch.from_numpy(np.ascontiguousarray(r.transpose(0, 2, 1))).to(dev)
    et = torch.from_numpy(np.ascontiguousarray(w.err_est(real).T)).to(dev)
    at = torch.from_numpy(np.ascontiguousarray(np.stack([w.accel(s, real) for s in range(S)]).transpose(0, 2, 1))).to(dev)
    ct = torch.from_numpy(np.ascontiguousarray(w.accel_cov(real).T)).to(dev)
    dts = np.array([w.dt_of(s) for s in range(S)])
    stream = torch.cuda.current_stream().cuda_stream

    def make():
        b = capi.KfposBank(model, T, w.anchors, **kw)
        if kind == "planar_sens":     # latch an IMU and a compass sample: ranging epochs carry their rows from now on
            wv, la = w.planar_imu(0)
            b.step_planar_imu(wv, np.tile(np.eye(3).ravel() * 1e-4, (T, 1)), la, w.accel_cov(), 0.1)
            b.step_compass(w.compass(0), 0.0)
        return b

    results = []
    for chunk in (None, "3", "1"):    # one launch for all epochs; launches of 3 + 3 + 1; one launch per epoch
        if chunk:
            monkeypatch.setenv("KFPOS_TRACE_CHUNK_STEPS", chunk)
        b = make()
        traj = torch.zeros(S, 3, T, dtype=torch.float64, device=dev)
        st = torch.zeros(T, dtype=torch.int32, device=dev)
        if model == 1:
            b.run_trace_dev(S, rt, A * T, et, 0, dts, accel=at, stride_accel=3 * T, cov=ct, stride_cov=0,
                            trajectory=traj, status=st, stream=stream)
        else:
            b.run_trace_dev(S, rt, A * T, et, 0, dts, trajectory=traj, status=st, stream=stream)
        torch.cuda.synchronize()
        x, P, fl = b.get_state()
        results.append((traj.cpu().numpy(), st.cpu().numpy(), x, P, fl, b.get_latch() if model in (1, 3) else None))
        b.close()
    # ... and through the single-epoch entry points
    b = make()
    st = torch.zeros(T, dtype=torch.int32, device=dev)
    for s in range(S):
        if model == 1:
            b.step_toa_imu_dev(rt[s], et, at[s], ct, dts[s], status=st, stream=stream)
        else:
            b.step_toa_dev(rt[s], et, dts[s], status=st, stream=stream)
    torch.cuda.synchronize()
    x, P, fl = b.get_state()
    ref = (None, st.cpu().numpy(), x, P, fl, b.get_latch() if model in (1, 3) else None)
    b.close()
    for k, got in enumerate(results):
        for name, a_, b_ in zip(("status", "x", "P", "flags"), got[1:5], ref[1:5]):
            assert np.array_equal(a_, b_, equal_nan=(name != "status" and name != "flags")), (k, name)
        if ref[5] is not None:
            assert np.array_equal(got[5], ref[5], equal_nan=True), (k, "latch")
        assert np.array_equal(got[0], results[0][0], equal_nan=True), (k, "trajectory")
    assert np.isfinite(results[0][2]).mean() > 0.95


@pytest.mark.parametrize("storage,A", [(0, 8), (2, 8), (3, 12)])
def test_trace_replay_with_a_sensor_covariance_per_epoch(storage, A):
    """kfpos_run_trace_dev with stride_cov != 0 -- an accelerometer covariance that changes from epoch to epoch -- runs one
    epoch per launch (the kernels whiten the covariance once per launch) and must equal the single-epoch entry point."""
    if not has_gpu():
        pytest.skip("no GPU")
    import torch
    from roskfpos_amd import capi
    T, S = 500, 6
    w = Workload(T, A)
    real = np.float64 if storage == capi.STORE_F64 else np.float32
    dev = "cuda:0"
    rt = torch.from_numpy(np.ascontiguousarray(np.stack([w.ranges_mm(s) for s in range(S)]).transpose(0, 2, 1))).to(dev)
    et = torch.from_numpy(np.ascontiguousarray(w.err_est(real).T)).to(dev)
    at = torch.from_numpy(np.ascontiguousarray(np.stack([w.accel(s, real) for s in range(S)]).transpose(0, 2, 1))).to(dev)
    covs = np.stack([w.accel_cov(real) * real(1.0 + 0.3 * s) for s in range(S)])          # [S][T][9]
    ct = torch.from_numpy(np.ascontiguousarray(covs.transpose(0, 2, 1))).to(dev)            # [S][9][T]
    dts = np.array([w.dt_of(s) for s in range(S)])
    stream = torch.cuda.current_stream().cuda_stream
    a = capi.KfposBank(1, T, w.anchors, storage=storage, init_pos=w.init_positions())
    b = capi.KfposBank(1, T, w.anchors, storage=storage, init_pos=w.init_positions())
    a.run_trace_dev(S, rt, A * T, et, 0, dts, accel=at, stride_accel=3 * T, cov=ct, stride_cov=9 * T, stream=stream)
    for s in range(S):
        b.step_toa_imu_dev(rt[s], et, at[s], ct[s], dts[s], stream=stream)
    torch.cuda.synchronize()
    xa, Pa, fa = a.get_state()
    xb, Pb, fb = b.get_state()
    assert np.array_equal(xa, xb) and np.array_equal(Pa, Pb) and np.array_equal(fa, fb)
    assert np.array_equal(a.get_latch(), b.get_latch())
    a.close()
    b.close()

"""GPU parity tests: libkfpos_hip.so through the C ABI against the oracle, on the same seeded traces.

Bar (BASELINE.json north_star): <= 1e-6 m RMS position difference vs the reference CPU EKF.
All arithmetic is float64; KFPOS_STORE_F32 only narrows what is kept in HBM between steps.
"""
import os

import numpy as np
import pytest

from cases import CASES, CASE_BY_NAME, Case, drive, rms_and_max
from impls import GpuImpl, OracleImpl

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
RMS_BAR = 1e-6  # metres, from BASELINE.json


def _gpu(storage):
    return lambda case, w, init: GpuImpl(case, w, init, storage=storage)


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_f64_storage_matches_oracle(case):
    fo, po, so = drive(case, OracleImpl, record=True)
    fg, pg, sg = drive(case, _gpu(0), record=True)
    rms, mx, same_nan = rms_and_max(pg, po)
    assert same_nan
    assert rms <= 1e-9 and mx <= 1e-8, (rms, mx)
    assert np.array_equal(so, sg)  # iteration counts, flags and ignored anchors identical
    xo, Po = fo.state()
    xg, Pg = fg.state()
    ok = np.isfinite(xo[:, 0])
    assert np.abs(Po[ok] - Pg[ok]).max() <= 1e-9 * np.abs(Po[ok]).max()


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_mixed_storage_matches_oracle(case):
    """KFPOS_STORE_MIXED (f64 covariance, f32 measurements): the configuration bench.py measures."""
    fo, po, so = drive(case, OracleImpl, real=np.float32, record=True)
    fg, pg, sg = drive(case, _gpu(2), real=np.float32, record=True)
    rms, mx, same_nan = rms_and_max(pg, po)
    assert same_nan
    assert rms <= 1e-9 and mx <= 1e-8, (rms, mx)
    assert np.array_equal(so, sg)


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_f32_storage_on_edge_case_traces(case):
    """KFPOS_STORE_F32: fp32 covariance + measurements, fp64 positions, velocities and arithmetic.

    The 6-state filter stays ~1e-8 m from the oracle. The 9-state filter amplifies the 6e-8 relative
    rounding of a 24-bit covariance to 1.6e-6 .. 2.5e-6 m RMS (also on the clean BASELINE trace), which is
    above the 1e-6 m bar: F32 is therefore a documented reduced-accuracy mode for that filter (bound
    checked here: 5e-6 m) and the parity configuration for BASELINE config 3 is KFPOS_STORE_MIXED."""
    fo, po, so = drive(case, OracleImpl, real=np.float32, record=True)
    fg, pg, sg = drive(case, _gpu(1), real=np.float32, record=True)
    rms, mx, same_nan = rms_and_max(pg, po)
    assert same_nan
    assert rms <= (5e-6 if case.model == 1 else RMS_BAR), (rms, mx)


@pytest.mark.parametrize("model,A,top_n,storage", [(1, 8, 0, 2), (0, 16, 2, 1), (0, 8, 0, 1),
                                                     (1, 8, 0, 3), (0, 16, 2, 3), (1, 12, 0, 3)])
def test_compact_storage_meets_the_rms_bar_on_baseline_traces(model, A, top_n, storage):
    """BASELINE configs 3 / 5 / 2 shapes on the clean synthetic trace of SURVEY.md 8d, 512 tags x 100 steps:
    config 3 with f32 measurements + f64 covariance, configs 5 and 2 with everything f32 -- and configs 3 and 5 as
    BASELINE.json words them, "fp32" = compact storage, in the mode that keeps the 9-state filter inside the bar:
    KFPOS_STORE_P48 (6-byte covariance entries, f32 measurements)."""
    from roskfpos_amd import capi
    from roskfpos_amd.synth import Workload
    import oracle_py
    T, S = 512, 100
    w = Workload(T, A)
    e32, c32 = w.err_est(np.float32), w.accel_cov(np.float32)
    gpu = capi.KfposBank(model, T, w.anchors, storage=storage, top_n=top_n, init_pos=w.init_positions())
    orc = oracle_py.OracleBank(model, T, w.anchors, top_n=top_n, init_pos=w.init_positions(), n_threads=8)
    sq = 0.0
    for s in range(S):
        r, a32, dt = w.ranges_mm(s), w.accel(s, np.float32), w.dt_of(s)
        if model == 1:
            gpu.step_toa_imu(r, e32, a32, c32, dt)
            orc.step_imu(a32.astype(np.float64), c32.astype(np.float64), 0.0)
        else:
            gpu.step_toa(r, e32, dt)
        orc.step_toa(r, e32.astype(np.float64), dt)
        xg, _, _ = gpu.get_state()
        xo, _ = orc.get_state()
        sq += ((xg[:, :3] - xo[:, :3]) ** 2).sum()
    rms = np.sqrt(sq / (T * S))
    assert rms <= RMS_BAR, rms
    if storage == 3:
        assert rms <= 5e-9, rms   # the encoding study on the host build of the kernel body: 5.5e-10 m (9-state)


@pytest.mark.parametrize("case", [c for c in CASES if c.name in (
    "toa6_A8_fixed", "toa6_A8_mlinit", "toa6_A16_top2", "toa6_A8_ignoreworst", "imu9_A8_fixed", "imu9_A8_latched",
    "imu9_A12_generic", "toa6_A5_generic")], ids=lambda c: c.name)
def test_p48_storage_on_edge_case_traces(case):
    """KFPOS_STORE_P48 on the ragged / degraded parity traces, every kernel family (register-resident, LDS-staged,
    generic, full covariance layout, 8 lanes per tag): 36 mantissa bits of covariance keep both filters within 1e-7 m
    of the oracle with the same flags (iteration counts may differ by one where a stop decision sits on its threshold)."""
    fo, po, so = drive(case, OracleImpl, real=np.float32, record=True)
    fg, pg, sg = drive(case, _gpu(3), real=np.float32, record=True)
    rms, mx, same_nan = rms_and_max(pg, po)
    assert same_nan
    assert rms <= 1e-7 and mx <= 1e-6, (rms, mx)
    assert np.array_equal(so & 0xFF, sg & 0xFF)
    xo, Po = fo.state()
    xg, Pg = fg.state()
    ok = np.isfinite(Po).all(axis=(1, 2))
    assert np.abs(Po[ok] - Pg[ok]).max() <= 1e-7 * np.abs(Po[ok]).max()


from p48 import p48_round_trip as _p48  # numpy mirror of roskfpos_amd/csrc/kfpos_p48.h


def test_p48_state_round_trip_and_fused_launch_equals_single_epochs():
    """kfpos_set_state / kfpos_get_state round and encode the covariance exactly as the kernels store it (48 bits: sign,
    8 exponent bits, 39 mantissa bits), and a fused multi-epoch launch rounds between its epochs like as many single-epoch
    launches do."""
    import torch
    from roskfpos_amd import capi
    from roskfpos_amd.synth import Workload
    T, A, S = 777, 8, 9
    w = Workload(T, A)
    r, a, dt, rt, at, et, ct = _torch_trace(w, S, np.float32)
    stream = torch.cuda.current_stream().cuda_stream
    one = capi.KfposBank(1, T, w.anchors, storage=3, init_pos=w.init_positions())
    fused = capi.KfposBank(1, T, w.anchors, storage=3, init_pos=w.init_positions())
    for s in range(S):
        one.step_toa_imu_dev(rt[s], et, at[s], ct, dt[s], stream=stream)
    fused.run_trace_dev(S, rt, A * T, et, 0, dt, accel=at, stride_accel=3 * T, cov=ct, stride_cov=0, stream=stream)
    torch.cuda.synchronize()
    x1, P1, _ = one.get_state()
    x2, P2, _ = fused.get_state()
    assert np.array_equal(x1, x2) and np.array_equal(P1, P2)
    bits = P1.view(np.uint64)
    assert np.all((bits & np.uint64(0x1FFF)) == 0) and np.any(bits & np.uint64(0x1FFFE000))   # 40 significant bits, all used
    assert np.array_equal(_p48(P1), P1)                                  # what the kernels keep is on the grid of the codec
    # round trip through the host encoders (roskfpos_amd/csrc/kfpos_p48.h, mirrored in numpy by _p48 above)
    rng = np.random.default_rng(3)
    P = rng.normal(size=(T, 9, 9)) * 10.0 ** rng.integers(-8, 3, size=(T, 1, 1))
    P = P + P.transpose(0, 2, 1)
    # special values: a NaN stays a NaN, an infinity becomes one (inf - inf inside the rounding: an overflowed covariance
    # entry is not a number either way), magnitudes below single's normal range are stored as signed zero, an exact power
    # of two and a value one step of the 40-bit grid below it keep their bits
    P[0, 0, 0], P[1, 2, 2], P[2, 3, 3], P[3, 4, 4] = np.inf, -np.inf, np.nan, -1e-39
    P[4, 5, 5], P[5, 6, 6], P[6, 7, 7] = 2.0 ** -20, 2.0 ** -20 * (1 - 2.0 ** -40), 0.0
    expect = _p48(P)
    one.set_state(x1, P)
    _, Pb, _ = one.get_state()
    nan = np.isnan(expect)
    assert nan[0, 0, 0] and nan[1, 2, 2] and nan[2, 3, 3] and nan.sum() == 3
    assert np.array_equal(np.isnan(Pb), nan) and np.array_equal(Pb[~nan], expect[~nan])
    assert Pb[3, 4, 4] == 0.0 and np.signbit(Pb[3, 4, 4]) and Pb[4, 5, 5] == P[4, 5, 5] and Pb[5, 6, 6] == P[5, 6, 6]
    fin = ~nan & (np.abs(P) > 1e-30)
    assert (np.abs(Pb[fin] - P[fin]) / np.abs(P[fin])).max() <= 2.0 ** -40
    one.set_state(x1, Pb)                                                 # idempotent: values on the grid pass unchanged
    _, Pc, _ = one.get_state()
    assert np.array_equal(Pc[~nan], Pb[~nan])
    # ... and through a kernel: getPose reads the stored covariance and reports NaN where it is NaN
    _, cov3, _, _ = one.get_pose(0.0)
    assert np.isnan(cov3[2]).any() and np.isfinite(cov3[5]).all()

@pytest.mark.parametrize("name", sorted(CASE_BY_NAME))
def test_gpu_reproduces_golden_fixture(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    c = CASE_BY_NAME[name]
    f, pos, st = drive(c, _gpu(0), steps=int(g["steps"]), record=True)
    rms, mx, same_nan = rms_and_max(pos, g["positions"])
    assert same_nan and rms <= 1e-9, (rms, mx)
    assert np.array_equal(st, g["status"])


@pytest.mark.parametrize("name", ["toa6_A8_fixed", "imu9_A8_fixed"])
def test_get_pose_and_idempotence(name):
    case = CASE_BY_NAME[name]
    fo = drive(case, OracleImpl, steps=20)
    fg = drive(case, _gpu(0), steps=20)
    x0, P0 = fg.state()
    for dt_ahead in (0.0, 0.05, 0.4):
        po, co, vo, _ = fo.pose(dt_ahead)
        pg, cg, vg, stg = fg.pose(dt_ahead)
        assert np.all(stg == 0)
        assert np.allclose(pg, po, atol=1e-9) and np.allclose(cg, co, rtol=1e-7, atol=1e-13)
        if case.model == 1:
            assert np.allclose(vg, vo, atol=1e-9)
    x1, P1 = fg.state()
    assert np.array_equal(x0, x1) and np.array_equal(P0, P1)  # getPose never modifies the filter


def test_not_started_and_empty_epochs():
    from roskfpos_amd import capi
    from roskfpos_amd.synth import Workload
    w = Workload(70, 8)  # not a multiple of the wavefront width: ragged last workgroup
    b = capi.KfposBank(capi.MODEL_TOA, 70, w.anchors, init_pos=w.init_positions())
    pos, cov, vel, st = b.get_pose(0.1)
    assert np.all(st == capi.ST_NOT_STARTED) and np.all(np.isnan(pos))
    st = b.step_toa(np.zeros((70, 8), dtype=np.int32), w.err_est(), 0.1)  # no range at all
    assert np.all(st & capi.ST_FEW_RANGES)
    x, P, fl = b.get_state()
    assert np.allclose(x[:, :3], w.init_positions()) and np.all(fl & 1)
    # state round trip (checkpoint / restore)
    b2 = capi.KfposBank(capi.MODEL_TOA, 70, w.anchors, init_pos=w.init_positions())
    b2.set_state(x, P, fl)
    r = w.ranges_mm(1)
    b.step_toa(r, w.err_est(), 0.05)
    b2.step_toa(r, w.err_est(), 0.05)
    xa, Pa, _ = b.get_state()
    xb, Pb, _ = b2.get_state()
    assert np.array_equal(xa, xb) and np.array_equal(Pa, Pb)


def test_argument_errors():
    from roskfpos_amd import capi
    from roskfpos_amd.synth import anchors_xyz
    with pytest.raises(capi.KfposError, match="model"):
        capi.KfposBank(7, 4, anchors_xyz(4), init_pos=np.zeros(3))          # unknown model
    with pytest.raises(capi.KfposError, match="n_tags"):
        capi.KfposBank(capi.MODEL_TOA, 0, anchors_xyz(4), init_pos=np.zeros(3))  # no tags
    with pytest.raises(capi.KfposError, match="6-state"):
        capi.KfposBank(capi.MODEL_TOA_IMU, 4, anchors_xyz(4), top_n=1, init_pos=np.zeros(3))
    b = capi.KfposBank(capi.MODEL_TOA, 4, anchors_xyz(4), init_pos=np.zeros(3))
    with pytest.raises(capi.KfposError):
        b.step_toa_imu(np.ones((4, 4), dtype=np.int32), np.ones((4, 4)), np.zeros((4, 3)), np.zeros((4, 9)), 0.1)


def _torch_trace(w, S, real):
    import torch
    r, a, dt = w.trace(S)
    dev = "cuda:0"
    rt = torch.from_numpy(np.ascontiguousarray(r.transpose(0, 2, 1))).to(dev)               # [S][A][T]
    at = torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 1)).astype(real)).to(dev)  # [S][3][T]
    et = torch.from_numpy(np.ascontiguousarray(w.err_est(real).T)).to(dev)                  # [A][T]
    ct = torch.from_numpy(np.ascontiguousarray(w.accel_cov(real).T)).to(dev)                # [9][T]
    return r, a, dt, rt, at, et, ct


@pytest.mark.parametrize("model,storage,A,generic", [
    (0, 0, 8, False), (1, 0, 8, False), (1, 1, 8, False), (1, 2, 8, False),
    # the run-time-anchor-count kernels (epoch staged in LDS per epoch): a multi-epoch launch must fetch every epoch's
    # accelerometer sample too -- in round 2 it re-fused the launch's first one unless the storage mode was MIXED
    (1, 0, 12, False), (1, 1, 12, False), (1, 2, 12, False), (1, 0, 8, True), (1, 1, 8, True), (0, 0, 12, False)])
def test_device_api_and_trace_replay_equal_host_api(model, storage, A, generic, monkeypatch):
    """Device-pointer entry points (component-major, torch-owned HBM) == host-buffer entry points."""
    import torch
    from roskfpos_amd import capi
    from roskfpos_amd.synth import Workload
    if generic:
        monkeypatch.setenv("KFPOS_GENERIC_KERNEL", "1")
    T, S = 1000, 12
    real = np.float32 if storage else np.float64
    w = Workload(T, A)
    r, a, dt, rt, at, et, ct = _torch_trace(w, S, real)
    host = capi.KfposBank(model, T, w.anchors, storage=storage, init_pos=w.init_positions())
    dev = capi.KfposBank(model, T, w.anchors, storage=storage, init_pos=w.init_positions())
    rep = capi.KfposBank(model, T, w.anchors, storage=storage, init_pos=w.init_positions())
    st_t = torch.zeros(T, dtype=torch.int32, device="cuda:0")
    traj_t = torch.zeros(S, 3, T, dtype=torch.float64, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    pos_hist = []
    for s in range(S):
        if model == 1:
            sh = host.step_toa_imu(r[s], w.err_est(real), a[s].astype(real), w.accel_cov(real), dt[s])
            dev.step_toa_imu_dev(rt[s], et, at[s], ct, dt[s], status=st_t, stream=stream)
        else:
            sh = host.step_toa(r[s], w.err_est(real), dt[s])
            dev.step_toa_dev(rt[s], et, dt[s], status=st_t, stream=stream)
        pos_hist.append(host.get_pose(0.0)[0])
    torch.cuda.synchronize()
    assert np.array_equal(sh, st_t.cpu().numpy().astype(np.uint32))
    if model == 1:
        rep.run_trace_dev(S, rt, A * T, et, 0, dt, accel=at, stride_accel=3 * T, cov=ct, stride_cov=0,
                          trajectory=traj_t, status=st_t, stream=stream)
    else:
        rep.run_trace_dev(S, rt, A * T, et, 0, dt, trajectory=traj_t, status=st_t, stream=stream)
    torch.cuda.synchronize()
    # one multi-epoch launch == S single-epoch launches: last status, every intermediate pose, final state
    assert np.array_equal(sh, st_t.cpu().numpy().astype(np.uint32))
    assert np.array_equal(traj_t.cpu().numpy().transpose(0, 2, 1), np.stack(pos_hist))
    xh, Ph, _ = host.get_state()
    for other in (dev, rep):
        xo, Po, _ = other.get_state()
        assert np.array_equal(xh, xo) and np.array_equal(Ph, Po)
        if model == 1:  # the sample left latched is the LAST epoch's (lastImuMeasurement, KalmanFilterTOAIMU.cpp:78-89)
            assert np.array_equal(host.get_latch(), other.get_latch())
    # pose straight into torch memory, component-major
    pos_t = torch.zeros(3, T, dtype=torch.float64, device="cuda:0")
    dev.get_pose_dev(0.05, pos=pos_t, stream=stream)
    torch.cuda.synchronize()
    ph, _, _, _ = host.get_pose(0.05)
    assert np.array_equal(pos_t.cpu().numpy().T, ph)


def test_full_size_properties_config3():
    """BASELINE config 3 at full size (65 536 tags x 8 anchors, 9-state, f32 measurements, f64 covariance):
    size-independent properties plus an oracle check on a strided sample of the tags."""
    import torch
    from roskfpos_amd import capi
    from roskfpos_amd.synth import Workload
    import oracle_py
    T, A, S = 65536, 8, 10
    w = Workload(T, A)
    r, a, dt, rt, at, et, ct = _torch_trace(w, S, np.float32)
    stream = torch.cuda.current_stream().cuda_stream
    full = capi.KfposBank(capi.MODEL_TOA_IMU, T, w.anchors, storage=2, init_pos=w.init_positions())
    full.run_trace_dev(S, rt, A * T, et, 0, dt, accel=at, stride_accel=3 * T, cov=ct, stride_cov=0, stream=stream)
    torch.cuda.synchronize()
    xf, Pf, _ = full.get_state()
    assert np.all(np.isfinite(xf)) and np.all(np.isfinite(Pf))
    # (1) shard equivalence: tags are independent, so any shard run alone is bit-identical
    lo, hi = 3 * 8192, 4 * 8192
    ws = Workload(hi - lo, A, tag0=lo)
    shard = capi.KfposBank(capi.MODEL_TOA_IMU, hi - lo, w.anchors, storage=2, init_pos=ws.init_positions())
    for s in range(S):
        shard.step_toa_imu(ws.ranges_mm(s), ws.err_est(np.float32), ws.accel(s, np.float32),
                           ws.accel_cov(np.float32), dt[s])
    xs, Ps, _ = shard.get_state()
    assert np.array_equal(xs, xf[lo:hi]) and np.array_equal(Ps, Pf[lo:hi])
    # (2) covariance stays symmetric positive semi-definite on its position block, and tracks the truth
    truth = w.position(w.time_of(S - 1))
    assert np.sqrt(((xf[:, :3] - truth) ** 2).sum(1).mean()) < 0.5
    assert np.all(np.linalg.eigvalsh(Pf[::997, :3, :3]) > -1e-9)
    # (3) oracle on every 257th tag
    idx = np.arange(0, T, 257)
    orc = oracle_py.OracleBank(1, len(idx), w.anchors, init_pos=w.init_positions()[idx], n_threads=8)
    e64 = w.err_est(np.float32).astype(np.float64)[idx]
    c64 = w.accel_cov(np.float32).astype(np.float64)[idx]
    for s in range(S):
        orc.step_imu(a[s].astype(np.float32).astype(np.float64)[idx], c64, 0.0)
        orc.step_toa(r[s][idx], e64, dt[s])
    xo, _ = orc.get_state()
    rms = np.sqrt(((xo[:, :3] - xf[idx, :3]) ** 2).sum(1).mean())
    assert rms <= RMS_BAR, rms


def test_diagonal_covariance_fast_path_is_bit_identical():
    """The 9-state iteration has a wave-uniform fast path for diagonal accelerometer covariances. A tag's result must
    not depend on its wave-mates: one tag with a full covariance sends the whole wavefront down the general path, and
    the other 63 must come out bit for bit as they do on the fast path."""
    from roskfpos_amd import capi
    from roskfpos_amd.synth import Workload
    T, S = 64, 25
    w = Workload(T, 8)
    cov_diag = w.accel_cov()
    cov_diag[:, 0], cov_diag[:, 4], cov_diag[:, 8] = 0.010, 0.012, 0.009  # diagonal, unequal
    cov_mixed = cov_diag.copy()
    cov_mixed[5] = np.array([[0.010, 0.002, -0.001], [0.002, 0.012, 0.003], [-0.001, 0.003, 0.009]]).ravel()
    out = []
    for cov in (cov_diag, cov_mixed):
        b = capi.KfposBank(capi.MODEL_TOA_IMU, T, w.anchors, init_pos=w.init_positions())
        for s in range(S):
            b.step_toa_imu(w.ranges_mm(s), w.err_est(), w.accel(s), cov, w.dt_of(s))
        out.append(b.get_state()[:2])
    keep = np.arange(T) != 5
    np.testing.assert_array_equal(out[0][0][keep], out[1][0][keep])
    np.testing.assert_array_equal(out[0][1][keep], out[1][1][keep])
    assert not np.array_equal(out[0][0][5], out[1][0][5])


@pytest.mark.parametrize("name,model,T,A,storage,top_n", [
    ("config2", 0, 4096, 8, 0, 0),          # 4 096 tags x 8 anchors, 6-state, fp64
    ("config4_shard", 0, 131072, 8, 0, 0),  # one GPU's share of 1 048 576 tags, 6-state
    ("config5", 0, 262144, 16, 1, 2),       # 262 144 tags x 16 anchors, top-N 2, f32 storage
])
def test_full_size_properties_six_state_configs(name, model, T, A, storage, top_n):
    """The other BASELINE configurations at their full sizes: finite state, shard equivalence (a slice run alone is
    bit-identical), and the oracle on a strided sample of the tags."""
    import torch
    from roskfpos_amd import capi
    from roskfpos_amd.synth import Workload
    import oracle_py
    S = 8
    real = np.float64 if storage == 0 else np.float32
    w = Workload(T, A)
    r = np.stack([w.ranges_mm(s) for s in range(S)])
    if top_n:
        r[:, ::5, 3] += 800  # NLOS-like bias for the ranking to react to
    dt = np.array([w.dt_of(s) for s in range(S)])
    rt = torch.from_numpy(np.ascontiguousarray(r.transpose(0, 2, 1))).to("cuda:0")
    et = torch.from_numpy(np.ascontiguousarray(w.err_est(real).T)).to("cuda:0")
    full = capi.KfposBank(model, T, w.anchors, storage=storage, top_n=top_n, init_pos=w.init_positions())
    full.run_trace_dev(S, rt, A * T, et, 0, dt, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    xf, Pf, _ = full.get_state()
    assert np.all(np.isfinite(xf)) and np.all(np.isfinite(Pf))
    lo, hi = T // 2, T // 2 + 1000  # not wave-aligned on purpose
    ws = Workload(hi - lo, A, tag0=lo)
    # bit-identity holds between banks served by the same kernel family: keep the 1000-tag slice off the
    # 8-lanes-per-tag kernel (banks <= 8 192 tags) when the full bank is too large for it
    old_env = os.environ.get("KFPOS_NO_COOP")
    if T > 8192:
        os.environ["KFPOS_NO_COOP"] = "1"
    try:
        shard = capi.KfposBank(model, hi - lo, w.anchors, storage=storage, top_n=top_n, init_pos=ws.init_positions())
    finally:
        if old_env is None:
            os.environ.pop("KFPOS_NO_COOP", None)
        else:
            os.environ["KFPOS_NO_COOP"] = old_env
    for s in range(S):
        shard.step_toa(r[s][lo:hi], ws.err_est(real), dt[s])
    xs, Ps, _ = shard.get_state()
    assert np.array_equal(xs, xf[lo:hi]) and np.array_equal(Ps, Pf[lo:hi])
    idx = np.arange(0, T, max(1, T // 200))
    orc = oracle_py.OracleBank(model, len(idx), w.anchors, top_n=top_n, init_pos=w.init_positions()[idx], n_threads=8)
    e64 = w.err_est(real).astype(np.float64)[idx]
    for s in range(S):
        orc.step_toa(r[s][idx], e64, dt[s])
    xo, _ = orc.get_state()
    rms = np.sqrt(((xo[:, :3] - xf[idx, :3]) ** 2).sum(1).mean())
    assert rms <= RMS_BAR, rms


COOP_CASES = ["toa6_A8_fixed", "toa6_A4_fixed", "toa6_A5_generic", "toa6_A8_zero_err"]


@pytest.mark.parametrize("name", COOP_CASES)
@pytest.mark.parametrize("storage", [0, 2])
def test_cooperative_small_batch_kernel_matches_oracle(name, storage):
    """Small plain 6-state banks run one tag per 8 lanes (k_step_toa6_coop: anchor sweeps spread over the lanes,
    partial sums combined by DPP). Same parity bar and status words as the one-tag-per-lane kernels; not bit-identical
    to them (another summation order)."""
    case = CASE_BY_NAME[name]
    real = np.float64 if storage == 0 else np.float32
    fo, po, so = drive(case, OracleImpl, real=real, record=True)
    fg, pg, sg = drive(case, lambda c, w, init: GpuImpl(c, w, init, storage=storage, coop=True), real=real, record=True)
    rms, mx, same_nan = rms_and_max(pg, po)
    assert same_nan
    assert rms <= 1e-9 and mx <= 1e-8, (rms, mx)
    assert np.array_equal(so, sg)
    xo, Po = fo.state()
    xg, Pg = fg.state()
    ok = np.isfinite(xo[:, 0])
    assert np.abs(Po[ok] - Pg[ok]).max() <= 1e-9 * np.abs(Po[ok]).max()


def test_cooperative_kernel_fused_trace_and_partial_groups():
    """T not a multiple of 8 (a partial last wavefront of groups), multi-epoch launch = per-epoch launches, dt < 0 skips."""
    import torch
    from roskfpos_amd import capi
    from roskfpos_amd.synth import Workload
    T, A, S = 1003, 8, 30
    w = Workload(T, A)
    seq = capi.KfposBank(capi.MODEL_TOA, T, w.anchors, init_pos=w.init_positions())
    rep = capi.KfposBank(capi.MODEL_TOA, T, w.anchors, init_pos=w.init_positions())
    r = np.stack([w.ranges_mm(s) for s in range(S)])
    dt = np.array([w.dt_of(s) for s in range(S)])
    for s in range(S):
        seq.step_toa(r[s], w.err_est(), dt[s])
    rt = torch.from_numpy(np.ascontiguousarray(r.transpose(0, 2, 1))).to("cuda:0")
    et = torch.from_numpy(np.ascontiguousarray(w.err_est().T)).to("cuda:0")
    traj = torch.zeros(S, 3, T, dtype=torch.float64, device="cuda:0")
    rep.run_trace_dev(S, rt, A * T, et, 0, dt, trajectory=traj, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    xs, Ps, _ = seq.get_state()
    xr, Pr, _ = rep.get_state()
    np.testing.assert_array_equal(xr, xs)
    np.testing.assert_array_equal(Pr, Ps)
    np.testing.assert_array_equal(traj[-1].cpu().numpy().T, xs[:, :3])
    d = np.full(T, 0.05)
    d[::4] = -1.0
    st = seq.step_toa(w.ranges_mm(S), w.err_est(), d)
    xa, _, _ = seq.get_state()
    assert np.all(st[::4] == 64) and np.array_equal(xa[::4], xs[::4]) and np.all(xa[1::4, :3] != xs[1::4, :3])

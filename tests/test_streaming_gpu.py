"""Host-side paths of the C ABI that do not change what is computed, only how epochs reach the GPU:
  - the streaming slot API (kfpos_slot_*: pinned component-major slots, copy / compute / return streams),
  - the mapped-block path of small banks (no staging copies) against the staged path,
  - anchor tables smaller than max_anchors.
Each is compared bit for bit with the synchronous host-buffer API (itself checked against the oracle in
test_gpu_parity.py) and once against the oracle directly.
"""
import os

import numpy as np
import pytest

from conftest import has_gpu
from roskfpos_amd.synth import Workload

pytestmark = pytest.mark.gpu


def _bank(model, T, w, storage, **kw):
    from roskfpos_amd import capi
    return capi.KfposBank(model, T, w.anchors, storage=storage, init_pos=w.init_positions(), **kw)


def _epoch(w, s):
    r = w.ranges_mm(s)
    if s % 7 == 3:
        r[:, 1] = -1
    if s % 11 == 5:
        r[::3, 2:] = 0
    return r


@pytest.mark.parametrize("model,storage", [(1, 2), (1, 0), (0, 0), (0, 1)])
def test_slot_api_equals_synchronous_api(model, storage):
    if not has_gpu():
        pytest.skip("no GPU")
    from roskfpos_amd import capi
    T, A, S = 3000, 8, 24          # > 4096 range elements: the synchronous bank takes the staged path
    w = Workload(T, A)
    real = np.float64 if storage == capi.STORE_F64 else np.float32
    err, cov = w.err_est(real), w.accel_cov(real)
    sync = _bank(model, T, w, storage)
    strm = _bank(model, T, w, storage)
    NS = strm.lib.kfpos_slot_count(strm._h)
    assert NS >= 2
    pos_sync, st_sync, pos_slot, st_slot = [], [], [], []
    pending = []

    def collect(slot, views):
        strm.slot_wait(slot)
        pos_slot.append(views["pos"].T.copy())
        st_slot.append(views["status"].copy())

    for s in range(S):
        r, dt = _epoch(w, s), w.dt_of(s)
        per_tag = s % 5 == 4       # tags report asynchronously: negative dt = no epoch for that tag
        dts = np.full(T, dt)
        if per_tag:
            dts[(np.arange(T) + s) % 6 == 1] = -1.0
        a = w.accel(s, real)
        if model == 1:
            st = sync.step_toa_imu(r, err, a, cov, dts if per_tag else dt)
        else:
            st = sync.step_toa(r, err, dts if per_tag else dt)
        st_sync.append(st)
        pos_sync.append(sync.get_pose(0.0)[0])
        slot = s % NS
        v = strm.slot_acquire(slot)    # waits for the submission of epoch s - NS
        v["range_mm"][:] = r.T
        flags = capi.SLOT_TOA_IMU if model == 1 else capi.SLOT_TOA
        if s == 0 or s % 4 == 0:
            v["err_est"][:] = err.T
        else:
            flags |= capi.SLOT_REUSE_ERR
        if model == 1:
            v["accel"][:] = a.T
            if s < 2 or s % 3 == 0:
                v["cov"][:] = cov.T
            else:
                flags |= capi.SLOT_REUSE_COV
        if per_tag:
            v["dt"][:] = dts
            flags |= capi.SLOT_DT_PER_TAG
        strm.slot_submit(slot, flags, dt)
        pending.append((slot, v))
        if len(pending) == NS:         # keep NS - 1 submissions in flight behind the one being collected
            collect(*pending.pop(0))
    while pending:
        collect(*pending.pop(0))
    assert len(pos_slot) == S
    for s in range(S):
        assert np.array_equal(st_slot[s], st_sync[s]), f"status words, epoch {s}"
        assert np.array_equal(pos_slot[s], pos_sync[s], equal_nan=True), f"poses, epoch {s}"
    xs, Ps, fs = sync.get_state()
    xt, Pt, ft = strm.get_state()      # waits for the slots by itself
    assert np.array_equal(xs, xt) and np.array_equal(Ps, Pt) and np.array_equal(fs, ft)
    sync.close()
    strm.close()


def test_slot_api_against_oracle_and_misuse():
    if not has_gpu():
        pytest.skip("no GPU")
    import oracle_py
    from roskfpos_amd import capi
    T, A, S = 2000, 8, 30
    w = Workload(T, A)
    b = _bank(1, T, w, capi.STORE_F64)
    o = oracle_py.OracleBank(1, T, w.anchors, init_pos=w.init_positions(), n_threads=4)
    err, cov = w.err_est(), w.accel_cov()
    with pytest.raises(capi.KfposError):   # nothing acquired yet
        b.slot_submit(0, capi.SLOT_TOA)
    v = b.slot_acquire(0)
    with pytest.raises(capi.KfposError):   # reuse before anything was uploaded
        b.slot_submit(0, capi.SLOT_TOA | capi.SLOT_REUSE_ERR, 0.1)
    for s in range(S):
        slot = s & 1
        v = b.slot_acquire(slot)
        v["range_mm"][:] = _epoch(w, s).T
        v["err_est"][:] = err.T
        v["accel"][:] = w.accel(s).T
        v["cov"][:] = cov.T
        b.slot_submit(slot, capi.SLOT_TOA_IMU | capi.SLOT_NO_POSE, w.dt_of(s))
        o.step_imu(w.accel(s), cov, 0.0)
        o.step_toa(_epoch(w, s), err, w.dt_of(s))
    x, _, _ = b.get_state()
    xo, _ = o.get_state()
    rms = float(np.sqrt(((x[:, :3] - xo[:, :3]) ** 2).sum(1).mean()))
    assert rms <= 1e-6, rms
    b.close()


@pytest.mark.parametrize("model", [0, 1])
def test_slot_upload_waits_for_readers_of_an_older_upload(model):
    """Slot 0 uploads errorEstimations (and the sensor covariance), slot 1 re-uses them, slot 2 uploads new ones, then
    slot 0 is filled and uploaded again while slot 1's kernel -- which reads slot 0's device block -- may still be
    queued: the upload has to wait for that reader although slot 0 no longer holds the current values (round 2 only
    guarded the slot that did). Values differ from upload to upload, so a torn read changes the result."""
    if not has_gpu():
        pytest.skip("no GPU")
    from roskfpos_amd import capi
    T, A, S = 65536, 8, 18
    w = Workload(T, A)
    storage = capi.STORE_MIXED if model == 1 else capi.STORE_F32
    real = np.float32
    sync, strm = _bank(model, T, w, storage), _bank(model, T, w, storage)
    NS = strm.lib.kfpos_slot_count(strm._h)
    assert NS == 3
    err0, cov0 = w.err_est(real), w.accel_cov(real)
    cur_err = cur_cov = None
    views = {}
    for s in range(S):
        r, dt, a = _epoch(w, s), w.dt_of(s), w.accel(s, real)
        upload = s % 3 != 1                 # s = 0 up (slot 0), 1 reuse (slot 1), 2 up (slot 2), 3 up (slot 0), ...
        if upload:
            cur_err = (err0 * real(1.0 + 0.25 * (s % 5))).astype(real)
            cur_cov = (cov0 * real(1.0 + 0.5 * (s % 4))).astype(real)
        if model == 1:
            sync.step_toa_imu(r, cur_err, a, cur_cov, dt)
        else:
            sync.step_toa(r, cur_err, dt)
        slot = s % NS
        v = views[slot] = strm.slot_acquire(slot)
        v["range_mm"][:] = r.T
        flags = (capi.SLOT_TOA_IMU if model == 1 else capi.SLOT_TOA) | capi.SLOT_NO_POSE
        if model == 1:
            v["accel"][:] = a.T
        if upload:
            v["err_est"][:] = cur_err.T
            if model == 1:
                v["cov"][:] = cur_cov.T
        else:
            flags |= capi.SLOT_REUSE_ERR | (capi.SLOT_REUSE_COV if model == 1 else 0)
        strm.slot_submit(slot, flags, dt)
    xs, Ps, _ = sync.get_state()
    xt, Pt, _ = strm.get_state()
    assert np.array_equal(xs, xt) and np.array_equal(Ps, Pt)
    sync.close()
    strm.close()


@pytest.mark.parametrize("model,A", [(0, 8), (0, 16), (1, 8), (0, 5)])
def test_small_bank_mapped_block_equals_staged_path(model, A):
    if not has_gpu():
        pytest.skip("no GPU")
    from roskfpos_amd import capi
    T, S = 48, 20
    w = Workload(T, A)
    err, cov = w.err_est(), w.accel_cov()

    def run(limit):
        old = os.environ.get("KFPOS_SMALL_BANK_ELEMS")
        os.environ["KFPOS_SMALL_BANK_ELEMS"] = str(limit)   # read at kfpos_create
        try:
            b = _bank(model, T, w, capi.STORE_F64)
        finally:
            if old is None:
                os.environ.pop("KFPOS_SMALL_BANK_ELEMS")
            else:
                os.environ["KFPOS_SMALL_BANK_ELEMS"] = old
        out = []
        for s in range(S):
            dts = np.full(T, w.dt_of(s))
            if s % 4 == 3:
                dts[::5] = -1.0
            if model == 1 and s % 3 != 2:
                st = b.step_toa_imu(_epoch(w, s), err, w.accel(s), cov, dts if s % 4 == 3 else w.dt_of(s))
            elif model == 1 and s % 3 == 2:
                b.step_imu(w.accel(s), cov, 0.4 * w.dt_of(s))
                st = b.step_toa(_epoch(w, s), err, 0.6 * w.dt_of(s))
            else:
                st = b.step_toa(_epoch(w, s), err, dts if s % 4 == 3 else w.dt_of(s))
            pos, c3, vel, pst = b.get_pose(0.02)
            xp, Pp, _ = b.get_predicted(np.full(T, 0.03) if s % 2 else 0.03)
            pe = b.get_pose_each(np.linspace(0.0, 0.05, T))[0]
            out.append((st, pos, c3, vel, pst, xp, Pp, pe))
        x, P, fl = b.get_state()
        b.close()
        return out, x, P, fl

    small, xs, Ps, fs = run(1 << 20)
    staged, xg, Pg, fg = run(0)
    assert np.array_equal(xs, xg) and np.array_equal(Ps, Pg) and np.array_equal(fs, fg)
    for s in range(S):
        for a, b in zip(small[s], staged[s]):
            assert np.array_equal(a, b, equal_nan=True), f"epoch {s}"


@pytest.mark.parametrize("model", [0, 1])
def test_fewer_anchors_than_capacity(model):
    """A handle created for up to 12 anchors with 9 set computes what a 9-anchor handle computes: columns beyond the
    anchors that are set do not exist (the adaptor objects are created with max_anchors = 64). 9 and 12: both banks run
    the run-time-loop kernels, so the comparison can be bit for bit."""
    if not has_gpu():
        pytest.skip("no GPU")
    from roskfpos_amd import capi
    T, S, n = 200, 15, 9
    w = Workload(T, n)
    err, cov = w.err_est(), w.accel_cov()
    exact = capi.KfposBank(model, T, w.anchors, init_pos=w.init_positions())
    roomy = capi.KfposBank(model, T, w.anchors, init_pos=w.init_positions(), max_anchors=12)
    pad_r = np.zeros((T, 12), dtype=np.int32)
    pad_e = np.ones((T, 12))
    for s in range(S):
        r = _epoch(w, s)
        pad_r[:, :n] = r
        pad_r[:, n:] = 4321          # whatever sits in the unused columns is never read
        pad_e[:, :n] = err
        if model == 1:
            st1 = exact.step_toa_imu(r, err, w.accel(s), cov, w.dt_of(s))
            st2 = roomy.step_toa_imu(pad_r, pad_e, w.accel(s), cov, w.dt_of(s))
        else:
            st1 = exact.step_toa(r, err, w.dt_of(s))
            st2 = roomy.step_toa(pad_r, pad_e, w.dt_of(s))
        assert np.array_equal(st1, st2)
    x1, P1, _ = exact.get_state()
    x2, P2, _ = roomy.get_state()
    assert np.array_equal(x1, x2) and np.array_equal(P1, P2)
    exact.close()
    roomy.close()

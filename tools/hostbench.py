#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-facing entry points (never bench.py's `value`), BASELINE configs[2] shape
(65 536 tags x 8 anchors, 9-state, f32/int32 measurements):

  sync      kfpos_step_toa_imu (+ kfpos_get_pose) per epoch with row-major numpy arrays in pageable memory: what the
            adaptor classes and a naive multi-tag node use
  slots     the streaming API (kfpos_slot_*): epochs already assembled in the two pinned, component-major slots (a
            front-end that writes ranges where they are wanted), submit / wait pipelined over the slots; poses and status words
            come back with every epoch
  slots+fill  the same, with a CPU producer that copies each epoch's 7.3 MB into the slot before submitting (numpy
            memcpy: the producer, not the library, is the bound here)
  reuse     `slots` with KFPOS_SLOT_REUSE_ERR | KFPOS_SLOT_REUSE_COV (constant errorEstimation / sensor covariance)

    python tools/hostbench.py [--tags 65536] [--steps 60]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roskfpos_amd import capi  # noqa: E402
from roskfpos_amd.synth import Workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tags", type=int, default=65536)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--model", type=int, default=1)
ap.add_argument("--modes", default="sync,slots,reuse,nopose,fill,fillreuse")
a = ap.parse_args()
MODES = set(a.modes.split(","))
T, A, S = a.tags, 8, a.steps
w = Workload(T, A)
imu = a.model == 1
err, cov = w.err_est(np.float32), w.accel_cov(np.float32)
NPRE = 8
inputs = [(w.ranges_mm(s), w.accel(s, np.float32), w.dt_of(s)) for s in range(NPRE)]
in_bytes = T * (A * 4 + A * 4 + (12 + 36 if imu else 0))


def new_bank():
    return capi.KfposBank(a.model, T, w.anchors, storage=capi.STORE_MIXED, init_pos=w.init_positions())


out = {"tags": T, "model": a.model, "input_MB_per_epoch": in_bytes / 1e6}

# ---- synchronous API, pageable row-major arrays ----
bank = new_bank()
S_sync = S if "sync" in MODES else 1
for r, ac, dt in inputs[:3]:
    bank.step_toa_imu(r, err, ac, cov, dt) if imu else bank.step_toa(r, err, dt)
t_step = t_pose = 0.0
for s in range(S_sync):
    r, ac, dt = inputs[s % NPRE]
    t0 = time.perf_counter()
    bank.step_toa_imu(r, err, ac, cov, 0.05) if imu else bank.step_toa(r, err, 0.05)
    t1 = time.perf_counter()
    bank.get_pose(0.0)
    t2 = time.perf_counter()
    t_step += t1 - t0
    t_pose += t2 - t1
bank.close()
if "sync" in MODES:
    out["sync"] = {"ms_per_step_call": t_step / S * 1e3, "ms_per_get_pose_call": t_pose / S * 1e3,
                   "tag_steps_per_s_step_only": T * S / t_step,
                   "tag_steps_per_s_step_plus_pose": T * S / (t_step + t_pose)}

# ---- streaming slots ----
cm = [(np.ascontiguousarray(r.T), np.ascontiguousarray(ac.T)) for r, ac, _ in inputs]
err_cm, cov_cm = np.ascontiguousarray(err.T), np.ascontiguousarray(cov.T)


def stream(fill, reuse, want_pose=True):
    bank = new_bank()
    kind = capi.SLOT_TOA_IMU if imu else capi.SLOT_TOA
    NS = bank.lib.kfpos_slot_count(bank._h)
    views = [bank.slot_acquire(k) for k in range(NS)]
    for k, v in enumerate(views):   # every slot holds a complete epoch
        v["range_mm"][:] = cm[k][0]
        v["err_est"][:] = err_cm
        v["accel"][:] = cm[k][1]
        v["cov"][:] = cov_cm
    flags = kind | (0 if want_pose else capi.SLOT_NO_POSE)
    for s in range(2 * NS):         # warm-up, uploads err / cov at least once
        bank.slot_acquire(s % NS)
        bank.slot_submit(s % NS, flags, 0.1 if s == 0 else 0.05)
    for k in range(NS):
        bank.slot_wait(k)
    if reuse:
        flags |= capi.SLOT_REUSE_ERR | capi.SLOT_REUSE_COV
    checksum = 0.0
    t0 = time.perf_counter()
    for s in range(S):
        k = s % NS
        v = bank.slot_acquire(k)    # epoch s - NS is complete: its poses / status words are in the slot
        if want_pose and s >= NS:
            checksum += float(v["pos"][0, 0])
        if fill:
            v["range_mm"][:] = cm[s % NPRE][0]
            v["accel"][:] = cm[s % NPRE][1]
            if not reuse:
                v["err_est"][:] = err_cm
                v["cov"][:] = cov_cm
        bank.slot_submit(k, flags, 0.05)
    for k in range(NS):
        bank.slot_wait(k)
    el = time.perf_counter() - t0
    x, _, _ = bank.get_state()
    bank.close()
    assert np.isfinite(x).all() and np.isfinite(checksum)
    return {"ms_per_epoch": el / S * 1e3, "tag_steps_per_s": T * S / el}


if "slots" in MODES:
    out["slots"] = stream(fill=False, reuse=False)
if "reuse" in MODES:
    out["slots_reuse_err_cov"] = stream(fill=False, reuse=True)
if "nopose" in MODES:
    out["slots_no_pose"] = stream(fill=False, reuse=True, want_pose=False)
if "fill" in MODES:
    out["slots_with_cpu_fill"] = stream(fill=True, reuse=False)
if "fillreuse" in MODES:
    out["slots_with_cpu_fill_reuse"] = stream(fill=True, reuse=True)
out["env"] = {k: os.environ[k] for k in ("HSA_ENABLE_SDMA", "GPU_FORCE_BLIT_COPY_SIZE", "HIP_FORCE_DEV_KERNARG") if k in os.environ}
print(json.dumps(out))

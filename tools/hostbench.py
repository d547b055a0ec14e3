#!/usr/bin/env python3
"""PCIe-inclusive rate of the synchronous host-buffer API (what the adaptor / ingest node use): one
kfpos_step_toa_imu + one kfpos_get_pose per epoch with numpy arrays in pageable host memory, BASELINE configs[2].

    python tools/hostbench.py [--tags 65536] [--steps 30]
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
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--model", type=int, default=1)
a = ap.parse_args()
T, A = a.tags, 8
w = Workload(T, A)
bank = capi.KfposBank(a.model, T, w.anchors, storage=capi.STORE_MIXED, init_pos=w.init_positions())
err, cov = w.err_est(np.float32), w.accel_cov(np.float32)
inputs = [(w.ranges_mm(s), w.accel(s, np.float32), w.dt_of(s)) for s in range(a.steps + 3)]
for r, ac, dt in inputs[:3]:
    bank.step_toa_imu(r, err, ac, cov, dt) if a.model == 1 else bank.step_toa(r, err, dt)
t_step = t_pose = 0.0
for r, ac, dt in inputs[3:]:
    t0 = time.perf_counter()
    bank.step_toa_imu(r, err, ac, cov, dt) if a.model == 1 else bank.step_toa(r, err, dt)
    t1 = time.perf_counter()
    bank.get_pose(0.0)
    t2 = time.perf_counter()
    t_step += t1 - t0
    t_pose += t2 - t1
in_bytes = T * (A * 4 + A * 4 + (12 + 36 if a.model == 1 else 0)) + 8
print(json.dumps({"tags": T, "model": a.model, "ms_per_step_call": t_step / a.steps * 1e3,
                  "ms_per_get_pose_call": t_pose / a.steps * 1e3,
                  "tag_steps_per_s_step_only": T * a.steps / t_step,
                  "tag_steps_per_s_step_plus_pose": T * a.steps / (t_step + t_pose),
                  "input_MB_per_step": in_bytes / 1e6}))

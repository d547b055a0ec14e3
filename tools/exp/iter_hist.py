#!/usr/bin/env python3
"""Iteration counts of the 9-state IEKF on the BASELINE configs[2] trace, per tag and per wavefront: the histogram, and
for survivor thresholds 8 / 16 / 32 the iteration at which a wavefront (64 consecutive tags) is down to that many
iterating lanes -- the input of the model in DESIGN section 6a."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from roskfpos_amd import capi
from roskfpos_amd.synth import Workload
T, S = 65536, 45
w = Workload(T, 8)
b = capi.KfposBank(capi.MODEL_TOA_IMU, T, w.anchors, storage=capi.STORE_MIXED, init_pos=w.init_positions())
err, cov = w.err_est(np.float32), w.accel_cov(np.float32)
its = []
for s in range(S):
    st = b.step_toa_imu(w.ranges_mm(s), err, w.accel(s, np.float32), cov, w.dt_of(s))
    if s >= 25:
        its.append(((st >> 8) & 0xFF).astype(np.int32))
its = np.array(its)                      # [epochs][tags] gain iterations (solves done); the loop body runs its+1 times, capped
out = {"tags": T, "epochs": int(its.shape[0]), "hist": np.bincount(its.ravel(), minlength=21).tolist(),
       "mean": float(its.mean())}
wv = its.reshape(its.shape[0], T // 64, 64)
out["mean_wave_max"] = float(wv.max(2).mean())
active = np.stack([(wv > k).sum(2) for k in range(21)], 0)     # [k][epoch][wave] lanes that still run after k solves
out["mean_active_after_k"] = [float(active[k].mean()) for k in range(21)]
for L in (8, 16, 32):
    first = (active <= L).argmax(0)      # first k with <= L lanes running
    out[f"switch_iteration_L{L}_hist"] = np.bincount(first.ravel(), minlength=21).tolist()
    never = ((active <= L).sum(0) == 0).mean()
    out[f"never_L{L}"] = float(never)
np.save("gpurun_out/iter_counts.npy", its.astype(np.int8))
print(json.dumps(out))

#!/usr/bin/env python3
"""How long does a 9-state tag stay 'slow'? P(IEKF at its cap of 20 in epoch e + lag | at the cap in epoch e) on the
BASELINE configs[2] trace -- what any scheme that groups slow tags (sorting, re-packing) has to live with."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from roskfpos_amd import capi
from roskfpos_amd.synth import Workload
T, S = 16384, 80
w = Workload(T, 8)
b = capi.KfposBank(capi.MODEL_TOA_IMU, T, w.anchors, storage=capi.STORE_MIXED, init_pos=w.init_positions())
err, cov = w.err_est(np.float32), w.accel_cov(np.float32)
caps = []
for s in range(S):
    st = b.step_toa_imu(w.ranges_mm(s), err, w.accel(s, np.float32), cov, w.dt_of(s))
    caps.append(((st >> 8) & 0xFF) >= 20)
caps = np.array(caps[20:])
out = {"tags": T, "fraction_at_cap": float(caps.mean())}
for lag in (1, 2, 3, 5, 10, 20, 40):
    a, c = caps[:-lag], caps[lag:]
    out[f"P(cap at e+{lag} | cap at e)"] = float((a & c).sum() / max(a.sum(), 1))
print(json.dumps(out))

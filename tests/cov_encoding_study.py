#!/usr/bin/env python3
"""Study (not a test): does any covariance encoding of at most 6 bytes per entry bring BASELINE
configs[2] (9-state filter, "fp32" storage) under the 1e-6 m bar? The kernel body (host emulation) runs the BASELINE
trace with the covariance passed through each encoding between epochs, against the oracle.
    python tests/cov_encoding_study.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import impls  # noqa: E402
from cases import Case  # noqa: E402
from roskfpos_amd.synth import Workload  # noqa: E402

T, S = 2048, 100
case = Case("baseline_c3", 1, 8, T=T, S=S)
w = Workload(T, 8)
err, cov = w.err_est(np.float32).astype(np.float64), w.accel_cov(np.float32).astype(np.float64)
lib = impls.emu_lib()
lib.kfe_round_storage.argtypes = [C.c_void_p, C.c_int]
orc = impls.OracleImpl(case, w, w.init_positions())
for s in range(S):
    orc.fused(w.ranges_mm(s), err, w.accel(s, np.float32).astype(np.float64), cov, w.dt_of(s))
po = orc.positions()
out = {}
for name, what in (("f64 (8 bytes)", 0), ("f32 (4 bytes)", 1), ("f32 + bf16 residual (6 bytes)", 4),
                   ("KFPOS_STORE_P48: single exponent + 39 mantissa bits (6 bytes; round 2 studied the upper 48 bits of the double here)", 8)):
    f = impls.EmuStaticImpl(case, w, w.init_positions())
    for s in range(S):
        f.fused(w.ranges_mm(s), err, w.accel(s, np.float32).astype(np.float64), cov, w.dt_of(s))
        if what:
            lib.kfe_round_storage(f.h, what)
    d = f.positions() - po
    out[name] = {"rms_m": float(np.sqrt((d ** 2).sum(1).mean())), "max_m": float(np.abs(d).max())}
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""When a 9-state tag runs the IEKF to its cap of 20, is it because the iteration has entered an exact cycle? The host
emulation of the kernel body (built with tools/exp/iter_trace.h) records, for every capped step on the BASELINE
configs[2] trace, the first iteration whose iterate equals an earlier one bit for bit, and the period.
    g++ ... -DKFPOS_EMU_ITER_TRACE -include tools/exp/iter_trace.h -o tools/exp/_build/libkfpos_emu_trace.so tools/exp/iter_cycle_emu.cpp
    python tools/exp/iter_cycle.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KFPOS_EMU_LIB"] = os.path.join(ROOT, "tools", "exp", "_build", "libkfpos_emu_trace.so")
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import impls  # noqa: E402
from cases import Case  # noqa: E402
from roskfpos_amd.synth import Workload  # noqa: E402


class Trace(C.Structure):
    _fields_ = [("n", C.c_int), ("first", C.c_int), ("period", C.c_int), ("ve", C.c_double * 6 * 21),
                ("c", C.c_double * 21), ("capped", C.c_long), ("hist_first", C.c_long * 21),
                ("hist_period", C.c_long * 21), ("never", C.c_long), ("min_rel_gap", C.c_double * 21)]


T, S = int(os.environ.get("TAGS", 2048)), int(os.environ.get("EPOCHS", 60))
case = Case("baseline_c3", 1, 8, T=T, S=S)
w = Workload(T, 8)
err, cov = w.err_est(np.float32).astype(np.float64), w.accel_cov(np.float32).astype(np.float64)
lib = impls.emu_lib()
f = impls.EmuStaticImpl(case, w, w.init_positions())
for s in range(S):
    f.fused(w.ranges_mm(s), err, w.accel(s, np.float32).astype(np.float64), cov, w.dt_of(s))
lib.kfpos_iter_trace_done()
t = Trace.in_dll(lib, "kfpos_iter_trace_cur")
print(json.dumps({"tag_epochs": T * S, "capped": t.capped, "never_repeats_within_20": t.never,
                  "first_repeat_at_iteration": {i: t.hist_first[i] for i in range(21) if t.hist_first[i]},
                  "period": {i: t.hist_period[i] for i in range(21) if t.hist_period[i]}}))

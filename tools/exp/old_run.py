#!/usr/bin/env python3
"""Runs a library built from an OLD revision (round-1 source before the register-resident 16-anchor kernel was
dropped) through the handful of C-ABI calls that already existed then, on full and partially filled wavefronts, three
runs each, against the oracle.   usage: old_run.py LIB [A]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: F401,E402  (same libamdhip64 as the other tools)
from roskfpos_amd.capi import _Config  # noqa: E402
from roskfpos_amd.synth import Workload  # noqa: E402
import oracle_py  # noqa: E402

L = C.CDLL(sys.argv[1])
A = int(sys.argv[2]) if len(sys.argv) > 2 else 16
vp = C.c_void_p
L.kfpos_create.argtypes = [C.POINTER(_Config), C.POINTER(vp)]
L.kfpos_set_anchors.argtypes = [vp, vp, vp, C.c_int32]
L.kfpos_set_init_positions.argtypes = [vp, vp]
L.kfpos_step_toa.argtypes = [vp, vp, vp, vp, C.c_int32, vp]
L.kfpos_get_state.argtypes = [vp, vp, vp, vp]
L.kfpos_destroy.argtypes = [vp]


def gpu_run(T, w, iw, topn, storage, err, eps):
    cfg = _Config()
    cfg.model, cfg.n_tags, cfg.max_anchors, cfg.storage = 0, T, A, storage
    cfg.accel_noise = cfg.jolt = 0.5
    cfg.ignore_worst, cfg.cost_threshold, cfg.top_n, cfg.use_init_pos, cfg.device = int(iw), 0.5, topn, 1, 0
    h = vp()
    assert L.kfpos_create(C.byref(cfg), C.byref(h)) == 0
    anc = np.ascontiguousarray(w.anchors)
    assert L.kfpos_set_anchors(h, anc.ctypes.data, None, A) == 0
    ip = np.ascontiguousarray(w.init_positions())
    assert L.kfpos_set_init_positions(h, ip.ctypes.data) == 0
    st = np.zeros(T, dtype=np.uint32)
    for r, dt in eps:
        d = np.array([dt])
        assert L.kfpos_step_toa(h, r.ctypes.data, err.ctypes.data, d.ctypes.data, 1, st.ctypes.data) == 0
    x, P = np.zeros((T, 6)), np.zeros((T, 6, 6))
    assert L.kfpos_get_state(h, x.ctypes.data, P.ctypes.data, None) == 0
    L.kfpos_destroy(h)
    return x[:, :3].copy()


for T in (64, 128, 48, 100, 37):
    for iw, topn, storage in ((False, 0, 0), (False, 2, 1), (True, 0, 0), (False, 0, 1)):
        w = Workload(T, A)
        real = np.float64 if storage == 0 else np.float32
        err = np.ascontiguousarray(w.err_est(real))
        eps = []
        for s in range(25):
            r = w.ranges_mm(s)
            if s % 7 == 3: r[:, 1] = -1
            if s % 11 == 5: r[::3, 2:] = 0
            r[::5, 3] += 800
            eps.append((np.ascontiguousarray(r), w.dt_of(s)))
        runs = [gpu_run(T, w, iw, topn, storage, err, eps) for _ in range(3)]
        o = oracle_py.OracleBank(0, T, w.anchors, ignore_worst=iw, top_n=topn, init_pos=w.init_positions(), n_threads=4)
        e64 = err.astype(np.float64)
        for r, dt in eps:
            o.step_toa(r, e64, dt)
        xo = o.get_state()[0][:, :3]
        d = runs[0] - xo
        bad = np.where(~(np.abs(d).max(1) <= 1e-6))[0]
        print(json.dumps({"lib": os.path.basename(sys.argv[1]), "T": T, "A": A, "ignore_worst": iw, "top_n": topn,
                          "storage": storage, "rms_vs_oracle": float(np.sqrt(np.nansum(d ** 2, 1).mean())),
                          "n_off": int(bad.size), "tags_off": bad.tolist()[:16],
                          "runs_identical": bool(np.array_equal(runs[0], runs[1], equal_nan=True) and
                                                 np.array_equal(runs[0], runs[2], equal_nan=True))}), flush=True)

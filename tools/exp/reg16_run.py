#!/usr/bin/env python3
"""Runs the experiment libraries of reg16_build.py on the GPU: 16-anchor 6-state banks whose size is / is not a
multiple of 64, leave-one-out (the instantiation that spills to scratch) and top-N (SGPR spills only), three runs each
to expose run-to-run variation, against the oracle. One child process per library (the ctypes binding caches it)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BUILD = os.path.join(ROOT, "tools", "exp", "_build")

CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from roskfpos_amd import capi
from roskfpos_amd.synth import Workload
import oracle_py
out = []
for T, skip in ((64, 0), (100, 0), (37, 0), (1000, 0), (64, 1), (100, 1), (1000, 1)):
    for iw, topn, st in ((True, 0, capi.STORE_F64), (False, 2, capi.STORE_F32), (False, 0, capi.STORE_F64)):
        def dts(s):  # skip = 1: tags report asynchronously, a negative dt = no epoch for that tag in this call
            d = np.full(T, w.dt_of(s))
            if skip and s > 1: d[(np.arange(T) + s) %% 5 == 2] = -1.0
            return d
        w = Workload(T, 16)
        real = np.float64 if st == capi.STORE_F64 else np.float32
        err = w.err_est(real)
        runs = []
        for rep in range(3):
            b = capi.KfposBank(capi.MODEL_TOA, T, w.anchors, storage=st, ignore_worst=iw, top_n=topn, init_pos=w.init_positions())
            for s in range(25):
                r = w.ranges_mm(s)
                if s %% 7 == 3: r[:, 1] = -1
                if s %% 11 == 5: r[::3, 2:] = 0
                r[::5, 3] += 800
                b.step_toa(r, err, dts(s))
            x, P, _ = b.get_state(); b.close()
            runs.append(x[:, :3].copy())
        o = oracle_py.OracleBank(0, T, w.anchors, ignore_worst=iw, top_n=topn, init_pos=w.init_positions(), n_threads=4)
        e64 = err.astype(np.float64)
        for s in range(25):
            r = w.ranges_mm(s)
            if s %% 7 == 3: r[:, 1] = -1
            if s %% 11 == 5: r[::3, 2:] = 0
            r[::5, 3] += 800
            o.step_toa(r, e64, dts(s))
        xo = o.get_state()[0][:, :3]
        d = runs[0] - xo
        bad = np.where(np.abs(d).max(1) > 1e-6)[0]
        out.append({"T": T, "skip_lanes": skip, "ignore_worst": iw, "top_n": topn, "storage": int(st),
                    "rms_vs_oracle": float(np.sqrt((d ** 2).sum(1).mean())), "max_vs_oracle": float(np.abs(d).max()),
                    "tags_off_by_more_than_1e-6": bad.tolist()[:20], "n_off": int(bad.size),
                    "runs_identical": bool(np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])),
                    "nonfinite": int((~np.isfinite(runs[0])).sum())})
print(json.dumps(out))
"""

for name in ("early", "clamp"):
    lib = os.path.join(BUILD, f"libkfpos_reg16_{name}.so")
    env = dict(os.environ, KFPOS_LIB_PATH=lib)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True)
    if r.returncode:
        print(name, "FAILED", r.stderr[-2000:])
        continue
    for row in json.loads(r.stdout.strip().splitlines()[-1]):
        print(json.dumps(dict(variant=name, **row)))

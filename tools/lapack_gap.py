#!/usr/bin/env python3
"""Oracle (hand-written dense helpers) vs the numpy restatement on Armadillo's LAPACK drivers, per parity case and per
generation of Armadillo's dispatch (oracle/numpy_oracle.py). Writes profiles/r03_oracle_vs_lapack_gap.json.
CPU only; ~1 minute.   python tools/lapack_gap.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy_oracle as N  # noqa: E402
from cases import CASE_BY_NAME, Case  # noqa: E402
from test_lapack_flavours import CASES, gap, small  # noqa: E402

if __name__ == "__main__":
    out = {"what": "max / RMS position difference [m] between oracle/kfpos_oracle.cpp and oracle/numpy_oracle.py over "
                   "every epoch, 4-6 tags; rel_P = max |dP| / max |P| of the final covariance", "cases": {}}
    rows = [(n, small(CASE_BY_NAME[n])) for n in CASES]
    for model in (0, 1):
        c = Case(f"baseline_{'imu9' if model else 'toa6'}_100_epochs", model, 8, T=6, S=100)
        c.epoch = lambda w, s: w.ranges_mm(s)
        rows.append((c.name, c))
    for name, case in rows:
        out["cases"][name] = {}
        for fl in N.FLAVOURS:
            N.set_flavour(fl)
            rms, mx, same, rel_P = gap(case)
            out["cases"][name][fl] = {"rms_m": rms, "max_m": mx, "rel_P": rel_P, "same_nan_pattern": bool(same)}
            print(f"{name:36s} {fl:7s} rms {rms:9.2e}  max {mx:9.2e}  rel_P {rel_P:9.2e}")
    json.dump(out, open(os.path.join(ROOT, "profiles", "r03_oracle_vs_lapack_gap.json"), "w"), indent=1)

"""Regenerate tests/golden/*.npz from the CPU oracle.

PARITY UNPINNED: these vectors come from this repository's own oracle (oracle/kfpos_oracle.cpp),
not from the reference -- the reference ships no fixtures and cannot be built in this image
(Armadillo / ROS headers absent). They pin the oracle against regressions and give the GPU box a
fixed target that does not depend on the oracle library being rebuilt there.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from cases import CASES, drive  # noqa: E402
from impls import OracleImpl  # noqa: E402

STEPS = 40

if __name__ == "__main__":
    for c in CASES:
        f, pos, st = drive(c, OracleImpl, steps=STEPS, record=True)
        x, P = f.state()
        w = c.workload()
        checksum = int(sum(int(c.epoch(w, s).astype(np.int64).sum()) for s in range(STEPS)))
        np.savez_compressed(os.path.join(HERE, c.name + ".npz"), steps=STEPS, positions=pos, status=st,
                            x_final=x, P_final=P, input_checksum=checksum)
        print(c.name, pos.shape, "finite:", bool(np.isfinite(pos[-1]).all()))

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
    # 8-state planar filter: per-epoch state after every ranging epoch of the shared interleaved trace (tests/planar.py)
    from planar import CFG, PlanarOracle, run_trace
    from roskfpos_amd.synth import Workload
    for name, sensors, fixed, fixed_height in [("planar_ranging_fixed", (), True, 1),
                                               ("planar_ranging_mlinit3d", (), False, 0),
                                               ("planar_all_sensors", ("imu", "px4", "mag", "compass"), True, 1)]:
        w = Workload(24, 8)
        orc = PlanarOracle(w, dict(CFG, use_fixed_height=fixed_height), w.init_positions() if fixed else None)
        xs = []
        st = run_trace([orc], w, STEPS, sensors, collect=lambda s, impls: xs.append(impls[0].get_state()[0].copy()))[0]
        x, P = orc.get_state()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), steps=STEPS, states=np.stack(xs),
                            status=np.stack([a for a in st]), x_final=x, P_final=P, height=orc.get_height())
        print(name, np.stack(xs).shape, "calls:", len(st))

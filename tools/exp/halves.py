#!/usr/bin/env python3
"""Experiment: do two half-batches on two streams hide the per-launch state traffic of
one-launch-per-epoch callers? One 65 536-tag 9-state bank on one stream against two 32 768-tag banks on two streams
(no join between epochs: each half runs ahead on its own stream), K epochs each, wall clock around the lot."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from roskfpos_amd import capi  # noqa: E402
from roskfpos_amd.dist import device_trace  # noqa: E402
from roskfpos_amd.synth import Workload  # noqa: E402

K, W = 200, 50
dev = "cuda:0"


def setup(T, tag0):
    w = Workload(T, 8, tag0=tag0)
    tr = device_trace(torch, w, W + K, dev, True, np.float32)
    b = capi.KfposBank(capi.MODEL_TOA_IMU, T, w.anchors, storage=capi.STORE_MIXED, init_pos=w.init_positions())
    return w, tr, b


def run(parts, streams):
    def go(lo, hi):
        for s in range(lo, hi):
            for (w, tr, b), st in zip(parts, streams):
                T = w.n_tags
                b.step_toa_imu_dev(tr["ranges"][s], tr["err"], tr["accel"][s], tr["cov"], tr["dts"][s], latch=False,
                                   stream=st.cuda_stream)
    go(0, W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(W, W + K)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6


one = [setup(65536, 0)]
us1 = run(one, [torch.cuda.Stream()])
two = [setup(32768, 0), setup(32768, 32768)]
us2 = run(two, [torch.cuda.Stream(), torch.cuda.Stream()])
four = [setup(16384, 16384 * k) for k in range(4)]
us4 = run(four, [torch.cuda.Stream() for _ in range(4)])
print(json.dumps({"per_epoch_us_one_bank_one_stream": us1, "per_epoch_us_two_halves_two_streams": us2,
                  "per_epoch_us_four_quarters_four_streams": us4}))

#!/usr/bin/env python3
"""Step-kernel timings for the BASELINE configurations (device-resident traces, HIP events).

    python tools/kbench.py [--steps 50] [--configs c2,c3,c4shard,c5,toa6_65k]

Prints one line per configuration: us per launch, tag-steps/s, algorithmic GB/s (SURVEY 8d byte figures).
Set KFPOS_GENERIC_KERNEL=1 to time the LDS-staged generic kernels instead of the anchor-count-specialised ones.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from roskfpos_amd import capi  # noqa: E402
from roskfpos_amd.synth import Workload  # noqa: E402

CONFIGS = {
    # name: (model, tags, anchors, storage, top_n, ignore_worst, algorithmic bytes per tag-step)
    "c2": (0, 4096, 8, capi.STORE_F64, 0, False, 560),
    "toa6_65k": (0, 65536, 8, capi.STORE_F64, 0, False, 560),
    "c4shard": (0, 131072, 8, capi.STORE_F64, 0, False, 560),
    "c3": (1, 65536, 8, capi.STORE_MIXED, 0, False, 544),
    "c3_f32": (1, 65536, 8, capi.STORE_F32, 0, False, 544),
    # BASELINE configs[2] / [4] as worded ("fp32" = compact storage) in the compact mode that meets the 1e-6 m bar:
    # 6-byte covariance entries (KFPOS_STORE_P48), f32 measurements
    "c3_p48": (1, 65536, 8, capi.STORE_P48, 0, False, 544),
    "c5_p48": (0, 262144, 16, capi.STORE_P48, 2, False, 344),
    "c5": (0, 262144, 16, capi.STORE_F32, 2, False, 344),
    "iw8": (0, 65536, 8, capi.STORE_F64, 0, True, 560),
    # the reference's default start: no initial position, every tag initialises from its first ML solve and keeps the
    # non-symmetric covariance that start leaves (KalmanFilterTOA.cpp:90-108) -- full 36-entry layout, 848 B per unit
    "toa6_65k_mlinit": (0, 65536, 8, capi.STORE_F64, 0, False, 848),
    "c3_mlinit": (1, 65536, 8, capi.STORE_MIXED, 0, False, 544),
    # small plain 6-state banks: where the 8-lanes-per-tag kernel (k_step_toa6_coop, KFPOS_NO_COOP=1 disables) pays off
    "t1k": (0, 1024, 8, capi.STORE_F64, 0, False, 560),
    "t8k": (0, 8192, 8, capi.STORE_F64, 0, False, 560),
    "t16k": (0, 16384, 8, capi.STORE_F64, 0, False, 560),
    "t32k": (0, 32768, 8, capi.STORE_F64, 0, False, 560),
    # 8-state planar filter (KalmanFilter): ranging-only bank, and the same with IMU + compass samples latched
    # (every ranging epoch then carries 4 sensor rows; LDS-staged kernel). state 7+36 doubles r/w, epoch 96 B,
    # trajectory 24 B, flags + status 12 B
    # standalone ML estimator (ALGORITHM_ML): position 3 + covariance 6 doubles written, epoch 96 B read
    "ml": (2, 65536, 8, capi.STORE_F64, 0, False, 96 + 72 + 24),
    "planar": (3, 65536, 8, capi.STORE_F64, 0, False, 820),
    "planar_sens": (3, 65536, 8, capi.STORE_F64, 0, False, 820 + 80),
}
PLANAR_CFG = dict(use_fixed_height=1, fixed_height=1.0, init_angle=0.3, px4_height=0.8, px4_arm_p1=0.05,
                  px4_arm_p2=-0.02, px4_cov_velocity=0.002, px4_cov_gyro_z=0.001, imu_use_fixed_cov_acc=0,
                  imu_cov_acc=0.02, imu_use_fixed_cov_ang_vel_z=1, imu_cov_ang_vel_z=0.0005, mag_angle_offset=0.0,
                  mag_cov=0.01)


def run(name, steps, warmup, epl=0):
    base, _, st = name.partition("@")  # "toa6_65k@p48": a configuration in another storage mode
    model, T, A, storage, top_n, iw, nbytes = CONFIGS[base]
    if st:
        storage = {"f64": capi.STORE_F64, "mixed": capi.STORE_MIXED, "f32": capi.STORE_F32, "p48": capi.STORE_P48}[st]
    w = Workload(T, A)
    real = np.float64 if storage == capi.STORE_F64 else np.float32
    S = steps + warmup
    dev = "cuda:0"
    ranges = torch.empty((S, A, T), dtype=torch.int32, device=dev)
    accel = torch.empty((S, 3, T), dtype=torch.float32 if real == np.float32 else torch.float64, device=dev)
    for s in range(S):
        ranges[s].copy_(torch.from_numpy(np.ascontiguousarray(w.ranges_mm(s).T)))
        if model == 1:
            accel[s].copy_(torch.from_numpy(np.ascontiguousarray(w.accel(s, real).T)))
    err = torch.from_numpy(np.ascontiguousarray(w.err_est(real).T)).to(dev)
    cov = torch.from_numpy(np.ascontiguousarray(w.accel_cov(real).T)).to(dev)
    dts = np.array([w.dt_of(s) for s in range(S)])
    bank = capi.KfposBank(model, T, w.anchors, storage=storage, top_n=top_n, ignore_worst=iw,
                          init_pos=None if base.endswith("_mlinit") else w.init_positions(),
                          planar=PLANAR_CFG if model == 3 else None)
    if base == "planar_sens":  # latch an IMU and a compass sample: ranging epochs carry their rows from now on
        wv, la = w.planar_imu(0)
        bank.step_planar_imu(wv, np.tile(np.eye(3).ravel() * 1e-4, (T, 1)), la, w.accel_cov(), 0.1)
        bank.step_compass(w.compass(0), 0.0)
    stream = torch.cuda.current_stream().cuda_stream
    status = torch.zeros(T, dtype=torch.int32, device=dev)

    def go(s0, n):
        kw = dict(accel=accel[s0], stride_accel=3 * T, cov=cov, stride_cov=0) if model == 1 else {}
        bank.run_trace_dev(n, ranges[s0], A * T, err, 0, dts[s0:s0 + n], status=status, stream=stream, **kw)

    def many(s0, n):  # epl epochs per launch (0: all n in one launch)
        for k in range(s0, s0 + n, epl or n):
            go(k, min(epl or n, s0 + n - k))

    many(0, warmup)
    torch.cuda.synchronize()
    bank.timing_begin(stream)
    many(warmup, steps)
    ms = bank.timing_end(stream)
    us = ms * 1e3 / steps
    st = status.cpu().numpy().astype(np.uint32)
    # getPose kernel (predict-only extrapolation of every tag), device buffers: state in, pos 3 + cov 9 + vel 3 out
    pos_t = torch.zeros((3, T), dtype=torch.float64, device=dev)
    cov_t = torch.zeros((9, T), dtype=torch.float64, device=dev)
    vel_t = torch.zeros((3, T), dtype=torch.float64, device=dev)
    for _ in range(3):
        bank.get_pose_dev(0.02, pos_t, cov_t, vel_t, status, stream)
    torch.cuda.synchronize()
    bank.timing_begin(stream)
    for _ in range(20):
        bank.get_pose_dev(0.02, pos_t, cov_t, vel_t, status, stream)
    pose_us = bank.timing_end(stream) * 1e3 / 20
    x, _, _ = bank.get_state()
    truth = w.position(w.time_of(S - 1))
    if model == 3:
        x, truth = x[:, :2], truth[:, :2]
    out = {"config": name, "tags": T, "anchors": A, "epochs_per_launch": epl or steps, "us_per_launch": round(us, 2),
           "tag_steps_per_s": T / us * 1e6, "get_pose_us": round(pose_us, 2), "algo_GBps": nbytes * T / us / 1e3,
           "hbm_frac": nbytes * T / us / 1e3 / 8000.0,
           "mean_gain_iters": float(((st >> 8) & 0xFF).mean()), "mean_ml_iters": float(((st >> 16) & 0xFF).mean()),
           "max_gain_iters": int(((st >> 8) & 0xFF).max()),
           "gain_iters_hist": np.bincount((st >> 8) & 0xFF, minlength=21).tolist(),
           "frac_waves_with_max_lane": float((((st >> 8) & 0xFF).reshape(-1, 64).max(1) == ((st >> 8) & 0xFF).max()).mean()),
           "mean_wave_max_gain": float(((st >> 8) & 0xFF).reshape(-1, 64).max(1).mean()),
           "mean_wave_max_ml": float(((st >> 16) & 0xFF).reshape(-1, 64).max(1).mean()),
           "rms_vs_truth": float(np.sqrt(((x[:, :3] - truth) ** 2).sum(1).mean())),
           "generic": os.environ.get("KFPOS_GENERIC_KERNEL", "0")}
    print(json.dumps(out), flush=True)
    bank.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--configs", default="c2,toa6_65k,c3,c5")
    ap.add_argument("--epochs-per-launch", type=int, default=int(os.environ.get("KBENCH_EPL", "0")),
                    help="0 = all epochs in one launch (default); 1 = what a live caller of kfpos_step_*_dev pays. "
                         "us_per_launch is per EPOCH either way (HIP events around all launches, gaps included)")
    a = ap.parse_args()
    for n in a.configs.split(","):
        run(n, a.steps, a.warmup, a.epochs_per_launch)

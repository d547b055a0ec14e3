#!/usr/bin/env python3
"""bench.py -- EKF predict+update throughput of the MI355X batched core on BASELINE.json's metric.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 65 536 tags x 8 anchors
PER GPU, UWB+IMU fused 9-state EKF (kfpos_toa_imu path), fp64 arithmetic, synthetic traces of SURVEY.md
8d. One "step" = one ranging epoch for every tag: predict + full iterated update fusing the epoch's
accelerometer sample, and the resulting pose written out. All inputs for every step are resident in HBM
before the timed region. The trace is replayed with kfpos_run_trace_dev: --epochs-per-launch epochs per
kernel launch (the per-tag state stays in registers between them; 1 = one launch per epoch, which is also
measured and reported as `per_epoch_launch`). With --gpus N > 1 (weak scaling: every rank owns 65 536 tags)
the poses at the end of every launch are all-gathered over RCCL, overlapped with the next launch.

Prints ONE JSON line on rank 0 (contract in the task description), with `roofline` for the step kernel
and `cpu_baseline` = the oracle timed on the host cores over a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from roskfpos_amd import capi  # noqa: E402
from roskfpos_amd.dist import PoseGather, env_world, shard_range  # noqa: E402
from roskfpos_amd.synth import Workload  # noqa: E402

TAGS_PER_GPU = 65536
ANCHORS = 8
# algorithmic bytes per tag-step of config 3 -- read x(9) + packed P(45) as f32, write them
# back, 8 int32 ranges + 8 f32 errorEstimations, accel(3) + covariance(9) f32:
#   (9+45)*4*2 + 8*4*2 + (3+9)*4 = 544 B
ALGO_BYTES_PER_TAG_STEP = 544
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Covariance storage in HBM. BASELINE configs[2] says fp32, but with a 24-bit covariance the 9-state filter
# sits at 1.6e-6 m RMS from the CPU reference over 100 steps (tests/test_gpu_parity.py), above the 1e-6 m
# bar, so the measured configuration keeps it in f64 (KFPOS_STORE_MIXED: measurements stay f32 / int32). The kernel is
# VALU-bound, so this does not change its duration; DESIGN.md "storage precision".
STORAGE = {"mixed": capi.STORE_MIXED, "f64": capi.STORE_F64, "f32": capi.STORE_F32}[
    os.environ.get("KFPOS_BENCH_STORAGE", "mixed")]


def upload_trace(torch, w, n_steps, device):
    """ranges [S][A][T] int32, accel [S][3][T] f32, err [A][T] f32, cov [9][T] f32, dt (S,)"""
    T, A = w.n_tags, w.n_anchors
    ranges = torch.empty((n_steps, A, T), dtype=torch.int32, device=device)
    accel = torch.empty((n_steps, 3, T), dtype=torch.float32, device=device)
    dts = np.zeros(n_steps)
    for s in range(n_steps):
        ranges[s].copy_(torch.from_numpy(np.ascontiguousarray(w.ranges_mm(s).T)))
        accel[s].copy_(torch.from_numpy(np.ascontiguousarray(w.accel(s, np.float32).T)))
        dts[s] = w.dt_of(s)
    err = torch.from_numpy(np.ascontiguousarray(w.err_est(np.float32).T)).to(device)
    cov = torch.from_numpy(np.ascontiguousarray(w.accel_cov(np.float32).T)).to(device)
    return ranges, accel, err, cov, dts


def cpu_baseline_and_rms(w, anchors, sample_tags, sample_steps, threads):
    """Oracle (the CPU restatement of the reference path) on a bounded sample of the same workload,
    and the GPU's RMS position difference against it on that sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    ws = Workload(sample_tags, ANCHORS, tag0=w.tag0)
    e32 = ws.err_est(np.float32)
    c32 = ws.accel_cov(np.float32)
    e64, c64 = e32.astype(np.float64), c32.astype(np.float64)
    orc = oracle_py.OracleBank(1, sample_tags, anchors, init_pos=ws.init_positions(), n_threads=threads)
    gpu = capi.KfposBank(capi.MODEL_TOA_IMU, sample_tags, anchors, storage=STORAGE,
                         init_pos=ws.init_positions())
    cpu_s = 0.0
    for s in range(sample_steps):
        r, a32, dt = ws.ranges_mm(s), ws.accel(s, np.float32), ws.dt_of(s)
        a64 = a32.astype(np.float64)
        t0 = time.perf_counter()
        orc.step_imu(a64, c64, 0.0)   # newIMUMeasurement at timeLag 0 (latch), then the ranging epoch
        orc.step_toa(r, e64, dt)
        cpu_s += time.perf_counter() - t0
        gpu.step_toa_imu(r, e32, a32, c32, dt)
    xo, _ = orc.get_state()
    xg, _, _ = gpu.get_state()
    gpu.close()
    rms = float(np.sqrt(((xo[:, :3] - xg[:, :3]) ** 2).sum(1).mean()))
    return sample_tags * sample_steps / cpu_s, cpu_s, rms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50,
                    help="untimed epochs first; the chip needs a few launches to settle its clocks under this fp64 load")
    ap.add_argument("--tags-per-gpu", type=int, default=TAGS_PER_GPU)
    ap.add_argument("--epochs-per-launch", type=int, default=25,
                    help="epochs fused into one kernel launch (state resident in registers); 1 = one launch per epoch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-per-epoch", action="store_true", help="skip the extra one-launch-per-epoch measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_world()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    # one rank per GPU; KFPOS_BENCH_BACKEND=gloo lets several ranks share one card to rehearse the N>1 code path
    backend = os.environ.get("KFPOS_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)

    T = args.tags_per_gpu
    K, W = args.steps, args.warmup
    E = max(1, min(args.epochs_per_launch, 128))
    lo, hi = shard_range(T * world, world, rank)
    w = Workload(T, ANCHORS, tag0=lo)
    ranges, accel, err, cov, dts = upload_trace(torch, w, W + K, device)
    stream = torch.cuda.current_stream().cuda_stream
    traj = torch.zeros((W + K, 3, T), dtype=torch.float64, device=device)  # pose after every epoch

    def make_bank():
        return capi.KfposBank(capi.MODEL_TOA_IMU, T, w.anchors, storage=STORAGE,
                              init_pos=w.init_positions(), device=local_rank)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(epochs_per_launch):
        """W warm-up epochs, then exactly K timed epochs. Every launch covers `epochs_per_launch` epochs
        (predict + update for every tag in each) and writes the pose after each epoch; with several
        ranks the poses at the end of every launch are all-gathered over RCCL, overlapped with the next
        launch. Returns (wall seconds, HIP-event ms around the step launches, number of launches, bank)."""
        bank = make_bank()
        gather = PoseGather(T, device) if world > 1 else None
        launches = 0

        def run(s0, n):
            nonlocal launches
            s = s0
            while s < s0 + n:
                m = min(epochs_per_launch, s0 + n - s)
                bank.run_trace_dev(m, ranges[s], ANCHORS * T, err, 0, dts[s:s + m], accel=accel[s],
                                   stride_accel=3 * T, cov=cov, stride_cov=0, trajectory=traj[s], stream=stream)
                launches += 1
                s += m
                if gather is not None:
                    gather.buffer().copy_(traj[s - 1], non_blocking=True)
                    gather.gather()
            if gather is not None:
                gather.wait()

        run(0, W)
        fence()
        launches = 0
        bank.timing_begin(stream)
        t0 = time.perf_counter()
        run(W, K)
        kernel_ms = bank.timing_end(stream)  # HIP events on the launch stream around the timed launches
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed, kernel_ms = float(tmax[0]), float(tmax[1])
        return elapsed, kernel_ms, launches, bank

    elapsed, kernel_ms, launches, bank = measure(E)
    x, P, _ = bank.get_state()
    truth = w.position(w.time_of(W + K - 1))
    track_rms = float(np.sqrt(((x[:, :3] - truth) ** 2).sum(1).mean()))
    finite = bool(np.isfinite(x).all() and np.isfinite(P).all())
    traj_ok = bool(np.array_equal(traj[W + K - 1].cpu().numpy().T, x[:, :3]))
    bank.close()
    per_epoch = None
    if not args.no_per_epoch and E != 1:
        e1, k1, l1, b1 = measure(1)
        b1.close()
        per_epoch = {"value": T * world * K / e1, "unit": "tag-steps/s", "ms_per_step": e1 * 1e3 / K,
                     "kernel_us_per_launch": k1 * 1e3 / l1,
                     "algorithmic_GBps": ALGO_BYTES_PER_TAG_STEP * T / (k1 * 1e-3 / l1) / 1e9}

    if rank == 0:
        total_steps = T * world * K
        value = total_steps / elapsed
        per_launch_s = kernel_ms * 1e-3 / launches
        units_per_launch = T * K / launches
        achieved = ALGO_BYTES_PER_TAG_STEP * units_per_launch / per_launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # what actually bounds the step kernel: fp64 VALU issue (committed PMC summary of this same command)
        valu = None
        ppath = os.path.join(ROOT, "profiles", "r01h_pmc_sq_counters.json")
        if os.path.exists(ppath):
            try:
                c = next(v for k, v in json.load(open(ppath)).items() if "k_step_imu9" in k)
                valu = {"busy_frac": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
                        "fp64_instr_per_wave_epoch": c["SQ_INSTS_VALU"] / c["SQ_WAVES"] / 25,
                        "source": "profiles/r01h_pmc_sq_counters.json (rocprofv3 --pmc, 25 epochs per launch)"}
            except Exception:
                valu = None
        out = {
            "metric": "EKF predict+update steps/s at 65536 tags x 8 anchors; RMS pos err vs CPU ref",
            "value": value, "unit": "tag-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: 65536 tags x 8 anchors per GPU, UWB+IMU fused 9-state "
                                   "IEKF (kfpos_toa_imu path), fp64 arithmetic, f32/int32 measurements, f64 "
                                   "covariance (KFPOS_STORE_MIXED: a 24-bit covariance misses the 1e-6 m bar)",
                       "tags_per_gpu": T, "anchors": ANCHORS, "total_tags": T * world,
                       "epochs_per_launch": E, "pose_output": "every epoch ([S][3][T] f64)",
                       "pose_gather": (("rccl" if backend == "nccl" else backend + " (rehearsal)") +
                                       " all_gather per launch, overlapped") if world > 1 else "none (single GPU)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_step_imu9<double,float,8>", "kernel_us_per_launch": per_launch_s * 1e6,
                         "units_per_launch": units_per_launch,
                         "algorithmic_bytes_per_tag_step": ALGO_BYTES_PER_TAG_STEP,
                         "valu": valu},
            "state_finite": finite, "trajectory_matches_state": traj_ok, "rms_vs_truth_m": track_rms,
        }
        if per_epoch is not None:
            out["per_epoch_launch"] = per_epoch
        if world == 1 and not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, 16)
            sample_tags, sample_steps = 16384, 220  # ~10 s of oracle time on 16 host threads
            v, secs, rms = cpu_baseline_and_rms(w, w.anchors, sample_tags, sample_steps, threads)
            v1, secs1, _ = cpu_baseline_and_rms(w, w.anchors, 1024, 60, 1)  # a 1-core figure beside it
            out["cpu_baseline"] = {"value": v, "unit": "tag-steps/s", "cores": threads, "kind": "port",
                                   "sample": f"first {sample_tags} tags x {sample_steps} steps of the same "
                                             f"workload ({secs:.1f} s of oracle time)",
                                   "one_core_value": v1,
                                   "one_core_sample": f"first 1024 tags x 60 steps ({secs1:.1f} s)"}
            out["rms_pos_err_vs_cpu_ref_m"] = rms
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- EKF predict+update throughput of the MI355X batched core on BASELINE.json's metric.

Default workload (--config c3 = BASELINE.json configs[2], the configuration the metric is quoted on): 65 536 tags x
8 anchors PER GPU (weak scaling), UWB+IMU fused 9-state IEKF (kfpos_toa_imu path), fp64 arithmetic, synthetic traces
of SURVEY.md 8d. One "step" = one ranging epoch for every tag: predict + full iterated update fusing the epoch's
accelerometer sample, and the resulting pose written out. All inputs for every step are resident in HBM before the
timed region. The trace is replayed with kfpos_run_trace_dev: --epochs-per-launch epochs per kernel launch (the
per-tag state stays in registers between them). The same K epochs are also replayed with ONE launch (and, with several
ranks, one pose all-gather) per epoch -- what a live 20 Hz node pays -- and reported as `per_epoch_launch`.

--config c4 = BASELINE.json configs[3]: 1 048 576 tags x 8 anchors in total, UWB-only 6-state EKF, f64, the tag batch
split over the ranks by dist.shard_range (strong scaling: the total is fixed), poses exchanged with one RCCL all-gather
per launch (--gather launch, the default), per epoch (--gather epoch) or per launch with every epoch's poses
(--gather trajectory).

Launching. `python bench.py --gpus N` works two ways:
  * under a launcher (torchrun / `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`): RANK,
    LOCAL_RANK, WORLD_SIZE, MASTER_* come from the environment and this process is one rank;
  * plainly, with WORLD_SIZE unset and N > 1: this process becomes the LAUNCHER -- before anything that could touch
    the GPU is imported it starts N fresh child processes of itself (one rank each, rendezvous on 127.0.0.1), relays
    rank 0's JSON line and exits with the first non-zero child status. It never initialises the GPU itself.
With --gpus N > 1 rank g owns the contiguous global tag range shard_range(total, N, g) and regenerates exactly its own
inputs; the pose all-gather runs on a side stream, overlapped with the next launch -- through the library's own RCCL
communicator behind the C ABI (kfpos_allgather_poses) when every rank has a GPU of its own, through torch.distributed
otherwise (gloo rehearsals of several ranks on one card; roskfpos_amd/dist.py).

Prints ONE JSON line on rank 0 (contract in the task description), with `roofline` for the step kernel
and `cpu_baseline` = the oracle timed on the host cores over a bounded sample of the same workload.

--dry-run: no GPU, no filter -- the launcher, the rendezvous, the shard arithmetic, the gather plumbing (gloo, CPU
tensors) and the JSON contract only. `value` is null and the line says so; used by the CPU test-suite.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from roskfpos_amd import capi  # noqa: E402  (ctypes only: nothing here touches the GPU until a bank is created)
from roskfpos_amd.dist import GATHER_MODES, ShardedReplay, device_trace, env_world, shard_range  # noqa: E402
from roskfpos_amd.synth import Workload  # noqa: E402

ANCHORS = 8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

# SURVEY.md 8d, algorithmic bytes per tag-step:
#   c3: read x(9) + packed P(45) as f32, write them back, 8 int32 ranges + 8 f32 errorEstimations,
#       accel(3) + covariance(9) f32:                          (9+45)*4*2 + 8*4*2 + (3+9)*4 = 544 B
#   c4 / 6-state f64: x(6) + packed P(21) f64 in and out, 8 ranges + 8 errorEstimations at 8 B:
#                                                              (6+21)*8*2 + 8*8*2           = 560 B
# Covariance storage of c3: BASELINE configs[2] says fp32, i.e. compact storage. With a 24-bit covariance
# (KFPOS_STORE_F32) the 9-state filter sits at 1.6e-6 m RMS from the CPU reference over 100 steps
# (tests/test_gpu_parity.py), above the 1e-6 m bar; KFPOS_STORE_P48 keeps every entry in 6 bytes with 40 significant bits
# (kfpos_p48.h; measurements f32 / int32): 1e-9 m on this trace, and the kernel is as fast as with an 8-byte covariance
# (same box, alternating: 36.99 against 37.06 us per epoch, profiles/r03m_ab_p48_vs_mixed_same_box.jsonl). What separates
# the two is the long run: the 9-state filter's capped steps multiply ANY difference, in rare epochs by 1e8 and more, so
# over 2 048 tags x 10 000 epochs P48's 2^-40 rounding ends at 2.7e-6 m RMS (one epoch at 5e-5 m) where the 8-byte
# covariance, which only carries rounding-level differences into those epochs, stays at 1e-8 m (profiles/r03p_*). The
# headline therefore runs MIXED -- f64 covariance in HBM, f32 / int32 measurements -- and the SAME line carries the P48 run
# of the same trace as secondary.c3_p48 (throughput, kernel time, RMS against the CPU reference): configs[2] as
# BASELINE.json words it. KFPOS_BENCH_STORAGE=p48|f64|f32 makes another mode the headline. roofline.achieved uses SURVEY
# 8d's 544 B either way (P48 moves 54*6*2 + 112 = 760 B per tag-step when every epoch is its own launch, MIXED 976 B).
_STORAGES = {"mixed": capi.STORE_MIXED, "f64": capi.STORE_F64, "f32": capi.STORE_F32, "p48": capi.STORE_P48}
STORAGE_C3_NAME = os.environ.get("KFPOS_BENCH_STORAGE", "mixed")
STORAGE_C3 = _STORAGES[STORAGE_C3_NAME]
_C3_TEXT = {
    "mixed": ("f64 arithmetic; f64 state and covariance, f32/int32 measurements in HBM (KFPOS_STORE_MIXED)",
              "f32/int32 measurements, f64 covariance (KFPOS_STORE_MIXED: a 24-bit covariance misses the 1e-6 m bar)"),
    "f64": ("f64", "f64 measurements and covariance (KFPOS_STORE_F64)"),
    "f32": ("f64 arithmetic; f32 covariance and measurements in HBM (KFPOS_STORE_F32)",
            "f32 measurements and covariance (KFPOS_STORE_F32: misses the 1e-6 m bar, 1.6e-6 m)"),
    "p48": ("f64 arithmetic; 48-bit covariance, f32/int32 measurements in HBM (KFPOS_STORE_P48)",
            "f32/int32 measurements, covariance in 6 bytes per entry (KFPOS_STORE_P48: 40 significant bits)"),
}[STORAGE_C3_NAME]
CONFIGS = {
    "c3": dict(model=capi.MODEL_TOA_IMU, storage=STORAGE_C3, bytes=544, scaling="weak", tags=65536,
               kernel={"mixed": "k_step_imu9<double,float,8,true>", "f64": "k_step_imu9<double,double,8,true>",
                       "f32": "k_step_imu9<float,float,8,true>", "p48": "k_step_imu9<p48,float,8,true>"}[STORAGE_C3_NAME],
               dtype=_C3_TEXT[0],
               workload="BASELINE configs[2]: 65536 tags x 8 anchors per GPU, UWB+IMU fused 9-state IEKF "
                        "(kfpos_toa_imu path), fp64 arithmetic, " + _C3_TEXT[1]),
    "c4": dict(model=capi.MODEL_TOA, storage=capi.STORE_F64, bytes=560, scaling="strong", tags=1048576,
               kernel="k_step_toa6<true,double,double,8,0>", dtype="f64",
               workload="BASELINE configs[3]: 1048576 tags x 8 anchors in total, UWB-only 6-state IEKF, f64, "
                        "sharded over the ranks (dist.shard_range), pose all-gather"),
}


# ------------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` with no launcher around it: become one. Starts N children of this script BEFORE this
    process has imported torch or made any HIP call (a process that has initialised the GPU must not be replaced or
    forked into ranks), gives each its RANK / LOCAL_RANK / WORLD_SIZE and a rendezvous on 127.0.0.1, relays rank 0's
    stdout (the JSON line), and ends the others by their exact PIDs if one fails."""
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this platform
    env["KFPOS_BENCH_SELF_LAUNCHED"] = "1"
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    def relay():  # rank 0 prints the one JSON line; whatever else lands on its stdout (gloo's connection banner) is
        for line in procs[0].stdout:  # diagnostics and goes to stderr, so that stdout carries the line and nothing else
            text = line.decode(errors="replace")
            out = sys.stdout if text.lstrip().startswith("{") else sys.stderr
            out.write(text)
            out.flush()

    import signal
    import threading
    pump = threading.Thread(target=relay, daemon=True)
    pump.start()

    def pass_on(signum, _frame):  # whoever stops the launcher stops its ranks too (exactly the PIDs started above)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        raise KeyboardInterrupt

    signal.signal(signal.SIGTERM, pass_on)
    rc = 0
    try:
        pending = set(range(n))
        kill_at = None
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:  # a rank died: the others would wait in a collective until its timeout
                    rc = code
                    sys.stderr.write(f"bench.py: rank {r} exited with status {code}; stopping the others\n")
                    for o in pending:
                        procs[o].terminate()  # exactly the PIDs started above
                    kill_at = time.time() + 10.0
            if kill_at is not None and pending and time.time() > kill_at:
                for o in pending:
                    procs[o].kill()
                kill_at = time.time() + 10.0
            time.sleep(0.05)
        pump.join(timeout=10.0)
    except KeyboardInterrupt:
        rc = 130
        for p in procs:
            if p.poll() is None:
                p.terminate()
    return rc


# ------------------------------------------------------------------------------------------------ helpers
def host_cores() -> int:
    """CPU cores this process may use: the scheduler affinity, capped by the cgroup quota of the box."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline_and_rms(cfg, w, sample_tags, sample_steps, threads):
    """Oracle (the CPU restatement of the reference path) on a bounded sample of the same workload,
    and the GPU's RMS position difference against it on that sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    imu = cfg["model"] == capi.MODEL_TOA_IMU
    real = np.float64 if cfg["storage"] == capi.STORE_F64 else np.float32
    ws = Workload(sample_tags, ANCHORS, tag0=w.tag0)
    er = ws.err_est(real)
    e64 = er.astype(np.float64)
    orc = oracle_py.OracleBank(cfg["model"], sample_tags, ws.anchors, init_pos=ws.init_positions(), n_threads=threads)
    gpu = capi.KfposBank(cfg["model"], sample_tags, ws.anchors, storage=cfg["storage"], init_pos=ws.init_positions())
    if imu:
        cr = ws.accel_cov(real)
        c64 = cr.astype(np.float64)
    cpu_s = 0.0
    for s in range(sample_steps):
        r, dt = ws.ranges_mm(s), ws.dt_of(s)
        if imu:
            ar = ws.accel(s, real)
            a64 = ar.astype(np.float64)
            t0 = time.perf_counter()
            orc.step_imu(a64, c64, 0.0)   # newIMUMeasurement at timeLag 0 (latch), then the ranging epoch
            orc.step_toa(r, e64, dt)
            cpu_s += time.perf_counter() - t0
            gpu.step_toa_imu(r, er, ar, cr, dt)
        else:
            t0 = time.perf_counter()
            orc.step_toa(r, e64, dt)
            cpu_s += time.perf_counter() - t0
            gpu.step_toa(r, er, dt)
    xo, _ = orc.get_state()
    xg, _, _ = gpu.get_state()
    gpu.close()
    rms = float(np.sqrt(((xo[:, :3] - xg[:, :3]) ** 2).sum(1).mean()))
    return sample_tags * sample_steps / cpu_s, cpu_s, rms


def static_counters(kernel_key, epochs_in_launch):
    """HBM traffic / VALU counters of the step kernel from the committed rocprofv3 --pmc summaries. They are NOT
    measured by this run (PMC collection needs its own profiler passes, MI355X_MICROARCH.md): every entry carries its
    source -- file, command, and the commit the profiled library was built from -- and the byte count is rebuilt for
    the launch length actually timed from the two components the summaries separate: state traffic paid once per
    launch, measurement + pose traffic paid per epoch."""
    out = {"traffic": None, "traffic_source": None, "valu": None}
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        t = json.load(open(tpath)).get(kernel_key)
        if t:
            out["traffic"] = t["bytes_per_launch_fixed"] + t["bytes_per_epoch"] * epochs_in_launch
            out["traffic_source"] = (f"profiles/traffic_latest.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                     f"passes of {t['measured_with']}; library at commit {t.get('commit', 'unrecorded')}; "
                                     f"{t['bytes_per_launch_fixed']:.0f} B per launch + "
                                     f"{t['bytes_per_epoch']:.0f} B per epoch, rebuilt for {epochs_in_launch:g} epochs)")
    except Exception:
        pass
    ppath = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        c = json.load(open(ppath)).get(kernel_key)
        if c:
            out["valu"] = {"busy_frac": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
                           "fp64_instr_per_wave_epoch": c["SQ_INSTS_VALU"] / c["SQ_WAVES"] / c["epochs_per_launch"],
                           "source": f"profiles/pmc_latest.json (static: {c['measured_with']}; library at commit "
                                     f"{c.get('commit', 'unrecorded')})"}
    except Exception:
        pass
    return out


class _DryBank:
    """--dry-run stand-in for capi.KfposBank: it runs NO filter (the trajectory buffer stays as it is), so that the
    launcher, the rendezvous, the sharding and the gather plumbing can be exercised on a machine without a GPU. Nothing
    it produces is a measurement and the JSON line says so."""

    def __init__(self, T):
        self.T = T

    def run_trace_dev(self, n_steps, *args, **kw):
        return None

    def timing_begin(self, stream=None):
        self._t0 = time.perf_counter()

    def timing_end(self, stream=None):
        return (time.perf_counter() - self._t0) * 1e3

    def close(self):
        pass


# ------------------------------------------------------------------------------------------------ one rank
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50,
                    help="untimed epochs first; the chip needs a few launches to settle its clocks under this fp64 load")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--tags-per-gpu", type=int, default=None, help="c3: tags per rank (default 65536)")
    ap.add_argument("--total-tags", type=int, default=None, help="c4: tags in total (default 1048576)")
    ap.add_argument("--epochs-per-launch", type=int, default=25,
                    help="epochs fused into one kernel launch (state resident in registers); 1 = one launch per epoch")
    ap.add_argument("--gather", choices=GATHER_MODES, default="launch",
                    help="pose all-gather of the main measurement with several ranks (dist.ShardedReplay)")
    ap.add_argument("--repeats", type=int, default=3,
                    help="the timed K epochs are run this many times, each on a fresh bank after its own warm-up; the "
                         "line reports the median run and the spread of all of them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-per-epoch", action="store_true", help="skip the extra one-launch-per-epoch measurement")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 6-state 65536 x 8 line of the c3 run")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU, no filter: launcher / rendezvous / gather plumbing / JSON contract only (value null)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    rank, local_rank, world = env_world()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: be one. Nothing has touched the GPU yet (torch is not even imported).
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                         f"(plain `python bench.py --gpus N` does that itself)")

    import torch
    import torch.distributed as dist

    dry = args.dry_run
    if dry:
        backend, device, n_dev = "gloo", "cpu", 0
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
        n_dev = torch.cuda.device_count()
        # one rank per GPU over RCCL. With fewer GPUs than ranks (a one-GPU box rehearsing the N > 1 code path) the ranks
        # share cards and the collective goes through gloo -- said so in the line; KFPOS_BENCH_BACKEND forces either.
        backend = os.environ.get("KFPOS_BENCH_BACKEND") or ("nccl" if world <= n_dev else "gloo")
        if backend == "nccl" and world > n_dev:
            raise SystemExit(f"RCCL needs one GPU per rank: {world} ranks, {n_dev} devices (KFPOS_BENCH_BACKEND=gloo rehearses)")
        local_rank = local_rank % n_dev
        torch.cuda.set_device(local_rank)  # before the first collective: RCCL binds to the current device
        device = f"cuda:{local_rank}"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)
    try:
        run(args, cfg, torch, dist, rank, local_rank, world, backend, device, n_dev, dry)
    finally:
        if world > 1 and dist.is_initialized():
            dist.destroy_process_group()


def run(args, cfg, torch, dist, rank, local_rank, world, backend, device, n_dev, dry):
    K, W = args.steps, args.warmup
    E = max(1, min(args.epochs_per_launch, 128))
    if cfg["scaling"] == "weak":
        per_rank = args.tags_per_gpu or cfg["tags"]
        total = per_rank * world
    else:
        total = args.total_tags or cfg["tags"]
    lo, hi = shard_range(total, world, rank)
    assert (lo, hi) == capi.shard_range(total, world, rank)  # the C ABI cuts the batch the same way
    T = hi - lo
    imu = cfg["model"] == capi.MODEL_TOA_IMU
    real = np.float64 if cfg["storage"] == capi.STORE_F64 else np.float32
    w = Workload(T, ANCHORS, tag0=lo)
    trace = device_trace(torch, w, W + K, device, imu, real)
    # every kfpos launch and every copy into the gather buffers goes to torch's current stream, so that the HIP
    # events of kfpos_timing_* and torch's stream semantics see the same queue
    stream = None if dry else torch.cuda.current_stream().cuda_stream

    def make_bank(model=cfg["model"], storage=cfg["storage"]):
        if dry:
            return _DryBank(T)
        return capi.KfposBank(model, T, w.anchors, storage=storage, init_pos=w.init_positions(), device=local_rank)

    def fence():
        if not dry:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    engines = set()
    fallbacks = []
    calibrations = {}
    gathers_made = {}

    def gather_for(mode, epochs_per_launch):
        """one PoseGather (= one communicator) per block shape, shared by every replay that needs it"""
        if world == 1 or mode == "none":
            return None
        from roskfpos_amd.dist import make_pose_gather, shard_sizes
        rows = 3 * epochs_per_launch if mode == "trajectory" else 3
        if rows not in gathers_made:
            # the library's RCCL communicator behind the C ABI when every rank has a GPU of its own -- created and
            # checked with one predictable gather first; torch.distributed's collective if any rank fails that
            gathers_made[rows] = make_pose_gather(T, device, rows=rows, sizes=shard_sizes(total, world))
            engines.add(gathers_made[rows].engine)
            if gathers_made[rows].fallback_reason:
                fallbacks.append(gathers_made[rows].fallback_reason)
            if gathers_made[rows].calibration:
                calibrations[f"{rows} rows"] = gathers_made[rows].calibration
        return gathers_made[rows]

    def measure(epochs_per_launch, gather_mode, tr=trace, bank=None):
        """W warm-up epochs, then exactly K timed epochs. Every launch covers `epochs_per_launch` epochs (predict +
        update for every tag in each) and writes the pose after each epoch; with several ranks the poses are
        all-gathered per `gather_mode`, overlapped with the next launch. Returns (wall seconds, HIP-event ms around
        the timed launches, launches, gathers, bank)."""
        bank = bank or make_bank()
        rep = ShardedReplay(bank, total, device, gather_mode=gather_mode if world > 1 else "none",
                            epochs_per_launch=epochs_per_launch, stream=stream,
                            gather=gather_for(gather_mode, epochs_per_launch))
        rep.run(tr, 0, W)
        fence()
        rep.launches = 0
        g0 = rep.gather.count if rep.gather else 0
        bank.timing_begin(stream)
        t0 = time.perf_counter()
        rep.run(tr, W, K)
        kernel_ms = bank.timing_end(stream)  # HIP events on the launch stream around the timed launches
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed, kernel_ms = float(tmax[0]), float(tmax[1])
        gathers = rep.gather.count - g0 if rep.gather else 0
        return elapsed, kernel_ms, rep.launches, gathers, bank

    # ---- secondary measurements first (they also bring the chip to its steady clocks under this fp64 load) ----
    per_epoch = None
    if not args.no_per_epoch and E != 1:
        e1, k1, l1, g1, b1 = measure(1, "epoch")
        b1.close()
        per_epoch = {"value": None if dry else total * K / e1, "unit": "tag-steps/s", "ms_per_step": e1 * 1e3 / K,
                     "kernel_us_per_launch": k1 * 1e3 / l1, "launches": l1, "pose_gathers": g1,
                     "algorithmic_GBps": cfg["bytes"] * T / (k1 * 1e-3 / l1) / 1e9,
                     "what": "the same K epochs with ONE launch" + (" and one pose all-gather" if world > 1 else "") +
                             " per epoch: the cost of a live 20 Hz caller of kfpos_step_*_dev"}
    secondary = None
    p48_line = None
    if args.config == "c3" and not args.no_secondary and not dry and cfg["storage"] != capi.STORE_P48:
        # configs[2] as BASELINE.json words it -- compact (6-byte) covariance storage -- on the same trace, same K epochs
        ep, kp, lp, _, bp = measure(E, "none", bank=make_bank(storage=capi.STORE_P48))
        bp.close()
        p48_line = {"workload": "the headline workload with KFPOS_STORE_P48: covariance in 6 bytes per entry (40 "
                                "significant bits), f32 / int32 measurements",
                    "kernel": "k_step_imu9<p48,float,8,true>", "value": total * K / ep, "unit": "tag-steps/s",
                    "ms_per_step": ep * 1e3 / K, "kernel_us_per_launch": kp * 1e3 / lp, "launches": lp,
                    "roofline_frac": 544 * (T * K / lp) / (kp * 1e-3 / lp) / 1e9 / HBM_PEAK_GBS}
    if args.config == "c3" and not args.no_secondary and not dry:
        # the 6-state filter on the same ranging trace (kbench's toa6_65k): the kernel BASELINE's >= 40 % HBM target
        # is reachable for; f64 storage, so its errorEstimations are f64
        tr6 = {"ranges": trace["ranges"], "dts": trace["dts"], "traj": trace["traj"],
               "err": torch.from_numpy(np.ascontiguousarray(w.err_est(np.float64).T)).to(device)}
        e6, k6, l6, _, b6 = measure(E, "none", tr=tr6, bank=make_bank(capi.MODEL_TOA, capi.STORE_F64))
        b6.close()
        a6 = 560 * (T * K / l6) / (k6 * 1e-3 / l6) / 1e9
        secondary = {"toa6_65k": {
            "workload": f"{T} tags x 8 anchors per GPU, UWB-only 6-state IEKF, f64 (the per-GPU shard shape of "
                        "BASELINE configs[1]/[3]), same ranging trace, same K timed epochs",
            "value": total * K / e6, "unit": "tag-steps/s", "ms_per_step": e6 * 1e3 / K,
            "roofline": {"bound": "hbm", "achieved": a6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": a6 / HBM_PEAK_GBS, "kernel": "k_step_toa6<true,double,double,8,0>",
                         "kernel_us_per_launch": k6 * 1e3 / l6, "epochs_per_timed_launch": K / l6,
                         "algorithmic_bytes_per_tag_step": 560}}}

    # ---- the main measurement: R repeats of "W warm-up epochs, K timed epochs" on fresh banks; the line reports the
    # median repeat (by wall time) and the spread of all R ----
    R = max(1, args.repeats)
    runs = []
    for _ in range(R):
        elapsed, kernel_ms, launches, gathers, bank = measure(E, args.gather)
        runs.append((elapsed, kernel_ms, launches, gathers))
        if len(runs) < R:
            bank.close()
    x = P = None
    finite = traj_ok = None
    track_rms = None
    if not dry:  # state of the last repeat (every repeat replays the same trace from the same start)
        x, P, _ = bank.get_state()
        truth = w.position(w.time_of(W + K - 1))
        track_rms = float(np.sqrt(((x[:, :3] - truth) ** 2).sum(1).mean()))
        finite = bool(np.isfinite(x).all() and np.isfinite(P).all())
        traj_ok = bool(np.array_equal(trace["traj"][W + K - 1].cpu().numpy().T, x[:, :3]))
    bank.close()
    fence()
    for g in gathers_made.values():
        g.close()
    order = sorted(range(R), key=lambda i: runs[i][0])
    elapsed, kernel_ms, launches, gathers = runs[order[(R - 1) // 2]]

    if rank == 0:
        total_steps = total * K
        value = total_steps / elapsed
        per_launch_s = kernel_ms * 1e-3 / launches
        epochs_in_launch = K / launches           # mean epochs per timed launch (K need not be a multiple of E)
        units_per_launch = T * epochs_in_launch
        achieved = cfg["bytes"] * units_per_launch / per_launch_s / 1e9
        st = static_counters(cfg["kernel"], epochs_in_launch)
        us = sorted(r[1] * 1e3 / r[2] for r in runs)
        ms = sorted(r[0] * 1e3 / K for r in runs)
        gather_txt = "none (single GPU)"
        if world > 1:
            how = {"cabi": "kfpos_allgather_poses (the library's RCCL communicator behind the C ABI: ncclAllGather or "
                           "grouped ncclSend / ncclRecv, whichever pose_gather_algorithms measured faster)",
                   "torch": ("torch.distributed all_gather_into_tensor over " +
                             ("rccl" if backend == "nccl" else
                              backend + (" on CPU tensors (dry run)" if dry else " (rehearsal: ranks share a card)")))}
            gather_txt = (" + ".join(how[e] for e in sorted(engines)) +
                          f", mode '{args.gather}' ({gathers} collectives in the timed region), "
                          "side stream, overlapped with the next launch")
        out = {
            "metric": "EKF predict+update steps/s at 65536 tags x 8 anchors; RMS pos err vs CPU ref",
            "value": None if dry else value, "unit": "tag-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": cfg["scaling"],
            "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": cfg["workload"], "tags_per_gpu": T, "anchors": ANCHORS, "total_tags": total,
                       "epochs_per_launch": E, "epochs_in_timed_launches": [min(E, K - s) for s in range(0, K, E)],
                       "pose_output": "every epoch ([S][3][T] f64)", "pose_gather": gather_txt,
                       "ranks": world, "devices": n_dev, "backend": backend if world > 1 else None,
                       "launched_by": "bench.py itself (one child process per rank)"
                       if os.environ.get("KFPOS_BENCH_SELF_LAUNCHED") else
                       ("an outer launcher (RANK / WORLD_SIZE from the environment)" if world > 1 else "direct")},
            "repeats": {"n": R, "reported": "median repeat by wall time",
                        "ms_per_step": {"min": ms[0], "median": ms[(R - 1) // 2], "max": ms[-1]},
                        "kernel_us_per_launch_spread": {"min": us[0], "median": us[(R - 1) // 2], "max": us[-1]}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": st["traffic"],
                         "traffic_source": st["traffic_source"],
                         "kernel": cfg["kernel"], "kernel_us_per_launch": per_launch_s * 1e6,
                         "epochs_per_timed_launch": epochs_in_launch, "units_per_launch": units_per_launch,
                         "algorithmic_bytes_per_tag_step": cfg["bytes"],
                         "algorithmic_bytes_per_launch": cfg["bytes"] * units_per_launch,
                         "valu": st["valu"]},
            "state_finite": finite, "trajectory_matches_state": traj_ok, "rms_vs_truth_m": track_rms,
        }
        if dry:
            out["dry_run"] = ("no GPU and no filter were used: launcher, rendezvous, sharding, gather plumbing and this "
                              "line's shape only -- nothing here is a measurement")
            out["roofline"] = None
            out["data"] = "none (dry run)"
        if fallbacks:
            out["config"]["pose_gather_fallback"] = fallbacks[0]
        if calibrations:  # RCCL's all-gather against every rank sending straight to every other one, timed on this machine
            out["config"]["pose_gather_algorithms"] = calibrations
        if "cabi" in engines:  # which librccl the communicator runs on (the tests' stand-in transport reports 99999)
            out["config"]["rccl_version"] = capi.load().kfpos_comm_backend_version()
        if world > 1 and n_dev and world > n_dev:
            out["config"]["note"] = (f"{world} ranks on {n_dev} device(s): a rehearsal of the N > 1 code path, not a "
                                     "scaling measurement")
        if per_epoch is not None:
            out["per_epoch_launch"] = per_epoch
        if p48_line is not None:
            secondary = dict(secondary or {}, c3_p48=p48_line)
        if secondary is not None:
            out["secondary"] = secondary
        if world == 1 and not args.no_cpu_baseline and not dry:
            cores = host_cores()  # all host cores, count stated
            # ~10 s of oracle time: the oracle does ~2.2e4 (9-state) / ~7e4 (6-state) tag-steps/s per core
            per_core = 2.2e4 if imu else 7e4
            sample_tags = 16384 if imu else 32768
            sample_steps = int(max(20, 10.0 * per_core * cores / sample_tags))
            v, secs, rms = cpu_baseline_and_rms(cfg, w, sample_tags, sample_steps, cores)
            v1, secs1, _ = cpu_baseline_and_rms(cfg, w, 1024, 60 if imu else 200, 1)  # a 1-core figure
            out["cpu_baseline"] = {"value": v, "unit": "tag-steps/s", "cores": cores, "kind": "port",
                                   "host_cores": cores, "os_cpu_count": os.cpu_count(),
                                   "sample": f"first {sample_tags} tags x {sample_steps} steps of the same "
                                             f"workload ({secs:.1f} s of oracle time on {cores} threads)",
                                   "one_core_value": v1,
                                   "one_core_sample": f"first 1024 tags x {60 if imu else 200} steps ({secs1:.1f} s)"}
            out["rms_pos_err_vs_cpu_ref_m"] = rms
            if p48_line is not None:  # the same sample with the 48-bit covariance against the same reference
                _, _, rms48 = cpu_baseline_and_rms(dict(cfg, storage=capi.STORE_P48), w, min(sample_tags, 4096),
                                                   sample_steps, cores)
                out["secondary"]["c3_p48"]["rms_pos_err_vs_cpu_ref_m"] = rms48
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()


if __name__ == "__main__":
    main()

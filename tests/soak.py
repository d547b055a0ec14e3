#!/usr/bin/env python3
"""Long-run parity: 2 048 tags x 2 000 epochs through the C ABI against the oracle (test infrastructure: it uses oracle/, hence it lives under tests/), both
filters; prints RMS / max position difference and the fraction of differing status words every 250 epochs, and at the
end what BASELINE.json's metric asks for -- the RMS over ALL tags and ALL epochs -- next to the worst single epoch.

    python tests/soak.py [f64|mixed|f32|p48] [EPOCHS]   (covariance / measurement storage of the GPU bank; default f64, 2000)
"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from roskfpos_amd import capi
from roskfpos_amd.synth import Workload
import oracle_py
T, S = 2048, (int(sys.argv[2]) if len(sys.argv) > 2 else 2000)
mode = sys.argv[1] if len(sys.argv) > 1 else "f64"
storage = {"f64": capi.STORE_F64, "mixed": capi.STORE_MIXED, "f32": capi.STORE_F32, "p48": capi.STORE_P48}[mode]
real = np.float64 if storage == capi.STORE_F64 else np.float32
print(f"storage {mode}", flush=True)
for model in (0, 1):
    w = Workload(T, 8)
    b = capi.KfposBank(model, T, w.anchors, storage=storage, init_pos=w.init_positions())
    o = oracle_py.OracleBank(model, T, w.anchors, init_pos=w.init_positions(), n_threads=16)
    err, cov = w.err_est(real).astype(np.float64), w.accel_cov(real).astype(np.float64)
    worst, worst_at, sq_sum, biggest, mism = 0.0, -1, 0.0, 0.0, 0
    for s in range(S):
        r, a, dt = w.ranges_mm(s), w.accel(s, real).astype(np.float64), w.dt_of(s)
        if model == 1:
            sb = b.step_toa_imu(r, err.astype(real), a.astype(real), cov.astype(real), dt); o.step_imu(a, cov, 0.0); so = o.step_toa(r, err, dt)
        else:
            sb = b.step_toa(r, err.astype(real), dt); so = o.step_toa(r, err, dt)
        xb = b.get_state()[0]; xo = o.get_state()[0]
        d2 = ((xb[:, :3] - xo[:, :3]) ** 2).sum(1)
        rms = float(np.sqrt(d2.mean()))
        sq_sum += float(d2.sum()); biggest = max(biggest, float(np.sqrt(d2.max()))); mism += int((sb != so).sum())
        if rms > worst:
            worst, worst_at = rms, s + 1
        if s % 250 == 249 or s == S - 1:
            print(f"model {model} step {s+1}: rms {rms:.3e} max {np.abs(xb[:, :3]-xo[:, :3]).max():.3e} status-mismatch {(sb != so).mean():.4f}", flush=True)
    print(f"model {model} ALL {T} tags x {S} epochs: rms {np.sqrt(sq_sum / (T * S)):.3e} m | worst epoch {worst_at}: rms {worst:.3e} m | "
          f"largest single difference {biggest:.3e} m | differing status words {mism} of {T * S}", flush=True)
    b.close()

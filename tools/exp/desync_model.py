#!/usr/bin/env python3
"""Priced, not built (round 3): would letting the lanes of a wavefront run DIFFERENT epochs of their tags help the fused
9-state replay? Tags are independent, so in a K-epoch launch a lane whose gain iteration converges early could start its
tag's next epoch while its wave-mates still iterate; the wave would then pay max-over-lanes of the SUM of trips instead
of the sum over epochs of the max (20 of 20 trips in 99.9 % of the wave-epochs today, mean 9.0 per tag).

Model: a wave is a state machine; per tick it executes the trip body (590 instructions) if any lane iterates, the ML sweep
body (340) if any lane is in its Gauss-Newton solve, and a turnaround block (covariance update + predict + B^-1 + loads and
stores) for the lanes waiting for one, batched: it runs when at least K lanes wait or nobody else can proceed. Every body
costs the same with 1 or 64 lanes enabled (tools/micro/exec_mask_fp64.hip). Trip counts are drawn from the measured
histogram (profiles/r03a_kbench_all_configs_after_tu_split.jsonl, c3).

Result: with the turnaround as ONE block of 3 600 instructions the best K (48) gains 4 %; with the solve split into its
own phase every K LOSES 8-90 % (lanes spread over the states make every tick pay for every body). Not built.
    python tools/exp/desync_model.py"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", "r03a_kbench_all_configs_after_tu_split.jsonl"))]
p = np.array([r for r in rows if r["config"] == "c3"][0]["gain_iters_hist"], float)
p /= p.sum()
rng = np.random.default_rng(0)
E, W, TRIP, SWEEP = 25, 100, 590, 340


def ml_sweeps(size):  # iterations + 1; mean 3.8, as measured
    return 3 + np.minimum(rng.poisson(0.77, size), 6)


def run(K, turn, split_ml):
    tot = 0
    for _ in range(W):
        trips, mls = rng.choice(len(p), size=(64, E), p=p), ml_sweeps((64, E))
        if K is None:  # today: epochs in lockstep
            tot += (trips.max(0) * TRIP).sum() + ((mls.max(0) * SWEEP).sum() if split_ml else 0) + E * turn
            continue
        ep, rem, remml, state = np.zeros(64, int), np.zeros(64, int), np.zeros(64, int), np.zeros(64, int)
        cost = 0
        while True:  # state: 0 waits for a turnaround, 1 Gauss-Newton solve, 2 gain iteration, 3 out of epochs
            waiting, active = state == 0, ((state == 1) | (state == 2)).sum()
            if waiting.any() and (waiting.sum() >= K or active == 0):
                cost += turn
                for i in np.where(waiting)[0]:
                    if ep[i] >= E:
                        state[i] = 3
                    else:
                        remml[i], rem[i] = (mls[i, ep[i]] if split_ml else 0), trips[i, ep[i]]
                        ep[i] += 1
                        state[i] = 1 if split_ml else 2
                        if state[i] == 2 and rem[i] <= 0:
                            state[i] = 0
                continue
            if active == 0:
                break
            ml, it = state == 1, state == 2
            if ml.any():
                cost += SWEEP
                remml[ml] -= 1
                done = ml & (remml <= 0)
                state[done] = 2
                state[done & (rem <= 0)] = 0
            if it.any():
                cost += TRIP
                rem[it] -= 1
                state[it & (rem <= 0)] = 0
        tot += cost
    return tot / W / E


if __name__ == "__main__":
    for split_ml, turn in ((False, 3600), (True, 1500)):
        base = run(None, turn, split_ml)
        print(f"turnaround {turn} instructions, solve {'in its own phase' if split_ml else 'inside it'}: "
              f"lockstep {base:.0f} instructions per wave-epoch")
        for K in (8, 16, 32, 48, 64):
            d = run(K, turn, split_ml)
            print(f"   batch of {K:2d} waiting lanes: {d:.0f}  ({d / base:.2f} x)")

"""Ranging ingest / epoch assembly for many tags: kfpos_ingest.h's BatchedRangingNode,
driven by the replay CLI in batched mode, against a Python restatement of PosGenerator's table logic
(Posgenerator.cpp:143-281, 476-496) feeding one oracle filter per tag.

The message stream exercises what the reference's table distinguishes: interleaved tags, anchors that
drop out, a skipped epoch (the 50 ms timer fires, and the next sequence number then flushes the SAME
row again), the 8-bit sequence number wrapping after 256 epochs with a dropped anchor (the stale range
of 256 epochs earlier is reused: only column 0 of a new row is reset), errorEstimation <= 0 on a
same-sequence message (kept from before).
"""
import os
import subprocess

import numpy as np
import pytest

import oracle_py
from roskfpos_amd.synth import Workload

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPLAY = os.path.join(ROOT, "roskfpos_amd", "csrc", "kfpos_replay")
A = 8


class RefTagNode:
    """PosGenerator + estimator for ONE tag (what the reference node is), time passed in."""

    def __init__(self, anchors, init_pos):
        self.orc = oracle_py.OracleBank(0, 1, anchors, init_pos=init_pos[None])
        self.value = -np.ones((256, A), dtype=np.int64)   # memset(-1), Posgenerator.cpp:505
        self.err = np.zeros((256, A))
        self.count = np.zeros(256, dtype=np.int64)
        self.seq, self.armed, self.deadline = -1, False, 0.0
        self.started, self.last, self.calls = False, 0.0, 0

    def _send(self, now):  # sendRangingMeasurementIfAvailable + calculateTagLocationWithRangings
        if self.seq == -1 or self.count[self.seq] < 1:
            return
        self.armed = False
        row = np.where(self.value[self.seq] > 0, self.value[self.seq], 0).astype(np.int32)
        dt = now - self.last if self.started else 0.1
        self.last, self.started = now, True
        self.orc.step_toa(row[None], self.err[self.seq][None].copy(), dt)
        self.calls += 1

    def _timer(self, now):
        if self.armed and self.deadline <= now:
            self.armed = False
            self._send(self.deadline)

    def on_ranging(self, now, a, rng, err, seq):  # processRangingNow
        self._timer(now)
        mm = int(np.floor(rng))
        if self.seq == seq:
            self.count[seq] += 1
            self.value[seq, a] = mm
            if err > 0.0:
                self.err[seq, a] = err
        else:
            self._send(now)
            self.value[seq, 0] = -1
            self.err[seq, 0] = 0.0
            self.count[seq] = 1
            self.seq = seq
            self.value[seq, a] = mm
            self.err[seq, a] = err
        self.deadline, self.armed = now + 0.05, True

    def pose(self, now):
        self._timer(now)
        if not self.started:
            return None
        pos, cov, _, _ = self.orc.get_pose(now - self.last)
        return pos[0], cov[0]


def _stream(w, tags, n_epochs):
    """(time, kind, payload) events: R = (anchor, tag, range_mm, err, seq), P = ()"""
    ev = []
    for k in range(n_epochs):
        t0 = 10.0 + 0.05 * k
        r = w.ranges_mm(k)
        for ti, tag in enumerate(tags):
            if tag == tags[1] and k in (40, 41):      # tag 1 misses two epochs: its timer fires
                continue
            for a in range(A):
                if a == 1 and k % 7 == 3:
                    continue                            # anchor 1 drops out now and then
                if tag == tags[2] and a == 5 and k == 300:
                    continue                            # after the seq wrap: stale range of epoch 44 is reused
                err = 0.0025
                if tag == tags[0] and a == 3 and k % 5 == 2:
                    err = 0.0                           # "no error estimation": the previous value stays (same seq)
                ev.append((t0 + 0.0007 * a + 0.0001 * ti, "R", (a, tag, float(r[ti, a]) + 0.6, err, k % 256)))
        if k % 10 == 9:
            ev.append((t0 + 0.03, "P", ()))
    ev.sort(key=lambda e: e[0])
    return ev


def test_replay_batched_mode_is_built():
    if not os.path.exists(REPLAY):
        import __graft_entry__
        __graft_entry__.build()
    assert os.path.exists(REPLAY)


@pytest.mark.gpu
def test_batched_ingest_matches_per_tag_reference_nodes(tmp_path):
    tags = [0x10, 0x2A, 0x3]
    n_epochs = 320
    w = Workload(len(tags), A)
    ev = _stream(w, tags, n_epochs)
    p0 = w.init_positions()[0]  # the node has ONE initial position parameter for every tag
    lines = [f"A {100 + a} {x:.17g} {y:.17g} {z:.17g}" for a, (x, y, z) in enumerate(w.anchors)]
    for t, kind, pl in ev:
        if kind == "R":
            a, tag, rng, err, seq = pl
            lines.append(f"R {t:.9f} {100 + a} {tag} {rng:.3f} {seq} {err:.17g}")
        else:
            lines.append(f"P {t:.9f}")
    trace = str(tmp_path / "multi.txt")
    open(trace, "w").write("\n".join(lines) + "\n")
    out = subprocess.run([REPLAY, "algorithm:=ALGORITHM_KF_TOA", "useStartPosition:=0",
                          f"initPositionX:={p0[0]:.17g}", f"initPositionY:={p0[1]:.17g}",
                          f"initPositionZ:={p0[2]:.17g}", "tagIds:=" + ",".join(f"{t:x}" for t in tags), trace],
                         capture_output=True, text=True, check=True).stdout
    got = {}
    for ln in out.splitlines():
        f = ln.split()
        got.setdefault(int(f[2], 16), []).append([float(f[1]), int(f[3])] + [float(v) for v in f[4:]])

    ref = {tag: RefTagNode(w.anchors, p0) for tag in tags}
    exp = {tag: [] for tag in tags}
    for t, kind, pl in ev:
        if kind == "R":
            a, tag, rng, err, seq = pl
            ref[tag].on_ranging(t, a, rng, err, seq)
        else:
            for tag in tags:
                ps = ref[tag].pose(t)
                exp[tag].append(ps)
    # the double flush really happened for the tag that skipped epochs: more estimator calls than epochs
    assert ref[tags[1]].calls > n_epochs - 2 - 1
    assert ref[tags[0]].calls == n_epochs - 1          # the last epoch is still open
    for tag in tags:
        g = np.array(got[tag])
        assert len(g) == len(exp[tag])
        for row, ps in zip(g, exp[tag]):
            assert ps is not None and row[1] == 1
            pos, cov = ps
            assert np.abs(row[2:5] - pos).max() < 1e-8
            assert np.allclose(row[5:8], [cov[0, 0], cov[1, 1], cov[2, 2]], rtol=1e-6, atol=1e-12)

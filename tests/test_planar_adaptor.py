"""Host mirror of the reference's KalmanFilter class (kfpos_adaptor.h) and ALGORITHM_KF in the replay tool."""
import os
import subprocess

import numpy as np
import pytest

from planar import CFG, PlanarOracle
from roskfpos_amd.synth import Workload

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPLAY = os.path.join(ROOT, "roskfpos_amd", "csrc", "kfpos_replay")

XML = {
    "configUWB": '<config>\n <!-- <uwb useFixedHeight="0" fixedHeight="9.9" tagId="6e5b"/> -->\n'
                 ' <uwb useFixedHeight="1" fixedHeight="1.0" tagId="0"/>\n</config>\n',
    "configPX4Flow": '<config>\n  <px4flow armP0="0.05" armP1="-0.02" useFixedSensorHeight="1" sensorHeight="0.8" '
                     'sensorInitAngle ="-1.57" covarianceVelocity="0.002" covarianceGyroZ="0.001"/>\n</config>',
    "configIMU": '<config>\n <imu useFixedCovarianceAcceleration="0" covarianceAcceleration="0.02"  '
                 'useFixedCovarianceAngularVelocityZ="1" covarianceAngularVelocityZ="0.0005"/> \n</config>',
    "configMAG": '<config>\n<mag angleOffset="0.1" covarianceMag="0.01"/>\n</config>',
    "configPos": '<config>\n <algorithm type="1" variant="0"/>\n</config>',
}


def test_xml_attribute_reader(tmp_path):
    """The slice of boost::property_tree the reference relies on: comments skipped, defaults, last element wins."""
    src = tmp_path / "x.cpp"
    src.write_text(r'''
#include "kfpos_adaptor.h"
#include <cstdio>
using namespace kfpos_host;
int main() {
    XmlAttributes a;
    if (!a.parse("<config>\n<!-- <uwb useFixedHeight=\"0\" fixedHeight=\"9.9\"/> -->\n"
                 "<uwb useFixedHeight=\"1\" fixedHeight=\"1.049\" tagId=\"0\"/>\n"
                 "<uwbx fixedHeight=\"7\"/><px4flow sensorInitAngle =\"-1.5\" armP0='2'/></config>")) return 1;
    if (a.getInt("uwb", "useFixedHeight", 0) != 1) return 2;
    if (a.getDouble("uwb", "fixedHeight", 0) != 1.049) return 3;
    if (a.getDouble("uwb", "missing", 4.5) != 4.5) return 4;
    if (a.getDouble("px4flow", "sensorInitAngle", 0) != -1.5) return 5;
    if (a.getDouble("px4flow", "armP0", 0) != 2.0) return 6;
    if (a.getDouble("mag", "angleOffset", -3) != -3) return 7;
    if (a.getDouble("uwb", "Height", -1) != -1) return 8;   /* no suffix match on "fixedHeight" */
    XmlAttributes b;
    if (b.parse("<notconfig/>")) return 9;
    XmlAttributes c;
    if (!c.parse("<config><mag angleOffset=\"1\"/><mag angleOffset=\"2\"/></config>")) return 10;
    if (c.getDouble("mag", "angleOffset", 0) != 2.0) return 11;
    std::puts("ok");
    return 0;
}
''')
    exe = tmp_path / "x"
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I",
                           os.path.join(ROOT, "roskfpos_amd", "csrc"), "-fsyntax-only", str(src)])
    # link against the HIP library only to resolve the inline classes' symbols; nothing on the GPU is called
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I",
                           os.path.join(ROOT, "roskfpos_amd", "csrc"), "-o", str(exe), str(src), "-L",
                           os.path.join(ROOT, "roskfpos_amd", "csrc"), "-lkfpos_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "roskfpos_amd", "csrc")])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stderr)


def _write_trace(path, w, S, tag):
    lines = [f"A {100 + a} {x:.17g} {y:.17g} {z:.17g}" for a, (x, y, z) in enumerate(w.anchors)]
    t, calls = 10.0, []
    cw = np.eye(3).ravel() * 1e-4
    ca = w.accel_cov()[tag]
    f17 = lambda v: " ".join("%.17g" % x for x in v)
    for s in range(S):
        t += w.dt_of(s) if s else 0.0
        if s >= 2:
            wv, la = w.planar_imu(s)
            lines.append(f"J {t - 0.03:.9f} {f17(wv[tag])} {f17(cw)} {f17(la[tag])} {f17(ca)}")
            calls.append(("imu", t - 0.03, s))
        if s >= 3:
            f = w.px4flow(s)[tag]
            lines.append(f"X {t - 0.02:.9f} {f17(f[:4])} {int(f[4])}")
            if f[4] > 0:
                calls.append(("px4", t - 0.02, s))
        if s >= 4:
            lines.append(f"C {t - 0.01:.9f} {w.compass(s)[tag]:.17g}")
            calls.append(("compass", t - 0.01, s))
        for a_idx, mm in enumerate(w.ranges_mm(s)[tag]):
            lines.append(f"R {t:.9f} {100 + a_idx} 0 {int(mm)}.3 {s % 256} 0.0025")
        lines.append(f"F {t:.9f}")
        calls.append(("toa", t, s))
        lines.append(f"P {t + 0.02:.9f}")
        calls.append(("pose", t + 0.02, s))
    open(path, "w").write("\n".join(lines) + "\n")
    return calls


@pytest.mark.gpu
@pytest.mark.parametrize("fixed,A", [(1, 8), (0, 8), (1, 4), (0, 4)])
def test_replay_algorithm_kf_matches_oracle(tmp_path, fixed, A):
    """A = 4 with the fixed height of config_uwb.xml is BASELINE configs[0]: one tag, four anchors, the 2-D
    fixed-height EKF, driven through the launch-file parameter surface."""
    S, tag = 40, 2
    w = Workload(8, A)
    trace = str(tmp_path / "trace.txt")
    calls = _write_trace(trace, w, S, tag)
    files = {}
    for k, v in XML.items():
        files[k] = str(tmp_path / (k + ".xml"))
        open(files[k], "w").write(v)
    p0 = w.init_positions()[tag]
    out = subprocess.run([REPLAY, "algorithm:=ALGORITHM_KF", f"useStartPosition:={fixed}", "initAngle:=0.3",
                          f"initPositionX:={p0[0]:.17g}", f"initPositionY:={p0[1]:.17g}", "initPositionZ:=5.5",
                          "accelNoise:=0.5", "jolt:=0.5", "usePX4Flow:=1", "useIMU:=1", "useMAG:=1"] +
                         [f"{k}:={v}" for k, v in files.items()] + [trace],
                         capture_output=True, text=True, check=True).stdout
    got = np.array([[float(v) for v in ln.split()[2:]] for ln in out.splitlines() if ln.startswith("P")])
    assert got.shape == (S, 7) and np.all(got[:, 0] == 1)

    # the same call sequence through the oracle; initAngle is only read with useStartPosition = 1
    cfg = dict(CFG, init_angle=0.3 if fixed else 0.0)
    w1 = Workload(1, A, tag0=tag)
    orc = PlanarOracle(w1, cfg, p0[None] if fixed else None)
    cw, ca = np.eye(3).reshape(1, 9) * 1e-4, w1.accel_cov()
    last, want = None, []
    for kind, t, s in calls:
        if kind == "pose":
            pos, cov, _, _ = orc.get_pose(t - last)
            want.append([pos[0, 0], pos[0, 1], pos[0, 2], cov[0, 0, 0], cov[0, 1, 1], cov[0, 2, 2]])
            continue
        dt = 0.1 if last is None else t - last
        last = t
        if kind == "imu":
            wv, la = w1.planar_imu(s)
            orc.step_planar_imu(wv, cw, la, ca, dt)
        elif kind == "px4":
            orc.step_px4flow(w1.px4flow(s), dt)
        elif kind == "compass":
            orc.step_compass(w1.compass(s), dt)
        else:
            orc.step_toa(w1.ranges_mm(s), w1.err_est(), dt)
    want = np.array(want)
    ok = np.isfinite(want[:, 0])
    assert ok.sum() >= S - 2
    np.testing.assert_allclose(got[ok, 1:4], want[ok, :3], rtol=0, atol=1e-8)
    np.testing.assert_allclose(got[ok, 4:7], want[ok, 3:6], rtol=1e-6, atol=1e-12)


@pytest.mark.gpu
def test_replay_algorithm_kf_init_fails_without_configuration(tmp_path):
    trace = str(tmp_path / "t.txt")
    open(trace, "w").write("A 1 0 0 0\n")
    r = subprocess.run([REPLAY, "algorithm:=ALGORITHM_KF", trace], capture_output=True, text=True)
    assert r.returncode == 1 and "init() failed" in r.stderr


class _RefPlanarNode:
    """PosGenerator + KalmanFilter for ONE tag: the table logic of processRangingNow (Posgenerator.cpp:201-281) in
    front of the oracle, sensor callbacks passed straight through (:99-141). Time is passed in."""

    def __init__(self, w1, cfg, init):
        self.orc = PlanarOracle(w1, cfg, init)
        self.A = w1.n_anchors
        self.value = -np.ones((256, self.A), dtype=np.int64)
        self.err = np.zeros((256, self.A))
        self.count = np.zeros(256, dtype=np.int64)
        self.seq, self.started, self.last = -1, False, 0.0

    def lag(self, now):
        dt = now - self.last if self.started else 0.1
        self.last, self.started = now, True
        return dt

    def on_ranging(self, now, a, rng, err, seq):
        mm = int(np.floor(rng))
        if self.seq == seq:
            self.count[seq] += 1
            self.value[seq, a] = mm
            if err > 0.0:
                self.err[seq, a] = err
            return
        if self.seq != -1 and self.count[self.seq] >= 1:
            row = np.where(self.value[self.seq] > 0, self.value[self.seq], 0).astype(np.int32)
            self.orc.step_toa(row[None], self.err[self.seq][None].copy(), self.lag(now))
        self.value[seq, 0], self.err[seq, 0] = -1, 0.0
        self.count[seq], self.seq = 1, seq
        self.value[seq, a], self.err[seq, a] = mm, err


@pytest.mark.gpu
def test_batched_node_with_sensors_matches_per_tag_reference_nodes(tmp_path):
    """Three vehicles on one GPU handle (tagIds:=...), their ranging and sensor messages interleaved."""
    S, tags = 30, [0x10, 0x11, 0x12]
    w = Workload(len(tags), 8)
    files = {}
    for k, v in XML.items():
        files[k] = str(tmp_path / (k + ".xml"))
        open(files[k], "w").write(v)
    cw = np.eye(3).ravel() * 1e-4
    f17 = lambda v: " ".join("%.17g" % x for x in v)
    lines = [f"A {100 + a} {x:.17g} {y:.17g} {z:.17g}" for a, (x, y, z) in enumerate(w.anchors)]
    for s in range(S):
        t = 5.0 + 0.04 * s  # 40 ms epochs: the 50 ms ranging timer never fires
        ca = w.accel_cov()
        for i, tag in enumerate(tags):
            off = 0.001 * i
            if s >= 2:
                wv, la = w.planar_imu(s)
                lines.append(f"j {t - 0.03 + off:.9f} {tag:x} {f17(wv[i])} {f17(cw)} {f17(la[i])} {f17(ca[i])}")
            if s >= 3:
                f = w.px4flow(s)[i]
                lines.append(f"x {t - 0.02 + off:.9f} {tag:x} {f17(f[:4])} {int(f[4])}")
            if s >= 4:
                c = w.compass(s)[i]
                lines.append(f"c {t - 0.01 + off:.9f} {tag:x} {c:.17g}")
        for a_idx in range(8):  # anchor-major: the tags' range messages interleave
            for i, tag in enumerate(tags):
                mm = w.ranges_mm(s)[i, a_idx]
                lines.append(f"R {t + 0.001 * i:.9f} {100 + a_idx} {tag} {int(mm)}.6 {s % 256} 0.0025")
        lines.append(f"P {t + 0.02:.9f}")
    trace = str(tmp_path / "trace.txt")
    open(trace, "w").write("\n".join(lines) + "\n")
    p0 = w.init_positions()
    # one shared initPosition for the bank is all the launch parameters offer; the per-tag starts differ, so start
    # every vehicle from tag 0's position in the reference nodes too
    nodes0 = [_RefPlanarNode(Workload(1, 8, tag0=i), CFG, p0[0][None]) for i in range(len(tags))]
    out = subprocess.run([REPLAY, "algorithm:=ALGORITHM_KF", "useStartPosition:=1", "initAngle:=0.3",
                          f"initPositionX:={p0[0][0]:.17g}", f"initPositionY:={p0[0][1]:.17g}", "initPositionZ:=0",
                          "usePX4Flow:=1", "useIMU:=1", "useMAG:=1", "tagIds:=" + ",".join(f"{t:x}" for t in tags)] +
                         [f"{k}:={v}" for k, v in files.items()] + [trace],
                         capture_output=True, text=True, check=True).stdout
    got = np.array([[float(v) for v in ln.split()[3:]] for ln in out.splitlines() if ln.startswith("P")])
    assert got.shape == (S * len(tags), 7)
    # replay the same event list through reference nodes that share tag 0's start
    want = []
    for ln in lines:
        k, rest = ln[0], ln.split()[1:]
        if k == "R":
            t, a, tag, mm, seq, e = float(rest[0]), int(rest[1]) - 100, int(rest[2]), float(rest[3]), int(rest[4]), float(rest[5])
            nodes0[tags.index(tag)].on_ranging(t, a, mm, e, seq)
        elif k in "jxc":
            t, i = float(rest[0]), tags.index(int(rest[1], 16))
            v = np.array([float(x) for x in rest[2:]])
            n = nodes0[i]
            if k == "j":
                n.orc.step_planar_imu(v[None, 0:3], v[None, 3:12], v[None, 12:15], v[None, 15:24], n.lag(t))
            elif k == "x" and v[4] > 0 and v[3] > 0:
                n.orc.step_px4flow(v[None], n.lag(t))
            elif k == "c":
                n.orc.step_compass(v[:1], n.lag(t))
        elif k == "P":
            t = float(rest[0])
            for n in nodes0:
                pos, cov, _, st = n.orc.get_pose(t - n.last if n.started else 0.0)
                want.append([float(st[0] != 16), pos[0, 0], pos[0, 1], pos[0, 2], cov[0, 0, 0], cov[0, 1, 1], cov[0, 2, 2]])
    want = np.array(want)
    np.testing.assert_array_equal(got[:, 0], want[:, 0])
    ok = want[:, 0] == 1
    assert ok.sum() > S * len(tags) - 10
    np.testing.assert_allclose(got[ok, 1:4], want[ok, 1:4], rtol=0, atol=1e-8)
    np.testing.assert_allclose(got[ok, 4:7], want[ok, 4:7], rtol=1e-6, atol=1e-12)

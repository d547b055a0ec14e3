"""The build-time performance rules of roskfpos_amd/csrc (make check = tools/check_scratch.py): no scratch access inside
a loop, and the kernels written to share a SIMD two at a time still fit twice. Neither can be seen by a parity test --
a kernel that lost its second wavefront computes the same bits 25-40 % slower -- so the rules themselves are tested:
they hold for the library as built, and they do fire."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "roskfpos_amd", "csrc")
TOOL = os.path.join(ROOT, "tools", "check_scratch.py")


def _built():
    if not (os.path.exists(os.path.join(CSRC, "libkfpos_hip.so")) and os.path.exists(os.path.join(CSRC, "build.log"))):
        import __graft_entry__
        __graft_entry__.build()


def _run(*rules):
    cmd = [sys.executable, TOOL, os.path.join(CSRC, "libkfpos_hip.so"), "--resource-log", os.path.join(CSRC, "build.log")]
    for r in rules:
        cmd += ["--min-occupancy", r]
    return subprocess.run(cmd, capture_output=True, text=True)


def test_make_check_holds_for_the_library_as_built():
    _built()
    res = subprocess.run(["make", "-s", "-C", CSRC, "check"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "3 occupancy rule(s) hold" in res.stdout


def test_occupancy_rule_fires():
    _built()
    ok = _run("k_step_toa6_w2=2")
    assert ok.returncode == 0 and "1 occupancy rule(s) hold" in ok.stdout
    # the 9-state kernel holds its state in 256 VGPRs + AGPRs: one wavefront per SIMD, and the rule says so
    bad = _run("k_step_imu9=2")
    assert bad.returncode == 1 and "1 wavefront(s) per SIMD, 2 required" in bad.stdout
    # a rule that matches nothing is an error too (a renamed kernel must not switch its rule off)
    gone = _run("k_no_such_kernel=2")
    assert gone.returncode == 1 and "no kernel of that name" in gone.stdout

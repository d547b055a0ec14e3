#!/usr/bin/env python3
"""Timing variants of the restructured 9-state iteration (to find what the per-lane path lost against the previous
commit): built from patched copies of the product source into tools/exp/_build/ (git-ignored)."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "roskfpos_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "exp", "_build")


def variant(name, edit):
    d = os.path.join(OUT, "v_" + name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    for f in os.listdir(CSRC):
        if f.endswith((".h", ".hip")):
            shutil.copy(os.path.join(CSRC, f), d)
    hip = open(os.path.join(d, "kfpos_hip.hip")).read().replace('#include "../../include/kfpos.h"', f'#include "{ROOT}/include/kfpos.h"')
    open(os.path.join(d, "kfpos_hip.hip"), "w").write(hip)
    p = os.path.join(d, "kfpos_core_imu9.h")
    src = edit(open(p).read())
    open(p, "w").write(src)
    lib = os.path.join(OUT, f"libkfpos_{name}.so")
    res = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                          "-Rpass-analysis=kernel-resource-usage", "-o", lib, os.path.join(d, "kfpos_hip.hip")],
                         capture_output=True, text=True)
    if res.returncode:
        sys.stderr.write(res.stderr[-3000:]); raise SystemExit(1)
    cur = None
    for line in res.stderr.splitlines():
        if "Function Name:" in line:
            cur = line.split("Function Name:")[1].split()[0]
        if cur and "k_step_imu9IdfLi8ELb1" in cur and ("ScratchSize" in line):
            print(name, line.split("remark:")[1].split("[-R")[0].strip())
    print("built", lib)


def rep(s, a, b):
    assert s.count(a) == 1, a
    return s.replace(a, b)


def plain_loop(s):   # the lane-divergent loop of the generic branch on the device too (no ballot, no pairs)
    return rep(s, "    if constexpr (RANGING && SC::NA == 8 && SC::CHUNK == 0 && !SC::COOP) {\n        /* every lane of the wavefront is here (none left",
               "    if constexpr (false && RANGING && SC::NA == 8 && SC::CHUNK == 0 && !SC::COOP) {\n        /* every lane of the wavefront is here (none left")


def plain_loop_nospread(s):
    s = plain_loop(s)
    return s


def plain_loop_hoist(s):  # + parked values read without the opaque index: the compiler may hoist them (AGPRs)
    s = plain_loop(s)
    return rep(s, "    const int z = kf_opaque_zero(); /* (read in every trip, not once before the loop) */", "    const int z = 0;")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["plain", "plainhoist"]
    if "plain" in which: variant("plain", plain_loop)
    if "plainhoist" in which: variant("plainhoist", plain_loop_hoist)

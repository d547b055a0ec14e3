#!/usr/bin/env python3
"""Timing variant: the 9-state per-lane sweep summing anchors 0..7 in one chain (no (0-3)+(4-7) split): what the sum order
kept for the opt-in lane pairs costs the default path. Output: tools/exp/_build/libkfpos_seqorder.so"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "roskfpos_amd", "csrc"); OUT = os.path.join(ROOT, "tools", "exp", "_build")
d = os.path.join(OUT, "v_seqorder"); shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
for f in os.listdir(CSRC):
    if f.endswith((".h", ".hip")): shutil.copy(os.path.join(CSRC, f), d)
hip = open(os.path.join(d, "kfpos_hip.hip")).read().replace('#include "../../include/kfpos.h"', f'#include "{ROOT}/include/kfpos.h"')
open(os.path.join(d, "kfpos_hip.hip"), "w").write(hip)
p = os.path.join(d, "kfpos_core_imu9.h"); s = open(p).read()
a = "    } else if constexpr (SC::NA == 8 && SC::CHUNK == 0 && !SC::COOP) {"
assert s.count(a) == 1
s = s.replace(a, "    } else if constexpr (false && SC::NA == 8 && SC::CHUNK == 0 && !SC::COOP) {")
open(p, "w").write(s)
res = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=on", "-o",
                      os.path.join(OUT, "libkfpos_seqorder.so"), os.path.join(d, "kfpos_hip.hip")], capture_output=True, text=True)
print(res.returncode, res.stderr[-500:])

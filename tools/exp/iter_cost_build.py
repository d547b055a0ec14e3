#!/usr/bin/env python3
"""What does one IEKF iteration of the 9-state kernel cost, and how much of it is the anchor sweep? Timing variants of
the product source (results are WRONG by construction, only their duration is of interest; nothing here ships):
  libkfpos_cap10.so    iteration cap 10 instead of 20: (T_product - T_cap10) / 10 = one iteration of the capped lane
  libkfpos_sweep2.so   the anchor sweep of every iteration is run twice (second copy behind an opaque copy of the
                       position, folded in with weight 0): (T_sweep2 - T_product) / 20 = one 8-anchor sweep
Output: tools/exp/_build/ (git-ignored).   python tools/exp/iter_cost_build.py && gpurun -- bash tools/gpu_ab3.sh ...
"""
import os
import shutil
import subprocess
import sys

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
    s = edit(open(p).read())
    open(p, "w").write(s)
    lib = os.path.join(OUT, f"libkfpos_{name}.so")
    res = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                          "-Rpass-analysis=kernel-resource-usage", "-o", lib, os.path.join(d, "kfpos_hip.hip")],
                         capture_output=True, text=True)
    if res.returncode:
        sys.stderr.write(res.stderr[-3000:])
        raise SystemExit(1)
    cur = None
    for line in res.stderr.splitlines():
        if "Function Name:" in line:
            cur = line.split("Function Name:")[1].split()[0]
        if cur and "k_step_imu9IdfLi8ELb1" in cur and ("ScratchSize" in line or "VGPRs:" in line or "AGPRs:" in line):
            print(name, line.split("remark:")[1].split("[-R")[0].strip())
    print("built", lib)


def cap10(s):
    a = "iekf9_info<false, RANGING>(xhat, binv, park.stride, sc, pr, imu, 20, 1e-4, o);"
    b = "iekf9<false, RANGING>(xhat, tg.P, sc, pr, imu, 20, 1e-4, o);"
    assert s.count(a) == 1 and s.count(b) == 1
    return s.replace(a, a.replace(", 20,", ", 10,")).replace(b, b.replace(", 20,", ", 10,"))


def sweep2(s):
    i = s.index("KFPOS_FN void iekf9_info(")
    head, body = s[:i], s[i:]
    old = "        const double m[6] = {m0, m1, m2, m3, m4, m5};\n"
    assert body.count(old) == 1
    extra = '''        if constexpr (RANGING) { /* EXPERIMENT: a second, independent copy of the sweep */
            double q0 = exp_opaque(p[0]), q1 = exp_opaque(p[1]), q2 = exp_opaque(p[2]);
            double e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0, e6 = 0, e7 = 0, e8 = 0, e9 = 0;
            for_anchors<SC>(pr, [&](int a) {
                const double dx = q0 - pr.anchors[3 * a], dy = q1 - pr.anchors[3 * a + 1], dz = q2 - pr.anchors[3 * a + 2];
                double d, invd;
                kf_sqrt_rsqrt_sweep(dx * dx + dy * dy + dz * dz, d, invd);
                const double w = sc.W(a), y = sc.R(a) - d;
                const double yw = y * w;
                e9 += y * yw;
                const double gx = dx * invd, gy = dy * invd, gz = dz * invd;
                e6 += gx * yw; e7 += gy * yw; e8 += gz * yw;
                const double wx = w * gx, wy = w * gy, wz = w * gz;
                e0 += wx * gx; e1 += wx * gy; e2 += wx * gz;
                e3 += wy * gy; e4 += wy * gz; e5 += wz * gz;
            });
            const double z = exp_opaque(0.0);
            c += z * e9; m0 += z * e0; m1 += z * e1; m2 += z * e2; m3 += z * e3; m4 += z * e4; m5 += z * e5;
            u0 += z * e6; u1 += z * e7; u2 += z * e8;
        }
'''
    helper = '''KFPOS_FN double exp_opaque(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}
'''
    j = head.rindex("template <bool DIAG, bool RANGING, class SC>")
    return head[:j] + helper + head[j:] + body.replace(old, extra + old)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    variant("cap10", cap10)
    variant("sweep2", sweep2)

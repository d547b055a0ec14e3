#!/usr/bin/env python3
"""Experiment build for the round-1 "scratch-spilling 16-anchor kernel returns wrong results on partial wavefronts"
finding (kfpos_hip.hip, static_anchors()): re-creates the register-resident 16-anchor instantiations that commit
ab16bcb dropped, in two variants of the SAME kernel text,
  libkfpos_reg16_early.so   lanes beyond the bank leave at the top (`if (t >= T) return;`), as every product kernel does
  libkfpos_reg16_clamp.so   those lanes stay: they redo the last tag's arithmetic and only their stores are masked,
                            so the whole kernel runs with a full EXEC mask
The product source is patched in memory; nothing here ships. Output: tools/exp/_build/ (git-ignored *.so).
    python tools/exp/reg16_build.py && gpurun -- python tools/exp/reg16_run.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "roskfpos_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "exp", "_build")


def patched(clamp: bool) -> str:
    s = open(os.path.join(CSRC, "kfpos_hip.hip")).read()
    s = s.replace('#include "kfpos_core.h"', f'#include "{CSRC}/kfpos_core.h"')
    s = s.replace('#include "../../include/kfpos.h"', f'#include "{ROOT}/include/kfpos.h"')

    def rep(old, new, count=1):
        nonlocal s
        assert s.count(old) >= 1, old
        s = s.replace(old, new, count)

    rep("    if (h->cfg.max_anchors == 16 && h->cfg.model == KFPOS_MODEL_TOA) return -16;",
        "    if (h->cfg.max_anchors == 16 && h->cfg.model == KFPOS_MODEL_TOA) return 16; /* EXPERIMENT */")
    rep("    if (as == -16) return heur == 1 ?",
        "    if (as == 16) return heur == 1 ? k_step_toa6<SYMM, REAL, MREAL, 16, 1> : k_step_toa6<SYMM, REAL, MREAL, 16>;\n"
        "    if (as == -16) return heur == 1 ?")
    if clamp:
        a = s.index("template <bool SYMM, typename REAL, typename MREAL, int AS, int HEUR = 2>\n__global__")
        b = s.index("/* ------------------------------------------------------------------ 6-state step kernel, small batches */")
        k = s[a:b]
        k = k.replace("    const size_t t = (size_t)blockIdx.x * WAVE + lane;\n    if (t >= (size_t)a.T) return;",
                      "    const size_t t_raw = (size_t)blockIdx.x * WAVE + lane;\n    const bool live = t_raw < (size_t)a.T;\n"
                      "    const size_t t = live ? t_raw : (size_t)a.T - 1; /* EXPERIMENT: full EXEC, stores masked */")
        k = k.replace("        if (a.status) a.status[t] = ST_SKIPPED;\n        return;", "        if (a.status && live) a.status[t] = ST_SKIPPED;\n        return;")
        k = k.replace("        if (a.traj) { /* the pose", "        if (a.traj && live) { /* the pose")
        k = k.replace("    bool fin = true;\n#pragma unroll\n    for (int k = 0; k < 3; ++k) {\n        (a.pos + k * T)[t32] = tg.pos[k];",
                      "    bool fin = true;\n    if (!live) return; /* all arithmetic is done: only the stores remain */\n#pragma unroll\n    for (int k = 0; k < 3; ++k) {\n        (a.pos + k * T)[t32] = tg.pos[k];")
        assert "t_raw" in k and "if (!live) return;" in k and "a.traj && live" in k
        s = s[:a] + k + s[b:]
    return s


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, clamp in (("early", False), ("clamp", True)):
        src = os.path.join(OUT, f"kfpos_reg16_{name}.hip")
        open(src, "w").write(patched(clamp))
        lib = os.path.join(OUT, f"libkfpos_reg16_{name}.so")
        res = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                              "-Rpass-analysis=kernel-resource-usage", "-o", lib, src], capture_output=True, text=True)
        if res.returncode:
            sys.stderr.write(res.stderr[-3000:])
            raise SystemExit(1)
        cur = None
        for line in res.stderr.splitlines():
            if "Function Name:" in line:
                cur = line.split("Function Name:")[1].split()[0]
            if cur and "Li16E" in cur and ("ScratchSize" in line or "SGPRs Spill" in line or "VGPRs Spill" in line):
                print(name, cur[24:70], line.split("remark:")[1].split("[-R")[0].strip())
        print("built", lib)


if __name__ == "__main__":
    main()

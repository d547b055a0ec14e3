#!/usr/bin/env python3
"""Build-time performance rule for libkfpos_hip.so: no kernel touches scratch memory inside a loop.

Every kernel keeps its per-tag state in VGPRs / AGPRs / LDS. A scratch (private memory) access inside the epoch loop or
an iteration loop costs a memory round trip per trip with nothing to hide it behind (one wavefront per SIMD), so it is a
build error; a few values parked in scratch ONCE per launch -- the compiler does that to a lane's tag offsets across the
9-state kernel's epoch loop -- cost nothing measurable and are tolerated up to 64 bytes per lane. This is a rule about
speed, not correctness.

The check disassembles the gfx950 code object, rebuilds each kernel's control-flow graph and fails on any scratch_*
instruction in a basic block that lies on a cycle. usage: check_scratch.py LIB [--max-bytes 64] [--resource-log build.log]
                                                         [--min-occupancy 'REGEX=N' ...]

--min-occupancy: kernels whose (mangled) name matches REGEX must reach N wavefronts per SIMD according to the compiler's
resource report. The kernels that serve banks with more wavefronts than the chip has SIMDs are written to share a SIMD
two at a time; two registers too many drop them to one and cost 25-40 % with every test still green (it happened: the
48-bit-covariance 16-anchor kernel, round 3), so the build checks what no parity test can.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="kfpos_co_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        cos = sorted(f for f in os.listdir(tmp) if "gfx950" in f)  # one code object per translation unit
        if not cos:
            raise SystemExit("no gfx950 code object in " + lib)
        return "\n".join(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, co)], check=True,
                                        capture_output=True, text=True).stdout for co in cos)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def scratch_in_cycles(ins):
    """Number of scratch_* instructions that lie on a cycle of the function's control-flow graph.
    ins: [(address, opcode, branch target offset from the function start or None)]."""
    base = ins[0][0]
    addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
    leaders = {0}
    for i, (a, op, off) in enumerate(ins):
        if off is not None:
            t = addr_index.get(base + off)
            if t is not None:
                leaders.add(t)
            if i + 1 < len(ins):
                leaders.add(i + 1)
        if op == "s_endpgm" and i + 1 < len(ins):
            leaders.add(i + 1)
    starts = sorted(leaders)
    block_of = {}
    for b, st in enumerate(starts):
        en = starts[b + 1] if b + 1 < len(starts) else len(ins)
        for i in range(st, en):
            block_of[i] = b
    succ = [[] for _ in starts]
    for b, st in enumerate(starts):
        en = (starts[b + 1] if b + 1 < len(starts) else len(ins)) - 1
        a, op, off = ins[en]
        if off is not None:
            t = addr_index.get(base + off)
            if t is not None:
                succ[b].append(block_of[t])
        if op not in ("s_branch", "s_endpgm") and en + 1 < len(ins):
            succ[b].append(block_of[en + 1])
    # Tarjan's strongly connected components, iterative
    n = len(starts)
    index, low, on, comp = [None] * n, [0] * n, [False] * n, [None] * n
    stack, counter, ncomp = [], 0, 0
    for root in range(n):
        if index[root] is not None:
            continue
        work = [(root, 0)]
        while work:
            v, pi = work.pop()
            if pi == 0:
                index[v] = low[v] = counter
                counter += 1
                stack.append(v)
                on[v] = True
            recurse = False
            for k in range(pi, len(succ[v])):
                w = succ[v][k]
                if index[w] is None:
                    work.append((v, k + 1))
                    work.append((w, 0))
                    recurse = True
                    break
                if on[w]:
                    low[v] = min(low[v], index[w])
            if recurse:
                continue
            if low[v] == index[v]:
                while True:
                    w = stack.pop()
                    on[w] = False
                    comp[w] = ncomp
                    if w == v:
                        break
                ncomp += 1
            if work:
                u = work[-1][0]
                low[u] = min(low[u], low[v])
    size = {}
    for c_ in comp:
        size[c_] = size.get(c_, 0) + 1
    cyclic = {b for b in range(n) if size[comp[b]] > 1 or b in succ[b]}
    return sum(1 for i, (a, op, _) in enumerate(ins) if op.startswith("scratch_") and block_of[i] in cyclic)


def check(lib, max_bytes=64, resource_log=None, min_occupancy=()):
    problems = []
    text = disassemble(lib)
    func, insns = None, {}
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            func = m.group(1)
            insns[func] = []
            continue
        m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-Fa-f ]+(?:<\S+?(?:\+0x([0-9a-fA-F]+))?>)?\s*$", line)
        if m and func:
            off = None
            if m.group(1).startswith(("s_cbranch", "s_branch")) and "<" in line:
                off = int(m.group(4), 16) if m.group(4) else 0
            insns[func].append((int(m.group(3), 16), m.group(1), off))
    n_scratch = 0
    for func, ins in insns.items():
        scratch = [a for a, op, _ in ins if op.startswith("scratch_")]
        if not scratch:
            continue
        n_scratch += len(scratch)
        inside = scratch_in_cycles(ins)
        if inside:
            problems.append(f"{func}: {inside} scratch access(es) inside a loop")
    if resource_log and os.path.exists(resource_log):
        name, seen = None, {}
        for line in open(resource_log):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and int(m.group(1)) > max_bytes:
                problems.append(f"{name}: {m.group(1)} bytes/lane of scratch (limit {max_bytes})")
            m = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", line)
            if m and name:
                for pattern, need in min_occupancy:
                    if re.search(pattern, name):
                        seen[pattern] = seen.get(pattern, 0) + 1
                        if int(m.group(1)) < need:
                            problems.append(f"{name}: {m.group(1)} wavefront(s) per SIMD, {need} required")
        for pattern, _ in min_occupancy:
            if not seen.get(pattern):
                problems.append(f"--min-occupancy {pattern}: no kernel of that name in {resource_log}")
    elif min_occupancy:
        problems.append("--min-occupancy needs --resource-log")
    return problems, n_scratch


if __name__ == "__main__":
    args = sys.argv[1:]
    mb, log = 64, None
    if "--max-bytes" in args:
        i = args.index("--max-bytes"); mb = int(args[i + 1]); del args[i:i + 2]
    if "--resource-log" in args:
        i = args.index("--resource-log"); log = args[i + 1]; del args[i:i + 2]
    occ = []
    while "--min-occupancy" in args:
        i = args.index("--min-occupancy"); pat, need = args[i + 1].rsplit("=", 1); occ.append((pat, int(need))); del args[i:i + 2]
    probs, n = check(args[0], mb, log, occ)
    if probs:
        print("\n".join(probs))
        sys.exit(1)
    print(f"no kernel touches scratch inside a loop ({n} scratch instructions in the library, all outside loops)"
          + (f"; {len(occ)} occupancy rule(s) hold" if occ else ""))

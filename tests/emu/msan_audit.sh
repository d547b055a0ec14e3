#!/bin/bash
# MemorySanitizer audit of the kernel body (see msan_driver.cpp). Needs ROCm's clang (its msan runtime); CPU only.
set -e
cd "$(dirname "$0")"
CLANG=${CLANG:-/opt/rocm/lib/llvm/bin/clang++}
OUT=${TMPDIR:-/tmp}/kfpos_msan_audit
$CLANG -std=c++17 -O1 -g -fsanitize=memory -fsanitize-memory-track-origins=2 -fno-omit-frame-pointer \
    -DKFE_MSAN -Wno-unknown-pragmas -Wno-pass-failed -o "$OUT" msan_driver.cpp kfpos_emu.cpp
"$OUT"

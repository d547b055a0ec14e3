#!/bin/bash
# MemorySanitizer audit of the kernel body (see msan_driver.cpp). Needs ROCm's clang (its msan runtime); CPU only.
set -e
cd "$(dirname "$0")"
CLANG=${CLANG:-/opt/rocm/lib/llvm/bin/clang++}
OUT=${TMPDIR:-/tmp}/kfpos_msan_audit
# MSAN_ORIGINS=1 adds origin tracking (slower build; use it to locate a report)
ORIGINS=${MSAN_ORIGINS:+-fsanitize-memory-track-origins=2 -g}
$CLANG -std=c++17 -O1 -fsanitize=memory $ORIGINS -fno-omit-frame-pointer \
    -DKFE_MSAN -Wno-unknown-pragmas -Wno-pass-failed -o "$OUT" msan_driver.cpp kfpos_emu.cpp
if "$OUT" --selftest > "$OUT.selftest.log" 2>&1; then
    echo "msan selftest: the sanitizer let a branch on a poisoned value pass -- audit void"; exit 2
fi
grep -q "use-of-uninitialized-value" "$OUT.selftest.log" || { echo "msan selftest: no report"; cat "$OUT.selftest.log"; exit 2; }
echo "msan selftest: report raised as expected"
"$OUT"

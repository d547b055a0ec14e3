"""MemorySanitizer audit of the per-lane kernel body (tests/emu/msan_driver.cpp): no read-before-write on any
lane-private array over ragged epochs, poisoned errorEstimations of absent ranges, a poisoned working-weight scratch,
skipped lanes and bank sizes that are not a multiple of 64. Needs ROCm's clang with its msan runtime; CPU only."""
import glob
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CLANG = os.environ.get("CLANG", "/opt/rocm/lib/llvm/bin/clang++")


def test_kernel_body_reads_no_uninitialised_value():
    if not os.path.exists(CLANG) or not glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.msan-x86_64.a"):
        pytest.skip("no clang with a MemorySanitizer runtime in this image")
    r = subprocess.run(["bash", os.path.join(HERE, "emu", "msan_audit.sh")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "msan selftest: report raised as expected" in r.stdout
    assert "msan audit: clean" in r.stdout

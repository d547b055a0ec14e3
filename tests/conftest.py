import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_test_infrastructure():
    """Build the oracle and the host emulation of the kernel body if they are missing."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    emu_dir = os.path.join(ROOT, "tests", "emu")
    so, src = os.path.join(emu_dir, "libkfpos_emu.so"), os.path.join(emu_dir, "kfpos_emu.cpp")
    core = os.path.join(ROOT, "roskfpos_amd", "csrc", "kfpos_core.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(core)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                               "-o", so, src])
    yield


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False

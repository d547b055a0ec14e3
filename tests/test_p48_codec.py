"""KFPOS_STORE_P48's codec (roskfpos_amd/csrc/kfpos_p48.h) compiled for the host: the numpy mirror the GPU tests use
equals it bit for bit, the rounding is to nearest on 40 significant bits, values on that grid pass unchanged, and the
special cases are what the header says."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from p48 import p48_round_trip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "%s/roskfpos_amd/csrc/kfpos_p48.h"
extern "C" void round_only(const double *v, double *o, int n) { for (int i = 0; i < n; ++i) o[i] = kfpos_p48_round(v[i]); }
extern "C" void codec(const double *v, double *o, uint32_t *hi, uint16_t *lo, int n) {
    for (int i = 0; i < n; ++i) {
        kfpos_p48_encode(kfpos_p48_round(v[i]), &hi[i], &lo[i]);
        o[i] = kfpos_p48_decode(hi[i], lo[i]);
    }
}
''' % ROOT


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("p48")
    (d / "c.cpp").write_text(SRC)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(d / "c.so"), str(d / "c.cpp")])
    return C.CDLL(str(d / "c.so"))


def _codec(lib, v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    o, hi, lo = np.zeros_like(v), np.zeros(v.size, np.uint32), np.zeros(v.size, np.uint16)
    lib.codec(v.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p),
              lo.ctypes.data_as(C.c_void_p), v.size)
    return o, hi, lo


def test_codec_rounds_to_nearest_on_40_bits_and_the_mirror_agrees(lib):
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.normal(size=300000) * 10.0 ** rng.integers(-30, 31, 300000),
                        [0.0, -0.0, 2.0 ** -126, -2.0 ** -126, 3.0e38, 1.0, 1.9999999999999998, 1 - 2.0 ** -41]])
    o, hi, lo = _codec(lib, v)
    assert np.array_equal(o.view(np.uint64), p48_round_trip(v).view(np.uint64))
    nz = np.abs(v) >= 2.0 ** -126                                                # (smaller magnitudes: signed zero, below)
    assert (np.abs(o[nz] - v[nz]) / np.abs(v[nz])).max() <= 2.0 ** -40          # half a step of a 40-bit significand
    assert np.all((o.view(np.uint64) & np.uint64(0x1FFF)) == 0)                 # 39 mantissa bits
    # the stored pair is the single that truncates the value + the next 16 mantissa bits
    f = hi.view(np.float32).astype(np.float64)
    assert np.all(np.abs(f) <= np.abs(o)) and np.all(np.abs(o[nz] - f[nz]) < np.abs(f[nz]) * 2.0 ** -23)
    o2, hi2, lo2 = _codec(lib, o)                                                # idempotent
    assert np.array_equal(o2.view(np.uint64), o.view(np.uint64)) and np.array_equal(hi2, hi) and np.array_equal(lo2, lo)


def test_codec_ties_go_to_even_and_carries_reach_the_exponent(lib):
    step = 2.0 ** -39                                                            # one step of the grid in [1, 2)
    v = np.array([1 + step / 2, 1 + 3 * step / 2, 1 + step / 2 + 2.0 ** -52, 2 - step / 2, -(2 - step / 2)])
    o, _, _ = _codec(lib, v)
    assert o.tolist() == [1.0, 1 + 2 * step, 1 + step, 2.0, -2.0]


def test_codec_special_values(lib):
    v = np.array([np.nan, np.inf, -np.inf, 1e-39, -1e-39, 2.0 ** -127, 0.0, -0.0])
    o, hi, lo = _codec(lib, v)
    assert np.isnan(o[:3]).all()                       # a NaN stays one; an infinity becomes one (inf - inf in the split)
    assert np.all(o[3:] == 0.0) and np.signbit(o[4]) and np.signbit(o[7]) and not np.signbit(o[3])
    assert np.all(lo[3:] == 0) and np.all((hi[3:] & 0x7FFFFFFF) == 0)
    assert np.array_equal(np.isnan(p48_round_trip(v)), np.isnan(o))

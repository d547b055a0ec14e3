"""numpy mirror of roskfpos_amd/csrc/kfpos_p48.h (KFPOS_STORE_P48: a covariance entry in 6 bytes -- sign, 8 exponent bits,
39 mantissa bits). tests/test_p48_codec.py holds it against the C functions themselves."""
import numpy as np


def p48_round_trip(v):
    """what kfpos_p48_decode(kfpos_p48_encode(kfpos_p48_round(v))) returns, elementwise"""
    v = np.ascontiguousarray(v, dtype=np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        c = v * 8193.0                                       # Veltkamp split, 53 - 13 = 40 significant bits
        r = c - (c - v)
        f = r.astype(np.float32)                             # to nearest; one step back where that went up = truncation
        h = f.view(np.uint32).copy()
        h -= (np.abs(f.astype(np.float64)) > np.abs(r)).astype(np.uint32)
        lo = (r.view(np.uint64) >> np.uint64(13)) & np.uint64(0xFFFF)
        tiny = (h & np.uint32(0x7F800000)) == 0              # below single's normal range: signed zero
        h[tiny] &= np.uint32(0x80000000)
        lo[tiny] = 0
        d = h.view(np.float32).astype(np.float64)
        return (d.view(np.uint64) | (lo << np.uint64(13))).view(np.float64)

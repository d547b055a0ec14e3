/*
 * kfpos_p48.h -- KFPOS_STORE_P48: a covariance entry in 6 bytes.
 *
 *   hi (32 bits) | lo (16 bits)   =   sign 1 | exponent 8 | mantissa 23 + 16 = 39
 *
 * i.e. the IEEE single that TRUNCATES the value, followed by the next 16 mantissa bits: single's exponent range
 * (1.2e-38 .. 3.4e38 -- covariance entries in m^2, m^2/s, ... live between 1e-30 and 1e10), 40 significant bits. The
 * upper 48 bits of the double, which this mode stored at first, spend 11 bits on the exponent and keep 37 significant
 * ones; the three bits matter because the 9-state filter multiplies a difference in P by ~1e5 in an epoch of capped steps
 * (DESIGN.md sections 3 and 5): over 2 048 tags x 2 000 epochs the old encoding stayed 3.8e-7 m RMS from the oracle with
 * single epochs up to 3e-6 m, above BASELINE's bar.
 *
 * Rounding is a Veltkamp split (three fp64 operations: cheaper in a kernel than integer arithmetic on the bit pattern),
 * round to nearest on 40 significant bits. It is applied (a) by the kernels between the epochs of a multi-epoch launch,
 * so that K epochs in one launch keep the bits of K launches of one epoch, (b) before every store, (c) by the host when it
 * writes a state (kfpos_set_state), (d) by the host build of the kernel body in tests/emu -- all through this one
 * function, so that they agree bit for bit. Values that are already on the 40-bit grid pass unchanged. NaN stays NaN; an
 * infinity becomes NaN (inf - inf inside the split): a covariance entry that overflowed is not a number the filter can use
 * either way, and the status word says so (KFPOS_ST_NONFINITE). Magnitudes below 2^-126 are stored as (signed) zero, above
 * single's range as garbage of that magnitude.
 */
#ifndef KFPOS_P48_H
#define KFPOS_P48_H

#include <stdint.h>
#include <string.h>

#ifndef KFPOS_HD
#define KFPOS_HD
#endif

/* v on the grid of 40 significant bits, round to nearest (Veltkamp, s = 53 - 40 = 13) */
KFPOS_HD inline double kfpos_p48_round(double v) {
    double c = v * 8193.0; /* 2^13 + 1 */
    /* the product is rounded before it is used: no fma(v, 8193, -v), whatever -ffp-contract and -march say */
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(c));
#elif defined(__GNUC__)
    __asm__("" : "+m"(c));
#endif
    const double t = c - v;
    return c - t;
}

/* r = kfpos_p48_round(something) -> the two stored words */
KFPOS_HD inline void kfpos_p48_encode(double r, uint32_t *hi, uint16_t *lo) {
    const float f = (float)r; /* to nearest; one step back (towards zero) where that went up = the truncation */
    uint32_t h;
    uint64_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    h = __float_as_uint(f);
    u = (uint64_t)__double_as_longlong(r);
#else
    memcpy(&h, &f, 4);
    memcpy(&u, &r, 8);
#endif
    const double back = (double)f;
    const double ab = back < 0.0 ? -back : back, ar = r < 0.0 ? -r : r;
    h -= (ab > ar) ? 1u : 0u;
    uint32_t l = (uint32_t)(u >> 13) & 0xFFFFu;
    if ((h & 0x7F800000u) == 0u) { /* below single's normal range: signed zero */
        h &= 0x80000000u;
        l = 0u;
    }
    *hi = h;
    *lo = (uint16_t)l;
}

KFPOS_HD inline double kfpos_p48_decode(uint32_t hi, uint32_t lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double d = (double)__uint_as_float(hi);
    return __longlong_as_double(__double_as_longlong(d) | (long long)((uint64_t)lo << 13));
#else
    float f;
    memcpy(&f, &hi, 4);
    double d = (double)f;
    uint64_t u;
    memcpy(&u, &d, 8);
    u |= (uint64_t)lo << 13;
    memcpy(&d, &u, 8);
    return d;
#endif
}

#endif /* KFPOS_P48_H */

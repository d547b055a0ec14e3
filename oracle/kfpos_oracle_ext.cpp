/*
 * kfpos_oracle_ext.cpp -- the SAME restatement (kfpos_oracle.cpp, every line of it) evaluated in x87 extended precision:
 * 64 mantissa bits instead of 53, unit roundoff 5.4e-20 instead of 1.1e-16.
 *
 * TEST INFRASTRUCTURE ONLY. PARITY UNPINNED (kfpos_oracle.h). What it is for: the reference's arithmetic is Armadillo /
 * LAPACK in double, and nothing reference-held says what those libraries return to the last bit. Whatever they return,
 * it is a double-precision evaluation of the algorithm this file evaluates with ~3.3 more decimal digits; so the distance
 * between the double oracle and this one measures how far ANY careful double-precision evaluation -- the reference's
 * included -- can sit from the oracle on a given trace: about as far as the oracle sits from the (nearly) exact result.
 * tests/test_oracle_extended.py states those distances per trace.
 *
 * How: the keyword `double` is redefined to `long double` for the oracle's translation unit (after every standard header
 * has been read), so that not one line of the restatement is duplicated. Decimal literals stay what they are in the
 * double build (1e-3, 0.5, ...: the same thresholds); machine epsilon follows the type (the pseudo-inverse's rank cut is
 * max(m,n) * sigma_max * eps of the arithmetic in use, as Armadillo defines it). Only the wrappers below, which carry
 * IEEE doubles across the C boundary, are compiled with the keyword restored.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#define double long double
#include "kfpos_oracle.cpp"
#undef double

static_assert(sizeof(long double) > sizeof(double) && std::numeric_limits<long double>::digits >= 64,
              "this build needs an extended long double (x86-64)");

namespace {
std::vector<long double> widen(const double *p, size_t n) { return std::vector<long double>(p, p + n); }
} // namespace

extern "C" {

/* the slice of the oracle's C API that a trace replay needs, with IEEE doubles at the boundary */
void *kfx_create(int model, int n_tags, int n_anchors, const double *anchors_xyz, double accel_noise, double jolt,
                 int ignore_worst, double cost_threshold, int top_n, int use_init_pos, const double *init_pos) {
    std::vector<long double> ip;
    if (use_init_pos && init_pos) ip = widen(init_pos, (size_t)n_tags * 3);
    kfo_filter_bank *o = kfo_create(model, n_tags, n_anchors, accel_noise, jolt, ignore_worst, cost_threshold, top_n,
                                    use_init_pos, ip.empty() ? nullptr : ip.data());
    const std::vector<long double> a = widen(anchors_xyz, (size_t)n_anchors * 3);
    kfo_set_anchors(o, a.data(), n_anchors);
    return o;
}
void kfx_destroy(void *o) { kfo_destroy((kfo_filter_bank *)o); }
void kfx_step_toa(void *o, int n_tags, int n_anchors, const int32_t *range_mm, const double *err_est, double dt,
                  uint32_t *status, int n_threads) {
    const std::vector<long double> e = widen(err_est, (size_t)n_tags * n_anchors);
    const long double d = dt;
    kfo_step_toa((kfo_filter_bank *)o, range_mm, e.data(), &d, 1, status, n_threads);
}
void kfx_step_imu(void *o, int n_tags, const double *accel, const double *cov, double dt, uint32_t *status, int n_threads) {
    const std::vector<long double> a = widen(accel, (size_t)n_tags * 3), c = widen(cov, (size_t)n_tags * 9);
    const long double d = dt;
    kfo_step_imu((kfo_filter_bank *)o, a.data(), c.data(), &d, 1, status, n_threads);
}
/* x: n_tags x n, P: n_tags x n x n, rounded to double on the way out */
void kfx_get_state(void *o, int n_tags, double *x, double *P) {
    const int n = kfo_state_dim((kfo_filter_bank *)o);
    std::vector<long double> xx((size_t)n_tags * n), pp((size_t)n_tags * n * n);
    kfo_get_state((kfo_filter_bank *)o, xx.data(), pp.data());
    for (size_t i = 0; i < xx.size(); ++i) x[i] = (double)xx[i];
    for (size_t i = 0; i < pp.size(); ++i) P[i] = (double)pp[i];
}
int kfx_mantissa_bits(void) { return std::numeric_limits<long double>::digits; }

} // extern "C"

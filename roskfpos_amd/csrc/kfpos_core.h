/*
 * kfpos_core.h -- per-tag arithmetic of the batched EKF core (one filter per lane).
 *
 * This is the body every HIP kernel in kfpos_k_*.hip runs for one tag. It is written against
 * plain doubles and compile-time-indexed arrays only (everything unrolls into registers; the
 * only device builtins are the v_rcp_f64 / v_rsq_f64 seeds in kf_rcp / kf_rsqrt), so the same
 * text also compiles with g++ into the host emulation used by the CPU tests (tests/emu) to
 * check the algebra without a GPU.
 *
 * What it computes is the reference's iterated EKF, restructured
 * for a register-resident, branch-light evaluation -- results equal the reference's to
 * rounding (parity tests: <= 1e-6 m RMS required, ~1e-12 m observed):
 *
 *  - H has non-zero columns only for position (and, with the IMU rows, acceleration), and R is
 *    diagonal (+ one 3x3 block), so with B = H restricted to those columns E and M = B' R^-1 B
 *        H' S^-1 v = E' (I + M P_ee)^-1 B' R^-1 v          (Woodbury; S = H P H' + R)
 *    replaces the reference's m x m inverse (KalmanFilterTOA.cpp:316-317) by a 3x3 (6-state) or
 *    6x6 (9-state, solved as 3x3 blocks + Schur complement) system that does not grow with the
 *    anchor count.
 *  - the state after a gain iteration is x = xhat + P E' w, hence delta = xhat - x = -P E' w and
 *        delta' pinv(P) delta = w' P_ee w = -w . delta_e
 *    for symmetric PSD P of any rank (and for any invertible P), which removes the reference's
 *    pinv(P) (KalmanFilterTOA.cpp:290): it only ever feeds the convergence cost.
 *  - P+ = (I - K H) P = P - P E' N E P with N = (I + M P_ee)^-1 M (KalmanFilterTOA.cpp:326).
 *  - F and Q are closed-form in dt (KalmanFilterTOA.cpp:362-391, KalmanFilterTOAIMU.cpp:392-421),
 *    so x <- F x, P <- F P F' + Q is a structured update of the packed covariance, not a GEMM.
 *
 * Reference quirks kept on purpose: velocity (6-state) / acceleration (9-state)
 * restart at 0 every step; the ML loop's first comparison is against the constant 1; the ML-init
 * covariance copy repeats column 1 (6-state, which makes P non-symmetric: COV_FULL layout);
 * H_imu = diag(a) instead of I; R = max(e_ML, errEst).
 */
#ifndef KFPOS_CORE_H
#define KFPOS_CORE_H

#include <math.h>
#include <stdint.h>

#ifndef KFPOS_HD
#define KFPOS_HD
#endif
#define KFPOS_FN KFPOS_HD inline __attribute__((always_inline))
#if defined(__clang__)
#define KFPOS_UNROLL _Pragma("unroll")
#define KFPOS_NOUNROLL _Pragma("nounroll")
#else
#define KFPOS_UNROLL
#define KFPOS_NOUNROLL
#endif

/* true when the predicate holds on every active lane of the wavefront (a wave-uniform value, so a branch on it
 * does not diverge); the host build runs one tag at a time */
#if defined(__HIP_DEVICE_COMPILE__)
#define KFPOS_WAVE_ALL(pred) (__all((pred) ? 1 : 0) != 0)
#else
#define KFPOS_WAVE_ALL(pred) (pred)
#endif

namespace kfpos {

/* per-tag status word (include/kfpos.h repeats these as KFPOS_ST_*) */
enum : uint32_t {
    ST_UPDATE_SKIPPED = 0x01u, /* the reference's swallowed std::runtime_error (KalmanFilterTOA.cpp:151-153) */
    ST_ML_FALLBACK    = 0x02u, /* ML position NaN -> predicted position (KalmanFilterTOA.cpp:270-272) */
    ST_FEW_RANGES     = 0x04u, /* < 4 ranges: ML returns its seed (MLLocation.cpp:158-161) */
    ST_ML_INIT        = 0x08u, /* this call was the ML initialisation (KalmanFilterTOA.cpp:90-108) */
    ST_NOT_STARTED    = 0x10u, /* getPose before any measurement (KalmanFilterTOA.cpp:442-447) */
    ST_NONFINITE      = 0x20u, /* state not finite after the call */
    ST_SKIPPED        = 0x40u, /* dt < 0: no epoch for this tag in this call, filter untouched */
};
/* persisted per-tag flag bits */
enum : uint32_t { FL_STARTED = 1u, FL_HAS_IMU = 2u };

KFPOS_FN uint32_t pack_status(uint32_t flags, int gain_iters, int ml_iters, int ignored) {
    const uint32_t g = gain_iters > 255 ? 255u : (uint32_t)gain_iters;
    const uint32_t m = ml_iters > 255 ? 255u : (uint32_t)ml_iters;
    return flags | (g << 8) | (m << 16) | ((uint32_t)(ignored + 1) << 24);
}

/* Reciprocal, reciprocal square root and square root to ~1 ulp: hardware seed + two Newton steps
 * (8-10 fp64 instructions instead of the ~20-30 of an IEEE-exact divide / sqrt sequence; the parity
 * bar is 1e-6 m, not the last bit -- the reference's own LAPACK arithmetic is not reproducible to
 * the last bit either). The host emulation uses the plain operators. */
KFPOS_FN double kf_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
#else
    return 1.0 / x;
#endif
}
/* fused multiply-add where the rounding has to be the same at two places of the device code; the host emulation, which
 * is compared with the oracle to a tolerance, not with the GPU bit for bit, multiplies and adds */
KFPOS_FN double kf_fma(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fma(a, b, c);
#else
    return a * b + c;
#endif
}
/* 0 that the compiler cannot see through: added to an index, it keeps a loop-invariant read of parked values inside
 * the loop (hoisted, they would occupy registers for the whole loop) without hiding which memory the pointer is in */
KFPOS_FN int kf_opaque_zero() {
    int z = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(z));
#endif
    return z;
}
/* index of the lowest set bit of a non-zero mask */
KFPOS_FN int kf_ctz64(uint64_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)m) - 1;
#else
    return __builtin_ctzll(m);
#endif
}
KFPOS_FN double kf_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    double e = __builtin_fma(-h * y, y, 0.5);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-h * y, y, 0.5);
    return __builtin_fma(y, e, y);
#else
    return 1.0 / sqrt(x);
#endif
}
/* d = sqrt(x) and invd = 1/sqrt(x) together (x > 0) */
KFPOS_FN void kf_sqrt_rsqrt(double x, double &d, double &invd) {
#if defined(__HIP_DEVICE_COMPILE__)
    invd = kf_rsqrt(x);
    const double s = x * invd;
    d = __builtin_fma(__builtin_fma(-s, s, x) * 0.5, invd, s);
#else
    d = sqrt(x);
    invd = 1.0 / d;
#endif
}

/* Variant for the anchor sweeps: d to full precision, invd after ONE Newton step (relative error
 * ~2e-14: the hardware seed carries ~23 bits). The residual r - d uses d; invd only scales the
 * Jacobian row (p - b) / d, where 2e-14 moves the result by ~1e-16 m. Saves 3 of 12 instructions. */
KFPOS_FN void kf_sqrt_rsqrt_sweep(double x, double &d, double &invd) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(0.5 * x) * y, y, 0.5);
    y = __builtin_fma(y, e, y);
    const double s = x * y;
    d = __builtin_fma(__builtin_fma(-s, s, x) * 0.5, y, s);
    invd = y;
#else
    d = sqrt(x);
    invd = 1.0 / d;
#endif
}

/* Integer millimetres -> metres with the value of the reference's `(double) mm / 1000`
 * (Posgenerator.cpp:484), bit for bit: q = mm * RN(1/1000) is within one ulp, r = mm - 1000 q is exact
 * in an fma, and q + r * RN(1/1000) rounds to the correctly rounded quotient (checked exhaustively for
 * every mm < 2^26 and sampled to 2^31 against the division). 3 instructions instead of ~20. */
KFPOS_FN double kf_mm_to_m(int32_t mm) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double x = (double)mm;
    const double q = x * 0.001;
    const double r = __builtin_fma(-q, 1000.0, x);
    return __builtin_fma(r, 0.001, q);
#else
    return (double)mm / 1000;
#endif
}

/* The reference's convergence test  fabs(cost - c) / cost < tol  (KalmanFilterTOA.cpp:307, KalmanFilterTOAIMU.cpp:316)
 * with the same decision for every input, without paying an IEEE division (a dozen instructions) in every iteration:
 * the product form decides unless the two sides are within a few ulps of each other, and only then -- wave-uniformly,
 * practically never -- is the quotient formed. NaN / 0 / inf operands fall through to the exact form as well. */
KFPOS_FN bool rel_change_below(double cost, double c, double tol) {
    const double lhs = fabs(cost - c), rhs = tol * cost;
    const bool clear = cost > 0.0 && fabs(lhs - rhs) > 1e-13 * rhs; /* false for NaN and for cost <= 0 or inf */
    if (KFPOS_WAVE_ALL(clear)) return lhs < rhs;
    return lhs / cost < tol;
}

/* The Gauss-Newton loop's test  fabs(cost - newCost) / cost > tol  (MLLocation.cpp:168; :79 for the 2-D solver), the same
 * way: every solve of every filter runs it once per pass. */
KFPOS_FN bool rel_change_above(double cost, double c, double tol) {
    const double lhs = fabs(cost - c), rhs = tol * cost;
    const bool clear = cost > 0.0 && fabs(lhs - rhs) > 1e-13 * rhs; /* false for NaN and for cost <= 0 or inf */
    if (KFPOS_WAVE_ALL(clear)) return lhs > rhs;
    return lhs / cost > tol;
}

/* std::max as the reference uses it: (a < b) ? b : a (matters for NaN) */
KFPOS_FN double stdmax(double a, double b) { return (a < b) ? b : a; }

/* ------------------------------------------------------------------ covariance storage */
/* Packed upper triangle (symmetric) or full row-major. All indices are compile-time after
 * unrolling, so the array lives in registers. */
template <int N, bool SYMM>
struct Cov {
    static constexpr int SZ = SYMM ? N * (N + 1) / 2 : N * N;
    double a[SZ];
    KFPOS_HD static constexpr int idx(int i, int j) {
        return SYMM ? (i <= j ? i * N - i * (i - 1) / 2 + (j - i) : j * N - j * (j - 1) / 2 + (i - j))
                    : i * N + j;
    }
    KFPOS_HD double operator()(int i, int j) const { return a[idx(i, j)]; }
    KFPOS_HD double &operator()(int i, int j) { return a[idx(i, j)]; }
};

/* Uniform (per-launch) parameters. anchors: xyz triples, wave-uniform reads. */
struct Params {
    const double *anchors;
    int n_anchors;
    double accel_noise, jolt, cost_threshold;
    int ignore_worst, top_n, use_init_pos;
    int ml_variant = 0; /* standalone ML estimator: 0 NORMAL / 1 IGNORE_N (acts through top_n) / 2 BEST (MLLocation.h:5-7) */
    /* 8-state planar filter only: what KalmanFilter::loadConfigurationFiles reads (KalmanFilter.cpp:748-842) */
    int use_fixed_height, imu_fixed_cov_acc, imu_fixed_cov_w;
    double px4_height, px4_arm_p1, px4_arm_p2, px4_cov_vel, px4_cov_gyro_z;
    double imu_cov_acc, imu_cov_w, mag_offset, mag_cov;
    const double *pair_anchor_tab = nullptr; /* 9-state kernel: [8][3] copy of the anchor table that lanes can index
                                                individually (LDS); non-null = the tail of the gain iteration runs two
                                                lanes per tag (iekf9_pairs) */
};

/* ------------------------------------------------------------------ 3x3 helpers */
/* inverse of a symmetric 3x3 {00,01,02,11,12,22} by cofactors; returns det */
KFPOS_FN double sym3_cofactors(const double h[6], double c[6]) {
    c[0] = h[3] * h[5] - h[4] * h[4];
    c[1] = h[2] * h[4] - h[1] * h[5];
    c[2] = h[1] * h[4] - h[2] * h[3];
    c[3] = h[0] * h[5] - h[2] * h[2];
    c[4] = h[1] * h[2] - h[0] * h[4];
    c[5] = h[0] * h[3] - h[1] * h[1];
    return h[0] * c[0] + h[1] * c[1] + h[2] * c[2];
}
/* adjugate of a general 3x3 (row-major), returns det; inverse = adj / det */
KFPOS_FN double gen3_adjugate(const double m[9], double adj[9]) {
    adj[0] = m[4] * m[8] - m[5] * m[7];
    adj[1] = m[2] * m[7] - m[1] * m[8];
    adj[2] = m[1] * m[5] - m[2] * m[4];
    adj[3] = m[5] * m[6] - m[3] * m[8];
    adj[4] = m[0] * m[8] - m[2] * m[6];
    adj[5] = m[2] * m[3] - m[0] * m[5];
    adj[6] = m[3] * m[7] - m[4] * m[6];
    adj[7] = m[1] * m[6] - m[0] * m[7];
    adj[8] = m[0] * m[4] - m[1] * m[3];
    return m[0] * adj[0] + m[1] * adj[3] + m[2] * adj[6];
}

/* X = A^-1 B for a general N x N A (row-major) and K right-hand sides (B, X: N x K row-major) by Gaussian
 * elimination with partial pivoting, branch-free (rows are exchanged through selects, every index is static). The
 * adjugates above lose digits when the eigenvalues of I + M P spread over many decades (determinant by
 * cancellation): a prior covariance of 1e4 m^2 against range variances of 1e-3 m^2 -- an ML initialisation from
 * coplanar anchors -- costs them 11 of 16. Used only for such steps (illconditioned()). */
template <int N, int K>
KFPOS_FN void gauss_solve(const double *a, const double *b, double *x) {
    double r[N][N + K];
    KFPOS_UNROLL
    for (int i = 0; i < N; ++i) {
        KFPOS_UNROLL
        for (int j = 0; j < N; ++j) r[i][j] = a[N * i + j];
        KFPOS_UNROLL
        for (int k = 0; k < K; ++k) r[i][N + k] = b[K * i + k];
    }
    KFPOS_UNROLL
    for (int c = 0; c < N; ++c) {
        KFPOS_UNROLL
        for (int i = c + 1; i < N; ++i) { /* largest |entry| of column c to row c */
            const bool sw = fabs(r[i][c]) > fabs(r[c][c]);
            KFPOS_UNROLL
            for (int k = c; k < N + K; ++k) {
                const double tc = r[c][k], ti = r[i][k];
                r[c][k] = sw ? ti : tc;
                r[i][k] = sw ? tc : ti;
            }
        }
        const double ip = 1.0 / r[c][c];
        KFPOS_UNROLL
        for (int i = c + 1; i < N; ++i) {
            const double f = r[i][c] * ip;
            KFPOS_UNROLL
            for (int k = c + 1; k < N + K; ++k) r[i][k] -= f * r[c][k];
        }
    }
    KFPOS_UNROLL
    for (int k = 0; k < K; ++k) {
        double sol[N];
        KFPOS_UNROLL
        for (int i = N - 1; i >= 0; --i) {
            double v = r[i][N + k];
            KFPOS_UNROLL
            for (int j = i + 1; j < N; ++j) v -= r[i][j] * sol[j];
            sol[i] = v / r[i][i];
        }
        KFPOS_UNROLL
        for (int i = 0; i < N; ++i) x[K * i + k] = sol[i];
    }
}

/* Lower Cholesky of a symmetric PSD 3x3 {00,01,02,11,12,22}; a pivot that cancels to
 * rounding level marks a rank-deficient direction (fewer than 3 independent ranges): its
 * column is zeroed, which is the semidefinite factor (M = L L' still holds).
 * l = {l00,l10,l20,l11,l21,l22}, il = reciprocals of the diagonal (0 for a dropped column). */
KFPOS_FN void chol3_psd(const double m[6], double l[6], double il[3]) {
    const double REL = 1e-12;
    double d = m[0], sq, isq;
    bool ok = d > 0.0;
    kf_sqrt_rsqrt(ok ? d : 1.0, sq, isq);
    l[0] = ok ? sq : 0.0;
    il[0] = ok ? isq : 0.0;
    l[1] = m[1] * il[0];
    l[2] = m[2] * il[0];
    d = m[3] - l[1] * l[1];
    ok = d > REL * m[3];
    kf_sqrt_rsqrt(ok ? d : 1.0, sq, isq);
    l[3] = ok ? sq : 0.0;
    il[1] = ok ? isq : 0.0;
    l[4] = (m[4] - l[2] * l[1]) * il[1];
    d = m[5] - l[2] * l[2] - l[4] * l[4];
    ok = d > REL * m[5];
    kf_sqrt_rsqrt(ok ? d : 1.0, sq, isq);
    l[5] = ok ? sq : 0.0;
    il[2] = ok ? isq : 0.0;
}

/* ------------------------------------------------------------------ measurement scratch */
/* Per-lane view of one tag's epoch: r = range in metres (<= 0: absent), e = errorEstimation,
 * w = working weight (1/e during ML, 1/R during the IEKF). Element a lives at base[a*stride]:
 * stride = wavefront width in LDS (conflict-free), 1 in the host emulation. */
struct Scratch {
    static constexpr int NA = 0; /* anchor count only known at run time */
    static constexpr int CHUNK = 0;
    static constexpr bool COOP = false;
    double *r, *e, *w;
    int stride;
    KFPOS_HD double R(int a) const { return r[a * stride]; }
    KFPOS_HD double E(int a) const { return e[a * stride]; }
    KFPOS_HD double W(int a) const { return w[a * stride]; }
    KFPOS_HD void setW(int a, double v) { w[a * stride] = v; }
    KFPOS_HD double Rdyn(int a) const { return r[a * stride]; } /* a not a compile-time constant */
    KFPOS_HD double Edyn(int a) const { return e[a * stride]; }
    KFPOS_HD void setWdyn(int a, double v) { w[a * stride] = v; }
};
/* Scratch with the anchor count fixed at compile time but the epoch still outside the register file: the anchor
 * loops unroll (coordinates become batched constant-offset scalar loads, the LDS reads of a sweep are issued
 * back to back) without the 6 registers per anchor that RegScratch costs -- for counts too large to keep in
 * registers next to the covariance (16 anchors: BASELINE config 5). */
template <int N>
struct StaticScratch : Scratch {
    static constexpr int NA = N;
    /* anchors per unrolled group: unrolling all 16 at once lets the scheduler interleave 16 rsqrt chains and spills
     * (156-452 bytes/lane measured); groups of 8 keep the live set of the 8-anchor kernels */
    static constexpr int CHUNK = N > 8 ? 8 : 0;
};
/* StaticScratch whose errorEstimations sit in LDS as the 4-byte values they arrived as (KFPOS_STORE_F32 / _MIXED): 20
 * instead of 24 bytes per anchor and lane. At 16 anchors that is 20 KB per wavefront, so EIGHT workgroups fit the 160 KB
 * of a CU instead of six, and BASELINE config 5 (4 096 wavefronts) runs in two full rounds of two wavefronts per SIMD
 * instead of 2.67 ragged ones. */
template <int N>
struct StaticScratchF {
    static constexpr int NA = N;
    static constexpr int CHUNK = N > 8 ? 8 : 0;
    static constexpr bool COOP = false;
    double *r, *w;
    float *e;
    int stride;
    KFPOS_HD double R(int a) const { return r[a * stride]; }
    KFPOS_HD double E(int a) const { return (double)e[a * stride]; }
    KFPOS_HD double W(int a) const { return w[a * stride]; }
    KFPOS_HD void setW(int a, double v) { w[a * stride] = v; }
    KFPOS_HD double Rdyn(int a) const { return r[a * stride]; }
    KFPOS_HD double Edyn(int a) const { return (double)e[a * stride]; }
    KFPOS_HD void setWdyn(int a, double v) { w[a * stride] = v; }
};
/* Same view with the anchor count fixed at compile time: the epoch stays in registers, every anchor
 * loop unrolls, anchor coordinates become constant-offset scalar loads that the compiler batches. */
template <int N>
struct RegScratch {
    static constexpr int NA = N;
    static constexpr int CHUNK = 0; /* register arrays need compile-time indices: one fully unrolled group */
    static constexpr bool COOP = false;
    double r[N], e[N], w[N];
    KFPOS_HD double R(int a) const { return r[a]; }
    KFPOS_HD double E(int a) const { return e[a]; }
    KFPOS_HD double W(int a) const { return w[a]; }
    KFPOS_HD void setW(int a, double v) { w[a] = v; }
    KFPOS_HD double Rdyn(int a) const { /* run-time index without sending the array to scratch memory */
        double v = 0.0;
        KFPOS_UNROLL
        for (int k = 0; k < N; ++k) v = (k == a) ? r[k] : v;
        return v;
    }
    KFPOS_HD double Edyn(int a) const {
        double v = 1.0;
        KFPOS_UNROLL
        for (int k = 0; k < N; ++k) v = (k == a) ? e[k] : v;
        return v;
    }
    KFPOS_HD void setWdyn(int a, double v) {
        KFPOS_UNROLL
        for (int k = 0; k < N; ++k) w[k] = (k == a) ? v : w[k];
    }
};
/* One tag per GROUP OF 8 LANES, one anchor per lane. For small batches (a few thousand tags) the machine is mostly
 * empty and what bounds a step is the instruction chain of a single lane; here the anchor sweeps of a tag -- most of
 * that chain -- are spread over 8 lanes and their partial sums combined with three DPP exchanges, while the small
 * solves run redundantly (bit-identically) on all 8. Anchor "loops" run once, for the lane's own anchor (index 0). */
struct CoopScratch {
    static constexpr int NA = 1;
    static constexpr int CHUNK = 0;
    static constexpr bool COOP = true;
    double r, e, w;    /* this lane's range (m, <= 0: absent / no such anchor), errorEstimation, working weight */
    double bx, by, bz; /* this lane's anchor */
    KFPOS_HD double R(int) const { return r; }
    KFPOS_HD double E(int) const { return e; }
    KFPOS_HD double W(int) const { return w; }
    KFPOS_HD void setW(int, double v) { w = v; }
    KFPOS_HD double Rdyn(int) const { return 0.0; } /* leave-one-out is not offered in this mode */
    KFPOS_HD double Edyn(int) const { return 1.0; }
    KFPOS_HD void setWdyn(int, double) {}
};

/* coordinates of anchor column a */
template <class SC>
KFPOS_FN void anchor_of(const SC &sc, const Params &pr, int a, double &bx, double &by, double &bz) {
    if constexpr (SC::COOP) {
        bx = sc.bx; by = sc.by; bz = sc.bz;
    } else {
        bx = pr.anchors[3 * a]; by = pr.anchors[3 * a + 1]; bz = pr.anchors[3 * a + 2];
    }
}

/* Sum of v over the 8 lanes of a group, delivered to all of them with identical bits (a butterfly: every lane adds
 * the same two partial sums at every stage). Identity for the one-tag-per-lane layouts. */
#if defined(__HIP_DEVICE_COMPILE__)
KFPOS_FN double dpp_exchange(double v, int ctrl_tag) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int plo, phi;
    if (ctrl_tag == 0) { /* quad_perm [1,0,3,2]: lane ^ 1 */
        plo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
        phi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
    } else if (ctrl_tag == 1) { /* quad_perm [2,3,0,1]: lane ^ 2 */
        plo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true);
        phi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);
    } else { /* row_half_mirror: lane i <-> 7 - i within each group of 8 */
        plo = __builtin_amdgcn_mov_dpp(lo, 0x141, 0xF, 0xF, true);
        phi = __builtin_amdgcn_mov_dpp(hi, 0x141, 0xF, 0xF, true);
    }
    return __hiloint2double(phi, plo);
}
#endif
template <class SC>
KFPOS_FN double group_sum(const SC &, double v) {
    if constexpr (SC::COOP) {
#if defined(__HIP_DEVICE_COMPILE__)
        v += dpp_exchange(v, 0);
        v += dpp_exchange(v, 1);
        v += dpp_exchange(v, 2);
#endif
    }
    return v;
}

/* f(a) for every anchor column; fully unrolled when the count is static */
template <class SC, class F>
KFPOS_FN void for_anchors(const Params &pr, F &&f) {
    if constexpr (SC::NA > 0 && SC::CHUNK > 0) {
        static_assert(SC::NA % SC::CHUNK == 0, "anchor count must be a multiple of the unroll group");
        KFPOS_NOUNROLL
        for (int c = 0; c < SC::NA; c += SC::CHUNK) {
            KFPOS_UNROLL
            for (int k = 0; k < SC::CHUNK; ++k) f(c + k);
        }
    } else if constexpr (SC::NA > 0) {
        KFPOS_UNROLL
        for (int a = 0; a < SC::NA; ++a) f(a);
    } else {
        for (int a = 0; a < pr.n_anchors; ++a) f(a);
    }
}
template <class SC>
KFPOS_FN bool used(const SC &sc, int a, uint64_t drop) {
    return !((drop >> a) & 1ull) && sc.R(a) > 0.0;
}
template <class SC>
KFPOS_FN int count_used(const SC &sc, const Params &pr, uint64_t drop) {
    int n = 0;
    for_anchors<SC>(pr, [&](int a) { n += used(sc, a, drop) ? 1 : 0; });
    if constexpr (SC::COOP) n = (int)group_sum(sc, (double)n);
    return n;
}

} // namespace kfpos

/* the estimators, in dependency order */
#include "kfpos_core_ml.h"
#include "kfpos_core_toa6.h"
#include "kfpos_core_imu9.h"
#include "kfpos_core_planar.h"
#endif

/*
 * kfpos_core.h -- per-tag arithmetic of the batched EKF core (one filter per lane).
 *
 * This is the body every HIP kernel in kfpos_hip.hip runs for one tag. It is written against
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
    /* 8-state planar filter only: what KalmanFilter::loadConfigurationFiles reads (KalmanFilter.cpp:748-842) */
    int use_fixed_height, imu_fixed_cov_acc, imu_fixed_cov_w;
    double px4_height, px4_arm_p1, px4_arm_p2, px4_cov_vel, px4_cov_gyro_z;
    double imu_cov_acc, imu_cov_w, mag_offset, mag_cov;
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

/* ------------------------------------------------------------------ MLLocation::estimatePosition */
/* One sweep over the anchors at position p: weighted cost sum (r-d)^2/e, unweighted SSE
 * (estimationError, MLLocation.cpp:263-278), gradient g and Hessian-like Hs of
 * MLLocation.cpp:174-204 (Hs symmetric, packed {00,01,02,11,12,22}). */
template <class SC>
KFPOS_FN void ml_sweep(const double p[3], const SC &sc, const Params &pr, uint64_t drop,
                              double &cw, double &sse, double g[3], double hs[6]) {
    double cw_ = 0.0, sse_ = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
    double h0 = 0.0, h1 = 0.0, h2 = 0.0, h3 = 0.0, h4 = 0.0, h5 = 0.0;
    for_anchors<SC>(pr, [&](int a) {
        /* branch-free: an absent / dropped range gets weight 0 (select, so a garbage errorEstimation of a
         * missing range never enters), which keeps the unrolled anchors in one basic block and lets the
         * scheduler interleave their independent rsqrt chains */
        const bool on = used(sc, a, drop);
        const double r = sc.R(a), w = sc.W(a); /* 0 for an absent / dropped range (set_weights_*) */
        double bx, by, bz;
        anchor_of(sc, pr, a, bx, by, bz);
        const double dx = bx - p[0], dy = by - p[1], dz = bz - p[2];
        double d, invd;
        kf_sqrt_rsqrt_sweep(dx * dx + dy * dy + dz * dz, d, invd);
        const double rd = r - d;
        cw_ += rd * rd * w;
        sse_ += on ? rd * rd : 0.0;
        const double gi = rd * invd * w;
        g0 += gi * dx;
        g1 += gi * dy;
        g2 += gi * dz;
        const double q = r * invd;
        const double c0 = w * (1.0 - q), c1 = w * q * invd * invd;
        h0 += c0 + c1 * dx * dx;
        h1 += c1 * dx * dy;
        h2 += c1 * dx * dz;
        h3 += c0 + c1 * dy * dy;
        h4 += c1 * dy * dz;
        h5 += c0 + c1 * dz * dz;
    });
    cw = group_sum(sc, cw_); sse = group_sum(sc, sse_);
    g[0] = group_sum(sc, g0); g[1] = group_sum(sc, g1); g[2] = group_sum(sc, g2);
    hs[0] = group_sum(sc, h0); hs[1] = group_sum(sc, h1); hs[2] = group_sum(sc, h2);
    hs[3] = group_sum(sc, h3); hs[4] = group_sum(sc, h4); hs[5] = group_sum(sc, h5);
}

/* SSE only (estimationError at a given position) */
template <class SC>
KFPOS_FN double ml_sse(const double p[3], const SC &sc, const Params &pr, uint64_t drop) {
    double sse = 0.0;
    for_anchors<SC>(pr, [&](int a) {
        double bx, by, bz;
        anchor_of(sc, pr, a, bx, by, bz);
        const double dx = bx - p[0], dy = by - p[1], dz = bz - p[2];
        double d, invd;
        kf_sqrt_rsqrt(dx * dx + dy * dy + dz * dz, d, invd);
        const double rd = d - sc.R(a);
        sse += used(sc, a, drop) ? rd * rd : 0.0;
    });
    return group_sum(sc, sse);
}

/* Gauss-Newton loop of MLLocation.cpp:164-225. p: seed in, estimate out. Requires sc.w = 1/e.
 * Returns the iteration count; sse_out = estimationError at the result. With n_used < 4 the
 * seed is returned untouched (MLLocation.cpp:158-161). The step p - Hs^-1 g equals the
 * reference's solve(Hs, Hs p - g). One sweep per pass yields the cost of the point just reached
 * and the gradient/Hessian for the next step (the reference evaluates them in two passes). */
template <class SC>
KFPOS_FN int ml_estimate(double p[3], const SC &sc, const Params &pr, uint64_t drop,
                                int n_used, double &sse_out) {
    if (n_used < 4) {
        sse_out = (n_used == 0) ? -1.0 : ml_sse(p, sc, pr, drop);
        return 0;
    }
    double cost = 1e20, newCost = 1.0, cw, sse, g[3], hs[6], c[6];
    int iter = 0;
    for (;;) {
        ml_sweep(p, sc, pr, drop, cw, sse, g, hs);
        if (iter > 0) newCost = cw;
        if (!((fabs(cost - newCost) / cost > 1e-3) && (iter < 10000))) break; /* MLLocation.cpp:168 */
        iter += 1;
        cost = newCost;
        const double idet = kf_rcp(sym3_cofactors(hs, c));
        p[0] -= (c[0] * g[0] + c[1] * g[1] + c[2] * g[2]) * idet;
        p[1] -= (c[1] * g[0] + c[3] * g[1] + c[4] * g[2]) * idet;
        p[2] -= (c[2] * g[0] + c[4] * g[1] + c[5] * g[2]) * idet;
    }
    sse_out = sse;
    return iter;
}

/* covariance of the ML estimate, inv(J' diag(max(e, e_ML))^-1 J) (MLLocation.cpp:229-252);
 * symmetric 3x3 packed. Only the ML initialisation uses it. */
template <class SC>
KFPOS_FN bool ml_covariance(const double p[3], const SC &sc, const Params &pr, double sse,
                                   double cov[6]) {
    double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, c[6];
    for_anchors<SC>(pr, [&](int a) {
        if (!used(sc, a, 0)) return;
        double bx, by, bz;
        anchor_of(sc, pr, a, bx, by, bz);
        const double dx = p[0] - bx, dy = p[1] - by, dz = p[2] - bz;
        const double invd = 1.0 / sqrt(dx * dx + dy * dy + dz * dz);
        const double w = 1.0 / stdmax(sc.E(a), sse);
        const double gx = dx * invd, gy = dy * invd, gz = dz * invd;
        m0 += w * gx * gx; m1 += w * gx * gy; m2 += w * gx * gz;
        m3 += w * gy * gy; m4 += w * gy * gz; m5 += w * gz * gz;
    });
    const double m[6] = {group_sum(sc, m0), group_sum(sc, m1), group_sum(sc, m2),
                         group_sum(sc, m3), group_sum(sc, m4), group_sum(sc, m5)};
    const double det = sym3_cofactors(m, c);
    const double idet = 1.0 / det;
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) cov[k] = c[k] * idet;
    return det != 0.0; /* inv() of an exactly singular J' W J throws (anchors and seed in one plane / on one line) */
}

/* MLLocation::estimatePosition ends with inv(diagmat(max(e_i, e_ML))) and inv(J' W J) (MLLocation.cpp:248-252).
 * The first one throws std::runtime_error when an entry is exactly 0 -- which is what an errorEstimation of 0
 * leads to: the 1/e weights make the ML position NaN, e_ML is NaN, std::max(e_i, NaN) = e_i = 0. The
 * 6-state filter swallows the exception and skips the update (KalmanFilterTOA.cpp:151-153). */
template <class SC>
KFPOS_FN bool ml_covariance_throws(const SC &sc, const Params &pr, uint64_t drop, int n_used, double sse,
                                   int min_used = 4) {
    if (n_used < min_used) return false; /* estimatePosition returned before getting there */
    bool bad = false;
    for_anchors<SC>(pr, [&](int a) { bad = bad || (used(sc, a, drop) && stdmax(sc.E(a), sse) == 0.0); });
    if constexpr (SC::COOP) bad = group_sum(sc, bad ? 1.0 : 0.0) > 0.0;
    return bad;
}

/* Not detected in the per-epoch solves: the second inverse, inv(J' W J), also throws when J' W J is EXACTLY singular
 * (all used anchors and the estimate on one line / in one plane with coordinates symmetric enough that every
 * cancellation is exact -- anchors at (0,0) and (10,10) with the estimate on the diagonal). The update path does not
 * form that covariance (an extra sweep per solve, 3-14 % of a step, for a measure-zero geometry); the ML
 * initialisation, which needs the covariance anyway, does report it (ml_covariance / ml2d_covariance). */
/* Working weights. An absent or dropped range gets weight 0 HERE (a select, so a garbage errorEstimation of a
 * missing range never enters): the sweeps, which run 10-60 times per step, then read the weight as it is. */
template <class SC>
KFPOS_FN void set_weights_ml(SC &sc, const Params &pr, uint64_t drop) {
    for_anchors<SC>(pr, [&](int a) { sc.setW(a, used(sc, a, drop) ? kf_rcp(sc.E(a)) : 0.0); });
}
template <class SC>
KFPOS_FN void set_weights_iekf(SC &sc, const Params &pr, double e_ml, uint64_t drop) {
    for_anchors<SC>(pr, [&](int a) { /* KalmanFilterTOA.cpp:281 */
        sc.setW(a, used(sc, a, drop) ? kf_rcp(stdmax(e_ml, sc.E(a))) : 0.0);
    });
}

/* Top-N composition (BASELINE config 5; MLLocation.cpp:284-300, 325-339): rank the residual^2
 * at the ML position of all ranges, drop the min(n-4, N) largest. Returns the drop mask. */
template <class SC>
KFPOS_FN uint64_t topn_mask(const double seed[3], SC &sc, const Params &pr, int n_valid) {
    int ndrop = n_valid - 4 < pr.top_n ? n_valid - 4 : pr.top_n;
    if (ndrop <= 0) return 0;
    double p[3] = {seed[0], seed[1], seed[2]}, sse;
    set_weights_ml(sc, pr, 0ull);
    ml_estimate(p, sc, pr, 0, n_valid, sse);
    uint64_t drop = 0;
    for (int k = 0; k < ndrop; ++k) {
        double worst = -1.0;
        int wi = -1;
        for_anchors<SC>(pr, [&](int a) {
            const double dx = pr.anchors[3 * a] - p[0], dy = pr.anchors[3 * a + 1] - p[1],
                         dz = pr.anchors[3 * a + 2] - p[2];
            double d, invd;
            kf_sqrt_rsqrt(dx * dx + dy * dy + dz * dz, d, invd);
            const double rd = d - sc.R(a);
            const bool take = used(sc, a, drop) && (rd * rd > worst);
            worst = take ? rd * rd : worst;
            wi = take ? a : wi;
        });
        if (wi < 0) break;
        drop |= 1ull << wi;
    }
    return drop;
}

/* ================================================================== standalone ML estimator (ALGORITHM_ML) */
/* MLLocation::newTOAMeasurement + getPose (MLLocation.cpp:421-486), variant NORMAL 3-D or IGNORE_N
 * (estimatePositionIgnoreN, :307-347): solve from the fixed seed (_previousEstimation is never updated),
 * optionally drop the min(n-4, N) largest residuals and solve again, return position + 3x3 covariance. */
template <class SC>
KFPOS_FN uint32_t step_ml(double pos[3], double cov[6], SC &sc, const Params &pr, const double seed[3]) {
    int n_valid = count_used(sc, pr, 0);
    if (n_valid < 4) { /* estimatePosition returns the seed; its covariance is empty (getPose would abort) */
        KFPOS_UNROLL
        for (int k = 0; k < 3; ++k) pos[k] = seed[k];
        KFPOS_UNROLL
        for (int k = 0; k < 6; ++k) cov[k] = NAN;
        return ST_FEW_RANGES;
    }
    uint64_t drop = 0;
    if (pr.top_n > 0) {
        drop = topn_mask(seed, sc, pr, n_valid); /* first solve + ranking */
        /* the first solve throws exactly when a used errorEstimation is 0 (its sse is NaN then) */
        if (ml_covariance_throws(sc, pr, 0, n_valid, NAN)) return ST_UPDATE_SKIPPED;
        n_valid = count_used(sc, pr, drop);
    }
    double p[3] = {seed[0], seed[1], seed[2]}, sse;
    set_weights_ml(sc, pr, drop);
    const int it = ml_estimate(p, sc, pr, drop, n_valid, sse);
    if (ml_covariance_throws(sc, pr, drop, n_valid, sse)) return ST_UPDATE_SKIPPED;
    double c[6];
    /* covariance over the kept ranges only */
    {
        double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, cf[6];
        for_anchors<SC>(pr, [&](int a) {
            const bool on = used(sc, a, drop);
            const double dx = p[0] - pr.anchors[3 * a], dy = p[1] - pr.anchors[3 * a + 1],
                         dz = p[2] - pr.anchors[3 * a + 2];
            const double invd = 1.0 / sqrt(dx * dx + dy * dy + dz * dz);
            const double w = on ? 1.0 / stdmax(sc.E(a), sse) : 0.0;
            const double gx = dx * invd, gy = dy * invd, gz = dz * invd;
            m0 += w * gx * gx; m1 += w * gx * gy; m2 += w * gx * gz;
            m3 += w * gy * gy; m4 += w * gy * gz; m5 += w * gz * gz;
        });
        const double m[6] = {m0, m1, m2, m3, m4, m5};
        const double det = sym3_cofactors(m, cf);
        if (det == 0.0) return ST_UPDATE_SKIPPED; /* inv() of an exactly singular J' W J throws */
        const double idet = 1.0 / det;
        KFPOS_UNROLL
        for (int k = 0; k < 6; ++k) c[k] = cf[k] * idet;
    }
    KFPOS_UNROLL
    for (int k = 0; k < 3; ++k) pos[k] = p[k];
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) cov[k] = c[k];
    return pack_status(0, 0, it, -1);
}

/* ================================================================== 6-state filter (KalmanFilterTOA) */
template <bool SYMM>
struct Tag6 {
    double pos[3];
    Cov<6, SYMM> P;
};

/* x <- F x is the identity on position (velocity restarts at 0); P <- F P F' + Q.
 * KalmanFilterTOA.cpp:115-123, 362-391. */
template <bool SYMM>
KFPOS_FN void predict6(Cov<6, SYMM> &P, double t, double accel_noise) {
    const double t2 = (t * t) / 2, a2 = accel_noise * accel_noise;
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) {
        KFPOS_UNROLL
        for (int j = SYMM ? i : 0; j < 3; ++j) {
            /* pp += t (pv + vp) + t^2 vv */
            P(i, j) = P(i, j) + t * (P(i, 3 + j) + P(3 + i, j)) + (t * t) * P(3 + i, 3 + j);
        }
    }
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) {
        KFPOS_UNROLL
        for (int j = 0; j < 3; ++j) {
            P(i, 3 + j) = P(i, 3 + j) + t * P(3 + i, 3 + j);
            if (!SYMM) P(3 + i, j) = P(3 + i, j) + t * P(3 + i, 3 + j);
        }
    }
    KFPOS_UNROLL
    for (int k = 0; k < 3; ++k) {
        P(k, k) += a2 * t2 * t2;
        P(k, 3 + k) += a2 * t2 * t;
        if (!SYMM) P(3 + k, k) += a2 * t2 * t;
        P(3 + k, 3 + k) += a2 * t * t;
    }
}

/* ---- exact delta' pinv(P) delta for the one case without a closed form --------------------------------------
 * With ML initialisation the 6-state filter's P is NON-symmetric (column slip, KalmanFilterTOA.cpp:102-104) and, until
 * enough process noise has been added (one epoch; longer while dt = 0), rank-deficient: then delta' pinv(P) delta of
 * the convergence cost (KalmanFilterTOA.cpp:290, 303) is not w' P_pp w. Those epochs take the reference's own route,
 * an SVD pseudo-inverse with its tolerance max(m,n) sigma_max eps. Everything else keeps the closed form. */
struct Pinv6 {
    bool on;     /* false: P is comfortably full rank (or symmetric), use the closed form */
    double *a;   /* pinv(P), row-major, element k at a[k * stride]: parked outside the register file (LDS on the GPU) */
    int stride;
};

/* May the 6x6 P be rank-deficient? Cholesky of P'P without pivoting: a pivot below 1e-10 of the largest diagonal
 * entry (singular-value ratio below 1e-5; rounding leaves up to ~1e-14 there for an exactly singular P) says
 * "suspect". Liberal on purpose: the SVD path is always right, the closed form only needs an invertible P. */
KFPOS_FN bool cov6_suspect(const Cov<6, false> &P) {
    double g[6][6], gmax = 0.0;
    KFPOS_UNROLL
    for (int i = 0; i < 6; ++i) {
        KFPOS_UNROLL
        for (int j = 0; j <= i; ++j) {
            double v = 0.0;
            KFPOS_UNROLL
            for (int k = 0; k < 6; ++k) v += P(k, i) * P(k, j);
            g[i][j] = v;
        }
        gmax = (g[i][i] > gmax) ? g[i][i] : gmax;
    }
    bool suspect = false;
    KFPOS_UNROLL
    for (int j = 0; j < 6; ++j) {
        double d = g[j][j];
        KFPOS_UNROLL
        for (int k = 0; k < j; ++k) d -= g[j][k] * g[j][k];
        suspect = suspect || !(d > 1e-10 * gmax);
        const double id = (d > 0.0) ? 1.0 / sqrt(d) : 0.0;
        KFPOS_UNROLL
        for (int i = j + 1; i < 6; ++i) {
            double v = g[i][j];
            KFPOS_UNROLL
            for (int k = 0; k < j; ++k) v -= g[i][k] * g[j][k];
            g[i][j] = v * id;
        }
    }
    return suspect;
}

/* pinv(P) by one-sided (Hestenes) Jacobi: rotate column pairs of A = P until they are orthogonal, accumulating the
 * rotations in V; then P = U S V' with s_j = |a_j|, u_j = a_j / s_j, and pinv(P) = sum over s_j > tol of
 * v_j a_j' / s_j^2. The same sweep order, rotation formulas and tolerance as the oracle's restatement of arma::pinv. */
KFPOS_FN void pinv6_jacobi(const Cov<6, false> &P, double *out, int stride) {
    const double EPS = 2.220446049250313e-16;
    double a[6][6], v[6][6];
    KFPOS_UNROLL
    for (int i = 0; i < 6; ++i) {
        KFPOS_UNROLL
        for (int j = 0; j < 6; ++j) { a[i][j] = P(i, j); v[i][j] = (i == j) ? 1.0 : 0.0; }
    }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        KFPOS_UNROLL
        for (int p = 0; p < 5; ++p) {
            KFPOS_UNROLL
            for (int q = p + 1; q < 6; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                KFPOS_UNROLL
                for (int i = 0; i < 6; ++i) {
                    alpha += a[i][p] * a[i][p];
                    beta += a[i][q] * a[i][q];
                    gamma += a[i][p] * a[i][q];
                }
                const bool rot = (gamma != 0.0) && (fabs(gamma) > EPS * sqrt(alpha * beta));
                rotated = rotated || rot;
                const double zeta = (beta - alpha) / (2.0 * (rot ? gamma : 1.0));
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c0 = 1.0 / sqrt(1.0 + t * t);
                const double cs = rot ? c0 : 1.0, sn = rot ? c0 * t : 0.0; /* identity where no rotation is due */
                KFPOS_UNROLL
                for (int i = 0; i < 6; ++i) {
                    const double up = a[i][p], uq = a[i][q];
                    a[i][p] = cs * up - sn * uq;
                    a[i][q] = sn * up + cs * uq;
                    const double vp = v[i][p], vq = v[i][q];
                    v[i][p] = cs * vp - sn * vq;
                    v[i][q] = sn * vp + cs * vq;
                }
            }
        }
        if (!rotated) break;
    }
    double s2[6], smax2 = 0.0;
    KFPOS_UNROLL
    for (int j = 0; j < 6; ++j) {
        double n2 = 0;
        KFPOS_UNROLL
        for (int i = 0; i < 6; ++i) n2 += a[i][j] * a[i][j];
        s2[j] = n2;
        smax2 = (n2 > smax2) ? n2 : smax2;
    }
    const double tol = 6.0 * sqrt(smax2) * EPS; /* max(m, n) * sigma_max * eps */
    double inv[6];
    KFPOS_UNROLL
    for (int j = 0; j < 6; ++j) inv[j] = (sqrt(s2[j]) > tol) ? 1.0 / s2[j] : 0.0;
    KFPOS_UNROLL
    for (int i = 0; i < 6; ++i) {
        KFPOS_UNROLL
        for (int k = 0; k < 6; ++k) {
            double acc = 0.0;
            KFPOS_UNROLL
            for (int j = 0; j < 6; ++j) acc += v[i][j] * inv[j] * a[k][j];
            out[(6 * i + k) * stride] = acc;
        }
    }
}

/* Is |M P_pp| (bounded by sum of weights x trace) large enough for the adjugate's cancellation to matter? */
template <class SC, class COV>
KFPOS_FN bool illconditioned(const SC &sc, const Params &pr, const COV &P) {
    double wsum = 0.0;
    for_anchors<SC>(pr, [&](int a) { wsum += sc.W(a); });
    wsum = group_sum(sc, wsum);
    return wsum * (fabs(P(0, 0)) + fabs(P(1, 1)) + fabs(P(2, 2))) > 1e3;
}

struct Iekf6Out {
    double p[3];      /* updated position */
    double mlast[6];  /* M = G' R^-1 G of the last gain iteration */
    double cost;
    int gain_iters, ml_iters;
    uint32_t flags;
    bool pivot;       /* this step's 3x3 systems go through gauss3_solve (illconditioned()) */
};

/* First half of kalmanStep3DIgnoreAnchor (KalmanFilterTOA.cpp:268-282): ML position -> observation
 * covariance. Touches the position and the epoch only, not P, so the kernels run it while the covariance
 * loads are still in flight. Leaves sc.w = 1/R. */
template <class SC>
KFPOS_FN void iekf6_weights(const double xhat_p[3], SC &sc, const Params &pr, uint64_t drop, int n_used,
                            Iekf6Out &o) {
    o.flags = (n_used < 4) ? ST_FEW_RANGES : 0u;
    double pml[3] = {xhat_p[0], xhat_p[1], xhat_p[2]}, e_ml;
    set_weights_ml(sc, pr, drop);
    o.ml_iters = ml_estimate(pml, sc, pr, drop, n_used, e_ml);
    if (ml_covariance_throws(sc, pr, drop, n_used, e_ml)) o.flags |= ST_UPDATE_SKIPPED;
    if (isnan(pml[0]) || isnan(pml[1]) || isnan(pml[2])) {
        o.flags |= ST_ML_FALLBACK;
        e_ml = (n_used == 0) ? -1.0 : ml_sse(xhat_p, sc, pr, drop);
    }
    set_weights_iekf(sc, pr, e_ml, drop);
}

/* Second half (KalmanFilterTOA.cpp:285-324): the IEKF loop, up to, not including, the covariance
 * update. xhat_p: predicted position; P: predicted covariance; drop: ignored anchors. */
template <bool SYMM, class SC>
KFPOS_FN void iekf6(const double xhat_p[3], const Cov<6, SYMM> &P, SC &sc,
                    const Params &pr, uint64_t drop, int max_steps, double tol, Iekf6Out &o, const Pinv6 &pinv) {
    double p[3] = {xhat_p[0], xhat_p[1], xhat_p[2]};
    double dp[3] = {0.0, 0.0, 0.0}; /* delta_p = xhat_p - p */
    o.pivot = illconditioned(sc, pr, P);
    double qd = 0.0;               /* delta' pinv(P) delta */
    double cost = 1e20;
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) o.mlast[k] = 0.0;
    o.gain_iters = 0;
    for (int iter = 0; iter < max_steps; ++iter) {
        double c = 0.0, m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, u0 = 0, u1 = 0, u2 = 0;
        for_anchors<SC>(pr, [&](int a) {
            double bx, by, bz;
            anchor_of(sc, pr, a, bx, by, bz);
            const double dx = p[0] - bx, dy = p[1] - by, dz = p[2] - bz;
            double d, invd;
            kf_sqrt_rsqrt_sweep(dx * dx + dy * dy + dz * dz, d, invd);
            const double w = sc.W(a), y = sc.R(a) - d; /* weight 0 = absent / dropped range (set_weights_iekf) */
            const double yw = y * w;
            c += y * yw;
            const double gx = dx * invd, gy = dy * invd, gz = dz * invd;
            u0 += gx * yw; u1 += gy * yw; u2 += gz * yw;
            const double wx = w * gx, wy = w * gy, wz = w * gz;
            m0 += wx * gx; m1 += wx * gy; m2 += wx * gz;
            m3 += wy * gy; m4 += wy * gz; m5 += wz * gz;
        });
        if constexpr (SC::COOP) { /* one anchor per lane: combine the group's partial sums */
            c = group_sum(sc, c);
            m0 = group_sum(sc, m0); m1 = group_sum(sc, m1); m2 = group_sum(sc, m2);
            m3 = group_sum(sc, m3); m4 = group_sum(sc, m4); m5 = group_sum(sc, m5);
            u0 = group_sum(sc, u0); u1 = group_sum(sc, u1); u2 = group_sum(sc, u2);
        }
        c += qd;
        /* u = G' R^-1 (y - G delta) = G' R^-1 y - M delta: the delta term once per pass, not once per anchor */
        const double m[6] = {m0, m1, m2, m3, m4, m5},
                     u[3] = {u0 - (m0 * dp[0] + m1 * dp[1] + m2 * dp[2]), u1 - (m1 * dp[0] + m3 * dp[1] + m4 * dp[2]),
                             u2 - (m2 * dp[0] + m4 * dp[1] + m5 * dp[2])};
        if (fabs(cost - c) / cost < tol) break; /* KalmanFilterTOA.cpp:307 */
        cost = c;
        KFPOS_UNROLL
        for (int k = 0; k < 6; ++k) o.mlast[k] = m[k];
        /* w3 = (I + M Ppp)^-1 u */
        const double mm[3][3] = {{m[0], m[1], m[2]}, {m[1], m[3], m[4]}, {m[2], m[4], m[5]}};
        double a33[9], adj[9];
        KFPOS_UNROLL
        for (int i = 0; i < 3; ++i) {
            KFPOS_UNROLL
            for (int j = 0; j < 3; ++j)
                a33[3 * i + j] = (i == j ? 1.0 : 0.0) + mm[i][0] * P(0, j) + mm[i][1] * P(1, j) + mm[i][2] * P(2, j);
        }
        const double idet = kf_rcp(gen3_adjugate(a33, adj));
        double w3[3];
        KFPOS_UNROLL
        for (int i = 0; i < 3; ++i) w3[i] = (adj[3 * i] * u[0] + adj[3 * i + 1] * u[1] + adj[3 * i + 2] * u[2]) * idet;
        if (o.pivot) gauss_solve<3, 1>(a33, u, w3);
        qd = 0.0;
        KFPOS_UNROLL
        for (int i = 0; i < 3; ++i) {
            const double s = P(i, 0) * w3[0] + P(i, 1) * w3[1] + P(i, 2) * w3[2];
            p[i] = xhat_p[i] + s;
            dp[i] = -s;
            qd += w3[i] * s;
        }
        if (!SYMM && pinv.on) { /* rank-deficient non-symmetric P: delta' pinv(P) delta as the reference forms it */
            double dl[6];
            KFPOS_UNROLL
            for (int i = 0; i < 6; ++i) dl[i] = -(P(i, 0) * w3[0] + P(i, 1) * w3[1] + P(i, 2) * w3[2]);
            qd = 0.0;
            KFPOS_UNROLL
            for (int i = 0; i < 6; ++i) {
                double r = 0.0;
                KFPOS_UNROLL
                for (int j = 0; j < 6; ++j) r += pinv.a[(6 * i + j) * pinv.stride] * dl[j];
                qd += dl[i] * r;
            }
        }
        o.gain_iters++;
    }
    o.p[0] = p[0]; o.p[1] = p[1]; o.p[2] = p[2];
    o.cost = cost;
}

/* P <- (I - K H) P = P - P[:,0:3] N P[0:3,:], N = (I + M Ppp)^-1 M (KalmanFilterTOA.cpp:326) */
template <bool SYMM>
KFPOS_FN void cov_update6(Cov<6, SYMM> &P, const double m[6], bool pivot) {
    const double mm[3][3] = {{m[0], m[1], m[2]}, {m[1], m[3], m[4]}, {m[2], m[4], m[5]}};
    double a33[9], adj[9], nn[3][3];
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) {
        KFPOS_UNROLL
        for (int j = 0; j < 3; ++j)
            a33[3 * i + j] = (i == j ? 1.0 : 0.0) + mm[i][0] * P(0, j) + mm[i][1] * P(1, j) + mm[i][2] * P(2, j);
    }
    const double idet = kf_rcp(gen3_adjugate(a33, adj));
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) {
        KFPOS_UNROLL
        for (int j = 0; j < 3; ++j)
            nn[i][j] = (adj[3 * i] * mm[0][j] + adj[3 * i + 1] * mm[1][j] + adj[3 * i + 2] * mm[2][j]) * idet;
    }
    if (pivot) gauss_solve<3, 3>(a33, &mm[0][0], &nn[0][0]);
    /* V = N P[0:3,:] (3x6), then P(i,j) -= sum_k P(i,k) V(k,j) using the old P(:,0:3) column block */
    double v[3][6], c0[6][3];
    KFPOS_UNROLL
    for (int k = 0; k < 3; ++k) {
        KFPOS_UNROLL
        for (int j = 0; j < 6; ++j) v[k][j] = nn[k][0] * P(0, j) + nn[k][1] * P(1, j) + nn[k][2] * P(2, j);
    }
    KFPOS_UNROLL
    for (int i = 0; i < 6; ++i) {
        KFPOS_UNROLL
        for (int k = 0; k < 3; ++k) c0[i][k] = P(i, k);
    }
    KFPOS_UNROLL
    for (int i = 0; i < 6; ++i) {
        KFPOS_UNROLL
        for (int j = SYMM ? i : 0; j < 6; ++j)
            P(i, j) = P(i, j) - (c0[i][0] * v[0][j] + c0[i][1] * v[1][j] + c0[i][2] * v[2][j]);
    }
}

/* KalmanFilterTOA::estimatePositionKF (KalmanFilterTOA.cpp:70-156) for one tag and one epoch.
 * sc holds the epoch (r in metres, e). Returns the status word. */
/* park: 36 doubles (element k at park[k * park_stride]) for the pseudo-inverse of the non-symmetric layout; unused
 * (may be null) with SYMM = true. */
/* HEUR: which outlier heuristics the bank may use -- 2: any (run-time flags decide), 1: top-N only (the caller
 * guarantees ignore_worst = 0), 0: none (ignore_worst = 0 and top_n = 0). Knowing it at compile time lets the
 * compiler drop the leave-one-out loop and the kept results: fewer instructions and registers for the plain filter
 * of BASELINE configs 2 and 4 and for the top-N composition of config 5. */
template <bool SYMM, int HEUR = 2, class SC>
KFPOS_FN uint32_t step_toa6(Tag6<SYMM> &tg, SC &sc, const Params &pr_in, double dt, double *park = nullptr,
                            int park_stride = 0) {
    Params pr = pr_in;
    if (HEUR < 2) pr.ignore_worst = 0;
    if (HEUR < 1) pr.top_n = 0;
    int n_valid = count_used(sc, pr, 0);
    if (!SYMM && !pr.use_init_pos && (isnan(tg.pos[0]) || isnan(tg.pos[1]) || isnan(tg.pos[2]))) {
        /* ML initialisation, KalmanFilterTOA.cpp:90-108. Only the non-symmetric layout gets here: a bank is created
         * with it exactly when use_init_pos = 0, so the symmetric instantiations carry no initialisation code */
        if (n_valid < 4) return ST_FEW_RANGES;
        double p[3] = {1.0, 1.0, 4.0}, sse, c[6];
        set_weights_ml(sc, pr, 0ull);
        const int it = ml_estimate(p, sc, pr, 0, n_valid, sse);
        if (ml_covariance_throws(sc, pr, 0, n_valid, sse)) return ST_UPDATE_SKIPPED; /* reference: abort */
        if (!ml_covariance(p, sc, pr, sse, c)) return ST_UPDATE_SKIPPED;
        tg.pos[0] = p[0]; tg.pos[1] = p[1]; tg.pos[2] = p[2];
        const double cm[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
        KFPOS_UNROLL
        for (int i = 0; i < 3; ++i) {
            tg.P(i, 0) = cm[i][0];
            tg.P(i, 1) = cm[i][1];
            if (!SYMM) tg.P(i, 2) = cm[i][1]; /* sic: column 1 again (KalmanFilterTOA.cpp:102-104) */
        }
        if (SYMM) tg.P(2, 2) = cm[2][1];
        return pack_status(ST_ML_INIT, 0, it, -1);
    }
    uint64_t drop = 0;
    if (pr.top_n > 0 && !(isnan(tg.pos[0]) || isnan(tg.pos[1]) || isnan(tg.pos[2]))) {
        drop = topn_mask(tg.pos, sc, pr, n_valid);
        n_valid = count_used(sc, pr, drop);
    }
    const double xhat_p[3] = {tg.pos[0], tg.pos[1], tg.pos[2]}; /* F x: velocity restarts at 0 */
    bool predicted = false;

    /* kalmanStep3DCanIgnoreAnAnchor (KalmanFilterTOA.cpp:185-238) runs one solve with every range and
     * one per left-out range, then adopts the left-out solve with the largest r_i - |p_(-i) - b_i| if
     * that is positive and lowers the cost by more than the threshold. Here one loop walks a virtual
     * index v: v = -1 solves with every range, v = 0..A-1 leaves range v out (lanes whose range v is
     * absent sit that trip out); the results of the all-ranges solve and of the best leave-one-out so far
     * are kept (a dozen registers each), so nothing is solved twice. Without the heuristic only v = A runs,
     * once. v is uniform across the wavefront (anchor coordinates stay scalar loads) and there is ONE call
     * site of the solver, so the kernel carries a single inlined copy of it. */
    Iekf6Out o = {}, o_all = {}, o_best = {};
    Pinv6 pinv{false, park, park_stride};
    int ignored = -1;
    const int A = SC::NA > 0 ? SC::NA : pr.n_anchors;
    const bool heuristic = n_valid > 4 && pr.ignore_worst;
    int i = 0, best_i = -1;
    double cost_all = 0.0, max_distance = 0.0, worst_cost = 0.0;
    bool thrown = false; /* one of the solves hit the reference's std::runtime_error */
    for (int v = pr.ignore_worst ? -1 : A; v <= A; ++v) {
        const bool last = (v == A);
        double ra = 0.0;
        bool active = last || (heuristic && v < 0);
        if (!last && v >= 0 && heuristic) {
            ra = sc.Rdyn(v);
            active = !((drop >> v) & 1ull) && ra > 0.0;
        }
        if (last && heuristic) { /* adopt one of the kept results (KalmanFilterTOA.cpp:225-233) */
            o = o_all;
            if (max_distance > 0.0 && (cost_all - worst_cost) > pr.cost_threshold) {
                o = o_best;
                ignored = best_i;
            }
            break;
        }
        if (!active) continue;
        const uint64_t mask = (last || v < 0) ? drop : (drop | (1ull << v));
        const int n_use = (last || v < 0) ? n_valid : n_valid - 1;
        iekf6_weights(xhat_p, sc, pr, mask, n_use, o);
        if (!predicted) { /* after the first ML solve: the covariance loads have landed by now */
            predict6(tg.P, dt, pr.accel_noise);
            predicted = true;
            if constexpr (!SYMM) {
                if (cov6_suspect(tg.P)) {
                    pinv.on = true;
                    pinv6_jacobi(tg.P, pinv.a, pinv.stride);
                }
            }
        }
        thrown = thrown || (o.flags & ST_UPDATE_SKIPPED);
        if (thrown) continue; /* the exception leaves kalmanStep3D*: nothing after it runs */
        iekf6(xhat_p, tg.P, sc, pr, mask, 10, 1e-3, o, pinv);
        if (last) break;
        if (v < 0) {
            cost_all = o.cost;
            o_all = o;
        } else {
            const double dx = pr.anchors[3 * v] - o.p[0], dy = pr.anchors[3 * v + 1] - o.p[1],
                         dz = pr.anchors[3 * v + 2] - o.p[2];
            const double diff = ra - sqrt(dx * dx + dy * dy + dz * dz);
            if (i == 0 || diff > max_distance) { /* KalmanFilterTOA.cpp:209-214 */
                max_distance = diff;
                worst_cost = o.cost;
                best_i = i;
                o_best = o;
            }
            ++i;
        }
    }
    if (thrown) return ST_UPDATE_SKIPPED; /* predicted covariance kept, position untouched (:151-153) */
    cov_update6(tg.P, o.mlast, o.pivot);
    tg.pos[0] = o.p[0]; tg.pos[1] = o.p[1]; tg.pos[2] = o.p[2];
    return pack_status(o.flags, o.gain_iters, o.ml_iters, ignored);
}

/* getPose (KalmanFilterTOA.cpp:438-473): predict-only; position block of F P F' + Q */
template <bool SYMM>
KFPOS_FN void pose6(const Tag6<SYMM> &tg, double t, double accel_noise, double pos[3], double cov[9]) {
    const double t2 = (t * t) / 2, a2 = accel_noise * accel_noise;
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) {
        pos[i] = tg.pos[i];
        KFPOS_UNROLL
        for (int j = 0; j < 3; ++j)
            cov[3 * i + j] = tg.P(i, j) + t * (tg.P(i, 3 + j) + tg.P(3 + i, j)) + (t * t) * tg.P(3 + i, 3 + j) +
                             (i == j ? a2 * t2 * t2 : 0.0);
    }
}

/* ================================================================== 9-state filter (KalmanFilterTOAIMU) */
struct Tag9 {
    double pos[3], vel[3];
    Cov<9, true> P;
};
struct Imu {
    bool has;      /* hasImuMeasurement */
    double acc[3]; /* linearAcceleration */
    double ci[6];  /* inverse Cholesky factor of the covariance, lower {00,10,20,11,21,22}: Sigma^-1 = Ci' Ci */
    double wi[6];  /* Sigma^-1 itself, symmetric {00,01,02,11,12,22} */
};

/* Sigma (row-major 3x3, symmetric positive definite; the lower triangle is read) -> Ci with
 * Sigma^-1 = Ci' Ci, and Sigma^-1 = Ci' Ci itself */
KFPOS_FN void imu_whitener(const double s[9], double ci[6], double wi[6]) {
    double c00, c11, c22, i00, i11, i22;
    kf_sqrt_rsqrt(s[0], c00, i00);
    const double c10 = s[3] * i00, c20 = s[6] * i00;
    kf_sqrt_rsqrt(s[4] - c10 * c10, c11, i11);
    const double c21 = (s[7] - c20 * c10) * i11;
    kf_sqrt_rsqrt(s[8] - c20 * c20 - c21 * c21, c22, i22);
    const double i10 = -c10 * i00 * i11;
    const double i21 = -c21 * i11 * i22;
    const double i20 = -(c20 * i00 + c21 * i10) * i22;
    ci[0] = i00; ci[1] = i10; ci[2] = i20; ci[3] = i11; ci[4] = i21; ci[5] = i22;
    wi[0] = i00 * i00 + i10 * i10 + i20 * i20;
    wi[1] = i10 * i11 + i20 * i21;
    wi[2] = i20 * i22;
    wi[3] = i11 * i11 + i21 * i21;
    wi[4] = i21 * i22;
    wi[5] = i22 * i22;
}

/* KalmanFilterTOAIMU.cpp:170-180, 392-421 */
KFPOS_FN void predict9(Cov<9, true> &P, double t, double jolt) {
    const double c = t * t / 2;
    /* blocks: p = 0..2, v = 3..5, a = 6..8; each block is updated from not-yet-overwritten ones */
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) {
        KFPOS_UNROLL
        for (int j = i; j < 3; ++j) {
            const double xpp = P(i, j) + t * P(3 + i, j) + c * P(6 + i, j);
            const double xpv = P(i, 3 + j) + t * P(3 + i, 3 + j) + c * P(6 + i, 3 + j);
            const double xpa = P(i, 6 + j) + t * P(3 + i, 6 + j) + c * P(6 + i, 6 + j);
            P(i, j) = xpp + t * xpv + c * xpa;
        }
    }
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) {
        KFPOS_UNROLL
        for (int j = 0; j < 3; ++j) {
            const double xpv = P(i, 3 + j) + t * P(3 + i, 3 + j) + c * P(6 + i, 3 + j);
            const double xpa = P(i, 6 + j) + t * P(3 + i, 6 + j) + c * P(6 + i, 6 + j);
            P(i, 3 + j) = xpv + t * xpa;
        }
    }
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) {
        KFPOS_UNROLL
        for (int j = 0; j < 3; ++j) P(i, 6 + j) = P(i, 6 + j) + t * P(3 + i, 6 + j) + c * P(6 + i, 6 + j);
    }
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) {
        KFPOS_UNROLL
        for (int j = i; j < 3; ++j) {
            const double xvv = P(3 + i, 3 + j) + t * P(6 + i, 3 + j);
            const double xva = P(3 + i, 6 + j) + t * P(6 + i, 6 + j);
            P(3 + i, 3 + j) = xvv + t * xva;
        }
    }
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) {
        KFPOS_UNROLL
        for (int j = 0; j < 3; ++j) P(3 + i, 6 + j) = P(3 + i, 6 + j) + t * P(6 + i, 6 + j);
    }
    const double u[3] = {(t * t * t) / 6, (t * t) / 2, t};
    KFPOS_UNROLL
    for (int k = 0; k < 3; ++k) {
        KFPOS_UNROLL
        for (int a = 0; a < 3; ++a) {
            KFPOS_UNROLL
            for (int b = a; b < 3; ++b) P(k + 3 * a, k + 3 * b) += jolt * u[a] * u[b];
        }
    }
}

/* index of the k-th updated state component: position 0..2, acceleration 6..8 */
KFPOS_HD constexpr int e9(int k) { return k < 3 ? k : k + 3; }

/* L = blockdiag(L_r, L_a): M_r = L_r L_r' (lower, psd Cholesky), M_a = D Sigma^-1 D = L_a L_a'
 * with L_a = D Ci' (upper). lt[k][l] = L(k, l) as a dense 6x6 with structural zeros. */
struct Factor9 {
    double lr[6], ilr[3]; /* chol3_psd of M_r */
    double la[6];         /* upper {00,01,02,11,12,22}: la(k,l) = a_k * ci(l,k) */
};
KFPOS_FN void factor9(const double mr[6], const double acc[3], const Imu &imu, Factor9 &f) {
    chol3_psd(mr, f.lr, f.ilr);
    if (imu.has) {
        f.la[0] = acc[0] * imu.ci[0]; f.la[1] = acc[0] * imu.ci[1]; f.la[2] = acc[0] * imu.ci[2];
        f.la[3] = acc[1] * imu.ci[3]; f.la[4] = acc[1] * imu.ci[4];
        f.la[5] = acc[2] * imu.ci[5];
    } else {
        KFPOS_UNROLL
        for (int k = 0; k < 6; ++k) f.la[k] = 0.0;
    }
}
/* dense access L(k,l), k,l in 0..5 (compile-time indices) */
KFPOS_FN double L9(const Factor9 &f, int k, int l) {
    if (k < 3 && l < 3) {
        if (k < l) return 0.0;
        /* lower packed {00,10,20,11,21,22} */
        return f.lr[l == 0 ? k : (l == 1 ? 2 + k : 5)];
    }
    if (k >= 3 && l >= 3) {
        const int i = k - 3, j = l - 3;
        if (i > j) return 0.0;
        return f.la[i == 0 ? j : (i == 1 ? 2 + j : 5)];
    }
    return 0.0;
}

struct Iekf9Out {
    double x[9];
    double mrlast[6], dlast[3];
    double cost;
    int gain_iters, ml_iters;
    uint32_t flags;
};

/* kalmanStep3D (KalmanFilterTOAIMU.cpp:242-340, with the 3-token repair), first part (:268-276): ML
 * position -> observation covariance of the ranging rows. Independent of P. */
template <class SC>
KFPOS_FN void iekf9_weights(const double xhat[9], SC &sc, const Params &pr, bool has_ranging, int n_used,
                            Iekf9Out &o) {
    o.flags = (has_ranging && n_used < 4) ? ST_FEW_RANGES : 0u;
    o.ml_iters = 0;
    if (has_ranging) {
        double pml[3] = {xhat[0], xhat[1], xhat[2]}, e_ml;
        set_weights_ml(sc, pr, 0ull);
        o.ml_iters = ml_estimate(pml, sc, pr, 0ull, n_used, e_ml); /* no NaN fallback in this filter */
        if (ml_covariance_throws(sc, pr, 0ull, n_used, e_ml)) o.flags |= ST_UPDATE_SKIPPED;
        set_weights_iekf(sc, pr, e_ml, 0ull);
    }
}

/* second part (:296-336): the IEKF loop, up to the covariance update */
/* DIAG: the accelerometer covariance of every lane of the wavefront is diagonal (the usual case: sensor_msgs::Imu
 * carries diag covariances), so Sigma^-1 and M_a = D Sigma^-1 D are diagonal: their 3x3 products collapse to
 * scalings (about 90 of the 725 instructions of an iteration). Same results: the skipped terms are exact zeros. */
template <bool DIAG, class SC>
KFPOS_FN void iekf9(const double xhat[9], const Cov<9, true> &P, SC &sc, const Params &pr,
                    bool has_ranging, const Imu &imu, int max_steps, double tol, Iekf9Out &o) {
    const uint64_t drop = has_ranging ? 0ull : ~0ull;
    double p[3] = {xhat[0], xhat[1], xhat[2]}, acc[3] = {xhat[6], xhat[7], xhat[8]};
    double de[6] = {0, 0, 0, 0, 0, 0}; /* delta on the updated components (position, acceleration) */
    double wl[6] = {0, 0, 0, 0, 0, 0};
    double qd = 0.0, cost = 1e20;
    const bool pivot = has_ranging && illconditioned(sc, pr, P);
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) o.mrlast[k] = 0.0;
    o.dlast[0] = o.dlast[1] = o.dlast[2] = 0.0;
    o.gain_iters = 0;
    for (int iter = 0; iter < max_steps; ++iter) {
        double c = qd, m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, u0 = 0, u1 = 0, u2 = 0;
        if (has_ranging) {
            for_anchors<SC>(pr, [&](int a) {
                const double dx = p[0] - pr.anchors[3 * a], dy = p[1] - pr.anchors[3 * a + 1],
                             dz = p[2] - pr.anchors[3 * a + 2];
                double d, invd;
                kf_sqrt_rsqrt_sweep(dx * dx + dy * dy + dz * dz, d, invd);
                const double w = sc.W(a), y = sc.R(a) - d; /* weight 0 = absent / dropped range (set_weights_iekf) */
                const double yw = y * w;
                c += y * yw;
                const double gx = dx * invd, gy = dy * invd, gz = dz * invd;
                u0 += gx * yw; u1 += gy * yw; u2 += gz * yw;
                const double wx = w * gx, wy = w * gy, wz = w * gz;
                m0 += wx * gx; m1 += wx * gy; m2 += wx * gz;
                m3 += wy * gy; m4 += wy * gz; m5 += wz * gz;
            });
        }
        /* u_r = G' R^-1 (y - G delta_p) = G' R^-1 y - M_r delta_p: the delta term once per pass */
        const double m[6] = {m0, m1, m2, m3, m4, m5},
                     u[3] = {u0 - (m0 * de[0] + m1 * de[1] + m2 * de[2]), u1 - (m1 * de[0] + m3 * de[1] + m4 * de[2]),
                             u2 - (m2 * de[0] + m4 * de[1] + m5 * de[2])};
        /* IMU rows: y_a = z_a - a, cost += y_a' Sigma^-1 y_a, u_a = D Sigma^-1 (y_a - D delta_a),
         * M_a = D Sigma^-1 D with D = diag(a) (sic, KalmanFilterTOAIMU.cpp:441-473) */
        double ua[3] = {0, 0, 0}, ma[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        if (imu.has) {
            const double ya[3] = {imu.acc[0] - acc[0], imu.acc[1] - acc[1], imu.acc[2] - acc[2]};
            const double va[3] = {ya[0] - acc[0] * de[3], ya[1] - acc[1] * de[4], ya[2] - acc[2] * de[5]};
            if constexpr (DIAG) {
                const double wd[3] = {imu.wi[0], imu.wi[3], imu.wi[5]};
                KFPOS_UNROLL
                for (int i = 0; i < 3; ++i) { /* same operation order as the full path: bit-identical results */
                    const double wy = wd[i] * ya[i];
                    c += ya[i] * wy;
                    ua[i] = acc[i] * (wd[i] * va[i]);
                    ma[i][i] = acc[i] * acc[i] * wd[i];
                }
            } else {
                const double wm[3][3] = {{imu.wi[0], imu.wi[1], imu.wi[2]},
                                         {imu.wi[1], imu.wi[3], imu.wi[4]},
                                         {imu.wi[2], imu.wi[4], imu.wi[5]}};
                KFPOS_UNROLL
                for (int i = 0; i < 3; ++i) {
                    const double wy = wm[i][0] * ya[0] + wm[i][1] * ya[1] + wm[i][2] * ya[2];
                    c += ya[i] * wy;
                    ua[i] = acc[i] * (wm[i][0] * va[0] + wm[i][1] * va[1] + wm[i][2] * va[2]);
                    KFPOS_UNROLL
                    for (int j = 0; j < 3; ++j) ma[i][j] = acc[i] * acc[j] * wm[i][j];
                }
            }
        }
        if (fabs(cost - c) / cost < tol) break; /* KalmanFilterTOAIMU.cpp:316 */
        cost = c;
        KFPOS_UNROLL
        for (int k = 0; k < 6; ++k) o.mrlast[k] = m[k];
        o.dlast[0] = acc[0]; o.dlast[1] = acc[1]; o.dlast[2] = acc[2];

        /* w = (I + M P_ee)^-1 [u_r; u_a], M = blockdiag(M_r, M_a), by 3x3 blocks:
         *   [A11 A12; A21 A22] = I + [M_r Ppp, M_r Ppa; M_a Pap, M_a Paa]
         * A11 and the Schur complement are inverted through their adjugates (both have real
         * eigenvalues >= 1: products of PSD matrices shifted by I). */
        const double mr[3][3] = {{m[0], m[1], m[2]}, {m[1], m[3], m[4]}, {m[2], m[4], m[5]}};
        double a11[9], a12[3][3], a21[3][3], a22[3][3];
        KFPOS_UNROLL
        for (int i = 0; i < 3; ++i) {
            KFPOS_UNROLL
            for (int j = 0; j < 3; ++j) {
                a11[3 * i + j] = (i == j ? 1.0 : 0.0) + mr[i][0] * P(0, j) + mr[i][1] * P(1, j) + mr[i][2] * P(2, j);
                a12[i][j] = mr[i][0] * P(0, 6 + j) + mr[i][1] * P(1, 6 + j) + mr[i][2] * P(2, 6 + j);
                if constexpr (DIAG) {
                    a21[i][j] = ma[i][i] * P(6 + i, j);
                    a22[i][j] = (i == j ? 1.0 : 0.0) + ma[i][i] * P(6 + i, 6 + j);
                } else {
                    a21[i][j] = ma[i][0] * P(6, j) + ma[i][1] * P(7, j) + ma[i][2] * P(8, j);
                    a22[i][j] = (i == j ? 1.0 : 0.0) + ma[i][0] * P(6, 6 + j) + ma[i][1] * P(7, 6 + j) + ma[i][2] * P(8, 6 + j);
                }
            }
        }
        if (!pivot) {
            double adj[9];
            const double id1 = kf_rcp(gen3_adjugate(a11, adj));
            double xx[3][3], y1[3]; /* X = A11^-1 A12, y1 = A11^-1 u_r */
            KFPOS_UNROLL
            for (int i = 0; i < 3; ++i) {
                y1[i] = (adj[3 * i] * u[0] + adj[3 * i + 1] * u[1] + adj[3 * i + 2] * u[2]) * id1;
                KFPOS_UNROLL
                for (int j = 0; j < 3; ++j)
                    xx[i][j] = (adj[3 * i] * a12[0][j] + adj[3 * i + 1] * a12[1][j] + adj[3 * i + 2] * a12[2][j]) * id1;
            }
            double sc9[9], rhs[3]; /* Schur complement A22 - A21 X, rhs u_a - A21 y1 */
            KFPOS_UNROLL
            for (int i = 0; i < 3; ++i) {
                rhs[i] = ua[i] - (a21[i][0] * y1[0] + a21[i][1] * y1[1] + a21[i][2] * y1[2]);
                KFPOS_UNROLL
                for (int j = 0; j < 3; ++j)
                    sc9[3 * i + j] = a22[i][j] - (a21[i][0] * xx[0][j] + a21[i][1] * xx[1][j] + a21[i][2] * xx[2][j]);
            }
            const double id2 = kf_rcp(gen3_adjugate(sc9, adj));
            KFPOS_UNROLL
            for (int i = 0; i < 3; ++i)
                wl[3 + i] = (adj[3 * i] * rhs[0] + adj[3 * i + 1] * rhs[1] + adj[3 * i + 2] * rhs[2]) * id2;
            KFPOS_UNROLL
            for (int i = 0; i < 3; ++i) wl[i] = y1[i] - (xx[i][0] * wl[3] + xx[i][1] * wl[4] + xx[i][2] * wl[5]);
        } else { /* ill-conditioned step: the same block elimination with pivoted 3x3 solves instead of adjugates */
            double b4[12], x4[12]; /* [A12 | u_r] -> [X | y1] = A11^-1 [A12 | u_r] */
            KFPOS_UNROLL
            for (int i = 0; i < 3; ++i) {
                KFPOS_UNROLL
                for (int j = 0; j < 3; ++j) b4[4 * i + j] = a12[i][j];
                b4[4 * i + 3] = u[i];
            }
            gauss_solve<3, 4>(a11, b4, x4);
            double s9[9], r3[3], wa[3];
            KFPOS_UNROLL
            for (int i = 0; i < 3; ++i) {
                r3[i] = ua[i] - (a21[i][0] * x4[3] + a21[i][1] * x4[7] + a21[i][2] * x4[11]);
                KFPOS_UNROLL
                for (int j = 0; j < 3; ++j)
                    s9[3 * i + j] = a22[i][j] - (a21[i][0] * x4[j] + a21[i][1] * x4[4 + j] + a21[i][2] * x4[8 + j]);
            }
            gauss_solve<3, 1>(s9, r3, wa);
            KFPOS_UNROLL
            for (int i = 0; i < 3; ++i) {
                wl[3 + i] = wa[i];
                wl[i] = x4[4 * i + 3] - (x4[4 * i] * wa[0] + x4[4 * i + 1] * wa[1] + x4[4 * i + 2] * wa[2]);
            }
        }
        /* x_e = xhat_e + P_ee w ; delta_e = -P_ee w ; delta' pinv(P) delta = w . P_ee w */
        qd = 0.0;
        KFPOS_UNROLL
        for (int i = 0; i < 6; ++i) {
            double v = 0.0;
            KFPOS_UNROLL
            for (int k = 0; k < 6; ++k) v += P(e9(i), e9(k)) * wl[k];
            de[i] = -v;
            qd += wl[i] * v;
            if (i < 3) p[i] = xhat[i] + v;
            else acc[i - 3] = xhat[3 + i] + v;
        }
        o.gain_iters++;
    }
    KFPOS_UNROLL
    for (int i = 0; i < 9; ++i) {
        double v = 0.0;
        KFPOS_UNROLL
        for (int k = 0; k < 6; ++k) v += P(i, e9(k)) * wl[k];
        o.x[i] = xhat[i] + v;
    }
    o.cost = cost;
}

/* P <- (I - K H) P (KalmanFilterTOAIMU.cpp:338) as six rank-1 downdates along the columns of
 * E' L (unit-noise pseudo-measurements): P -= (P t)(P t)' / (1 + t' P t). */
KFPOS_FN void cov_update9(Cov<9, true> &P, const double mr[6], const double d[3], const Imu &imu) {
    Factor9 f;
    factor9(mr, d, imu, f);
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) {
        double pt[9], s = 1.0;
        KFPOS_UNROLL
        for (int i = 0; i < 9; ++i) {
            double v = 0.0;
            KFPOS_UNROLL
            for (int l = 0; l < 6; ++l) {
                const bool nz = (l < 3 && k < 3 && l >= k) || (l >= 3 && k >= 3 && l <= k);
                if (nz) v += P(i, e9(l)) * L9(f, l, k);
            }
            pt[i] = v;
        }
        KFPOS_UNROLL
        for (int l = 0; l < 6; ++l) {
            const bool nz = (l < 3 && k < 3 && l >= k) || (l >= 3 && k >= 3 && l <= k);
            if (nz) s += L9(f, l, k) * pt[e9(l)];
        }
        const double is = kf_rcp(s);
        KFPOS_UNROLL
        for (int i = 0; i < 9; ++i) {
            const double pi = pt[i] * is;
            KFPOS_UNROLL
            for (int j = i; j < 9; ++j) P(i, j) -= pi * pt[j];
        }
    }
}

/* KalmanFilterTOAIMU::estimatePositionKF (KalmanFilterTOAIMU.cpp:100-195) for one tag.
 * has_ranging = false is the IMU-only call of newIMUMeasurement (:91). */
template <class SC>
KFPOS_FN uint32_t step_imu9(Tag9 &tg, SC &sc, const Params &pr, double dt,
                                   bool has_ranging, const Imu &imu) {
    const int n_valid = has_ranging ? count_used(sc, pr, 0) : 0;
    if (!pr.use_init_pos && (isnan(tg.pos[0]) || isnan(tg.pos[1]))) { /* :121-122, z is not tested */
        if (!has_ranging) return 0;
        if (n_valid < 4) return ST_FEW_RANGES;
        double p[3] = {1.0, 1.0, 4.0}, sse, c[6];
        set_weights_ml(sc, pr, 0ull);
        const int it = ml_estimate(p, sc, pr, 0, n_valid, sse);
        if (ml_covariance_throws(sc, pr, 0, n_valid, sse)) return ST_UPDATE_SKIPPED; /* reference: abort */
        if (!ml_covariance(p, sc, pr, sse, c)) return ST_UPDATE_SKIPPED;
        tg.pos[0] = p[0]; tg.pos[1] = p[1]; tg.pos[2] = p[2];
        tg.P(0, 0) = c[0]; tg.P(0, 1) = c[1]; tg.P(1, 1) = c[3]; /* xy block only, :134-137 */
        return pack_status(ST_ML_INIT, 0, it, -1);
    }
    const double c = dt * dt / 2;
    double xhat[9];
    KFPOS_UNROLL
    for (int k = 0; k < 3; ++k) { /* acceleration restarts at 0 */
        xhat[k] = tg.pos[k] + dt * tg.vel[k] + c * 0.0;
        xhat[3 + k] = tg.vel[k] + dt * 0.0;
        xhat[6 + k] = 0.0;
    }
    Iekf9Out o;
    iekf9_weights(xhat, sc, pr, has_ranging, n_valid, o); /* needs position + epoch only ... */
    predict9(tg.P, dt, pr.jolt);                          /* ... so the covariance is first touched here */
    /* The 9-state filter has no try/catch: the reference node aborts here. This core keeps the predicted
     * covariance and reports the tag instead. */
    if (o.flags & ST_UPDATE_SKIPPED) return ST_UPDATE_SKIPPED;
    /* exact zeros only: the whitener of a diagonal covariance produces them */
    const bool diag = !imu.has || (imu.wi[1] == 0.0 && imu.wi[2] == 0.0 && imu.wi[4] == 0.0);
    if (KFPOS_WAVE_ALL(diag)) iekf9<true>(xhat, tg.P, sc, pr, has_ranging, imu, 20, 1e-4, o);
    else iekf9<false>(xhat, tg.P, sc, pr, has_ranging, imu, 20, 1e-4, o);
    cov_update9(tg.P, o.mrlast, o.dlast, imu);
    KFPOS_UNROLL
    for (int k = 0; k < 3; ++k) { tg.pos[k] = o.x[k]; tg.vel[k] = o.x[3 + k]; } /* :189-194 */
    return pack_status(o.flags, o.gain_iters, o.ml_iters, -1);
}

/* getPose (KalmanFilterTOAIMU.cpp:476-510): predicted position / velocity and the position block */
KFPOS_FN void pose9(const Tag9 &tg, double t, double jolt, double pos[3], double vel[3], double cov[9]) {
    const double c = t * t / 2, t3 = (t * t * t) / 6;
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) {
        pos[i] = tg.pos[i] + t * tg.vel[i];
        vel[i] = tg.vel[i];
        KFPOS_UNROLL
        for (int j = 0; j < 3; ++j) {
            const Cov<9, true> &P = tg.P;
            const double xpp = P(i, j) + t * P(3 + i, j) + c * P(6 + i, j);
            const double xpv = P(i, 3 + j) + t * P(3 + i, 3 + j) + c * P(6 + i, 3 + j);
            const double xpa = P(i, 6 + j) + t * P(3 + i, 6 + j) + c * P(6 + i, 6 + j);
            cov[3 * i + j] = xpp + t * xpv + c * xpa + (i == j ? jolt * t3 * t3 : 0.0);
        }
    }
}

/* ================================================================== 8-state planar filter (KalmanFilter) */
/* State [x y vx vy ax ay theta omega] at a fixed height z (mUWBtagZ). Ranging rows touch (x, y) only; the
 * PX4Flow, IMU and magnetometer / compass rows of kalmanStep3D (KalmanFilter.cpp:365-501) are optional
 * (template flag SENSORS) so that a ranging-only bank runs the closed-form 2x2 update. */
struct Tag8 {
    double xy[2], z, vel[2], ang, om;
    Cov<8, true> P;
};
/* rows of one estimatePositionKF call / latched samples of a tag (mHas*Measurement) */
enum : uint32_t { ROW_RANGING = 1u, ROW_PX4 = 2u, ROW_IMU = 4u, ROW_MAG = 8u };
struct Latch8 {
    uint32_t has;   /* ROW_PX4 | ROW_IMU | ROW_MAG */
    double px4[5];  /* vx, vy, gyroz, covarianceVelocity, covarianceGyroZ */
    double imu[8];  /* ax, ay, angularVelocityZ, covarianceAccelerationXY[4], covarianceAngularVelocityZ */
    double mag[2];  /* angle, covarianceMag */
};

KFPOS_FN double normalize_angle(double a) { /* KalmanFilter.cpp:699-706 */
    const double pi = 3.14159265358979323846;
    if (a > pi) return a - 2 * pi;
    if (a <= -pi) return a + 2 * pi;
    return a;
}

/* newPX4FlowMeasurement, KalmanFilter.cpp:102-128. f = integrationX, integrationY, integrationRotationZ,
 * integrationTime [us], quality. Returns false for quality 0: the sample is dropped before anything happens. */
KFPOS_FN bool px4_sample(const Params &pr, const double f[5], double out[5]) {
    const int quality = (int)f[4];
    const double sec = f[3] / 1000000.0;
    out[1] = f[1] / sec * pr.px4_height;
    out[0] = f[0] / sec * pr.px4_height;
    out[2] = f[2] / sec;
    if (quality == 0) return false;
    out[3] = (f[3] > 0) ? pr.px4_cov_vel / sec * pr.px4_height / quality : pr.px4_cov_vel * quality;
    out[4] = pr.px4_cov_gyro_z;
    return true;
}
/* newIMUMeasurement, KalmanFilter.cpp:139-170 */
KFPOS_FN void imu_sample8(const Params &pr, const double ang_vel[3], const double cov_ang_vel[9],
                          const double lin_acc[3], const double cov_acc[9], double out[8]) {
    out[0] = lin_acc[0];
    out[1] = lin_acc[1];
    out[2] = ang_vel[2];
    out[3] = pr.imu_fixed_cov_acc ? pr.imu_cov_acc : cov_acc[0];
    out[4] = cov_acc[1];
    out[5] = cov_acc[3];
    out[6] = pr.imu_fixed_cov_acc ? pr.imu_cov_acc : cov_acc[4];
    out[7] = pr.imu_fixed_cov_w ? pr.imu_cov_w : cov_ang_vel[8];
}

/* P <- E P E' for E = I + c e_i e_k' (row i += c row k, column i += c column k), in place on the packed
 * upper triangle */
template <int I, int K>
KFPOS_FN void congruence8(Cov<8, true> &P, double c) {
    const double pik = P(I, K), pkk = P(K, K);
    KFPOS_UNROLL
    for (int j = 0; j < 8; ++j)
        if (j != I) P(I, j) = P(I, j) + c * P(K, j);
    P(I, I) = P(I, I) + c * (2.0 * pik + c * pkk);
}
/* P <- F P F' + Q, KalmanFilter.cpp:583-609. F factors into elementary congruences per chain (x, vx, ax),
 * (y, vy, ay), (theta, omega). Note accelerationNoise enters the angle block un-squared (:606-607). */
KFPOS_FN void predict8(Cov<8, true> &P, double t, double accel_noise, double jolt) {
    const double t2 = (t * t) / 2;
    congruence8<0, 4>(P, t2); congruence8<0, 2>(P, t); congruence8<2, 4>(P, t);
    congruence8<1, 5>(P, t2); congruence8<1, 3>(P, t); congruence8<3, 5>(P, t);
    congruence8<6, 7>(P, t);
    const double u[3] = {(t * t * t) / 6, t2, t};
    KFPOS_UNROLL
    for (int k = 0; k < 2; ++k) {
        KFPOS_UNROLL
        for (int a = 0; a < 3; ++a) {
            KFPOS_UNROLL
            for (int b = a; b < 3; ++b) P(k + 2 * a, k + 2 * b) += jolt * u[a] * u[b];
        }
    }
    P(6, 6) += accel_noise * t2 * t2;
    P(6, 7) += accel_noise * t2 * t;
    P(7, 7) += accel_noise * t * t;
}

/* One sweep of MLLocation::estimatePosition2D (MLLocation.cpp:73-98) at (p0, p1, z): unweighted SSE
 * (estimationError is this routine's cost, :65, :107), gradient g and the 2x2 hs = {00, 01, 11} with the 1/e
 * weights in sc.w. Distances are 3-D. */
template <class SC>
KFPOS_FN void ml2d_sweep(const double p[2], double z, const SC &sc, const Params &pr, double &sse, double g[2],
                         double hs[3]) {
    double sse_ = 0.0, g0 = 0.0, g1 = 0.0, h0 = 0.0, h1 = 0.0, h3 = 0.0;
    for_anchors<SC>(pr, [&](int a) {
        const bool on = used(sc, a, 0);
        const double r = sc.R(a), w = sc.W(a); /* 0 for an absent / dropped range (set_weights_*) */
        const double dx = pr.anchors[3 * a] - p[0], dy = pr.anchors[3 * a + 1] - p[1], dz = pr.anchors[3 * a + 2] - z;
        double d, invd;
        kf_sqrt_rsqrt_sweep(dx * dx + dy * dy + dz * dz, d, invd);
        const double rd = r - d;
        sse_ += on ? rd * rd : 0.0;
        const double gi = rd * invd * w;
        g0 += gi * dx;
        g1 += gi * dy;
        const double q = r * invd;
        const double c0 = w * (1.0 - q), c1 = w * q * invd * invd;
        h0 += c0 + c1 * dx * dx;
        h1 += c1 * dx * dy;
        h3 += c0 + c1 * dy * dy;
    });
    sse = sse_;
    g[0] = g0; g[1] = g1;
    hs[0] = h0; hs[1] = h1; hs[2] = h3;
}

/* MLLocation::estimatePosition2D, MLLocation.cpp:48-143. p: seed in, estimate out (z stays). Requires
 * sc.w = 1/e. A tentative point that raises the SSE is rejected and, since newCost then equals cost, the loop
 * ends at the next test: the halved step is never used. One sweep per pass: the sweep at the tentative
 * point is also the gradient / Hessian of the next pass. (`tentativePos.z` is uninitialised in the
 * reference, :64; it is taken as the fixed height, see DESIGN.md.) Fewer than 3 ranges: seed returned. */
template <class SC>
KFPOS_FN int ml2d_estimate(double p[2], double z, const SC &sc, const Params &pr, int n_used, double &sse_out) {
    double sse, g[2], hs[3];
    if (n_used < 3) {
        if (n_used == 0) { sse_out = -1.0; return 0; }
        ml2d_sweep(p, z, sc, pr, sse, g, hs);
        sse_out = sse;
        return 0;
    }
    ml2d_sweep(p, z, sc, pr, sse, g, hs);
    double cost = 1e20, newCost = sse;
    int iter = 0;
    while ((fabs(cost - newCost) / cost > 1e-3) && (iter < 10000)) {
        iter += 1;
        cost = newCost;
        const double idet = kf_rcp(hs[0] * hs[2] - hs[1] * hs[1]);
        const double q[2] = {p[0] - (hs[2] * g[0] - hs[1] * g[1]) * idet, p[1] - (hs[0] * g[1] - hs[1] * g[0]) * idet};
        double tc, g2[2], hs2[3];
        ml2d_sweep(q, z, sc, pr, tc, g2, hs2);
        if (tc > cost) break; /* :110-112 */
        newCost = tc;
        p[0] = q[0]; p[1] = q[1];
        g[0] = g2[0]; g[1] = g2[1];
        hs[0] = hs2[0]; hs[1] = hs2[1]; hs[2] = hs2[2];
    }
    sse_out = newCost;
    return iter;
}

/* inv(J' diag(max(e, e_ML))^-1 J) of estimatePosition2D (MLLocation.cpp:122-140), {00, 01, 11} */
template <class SC>
KFPOS_FN bool ml2d_covariance(const double p[2], double z, const SC &sc, const Params &pr, double sse, double cov[3]) {
    double m0 = 0, m1 = 0, m3 = 0;
    for_anchors<SC>(pr, [&](int a) {
        if (!used(sc, a, 0)) return;
        const double dx = p[0] - pr.anchors[3 * a], dy = p[1] - pr.anchors[3 * a + 1], dz = z - pr.anchors[3 * a + 2];
        const double invd = 1.0 / sqrt(dx * dx + dy * dy + dz * dz);
        const double w = 1.0 / stdmax(sc.E(a), sse);
        const double gx = dx * invd, gy = dy * invd;
        m0 += w * gx * gx; m1 += w * gx * gy; m3 += w * gy * gy;
    });
    const double det = m0 * m3 - m1 * m1;
    const double idet = 1.0 / det;
    cov[0] = m3 * idet; cov[1] = -m1 * idet; cov[2] = m0 * idet;
    return det != 0.0; /* as ml_covariance */
}

/* One scalar row of the linearised update, processed sequentially (rows with uncorrelated noise may be
 * absorbed one after the other): h has its non-zeros hv at the compile-time columns IX, noise variance R,
 * linearised innovation r. dl is the running state offset from the prediction, P the running covariance.
 *   s = P h; alpha = 1 / (h.s + R); dl += alpha s (r - h.dl); P -= alpha s s'.
 * Also accumulates what the cost and delta' pinv(P) delta need. `on` = false is a branch-free no-op. */
template <int... IX>
KFPOS_FN void seq_row8(Cov<8, true> &P, double dl[8], const double (&hv)[sizeof...(IX)], double R, double r, bool on) {
    constexpr int NNZ = sizeof...(IX);
    constexpr int ix[NNZ] = {IX...}; /* compile-time columns: every array index below is static after unrolling */
    double h[NNZ], s[8], hs = on ? R : 1.0, hd = 0.0;
    KFPOS_UNROLL
    for (int k = 0; k < NNZ; ++k) h[k] = on ? hv[k] : 0.0; /* an absent row may carry garbage (0/0 variances) */
    KFPOS_UNROLL
    for (int i = 0; i < 8; ++i) {
        double v = 0.0;
        KFPOS_UNROLL
        for (int k = 0; k < NNZ; ++k) v += P(i, ix[k]) * h[k];
        s[i] = v;
    }
    KFPOS_UNROLL
    for (int k = 0; k < NNZ; ++k) { hs += h[k] * s[ix[k]]; hd += h[k] * dl[ix[k]]; }
    const double alpha = kf_rcp(hs);
    const double gain = on ? alpha * (r - hd) : 0.0;
    KFPOS_UNROLL
    for (int i = 0; i < 8; ++i) {
        dl[i] += gain * s[i];
        const double as = alpha * s[i];
        KFPOS_UNROLL
        for (int j = i; j < 8; ++j) P(i, j) -= as * s[j];
    }
}
/* h . dl for a sparse row */
template <int... IX>
KFPOS_FN double row_dot8(const double dl[8], const double (&hv)[sizeof...(IX)]) {
    constexpr int NNZ = sizeof...(IX);
    constexpr int ix[NNZ] = {IX...};
    double v = 0.0;
    KFPOS_UNROLL
    for (int k = 0; k < NNZ; ++k) v += hv[k] * dl[ix[k]];
    return v;
}

/* Read-only view of a packed 8x8 covariance parked outside the register file (LDS on the GPU, element k at
 * base[k * stride]): the sensor-row update needs the predicted covariance of every iteration while it
 * downdates a working copy, and two register-resident copies do not fit next to the epoch. */
struct CovSpill8 {
    double *base;
    int stride;
    KFPOS_HD double operator()(int i, int j) const { return base[Cov<8, true>::idx(i, j) * stride]; }
};

struct Iekf8Out {
    double x[8];
    double mlast[3]; /* ranging information block of the last gain iteration (SENSORS = false) */
    int gain_iters, ml_iters;
    uint32_t flags;
};

/* kalmanStep3D first part (KalmanFilter.cpp:403-411): 2-D ML position from the predicted (x, y) ->
 * observation variance max(e_ML, e_i) of the ranging rows. Leaves sc.w = 1/R. Independent of P. */
template <class SC>
KFPOS_FN void iekf8_weights(const double xhat[8], double z, SC &sc, const Params &pr, int n_used, Iekf8Out &o) {
    o.flags = (n_used < 3) ? ST_FEW_RANGES : 0u;
    double pml[2] = {xhat[0], xhat[1]}, e_ml;
    set_weights_ml(sc, pr, 0ull);
    o.ml_iters = ml2d_estimate(pml, z, sc, pr, n_used, e_ml);
    if (ml_covariance_throws(sc, pr, 0ull, n_used, e_ml, 3)) o.flags |= ST_UPDATE_SKIPPED;
    set_weights_iekf(sc, pr, e_ml, 0ull);
}

/* kalmanStep3D second part (KalmanFilter.cpp:444-500). rows: which row groups this call carries. With
 * SENSORS = false only ROW_RANGING may be set and P is left untouched (cov_update8 finishes the job);
 * with SENSORS = true, Pout receives (I - K H) P of the last gain iteration.
 *
 * The ranging block is absorbed in information form on (x, y): M = G' R^-1 G (2x2), u = G' R^-1 (y - G delta),
 *   w = (I + M Pxy)^-1 u, offset = P[:, xy] w, P -= P[:, xy] (I + M Pxy)^-1 M P[xy, :].
 * The sensor rows follow one at a time (seq_row8); the two accelerometer rows, whose noise is correlated,
 * go as one 2x2 block. delta' pinv(P) delta of the cost is u_tot . dl - dl' M_tot dl, summed row group by
 * row group (w = u - M P w for the joint solve, so w' P w = (u - M dl) . dl). */
template <bool SENSORS, class PM, class SC>
KFPOS_FN void iekf8(const double xhat[8], double z, const PM &P, Cov<8, true> &Pout, SC &sc,
                    const Params &pr, uint32_t rows, const Latch8 &lt, double t, Iekf8Out &o) {
    const bool has_r = (rows & ROW_RANGING) != 0;
    const uint64_t drop = has_r ? 0ull : ~0ull;
    double x[8], dl[8]; /* dl = x - xhat = -delta */
    KFPOS_UNROLL
    for (int k = 0; k < 8; ++k) { x[k] = xhat[k]; dl[k] = 0.0; }
    double qd = 0.0, cost = 1e20;
    o.mlast[0] = o.mlast[1] = o.mlast[2] = 0.0;
    o.gain_iters = 0;
    for (int iter = 0; iter < 20; ++iter) {
        double c = qd, m0 = 0, m1 = 0, m3 = 0, u0 = 0, u1 = 0;
        if (has_r || !SENSORS) {
            for_anchors<SC>(pr, [&](int a) {
                const double dx = x[0] - pr.anchors[3 * a], dy = x[1] - pr.anchors[3 * a + 1], dz = z - pr.anchors[3 * a + 2];
                double d, invd;
                kf_sqrt_rsqrt_sweep(dx * dx + dy * dy + dz * dz, d, invd);
                const double w = sc.W(a), y = sc.R(a) - d; /* weight 0 = absent range (set_weights_iekf) */
                const double yw = y * w;
                c += y * yw;
                const double gx = dx * invd, gy = dy * invd;
                u0 += gx * yw; u1 += gy * yw;
                const double wx = w * gx, wy = w * gy;
                m0 += wx * gx; m1 += wx * gy; m3 += wy * gy;
            });
            /* u = G' R^-1 (y - G delta) with delta = -dl: G' R^-1 y + M dl, the delta term once per pass */
            u0 += m0 * dl[0] + m1 * dl[1];
            u1 += m1 * dl[0] + m3 * dl[1];
        }
        /* sensor rows at the current linearisation point */
        double sn = 0.0, cs = 1.0, sw = 0.0, cw = 1.0;
        double ypx[3] = {0, 0, 0}, yim[3] = {0, 0, 0}, ymag = 0.0, idet_im = 0.0;
        if (SENSORS) {
            const double vx = x[2], vy = x[3], ax = x[4], ay = x[5], th = x[6], om = x[7];
            if (rows & (ROW_PX4 | ROW_IMU)) { sn = sin(th); cs = cos(th); }
            if (rows & ROW_PX4) { /* px4flowOutput, :558-566 */
                sw = sin(om * t); cw = cos(om * t);
                ypx[0] = lt.px4[0] - (cs * vx + sn * vy + 1 / t * ((1 - cw) * pr.px4_arm_p1 - sw * pr.px4_arm_p2));
                ypx[1] = lt.px4[1] - (-sn * vx + cs * vy + 1 / t * (sw * pr.px4_arm_p1 + (1 - cw) * pr.px4_arm_p2));
                ypx[2] = lt.px4[2] - om;
                c += ypx[0] * ypx[0] / lt.px4[3] + ypx[1] * ypx[1] / lt.px4[3] + ypx[2] * ypx[2] / lt.px4[4];
            }
            if (rows & ROW_IMU) { /* imuOutput, :568-576 */
                yim[0] = lt.imu[0] - (cs * ax + sn * ay);
                yim[1] = lt.imu[1] - (-sn * ax + cs * ay);
                yim[2] = lt.imu[2] - om;
                /* y' R^-1 y for the 2x2 block {c00 c01; c10 c11} through its adjugate */
                idet_im = 1.0 / (lt.imu[3] * lt.imu[6] - lt.imu[4] * lt.imu[5]);
                c += (yim[0] * yim[0] * lt.imu[6] - yim[0] * yim[1] * (lt.imu[4] + lt.imu[5]) + yim[1] * yim[1] * lt.imu[3]) * idet_im +
                     yim[2] * yim[2] / lt.imu[7];
            }
            if (rows & ROW_MAG) {
                ymag = normalize_angle(lt.mag[0] - th); /* :460-462 */
                c += ymag * ymag / lt.mag[1];
            }
        }
        if (fabs(cost - c) / cost < 1e-4) break; /* :473 */
        cost = c;
        o.mlast[0] = m0; o.mlast[1] = m1; o.mlast[2] = m3;

        /* ranging block: w = (I + M Pxy)^-1 u */
        const double a00 = 1.0 + m0 * P(0, 0) + m1 * P(0, 1), a01 = m0 * P(0, 1) + m1 * P(1, 1);
        const double a10 = m1 * P(0, 0) + m3 * P(0, 1), a11 = 1.0 + m1 * P(0, 1) + m3 * P(1, 1);
        const double idet = kf_rcp(a00 * a11 - a01 * a10);
        const double w0 = (a11 * u0 - a01 * u1) * idet, w1 = (a00 * u1 - a10 * u0) * idet;
        const double dprev[8] = {dl[0], dl[1], dl[2], dl[3], dl[4], dl[5], dl[6], dl[7]};
        KFPOS_UNROLL
        for (int i = 0; i < 8; ++i) dl[i] = P(i, 0) * w0 + P(i, 1) * w1;
        if (!SENSORS) {
            qd = w0 * dl[0] + w1 * dl[1]; /* w' Pxy w */
        } else {
            /* running covariance: P - P[:, xy] N P[xy, :], N = (I + M Pxy)^-1 M */
            const double n00 = (a11 * m0 - a01 * m1) * idet, n01 = (a11 * m1 - a01 * m3) * idet;
            const double n10 = (a00 * m1 - a10 * m0) * idet, n11 = (a00 * m3 - a10 * m1) * idet;
            double v0[8], v1[8];
            KFPOS_UNROLL
            for (int j = 0; j < 8; ++j) {
                v0[j] = n00 * P(0, j) + n01 * P(1, j);
                v1[j] = n10 * P(0, j) + n11 * P(1, j);
            }
            KFPOS_UNROLL
            for (int i = 0; i < 8; ++i) {
                KFPOS_UNROLL
                for (int j = i; j < 8; ++j) Pout(i, j) = P(i, j) - (P(i, 0) * v0[j] + P(i, 1) * v1[j]);
            }
            const double vx = x[2], vy = x[3], ax = x[4], ay = x[5];
            /* linearised innovations r = y - H delta = y + H dprev; the quadratic terms of qd use the final dl
             * and are added after the last row */
            const double hp0[4] = {cs, sn, -sn * vx + cs * vy, pr.px4_arm_p1 * sw - pr.px4_arm_p2 * cw};   /* :627-655 */
            const double hp1[4] = {-sn, cs, -cs * vx - sn * vy, pr.px4_arm_p1 * cw + pr.px4_arm_p2 * sw};
            const double one[1] = {1.0};
            const double rp0 = ypx[0] + row_dot8<2, 3, 6, 7>(dprev, hp0), rp1 = ypx[1] + row_dot8<2, 3, 6, 7>(dprev, hp1),
                         rp2 = ypx[2] + dprev[7];
            /* a row group no lane of the wavefront carries is skipped altogether (wave-uniform branch); within a
             * wavefront that carries it, lanes without it run it as a no-op */
            const bool onp = (rows & ROW_PX4) != 0;
            if (!KFPOS_WAVE_ALL(!onp)) {
                seq_row8<2, 3, 6, 7>(Pout, dl, hp0, lt.px4[3], rp0, onp);
                seq_row8<2, 3, 6, 7>(Pout, dl, hp1, lt.px4[3], rp1, onp);
                seq_row8<7>(Pout, dl, one, lt.px4[4], rp2, onp);
            }
            /* accelerometer pair (:657-686): decorrelate with the LDL' of its 2x2 noise block (taken as
             * symmetric: c01 is used for both off-diagonal entries): row1' = row1 - (c01 / c00) row0 */
            const double hi0[3] = {cs, sn, -sn * ax + cs * ay};
            const double hi1[3] = {-sn, cs, -cs * ax - sn * ay};
            const bool oni = (rows & ROW_IMU) != 0;
            const double ri0 = yim[0] + row_dot8<4, 5, 6>(dprev, hi0), ri1 = yim[1] + row_dot8<4, 5, 6>(dprev, hi1),
                         ri2 = yim[2] + dprev[7];
            const double lc = lt.imu[4] / lt.imu[3];
            const double hi1d[3] = {hi1[0] - lc * hi0[0], hi1[1] - lc * hi0[1], hi1[2] - lc * hi0[2]};
            if (!KFPOS_WAVE_ALL(!oni)) {
                seq_row8<4, 5, 6>(Pout, dl, hi0, lt.imu[3], ri0, oni);
                seq_row8<4, 5, 6>(Pout, dl, hi1d, lt.imu[6] - lc * lt.imu[4], ri1 - lc * ri0, oni);
                seq_row8<7>(Pout, dl, one, lt.imu[7], ri2, oni);
            }
            const bool onm = (rows & ROW_MAG) != 0;
            const double rm = ymag + dprev[6];
            if (!KFPOS_WAVE_ALL(!onm)) seq_row8<6>(Pout, dl, one, lt.mag[1], rm, onm);
            /* delta' pinv(P) delta = sum over row groups of (H dl)' R^-1 (r - H dl), at the final dl */
            qd = u0 * dl[0] + u1 * dl[1] - (m0 * dl[0] * dl[0] + 2.0 * m1 * dl[0] * dl[1] + m3 * dl[1] * dl[1]);
            if (onp) {
                const double a0 = row_dot8<2, 3, 6, 7>(dl, hp0), a1 = row_dot8<2, 3, 6, 7>(dl, hp1), a2 = dl[7];
                qd += a0 * (rp0 - a0) / lt.px4[3] + a1 * (rp1 - a1) / lt.px4[3] + a2 * (rp2 - a2) / lt.px4[4];
            }
            if (oni) {
                const double a0 = row_dot8<4, 5, 6>(dl, hi0), a1 = row_dot8<4, 5, 6>(dl, hi1), a2 = dl[7];
                const double e0 = ri0 - a0, e1 = ri1 - a1; /* a' R^-1 e, R symmetric */
                qd += (a0 * e0 * lt.imu[6] - (a0 * e1 + a1 * e0) * lt.imu[4] + a1 * e1 * lt.imu[3]) /
                          (lt.imu[3] * lt.imu[6] - lt.imu[4] * lt.imu[4]) +
                      a2 * (ri2 - a2) / lt.imu[7];
            }
            if (onm) qd += dl[6] * (rm - dl[6]) / lt.mag[1];
        }
        KFPOS_UNROLL
        for (int i = 0; i < 8; ++i) x[i] = xhat[i] + dl[i];
        o.gain_iters++;
    }
    KFPOS_UNROLL
    for (int i = 0; i < 8; ++i) o.x[i] = x[i];
}

/* P <- (I - K H) P for ranging rows only: P - P[:, xy] (I + M Pxy)^-1 M P[xy, :] */
KFPOS_FN void cov_update8(Cov<8, true> &P, const double m[3]) {
    const double a00 = 1.0 + m[0] * P(0, 0) + m[1] * P(0, 1), a01 = m[0] * P(0, 1) + m[1] * P(1, 1);
    const double a10 = m[1] * P(0, 0) + m[2] * P(0, 1), a11 = 1.0 + m[1] * P(0, 1) + m[2] * P(1, 1);
    const double idet = kf_rcp(a00 * a11 - a01 * a10);
    const double n00 = (a11 * m[0] - a01 * m[1]) * idet, n01 = (a11 * m[1] - a01 * m[2]) * idet;
    const double n10 = (a00 * m[1] - a10 * m[0]) * idet, n11 = (a00 * m[2] - a10 * m[1]) * idet;
    double v0[8], v1[8], c0[8], c1[8];
    KFPOS_UNROLL
    for (int j = 0; j < 8; ++j) {
        c0[j] = P(0, j); c1[j] = P(1, j);
        v0[j] = n00 * c0[j] + n01 * c1[j];
        v1[j] = n10 * c0[j] + n11 * c1[j];
    }
    KFPOS_UNROLL
    for (int i = 0; i < 8; ++i) {
        KFPOS_UNROLL
        for (int j = i; j < 8; ++j) P(i, j) = P(i, j) - (c0[i] * v0[j] + c1[i] * v1[j]);
    }
}

/* KalmanFilter::estimatePositionKF (KalmanFilter.cpp:224-321) for one tag and one call carrying `rows`. */
template <bool SENSORS, class SC>
KFPOS_FN uint32_t step_planar8(Tag8 &tg, SC &sc, const Params &pr, double dt, uint32_t rows, const Latch8 &lt,
                               CovSpill8 spill = CovSpill8{nullptr, 0}) {
    const bool has_r = (rows & ROW_RANGING) != 0;
    const int n_valid = has_r ? count_used(sc, pr, 0) : 0;
    if (!pr.use_init_pos && (isnan(tg.xy[0]) || isnan(tg.xy[1]))) { /* :243-278 */
        if (!has_r) return 0;
        int it;
        double c00, c01, c11;
        set_weights_ml(sc, pr, 0ull);
        if (pr.use_fixed_height) {
            if (n_valid < 3) return ST_FEW_RANGES; /* the reference indexes an empty covariance here: abort */
            double p[2] = {1.0, 1.0}, sse, c[3];
            it = ml2d_estimate(p, tg.z, sc, pr, n_valid, sse);
            if (ml_covariance_throws(sc, pr, 0, n_valid, sse, 3)) return ST_UPDATE_SKIPPED;
            if (!ml2d_covariance(p, tg.z, sc, pr, sse, c)) return ST_UPDATE_SKIPPED;
            tg.xy[0] = p[0]; tg.xy[1] = p[1];
            c00 = c[0]; c01 = c[1]; c11 = c[2];
        } else {
            if (n_valid < 4) return ST_FEW_RANGES;
            double p[3] = {1.0, 1.0, 4.0}, sse, c[6];
            it = ml_estimate(p, sc, pr, 0, n_valid, sse);
            if (ml_covariance_throws(sc, pr, 0, n_valid, sse)) return ST_UPDATE_SKIPPED;
            if (!ml_covariance(p, sc, pr, sse, c)) return ST_UPDATE_SKIPPED;
            tg.xy[0] = p[0]; tg.xy[1] = p[1];
            tg.z = p[2]; /* mUWBtagZ = mPosition.z, :257 */
            c00 = c[0]; c01 = c[1]; c11 = c[3];
        }
        tg.P(0, 0) = c00; tg.P(0, 1) = c01; tg.P(1, 1) = c11;
        return pack_status(ST_ML_INIT, 0, it, -1);
    }
    /* predicted state: the acceleration restarts at 0 (mAcceleration is never written back) */
    const double xhat[8] = {tg.xy[0] + dt * tg.vel[0], tg.xy[1] + dt * tg.vel[1], tg.vel[0], tg.vel[1], 0.0, 0.0,
                            normalize_angle(tg.ang + dt * tg.om), tg.om};
    Iekf8Out o;
    o.flags = 0;
    o.ml_iters = 0;
    if (has_r) iekf8_weights(xhat, tg.z, sc, pr, n_valid, o);
    predict8(tg.P, dt, pr.accel_noise, pr.jolt);
    if (SENSORS) { /* inv(observationCovariance) throws on a zero variance (:446) */
        bool bad = false;
        if (rows & ROW_PX4) bad = bad || lt.px4[3] == 0.0 || lt.px4[4] == 0.0;
        if (rows & ROW_IMU) bad = bad || (lt.imu[3] * lt.imu[6] - lt.imu[4] * lt.imu[5]) == 0.0 || lt.imu[7] == 0.0;
        if (rows & ROW_MAG) bad = bad || lt.mag[1] == 0.0;
        if (bad) o.flags |= ST_UPDATE_SKIPPED;
    }
    /* no try/catch in this filter: the reference node aborts; here the predicted covariance is kept */
    if (o.flags & ST_UPDATE_SKIPPED) return ST_UPDATE_SKIPPED;
    if (SENSORS) { /* predicted covariance parked in `spill`, tg.P becomes the working copy */
        KFPOS_UNROLL
        for (int k = 0; k < 36; ++k) spill.base[k * spill.stride] = tg.P.a[k];
        iekf8<true>(xhat, tg.z, spill, tg.P, sc, pr, rows, lt, dt, o);
    } else {
        iekf8<false>(xhat, tg.z, tg.P, tg.P, sc, pr, rows, lt, dt, o);
        cov_update8(tg.P, o.mlast);
    }
    tg.xy[0] = o.x[0]; tg.xy[1] = o.x[1];
    tg.vel[0] = o.x[2]; tg.vel[1] = o.x[3];
    tg.ang = o.x[6]; tg.om = o.x[7]; /* :316-319 */
    return pack_status(o.flags, o.gain_iters, o.ml_iters, -1);
}

/* getPose (KalmanFilter.cpp:709-745): predicted state and covariance; pos = (x, y, mUWBtagZ) and the position
 * block of stateToPose's 6x6 (0.01 on the z diagonal, :349) */
KFPOS_FN void pose8(const Tag8 &tg, double t, double accel_noise, double jolt, double x[8], Cov<8, true> &Pp) {
    x[0] = tg.xy[0] + t * tg.vel[0]; x[1] = tg.xy[1] + t * tg.vel[1];
    x[2] = tg.vel[0]; x[3] = tg.vel[1]; x[4] = 0.0; x[5] = 0.0;
    x[6] = normalize_angle(tg.ang + t * tg.om); x[7] = tg.om;
    Pp = tg.P;
    predict8(Pp, t, accel_noise, jolt);
}

} // namespace kfpos
#endif

/*
 * kfpos_core_planar.h -- part of kfpos_core.h (include that, not this): per-tag arithmetic shared by the HIP kernels and
 * the host emulation of the tests.
 * 8-state planar filter: KalmanFilter (KalmanFilter.cpp:224-321, 365-501) with ranging, PX4Flow, IMU and magnetometer /
 * compass rows; MLLocation::estimatePosition2D (MLLocation.cpp:48-143).
 */
#ifndef KFPOS_CORE_PLANAR_H
#define KFPOS_CORE_PLANAR_H

namespace kfpos {

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
    while (rel_change_above(cost, newCost, 1e-3) && (iter < 10000)) { /* MLLocation.cpp:79 */
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
        if (rel_change_below(cost, c, 1e-4)) break; /* :473 */
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

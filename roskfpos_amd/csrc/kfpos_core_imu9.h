/*
 * kfpos_core_imu9.h -- part of kfpos_core.h (include that, not this): per-tag arithmetic shared by the HIP kernels and
 * the host emulation of the tests.
 * 9-state filter: KalmanFilterTOAIMU (KalmanFilterTOAIMU.cpp:100-195, 242-340, with the 3-token repair).
 */
#ifndef KFPOS_CORE_IMU9_H
#define KFPOS_CORE_IMU9_H

namespace kfpos {

/* ================================================================== 9-state filter (KalmanFilterTOAIMU) */
/* smallest LDL' pivot of P_ee, relative to its diagonal entry, for which the information-form iteration is used */
constexpr double INFO_FORM_MIN_PIVOT = 1e-7;

struct Tag9 {
    double pos[3], vel[3];
    Cov<9, true> P;
};
struct Imu {
    bool has;      /* hasImuMeasurement */
    double acc[3]; /* linearAcceleration */
    double *ci;    /* 12 entries, entry k at ci[k * ci_stride]: 0-5 the inverse Cholesky factor of the covariance, lower
                      {00,10,20,11,21,22} (Sigma^-1 = Ci' Ci; used once per step, in the covariance update), 6-11
                      Sigma^-1 itself, symmetric {00,01,02,11,12,22} (six reads per pass of the gain iteration).
                      Constant over a launch, so both wait outside the register file (LDS on the GPU) */
    int ci_stride;
    KFPOS_HD double Wi(int k) const { return ci[(6 + k) * ci_stride]; }
};

/* Sigma (row-major 3x3, symmetric positive definite; the lower triangle is read) -> Ci with
 * Sigma^-1 = Ci' Ci, and Sigma^-1 = Ci' Ci itself */
KFPOS_FN void imu_whitener(const double s[9], double *ci, int ci_stride) {
    double c00, c11, c22, i00, i11, i22;
    kf_sqrt_rsqrt(s[0], c00, i00);
    const double c10 = s[3] * i00, c20 = s[6] * i00;
    kf_sqrt_rsqrt(s[4] - c10 * c10, c11, i11);
    const double c21 = (s[7] - c20 * c10) * i11;
    kf_sqrt_rsqrt(s[8] - c20 * c20 - c21 * c21, c22, i22);
    const double i10 = -c10 * i00 * i11;
    const double i21 = -c21 * i11 * i22;
    const double i20 = -(c20 * i00 + c21 * i10) * i22;
    ci[0] = i00; ci[ci_stride] = i10; ci[2 * ci_stride] = i20; ci[3 * ci_stride] = i11; ci[4 * ci_stride] = i21;
    ci[5 * ci_stride] = i22;
    double *wi = ci + 6 * ci_stride;
    wi[0] = i00 * i00 + i10 * i10 + i20 * i20;
    wi[ci_stride] = i10 * i11 + i20 * i21;
    wi[2 * ci_stride] = i20 * i22;
    wi[3 * ci_stride] = i11 * i11 + i21 * i21;
    wi[4 * ci_stride] = i21 * i22;
    wi[5 * ci_stride] = i22 * i22;
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
        const int cs = imu.ci_stride;
        f.la[0] = acc[0] * imu.ci[0]; f.la[1] = acc[0] * imu.ci[cs]; f.la[2] = acc[0] * imu.ci[2 * cs];
        f.la[3] = acc[1] * imu.ci[3 * cs]; f.la[4] = acc[1] * imu.ci[4 * cs];
        f.la[5] = acc[2] * imu.ci[5 * cs];
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
    double w[6];      /* w of the last gain iteration: the updated state is xhat + P E' w */
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
 * scalings (about 90 of the 725 instructions of an iteration). Same results: the skipped terms are exact zeros.
 * RANGING: the call carries ranging rows (compile-time: the sums of the sweep then need no zero start value and the
 * loop body no branch around it). */
template <bool DIAG, bool RANGING, class SC>
KFPOS_FN void iekf9(const double xhat[9], const Cov<9, true> &P, SC &sc, const Params &pr,
                    const Imu &imu, int max_steps, double tol, Iekf9Out &o) {
    double p[3] = {xhat[0], xhat[1], xhat[2]}, acc[3] = {xhat[6], xhat[7], xhat[8]};
    double ve[6] = {0, 0, 0, 0, 0, 0}; /* P_ee w = x_e - xhat_e = -delta on the updated components (position, acceleration) */
    double wl[6] = {0, 0, 0, 0, 0, 0};
    double qd = 0.0, cost = 1e20;
    const bool pivot = RANGING && illconditioned(sc, pr, P);
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) o.mrlast[k] = 0.0;
    o.dlast[0] = o.dlast[1] = o.dlast[2] = 0.0;
    o.gain_iters = 0;
    for (int iter = 0; iter < max_steps; ++iter) {
        double c = qd, m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, u0 = 0, u1 = 0, u2 = 0;
        if constexpr (RANGING) {
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
        /* u_r = G' R^-1 (y - G delta_p) = G' R^-1 y + M_r (P_ee w)_p: the delta term once per pass */
        const double m[6] = {m0, m1, m2, m3, m4, m5},
                     u[3] = {u0 + (m0 * ve[0] + m1 * ve[1] + m2 * ve[2]), u1 + (m1 * ve[0] + m3 * ve[1] + m4 * ve[2]),
                             u2 + (m2 * ve[0] + m4 * ve[1] + m5 * ve[2])};
        /* IMU rows: y_a = z_a - a, cost += y_a' Sigma^-1 y_a, u_a = D Sigma^-1 (y_a - D delta_a),
         * M_a = D Sigma^-1 D with D = diag(a) (sic, KalmanFilterTOAIMU.cpp:441-473) */
        double ua[3] = {0, 0, 0}, ma[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        if (imu.has) {
            const double ya[3] = {imu.acc[0] - acc[0], imu.acc[1] - acc[1], imu.acc[2] - acc[2]};
            const double va[3] = {ya[0] + acc[0] * ve[3], ya[1] + acc[1] * ve[4], ya[2] + acc[2] * ve[5]};
            if constexpr (DIAG) {
                const double wd[3] = {imu.Wi(0), imu.Wi(3), imu.Wi(5)};
                KFPOS_UNROLL
                for (int i = 0; i < 3; ++i) { /* same operation order as the full path: bit-identical results */
                    const double wy = wd[i] * ya[i];
                    c += ya[i] * wy;
                    ua[i] = acc[i] * (wd[i] * va[i]);
                    ma[i][i] = acc[i] * acc[i] * wd[i];
                }
            } else {
                const double wm[3][3] = {{imu.Wi(0), imu.Wi(1), imu.Wi(2)},
                                         {imu.Wi(1), imu.Wi(3), imu.Wi(4)},
                                         {imu.Wi(2), imu.Wi(4), imu.Wi(5)}};
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
        if (rel_change_below(cost, c, tol)) break; /* KalmanFilterTOAIMU.cpp:316 */
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
            ve[i] = v;
            qd += wl[i] * v;
            if (i < 3) p[i] = xhat[i] + v;
            else acc[i - 3] = xhat[3 + i] + v;
        }
        o.gain_iters++;
    }
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) o.w[k] = wl[k];
    o.cost = cost;
}

/* ---- information-form gain iteration ------------------------------------------------------------------------
 * With B = P_ee (the 6x6 block of the predicted covariance on position and acceleration) invertible, the state shift
 * v = P_ee w of a gain iteration solves the SYMMETRIC POSITIVE DEFINITE system
 *        (B^-1 + M) v = u,          M = blockdiag(M_r, M_a),  u = H' R^-1 (y - H delta)
 * (multiply w = (I + M B)^-1 u by B), and w = u - M v. B is fixed during the iterations of a step, so B^-1 is formed
 * once per step and an iteration costs one 6x6 LDL' solve (~140 instructions) instead of the two 3x3-block
 * eliminations on the non-symmetric I + M B plus the product P_ee w (~360): 700 -> ~510 instructions per iteration.
 * No pivoting is needed (SPD), so the adjugate / pivoted split of the other form does not exist here. B is singular
 * right after a fixed start (P0 = 0, Q has rank 1 per axis) and close to it for a few epochs: sym6_inverse() says
 * so, and such steps take the (I + M B) form below, which needs no inverse of B. */
KFPOS_HD constexpr int s6(int i, int j) { return Cov<6, true>::idx(i, j); }

/* LDL' of a symmetric 6x6 (packed upper): unit lower L (strictly lower entries), pivots d, reciprocals id.
 * Returns false when a pivot is not safely positive: d_j <= rel * a_jj. */
KFPOS_FN bool ldl6(const Cov<6, true> &a, double L[6][6], double id[6], double rel) {
    bool ok = true;
    double d[6];
    KFPOS_UNROLL
    for (int j = 0; j < 6; ++j) {
        double v[6], dj = a(j, j);
        KFPOS_UNROLL
        for (int k = 0; k < j; ++k) { v[k] = L[j][k] * d[k]; dj -= L[j][k] * v[k]; }
        ok = ok && (dj > rel * a(j, j));
        d[j] = dj;
        id[j] = kf_rcp(dj > 0.0 ? dj : 1.0);
        KFPOS_UNROLL
        for (int i = j + 1; i < 6; ++i) {
            double t = a(i, j);
            KFPOS_UNROLL
            for (int k = 0; k < j; ++k) t -= L[i][k] * v[k];
            L[i][j] = t * id[j];
        }
    }
    return ok;
}
/* inverse of a symmetric positive definite 6x6, packed entry k written to out[k * stride] as soon as it is formed (the
 * caller parks it in LDS: holding all 21 in registers next to L and L^-1 is what made this kernel spill); false: not
 * safely invertible (out then holds nothing useful) */
KFPOS_FN bool sym6_inverse(const Cov<6, true> &a, double *out, int stride, double rel) {
    double L[6][6], id[6], W[6][6]; /* W = L^-1, unit lower */
    const bool ok = ldl6(a, L, id, rel);
    KFPOS_UNROLL
    for (int j = 0; j < 6; ++j) {
        KFPOS_UNROLL
        for (int i = j + 1; i < 6; ++i) {
            double t = L[i][j];
            KFPOS_UNROLL
            for (int k = j + 1; k < i; ++k) t += L[i][k] * W[k][j];
            W[i][j] = -t;
        }
    }
    KFPOS_UNROLL
    for (int i = 0; i < 6; ++i) {
        KFPOS_UNROLL
        for (int j = i; j < 6; ++j) { /* sum over k >= j of W(k,i) W(k,j) / d_k, W(k,k) = 1 */
            double t = (i == j ? 1.0 : W[j][i]) * id[j];
            KFPOS_UNROLL
            for (int k = j + 1; k < 6; ++k) t += W[k][i] * W[k][j] * id[k];
            out[s6(i, j) * stride] = t;
        }
    }
    return ok;
}
/* x = K^-1 b for a symmetric positive definite 6x6 K, by 3x3 blocks: K = [Kpp Kpa; Kpa' Kaa], Kpp and the Schur
 * complement Kaa - Kpa' Kpp^-1 Kpa (both SPD) inverted through their cofactors. About as many operations as an LDL'
 * solve, but a dependency chain a third as long (two reciprocals instead of six on it, every cofactor independent of
 * the others) -- with one wavefront per SIMD nothing else hides that latency. */
KFPOS_HD constexpr int s3(int i, int j) { return i <= j ? (i == 0 ? j : (i == 1 ? 2 + j : 5)) : (j == 0 ? i : (j == 1 ? 2 + i : 5)); }
KFPOS_FN void sym6_solve(const Cov<6, true> &K, const double b[6], double x[6]) {
    const double kpp[6] = {K(0, 0), K(0, 1), K(0, 2), K(1, 1), K(1, 2), K(2, 2)};
    double c1[6], c2[6];
    const double id1 = kf_rcp(sym3_cofactors(kpp, c1));
    double X[3][3], y1[3]; /* Kpp^-1 [Kpa | b_p] */
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) {
        y1[i] = (c1[s3(i, 0)] * b[0] + c1[s3(i, 1)] * b[1] + c1[s3(i, 2)] * b[2]) * id1;
        KFPOS_UNROLL
        for (int j = 0; j < 3; ++j)
            X[i][j] = (c1[s3(i, 0)] * K(0, 3 + j) + c1[s3(i, 1)] * K(1, 3 + j) + c1[s3(i, 2)] * K(2, 3 + j)) * id1;
    }
    double S[6], rhs[3];
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) {
        rhs[i] = b[3 + i] - (K(0, 3 + i) * y1[0] + K(1, 3 + i) * y1[1] + K(2, 3 + i) * y1[2]);
        KFPOS_UNROLL
        for (int j = i; j < 3; ++j)
            S[s3(i, j)] = K(3 + i, 3 + j) - (K(0, 3 + i) * X[0][j] + K(1, 3 + i) * X[1][j] + K(2, 3 + i) * X[2][j]);
    }
    const double id2 = kf_rcp(sym3_cofactors(S, c2));
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) x[3 + i] = (c2[s3(i, 0)] * rhs[0] + c2[s3(i, 1)] * rhs[1] + c2[s3(i, 2)] * rhs[2]) * id2;
    KFPOS_UNROLL
    for (int i = 0; i < 3; ++i) x[i] = y1[i] - (X[i][0] * x[3] + X[i][1] * x[4] + X[i][2] * x[5]);
}

/* ---- the information-form iteration, in pieces: sweep (sums over the anchors at the current iterate), pass (cost,
 * convergence test, next iterate). One tag per lane runs sweep + pass in a loop; once at most half of a wavefront's
 * lanes are still iterating, iekf9_pairs hands every survivor to a PAIR of lanes that splits its sweep. */
struct Sweep9 {
    double c, m[6], u[3]; /* sum of w y^2; M = sum of w g g' {00,01,02,11,12,22}; sum of w y g */
};
/* what a lane carries from pass to pass */
struct Iekf9Iter {
    double ve[6]; /* x_e - xhat_e = P_ee w */
    double wl[6]; /* w of the last solve */
    double qd, cost;
};

/* Four anchors, one fused multiply-add chain per sum in anchor order -- written with explicit fma so that the two
 * places that run it (a lane sweeping all 8 anchors of its tag as 4 + 4, a pair of lanes sweeping 4 each) round
 * identically whatever the compiler would have contracted. r, w: ranges and working weights (0 = absent / dropped),
 * b: xyz of the four anchors. */
KFPOS_FN void iekf9_sweep4(const double p[3], const double *r, const double *w, const double *b, Sweep9 &s) {
    KFPOS_UNROLL
    for (int a = 0; a < 4; ++a) {
        const double dx = p[0] - b[3 * a], dy = p[1] - b[3 * a + 1], dz = p[2] - b[3 * a + 2];
        double d, invd;
        kf_sqrt_rsqrt_sweep(kf_fma(dz, dz, kf_fma(dy, dy, dx * dx)), d, invd);
        const double y = r[a] - d, yw = y * w[a];
        const double gx = dx * invd, gy = dy * invd, gz = dz * invd;
        const double wx = w[a] * gx, wy = w[a] * gy, wz = w[a] * gz;
        if (a == 0) {
            s.c = y * yw;
            s.u[0] = gx * yw; s.u[1] = gy * yw; s.u[2] = gz * yw;
            s.m[0] = wx * gx; s.m[1] = wx * gy; s.m[2] = wx * gz;
            s.m[3] = wy * gy; s.m[4] = wy * gz; s.m[5] = wz * gz;
        } else {
            s.c = kf_fma(y, yw, s.c);
            s.u[0] = kf_fma(gx, yw, s.u[0]); s.u[1] = kf_fma(gy, yw, s.u[1]); s.u[2] = kf_fma(gz, yw, s.u[2]);
            s.m[0] = kf_fma(wx, gx, s.m[0]); s.m[1] = kf_fma(wx, gy, s.m[1]); s.m[2] = kf_fma(wx, gz, s.m[2]);
            s.m[3] = kf_fma(wy, gy, s.m[3]); s.m[4] = kf_fma(wy, gz, s.m[4]); s.m[5] = kf_fma(wz, gz, s.m[5]);
        }
    }
}
KFPOS_FN void sweep9_add(Sweep9 &s, const Sweep9 &t) {
    s.c += t.c;
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) s.m[k] += t.m[k];
    KFPOS_UNROLL
    for (int k = 0; k < 3; ++k) s.u[k] += t.u[k];
}

/* the sums of one lane's tag over all its anchors */
template <bool RANGING, class SC>
KFPOS_FN void iekf9_sweep(const double p[3], const SC &sc, const Params &pr, Sweep9 &s) {
    if constexpr (!RANGING) {
        s.c = 0.0;
        KFPOS_UNROLL
        for (int k = 0; k < 6; ++k) s.m[k] = 0.0;
        s.u[0] = s.u[1] = s.u[2] = 0.0;
    } else if constexpr (SC::NA == 8 && SC::CHUNK == 0 && !SC::COOP) {
        /* (anchors 0-3) + (anchors 4-7): the order a pair of lanes can reproduce bit for bit */
        Sweep9 t;
        iekf9_sweep4(p, &sc.r[0], &sc.w[0], &pr.anchors[0], s);
        iekf9_sweep4(p, &sc.r[4], &sc.w[4], &pr.anchors[12], t);
        sweep9_add(s, t);
    } else {
        double c = 0, m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, u0 = 0, u1 = 0, u2 = 0;
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
        s.c = c;
        s.m[0] = m0; s.m[1] = m1; s.m[2] = m2; s.m[3] = m3; s.m[4] = m4; s.m[5] = m5;
        s.u[0] = u0; s.u[1] = u1; s.u[2] = u2;
    }
}

/* One pass at the iterate xhat_e + it.ve, whose anchor sums are sw: cost, convergence test (false: converged, nothing
 * changed) and -- if the iteration goes on -- the solve that yields the next iterate. */
/* What a pass reads from the park: B^-1 and Sigma^-1. Fetched at the top of every trip, BEFORE the sweep, so that the
 * LDS round trip hides behind the sweep's arithmetic instead of stalling the solve (one wavefront per SIMD: nothing else
 * would cover it); 54 registers that are free while the sweep runs. */
/* The pairs' loop reads its parked values in every trip; this asks the scheduler to spread those reads over the sweep,
 * one per ten arithmetic instructions, instead of queueing them in a row at the top of the trip (a wavefront issues in
 * order, and the CU's four wavefronts share one LDS queue). For the one-tag-per-lane loop neither placement beat letting
 * the compiler keep the values in registers. */
KFPOS_FN void iekf9_spread_reads() {
#if defined(__HIP_DEVICE_COMPILE__)
    KFPOS_UNROLL
    for (int k = 0; k < 20; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); /* one DS read */
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0); /* ten VALU */
    }
#endif
}
struct Iekf9Parked {
    double binv[21], wi[6];
};
template <bool EVERY_TRIP>
KFPOS_FN void iekf9_fetch(const double *binv, int binv_stride, const Imu &imu, Iekf9Parked &pk) {
    const int z = EVERY_TRIP ? kf_opaque_zero() : 0; /* (read in every trip / wherever the compiler sees fit) */
    KFPOS_UNROLL
    for (int k = 0; k < 21; ++k) pk.binv[k] = binv[k * binv_stride + z];
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) pk.wi[k] = imu.ci[(6 + k) * imu.ci_stride + z]; /* (never looked at without a sample) */
}
template <bool DIAG>
KFPOS_FN bool iekf9_pass(const double xhat[9], const Iekf9Parked &pk, const Imu &imu, double tol,
                         const Sweep9 &sw, Iekf9Iter &it, Iekf9Out &o) {
    const double *ve = it.ve;
    const double acc[3] = {xhat[6] + ve[3], xhat[7] + ve[4], xhat[8] + ve[5]};
    double c = it.qd + sw.c;
    const double m0 = sw.m[0], m1 = sw.m[1], m2 = sw.m[2], m3 = sw.m[3], m4 = sw.m[4], m5 = sw.m[5];
    const double m[6] = {m0, m1, m2, m3, m4, m5};
    double u[6] = {sw.u[0] + (m0 * ve[0] + m1 * ve[1] + m2 * ve[2]), sw.u[1] + (m1 * ve[0] + m3 * ve[1] + m4 * ve[2]),
                   sw.u[2] + (m2 * ve[0] + m4 * ve[1] + m5 * ve[2]), 0.0, 0.0, 0.0};
    double ma[6] = {0, 0, 0, 0, 0, 0}; /* M_a = D Sigma^-1 D, symmetric {00,01,02,11,12,22} */
    if (imu.has) {
        const double ya[3] = {imu.acc[0] - acc[0], imu.acc[1] - acc[1], imu.acc[2] - acc[2]};
        const double va[3] = {ya[0] + acc[0] * ve[3], ya[1] + acc[1] * ve[4], ya[2] + acc[2] * ve[5]};
        if constexpr (DIAG) {
            const double wd[3] = {pk.wi[0], pk.wi[3], pk.wi[5]};
            KFPOS_UNROLL
            for (int i = 0; i < 3; ++i) {
                const double wy = wd[i] * ya[i];
                c += ya[i] * wy;
                u[3 + i] = acc[i] * (wd[i] * va[i]);
            }
            ma[0] = acc[0] * acc[0] * wd[0]; ma[3] = acc[1] * acc[1] * wd[1]; ma[5] = acc[2] * acc[2] * wd[2];
        } else {
            const double wm[3][3] = {{pk.wi[0], pk.wi[1], pk.wi[2]},
                                     {pk.wi[1], pk.wi[3], pk.wi[4]},
                                     {pk.wi[2], pk.wi[4], pk.wi[5]}};
            KFPOS_UNROLL
            for (int i = 0; i < 3; ++i) {
                const double wy = wm[i][0] * ya[0] + wm[i][1] * ya[1] + wm[i][2] * ya[2];
                c += ya[i] * wy;
                u[3 + i] = acc[i] * (wm[i][0] * va[0] + wm[i][1] * va[1] + wm[i][2] * va[2]);
            }
            ma[0] = acc[0] * acc[0] * wm[0][0]; ma[1] = acc[0] * acc[1] * wm[0][1]; ma[2] = acc[0] * acc[2] * wm[0][2];
            ma[3] = acc[1] * acc[1] * wm[1][1]; ma[4] = acc[1] * acc[2] * wm[1][2]; ma[5] = acc[2] * acc[2] * wm[2][2];
        }
    }
    if (rel_change_below(it.cost, c, tol)) return false; /* KalmanFilterTOAIMU.cpp:316 */
#ifdef KFPOS_EMU_ITER_TRACE /* tools/exp/iter_cycle.py: a host build that records every iterate */
    kfpos_emu_iter_trace(o.gain_iters, ve, c);
#endif
    it.cost = c;
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) o.mrlast[k] = m[k];
    o.dlast[0] = acc[0]; o.dlast[1] = acc[1]; o.dlast[2] = acc[2];

    /* (B^-1 + M) v = u */
    Cov<6, true> K;
    KFPOS_UNROLL
    for (int k = 0; k < 21; ++k) K.a[k] = pk.binv[k];
    K(0, 0) += m[0]; K(0, 1) += m[1]; K(0, 2) += m[2]; K(1, 1) += m[3]; K(1, 2) += m[4]; K(2, 2) += m[5];
    K(3, 3) += ma[0]; K(4, 4) += ma[3]; K(5, 5) += ma[5];
    if constexpr (!DIAG) { K(3, 4) += ma[1]; K(3, 5) += ma[2]; K(4, 5) += ma[4]; }
    double v[6];
    sym6_solve(K, u, v);
    /* w = u - M v ; delta' pinv(P) delta = w . P_ee w = w . v */
    double *wl = it.wl;
    wl[0] = u[0] - (m[0] * v[0] + m[1] * v[1] + m[2] * v[2]);
    wl[1] = u[1] - (m[1] * v[0] + m[3] * v[1] + m[4] * v[2]);
    wl[2] = u[2] - (m[2] * v[0] + m[4] * v[1] + m[5] * v[2]);
    /* (written out in full for the diagonal case too: a shorter form would round differently, and a tag must not
     * see which wave-mates it has) */
    wl[3] = u[3] - (ma[0] * v[3] + ma[1] * v[4] + ma[2] * v[5]);
    wl[4] = u[4] - (ma[1] * v[3] + ma[3] * v[4] + ma[4] * v[5]);
    wl[5] = u[5] - (ma[2] * v[3] + ma[4] * v[4] + ma[5] * v[5]);
    double qd = 0.0;
    KFPOS_UNROLL
    for (int i = 0; i < 6; ++i) qd += wl[i] * v[i];
    it.qd = qd;
    KFPOS_UNROLL
    for (int i = 0; i < 6; ++i) it.ve[i] = v[i];
    o.gain_iters++;
    return true;
}

#if defined(__HIP_DEVICE_COMPILE__)
/* value of v on lane src (byte address src * 4), for every lane of a fully active wavefront */
KFPOS_FN double lane_pull(double v, int src4) {
    const int lo = __builtin_amdgcn_ds_bpermute(src4, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src4, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
/* position of the n-th (0-based) set bit of m; n < popcount(m) */
KFPOS_FN int nth_set_bit(uint64_t m, int n) {
    uint32_t w = (uint32_t)m;
    int base = 0;
    const int c0 = __popc(w);
    if (n >= c0) { n -= c0; w = (uint32_t)(m >> 32); base = 32; }
    KFPOS_UNROLL
    for (int width = 16; width >= 1; width >>= 1) {
        const uint32_t low = w & ((1u << width) - 1u);
        const int c = __popc(low);
        if (n >= c) { n -= c; w >>= width; base += width; }
        else w = low;
    }
    return base;
}

/* The tail of the iteration, two lanes per tag. m: the lanes still iterating (at most 32 of a wavefront that entered
 * the iteration with all 64 lanes); pair g = lanes 2g, 2g+1 takes over the g-th of them: its iterate, ranges and
 * weights come over by ds_bpermute, its B^-1 is read where its owner parked it, the anchors from a copy of the anchor
 * table in LDS (anchor_tab: [8][3], selected by lane parity). Each lane of a pair sweeps four of the eight anchors, one exchange (quad_perm [1,0,3,2]) completes the ten sums -- (0-3) + (4-7), exactly what the
 * owner would have formed -- and both lanes run the pass on identical numbers. Results go back to the owners, which
 * cannot tell that it happened: same bits. (The sweep is written with explicit fma for that; the pass is the same
 * inlined function at both places, and that the compiler contracts it the same way twice is what
 * tests/test_pairs9_gpu.py checks on every build -- it is not a guarantee of the language.) */
template <bool DIAG, bool ACC0>
KFPOS_FN void iekf9_pairs(uint64_t m, const double xhat[9], const double *binv, int binv_stride,
                          const RegScratch<8> &sc, const double *anchor_tab, const Imu &imu, int max_steps, double tol,
                          Iekf9Iter &it, Iekf9Out &o) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int g = lane >> 1, h = lane & 1;
    const int n_surv = __popcll(m);
    bool act = g < n_surv;
    const int src = nth_set_bit(m, act ? g : 0);
    const int src4 = src << 2;
    double xh[9];
    KFPOS_UNROLL
    for (int k = 0; k < 3; ++k) {
        xh[k] = lane_pull(xhat[k], src4);
        xh[3 + k] = 0.0; /* (the velocity prediction plays no role in the iteration) */
        xh[6 + k] = ACC0 ? 0.0 : lane_pull(xhat[6 + k], src4);
    }
    Iekf9Iter ct;
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) { ct.ve[k] = lane_pull(it.ve[k], src4); ct.wl[k] = lane_pull(it.wl[k], src4); }
    ct.qd = lane_pull(it.qd, src4);
    ct.cost = lane_pull(it.cost, src4);
    Iekf9Out co;
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) co.mrlast[k] = lane_pull(o.mrlast[k], src4);
    KFPOS_UNROLL
    for (int k = 0; k < 3; ++k) co.dlast[k] = lane_pull(o.dlast[k], src4);
    co.gain_iters = __builtin_amdgcn_ds_bpermute(src4, o.gain_iters);
    Imu ci;
    ci.has = __builtin_amdgcn_ds_bpermute(src4, imu.has ? 1 : 0) != 0;
    KFPOS_UNROLL
    for (int k = 0; k < 3; ++k) ci.acc[k] = lane_pull(imu.acc[k], src4);
    ci.ci = imu.ci + (src - lane); /* the owner's column of the park */
    ci.ci_stride = imu.ci_stride;
    double r4[4], w4[4];
    KFPOS_UNROLL
    for (int k = 0; k < 4; ++k) {
        const double r0 = lane_pull(sc.r[k], src4), r1 = lane_pull(sc.r[4 + k], src4);
        const double w0 = lane_pull(sc.w[k], src4), w1 = lane_pull(sc.w[4 + k], src4);
        r4[k] = h ? r1 : r0;
        w4[k] = h ? w1 : w0;
    }
    const double *cbinv = binv + (src - lane); /* the owner's column of the park */
    const double *b4 = anchor_tab + 12 * h; /* this lane's four anchors, xyz */
    const double b0[3] = {b4[0], b4[1], b4[2]};
    while (act) { /* (both lanes of a pair leave together) */
        /* the first anchor stays in registers, the other three are read in every trip (held across the solve they
         * would cost 18 more registers), ahead of B^-1: LDS answers in order; the sweep starts on the first anchor
         * while the rest arrives */
        double bb[12];
        const int z = kf_opaque_zero();
        bb[0] = b0[0]; bb[1] = b0[1]; bb[2] = b0[2];
        KFPOS_UNROLL
        for (int k = 3; k < 12; ++k) bb[k] = b4[k + z];
        Iekf9Parked pk;
        iekf9_fetch<true>(cbinv, binv_stride, ci, pk);
        const double p[3] = {xh[0] + ct.ve[0], xh[1] + ct.ve[1], xh[2] + ct.ve[2]};
        Sweep9 sw, other;
        iekf9_sweep4(p, r4, w4, bb, sw);
        other.c = dpp_exchange(sw.c, 0);
        KFPOS_UNROLL
        for (int k = 0; k < 6; ++k) other.m[k] = dpp_exchange(sw.m[k], 0);
        KFPOS_UNROLL
        for (int k = 0; k < 3; ++k) other.u[k] = dpp_exchange(sw.u[k], 0);
        /* the lower lane holds (0-3) and adds (4-7); the upper one adds them the other way round: same sums */
        sweep9_add(sw, other);
        iekf9_spread_reads();
        const bool more = iekf9_pass<DIAG>(xh, pk, ci, tol, sw, ct, co);
        act = more && co.gain_iters < max_steps;
    }
    /* back to the owners: the k-th survivor reads lane 2k */
    const bool owner = (m >> lane) & 1ull;
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    const int from4 = (owner ? 2 * rank : lane) << 2;
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) {
        const double a = lane_pull(ct.wl[k], from4), b = lane_pull(co.mrlast[k], from4);
        if (owner) { it.wl[k] = a; o.mrlast[k] = b; }
    }
    KFPOS_UNROLL
    for (int k = 0; k < 3; ++k) {
        const double a = lane_pull(co.dlast[k], from4);
        if (owner) o.dlast[k] = a;
    }
    const double cst = lane_pull(ct.cost, from4);
    const int gi = __builtin_amdgcn_ds_bpermute(from4, co.gain_iters);
    if (owner) { it.cost = cst; o.gain_iters = gi; }
}
#endif

/* ACC0: the predicted acceleration xhat[6..8] is the literal 0 on every lane (KalmanFilterTOAIMU.cpp restarts it at
 * every step), so a pair does not fetch it */
template <bool DIAG, bool RANGING, bool ACC0 = false, class SC>
KFPOS_FN void iekf9_info(const double xhat[9], const double *binv, int binv_stride, SC &sc, const Params &pr,
                         const Imu &imu, int max_steps, double tol, Iekf9Out &o) {
    Iekf9Iter it;
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) { it.ve[k] = 0.0; it.wl[k] = 0.0; o.mrlast[k] = 0.0; }
    it.qd = 0.0;
    it.cost = 1e20;
    o.dlast[0] = o.dlast[1] = o.dlast[2] = 0.0;
    o.gain_iters = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool PAIRS = RANGING && SC::NA == 8 && SC::CHUNK == 0 && !SC::COOP;
    /* every lane of the wavefront is here (none left early, none took the other form): pairs can be formed */
    const bool pairs = PAIRS && pr.pair_anchor_tab && __builtin_amdgcn_ballot_w64(true) == ~0ull;
#endif
    bool more = true; /* false: converged */
    int stop = max_steps;
#if defined(__HIP_DEVICE_COMPILE__)
    /* with pairs the lanes meet again after 8 solves, then after every 4: by then (BASELINE configs[2] trace: after 8
     * in 97 % of the wavefronts) at most half of them are still iterating. Looking after every pass would cost the
     * per-lane loop more than the pairs give back (measured: 1.2 us per epoch for a ballot and a branch per trip). */
    if (pairs) stop = 8 < max_steps ? 8 : max_steps;
#endif
    for (;;) {
        while (more && o.gain_iters < stop) {
            Iekf9Parked pk;
            iekf9_fetch<false>(binv, binv_stride, imu, pk);
            const double p[3] = {xhat[0] + it.ve[0], xhat[1] + it.ve[1], xhat[2] + it.ve[2]};
            Sweep9 sw;
            iekf9_sweep<RANGING>(p, sc, pr, sw);
            more = iekf9_pass<DIAG>(xhat, pk, imu, tol, sw, it, o);
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (PAIRS) {
            if (pairs) { /* (wave-uniform: every lane is back here) */
                const uint64_t m = __builtin_amdgcn_ballot_w64(more && o.gain_iters < max_steps);
                if (m == 0) break;
                if (__popcll(m) <= 32) {
                    iekf9_pairs<DIAG, ACC0>(m, xhat, binv, binv_stride, sc, pr.pair_anchor_tab, imu, max_steps, tol, it, o);
                    break;
                }
                stop = stop + 4 < max_steps ? stop + 4 : max_steps;
                continue;
            }
        }
#endif
        break;
    }
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) o.w[k] = it.wl[k];
    o.cost = it.cost;
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
 * RANGING = false is the IMU-only call of newIMUMeasurement (:91): its own instantiation (and kernel), so that a
 * ranging epoch carries no code of it. */
/* where the 45 covariance entries wait while the information-form iteration runs (it does not touch P), followed by
 * the 21 entries of B^-1, which each iteration reads back, and the 6 of the accelerometer whitener (Imu::ci): element k
 * at a[k * stride] -- LDS on the GPU ([78][lane] with Imu::ci behind them: 39 KB per wavefront), a stack array in the host emulation. The iteration then has the directly addressable half of
 * the register file to itself instead of shuffling P and B^-1 through the accumulation registers. */
struct CovPark9 {
    double *a;
    int stride;
};

template <bool RANGING, class SC>
KFPOS_FN uint32_t step_imu9(Tag9 &tg, SC &sc, const Params &pr, double dt, const Imu &imu, const CovPark9 &park) {
    constexpr bool has_ranging = RANGING;
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
    /* Prediction of the covariance, then B = P_ee and its inverse, BEFORE the ML solve: fewer values are alive here
     * than behind it, and from here to the end of the iteration P waits in LDS (inversion and iteration work on B
     * alone). In multi-epoch launches P is in registers at this point anyway; a single-epoch launch waits for its
     * covariance loads here instead of behind the ML solve. */
    predict9(tg.P, dt, pr.jolt);
    Cov<6, true> B;
    KFPOS_UNROLL
    for (int i = 0; i < 6; ++i) {
        KFPOS_UNROLL
        for (int j = i; j < 6; ++j) B(i, j) = tg.P(e9(i), e9(j));
    }
    KFPOS_UNROLL
    for (int k = 0; k < 45; ++k) park.a[k * park.stride] = tg.P.a[k];
    double *binv = park.a + 45 * park.stride;
    /* B safely invertible on every lane: the information-form iteration (the usual case); otherwise -- right after a
     * fixed start, B is singular -- the (I + M B) form */
    const bool invertible = sym6_inverse(B, binv, park.stride, INFO_FORM_MIN_PIVOT);
    Iekf9Out o;
    iekf9_weights(xhat, sc, pr, has_ranging, n_valid, o); /* position + epoch only */
    /* The 9-state filter has no try/catch: the reference node aborts here. This core keeps the predicted
     * covariance and reports the tag instead. */
    if (o.flags & ST_UPDATE_SKIPPED) {
        KFPOS_UNROLL
        for (int k = 0; k < 45; ++k) tg.P.a[k] = park.a[k * park.stride];
        return ST_UPDATE_SKIPPED;
    }
    /* per lane, not per wavefront: a tag's arithmetic must not depend on its wave-mates (a wavefront whose lanes
     * disagree runs both forms one after the other, each under its lanes' mask) */
    if (invertible) {
        iekf9_info<false, RANGING, true>(xhat, binv, park.stride, sc, pr, imu, 20, 1e-4, o);
        KFPOS_UNROLL
        for (int k = 0; k < 45; ++k) tg.P.a[k] = park.a[k * park.stride];
    } else {
        KFPOS_UNROLL
        for (int k = 0; k < 45; ++k) tg.P.a[k] = park.a[k * park.stride];
        iekf9<false, RANGING>(xhat, tg.P, sc, pr, imu, 20, 1e-4, o);
    }
    KFPOS_UNROLL
    for (int i = 0; i < 6; ++i) { /* x = xhat + P E' w: position and velocity are what the filter keeps (:189-194) */
        double v = 0.0;
        KFPOS_UNROLL
        for (int k = 0; k < 6; ++k) v += tg.P(i, e9(k)) * o.w[k];
        if (i < 3) tg.pos[i] = xhat[i] + v;
        else tg.vel[i - 3] = xhat[i] + v;
    }
    cov_update9(tg.P, o.mrlast, o.dlast, imu);
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

} // namespace kfpos
#endif

/*
 * kfpos_core_toa6.h -- part of kfpos_core.h (include that, not this): per-tag arithmetic shared by the HIP kernels and
 * the host emulation of the tests.
 * 6-state filter: KalmanFilterTOA (KalmanFilterTOA.cpp:70-156, 185-338), incl. the SVD pseudo-inverse route of the
 * non-symmetric layout and the leave-one-out heuristic.
 */
#ifndef KFPOS_CORE_TOA6_H
#define KFPOS_CORE_TOA6_H

namespace kfpos {

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
    int pivot;        /* this step's 3x3 systems go through gauss_solve (illconditioned()); an int: a trailing bool leaves padding bytes that the struct copies of the leave-one-out loop would move through scratch memory */
};

/* First half of kalmanStep3DIgnoreAnchor (KalmanFilterTOA.cpp:268-282): ML position -> observation
 * covariance. Touches the position and the epoch only, not P, so the kernels run it while the covariance
 * loads are still in flight. Leaves sc.w = 1/R. */
template <class SC>
KFPOS_FN void iekf6_weights(const double xhat_p[3], SC &sc, const Params &pr, uint64_t drop, int n_used,
                            Iekf6Out &o, bool use_first, const MlFirst &first, bool keep_first, MlFirst &first_out,
                            bool weights_ready = false) {
    o.flags = (n_used < 4) ? ST_FEW_RANGES : 0u;
    double pml[3] = {xhat_p[0], xhat_p[1], xhat_p[2]}, e_ml;
    if (!weights_ready) set_weights_ml(sc, pr, drop); /* (the top-N ranking may have left them in place) */
    o.ml_iters = ml_estimate(pml, sc, pr, drop, n_used, e_ml, use_first, first, keep_first, first_out);
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
        if (rel_change_below(cost, c, tol)) break; /* KalmanFilterTOA.cpp:307 */
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
/* PARKN (symmetric layout only): the first PARKN covariance entries wait at park[k * park_stride] (LDS) while the
 * ML solve runs -- it does not touch P -- so that the solve has their registers: what lets the plain filter fit 256
 * registers, two wavefronts per SIMD (k_step_toa6_w2). */
template <bool SYMM, int HEUR = 2, int PARKN = 0, class SC>
KFPOS_FN uint32_t step_toa6(Tag6<SYMM> &tg, SC &sc, const Params &pr_in, double dt, double *park = nullptr,
                            int park_stride = 0) {
    static_assert(PARKN == 0 || SYMM, "the parking slots of the non-symmetric layout hold its pseudo-inverse");
    Params pr = pr_in;
    if constexpr (PARKN > 0) {
        KFPOS_UNROLL
        for (int k = 0; k < PARKN; ++k) park[k * park_stride] = tg.P.a[k];
    }
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
    /* top-N only (HEUR == 1): the solve that ranks the residuals and the filter's own solve start at the same seed, so
     * the second one gets its first sweep from the first one's (minus the dropped ranges' terms), and the working
     * weights of the kept set straight from the ranking (topn_mask<true>) */
    constexpr bool SHARE_TOPN = HEUR == 1 && SYMM && !SC::COOP;
    MlFirst first_top = {};
    bool top_shared = false, top_weights = false;
    if (pr.top_n > 0 && !(isnan(tg.pos[0]) || isnan(tg.pos[1]) || isnan(tg.pos[2]))) {
        bool first_valid = false;
        drop = topn_mask<SHARE_TOPN>(tg.pos, sc, pr, n_valid, first_top, first_valid, top_weights);
        if (SHARE_TOPN && first_valid) { /* (nothing to drop: no ranking solve has run) */
            uint64_t rest = drop;
            KFPOS_UNROLL
            for (int k = 0; k < 2; ++k) { /* up to two dropped ranges are taken out of the sums; a lane with more sweeps as before */
                const bool any = rest != 0;
                const int v = any ? kf_ctz64(rest) : 0;
                MlFirst t;
                ml_terms_of_lane<SC>(tg.pos, pr, v, sc.Rdyn(v), any ? kf_rcp(sc.Edyn(v)) : 0.0, t);
                KFPOS_UNROLL
                for (int j = 0; j < 3; ++j) first_top.g[j] -= t.g[j];
                KFPOS_UNROLL
                for (int j = 0; j < 6; ++j) first_top.hs[j] -= t.hs[j];
                rest &= rest - 1;
            }
            top_shared = rest == 0; /* more than two dropped: this lane sweeps as before */
        }
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
    MlFirst first_all = {};
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
        /* every solve of the heuristic starts at the same seed: the all-ranges solve (v = -1) leaves the gradient /
         * Hessian sums of its first sweep, a leave-one-out solve subtracts anchor v's terms from them instead of
         * sweeping the other ranges again (one anchor instead of A per solve). Still ONE call site of the solver. */
        MlFirst f = {};
        bool use_first = false, keep_first = false;
        if constexpr (HEUR == 2 && SYMM && !SC::COOP) { /* (the non-symmetric layout's kernels have no register to spare) */
            if (v < 0) {
                keep_first = true;
            } else if (!last) {
                MlFirst t;
                ml_terms_of<SC>(xhat_p, pr, v, ra, kf_rcp(sc.Edyn(v)), t);
                KFPOS_UNROLL
                for (int k = 0; k < 3; ++k) f.g[k] = first_all.g[k] - t.g[k];
                KFPOS_UNROLL
                for (int k = 0; k < 6; ++k) f.hs[k] = first_all.hs[k] - t.hs[k];
                use_first = true;
            }
        }
        if constexpr (SHARE_TOPN) {
            if (last && top_shared) {
                f = first_top;
                use_first = true;
            }
        }
        iekf6_weights(xhat_p, sc, pr, mask, n_use, o, use_first, f, keep_first, first_all,
                      SHARE_TOPN && last && top_weights);
        if (!predicted) { /* after the first ML solve: the covariance loads have landed by now */
            if constexpr (PARKN > 0) {
                KFPOS_UNROLL
                for (int k = 0; k < PARKN; ++k) tg.P.a[k] = park[k * park_stride];
            }
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

} // namespace kfpos
#endif

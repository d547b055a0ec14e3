/*
 * kfpos_core_ml.h -- part of kfpos_core.h (include that, not this): per-tag arithmetic shared by the HIP kernels and
 * the host emulation of the tests.
 * Gauss-Newton multilateration (MLLocation::estimatePosition, MLLocation.cpp:153-257), its covariance, the top-N
 * composition and the standalone ML estimator (ALGORITHM_ML).
 */
#ifndef KFPOS_CORE_ML_H
#define KFPOS_CORE_ML_H

namespace kfpos {

/* ------------------------------------------------------------------ MLLocation::estimatePosition */
/* One sweep over the anchors at position p: weighted cost sum (r-d)^2/e, unweighted SSE
 * (estimationError, MLLocation.cpp:263-278), gradient g and Hessian-like Hs of
 * MLLocation.cpp:174-204 (Hs symmetric, packed {00,01,02,11,12,22}). */
template <class SC>
KFPOS_FN void ml_sweep(const double p[3], const SC &sc, const Params &pr, uint64_t drop,
                              double &cw, double &sse, double g[3], double hs[6]) {
    double cw_ = 0.0, sse_ = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0, c0s = 0.0;
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
        const double rd = r - d, rd2 = rd * rd;
        cw_ += rd2 * w;
        sse_ += on ? rd2 : 0.0;
        /* with wi = w / d:  gradient weight (r - d) w / d,  Hessian terms w (1 - r/d) on the diagonal (summed once:
         * the same number goes to all three entries) and w r / d^3 on the dyadic part */
        const double wi = w * invd, gi = rd * wi, wq = r * wi;
        g0 += gi * dx;
        g1 += gi * dy;
        g2 += gi * dz;
        c0s += w - wq;
        const double c1 = wq * (invd * invd);
        const double tx = c1 * dx, ty = c1 * dy, tz = c1 * dz;
        h0 += tx * dx;
        h1 += tx * dy;
        h2 += tx * dz;
        h3 += ty * dy;
        h4 += ty * dz;
        h5 += tz * dz;
    });
    /* 8 lanes per tag: the SSE is wanted of the LAST sweep only, so it stays a per-lane partial sum here and the caller
     * combines it once, behind the loop (three DPP exchanges less in every pass) */
    cw = group_sum(sc, cw_); sse = SC::COOP ? sse_ : group_sum(sc, sse_);
    g[0] = group_sum(sc, g0); g[1] = group_sum(sc, g1); g[2] = group_sum(sc, g2);
    /* (the diagonal term joins the partial sums before they are combined: no extra exchange in the 8-lanes-per-tag mode) */
    hs[0] = group_sum(sc, h0 + c0s); hs[1] = group_sum(sc, h1); hs[2] = group_sum(sc, h2);
    hs[3] = group_sum(sc, h3 + c0s); hs[4] = group_sum(sc, h4); hs[5] = group_sum(sc, h5 + c0s);
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

/* Gradient and Hessian sums of the FIRST sweep of a solve (its cost and SSE are never looked at: the loop's first
 * comparison is against the constant 1). The leave-one-out heuristic runs up to 65 solves from the same seed over
 * subsets that differ by one range: their first sweeps are the sweep over all ranges minus one anchor's terms. */
struct MlFirst {
    double g[3], hs[6];
};
/* the terms anchor v (wave-uniform index) contributes to a sweep at p with weight w (0: none) and range r */
template <class SC>
KFPOS_FN void ml_terms_of(const double p[3], const Params &pr, int v, double r, double w, MlFirst &t) {
    const double dx = pr.anchors[3 * v] - p[0], dy = pr.anchors[3 * v + 1] - p[1], dz = pr.anchors[3 * v + 2] - p[2];
    double d, invd;
    kf_sqrt_rsqrt_sweep(dx * dx + dy * dy + dz * dz, d, invd);
    const double rd = r - d;
    const double wi = w * invd, gi = rd * wi, wq = r * wi;
    const double c0 = w - wq, c1 = wq * (invd * invd);
    const double tx = c1 * dx, ty = c1 * dy, tz = c1 * dz;
    t.g[0] = gi * dx; t.g[1] = gi * dy; t.g[2] = gi * dz;
    t.hs[0] = tx * dx + c0; t.hs[1] = tx * dy; t.hs[2] = tx * dz;
    t.hs[3] = ty * dy + c0; t.hs[4] = ty * dz; t.hs[5] = tz * dz + c0;
}

/* the same for an anchor index that differs from lane to lane (the ranges a top-N composition dropped): the anchor
 * table is read with a vector load. v must be a valid index on every lane; w = 0 yields exact zeros. */
template <class SC>
KFPOS_FN void ml_terms_of_lane(const double p[3], const Params &pr, int v, double r, double w, MlFirst &t) {
    ml_terms_of<SC>(p, pr, v, r, w, t);
}

/* Gauss-Newton loop of MLLocation.cpp:164-225. p: seed in, estimate out. Requires sc.w = 1/e.
 * Returns the iteration count; sse_out = estimationError at the result. With n_used < 4 the
 * seed is returned untouched (MLLocation.cpp:158-161). The step p - Hs^-1 g equals the
 * reference's solve(Hs, Hs p - g). One sweep per pass yields the cost of the point just reached
 * and the gradient/Hessian for the next step (the reference evaluates them in two passes). */
/* The leave-one-out heuristic hands in the gradient / Hessian sums of the first sweep (use_first, first) or asks for
 * them (keep_first, first_out); flags and references rather than pointers, so that the sums stay in registers. */
template <class SC>
KFPOS_FN int ml_estimate(double p[3], const SC &sc, const Params &pr, uint64_t drop, int n_used, double &sse_out,
                         bool use_first, const MlFirst &first, bool keep_first, MlFirst &first_out) {
    if (n_used < 4) {
        sse_out = (n_used == 0) ? -1.0 : ml_sse(p, sc, pr, drop);
        return 0;
    }
    double cost = 1e20, newCost = 1.0, cw = 0.0, sse = 0.0, g[3], hs[6], c[6];
    int iter = 0;
    for (;;) {
        if (use_first && iter == 0) {
            KFPOS_UNROLL
            for (int k = 0; k < 3; ++k) g[k] = first.g[k];
            KFPOS_UNROLL
            for (int k = 0; k < 6; ++k) hs[k] = first.hs[k];
        } else {
            ml_sweep(p, sc, pr, drop, cw, sse, g, hs);
        }
        if (keep_first && iter == 0) {
            KFPOS_UNROLL
            for (int k = 0; k < 3; ++k) first_out.g[k] = g[k];
            KFPOS_UNROLL
            for (int k = 0; k < 6; ++k) first_out.hs[k] = hs[k];
        }
        if (iter > 0) newCost = cw;
        /* MLLocation.cpp:168. One tag per lane: the product form, no IEEE division per pass (-54 instructions per
         * wave-epoch, 1-2.4 % of a 6-state step). 8 lanes per tag: that kernel is one dependent chain at 75 % issue
         * occupancy, and there the vote and the extra branch of the product form cost more than the division they
         * replace (same box: 4.38 -> 4.55 us per epoch, profiles/r03b_*): it keeps the quotient. */
        bool go_on;
        if constexpr (SC::COOP) go_on = fabs(cost - newCost) / cost > 1e-3;
        else go_on = rel_change_above(cost, newCost, 1e-3);
        if (!(go_on && (iter < 10000))) break;
        iter += 1;
        cost = newCost;
        const double idet = kf_rcp(sym3_cofactors(hs, c));
        p[0] -= (c[0] * g[0] + c[1] * g[1] + c[2] * g[2]) * idet;
        p[1] -= (c[1] * g[0] + c[3] * g[1] + c[4] * g[2]) * idet;
        p[2] -= (c[2] * g[0] + c[4] * g[1] + c[5] * g[2]) * idet;
    }
    sse_out = group_sum(sc, sse); /* (identity unless 8 lanes share the tag: ml_sweep) */
    return iter;
}

template <class SC>
KFPOS_FN int ml_estimate(double p[3], const SC &sc, const Params &pr, uint64_t drop, int n_used, double &sse_out) {
    MlFirst unused = {};
    return ml_estimate(p, sc, pr, drop, n_used, sse_out, false, unused, false, unused);
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
/* REQUIRES the weights of set_weights_ml(sc, pr, drop) in place: 1 / max(e_ML, e_a) is either that weight (1 / e_a,
 * the same kf_rcp of the same number) or 1 / e_ML, one reciprocal for all anchors instead of one per anchor */
template <class SC>
KFPOS_FN void set_weights_iekf(SC &sc, const Params &pr, double e_ml, uint64_t drop) {
    if constexpr (SC::COOP) { /* one anchor per lane: one reciprocal either way, the direct form is shorter */
        sc.setW(0, used(sc, 0, drop) ? kf_rcp(stdmax(e_ml, sc.E(0))) : 0.0);
        return;
    }
    const double r_ml = kf_rcp(e_ml);
    for_anchors<SC>(pr, [&](int a) { /* KalmanFilterTOA.cpp:281; stdmax(e_ml, e) = e_ml < e ? e : e_ml */
        sc.setW(a, used(sc, a, drop) ? ((e_ml < sc.E(a)) ? sc.W(a) : r_ml) : 0.0);
    });
}

/* Top-N composition (BASELINE config 5; MLLocation.cpp:284-300, 325-339): rank the residual^2
 * at the ML position of all ranges, drop the min(n-4, N) largest. Returns the drop mask.
 * SHARE (the top-N-only kernels): the caller's next solve starts at the same seed over the kept ranges, so
 *  - first_all receives the gradient / Hessian sums of this solve's first sweep (the caller subtracts the dropped
 *    anchors' terms instead of sweeping again), and
 *  - with at most two ranges to drop the two largest residuals are tracked in registers while they are computed, the
 *    working weights (1/e of every used range) stay where they are and only the dropped ones are zeroed: on return
 *    weights_kept says that the weights of the kept set are already in place (set_weights_ml(sc, pr, drop) done). */
template <bool SHARE, class SC>
KFPOS_FN uint64_t topn_mask(const double seed[3], SC &sc, const Params &pr, int n_valid, MlFirst &first_all,
                            bool &first_valid, bool &weights_kept, int *rank_iters = nullptr) {
    weights_kept = false;
    first_valid = false;
    if (rank_iters) *rank_iters = 0;
    int ndrop = n_valid - 4 < pr.top_n ? n_valid - 4 : pr.top_n;
    if (ndrop <= 0) return 0;
    first_valid = SHARE; /* n_valid >= 5 here: the solve below does sweep */
    double p[3] = {seed[0], seed[1], seed[2]}, sse;
    set_weights_ml(sc, pr, 0ull);
    MlFirst unused = {};
    const int it_rank = ml_estimate(p, sc, pr, 0, n_valid, sse, false, unused, SHARE, first_all);
    if (rank_iters) *rank_iters = it_rank;
    uint64_t drop = 0;
    if (SHARE && ndrop <= 2) {
        double v1 = -1.0, v2 = -1.0; /* largest, second largest residual^2; an absent range (-1) never gets in */
        int i1 = -1, i2 = -1;
        for_anchors<SC>(pr, [&](int a) {
            const double dx = pr.anchors[3 * a] - p[0], dy = pr.anchors[3 * a + 1] - p[1], dz = pr.anchors[3 * a + 2] - p[2];
            double d, invd;
            kf_sqrt_rsqrt(dx * dx + dy * dy + dz * dz, d, invd);
            const double rd = d - sc.R(a);
            const double r2 = used(sc, a, 0) ? rd * rd : -1.0;
            /* the order of the selection passes below: the first of equal values wins, the next one comes second */
            const bool first = r2 > v1, second = !first && r2 > v2;
            v2 = first ? v1 : (second ? r2 : v2);
            i2 = first ? i1 : (second ? a : i2);
            v1 = first ? r2 : v1;
            i1 = first ? a : i1;
        });
        if (i1 >= 0) {
            drop |= 1ull << i1;
            sc.setWdyn(i1, 0.0);
        }
        if (ndrop == 2 && i2 >= 0) {
            drop |= 1ull << i2;
            sc.setWdyn(i2, 0.0);
        }
        weights_kept = true;
        return drop;
    }
    /* residual^2 of every range at the ML position, once, into the working-weight slots (free between the two solves:
     * the caller sets the weights of the kept set afterwards); then ndrop selection passes over those 16 numbers
     * instead of ndrop passes that each recompute all the distances */
    for_anchors<SC>(pr, [&](int a) {
        const double dx = pr.anchors[3 * a] - p[0], dy = pr.anchors[3 * a + 1] - p[1], dz = pr.anchors[3 * a + 2] - p[2];
        double d, invd;
        kf_sqrt_rsqrt(dx * dx + dy * dy + dz * dz, d, invd);
        const double rd = d - sc.R(a);
        sc.setW(a, used(sc, a, 0) ? rd * rd : -1.0); /* an absent range never wins: every real residual^2 is >= 0 */
    });
    for (int k = 0; k < ndrop; ++k) {
        double worst = -1.0;
        int wi = -1;
        for_anchors<SC>(pr, [&](int a) {
            const double r2 = sc.W(a);
            const bool take = !((drop >> a) & 1ull) && (r2 > worst);
            worst = take ? r2 : worst;
            wi = take ? a : wi;
        });
        if (wi < 0) break;
        drop |= 1ull << wi;
    }
    return drop;
}
template <class SC>
KFPOS_FN uint64_t topn_mask(const double seed[3], SC &sc, const Params &pr, int n_valid, int *rank_iters = nullptr) {
    MlFirst unused = {};
    bool valid, kept;
    return topn_mask<false>(seed, sc, pr, n_valid, unused, valid, kept, rank_iters);
}

/* ================================================================== standalone ML estimator (ALGORITHM_ML) */
/* MLLocation::newTOAMeasurement + getPose (MLLocation.cpp:421-486): solve from the fixed seed (_previousEstimation is
 * never updated), return position + 3x3 covariance. Variants (Params::ml_variant, MLLocation.h:5-7):
 *   NORMAL 3-D   estimatePosition (:153-257)
 *   IGNORE_N     estimatePositionIgnoreN (:307-347): drop the min(n-4, N) largest residuals and solve again
 *   BEST         estimatePositionBestGroup (:348-414), where the reference defines it: 4 or 5 ranges (below) */
constexpr int ML_VARIANT_BEST_GROUP = 2;

/* one solve over the ranges not in `drop`, with its covariance (over those ranges only). false: the reference throws
 * (an errorEstimation of exactly 0, or an exactly singular J' W J) */
template <class SC>
KFPOS_FN bool ml_solve_cov(const double seed[3], SC &sc, const Params &pr, uint64_t drop, int n_used, double p[3],
                           double c[6], int &it) {
    p[0] = seed[0]; p[1] = seed[1]; p[2] = seed[2];
    double sse;
    set_weights_ml(sc, pr, drop);
    it = ml_estimate(p, sc, pr, drop, n_used, sse);
    if (ml_covariance_throws(sc, pr, drop, n_used, sse)) return false;
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
    if (det == 0.0) return false; /* inv() of an exactly singular J' W J throws */
    const double idet = 1.0 / det;
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) c[k] = cf[k] * idet;
    return true;
}

/* anchor index of the k-th (0-based) range of this epoch that is present */
template <class SC>
KFPOS_FN int nth_present(const SC &sc, const Params &pr, int k) {
    int seen = 0, at = 0;
    for_anchors<SC>(pr, [&](int a) {
        const bool on = used(sc, a, 0ull);
        at = (on && seen == k) ? a : at;
        seen += on ? 1 : 0;
    });
    return at;
}

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
    double p[3], c[6];
    int it = 0;
    if (pr.ml_variant == ML_VARIANT_BEST_GROUP) {
        /* estimatePositionBestGroup enumerates the subsets of 4 ranges with std::prev_permutation over a mask and
         * removes the unselected ranges with erase(begin() + i) on the vector it is shrinking (:377-381): with two or
         * more ranges to remove the second erase addresses end() or beyond -- undefined from the first group on, so 6
         * ranges and more have no result to restate (kfpos_set_anchors refuses such a table for this variant). With 4
         * ranges there is one group, all of them; with 5 there are five, each without exactly one range, in the order
         * "without range 4, 3, 2, 1, 0" (indices among the ranges present). The smallest covariance trace wins, a
         * later group wins a tie (`<=`, :407), a NaN trace only if it is the first group's. */
        if (n_valid > 5) return ST_UPDATE_SKIPPED;
        /* the solve over ALL ranges the reference runs first and throws away (:357-366) still throws when an
         * errorEstimation is 0 */
        if (ml_covariance_throws(sc, pr, 0, n_valid, NAN)) return ST_UPDATE_SKIPPED;
        const int groups = n_valid == 4 ? 1 : 5;
        double best_trace = 0.0;
        int best_it = 0, best_out = -1, most = 0;
        for (int g = 0; g < groups; ++g) {
            const int out = n_valid == 4 ? -1 : 4 - g;
            const uint64_t drop = out < 0 ? 0ull : (1ull << nth_present(sc, pr, out));
            double gp[3], gc[6];
            int git;
            if (!ml_solve_cov(seed, sc, pr, drop, 4, gp, gc, git)) return ST_UPDATE_SKIPPED;
            const double trace = gc[0] + gc[3] + gc[5];
            most = git > most ? git : most;
            if (g == 0 || trace <= best_trace) {
                best_trace = trace;
                best_it = git;
                best_out = out;
                KFPOS_UNROLL
                for (int k = 0; k < 3; ++k) p[k] = gp[k];
                KFPOS_UNROLL
                for (int k = 0; k < 6; ++k) c[k] = gc[k];
            }
        }
        KFPOS_UNROLL
        for (int k = 0; k < 3; ++k) pos[k] = p[k];
        KFPOS_UNROLL
        for (int k = 0; k < 6; ++k) cov[k] = c[k];
        /* gain-iteration byte: the most Gauss-Newton passes any group took; ML byte: the winner's; ignored: the range
         * the winning group did without (index among this epoch's ranges), -1 with four ranges */
        return pack_status(0, most, best_it, best_out);
    }
    uint64_t drop = 0;
    int it_rank = 0; /* Gauss-Newton passes of the ranking solve: reported in the status word's gain-iteration byte */
    if (pr.top_n > 0) {
        drop = topn_mask(seed, sc, pr, n_valid, &it_rank); /* first solve + ranking */
        /* the first solve throws exactly when a used errorEstimation is 0 (its sse is NaN then) */
        if (ml_covariance_throws(sc, pr, 0, n_valid, NAN)) return ST_UPDATE_SKIPPED;
        n_valid = count_used(sc, pr, drop);
    }
    if (!ml_solve_cov(seed, sc, pr, drop, n_valid, p, c, it)) return ST_UPDATE_SKIPPED;
    KFPOS_UNROLL
    for (int k = 0; k < 3; ++k) pos[k] = p[k];
    KFPOS_UNROLL
    for (int k = 0; k < 6; ++k) cov[k] = c[k];
    return pack_status(0, it_rank, it, -1);
}

} // namespace kfpos
#endif

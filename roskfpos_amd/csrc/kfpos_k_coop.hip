/*
 * kfpos_k_coop.hip -- k_step_toa6_coop: the 6-state step for small banks, one tag per group of 8 lanes.
 *
 * This translation unit alone is compiled with -ffp-contract=fast (hipcc's default: multiply-adds fused across statements
 * too); the rest of the library uses -ffp-contract=on so that every one-tag-per-lane kernel rounds alike (Makefile). This
 * kernel sums in another order anyway (three DPP exchanges per partial sum), is nothing but one dependent chain, and runs
 * 5 % faster with the chain the freer contraction leaves (BASELINE configs[1]: 4.54 -> 4.29 us per epoch).
 */
#include "kfpos_kernels.h"

namespace {

/* ------------------------------------------------------------------ 6-state step kernel, small batches */
/* One tag per group of 8 lanes, one anchor per lane (kfpos_core.h: CoopScratch): for banks of a few thousand tags
 * the chip is mostly empty and a step costs the instruction chain of one lane, so the anchor sweeps are spread
 * over the idle lanes (three DPP exchanges per partial sum) and the chain shrinks ~2.5x. Plain 6-state filter only:
 * fixed start (symmetric layout), no outlier heuristic, at most 8 anchors. Every lane of a group carries the tag's
 * whole state (identical bits); lane 0 of the group writes it back. */

template <typename REAL, typename MREAL>
__global__ __launch_bounds__(WAVE) void k_step_toa6_coop(const KArgs a) {
    __shared__ double s_anchor[COOP_LANES * 3];
    const int lane = threadIdx.x, al = lane & (COOP_LANES - 1);
    if (lane < COOP_LANES * 3) s_anchor[lane] = (lane < a.A * 3) ? a.anchors[lane] : 0.0; /* wave-uniform table -> LDS */
    __syncthreads();
    const size_t t = (size_t)blockIdx.x * COOP_TAGS_PER_WAVE + (lane >> 3);
    if (t >= (size_t)a.T) return; /* whole groups only: the exchanges never cross a group */
    const size_t T = a.T;
    const uint32_t t32 = (uint32_t)t;
    namespace kc = kfpos; /* (this translation unit is compiled with -ffp-contract=fast: Makefile) */
    const kc::Params pr = make_params_of<kc::Params>(a);
    if (a.n_steps == 1 && a.dt && a.dt[t] < 0.0) {
        skipped_lane(a, t, al == 0);
        return;
    }
    const auto fl0 = a.flags[t]; /* (read in front of the state loads, written back at the end: see k_step_toa6) */
    const bool has_anchor = al < a.A;
    kc::CoopScratch sc;
    sc.bx = s_anchor[3 * al]; sc.by = s_anchor[3 * al + 1]; sc.bz = s_anchor[3 * al + 2];
    kc::Tag6<true> tg;
#pragma unroll
    for (int k = 0; k < 3; ++k) tg.pos[k] = (a.pos + k * T)[t32];
#pragma unroll
    for (int k = 0; k < 21; ++k) tg.P.a[k] = ldcov<REAL>(a.P, k, 21, T, t32);
    int32_t mm = 0;
    MREAL ee = (MREAL)1;
    if (has_anchor) {
        mm = (a.ranges + (size_t)al * T)[t32];
        ee = ((const MREAL *)a.err + (size_t)al * T)[t32];
    }
    uint32_t s = 0;
    for (int e = 0; e < a.n_steps; ++e) {
        const double dt = epoch_dt(a, t, e);
        sc.r = mm > 0 ? kc::kf_mm_to_m(mm) : 0.0;
        sc.e = (double)ee;
        sc.w = 0.0;
        if (e + 1 < a.n_steps && has_anchor) { /* next epoch in flight */
            mm = (a.ranges + (size_t)(e + 1) * a.stride_ranges + (size_t)al * T)[t32];
            ee = ((const MREAL *)a.err + (size_t)(e + 1) * a.stride_err + (size_t)al * T)[t32];
        }
        s = kc::step_toa6<true, 0>(tg, sc, pr, dt);
        if (a.traj && al == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) (a.traj + ((size_t)e * 3 + k) * T)[t32] = tg.pos[k];
        }
        if constexpr (cov_is_rounded<REAL>()) {
            if (e + 1 < a.n_steps) {
#pragma unroll
                for (int k = 0; k < 21; ++k) tg.P.a[k] = round_cov<REAL>(tg.P.a[k]);
            }
        }
    }
    if (al != 0) return;
    bool fin = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        (a.pos + k * T)[t32] = tg.pos[k];
        fin &= isfinite(tg.pos[k]);
    }
#pragma unroll
    for (int k = 0; k < 21; ++k) {
        stcov<REAL>(a.P, k, 21, T, t32, tg.P.a[k]);
        fin &= isfinite(tg.P.a[k]);
    }
    if (!fin) s |= ST_NONFINITE;
    a.flags[t] = fl0 | FL_STARTED;
    if (a.status) a.status[t] = s;
}

} // namespace

kfpos_k::step_kernel_t kfpos_k::toa6_coop_kernel(int st) {
    return st == KFPOS_STORE_F32 ? k_step_toa6_coop<float, float>
         : st == KFPOS_STORE_MIXED ? k_step_toa6_coop<double, float>
         : st == KFPOS_STORE_P48 ? k_step_toa6_coop<p48, float> : k_step_toa6_coop<double, double>;
}

/*
 * kfpos_k_imu9.hip -- k_step_imu9: the 9-state UWB + IMU step (KalmanFilterTOAIMU.cpp:100-195), the bench kernel
 */
#include "kfpos_kernels.h"

namespace {

/* ------------------------------------------------------------------ 9-state step kernel */
/* The acceleration sample is fetched one epoch ahead like the ranges. Its covariance is loaded and whitened ONCE per
 * launch: a multi-epoch launch always has one covariance array for all its epochs (stride_cov = 0: a sensor with a
 * fixed covariance); a trace with a covariance per epoch is replayed one epoch per launch (kfpos_run_trace_dev). */
template <typename MREAL>
struct RawImu {
    MREAL acc[3];
};
template <typename MREAL>
__device__ inline void fetch_imu(const KArgs &a, size_t t, int s, RawImu<MREAL> &raw) {
    const MREAL *ap = (const MREAL *)a.accel + (size_t)s * a.stride_accel;
#pragma unroll
    for (int k = 0; k < 3; ++k) raw.acc[k] = (ap + (size_t)k * a.T)[(uint32_t)t];
}
template <typename MREAL>
__device__ inline void fetch_imu_cov(const KArgs &a, size_t t, int s, MREAL raw[9]) {
    const MREAL *cp = (const MREAL *)a.cov + (size_t)s * a.stride_cov;
#pragma unroll
    for (int k = 0; k < 9; ++k) raw[k] = (cp + (size_t)k * a.T)[(uint32_t)t];
}
template <typename MREAL>
__device__ inline void latch_imu_cov(const KArgs &a, size_t T, uint32_t t32, const double cv[9]) {
    strow<MREAL>(a.imu_cov, 0, T, t32, cv[0]);
    strow<MREAL>(a.imu_cov, 1, T, t32, cv[3]);
    strow<MREAL>(a.imu_cov, 2, T, t32, cv[4]);
    strow<MREAL>(a.imu_cov, 3, T, t32, cv[6]);
    strow<MREAL>(a.imu_cov, 4, T, t32, cv[7]);
    strow<MREAL>(a.imu_cov, 5, T, t32, cv[8]);
}

/* RANGING = false: the IMU-only call (MODE_IMU_ONLY), a kernel of its own */
template <typename REAL, typename MREAL, int AS, bool RANGING = true>
__global__ __launch_bounds__(WAVE) void k_step_imu9(const KArgs a) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const size_t t = (size_t)blockIdx.x * WAVE + lane;
    if (t >= (size_t)a.T) return;
    const size_t T = a.T;
    const uint32_t t32 = (uint32_t)t;
    Params pr = make_params(a);
    constexpr bool has_ranging = RANGING;
    /* the next epoch's measurements are fetched one epoch AHEAD, behind the current epoch's arithmetic -- in the
     * KFPOS_STORE_MIXED instantiation (the bench configuration); the other two (8-byte measurements: 22 more registers
     * per lane across the whole step; 4-byte covariance: its rounding code) do not have the registers for that (they
     * spill), so they fetch between two epochs instead */
    constexpr bool AHEAD = sizeof(MREAL) == 4 && !std::is_same<REAL, float>::value;
    const bool fresh_imu = a.mode != MODE_TOA;
    constexpr int NA = AS > 0 ? AS : 1;
    if (a.n_steps == 1 && a.dt && a.dt[t32] < 0.0) { /* no epoch / sample for this tag in this call */
        skipped_lane(a, t, true);
        return;
    }

    /* load order = order of first use (see k_step_toa6): flags, epoch, position, velocity, IMU sample, then the
     * 45 covariance entries, which are not needed until the ML solve is over. The flags word goes first: the compiler
     * parks it in an AGPR straight away, and vmcnt counts loads in order -- as the last load of the first group it
     * made that move wait for the whole group before the second group (IMU sample, covariance) was even issued */
    uint32_t fl = a.flags[t32];
    RawEpoch<MREAL, NA> raw;
    RawImu<MREAL> rawi;
    if constexpr (AS > 0) {
        if (has_ranging) fetch_epoch<MREAL, AS>(a, t, 0, raw);
    }
    Tag9 tg;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        tg.pos[k] = (a.pos + k * T)[t32];
        tg.vel[k] = (a.vel + k * T)[t32];
    }
    /* the covariance and B^-1 wait in LDS while the gain iteration runs, the accelerometer whitener for the whole
     * launch: [78][lane], behind the generic kernel's epoch scratch */
    const CovPark9 park{lds + ((AS == 0 && has_ranging) ? 3 * (size_t)a.A * WAVE : 0) + lane, WAVE};
    Imu imu;
    imu.has = false;
    imu.ci = park.a + 66 * WAVE;
    imu.ci_stride = WAVE;
    if constexpr (AS == 8 && RANGING) { /* the anchor table once more, where lanes can index it one by one (iekf9_pairs) */
        if (a.pair9) {
            double *tab = lds + 78 * WAVE;
#pragma unroll
            for (int k = 0; k < 24; ++k) tab[k] = a.anchors[k]; /* (every lane writes the same 24 numbers) */
            pr.pair_anchor_tab = tab;
        }
    }
    double cv[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    MREAL rawc[9];
    if (fresh_imu) {
        fetch_imu<MREAL>(a, t, 0, rawi);
        fetch_imu_cov<MREAL>(a, t, 0, rawc);
    } else if (fl & FL_HAS_IMU) { /* re-fuse the latched sample (KalmanFilterTOAIMU.cpp:68-72) */
        imu.has = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) imu.acc[k] = ldrow<MREAL>(a.imu_acc, k, T, t32);
        cv[0] = ldrow<MREAL>(a.imu_cov, 0, T, t32);
        cv[3] = ldrow<MREAL>(a.imu_cov, 1, T, t32);
        cv[4] = ldrow<MREAL>(a.imu_cov, 2, T, t32);
        cv[6] = ldrow<MREAL>(a.imu_cov, 3, T, t32);
        cv[7] = ldrow<MREAL>(a.imu_cov, 4, T, t32);
        cv[8] = ldrow<MREAL>(a.imu_cov, 5, T, t32);
    }
#pragma unroll
    for (int k = 0; k < 45; ++k) tg.P.a[k] = ldcov<REAL>(a.P, k, 45, T, t32);
    if (fresh_imu) { /* the covariance of the first (usually: of every) epoch of this launch */
        imu.has = true;
#pragma unroll
        for (int k = 0; k < 9; ++k) cv[k] = (double)rawc[k];
        if (a.latch) latch_imu_cov<MREAL>(a, T, t32, cv);
    }
    if (imu.has) imu_whitener(cv, imu.ci, imu.ci_stride);


    uint32_t s = 0;
    for (int e = 0; e < a.n_steps; ++e) { /* the state stays in registers from epoch to epoch */
        const double dt = epoch_dt(a, t, e);
        if (fresh_imu) { /* fresh sample: newIMUMeasurement latches it (KalmanFilterTOAIMU.cpp:78-89) */
#pragma unroll
            for (int k = 0; k < 3; ++k) imu.acc[k] = (double)rawi.acc[k];
            if (e + 1 < a.n_steps) {
                if constexpr (AHEAD) fetch_imu<MREAL>(a, opaque_lane(t), opaque_uniform(e + 1), rawi);
            }
        }
        if constexpr (AS > 0) {
            RegScratch<AS> sc;
            if (has_ranging) {
                unpack_epoch<MREAL, AS>(raw, sc);
                if (e + 1 < a.n_steps) {
                    if constexpr (AHEAD) fetch_epoch<MREAL, AS>(a, opaque_lane(t), opaque_uniform(e + 1), raw);
                }
            } else {
#pragma unroll
                for (int k = 0; k < AS; ++k) sc.r[k] = sc.e[k] = sc.w[k] = 0.0;
            }
            s = step_imu9<RANGING>(tg, sc, pr, dt, imu, park);
            if constexpr (!AHEAD) { /* 8-byte measurements: the next epoch is fetched when this one is over */
                if (e + 1 < a.n_steps) {
                    if (fresh_imu) fetch_imu<MREAL>(a, opaque_lane(t), opaque_uniform(e + 1), rawi);
                    if (has_ranging) fetch_epoch<MREAL, AS>(a, opaque_lane(t), opaque_uniform(e + 1), raw);
                }
            }
        } else {
            Scratch sc{nullptr, nullptr, nullptr, WAVE};
            if (has_ranging) sc = stage_epoch_lds<MREAL>(a, lds, lane, t, opaque_uniform(e));
            s = step_imu9<RANGING>(tg, sc, pr, dt, imu, park);
            if constexpr (!AHEAD) { /* the ranges are staged per epoch above; the next accelerometer sample is not */
                if (e + 1 < a.n_steps && fresh_imu) fetch_imu<MREAL>(a, opaque_lane(t), opaque_uniform(e + 1), rawi);
            }
        }
        if (a.traj) { /* the pose a per-epoch caller would have read back (getPose at timeLag 0) */
#pragma unroll
            for (int k = 0; k < 3; ++k) (a.traj + ((size_t)opaque_uniform(e) * 3 + k) * T)[t32] = tg.pos[k];
        }
        if constexpr (cov_is_rounded<REAL>()) { /* what n single-epoch launches would have kept in HBM */
            if (e + 1 < a.n_steps) {
#pragma unroll
                for (int k = 0; k < 45; ++k) tg.P.a[k] = round_cov<REAL>(tg.P.a[k]);
            }
        }
    }

    if (fresh_imu && a.latch) { /* the last epoch's sample stays latched (lastImuMeasurement, KalmanFilterTOAIMU.cpp:78-89) */
#pragma unroll
        for (int k = 0; k < 3; ++k) strow<MREAL>(a.imu_acc, k, T, t32, imu.acc[k]);
        fl |= FL_HAS_IMU;
    }
    bool fin = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        (a.pos + k * T)[t32] = tg.pos[k];
        (a.vel + k * T)[t32] = tg.vel[k];
        fin &= isfinite(tg.pos[k]) & isfinite(tg.vel[k]);
    }
#pragma unroll
    for (int k = 0; k < 45; ++k) {
        stcov<REAL>(a.P, k, 45, T, t32, tg.P.a[k]);
        fin &= isfinite(tg.P.a[k]);
    }
    const bool waiting = !a.use_init_pos && isnan(tg.pos[0]);
    if (!fin && !waiting) s |= ST_NONFINITE;
    a.flags[t32] = fl | FL_STARTED;
    if (a.status) a.status[t32] = s;
}

} // namespace

template <typename REAL, typename MREAL>
static kfpos_k::step_kernel_t imu9_of(int as, bool ranging) {
    if (!ranging) return k_step_imu9<REAL, MREAL, 0, false>; /* no epoch: the anchor count plays no role */
    if (as == 8) return k_step_imu9<REAL, MREAL, 8>;
    return k_step_imu9<REAL, MREAL, 0>;
}
kfpos_k::step_kernel_t kfpos_k::imu9_kernel(int st, int as, bool ranging) {
    return KFPOS_BY_STORAGE(st, imu9_of, as, ranging);
}

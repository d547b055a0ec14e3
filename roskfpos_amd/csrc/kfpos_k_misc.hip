/*
 * kfpos_k_misc.hip -- k_step_ml (MLLocation as the estimator), k_step_planar (KalmanFilter, 8 states), k_get_pose, layout turns
 */
#include "kfpos_kernels.h"

namespace {

/* ------------------------------------------------------------------ standalone ML estimator kernel */
template <typename REAL, typename MREAL, int AS>
__global__ __launch_bounds__(WAVE) void k_step_ml(const KArgs a) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const size_t t = (size_t)blockIdx.x * WAVE + lane;
    if (t >= (size_t)a.T) return;
    const size_t T = a.T;
    const uint32_t t32 = (uint32_t)t;
    const Params pr = make_params(a);
    if (a.n_steps == 1 && a.dt && a.dt[t] < 0.0) {
        skipped_lane(a, t, true);
        return;
    }
    const auto fl0 = a.flags[t]; /* (read at the head, written back at the end: see k_step_toa6) */
    /* _previousEstimation: the per-tag seed lives in the velocity slot of the handle, it is never updated */
    double seed[3] = {1.0, 1.0, 4.0};
    if (a.use_init_pos) {
#pragma unroll
        for (int k = 0; k < 3; ++k) seed[k] = (a.vel + k * T)[t32];
    }
    double pos[3], cov[6];
    uint32_t s = 0;
    for (int e = 0; e < a.n_steps; ++e) {
        if constexpr (AS > 0) {
            RawEpoch<MREAL, AS> raw;
            fetch_epoch<MREAL, AS>(a, t, e, raw);
            RegScratch<AS> sc;
            unpack_epoch<MREAL, AS>(raw, sc);
            s = step_ml(pos, cov, sc, pr, seed);
        } else {
            Scratch sc = stage_epoch_lds<MREAL>(a, lds, lane, t, e);
            s = step_ml(pos, cov, sc, pr, seed);
        }
        if (a.traj && !(s & ST_UPDATE_SKIPPED)) {
#pragma unroll
            for (int k = 0; k < 3; ++k) (a.traj + ((size_t)e * 3 + k) * T)[t32] = pos[k];
        }
    }
    if (!(s & ST_UPDATE_SKIPPED)) {
#pragma unroll
        for (int k = 0; k < 3; ++k) (a.pos + k * T)[t32] = pos[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) stcov<REAL>(a.P, k, 6, T, t32, cov[k]);
    }
    a.flags[t] = fl0 | FL_STARTED;
    if (a.status) a.status[t] = s;
}

/* ------------------------------------------------------------------ 8-state planar step kernel */
/* KalmanFilter (ALGORITHM_KF). a.mode: 0 = ranging epoch (carries whatever the tag has latched), 1..4 = one of
 * the four other sensor entry points (KFPOS_SENSOR_*), which latch their sample and run an update without
 * ranging rows. SENS = false is the ranging-only bank: no latch traffic, closed-form 2x2 update. Flags word:
 * bit 0 started, bits 5..7 = latched PX4Flow / IMU / magnetometer. */

template <bool SENS, typename REAL, typename MREAL, int AS>
__global__ __launch_bounds__(WAVE) void k_step_planar(const KArgs a) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const size_t t = (size_t)blockIdx.x * WAVE + lane;
    if (t >= (size_t)a.T) return;
    const size_t T = a.T;
    const uint32_t t32 = (uint32_t)t;
    const Params pr = make_params(a);
    const int kind = a.mode;
    const bool has_ranging = kind == 0;
    constexpr int NA = AS > 0 ? AS : 1;
    if (a.n_steps == 1 && a.dt && a.dt[t] < 0.0) { /* no epoch / sample for this tag in this call */
        skipped_lane(a, t, true);
        return;
    }
    RawEpoch<MREAL, NA> raw;
    if constexpr (AS > 0) {
        if (has_ranging) fetch_epoch<MREAL, AS>(a, t, 0, raw);
    }
    Tag8 tg;
    tg.xy[0] = (a.pos + 0 * T)[t32];
    tg.xy[1] = (a.pos + 1 * T)[t32];
    tg.z = (a.pos + 2 * T)[t32];
    tg.vel[0] = (a.vel + 0 * T)[t32];
    tg.vel[1] = (a.vel + 1 * T)[t32];
    tg.ang = (a.vel + 2 * T)[t32];
    tg.om = (a.vel + 3 * T)[t32];
    uint32_t fl = a.flags[t];
    Latch8 lt;
    lt.has = SENS ? ((fl >> PLANAR_HAS_SHIFT) & (ROW_PX4 | ROW_IMU | ROW_MAG)) : 0u;
#pragma unroll
    for (int k = 0; k < 5; ++k) lt.px4[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) lt.imu[k] = 0.0;
    lt.mag[0] = lt.mag[1] = 0.0;
    uint32_t rows = ROW_RANGING;
    if constexpr (SENS) {
        /* this call's sample (KalmanFilter.cpp:102-229); a PX4Flow sample of quality 0 is dropped on entry */
        if (kind == KFPOS_SENSOR_PX4FLOW) {
            double f[5], m[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) f[k] = (a.sensor + k * T)[t32];
            if (!px4_sample(pr, f, m)) {
                skipped_lane(a, t, true);
                return;
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) { lt.px4[k] = m[k]; (a.platch + k * T)[t32] = m[k]; }
            lt.has |= ROW_PX4;
            rows = ROW_PX4;
        } else if (kind == KFPOS_SENSOR_IMU) {
            double w3[3], cw[9], la[3], ca[9];
#pragma unroll
            for (int k = 0; k < 3; ++k) { w3[k] = (a.sensor + k * T)[t32]; la[k] = (a.sensor + (12 + k) * T)[t32]; }
#pragma unroll
            for (int k = 0; k < 9; ++k) { cw[k] = (a.sensor + (3 + k) * T)[t32]; ca[k] = (a.sensor + (15 + k) * T)[t32]; }
            imu_sample8(pr, w3, cw, la, ca, lt.imu);
#pragma unroll
            for (int k = 0; k < 8; ++k) (a.platch + (5 + k) * T)[t32] = lt.imu[k];
            lt.has |= ROW_IMU;
            rows = ROW_IMU;
        } else if (kind == KFPOS_SENSOR_MAG || kind == KFPOS_SENSOR_COMPASS) {
            if (kind == KFPOS_SENSOR_MAG)
                lt.mag[0] = atan2((a.sensor + 1 * T)[t32], (a.sensor + 0 * T)[t32]) - pr.mag_offset; /* :188 */
            else
                lt.mag[0] = normalize_angle(a.sensor[t32]); /* :207 */
            lt.mag[1] = pr.mag_cov;
            (a.platch + 13 * T)[t32] = lt.mag[0];
            (a.platch + 14 * T)[t32] = lt.mag[1];
            const uint32_t before = lt.has;
            lt.has |= ROW_MAG;
            rows = kind == KFPOS_SENSOR_MAG ? ROW_MAG : (ROW_MAG | (before & (ROW_PX4 | ROW_IMU)));
        } else {
            rows = ROW_RANGING | lt.has; /* newTOAMeasurement: the latched samples ride along (:84-98) */
        }
        /* latched samples this call carries but did not bring itself */
        if ((rows & ROW_PX4) && kind != KFPOS_SENSOR_PX4FLOW) {
#pragma unroll
            for (int k = 0; k < 5; ++k) lt.px4[k] = (a.platch + k * T)[t32];
        }
        if ((rows & ROW_IMU) && kind != KFPOS_SENSOR_IMU) {
#pragma unroll
            for (int k = 0; k < 8; ++k) lt.imu[k] = (a.platch + (5 + k) * T)[t32];
        }
        if ((rows & ROW_MAG) && kind == 0) {
            lt.mag[0] = (a.platch + 13 * T)[t32];
            lt.mag[1] = (a.platch + 14 * T)[t32];
        }
    }
#pragma unroll
    for (int k = 0; k < 36; ++k) tg.P.a[k] = ldcov<REAL>(a.P, k, 36, T, t32);

    /* SENS: the predicted covariance is parked in LDS, [36][lane], behind the generic kernel's epoch scratch */
    const CovSpill8 park{lds + (AS < 0 ? 3 * (size_t)(-AS) * WAVE : ((AS == 0 && has_ranging) ? 3 * (size_t)a.A * WAVE : 0)) + lane, WAVE};
    uint32_t s = 0;
    for (int e = 0; e < a.n_steps; ++e) { /* the state stays in registers from epoch to epoch */
        const double dt = epoch_dt(a, t, e);
        if constexpr (AS > 0) {
            RegScratch<AS> sc;
            if (has_ranging) {
                unpack_epoch<MREAL, AS>(raw, sc);
                if (e + 1 < a.n_steps) fetch_epoch<MREAL, AS>(a, t, e + 1, raw);
            } else {
#pragma unroll
                for (int k = 0; k < AS; ++k) sc.r[k] = sc.e[k] = sc.w[k] = 0.0;
            }
            s = step_planar8<SENS>(tg, sc, pr, dt, rows, lt, park);
        } else if constexpr (AS < 0) { /* compile-time anchor loops over the LDS-resident epoch (ranging epochs only) */
            StaticScratch<-AS> sc = stage_epoch_lds_n<MREAL, -AS>(a, lds, lane, t, e);
            s = step_planar8<SENS>(tg, sc, pr, dt, rows, lt, park);
        } else {
            Scratch sc{nullptr, nullptr, nullptr, WAVE};
            if (has_ranging) sc = stage_epoch_lds<MREAL>(a, lds, lane, t, e);
            s = step_planar8<SENS>(tg, sc, pr, dt, rows, lt, park);
        }
        if (a.traj) { /* the pose a per-epoch caller would have read back (getPose at timeLag 0) */
            (a.traj + ((size_t)e * 3 + 0) * T)[t32] = tg.xy[0];
            (a.traj + ((size_t)e * 3 + 1) * T)[t32] = tg.xy[1];
            (a.traj + ((size_t)e * 3 + 2) * T)[t32] = tg.z;
        }
        if constexpr (cov_is_rounded<REAL>()) { /* what n single-epoch launches would have kept in HBM */
            if (e + 1 < a.n_steps) {
#pragma unroll
                for (int k = 0; k < 36; ++k) tg.P.a[k] = round_cov<REAL>(tg.P.a[k]);
            }
        }
    }

    bool fin = isfinite(tg.xy[0]) & isfinite(tg.xy[1]) & isfinite(tg.z) & isfinite(tg.vel[0]) & isfinite(tg.vel[1]) &
               isfinite(tg.ang) & isfinite(tg.om);
    (a.pos + 0 * T)[t32] = tg.xy[0];
    (a.pos + 1 * T)[t32] = tg.xy[1];
    (a.pos + 2 * T)[t32] = tg.z;
    (a.vel + 0 * T)[t32] = tg.vel[0];
    (a.vel + 1 * T)[t32] = tg.vel[1];
    (a.vel + 2 * T)[t32] = tg.ang;
    (a.vel + 3 * T)[t32] = tg.om;
#pragma unroll
    for (int k = 0; k < 36; ++k) {
        stcov<REAL>(a.P, k, 36, T, t32, tg.P.a[k]);
        fin &= isfinite(tg.P.a[k]);
    }
    const bool waiting = !a.use_init_pos && isnan(tg.xy[0]);
    if (!fin && !waiting) s |= ST_NONFINITE;
    fl |= FL_STARTED;
    if constexpr (SENS) fl |= lt.has << PLANAR_HAS_SHIFT;
    a.flags[t] = fl;
    if (a.status) a.status[t] = s;
}

/* ------------------------------------------------------------------ pose kernel (getPose) */
template <int MODEL, bool SYMM, typename REAL>
__global__ __launch_bounds__(WAVE) void k_get_pose(const PoseArgs a) {
    const size_t t = (size_t)blockIdx.x * WAVE + threadIdx.x;
    if (t >= (size_t)a.T) return;
    const size_t T = a.T;
    const uint32_t t32 = (uint32_t)t;
    double pos[3], vel[3] = {0, 0, 0}, cov[9];
    uint32_t s = 0;
    const double ahead = a.dt_each ? a.dt_each[t] : a.dt_ahead;
    if (!(a.flags[t] & FL_STARTED)) {
        s = ST_NOT_STARTED;
#pragma unroll
        for (int k = 0; k < 3; ++k) pos[k] = vel[k] = NAN;
#pragma unroll
        for (int k = 0; k < 9; ++k) cov[k] = NAN;
        if (a.full_P) {
            constexpr int N = MODEL == 6 ? 6 : (MODEL == 3 ? 3 : (MODEL == 8 ? 8 : 9));
#pragma unroll
            for (int i = 0; i < N; ++i) (a.full_x + i * T)[t32] = NAN;
            for (int i = 0; i < N * N; ++i) (a.full_P + (size_t)i * T)[t32] = NAN;
        }
    } else if (MODEL == 3) { /* MLLocation::getPose: the estimate as it is */
#pragma unroll
        for (int k = 0; k < 3; ++k) pos[k] = (a.pos_in + k * T)[t32];
        const double c[6] = {ldcov<REAL>(a.P, 0, 6, T, t32), ldcov<REAL>(a.P, 1, 6, T, t32), ldcov<REAL>(a.P, 2, 6, T, t32),
                             ldcov<REAL>(a.P, 3, 6, T, t32), ldcov<REAL>(a.P, 4, 6, T, t32), ldcov<REAL>(a.P, 5, 6, T, t32)};
        cov[0] = c[0]; cov[1] = c[1]; cov[2] = c[2]; cov[3] = c[1]; cov[4] = c[3]; cov[5] = c[4];
        cov[6] = c[2]; cov[7] = c[4]; cov[8] = c[5];
        if (a.full_P) {
#pragma unroll
            for (int i = 0; i < 3; ++i) (a.full_x + i * T)[t32] = pos[i];
#pragma unroll
            for (int i = 0; i < 9; ++i) (a.full_P + (size_t)i * T)[t32] = cov[i];
        }
    } else if (MODEL == 8) { /* KalmanFilter::getPose, KalmanFilter.cpp:709-745 */
        Tag8 tg;
        tg.xy[0] = (a.pos_in + 0 * T)[t32];
        tg.xy[1] = (a.pos_in + 1 * T)[t32];
        tg.z = (a.pos_in + 2 * T)[t32];
        tg.vel[0] = (a.vel_in + 0 * T)[t32];
        tg.vel[1] = (a.vel_in + 1 * T)[t32];
        tg.ang = (a.vel_in + 2 * T)[t32];
        tg.om = (a.vel_in + 3 * T)[t32];
#pragma unroll
        for (int k = 0; k < 36; ++k) tg.P.a[k] = ldcov<REAL>(a.P, k, 36, T, t32);
        double x8[8];
        Cov<8, true> Pp;
        pose8(tg, ahead, a.accel_noise, a.jolt, x8, Pp);
        pos[0] = x8[0]; pos[1] = x8[1]; pos[2] = tg.z;
        vel[0] = x8[2]; vel[1] = x8[3]; vel[2] = 0.0;
        /* position block of stateToPose's 6x6: eye * 0.01 with the xy block of P (:349-354) */
        cov[0] = Pp(0, 0); cov[1] = Pp(0, 1); cov[2] = 0.0;
        cov[3] = Pp(0, 1); cov[4] = Pp(1, 1); cov[5] = 0.0;
        cov[6] = 0.0; cov[7] = 0.0; cov[8] = 0.01;
        if (a.full_P) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                (a.full_x + i * T)[t32] = x8[i];
#pragma unroll
                for (int j = 0; j < 8; ++j) (a.full_P + (size_t)(i * 8 + j) * T)[t32] = Pp(i, j);
            }
        }
    } else if (MODEL == 6) {
        Tag6<SYMM> tg;
#pragma unroll
        for (int k = 0; k < 3; ++k) tg.pos[k] = (a.pos_in + k * T)[t32];
#pragma unroll
        for (int k = 0; k < Cov<6, SYMM>::SZ; ++k) tg.P.a[k] = ldcov<REAL>(a.P, k, Cov<6, SYMM>::SZ, T, t32);
        pose6<SYMM>(tg, ahead, a.accel_noise, pos, cov);
        if (a.full_P) { /* the whole predicted covariance, as getPose computes it (KalmanFilterTOA.cpp:467-468) */
            predict6(tg.P, ahead, a.accel_noise);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                (a.full_x + i * T)[t32] = i < 3 ? tg.pos[i] : 0.0;
#pragma unroll
                for (int j = 0; j < 6; ++j) (a.full_P + (size_t)(i * 6 + j) * T)[t32] = tg.P(i, j);
            }
        }
    } else {
        Tag9 tg;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            tg.pos[k] = (a.pos_in + k * T)[t32];
            tg.vel[k] = (a.vel_in + k * T)[t32];
        }
#pragma unroll
        for (int k = 0; k < 45; ++k) tg.P.a[k] = ldcov<REAL>(a.P, k, 45, T, t32);
        pose9(tg, ahead, a.jolt, pos, vel, cov);
        if (a.full_P) { /* KalmanFilterTOAIMU.cpp:503-506 */
            predict9(tg.P, ahead, a.jolt);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                (a.full_x + i * T)[t32] = i < 3 ? pos[i] : (i < 6 ? vel[i - 3] : 0.0);
#pragma unroll
                for (int j = 0; j < 9; ++j) (a.full_P + (size_t)(i * 9 + j) * T)[t32] = tg.P(i, j);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (a.pos) (a.pos + k * T)[t32] = pos[k];
        if (a.vel) (a.vel + k * T)[t32] = vel[k];
    }
    if (a.cov) {
#pragma unroll
        for (int k = 0; k < 9; ++k) (a.cov + k * T)[t32] = cov[k];
    }
    if (a.status) a.status[t] = s;
}

/* ------------------------------------------------------------------ layout kernels of the host-buffer API */
/* The host API takes and returns row-major [tag][component] arrays (one row per reference call); the step kernels
 * want [component][tag]. The turn is done on the device, next to one plain copy per array, instead of element by
 * element on the CPU. E = 4- or 8-byte element. */
template <typename E>
__global__ __launch_bounds__(256) void k_rows_to_cols(const E *src, E *dst, int T, int C) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)T) return;
    for (int c = 0; c < C; ++c) dst[(size_t)c * T + t] = src[t * C + c];
}
template <typename E>
__global__ __launch_bounds__(256) void k_cols_to_rows(const E *src, E *dst, int T, int C) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)T) return;
    for (int c = 0; c < C; ++c) dst[t * C + c] = src[(size_t)c * T + t];
}

} // namespace

template <typename REAL, typename MREAL>
static kfpos_k::step_kernel_t ml_of(int as) {
    if (as == 8) return k_step_ml<REAL, MREAL, 8>;
    return k_step_ml<REAL, MREAL, 0>;
}
kfpos_k::step_kernel_t kfpos_k::ml_kernel(int st, int as) {
    return KFPOS_BY_STORAGE(st, ml_of, as);
}

/* With sensor rows the register-resident epoch of the 8-anchor specialisation no longer fits (measured: 212-244
 * bytes/lane of scratch), so those banks always run the LDS-staged kernel. */
template <bool SENS, typename REAL, typename MREAL>
static kfpos_k::step_kernel_t planar_of(int as) {
    if constexpr (!SENS) {
        if (as == 8) return k_step_planar<false, REAL, MREAL, 8>;
    } else {
        if (as == 8) return k_step_planar<true, REAL, MREAL, -8>; /* epoch in LDS, anchor loops compile-time */
    }
    return k_step_planar<SENS, REAL, MREAL, 0>;
}
template <typename REAL, typename MREAL>
static kfpos_k::step_kernel_t planar_sens(int as) { return planar_of<true, REAL, MREAL>(as); }
template <typename REAL, typename MREAL>
static kfpos_k::step_kernel_t planar_plain(int as) { return planar_of<false, REAL, MREAL>(as); }
kfpos_k::step_kernel_t kfpos_k::planar_kernel(int st, bool sensors, int as) {
    return sensors ? KFPOS_BY_STORAGE(st, planar_sens, as) : KFPOS_BY_STORAGE(st, planar_plain, as);
}

template <int MODEL, bool SYMM>
static void pose_by_storage(int st, int blocks, hipStream_t s, const PoseArgs &a) {
    if (st == KFPOS_STORE_F32) hipLaunchKernelGGL((k_get_pose<MODEL, SYMM, float>), dim3(blocks), dim3(WAVE), 0, s, a);
    else if (st == KFPOS_STORE_P48) hipLaunchKernelGGL((k_get_pose<MODEL, SYMM, p48>), dim3(blocks), dim3(WAVE), 0, s, a);
    else hipLaunchKernelGGL((k_get_pose<MODEL, SYMM, double>), dim3(blocks), dim3(WAVE), 0, s, a);
}
void kfpos_k::launch_get_pose(int model, bool full, int st, int blocks, hipStream_t s, const PoseArgs &a) {
    if (model == KFPOS_MODEL_PLANAR) pose_by_storage<8, true>(st, blocks, s, a);
    else if (model == KFPOS_MODEL_ML) pose_by_storage<3, true>(st, blocks, s, a);
    else if (model == KFPOS_MODEL_TOA_IMU) pose_by_storage<9, true>(st, blocks, s, a);
    else if (full) pose_by_storage<6, false>(st, blocks, s, a);
    else pose_by_storage<6, true>(st, blocks, s, a);
}

void kfpos_k::launch_rows_to_cols(size_t esz, hipStream_t s, const void *src, void *dst, int T, int C) {
    const int blocks = (T + 255) / 256;
    if (esz == 4) hipLaunchKernelGGL(k_rows_to_cols<uint32_t>, dim3(blocks), dim3(256), 0, s, (const uint32_t *)src, (uint32_t *)dst, T, C);
    else hipLaunchKernelGGL(k_rows_to_cols<uint64_t>, dim3(blocks), dim3(256), 0, s, (const uint64_t *)src, (uint64_t *)dst, T, C);
}
void kfpos_k::launch_cols_to_rows(hipStream_t s, const double *src, double *dst, int T, int C) {
    const int blocks = (T + 255) / 256;
    hipLaunchKernelGGL(k_cols_to_rows<uint64_t>, dim3(blocks), dim3(256), 0, s, (const uint64_t *)src, (uint64_t *)dst, T, C);
}

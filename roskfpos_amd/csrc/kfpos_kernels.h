/*
 * kfpos_kernels.h -- what the kernel translation units of libkfpos_hip.so share: the kernel-argument blocks, the
 * device-side helpers that stage an epoch (component-major HBM -> registers / LDS), and the selector functions through
 * which the host side (kfpos_hip.hip) obtains a kernel without seeing its template.
 *
 * Execution model: ONE FILTER PER LANE, 64 filters per wavefront, one wavefront per workgroup.
 *  - The whole per-tag state (position, velocity, packed covariance: 21 / 36 / 45 doubles) lives in
 *    VGPRs/AGPRs for the duration of a step -- and across the epochs of a multi-epoch launch; every array
 *    index in kfpos_core.h is a compile-time constant after unrolling. At 65 536 tags there is exactly one
 *    wavefront per SIMD (1024 waves on 256 CUs x 4 SIMDs), so the 512-register file per lane is free to use.
 *    No kernel may spill to scratch inside a loop (checked at build time, tools/check_scratch.py).
 *  - HBM layout is component-major ([component][tag]): lane l of a wave reads element
 *    base + tag0 + l, so every state / measurement access is one fully coalesced
 *    512-byte (f64) or 256-byte (f32 / int32) wave transaction, each byte touched once.
 *  - The epoch's measurements -- range in metres (the integer-mm wire value converted once, with the
 *    reference's exact `(double) mm / 1000`, Posgenerator.cpp:484), errorEstimation, working weight (1/e for
 *    the ML sweeps, 1/R for the IEKF sweeps) -- live in registers for 8 anchors (RegScratch) and per lane in
 *    LDS otherwise, [anchor][lane] (lane-consecutive 8-byte words: conflict-free ds_read_b64), with
 *    compile-time anchor loops for 16 anchors (StaticScratch) and a run-time loop for every other count. The
 *    inner sweeps (2-4 ML + 3-20 IEKF per step) then touch only registers / LDS + SGPRs, never HBM.
 *  - Anchor coordinates are wave-uniform: they travel in the kernel-argument segment and are read
 *    with scalar loads into SGPRs.
 *  - No MFMA: the largest dense object is 9x9 per filter; lanes are independent filters, so there is nothing to
 *    shuffle and no barrier (the exception: the 8-lanes-per-tag kernel of small banks, kfpos_k_coop.hip).
 *
 * Translation units (one code object each, built in parallel): kfpos_k_toa6s / kfpos_k_toa6f (6-state filter, symmetric /
 * full covariance layout), kfpos_k_coop (6-state, 8 lanes per tag), kfpos_k_imu9 (9-state), kfpos_k_misc (8-state planar
 * filter, standalone ML estimator, getPose, layout turns), kfpos_hip (host side + C ABI), kfpos_comm (RCCL gather).
 */
#ifndef KFPOS_KERNELS_H
#define KFPOS_KERNELS_H

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <type_traits>

#define KFPOS_HD __host__ __device__
#include "kfpos_core.h"
#include "kfpos_internal.h"
#include "kfpos_p48.h"

namespace kfpos_k {

/* kernel arguments: everything wave-uniform, read through scalar loads */
struct KArgs {
    double anchors[KFPOS_MAX_ANCHORS * 3];
    int T, A;
    double accel_noise, jolt, cost_threshold;
    int ignore_worst, top_n, use_init_pos, ml_variant;
    int pair9;        /* 9-state kernel: two lanes per tag for the tail of the gain iteration (KFPOS_PAIR9=1; off by default: DESIGN 6a) */
    /* planar filter configuration (kfpos_planar_config) */
    int use_fixed_height, imu_fixed_cov_acc, imu_fixed_cov_w;
    double px4_height, px4_arm_p1, px4_arm_p2, px4_cov_vel, px4_cov_gyro_z;
    double imu_cov_acc, imu_cov_w, mag_offset, mag_cov;
    double *platch;   /* [15][T] planar filter: latched PX4Flow (5), IMU (8), magnetometer (2) samples */
    const double *sensor; /* planar sensor call: [C][T] sample of this call (C = 5 / 24 / 3 / 1) */
    /* persistent state, component-major */
    double *pos;      /* [3][T] */
    double *vel;      /* [3][T] (9-state; always f64: the 9-state filter amplifies velocity rounding);
                         planar filter: [4][T] = vx, vy, theta, omega */
    void *P;          /* [SZ][T] real */
    uint32_t *flags;  /* [T] */
    void *imu_acc;    /* [3][T] real, latched sample (9-state) */
    void *imu_cov;    /* [6][T] real, lower triangle {00,10,11,20,21,22} of the latched covariance */
    /* epoch inputs */
    const int32_t *ranges; /* [A][T] */
    const void *err;       /* [A][T] real */
    const double *dt;      /* [T] or null */
    double dt_shared;
    const void *accel;     /* [3][T] real */
    const void *cov;       /* [9][T] real */
    int mode, latch;
    uint32_t *status;      /* [T] or null */
    /* multi-epoch launches (kfpos_run_trace_dev): epoch s reads its inputs at base + s * stride (elements),
     * dt_steps[s] is its shared dt. n_steps = 1 is the single-epoch case and uses dt / dt_shared. */
    int n_steps;
    long long stride_ranges, stride_err, stride_accel, stride_cov;
    double *traj;          /* [n_steps][3][T] positions after each epoch, or null */
    double dt_steps[KFPOS_TRACE_CHUNK];
};


struct PoseArgs {
    int T, model, full;
    double accel_noise, jolt, dt_ahead;
    const double *dt_each; /* [T] per-tag extrapolation time, or null to use dt_ahead */
    const double *pos_in;
    const double *vel_in;
    const void *P;
    const uint32_t *flags;
    double *pos, *cov, *vel; /* [3][T], [9][T], [3][T]; any may be null */
    double *full_x, *full_P; /* [n][T], [n*n][T] predicted state / covariance (row-major index first), or null */
    uint32_t *status;
};

typedef void (*step_kernel_t)(const KArgs);

/* ---- selectors: each is defined in the translation unit that instantiates the kernels it hands out ----
 * st = KFPOS_STORE_*; as = anchor-count specialisation (8: epoch in registers; -8 / -16: compile-time loops over an
 * LDS-resident epoch; 0: run-time loop); heur: 0 = no outlier heuristic, 1 = top-N only, 2 = leave-one-out */
step_kernel_t toa6_sym_kernel(int st, int as, int heur, bool two_waves);   /* kfpos_k_toa6s.hip */
step_kernel_t toa6_full_kernel(int st, int as, int heur);                  /* kfpos_k_toa6f.hip */
step_kernel_t toa6_coop_kernel(int st);                                    /* kfpos_k_coop.hip */
step_kernel_t imu9_kernel(int st, int as, bool ranging);                   /* kfpos_k_imu9.hip */
step_kernel_t ml_kernel(int st, int as);                                   /* kfpos_k_misc.hip */
step_kernel_t planar_kernel(int st, bool sensors, int as);                 /* kfpos_k_misc.hip */
void launch_get_pose(int model, bool full, int st, int blocks, hipStream_t s, const PoseArgs &a); /* kfpos_k_misc.hip */
void launch_rows_to_cols(size_t esz, hipStream_t s, const void *src, void *dst, int T, int C);    /* kfpos_k_misc.hip */
void launch_cols_to_rows(hipStream_t s, const double *src, double *dst, int T, int C);            /* kfpos_k_misc.hip */

constexpr int COOP_LANES = 8;                    /* kfpos_k_coop.hip: one tag per group of 8 lanes */
constexpr int COOP_TAGS_PER_WAVE = 64 / COOP_LANES;
constexpr int PLANAR_HAS_SHIFT = 4;              /* planar flags word: bits 5..7 = latched PX4Flow / IMU / magnetometer */
constexpr int LATCH_ROWS = 15;

} // namespace kfpos_k

namespace {

using namespace kfpos;
using kfpos_k::KArgs;
using kfpos_k::PoseArgs;
using kfpos_k::step_kernel_t;
using kfpos_k::COOP_LANES;
using kfpos_k::COOP_TAGS_PER_WAVE;
using kfpos_k::PLANAR_HAS_SHIFT;
using kfpos_k::LATCH_ROWS;

constexpr int WAVE = 64; /* lanes per workgroup = one wavefront */

enum StepMode : int { MODE_TOA = 0, MODE_IMU_ONLY = 1, MODE_FUSED = 2 };

/* Component-major arrays are addressed as (wave-uniform row base) + (32-bit lane offset): the row base
 * stays in SGPRs (global_load ... v_off, s[base]) and one VGPR serves every array, instead of a 64-bit
 * per-lane address kept alive for each of the 30-60 rows between the loads and the final stores. */
template <typename REAL>
__device__ inline double ldrow(const void *p, size_t row, size_t T, uint32_t t) {
    return (double)(((const REAL *)p) + row * T)[t];
}
template <typename REAL>
__device__ inline void strow(void *p, size_t row, size_t T, uint32_t t, double v) {
    (((REAL *)p) + row * T)[t] = (REAL)v;
}

/* ---- the covariance in HBM: [entries][tag] of double / float, or KFPOS_STORE_P48 ----
 * P48 (kfpos_p48.h): sign, 8 exponent bits, 39 mantissa bits -- the single that truncates the value plus the next 16
 * mantissa bits; 9e-13 relative, 6 bytes per entry. Two planes so that both stay coalesced: [entries][T] uint32 followed
 * by [entries][T] uint16; a load is two loads, a conversion and a shift-or. (The 9-state filter amplifies a 24-bit
 * covariance to 1.6e-6 m over 100 epochs of the BASELINE trace, above the 1e-6 m bar.) */
struct p48 {};
template <typename REAL>
constexpr bool cov_is_rounded() { return !std::is_same<REAL, double>::value; }
/* what the storage type keeps of a value: applied between the epochs of a multi-epoch launch, so that it computes what
 * as many single-epoch launches would (KFPOS_STORE_P48: kfpos_p48.h -- three fp64 operations per entry) */
template <typename REAL>
__device__ inline double round_cov(double v) {
    if constexpr (std::is_same<REAL, p48>::value) return kfpos_p48_round(v);
    else return (double)(REAL)v;
}
template <typename REAL>
__device__ inline double ldcov(const void *p, size_t row, size_t rows, size_t T, uint32_t t) {
    if constexpr (std::is_same<REAL, p48>::value) { /* two coalesced planes: [rows][T] uint32, then [rows][T] uint16 */
        const uint32_t hi = (((const uint32_t *)p) + row * T)[t];
        const uint32_t lo = (((const uint16_t *)(((const uint32_t *)p) + rows * T)) + row * T)[t];
        return kfpos_p48_decode(hi, lo);
    } else {
        (void)rows;
        return (double)(((const REAL *)p) + row * T)[t];
    }
}
template <typename REAL>
__device__ inline void stcov(void *p, size_t row, size_t rows, size_t T, uint32_t t, double v) {
    if constexpr (std::is_same<REAL, p48>::value) {
        uint32_t hi;
        uint16_t lo;
        kfpos_p48_encode(kfpos_p48_round(v), &hi, &lo);
        (((uint32_t *)p) + row * T)[t] = hi;
        (((uint16_t *)(((uint32_t *)p) + rows * T)) + row * T)[t] = lo;
    } else {
        (void)rows;
        (((REAL *)p) + row * T)[t] = (REAL)v;
    }
}
/* one kernel per storage mode: F<covariance type, measurement type>(args) */
#define KFPOS_BY_STORAGE(st, F, ...)                                                                          \
    ((st) == KFPOS_STORE_F32 ? F<float, float>(__VA_ARGS__)                                                   \
     : (st) == KFPOS_STORE_MIXED ? F<double, float>(__VA_ARGS__)                                              \
     : (st) == KFPOS_STORE_P48 ? F<p48, float>(__VA_ARGS__) : F<double, double>(__VA_ARGS__))

template <class P>
__device__ inline P make_params_of(const KArgs &a) {
    P pr;
    pr.anchors = a.anchors;
    pr.n_anchors = a.A;
    pr.accel_noise = a.accel_noise;
    pr.jolt = a.jolt;
    pr.cost_threshold = a.cost_threshold;
    pr.ignore_worst = a.ignore_worst;
    pr.top_n = a.top_n;
    pr.ml_variant = a.ml_variant;
    pr.use_init_pos = a.use_init_pos;
    pr.use_fixed_height = a.use_fixed_height;
    pr.imu_fixed_cov_acc = a.imu_fixed_cov_acc;
    pr.imu_fixed_cov_w = a.imu_fixed_cov_w;
    pr.px4_height = a.px4_height;
    pr.px4_arm_p1 = a.px4_arm_p1;
    pr.px4_arm_p2 = a.px4_arm_p2;
    pr.px4_cov_vel = a.px4_cov_vel;
    pr.px4_cov_gyro_z = a.px4_cov_gyro_z;
    pr.imu_cov_acc = a.imu_cov_acc;
    pr.imu_cov_w = a.imu_cov_w;
    pr.mag_offset = a.mag_offset;
    pr.mag_cov = a.mag_cov;
    return pr;
}
__device__ inline Params make_params(const KArgs &a) { return make_params_of<Params>(a); }

/* Raw epoch of one tag as it sits in HBM: fetched one epoch ahead in multi-epoch launches, so its
 * latency hides behind the previous epoch's arithmetic. */
template <typename MREAL, int AS>
struct RawEpoch {
    int32_t mm[AS];
    MREAL e[AS];
};
template <typename MREAL, int AS>
__device__ inline void fetch_epoch(const KArgs &a, size_t t, int s, RawEpoch<MREAL, AS> &raw) {
    const int32_t *rp = a.ranges + (size_t)s * a.stride_ranges;
    const MREAL *ep = (const MREAL *)a.err + (size_t)s * a.stride_err;
#pragma unroll
    for (int k = 0; k < AS; ++k) { /* all loads first: one latency, not AS of them */
        raw.mm[k] = (rp + (size_t)k * a.T)[(uint32_t)t];
        raw.e[k] = (ep + (size_t)k * a.T)[(uint32_t)t];
    }
}
template <typename MREAL, int AS>
__device__ inline void unpack_epoch(const RawEpoch<MREAL, AS> &raw, RegScratch<AS> &sc) {
#pragma unroll
    for (int k = 0; k < AS; ++k) {
        sc.r[k] = raw.mm[k] > 0 ? kf_mm_to_m(raw.mm[k]) : 0.0; /* Posgenerator.cpp:483-484 */
        sc.e[k] = (double)raw.e[k];
        sc.w[k] = 0.0;
    }
}
/* generic anchor count: epoch s -> per-lane LDS scratch */
template <typename MREAL>
__device__ inline Scratch stage_epoch_lds(const KArgs &a, double *lds, int lane, size_t t, int s) {
    Scratch sc;
    sc.r = lds + lane;
    sc.e = lds + (size_t)a.A * WAVE + lane;
    sc.w = lds + 2 * (size_t)a.A * WAVE + lane;
    sc.stride = WAVE;
    const int32_t *rp = a.ranges + (size_t)s * a.stride_ranges;
    const MREAL *ep = (const MREAL *)a.err + (size_t)s * a.stride_err;
    for (int k = 0; k < a.A; ++k) {
        const int32_t mm = (rp + (size_t)k * a.T)[(uint32_t)t];
        sc.r[k * WAVE] = mm > 0 ? kf_mm_to_m(mm) : 0.0;
        sc.e[k * WAVE] = (double)(ep + (size_t)k * a.T)[(uint32_t)t];
    }
    return sc;
}
/* anchor count known at compile time: all 2 N loads first, then the conversions and the LDS stores */
template <typename MREAL, int N>
__device__ inline StaticScratch<N> stage_epoch_lds_n(const KArgs &a, double *lds, int lane, size_t t, int s) {
    StaticScratch<N> sc;
    sc.r = lds + lane;
    sc.e = lds + (size_t)N * WAVE + lane;
    sc.w = lds + 2 * (size_t)N * WAVE + lane;
    sc.stride = WAVE;
    RawEpoch<MREAL, N> raw;
    fetch_epoch<MREAL, N>(a, t, s, raw);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        sc.r[k * WAVE] = raw.mm[k] > 0 ? kf_mm_to_m(raw.mm[k]) : 0.0;
        sc.e[k * WAVE] = (double)raw.e[k];
    }
    return sc;
}
/* A wave-uniform epoch index the optimiser cannot see through: addresses derived from it are formed anew in every
 * epoch (a few scalar instructions) instead of living as two dozen running row pointers across the whole epoch loop,
 * where they exhaust the scalar registers and end up as spilled 64-bit per-lane addresses. */
__device__ inline int opaque_uniform(int v) {
    asm volatile("" : "+s"(v));
    return v;
}
/* the same for the lane's tag index: its per-array 64-bit addresses are then formed where they are used instead of
 * being carried (spilled) across the epoch loop */
__device__ inline size_t opaque_lane(size_t t) {
    uint32_t v = (uint32_t)t;
    asm volatile("" : "+v"(v));
    return v;
}
/* the same with 4-byte errorEstimations kept as they are: LDS = r [N][lane] f64 | w [N][lane] f64 | e [N][lane] f32 */
template <int N>
__device__ inline StaticScratchF<N> stage_epoch_lds_nf(const KArgs &a, double *lds, int lane, size_t t, int s) {
    StaticScratchF<N> sc;
    sc.r = lds + lane;
    sc.w = lds + (size_t)N * WAVE + lane;
    sc.e = (float *)(lds + 2 * (size_t)N * WAVE) + lane;
    sc.stride = WAVE;
    RawEpoch<float, N> raw;
    fetch_epoch<float, N>(a, t, s, raw);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        sc.r[k * WAVE] = raw.mm[k] > 0 ? kf_mm_to_m(raw.mm[k]) : 0.0;
        sc.e[k * WAVE] = raw.e[k];
    }
    return sc;
}
/* doubles of LDS the epoch of a compile-time-count kernel takes per workgroup */
template <typename MREAL, int N>
constexpr size_t static_epoch_doubles() { return sizeof(MREAL) == 4 ? (size_t)N * WAVE * 5 / 2 : (size_t)N * WAVE * 3; }
__device__ inline double epoch_dt(const KArgs &a, size_t t, int s) {
    return a.n_steps > 1 ? a.dt_steps[s] : (a.dt ? a.dt[(uint32_t)t] : a.dt_shared);
}

/* a lane that sits a call out (dt < 0, or a dropped PX4Flow sample) still reports where its tag is: the trajectory /
 * pose output of the call carries the untouched position (what getPose at timeLag 0 would return) */
__device__ inline void skipped_lane(const KArgs &a, size_t t, bool write) {
    if (!write) return;
    if (a.status) a.status[t] = ST_SKIPPED;
    if (a.traj) {
#pragma unroll
        for (int k = 0; k < 3; ++k) (a.traj + (size_t)k * a.T)[(uint32_t)t] = (a.pos + (size_t)k * a.T)[(uint32_t)t];
    }
}

} // namespace
#endif /* KFPOS_KERNELS_H */

/*
 * kfpos_hip.hip -- host side of libkfpos_hip.so: the handle, launch selection, the three ways in (synchronous
 * host-buffer API, streaming slots, device-buffer API) and the C ABI of include/kfpos.h. The kernels live in
 * kfpos_k_*.hip (kfpos_kernels.h says which is where); the RCCL pose gather in kfpos_comm.hip.
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "kfpos_kernels.h"

std::string &kfpos_error_text() {
    thread_local std::string text;
    return text;
}

namespace {

/* ------------------------------------------------------------------ host side */
#define g_err (kfpos_error_text())

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                            \
            return KFPOS_ERR_HIP;                                                                 \
        }                                                                                         \
    } while (0)

} // namespace


namespace {

size_t lds_bytes(const kfpos_handle *h) { return (size_t)3 * h->cfg.max_anchors * WAVE * sizeof(double); }
size_t park_bytes() { return (size_t)36 * WAVE * sizeof(double); } /* 36 doubles per lane: CovSpill8 (planar), Pinv6 (6-state, full) */
size_t park9_bytes() { return (size_t)78 * WAVE * sizeof(double) + 24 * sizeof(double); } /* CovPark9: the 9-state covariance (45) and B^-1 (21) during the gain iteration, the accelerometer whitener and Sigma^-1 (12), a lane-addressable copy of 8 anchors: 39.2 KB per wavefront, four wavefronts per CU */

void fill_args(const kfpos_handle *h, KArgs &a) {
    std::memcpy(a.anchors, h->anchors, sizeof(a.anchors));
    a.T = h->cfg.n_tags;
    a.A = h->have_anchors ? h->A : 0; /* anchors set (<= max_anchors): columns beyond them do not exist */
    a.accel_noise = h->cfg.accel_noise;
    a.jolt = h->cfg.jolt;
    a.cost_threshold = h->cfg.cost_threshold;
    a.ignore_worst = h->cfg.ignore_worst;
    a.top_n = h->cfg.top_n;
    a.ml_variant = h->cfg.model == KFPOS_MODEL_ML ? h->cfg.ml_variant : 0;
    a.pair9 = h->pair9 ? 1 : 0;
    a.use_init_pos = h->cfg.use_init_pos;
    a.use_fixed_height = h->planar.use_fixed_height;
    a.imu_fixed_cov_acc = h->planar.imu_use_fixed_cov_acc;
    a.imu_fixed_cov_w = h->planar.imu_use_fixed_cov_ang_vel_z;
    a.px4_height = h->planar.px4_height;
    a.px4_arm_p1 = h->planar.px4_arm_p1;
    a.px4_arm_p2 = h->planar.px4_arm_p2;
    a.px4_cov_vel = h->planar.px4_cov_velocity;
    a.px4_cov_gyro_z = h->planar.px4_cov_gyro_z;
    a.imu_cov_acc = h->planar.imu_cov_acc;
    a.imu_cov_w = h->planar.imu_cov_ang_vel_z;
    a.mag_offset = h->planar.mag_angle_offset;
    a.mag_cov = h->planar.mag_cov;
    a.platch = h->d_latch;
    a.sensor = nullptr;
    a.pos = h->d_pos;
    a.vel = h->d_vel;
    a.P = h->d_P;
    a.flags = h->d_flags;
    a.imu_acc = h->d_imu_acc;
    a.imu_cov = h->d_imu_cov;
    a.ranges = nullptr;
    a.err = nullptr;
    a.dt = nullptr;
    a.dt_shared = 0.0;
    a.accel = nullptr;
    a.cov = nullptr;
    a.mode = MODE_TOA;
    a.latch = 1;
    a.status = nullptr;
    a.n_steps = 1;
    a.stride_ranges = a.stride_err = a.stride_accel = a.stride_cov = 0;
    a.traj = nullptr;
}


/* Anchor-count specialisation. 8 anchors (BASELINE configs 2-4): epoch in registers (RegScratch), returns 8. 16 anchors
 * (config 5, 6-state): a register-resident epoch costs 96 more live registers and spills to scratch, which is no faster
 * than the run-time loop (measured in round 1), so the epoch stays in LDS and only the anchor loops are compile-time, in
 * groups of 8 (StaticScratch): returns -16. Everything else: 0, the run-time loop.
 * "No scratch" is a PERFORMANCE rule of this library (checked at build time), not a correctness crutch: round 1 saw
 * wrong, run-to-run varying positions from a work-in-progress build of the spilling 16-anchor kernel and dropped it
 * without a diagnosis. Round 2 could not reproduce that with the committed sources of either round -- the library as it
 * stood before that commit and today's kernels with the register-resident 16-anchor instantiations re-enabled
 * (236-960 bytes/lane of scratch) match the oracle and repeat bit for bit on full, partially filled and
 * skipped-lane wavefronts (tools/exp/, profiles/r02b_*) -- and a MemorySanitizer build of the kernel body with every
 * undefined input poisoned is clean (tests/emu/msan_audit.sh). DESIGN.md section 2. */
/* the plain 8-anchor 6-state bank with more wavefronts than SIMDs: the two-wavefronts-per-SIMD build (k_step_toa6_w2) */
bool toa6_two_waves(const kfpos_handle *h) {
    return h->two_waves && h->cfg.model == KFPOS_MODEL_TOA && !h->full && !h->force_generic && !h->coop &&
           h->cfg.max_anchors == 8 && h->cfg.ignore_worst == 0 && h->cfg.top_n == 0;
}
int static_anchors(const kfpos_handle *h) {
    if (h->cfg.max_anchors == 8) return 8;
    /* 16 anchors (BASELINE config 5): compile-time loops over an LDS-resident epoch (StaticScratch), 6-state only */
    if (h->cfg.max_anchors == 16 && h->cfg.model == KFPOS_MODEL_TOA) return -16;
    return 0;
}

step_kernel_t step_kernel(const kfpos_handle *h, bool sensor_call = false, bool imu_only = false) {
    const int st = h->cfg.storage, as = (h->force_generic || sensor_call) ? 0 : static_anchors(h);
    if (h->cfg.model == KFPOS_MODEL_PLANAR) return kfpos_k::planar_kernel(st, h->planar_sensors || sensor_call, as);
    if (h->cfg.model == KFPOS_MODEL_ML) return kfpos_k::ml_kernel(st, as);
    if (h->coop) return kfpos_k::toa6_coop_kernel(st);
    if (h->cfg.model == KFPOS_MODEL_TOA) {
        /* heur: 0 = the bank has no outlier heuristic (BASELINE configs 2 and 4), 1 = top-N only (config 5), 2 =
         * leave-one-out (with or without top-N) */
        const int heur = h->cfg.ignore_worst != 0 ? 2 : (h->cfg.top_n != 0 ? 1 : 0);
        if (h->full) return kfpos_k::toa6_full_kernel(st, as, heur);
        /* more wavefronts than SIMDs: the 256-register build of the plain kernel */
        return kfpos_k::toa6_sym_kernel(st, as, heur, toa6_two_waves(h));
    }
    return kfpos_k::imu9_kernel(st, as, !imu_only);
}

int launch_step(kfpos_handle *h, const KArgs &a, hipStream_t s) {
    if (h->coop) {
        const int groups = (h->cfg.n_tags + COOP_TAGS_PER_WAVE - 1) / COOP_TAGS_PER_WAVE;
        hipLaunchKernelGGL(step_kernel(h), dim3(groups), dim3(WAVE), 0, s, a);
        HIPCHK(hipGetLastError());
        h->stepped = true;
        return KFPOS_OK;
    }
    const int blocks = (h->cfg.n_tags + WAVE - 1) / WAVE;
    /* does the selected kernel stage the epoch in LDS? */
    const bool generic = h->force_generic || static_anchors(h) <= 0 || (h->cfg.model == KFPOS_MODEL_TOA && h->full) ||
                         (h->cfg.model == KFPOS_MODEL_PLANAR && h->planar_sensors) || toa6_two_waves(h);
    const bool planar_sensor = h->cfg.model == KFPOS_MODEL_PLANAR && a.mode != 0;
    size_t lds = (a.mode == MODE_IMU_ONLY || planar_sensor || !generic) ? 0 : lds_bytes(h);
    /* the 6-state compile-time-count kernels keep 4-byte errorEstimations as they are: 20 bytes per anchor and lane */
    if (lds && h->cfg.model == KFPOS_MODEL_TOA && !h->force_generic && static_anchors(h) != 0 && h->msz == 4)
        lds = lds * 5 / 6;
    if (h->cfg.model == KFPOS_MODEL_PLANAR && (h->planar_sensors || planar_sensor)) lds += park_bytes();
    if (h->cfg.model == KFPOS_MODEL_TOA && h->full) lds += park_bytes(); /* Pinv6 of the non-symmetric layout */
    if (toa6_two_waves(h)) lds += (size_t)(h->msz == 4 ? 19 : 15) * WAVE * sizeof(double); /* covariance entries parked during the ML solve */
    if (h->cfg.model == KFPOS_MODEL_TOA_IMU) lds += park9_bytes();
    hipLaunchKernelGGL(step_kernel(h, planar_sensor, a.mode == MODE_IMU_ONLY && h->cfg.model == KFPOS_MODEL_TOA_IMU), dim3(blocks), dim3(WAVE), lds, s, a);
    HIPCHK(hipGetLastError());
    h->stepped = true;
    return KFPOS_OK;
}

/* a region of the row-major staging area; regions live until the next stage_reset() (start of an API call) */
int stage_region(kfpos_handle *h, size_t bytes, void **out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (h->stage_used + bytes > h->stage_cap) {
        /* grow: everything queued so far still reads the old buffer, so drain first */
        HIPCHK(hipDeviceSynchronize());
        const size_t cap = (h->stage_used + bytes) * 2;
        unsigned char *nb = nullptr;
        HIPCHK(hipMalloc((void **)&nb, cap));
        if (h->d_stage) (void)hipFree(h->d_stage);
        h->d_stage = nb;
        h->stage_cap = cap;
        h->stage_used = 0; /* regions handed out before the growth have been consumed (synchronised above) */
    }
    *out = h->d_stage + h->stage_used;
    h->stage_used += bytes;
    return KFPOS_OK;
}
void stage_reset(kfpos_handle *h) { h->stage_used = 0; }

/* host row-major [T][C] -> component-major [C][T] of `esz`-byte elements: one copy + a device-side turn */
int stage_in(kfpos_handle *h, void *dst, const void *src, int C, size_t esz) {
    const size_t T = h->cfg.n_tags;
    if (C == 1) {
        HIPCHK(hipMemcpy(dst, src, T * esz, hipMemcpyHostToDevice));
        return KFPOS_OK;
    }
    void *rows = nullptr;
    const int rc = stage_region(h, T * C * esz, &rows);
    if (rc) return rc;
    HIPCHK(hipMemcpy(rows, src, T * C * esz, hipMemcpyHostToDevice));
    kfpos_k::launch_rows_to_cols(esz, nullptr, rows, dst, (int)T, C);
    HIPCHK(hipGetLastError());
    return KFPOS_OK;
}
/* device [C][T] doubles -> host row-major [T][C] */
int stage_out(kfpos_handle *h, double *dst, const double *dsrc, int C) {
    const size_t T = h->cfg.n_tags;
    if (C == 1) {
        HIPCHK(hipMemcpy(dst, dsrc, T * sizeof(double), hipMemcpyDeviceToHost));
        return KFPOS_OK;
    }
    void *rows = nullptr;
    const int rc = stage_region(h, T * C * sizeof(double), &rows);
    if (rc) return rc;
    kfpos_k::launch_cols_to_rows(nullptr, dsrc, (double *)rows, (int)T, C);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(dst, rows, T * C * sizeof(double), hipMemcpyDeviceToHost));
    return KFPOS_OK;
}

int stage_dt(kfpos_handle *h, const double *dt, int32_t dt_len, const double **d_dt, double *shared) {
    if (!dt || (dt_len != 1 && dt_len != h->cfg.n_tags)) return KFPOS_ERR_ARG;
    if (dt_len == 1 && h->cfg.n_tags != 1) {
        *d_dt = nullptr;
        *shared = dt[0];
        return KFPOS_OK;
    }
    HIPCHK(hipMemcpy(h->d_dt, dt, sizeof(double) * h->cfg.n_tags, hipMemcpyHostToDevice));
    *d_dt = h->d_dt;
    *shared = dt[0];
    return KFPOS_OK;
}

int fetch_status(kfpos_handle *h, uint32_t *status) {
    HIPCHK(hipDeviceSynchronize());
    if (status) HIPCHK(hipMemcpy(status, h->d_status, sizeof(uint32_t) * h->cfg.n_tags, hipMemcpyDeviceToHost));
    return KFPOS_OK;
}


/* ---- small banks: inputs and outputs in one host-pinned, device-mapped block (kfpos_handle::sm_*) ---- */
size_t small_bank_limit() { /* elements of the range matrix (n_tags x max_anchors) up to which a bank is "small" */
    const char *e = getenv("KFPOS_SMALL_BANK_ELEMS");
    return e ? (size_t)atol(e) : 4096;
}
template <typename E>
void rows_to_cols_host(void *dst, const void *src, size_t T, int C) {
    const E *s = (const E *)src;
    E *d = (E *)dst;
    for (size_t t = 0; t < T; ++t)
        for (int c = 0; c < C; ++c) d[(size_t)c * T + t] = s[t * C + c];
}
/* host row-major [T][C] -> component-major [C][T] inside the mapped block */
void small_in(kfpos_handle *h, size_t off, const void *src, int C, size_t esz) {
    if (esz == 4) rows_to_cols_host<uint32_t>(h->sm_h + off, src, h->cfg.n_tags, C);
    else rows_to_cols_host<uint64_t>(h->sm_h + off, src, h->cfg.n_tags, C);
}
/* component-major [C][T] doubles in the mapped block -> host row-major [T][C] */
void small_out(const kfpos_handle *h, double *dst, size_t off_doubles, int C) {
    const size_t T = h->cfg.n_tags;
    const double *s = (const double *)(h->sm_h + h->sm_out) + off_doubles;
    for (size_t t = 0; t < T; ++t)
        for (int c = 0; c < C; ++c) dst[t * C + c] = s[(size_t)c * T + t];
}
int small_dt(kfpos_handle *h, const double *dt, int32_t dt_len, const double **d_dt, double *shared) {
    if (!dt || (dt_len != 1 && dt_len != h->cfg.n_tags)) return KFPOS_ERR_ARG;
    *shared = dt[0];
    *d_dt = nullptr;
    if (dt_len != 1) {
        std::memcpy(h->sm_h + h->sm_dt, dt, sizeof(double) * h->cfg.n_tags);
        *d_dt = (const double *)(h->sm_d + h->sm_dt);
    }
    return KFPOS_OK;
}
int small_finish(kfpos_handle *h, uint32_t *status) {
    HIPCHK(hipStreamSynchronize(nullptr)); /* the kernel's writes into the mapped block are visible now */
    if (status) std::memcpy(status, h->sm_h + h->sm_status, sizeof(uint32_t) * h->cfg.n_tags);
    return KFPOS_OK;
}

/* ---- streaming host API: wait for whatever the slots still have in flight ---- */
int drain_slots(kfpos_handle *h) {
    for (auto &sl : h->slot)
        if (sl.busy) {
            HIPCHK(hipEventSynchronize(sl.done));
            sl.busy = false;
        }
    return KFPOS_OK;
}

using DevScope = KfposDevScope; /* kfpos_internal.h */

/* packed index of the stored covariance entry (i, j) */
inline int pidx(const kfpos_handle *h, int i, int j) {
    if (h->full) return i * h->n + j;
    const int N = h->n;
    if (i > j) { const int tt = i; i = j; j = tt; }
    return i * N - i * (i - 1) / 2 + (j - i);
}

} // namespace

extern "C" {

const char *kfpos_last_error(void) { return g_err.c_str(); }
const char *kfpos_strerror(int code) {
    switch (code) {
    case KFPOS_OK: return "ok";
    case KFPOS_ERR_ARG: return "invalid argument";
    case KFPOS_ERR_HIP: return "HIP runtime error";
    case KFPOS_ERR_NO_DEVICE: return "no usable GPU";
    case KFPOS_ERR_MODEL: return "call not defined for this model";
    case KFPOS_ERR_STATE: return "call out of sequence";
    case KFPOS_ERR_COMM: return "RCCL error";
    default: return "unknown error";
    }
}
int kfpos_version(void) { return KFPOS_VERSION; }

int kfpos_create(const kfpos_config *cfg, kfpos_handle **out) {
    if (!cfg || !out) return KFPOS_ERR_ARG;
    *out = nullptr;
    g_err.clear();
    auto bad = [](const char *why) {
        g_err = why;
        return KFPOS_ERR_ARG;
    };
    if (cfg->model != KFPOS_MODEL_TOA && cfg->model != KFPOS_MODEL_TOA_IMU && cfg->model != KFPOS_MODEL_ML &&
        cfg->model != KFPOS_MODEL_PLANAR)
        return bad("kfpos_config.model is not one of KFPOS_MODEL_*");
    if (cfg->storage != KFPOS_STORE_F64 && cfg->storage != KFPOS_STORE_F32 && cfg->storage != KFPOS_STORE_MIXED &&
        cfg->storage != KFPOS_STORE_P48)
        return bad("kfpos_config.storage is not one of KFPOS_STORE_*");
    if (cfg->n_tags < 1) return bad("kfpos_config.n_tags must be >= 1");
    if (cfg->max_anchors < 1 || cfg->max_anchors > KFPOS_MAX_ANCHORS)
        return bad("kfpos_config.max_anchors must be 1..64 (MAX_NUM_ANCS, Posgenerator.h:74)");
    if (cfg->top_n < 0) return bad("kfpos_config.top_n must be >= 0");
    if (cfg->ml_variant != KFPOS_ML_NORMAL && cfg->ml_variant != KFPOS_ML_IGNORE_N && cfg->ml_variant != KFPOS_ML_BEST)
        return bad("kfpos_config.ml_variant is not one of KFPOS_ML_*");
    if (cfg->ml_variant != KFPOS_ML_NORMAL && cfg->model != KFPOS_MODEL_ML)
        return bad("kfpos_config.ml_variant belongs to KFPOS_MODEL_ML (MLLocation's variants, MLLocation.h:5-7)");
    if ((cfg->model == KFPOS_MODEL_TOA_IMU && (cfg->top_n || cfg->ignore_worst)) ||
        (cfg->model == KFPOS_MODEL_ML && cfg->ignore_worst) ||
        (cfg->model == KFPOS_MODEL_PLANAR && (cfg->top_n || cfg->ignore_worst)))
        return bad("ignore_worst exists for the 6-state filter only, top_n for the 6-state filter and the ML estimator");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
        g_err = "no HIP device";
        return KFPOS_ERR_NO_DEVICE;
    }
    DevScope dev_(cfg->device); /* the caller's current device is left as it was */
    kfpos_handle *h = new (std::nothrow) kfpos_handle();
    if (!h) return KFPOS_ERR_ARG;
    h->cfg = *cfg;
    h->n = cfg->model == KFPOS_MODEL_TOA_IMU ? 9 : (cfg->model == KFPOS_MODEL_ML ? 3 : (cfg->model == KFPOS_MODEL_PLANAR ? 8 : 6));
    h->full = (cfg->model == KFPOS_MODEL_TOA && !cfg->use_init_pos) ? 1 : 0;
    h->psz = h->full ? h->n * h->n : h->n * (h->n + 1) / 2;
    h->rsz = cfg->storage == KFPOS_STORE_F32 ? 4 : (cfg->storage == KFPOS_STORE_P48 ? 6 : 8);
    h->msz = cfg->storage == KFPOS_STORE_F64 ? 8 : 4;
    h->A = 0;
    h->have_anchors = false;
    h->stepped = false;
    {
        const char *g = getenv("KFPOS_GENERIC_KERNEL");
        h->force_generic = g && g[0] == '1';
        /* one combination has no anchor-count-specialised kernel: ML start (36-entry covariance + its SVD scratchpad) with
         * leave-one-out and the 48-bit covariance -- the LDS-resident 8-anchor instantiation would touch scratch memory
         * inside its epoch loop (make check), the run-time-loop kernel does not */
        if (h->full && cfg->ignore_worst && cfg->storage == KFPOS_STORE_P48) h->force_generic = true;
        {
            hipDeviceProp_t prop;
            const char *ow = getenv("KFPOS_ONE_WAVE_BUILD");
            h->two_waves = false;
            if (!(ow && ow[0] == '1') && hipGetDeviceProperties(&prop, cfg->device) == hipSuccess)
                h->two_waves = ((size_t)cfg->n_tags + WAVE - 1) / WAVE > (size_t)prop.multiProcessorCount * 4;
        }
        const char *np = getenv("KFPOS_PAIR9");
        h->pair9 = np && np[0] == '1';
        const char *nc = getenv("KFPOS_NO_COOP");
        /* up to 8 192 tags (1024 groups-of-8 wavefronts = one per SIMD) 8 lanes per tag pay off: 4.4-5.0 us per epoch
         * against 7.5 us; at 16 384 the one-tag-per-lane grid wins again (7.7 vs 8.3 us, measured) */
        h->coop = !(nc && nc[0] == '1') && !h->force_generic && cfg->model == KFPOS_MODEL_TOA && cfg->use_init_pos &&
                  !cfg->ignore_worst && !cfg->top_n && cfg->max_anchors <= COOP_LANES && cfg->n_tags <= 8192;
        const char *c = getenv("KFPOS_TRACE_CHUNK_STEPS");
        int n = c ? atoi(c) : KFPOS_TRACE_CHUNK;
        h->trace_chunk = n < 1 ? 1 : (n > KFPOS_TRACE_CHUNK ? KFPOS_TRACE_CHUNK : n);
    }
    std::memset(h->anchors, 0, sizeof(h->anchors));
    const size_t T = cfg->n_tags, A = cfg->max_anchors, r = h->rsz, m = h->msz;
#define ALLOC(ptr, bytes)                                                     \
    do {                                                                      \
        hipError_t e_ = hipMalloc((void **)&(ptr), (bytes));                  \
        if (e_ != hipSuccess) {                                               \
            g_err = std::string("hipMalloc: ") + hipGetErrorString(e_);       \
            kfpos_destroy(h);                                                 \
            return KFPOS_ERR_HIP;                                             \
        }                                                                     \
        (void)hipMemset((ptr), 0, (bytes));                                   \
    } while (0)
    ALLOC(h->d_pos, 3 * T * sizeof(double));
    ALLOC(h->d_P, h->psz * T * r);
    ALLOC(h->d_flags, T * sizeof(uint32_t));
    if (h->n == 3) ALLOC(h->d_vel, 3 * T * sizeof(double)); /* ALGORITHM_ML: the solver's per-tag seed */
    if (h->n == 8) {
        ALLOC(h->d_vel, 4 * T * sizeof(double)); /* vx, vy, theta, omega */
        ALLOC(h->d_latch, LATCH_ROWS * T * sizeof(double));
        ALLOC(h->d_sensor, 24 * T * sizeof(double));
    }
    if (h->n == 9) {
        ALLOC(h->d_vel, 3 * T * sizeof(double));
        ALLOC(h->d_imu_acc, 3 * T * m);
        ALLOC(h->d_imu_cov, 6 * T * m);
        ALLOC(h->d_accel, 3 * T * m);
        ALLOC(h->d_cov, 9 * T * m);
    }
    ALLOC(h->d_ranges, A * T * sizeof(int32_t));
    ALLOC(h->d_err, A * T * m);
    ALLOC(h->d_dt, T * sizeof(double));
    ALLOC(h->d_out, 15 * T * sizeof(double));
    ALLOC(h->d_status, T * sizeof(uint32_t));
#undef ALLOC
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
        g_err = "hipEventCreate failed";
        kfpos_destroy(h);
        return KFPOS_ERR_HIP;
    }
    if (T * A <= small_bank_limit()) { /* small bank: one mapped block instead of staging copies (kfpos_handle::sm_*) */
        size_t off = 0;
        auto region = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
        h->sm_ranges = region(A * T * sizeof(int32_t));
        h->sm_err = region(A * T * m);
        h->sm_accel = region(3 * T * m);
        h->sm_cov = region(9 * T * m);
        h->sm_dt = region(T * sizeof(double));
        h->sm_sensor = region(24 * T * sizeof(double));
        h->sm_status = region(T * sizeof(uint32_t));
        h->sm_out = region((size_t)(15 + h->n + h->n * h->n) * T * sizeof(double));
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, off, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
            g_err = "hipHostMalloc(mapped block of a small bank) failed";
            if (hp) (void)hipHostFree(hp);
            kfpos_destroy(h);
            return KFPOS_ERR_HIP;
        }
        std::memset(hp, 0, off);
        h->sm_h = (unsigned char *)hp;
        h->sm_d = (unsigned char *)dp;
    }
    const bool parks = cfg->model == KFPOS_MODEL_PLANAR || (cfg->model == KFPOS_MODEL_TOA && h->full);
    if (cfg->model == KFPOS_MODEL_TOA_IMU && lds_bytes(h) + park9_bytes() > 64 * 1024) {
        hipError_t e_ = hipFuncSetAttribute((const void *)step_kernel(h), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)(lds_bytes(h) + park9_bytes()));
        if (e_ != hipSuccess) {
            g_err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e_);
            kfpos_destroy(h);
            return KFPOS_ERR_HIP;
        }
    } else if (lds_bytes(h) + (parks ? park_bytes() : 0) > 64 * 1024) {
        hipError_t e_ = hipFuncSetAttribute((const void *)step_kernel(h), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)(lds_bytes(h) + (h->full ? park_bytes() : 0)));
        if (e_ == hipSuccess && cfg->model == KFPOS_MODEL_PLANAR) { /* the instantiation ranging epochs switch to */
            h->planar_sensors = true;
            e_ = hipFuncSetAttribute((const void *)step_kernel(h), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(lds_bytes(h) + park_bytes()));
            h->planar_sensors = false;
        }
        if (e_ != hipSuccess) {
            g_err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e_);
            kfpos_destroy(h);
            return KFPOS_ERR_HIP;
        }
    }
    /* initial position: fixed start (P0 = 0) or NaN until the ML initialisation */
    std::vector<double> p0(3 * T);
    for (size_t t = 0; t < T; ++t)
        for (int k = 0; k < 3; ++k) {
            p0[(size_t)k * T + t] = cfg->use_init_pos ? cfg->init_pos[k] : NAN;
            if (h->n == 8 && k == 2) p0[(size_t)k * T + t] = 0.0; /* mUWBtagZ: kfpos_set_planar, not initialPosition.z */
        }
    if (h->n == 3 && hipMemcpy(h->d_vel, p0.data(), p0.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        g_err = "hipMemcpy(ml seed) failed";
        kfpos_destroy(h);
        return KFPOS_ERR_HIP;
    }
    if (hipMemcpy(h->d_pos, p0.data(), p0.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        g_err = "hipMemcpy(init pos) failed";
        kfpos_destroy(h);
        return KFPOS_ERR_HIP;
    }
    *out = h;
    return KFPOS_OK;
}

int kfpos_destroy(kfpos_handle *h) {
    if (!h) return KFPOS_ERR_ARG;
    void *ptrs[] = {h->d_pos, h->d_vel, h->d_P, h->d_imu_acc, h->d_imu_cov, h->d_flags, h->d_ranges,
                    h->d_err, h->d_accel, h->d_cov, h->d_dt, h->d_out, h->d_status, h->d_latch, h->d_sensor,
                    h->d_stage};
    DevScope dev_(h->cfg.device);
    (void)hipDeviceSynchronize(); /* nothing of this handle may still be in flight (streaming slots) */
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (h->sm_h) (void)hipHostFree(h->sm_h);
    for (auto &sl : h->slot) {
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.copied) (void)hipEventDestroy(sl.copied);
        if (sl.copied2) (void)hipEventDestroy(sl.copied2);
        if (sl.computed) (void)hipEventDestroy(sl.computed);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (h->s_copy) (void)hipStreamDestroy(h->s_copy);
    if (h->s_copy2) (void)hipStreamDestroy(h->s_copy2);
    if (h->s_comp) (void)hipStreamDestroy(h->s_comp);
    if (h->s_back) (void)hipStreamDestroy(h->s_back);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    delete h;
    return KFPOS_OK;
}

int kfpos_init(kfpos_handle *h) { return h ? KFPOS_OK : KFPOS_ERR_ARG; }

int kfpos_set_anchors(kfpos_handle *h, const double *xyz, const int32_t *ids, int32_t n_anchors) {
    g_err.clear();
    (void)ids;
    if (!h || !xyz || n_anchors < 1 || n_anchors > h->cfg.max_anchors) return KFPOS_ERR_ARG;
    if (h->cfg.model == KFPOS_MODEL_ML && h->cfg.ml_variant == KFPOS_ML_BEST && n_anchors > 5) {
        g_err = "estimatePositionBestGroup has no defined result for more than 5 ranges: its erase loop removes by an index "
                "into the vector it is shrinking and runs past the end from the first group on (MLLocation.cpp:377-381)";
        return KFPOS_ERR_MODEL;
    }
    std::memset(h->anchors, 0, sizeof(h->anchors));
    std::memcpy(h->anchors, xyz, sizeof(double) * 3 * n_anchors);
    h->A = n_anchors;
    h->have_anchors = true;
    return KFPOS_OK;
}

int kfpos_set_init_positions(kfpos_handle *h, const double *xyz) {
    g_err.clear();
    if (!h || !xyz) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    if (!h->cfg.use_init_pos || h->stepped) return KFPOS_ERR_STATE;
    stage_reset(h);
    if (h->n == 3) { /* ALGORITHM_ML: the seed of every solve */
        const int rc = stage_in(h, h->d_vel, xyz, 3, sizeof(double));
        if (rc) return rc;
    }
    const int rc = stage_in(h, h->d_pos, xyz, 3, sizeof(double));
    if (rc == KFPOS_OK && h->n == 8) { /* the filter works at mUWBtagZ, whatever initialPosition.z says */
        std::vector<double> z(h->cfg.n_tags, h->planar.fixed_height);
        HIPCHK(hipMemcpy(h->d_pos + 2 * (size_t)h->cfg.n_tags, z.data(), z.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return rc;
}

int kfpos_set_planar(kfpos_handle *h, const kfpos_planar_config *cfg) {
    g_err.clear();
    if (!h || !cfg) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    if (h->cfg.model != KFPOS_MODEL_PLANAR) return KFPOS_ERR_MODEL;
    if (h->stepped) return KFPOS_ERR_STATE;
    h->planar = *cfg;
    /* mUWBtagZ and mAngle of every tag: pos row 2, vel row 2 (angular speed, row 3, stays 0) */
    const size_t T = h->cfg.n_tags;
    std::vector<double> v(T, cfg->fixed_height);
    HIPCHK(hipMemcpy(h->d_pos + 2 * T, v.data(), T * sizeof(double), hipMemcpyHostToDevice));
    v.assign(T, cfg->init_angle);
    HIPCHK(hipMemcpy(h->d_vel + 2 * T, v.data(), T * sizeof(double), hipMemcpyHostToDevice));
    return KFPOS_OK;
}

int kfpos_real_size(const kfpos_handle *h) { return h ? h->msz : 0; }
int kfpos_state_dim(const kfpos_handle *h) { return h ? h->n : 0; }

/* ---- device-buffer API ---- */
int kfpos_step_toa_dev(kfpos_handle *h, const int32_t *range_mm, const void *err_est, const double *dt,
                       double dt_shared, uint32_t *status, void *stream) {
    g_err.clear();
    if (!h || !range_mm || !err_est) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    if (!h->have_anchors) {
        g_err = "kfpos_set_anchors has not been called (the node drops ranges until the anchors are known, Posgenerator.cpp:92-96)";
        return KFPOS_ERR_STATE;
    }
    KArgs a;
    fill_args(h, a);
    a.ranges = range_mm;
    a.err = err_est;
    a.dt = dt;
    a.dt_shared = dt_shared;
    a.status = status;
    a.mode = MODE_TOA;
    return launch_step(h, a, (hipStream_t)stream);
}

int kfpos_step_imu_dev(kfpos_handle *h, const void *accel, const void *cov, const double *dt,
                       double dt_shared, uint32_t *status, void *stream) {
    g_err.clear();
    if (!h || !accel || !cov) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    if (h->cfg.model != KFPOS_MODEL_TOA_IMU) return KFPOS_OK; /* KalmanFilterTOA::newIMUMeasurement is empty */
    KArgs a;
    fill_args(h, a);
    a.accel = accel;
    a.cov = cov;
    a.dt = dt;
    a.dt_shared = dt_shared;
    a.status = status;
    a.mode = MODE_IMU_ONLY;
    a.latch = 1;
    return launch_step(h, a, (hipStream_t)stream);
}

static int sensor_width(int32_t kind) {
    switch (kind) {
    case KFPOS_SENSOR_PX4FLOW: return 5;
    case KFPOS_SENSOR_IMU: return 24;
    case KFPOS_SENSOR_MAG: return 3;
    case KFPOS_SENSOR_COMPASS: return 1;
    default: return 0;
    }
}

int kfpos_step_sensor_dev(kfpos_handle *h, int32_t kind, const double *data, const double *dt, double dt_shared,
                          uint32_t *status, void *stream) {
    g_err.clear();
    if (!h || !data || sensor_width(kind) == 0) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    if (h->cfg.model != KFPOS_MODEL_PLANAR) { /* the empty virtuals of PositionEstimationAlgorithm.h:26-35 */
        if (status) HIPCHK(hipMemsetAsync(status, 0, sizeof(uint32_t) * h->cfg.n_tags, (hipStream_t)stream));
        return KFPOS_OK;
    }
    KArgs a;
    fill_args(h, a);
    a.sensor = data;
    a.dt = dt;
    a.dt_shared = dt_shared;
    a.status = status;
    a.mode = kind;
    const int rc = launch_step(h, a, (hipStream_t)stream);
    h->planar_sensors = true; /* ranging epochs now carry the latched samples */
    return rc;
}

int kfpos_step_toa_imu_dev(kfpos_handle *h, const int32_t *range_mm, const void *err_est, const void *accel,
                           const void *cov, int32_t latch, const double *dt, double dt_shared,
                           uint32_t *status, void *stream) {
    g_err.clear();
    if (!h || !range_mm || !err_est || !accel || !cov) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    if (h->cfg.model != KFPOS_MODEL_TOA_IMU) return KFPOS_ERR_MODEL;
    if (!h->have_anchors) return KFPOS_ERR_STATE;
    KArgs a;
    fill_args(h, a);
    a.ranges = range_mm;
    a.err = err_est;
    a.accel = accel;
    a.cov = cov;
    a.dt = dt;
    a.dt_shared = dt_shared;
    a.status = status;
    a.mode = MODE_FUSED;
    a.latch = latch ? 1 : 0;
    return launch_step(h, a, (hipStream_t)stream);
}

int kfpos_run_trace_dev(kfpos_handle *h, int32_t n_steps, const int32_t *range_mm, int64_t stride_ranges,
                        const void *err_est, int64_t stride_err, const void *accel, int64_t stride_accel,
                        const void *cov, int64_t stride_cov, const double *dt_steps, double *trajectory,
                        uint32_t *status, void *stream) {
    g_err.clear();
    if (!h || n_steps < 0 || !range_mm || !err_est || !dt_steps) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    if (accel && (!cov || h->cfg.model != KFPOS_MODEL_TOA_IMU)) return KFPOS_ERR_MODEL;
    if (!h->have_anchors) return KFPOS_ERR_STATE;
    KArgs a;
    fill_args(h, a);
    a.status = status;
    a.mode = accel ? MODE_FUSED : MODE_TOA;
    a.stride_ranges = stride_ranges;
    a.stride_err = stride_err;
    a.stride_accel = stride_accel;
    a.stride_cov = stride_cov;
    const size_t r = h->msz;
    /* the kernels whiten the accelerometer covariance once per launch: a covariance per epoch means one epoch per launch */
    const int chunk = (accel && stride_cov != 0) ? 1 : h->trace_chunk;
    /* Up to `chunk` epochs per launch: the kernel keeps every tag's state in registers across them, so
     * state traffic and launch boundaries are paid once per chunk instead of once per epoch. */
    for (int s0 = 0; s0 < n_steps; s0 += chunk) {
        const int n = n_steps - s0 < chunk ? n_steps - s0 : chunk;
        a.n_steps = n;
        a.ranges = range_mm + (size_t)s0 * stride_ranges;
        a.err = (const char *)err_est + (size_t)s0 * stride_err * r;
        if (accel) {
            a.accel = (const char *)accel + (size_t)s0 * stride_accel * r;
            a.cov = (const char *)cov + (size_t)s0 * stride_cov * r;
            a.latch = (s0 + n == n_steps) ? 1 : 0; /* leave the last sample latched, as n separate calls would */
        }
        for (int k = 0; k < n; ++k) a.dt_steps[k] = dt_steps[s0 + k];
        a.dt_shared = dt_steps[s0];
        a.traj = trajectory ? trajectory + (size_t)s0 * 3 * h->cfg.n_tags : nullptr;
        const int rc = launch_step(h, a, (hipStream_t)stream);
        if (rc != KFPOS_OK) return rc;
    }
    return KFPOS_OK;
}

static int launch_pose(kfpos_handle *h, double dt_ahead, const double *dt_each, double *pos, double *cov3x3,
                       double *vel, uint32_t *status, void *stream, double *full_x = nullptr,
                       double *full_P = nullptr) {
    if (!h) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    PoseArgs a;
    a.dt_each = dt_each;
    a.full_x = full_x;
    a.full_P = full_P;
    a.T = h->cfg.n_tags;
    a.model = h->cfg.model;
    a.full = h->full;
    a.accel_noise = h->cfg.accel_noise;
    a.jolt = h->cfg.jolt;
    a.dt_ahead = dt_ahead;
    a.pos_in = h->d_pos;
    a.vel_in = h->d_vel;
    a.P = h->d_P;
    a.flags = h->d_flags;
    a.pos = pos;
    a.cov = cov3x3;
    a.vel = vel;
    a.status = status;
    kfpos_k::launch_get_pose(h->cfg.model, h->full != 0, h->cfg.storage, (a.T + WAVE - 1) / WAVE, (hipStream_t)stream, a);
    HIPCHK(hipGetLastError());
    return KFPOS_OK;
}

int kfpos_get_pose_dev(kfpos_handle *h, double dt_ahead, double *pos, double *cov3x3, double *vel,
                       uint32_t *status, void *stream) {
    g_err.clear();
    return launch_pose(h, dt_ahead, nullptr, pos, cov3x3, vel, status, stream);
}

/* ---- host-buffer API ---- */
int kfpos_step_toa(kfpos_handle *h, const int32_t *range_mm, const void *err_est, const double *dt,
                   int32_t dt_len, uint32_t *status) {
    g_err.clear();
    if (!h || !range_mm || !err_est) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    if (!h->have_anchors) {
        g_err = "kfpos_set_anchors has not been called (the node drops ranges until the anchors are known, Posgenerator.cpp:92-96)";
        return KFPOS_ERR_STATE;
    }
    int rc = drain_slots(h);
    if (rc) return rc;
    const double *d_dt;
    double shared;
    if (h->sm_h) {
        if ((rc = small_dt(h, dt, dt_len, &d_dt, &shared))) return rc;
        small_in(h, h->sm_ranges, range_mm, h->cfg.max_anchors, sizeof(int32_t));
        small_in(h, h->sm_err, err_est, h->cfg.max_anchors, h->msz);
        if ((rc = kfpos_step_toa_dev(h, (const int32_t *)(h->sm_d + h->sm_ranges), h->sm_d + h->sm_err, d_dt, shared,
                                     (uint32_t *)(h->sm_d + h->sm_status), nullptr)))
            return rc;
        return small_finish(h, status);
    }
    stage_reset(h);
    if ((rc = stage_dt(h, dt, dt_len, &d_dt, &shared))) return rc;
    if ((rc = stage_in(h, h->d_ranges, range_mm, h->cfg.max_anchors, sizeof(int32_t)))) return rc;
    if ((rc = stage_in(h, h->d_err, err_est, h->cfg.max_anchors, h->msz))) return rc;
    if ((rc = kfpos_step_toa_dev(h, h->d_ranges, h->d_err, d_dt, shared, h->d_status, nullptr))) return rc;
    return fetch_status(h, status);
}

int kfpos_step_imu(kfpos_handle *h, const void *accel, const void *cov, const double *dt, int32_t dt_len,
                   uint32_t *status) {
    g_err.clear();
    if (!h || !accel || !cov) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    if (h->cfg.model != KFPOS_MODEL_TOA_IMU) {
        if (status) std::memset(status, 0, sizeof(uint32_t) * h->cfg.n_tags);
        return KFPOS_OK;
    }
    const double *d_dt;
    double shared;
    int rc = drain_slots(h);
    if (rc) return rc;
    if (h->sm_h) {
        if ((rc = small_dt(h, dt, dt_len, &d_dt, &shared))) return rc;
        small_in(h, h->sm_accel, accel, 3, h->msz);
        small_in(h, h->sm_cov, cov, 9, h->msz);
        if ((rc = kfpos_step_imu_dev(h, h->sm_d + h->sm_accel, h->sm_d + h->sm_cov, d_dt, shared,
                                     (uint32_t *)(h->sm_d + h->sm_status), nullptr)))
            return rc;
        return small_finish(h, status);
    }
    stage_reset(h);
    if ((rc = stage_dt(h, dt, dt_len, &d_dt, &shared))) return rc;
    if ((rc = stage_in(h, h->d_accel, accel, 3, h->msz))) return rc;
    if ((rc = stage_in(h, h->d_cov, cov, 9, h->msz))) return rc;
    if ((rc = kfpos_step_imu_dev(h, h->d_accel, h->d_cov, d_dt, shared, h->d_status, nullptr))) return rc;
    return fetch_status(h, status);
}

int kfpos_step_sensor(kfpos_handle *h, int32_t kind, const double *data, const double *dt, int32_t dt_len,
                      uint32_t *status) {
    g_err.clear();
    const int C = sensor_width(kind);
    if (!h || !data || C == 0) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    if (h->cfg.model != KFPOS_MODEL_PLANAR) {
        if (status) std::memset(status, 0, sizeof(uint32_t) * h->cfg.n_tags);
        return KFPOS_OK;
    }
    const double *d_dt;
    double shared;
    int rc = drain_slots(h);
    if (rc) return rc;
    if (h->sm_h) {
        if ((rc = small_dt(h, dt, dt_len, &d_dt, &shared))) return rc;
        small_in(h, h->sm_sensor, data, C, sizeof(double));
        if ((rc = kfpos_step_sensor_dev(h, kind, (const double *)(h->sm_d + h->sm_sensor), d_dt, shared,
                                        (uint32_t *)(h->sm_d + h->sm_status), nullptr)))
            return rc;
        return small_finish(h, status);
    }
    stage_reset(h);
    if ((rc = stage_dt(h, dt, dt_len, &d_dt, &shared))) return rc;
    if ((rc = stage_in(h, h->d_sensor, data, C, sizeof(double)))) return rc;
    if ((rc = kfpos_step_sensor_dev(h, kind, h->d_sensor, d_dt, shared, h->d_status, nullptr))) return rc;
    return fetch_status(h, status);
}


/* ---- streaming host API (kfpos.h: kfpos_slot_*) ---- */
static int slots_init(kfpos_handle *h) {
    if (h->s_copy) return KFPOS_OK;
    {
        const char *e = getenv("KFPOS_SLOT_SPLIT_BYTES");
        h->split_bytes = e ? (size_t)atol(e) : 0; /* off: measured slower inside the pipeline (profiles/r02i_*) */
    }
    const size_t T = h->cfg.n_tags, A = h->cfg.max_anchors, m = h->msz;
    size_t off = 0;
    auto region = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    h->so_ranges = region(A * T * sizeof(int32_t));
    h->so_err = region(A * T * m);
    h->so_accel = region(3 * T * m);
    h->so_cov = region(9 * T * m);
    h->so_dt = region(T * sizeof(double));
    h->so_status = region(T * sizeof(uint32_t));
    h->so_pos = region(3 * T * sizeof(double));
    h->so_bytes = off;
    HIPCHK(hipStreamCreateWithFlags(&h->s_copy, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&h->s_copy2, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&h->s_comp, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&h->s_back, hipStreamNonBlocking));
    for (auto &sl : h->slot) {
        HIPCHK(hipHostMalloc((void **)&sl.host, off, hipHostMallocDefault));
        HIPCHK(hipMalloc((void **)&sl.dev, off));
        std::memset(sl.host, 0, off);
        HIPCHK(hipMemset(sl.dev, 0, off));
        HIPCHK(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl.copied2, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl.computed, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    HIPCHK(hipDeviceSynchronize()); /* whatever the other entry points queued on the default stream comes first */
    return KFPOS_OK;
}

int kfpos_slot_count(const kfpos_handle *h) { return h ? KFPOS_N_SLOTS : 0; }

int kfpos_slot_acquire(kfpos_handle *h, int32_t slot, kfpos_epoch_slot *out) {
    g_err.clear();
    if (!h || !out || slot < 0 || slot >= KFPOS_N_SLOTS) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    const int rc = slots_init(h);
    if (rc) return rc;
    auto &sl = h->slot[slot];
    if (sl.busy) { /* its last submission: inputs consumed, outputs delivered */
        HIPCHK(hipEventSynchronize(sl.done));
        sl.busy = false;
    }
    out->range_mm = (int32_t *)(sl.host + h->so_ranges);
    out->err_est = sl.host + h->so_err;
    out->accel = sl.host + h->so_accel;
    out->cov = sl.host + h->so_cov;
    out->dt = (double *)(sl.host + h->so_dt);
    out->status = (uint32_t *)(sl.host + h->so_status);
    out->pos = (double *)(sl.host + h->so_pos);
    return KFPOS_OK;
}

int kfpos_slot_submit(kfpos_handle *h, int32_t slot, int32_t flags, double dt_shared) {
    g_err.clear();
    if (!h || slot < 0 || slot >= KFPOS_N_SLOTS) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    if (!h->s_copy || h->slot[slot].busy) {
        g_err = "kfpos_slot_submit: acquire the slot first (kfpos_slot_acquire)";
        return KFPOS_ERR_STATE;
    }
    const int kind = flags & 0xff;
    if (kind != KFPOS_SLOT_TOA && kind != KFPOS_SLOT_IMU && kind != KFPOS_SLOT_TOA_IMU) return KFPOS_ERR_ARG;
    if (h->cfg.model == KFPOS_MODEL_PLANAR && kind != KFPOS_SLOT_TOA) return KFPOS_ERR_MODEL; /* its sensors: kfpos_step_sensor */
    const bool imu9 = h->cfg.model == KFPOS_MODEL_TOA_IMU;
    if (kind == KFPOS_SLOT_TOA_IMU && !imu9) return KFPOS_ERR_MODEL;
    const bool has_rng = kind != KFPOS_SLOT_IMU, has_imu = imu9 && kind != KFPOS_SLOT_TOA;
    if (kind == KFPOS_SLOT_IMU && !imu9) { /* KalmanFilterTOA::newIMUMeasurement is empty */
        std::memset(h->slot[slot].host + h->so_status, 0, sizeof(uint32_t) * h->cfg.n_tags);
        return KFPOS_OK;
    }
    if (has_rng && !h->have_anchors) {
        g_err = "kfpos_set_anchors has not been called";
        return KFPOS_ERR_STATE;
    }
    auto &sl = h->slot[slot];
    const size_t T = h->cfg.n_tags, A = h->cfg.max_anchors, m = h->msz;
    /* 1. inputs: pinned host -> device on the copy stream. A complete fused epoch is one contiguous block of the slot
     * (ranges | err | accel | cov | dt): one DMA instead of five. Measured on this platform (tools/exp/h2d_probe.hip,
     * profiles/r02g_*): a lone 7.3 MB copy moves at 27-35 GB/s while the GPU is otherwise idle and at 53 GB/s while a
     * kernel is running; two halves on two streams reach 55 GB/s when ALONE but were slower inside this pipeline
     * (KFPOS_SLOT_SPLIT_BYTES=n turns that on for copies of n bytes and more; default off). */
    bool second = false;
    auto h2d = [&](size_t off, size_t bytes) -> hipError_t {
        if (h->split_bytes == 0 || bytes < h->split_bytes) return hipMemcpyAsync(sl.dev + off, sl.host + off, bytes, hipMemcpyHostToDevice, h->s_copy);
        const size_t half = (bytes / 2 + 255) & ~(size_t)255;
        hipError_t e = hipMemcpyAsync(sl.dev + off, sl.host + off, half, hipMemcpyHostToDevice, h->s_copy);
        if (e != hipSuccess) return e;
        second = true;
        return hipMemcpyAsync(sl.dev + off + half, sl.host + off + half, bytes - half, hipMemcpyHostToDevice, h->s_copy2);
    };
    /* a slot's device copy of err / cov may still be read by a later submission of another slot that reused it
     * (KFPOS_SLOT_REUSE_*) -- also after a third slot has uploaded newer ones in between */
    auto guard = [&](hipEvent_t last_use) -> hipError_t {
        hipError_t e = hipStreamWaitEvent(h->s_copy, last_use, 0);
        return e != hipSuccess ? e : hipStreamWaitEvent(h->s_copy2, last_use, 0);
    };
    const bool up_err = has_rng && !(flags & KFPOS_SLOT_REUSE_ERR), up_cov = has_imu && !(flags & KFPOS_SLOT_REUSE_COV);
    if (has_rng && !up_err && h->err_slot < 0) {
        g_err = "KFPOS_SLOT_REUSE_ERR before any errorEstimation was submitted";
        return KFPOS_ERR_STATE;
    }
    if (has_imu && !up_cov && h->cov_slot < 0) {
        g_err = "KFPOS_SLOT_REUSE_COV before any covariance was submitted";
        return KFPOS_ERR_STATE;
    }
    if (up_err && sl.err_reader) HIPCHK(guard(sl.err_reader));
    if (up_cov && sl.cov_reader) HIPCHK(guard(sl.cov_reader));
    if (up_err && up_cov) { /* the whole block in one go */
        const size_t end = (flags & KFPOS_SLOT_DT_PER_TAG) ? h->so_dt + T * sizeof(double) : h->so_cov + 9 * T * m;
        HIPCHK(h2d(h->so_ranges, end - h->so_ranges));
    } else {
        if (has_rng) HIPCHK(h2d(h->so_ranges, up_err ? h->so_err + A * T * m - h->so_ranges : A * T * sizeof(int32_t)));
        if (has_imu) HIPCHK(h2d(h->so_accel, up_cov ? h->so_cov + 9 * T * m - h->so_accel : 3 * T * m));
        if (flags & KFPOS_SLOT_DT_PER_TAG) HIPCHK(h2d(h->so_dt, T * sizeof(double)));
    }
    if (up_err) h->err_slot = slot;
    if (up_cov) h->cov_slot = slot;
    HIPCHK(hipEventRecord(sl.copied, h->s_copy));
    if (second) HIPCHK(hipEventRecord(sl.copied2, h->s_copy2));
    /* 2. the step, on the compute stream (submissions run in order: the filter state is sequential) */
    HIPCHK(hipStreamWaitEvent(h->s_comp, sl.copied, 0));
    if (second) HIPCHK(hipStreamWaitEvent(h->s_comp, sl.copied2, 0));
    KArgs a;
    fill_args(h, a);
    a.mode = kind == KFPOS_SLOT_TOA ? MODE_TOA : (kind == KFPOS_SLOT_IMU ? MODE_IMU_ONLY : MODE_FUSED);
    a.latch = 1;
    if (has_rng) {
        a.ranges = (const int32_t *)(sl.dev + h->so_ranges);
        a.err = h->slot[h->err_slot].dev + h->so_err;
    }
    if (has_imu) {
        a.accel = sl.dev + h->so_accel;
        a.cov = h->slot[h->cov_slot].dev + h->so_cov;
    }
    a.dt = (flags & KFPOS_SLOT_DT_PER_TAG) ? (const double *)(sl.dev + h->so_dt) : nullptr;
    a.dt_shared = dt_shared;
    a.status = (uint32_t *)(sl.dev + h->so_status);
    const bool want_pose = !(flags & KFPOS_SLOT_NO_POSE);
    a.traj = want_pose ? (double *)(sl.dev + h->so_pos) : nullptr;
    const int rc = launch_step(h, a, h->s_comp);
    if (rc) return rc;
    HIPCHK(hipEventRecord(sl.computed, h->s_comp));
    if (has_rng) h->slot[h->err_slot].err_reader = sl.computed;
    if (has_imu) h->slot[h->cov_slot].cov_reader = sl.computed;
    /* 3. outputs: device -> pinned host, on the return stream */
    HIPCHK(hipStreamWaitEvent(h->s_back, sl.computed, 0));
    /* status words and poses sit side by side in the slot: one copy */
    HIPCHK(hipMemcpyAsync(sl.host + h->so_status, sl.dev + h->so_status,
                          want_pose ? h->so_pos + 3 * T * sizeof(double) - h->so_status : T * sizeof(uint32_t),
                          hipMemcpyDeviceToHost, h->s_back));
    HIPCHK(hipEventRecord(sl.done, h->s_back));
    sl.busy = true;
    return KFPOS_OK;
}

int kfpos_slot_wait(kfpos_handle *h, int32_t slot) {
    g_err.clear();
    if (!h || slot < 0 || slot >= KFPOS_N_SLOTS) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    auto &sl = h->slot[slot];
    if (sl.busy) {
        HIPCHK(hipEventSynchronize(sl.done));
        sl.busy = false;
    }
    return KFPOS_OK;
}

int kfpos_get_height(kfpos_handle *h, double *z) {
    g_err.clear();
    if (!h || !z) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    if (h->cfg.model != KFPOS_MODEL_PLANAR) return KFPOS_ERR_MODEL;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(z, h->d_pos + 2 * (size_t)h->cfg.n_tags, sizeof(double) * h->cfg.n_tags, hipMemcpyDeviceToHost));
    return KFPOS_OK;
}

int kfpos_set_height(kfpos_handle *h, const double *z) {
    g_err.clear();
    if (!h || !z) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    if (h->cfg.model != KFPOS_MODEL_PLANAR) return KFPOS_ERR_MODEL;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(h->d_pos + 2 * (size_t)h->cfg.n_tags, z, sizeof(double) * h->cfg.n_tags, hipMemcpyHostToDevice));
    return KFPOS_OK;
}

int kfpos_step_toa_imu(kfpos_handle *h, const int32_t *range_mm, const void *err_est, const void *accel,
                       const void *cov, const double *dt, int32_t dt_len, uint32_t *status) {
    g_err.clear();
    if (!h || !range_mm || !err_est || !accel || !cov) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    if (h->cfg.model != KFPOS_MODEL_TOA_IMU) return KFPOS_ERR_MODEL;
    if (!h->have_anchors) return KFPOS_ERR_STATE;
    const double *d_dt;
    double shared;
    int rc = drain_slots(h);
    if (rc) return rc;
    if (h->sm_h) {
        if ((rc = small_dt(h, dt, dt_len, &d_dt, &shared))) return rc;
        small_in(h, h->sm_ranges, range_mm, h->cfg.max_anchors, sizeof(int32_t));
        small_in(h, h->sm_err, err_est, h->cfg.max_anchors, h->msz);
        small_in(h, h->sm_accel, accel, 3, h->msz);
        small_in(h, h->sm_cov, cov, 9, h->msz);
        if ((rc = kfpos_step_toa_imu_dev(h, (const int32_t *)(h->sm_d + h->sm_ranges), h->sm_d + h->sm_err,
                                         h->sm_d + h->sm_accel, h->sm_d + h->sm_cov, 1, d_dt, shared,
                                         (uint32_t *)(h->sm_d + h->sm_status), nullptr)))
            return rc;
        return small_finish(h, status);
    }
    stage_reset(h);
    if ((rc = stage_dt(h, dt, dt_len, &d_dt, &shared))) return rc;
    if ((rc = stage_in(h, h->d_ranges, range_mm, h->cfg.max_anchors, sizeof(int32_t)))) return rc;
    if ((rc = stage_in(h, h->d_err, err_est, h->cfg.max_anchors, h->msz))) return rc;
    if ((rc = stage_in(h, h->d_accel, accel, 3, h->msz))) return rc;
    if ((rc = stage_in(h, h->d_cov, cov, 9, h->msz))) return rc;
    if ((rc = kfpos_step_toa_imu_dev(h, h->d_ranges, h->d_err, h->d_accel, h->d_cov, 1, d_dt, shared,
                                     h->d_status, nullptr)))
        return rc;
    return fetch_status(h, status);
}

static int get_pose_host(kfpos_handle *h, double dt_ahead, const double *dt_each, double *pos, double *cov3x3,
                         double *vel, uint32_t *status) {
    if (!h) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    const size_t T = h->cfg.n_tags;
    int rc0 = drain_slots(h);
    if (rc0) return rc0;
    if (h->sm_h) {
        const double *d_each = nullptr;
        if (dt_each) {
            std::memcpy(h->sm_h + h->sm_dt, dt_each, sizeof(double) * T);
            d_each = (const double *)(h->sm_d + h->sm_dt);
        }
        double *o = (double *)(h->sm_d + h->sm_out);
        const int rc = launch_pose(h, dt_ahead, d_each, o, o + 3 * T, o + 12 * T, (uint32_t *)(h->sm_d + h->sm_status), nullptr);
        if (rc) return rc;
        if ((rc0 = small_finish(h, status))) return rc0;
        if (pos) small_out(h, pos, 0, 3);
        if (cov3x3) small_out(h, cov3x3, 3 * T, 9);
        if (vel) small_out(h, vel, 12 * T, 3);
        return KFPOS_OK;
    }
    stage_reset(h);
    double *dp = h->d_out, *dc = h->d_out + 3 * T, *dv = h->d_out + 12 * T;
    const double *d_each = nullptr;
    if (dt_each) {
        HIPCHK(hipMemcpy(h->d_dt, dt_each, sizeof(double) * T, hipMemcpyHostToDevice));
        d_each = h->d_dt;
    }
    int rc = launch_pose(h, dt_ahead, d_each, dp, dc, dv, h->d_status, nullptr);
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    if (pos && (rc = stage_out(h, pos, dp, 3))) return rc;
    if (cov3x3 && (rc = stage_out(h, cov3x3, dc, 9))) return rc;
    if (vel && (rc = stage_out(h, vel, dv, 3))) return rc;
    if (status) HIPCHK(hipMemcpy(status, h->d_status, sizeof(uint32_t) * T, hipMemcpyDeviceToHost));
    return KFPOS_OK;
}

int kfpos_get_pose(kfpos_handle *h, double dt_ahead, double *pos, double *cov3x3, double *vel,
                   uint32_t *status) {
    g_err.clear();
    return get_pose_host(h, dt_ahead, nullptr, pos, cov3x3, vel, status);
}

int kfpos_get_pose_each(kfpos_handle *h, const double *dt_ahead, double *pos, double *cov3x3, double *vel,
                        uint32_t *status) {
    g_err.clear();
    if (!dt_ahead) return KFPOS_ERR_ARG;
    return get_pose_host(h, 0.0, dt_ahead, pos, cov3x3, vel, status);
}

int kfpos_get_predicted(kfpos_handle *h, const double *dt_ahead, int32_t dt_len, double *x, double *P,
                        uint32_t *status) {
    g_err.clear();
    if (!h || !dt_ahead || !x || !P || (dt_len != 1 && dt_len != h->cfg.n_tags)) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    const size_t T = h->cfg.n_tags, n = h->n;
    int rc0 = drain_slots(h);
    if (rc0) return rc0;
    if (h->sm_h) {
        const double *d_each = nullptr;
        if (dt_len > 1) {
            std::memcpy(h->sm_h + h->sm_dt, dt_ahead, sizeof(double) * T);
            d_each = (const double *)(h->sm_d + h->sm_dt);
        }
        double *o = (double *)(h->sm_d + h->sm_out) + 15 * T;
        if ((rc0 = launch_pose(h, dt_ahead[0], d_each, nullptr, nullptr, nullptr, (uint32_t *)(h->sm_d + h->sm_status),
                               nullptr, o, o + n * T)))
            return rc0;
        if ((rc0 = small_finish(h, status))) return rc0;
        small_out(h, x, 15 * T, (int)n);
        small_out(h, P, 15 * T + n * T, (int)(n * n));
        return KFPOS_OK;
    }
    stage_reset(h);
    /* component-major results + their row-major turn, all from the staging area: reserve the lot first so that no
     * region moves while another is in use */
    const size_t bx = n * T * sizeof(double), bP = n * n * T * sizeof(double);
    void *probe = nullptr;
    int rc = stage_region(h, 2 * (bx + bP) + 1024, &probe);
    if (rc) return rc;
    stage_reset(h);
    void *vx = nullptr, *vP = nullptr;
    if ((rc = stage_region(h, bx, &vx)) || (rc = stage_region(h, bP, &vP))) return rc;
    double *dx = (double *)vx, *dP = (double *)vP;
    const double *d_each = nullptr;
    if (dt_len > 1 || T == 1) {
        HIPCHK(hipMemcpy(h->d_dt, dt_ahead, sizeof(double) * T, hipMemcpyHostToDevice));
        d_each = h->d_dt;
    }
    if ((rc = launch_pose(h, dt_ahead[0], d_each, nullptr, nullptr, nullptr, h->d_status, nullptr, dx, dP))) return rc;
    HIPCHK(hipDeviceSynchronize());
    if ((rc = stage_out(h, x, dx, (int)n))) return rc;
    if ((rc = stage_out(h, P, dP, (int)(n * n)))) return rc;
    if (status) HIPCHK(hipMemcpy(status, h->d_status, sizeof(uint32_t) * T, hipMemcpyDeviceToHost));
    return KFPOS_OK;
}

int kfpos_get_state(kfpos_handle *h, double *x, double *P, uint32_t *flags) {
    g_err.clear();
    if (!h) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    HIPCHK(hipDeviceSynchronize());
    const size_t T = h->cfg.n_tags;
    const int n = h->n;
    if (x) {
        std::vector<double> pos(3 * T);
        HIPCHK(hipMemcpy(pos.data(), h->d_pos, pos.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<double> vel;
        if (n == 9 || n == 8) {
            vel.resize((n == 8 ? 4 : 3) * T);
            HIPCHK(hipMemcpy(vel.data(), h->d_vel, vel.size() * sizeof(double), hipMemcpyDeviceToHost));
        }
        for (size_t t = 0; t < T; ++t) {
            for (int k = 0; k < n; ++k) x[t * n + k] = 0.0;
            if (n == 8) { /* [x y vx vy 0 0 theta omega]; the height: kfpos_get_height */
                x[t * 8 + 0] = pos[t];
                x[t * 8 + 1] = pos[T + t];
                x[t * 8 + 2] = vel[t];
                x[t * 8 + 3] = vel[T + t];
                x[t * 8 + 6] = vel[2 * T + t];
                x[t * 8 + 7] = vel[3 * T + t];
                continue;
            }
            for (int k = 0; k < 3; ++k) {
                x[t * n + k] = pos[(size_t)k * T + t];
                if (n == 9) x[t * n + 3 + k] = vel[(size_t)k * T + t];
            }
        }
    }
    if (P) {
        std::vector<unsigned char> buf((size_t)h->psz * T * h->rsz);
        HIPCHK(hipMemcpy(buf.data(), h->d_P, buf.size(), hipMemcpyDeviceToHost));
        for (size_t t = 0; t < T; ++t)
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    const size_t k = (size_t)pidx(h, i, j) * T + t;
                    if (h->rsz == 6) { /* KFPOS_STORE_P48 (kfpos_p48.h): [psz][T] uint32 | [psz][T] uint16 */
                        P[(t * n + i) * n + j] = kfpos_p48_decode(((const uint32_t *)buf.data())[k],
                                                                  ((const uint16_t *)(buf.data() + (size_t)h->psz * T * 4))[k]);
                        continue;
                    }
                    P[(t * n + i) * n + j] = h->rsz == 4 ? (double)((const float *)buf.data())[k]
                                                          : ((const double *)buf.data())[k];
                }
    }
    if (flags) HIPCHK(hipMemcpy(flags, h->d_flags, sizeof(uint32_t) * T, hipMemcpyDeviceToHost));
    return KFPOS_OK;
}

int kfpos_set_state(kfpos_handle *h, const double *x, const double *P, const uint32_t *flags) {
    g_err.clear();
    if (!h) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    HIPCHK(hipDeviceSynchronize());
    const size_t T = h->cfg.n_tags;
    const int n = h->n;
    if (x && n == 8) { /* the height row of d_pos is kept */
        std::vector<double> xy(2 * T), vel(4 * T);
        for (size_t t = 0; t < T; ++t) {
            xy[t] = x[t * 8 + 0];
            xy[T + t] = x[t * 8 + 1];
            vel[t] = x[t * 8 + 2];
            vel[T + t] = x[t * 8 + 3];
            vel[2 * T + t] = x[t * 8 + 6];
            vel[3 * T + t] = x[t * 8 + 7];
        }
        HIPCHK(hipMemcpy(h->d_pos, xy.data(), xy.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_vel, vel.data(), vel.size() * sizeof(double), hipMemcpyHostToDevice));
    } else if (x) {
        std::vector<double> pos(3 * T);
        std::vector<double> vel(n == 9 ? 3 * T : 0);
        for (size_t t = 0; t < T; ++t)
            for (int k = 0; k < 3; ++k) {
                pos[(size_t)k * T + t] = x[t * n + k];
                if (n == 9) vel[(size_t)k * T + t] = x[t * n + 3 + k];
            }
        HIPCHK(hipMemcpy(h->d_pos, pos.data(), pos.size() * sizeof(double), hipMemcpyHostToDevice));
        if (n == 9) HIPCHK(hipMemcpy(h->d_vel, vel.data(), vel.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (P) {
        std::vector<unsigned char> buf((size_t)h->psz * T * h->rsz);
        for (size_t t = 0; t < T; ++t)
            for (int i = 0; i < n; ++i)
                for (int j = h->full ? 0 : i; j < n; ++j) {
                    const size_t k = (size_t)pidx(h, i, j) * T + t;
                    const double v = P[(t * n + i) * n + j];
                    if (h->rsz == 6) { /* rounded and encoded exactly as the kernels do (kfpos_p48.h) */
                        kfpos_p48_encode(kfpos_p48_round(v), &((uint32_t *)buf.data())[k],
                                         &((uint16_t *)(buf.data() + (size_t)h->psz * T * 4))[k]);
                    } else if (h->rsz == 4) ((float *)buf.data())[k] = (float)v;
                    else ((double *)buf.data())[k] = v;
                }
        HIPCHK(hipMemcpy(h->d_P, buf.data(), buf.size(), hipMemcpyHostToDevice));
    }
    if (flags) {
        HIPCHK(hipMemcpy(h->d_flags, flags, sizeof(uint32_t) * T, hipMemcpyHostToDevice));
        if (n == 8) /* restored latch bits: ranging epochs must run the instantiation that honours them */
            for (size_t t = 0; t < T; ++t) h->planar_sensors = h->planar_sensors || ((flags[t] >> PLANAR_HAS_SHIFT) != 0);
    }
    h->stepped = true;
    return KFPOS_OK;
}

int kfpos_latch_dim(const kfpos_handle *h) {
    g_err.clear();
    if (!h) return 0;
    return h->cfg.model == KFPOS_MODEL_TOA_IMU ? 12 : (h->cfg.model == KFPOS_MODEL_PLANAR ? LATCH_ROWS : (h->cfg.model == KFPOS_MODEL_ML ? 3 : 0));
}

int kfpos_get_latch(kfpos_handle *h, double *latch) {
    g_err.clear();
    if (!h) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    const int L = kfpos_latch_dim(h);
    if (L == 0) return KFPOS_OK;
    if (!latch) return KFPOS_ERR_ARG;
    HIPCHK(hipDeviceSynchronize());
    stage_reset(h);
    const size_t T = h->cfg.n_tags;
    if (L == LATCH_ROWS) return stage_out(h, latch, h->d_latch, LATCH_ROWS);
    if (L == 3) return stage_out(h, latch, h->d_vel, 3); /* ALGORITHM_ML: _previousEstimation, the seed of every solve */
    /* 9-state: acceleration [3][T] and the lower triangle {00,10,11,20,21,22} [6][T], kfpos_real */
    std::vector<unsigned char> a(3 * T * h->msz), c(6 * T * h->msz);
    HIPCHK(hipMemcpy(a.data(), h->d_imu_acc, a.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(c.data(), h->d_imu_cov, c.size(), hipMemcpyDeviceToHost));
    auto rd = [&](const std::vector<unsigned char> &b, size_t k) {
        return h->msz == 4 ? (double)((const float *)b.data())[k] : ((const double *)b.data())[k];
    };
    static const int tri[3][3] = {{0, 1, 3}, {1, 2, 4}, {3, 4, 5}};
    for (size_t t = 0; t < T; ++t) {
        for (int k = 0; k < 3; ++k) latch[t * 12 + k] = rd(a, (size_t)k * T + t);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) latch[t * 12 + 3 + 3 * i + j] = rd(c, (size_t)tri[i][j] * T + t);
    }
    return KFPOS_OK;
}

int kfpos_set_latch(kfpos_handle *h, const double *latch) {
    g_err.clear();
    if (!h) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    const int L = kfpos_latch_dim(h);
    if (L == 0) return KFPOS_OK;
    if (!latch) return KFPOS_ERR_ARG;
    HIPCHK(hipDeviceSynchronize());
    stage_reset(h);
    const size_t T = h->cfg.n_tags;
    if (L == LATCH_ROWS) return stage_in(h, h->d_latch, latch, LATCH_ROWS, sizeof(double));
    if (L == 3) return stage_in(h, h->d_vel, latch, 3, sizeof(double));
    std::vector<unsigned char> a(3 * T * h->msz), c(6 * T * h->msz);
    auto wr = [&](std::vector<unsigned char> &b, size_t k, double v) {
        if (h->msz == 4) ((float *)b.data())[k] = (float)v;
        else ((double *)b.data())[k] = v;
    };
    static const int li[6] = {0, 1, 1, 2, 2, 2}, lj[6] = {0, 0, 1, 0, 1, 2};
    for (size_t t = 0; t < T; ++t) {
        for (int k = 0; k < 3; ++k) wr(a, (size_t)k * T + t, latch[t * 12 + k]);
        for (int k = 0; k < 6; ++k) wr(c, (size_t)k * T + t, latch[t * 12 + 3 + 3 * li[k] + lj[k]]);
    }
    HIPCHK(hipMemcpy(h->d_imu_acc, a.data(), a.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_imu_cov, c.data(), c.size(), hipMemcpyHostToDevice));
    return KFPOS_OK;
}

int kfpos_timing_begin(kfpos_handle *h, void *stream) {
    g_err.clear();
    if (!h) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    HIPCHK(hipEventRecord(h->ev0, (hipStream_t)stream));
    return KFPOS_OK;
}
int kfpos_timing_end(kfpos_handle *h, void *stream, float *elapsed_ms) {
    g_err.clear();
    if (!h || !elapsed_ms) return KFPOS_ERR_ARG;
    DevScope dev_(h->cfg.device);
    HIPCHK(hipEventRecord(h->ev1, (hipStream_t)stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    return KFPOS_OK;
}

} // extern "C"

/*
 * kfpos_emu.cpp -- host emulation of the per-lane kernel body (roskfpos_amd/csrc/kfpos_core.h).
 *
 * TEST INFRASTRUCTURE ONLY. There is no GPU in the development container, so the CPU test
 * suite compiles the exact per-tag arithmetic the HIP kernels run (same header, same
 * templates) with g++ and checks it against the oracle. It is never loaded by the product:
 * roskfpos_amd fails loudly when libkfpos_hip.so or a GPU is missing. The emulation keeps
 * the kernel's data model: integer-mm ranges in, per-lane scratch (stride 1 instead of the
 * LDS stride), packed covariance.
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../roskfpos_amd/csrc/kfpos_core.h"
#include "../../roskfpos_amd/csrc/kfpos_p48.h"

using namespace kfpos;

/* -DKFE_MSAN (tests/emu/msan_audit.sh, clang -fsanitize=memory): everything the GPU kernels leave undefined is
 * poisoned here too -- the working-weight row of the epoch scratch (stage_epoch_lds* never writes it) and the
 * errorEstimation of an absent range (whatever the caller left there) -- so that a read-before-write in the kernel
 * body shows up as a MemorySanitizer report instead of as a run-to-run varying result on the GPU. */
#ifdef KFE_MSAN
#include <sanitizer/msan_interface.h>
#define KFE_POISON(ptr, bytes) __msan_poison((ptr), (bytes))
#else
#define KFE_POISON(ptr, bytes) ((void)0)
#endif

/* the instantiation the library would launch: without an outlier heuristic the HEUR = false build of the step */
template <bool SYMM, class SC>
static uint32_t toa6_step(Tag6<SYMM> &tg, SC &sc, const Params &pr, double lag, double *park = nullptr, int stride = 0) {
    if (pr.ignore_worst) return step_toa6<SYMM, 2>(tg, sc, pr, lag, park, stride);
    if (pr.top_n) return step_toa6<SYMM, 1>(tg, sc, pr, lag, park, stride);
    return step_toa6<SYMM, 0>(tg, sc, pr, lag, park, stride);
}

struct kfe_bank {
    int model, T, A, full; /* full: COV_FULL layout (6-state ML-init mode) */
    int use_static = 0;    /* run the anchor-count-specialised (register-resident epoch) code path */
    Params pr;
    std::vector<double> anchors;
    std::vector<Tag6<true>> t6s;
    std::vector<Tag6<false>> t6f;
    std::vector<Tag9> t9;
    std::vector<Imu> imu;
    std::vector<double> imu_ci; /* Imu::ci storage, 12 per tag (LDS on the GPU) */
    std::vector<uint32_t> flags;
    std::vector<double> ml_pos, ml_cov, ml_seed; /* model 2: standalone ML estimator */
    std::vector<Tag8> t8;                        /* model 3: planar filter */
    std::vector<Latch8> l8;
    int sensors = 0;                             /* run the SENSORS = true instantiation */
};

extern "C" {

kfe_bank *kfe_create(int model, int n_tags, int n_anchors, const double *anchors, double accel_noise,
                     double jolt, int ignore_worst, double cost_threshold, int top_n, int use_init_pos,
                     const double *init_pos) {
    kfe_bank *b = new kfe_bank();
    b->model = model;
    b->T = n_tags;
    b->A = n_anchors;
    b->full = (model == 0 && !use_init_pos) ? 1 : 0;
    b->anchors.assign(anchors, anchors + 3 * n_anchors);
    b->pr.anchors = b->anchors.data();
    b->pr.n_anchors = n_anchors;
    b->pr.accel_noise = accel_noise;
    b->pr.jolt = jolt;
    b->pr.cost_threshold = cost_threshold;
    b->pr.ignore_worst = ignore_worst;
    b->pr.top_n = top_n;
    b->pr.use_init_pos = use_init_pos;
    b->pr.use_fixed_height = b->pr.imu_fixed_cov_acc = b->pr.imu_fixed_cov_w = 0;
    b->pr.px4_height = b->pr.px4_arm_p1 = b->pr.px4_arm_p2 = b->pr.px4_cov_vel = b->pr.px4_cov_gyro_z = 0.0;
    b->pr.imu_cov_acc = b->pr.imu_cov_w = b->pr.mag_offset = b->pr.mag_cov = 0.0;
    b->flags.assign(n_tags, 0u);
    auto ip = [&](int t, int k) { return use_init_pos ? (init_pos ? init_pos[3 * t + k] : 0.0) : NAN; };
    if (model == 3) {
        b->t8.resize(n_tags);
        b->l8.resize(n_tags);
        for (int t = 0; t < n_tags; ++t) {
            std::memset(&b->t8[t], 0, sizeof(Tag8));
            std::memset(&b->l8[t], 0, sizeof(Latch8));
            b->t8[t].xy[0] = ip(t, 0);
            b->t8[t].xy[1] = ip(t, 1);
        }
    } else if (model == 2) {
        b->ml_pos.assign((size_t)3 * n_tags, NAN);
        b->ml_cov.assign((size_t)6 * n_tags, NAN);
        b->ml_seed.assign((size_t)3 * n_tags, 0.0);
        for (int t = 0; t < n_tags; ++t)
            for (int k = 0; k < 3; ++k) b->ml_seed[3 * t + k] = use_init_pos ? ip(t, k) : (k < 2 ? 1.0 : 4.0);
    } else if (model == 0 && !b->full) {
        b->t6s.resize(n_tags);
        for (int t = 0; t < n_tags; ++t) {
            std::memset(&b->t6s[t], 0, sizeof(Tag6<true>));
            for (int k = 0; k < 3; ++k) b->t6s[t].pos[k] = ip(t, k);
        }
    } else if (model == 0) {
        b->t6f.resize(n_tags);
        for (int t = 0; t < n_tags; ++t) {
            std::memset(&b->t6f[t], 0, sizeof(Tag6<false>));
            for (int k = 0; k < 3; ++k) b->t6f[t].pos[k] = ip(t, k);
        }
    } else {
        b->t9.resize(n_tags);
        b->imu.resize(n_tags);
        b->imu_ci.assign((size_t)12 * n_tags, 0.0);
        for (int t = 0; t < n_tags; ++t) {
            std::memset(&b->t9[t], 0, sizeof(Tag9));
            std::memset(&b->imu[t], 0, sizeof(Imu));
            b->imu[t].ci = &b->imu_ci[(size_t)12 * t];
            b->imu[t].ci_stride = 1;
            for (int k = 0; k < 3; ++k) b->t9[t].pos[k] = ip(t, k);
        }
    }
    return b;
}
void kfe_destroy(kfe_bank *b) { delete b; }
void kfe_set_static(kfe_bank *b, int on) { b->use_static = on; }
void kfe_set_ml_variant(kfe_bank *b, int variant) { b->pr.ml_variant = variant; } /* ALGORITHM_ML: 2 = BEST */
/* cfg: the 14 fields of kfpos_planar_config in order, as doubles */
void kfe_set_planar(kfe_bank *b, const double *cfg, int sensors) {
    Params &p = b->pr;
    p.use_fixed_height = cfg[0] != 0;
    p.px4_height = cfg[3]; p.px4_arm_p1 = cfg[4]; p.px4_arm_p2 = cfg[5]; p.px4_cov_vel = cfg[6]; p.px4_cov_gyro_z = cfg[7];
    p.imu_fixed_cov_acc = cfg[8] != 0; p.imu_cov_acc = cfg[9];
    p.imu_fixed_cov_w = cfg[10] != 0; p.imu_cov_w = cfg[11];
    p.mag_offset = cfg[12]; p.mag_cov = cfg[13];
    b->sensors = sensors;
    for (Tag8 &tg : b->t8) { tg.z = cfg[1]; tg.ang = cfg[2]; tg.om = 0.0; }
}

static void fill_scratch(const kfe_bank *b, const int32_t *mm, const double *err, std::vector<double> &buf,
                         Scratch &sc) {
    const int A = b->A;
    buf.assign(3 * A, 0.0);
    sc.r = buf.data();
    sc.e = buf.data() + A;
    sc.w = buf.data() + 2 * A;
    sc.stride = 1;
    for (int a = 0; a < A; ++a) {
        sc.r[a] = mm[a] > 0 ? (double)mm[a] / 1000 : 0.0; /* Posgenerator.cpp:483-484 */
        sc.e[a] = err[a];
    }
    KFE_POISON(sc.w, sizeof(double) * A);
}

} // extern "C"

template <int AS>
static uint32_t step_static(kfe_bank *b, int t, const int32_t *mm, const double *err, double lag) {
    RegScratch<AS> sc;
    for (int a = 0; a < AS; ++a) {
        sc.r[a] = mm[a] > 0 ? (double)mm[a] / 1000 : 0.0;
        sc.e[a] = err[a];
        sc.w[a] = 0.0;
    }
    if (b->model == 3) {
        const uint32_t rows = ROW_RANGING | b->l8[t].has;
        double park[36];
        return b->sensors ? step_planar8<true>(b->t8[t], sc, b->pr, lag, rows, b->l8[t], CovSpill8{park, 1})
                          : step_planar8<false>(b->t8[t], sc, b->pr, lag, rows, b->l8[t]);
    }
    if (b->model == 2) return step_ml(&b->ml_pos[3 * t], &b->ml_cov[6 * t], sc, b->pr, &b->ml_seed[3 * t]);
    if (b->model == 0 && !b->full) return toa6_step(b->t6s[t], sc, b->pr, lag);
    if (b->model == 0) { double park[36]; return toa6_step(b->t6f[t], sc, b->pr, lag, park, 1); }
    double park9[66];
    return step_imu9<true>(b->t9[t], sc, b->pr, lag, b->imu[t], CovPark9{park9, 1});
}

/* compile-time anchor count over the strided scratch (the LDS-resident epoch of the 16-anchor kernels) */
template <int AS>
static uint32_t step_static_lds(kfe_bank *b, int t, const int32_t *mm, const double *err, double lag,
                                std::vector<double> &buf) {
    StaticScratch<AS> sc;
    fill_scratch(b, mm, err, buf, sc);
    if (b->model == 0 && !b->full) return toa6_step(b->t6s[t], sc, b->pr, lag);
    if (b->model == 0) { double park[36]; return toa6_step(b->t6f[t], sc, b->pr, lag, park, 1); }
    double park9[66];
    return step_imu9<true>(b->t9[t], sc, b->pr, lag, b->imu[t], CovPark9{park9, 1});
}

extern "C" {

void kfe_step_toa(kfe_bank *b, const int32_t *range_mm, const double *err_est, const double *dt, int dt_len,
                  uint32_t *status) {
    std::vector<double> buf;
    Scratch sc;
    for (int t = 0; t < b->T; ++t) {
        const int32_t *mm = range_mm + (size_t)t * b->A;
        const double *err = err_est + (size_t)t * b->A;
        const double lag = dt[dt_len > 1 ? t : 0];
        uint32_t st;
        if (dt_len > 1 && lag < 0) { /* no epoch for this tag */
            if (status) status[t] = ST_SKIPPED;
            continue;
        }
        if (b->use_static == 2 && b->A == 16 && b->model <= 1) st = step_static_lds<16>(b, t, mm, err, lag, buf);
        else if (b->use_static && b->A == 8) st = step_static<8>(b, t, mm, err, lag);
        else if (b->use_static && b->A == 16) st = step_static<16>(b, t, mm, err, lag);
        else if (b->use_static && b->A == 4) st = step_static<4>(b, t, mm, err, lag);
        else {
            fill_scratch(b, mm, err, buf, sc);
            if (b->model == 3) {
                const uint32_t rows = ROW_RANGING | b->l8[t].has;
                double park[36];
                st = b->sensors ? step_planar8<true>(b->t8[t], sc, b->pr, lag, rows, b->l8[t], CovSpill8{park, 1})
                                : step_planar8<false>(b->t8[t], sc, b->pr, lag, rows, b->l8[t]);
            } else if (b->model == 2) st = step_ml(&b->ml_pos[3 * t], &b->ml_cov[6 * t], sc, b->pr, &b->ml_seed[3 * t]);
            else if (b->model == 0 && !b->full) st = toa6_step(b->t6s[t], sc, b->pr, lag);
            else if (b->model == 0) { double park[36]; st = toa6_step(b->t6f[t], sc, b->pr, lag, park, 1); }
            else { double park9[66]; st = step_imu9<true>(b->t9[t], sc, b->pr, lag, b->imu[t], CovPark9{park9, 1}); }
        }
        b->flags[t] |= FL_STARTED;
        if (status) status[t] = st;
    }
}

void kfe_latch_imu(kfe_bank *b, const double *accel, const double *cov) {
    if (b->model != 1) return;
    for (int t = 0; t < b->T; ++t) {
        Imu &im = b->imu[t];
        im.has = true;
        for (int k = 0; k < 3; ++k) im.acc[k] = accel[3 * t + k];
        imu_whitener(cov + 9 * (size_t)t, im.ci, im.ci_stride);
        b->flags[t] |= FL_HAS_IMU;
    }
}

void kfe_step_imu(kfe_bank *b, const double *accel, const double *cov, const double *dt, int dt_len,
                  uint32_t *status) {
    if (b->model != 1) return;
    kfe_latch_imu(b, accel, cov);
    Scratch sc{nullptr, nullptr, nullptr, 1};
    for (int t = 0; t < b->T; ++t) {
        double park9[66];
        uint32_t st = step_imu9<false>(b->t9[t], sc, b->pr, dt[dt_len > 1 ? t : 0], b->imu[t], CovPark9{park9, 1});
        b->flags[t] |= FL_STARTED;
        if (status) status[t] = st;
    }
}

/* The other four sensor entry points of the planar filter. kind: 1 PX4Flow (T x 5), 2 IMU (T x 24: angular
 * velocity 3, its covariance 9, linear acceleration 3, its covariance 9), 3 magnetometer (T x 3), 4 compass (T). */
void kfe_planar_sensor(kfe_bank *b, int kind, const double *data, const double *dt, int dt_len, uint32_t *status) {
    if (b->model != 3) return;
    Scratch sc{nullptr, nullptr, nullptr, 1};
    for (int t = 0; t < b->T; ++t) {
        const double lag = dt[dt_len > 1 ? t : 0];
        if (dt_len > 1 && lag < 0) {
            if (status) status[t] = ST_SKIPPED;
            continue;
        }
        Latch8 &lt = b->l8[t];
        uint32_t rows = 0;
        if (kind == 1) {
            double m[5];
            if (!px4_sample(b->pr, data + 5 * (size_t)t, m)) {
                if (status) status[t] = ST_SKIPPED;
                continue;
            }
            for (int k = 0; k < 5; ++k) lt.px4[k] = m[k];
            lt.has |= ROW_PX4;
            rows = ROW_PX4;
        } else if (kind == 2) {
            const double *d = data + 24 * (size_t)t;
            imu_sample8(b->pr, d, d + 3, d + 12, d + 15, lt.imu);
            lt.has |= ROW_IMU;
            rows = ROW_IMU;
        } else if (kind == 3) {
            lt.mag[0] = atan2(data[3 * (size_t)t + 1], data[3 * (size_t)t]) - b->pr.mag_offset; /* KalmanFilter.cpp:188 */
            lt.mag[1] = b->pr.mag_cov;
            lt.has |= ROW_MAG;
            rows = ROW_MAG;
        } else {
            lt.mag[0] = normalize_angle(data[t]); /* KalmanFilter.cpp:207 */
            lt.mag[1] = b->pr.mag_cov;
            lt.has |= ROW_MAG;
            rows = ROW_MAG | (lt.has & (ROW_PX4 | ROW_IMU));
        }
        double park[36];
        const uint32_t st = step_planar8<true>(b->t8[t], sc, b->pr, lag, rows, lt, CovSpill8{park, 1});
        b->flags[t] |= FL_STARTED;
        if (status) status[t] = st;
    }
}
void kfe_get_height(const kfe_bank *b, double *z) {
    for (int t = 0; t < b->T; ++t) z[t] = b->t8[t].z;
}

/* candidate 6-byte encodings of a covariance entry (study only: tests/cov_encoding_study.py) */
static double enc_f32_bf16(double v) { /* f32 value + bf16 residual */
    const float hi = (float)v;
    float lo = (float)(v - (double)hi);
    uint32_t u;
    std::memcpy(&u, &lo, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u; /* round to nearest even on the upper 16 bits */
    std::memcpy(&lo, &u, 4);
    return (double)hi + (double)lo;
}
static double enc_f48(double v) { /* KFPOS_STORE_P48 as the kernels keep it: through the same codec (kfpos_p48.h) */
    uint32_t hi;
    uint16_t lo;
    kfpos_p48_encode(kfpos_p48_round(v), &hi, &lo);
    return kfpos_p48_decode(hi, lo);
}

/* Emulate KFPOS_STORE_F32: what the step kernels keep in HBM between epochs is rounded to float
 * (covariance when what&1, velocity when what&2); positions always stay double. what&4 / what&8: the covariance
 * through the f32 + bf16 / 48-bit encodings above instead. */
void kfe_round_storage(kfe_bank *b, int what) {
    for (int t = 0; t < b->T; ++t) {
        if (b->model == 1 && (what & 12)) {
            for (double &v : b->t9[t].P.a) v = (what & 4) ? enc_f32_bf16(v) : enc_f48(v);
            continue;
        }
        if (b->model == 1) {
            if (what & 1) for (double &v : b->t9[t].P.a) v = (double)(float)v;
            if (what & 2) for (double &v : b->t9[t].vel) v = (double)(float)v;
            /* bits 4..9: round one 3x3 block pair only: pp, pv, pa, vv, va, aa */
            const int bi[6] = {0, 0, 0, 3, 3, 6}, bj[6] = {0, 3, 6, 3, 6, 6};
            for (int k = 0; k < 6; ++k)
                if (what & (16 << k))
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j) {
                            double &v = b->t9[t].P(bi[k] + i, bj[k] + j);
                            v = (double)(float)v;
                        }
        } else if (b->full) {
            if (what & 1) for (double &v : b->t6f[t].P.a) v = (double)(float)v;
            if (what & 8) for (double &v : b->t6f[t].P.a) v = enc_f48(v);
        } else {
            if (what & 1) for (double &v : b->t6s[t].P.a) v = (double)(float)v;
            if (what & 8) for (double &v : b->t6s[t].P.a) v = enc_f48(v);
        }
    }
}

/* x: T*n ([pos, vel(, 0)]), P: T*n*n full row-major */
void kfe_get_state(const kfe_bank *b, double *x, double *P) {
    if (b->model == 3) {
        for (int t = 0; t < b->T; ++t) {
            const Tag8 &tg = b->t8[t];
            const double s8[8] = {tg.xy[0], tg.xy[1], tg.vel[0], tg.vel[1], 0.0, 0.0, tg.ang, tg.om};
            for (int k = 0; k < 8; ++k) x[8 * (size_t)t + k] = s8[k];
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 8; ++j) P[64 * (size_t)t + 8 * i + j] = tg.P(i, j);
        }
        return;
    }
    if (b->model == 2) {
        for (int t = 0; t < b->T; ++t) {
            const double *c = &b->ml_cov[6 * t];
            const double full[9] = {c[0], c[1], c[2], c[1], c[3], c[4], c[2], c[4], c[5]};
            for (int k = 0; k < 3; ++k) x[3 * t + k] = b->ml_pos[3 * t + k];
            for (int k = 0; k < 9; ++k) P[9 * t + k] = full[k];
        }
        return;
    }
    const int n = b->model == 1 ? 9 : 6;
    for (int t = 0; t < b->T; ++t) {
        double *xt = x + (size_t)t * n, *Pt = P + (size_t)t * n * n;
        for (int k = 0; k < n; ++k) xt[k] = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                if (b->model == 1) Pt[i * n + j] = b->t9[t].P(i, j);
                else if (b->full) Pt[i * n + j] = b->t6f[t].P(i, j);
                else Pt[i * n + j] = b->t6s[t].P(i, j);
            }
        for (int k = 0; k < 3; ++k) {
            if (b->model == 1) { xt[k] = b->t9[t].pos[k]; xt[3 + k] = b->t9[t].vel[k]; }
            else xt[k] = b->full ? b->t6f[t].pos[k] : b->t6s[t].pos[k];
        }
    }
}

/* unit access to the rank test and the Jacobi pseudo-inverse of the non-symmetric 6x6 path */
int kfe_pinv6(const double *P, double *out) {
    Cov<6, false> c;
    for (int i = 0; i < 36; ++i) c.a[i] = P[i];
    pinv6_jacobi(c, out, 1);
    return cov6_suspect(c) ? 1 : 0;
}

void kfe_get_pose(const kfe_bank *b, double dt_ahead, double *pos, double *cov3x3, double *vel) {
    for (int t = 0; t < b->T; ++t) {
        double v[3] = {0, 0, 0};
        if (!(b->flags[t] & FL_STARTED)) {
            for (int k = 0; k < 3; ++k) { pos[3 * t + k] = NAN; vel[3 * t + k] = NAN; }
            for (int k = 0; k < 9; ++k) cov3x3[9 * t + k] = NAN;
            continue;
        }
        if (b->model == 3) {
            double x8[8];
            Cov<8, true> Pp;
            pose8(b->t8[t], dt_ahead, b->pr.accel_noise, b->pr.jolt, x8, Pp);
            const double c[9] = {Pp(0, 0), Pp(0, 1), 0, Pp(0, 1), Pp(1, 1), 0, 0, 0, 0.01};
            pos[3 * t] = x8[0]; pos[3 * t + 1] = x8[1]; pos[3 * t + 2] = b->t8[t].z;
            vel[3 * t] = x8[2]; vel[3 * t + 1] = x8[3]; vel[3 * t + 2] = 0.0;
            for (int k = 0; k < 9; ++k) cov3x3[9 * t + k] = c[k];
            continue;
        }
        if (b->model == 1) pose9(b->t9[t], dt_ahead, b->pr.jolt, pos + 3 * t, v, cov3x3 + 9 * t);
        else if (b->full) pose6(b->t6f[t], dt_ahead, b->pr.accel_noise, pos + 3 * t, cov3x3 + 9 * t);
        else pose6(b->t6s[t], dt_ahead, b->pr.accel_noise, pos + 3 * t, cov3x3 + 9 * t);
        for (int k = 0; k < 3; ++k) vel[3 * t + k] = v[k];
    }
}

} // extern "C"

/*
 * kfpos_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (GTEC-UDC/roskfpos) ships no tests, golden
 * vectors or fixtures for this path, and its arithmetic lives in Armadillo +
 * LAPACK, which are neither vendored in the reference nor installed here, so
 * the reference cannot be compiled in this image without writing stand-ins
 * for those headers (not allowed).  This oracle is therefore a from-scratch
 * restatement of the reference algorithm, dense and in the reference's own
 * operation order; see kfpos_oracle.cpp for the file:line map.
 */
#ifndef KFPOS_ORACLE_H
#define KFPOS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kfo_filter_bank kfo_filter_bank;

/* model ids follow node_pos.cpp:50-57 (ALGORITHM_KF_TOA / ALGORITHM_KF_TOA_IMU) */
#define KFO_MODEL_TOA     0 /* KalmanFilterTOA,    6 states */
#define KFO_MODEL_TOA_IMU 1 /* KalmanFilterTOAIMU, 9 states (with the 3-token repair) */
#define KFO_MODEL_ML      2 /* MLLocation as a standalone estimator (ALGORITHM_ML), 3-D, variant NORMAL / IGNORE_N */
#define KFO_MODEL_PLANAR  3 /* KalmanFilter (ALGORITHM_KF), 8 states [x y vx vy ax ay theta omega], ranging rows */

/* per-tag status bits reported by a step (same layout as include/kfpos.h) */
#define KFO_ST_UPDATE_SKIPPED 0x01u /* inv/pinv/solve "threw": predicted P kept (KalmanFilterTOA.cpp:151-153) */
#define KFO_ST_ML_FALLBACK    0x02u /* ML position NaN -> predicted position (KalmanFilterTOA.cpp:270-272) */
#define KFO_ST_FEW_RANGES     0x04u /* <4 ranges: ML returned its seed (MLLocation.cpp:158-161) */
#define KFO_ST_ML_INIT        0x08u /* this call was the ML initialisation (KalmanFilterTOA.cpp:90-108) */
#define KFO_ST_NOT_STARTED    0x10u /* getPose before the first measurement (KalmanFilterTOA.cpp:442-447) */
#define KFO_ST_NONFINITE      0x20u /* state is not finite after the call */
#define KFO_ST_SKIPPED        0x40u /* dt < 0: this tag's filter was not called */
/* bits 8..15: IEKF gain iterations; 16..23: ML Gauss-Newton iterations (sat. 255);
 * 24..31: 1 + index (into the >0 ranges of the epoch) of the anchor dropped by
 * the leave-one-out heuristic, 0 if none. */

/* ALGORITHM_ML: 0 NORMAL / 1 IGNORE_N (= top_n > 0) / 2 BEST (estimatePositionBestGroup; defined for 4 or 5 ranges) */
void kfo_set_ml_variant(kfo_filter_bank *o, int variant);
kfo_filter_bank *kfo_create(int model, int n_tags, int max_anchors,
                            double accel_noise, double jolt,
                            int ignore_worst, double cost_threshold, int top_n,
                            int use_init_pos, const double *init_pos /* n_tags*3 or NULL */);
void kfo_destroy(kfo_filter_bank *);
int  kfo_state_dim(const kfo_filter_bank *);
/* KFO_MODEL_PLANAR: what KalmanFilter::loadConfigurationFiles reads (KalmanFilter.cpp:748-842) plus the
 * constructor's initialAngle; call before the first step. kfo_get_height returns mUWBtagZ per tag. */
typedef struct {
    int use_fixed_height; double fixed_height; /* <uwb useFixedHeight fixedHeight/> */
    double init_angle;
    double px4_height, px4_arm_p1, px4_arm_p2, px4_cov_velocity, px4_cov_gyro_z; /* <px4flow sensorHeight armP0 armP1 .../> */
    int imu_use_fixed_cov_acc; double imu_cov_acc;                               /* <imu .../> */
    int imu_use_fixed_cov_ang_vel_z; double imu_cov_ang_vel_z;
    double mag_angle_offset, mag_cov;                                            /* <mag angleOffset covarianceMag/> */
} kfo_planar_config;
void kfo_set_planar(kfo_filter_bank *, const kfo_planar_config *);
void kfo_get_height(const kfo_filter_bank *, double *z);
/* The other four sensor entry points of KalmanFilter (dt as in kfo_step_toa; dt < 0 with dt_len = n_tags skips a tag).
 * flow: T x 5 (integrationX, integrationY, integrationRotationZ, integrationTime [us], quality); a sample with
 * quality 0 is dropped exactly like the reference does (status KFO_ST_SKIPPED). */
void kfo_planar_px4flow(kfo_filter_bank *, const double *flow, const double *dt, int dt_len, uint32_t *status, int n_threads);
void kfo_planar_imu(kfo_filter_bank *, const double *ang_vel, const double *cov_ang_vel, const double *lin_acc,
                    const double *cov_acc, const double *dt, int dt_len, uint32_t *status, int n_threads);
void kfo_planar_mag(kfo_filter_bank *, const double *mag_xyz, const double *dt, int dt_len, uint32_t *status, int n_threads);
void kfo_planar_compass(kfo_filter_bank *, const double *compass, const double *dt, int dt_len, uint32_t *status, int n_threads);
void kfo_set_anchors(kfo_filter_bank *, const double *xyz /* A*3 */, int n_anchors);

/* One ranging epoch for every tag: KalmanFilterTOA::newTOAMeasurement
 * (KalmanFilterTOA.cpp:43-61) / KalmanFilterTOAIMU::newTOAMeasurement
 * (KalmanFilterTOAIMU.cpp:49-73) with the wall-clock dt passed as data.
 * range_mm[t*A+a] <= 0 means "no range" (Posgenerator.cpp:483).
 * dt_len is 1 (shared) or n_tags. status may be NULL. */
void kfo_step_toa(kfo_filter_bank *, const int32_t *range_mm, const double *err_est,
                  const double *dt, int dt_len, uint32_t *status, int n_threads);

/* KalmanFilterTOAIMU::newIMUMeasurement (KalmanFilterTOAIMU.cpp:76-92):
 * latch the sample and run the IMU-only estimate. accel n_tags*3, cov n_tags*9 row-major. */
void kfo_step_imu(kfo_filter_bank *, const double *accel, const double *cov,
                  const double *dt, int dt_len, uint32_t *status, int n_threads);

/* getPose (KalmanFilterTOA.cpp:438-473, KalmanFilterTOAIMU.cpp:476-510):
 * predict-only extrapolation by dt_ahead; pos n_tags*3, cov n_tags*9 (3x3 position block),
 * vel n_tags*3 (may be NULL). */
void kfo_get_pose(const kfo_filter_bank *, double dt_ahead, double *pos, double *cov3x3,
                  double *vel, uint32_t *status);

/* raw filter members: x = [pos, vel(, acc=0)] n_tags*n, P n_tags*n*n row-major */
void kfo_get_state(const kfo_filter_bank *, double *x, double *P);
void kfo_set_state(kfo_filter_bank *, const double *x, const double *P, int started);

/* ---- unit-level entry points for known-answer tests ---- */
/* MLLocation::estimatePosition (MLLocation.cpp:153-257). Returns iterations, -1 if an
 * inverse/solve failed. cov may be NULL. */
int kfo_ml_estimate(int n, const double *anchors_xyz, const double *ranges,
                    const double *err_est, const double *seed, double *pos, double *cov3x3);
/* predictionMatrix / predictionErrorCovariance, n*n row-major each */
void kfo_predict_matrices(int model, double dt, double accel_noise, double jolt,
                          double *F, double *Q);
int kfo_inv(int n, const double *A, double *out);   /* 0 ok, 1 singular/non-finite */
int kfo_pinv(int n, const double *A, double *out);  /* 0 ok, 1 svd failed */
int kfo_solve_equilibrate(int n, const double *A, const double *b, double *x);
/* residual^2 ranking + drop rule of the top-N composition (MLLocation.cpp:284-300,325-339).
 * keep[i] = 1/0 for the n given ranges. */
void kfo_topn_keep(int n, const double *anchors_xyz, const double *ranges,
                   const double *err_est, const double *seed, int top_n, int *keep);

#ifdef __cplusplus
}
#endif
#endif

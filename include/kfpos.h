/*
 * kfpos.h -- C ABI of the MI355X batched EKF positioning core (libkfpos_hip.so).
 *
 * Drop-in boundary: the reference's estimator interface
 *   class PositionEstimationAlgorithm   src/kfpos/algorithms/PositionEstimationAlgorithm.h:8-37
 * as implemented by
 *   KalmanFilterTOA      src/kfpos/algorithms/KalmanFilterTOA.{h,cpp}     (ALGORITHM_KF_TOA)
 *   KalmanFilterTOAIMU   src/kfpos/algorithms/KalmanFilterTOAIMU.{h,cpp}  (ALGORITHM_KF_TOA_IMU)
 * and called from PosGenerator (src/kfpos/publishers/Posgenerator.cpp:491 newTOAMeasurement,
 * :139 newIMUMeasurement, :537 init, :543 getPose). The reference runs ONE filter for ONE tag per
 * process; this library runs T independent filters per handle, one per GPU lane, with the same
 * per-filter semantics. The wall clock the reference reads inside the estimator
 * (KalmanFilterTOA.cpp:76-88) is passed in as dt. Plain C types only: no exceptions cross this
 * boundary, per-tag conditions are reported in a status word instead.
 *
 * Host-side mirrors of the reference classes on top of this ABI: roskfpos_amd/csrc/kfpos_adaptor.h.
 * The binding a reference maintainer would add: INTEGRATION.md.
 */
#ifndef KFPOS_H
#define KFPOS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KFPOS_VERSION 102 /* 0.1.2: KFPOS_STORE_P48 keeps 40 significant bits (0.1.1: kfpos_config.ml_variant, kfpos_comm_*) */

/* ---- return codes (every entry point returns one; 0 = success) ---- */
#define KFPOS_OK            0
#define KFPOS_ERR_ARG       1 /* NULL / out-of-range argument */
#define KFPOS_ERR_HIP       2 /* a HIP call failed; kfpos_last_error() has the text */
#define KFPOS_ERR_NO_DEVICE 3 /* no usable gfx950 device */
#define KFPOS_ERR_MODEL     4 /* call not defined for this handle's model */
#define KFPOS_ERR_STATE     5 /* e.g. anchors not set yet (Posgenerator.cpp:92-96 drops ranges until then) */
#define KFPOS_ERR_COMM      6 /* an RCCL call failed, or librccl could not be opened; kfpos_last_error() has the text */

/* ---- models: the `algorithm` launch parameter (node_pos.cpp:48-58) ---- */
#define KFPOS_MODEL_TOA     0 /* ALGORITHM_KF_TOA     -> KalmanFilterTOA, 6 states p,v */
#define KFPOS_MODEL_TOA_IMU 1 /* ALGORITHM_KF_TOA_IMU -> KalmanFilterTOAIMU, 9 states p,v,a (3-token repair) */
#define KFPOS_MODEL_ML      2 /* ALGORITHM_ML -> MLLocation as the estimator (MLLocation.cpp:421-486): 3-D, variant NORMAL
                               (top_n = 0), IGNORE_N (top_n = numRangingsToIgnore) or BEST (kfpos_config.ml_variant);
                               state = position, P = its 3x3 covariance; dt is ignored, use_init_pos selects the
                               solver's seed ({1,1,4} otherwise) */
#define KFPOS_MODEL_PLANAR  3 /* ALGORITHM_KF -> KalmanFilter (KalmanFilter.cpp): 8 states [x y vx vy ax ay theta omega] at a
                               fixed height, fed by ranging epochs and, optionally, PX4Flow / IMU / magnetometer /
                               compass samples (kfpos_step_sensor). Configure with kfpos_set_planar() before stepping */

/* ---- storage precision of the covariance and of the measurements in HBM; arithmetic is always f64 ----
 * Positions and velocities are kept as double in every mode, ranges are exact integer mm. F32 rounds the covariance to
 * 24 bits between epochs: the 6-state filter stays at 3e-9 m from the CPU reference, the 9-state filter amplifies that
 * rounding to 1.6e-6 m RMS over 100 epochs of the BASELINE trace -- above the 1e-6 m bar: use P48 or MIXED there.
 * F64 and MIXED keep the covariance exactly. */
#define KFPOS_STORE_F64   0 /* covariance double, measurements (kfpos_real) double */
#define KFPOS_STORE_F32   1 /* covariance float,  measurements (kfpos_real) float */
#define KFPOS_STORE_MIXED 2 /* covariance double, measurements (kfpos_real) float: exact filter state, compact inputs */
#define KFPOS_STORE_P48   3 /* covariance in 6 bytes per entry -- sign, 8 exponent bits (single's range), 39 mantissa bits,
                               rounded to nearest (roskfpos_amd/csrc/kfpos_p48.h) --, measurements float: the compact mode
                               that keeps the 9-state filter inside the 1e-6 m bar (1e-9 m on the 100-epoch BASELINE trace;
                               6.6e-8 m RMS over 2 048 tags x 2 000 epochs, no single epoch above 6e-7 m; F32's 24 bits give
                               1.6e-6 m after 100 epochs). Over 10 000 epochs one epoch of capped steps takes it to 2.7e-6 m
                               (MIXED: 1e-8 m): the mode for the 25 % smaller state, not for the last digits. An infinite
                               entry is stored as NaN. DESIGN.md section 3 */

#define KFPOS_MAX_ANCHORS 64 /* MAX_NUM_ANCS, Posgenerator.h:74 */

/* ---- per-tag status word ---- */
#define KFPOS_ST_UPDATE_SKIPPED 0x01u /* the std::runtime_error the reference swallows (KalmanFilterTOA.cpp:151-153); the
                                         9-state and planar filters have no try/catch (the node would abort): here the
                                         tag keeps its predicted covariance and reports this bit. Also: ML initialisation
                                         whose covariance is exactly singular, planar sensor row with zero variance */
#define KFPOS_ST_ML_FALLBACK    0x02u /* ML position NaN -> predicted position (KalmanFilterTOA.cpp:270-272) */
#define KFPOS_ST_FEW_RANGES     0x04u /* < 4 ranges this epoch (< 3 for the planar filter's 2-D solve): ML returned its seed
                                         (MLLocation.cpp:158-161, :54-58) */
#define KFPOS_ST_ML_INIT        0x08u /* this epoch was consumed by the ML initialisation (KalmanFilterTOA.cpp:90-108) */
#define KFPOS_ST_NOT_STARTED    0x10u /* getPose() == false: no measurement yet (KalmanFilterTOA.cpp:442-447) */
#define KFPOS_ST_NONFINITE      0x20u /* state not finite after the call */
#define KFPOS_ST_SKIPPED        0x40u /* dt < 0 was passed for this tag: no estimator call, filter untouched */
#define KFPOS_ST_GAIN_ITERS(s)  (((s) >> 8) & 0xffu)  /* IEKF gain iterations (KalmanFilterTOA.cpp:293-324); KFPOS_MODEL_ML with
                                                         top_n > 0: Gauss-Newton iterations of the solve that ranks the
                                                         residuals (estimatePositionIgnoreN's first, MLLocation.cpp:318) */
#define KFPOS_ST_ML_ITERS(s)    (((s) >> 16) & 0xffu) /* ML Gauss-Newton iterations, saturating (MLLocation.cpp:168-225) */
#define KFPOS_ST_IGNORED(s)     ((int)(((s) >> 24) & 0xffu) - 1) /* index among this epoch's >0 ranges of the anchor the
                                                                    leave-one-out heuristic dropped, -1 if none */

/* Construction parameters: the constructor arguments of KalmanFilterTOA (KalmanFilterTOA.h:21-22) and
 * KalmanFilterTOAIMU (KalmanFilterTOAIMU.h:19-20) plus the batch shape. Launch-file names in brackets. */
typedef struct kfpos_config {
    int32_t model;          /* KFPOS_MODEL_*                                   [algorithm] */
    int32_t n_tags;         /* T, filters in this handle (>= 1) */
    int32_t max_anchors;    /* A, columns of the range matrix, 1..KFPOS_MAX_ANCHORS */
    int32_t storage;        /* KFPOS_STORE_* */
    double  accel_noise;    /* accelerationNoise                               [accelNoise], kfpos_toa.launch:13 */
    double  jolt;           /* jolt (9-state process noise)                    [jolt], kfpos_toa_imu.launch */
    int32_t ignore_worst;   /* ignoreWorstAnchorMode (6-state only)            [useHeuristicIgnoreWorst] */
    double  cost_threshold; /* ignoreCostThreshold                             [heuristicIgnoreThreshold] */
    int32_t top_n;          /* ranges dropped by residual ranking, 0 = off     [numRangingsToIgnore], BASELINE config 5 */
    int32_t use_init_pos;   /* 1: fixed initial position, P0 = 0; 0: ML initialisation on the first epoch
                               (the reference's mUseFixedInitialPosition)      [useStartPosition] */
    double  init_pos[3];    /* initialPosition for every tag                   [initPositionX/Y/Z]; per-tag values:
                               kfpos_set_init_positions() */
    int32_t device;         /* HIP device ordinal */
    int32_t ml_variant;     /* KFPOS_MODEL_ML only (0 otherwise): KFPOS_ML_NORMAL / _IGNORE_N / _BEST  [variant],
                               MLLocation.h:5-7, Posgenerator.cpp:79-83 */
} kfpos_config;
#define KFPOS_ML_NORMAL   0 /* estimatePosition (MLLocation.cpp:153-257); with top_n > 0 it acts as IGNORE_N */
#define KFPOS_ML_IGNORE_N 1 /* estimatePositionIgnoreN (:307-347) with numRangingsToIgnore = top_n */
#define KFPOS_ML_BEST     2 /* estimatePositionBestGroup (:348-414): every subset of 4 ranges, smallest covariance trace
                               wins. Defined by the reference for 4 or 5 ranges only -- its erase loop (:377-381) runs past
                               the end of the vector from 6 ranges on -- so kfpos_set_anchors refuses more than 5 anchors
                               for this variant (KFPOS_ERR_MODEL); top_n is ignored, as numRangingsToIgnore is there.
                               Status word: KFPOS_ST_IGNORED = the range the winning group did without, KFPOS_ST_GAIN_ITERS
                               = the most Gauss-Newton passes any group took */

/* KFPOS_MODEL_PLANAR only: the attributes KalmanFilter::loadConfigurationFiles reads from the four XML
 * parameters (KalmanFilter.cpp:748-842; config_uwb.xml / config_px4flow.xml / config_imu.xml / config_mag.xml
 * of src/kfpos/config) plus the constructor's initialAngle (KalmanFilter.h:32). */
typedef struct kfpos_planar_config {
    int32_t use_fixed_height;  /* <uwb useFixedHeight/>: 1 = 2-D ML initialisation at fixed_height, 0 = 3-D ML
                                  initialisation whose z becomes the height (KalmanFilter.cpp:252-258) */
    double  fixed_height;      /* <uwb fixedHeight/> = mUWBtagZ */
    double  init_angle;        /* initialAngle [initAngle] */
    double  px4_height;        /* <px4flow sensorHeight/> */
    double  px4_arm_p1;        /* <px4flow armP0/> */
    double  px4_arm_p2;        /* <px4flow armP1/> */
    double  px4_cov_velocity;  /* <px4flow covarianceVelocity/> */
    double  px4_cov_gyro_z;    /* <px4flow covarianceGyroZ/> */
    int32_t imu_use_fixed_cov_acc;       /* <imu useFixedCovarianceAcceleration/> */
    double  imu_cov_acc;                 /* <imu covarianceAcceleration/> */
    int32_t imu_use_fixed_cov_ang_vel_z; /* <imu useFixedCovarianceAngularVelocityZ/> */
    double  imu_cov_ang_vel_z;           /* <imu covarianceAngularVelocityZ/> */
    double  mag_angle_offset;  /* <mag angleOffset/> */
    double  mag_cov;           /* <mag covarianceMag/> */
} kfpos_planar_config;

/* sensor kinds of kfpos_step_sensor: the other four entry points of PositionEstimationAlgorithm */
#define KFPOS_SENSOR_PX4FLOW 1 /* newPX4FlowMeasurement (KalmanFilter.cpp:102-135): 5 values per tag = integrationX,
                                  integrationY, integrationRotationZ, integrationTime [us], quality. A sample with
                                  quality 0 is dropped on entry (status KFPOS_ST_SKIPPED) */
#define KFPOS_SENSOR_IMU     2 /* newIMUMeasurement (:139-182): 24 values = angularVelocity[3], its covariance[9],
                                  linearAcceleration[3], its covariance[9] (row-major 3x3) */
#define KFPOS_SENSOR_MAG     3 /* newMAGMeasurement (:185-199): 3 values = mag x, y, z (its covariance argument is
                                  ignored by the reference and is not part of this ABI) */
#define KFPOS_SENSOR_COMPASS 4 /* newCompassMeasurement (:201-229): 1 value = heading in radians */

typedef struct kfpos_handle kfpos_handle;

/* ---- lifetime: PosGenerator::setAlgorithm (Posgenerator.cpp:510-538) ---- */
int kfpos_create(const kfpos_config *cfg, kfpos_handle **out);
int kfpos_destroy(kfpos_handle *h);
/* PositionEstimationAlgorithm::init(). Returns KFPOS_OK (the reference's `true`). */
int kfpos_init(kfpos_handle *h);
/* Beacon table: what PosGenerator keeps in `beacons` from the anchors topic (Posgenerator.cpp:15-39)
 * and zips into newTOAMeasurement's `beacons` argument (:486-487). xyz: n_anchors*3 doubles, metres.
 * ids may be NULL (kept for the caller's id->column map only). */
int kfpos_set_anchors(kfpos_handle *h, const double *xyz, const int32_t *ids, int32_t n_anchors);
/* Per-tag fixed start positions, n_tags*3 doubles (batched form of the initialPosition ctor argument).
 * Only before the first step and only with use_init_pos = 1. */
int kfpos_set_init_positions(kfpos_handle *h, const double *xyz);

/* KFPOS_MODEL_PLANAR: what KalmanFilter::init() loads. Before the first step; every tag starts at
 * fixed_height / init_angle. KFPOS_ERR_MODEL on other models. */
int kfpos_set_planar(kfpos_handle *h, const kfpos_planar_config *cfg);

/* size in bytes of kfpos_real for this handle (8 or 4) */
int kfpos_real_size(const kfpos_handle *h);

/* ---- synchronous host-buffer API (pointers are borrowed until the call returns) ---- */

/* One ranging epoch for every tag = T calls of newTOAMeasurement (KalmanFilterTOA.cpp:43-61,
 * KalmanFilterTOAIMU.cpp:49-73) -> estimatePositionKF (predict + iterated update).
 *   range_mm  n_tags x max_anchors, row-major, integer millimetres exactly as PosGenerator stores them
 *             (Posgenerator.cpp:213); <= 0 = no range from that anchor this epoch (:483)
 *   err_est   n_tags x max_anchors, kfpos_real, m^2, must be > 0 where a range is present (:485)
 *   dt        seconds since the previous estimate of each tag (the reference's wall-clock timeLag;
 *             0.1 on a filter's first call, KalmanFilterTOA.cpp:81); dt_len = 1 (shared) or n_tags.
 *             With dt_len = n_tags a NEGATIVE dt[t] means "tag t has no epoch in this call": its filter is
 *             left untouched (tags report asynchronously; kfpos_ingest.h batches whoever flushed)
 *   status    n_tags words or NULL
 * A 9-state handle fuses the latched IMU sample again, as the reference does (KalmanFilterTOAIMU.cpp:68-72). */
int kfpos_step_toa(kfpos_handle *h, const int32_t *range_mm, const void *err_est,
                   const double *dt, int32_t dt_len, uint32_t *status);

/* One IMU sample for every tag = T calls of newIMUMeasurement (KalmanFilterTOAIMU.cpp:76-92): latch
 * linearAcceleration + covarianceAcceleration, then predict + IMU-only update. angularVelocity and its
 * covariance are ignored by the reference and are not part of this ABI. 6-state handles: no-op
 * (KalmanFilterTOA.cpp:64), returns KFPOS_OK.
 *   accel n_tags x 3, cov n_tags x 9 (row-major 3x3, symmetric positive definite), kfpos_real */
int kfpos_step_imu(kfpos_handle *h, const void *accel, const void *cov,
                   const double *dt, int32_t dt_len, uint32_t *status);

/* Fused epoch: latch the IMU sample, then the ranging epoch -- exactly the reference sequence
 * { newIMUMeasurement at timeLag 0 ; newTOAMeasurement at timeLag dt } (the IMU-only estimate is the
 * identity at timeLag 0). 9-state handles only. */
int kfpos_step_toa_imu(kfpos_handle *h, const int32_t *range_mm, const void *err_est,
                       const void *accel, const void *cov,
                       const double *dt, int32_t dt_len, uint32_t *status);

/* KFPOS_MODEL_PLANAR: one sample of one of the other four sensors for every tag = T calls of
 * newPX4FlowMeasurement / newIMUMeasurement / newMAGMeasurement / newCompassMeasurement: latch the sample, then
 * predict + update with that sensor's rows (the compass call also carries the latched PX4Flow and IMU samples,
 * KalmanFilter.cpp:213-226). From then on every ranging epoch of that tag carries its latched samples
 * (:84-98). data: n_tags x C doubles row-major, C by kind (KFPOS_SENSOR_*); dt / dt_len / status as in
 * kfpos_step_toa. Other models: the reference's empty virtuals, returns KFPOS_OK and status 0. */
int kfpos_step_sensor(kfpos_handle *h, int32_t kind, const double *data,
                      const double *dt, int32_t dt_len, uint32_t *status);
/* mUWBtagZ of every tag (n_tags doubles): the configured height, or the ML initialisation's z. kfpos_set_height puts
 * it back when a checkpoint is restored (with useFixedHeight = 0 the 3-D ML initialisation replaced the configured
 * value, KalmanFilter.cpp:252-258). KFPOS_MODEL_PLANAR only. */
int kfpos_get_height(kfpos_handle *h, double *z);
int kfpos_set_height(kfpos_handle *h, const double *z);

/* getPose for every tag (KalmanFilterTOA.cpp:438-473, KalmanFilterTOAIMU.cpp:476-510): predict-only
 * extrapolation by dt_ahead, filter state untouched. pos n_tags x 3, cov3x3 n_tags x 9 (position block of
 * the predicted covariance; the only block stateToPose fills, KalmanFilterTOA.cpp:159-183), vel n_tags x 3
 * (linearSpeed*, 9-state; zeros for 6-state), all double; any may be NULL. A tag without a measurement
 * yet reports KFPOS_ST_NOT_STARTED and NaN (getPose() == false). */
int kfpos_get_pose(kfpos_handle *h, double dt_ahead, double *pos, double *cov3x3, double *vel,
                   uint32_t *status);
/* Same with one extrapolation time per tag (dt_ahead: n_tags doubles): tags report asynchronously, so
 * "now minus the time of my last estimate" differs from tag to tag (KalmanFilterTOA.cpp:451-452). */
int kfpos_get_pose_each(kfpos_handle *h, const double *dt_ahead, double *pos, double *cov3x3, double *vel,
                        uint32_t *status);

/* The whole predict-only extrapolation getPose computes before stateToPose thins it out
 * (KalmanFilterTOA.cpp:455-468, KalmanFilterTOAIMU.cpp:492-506): x n_tags x n, P n_tags x n x n row-major,
 * double; dt_len = 1 (shared) or n_tags. Tags without a measurement yet report KFPOS_ST_NOT_STARTED.
 * Used by the adaptor to fill Vector3::covarianceMatrix exactly as stateToPose does. */
int kfpos_get_predicted(kfpos_handle *h, const double *dt_ahead, int32_t dt_len, double *x, double *P,
                        uint32_t *status);

/* Raw filter members for tests and checkpoint/restore: x n_tags x n ([p, v(, a = 0)]; planar:
 * [x y vx vy 0 0 theta omega]), P n_tags x n x n row-major, double. n = kfpos_state_dim(). flags: n_tags
 * words (bit 0 started, bit 1 has latched IMU; planar: bits 5..7 latched PX4Flow / IMU / magnetometer), may
 * be NULL. */
int kfpos_state_dim(const kfpos_handle *h);
int kfpos_get_state(kfpos_handle *h, double *x, double *P, uint32_t *flags);
int kfpos_set_state(kfpos_handle *h, const double *x, const double *P, const uint32_t *flags);
/* The latched sensor samples (lastImuMeasurement etc.), the rest of a checkpoint: n_tags x kfpos_latch_dim() doubles.
 * 9-state: 12 = linearAcceleration[3] + its covariance 3x3 row-major (KalmanFilterTOAIMU.h:58-59); planar: 15 =
 * PX4Flow {vx, vy, gyroz, covarianceVelocity, covarianceGyroZ}, IMU {ax, ay, angularVelocityZ,
 * covarianceAccelerationXY[4], covarianceAngularVelocityZ}, magnetometer {angle, covarianceMag}
 * (sensor_types.h:32-60); ALGORITHM_ML: 3 = _previousEstimation, the seed of every solve (MLLocation.cpp:3-22);
 * 6-state filter: 0, both calls are no-ops. Which samples are live is in the flags word. A complete checkpoint of a
 * handle = kfpos_get_state + flags, kfpos_get_latch and, for the planar filter, kfpos_get_height. */
int kfpos_latch_dim(const kfpos_handle *h);
int kfpos_get_latch(kfpos_handle *h, double *latch);
int kfpos_set_latch(kfpos_handle *h, const double *latch);

/* ---- streaming host API: epochs pipelined through kfpos_slot_count() slots of pinned host memory ----
 * For a node that feeds epoch after epoch from the CPU (PosGenerator's table flushes, Posgenerator.cpp:155-198, batched
 * over many tags): the synchronous calls above copy pageable arrays, turn their layout on the device and wait; here
 * the caller ASSEMBLES each epoch directly in a slot -- host-pinned memory owned by the handle, component-major like the
 * device layout ([max_anchors][n_tags] etc., see below) -- and submits it. A submission only enqueues: inputs go to the
 * GPU by DMA on a copy stream, the step runs on a compute stream, status words and the poses after the epoch come back
 * on a third; while earlier slots are in flight the caller fills the next one. Submissions execute in the order they
 * were made.
 *   kfpos_slot_acquire  waits until the slot's previous submission has completed (its outputs are then in the slot,
 *                       its inputs may be overwritten) and returns the slot's pointers
 *   kfpos_slot_submit   flags = one KFPOS_SLOT_* kind | options; dt_shared is used unless KFPOS_SLOT_DT_PER_TAG
 *   kfpos_slot_wait     waits for the slot's submission; status / pos of the slot are valid afterwards
 * The synchronous host API and the state accessors first wait for everything the slots have in flight. */
typedef struct kfpos_epoch_slot {
    int32_t  *range_mm; /* [max_anchors][n_tags] integer mm, <= 0 = absent */
    void     *err_est;  /* [max_anchors][n_tags] kfpos_real */
    void     *accel;    /* [3][n_tags] kfpos_real (9-state handles) */
    void     *cov;      /* [9][n_tags] kfpos_real, row-major 3x3 index first */
    double   *dt;       /* [n_tags], read with KFPOS_SLOT_DT_PER_TAG (negative = tag has no epoch in this submission) */
    uint32_t *status;   /* out: [n_tags] status words of the submission */
    double   *pos;      /* out: [3][n_tags] positions after the epoch (getPose at timeLag 0), unless KFPOS_SLOT_NO_POSE */
} kfpos_epoch_slot;
#define KFPOS_SLOT_TOA        0 /* kfpos_step_toa: a ranging epoch (a 9-state handle re-fuses its latched IMU sample) */
#define KFPOS_SLOT_IMU        1 /* kfpos_step_imu: latch accel / cov, IMU-only estimate */
#define KFPOS_SLOT_TOA_IMU    2 /* kfpos_step_toa_imu: latch + ranging epoch */
#define KFPOS_SLOT_DT_PER_TAG 0x100 /* dt comes from the slot's dt array instead of dt_shared */
#define KFPOS_SLOT_REUSE_ERR  0x200 /* errorEstimations are those of the previous submission: not copied again */
#define KFPOS_SLOT_REUSE_COV  0x400 /* the accelerometer covariance is that of the previous submission (a sensor with a
                                       fixed covariance): not copied again */
#define KFPOS_SLOT_NO_POSE    0x800 /* do not write / return pos */
int kfpos_slot_count(const kfpos_handle *h); /* 3: one being filled, one on the bus, one computing / returning */
int kfpos_slot_acquire(kfpos_handle *h, int32_t slot, kfpos_epoch_slot *out);
int kfpos_slot_submit(kfpos_handle *h, int32_t slot, int32_t flags, double dt_shared);
int kfpos_slot_wait(kfpos_handle *h, int32_t slot);

/* ---- asynchronous device-buffer API (inputs already resident in HBM) ----
 * All pointers are device pointers; `stream` is a hipStream_t (NULL = the default stream). Calls
 * enqueue work and return; the caller synchronises. Device layouts are component-major so that the
 * 64 lanes of a wavefront read 64 consecutive elements:
 *   range_mm [max_anchors][n_tags] int32      err_est [max_anchors][n_tags] kfpos_real
 *   accel    [3][n_tags] kfpos_real           cov     [9][n_tags] kfpos_real (row-major 3x3 index first)
 *   dt       [n_tags] double, or NULL to use the shared dt_shared
 *   status   [n_tags] uint32 or NULL
 *   pos      [3][n_tags] double               cov3x3  [9][n_tags] double      vel [3][n_tags] double */
int kfpos_step_toa_dev(kfpos_handle *h, const int32_t *range_mm, const void *err_est,
                       const double *dt, double dt_shared, uint32_t *status, void *stream);
int kfpos_step_imu_dev(kfpos_handle *h, const void *accel, const void *cov,
                       const double *dt, double dt_shared, uint32_t *status, void *stream);
/* latch != 0 also stores the sample in the handle (needed if a later kfpos_step_toa* call is to
 * re-fuse it); latch = 0 skips that write when every epoch brings its own sample. */
int kfpos_step_toa_imu_dev(kfpos_handle *h, const int32_t *range_mm, const void *err_est,
                           const void *accel, const void *cov, int32_t latch,
                           const double *dt, double dt_shared, uint32_t *status, void *stream);
/* kfpos_step_sensor on device buffers: data is component-major [C][n_tags] doubles */
int kfpos_step_sensor_dev(kfpos_handle *h, int32_t kind, const double *data,
                          const double *dt, double dt_shared, uint32_t *status, void *stream);
int kfpos_get_pose_dev(kfpos_handle *h, double dt_ahead, double *pos, double *cov3x3, double *vel,
                       uint32_t *status, void *stream);

/* Replay a whole trace resident in HBM: n_steps epochs, epoch s reading range_mm + s*stride_ranges
 * elements etc. (strides in elements; err_est / cov strides may be 0 to reuse one array). Equivalent, bit
 * for bit, to n_steps calls of kfpos_step_toa_dev / kfpos_step_toa_imu_dev (accel != NULL) with
 * dt_steps[s] (host array) shared by all tags -- but up to 128 epochs run inside ONE launch with every
 * tag's state resident in registers, so state traffic and launch boundaries are paid once per launch
 * instead of once per epoch (KFPOS_TRACE_CHUNK_STEPS=n in the environment caps the epochs per launch;
 * 1 = one launch per epoch).
 *   trajectory  [n_steps][3][n_tags] double or NULL: the position after every epoch, i.e. what a caller
 *               polling getPose at timeLag 0 after each epoch would have read (Posgenerator.cpp:541-548)
 *   status      [n_tags] status words of the LAST epoch, or NULL */
int kfpos_run_trace_dev(kfpos_handle *h, int32_t n_steps,
                        const int32_t *range_mm, int64_t stride_ranges,
                        const void *err_est, int64_t stride_err,
                        const void *accel, int64_t stride_accel,
                        const void *cov, int64_t stride_cov,
                        const double *dt_steps, double *trajectory, uint32_t *status, void *stream);

/* ---- multi-GPU: contiguous tag shards + ONE collective, the RCCL all-gather of poses ----
 * The reference runs one filter in one process (node_pos.cpp:176-181) and has no counterpart. Here a node that serves
 * more tags than one GPU holds cuts the batch into contiguous ranges, one handle per GPU; the filters never talk to each
 * other, and what is published -- the pose of EVERY tag (Posgenerator.cpp:541-548, batched) -- is brought together by one
 * ncclAllGather per epoch (or per launch of K epochs) over xGMI, on a side stream, double-buffered, so that it overlaps
 * the next epoch's compute. librccl is opened on first use; a process that never calls kfpos_comm_* does not load it.
 *
 * One process per GPU (MPI / a launcher of your own / torchrun):
 *     rank 0:     kfpos_comm_unique_id(id);  ...ship the 128 bytes to every rank by whatever you bootstrap with...
 *     every rank: kfpos_comm_create(world, rank, id, device, &comm);
 *                 kfpos_comm_set_total(comm, total_tags, &lo, &hi);   -> this rank's handle holds tags [lo, hi)
 *                 per epoch: kfpos_step_*_dev(h, ...); kfpos_allgather_poses(h, comm, NULL, 3, pos_all, stream);
 *                 before reading pos_all: kfpos_comm_wait(comm, stream)  (or kfpos_comm_sync(comm) on the host)
 * One process driving n GPUs from one thread:
 *     kfpos_comm_create_all(n, devices, comms);  then kfpos_allgather_poses_multi(...) once per epoch for all of them.
 */
#define KFPOS_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
typedef struct kfpos_comm kfpos_comm;

/* Contiguous shard [lo, hi) of `rank` out of `world`: sizes differ by at most one tag, the first total % world ranks
 * hold the extra one. Pure host arithmetic (no GPU, no RCCL). */
int kfpos_shard_range(int64_t total_tags, int32_t world, int32_t rank, int64_t *lo, int64_t *hi);

/* ncclGetUniqueId: 128 bytes that rank 0 creates and every rank passes to kfpos_comm_create. */
int kfpos_comm_unique_id(void *id_out);
/* ncclCommInitRank on `device` (collective: returns when all `world` ranks have called it). */
int kfpos_comm_create(int32_t world, int32_t rank, const void *unique_id, int32_t device, kfpos_comm **out);
/* ncclCommInitAll: n communicators of one clique for one process; devices = NULL means 0..n-1. out: n pointers. */
int kfpos_comm_create_all(int32_t n_devices, const int32_t *devices, kfpos_comm **out);
int kfpos_comm_destroy(kfpos_comm *c);
int kfpos_comm_world(const kfpos_comm *c);
int kfpos_comm_rank(const kfpos_comm *c);
/* The size of the whole batch; fixes this rank's shard (returned in lo / hi, may be NULL) and the padded block size of
 * the collective. Call before the first gather, on every rank with the same total (>= world). */
int kfpos_comm_set_total(kfpos_comm *c, int64_t total_tags, int64_t *lo, int64_t *hi);

/* All-gather of pose blocks (device pointers, component-major, double):
 *   pos_local  [rows][t_local] this rank's block, or NULL with rows = 3 for the handle's current positions (what
 *              getPose at timeLag 0 returns for every tag of the shard); rows = 3 for one epoch, 3 K for the K epochs
 *              of a kfpos_run_trace_dev trajectory. It is copied on `stream` before the call returns to the caller,
 *              so it may be overwritten by later work on `stream` at once
 *   pos_all    [rows][total_tags] in global tag order, written on the communicator's side stream: valid after
 *              kfpos_comm_wait (device-side: `stream` waits for the last gather) or kfpos_comm_sync (host blocks)
 *   h          the shard's handle (checked against the communicator's shard size and device), or NULL when pos_local is given
 * Unequal shards are padded to the largest inside the call (an all-gather wants equal contributions) and the padding
 * is dropped again when pos_all is assembled. Two gathers may be in flight: the third waits for the first. */
int kfpos_allgather_poses(kfpos_handle *h, kfpos_comm *c, const double *pos_local, int32_t rows, double *pos_all,
                          void *stream);
/* The same for n communicators of one process (kfpos_comm_create_all) in ONE RCCL group; arrays of n entries;
 * handles, pos_local and streams may be NULL (= all NULL). */
int kfpos_allgather_poses_multi(int32_t n, kfpos_handle *const *handles, kfpos_comm *const *comms,
                                const double *const *pos_local, int32_t rows, double *const *pos_all,
                                void *const *streams);
/* How the blocks of a gather travel. COLLECTIVE: ncclAllGather, RCCL's own schedule. DIRECT: every rank sends its block
 * to each other rank and receives theirs in one RCCL group (ncclSend / ncclRecv) -- on MI355X every pair of GPUs has its
 * own xGMI link, so a rank's world-1 transfers run side by side, one hop each, where a ring passes every block through
 * world-1 hops. Same result, same buffers; which is faster at a given size is for the machine to say (bench.py times
 * both). All ranks must choose alike; default COLLECTIVE, or KFPOS_GATHER_ALGO=direct in the environment at creation. */
#define KFPOS_GATHER_COLLECTIVE 0
#define KFPOS_GATHER_DIRECT     1
int kfpos_comm_set_algorithm(kfpos_comm *c, int32_t algorithm);
int kfpos_comm_algorithm(const kfpos_comm *c);
int kfpos_comm_wait(kfpos_comm *c, void *stream);
int kfpos_comm_sync(kfpos_comm *c);
/* The assembly step alone, for a caller that gathers with a collective of its own (torch.distributed, MPI):
 * staged [world][rows][t_pad] (t_pad = largest shard of kfpos_shard_range) -> out [rows][total_tags]. */
int kfpos_assemble_poses_dev(int32_t world, int32_t rows, int64_t total_tags, const double *staged, double *out,
                             int32_t device, void *stream);
/* NCCL version code of the RCCL that was opened (e.g. 22707), 0 if none could be */
int kfpos_comm_backend_version(void);

/* ---- diagnostics ---- */
const char *kfpos_last_error(void);  /* thread-local detail of the last error (HIP failures, rejected configurations, calls out of sequence) */
const char *kfpos_strerror(int code);
int kfpos_version(void);
/* name / duration helpers for benchmarks: time the last n enqueued step kernels with HIP events on the
 * stream they were launched on. kfpos_timing_begin records, kfpos_timing_end synchronises and returns ms. */
int kfpos_timing_begin(kfpos_handle *h, void *stream);
int kfpos_timing_end(kfpos_handle *h, void *stream, float *elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* KFPOS_H */

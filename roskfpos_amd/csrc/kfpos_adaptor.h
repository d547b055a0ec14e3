/*
 * kfpos_adaptor.h -- host C++ mirror of the reference estimator classes on top of the C ABI.
 *
 * A PosGenerator-like caller keeps its code: same class names, constructor arguments, the seven
 * virtual methods of PositionEstimationAlgorithm and their argument meaning
 *   PositionEstimationAlgorithm   src/kfpos/algorithms/PositionEstimationAlgorithm.h:8-37
 *   KalmanFilterTOA               src/kfpos/algorithms/KalmanFilterTOA.h:18-54
 *   KalmanFilterTOAIMU            src/kfpos/algorithms/KalmanFilterTOAIMU.h:16-78
 *   Vector3 / VectorDim3 / Beacon src/kfpos/algorithms/sensor_types.h:7-25
 * Differences, all at the type level: Vector3::covarianceMatrix is a plain row-major array with its
 * dimension (the reference embeds an arma::mat); the estimator reads time from an injectable clock
 * (default std::chrono::steady_clock, as KalmanFilterTOA.cpp:76-88 does) so tests can replay traces.
 *
 * Each object is ONE filter (n_tags = 1), which is what the reference node runs; batched callers use
 * the C ABI (include/kfpos.h) or BatchedEstimator below directly. Header-only; link libkfpos_hip.so.
 */
#ifndef KFPOS_ADAPTOR_H
#define KFPOS_ADAPTOR_H

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "kfpos.h"

namespace kfpos_host {

struct Vector3 { /* sensor_types.h:7-13 */
    double x = 0, y = 0, z = 0;
    double rotX = 0, rotY = 0, rotZ = 0, rotW = 0;
    double linearSpeedX = 0, linearSpeedY = 0, linearSpeedZ = 0;
    double angularSpeedX = 0, angularSpeedY = 0, angularSpeedZ = 0;
    int covarianceDim = 0;             /* 6 (TOA) or 9 (TOA+IMU): stateToPose */
    double covarianceMatrix[81] = {0}; /* row-major covarianceDim x covarianceDim */
};
struct VectorDim3 { double x, y, z; };  /* sensor_types.h:15-17 */
struct Beacon {                          /* sensor_types.h:19-23 */
    int id;
    int index;
    Vector3 position;
};

/* seconds on a monotonic clock */
using Clock = std::function<double()>;
inline double steady_seconds() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

class PositionEstimationAlgorithm { /* PositionEstimationAlgorithm.h:8-37 */
public:
    virtual ~PositionEstimationAlgorithm() {}
    virtual bool init() { return true; }
    virtual bool getPose(Vector3 &) { return false; }
    virtual void newPX4FlowMeasurement(double, double, double, double, int) {}
    virtual void newTOAMeasurement(const std::vector<double> &, const std::vector<Beacon> &,
                                   const std::vector<double> &, double) {}
    virtual void newIMUMeasurement(VectorDim3, double[9], VectorDim3, double[9]) {}
    virtual void newMAGMeasurement(VectorDim3, double[9]) {}
    virtual void newCompassMeasurement(double) {}
};

/* Shared plumbing of the two filters: one handle with n_tags = 1. */
class SingleTagFilter : public PositionEstimationAlgorithm {
public:
    ~SingleTagFilter() override {
        if (h_) kfpos_destroy(h_);
    }
    bool init() override { return h_ && kfpos_init(h_) == KFPOS_OK; }
    void setClock(Clock c) { clock_ = std::move(c); }
    uint32_t lastStatus() const { return status_; }
    kfpos_handle *handle() { return h_; }

    /* newTOAMeasurement: ranges in metres as PosGenerator passes them ((double) mm / 1000,
     * Posgenerator.cpp:484); the estimator drops entries <= 0 (KalmanFilterTOA.cpp:50). The timeLag
     * argument is ignored, as in the reference (KalmanFilterTOA.cpp:43-60). */
    void newTOAMeasurement(const std::vector<double> &rangings, const std::vector<Beacon> &beacons,
                           const std::vector<double> &errorEstimations, double /*timeLag*/) override {
        const int n = (int)rangings.size();
        if (n > KFPOS_MAX_ANCHORS) throw std::invalid_argument("more than 64 ranges in one epoch");
        /* the beacon set may change from epoch to epoch: re-send the table, column i = beacon i */
        double xyz[KFPOS_MAX_ANCHORS * 3] = {0};
        int32_t mm[KFPOS_MAX_ANCHORS] = {0};
        double err[KFPOS_MAX_ANCHORS];
        for (int i = 0; i < KFPOS_MAX_ANCHORS; ++i) err[i] = 1.0;
        for (int i = 0; i < n; ++i) {
            xyz[3 * i] = beacons[i].position.x;
            xyz[3 * i + 1] = beacons[i].position.y;
            xyz[3 * i + 2] = beacons[i].position.z;
            /* back to the node's integer millimetres: exact, the metres value came from mm / 1000 */
            mm[i] = rangings[i] > 0 ? (int32_t)std::llround(rangings[i] * 1000.0) : 0;
            err[i] = errorEstimations[i];
        }
        check(kfpos_set_anchors(h_, xyz, nullptr, n > 0 ? n : 1));
        const double dt = lag();
        check(kfpos_step_toa(h_, mm, err, &dt, 1, &status_));
    }

    bool getPose(Vector3 &pose) override {
        if (!started_) return false; /* KalmanFilterTOA.cpp:442-447 */
        const double ahead = clock_() - last_;
        double x[9], P[81];
        uint32_t st = 0;
        check(kfpos_get_predicted(h_, &ahead, 1, x, P, &st)); /* predicted state and covariance, :455-468 */
        fillPose(pose, x, P);
        return true;
    }

protected:
    SingleTagFilter(int model, double accelNoise, double jolt, bool ignoreWorst, double costThreshold,
                    bool fixed, const Vector3 *init) {
        kfpos_config c;
        std::memset(&c, 0, sizeof(c));
        c.model = model;
        c.n_tags = 1;
        c.max_anchors = KFPOS_MAX_ANCHORS;
        c.storage = KFPOS_STORE_F64;
        c.accel_noise = accelNoise;
        c.jolt = jolt;
        c.ignore_worst = ignoreWorst ? 1 : 0;
        c.cost_threshold = costThreshold;
        c.use_init_pos = fixed ? 1 : 0;
        if (init) { c.init_pos[0] = init->x; c.init_pos[1] = init->y; c.init_pos[2] = init->z; }
        check(kfpos_create(&c, &h_));
        clock_ = steady_seconds;
    }
    /* wall-clock timeLag: 0.1 on the first call (KalmanFilterTOA.cpp:78-88) */
    double lag() {
        const double now = clock_();
        const double dt = started_ ? now - last_ : 0.1;
        last_ = now;
        started_ = true;
        return dt;
    }
    static void check(int rc) {
        if (rc != KFPOS_OK)
            throw std::runtime_error(std::string("kfpos: ") + kfpos_strerror(rc) + " " + kfpos_last_error());
    }
    /* stateToPose: x = predicted state (n), P = predicted covariance (n x n row-major) */
    virtual void fillPose(Vector3 &pose, const double *x, const double *P) = 0;

    kfpos_handle *h_ = nullptr;
    Clock clock_;
    bool started_ = false;
    double last_ = 0.0;
    uint32_t status_ = 0;
};

/* ALGORITHM_KF_TOA */
class KalmanFilterTOA : public SingleTagFilter {
public:
    KalmanFilterTOA(double accelerationNoise, bool ignoreWorstAnchorMode, double ignoreCostThreshold)
        : SingleTagFilter(KFPOS_MODEL_TOA, accelerationNoise, 0.0, ignoreWorstAnchorMode, ignoreCostThreshold,
                          false, nullptr) {}
    KalmanFilterTOA(double accelerationNoise, bool ignoreWorstAnchorMode, double ignoreCostThreshold,
                    Vector3 initialPosition)
        : SingleTagFilter(KFPOS_MODEL_TOA, accelerationNoise, 0.0, ignoreWorstAnchorMode, ignoreCostThreshold,
                          true, &initialPosition) {}
    /* the other four sensors are no-ops in this filter (KalmanFilterTOA.cpp:63-66) */
protected:
    void fillPose(Vector3 &pose, const double *x, const double *P) override {
        pose = Vector3(); /* stateToPose, KalmanFilterTOA.cpp:159-183: zero quaternion, 6x6 with the position block */
        pose.x = x[0]; pose.y = x[1]; pose.z = x[2];
        pose.covarianceDim = 6;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) pose.covarianceMatrix[i * 6 + j] = P[i * 6 + j];
    }
};

/* ALGORITHM_KF_TOA_IMU */
class KalmanFilterTOAIMU : public SingleTagFilter {
public:
    KalmanFilterTOAIMU(double accelerationNoise, double jolt)
        : SingleTagFilter(KFPOS_MODEL_TOA_IMU, accelerationNoise, jolt, false, 0.0, false, nullptr) {}
    KalmanFilterTOAIMU(double accelerationNoise, double jolt, Vector3 initialPosition)
        : SingleTagFilter(KFPOS_MODEL_TOA_IMU, accelerationNoise, jolt, false, 0.0, true, &initialPosition) {}

    /* KalmanFilterTOAIMU.cpp:76-92: angular velocity is ignored, the acceleration sample is latched
     * and an IMU-only estimate runs */
    void newIMUMeasurement(VectorDim3, double[9], VectorDim3 linearAcceleration,
                           double covarianceAcceleration[9]) override {
        const double acc[3] = {linearAcceleration.x, linearAcceleration.y, linearAcceleration.z};
        const double dt = lag();
        check(kfpos_step_imu(h_, acc, covarianceAcceleration, &dt, 1, &status_));
    }

protected:
    void fillPose(Vector3 &pose, const double *x, const double *P) override {
        pose = Vector3(); /* stateToPose, KalmanFilterTOAIMU.cpp:198-240 */
        pose.x = x[0]; pose.y = x[1]; pose.z = x[2];
        pose.linearSpeedX = x[3]; pose.linearSpeedY = x[4]; pose.linearSpeedZ = x[5];
        /* sic: the acceleration state is reported in the angularSpeed fields (:213-215); it is never
         * persisted, so the predicted value is 0 */
        pose.angularSpeedX = x[6]; pose.angularSpeedY = x[7]; pose.angularSpeedZ = x[8];
        pose.covarianceDim = 9;
        for (int i = 0; i < 9; ++i) pose.covarianceMatrix[i * 9 + i] = 0.01; /* eye(9,9) * 0.01, :217 */
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) pose.covarianceMatrix[i * 9 + j] = P[i * 9 + j];
        for (int i = 0; i < 3; ++i) { /* sic: row/column 7 receive P(.,8) (:231-239) */
            pose.covarianceMatrix[i * 9 + 7] = P[i * 9 + 8];
            pose.covarianceMatrix[7 * 9 + i] = P[8 * 9 + i];
        }
        pose.covarianceMatrix[7 * 9 + 7] = P[8 * 9 + 8];
    }
};

} // namespace kfpos_host
#endif

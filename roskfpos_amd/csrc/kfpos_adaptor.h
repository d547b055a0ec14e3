/*
 * kfpos_adaptor.h -- host C++ mirror of the reference estimator classes on top of the C ABI.
 *
 * A PosGenerator-like caller keeps its code: same class names, constructor arguments, the seven
 * virtual methods of PositionEstimationAlgorithm and their argument meaning
 *   PositionEstimationAlgorithm   src/kfpos/algorithms/PositionEstimationAlgorithm.h:8-37
 *   KalmanFilterTOA               src/kfpos/algorithms/KalmanFilterTOA.h:18-54
 *   KalmanFilterTOAIMU            src/kfpos/algorithms/KalmanFilterTOAIMU.h:16-78
 *   KalmanFilter                  src/kfpos/algorithms/KalmanFilter.h:29-133 (8-state planar filter, ALGORITHM_KF)
 *   MLLocation                    src/kfpos/algorithms/MLLocation.h:26-70 as an estimator (ALGORITHM_ML; 3-D, variants 0, 1 and -- up to 5 beacons -- 2)
 *   Vector3 / VectorDim3 / Beacon src/kfpos/algorithms/sensor_types.h:7-25
 * Differences, all at the type level: Vector3::covarianceMatrix is a fixed-capacity matrix with the arma::mat
 * accessors the node uses (the reference embeds an arma::mat); the estimator reads time from an injectable clock
 * (default std::chrono::steady_clock, as KalmanFilterTOA.cpp:76-88 does) so tests can replay traces.
 *
 * Each object is ONE filter (n_tags = 1), which is what the reference node runs; batched callers use
 * the C ABI (include/kfpos.h) or BatchedEstimator below directly. Header-only; link libkfpos_hip.so.
 */
#ifndef KFPOS_ADAPTOR_H
#define KFPOS_ADAPTOR_H

#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "kfpos.h"

namespace kfpos_host {

/* The slice of arma::mat that code holding a Vector3 uses on covarianceMatrix (sensor_types.h:12): element access
 * (i, j), Armadillo's column-major LINEAR access (i) -- how publishPositionReport copies the first 36 elements
 * into the ROS messages, Posgenerator.cpp:397-399, 448-449 --, n_rows / n_cols / n_elem, zeros() / eye(). Fixed
 * capacity 9 x 9 (the largest stateToPose fills, KalmanFilterTOAIMU.cpp:217), no heap. Out-of-range access throws
 * std::logic_error, as Armadillo's default bounds check does. */
struct CovarianceMatrix {
    int n_rows, n_cols, n_elem;
    double mem[81]; /* column-major, like arma::mat::mem */
    CovarianceMatrix() : n_rows(0), n_cols(0), n_elem(0) { std::memset(mem, 0, sizeof(mem)); }
    CovarianceMatrix(int rows, int cols) { zeros(rows, cols); }
    void zeros(int rows, int cols) {
        if (rows < 0 || cols < 0 || rows * cols > 81) throw std::logic_error("CovarianceMatrix: at most 81 elements");
        n_rows = rows; n_cols = cols; n_elem = rows * cols;
        std::memset(mem, 0, sizeof(mem));
    }
    void eye(int rows, int cols, double scale = 1.0) { /* arma::eye<arma::mat>(rows, cols) * scale */
        zeros(rows, cols);
        for (int i = 0; i < rows && i < cols; ++i) mem[i * rows + i] = scale;
    }
    double &operator()(int i) { return mem[lin(i)]; }
    const double &operator()(int i) const { return mem[lin(i)]; }
    double &operator()(int r, int c) { return mem[at(r, c)]; }
    const double &operator()(int r, int c) const { return mem[at(r, c)]; }

private:
    int lin(int i) const {
        if (i < 0 || i >= n_elem) throw std::logic_error("Mat::operator(): index out of bounds");
        return i;
    }
    int at(int r, int c) const {
        if (r < 0 || c < 0 || r >= n_rows || c >= n_cols) throw std::logic_error("Mat::operator(): index out of bounds");
        return c * n_rows + r;
    }
};

/* sensor_types.h:7-13. An aggregate with the reference's member order and no default member initialisers, so that
 * `Vector3 pose = {NAN, NAN, NAN};` (Posgenerator.cpp:542) is aggregate initialisation under the reference's
 * -std=c++11 (CMakeLists.txt:7-9): the members not named are value-initialised (0, empty matrix). */
struct Vector3 {
    double x, y, z;
    double rotX, rotY, rotZ, rotW;
    double linearSpeedX, linearSpeedY, linearSpeedZ;
    double angularSpeedX, angularSpeedY, angularSpeedZ;
    CovarianceMatrix covarianceMatrix;
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
                    bool fixed, const Vector3 *init, int topN = 0, int mlVariant = KFPOS_ML_NORMAL, bool use2d = false,
                    int variantAsGiven = 0) {
        if (use2d) throw std::invalid_argument("MLLocation: use2d has no defined result in the reference (getPose, MLLocation.cpp:456-464)");
        if (variantAsGiven < 0 || variantAsGiven > 2) throw std::invalid_argument("MLLocation: variant is not one of ML_VARIANT_*");
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
        c.top_n = topN;
        c.ml_variant = mlVariant;
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
        pose.covarianceMatrix.zeros(6, 6);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) pose.covarianceMatrix(i, j) = P[i * 6 + j];
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
        pose.covarianceMatrix.eye(9, 9, 0.01); /* eye(9,9) * 0.01, :217 */
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) pose.covarianceMatrix(i, j) = P[i * 9 + j];
        for (int i = 0; i < 3; ++i) { /* sic: row/column 7 receive P(.,8) (:231-239) */
            pose.covarianceMatrix(i, 7) = P[i * 9 + 8];
            pose.covarianceMatrix(7, i) = P[8 * 9 + i];
        }
        pose.covarianceMatrix(7, 7) = P[8 * 9 + 8];
    }
};

/* ---- ALGORITHM_ML: MLLocation as the estimator (MLLocation.cpp:421-486) ---- */
#ifndef ML_VARIANT_NORMAL /* MLLocation.h:5-7; kept as macros so that node code using them compiles unchanged */
#define ML_VARIANT_NORMAL 0
#define ML_VARIANT_IGNORE_N 1
#define ML_VARIANT_BEST 2
#endif

class MLLocation : public SingleTagFilter {
public:
    /* _previousEstimation = {1,1,4} (MLLocation.cpp:3-12) */
    MLLocation() : SingleTagFilter(KFPOS_MODEL_ML, 0.0, 0.0, false, 0.0, false, nullptr) {}
    /* MLLocation.cpp:14-22. Offered: the 3-D solver, variants NORMAL, IGNORE_N and BEST. The 2-D variant indexes (0,2)
     * of a 2x2 covariance in getPose (:456-464): no defined result in the reference, refused here. BEST is defined for 4
     * or 5 ranges (its erase loop, :377-381, runs past the end of the vector from 6 on): newTOAMeasurement with more
     * than 5 beacons throws std::runtime_error (kfpos_set_anchors: KFPOS_ERR_MODEL). */
    MLLocation(bool use2d, int variant, int numRangingsToIgnore, const Vector3 &previousEstimation)
        : SingleTagFilter(KFPOS_MODEL_ML, 0.0, 0.0, false, 0.0, true, &previousEstimation,
                          variant == ML_VARIANT_IGNORE_N ? numRangingsToIgnore : 0,
                          variant == ML_VARIANT_BEST ? KFPOS_ML_BEST : KFPOS_ML_NORMAL, use2d, variant) {}
    /* getPose solves from the stored ranges and the fixed seed every time (:426-441): no extrapolation, no clock.
     * Before the first ranging epoch the reference indexes an empty covariance; here getPose returns false. */
    bool getPose(Vector3 &pose) override {
        if (!started_) return false;
        double pos[3], cov[9], vel[3];
        uint32_t st = 0;
        check(kfpos_get_pose(h_, 0.0, pos, cov, vel, &st));
        if (st & KFPOS_ST_NOT_STARTED) return false;
        pose = Vector3();
        pose.x = pos[0]; pose.y = pos[1]; pose.z = pos[2];
        pose.covarianceMatrix.zeros(6, 6); /* eye(6,6) * 0 with the 3x3 block, :449-464 */
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) pose.covarianceMatrix(i, j) = cov[3 * i + j];
        return true;
    }

protected:
    void fillPose(Vector3 &, const double *, const double *) override {}
};

/* ---- ALGORITHM_KF: the 8-state planar filter ---- */

/* The reference reads its configuration from five ROS parameters whose VALUES are XML documents
 * (node_pos.cpp:111: "configPos", "configPX4Flow", "configUWB", "configIMU", "configMAG"; the launch files fill
 * them with <param textfile=.../>). A ParamSource stands for NodeHandle::getParam(name, content). */
using ParamSource = std::function<bool(const std::string &name, std::string &content)>;

/* ParamSource over a name -> file path table (what <param name=... textfile=.../> does) */
inline ParamSource fileParamSource(std::vector<std::pair<std::string, std::string>> table) {
    return [table](const std::string &name, std::string &content) {
        for (const auto &kv : table)
            if (kv.first == name) {
                std::ifstream f(kv.second);
                if (!f) return false;
                std::stringstream ss;
                ss << f.rdbuf();
                content = ss.str();
                return true;
            }
        return false;
    };
}

/* The slice of boost::property_tree's XML reader the reference relies on (KalmanFilter.cpp:763-842): children of
 * <config> named `element`, attribute lookup with a default, comments skipped, later elements overriding
 * earlier ones. Returns false when there is no <config> root (property_tree would throw -> init() false). */
class XmlAttributes {
public:
    bool parse(const std::string &xml) {
        text_.clear();
        for (size_t i = 0; i < xml.size();) { /* drop <!-- ... --> */
            if (xml.compare(i, 4, "<!--") == 0) {
                const size_t e = xml.find("-->", i + 4);
                if (e == std::string::npos) break;
                i = e + 3;
            } else {
                text_ += xml[i++];
            }
        }
        return text_.find("<config") != std::string::npos;
    }
    /* value of `attr` in the last <element .../>, or false if absent */
    bool find(const std::string &element, const std::string &attr, std::string &out) const {
        bool found = false;
        size_t pos = 0;
        const std::string open = "<" + element;
        while ((pos = text_.find(open, pos)) != std::string::npos) {
            const size_t after = pos + open.size();
            pos = after;
            if (after >= text_.size() || !(std::isspace((unsigned char)text_[after]) || text_[after] == '/' || text_[after] == '>'))
                continue;
            const size_t end = text_.find('>', after);
            const std::string body = text_.substr(after, end == std::string::npos ? std::string::npos : end - after);
            size_t a = 0;
            while ((a = body.find(attr, a)) != std::string::npos) {
                const bool starts = a == 0 || std::isspace((unsigned char)body[a - 1]);
                size_t q = a + attr.size();
                while (q < body.size() && std::isspace((unsigned char)body[q])) ++q;
                if (starts && q < body.size() && body[q] == '=') {
                    ++q;
                    while (q < body.size() && std::isspace((unsigned char)body[q])) ++q;
                    if (q < body.size() && (body[q] == '"' || body[q] == '\'')) {
                        const char quote = body[q];
                        const size_t e = body.find(quote, q + 1);
                        if (e != std::string::npos) {
                            out = body.substr(q + 1, e - q - 1);
                            found = true;
                        }
                    }
                    break;
                }
                a += attr.size();
            }
        }
        return found;
    }
    double getDouble(const std::string &element, const std::string &attr, double dflt) const {
        std::string v;
        return find(element, attr, v) ? std::strtod(v.c_str(), nullptr) : dflt;
    }
    int getInt(const std::string &element, const std::string &attr, int dflt) const {
        std::string v;
        return find(element, attr, v) ? (int)std::strtol(v.c_str(), nullptr, 10) : dflt;
    }

private:
    std::string text_;
};

class KalmanFilter : public SingleTagFilter { /* KalmanFilter.h:29-133 */
public:
    KalmanFilter(double accelerationNoise, double initialAngle, double jolt, std::string filenamePos,
                 std::string filenamePX4Flow, std::string filenameTag, std::string filenameImu, std::string filenameMag)
        : SingleTagFilter(KFPOS_MODEL_PLANAR, accelerationNoise, jolt, false, 0.0, false, nullptr),
          names_{filenamePos, filenamePX4Flow, filenameTag, filenameImu, filenameMag} {
        std::memset(&cfg_, 0, sizeof(cfg_));
        cfg_.init_angle = initialAngle;
    }
    KalmanFilter(double accelerationNoise, double initialAngle, double jolt, std::string filenamePos,
                 std::string filenamePX4Flow, std::string filenameTag, std::string filenameImu, std::string filenameMag,
                 Vector3 initialPosition)
        : SingleTagFilter(KFPOS_MODEL_PLANAR, accelerationNoise, jolt, false, 0.0, true, &initialPosition),
          names_{filenamePos, filenamePX4Flow, filenameTag, filenameImu, filenameMag} {
        std::memset(&cfg_, 0, sizeof(cfg_));
        cfg_.init_angle = initialAngle;
    }
    /* where init() finds the five XML documents; default: none, init() then returns false as the reference does
     * when a parameter is missing (property_tree throws on the empty document, KalmanFilter.cpp:890-892) */
    void setParamSource(ParamSource src) { source_ = std::move(src); }
    const kfpos_planar_config &configuration() const { return cfg_; }

    /* loadConfigurationFiles, KalmanFilter.cpp:748-897 */
    bool init() override {
        if (!h_ || !source_) return false;
        XmlAttributes px4, uwb, imu, mag, pos;
        std::string content;
        if (!source_(names_[1], content) || !px4.parse(content)) return false;
        if (!source_(names_[2], content) || !uwb.parse(content)) return false;
        if (!source_(names_[3], content) || !imu.parse(content)) return false;
        if (!source_(names_[4], content) || !mag.parse(content)) return false;
        if (!source_(names_[0], content) || !pos.parse(content)) return false; /* read, nothing used (:849-887) */
        cfg_.px4_height = px4.getDouble("px4flow", "sensorHeight", 0);
        cfg_.px4_arm_p1 = px4.getDouble("px4flow", "armP0", 0);
        cfg_.px4_arm_p2 = px4.getDouble("px4flow", "armP1", 0);
        cfg_.px4_cov_velocity = px4.getDouble("px4flow", "covarianceVelocity", 0);
        cfg_.px4_cov_gyro_z = px4.getDouble("px4flow", "covarianceGyroZ", 0);
        cfg_.use_fixed_height = uwb.getInt("uwb", "useFixedHeight", 0) == 1;
        cfg_.fixed_height = uwb.getDouble("uwb", "fixedHeight", 0);
        cfg_.imu_use_fixed_cov_acc = imu.getInt("imu", "useFixedCovarianceAcceleration", 0) == 1;
        cfg_.imu_cov_acc = imu.getDouble("imu", "covarianceAcceleration", 0);
        cfg_.imu_use_fixed_cov_ang_vel_z = imu.getInt("imu", "useFixedCovarianceAngularVelocityZ", 0) == 1;
        cfg_.imu_cov_ang_vel_z = imu.getDouble("imu", "covarianceAngularVelocityZ", 0);
        cfg_.mag_angle_offset = mag.getDouble("mag", "angleOffset", 0);
        cfg_.mag_cov = mag.getDouble("mag", "covarianceMag", 0);
        return kfpos_set_planar(h_, &cfg_) == KFPOS_OK && kfpos_init(h_) == KFPOS_OK;
    }

    /* KalmanFilter.cpp:102-135. quality 0: the sample is dropped before the clock is read */
    void newPX4FlowMeasurement(double integrationX, double integrationY, double integrationRotationZ,
                               double integrationTime, int quality) override {
        if (quality == 0) return;
        const double f[5] = {integrationX, integrationY, integrationRotationZ, integrationTime, (double)quality};
        const double dt = lag();
        check(kfpos_step_sensor(h_, KFPOS_SENSOR_PX4FLOW, f, &dt, 1, &status_));
    }
    /* KalmanFilter.cpp:139-182 */
    void newIMUMeasurement(VectorDim3 angularVelocity, double covarianceAngularVelocity[9], VectorDim3 linearAcceleration,
                           double covarianceAcceleration[9]) override {
        double d[24] = {angularVelocity.x, angularVelocity.y, angularVelocity.z};
        std::memcpy(d + 3, covarianceAngularVelocity, 9 * sizeof(double));
        d[12] = linearAcceleration.x; d[13] = linearAcceleration.y; d[14] = linearAcceleration.z;
        std::memcpy(d + 15, covarianceAcceleration, 9 * sizeof(double));
        const double dt = lag();
        check(kfpos_step_sensor(h_, KFPOS_SENSOR_IMU, d, &dt, 1, &status_));
    }
    /* KalmanFilter.cpp:185-199: covarianceMag is ignored, the configured one is used */
    void newMAGMeasurement(VectorDim3 mag, double /*covarianceMag*/[9]) override {
        const double d[3] = {mag.x, mag.y, mag.z};
        const double dt = lag();
        check(kfpos_step_sensor(h_, KFPOS_SENSOR_MAG, d, &dt, 1, &status_));
    }
    /* KalmanFilter.cpp:201-229 */
    void newCompassMeasurement(double compass) override {
        const double dt = lag();
        check(kfpos_step_sensor(h_, KFPOS_SENSOR_COMPASS, &compass, &dt, 1, &status_));
    }

protected:
    void fillPose(Vector3 &pose, const double *x, const double *P) override {
        pose = Vector3(); /* stateToPose, KalmanFilter.cpp:324-363 */
        double z = 0.0;
        check(kfpos_get_height(h_, &z));
        pose.x = x[0]; pose.y = x[1]; pose.z = z;
        const double half = x[6] * 0.5;
        pose.rotZ = std::sin(half);
        pose.rotW = std::cos(half);
        pose.linearSpeedX = x[2]; pose.linearSpeedY = x[3];
        pose.angularSpeedZ = x[7];
        pose.covarianceMatrix.eye(6, 6, 0.01);
        pose.covarianceMatrix(0, 0) = P[0 * 8 + 0];
        pose.covarianceMatrix(0, 1) = P[0 * 8 + 1];
        pose.covarianceMatrix(1, 0) = P[1 * 8 + 0];
        pose.covarianceMatrix(1, 1) = P[1 * 8 + 1];
        pose.covarianceMatrix(0, 5) = P[0 * 8 + 6];
        pose.covarianceMatrix(1, 5) = P[1 * 8 + 6];
        pose.covarianceMatrix(5, 0) = P[6 * 8 + 0];
        pose.covarianceMatrix(5, 1) = P[6 * 8 + 1];
        pose.covarianceMatrix(5, 5) = P[6 * 8 + 6];
    }

private:
    std::string names_[5];
    ParamSource source_;
    kfpos_planar_config cfg_;
};

} // namespace kfpos_host
#endif

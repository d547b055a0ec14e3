/*
 * kfpos_ingest.h -- ranging ingest / epoch assembly for MANY tags in front of the batched core
 * (the step immediately before the hot path).
 *
 * Per tag this is PosGenerator's table logic, kept quirk for quirk:
 *   processRangingNow                src/kfpos/publishers/Posgenerator.cpp:201-281
 *   sendRangingMeasurementIfAvailable                              :155-198
 *   calculateTagLocationWithRangings                               :476-496
 *   timerRangingCallback (50 ms one-shot, MAX_TIME_TO_SEND_RANGING) :143-152, Posgenerator.h:77
 *   tag_reports_t                    src/kfpos/publishers/Posgenerator.h:78-105
 * i.e. ranges are floor()ed to integer millimetres (:213) and stored per 8-bit sequence number and
 * anchor column; a message with a NEW sequence number first flushes the previous sequence to the
 * estimator (:246-247); only column 0 of the new row is reset (:251-255 -- sic: columns 1..63 keep what
 * the same sequence number held 256 epochs ago unless a message overwrites them); errorEstimation is
 * stored on a same-sequence message only when > 0 (:241-243, :95); a flush does not clear anything, so
 * a timer flush followed by the next sequence number hands the same epoch to the estimator twice.
 *
 * What changes is the fan-out: the reference node drops every tag but one (:203) and its single timer
 * serves list index 0 (:148-150). Here every tag owns a row of the batch and its own 50 ms deadline, and
 * flushed epochs are handed to the GPU together: one kfpos_step_toa call per round, with the
 * estimator's wall-clock timeLag computed per tag and dt < 0 for the tags that have nothing this round.
 * The caller supplies time (seconds, monotonic) with every event, as the adaptor does.
 *
 * The other sensor callbacks of PosGenerator (Posgenerator.cpp:99-141) are accepted per tag too (onImu,
 * onPX4Flow, onCompass, onMag): they queue behind that tag's pending ranging epochs, so each filter sees its
 * calls in arrival order, and a round hands every kind to the GPU in one call (kfpos_step_imu for the 9-state
 * filter, kfpos_step_sensor for the planar one; the 6-state filter ignores them like the reference does).
 */
#ifndef KFPOS_INGEST_H
#define KFPOS_INGEST_H

#include <cmath>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "kfpos.h"

namespace kfpos_host {

class BatchedRangingNode {
public:
    static constexpr double kMaxTimeToSendRanging = 0.05; /* MAX_TIME_TO_SEND_RANGING */

    /* tagIds: the tag id served by batch row t; handle: n_tags == tagIds.size(), anchors already set in
     * the same order as anchorIds (column a = anchorIds[a], the reference's _anchorIndexById). */
    BatchedRangingNode(kfpos_handle *h, const std::vector<int> &tagIds, const std::vector<int> &anchorIds)
        : h_(h), T_((int)tagIds.size()), A_((int)anchorIds.size()) {
        for (int t = 0; t < T_; ++t) row_[tagIds[t]] = t;
        for (int a = 0; a < A_; ++a) col_[anchorIds[a]] = a;
        tags_.resize(T_);
        real_ = kfpos_real_size(h);
    }

    /* gtec_msgs::Ranging -> PosGenerator::newTOAMeasurement -> processRangingNow */
    void onRanging(double now, int anchorId, int tagId, double range, double errorEstimation, int seq) {
        auto it = row_.find(tagId);
        if (it == row_.end()) return; /* not one of ours (Posgenerator.cpp:203) */
        if (!col_.count(anchorId)) return;
        const int a = col_[anchorId];
        Tag &tg = tags_[it->second];
        if (tg.value.empty()) { /* initialiseTagList, Posgenerator.cpp:499-507 */
            tg.value.assign((size_t)256 * A_, -1);
            tg.err.assign((size_t)256 * A_, 0.0);
            tg.count.assign(256, 0);
            tg.seq = -1;
        }
        if (tg.armed && tg.deadline <= now) { /* this tag's 50 ms timer fired before the message arrived */
            tg.armed = false;
            enqueueFlush(it->second, tg.deadline);
        }
        const int32_t mm = (int32_t)std::floor(range); /* :213 */
        const int s = seq & 0xff;
        if (tg.seq == s) {
            tg.count[s]++;
            tg.value[(size_t)s * A_ + a] = mm;
            if (errorEstimation > 0.0) tg.err[(size_t)s * A_ + a] = errorEstimation;
        } else {
            enqueueFlush(it->second, now); /* :246-247 */
            tg.value[(size_t)s * A_ + 0] = -1; /* sic: column 0 only, :251-255 */
            tg.err[(size_t)s * A_ + 0] = 0.0;
            tg.count[s] = 1;
            tg.seq = s;
            tg.value[(size_t)s * A_ + a] = mm;
            tg.err[(size_t)s * A_ + a] = errorEstimation;
        }
        tg.deadline = now + kMaxTimeToSendRanging; /* timerRanging.stop(); start(); :274-277 */
        tg.armed = true;
    }

    /* sensor_msgs::Imu -> PosGenerator::newIMUMeasurement (Posgenerator.cpp:126-141) */
    void onImu(double now, int tagId, const double angVel[3], const double covAngVel[9], const double linAcc[3],
               const double covAcc[9]) {
        double d[24];
        for (int k = 0; k < 3; ++k) { d[k] = angVel[k]; d[12 + k] = linAcc[k]; }
        for (int k = 0; k < 9; ++k) { d[3 + k] = covAngVel[k]; d[15 + k] = covAcc[k]; }
        enqueueSensor(now, tagId, KFPOS_SENSOR_IMU, d, 24);
    }
    /* mavros_msgs::OpticalFlowRad -> newPX4FlowMeasurement, gated as Posgenerator.cpp:100 does */
    void onPX4Flow(double now, int tagId, double integratedX, double integratedY, double integratedZgyro,
                   double integrationTimeUs, int quality) {
        if (!(integrationTimeUs > 0 && quality > 0)) return;
        const double d[5] = {integratedX, integratedY, integratedZgyro, integrationTimeUs, (double)quality};
        enqueueSensor(now, tagId, KFPOS_SENSOR_PX4FLOW, d, 5);
    }
    /* std_msgs::Float64 -> newCompassMeasurement (:121-123) */
    void onCompass(double now, int tagId, double heading) { enqueueSensor(now, tagId, KFPOS_SENSOR_COMPASS, &heading, 1); }
    /* sensor_msgs::MagneticField -> newMAGMeasurement (:106-118) */
    void onMag(double now, int tagId, const double field[3]) { enqueueSensor(now, tagId, KFPOS_SENSOR_MAG, field, 3); }

    /* Advance time: fire the 50 ms deadlines that have passed (timerRangingCallback), then hand every
     * pending epoch to the GPU. Returns the number of estimator calls made (tag-epochs). */
    int poll(double now) {
        for (int t = 0; t < T_; ++t) {
            Tag &tg = tags_[t];
            if (tg.armed && tg.deadline <= now) {
                tg.armed = false; /* one-shot */
                enqueueFlush(t, tg.deadline);
            }
        }
        return drain();
    }

    /* the tags' estimator clocks, for getPose extrapolation: seconds since each tag's last estimate */
    double sinceLastEstimate(int row, double now) const { return tags_[row].started ? now - tags_[row].last : 0.0; }
    bool started(int row) const { return tags_[row].started; }
    int rows() const { return T_; }

private:
    struct Tag {
        std::vector<int32_t> value; /* rangeValue[256][A] */
        std::vector<double> err;    /* errorEstimation[256][A] */
        std::vector<int> count;     /* rangeCount[256] */
        int seq = -1;               /* rangeSeq */
        bool armed = false;
        double deadline = 0.0;
        bool started = false;       /* the estimator's mLastKFTimestamp != min */
        double last = 0.0;
    };
    struct Pending {
        int row;
        double time;
        int kind = 0; /* 0: ranging epoch, else KFPOS_SENSOR_* */
        std::vector<int32_t> mm;
        std::vector<double> err; /* ranging: errorEstimation row; sensor: the sample */
    };

    void enqueueSensor(double now, int tagId, int kind, const double *data, int n) {
        auto it = row_.find(tagId);
        if (it == row_.end()) return;
        /* the empty virtuals of PositionEstimationAlgorithm.h:26-35 do not touch the filter, nor its clock */
        const int dim = kfpos_state_dim(h_);
        if (!(dim == 8 || (dim == 9 && kind == KFPOS_SENSOR_IMU))) return;
        Tag &tg = tags_[it->second];
        if (tg.armed && tg.deadline <= now) { /* the ranging timer fired first */
            tg.armed = false;
            enqueueFlush(it->second, tg.deadline);
        }
        Pending p;
        p.row = it->second;
        p.time = now;
        p.kind = kind;
        p.err.assign(data, data + n);
        queue_.push_back(std::move(p));
    }

    /* sendRangingMeasurementIfAvailable (:155-198): a snapshot of the current sequence row */
    void enqueueFlush(int row, double now) {
        Tag &tg = tags_[row];
        if (tg.seq == -1 || tg.count[tg.seq] < 1) return;
        tg.armed = false; /* timerRanging.stop(), :172 */
        Pending p;
        p.row = row;
        p.time = now;
        p.mm.assign(tg.value.begin() + (size_t)tg.seq * A_, tg.value.begin() + (size_t)(tg.seq + 1) * A_);
        p.err.assign(tg.err.begin() + (size_t)tg.seq * A_, tg.err.begin() + (size_t)(tg.seq + 1) * A_);
        queue_.push_back(std::move(p));
    }

    static int sensorWidth(int kind) {
        return kind == KFPOS_SENSOR_PX4FLOW ? 5 : kind == KFPOS_SENSOR_IMU ? 24 : kind == KFPOS_SENSOR_MAG ? 3 : 1;
    }
    static void check(int rc, const char *what) {
        if (rc != KFPOS_OK)
            throw std::runtime_error(std::string(what) + ": " + kfpos_strerror(rc) + " " + kfpos_last_error());
    }

    /* A round takes at most one pending call per tag, in arrival order, and makes one batched call per kind
     * present (tags without a call of that kind get dt < 0). Tags are independent, so the order of the kinds
     * inside a round does not matter. */
    int drain() {
        int calls = 0;
        const int state_dim = kfpos_state_dim(h_);
        while (!queue_.empty()) {
            std::vector<char> taken(T_, 0);
            std::vector<Pending> round, rest;
            for (Pending &p : queue_) {
                if (taken[p.row]) { rest.push_back(std::move(p)); continue; }
                taken[p.row] = 1;
                round.push_back(std::move(p));
            }
            queue_.swap(rest);
            std::vector<double> lag(round.size());
            for (size_t i = 0; i < round.size(); ++i) { /* every call reads the filter's clock, KalmanFilterTOA.cpp:78-88 */
                Tag &tg = tags_[round[i].row];
                lag[i] = tg.started ? round[i].time - tg.last : 0.1;
                tg.last = round[i].time;
                tg.started = true;
            }
            for (int kind = 0; kind <= KFPOS_SENSOR_COMPASS; ++kind) {
                std::vector<double> dt(T_, -1.0);
                int n = 0;
                for (size_t i = 0; i < round.size(); ++i)
                    if (round[i].kind == kind) { dt[round[i].row] = lag[i]; ++n; }
                if (n == 0) continue;
                calls += n;
                if (kind == 0) {
                    std::vector<int32_t> mm((size_t)T_ * A_, 0);
                    std::vector<double> err64((size_t)T_ * A_, 1.0);
                    for (const Pending &p : round) {
                        if (p.kind != 0) continue;
                        for (int a = 0; a < A_; ++a) {
                            mm[(size_t)p.row * A_ + a] = p.mm[a] > 0 ? p.mm[a] : 0; /* only entries > 0 (:483) */
                            err64[(size_t)p.row * A_ + a] = p.err[a];
                        }
                    }
                    if (real_ == 4) {
                        std::vector<float> e32(err64.begin(), err64.end());
                        check(kfpos_step_toa(h_, mm.data(), e32.data(), dt.data(), T_, nullptr), "kfpos_step_toa");
                    } else {
                        check(kfpos_step_toa(h_, mm.data(), err64.data(), dt.data(), T_, nullptr), "kfpos_step_toa");
                    }
                } else if (state_dim == 8) { /* planar filter: the sample as the reference callback passes it */
                    const int C = sensorWidth(kind);
                    std::vector<double> data((size_t)T_ * C, 0.0);
                    for (const Pending &p : round)
                        if (p.kind == kind) std::copy(p.err.begin(), p.err.end(), data.begin() + (size_t)p.row * C);
                    check(kfpos_step_sensor(h_, kind, data.data(), dt.data(), T_, nullptr), "kfpos_step_sensor");
                } else if (state_dim == 9 && kind == KFPOS_SENSOR_IMU) { /* KalmanFilterTOAIMU: acceleration + its covariance */
                    std::vector<double> acc((size_t)T_ * 3, 0.0), cov((size_t)T_ * 9, 0.0);
                    for (int t = 0; t < T_; ++t) cov[(size_t)t * 9] = cov[(size_t)t * 9 + 4] = cov[(size_t)t * 9 + 8] = 1.0;
                    for (const Pending &p : round) {
                        if (p.kind != kind) continue;
                        std::copy(p.err.begin() + 12, p.err.begin() + 15, acc.begin() + (size_t)p.row * 3);
                        std::copy(p.err.begin() + 15, p.err.begin() + 24, cov.begin() + (size_t)p.row * 9);
                    }
                    if (real_ == 4) {
                        std::vector<float> a32(acc.begin(), acc.end()), c32(cov.begin(), cov.end());
                        check(kfpos_step_imu(h_, a32.data(), c32.data(), dt.data(), T_, nullptr), "kfpos_step_imu");
                    } else {
                        check(kfpos_step_imu(h_, acc.data(), cov.data(), dt.data(), T_, nullptr), "kfpos_step_imu");
                    }
                }
            }
        }
        return calls;
    }

    kfpos_handle *h_;
    int T_, A_, real_;
    std::map<int, int> row_, col_;
    std::vector<Tag> tags_;
    std::vector<Pending> queue_;
};

} // namespace kfpos_host
#endif

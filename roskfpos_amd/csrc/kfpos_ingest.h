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
 * flushed epochs are handed to the GPU together, with the estimator's wall-clock timeLag computed per tag
 * and dt < 0 for the tags that have nothing in a round. The caller supplies time (seconds, monotonic) with
 * every event, as the adaptor does.
 *
 * Built for message rates, not for one tag (BASELINE configs[2]: 65 536 tags x 8 anchors x 20 Hz = 10.5 M
 * messages/s in front of a 40 us kernel):
 *   - tag and anchor ids resolve through flat open-addressing tables, the per-(tag, sequence) rows live in one
 *     allocation, count + ranges + errorEstimations of a row side by side; onRangingBatch() prefetches the
 *     rows of the messages a few places ahead, so the table's DRAM latency overlaps the work on the
 *     current message;
 *   - a flushed epoch is written ONCE, straight into the pinned, component-major slot the GPU's DMA engine
 *     reads (kfpos_slot_*): no per-epoch vectors, no row-major intermediate, no layout turn on the device; only
 *     a tag's second call within one round goes through a (re-used) overflow arena;
 *   - rounds are submitted asynchronously: while the GPU works on round r the CPU assembles round r + 1 in the
 *     next slot.
 *
 * The other sensor callbacks of PosGenerator (Posgenerator.cpp:99-141) are accepted per tag too (onImu,
 * onPX4Flow, onCompass, onMag): they queue behind that tag's pending ranging epochs, so each filter sees its
 * calls in arrival order, and a round hands every kind to the GPU in one call (the IMU slot kind for the 9-state
 * filter, kfpos_step_sensor for the planar one; the 6-state filter ignores them like the reference does).
 */
#ifndef KFPOS_INGEST_H
#define KFPOS_INGEST_H

#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "kfpos.h"

namespace kfpos_host {

/* gtec_msgs::Ranging as PosGenerator::newTOAMeasurement receives it, plus its arrival time */
struct RangingMsg {
    double now;             /* seconds, monotonic */
    int anchorId, tagId;
    double range;           /* millimetres; floor()ed on entry (Posgenerator.cpp:213) */
    double errorEstimation; /* m^2 */
    int seq;
};

/* int -> int, open addressing, never erased from: one probe for almost every lookup */
class FlatIdMap {
public:
    void build(const std::vector<int> &ids) {
        size_t cap = 16;
        while (cap < 2 * ids.size() + 2) cap <<= 1;
        mask_ = cap - 1;
        key_.assign(cap, empty());
        val_.assign(cap, -1);
        for (size_t i = 0; i < ids.size(); ++i) {
            size_t p = hash(ids[i]);
            while (key_[p] != empty() && key_[p] != ids[i]) p = (p + 1) & mask_;
            key_[p] = ids[i];
            val_[p] = (int)i; /* a repeated id keeps its last position, like std::map::operator[] did */
        }
    }
    int find(int id) const {
        size_t p = hash(id);
        while (key_[p] != empty()) {
            if (key_[p] == id) return val_[p];
            p = (p + 1) & mask_;
        }
        return -1;
    }

private:
    static int64_t empty() { return INT64_MIN; }
    size_t hash(int id) const { return ((uint64_t)(uint32_t)id * 0x9E3779B97F4A7C15ull >> 32) & mask_; }
    std::vector<int64_t> key_;
    std::vector<int> val_;
    size_t mask_ = 0;
};

class BatchedRangingNode {
public:
    static constexpr double kMaxTimeToSendRanging = 0.05; /* MAX_TIME_TO_SEND_RANGING */

    /* tagIds: the tag id served by batch row t; handle: n_tags == tagIds.size(), anchors already set in
     * the same order as anchorIds (column a = anchorIds[a], the reference's _anchorIndexById). */
    BatchedRangingNode(kfpos_handle *h, const std::vector<int> &tagIds, const std::vector<int> &anchorIds)
        : h_(h), T_((int)tagIds.size()), A_((int)anchorIds.size()) {
        row_.build(tagIds);
        col_.build(anchorIds);
        tags_.resize(T_);
        real_ = kfpos_real_size(h);
        dim_ = kfpos_state_dim(h);
        nSlots_ = kfpos_slot_count(h) < kMaxSlots ? kfpos_slot_count(h) : kMaxSlots;
        /* one (tag, sequence) row: count, A ranges, A errorEstimations, padded to whole cache lines */
        rowBytes_ = ((sizeof(int32_t) * (1 + A_) + 7) & ~(size_t)7) + sizeof(double) * A_;
        rowBytes_ = (rowBytes_ + 63) & ~(size_t)63;
        errOff_ = (sizeof(int32_t) * (1 + A_) + 7) & ~(size_t)7;
        table_.reset(new unsigned char[(size_t)T_ * 256 * rowBytes_ + 64]); /* untouched until a tag speaks up */
        tableBase_ = (unsigned char *)(((uintptr_t)table_.get() + 63) & ~(uintptr_t)63);
        taken_.assign(T_, 0);
        for (int k = 0; k < 5; ++k) roundSlot_[k] = -1;
        if (dim_ == 8) { /* planar filter: its other four sensors go through the synchronous kfpos_step_sensor */
            for (int k = 1; k <= 4; ++k) sens_[k].assign((size_t)T_ * sensorWidth(k), 0.0);
            sensDt_.assign(T_, -1.0);
        }
    }

    /* gtec_msgs::Ranging -> PosGenerator::newTOAMeasurement -> processRangingNow */
    void onRanging(double now, int anchorId, int tagId, double range, double errorEstimation, int seq) {
        const RangingMsg m = {now, anchorId, tagId, range, errorEstimation, seq};
        onRangingBatch(&m, 1);
    }

    /* A burst of ranging messages in arrival order (a network packet, a ROS queue drain). Same effect as calling
     * onRanging for each; the table rows of the messages kPrefetch places ahead are requested from memory early. */
    void onRangingBatch(const RangingMsg *m, size_t n) {
        constexpr size_t kPrefetch = 12;
        for (size_t i = 0; i < n; ++i) {
            if (i + kPrefetch < n) {
                const int r = row_.find(m[i + kPrefetch].tagId);
                if (r >= 0) {
                    __builtin_prefetch(&tags_[r], 1);
                    unsigned char *p = seqRow(r, m[i + kPrefetch].seq & 0xff);
                    __builtin_prefetch(p, 1);
                    __builtin_prefetch(p + 64, 1);
                }
            }
            ranging(m[i]);
        }
        messages_ += n;
    }

    /* sensor_msgs::Imu -> PosGenerator::newIMUMeasurement (Posgenerator.cpp:126-141) */
    void onImu(double now, int tagId, const double angVel[3], const double covAngVel[9], const double linAcc[3],
               const double covAcc[9]) {
        double d[24];
        for (int k = 0; k < 3; ++k) { d[k] = angVel[k]; d[12 + k] = linAcc[k]; }
        for (int k = 0; k < 9; ++k) { d[3 + k] = covAngVel[k]; d[15 + k] = covAcc[k]; }
        sensor(now, tagId, KFPOS_SENSOR_IMU, d, 24);
    }
    /* mavros_msgs::OpticalFlowRad -> newPX4FlowMeasurement, gated as Posgenerator.cpp:100 does */
    void onPX4Flow(double now, int tagId, double integratedX, double integratedY, double integratedZgyro,
                   double integrationTimeUs, int quality) {
        if (!(integrationTimeUs > 0 && quality > 0)) return;
        const double d[5] = {integratedX, integratedY, integratedZgyro, integrationTimeUs, (double)quality};
        sensor(now, tagId, KFPOS_SENSOR_PX4FLOW, d, 5);
    }
    /* std_msgs::Float64 -> newCompassMeasurement (:121-123) */
    void onCompass(double now, int tagId, double heading) { sensor(now, tagId, KFPOS_SENSOR_COMPASS, &heading, 1); }
    /* sensor_msgs::MagneticField -> newMAGMeasurement (:106-118) */
    void onMag(double now, int tagId, const double field[3]) { sensor(now, tagId, KFPOS_SENSOR_MAG, field, 3); }

    /* Advance time: fire the 50 ms deadlines that have passed (timerRangingCallback), then hand every
     * pending epoch to the GPU. Returns the number of estimator calls made (tag-epochs). The GPU may still be
     * working on them when this returns; every synchronous kfpos_* call (kfpos_get_pose_each, ...) waits first. */
    int poll(double now) {
        for (int t = 0; t < T_; ++t) {
            Tag &tg = tags_[t];
            if (tg.armed && tg.deadline <= now) {
                tg.armed = false; /* one-shot */
                flush(t, tg.deadline);
            }
        }
        return drain();
    }

    /* the tags' estimator clocks, for getPose extrapolation: seconds since each tag's last estimate */
    double sinceLastEstimate(int row, double now) const { return tags_[row].started ? now - tags_[row].last : 0.0; }
    bool started(int row) const { return tags_[row].started; }
    int rows() const { return T_; }
    uint64_t messages() const { return messages_; }    /* ranging messages taken in so far */
    uint64_t overflowCalls() const { return overflowed_; } /* calls that had to wait for a later round */

private:
    struct Tag {
        double deadline = 0.0;
        double last = 0.0;    /* the estimator's mLastKFTimestamp */
        int seq = -1;         /* rangeSeq */
        bool armed = false;
        bool started = false; /* mLastKFTimestamp != min */
        bool live = false;    /* its rows of the table are initialised (initialiseTagList) */
    };
    struct Overflow { /* a call that found its tag already taken in the round being assembled */
        int row, kind;
        double lag;
        size_t off; /* ranging: A ints in ovInt_ and A doubles in ovDbl_; sensor: its sample in ovDbl_ */
    };

    static int sensorWidth(int kind) {
        return kind == KFPOS_SENSOR_PX4FLOW ? 5 : kind == KFPOS_SENSOR_IMU ? 24 : kind == KFPOS_SENSOR_MAG ? 3 : 1;
    }
    static void check(int rc, const char *what) {
        if (rc != KFPOS_OK)
            throw std::runtime_error(std::string(what) + ": " + kfpos_strerror(rc) + " " + kfpos_last_error());
    }
    unsigned char *seqRow(int row, int s) const { return tableBase_ + ((size_t)row * 256 + s) * rowBytes_; }
    static int32_t &count(unsigned char *p) { return *(int32_t *)p; }
    static int32_t *values(unsigned char *p) { return (int32_t *)p + 1; }
    double *errs(unsigned char *p) const { return (double *)(p + errOff_); }

    void ranging(const RangingMsg &m) {
        const int r = row_.find(m.tagId);
        if (r < 0) return; /* not one of ours (Posgenerator.cpp:203) */
        const int a = col_.find(m.anchorId);
        if (a < 0) return;
        Tag &tg = tags_[r];
        if (!tg.live) { /* initialiseTagList, Posgenerator.cpp:499-507 */
            for (int s = 0; s < 256; ++s) {
                unsigned char *p = seqRow(r, s);
                count(p) = 0;
                for (int k = 0; k < A_; ++k) { values(p)[k] = -1; errs(p)[k] = 0.0; }
            }
            tg.seq = -1;
            tg.live = true;
        }
        if (tg.armed && tg.deadline <= m.now) { /* this tag's 50 ms timer fired before the message arrived */
            tg.armed = false;
            flush(r, tg.deadline);
        }
        const int32_t mm = (int32_t)std::floor(m.range); /* :213 */
        const int s = m.seq & 0xff;
        unsigned char *p = seqRow(r, s);
        if (tg.seq == s) {
            count(p)++;
            values(p)[a] = mm;
            if (m.errorEstimation > 0.0) errs(p)[a] = m.errorEstimation;
        } else {
            flush(r, m.now); /* :246-247 */
            values(p)[0] = -1; /* sic: column 0 only, :251-255 */
            errs(p)[0] = 0.0;
            count(p) = 1;
            tg.seq = s;
            values(p)[a] = mm;
            errs(p)[a] = m.errorEstimation;
        }
        tg.deadline = m.now + kMaxTimeToSendRanging; /* timerRanging.stop(); start(); :274-277 */
        tg.armed = true;
    }

    /* every estimator call reads the filter's clock on entry (KalmanFilterTOA.cpp:78-88): calls of one tag reach
     * here in arrival order, so its timeLag can be settled now */
    double lagOf(Tag &tg, double time) {
        const double lag = tg.started ? time - tg.last : 0.1;
        tg.last = time;
        tg.started = true;
        return lag;
    }

    void sensor(double now, int tagId, int kind, const double *data, int n) {
        const int r = row_.find(tagId);
        if (r < 0) return;
        /* the empty virtuals of PositionEstimationAlgorithm.h:26-35 do not touch the filter, nor its clock */
        if (!(dim_ == 8 || (dim_ == 9 && kind == KFPOS_SENSOR_IMU))) return;
        Tag &tg = tags_[r];
        if (tg.armed && tg.deadline <= now) { /* the ranging timer fired first */
            tg.armed = false;
            flush(r, tg.deadline);
        }
        const double lag = lagOf(tg, now);
        if (!taken_[r]) {
            place(r, kind, lag, nullptr, nullptr, data);
        } else {
            Overflow o = {r, kind, lag, ovDbl_.size()};
            ovDbl_.insert(ovDbl_.end(), data, data + n);
            overflow_.push_back(o);
            ++overflowed_;
        }
    }

    /* sendRangingMeasurementIfAvailable (:155-198): the current sequence row goes to the estimator */
    void flush(int row, double time) {
        Tag &tg = tags_[row];
        if (tg.seq == -1) return;
        unsigned char *p = seqRow(row, tg.seq);
        if (count(p) < 1) return;
        tg.armed = false; /* timerRanging.stop(), :172 */
        const double lag = lagOf(tg, time);
        if (!taken_[row]) {
            place(row, 0, lag, values(p), errs(p), nullptr);
        } else { /* a second epoch of this tag before the round left: keep a snapshot for the next round */
            const Overflow o = {row, 0, lag, ovDbl_.size()};
            ovIntOff_.push_back(ovInt_.size());
            ovInt_.insert(ovInt_.end(), values(p), values(p) + A_);
            ovDbl_.insert(ovDbl_.end(), errs(p), errs(p) + A_);
            overflow_.push_back(o);
            ++overflowed_;
        }
    }

    /* the slot (pinned, component-major) that collects this round's calls of `kind` */
    kfpos_epoch_slot &slotFor(int kind) {
        if (roundSlot_[kind] < 0) {
            const int s = nextSlot_++ % nSlots_;
            check(kfpos_slot_acquire(h_, s, &slot_[s]), "kfpos_slot_acquire"); /* waits for its previous round */
            if (!slotInit_[s]) {
                for (int t = 0; t < T_; ++t) slot_[s].dt[t] = -1.0;
                slotInit_[s] = true;
            } else {
                for (int t : written_[s]) slot_[s].dt[t] = -1.0;
            }
            written_[s].clear();
            roundSlot_[kind] = s;
        }
        return slot_[roundSlot_[kind]];
    }

    /* one call of one tag into the round being assembled */
    void place(int row, int kind, double lag, const int32_t *mm, const double *err, const double *sample) {
        taken_[row] = 1;
        touched_.push_back(row);
        ++roundCalls_;
        const size_t T = (size_t)T_;
        if (kind == 0) {
            kfpos_epoch_slot &sl = slotFor(0);
            for (int a = 0; a < A_; ++a) sl.range_mm[(size_t)a * T + row] = mm[a] > 0 ? mm[a] : 0; /* only entries > 0 (:483) */
            if (real_ == 4) {
                float *e = (float *)sl.err_est;
                for (int a = 0; a < A_; ++a) e[(size_t)a * T + row] = (float)err[a];
            } else {
                double *e = (double *)sl.err_est;
                for (int a = 0; a < A_; ++a) e[(size_t)a * T + row] = err[a];
            }
            sl.dt[row] = lag;
            written_[roundSlot_[0]].push_back(row);
        } else if (dim_ == 9) { /* KalmanFilterTOAIMU::newIMUMeasurement: acceleration + its covariance */
            kfpos_epoch_slot &sl = slotFor(KFPOS_SENSOR_IMU);
            if (real_ == 4) {
                float *ac = (float *)sl.accel, *cv = (float *)sl.cov;
                for (int k = 0; k < 3; ++k) ac[(size_t)k * T + row] = (float)sample[12 + k];
                for (int k = 0; k < 9; ++k) cv[(size_t)k * T + row] = (float)sample[15 + k];
            } else {
                double *ac = (double *)sl.accel, *cv = (double *)sl.cov;
                for (int k = 0; k < 3; ++k) ac[(size_t)k * T + row] = sample[12 + k];
                for (int k = 0; k < 9; ++k) cv[(size_t)k * T + row] = sample[15 + k];
            }
            sl.dt[row] = lag;
            written_[roundSlot_[KFPOS_SENSOR_IMU]].push_back(row);
        } else { /* planar filter: the sample as the reference callback passes it */
            const int C = sensorWidth(kind);
            std::memcpy(&sens_[kind][(size_t)row * C], sample, sizeof(double) * C);
            if (sensRows_[kind].empty()) sensKinds_.push_back(kind);
            sensRows_[kind].push_back(row);
            sensLag_[kind].push_back(lag);
        }
    }

    /* hand the assembled round to the GPU: one submission per kind present */
    void submitRound() {
        if (roundSlot_[0] >= 0) {
            check(kfpos_slot_submit(h_, roundSlot_[0], KFPOS_SLOT_TOA | KFPOS_SLOT_DT_PER_TAG | KFPOS_SLOT_NO_POSE, 0.0),
                  "kfpos_slot_submit");
            roundSlot_[0] = -1;
        }
        if (roundSlot_[KFPOS_SENSOR_IMU] >= 0) {
            check(kfpos_slot_submit(h_, roundSlot_[KFPOS_SENSOR_IMU], KFPOS_SLOT_IMU | KFPOS_SLOT_DT_PER_TAG | KFPOS_SLOT_NO_POSE, 0.0),
                  "kfpos_slot_submit");
            roundSlot_[KFPOS_SENSOR_IMU] = -1;
        }
        for (int kind : sensKinds_) { /* planar sensors: synchronous (the call waits for the slots first) */
            for (size_t i = 0; i < sensRows_[kind].size(); ++i) sensDt_[sensRows_[kind][i]] = sensLag_[kind][i];
            check(kfpos_step_sensor(h_, kind, sens_[kind].data(), sensDt_.data(), T_, nullptr), "kfpos_step_sensor");
            for (int r : sensRows_[kind]) sensDt_[r] = -1.0;
            sensRows_[kind].clear();
            sensLag_[kind].clear();
        }
        sensKinds_.clear();
        for (int r : touched_) taken_[r] = 0;
        touched_.clear();
    }

    /* A round takes at most one pending call per tag, in arrival order, and makes one submission per kind
     * present (tags without a call of that kind carry dt < 0). Tags are independent, so the order of the kinds
     * inside a round does not matter. The first round has been assembled in place as the calls came in; what did
     * not fit (a tag's second call) follows in further rounds. */
    int drain() {
        int calls = roundCalls_;
        submitRound();
        while (!overflow_.empty()) {
            std::vector<Overflow> rest;
            std::vector<size_t> restInt;
            size_t iInt = 0;
            for (const Overflow &o : overflow_) {
                const bool ranging = o.kind == 0;
                const size_t intOff = ranging ? ovIntOff_[iInt++] : 0;
                if (taken_[o.row]) {
                    rest.push_back(o);
                    if (ranging) restInt.push_back(intOff);
                    continue;
                }
                if (ranging) place(o.row, 0, o.lag, &ovInt_[intOff], &ovDbl_[o.off], nullptr);
                else place(o.row, o.kind, o.lag, nullptr, nullptr, &ovDbl_[o.off]);
            }
            overflow_.swap(rest);
            ovIntOff_.swap(restInt);
            calls = roundCalls_;
            submitRound();
        }
        ovInt_.clear();
        ovDbl_.clear();
        ovIntOff_.clear();
        roundCalls_ = 0;
        return calls;
    }

    kfpos_handle *h_;
    int T_, A_, real_ = 8, dim_ = 6;
    FlatIdMap row_, col_;
    std::vector<Tag> tags_;
    std::unique_ptr<unsigned char[]> table_;
    unsigned char *tableBase_ = nullptr;
    size_t rowBytes_ = 0, errOff_ = 0;
    /* the round being assembled */
    std::vector<unsigned char> taken_;
    std::vector<int> touched_;
    int roundCalls_ = 0;
    static constexpr int kMaxSlots = 8;
    int nSlots_ = 2;
    kfpos_epoch_slot slot_[kMaxSlots];
    bool slotInit_[kMaxSlots] = {false};
    std::vector<int> written_[kMaxSlots];
    int roundSlot_[5];
    int nextSlot_ = 0;
    std::vector<double> sens_[5], sensLag_[5], sensDt_;
    std::vector<int> sensRows_[5], sensKinds_;
    /* calls waiting for a later round */
    std::vector<Overflow> overflow_;
    std::vector<int32_t> ovInt_;
    std::vector<double> ovDbl_;
    std::vector<size_t> ovIntOff_;
    uint64_t messages_ = 0, overflowed_ = 0;
};

} // namespace kfpos_host
#endif

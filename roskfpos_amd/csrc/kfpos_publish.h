/*
 * kfpos_publish.h -- pose -> published messages (the step right after the hot path).
 *
 * ROS is not available in this image, so the three messages and the TF the reference publishes are
 * mirrored as plain structs with the reference's exact field mapping:
 *   PosGenerator::publishFixedRateReport    src/kfpos/publishers/Posgenerator.cpp:541-548
 *   PosGenerator::publishPositionReport                                           :385-473
 *   topic names                             src/kfpos/publishers/node_pos.cpp:119-135
 * Quirks kept: nothing is published while report.x is NaN (:387); pose.covariance takes the first 36
 * LINEAR elements of covarianceMatrix in Armadillo's column-major order (:397-399) -- the whole 6x6 for
 * KalmanFilterTOA, but columns 0..3 of the 9x9 for KalmanFilterTOAIMU; the same 36 numbers go into
 * twist.covariance (:447); the path keeps the newest 1000 poses (Posgenerator.h:194, :422-426);
 * frame ids "world" (pose, path) and "odom" (odometry), TF odom -> pioneer3at::chassis (:459-469);
 * `nodeName` already starts with '/', so the topic names contain "//" (node_pos.cpp:120).
 */
#ifndef KFPOS_PUBLISH_H
#define KFPOS_PUBLISH_H

#include <cmath>
#include <deque>
#include <string>

#include "kfpos_adaptor.h"

namespace kfpos_host {

struct PoseMsg { /* geometry_msgs::Pose */
    double px, py, pz, qx, qy, qz, qw;
};
struct PoseWithCovarianceStamped {
    std::string frame_id;
    double stamp;
    PoseMsg pose;
    double covariance[36];
};
struct PoseStamped {
    std::string frame_id;
    double stamp;
    PoseMsg pose;
};
struct Path {
    std::string frame_id;
    double stamp;
    std::deque<PoseStamped> poses;
};
struct Odometry {
    std::string frame_id, child_frame_id;
    double stamp;
    PoseMsg pose;
    double pose_covariance[36];
    double twist_linear[3], twist_angular[3];
    double twist_covariance[36];
};
struct StampedTransform {
    std::string frame_id, child_frame_id;
    double stamp;
    PoseMsg transform;
};

struct Topics {
    std::string pose, path, odom;
};
/* node_pos.cpp:119-135 */
inline Topics topicNames(const std::string &nodeName, const std::string &targetDeviceId) {
    Topics t;
    t.pose = "/gtec/" + nodeName + "/" + targetDeviceId;
    t.path = "/gtec/" + nodeName + "/path/" + targetDeviceId;
    t.odom = "/gtec/" + nodeName + "/odom/" + targetDeviceId;
    return t;
}

class PosePublisher {
public:
    static constexpr size_t kMaxPathSize = 1000; /* Posgenerator.h:194 */

    /* publishPositionReport: returns false (nothing published) while the pose is NaN */
    bool publish(const Vector3 &report, double now) {
        if (std::isnan(report.x)) return false; /* :387 */
        const PoseMsg p = {report.x, report.y, report.z, report.rotX, report.rotY, report.rotZ, report.rotW};
        double lin[36];
        for (int i = 0; i < 36; ++i) lin[i] = report.covarianceMatrix(i); /* column-major linear index, :397-399 */
        msg.frame_id = "world";
        msg.stamp = now;
        msg.pose = p;
        for (int i = 0; i < 36; ++i) msg.covariance[i] = lin[i];

        PoseStamped ps;
        ps.frame_id = "world";
        ps.stamp = now;
        ps.pose = p;
        path.poses.push_back(ps);
        if (path.poses.size() > kMaxPathSize) path.poses.pop_front(); /* drop the oldest */
        path.frame_id = "world";
        path.stamp = now;

        odom.frame_id = "odom";
        odom.child_frame_id = "pioneer3at::chassis";
        odom.stamp = now;
        odom.pose = p;
        for (int i = 0; i < 36; ++i) odom.pose_covariance[i] = odom.twist_covariance[i] = lin[i];
        odom.twist_linear[0] = report.linearSpeedX; odom.twist_linear[1] = report.linearSpeedY;
        odom.twist_linear[2] = report.linearSpeedZ;
        odom.twist_angular[0] = report.angularSpeedX; odom.twist_angular[1] = report.angularSpeedY;
        odom.twist_angular[2] = report.angularSpeedZ;

        tf.frame_id = "odom";
        tf.child_frame_id = "pioneer3at::chassis";
        tf.stamp = now;
        tf.transform = p;
        return true;
    }

    /* publishFixedRateReport: pose = {NaN, NaN, NaN}; getPose; publish if it returned true */
    bool fixedRateReport(PositionEstimationAlgorithm &alg, double now) {
        Vector3 pose = {NAN, NAN, NAN}; /* Posgenerator.cpp:542 */
        if (!alg.getPose(pose)) return false;
        return publish(pose, now);
    }

    PoseWithCovarianceStamped msg;
    Path path;
    Odometry odom;
    StampedTransform tf;
};

} // namespace kfpos_host
#endif

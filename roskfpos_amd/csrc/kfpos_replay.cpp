/*
 * kfpos_replay.cpp -- ROS-free trace replay through the host adaptor (kfpos_adaptor.h).
 *
 * Stands where PosGenerator + node_pos.cpp stand in the reference: it takes the launch-file
 * parameters by their reference names (node_pos.cpp:48-109, roslaunch `name:=value` syntax), builds
 * the estimator the way PosGenerator::setAlgorithm does (Posgenerator.cpp:510-538), assembles ranging
 * messages into epochs by sequence number (Posgenerator.cpp:201-281, 476-496) and prints poses.
 * ROS itself is not available in this image, so messages come from a text trace:
 *
 *   A <anchorId> <x> <y> <z>                          anchors topic (Posgenerator.cpp:15-39)
 *   R <t> <anchorId> <tagId> <range_mm> <seq> <err>   gtec_msgs::Ranging (one per tag-anchor range)
 *   I <t> <ax> <ay> <az> <c0> ... <c8>                sensor_msgs::Imu linear acceleration + covariance
 *   J <t> <wx> <wy> <wz> <cw0..8> <ax> <ay> <az> <ca0..8>   the whole sensor_msgs::Imu (ALGORITHM_KF reads the yaw rate too)
 *   X <t> <int_x> <int_y> <int_zgyro> <int_time_us> <quality>   mavros_msgs::OpticalFlowRad (Posgenerator.cpp:99-103)
 *   C <t> <heading>                                   std_msgs::Float64 compass, what node_pos.cpp subscribes as "MAG" (:169-173)
 *   G <t> <mx> <my> <mz>                              sensor_msgs::MagneticField -> PosGenerator::newMAGMeasurement (:106-118)
 *   P <t>                                             fixed-rate publish tick -> getPose (Posgenerator.cpp:541-548)
 *   F <t>                                             flush the open epoch (the 50 ms timer, Posgenerator.cpp:143-152)
 *
 * Output: one line per P tick: "P <t> <ok> <x> <y> <z> <cov00> <cov11> <cov22>".
 */
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>

#include "kfpos_adaptor.h"
#include "kfpos_ingest.h"
#include "kfpos_publish.h"

using namespace kfpos_host;

struct NodeParams { /* names and defaults: node_pos.cpp:48-109, kfpos_toa.launch, kfpos_toa_imu.launch */
    std::string algorithm = "ALGORITHM_KF_TOA";
    std::string toaTagId = ""; /* the node reads toaTagId, not the launch files' tagId (node_pos.cpp:63) */
    double accelNoise = 0.5, jolt = 0.5;
    int useStartPosition = 0;
    double initPositionX = 0, initPositionY = 0, initPositionZ = 0;
    int useHeuristicIgnoreWorst = 0;
    double heuristicIgnoreThreshold = 0.5;
    /* ALGORITHM_KF (node_pos.cpp:64-97): initAngle is only read with useStartPosition = 1; the use* switches decide
     * which topics are subscribed; config*: files behind the five XML parameters (<param textfile=.../>) */
    int use2d = 0, variant = 0, numRangingsToIgnore = 0; /* ALGORITHM_ML, node_pos.cpp:79-81 */
    double initAngle = 0;
    int usePX4Flow = 0, useTOA = 1, useIMU = 0, useMAG = 0;
    std::string configPos, configPX4Flow, configUWB, configIMU, configMAG;
    std::string targetDeviceId = "", nodeName = "/kfpos"; /* topic names, node_pos.cpp:119-135 */
    int dumpMessages = 0; /* not a reference parameter: print every published message field on P ticks */
    std::string tagIds = ""; /* not a reference parameter: comma-separated hex tag ids -> batched multi-tag mode */
};

static bool set_param(NodeParams &p, const std::string &k, const std::string &v) {
    if (k == "algorithm") p.algorithm = v;
    else if (k == "toaTagId") p.toaTagId = v;
    else if (k == "accelNoise") p.accelNoise = atof(v.c_str());
    else if (k == "jolt") p.jolt = atof(v.c_str());
    else if (k == "useStartPosition") p.useStartPosition = atoi(v.c_str());
    else if (k == "initPositionX") p.initPositionX = atof(v.c_str());
    else if (k == "initPositionY") p.initPositionY = atof(v.c_str());
    else if (k == "initPositionZ") p.initPositionZ = atof(v.c_str());
    else if (k == "useHeuristicIgnoreWorst") p.useHeuristicIgnoreWorst = atoi(v.c_str());
    else if (k == "heuristicIgnoreThreshold") p.heuristicIgnoreThreshold = atof(v.c_str());
    else if (k == "use2d") p.use2d = atoi(v.c_str());
    else if (k == "variant") p.variant = atoi(v.c_str());
    else if (k == "numRangingsToIgnore") p.numRangingsToIgnore = atoi(v.c_str());
    else if (k == "initAngle") p.initAngle = atof(v.c_str());
    else if (k == "usePX4Flow") p.usePX4Flow = atoi(v.c_str());
    else if (k == "useTOA") p.useTOA = atoi(v.c_str());
    else if (k == "useIMU") p.useIMU = atoi(v.c_str());
    else if (k == "useMAG") p.useMAG = atoi(v.c_str());
    else if (k == "configPos") p.configPos = v;
    else if (k == "configPX4Flow") p.configPX4Flow = v;
    else if (k == "configUWB") p.configUWB = v;
    else if (k == "configIMU") p.configIMU = v;
    else if (k == "configMAG") p.configMAG = v;
    else if (k == "tagIds") p.tagIds = v;
    else if (k == "targetDeviceId") p.targetDeviceId = v;
    else if (k == "nodeName") p.nodeName = v;
    else if (k == "dumpMessages") p.dumpMessages = atoi(v.c_str());
    else return false;
    return true;
}

/* PosGenerator::setAlgorithm, Posgenerator.cpp:510-538 */
static std::unique_ptr<SingleTagFilter> make_algorithm(const NodeParams &p) {
    Vector3 init = Vector3();
    init.x = p.initPositionX; init.y = p.initPositionY; init.z = p.initPositionZ;
    if (p.algorithm == "ALGORITHM_KF_TOA") {
        /* sic: the reference's branch is inverted here (Posgenerator.cpp:512-516): WITHOUT
         * useStartPosition the fixed-initial-position constructor is chosen */
        if (!p.useStartPosition)
            return std::unique_ptr<SingleTagFilter>(new KalmanFilterTOA(
                p.accelNoise, p.useHeuristicIgnoreWorst != 0, p.heuristicIgnoreThreshold, init));
        return std::unique_ptr<SingleTagFilter>(
            new KalmanFilterTOA(p.accelNoise, p.useHeuristicIgnoreWorst != 0, p.heuristicIgnoreThreshold));
    }
    if (p.algorithm == "ALGORITHM_KF_TOA_IMU") {
        if (!p.useStartPosition) return std::unique_ptr<SingleTagFilter>(new KalmanFilterTOAIMU(p.accelNoise, p.jolt));
        return std::unique_ptr<SingleTagFilter>(new KalmanFilterTOAIMU(p.accelNoise, p.jolt, init));
    }
    if (p.algorithm == "ALGORITHM_KF") { /* Posgenerator.cpp:517-522 */
        const double angle = p.useStartPosition == 1 ? p.initAngle : 0.0; /* node_pos.cpp:68-73 */
        std::unique_ptr<KalmanFilter> kf(
            !p.useStartPosition ? new KalmanFilter(p.accelNoise, angle, p.jolt, "configPos", "configPX4Flow", "configUWB",
                                                   "configIMU", "configMAG")
                                : new KalmanFilter(p.accelNoise, angle, p.jolt, "configPos", "configPX4Flow", "configUWB",
                                                   "configIMU", "configMAG", init));
        kf->setParamSource(fileParamSource({{"configPos", p.configPos}, {"configPX4Flow", p.configPX4Flow},
                                            {"configUWB", p.configUWB}, {"configIMU", p.configIMU},
                                            {"configMAG", p.configMAG}}));
        return std::unique_ptr<SingleTagFilter>(kf.release());
    }
    if (p.algorithm == "ALGORITHM_ML") { /* Posgenerator.cpp:529-534 */
        Vector3 seed = Vector3();
        seed.x = 1; seed.y = 1; seed.z = 4;
        return std::unique_ptr<SingleTagFilter>(
            new MLLocation(p.use2d != 0, p.variant, p.numRangingsToIgnore, p.useStartPosition ? init : seed));
    }
    throw std::invalid_argument("algorithm must be ALGORITHM_KF_TOA, ALGORITHM_KF_TOA_IMU, ALGORITHM_KF or ALGORITHM_ML");
}

/* The ranging epoch table of PosGenerator for one tag: ranges keyed by anchor column, flushed when a
 * message with a new sequence number arrives (Posgenerator.cpp:229-272). */
struct EpochAssembler {
    std::map<int, int> column; /* anchor id -> column (_anchorIndexById) */
    std::vector<Beacon> beacons;
    int seq = -1;
    std::vector<int> mm;
    std::vector<double> err;

    void addAnchor(int id, double x, double y, double z) {
        if (column.count(id)) return;
        Beacon b = Beacon();
        b.id = id;
        b.index = (int)beacons.size();
        b.position.x = x; b.position.y = y; b.position.z = z;
        column[id] = b.index;
        beacons.push_back(b);
        mm.assign(beacons.size(), -1);
        err.assign(beacons.size(), 0.0);
    }
    /* calculateTagLocationWithRangings, Posgenerator.cpp:476-496 */
    void flush(PositionEstimationAlgorithm &alg) {
        if (seq < 0) return;
        std::vector<double> r, e;
        std::vector<Beacon> sel;
        for (size_t i = 0; i < mm.size(); ++i)
            if (mm[i] > 0) {
                r.push_back((double)mm[i] / 1000);
                e.push_back(err[i]);
                sel.push_back(beacons[i]);
            }
        alg.newTOAMeasurement(r, sel, e, 0.0);
        std::fill(mm.begin(), mm.end(), -1);
        seq = -1;
    }
    void ranging(PositionEstimationAlgorithm &alg, int anchorId, double range_mm, int s, double e) {
        if (!column.count(anchorId)) return; /* dropped until the anchor is known (Posgenerator.cpp:92-96) */
        if (seq >= 0 && s != seq) flush(alg);
        seq = s;
        mm[column[anchorId]] = (int)std::floor(range_mm); /* Posgenerator.cpp:213 */
        err[column[anchorId]] = e;
    }
};

/* Batched mode (tagIds:=...): every listed tag gets a row of ONE GPU handle; messages go through
 * kfpos_ingest.h's BatchedRangingNode, 'F'/'P' lines advance time (timers + GPU rounds), 'P' prints one
 * line per tag: "P <t> <tagId> <ok> <x> <y> <z> <cov00> <cov11> <cov22>". Sensor messages carry the tag they
 * belong to (a multi-tag node has one sensor topic per vehicle): lower-case kinds with the hex tag id after the
 * time stamp -- "j <t> <tag> ...", "x <t> <tag> ...", "c <t> <tag> <heading>", "g <t> <tag> <mx> <my> <mz>" --
 * and the same subscription switches as the single-tag node. */
static int run_batched(const NodeParams &p, const std::string &trace) {
    std::vector<int> tagIds;
    {
        std::stringstream ss(p.tagIds);
        std::string tok;
        while (std::getline(ss, tok, ',')) tagIds.push_back((int)strtol(tok.c_str(), nullptr, 16));
    }
    const int T = (int)tagIds.size();
    std::vector<int> anchorIds;
    std::vector<double> anchorXyz;
    kfpos_handle *h = nullptr;
    std::unique_ptr<BatchedRangingNode> node;
    auto ensure = [&]() {
        if (node) return;
        kfpos_config c;
        std::memset(&c, 0, sizeof(c));
        c.model = p.algorithm == "ALGORITHM_KF_TOA_IMU" ? KFPOS_MODEL_TOA_IMU
                : p.algorithm == "ALGORITHM_KF" ? KFPOS_MODEL_PLANAR : KFPOS_MODEL_TOA;
        c.n_tags = T;
        c.max_anchors = (int)anchorIds.size();
        c.storage = KFPOS_STORE_F64;
        c.accel_noise = p.accelNoise;
        c.jolt = p.jolt;
        c.ignore_worst = (c.model == KFPOS_MODEL_TOA) ? p.useHeuristicIgnoreWorst : 0;
        c.cost_threshold = p.heuristicIgnoreThreshold;
        /* same (inverted for KF_TOA) start-position rule as the single-tag factory above */
        const bool fixed = (c.model == KFPOS_MODEL_TOA) ? !p.useStartPosition : (p.useStartPosition != 0);
        c.use_init_pos = fixed ? 1 : 0;
        c.init_pos[0] = p.initPositionX; c.init_pos[1] = p.initPositionY; c.init_pos[2] = p.initPositionZ;
        if (kfpos_create(&c, &h) != KFPOS_OK) throw std::runtime_error(std::string("kfpos_create: ") + kfpos_last_error());
        if (c.model == KFPOS_MODEL_PLANAR) { /* KalmanFilter::init(): the five XML parameters, read once for the bank */
            KalmanFilter loader(p.accelNoise, p.useStartPosition == 1 ? p.initAngle : 0.0, p.jolt, "configPos",
                                "configPX4Flow", "configUWB", "configIMU", "configMAG");
            loader.setParamSource(fileParamSource({{"configPos", p.configPos}, {"configPX4Flow", p.configPX4Flow},
                                                   {"configUWB", p.configUWB}, {"configIMU", p.configIMU},
                                                   {"configMAG", p.configMAG}}));
            if (!loader.init()) throw std::runtime_error("init() failed: a config* XML parameter is missing or malformed");
            if (kfpos_set_planar(h, &loader.configuration()) != KFPOS_OK)
                throw std::runtime_error(std::string("kfpos_set_planar: ") + kfpos_last_error());
        }
        kfpos_set_anchors(h, anchorXyz.data(), anchorIds.data(), (int)anchorIds.size());
        node.reset(new BatchedRangingNode(h, tagIds, anchorIds));
    };
    const bool kf = p.algorithm == "ALGORITHM_KF"; /* which topics node_pos.cpp subscribes (:146-173) */
    const bool subTOA = kf ? p.useTOA == 1 : true, subIMU = kf ? p.useIMU == 1 : p.algorithm == "ALGORITHM_KF_TOA_IMU";
    const bool subPX4 = kf && p.usePX4Flow == 1, subMAG = kf && p.useMAG == 1;
    std::ifstream in(trace);
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        char kind;
        ss >> kind;
        if (kind == 'A' && !node) {
            int id; double x, y, z;
            ss >> id >> x >> y >> z;
            anchorIds.push_back(id);
            anchorXyz.push_back(x); anchorXyz.push_back(y); anchorXyz.push_back(z);
        } else if (kind == 'R' && subTOA) {
            int anchorId, tag, seq; double t, mm, e;
            ss >> t >> anchorId >> tag >> mm >> seq >> e;
            ensure();
            node->onRanging(t, anchorId, tag, mm, e, seq);
        } else if (kind == 'j' && subIMU) {
            double t, w[3], cw[9], a[3], ca[9]; std::string tag;
            ss >> t >> tag >> w[0] >> w[1] >> w[2];
            for (double &v : cw) ss >> v;
            ss >> a[0] >> a[1] >> a[2];
            for (double &v : ca) ss >> v;
            ensure();
            node->onImu(t, (int)strtol(tag.c_str(), nullptr, 16), w, cw, a, ca);
        } else if (kind == 'x' && subPX4) {
            double t, ix, iy, iz, us; int q; std::string tag;
            ss >> t >> tag >> ix >> iy >> iz >> us >> q;
            ensure();
            node->onPX4Flow(t, (int)strtol(tag.c_str(), nullptr, 16), ix, iy, iz, us, q);
        } else if (kind == 'c' && subMAG) {
            double t, heading; std::string tag;
            ss >> t >> tag >> heading;
            ensure();
            node->onCompass(t, (int)strtol(tag.c_str(), nullptr, 16), heading);
        } else if (kind == 'g' && subMAG) {
            double t, m[3]; std::string tag;
            ss >> t >> tag >> m[0] >> m[1] >> m[2];
            ensure();
            node->onMag(t, (int)strtol(tag.c_str(), nullptr, 16), m);
        } else if (kind == 'F') {
            double t; ss >> t;
            ensure();
            node->poll(t);
        } else if (kind == 'P') {
            double t; ss >> t;
            ensure();
            node->poll(t);
            std::vector<double> ahead(T), pos(3 * T), cov(9 * T), vel(3 * T);
            std::vector<uint32_t> st(T);
            for (int r = 0; r < T; ++r) ahead[r] = node->sinceLastEstimate(r, t);
            if (kfpos_get_pose_each(h, ahead.data(), pos.data(), cov.data(), vel.data(), st.data()) != KFPOS_OK)
                throw std::runtime_error(std::string("kfpos_get_pose_each: ") + kfpos_last_error());
            for (int r = 0; r < T; ++r)
                printf("P %.9f %x %d %.17g %.17g %.17g %.17g %.17g %.17g\n", t, tagIds[r],
                       (st[r] & KFPOS_ST_NOT_STARTED) ? 0 : 1, pos[3 * r], pos[3 * r + 1], pos[3 * r + 2],
                       cov[9 * r], cov[9 * r + 4], cov[9 * r + 8]);
        }
    }
    node.reset();
    if (h) kfpos_destroy(h);
    return 0;
}

int main(int argc, char **argv) {
    NodeParams p;
    std::string trace;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        const size_t k = a.find(":=");
        if (k == std::string::npos) { trace = a; continue; }
        if (!set_param(p, a.substr(0, k), a.substr(k + 2))) {
            fprintf(stderr, "unknown parameter %s\n", a.substr(0, k).c_str());
            return 2;
        }
    }
    if (trace.empty()) {
        fprintf(stderr, "usage: kfpos_replay [name:=value ...] trace.txt\n");
        return 2;
    }
    if (!p.tagIds.empty()) {
        try {
            return run_batched(p, trace);
        } catch (const std::exception &e) {
            fprintf(stderr, "kfpos_replay: %s\n", e.what());
            return 1;
        }
    }
    int tagId = 0; /* hex parse, default 0 (node_pos.cpp:139-144) */
    if (!p.toaTagId.empty()) tagId = (int)strtol(p.toaTagId.c_str(), nullptr, 16);

    try {
        std::unique_ptr<SingleTagFilter> alg = make_algorithm(p);
        double now = 0.0;
        alg->setClock([&now] { return now; });
        if (!alg->init()) throw std::runtime_error("init() failed: a config* XML parameter is missing or malformed");
        /* which topics node_pos.cpp subscribes (:146-173) */
        const bool kf = p.algorithm == "ALGORITHM_KF";
        const bool subTOA = kf ? p.useTOA == 1 : true, subIMU = kf ? p.useIMU == 1 : p.algorithm == "ALGORITHM_KF_TOA_IMU";
        const bool subPX4 = kf && p.usePX4Flow == 1, subMAG = kf && p.useMAG == 1;
        EpochAssembler ep;
        PosePublisher publisher;
        std::ifstream in(trace);
        std::string line;
        while (std::getline(in, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream ss(line);
            char kind;
            ss >> kind;
            if (kind == 'A') {
                int id; double x, y, z;
                ss >> id >> x >> y >> z;
                ep.addAnchor(id, x, y, z);
            } else if (kind == 'R') {
                int anchorId, tag, seq; double t, mm, e;
                ss >> t >> anchorId >> tag >> mm >> seq >> e;
                if (tag != tagId || !subTOA) continue; /* Posgenerator.cpp:203 */
                now = t;
                ep.ranging(*alg, anchorId, mm, seq, e);
            } else if (kind == 'F') {
                ss >> now;
                ep.flush(*alg);
            } else if (kind == 'I' && subIMU) {
                double t, c[9]; VectorDim3 a, w = {0, 0, 0};
                ss >> t >> a.x >> a.y >> a.z;
                for (double &v : c) ss >> v;
                now = t;
                double cw[9] = {0};
                alg->newIMUMeasurement(w, cw, a, c);
            } else if (kind == 'J' && subIMU) {
                double t, cw[9], ca[9]; VectorDim3 a, w;
                ss >> t >> w.x >> w.y >> w.z;
                for (double &v : cw) ss >> v;
                ss >> a.x >> a.y >> a.z;
                for (double &v : ca) ss >> v;
                now = t;
                alg->newIMUMeasurement(w, cw, a, ca);
            } else if (kind == 'X' && subPX4) {
                double t, ix, iy, iz, us; int q;
                ss >> t >> ix >> iy >> iz >> us >> q;
                now = t;
                if (us > 0 && q > 0) alg->newPX4FlowMeasurement(ix, iy, iz, us, q); /* Posgenerator.cpp:100 */
            } else if (kind == 'C' && subMAG) {
                double t, heading;
                ss >> t >> heading;
                now = t;
                alg->newCompassMeasurement(heading);
            } else if (kind == 'G' && subMAG) {
                double t, c[9] = {0}; VectorDim3 m;
                ss >> t >> m.x >> m.y >> m.z;
                now = t;
                alg->newMAGMeasurement(m, c);
            } else if (kind == 'P' && p.dumpMessages) {
                ss >> now;
                const bool ok = publisher.fixedRateReport(*alg, now);
                const Topics tn = topicNames(p.nodeName, p.targetDeviceId);
                printf("M %.9f %d %s %s %s %s %s %s %zu", now, ok ? 1 : 0, tn.pose.c_str(), tn.path.c_str(),
                       tn.odom.c_str(), publisher.msg.frame_id.c_str(), publisher.odom.frame_id.c_str(),
                       publisher.odom.child_frame_id.c_str(), publisher.path.poses.size());
                if (ok) {
                    printf(" %.17g %.17g %.17g %.17g", publisher.msg.pose.px, publisher.msg.pose.py,
                           publisher.msg.pose.pz, publisher.msg.pose.qw);
                    for (int i = 0; i < 36; ++i) printf(" %.17g", publisher.msg.covariance[i]);
                    for (int i = 0; i < 3; ++i) printf(" %.17g", publisher.odom.twist_linear[i]);
                    for (int i = 0; i < 3; ++i) printf(" %.17g", publisher.odom.twist_angular[i]);
                }
                printf("\n");
            } else if (kind == 'P') {
                ss >> now;
                Vector3 pose = {NAN, NAN, NAN};
                const bool ok = alg->getPose(pose);
                const bool filled = pose.covarianceMatrix.n_rows >= 3;
                printf("P %.9f %d %.17g %.17g %.17g %.17g %.17g %.17g\n", now, ok ? 1 : 0, pose.x, pose.y, pose.z,
                       filled ? pose.covarianceMatrix(0, 0) : 0.0, filled ? pose.covarianceMatrix(1, 1) : 0.0,
                       filled ? pose.covarianceMatrix(2, 2) : 0.0);
            }
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "kfpos_replay: %s\n", e.what());
        return 1;
    }
    return 0;
}

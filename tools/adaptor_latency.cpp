/*
 * adaptor_latency.cpp -- latency of the single-tag estimator objects a drop-in node runs (kfpos_adaptor.h):
 * newTOAMeasurement (one ranging epoch of one tag: predict + iterated update on the GPU, synchronous) and getPose
 * (predict-only extrapolation), as PosGenerator calls them (Posgenerator.cpp:491, :543), plus newIMUMeasurement for the
 * 9-state filter. Prints one JSON line per estimator: median / mean / p99 microseconds per call.
 *   g++ -O2 -std=c++17 -I include -I roskfpos_amd/csrc -o gpurun_out/adaptor_latency tools/adaptor_latency.cpp \
 *       -L roskfpos_amd/csrc -lkfpos_hip -Wl,-rpath,$PWD/roskfpos_amd/csrc
 */
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "kfpos_adaptor.h"

using namespace kfpos_host;

static double now_us() { return steady_seconds() * 1e6; }

struct Stats { double med, mean, p99; };
static Stats stats(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    double s = 0;
    for (double x : v) s += x;
    return {v[v.size() / 2], s / v.size(), v[(size_t)(v.size() * 0.99)]};
}

template <class F>
static void run(const char *name, F &f, bool imu, int n_anchors, int iters) {
    std::vector<Beacon> beacons;
    for (int i = 0; i < n_anchors; ++i) {
        Beacon b = Beacon();
        b.id = 100 + i; b.index = i;
        b.position.x = 10.0 * (i & 1); b.position.y = 10.0 * ((i >> 1) & 1); b.position.z = 0.3 + 2.7 * ((i >> 2) & 1);
        if (i >= 8) { b.position.x = 5 + 3 * std::cos((double)i); b.position.y = 5 + 3 * std::sin((double)i); b.position.z = 1.5 + 0.1 * i; }
        beacons.push_back(b);
    }
    std::mt19937 rng(7);
    std::normal_distribution<double> noise(0.0, 0.05);
    std::vector<double> t_toa, t_imu, t_pose;
    double px = 4.0, py = 6.0, pz = 1.1;
    for (int it = 0; it < iters + 20; ++it) {
        px += 0.01; py -= 0.005;
        std::vector<double> ranges, errs;
        for (const Beacon &b : beacons) {
            const double d = std::sqrt((px - b.position.x) * (px - b.position.x) + (py - b.position.y) * (py - b.position.y) +
                                       (pz - b.position.z) * (pz - b.position.z));
            ranges.push_back(std::floor((d + noise(rng)) * 1000.0) / 1000.0); /* Posgenerator.cpp:213, :484 */
            errs.push_back(0.0025);
        }
        double t0 = now_us();
        if (imu) {
            VectorDim3 w = {0, 0, 0}, a = {0.01, -0.02, 0.0};
            double cw[9] = {1e-4, 0, 0, 0, 1e-4, 0, 0, 0, 1e-4}, ca[9] = {0.01, 0, 0, 0, 0.01, 0, 0, 0, 0.01};
            f.newIMUMeasurement(w, cw, a, ca);
            const double t1 = now_us();
            if (it >= 20) t_imu.push_back(t1 - t0);
            t0 = t1;
        }
        f.newTOAMeasurement(ranges, beacons, errs, 0.0);
        const double t1 = now_us();
        Vector3 pose = {NAN, NAN, NAN};
        const bool ok = f.getPose(pose);
        const double t2 = now_us();
        if (!ok || !std::isfinite(pose.x)) { std::printf("{\"estimator\": \"%s\", \"error\": \"no pose\"}\n", name); return; }
        if (it >= 20) { t_toa.push_back(t1 - t0); t_pose.push_back(t2 - t1); }
    }
    const Stats a = stats(t_toa), p = stats(t_pose);
    std::printf("{\"estimator\": \"%s\", \"anchors\": %d, \"calls\": %d, \"newTOAMeasurement_us\": {\"median\": %.1f, \"mean\": %.1f, \"p99\": %.1f}, "
                "\"getPose_us\": {\"median\": %.1f, \"mean\": %.1f, \"p99\": %.1f}",
                name, n_anchors, iters, a.med, a.mean, a.p99, p.med, p.mean, p.p99);
    if (imu) {
        const Stats i = stats(t_imu);
        std::printf(", \"newIMUMeasurement_us\": {\"median\": %.1f, \"mean\": %.1f, \"p99\": %.1f}", i.med, i.mean, i.p99);
    }
    std::printf("}\n");
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 500;
    Vector3 init = Vector3();
    init.x = 4.0; init.y = 6.0; init.z = 1.1;
    for (int n : {4, 8, 16}) {
        KalmanFilterTOA toa(0.5, false, 0.5, init);
        toa.init();
        run("KalmanFilterTOA", toa, false, n, iters);
    }
    {
        KalmanFilterTOA iw(0.5, true, 0.5, init);
        iw.init();
        run("KalmanFilterTOA ignoreWorst", iw, false, 8, iters);
    }
    {
        KalmanFilterTOAIMU imu(0.5, 0.5, init);
        imu.init();
        run("KalmanFilterTOAIMU", imu, true, 8, iters);
    }
    {
        kfpos_host::MLLocation ml(false, ML_VARIANT_NORMAL, 0, init);
        ml.init();
        run("MLLocation", ml, false, 8, iters);
    }
    return 0;
}

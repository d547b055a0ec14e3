// shard_node.cpp -- a multi-tag node in C++ that shards its tag batch over the GPUs of one machine, no Python anywhere:
// one kfpos handle per device, contiguous tag ranges (kfpos_shard_range), ONE collective per epoch -- the RCCL pose
// all-gather behind the C ABI (kfpos_allgather_poses_multi: the single-process form, kfpos_comm_create_all). What
// INTEGRATION.md section 3 describes; tests/test_comm_gpu.py builds and runs it on the one device a GPU box has.
//
//   hipcc -O2 -std=c++17 -I include -o shard_node tools/shard_node.cpp -L roskfpos_amd/csrc -lkfpos_hip -Wl,-rpath,$PWD/roskfpos_amd/csrc
//   ./shard_node N_DEVICES TOTAL_TAGS EPOCHS [DEVICES]   -> one JSON line; exit status 0 iff every check passed
//   DEVICES: comma-separated HIP device per shard (default 0,1,..,N-1). RCCL wants them distinct; the tests' stand-in
//   transport (tests/fake_rccl) accepts "0,0,0,0" and so runs the 4-shard node on the one card of a GPU box.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kfpos.h"

#define CHECK(call)                                                                                         \
    do {                                                                                                    \
        int rc_ = (call);                                                                                   \
        if (rc_ != 0) {                                                                                     \
            std::fprintf(stderr, "%s -> %s: %s\n", #call, kfpos_strerror(rc_), kfpos_last_error());         \
            return 1;                                                                                       \
        }                                                                                                   \
    } while (0)
#define HIP(call)                                                                            \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));                \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

// a room with 8 anchors in its corners and tags on circles inside it; ranges in the node's integer millimetres
static const double ANCHORS[8][3] = {{0, 0, 0.3}, {10, 0, 0.3}, {0, 10, 0.3}, {10, 10, 0.3},
                                     {0, 0, 3.0}, {10, 0, 3.0}, {0, 10, 3.0}, {10, 10, 3.0}};
static void tag_position(int64_t tag, double t, double p[3]) {
    const double rho = 1.0 + 3.0 * ((tag * 2654435761u) % 1000) / 1000.0, om = 0.1 + 0.3 * ((tag * 40503u) % 997) / 997.0;
    const double ph = 6.283185307179586 * ((tag * 69069u) % 991) / 991.0;
    p[0] = 5.0 + rho * std::cos(om * t + ph);
    p[1] = 5.0 + rho * std::sin(om * t + ph);
    p[2] = 1.0 + 0.2 * std::sin(0.1 * t + ph);
}
static uint32_t lcg(uint32_t &s) { return s = s * 1664525u + 1013904223u; }

int main(int argc, char **argv) {
    const int n_dev = argc > 1 ? std::atoi(argv[1]) : 1;
    const int64_t total = argc > 2 ? std::atoll(argv[2]) : 100000;
    const int epochs = argc > 3 ? std::atoi(argv[3]) : 20;
    const int A = 8;
    if (n_dev < 1 || total < n_dev) return 2;
    std::vector<int32_t> dev(n_dev);
    for (int d = 0; d < n_dev; ++d) dev[d] = d;
    if (argc > 4) {
        const char *p = argv[4];
        for (int d = 0; d < n_dev; ++d) {
            dev[d] = (int32_t)std::strtol(p, (char **)&p, 10);
            if (*p == ',') ++p;
        }
    }

    std::vector<kfpos_comm *> comms(n_dev, nullptr);
    CHECK(kfpos_comm_create_all(n_dev, dev.data(), comms.data())); // one RCCL clique over these devices
    std::vector<kfpos_handle *> h(n_dev, nullptr);
    std::vector<int64_t> lo(n_dev), hi(n_dev);
    std::vector<double *> pos_all(n_dev, nullptr); // [3][total] on every device: the poses of ALL tags
    for (int d = 0; d < n_dev; ++d) {
        CHECK(kfpos_comm_set_total(comms[d], total, &lo[d], &hi[d]));
        kfpos_config c = {};
        c.model = KFPOS_MODEL_TOA; // ALGORITHM_KF_TOA, 6 states
        c.n_tags = (int32_t)(hi[d] - lo[d]);
        c.max_anchors = A;
        c.storage = KFPOS_STORE_F64;
        c.accel_noise = 0.5;
        c.use_init_pos = 1;
        c.device = dev[d];
        CHECK(kfpos_create(&c, &h[d]));
        CHECK(kfpos_set_anchors(h[d], &ANCHORS[0][0], nullptr, A));
        std::vector<double> init((size_t)c.n_tags * 3);
        for (int64_t t = lo[d]; t < hi[d]; ++t) tag_position(t, 0.0, &init[(size_t)(t - lo[d]) * 3]);
        CHECK(kfpos_set_init_positions(h[d], init.data()));
        HIP(hipSetDevice(dev[d]));
        HIP(hipMalloc((void **)&pos_all[d], sizeof(double) * 3 * total));
    }

    std::vector<std::vector<int32_t>> mm(n_dev);
    std::vector<std::vector<double>> err(n_dev);
    for (int d = 0; d < n_dev; ++d) {
        mm[d].resize((size_t)(hi[d] - lo[d]) * A);
        err[d].assign((size_t)(hi[d] - lo[d]) * A, 0.0025);
    }
    double gather_ms = 0.0;
    for (int s = 0; s < epochs; ++s) {
        const double t = 0.1 + 0.05 * s, dt = s ? 0.05 : 0.1;
        for (int d = 0; d < n_dev; ++d) { // every device steps its own shard ...
            for (int64_t g = lo[d]; g < hi[d]; ++g) {
                double p[3];
                tag_position(g, t, p);
                uint32_t seed = (uint32_t)(g * 7919 + s * 104729);
                for (int a = 0; a < A; ++a) {
                    const double dx = p[0] - ANCHORS[a][0], dy = p[1] - ANCHORS[a][1], dz = p[2] - ANCHORS[a][2];
                    const double noise = ((int)(lcg(seed) >> 16) % 101 - 50) * 1e-3; // +-5 cm
                    mm[d][(size_t)(g - lo[d]) * A + a] = (int32_t)std::floor((std::sqrt(dx * dx + dy * dy + dz * dz) + noise) * 1000.0);
                }
            }
            CHECK(kfpos_step_toa(h[d], mm[d].data(), err[d].data(), &dt, 1, nullptr));
        }
        // ... and ONE collective brings the poses of all tags to all devices (pos_local = NULL: the handles' positions)
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(kfpos_allgather_poses_multi(n_dev, h.data(), comms.data(), nullptr, 3, pos_all.data(), nullptr));
        for (int d = 0; d < n_dev; ++d) CHECK(kfpos_comm_sync(comms[d]));
        gather_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }

    // check: what every device received == what each shard's own getPose (timeLag 0) says, bit for bit
    std::vector<double> want((size_t)3 * total), got((size_t)3 * total);
    for (int d = 0; d < n_dev; ++d) {
        const int64_t n = hi[d] - lo[d];
        std::vector<double> pos((size_t)n * 3);
        CHECK(kfpos_get_pose(h[d], 0.0, pos.data(), nullptr, nullptr, nullptr));
        for (int64_t t = 0; t < n; ++t)
            for (int k = 0; k < 3; ++k) want[(size_t)k * total + lo[d] + t] = pos[(size_t)t * 3 + k];
    }
    long mismatches = 0;
    double worst_track = 0.0;
    for (int d = 0; d < n_dev; ++d) {
        HIP(hipSetDevice(dev[d]));
        HIP(hipMemcpy(got.data(), pos_all[d], sizeof(double) * 3 * total, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < got.size(); ++i) mismatches += got[i] != want[i];
    }
    for (int64_t g = 0; g < total; g += 97) { // the filters do track: a sanity check of the example itself
        double p[3];
        tag_position(g, 0.1 + 0.05 * (epochs - 1), p);
        for (int k = 0; k < 3; ++k) worst_track = std::fmax(worst_track, std::fabs(want[(size_t)k * total + g] - p[k]));
    }
    std::printf("{\"devices\": %d, \"total_tags\": %lld, \"epochs\": %d, \"gather_ms_per_epoch\": %.4f, "
                "\"mismatches\": %ld, \"worst_distance_to_truth_m\": %.3f, \"rccl_version\": %d}\n",
                n_dev, (long long)total, epochs, gather_ms / epochs, mismatches, worst_track, kfpos_comm_backend_version());
    for (int d = 0; d < n_dev; ++d) {
        (void)hipFree(pos_all[d]);
        kfpos_destroy(h[d]);
        kfpos_comm_destroy(comms[d]);
    }
    return (mismatches == 0 && worst_track < 0.5) ? 0 : 1;
}

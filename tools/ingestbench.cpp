/*
 * ingestbench.cpp -- message rate of the ranging ingest (kfpos_ingest.h: BatchedRangingNode) in front of the GPU core,
 * BASELINE configs[2] shape by default: 65 536 tags x 8 anchors, one ranging message per (tag, anchor, epoch), the
 * messages of an epoch interleaved across tags (anchor-major: consecutive messages belong to different tags, the worst
 * case for the per-tag table). Times, per epoch of 524 288 messages: onRangingBatch (the table logic + the flushes it
 * triggers, written straight into the pinned slot) and poll (timers + submitting the round); the GPU works on round r
 * while the CPU ingests epoch r + 1. Prints one JSON line.
 *   g++ -O2 -std=c++17 -I include -I roskfpos_amd/csrc -o ingestbench tools/ingestbench.cpp -L roskfpos_amd/csrc -lkfpos_hip
 *   ./ingestbench [tags] [epochs] [model 0|1] [batch]
 */
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kfpos_ingest.h"

using namespace kfpos_host;

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
    const int T = argc > 1 ? std::atoi(argv[1]) : 65536, K = argc > 2 ? std::atoi(argv[2]) : 12;
    const int model = argc > 3 ? std::atoi(argv[3]) : 0;
    const size_t batch = argc > 4 ? (size_t)std::atol(argv[4]) : 4096;
    const int A = 8;
    kfpos_config c;
    std::memset(&c, 0, sizeof(c));
    c.model = model ? KFPOS_MODEL_TOA_IMU : KFPOS_MODEL_TOA;
    c.n_tags = T; c.max_anchors = A; c.storage = model ? KFPOS_STORE_MIXED : KFPOS_STORE_F64;
    c.accel_noise = c.jolt = 0.5; c.use_init_pos = 1;
    c.init_pos[0] = 5; c.init_pos[1] = 5; c.init_pos[2] = 1;
    kfpos_handle *h = nullptr;
    if (kfpos_create(&c, &h) != KFPOS_OK) { std::fprintf(stderr, "kfpos_create: %s\n", kfpos_last_error()); return 1; }
    double xyz[3 * A];
    std::vector<int> anchorIds, tagIds;
    for (int i = 0; i < A; ++i) {
        xyz[3 * i] = 10.0 * (i & 1); xyz[3 * i + 1] = 10.0 * ((i >> 1) & 1); xyz[3 * i + 2] = 0.3 + 2.7 * ((i >> 2) & 1);
        anchorIds.push_back(100 + i);
    }
    kfpos_set_anchors(h, xyz, anchorIds.data(), A);
    for (int t = 0; t < T; ++t) tagIds.push_back(0x1000 + 3 * t); /* sparse ids */
    BatchedRangingNode node(h, tagIds, anchorIds);
    /* one epoch of messages, re-stamped per epoch: tag k sits at (5 + cos k, 5 + sin k, 1) */
    std::vector<RangingMsg> msgs((size_t)T * A);
    for (int a = 0; a < A; ++a)
        for (int t = 0; t < T; ++t) {
            const double px = 5 + std::cos(0.001 * t), py = 5 + std::sin(0.001 * t), pz = 1.0;
            const double d = std::sqrt((px - xyz[3 * a]) * (px - xyz[3 * a]) + (py - xyz[3 * a + 1]) * (py - xyz[3 * a + 1]) +
                                       (pz - xyz[3 * a + 2]) * (pz - xyz[3 * a + 2]));
            msgs[(size_t)a * T + t] = RangingMsg{0.0, 100 + a, 0x1000 + 3 * t, d * 1000.0 + (t % 7), 0.0025, 0};
        }
    double t_ingest = 0, t_poll = 0;
    int calls = 0;
    const int warm = 3;
    double wall0 = 0;
    for (int k = 0; k < K + warm; ++k) {
        if (k == warm) { t_ingest = t_poll = 0; calls = 0; wall0 = now_s(); }
        const double t0 = 10.0 + 0.05 * k;
        for (size_t i = 0; i < msgs.size(); ++i) { msgs[i].now = t0 + 1e-8 * i; msgs[i].seq = k & 0xff; }
        const double a0 = now_s();
        for (size_t i = 0; i < msgs.size(); i += batch) node.onRangingBatch(&msgs[i], std::min(batch, msgs.size() - i));
        const double a1 = now_s();
        calls += node.poll(t0 + 0.04); /* before the 50 ms timers: the epoch flushed by the new sequence numbers */
        const double a2 = now_s();
        t_ingest += a1 - a0;
        t_poll += a2 - a1;
    }
    std::vector<double> ahead(T, 0.0), pos(3 * (size_t)T);
    std::vector<uint32_t> st(T);
    kfpos_get_pose_each(h, ahead.data(), pos.data(), nullptr, nullptr, st.data()); /* waits for the last round */
    const double wall = now_s() - wall0;
    const double n = (double)msgs.size() * K;
    std::printf("{\"tags\": %d, \"anchors\": %d, \"model\": %d, \"epochs\": %d, \"messages\": %.0f, \"batch\": %zu, "
                "\"ingest_Mmsg_per_s\": %.2f, \"ingest_ms_per_epoch\": %.3f, \"poll_ms_per_epoch\": %.3f, "
                "\"end_to_end_Mmsg_per_s\": %.2f, \"end_to_end_ms_per_epoch\": %.3f, \"estimator_calls\": %d, "
                "\"overflow_calls\": %llu, \"pose0\": [%.4f, %.4f, %.4f], \"realtime_factor_at_20Hz\": %.2f}\n",
                T, A, model, K, n, batch, n / t_ingest / 1e6, t_ingest / K * 1e3, t_poll / K * 1e3, n / wall / 1e6,
                wall / K * 1e3, calls, (unsigned long long)node.overflowCalls(), pos[0], pos[1], pos[2],
                0.05 / (wall / K));
    kfpos_destroy(h);
    return 0;
}

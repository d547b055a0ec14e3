/*
 * msan_driver.cpp -- MemorySanitizer audit of the per-lane kernel body (TEST INFRASTRUCTURE; CPU only).
 *
 * Round 1 dropped a register-resident 16-anchor kernel whose scratch-spilling build returned wrong, run-to-run varying
 * positions on partially filled wavefronts. A correct program does not change its results when it spills, and
 * run-to-run variation means an undefined value is consumed somewhere -- either by the source (a lane-private array read
 * before it is written) or by the generated code. This driver rules the first out: it runs every code path of
 * kfpos_core*.h that the step kernels instantiate (register-resident epoch at 4 / 8 / 16 anchors, compile-time loops over
 * the strided scratch at 16, run-time loop; no heuristic / top-N / leave-one-out; fixed start and ML initialisation;
 * 9-state with fresh, latched and absent IMU samples) over ragged traces -- absent ranges whose errorEstimation is
 * POISONED, epochs with 0-3 ranges, lanes skipped with dt < 0, banks whose size is not a multiple of 64 -- with the
 * working-weight scratch poisoned before every step (kfpos_emu.cpp, KFE_MSAN), and checks every output word.
 * Build + run: tests/emu/msan_audit.sh. Exit code 0 and "msan audit: clean" = no use of an uninitialised value.
 */
#include <sanitizer/msan_interface.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

struct kfe_bank;
extern "C" {
kfe_bank *kfe_create(int model, int n_tags, int n_anchors, const double *anchors, double accel_noise, double jolt,
                     int ignore_worst, double cost_threshold, int top_n, int use_init_pos, const double *init_pos);
void kfe_destroy(kfe_bank *b);
void kfe_set_static(kfe_bank *b, int on);
void kfe_step_toa(kfe_bank *b, const int32_t *range_mm, const double *err_est, const double *dt, int dt_len,
                  uint32_t *status);
void kfe_step_imu(kfe_bank *b, const double *accel, const double *cov, const double *dt, int dt_len, uint32_t *status);
void kfe_latch_imu(kfe_bank *b, const double *accel, const double *cov);
void kfe_get_state(const kfe_bank *b, double *x, double *P);
void kfe_get_pose(const kfe_bank *b, double dt_ahead, double *pos, double *cov3x3, double *vel);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double uni() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}

static void anchors_of(int A, std::vector<double> &xyz) {
    xyz.resize(3 * A);
    for (int i = 0; i < A; ++i) {
        if (i < 8) { xyz[3 * i] = 10.0 * (i & 1); xyz[3 * i + 1] = 10.0 * ((i >> 1) & 1); xyz[3 * i + 2] = 0.3 + 2.7 * ((i >> 2) & 1); }
        else { xyz[3 * i] = 5.0 + 3.0 * std::cos((double)i); xyz[3 * i + 1] = 5.0 + 3.0 * std::sin((double)i); xyz[3 * i + 2] = 1.5 + 0.1 * i; }
    }
}

static int run_case(const char *name, int model, int A, int use_static, int ignore_worst, int top_n, int fixed) {
    const int T = 37, S = 30; /* 37: a partially filled wavefront on the GPU */
    std::vector<double> anchors;
    anchors_of(A, anchors);
    std::vector<double> init(3 * T), truth(3 * T);
    for (int t = 0; t < T; ++t) { truth[3 * t] = 2 + 6 * uni(); truth[3 * t + 1] = 2 + 6 * uni(); truth[3 * t + 2] = 0.8 + 0.6 * uni(); }
    init = truth;
    kfe_bank *b = kfe_create(model, T, A, anchors.data(), 0.5, 0.5, ignore_worst, 0.5, top_n, fixed, fixed ? init.data() : nullptr);
    kfe_set_static(b, use_static);
    const int n = model == 1 ? 9 : 6;
    std::vector<int32_t> mm(T * A);
    std::vector<double> err(T * A), dt(T), acc(3 * T), cov(9 * T), x(T * n), P(T * n * n), pos(3 * T), c3(9 * T), vel(3 * T);
    std::vector<uint32_t> st(T);
    for (int s = 0; s < S; ++s) {
        for (int t = 0; t < T; ++t) {
            truth[3 * t] += 0.02 * (uni() - 0.5); truth[3 * t + 1] += 0.02 * (uni() - 0.5);
            dt[t] = (s == 0) ? 0.1 : 0.05;
            if (s > 2 && (t + s) % 9 == 4) dt[t] = -1.0; /* no epoch for this tag in this call */
            for (int a = 0; a < A; ++a) {
                const double dx = truth[3 * t] - anchors[3 * a], dy = truth[3 * t + 1] - anchors[3 * a + 1], dz = truth[3 * t + 2] - anchors[3 * a + 2];
                double r = std::sqrt(dx * dx + dy * dy + dz * dz) + 0.05 * (uni() - 0.5);
                if (a == 3 && t % 5 == 0) r += 0.8; /* NLOS-like bias: what the heuristics react to */
                mm[t * A + a] = (int32_t)std::floor(r * 1000.0);
                err[t * A + a] = 0.0025 * (1.0 + (a % 3));
                bool absent = (s % 7 == 3 && a == 1) || (s % 11 == 5 && t % 3 == 0 && a >= 2) || (s % 13 == 9 && t % 4 == 1);
                if (absent) {
                    mm[t * A + a] = (a & 1) ? 0 : -1;
                    __msan_poison(&err[t * A + a], sizeof(double)); /* the caller left garbage there */
                }
            }
            for (int k = 0; k < 3; ++k) acc[3 * t + k] = 0.1 * (uni() - 0.5);
            for (int k = 0; k < 9; ++k) cov[9 * t + k] = (k % 4 == 0) ? 0.01 : ((t & 1) ? 0.001 : 0.0);
        }
        if (model == 1) {
            if (s % 3 == 0) kfe_latch_imu(b, acc.data(), cov.data());          /* fused epoch */
            else if (s % 3 == 1) kfe_step_imu(b, acc.data(), cov.data(), dt.data(), T, st.data()); /* separate IMU call */
        }
        kfe_step_toa(b, mm.data(), err.data(), dt.data(), T, st.data());
        kfe_get_state(b, x.data(), P.data());
        kfe_get_pose(b, 0.02, pos.data(), c3.data(), vel.data());
        if (__msan_test_shadow(st.data(), st.size() * sizeof(uint32_t)) >= 0 ||
            __msan_test_shadow(x.data(), x.size() * sizeof(double)) >= 0 ||
            __msan_test_shadow(P.data(), P.size() * sizeof(double)) >= 0 ||
            __msan_test_shadow(pos.data(), pos.size() * sizeof(double)) >= 0 ||
            __msan_test_shadow(c3.data(), c3.size() * sizeof(double)) >= 0) {
            std::printf("%s: step %d leaves uninitialised bytes in its outputs\n", name, s);
            return 1;
        }
        for (size_t k = 0; k < err.size(); ++k) __msan_unpoison(&err[k], sizeof(double));
    }
    kfe_destroy(b);
    std::printf("%-28s clean\n", name);
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "--selftest") {
        /* the sanitizer must abort here: a branch on a poisoned value (proves the audit can see what it looks for) */
        volatile double v = 1.0;
        double w = v;
        __msan_poison(&w, sizeof(w));
        if (w > 0.5) std::printf("selftest: branch taken\n");
        std::printf("selftest: MemorySanitizer did NOT stop a branch on an uninitialised value\n");
        return 0;
    }
    int bad = 0;
    /*            name                         model A  static iw topn fixed */
    bad += run_case("toa6 A16 regs",              0, 16, 1, 0, 0, 1);
    bad += run_case("toa6 A16 regs top-N",        0, 16, 1, 0, 2, 1);
    bad += run_case("toa6 A16 regs leave-one-out", 0, 16, 1, 1, 0, 1);
    bad += run_case("toa6 A16 regs loo+topN",     0, 16, 1, 1, 2, 1);
    bad += run_case("toa6 A16 regs ML-init",      0, 16, 1, 1, 0, 0);
    bad += run_case("toa6 A16 static-lds top-N",  0, 16, 2, 0, 2, 1);
    bad += run_case("toa6 A16 static-lds loo",    0, 16, 2, 1, 0, 1);
    bad += run_case("toa6 A8 regs",               0, 8, 1, 0, 0, 1);
    bad += run_case("toa6 A8 regs leave-one-out", 0, 8, 1, 1, 0, 1);
    bad += run_case("toa6 A8 regs ML-init",       0, 8, 1, 0, 0, 0);
    bad += run_case("toa6 A4 regs",               0, 4, 1, 0, 0, 1);
    bad += run_case("toa6 A5 run-time loop",      0, 5, 0, 1, 1, 1);
    bad += run_case("toa6 A12 run-time ML-init",  0, 12, 0, 1, 0, 0);
    bad += run_case("imu9 A8 regs",               1, 8, 1, 0, 0, 1);
    bad += run_case("imu9 A8 regs ML-init",       1, 8, 1, 0, 0, 0);
    bad += run_case("imu9 A16 static-lds",        1, 16, 2, 0, 0, 1);
    bad += run_case("imu9 A12 run-time loop",     1, 12, 0, 0, 0, 1);
    if (bad) return 1;
    std::printf("msan audit: clean\n");
    return 0;
}

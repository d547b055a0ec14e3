/*
 * kfpos_internal.h -- what the translation units of libkfpos_hip.so share and the ABI does not show: the handle.
 * (kfpos_k_*.hip: the kernels; kfpos_hip.hip: host side and filter entry points; kfpos_comm.hip: the RCCL pose gather.)
 */
#ifndef KFPOS_INTERNAL_H
#define KFPOS_INTERNAL_H

#include <hip/hip_runtime.h>

#include <string>

#include "../../include/kfpos.h"

#define KFPOS_TRACE_CHUNK 128 /* epochs per multi-epoch launch (their dt values travel in the kernel arguments) */
#define KFPOS_N_SLOTS 3       /* streaming host API: one slot being filled, one on the bus, one computing / returning */

/* thread-local text behind kfpos_last_error() */
__attribute__((visibility("hidden"))) std::string &kfpos_error_text();

/* Every entry point runs on its handle's device whatever the calling thread's current device is (a process that
 * drives one handle per GPU from one thread: kfpos_comm_create_all), and leaves the caller's device as it found it. */
struct KfposDevScope {
    int prev = -1;
    bool switched = false;
    explicit KfposDevScope(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~KfposDevScope() {
        if (switched) (void)hipSetDevice(prev);
    }
    KfposDevScope(const KfposDevScope &) = delete;
    KfposDevScope &operator=(const KfposDevScope &) = delete;
};

struct kfpos_handle {
    kfpos_config cfg;
    int n;        /* state dimension */
    int full;     /* COV_FULL layout: 6-state with ML initialisation (non-symmetric P) */
    int psz;      /* stored covariance entries per tag */
    int rsz;      /* bytes per stored covariance entry */
    int msz;      /* sizeof(kfpos_real): bytes per measurement element */
    int A;        /* anchors set */
    bool have_anchors, stepped;
    int trace_chunk;    /* epochs per launch in kfpos_run_trace_dev (KFPOS_TRACE_CHUNK_STEPS, 1..128) */
    bool force_generic; /* KFPOS_GENERIC_KERNEL=1: always the LDS-staged kernel (A/B measurements, tests) */
    bool two_waves;     /* n_tags / 64 exceeds the device's SIMD count (KFPOS_ONE_WAVE_BUILD=1: never) */
    bool pair9;         /* 9-state bank: iekf9_pairs for the tail of the gain iteration; KFPOS_PAIR9=1 enables (built, bit-identical, measured: no gain worth having -- DESIGN 6a) */
    bool coop;          /* small plain 6-state bank: one tag per 8 lanes (k_step_toa6_coop); KFPOS_NO_COOP=1 disables */
    double anchors[KFPOS_MAX_ANCHORS * 3];
    /* device state */
    double *d_pos = nullptr;
    double *d_vel = nullptr;
    void *d_P = nullptr, *d_imu_acc = nullptr, *d_imu_cov = nullptr;
    uint32_t *d_flags = nullptr;
    /* staging for the host-buffer API */
    int32_t *d_ranges = nullptr;
    void *d_err = nullptr, *d_accel = nullptr, *d_cov = nullptr;
    double *d_dt = nullptr, *d_out = nullptr; /* d_out: [15][T] doubles for pose results */
    uint32_t *d_status = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    /* planar filter */
    kfpos_planar_config planar = {};
    bool planar_sensors = false; /* a PX4Flow / IMU / magnetometer / compass sample has been fed: latches are live */
    double *d_latch = nullptr;   /* [15][T] */
    double *d_sensor = nullptr;  /* [24][T] staging of one sensor sample */
    /* row-major staging area of the host-buffer API: one region per array of a call (bump-allocated) */
    unsigned char *d_stage = nullptr;
    size_t stage_cap = 0, stage_used = 0;
    /* small banks (the single-tag adaptor objects, test banks): ONE host-pinned, device-mapped block holds every
     * input and output of a host-API call in component-major form; the kernels read and write it in place over the
     * bus, so a call is "turn the layout on the CPU, launch, synchronise" -- no hipMemcpy, no layout kernels */
    unsigned char *sm_h = nullptr, *sm_d = nullptr;
    size_t sm_ranges = 0, sm_err = 0, sm_accel = 0, sm_cov = 0, sm_dt = 0, sm_sensor = 0, sm_status = 0, sm_out = 0;
    /* streaming host API: KFPOS_N_SLOTS slots of pinned host + device buffers, three streams (kfpos_slot_*) */
    struct Slot {
        unsigned char *host = nullptr; /* pinned block: ranges | err | accel | cov | dt | status | pos */
        unsigned char *dev = nullptr;  /* device block, same layout */
        hipEvent_t copied = nullptr, copied2 = nullptr, computed = nullptr, done = nullptr;
        /* `computed` of the last submission (of ANY slot) whose kernel read this slot's device errorEstimations /
         * sensor covariance: an upload into those regions waits for it, whichever slot holds the current ones by then */
        hipEvent_t err_reader = nullptr, cov_reader = nullptr;
        bool busy = false;
    } slot[KFPOS_N_SLOTS];
    size_t so_ranges = 0, so_err = 0, so_accel = 0, so_cov = 0, so_dt = 0, so_status = 0, so_pos = 0, so_bytes = 0;
    hipStream_t s_copy = nullptr, s_copy2 = nullptr, s_comp = nullptr, s_back = nullptr;
    size_t split_bytes = 0;                            /* H2D copies from this size on travel as two halves on two streams */
    int err_slot = -1, cov_slot = -1;                  /* which slot's device block holds the current err / cov */
};

#endif /* KFPOS_INTERNAL_H */

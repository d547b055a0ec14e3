/*
 * kfpos_comm.hip -- the one exchange of the sharded tag batch: an RCCL all-gather of poses over xGMI, behind the C ABI
 * (include/kfpos.h: kfpos_comm_*, kfpos_allgather_poses*). SURVEY.md 8(e); the reference has no counterpart (one filter,
 * one process: node_pos.cpp:176-181).
 *
 * Tags share nothing but the read-only anchor table, so rank g keeps the contiguous global tag range
 * kfpos_shard_range(total, world, g) on its own GPU for good and the filters never communicate. What a multi-tag node
 * publishes is the pose of EVERY tag, so once per epoch (or per launch of K epochs) the ranks exchange their pose blocks:
 *
 *   caller's stream:  [pack]   pos_local [rows][t_local] -> send[b] [rows][t_pad]   (pad to the largest shard: an
 *                              all-gather wants equal contributions; shards differ by at most one tag)
 *                     record ready[b]
 *   side stream:      wait ready[b]; ncclAllGather(send[b] -> staged[b] [world][rows][t_pad]);
 *                     [assemble] staged[b] -> pos_all [rows][total] in global tag order (padding dropped);
 *                     record done[b]
 *
 * Two buffer sets alternate (b = call parity), so the exchange of one epoch overlaps the compute of the next: a
 * 131 072-tag f64 pose shard is 3 MB -- on point-to-point xGMI links that is a latency-bound collective, and hiding it
 * is what protects scaling. pos_local may be overwritten as soon as the call returns (stream-ordered behind the pack);
 * pos_all is valid after kfpos_comm_wait() / kfpos_comm_sync().
 *
 * How the blocks travel (kfpos_comm_set_algorithm; KFPOS_GATHER_ALGO=direct|collective at communicator creation):
 *   collective  ncclAllGather: RCCL's own schedule (rings over the xGMI links)
 *   direct      every rank SENDS its block to each of the other world-1 ranks and receives theirs, all in one RCCL
 *               group (ncclSend / ncclRecv): on MI355X every pair of GPUs has its own xGMI link, so the world-1
 *               transfers of a rank run side by side, one hop each -- 3 MB per link for a 131 072-tag shard -- where a
 *               ring passes every block through world-1 hops. Which of the two is faster at a given size is a property
 *               of the machine: bench.py times both before its timed region and says what it picked.
 * Both fill the same staged [world][rows][t_pad] image, so everything around them is shared.
 *
 * librccl is opened on first use (dlopen), not linked: a process that never shards -- the single-tag adaptor objects, a
 * one-GPU node -- does not map a 570 MB library, and a process that already has RCCL (PyTorch's copy has the same
 * soname) shares it.
 */
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "kfpos_internal.h"

namespace {

#define g_err (kfpos_error_text())

#define HIPCHK(expr)                                                            \
    do {                                                                        \
        hipError_t e_ = (expr);                                                 \
        if (e_ != hipSuccess) {                                                 \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e_);          \
            return KFPOS_ERR_HIP;                                               \
        }                                                                       \
    } while (0)

/* ---- librccl, resolved at first use ---- */
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string why;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("KFPOS_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
            r.why += std::string(r.why.empty() ? "" : "; ") + dlerror();
        }
        if (!r.lib) return;
        bool ok = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(r.lib, name);
            if (!p) {
                ok = false;
                r.why += std::string("missing symbol ") + name;
            }
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
        if (!ok) {
            dlclose(r.lib);
            r.lib = nullptr;
        }
    });
    return r;
}

int need_rccl() {
    Rccl &r = rccl();
    if (r.lib) return KFPOS_OK;
    g_err = "RCCL is not available (" + r.why + "); set KFPOS_RCCL_PATH to librccl.so";
    return KFPOS_ERR_COMM;
}

#define NCCLCHK(expr)                                                                           \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess) {                                                                \
            g_err = std::string(#expr) + ": " + rccl().GetErrorString(r_);                      \
            return KFPOS_ERR_COMM;                                                              \
        }                                                                                       \
    } while (0)

/* ---- layout kernels (8-byte words; 256 lanes sweep consecutive tags: coalesced on both sides) ---- */

/* [rows][t_local] -> [rows][t_pad], padding columns zeroed */
__global__ __launch_bounds__(256) void k_pack_poses(const double *__restrict__ src, double *__restrict__ dst, int rows,
                                                    long long t_local, long long t_pad) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= t_pad) return;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) dst[(size_t)r * t_pad + t] = t < t_local ? src[(size_t)r * t_local + t] : 0.0;
}

/* staged [world][rows][t_pad] -> out [rows][total] in global tag order. The shards are the contiguous ranges of
 * kfpos_shard_range: the first `rem` ranks hold base + 1 tags, the others base, so the owner of global tag g is closed
 * form -- no offset table to read. */
__global__ __launch_bounds__(256) void k_assemble_poses(const double *__restrict__ staged, double *__restrict__ out,
                                                        int rows, long long total, long long base, int rem,
                                                        long long t_pad) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const long long cut = (base + 1) * rem; /* tags held by the ranks with one tag more */
    long long rank, col;
    if (g < cut) {
        rank = g / (base + 1);
        col = g - rank * (base + 1);
    } else {
        rank = rem + (g - cut) / base;
        col = (g - cut) - (rank - rem) * base;
    }
    for (int r = blockIdx.y; r < rows; r += gridDim.y)
        out[(size_t)r * total + g] = staged[((size_t)rank * rows + r) * t_pad + col];
}

} // namespace

struct kfpos_comm {
    ncclComm_t nccl = nullptr;
    int world = 1, rank = 0, device = 0;
    long long total = 0, base = 0, t_local = 0, t_pad = 0; /* set by kfpos_comm_set_total */
    int rem = 0;
    hipStream_t side = nullptr;
    hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    double *send[2] = {nullptr, nullptr}, *staged[2] = {nullptr, nullptr};
    size_t cap_rows = 0; /* rows the buffers are sized for */
    unsigned long long calls = 0;
    int last = -1; /* buffer set of the last gather */
    int algo = KFPOS_GATHER_COLLECTIVE; /* how the blocks travel: kfpos_comm_set_algorithm */
};

namespace {

int comm_finish_init(kfpos_comm *c) {
    KfposDevScope dev(c->device);
    const char *algo = getenv("KFPOS_GATHER_ALGO");
    if (algo && std::strcmp(algo, "direct") == 0) c->algo = KFPOS_GATHER_DIRECT;
    HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
        HIPCHK(hipEventCreateWithFlags(&c->ready[b], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->done[b], hipEventDisableTiming));
    }
    return KFPOS_OK;
}

int comm_reserve(kfpos_comm *c, int rows) {
    if ((size_t)rows <= c->cap_rows) return KFPOS_OK;
    /* growing: nothing may still be reading the old buffers */
    HIPCHK(hipStreamSynchronize(c->side));
    for (int b = 0; b < 2; ++b) {
        if (c->send[b]) (void)hipFree(c->send[b]);
        if (c->staged[b]) (void)hipFree(c->staged[b]);
        c->send[b] = c->staged[b] = nullptr;
        c->used[b] = false;
    }
    c->cap_rows = 0;
    const size_t one = (size_t)rows * c->t_pad * sizeof(double);
    for (int b = 0; b < 2; ++b) {
        HIPCHK(hipMalloc((void **)&c->send[b], one));
        HIPCHK(hipMalloc((void **)&c->staged[b], one * c->world));
    }
    c->cap_rows = rows;
    return KFPOS_OK;
}

struct GatherPlan {
    kfpos_comm *c;
    int b, rows;
    double *out;
};

/* everything in front of the collective: pack on the caller's stream, hand over to the side stream */
int gather_front(kfpos_handle *h, kfpos_comm *c, const double *pos_local, int rows, double *pos_all, hipStream_t s,
                 GatherPlan &plan) {
    if (!c || !pos_all || rows < 1) return KFPOS_ERR_ARG;
    if (c->total <= 0) {
        g_err = "kfpos_comm_set_total has not been called";
        return KFPOS_ERR_STATE;
    }
    if (!pos_local) { /* the handle's current positions: what getPose at timeLag 0 returns for every tag of the shard */
        if (!h || rows != 3) return KFPOS_ERR_ARG;
        pos_local = h->d_pos;
    }
    if (h && (h->cfg.n_tags != c->t_local || h->cfg.device != c->device)) {
        g_err = "handle and communicator disagree: the handle holds " + std::to_string(h->cfg.n_tags) + " tags on device " +
                std::to_string(h->cfg.device) + ", rank " + std::to_string(c->rank) + "'s shard is " +
                std::to_string(c->t_local) + " tags on device " + std::to_string(c->device);
        return KFPOS_ERR_ARG;
    }
    int rc = comm_reserve(c, rows);
    if (rc) return rc;
    const int b = (int)(c->calls & 1);
    if (c->used[b]) HIPCHK(hipStreamWaitEvent(s, c->done[b], 0)); /* the gather two calls ago has drained this set */
    const dim3 grid((unsigned)((c->t_pad + 255) / 256), (unsigned)(rows < 64 ? rows : 64));
    hipLaunchKernelGGL(k_pack_poses, grid, dim3(256), 0, s, pos_local, c->send[b], rows, c->t_local, c->t_pad);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ready[b], s));
    HIPCHK(hipStreamWaitEvent(c->side, c->ready[b], 0));
    plan = GatherPlan{c, b, rows, pos_all};
    return KFPOS_OK;
}

/* the exchange itself, enqueued on the side stream; inside an RCCL group when `grouped` (the caller opened it) */
ncclResult_t exchange(const GatherPlan &p, bool grouped) {
    kfpos_comm *c = p.c;
    const size_t count = (size_t)p.rows * c->t_pad;
    if (c->algo != KFPOS_GATHER_DIRECT)
        return rccl().AllGather(c->send[p.b], c->staged[p.b], count, ncclDouble, c->nccl, c->side);
    ncclResult_t first = ncclSuccess;
    if (!grouped) first = rccl().GroupStart();
    for (int peer = 0; peer < c->world && first == ncclSuccess; ++peer) {
        if (peer == c->rank) continue;
        first = rccl().Send(c->send[p.b], count, ncclDouble, peer, c->nccl, c->side);
        if (first == ncclSuccess)
            first = rccl().Recv(c->staged[p.b] + (size_t)peer * count, count, ncclDouble, peer, c->nccl, c->side);
    }
    if (!grouped) {
        const ncclResult_t e = rccl().GroupEnd();
        if (first == ncclSuccess) first = e;
    }
    /* this rank's own block does not travel */
    if (first == ncclSuccess &&
        hipMemcpyAsync(c->staged[p.b] + (size_t)c->rank * count, c->send[p.b], count * sizeof(double),
                       hipMemcpyDeviceToDevice, c->side) != hipSuccess)
        first = ncclUnhandledCudaError;
    return first;
}

/* everything behind it, on the side stream */
int gather_back(const GatherPlan &p) {
    kfpos_comm *c = p.c;
    const dim3 grid((unsigned)((c->total + 255) / 256), (unsigned)(p.rows < 64 ? p.rows : 64));
    hipLaunchKernelGGL(k_assemble_poses, grid, dim3(256), 0, c->side, c->staged[p.b], p.out, p.rows, c->total, c->base,
                       c->rem, c->t_pad);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->done[p.b], c->side));
    c->used[p.b] = true;
    c->last = p.b;
    c->calls++;
    return KFPOS_OK;
}

} // namespace

extern "C" {

int kfpos_shard_range(int64_t total_tags, int32_t world, int32_t rank, int64_t *lo, int64_t *hi) {
    if (total_tags < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return KFPOS_ERR_ARG;
    const int64_t base = total_tags / world, rem = total_tags % world;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
    return KFPOS_OK;
}

int kfpos_comm_unique_id(void *id_out) {
    g_err.clear();
    if (!id_out) return KFPOS_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) == KFPOS_COMM_ID_BYTES, "kfpos.h: KFPOS_COMM_ID_BYTES");
    int rc = need_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    NCCLCHK(rccl().GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return KFPOS_OK;
}

int kfpos_comm_create(int32_t world, int32_t rank, const void *unique_id, int32_t device, kfpos_comm **out) {
    g_err.clear();
    if (!out) return KFPOS_ERR_ARG;
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || !unique_id) return KFPOS_ERR_ARG;
    int rc = need_rccl();
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        g_err = "no such HIP device";
        return KFPOS_ERR_NO_DEVICE;
    }
    kfpos_comm *c = new (std::nothrow) kfpos_comm();
    if (!c) return KFPOS_ERR_ARG;
    c->world = world;
    c->rank = rank;
    c->device = device;
    KfposDevScope dev(device); /* ncclCommInitRank binds the communicator to the current device */
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = rccl().CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
        g_err = std::string("ncclCommInitRank: ") + rccl().GetErrorString(r);
        delete c;
        return KFPOS_ERR_COMM;
    }
    if ((rc = comm_finish_init(c))) {
        kfpos_comm_destroy(c);
        return rc;
    }
    *out = c;
    return KFPOS_OK;
}

int kfpos_comm_create_all(int32_t n_devices, const int32_t *devices, kfpos_comm **out) {
    g_err.clear();
    if (n_devices < 1 || !out) return KFPOS_ERR_ARG;
    for (int i = 0; i < n_devices; ++i) out[i] = nullptr;
    int rc = need_rccl();
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        g_err = "no HIP device";
        return KFPOS_ERR_NO_DEVICE;
    }
    std::vector<int> devs(n_devices);
    for (int i = 0; i < n_devices; ++i) {
        devs[i] = devices ? devices[i] : i;
        if (devs[i] < 0 || devs[i] >= ndev) {
            g_err = "no such HIP device";
            return KFPOS_ERR_NO_DEVICE;
        }
    }
    std::vector<ncclComm_t> comms(n_devices, nullptr);
    int prev = 0;
    (void)hipGetDevice(&prev);
    ncclResult_t r = rccl().CommInitAll(comms.data(), n_devices, devs.data());
    (void)hipSetDevice(prev); /* ncclCommInitAll walks the devices */
    if (r != ncclSuccess) {
        g_err = std::string("ncclCommInitAll: ") + rccl().GetErrorString(r);
        return KFPOS_ERR_COMM;
    }
    rc = KFPOS_OK;
    for (int i = 0; i < n_devices; ++i) { /* every communicator gets an owner first, so that a failure frees them all */
        kfpos_comm *c = new (std::nothrow) kfpos_comm();
        if (!c) {
            (void)rccl().CommAbort(comms[i]);
            rc = KFPOS_ERR_ARG;
            continue;
        }
        c->nccl = comms[i];
        c->world = n_devices;
        c->rank = i;
        c->device = devs[i];
        out[i] = c;
    }
    for (int i = 0; i < n_devices && rc == KFPOS_OK; ++i) rc = comm_finish_init(out[i]);
    if (rc) {
        const std::string why = g_err;
        for (int i = 0; i < n_devices; ++i) {
            if (out[i]) kfpos_comm_destroy(out[i]);
            out[i] = nullptr;
        }
        g_err = why;
    }
    return rc;
}

int kfpos_comm_destroy(kfpos_comm *c) {
    if (!c) return KFPOS_ERR_ARG;
    KfposDevScope dev(c->device);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->nccl) (void)rccl().CommDestroy(c->nccl);
    for (int b = 0; b < 2; ++b) {
        if (c->send[b]) (void)hipFree(c->send[b]);
        if (c->staged[b]) (void)hipFree(c->staged[b]);
        if (c->ready[b]) (void)hipEventDestroy(c->ready[b]);
        if (c->done[b]) (void)hipEventDestroy(c->done[b]);
    }
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    return KFPOS_OK;
}

int kfpos_comm_set_algorithm(kfpos_comm *c, int32_t algorithm) {
    g_err.clear();
    if (!c || (algorithm != KFPOS_GATHER_COLLECTIVE && algorithm != KFPOS_GATHER_DIRECT)) return KFPOS_ERR_ARG;
    KfposDevScope dev(c->device);
    if (c->side) HIPCHK(hipStreamSynchronize(c->side)); /* between two gathers, never inside one */
    c->algo = algorithm;
    return KFPOS_OK;
}
int kfpos_comm_algorithm(const kfpos_comm *c) { return c ? c->algo : -1; }

int kfpos_comm_world(const kfpos_comm *c) { return c ? c->world : 0; }
int kfpos_comm_rank(const kfpos_comm *c) { return c ? c->rank : -1; }

int kfpos_comm_set_total(kfpos_comm *c, int64_t total_tags, int64_t *lo, int64_t *hi) {
    g_err.clear();
    if (!c || total_tags < c->world) return KFPOS_ERR_ARG; /* every rank holds at least one tag */
    KfposDevScope dev(c->device);
    if (c->side) HIPCHK(hipStreamSynchronize(c->side));
    int64_t l = 0, u = 0;
    kfpos_shard_range(total_tags, c->world, c->rank, &l, &u);
    c->total = total_tags;
    c->base = total_tags / c->world;
    c->rem = (int)(total_tags % c->world);
    c->t_local = u - l;
    c->t_pad = c->base + (c->rem ? 1 : 0);
    for (int b = 0; b < 2; ++b) { /* buffers are sized by t_pad: start over */
        if (c->send[b]) (void)hipFree(c->send[b]);
        if (c->staged[b]) (void)hipFree(c->staged[b]);
        c->send[b] = c->staged[b] = nullptr;
        c->used[b] = false;
    }
    c->cap_rows = 0;
    if (lo) *lo = l;
    if (hi) *hi = u;
    return KFPOS_OK;
}

int kfpos_allgather_poses(kfpos_handle *h, kfpos_comm *c, const double *pos_local, int32_t rows, double *pos_all,
                          void *stream) {
    g_err.clear();
    if (!c) return KFPOS_ERR_ARG;
    KfposDevScope dev(c->device);
    GatherPlan p;
    int rc = gather_front(h, c, pos_local, rows, pos_all, (hipStream_t)stream, p);
    if (rc) return rc;
    NCCLCHK(exchange(p, false));
    return gather_back(p);
}

int kfpos_allgather_poses_multi(int32_t n, kfpos_handle *const *handles, kfpos_comm *const *comms,
                                const double *const *pos_local, int32_t rows, double *const *pos_all,
                                void *const *streams) {
    g_err.clear();
    if (n < 1 || !comms || !pos_all) return KFPOS_ERR_ARG;
    std::vector<GatherPlan> plans(n);
    for (int i = 0; i < n; ++i) {
        if (!comms[i]) return KFPOS_ERR_ARG;
        KfposDevScope dev(comms[i]->device);
        int rc = gather_front(handles ? handles[i] : nullptr, comms[i], pos_local ? pos_local[i] : nullptr, rows,
                              pos_all[i], streams ? (hipStream_t)streams[i] : nullptr, plans[i]);
        if (rc) return rc;
    }
    /* one thread, several devices: the collective calls of all ranks form ONE group, or the first would wait for the
     * others for ever */
    NCCLCHK(rccl().GroupStart());
    ncclResult_t first = ncclSuccess;
    for (int i = 0; i < n; ++i) {
        KfposDevScope dev(comms[i]->device); /* (the copy of a rank's own block is enqueued by the HIP runtime, not by RCCL) */
        const ncclResult_t r = exchange(plans[i], true);
        if (r != ncclSuccess && first == ncclSuccess) first = r;
    }
    ncclResult_t e = rccl().GroupEnd();
    if (first == ncclSuccess) first = e;
    if (first != ncclSuccess) {
        g_err = std::string("pose exchange (RCCL group): ") + rccl().GetErrorString(first);
        return KFPOS_ERR_COMM;
    }
    for (int i = 0; i < n; ++i) {
        KfposDevScope dev(comms[i]->device);
        int rc = gather_back(plans[i]);
        if (rc) return rc;
    }
    return KFPOS_OK;
}

int kfpos_comm_wait(kfpos_comm *c, void *stream) {
    g_err.clear();
    if (!c) return KFPOS_ERR_ARG;
    KfposDevScope dev(c->device);
    if (c->last >= 0) HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->done[c->last], 0));
    return KFPOS_OK;
}

int kfpos_comm_sync(kfpos_comm *c) {
    g_err.clear();
    if (!c) return KFPOS_ERR_ARG;
    KfposDevScope dev(c->device);
    HIPCHK(hipStreamSynchronize(c->side));
    return KFPOS_OK;
}

int kfpos_assemble_poses_dev(int32_t world, int32_t rows, int64_t total_tags, const double *staged, double *out,
                             int32_t device, void *stream) {
    g_err.clear();
    if (world < 1 || rows < 1 || total_tags < world || !staged || !out) return KFPOS_ERR_ARG;
    KfposDevScope dev(device);
    const long long base = total_tags / world;
    const int rem = (int)(total_tags % world);
    const long long t_pad = base + (rem ? 1 : 0);
    const dim3 grid((unsigned)((total_tags + 255) / 256), (unsigned)(rows < 64 ? rows : 64));
    hipLaunchKernelGGL(k_assemble_poses, grid, dim3(256), 0, (hipStream_t)stream, staged, out, rows, (long long)total_tags,
                       base, rem, t_pad);
    HIPCHK(hipGetLastError());
    return KFPOS_OK;
}

int kfpos_comm_backend_version(void) {
    if (need_rccl()) return 0;
    int v = 0;
    return rccl().GetVersion(&v) == ncclSuccess ? v : 0;
}

} // extern "C"

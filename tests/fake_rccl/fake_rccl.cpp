/*
 * fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for the dozen librccl entry points kfpos_comm.hip resolves with
 * dlopen, so that the pose exchange of the C ABI (include/kfpos.h: kfpos_comm_*, kfpos_allgather_poses*) can run at
 * world sizes 2..16 on the ONE card of a GPU box. RCCL itself refuses two ranks on one device, so without this the
 * world > 1 branches -- padding of unequal shards, the peer loop of the direct exchange, the grouped single-process
 * form, the assembly of blocks that really came from other ranks -- would meet a transport for the first time on the
 * 8-GPU node. What it does NOT stand in for is RCCL's own behaviour (topology, channels, kernels): that stays "first
 * contact". The product never loads this: it is selected by KFPOS_RCCL_PATH=<this library> in tests only.
 *
 * Transport: ranks rendezvous in a POSIX shared-memory segment named by the "unique id". An exchange round is
 *   1. wait for the stream the operation was enqueued on (what precedes it in stream order is therefore visible, what
 *      the caller forgot to order is not -- as with the real library), copy the payloads device -> mailbox[rank];
 *   2. barrier over all ranks of the communicator; every rank CHECKS the others' postings against its own -- equal
 *      count and type for an all-gather, a matching send for every receive and a matching receive for every send --
 *      and fails with ncclInvalidUsage / ncclInvalidArgument where RCCL would hang or corrupt memory;
 *   3. copy mailbox[peer] -> receive buffers; barrier (the mailboxes are free again).
 * The calls are synchronous on the host (a stronger ordering than "enqueued on the stream", never a weaker one). Every
 * round involves every rank of the communicator -- true for an all-gather and for kfpos's direct exchange; a round a
 * rank stays away from ends in ncclSystemError after FAKE_RCCL_TIMEOUT_S (default 30 s), never in a hang.
 * Several ranks in one process (ncclCommInitAll, or the same device listed more than once -- allowed here, which is the
 * point) must issue a round inside ONE ncclGroupStart/End, as with RCCL.
 *
 *   hipcc -O2 -std=c++17 -shared -fPIC -o libfake_rccl.so fake_rccl.cpp -lrt
 */
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

constexpr int MAX_RANKS = 16, MAX_P2P = 32;
constexpr int FAKE_VERSION = 99999; /* what ncclGetVersion reports: tests can tell the stand-in from RCCL */

enum Kind : int { K_NONE = 0, K_ALLGATHER = 1, K_P2P = 2 };

struct Posting { /* what one rank announces for one round */
    uint64_t seq;
    int kind;
    uint64_t count; /* all-gather: elements per rank */
    int dtype;
    int n_send, n_recv;
    struct {
        int peer, dtype;
        uint64_t count, offset; /* offset of the payload in this rank's mailbox */
    } send[MAX_P2P], recv[MAX_P2P];
};

struct Segment { /* the shared-memory image; a fresh segment is all zeroes, which is its valid initial state */
    std::atomic<int> world;
    std::atomic<uint64_t> joined, arrive_a, arrive_b;
    std::atomic<int> failed; /* sticky: one rank's verdict is everybody's */
    uint64_t mailbox_bytes;
    Posting post[MAX_RANKS];
    /* mailboxes follow: MAX_RANKS x mailbox_bytes */
};

size_t mailbox_bytes_env() {
    const char *e = getenv("FAKE_RCCL_MAILBOX_MB");
    const long mb = e ? atol(e) : 32;
    return (size_t)(mb > 0 ? mb : 32) << 20;
}
double timeout_s() {
    const char *e = getenv("FAKE_RCCL_TIMEOUT_S");
    const double t = e ? atof(e) : 30.0;
    return t > 0 ? t : 30.0;
}
double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

struct Mapping {
    Segment *seg = nullptr;
    size_t bytes = 0;
    int refs = 0;
    char *mailbox(int rank) const { return (char *)(seg + 1) + (size_t)rank * seg->mailbox_bytes; }
};

std::atomic<uint64_t> g_stats[4]; /* rounds, payload bytes copied in, all-gather operations, send+recv operations */
std::atomic<uint64_t> g_ids;

size_t dtype_size(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

bool wait_for(std::atomic<uint64_t> &counter, uint64_t target, Segment *seg) {
    const double deadline = now_s() + timeout_s();
    unsigned spins = 0;
    while (counter.load(std::memory_order_acquire) < target) {
        if (++spins > 2000) {
            usleep(50);
            if (now_s() > deadline) {
                seg->failed.store((int)ncclSystemError);
                return false;
            }
        }
    }
    return true;
}

} // namespace

struct ncclComm {
    Mapping *map = nullptr;
    int world = 0, rank = 0, device = 0;
    uint64_t seq = 0;
};

namespace {

struct Op {
    int kind; /* K_ALLGATHER, or K_P2P with send = true / false */
    bool send;
    ncclComm *comm;
    const void *src;
    void *dst;
    size_t count;
    ncclDataType_t dtype;
    int peer;
    hipStream_t stream;
};

thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

Mapping *open_segment(const char *name, int world, ncclResult_t &why) {
    why = ncclSystemError;
    const size_t mb = mailbox_bytes_env(), bytes = sizeof(Segment) + (size_t)MAX_RANKS * mb;
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return nullptr;
    if (ftruncate(fd, (off_t)bytes) != 0) { /* every rank sets the same size; new pages read as zero */
        close(fd);
        return nullptr;
    }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return nullptr;
    Mapping *m = new Mapping();
    m->seg = (Segment *)p;
    m->bytes = bytes;
    int expect = 0;
    if (!m->seg->world.compare_exchange_strong(expect, world) && expect != world) {
        munmap(p, bytes);
        delete m;
        why = ncclInvalidArgument; /* the ranks disagree about the size of the communicator */
        return nullptr;
    }
    m->seg->mailbox_bytes = mb; /* same value from every rank */
    why = ncclSuccess;
    return m;
}

void release(Mapping *m) {
    if (m && --m->refs <= 0) {
        munmap(m->seg, m->bytes);
        delete m;
    }
}

/* one exchange round of the ranks this thread drives; ops are grouped by communicator */
ncclResult_t run_round(std::vector<Op> &ops) {
    if (ops.empty()) return ncclSuccess;
    std::vector<ncclComm *> comms;
    for (const Op &o : ops) {
        bool seen = false;
        for (ncclComm *c : comms) seen = seen || c == o.comm;
        if (!seen) comms.push_back(o.comm);
    }
    Mapping *map = comms[0]->map;
    for (ncclComm *c : comms)
        if (c->map != map) return ncclInvalidUsage; /* one clique per group is all the stand-in supports */
    Segment *seg = map->seg;
    ncclResult_t mine = ncclSuccess;
    int prev = 0;
    (void)hipGetDevice(&prev);

    /* 1. post and copy out */
    for (ncclComm *c : comms) {
        Posting &p = seg->post[c->rank];
        std::memset(&p, 0, sizeof(p));
        p.seq = c->seq;
        (void)hipSetDevice(c->device);
        size_t used = 0;
        for (const Op &o : ops) {
            if (o.comm != c) continue;
            const size_t es = dtype_size(o.dtype), bytes = o.count * es;
            if (!es) mine = ncclInvalidArgument;
            if (hipStreamSynchronize(o.stream) != hipSuccess) mine = ncclUnhandledCudaError;
            if (o.kind == K_ALLGATHER) {
                if (p.kind != K_NONE) mine = ncclInvalidUsage; /* one collective per round */
                p.kind = K_ALLGATHER;
                p.count = o.count;
                p.dtype = (int)o.dtype;
                g_stats[2]++;
            } else {
                if (p.kind == K_ALLGATHER) mine = ncclInvalidUsage;
                p.kind = K_P2P;
                g_stats[3]++;
                if (o.peer < 0 || o.peer >= c->world || o.peer == c->rank) {
                    mine = ncclInvalidArgument;
                    continue;
                }
                if (!o.send) {
                    if (p.n_recv >= MAX_P2P) { mine = ncclInvalidUsage; continue; }
                    auto &r = p.recv[p.n_recv++];
                    r.peer = o.peer, r.dtype = (int)o.dtype, r.count = o.count, r.offset = 0;
                    continue;
                }
                if (p.n_send >= MAX_P2P) { mine = ncclInvalidUsage; continue; }
                auto &s = p.send[p.n_send++];
                s.peer = o.peer, s.dtype = (int)o.dtype, s.count = o.count, s.offset = used;
            }
            if (used + bytes > seg->mailbox_bytes) {
                std::fprintf(stderr, "fake_rccl: %zu bytes do not fit a %llu-byte mailbox (FAKE_RCCL_MAILBOX_MB)\n",
                             used + bytes, (unsigned long long)seg->mailbox_bytes);
                mine = ncclInvalidArgument;
                continue;
            }
            if (bytes && hipMemcpy(map->mailbox(c->rank) + used, o.src, bytes, hipMemcpyDeviceToHost) != hipSuccess)
                mine = ncclUnhandledCudaError;
            used += bytes;
        }
    }
    if (mine != ncclSuccess) seg->failed.store((int)mine);
    const uint64_t round = comms[0]->seq;
    for (ncclComm *c : comms)
        if (c->seq != round) mine = ncclInvalidUsage;
    seg->arrive_a.fetch_add(comms.size(), std::memory_order_acq_rel);
    if (!wait_for(seg->arrive_a, (round + 1) * (uint64_t)comms[0]->world, seg)) {
        (void)hipSetDevice(prev);
        return ncclSystemError;
    }

    /* 2. check the others' postings, 3. copy in */
    for (ncclComm *c : comms) {
        const Posting &p = seg->post[c->rank];
        (void)hipSetDevice(c->device);
        for (int r = 0; r < c->world; ++r) {
            const Posting &q = seg->post[r];
            if (q.seq != p.seq || q.kind != p.kind) mine = ncclInvalidUsage;
            if (p.kind == K_ALLGATHER && (q.count != p.count || q.dtype != p.dtype)) mine = ncclInvalidArgument;
        }
        if (mine != ncclSuccess) continue;
        if (p.kind == K_ALLGATHER) {
            for (const Op &o : ops) {
                if (o.comm != c) continue;
                const size_t bytes = o.count * dtype_size(o.dtype);
                for (int r = 0; r < c->world && bytes; ++r) {
                    if (hipMemcpy((char *)o.dst + (size_t)r * bytes, map->mailbox(r), bytes, hipMemcpyHostToDevice) != hipSuccess)
                        mine = ncclUnhandledCudaError;
                    g_stats[1] += bytes;
                }
            }
            continue;
        }
        /* the k-th receive from peer P pairs with P's k-th send to this rank; counts and types must agree, and nothing
         * may stay unpaired on either side */
        std::vector<int> taken(c->world, 0);
        for (const Op &o : ops) {
            if (o.comm != c || o.send) continue;
            const Posting &q = seg->post[o.peer];
            int k = 0, found = -1;
            for (int i = 0; i < q.n_send; ++i)
                if (q.send[i].peer == c->rank && k++ == taken[o.peer]) found = i;
            taken[o.peer]++;
            if (found < 0) { mine = ncclInvalidUsage; continue; }
            if (q.send[found].count != o.count || q.send[found].dtype != (int)o.dtype) { mine = ncclInvalidArgument; continue; }
            const size_t bytes = o.count * dtype_size(o.dtype);
            if (bytes && hipMemcpy(o.dst, map->mailbox(o.peer) + q.send[found].offset, bytes, hipMemcpyHostToDevice) != hipSuccess)
                mine = ncclUnhandledCudaError;
            g_stats[1] += bytes;
        }
        for (int r = 0; r < c->world; ++r) {
            int to_r = 0, r_expects = 0;
            for (int i = 0; i < p.n_send; ++i) to_r += p.send[i].peer == r;
            for (int i = 0; i < seg->post[r].n_recv; ++i) r_expects += seg->post[r].recv[i].peer == c->rank;
            if (to_r != r_expects) mine = ncclInvalidUsage;
        }
    }
    if (mine != ncclSuccess) seg->failed.store((int)mine);
    seg->arrive_b.fetch_add(comms.size(), std::memory_order_acq_rel);
    const bool drained = wait_for(seg->arrive_b, (round + 1) * (uint64_t)comms[0]->world, seg);
    for (ncclComm *c : comms) c->seq++;
    g_stats[0]++;
    (void)hipSetDevice(prev);
    if (!drained) return ncclSystemError;
    const int verdict = seg->failed.load();
    return mine != ncclSuccess ? mine : (ncclResult_t)verdict;
}

ncclResult_t submit(const Op &o) {
    if (!o.comm || (!o.src && !o.dst)) return ncclInvalidArgument;
    t_ops.push_back(o);
    if (t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_round(ops);
}

} // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version) {
    if (!version) return ncclInvalidArgument;
    *version = FAKE_VERSION;
    return ncclSuccess;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "/fake_rccl_%d_%llu_%llu", (int)getpid(),
                  (unsigned long long)g_ids++, (unsigned long long)(now_s() * 1e6));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    if (id.internal[0] != '/') return ncclInvalidArgument;
    ncclResult_t why;
    Mapping *m = open_segment(id.internal, nranks, why);
    if (!m) return why;
    m->refs = 1;
    ncclComm *c = new ncclComm();
    c->map = m;
    c->world = nranks;
    c->rank = rank;
    (void)hipGetDevice(&c->device); /* bound to the current device, as with RCCL */
    m->seg->joined.fetch_add(1, std::memory_order_acq_rel);
    const bool all = wait_for(m->seg->joined, (uint64_t)nranks, m->seg); /* collective, like the real call */
    if (rank == 0) shm_unlink(id.internal);                              /* the mappings keep it alive; no name is left behind */
    if (!all) {
        release(m);
        delete c;
        return ncclSystemError;
    }
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
    if (!comms || ndev < 1 || ndev > MAX_RANKS) return ncclInvalidArgument;
    ncclUniqueId id;
    ncclGetUniqueId(&id);
    ncclResult_t why;
    Mapping *m = open_segment(id.internal, ndev, why);
    if (!m) return why;
    shm_unlink(id.internal);
    m->refs = ndev;
    m->seg->joined.store((uint64_t)ndev);
    for (int i = 0; i < ndev; ++i) {
        ncclComm *c = new ncclComm();
        c->map = m;
        c->world = ndev;
        c->rank = i;
        c->device = devlist ? devlist[i] : i;
        comms[i] = c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclInvalidArgument;
    release(comm->map);
    delete comm;
    return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t comm) { return ncclCommDestroy(comm); }

ncclResult_t ncclGroupStart() {
    ++t_depth;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_round(ops);
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    return submit(Op{K_ALLGATHER, true, comm, sendbuff, recvbuff, sendcount, datatype, -1, stream});
}
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return submit(Op{K_P2P, true, comm, sendbuff, nullptr, count, datatype, peer, stream});
}
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return submit(Op{K_P2P, false, comm, nullptr, recvbuff, count, datatype, peer, stream});
}

const char *ncclGetErrorString(ncclResult_t result) {
    switch (result) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake_rccl: a HIP call failed";
    case ncclSystemError: return "fake_rccl: a rank did not arrive (timeout) or the shared segment could not be opened";
    case ncclInvalidArgument: return "fake_rccl: invalid argument (counts / types / world size disagree between ranks, or the mailbox is too small)";
    case ncclInvalidUsage: return "fake_rccl: invalid usage (unpaired send/recv, mixed operations in one round, ranks out of step)";
    default: return "fake_rccl: error";
    }
}

/* for the tests: rounds, payload bytes delivered, all-gather operations, send + recv operations (this process) */
void fake_rccl_stats(uint64_t out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = g_stats[i].load();
}

} // extern "C"

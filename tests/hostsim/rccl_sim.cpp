// TEST INFRASTRUCTURE ONLY -- see tests/hostsim/rccl/rccl.h.  A communicator = one POSIX shared-memory control segment
// named after the unique id: a join counter and, per ordered pair (src, dst), the sequence numbers "posted" and "taken".
// A send number k from src to dst is its own small segment "<id>.<src>.<dst>.<k>" (u64 byte count + the bytes), published
// by posted = k; the receive number k waits for it, checks the count, copies, unlinks, sets taken = k.  Deadline on every
// wait (SMR_RCCL_SIM_TIMEOUT_S, default 20 s).
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <vector>

#include "rccl/rccl.h"

namespace {

constexpr int MAX_RANKS = 16;
struct Control {
    std::atomic<uint32_t> joined, left;
    std::atomic<uint64_t> posted[MAX_RANKS][MAX_RANKS], taken[MAX_RANKS][MAX_RANKS];
};

struct Op { bool send; int peer; const void *src; void *dst; size_t bytes; ncclSimComm *comm; };

thread_local int group_depth = 0;
thread_local std::vector<Op> group_ops;
thread_local std::string last_log;

double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
double deadline_s() { const char *e = getenv("SMR_RCCL_SIM_TIMEOUT_S"); return now_s() + (e ? atof(e) : 20.0); }
void nap() { timespec t = {0, 200000}; nanosleep(&t, nullptr); }

}  // namespace

struct ncclSimComm {
    std::string name;
    int rank = 0, world = 1;
    Control *ctl = nullptr;
    uint64_t sent[MAX_RANKS] = {0}, received[MAX_RANKS] = {0};   // my posting counters per peer
};

namespace {

std::string seg_name(const ncclSimComm *c, int src, int dst, uint64_t k) {
    char b[96];
    snprintf(b, sizeof b, "%s.%d.%d.%llu", c->name.c_str(), src, dst, (unsigned long long)k);
    return b;
}

ncclResult_t post_send(ncclSimComm *c, const Op &o, uint64_t *seq_out) {
    const uint64_t k = ++c->sent[o.peer];
    const std::string n = seg_name(c, c->rank, o.peer, k);
    int fd = shm_open(n.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    const size_t len = sizeof(uint64_t) + o.bytes;
    if (ftruncate(fd, (off_t)len) != 0) { close(fd); shm_unlink(n.c_str()); return ncclSystemError; }
    void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { shm_unlink(n.c_str()); return ncclSystemError; }
    *(uint64_t *)p = o.bytes;
    if (o.bytes) memcpy((char *)p + sizeof(uint64_t), o.src, o.bytes);
    munmap(p, len);
    c->ctl->posted[c->rank][o.peer].store(k, std::memory_order_release);
    *seq_out = k;
    return ncclSuccess;
}

ncclResult_t complete_recv(ncclSimComm *c, const Op &o) {
    const uint64_t k = ++c->received[o.peer];
    const double dl = deadline_s();
    while (c->ctl->posted[o.peer][c->rank].load(std::memory_order_acquire) < k) {
        if (now_s() > dl) { fprintf(stderr, "rccl_sim: rank %d: receive %llu from %d never met a send\n", c->rank, (unsigned long long)k, o.peer); return ncclSystemError; }
        nap();
    }
    const std::string n = seg_name(c, o.peer, c->rank, k);
    int fd = shm_open(n.c_str(), O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return ncclSystemError; }
    void *p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    const uint64_t got = *(const uint64_t *)p;
    ncclResult_t r = ncclSuccess;
    if (got != o.bytes) {                                       // RCCL would hang or overrun here: the stand-in says so
        fprintf(stderr, "rccl_sim: rank %d expected %zu bytes from rank %d (its message %llu), the send has %llu\n", c->rank, o.bytes, o.peer,
                (unsigned long long)k, (unsigned long long)got);
        r = ncclInvalidUsage;
    } else if (got) {
        memcpy(o.dst, (const char *)p + sizeof(uint64_t), got);
    }
    munmap(p, (size_t)st.st_size);
    shm_unlink(n.c_str());
    c->ctl->taken[o.peer][c->rank].store(k, std::memory_order_release);
    return r;
}

ncclResult_t wait_taken(ncclSimComm *c, int peer, uint64_t k) {
    const double dl = deadline_s();
    while (c->ctl->taken[c->rank][peer].load(std::memory_order_acquire) < k) {
        if (now_s() > dl) { fprintf(stderr, "rccl_sim: rank %d: send %llu to %d was never received\n", c->rank, (unsigned long long)k, peer); return ncclSystemError; }
        nap();
    }
    return ncclSuccess;
}

// the operations of one group (or one lone call): every send published, then every receive completed in posting order, then
// every send waited for -- a group's operations complete together, whatever order they were posted in
ncclResult_t run_ops(std::vector<Op> &ops) {
    ncclResult_t r = ncclSuccess;
    std::vector<std::pair<const Op *, uint64_t>> sends;
    last_log.clear();
    for (const Op &o : ops) {
        char b[48];
        snprintf(b, sizeof b, "%s%c%d:%zu", last_log.empty() ? "" : " ", o.send ? 'S' : 'R', o.peer, o.bytes);
        last_log += b;
    }
    for (const Op &o : ops)
        if (o.send && r == ncclSuccess) { uint64_t k = 0; r = post_send(o.comm, o, &k); sends.push_back({&o, k}); }
    for (const Op &o : ops) {                                   // a size mismatch fails THIS rank's call; the other receives are still taken so
        if (o.send || (r != ncclSuccess && r != ncclInvalidUsage)) continue;   // that the peers' groups complete (the message is consumed)
        const ncclResult_t e = complete_recv(o.comm, o);
        if (r == ncclSuccess) r = e;
    }
    if (r == ncclInvalidUsage) { for (auto &s : sends) (void)wait_taken(s.first->comm, s.first->peer, s.second); ops.clear(); return r; }
    for (auto &s : sends)
        if (r == ncclSuccess) r = wait_taken(s.first->comm, s.first->peer, s.second);
    ops.clear();
    return r;
}

size_t width(ncclDataType_t dt) { return dt == ncclInt8 || dt == ncclUint8 ? 1 : dt == ncclInt32 || dt == ncclUint32 ? 4 : 8; }

ncclResult_t enqueue(bool send, const void *src, void *dst, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm) {
    if (!comm || peer < 0 || peer >= comm->world) return ncclInvalidArgument;
    group_ops.push_back(Op{send, peer, src, dst, count * width(dt), comm});
    return group_depth ? ncclSuccess : run_ops(group_ops);
}

}  // namespace

extern "C" {

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclSystemError: return "unhandled system error (rccl_sim: a wait ran into its deadline, or shared memory failed)";
        case ncclInvalidArgument: return "invalid argument";
        case ncclInvalidUsage: return "invalid usage (rccl_sim: a receive's byte count differs from its send's)";
        default: return "internal error";
    }
}

const char *ncclSimLastGroupLog(void) { return last_log.c_str(); }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0, sizeof id->internal);
    static std::atomic<unsigned> serial{0};
    timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof id->internal, "/smr_rccl_sim.%d.%u.%lx", (int)getpid(), serial++, (unsigned long)t.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    if (id.internal[0] != '/') return ncclInvalidArgument;     // not an id this stand-in made
    ncclSimComm *c = new ncclSimComm();
    c->name = id.internal; c->rank = rank; c->world = nranks;
    int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);  // whoever comes first creates it; a fresh segment is all zeros
    if (fd < 0 || ftruncate(fd, sizeof(Control)) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    void *p = mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->ctl = (Control *)p;
    c->ctl->joined.fetch_add(1);
    const double dl = deadline_s();
    while (c->ctl->joined.load() < (uint32_t)nranks) {          // blocks until every rank of the world has called
        if (now_s() > dl) { fprintf(stderr, "rccl_sim: rank %d: only %u of %d ranks joined\n", rank, c->ctl->joined.load(), nranks); ncclCommDestroy(c); return ncclSystemError; }
        nap();
    }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (c->ctl) {
        const uint32_t gone = c->ctl->left.fetch_add(1) + 1;
        const bool last = gone >= c->ctl->joined.load();
        munmap(c->ctl, sizeof(Control));
        if (last) shm_unlink(c->name.c_str());
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { group_depth++; return ncclSuccess; }

ncclResult_t ncclGroupEnd(void) {
    if (group_depth <= 0) return ncclInvalidUsage;
    if (--group_depth) return ncclSuccess;
    return run_ops(group_ops);
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t) {
    return enqueue(true, buf, nullptr, count, dt, peer, comm);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t) {
    return enqueue(false, nullptr, buf, count, dt, peer, comm);
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t) {
    if (!c || (count && (!send || !recv))) return ncclInvalidArgument;
    if (dt != ncclUint64 && dt != ncclInt64) return ncclInvalidArgument;   // what comm.hip reduces
    if (op != ncclSum && op != ncclMax) return ncclInvalidArgument;
    std::vector<uint64_t> mine((const uint64_t *)send, (const uint64_t *)send + count), acc = mine, in(count);
    ncclResult_t r = ncclGroupStart();
    std::vector<std::vector<uint64_t>> got(c->world, std::vector<uint64_t>(count));
    for (int k = 0; k < c->world && r == ncclSuccess; k++)
        if (k != c->rank) r = ncclRecv(got[k].data(), count, dt, k, c, nullptr);
    for (int k = 0; k < c->world && r == ncclSuccess; k++)
        if (k != c->rank) r = ncclSend(mine.data(), count, dt, k, c, nullptr);
    const ncclResult_t e = ncclGroupEnd();
    if (r != ncclSuccess) return r;
    if (e != ncclSuccess) return e;
    for (int k = 0; k < c->world; k++) {
        if (k == c->rank) continue;
        for (size_t i = 0; i < count; i++) {
            if (op == ncclSum) acc[i] += got[k][i];
            else if (dt == ncclInt64) acc[i] = (uint64_t)((int64_t)got[k][i] > (int64_t)acc[i] ? got[k][i] : acc[i]);
            else acc[i] = got[k][i] > acc[i] ? got[k][i] : acc[i];
        }
    }
    memcpy(recv, acc.data(), count * sizeof(uint64_t));
    return ncclSuccess;
}

}  // extern "C"

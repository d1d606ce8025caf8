// TEST INFRASTRUCTURE ONLY: the smr_comm_* entry points of the emulator build (tests/hostsim).  The shipped library issues
// RCCL sends / receives (summerset_amd/csrc/comm.hip); the emulator has no RCCL and no second device, so this stands in
// with the one case a single process can hold -- a world of ONE rank, whose only segment is the one a rank keeps for
// itself -- so that the C-ABI paths above it (smr_mp_spread_bind_comm / smr_mp_spread_tick, argument checks) run in
// the CPU suite.  Multi-rank jobs on the emulator go through torch.distributed (gloo), as before.
#include <stdint.h>
#include <string.h>

#include <string>

#include "../../include/summerset_hip.h"

namespace smr { void set_error(const std::string &msg); }

struct smr_comm { uint32_t rank = 0, world = 1; uint64_t exchanges = 0; };

static int bad(int code, const char *msg) { smr::set_error(msg); return code; }

extern "C" {

int smr_comm_unique_id(uint8_t *out, uint64_t cap) {
    if (!out || cap < SMR_COMM_ID_BYTES) return bad(SMR_ERR_ARG, "comm: the id buffer needs SMR_COMM_ID_BYTES bytes");
    memset(out, 0x5A, SMR_COMM_ID_BYTES);
    return SMR_OK;
}

int smr_comm_init_rank(const uint8_t *id, uint64_t id_bytes, uint32_t rank, uint32_t world, smr_comm **out) {
    if (!id || !out) return bad(SMR_ERR_ARG, "comm: null argument");
    if (id_bytes != SMR_COMM_ID_BYTES) return bad(SMR_ERR_ARG, "comm: the id is SMR_COMM_ID_BYTES bytes");
    if (world == 0 || rank >= world) return bad(SMR_ERR_ARG, "comm: rank must be below world");
    if (world != 1) return bad(SMR_ERR_DEVICE, "comm: the emulator build holds a world of one rank (no RCCL)");
    *out = new smr_comm();
    return SMR_OK;
}

void smr_comm_destroy(smr_comm *c) { delete c; }

int smr_comm_exchange(smr_comm *c, const void *send_dev, const uint64_t *send_bytes, void *recv_dev, const uint64_t *recv_bytes,
                      uint32_t flags, void *stream) {
    (void)stream;
    if (!c || !send_bytes || !recv_bytes) return bad(SMR_ERR_ARG, "comm: null argument");
    if (flags & ~(uint32_t)SMR_COMM_SELF_VIA_RCCL) return bad(SMR_ERR_ARG, "comm: unknown flag");
    if ((send_bytes[0] && !send_dev) || (recv_bytes[0] && !recv_dev)) return bad(SMR_ERR_ARG, "comm: bytes to move but no buffer");
    if (send_bytes[0] != recv_bytes[0]) return bad(SMR_ERR_ARG, "comm: a rank's segment for itself must be as long as the one it expects from itself");
    if (send_bytes[0]) memmove(recv_dev, send_dev, send_bytes[0]);
    c->exchanges++;
    return SMR_OK;
}

int smr_comm_all_reduce_u64(smr_comm *c, uint64_t *inout_dev, uint64_t n, int op, void *stream) {
    (void)stream;
    if (!c || (n && !inout_dev)) return bad(SMR_ERR_ARG, "comm: null argument");
    if (op != SMR_COMM_SUM && op != SMR_COMM_MAX) return bad(SMR_ERR_ARG, "comm: op is SMR_COMM_SUM or SMR_COMM_MAX");
    return SMR_OK;
}

int smr_comm_info(smr_comm *c, uint64_t out[5]) {
    if (!c || !out) return bad(SMR_ERR_ARG, "comm: null argument");
    out[0] = c->rank; out[1] = c->world; out[2] = c->exchanges; out[3] = 0; out[4] = 0;
    return SMR_OK;
}

}  // extern "C"

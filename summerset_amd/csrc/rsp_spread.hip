// Layout L2 of the RSPaxos replica engine INSIDE the library (round 6, VERDICT r5 missing #3): BASELINE config 4 as it is written --
// "RSPaxos, 16 384 groups x 5 replicas, 4 KiB values, RS(3,2) GF(2^8) encode, 1 -> 8 GPU shard over xGMI" -- the steady state of
// summerset_amd/rsp_cluster.SteadyLoop with the replicas of every group on DIFFERENT ranks, behind one C call per tick.
// Rounds 3-5 drove it from Python (summerset_amd/spread_rsp.py: every handler a ctypes call, every header copy a torch op).
//
// The job's groups are block-partitioned over `world` ranks; replica r of block b lives on rank (b + r) mod world, so block b is
// led (replica 0, prepared) from rank b and each of its Accepts -- header AND the follower's shard of the batch's codeword, the
// one exchange of the path with real bytes (rspaxos/request.rs:127-142: one shard per peer; messages.rs:343-403) -- crosses to
// another rank, and each AcceptReply crosses back (server/transport.rs:208-275 `send_msg`).  Per tick and rank:
//   segment 0  for the block it leads: from_data + RS encode of the tick's batches with every follower's shard written STRAIGHT into
//              that follower's slice of the send buffer (smr_rs_from_data_encode_scatter), handle_req_batch, the Accept header in
//              front of every shard (flags behind that follower's loss mask)
//   exchange 0 (accept)
//   segment 1  for every (block, follower) it holds: handle_msg_accept on the receive buffer's words with the mask of the one shard it
//              was sent; the reply ballot is written by the kernel straight into the backward send buffer
//   exchange 1 (accept_reply)
//   segment 2  the leader's handle_msg_accept_reply tally (majority + f, the shard-availability gate behind it)
//   on a heartbeat tick: segment 3 (the leader's Heartbeat out), exchange 2, segment 4 (heard_heartbeat + the Heartbeats back),
//   exchange 3, segment 5 (the leader hears them).
// A message is a slot of the exchange's send buffer at its sender and of the receive buffer at its receiver.
#include <string.h>

#include <algorithm>
#include <vector>

#include "smr_common.h"

namespace smr {

constexpr int RS_MAXR = SMR_MAX_REPLICAS;
constexpr uint32_t RSS_OPS = 64;
// kind 0: dst[i] = drop && drop[i] ? 0 : src[i] (bytes);  1: dst64[i] = drop[i] ? 0 : dst64[i];  2: dst8[i] = src32[i] != 0;
// 3: dst8[i] = src64[i] != 0;  4: dst8[i] = dst8[i] & src[i]
struct RssOp { const uint8_t *src; uint8_t *dst; const uint8_t *drop; uint32_t n, kind; };
struct RssOps { uint32_t n; RssOp op[RSS_OPS]; };

__global__ __launch_bounds__(256) void rss_ops_kernel(const RssOps A) {
    const RssOp &o = A.op[blockIdx.y];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= o.n) return;
    if (o.kind == 0) o.dst[i] = (o.drop && o.drop[i]) ? (uint8_t)0 : o.src[i];
    else if (o.kind == 1) { if (o.drop[i]) ((uint64_t *)o.dst)[i] = 0; }
    else if (o.kind == 2) o.dst[i] = ((const uint32_t *)o.src)[i] != 0;
    else if (o.kind == 3) o.dst[i] = ((const uint64_t *)o.src)[i] != 0;
    else o.dst[i] = o.dst[i] & o.src[i];
}

static inline uint64_t a16(uint64_t n) { return (n + 15) / 16 * 16; }
enum { RSS_ACCEPT = 0, RSS_ACCEPT_REPLY = 1, RSS_HB = 2, RSS_HB_BACK = 3 };
static inline uint64_t rss_msg_bytes(int kind, uint64_t G, uint64_t sl) {
    return kind == RSS_ACCEPT ? a16(G * 17) + a16(G * sl) : kind == RSS_ACCEPT_REPLY ? a16(G * 8) : kind == RSS_HB ? a16(G * 20) : a16(G * 21);
}

}  // namespace smr

using namespace smr;

struct RssPlan {
    std::vector<uint64_t> in_split, out_split;
    uint64_t n_send = 0, n_recv = 0;
    uint8_t *sbuf = nullptr, *rbuf = nullptr;
    // [b * R + q]: the message between block b's leader and its follower q -- where its sender on this rank writes it (sslot), where
    // its receiver on this rank reads it (rslot).  A message between two replicas of this rank goes through the exchange like any
    // other (the rank's own segment: a device copy), as in summerset_amd/spread_rsp.py's plans.
    std::vector<uint8_t *> sslot, rslot;
};
struct RssRep {
    smr_rsp_replica *e = nullptr;
    uint32_t b = 0, r = 0, G = 0;
    uint8_t *mask = nullptr;                        // u8 [G] = 1 << r: the one shard this follower is sent
    uint32_t *r_slot = nullptr;
    // the leader's
    uint8_t *cw = nullptr;                          // [G][R * sl] the codeword buffer of the fused encode
    uint32_t *a_n = nullptr, *a_slot = nullptr, *a_val = nullptr;   // [G], [W][G], [W][G]
    uint64_t *a_ballot = nullptr, *st_ballot = nullptr;             // [G], [R][G]
    uint8_t *live = nullptr, *st_flags = nullptr, *ones = nullptr, *hb_reply = nullptr;
    uint64_t *hb_ballot = nullptr;
    uint32_t *hb_c = nullptr;                       // [3][G] scratch of the Heartbeats the leader "sends back" (dropped)
};
struct smr_rsp_spread {
    uint32_t world = 0, rank = 0, R = 0, W = 0, d = 0;
    uint64_t L = 0, sl = 0;
    std::vector<uint32_t> block_groups;
    std::vector<RssRep> reps;
    std::vector<int> rep_of;
    RssPlan plans[4];
    std::vector<uint8_t *> peer_c;                  // [b * R + r]: u8 [G_b] = r
    char *arena = nullptr;
    smr_comm *comm = nullptr;
    uint32_t next_seg = 0;
    int open_heartbeat = 0;
    uint64_t bytes_sent = 0;
    RssOps ops;
};

namespace smr {
static inline uint32_t rss_home(uint32_t b, uint32_t r, uint32_t world) { return (b + r) % world; }
static int rss_flush(smr_rsp_spread *s, hipStream_t st) {
    if (!s->ops.n) return SMR_OK;
    uint32_t most = 0;
    for (uint32_t i = 0; i < s->ops.n; i++) most = std::max(most, s->ops.op[i].n);
    hipLaunchKernelGGL(rss_ops_kernel, dim3((most + 255u) / 256u, s->ops.n), dim3(256), 0, st, s->ops);
    s->ops.n = 0;
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}
static int rss_op(smr_rsp_spread *s, hipStream_t st, uint32_t kind, const void *src, void *dst, uint64_t n, const uint8_t *drop = nullptr) {
    if (!n) return SMR_OK;
    if (s->ops.n == RSS_OPS) { int rc = rss_flush(s, st); if (rc) return rc; }
    s->ops.op[s->ops.n++] = RssOp{(const uint8_t *)src, (uint8_t *)dst, drop, (uint32_t)n, kind};
    return SMR_OK;
}
struct RssAccept { uint64_t *ballot; uint32_t *slot, *val; uint8_t *flags, *shard; };
static inline RssAccept rss_accept(uint8_t *p, uint64_t G) { return RssAccept{(uint64_t *)p, (uint32_t *)(p + 8 * G), (uint32_t *)(p + 12 * G), p + 16 * G, p + a16(G * 17)}; }
struct RssHb { uint64_t *ballot; uint32_t *commit, *exec, *snap; uint8_t *reply; };
static inline RssHb rss_hb(uint8_t *p, uint64_t G) { return RssHb{(uint64_t *)p, (uint32_t *)(p + 8 * G), (uint32_t *)(p + 12 * G), (uint32_t *)(p + 16 * G), p + 20 * G}; }
}  // namespace smr

extern "C" {

int smr_rsp_spread_create(smr_rsp_replica *const *reps, const uint32_t *rep_block, const uint8_t *rep_id, uint32_t n_reps,
                          const uint32_t *block_groups, uint32_t world, uint32_t rank, uint8_t population, uint32_t window, uint64_t data_len,
                          smr_rsp_spread **out) {
    if (!out || !block_groups || (n_reps && (!reps || !rep_block || !rep_id))) return fail(SMR_ERR_ARG, "rspaxos spread: null argument");
    if (world == 0 || rank >= world) return fail(SMR_ERR_ARG, "rspaxos spread: rank / world");
    if (population < 3 || population > RS_MAXR) return fail(SMR_ERR_ARG, "rspaxos spread: population must be in 3..8");
    if (data_len == 0 || window == 0) return fail(SMR_ERR_ARG, "rspaxos spread: data_len / window is zero");
    smr_rsp_spread *s = new smr_rsp_spread();
    s->world = world; s->rank = rank; s->R = population; s->W = window; s->L = data_len;
    s->d = population / 2 + 1;
    s->sl = smr_rs_shard_len(data_len, (int)s->d);
    s->block_groups.assign(block_groups, block_groups + world);
    s->rep_of.assign((size_t)world * population, -1);
    const uint64_t R = population, sl = s->sl;
    for (uint32_t i = 0; i < n_reps; i++) {
        const uint32_t b = rep_block[i], r = rep_id[i];
        if (!reps[i] || b >= world || r >= population || rss_home(b, r, world) != rank || !block_groups[b] || s->rep_of[(size_t)b * R + r] >= 0) {
            delete s;
            return fail(SMR_ERR_ARG, "rspaxos spread: replica r of block b lives on rank (b + r) mod world, once, and only where the block has groups");
        }
        RssRep x;
        x.e = reps[i]; x.b = b; x.r = r; x.G = block_groups[b];
        s->rep_of[(size_t)b * R + r] = (int)s->reps.size();
        s->reps.push_back(x);
    }
    for (uint32_t b = 0; b < world; b++)
        for (uint32_t r = 0; r < population; r++)
            if (block_groups[b] && rss_home(b, r, world) == rank && s->rep_of[(size_t)b * R + r] < 0) {
                delete s;
                return fail(SMR_ERR_ARG, "rspaxos spread: a replica that lives on this rank was not handed over");
            }
    size_t bytes = 0;
    auto take = [&](size_t n) { size_t o = bytes; bytes = (bytes + n + 255) & ~(size_t)255; return o; };
    struct Off { size_t sbuf, rbuf, lbuf; };
    Off poff[4];
    std::vector<std::pair<size_t, uint64_t>> psend[4], precv[4];            // (b * R + q, offset)
    for (int k = 0; k < 4; k++) {
        RssPlan &p = s->plans[k];
        p.in_split.assign(world, 0); p.out_split.assign(world, 0);
        struct M { uint32_t src, dst, b, q; };
        std::vector<M> msgs;
        for (uint32_t b = 0; b < world; b++) {
            if (!block_groups[b]) continue;
            const uint32_t hl = rss_home(b, 0, world);
            for (uint32_t q = 1; q < population; q++) {
                const uint32_t hq = rss_home(b, q, world);
                msgs.push_back((k == RSS_ACCEPT || k == RSS_HB) ? M{hl, hq, b, q} : M{hq, hl, b, q});
            }
        }
        std::vector<M> send, recv;
        for (const M &m : msgs) {
            if (m.src == rank) send.push_back(m);
            if (m.dst == rank) recv.push_back(m);
        }
        std::stable_sort(send.begin(), send.end(), [](const M &x, const M &y) { return x.dst < y.dst; });
        std::stable_sort(recv.begin(), recv.end(), [](const M &x, const M &y) { return x.src < y.src; });
        for (const M &m : send) { const uint64_t n = rss_msg_bytes(k, block_groups[m.b], sl); psend[k].push_back({(size_t)m.b * R + m.q, p.n_send}); p.n_send += n; p.in_split[m.dst] += n; }
        for (const M &m : recv) { const uint64_t n = rss_msg_bytes(k, block_groups[m.b], sl); precv[k].push_back({(size_t)m.b * R + m.q, p.n_recv}); p.n_recv += n; p.out_split[m.src] += n; }
        poff[k] = Off{take(std::max<uint64_t>(p.n_send, 16)), take(std::max<uint64_t>(p.n_recv, 16)), 0};
    }
    struct RepOff { size_t mask, r_slot, cw, a_n, a_slot, a_val, a_ballot, st_ballot, live, st_flags, ones, hb_reply, hb_ballot, hb_c; };
    std::vector<RepOff> roff(s->reps.size());
    const size_t W = window;
    for (size_t i = 0; i < s->reps.size(); i++) {
        const size_t G = s->reps[i].G;
        const bool lead = s->reps[i].r == 0;
        roff[i] = RepOff{take(G), take(4 * G), lead ? take(G * a16(R * sl)) : 0, lead ? take(4 * G) : 0, lead ? take(4 * W * G) : 0, lead ? take(4 * W * G) : 0,
                         lead ? take(8 * G) : 0, lead ? take(8 * R * G) : 0, lead ? take(G) : 0, lead ? take(R * G) : 0, take(G), lead ? take(G) : 0,
                         lead ? take(8 * G) : 0, lead ? take(12 * G) : 0};
    }
    std::vector<size_t> pc_off((size_t)world * R, 0);
    for (uint32_t b = 0; b < world; b++) {
        bool here = false;
        for (uint32_t r = 0; r < population; r++) here = here || s->rep_of[(size_t)b * R + r] >= 0;
        if (!here) continue;
        for (uint32_t r = 0; r < population; r++) pc_off[(size_t)b * R + r] = take(block_groups[b]) + 1;      // (+ 1: 0 = none)
    }
    if (hipMalloc((void **)&s->arena, bytes + 256) != hipSuccess) { delete s; return fail(SMR_ERR_DEVICE, "rspaxos spread: hipMalloc failed"); }
    hipError_t err = hipMemset(s->arena, 0, bytes + 256);
    for (int k = 0; k < 4; k++) {
        RssPlan &p = s->plans[k];
        p.sbuf = (uint8_t *)s->arena + poff[k].sbuf; p.rbuf = (uint8_t *)s->arena + poff[k].rbuf;
        p.sslot.assign((size_t)world * R, nullptr); p.rslot.assign((size_t)world * R, nullptr);
        for (auto &x : psend[k]) p.sslot[x.first] = p.sbuf + x.second;
        for (auto &x : precv[k]) p.rslot[x.first] = p.rbuf + x.second;
    }
    for (size_t i = 0; i < s->reps.size() && err == hipSuccess; i++) {
        RssRep &x = s->reps[i];
        char *a = s->arena;
        const RepOff &o = roff[i];
        x.mask = (uint8_t *)a + o.mask; x.r_slot = (uint32_t *)(a + o.r_slot); x.ones = (uint8_t *)a + o.ones;
        err = hipMemset(x.mask, (int)(1u << x.r), x.G);
        if (err == hipSuccess) err = hipMemset(x.ones, 1, x.G);
        if (x.r == 0) {
            x.cw = (uint8_t *)a + o.cw; x.a_n = (uint32_t *)(a + o.a_n); x.a_slot = (uint32_t *)(a + o.a_slot); x.a_val = (uint32_t *)(a + o.a_val);
            x.a_ballot = (uint64_t *)(a + o.a_ballot); x.st_ballot = (uint64_t *)(a + o.st_ballot); x.live = (uint8_t *)a + o.live;
            x.st_flags = (uint8_t *)a + o.st_flags; x.hb_reply = (uint8_t *)a + o.hb_reply; x.hb_ballot = (uint64_t *)(a + o.hb_ballot);
            x.hb_c = (uint32_t *)(a + o.hb_c);
        }
    }
    s->peer_c.assign((size_t)world * R, nullptr);
    for (size_t k = 0; k < pc_off.size() && err == hipSuccess; k++)
        if (pc_off[k]) {
            s->peer_c[k] = (uint8_t *)s->arena + pc_off[k] - 1;
            err = hipMemset(s->peer_c[k], (int)(k % R), block_groups[k / R]);
        }
    if (err == hipSuccess) err = hipDeviceSynchronize();
    if (err != hipSuccess) { (void)hipFree(s->arena); delete s; return fail(SMR_ERR_DEVICE, std::string("rspaxos spread: init: ") + hipGetErrorString(err)); }
    s->ops.n = 0;
    *out = s;
    return SMR_OK;
}

void smr_rsp_spread_destroy(smr_rsp_spread *s) {
    if (!s) return;
    if (s->arena) (void)hipFree(s->arena);
    delete s;
}

int smr_rsp_spread_buffers(smr_rsp_spread *s, uint32_t exchange, void **send_dev, uint64_t *send_bytes, void **recv_dev, uint64_t *recv_bytes) {
    if (!s || exchange >= 4 || !send_dev || !send_bytes || !recv_dev || !recv_bytes) return fail(SMR_ERR_ARG, "rspaxos spread: bad argument");
    const RssPlan &p = s->plans[exchange];
    *send_dev = p.sbuf; *recv_dev = p.rbuf;
    for (uint32_t k = 0; k < s->world; k++) { send_bytes[k] = p.in_split[k]; recv_bytes[k] = p.out_split[k]; }
    return SMR_OK;
}

int smr_rsp_spread_bind_comm(smr_rsp_spread *s, smr_comm *comm) {
    if (!s) return fail(SMR_ERR_ARG, "rspaxos spread: null argument");
    if (comm) {
        uint64_t info[5];
        int rc = smr_comm_info(comm, info);
        if (rc != SMR_OK) return rc;
        if (info[0] != s->rank || info[1] != s->world) return fail(SMR_ERR_ARG, "rspaxos spread: the communicator's rank / world are not the job's");
    }
    s->comm = comm;
    return SMR_OK;
}

// segment 0 .. 2 (5 on a heartbeat tick), see the file's header.  data_dev / val_dev: the led block's batches (u8 [G][data_len], rows
// data_len apart) and their tokens (u32 [G], SMR_RSP_NULL: none) -- NULL on a rank that leads no block; lost_dev (may be NULL):
// [world * 4 * R] pointers, entry (b * 4 + k) * R + q (may be NULL) = u8 [G_b], 1 where block b's message is lost: k = 0 Accept
// leader -> q, 1 AcceptReply q -> leader, 2 Heartbeat leader -> q, 3 Heartbeat q -> leader; committed_dev (u8 [G] of the led block):
// written by segment 2.
int smr_rsp_spread_segment(smr_rsp_spread *s, uint32_t seg, const uint8_t *data_dev, const uint32_t *val_dev, const uint8_t *const *lost_dev,
                           int heartbeat, uint8_t *committed_dev, void *stream) {
    if (!s) return fail(SMR_ERR_ARG, "rspaxos spread: null argument");
    const uint32_t last = heartbeat ? 5u : 2u;
    if (seg > last) return fail(SMR_ERR_ARG, "rspaxos spread: no such segment");
    if (seg != s->next_seg) return fail(SMR_ERR_STATE, "rspaxos spread: segment " + std::to_string(seg) + " out of order (the open tick expects " + std::to_string(s->next_seg) + ")");
    if (seg == 0) s->open_heartbeat = heartbeat ? 1 : 0;
    else if ((heartbeat ? 1 : 0) != s->open_heartbeat) return fail(SMR_ERR_STATE, "rspaxos spread: `heartbeat` differs from the one the tick's segment 0 was called with");
    const int led = s->rep_of[(size_t)s->rank * s->R + 0] >= 0 && rss_home(s->rank, 0, s->world) == s->rank ? s->rep_of[(size_t)s->rank * s->R + 0] : -1;
    if (led >= 0 && (!data_dev || !val_dev || !committed_dev)) return fail(SMR_ERR_ARG, "rspaxos spread: this rank leads a block: its batches, tokens and the committed array");
    s->next_seg = seg == last ? 0 : seg + 1;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t R = s->R;
    const uint64_t sl = s->sl;
    int rc;
    s->ops.n = 0;
    auto lost = [&](uint32_t b, uint32_t k, uint32_t q) -> const uint8_t * { return lost_dev ? lost_dev[((size_t)b * 4 + k) * R + q] : nullptr; };
    if (seg == 0) {                                          // ---- the leader: encode + handle_req_batch + the Accepts
        if (led >= 0) {
            RssRep &x = s->reps[led];
            const size_t G = x.G;
            RssPlan &p = s->plans[RSS_ACCEPT];
            uint8_t *dst[RS_MAXR] = {};
            for (uint32_t q = 1; q < R; q++) dst[q] = rss_accept(p.sslot[(size_t)x.b * R + q], G).shard;
            if ((rc = smr_rs_from_data_encode_scatter(data_dev, s->L, s->L, G, (int)s->d, (int)(R - s->d), x.cw, a16(R * sl), dst, sl, stream)) != SMR_OK) return rc;
            const smr_rsp_accepts acc{x.a_n, x.a_slot, x.a_val, x.a_ballot};
            if ((rc = smr_rsp_req_batch(x.e, val_dev, &acc, stream)) != SMR_OK) return rc;
            if ((rc = rss_op(s, st, 2, x.a_n, x.live, G))) return rc;
            if ((rc = rss_flush(s, st))) return rc;                                   // (live first: the flags below read it)
            for (uint32_t q = 1; q < R; q++) {
                const RssAccept m = rss_accept(p.sslot[(size_t)x.b * R + q], G);
                if ((rc = rss_op(s, st, 0, x.a_ballot, m.ballot, 8 * G)) || (rc = rss_op(s, st, 0, x.a_slot, m.slot, 4 * G)) || (rc = rss_op(s, st, 0, x.a_val, m.val, 4 * G)) ||
                    (rc = rss_op(s, st, 0, x.live, m.flags, G, lost(x.b, 0, q))))
                    return rc;
            }
            if ((rc = rss_flush(s, st))) return rc;
        }
        s->bytes_sent += s->plans[RSS_ACCEPT].n_send;
    } else if (seg == 1) {                                   // ---- followers: handle_msg_accept; the reply ballot into the backward buffer
        for (RssRep &x : s->reps) {
            if (x.r == 0) continue;
            const size_t G = x.G;
            const RssAccept m = rss_accept(s->plans[RSS_ACCEPT].rslot[(size_t)x.b * R + x.r], G);
            uint64_t *r_ballot = (uint64_t *)s->plans[RSS_ACCEPT_REPLY].sslot[(size_t)x.b * R + x.r];
            if ((rc = smr_rsp_handle_accept(x.e, m.flags, s->peer_c[(size_t)x.b * R + 0], m.slot, m.ballot, m.val, x.mask, r_ballot, x.r_slot, stream)) != SMR_OK) return rc;
            const uint8_t *g = lost(x.b, 1, x.r);
            if (g && (rc = rss_op(s, st, 1, nullptr, r_ballot, G, g))) return rc;    // a lost reply
        }
        if ((rc = rss_flush(s, st))) return rc;
        s->bytes_sent += s->plans[RSS_ACCEPT_REPLY].n_send;
    } else if (seg == 2) {                                   // ---- the leader: the AcceptReply tally
        if (led >= 0) {
            RssRep &x = s->reps[led];
            const size_t G = x.G;
            for (uint32_t q = 1; q < R; q++)
                if ((rc = rss_op(s, st, 0, s->plans[RSS_ACCEPT_REPLY].rslot[(size_t)x.b * R + q], x.st_ballot + q * G, 8 * G))) return rc;
            if ((rc = rss_flush(s, st))) return rc;
            if ((rc = rss_op(s, st, 3, x.st_ballot, x.st_flags, R * G))) return rc;  // flags = ballot != 0 (row 0, my own, stays zero)
            if ((rc = rss_flush(s, st))) return rc;
            if ((rc = smr_rsp_handle_accept_replies(x.e, x.a_slot, x.st_ballot, x.st_flags, nullptr, committed_dev, stream)) != SMR_OK) return rc;
            if ((rc = rss_op(s, st, 4, x.live, committed_dev, G))) return rc;        // committed &= live
            if ((rc = rss_flush(s, st))) return rc;
        }
    } else if (seg == 3) {                                   // ---- the leader's Heartbeat to every follower
        if (led >= 0) {
            RssRep &x = s->reps[led];
            const size_t G = x.G;
            RssPlan &p = s->plans[RSS_HB];
            const RssHb first = rss_hb(p.sslot[(size_t)x.b * R + 1], G);
            const smr_rsp_heartbeat hb{nullptr, first.ballot, first.commit, first.exec, first.snap};
            if ((rc = smr_rsp_bcast_heartbeat(x.e, x.ones, &hb, stream)) != SMR_OK) return rc;
            for (uint32_t q = 2; q < R; q++)
                if ((rc = rss_op(s, st, 0, p.sslot[(size_t)x.b * R + 1], p.sslot[(size_t)x.b * R + q], 20 * G))) return rc;
            if ((rc = rss_flush(s, st))) return rc;
        }
        s->bytes_sent += s->plans[RSS_HB].n_send;
    } else if (seg == 4) {                                   // ---- followers: heard_heartbeat, their Heartbeats back
        for (RssRep &x : s->reps) {
            if (x.r == 0) continue;
            const size_t G = x.G;
            const RssHb m = rss_hb(s->plans[RSS_HB].rslot[(size_t)x.b * R + x.r], G);
            const RssHb back = rss_hb(s->plans[RSS_HB_BACK].sslot[(size_t)x.b * R + x.r], G);
            const uint8_t *g = lost(x.b, 2, x.r);
            const uint8_t *fl = x.ones;
            if (g) {                                         // (x.r_slot's bytes as the flags' scratch: ones & ~lost)
                if ((rc = rss_op(s, st, 0, x.ones, (uint8_t *)x.r_slot, G, g)) || (rc = rss_flush(s, st))) return rc;
                fl = (const uint8_t *)x.r_slot;
            }
            const smr_rsp_heartbeat in{(uint8_t *)fl, m.ballot, m.commit, m.exec, m.snap};
            const smr_rsp_heartbeat outm{nullptr, back.ballot, back.commit, back.exec, back.snap};
            if ((rc = smr_rsp_handle_heartbeat(x.e, s->peer_c[(size_t)x.b * R + 0], &in, back.reply, &outm, stream)) != SMR_OK) return rc;
            const uint8_t *gb = lost(x.b, 3, x.r);
            if (gb && (rc = rss_op(s, st, 0, back.reply, back.reply, G, gb))) return rc;
        }
        if ((rc = rss_flush(s, st))) return rc;
        s->bytes_sent += s->plans[RSS_HB_BACK].n_send;
    } else {                                                 // ---- the leader hears the followers' Heartbeats, peers ascending
        if (led >= 0) {
            RssRep &x = s->reps[led];
            const size_t G = x.G;
            for (uint32_t q = 1; q < R; q++) {
                const RssHb m = rss_hb(s->plans[RSS_HB_BACK].rslot[(size_t)x.b * R + q], G);
                const smr_rsp_heartbeat in{m.reply, m.ballot, m.commit, m.exec, m.snap};
                const smr_rsp_heartbeat outm{nullptr, x.hb_ballot, x.hb_c, x.hb_c + G, x.hb_c + 2 * G};
                if ((rc = smr_rsp_handle_heartbeat(x.e, s->peer_c[(size_t)x.b * R + q], &in, x.hb_reply, &outm, stream)) != SMR_OK) return rc;
            }
        }
    }
    return SMR_OK;
}

int smr_rsp_spread_abort_tick(smr_rsp_spread *s) {
    if (!s) return fail(SMR_ERR_ARG, "rspaxos spread: null argument");
    s->next_seg = 0;
    return SMR_OK;
}

// the whole tick: segments and exchanges back to back on `stream` (smr_rsp_spread_bind_comm first; a job of one rank needs none)
int smr_rsp_spread_tick(smr_rsp_spread *s, const uint8_t *data_dev, const uint32_t *val_dev, const uint8_t *const *lost_dev, int heartbeat,
                        uint8_t *committed_dev, void *stream) {
    if (!s) return fail(SMR_ERR_ARG, "rspaxos spread: null argument");
    if (s->world > 1 && !s->comm) return fail(SMR_ERR_STATE, "rspaxos spread: no communicator bound (smr_rsp_spread_bind_comm)");
    if (s->next_seg != 0) return fail(SMR_ERR_STATE, "rspaxos spread: a tick is open (smr_rsp_spread_abort_tick closes it)");
    static const int exch_behind[6] = {RSS_ACCEPT, RSS_ACCEPT_REPLY, -1, RSS_HB, RSS_HB_BACK, -1};
    int rc = SMR_OK;
    for (uint32_t seg = 0; seg <= (heartbeat ? 5u : 2u) && rc == SMR_OK; seg++) {
        rc = smr_rsp_spread_segment(s, seg, data_dev, val_dev, lost_dev, heartbeat, committed_dev, stream);
        if (rc == SMR_OK && exch_behind[seg] >= 0) {
            const RssPlan &p = s->plans[exch_behind[seg]];
            if (s->world > 1) rc = smr_comm_exchange(s->comm, p.sbuf, p.in_split.data(), p.rbuf, p.out_split.data(), 0u, stream);
            else if (p.n_send && hipMemcpyAsync(p.rbuf, p.sbuf, p.n_send, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
                rc = fail(SMR_ERR_DEVICE, "rspaxos spread: the one rank's own exchange (a device copy) failed");
        }
    }
    if (rc != SMR_OK) s->next_seg = 0;
    return rc;
}

int smr_rsp_spread_info(const smr_rsp_spread *s, uint64_t out[2]) {
    if (!s || !out) return fail(SMR_ERR_ARG, "rspaxos spread: null argument");
    out[0] = 4; out[1] = s->bytes_sent;
    return SMR_OK;
}

}  // extern "C"

// Layout L2 of the EPaxos cluster INSIDE the library (round 6, VERDICT r5 missing #3): BASELINE config 5 as it is written --
// "EPaxos, 65 536 groups x 5 replicas, 8 x MI355X, dependency-graph + fast-quorum kernel with RCCL all-to-all" -- with the whole
// tick behind one C call.  Rounds 3-5 drove it from Python (summerset_amd/spread_ep.py: every handler a ctypes call, every
// message packed by torch.cat, every reply stack a torch.stack; only the exchange itself was smr_comm_exchange): here the
// schedule, the message plan, the packing and the exchanges are the library's.
//
// The job's groups are block-partitioned over `world` ranks; replica r of block b lives on rank (b + r) mod world (SURVEY 8e L2,
// DESIGN 6).  EPaxos has no leader: every replica proposes, so every rank sends and receives in every exchange.  One tick is
// the closed loop of smr_ep_cluster_tick's handler-by-handler mode (epaxos/request.rs:10-108, messages.rs:10-508) cut at the
// points where a message crosses replicas -- PreAccept, PreAcceptReply, Accept, AcceptReply, CommitNotice: 5 exchanges with
// all command leaders tallying together (the co-located loop's phase-by-phase order), or 2 + 3 R with the Accept / AcceptReply /
// CommitNotice exchanges once per command leader (`ordered`: the leader-by-leader order, execution state included).  An exchange
// is ONE all-to-all with static split sizes (smr_comm_exchange: grouped ncclSend / ncclRecv pairs over xGMI), the stand-in for
// TransportHub::send_msg / bcast_msg (server/transport.rs:208-275).
//
// A message is a SLOT of a buffer -- the exchange's send buffer (to another rank), its receive buffer (from one), or a local
// buffer (both replicas live here) -- with the handler's fields as arrays [field][G], widest first, the slot padded to 8 bytes.
// A reply handler writes its reply straight into its slot; a broadcast (one proposal / decision to R - 1 acceptors, the
// PreAccept's flags behind each acceptor's drop mask) and a leader's reply stacks [peer][G] are filled by ONE copy launch per
// stage (es_copy_kernel: a list of (src, dst, bytes) in the kernel's arguments).  Nothing of a tick touches the host's memory.
#include <string.h>

#include <algorithm>
#include <vector>

#include "smr_common.h"

namespace smr {

constexpr int ES_MAXR = SMR_MAX_REPLICAS;
constexpr uint32_t ES_OPS = 96;                  // copy operations per launch (the kernel's arguments hold them)
struct EsOp { const uint8_t *src; uint8_t *dst; const uint8_t *drop; uint32_t bytes, fill; };   // drop: dst[i] = drop[i] ? 0 : src[i]; src NULL: dst[i] = fill
struct EsOps { uint32_t n; EsOp op[ES_OPS]; };

__global__ __launch_bounds__(256) void es_copy_kernel(const EsOps A) {
    const EsOp &o = A.op[blockIdx.y];
    const uint32_t i0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (i0 >= o.bytes) return;
    if (!o.drop && o.src && i0 + 4u <= o.bytes && ((((uintptr_t)o.src) | ((uintptr_t)o.dst)) & 3u) == 0) {
        *(uint32_t *)(o.dst + i0) = *(const uint32_t *)(o.src + i0);
        return;
    }
    for (uint32_t i = i0; i < i0 + 4u && i < o.bytes; i++) o.dst[i] = !o.src ? (uint8_t)o.fill : (o.drop && o.drop[i]) ? (uint8_t)0 : o.src[i];
}
// out[g] = (a[g] == va) | (b != NULL && b[g] == vb)
__global__ __launch_bounds__(256) void es_flags_eq_kernel(uint32_t G, const uint8_t *__restrict__ a, uint8_t va, const uint8_t *__restrict__ b, uint8_t vb,
                                                          uint8_t *__restrict__ out) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g < G) out[g] = (uint8_t)((a[g] == va) || (b && b[g] == vb));
}
__global__ __launch_bounds__(256) void es_fill64_kernel(uint64_t *p, uint64_t v, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

enum { ES_PRE_ACCEPT = 0, ES_PA_REPLY = 1, ES_ACCEPT = 2, ES_ACC_REPLY = 3, ES_COMMIT = 4 };
static inline bool es_to_acceptor(int kind) { return kind == ES_PRE_ACCEPT || kind == ES_ACCEPT || kind == ES_COMMIT; }
static inline uint64_t es_msg_bytes(int kind, uint64_t G, uint64_t R) {
    const uint64_t per = es_to_acceptor(kind) ? 8 + 4 * R + 4 + 1 + 1 : kind == ES_PA_REPLY ? 8 + 8 + 4 * R + 1 : 8 + 1;
    return (per * G + 7) & ~7ull;
}

}  // namespace smr

using namespace smr;

struct EsSlot { uint8_t *p = nullptr; };            // where a message lies this tick (a slot of sbuf / rbuf / lbuf)
struct EsMsg { uint32_t b, a, z; };                 // block, from replica, to replica
struct EsPlan {
    int kind = 0;
    std::vector<uint32_t> leaders;
    std::vector<uint64_t> in_split, out_split;      // bytes to / from every rank
    uint64_t n_send = 0, n_recv = 0, n_local = 0;
    uint8_t *sbuf = nullptr, *rbuf = nullptr, *lbuf = nullptr;
    // slot of the message (b, a, z) that touches this rank: [b][a][z] -> pointer (NULL: not mine)
    std::vector<uint8_t *> slot;
};
struct EsRep {                                      // a (block, replica) that lives on this rank
    smr_ep_replica *e = nullptr;
    uint32_t b = 0, r = 0, G = 0;
    uint8_t *pa_flags = nullptr, *masked = nullptr, *slow = nullptr, *acc = nullptr;
    uint32_t *pa_col = nullptr, *pa_deps = nullptr;
    uint64_t *pa_seq = nullptr;
    uint8_t *st_flags = nullptr, *a_flags = nullptr;       // [R][G] stacked replies (my own row stays zero / None)
    uint64_t *st_ballot = nullptr, *st_seq = nullptr, *a_ballot = nullptr;
    uint32_t *st_deps = nullptr;                           // [R][R][G]
};
struct smr_ep_spread {
    uint32_t world = 0, rank = 0, R = 0, ordered = 0;
    std::vector<uint32_t> block_groups;
    std::vector<EsRep> reps;
    std::vector<int> rep_of;                        // [b * R + r] -> index into reps, -1
    std::vector<EsPlan> plans;                      // in exchange order
    std::vector<uint8_t *> peer_c;                  // [b * R + s]: u8 [G_b] = s      (blocks with a replica here)
    std::vector<uint64_t *> bal_c;                  // [b * R + s]: u64 [G_b] = s + 1
    char *arena = nullptr;
    smr_comm *comm = nullptr;
    uint32_t next_seg = 0;
    uint64_t bytes_sent = 0;
    EsOps ops;
};

namespace smr {
static inline uint32_t es_home(uint32_t b, uint32_t r, uint32_t world) { return (b + r) % world; }
static inline size_t es_ix(const smr_ep_spread *s, uint32_t b, uint32_t a, uint32_t z) { return ((size_t)b * s->R + a) * s->R + z; }

// the message list of one exchange, in the order every rank derives it in (spread_ep.py _plan): blocks, then leaders, then peers
static void es_plan_msgs(const smr_ep_spread *s, const EsPlan &p, std::vector<EsMsg> &out) {
    for (uint32_t b = 0; b < s->world; b++) {
        if (!s->block_groups[b]) continue;
        for (uint32_t ld : p.leaders)
            for (uint32_t q = 0; q < s->R; q++)
                if (q != ld) out.push_back(es_to_acceptor(p.kind) ? EsMsg{b, ld, q} : EsMsg{b, q, ld});
    }
}

static int es_flush(smr_ep_spread *s, hipStream_t st) {
    if (!s->ops.n) return SMR_OK;
    uint32_t most = 0;
    for (uint32_t i = 0; i < s->ops.n; i++) most = std::max(most, s->ops.op[i].bytes);
    hipLaunchKernelGGL(es_copy_kernel, dim3((most + 1023u) / 1024u, s->ops.n), dim3(256), 0, st, s->ops);
    s->ops.n = 0;
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}
static int es_copy(smr_ep_spread *s, hipStream_t st, const void *src, void *dst, uint64_t bytes, const uint8_t *drop = nullptr) {
    if (!bytes) return SMR_OK;
    if (s->ops.n == ES_OPS) { int rc = es_flush(s, st); if (rc) return rc; }
    s->ops.op[s->ops.n++] = EsOp{(const uint8_t *)src, (uint8_t *)dst, drop, (uint32_t)bytes, 0u};
    return SMR_OK;
}
// the fields of a slot
struct EsToAcc { uint64_t *seq; uint32_t *deps, *col; uint8_t *flags, *key; };
static inline EsToAcc es_to_acc(uint8_t *p, uint64_t G, uint64_t R) {
    return EsToAcc{(uint64_t *)p, (uint32_t *)(p + 8 * G), (uint32_t *)(p + 8 * G + 4 * R * G), p + 8 * G + 4 * R * G + 4 * G, p + 8 * G + 4 * R * G + 5 * G};
}
struct EsPaRep { uint64_t *ballot, *seq; uint32_t *deps; uint8_t *flags; };
static inline EsPaRep es_pa_rep(uint8_t *p, uint64_t G, uint64_t R) {
    return EsPaRep{(uint64_t *)p, (uint64_t *)(p + 8 * G), (uint32_t *)(p + 16 * G), p + 16 * G + 4 * R * G};
}
struct EsAccRep { uint64_t *ballot; uint8_t *flags; };
static inline EsAccRep es_acc_rep(uint8_t *p, uint64_t G) { return EsAccRep{(uint64_t *)p, p + 8 * G}; }
}  // namespace smr

extern "C" {

int smr_ep_spread_create(smr_ep_replica *const *reps, const uint32_t *rep_block, const uint8_t *rep_id, uint32_t n_reps,
                         const uint32_t *block_groups, uint32_t world, uint32_t rank, uint8_t population, int ordered, smr_ep_spread **out) {
    if (!out || !block_groups || (n_reps && (!reps || !rep_block || !rep_id))) return fail(SMR_ERR_ARG, "epaxos spread: null argument");
    if (world == 0 || rank >= world) return fail(SMR_ERR_ARG, "epaxos spread: rank / world");
    if (population < 3 || population > ES_MAXR) return fail(SMR_ERR_ARG, "epaxos spread: population must be in 3..8");
    smr_ep_spread *s = new smr_ep_spread();
    s->world = world; s->rank = rank; s->R = population; s->ordered = ordered ? 1 : 0;
    s->block_groups.assign(block_groups, block_groups + world);
    s->rep_of.assign((size_t)world * population, -1);
    const uint64_t R = population;
    for (uint32_t i = 0; i < n_reps; i++) {
        const uint32_t b = rep_block[i], r = rep_id[i];
        if (!reps[i] || b >= world || r >= population || es_home(b, r, world) != rank || !block_groups[b] || s->rep_of[(size_t)b * R + r] >= 0) {
            delete s;
            return fail(SMR_ERR_ARG, "epaxos spread: replica r of block b lives on rank (b + r) mod world, once, and only where the block has groups");
        }
        EsRep x;
        x.e = reps[i]; x.b = b; x.r = r; x.G = block_groups[b];
        s->rep_of[(size_t)b * R + r] = (int)s->reps.size();
        s->reps.push_back(x);
    }
    for (uint32_t b = 0; b < world; b++)
        for (uint32_t r = 0; r < population; r++)
            if (block_groups[b] && es_home(b, r, world) == rank && s->rep_of[(size_t)b * R + r] < 0) {
                delete s;
                return fail(SMR_ERR_ARG, "epaxos spread: a replica that lives on this rank was not handed over");
            }
    // the exchanges: PreAccept, PreAcceptReply for all leaders; then Accept / AcceptReply / CommitNotice per leader set
    std::vector<std::vector<uint32_t>> sets;
    std::vector<uint32_t> all;
    for (uint32_t r = 0; r < population; r++) all.push_back(r);
    if (ordered) for (uint32_t r = 0; r < population; r++) sets.push_back({r});
    else sets.push_back(all);
    auto add = [&](int kind, const std::vector<uint32_t> &ld) { EsPlan p; p.kind = kind; p.leaders = ld; s->plans.push_back(p); };
    add(ES_PRE_ACCEPT, all); add(ES_PA_REPLY, all);
    for (auto &ld : sets) { add(ES_ACCEPT, ld); add(ES_ACC_REPLY, ld); add(ES_COMMIT, ld); }
    // sizes first, one arena, then the pointers
    size_t bytes = 0;
    auto take = [&](size_t n) { size_t o = bytes; bytes = (bytes + n + 255) & ~(size_t)255; return o; };
    struct Off { size_t sbuf, rbuf, lbuf; };
    std::vector<Off> poff(s->plans.size());
    std::vector<std::vector<std::pair<EsMsg, uint64_t>>> psend(s->plans.size()), precv(s->plans.size()), ploc(s->plans.size());
    for (size_t k = 0; k < s->plans.size(); k++) {
        EsPlan &p = s->plans[k];
        p.in_split.assign(world, 0); p.out_split.assign(world, 0);
        std::vector<EsMsg> msgs;
        es_plan_msgs(s, p, msgs);
        std::vector<EsMsg> send, recv;
        for (const EsMsg &m : msgs) {
            const uint32_t src = es_home(m.b, m.a, world), dst = es_home(m.b, m.z, world);
            if (src == rank && dst != rank) send.push_back(m);
            if (dst == rank && src != rank) recv.push_back(m);
            if (src == rank && dst == rank) { ploc[k].push_back({m, p.n_local}); p.n_local += es_msg_bytes(p.kind, block_groups[m.b], R); }
        }
        std::stable_sort(send.begin(), send.end(), [&](const EsMsg &x, const EsMsg &y) { return es_home(x.b, x.z, world) < es_home(y.b, y.z, world); });
        std::stable_sort(recv.begin(), recv.end(), [&](const EsMsg &x, const EsMsg &y) { return es_home(x.b, x.a, world) < es_home(y.b, y.a, world); });
        for (const EsMsg &m : send) { const uint64_t n = es_msg_bytes(p.kind, block_groups[m.b], R); psend[k].push_back({m, p.n_send}); p.n_send += n; p.in_split[es_home(m.b, m.z, world)] += n; }
        for (const EsMsg &m : recv) { const uint64_t n = es_msg_bytes(p.kind, block_groups[m.b], R); precv[k].push_back({m, p.n_recv}); p.n_recv += n; p.out_split[es_home(m.b, m.a, world)] += n; }
        poff[k] = Off{take(std::max<uint64_t>(p.n_send, 8)), take(std::max<uint64_t>(p.n_recv, 8)), take(std::max<uint64_t>(p.n_local, 8))};
    }
    struct RepOff { size_t pa_flags, masked, slow, acc, pa_col, pa_deps, pa_seq, st_flags, a_flags, st_ballot, st_seq, a_ballot, st_deps; };
    std::vector<RepOff> roff(s->reps.size());
    for (size_t i = 0; i < s->reps.size(); i++) {
        const size_t G = s->reps[i].G;
        roff[i] = RepOff{take(G), take(G), take(G), take(G), take(4 * G), take(4 * R * G), take(8 * G), take(R * G), take(R * G), take(8 * R * G), take(8 * R * G),
                         take(8 * R * G), take(4 * R * R * G)};
    }
    std::vector<size_t> pc_off((size_t)world * R, 0), bc_off((size_t)world * R, 0);
    for (uint32_t b = 0; b < world; b++) {
        bool here = false;
        for (uint32_t r = 0; r < population; r++) here = here || s->rep_of[(size_t)b * R + r] >= 0;
        if (!here) continue;
        for (uint32_t r = 0; r < population; r++) { pc_off[(size_t)b * R + r] = take(block_groups[b]); bc_off[(size_t)b * R + r] = take(8 * (size_t)block_groups[b]); }
    }
    if (hipMalloc((void **)&s->arena, bytes + 256) != hipSuccess) { delete s; return fail(SMR_ERR_DEVICE, "epaxos spread: hipMalloc failed"); }
    if (hipMemset(s->arena, 0, bytes + 256) != hipSuccess) { (void)hipFree(s->arena); delete s; return fail(SMR_ERR_DEVICE, "epaxos spread: hipMemset failed"); }
    for (size_t k = 0; k < s->plans.size(); k++) {
        EsPlan &p = s->plans[k];
        p.sbuf = (uint8_t *)s->arena + poff[k].sbuf; p.rbuf = (uint8_t *)s->arena + poff[k].rbuf; p.lbuf = (uint8_t *)s->arena + poff[k].lbuf;
        p.slot.assign((size_t)world * R * R, nullptr);
        for (auto &x : psend[k]) p.slot[es_ix(s, x.first.b, x.first.a, x.first.z)] = p.sbuf + x.second;
        for (auto &x : precv[k]) p.slot[es_ix(s, x.first.b, x.first.a, x.first.z)] = p.rbuf + x.second;
        for (auto &x : ploc[k]) p.slot[es_ix(s, x.first.b, x.first.a, x.first.z)] = p.lbuf + x.second;
    }
    s->peer_c.assign((size_t)world * R, nullptr); s->bal_c.assign((size_t)world * R, nullptr);
    hipError_t err = hipSuccess;
    for (size_t i = 0; i < s->reps.size(); i++) {
        EsRep &x = s->reps[i];
        char *a = s->arena;
        const RepOff &o = roff[i];
        x.pa_flags = (uint8_t *)a + o.pa_flags; x.masked = (uint8_t *)a + o.masked; x.slow = (uint8_t *)a + o.slow; x.acc = (uint8_t *)a + o.acc;
        x.pa_col = (uint32_t *)(a + o.pa_col); x.pa_deps = (uint32_t *)(a + o.pa_deps); x.pa_seq = (uint64_t *)(a + o.pa_seq);
        x.st_flags = (uint8_t *)a + o.st_flags; x.a_flags = (uint8_t *)a + o.a_flags;
        x.st_ballot = (uint64_t *)(a + o.st_ballot); x.st_seq = (uint64_t *)(a + o.st_seq); x.a_ballot = (uint64_t *)(a + o.a_ballot);
        x.st_deps = (uint32_t *)(a + o.st_deps);
        if (err == hipSuccess) err = hipMemset(x.st_deps, 0xFF, 4 * R * R * (size_t)x.G);        // (my own row of the stack: no dependencies, never written)
    }
    for (uint32_t b = 0; b < world && err == hipSuccess; b++)
        for (uint32_t r = 0; r < population && err == hipSuccess; r++) {
            if (!bc_off[(size_t)b * R + r] && !pc_off[(size_t)b * R + r]) continue;
            uint8_t *pc = (uint8_t *)s->arena + pc_off[(size_t)b * R + r];
            uint64_t *bc = (uint64_t *)(s->arena + bc_off[(size_t)b * R + r]);
            s->peer_c[(size_t)b * R + r] = pc; s->bal_c[(size_t)b * R + r] = bc;
            err = hipMemset(pc, (int)r, block_groups[b]);
            if (err == hipSuccess) {
                hipLaunchKernelGGL(es_fill64_kernel, dim3((block_groups[b] + 255) / 256), dim3(256), 0, (hipStream_t)nullptr, bc, (uint64_t)(r + 1), block_groups[b]);
                err = hipGetLastError();
            }
        }
    if (err == hipSuccess) err = hipDeviceSynchronize();
    if (err != hipSuccess) { (void)hipFree(s->arena); delete s; return fail(SMR_ERR_DEVICE, std::string("epaxos spread: init: ") + hipGetErrorString(err)); }
    s->ops.n = 0;
    *out = s;
    return SMR_OK;
}

void smr_ep_spread_destroy(smr_ep_spread *s) {
    if (!s) return;
    if (s->arena) (void)hipFree(s->arena);
    delete s;
}

int smr_ep_spread_n_exchanges(const smr_ep_spread *s) { return s ? (int)s->plans.size() : SMR_ERR_ARG; }

int smr_ep_spread_buffers(smr_ep_spread *s, uint32_t exchange, void **send_dev, uint64_t *send_bytes, void **recv_dev, uint64_t *recv_bytes) {
    if (!s || exchange >= s->plans.size() || !send_dev || !send_bytes || !recv_dev || !recv_bytes) return fail(SMR_ERR_ARG, "epaxos spread: bad argument");
    const EsPlan &p = s->plans[exchange];
    *send_dev = p.sbuf; *recv_dev = p.rbuf;
    for (uint32_t k = 0; k < s->world; k++) { send_bytes[k] = p.in_split[k]; recv_bytes[k] = p.out_split[k]; }
    return SMR_OK;
}

int smr_ep_spread_bind_comm(smr_ep_spread *s, smr_comm *comm) {
    if (!s) return fail(SMR_ERR_ARG, "epaxos spread: null argument");
    if (comm) {
        uint64_t info[5];
        int rc = smr_comm_info(comm, info);
        if (rc != SMR_OK) return rc;
        if (info[0] != s->rank || info[1] != s->world) return fail(SMR_ERR_ARG, "epaxos spread: the communicator's rank / world are not the job's");
    }
    s->comm = comm;
    return SMR_OK;
}

// segment `seg` of the tick: the compute between exchange seg - 1 (whose receive buffer it reads) and exchange seg (whose send
// buffer it fills); seg == n_exchanges: behind the last one.  keys_dev / out: per replica of this rank, in smr_ep_spread_create's
// order; drop_dev (may be NULL): [n_reps * R] pointers, entry i * R + q (may be NULL) = u8 [G], 1 where replica i's PreAccept to
// q is lost (with its reply).
int smr_ep_spread_segment(smr_ep_spread *s, uint32_t seg, const uint8_t *const *keys_dev, const uint8_t *const *drop_dev,
                          const smr_ep_cluster_out *out, void *stream) {
    if (!s || (!keys_dev && !s->reps.empty()) || (!out && !s->reps.empty())) return fail(SMR_ERR_ARG, "epaxos spread: null argument");
    if (seg > s->plans.size()) return fail(SMR_ERR_ARG, "epaxos spread: no such segment");
    if (seg != s->next_seg) return fail(SMR_ERR_STATE, "epaxos spread: segment " + std::to_string(seg) + " out of order (the open tick expects " + std::to_string(s->next_seg) + ")");
    for (size_t i = 0; i < s->reps.size(); i++)
        if (!keys_dev[i] || !out[i].proposed || !out[i].col || !out[i].seq0 || !out[i].deps0 || !out[i].decision || !out[i].committed || !out[i].seq || !out[i].deps)
            return fail(SMR_ERR_ARG, "epaxos spread: null key or output array");
    s->next_seg = seg == s->plans.size() ? 0 : seg + 1;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t R = s->R;
    int rc;
    s->ops.n = 0;
    // ---- what consumes the previous exchange --------------------------------------------------------------------------
    if (seg > 0) {
        const EsPlan &p = s->plans[seg - 1];
        if (p.kind == ES_PRE_ACCEPT) {                       // acceptors: one sender's PreAccept at a time, senders ascending; the reply into its slot
            const EsPlan &pn = s->plans[seg];
            for (EsRep &x : s->reps)
                for (uint32_t sd = 0; sd < R; sd++) {
                    if (sd == x.r) continue;
                    const EsToAcc m = es_to_acc(p.slot[es_ix(s, x.b, sd, x.r)], x.G, R);
                    const EsPaRep rp = es_pa_rep(pn.slot[es_ix(s, x.b, x.r, sd)], x.G, R);
                    const smr_ep_msg msg{m.flags, s->peer_c[(size_t)x.b * R + sd], m.col, s->bal_c[(size_t)x.b * R + sd], m.seq, m.deps, m.key, nullptr};
                    const smr_ep_msg rmsg{rp.flags, nullptr, nullptr, rp.ballot, rp.seq, rp.deps, nullptr, nullptr};
                    if ((rc = smr_ep_handle_pre_accept(x.e, &msg, &rmsg, stream)) != SMR_OK) return rc;
                }
        } else if (p.kind == ES_PA_REPLY) {                  // the replies into the leaders' stacks (the decision itself: below, per leader set)
            for (EsRep &x : s->reps)
                for (uint32_t q = 0; q < R; q++) {
                    if (q == x.r) continue;
                    const EsPaRep rp = es_pa_rep(p.slot[es_ix(s, x.b, q, x.r)], x.G, R);
                    const size_t G = x.G;
                    if ((rc = es_copy(s, st, rp.flags, x.st_flags + q * G, G)) || (rc = es_copy(s, st, rp.ballot, x.st_ballot + q * G, 8 * G)) ||
                        (rc = es_copy(s, st, rp.seq, x.st_seq + q * G, 8 * G)) || (rc = es_copy(s, st, rp.deps, x.st_deps + (size_t)q * R * G, 4 * R * G)))
                        return rc;
                }
            if ((rc = es_flush(s, st))) return rc;
        } else if (p.kind == ES_ACCEPT) {                    // acceptors: the Accepts of this leader set, leaders ascending
            const EsPlan &pn = s->plans[seg];
            for (EsRep &x : s->reps)
                for (uint32_t ld : p.leaders) {
                    if (ld == x.r) continue;
                    const EsToAcc m = es_to_acc(p.slot[es_ix(s, x.b, ld, x.r)], x.G, R);
                    const EsAccRep rp = es_acc_rep(pn.slot[es_ix(s, x.b, x.r, ld)], x.G);
                    const smr_ep_msg msg{m.flags, s->peer_c[(size_t)x.b * R + ld], m.col, s->bal_c[(size_t)x.b * R + ld], m.seq, m.deps, m.key, nullptr};
                    const smr_ep_msg rmsg{rp.flags, nullptr, nullptr, rp.ballot, nullptr, nullptr, nullptr, nullptr};
                    if ((rc = smr_ep_handle_accept(x.e, &msg, &rmsg, stream)) != SMR_OK) return rc;
                }
        } else if (p.kind == ES_ACC_REPLY) {                 // command leaders: the slow-path tally, what is committed
            for (EsRep &x : s->reps) {
                if (std::find(p.leaders.begin(), p.leaders.end(), x.r) == p.leaders.end()) continue;
                const size_t G = x.G;
                for (uint32_t q = 0; q < R; q++) {
                    if (q == x.r) continue;
                    const EsAccRep rp = es_acc_rep(p.slot[es_ix(s, x.b, q, x.r)], G);
                    if ((rc = es_copy(s, st, rp.flags, x.a_flags + q * G, G)) || (rc = es_copy(s, st, rp.ballot, x.a_ballot + q * G, 8 * G))) return rc;
                }
            }
            if ((rc = es_flush(s, st))) return rc;
            for (size_t i = 0; i < s->reps.size(); i++) {
                EsRep &x = s->reps[i];
                if (std::find(p.leaders.begin(), p.leaders.end(), x.r) == p.leaders.end()) continue;
                if ((rc = smr_ep_handle_accept_replies(x.e, x.pa_col, x.a_ballot, x.a_flags, nullptr, x.acc, stream)) != SMR_OK) return rc;
                hipLaunchKernelGGL(es_flags_eq_kernel, dim3((x.G + 255) / 256), dim3(256), 0, st, x.G, out[i].decision, (uint8_t)3, x.acc, (uint8_t)1, out[i].committed);
                SMR_HIP_TRY(hipGetLastError());
            }
        } else {                                             // ES_COMMIT: acceptors take the CommitNotices
            for (EsRep &x : s->reps)
                for (uint32_t ld : p.leaders) {
                    if (ld == x.r) continue;
                    const EsToAcc m = es_to_acc(p.slot[es_ix(s, x.b, ld, x.r)], x.G, R);
                    const smr_ep_msg msg{m.flags, s->peer_c[(size_t)x.b * R + ld], m.col, s->bal_c[(size_t)x.b * R + ld], m.seq, m.deps, m.key, nullptr};
                    if ((rc = smr_ep_handle_commit_notice(x.e, &msg, stream)) != SMR_OK) return rc;
                }
        }
    }
    if (seg == s->plans.size()) return SMR_OK;
    // ---- what fills the next exchange ----------------------------------------------------------------------------------
    const EsPlan &p = s->plans[seg];
    // a broadcast of replica x: (seq, deps, col, flags, key) into the slot of every peer q
    auto bcast = [&](const EsRep &x, size_t i, const uint64_t *seq, const uint32_t *deps, const uint8_t *flags, bool with_drop) -> int {
        const size_t G = x.G;
        for (uint32_t q = 0; q < R; q++) {
            if (q == x.r) continue;
            const EsToAcc m = es_to_acc(p.slot[es_ix(s, x.b, x.r, q)], G, R);
            const uint8_t *dm = with_drop && drop_dev ? drop_dev[i * R + q] : nullptr;
            int r2;
            if ((r2 = es_copy(s, st, seq, m.seq, 8 * G)) || (r2 = es_copy(s, st, deps, m.deps, 4 * R * G)) || (r2 = es_copy(s, st, x.pa_col, m.col, 4 * G)) ||
                (r2 = es_copy(s, st, flags, m.flags, G, dm)) || (r2 = es_copy(s, st, keys_dev[i], m.key, G)))
                return r2;
        }
        return SMR_OK;
    };
    if (p.kind == ES_PRE_ACCEPT) {                           // every replica proposes; the PreAccept to every peer
        for (size_t i = 0; i < s->reps.size(); i++) {
            EsRep &x = s->reps[i];
            const smr_ep_msg pa{out[i].proposed, nullptr, x.pa_col, nullptr, out[i].seq0, out[i].deps0, nullptr, nullptr};
            if ((rc = smr_ep_propose(x.e, keys_dev[i], nullptr, &pa, stream)) != SMR_OK) return rc;
            if ((rc = es_copy(s, st, x.pa_col, out[i].col, 4 * (size_t)x.G))) return rc;
            if ((rc = bcast(x, i, out[i].seq0, out[i].deps0, out[i].proposed, true))) return rc;
        }
        if ((rc = es_flush(s, st))) return rc;
    } else if (p.kind == ES_ACCEPT) {                        // command leaders of this set: the decision; Accepts where the slow path was taken
        for (size_t i = 0; i < s->reps.size(); i++) {
            EsRep &x = s->reps[i];
            if (std::find(p.leaders.begin(), p.leaders.end(), x.r) == p.leaders.end()) continue;
            if ((rc = smr_ep_handle_pre_accept_replies(x.e, x.pa_col, x.st_ballot, x.st_seq, x.st_deps, x.st_flags, nullptr, nullptr, out[i].decision, out[i].seq,
                                                       out[i].deps, stream)) != SMR_OK) return rc;
            hipLaunchKernelGGL(es_flags_eq_kernel, dim3((x.G + 255) / 256), dim3(256), 0, st, x.G, out[i].decision, (uint8_t)2, (const uint8_t *)nullptr, (uint8_t)0, x.slow);
            SMR_HIP_TRY(hipGetLastError());
            if ((rc = bcast(x, i, out[i].seq, out[i].deps, x.slow, false))) return rc;
        }
        if ((rc = es_flush(s, st))) return rc;
    } else if (p.kind == ES_COMMIT) {                        // CommitNotice to every peer
        for (size_t i = 0; i < s->reps.size(); i++) {
            EsRep &x = s->reps[i];
            if (std::find(p.leaders.begin(), p.leaders.end(), x.r) == p.leaders.end()) continue;
            if ((rc = bcast(x, i, out[i].seq, out[i].deps, out[i].committed, false))) return rc;
        }
        if ((rc = es_flush(s, st))) return rc;
    }
    // (ES_PA_REPLY, ES_ACC_REPLY: the acceptors' handlers above wrote their replies into this exchange's slots)
    s->bytes_sent += p.n_send;
    return SMR_OK;
}

int smr_ep_spread_abort_tick(smr_ep_spread *s) {
    if (!s) return fail(SMR_ERR_ARG, "epaxos spread: null argument");
    s->next_seg = 0;
    return SMR_OK;
}

// the whole tick: segments and exchanges back to back on `stream` (smr_ep_spread_bind_comm first; world 1 needs none)
int smr_ep_spread_tick(smr_ep_spread *s, const uint8_t *const *keys_dev, const uint8_t *const *drop_dev, const smr_ep_cluster_out *out, void *stream) {
    if (!s) return fail(SMR_ERR_ARG, "epaxos spread: null argument");
    if (s->world > 1 && !s->comm) return fail(SMR_ERR_STATE, "epaxos spread: no communicator bound (smr_ep_spread_bind_comm)");
    if (s->next_seg != 0) return fail(SMR_ERR_STATE, "epaxos spread: a tick is open (smr_ep_spread_abort_tick closes it)");
    int rc = SMR_OK;
    for (uint32_t seg = 0; seg <= s->plans.size() && rc == SMR_OK; seg++) {
        rc = smr_ep_spread_segment(s, seg, keys_dev, drop_dev, out, stream);
        if (rc == SMR_OK && seg < s->plans.size() && s->world > 1) {
            const EsPlan &p = s->plans[seg];
            rc = smr_comm_exchange(s->comm, p.sbuf, p.in_split.data(), p.rbuf, p.out_split.data(), 0u, stream);
        }
    }
    if (rc != SMR_OK) s->next_seg = 0;                       // (a failed segment / exchange: the tick is closed, the replicas' state is the host's to restore)
    return rc;
}

int smr_ep_spread_info(const smr_ep_spread *s, uint64_t out[2]) {
    if (!s || !out) return fail(SMR_ERR_ARG, "epaxos spread: null argument");
    out[0] = s->plans.size(); out[1] = s->bytes_sent;
    return SMR_OK;
}

}  // extern "C"

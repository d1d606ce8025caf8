// MultiPaxos near-quorum reads (src/protocols/multipaxos/quorumread.rs, request.rs:55-101) over G groups, one
// replica per group, lane = group: the highest-slot table per key (refresh_highest_slot :8-26), the responder
// (handle_msg_read_query :75-188 with inspect_highest_slot :30-73) and the issuer's read-quorum tally
// (handle_msg_read_query_reply :190-346: max_replies merge, rq_acks bitmap, quorum_cnt, the clients' answers).
// Model: keys < K, values are 32-bit tokens (0 = none), a batch = its Put keys + one token, the log
// is the caller's (start_slot, length, status / token rings of W), a query id is a slot q < Q of the table.
// All arrays [..][G], group fastest: a wavefront reads 64 consecutive groups per row.
#include <string.h>

#include <vector>

#include "smr_common.h"

namespace smr {

constexpr uint32_t QR_NONE = 0xFFFFFFFFu;
enum { QR_ACCEPTING = 2, QR_COMMITTED = 3, QR_EXECUTED = 4 };
enum { RP_NONE = 0, RP_SLOT = 1, RP_VALUE = 2 };
enum { OUT_PENDING = 0, OUT_NOT_FOUND = 1, OUT_RETRY = 2, OUT_VALUE = 3 };

struct QrView {
    uint32_t G, K, B, Q, R, me, quorum;
    uint32_t *highest_slot;                              // [K][G]
    uint8_t *live, *n, *acks;                            // [Q][G]
    uint8_t *mx_state;                                   // [Q][B][G]
    uint32_t *mx_slot, *mx_val;                          // [Q][B][G]
    unsigned long long *counters;                        // good, retry, not found, conflicting values
};

// quorumread.rs:8-26
__global__ __launch_bounds__(256) void qr_refresh_kernel(const QrView v, const uint32_t *__restrict__ slot,
                                                         const uint8_t *__restrict__ put_keys) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    const uint32_t s = slot[g];
    if (s == QR_NONE) return;
    for (uint32_t i = 0; i < v.B; i++) {
        const uint32_t key = put_keys[(size_t)i * v.G + g];
        if (key == 0xFF || key >= v.K) continue;
        uint32_t *hp = &v.highest_slot[(size_t)key * v.G + g];
        const uint32_t h = *hp;
        if (h == QR_NONE || s > h) *hp = s;                 // insert, or max
    }
}

// quorumread.rs:75-188; the not-a-stable-leader arm is inspect_highest_slot (:30-73) per key
__global__ __launch_bounds__(256) void qr_read_query_kernel(const QrView v, const uint8_t *__restrict__ keys,
                                                            const uint8_t *__restrict__ n, const uint8_t *__restrict__ stable_leader,
                                                            const uint32_t *__restrict__ kv, const uint32_t *__restrict__ start_slot,
                                                            const uint32_t *__restrict__ log_end, const void *__restrict__ status,
                                                            const uint32_t *__restrict__ token, uint32_t Wmask, uint32_t mp_layout,
                                                            const uint32_t *__restrict__ run_lo, const uint32_t *__restrict__ run_hi,
                                                            const uint8_t *__restrict__ run_leader, uint32_t run_rep,
                                                            uint8_t *__restrict__ o_state, uint32_t *__restrict__ o_slot,
                                                            uint32_t *__restrict__ o_val, uint8_t *__restrict__ from_leader) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    const uint32_t cnt = n[g] < v.B ? n[g] : v.B;
    const bool stable = cnt && stable_leader && stable_leader[g];
    from_leader[g] = stable ? 1 : 0;
    const uint32_t start = start_slot[g], end = log_end[g];
    for (uint32_t i = 0; i < v.B; i++) {
        const size_t o = (size_t)i * v.G + g;
        uint32_t st = RP_NONE, sl = 0, vl = 0;
        if (i < cnt) {
            const uint32_t key = keys[o];
            if (stable) {
                const uint32_t x = kv[(size_t)key * v.G + g];   // :124-131 value.map(|v| (0, Some(v)))
                if (x) { st = RP_VALUE; vl = x; }
            } else {
                const uint32_t h = v.highest_slot[(size_t)key * v.G + g];
                if (h != QR_NONE) {
                    st = RP_SLOT; sl = h;
                    if (h >= start && h < end) {
                        // mp_layout: the MultiPaxos engine's own rings -- wave-tiled (mp_types.h tix) meta words, Status in the low 3 bits
                        const size_t w = mp_layout ? ((size_t)(g >> 6) * (Wmask + 1) + (h & Wmask)) * 64 + (g & 63) : (size_t)(h & Wmask) * v.G + g;
                        uint32_t sv = mp_layout ? (((const uint32_t *)status)[w] & 7u) : ((const uint8_t *)status)[w];
                        if (run_lo && h >= run_lo[g]) {
                            if (h < run_hi[g]) sv = QR_EXECUTED;                        // status implied by the bars
                            else if (run_leader && run_leader[g] != run_rep) sv = QR_ACCEPTING;   // a follower's run stores no status word
                        }
                        if (sv >= QR_COMMITTED) { st = RP_VALUE; vl = token[w]; }
                    }
                }
            }
        }
        o_state[o] = (uint8_t)st; o_slot[o] = sl; o_val[o] = vl;
    }
}

// request.rs:55-101: bookkeeping of query q
__global__ __launch_bounds__(256) void qr_issue_kernel(const QrView v, uint32_t q, const uint8_t *__restrict__ n,
                                                       const uint8_t *__restrict__ state, const uint32_t *__restrict__ slot,
                                                       const uint32_t *__restrict__ val) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    const uint32_t cnt = n[g] < v.B ? n[g] : v.B;
    if (!cnt) return;
    const size_t qo = (size_t)q * v.G + g;
    v.live[qo] = 1; v.n[qo] = (uint8_t)cnt; v.acks[qo] = (uint8_t)(1u << v.me);   // :100
    for (uint32_t i = 0; i < cnt; i++) {
        const size_t o = (size_t)i * v.G + g, m = ((size_t)q * v.B + i) * v.G + g;
        v.mx_state[m] = state[o]; v.mx_slot[m] = slot[o]; v.mx_val[m] = val[o];
    }
}

// quorumread.rs:190-346, the replies of up to R - 1 peers to query q in one launch, in `order`
__global__ __launch_bounds__(256) void qr_replies_kernel(const QrView v, uint32_t q, const uint8_t *__restrict__ state,
                                                         const uint32_t *__restrict__ slot, const uint32_t *__restrict__ val,
                                                         const uint8_t *__restrict__ flags, const uint32_t *__restrict__ order,
                                                         uint8_t *__restrict__ outcome, uint32_t *__restrict__ out_val,
                                                         uint8_t *__restrict__ done) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    unsigned int c[4] = {0, 0, 0, 0};
    if (g < v.G) {
        const size_t qo = (size_t)q * v.G + g;
        bool live = v.live[qo] != 0;
        const uint32_t cnt = live ? v.n[qo] : 0;
        uint32_t acks = v.acks[qo];
        const uint32_t o_acks = acks;
        bool answered = false;
        const uint32_t ctl = order ? order[g] : SMR_CTL_IDENTITY;
        for (uint32_t oi = 0; oi < v.R && live; oi++) {
            const uint32_t p = (ctl >> (3 * oi)) & 7u;
            if (p == v.me || p >= v.R) continue;
            const uint32_t f = flags[(size_t)p * v.G + g];
            if (!(f & 1)) continue;
            bool can_reply = false;
            if (f & 2) {                                        // :206-210 from the stable leader: take its answers
                for (uint32_t i = 0; i < cnt; i++) {
                    const size_t r = ((size_t)p * v.B + i) * v.G + g, m = ((size_t)q * v.B + i) * v.G + g;
                    v.mx_state[m] = state[r]; v.mx_slot[m] = slot[r]; v.mx_val[m] = val[r];
                }
                can_reply = true;
            } else if (!((acks >> p) & 1u)) {                   // :211
                bool err = false;
                for (uint32_t i = 0; i < cnt && !err; i++) {    // :215-253
                    const size_t r = ((size_t)p * v.B + i) * v.G + g, m = ((size_t)q * v.B + i) * v.G + g;
                    const uint32_t rs = state[r];
                    if (rs == RP_NONE) continue;
                    const uint32_t rsl = slot[r], ms = v.mx_state[m], msl = v.mx_slot[m];
                    if (rs == RP_SLOT || ms == RP_NONE) {       // incl. :231-233, where a committed value is not kept
                        if (ms == RP_NONE || (rs == RP_SLOT && rsl > msl)) { v.mx_state[m] = RP_SLOT; v.mx_slot[m] = rsl; v.mx_val[m] = 0; }
                    } else if (ms == RP_SLOT) {
                        if (rsl >= msl) { v.mx_state[m] = RP_VALUE; v.mx_slot[m] = rsl; v.mx_val[m] = val[r]; }
                    } else if (rsl > msl) {
                        v.mx_slot[m] = rsl; v.mx_val[m] = val[r];
                    } else if (rsl == msl && v.mx_val[m] != val[r]) {
                        err = true; c[3]++;                     // :243-250 logged_err, the handler returns
                    }
                }
                if (!err) {
                    acks |= 1u << p;                            // :254
                    if ((uint32_t)__popc(acks) >= v.quorum) can_reply = true;   // :258-265
                }
            }
            if (can_reply) {                                    // :270-318
                live = false; answered = true;
                for (uint32_t i = 0; i < cnt; i++) {
                    const size_t o = (size_t)i * v.G + g, m = ((size_t)q * v.B + i) * v.G + g;
                    const uint32_t ms = v.mx_state[m];
                    outcome[o] = ms == RP_NONE ? OUT_NOT_FOUND : (ms == RP_SLOT ? OUT_RETRY : OUT_VALUE);
                    out_val[o] = ms == RP_VALUE ? v.mx_val[m] : 0;
                    c[ms == RP_NONE ? 2 : (ms == RP_SLOT ? 1 : 0)]++;
                }
            }
        }
        for (uint32_t i = answered ? cnt : 0; i < v.B; i++) { outcome[(size_t)i * v.G + g] = OUT_PENDING; out_val[(size_t)i * v.G + g] = 0; }
        done[g] = answered ? 1 : 0;
        if (answered) v.live[qo] = 0;
        else if (acks != o_acks) v.acks[qo] = (uint8_t)acks;
    }
    for (int k = 0; k < 4; k++) {
        unsigned int x = c[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if (__lane_id() == 0 && x) ctr_add(v.counters, k, (unsigned long long)x);
    }
}

}  // namespace smr

using namespace smr;

struct smr_qread {
    smr_qread_cfg cfg;
    QrView v;
    Arena arena;
};

namespace smr {
template <typename T> static void qcarve(Arena &a, T *&p, size_t n, bool dry) {
    size_t off = a.reserve(n * sizeof(T));
    if (!dry) p = a.at<T>(off);
}
static void qr_layout(smr_qread *h, bool dry) {
    Arena &a = h->arena;
    a.used = 0;
    QrView &v = h->v;
    const size_t G = h->cfg.n_groups, K = h->cfg.n_keys, B = h->cfg.max_reads, Q = h->cfg.n_queries;
    qcarve(a, v.highest_slot, K * G, dry);
    qcarve(a, v.live, Q * G, dry); qcarve(a, v.n, Q * G, dry); qcarve(a, v.acks, Q * G, dry);
    qcarve(a, v.mx_state, Q * B * G, dry); qcarve(a, v.mx_slot, Q * B * G, dry); qcarve(a, v.mx_val, Q * B * G, dry);
    qcarve(a, v.counters, SMR_CTR_WORDS, dry);
}
}  // namespace smr

extern "C" {

int smr_qread_create(const smr_qread_cfg *cfg, smr_qread **out) {
    if (!cfg || !out) return fail(SMR_ERR_ARG, "qread: null argument");
    if (cfg->n_groups == 0) return fail(SMR_ERR_ARG, "qread: n_groups is zero");
    if (cfg->population < 3 || cfg->population > SMR_MAX_REPLICAS) return fail(SMR_ERR_ARG, "qread: population must be in 3..8");
    if (cfg->replica_id >= cfg->population) return fail(SMR_ERR_ARG, "qread: replica_id out of range");
    if (cfg->n_keys == 0 || cfg->n_keys > 255) return fail(SMR_ERR_ARG, "qread: n_keys must be in 1..255");
    if (cfg->max_reads == 0 || cfg->max_reads > 255) return fail(SMR_ERR_ARG, "qread: max_reads must be in 1..255");
    if (cfg->n_queries == 0) return fail(SMR_ERR_ARG, "qread: n_queries is zero");
    smr_qread *h = new smr_qread();
    h->cfg = *cfg;
    memset(&h->v, 0, sizeof(h->v));
    qr_layout(h, true);
    h->arena.size = h->arena.used + 256;
    hipError_t e = hipMalloc((void **)&h->arena.base, h->arena.size);
    if (e != hipSuccess) { delete h; return fail(SMR_ERR_DEVICE, std::string("qread: hipMalloc: ") + hipGetErrorString(e)); }
    qr_layout(h, false);
    QrView &v = h->v;
    v.G = cfg->n_groups; v.K = cfg->n_keys; v.B = cfg->max_reads; v.Q = cfg->n_queries; v.R = cfg->population;
    v.me = cfg->replica_id; v.quorum = cfg->population / 2 + 1;
    e = hipMemset(h->arena.base, 0, h->arena.size);
    if (e == hipSuccess) e = hipMemset(v.highest_slot, 0xFF, (size_t)v.K * v.G * 4);
    if (e != hipSuccess) {
        (void)hipFree(h->arena.base); delete h;
        return fail(SMR_ERR_DEVICE, std::string("qread: init: ") + hipGetErrorString(e));
    }
    *out = h;
    return SMR_OK;
}

void smr_qread_destroy(smr_qread *h) {
    if (!h) return;
    if (h->arena.base) (void)hipFree(h->arena.base);
    delete h;
}

#define QR_GRID(h) dim3(((h)->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream

int smr_qread_refresh_highest_slot(smr_qread *h, const uint32_t *slot_dev, const uint8_t *put_keys_dev, void *stream) {
    if (!h || !slot_dev || !put_keys_dev) return fail(SMR_ERR_ARG, "qread: null argument");
    hipLaunchKernelGGL(qr_refresh_kernel, QR_GRID(h), h->v, slot_dev, put_keys_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_qread_handle_read_query(smr_qread *h, const uint8_t *keys_dev, const uint8_t *n_dev, const uint8_t *stable_leader_dev,
                                const uint32_t *kv_dev, const smr_qread_log *log, const smr_qread_replies *out,
                                uint8_t *from_leader_dev, void *stream) {
    if (!h || !keys_dev || !n_dev || !log || !out || !from_leader_dev) return fail(SMR_ERR_ARG, "qread: null argument");
    if (!log->start_slot || !log->log_end || !log->status || !log->token || !out->state || !out->slot || !out->val)
        return fail(SMR_ERR_ARG, "qread: null argument");
    if (stable_leader_dev && !kv_dev) return fail(SMR_ERR_ARG, "qread: stable_leader without a kv table");
    if (!log->window || (log->window & (log->window - 1))) return fail(SMR_ERR_ARG, "qread: log window must be a power of two");
    if (!log->run_lo != !log->run_hi) return fail(SMR_ERR_ARG, "qread: run_lo and run_hi go together");
    hipLaunchKernelGGL(qr_read_query_kernel, QR_GRID(h), h->v, keys_dev, n_dev, stable_leader_dev, kv_dev, log->start_slot,
                       log->log_end, log->status, log->token, log->window - 1, log->mp_layout, log->run_lo, log->run_hi, log->run_leader, log->run_rep, out->state, out->slot, out->val, from_leader_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_qread_issue(smr_qread *h, uint32_t q, const uint8_t *n_dev, const smr_qread_replies *own, void *stream) {
    if (!h || !n_dev || !own || !own->state || !own->slot || !own->val) return fail(SMR_ERR_ARG, "qread: null argument");
    if (q >= h->v.Q) return fail(SMR_ERR_ARG, "qread: query index out of range");
    hipLaunchKernelGGL(qr_issue_kernel, QR_GRID(h), h->v, q, n_dev, (const uint8_t *)own->state, (const uint32_t *)own->slot,
                       (const uint32_t *)own->val);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_qread_handle_replies(smr_qread *h, uint32_t q, const smr_qread_replies *replies, const uint8_t *flags_dev,
                             const uint32_t *order_dev, uint8_t *outcome_dev, uint32_t *out_val_dev, uint8_t *done_dev, void *stream) {
    if (!h || !replies || !replies->state || !replies->slot || !replies->val || !flags_dev || !outcome_dev || !out_val_dev || !done_dev)
        return fail(SMR_ERR_ARG, "qread: null argument");
    if (q >= h->v.Q) return fail(SMR_ERR_ARG, "qread: query index out of range");
    hipLaunchKernelGGL(qr_replies_kernel, QR_GRID(h), h->v, q, (const uint8_t *)replies->state, (const uint32_t *)replies->slot,
                       (const uint32_t *)replies->val, flags_dev, order_dev, outcome_dev, out_val_dev, done_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_qread_dump(smr_qread *h, uint32_t *highest_slot_host, uint8_t *live_host, uint8_t *n_host, uint8_t *rq_acks_host,
                   uint8_t *mx_state_host, uint32_t *mx_slot_host, uint32_t *mx_val_host, uint64_t *counters_host) {
    if (!h || !highest_slot_host || !live_host || !n_host || !rq_acks_host || !mx_state_host || !mx_slot_host || !mx_val_host ||
        !counters_host)
        return fail(SMR_ERR_ARG, "qread: null argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const QrView &v = h->v;
    const size_t G = v.G, K = v.K, B = v.B, Q = v.Q;
    SMR_HIP_TRY(hipMemcpy(highest_slot_host, v.highest_slot, K * G * 4, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(live_host, v.live, Q * G, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(n_host, v.n, Q * G, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(rq_acks_host, v.acks, Q * G, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(mx_state_host, v.mx_state, Q * B * G, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(mx_slot_host, v.mx_slot, Q * B * G * 4, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(mx_val_host, v.mx_val, Q * B * G * 4, hipMemcpyDeviceToHost));
    unsigned long long c[4];
    SMR_HIP_TRY(ctr_read(v.counters, 4, c));
    for (int k = 0; k < 4; k++) counters_host[k] = c[k];
    // canonical form: rows of queries that are gone, rows past a query's reads and fields its state does not use read 0
    for (size_t q = 0; q < Q; q++)
        for (size_t g = 0; g < G; g++) {
            const size_t qo = q * G + g;
            const bool live = live_host[qo] != 0;
            if (!live) { n_host[qo] = 0; rq_acks_host[qo] = 0; }
            for (size_t i = 0; i < B; i++) {
                const size_t m = (q * B + i) * G + g;
                const bool on = live && i < n_host[qo];
                if (!on) mx_state_host[m] = 0;
                if (!on || mx_state_host[m] == RP_NONE) mx_slot_host[m] = 0;
                if (!on || mx_state_host[m] != RP_VALUE) mx_val_host[m] = 0;
            }
        }
    return SMR_OK;
}

}  // extern "C"

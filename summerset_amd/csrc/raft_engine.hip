// Raft leader-side hot path over G groups (lane = group): log append and the
// AppendEntriesReply match-index quorum kernel.
//
// Stands in for RaftReplica::{handle_req_batch (raft/request.rs:10-91),
// handle_logged_leader_append (raft/durability.rs:12-94, try_next_slot),
// check_term (raft/leadership.rs:16-72), handle_msg_append_entries_reply
// (raft/messages.rs:222-388)}.  The reference rescans the whole uncommitted
// log tail per reply (messages.rs:256-275); here the new commit index is the
// closed form: m = (thresh-1)-th largest peer match_slot, then the highest
// slot <= min(m, log_end-1) above last_commit whose entry term is curr_term.
#include <string.h>

#include <vector>

#include "smr_common.h"

namespace smr {

constexpr int RMAX = SMR_MAX_REPLICAS;
enum { ROLE_FOLLOWER = 0, ROLE_CANDIDATE = 1, ROLE_LEADER = 2 };

struct RaftView {
    uint32_t G, W, Wmask, R, me, thresh;
    uint8_t *role, *leader;
    uint64_t *curr_term;
    uint32_t *log_len, *start_slot, *last_commit, *last_snap;
    uint32_t *next_slot, *try_next_slot, *match_slot;   // [R][G]
    uint64_t *entry_term;                               // [W][G]
    unsigned long long *counters;                       // commits, redirects, rejects, entries sent
};

__device__ __forceinline__ void raft_flush(const RaftView &v, unsigned int c[4]) {
    for (int k = 0; k < 4; k++) {
        unsigned int x = c[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if (__lane_id() == 0 && x) atomicAdd(&v.counters[k], (unsigned long long)x);
    }
}

__global__ __launch_bounds__(256) void raft_append_kernel(const RaftView v, const uint32_t *__restrict__ n_new) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    unsigned int c[4] = {0, 0, 0, 0};
    if (g < v.G) {
        const uint32_t n = n_new[g];
        if (n) {
            if (v.role[g] != ROLE_LEADER) c[1] = n;            // request.rs:19-42 redirect
            else {
                uint32_t len = v.log_len[g];
                const uint32_t start = v.start_slot[g], snap = v.last_snap[g];
                const uint64_t term = v.curr_term[g];
                uint32_t tn[RMAX];
#pragma unroll
                for (int p = 0; p < RMAX; p++) tn[p] = (uint32_t)p < v.R ? v.try_next_slot[(size_t)p * v.G + g] : 0;
                for (uint32_t k = 0; k < n; k++) {
                    if (len - snap >= v.W) { c[2]++; continue; }   // ring back-pressure
                    const uint32_t slot = len;                   // request.rs:77
                    v.entry_term[(size_t)(slot & v.Wmask) * v.G + g] = term;
                    len++;
                    // durability.rs:28-88: who is sent entries, try_next_slot
#pragma unroll
                    for (int p = 0; p < RMAX; p++) {
                        if ((uint32_t)p >= v.R || (uint32_t)p == v.me || tn[p] < 1) continue;
                        uint32_t prev = tn[p] - 1;
                        if (prev < start) break;                 // logged_err
                        if (prev >= len) continue;
                        if (slot >= tn[p]) { c[3] += slot + 1 - tn[p]; tn[p] = slot + 1; }
                    }
                }
                v.log_len[g] = len;
#pragma unroll
                for (int p = 0; p < RMAX; p++)
                    if ((uint32_t)p < v.R && (uint32_t)p != v.me) v.try_next_slot[(size_t)p * v.G + g] = tn[p];
            }
        }
    }
    raft_flush(v, c);
}

__global__ __launch_bounds__(256) void raft_replies_kernel(const RaftView v, const uint64_t *__restrict__ reply_term,
                                                           const uint32_t *__restrict__ end_slot,
                                                           const uint64_t *__restrict__ conflict_term,
                                                           const uint32_t *__restrict__ conflict_slot,
                                                           const uint8_t *__restrict__ flags,
                                                           const uint32_t *__restrict__ order) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    unsigned int c[4] = {0, 0, 0, 0};
    if (g < v.G) {
        uint32_t role = v.role[g], leader = v.leader[g];
        uint64_t term = v.curr_term[g];
        const uint32_t len = v.log_len[g], start = v.start_slot[g];
        uint32_t commit = v.last_commit[g], snap = v.last_snap[g];
        const uint32_t o_role = role, o_leader = leader, o_commit = commit, o_snap = snap;
        const uint64_t o_term = term;
        uint32_t nx[RMAX], tn[RMAX], mt[RMAX];
        bool dirty[RMAX];
#pragma unroll
        for (int p = 0; p < RMAX; p++) {
            bool on = (uint32_t)p < v.R && (uint32_t)p != v.me;
            size_t o = (size_t)p * v.G + g;
            nx[p] = on ? v.next_slot[o] : 0; tn[p] = on ? v.try_next_slot[o] : 0; mt[p] = on ? v.match_slot[o] : 0;
            dirty[p] = false;
        }
        const uint32_t ctl = order ? order[g] : SMR_CTL_IDENTITY;
        for (uint32_t oi = 0; oi < v.R; oi++) {
            const uint32_t p = (ctl >> (3 * oi)) & 7u;
            if (p == v.me || p >= v.R) continue;
            const size_t o = (size_t)p * v.G + g;
            const uint32_t f = flags[o];
            if (!(f & 1)) continue;
            const uint64_t rt = reply_term[o];
            const uint32_t es = end_slot[o];
            // leadership.rs:16-72 check_term
            bool stepped = false;
            if (rt > term) {
                term = rt; leader = p;
                if (role != ROLE_FOLLOWER) { role = ROLE_FOLLOWER; stepped = true; }
            }
            if (stepped || role != ROLE_LEADER) continue;      // messages.rs:239-241
            // registers indexed by a runtime peer id: unrolled select
            uint32_t nxp = 0, tnp = 0;
#pragma unroll
            for (int q = 0; q < RMAX; q++) if ((uint32_t)q == p) { nxp = nx[q]; tnp = tn[q]; }
            uint32_t mtp;
            if (!(f & 2)) {
                if (nxp > es + 1) continue;                     // :245-247
                nxp = es + 1;
                if (tnp < es + 1) tnp = es + 1;
                mtp = es;
#pragma unroll
                for (int q = 0; q < RMAX; q++) if ((uint32_t)q == p) { nx[q] = nxp; tn[q] = tnp; mt[q] = mtp; dirty[q] = true; }
                // commit index, closed form of :256-275
                const uint32_t need = v.thresh - 1;             // peers needed besides me
                uint32_t m = 0xFFFFFFFFu;
                if (need > 0) {
                    m = 0;
#pragma unroll
                    for (int q = 0; q < RMAX; q++) {
                        if ((uint32_t)q >= v.R || (uint32_t)q == v.me) continue;
                        uint32_t ge = 0;
#pragma unroll
                        for (int q2 = 0; q2 < RMAX; q2++)
                            if ((uint32_t)q2 < v.R && (uint32_t)q2 != v.me && mt[q2] >= mt[q]) ge++;
                        if (ge >= need && mt[q] > m) m = mt[q];
                    }
                }
                uint32_t hi = m < len - 1 ? m : len - 1;
                for (uint32_t s = hi; s > commit; s--) {
                    if (s + v.W < len) break;                   // beyond the term ring
                    if (v.entry_term[(size_t)(s & v.Wmask) * v.G + g] == term) {
                        c[0] += s - commit;                     // :278-293 exec submissions
                        commit = s;
                        break;
                    }
                }
                // snapshot-safe index, closed form of :298-309
                uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
                for (int q = 0; q < RMAX; q++)
                    if ((uint32_t)q < v.R && (uint32_t)q != v.me && mt[q] < mn) mn = mt[q];
                uint32_t cand = mn < es ? mn : es;
                if (cand > snap) snap = cand;
            } else {
                if (nxp == 1) {                                 // :313-316
                    tnp = 1;
                } else {
                    nxp -= 1;                                   // :318
                    const uint64_t ct = conflict_term ? conflict_term[o] : 0;
                    const uint32_t cs = conflict_slot ? conflict_slot[o] : 0;
                    for (;;) {                                  // :320-330
                        bool readable = nxp >= start && nxp < len && nxp + v.W >= len;
                        if (!(nxp > start && readable && nxp >= cs && nxp > 1)) break;
                        if (v.entry_term[(size_t)(nxp & v.Wmask) * v.G + g] != ct) break;
                        nxp -= 1;
                    }
                    tnp = nxp;                                  // :331
                    uint32_t prev = nxp - 1;
                    if (prev >= start && prev < len) {          // :335-340
                        if (es + 1 > nxp) c[3] += es + 1 - nxp;
                        tnp = es + 1;                           // :384
                    }
                }
#pragma unroll
                for (int q = 0; q < RMAX; q++) if ((uint32_t)q == p) { nx[q] = nxp; tn[q] = tnp; dirty[q] = true; }
            }
        }
        if (role != o_role) v.role[g] = (uint8_t)role;
        if (leader != o_leader) v.leader[g] = (uint8_t)leader;
        if (term != o_term) v.curr_term[g] = term;
        if (commit != o_commit) v.last_commit[g] = commit;
        if (snap != o_snap) v.last_snap[g] = snap;
#pragma unroll
        for (int p = 0; p < RMAX; p++)
            if (dirty[p]) {
                size_t o = (size_t)p * v.G + g;
                v.next_slot[o] = nx[p]; v.try_next_slot[o] = tn[p]; v.match_slot[o] = mt[p];
            }
    }
    raft_flush(v, c);
}

}  // namespace smr

using namespace smr;

struct smr_raft_leader {
    smr_raft_cfg cfg;
    RaftView v;
    Arena arena;
};

namespace smr {
template <typename T> static void rcarve(Arena &a, T *&p, size_t n, bool dry) {
    size_t off = a.reserve(n * sizeof(T));
    if (!dry) p = a.at<T>(off);
}
static void raft_layout(smr_raft_leader *l, bool dry) {
    Arena &a = l->arena;
    a.used = 0;
    RaftView &v = l->v;
    const size_t G = l->cfg.n_groups, W = l->cfg.window, R = l->cfg.population;
    rcarve(a, v.role, G, dry); rcarve(a, v.leader, G, dry); rcarve(a, v.curr_term, G, dry);
    rcarve(a, v.log_len, G, dry); rcarve(a, v.start_slot, G, dry); rcarve(a, v.last_commit, G, dry);
    rcarve(a, v.last_snap, G, dry);
    rcarve(a, v.next_slot, R * G, dry); rcarve(a, v.try_next_slot, R * G, dry); rcarve(a, v.match_slot, R * G, dry);
    rcarve(a, v.entry_term, W * G, dry);
    rcarve(a, v.counters, 4, dry);
}
}  // namespace smr

extern "C" {

int smr_raft_leader_create(const smr_raft_cfg *cfg, smr_raft_leader **out) {
    if (!cfg || !out) return fail(SMR_ERR_ARG, "raft: null argument");
    if (cfg->n_groups == 0) return fail(SMR_ERR_ARG, "raft: n_groups is zero");
    if (cfg->population < 3 || cfg->population > SMR_MAX_REPLICAS) return fail(SMR_ERR_ARG, "raft: population must be in 3..8");
    if (cfg->leader_id >= cfg->population) return fail(SMR_ERR_ARG, "raft: leader_id out of range");
    if (!cfg->window || (cfg->window & (cfg->window - 1)) || cfg->window < 8)
        return fail(SMR_ERR_ARG, "raft: window must be a power of two >= 8");
    uint32_t quorum = cfg->population / 2 + 1;
    if (cfg->commit_extra > cfg->population - quorum) return fail(SMR_ERR_ARG, "raft: commit_extra too large");
    smr_raft_leader *l = new smr_raft_leader();
    l->cfg = *cfg;
    memset(&l->v, 0, sizeof(l->v));
    raft_layout(l, true);
    l->arena.size = l->arena.used + 256;
    hipError_t e = hipMalloc((void **)&l->arena.base, l->arena.size);
    if (e != hipSuccess) { delete l; return fail(SMR_ERR_DEVICE, std::string("raft: hipMalloc: ") + hipGetErrorString(e)); }
    raft_layout(l, false);
    RaftView &v = l->v;
    v.G = cfg->n_groups; v.W = cfg->window; v.Wmask = cfg->window - 1; v.R = cfg->population;
    v.me = cfg->leader_id; v.thresh = quorum + cfg->commit_extra;
    const size_t G = v.G;
    e = hipMemset(l->arena.base, 0, l->arena.size);
    // state right after become_the_leader on a log holding only the dummy entry
    std::vector<uint64_t> t(G, cfg->term);
    std::vector<uint32_t> one(G * v.R, 1u);
    if (e == hipSuccess) e = hipMemset(v.role, ROLE_LEADER, G);
    if (e == hipSuccess) e = hipMemset(v.leader, cfg->leader_id, G);
    if (e == hipSuccess) e = hipMemcpy(v.curr_term, t.data(), G * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v.log_len, one.data(), G * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v.next_slot, one.data(), G * v.R * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v.try_next_slot, one.data(), G * v.R * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(l->arena.base); delete l;
        return fail(SMR_ERR_DEVICE, std::string("raft: init: ") + hipGetErrorString(e));
    }
    *out = l;
    return SMR_OK;
}

void smr_raft_leader_destroy(smr_raft_leader *l) {
    if (!l) return;
    if (l->arena.base) (void)hipFree(l->arena.base);
    delete l;
}

int smr_raft_leader_append(smr_raft_leader *l, const uint32_t *n_new_dev, void *stream) {
    if (!l || !n_new_dev) return fail(SMR_ERR_ARG, "raft: null argument");
    hipLaunchKernelGGL(raft_append_kernel, dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v, n_new_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_leader_handle_replies(smr_raft_leader *l, const uint64_t *reply_term_dev, const uint32_t *end_slot_dev,
                                   const uint64_t *conflict_term_dev, const uint32_t *conflict_slot_dev,
                                   const uint8_t *flags_dev, const uint32_t *order_dev, void *stream) {
    if (!l || !reply_term_dev || !end_slot_dev || !flags_dev) return fail(SMR_ERR_ARG, "raft: null argument");
    hipLaunchKernelGGL(raft_replies_kernel, dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v,
                       reply_term_dev, end_slot_dev, conflict_term_dev, conflict_slot_dev, flags_dev, order_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_leader_dump(smr_raft_leader *l, const smr_raft_dump_bufs *hb) {
    if (!l || !hb) return fail(SMR_ERR_ARG, "raft: null argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const RaftView &v = l->v;
    const size_t G = v.G, W = v.W, R = v.R;
#define D2H(dst, src, n) SMR_HIP_TRY(hipMemcpy((dst), (src), (n), hipMemcpyDeviceToHost))
    D2H(hb->role, v.role, G); D2H(hb->leader, v.leader, G); D2H(hb->curr_term, v.curr_term, G * 8);
    D2H(hb->log_len, v.log_len, G * 4); D2H(hb->start_slot, v.start_slot, G * 4);
    D2H(hb->last_commit, v.last_commit, G * 4); D2H(hb->last_snap, v.last_snap, G * 4);
    D2H(hb->next_slot, v.next_slot, R * G * 4); D2H(hb->try_next_slot, v.try_next_slot, R * G * 4);
    D2H(hb->match_slot, v.match_slot, R * G * 4);
    std::vector<uint64_t> et(W * G);
    D2H(et.data(), v.entry_term, W * G * 8);
#undef D2H
    for (size_t g = 0; g < G; g++) {
        hb->next_slot[(size_t)v.me * G + g] = 0; hb->try_next_slot[(size_t)v.me * G + g] = 0;
        hb->match_slot[(size_t)v.me * G + g] = 0;
        for (size_t w = 0; w < W; w++) hb->entry_term[w * G + g] = 0;
        uint32_t len = hb->log_len[g], lo = len > W ? len - (uint32_t)W : hb->start_slot[g];
        for (uint32_t s = lo; s < len; s++) hb->entry_term[(size_t)(s & (W - 1)) * G + g] = et[(size_t)(s & (W - 1)) * G + g];
    }
    return SMR_OK;
}

int smr_raft_leader_total_commits(smr_raft_leader *l, uint64_t *out) {
    if (!l || !out) return fail(SMR_ERR_ARG, "raft: null argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[4];
    SMR_HIP_TRY(hipMemcpy(h, l->v.counters, sizeof(h), hipMemcpyDeviceToHost));
    *out = h[0];
    return SMR_OK;
}

}  // extern "C"

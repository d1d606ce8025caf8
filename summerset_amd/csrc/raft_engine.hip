// Raft hot path over G groups (lane = group, one replica per group): leader side (log
// append and the AppendEntriesReply match-index quorum kernel), follower side
// (handle_msg_append_entries, raft/messages.rs:13-218 + durability.rs:97-132) and the
// term / vote state machine (become_a_candidate leadership.rs:76-142, RequestVote
// messages.rs:391-482, RequestVoteReply messages.rs:485-510, become_the_leader
// leadership.rs:145-218).  WAL completions are inline (LS-1 rule 0).
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
#include "raft_peek.h"
#include "wire_rd.h"

namespace smr {

constexpr int RMAX = SMR_MAX_REPLICAS;
constexpr uint8_t NO_REP = 0xFF;
enum { ROLE_FOLLOWER = 0, ROLE_CANDIDATE = 1, ROLE_LEADER = 2 };

struct RaftView {
    uint32_t G, W, Wmask, R, me, thresh;
    uint8_t *role, *leader;
    uint8_t *voted_for, *votes;                          // Option<ReplicaId> (0xFF = None); votes_granted bitmask
    uint32_t *n_exec, *n_trunc;                          // follower: entries submitted for execution, truncations
    uint64_t *curr_term;
    uint32_t *log_len, *start_slot, *last_commit, *last_snap;
    uint32_t *ring_lo;                                  // lowest slot whose term is still in the W-entry ring
    uint32_t *next_slot, *try_next_slot, *match_slot;   // [R][G]
    uint64_t *entry_term;                               // [W][G]
    uint8_t *entry_mask;                                // [W][G] CRaft: avail_shards_map of the entry's codeword (NULL: plain Raft)
    unsigned long long *counters;                       // commits, redirects, rejects, entries sent; CRaft follower: reconstruct_data
                                                        // calls, executions postponed for lack of shards
};

// CRaft leader variant (src/protocols/craft/, a fork of raft/): the fall-back flag of craft/mod.rs:283 and the
// Heartbeater's reply counters (server/heartbeat.rs:52-57) per group; only kernels of the variant see it
struct CraftView {
    uint32_t ft, rep_thr, quorum;
    uint8_t *full_copy, *alive;                          // [G]; alive = peer_alive bitmap
    uint8_t *hb_repeat;                                  // [R][G] reply_cnts.2
    uint64_t *hb_replied, *hb_seen;                      // [R][G] reply_cnts.0 / .1
    // a leader that did not create its whole log (it was a follower once): the shard gate of craft/messages.rs:315-358
    uint8_t *partial;                                    // [G] the log may hold an entry without every shard (set by the follower kernel)
    uint32_t *last_recon, *rq_n;                         // [G] highest slot a Reconstruct was asked for; slots queued since the last poll
    uint32_t *rq_slot;                                   // [CRAFT_RQ][G]
    uint64_t *rq_term;                                   // [CRAFT_RQ][G]
};
constexpr uint32_t CRAFT_RQ = 16;

__device__ __forceinline__ void raft_flush(const RaftView &v, unsigned int c[4]) {
    for (int k = 0; k < 4; k++) {
        unsigned int x = c[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if (__lane_id() == 0 && x) ctr_add(v.counters, k, (unsigned long long)x);
    }
}

// handle_req_batch x n + handle_logged_leader_append per entry (request.rs:10-91, durability.rs:12-94) for one group's leader,
// on registers: `len` and the peers' try_next_slot go in and out, the entries' terms (and CRaft masks) are stored, fs[p] = the
// first slot sent to peer p by these appends (0xFFFFFFFF: none).  c[2] rejects (ring back-pressure), c[3] entries sent.
template <int NR>
__device__ __forceinline__ void raft_append_body(const RaftView &v, uint32_t g, uint32_t n, uint32_t &len, uint32_t start, uint32_t snap,
                                                 uint64_t term, uint32_t (&tn)[NR], uint32_t (&fs)[NR], unsigned int (&c)[4]) {
    // The steady state in closed form (round 4): none of the n appends meets the ring's back-pressure and every peer's
    // try_next_slot lies inside the log (1 <= try_next <= len, its predecessor still held), so the loop below would, per peer,
    // send [try_next, len] with the first append and one more entry with each further one: first slot sent = try_next,
    // entries sent = (len + 1 - try_next) + (n - 1), try_next = len + n.  What is left of the n iterations is their n
    // stores, independent of each other.  (One lane per group is one wavefront per SIMD: the loop's ~45 dependent
    // instructions per append were MORE THAN HALF of the batched tick, 26.1 -> 11.3 us, profiles/r5k, r5l.  Asking for a
    // conflict reply's words and candidate terms a round of loads ahead, with the next tick's inputs, gained nothing --
    // 11.2 us, and 9.4 instead of 9.2 without conflicts, profiles/r5n: what a conflict costs is its divergent code, not
    // its round trips.)
    bool simple = n > 0 && len + n - 1 - snap < v.W;
#pragma unroll
    for (int p = 0; p < NR; p++)
        if ((uint32_t)p < v.R && (uint32_t)p != v.me) simple = simple && tn[p] >= 1 && tn[p] - 1 >= start && tn[p] <= len;
    if (simple) {
        const uint32_t len0 = len;
        for (uint32_t k0 = 0; k0 < n; k0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t k = k0 + (uint32_t)u;
                if (k >= n) break;
                const size_t i = (size_t)((len0 + k) & v.Wmask) * v.G + g;
                v.entry_term[i] = term;                                          // request.rs:77
                if (v.entry_mask) v.entry_mask[i] = (uint8_t)((1u << v.R) - 1u);   // craft/request.rs:71-76: every shard
            }
        }
#pragma unroll
        for (int p = 0; p < NR; p++) {
            if ((uint32_t)p >= v.R || (uint32_t)p == v.me) continue;
            if (fs[p] == 0xFFFFFFFFu) fs[p] = tn[p];
            c[3] += (len0 + 1 - tn[p]) + (n - 1);
            tn[p] = len0 + n;
        }
        len = len0 + n;
        return;
    }
    for (uint32_t k = 0; k < n; k++) {
        if (len - snap >= v.W) { c[2]++; continue; }   // ring back-pressure
        const uint32_t slot = len;                   // request.rs:77
        v.entry_term[(size_t)(slot & v.Wmask) * v.G + g] = term;
        if (v.entry_mask) v.entry_mask[(size_t)(slot & v.Wmask) * v.G + g] = (uint8_t)((1u << v.R) - 1u);   // craft/request.rs:71-76: every shard
        len++;
        // durability.rs:28-88: who is sent entries, try_next_slot
#pragma unroll
        for (int p = 0; p < NR; p++) {
            if ((uint32_t)p >= v.R || (uint32_t)p == v.me || tn[p] < 1) continue;
            uint32_t prev = tn[p] - 1;
            if (prev < start) break;                 // logged_err
            if (prev >= len) continue;
            if (slot >= tn[p]) { if (fs[p] == 0xFFFFFFFFu) fs[p] = tn[p]; c[3] += slot + 1 - tn[p]; tn[p] = slot + 1; }
        }
    }
}

// smr_raft_leader_append(_emit) for one group's lane (raft_append_kernel; the first step of raft_cluster_tick_kernel)
template <int NR>
__device__ __forceinline__ void raft_append_lane(const RaftView &v, uint32_t g, const uint32_t *__restrict__ n_new, uint32_t *__restrict__ ae_first,
                                                 unsigned int (&c)[4]) {
    uint32_t fs[NR];                                     // first slot sent to each peer by this call's appends
#pragma unroll
    for (int p = 0; p < NR; p++) fs[p] = 0xFFFFFFFFu;
    const uint32_t n = n_new[g];
    if (n) {
        if (v.role[g] != ROLE_LEADER) c[1] += n;           // request.rs:19-42 redirect
        else {
            uint32_t len = v.log_len[g];
            const uint32_t start = v.start_slot[g], snap = v.last_snap[g];
            const uint64_t term = v.curr_term[g];
            uint32_t tn[NR];
#pragma unroll
            for (int p = 0; p < NR; p++) tn[p] = (uint32_t)p < v.R ? v.try_next_slot[(size_t)p * v.G + g] : 0;
            raft_append_body<NR>(v, g, n, len, start, snap, term, tn, fs, c);
            v.log_len[g] = len;
            if (len > v.W && len - v.W > v.ring_lo[g]) v.ring_lo[g] = len - v.W;
#pragma unroll
            for (int p = 0; p < NR; p++)
                if ((uint32_t)p < v.R && (uint32_t)p != v.me) v.try_next_slot[(size_t)p * v.G + g] = tn[p];
        }
    }
    if (ae_first) {
#pragma unroll
        for (int p = 0; p < NR; p++) if ((uint32_t)p < v.R) ae_first[(size_t)p * v.G + g] = fs[p];
    }
}
template <int NR>
__global__ __launch_bounds__(256) void raft_append_kernel(const RaftView v, const uint32_t *__restrict__ n_new,
                                                          uint32_t *__restrict__ ae_first) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    unsigned int c[4] = {0, 0, 0, 0};
    if (g < v.G) raft_append_lane<NR>(v, g, n_new, ae_first, c);
    raft_flush(v, c);
}

// CRAFT: craft/messages.rs:256-404 -- every reply a leader takes also counts as a heard heartbeat (:275 ->
// heartbeat.rs:280-296), a success reply is not tested for staleness (:279 is a debug_assert; a release build goes
// on), and the commit rule is `majority + fault_tolerance` matches, `majority` in full-copy mode (:307-313)
// (Measured and dropped, profiles/round2/r2s_raft_lanes_ab.log: eight lanes per group with width-8 shuffles -- 8192 wavefronts --
// ran 29.5 us against this kernel's 17.3: each load then touches eight 32-byte pieces of eight rows instead of one
// contiguous run.  What HAD made one lane per group slow were the shared counter words, see ctr_add.)
// NR >= population: every per-peer array and every loop over peers is NR wide (5 for the common populations -- the
// R x R match-index comparison is 25 selects instead of 64; one lane per group is one wavefront per SIMD, so the kernel's
// time is its straight-line ALU between the two load rounds)
// The leader's per-group state the reply handler works on, in registers (raft_replies_kernel loads and stores it around one
// round of replies; raft_ticks_kernel keeps it across the ticks of a batch)
template <int NR>
struct RaftLeaderRegs {
    uint32_t role, leader, len, start, rlo, commit, snap;
    uint64_t term;
    uint32_t nx[NR], tn[NR], mt[NR];
    bool dirty[NR];
    uint32_t o_role, o_leader, o_commit, o_snap;
    uint64_t o_term;
    __device__ __forceinline__ void load(const RaftView &v, uint32_t g) {
        role = v.role[g]; leader = v.leader[g]; term = v.curr_term[g];
        len = v.log_len[g]; start = v.start_slot[g]; rlo = v.ring_lo[g];
        commit = v.last_commit[g]; snap = v.last_snap[g];
        o_role = role; o_leader = leader; o_commit = commit; o_snap = snap; o_term = term;
#pragma unroll
        for (int p = 0; p < NR; p++) {
            const bool on = (uint32_t)p < v.R && (uint32_t)p != v.me;
            const size_t o = (size_t)p * v.G + g;
            nx[p] = on ? v.next_slot[o] : 0; tn[p] = on ? v.try_next_slot[o] : 0; mt[p] = on ? v.match_slot[o] : 0;
            dirty[p] = false;
        }
    }
    __device__ __forceinline__ void store(const RaftView &v, uint32_t g) const {
        if (role != o_role) v.role[g] = (uint8_t)role;
        if (leader != o_leader) v.leader[g] = (uint8_t)leader;
        if (term != o_term) v.curr_term[g] = term;
        if (commit != o_commit) v.last_commit[g] = commit;
        if (snap != o_snap) v.last_snap[g] = snap;
#pragma unroll
        for (int p = 0; p < NR; p++)
            if (dirty[p]) {
                const size_t o = (size_t)p * v.G + g;
                v.next_slot[o] = nx[p]; v.try_next_slot[o] = tn[p]; v.match_slot[o] = mt[p];
            }
    }
};

// One AppendEntriesReply per peer (rf / rtm / res by peer id; conflict words read where a reply carries one), applied in `ctl`
// order: handle_msg_append_entries_reply (raft/messages.rs:222-388) with check_term.
// One lane per group and 65 536 groups are one wavefront per SIMD: the handler is a chain of dependent memory round trips,
// so the chain is kept at TWO rounds of loads.  Round 1 (the caller's): the group's scalars, every peer's state and every
// peer's reply (addressed by peer id, not by the delivery order, so nothing waits for the order word).  Pass A then replays
// the replies in delivery order in registers -- everything of the handler except the commit scan, whose upper end `hi`
// depends only on the match indices -- and round 2 loads the entry terms at those (<= R - 1) slots together.  Pass B
// finishes the commit scans in the same order; it goes back to memory only when the entry at `hi` is of an older term (never
// in a steady term).
template <bool CRAFT, int NR>
__device__ __forceinline__ void raft_replies_body(const RaftView &v, const CraftView &cv, uint32_t g, RaftLeaderRegs<NR> &S, uint32_t ctl,
                                                  const uint32_t (&rf)[NR], const uint64_t (&rtm)[NR], const uint32_t (&res)[NR],
                                                  const uint64_t *__restrict__ conflict_term, const uint32_t *__restrict__ conflict_slot,
                                                  uint32_t commit_need, uint32_t &heard, unsigned int (&c)[4],
                                                  uint32_t conf_stride = 0, uint32_t conf_g = 0) {   // conf_stride != 0: the conflict arrays are [peer][conf_stride], mine at conf_g
    uint32_t &role = S.role, &leader = S.leader, &commit = S.commit, &snap = S.snap;
    uint64_t &term = S.term;
    const uint32_t len = S.len, start = S.start, rlo = S.rlo;
    uint32_t (&nx)[NR] = S.nx, (&tn)[NR] = S.tn, (&mt)[NR] = S.mt;
    bool (&dirty)[NR] = S.dirty;
    uint32_t hi_at[NR];                                   // by delivery position: upper end of that reply's commit scan
#pragma unroll
    for (int q = 0; q < NR; q++) hi_at[q] = 0xFFFFFFFFu;
    const uint64_t lead_term = term;                        // commit scans only happen while I lead: in this term
    // ---- pass A -------------------------------------------------------------------------------------------
    for (uint32_t oi = 0; oi < v.R; oi++) {
        const uint32_t p = (ctl >> (3 * oi)) & 7u;
        if (p == v.me || p >= v.R) continue;
        const size_t o = (size_t)p * v.G + g;
        // registers indexed by a runtime peer id: unrolled select
        uint32_t f = 0, es = 0, nxp = 0, tnp = 0;
        uint64_t rt = 0;
#pragma unroll
        for (int q = 0; q < NR; q++) if ((uint32_t)q == p) { f = rf[q]; rt = rtm[q]; es = res[q]; nxp = nx[q]; tnp = tn[q]; }
        if (!(f & 1)) continue;
        // leadership.rs:16-72 check_term
        bool stepped = false;
        if (rt > term) {
            term = rt; leader = p;
            v.voted_for[g] = NO_REP; v.votes[g] = 0;        // :21-22
            if (role != ROLE_FOLLOWER) { role = ROLE_FOLLOWER; stepped = true; }
        }
        if (stepped || role != ROLE_LEADER) continue;      // messages.rs:239-241
        if (CRAFT) heard |= 1u << p;
        uint32_t mtp;
        if (!(f & 2)) {
            if (!CRAFT && nxp > es + 1) continue;           // :245-247
            nxp = es + 1;
            if (tnp < es + 1) tnp = es + 1;
            mtp = es;
#pragma unroll
            for (int q = 0; q < NR; q++) if ((uint32_t)q == p) { nx[q] = nxp; tn[q] = tnp; mt[q] = mtp; dirty[q] = true; }
            // commit index, closed form of :256-275
            const uint32_t need = CRAFT ? commit_need : v.thresh - 1;   // peers needed besides me
            uint32_t m = 0xFFFFFFFFu;
            if (need > 0) {
                m = 0;
#pragma unroll
                for (int q = 0; q < NR; q++) {
                    if ((uint32_t)q >= v.R || (uint32_t)q == v.me) continue;
                    uint32_t ge = 0;
#pragma unroll
                    for (int q2 = 0; q2 < NR; q2++)
                        if ((uint32_t)q2 < v.R && (uint32_t)q2 != v.me && mt[q2] >= mt[q]) ge++;
                    if (ge >= need && mt[q] > m) m = mt[q];
                }
            }
            const uint32_t hi = m < len - 1 ? m : len - 1;
#pragma unroll
            for (int q = 0; q < NR; q++) if ((uint32_t)q == oi) hi_at[q] = hi;
            // snapshot-safe index, closed form of :298-309
            uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < NR; q++)
                if ((uint32_t)q < v.R && (uint32_t)q != v.me && mt[q] < mn) mn = mt[q];
            uint32_t cand = mn < es ? mn : es;
            if (cand > snap) snap = cand;
        } else {
            if (nxp == 1) {                                 // :313-316
                tnp = 1;
            } else {
                nxp -= 1;                                   // :318
                // (where the reply's conflict words lie: [peer][G] arrays, or -- raft_wire_replies_kernel, conf_stride = all ones -- a group's
                //  followers side by side from word conf_g on, the leader's own id left out)
                const size_t oc = conf_stride == 0xFFFFFFFFu ? (size_t)conf_g + (p < v.me ? p : p - 1u) : conf_stride ? (size_t)p * conf_stride + conf_g : o;
                const uint64_t ct = conflict_term ? conflict_term[oc] : 0;
                const uint32_t cs = conflict_slot ? conflict_slot[oc] : 0;
                for (bool more = true; more;) {              // :320-330, eight candidates per round of loads: the
                    uint64_t e8[8]; bool ok8[8];            // loop's tests depend on next_slot alone
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const uint32_t cnd = nxp - (uint32_t)k;
                        ok8[k] = (uint32_t)k < nxp && cnd > start && cnd < len && cnd >= rlo && cnd >= cs && cnd > 1;
                        e8[k] = ok8[k] ? v.entry_term[(size_t)(cnd & v.Wmask) * v.G + g] : 0ull;
                    }
                    more = false;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (!ok8[k] || e8[k] != ct) break;
                        nxp -= 1;
                        if (k == 7) more = true;
                    }
                }
                tnp = nxp;                                  // :331
                uint32_t prev = nxp - 1;
                if (prev >= start && prev < len) {          // :335-340
                    if (es + 1 > nxp) c[3] += es + 1 - nxp;
                    tnp = es + 1;                           // :384
                }
            }
#pragma unroll
            for (int q = 0; q < NR; q++) if ((uint32_t)q == p) { nx[q] = nxp; tn[q] = tnp; dirty[q] = true; }
        }
    }
    // ---- round 2: the entry terms at the scans' upper ends, together -------------------------------------------
    uint64_t et[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) {
        const uint32_t h = hi_at[q];
        et[q] = (h != 0xFFFFFFFFu && h > commit && h >= rlo) ? v.entry_term[(size_t)(h & v.Wmask) * v.G + g] : 0ull;
    }
    // ---- pass B: the scans of :256-275 / :278-293 in delivery order ----------------------------------------------
#pragma unroll
    for (int q = 0; q < NR; q++) {
        const uint32_t hi = hi_at[q];
        if (hi == 0xFFFFFFFFu) continue;
        for (uint32_t s2 = hi; s2 > commit; s2--) {
            if (s2 < rlo) break;                            // beyond the term ring
            const uint64_t e = s2 == hi ? et[q] : v.entry_term[(size_t)(s2 & v.Wmask) * v.G + g];
            if (e == lead_term) {
                if (!CRAFT || !cv.partial[g]) {
                    c[0] += s2 - commit;                    // :278-293 exec submissions
                    commit = s2;
                } else {                                    // craft/messages.rs:315-358: only what I hold enough shards of
                    const uint32_t data = (1u << cv.quorum) - 1u;
                    bool can_execute = true;
                    for (uint32_t sl = commit + 1; sl <= s2; sl++) {
                        if (sl < rlo) break;                // harness guard: the entry left the ring
                        const size_t mi = (size_t)(sl & v.Wmask) * v.G + g;
                        const uint32_t m = v.entry_mask[mi];
                        if ((uint32_t)__popc(m) < cv.quorum) {              // :318-325 ask the peers for its shards, once
                            if (sl > cv.last_recon[g]) {
                                const uint32_t k = cv.rq_n[g];
                                if (k < CRAFT_RQ) {
                                    cv.rq_slot[(size_t)k * v.G + g] = sl; cv.rq_term[(size_t)k * v.G + g] = v.entry_term[mi];
                                    cv.rq_n[g] = k + 1;
                                }
                                cv.last_recon[g] = sl;
                            }
                            can_execute = false;
                            continue;
                        } else if ((uint32_t)__popc(m & data) < cv.quorum) { v.entry_mask[mi] = (uint8_t)(m | data); ctr_add(v.counters, 4, 1); }   // :326-328
                        if (can_execute) { c[0]++; commit = sl; }           // :329-345
                    }
                }
                break;
            }
        }
    }
}

// smr_raft_leader_handle_replies for one group's lane (raft_replies_kernel; the last step of raft_cluster_tick_kernel)
template <bool CRAFT, int NR>
__device__ __forceinline__ void raft_replies_lane(const RaftView &v, const CraftView &cv, uint32_t g, const uint64_t *__restrict__ reply_term,
                                                  const uint32_t *__restrict__ end_slot, const uint64_t *__restrict__ conflict_term,
                                                  const uint32_t *__restrict__ conflict_slot, const uint8_t *__restrict__ flags,
                                                  const uint32_t *__restrict__ order, unsigned int (&c)[4]) {
    uint32_t heard = 0, commit_need = v.thresh - 1;     // peers needed besides me
    if (CRAFT && cv.full_copy[g]) commit_need = cv.quorum - 1;
    RaftLeaderRegs<NR> S;
    S.load(v, g);
    uint32_t rf[NR], res[NR];
    uint64_t rtm[NR];
#pragma unroll
    for (int p = 0; p < NR; p++) {
        const bool on = (uint32_t)p < v.R && (uint32_t)p != v.me;
        const size_t o = (size_t)p * v.G + g;
        rf[p] = on ? flags[o] : 0u; rtm[p] = on ? reply_term[o] : 0ull; res[p] = on ? end_slot[o] : 0u;
    }
    raft_replies_body<CRAFT, NR>(v, cv, g, S, order ? order[g] : SMR_CTL_IDENTITY, rf, rtm, res, conflict_term, conflict_slot, commit_need, heard, c);
    S.store(v, g);
    if (CRAFT && heard) {                                   // heartbeat.rs:284-290 update_heard_cnt
        for (uint32_t p = 0; p < v.R; p++)
            if ((heard >> p) & 1u) cv.hb_replied[(size_t)p * v.G + g] += 1;
        const uint32_t al = cv.alive[g];
        if ((al | heard) != al) cv.alive[g] = (uint8_t)(al | heard);
    }
}
template <bool CRAFT, int NR>
__global__ __launch_bounds__(256) void raft_replies_kernel(const RaftView v, const uint64_t *__restrict__ reply_term,
                                                           const uint32_t *__restrict__ end_slot,
                                                           const uint64_t *__restrict__ conflict_term,
                                                           const uint32_t *__restrict__ conflict_slot,
                                                           const uint8_t *__restrict__ flags,
                                                           const uint32_t *__restrict__ order, const CraftView cv) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    unsigned int c[4] = {0, 0, 0, 0};
    if (g < v.G) raft_replies_lane<CRAFT, NR>(v, cv, g, reply_term, end_slot, conflict_term, conflict_slot, flags, order, c);
    raft_flush(v, c);
}

// ---- the leader's receive side of a tick in ONE launch (round 6) ---------------------------------------------------
// smr_wire_ingest_raft_replies parses the followers' connections into [R][G] arrays (two memsets in front: the flag plane, the
// counters) and smr_raft_leader_handle_replies reads them back: three stream operations and 13 bytes per connection out and in
// again for 25-byte frames.  Here the parse is the prologue of the handler.  The connections come DENSE: connection
// c = g * (R - 1) + k is group g's k-th follower (peer ids ascending, the leader's own left out) -- the order a server keeps
// its peers' sockets in -- so a block's 256 / (R - 1) groups have all their connections in the block: every lane parses its
// connection out of the block's LDS copy of the buffer span (wire_rd.h, as wire_ingest_replies_kernel does: same frames taken,
// same frames located, same `consumed` / `status`), leaves the reply in LDS [peer slot][group], and behind one barrier the
// block's first lanes run raft_replies_body on their group (messages.rs:222-309) out of those words.  No intermediate array,
// nothing to zero: a reply that did not come is a flag of zero in LDS.  The call's four counters are summed per block into
// a scratch of the leader's and handed out by the last block (which leaves the scratch zero for the next call).
struct RaftWire {
    const uint8_t *buf; uint64_t buf_len;
    const uint64_t *conn_off; const uint8_t *conn_len; uint32_t n_conn;
    const uint32_t *order;
    smr_wire_other *others; uint64_t other_cap;
    uint64_t *counts, *consumed; int32_t *status;
    unsigned long long *acc;                     // the leader's: [0..3] the counters of the call in flight, [4] blocks done
};

// every counter word of the call lives in L2 and is only ever touched by atomics, so "my adds before my ticket" needs my adds
// ACKNOWLEDGED, not a fence: __threadfence() is an agent-scope release -- an L2 write-back on gfx950 -- once per block,
// and the call's time grew with its number of blocks (47 / 33 / 28 us at 1024 / 512 / 256 blocks, profiles/s14)
#if defined(__HIP_DEVICE_COMPILE__)
#define RW_DRAIN() __builtin_amdgcn_s_waitcnt(0)
#else
#define RW_DRAIN() __threadfence()
#endif
#ifndef RW_BLOCK_N
#define RW_BLOCK_N 1024          // (256 / 512 / 1024 lanes per block: 47.2 / 33.4 / 28.5 us for 262 144 connections, profiles/s14 -- every block
#endif                           //  that located a frame asks ONE word for its place, a returning same-address atomic per block)
constexpr uint32_t RW_BLOCK = RW_BLOCK_N, RW_STAGE = RW_BLOCK * 40, RW_LOC = RW_BLOCK / 2;
// the call's counters without a word every block adds to (a same-address atomic is ~10 ns, a returning one several times that, and
// there are up to 1024 blocks): 16 shards of the three plain counters, arrival tickets in two levels of <= 32 arrivals each --
// only the located frames' places come from ONE word (acc[1]), asked for by blocks that located something
constexpr uint32_t RW_SHARDS = 16, RW_ACC_WORDS = 8 + RW_SHARDS * 4 + 64;   // acc[1]: located; [4]: top ticket; [8 ..): shards; then 64 level-1 tickets
template <bool CRAFT, int NR>
__global__ __launch_bounds__(RW_BLOCK) void raft_wire_replies_kernel(const RaftView v, const CraftView cv, const RaftWire A) {
    __shared__ uint32_t stage[RW_STAGE / 4 + 4];
    __shared__ smr_wire_other loc[RW_LOC];
    __shared__ uint32_t blk[4];
    __shared__ unsigned long long loc_base;
    __shared__ uint64_t sh_term[RW_BLOCK], sh_ct[RW_BLOCK];
    __shared__ uint32_t sh_end[RW_BLOCK], sh_cs[RW_BLOCK];
    __shared__ uint8_t sh_fl[RW_BLOCK];
    if (threadIdx.x < 4) blk[threadIdx.x] = 0;
    const uint32_t F = v.R - 1u, GPB = RW_BLOCK / F;                          // followers per group, groups per block
    const uint32_t g0 = blockIdx.x * GPB, c0 = g0 * F;
    const uint32_t n_here = (A.n_conn - c0) < GPB * F ? (A.n_conn - c0) : GPB * F;
    const bool live = threadIdx.x < n_here;
    const uint32_t c = c0 + threadIdx.x;
    const uint64_t start = live ? A.conn_off[c] : 0, end = !live ? 0 : A.conn_len ? start + A.conn_len[c] : A.conn_off[c + 1];
    const uint32_t c1 = c0 + n_here;
    const uint64_t s0 = A.conn_off[c0] & ~15ull, s1 = A.conn_len ? A.conn_off[c1 - 1] + A.conn_len[c1 - 1] : A.conn_off[c1];
    uint64_t slo = 0, shi = 0;
    if (s0 < s1 && s1 <= A.buf_len) {
        slo = s0; shi = s1 - s0 <= RW_STAGE ? s1 : s0 + RW_STAGE;
        const uint32_t n16 = (uint32_t)((shi - slo + 15) / 16);
        for (uint32_t i = threadIdx.x; i <= n16; i += RW_BLOCK) {
            const uint64_t off = slo + 16ull * i;
            uint32_t w[4] = {0, 0, 0, 0};
            if (off + 16 <= A.buf_len) {
                const wr_u32x4 x = *(const wr_u32x4 *)(A.buf + off);
                w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w;
            } else {
                for (uint32_t b = 0; b < 16 && off + b < A.buf_len; b++) w[b >> 2] |= (uint32_t)A.buf[off + b] << (8 * (b & 3));
            }
            if (4 * i + 3 < RW_STAGE / 4 + 4) { stage[4 * i] = w[0]; stage[4 * i + 1] = w[1]; stage[4 * i + 2] = w[2]; stage[4 * i + 3] = w[3]; }
        }
    }
    sh_fl[threadIdx.x] = 0;
    __syncthreads();
    uint64_t pos = start;
    int st = 0;
    bool have = false, deferred = false;
    if (live && (end < start || end > A.buf_len)) st = 1;
    while (live && st == 0) {
        const uint64_t avail = end - pos;
        if (avail < 8) break;
        GlRd r{A.buf, pos, pos + 8, A.buf_len, true, stage, slo, shi};
        const uint64_t plen = __builtin_bswap64(r.peek64());
        if (plen > 1000000000000ull) { st = 1; break; }                             // safetcp.rs:56-66
        if (avail - 8 < plen) break;
        r.n = pos + 8; r.end = pos + 8 + plen;
        uint32_t kind = SMR_WIRE_OTHER;
        bool mine = false;
        const uint64_t outer = r.varint();
        if (outer == 2) kind = SMR_WIRE_LEAVE;
        else if (outer == 0) {
            const uint64_t var = r.varint();
            if (!r.ok || var > 3) { st = 1; break; }
            kind = (uint32_t)var;
            if (var == 1) {                                                         // AppendEntriesReply
                uint64_t term, end_slot, ct, cs;
                uint8_t has;
                if (!wr_raft_append_reply(r, term, end_slot, has, ct, cs)) { st = 1; break; }
                mine = end_slot <= 0xFFFFFFFFull && cs <= 0xFFFFFFFFull;
                if (mine) {
                    if (have) { deferred = true; break; }
                    sh_term[threadIdx.x] = term; sh_end[threadIdx.x] = (uint32_t)end_slot; sh_ct[threadIdx.x] = ct; sh_cs[threadIdx.x] = (uint32_t)cs;
                    sh_fl[threadIdx.x] = (uint8_t)(1u | (has ? 2u : 0u));
                    have = true;
                }
            }
        }
        if (!r.ok) { st = 1; break; }
        if (!mine) {
            smr_wire_other o; o.conn = c; o.kind = kind; o.off = pos; o.len = 8 + plen;
            const uint32_t at = atomicAdd(&blk[1], 1u);
            if (at < RW_LOC) loc[at] = o;
            else {
                const unsigned long long far = atomicAdd(&A.acc[1], 1ull);        // (past the block's own list: rare)
                if (far < A.other_cap) A.others[far] = o;
            }
        }
        pos += 8 + plen;
    }
    if (live) {
        A.consumed[c] = st ? 0 : pos - start;
        A.status[c] = st;
    }
    const unsigned long long m0 = __ballot(have && st == 0), m2 = __ballot(st != 0), m3 = __ballot(deferred);
    if (threadIdx.x % 64 == 0) {
        if (m0) atomicAdd(&blk[0], (uint32_t)__popcll(m0));
        if (m2) atomicAdd(&blk[2], (uint32_t)__popcll(m2));
        if (m3) atomicAdd(&blk[3], (uint32_t)__popcll(m3));
    }
    __syncthreads();
    // (the located frames' place in the call's list: asked for NOW, used behind the handler -- the word is one for all blocks, and the
    //  round trip of a contended returning atomic is as long as the handler)
    const uint32_t n_kept = blk[1] < RW_LOC ? blk[1] : RW_LOC;
    unsigned long long my_base = 0;
    if (threadIdx.x == 0 && n_kept) my_base = atomicAdd(&A.acc[1], (unsigned long long)n_kept);
    // ---- the handler: one lane per group of the block, its followers' replies out of LDS ----
    unsigned int cn[4] = {0, 0, 0, 0};
    const uint32_t g = g0 + threadIdx.x;
    if (threadIdx.x < GPB && g < v.G) {
        uint32_t heard = 0, commit_need = v.thresh - 1;
        if (CRAFT && cv.full_copy[g]) commit_need = cv.quorum - 1;
        RaftLeaderRegs<NR> S;
        S.load(v, g);
        uint32_t rf[NR], res[NR];
        uint64_t rtm[NR];
#pragma unroll
        for (int p = 0; p < NR; p++) {
            const bool on = (uint32_t)p < v.R && (uint32_t)p != v.me;
            const uint32_t slot = threadIdx.x * F + ((uint32_t)p < v.me ? (uint32_t)p : (uint32_t)p - 1u);   // peer p's connection of my group
            const bool there = on && slot < n_here;
            rf[p] = there ? sh_fl[there ? slot : 0] : 0u; rtm[p] = there ? sh_term[there ? slot : 0] : 0ull; res[p] = there ? sh_end[there ? slot : 0] : 0u;
        }
        const uint32_t base = threadIdx.x * F;                                      // (my group's followers' words start here)
        raft_replies_body<CRAFT, NR>(v, cv, g, S, A.order ? A.order[g] : SMR_CTL_IDENTITY, rf, rtm, res, sh_ct, sh_cs, commit_need, heard, cn, 0xFFFFFFFFu, base);
        S.store(v, g);
        if (CRAFT && heard) {
            for (uint32_t p = 0; p < v.R; p++)
                if ((heard >> p) & 1u) cv.hb_replied[(size_t)p * v.G + g] += 1;
            const uint32_t al = cv.alive[g];
            if ((al | heard) != al) cv.alive[g] = (uint8_t)(al | heard);
        }
    }
    raft_flush(v, cn);
    // ---- the call's counters: this block's share, the located frames' place, and -- the last block -- the totals ----
    unsigned long long *const shard = A.acc + 8 + (blockIdx.x % RW_SHARDS) * 4u;
    if (threadIdx.x == 0) {
        if (blk[0]) atomicAdd(&shard[0], (unsigned long long)blk[0]);
        if (blk[2]) atomicAdd(&shard[2], (unsigned long long)blk[2]);
        if (blk[3]) atomicAdd(&shard[3], (unsigned long long)blk[3]);
        loc_base = my_base;
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < n_kept; j += RW_BLOCK)
        if (loc_base + j < A.other_cap) A.others[loc_base + j] = loc[j];
    if (threadIdx.x == 0) {
        RW_DRAIN();
        // arrivals: my level-1 ticket (blocks 32 k .. 32 k + 31), its last arrival goes on to the top ticket; the last of all hands the totals out
        const uint32_t n1 = (gridDim.x + 31u) / 32u, t1 = blockIdx.x / 32u;
        const uint32_t mine = t1 + 1u < n1 ? 32u : gridDim.x - 32u * (n1 - 1u);
        unsigned long long *const tick1 = A.acc + 8 + RW_SHARDS * 4 + (t1 % 64u);
        bool last = false;
        if (n1 > 64u) last = atomicAdd(&A.acc[4], 1ull) == (unsigned long long)gridDim.x - 1ull;   // (more level-1 groups than tickets: one level)
        else if (atomicAdd(tick1, 1ull) == (unsigned long long)mine - 1ull) {
            atomicExch(tick1, 0ull);
            RW_DRAIN();
            last = atomicAdd(&A.acc[4], 1ull) == (unsigned long long)n1 - 1ull;
        }
        if (last) {                                                                 // every other block's adds are behind their fences
            unsigned long long tot[4] = {0ull, 0ull, 0ull, 0ull};
            for (uint32_t sh = 0; sh < RW_SHARDS; sh++)
#pragma unroll
                for (int q = 0; q < 4; q++) if (q != 1) tot[q] += atomicExch(&A.acc[8 + sh * 4 + q], 0ull);
            tot[1] = atomicExch(&A.acc[1], 0ull);
#pragma unroll
            for (int q = 0; q < 4; q++) A.counts[q] = tot[q];
            atomicExch(&A.acc[4], 0ull);
        }
    }
}

// A batch of <= RAFT_MAX_BATCH ticks in ONE launch: per tick the appends of that tick, then the peers' AppendEntriesReplies of
// that tick -- what smr_raft_leader_append + smr_raft_leader_handle_replies do call by call, with the group's state kept in
// registers across the ticks (groups never talk to each other, so a lane can run its group's ticks back to back) and the
// next tick's inputs requested before this tick's are worked on.  Plain Raft leaders.
constexpr int RAFT_MAX_BATCH = 16;
struct RaftTickBatch {
    uint32_t n;
    const uint32_t *n_new[RAFT_MAX_BATCH];
    const uint64_t *reply_term[RAFT_MAX_BATCH];
    const uint32_t *end_slot[RAFT_MAX_BATCH];
    const uint64_t *conflict_term[RAFT_MAX_BATCH];
    const uint32_t *conflict_slot[RAFT_MAX_BATCH];
    const uint8_t *flags[RAFT_MAX_BATCH];
    const uint32_t *order[RAFT_MAX_BATCH];
};

template <int NR>
__global__ __launch_bounds__(256) void raft_ticks_kernel(const RaftView v, const RaftTickBatch b) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    unsigned int c[4] = {0, 0, 0, 0};
    if (g < v.G) {
        const CraftView cv{};
        RaftLeaderRegs<NR> S;
        S.load(v, g);
        const uint32_t len0 = S.len, rlo0 = S.rlo;
        uint32_t heard = 0;
        // the first tick's inputs (a tick without appends / without replies passes NULL)
        uint32_t n = b.n_new[0] ? b.n_new[0][g] : 0u, ctl = b.order[0] ? b.order[0][g] : SMR_CTL_IDENTITY;
        uint32_t rf[NR], res[NR];
        uint64_t rtm[NR];
#pragma unroll
        for (int p = 0; p < NR; p++) {
            const bool on = (uint32_t)p < v.R && (uint32_t)p != v.me;
            const size_t o = (size_t)p * v.G + g;
            const bool on0 = on && b.flags[0];
            rf[p] = on0 ? b.flags[0][o] : 0u; rtm[p] = on0 ? b.reply_term[0][o] : 0ull; res[p] = on0 ? b.end_slot[0][o] : 0u;
        }
        for (uint32_t t = 0; t < b.n; t++) {
            // the next tick's inputs, on their way while this tick runs (the last tick re-reads its own)
            const uint32_t tn_ = t + 1 < b.n ? t + 1 : t;
            const uint32_t n2 = b.n_new[tn_] ? b.n_new[tn_][g] : 0u, ctl2 = b.order[tn_] ? b.order[tn_][g] : SMR_CTL_IDENTITY;
            uint32_t rf2[NR], res2[NR];
            uint64_t rtm2[NR];
#pragma unroll
            for (int p = 0; p < NR; p++) {
                const bool on = (uint32_t)p < v.R && (uint32_t)p != v.me;
                const size_t o = (size_t)p * v.G + g;
                const bool on2 = on && b.flags[tn_];
                rf2[p] = on2 ? b.flags[tn_][o] : 0u; rtm2[p] = on2 ? b.reply_term[tn_][o] : 0ull; res2[p] = on2 ? b.end_slot[tn_][o] : 0u;
            }
            // smr_raft_leader_append
            if (n) {
                if (S.role != ROLE_LEADER) c[1] += n;          // request.rs:19-42 redirect
                else {
                    uint32_t fs[NR];
#pragma unroll
                    for (int p = 0; p < NR; p++) fs[p] = 0xFFFFFFFFu;
                    const uint32_t before = S.len;
                    raft_append_body<NR>(v, g, n, S.len, S.start, S.snap, S.term, S.tn, fs, c);
                    if (S.len != before) {
#pragma unroll
                        for (int p = 0; p < NR; p++) if ((uint32_t)p < v.R && (uint32_t)p != v.me) S.dirty[p] = true;   // (try_next_slot moved)
                    }
                    if (S.len > v.W && S.len - v.W > S.rlo) S.rlo = S.len - v.W;
                }
            }
            // smr_raft_leader_handle_replies
            raft_replies_body<false, NR>(v, cv, g, S, ctl, rf, rtm, res, b.conflict_term[t], b.conflict_slot[t], v.thresh - 1, heard, c);
            n = n2; ctl = ctl2;
#pragma unroll
            for (int p = 0; p < NR; p++) { rf[p] = rf2[p]; rtm[p] = rtm2[p]; res[p] = res2[p]; }
        }
        S.store(v, g);
        if (S.len != len0) v.log_len[g] = S.len;
        if (S.rlo != rlo0) v.ring_lo[g] = S.rlo;
    }
    raft_flush(v, c);
}

// craft/leadership.rs:249-291 bcast_heartbeats on a leader's send tick: the empty AppendEntries per peer (:252-273),
// Heartbeater::update_bcast_cnts (heartbeat.rs:240-276), then the fall-back test (:283-288)
__global__ __launch_bounds__(256) void craft_heartbeat_kernel(const RaftView v, const CraftView cv, uint8_t *__restrict__ hb_flags,
                                                              uint32_t *__restrict__ prev_slot, uint64_t *__restrict__ prev_term,
                                                              uint32_t *__restrict__ leader_commit,
                                                              uint32_t *__restrict__ last_snap) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    leader_commit[g] = v.last_commit[g]; last_snap[g] = v.last_snap[g];
    const bool sending = v.role[g] == ROLE_LEADER;            // only a leader's Heartbeater ticks (leadership.rs:65,217)
    const uint32_t len = v.log_len[g], start = v.start_slot[g], rlo = v.ring_lo[g];
    uint32_t alive = cv.alive[g];
    const uint32_t o_alive = alive;
    for (uint32_t p = 0; p < v.R; p++) {
        const size_t o = (size_t)p * v.G + g;
        uint8_t f = 0; uint32_t ps = 0; uint64_t pt = 0;
        if (sending && p != v.me) {
            ps = v.try_next_slot[o] - 1;
            if (ps > len - 1) ps = len - 1;
            if (ps >= start && ps >= rlo) { f = 1; pt = v.entry_term[(size_t)(ps & v.Wmask) * v.G + g]; } else { ps = 0; }
            const uint64_t c0 = cv.hb_replied[o], c1 = cv.hb_seen[o];
            uint32_t rep = cv.hb_repeat[o];
            if (c0 > c1) {
                cv.hb_seen[o] = c0;
                if (rep) cv.hb_repeat[o] = 0;
            } else {
                rep += 1;
                if (rep > cv.rep_thr) { alive &= ~(1u << p); rep = 0; }
                cv.hb_repeat[o] = (uint8_t)rep;
            }
        }
        hb_flags[o] = f; prev_slot[o] = ps; prev_term[o] = pt;
    }
    if (!sending) return;
    if (alive != o_alive) cv.alive[g] = (uint8_t)alive;
    if (!cv.full_copy[g] && v.R - (uint32_t)__popc(alive) >= cv.ft) cv.full_copy[g] = 1;   // switch_assignment_mode(true)
}

// switch_assignment_mode (craft/leadership.rs:80-141; to_full NULL = no call) and the shard assignment of a new entry:
// persist = what the leader's WAL entry holds (craft/request.rs:86-100), send[p] = what an AppendEntries to p carries
// (craft/durability.rs:41-80, messages.rs:416-460): the data shards 0..majority in full-copy mode, else one's own shard
__global__ __launch_bounds__(256) void craft_mode_kernel(const RaftView v, const CraftView cv, const uint8_t *__restrict__ to_full,
                                                         uint32_t *__restrict__ persist, uint32_t *__restrict__ send) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    uint32_t full = cv.full_copy[g];
    if (to_full) {
        const uint32_t t = to_full[g];
        if (t <= 1 && t != full) { full = t; cv.full_copy[g] = (uint8_t)t; }
    }
    if (persist) {
        const uint32_t data = (1u << cv.quorum) - 1u;
        persist[g] = full ? data : (1u << v.me);
        for (uint32_t p = 0; p < v.R; p++) send[(size_t)p * v.G + g] = p == v.me ? 0u : (full ? data : (1u << p));
    }
}

// ---- follower side and elections ------------------------------------------------------------
// scalar replica state of one group in registers
struct RaftLane {
    const RaftView &v;
    const uint32_t g;
    uint32_t role, leader, voted_for, votes, len, start, commit, snap, rlo;
    uint64_t term;
    uint32_t o_role, o_leader, o_voted, o_votes, o_len, o_commit, o_snap, o_rlo;
    uint64_t o_term;
    __device__ __forceinline__ RaftLane(const RaftView &v_, uint32_t g_) : v(v_), g(g_) {
        o_role = role = v.role[g]; o_leader = leader = v.leader[g]; o_voted = voted_for = v.voted_for[g];
        o_votes = votes = v.votes[g]; o_len = len = v.log_len[g]; start = v.start_slot[g];
        o_commit = commit = v.last_commit[g]; o_snap = snap = v.last_snap[g]; o_term = term = v.curr_term[g];
        o_rlo = rlo = v.ring_lo[g];
    }
    __device__ __forceinline__ void store() {
        if (role != o_role) v.role[g] = (uint8_t)role;
        if (leader != o_leader) v.leader[g] = (uint8_t)leader;
        if (voted_for != o_voted) v.voted_for[g] = (uint8_t)voted_for;
        if (votes != o_votes) v.votes[g] = (uint8_t)votes;
        if (len != o_len) v.log_len[g] = len;
        if (commit != o_commit) v.last_commit[g] = commit;
        if (snap != o_snap) v.last_snap[g] = snap;
        if (term != o_term) v.curr_term[g] = term;
        if (rlo != o_rlo) v.ring_lo[g] = rlo;
    }
    // leadership.rs:16-72; true iff the role was not Follower and now is
    __device__ __forceinline__ bool check_term(uint32_t peer, uint64_t t) {
        if (t <= term) return false;
        term = t; voted_for = NO_REP; votes = 0; leader = peer;
        if (role == ROLE_FOLLOWER) return false;
        role = ROLE_FOLLOWER;
        return true;
    }
    // entry term if the slot is in the log and still in the W-entry ring (the harness guard)
    __device__ __forceinline__ bool term_at(uint32_t slot, uint64_t &t) const {
        if (slot < start || slot >= len || slot < rlo) return false;
        t = v.entry_term[(size_t)(slot & v.Wmask) * v.G + g];
        return true;
    }
};

// messages.rs:13-218 + durability.rs:97-132.  CRAFT: the fork's follower (craft/messages.rs:14-254): the consistency check
// also on heartbeats (:43-47), the leader recorded also when it fails (:81-84), the shards of a re-sent entry absorbed
// (:133-146, RSCodeword::absorb_other as an OR of availability bitmaps), and execution only of entries with `majority`
// shards, after reconstruct_data when too few of them are data shards (:193-233); entry_mask[s][g] = avail_shards_map of
// the s-th sent entry
// (the pointers carry no __restrict__ here: raft_replicate_kernel hands this body the message its own lane has just written)
template <bool CRAFT>
__device__ __forceinline__ void raft_append_entries_body(
    const RaftView &v, uint32_t g, const uint8_t *flags, const uint8_t *leader_id, const uint64_t *term, const uint32_t *prev_slot,
    const uint64_t *prev_term, const uint32_t *n_entries, const uint64_t *entry_term, const uint8_t *entry_mask, uint8_t *partial, uint32_t K,
    const uint32_t *leader_commit, const uint32_t *last_snap, uint8_t *r_flags, uint64_t *r_term, uint32_t *r_end, uint64_t *r_cterm,
    uint32_t *r_cslot) {
    uint8_t of = 0; uint64_t ot = 0, oct = 0; uint32_t oe = 0, ocs = 0;
    if (flags[g] & 1) {
        RaftLane L(v, g);
        const uint32_t ld = leader_id[g], ps = prev_slot[g];
        const uint64_t tm = term[g], pt = prev_term[g];
        const uint32_t n = n_entries[g] < K ? n_entries[g] : K;
        bool go = true;
        if (L.check_term(ld, tm) || L.role != ROLE_FOLLOWER) {                  // :32
            if (tm == L.term && L.role == ROLE_CANDIDATE) { L.term -= 1; L.check_term(ld, tm); }   // :33-39
            else go = false;
        }
        if (go) {
            uint64_t t_prev = 0;
            const bool ok = L.term_at(ps, t_prev);
            if ((CRAFT || n != 0) && (tm < L.term || ps < L.start || ps >= L.len || !ok || t_prev != pt)) {   // :46-51
                const uint64_t ct = (ps >= L.start && ps < L.len && ok) ? t_prev : 0;
                uint32_t cs = ps;
                while (ct > 0 && cs > L.start) {                                // :60-68
                    uint64_t t;
                    if (L.term_at(cs - 1, t) && t == ct) cs--; else break;
                }
                of = 3; ot = L.term; oe = ps + n; oct = ct; ocs = cs;            // :70-77
                if (CRAFT && tm >= L.term) L.leader = ld;                       // craft/messages.rs:81-84
            } else {
                const uint32_t quorum = v.R / 2 + 1, data = (1u << quorum) - 1u;
                L.leader = ld;                                                  // :94
                uint32_t first_new = ps + 1;                                    // :99-139
                for (uint32_t s = 0; s < n; s++) {
                    const uint32_t slot = ps + 1 + s;
                    if (slot >= L.len) { first_new = slot; break; }
                    uint64_t t;
                    // (harness: an entry that has left the W-entry term ring is >= W entries behind the log's end -- its term is
                    // no longer held, which is NOT a term conflict: it is taken as matching, never as a reason to truncate a
                    // suffix that may be committed; the oracle shares the rule)
                    if (slot >= L.start && slot < L.rlo) { ctr_add(v.counters, 6, 1); continue; }   // (counted: smr_raft_ring_guard_hits)
                    if (!L.term_at(slot, t) || t != entry_term[(size_t)s * v.G + g]) {
                        L.len = slot;                                           // :136 truncate
                        v.n_trunc[g] += 1;
                        first_new = slot;
                        break;
                    }
                    if (CRAFT) {                                                // craft/messages.rs:133-146 absorb the sent shards
                        const size_t mi = (size_t)(slot & v.Wmask) * v.G + g;
                        const uint32_t m = v.entry_mask[mi], em = entry_mask[(size_t)s * v.G + g];
                        if ((uint32_t)__popc(m & data) < quorum && m != em) v.entry_mask[mi] = (uint8_t)(m | em);
                    }
                }
                // :143-167: everything from first_new on is PUSHED (also when nothing differed)
                const uint32_t skipped = first_new - ps - 1, slot_e = ps + n;
                uint32_t appended = 0;
                for (uint32_t s = skipped; s < n; s++) {
                    const uint32_t slot = (s - skipped) + first_new;
                    v.entry_term[(size_t)(L.len & v.Wmask) * v.G + g] = entry_term[(size_t)s * v.G + g];
                    if (CRAFT) {
                        const uint8_t em = entry_mask[(size_t)s * v.G + g];
                        v.entry_mask[(size_t)(L.len & v.Wmask) * v.G + g] = em;
                        if (em != (uint8_t)((1u << v.R) - 1u) && !partial[g]) partial[g] = 1;   // a later leadership has to gate on shards
                    }
                    L.len++;
                    if (L.len > v.W && L.len - v.W > L.rlo) L.rlo = L.len - v.W;
                    appended++;
                    if (slot >= L.start && L.role == ROLE_FOLLOWER && slot == slot_e && L.leader != NO_REP) {   // durability.rs:104-128
                        of = 1; ot = L.term; oe = slot_e;
                    }
                }
                if (appended == 0) { of = 1; ot = L.term; oe = first_new - 1; }  // :172-181
                const uint32_t lc = leader_commit[g];
                if (lc > L.commit) {                                            // :184-208 (entries.len() == skipped by now)
                    uint32_t nc = lc < ps + skipped ? lc : ps + skipped;
                    if (nc > L.len - 1) nc = L.len - 1;
                    if (!CRAFT) {
                        if (nc > L.commit) v.n_exec[g] += nc - L.commit;
                        L.commit = nc;
                    } else {
                        uint32_t ex = 0;
                        for (uint32_t slot = L.commit + 1; slot <= nc; slot++) {   // craft/messages.rs:193-233
                            if (slot < L.rlo) break;                            // harness guard: the entry left the ring
                            const size_t mi = (size_t)(slot & v.Wmask) * v.G + g;
                            const uint32_t m = v.entry_mask[mi];
                            if ((uint32_t)__popc(m) < quorum) { ctr_add(v.counters, 5, 1); break; }          // not enough shards yet
                            if ((uint32_t)__popc(m & data) < quorum) { v.entry_mask[mi] = (uint8_t)(m | data); ctr_add(v.counters, 4, 1); }   // reconstruct_data
                            ex++;
                            L.commit = slot;
                        }
                        if (ex) v.n_exec[g] += ex;
                    }
                }
                const uint32_t ls = last_snap[g];
                if (ls > L.snap) L.snap = ls;                                   // :211-213
            }
        }
        L.store();
    }
    r_flags[g] = of; r_term[g] = ot; r_end[g] = oe; r_cterm[g] = oct; r_cslot[g] = ocs;
}
template <bool CRAFT>
__global__ __launch_bounds__(256) void raft_append_entries_kernel(
    const RaftView v, const uint8_t *__restrict__ flags, const uint8_t *__restrict__ leader_id,
    const uint64_t *__restrict__ term, const uint32_t *__restrict__ prev_slot, const uint64_t *__restrict__ prev_term,
    const uint32_t *__restrict__ n_entries, const uint64_t *__restrict__ entry_term, const uint8_t *__restrict__ entry_mask,
    uint8_t *__restrict__ partial, uint32_t K,
    const uint32_t *__restrict__ leader_commit, const uint32_t *__restrict__ last_snap, uint8_t *__restrict__ r_flags,
    uint64_t *__restrict__ r_term, uint32_t *__restrict__ r_end, uint64_t *__restrict__ r_cterm,
    uint32_t *__restrict__ r_cslot) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    raft_append_entries_body<CRAFT>(v, g, flags, leader_id, term, prev_slot, prev_term, n_entries, entry_term, entry_mask, partial, K, leader_commit,
                                    last_snap, r_flags, r_term, r_end, r_cterm, r_cslot);
}

// craft/messages.rs:622-663 handle_msg_reconstruct: the codeword (as its availability bitmap) of every asked slot I hold
// under the asked term
__global__ __launch_bounds__(256) void craft_reconstruct_kernel(const RaftView v, const uint32_t *__restrict__ n, const uint32_t *__restrict__ slot,
                                                                const uint64_t *__restrict__ term, uint32_t K, uint32_t *__restrict__ r_n,
                                                                uint8_t *__restrict__ r_has, uint8_t *__restrict__ r_mask) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    const uint32_t start = v.start_slot[g], len = v.log_len[g], rlo = v.ring_lo[g], cnt = n[g];
    uint32_t out = 0;
    for (uint32_t k = 0; k < K; k++) {
        const size_t o = (size_t)k * v.G + g;
        uint8_t has = 0, m = 0;
        if (k < cnt) {
            const uint32_t sl = slot[o];
            if (sl >= start && sl < len && sl >= rlo && v.entry_term[(size_t)(sl & v.Wmask) * v.G + g] == term[o]) {   // :631-636
                has = 1; m = v.entry_mask[(size_t)(sl & v.Wmask) * v.G + g]; out++;
            }
        }
        r_has[o] = has; r_mask[o] = m;
    }
    r_n[g] = out;
}

// craft/messages.rs:665-745 handle_msg_reconstruct_reply: ReconstructReply { slots_data } from peer[g] (NO_REP: none): absorb the
// shards, and when the slot behind last_commit arrived go on executing up to the shadow commit index
__global__ __launch_bounds__(256) void craft_reconstruct_reply_kernel(const RaftView v, const CraftView cv, const uint8_t *__restrict__ peer,
                                                                      const uint32_t *__restrict__ n, const uint32_t *__restrict__ slot,
                                                                      const uint8_t *__restrict__ mask, uint32_t K) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    unsigned int c[4] = {0, 0, 0, 0};
    if (g < v.G && peer[g] != NO_REP && peer[g] < v.R) {
        const uint32_t pe = peer[g];
        if (pe != v.me) {                                                       // :669 heard_heartbeat -> heartbeat.rs:284-290
            cv.hb_replied[(size_t)pe * v.G + g] += 1;
            const uint32_t al = cv.alive[g];
            if (!((al >> pe) & 1u)) cv.alive[g] = (uint8_t)(al | (1u << pe));
        }
        uint32_t ms[RMAX]; int nm = 0;                                          // :670-683 shadow_last_commit
        for (uint32_t q = 0; q < v.R; q++) if (q != v.me) ms[nm++] = v.match_slot[(size_t)q * v.G + g];
        for (int a = 0; a < nm; a++) for (int b = a + 1; b < nm; b++) if (ms[b] > ms[a]) { const uint32_t t = ms[a]; ms[a] = ms[b]; ms[b] = t; }
        const uint32_t idx = cv.full_copy[g] ? cv.quorum - 2 : cv.quorum + cv.ft - 2;
        const uint32_t shadow = ms[idx];
        const uint32_t start = v.start_slot[g], len = v.log_len[g], rlo = v.ring_lo[g], data = (1u << cv.quorum) - 1u;
        uint32_t commit = v.last_commit[g];
        const uint32_t o_commit = commit, cnt = n[g] < K ? n[g] : K;
        for (uint32_t k = 0; k < cnt; k++) {
            const uint32_t sl = slot[(size_t)k * v.G + g];
            if (sl < start || sl >= len || sl < rlo) continue;                  // :685-687
            const size_t mi = (size_t)(sl & v.Wmask) * v.G + g;
            v.entry_mask[mi] = (uint8_t)(v.entry_mask[mi] | mask[(size_t)k * v.G + g]);   // :697 absorb_other
            if (sl == commit + 1) {                                             // :699-737
                while (commit < shadow) {
                    const uint32_t nx = commit + 1;
                    if (nx >= len || nx < rlo) break;
                    const size_t ni = (size_t)(nx & v.Wmask) * v.G + g;
                    const uint32_t m = v.entry_mask[ni];
                    if ((uint32_t)__popc(m) < cv.quorum) break;
                    if ((uint32_t)__popc(m & data) < cv.quorum) { v.entry_mask[ni] = (uint8_t)(m | data); ctr_add(v.counters, 4, 1); }
                    c[0]++;
                    commit = nx;
                }
            }
        }
        if (commit != o_commit) v.last_commit[g] = commit;
    }
    raft_flush(v, c);
}

// the Reconstruct { slots } the reply kernel queued (craft/messages.rs:347-358), handed over and cleared
__global__ __launch_bounds__(256) void craft_poll_reconstructs_kernel(const RaftView v, const CraftView cv, uint32_t K, uint32_t *__restrict__ n,
                                                                      uint32_t *__restrict__ slot, uint64_t *__restrict__ term) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    const uint32_t have = cv.rq_n[g], cnt = have < K ? have : K;
    for (uint32_t k = 0; k < K; k++) {
        slot[(size_t)k * v.G + g] = k < cnt ? cv.rq_slot[(size_t)k * v.G + g] : 0u;
        term[(size_t)k * v.G + g] = k < cnt ? cv.rq_term[(size_t)k * v.G + g] : 0ull;
    }
    n[g] = cnt;
    if (have) cv.rq_n[g] = 0;
}

// leadership.rs:76-142
__global__ __launch_bounds__(256) void raft_become_candidate_kernel(const RaftView v, const uint8_t *__restrict__ timeout_src,
                                                                    uint8_t *__restrict__ rv_flags, uint64_t *__restrict__ rv_term,
                                                                    uint32_t *__restrict__ rv_last_slot,
                                                                    uint64_t *__restrict__ rv_last_term) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    uint8_t of = 0; uint64_t ot = 0, olt = 0; uint32_t ols = 0;
    const uint32_t src = timeout_src[g];
    if (src != NO_REP) {
        RaftLane L(v, g);
        if (L.role == ROLE_FOLLOWER && !(L.leader != NO_REP && L.leader != src)) {   // :80-85
            L.role = ROLE_CANDIDATE;
            L.term += 1; L.voted_for = v.me; L.votes = 1u << v.me;              // :90-92
            ols = L.len - 1;                                                    // :99-101
            uint64_t t = 0;
            olt = L.term_at(ols, t) ? t : 0;
            of = 1; ot = L.term;
            L.store();
        }
    }
    rv_flags[g] = of; rv_term[g] = ot; rv_last_slot[g] = ols; rv_last_term[g] = olt;
}

// messages.rs:391-482
__global__ __launch_bounds__(256) void raft_request_vote_kernel(const RaftView v, const uint8_t *__restrict__ flags,
                                                                const uint8_t *__restrict__ cand, const uint64_t *__restrict__ term,
                                                                const uint32_t *__restrict__ last_slot,
                                                                const uint64_t *__restrict__ last_term,
                                                                uint8_t *__restrict__ r_flags, uint64_t *__restrict__ r_term) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    uint8_t of = 0; uint64_t ot = 0;
    if (flags[g] & 1) {
        RaftLane L(v, g);
        const uint32_t c = cand[g];
        const uint64_t tm = term[g];
        L.check_term(c, tm);                                                    // :405
        if (tm < L.term) { of = 1; ot = L.term; }                               // :408-422
        else if (L.voted_for == NO_REP || L.voted_for == c) {                   // :427
            uint64_t my_last = 0;
            L.term_at(L.len - 1, my_last);
            const uint64_t lt = last_term[g];
            if (lt >= my_last || (lt == L.term && last_slot[g] + 1 >= L.len)) { // :428-430
                of = 3; ot = L.term;
                L.voted_for = c;                                                // :450
            }
        }
        L.store();
    }
    r_flags[g] = of; r_term[g] = ot;
}

// messages.rs:485-510 + leadership.rs:145-218
__global__ __launch_bounds__(256) void raft_vote_replies_kernel(const RaftView v, const uint64_t *__restrict__ term,
                                                                const uint8_t *__restrict__ flags,
                                                                const uint32_t *__restrict__ order,
                                                                uint32_t *__restrict__ hb_prev, uint8_t *__restrict__ elected) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    RaftLane L(v, g);
    uint8_t el = 0;
    for (uint32_t p = 0; p < v.R; p++) hb_prev[(size_t)p * v.G + g] = 0xFFFFFFFFu;
    const uint32_t ctl = order ? order[g] : SMR_CTL_IDENTITY;
    const uint32_t quorum = v.R / 2 + 1;
    for (uint32_t oi = 0; oi < v.R; oi++) {
        const uint32_t p = (ctl >> (3 * oi)) & 7u;
        if (p == v.me || p >= v.R) continue;
        const size_t o = (size_t)p * v.G + g;
        if (!(flags[o] & 1)) continue;
        if (L.check_term(p, term[o]) || L.role != ROLE_CANDIDATE) continue;     // :498-500
        L.votes |= 1u << p;                                                     // :503 (whatever `granted` says)
        if ((uint32_t)__popc(L.votes) >= quorum) {                              // :506-508
            L.role = ROLE_LEADER;                                               // leadership.rs:149
            for (uint32_t q = 0; q < v.R; q++) {                                // :156 -> :186-196, before the re-init below
                if (q == v.me) continue;
                const uint32_t a = v.try_next_slot[(size_t)q * v.G + g] - 1, b = L.len - 1;
                hb_prev[(size_t)q * v.G + g] = a < b ? a : b;
            }
            for (uint32_t q = 0; q < v.R; q++) {                                // :159-168
                const size_t oq = (size_t)q * v.G + g;
                v.next_slot[oq] = L.len; v.try_next_slot[oq] = L.len; v.match_slot[oq] = 0;
            }
            el = 1;
        }
    }
    L.store();
    elected[g] = el;
}

// The AppendEntries a leader's appends produced for one peer, as ONE message per group (the reference
// sends one per appended batch, durability.rs:57-80; a follower handling them in order ends in the same
// state): entries [first, min(first + K, log end)), prev = first - 1.
__device__ __forceinline__ void raft_gather_body(const RaftView &v, uint32_t g, const uint32_t *first, uint32_t K, uint8_t *flags, uint8_t *leader,
                                                 uint64_t *term, uint32_t *prev_slot, uint64_t *prev_term, uint32_t *n_entries, uint64_t *entry_term,
                                                 uint32_t *leader_commit, uint32_t *last_snap) {
    RaftLane L(v, g);
    const uint32_t f = first[g];
    uint8_t fl = 0; uint32_t ps = 0, n = 0; uint64_t pt = 0;
    if (f != 0xFFFFFFFFu && L.role == ROLE_LEADER && f >= 1 && f <= L.len && L.term_at(f - 1, pt)) {
        n = L.len - f;
        if (n > K) n = K;
        fl = 1; ps = f - 1;
    } else pt = 0;
    for (uint32_t k = 0; k < K; k++) {
        uint64_t t = 0;
        if (k < n) L.term_at(f + k, t);
        entry_term[(size_t)k * v.G + g] = t;
    }
    flags[g] = fl; leader[g] = (uint8_t)v.me; term[g] = L.term; prev_slot[g] = ps; prev_term[g] = pt; n_entries[g] = n;
    leader_commit[g] = L.commit; last_snap[g] = L.snap;
}
__global__ __launch_bounds__(256) void raft_gather_kernel(const RaftView v, const uint32_t *__restrict__ first, uint32_t K,
                                                          uint8_t *__restrict__ flags, uint8_t *__restrict__ leader,
                                                          uint64_t *__restrict__ term, uint32_t *__restrict__ prev_slot,
                                                          uint64_t *__restrict__ prev_term, uint32_t *__restrict__ n_entries,
                                                          uint64_t *__restrict__ entry_term, uint32_t *__restrict__ leader_commit,
                                                          uint32_t *__restrict__ last_snap) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    raft_gather_body(v, g, first, K, flags, leader, term, prev_slot, prev_term, n_entries, entry_term, leader_commit, last_snap);
}

// The leader's AppendEntries for n co-located followers and the followers' handlers in ONE launch (blockIdx.y = which follower):
// the gather of follower k's message out of the leader's log, then -- same lane, the message it has just written -- follower k's
// handle_msg_append_entries.  What smr_raft_leader_gather_entries + smr_raft_replica_handle_append_entries do in 2 n launches; the
// leader's state is only read, follower k's only touched by its own blocks.  The followers' views are read from the device copies
// their objects keep (a by-value table indexed by the block goes to scratch, csrc/rsp_payload.hip PsMany); the per-follower
// pointers are picked out of the argument with compile-time indices.
struct RaftRepl {
    const RaftView *fv[RMAX];
    uint8_t *partial[RMAX];
    const uint32_t *first[RMAX];
    uint8_t *m_flags[RMAX], *m_leader[RMAX];
    uint64_t *m_term[RMAX], *m_prev_term[RMAX], *m_eterm[RMAX];
    uint32_t *m_prev_slot[RMAX], *m_n[RMAX], *m_lc[RMAX], *m_ls[RMAX];
    const uint8_t *m_emask[RMAX];
    uint8_t *r_flags[RMAX];
    uint64_t *r_term[RMAX], *r_cterm[RMAX];
    uint32_t *r_end[RMAX], *r_cslot[RMAX];
    uint32_t K;
};
template <bool CRAFT>
__global__ __launch_bounds__(256) void raft_replicate_kernel(const RaftView lv, const RaftRepl A) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    const RaftView *fvp = nullptr;
    uint8_t *partial = nullptr, *m_flags = nullptr, *m_leader = nullptr, *r_flags = nullptr;
    const uint32_t *first = nullptr;
    uint64_t *m_term = nullptr, *m_prev_term = nullptr, *m_eterm = nullptr, *r_term = nullptr, *r_cterm = nullptr;
    uint32_t *m_prev_slot = nullptr, *m_n = nullptr, *m_lc = nullptr, *m_ls = nullptr, *r_end = nullptr, *r_cslot = nullptr;
    const uint8_t *m_emask = nullptr;
#pragma unroll
    for (int k = 0; k < (int)RMAX; k++)
        if (blockIdx.y == (unsigned)k) {
            fvp = A.fv[k]; partial = A.partial[k]; first = A.first[k]; m_flags = A.m_flags[k]; m_leader = A.m_leader[k]; m_term = A.m_term[k];
            m_prev_term = A.m_prev_term[k]; m_eterm = A.m_eterm[k]; m_prev_slot = A.m_prev_slot[k]; m_n = A.m_n[k]; m_lc = A.m_lc[k];
            m_ls = A.m_ls[k]; m_emask = A.m_emask[k]; r_flags = A.r_flags[k]; r_term = A.r_term[k]; r_cterm = A.r_cterm[k]; r_end = A.r_end[k];
            r_cslot = A.r_cslot[k];
        }
    const RaftView fv = *fvp;                         // a copy in registers
    if (g >= lv.G) return;
    raft_gather_body(lv, g, first, A.K, m_flags, m_leader, m_term, m_prev_slot, m_prev_term, m_n, m_eterm, m_lc, m_ls);
    raft_append_entries_body<CRAFT>(fv, g, m_flags, m_leader, m_term, m_prev_slot, m_prev_term, m_n, m_eterm, m_emask, partial, A.K, m_lc, m_ls, r_flags,
                                    r_term, r_end, r_cterm, r_cslot);
}

// A co-located cluster's whole steady tick in ONE launch (round 6, VERDICT r5 #5: "CRaft's engine tick as one launch, like RSPaxos's"):
// smr_raft_leader_append_emit + smr_raft_cluster_replicate + smr_raft_leader_handle_replies.  Groups never talk to each other, so the
// three steps only have to be ordered within a group: a block owns 64 groups, wavefront 0 is the leader, wavefront 1 + k follower k,
// and a block barrier stands where the three launches have a kernel boundary (HIP's barrier carries a workgroup-scope release /
// acquire and the wavefronts of a block share their CU's L1: what the leader's wavefront stored is what the followers' load, and
// back -- as mp_ticks_fused relies on).  Same state, messages, replies and counters as the three calls.
struct RaftTickRest {
    const uint32_t *n_new; uint32_t *ae_first;
    const uint64_t *reply_term, *conflict_term; const uint32_t *end_slot, *conflict_slot, *order; const uint8_t *flags;
    uint32_t n;
};
template <bool CRAFT, int NR>
__global__ __launch_bounds__(64 * (RMAX + 1)) void raft_cluster_tick_kernel(const RaftView lv, const CraftView cv, const RaftRepl A, const RaftTickRest T) {
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u, g = blockIdx.x * 64u + lane;
    unsigned int c[4] = {0, 0, 0, 0};
    if (w == 0 && g < lv.G) raft_append_lane<NR>(lv, g, T.n_new, T.ae_first, c);
    __syncthreads();
    if (w >= 1u && w <= T.n) {
        const RaftView *fvp = nullptr;
        uint8_t *partial = nullptr, *m_flags = nullptr, *m_leader = nullptr, *r_flags = nullptr;
        const uint32_t *first = nullptr;
        uint64_t *m_term = nullptr, *m_prev_term = nullptr, *m_eterm = nullptr, *r_term = nullptr, *r_cterm = nullptr;
        uint32_t *m_prev_slot = nullptr, *m_n = nullptr, *m_lc = nullptr, *m_ls = nullptr, *r_end = nullptr, *r_cslot = nullptr;
        const uint8_t *m_emask = nullptr;
#pragma unroll
        for (int k = 0; k < (int)RMAX; k++)
            if (w == (unsigned)k + 1u) {
                fvp = A.fv[k]; partial = A.partial[k]; first = A.first[k]; m_flags = A.m_flags[k]; m_leader = A.m_leader[k]; m_term = A.m_term[k];
                m_prev_term = A.m_prev_term[k]; m_eterm = A.m_eterm[k]; m_prev_slot = A.m_prev_slot[k]; m_n = A.m_n[k]; m_lc = A.m_lc[k];
                m_ls = A.m_ls[k]; m_emask = A.m_emask[k]; r_flags = A.r_flags[k]; r_term = A.r_term[k]; r_cterm = A.r_cterm[k]; r_end = A.r_end[k];
                r_cslot = A.r_cslot[k];
            }
        const RaftView fv = *fvp;                         // a copy in registers
        if (g < lv.G) {
            raft_gather_body(lv, g, first, A.K, m_flags, m_leader, m_term, m_prev_slot, m_prev_term, m_n, m_eterm, m_lc, m_ls);
            raft_append_entries_body<CRAFT>(fv, g, m_flags, m_leader, m_term, m_prev_slot, m_prev_term, m_n, m_eterm, m_emask, partial, A.K, m_lc, m_ls,
                                            r_flags, r_term, r_end, r_cterm, r_cslot);
        }
    }
    __syncthreads();
    if (w == 0) {
        if (g < lv.G) raft_replies_lane<CRAFT, NR>(lv, cv, g, T.reply_term, T.end_slot, T.conflict_term, T.conflict_slot, T.flags, T.order, c);
        raft_flush(lv, c);
    }
}

}  // namespace smr

using namespace smr;

struct smr_raft_leader {
    smr_raft_cfg cfg;
    RaftView v;
    Arena arena;
    bool craft = false;
    CraftView cv;
    uint8_t *craft_base = nullptr;
    int device = -1;                 // the device current when the object was created: where its arena is
    unsigned long long *wire_acc = nullptr;   // smr_raft_leader_handle_wire_replies: the counters of the call in flight + blocks done (zero between calls)
    RaftView *d_view = nullptr;      // a device copy of v for smr_raft_cluster_replicate (made on first use; craft_enable changes v)
    bool d_view_ok = false;
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
    rcarve(a, v.voted_for, G, dry); rcarve(a, v.votes, G, dry); rcarve(a, v.n_exec, G, dry); rcarve(a, v.n_trunc, G, dry);
    rcarve(a, v.log_len, G, dry); rcarve(a, v.start_slot, G, dry); rcarve(a, v.last_commit, G, dry);
    rcarve(a, v.last_snap, G, dry); rcarve(a, v.ring_lo, G, dry);
    rcarve(a, v.next_slot, R * G, dry); rcarve(a, v.try_next_slot, R * G, dry); rcarve(a, v.match_slot, R * G, dry);
    rcarve(a, v.entry_term, W * G, dry);
    rcarve(a, v.counters, SMR_CTR_WORDS, dry);
}
RaftPeek raft_peek(const smr_raft_leader *l) {
    const RaftView &v = l->v;
    return RaftPeek{v.G, v.W, v.R, l->craft ? l->cv.quorum : v.R / 2 + 1, v.entry_term, l->craft ? v.entry_mask : nullptr,
                    v.log_len, v.start_slot, v.ring_lo};
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
    if (e == hipSuccess) e = hipMemset(v.voted_for, 0xFF, G);
    if (e == hipSuccess) e = hipMemcpy(v.curr_term, t.data(), G * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v.log_len, one.data(), G * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v.next_slot, one.data(), G * v.R * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v.try_next_slot, one.data(), G * v.R * 4, hipMemcpyHostToDevice);
    // the device copy of the view smr_raft_cluster_replicate's followers are read through: made HERE (and again by craft_enable),
    // so that the stream-ordered call allocates and copies nothing (a blocking hipMalloc + hipMemcpy inside it)
    if (e == hipSuccess) e = hipGetDevice(&l->device);
    if (e == hipSuccess) e = hipMalloc((void **)&l->wire_acc, RW_ACC_WORDS * 8);
    if (e == hipSuccess) e = hipMemset(l->wire_acc, 0, RW_ACC_WORDS * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&l->d_view, sizeof(RaftView));
    if (e == hipSuccess) e = hipMemcpy(l->d_view, &l->v, sizeof(RaftView), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (l->d_view) (void)hipFree(l->d_view);
        if (l->wire_acc) (void)hipFree(l->wire_acc);
        (void)hipFree(l->arena.base); delete l;
        return fail(SMR_ERR_DEVICE, std::string("raft: init: ") + hipGetErrorString(e));
    }
    l->d_view_ok = true;
    *out = l;
    return SMR_OK;
}

void smr_raft_leader_destroy(smr_raft_leader *l) {
    if (!l) return;
    if (l->arena.base) (void)hipFree(l->arena.base);
    if (l->craft_base) (void)hipFree(l->craft_base);
    if (l->d_view) (void)hipFree(l->d_view);
    if (l->wire_acc) (void)hipFree(l->wire_acc);
    delete l;
}

int smr_raft_leader_append(smr_raft_leader *l, const uint32_t *n_new_dev, void *stream) {
    if (!l || !n_new_dev) return fail(SMR_ERR_ARG, "raft: null argument");
    hipLaunchKernelGGL((l->v.R <= 5 ? raft_append_kernel<5> : raft_append_kernel<RMAX>), dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v, n_new_dev,
                       (uint32_t *)nullptr);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_leader_handle_replies(smr_raft_leader *l, const uint64_t *reply_term_dev, const uint32_t *end_slot_dev,
                                   const uint64_t *conflict_term_dev, const uint32_t *conflict_slot_dev,
                                   const uint8_t *flags_dev, const uint32_t *order_dev, void *stream) {
    if (!l || !reply_term_dev || !end_slot_dev || !flags_dev) return fail(SMR_ERR_ARG, "raft: null argument");
    const dim3 grid((l->v.G + 255) / 256), block(256);
    hipStream_t st = (hipStream_t)stream;
    const CraftView cv = l->craft ? l->cv : CraftView{};
#define RAFT_REPLIES(C, N) hipLaunchKernelGGL((raft_replies_kernel<C, N>), grid, block, 0, st, l->v, reply_term_dev, end_slot_dev, \
                                              conflict_term_dev, conflict_slot_dev, flags_dev, order_dev, cv)
    if (l->v.R <= 5) { if (l->craft) RAFT_REPLIES(true, 5); else RAFT_REPLIES(false, 5); }
    else { if (l->craft) RAFT_REPLIES(true, RMAX); else RAFT_REPLIES(false, RMAX); }
#undef RAFT_REPLIES
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_leader_handle_wire_replies(smr_raft_leader *l, const uint8_t *buf_dev, uint64_t buf_len, const uint64_t *conn_off_dev,
                                        const uint8_t *conn_len_dev, uint32_t n_conn, const uint32_t *order_dev, smr_wire_other *others_dev,
                                        uint64_t other_cap, uint64_t *counts_dev, uint64_t *consumed_dev, int32_t *status_dev, void *stream) {
    if (!l || !conn_off_dev || !counts_dev || !consumed_dev || !status_dev || (buf_len && !buf_dev) || (other_cap && !others_dev))
        return fail(SMR_ERR_ARG, "raft wire replies: null argument");
    if ((uintptr_t)buf_dev & 15) return fail(SMR_ERR_ARG, "raft wire replies: the byte buffer must be 16-byte aligned");
    const uint32_t F = l->v.R - 1u;
    if (n_conn != l->v.G * F)
        return fail(SMR_ERR_ARG, "raft wire replies: the connections come dense -- n_groups * (population - 1) of them, connection g * (population - 1) + k "
                                 "= group g's k-th follower in ascending id");
    const uint32_t GPB = RW_BLOCK / F;
    const RaftWire A{buf_dev, buf_len, conn_off_dev, conn_len_dev, n_conn, order_dev, others_dev, other_cap, counts_dev, consumed_dev, status_dev, l->wire_acc};
    const dim3 grid((l->v.G + GPB - 1) / GPB), block(RW_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    const CraftView cv = l->craft ? l->cv : CraftView{};
#define RAFT_WIRE(C, N) hipLaunchKernelGGL((raft_wire_replies_kernel<C, N>), grid, block, 0, st, l->v, cv, A)
    if (l->v.R <= 5) { if (l->craft) RAFT_WIRE(true, 5); else RAFT_WIRE(false, 5); }
    else { if (l->craft) RAFT_WIRE(true, RMAX); else RAFT_WIRE(false, RMAX); }
#undef RAFT_WIRE
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_leader_run_ticks(smr_raft_leader *l, const smr_raft_tick *ticks, uint32_t n_ticks, void *stream) {
    if (!l || (n_ticks && !ticks)) return fail(SMR_ERR_ARG, "raft: null argument");
    if (l->craft) return fail(SMR_ERR_STATE, "raft: batches of ticks are for plain Raft leaders (a CRaft leader's tick has its own heartbeat step)");
    for (uint32_t t = 0; t < n_ticks; t++)
        if (ticks[t].flags && (!ticks[t].reply_term || !ticks[t].end_slot)) return fail(SMR_ERR_ARG, "raft: a tick with replies needs reply_term and end_slot");
    const dim3 grid((l->v.G + 255) / 256), block(256);
    for (uint32_t t0 = 0; t0 < n_ticks; t0 += RAFT_MAX_BATCH) {
        RaftTickBatch b;
        memset(&b, 0, sizeof(b));
        b.n = n_ticks - t0 < (uint32_t)RAFT_MAX_BATCH ? n_ticks - t0 : (uint32_t)RAFT_MAX_BATCH;
        for (uint32_t k = 0; k < b.n; k++) {
            const smr_raft_tick &x = ticks[t0 + k];
            b.n_new[k] = x.n_new; b.reply_term[k] = x.reply_term; b.end_slot[k] = x.end_slot; b.conflict_term[k] = x.conflict_term;
            b.conflict_slot[k] = x.conflict_slot; b.flags[k] = x.flags; b.order[k] = x.order;
        }
        if (l->v.R <= 5) hipLaunchKernelGGL(raft_ticks_kernel<5>, grid, block, 0, (hipStream_t)stream, l->v, b);
        else hipLaunchKernelGGL(raft_ticks_kernel<RMAX>, grid, block, 0, (hipStream_t)stream, l->v, b);
        SMR_HIP_TRY(hipGetLastError());
    }
    return SMR_OK;
}

int smr_raft_craft_enable(smr_raft_leader *l, uint8_t fault_tolerance, uint8_t repeat_threshold) {
    if (!l) return fail(SMR_ERR_ARG, "craft: null argument");
    if (l->craft) return fail(SMR_ERR_ARG, "craft: already enabled");
    const uint32_t R = l->v.R, quorum = R / 2 + 1;
    if (fault_tolerance > R - quorum) return fail(SMR_ERR_ARG, "craft: fault_tolerance too large");   // craft/mod.rs:528-533
    const size_t G = l->v.G;
    const size_t n8 = (R * G * 8 + 255) & ~(size_t)255, n1 = (R * G + 255) & ~(size_t)255, ng = (G + 255) & ~(size_t)255;
    const size_t nm = ((size_t)l->v.W * G + 255) & ~(size_t)255;
    const size_t ng4 = (G * 4 + 255) & ~(size_t)255, nq4 = ((size_t)CRAFT_RQ * G * 4 + 255) & ~(size_t)255, nq8 = ((size_t)CRAFT_RQ * G * 8 + 255) & ~(size_t)255;
    const size_t total = 2 * n8 + n1 + 2 * ng + nm + ng + 2 * ng4 + nq4 + nq8;
    SMR_HIP_TRY(hipMalloc((void **)&l->craft_base, total));
    SMR_HIP_TRY(hipMemset(l->craft_base, 0, total));
    CraftView &cv = l->cv;
    uint8_t *b = l->craft_base;
    cv.hb_replied = (uint64_t *)b; b += n8; cv.hb_seen = (uint64_t *)b; b += n8;
    cv.hb_repeat = b; b += n1; cv.full_copy = b; b += ng; cv.alive = b; b += ng;
    l->v.entry_mask = b;                                       // the dummy entry 0 and whatever the log holds so far: every shard
    SMR_HIP_TRY(hipMemset(l->v.entry_mask, (int)((1u << R) - 1u), (size_t)l->v.W * G));
    b += nm;
    cv.rq_term = (uint64_t *)b; b += nq8; cv.rq_slot = (uint32_t *)b; b += nq4; cv.last_recon = (uint32_t *)b; b += ng4;
    cv.rq_n = (uint32_t *)b; b += ng4; cv.partial = b;
    cv.ft = fault_tolerance; cv.rep_thr = repeat_threshold; cv.quorum = quorum;
    std::vector<uint64_t> one(R * G, 1);                      // heartbeat.rs:117-119 reply_cnts start at (1, 0, 0)
    for (size_t g = 0; g < G; g++) one[(size_t)l->v.me * G + g] = 0;
    SMR_HIP_TRY(hipMemcpy(cv.hb_replied, one.data(), R * G * 8, hipMemcpyHostToDevice));
    SMR_HIP_TRY(hipMemset(cv.alive, (int)((1u << R) - 1u), G));   // heartbeat.rs:131
    l->v.thresh = quorum + fault_tolerance;
    l->craft = true;
    SMR_HIP_TRY(hipMemcpy(l->d_view, &l->v, sizeof(RaftView), hipMemcpyHostToDevice));   // (the threshold moved: the view's device copy follows)
    l->d_view_ok = true;
    return SMR_OK;
}

int smr_raft_craft_switch_assignment_mode(smr_raft_leader *l, const uint8_t *to_full_dev, void *stream) {
    if (!l || !to_full_dev) return fail(SMR_ERR_ARG, "craft: null argument");
    if (!l->craft) return fail(SMR_ERR_ARG, "craft: not enabled on this leader");
    hipLaunchKernelGGL(craft_mode_kernel, dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v, l->cv, to_full_dev,
                       (uint32_t *)nullptr, (uint32_t *)nullptr);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_craft_assignment(smr_raft_leader *l, uint32_t *persist_dev, uint32_t *send_dev, void *stream) {
    if (!l || !persist_dev || !send_dev) return fail(SMR_ERR_ARG, "craft: null argument");
    if (!l->craft) return fail(SMR_ERR_ARG, "craft: not enabled on this leader");
    hipLaunchKernelGGL(craft_mode_kernel, dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v, l->cv,
                       (const uint8_t *)nullptr, persist_dev, send_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_craft_bcast_heartbeats(smr_raft_leader *l, uint8_t *hb_flags_dev, uint32_t *prev_slot_dev, uint64_t *prev_term_dev,
                                    uint32_t *leader_commit_dev, uint32_t *last_snap_dev, void *stream) {
    if (!l || !hb_flags_dev || !prev_slot_dev || !prev_term_dev || !leader_commit_dev || !last_snap_dev)
        return fail(SMR_ERR_ARG, "craft: null argument");
    if (!l->craft) return fail(SMR_ERR_ARG, "craft: not enabled on this leader");
    hipLaunchKernelGGL(craft_heartbeat_kernel, dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v, l->cv,
                       hb_flags_dev, prev_slot_dev, prev_term_dev, leader_commit_dev, last_snap_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_craft_dump(smr_raft_leader *l, uint8_t *full_copy_host, uint8_t *alive_host, uint64_t *hb_replied_host,
                        uint64_t *hb_seen_host, uint8_t *hb_repeat_host) {
    if (!l || !full_copy_host || !alive_host || !hb_replied_host || !hb_seen_host || !hb_repeat_host)
        return fail(SMR_ERR_ARG, "craft: null argument");
    if (!l->craft) return fail(SMR_ERR_ARG, "craft: not enabled on this leader");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const size_t G = l->v.G, R = l->v.R;
    SMR_HIP_TRY(hipMemcpy(full_copy_host, l->cv.full_copy, G, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(alive_host, l->cv.alive, G, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(hb_replied_host, l->cv.hb_replied, R * G * 8, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(hb_seen_host, l->cv.hb_seen, R * G * 8, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(hb_repeat_host, l->cv.hb_repeat, R * G, hipMemcpyDeviceToHost));
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
    std::vector<uint32_t> rlo(G);
    D2H(et.data(), v.entry_term, W * G * 8);
    D2H(rlo.data(), v.ring_lo, G * 4);
#undef D2H
    for (size_t g = 0; g < G; g++) {
        hb->next_slot[(size_t)v.me * G + g] = 0; hb->try_next_slot[(size_t)v.me * G + g] = 0;
        hb->match_slot[(size_t)v.me * G + g] = 0;
        for (size_t w = 0; w < W; w++) hb->entry_term[w * G + g] = 0;
        uint32_t len = hb->log_len[g], lo = rlo[g] > hb->start_slot[g] ? rlo[g] : hb->start_slot[g];
        for (uint32_t s = lo; s < len; s++) hb->entry_term[(size_t)(s & (W - 1)) * G + g] = et[(size_t)(s & (W - 1)) * G + g];
    }
    return SMR_OK;
}

int smr_raft_leader_total_commits(smr_raft_leader *l, uint64_t *out) {
    if (!l || !out) return fail(SMR_ERR_ARG, "raft: null argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[4];
    SMR_HIP_TRY(ctr_read(l->v.counters, 4, h));
    *out = h[0];
    return SMR_OK;
}

int smr_raft_replica_preset(smr_raft_leader *l, uint8_t role, uint8_t leader, uint64_t term, uint8_t voted_for) {
    if (!l || role > ROLE_LEADER) return fail(SMR_ERR_ARG, "raft: bad argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const size_t G = l->v.G;
    std::vector<uint64_t> t(G, term);
    SMR_HIP_TRY(hipMemset(l->v.role, role, G));
    SMR_HIP_TRY(hipMemset(l->v.leader, leader, G));
    SMR_HIP_TRY(hipMemset(l->v.voted_for, voted_for, G));
    SMR_HIP_TRY(hipMemset(l->v.votes, 0, G));
    SMR_HIP_TRY(hipMemcpy(l->v.curr_term, t.data(), G * 8, hipMemcpyHostToDevice));
    return SMR_OK;
}

int smr_raft_replica_handle_append_entries(smr_raft_leader *l, const smr_raft_append_entries *m,
                                           const smr_raft_append_reply *r, void *stream) {
    if (!l || !m || !r || !m->flags || !m->leader || !m->term || !m->prev_slot || !m->prev_term || !m->n_entries ||
        !m->leader_commit || !m->last_snap || (m->max_entries && !m->entry_term) || !r->flags || !r->term ||
        !r->end_slot || !r->conflict_term || !r->conflict_slot)
        return fail(SMR_ERR_ARG, "raft: null argument");
    if (l->craft) {
        if (m->max_entries && !m->entry_mask) return fail(SMR_ERR_ARG, "craft: AppendEntries without the entries' shard bitmaps");
        hipLaunchKernelGGL(raft_append_entries_kernel<true>, dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v,
                           m->flags, m->leader, m->term, m->prev_slot, m->prev_term, m->n_entries, m->entry_term, m->entry_mask,
                           l->cv.partial, m->max_entries, m->leader_commit, m->last_snap, r->flags, r->term, r->end_slot, r->conflict_term,
                           r->conflict_slot);
    } else
        hipLaunchKernelGGL(raft_append_entries_kernel<false>, dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v,
                           m->flags, m->leader, m->term, m->prev_slot, m->prev_term, m->n_entries, m->entry_term, (const uint8_t *)nullptr,
                           (uint8_t *)nullptr, m->max_entries, m->leader_commit, m->last_snap, r->flags, r->term, r->end_slot, r->conflict_term,
                           r->conflict_slot);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_craft_handle_reconstruct(smr_raft_leader *l, const uint32_t *n_dev, const uint32_t *slot_dev, const uint64_t *term_dev,
                                      uint32_t max_slots, uint32_t *r_n_dev, uint8_t *r_has_dev, uint8_t *r_mask_dev, void *stream) {
    if (!l || !n_dev || !slot_dev || !term_dev || !r_n_dev || !r_has_dev || !r_mask_dev) return fail(SMR_ERR_ARG, "craft: null argument");
    if (!l->craft) return fail(SMR_ERR_ARG, "craft: not enabled on this replica");
    hipLaunchKernelGGL(craft_reconstruct_kernel, dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v, n_dev, slot_dev,
                       term_dev, max_slots, r_n_dev, r_has_dev, r_mask_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_craft_handle_reconstruct_reply(smr_raft_leader *l, const uint8_t *peer_dev, const uint32_t *n_dev, const uint32_t *slot_dev,
                                            const uint8_t *mask_dev, uint32_t max_slots, void *stream) {
    if (!l || !peer_dev || !n_dev || !slot_dev || !mask_dev) return fail(SMR_ERR_ARG, "craft: null argument");
    if (!l->craft) return fail(SMR_ERR_ARG, "craft: not enabled on this replica");
    hipLaunchKernelGGL(craft_reconstruct_reply_kernel, dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v, l->cv, peer_dev,
                       n_dev, slot_dev, mask_dev, max_slots);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_craft_poll_reconstructs(smr_raft_leader *l, uint32_t max_slots, uint32_t *n_dev, uint32_t *slot_dev, uint64_t *term_dev, void *stream) {
    if (!l || !n_dev || !slot_dev || !term_dev) return fail(SMR_ERR_ARG, "craft: null argument");
    if (!l->craft) return fail(SMR_ERR_ARG, "craft: not enabled on this replica");
    if (max_slots > CRAFT_RQ) max_slots = CRAFT_RQ;
    hipLaunchKernelGGL(craft_poll_reconstructs_kernel, dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v, l->cv, max_slots,
                       n_dev, slot_dev, term_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_ring_guard_hits(smr_raft_leader *l, uint64_t *out) {
    if (!l || !out) return fail(SMR_ERR_ARG, "raft: null argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    unsigned long long c[8];
    SMR_HIP_TRY(ctr_read(l->v.counters, 8, c));
    *out = c[6];
    return SMR_OK;
}

int smr_raft_craft_dump_masks(smr_raft_leader *l, uint8_t *mask_host, uint64_t *counters) {
    if (!l || !mask_host || !counters) return fail(SMR_ERR_ARG, "craft: null argument");
    if (!l->craft) return fail(SMR_ERR_ARG, "craft: not enabled on this replica");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const RaftView &v = l->v;
    const size_t G = v.G, W = v.W;
    std::vector<uint8_t> m(W * G);
    std::vector<uint32_t> len(G), start(G), rlo(G);
    SMR_HIP_TRY(hipMemcpy(m.data(), v.entry_mask, W * G, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(len.data(), v.log_len, G * 4, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(start.data(), v.start_slot, G * 4, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(rlo.data(), v.ring_lo, G * 4, hipMemcpyDeviceToHost));
    unsigned long long c[8];
    SMR_HIP_TRY(ctr_read(v.counters, 6, c));
    counters[0] = c[4]; counters[1] = c[5];
    memset(mask_host, 0, W * G);                               // canonical form: only the slots the log (and the ring) holds
    for (size_t g = 0; g < G; g++) {
        uint32_t end = len[g], lo = end > W ? end - (uint32_t)W : start[g];
        if (lo < rlo[g]) lo = rlo[g];
        for (uint32_t s2 = lo; s2 < end; s2++) mask_host[(size_t)(s2 & (W - 1)) * G + g] = m[(size_t)(s2 & (W - 1)) * G + g];
    }
    return SMR_OK;
}

int smr_raft_replica_become_candidate(smr_raft_leader *l, const uint8_t *timeout_src_dev, uint8_t *rv_flags_dev,
                                      uint64_t *rv_term_dev, uint32_t *rv_last_slot_dev, uint64_t *rv_last_term_dev,
                                      void *stream) {
    if (!l || !timeout_src_dev || !rv_flags_dev || !rv_term_dev || !rv_last_slot_dev || !rv_last_term_dev)
        return fail(SMR_ERR_ARG, "raft: null argument");
    hipLaunchKernelGGL(raft_become_candidate_kernel, dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v,
                       timeout_src_dev, rv_flags_dev, rv_term_dev, rv_last_slot_dev, rv_last_term_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_replica_handle_request_vote(smr_raft_leader *l, const uint8_t *flags_dev, const uint8_t *candidate_dev,
                                         const uint64_t *term_dev, const uint32_t *last_slot_dev,
                                         const uint64_t *last_term_dev, uint8_t *r_flags_dev, uint64_t *r_term_dev,
                                         void *stream) {
    if (!l || !flags_dev || !candidate_dev || !term_dev || !last_slot_dev || !last_term_dev || !r_flags_dev || !r_term_dev)
        return fail(SMR_ERR_ARG, "raft: null argument");
    hipLaunchKernelGGL(raft_request_vote_kernel, dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v,
                       flags_dev, candidate_dev, term_dev, last_slot_dev, last_term_dev, r_flags_dev, r_term_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_replica_handle_vote_replies(smr_raft_leader *l, const uint64_t *term_dev, const uint8_t *flags_dev,
                                         const uint32_t *order_dev, uint32_t *hb_prev_slot_dev, uint8_t *elected_dev,
                                         void *stream) {
    if (!l || !term_dev || !flags_dev || !hb_prev_slot_dev || !elected_dev) return fail(SMR_ERR_ARG, "raft: null argument");
    hipLaunchKernelGGL(raft_vote_replies_kernel, dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v,
                       term_dev, flags_dev, order_dev, hb_prev_slot_dev, elected_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_replica_dump_votes(smr_raft_leader *l, uint8_t *voted_for_host, uint8_t *votes_host, uint32_t *n_exec_host,
                                uint32_t *n_trunc_host) {
    if (!l || !voted_for_host || !votes_host) return fail(SMR_ERR_ARG, "raft: null argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const size_t G = l->v.G;
    SMR_HIP_TRY(hipMemcpy(voted_for_host, l->v.voted_for, G, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(votes_host, l->v.votes, G, hipMemcpyDeviceToHost));
    if (n_exec_host) SMR_HIP_TRY(hipMemcpy(n_exec_host, l->v.n_exec, G * 4, hipMemcpyDeviceToHost));
    if (n_trunc_host) SMR_HIP_TRY(hipMemcpy(n_trunc_host, l->v.n_trunc, G * 4, hipMemcpyDeviceToHost));
    return SMR_OK;
}

int smr_raft_leader_append_emit(smr_raft_leader *l, const uint32_t *n_new_dev, uint32_t *first_sent_dev, void *stream) {
    if (!l || !n_new_dev || !first_sent_dev) return fail(SMR_ERR_ARG, "raft: null argument");
    hipLaunchKernelGGL((l->v.R <= 5 ? raft_append_kernel<5> : raft_append_kernel<RMAX>), dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v, n_new_dev,
                       first_sent_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_leader_gather_entries(smr_raft_leader *l, const uint32_t *first_dev, const smr_raft_append_entries *m,
                                   void *stream) {
    if (!l || !first_dev || !m || !m->flags || !m->leader || !m->term || !m->prev_slot || !m->prev_term || !m->n_entries ||
        !m->leader_commit || !m->last_snap || (m->max_entries && !m->entry_term))
        return fail(SMR_ERR_ARG, "raft: null argument");
    hipLaunchKernelGGL(raft_gather_kernel, dim3((l->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, l->v, first_dev,
                       m->max_entries, (uint8_t *)m->flags, (uint8_t *)m->leader, (uint64_t *)m->term, (uint32_t *)m->prev_slot,
                       (uint64_t *)m->prev_term, (uint32_t *)m->n_entries, (uint64_t *)m->entry_term,
                       (uint32_t *)m->leader_commit, (uint32_t *)m->last_snap);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

static int raft_repl_setup(smr_raft_leader *leader, uint32_t n, smr_raft_leader *const *followers, const uint32_t *const *first_dev,
                           const smr_raft_append_entries *msgs, const smr_raft_append_reply *replies, RaftRepl &A) {
    if (!leader || !followers || !first_dev || !msgs || !replies) return fail(SMR_ERR_ARG, "raft replicate: null argument");
    if (n == 0 || n > RMAX) return fail(SMR_ERR_ARG, "raft replicate: 1 .. 8 followers");
    memset(&A, 0, sizeof(A));
    A.K = msgs[0].max_entries;
    for (uint32_t k = 0; k < n; k++) {
        smr_raft_leader *f = followers[k];
        const smr_raft_append_entries &m = msgs[k];
        const smr_raft_append_reply &r = replies[k];
        if (!f || f == leader) return fail(SMR_ERR_ARG, "raft replicate: a follower is null or the leader itself");
        for (uint32_t j = 0; j < k; j++) if (followers[j] == f) return fail(SMR_ERR_ARG, "raft replicate: a follower is listed twice");
        if (f->v.G != leader->v.G || f->v.W != leader->v.W || f->v.R != leader->v.R || f->craft != leader->craft)
            return fail(SMR_ERR_ARG, "raft replicate: a follower differs from the leader in groups / window / population / variant");
        if (!first_dev[k] || !m.flags || !m.leader || !m.term || !m.prev_slot || !m.prev_term || !m.n_entries || !m.leader_commit || !m.last_snap ||
            (m.max_entries && !m.entry_term) || !r.flags || !r.term || !r.end_slot || !r.conflict_term || !r.conflict_slot)
            return fail(SMR_ERR_ARG, "raft replicate: null argument");
        if (m.max_entries != A.K) return fail(SMR_ERR_ARG, "raft replicate: the messages differ in max_entries");
        if (f->craft && m.max_entries && !m.entry_mask) return fail(SMR_ERR_ARG, "craft: AppendEntries without the entries' shard bitmaps");
        if (!f->d_view || !f->d_view_ok) return fail(SMR_ERR_STATE, "raft replicate: a follower has no device copy of its view");
        // blocks of different blockIdx.y write follower k's message and reply arrays: two followers must not share them
        for (uint32_t j = 0; j < k; j++)
            if (msgs[j].flags == m.flags || msgs[j].n_entries == m.n_entries || msgs[j].term == m.term || replies[j].flags == r.flags ||
                replies[j].end_slot == r.end_slot || replies[j].term == r.term)
                return fail(SMR_ERR_ARG, "raft replicate: two followers share a message or reply buffer");
        if (f->device != leader->device)                         // ... and the follower's state must be on the device the launch runs on
            return fail(SMR_ERR_ARG, "raft replicate: a follower lives on another device than the leader");
        A.fv[k] = f->d_view; A.partial[k] = f->craft ? f->cv.partial : nullptr; A.first[k] = first_dev[k];
        A.m_flags[k] = (uint8_t *)m.flags; A.m_leader[k] = (uint8_t *)m.leader; A.m_term[k] = (uint64_t *)m.term;
        A.m_prev_slot[k] = (uint32_t *)m.prev_slot; A.m_prev_term[k] = (uint64_t *)m.prev_term; A.m_n[k] = (uint32_t *)m.n_entries;
        A.m_eterm[k] = (uint64_t *)m.entry_term; A.m_lc[k] = (uint32_t *)m.leader_commit; A.m_ls[k] = (uint32_t *)m.last_snap;
        A.m_emask[k] = f->craft ? m.entry_mask : nullptr;
        A.r_flags[k] = r.flags; A.r_term[k] = r.term; A.r_end[k] = r.end_slot; A.r_cterm[k] = r.conflict_term; A.r_cslot[k] = r.conflict_slot;
    }
    return SMR_OK;
}

int smr_raft_cluster_replicate(smr_raft_leader *leader, uint32_t n, smr_raft_leader *const *followers, const uint32_t *const *first_dev,
                               const smr_raft_append_entries *msgs, const smr_raft_append_reply *replies, void *stream) {
    RaftRepl A;
    if (int rc = raft_repl_setup(leader, n, followers, first_dev, msgs, replies, A)) return rc;
    const dim3 grid((leader->v.G + 255) / 256, n), block(256);
    if (leader->craft) hipLaunchKernelGGL(raft_replicate_kernel<true>, grid, block, 0, (hipStream_t)stream, leader->v, A);
    else hipLaunchKernelGGL(raft_replicate_kernel<false>, grid, block, 0, (hipStream_t)stream, leader->v, A);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_raft_cluster_tick(smr_raft_leader *leader, const uint32_t *n_new_dev, uint32_t *first_sent_dev, uint32_t n, smr_raft_leader *const *followers,
                          const uint32_t *const *first_dev, const smr_raft_append_entries *msgs, const smr_raft_append_reply *replies,
                          const uint64_t *reply_term_dev, const uint32_t *end_slot_dev, const uint64_t *conflict_term_dev,
                          const uint32_t *conflict_slot_dev, const uint8_t *flags_dev, const uint32_t *order_dev, void *stream) {
    if (!leader || !n_new_dev || !first_sent_dev || !reply_term_dev || !end_slot_dev || !flags_dev) return fail(SMR_ERR_ARG, "raft cluster tick: null argument");
    RaftRepl A;
    if (int rc = raft_repl_setup(leader, n, followers, first_dev, msgs, replies, A)) return rc;
    const RaftTickRest T{n_new_dev, first_sent_dev, reply_term_dev, conflict_term_dev, end_slot_dev, conflict_slot_dev, order_dev, flags_dev, n};
    const CraftView cv = leader->craft ? leader->cv : CraftView{};
    const dim3 grid((leader->v.G + 63) / 64), block(64 * (n + 1));
    hipStream_t st = (hipStream_t)stream;
#define RAFT_TICK(C, N) hipLaunchKernelGGL((raft_cluster_tick_kernel<C, N>), grid, block, 0, st, leader->v, cv, A, T)
    if (leader->v.R <= 5) { if (leader->craft) RAFT_TICK(true, 5); else RAFT_TICK(false, 5); }
    else { if (leader->craft) RAFT_TICK(true, RMAX); else RAFT_TICK(false, RMAX); }
#undef RAFT_TICK
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

}  // extern "C"

// RSPaxos replica over G groups (lane = group), one replica id per object: leader append with one
// shard per peer, follower accept, the accept tally with threshold majority + fault_tolerance, the
// commit-bar run gated on shard availability, leader change (Prepare phase with shard merging,
// re-Accept), reconstruction reads, heartbeat commit learning, the exec bar.
//
// Stands in for RSPaxosReplica::{handle_req_batch (rspaxos/request.rs:10-151), handle_msg_prepare
// (messages.rs:12-84), handle_msg_prepare_reply (:87-340), handle_msg_accept (:343-403),
// handle_msg_accept_reply (:406-464), handle_msg_reconstruct (:467-515),
// handle_msg_reconstruct_reply (:518-594), handle_logged_* (durability.rs:10-186),
// handle_cmd_result (execution.rs:10-65), check_leader / become_a_leader / bcast_heartbeats /
// heard_heartbeat (leadership.rs:11-340)}.  A request batch is an opaque token, a codeword is
// (token, mask of the shards present) -- rscoding.rs:255-346 subset_copy / absorb_other on masks;
// the shard BYTES are rs_kernels.hip's business.  WAL appends and commands complete right after
// the handler that submitted them returns (LS-1 rule 0): WAL completions are inlined where that is
// the same thing (see each handler), command results wait in a per-group list until the handler
// is done.  First version: one lane per group, every access coalesced over groups, no tuning yet.
//
// Layout (group fastest): per-slot fields X[(slot & (W-1)) * G + g] over a ring of W slots; an
// instance that left the ring is ignored like a slot below start_slot (harness guard).
#include <string.h>

#include <vector>

#include "smr_common.h"
#include "rsp_peek.h"

namespace smr {

constexpr uint32_t RSP_NULL = 0xFFFFFFFFu;                           // from_null(): no data length yet
constexpr uint8_t RSP_NO_REP = 0xFF;
constexpr uint64_t RSP_DG_MUL = 0x100000001B3ull;
enum { RST_NULL = 0, RST_PREPARING = 1, RST_ACCEPTING = 2, RST_COMMITTED = 3, RST_EXECUTED = 4 };
enum { RFL_LBK = 1, RFL_RBK = 2, RFL_EXT = 4 };

struct RspView {
    uint32_t G, W, Wmask, R, me, majority, ft;
    uint8_t *leader;
    uint64_t *bps, *bpd, *bms;           // bal_prep_sent, bal_prepared, bal_max_seen [G]
    uint32_t *len, *cbar, *ebar, *snap;  // [G]
    uint32_t *peb;                       // peer_exec_bar [R][G]
    uint64_t *digest;                    // [G]
    uint64_t *s_bal, *s_vbal, *s_pmax;   // [W][G]
    uint8_t *s_st, *s_mask, *s_vmask, *s_fl, *s_packs, *s_aacks, *s_rsrc;
    uint32_t *s_val, *s_vval, *s_ltrig, *s_lendp, *s_rtrig, *s_rendp;
    uint32_t *xq;                        // [W][G] slots whose command the call in flight submitted, in order ...
    uint32_t *xn;                        // ... [G] how many: what smr_rsp_exec_poll reads
    unsigned long long *counters;        // commits, commands executed, mixed absorbs, redirects
};

struct RspLane {
    const RspView &v;
    const uint32_t g;
    uint32_t leader, len, cbar, ebar, snap;
    uint64_t bps, bpd, bms;
    uint32_t n_xq = 0, x_lo = 0;          // submitted so far in this call / of them already executed
    unsigned int c_commit = 0, c_exec = 0, c_mixed = 0, c_redirect = 0;
    __device__ __forceinline__ RspLane(const RspView &v_, uint32_t g_) : v(v_), g(g_) {
        leader = v.leader[g]; len = v.len[g]; cbar = v.cbar[g]; ebar = v.ebar[g]; snap = v.snap[g];
        bps = v.bps[g]; bpd = v.bpd[g]; bms = v.bms[g];
    }
    __device__ __forceinline__ void store() {
        v.leader[g] = (uint8_t)leader; v.len[g] = len; v.cbar[g] = cbar; v.ebar[g] = ebar; v.snap[g] = snap;
        v.bps[g] = bps; v.bpd[g] = bpd; v.bms[g] = bms;
        v.xn[g] = n_xq;
    }
    __device__ __forceinline__ size_t ix(uint32_t slot) const { return (size_t)(slot & v.Wmask) * v.G + g; }
    __device__ __forceinline__ bool held(uint32_t slot) const { return slot < len && slot + v.W >= len; }
    __device__ __forceinline__ uint32_t ring_lo() const { return len > v.W ? len - v.W : 0u; }
    __device__ __forceinline__ bool is_leader() const { return leader == v.me; }
    __device__ __forceinline__ uint32_t data_mask() const { return (1u << v.majority) - 1u; }
    __device__ __forceinline__ uint32_t all_mask() const { return (1u << v.R) - 1u; }
    __device__ __forceinline__ void push_null() {                        // mod.rs:411-431
        const size_t i = ix(len);
        v.s_bal[i] = 0; v.s_st[i] = RST_NULL; v.s_val[i] = RSP_NULL; v.s_mask[i] = 0;
        v.s_vbal[i] = 0; v.s_vval[i] = RSP_NULL; v.s_vmask[i] = 0; v.s_fl[i] = 0;
        v.s_pmax[i] = 0;       // read by handle_msg_prepare_reply even without leader bookkeeping (the reference would panic there)
        len++;
    }
    // while (len <= slot) push(null); only the last W pushes leave anything behind
    __device__ __forceinline__ void pad_to(uint32_t slot) {
        if (slot + 1 > len + v.W) len = slot + 1 - v.W;
        while (len <= slot) push_null();
    }
    // rscoding.rs:296-346 on (token, mask)
    __device__ __forceinline__ void absorb(size_t i, uint32_t oval, uint32_t omask) {
        const uint32_t sval = v.s_val[i];
        if (sval != RSP_NULL && oval == RSP_NULL) { c_mixed++; return; }  // data_len mismatch: Err
        if (sval == RSP_NULL) v.s_val[i] = oval;
        else if (sval != oval) c_mixed++;
        v.s_mask[i] = (uint8_t)(v.s_mask[i] | omask);
    }
    __device__ __forceinline__ void check_leader(uint32_t peer, uint64_t ballot) {   // leadership.rs:11-42
        if (ballot > bms) { bms = ballot; leader = peer; }
    }
    // the commit-bar run of durability.rs:140-181 (check_status) / messages.rs:547-590
    __device__ __forceinline__ void commit_bar_run(bool check_status) {
        while (cbar < len && held(cbar)) {
            const size_t i = ix(cbar);
            const uint32_t st = v.s_st[i];
            if (st < RST_COMMITTED) break;
            uint32_t mask = v.s_mask[i];
            if ((uint32_t)__popc(mask) < v.majority) break;             // cannot execute without the whole batch
            if ((uint32_t)__popc(mask & data_mask()) < v.majority) v.s_mask[i] = (uint8_t)(mask | data_mask());   // reconstruct_data
            if (v.s_val[i] == 0) v.s_st[i] = RST_EXECUTED;               // reqs.is_empty()
            else if (!check_status || st == RST_COMMITTED) v.xq[(size_t)(n_xq++ & v.Wmask) * v.G + g] = cbar;
            cbar++;
        }
    }
    // execution.rs:10-65 for the commands the handler submitted, in order (rule 0)
    __device__ __forceinline__ void drain_exec() {
        if (x_lo == n_xq) return;
        uint64_t dg = v.digest[g];
        for (uint32_t k = x_lo; k < n_xq; k++) {
            const uint32_t slot = v.xq[(size_t)(k & v.Wmask) * v.G + g];
            if (!held(slot)) continue;
            const size_t i = ix(slot);
            dg = (dg ^ (((uint64_t)slot << 32) | v.s_val[i])) * RSP_DG_MUL;
            c_exec++;
            v.s_st[i] = RST_EXECUTED;
            if (slot == ebar)
                while (ebar < len && held(ebar) && v.s_st[ix(ebar)] >= RST_EXECUTED) ebar++;
        }
        v.digest[g] = dg;
        x_lo = n_xq;                                                       // the list stays: the host polls it after the call
    }
    // messages.rs:406-464 + the CommitSlot completion (durability.rs:125-186) + the command results
    __device__ __forceinline__ void accept_reply(uint32_t peer, uint32_t slot, uint64_t ballot) {
        if (!held(slot)) return;
        if (ballot != bpd) return;
        const size_t i = ix(slot);
        if (!is_leader() || v.s_st[i] != RST_ACCEPTING || ballot < v.s_bal[i]) return;
        if (!(v.s_fl[i] & RFL_LBK)) return;
        uint32_t acks = v.s_aacks[i];
        if ((acks >> peer) & 1u) return;
        acks |= 1u << peer;
        v.s_aacks[i] = (uint8_t)acks;
        if ((uint32_t)__popc(acks) >= v.majority + v.ft) {               // :437-440
            v.s_st[i] = RST_COMMITTED;
            c_commit++;
            if (slot == cbar) commit_bar_run(true);
            drain_exec();
        }
    }
    __device__ __forceinline__ void set_lbk(size_t i, uint32_t trig, uint32_t endp) {
        v.s_fl[i] = (uint8_t)(v.s_fl[i] | RFL_LBK);
        v.s_ltrig[i] = trig; v.s_lendp[i] = endp; v.s_packs[i] = 0; v.s_pmax[i] = 0; v.s_aacks[i] = 0;
    }
    // messages.rs:87-340; Accepts it releases are appended to (a_slot, a_val)[n_acc++]
    __device__ __forceinline__ void prepare_reply(uint32_t peer, uint32_t slot, uint32_t trig, uint32_t endp, uint64_t ballot,
                                                  bool has_voted, uint64_t vbal, uint32_t vval, uint32_t vmask, uint32_t *a_slot,
                                                  uint32_t *a_val, uint32_t &n_acc) {
        if (ballot != bps) return;                                       // :109
        if (!is_leader()) return;
        if (!held(trig) || !(v.s_fl[ix(trig)] & RFL_LBK)) return;        // :119-125
        const uint32_t my_endp = v.s_lendp[ix(trig)];
        if (len <= slot) {                                               // :131-168 slots I did not know of
            if (slot + 1 > len + v.W) len = slot + 1 - v.W;
            while (len <= slot) {
                const size_t i = ix(len);
                push_null();
                v.s_fl[i] = RFL_EXT; v.s_bal[i] = bps; v.s_st[i] = RST_PREPARING;
                set_lbk(i, trig, my_endp);                               // (its PrepareBal completion: slot > endprep, nothing)
            }
        }
        if (!held(slot) || !held(trig)) return;
        {
            const size_t i = ix(slot);
            if (v.s_st[i] != RST_PREPARING || ballot < v.s_bal[i]) return;   // :173-176
            if (has_voted) {                                             // :180-194
                const uint64_t pm = v.s_pmax[i];
                if (vbal > pm) { v.s_pmax[i] = vbal; v.s_val[i] = vval; v.s_mask[i] = (uint8_t)vmask; }
                else if (vbal == pm) absorb(i, vval, vmask);
            }
        }
        if (slot != endp) return;                                        // :200-338
        const size_t ti = ix(trig);
        const uint32_t pa = v.s_packs[ti] | (1u << peer);
        v.s_packs[ti] = (uint8_t)pa;
        const uint32_t cnt = __popc(pa);
        if (cnt < v.majority) return;
        bpd = ballot;
        for (uint32_t s = trig > ring_lo() ? trig : ring_lo(); s < len; s++) {
            const size_t i = ix(s);
            if (v.s_st[i] != RST_PREPARING) continue;
            uint32_t mask = v.s_mask[i], val = v.s_val[i];
            if ((uint32_t)__popc(mask) >= v.majority) {
                if ((uint32_t)__popc(mask & data_mask()) < v.majority) mask |= data_mask();   // reconstruct_data
            } else if (cnt >= v.R - v.ft) {
                val = 0; mask = data_mask();                             // from_data(ReqBatch::new())
            } else continue;
            if ((uint32_t)__popc(mask) < v.R) mask = all_mask();         // compute_parity
            v.s_val[i] = val; v.s_mask[i] = (uint8_t)mask;
            v.s_st[i] = RST_ACCEPTING;
            v.s_vbal[i] = ballot; v.s_vval[i] = val; v.s_vmask[i] = (uint8_t)(mask & (1u << v.me));
            if (a_slot) { a_slot[(size_t)(n_acc & v.Wmask) * v.G + g] = s; a_val[(size_t)(n_acc & v.Wmask) * v.G + g] = val; }
            n_acc++;
            accept_reply(v.me, s, v.s_bal[i]);                           // my own AcceptData completion carries inst.bal (durability.rs:100-104),
                                                                         // which the quorum step does not raise to `ballot`
        }
    }
    // leadership.rs:236-340; true = my Heartbeat goes back to `peer`
    __device__ __forceinline__ bool heard_heartbeat(uint32_t peer, uint64_t ballot, uint32_t hb_commit, uint32_t hb_exec, uint32_t hb_snap) {
        bool reply = false;
        if (peer != v.me) {
            check_leader(peer, ballot);
            reply = leader == peer;
        }
        if (ballot < bms) return reply;
        if (hb_exec < ebar) return reply;
        if (hb_commit > cbar) {
            if (len < hb_commit) pad_to(hb_commit - 1);
            bool at_bar = false;                                         // a CommitSlot completion that finds slot == commit_bar
            const uint32_t cb0 = cbar;
            for (uint32_t s = cb0; s < hb_commit; s++) {
                if (!held(s)) continue;                                  // harness: left the ring
                const size_t i = ix(s);
                const uint32_t st = v.s_st[i];
                if (v.s_bal[i] < ballot || st < RST_ACCEPTING) break;
                if (st >= RST_COMMITTED) continue;
                v.s_st[i] = RST_COMMITTED;
                if (s == cb0) at_bar = true;
            }
            if (at_bar) commit_bar_run(true);                            // later completions find the bar moved or stuck: no-ops
            drain_exec();
        }
        if (peer != v.me) {
            const size_t o = (size_t)peer * v.G + g;
            if (hb_exec > v.peb[o]) {
                v.peb[o] = hb_exec;
                uint32_t passed = 1;
                for (uint32_t p = 0; p < v.R; p++) if (p != v.me && v.peb[(size_t)p * v.G + g] >= hb_exec) passed++;
                if (passed == v.R) snap = hb_exec;
            }
            if (hb_snap > snap) snap = hb_snap;
        }
        return reply;
    }
    __device__ __forceinline__ void flush() {
        unsigned int c[4] = {c_commit, c_exec, c_mixed, c_redirect};
        for (int k = 0; k < 4; k++) {
            unsigned int x = c[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if (__lane_id() == 0 && x) ctr_add(v.counters, k, (unsigned long long)x);
        }
    }
};

#define RSP_LANE_BEGIN                                         \
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;         \
    const bool on = g < v.G;                                   \
    RspLane L(v, on ? g : 0);                                  \
    if (on) {
#define RSP_LANE_END \
        L.store();   \
    }                \
    L.flush();

// request.rs:10-151 + my AcceptData completion on one lane: the batch token x (RSP_NULL: none); n = 1 with (slot, ab) = the Accept to send
__device__ __forceinline__ void rsp_req_batch_lane(RspLane &L, uint32_t x, uint32_t &n, uint32_t &slot, uint64_t &ab) {
    const RspView &v = L.v;
    n = 0; ab = 0; slot = 0;
    if (x == RSP_NULL) return;
    if (!L.is_leader() || L.bpd == 0) { L.c_redirect++; return; }       // :19-42
    slot = RSP_NULL;                                                     // mod.rs:434-442
    for (uint32_t s = L.ebar > L.ring_lo() ? L.ebar : L.ring_lo(); s < L.len; s++)
        if (v.s_st[L.ix(s)] == RST_NULL) { slot = s; break; }
    if (slot == RSP_NULL) { L.push_null(); slot = L.len - 1; }
    const size_t i = L.ix(slot);
    v.s_val[i] = x; v.s_mask[i] = (uint8_t)L.all_mask();                 // from_data + compute_parity
    v.s_fl[i] = (uint8_t)(v.s_fl[i] | RFL_EXT);
    L.set_lbk(i, 0, 0);
    v.s_bal[i] = L.bpd; v.s_st[i] = RST_ACCEPTING;
    v.s_vbal[i] = L.bpd; v.s_vval[i] = x; v.s_vmask[i] = (uint8_t)(1u << v.me);
    n = 1; ab = L.bpd;
    L.accept_reply(v.me, slot, L.bpd);                                   // durability.rs:100-104
}

// messages.rs:343-403 + the AcceptData completion (durability.rs:85-122) on one lane; (rb, rs) = the AcceptReply (rb = 0: none)
__device__ __forceinline__ void rsp_accept_lane(RspLane &L, bool on, uint32_t peer, uint32_t s, uint64_t b, uint32_t val, uint32_t mask, uint64_t &rb,
                                                uint32_t &rs) {
    const RspView &v = L.v;
    rb = 0; rs = 0;
    if (!(on && !(s < L.len && !L.held(s)) && b >= L.bms)) return;
    L.check_leader(peer, b);
    L.pad_to(s);
    const size_t i = L.ix(s);
    v.s_bal[i] = b; v.s_st[i] = RST_ACCEPTING;
    v.s_val[i] = val; v.s_mask[i] = (uint8_t)mask;
    v.s_fl[i] = (uint8_t)(v.s_fl[i] | RFL_RBK); v.s_rsrc[i] = (uint8_t)peer; v.s_rtrig[i] = 0; v.s_rendp[i] = 0;
    v.s_vbal[i] = b; v.s_vval[i] = val; v.s_vmask[i] = (uint8_t)mask;
    if (L.is_leader()) L.accept_reply(v.me, s, b);
    else { rb = b; rs = s; }
}

__global__ __launch_bounds__(256) void rsp_req_batch_kernel(const RspView v, const uint32_t *__restrict__ val, uint32_t *__restrict__ a_n,
                                                            uint32_t *__restrict__ a_slot, uint32_t *__restrict__ a_val,
                                                            uint64_t *__restrict__ a_ballot) {
    RSP_LANE_BEGIN
    uint32_t n, slot; uint64_t ab;
    const uint32_t x = val[g];
    rsp_req_batch_lane(L, x, n, slot, ab);
    if (n) { a_slot[g] = slot; a_val[g] = x; }
    a_n[g] = n; a_ballot[g] = ab;
    RSP_LANE_END
}

// messages.rs:343-403 + the AcceptData completion (durability.rs:85-122)
__global__ __launch_bounds__(256) void rsp_accept_kernel(const RspView v, const uint8_t *__restrict__ flags, const uint8_t *__restrict__ peer,
                                                         const uint32_t *__restrict__ slot, const uint64_t *__restrict__ ballot,
                                                         const uint32_t *__restrict__ val, const uint8_t *__restrict__ mask,
                                                         uint64_t *__restrict__ r_ballot, uint32_t *__restrict__ r_slot) {
    RSP_LANE_BEGIN
    uint64_t rb; uint32_t rs;
    rsp_accept_lane(L, flags[g] & 1, peer[g], slot[g], ballot[g], val[g], mask[g], rb, rs);
    r_ballot[g] = rb; r_slot[g] = rs;
    RSP_LANE_END
}

__global__ __launch_bounds__(256) void rsp_accept_replies_kernel(const RspView v, const uint32_t *__restrict__ slot,
                                                                 const uint64_t *__restrict__ ballot, const uint8_t *__restrict__ flags,
                                                                 const uint32_t *__restrict__ order, uint8_t *__restrict__ committed) {
    RSP_LANE_BEGIN
    const uint32_t s = slot[g], ctl = order ? order[g] : SMR_CTL_IDENTITY;
    const uint32_t before = L.held(s) ? v.s_st[L.ix(s)] : 0u;
    for (uint32_t oi = 0; oi < v.R; oi++) {
        const uint32_t p = (ctl >> (3 * oi)) & 7u;
        if (p == v.me || p >= v.R) continue;
        const size_t o = (size_t)p * v.G + g;
        if (!(flags[o] & 1)) continue;
        L.accept_reply(p, s, ballot[o]);
    }
    committed[g] = (before == RST_ACCEPTING && L.held(s) && v.s_st[L.ix(s)] >= RST_COMMITTED) ? 1 : 0;
    RSP_LANE_END
}

// leadership.rs:47-185 on HearTimeout
__global__ __launch_bounds__(256) void rsp_become_leader_kernel(const RspView v, const uint8_t *__restrict__ src, uint8_t *__restrict__ hb_flags,
                                                                uint64_t *__restrict__ hb_ballot, uint32_t *__restrict__ hb_commit,
                                                                uint32_t *__restrict__ hb_exec, uint32_t *__restrict__ hb_snap,
                                                                uint8_t *__restrict__ p_flags, uint32_t *__restrict__ p_trig,
                                                                uint64_t *__restrict__ p_ballot, uint32_t *__restrict__ rc_n,
                                                                uint32_t *__restrict__ rc_slot) {
    RSP_LANE_BEGIN
    uint8_t hf = 0, pf = 0; uint64_t hb = 0, pb = 0; uint32_t hc = 0, he = 0, hs = 0, pt = 0, rn = 0;
    const uint32_t sr = src[g];
    if (sr != RSP_NO_REP && !(L.leader != RSP_NO_REP && L.leader != sr)) {   // :51-55
        L.leader = v.me;
        hf = 1; hb = L.bms; hc = L.cbar; he = L.ebar; hs = L.snap;          // :64 bcast_heartbeats
        (void)L.heard_heartbeat(v.me, L.bms, L.cbar, L.ebar, L.snap);
        for (uint32_t p = 0; p < v.R; p++) v.peb[(size_t)p * v.G + g] = 0;
        L.bpd = 0;                                                       // :72-74
        L.bps = (((L.bms >> 8) + 1) << 8) | (uint64_t)(v.me + 1);
        L.bms = L.bps;
        const uint32_t lo = L.ring_lo();                                 // :77-92
        uint32_t trig = L.len, endp = L.len;
        for (uint32_t s = lo; s < L.len; s++) if (v.s_st[L.ix(s)] < RST_COMMITTED) { trig = s; break; }
        for (uint32_t s = L.len; s > lo; s--) if (v.s_st[L.ix(s - 1)] < RST_COMMITTED) { endp = s - 1; break; }
        if (trig == L.len) L.push_null();
        const uint32_t from = L.ebar > L.ring_lo() ? L.ebar : L.ring_lo();
        for (uint32_t s = from; s < L.len; s++) {                        // :100-149
            const size_t i = L.ix(s);
            uint32_t st = v.s_st[i];
            if (st == RST_EXECUTED) continue;
            v.s_fl[i] = (uint8_t)(v.s_fl[i] | RFL_EXT);
            if (st < RST_COMMITTED) {
                v.s_bal[i] = L.bps; v.s_st[i] = RST_PREPARING;
                L.set_lbk(i, trig, endp);
            } else if ((uint32_t)__popc(v.s_mask[i]) < v.majority && rn < v.W)
                rc_slot[(size_t)(rn++) * v.G + g] = s;
        }
        // the PrepareBal completions, in order: a Prepare reply from myself (durability.rs:27-46)
        uint32_t na = 0;
        for (uint32_t s = from; s < L.len; s++) {
            const size_t i = L.ix(s);
            if (v.s_st[i] != RST_PREPARING || !(v.s_fl[i] & RFL_LBK) || s > v.s_lendp[i]) continue;
            const uint64_t vb = v.s_vbal[i];
            L.prepare_reply(v.me, s, v.s_ltrig[i], v.s_lendp[i], v.s_bal[i], vb > 0, vb, v.s_vval[i], v.s_vmask[i], nullptr, nullptr, na);
        }
        pf = 1; pt = trig; pb = L.bps;
    }
    hb_flags[g] = hf; hb_ballot[g] = hb; hb_commit[g] = hc; hb_exec[g] = he; hb_snap[g] = hs;
    p_flags[g] = pf; p_trig[g] = pt; p_ballot[g] = pb; rc_n[g] = rn;
    RSP_LANE_END
}

// messages.rs:12-84 + the PrepareBal completions on a follower (durability.rs:47-79): the reply batch
__global__ __launch_bounds__(256) void rsp_prepare_kernel(const RspView v, const uint8_t *__restrict__ flags, const uint8_t *__restrict__ peer,
                                                          const uint32_t *__restrict__ trig, const uint64_t *__restrict__ ballot,
                                                          uint32_t *__restrict__ pr_n, uint32_t *__restrict__ pr_trig,
                                                          uint32_t *__restrict__ pr_endp, uint64_t *__restrict__ pr_ballot,
                                                          uint64_t *__restrict__ pr_vbal, uint32_t *__restrict__ pr_vval,
                                                          uint8_t *__restrict__ pr_vmask) {
    RSP_LANE_BEGIN
    uint32_t n = 0, o_trig = 0, o_endp = 0; uint64_t o_bal = 0;
    const uint32_t t = trig[g];
    const uint64_t b = ballot[g];
    if ((flags[g] & 1) && !(t < L.len && !L.held(t)) && b >= L.bms) {
        L.check_leader(peer[g], b);
        L.pad_to(t);
        uint32_t last = 0;                                               // :40-48
        for (uint32_t s = L.len; s > L.ring_lo(); s--) if (v.s_st[L.ix(s - 1)] > RST_NULL) { last = s - 1; break; }
        const uint32_t endp = last > t ? last : t;
        const bool follower = !L.is_leader();
        for (uint32_t s = t; s <= endp; s++) {
            if (!L.held(s)) continue;
            const size_t i = L.ix(s);
            v.s_bal[i] = b; v.s_st[i] = RST_PREPARING;
            v.s_fl[i] = (uint8_t)(v.s_fl[i] | RFL_RBK); v.s_rsrc[i] = peer[g]; v.s_rtrig[i] = t; v.s_rendp[i] = endp;
        }
        // the PrepareBal completions come once every slot above is Preparing (rule 0), in slot order: on a follower
        // one row of the reply each; on a replica that still leads (a Prepare carrying its own ballot) a Prepare
        // reply from itself, whose quorum step may turn LATER slots of this very range back to Accepting
        for (uint32_t s = t; s <= endp; s++) {
            if (!L.held(s)) continue;
            const size_t i = L.ix(s);
            if (follower) {
                const uint32_t k = s - t;
                if (n == 0) { o_trig = t; o_endp = endp; o_bal = v.s_bal[i]; }
                if (k < v.W) {
                    const uint64_t vb = v.s_vbal[i];
                    const size_t o = (size_t)k * v.G + g;
                    pr_vbal[o] = vb; pr_vval[o] = vb > 0 ? v.s_vval[i] : RSP_NULL; pr_vmask[o] = vb > 0 ? v.s_vmask[i] : (uint8_t)0;
                    if (k + 1 > n) n = k + 1;
                }
            } else if ((v.s_fl[i] & RFL_LBK) && s <= v.s_lendp[i]) {
                uint32_t na = 0;
                const uint64_t vb = v.s_vbal[i];
                L.prepare_reply(v.me, s, v.s_ltrig[i], v.s_lendp[i], v.s_bal[i], vb > 0, vb, v.s_vval[i], v.s_vmask[i], nullptr, nullptr, na);
            }
        }
    }
    pr_n[g] = n; pr_trig[g] = o_trig; pr_endp[g] = o_endp; pr_ballot[g] = o_bal;
    RSP_LANE_END
}

// one peer's PrepareReply batch, slot by slot
__global__ __launch_bounds__(256) void rsp_prepare_replies_kernel(const RspView v, const uint8_t *__restrict__ peer,
                                                                  const uint32_t *__restrict__ pr_n, const uint32_t *__restrict__ pr_trig,
                                                                  const uint32_t *__restrict__ pr_endp, const uint64_t *__restrict__ pr_ballot,
                                                                  const uint64_t *__restrict__ pr_vbal, const uint32_t *__restrict__ pr_vval,
                                                                  const uint8_t *__restrict__ pr_vmask, uint32_t *__restrict__ a_n,
                                                                  uint32_t *__restrict__ a_slot, uint32_t *__restrict__ a_val,
                                                                  uint64_t *__restrict__ a_ballot) {
    RSP_LANE_BEGIN
    uint32_t na = 0;
    const uint32_t n = pr_n[g], t = pr_trig[g], e = pr_endp[g];
    const uint64_t b = pr_ballot[g];
    for (uint32_t k = 0; k < n; k++) {
        const size_t o = (size_t)k * v.G + g;
        const uint64_t vb = pr_vbal[o];
        L.prepare_reply(peer[g], t + k, t, e, b, vb > 0, vb, pr_vval[o], pr_vmask[o], a_slot, a_val, na);
    }
    a_n[g] = na; a_ballot[g] = na ? b : 0ull;
    RSP_LANE_END
}

// messages.rs:467-515
__global__ __launch_bounds__(256) void rsp_reconstruct_kernel(const RspView v, const uint8_t *__restrict__ flags, const uint32_t *__restrict__ rc_n,
                                                              const uint32_t *__restrict__ rc_slot, uint32_t *__restrict__ rr_n,
                                                              uint32_t *__restrict__ rr_slot, uint64_t *__restrict__ rr_bal,
                                                              uint32_t *__restrict__ rr_val, uint8_t *__restrict__ rr_mask) {
    RSP_LANE_BEGIN
    uint32_t n = 0;
    if (flags[g] & 1) {
        const uint32_t cnt = rc_n[g];
        for (uint32_t k = 0; k < cnt; k++) {
            const uint32_t s = rc_slot[(size_t)k * v.G + g];
            if (s < L.len && !L.held(s)) continue;
            L.pad_to(s);
            const size_t i = L.ix(s);
            if (v.s_st[i] < RST_ACCEPTING || v.s_mask[i] == 0) continue;
            const size_t o = (size_t)(n++) * v.G + g;
            rr_slot[o] = s; rr_bal[o] = v.s_bal[i]; rr_val[o] = v.s_val[i]; rr_mask[o] = v.s_mask[i];
        }
    }
    rr_n[g] = n;
    RSP_LANE_END
}

// messages.rs:518-594
__global__ __launch_bounds__(256) void rsp_reconstruct_reply_kernel(const RspView v, const uint8_t *__restrict__ flags,
                                                                    const uint32_t *__restrict__ rr_n, const uint32_t *__restrict__ rr_slot,
                                                                    const uint64_t *__restrict__ rr_bal, const uint32_t *__restrict__ rr_val,
                                                                    const uint8_t *__restrict__ rr_mask) {
    RSP_LANE_BEGIN
    if (flags[g] & 1) {
        const uint32_t cnt = rr_n[g];
        for (uint32_t k = 0; k < cnt; k++) {
            const size_t o = (size_t)k * v.G + g;
            const uint32_t s = rr_slot[o];
            if (!L.held(s)) continue;
            const size_t i = L.ix(s);
            if (v.s_st[i] < RST_EXECUTED && rr_bal[o] >= v.s_bal[i]) {
                L.absorb(i, rr_val[o], rr_mask[o]);
                if (s == L.cbar) L.commit_bar_run(false);
            }
        }
        L.drain_exec();                                                  // the results wait for the handler to finish
    }
    RSP_LANE_END
}

// a Heartbeat from a peer (MODE 0) / my own periodic broadcast (MODE 1: fields out, then I hear myself)
template <int MODE>
__global__ __launch_bounds__(256) void rsp_heartbeat_kernel(const RspView v, const uint8_t *__restrict__ flags, const uint8_t *__restrict__ peer,
                                                            const uint64_t *__restrict__ ballot, const uint32_t *__restrict__ commit_bar,
                                                            const uint32_t *__restrict__ exec_bar, const uint32_t *__restrict__ snap_bar,
                                                            uint8_t *__restrict__ reply, uint64_t *__restrict__ o_ballot,
                                                            uint32_t *__restrict__ o_commit, uint32_t *__restrict__ o_exec,
                                                            uint32_t *__restrict__ o_snap) {
    RSP_LANE_BEGIN
    uint8_t rp = 0; uint64_t ob = 0; uint32_t oc = 0, oe = 0, os = 0;
    if (flags[g] & 1) {
        if (MODE == 1) {
            ob = L.bms; oc = L.cbar; oe = L.ebar; os = L.snap;
            (void)L.heard_heartbeat(v.me, L.bms, L.cbar, L.ebar, L.snap);
        } else {
            // my Heartbeat back is sent right after check_leader, before the commit learning (leadership.rs:253-265)
            const uint64_t b = ballot[g];
            const uint64_t bms_after = b > L.bms ? b : L.bms;
            const uint32_t cb = L.cbar, eb = L.ebar, sb = L.snap;
            if (L.heard_heartbeat(peer[g], b, commit_bar[g], exec_bar[g], snap_bar[g])) { rp = 1; ob = bms_after; oc = cb; oe = eb; os = sb; }
        }
    }
    if (MODE == 0) reply[g] = rp;
    o_ballot[g] = ob; o_commit[g] = oc; o_exec[g] = oe; o_snap[g] = os;
    RSP_LANE_END
}


// ---- the steady state of a co-located cluster as ONE launch per tick (smr_rsp_cluster_steady_tick) ------------------------
// A block = R wavefronts over 64 groups, wavefront q = replica q (as ep_cluster_tick_kernel): the leader's handle_req_batch |
// the followers' handle_msg_accept with the ONE shard each was sent | the leader's handle_msg_accept_reply tally, peers
// ascending | on a heartbeat tick: the leader's Heartbeat, the followers' heard_heartbeat + their Heartbeats back, the leader
// hearing those, peers ascending.  Messages cross wavefronts through LDS behind block barriers.  Same handler bodies, same
// order as summerset_amd/rsp_cluster.SteadyLoop call by call.
struct RspClusterArgs {
    uint32_t R, G, leader, heartbeat;
    RspView v0;                                  // replica 0's view; replica q's arrays are delta[q] bytes further on (one layout)
    int64_t delta[SMR_MAX_REPLICAS];
    const uint32_t *val;                         // [G] the tick's batch tokens (RSP_NULL: none)
    const uint8_t *lost[4][SMR_MAX_REPLICAS];    // [kind][q] (NULL: nothing lost): 0 Accept leader -> q, 1 AcceptReply q -> leader, 2 / 3 Heartbeat out / back
    uint8_t *committed;                          // [G] out
};

template <typename T> __device__ __forceinline__ void rsp_shift_ptr(T *&p, int64_t d) { p = (T *)((char *)p + d); }
__device__ __forceinline__ void rsp_shift(RspView &v, int64_t d) {
    rsp_shift_ptr(v.leader, d); rsp_shift_ptr(v.bps, d); rsp_shift_ptr(v.bpd, d); rsp_shift_ptr(v.bms, d);
    rsp_shift_ptr(v.len, d); rsp_shift_ptr(v.cbar, d); rsp_shift_ptr(v.ebar, d); rsp_shift_ptr(v.snap, d); rsp_shift_ptr(v.peb, d);
    rsp_shift_ptr(v.digest, d); rsp_shift_ptr(v.s_bal, d); rsp_shift_ptr(v.s_vbal, d); rsp_shift_ptr(v.s_pmax, d);
    rsp_shift_ptr(v.s_st, d); rsp_shift_ptr(v.s_mask, d); rsp_shift_ptr(v.s_vmask, d); rsp_shift_ptr(v.s_fl, d); rsp_shift_ptr(v.s_packs, d);
    rsp_shift_ptr(v.s_aacks, d); rsp_shift_ptr(v.s_rsrc, d); rsp_shift_ptr(v.s_val, d); rsp_shift_ptr(v.s_vval, d); rsp_shift_ptr(v.s_ltrig, d);
    rsp_shift_ptr(v.s_lendp, d); rsp_shift_ptr(v.s_rtrig, d); rsp_shift_ptr(v.s_rendp, d); rsp_shift_ptr(v.xq, d); rsp_shift_ptr(v.xn, d);
    rsp_shift_ptr(v.counters, d);
}

__global__ __launch_bounds__(SMR_MAX_REPLICAS * 64) void rsp_cluster_tick_kernel(const RspClusterArgs a) {
    __shared__ uint32_t m_n[64], m_slot[64], m_val[64];
    __shared__ uint64_t m_bal[64];
    __shared__ uint64_t r_bal[SMR_MAX_REPLICAS][64];
    __shared__ uint64_t h_bal[64];
    __shared__ uint32_t h_commit[64], h_exec[64], h_snap[64];
    __shared__ uint64_t b_bal[SMR_MAX_REPLICAS][64];
    __shared__ uint32_t b_commit[SMR_MAX_REPLICAS][64], b_exec[SMR_MAX_REPLICAS][64], b_snap[SMR_MAX_REPLICAS][64];
    __shared__ uint8_t b_reply[SMR_MAX_REPLICAS][64];
    const uint32_t q = SMR_WAVE_UNIFORM(threadIdx.x >> 6), lane = threadIdx.x & 63u, s = a.leader;
    const uint32_t g0 = blockIdx.x * 64u + lane;
    const bool live = g0 < a.G;
    const uint32_t g = live ? g0 : 0u;
    RspView v = a.v0;
    rsp_shift(v, a.delta[q]);
    v.me = q;
    RspLane L(v, g);                                                     // (a dead lane reads group 0's scalars and writes nothing)
    // 1. the leader: handle_req_batch
    if (q == s) {
        uint32_t n = 0, slot = 0; uint64_t ab = 0;
        const uint32_t x = live ? a.val[g] : RSP_NULL;
        if (live) rsp_req_batch_lane(L, x, n, slot, ab);
        m_n[lane] = n; m_slot[lane] = slot; m_val[lane] = x; m_bal[lane] = ab;
    }
    __syncthreads();
    // 2. the followers: handle_msg_accept with the mask of the one shard they hold
    if (q != s) {
        uint64_t rb = 0; uint32_t rs = 0;
        if (live) {
            const uint8_t *la = a.lost[0][q], *lr = a.lost[1][q];
            const bool on = m_n[lane] > 0 && !(la && la[g]);
            rsp_accept_lane(L, on, s, m_slot[lane], m_bal[lane], m_val[lane], 1u << q, rb, rs);
            if (lr && lr[g]) rb = 0;                                     // the AcceptReply is lost
        }
        r_bal[q][lane] = rb;
    }
    __syncthreads();
    // 3. the leader: the AcceptReplies, peers ascending
    if (q == s && live) {
        const uint32_t slot = m_slot[lane];
        const bool was_live = m_n[lane] > 0;
        const uint32_t before = L.held(slot) ? v.s_st[L.ix(slot)] : 0u;
        for (uint32_t p = 0; p < a.R; p++) {
            if (p == s) continue;
            const uint64_t rb = r_bal[p][lane];
            if (rb != 0) L.accept_reply(p, slot, rb);
        }
        a.committed[g] = (was_live && before == RST_ACCEPTING && L.held(slot) && v.s_st[L.ix(slot)] >= RST_COMMITTED) ? 1 : 0;
    }
    if (a.heartbeat) {
        // 4. the leader's periodic Heartbeat (it hears itself)
        if (q == s) {
            h_bal[lane] = L.bms; h_commit[lane] = L.cbar; h_exec[lane] = L.ebar; h_snap[lane] = L.snap;
            if (live) (void)L.heard_heartbeat(v.me, L.bms, L.cbar, L.ebar, L.snap);
        }
        __syncthreads();
        // 5. the followers hear it; their Heartbeat back is sent right after check_leader, before the commit learning
        if (q != s) {
            uint8_t rp = 0; uint64_t ob = 0; uint32_t oc = 0, oe = 0, os = 0;
            const uint8_t *lo = a.lost[2][q], *lb = a.lost[3][q];
            if (live && !(lo && lo[g])) {
                const uint64_t b = h_bal[lane];
                const uint64_t bms_after = b > L.bms ? b : L.bms;
                const uint32_t cb = L.cbar, eb = L.ebar, sb = L.snap;
                if (L.heard_heartbeat(s, b, h_commit[lane], h_exec[lane], h_snap[lane])) { rp = 1; ob = bms_after; oc = cb; oe = eb; os = sb; }
                if (lb && lb[g]) rp = 0;
            }
            b_reply[q][lane] = rp; b_bal[q][lane] = ob; b_commit[q][lane] = oc; b_exec[q][lane] = oe; b_snap[q][lane] = os;
        }
        __syncthreads();
        // 6. the leader hears the followers' Heartbeats, peers ascending
        if (q == s && live)
            for (uint32_t p = 0; p < a.R; p++) {
                if (p == s || !b_reply[p][lane]) continue;
                (void)L.heard_heartbeat(p, b_bal[p][lane], b_commit[p][lane], b_exec[p][lane], b_snap[p][lane]);
            }
    }
    if (live) L.store();
    L.flush();
}

}  // namespace smr

using namespace smr;

struct smr_rsp_replica {
    smr_rsp_cfg cfg;
    RspView v;
    Arena arena;
};

namespace smr {
template <typename T> static void qcarve(Arena &a, T *&p, size_t n, bool dry) {
    size_t off = a.reserve(n * sizeof(T));
    if (!dry) p = a.at<T>(off);
}
static void rsp_layout(smr_rsp_replica *e, bool dry) {
    Arena &a = e->arena;
    a.used = 0;
    RspView &v = e->v;
    const size_t G = e->cfg.n_groups, W = e->cfg.window, R = e->cfg.population;
    qcarve(a, v.leader, G, dry); qcarve(a, v.bps, G, dry); qcarve(a, v.bpd, G, dry); qcarve(a, v.bms, G, dry);
    qcarve(a, v.len, G, dry); qcarve(a, v.cbar, G, dry); qcarve(a, v.ebar, G, dry); qcarve(a, v.snap, G, dry);
    qcarve(a, v.peb, R * G, dry); qcarve(a, v.digest, G, dry);
    qcarve(a, v.s_bal, W * G, dry); qcarve(a, v.s_vbal, W * G, dry); qcarve(a, v.s_pmax, W * G, dry);
    qcarve(a, v.s_st, W * G, dry); qcarve(a, v.s_mask, W * G, dry); qcarve(a, v.s_vmask, W * G, dry); qcarve(a, v.s_fl, W * G, dry);
    qcarve(a, v.s_packs, W * G, dry); qcarve(a, v.s_aacks, W * G, dry); qcarve(a, v.s_rsrc, W * G, dry);
    qcarve(a, v.s_val, W * G, dry); qcarve(a, v.s_vval, W * G, dry); qcarve(a, v.s_ltrig, W * G, dry); qcarve(a, v.s_lendp, W * G, dry);
    qcarve(a, v.s_rtrig, W * G, dry); qcarve(a, v.s_rendp, W * G, dry);
    qcarve(a, v.xq, W * G, dry); qcarve(a, v.xn, G, dry);
    qcarve(a, v.counters, SMR_CTR_WORDS, dry);
}
RspPeek rsp_peek(const smr_rsp_replica *e) {
    const RspView &v = e->v;
    return RspPeek{v.G, v.W, v.R, v.me, v.majority, v.s_val, v.s_vval, v.s_mask, v.s_vmask};
}
}  // namespace smr

extern "C" {

int smr_rsp_replica_create(const smr_rsp_cfg *cfg, smr_rsp_replica **out) {
    if (!cfg || !out) return fail(SMR_ERR_ARG, "rspaxos: null argument");
    if (cfg->n_groups == 0) return fail(SMR_ERR_ARG, "rspaxos: n_groups is zero");
    if (cfg->population < 3 || cfg->population > SMR_MAX_REPLICAS) return fail(SMR_ERR_ARG, "rspaxos: population must be in 3..8");
    if (cfg->me >= cfg->population) return fail(SMR_ERR_ARG, "rspaxos: replica id out of range");
    if (!cfg->window || (cfg->window & (cfg->window - 1)) || cfg->window < 8)
        return fail(SMR_ERR_ARG, "rspaxos: window must be a power of two >= 8");
    const uint32_t majority = cfg->population / 2 + 1;
    if (cfg->fault_tolerance > cfg->population - majority)               // mod.rs:600-604
        return fail(SMR_ERR_ARG, "rspaxos: invalid fault_tolerance");
    smr_rsp_replica *e = new smr_rsp_replica();
    e->cfg = *cfg;
    memset(&e->v, 0, sizeof(e->v));
    rsp_layout(e, true);
    e->arena.size = e->arena.used + 256;
    hipError_t err = hipMalloc((void **)&e->arena.base, e->arena.size);
    if (err != hipSuccess) { delete e; return fail(SMR_ERR_DEVICE, std::string("rspaxos: hipMalloc: ") + hipGetErrorString(err)); }
    rsp_layout(e, false);
    RspView &v = e->v;
    v.G = cfg->n_groups; v.W = cfg->window; v.Wmask = cfg->window - 1; v.R = cfg->population; v.me = cfg->me;
    v.majority = majority; v.ft = cfg->fault_tolerance;
    err = hipMemset(e->arena.base, 0, e->arena.size);
    if (err == hipSuccess) err = hipMemset(v.leader, 0xFF, v.G);
    if (err == hipSuccess) err = hipMemset(v.s_val, 0xFF, (size_t)v.W * v.G * 4);
    if (err == hipSuccess) err = hipMemset(v.s_vval, 0xFF, (size_t)v.W * v.G * 4);
    if (err != hipSuccess) {
        (void)hipFree(e->arena.base); delete e;
        return fail(SMR_ERR_DEVICE, std::string("rspaxos: init: ") + hipGetErrorString(err));
    }
    *out = e;
    return SMR_OK;
}

void smr_rsp_replica_destroy(smr_rsp_replica *e) {
    if (!e) return;
    if (e->arena.base) (void)hipFree(e->arena.base);
    delete e;
}

int smr_rsp_preset_leader(smr_rsp_replica *e, uint8_t leader) {
    if (!e || leader >= e->cfg.population) return fail(SMR_ERR_ARG, "rspaxos: bad argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const RspView &v = e->v;
    const uint64_t b = (1ull << 8) | (uint64_t)(leader + 1);
    std::vector<uint64_t> bal(v.G, b);
    SMR_HIP_TRY(hipMemset(v.leader, leader, v.G));
    SMR_HIP_TRY(hipMemcpy(v.bms, bal.data(), v.G * 8, hipMemcpyHostToDevice));
    if (leader == v.me) {
        SMR_HIP_TRY(hipMemcpy(v.bps, bal.data(), v.G * 8, hipMemcpyHostToDevice));
        SMR_HIP_TRY(hipMemcpy(v.bpd, bal.data(), v.G * 8, hipMemcpyHostToDevice));
    }
    return SMR_OK;
}

#define RSP_GRID(e) dim3(((e)->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream
#define RSP_NEED(cond) do { if (!(cond)) return fail(SMR_ERR_ARG, "rspaxos: null argument"); } while (0)

int smr_rsp_req_batch(smr_rsp_replica *e, const uint32_t *val_dev, const smr_rsp_accepts *out, void *stream) {
    RSP_NEED(e && val_dev && out && out->n && out->slot && out->val && out->ballot);
    hipLaunchKernelGGL(rsp_req_batch_kernel, RSP_GRID(e), e->v, val_dev, out->n, out->slot, out->val, out->ballot);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_handle_accept(smr_rsp_replica *e, const uint8_t *flags_dev, const uint8_t *peer_dev, const uint32_t *slot_dev,
                          const uint64_t *ballot_dev, const uint32_t *val_dev, const uint8_t *mask_dev, uint64_t *r_ballot_dev,
                          uint32_t *r_slot_dev, void *stream) {
    RSP_NEED(e && flags_dev && peer_dev && slot_dev && ballot_dev && val_dev && mask_dev && r_ballot_dev && r_slot_dev);
    hipLaunchKernelGGL(rsp_accept_kernel, RSP_GRID(e), e->v, flags_dev, peer_dev, slot_dev, ballot_dev, val_dev, mask_dev, r_ballot_dev,
                       r_slot_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_handle_accept_replies(smr_rsp_replica *e, const uint32_t *slot_dev, const uint64_t *ballot_dev, const uint8_t *flags_dev,
                                  const uint32_t *order_dev, uint8_t *committed_dev, void *stream) {
    RSP_NEED(e && slot_dev && ballot_dev && flags_dev && committed_dev);
    hipLaunchKernelGGL(rsp_accept_replies_kernel, RSP_GRID(e), e->v, slot_dev, ballot_dev, flags_dev, order_dev, committed_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_become_leader(smr_rsp_replica *e, const uint8_t *src_dev, const smr_rsp_heartbeat *hb, uint8_t *p_flags_dev,
                          uint32_t *p_trig_dev, uint64_t *p_ballot_dev, uint32_t *rc_n_dev, uint32_t *rc_slot_dev, void *stream) {
    RSP_NEED(e && src_dev && hb && hb->flags && hb->ballot && hb->commit_bar && hb->exec_bar && hb->snap_bar && p_flags_dev && p_trig_dev &&
             p_ballot_dev && rc_n_dev && rc_slot_dev);
    hipLaunchKernelGGL(rsp_become_leader_kernel, RSP_GRID(e), e->v, src_dev, hb->flags, hb->ballot, hb->commit_bar, hb->exec_bar, hb->snap_bar,
                       p_flags_dev, p_trig_dev, p_ballot_dev, rc_n_dev, rc_slot_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_handle_prepare(smr_rsp_replica *e, const uint8_t *flags_dev, const uint8_t *peer_dev, const uint32_t *trig_dev,
                           const uint64_t *ballot_dev, const smr_rsp_prepare_reply *out, void *stream) {
    RSP_NEED(e && flags_dev && peer_dev && trig_dev && ballot_dev && out && out->n && out->trig && out->endp && out->ballot && out->vbal &&
             out->vval && out->vmask);
    hipLaunchKernelGGL(rsp_prepare_kernel, RSP_GRID(e), e->v, flags_dev, peer_dev, trig_dev, ballot_dev, out->n, out->trig, out->endp,
                       out->ballot, out->vbal, out->vval, out->vmask);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_handle_prepare_replies(smr_rsp_replica *e, const uint8_t *peer_dev, const smr_rsp_prepare_reply *in, const smr_rsp_accepts *out,
                                   void *stream) {
    RSP_NEED(e && peer_dev && in && in->n && in->trig && in->endp && in->ballot && in->vbal && in->vval && in->vmask && out && out->n &&
             out->slot && out->val && out->ballot);
    hipLaunchKernelGGL(rsp_prepare_replies_kernel, RSP_GRID(e), e->v, peer_dev, in->n, in->trig, in->endp, in->ballot, in->vbal, in->vval,
                       in->vmask, out->n, out->slot, out->val, out->ballot);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_handle_reconstruct(smr_rsp_replica *e, const uint8_t *flags_dev, const uint32_t *rc_n_dev, const uint32_t *rc_slot_dev,
                               const smr_rsp_shards *out, void *stream) {
    RSP_NEED(e && flags_dev && rc_n_dev && rc_slot_dev && out && out->n && out->slot && out->bal && out->val && out->mask);
    hipLaunchKernelGGL(rsp_reconstruct_kernel, RSP_GRID(e), e->v, flags_dev, rc_n_dev, rc_slot_dev, out->n, out->slot, out->bal, out->val,
                       out->mask);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_handle_reconstruct_reply(smr_rsp_replica *e, const uint8_t *flags_dev, const smr_rsp_shards *in, void *stream) {
    RSP_NEED(e && flags_dev && in && in->n && in->slot && in->bal && in->val && in->mask);
    hipLaunchKernelGGL(rsp_reconstruct_reply_kernel, RSP_GRID(e), e->v, flags_dev, in->n, in->slot, in->bal, in->val, in->mask);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_handle_heartbeat(smr_rsp_replica *e, const uint8_t *peer_dev, const smr_rsp_heartbeat *in, uint8_t *reply_dev,
                             const smr_rsp_heartbeat *out, void *stream) {
    RSP_NEED(e && peer_dev && in && in->flags && in->ballot && in->commit_bar && in->exec_bar && in->snap_bar && reply_dev && out &&
             out->ballot && out->commit_bar && out->exec_bar && out->snap_bar);
    hipLaunchKernelGGL(rsp_heartbeat_kernel<0>, RSP_GRID(e), e->v, in->flags, peer_dev, in->ballot, in->commit_bar, in->exec_bar, in->snap_bar,
                       reply_dev, out->ballot, out->commit_bar, out->exec_bar, out->snap_bar);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_bcast_heartbeat(smr_rsp_replica *e, const uint8_t *flags_dev, const smr_rsp_heartbeat *out, void *stream) {
    RSP_NEED(e && flags_dev && out && out->ballot && out->commit_bar && out->exec_bar && out->snap_bar);
    hipLaunchKernelGGL(rsp_heartbeat_kernel<1>, RSP_GRID(e), e->v, flags_dev, (const uint8_t *)nullptr, (const uint64_t *)nullptr,
                       (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (uint8_t *)nullptr, out->ballot,
                       out->commit_bar, out->exec_bar, out->snap_bar);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

struct smr_rsp_cluster {
    uint32_t R = 0, G = 0;
    smr_rsp_replica *rep[SMR_MAX_REPLICAS] = {};
};

int smr_rsp_cluster_create(smr_rsp_replica *const *reps, uint32_t n, smr_rsp_cluster **out) {
    if (!reps || !out) return fail(SMR_ERR_ARG, "rspaxos cluster: null argument");
    if (n < 3 || n > SMR_MAX_REPLICAS) return fail(SMR_ERR_ARG, "rspaxos cluster: 3..8 replicas");
    for (uint32_t r = 0; r < n; r++) {
        if (!reps[r]) return fail(SMR_ERR_ARG, "rspaxos cluster: null replica");
        const smr_rsp_cfg &k = reps[r]->cfg, &k0 = reps[0]->cfg;
        if (k.population != n || k.me != r || k.n_groups != k0.n_groups || k.window != k0.window || k.fault_tolerance != k0.fault_tolerance)
            return fail(SMR_ERR_ARG, "rspaxos cluster: replica r must be created with me = r, population = n and the same groups, window and fault_tolerance");
        if ((char *)reps[r]->v.counters - reps[r]->arena.base != (char *)reps[0]->v.counters - reps[0]->arena.base)
            return fail(SMR_ERR_STATE, "rspaxos cluster: the replicas' arenas differ in layout");
    }
    smr_rsp_cluster *c = new smr_rsp_cluster();
    c->R = n; c->G = reps[0]->cfg.n_groups;
    for (uint32_t r = 0; r < n; r++) c->rep[r] = reps[r];
    *out = c;
    return SMR_OK;
}

void smr_rsp_cluster_destroy(smr_rsp_cluster *c) { delete c; }

int smr_rsp_cluster_steady_tick(smr_rsp_cluster *c, uint8_t leader, const uint32_t *val_dev, const uint8_t *const *lost_dev, int heartbeat,
                                uint8_t *committed_dev, void *stream) {
    if (!c || !val_dev || !committed_dev) return fail(SMR_ERR_ARG, "rspaxos cluster: null argument");
    if (leader >= c->R) return fail(SMR_ERR_ARG, "rspaxos cluster: leader out of range");
    RspClusterArgs a;
    memset(&a, 0, sizeof(a));
    a.R = c->R; a.G = c->G; a.leader = leader; a.heartbeat = heartbeat ? 1u : 0u;
    a.v0 = c->rep[0]->v;
    for (uint32_t r = 0; r < c->R; r++) a.delta[r] = (int64_t)(c->rep[r]->arena.base - c->rep[0]->arena.base);
    a.val = val_dev; a.committed = committed_dev;
    for (int k = 0; k < 4; k++)
        for (uint32_t r = 0; r < c->R; r++) a.lost[k][r] = lost_dev ? lost_dev[(size_t)k * c->R + r] : nullptr;
    hipLaunchKernelGGL(rsp_cluster_tick_kernel, dim3((c->G + 63) / 64), dim3(c->R * 64), 0, (hipStream_t)stream, a);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_dump(smr_rsp_replica *e, const smr_rsp_dump_bufs *hb) {
    if (!e || !hb) return fail(SMR_ERR_ARG, "rspaxos: null argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const RspView &v = e->v;
    const size_t G = v.G, W = v.W, R = v.R;
#define D2H(dst, src, n) SMR_HIP_TRY(hipMemcpy((dst), (src), (n), hipMemcpyDeviceToHost))
    D2H(hb->leader, v.leader, G); D2H(hb->bal_prep_sent, v.bps, G * 8); D2H(hb->bal_prepared, v.bpd, G * 8);
    D2H(hb->bal_max_seen, v.bms, G * 8); D2H(hb->len, v.len, G * 4); D2H(hb->commit_bar, v.cbar, G * 4);
    D2H(hb->exec_bar, v.ebar, G * 4); D2H(hb->snap_bar, v.snap, G * 4); D2H(hb->peer_exec_bar, v.peb, R * G * 4);
    D2H(hb->digest, v.digest, G * 8);
    D2H(hb->s_bal, v.s_bal, W * G * 8); D2H(hb->s_status, v.s_st, W * G); D2H(hb->s_val, v.s_val, W * G * 4);
    D2H(hb->s_mask, v.s_mask, W * G); D2H(hb->s_vbal, v.s_vbal, W * G * 8); D2H(hb->s_vval, v.s_vval, W * G * 4);
    D2H(hb->s_vmask, v.s_vmask, W * G); D2H(hb->s_flags, v.s_fl, W * G); D2H(hb->s_ltrig, v.s_ltrig, W * G * 4);
    D2H(hb->s_lendp, v.s_lendp, W * G * 4); D2H(hb->s_packs, v.s_packs, W * G); D2H(hb->s_aacks, v.s_aacks, W * G);
    D2H(hb->s_pmax, v.s_pmax, W * G * 8); D2H(hb->s_rsrc, v.s_rsrc, W * G); D2H(hb->s_rtrig, v.s_rtrig, W * G * 4);
    D2H(hb->s_rendp, v.s_rendp, W * G * 4);
    unsigned long long c[4];
    SMR_HIP_TRY(ctr_read(v.counters, 4, c));
#undef D2H
    for (int k = 0; k < 4; k++) hb->counters[k] = c[k];
    // canonical form: cells outside the ring of the last W slots read as null instances; bookkeeping fields only
    // where the instance has that bookkeeping
    for (size_t w = 0; w < W; w++)
        for (size_t g = 0; g < G; g++) {
            const size_t o = w * G + g;
            const uint32_t end = hb->len[g], lo = end > W ? end - (uint32_t)W : 0;
            uint32_t s = (lo & ~(uint32_t)(W - 1)) | (uint32_t)w;
            if (s < lo) s += (uint32_t)W;
            const bool live = s < end;
            if (!live) {
                hb->s_bal[o] = 0; hb->s_status[o] = 0; hb->s_val[o] = 0xFFFFFFFFu; hb->s_mask[o] = 0; hb->s_vbal[o] = 0;
                hb->s_vval[o] = 0xFFFFFFFFu; hb->s_vmask[o] = 0; hb->s_flags[o] = 0;
            }
            const uint8_t fl = hb->s_flags[o];
            if (!(fl & 1)) { hb->s_ltrig[o] = 0; hb->s_lendp[o] = 0; hb->s_packs[o] = 0; hb->s_aacks[o] = 0; hb->s_pmax[o] = 0; }
            if (!(fl & 2)) { hb->s_rsrc[o] = 0xFF; hb->s_rtrig[o] = 0; hb->s_rendp[o] = 0; }
        }
    return SMR_OK;
}

int smr_rsp_exec_poll(smr_rsp_replica *e, uint32_t *group_host, uint32_t *slot_host, uint32_t *val_host, uint64_t cap, uint64_t *n_out) {
    if (!e || !n_out) return fail(SMR_ERR_ARG, "rspaxos: null argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const RspView &v = e->v;
    const size_t G = v.G, W = v.W;
    std::vector<uint32_t> xn(G);
    SMR_HIP_TRY(hipMemcpy(xn.data(), v.xn, G * 4, hipMemcpyDeviceToHost));
    uint32_t most = 0;
    for (size_t g = 0; g < G; g++) most = xn[g] > most ? xn[g] : most;
    if (most > W) most = (uint32_t)W;
    std::vector<uint32_t> xq((size_t)most * G), val(W * G);
    if (most) {
        SMR_HIP_TRY(hipMemcpy(xq.data(), v.xq, (size_t)most * G * 4, hipMemcpyDeviceToHost));
        SMR_HIP_TRY(hipMemcpy(val.data(), v.s_val, W * G * 4, hipMemcpyDeviceToHost));
    }
    uint64_t n = 0;
    for (size_t g = 0; g < G; g++)
        for (uint32_t k = 0; k < xn[g] && k < most; k++, n++) {
            if (n >= cap || !group_host || !slot_host || !val_host) continue;
            const uint32_t slot = xq[(size_t)k * G + g];
            group_host[n] = (uint32_t)g; slot_host[n] = slot; val_host[n] = val[(size_t)(slot & v.Wmask) * G + g];
        }
    *n_out = n;
    return SMR_OK;
}

}  // extern "C"

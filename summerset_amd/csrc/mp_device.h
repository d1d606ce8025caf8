// Per-lane MultiPaxos replica handlers for gfx950 (lane = one replica group).
//
// Each function restates one reference handler of
// src/protocols/multipaxos/{request,messages,durability,leadership,execution}.rs
// on the SoA ring layout of mp_types.h, together with the WAL / executor
// completions that handler triggers under schedule LS-1 ("every WAL append and
// every state-machine command completes right after the handler that submitted
// it returns, in submission order").  Written independently of
// oracle/mp_oracle.c, which replays the same handlers over explicit queues.
#pragma once
#include <hip/hip_runtime.h>

#include "mp_types.h"

// phase stamps inside a handler (experiments only: a -DSMR_JOB_STAMPS build, tools/dbg_stamps.py)
#ifdef SMR_JOB_STAMPS
#define LSTAMP(k) do { if (__lane_id() == 0) P.dbg[(k)] = wall_clock64(); } while (0)
#else
#define LSTAMP(k) do { } while (0)
#endif

namespace smr {

__device__ __forceinline__ uint32_t m_st(uint32_t m) { return m & M_STATUS; }
__device__ __forceinline__ uint32_t m_set_st(uint32_t m, uint32_t st) { return (m & ~M_STATUS) | st; }
__device__ __forceinline__ uint32_t m_acks(uint32_t m) { return (m >> M_ACKS_SH) & 0xFFu; }
__device__ __forceinline__ uint32_t m_packs(uint32_t m) { return (m >> M_PACKS_SH) & 0xFFu; }
__device__ __forceinline__ uint32_t m_src(uint32_t m) { return (m >> M_SRC_SH) & 0x7u; }
__device__ __forceinline__ uint32_t m_vmode(uint32_t m) { return (m >> M_VMODE_SH) & 0x3u; }
__device__ __forceinline__ uint32_t m_set_vmode(uint32_t m, uint32_t vm) {
    return (m & ~(0x3u << M_VMODE_SH)) | (vm << M_VMODE_SH);
}
__device__ __forceinline__ uint32_t m_set_src(uint32_t m, uint32_t s) {
    return (m & ~(0x7u << M_SRC_SH)) | (s << M_SRC_SH);
}
__device__ __forceinline__ uint32_t ctl_order(uint32_t ctl, int i) { return (ctl >> (3 * i)) & 7u; }
__device__ __forceinline__ uint32_t ctl_drop(uint32_t ctl) { return ctl >> 24; }

// wave-wide reductions (all 64 lanes must be active: uniform mode only)
__device__ __forceinline__ uint32_t wave_min(uint32_t x) {
    for (int off = 32; off > 0; off >>= 1) { uint32_t y = __shfl_xor(x, off); x = y < x ? y : x; }
    return x;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t x) {
    for (int off = 32; off > 0; off >>= 1) { uint32_t y = __shfl_xor(x, off); x = y > x ? y : x; }
    return x;
}

// Lane context: the group's scalar state of one replica cached in registers.
struct Lane {
    const MpParams &P;
    const RepView v;               // my replica's arrays (the replica index may differ from lane to lane)
    const uint32_t g;
    const uint32_t me;
    int par;                       // outbox parity of the current tick
    uint32_t leader;
    uint64_t bps, bpd, bms;        // bal_prep_sent, bal_prepared, bal_max_seen
    uint32_t start, len, abar, cbar, ebar, snap, nlb;
    // values as loaded, for write-back of only what changed
    uint32_t o_leader; uint64_t o_bps, o_bpd, o_bms;
    uint32_t o_start, o_len, o_abar, o_cbar, o_ebar, o_snap, o_nlb;
    // Experiment (tools/experiments/README.md): [brun, len) is a run of slots all holding bal == bal_max_seen, kept by the
    // follower's steady-state append path and dropped (BAL_TOUCH) by everything else that writes a ballot or moves
    // bal_max_seen; the heartbeat's commit learning then need not load s_bal for slots inside it (8 of its 16 B per slot).
    uint32_t brun, o_brun;
    // ... and, one step further (-DSMR_BAL_LAZY): the follower's steady-state append does not STORE the ballot of a slot
    // inside the run (8 of the 16 B it writes per slot) -- the true value is bal_max_seen; whoever ends the run writes
    // the ballots out first (uniform mode: every lane its share), with the bal_max_seen the run was built under.
    // -DSMR_STATUS_LAZY: inside the run the heartbeat's commit learning (hb_advance, fused prefix) moves the bars over the
    // slots without touching them -- every slot of the run is as appended (Accepting at bal_max_seen), so the prefix
    // passes them all; a run slot below commit_bar is Executed by definition (stored: Accepting), and whoever ends the
    // run writes those statuses out first.  Nothing reads a slot below commit_bar before that: the bar scans start at
    // commit_bar / accept_bar, every generic handler ends the run on entry.
    // Round 4, one step further for the FOLLOWER's run: its steady-state append stores neither ballot NOR meta word (8 + 4 of
    // the 16 B it wrote per slot in round 1; only the batch token remains) -- a slot it appended under leader `leader` is
    // `follower_run_meta(token, leader, slot < commit_bar)`: fresh ReplicaBookkeeping from that leader, voted = (ballot,
    // reqs), Accepting, or Executed once the bars passed it.  Whoever ends the run (end_run(): everything that writes a
    // ballot, moves bal_max_seen or changes `leader` -- always BEFORE it does) writes ballots and metas out first.  The only
    // handler that meets run slots without ending the run is the heartbeat's commit learning (hb_advance), which knows.
    // The LEADER's run (mp_round_local's fast path) keeps its metas stored -- they carry the accept_acks -- and only leaves
    // out the ballots.  Which kind a run is follows from leader == me: a role change ends the run first.
#define BAL_TOUCH() end_run()
#define BAL_EXTEND(from) do { if (brun == 0xFFFFFFFFu || brun > (from)) brun = (from); } while (0)
#define HB_BAL(slot, i) ((slot) >= brun ? ballot : v.s_bal()[i])
    uint32_t obn0, obn1;           // outbox counts of parity 0 / 1 (scalars: a runtime-indexed
    bool obl0, obl1;               // array would live in scratch memory)
    uint32_t n_commit, n_redirect, n_reject;
    uint32_t n_generic;            // debug: units of work that left the fast paths
    bool ovf;
    // Wave-cooperative ("uniform") mode: all 64 lanes of a wavefront run the SAME
    // (group, replica) handler with identical scalar state; only lane 0 commits
    // stores (`wr`), and converted slot loops are strided by lane (`cl`, `cn`).
    // In the normal per-lane mode wr = true, cl = 0, cn = 1.
    bool wr;
    uint32_t cl, cn;

    __device__ __forceinline__ Lane(const MpParams &P_, uint32_t rep, uint32_t g_, int par_)
        : P(P_), v{P_.rep[0], (size_t)rep * P_.rep_stride}, g(g_), me(rep), par(par_), n_commit(0), n_redirect(0), n_reject(0), n_generic(0), ovf(false),
          wr(true), cl(0), cn(1) {
        obl0 = obl1 = false;
        obn0 = obn1 = 0;
    }

    __device__ __forceinline__ void bal_touch() { BAL_TOUCH(); }   // for callers outside the struct (experiments, see above)
    static __device__ __forceinline__ uint32_t follower_run_meta(uint32_t tok, uint32_t ldr, bool executed) {
        return (executed ? SMR_ST_EXECUTED : SMR_ST_ACCEPTING) | M_RBK | (ldr << M_SRC_SH) | (VM_SAME << M_VMODE_SH) |
               (tok ? M_NONEMPTY : 0u);
    }
    // is `slot` inside a follower's run: its meta word is not stored (see above)
    __device__ __forceinline__ bool meta_unstored(uint32_t slot) const { return slot >= brun && leader != me; }
    // end the run [brun, len): write out what its slots left unstored.  Uniform mode: every lane its share.
    __device__ __forceinline__ void end_run() {
        if (brun == 0xFFFFFFFFu) return;
        const uint32_t lo = brun > start ? brun : start;
        if (leader != me) {                                      // a follower's run: ballots and metas (4 token loads per round)
            for (uint32_t s0 = lo + cl; s0 < len; s0 += 4 * cn) {
                uint32_t tk[4];
#pragma unroll
                for (int u = 0; u < 4; u++) tk[u] = (s0 + u * cn < len) ? v.s_val()[ix(s0 + u * cn)] : 0u;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t s_ = s0 + u * cn;
                    if (s_ >= len) break;
                    const size_t i_ = ix(s_);
                    v.s_bal()[i_] = bms;
                    v.s_meta()[i_] = follower_run_meta(tk[u], leader, s_ < cbar);
                }
            }
        } else {                                                 // the leader's: ballots; metas are stored (a slot below commit_bar is Executed)
            for (uint32_t s_ = lo + cl; s_ < len; s_ += cn) {
                const size_t i_ = ix(s_);
                v.s_bal()[i_] = bms;
                if (s_ < cbar) v.s_meta()[i_] = m_set_st(v.s_meta()[i_], SMR_ST_EXECUTED);
            }
        }
        brun = 0xFFFFFFFFu;
    }
    __device__ __forceinline__ void set_uniform() { wr = __lane_id() == 0; cl = (uint32_t)__lane_id(); cn = 64; }
    __device__ __forceinline__ bool coop() const { return cn != 1; }

    __device__ __forceinline__ void load() {
        o_leader = leader = v.leader()[g];
        o_bps = bps = v.bal_prep_sent()[g];
        o_bpd = bpd = v.bal_prepared()[g];
        o_bms = bms = v.bal_max_seen()[g];
        o_start = start = v.start_slot()[g];
        o_len = len = v.log_len()[g];
        o_abar = abar = v.accept_bar()[g];
        o_cbar = cbar = v.commit_bar()[g];
        o_ebar = ebar = v.exec_bar()[g];
        o_snap = snap = v.snap_bar()[g];
        o_nlb = nlb = v.null_lb()[g];
        o_brun = brun = v.bal_lo()[g];
    }
    __device__ __forceinline__ void store() {
        if (leader != o_leader) if (wr) v.leader()[g] = (uint8_t)leader;
        if (bps != o_bps) if (wr) v.bal_prep_sent()[g] = bps;
        if (bpd != o_bpd) if (wr) v.bal_prepared()[g] = bpd;
        if (bms != o_bms) if (wr) v.bal_max_seen()[g] = bms;
        if (start != o_start) if (wr) v.start_slot()[g] = start;
        if (len != o_len) if (wr) v.log_len()[g] = len;
        if (abar != o_abar) if (wr) v.accept_bar()[g] = abar;
        if (cbar != o_cbar) if (wr) v.commit_bar()[g] = cbar;
        if (ebar != o_ebar) if (wr) v.exec_bar()[g] = ebar;
        if (snap != o_snap) if (wr) v.snap_bar()[g] = snap;
        if (nlb != o_nlb) if (wr) v.null_lb()[g] = nlb;
        if (brun != o_brun) if (wr) v.bal_lo()[g] = brun;
        if (obl0) if (wr) v.ob_cnt(0)[g] = obn0;
        if (obl1) if (wr) v.ob_cnt(1)[g] = obn1;
        if (ovf && wr) P.overflow[g] = 1;
    }

    __device__ __forceinline__ size_t ix(uint32_t slot) const { return tix(P.W, slot & P.Wmask, g); }
    // a slot's ballot: inside the run [brun, len) it is bal_max_seen and not stored (the follower's steady appends since
    // round 2, the leader's since round 4)
    __device__ __forceinline__ uint64_t slot_bal(uint32_t slot, size_t i) const { return slot >= brun ? bms : v.s_bal()[i]; }
    __device__ __forceinline__ bool is_leader() const { return leader == me; }

    // mod.rs:553-561
    __device__ __forceinline__ uint64_t make_greater_ballot(uint64_t bal) const {
        return (((bal >> 8) + 1) << 8) | (uint64_t)(me + 1);
    }

    // Vec::push(null_instance()) (mod.rs:527-538) on the ring; false = window exhausted
    __device__ __forceinline__ bool push_null() {
        if (len - start >= P.W) { ovf = true; return false; }
        size_t i = ix(len);
        if (wr) v.s_meta()[i] = 0; if (wr) v.s_bal()[i] = 0; if (wr) v.s_val()[i] = 0;
        BAL_TOUCH();
        len++;
        return true;
    }
    // pad with nulls until `slot` exists; records that Nulls may now sit at [old_len, slot)
    __device__ __forceinline__ bool pad_to(uint32_t slot) {
        if (len <= slot && len < nlb) nlb = len;
        while (len <= slot)
            if (!push_null()) return false;
        return true;
    }

    // Instance::voted accessors
    // `w`: does this lane commit the stores (uniform mode: lane 0, unless the caller is a
    // lane-strided loop in which every lane owns its slots)
    __device__ __forceinline__ uint32_t materialize_voted(size_t i, uint32_t m, uint64_t bal, uint32_t val, bool w) {
        if (m_vmode(m) == VM_SAME) {
            if (w) { v.s_vbal()[i] = bal; v.s_vval()[i] = val; }
            m = m_set_vmode(m, VM_SIDE);
        }
        return m;
    }
    __device__ __forceinline__ uint32_t materialize_voted(size_t i, uint32_t m, uint64_t bal, uint32_t val) {
        return materialize_voted(i, m, bal, val, wr);
    }
    __device__ __forceinline__ void get_voted(size_t i, uint32_t m, uint64_t bal, uint32_t val, uint64_t &vb,
                                              uint32_t &vv) const {
        uint32_t vm = m_vmode(m);
        if (vm == VM_SAME) { vb = bal; vv = val; }
        else if (vm == VM_SIDE) { vb = v.s_vbal()[i]; vv = v.s_vval()[i]; }
        else { vb = 0; vv = 0; }
    }

    __device__ __forceinline__ void ob_load(int p) {
        if (p == 0) { if (!obl0) { obn0 = v.ob_cnt(0)[g]; obl0 = true; } }
        else { if (!obl1) { obn1 = v.ob_cnt(1)[g]; obl1 = true; } }
    }
    __device__ __forceinline__ void ob_set(int p, uint32_t n) {
        ob_load(p);
        if (p == 0) obn0 = n; else obn1 = n;
    }
    // The outbox of parity p stops being a pure steady-state append run (ob_reg != 0: entry j is the Accept for slot
    // ob_reg - 1 + j at ballot ob_rbal).  Experiment -DSMR_SKIP_REG_OUTBOX (tools/experiments/README.md): such a run does
    // not store ob_slot / ob_bal at all -- 12 of the 32 bytes mp_round_local writes per new slot -- and readers derive
    // them; whoever ends the run writes the `c` entries out first.
    __device__ __forceinline__ void ob_end_run(int p, uint32_t c) {
        const uint32_t reg = v.ob_reg(p)[g];
        if (reg != 0) {
            const uint64_t rb = v.ob_rbal(p)[g];
            for (uint32_t j = cl; j < c && j < P.cap; j += cn) {  // uniform mode: every lane its share
                const size_t oj = tix(P.cap, j, g);
                v.ob_slot(p)[oj] = (OB_ACCEPT << OB_KIND_SH) | ((reg - 1 + j) & OB_SLOT_MASK);
                v.ob_bal(p)[oj] = rb;
            }
        }
        if (wr) v.ob_reg(p)[g] = 0;
    }
    // transport_hub.bcast_msg(): append to my outbox of parity p
    __device__ __forceinline__ void ob_push(int p, uint32_t kind, uint32_t slot, uint64_t bal, uint32_t val, uint32_t aux) {
        ob_load(p);
        uint32_t c = p == 0 ? obn0 : obn1;
        ob_end_run(p, c);                                       // no longer (only) a steady-state append run
        if (c >= P.cap) { ovf = true; return; }
        size_t o = tix(P.cap, c, g);
        if (wr) v.ob_slot(p)[o] = (kind << OB_KIND_SH) | (slot & OB_SLOT_MASK);
        if (wr) v.ob_bal(p)[o] = bal;
        if (wr) v.ob_val(p)[o] = val;
        if (kind == OB_HEARTBEAT) if (wr) v.ob_aux(p)[o] = aux;
        if (p == 0) obn0 = c + 1; else obn1 = c + 1;
    }

    // committed-slot list: wave-aggregated append ((group<<32)|slot)
    __device__ __forceinline__ void record_commit(uint32_t slot) {
        n_commit++;
        if (P.clist_cap == 0) return;
        if (coop()) {                              // uniform mode: one entry, appended by lane 0
            if (wr) {
                unsigned int idx = atomicAdd((unsigned int *)v.clist_n(), 1u);
                if (idx < P.clist_cap) v.clist()[idx] = ((unsigned long long)g << 32) | slot;
            }
            return;
        }
        // lanes of a wavefront may stand for different replicas, each with its own list
        const int lane = __lane_id();
        for (unsigned long long todo = __ballot(1); todo;) {
            const int first = __ffsll((long long)todo) - 1;
            const unsigned long long mask = __ballot(me == __shfl(me, first)) & todo;
            todo &= ~mask;
            if (!((mask >> lane) & 1ull)) continue;
            unsigned int base = 0;
            if (lane == first) base = atomicAdd((unsigned int *)v.clist_n(), (unsigned int)__popcll(mask));
            base = __shfl(base, first);
            const unsigned int idx = base + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
            if (idx < P.clist_cap) v.clist()[idx] = ((unsigned long long)g << 32) | slot;
            break;
        }
    }

    // leadership.rs:11-67 check_leader (lease branches are config-off)
    __device__ __forceinline__ void check_leader(uint32_t peer, uint64_t ballot) {
        if (ballot > bms) { BAL_TOUCH(); leader = peer; bms = ballot; }
    }

    // meta of N consecutive slots [s0, s0+N) below lim, as independent loads
    template <int N>
    __device__ __forceinline__ void fetch_meta(uint32_t s0, uint32_t lim, uint32_t (&mm)[N]) const {
#pragma unroll
        for (int k = 0; k < N; k++) mm[k] = (s0 + k < lim) ? v.s_meta()[ix(s0 + k)] : 0u;
    }

    // durability.rs:134-142: accept_bar forward scan after logging `slot`, whose
    // own status the caller knows to be >= Accepting
    __device__ __forceinline__ void accept_bar_scan(uint32_t slot) {
        if (slot != abar) return;
        abar++;
        while (abar < len) {
            uint32_t mm[4];
            fetch_meta<4>(abar, len, mm);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (abar >= len) return;
                if (m_st(mm[k]) < SMR_ST_ACCEPTING) return;
                abar++;
            }
        }
    }

    // durability.rs:148-218 handle_logged_commit_slot(slot) followed by the
    // executor results of what it submitted (execution.rs:56-79 handle_cmd_result,
    // one per non-empty batch, in submission order), fused into ONE forward pass:
    //   * every slot of the run [commit_bar, first slot < Committed or accept_bar)
    //     ends Executed: empty batches at commit time (durability.rs:171-172),
    //     the others when their command result arrives (execution.rs:57);
    //   * exec_bar only ever moves when a command result arrives for the slot AT
    //     exec_bar (execution.rs:70); once that happens inside the run it walks
    //     over the whole rest of the run (each later slot is either already
    //     Executed or is itself the next result) and on through Executed slots
    //     beyond it.  A run whose exec_bar slot is an EMPTY batch never moves
    //     exec_bar -- the reference's no-op pin, kept as is.
    // `m_slot` is the (already stored) meta of `slot`; `next_hint` optionally the
    // meta of slot + 1 if the caller holds it (0xFFFFFFFF = unknown); NB = slots
    // fetched per batch of independent loads when the run goes on.
    // `chase_in`: the caller already walked a prefix of this run and exec_bar is riding along.
    template <int NB>
    __device__ __forceinline__ void commit_complete(uint32_t slot, uint32_t m_slot, uint32_t next_hint = 0xFFFFFFFFu,
                                                    bool chase_in = false) {
        if (slot < start || slot != cbar) return;
        const uint32_t e0 = ebar;
        bool chase = chase_in;
        uint32_t unexec_at = 0xFFFFFFFFu;          // where the run stopped on a slot < Committed
        uint32_t s = cbar;
        auto visit = [&](uint32_t m) -> bool {      // false = run ends here
            if (m_st(m) < SMR_ST_COMMITTED) { unexec_at = s; return false; }   // durability.rs:164-166
            if (m_st(m) == SMR_ST_COMMITTED) {
                if ((m & M_NONEMPTY) && s == e0) chase = true;
                if (wr) v.s_meta()[ix(s)] = m_set_st(m, SMR_ST_EXECUTED);
            }
            s++;                                    // durability.rs:189
            return true;
        };
        bool go = s < abar && visit(m_slot);
        if (go && s < abar && next_hint != 0xFFFFFFFFu && m_st(next_hint) < SMR_ST_COMMITTED) {
            unexec_at = s;                          // caller's copy of slot + 1: still in flight
            go = false;
        }
        while (go && s < abar) {                    // durability.rs:162
            uint32_t mm[NB];
            fetch_meta<NB>(s, abar, mm);
#pragma unroll
            for (int k = 0; k < NB; k++) {
                if (s >= abar) { go = false; break; }
                if (!visit(mm[k])) { go = false; break; }
            }
        }
        cbar = s;
        if (chase) {                                // execution.rs:70-78
            ebar = cbar;
            bool scan = ebar != unexec_at;
            while (scan && ebar < len) {
                uint32_t ee[NB];
                fetch_meta<NB>(ebar, len, ee);
#pragma unroll
                for (int q = 0; q < NB; q++) {
                    if (ebar >= len) { scan = false; break; }
                    if (m_st(ee[q]) < SMR_ST_EXECUTED) { scan = false; break; }
                    ebar++;
                }
            }
        }
    }

    // messages.rs:370-443 handle_msg_accept_reply + its CommitSlot completion
    __device__ __forceinline__ void accept_reply(uint32_t peer, uint32_t slot, uint64_t ballot) {
        if (slot < start) return;                               // :377-379
        if (ballot != bpd) return;                              // :388
        if (slot >= len) return;                                // debug_assert :389
        size_t i = ix(slot);
        uint32_t m = v.s_meta()[i];
        if (!is_leader() || m_st(m) != SMR_ST_ACCEPTING) return;   // :394-399
        if (ballot < slot_bal(slot, i)) return;
        if (!(m & M_LBK)) return;                               // debug_assert :402
        uint32_t bit = 1u << (peer + M_ACKS_SH);
        if (m & bit) return;                                    // :404-406
        m |= bit;                                               // :409
        bool committed = (uint32_t)__popc(m_acks(m)) >= P.thresh;   // :412 (rspaxos/messages.rs:438-439)
        if (committed) m = m_set_st(m, SMR_ST_COMMITTED);
        if (wr) v.s_meta()[i] = m;
        if (committed) {
            record_commit(slot);
            commit_complete<2>(slot, m);                        // WAL CommitSlot :427-433 -> durability.rs:148
        }
    }

    // All AcceptReplies to ONE of my Accepts (an ack-matrix row), applied in
    // registers: `m`, `b` are the slot's meta / ballot as prefetched, `acks[s]`
    // the reply ballot of replica s (0 = none), `ctl` the delivery order / loss
    // word.  Same per-reply filter chain as accept_reply (messages.rs:377-412).
    __device__ __forceinline__ void accept_entry(uint32_t slot, uint32_t m, uint64_t b, uint32_t ctl,
                                                 const uint64_t (&acks)[MAXR], uint32_t next_hint = 0xFFFFFFFFu) {
        if (slot < start || slot >= len) return;                // :377-379, :389
        if (!is_leader() || !(m & M_LBK)) return;               // :394, :402
        const uint32_t drop = ctl_drop(ctl);
        bool changed = false, committed = false;
#pragma unroll
        for (int oi = 0; oi < MAXR; oi++) {
            const uint32_t s = ctl_order(ctl, oi);
            if ((uint32_t)oi >= P.R || s == me || s >= P.R || ((drop >> s) & 1u)) continue;
            uint64_t a = 0;
#pragma unroll
            for (int q = 0; q < MAXR; q++) a = (s == (uint32_t)q) ? acks[q] : a;
            if (a == 0 || a != bpd) continue;                   // no reply / :388
            if (m_st(m) != SMR_ST_ACCEPTING || a < b) continue; // :394-399 (a Committed slot ignores the rest)
            const uint32_t bit = 1u << (s + M_ACKS_SH);
            if (m & bit) continue;                              // :404-406
            m |= bit;                                           // :409
            changed = true;
            if ((uint32_t)__popc(m_acks(m)) >= P.thresh) {      // :412
                m = m_set_st(m, SMR_ST_COMMITTED);
                committed = true;
            }
        }
        if (!changed) return;
        if (wr) v.s_meta()[ix(slot)] = m;
        if (committed) {
            record_commit(slot);
            commit_complete<2>(slot, m, next_hint);
        }
    }

    // durability.rs:85-145 handle_logged_accept_data, leader branch
    __device__ __forceinline__ void self_accept_logged(uint32_t slot) {
        size_t i = ix(slot);
        accept_reply(me, slot, slot_bal(slot, i));                // :99-103
        accept_bar_scan(slot);
    }

    // request.rs:112-224 handle_req_batch + durability.rs:85-107 (self ack)
    __device__ __forceinline__ void req_batch(uint32_t reqs) {
        n_generic++;
        if (!is_leader() || bpd == 0) { n_redirect++; return; }    // :128-154
        // mod.rs:541-549 first_null_slot, scanning only where a Null can be
        uint32_t slot = 0xFFFFFFFFu;
        uint32_t s0 = ebar > nlb ? ebar : nlb;
        for (uint32_t s = s0; s < len; s++)
            if (m_st(v.s_meta()[ix(s)]) == SMR_ST_NULL) { slot = s; break; }
        if (slot == 0xFFFFFFFFu) {
            if ((len - start) + P.win_reserve >= P.W) { n_reject++; return; }   // ring back-pressure
            slot = len;
            len++;                                              // push; filled right below
        }
        nlb = slot + 1;
        size_t i = ix(slot);
        // :158-182 fresh LeaderBookkeeping (all-zero side fields), Accepting at bal_prepared,
        // voted = (bal, reqs) :190.  The WAL AcceptData append (:191-201) completes at once:
        // durability.rs:99-103 self AcceptReply -- every filter of messages.rs:377-406 passes by
        // construction (slot in range, ballot == bal_prepared == inst.bal, leader, Accepting,
        // fresh accept_acks), so the self bit is set right here.
        uint32_t m = SMR_ST_ACCEPTING | M_EXT | M_LBK | (VM_SAME << M_VMODE_SH) | (reqs ? M_NONEMPTY : 0u) |
                     (1u << (me + M_ACKS_SH));
        const bool committed = P.thresh <= 1;                   // messages.rs:412 (only for a 1-ack threshold)
        if (committed) m = m_set_st(m, SMR_ST_COMMITTED);
        if (wr) v.s_bal()[i] = bpd;
        BAL_TOUCH();
        if (wr) v.s_val()[i] = reqs;
        if (wr) v.s_meta()[i] = m;
        ob_push(par, OB_ACCEPT, slot, bpd, reqs, 0);            // :209-216
        if (committed) { record_commit(slot); commit_complete<2>(slot, m); }
        accept_bar_scan(slot);                                  // durability.rs:134-142
    }

    // first slot of [lo, hi) whose status is below `bound`, else hi.  Lane-strided in
    // uniform mode (each lane checks every 64th slot, then a wave-wide min).
    __device__ __forceinline__ uint32_t first_status_below(uint32_t lo, uint32_t hi, uint32_t bound) const {
        uint32_t found = hi;
        for (uint32_t s0 = lo + cl; s0 < hi && found == hi; s0 += 4 * cn) {
            uint32_t mm[4];
#pragma unroll
            for (int u = 0; u < 4; u++) mm[u] = (s0 + u * cn < hi) ? v.s_meta()[ix(s0 + u * cn)] : 0xFFFFFFFFu;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (found == hi && mm[u] != 0xFFFFFFFFu && m_st(mm[u]) < bound) found = s0 + u * cn;
        }
        return coop() ? wave_min(found) : found;
    }
    // last slot of [lo, hi) whose status is below `bound` (below = true) or above it, else `none`
    __device__ __forceinline__ uint32_t last_status(uint32_t lo, uint32_t hi, uint32_t bound, bool below,
                                                    uint32_t none) const {
        uint32_t found = 0;                                       // slot + 1, 0 = none
        if (hi > lo)
            for (uint32_t t = cl; t < hi - lo; t += cn) {          // t-th slot from the top
                const uint32_t s = hi - 1 - t;
                const uint32_t st = m_st(v.s_meta()[ix(s)]);
                if (below ? st < bound : st > bound) { found = s + 1; break; }
            }
        if (coop()) found = wave_max(found);
        return found ? found - 1 : none;
    }

    // messages.rs:219-287: the sender's PrepareReplies are complete up to its
    // endprep_slot -> count it on the trigger slot; on a quorum, bal_prepared = ballot
    // and every Preparing slot from the trigger on enters the Accept phase.
    // The AcceptData completions (durability.rs:85-145: self ack + accept_bar scan)
    // are folded into the same ascending pass: each touches only its own slot, and
    // the accept_bar scan that the completion of the slot AT accept_bar starts runs
    // over the slots' final statuses, which an ascending pass knows as it goes.
    __device__ __forceinline__ void prepare_quorum_step(uint32_t peer, uint32_t trig, uint64_t ballot) {
        const size_t ti = ix(trig);
        uint32_t tm = v.s_meta()[ti];
        tm |= 1u << (peer + M_PACKS_SH);                        // :228
        if (wr) v.s_meta()[ti] = tm;
        if ((uint32_t)__popc(m_packs(tm)) < P.quorum) return;   // :233
        bpd = ballot;                                           // :236
        if (coop() && P.thresh > 1) {
            // 64 consecutive slots per step, one per lane: the Accepts go to the outbox in slot
            // order (ballot + prefix count), the accept_bar scan runs on the bitmaps.
            ob_load(par ^ 1);
            uint32_t c = (par ^ 1) == 0 ? obn0 : obn1;
            ob_end_run(par ^ 1, c);
            int chase = 0;
            for (uint32_t base = trig; base < len; base += 64) {
                const uint32_t sl = base + cl;
                const bool in = sl < len;
                const size_t i = ix(sl);
                uint32_t m = in ? v.s_meta()[i] : 0u;
                const bool moved = in && m_st(m) == SMR_ST_PREPARING;
                if (moved) {
                    m = m_set_st(m, SMR_ST_ACCEPTING);
                    if (v.s_bal()[i] == ballot && (m & M_LBK) && !(m_acks(m) & (1u << me))) m |= 1u << (me + M_ACKS_SH);
                    v.s_meta()[i] = m;
                }
                const unsigned long long mv = __ballot(moved);
                const unsigned long long acc = __ballot(in && m_st(m) >= SMR_ST_ACCEPTING);
                if (moved) {
                    const uint32_t pos = c + (uint32_t)__popcll(mv & ((1ull << cl) - 1ull));
                    if (pos < P.cap) {
                        const size_t o = tix(P.cap, pos, g);
                        v.ob_slot(par ^ 1)[o] = (OB_ACCEPT << OB_KIND_SH) | (sl & OB_SLOT_MASK);
                        v.ob_bal(par ^ 1)[o] = ballot;
                        v.ob_val(par ^ 1)[o] = v.s_val()[i];
                    }
                }
                c += (uint32_t)__popcll(mv);
                // durability.rs:134-142 on the bitmaps: the scan starts at the completion of the
                // moved slot sitting AT accept_bar and runs while slots are >= Accepting
                if (chase != 2 && abar >= base && abar < base + 64 && abar < len) {
                    const uint32_t pbit = abar - base;
                    if (chase == 0 && ((mv >> pbit) & 1ull)) chase = 1;
                    if (chase == 1) {
                        const unsigned long long run = ~(acc >> pbit);           // first 0 = first slot < Accepting
                        uint32_t n = run ? (uint32_t)(__ffsll((long long)run) - 1) : 64u - pbit;
                        if (n > 64u - pbit) n = 64u - pbit;
                        abar += n;
                        if (abar < base + 64 && abar < len) chase = 2;           // stopped inside the chunk
                        if (abar > len) abar = len;
                    }
                }
            }
            if (c > P.cap) { ovf = true; c = P.cap; }
            if ((par ^ 1) == 0) obn0 = c; else obn1 = c;
            return;
        }
        int chase = 0;                                          // 0 not started, 1 running, 2 over
        for (uint32_t s0 = trig; s0 < len; s0 += 8) {           // :238-286
            uint32_t mm[8], vv[8]; uint64_t bb[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const bool in = s0 + k < len;
                const size_t i = ix(s0 + k);
                mm[k] = in ? v.s_meta()[i] : 0u; vv[k] = in ? v.s_val()[i] : 0u; bb[k] = in ? v.s_bal()[i] : 0ull;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t s = s0 + k;
                if (s >= len) break;
                uint32_t m = mm[k];
                const bool moved = m_st(m) == SMR_ST_PREPARING;
                if (moved) {
                    m = m_set_st(m, SMR_ST_ACCEPTING);
                    ob_push(par ^ 1, OB_ACCEPT, s, ballot, vv[k], 0);   // travels in the next tick
                    // its AcceptData completion: messages.rs:377-412 with peer = me, ballot = inst.bal
                    if (bb[k] == ballot && (m & M_LBK) && !(m_acks(m) & (1u << me))) {
                        m |= 1u << (me + M_ACKS_SH);
                        if ((uint32_t)__popc(m_acks(m)) >= P.thresh) {   // only with a 1-ack threshold
                            m = m_set_st(m, SMR_ST_COMMITTED);
                            if (wr) v.s_meta()[ix(s)] = m;
                            record_commit(s);
                            commit_complete<2>(s, m);
                            m = v.s_meta()[ix(s)];
                        }
                    }
                    if (wr) v.s_meta()[ix(s)] = m;
                }
                if (chase == 0 && moved && s == abar) chase = 1;            // durability.rs:134
                if (chase == 1 && s == abar) {
                    if (m_st(m) >= SMR_ST_ACCEPTING) abar = s + 1; else chase = 2;
                }
            }
        }
    }

    // messages.rs:87-292 handle_msg_prepare_reply.  peer_accept_bar bookkeeping is lease-only.
    __device__ __forceinline__ void prepare_reply(uint32_t peer, uint32_t slot, uint32_t trig, uint32_t endp, uint64_t ballot,
                                  bool has_voted, uint64_t vbal, uint32_t vval) {
        if (slot < start) return;                               // :97-99
        if (ballot != bps) return;                              // :110
        if (!is_leader()) return;                               // :112-114
        if (trig < start || trig >= len) return;                // debug_assert :116-119
        const size_t ti = ix(trig);
        uint32_t tm = v.s_meta()[ti];
        if (!(tm & M_LBK)) return;                              // :120-125
        const uint32_t my_endp = (tm & M_LBKX) ? v.s_lendp()[ti] : 0;   // :149-153
        while (len <= slot) {                                   // :154-190 slot unknown at become_a_leader
            uint32_t this_slot = len;
            if (!push_null()) return;
            size_t i = ix(this_slot);
            if (wr) v.s_bal()[i] = bps;
            if (wr) v.s_ltrig()[i] = trig; if (wr) v.s_lendp()[i] = my_endp; if (wr) v.s_pmax()[i] = 0;
            if (wr) v.s_meta()[i] = SMR_ST_PREPARING | M_EXT | M_LBK | M_LBKX;
            if (nlb == this_slot) nlb = this_slot + 1;          // filled at once: still no Null below the log end
            // its PrepareBal completion is a no-op on the leader (this_slot > endprep)
        }
        {
            size_t i = ix(slot);
            uint32_t m = v.s_meta()[i];
            uint64_t b = v.s_bal()[i];
            if (m_st(m) != SMR_ST_PREPARING || ballot < b) return;   // :196-198
            if (has_voted && (m & M_LBK)) {                     // :203-216
                uint64_t pm = (m & M_LBKX) ? v.s_pmax()[i] : 0;
                if (vbal > pm) {
                    if (!(m & M_LBKX)) { if (wr) v.s_ltrig()[i] = 0; if (wr) v.s_lendp()[i] = 0; m |= M_LBKX; }
                    if (wr) v.s_pmax()[i] = vbal;
                    m = materialize_voted(i, m, b, v.s_val()[i]);
                    if (wr) v.s_val()[i] = vval;                          // inst.reqs = val
                    m = vval ? (m | M_NONEMPTY) : (m & ~M_NONEMPTY);
                    if (wr) v.s_meta()[i] = m;
                }
            }
        }
        if (slot != endp) return;                               // :222
        prepare_quorum_step(peer, trig, ballot);
    }

    // One sender's whole PrepareReply batch (slots trig .. trig+n-1, FIFO): the checks of
    // messages.rs:97-125 do not depend on the slot, so they are made once; the per-slot
    // part (:196-216) runs on 8 slots per batch of loads.  A slot beyond my log end
    // (:154-190) goes through prepare_reply() one by one.
    __device__ __forceinline__ void prepare_reply_batch(uint32_t peer, uint32_t trig, uint32_t endp, uint64_t ballot,
                                                        uint32_t n, SMR_G const uint64_t *pr_vbal,
                                                        SMR_G const uint32_t *pr_vval) {
        if (ballot != bps || !is_leader()) return;              // :110-114
        if (trig < start || trig >= len) return;                // :97-99 (slot >= trig), :116-119
        if (!(v.s_meta()[ix(trig)] & M_LBK)) return;              // :120-125
        const uint32_t len0 = len;
        const uint32_t n_mine = trig + n <= len0 ? n : len0 - trig;      // replies for slots I already hold
        // This ballot already has its quorum and every slot of the batch is one I hold: the quorum
        // step moved all Preparing slots from the trigger on to Accepting, so each reply fails the
        // status test of :196-198 and nothing (not even prepare_acks, :222 comes after) changes.
        if (bpd == ballot && n_mine == n) return;
        for (uint32_t k0 = cl; k0 < n_mine; k0 += 4 * cn) {      // :196-216, every lane owns its slots;
            uint64_t vb[4], bb[4], pm[4]; uint32_t mm[4], vv[4], vl[4];   // 4 strided slots per lane, loads first
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t k = k0 + u * cn;
                const bool in = k < n_mine;
                const size_t o = tix(P.pcap, in ? k : 0, g), i = ix(trig + (in ? k : 0));
                vb[u] = in ? pr_vbal[o] : 0ull; vv[u] = in ? pr_vval[o] : 0u;
                mm[u] = in ? v.s_meta()[i] : 0u; bb[u] = in ? v.s_bal()[i] : 0ull;
                pm[u] = in ? v.s_pmax()[i] : 0ull; vl[u] = in ? v.s_val()[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t k = k0 + u * cn;
                if (k >= n_mine || vb[u] == 0) continue;         // voted: None
                uint32_t m = mm[u];
                if (m_st(m) != SMR_ST_PREPARING || ballot < bb[u] || !(m & M_LBK)) continue;
                const uint64_t cur = (m & M_LBKX) ? pm[u] : 0ull;
                if (vb[u] > cur) {
                    const size_t i = ix(trig + k);
                    if (!(m & M_LBKX)) { v.s_ltrig()[i] = 0; v.s_lendp()[i] = 0; m |= M_LBKX; }
                    v.s_pmax()[i] = vb[u];
                    m = materialize_voted(i, m, bb[u], vl[u], true);
                    v.s_val()[i] = vv[u];
                    m = vv[u] ? (m | M_NONEMPTY) : (m & ~M_NONEMPTY);
                    v.s_meta()[i] = m;
                }
            }
        }
        if (n_mine == n) {                                       // the batch ends inside my log: :222
            if (endp >= trig && endp < trig + n) {
                const size_t i = ix(endp);
                if (m_st(v.s_meta()[i]) == SMR_ST_PREPARING && ballot >= v.s_bal()[i]) prepare_quorum_step(peer, trig, ballot);
            }
            return;
        }
        if (coop() && (len0 - start) + (n - n_mine) <= P.W) {
            // Unknown slots (:154-190), all at once: slot len0 + t is pushed, becomes a Preparing
            // instance of this Prepare phase with a fresh LeaderBookkeeping, and then takes the reply
            // (:196-216 with prepare_max_bal == 0): the voted value, if any.  One lane per slot.
            const size_t ti = ix(trig);
            const uint32_t tm = v.s_meta()[ti];
            const uint32_t my_endp = (tm & M_LBKX) ? v.s_lendp()[ti] : 0;   // :149-153
            for (uint32_t k = n_mine + cl; k < n; k += cn) {
                const size_t o = tix(P.pcap, k, g), i = ix(trig + k);
                const uint64_t vb = pr_vbal[o];
                const uint32_t vv = vb > 0 ? pr_vval[o] : 0u;
                v.s_bal()[i] = bps;
                BAL_TOUCH();
                v.s_val()[i] = vv;
                v.s_ltrig()[i] = trig; v.s_lendp()[i] = my_endp; v.s_pmax()[i] = vb;
                v.s_meta()[i] = SMR_ST_PREPARING | M_EXT | M_LBK | M_LBKX | (vv ? M_NONEMPTY : 0u);
            }
            if (nlb == len0) nlb = trig + n;                     // filled at once: no Null below the log end
            len = trig + n;
            if (endp >= trig + n_mine && endp < trig + n) prepare_quorum_step(peer, trig, ballot);   // :222
            return;
        }
        for (uint32_t k = n_mine; k < n && !ovf; k++) {          // unknown slots: pad + reply, one by one
            const size_t o = tix(P.pcap, k, g);
            const uint64_t vb = pr_vbal[o];
            prepare_reply(peer, trig + k, trig, endp, ballot, vb > 0, vb, pr_vval[o]);
        }
    }

    // leadership.rs:73-214 become_a_leader + its PrepareBal completions
    // (durability.rs:10-49: the leader's own PrepareReply, messages.rs:196-228).
    // Those completions touch only their own slot (and, for the last one, the trigger
    // slot's prepare_acks), so they are folded into the one ascending pass that rewrites
    // the in-progress instances; 8 slots per batch of loads.
    __device__ __forceinline__ void become_a_leader(uint32_t src) {
        if (leader != NO_REP && leader != src) return;          // :77-81
        BAL_TOUCH();                                            // (a follower's run is written out under the leader it was built under)
        leader = me;                                            // :98
        // :104 bcast_heartbeats() now, still carrying the old bal_max_seen (:240-247)
        ob_push(par, OB_HEARTBEAT, cbar, bms, ebar, snap);
        for (uint32_t p = 0; p < P.R; p++) if (wr) v.peer_exec_bar()[(size_t)p * P.G + g] = 0;   // :107-109
        bpd = 0;                                                // :112-114
        bps = make_greater_ballot(bms);
        BAL_TOUCH();
        bms = bps;
        uint32_t trig = first_status_below(start, len, SMR_ST_COMMITTED);          // :117-123 (else log end)
        const uint32_t endp = last_status(start, len, SMR_ST_COMMITTED, true, len); // :124-130 (else log end)
        if (trig == len) {                                      // :131-134
            if (!push_null()) return;
            if (nlb == len - 1) nlb = len;                      // it turns Preparing below: not a Null hole
        }
        const uint32_t e0 = ebar;
        // Will my own PrepareReplies be counted?  messages.rs:116-125 looks at the trigger
        // slot's leader_bk, which the pass below creates when the trigger lies in it.
        const bool self_ok = trig >= e0 ? true : (v.s_meta()[ix(trig)] & M_LBK) != 0;
        for (uint32_t s = e0 + cl; s < len; s += cn) {          // :142-183, every lane owns its slots
            const size_t i = ix(s);
            uint32_t m = v.s_meta()[i];
            const uint32_t st = m_st(m);
            if (st == SMR_ST_EXECUTED) continue;
            m |= M_EXT;
            if (st == SMR_ST_COMMITTED) { v.s_meta()[i] = m; continue; }
            const uint64_t b = v.s_bal()[i];
            const uint32_t val = v.s_val()[i];
            uint64_t vb; uint32_t vv;
            get_voted(i, m, b, val, vb, vv);
            m = materialize_voted(i, m, b, val, true);
            m = m_set_st(m, SMR_ST_PREPARING) | M_LBK | M_LBKX;
            m &= ~((0xFFu << M_ACKS_SH) | (0xFFu << M_PACKS_SH));
            uint64_t pmax = 0;
            // PrepareBal completion -> my own PrepareReply for this slot (durability.rs:33-48):
            // keep the value I voted for, if any (messages.rs:203-216 with prepare_max_bal == 0)
            if (self_ok && s <= endp && vb > 0) {
                pmax = vb;
                if (vv != val) v.s_val()[i] = vv;
                m = vv ? (m | M_NONEMPTY) : (m & ~M_NONEMPTY);
            }
            v.s_bal()[i] = bps;
            v.s_ltrig()[i] = trig; v.s_lendp()[i] = endp; v.s_pmax()[i] = pmax;
            v.s_meta()[i] = m;
        }
        ob_push(par, OB_PREPARE, trig, bps, 0, 0);              // :192-198
        // the completion of slot endprep counts me in (messages.rs:222-233)
        if (self_ok && endp >= e0 && endp < len) prepare_quorum_step(me, trig, bps);
    }

    // messages.rs:12-83 handle_msg_prepare + durability.rs:50-78 (PrepareReply
    // per slot, written as one batch: header + (voted_bal, voted_reqs) entries);
    // 8 slots per batch of loads
    __device__ __forceinline__ void msg_prepare(uint32_t peer, uint32_t trig, uint64_t ballot) {
        if (trig < start) return;                               // :18-20
        if (ballot < bms) return;                               // :29
        LSTAMP(40);
        BAL_TOUCH();                                            // (reads slot ballots below)
        LSTAMP(41);
        check_leader(peer, ballot);
        LSTAMP(42);
        if (!pad_to(trig)) return;                              // :37-39
        const uint32_t last = last_status(start, len, SMR_ST_NULL, false, start);   // :43-52 (unwrap_or(0))
        LSTAMP(43);
        const uint32_t endp = last > trig ? last : trig;
        const uint32_t n = endp - trig + 1;
        const bool follower = !is_leader();
        if (follower && (v.pr_cnt()[g] != 0 || n > P.pcap)) { ovf = true; return; }
        for (uint32_t s = trig + cl; s <= endp; s += cn) {      // :55-79, every lane owns its slots
            const size_t i = ix(s);
            uint32_t m = v.s_meta()[i];
            const uint64_t b = v.s_bal()[i];
            const uint32_t val = v.s_val()[i];
            uint64_t vb; uint32_t vv;
            get_voted(i, m, b, val, vb, vv);
            m = materialize_voted(i, m, b, val, true);
            v.s_bal()[i] = ballot;
            BAL_TOUCH();
            m = m_set_src(m_set_st(m, SMR_ST_PREPARING) | M_RBK | M_RBKX, peer);
            v.s_rtrig()[i] = trig; v.s_rendp()[i] = endp;
            v.s_meta()[i] = m;
            if (follower) {                                     // durability.rs:50-78
                size_t o = tix(P.pcap, s - trig, g);
                v.pr_vbal()[o] = vb; v.pr_vval()[o] = vv;
            }
        }
        LSTAMP(44);
        if (follower) {
            if (wr) v.pr_dest()[g] = (uint8_t)peer; if (wr) v.pr_trig()[g] = trig; if (wr) v.pr_endp()[g] = endp;
            if (wr) v.pr_bal()[g] = ballot; if (wr) v.pr_abar()[g] = abar;
            if (wr) v.pr_cnt()[g] = n;
        }
    }

    // messages.rs:295-367 handle_msg_accept + durability.rs:85-145; returns the
    // AcceptReply ballot for the sender (0 = no reply)
    __device__ __forceinline__ uint64_t msg_accept(uint32_t peer, uint32_t slot, uint64_t ballot, uint32_t reqs) {
        if (slot < start) return 0;                             // :302-304
        if (ballot < bms) return 0;                             // :313
        BAL_TOUCH();                                            // (reads the slot's ballot below)
        check_leader(peer, ballot);
        uint32_t m = 0;
        size_t i = ix(slot);
        if (slot < len) m = v.s_meta()[i];
        else if (slot == len) {                                 // common case: push + fill fused
            if (len - start >= P.W) { ovf = true; return 0; }
            if (nlb == len) nlb = len + 1;                      // still no Null below the log end
            len++;
        } else {
            if (!pad_to(slot)) return 0;                        // :321-323
        }
        m = m_set_st(m, SMR_ST_ACCEPTING);                      // :327-329
        if (!(m & M_RBK)) m = (m | M_RBK) & ~M_RBKX;            // :331-339
        m = m_set_src(m, peer);
        m = m_set_vmode(m, VM_SAME);                            // :351 voted = (ballot, reqs)
        m = reqs ? (m | M_NONEMPTY) : (m & ~M_NONEMPTY);
        if (wr) v.s_bal()[i] = ballot;
        BAL_TOUCH();
        if (wr) v.s_val()[i] = reqs;
        if (wr) v.s_meta()[i] = m;
        uint64_t reply = 0;
        if (is_leader()) accept_reply(me, slot, ballot);        // durability.rs:99-103 (not reachable: a
                                                                // peer's ballot >= mine deposes me)
        else reply = ballot;                                    // durability.rs:108-131 -> source == peer
        accept_bar_scan(slot);                                  // durability.rs:134-142
        return reply;
    }

    // leadership.rs:270-346 heard_heartbeat + :372-427 advance_commit_bar
    // in three parts, so that a kernel can run the commit-bar pass of lanes whose senders differ together
    __device__ __forceinline__ void heard_heartbeat(uint32_t peer, uint64_t ballot, uint32_t hb_commit, uint32_t hb_exec,
                                    uint32_t hb_snap) {
        if (!hb_gate(peer, ballot, hb_exec)) return;
        if (hb_commit > cbar && !hb_advance(ballot, hb_commit)) return;   // :379
        hb_peer(peer, hb_exec, hb_snap);
    }
    __device__ __forceinline__ bool hb_gate(uint32_t peer, uint64_t ballot, uint32_t hb_exec) {
        if (peer != me) check_leader(peer, ballot);             // :278-285
        if (ballot < bms) return false;                         // :303-305
        if (hb_exec < ebar) return false;                       // :312-314
        return true;
    }
    // advance_commit_bar (:372-427) for hb_commit > commit_bar; false = the heartbeat is dropped here
    __device__ __forceinline__ bool hb_advance(uint64_t ballot, uint32_t hb_commit) {
        {
            if (len < hb_commit && !pad_to(hb_commit - 1)) return false;   // :380-382
            // Fused prefix (the steady state of a follower): while slots are Accepting at a ballot
            // >= the heartbeat's and below accept_bar, advance_commit_bar marks them Committed
            // (:385-416), the CommitSlot completion of the first one starts the commit-bar run
            // (durability.rs:162-189) which passes each of them, and their command results bring
            // them to Executed (execution.rs:57) -- one read and one write per slot instead of three
            // passes.  Exact: the run was started by slot commit_bar itself (it is Accepting here),
            // and exec_bar rides along iff that slot is a non-empty batch sitting at exec_bar.
            const uint32_t c0 = cbar, e0 = ebar;
            uint32_t sfx = cbar;
            bool chase = false;
            if (!coop()) {
                bool simple = true;
                // slots below the run the usual way, then the run in one step
                const bool lz = brun != 0xFFFFFFFFu && bms >= ballot;
                const uint32_t eager_end = lz && brun < hb_commit ? (brun > sfx ? brun : sfx) : hb_commit;
                while (simple && sfx < eager_end) {
                    uint32_t mm[8]; uint64_t bb[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        bool in = sfx + k < eager_end;
                        size_t i = ix(sfx + k);
                        mm[k] = in ? v.s_meta()[i] : 0u;
                        bb[k] = in ? HB_BAL(sfx + k, i) : 0ull;
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (sfx >= eager_end) break;
                        if (!(m_st(mm[k]) == SMR_ST_ACCEPTING && bb[k] >= ballot && sfx < abar)) { simple = false; break; }
                        if ((mm[k] & M_NONEMPTY) && sfx == e0) chase = true;
                        v.s_meta()[ix(sfx)] = m_set_st(mm[k], SMR_ST_EXECUTED);
                        sfx++;
                    }
                }
                if (simple && lz && sfx >= brun && sfx < hb_commit) {
                    const uint32_t tgt = hb_commit < abar ? hb_commit : abar;
                    if (tgt > sfx) {
                        // (M_NONEMPTY mirrors s_val != 0; a follower's run stores no meta word)
                        if (sfx == e0 && (meta_unstored(sfx) ? v.s_val()[ix(sfx)] != 0u : (v.s_meta()[ix(sfx)] & M_NONEMPTY) != 0u)) chase = true;
                        sfx = tgt;
                    }
                }
                if (sfx > c0) { cbar = sfx; if (chase) ebar = sfx; }
            }
            uint32_t first = 0xFFFFFFFFu, first_m = 0;
            {                                                   // :385-416, 8 slots per batch of loads
                uint32_t s = sfx;
                bool go = true;
                while (go && s < hb_commit) {
                    uint32_t mm[8]; uint64_t bb[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        bool in = s + k < hb_commit;
                        size_t i = ix(s + k);
                        // (a slot of a follower's run: its meta is what the append would have stored -- Accepting here, the bars
                        // have not passed it)
                        mm[k] = !in ? 0u : (meta_unstored(s + k) ? follower_run_meta(v.s_val()[i], leader, false) : v.s_meta()[i]);
                        bb[k] = in ? HB_BAL(s + k, i) : 0ull;
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (s >= hb_commit) { go = false; break; }
                        uint32_t st = m_st(mm[k]);
                        if (bb[k] < ballot || st < SMR_ST_ACCEPTING) { go = false; break; }
                        if (st < SMR_ST_COMMITTED) {
                            uint32_t m = m_set_st(mm[k], SMR_ST_COMMITTED);
                            BAL_TOUCH();                        // a slot of the run marked one by one: the run ends here
                            if (wr) v.s_meta()[ix(s)] = m;
                            if (first == 0xFFFFFFFFu) { first = s; first_m = m; }
                        }
                        s++;
                    }
                }
            }
            if (sfx > c0) {
                // the run started in the fused prefix goes on from commit_bar, whatever marked it
                // a slot of a follower's run at commit_bar is Accepting: the run of commits ends in front of it, exec_bar rides
                // up to it (what commit_complete does with such a slot, without reading the meta word that is not there)
                if (cbar < len && !meta_unstored(cbar)) commit_complete<8>(cbar, v.s_meta()[ix(cbar)], 0xFFFFFFFFu, chase);
                else if (chase) ebar = cbar;
            } else if (first != 0xFFFFFFFFu) {
                // CommitSlot completions: only the first can sit at commit_bar
                commit_complete<8>(first, first_m);
            }
        }
        return true;
    }
    __device__ __forceinline__ void hb_peer(uint32_t peer, uint32_t hb_exec, uint32_t hb_snap) {
        if (peer != me) {                                       // :320-342
            size_t po = (size_t)peer * P.G + g;
            if (hb_exec > v.peer_exec_bar()[po]) {
                if (wr) v.peer_exec_bar()[po] = hb_exec;
                uint32_t passed = 1;
                for (uint32_t p = 0; p < P.R; p++)
                    if (p != me && v.peer_exec_bar()[(size_t)p * P.G + g] >= hb_exec) passed++;
                if (passed == P.R) snap = hb_exec;
            }
            if (hb_snap > snap) snap = hb_snap;
        }
    }
};

}  // namespace smr

// MultiPaxos / RSPaxos lock-step cluster: round kernels + C-ABI.
//
// Grid shape of every round kernel: blockIdx.y = replica id, 256 lanes per
// block = 256 consecutive groups, so every per-group array access of a
// wavefront is one contiguous request.  (mp_quorum_tally is the exception: its
// lanes pick their group's leader, whichever replica that is.)  Rounds are separate launches because
// each consumes what the previous one wrote for OTHER replicas (and, in the
// multi-GPU layout, the exchange sits between them).
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <vector>

#include "mp_device.h"
#include "smr_common.h"

// experiment (profiles/r9h): the bulk round kernels' wavefronts at a raised issue priority, so that on a SIMD they share with the
// side launch's wavefronts they go first.  Off unless built with -DSMR_BULK_PRIO=1..3
#ifdef SMR_BULK_PRIO
#define SMR_RAISE_PRIO() __builtin_amdgcn_s_setprio(SMR_BULK_PRIO)
#else
#define SMR_RAISE_PRIO() ((void)0)
#endif

namespace smr {

// wave-reduce the counters, one atomic per wave and replica
__device__ __forceinline__ void flush_counters(const Lane &L, bool active) {
    unsigned int c[4] = {active ? L.n_commit : 0u, active ? L.n_redirect : 0u, active ? L.n_reject : 0u,
                         active ? L.n_generic : 0u};
    const bool any = (c[0] | c[1] | c[2] | c[3]) != 0;
    // lanes of a wavefront may stand for different replicas: one reduction per replica that counted
    for (unsigned long long todo = __ballot(any); todo;) {
        const uint32_t rep = __shfl(L.me, __ffsll((long long)todo) - 1);
        const bool mine = any && L.me == rep;
        todo &= ~__ballot(mine);
        unsigned int x[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            x[k] = mine ? c[k] : 0u;
            for (int off = 32; off > 0; off >>= 1) x[k] += __shfl_xor(x[k], off);
        }
        if (__lane_id() == 0) {
            SMR_G unsigned long long *const ctr = RepView{L.P.rep[0], (size_t)rep * L.P.rep_stride}.counters();
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (x[k]) ctr_add((unsigned long long *)ctr, k, (unsigned long long)x[k]);
        }
    }
}

// a cooperative job's counters: identical in all lanes, added once
__device__ __forceinline__ void flush_job(const Lane &J) {
    if (__lane_id() != 0) return;
    SMR_G unsigned long long *const ctr = J.v.counters();
    if (J.n_commit) ctr_add((unsigned long long *)ctr, 0, (unsigned long long)J.n_commit);
    if (J.n_redirect) ctr_add((unsigned long long *)ctr, 1, (unsigned long long)J.n_redirect);
    if (J.n_reject) ctr_add((unsigned long long *)ctr, 2, (unsigned long long)J.n_reject);
}

// Rare, long-running work (leader changes) is not run by one lane while 63 idle: the
// wavefront takes the lanes that need it one at a time and ALL 64 lanes execute that
// lane's (group, replica) handler in uniform mode (Lane::set_uniform), slot loops
// strided by lane.  `pending` = this lane has such a job.
// phase stamps of the cooperative jobs (read back by smr_mp_debug_stamps): only in a -DSMR_JOB_STAMPS
// build (tools/dbg_stamps.py), nothing in the shipped kernels
#ifdef SMR_JOB_STAMPS
#define JSTAMP(k) do { if (__lane_id() == 0) P.dbg[(k)] = wall_clock64(); } while (0)
/* the longest job of a kind so far (dbg[48 + kind]) and how many there were (dbg[52 + kind]) */
#define JBEGIN() const unsigned long long _jt0 = wall_clock64()
#define JEND(kind) do { if (__lane_id() == 0) { atomicMax(&P.dbg[48 + (kind)], wall_clock64() - _jt0); atomicAdd(&P.dbg[52 + (kind)], 1ull); atomicAdd(&P.dbg[56 + (kind)], wall_clock64() - _jt0); } } while (0)
#else
#define JSTAMP(k) do { } while (0)
#define JBEGIN() do { } while (0)
#define JEND(kind) do { } while (0)
#endif
#define SMR_FOR_EACH_JOB(pending, src)                                            \
    for (unsigned long long _jm = __ballot(pending); _jm; _jm &= _jm - 1)        \
        if (const int src = __ffsll((long long)_jm) - 1; true)

// Which group a lane works on.  Bulk launch (side == 0): lane = group, minus the groups the
// tick's mark pass moved to the straggler list.  Straggler launch (side = 1 + list parity): one
// listed group per wavefront, on lane 0, so that its handlers run as wave-cooperative jobs
// beside the bulk kernels instead of at their tail.
__device__ __forceinline__ bool pick_group(const MpParams &P, int side, uint32_t &g) {
    if (side == 0) {
        g = blockIdx.x * blockDim.x + threadIdx.x;
        return g < P.G && !P.overflow[g] && !P.slow[g];
    }
    const uint32_t idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t n = P.slow_n[side - 1];
    if (n > P.slow_cap) n = P.slow_cap;
    g = P.G;
    if ((threadIdx.x & 63) != 0 || idx >= n) return false;
    g = P.slow_list[idx];
    return !P.overflow[g];
}

// Which replica of its group a lane of a bulk launch stands for: block row y, rotated by the group's leader where role
// rotation is on (MpParams::role_rot).  Every (group, replica) pair is still taken exactly once per launch.
__device__ __forceinline__ uint32_t pick_replica(const MpParams &P, int side, uint32_t g) {
    uint32_t d = blockIdx.y;
    if (P.rot_on && side == 0) {
        d += P.role_rot[g < P.G ? g : 0];
        if (d >= P.R) d -= P.R;
    }
    return d;
}
// (the mark pass: the leader replica 0 knows of, which is the leader wherever no change is in flight)
__device__ __forceinline__ void mark_role(const MpParams &P, uint32_t g) {
    if (!P.rot_on) return;
    const uint32_t l = P.rep[0].leader[g];
    const uint8_t r = (uint8_t)(l < P.R ? l : 0u);
    if (P.role_rot[g] != r) P.role_rot[g] = r;
}

// start of a tick: a HearTimeout puts its group on the straggler list for `ttl` ticks
__global__ __launch_bounds__(256) void mp_mark_stragglers(const MpParams *__restrict__ Pp, int lpar,
                                                          const uint8_t *__restrict__ timeout_rep, uint32_t ttl) {
    const MpParams &P = *Pp;
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g == 0) P.slow_n[lpar ^ 1] = 0;                          // next tick's counter
    if (g >= P.G) return;
    mark_role(P, g);
    uint32_t t = P.slow_ttl[g];
    if (timeout_rep && timeout_rep[g] != NO_REP) t = ttl;
    uint8_t s = 0;
    if (t > 0) {
        const uint32_t idx = atomicAdd(&P.slow_n[lpar], 1u);
        if (idx < P.slow_cap) { P.slow_list[idx] = g; s = 1; }   // list full: the group stays with the bulk
        t--;
        P.slow_ttl[g] = (uint8_t)t;
    }
    if (P.slow[g] != s) P.slow[g] = s;
}

// ---- R1 ---------------------------------------------------------------------
__device__ __forceinline__ void r1_generic_batches(Lane &L, const uint32_t *__restrict__ req_val, uint32_t k0,
                                                   uint32_t n_req) {
    for (; k0 < n_req && !L.ovf; k0 += 8) {                      // 8 token loads per batch
        uint32_t tok[8];
#pragma unroll
        for (int k = 0; k < 8; k++) tok[k] = (k0 + k < n_req) ? req_val[(size_t)(k0 + k) * L.P.G + L.g] : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k0 + k < n_req && !L.ovf) L.req_batch(tok[k]);
    }
}

// handle_req_batch x n_req of a prepared leader in its steady state as ONE step of a wavefront in uniform mode: lane k appends
// batch k -- r1_body's store-only fast path turned by ninety degrees (same preconditions, same words stored).  For the side
// stream's batch kernel (round 5), where a listed group's lane has the wavefront to itself and every round of loads a lane
// makes is a round trip nobody hides.  false: not that state (or more batches than lanes): the caller takes the generic loop.
__device__ __forceinline__ bool r1_coop_append(Lane &J, int par, const uint32_t *__restrict__ req_val, uint32_t n_req) {
    const MpParams &P = J.P;
    const uint32_t g = J.g, r = J.me;
    if (!(J.is_leader() && J.bpd != 0 && J.bpd == J.bms && P.thresh > 1 && J.nlb >= J.len && J.abar == J.len && n_req <= 64u &&
          (J.len - J.start) + n_req - 1 + P.win_reserve < P.W))
        return false;
    J.ob_load(par);
    const uint32_t c0 = par == 0 ? J.obn0 : J.obn1;
    if (c0 + n_req > P.cap) return false;
    const RepView &v = J.v;
    const uint64_t bal = J.bpd;
    const uint32_t base = J.len, k = J.cl;
    const uint32_t m0 = SMR_ST_ACCEPTING | M_EXT | M_LBK | (VM_SAME << M_VMODE_SH) | (1u << (r + M_ACKS_SH));
    if (k < n_req) {
        const uint32_t tok = req_val[(size_t)k * P.G + g], slot = base + k;
        const size_t i = tix(P.W, slot & P.Wmask, g), o = tix(P.cap, c0 + k, g);
        v.s_val()[i] = tok; v.s_meta()[i] = m0 | (tok ? M_NONEMPTY : 0u);
        if (c0 != 0) { v.ob_slot(par)[o] = (OB_ACCEPT << OB_KIND_SH) | (slot & OB_SLOT_MASK); v.ob_bal(par)[o] = bal; }
        v.ob_val(par)[o] = tok;
    }
    if (J.brun == 0xFFFFFFFFu || J.brun > base) J.brun = base;
    J.len = base + n_req; J.abar = J.len; J.nlb = J.len;
    if (par == 0) J.obn0 = c0 + n_req; else J.obn1 = c0 + n_req;
    if (J.wr) {
        v.ob_reg(par)[g] = c0 == 0 ? base + 1 : 0u;
        if (c0 == 0) v.ob_rbal(par)[g] = bal;
    }
    return true;
}

#ifndef R1_PF
#define R1_PF 32u     // client batches per group and tick whose tokens R1 prefetches
#endif
// R1: HearTimeout -> become_a_leader; client batches -> handle_req_batch
__device__ __forceinline__ void r1_body(const MpParams &P, int par, const uint8_t *__restrict__ timeout_rep,
                                        const uint8_t *__restrict__ timeout_src, const uint8_t *__restrict__ req_target,
                                        const uint32_t *__restrict__ req_cnt, const uint32_t *__restrict__ req_val,
                                        uint32_t S, const uint32_t g, bool active, const uint32_t r, const bool force_coop = false) {
    Lane L(P, r, g < P.G ? g : 0, par);
    const bool has_to = active && timeout_rep && timeout_rep[g] == r;
    uint32_t n_req = (active && req_target && req_target[g] == r) ? req_cnt[g] : 0;
    if (n_req > S) n_req = S;
    // (round 6) the previous tick's mp_quorum_tally may have run these batches already (MpNextLocal): nothing left of them here
    const uint32_t r1_dn = P.r1_done[g < P.G ? g : 0];
    if (n_req != 0 && r1_dn != 0) { n_req = 0; P.r1_done[g] = 0; }
    active = !has_to && n_req > 0;
    // force_coop (the side stream's batch kernel): the appends too are a job of the whole wavefront, one lane per batch
    const bool app_job = force_coop && active;
    if (app_job) active = false;
    if (active) {
        // One round of loads for everything the fast path reads: the client batches' tokens (up to R1_PF of them, in
        // registers), the replica's scalars, the outbox count.  (They used to go out as seven dependent rounds -- scalars,
        // outbox count, then four batches of eight tokens -- and the leaders' wavefronts are one per SIMD: nothing else
        // hides a round trip.)
        uint32_t ptok[R1_PF];
#pragma unroll
        for (int q = 0; q < (int)R1_PF; q++) ptok[q] = ((uint32_t)q < n_req) ? req_val[(size_t)q * P.G + g] : 0u;
        L.load();
        L.ob_load(par);
        uint32_t k0 = 0;
        // Steady-state fast path: a prepared leader whose log has no holes and
        // whose accept_bar sits at the log end appends all n_req batches as a
        // tight store-only loop.  Exactly what req_batch() does per batch in that
        // state (first_null_slot -> push, fresh LeaderBookkeeping + self ack,
        // Accept bcast, accept_bar + 1), with the array pointers hoisted.
        // (bpd == bal_max_seen for every prepared leader -- a higher ballot seen deposes it (check_leader) -- and is tested so
        // that the ballot run below rests on nothing unstated)
        if (L.is_leader() && L.bpd != 0 && L.bpd == L.bms && P.thresh > 1 && L.nlb >= L.len && L.abar == L.len &&
            (L.len - L.start) + n_req - 1 + P.win_reserve < P.W) {
            L.ob_load(par);
            const uint32_t c0 = par == 0 ? L.obn0 : L.obn1;
            if (c0 + n_req <= P.cap) {
                const RepView &v = L.v;
                SMR_G uint32_t *const sv = v.s_val(); SMR_G uint32_t *const sm = v.s_meta();
                SMR_G uint32_t *const os = v.ob_slot(par); SMR_G uint64_t *const obl = v.ob_bal(par);
                SMR_G uint32_t *const ov = v.ob_val(par);
                // Leader-side ballot run (round 4): the appended slots' ballot IS bal_prepared == bal_max_seen, so it is not
                // stored -- [bal_lo, log_len) is the run the follower's append path already keeps (mp_device.h: brun), its
                // readers (the tally, r3_accept_replies, accept_reply) take bal_max_seen for a slot inside it, and whoever
                // writes another ballot or moves bal_max_seen writes the run's ballots out first (BAL_TOUCH).  8 of the 16 B
                // this loop stored per slot, and 8 of the 12 B the tally loaded per row.
                const uint64_t bal = L.bpd;
                const uint32_t base = L.len, G = P.G, Wm = P.Wmask;
                const uint32_t m0 = SMR_ST_ACCEPTING | M_EXT | M_LBK | (VM_SAME << M_VMODE_SH) | (1u << (r + M_ACKS_SH));
                auto put8 = [&](uint32_t k, const uint32_t (&tok)[8]) {
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        if (k + q >= n_req) break;
                        const uint32_t slot = base + k + q;
                        const size_t i = tix(P.W, slot & Wm, g);
                        const size_t o = tix(P.cap, c0 + k + q, g);
                        sv[i] = tok[q]; sm[i] = m0 | (tok[q] ? M_NONEMPTY : 0u);
                        if (c0 != 0) { os[o] = (OB_ACCEPT << OB_KIND_SH) | (slot & OB_SLOT_MASK); obl[o] = bal; }   // else: follow from ob_reg / ob_rbal
                        ov[o] = tok[q];
                    }
                };
#pragma unroll
                for (int kk = 0; kk < (int)R1_PF; kk += 8) {         // the first R1_PF tokens are in registers already
                    if ((uint32_t)kk >= n_req) break;
                    uint32_t tok[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) tok[q] = ptok[kk + q];
                    put8((uint32_t)kk, tok);
                }
                for (uint32_t k = R1_PF; k < n_req; k += 8) {
                    uint32_t tok[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) tok[q] = (k + q < n_req) ? req_val[(size_t)(k + q) * G + g] : 0u;
                    put8(k, tok);
                }
                if (L.brun == 0xFFFFFFFFu || L.brun > base) L.brun = base;   // the run starts here, or goes on
                L.len = base + n_req; L.abar = L.len; L.nlb = L.len;
                if (par == 0) L.obn0 = c0 + n_req; else L.obn1 = c0 + n_req;
                v.ob_reg(par)[g] = c0 == 0 ? base + 1 : 0u;
                if (c0 == 0) v.ob_rbal(par)[g] = bal;
                k0 = n_req;
            }
        }
        r1_generic_batches(L, req_val, k0, n_req);
        L.store();
    }
    SMR_FOR_EACH_JOB(has_to, src) {                             // leader change: the whole wave on one lane's group
        const uint32_t gj = __shfl(g, src), nj = __shfl(n_req, src), rj = __shfl(r, src);   // (lanes may stand for different replicas)
        Lane J(P, rj, gj, par);
        J.set_uniform();
        JBEGIN();
        JSTAMP(8);
        J.load();
        JSTAMP(9);
        J.become_a_leader(timeout_src[gj]);
        JSTAMP(10);
        r1_generic_batches(J, req_val, 0, nj);
        J.store();
        JSTAMP(11);
        JEND(0);
        flush_job(J);
    }
    SMR_FOR_EACH_JOB(app_job, src) {
        const uint32_t gj = __shfl(g, src), nj = __shfl(n_req, src), rj = __shfl(r, src);
        Lane J(P, rj, gj, par);
        J.set_uniform();
        J.load();
        if (!r1_coop_append(J, par, req_val, nj)) r1_generic_batches(J, req_val, 0, nj);
        J.store();
        flush_job(J);
    }
    flush_counters(L, active);
}

// (min wavefronts per SIMD, i.e. the register budget of the bulk round kernels; profiles/round2/r2g1_occupancy.log, r2g2: capping R2 at
// 96 VGPRs -- its rare paths then spill 548 B per lane -- takes it from 29 to 21.5 us in the steady state; R1, the rest of R3
// and R4 gain nothing from tighter caps, the tally loses)
#ifdef MP_R1_MINW
#define MP_R1_BOUNDS __launch_bounds__(256, MP_R1_MINW)
#else
#define MP_R1_BOUNDS __launch_bounds__(256)
#endif
__global__ MP_R1_BOUNDS void mp_round_local(const MpParams *__restrict__ Pp, int par,
                                                      const uint8_t *__restrict__ timeout_rep,
                                                      const uint8_t *__restrict__ timeout_src,
                                                      const uint8_t *__restrict__ req_target,
                                                      const uint32_t *__restrict__ req_cnt,
                                                      const uint32_t *__restrict__ req_val, uint32_t S, int side) {
    const MpParams &P = *Pp;
    if (!side) SMR_RAISE_PRIO();
    if (!((P.live >> blockIdx.y) & 1u)) return;                 // spread layout: that replica lives on another rank
    uint32_t g;
    const bool active = pick_group(P, side, g);
    r1_body(P, par, timeout_rep, timeout_src, req_target, req_cnt, req_val, S, g, active, pick_replica(P, side, g));
}

// ---- R2 ---------------------------------------------------------------------
#define ACK_STORE(arr, j, val) do { if ((j) >= 64u) (arr)[ack_ix(P.cap, (j), r, g)] = (val); } while (0)
template <int NR>
__device__ __forceinline__ uint64_t ack_word_from_bits(const uint64_t (&ab)[NR], uint32_t j) {   // j < 64
    uint64_t a = 0;
#pragma unroll
    for (int q = 0; q < NR; q++) a |= ((ab[q] >> j) & 1ull) << (8 * q);
    return a;
}
// every outbox but mine, sender-major, FIFO; `first_sender` / `first_j`: resume point
// left by the fast path
__device__ __forceinline__ void r2_generic(Lane &L, uint32_t first_sender, uint32_t first_j) {
    const MpParams &P = L.P;
    const uint32_t g = L.g, r = L.me;
    const int par = L.par;
    for (uint32_t s = first_sender; s < P.R && !L.ovf; s++) {
        if (s == r) continue;
        const MpRep &snd = P.rep[s];
        const uint32_t cnt = snd.ob_cnt[par][g];
        uint32_t jstart = (s == first_sender ? first_j : 0u);
        // a pure append run stores no ob_slot / ob_bal: entry j = Accept for slot ob_reg - 1 + j at ob_rbal
        const uint32_t sreg = snd.ob_reg[par][g];
        const uint64_t srbal = sreg ? snd.ob_rbal[par][g] : 0ull;
#define SND_SLOT(j, o) (sreg ? ((OB_ACCEPT << OB_KIND_SH) | ((sreg - 1 + (j)) & OB_SLOT_MASK)) : snd.ob_slot[par][o])
#define SND_BAL(o) (sreg ? srbal : snd.ob_bal[par][o])
        uint64_t abits = 0;                                      // my answers to this sender's entries < 64 (uniform in a job)
        const uint32_t ab_first = jstart;
        // Uniform mode, first choice: the whole rest of this outbox (<= 512 messages) is ONE run of
        // Accepts at one ballot >= bal_max_seen for consecutive slots that start inside or right at
        // the end of my log (the re-Accept round of a new leader followed by its new batches).  Each
        // lane takes every 64th message; all loads of the run go out in two rounds.  Per message this
        // is msg_accept(); the accept_bar scan that the message AT accept_bar starts ends behind the
        // run (every slot of it is Accepting by then) and goes on in memory.
        if (L.coop() && jstart < cnt && cnt - jstart <= 512 && !L.ovf) {
            const uint32_t n = cnt - jstart;
            uint32_t e[8], tok[8]; uint64_t bl[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t t = L.cl + 64u * u;
                const bool in = t < n;
                const size_t o = tix(P.cap, jstart + (in ? t : 0), g);
                e[u] = in ? SND_SLOT(jstart + t, o) : 0u; bl[u] = in ? SND_BAL(o) : 0ull;
                tok[u] = in ? snd.ob_val[par][o] : 0u;
            }
            const uint32_t slot0 = __shfl(e[0], 0) & OB_SLOT_MASK;
            const uint64_t bal0 = __shfl(bl[0], 0);
            bool fits = true;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t t = L.cl + 64u * u;
                if (t < n) fits = fits && e[u] == ((OB_ACCEPT << OB_KIND_SH) | ((slot0 + t) & OB_SLOT_MASK)) && bl[u] == bal0;
            }
            const uint32_t len0 = L.len;
            if (__all(fits) && bal0 >= L.bms && slot0 >= L.start && slot0 <= len0) {
                const uint32_t n_old = (len0 - slot0 < n) ? len0 - slot0 : n, n_new = n - n_old;
                const bool mine_before = L.is_leader();
                uint32_t ldr = L.leader; uint64_t bm = L.bms;
                if (bal0 > bm) { ldr = s; bm = bal0; }            // check_leader, messages.rs:313-316
                if ((len0 - L.start) + n_new <= P.W && ldr != r) {
                    (void)mine_before;
                    L.bal_touch();                                   // ballots of a re-Accept run: not tracked
                    L.leader = ldr; L.bms = bm;
                    const RepView &v = L.v;
                    uint32_t mo[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t t = L.cl + 64u * u;
                        mo[u] = (t < n_old) ? v.s_meta()[L.ix(slot0 + t)] : 0u;   // fresh instance beyond my log end
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t t = L.cl + 64u * u;
                        if (t >= n) continue;
                        const size_t i = L.ix(slot0 + t);
                        uint32_t m = m_set_st(mo[u], SMR_ST_ACCEPTING);     // :327-329
                        if (!(m & M_RBK)) m = (m | M_RBK) & ~M_RBKX;        // :331-339
                        m = m_set_src(m, s);
                        m = m_set_vmode(m, VM_SAME);                         // :351
                        m = tok[u] ? (m | M_NONEMPTY) : (m & ~M_NONEMPTY);
                        v.s_bal()[i] = bal0; v.s_val()[i] = tok[u]; v.s_meta()[i] = m;
                        ACK_STORE(snd.ack, jstart + t, 1);   // durability.rs:108-131
                    }
                    abits |= ack_range_bits(jstart, jstart + n);
                    if (n_new) {
                        if (L.nlb == len0) L.nlb = len0 + n_new;  // still no Null below the log end
                        L.len = len0 + n_new;
                    }
                    if (L.abar >= slot0 && L.abar < slot0 + n) {  // durability.rs:134-142
                        L.abar = slot0 + n;
                        while (L.abar < L.len) {
                            if (m_st(v.s_meta()[L.ix(L.abar)]) < SMR_ST_ACCEPTING) break;
                            L.abar++;
                        }
                    }
                    jstart = cnt;
                }
            }
        }
        // Uniform mode, second choice: 64 messages per step, one per lane, when they are a run as
        // above.  Per message this is msg_accept(); the accept_bar scan runs on the bitmap.
        while (L.coop() && jstart < cnt && !L.ovf) {
            const uint32_t j = jstart + L.cl;
            const bool in = j < cnt;
            const size_t o = tix(P.cap, j, g);
            const uint32_t e = in ? SND_SLOT(j, o) : 0u;
            const uint64_t bl = in ? SND_BAL(o) : 0ull;
            const uint32_t tok = in ? snd.ob_val[par][o] : 0u;   // fetched with the header: one round of loads
            const uint32_t slot = e & OB_SLOT_MASK;
            const uint32_t slot0 = __shfl(slot, 0);
            const uint64_t bal0 = __shfl(bl, 0);
            const uint32_t nin = cnt - jstart < 64 ? cnt - jstart : 64u;
            const bool fits = !in || ((e >> OB_KIND_SH) == OB_ACCEPT && bl == bal0 && slot == slot0 + L.cl);
            if (!__all(fits) || bal0 < L.bms || slot0 < L.start) break;
            const bool appending = slot0 == L.len;               // brand-new slots (push + fill fused)
            if (!appending && slot0 + nin > L.len) break;        // a mix of old and new slots: one by one
            if (appending && (L.len - L.start) + nin > P.W) break;   // ring window: let the serial path flag it
            L.bal_touch();
            L.check_leader(s, bal0);                             // messages.rs:313-316
            if (L.is_leader()) break;
            const RepView &v = L.v;
            uint32_t m = 0;
            if (in) {
                const size_t i = L.ix(slot);
                const uint32_t val = tok;
                m = appending ? 0u : v.s_meta()[i];
                m = m_set_st(m, SMR_ST_ACCEPTING);              // :327-329
                if (!(m & M_RBK)) m = (m | M_RBK) & ~M_RBKX;    // :331-339
                m = m_set_src(m, s);
                m = m_set_vmode(m, VM_SAME);                     // :351
                m = val ? (m | M_NONEMPTY) : (m & ~M_NONEMPTY);
                v.s_bal()[i] = bal0; v.s_val()[i] = val; v.s_meta()[i] = m;
                ACK_STORE(snd.ack, j, 1); // durability.rs:108-131
            }
            abits |= ack_range_bits(jstart, jstart + nin);
            // durability.rs:134-142: the completion of the slot AT accept_bar starts the scan; every
            // slot of the run is Accepting now, beyond it the scan reads memory
            if (appending) {
                if (L.nlb == L.len) L.nlb = L.len + nin;         // still no Null below the log end
                L.len += nin;
            }
            if (L.abar >= slot0 && L.abar < slot0 + nin) {
                L.abar = slot0 + nin;
                while (L.abar < L.len) {
                    if (m_st(v.s_meta()[L.ix(L.abar)]) < SMR_ST_ACCEPTING) break;
                    L.abar++;
                }
            }
            jstart += nin;
        }
        for (uint32_t j0 = jstart; j0 < cnt && !L.ovf; j0 += 8) {   // 8 messages per batch
            uint32_t e[8], val[8]; uint64_t bal[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                bool in = j0 + k < cnt;
                size_t o = tix(P.cap, j0 + k, g);
                e[k] = in ? SND_SLOT(j0 + k, o) : 0u;
                bal[k] = in ? SND_BAL(o) : 0ull;
                val[k] = in ? snd.ob_val[par][o] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (j0 + k >= cnt || L.ovf) break;
                const uint32_t j = j0 + k;
                const uint32_t kind = e[k] >> OB_KIND_SH, slot = e[k] & OB_SLOT_MASK;
                if (kind == OB_ACCEPT) {
                    uint64_t rep = L.msg_accept(s, slot, bal[k], val[k]);
                    if (L.wr) ACK_STORE(snd.ack, j, rep ? 1 : 0);   // rep == bal[k] or none
                    if (rep && j < 64u) abits |= 1ull << j;
                } else if (kind == OB_PREPARE) {
                    L.msg_prepare(s, slot, bal[k]);
                } else if (kind == OB_HEARTBEAT) {
                    L.heard_heartbeat(s, bal[k], slot, val[k], snd.ob_aux[par][tix(P.cap, j, g)]);
                }
            }
        }
        if (cnt != 0 && L.wr) {                                  // one word for this sender: the tick's first writer stores, a resumed one adds
            SMR_G uint64_t *const w = &ack_bits_base(snd.ack, P.cap, P.G)[tix(MAXR, r, g)];
            *w = ab_first == 0 ? abits : (*w | abits);
        }
    }
#undef SND_SLOT
#undef SND_BAL
}

// R2: every replica consumes the other replicas' outboxes (sender-major, FIFO)
// MODE 0: all of it (the side stream's kernels, the fused tick, quiet stretches).  Round 5, for the bulk launch beside a busy side
// stream: MODE 1 = the steady-state fast path ALONE -- a lane whose outboxes it cannot finish leaves (sender, entry) in r2_res and
// raises its tile's r2_need flag -- and MODE 2 = the cooperative jobs of exactly those lanes, a launch of its own right behind
// (mp_round_deliver_rest: exits on the flags).  The generic handlers inlined into one kernel with the fast path set its register
// budget (96 VGPRs with 852 B of scratch per lane) for work a handful of groups per tick have; alone, the fast path
// is 79 VGPRs and no scratch: six wavefronts per SIMD instead of five, and three instead of two on a SIMD that hosts a 232-VGPR
// wavefront of the side stream -- the launch fits the chip in one pass with ~110 CUs half taken.  MEASURED (profiles/r8i, same
// call A/B on the driver's command): the fast launch takes 24.2 us where the one launch took 26.8 (15.3 against 14.2 at best), and
// the rest launch costs its 4.9 us every tick: 0.0906-0.0921 ms per tick against 0.0905-0.0907.  Occupancy is NOT what the side
// stream costs R2 -- an EMPTY launch (mp_round_replies with nothing flagged) is 30 % slower beside it too (4.9-5.5 us against
// 3.7-4.0, profiles/r8c) -- so the split stays off (SMR_MP_SPLIT_R2 in the environment turns it on; the tests run both).
template <int MODE>
__device__ __forceinline__ void r2_body(const MpParams &P, int par, const uint32_t g, bool active, const uint32_t r, const bool force_coop = false) {
    Lane L(P, r, g < P.G ? g : 0, par);
    bool job = false;
    uint32_t job_sender = 0, job_j = 0;
    bool loaded = false;
    if (MODE == 2) {
        if (active) {
            const uint32_t w = P.r2_res[(size_t)r * P.G + g];
            if (w) { job = true; job_sender = (w >> 4) & 0xFu; job_j = w >> 8; P.r2_res[(size_t)r * P.G + g] = 0; }
        }
    } else if (active) {
        if (L.v.pr_cnt()[g]) L.v.pr_cnt()[g] = 0;
        uint32_t cnts[MAXR];
        uint32_t n_senders = 0, the_sender = 0, the_cnt = 0, first = MAXR;
#pragma unroll
        for (int s = 0; s < MAXR; s++) {
            cnts[s] = ((uint32_t)s < P.R && (uint32_t)s != r) ? P.rep[s].ob_cnt[par][g] : 0u;
            if (cnts[s]) { n_senders++; the_sender = (uint32_t)s; the_cnt = cnts[s]; if (first == MAXR) first = (uint32_t)s; }
        }
        uint32_t fast_done = 0;
        // Steady-state fast path: the only sender is the leader I already follow and
        // each message is an Accept at my bal_max_seen for the slot right at my log
        // end, with accept_bar at the log end too.  Per message this is exactly
        // msg_accept(): check_leader is a no-op (ballot == bal_max_seen), push + fill
        // fused (fresh ReplicaBookkeeping, voted = (ballot, reqs)), the WAL completion
        // answers with the ballot, accept_bar + 1.  The first message that does not
        // fit hands the rest of the outbox to the generic handlers.
        if (n_senders == 1 && !force_coop) {                     // (force_coop: every outbox is a job of the whole wavefront)
            L.load(); loaded = true;
            const uint32_t s = the_sender;                       // differs from lane to lane: addressed like my own arrays
            if (L.leader == s && L.abar == L.len) {
                const RepView snd{P.rep[0], (size_t)s * P.rep_stride};
                const RepView &v = L.v;
                SMR_G const uint32_t *const os = snd.ob_slot(par); SMR_G const uint64_t *const obl = snd.ob_bal(par);
                SMR_G const uint32_t *const ov = snd.ob_val(par); SMR_G uint8_t *const ack = snd.ack();
                SMR_G uint64_t *const sb = v.s_bal(); SMR_G uint32_t *const sv = v.s_val(); SMR_G uint32_t *const sm = v.s_meta();
                const uint32_t cnt = the_cnt, Wm = P.Wmask, W = P.W, start = L.start;
                const uint64_t bms = L.bms;
                const uint32_t m0 = SMR_ST_ACCEPTING | M_RBK | (s << M_SRC_SH) | (VM_SAME << M_VMODE_SH);
                uint32_t len = L.len;
                bool ok = true;
                const uint32_t reg = snd.ob_reg(par)[g];
                if (reg != 0 && reg - 1 == len && snd.ob_rbal(par)[g] == bms && (len - start) + cnt <= W) {
                    // regular outbox: Accepts for slots len, len+1, ... at my bal_max_seen, all of
                    // which fit the window -- only the batch tokens are read
                    for (uint32_t j0 = 0; j0 < cnt; j0 += 8) {
                        uint32_t val[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) val[k] = (j0 + k < cnt) ? ov[tix(P.cap, j0 + k, g)] : 0u;
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            if (j0 + k >= cnt) break;
                            const size_t i = tix(W, (len + j0 + k) & Wm, g);
                            (void)sb; (void)sm; (void)m0;            // neither ballot nor meta: the run (mp_device.h: end_run)
                            sv[i] = val[k];
                            ACK_STORE(ack, j0 + k, 1);
                        }
                    }
                    len += cnt;
                    fast_done = cnt;
                    ok = false;                                  // nothing left for the per-message loop
                }
                for (uint32_t j0 = 0; j0 < cnt && ok; j0 += 8) {
                    uint32_t e[8], val[8]; uint64_t bal[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const bool in = j0 + k < cnt;
                        const size_t o = tix(P.cap, j0 + k, g);
                        e[k] = !in ? 0u : (reg ? ((OB_ACCEPT << OB_KIND_SH) | ((reg - 1 + j0 + k) & OB_SLOT_MASK)) : os[o]);
                        bal[k] = !in ? 0ull : (reg ? snd.ob_rbal(par)[g] : obl[o]);
                        val[k] = in ? ov[o] : 0u;
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (j0 + k >= cnt || !ok) break;
                        if (e[k] != ((OB_ACCEPT << OB_KIND_SH) | (len & OB_SLOT_MASK)) || bal[k] != bms || len - start >= W) {
                            ok = false;
                            break;
                        }
                        const size_t i = tix(W, len & Wm, g);
                        (void)sb;
                        sv[i] = val[k];
                        ACK_STORE(ack, j0 + k, 1);
                        len++;
                        fast_done++;
                    }
                }
                if (L.nlb == L.len) L.nlb = len;                 // still no Null below the log end
                if (fast_done) { if (L.brun == 0xFFFFFFFFu || L.brun > L.len) L.brun = L.len; }   // appended at bal_max_seen
                L.len = len; L.abar = len;
                if (fast_done) ack_bits_base(ack, P.cap, P.G)[tix(MAXR, r, g)] = ack_range_bits(0, fast_done);   // all of them accepted
            }
        }
        // whatever is left goes to the wave as a cooperative job
        if (n_senders > 1 || (force_coop && n_senders == 1)) { job = true; job_sender = first; job_j = 0; }
        else if (n_senders == 1 && fast_done < the_cnt) { job = true; job_sender = the_sender; job_j = fast_done; }
        if (loaded) L.store();
        if (MODE == 1 && job) {                                  // ... of the launch behind this one
            P.r2_res[(size_t)r * P.G + g] = 1u | (job_sender << 4) | (job_j << 8);
            P.r2_need[(size_t)blockIdx.y * ((P.G + 63) / 64) + (g >> 6)] = 1;   // (lanes of a wavefront share a tile: same byte, same value)
            job = false;
        }
    }
    if (MODE != 1)
    SMR_FOR_EACH_JOB(job, src) {
        const uint32_t gj = __shfl(g, src), sj = __shfl(job_sender, src), jj = __shfl(job_j, src), rj = __shfl(r, src);
        Lane J(P, rj, gj, par);
        J.set_uniform();
        JBEGIN();
        JSTAMP(16);
        J.load();
        JSTAMP(17);
        r2_generic(J, sj, jj);
        JSTAMP(18);
        J.store();
        JSTAMP(19);
        JEND(1);
        flush_job(J);
    }
    flush_counters(L, active && loaded);
}

#ifndef MP_R2_MINW
#define MP_R2_MINW 5
#endif
// everything of R2 in one launch: the side stream's blocks, the spread layout, stretches without a leader change
__global__ __launch_bounds__(256, MP_R2_MINW) void mp_round_deliver_all(const MpParams *__restrict__ Pp, int par, int side) {
    const MpParams &P = *Pp;
    if (!side) SMR_RAISE_PRIO();
    if (!((P.live >> blockIdx.y) & 1u)) return;
    uint32_t g;
    const bool active = pick_group(P, side, g);
    r2_body<0>(P, par, g, active, pick_replica(P, side, g));
}
// the bulk launch beside a busy side stream: the fast path alone ...
__global__ __launch_bounds__(256) void mp_round_deliver(const MpParams *__restrict__ Pp, int par) {
    const MpParams &P = *Pp;
    if (!((P.live >> blockIdx.y) & 1u)) return;
    uint32_t g;
    const bool active = pick_group(P, 0, g);
    r2_body<1>(P, par, g, active, pick_replica(P, 0, g));
}
// ... and what it left, for the tiles whose flag is up (same grid, same lane -> (group, replica) mapping)
__global__ __launch_bounds__(256) void mp_round_deliver_rest(const MpParams *__restrict__ Pp, int par) {
    const MpParams &P = *Pp;
    if (!((P.live >> blockIdx.y) & 1u)) return;
    const uint32_t ntile = (P.G + 63) / 64, tpb = blockDim.x >> 6, t0 = blockIdx.x * tpb;
    uint32_t any = 0;
    for (uint32_t k = 0; k < tpb; k++) any |= (t0 + k < ntile) ? P.r2_need[(size_t)blockIdx.y * ntile + t0 + k] : 0u;
    if (!any) return;                                            // (block-uniform)
    uint32_t g;
    const bool active = pick_group(P, 0, g);
    r2_body<2>(P, par, g, active, pick_replica(P, 0, g));
    __syncthreads();
    if (threadIdx.x < tpb && t0 + threadIdx.x < ntile) P.r2_need[(size_t)blockIdx.y * ntile + t0 + threadIdx.x] = 0;
}

// ---- R3 ---------------------------------------------------------------------
// (a) PrepareReplies addressed to me: sender order of the tick's ackctl word, FIFO per sender
__device__ __forceinline__ void r3_prepare_replies(Lane &L, uint32_t tickctl) {
    const MpParams &P = L.P;
    const uint32_t g = L.g, d = L.me;
    for (uint32_t oi = 0; oi < P.R; oi++) {
        uint32_t s = ctl_order(tickctl, oi);
        if (s == d || s >= P.R) continue;
        const MpRep &snd = P.rep[s];
        uint32_t n = snd.pr_cnt[g];
        if (n == 0 || snd.pr_dest[g] != d) continue;
        L.prepare_reply_batch(s, snd.pr_trig[g], snd.pr_endp[g], snd.pr_bal[g], n, snd.pr_vbal, snd.pr_vval);
    }
}

// One ack-matrix row applied to its slot, branch-free, in 32-bit integer ops.
// Filter chain of handle_msg_accept_reply (messages.rs:377-412) for every reply of the
// row: reply present and == bal_prepared (:388), instance Accepting (:394), reply ballot
// >= inst.bal (:396; all valid replies equal bal_prepared, so this is one test per row),
// not a duplicate (:404-406); replies taken in the ackctl order, lost ones skipped; the
// mask freezes once popcount reaches the threshold (:412).
template <int NR>
__device__ __forceinline__ uint32_t tally_valid(uint32_t m, uint64_t b, uint32_t ctl, uint32_t valid, uint64_t bpd,
                                                uint32_t thresh, uint32_t R, bool &changed, bool &committed) {
    valid &= ~ctl_drop(ctl);                                     // bit s: replica s answered with bal_prepared
    uint32_t accepting = (m_st(m) == SMR_ST_ACCEPTING && bpd >= b) ? 1u : 0u;
    uint32_t acks = m_acks(m), chg = 0, done = 0;
#pragma unroll
    for (int oi = 0; oi < NR; oi++) {
        const uint32_t s = ctl_order(ctl, oi);                   // ids >= R never have a valid bit
        const uint32_t nb = ((valid >> s) & ~(acks >> s) & accepting & ((uint32_t)oi < R ? 1u : 0u)) & 1u;
        acks |= nb << s;
        chg |= nb;
        const uint32_t hit = nb & ((uint32_t)__popc(acks) >= thresh ? 1u : 0u);
        done |= hit;
        accepting &= ~hit;
    }
    changed = chg != 0;
    committed = done != 0;
    m = (m & ~(0xFFu << M_ACKS_SH)) | (acks << M_ACKS_SH);
    if (committed) m = m_set_st(m, SMR_ST_COMMITTED);
    return m;
}

// `aw` = the row's ack word (byte q = 1 iff replica q answered), `eb` = the ballot of the Accept the row
// is about, which is also the ballot every answer carries; `me` never answers itself
template <int NR>
__device__ __forceinline__ uint32_t tally_row(uint32_t m, uint64_t b, uint32_t ctl, uint64_t aw, uint32_t me, uint64_t eb,
                                              uint64_t bpd, uint32_t thresh, uint32_t R, bool &changed,
                                              bool &committed) {
    uint32_t valid = 0;
    if (eb != 0 && eb == bpd) {                                    // messages.rs:388: reply ballot == bal_prepared
#pragma unroll
        for (int q = 0; q < NR; q++) valid |= ((uint32_t)(aw >> (8 * q)) & 1u) << q;
        valid &= ~(1u << me);
    }
    return tally_valid<NR>(m, b, ctl, valid, bpd, thresh, R, changed, committed);
}

// (b) AcceptReplies to my Accepts of this tick: the ack matrix of my outbox, entry-major,
// per-entry peer order / loss from ackctl; C rows per batch of loads, tally in registers
template <int NR>   // NR >= population: replica columns held in registers
__device__ __forceinline__ void r3_accept_replies(Lane &L, const uint32_t *__restrict__ ackctl, uint32_t cnt) {
    const MpParams &P = L.P;
    const uint32_t g = L.g, d = L.me;
    const int par = L.par;
    const RepView &v = L.v;
    constexpr int C = 8;                                         // ack-matrix rows per batch of loads
    SMR_G const uint32_t *const os = v.ob_slot(par);
    SMR_G const uint64_t *const ackw = (SMR_G const uint64_t *)v.ack();   // one word per (entry, group), byte q = replica q
    uint64_t ab[NR];                                             // entries < 64: one word per follower instead
#pragma unroll
    for (int q = 0; q < NR; q++) ab[q] = (uint32_t)q < P.R ? ack_bits_base(v.ack(), P.cap, P.G)[tix(MAXR, q, g)] : 0ull;
    SMR_G const uint64_t *const obl = v.ob_bal(par);
    SMR_G uint32_t *const sm = v.s_meta(); SMR_G const uint64_t *const sb = v.s_bal();
    const uint32_t G = P.G, Wm = P.Wmask, R = P.R, thresh = P.thresh;
    const bool lead = L.is_leader();
    const uint64_t bpd = L.bpd;
    if (L.coop()) {
        // Uniform mode (long outbox of a re-Accept round): 64 ack-matrix rows per step, one per
        // lane, tallied in parallel; rows that reach the quorum are then committed one by one in
        // entry order -- their Committed status is only written then, so every commit-bar run sees
        // exactly the slots an entry-by-entry replay would have committed so far.
        for (uint32_t j0 = 0; j0 < cnt; j0 += 64) {
            const uint32_t j = j0 + L.cl;
            const bool in = j < cnt;
            const size_t o = tix(P.cap, j, g);
            const uint32_t creg = v.ob_reg(par)[g];
            const uint32_t e = !in ? 0u : (creg ? ((OB_ACCEPT << OB_KIND_SH) | ((creg - 1 + j) & OB_SLOT_MASK)) : os[o]);
            const uint64_t eb = !in ? 0ull : (creg ? v.ob_rbal(par)[g] : obl[o]);
            const uint32_t ctl = (in && ackctl) ? ackctl[(size_t)j * G + g] : SMR_CTL_IDENTITY;
            const uint64_t a = !in ? 0ull : (j < 64u ? ack_word_from_bits<NR>(ab, j) : ackw[o]);
            const uint32_t slot = e & OB_SLOT_MASK;
            const bool have = in && (e >> OB_KIND_SH) == OB_ACCEPT && slot >= L.start && slot < L.len;
            const size_t i = tix(P.W, slot & Wm, g);
            const uint32_t m0 = have ? sm[i] : 0u;
            const uint64_t b = !have ? 0ull : (slot >= L.brun ? L.bms : sb[i]);   // (a leader's ballot run: unstored)
            uint32_t mk = m0;
            bool changed = false, committed = false;
            if (have && lead && (mk & M_LBK)) mk = tally_row<NR>(m0, b, ctl, a, d, eb, bpd, thresh, R, changed, committed);
            if (changed && !committed) sm[i] = mk;               // every lane owns its row's slot
            unsigned long long cm = __ballot(changed && committed);
            {
                // Common shape of a re-Accept round: the step's rows are consecutive slots starting at
                // commit_bar == exec_bar, all below accept_bar, all non-empty, and ALL reach the
                // quorum.  Replayed one by one each would take the register fast path above (commit,
                // execute, both bars + 1), so the whole step is: every slot Executed, bars + n.
                const uint32_t nin = cnt - j0 < 64 ? cnt - j0 : 64u;
                const unsigned long long full = nin == 64 ? ~0ull : ((1ull << nin) - 1ull);
                const uint32_t slot0 = __shfl(slot, 0);
                const bool shape = !in || (have && slot == slot0 + L.cl && (mk & M_NONEMPTY));
                if (cm == full && __all(shape) && slot0 == L.cbar && slot0 + nin <= L.abar && P.clist_cap == 0) {
                    bool tail_ok = slot0 + nin >= L.abar && slot0 + nin >= L.len;   // run ends with the log ...
                    if (!tail_ok && slot0 + nin < L.len)
                        tail_ok = m_st(sm[tix(P.W, (slot0 + nin) & Wm, g)]) < SMR_ST_COMMITTED;   // next slot not committed yet
                    if (tail_ok) {
                        if (in) sm[i] = m_set_st(mk, SMR_ST_EXECUTED);
                        L.n_commit += nin;
                        L.cbar = slot0 + nin;
                        if (L.ebar >= slot0 && L.ebar < slot0 + nin) L.ebar = slot0 + nin;   // rides from the row at it
                        cm = 0;
                    }
                }
            }
            for (; cm; cm &= cm - 1) {
                const int src = __ffsll((long long)cm) - 1;
                const int nx = src < 63 ? src + 1 : src;
                const uint32_t cs = __shfl(slot, src), cmk = __shfl(mk, src);
                const uint32_t ns = __shfl(slot, nx), nm = __shfl(m0, nx);
                const bool next_known = src < 63 && __shfl((int)have, nx) && ns == cs + 1;
                const size_t ci = tix(P.W, cs & Wm, g);
                L.record_commit(cs);
                const bool stops = (next_known && m_st(nm) < SMR_ST_COMMITTED) || (cs + 1 >= L.abar && cs + 1 >= L.len);
                if (cs == L.cbar && cs < L.abar && stops) {
                    if (L.wr) sm[ci] = m_set_st(cmk, SMR_ST_EXECUTED);
                    L.cbar = cs + 1;
                    if ((cmk & M_NONEMPTY) && cs == L.ebar) L.ebar = cs + 1;   // execution.rs:70
                } else {
                    if (L.wr) sm[ci] = cmk;
                    L.commit_complete<2>(cs, cmk, next_known ? nm : 0xFFFFFFFFu);
                }
            }
        }
        L.ob_set(par, 0);
        return;
    }
    // regular outbox (steady-state appends): entry j is the Accept for slot (reg - 1) + j, so
    // nothing depends on ob_slot and all loads of a batch go out together
    const uint32_t reg = v.ob_reg(par)[g];
    const uint64_t rbal = reg ? v.ob_rbal(par)[g] : 0ull;
    for (uint32_t j0 = 0; j0 < cnt; j0 += C) {
        uint32_t e[C], ctl[C], m[C];
        uint64_t b[C], eb[C], a[C];
        bool have[C];
#pragma unroll
        for (int k = 0; k < C; k++) {                            // wave 1: everything addressed by (j, g) alone
            const bool in = j0 + k < cnt;
            const size_t o = tix(P.cap, j0 + k, g);
            e[k] = !in ? 0u : (reg ? ((OB_ACCEPT << OB_KIND_SH) | ((reg - 1 + j0 + k) & OB_SLOT_MASK)) : os[o]);
            eb[k] = !in ? 0ull : (reg ? rbal : obl[o]);
            ctl[k] = (in && ackctl) ? ackctl[(size_t)(j0 + k) * G + g] : SMR_CTL_IDENTITY;
            a[k] = !in ? 0ull : (j0 + k < 64u ? ack_word_from_bits<NR>(ab, j0 + k) : ackw[o]);
        }
#pragma unroll
        for (int k = 0; k < C; k++) {                            // wave 2: the slots those Accepts name
            const uint32_t slot = e[k] & OB_SLOT_MASK;
            have[k] = (e[k] >> OB_KIND_SH) == OB_ACCEPT && slot >= L.start && slot < L.len;
            const size_t i = tix(P.W, slot & Wm, g);
            m[k] = have[k] ? sm[i] : 0u;
            b[k] = !have[k] ? 0ull : (slot >= L.brun ? L.bms : sb[i]);
        }
#pragma unroll
        for (int k = 0; k < C; k++) {
            if (!have[k]) continue;
            const uint32_t slot = e[k] & OB_SLOT_MASK;
            const int kn = k + 1 < C ? k + 1 : k;
            const bool next_known = k + 1 < C && have[kn] && (e[kn] & OB_SLOT_MASK) == slot + 1;
            // The row's replies applied in registers: same filter chain as
            // accept_reply() / accept_entry() (messages.rs:377-412).
            uint32_t mk = m[k];
            if (!lead || !(mk & M_LBK)) continue;
            bool changed = false, committed = false;
            mk = tally_row<NR>(mk, b[k], ctl[k], a[k], d, eb[k], bpd, thresh, R, changed, committed);
            if (!changed) continue;
            const size_t i = tix(P.W, slot & Wm, g);
            if (!committed) { if (L.wr) sm[i] = mk; continue; }
            L.record_commit(slot);
            // commit_complete() in its common shape, in registers: the slot sits at
            // commit_bar == exec_bar below accept_bar with a non-empty batch, and the
            // run ends right behind it (next slot still Accepting, or the log ends).
            const bool stops = (next_known && m_st(m[kn]) < SMR_ST_COMMITTED) || (slot + 1 >= L.abar && slot + 1 >= L.len);
            if (slot == L.cbar && slot < L.abar && stops) {
                if (L.wr) sm[i] = m_set_st(mk, SMR_ST_EXECUTED);
                L.cbar = slot + 1;
                if ((mk & M_NONEMPTY) && slot == L.ebar) L.ebar = slot + 1;   // execution.rs:70
            } else {
                if (L.wr) sm[i] = mk;
                L.commit_complete<2>(slot, mk, next_known ? m[kn] : 0xFFFFFFFFu);
            }
        }
    }
    L.ob_set(par, 0);                                            // outbox consumed
}

__device__ __forceinline__ void r3_publish_hb(Lane &L) {         // leadership.rs:240-247 record
    const RepView &v = L.v;
    if (L.wr) { v.hb_bal()[L.g] = L.bms; v.hb_commit()[L.g] = L.cbar; v.hb_exec()[L.g] = L.ebar; v.hb_snap()[L.g] = L.snap; }
}

// R3: replies reach their destination: PrepareReplies (sender order of the
// tick's ackctl word, FIFO per sender), then the AcceptReply matrix of my own
// outbox, entry-major with per-entry peer order / loss.  THE quorum kernel, in two launches:
//
//  * mp_quorum_tally -- the steady state: a prepared leader whose outbox is "regular"
//    (<= 64 Accepts for consecutive slots at bal_prepared).  Block = 4 wavefronts over the
//    SAME 64 groups; each tallies a quarter of the ack-matrix rows (every load of a row batch
//    is independent of the others) and leaves the per-row outcome in LDS.  After the barrier:
//    if every row reached the quorum, the rows are the consecutive slots at commit_bar ==
//    exec_bar and all batches are non-empty, then an entry-by-entry replay would commit and
//    execute each in turn (both bars + 1 per row), so each wavefront writes its rows Executed
//    and wavefront 0 moves the bars; otherwise wavefront 0 replays the commits in entry order.
//  * mp_round_replies -- every other lane (irregular outbox, non-leader, PrepareReplies, the
//    long outbox of a re-Accept round): per lane, or as a cooperative job for the wavefront.
// which replicas have PrepareReplies addressed to them (bit d), for group g
__device__ __forceinline__ uint32_t r3_pr_dest_mask(const MpParams &P, uint32_t g) {
    uint32_t mask = 0;
#pragma unroll
    for (int s = 0; s < MAXR; s++)
        if ((uint32_t)s < P.R && P.rep[s].pr_cnt[g] != 0) mask |= 1u << P.rep[s].pr_dest[g];
    return mask;
}

// the same array of another replica: the arena lays every replica out identically
template <typename T>
__device__ __forceinline__ T *rep_shift(T *p0, size_t byte_off) { return (T *)((SMR_G char *)p0 + byte_off); }

// Block = 4 wavefronts over the same 64 groups, ALL replicas: every lane works for the one
// replica of its group that has a steady-state tally to do (its prepared leader), so there is
// no block without work.  The kernel also publishes the heartbeat record of every replica whose
// round is already complete, and tells mp_round_replies which (replica, tile) pairs still need it.
// Loads are arranged in three dependent rounds: (1) who has an outbox, (2) that replica's
// scalars + its ack-matrix rows, (3) the ring rows.
// TALLY_SPEC: the round-2 loads (scalars + ack words) of replica `hint` -- the job's preset leader, wave-uniform --
// go out WITH round 1; a lane whose replica turns out to be `hint` (all of them in the steady state) starts its ring
// rows as soon as round 1 lands: two dependent load rounds instead of three.  Other lanes reload as before.
// TALLY_WAVEFLAGS: a wavefront ANDs the flags of its own rows in a register and hands over ONE byte per lane, so the
// closed-form test reads 4 LDS bytes per lane instead of one per row.
#ifndef TALLY_SPEC
#define TALLY_SPEC 0
#endif
#ifndef TALLY_WAVEFLAGS
#define TALLY_WAVEFLAGS 0
#endif
template <int NR>
__device__ __forceinline__ void quorum_tally_block(const MpParams &P, int par, const uint32_t *__restrict__ ackctl,
                                                   int publish_hb, uint8_t *sh_fl, uint32_t *sh_mk, uint32_t hint,
                                                   const MpNextLocal nx = MpNextLocal{nullptr, nullptr, nullptr, nullptr, 0u}) {
#ifndef TALLY_C
#define TALLY_C 4          // rows per wavefront and pass.  Since round 4 a row in flight is its meta word alone -- the ballots of the leader's
#endif                     // run are not stored (one bit per row says "bal_prepared >= the slot's ballot", loaded only for rows outside
                           // the run) and the answers are taken from the followers' bit words as they are needed: 82 VGPRs (was 96).
                           // Same-call A/Bs (profiles/r5e_tally_rows_ab.log, r5f_tally_w6_rsp_stores.log): 8 rows per pass (one pass
                           // for S = 32, 102 VGPRs) the same in the steady state and worse beside the side stream's blocks (27.9 vs
                           // 26.4 us); capped at 80 VGPRs for a sixth wavefront per SIMD (-DTALLY_MINW=6, 12 B of scratch): no
                           // difference.  (Round 2: profiles/round2/r2u_tally_rows_per_pass.log.)
    constexpr int C = NR <= 5 ? TALLY_C : 4;                    // rows per wavefront per pass
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (w >= 4) { __syncthreads(); return; }                    // (the fused tick kernel's block has a wavefront per replica)
    const uint32_t g = blockIdx.x * 64 + lane;
    const uint32_t gg = g < P.G ? g : 0;
    const uint32_t R = P.R, G = P.G, Wm = P.Wmask, thresh = P.thresh;
    // ---- round 1: every replica's outbox count and who is owed PrepareReplies --------------------
    const uint32_t ovf = (uint32_t)P.overflow[gg] | (uint32_t)P.slow[gg];
    // (every load of the round goes out before anything is looked at: the addresses are replica 0's arrays plus a uniform
    // multiple of rep_stride, clamped to a real replica, so no load sits behind a branch -- the per-replica `d < R ? load : 0`
    // this replaces compiled to one basic block per replica with a wait on its pr_cnt at the end: R serial round trips)
    const MpRep &v0 = P.rep[0];
    uint32_t cnts[NR], prc[NR], prd[NR];
#pragma unroll
    for (int d = 0; d < NR; d++) {
        const size_t rd = (size_t)((uint32_t)d < R ? d : 0) * P.rep_stride;
        cnts[d] = rep_shift(v0.ob_cnt[par], rd)[gg];
        prc[d] = rep_shift(v0.pr_cnt, rd)[gg];
        prd[d] = rep_shift(v0.pr_dest, rd)[gg];
    }
    uint32_t ctl[C];                                             // the reply order / loss words of my first C rows
#pragma unroll
    for (int k = 0; k < C; k++) {
        const uint32_t j = w + 4u * (uint32_t)k;
        ctl[k] = (ackctl && j < P.cap) ? ackctl[(size_t)j * G + gg] : SMR_CTL_IDENTITY;
    }
    // (round 6) the NEXT tick's client batches of this group: whom they are addressed to, how many, and that replica's HearTimeout
    uint32_t nx_tgt = NO_REP, nx_cnt = 0, nx_to = NO_REP;
    if (nx.req_target) {
        nx_tgt = nx.req_target[gg]; nx_cnt = nx.req_cnt[gg];
        if (nx.timeout_rep) nx_to = nx.timeout_rep[gg];
    }
#if TALLY_SPEC
    const size_t roh = (size_t)(hint < R ? hint : 0u) * P.rep_stride;   // wave-uniform
    uint64_t abh[NR];
#pragma unroll
    for (int q = 0; q < NR; q++)
        abh[q] = ack_bits_base(rep_shift(v0.ack, roh), P.cap, P.G)[tix(MAXR, q, gg)];   // (rows R.. MAXR exist: masked below)
    const uint32_t h_reg = rep_shift(v0.ob_reg[par], roh)[gg], h_leader = rep_shift(v0.leader, roh)[gg];
    const uint64_t h_bpd = rep_shift(v0.bal_prepared, roh)[gg], h_rbal = rep_shift(v0.ob_rbal[par], roh)[gg];
    const uint32_t h_start = rep_shift(v0.start_slot, roh)[gg], h_len = rep_shift(v0.log_len, roh)[gg];
    const uint32_t h_cbar = rep_shift(v0.commit_bar, roh)[gg], h_ebar = rep_shift(v0.exec_bar, roh)[gg];
    const uint32_t h_abar = rep_shift(v0.accept_bar, roh)[gg];
    const uint32_t h_brun = rep_shift(v0.bal_lo, roh)[gg];
    const uint64_t h_bms = rep_shift(v0.bal_max_seen, roh)[gg];
#endif
    const bool active = g < G && !ovf;                          // ovf: frozen, or on the straggler list
    uint32_t prmask = 0;
#pragma unroll
    for (int d = 0; d < NR; d++) {
        const bool in = (uint32_t)d < R;
        if (!in || !active || !((P.live >> d) & 1u)) cnts[d] = 0;   // (an image's outbox is its own rank's to tally)
        if (in && active && prc[d] != 0) prmask |= 1u << prd[d];
    }
    prmask &= P.live;
    // my replica: the lowest one with a non-empty outbox (a second one, if any, is left to mp_round_replies)
    uint32_t dl = R, cnt = 0;
#pragma unroll
    for (int d = NR - 1; d >= 0; d--) if (cnts[d] != 0) { dl = (uint32_t)d; cnt = cnts[d]; }
    uint32_t nzmask = 0;                                         // bit d: replica d has a non-empty outbox (all that is needed of cnts[] below)
#pragma unroll
    for (int d = 0; d < NR; d++) nzmask |= cnts[d] != 0 ? 1u << d : 0u;
    const bool cand = dl < R && !((prmask >> dl) & 1u) && cnt <= 64;
    const size_t ro = (size_t)(dl < R ? dl : 0) * P.rep_stride;  // per-lane replica: addresses are vector values
#if TALLY_SPEC
    const bool spec = cand && dl == hint;
#else
    constexpr bool spec = false;
#endif
    SMR_G uint32_t *const sm = rep_shift(v0.s_meta, ro);
    SMR_G const uint64_t *const sb = rep_shift(v0.s_bal, ro);
    uint64_t ab[NR];                                             // my replica's followers, entries < 64 (cand: cnt <= 64)
#pragma unroll
    for (int q = 0; q < NR; q++)
        ab[q] = (cand && !spec && (uint32_t)q < R) ? ack_bits_base(rep_shift(v0.ack, ro), P.cap, P.G)[tix(MAXR, q, gg)] : 0ull;
    SMR_G uint32_t *const p_cbar = rep_shift(v0.commit_bar, ro), *const p_ebar = rep_shift(v0.exec_bar, ro);
    // Rows are dealt to the four wavefronts round robin -- wavefront w takes rows w, w + 4, w + 8, ... -- so WHICH rows a
    // wavefront tallies depends on nothing it has to load first: their ackctl words went out with round 1 above.
    // ---- round 2: my replica's scalars -------------------------------------------------------------
    uint32_t reg = 0, leader = NO_REP, start = 0, len = 0, cbar = 0, ebar = 0, abar = 0, brun = 0xFFFFFFFFu;
    uint64_t bpd = 0, bms = 0;
    uint64_t rbal = 0;
#if TALLY_SPEC
    if (spec) {
#pragma unroll
        for (int q = 0; q < NR; q++) ab[q] = (uint32_t)q < R ? abh[q] : 0ull;
        reg = h_reg; bpd = h_bpd; rbal = h_rbal; leader = h_leader; start = h_start; len = h_len;
        cbar = h_cbar; ebar = h_ebar; abar = h_abar; brun = h_brun; bms = h_bms;
    }
#endif
    if (cand && !spec) {
        reg = rep_shift(v0.ob_reg[par], ro)[gg]; bpd = rep_shift(v0.bal_prepared, ro)[gg];
        rbal = rep_shift(v0.ob_rbal[par], ro)[gg];
        leader = rep_shift(v0.leader, ro)[gg];
        start = rep_shift(v0.start_slot, ro)[gg]; len = rep_shift(v0.log_len, ro)[gg];
        cbar = p_cbar[gg]; ebar = p_ebar[gg]; abar = rep_shift(v0.accept_bar, ro)[gg];
        brun = rep_shift(v0.bal_lo, ro)[gg]; bms = rep_shift(v0.bal_max_seen, ro)[gg];   // the leader's ballot run (R1)
    }
    const bool fast4 = cand && reg != 0 && bpd != 0 && leader == dl;
    // (round 6) mp_round_local's steady-state fast path of the NEXT tick, decided on what this tick's closed form leaves behind:
    // fold_n = the batches my replica will append below, 0 = that tick's R1 launch does it (r1_body's preconditions, one by one;
    // the closed form adds accept_bar == log end and an empty outbox of this parity, and moves nothing r1_body looks at)
    uint32_t fold_n = 0;
    if (nx.req_target && fast4 && nx_tgt == dl && nx_to != dl && !publish_hb) {
        const uint32_t nlb = rep_shift(v0.null_lb, ro)[gg], c0n = rep_shift(v0.ob_cnt[par ^ 1], ro)[gg];
        const uint32_t n_req = nx_cnt < nx.S ? nx_cnt : nx.S;
        if (n_req != 0 && bpd == bms && thresh > 1 && nlb >= len && abar == len && c0n == 0 && n_req <= P.cap &&
            (len - start) + n_req - 1 + P.win_reserve < P.W)
            fold_n = n_req;
    }
    // (what the loop needs of the 64-bit scalars, as bits: they are dead from here on -- the kernel is 2 VGPRs from a sixth wavefront)
    const bool run_bal_ok = bpd >= bms;                          // a slot of the run: bal_prepared >= its ballot (= bal_max_seen)
    const bool answers_ok = rbal != 0 && rbal == bpd;            // an answer carries the Accept's ballot (ob_rbal): messages.rs:388
    uint32_t wall = 0xFF;                                        // TALLY_WAVEFLAGS: AND of my rows' flags
    (void)wall;
    // ---- round 3 + tally: C rows per pass (one pass unless the outbox is longer than 4 * C) -------
#pragma unroll 1
    for (uint32_t k0 = 0; w + 4u * k0 < cnt; k0 += C) {
        if (k0 != 0) {
#pragma unroll
            for (int k = 0; k < C; k++) {
                const uint32_t j = w + 4u * (k0 + k);
                ctl[k] = (cand && j < cnt && ackctl) ? ackctl[(size_t)j * G + g] : SMR_CTL_IDENTITY;
            }
        }
        uint32_t m[C];
        // the followers' answer bits of this pass's rows (j = w + 4 (k0 + k): bit 4 k of the word shifted down to the pass's first
        // row), 32 bits wide: one 64-bit shift per follower and pass instead of one per follower and ROW (quarter rate on this chip)
        uint32_t a32[NR];
#pragma unroll
        for (int q = 0; q < NR; q++) a32[q] = (uint32_t)(ab[q] >> (w + 4u * k0));
        static_assert(4 * (C - 1) < 32, "a pass's rows must fit the 32-bit window");
        // bit k: bal_prepared >= the ballot of row k's slot (messages.rs:396).  Inside the leader's run the ballot is
        // bal_max_seen, unstored: one comparison for all rows; a row below the run (none in the steady state: the run starts
        // with the first steady append) loads its ballot -- one by one, so that no row holds two more registers in flight
        uint32_t bok = run_bal_ok ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int k = 0; k < C; k++) {
            const uint32_t j = w + 4u * (k0 + k);
            const uint32_t slot = reg - 1 + j;
            const bool have = fast4 && j < cnt && slot >= start && slot < len;
            const size_t i = tix(P.W, slot & Wm, g);
            m[k] = have ? sm[i] : 0xFFFFFFFFu;
        }
        if (fast4 && reg - 1 + w + 4u * k0 < brun) {                 // (rows are consecutive slots: the pass's first row tells)
#pragma unroll 1
            for (int k = 0; k < C; k++) {
                const uint32_t j = w + 4u * (k0 + k);
                const uint32_t slot = reg - 1 + j;
                if (!(j < cnt && slot >= start && slot < len) || slot >= brun) continue;
                const bool ok = bpd >= sb[tix(P.W, slot & Wm, g)];
                bok = ok ? (bok | (1u << k)) : (bok & ~(1u << k));
            }
        }
#pragma unroll
        for (int k = 0; k < C; k++) {
            const uint32_t j = w + 4u * (k0 + k);
            if (!fast4 || j >= cnt) continue;
            const bool have = m[k] != 0xFFFFFFFFu;
            uint32_t mk = have ? m[k] : 0u;
            bool changed = false, committed = false;
            if (have && (mk & M_LBK)) {
                uint32_t valid = 0;                              // an answer carries the Accept's ballot: ob_rbal
                if (answers_ok) {
#pragma unroll
                    for (int q = 0; q < NR; q++) valid |= ((a32[q] >> (4 * k)) & 1u) << q;        // (cand: j < cnt <= 64)
                    valid &= ~(1u << dl);
                }
                // (tally_valid's ballot test is `bpd >= b`: b = 0 passes, b = ~0 fails)
                mk = tally_valid<NR>(mk, ((bok >> k) & 1u) ? 0ull : ~0ull, ctl[k], valid, bpd, thresh, R, changed, committed);
            }
            // a row short of the quorum keeps its new acks; re-tallying it later changes nothing
            if (changed && !committed) sm[tix(P.W, (reg - 1 + j) & Wm, g)] = mk;
            sh_mk[j * 64 + lane] = mk;
            const uint32_t fl = (have ? 1u : 0u) | (changed ? 2u : 0u) | (committed ? 4u : 0u) | ((mk & M_NONEMPTY) ? 16u : 0u);
#if TALLY_WAVEFLAGS
            wall &= fl;
#else
            sh_fl[j * 64 + lane] = (uint8_t)fl;
#endif
        }
    }
#if TALLY_WAVEFLAGS
    sh_fl[w * 64 + lane] = (uint8_t)wall;                        // (a wavefront without rows hands over 0xFF)
#endif
    // (round 6) my share of the next tick's batch tokens -- batches w, w + 4, ... -- in flight across the barrier
    constexpr int NXQ = 8;
    uint32_t nxtok[NXQ];
#pragma unroll
    for (int q = 0; q < NXQ; q++) {
        const uint32_t k = w + 4u * (uint32_t)q;
        nxtok[q] = k < fold_n ? nx.req_val[(size_t)k * G + gg] : 0u;
    }
    __syncthreads();
    // ---- the all-commit closed form, or leave the lane to mp_round_replies ------------------------
    bool closed = false;
    const uint32_t first = reg - 1;
    if (fast4) {
#if TALLY_WAVEFLAGS
        const uint32_t all = (uint32_t)sh_fl[lane] & sh_fl[64 + lane] & sh_fl[128 + lane] & sh_fl[192 + lane];
#else
        uint32_t all = 0xFF;
        for (uint32_t j = 0; j < cnt; j++) all &= sh_fl[j * 64 + lane];
#endif
        // every row: present, changed, committed, non-empty; rows = slots commit_bar.., all below
        // accept_bar, and the run ends behind the last one (accept_bar and log end).  exec_bar rides
        // along from the row that sits AT it (execution.rs:70), if any -- a pinned exec_bar stays.
        closed = (all & 23) == 23 && first == cbar && first + cnt == abar && first + cnt >= len && P.clist_cap == 0;
        if (closed)
            for (uint32_t j = w; j < cnt; j += 4)
                sm[tix(P.W, (first + j) & Wm, g)] = m_set_st(sh_mk[j * 64 + lane], SMR_ST_EXECUTED);
    }
    // which replicas of my group still need mp_round_replies: a non-empty outbox I did not close,
    // or PrepareReplies waiting; everybody else's round is complete
    const uint32_t need = prmask | (closed ? nzmask & ~(1u << dl) : nzmask);
    if (w == 0 && closed) {
        p_cbar[gg] = first + cnt;
        if (ebar >= first && ebar < first + cnt) p_ebar[gg] = first + cnt;
        rep_shift(v0.ob_cnt[par], ro)[gg] = 0;                 // outbox consumed: nothing left for mp_round_replies
    }
    // (round 6) the group's tick is complete (nobody of it goes to mp_round_replies): my replica's handle_req_batch calls of the
    // NEXT tick, exactly r1_body's store-only fast path -- slots len .. len + n - 1 Accepting with my self-ack, the regular outbox
    // of the next parity (ob_reg / ob_rbal name slot and ballot, only the tokens are stored), the run of unstored ballots
    // extended, accept_bar and log end + n -- with the four wavefronts taking every fourth batch.  r1_done tells that tick's R1.
    const bool folded = closed && need == 0 && fold_n != 0;
    if (folded) {
        const int np = par ^ 1;
        SMR_G uint32_t *const sv = rep_shift(v0.s_val, ro);
        SMR_G uint32_t *const ov = rep_shift(v0.ob_val[np], ro);
        const uint32_t m0 = SMR_ST_ACCEPTING | M_EXT | M_LBK | (VM_SAME << M_VMODE_SH) | (1u << (dl + M_ACKS_SH));
#pragma unroll
        for (int q = 0; q < NXQ; q++) {
            const uint32_t k = w + 4u * (uint32_t)q;
            if (k >= fold_n) break;
            const size_t i = tix(P.W, (len + k) & Wm, g);
            sv[i] = nxtok[q]; sm[i] = m0 | (nxtok[q] ? M_NONEMPTY : 0u);
            ov[tix(P.cap, k, g)] = nxtok[q];
        }
        for (uint32_t k = w + 4u * NXQ; k < fold_n; k += 4) {    // (more than 32 batches per tick)
            const uint32_t tok = nx.req_val[(size_t)k * G + gg];
            const size_t i = tix(P.W, (len + k) & Wm, g);
            sv[i] = tok; sm[i] = m0 | (tok ? M_NONEMPTY : 0u);
            ov[tix(P.cap, k, g)] = tok;
        }
        if (w == 0) {
            if (brun == 0xFFFFFFFFu || brun > len) rep_shift(v0.bal_lo, ro)[gg] = len;
            rep_shift(v0.log_len, ro)[gg] = len + fold_n;
            rep_shift(v0.accept_bar, ro)[gg] = len + fold_n;
            rep_shift(v0.null_lb, ro)[gg] = len + fold_n;
            rep_shift(v0.ob_cnt[np], ro)[gg] = fold_n;
            rep_shift(v0.ob_reg[np], ro)[gg] = len + 1;
            rep_shift(v0.ob_rbal[np], ro)[gg] = rep_shift(v0.ob_rbal[par], ro)[gg];   // = bal_prepared (answers_ok: every row committed)
            P.r1_done[gg] = 1;
        }
    }
    if (publish_hb && active) {                                 // leadership.rs:240-247 record, complete rounds only
        for (uint32_t d = w; d < R; d += 4) {                   // replicas dealt over the four wavefronts
            if (((need >> d) & 1u) || !((P.live >> d) & 1u)) continue;   // mp_round_replies publishes after its work
            const MpRep &u = P.rep[d];
            uint32_t cb = u.commit_bar[gg], eb = u.exec_bar[gg];
            if (closed && d == dl) {                            // wavefront 0's stores above may not have landed
                cb = first + cnt;
                eb = (ebar >= first && ebar < first + cnt) ? first + cnt : ebar;
            }
            u.hb_bal[gg] = u.bal_max_seen[gg]; u.hb_commit[gg] = cb; u.hb_exec[gg] = eb;
            u.hb_snap[gg] = u.snap_bar[gg];
        }
    }
    if (w != 0) return;
    // commit counter of each lane's replica: one sum when the whole wavefront serves one replica
    {
        const unsigned int nc = closed ? cnt : 0u;
        const uint32_t d0 = __shfl(dl, __ffsll((long long)(__ballot(closed) | (1ull << 63))) - 1);
        if (__all(!closed || dl == d0)) {
            unsigned int x = nc;
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if (lane == 0 && x) ctr_add((unsigned long long *)P.rep[d0].counters, 0, (unsigned long long)x);
        } else {
            for (uint32_t d = 0; d < R; d++) {
                unsigned int x = (closed && dl == d) ? nc : 0u;
                for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
                if (lane == 0 && x) ctr_add((unsigned long long *)P.rep[d].counters, 0, (unsigned long long)x);
            }
        }
    }
    if (__any(folded)) {                                         // counter 4: client batches appended here instead of in R1 (a performance figure)
        for (uint32_t d = 0; d < R; d++) {
            if (!__any(folded && dl == d)) continue;
            unsigned int x = (folded && dl == d) ? fold_n : 0u;
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if (lane == 0 && x) ctr_add((unsigned long long *)P.rep[d].counters, 4, (unsigned long long)x);
        }
    }
    uint32_t nd = need;
    for (int off = 32; off > 0; off >>= 1) nd |= __shfl_xor(nd, off);
    if (lane == 0) {
        const uint32_t ntile = (G + 63) / 64;
        for (uint32_t k = 0; k < R; k++) P.r3_need[(size_t)k * ntile + blockIdx.x] = (uint8_t)((nd >> k) & 1u);
    }
}

#ifndef TALLY_MINW
#define TALLY_MINW 1
#endif
template <int NR>
__global__ __launch_bounds__(256, TALLY_MINW) void mp_quorum_tally(const MpParams *__restrict__ Pp, int par,
                                                       const uint32_t *__restrict__ ackctl, int publish_hb, uint32_t hint,
                                                       const MpNextLocal nx) {
    __shared__ uint8_t sh_fl[64 * 64];
    __shared__ uint32_t sh_mk[64 * 64];
    SMR_RAISE_PRIO();
    quorum_tally_block<NR>(*Pp, par, ackctl, publish_hb, sh_fl, sh_mk, hint, nx);
}

__device__ __forceinline__ void r3_body(const MpParams &P, int par, const uint32_t *__restrict__ ackctl, int publish_hb,
                                        const uint32_t g, bool active, const uint32_t d, const bool force_coop = false) {
    Lane L(P, d, g < P.G ? g : 0, par);
    bool loaded = false, job = false;
    if (active) {
        const RepView &v = L.v;
        bool has_pr = false;
#pragma unroll
        for (int s = 0; s < MAXR; s++)
            if ((uint32_t)s < P.R && (uint32_t)s != d && P.rep[s].pr_cnt[g] != 0 && P.rep[s].pr_dest[g] == d) has_pr = true;
        const uint32_t cnt = v.ob_cnt(par)[g];
        // (a lane mp_quorum_tally closed shows up here with an empty outbox)
        // a leader change in flight (PrepareReplies for me, or the long outbox of the
        // re-Accept round) is a cooperative job for the whole wave
        job = has_pr || cnt > 64 || (force_coop && (cnt != 0 || publish_hb));
        if (!job) {
            if (cnt) {
                L.load(); loaded = true;
                if (P.R <= 5) r3_accept_replies<5>(L, ackctl, cnt); else r3_accept_replies<MAXR>(L, ackctl, cnt);
            }
            if (publish_hb) { if (!loaded) { L.load(); loaded = true; } r3_publish_hb(L); }
            if (loaded) L.store();
        }
    }
    SMR_FOR_EACH_JOB(job, src) {
        const uint32_t gj = __shfl(g, src), dj = __shfl(d, src);
        Lane J(P, dj, gj, par);
        J.set_uniform();
        JBEGIN();
        JSTAMP(24);
        J.load();
        JSTAMP(25);
        r3_prepare_replies(J, ackctl ? ackctl[gj] : SMR_CTL_IDENTITY);
        JSTAMP(26);
        const uint32_t cj = J.v.ob_cnt(par)[gj];
        if (cj) { if (P.R <= 5) r3_accept_replies<5>(J, ackctl, cj); else r3_accept_replies<MAXR>(J, ackctl, cj); }
        JSTAMP(27);
        if (publish_hb) r3_publish_hb(J);
        J.store();
        JSTAMP(28);
        JEND(2);
        flush_job(J);
    }
    flush_counters(L, active && loaded);
}

#ifdef MP_R3_MINW
#define MP_R3_BOUNDS __launch_bounds__(256, MP_R3_MINW)
#else
#define MP_R3_BOUNDS __launch_bounds__(256)
#endif
// (Round 4 measured the bulk launch as ONE row of blocks that walk the replica rows themselves -- G / 256 blocks instead of R times
// as many to read the tiles' flag bytes and leave: no difference, 31.8 against 32.0 us for the tally + this launch
// (profiles/r5o_r3rest_one_row_negative.log).  The ~5 us this launch costs a steady tick are a launch's fixed cost, not its
// 5120 wavefronts coming and going.)
__global__ MP_R3_BOUNDS void mp_round_replies(const MpParams *__restrict__ Pp, int par,
                                                        const uint32_t *__restrict__ ackctl,
                                                        int publish_hb, int side) {
    const MpParams &P = *Pp;
    if (!side) SMR_RAISE_PRIO();
    if (!((P.live >> blockIdx.y) & 1u)) return;
    if (side == 0) {   // mp_quorum_tally left a flag per (replica, 64-group tile): nothing flagged, nothing to do
        const uint32_t ntile = (P.G + 63) / 64, tpb = blockDim.x >> 6, t0 = blockIdx.x * tpb;   // 64-group tiles of this block
        uint32_t any = 0;
        for (uint32_t k = 0; k < tpb; k++) any |= (t0 + k < ntile) ? P.r3_need[(size_t)blockIdx.y * ntile + t0 + k] : 0u;
        if (P.rot_on)                                            // (my row's replicas differ from group to group: any replica's flag)
            for (uint32_t y = 0; y < P.R; y++)
                for (uint32_t k = 0; k < tpb; k++) any |= (t0 + k < ntile) ? P.r3_need[(size_t)y * ntile + t0 + k] : 0u;
        if (!any) return;
    }
    uint32_t g;
    const bool active = pick_group(P, side, g);
    r3_body(P, par, ackctl, publish_hb, g, active, pick_replica(P, side, g));
}

// Inside a batch of ticks (smr_mp_run_ticks with the list on) the rest of tick t's R3 rides in the launch of tick t + 1's R1
// (round 4): mp_round_replies has nothing to do in the steady state and still cost the tick a launch (~5 us: a launch's
// fixed cost, profiles/r5o), and what it does do -- lanes the tally's closed form left -- needs nothing of tick t + 1 and
// nothing of it is needed before R2 of tick t + 1: it reads its own replica's state, the ack matrix and the PrepareReplies
// R2 of tick t wrote, and appends to its own outbox of the NEXT parity, which is the outbox R1 of tick t + 1 appends to
// right behind it in the same lane.  Not on a heartbeat tick (R4 reads the records R3 publishes) and not for a batch's last
// tick.  Same results: the handlers of a (group, replica) still run in the same order.  The one value that crosses replicas between
// the two halves is the group's `overflow` flag (a frozen group skips R1): the engine only defers in a quiet stretch (no HearTimeout
// for 2 x 16 + ttl ticks: smr_mp_run_ticks), and without Prepare traffic the rest of R3 has no path that freezes a group -- the
// window and outbox checks that set the flag sit in R1 / R2 and in the PrepareReply handlers (mp_device.h) -- so no replica's R1
// can run ahead of a flag another replica's deferred R3 would have set (ADVICE r4;
// tests/test_mp_gpu.py::test_window_overflow_while_the_r3_rest_rides_in_the_next_r1).
__global__ __launch_bounds__(256) void mp_rest_then_local(const MpParams *__restrict__ Pp, int par_prev,
                                                          const uint32_t *__restrict__ ackctl_prev, int par,
                                                          const uint8_t *__restrict__ timeout_rep, const uint8_t *__restrict__ timeout_src,
                                                          const uint8_t *__restrict__ req_target, const uint32_t *__restrict__ req_cnt,
                                                          const uint32_t *__restrict__ req_val, uint32_t S) {
    const MpParams &P = *Pp;
    SMR_RAISE_PRIO();
    if (!((P.live >> blockIdx.y) & 1u)) return;
    uint32_t g;
    const bool active = pick_group(P, 0, g);
    const uint32_t d = pick_replica(P, 0, g);
    {
        const uint32_t ntile = (P.G + 63) / 64, tpb = blockDim.x >> 6, t0 = blockIdx.x * tpb;
        uint32_t any = 0;
        for (uint32_t k = 0; k < tpb; k++) any |= (t0 + k < ntile) ? P.r3_need[(size_t)blockIdx.y * ntile + t0 + k] : 0u;
        if (P.rot_on)
            for (uint32_t y = 0; y < P.R; y++)
                for (uint32_t k = 0; k < tpb; k++) any |= (t0 + k < ntile) ? P.r3_need[(size_t)y * ntile + t0 + k] : 0u;
        if (any) {                                                // (block-uniform)
            r3_body(P, par_prev, ackctl_prev, 0, g, active, d);
            __syncthreads();                                      // (a cooperative job's stores -- a group frozen by lane 0 -- before R1 looks)
        }
    }
    r1_body(P, par, timeout_rep, timeout_src, req_target, req_cnt, req_val, S, g, active && !P.overflow[g < P.G ? g : 0], d);
}

// R4: all-to-all heartbeats (mod.rs:695, leadership.rs:217-265), then the ring
// trim (snapshot.rs:121-186, in-memory part) to min(my exec_bar, peers' exec_bar
// as carried by this round's heartbeats).
__device__ __forceinline__ void r4_body(const MpParams &P, int par, const uint32_t g, bool active, const uint32_t r) {
    Lane L(P, r, g < P.G ? g : 0, par);
    if (active) {
        L.load();
        const RepView &v = L.v;
        L.heard_heartbeat(r, v.hb_bal()[g], v.hb_commit()[g], v.hb_exec()[g], v.hb_snap()[g]);   // :254-261
        uint32_t bound = 0xFFFFFFFFu;
        // LS-1 order: the replica I follow first -- its heartbeat is the one that moves my commit bar, and
        // with it in the same position for every lane the long pass runs once per wavefront whichever
        // replicas lead the lanes' groups -- then the other peers by ascending id
        const uint32_t first = (L.leader != NO_REP && L.leader != r && L.leader < P.R) ? L.leader : NO_REP;
        uint32_t nxt = 0;                                       // ascending cursor over the other peers
        for (uint32_t k = 0; k + 1 < P.R; k++) {                // one call site: the sender is a per-lane value
            uint32_t s;
            if (k == 0 && first != NO_REP) s = first;
            else {
                while (nxt == r || nxt == first) nxt++;
                s = nxt++;
            }
            const RepView snd{P.rep[0], (size_t)s * P.rep_stride};
            const uint32_t he = snd.hb_exec()[g];
            if (he < bound) bound = he;
            if (!L.ovf) L.heard_heartbeat(s, snd.hb_bal()[g], snd.hb_commit()[g], he, snd.hb_snap()[g]);
        }
        if (L.ebar < bound) bound = L.ebar;
        if (bound > L.start) L.start = bound;
        L.store();
    }
    flush_counters(L, active);
}

#ifdef MP_R4_MINW
#define MP_R4_BOUNDS __launch_bounds__(256, MP_R4_MINW)
#else
#define MP_R4_BOUNDS __launch_bounds__(256)
#endif
__global__ MP_R4_BOUNDS void mp_round_heartbeat(const MpParams *__restrict__ Pp, int par, int side) {
    const MpParams &P = *Pp;
    if (!side) SMR_RAISE_PRIO();
    if (!((P.live >> blockIdx.y) & 1u)) return;
    uint32_t g;
    const bool active = pick_group(P, side, g);
    r4_body(P, par, g, active, pick_replica(P, side, g));
}

// One launch for the whole tick of the straggler list: a block per listed group (round robin), one
// wavefront per replica with the group on lane 0, the four rounds back to back with a block barrier in
// between -- what the bulk launches get from stream order.  Removes three launch boundaries from the
// stragglers' critical path (a leader change's handlers are serial latency, not bandwidth).
// (Round 2 tried packing the list tighter, twice.  A listed group's replicas on neighbouring lanes of ONE wavefront:
// 0.208 ms per tick against 0.157 (profiles/round2/r2w_strag_lanes.log) -- the replicas' handlers are divergent serial chains; on
// one wavefront they run one after the other, on five they overlap.  STRAG_K listed groups per block, on lanes 0..K-1 of
// every replica's wavefront: 0.19 ms at K = 2, 0.24 at K = 4 (profiles/round2/r2x_strag_k.log) -- cooperative jobs of the same
// wavefront queue behind each other.  What both runs showed: bench.py's list is a few dozen groups per tick, nowhere
// near the 256 blocks; this launch lasts as long as ONE group's leader change takes through its four rounds (~128 us
// of dependent memory round trips), and neither residency nor packing touches that.)
#ifndef STRAG_MINW
#define STRAG_MINW 1
#endif
#ifndef STRAG_K
#define STRAG_K 1      // listed groups per block: lanes 0 .. STRAG_K - 1 of every replica's wavefront
#endif
__global__ __launch_bounds__(512, STRAG_MINW) void mp_straggler_tick(const MpParams *__restrict__ Pp, int par, int lpar,
                                                          const uint8_t *__restrict__ timeout_rep,
                                                          const uint8_t *__restrict__ timeout_src,
                                                          const uint8_t *__restrict__ req_target,
                                                          const uint32_t *__restrict__ req_cnt,
                                                          const uint32_t *__restrict__ req_val, uint32_t S,
                                                          const uint32_t *__restrict__ ackctl, int do_heartbeat) {
    const MpParams &P = *Pp;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t n = P.slow_n[lpar];
    if (n > P.slow_cap) n = P.slow_cap;
    // entry (pass, lane, block) of the list = (pass * STRAG_K + lane) * gridDim.x + block: every block gets a group before
    // any block gets a second one
    for (uint32_t idx0 = blockIdx.x; idx0 < n; idx0 += gridDim.x * STRAG_K) {  // uniform per block
        const uint32_t idx = idx0 + lane * gridDim.x;
        const bool mine = w < P.R && lane < STRAG_K && idx < n;
        const uint32_t g = mine ? P.slow_list[idx] : P.G;
        const uint32_t r = w < P.R ? w : 0;
        if (timeout_rep || req_target)
            r1_body(P, par, timeout_rep, timeout_src, req_target, req_cnt, req_val, S, g, mine && !P.overflow[g], r);
        __syncthreads();
        r2_body<0>(P, par, g, mine && !P.overflow[g], r);                      // (a round may freeze the group)
        __syncthreads();
        r3_body(P, par, ackctl, do_heartbeat, g, mine && !P.overflow[g], r);
        __syncthreads();
        if (do_heartbeat) {
            r4_body(P, par, g, mine && !P.overflow[g], r);
            __syncthreads();
        }
    }
}

// The list's work for a BATCH of ticks in one launch (smr_mp_run_ticks with the straggler list on): a listed group never
// meets the bulk kernels, and groups never talk to each other, so its block simply runs the batch's ticks back to back
// while the bulk kernels go through them one launch at a time on the caller's stream.  What this buys over the per-tick
// launch above: a leader change costs its group ~130 us in the tick it happens and ~40 us in the next, far more than a
// bulk tick -- per tick, that chain IS the tick; per batch it only has to fit into the time the bulk takes for the whole
// batch, and the two meet once, at the batch's end.
// (STRAG_BATCH_K listed groups per block, on lanes 0 .. K-1 of every replica's wavefront: per tick that packing lost --
// the cooperative jobs of a wavefront queue behind each other and the tick waits for the longest queue -- but a batch has
// slack, and what counts here is how many listed groups the 256 blocks get through in the time the bulk needs.)
// 192 blocks x 6 groups: a quarter of the CUs stay free of the 5-wavefront, 220-VGPR side blocks, which is where the tally's
// 4-wavefront LDS blocks find their slots when the list is long (26 leader changes per tick: 0.157 -> 0.143 ms per tick against
// 256 x 4; no difference on the default workload, whose list is shorter than 192 groups -- profiles/round2/r2g6_side_blocks.log)
#ifndef STRAG_BATCH_K
#define STRAG_BATCH_K 6
#endif
// Round 5: a listed group's lane has its wavefront to itself (one listed group per block while the list is shorter than the grid),
// and in a quiet tick of its listing it ran the bulk kernels' per-lane fast paths -- three or four dependent rounds of loads per
// round, nothing to hide them: ~58 us per tick, 8 ticks per launch, which made the side launch as long as the bulk's whole batch
// (575 of 640 us, profiles/r8c_headline_timeline.txt) and the bulk kernels run beside it all the time.  With STRAG_COOP every
// round of a listed (group, replica) is a wave-cooperative job -- one lane per client batch (r1_coop_append), per message
// (r2_generic's append run), per ack-matrix row (r3_accept_replies' uniform mode) -- the paths a leader change takes anyway.
#ifndef STRAG_COOP
#define STRAG_COOP true
#endif
#ifndef STRAG_BATCH_BLOCKS
#define STRAG_BATCH_BLOCKS 192
#endif
#ifndef STRAG_BATCH_MINW
#define STRAG_BATCH_MINW STRAG_MINW
#endif
__global__ __launch_bounds__(512, STRAG_BATCH_MINW) void mp_straggler_batch(const MpParams *__restrict__ Pp, int lpar, const MpTickBatch B) {
    const MpParams &P = *Pp;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t n = P.slow_n[lpar];
    if (n > P.slow_cap) n = P.slow_cap;
#ifdef STRAG_PACK                                                 // (experiments: STRAG_PACK listed groups side by side in a block from the first one on)
    for (uint32_t idx0 = blockIdx.x * STRAG_PACK; idx0 < n; idx0 += gridDim.x * STRAG_PACK) {
        const uint32_t idx = idx0 + lane;
        const bool mine = w < P.R && lane < STRAG_PACK && idx < n;
#else
    for (uint32_t idx0 = blockIdx.x; idx0 < n; idx0 += gridDim.x * STRAG_BATCH_K) {  // uniform per block
        const uint32_t idx = idx0 + lane * gridDim.x;
        const bool mine = w < P.R && lane < STRAG_BATCH_K && idx < n;
#endif
        const uint32_t g = mine ? P.slow_list[idx] : P.G;
        const uint32_t r = w < P.R ? w : 0;
        for (uint32_t t = 0; t < B.n; t++) {
            const MpTickIn &in = B.t[t];
            const int par = B.par0 ^ (int)(t & 1u);
            if (in.timeout_rep || in.req_target)
                r1_body(P, par, in.timeout_rep, in.timeout_src, in.req_target, in.req_cnt, in.req_val, in.S, g, mine && !P.overflow[g], r, STRAG_COOP);
            __syncthreads();
            r2_body<0>(P, par, g, mine && !P.overflow[g], r, STRAG_COOP);
            __syncthreads();
            r3_body(P, par, in.ackctl, in.heartbeat, g, mine && !P.overflow[g], r, STRAG_COOP);
            __syncthreads();
            if (in.heartbeat) {
                r4_body(P, par, g, mine && !P.overflow[g], r);
                __syncthreads();
            }
        }
    }
}

// mp_mark_stragglers for a batch: a group that would be on the list in ANY tick of the batch is on it for the whole batch
// (its ttl evolves tick by tick exactly as the per-tick mark pass would have it, and carries over to the next batch)
__global__ __launch_bounds__(256) void mp_mark_batch(const MpParams *__restrict__ Pp, int lpar, const MpTickBatch B, uint32_t ttl) {
    const MpParams &P = *Pp;
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g == 0) P.slow_n[lpar ^ 1] = 0;
    if (g >= P.G) return;
    mark_role(P, g);
    uint32_t t = P.slow_ttl[g];
    bool any = false;
    for (uint32_t k = 0; k < B.n; k++) {
        const uint8_t *tr = B.t[k].timeout_rep;
        if (tr && tr[g] != NO_REP) t = ttl;
        if (t > 0) { any = true; t--; }
    }
    uint8_t s = 0;
    if (any) {
        const uint32_t idx = atomicAdd(&P.slow_n[lpar], 1u);
        if (idx < P.slow_cap) { P.slow_list[idx] = g; s = 1; }   // list full: the group stays with the bulk
    }
    P.slow_ttl[g] = (uint8_t)t;
    if (P.slow[g] != s) P.slow[g] = s;
}

// The whole tick -- and a batch of consecutive ticks -- in ONE launch.  Groups never talk to each other, so the round
// boundaries only have to order the replicas of the SAME group: a block owns 64 groups with all their replicas (a
// wavefront per replica for R1 / R2 / R3-rest / R4, four wavefronts for the quorum tally's row split) and a block
// barrier stands where the per-round launches have a kernel boundary.  What that buys on the co-located layout:
// no launch gaps (5 launches per tick -> 1 per batch); blocks drift apart, so one block's load latency is another's
// bandwidth; and a leader change -- serial latency of one wavefront, 100 us and more over its rounds -- delays only its
// own 64 groups while every other block keeps going, through the following ticks of the batch as well.  The per-round
// kernels above remain what the spread layout (an exchange between the rounds) and hosts with real I/O call.
// (the phases are separate functions on purpose: inlined into one body the register allocator keeps the union of
// their working sets -- 229 VGPRs, two wavefronts per SIMD -- instead of the largest one)
__device__ __forceinline__ void fused_r1(const MpParams *Pp, int par, const uint8_t *timeout_rep, const uint8_t *timeout_src,
                                      const uint8_t *req_target, const uint32_t *req_cnt, const uint32_t *req_val, uint32_t S,
                                      uint32_t g, uint32_t r) {
    r1_body(*Pp, par, timeout_rep, timeout_src, req_target, req_cnt, req_val, S, g, g < Pp->G && !Pp->overflow[g], r);
}
__device__ __forceinline__ void fused_r2(const MpParams *Pp, int par, uint32_t g, uint32_t r) {
    r2_body<0>(*Pp, par, g, g < Pp->G && !Pp->overflow[g], r);  // (a round may freeze the group)
}
template <int NR>
__device__ __forceinline__ void fused_tally(const MpParams *Pp, int par, const uint32_t *ackctl, int publish_hb, uint8_t *sh_fl,
                                         uint32_t *sh_mk) {
    quorum_tally_block<NR>(*Pp, par, ackctl, publish_hb, sh_fl, sh_mk, 0u);
}
__device__ __forceinline__ void fused_r3(const MpParams *Pp, int par, const uint32_t *ackctl, int publish_hb, uint32_t g, uint32_t r) {
    r3_body(*Pp, par, ackctl, publish_hb, g, g < Pp->G && !Pp->overflow[g], r);
}
__device__ __forceinline__ void fused_r4(const MpParams *Pp, int par, uint32_t g, uint32_t r) {
    r4_body(*Pp, par, g, g < Pp->G && !Pp->overflow[g], r);
}

#ifndef FUSED_MINW
#define FUSED_MINW 2
#endif
template <int NR, int NW>
__global__ __launch_bounds__(NW * 64, FUSED_MINW) void mp_ticks_fused(const MpParams *__restrict__ Pp, const MpTickBatch B) {
    __shared__ uint8_t sh_fl[64 * 64];
    __shared__ uint32_t sh_mk[64 * 64];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x * 64 + lane;
    const uint32_t R = Pp->R;
    const bool rw = w < R;                                       // this wavefront stands for replica w
    const uint32_t r = rw ? w : 0;
    const uint32_t ntile = (Pp->G + 63) / 64;
    // Round boundaries are `__syncthreads()` alone: HIP's barrier carries a workgroup-scope release / acquire, and the
    // wavefronts of a block share their CU's L1, so what one replica's wavefront stored is what the others load.  (A
    // `__threadfence()` here is an AGENT-scope fence -- on gfx950 an L2 write-back per barrier: measured 0.67 ms per tick.)
    for (uint32_t t = 0; t < B.n; t++) {
        const MpTickIn &in = B.t[t];
        const int par = B.par0 ^ (int)(t & 1u);
        if (rw && (in.timeout_rep || in.req_target))
            fused_r1(Pp, par, in.timeout_rep, in.timeout_src, in.req_target, in.req_cnt, in.req_val, in.S, g, r);
        __syncthreads();
        if (rw) fused_r2(Pp, par, g, r);
        __syncthreads();
        fused_tally<NR>(Pp, par, in.ackctl, in.heartbeat, sh_fl, sh_mk);
        __syncthreads();
        if (rw && Pp->r3_need[(size_t)r * ntile + blockIdx.x]) fused_r3(Pp, par, in.ackctl, in.heartbeat, g, r);
        __syncthreads();
        if (in.heartbeat) {
            if (rw) fused_r4(Pp, par, g, r);
            __syncthreads();
        }
    }
}

// ---- spread layout: what the rounds hand from one replica to another, as fixed-size images ------
// Replica r of a group may live on another rank.  This rank then holds r as an IMAGE: of everything r owns only the
// arrays the other replicas read or write between rounds exist here -- its outbox (R1 -> R2), the ack words / cells
// its followers fill (R2 -> R3), its PrepareReply batch (R2 -> R3) and its heartbeat record (R3 -> R4).  After a
// round, each rank packs those pieces of its LIVE replicas into contiguous buffers, one collective moves them
// (summerset_amd/spread.py: all_to_all_single over RCCL, static split sizes), and the receivers unpack them into
// the images.  Sizes are fixed per (kind, G, rows): a per-group part for the steady state and an overflow list of
// MP_IMG_OVF entries for what a leader change adds (irregular outboxes, ack cells of entries >= 64, PrepareReply rows).
struct ImgHdr { uint32_t ovf_n, ovf_cap, dropped, pad; };
struct ImgOvf { uint32_t g, j, a, b, c, pad; uint64_t d; };      // 32 B; meaning of a..d per kind, below
__device__ __forceinline__ void img_push(ImgHdr *h, ImgOvf *ov, uint32_t cap, const ImgOvf &e) {
    const uint32_t i = atomicAdd(&h->ovf_n, 1u);                 // (the header is zeroed by a memset in front of the pack kernel)
    if (i < cap) ov[i] = e; else atomicAdd(&h->dropped, 1u);
}
__host__ __device__ inline size_t img_align(size_t x) { return (x + 255) & ~(size_t)255; }
// byte offsets of the parts of an image
struct ImgLayout { size_t hdr, a, b, c, d, e, ovf, total; };
__host__ __device__ inline ImgLayout img_layout(uint32_t kind, uint32_t G, uint32_t rows, uint32_t ovf_cap) {
    ImgLayout L{};
    size_t o = img_align(sizeof(ImgHdr));
    L.hdr = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = img_align(o + bytes); return at; };
    if (kind == SMR_IMG_OUTBOX) { L.a = take((size_t)G * 4); L.b = take((size_t)G * 4); L.c = take((size_t)G * 8); L.d = take((size_t)rows * G * 4); }
    else if (kind == SMR_IMG_ACKS) { L.a = take((size_t)G * 8); }
    else if (kind == SMR_IMG_PREPARE_REPLIES) { L.a = take((size_t)G * 4); L.b = take((size_t)G * 4); L.c = take((size_t)G * 4); L.d = take((size_t)G * 4); L.e = take((size_t)G * 8); }
    else { L.a = take((size_t)G * 8); L.b = take((size_t)G * 4); L.c = take((size_t)G * 4); L.d = take((size_t)G * 4); }
    L.ovf = take((size_t)ovf_cap * sizeof(ImgOvf));
    L.total = o;
    return L;
}

// OUTBOX of replica `rep` (parity par): cnt, reg, rbal per group; the tokens of a regular outbox's first `rows`
// entries; everything else as overflow entries (a = ob_slot, b = ob_val, c = ob_aux, d = ob_bal; regular: b only)
__device__ __forceinline__ void img_pack_outbox(const MpParams &P, int par, uint32_t rep, uint32_t rows, uint8_t *img, const ImgLayout &L,
                                                uint32_t ocap, const uint32_t g, const uint32_t stride) {
    (void)stride;
    if (g == 0) ((ImgHdr *)img)->ovf_cap = ocap;
    if (g >= P.G) return;
    const RepView v{P.rep[0], (size_t)rep * P.rep_stride};
    ImgHdr *h = (ImgHdr *)img; ImgOvf *ov = (ImgOvf *)(img + L.ovf);
    uint32_t cnt = v.ob_cnt(par)[g];
    if (cnt > P.cap) cnt = P.cap;
    const uint32_t reg = v.ob_reg(par)[g];
    ((uint32_t *)(img + L.a))[g] = cnt; ((uint32_t *)(img + L.b))[g] = reg;
    ((uint64_t *)(img + L.c))[g] = reg ? v.ob_rbal(par)[g] : 0ull;
    uint32_t *tok = (uint32_t *)(img + L.d);
    for (uint32_t j = 0; j < cnt; j++) {
        const size_t o = tix(P.cap, j, g);
        if (reg && j < rows) tok[(size_t)j * P.G + g] = v.ob_val(par)[o];
        else if (reg) img_push(h, ov, ocap, ImgOvf{g, j, 0u, v.ob_val(par)[o], 0u, 0u, 0ull});
        else img_push(h, ov, ocap, ImgOvf{g, j, v.ob_slot(par)[o], v.ob_val(par)[o], v.ob_aux(par)[o], 0u, v.ob_bal(par)[o]});
    }
}
__device__ __forceinline__ void img_unpack_outbox(const MpParams &P, int par, uint32_t rep, uint32_t rows, const uint8_t *img,
                                                  const ImgLayout &L, const uint32_t g, const uint32_t stride) {
    const RepView v{P.rep[0], (size_t)rep * P.rep_stride};
    const ImgHdr *h = (const ImgHdr *)img; const ImgOvf *ov = (const ImgOvf *)(img + L.ovf);
    const uint32_t *cnts = (const uint32_t *)(img + L.a), *regs = (const uint32_t *)(img + L.b);
    if (g < P.G) {
        const uint32_t cnt = cnts[g], reg = regs[g];
        v.ob_cnt(par)[g] = cnt; v.ob_reg(par)[g] = reg;
        if (reg) v.ob_rbal(par)[g] = ((const uint64_t *)(img + L.c))[g];
        const uint32_t *tok = (const uint32_t *)(img + L.d);
        if (reg) for (uint32_t j = 0; j < cnt && j < rows; j++) v.ob_val(par)[tix(P.cap, j, g)] = tok[(size_t)j * P.G + g];
    }
    const uint32_t n = h->ovf_n < h->ovf_cap ? h->ovf_n : h->ovf_cap;
    for (uint32_t i = g; i < n; i += stride) {
        const ImgOvf e = ov[i];
        if (e.g >= P.G || e.j >= P.cap) continue;
        const size_t o = tix(P.cap, e.j, e.g);
        v.ob_val(par)[o] = e.b;
        if (!regs[e.g]) { v.ob_slot(par)[o] = e.a; v.ob_aux(par)[o] = e.c; v.ob_bal(par)[o] = e.d; }
    }
}
// ACKS follower `fol` gave sender `rep` this tick: the word for entries < 64; the byte cells of entries >= 64 (both
// values: the sender's cells are overwritten) as overflow entries (a = value)
__device__ __forceinline__ void img_pack_acks(const MpParams &P, int par, uint32_t rep, uint32_t fol, uint8_t *img, const ImgLayout &L,
                                              uint32_t ocap, const uint32_t g, const uint32_t stride) {
    (void)stride;
    if (g == 0) ((ImgHdr *)img)->ovf_cap = ocap;
    if (g >= P.G) return;
    const RepView v{P.rep[0], (size_t)rep * P.rep_stride};
    ImgHdr *h = (ImgHdr *)img; ImgOvf *ov = (ImgOvf *)(img + L.ovf);
    uint32_t cnt = v.ob_cnt(par)[g];
    if (cnt > P.cap) cnt = P.cap;
    ((uint64_t *)(img + L.a))[g] = cnt ? ack_bits_base(v.ack(), P.cap, P.G)[tix(MAXR, fol, g)] : 0ull;
    for (uint32_t j = 64; j < cnt; j++) img_push(h, ov, ocap, ImgOvf{g, j, (uint32_t)v.ack()[ack_ix(P.cap, j, fol, g)], 0u, 0u, 0u, 0ull});
}
__device__ __forceinline__ void img_unpack_acks(const MpParams &P, int par, uint32_t rep, uint32_t fol, const uint8_t *img,
                                                const ImgLayout &L, const uint32_t g, const uint32_t stride) {
    const RepView v{P.rep[0], (size_t)rep * P.rep_stride};
    const ImgHdr *h = (const ImgHdr *)img; const ImgOvf *ov = (const ImgOvf *)(img + L.ovf);
    if (g < P.G && v.ob_cnt(par)[g]) ack_bits_base(v.ack(), P.cap, P.G)[tix(MAXR, fol, g)] = ((const uint64_t *)(img + L.a))[g];
    const uint32_t n = h->ovf_n < h->ovf_cap ? h->ovf_n : h->ovf_cap;
    for (uint32_t i = g; i < n; i += stride) {
        const ImgOvf e = ov[i];
        if (e.g < P.G && e.j < P.cap) v.ack()[ack_ix(P.cap, e.j, fol, e.g)] = (uint8_t)e.a;
    }
}
// PREPARE_REPLIES of replica `rep`: header per group; the (voted_bal, voted_reqs) rows as overflow entries (a = vval, d = vbal)
__device__ __forceinline__ void img_pack_pr(const MpParams &P, uint32_t rep, uint8_t *img, const ImgLayout &L, uint32_t ocap, const uint32_t g,
                                            const uint32_t stride) {
    (void)stride;
    if (g == 0) ((ImgHdr *)img)->ovf_cap = ocap;
    if (g >= P.G) return;
    const RepView v{P.rep[0], (size_t)rep * P.rep_stride};
    ImgHdr *h = (ImgHdr *)img; ImgOvf *ov = (ImgOvf *)(img + L.ovf);
    uint32_t n = v.pr_cnt()[g];
    ((uint32_t *)(img + L.a))[g] = n; ((uint32_t *)(img + L.b))[g] = n ? v.pr_dest()[g] : 0u;
    ((uint32_t *)(img + L.c))[g] = n ? v.pr_trig()[g] : 0u; ((uint32_t *)(img + L.d))[g] = n ? v.pr_endp()[g] : 0u;
    ((uint64_t *)(img + L.e))[g] = n ? v.pr_bal()[g] : 0ull;
    if (n > P.pcap) n = P.pcap;
    for (uint32_t k = 0; k < n; k++) {
        const size_t o = tix(P.pcap, k, g);
        img_push(h, ov, ocap, ImgOvf{g, k, v.pr_vval()[o], 0u, 0u, 0u, v.pr_vbal()[o]});
    }
}
__device__ __forceinline__ void img_unpack_pr(const MpParams &P, uint32_t rep, const uint8_t *img, const ImgLayout &L, const uint32_t g,
                                              const uint32_t stride) {
    const RepView v{P.rep[0], (size_t)rep * P.rep_stride};
    const ImgHdr *h = (const ImgHdr *)img; const ImgOvf *ov = (const ImgOvf *)(img + L.ovf);
    if (g < P.G) {
        const uint32_t n = ((const uint32_t *)(img + L.a))[g];
        v.pr_cnt()[g] = n;
        if (n) {
            v.pr_dest()[g] = (uint8_t)((const uint32_t *)(img + L.b))[g]; v.pr_trig()[g] = ((const uint32_t *)(img + L.c))[g];
            v.pr_endp()[g] = ((const uint32_t *)(img + L.d))[g]; v.pr_bal()[g] = ((const uint64_t *)(img + L.e))[g];
        }
    }
    const uint32_t n = h->ovf_n < h->ovf_cap ? h->ovf_n : h->ovf_cap;
    for (uint32_t i = g; i < n; i += stride) {
        const ImgOvf e = ov[i];
        if (e.g >= P.G || e.j >= P.pcap) continue;
        const size_t o = tix(P.pcap, e.j, e.g);
        v.pr_vval()[o] = e.a; v.pr_vbal()[o] = e.d;
    }
}
// HEARTBEAT record of replica `rep` (leadership.rs:240-247): (bal_max_seen, commit_bar, exec_bar, snap_bar)
__device__ __forceinline__ void img_heartbeat(const MpParams &P, uint32_t rep, uint8_t *img, const ImgLayout &L, int unpack, const uint32_t g) {
    if (g >= P.G) return;
    const RepView v{P.rep[0], (size_t)rep * P.rep_stride};
    uint64_t *b = (uint64_t *)(img + L.a); uint32_t *c = (uint32_t *)(img + L.b), *x = (uint32_t *)(img + L.c), *sn = (uint32_t *)(img + L.d);
    if (unpack) { v.hb_bal()[g] = b[g]; v.hb_commit()[g] = c[g]; v.hb_exec()[g] = x[g]; v.hb_snap()[g] = sn[g]; }
    else { b[g] = v.hb_bal()[g]; c[g] = v.hb_commit()[g]; x[g] = v.hb_exec()[g]; sn[g] = v.hb_snap()[g]; }
}

// one image operation, as the plan kernels see it (smr_mp_image_plan_*): the cluster's device parameters, what to move
struct ImgOp {
    const MpParams *dp; uint8_t *img; const uint8_t *src;       // src: SMR_IMG_COPY only (the image this one duplicates)
    uint32_t kind, rep, other, G;
    uint64_t bytes;
    ImgLayout L;
};
constexpr uint32_t SMR_IMG_COPY = 4;

__device__ __forceinline__ void img_run(const ImgOp &op, int par, uint32_t rows, uint32_t ocap, bool unpack, uint32_t g, uint32_t stride) {
    const MpParams &P = *op.dp;
    if (!unpack) {
        if (g >= P.G) return;
        if (op.kind == SMR_IMG_OUTBOX) img_pack_outbox(P, par, op.rep, rows, op.img, op.L, ocap, g, stride);
        else if (op.kind == SMR_IMG_ACKS) img_pack_acks(P, par, op.rep, op.other, op.img, op.L, ocap, g, stride);
        else if (op.kind == SMR_IMG_PREPARE_REPLIES) img_pack_pr(P, op.rep, op.img, op.L, ocap, g, stride);
        else if (op.kind == SMR_IMG_HEARTBEAT) img_heartbeat(P, op.rep, op.img, op.L, 0, g);
    } else {
        if (op.kind == SMR_IMG_OUTBOX) img_unpack_outbox(P, par, op.rep, rows, op.img, op.L, g, stride);
        else if (op.kind == SMR_IMG_ACKS) img_unpack_acks(P, par, op.rep, op.other, op.img, op.L, g, stride);
        else if (op.kind == SMR_IMG_PREPARE_REPLIES) img_unpack_pr(P, op.rep, op.img, op.L, g, stride);
        else if (op.kind == SMR_IMG_HEARTBEAT && g < P.G) img_heartbeat(P, op.rep, op.img, op.L, 1, g);
    }
}
// one image per launch (smr_mp_image_pack / _unpack)
__global__ __launch_bounds__(256) void mp_img_one(const ImgOp op, int par, uint32_t rows, uint32_t ocap, int unpack) {
    img_run(op, par, rows, ocap, unpack != 0, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}
// a whole exchange's images in one launch: blockIdx.y = the operation
__global__ __launch_bounds__(256) void mp_img_many(const ImgOp *__restrict__ ops, int par, uint32_t rows, uint32_t ocap, int unpack) {
    const ImgOp op = ops[blockIdx.y];
    if (blockIdx.x * 256 >= ((op.G + 255) / 256) * 256) return;
    img_run(op, par, rows, ocap, unpack != 0, blockIdx.x * 256 + threadIdx.x, ((op.G + 255) / 256) * 256);
}
// pack side, in front of mp_img_many: the images' headers cleared (their overflow counters are bumped with atomics)
__global__ __launch_bounds__(256) void mp_img_zero_headers(const ImgOp *__restrict__ ops, uint32_t n, uint32_t ocap) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) *(ImgHdr *)ops[i].img = ImgHdr{0u, ocap, 0u, 0u};
}
// pack side, behind mp_img_many: the same piece for another rank is a copy of the packed one (16 bytes per lane and step)
__global__ __launch_bounds__(256) void mp_img_copy_many(const ImgOp *__restrict__ ops) {
    const ImgOp op = ops[blockIdx.y];
    struct alignas(16) B16 { uint64_t a, b; };
    const uint64_t n16 = op.bytes / 16;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256)
        ((B16 *)op.img)[i] = ((const B16 *)op.src)[i];
}

// ------------------------------------------------------------------ host ---
struct ProfEv { hipEvent_t a, b; int which; };


// ---- AcceptReply records <-> ack matrix (smr_mp_deliver_acks / smr_mp_collect_acks) ---------------
// The ack matrix is the batched form of "the AcceptReply messages that reached replica `rep` this tick"
// (PeerMsg::AcceptReply { slot, ballot } from `peer`, multipaxos/messages.rs:370-443).  A host that owns real
// I/O, or the multi-GPU exchange, holds them as records; these two kernels convert, one way and back.
// my outbox entry that is the Accept (slot, ballot) of group g, or NO_ENTRY
constexpr uint32_t NO_ENTRY = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t find_accept_entry(const MpParams &P, const RepView &v, int par, uint32_t g,
                                                      uint32_t slot, uint64_t ballot) {
    uint32_t cnt = v.ob_cnt(par)[g];
    if (cnt > P.cap) cnt = P.cap;
    const uint32_t reg = v.ob_reg(par)[g];
    if (reg) {                                                   // Accepts of the consecutive slots (reg - 1) + j at ob_rbal
        const uint32_t j = slot - (reg - 1);
        return (slot >= reg - 1 && j < cnt && ballot == v.ob_rbal(par)[g]) ? j : NO_ENTRY;
    }
    for (uint32_t j = 0; j < cnt; j++) {
        const size_t o = tix(P.cap, j, g);
        const uint32_t e = v.ob_slot(par)[o];
        if ((e >> OB_KIND_SH) == OB_ACCEPT && (e & OB_SLOT_MASK) == slot && v.ob_bal(par)[o] == ballot) return j;
    }
    return NO_ENTRY;
}

// one lane per record; a record that answers no Accept of this tick's outbox (other ballot, unknown slot, bad group /
// peer, my own id) is counted in *dropped and ignored, as the reference ignores an outdated AcceptReply
__global__ __launch_bounds__(256) void mp_deliver_acks_kernel(const MpParams *__restrict__ Pp, int par, uint32_t rep,
                                                              const smr_mp_ack *__restrict__ recs, uint64_t n,
                                                              unsigned long long *__restrict__ dropped) {
    const MpParams &P = *Pp;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t drop = 0;
    if (i < n) {
        const smr_mp_ack a = recs[i];
        drop = 1;
        if (a.group < P.G && a.peer < P.R && a.peer != rep) {
            const RepView v{P.rep[0], (size_t)rep * P.rep_stride};
            const uint32_t j = find_accept_entry(P, v, par, a.group, a.slot, a.ballot);
            if (j != NO_ENTRY) {
                drop = 0;
                if (j < 64u) atomicOr((unsigned long long *)&ack_bits_base(v.ack(), P.cap, P.G)[tix(MAXR, a.peer, a.group)], 1ull << j);
                else
                v.ack()[ack_ix(P.cap, j, a.peer, a.group)] = 1;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) drop += __shfl_xor(drop, off);
    if (__lane_id() == 0 && drop && dropped) atomicAdd(dropped, (unsigned long long)drop);
}

// the same for records in one segment per connection (smr_wire_ingest_mp_conn: connection c's cnt[c][0] 12-byte records
// { slot, ballot } from record conn_off[c] / 13 on; the group and the peer are the connection's): a wavefront per connection, a
// lane per record of it
__global__ __launch_bounds__(256) void mp_deliver_acks_conn_kernel(const MpParams *__restrict__ Pp, int par, uint32_t rep,
                                                                   const uint32_t *__restrict__ recs, uint64_t cap, const uint64_t *__restrict__ conn_off,
                                                                   const uint32_t *__restrict__ conn_group, const uint8_t *__restrict__ conn_peer,
                                                                   const uint32_t *__restrict__ cnt, uint32_t n_conn,
                                                                   unsigned long long *__restrict__ dropped) {
    const MpParams &P = *Pp;
    const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    uint32_t drop = 0;
    if (c < n_conn) {
        const uint32_t n_said = cnt[(size_t)c * 3], group = conn_group[c], peer = conn_peer[c];
        const uint64_t base = conn_off[c] / 13u;
        // a connection's segment ends where the next one's begins (conn_off has n_conn + 1 entries) and at `cap`: a count that
        // says more -- arrays of two different ingest calls -- must not make a neighbour's records this connection's;
        // what is cut off is counted as dropped
        const uint64_t room_seg = conn_off[c + 1] / 13u - base, room_cap = cap > base ? cap - base : 0ull;
        const uint64_t room = room_seg < room_cap ? room_seg : room_cap;
        const uint32_t n = (uint64_t)n_said < room ? n_said : (uint32_t)room;
        if (lane == 0) drop += n_said - n;
        const RepView v{P.rep[0], (size_t)rep * P.rep_stride};
        const bool mine = group < P.G && peer < P.R && peer != rep;
        for (uint32_t j = lane; j < n; j += 64u) {
            const uint32_t *r = recs + (base + j) * 3u;
            uint32_t d = 1;
            if (mine) {
                const uint32_t e = find_accept_entry(P, v, par, group, r[0], (uint64_t)r[1] | ((uint64_t)r[2] << 32));
                if (e != NO_ENTRY) {
                    d = 0;
                    if (e < 64u) atomicOr((unsigned long long *)&ack_bits_base(v.ack(), P.cap, P.G)[tix(MAXR, peer, group)], 1ull << e);
                    else
                    v.ack()[ack_ix(P.cap, e, peer, group)] = 1;
                }
            }
            drop += d;
        }
    }
    for (int off = 32; off > 0; off >>= 1) drop += __shfl_xor(drop, off);
    if (__lane_id() == 0 && drop && dropped) atomicAdd(dropped, (unsigned long long)drop);
}

// one lane per group: every set cell of my ack matrix as a record (any order); *n_out counts them all, records past
// `cap` are not stored
__global__ __launch_bounds__(256) void mp_collect_acks_kernel(const MpParams *__restrict__ Pp, int par, uint32_t rep,
                                                              smr_mp_ack *__restrict__ out, uint64_t cap,
                                                              unsigned long long *__restrict__ n_out) {
    const MpParams &P = *Pp;
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= P.G) return;
    const RepView v{P.rep[0], (size_t)rep * P.rep_stride};
    uint32_t cnt = v.ob_cnt(par)[g];
    if (cnt > P.cap) cnt = P.cap;
    const uint32_t reg = v.ob_reg(par)[g];
    const uint64_t rbal = reg ? v.ob_rbal(par)[g] : 0ull;
    SMR_G const uint64_t *const ackw = (SMR_G const uint64_t *)v.ack();
    for (uint32_t j = 0; j < cnt; j++) {
        const size_t o = tix(P.cap, j, g);
        uint32_t slot; uint64_t bal;
        if (reg) { slot = reg - 1 + j; bal = rbal; }
        else {
            const uint32_t e = v.ob_slot(par)[o];
            if ((e >> OB_KIND_SH) != OB_ACCEPT) continue;
            slot = e & OB_SLOT_MASK; bal = v.ob_bal(par)[o];
        }
        uint64_t w = ackw[o];
        if (j < 64u) {
            w = 0;
            for (uint32_t q = 0; q < P.R; q++) w |= ((ack_bits_base(v.ack(), P.cap, P.G)[tix(MAXR, q, g)] >> j) & 1ull) << (8 * q);
        }
        for (uint32_t q = 0; q < P.R; q++) {
            if (q == rep || !((w >> (8 * q)) & 0xFFull)) continue;
            const unsigned long long idx = atomicAdd(n_out, 1ull);
            if (idx < cap) { smr_mp_ack a; a.group = g; a.slot = slot; a.ballot = bal; a.peer = q; a.reserved = 0; out[idx] = a; }
        }
    }
}
// ---- the rounds of SEVERAL clusters in one launch (round 5, layout L2): a rank holds a cluster per block of groups it takes part
// in, and a round on block b has nothing to do with the round on block b'.  Round 4 ran them side by side on streams of the spread
// object's own -- a fork event, a launch per block, a join event per block: ~12 host calls per segment where the launches are
// ~10 us of device time each, so a rank's tick was its host's.  blockIdx.z = the cluster; a cluster with fewer groups leaves its
// surplus blocks at once.  Only for clusters without a straggler list (the spread layout's).
struct MpMulti {
    const MpParams *p[MAXR];
    int par[MAXR];
    uint32_t hint[MAXR];
    MpTickIn in[MAXR];
};
__global__ __launch_bounds__(256) void mp_round_local_multi(const MpMulti M) {
    const MpParams &P = *M.p[blockIdx.z];
    if (blockIdx.x * blockDim.x >= P.G || !((P.live >> blockIdx.y) & 1u)) return;
    const MpTickIn &in = M.in[blockIdx.z];
    uint32_t g;
    const bool active = pick_group(P, 0, g);
    r1_body(P, M.par[blockIdx.z], in.timeout_rep, in.timeout_src, in.req_target, in.req_cnt, in.req_val, in.S, g, active, pick_replica(P, 0, g));
}
__global__ __launch_bounds__(256, MP_R2_MINW) void mp_round_deliver_multi(const MpMulti M) {
    const MpParams &P = *M.p[blockIdx.z];
    if (blockIdx.x * blockDim.x >= P.G || !((P.live >> blockIdx.y) & 1u)) return;
    uint32_t g;
    const bool active = pick_group(P, 0, g);
    r2_body<0>(P, M.par[blockIdx.z], g, active, pick_replica(P, 0, g));
}
template <int NR>
__global__ __launch_bounds__(256, TALLY_MINW) void mp_quorum_tally_multi(const MpMulti M, int publish_hb) {
    __shared__ uint8_t sh_fl[64 * 64];
    __shared__ uint32_t sh_mk[64 * 64];
    const MpParams &P = *M.p[blockIdx.z];
    if (blockIdx.x * 64u >= P.G) return;                          // (block-uniform: nobody is left at a barrier)
    quorum_tally_block<NR>(P, M.par[blockIdx.z], M.in[blockIdx.z].ackctl, publish_hb, sh_fl, sh_mk, M.hint[blockIdx.z]);
}
__global__ __launch_bounds__(256) void mp_round_replies_multi(const MpMulti M, int publish_hb) {
    const MpParams &P = *M.p[blockIdx.z];
    if (blockIdx.x * blockDim.x >= P.G || !((P.live >> blockIdx.y) & 1u)) return;
    {
        const uint32_t ntile = (P.G + 63) / 64, tpb = blockDim.x >> 6, t0 = blockIdx.x * tpb;
        uint32_t any = 0;
        for (uint32_t k = 0; k < tpb; k++) any |= (t0 + k < ntile) ? P.r3_need[(size_t)blockIdx.y * ntile + t0 + k] : 0u;
        if (!any) return;
    }
    uint32_t g;
    const bool active = pick_group(P, 0, g);
    r3_body(P, M.par[blockIdx.z], M.in[blockIdx.z].ackctl, publish_hb, g, active, pick_replica(P, 0, g));
}
__global__ __launch_bounds__(256) void mp_round_heartbeat_multi(const MpMulti M) {
    const MpParams &P = *M.p[blockIdx.z];
    if (blockIdx.x * blockDim.x >= P.G || !((P.live >> blockIdx.y) & 1u)) return;
    uint32_t g;
    const bool active = pick_group(P, 0, g);
    r4_body(P, M.par[blockIdx.z], g, active, pick_replica(P, 0, g));
}

}  // namespace smr

using namespace smr;

struct smr_mp_cluster {
    smr_mp_cfg cfg;
    MpParams hp;            // host copy (device pointers inside)
    MpParams *dp = nullptr; // device copy
    Arena arena;
    uint32_t pcap = 0;
    int par = 0;
    int lpar = 0;                    // straggler-list counter parity
    uint32_t ttl = 0;                // ticks on the side stream after a HearTimeout (0 = never)
    hipStream_t side = nullptr;      // straggler launches
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool forked = false, marked = false, side_on = false, side_fused = false;
    uint32_t side_live = 0;          // ticks the side stream stays on without a new HearTimeout array
    uint32_t quiet_ticks = 0;        // smr_mp_run_ticks: ticks in a row without a HearTimeout array (the side launch has nothing listed)
    bool always_defer = false;       // SMR_MP_ALWAYS_DEFER_REST in the environment at smr_mp_create (A/B runs): the R3 rest rides in the next R1 launch
                                     // beside a busy side stream too
    bool no_split_r2 = true;         // the split is OFF unless SMR_MP_SPLIT_R2 is in the environment at smr_mp_create: measured, it does not
                                     // pay (profiles/r8i: fast path alone 24.2 us + rest 4.9 us against 26.8 us for the one launch)
    bool split_r2 = false;           // smr_mp_run_ticks, the side stream busy: R2's bulk launch = fast path + rest (r2_body)
    MpNextLocal next_local = MpNextLocal{nullptr, nullptr, nullptr, nullptr, 0u};   // smr_mp_run_ticks: the NEXT tick's client batches, for the tally launch of this one (round 6)
    bool fold_r1 = false;            // SMR_MP_FOLD_R1 in the environment at smr_mp_create: the tally launch of tick t also runs the leaders' steady-state
                                     // appends of tick t + 1 (built, tested both ways, off by default: a wash -- profiles/s18, s19)
    bool defer_rest = false;         // smr_mp_run_ticks: the next smr_mp_round_replies launches the tally only ...
    bool rest_pending = false;       // ... and the rest of that R3 rides in the next R1 launch (mp_rest_then_local)
    int rest_par = 0;
    const uint32_t *rest_ackctl = nullptr;
    uint32_t lead_hint = 0;          // the replica most groups are led by (smr_mp_preset_leader): the tally's speculative loads
    bool profile = false;
    std::vector<ProfEv> evs;
    double prof_ms[5] = {0, 0, 0, 0, 0};
    uint64_t prof_n[5] = {0, 0, 0, 0, 0};
};

namespace smr {

#ifndef SMR_SLOW_CAP
#define SMR_SLOW_CAP 1024
#endif
constexpr uint32_t SLOW_CAP = SMR_SLOW_CAP;   // groups per tick the side stream takes

static bool is_pow2(uint32_t x) { return x && !(x & (x - 1)); }

template <typename T> static void carve(Arena &a, T *&p, size_t n, bool dry) {
    size_t off = a.reserve(n * sizeof(T));
    if (!dry) p = a.at<T>(off);
}

static void layout(smr_mp_cluster *c, bool dry) {
    Arena &a = c->arena;
    a.used = 0;
    const size_t G = c->cfg.n_groups, W = c->cfg.window, R = c->cfg.population, cap = c->cfg.outbox_cap,
                 pcap = c->pcap;
    const size_t Gp = (G + 63) / 64 * 64;               // wave-tiled arrays hold whole 64-group tiles
    MpParams &P = c->hp;
    carve(a, P.overflow, G, dry);
    carve(a, P.dbg, 64, dry);
    carve(a, P.r3_need, R * ((G + 63) / 64), dry);
    carve(a, P.r2_need, R * ((G + 63) / 64), dry); carve(a, P.r2_res, R * G, dry);
    carve(a, P.slow, G, dry); carve(a, P.slow_ttl, G, dry); carve(a, P.role_rot, G, dry);
    carve(a, P.r1_done, G, dry);
    carve(a, P.slow_list, (size_t)SLOW_CAP, dry); carve(a, P.slow_n, 2, dry);
    for (size_t r = 0; r < R; r++) {
        MpRep &v = P.rep[r];
        carve(a, v.leader, G, dry);
        carve(a, v.bal_prep_sent, G, dry); carve(a, v.bal_prepared, G, dry); carve(a, v.bal_max_seen, G, dry);
        carve(a, v.start_slot, G, dry); carve(a, v.log_len, G, dry); carve(a, v.accept_bar, G, dry);
        carve(a, v.commit_bar, G, dry); carve(a, v.exec_bar, G, dry); carve(a, v.snap_bar, G, dry);
        carve(a, v.null_lb, G, dry);
        carve(a, v.bal_lo, G, dry);
        carve(a, v.peer_exec_bar, R * G, dry);
        carve(a, v.s_bal, W * Gp, dry); carve(a, v.s_val, W * Gp, dry); carve(a, v.s_meta, W * Gp, dry);
        carve(a, v.s_vbal, W * Gp, dry); carve(a, v.s_vval, W * Gp, dry); carve(a, v.s_pmax, W * Gp, dry);
        carve(a, v.s_ltrig, W * Gp, dry); carve(a, v.s_lendp, W * Gp, dry);
        carve(a, v.s_rtrig, W * Gp, dry); carve(a, v.s_rendp, W * Gp, dry);
        for (int p = 0; p < 2; p++) {
            carve(a, v.ob_cnt[p], G, dry);
            carve(a, v.ob_slot[p], cap * Gp, dry); carve(a, v.ob_bal[p], cap * Gp, dry);
            carve(a, v.ob_val[p], cap * Gp, dry); carve(a, v.ob_aux[p], cap * Gp, dry);
            carve(a, v.ob_reg[p], G, dry); carve(a, v.ob_rbal[p], G, dry);
        }
        carve(a, v.ack, cap * Gp * 8 + (size_t)MAXR * Gp * 8, dry);   // + one word per (follower, group): ack_bits_base()
        carve(a, v.pr_cnt, G, dry); carve(a, v.pr_dest, G, dry);
        carve(a, v.pr_trig, G, dry); carve(a, v.pr_endp, G, dry); carve(a, v.pr_abar, G, dry);
        carve(a, v.pr_bal, G, dry);
        carve(a, v.pr_vbal, pcap * Gp, dry); carve(a, v.pr_vval, pcap * Gp, dry);
        carve(a, v.hb_bal, G, dry); carve(a, v.hb_commit, G, dry); carve(a, v.hb_exec, G, dry);
        carve(a, v.hb_snap, G, dry);
        carve(a, v.counters, SMR_CTR_WORDS, dry);
        carve(a, v.clist, (size_t)c->cfg.commit_list_cap, dry);
        carve(a, v.clist_n, 1, dry);
    }
    // identical carve sequence per replica => constant distance between the replicas' arrays
    if (!dry) P.rep_stride = R > 1 ? (size_t)((char *)P.rep[1].leader - (char *)P.rep[0].leader) : 0;
}

static bool stride_ok(const MpParams &P) {
    for (uint32_t r = 1; r < P.R; r++) {
        const size_t o = r * P.rep_stride;
        if ((char *)P.rep[r].s_meta != (char *)P.rep[0].s_meta + o || (char *)P.rep[r].ack != (char *)P.rep[0].ack + o ||
            (char *)P.rep[r].ob_cnt[1] != (char *)P.rep[0].ob_cnt[1] + o ||
            (char *)P.rep[r].clist_n != (char *)P.rep[0].clist_n + o)
            return false;
    }
    return true;
}

// bracket [begin, end) of launches on `st` with a HIP event pair; idx = which pair to close
static int prof_begin(smr_mp_cluster *c, int which, hipStream_t st, int &idx) {
    idx = -1;
    if (!c->profile) return SMR_OK;
    ProfEv e; e.which = which;
    SMR_HIP_TRY(hipEventCreate(&e.a));
    SMR_HIP_TRY(hipEventCreate(&e.b));
    SMR_HIP_TRY(hipEventRecord(e.a, st));
    c->evs.push_back(e);
    idx = (int)c->evs.size() - 1;
    return SMR_OK;
}
static int prof_end(smr_mp_cluster *c, int idx, hipStream_t st) {
    if (idx < 0) return SMR_OK;
    SMR_HIP_TRY(hipEventRecord(c->evs[(size_t)idx].b, st));
    return SMR_OK;
}

// Bulk round kernels: lane = group, blockIdx.y = replica.  MP_BLOCK lanes per block: nothing in them crosses a wavefront
// (no LDS, no block barrier), so a block is ONE wavefront -- it fits any free wavefront slot, also on the CUs the side
// stream's long-lived blocks share with it, where a 4-wavefront block often does not.
#ifndef MP_BLOCK
#define MP_BLOCK 64
#endif
static dim3 mp_grid(const smr_mp_cluster *c) { return dim3((c->cfg.n_groups + MP_BLOCK - 1) / MP_BLOCK, c->cfg.population); }
static dim3 side_grid(const smr_mp_cluster *c) { return dim3(SLOW_CAP / 4, c->cfg.population); }

// Straggler side stream.  The first round call of a tick builds the tick's list; a round then
// launches its kernel twice -- bulk groups on the caller's stream, listed groups on the side
// stream -- between a fork (side waits for the caller's stream) and a join (the caller's stream
// waits for the side).  smr_mp_tick forks once around all rounds, a lone round call around itself.
static int ensure_marked(smr_mp_cluster *c, const uint8_t *timeout_rep_dev, hipStream_t st) {
    if (!c->ttl || c->marked) return SMR_OK;
    c->marked = true;
    // no HearTimeout array for ttl + 1 ticks: every list entry has expired and the last mark pass
    // cleared the flags, so the tick runs without the side stream at all
    if (timeout_rep_dev) c->side_live = c->ttl + 1;
    c->side_on = c->side_live != 0;
    if (!c->side_on) return SMR_OK;
    c->side_live--;
    hipLaunchKernelGGL(mp_mark_stragglers, dim3((c->cfg.n_groups + 255) / 256), dim3(256), 0, st, c->dp, c->lpar,
                       timeout_rep_dev, c->ttl);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}
static int fork_side(smr_mp_cluster *c, hipStream_t st, bool &own) {
    own = false;
    if (!c->side_on || c->forked) return SMR_OK;
    SMR_HIP_TRY(hipEventRecord(c->ev_fork, st));
    SMR_HIP_TRY(hipStreamWaitEvent(c->side, c->ev_fork, 0));
    c->forked = own = true;
    return SMR_OK;
}
static int join_side(smr_mp_cluster *c, hipStream_t st, bool own) {
    if (!own) return SMR_OK;
    SMR_HIP_TRY(hipEventRecord(c->ev_join, c->side));
    SMR_HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0));
    c->forked = false;
    return SMR_OK;
}

}  // namespace smr

extern "C" {

int smr_mp_cluster_create(const smr_mp_cfg *cfg, smr_mp_cluster **out) {
    if (!cfg || !out) return fail(SMR_ERR_ARG, "mp: null argument");
    if (cfg->n_groups == 0) return fail(SMR_ERR_ARG, "mp: n_groups is zero");
    if (cfg->population < 3 || cfg->population > SMR_MAX_REPLICAS)
        return fail(SMR_ERR_ARG, "mp: population must be in 3..8");
    if (!is_pow2(cfg->window) || cfg->window < 8 || cfg->window > (1u << 20))
        return fail(SMR_ERR_ARG, "mp: window must be a power of two in 8..2^20");
    if (cfg->win_reserve >= cfg->window) return fail(SMR_ERR_ARG, "mp: win_reserve >= window");
    if (cfg->outbox_cap < 4) return fail(SMR_ERR_ARG, "mp: outbox_cap < 4");
    uint32_t quorum = cfg->population / 2 + 1;                    // multipaxos/mod.rs:774
    if (cfg->commit_extra > cfg->population - quorum)             // rspaxos/mod.rs:600-605
        return fail(SMR_ERR_ARG, "mp: commit_extra (fault_tolerance) exceeds population - majority");
    smr_mp_cluster *c = new smr_mp_cluster();
    c->cfg = *cfg;
    c->pcap = cfg->window;
    memset(&c->hp, 0, sizeof(c->hp));
    layout(c, true);
    c->arena.size = c->arena.used + 256;
    hipError_t e = hipMalloc((void **)&c->arena.base, c->arena.size);
    if (e != hipSuccess) {
        delete c;
        return fail(SMR_ERR_DEVICE, std::string("mp: hipMalloc of engine state: ") + hipGetErrorString(e));
    }
    layout(c, false);
    MpParams &P = c->hp;
    P.G = cfg->n_groups; P.W = cfg->window; P.Wmask = cfg->window - 1; P.cap = cfg->outbox_cap;
    P.pcap = c->pcap; P.win_reserve = cfg->win_reserve; P.clist_cap = cfg->commit_list_cap;
    P.R = cfg->population; P.quorum = quorum; P.thresh = quorum + cfg->commit_extra; P.rspaxos = 0;
    P.slow_cap = SLOW_CAP;
    P.live = (1u << cfg->population) - 1u;                    // co-located: every replica runs here
    if (!stride_ok(P)) {
        (void)hipFree(c->arena.base);
        delete c;
        return fail(SMR_ERR_DEVICE, "mp: replica arrays are not equally spaced in the arena");
    }
    e = hipMemset(c->arena.base, 0, c->arena.size);
    for (uint32_t r = 0; e == hipSuccess && r < P.R; r++) e = hipMemset(P.rep[r].leader, 0xFF, P.G);
    if (e == hipSuccess) e = hipMalloc((void **)&c->dp, sizeof(MpParams));
    if (e == hipSuccess) e = hipMemcpy(c->dp, &c->hp, sizeof(MpParams), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(c->arena.base);
        if (c->dp) (void)hipFree(c->dp);
        delete c;
        return fail(SMR_ERR_DEVICE, std::string("mp: init: ") + hipGetErrorString(e));
    }
    c->ttl = cfg->straggler_ticks == SMR_STRAGGLER_OFF ? 0u : cfg->straggler_ticks;
    c->no_split_r2 = getenv("SMR_MP_SPLIT_R2") == nullptr;
    c->always_defer = getenv("SMR_MP_ALWAYS_DEFER_REST") != nullptr;
    c->fold_r1 = getenv("SMR_MP_FOLD_R1") != nullptr;
    if (c->ttl) {
        e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
        if (e != hipSuccess) {
            smr_mp_cluster_destroy(c);
            return fail(SMR_ERR_DEVICE, std::string("mp: side stream: ") + hipGetErrorString(e));
        }
    }
    *out = c;
    return SMR_OK;
}

void smr_mp_cluster_destroy(smr_mp_cluster *c) {
    if (!c) return;
    (void)hipDeviceSynchronize();
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->side) (void)hipStreamDestroy(c->side);
    for (auto &e : c->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (c->dp) (void)hipFree(c->dp);
    if (c->arena.base) (void)hipFree(c->arena.base);
    delete c;
}

int smr_mp_preset_leader(smr_mp_cluster *c, uint8_t rep) {
    if (!c || rep >= c->cfg.population) return fail(SMR_ERR_ARG, "mp: bad replica id");
    const uint32_t G = c->cfg.n_groups;
    std::vector<uint64_t> b(G, (1ull << 8) | (uint64_t)(rep + 1));
    for (uint32_t r = 0; r < c->cfg.population; r++) {
        MpRep &v = c->hp.rep[r];
        SMR_HIP_TRY(hipMemset(v.leader, rep, G));
        SMR_HIP_TRY(hipMemcpy(v.bal_max_seen, b.data(), G * 8, hipMemcpyHostToDevice));
        if (r == rep) {
            SMR_HIP_TRY(hipMemcpy(v.bal_prep_sent, b.data(), G * 8, hipMemcpyHostToDevice));
            SMR_HIP_TRY(hipMemcpy(v.bal_prepared, b.data(), G * 8, hipMemcpyHostToDevice));
        }
    }
    c->lead_hint = rep;
    return SMR_OK;
}

int smr_mp_round_local(smr_mp_cluster *c, const uint8_t *timeout_rep_dev, const uint8_t *timeout_src_dev,
                       const uint8_t *req_target_dev, const uint32_t *req_cnt_dev, const uint32_t *req_val_dev,
                       uint32_t S, void *stream) {
    if (!c) return fail(SMR_ERR_ARG, "mp: null cluster");
    if (timeout_rep_dev && !timeout_src_dev) return fail(SMR_ERR_ARG, "mp: timeout_rep without timeout_src");
    if (req_target_dev && (!req_cnt_dev || !req_val_dev)) return fail(SMR_ERR_ARG, "mp: incomplete request arrays");
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_marked(c, timeout_rep_dev, st); if (rc) return rc;
    if (!timeout_rep_dev && !req_target_dev) {
        if (c->rest_pending) {                                   // no R1 launch to ride in: the previous tick's R3 rest on its own
            c->rest_pending = false;
            hipLaunchKernelGGL(mp_round_replies, mp_grid(c), dim3(MP_BLOCK), 0, st, c->dp, c->rest_par, c->rest_ackctl, 0, 0);
            SMR_HIP_TRY(hipGetLastError());
        }
        return SMR_OK;
    }
    bool own;
    if ((rc = fork_side(c, st, own))) return rc;
    if (c->side_on && !c->side_fused) {
        hipLaunchKernelGGL(mp_round_local, side_grid(c), dim3(256), 0, c->side, c->dp, c->par, timeout_rep_dev,
                           timeout_src_dev, req_target_dev, req_cnt_dev, req_val_dev, S, 1 + c->lpar);
        SMR_HIP_TRY(hipGetLastError());
    }
    int pi;
    if ((rc = prof_begin(c, 0, st, pi))) return rc;
    if (c->rest_pending) {                                       // the previous tick's R3 rest, then this tick's R1: one launch
        c->rest_pending = false;
        hipLaunchKernelGGL(mp_rest_then_local, mp_grid(c), dim3(MP_BLOCK), 0, st, c->dp, c->rest_par, c->rest_ackctl, c->par,
                           timeout_rep_dev, timeout_src_dev, req_target_dev, req_cnt_dev, req_val_dev, S);
    } else
        hipLaunchKernelGGL(mp_round_local, mp_grid(c), dim3(MP_BLOCK), 0, st, c->dp, c->par, timeout_rep_dev,
                           timeout_src_dev, req_target_dev, req_cnt_dev, req_val_dev, S, 0);
    SMR_HIP_TRY(hipGetLastError());
    if ((rc = prof_end(c, pi, st))) return rc;
    return join_side(c, st, own);
}

int smr_mp_round_deliver(smr_mp_cluster *c, void *stream) {
    if (!c) return fail(SMR_ERR_ARG, "mp: null cluster");
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_marked(c, nullptr, st); if (rc) return rc;
    bool own;
    if ((rc = fork_side(c, st, own))) return rc;
    if (c->side_on && !c->side_fused) {
        hipLaunchKernelGGL(mp_round_deliver_all, side_grid(c), dim3(256), 0, c->side, c->dp, c->par, 1 + c->lpar);
        SMR_HIP_TRY(hipGetLastError());
    }
    int pi;
    if ((rc = prof_begin(c, 1, st, pi))) return rc;
    if (c->split_r2) {                                           // (smr_mp_run_ticks beside a busy side stream: see r2_body)
        hipLaunchKernelGGL(mp_round_deliver, mp_grid(c), dim3(MP_BLOCK), 0, st, c->dp, c->par);
        SMR_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(mp_round_deliver_rest, mp_grid(c), dim3(MP_BLOCK), 0, st, c->dp, c->par);
    } else {
        hipLaunchKernelGGL(mp_round_deliver_all, mp_grid(c), dim3(MP_BLOCK), 0, st, c->dp, c->par, 0);
    }
    SMR_HIP_TRY(hipGetLastError());
    if ((rc = prof_end(c, pi, st))) return rc;
    return join_side(c, st, own);
}

int smr_mp_round_replies(smr_mp_cluster *c, const uint32_t *ackctl_dev, int publish_heartbeat, void *stream) {
    if (!c) return fail(SMR_ERR_ARG, "mp: null cluster");
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_marked(c, nullptr, st); if (rc) return rc;
    bool own;
    if ((rc = fork_side(c, st, own))) return rc;
    if (c->side_on && !c->side_fused) {
        hipLaunchKernelGGL(mp_round_replies, side_grid(c), dim3(256), 0, c->side, c->dp, c->par, ackctl_dev,
                           publish_heartbeat, 1 + c->lpar);
        SMR_HIP_TRY(hipGetLastError());
    }
    int pi;
    if ((rc = prof_begin(c, 2, st, pi))) return rc;
    int pt;
    if ((rc = prof_begin(c, 4, st, pt))) return rc;             // the quorum-tally kernel alone
    if (c->cfg.population <= 5)
        hipLaunchKernelGGL(mp_quorum_tally<5>, dim3((c->cfg.n_groups + 63) / 64), dim3(256), 0, st,
                           c->dp, c->par, ackctl_dev, publish_heartbeat, c->lead_hint, c->next_local);
    else
        hipLaunchKernelGGL(mp_quorum_tally<MAXR>, dim3((c->cfg.n_groups + 63) / 64), dim3(256), 0, st,
                           c->dp, c->par, ackctl_dev, publish_heartbeat, c->lead_hint, c->next_local);
    SMR_HIP_TRY(hipGetLastError());
    if ((rc = prof_end(c, pt, st))) return rc;
    if (c->defer_rest && !publish_heartbeat) {                   // (smr_mp_run_ticks: the rest rides in the next tick's R1 launch)
        c->rest_pending = true; c->rest_par = c->par; c->rest_ackctl = ackctl_dev;
    } else {
        hipLaunchKernelGGL(mp_round_replies, mp_grid(c), dim3(MP_BLOCK), 0, st, c->dp, c->par, ackctl_dev,
                           publish_heartbeat, 0);
        SMR_HIP_TRY(hipGetLastError());
    }
    if ((rc = prof_end(c, pi, st))) return rc;
    return join_side(c, st, own);
}

int smr_mp_round_heartbeat(smr_mp_cluster *c, void *stream) {
    if (!c) return fail(SMR_ERR_ARG, "mp: null cluster");
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_marked(c, nullptr, st); if (rc) return rc;
    bool own;
    if ((rc = fork_side(c, st, own))) return rc;
    if (c->side_on && !c->side_fused) {
        hipLaunchKernelGGL(mp_round_heartbeat, side_grid(c), dim3(256), 0, c->side, c->dp, c->par, 1 + c->lpar);
        SMR_HIP_TRY(hipGetLastError());
    }
    int pi;
    if ((rc = prof_begin(c, 3, st, pi))) return rc;
    hipLaunchKernelGGL(mp_round_heartbeat, mp_grid(c), dim3(MP_BLOCK), 0, st, c->dp, c->par, 0);
    SMR_HIP_TRY(hipGetLastError());
    if ((rc = prof_end(c, pi, st))) return rc;
    return join_side(c, st, own);
}

int smr_mp_end_tick(smr_mp_cluster *c) {
    if (!c) return fail(SMR_ERR_ARG, "mp: null cluster");
    if (c->forked) return fail(SMR_ERR_STATE, "mp: end_tick inside an open side-stream fork");
    c->par ^= 1;
    if (c->marked && c->side_on) c->lpar ^= 1;
    c->marked = false;
    c->side_on = false;
    return SMR_OK;
}

int smr_mp_tick(smr_mp_cluster *c, const uint8_t *timeout_rep_dev, const uint8_t *timeout_src_dev,
                const uint8_t *req_target_dev, const uint32_t *req_cnt_dev, const uint32_t *req_val_dev,
                uint32_t S, const uint32_t *ackctl_dev, int do_heartbeat, void *stream) {
    if (!c) return fail(SMR_ERR_ARG, "mp: null cluster");
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_marked(c, timeout_rep_dev, st); if (rc) return rc;
    if (timeout_rep_dev && !timeout_src_dev) return fail(SMR_ERR_ARG, "mp: timeout_rep without timeout_src");
    if (req_target_dev && (!req_cnt_dev || !req_val_dev)) return fail(SMR_ERR_ARG, "mp: incomplete request arrays");
    bool own;
    if ((rc = fork_side(c, st, own))) return rc;                // one fork around the whole tick
    if (own) {                                                  // the list's whole tick: ONE launch on the side stream
        // a wavefront per replica: 5 x 64 lanes for the common populations (the idle wavefronts of a 512-lane block would
        // hold 227 VGPRs each on the CU the bulk kernels share with it)
        hipLaunchKernelGGL(mp_straggler_tick, dim3(256), dim3(c->cfg.population <= 5 ? 320 : 512), 0, c->side, c->dp, c->par, c->lpar,
                           timeout_rep_dev, timeout_src_dev, req_target_dev, req_cnt_dev, req_val_dev, S, ackctl_dev,
                           do_heartbeat);
        SMR_HIP_TRY(hipGetLastError());
        c->side_fused = true;                                   // the round calls below launch the bulk only
    }
    rc = smr_mp_round_local(c, timeout_rep_dev, timeout_src_dev, req_target_dev, req_cnt_dev, req_val_dev, S, stream);
    if (!rc) rc = smr_mp_round_deliver(c, stream);
    if (!rc) rc = smr_mp_round_replies(c, ackctl_dev, do_heartbeat, stream);
    if (!rc && do_heartbeat) rc = smr_mp_round_heartbeat(c, stream);
    c->side_fused = false;
    const int rj = join_side(c, st, own);
    if (rc) return rc;
    if (rj) return rj;
    return smr_mp_end_tick(c);
}

int smr_mp_run_ticks(smr_mp_cluster *c, const smr_mp_tick_in *ticks, uint32_t n, void *stream) {
    if (!c || (n && !ticks)) return fail(SMR_ERR_ARG, "mp: null argument");
    if (c->hp.live != (1u << c->cfg.population) - 1u) return fail(SMR_ERR_STATE, "mp: run_ticks needs every replica live (co-located layout)");
    if (c->forked || c->marked) return fail(SMR_ERR_STATE, "mp: run_ticks inside an open tick");
    hipStream_t st = (hipStream_t)stream;
    for (uint32_t i = 0; i < n; i++) {
        const smr_mp_tick_in &x = ticks[i];
        if (x.timeout_rep_dev && !x.timeout_src_dev) return fail(SMR_ERR_ARG, "mp: timeout_rep without timeout_src");
        if (x.req_target_dev && (!x.req_cnt_dev || !x.req_val_dev)) return fail(SMR_ERR_ARG, "mp: incomplete request arrays");
    }
    const uint32_t R = c->cfg.population;
    const dim3 grid((c->cfg.n_groups + 63) / 64);
    for (uint32_t i0 = 0; i0 < n; i0 += MP_FUSED_MAXT) {
        MpTickBatch b;
        memset(&b, 0, sizeof(b));
        b.n = n - i0 < MP_FUSED_MAXT ? n - i0 : MP_FUSED_MAXT;
        b.par0 = c->par;
        for (uint32_t k = 0; k < b.n; k++) {
            const smr_mp_tick_in &x = ticks[i0 + k];
            b.t[k] = MpTickIn{x.timeout_rep_dev, x.timeout_src_dev, x.req_target_dev, x.req_cnt_dev, x.req_val_dev, x.ackctl_dev,
                              x.S, x.do_heartbeat ? 1 : 0};
        }
        if (!c->ttl) {                                          // no straggler list: the whole batch in one fused launch
            if (R <= 5) hipLaunchKernelGGL((mp_ticks_fused<5, 5>), grid, dim3(5 * 64), 0, st, c->dp, b);
            else hipLaunchKernelGGL((mp_ticks_fused<MAXR, MAXR>), grid, dim3(MAXR * 64), 0, st, c->dp, b);
            SMR_HIP_TRY(hipGetLastError());
            c->par ^= (int)(b.n & 1u);
            continue;
        }
        bool any_to = false;
        for (uint32_t k = 0; k < b.n; k++) any_to = any_to || ticks[i0 + k].timeout_rep_dev != nullptr;
        c->quiet_ticks = any_to ? 0u : c->quiet_ticks + b.n;
        const bool quiet = c->quiet_ticks >= 2u * MP_FUSED_MAXT + c->ttl;   // every ttl has run out batches ago: the list is empty
        // Straggler list on: the list of the whole batch, its groups' ticks back to back in ONE launch on the side stream,
        // the bulk kernels tick by tick on the caller's stream; the streams meet at the end of the batch.
        hipLaunchKernelGGL(mp_mark_batch, dim3((c->cfg.n_groups + 255) / 256), dim3(256), 0, st, c->dp, c->lpar, b, c->ttl);
        SMR_HIP_TRY(hipGetLastError());
        SMR_HIP_TRY(hipEventRecord(c->ev_fork, st));
        SMR_HIP_TRY(hipStreamWaitEvent(c->side, c->ev_fork, 0));
        hipLaunchKernelGGL(mp_straggler_batch, dim3(STRAG_BATCH_BLOCKS), dim3(R <= 5 ? 320 : 512), 0, c->side, c->dp, c->lpar, b);
        SMR_HIP_TRY(hipGetLastError());
        c->marked = c->side_on = c->forked = c->side_fused = true;   // the round calls below: bulk only, no fork of their own
        c->split_r2 = !quiet && !c->no_split_r2;
        int rc = SMR_OK;
        for (uint32_t k = 0; k < b.n && !rc; k++) {
            const smr_mp_tick_in &x = ticks[i0 + k];
            rc = smr_mp_round_local(c, x.timeout_rep_dev, x.timeout_src_dev, x.req_target_dev, x.req_cnt_dev, x.req_val_dev, x.S, stream);
            if (!rc) rc = smr_mp_round_deliver(c, stream);
            // (not on a heartbeat tick: smr_mp_round_replies looks.  And only while nothing is on the straggler list: the fused
            // launch needs 171 VGPRs where R1 alone needs 158 -- two wavefronts per SIMD instead of three -- and beside the side
            // stream's blocks that costs R1 more than the launch saves: driver's command 0.0972 -> 0.0987 ms per tick, steady
            // 0.0568 -> 0.0542, profiles/r5p_rest_in_next_r1_ab.log)
            c->defer_rest = k + 1 < b.n && (quiet || c->always_defer);
            // (round 6, SMR_MP_FOLD_R1) ... and the leaders' steady-state appends of tick k + 1 ride in the tally launch of tick k: a group
            // the tally's closed form completes has nothing between the two but the heartbeat round, so not on a heartbeat tick.
            // MEASURED (profiles/s18, s19): the appends cost what they cost wherever they run -- tally 17.9 -> 22.3 us, the R1 launch
            // 12.4 -> 7.0 (steady), 24.2 -> 30.0 / 16.3 -> 10.4 beside the side launch: 0.0570 -> 0.0565 and 0.0892 -> 0.0895 ms per tick.
            // R1 is bound by the 39 MB it moves, not by its launch or its one wavefront per SIMD.
            if (k + 1 < b.n && !x.do_heartbeat && c->fold_r1 && ticks[i0 + k + 1].req_target_dev) {
                const smr_mp_tick_in &y = ticks[i0 + k + 1];
                c->next_local = MpNextLocal{y.timeout_rep_dev, y.req_target_dev, y.req_cnt_dev, y.req_val_dev, y.S};
            }
            if (!rc) rc = smr_mp_round_replies(c, x.ackctl_dev, x.do_heartbeat ? 1 : 0, stream);
            c->next_local = MpNextLocal{nullptr, nullptr, nullptr, nullptr, 0u};
            c->defer_rest = false;
            if (!rc && x.do_heartbeat) rc = smr_mp_round_heartbeat(c, stream);
            c->par ^= 1;
        }
        if (c->rest_pending) {                                   // (a round failed behind a deferral: nothing may stay pending)
            c->rest_pending = false;
            hipLaunchKernelGGL(mp_round_replies, mp_grid(c), dim3(MP_BLOCK), 0, st, c->dp, c->rest_par, c->rest_ackctl, 0, 0);
        }
        c->marked = c->side_on = c->forked = c->side_fused = false;
        c->split_r2 = false;
        hipError_t e1 = hipEventRecord(c->ev_join, c->side);
        hipError_t e2 = hipStreamWaitEvent(st, c->ev_join, 0);
        c->lpar ^= 1;
        if (rc) return rc;
        SMR_HIP_TRY(e1);
        SMR_HIP_TRY(e2);
    }
    return SMR_OK;
}

int smr_mp_set_role_rotation(smr_mp_cluster *c, int on) {
    if (!c) return fail(SMR_ERR_ARG, "mp: null cluster");
    const uint32_t all = (1u << c->cfg.population) - 1u;
    if (on && c->hp.live != all) return fail(SMR_ERR_STATE, "mp: role rotation needs every replica on this device");
    if (on && !c->ttl) return fail(SMR_ERR_STATE, "mp: role rotation rides on the straggler mark pass (straggler_ticks > 0)");
    SMR_HIP_TRY(hipDeviceSynchronize());
    c->hp.rot_on = on ? 1u : 0u;
    SMR_HIP_TRY(hipMemset(c->hp.role_rot, 0, c->cfg.n_groups));
    SMR_HIP_TRY(hipMemcpy(c->dp, &c->hp, sizeof(MpParams), hipMemcpyHostToDevice));
    return SMR_OK;
}

int smr_mp_set_live(smr_mp_cluster *c, uint32_t live_mask) {
    if (!c) return fail(SMR_ERR_ARG, "mp: null cluster");
    const uint32_t all = (1u << c->cfg.population) - 1u;
    if ((live_mask & ~all) != 0) return fail(SMR_ERR_ARG, "mp: live mask names a replica beyond the population");
    if (c->ttl && live_mask != all) return fail(SMR_ERR_STATE, "mp: the spread layout and the straggler side stream exclude each other");
    if (c->hp.rot_on && live_mask != all) return fail(SMR_ERR_STATE, "mp: the spread layout and role rotation exclude each other");
    SMR_HIP_TRY(hipDeviceSynchronize());
    c->hp.live = live_mask;
    SMR_HIP_TRY(hipMemcpy(c->dp, &c->hp, sizeof(MpParams), hipMemcpyHostToDevice));
    return SMR_OK;
}

int64_t smr_mp_image_bytes(smr_mp_cluster *c, int kind, uint32_t rows, uint32_t ovf_cap) {
    if (!c || kind < SMR_IMG_OUTBOX || kind > SMR_IMG_HEARTBEAT) return fail(SMR_ERR_ARG, "mp: bad image kind");
    return (int64_t)img_layout((uint32_t)kind, c->cfg.n_groups, rows, ovf_cap).total;
}

static int img_check(smr_mp_cluster *c, int kind, uint8_t rep, uint8_t other, const void *img, uint64_t img_bytes, uint32_t rows,
                     uint32_t ovf_cap, ImgLayout &L) {
    if (!c || !img || kind < SMR_IMG_OUTBOX || kind > SMR_IMG_HEARTBEAT) return fail(SMR_ERR_ARG, "mp: bad image argument");
    if (rep >= c->cfg.population || (kind == SMR_IMG_ACKS && (other >= c->cfg.population || other == rep)))
        return fail(SMR_ERR_ARG, "mp: bad replica for an image");
    if (((uintptr_t)img & 7u) != 0) return fail(SMR_ERR_ARG, "mp: image buffers must be 8-byte aligned");
    L = img_layout((uint32_t)kind, c->cfg.n_groups, rows, ovf_cap);
    if (img_bytes < L.total) return fail(SMR_ERR_ARG, "mp: image buffer smaller than smr_mp_image_bytes()");
    return SMR_OK;
}

static ImgOp img_op(smr_mp_cluster *c, int kind, uint8_t rep, uint8_t other, uint8_t *img, const ImgLayout &L) {
    ImgOp op{};
    op.dp = c->dp; op.img = img; op.src = nullptr; op.kind = (uint32_t)kind; op.rep = rep; op.other = other; op.G = c->cfg.n_groups;
    op.bytes = L.total; op.L = L;
    return op;
}

int smr_mp_image_pack(smr_mp_cluster *c, int kind, uint8_t rep, uint8_t other, uint8_t *img_dev, uint64_t img_bytes, uint32_t rows,
                      uint32_t ovf_cap, void *stream) {
    ImgLayout L;
    if (int rc = img_check(c, kind, rep, other, img_dev, img_bytes, rows, ovf_cap, L)) return rc;
    hipStream_t st = (hipStream_t)stream;
    SMR_HIP_TRY(hipMemsetAsync(img_dev, 0, sizeof(ImgHdr), st));
    hipLaunchKernelGGL(mp_img_one, dim3((c->cfg.n_groups + 255) / 256), dim3(256), 0, st, img_op(c, kind, rep, other, img_dev, L), c->par, rows,
                       ovf_cap, 0);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_mp_image_unpack(smr_mp_cluster *c, int kind, uint8_t rep, uint8_t other, const uint8_t *img_dev, uint64_t img_bytes,
                        uint32_t rows, uint32_t ovf_cap, void *stream) {
    ImgLayout L;
    if (int rc = img_check(c, kind, rep, other, img_dev, img_bytes, rows, ovf_cap, L)) return rc;
    hipLaunchKernelGGL(mp_img_one, dim3((c->cfg.n_groups + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       img_op(c, kind, rep, other, (uint8_t *)img_dev, L), c->par, rows, ovf_cap, 1);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

// ---- a whole exchange as a plan: the operations are static (which replica's piece goes into which buffer slice), so
// their descriptors are uploaded once and an exchange is three launches to pack (clear headers, pack, duplicate) and one
// to unpack, whatever the number of images
struct smr_mp_image_plan {
    std::vector<smr_mp_cluster *> cl;
    ImgOp *ops_dev = nullptr, *dup_dev = nullptr;
    uint32_t n_ops = 0, n_dup = 0, rows = 0, ovf_cap = 0, max_g = 0;
    uint64_t max_bytes = 0;
};

int smr_mp_image_plan_create(const smr_mp_image_op *ops, uint32_t n, uint32_t rows, uint32_t ovf_cap, smr_mp_image_plan **out) {
    if (!out || (n && !ops)) return fail(SMR_ERR_ARG, "mp: null argument");
    std::vector<ImgOp> main_ops, dup_ops;
    smr_mp_image_plan *p = new smr_mp_image_plan();
    p->rows = rows; p->ovf_cap = ovf_cap;
    for (uint32_t i = 0; i < n; i++) {
        const smr_mp_image_op &o = ops[i];
        ImgLayout L;
        if (int rc = img_check(o.cluster, o.kind, o.rep, o.other, o.img_dev, o.img_bytes, rows, ovf_cap, L)) { delete p; return rc; }
        ImgOp d = img_op(o.cluster, o.kind, o.rep, o.other, o.img_dev, L);
        if (o.copy_of_dev) {
            if (((uintptr_t)o.copy_of_dev & 15u) || ((uintptr_t)o.img_dev & 15u)) { delete p; return fail(SMR_ERR_ARG, "mp: duplicated images must be 16-byte aligned"); }
            d.kind = SMR_IMG_COPY; d.src = o.copy_of_dev;
            dup_ops.push_back(d);
            if (L.total > p->max_bytes) p->max_bytes = L.total;
        } else {
            main_ops.push_back(d);
            if (d.G > p->max_g) p->max_g = d.G;
        }
        p->cl.push_back(o.cluster);
    }
    p->n_ops = (uint32_t)main_ops.size(); p->n_dup = (uint32_t)dup_ops.size();
    hipError_t e = hipSuccess;
    if (p->n_ops) {
        e = hipMalloc((void **)&p->ops_dev, sizeof(ImgOp) * p->n_ops);
        if (e == hipSuccess) e = hipMemcpy(p->ops_dev, main_ops.data(), sizeof(ImgOp) * p->n_ops, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && p->n_dup) {
        e = hipMalloc((void **)&p->dup_dev, sizeof(ImgOp) * p->n_dup);
        if (e == hipSuccess) e = hipMemcpy(p->dup_dev, dup_ops.data(), sizeof(ImgOp) * p->n_dup, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) { smr_mp_image_plan_destroy(p); return fail(SMR_ERR_DEVICE, std::string("mp: image plan: ") + hipGetErrorString(e)); }
    *out = p;
    return SMR_OK;
}

void smr_mp_image_plan_destroy(smr_mp_image_plan *p) {
    if (!p) return;
    if (p->ops_dev) (void)hipFree(p->ops_dev);
    if (p->dup_dev) (void)hipFree(p->dup_dev);
    delete p;
}

int smr_mp_image_plan_run(smr_mp_image_plan *p, int unpack, void *stream) {
    if (!p) return fail(SMR_ERR_ARG, "mp: null plan");
    if (p->cl.empty()) return SMR_OK;
    const int par = p->cl[0]->par;
    for (smr_mp_cluster *c : p->cl)
        if (c->par != par) return fail(SMR_ERR_STATE, "mp: the clusters of an exchange must be in the same tick (outbox parity differs)");
    hipStream_t st = (hipStream_t)stream;
    if (unpack && p->n_dup) return fail(SMR_ERR_ARG, "mp: a receive plan has no duplicated images");
    if (p->n_ops) {
        if (!unpack) hipLaunchKernelGGL(mp_img_zero_headers, dim3((p->n_ops + 255) / 256), dim3(256), 0, st, p->ops_dev, p->n_ops, p->ovf_cap);
        hipLaunchKernelGGL(mp_img_many, dim3((p->max_g + 255) / 256, p->n_ops), dim3(256), 0, st, p->ops_dev, par, p->rows, p->ovf_cap, unpack);
    }
    if (!unpack && p->n_dup) {
        const uint64_t blocks = (p->max_bytes / 16 + 255) / 256;
        hipLaunchKernelGGL(mp_img_copy_many, dim3((unsigned)(blocks < 1024 ? (blocks ? blocks : 1) : 1024), p->n_dup), dim3(256), 0, st, p->dup_dev);
    }
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_mp_ack_matrix(smr_mp_cluster *c, uint8_t rep, uint8_t **ack_dev, uint64_t *n_bytes) {
    if (!c || rep >= c->cfg.population || !ack_dev) return fail(SMR_ERR_ARG, "mp: bad argument");
    *ack_dev = c->hp.rep[rep].ack;
    if (n_bytes) *n_bytes = (uint64_t)c->cfg.outbox_cap * ((c->cfg.n_groups + 63) / 64 * 64) * 8;
    return SMR_OK;
}

int smr_mp_deliver_acks(smr_mp_cluster *c, uint8_t rep, const smr_mp_ack *acks_dev, uint64_t n, uint64_t *dropped_dev,
                        void *stream) {
    if (!c || rep >= c->cfg.population || (n && !acks_dev)) return fail(SMR_ERR_ARG, "mp: bad argument");
    if (!n) return SMR_OK;
    if (n > 0x7FFFFFFFull * 256) return fail(SMR_ERR_ARG, "mp: too many ack records for one call");
    hipLaunchKernelGGL(mp_deliver_acks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c->dp, c->par,
                       (uint32_t)rep, acks_dev, n, (unsigned long long *)dropped_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_mp_deliver_acks_conn(smr_mp_cluster *c, uint8_t rep, const void *acks12_dev, uint64_t ack_cap, const uint64_t *conn_off_dev,
                             const uint32_t *conn_group_dev, const uint8_t *conn_peer_dev, const uint32_t *cnt_dev, uint32_t n_conn,
                             uint64_t *dropped_dev, void *stream) {
    if (!c || rep >= c->cfg.population || (n_conn && (!acks12_dev || !conn_off_dev || !conn_group_dev || !conn_peer_dev || !cnt_dev)))
        return fail(SMR_ERR_ARG, "mp: bad argument");
    if (!n_conn) return SMR_OK;
    hipLaunchKernelGGL(mp_deliver_acks_conn_kernel, dim3((n_conn + 3) / 4), dim3(256), 0, (hipStream_t)stream, c->dp, c->par, (uint32_t)rep,
                       (const uint32_t *)acks12_dev, ack_cap, conn_off_dev, conn_group_dev, conn_peer_dev, cnt_dev, n_conn, (unsigned long long *)dropped_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_mp_collect_acks(smr_mp_cluster *c, uint8_t rep, smr_mp_ack *out_dev, uint64_t cap, uint64_t *n_dev, void *stream) {
    if (!c || rep >= c->cfg.population || !n_dev || (cap && !out_dev)) return fail(SMR_ERR_ARG, "mp: bad argument");
    SMR_HIP_TRY(hipMemsetAsync(n_dev, 0, 8, (hipStream_t)stream));
    hipLaunchKernelGGL(mp_collect_acks_kernel, dim3((c->cfg.n_groups + 255) / 256), dim3(256), 0, (hipStream_t)stream, c->dp,
                       c->par, (uint32_t)rep, out_dev, cap, (unsigned long long *)n_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_mp_clear_acks(smr_mp_cluster *c, uint8_t rep, void *stream) {
    if (!c || rep >= c->cfg.population) return fail(SMR_ERR_ARG, "mp: bad argument");
    size_t nb = (size_t)c->cfg.outbox_cap * ((c->cfg.n_groups + 63) / 64 * 64) * 8;
    nb += (size_t)MAXR * ((c->cfg.n_groups + 63) / 64 * 64) * 8;
    SMR_HIP_TRY(hipMemsetAsync(c->hp.rep[rep].ack, 0, nb, (hipStream_t)stream));
    return SMR_OK;
}

int smr_mp_replica_log_view(smr_mp_cluster *c, uint8_t rep, smr_qread_log *out) {
    if (!c || !out || rep >= c->cfg.population) return fail(SMR_ERR_ARG, "mp: bad argument");
    const MpRep &v = c->hp.rep[rep];
    out->start_slot = v.start_slot; out->log_end = v.log_len;       // log_len is kept as start_slot + insts.len()
    out->status = v.s_meta; out->token = v.s_val; out->window = c->cfg.window; out->mp_layout = 1;
    out->run_lo = v.bal_lo; out->run_hi = v.commit_bar;             // inside the run a slot below commit_bar is Executed, unwritten
    out->run_leader = v.leader; out->run_rep = rep;                 // ... and one at or above it Accepting, unwritten, in a follower's run
    return SMR_OK;
}

int smr_mp_read_group_state(smr_mp_cluster *c, uint32_t group, uint8_t rep, smr_mp_group_state *out) {
    if (!c || !out || rep >= c->cfg.population || group >= c->cfg.n_groups)
        return fail(SMR_ERR_ARG, "mp: bad group / replica");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const MpRep &v = c->hp.rep[rep];
    memset(out, 0, sizeof(*out));
#define RD(dst, src) SMR_HIP_TRY(hipMemcpy(&(dst), (src) + group, sizeof(dst), hipMemcpyDeviceToHost))
    RD(out->leader, v.leader); RD(out->overflow, c->hp.overflow);
    RD(out->bal_prep_sent, v.bal_prep_sent); RD(out->bal_prepared, v.bal_prepared);
    RD(out->bal_max_seen, v.bal_max_seen);
    RD(out->start_slot, v.start_slot); RD(out->log_len, v.log_len); RD(out->accept_bar, v.accept_bar);
    RD(out->commit_bar, v.commit_bar); RD(out->exec_bar, v.exec_bar); RD(out->snap_bar, v.snap_bar);
#undef RD
    for (uint32_t p = 0; p < c->cfg.population; p++)
        if (p != rep)
            SMR_HIP_TRY(hipMemcpy(&out->peer_exec_bar[p], v.peer_exec_bar + (size_t)p * c->cfg.n_groups + group, 4,
                                  hipMemcpyDeviceToHost));
    return SMR_OK;
}

int smr_mp_dump_range(smr_mp_cluster *c, uint8_t rep, uint32_t g0, uint32_t n, const smr_mp_dump_bufs *hb) {
    if (!c || !hb || rep >= c->cfg.population) return fail(SMR_ERR_ARG, "mp: bad argument");
    const size_t GT = c->cfg.n_groups, W = c->cfg.window, R = c->cfg.population;
    if ((g0 & 63u) || (size_t)g0 + n > GT || (size_t)g0 + n < g0)
        return fail(SMR_ERR_ARG, "mp: dump range must start on a multiple of 64 and lie inside the cluster");
    if (n == 0) return SMR_OK;
    SMR_HIP_TRY(hipDeviceSynchronize());
    const MpRep &v = c->hp.rep[rep];
    const size_t G = n;                                                          // host buffers are [..][n]
#define D2H(dst, src, bytes) SMR_HIP_TRY(hipMemcpy((dst), (src), (bytes), hipMemcpyDeviceToHost))
    D2H(hb->leader, v.leader + g0, G);
    D2H(hb->bal_prep_sent, v.bal_prep_sent + g0, G * 8); D2H(hb->bal_prepared, v.bal_prepared + g0, G * 8);
    D2H(hb->bal_max_seen, v.bal_max_seen + g0, G * 8);
    D2H(hb->start_slot, v.start_slot + g0, G * 4); D2H(hb->log_len, v.log_len + g0, G * 4);
    D2H(hb->accept_bar, v.accept_bar + g0, G * 4); D2H(hb->commit_bar, v.commit_bar + g0, G * 4);
    D2H(hb->exec_bar, v.exec_bar + g0, G * 4); D2H(hb->snap_bar, v.snap_bar + g0, G * 4);
    for (size_t r = 0; r < R; r++) D2H(hb->peer_exec_bar + r * G, v.peer_exec_bar + r * GT + g0, G * 4);
    D2H(hb->overflow, c->hp.overflow + g0, G);
    for (size_t g = 0; g < G; g++) hb->peer_exec_bar[(size_t)rep * G + g] = 0;
    // the rows of groups [g0, g0 + n) are the wave tiles g0/64 .. : one contiguous piece of every tiled array
    const size_t Gp = (G + 63) / 64 * 64, t0 = (size_t)(g0 >> 6) * W * 64;
    std::vector<uint64_t> bal(W * Gp), vbal(W * Gp), pmax(W * Gp);
    std::vector<uint32_t> val(W * Gp), meta(W * Gp), vval(W * Gp), ltrig(W * Gp), lendp(W * Gp), rtrig(W * Gp), rendp(W * Gp);
    D2H(bal.data(), v.s_bal + t0, W * Gp * 8); D2H(vbal.data(), v.s_vbal + t0, W * Gp * 8); D2H(pmax.data(), v.s_pmax + t0, W * Gp * 8);
    D2H(val.data(), v.s_val + t0, W * Gp * 4); D2H(meta.data(), v.s_meta + t0, W * Gp * 4); D2H(vval.data(), v.s_vval + t0, W * Gp * 4);
    D2H(ltrig.data(), v.s_ltrig + t0, W * Gp * 4); D2H(lendp.data(), v.s_lendp + t0, W * Gp * 4);
    D2H(rtrig.data(), v.s_rtrig + t0, W * Gp * 4); D2H(rendp.data(), v.s_rendp + t0, W * Gp * 4);
    std::vector<uint32_t> bal_lo(G);
    D2H(bal_lo.data(), v.bal_lo + g0, G * 4);
#undef D2H
    // canonicalise: explicit Instance fields, zero outside [start_slot, log_len); the host
    // buffers are plain [W][n], the device arrays wave-tiled (tix)
    memset(hb->s_bal, 0, W * G * 8); memset(hb->s_status, 0, W * G); memset(hb->s_reqs, 0, W * G * 4);
    memset(hb->s_vbal, 0, W * G * 8); memset(hb->s_vreqs, 0, W * G * 4); memset(hb->s_flags, 0, W * G);
    memset(hb->s_acks, 0, W * G); memset(hb->s_packs, 0, W * G); memset(hb->s_pmax, 0, W * G * 8);
    memset(hb->s_ltrig, 0, W * G * 4); memset(hb->s_lendp, 0, W * G * 4); memset(hb->s_src, 0, W * G);
    memset(hb->s_rtrig, 0, W * G * 4); memset(hb->s_rendp, 0, W * G * 4);
    for (size_t g = 0; g < G; g++) {
        uint32_t lo = hb->start_slot[g], hi = hb->log_len[g];
        if (hi - lo > W) hi = lo + (uint32_t)W;
        for (uint32_t s = lo; s < hi; s++) {
            const size_t o = (size_t)(s & (W - 1)) * G + g;                     // host index
            const size_t t = tix((uint32_t)W, s & (uint32_t)(W - 1), (uint32_t)g);   // index inside the copied tiles
            uint32_t m = meta[t];
            if (s >= bal_lo[g]) bal[t] = hb->bal_max_seen[g];                    // inside the run the ballot is not stored
            if (s >= bal_lo[g] && hb->leader[g] != rep)                          // a follower's run: nor the meta word (mp_device.h)
                m = (s < hb->commit_bar[g] ? SMR_ST_EXECUTED : SMR_ST_ACCEPTING) | M_RBK | ((uint32_t)hb->leader[g] << M_SRC_SH) |
                    (VM_SAME << M_VMODE_SH) | (val[t] ? M_NONEMPTY : 0u);
            else if (s >= bal_lo[g] && s < hb->commit_bar[g]) m = (m & ~M_STATUS) | SMR_ST_EXECUTED;   // nor the statuses the bars imply
            hb->s_bal[o] = bal[t]; hb->s_status[o] = (uint8_t)(m & M_STATUS); hb->s_reqs[o] = val[t];
            uint32_t vm = (m >> M_VMODE_SH) & 3u;
            hb->s_vbal[o] = vm == VM_SAME ? bal[t] : (vm == VM_SIDE ? vbal[t] : 0);
            hb->s_vreqs[o] = vm == VM_SAME ? val[t] : (vm == VM_SIDE ? vval[t] : 0);
            bool lbk = m & M_LBK, rbk = m & M_RBK;
            hb->s_flags[o] = (uint8_t)((lbk ? 1 : 0) | (rbk ? 2 : 0) | ((m & M_EXT) ? 4 : 0));
            hb->s_acks[o] = lbk ? (uint8_t)((m >> M_ACKS_SH) & 0xFF) : 0;
            hb->s_packs[o] = lbk ? (uint8_t)((m >> M_PACKS_SH) & 0xFF) : 0;
            bool lx = lbk && (m & M_LBKX), rx = rbk && (m & M_RBKX);
            hb->s_pmax[o] = lx ? pmax[t] : 0; hb->s_ltrig[o] = lx ? ltrig[t] : 0; hb->s_lendp[o] = lx ? lendp[t] : 0;
            hb->s_src[o] = rbk ? (uint8_t)((m >> M_SRC_SH) & 7) : 0;
            hb->s_rtrig[o] = rx ? rtrig[t] : 0; hb->s_rendp[o] = rx ? rendp[t] : 0;
        }
    }
    return SMR_OK;
}

int smr_mp_dump(smr_mp_cluster *c, uint8_t rep, const smr_mp_dump_bufs *hb) {
    if (!c) return fail(SMR_ERR_ARG, "mp: bad argument");
    return smr_mp_dump_range(c, rep, 0, c->cfg.n_groups, hb);
}

int smr_mp_counters(smr_mp_cluster *c, uint8_t rep, uint64_t out[3]) {
    if (!c || !out || rep >= c->cfg.population) return fail(SMR_ERR_ARG, "mp: bad argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[4];
    SMR_HIP_TRY(ctr_read((const unsigned long long *)c->hp.rep[rep].counters, 4, h));
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2];
    return SMR_OK;
}

int smr_mp_straggler_stats(smr_mp_cluster *c, uint64_t out[2]) {
    if (!c || !out) return fail(SMR_ERR_ARG, "mp: bad argument");
    out[0] = c->ttl ? (uint64_t)SLOW_CAP : 0u; out[1] = 0;
    if (!c->ttl) return SMR_OK;
    SMR_HIP_TRY(hipDeviceSynchronize());
    uint32_t n[2];
    SMR_HIP_TRY(hipMemcpy(n, c->hp.slow_n, 8, hipMemcpyDeviceToHost));
    out[1] = n[c->lpar ^ 1];                                   // the parity the last mark pass counted into
    return SMR_OK;
}

int smr_mp_debug_generic_units(smr_mp_cluster *c, uint8_t rep, uint64_t *out) {
    if (!c || !out || rep >= c->cfg.population) return fail(SMR_ERR_ARG, "mp: bad argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[4];
    SMR_HIP_TRY(ctr_read((const unsigned long long *)c->hp.rep[rep].counters, 4, h));
    *out = h[3];
    return SMR_OK;
}

int smr_mp_debug_folded_batches(smr_mp_cluster *c, uint8_t rep, uint64_t *out) {
    if (!c || !out || rep >= c->cfg.population) return fail(SMR_ERR_ARG, "mp: bad argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[5];
    SMR_HIP_TRY(ctr_read((const unsigned long long *)c->hp.rep[rep].counters, 5, h));
    *out = h[4];
    return SMR_OK;
}

int smr_mp_debug_stamps(smr_mp_cluster *c, uint64_t *out64) {
    if (!c || !out64) return fail(SMR_ERR_ARG, "mp: bad argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    SMR_HIP_TRY(hipMemcpy(out64, c->hp.dbg, 64 * 8, hipMemcpyDeviceToHost));
    return SMR_OK;
}

int smr_mp_poll_commits(smr_mp_cluster *c, uint8_t rep, uint32_t *groups_host, uint32_t *slots_host, uint64_t cap,
                        uint64_t *n_out) {
    if (!c || rep >= c->cfg.population || !n_out) return fail(SMR_ERR_ARG, "mp: bad argument");
    if (c->cfg.commit_list_cap == 0) return fail(SMR_ERR_STATE, "mp: cluster was created without a commit list");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const MpRep &v = c->hp.rep[rep];
    unsigned int n = 0;
    SMR_HIP_TRY(hipMemcpy(&n, v.clist_n, 4, hipMemcpyDeviceToHost));
    *n_out = n;
    uint64_t take = n;
    if (take > c->cfg.commit_list_cap) take = c->cfg.commit_list_cap;
    if (take > cap) take = cap;
    if (take && groups_host && slots_host) {
        std::vector<unsigned long long> tmp(take);
        SMR_HIP_TRY(hipMemcpy(tmp.data(), v.clist, take * 8, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < take; i++) {
            groups_host[i] = (uint32_t)(tmp[i] >> 32);
            slots_host[i] = (uint32_t)tmp[i];
        }
    }
    SMR_HIP_TRY(hipMemset(v.clist_n, 0, 4));
    return SMR_OK;
}

int smr_mp_profile_enable(smr_mp_cluster *c, int on) {
    if (!c) return fail(SMR_ERR_ARG, "mp: null cluster");
    c->profile = on != 0;
    return SMR_OK;
}

int smr_mp_profile_read(smr_mp_cluster *c, int which, double *total_ms, uint64_t *launches) {
    if (!c || which < 0 || which > 4) return fail(SMR_ERR_ARG, "mp: bad argument");
    if (!c->evs.empty()) {
        SMR_HIP_TRY(hipDeviceSynchronize());
        for (auto &e : c->evs) {
            float ms = 0;
            SMR_HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
            c->prof_ms[e.which] += ms;
            c->prof_n[e.which]++;
            (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
        }
        c->evs.clear();
    }
    if (total_ms) *total_ms = c->prof_ms[which];
    if (launches) *launches = c->prof_n[which];
    return SMR_OK;
}

/* ---- a rank's part of an L2 (spread) job as ONE object: the tick's orchestration in the library --------------------------------
 * What summerset_amd/spread_mp.py did call by call -- a round on every block cluster of the rank, the exchange's pack in front
 * of the collective, its unpack behind it -- as one C call per SEGMENT of the tick; the collectives between the segments stay the
 * caller's (torch.distributed over RCCL: all_to_all_single on the plans' buffers).  Stand-in for server/transport.rs:208-275. */
struct smr_mp_spread {
    std::vector<smr_mp_cluster *> cl;
    smr_mp_image_plan *pack[3] = {nullptr, nullptr, nullptr}, *unpack[3] = {nullptr, nullptr, nullptr};   // outbox, replies, heartbeat
    // A round on block b has nothing to do with the round on block b': the blocks' launches go to side streams (block 0 stays on
    // the caller's), forked behind the unpack and joined in front of the pack.  A block's round kernel is latency-bound at a
    // quarter or an eighth of the groups (~20-40 us whatever its size): back to back they were the tick.
    std::vector<hipStream_t> side;               // [n_blocks - 1]
    std::vector<hipEvent_t> done;                // [n_blocks - 1]
    hipEvent_t fork = nullptr;
    bool concurrent = true;
    bool multi = true;                           // round 5: the blocks' rounds as ONE launch (blockIdx.z = block) instead of side by side on streams
    int next_segment = 0, tick_heartbeat = 0;    // the segment the open tick expects next, and the `heartbeat` its segment 0 came with
    // the exchange in the library (smr_mp_spread_bind_comm): the three exchanges' buffers and per-peer byte counts
    smr_comm *comm = nullptr;
    const void *sbuf[3] = {nullptr, nullptr, nullptr};
    void *rbuf[3] = {nullptr, nullptr, nullptr};
    std::vector<uint64_t> in_split[3], out_split[3];
};

// a round on every block: block 0 on the caller's stream, the others on their side streams between a fork and a join
static int spread_each_block(smr_mp_spread *s, hipStream_t st, const std::function<int(size_t, void *)> &round) {
    const size_t n = s->cl.size();
    int rc;
    if (!s->concurrent || n < 2) {
        for (size_t b = 0; b < n; b++)
            if ((rc = round(b, (void *)st)) != SMR_OK) return rc;
        return SMR_OK;
    }
    SMR_HIP_TRY(hipEventRecord(s->fork, st));
    // a round that fails leaves no side stream unjoined: whatever was forked is recorded and waited for before the error goes back
    size_t forked = 0;
    rc = SMR_OK;
    hipError_t he = hipSuccess;
    for (size_t b = 1; b < n && rc == SMR_OK && he == hipSuccess; b++) {
        if ((he = hipStreamWaitEvent(s->side[b - 1], s->fork, 0)) != hipSuccess) break;
        forked = b;
        rc = round(b, (void *)s->side[b - 1]);
    }
    if (rc == SMR_OK && he == hipSuccess) rc = round(0, (void *)st);
    for (size_t b = 1; b <= forked; b++) {
        const hipError_t e1 = hipEventRecord(s->done[b - 1], s->side[b - 1]);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(st, s->done[b - 1], 0) : e1;
        if (he == hipSuccess) he = e2;
    }
    if (rc != SMR_OK) return rc;
    SMR_HIP_TRY(he);
    return SMR_OK;
}

int smr_mp_spread_create(smr_mp_cluster *const *clusters, uint32_t n_blocks, smr_mp_image_plan *const *pack, smr_mp_image_plan *const *unpack,
                         smr_mp_spread **out) {
    if (!out || (n_blocks && !clusters) || !pack || !unpack) return fail(SMR_ERR_ARG, "mp spread: null argument");
    for (uint32_t b = 0; b < n_blocks; b++)
        if (!clusters[b]) return fail(SMR_ERR_ARG, "mp spread: null cluster");
    for (int k = 0; k < 3; k++)
        if (!pack[k] || !unpack[k]) return fail(SMR_ERR_ARG, "mp spread: the three exchanges (outbox, replies, heartbeat) need a pack and an unpack plan each");
    smr_mp_spread *s = new smr_mp_spread();
    s->cl.assign(clusters, clusters + n_blocks);
    for (int k = 0; k < 3; k++) { s->pack[k] = pack[k]; s->unpack[k] = unpack[k]; }
    bool ok = hipEventCreateWithFlags(&s->fork, hipEventDisableTiming) == hipSuccess;
    for (uint32_t b = 1; ok && b < n_blocks; b++) {
        hipStream_t st = nullptr; hipEvent_t ev = nullptr;
        ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
        if (st) s->side.push_back(st);
        if (ev) s->done.push_back(ev);
    }
    if (!ok) { smr_mp_spread_destroy(s); return fail(SMR_ERR_DEVICE, "mp spread: stream / event creation failed"); }
    *out = s;
    return SMR_OK;
}

void smr_mp_spread_destroy(smr_mp_spread *s) {
    if (!s) return;
    for (hipStream_t st : s->side) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : s->done) (void)hipEventDestroy(ev);
    if (s->fork) (void)hipEventDestroy(s->fork);
    delete s;
}

int smr_mp_spread_set_concurrent(smr_mp_spread *s, int on) {
    if (!s) return fail(SMR_ERR_ARG, "mp spread: null argument");
    s->concurrent = on != 0;
    s->multi = on >= 2 || on < 0;                                 // 2 (the default since round 5): the blocks' rounds in ONE launch; 1: side by side
    return SMR_OK;                                                // on streams of the object's own (round 4); 0: one after the other
}

// the blocks' round `which` (0 R1, 1 R2, 2 R3, 3 R4) as one launch (two for R3: the tally, then what it left); false: this
// object's clusters cannot go together (a straggler list, a profile pass, more blocks than a launch takes): the per-block path
static bool spread_multi_ok(const smr_mp_spread *s) {
    if (!s->multi || s->cl.size() < 2 || s->cl.size() > (size_t)MAXR) return false;
    for (const smr_mp_cluster *c : s->cl)
        if (c->ttl || c->profile || c->rest_pending || c->cfg.population != s->cl[0]->cfg.population) return false;
    return true;
}
static int spread_multi_round(smr_mp_spread *s, int which, const smr_mp_tick_in *in, int heartbeat, hipStream_t st) {
    MpMulti M;
    memset(&M, 0, sizeof(M));
    uint32_t gmax = 0;
    const size_t n = s->cl.size();
    for (size_t b = 0; b < n; b++) {
        const smr_mp_cluster *c = s->cl[b];
        M.p[b] = c->dp; M.par[b] = c->par; M.hint[b] = c->lead_hint;
        if (in) {
            const smr_mp_tick_in &x = in[b];
            // (the checks smr_mp_round_local makes per cluster: here the arrays go to the device as they are -- ADVICE r5)
            if (which == 0 && x.timeout_rep_dev && !x.timeout_src_dev) return fail(SMR_ERR_ARG, "mp: timeout_rep without timeout_src");
            if (which == 0 && x.req_target_dev && (!x.req_cnt_dev || !x.req_val_dev)) return fail(SMR_ERR_ARG, "mp: incomplete request arrays");
            M.in[b] = MpTickIn{x.timeout_rep_dev, x.timeout_src_dev, x.req_target_dev, x.req_cnt_dev, x.req_val_dev, x.ackctl_dev, x.S, heartbeat};
        }
        if (c->cfg.n_groups > gmax) gmax = c->cfg.n_groups;
    }
    const uint32_t R = s->cl[0]->cfg.population;
    const dim3 grid((gmax + MP_BLOCK - 1) / MP_BLOCK, R, (unsigned)n);
    switch (which) {
    case 0: hipLaunchKernelGGL(mp_round_local_multi, grid, dim3(MP_BLOCK), 0, st, M); break;
    case 1: hipLaunchKernelGGL(mp_round_deliver_multi, grid, dim3(MP_BLOCK), 0, st, M); break;
    case 2:
        if (R <= 5) hipLaunchKernelGGL(mp_quorum_tally_multi<5>, dim3((gmax + 63) / 64, 1, (unsigned)n), dim3(256), 0, st, M, heartbeat);
        else hipLaunchKernelGGL(mp_quorum_tally_multi<MAXR>, dim3((gmax + 63) / 64, 1, (unsigned)n), dim3(256), 0, st, M, heartbeat);
        SMR_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(mp_round_replies_multi, grid, dim3(MP_BLOCK), 0, st, M, heartbeat);
        break;
    default: hipLaunchKernelGGL(mp_round_heartbeat_multi, grid, dim3(MP_BLOCK), 0, st, M); break;
    }
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_mp_spread_segment(smr_mp_spread *s, int segment, const smr_mp_tick_in *in, int heartbeat, void *stream) {
    if (!s) return fail(SMR_ERR_ARG, "mp spread: null argument");
    if (segment < 0 || segment > 3) return fail(SMR_ERR_ARG, "mp spread: segment must be 0..3");
    if ((segment == 0 || segment == 2) && !s->cl.empty() && !in) return fail(SMR_ERR_ARG, "mp spread: segments 0 and 2 take the blocks' inputs");
    if (segment == 3 && !heartbeat) return fail(SMR_ERR_ARG, "mp spread: segment 3 exists on heartbeat ticks only");
    // the segments of a tick come in order and with the `heartbeat` the tick opened with: segment 2 without it ends the tick, a
    // later segment 3 with it would end it again and flip the outbox parity twice
    if (segment != s->next_segment)
        return fail(SMR_ERR_STATE, "mp spread: segment " + std::to_string(segment) + " out of order (the open tick expects segment " +
                                       std::to_string(s->next_segment) + ")");
    if (segment == 0) s->tick_heartbeat = heartbeat ? 1 : 0;
    else if ((heartbeat ? 1 : 0) != s->tick_heartbeat)
        return fail(SMR_ERR_STATE, "mp spread: `heartbeat` differs from the one the tick's segment 0 was called with");
    s->next_segment = (segment == 3 || (segment == 2 && !heartbeat)) ? 0 : segment + 1;
    int rc;
    const size_t n = s->cl.size();
    hipStream_t st = (hipStream_t)stream;
    switch (segment) {
    case 0:                                                  // R1 everywhere, then the outboxes into the send buffer
        if (spread_multi_ok(s)) { if ((rc = spread_multi_round(s, 0, in, heartbeat, st)) != SMR_OK) return rc; }
        else if ((rc = spread_each_block(s, st, [&](size_t b, void *sb) {
                 return smr_mp_round_local(s->cl[b], in[b].timeout_rep_dev, in[b].timeout_src_dev, in[b].req_target_dev, in[b].req_cnt_dev,
                                           in[b].req_val_dev, in[b].S, sb); })) != SMR_OK) return rc;
        return smr_mp_image_plan_run(s->pack[0], 0, stream);
    case 1:                                                  // the peers' outboxes arrive; R2; replies into the send buffer
        if ((rc = smr_mp_image_plan_run(s->unpack[0], 1, stream)) != SMR_OK) return rc;
        if (spread_multi_ok(s)) { if ((rc = spread_multi_round(s, 1, nullptr, heartbeat, st)) != SMR_OK) return rc; }
        else if ((rc = spread_each_block(s, st, [&](size_t b, void *sb) { return smr_mp_round_deliver(s->cl[b], sb); })) != SMR_OK) return rc;
        return smr_mp_image_plan_run(s->pack[1], 0, stream);
    case 2:                                                  // the replies arrive; R3; a heartbeat tick publishes and packs, else the tick ends
        if ((rc = smr_mp_image_plan_run(s->unpack[1], 1, stream)) != SMR_OK) return rc;
        if (spread_multi_ok(s)) { if ((rc = spread_multi_round(s, 2, in, heartbeat, st)) != SMR_OK) return rc; }
        else if ((rc = spread_each_block(s, st, [&](size_t b, void *sb) { return smr_mp_round_replies(s->cl[b], in[b].ackctl_dev, heartbeat, sb); })) != SMR_OK)
            return rc;
        if (heartbeat) return smr_mp_image_plan_run(s->pack[2], 0, stream);
        for (size_t b = 0; b < n; b++)
            if ((rc = smr_mp_end_tick(s->cl[b])) != SMR_OK) return rc;
        return SMR_OK;
    default:                                                 // the heartbeats arrive; R4; the tick ends
        if ((rc = smr_mp_image_plan_run(s->unpack[2], 1, stream)) != SMR_OK) return rc;
        if (spread_multi_ok(s)) { if ((rc = spread_multi_round(s, 3, nullptr, heartbeat, st)) != SMR_OK) return rc; }
        else if ((rc = spread_each_block(s, st, [&](size_t b, void *sb) { return smr_mp_round_heartbeat(s->cl[b], sb); })) != SMR_OK) return rc;
        for (size_t b = 0; b < n; b++)
            if ((rc = smr_mp_end_tick(s->cl[b])) != SMR_OK) return rc;
        return SMR_OK;
    }
}

// A segment or an exchange that fails leaves the tick OPEN (the next segment is still expected) and the blocks in the state of a
// partly run tick.  The host's way out (the object used to answer SMR_ERR_STATE for ever): abort the tick -- the
// object takes segment 0, bind_comm and destroy again -- and bring the clusters back to a state it trusts (smr_mp_load_state /
// new clusters) before the next tick.  Nothing is run or undone here.
int smr_mp_spread_abort_tick(smr_mp_spread *s) {
    if (!s) return fail(SMR_ERR_ARG, "mp spread: null argument");
    s->next_segment = 0;
    s->tick_heartbeat = 0;
    return SMR_OK;
}

int smr_mp_spread_bind_comm(smr_mp_spread *s, smr_comm *comm, const void *const send_dev[3], const uint64_t *send_bytes,
                            void *const recv_dev[3], const uint64_t *recv_bytes, uint32_t world) {
    if (!s) return fail(SMR_ERR_ARG, "mp spread: null argument");
    if (s->next_segment != 0) return fail(SMR_ERR_STATE, "mp spread: bind_comm inside an open tick");
    if (!comm) { s->comm = nullptr; return SMR_OK; }           // unbind: the collectives are the caller's again
    if (!send_dev || !send_bytes || !recv_dev || !recv_bytes) return fail(SMR_ERR_ARG, "mp spread: null argument");
    uint64_t info[5];
    int rc = smr_comm_info(comm, info);
    if (rc != SMR_OK) return rc;
    if (info[1] != world) return fail(SMR_ERR_ARG, "mp spread: the byte counts are per rank of the communicator's world");
    for (int k = 0; k < 3; k++) {
        s->sbuf[k] = send_dev[k]; s->rbuf[k] = recv_dev[k];
        s->in_split[k].assign(send_bytes + (size_t)k * world, send_bytes + (size_t)(k + 1) * world);
        s->out_split[k].assign(recv_bytes + (size_t)k * world, recv_bytes + (size_t)(k + 1) * world);
    }
    s->comm = comm;
    return SMR_OK;
}

int smr_mp_spread_tick(smr_mp_spread *s, const smr_mp_tick_in *in, int heartbeat, void *stream) {
    if (!s) return fail(SMR_ERR_ARG, "mp spread: null argument");
    if (!s->comm) return fail(SMR_ERR_STATE, "mp spread: no communicator bound (smr_mp_spread_bind_comm)");
    const int n_seg = heartbeat ? 4 : 3;
    if (s->next_segment != 0) return fail(SMR_ERR_STATE, "mp spread: a tick is open (its segments were called one by one, or it failed: smr_mp_spread_abort_tick)");
    for (int k = 0; k < n_seg; k++) {
        int rc = smr_mp_spread_segment(s, k, in, heartbeat, stream);
        if (rc == SMR_OK && k + 1 < n_seg)                       // exchange k sits between segments k and k + 1, on the same stream
            rc = smr_comm_exchange(s->comm, s->sbuf[k], s->in_split[k].data(), s->rbuf[k], s->out_split[k].data(), 0, stream);
        if (rc != SMR_OK) {                                      // this call opened the tick, so it closes it: the object stays usable,
            (void)smr_mp_spread_abort_tick(s);                   // the blocks' state is the host's to restore (see abort_tick)
            return rc;
        }
    }
    return SMR_OK;
}

}  // extern "C"

/*
 * oracle/rsp_oracle.c -- CPU restatement of the RSPaxos replica of Summerset over G independent
 * groups, one replica (id `me`) per group: leader append with one shard per peer, follower accept,
 * the accept tally with threshold majority + fault_tolerance, the commit-bar run gated on shard
 * availability, leader change (Prepare phase with shard merging, re-Accept), reconstruction reads,
 * heartbeat commit learning, the exec bar.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/mp_oracle.c header for the rules).
 *
 * Follows src/protocols/rspaxos/:
 *   handle_req_batch                    request.rs:10-151
 *   handle_msg_prepare                  messages.rs:12-84
 *   handle_msg_prepare_reply            messages.rs:87-340
 *   handle_msg_accept                   messages.rs:343-403
 *   handle_msg_accept_reply             messages.rs:406-464
 *   handle_msg_reconstruct              messages.rs:467-515
 *   handle_msg_reconstruct_reply        messages.rs:518-594
 *   handle_logged_{prepare_bal,accept_data,commit_slot}   durability.rs:10-186
 *   handle_cmd_result                   execution.rs:10-65
 *   check_leader, become_a_leader, bcast_heartbeats, heard_heartbeat   leadership.rs:11-340
 *   null_instance, first_null_slot, ballots                mod.rs:411-454
 * and, for what a codeword holds, src/utils/rscoding.rs: from_data / from_null (:165-251),
 * subset_copy (:255-293), absorb_other (:296-346), avail_shards / avail_data_shards,
 * compute_parity / reconstruct_data (:447-537).
 *
 * A request batch is an opaque token (0 = the empty batch, which is a real, non-null codeword;
 * NULL_CW = from_null).  A codeword is (token, mask of the shards present); RS(d, p) with
 * d = majority, p = population - majority.  Shard BYTES are the business of the RS kernels
 * (oracle/rs_oracle.c); here only which shards exist matters.  absorb_other of a codeword with another
 * token cannot happen in a correct run (the reference merges only at equal ballots resp. for committed
 * instances); it is counted (n_mixed) and keeps the absorbing side's token.
 * Rule 0: WAL appends and state-machine commands complete right after the handler
 * that submitted them returns, in submission order.  The state machine: digest = (digest ^ (slot << 32
 * | token)) * 0x100000001B3 per executed command.  NOT modelled: snapshots (start_slot = 0), leases,
 * timers (a HearTimeout is an input), msg_chunk_size (one Reconstruct message per step-up; the Heartbeat the
 * reference injects behind each chunk, leadership.rs:173-183, is a plain broadcast of fields this call returns and
 * is delivered by the schedule, summerset_amd/rsp_cluster.py).  ReconstructReply rows are taken in the order
 * given (the reference iterates a HashMap; the outcome does not depend on the order).
 * Harness guard shared with the engine: an instance that left the ring of the last W slots is ignored
 * like a slot below start_slot.
 *
 * PARITY STATUS: "parity unpinned" -- the reference has no unit tests or fixtures for these handlers
 * and cannot be built here; pinned by hand-derived traces (tests/test_oracle_rsp.py) and by the
 * protocol's safety properties on a closed loop (tests/test_oracle_rsp_cluster.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ST_NULL = 0, ST_PREPARING = 1, ST_ACCEPTING = 2, ST_COMMITTED = 3, ST_EXECUTED = 4 };
#define MAXR 8
#define NO_REP 0xFF
#define NULL_CW 0xFFFFFFFFu
#define DG_MUL 0x100000001B3ull

typedef struct { uint32_t val; uint8_t mask; } Cw;

typedef struct {
    uint64_t bal;
    uint8_t status;
    Cw cw;
    uint64_t voted_bal; Cw voted;
    uint8_t has_lbk, has_rbk, external;
    uint32_t l_trig, l_endp; uint8_t p_acks, a_acks; uint64_t p_max;    /* LeaderBookkeeping */
    uint8_t r_src; uint32_t r_trig, r_endp;                             /* ReplicaBookkeeping */
} Inst;

enum { WAL_PREPARE_BAL = 1, WAL_ACCEPT_DATA, WAL_COMMIT_SLOT };
typedef struct { uint8_t kind; uint32_t slot; } Act;

typedef struct {
    uint8_t id, population, majority, ft;
    uint32_t W;
    uint8_t leader;
    uint64_t bal_prep_sent, bal_prepared, bal_max_seen;
    Inst *insts; uint32_t len, cap;
    uint32_t commit_bar, exec_bar, snap_bar;
    uint32_t peer_exec_bar[MAXR];
    Act *wal; uint32_t n_wal, cap_wal;
    uint32_t *execq; uint32_t n_exec, cap_exec;
    uint64_t digest;
    uint32_t *xlog, *xlog_val; uint32_t n_xlog, cap_xlog;               /* executed since the last orc_rsp_take_executed */
    uint64_t n_commit, n_exec_total, n_mixed, n_redirect;
    /* messages produced by the handler in flight */
    uint64_t out_acc_reply;                                             /* AcceptReply ballot (0 = none) */
    uint32_t out_acc_reply_slot;
    uint32_t *out_acc_slot, *out_acc_val; uint32_t n_out_acc; uint64_t out_acc_ballot;   /* Accepts (to every peer) */
    uint32_t pr_n, pr_trig, pr_endp; uint64_t pr_ballot; uint8_t pr_dest;              /* PrepareReply batch */
    uint64_t *pr_vbal; uint32_t *pr_vval; uint8_t *pr_vmask;
} Rep;

typedef struct { uint32_t G; uint8_t R; Rep *reps; } Cl;

static int popc(uint32_t x) { return __builtin_popcount(x); }
static Cw cw_null(void) { Cw c = {NULL_CW, 0}; return c; }
static uint8_t data_mask(const Rep *r) { return (uint8_t)((1u << r->majority) - 1u); }
static uint8_t all_mask(const Rep *r) { return (uint8_t)((1u << r->population) - 1u); }
/* rscoding.rs:296-346 */
static void cw_absorb(Rep *r, Cw *self, Cw other) {
    if (self->val != NULL_CW && other.val == NULL_CW) { r->n_mixed++; return; }   /* data_len mismatch: Err */
    if (self->val == NULL_CW) self->val = other.val;
    else if (self->val != other.val) r->n_mixed++;
    self->mask |= other.mask;
}
/* rscoding.rs:255-293 (the source is never null where the handlers call it) */
static Cw cw_subset(Cw c, uint8_t subset) { Cw o = {c.val, (uint8_t)(c.mask & subset)}; return o; }

static Inst null_instance(void) {                                   /* mod.rs:411-431 */
    Inst in; memset(&in, 0, sizeof(in));
    in.cw = cw_null(); in.voted = cw_null(); in.r_src = NO_REP;
    return in;
}
static void push(Rep *r, Inst in) {
    if (r->len == r->cap) { r->cap = r->cap ? r->cap * 2 : 32; r->insts = (Inst *)realloc(r->insts, sizeof(Inst) * r->cap); }
    r->insts[r->len++] = in;
}
static int held(const Rep *r, uint32_t slot) { return slot < r->len && slot + r->W >= r->len; }
static uint32_t ring_lo(const Rep *r) { return r->len > r->W ? r->len - r->W : 0; }
static int is_leader(const Rep *r) { return r->leader == r->id; }
static void wal_submit(Rep *r, uint8_t kind, uint32_t slot) {
    if (r->n_wal == r->cap_wal) { r->cap_wal = r->cap_wal ? r->cap_wal * 2 : 64; r->wal = (Act *)realloc(r->wal, sizeof(Act) * r->cap_wal); }
    r->wal[r->n_wal].kind = kind; r->wal[r->n_wal].slot = slot; r->n_wal++;
}
static void exec_submit(Rep *r, uint32_t slot) {
    if (r->n_exec == r->cap_exec) { r->cap_exec = r->cap_exec ? r->cap_exec * 2 : 64; r->execq = (uint32_t *)realloc(r->execq, 4 * r->cap_exec); }
    r->execq[r->n_exec++] = slot;
}
static uint64_t make_greater_ballot(const Rep *r, uint64_t bal) { return (((bal >> 8) + 1) << 8) | (uint64_t)(r->id + 1); }

/* leadership.rs:11-42 */
static void check_leader(Rep *r, uint8_t peer, uint64_t ballot) {
    if (ballot > r->bal_max_seen) { r->bal_max_seen = ballot; r->leader = peer; }
}

/* Accepts to every peer for `slot`: one list per handler call, shard {peer} each */
static void emit_accept(Rep *r, uint32_t slot, uint64_t ballot, uint32_t val) {
    r->out_acc_slot[r->n_out_acc] = slot; r->out_acc_val[r->n_out_acc] = val; r->n_out_acc++;
    r->out_acc_ballot = ballot;
}

/* messages.rs:406-464 */
static void handle_msg_accept_reply(Rep *r, uint8_t peer, uint32_t slot, uint64_t ballot) {
    if (!held(r, slot)) return;
    if (ballot != r->bal_prepared) return;
    Inst *in = &r->insts[slot];
    if (!is_leader(r) || in->status != ST_ACCEPTING || ballot < in->bal) return;
    if (!in->has_lbk) return;                                        /* debug_assert in the reference */
    if ((in->a_acks >> peer) & 1) return;
    in->a_acks |= (uint8_t)(1u << peer);
    if (popc(in->a_acks) >= r->majority + r->ft) {                   /* :437-440 */
        in->status = ST_COMMITTED;
        r->n_commit++;
        wal_submit(r, WAL_COMMIT_SLOT, slot);
    }
}

/* messages.rs:87-340 */
static void handle_msg_prepare_reply(Rep *r, uint8_t peer, uint32_t slot, uint32_t trig, uint32_t endp, uint64_t ballot,
                                     int has_voted, uint64_t vbal, Cw vcw) {
    if (ballot != r->bal_prep_sent) return;                          /* :109 */
    if (!is_leader(r)) return;
    if (!held(r, trig) || !r->insts[trig].has_lbk) return;           /* :119-125 */
    const uint32_t my_endp = r->insts[trig].l_endp;
    while (r->len <= slot) {                                         /* :131-168 a slot I did not know of */
        Inst in = null_instance();
        in.external = 1; in.bal = r->bal_prep_sent; in.status = ST_PREPARING;
        in.has_lbk = 1; in.l_trig = trig; in.l_endp = my_endp; in.p_acks = 0; in.p_max = 0; in.a_acks = 0;
        push(r, in);
        wal_submit(r, WAL_PREPARE_BAL, r->len - 1);
    }
    if (!held(r, slot) || !held(r, trig)) return;
    {
        Inst *in = &r->insts[slot];
        if (in->status != ST_PREPARING || ballot < in->bal) return;  /* :173-176 */
        if (has_voted) {                                             /* :180-194 */
            if (vbal > in->p_max) { in->p_max = vbal; in->cw = vcw; }
            else if (vbal == in->p_max) cw_absorb(r, &in->cw, vcw);
        }
    }
    if (slot == endp) {                                              /* :200-338 */
        Inst *ti = &r->insts[trig];
        ti->p_acks |= (uint8_t)(1u << peer);
        const int cnt = popc(ti->p_acks);
        if (cnt >= r->majority) {
            r->bal_prepared = ballot;
            for (uint32_t s = trig; s < r->len; s++) {
                if (!held(r, s)) continue;
                Inst *in = &r->insts[s];
                if (in->status != ST_PREPARING) continue;
                if (popc(in->cw.mask) >= r->majority) {
                    if (popc(in->cw.mask & data_mask(r)) < r->majority) in->cw.mask |= data_mask(r);   /* reconstruct_data */
                } else if (cnt >= r->population - r->ft) {
                    in->cw.val = 0; in->cw.mask = data_mask(r);        /* from_data(ReqBatch::new()) */
                } else continue;
                if (popc(in->cw.mask) < r->population) in->cw.mask = all_mask(r);   /* compute_parity */
                in->status = ST_ACCEPTING;
                in->voted_bal = ballot; in->voted = cw_subset(in->cw, (uint8_t)(1u << r->id));
                wal_submit(r, WAL_ACCEPT_DATA, s);
                emit_accept(r, s, ballot, in->cw.val);
            }
        }
    }
}

/* durability.rs:10-82 */
static void handle_logged_prepare_bal(Rep *r, uint32_t slot) {
    if (!held(r, slot)) return;
    Inst *in = &r->insts[slot];
    const int has_voted = in->voted_bal > 0;
    if (is_leader(r)) {
        if (in->has_lbk && slot <= in->l_endp)
            handle_msg_prepare_reply(r, r->id, slot, in->l_trig, in->l_endp, in->bal, has_voted, in->voted_bal, in->voted);
    } else if (in->has_rbk) {                                        /* one entry of the PrepareReply batch to r_src */
        const uint32_t k = slot - in->r_trig;
        if (r->pr_n == 0) { r->pr_trig = in->r_trig; r->pr_endp = in->r_endp; r->pr_ballot = in->bal; r->pr_dest = in->r_src; }
        if (k < r->W) {
            r->pr_vbal[k] = has_voted ? in->voted_bal : 0; r->pr_vval[k] = has_voted ? in->voted.val : NULL_CW;
            r->pr_vmask[k] = has_voted ? in->voted.mask : 0;
            if (k + 1 > r->pr_n) r->pr_n = k + 1;
        }
    }
}

/* durability.rs:85-122 */
static void handle_logged_accept_data(Rep *r, uint32_t slot) {
    if (!held(r, slot)) return;
    Inst *in = &r->insts[slot];
    if (is_leader(r)) handle_msg_accept_reply(r, r->id, slot, in->bal);
    else if (in->has_rbk) { r->out_acc_reply = in->bal; r->out_acc_reply_slot = slot; }
}

/* the commit-bar run of durability.rs:140-181 and messages.rs:547-590 (`check_status`: the former submits only
 * for an instance still Committed) */
static void commit_bar_run(Rep *r, int check_status) {
    while (r->commit_bar < r->len && held(r, r->commit_bar)) {
        Inst *in = &r->insts[r->commit_bar];
        if (in->status < ST_COMMITTED) break;
        if (popc(in->cw.mask) < r->majority) break;                  /* cannot execute without the whole batch */
        if (popc(in->cw.mask & data_mask(r)) < r->majority) in->cw.mask |= data_mask(r);   /* reconstruct_data */
        if (in->cw.val == 0) in->status = ST_EXECUTED;               /* reqs.is_empty() */
        else if (!check_status || in->status == ST_COMMITTED) exec_submit(r, r->commit_bar);
        r->commit_bar++;
    }
}
/* durability.rs:125-186 */
static void handle_logged_commit_slot(Rep *r, uint32_t slot) {
    if (slot == r->commit_bar) commit_bar_run(r, 1);
}
/* execution.rs:10-65, one command per batch */
static void handle_cmd_result(Rep *r, uint32_t slot) {
    if (!held(r, slot)) return;
    Inst *in = &r->insts[slot];
    r->digest = (r->digest ^ (((uint64_t)slot << 32) | in->cw.val)) * DG_MUL;
    r->n_exec_total++;
    if (r->n_xlog == r->cap_xlog) {
        r->cap_xlog = r->cap_xlog ? r->cap_xlog * 2 : 32;
        r->xlog = (uint32_t *)realloc(r->xlog, 4 * r->cap_xlog); r->xlog_val = (uint32_t *)realloc(r->xlog_val, 4 * r->cap_xlog);
    }
    r->xlog[r->n_xlog] = slot; r->xlog_val[r->n_xlog] = in->cw.val; r->n_xlog++;
    in->status = ST_EXECUTED;
    if (slot == r->exec_bar)
        while (r->exec_bar < r->len && held(r, r->exec_bar)) {
            if (r->insts[r->exec_bar].status < ST_EXECUTED) break;
            r->exec_bar++;
        }
}

/* Rule 0 */
static void drain(Rep *r) {
    uint32_t wi = 0, ei = 0;
    for (;;) {
        if (wi < r->n_wal) {
            Act a = r->wal[wi++];
            if (a.kind == WAL_PREPARE_BAL) handle_logged_prepare_bal(r, a.slot);
            else if (a.kind == WAL_ACCEPT_DATA) handle_logged_accept_data(r, a.slot);
            else handle_logged_commit_slot(r, a.slot);
            continue;
        }
        if (ei < r->n_exec) { handle_cmd_result(r, r->execq[ei++]); continue; }
        break;
    }
    r->n_wal = 0; r->n_exec = 0;
}
static void begin(Rep *r) {
    r->out_acc_reply = 0; r->out_acc_reply_slot = 0; r->n_out_acc = 0; r->out_acc_ballot = 0; r->pr_n = 0;
    r->pr_trig = r->pr_endp = 0; r->pr_ballot = 0;
    for (uint32_t k = 0; k < r->W; k++) { r->pr_vbal[k] = 0; r->pr_vval[k] = NULL_CW; r->pr_vmask[k] = 0; }
}

/* leadership.rs:236-340; returns 1 if a Heartbeat reply goes back to `peer` */
static int heard_heartbeat(Rep *r, uint8_t peer, uint64_t ballot, uint32_t commit_bar, uint32_t exec_bar, uint32_t snap_bar) {
    int reply = 0;
    if (peer != r->id) {
        check_leader(r, peer, ballot);
        if (r->leader == peer) reply = 1;
    }
    if (ballot < r->bal_max_seen) return reply;
    if (exec_bar < r->exec_bar) return reply;
    if (commit_bar > r->commit_bar) {
        while (r->len < commit_bar) push(r, null_instance());
        for (uint32_t s = r->commit_bar; s < commit_bar; s++) {
            if (!held(r, s)) continue;                               /* harness: left the ring */
            Inst *in = &r->insts[s];
            if (in->bal < ballot || in->status < ST_ACCEPTING) break;
            else if (in->status >= ST_COMMITTED) continue;
            in->status = ST_COMMITTED;
            wal_submit(r, WAL_COMMIT_SLOT, s);
        }
    }
    if (peer != r->id) {
        if (exec_bar > r->peer_exec_bar[peer]) {
            r->peer_exec_bar[peer] = exec_bar;
            int passed = 1;
            for (int p = 0; p < r->population; p++) if (p != r->id && r->peer_exec_bar[p] >= exec_bar) passed++;
            if (passed == r->population) r->snap_bar = exec_bar;
        }
        if (snap_bar > r->snap_bar) r->snap_bar = snap_bar;
    }
    return reply;
}

/* ---- API: one call = one handler per group (+ its completions) ------------------------------- */
void *orc_rsp_new(uint32_t G, uint8_t R, uint8_t me, uint32_t W, uint8_t fault_tolerance) {
    Cl *cl = (Cl *)calloc(1, sizeof(Cl));
    cl->G = G; cl->R = R;
    cl->reps = (Rep *)calloc(G, sizeof(Rep));
    for (uint32_t g = 0; g < G; g++) {
        Rep *r = &cl->reps[g];
        r->id = me; r->population = R; r->majority = (uint8_t)(R / 2 + 1); r->ft = fault_tolerance; r->W = W;
        r->leader = NO_REP;
        r->out_acc_slot = (uint32_t *)calloc(W + 1, 4); r->out_acc_val = (uint32_t *)calloc(W + 1, 4);
        r->pr_vbal = (uint64_t *)calloc(W, 8); r->pr_vval = (uint32_t *)calloc(W, 4); r->pr_vmask = (uint8_t *)calloc(W, 1);
    }
    return cl;
}
void orc_rsp_free(void *h) {
    Cl *cl = (Cl *)h;
    for (uint32_t g = 0; g < cl->G; g++) {
        Rep *r = &cl->reps[g];
        free(r->insts); free(r->wal); free(r->execq); free(r->out_acc_slot); free(r->out_acc_val);
        free(r->pr_vbal); free(r->pr_vval); free(r->pr_vmask); free(r->xlog); free(r->xlog_val);
    }
    free(cl->reps); free(cl);
}
/* every replica starts believing in `leader`, prepared at make_unique_ballot(1) of that leader */
void orc_rsp_preset_leader(void *h, uint8_t leader) {
    Cl *cl = (Cl *)h;
    for (uint32_t g = 0; g < cl->G; g++) {
        Rep *r = &cl->reps[g];
        const uint64_t b = (1ull << 8) | (uint64_t)(leader + 1);
        r->leader = leader; r->bal_max_seen = b;
        if (leader == r->id) { r->bal_prep_sent = b; r->bal_prepared = b; }
    }
}

static void out_accepts(Rep *r, uint32_t g, uint32_t G, uint32_t *a_n, uint32_t *a_slot, uint32_t *a_val, uint64_t *a_ballot) {
    a_n[g] = r->n_out_acc; a_ballot[g] = r->out_acc_ballot;
    for (uint32_t k = 0; k < r->W; k++) {
        a_slot[(size_t)k * G + g] = k < r->n_out_acc ? r->out_acc_slot[k] : 0;
        a_val[(size_t)k * G + g] = k < r->n_out_acc ? r->out_acc_val[k] : 0;
    }
}

/* request.rs:10-151: val[g] (NULL_CW = nothing).  Accepts: a_n (0/1), a_slot[0], a_val[0], a_ballot */
void orc_rsp_req_batch(void *h, const uint32_t *val, uint32_t *a_n, uint32_t *a_slot, uint32_t *a_val, uint64_t *a_ballot) {
    Cl *cl = (Cl *)h;
    for (uint32_t g = 0; g < cl->G; g++) {
        Rep *r = &cl->reps[g];
        begin(r);
        if (val[g] != NULL_CW) {
            if (!is_leader(r) || r->bal_prepared == 0) r->n_redirect++;   /* request.rs:26-58: not a prepared leader */
            else {
                uint32_t slot = NULL_CW;                             /* mod.rs:434-442 */
                for (uint32_t s = r->exec_bar > ring_lo(r) ? r->exec_bar : ring_lo(r); s < r->len; s++)
                    if (r->insts[s].status == ST_NULL) { slot = s; break; }
                if (slot == NULL_CW) { push(r, null_instance()); slot = r->len - 1; }
                Inst *in = &r->insts[slot];
                in->cw.val = val[g]; in->cw.mask = all_mask(r);       /* from_data + compute_parity */
                in->has_lbk = 1; in->l_trig = 0; in->l_endp = 0; in->p_acks = 0; in->p_max = 0; in->a_acks = 0;
                in->external = 1;
                in->bal = r->bal_prepared; in->status = ST_ACCEPTING;
                in->voted_bal = in->bal; in->voted = cw_subset(in->cw, (uint8_t)(1u << r->id));
                wal_submit(r, WAL_ACCEPT_DATA, slot);
                emit_accept(r, slot, in->bal, in->cw.val);
            }
        }
        drain(r);
        out_accepts(r, g, cl->G, a_n, a_slot, a_val, a_ballot);
    }
}

/* messages.rs:343-403 + the AcceptData completion: reply r_ballot (0 = none) for r_slot */
void orc_rsp_accept(void *h, const uint8_t *flags, const uint8_t *peer, const uint32_t *slot, const uint64_t *ballot,
                    const uint32_t *val, const uint8_t *mask, uint64_t *r_ballot, uint32_t *r_slot) {
    Cl *cl = (Cl *)h;
    for (uint32_t g = 0; g < cl->G; g++) {
        Rep *r = &cl->reps[g];
        begin(r);
        if ((flags[g] & 1) && !(slot[g] < r->len && !held(r, slot[g])) && ballot[g] >= r->bal_max_seen) {
            check_leader(r, peer[g], ballot[g]);
            while (r->len <= slot[g]) push(r, null_instance());
            Inst *in = &r->insts[slot[g]];
            in->bal = ballot[g]; in->status = ST_ACCEPTING;
            in->cw.val = val[g]; in->cw.mask = mask[g];
            in->has_rbk = 1; in->r_src = peer[g]; in->r_trig = 0; in->r_endp = 0;
            in->voted_bal = ballot[g]; in->voted = in->cw;
            wal_submit(r, WAL_ACCEPT_DATA, slot[g]);
        }
        drain(r);
        r_ballot[g] = r->out_acc_reply; r_slot[g] = r->out_acc_reply_slot;
    }
}

static uint32_t ctl_order(uint32_t ctl, int i) { return (ctl >> (3 * i)) & 7u; }
#define CTL_IDENTITY 0x00FAC688u

/* AcceptReplies to my slot[g]: ballot / flags [R][G], peers in `order` order; committed[g] = it commits here */
void orc_rsp_accept_replies(void *h, const uint32_t *slot, const uint64_t *ballot, const uint8_t *flags, const uint32_t *order,
                            uint8_t *committed) {
    Cl *cl = (Cl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        Rep *r = &cl->reps[g];
        const uint32_t ctl = order ? order[g] : CTL_IDENTITY;
        const uint8_t before = held(r, slot[g]) ? r->insts[slot[g]].status : 0;
        for (int oi = 0; oi < cl->R; oi++) {
            const int p = (int)ctl_order(ctl, oi);
            if (p == r->id || p >= cl->R) continue;
            if (!(flags[(size_t)p * G + g] & 1)) continue;
            begin(r);
            handle_msg_accept_reply(r, (uint8_t)p, slot[g], ballot[(size_t)p * G + g]);
            drain(r);
        }
        committed[g] = (before == ST_ACCEPTING && held(r, slot[g]) && r->insts[slot[g]].status >= ST_COMMITTED) ? 1 : 0;
    }
}

/* leadership.rs:47-185 on HeartbeatEvent::HearTimeout: src[g] = the timeout source (NO_REP = no event here).
 * Out: the step-up Heartbeat (hb_flags, 4 fields), the Prepare (p_flags, trig, ballot), the Reconstruct list */
void orc_rsp_become_leader(void *h, const uint8_t *src, uint8_t *hb_flags, uint64_t *hb_ballot, uint32_t *hb_commit,
                           uint32_t *hb_exec, uint32_t *hb_snap, uint8_t *p_flags, uint32_t *p_trig, uint64_t *p_ballot,
                           uint32_t *rc_n, uint32_t *rc_slot) {
    Cl *cl = (Cl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        Rep *r = &cl->reps[g];
        begin(r);
        hb_flags[g] = 0; hb_ballot[g] = 0; hb_commit[g] = hb_exec[g] = hb_snap[g] = 0;
        p_flags[g] = 0; p_trig[g] = 0; p_ballot[g] = 0; rc_n[g] = 0;
        for (uint32_t k = 0; k < r->W; k++) rc_slot[(size_t)k * G + g] = 0;
        if (src[g] == NO_REP) continue;
        if (r->leader != NO_REP && r->leader != src[g]) continue;    /* :51-55 */
        r->leader = r->id;
        hb_flags[g] = 1; hb_ballot[g] = r->bal_max_seen; hb_commit[g] = r->commit_bar; hb_exec[g] = r->exec_bar;   /* :64 bcast_heartbeats */
        hb_snap[g] = r->snap_bar;
        (void)heard_heartbeat(r, r->id, r->bal_max_seen, r->commit_bar, r->exec_bar, r->snap_bar);
        for (int p = 0; p < r->population; p++) r->peer_exec_bar[p] = 0;
        r->bal_prepared = 0;                                         /* :72-74 */
        r->bal_prep_sent = make_greater_ballot(r, r->bal_max_seen);
        r->bal_max_seen = r->bal_prep_sent;
        const uint32_t lo = ring_lo(r);                              /* :77-92 (the ring stands in for the whole log) */
        uint32_t trig = r->len, endp = r->len;
        for (uint32_t s = lo; s < r->len; s++) if (r->insts[s].status < ST_COMMITTED) { trig = s; break; }
        for (uint32_t s = r->len; s > lo; s--) if (r->insts[s - 1].status < ST_COMMITTED) { endp = s - 1; break; }
        if (trig == r->len) push(r, null_instance());
        for (uint32_t s = r->exec_bar > ring_lo(r) ? r->exec_bar : ring_lo(r); s < r->len; s++) {   /* :100-149 */
            Inst *in = &r->insts[s];
            if (in->status == ST_EXECUTED) continue;
            in->external = 1;
            if (in->status < ST_COMMITTED) {
                in->bal = r->bal_prep_sent; in->status = ST_PREPARING;
                in->has_lbk = 1; in->l_trig = trig; in->l_endp = endp; in->p_acks = 0; in->p_max = 0; in->a_acks = 0;
                wal_submit(r, WAL_PREPARE_BAL, s);
            }
            if (in->status == ST_COMMITTED && popc(in->cw.mask) < r->majority && rc_n[g] < r->W)
                rc_slot[(size_t)(rc_n[g]++) * G + g] = s;
        }
        p_flags[g] = 1; p_trig[g] = trig; p_ballot[g] = r->bal_prep_sent;
        drain(r);
    }
}

/* messages.rs:12-84 + the PrepareBal completions: the PrepareReply batch (pr_n = 0: none) */
void orc_rsp_prepare(void *h, const uint8_t *flags, const uint8_t *peer, const uint32_t *trig, const uint64_t *ballot,
                     uint32_t *pr_n, uint32_t *pr_trig, uint32_t *pr_endp, uint64_t *pr_ballot, uint64_t *pr_vbal,
                     uint32_t *pr_vval, uint8_t *pr_vmask) {
    Cl *cl = (Cl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        Rep *r = &cl->reps[g];
        begin(r);
        if ((flags[g] & 1) && !(trig[g] < r->len && !held(r, trig[g])) && ballot[g] >= r->bal_max_seen) {
            check_leader(r, peer[g], ballot[g]);
            while (r->len <= trig[g]) push(r, null_instance());
            uint32_t last = 0;                                       /* :40-48 rposition(status > Null).unwrap_or(0) */
            for (uint32_t s = r->len; s > ring_lo(r); s--) if (r->insts[s - 1].status > ST_NULL) { last = s - 1; break; }
            const uint32_t endp = last > trig[g] ? last : trig[g];
            for (uint32_t s = trig[g]; s <= endp; s++) {
                if (!held(r, s)) continue;
                Inst *in = &r->insts[s];
                in->bal = ballot[g]; in->status = ST_PREPARING;
                in->has_rbk = 1; in->r_src = peer[g]; in->r_trig = trig[g]; in->r_endp = endp;
                wal_submit(r, WAL_PREPARE_BAL, s);
            }
        }
        drain(r);
        pr_n[g] = r->pr_n; pr_trig[g] = r->pr_trig; pr_endp[g] = r->pr_endp; pr_ballot[g] = r->pr_ballot;
        for (uint32_t k = 0; k < r->W; k++) {
            const size_t o = (size_t)k * G + g;
            pr_vbal[o] = r->pr_vbal[k]; pr_vval[o] = r->pr_vval[k]; pr_vmask[o] = r->pr_vmask[k];
        }
    }
}

/* one peer's PrepareReply batch, slot by slot (messages.rs:87-340); out: the Accepts it lets me send */
void orc_rsp_prepare_replies(void *h, const uint8_t *peer, const uint32_t *pr_n, const uint32_t *pr_trig, const uint32_t *pr_endp,
                             const uint64_t *pr_ballot, const uint64_t *pr_vbal, const uint32_t *pr_vval, const uint8_t *pr_vmask,
                             uint32_t *a_n, uint32_t *a_slot, uint32_t *a_val, uint64_t *a_ballot) {
    Cl *cl = (Cl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        Rep *r = &cl->reps[g];
        begin(r);
        for (uint32_t k = 0; k < pr_n[g]; k++) {
            const size_t o = (size_t)k * G + g;
            Cw v = {pr_vval[o], pr_vmask[o]};
            handle_msg_prepare_reply(r, peer[g], pr_trig[g] + k, pr_trig[g], pr_endp[g], pr_ballot[g], pr_vbal[o] > 0, pr_vbal[o], v);
            drain(r);
        }
        out_accepts(r, g, G, a_n, a_slot, a_val, a_ballot);
    }
}

/* messages.rs:467-515: rc_n[g] slots asked for; the reply rows (rr_n of them) */
void orc_rsp_reconstruct(void *h, const uint8_t *flags, const uint32_t *rc_n, const uint32_t *rc_slot, uint32_t *rr_n,
                         uint32_t *rr_slot, uint64_t *rr_bal, uint32_t *rr_val, uint8_t *rr_mask) {
    Cl *cl = (Cl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        Rep *r = &cl->reps[g];
        begin(r);
        rr_n[g] = 0;
        for (uint32_t k = 0; k < r->W; k++) { const size_t o = (size_t)k * G + g; rr_slot[o] = 0; rr_bal[o] = 0; rr_val[o] = NULL_CW; rr_mask[o] = 0; }
        if (!(flags[g] & 1)) continue;
        for (uint32_t k = 0; k < rc_n[g]; k++) {
            const uint32_t s = rc_slot[(size_t)k * G + g];
            if (s < r->len && !held(r, s)) continue;
            while (r->len <= s) push(r, null_instance());
            Inst *in = &r->insts[s];
            if (in->status < ST_ACCEPTING || popc(in->cw.mask) == 0) continue;
            const size_t o = (size_t)(rr_n[g]++) * G + g;
            rr_slot[o] = s; rr_bal[o] = in->bal; rr_val[o] = in->cw.val; rr_mask[o] = in->cw.mask;
        }
        drain(r);
    }
}

/* messages.rs:518-594 */
void orc_rsp_reconstruct_reply(void *h, const uint8_t *flags, const uint32_t *rr_n, const uint32_t *rr_slot, const uint64_t *rr_bal,
                               const uint32_t *rr_val, const uint8_t *rr_mask) {
    Cl *cl = (Cl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        Rep *r = &cl->reps[g];
        begin(r);
        if (!(flags[g] & 1)) continue;
        for (uint32_t k = 0; k < rr_n[g]; k++) {
            const size_t o = (size_t)k * G + g;
            const uint32_t s = rr_slot[o];
            if (!held(r, s)) continue;
            Inst *in = &r->insts[s];
            if (in->status < ST_EXECUTED && rr_bal[o] >= in->bal) {
                Cw c = {rr_val[o], rr_mask[o]};
                cw_absorb(r, &in->cw, c);
                if (s == r->commit_bar) commit_bar_run(r, 0);
            }
        }
        drain(r);
    }
}

/* a Heartbeat from `peer` (leadership.rs:236-340); reply[g] = 1: my Heartbeat goes back to it (fields out) */
void orc_rsp_heartbeat(void *h, const uint8_t *flags, const uint8_t *peer, const uint64_t *ballot, const uint32_t *commit_bar,
                       const uint32_t *exec_bar, const uint32_t *snap_bar, uint8_t *reply, uint64_t *o_ballot, uint32_t *o_commit,
                       uint32_t *o_exec, uint32_t *o_snap) {
    Cl *cl = (Cl *)h;
    for (uint32_t g = 0; g < cl->G; g++) {
        Rep *r = &cl->reps[g];
        begin(r);
        reply[g] = 0; o_ballot[g] = 0; o_commit[g] = o_exec[g] = o_snap[g] = 0;
        if (!(flags[g] & 1)) continue;
        /* the reply is sent inside check_leader's wake, BEFORE the commit learning below (leadership.rs:253-265) */
        const uint64_t bms_after = ballot[g] > r->bal_max_seen ? ballot[g] : r->bal_max_seen;
        const uint32_t cb = r->commit_bar, eb = r->exec_bar, sb = r->snap_bar;
        if (heard_heartbeat(r, peer[g], ballot[g], commit_bar[g], exec_bar[g], snap_bar[g])) {
            reply[g] = 1; o_ballot[g] = bms_after; o_commit[g] = cb; o_exec[g] = eb; o_snap[g] = sb;
        }
        drain(r);
    }
}

/* leadership.rs:187-217 bcast_heartbeats by a replica that is sending (flags): fields out + hears itself */
void orc_rsp_bcast_heartbeat(void *h, const uint8_t *flags, uint64_t *o_ballot, uint32_t *o_commit, uint32_t *o_exec, uint32_t *o_snap) {
    Cl *cl = (Cl *)h;
    for (uint32_t g = 0; g < cl->G; g++) {
        Rep *r = &cl->reps[g];
        begin(r);
        o_ballot[g] = 0; o_commit[g] = o_exec[g] = o_snap[g] = 0;
        if (!(flags[g] & 1)) continue;
        o_ballot[g] = r->bal_max_seen; o_commit[g] = r->commit_bar; o_exec[g] = r->exec_bar; o_snap[g] = r->snap_bar;
        (void)heard_heartbeat(r, r->id, r->bal_max_seen, r->commit_bar, r->exec_bar, r->snap_bar);
        drain(r);
    }
}

/* canonical dump: scalars [G]; per slot [W][G] by slot % W (cells outside the ring read as null) */
void orc_rsp_dump(void *h, uint8_t *leader, uint64_t *bal_prep_sent, uint64_t *bal_prepared, uint64_t *bal_max_seen, uint32_t *len,
                  uint32_t *commit_bar, uint32_t *exec_bar, uint32_t *snap_bar, uint32_t *peer_exec_bar, uint64_t *digest,
                  uint64_t *s_bal, uint8_t *s_status, uint32_t *s_val, uint8_t *s_mask, uint64_t *s_vbal, uint32_t *s_vval,
                  uint8_t *s_vmask, uint8_t *s_flags, uint32_t *s_ltrig, uint32_t *s_lendp, uint8_t *s_packs, uint8_t *s_aacks,
                  uint64_t *s_pmax, uint8_t *s_rsrc, uint32_t *s_rtrig, uint32_t *s_rendp, uint64_t *counters) {
    Cl *cl = (Cl *)h;
    const uint32_t G = cl->G;
    counters[0] = counters[1] = counters[2] = counters[3] = 0;
    for (uint32_t g = 0; g < G; g++) {
        Rep *r = &cl->reps[g];
        const uint32_t W = r->W;
        leader[g] = r->leader; bal_prep_sent[g] = r->bal_prep_sent; bal_prepared[g] = r->bal_prepared; bal_max_seen[g] = r->bal_max_seen;
        len[g] = r->len; commit_bar[g] = r->commit_bar; exec_bar[g] = r->exec_bar; snap_bar[g] = r->snap_bar; digest[g] = r->digest;
        for (int p = 0; p < cl->R; p++) peer_exec_bar[(size_t)p * G + g] = r->peer_exec_bar[p];
        counters[0] += r->n_commit; counters[1] += r->n_exec_total; counters[2] += r->n_mixed; counters[3] += r->n_redirect;
        Inst nul = null_instance();
        for (uint32_t w = 0; w < W; w++) {
            const size_t o = (size_t)w * G + g;
            const Inst *in = &nul;
            const uint32_t lo = ring_lo(r);
            uint32_t s = (lo & ~(W - 1)) | w;
            if (s < lo) s += W;
            if (s < r->len) in = &r->insts[s];
            s_bal[o] = in->bal; s_status[o] = in->status; s_val[o] = in->cw.val; s_mask[o] = in->cw.mask;
            s_vbal[o] = in->voted_bal; s_vval[o] = in->voted.val; s_vmask[o] = in->voted.mask;
            s_flags[o] = (uint8_t)(in->has_lbk | (in->has_rbk << 1) | (in->external << 2));
            s_ltrig[o] = in->has_lbk ? in->l_trig : 0; s_lendp[o] = in->has_lbk ? in->l_endp : 0;
            s_packs[o] = in->has_lbk ? in->p_acks : 0; s_aacks[o] = in->has_lbk ? in->a_acks : 0; s_pmax[o] = in->has_lbk ? in->p_max : 0;
            s_rsrc[o] = in->has_rbk ? in->r_src : NO_REP; s_rtrig[o] = in->has_rbk ? in->r_trig : 0; s_rendp[o] = in->has_rbk ? in->r_endp : 0;
        }
    }
}

/* the commands executed since the last call: (group, slot, token), group-major, execution order within a group */
uint64_t orc_rsp_take_executed(void *h, uint32_t *group, uint32_t *slot, uint32_t *val, uint64_t cap) {
    Cl *cl = (Cl *)h;
    uint64_t n = 0;
    for (uint32_t g = 0; g < cl->G; g++) {
        Rep *r = &cl->reps[g];
        for (uint32_t i = 0; i < r->n_xlog; i++, n++)
            if (n < cap) { group[n] = g; slot[n] = r->xlog[i]; val[n] = r->xlog_val[i]; }
        r->n_xlog = 0;
    }
    return n;
}

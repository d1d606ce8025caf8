/*
 * oracle/mp_oracle.c -- CPU restatement of Summerset's MultiPaxos (and the
 * RSPaxos commit-rule variant) replica event handlers, replayed over G
 * independent replica groups under the lock-step schedule LS-1.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product path; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / CPU
 * baseline.
 *
 * Deliberately literal: one `Replica` struct per (group, replica) with a
 * growable `insts` vector and `start_slot` exactly like the reference's
 * `MultiPaxosReplica` (src/protocols/multipaxos/mod.rs:387-514), handlers
 * that push WAL actions / exec submissions / peer messages onto queues, and a
 * scheduler that delivers them.  The HIP engine is written independently
 * against SoA ring windows; the two only share the input stream format.
 *
 * PARITY STATUS: the reference has NO unit tests, golden vectors or fixtures
 * for protocol outcomes and cannot be built here (no Rust
 * toolchain), so this restatement is "parity unpinned" against the reference
 * binary.  It is pinned only by hand-derived traces and the invariants the
 * reference states (mod.rs:465-468, tla+ specs) in tests/test_oracle_mp.py.
 *
 * Each handler cites the reference lines it follows
 * (paths relative to src/protocols/multipaxos/).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ST_NULL = 0, ST_PREPARING = 1, ST_ACCEPTING = 2, ST_COMMITTED = 3, ST_EXECUTED = 4 };
enum { WAL_PREPARE_BAL = 1, WAL_ACCEPT_DATA = 2, WAL_COMMIT_SLOT = 3 };
enum { MSG_PREPARE = 1, MSG_ACCEPT = 2, MSG_HEARTBEAT = 3 };
#define NO_LEADER 0xFF
#define MAXR 8

/* mod.rs:207-233 Instance (+ LeaderBookkeeping :173-190, ReplicaBookkeeping
 * :193-204).  `reqs` is an opaque batch token; 0 == empty batch. */
typedef struct {
    uint64_t bal;
    uint8_t status;
    uint32_t reqs;
    uint64_t voted_bal;
    uint32_t voted_reqs;
    uint8_t has_lbk;
    uint32_t l_trigger, l_endprep;
    uint8_t prepare_acks;
    uint64_t prepare_max_bal;
    uint8_t accept_acks;
    uint8_t has_rbk;
    uint8_t r_source;
    uint32_t r_trigger, r_endprep;
    uint8_t external;
} Inst;

typedef struct { uint8_t kind; uint32_t slot; } WalAct;

typedef struct {
    uint8_t kind;
    uint32_t slot;       /* Accept: slot; Prepare: trigger_slot; HB: commit_bar */
    uint64_t ballot;
    uint32_t reqs;       /* Accept: batch token; HB: exec_bar */
    uint32_t snap_bar;   /* HB only */
} DownMsg;

typedef struct {
    uint8_t dest;
    uint32_t slot, trigger, endprep;
    uint64_t ballot;
    uint8_t has_voted;
    uint64_t voted_bal;
    uint32_t voted_reqs;
    uint32_t accept_bar;
} PrepReply;

typedef struct { uint32_t group, slot; } CommitEv;

typedef struct Replica {
    uint8_t id, population, quorum_cnt, commit_thresh;
    uint8_t leader;
    Inst *insts;
    uint32_t n_insts, cap_insts, start_slot;
    uint64_t bal_prep_sent, bal_prepared, bal_max_seen;
    uint32_t accept_bar, commit_bar, exec_bar, snap_bar;
    uint32_t peer_exec_bar[MAXR];
    /* queues drained by the scheduler */
    WalAct *wal; uint32_t n_wal, cap_wal;
    uint32_t *execq; uint32_t n_exec, cap_exec;
    /* messages */
    DownMsg *ob[2]; uint32_t n_ob[2], cap_ob[2];   /* outbox, by tick parity */
    PrepReply *pr; uint32_t n_pr, cap_pr;          /* my PrepareReplies of this tick */
    /* heartbeat record published at the start of the heartbeat round */
    uint64_t hb_bal; uint32_t hb_commit, hb_exec, hb_snap;
    /* counters */
    uint64_t n_commits, n_redirect, n_reject;
} Replica;

typedef struct {
    uint32_t G; uint8_t R; uint32_t W, win_reserve, cap;
    uint32_t tick;
    Replica *reps;          /* [G][R] */
    uint8_t *overflow;      /* [G] sticky */
    uint64_t *ack;          /* per group scratch: [R senders][cap][R] */
    CommitEv *commits[MAXR]; uint64_t n_commits[MAXR], cap_commits[MAXR];
    int record_commits;
    int cur_group; int cur_parity;
} Cluster;

static Cluster *g_cl; /* current cluster, for the overflow flag from handlers */

/* ---------- small vector helpers ---------------------------------------- */
#define VPUSH(arr, n, cap, T, val) do { \
    if ((n) == (cap)) { (cap) = (cap) ? (cap) * 2 : 8; (arr) = (T *)realloc((arr), sizeof(T) * (cap)); } \
    (arr)[(n)++] = (val); } while (0)

static int is_leader(const Replica *r) { return r->leader == r->id; }
static uint32_t log_end(const Replica *r) { return r->start_slot + r->n_insts; }
static Inst *inst_at(Replica *r, uint32_t slot) { return &r->insts[slot - r->start_slot]; }

/* mod.rs:527-538 null_instance */
static Inst null_instance(void) { Inst i; memset(&i, 0, sizeof(i)); return i; }

/* push a null instance; ring-window harness guard (ours):
 * a replica may hold at most W live instances. Returns 0 on overflow. */
static int push_null(Replica *r) {
    if (r->n_insts >= g_cl->W) { g_cl->overflow[g_cl->cur_group] = 1; return 0; }
    VPUSH(r->insts, r->n_insts, r->cap_insts, Inst, null_instance());
    return 1;
}

/* mod.rs:553-561 */
static uint64_t make_unique_ballot(const Replica *r, uint64_t base) { return (base << 8) | (uint64_t)(r->id + 1); }
static uint64_t make_greater_ballot(const Replica *r, uint64_t bal) { return make_unique_ballot(r, (bal >> 8) + 1); }

static void wal_submit(Replica *r, uint8_t kind, uint32_t slot) {
    WalAct a = { kind, slot };
    VPUSH(r->wal, r->n_wal, r->cap_wal, WalAct, a);
}
static void send_down(Replica *r, int parity, DownMsg m) {
    VPUSH(r->ob[parity], r->n_ob[parity], r->cap_ob[parity], DownMsg, m);
}

static int popcnt8(uint8_t v) { int c = 0; while (v) { c += v & 1; v >>= 1; } return c; }

/* forward decls */
static void handle_msg_accept_reply(Replica *r, uint8_t peer, uint32_t slot, uint64_t ballot);
static void handle_msg_prepare_reply(Replica *r, uint8_t peer, uint32_t slot, uint32_t trigger_slot,
                                     uint32_t endprep_slot, uint64_t ballot, int has_voted,
                                     uint64_t voted_bal, uint32_t voted_reqs, uint32_t accept_bar);

/* leadership.rs:11-67 check_leader (lease branches are config-off) */
static void check_leader(Replica *r, uint8_t peer, uint64_t ballot) {
    if (ballot > r->bal_max_seen) {
        r->leader = peer;
        r->bal_max_seen = ballot;
    }
}

/* mod.rs:541-549 first_null_slot, plus the harness back-pressure guard: the
 * leader refuses to grow its log beyond W - win_reserve live instances
 * (returns UINT32_MAX). */
static uint32_t first_null_slot(Replica *r) {
    for (uint32_t s = r->exec_bar; s < log_end(r); s++)
        if (inst_at(r, s)->status == ST_NULL) return s;
    if (r->n_insts + g_cl->win_reserve >= g_cl->W) return UINT32_MAX;
    if (!push_null(r)) return UINT32_MAX;
    return log_end(r) - 1;
}

/* request.rs:112-224 handle_req_batch (read-only shortcuts are config-off) */
static void handle_req_batch(Replica *r, uint32_t reqs, int parity) {
    if (!is_leader(r) || r->bal_prepared == 0) { r->n_redirect++; return; } /* :128-154 */
    uint32_t slot = first_null_slot(r);
    if (slot == UINT32_MAX) { r->n_reject++; return; }
    Inst *inst = inst_at(r, slot);
    inst->reqs = reqs;                                   /* :162 */
    inst->has_lbk = 1; inst->l_trigger = 0; inst->l_endprep = 0; /* :168-174 */
    inst->prepare_acks = 0; inst->prepare_max_bal = 0; inst->accept_acks = 0;
    inst->external = 1;                                  /* :175 */
    inst->bal = r->bal_prepared;                         /* :180 */
    inst->status = ST_ACCEPTING;                         /* :181 */
    inst->voted_bal = inst->bal; inst->voted_reqs = reqs; /* :190 */
    wal_submit(r, WAL_ACCEPT_DATA, slot);                /* :191-201 */
    DownMsg m = { MSG_ACCEPT, slot, inst->bal, reqs, 0 }; /* :209-216 */
    send_down(r, parity, m);
}

/* leadership.rs:73-214 become_a_leader */
static void become_a_leader(Replica *r, uint8_t timeout_source, int parity) {
    if (r->leader != NO_LEADER && r->leader != timeout_source) return; /* :77-81 */
    r->leader = r->id;                                    /* :98 */
    /* :104 bcast_heartbeats() right now, carrying the OLD bal_max_seen
     * (leadership.rs:240-247); self-heard copy is a no-op */
    DownMsg hb = { MSG_HEARTBEAT, r->commit_bar, r->bal_max_seen, r->exec_bar, r->snap_bar };
    send_down(r, parity, hb);
    for (int p = 0; p < r->population; p++) r->peer_exec_bar[p] = 0; /* :107-109 */
    r->bal_prepared = 0;                                  /* :112 */
    r->bal_prep_sent = make_greater_ballot(r, r->bal_max_seen);
    r->bal_max_seen = r->bal_prep_sent;
    /* :117-130 first / last slot with status < Committed (else log end) */
    uint32_t trigger_slot = log_end(r), endprep_slot = log_end(r);
    for (uint32_t i = 0; i < r->n_insts; i++)
        if (r->insts[i].status < ST_COMMITTED) { trigger_slot = r->start_slot + i; break; }
    for (uint32_t i = r->n_insts; i > 0; i--)
        if (r->insts[i - 1].status < ST_COMMITTED) { endprep_slot = r->start_slot + i - 1; break; }
    if (trigger_slot == log_end(r)) {                     /* :131-134 */
        if (!push_null(r)) return;
    }
    for (uint32_t s = r->exec_bar; s < log_end(r); s++) { /* :142-183 */
        Inst *inst = inst_at(r, s);
        if (inst->status == ST_EXECUTED) continue;
        inst->external = 1;
        if (inst->status == ST_COMMITTED) continue;
        inst->bal = r->bal_prep_sent;
        inst->status = ST_PREPARING;
        inst->has_lbk = 1; inst->l_trigger = trigger_slot; inst->l_endprep = endprep_slot;
        inst->prepare_acks = 0; inst->prepare_max_bal = 0; inst->accept_acks = 0;
        wal_submit(r, WAL_PREPARE_BAL, s);
    }
    DownMsg m = { MSG_PREPARE, trigger_slot, r->bal_prep_sent, 0, 0 }; /* :192-198 */
    send_down(r, parity, m);
}

/* messages.rs:12-83 handle_msg_prepare */
static void handle_msg_prepare(Replica *r, uint8_t peer, uint32_t trigger_slot, uint64_t ballot) {
    if (trigger_slot < r->start_slot) return;             /* :18-20 */
    if (ballot >= r->bal_max_seen) {                      /* :29 */
        check_leader(r, peer, ballot);
        while (log_end(r) <= trigger_slot)                /* :37-39 */
            if (!push_null(r)) return;
        /* :43-52 last non-null slot (unwrap_or(0)), max'ed with trigger */
        uint32_t last = r->start_slot;
        for (uint32_t i = r->n_insts; i > 0; i--)
            if (r->insts[i - 1].status > ST_NULL) { last = r->start_slot + i - 1; break; }
        uint32_t endprep_slot = last > trigger_slot ? last : trigger_slot;
        for (uint32_t slot = trigger_slot; slot <= endprep_slot; slot++) { /* :55-79 */
            Inst *inst = inst_at(r, slot);
            inst->bal = ballot;
            inst->status = ST_PREPARING;
            inst->has_rbk = 1; inst->r_source = peer;
            inst->r_trigger = trigger_slot; inst->r_endprep = endprep_slot;
            wal_submit(r, WAL_PREPARE_BAL, slot);
        }
    }
}

/* messages.rs:87-292 handle_msg_prepare_reply (peer_accept_bar bookkeeping
 * :129-144 is lease-only and not modelled) */
static void handle_msg_prepare_reply(Replica *r, uint8_t peer, uint32_t slot, uint32_t trigger_slot,
                                     uint32_t endprep_slot, uint64_t ballot, int has_voted,
                                     uint64_t voted_bal, uint32_t voted_reqs, uint32_t accept_bar) {
    (void)accept_bar;
    if (slot < r->start_slot) return;                     /* :97-99 */
    if (ballot != r->bal_prep_sent) return;               /* :110 */
    if (!is_leader(r)) return;                            /* :112-114 */
    if (trigger_slot < r->start_slot || trigger_slot >= log_end(r)) return; /* debug_assert :116-119 */
    if (!inst_at(r, trigger_slot)->has_lbk) return;       /* :120-125 */
    uint32_t my_endprep_slot = inst_at(r, trigger_slot)->l_endprep; /* :149-153 */
    while (log_end(r) <= slot) {                          /* :154-190 */
        uint32_t this_slot = log_end(r);
        if (!push_null(r)) return;
        Inst *inst = inst_at(r, this_slot);
        inst->external = 1;
        inst->bal = r->bal_prep_sent;
        inst->status = ST_PREPARING;
        inst->has_lbk = 1; inst->l_trigger = trigger_slot; inst->l_endprep = my_endprep_slot;
        inst->prepare_acks = 0; inst->prepare_max_bal = 0; inst->accept_acks = 0;
        wal_submit(r, WAL_PREPARE_BAL, this_slot);
    }
    {
        Inst *inst = inst_at(r, slot);
        if (inst->status != ST_PREPARING || ballot < inst->bal) return; /* :196-198 */
        if (has_voted) {                                  /* :203-216 */
            if (voted_bal > inst->prepare_max_bal) {
                inst->prepare_max_bal = voted_bal;
                inst->reqs = voted_reqs;
            }
        }
    }
    if (slot == endprep_slot) {                           /* :222 */
        Inst *tinst = inst_at(r, trigger_slot);
        tinst->prepare_acks |= (uint8_t)(1u << peer);     /* :228 */
        if (popcnt8(tinst->prepare_acks) >= r->quorum_cnt) { /* :233 */
            r->bal_prepared = ballot;                     /* :236 */
            for (uint32_t s = trigger_slot; s < log_end(r); s++) { /* :238-286 */
                Inst *inst = inst_at(r, s);
                if (inst->status != ST_PREPARING) continue;
                inst->status = ST_ACCEPTING;
                wal_submit(r, WAL_ACCEPT_DATA, s);
                DownMsg m = { MSG_ACCEPT, s, ballot, inst->reqs, 0 };
                /* Accepts born in the reply round travel in the NEXT tick */
                send_down(r, g_cl->cur_parity ^ 1, m);
            }
        }
    }
}

/* messages.rs:295-367 handle_msg_accept */
static void handle_msg_accept(Replica *r, uint8_t peer, uint32_t slot, uint64_t ballot, uint32_t reqs) {
    if (slot < r->start_slot) return;                     /* :302-304 */
    if (ballot >= r->bal_max_seen) {                      /* :313 */
        check_leader(r, peer, ballot);
        while (log_end(r) <= slot)                        /* :321-323 */
            if (!push_null(r)) return;
        Inst *inst = inst_at(r, slot);
        inst->bal = ballot;                               /* :327 */
        inst->status = ST_ACCEPTING;
        inst->reqs = reqs;
        if (inst->has_rbk) inst->r_source = peer;         /* :331-339 */
        else { inst->has_rbk = 1; inst->r_source = peer; inst->r_trigger = 0; inst->r_endprep = 0; }
        inst->voted_bal = ballot; inst->voted_reqs = reqs; /* :351 */
        wal_submit(r, WAL_ACCEPT_DATA, slot);             /* :352-358 */
    }
}

/* messages.rs:370-443 handle_msg_accept_reply.  commit_thresh == quorum_cnt
 * for MultiPaxos (:412); == majority + fault_tolerance for RSPaxos
 * (rspaxos/messages.rs:438-439). */
static void handle_msg_accept_reply(Replica *r, uint8_t peer, uint32_t slot, uint64_t ballot) {
    if (slot < r->start_slot) return;                     /* :377-379 */
    if (ballot == r->bal_prepared) {                      /* :388 */
        if (slot >= log_end(r)) return;                   /* debug_assert :389 */
        Inst *inst = inst_at(r, slot);
        if (!is_leader(r) || inst->status != ST_ACCEPTING || ballot < inst->bal) return; /* :394-399 */
        if (!inst->has_lbk) return;                       /* debug_assert :402 */
        if (inst->accept_acks & (1u << peer)) return;     /* :404-406 */
        inst->accept_acks |= (uint8_t)(1u << peer);       /* :409 */
        if (popcnt8(inst->accept_acks) >= r->commit_thresh) { /* :412 */
            inst->status = ST_COMMITTED;
            r->n_commits++;
            if (g_cl->record_commits) {
                CommitEv ev = { (uint32_t)g_cl->cur_group, slot };
                VPUSH(g_cl->commits[r->id], g_cl->n_commits[r->id], g_cl->cap_commits[r->id], CommitEv, ev);
            }
            wal_submit(r, WAL_COMMIT_SLOT, slot);         /* :427-433 */
        }
    }
}

/* durability.rs:10-82 handle_logged_prepare_bal */
static void handle_logged_prepare_bal(Replica *r, uint32_t slot) {
    if (slot < r->start_slot) return;
    Inst *inst = inst_at(r, slot);
    int has_voted = inst->voted_bal > 0;                  /* :23-27 */
    if (is_leader(r)) {                                   /* :29-49 */
        if (inst->has_lbk && slot <= inst->l_endprep)
            handle_msg_prepare_reply(r, r->id, slot, inst->l_trigger, inst->l_endprep, inst->bal,
                                     has_voted, inst->voted_bal, inst->voted_reqs, r->accept_bar);
    } else if (inst->has_rbk) {                           /* :50-78 */
        PrepReply pr = { inst->r_source, slot, inst->r_trigger, inst->r_endprep, inst->bal,
                         (uint8_t)has_voted, inst->voted_bal, inst->voted_reqs, r->accept_bar };
        VPUSH(r->pr, r->n_pr, r->cap_pr, PrepReply, pr);
    }
}

/* durability.rs:85-145 handle_logged_accept_data.  Returns the AcceptReply
 * ballot a follower sends (0 = none) and its destination through *dest. */
static uint64_t handle_logged_accept_data(Replica *r, uint32_t slot, uint8_t *dest) {
    uint64_t reply = 0;
    if (slot < r->start_slot) return 0;
    Inst *inst = inst_at(r, slot);
    if (is_leader(r)) {                                   /* :99-107 */
        handle_msg_accept_reply(r, r->id, slot, inst->bal);
    } else if (inst->has_rbk) {                           /* :108-131 */
        reply = inst->bal;
        *dest = inst->r_source;
    }
    if (slot == r->accept_bar) {                          /* :134-142 */
        while (r->accept_bar < log_end(r)) {
            if (inst_at(r, r->accept_bar)->status < ST_ACCEPTING) break;
            r->accept_bar++;
        }
    }
    return reply;
}

/* durability.rs:148-218 handle_logged_commit_slot (urgent_commit_notice off).
 * rspaxos == 1 selects rspaxos/durability.rs:125-186: bounded by the log end
 * and gated on shard availability (`avail` = shards this replica holds). */
static void handle_logged_commit_slot(Replica *r, uint32_t slot) {
    if (slot < r->start_slot) return;
    if (slot == r->commit_bar) {                          /* :161 */
        while (r->commit_bar < r->accept_bar) {           /* :162 */
            Inst *inst = inst_at(r, r->commit_bar);
            if (inst->status < ST_COMMITTED) break;       /* :164-166 */
            if (inst->reqs == 0) inst->status = ST_EXECUTED; /* :171-172 */
            else if (inst->status == ST_COMMITTED)        /* :173-181 */
                VPUSH(r->execq, r->n_exec, r->cap_exec, uint32_t, r->commit_bar);
            r->commit_bar++;                              /* :189 */
        }
    }
}

/* execution.rs:10-82 handle_cmd_result (one command per batch token) */
static void handle_cmd_result(Replica *r, uint32_t slot) {
    if (slot < r->start_slot) return;
    Inst *inst = inst_at(r, slot);
    inst->status = ST_EXECUTED;                           /* :57 */
    if (slot == r->exec_bar) {                            /* :70-78 */
        while (r->exec_bar < log_end(r)) {
            if (inst_at(r, r->exec_bar)->status < ST_EXECUTED) break;
            r->exec_bar++;
        }
    }
}

/* leadership.rs:372-427 advance_commit_bar */
static void advance_commit_bar(Replica *r, uint64_t ballot, uint32_t commit_bar) {
    if (commit_bar > r->commit_bar) {
        while (log_end(r) < commit_bar)                   /* :380-382 */
            if (!push_null(r)) return;
        for (uint32_t slot = r->commit_bar; slot < commit_bar; slot++) { /* :385-416 */
            Inst *inst = inst_at(r, slot);
            if (inst->bal < ballot || inst->status < ST_ACCEPTING) break;
            else if (inst->status >= ST_COMMITTED) continue;
            inst->status = ST_COMMITTED;
            wal_submit(r, WAL_COMMIT_SLOT, slot);
        }
    }
}

/* leadership.rs:270-346 heard_heartbeat */
static void heard_heartbeat(Replica *r, uint8_t peer, uint64_t ballot, uint32_t commit_bar,
                            uint32_t exec_bar, uint32_t snap_bar) {
    if (peer != r->id) check_leader(r, peer, ballot);     /* :278-285 */
    if (ballot < r->bal_max_seen) return;                 /* :303-305 */
    if (exec_bar < r->exec_bar) return;                   /* :312-314 */
    advance_commit_bar(r, ballot, commit_bar);            /* :318 */
    if (peer != r->id) {                                  /* :320-342 */
        if (exec_bar > r->peer_exec_bar[peer]) {
            r->peer_exec_bar[peer] = exec_bar;
            int passed_cnt = 1;
            for (int p = 0; p < r->population; p++)
                if (p != r->id && r->peer_exec_bar[p] >= exec_bar) passed_cnt++;
            if (passed_cnt == r->population) r->snap_bar = exec_bar;
        }
        if (snap_bar > r->snap_bar) r->snap_bar = snap_bar;
    }
}

/* snapshot.rs:121-186 take_new_snapshot, in-memory log trim only (:137-140,
 * :170-171): drop instances below `new_start`.  The KV dump, WAL discard and
 * the two appeasement heartbeat broadcasts are not modelled.  The reference
 * trims to min(snap_bar, exec_bar) on a timer that is off by default, and its
 * snap_bar never moves under sustained load (heard_heartbeat returns at
 * leadership.rs:312-314 before the peer_exec_bar update whenever the sender
 * lags).  The ring-window harness instead trims, at the end of every
 * heartbeat round, to the same invariant computed exactly: the minimum of my
 * exec_bar and every peer's exec_bar as carried by this round's heartbeats
 * (DESIGN.md §3.4).  snap_bar / peer_exec_bar stay faithful state. */
static void trim_log(Replica *r, uint32_t new_start) {
    if (new_start <= r->start_slot) return;
    uint32_t k = new_start - r->start_slot;
    memmove(r->insts, r->insts + k, sizeof(Inst) * (r->n_insts - k));
    r->n_insts -= k;
    r->start_slot = new_start;
}

/* ---------- scheduler (LS-1) -------------------------------------------- */

/* Drain the WAL queue ("every WAL append completes right after the handler
 * that submitted it returns", durability.rs:221-254 handle_log_result) and the
 * executor queue.  AcceptReplies produced on the way are stored through
 * ack_out (the cell for the Accept being processed), PrepareReplies go to the
 * replica's reply list. */
static void drain(Replica *r, uint64_t *ack_out, uint8_t sender) {
    uint32_t wi = 0, ei = 0;
    for (;;) {
        if (wi < r->n_wal) {
            WalAct a = r->wal[wi++];
            switch (a.kind) {
            case WAL_PREPARE_BAL: handle_logged_prepare_bal(r, a.slot); break;
            case WAL_ACCEPT_DATA: {
                uint8_t dest = NO_LEADER;
                uint64_t rep = handle_logged_accept_data(r, a.slot, &dest);
                if (rep) {
                    /* LS-1 regularity: a follower's AcceptReply always answers
                     * the Accept being processed, to its sender */
                    if (ack_out && dest == sender) *ack_out = rep;
                    else g_cl->overflow[g_cl->cur_group] |= 2; /* would be a schedule bug */
                }
            } break;
            case WAL_COMMIT_SLOT: handle_logged_commit_slot(r, a.slot); break;
            }
            continue;
        }
        if (ei < r->n_exec) { handle_cmd_result(r, r->execq[ei++]); continue; }
        break;
    }
    r->n_wal = 0; r->n_exec = 0;
}

/* ackctl word: bits 0..23 = R replica ids, 3 bits each, in delivery order;
 * bits 24..31 = drop mask by replica id.  CTL_IDENTITY = ids 0,1,2,...,7. */
#define CTL_IDENTITY 0x00FAC688u
static uint32_t ctl_order(uint32_t ctl, int i) { return (ctl >> (3 * i)) & 7u; }
static uint32_t ctl_drop(uint32_t ctl) { return ctl >> 24; }

void *orc_mp_new(uint32_t G, uint8_t R, uint32_t W, uint32_t win_reserve, uint32_t cap,
                 uint8_t commit_extra, int record_commits) {
    Cluster *cl = (Cluster *)calloc(1, sizeof(Cluster));
    cl->G = G; cl->R = R; cl->W = W; cl->win_reserve = win_reserve; cl->cap = cap;
    cl->record_commits = record_commits;
    cl->reps = (Replica *)calloc((size_t)G * R, sizeof(Replica));
    cl->overflow = (uint8_t *)calloc(G, 1);
    cl->ack = (uint64_t *)calloc((size_t)R * cap * R, sizeof(uint64_t));
    for (uint32_t g = 0; g < G; g++)
        for (uint8_t i = 0; i < R; i++) {
            Replica *r = &cl->reps[(size_t)g * R + i];
            r->id = i; r->population = R;
            r->quorum_cnt = (uint8_t)(R / 2 + 1);         /* mod.rs:774 */
            r->commit_thresh = (uint8_t)(r->quorum_cnt + commit_extra);
            r->leader = NO_LEADER;
        }
    return cl;
}

void orc_mp_free(void *h) {
    Cluster *cl = (Cluster *)h;
    for (size_t i = 0; i < (size_t)cl->G * cl->R; i++) {
        Replica *r = &cl->reps[i];
        free(r->insts); free(r->wal); free(r->execq); free(r->ob[0]); free(r->ob[1]); free(r->pr);
    }
    for (int i = 0; i < MAXR; i++) free(cl->commits[i]);
    free(cl->reps); free(cl->overflow); free(cl->ack); free(cl);
}

/*
 * One lock-step tick for every group.  Inputs are SoA over groups:
 *   timeout_rep[G], timeout_src[G] : HearTimeout event (rep 0xFF = none)
 *   req_target[G], req_cnt[G], req_val[S][G] : client batches for the tick
 *   ackctl[cap][G] : per outbox entry, peer order (3 bits each) | drop mask<<24
 *   do_heartbeat   : run the heartbeat round + log trim this tick
 */
void orc_mp_tick(void *h, const uint8_t *timeout_rep, const uint8_t *timeout_src,
                 const uint8_t *req_target, const uint32_t *req_cnt, const uint32_t *req_val,
                 uint32_t S, const uint32_t *ackctl, int do_heartbeat) {
    Cluster *cl = (Cluster *)h;
    g_cl = cl;
    const uint32_t G = cl->G; const int R = cl->R; const uint32_t cap = cl->cap;
    const int par = (int)(cl->tick & 1);
    cl->cur_parity = par;
    for (uint32_t g = 0; g < G; g++) {
        if (cl->overflow[g]) continue;
        cl->cur_group = (int)g;
        Replica *reps = &cl->reps[(size_t)g * R];
        /* R1: local events */
        for (int i = 0; i < R; i++) {
            Replica *r = &reps[i];
            r->n_pr = 0;
            if (timeout_rep && timeout_rep[g] == i) {
                become_a_leader(r, timeout_src[g], par);
                drain(r, NULL, NO_LEADER);
            }
            if (req_target && req_target[g] == i)
                for (uint32_t k = 0; k < req_cnt[g] && k < S; k++) {
                    handle_req_batch(r, req_val[(size_t)k * G + g], par);
                    drain(r, NULL, NO_LEADER);
                }
            if (r->n_ob[par] > cap) cl->overflow[g] = 1;
        }
        if (cl->overflow[g]) continue;
        /* R2: deliver leader->peers messages, sender-major */
        memset(cl->ack, 0, sizeof(uint64_t) * (size_t)R * cap * R);
        for (int i = 0; i < R; i++) {
            Replica *r = &reps[i];
            for (int s = 0; s < R; s++) {
                if (s == i) continue;
                Replica *snd = &reps[s];
                for (uint32_t j = 0; j < snd->n_ob[par]; j++) {
                    DownMsg m = snd->ob[par][j];
                    uint64_t *cell = &cl->ack[((size_t)s * cap + j) * R + i];
                    switch (m.kind) {
                    case MSG_PREPARE: handle_msg_prepare(r, (uint8_t)s, m.slot, m.ballot); drain(r, NULL, (uint8_t)s); break;
                    case MSG_ACCEPT: handle_msg_accept(r, (uint8_t)s, m.slot, m.ballot, m.reqs); drain(r, cell, (uint8_t)s); break;
                    case MSG_HEARTBEAT: heard_heartbeat(r, (uint8_t)s, m.ballot, m.slot, m.reqs, m.snap_bar); drain(r, NULL, (uint8_t)s); break;
                    }
                    if (cl->overflow[g]) break;
                }
                if (cl->overflow[g]) break;
            }
            if (cl->overflow[g]) break;
        }
        if (cl->overflow[g]) continue;
        /* R3: deliver replies to their destination, for each receiver d */
        for (int d = 0; d < R; d++) {
            Replica *r = &reps[d];
            uint32_t tickctl = ackctl ? ackctl[g] : CTL_IDENTITY;
            /* (a) PrepareReplies: senders in the tick's peer order, FIFO each */
            for (int oi = 0; oi < R; oi++) {
                int s = (int)ctl_order(tickctl, oi);
                if (s == d || s >= R) continue;
                Replica *snd = &reps[s];
                for (uint32_t k = 0; k < snd->n_pr; k++) {
                    PrepReply *p = &snd->pr[k];
                    if (p->dest != d) continue;
                    handle_msg_prepare_reply(r, (uint8_t)s, p->slot, p->trigger, p->endprep, p->ballot,
                                             p->has_voted, p->voted_bal, p->voted_reqs, p->accept_bar);
                    drain(r, NULL, NO_LEADER);
                    if (cl->overflow[g]) break;
                }
            }
            /* (b) AcceptReplies to my Accepts of this tick, entry-major */
            for (uint32_t j = 0; j < r->n_ob[par]; j++) {
                DownMsg m = r->ob[par][j];
                if (m.kind != MSG_ACCEPT) continue;
                uint32_t ctl = ackctl ? ackctl[(size_t)j * G + g] : CTL_IDENTITY;
                for (int oi = 0; oi < R; oi++) {
                    int s = (int)ctl_order(ctl, oi);
                    if (s == d || s >= R) continue;
                    if (ctl_drop(ctl) & (1u << s)) continue;
                    uint64_t bal = cl->ack[((size_t)d * cap + j) * R + s];
                    if (!bal) continue;
                    handle_msg_accept_reply(r, (uint8_t)s, m.slot, bal);
                    drain(r, NULL, NO_LEADER);
                }
            }
            r->n_ob[par] = 0;
            if (r->n_ob[par ^ 1] > cap) cl->overflow[g] = 1;
        }
        if (cl->overflow[g]) continue;
        /* R4: all-to-all heartbeats (leadership.rs:217-265, mod.rs:695), then trim */
        if (do_heartbeat) {
            for (int i = 0; i < R; i++) {
                Replica *r = &reps[i];
                r->hb_bal = r->bal_max_seen; r->hb_commit = r->commit_bar;
                r->hb_exec = r->exec_bar; r->hb_snap = r->snap_bar;
                heard_heartbeat(r, r->id, r->hb_bal, r->hb_commit, r->hb_exec, r->hb_snap);
                drain(r, NULL, NO_LEADER);
            }
            /* LS-1 order of the peers' heartbeats at a replica: the replica it currently follows first
             * (the one heartbeat that moves its commit bar), then the others by ascending id */
            for (int i = 0; i < R; i++) {
                Replica *r = &reps[i];
                const int first = (r->leader != NO_LEADER && r->leader != i && r->leader < R) ? r->leader : -1;
                for (int k = -1; k < R; k++) {
                    const int s = k < 0 ? first : k;
                    if (s < 0 || s == i || (k >= 0 && s == first)) continue;
                    Replica *snd = &reps[s];
                    heard_heartbeat(r, (uint8_t)s, snd->hb_bal, snd->hb_commit, snd->hb_exec, snd->hb_snap);
                    drain(r, NULL, NO_LEADER);
                    if (cl->overflow[g]) break;
                }
            }
            for (int i = 0; i < R; i++) {
                uint32_t bound = reps[i].exec_bar;
                for (int s = 0; s < R; s++)
                    if (s != i && reps[s].hb_exec < bound) bound = reps[s].hb_exec;
                trim_log(&reps[i], bound);
            }
        }
    }
    cl->tick++;
}

/* ---------- state export (canonical, window-indexed by slot % W) -------- */
typedef struct {
    uint8_t leader;
    uint64_t bal_prep_sent, bal_prepared, bal_max_seen;
    uint32_t start_slot, log_len, accept_bar, commit_bar, exec_bar, snap_bar;
    uint32_t peer_exec_bar[MAXR];
    uint64_t n_commits, n_redirect, n_reject;
} OrcMpScalars;

void orc_mp_get_scalars(void *h, uint32_t g, uint8_t rep, OrcMpScalars *out) {
    Cluster *cl = (Cluster *)h;
    Replica *r = &cl->reps[(size_t)g * cl->R + rep];
    memset(out, 0, sizeof(*out));
    out->leader = r->leader;
    out->bal_prep_sent = r->bal_prep_sent; out->bal_prepared = r->bal_prepared; out->bal_max_seen = r->bal_max_seen;
    out->start_slot = r->start_slot; out->log_len = log_end(r);
    out->accept_bar = r->accept_bar; out->commit_bar = r->commit_bar; out->exec_bar = r->exec_bar;
    out->snap_bar = r->snap_bar;
    for (int p = 0; p < MAXR; p++) out->peer_exec_bar[p] = (p == r->id) ? 0 : r->peer_exec_bar[p];
    out->n_commits = r->n_commits; out->n_redirect = r->n_redirect; out->n_reject = r->n_reject;
}

/* SoA dumps over all groups for one replica id; arrays sized [G] / [W][G].
 * Slots outside [start_slot, log_len) read as all-zero. */
void orc_mp_dump(void *h, uint8_t rep,
                 uint8_t *leader, uint64_t *bal_prep_sent, uint64_t *bal_prepared, uint64_t *bal_max_seen,
                 uint32_t *start_slot, uint32_t *log_len, uint32_t *accept_bar, uint32_t *commit_bar,
                 uint32_t *exec_bar, uint32_t *snap_bar, uint32_t *peer_exec_bar /*[R][G]*/,
                 uint64_t *s_bal, uint8_t *s_status, uint32_t *s_reqs, uint64_t *s_vbal, uint32_t *s_vreqs,
                 uint8_t *s_flags, uint8_t *s_acks, uint8_t *s_packs, uint64_t *s_pmax,
                 uint32_t *s_ltrig, uint32_t *s_lendp, uint8_t *s_src, uint32_t *s_rtrig, uint32_t *s_rendp,
                 uint8_t *overflow) {
    Cluster *cl = (Cluster *)h;
    const uint32_t G = cl->G, W = cl->W;
    for (uint32_t g = 0; g < G; g++) {
        Replica *r = &cl->reps[(size_t)g * cl->R + rep];
        leader[g] = r->leader;
        bal_prep_sent[g] = r->bal_prep_sent; bal_prepared[g] = r->bal_prepared; bal_max_seen[g] = r->bal_max_seen;
        start_slot[g] = r->start_slot; log_len[g] = log_end(r);
        accept_bar[g] = r->accept_bar; commit_bar[g] = r->commit_bar; exec_bar[g] = r->exec_bar;
        snap_bar[g] = r->snap_bar;
        for (int p = 0; p < cl->R; p++) peer_exec_bar[(size_t)p * G + g] = (p == r->id) ? 0 : r->peer_exec_bar[p];
        overflow[g] = cl->overflow[g];
        for (uint32_t w = 0; w < W; w++) {
            size_t o = (size_t)w * G + g;
            s_bal[o] = 0; s_status[o] = 0; s_reqs[o] = 0; s_vbal[o] = 0; s_vreqs[o] = 0; s_flags[o] = 0;
            s_acks[o] = 0; s_packs[o] = 0; s_pmax[o] = 0; s_ltrig[o] = 0; s_lendp[o] = 0; s_src[o] = 0;
            s_rtrig[o] = 0; s_rendp[o] = 0;
        }
        for (uint32_t i = 0; i < r->n_insts; i++) {
            uint32_t slot = r->start_slot + i;
            size_t o = (size_t)(slot % W) * G + g;
            Inst *in = &r->insts[i];
            s_bal[o] = in->bal; s_status[o] = in->status; s_reqs[o] = in->reqs;
            s_vbal[o] = in->voted_bal; s_vreqs[o] = in->voted_reqs;
            s_flags[o] = (uint8_t)((in->has_lbk ? 1 : 0) | (in->has_rbk ? 2 : 0) | (in->external ? 4 : 0));
            s_acks[o] = in->has_lbk ? in->accept_acks : 0;
            s_packs[o] = in->has_lbk ? in->prepare_acks : 0;
            s_pmax[o] = in->has_lbk ? in->prepare_max_bal : 0;
            s_ltrig[o] = in->has_lbk ? in->l_trigger : 0;
            s_lendp[o] = in->has_lbk ? in->l_endprep : 0;
            s_src[o] = in->has_rbk ? in->r_source : 0;
            s_rtrig[o] = in->has_rbk ? in->r_trigger : 0;
            s_rendp[o] = in->has_rbk ? in->r_endprep : 0;
        }
    }
}

uint64_t orc_mp_total_commits(void *h, uint8_t rep) {
    Cluster *cl = (Cluster *)h;
    uint64_t t = 0;
    for (uint32_t g = 0; g < cl->G; g++) t += cl->reps[(size_t)g * cl->R + rep].n_commits;
    return t;
}

/* recorded leader-side commit events of replica `rep`: returns count, copies
 * up to max (group, slot) pairs, then clears the list */
uint64_t orc_mp_take_commits(void *h, uint8_t rep, uint32_t *groups, uint32_t *slots, uint64_t max) {
    Cluster *cl = (Cluster *)h;
    uint64_t n = cl->n_commits[rep];
    for (uint64_t i = 0; i < n && i < max; i++) {
        groups[i] = cl->commits[rep][i].group;
        slots[i] = cl->commits[rep][i].slot;
    }
    cl->n_commits[rep] = 0;
    return n;
}

/* Synthetic initial condition of SURVEY.md §8d config 2: replica `rep` has
 * already completed phase 1 on an empty log with ballot
 * make_unique_ballot(1) and everyone has seen that ballot. */
void orc_mp_preset_leader(void *h, uint8_t rep) {
    Cluster *cl = (Cluster *)h;
    for (uint32_t g = 0; g < cl->G; g++)
        for (uint8_t i = 0; i < cl->R; i++) {
            Replica *r = &cl->reps[(size_t)g * cl->R + i];
            uint64_t b = (1ull << 8) | (uint64_t)(rep + 1);
            r->leader = rep;
            r->bal_max_seen = b;
            if (i == rep) { r->bal_prep_sent = b; r->bal_prepared = b; }
        }
}

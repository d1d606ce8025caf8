/*
 * oracle/raft_oracle.c -- CPU restatement of the Raft hot path of Summerset over
 * G independent groups, one replica per group: leader side (log append +
 * AppendEntriesReply handling: match-index quorum -> last_commit, last_snap,
 * next_slot back-off), follower side (AppendEntries: consistency check, conflict
 * hint, truncate, append, commit learning) and the term / vote state machine
 * (become_a_candidate, RequestVote, RequestVoteReply, become_the_leader).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/mp_oracle.c header for the rules).
 *
 * Follows src/protocols/raft/:
 *   handle_req_batch                    request.rs:10-91
 *   handle_logged_leader_append         durability.rs:12-94   (try_next_slot)
 *   check_term                          leadership.rs:16-72
 *   handle_msg_append_entries_reply     messages.rs:222-388
 *   dummy 0-th entry                    recovery.rs:96-102
 *   become_the_leader init              leadership.rs:145-179, mod.rs:553-562
 *   handle_msg_append_entries           messages.rs:13-218
 *   handle_logged_follower_append       durability.rs:97-132
 *   become_a_candidate                  leadership.rs:76-142
 *   handle_msg_request_vote             messages.rs:391-482
 *   handle_msg_request_vote_reply       messages.rs:485-510
 *   bcast_heartbeats                    leadership.rs:182-218
 * and, for the CRaft leader variant (orc_craft_*; src/protocols/craft/ is a fork of raft/ -- only the leader's
 * reply / heartbeat path of the fork is restated, on a leader that created every entry of its log itself):
 *   handle_msg_append_entries_reply     craft/messages.rs:256-404  (no stale-success test, the
 *                                       `majority + fault_tolerance` / full-copy commit rule :301-313)
 *   bcast_heartbeats + fallback check   craft/leadership.rs:249-291
 *   switch_assignment_mode              craft/leadership.rs:80-141
 *   shard assignment of an entry        craft/request.rs:71-100, craft/messages.rs:416-460
 *   Heartbeater reply counters          server/heartbeat.rs:117-119,240-296 (update_bcast_cnts, update_heard_cnt)
 * and the CRaft FOLLOWER (orc_craft_handle_append_entries, orc_craft_handle_reconstruct):
 *   handle_msg_append_entries           craft/messages.rs:14-254   (the consistency check also on heartbeats, the leader
 *                                       recorded also on a failed check, shards of a re-sent entry absorbed, execution
 *                                       only with `majority` shards and after reconstruct_data when too few are data)
 *   handle_logged_follower_append       craft/durability.rs:129-165 (identical to raft's)
 *   handle_msg_reconstruct              craft/messages.rs:622-663
 *   RSCodeword::absorb_other / reconstruct_data / avail_shards / avail_data_shards   utils/rscoding.rs:296-, as bitmaps
 * An entry's codeword is its availability bitmap over the population's shards (data shards 0 .. majority-1); payload
 * bytes are the RS kernels' business.  data_len of a re-sent entry equals the stored one (same slot, same term = same
 * entry), so craft/messages.rs:133-134's data_len test is always true here.  The leader's own shard gate with the
 * Reconstruct slots it asks for (craft/messages.rs:315-358; orc_craft_take_reconstructs drains what a call queued) and
 * handle_msg_reconstruct_reply (:665-745; its HashMap of slots is walked in the order given -- the outcome does not
 * depend on it: absorbs commute and the execution loop restarts whenever the slot behind last_commit arrives).
 * WAL completions are inline (LS-1 rule 0); timers, the WAL file
 * offsets and the `external` reply flag of entries are not modelled.
 * Deliberately literal (forward loops over the log tail exactly as written).
 *
 * PARITY STATUS: "parity unpinned" -- the reference has no unit tests or
 * fixtures for these handlers and cannot be built here.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ROLE_FOLLOWER = 0, ROLE_CANDIDATE = 1, ROLE_LEADER = 2 };
#define MAXR 8
#define NO_LEADER 0xFF
#define NONE32 0xFFFFFFFFu

typedef struct {
    uint8_t id, population, quorum_cnt, commit_thresh;
    uint8_t role, leader;
    uint8_t voted_for;        /* NO_LEADER = None */
    uint8_t votes;            /* votes_granted as a bitmask */
    uint64_t curr_term;
    uint64_t *log_term;       /* term of every entry; index = slot - start_slot */
    uint8_t *log_mask;        /* CRaft: avail_shards_map of every entry's codeword */
    uint64_t n_recon_data, n_postponed;   /* CRaft follower: reconstruct_data calls, executions postponed for lack of shards */
    uint32_t last_recon;                  /* CRaft leader: highest slot a Reconstruct was asked for (craft/mod.rs) */
    uint32_t rq_n, rq_slot[16]; uint64_t rq_term[16];   /* Reconstruct slots queued since the last take (cap 16, more are dropped) */
    uint32_t n_log, cap_log, start_slot;
    uint32_t ring_W, ring_lo; /* harness guard shared with the engine, whose log is a ring of W entry terms: once the
                               * log has reached length n, slots below n - W are gone for good (also after a truncation) */
    uint32_t last_commit, last_snap;
    uint32_t next_slot[MAXR], try_next_slot[MAXR], match_slot[MAXR];
    uint64_t n_committed, n_redirect, n_reject, n_sent;
    uint64_t n_exec, n_trunc; /* follower: entries submitted for execution, log truncations */
    uint64_t n_ring_guard;    /* follower: entries of an AppendEntries skipped by the ring guard (NOT the reference's rule: see
                                 orc_raft_ring_guard_hits) */
    uint32_t ae_first[MAXR];  /* first slot sent to each peer during the current append call (NONE32 = nothing) */
    /* CRaft leader variant */
    uint8_t craft, fault_tolerance, full_copy_mode, repeat_threshold;
    uint8_t peer_alive;       /* heartbeat.rs:57 Bitmap, bit p; starts all true (:131) */
    uint64_t hb_replied[MAXR], hb_seen[MAXR];   /* heartbeat.rs:52 reply_cnts .0 / .1, start (1, 0, 0) (:117-119) */
    uint8_t hb_repeat[MAXR];                    /* .2 */
} RaftRep;

typedef struct {
    uint32_t G, W; uint8_t R;
    RaftRep *reps;
} RaftCl;

static uint32_t log_end(const RaftRep *r) { return r->start_slot + r->n_log; }

static void log_push_m(RaftRep *r, uint64_t term, uint8_t mask) {
    if (r->n_log == r->cap_log) {
        r->cap_log = r->cap_log ? r->cap_log * 2 : 16;
        r->log_term = (uint64_t *)realloc(r->log_term, sizeof(uint64_t) * r->cap_log);
        r->log_mask = (uint8_t *)realloc(r->log_mask, r->cap_log);
    }
    r->log_mask[r->n_log] = mask;
    r->log_term[r->n_log++] = term;
    if (r->ring_W && log_end(r) > r->ring_W && log_end(r) - r->ring_W > r->ring_lo) r->ring_lo = log_end(r) - r->ring_W;
}

/* an entry its holder created (or, in plain Raft, any entry): every shard (craft/request.rs:71-76) */
static void log_push(RaftRep *r, uint64_t term) { log_push_m(r, term, (uint8_t)((1u << r->population) - 1u)); }

void *orc_raft_new(uint32_t G, uint8_t R, uint32_t W, uint8_t leader_id, uint64_t term, uint8_t commit_extra) {
    RaftCl *cl = (RaftCl *)calloc(1, sizeof(RaftCl));
    cl->G = G; cl->R = R; cl->W = W;
    cl->reps = (RaftRep *)calloc(G, sizeof(RaftRep));
    for (uint32_t g = 0; g < G; g++) {
        RaftRep *r = &cl->reps[g];
        r->id = leader_id; r->population = R; r->ring_W = W;
        r->quorum_cnt = (uint8_t)(R / 2 + 1);
        r->commit_thresh = (uint8_t)(r->quorum_cnt + commit_extra);
        log_push(r, 0);                                   /* recovery.rs:96-102 dummy entry */
        r->role = ROLE_LEADER; r->leader = leader_id; r->curr_term = term; r->voted_for = NO_LEADER;
        for (int p = 0; p < R; p++) {                     /* leadership.rs:159-168 */
            r->next_slot[p] = log_end(r); r->try_next_slot[p] = log_end(r); r->match_slot[p] = 0;
        }
    }
    return cl;
}

void orc_raft_free(void *h) {
    RaftCl *cl = (RaftCl *)h;
    for (uint32_t g = 0; g < cl->G; g++) { free(cl->reps[g].log_term); free(cl->reps[g].log_mask); }
    free(cl->reps); free(cl);
}

/* durability.rs:12-94 (state effects only: which peers get entries, try_next) */
static void handle_logged_leader_append(RaftRep *r, uint32_t slot) {
    if (slot < r->start_slot || r->role != ROLE_LEADER) return;
    for (int peer = 0; peer < r->population; peer++) {
        if (peer == r->id || r->try_next_slot[peer] < 1) continue;
        uint32_t prev_slot = r->try_next_slot[peer] - 1;
        if (prev_slot < r->start_slot) return;            /* logged_err */
        if (prev_slot >= log_end(r)) continue;
        if (slot >= r->try_next_slot[peer]) {
            if (r->ae_first[peer] == NONE32) r->ae_first[peer] = r->try_next_slot[peer];   /* :44-52 entries from here */
            r->n_sent += slot + 1 - r->try_next_slot[peer];
            r->try_next_slot[peer] = slot + 1;            /* :85 */
        }
    }
}

/* request.rs:10-91; W-bounded ring back-pressure is the harness guard */
static void handle_req_batch(RaftRep *r, uint32_t W) {
    if (r->role != ROLE_LEADER) { r->n_redirect++; return; }   /* :19-42 */
    if (log_end(r) - r->last_snap >= W) { r->n_reject++; return; }
    uint32_t slot = log_end(r);                            /* :77 */
    log_push(r, r->curr_term);
    handle_logged_leader_append(r, slot);                  /* WAL completes at once */
}

static uint64_t term_at(const RaftRep *r, uint32_t slot, uint32_t W, int *ok);

/* ae_first (may be NULL) [R][G]: the first slot of the entries sent to each peer by this call's appends
 * (durability.rs:44-88), NONE32 if nothing was sent: all AppendEntries of the call to one peer, taken
 * together, carry the slots [ae_first, log end) */
void orc_raft_leader_append_emit(void *h, const uint32_t *n_new, uint32_t *ae_first) {
    RaftCl *cl = (RaftCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        RaftRep *r = &cl->reps[g];
        for (int p = 0; p < MAXR; p++) r->ae_first[p] = NONE32;
        for (uint32_t k = 0; k < n_new[g]; k++) handle_req_batch(r, cl->W);
        if (ae_first) for (int p = 0; p < cl->R; p++) ae_first[(size_t)p * G + g] = r->ae_first[p];
    }
}
void orc_raft_leader_append(void *h, const uint32_t *n_new) { orc_raft_leader_append_emit(h, n_new, NULL); }

/* The AppendEntries a leader's appends produced for one peer, as ONE message per group (the reference sends
 * one per appended batch, durability.rs:57-80; a follower that handles them in order ends in the same
 * state): entries [first, min(first + K, log end)), prev = first - 1. */
void orc_raft_gather_entries(void *h, const uint32_t *first, uint32_t K, uint8_t *flags, uint8_t *leader, uint64_t *term,
                             uint32_t *prev_slot, uint64_t *prev_term, uint32_t *n_entries, uint64_t *entry_term,
                             uint32_t *leader_commit, uint32_t *last_snap) {
    RaftCl *cl = (RaftCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        RaftRep *r = &cl->reps[g];
        flags[g] = 0; leader[g] = r->id; term[g] = r->curr_term; prev_slot[g] = 0; prev_term[g] = 0; n_entries[g] = 0;
        leader_commit[g] = r->last_commit; last_snap[g] = r->last_snap;
        for (uint32_t k = 0; k < K; k++) entry_term[(size_t)k * G + g] = 0;
        if (first[g] == NONE32 || r->role != ROLE_LEADER || first[g] < 1 || first[g] > log_end(r)) continue;
        int ok; uint64_t pt = term_at(r, first[g] - 1, cl->W, &ok);
        if (!ok) continue;                                  /* prev fell out of the ring: nothing to send (harness) */
        uint32_t n = log_end(r) - first[g];
        if (n > K) n = K;
        flags[g] = 1; prev_slot[g] = first[g] - 1; prev_term[g] = pt; n_entries[g] = n;
        for (uint32_t k = 0; k < n; k++) {
            int ok2; entry_term[(size_t)k * G + g] = term_at(r, first[g] + k, cl->W, &ok2);
        }
    }
}

/* leadership.rs:16-72; returns 1 iff the role was not Follower and now is */
static int check_term(RaftRep *r, uint8_t peer, uint64_t term) {
    if (term > r->curr_term) {
        r->curr_term = term;
        r->voted_for = NO_LEADER;                         /* :21-22 */
        r->votes = 0;
        r->leader = peer;
        if (r->role == ROLE_FOLLOWER) return 0;
        r->role = ROLE_FOLLOWER;
        return 1;
    }
    return 0;
}

static uint64_t term_at(const RaftRep *r, uint32_t slot, uint32_t W, int *ok) {
    /* ring guard shared with the engine: only the last W entries are readable */
    (void)W;
    if (slot < r->start_slot || slot >= log_end(r) || slot < r->ring_lo) { *ok = 0; return 0; }
    *ok = 1;
    return r->log_term[slot - r->start_slot];
}

/* heartbeat.rs:280-296 */
static void update_heard_cnt(RaftRep *r, uint8_t peer) {
    r->hb_replied[peer] += 1;
    if (!((r->peer_alive >> peer) & 1)) r->peer_alive |= (uint8_t)(1u << peer);
}

/* messages.rs:222-388; CRaft: craft/messages.rs:256-404 */
static void handle_msg_append_entries_reply(RaftRep *r, uint32_t W, uint8_t peer, uint64_t term, uint32_t end_slot,
                                            int has_conflict, uint64_t conflict_term, uint32_t conflict_slot) {
    if (check_term(r, peer, term) || r->role != ROLE_LEADER) return;   /* :239-241 */
    if (r->craft) update_heard_cnt(r, peer);               /* craft/messages.rs:275 heard_heartbeat -> leadership.rs:300-303 */
    if (!has_conflict) {
        /* raft :245-247; the fork only debug_asserts it (craft/messages.rs:279): a release build goes on */
        if (!r->craft && r->next_slot[peer] > end_slot + 1) return;
        r->next_slot[peer] = end_slot + 1;
        if (r->try_next_slot[peer] < end_slot + 1) r->try_next_slot[peer] = end_slot + 1;
        r->match_slot[peer] = end_slot;
        uint32_t new_commit = r->last_commit;              /* :256-275 */
        for (uint32_t slot = r->last_commit + 1; slot < log_end(r); slot++) {
            int ok; uint64_t t = term_at(r, slot, W, &ok);
            if (!ok || t != r->curr_term) continue;
            int match_cnt = 1;
            for (int q = 0; q < r->population; q++)
                if (q != r->id && r->match_slot[q] >= slot) match_cnt++;
            if (!r->craft) {
                if (match_cnt >= r->commit_thresh) new_commit = slot;
            } else if ((!r->full_copy_mode && match_cnt >= r->quorum_cnt + r->fault_tolerance) ||
                       (r->full_copy_mode && match_cnt >= r->quorum_cnt)) {   /* craft/messages.rs:307-313 */
                new_commit = slot;
            }
        }
        if (!r->craft) {
            r->n_committed += new_commit - r->last_commit; /* :278-293 exec submission */
            r->last_commit = new_commit;                   /* :295 */
        } else {                                           /* craft/messages.rs:315-358 */
            const uint8_t data = (uint8_t)((1u << r->quorum_cnt) - 1u);
            int can_execute = 1;
            for (uint32_t slot = r->last_commit + 1; slot <= new_commit; slot++) {
                if (slot < r->ring_lo) break;              /* harness guard shared with the engine */
                uint8_t *m = &r->log_mask[slot - r->start_slot];
                if (__builtin_popcount(*m) < r->quorum_cnt) {            /* :318-325 ask the peers for its shards, once */
                    if (slot > r->last_recon) {
                        if (r->rq_n < 16) { r->rq_slot[r->rq_n] = slot; r->rq_term[r->rq_n] = r->log_term[slot - r->start_slot]; r->rq_n++; }
                        r->last_recon = slot;
                    }
                    can_execute = 0;
                    continue;
                } else if (__builtin_popcount(*m & data) < r->quorum_cnt) { *m |= data; r->n_recon_data++; }   /* :326-328 */
                if (can_execute) { r->n_committed++; r->last_commit = slot; }                              /* :329-345 */
            }
        }
        for (uint32_t slot = r->last_snap + 1; slot <= end_slot; slot++) {   /* :298-309 */
            int match_cnt = 1;
            for (int q = 0; q < r->population; q++)
                if (q != r->id && r->match_slot[q] >= slot) match_cnt++;
            if (match_cnt == r->population) r->last_snap = slot;
        }
    } else {
        if (r->next_slot[peer] == 1) { r->try_next_slot[peer] = 1; return; }   /* :313-316 */
        r->next_slot[peer] -= 1;                           /* :318 */
        for (;;) {                                         /* :320-330 */
            uint32_t ns = r->next_slot[peer];
            int ok; uint64_t t = term_at(r, ns, W, &ok);
            if (!(ns > r->start_slot && ok && t == conflict_term && ns >= conflict_slot && ns > 1)) break;
            r->next_slot[peer] -= 1;
        }
        r->try_next_slot[peer] = r->next_slot[peer];       /* :331 */
        uint32_t prev_slot = r->next_slot[peer] - 1;
        if (prev_slot < r->start_slot) return;             /* :335-337 */
        if (prev_slot >= log_end(r)) return;               /* :338-340 */
        if (end_slot + 1 > r->next_slot[peer]) r->n_sent += end_slot + 1 - r->next_slot[peer];
        r->try_next_slot[peer] = end_slot + 1;             /* :384 */
    }
}

static uint32_t ctl_order(uint32_t ctl, int i) { return (ctl >> (3 * i)) & 7u; }
#define CTL_IDENTITY 0x00FAC688u

/* One reply per (peer, group): arrays [R][G]; flags bit0 valid, bit1 conflict */
void orc_raft_handle_replies(void *h, const uint64_t *reply_term, const uint32_t *end_slot,
                             const uint64_t *conflict_term, const uint32_t *conflict_slot, const uint8_t *flags,
                             const uint32_t *order) {
    RaftCl *cl = (RaftCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        RaftRep *r = &cl->reps[g];
        uint32_t ctl = order ? order[g] : CTL_IDENTITY;
        for (int oi = 0; oi < cl->R; oi++) {
            int p = (int)ctl_order(ctl, oi);
            if (p == r->id || p >= cl->R) continue;
            size_t o = (size_t)p * G + g;
            if (!(flags[o] & 1)) continue;
            handle_msg_append_entries_reply(r, cl->W, (uint8_t)p, reply_term[o], end_slot[o], (flags[o] >> 1) & 1,
                                            conflict_term ? conflict_term[o] : 0,
                                            conflict_slot ? conflict_slot[o] : 0);
        }
    }
}

/* ---- follower side and elections ---------------------------------------- */

/* test set-up: put every group's replica into a given role / term / leader / vote */
void orc_raft_preset(void *h, uint8_t role, uint8_t leader, uint64_t term, uint8_t voted_for) {
    RaftCl *cl = (RaftCl *)h;
    for (uint32_t g = 0; g < cl->G; g++) {
        RaftRep *r = &cl->reps[g];
        r->role = role; r->leader = leader; r->curr_term = term; r->voted_for = voted_for; r->votes = 0;
    }
}

/* messages.rs:13-218, durability.rs:97-132.  Reply: flags bit0 = a reply is sent, bit1 = conflict */
static void handle_msg_append_entries(RaftRep *r, uint32_t W, uint8_t leader, uint64_t term, uint32_t prev_slot,
                                      uint64_t prev_term, uint32_t n, const uint64_t *ent, size_t ent_stride,
                                      uint32_t leader_commit, uint32_t last_snap, uint8_t *r_flags, uint64_t *r_term,
                                      uint32_t *r_end, uint64_t *r_cterm, uint32_t *r_cslot) {
    *r_flags = 0; *r_term = 0; *r_end = 0; *r_cterm = 0; *r_cslot = 0;
    if (check_term(r, leader, term) || r->role != ROLE_FOLLOWER) {            /* :32 */
        if (term == r->curr_term && r->role == ROLE_CANDIDATE) {              /* :33-39 */
            r->curr_term -= 1;
            check_term(r, leader, term);
        } else return;
    }
    int ok; uint64_t t_prev = term_at(r, prev_slot, W, &ok);
    if (n != 0 && (term < r->curr_term || prev_slot < r->start_slot || prev_slot >= log_end(r) || !ok ||
                   t_prev != prev_term)) {                                    /* :46-51 (!ok: ring guard) */
        uint64_t conflict_term = (prev_slot >= r->start_slot && prev_slot < log_end(r) && ok) ? t_prev : 0;
        uint32_t conflict_slot = prev_slot;
        while (conflict_term > 0 && conflict_slot > r->start_slot) {          /* :60-68 */
            int ok2; uint64_t t = term_at(r, conflict_slot - 1, W, &ok2);
            if (ok2 && t == conflict_term) conflict_slot--; else break;
        }
        *r_flags = 3; *r_term = r->curr_term; *r_end = prev_slot + n;         /* :70-77 */
        *r_cterm = conflict_term; *r_cslot = conflict_slot;
        return;
    }
    r->leader = leader;                                                       /* :94 */
    uint32_t first_new = prev_slot + 1;                                       /* :99-139 */
    for (uint32_t s = 0; s < n; s++) {
        uint32_t slot = prev_slot + 1 + s;
        if (slot >= log_end(r)) { first_new = slot; break; }
        if (slot >= r->start_slot && slot < r->ring_lo) { r->n_ring_guard++; continue; }   /* harness guard shared with the engine: an entry that left the
                                                                               W-entry term ring is taken as matching, never as a conflict */
        int ok3; uint64_t t = term_at(r, slot, W, &ok3);
        if (!ok3 || t != ent[s * ent_stride]) {
            r->n_log = slot - r->start_slot;                                  /* :136 truncate */
            r->n_trunc++;
            first_new = slot;
            break;
        }
    }
    /* :143-167 entries.drain(first_new - prev_slot - 1 ..): every drained entry is PUSHED -- also
     * when the loop above never broke (all entries already present): they are appended again */
    uint32_t skipped = first_new - prev_slot - 1, num_appended = 0;
    uint32_t slot_e = prev_slot + n;
    for (uint32_t s = skipped; s < n; s++) {
        uint32_t slot = (s - skipped) + first_new;
        log_push(r, ent[s * ent_stride]);
        num_appended++;
        /* WAL completion, durability.rs:97-132 */
        if (!(slot < r->start_slot || r->role != ROLE_FOLLOWER) && slot == slot_e && r->leader != NO_LEADER) {
            *r_flags = 1; *r_term = r->curr_term; *r_end = slot_e;
        }
    }
    if (num_appended == 0) { *r_flags = 1; *r_term = r->curr_term; *r_end = first_new - 1; }   /* :172-181 */
    if (leader_commit > r->last_commit) {                                     /* :184-208; entries.len() is now `skipped` */
        uint32_t new_commit = leader_commit < prev_slot + skipped ? leader_commit : prev_slot + skipped;
        if (new_commit > log_end(r) - 1) new_commit = log_end(r) - 1;
        if (new_commit > r->last_commit) r->n_exec += new_commit - r->last_commit;
        r->last_commit = new_commit;
    }
    if (last_snap > r->last_snap) r->last_snap = last_snap;                   /* :211-213 */
}

void orc_raft_handle_append_entries(void *h, const uint8_t *flags, const uint8_t *leader, const uint64_t *term,
                                    const uint32_t *prev_slot, const uint64_t *prev_term, const uint32_t *n_entries,
                                    const uint64_t *entry_term, uint32_t K, const uint32_t *leader_commit,
                                    const uint32_t *last_snap, uint8_t *r_flags, uint64_t *r_term, uint32_t *r_end,
                                    uint64_t *r_cterm, uint32_t *r_cslot) {
    RaftCl *cl = (RaftCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        r_flags[g] = 0; r_term[g] = 0; r_end[g] = 0; r_cterm[g] = 0; r_cslot[g] = 0;
        if (!(flags[g] & 1)) continue;
        uint32_t n = n_entries[g] < K ? n_entries[g] : K;
        handle_msg_append_entries(&cl->reps[g], cl->W, leader[g], term[g], prev_slot[g], prev_term[g], n,
                                  entry_term + g, G, leader_commit[g], last_snap[g], &r_flags[g], &r_term[g],
                                  &r_end[g], &r_cterm[g], &r_cslot[g]);
    }
}

/* craft/messages.rs:14-254 + craft/durability.rs:129-165.  emask[s]: avail_shards_map of the s-th entry's codeword.
 * Reply: flags bit0 = a reply is sent, bit1 = conflict */
static void craft_handle_msg_append_entries(RaftRep *r, uint32_t W, uint8_t leader, uint64_t term, uint32_t prev_slot,
                                            uint64_t prev_term, uint32_t n, const uint64_t *ent, const uint8_t *emask,
                                            size_t ent_stride, uint32_t leader_commit, uint32_t last_snap, uint8_t *r_flags,
                                            uint64_t *r_term, uint32_t *r_end, uint64_t *r_cterm, uint32_t *r_cslot) {
    *r_flags = 0; *r_term = 0; *r_end = 0; *r_cterm = 0; *r_cslot = 0;
    if (check_term(r, leader, term) || r->role != ROLE_FOLLOWER) {            /* :33-40 */
        if (term == r->curr_term && r->role == ROLE_CANDIDATE) {
            r->curr_term -= 1;
            check_term(r, leader, term);
        } else return;
    }
    int ok; uint64_t t_prev = term_at(r, prev_slot, W, &ok);
    if (term < r->curr_term || prev_slot < r->start_slot || prev_slot >= log_end(r) || !ok || t_prev != prev_term) {   /* :43-47, heartbeats too */
        uint64_t conflict_term = (prev_slot >= r->start_slot && prev_slot < log_end(r) && ok) ? t_prev : 0;
        uint32_t conflict_slot = prev_slot;
        while (conflict_term > 0 && conflict_slot > r->start_slot) {          /* :56-64 */
            int ok2; uint64_t t = term_at(r, conflict_slot - 1, W, &ok2);
            if (ok2 && t == conflict_term) conflict_slot--; else break;
        }
        *r_flags = 3; *r_term = r->curr_term; *r_end = prev_slot + n;         /* :66-73 */
        *r_cterm = conflict_term; *r_cslot = conflict_slot;
        if (term >= r->curr_term) r->leader = leader;                         /* :81-84 (+ heard_heartbeat: a timer) */
        return;
    }
    r->leader = leader;                                                       /* :89-90 */
    const uint8_t data = (uint8_t)((1u << r->quorum_cnt) - 1u);
    uint32_t first_new = prev_slot + 1;                                       /* :93-147 */
    for (uint32_t s = 0; s < n; s++) {
        uint32_t slot = prev_slot + 1 + s;
        if (slot >= log_end(r)) { first_new = slot; break; }
        if (slot >= r->start_slot && slot < r->ring_lo) { r->n_ring_guard++; continue; }   /* harness guard shared with the engine: an entry that left the
                                                                               W-entry term ring is taken as matching, never as a conflict */
        int ok3; uint64_t t = term_at(r, slot, W, &ok3);
        if (!ok3 || t != ent[s * ent_stride]) {
            r->n_log = slot - r->start_slot;                                  /* :129 truncate */
            r->n_trunc++;
            first_new = slot;
            break;
        }
        uint8_t *m = &r->log_mask[slot - r->start_slot];                      /* :133-146 no conflict: absorb the sent shards */
        const uint8_t em = emask[s * ent_stride];
        if (__builtin_popcount(*m & data) < r->quorum_cnt && *m != em) *m |= em;
    }
    uint32_t skipped = first_new - prev_slot - 1, num_appended = 0;           /* :149-173 */
    uint32_t slot_e = prev_slot + n;
    for (uint32_t s = skipped; s < n; s++) {
        uint32_t slot = (s - skipped) + first_new;
        log_push_m(r, ent[s * ent_stride], emask[s * ent_stride]);
        num_appended++;
        if (!(slot < r->start_slot || r->role != ROLE_FOLLOWER) && slot == slot_e && r->leader != NO_LEADER) {   /* durability.rs:129-165 */
            *r_flags = 1; *r_term = r->curr_term; *r_end = slot_e;
        }
    }
    if (num_appended == 0) { *r_flags = 1; *r_term = r->curr_term; *r_end = first_new - 1; }   /* :176-185 */
    if (leader_commit > r->last_commit) {                                     /* :188-237; entries.len() is now `skipped` */
        uint32_t new_commit = leader_commit < prev_slot + skipped ? leader_commit : prev_slot + skipped;
        if (new_commit > log_end(r) - 1) new_commit = log_end(r) - 1;
        for (uint32_t slot = r->last_commit + 1; slot <= new_commit; slot++) {
            if (slot < r->ring_lo) break;                                     /* harness guard shared with the engine: left the ring */
            uint8_t *m = &r->log_mask[slot - r->start_slot];
            if (__builtin_popcount(*m) < r->quorum_cnt) { r->n_postponed++; break; }      /* :197-208 not enough shards yet */
            else if (__builtin_popcount(*m & data) < r->quorum_cnt) { *m |= data; r->n_recon_data++; }   /* :209-212 reconstruct_data */
            r->n_exec++;                                                      /* :213-229 */
            r->last_commit = slot;                                            /* :233 */
        }
    }
    if (last_snap > r->last_snap) r->last_snap = last_snap;                   /* :240-242 */
}

void orc_craft_handle_append_entries(void *h, const uint8_t *flags, const uint8_t *leader, const uint64_t *term,
                                     const uint32_t *prev_slot, const uint64_t *prev_term, const uint32_t *n_entries,
                                     const uint64_t *entry_term, const uint8_t *entry_mask, uint32_t K, const uint32_t *leader_commit,
                                     const uint32_t *last_snap, uint8_t *r_flags, uint64_t *r_term, uint32_t *r_end,
                                     uint64_t *r_cterm, uint32_t *r_cslot) {
    RaftCl *cl = (RaftCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        r_flags[g] = 0; r_term[g] = 0; r_end[g] = 0; r_cterm[g] = 0; r_cslot[g] = 0;
        if (!(flags[g] & 1)) continue;
        uint32_t n = n_entries[g] < K ? n_entries[g] : K;
        craft_handle_msg_append_entries(&cl->reps[g], cl->W, leader[g], term[g], prev_slot[g], prev_term[g], n,
                                        entry_term + g, entry_mask + g, G, leader_commit[g], last_snap[g], &r_flags[g], &r_term[g],
                                        &r_end[g], &r_cterm[g], &r_cslot[g]);
    }
}

/* craft/messages.rs:622-663: Reconstruct { slots } from `peer`: n[g] (slot, term) pairs [K][G]; the ReconstructReply holds the
 * codeword of every slot I have under that term: r_has[K][G] (1 = in the reply), r_mask[K][G]; r_n[g] = how many (0: no reply) */
void orc_craft_handle_reconstruct(void *h, const uint32_t *n, const uint32_t *slot, const uint64_t *term, uint32_t K, uint32_t *r_n,
                                  uint8_t *r_has, uint8_t *r_mask) {
    RaftCl *cl = (RaftCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        RaftRep *r = &cl->reps[g];
        r_n[g] = 0;
        for (uint32_t k = 0; k < K; k++) {
            const size_t o = (size_t)k * G + g;
            r_has[o] = 0; r_mask[o] = 0;
            if (k >= n[g]) continue;
            int ok; uint64_t t = term_at(r, slot[o], cl->W, &ok);
            if (slot[o] < r->start_slot || slot[o] >= log_end(r) || !ok || t != term[o]) continue;   /* :631-636 */
            r_has[o] = 1; r_mask[o] = r->log_mask[slot[o] - r->start_slot]; r_n[g]++;
        }
    }
}

/* the Reconstruct { slots } broadcasts the reply handler queued since the last take (craft/messages.rs:347-358): n[g], slot /
 * term [K][G] */
void orc_craft_take_reconstructs(void *h, uint32_t K, uint32_t *n, uint32_t *slot, uint64_t *term) {
    RaftCl *cl = (RaftCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        RaftRep *r = &cl->reps[g];
        n[g] = r->rq_n < K ? r->rq_n : K;
        for (uint32_t k = 0; k < K; k++) {
            slot[(size_t)k * G + g] = k < n[g] ? r->rq_slot[k] : 0;
            term[(size_t)k * G + g] = k < n[g] ? r->rq_term[k] : 0;
        }
        r->rq_n = 0;
    }
}

/* craft/messages.rs:665-745: ReconstructReply { slots_data } from peer[g] (NO_LEADER: none): n[g] (slot, bitmap) pairs [K][G] */
void orc_craft_handle_reconstruct_reply(void *h, const uint8_t *peer, const uint32_t *n, const uint32_t *slot, const uint8_t *mask,
                                        uint32_t K) {
    RaftCl *cl = (RaftCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        RaftRep *r = &cl->reps[g];
        if (peer[g] == NO_LEADER || peer[g] >= r->population) continue;
        if (peer[g] != r->id && r->craft) update_heard_cnt(r, peer[g]);       /* :669 heard_heartbeat -> leadership.rs:300-303 */
        uint32_t ms[MAXR]; int nm = 0;                                        /* :670-683 shadow_last_commit */
        for (int q = 0; q < r->population; q++) if (q != r->id) ms[nm++] = r->match_slot[q];
        for (int a = 0; a < nm; a++) for (int b = a + 1; b < nm; b++) if (ms[b] > ms[a]) { uint32_t t = ms[a]; ms[a] = ms[b]; ms[b] = t; }
        const int idx = r->full_copy_mode ? r->quorum_cnt - 2 : r->quorum_cnt + r->fault_tolerance - 2;
        const uint32_t shadow = ms[idx];
        const uint8_t data = (uint8_t)((1u << r->quorum_cnt) - 1u);
        for (uint32_t k = 0; k < K && k < n[g]; k++) {
            const uint32_t sl = slot[(size_t)k * G + g];
            if (sl < r->start_slot || sl >= log_end(r) || sl < r->ring_lo) continue;   /* :685-687 (+ the debug_assert, the ring guard) */
            r->log_mask[sl - r->start_slot] |= mask[(size_t)k * G + g];                 /* :697 absorb_other */
            if (sl == r->last_commit + 1) {                                   /* :699-737 */
                while (r->last_commit < shadow) {
                    const uint32_t nx = r->last_commit + 1;
                    if (nx >= log_end(r) || nx < r->ring_lo) break;           /* (harness guards) */
                    uint8_t *m = &r->log_mask[nx - r->start_slot];
                    if (__builtin_popcount(*m) < r->quorum_cnt) break;
                    if (__builtin_popcount(*m & data) < r->quorum_cnt) { *m |= data; r->n_recon_data++; }
                    r->n_committed++;
                    r->last_commit += 1;
                }
            }
        }
    }
}

/* CRaft follower state: the entries' shard bitmaps [W][G] by slot % W (slots outside the log or the ring: 0), counters[2] =
 * reconstruct_data calls, postponed executions */
void orc_craft_dump_masks(void *h, uint8_t *mask, uint64_t *counters) {
    RaftCl *cl = (RaftCl *)h;
    const uint32_t G = cl->G, W = cl->W;
    counters[0] = counters[1] = 0;
    for (uint32_t g = 0; g < G; g++) {
        RaftRep *r = &cl->reps[g];
        counters[0] += r->n_recon_data; counters[1] += r->n_postponed;
        for (uint32_t w = 0; w < W; w++) mask[(size_t)w * G + g] = 0;
        uint32_t end = log_end(r), lo = end > W ? end - W : r->start_slot;
        if (lo < r->ring_lo) lo = r->ring_lo;
        for (uint32_t s = lo; s < end; s++) mask[(size_t)(s % W) * G + g] = r->log_mask[s - r->start_slot];
    }
}

/* leadership.rs:76-142; rv_flags bit0 = RequestVote broadcast */
void orc_raft_become_candidate(void *h, const uint8_t *timeout_src, uint8_t *rv_flags, uint64_t *rv_term,
                               uint32_t *rv_last_slot, uint64_t *rv_last_term) {
    RaftCl *cl = (RaftCl *)h;
    for (uint32_t g = 0; g < cl->G; g++) {
        RaftRep *r = &cl->reps[g];
        rv_flags[g] = 0; rv_term[g] = 0; rv_last_slot[g] = 0; rv_last_term[g] = 0;
        if (timeout_src[g] == NO_LEADER) continue;
        if (r->role != ROLE_FOLLOWER || (r->leader != NO_LEADER && r->leader != timeout_src[g])) continue;   /* :80-85 */
        r->role = ROLE_CANDIDATE;
        r->curr_term += 1;                                                    /* :90-92 */
        r->voted_for = r->id;
        r->votes = (uint8_t)(1u << r->id);
        uint32_t last_slot = log_end(r) - 1;                                  /* :99-101 */
        int ok; uint64_t lt = term_at(r, last_slot, cl->W, &ok);
        rv_flags[g] = 1; rv_term[g] = r->curr_term; rv_last_slot[g] = last_slot; rv_last_term[g] = ok ? lt : 0;
    }
}

/* messages.rs:391-482; r_flags bit0 = a reply is sent, bit1 = granted */
void orc_raft_handle_request_vote(void *h, const uint8_t *flags, const uint8_t *cand, const uint64_t *term,
                                  const uint32_t *last_slot, const uint64_t *last_term, uint8_t *r_flags,
                                  uint64_t *r_term) {
    RaftCl *cl = (RaftCl *)h;
    for (uint32_t g = 0; g < cl->G; g++) {
        RaftRep *r = &cl->reps[g];
        r_flags[g] = 0; r_term[g] = 0;
        if (!(flags[g] & 1)) continue;
        check_term(r, cand[g], term[g]);                                      /* :405 */
        if (term[g] < r->curr_term) { r_flags[g] = 1; r_term[g] = r->curr_term; continue; }   /* :408-422 */
        if (r->voted_for == NO_LEADER || r->voted_for == cand[g]) {           /* :427 */
            int ok; uint64_t my_last = term_at(r, log_end(r) - 1, cl->W, &ok);
            if (last_term[g] >= my_last || (last_term[g] == r->curr_term && last_slot[g] + 1 >= log_end(r))) {   /* :428-430 */
                r_flags[g] = 3; r_term[g] = r->curr_term;
                r->voted_for = cand[g];                                       /* :450 */
            }
        }
    }
}

/* messages.rs:485-510 + leadership.rs:145-218.  hb_prev[p][g] = prev_slot of the heartbeat
 * computed for peer p when the replica gets elected in this call (0xFFFFFFFF otherwise); the
 * reference broadcasts EACH of them to ALL peers (bcast_msg inside the per-peer loop). */
void orc_raft_handle_vote_replies(void *h, const uint64_t *term, const uint8_t *granted, const uint8_t *flags,
                                  const uint32_t *order, uint32_t *hb_prev, uint8_t *elected) {
    RaftCl *cl = (RaftCl *)h;
    const uint32_t G = cl->G;
    (void)granted;                                                            /* :503 inserts the peer whatever it answered */
    for (uint32_t g = 0; g < G; g++) {
        RaftRep *r = &cl->reps[g];
        elected[g] = 0;
        for (int p = 0; p < cl->R; p++) hb_prev[(size_t)p * G + g] = 0xFFFFFFFFu;
        uint32_t ctl = order ? order[g] : CTL_IDENTITY;
        for (int oi = 0; oi < cl->R; oi++) {
            int p = (int)ctl_order(ctl, oi);
            if (p == r->id || p >= cl->R) continue;
            size_t o = (size_t)p * G + g;
            if (!(flags[o] & 1)) continue;
            if (check_term(r, (uint8_t)p, term[o]) || r->role != ROLE_CANDIDATE) continue;   /* :498-500 */
            r->votes |= (uint8_t)(1u << p);                                   /* :503 */
            if (__builtin_popcount(r->votes) >= r->quorum_cnt) {              /* :506-508 */
                r->role = ROLE_LEADER;                                        /* leadership.rs:149 */
                for (int q = 0; q < cl->R; q++) {                             /* :156 bcast_heartbeats, :186-196 */
                    if (q == r->id) continue;
                    uint32_t a = r->try_next_slot[q] - 1, b = log_end(r) - 1;
                    hb_prev[(size_t)q * G + g] = a < b ? a : b;
                }
                for (int q = 0; q < cl->R; q++) {                             /* :159-168 */
                    r->next_slot[q] = log_end(r); r->try_next_slot[q] = log_end(r); r->match_slot[q] = 0;
                }
                elected[g] = 1;
            }
        }
    }
}

void orc_raft_dump_votes(void *h, uint8_t *voted_for, uint8_t *votes, uint64_t *n_exec, uint64_t *n_trunc) {
    RaftCl *cl = (RaftCl *)h;
    for (uint32_t g = 0; g < cl->G; g++) {
        voted_for[g] = cl->reps[g].voted_for; votes[g] = cl->reps[g].votes;
        if (n_exec) n_exec[g] = cl->reps[g].n_exec;
        if (n_trunc) n_trunc[g] = cl->reps[g].n_trunc;
    }
}

void orc_raft_dump(void *h, uint8_t *role, uint64_t *curr_term, uint32_t *log_len, uint32_t *last_commit,
                   uint32_t *last_snap, uint32_t *next_slot, uint32_t *try_next_slot, uint32_t *match_slot,
                   uint64_t *entry_term, uint8_t *leader, uint32_t *start_slot) {
    RaftCl *cl = (RaftCl *)h;
    const uint32_t G = cl->G, W = cl->W;
    for (uint32_t g = 0; g < G; g++) {
        RaftRep *r = &cl->reps[g];
        role[g] = r->role; curr_term[g] = r->curr_term; log_len[g] = log_end(r);
        last_commit[g] = r->last_commit; last_snap[g] = r->last_snap;
        leader[g] = r->leader; start_slot[g] = r->start_slot;
        for (int p = 0; p < cl->R; p++) {
            size_t o = (size_t)p * G + g;
            next_slot[o] = p == r->id ? 0 : r->next_slot[p];
            try_next_slot[o] = p == r->id ? 0 : r->try_next_slot[p];
            match_slot[o] = p == r->id ? 0 : r->match_slot[p];
        }
        for (uint32_t w = 0; w < W; w++) entry_term[(size_t)w * G + g] = 0;
        uint32_t lo = r->ring_lo > r->start_slot ? r->ring_lo : r->start_slot;
        for (uint32_t s = lo; s < log_end(r); s++) entry_term[(size_t)(s % W) * G + g] = r->log_term[s - r->start_slot];
    }
}

uint64_t orc_raft_total_commits(void *h) {
    RaftCl *cl = (RaftCl *)h;
    uint64_t t = 0;
    for (uint32_t g = 0; g < cl->G; g++) t += cl->reps[g].n_committed;
    return t;
}

/* How often the follower's prev-term check met an entry that had left the W-entry term ring and took it as matching -- where
 * raft/messages.rs:128-140 compares terms.  Engine and oracle share the deviation (a ring cannot hold what the reference's
 * Vec holds), so a parity run in which this is not 0 proves nothing about those entries: the tests assert it. */
uint64_t orc_raft_ring_guard_hits(void *h) {
    RaftCl *cl = (RaftCl *)h;
    uint64_t t = 0;
    for (uint32_t g = 0; g < cl->G; g++) t += cl->reps[g].n_ring_guard;
    return t;
}

void orc_raft_counters(void *h, uint64_t out[4]) {
    RaftCl *cl = (RaftCl *)h;
    memset(out, 0, sizeof(uint64_t) * 4);
    for (uint32_t g = 0; g < cl->G; g++) {
        out[0] += cl->reps[g].n_committed; out[1] += cl->reps[g].n_redirect;
        out[2] += cl->reps[g].n_reject; out[3] += cl->reps[g].n_sent;
    }
}

/* ---- CRaft leader variant ------------------------------------------------ */

/* turn every group's leader into a CRaft leader: craft/mod.rs:573 (full_copy_mode false), heartbeat.rs:117-131 */
void orc_craft_enable(void *h, uint8_t fault_tolerance, uint8_t repeat_threshold) {
    RaftCl *cl = (RaftCl *)h;
    for (uint32_t g = 0; g < cl->G; g++) {
        RaftRep *r = &cl->reps[g];
        r->craft = 1; r->fault_tolerance = fault_tolerance; r->repeat_threshold = repeat_threshold;
        r->full_copy_mode = 0;
        r->peer_alive = (uint8_t)((1u << cl->R) - 1u);
        for (int p = 0; p < cl->R; p++) {
            r->hb_replied[p] = p == r->id ? 0 : 1; r->hb_seen[p] = 0; r->hb_repeat[p] = 0;
        }
    }
}

/* craft/leadership.rs:80-141 (state effect; the re-send it once did is commented out there) */
static void switch_assignment_mode(RaftRep *r, int to_full_copy) {
    if (r->full_copy_mode == (uint8_t)to_full_copy) return;
    r->full_copy_mode = (uint8_t)to_full_copy;
}

/* to_full[g]: 0 / 1, anything else = no call for the group */
void orc_craft_switch_assignment_mode(void *h, const uint8_t *to_full) {
    RaftCl *cl = (RaftCl *)h;
    for (uint32_t g = 0; g < cl->G; g++)
        if (to_full[g] <= 1) switch_assignment_mode(&cl->reps[g], to_full[g]);
}

/* heartbeat.rs:240-276 */
static int update_bcast_cnts(RaftRep *r) {
    int peer_death = 0;
    for (int peer = 0; peer < r->population; peer++) {
        if (peer == r->id) continue;
        if (r->hb_replied[peer] > r->hb_seen[peer]) {
            r->hb_seen[peer] = r->hb_replied[peer];
            r->hb_repeat[peer] = 0;
        } else {
            r->hb_repeat[peer] += 1;
            if (r->hb_repeat[peer] > r->repeat_threshold) {
                if ((r->peer_alive >> peer) & 1) {
                    r->peer_alive &= (uint8_t)~(1u << peer);
                    peer_death = 1;
                }
                r->hb_repeat[peer] = 0;
            }
        }
    }
    return peer_death;
}

/* craft/leadership.rs:249-291, on the send tick (only a leader's Heartbeater ticks: leadership.rs:65,217).
 * hb_flags[R][G]: 1 = a heartbeat goes to that peer (prev_slot, prev_term [R][G]; leader_commit, last_snap [G]) */
void orc_craft_bcast_heartbeats(void *h, uint8_t *hb_flags, uint32_t *prev_slot, uint64_t *prev_term,
                                uint32_t *leader_commit, uint32_t *last_snap) {
    RaftCl *cl = (RaftCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        RaftRep *r = &cl->reps[g];
        leader_commit[g] = r->last_commit; last_snap[g] = r->last_snap;
        for (int p = 0; p < cl->R; p++) {
            size_t o = (size_t)p * G + g;
            hb_flags[o] = 0; prev_slot[o] = 0; prev_term[o] = 0;
        }
        if (r->role != ROLE_LEADER) continue;
        for (int peer = 0; peer < r->population; peer++) {       /* :252-273 */
            if (peer == r->id) continue;
            uint32_t ps = r->try_next_slot[peer] - 1;
            if (ps > log_end(r) - 1) ps = log_end(r) - 1;
            int ok; uint64_t pt = term_at(r, ps, cl->W, &ok);
            if (!ok) continue;                                   /* out of the term ring (harness) */
            size_t o = (size_t)peer * G + g;
            hb_flags[o] = 1; prev_slot[o] = ps; prev_term[o] = pt;
        }
        (void)update_bcast_cnts(r);                              /* :277 */
        /* :280 heard_heartbeat(self.id): nothing for peer == id (:300) */
        int alive = 0;
        for (int p = 0; p < r->population; p++) alive += (r->peer_alive >> p) & 1;
        if (!r->full_copy_mode && r->population - alive >= r->fault_tolerance) switch_assignment_mode(r, 1);   /* :283-288 */
    }
}

/* Which shards of a new entry's codeword go where (bitmask over shard indices):
 * persist[g]: what the leader's WAL entry holds (craft/request.rs:86-100): full copy = the data shards 0..majority,
 * else its own shard; send[R][G]: what an AppendEntries to peer p carries (craft/messages.rs:420-460,
 * durability.rs:44-70): full copy = the data shards, else shard p */
void orc_craft_assignment(void *h, uint32_t *persist, uint32_t *send) {
    RaftCl *cl = (RaftCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        RaftRep *r = &cl->reps[g];
        uint32_t data = (1u << r->quorum_cnt) - 1u;
        persist[g] = r->full_copy_mode ? data : (1u << r->id);
        for (int p = 0; p < cl->R; p++) send[(size_t)p * G + g] = p == r->id ? 0 : (r->full_copy_mode ? data : (1u << p));
    }
}

void orc_craft_dump(void *h, uint8_t *full_copy_mode, uint8_t *peer_alive, uint64_t *hb_replied, uint64_t *hb_seen,
                    uint8_t *hb_repeat) {
    RaftCl *cl = (RaftCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        RaftRep *r = &cl->reps[g];
        full_copy_mode[g] = r->full_copy_mode; peer_alive[g] = r->peer_alive;
        for (int p = 0; p < cl->R; p++) {
            size_t o = (size_t)p * G + g;
            hb_replied[o] = r->hb_replied[p]; hb_seen[o] = r->hb_seen[p]; hb_repeat[o] = r->hb_repeat[p];
        }
    }
}

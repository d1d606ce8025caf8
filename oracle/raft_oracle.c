/*
 * oracle/raft_oracle.c -- CPU restatement of the Raft LEADER side hot path of
 * Summerset over G independent groups: log append + AppendEntriesReply
 * handling (match-index quorum -> last_commit, last_snap, next_slot back-off).
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

typedef struct {
    uint8_t id, population, quorum_cnt, commit_thresh;
    uint8_t role, leader;
    uint64_t curr_term;
    uint64_t *log_term;       /* term of every entry; index = slot - start_slot */
    uint32_t n_log, cap_log, start_slot;
    uint32_t last_commit, last_snap;
    uint32_t next_slot[MAXR], try_next_slot[MAXR], match_slot[MAXR];
    uint64_t n_committed, n_redirect, n_reject, n_sent;
} RaftRep;

typedef struct {
    uint32_t G, W; uint8_t R;
    RaftRep *reps;
} RaftCl;

static uint32_t log_end(const RaftRep *r) { return r->start_slot + r->n_log; }

static void log_push(RaftRep *r, uint64_t term) {
    if (r->n_log == r->cap_log) {
        r->cap_log = r->cap_log ? r->cap_log * 2 : 16;
        r->log_term = (uint64_t *)realloc(r->log_term, sizeof(uint64_t) * r->cap_log);
    }
    r->log_term[r->n_log++] = term;
}

void *orc_raft_new(uint32_t G, uint8_t R, uint32_t W, uint8_t leader_id, uint64_t term, uint8_t commit_extra) {
    RaftCl *cl = (RaftCl *)calloc(1, sizeof(RaftCl));
    cl->G = G; cl->R = R; cl->W = W;
    cl->reps = (RaftRep *)calloc(G, sizeof(RaftRep));
    for (uint32_t g = 0; g < G; g++) {
        RaftRep *r = &cl->reps[g];
        r->id = leader_id; r->population = R;
        r->quorum_cnt = (uint8_t)(R / 2 + 1);
        r->commit_thresh = (uint8_t)(r->quorum_cnt + commit_extra);
        log_push(r, 0);                                   /* recovery.rs:96-102 dummy entry */
        r->role = ROLE_LEADER; r->leader = leader_id; r->curr_term = term;
        for (int p = 0; p < R; p++) {                     /* leadership.rs:159-168 */
            r->next_slot[p] = log_end(r); r->try_next_slot[p] = log_end(r); r->match_slot[p] = 0;
        }
    }
    return cl;
}

void orc_raft_free(void *h) {
    RaftCl *cl = (RaftCl *)h;
    for (uint32_t g = 0; g < cl->G; g++) free(cl->reps[g].log_term);
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

void orc_raft_leader_append(void *h, const uint32_t *n_new) {
    RaftCl *cl = (RaftCl *)h;
    for (uint32_t g = 0; g < cl->G; g++)
        for (uint32_t k = 0; k < n_new[g]; k++) handle_req_batch(&cl->reps[g], cl->W);
}

/* leadership.rs:16-72; returns 1 iff the role was not Follower and now is */
static int check_term(RaftRep *r, uint8_t peer, uint64_t term) {
    if (term > r->curr_term) {
        r->curr_term = term;
        r->leader = peer;
        if (r->role == ROLE_FOLLOWER) return 0;
        r->role = ROLE_FOLLOWER;
        return 1;
    }
    return 0;
}

static uint64_t term_at(const RaftRep *r, uint32_t slot, uint32_t W, int *ok) {
    /* ring guard shared with the engine: only the last W entries are readable */
    if (slot < r->start_slot || slot >= log_end(r) || slot + W < log_end(r)) { *ok = 0; return 0; }
    *ok = 1;
    return r->log_term[slot - r->start_slot];
}

/* messages.rs:222-388 */
static void handle_msg_append_entries_reply(RaftRep *r, uint32_t W, uint8_t peer, uint64_t term, uint32_t end_slot,
                                            int has_conflict, uint64_t conflict_term, uint32_t conflict_slot) {
    if (check_term(r, peer, term) || r->role != ROLE_LEADER) return;   /* :239-241 */
    if (!has_conflict) {
        if (r->next_slot[peer] > end_slot + 1) return;     /* :245-247 */
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
            if (match_cnt >= r->commit_thresh) new_commit = slot;
        }
        r->n_committed += new_commit - r->last_commit;     /* :278-293 exec submission */
        r->last_commit = new_commit;                       /* :295 */
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
        uint32_t lo = log_end(r) > W ? log_end(r) - W : r->start_slot;
        for (uint32_t s = lo; s < log_end(r); s++) entry_term[(size_t)(s % W) * G + g] = r->log_term[s - r->start_slot];
    }
}

uint64_t orc_raft_total_commits(void *h) {
    RaftCl *cl = (RaftCl *)h;
    uint64_t t = 0;
    for (uint32_t g = 0; g < cl->G; g++) t += cl->reps[g].n_committed;
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

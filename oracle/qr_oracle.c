/*
 * oracle/qr_oracle.c -- CPU restatement of MultiPaxos' near-quorum-read path of Summerset over G
 * independent groups, one replica per group.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/mp_oracle.c header for the rules).
 *
 * Follows src/protocols/multipaxos/:
 *   refresh_highest_slot            quorumread.rs:8-26
 *   inspect_highest_slot            quorumread.rs:30-73
 *   issuing a ReadQuery             request.rs:55-101   (bookkeeping: reads, rq_acks = {me}, max_replies = own inspect)
 *   handle_msg_read_query           quorumread.rs:75-188
 *   handle_msg_read_query_reply     quorumread.rs:190-346
 * Model: keys are small integers (< K), a value is a 32-bit token (0 = no value), a request batch is its list of
 * Put keys and one token -- the value every Put of the batch writes --, so "the latest value for the key in the
 * batch" (quorumread.rs:50-60) is the token of the slot; the error of a highest_slot entry whose batch does not
 * write the key (:62-66) cannot arise.  The log is what the caller shows (start_slot, end, status and token
 * per slot, a ring of W); a query id (client, request id) is a slot index q < Q of the outstanding-query table.
 * Leases (is_stable_leader) are an input flag; the stable leader answers from the KV table (kv[key], 0 = None).
 * Deliberately literal (the match arms of :218-252 as written, incl. the arm that drops a committed value when
 * nothing was known before).
 *
 * PARITY STATUS: "parity unpinned" -- the reference has no unit tests or fixtures for these handlers and cannot be
 * built here; pinned by hand-derived traces (tests/test_oracle_qr.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NONE32 0xFFFFFFFFu
enum { ST_COMMITTED = 3 };
enum { RP_NONE = 0, RP_SLOT = 1, RP_VALUE = 2 };                 /* None / Some((slot, None)) / Some((slot, Some(v))) */
enum { OUT_PENDING = 0, OUT_NOT_FOUND = 1, OUT_RETRY = 2, OUT_VALUE = 3 };

typedef struct { uint8_t state; uint32_t slot, val; } Reply;

typedef struct {
    uint8_t live, n, rq_acks;                                   /* quorum_reads entry: exists, reads.len(), Bitmap */
    Reply *max_replies;                                         /* [B] */
} ReadQueryBk;

typedef struct {
    uint32_t *highest_slot;                                     /* [K], NONE32 = key never seen */
    ReadQueryBk *bk;                                            /* [Q] */
} QrRep;

typedef struct {
    uint32_t G, K, B, Q; uint8_t R, me, quorum_cnt;
    QrRep *reps;
    uint64_t n_good, n_retry, n_not_found, n_conflict;
} QrCl;

void *orc_qr_new(uint32_t G, uint8_t R, uint8_t me, uint32_t K, uint32_t B, uint32_t Q) {
    QrCl *cl = (QrCl *)calloc(1, sizeof(QrCl));
    cl->G = G; cl->R = R; cl->me = me; cl->K = K; cl->B = B; cl->Q = Q;
    cl->quorum_cnt = (uint8_t)(R / 2 + 1);
    cl->reps = (QrRep *)calloc(G, sizeof(QrRep));
    for (uint32_t g = 0; g < G; g++) {
        QrRep *r = &cl->reps[g];
        r->highest_slot = (uint32_t *)malloc(sizeof(uint32_t) * K);
        for (uint32_t k = 0; k < K; k++) r->highest_slot[k] = NONE32;
        r->bk = (ReadQueryBk *)calloc(Q, sizeof(ReadQueryBk));
        for (uint32_t q = 0; q < Q; q++) r->bk[q].max_replies = (Reply *)calloc(B, sizeof(Reply));
    }
    return cl;
}

void orc_qr_free(void *h) {
    QrCl *cl = (QrCl *)h;
    for (uint32_t g = 0; g < cl->G; g++) {
        for (uint32_t q = 0; q < cl->Q; q++) free(cl->reps[g].bk[q].max_replies);
        free(cl->reps[g].bk); free(cl->reps[g].highest_slot);
    }
    free(cl->reps); free(cl);
}

/* quorumread.rs:8-26; put_keys[B][G], 0xFF = that request of the batch is not a Put */
void orc_qr_refresh_highest_slot(void *h, const uint32_t *slot, const uint8_t *put_keys) {
    QrCl *cl = (QrCl *)h;
    for (uint32_t g = 0; g < cl->G; g++) {
        if (slot[g] == NONE32) continue;                         /* no batch for the group */
        QrRep *r = &cl->reps[g];
        for (uint32_t i = 0; i < cl->B; i++) {
            uint8_t key = put_keys[(size_t)i * cl->G + g];
            if (key == 0xFF || key >= cl->K) continue;
            if (r->highest_slot[key] != NONE32) {
                if (slot[g] > r->highest_slot[key]) r->highest_slot[key] = slot[g];
            } else {
                r->highest_slot[key] = slot[g];
            }
        }
    }
}

/* quorumread.rs:30-73 */
static Reply inspect_highest_slot(const QrCl *cl, const QrRep *r, uint32_t g, uint8_t key, const uint32_t *start_slot,
                                  const uint32_t *log_end, const uint8_t *status, const uint32_t *token, uint32_t W) {
    Reply out = {RP_NONE, 0, 0};
    uint32_t slot = r->highest_slot[key];
    if (slot == NONE32) return out;                              /* never seen this key */
    out.state = RP_SLOT; out.slot = slot;
    if (slot < start_slot[g] || slot >= log_end[g]) return out;   /* GCed / not locatable; log_end = start_slot + insts.len() */
    size_t o = (size_t)(slot % W) * cl->G + g;
    if (status[o] < ST_COMMITTED) return out;                    /* not committed on me yet */
    out.state = RP_VALUE; out.val = token[o];
    return out;
}

/* quorumread.rs:75-188: the reply to a ReadQuery of n[g] Gets (keys[B][G]).  stable_leader (may be NULL) [G]: answer
 * from the KV table kv[K][G] instead (:99-147).  Out: state / slot / val [B][G], from_leader[G]. */
void orc_qr_handle_read_query(void *h, const uint8_t *keys, const uint8_t *n, const uint8_t *stable_leader, const uint32_t *kv,
                              const uint32_t *start_slot, const uint32_t *log_end, const uint8_t *status, const uint32_t *token,
                              uint32_t W, uint8_t *o_state, uint32_t *o_slot, uint32_t *o_val, uint8_t *from_leader) {
    QrCl *cl = (QrCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        QrRep *r = &cl->reps[g];
        from_leader[g] = 0;
        for (uint32_t i = 0; i < cl->B; i++) {
            size_t o = (size_t)i * G + g;
            o_state[o] = 0; o_slot[o] = 0; o_val[o] = 0;
        }
        if (n[g] == 0) continue;
        int stable = stable_leader && stable_leader[g];
        from_leader[g] = (uint8_t)stable;
        for (uint32_t i = 0; i < n[g] && i < cl->B; i++) {
            size_t o = (size_t)i * G + g;
            uint8_t key = keys[o];
            Reply rp = {RP_NONE, 0, 0};
            if (stable) {
                uint32_t v = kv[(size_t)key * G + g];           /* do_sync_cmd(Get) -> value.map(|v| (0, Some(v))) :124-131 */
                if (v) { rp.state = RP_VALUE; rp.slot = 0; rp.val = v; }
            } else {
                rp = inspect_highest_slot(cl, r, g, key, start_slot, log_end, status, token, W);   /* :158 */
            }
            o_state[o] = rp.state; o_slot[o] = rp.slot; o_val[o] = rp.val;
        }
    }
}

/* request.rs:55-101: the issuer's bookkeeping of query q; n[g] = 0: no query issued for the group.
 * own state / slot / val [B][G] = its own inspect_highest_slot of every key (:76) */
void orc_qr_issue(void *h, uint32_t q, const uint8_t *n, const uint8_t *state, const uint32_t *slot, const uint32_t *val) {
    QrCl *cl = (QrCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        if (n[g] == 0) continue;
        ReadQueryBk *bk = &cl->reps[g].bk[q];
        bk->live = 1; bk->n = n[g] < cl->B ? n[g] : (uint8_t)cl->B;
        bk->rq_acks = (uint8_t)(1u << cl->me);                   /* :100 mark myself as replied */
        for (uint32_t i = 0; i < bk->n; i++) {
            size_t o = (size_t)i * G + g;
            bk->max_replies[i].state = state[o]; bk->max_replies[i].slot = slot[o]; bk->max_replies[i].val = val[o];
        }
    }
}

/* quorumread.rs:190-346 for one reply; returns 1 if the clients were answered (outcome / out_val rows written) */
static int handle_msg_read_query_reply(QrCl *cl, QrRep *r, uint32_t g, uint32_t q, uint8_t peer, const Reply *replies,
                                       int from_leader, uint8_t *outcome, uint32_t *out_val) {
    int can_reply = 0;
    ReadQueryBk *bk = &r->bk[q];
    if (bk->live) {                                              /* :205 */
        if (from_leader) {                                       /* :206-210 */
            for (uint32_t i = 0; i < bk->n; i++) bk->max_replies[i] = replies[i];
            can_reply = 1;
        } else if (!((bk->rq_acks >> peer) & 1)) {               /* :211 */
            for (uint32_t i = 0; i < bk->n; i++) {               /* :215-253 */
                const Reply *reply = &replies[i];
                Reply *max_reply = &bk->max_replies[i];
                if (reply->state == RP_NONE) {
                } else if (reply->state == RP_SLOT) {
                    if (max_reply->state == RP_NONE) {
                        max_reply->state = RP_SLOT; max_reply->slot = reply->slot; max_reply->val = 0;
                    } else if (reply->slot > max_reply->slot) {
                        max_reply->state = RP_SLOT; max_reply->slot = reply->slot; max_reply->val = 0;
                    }
                } else {
                    if (max_reply->state == RP_NONE) {           /* :231-233: the value is not kept */
                        max_reply->state = RP_SLOT; max_reply->slot = reply->slot; max_reply->val = 0;
                    } else if (max_reply->state == RP_SLOT) {
                        if (reply->slot >= max_reply->slot) *max_reply = *reply;
                    } else {
                        if (reply->slot > max_reply->slot) {
                            *max_reply = *reply;
                        } else if (reply->slot == max_reply->slot && max_reply->val != reply->val) {
                            cl->n_conflict++;                    /* :243-250 logged_err: the handler returns here */
                            return 0;
                        }
                    }
                }
            }
            bk->rq_acks |= (uint8_t)(1u << peer);               /* :254 */
            if (__builtin_popcount(bk->rq_acks) >= cl->quorum_cnt) can_reply = 1;   /* :258-265 */
        }
    }
    if (!can_reply) return 0;
    bk->live = 0;                                                /* :271 remove */
    for (uint32_t i = 0; i < bk->n; i++) {                       /* :272-318 */
        size_t o = (size_t)i * cl->G + g;
        const Reply *rp = &bk->max_replies[i];
        if (rp->state == RP_NONE) { outcome[o] = OUT_NOT_FOUND; out_val[o] = 0; cl->n_not_found++; }
        else if (rp->state == RP_SLOT) { outcome[o] = OUT_RETRY; out_val[o] = 0; cl->n_retry++; }
        else { outcome[o] = OUT_VALUE; out_val[o] = rp->val; cl->n_good++; }
    }
    return 1;
}

#define CTL_IDENTITY 0x00FAC688u

/* One ReadQueryReply per (peer, group) for query q: state / slot / val [R][B][G]; flags [R][G] bit0 a reply is there,
 * bit1 from_leader; peers in `order[g]` order (ackctl encoding, NULL = identity).  Out: outcome / out_val [B][G]
 * (OUT_PENDING rows where nothing was answered), done[G]. */
void orc_qr_handle_replies(void *h, uint32_t q, const uint8_t *state, const uint32_t *slot, const uint32_t *val,
                           const uint8_t *flags, const uint32_t *order, uint8_t *outcome, uint32_t *out_val, uint8_t *done) {
    QrCl *cl = (QrCl *)h;
    const uint32_t G = cl->G, B = cl->B;
    Reply *tmp = (Reply *)calloc(B, sizeof(Reply));
    for (uint32_t g = 0; g < G; g++) {
        done[g] = 0;
        for (uint32_t i = 0; i < B; i++) { outcome[(size_t)i * G + g] = OUT_PENDING; out_val[(size_t)i * G + g] = 0; }
        uint32_t ctl = order ? order[g] : CTL_IDENTITY;
        for (int oi = 0; oi < cl->R; oi++) {
            uint32_t p = (ctl >> (3 * oi)) & 7u;
            if (p == cl->me || p >= cl->R) continue;
            uint8_t f = flags[(size_t)p * G + g];
            if (!(f & 1)) continue;
            for (uint32_t i = 0; i < B; i++) {
                size_t o = ((size_t)p * B + i) * G + g;
                tmp[i].state = state[o]; tmp[i].slot = slot[o]; tmp[i].val = val[o];
            }
            if (handle_msg_read_query_reply(cl, &cl->reps[g], g, q, (uint8_t)p, tmp, (f >> 1) & 1, outcome, out_val)) done[g] = 1;
        }
    }
    free(tmp);
}

/* highest_slot [K][G]; per query: live / n / rq_acks [Q][G], max_replies state / slot / val [Q][B][G] (rows of dead
 * queries and rows >= n read 0) */
void orc_qr_dump(void *h, uint32_t *highest_slot, uint8_t *live, uint8_t *n, uint8_t *rq_acks, uint8_t *mx_state,
                 uint32_t *mx_slot, uint32_t *mx_val, uint64_t counters[4]) {
    QrCl *cl = (QrCl *)h;
    const uint32_t G = cl->G, B = cl->B;
    for (uint32_t g = 0; g < G; g++) {
        QrRep *r = &cl->reps[g];
        for (uint32_t k = 0; k < cl->K; k++) highest_slot[(size_t)k * G + g] = r->highest_slot[k];
        for (uint32_t q = 0; q < cl->Q; q++) {
            ReadQueryBk *bk = &r->bk[q];
            size_t o = (size_t)q * G + g;
            live[o] = bk->live; n[o] = bk->live ? bk->n : 0; rq_acks[o] = bk->live ? bk->rq_acks : 0;
            for (uint32_t i = 0; i < B; i++) {
                size_t oo = ((size_t)q * B + i) * G + g;
                int on = bk->live && i < bk->n;
                mx_state[oo] = on ? bk->max_replies[i].state : 0;
                mx_slot[oo] = on && bk->max_replies[i].state ? bk->max_replies[i].slot : 0;
                mx_val[oo] = on && bk->max_replies[i].state == RP_VALUE ? bk->max_replies[i].val : 0;
            }
        }
    }
    counters[0] = cl->n_good; counters[1] = cl->n_retry; counters[2] = cl->n_not_found; counters[3] = cl->n_conflict;
}

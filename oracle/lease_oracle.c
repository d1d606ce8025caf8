/* CPU ORACLE -- test infrastructure only (never linked into or imported by summerset_amd/).
 *
 * `LeaseManager` (src/server/leaseman.rs:132-935: the public object + LeaseManagerLogicTask) for G independent groups,
 * one replica id `me`: the push-based guard + promise leases (grantor side: guards_sent -> promises_sent; holder side:
 * guards_held -> promises_held), revocation, lease numbers, and the timers behind them (utils/timer.rs: kickoff / extend /
 * explode).  The reference runs on tokio timers and channels; here time is an argument and one call =
 * one notice per group: `orc_lease_step(now, notice)` first delivers, in deadline order, the timeout notices of every
 * timer that exploded up to `now` (they were sent into the notice channel when they exploded, i.e. before the notice that
 * arrives now: the channel is FIFO), then handles the notice exactly as `run()` (:837-933) + `handle_notice` (:791-835) do,
 * and returns the actions the protocol module would drain with get_action() (:275-290; NextRefresh marks the peer there).
 *
 * PARITY STATUS: pinned by the reference's OWN unit tests -- leaseman.rs:1079-2301 guard_expired, promise_expired,
 * promise_refresh, revoke_replied, revoke_expired, regrant_higher, mutual_leases -- restated step by step with their
 * timings in tests/test_oracle_lease.py (every add_notice, every asserted get_action / grant_set / lease_cnt /
 * attempt_refresh).  What those tests do not reach (accept_bar, ClearHeld, outdated Revoke, PromiseReply{held: false}) is
 * covered by hand-derived traces of the cited lines. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXR 8
enum { N_NONE = 0, N_NEW_GRANTS = 1, N_DO_REVOKE = 2, N_CLEAR_HELD = 3, N_RECV_MSG = 4 };
enum { M_GUARD = 0, M_GUARD_REPLY = 1, M_PROMISE = 2, M_PROMISE_REPLY = 3, M_REVOKE = 4, M_REVOKE_REPLY = 5 };
enum { A_SEND = 1, A_BCAST = 2, A_NEXT_REFRESH = 3, A_GRANT_REMOVED = 4, A_LEASE_CLEARED = 5, A_GRANT_TIMEOUT = 6,
       A_LEASE_TIMEOUT = 7, A_HIGHER_NUMBER = 8, A_GUARD_ACCEPT_BAR = 9 };
#define PEERS_ALL 0xFF          /* Option<Bitmap>::None */
#define ACT_CAP 20

typedef struct { uint8_t on; uint64_t deadline; uint8_t exploded; uint64_t num; } Tm;   /* deadline 0 = not kicked off */
typedef struct {
    uint64_t active_num;
    Tm gs[MAXR], gh[MAXR], ps[MAXR], ph[MAXR];      /* guards_sent, guards_held, promises_sent (.0), promises_held */
    uint8_t revoking[MAXR];                          /* promises_sent .1 (never set in this version of the reference) */
    uint8_t refresh_mark;                            /* LeaseManager::refresh_mark */
} Lm;
typedef struct { uint32_t G; uint8_t R, me; uint64_t expire; Lm *g; } LmCl;

typedef struct {                                     /* one group's actions of a call */
    uint8_t n, kind[ACT_CAP], peer[ACT_CAP], mask[ACT_CAP], msg[ACT_CAP], flag[ACT_CAP];
    uint64_t num[ACT_CAP], bar[ACT_CAP];
} Acts;
static void act(Acts *a, uint64_t num, uint8_t kind, uint8_t peer, uint8_t mask, uint8_t msg, uint8_t flag, uint64_t bar) {
    if (a->n >= ACT_CAP) return;
    const int i = a->n++;
    a->num[i] = num; a->kind[i] = kind; a->peer[i] = peer; a->mask[i] = mask; a->msg[i] = msg; a->flag[i] = flag; a->bar[i] = bar;
    (void)0;
}

/* new_and_setup (:169-233) */
void *orc_lease_new(uint32_t G, uint8_t R, uint8_t me, uint64_t expire_ms, uint64_t hb_send_ms) {
    if (expire_ms < 100 || expire_ms > 10000) return NULL;            /* :178-185 */
    if (2 * hb_send_ms >= expire_ms) return NULL;                       /* :186-193 */
    if (R == 0 || R > MAXR || me >= R) return NULL;
    LmCl *c = (LmCl *)calloc(1, sizeof(LmCl));
    c->G = G; c->R = R; c->me = me; c->expire = expire_ms;
    c->g = (Lm *)calloc(G, sizeof(Lm));
    return c;
}
void orc_lease_free(void *h) { if (h) { free(((LmCl *)h)->g); free(h); } }

static void tm_kickoff(Tm *t, uint64_t now, uint64_t dur) { t->deadline = now + dur; t->exploded = 0; }   /* timer.rs kickoff */
static void tm_extend(Tm *t, uint64_t now, uint64_t dur) {              /* timer.rs extend */
    if (t->deadline != 0) { if (t->deadline < now) t->deadline = now; t->deadline += dur; } else t->deadline = now + dur;
    t->exploded = 0;
}

static void handle_grant_timeout(LmCl *c, Lm *m, Acts *a, uint64_t num, uint8_t peer) {   /* :752-767 */
    (void)c;
    m->gs[peer].on = 0; m->ps[peer].on = 0;
    act(a, num, A_GRANT_TIMEOUT, peer, 0, 0, 0, 0);
}
static void handle_lease_timeout(LmCl *c, Lm *m, Acts *a, uint64_t num, uint8_t peer) {   /* :770-788 */
    (void)c;
    m->gh[peer].on = 0; m->ph[peer].on = 0;
    act(a, num, A_LEASE_TIMEOUT, peer, 0, 0, 0, 0);
}

/* run() :840-926: the lease number filter in front of every notice; returns 1 if the notice is to be handled */
static int admit(LmCl *c, Lm *m, Acts *a, uint64_t num, int is_revoke_msg, uint8_t peer) {
    if (num < m->active_num) {
        if (is_revoke_msg) act(a, num, A_SEND, peer, 0, M_REVOKE_REPLY, 0, 0);   /* :851-877 prompt RevokeReply { held: false } */
        return 0;
    }
    if (num > m->active_num) {                                          /* :880-915 */
        for (int p = 0; p < c->R; p++) { m->gs[p].on = 0; m->gh[p].on = 0; if (p != c->me) { m->ps[p].on = 0; m->ph[p].on = 0; } }
        m->active_num = num;
        act(a, num, A_HIGHER_NUMBER, 0, 0, 0, 0, 0);
    }
    return 1;
}

/* the timers that exploded up to `now`, as the notices they sent, in the order they exploded */
static void fire_timers(LmCl *c, Lm *m, Acts *a, uint64_t now) {
    for (;;) {
        int best = -1, side = 0; uint64_t dl = ~0ull;
        for (int p = 0; p < c->R; p++) {
            /* grantor side: only promises_sent timers are ever kicked off (guards_sent ones are created, never started, :399-411) */
            if (m->ps[p].on && m->ps[p].deadline && !m->ps[p].exploded && m->ps[p].deadline <= now && m->ps[p].deadline < dl) { best = p; side = 0; dl = m->ps[p].deadline; }
            if (m->gh[p].on && m->gh[p].deadline && !m->gh[p].exploded && m->gh[p].deadline <= now && m->gh[p].deadline < dl) { best = p; side = 1; dl = m->gh[p].deadline; }
            if (m->ph[p].on && m->ph[p].deadline && !m->ph[p].exploded && m->ph[p].deadline <= now && m->ph[p].deadline < dl) { best = p; side = 2; dl = m->ph[p].deadline; }
        }
        if (best < 0) return;
        Tm *t = side == 0 ? &m->ps[best] : (side == 1 ? &m->gh[best] : &m->ph[best]);
        t->exploded = 1;
        const uint64_t num = t->num;                                    /* the number the timer's closure captured */
        if (!admit(c, m, a, num, 0, (uint8_t)best)) { t->on = 0; continue; }   /* (cannot happen: a higher number drops the timers) */
        if (side == 0) handle_grant_timeout(c, m, a, num, (uint8_t)best); else handle_lease_timeout(c, m, a, num, (uint8_t)best);
    }
}

/* One call: timers up to now, then one notice per group (kind[g] = N_NONE: timers only).
 * notice fields: num, peer (N_RECV_MSG), peers (bitmap or PEERS_ALL: N_NEW_GRANTS / N_DO_REVOKE), msg + held (N_RECV_MSG),
 * has_bar + bar (accept_bar of NewGrants / of a Guard message).  Outputs: act_n[G] and the action arrays [ACT_CAP][G]. */
void orc_lease_step(void *hh, uint64_t now, const uint8_t *kind, const uint64_t *num, const uint8_t *peer, const uint8_t *peers,
                    const uint8_t *msg, const uint8_t *held, const uint8_t *has_bar, const uint64_t *bar, uint8_t *act_n,
                    uint64_t *act_num, uint8_t *act_kind, uint8_t *act_peer, uint8_t *act_mask, uint8_t *act_msg, uint8_t *act_flag,
                    uint64_t *act_bar) {
    LmCl *c = (LmCl *)hh;
    const uint8_t all = (uint8_t)((1u << c->R) - 1u);
    for (uint32_t g = 0; g < c->G; g++) {
        Lm *m = &c->g[g];
        Acts a; a.n = 0;
        fire_timers(c, m, &a, now);
        const uint8_t k = kind ? kind[g] : N_NONE;
        if (k != N_NONE) {
            const uint64_t ln = num[g];
            const uint8_t pe = peer ? peer[g] : 0;
            if (admit(c, m, &a, ln, k == N_RECV_MSG && msg[g] == M_REVOKE, pe)) {
                if (k == N_NEW_GRANTS) {                                /* :385-439 */
                    const uint8_t ps = peers[g] == PEERS_ALL ? all : (uint8_t)(peers[g] & all);
                    uint8_t bc = ps;
                    for (int p = 0; p < c->R; p++) {
                        if (p == c->me || !((ps >> p) & 1)) { bc &= (uint8_t)~(1u << p); continue; }
                        if (m->ps[p].on) { bc &= (uint8_t)~(1u << p); continue; }   /* already granting */
                        m->gs[p].on = 1; m->gs[p].deadline = 0; m->gs[p].exploded = 0; m->gs[p].num = ln;
                    }
                    act(&a, ln, A_BCAST, 0, bc, M_GUARD, has_bar[g], has_bar[g] ? bar[g] : 0);
                } else if (k == N_DO_REVOKE) {                          /* :442-481 */
                    const uint8_t ps = peers[g] == PEERS_ALL ? all : (uint8_t)(peers[g] & all);
                    uint8_t bc = ps;
                    for (int p = 0; p < c->R; p++) {
                        if (p == c->me || !((ps >> p) & 1)) { bc &= (uint8_t)~(1u << p); continue; }
                        m->gs[p].on = 0;
                        if (!m->ps[p].on) bc &= (uint8_t)~(1u << p);
                    }
                    if (bc) act(&a, ln, A_BCAST, 0, bc, M_REVOKE, 0, 0);
                } else if (k == N_CLEAR_HELD) {                         /* :484-498 */
                    for (int p = 0; p < c->R; p++) { m->gh[p].on = 0; if (p != c->me) m->ph[p].on = 0; }
                    act(&a, ln, A_LEASE_CLEARED, 0, 0, 0, 0, 0);
                } else if (k == N_RECV_MSG && pe < c->R && pe != c->me) {
                    switch (msg[g]) {
                    case M_GUARD:                                       /* :501-555 */
                        if (m->ph[pe].on) break;
                        m->gh[pe].on = 1; m->gh[pe].num = ln; tm_kickoff(&m->gh[pe], now, c->expire);
                        if (has_bar[g]) act(&a, ln, A_GUARD_ACCEPT_BAR, pe, 0, 0, 1, bar[g]);
                        act(&a, ln, A_SEND, pe, 0, M_GUARD_REPLY, 0, 0);
                        break;
                    case M_GUARD_REPLY:                                 /* :558-592 */
                        if (!m->gs[pe].on) break;
                        { Tm t = m->gs[pe]; m->gs[pe].on = 0;
                          if (t.exploded) break;
                          tm_kickoff(&t, now, c->expire + c->expire);     /* T_guard + T_promise */
                          m->ps[pe] = t; m->ps[pe].on = 1; m->revoking[pe] = 0; }
                        act(&a, ln, A_SEND, pe, 0, M_PROMISE, 0, 0);
                        break;
                    case M_PROMISE:                                     /* :595-645 */
                        if (m->gh[pe].on) {
                            Tm t = m->gh[pe]; m->gh[pe].on = 0;
                            tm_kickoff(&t, now, c->expire);
                            m->ph[pe] = t; m->ph[pe].on = 1;
                            act(&a, ln, A_SEND, pe, 0, M_PROMISE_REPLY, 1, 0);
                        } else if (m->ph[pe].on) {
                            tm_kickoff(&m->ph[pe], now, c->expire);
                            act(&a, ln, A_SEND, pe, 0, M_PROMISE_REPLY, 1, 0);
                        } else act(&a, ln, A_SEND, pe, 0, M_PROMISE_REPLY, 0, 0);
                        break;
                    case M_PROMISE_REPLY:                               /* :648-693 */
                        if (!m->ps[pe].on) break;
                        if (!held[g]) { m->gs[pe].on = 0; m->ps[pe].on = 0; act(&a, ln, A_GRANT_REMOVED, pe, 0, 0, 0, 0); break; }
                        if (m->ps[pe].exploded) break;
                        if (!m->revoking[pe]) {
                            tm_kickoff(&m->ps[pe], now, c->expire);
                            act(&a, ln, A_NEXT_REFRESH, pe, 0, 0, 0, 0);
                            m->refresh_mark |= (uint8_t)(1u << pe);     /* get_action :281-284 */
                        }
                        break;
                    case M_REVOKE: {                                    /* :696-725 */
                        m->gh[pe].on = 0;
                        const uint8_t h = m->ph[pe].on; m->ph[pe].on = 0;
                        act(&a, ln, A_SEND, pe, 0, M_REVOKE_REPLY, h, 0);
                        break; }
                    case M_REVOKE_REPLY:                                /* :728-749 */
                        m->gs[pe].on = 0; m->ps[pe].on = 0;
                        act(&a, ln, A_GRANT_REMOVED, pe, 0, 0, held[g], 0);
                        break;
                    default: break;
                    }
                }
            }
        }
        act_n[g] = a.n;
        for (int i = 0; i < ACT_CAP; i++) {
            const size_t o = (size_t)i * c->G + g;
            const int in = i < a.n;
            act_num[o] = in ? a.num[i] : 0; act_kind[o] = in ? a.kind[i] : 0; act_peer[o] = in ? a.peer[i] : 0;
            act_mask[o] = in ? a.mask[i] : 0; act_msg[o] = in ? a.msg[i] : 0; act_flag[o] = in ? a.flag[i] : 0; act_bar[o] = in ? a.bar[i] : 0;
        }
    }
}

/* attempt_refresh (:296-317): peers[g] = bitmap or PEERS_ALL (None); 0 with call[g] = 0: no call.  to_refresh[g] out. */
void orc_lease_attempt_refresh(void *hh, uint64_t now, const uint8_t *call, const uint8_t *peers, uint8_t *to_refresh) {
    LmCl *c = (LmCl *)hh;
    for (uint32_t g = 0; g < c->G; g++) {
        Lm *m = &c->g[g];
        uint8_t out = 0;
        if (call[g])
            for (int p = 0; p < c->R; p++) {
                if (!m->ps[p].on || p == c->me) continue;
                if (!(peers[g] == PEERS_ALL || ((peers[g] >> p) & 1))) continue;
                if (!((m->refresh_mark >> p) & 1)) continue;            /* refresh_mark.remove(&peer) */
                m->refresh_mark &= (uint8_t)~(1u << p);
                if (!m->revoking[p]) { tm_extend(&m->ps[p], now, c->expire); out |= (uint8_t)(1u << p); }
            }
        to_refresh[g] = out;
    }
}

/* grant_set (:236-243), lease_set (:246-253), lease_cnt (:257-259) and the rest of the state */
void orc_lease_dump(void *hh, uint64_t *active_num, uint8_t *grant_set, uint8_t *lease_set, uint8_t *lease_cnt, uint8_t *guards_sent,
                    uint8_t *guards_held, uint8_t *refresh_mark, uint64_t *ps_deadline, uint64_t *gh_deadline, uint64_t *ph_deadline) {
    LmCl *c = (LmCl *)hh;
    for (uint32_t g = 0; g < c->G; g++) {
        const Lm *m = &c->g[g];
        uint8_t gs = 0, ls = 0, a = 0, b = 0;
        for (int p = 0; p < c->R; p++) {
            gs |= (uint8_t)(m->ps[p].on << p); ls |= (uint8_t)(m->ph[p].on << p); a |= (uint8_t)(m->gs[p].on << p); b |= (uint8_t)(m->gh[p].on << p);
            const size_t o = (size_t)p * c->G + g;
            ps_deadline[o] = m->ps[p].on ? m->ps[p].deadline : 0; gh_deadline[o] = m->gh[p].on ? m->gh[p].deadline : 0;
            ph_deadline[o] = m->ph[p].on ? m->ph[p].deadline : 0;
        }
        active_num[g] = m->active_num; grant_set[g] = gs; lease_set[g] = ls; lease_cnt[g] = (uint8_t)(1 + __builtin_popcount(ls));
        guards_sent[g] = a; guards_held[g] = b; refresh_mark[g] = m->refresh_mark;
    }
}

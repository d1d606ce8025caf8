/* CPU ORACLE -- test infrastructure only (never linked into or imported by summerset_amd/).
 *
 * `Heartbeater` (src/server/heartbeat.rs:26-296) for G independent groups, one replica id `me`: the hear timers
 * (one per peer, random timeout in [hear_timeout_min, hear_timeout_max]), the send ticker, and the reply counters /
 * peer_alive bitmap.  The reference runs on tokio timers and `rand::rng()`; both are made explicit here, as SURVEY.md
 * §8c prescribes ("randomised election timeouts => timeouts are explicit stream events"): every call that looks at a
 * clock takes `now_ms`, every kickoff takes the random draw it would have made (a u32 per timer; timeout = min + draw
 * mod (max - min + 1), `random_range(min..=max)` :179-182).
 *
 * What a timer is (utils/timer.rs): `kickoff(dur)` arms it and clears `exploded`; when `dur` has passed it sets
 * `exploded` and fires the callback (heartbeat.rs:83-87: the peer id goes into the timeout channel); `cancel()` disarms.
 * get_event (:130-160) pops the channel and drops an entry whose timer is no longer exploded (re-armed since, :137-139).
 * LS-1 reading: `poll(now)` delivers, in peer order, ONE HearTimeout event per timer that exploded since the last
 * poll and has not been re-armed, then the SendTicked event if the ticker is due.  The ticker is a tokio interval with
 * MissedTickBehavior::Skip (:97-98): first tick immediately, then at multiples of the period from its start; late ticks
 * are skipped, not bunched.
 *
 * PARITY STATUS: "parity unpinned" (the reference has no unit tests for this module: "TODO: add Heartbeater module unit
 * tests", :299); pinned by hand-derived traces (tests/test_oracle_hb.py). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXR 8
#define NO_REP 0xFF
#define ALL_PEERS 0xFE

typedef struct {
    uint64_t deadline[MAXR];        /* 0 = not armed */
    uint8_t exploded[MAXR], queued[MAXR];
    uint8_t is_sending;
    uint64_t tick_start, next_tick; /* interval: created at tick_start (the create call), next due time */
    uint64_t cnt0[MAXR], cnt1[MAXR]; uint8_t rep[MAXR];   /* reply_cnts: (#hb_replied, # seen at last send, repetition) */
    uint8_t alive;                  /* peer_alive bitmap, bit p */
} Hb;

typedef struct { uint32_t G; uint8_t R, me; uint64_t tmin, tmax, period; Hb *g; } HbCl;

/* new_and_setup (:62-127): the three validity checks, all-true peer_alive, reply_cnts (1, 0, 0) */
void *orc_hb_new(uint32_t G, uint8_t R, uint8_t me, uint64_t hear_min_ms, uint64_t hear_max_ms, uint64_t send_ms, uint64_t now_ms) {
    if (hear_min_ms < 100) return NULL;                                   /* :69-74 */
    if (hear_max_ms < hear_min_ms + 100) return NULL;                     /* :75-81 */
    if (send_ms < 1 || send_ms > hear_max_ms) return NULL;                /* :82-89 */
    if (R == 0 || R > MAXR || me >= R) return NULL;
    HbCl *c = (HbCl *)calloc(1, sizeof(HbCl));
    c->G = G; c->R = R; c->me = me; c->tmin = hear_min_ms; c->tmax = hear_max_ms; c->period = send_ms;
    c->g = (Hb *)calloc(G, sizeof(Hb));
    for (uint32_t i = 0; i < G; i++) {
        Hb *h = &c->g[i];
        for (int p = 0; p < R; p++) h->cnt0[p] = 1;                       /* :112-114 (the entry of `me` is never touched) */
        h->alive = (uint8_t)((1u << R) - 1u);                             /* :125 Bitmap::new(population, true) */
        h->tick_start = h->next_tick = now_ms;                            /* time::interval: first tick completes immediately */
    }
    return c;
}
void orc_hb_free(void *h) { if (h) { free(((HbCl *)h)->g); free(h); } }

/* set_sending (:130-132) */
void orc_hb_set_sending(void *hh, const uint8_t *sending) {
    HbCl *c = (HbCl *)hh;
    for (uint32_t g = 0; g < c->G; g++) if (sending[g] != NO_REP) c->g[g].is_sending = sending[g] ? 1 : 0;
}

static void kickoff(HbCl *c, Hb *h, int p, uint64_t now, uint32_t draw) {  /* :174-185 */
    h->exploded[p] = 0; h->queued[p] = 0;                                 /* timer.cancel(), then kickoff clears `exploded` */
    h->deadline[p] = now + c->tmin + (uint64_t)draw % (c->tmax - c->tmin + 1);
}
/* kickoff_hear_timer (:189-210): peer[g] = a peer id, ALL_PEERS (None), or NO_REP (no call for this group);
 * draw[p][g]: the random draw of timer p */
void orc_hb_kickoff_hear_timer(void *hh, const uint8_t *peer, uint64_t now_ms, const uint32_t *draw) {
    HbCl *c = (HbCl *)hh;
    for (uint32_t g = 0; g < c->G; g++) {
        Hb *h = &c->g[g];
        if (peer[g] == NO_REP) continue;
        if (peer[g] == ALL_PEERS) {
            for (int p = 0; p < c->R; p++) if (p != c->me) kickoff(c, h, p, now_ms, draw[(size_t)p * c->G + g]);   /* :204-208 */
        } else if (peer[g] < c->R && peer[g] != c->me) {                   /* :194-195: my own id is a no-op */
            kickoff(c, h, peer[g], now_ms, draw[(size_t)peer[g] * c->G + g]);
        }
    }
}

/* get_event (:134-160), drained: timeouts[p][g] = 1 for every HearTimeout{peer: p} delivered now, send_ticked[g] */
void orc_hb_poll(void *hh, uint64_t now_ms, uint8_t *timeouts, uint8_t *send_ticked) {
    HbCl *c = (HbCl *)hh;
    for (uint32_t g = 0; g < c->G; g++) {
        Hb *h = &c->g[g];
        for (int p = 0; p < c->R; p++) {
            timeouts[(size_t)p * c->G + g] = 0;
            if (p == c->me) continue;
            if (h->deadline[p] != 0 && now_ms >= h->deadline[p]) {         /* the timer task: explode, send p into the channel */
                h->deadline[p] = 0; h->exploded[p] = 1; h->queued[p] = 1;
            }
            if (h->queued[p]) {                                            /* :136-141 */
                h->queued[p] = 0;
                if (h->exploded[p]) timeouts[(size_t)p * c->G + g] = 1;
            }
        }
        send_ticked[g] = 0;
        if (h->is_sending && now_ms >= h->next_tick) {                     /* :154-156; Skip: next tick on the period grid */
            send_ticked[g] = 1;
            h->next_tick = h->tick_start + ((now_ms - h->tick_start) / c->period + 1) * c->period;
        }
    }
}

/* clear_reply_cnts (:223-241) */
void orc_hb_clear_reply_cnts(void *hh, const uint8_t *peer) {
    HbCl *c = (HbCl *)hh;
    for (uint32_t g = 0; g < c->G; g++) {
        Hb *h = &c->g[g];
        if (peer[g] == NO_REP) continue;
        for (int p = 0; p < c->R; p++)
            if (p != c->me && (peer[g] == ALL_PEERS || peer[g] == p)) { h->cnt0[p] = 1; h->cnt1[p] = 0; h->rep[p] = 0; }
    }
}

/* update_bcast_cnts (:247-281): flags[g] != 0 = the call happens for this group; peer_death[g] = its return value */
void orc_hb_update_bcast_cnts(void *hh, const uint8_t *flags, uint8_t *peer_death) {
    HbCl *c = (HbCl *)hh;
    const uint8_t thresh = (uint8_t)(c->tmin / c->period);                /* :262-264 `as u8` */
    for (uint32_t g = 0; g < c->G; g++) {
        Hb *h = &c->g[g];
        peer_death[g] = 0;
        if (!flags[g]) continue;
        for (int p = 0; p < c->R; p++) {
            if (p == c->me) continue;
            if (h->cnt0[p] > h->cnt1[p]) { h->cnt1[p] = h->cnt0[p]; h->rep[p] = 0; }   /* :251-255 */
            else {
                h->rep[p]++;                                               /* :259 */
                if (h->rep[p] > thresh) {                                  /* :266-276 */
                    if ((h->alive >> p) & 1) { h->alive &= (uint8_t)~(1u << p); peer_death[g] = 1; }
                    h->rep[p] = 0;
                }
            }
        }
    }
}

/* update_heard_cnt (:285-300) */
void orc_hb_update_heard_cnt(void *hh, const uint8_t *peer) {
    HbCl *c = (HbCl *)hh;
    for (uint32_t g = 0; g < c->G; g++) {
        Hb *h = &c->g[g];
        if (peer[g] == NO_REP || peer[g] >= c->R || peer[g] == c->me) continue;
        h->cnt0[peer[g]]++;
        h->alive |= (uint8_t)(1u << peer[g]);
    }
}

void orc_hb_dump(void *hh, uint64_t *deadline, uint8_t *exploded, uint8_t *is_sending, uint64_t *next_tick, uint64_t *cnt0,
                 uint64_t *cnt1, uint8_t *rep, uint8_t *alive) {
    HbCl *c = (HbCl *)hh;
    for (uint32_t g = 0; g < c->G; g++) {
        const Hb *h = &c->g[g];
        for (int p = 0; p < c->R; p++) {
            const size_t o = (size_t)p * c->G + g;
            deadline[o] = h->deadline[p]; exploded[o] = h->exploded[p]; cnt0[o] = p == c->me ? 0 : h->cnt0[p]; cnt1[o] = h->cnt1[p]; rep[o] = h->rep[p];
        }
        is_sending[g] = h->is_sending; next_tick[g] = h->next_tick; alive[g] = h->alive;
    }
}

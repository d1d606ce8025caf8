// Batched `Heartbeater` (src/server/heartbeat.rs:26-296) for G groups, one replica id per object,
// lane = group: the hear timers (a deadline per (peer, group); a kickoff takes the random draw the reference makes with
// rand::rng(), so runs are reproducible), the send ticker (tokio interval, MissedTickBehavior::Skip: ticks on the
// period grid, late ones skipped) and the reply counters / peer_alive bitmap.  Clocks are explicit: every call that
// looks at time takes now_ms.  `smr_hb_poll` is get_event (:134-160) drained: per (peer, group) whether a HearTimeout is
// delivered now, per group whether the send ticker fired -- arrays a host hands straight to the protocol engines
// (MultiPaxos `timeout_rep` / `timeout_src`, RSPaxos `become_leader`, the CRaft leader's heartbeat tick) without a
// round trip through host memory.  State arrays are [R][G] / [G], every access one contiguous request per wavefront.
#include <string.h>

#include "smr_common.h"

namespace smr {

constexpr uint32_t HB_MAXR = 8;
constexpr uint8_t HB_NONE = 0xFF, HB_ALL = 0xFE;

struct HbView {
    uint32_t G, R, me;
    uint64_t tmin, tmax, period;
    uint64_t *deadline;             // [R][G] 0 = not armed
    uint8_t *exploded, *queued;     // [R][G]
    uint8_t *is_sending;            // [G]
    uint64_t *tick_start, *next_tick;   // [G]
    uint64_t *cnt0, *cnt1;          // [R][G] reply_cnts .0 / .1
    uint8_t *rep;                   // [R][G] reply_cnts .2
    uint8_t *alive;                 // [G] peer_alive bitmap
};

__global__ __launch_bounds__(256) void hb_init_kernel(const HbView v, uint64_t now) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    for (uint32_t p = 0; p < v.R; p++) v.cnt0[(size_t)p * v.G + g] = 1;   // :112-114 (1, 0, 0)
    v.alive[g] = (uint8_t)((1u << v.R) - 1u);                              // :125
    v.tick_start[g] = now; v.next_tick[g] = now;                            // first tick completes immediately
}

// kickoff_hear_timer (:189-210) with kickoff_timer_inner (:174-185)
__global__ __launch_bounds__(256) void hb_kickoff_kernel(const HbView v, const uint8_t *__restrict__ peer, uint64_t now,
                                                         const uint32_t *__restrict__ draw) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    const uint32_t pe = peer[g];
    if (pe == HB_NONE) return;
    for (uint32_t p = 0; p < v.R; p++) {
        if (p == v.me || !(pe == HB_ALL || pe == p)) continue;            // :194-195: my own id is a no-op
        const size_t o = (size_t)p * v.G + g;
        v.exploded[o] = 0; v.queued[o] = 0;                                // cancel(), kickoff() clears `exploded`
        v.deadline[o] = now + v.tmin + (uint64_t)draw[o] % (v.tmax - v.tmin + 1);   // random_range(min..=max)
    }
}

// get_event (:134-160), drained
__global__ __launch_bounds__(256) void hb_poll_kernel(const HbView v, uint64_t now, uint8_t *__restrict__ timeouts,
                                                      uint8_t *__restrict__ send_ticked) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    for (uint32_t p = 0; p < v.R; p++) {
        const size_t o = (size_t)p * v.G + g;
        uint8_t t = 0;
        if (p != v.me) {
            const uint64_t d = v.deadline[o];
            uint8_t q = v.queued[o], ex = v.exploded[o];
            if (d != 0 && now >= d) { v.deadline[o] = 0; ex = 1; q = 1; v.exploded[o] = 1; }   // the timer task fires
            if (q) { q = 0; t = ex; }                                       // :136-141: dropped if re-armed since
            v.queued[o] = q;
        }
        timeouts[o] = t;
    }
    uint8_t s = 0;
    if (v.is_sending[g] && now >= v.next_tick[g]) {                        // :154-156, MissedTickBehavior::Skip
        s = 1;
        const uint64_t t0 = v.tick_start[g];
        v.next_tick[g] = t0 + ((now - t0) / v.period + 1) * v.period;
    }
    send_ticked[g] = s;
}

__global__ __launch_bounds__(256) void hb_set_sending_kernel(const HbView v, const uint8_t *__restrict__ sending) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g < v.G && sending[g] != HB_NONE) v.is_sending[g] = sending[g] ? 1 : 0;   // :130-132
}

// clear_reply_cnts (:223-241)
__global__ __launch_bounds__(256) void hb_clear_kernel(const HbView v, const uint8_t *__restrict__ peer) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    const uint32_t pe = peer[g];
    if (pe == HB_NONE) return;
    for (uint32_t p = 0; p < v.R; p++)
        if (p != v.me && (pe == HB_ALL || pe == p)) { const size_t o = (size_t)p * v.G + g; v.cnt0[o] = 1; v.cnt1[o] = 0; v.rep[o] = 0; }
}

// update_bcast_cnts (:247-281)
__global__ __launch_bounds__(256) void hb_bcast_kernel(const HbView v, const uint8_t *__restrict__ flags, uint8_t *__restrict__ peer_death) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    uint8_t death = 0;
    if (flags[g]) {
        const uint8_t thresh = (uint8_t)(v.tmin / v.period);               // :262-264 `as u8`
        uint8_t alive = v.alive[g];
        const uint8_t alive0 = alive;
        for (uint32_t p = 0; p < v.R; p++) {
            if (p == v.me) continue;
            const size_t o = (size_t)p * v.G + g;
            const uint64_t c0 = v.cnt0[o], c1 = v.cnt1[o];
            if (c0 > c1) { v.cnt1[o] = c0; v.rep[o] = 0; }                 // :251-255
            else {
                uint8_t r = (uint8_t)(v.rep[o] + 1);                       // :259
                if (r > thresh) {                                          // :266-276
                    if ((alive >> p) & 1) { alive &= (uint8_t)~(1u << p); death = 1; }
                    r = 0;
                }
                v.rep[o] = r;
            }
        }
        if (alive != alive0) v.alive[g] = alive;
    }
    peer_death[g] = death;
}

// update_heard_cnt (:285-300)
__global__ __launch_bounds__(256) void hb_heard_kernel(const HbView v, const uint8_t *__restrict__ peer) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    const uint32_t p = peer[g];
    if (p == HB_NONE || p >= v.R || p == v.me) return;
    v.cnt0[(size_t)p * v.G + g] += 1;
    const uint8_t a = v.alive[g];
    if (!((a >> p) & 1)) v.alive[g] = (uint8_t)(a | (1u << p));
}

}  // namespace smr

using namespace smr;

struct smr_hb { HbView v; Arena arena; };

#define HB_GRID(h) dim3(((h)->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream

extern "C" {

int smr_hb_create(const smr_hb_cfg *cfg, uint64_t now_ms, smr_hb **out) {
    if (!cfg || !out) return fail(SMR_ERR_ARG, "heartbeater: null argument");
    if (cfg->n_groups == 0 || cfg->population == 0 || cfg->population > HB_MAXR || cfg->replica_id >= cfg->population)
        return fail(SMR_ERR_ARG, "heartbeater: bad population / replica id");
    if (cfg->hear_timeout_min_ms < 100) return fail(SMR_ERR_ARG, "invalid heartbeat min hear_timeout");              // :69-74
    if (cfg->hear_timeout_max_ms < cfg->hear_timeout_min_ms + 100)
        return fail(SMR_ERR_ARG, "heartbeat max hear_timeout must be >= 100ms + min hear_timeout");                  // :75-81
    if (cfg->send_interval_ms < 1 || cfg->send_interval_ms > cfg->hear_timeout_max_ms)
        return fail(SMR_ERR_ARG, "invalid heartbeat send_interval");                                                 // :82-89
    smr_hb *h = new smr_hb();
    HbView &v = h->v;
    v.G = cfg->n_groups; v.R = cfg->population; v.me = cfg->replica_id;
    v.tmin = cfg->hear_timeout_min_ms; v.tmax = cfg->hear_timeout_max_ms; v.period = cfg->send_interval_ms;
    const size_t G = v.G, RG = (size_t)v.R * G;
    Arena &a = h->arena;
    const size_t o_dl = a.reserve(RG * 8), o_ex = a.reserve(RG), o_q = a.reserve(RG), o_s = a.reserve(G), o_ts = a.reserve(G * 8),
                 o_nt = a.reserve(G * 8), o_c0 = a.reserve(RG * 8), o_c1 = a.reserve(RG * 8), o_rp = a.reserve(RG), o_al = a.reserve(G);
    a.size = a.used;
    hipError_t e = hipMalloc((void **)&a.base, a.size);
    if (e == hipSuccess) e = hipMemset(a.base, 0, a.size);
    if (e != hipSuccess) { delete h; return fail(SMR_ERR_DEVICE, std::string("heartbeater: ") + hipGetErrorString(e)); }
    v.deadline = a.at<uint64_t>(o_dl); v.exploded = a.at<uint8_t>(o_ex); v.queued = a.at<uint8_t>(o_q); v.is_sending = a.at<uint8_t>(o_s);
    v.tick_start = a.at<uint64_t>(o_ts); v.next_tick = a.at<uint64_t>(o_nt); v.cnt0 = a.at<uint64_t>(o_c0); v.cnt1 = a.at<uint64_t>(o_c1);
    v.rep = a.at<uint8_t>(o_rp); v.alive = a.at<uint8_t>(o_al);
    void *stream = nullptr;
    hipLaunchKernelGGL(hb_init_kernel, HB_GRID(h), h->v, now_ms);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(a.base); delete h; return fail(SMR_ERR_DEVICE, std::string("heartbeater: ") + hipGetErrorString(e)); }
    *out = h;
    return SMR_OK;
}

void smr_hb_destroy(smr_hb *h) {
    if (!h) return;
    (void)hipDeviceSynchronize();
    if (h->arena.base) (void)hipFree(h->arena.base);
    delete h;
}

int smr_hb_set_sending(smr_hb *h, const uint8_t *sending_dev, void *stream) {
    if (!h || !sending_dev) return fail(SMR_ERR_ARG, "heartbeater: null argument");
    hipLaunchKernelGGL(hb_set_sending_kernel, HB_GRID(h), h->v, sending_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_hb_kickoff_hear_timer(smr_hb *h, const uint8_t *peer_dev, uint64_t now_ms, const uint32_t *draw_dev, void *stream) {
    if (!h || !peer_dev || !draw_dev) return fail(SMR_ERR_ARG, "heartbeater: null argument");
    hipLaunchKernelGGL(hb_kickoff_kernel, HB_GRID(h), h->v, peer_dev, now_ms, draw_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_hb_poll(smr_hb *h, uint64_t now_ms, uint8_t *timeouts_dev, uint8_t *send_ticked_dev, void *stream) {
    if (!h || !timeouts_dev || !send_ticked_dev) return fail(SMR_ERR_ARG, "heartbeater: null argument");
    hipLaunchKernelGGL(hb_poll_kernel, HB_GRID(h), h->v, now_ms, timeouts_dev, send_ticked_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_hb_clear_reply_cnts(smr_hb *h, const uint8_t *peer_dev, void *stream) {
    if (!h || !peer_dev) return fail(SMR_ERR_ARG, "heartbeater: null argument");
    hipLaunchKernelGGL(hb_clear_kernel, HB_GRID(h), h->v, peer_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_hb_update_bcast_cnts(smr_hb *h, const uint8_t *flags_dev, uint8_t *peer_death_dev, void *stream) {
    if (!h || !flags_dev || !peer_death_dev) return fail(SMR_ERR_ARG, "heartbeater: null argument");
    hipLaunchKernelGGL(hb_bcast_kernel, HB_GRID(h), h->v, flags_dev, peer_death_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_hb_update_heard_cnt(smr_hb *h, const uint8_t *peer_dev, void *stream) {
    if (!h || !peer_dev) return fail(SMR_ERR_ARG, "heartbeater: null argument");
    hipLaunchKernelGGL(hb_heard_kernel, HB_GRID(h), h->v, peer_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_hb_dump(smr_hb *h, uint64_t *deadline, uint8_t *exploded, uint8_t *is_sending, uint64_t *next_tick, uint64_t *cnt0,
                uint64_t *cnt1, uint8_t *rep, uint8_t *alive) {
    if (!h || !deadline || !exploded || !is_sending || !next_tick || !cnt0 || !cnt1 || !rep || !alive)
        return fail(SMR_ERR_ARG, "heartbeater: null argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const HbView &v = h->v;
    const size_t G = v.G, RG = (size_t)v.R * G;
    SMR_HIP_TRY(hipMemcpy(deadline, v.deadline, RG * 8, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(exploded, v.exploded, RG, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(is_sending, v.is_sending, G, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(next_tick, v.next_tick, G * 8, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(cnt0, v.cnt0, RG * 8, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(cnt1, v.cnt1, RG * 8, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(rep, v.rep, RG, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(alive, v.alive, G, hipMemcpyDeviceToHost));
    for (size_t g = 0; g < G; g++) cnt0[(size_t)v.me * G + g] = 0;        // my own entry does not exist in the reference's maps
    return SMR_OK;
}

}  // extern "C"

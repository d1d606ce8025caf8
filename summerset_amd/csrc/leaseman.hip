// Batched `LeaseManager` (src/server/leaseman.rs:132-935) for G groups, one replica id per object,
// lane = group.  The reference keeps four maps of tokio timers per replica (guards_sent / promises_sent on the grantor
// side, guards_held / promises_held on the holder side) and a task that takes one notice at a time from a channel; here a
// (peer, group) pair is one phase byte and two deadlines, time is an argument, and one launch takes one notice per group:
// first the timers that exploded up to now_ms are delivered in deadline order (their timeout notices were sent into the
// channel before the notice arriving now), then the notice goes through run()'s lease-number filter (:840-926) and its
// handler (:385-788).  What comes out is the action list get_action() (:275-290) would have drained, packed.
//
// What the maps hold that this layout does not: the timers' captured lease number (every live timer carries active_num:
// a higher number drops all four maps, :887-906) and their `exploded` flag (only read between a timer firing and its
// timeout notice being handled -- here both happen at the top of the same launch).  promises_sent's `revoking` flag is
// never set by the reference, so the branches testing it (:306, :681) are always taken as `false`.
//
// State arrays are [R][G] / [G]; a launch reads and writes 17 R + 9 bytes per group plus the notice (24 B) and the
// actions it emits (24 B each), every access one contiguous request per wavefront.
#include <string.h>

#include "smr_common.h"

namespace smr {

constexpr uint32_t LM_MAXR = 8;
constexpr uint8_t LM_GS = 1, LM_GH = 2, LM_PS = 4, LM_PH = 8;          // phase bits of a (peer, group)

struct LmView {
    uint32_t G, R, me;
    uint64_t expire;
    uint64_t *active;               // [G] active_num
    uint8_t *phase;                 // [R][G]
    uint64_t *grant_dl, *hold_dl;   // [R][G] promises_sent timer / guards_held-or-promises_held timer (never both, :512-516)
    uint8_t *mark;                  // [G] refresh_mark
};

struct LmOut {
    uint8_t *n; uint64_t *num, *meta, *bar;
    uint32_t G, cnt;
    __device__ void put(uint32_t g, uint64_t lnum, uint32_t kind, uint32_t peer, uint32_t mask, uint32_t msg, uint32_t flag, uint64_t b) {
        if (cnt >= SMR_LEASE_ACT_CAP) return;
        const size_t o = (size_t)cnt * G + g;
        num[o] = lnum;
        meta[o] = (uint64_t)kind | ((uint64_t)peer << 8) | ((uint64_t)mask << 16) | ((uint64_t)msg << 24) | ((uint64_t)flag << 32);
        bar[o] = b;
        cnt++;
    }
};

__global__ __launch_bounds__(256) void lm_step_kernel(const LmView v, uint64_t now, const uint64_t *__restrict__ in_num,
                                                      const uint64_t *__restrict__ in_meta, const uint64_t *__restrict__ in_bar,
                                                      LmOut out) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    const uint32_t R = v.R, me = v.me;
    const size_t G = v.G;
    // everything about this group, loaded up front
    uint8_t ph[LM_MAXR]; uint64_t gd[LM_MAXR], hd[LM_MAXR];
#pragma unroll
    for (uint32_t p = 0; p < LM_MAXR; p++) {
        const bool in = p < R;
        ph[p] = in ? v.phase[p * G + g] : 0; gd[p] = in ? v.grant_dl[p * G + g] : 0; hd[p] = in ? v.hold_dl[p * G + g] : 0;
    }
    uint64_t active = v.active[g];
    uint8_t mark = v.mark[g];
    const uint64_t meta = in_meta ? in_meta[g] : 0;
    const uint64_t lnum = in_meta ? in_num[g] : 0, nbar = in_meta ? in_bar[g] : 0;
    out.cnt = 0;

    // timers up to now, earliest first (ties: lower peer, grantor side first)
    for (uint32_t it = 0; it < 2 * LM_MAXR; it++) {
        uint64_t best = ~0ull; uint32_t bp = 0, side = 0;
#pragma unroll
        for (uint32_t p = 0; p < LM_MAXR; p++) {
            if ((ph[p] & LM_PS) && gd[p] != 0 && gd[p] <= now && gd[p] < best) { best = gd[p]; bp = p; side = 0; }
            if ((ph[p] & (LM_GH | LM_PH)) && hd[p] != 0 && hd[p] <= now && hd[p] < best) { best = hd[p]; bp = p; side = 1; }
        }
        if (best == ~0ull) break;
#pragma unroll
        for (uint32_t p = 0; p < LM_MAXR; p++)
            if (p == bp) {
                if (side == 0) { ph[p] &= (uint8_t)~(LM_GS | LM_PS); gd[p] = 0; }       // handle_grant_timeout :752-767
                else { ph[p] &= (uint8_t)~(LM_GH | LM_PH); hd[p] = 0; }                 // handle_lease_timeout :770-788
            }
        out.put(g, active, side == 0 ? SMR_LEASE_A_GRANT_TIMEOUT : SMR_LEASE_A_LEASE_TIMEOUT, bp, 0, 0, 0, 0);
    }

    const uint32_t kind = (uint32_t)(meta & 0xFF), pe = (uint32_t)((meta >> 8) & 0xFF), peers = (uint32_t)((meta >> 16) & 0xFF),
                   msg = (uint32_t)((meta >> 24) & 0xFF), held = (uint32_t)((meta >> 32) & 1), has_bar = (uint32_t)((meta >> 40) & 1);
    bool handle = kind != SMR_LEASE_N_NONE;
    if (handle && lnum < active) {                                     // :843-877
        if (kind == SMR_LEASE_N_RECV_MSG && msg == SMR_LEASE_M_REVOKE) out.put(g, lnum, SMR_LEASE_A_SEND, pe, 0, SMR_LEASE_M_REVOKE_REPLY, 0, 0);
        handle = false;
    }
    if (handle && lnum > active) {                                     // :880-915
#pragma unroll
        for (uint32_t p = 0; p < LM_MAXR; p++) { ph[p] = 0; gd[p] = 0; hd[p] = 0; }
        active = lnum;
        out.put(g, lnum, SMR_LEASE_A_HIGHER_NUMBER, 0, 0, 0, 0, 0);
    }
    if (handle) {
        const uint32_t all = (1u << R) - 1u;
        const uint32_t sel = (peers == SMR_LEASE_ALL ? all : (peers & all)) & ~(1u << me);
        if (kind == SMR_LEASE_N_NEW_GRANTS) {                          // :385-439
            uint32_t bc = 0;
#pragma unroll
            for (uint32_t p = 0; p < LM_MAXR; p++)
                if (((sel >> p) & 1) && !(ph[p] & LM_PS)) { ph[p] |= LM_GS; bc |= 1u << p; }
            out.put(g, lnum, SMR_LEASE_A_BCAST, 0, bc, SMR_LEASE_M_GUARD, has_bar, has_bar ? nbar : 0);
        } else if (kind == SMR_LEASE_N_DO_REVOKE) {                    // :442-481
            uint32_t bc = 0;
#pragma unroll
            for (uint32_t p = 0; p < LM_MAXR; p++)
                if ((sel >> p) & 1) { ph[p] &= (uint8_t)~LM_GS; if (ph[p] & LM_PS) bc |= 1u << p; }
            if (bc) out.put(g, lnum, SMR_LEASE_A_BCAST, 0, bc, SMR_LEASE_M_REVOKE, 0, 0);
        } else if (kind == SMR_LEASE_N_CLEAR_HELD) {                   // :484-498
#pragma unroll
            for (uint32_t p = 0; p < LM_MAXR; p++) { ph[p] &= (uint8_t)~(LM_GH | LM_PH); hd[p] = 0; }
            out.put(g, lnum, SMR_LEASE_A_LEASE_CLEARED, 0, 0, 0, 0, 0);
        } else if (kind == SMR_LEASE_N_RECV_MSG && pe < R && pe != me) {
            // the one (peer, group) the message is about, pulled out of the register arrays and put back
            uint8_t s = 0; uint64_t gdl = 0, hdl = 0;
#pragma unroll
            for (uint32_t p = 0; p < LM_MAXR; p++) if (p == pe) { s = ph[p]; gdl = gd[p]; hdl = hd[p]; }
            switch (msg) {
            case SMR_LEASE_M_GUARD:                                    // :501-555
                if (s & LM_PH) break;
                s |= LM_GH; hdl = now + v.expire;
                if (has_bar) out.put(g, lnum, SMR_LEASE_A_GUARD_ACCEPT_BAR, pe, 0, 0, 1, nbar);
                out.put(g, lnum, SMR_LEASE_A_SEND, pe, 0, SMR_LEASE_M_GUARD_REPLY, 0, 0);
                break;
            case SMR_LEASE_M_GUARD_REPLY:                              // :558-592
                if (!(s & LM_GS)) break;
                s = (uint8_t)((s & ~LM_GS) | LM_PS); gdl = now + 2 * v.expire;
                out.put(g, lnum, SMR_LEASE_A_SEND, pe, 0, SMR_LEASE_M_PROMISE, 0, 0);
                break;
            case SMR_LEASE_M_PROMISE:                                  // :595-645
                if (s & (LM_GH | LM_PH)) {
                    s = (uint8_t)((s & ~LM_GH) | LM_PH); hdl = now + v.expire;
                    out.put(g, lnum, SMR_LEASE_A_SEND, pe, 0, SMR_LEASE_M_PROMISE_REPLY, 1, 0);
                } else out.put(g, lnum, SMR_LEASE_A_SEND, pe, 0, SMR_LEASE_M_PROMISE_REPLY, 0, 0);
                break;
            case SMR_LEASE_M_PROMISE_REPLY:                            // :648-693
                if (!(s & LM_PS)) break;
                if (!held) { s &= (uint8_t)~(LM_GS | LM_PS); gdl = 0; out.put(g, lnum, SMR_LEASE_A_GRANT_REMOVED, pe, 0, 0, 0, 0); break; }
                gdl = now + v.expire;
                out.put(g, lnum, SMR_LEASE_A_NEXT_REFRESH, pe, 0, 0, 0, 0);
                mark |= (uint8_t)(1u << pe);                            // get_action :281-284
                break;
            case SMR_LEASE_M_REVOKE: {                                 // :696-725
                const uint32_t h = (s & LM_PH) ? 1 : 0;
                s &= (uint8_t)~(LM_GH | LM_PH); hdl = 0;
                out.put(g, lnum, SMR_LEASE_A_SEND, pe, 0, SMR_LEASE_M_REVOKE_REPLY, h, 0);
                break; }
            case SMR_LEASE_M_REVOKE_REPLY:                             // :728-749
                s &= (uint8_t)~(LM_GS | LM_PS); gdl = 0;
                out.put(g, lnum, SMR_LEASE_A_GRANT_REMOVED, pe, 0, 0, held, 0);
                break;
            default: break;
            }
#pragma unroll
            for (uint32_t p = 0; p < LM_MAXR; p++) if (p == pe) { ph[p] = s; gd[p] = gdl; hd[p] = hdl; }
        }
    }
#pragma unroll
    for (uint32_t p = 0; p < LM_MAXR; p++)
        if (p < R) { v.phase[p * G + g] = ph[p]; v.grant_dl[p * G + g] = gd[p]; v.hold_dl[p * G + g] = hd[p]; }
    v.active[g] = active; v.mark[g] = mark;
    out.n[g] = (uint8_t)out.cnt;
}

// attempt_refresh (:296-317); timer.extend (utils/timer.rs:94-115): beyond the deadline, or beyond now if that is past
__global__ __launch_bounds__(256) void lm_refresh_kernel(const LmView v, uint64_t now, const uint8_t *__restrict__ call,
                                                         const uint8_t *__restrict__ peers, uint8_t *__restrict__ to_refresh) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    uint32_t out = 0;
    if (call[g]) {
        uint32_t mark = v.mark[g];
        const uint32_t sel = peers[g] == SMR_LEASE_ALL ? 0xFFu : peers[g];
        for (uint32_t p = 0; p < v.R; p++) {
            const size_t o = (size_t)p * v.G + g;
            if (p == v.me || !((sel >> p) & 1) || !((mark >> p) & 1) || !(v.phase[o] & LM_PS)) continue;
            mark &= ~(1u << p);
            uint64_t d = v.grant_dl[o];
            if (d < now) d = now;
            v.grant_dl[o] = d + v.expire;
            out |= 1u << p;
        }
        v.mark[g] = (uint8_t)mark;
    }
    to_refresh[g] = (uint8_t)out;
}

// grant_set (:236-243), lease_set (:246-253), lease_cnt (:257-259) for every group
__global__ __launch_bounds__(256) void lm_sets_kernel(const LmView v, uint8_t *__restrict__ grant_set, uint8_t *__restrict__ lease_set,
                                                      uint8_t *__restrict__ lease_cnt) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    uint32_t gs = 0, ls = 0;
    for (uint32_t p = 0; p < v.R; p++) {
        const uint8_t s = v.phase[(size_t)p * v.G + g];
        gs |= ((s & LM_PS) ? 1u : 0u) << p; ls |= ((s & LM_PH) ? 1u : 0u) << p;
    }
    if (grant_set) grant_set[g] = (uint8_t)gs;
    if (lease_set) lease_set[g] = (uint8_t)ls;
    if (lease_cnt) lease_cnt[g] = (uint8_t)(1 + __popc(ls));
}

}  // namespace smr

using namespace smr;

struct smr_lease { LmView v; Arena arena; };

#define LM_GRID(h) dim3(((h)->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream

extern "C" {

int smr_lease_create(const smr_lease_cfg *cfg, smr_lease **out) {
    if (!cfg || !out) return fail(SMR_ERR_ARG, "lease manager: null argument");
    if (cfg->n_groups == 0 || cfg->population == 0 || cfg->population > LM_MAXR || cfg->replica_id >= cfg->population)
        return fail(SMR_ERR_ARG, "lease manager: bad population / replica id");
    if (cfg->expire_timeout_ms < 100 || cfg->expire_timeout_ms > 10000)
        return fail(SMR_ERR_ARG, "invalid lease expire_timeout");                                       // :178-185
    if (2 * cfg->hb_send_interval_ms >= cfg->expire_timeout_ms)
        return fail(SMR_ERR_ARG, "heartbeat interval too long for lease expire_timeout");              // :186-193
    smr_lease *h = new smr_lease();
    LmView &v = h->v;
    v.G = cfg->n_groups; v.R = cfg->population; v.me = cfg->replica_id; v.expire = cfg->expire_timeout_ms;
    const size_t G = v.G, RG = (size_t)v.R * G;
    Arena &a = h->arena;
    const size_t o_ac = a.reserve(G * 8), o_ph = a.reserve(RG), o_gd = a.reserve(RG * 8), o_hd = a.reserve(RG * 8), o_mk = a.reserve(G);
    a.size = a.used;
    hipError_t e = hipMalloc((void **)&a.base, a.size);
    if (e == hipSuccess) e = hipMemset(a.base, 0, a.size);
    if (e != hipSuccess) { delete h; return fail(SMR_ERR_DEVICE, std::string("lease manager: ") + hipGetErrorString(e)); }
    v.active = a.at<uint64_t>(o_ac); v.phase = a.at<uint8_t>(o_ph); v.grant_dl = a.at<uint64_t>(o_gd); v.hold_dl = a.at<uint64_t>(o_hd);
    v.mark = a.at<uint8_t>(o_mk);
    *out = h;
    return SMR_OK;
}

void smr_lease_destroy(smr_lease *h) {
    if (!h) return;
    (void)hipDeviceSynchronize();
    if (h->arena.base) (void)hipFree(h->arena.base);
    delete h;
}

int smr_lease_step(smr_lease *h, uint64_t now_ms, const uint64_t *num_dev, const uint64_t *meta_dev, const uint64_t *bar_dev,
                   uint8_t *act_n_dev, uint64_t *act_num_dev, uint64_t *act_meta_dev, uint64_t *act_bar_dev, void *stream) {
    if (!h || !act_n_dev || !act_num_dev || !act_meta_dev || !act_bar_dev) return fail(SMR_ERR_ARG, "lease manager: null argument");
    if (meta_dev && (!num_dev || !bar_dev)) return fail(SMR_ERR_ARG, "lease manager: a notice batch needs num, meta and bar");
    LmOut o; o.n = act_n_dev; o.num = act_num_dev; o.meta = act_meta_dev; o.bar = act_bar_dev; o.G = h->v.G; o.cnt = 0;
    hipLaunchKernelGGL(lm_step_kernel, LM_GRID(h), h->v, now_ms, num_dev, meta_dev, bar_dev, o);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_lease_attempt_refresh(smr_lease *h, uint64_t now_ms, const uint8_t *call_dev, const uint8_t *peers_dev, uint8_t *to_refresh_dev,
                              void *stream) {
    if (!h || !call_dev || !peers_dev || !to_refresh_dev) return fail(SMR_ERR_ARG, "lease manager: null argument");
    hipLaunchKernelGGL(lm_refresh_kernel, LM_GRID(h), h->v, now_ms, call_dev, peers_dev, to_refresh_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_lease_sets(smr_lease *h, uint8_t *grant_set_dev, uint8_t *lease_set_dev, uint8_t *lease_cnt_dev, void *stream) {
    if (!h) return fail(SMR_ERR_ARG, "lease manager: null argument");
    hipLaunchKernelGGL(lm_sets_kernel, LM_GRID(h), h->v, grant_set_dev, lease_set_dev, lease_cnt_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_lease_dump(smr_lease *h, uint64_t *active_num, uint8_t *phase, uint64_t *grant_deadline, uint64_t *hold_deadline, uint8_t *refresh_mark) {
    if (!h || !active_num || !phase || !grant_deadline || !hold_deadline || !refresh_mark) return fail(SMR_ERR_ARG, "lease manager: null argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const LmView &v = h->v;
    const size_t G = v.G, RG = (size_t)v.R * G;
    SMR_HIP_TRY(hipMemcpy(active_num, v.active, G * 8, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(phase, v.phase, RG, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(grant_deadline, v.grant_dl, RG * 8, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(hold_deadline, v.hold_dl, RG * 8, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(refresh_mark, v.mark, G, hipMemcpyDeviceToHost));
    return SMR_OK;
}

}  // extern "C"

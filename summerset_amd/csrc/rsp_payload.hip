// RSPaxos payload store: the shard BYTES behind an RSPaxos replica's instances, resident in HBM, keyed by (slot, shard).
//
// The replica engine (rsp_engine.hip) keeps a codeword as (batch token, mask of shards present); this object keeps what
// the reference keeps in `inst.reqs_cw` and `inst.voted.1` (rspaxos/mod.rs:168-233): the shards themselves.  Two planes
// per replica -- REQS and VOTED -- each a ring of W rows x n shards x G groups x cap_sl bytes, plus per (row, group) the
// token whose bytes the row holds, the shards present and the data length; a VOTED shard that equals the REQS row's (the
// vote is a clone of reqs_cw, messages.rs:373-380) is an ALIAS of it -- a bit per shard, no second copy (ps_plan_body).
// The engine decides which shards exist where; the store makes the bytes FOLLOW that decision:
//   smr_rsp_pstore_put     the leader's RSCodeword::from_data + compute_parity of a tick's batches (request.rs:71-101,
//                          rscoding.rs:165-243,447-486) into the rows the engine's handle_req_batch put them
//   smr_rsp_pstore_follow  after any handler call: for every ring cell, take the shards the engine's mask has and the row
//                          has not from the given sources where they hold the same token (subset_copy at the sender +
//                          `inst.reqs_cw = reqs_cw` / absorb_other at the receiver: rscoding.rs:255-346, messages.rs:180-194,
//                          373-380, 547-560), then rebuild what is still missing from any d present shards
//                          (reconstruct_data on commit / at the prepare quorum, compute_parity for the re-Accepts:
//                          durability.rs:140-160, messages.rs:227-259; ReedSolomon::reconstruct's first-d-present rule)
//   smr_rsp_pstore_get_data  the serialized batch of an instance for execution (rscoding.rs:583-609)
// Payload identity is the token: the bytes of a shard are a function of (token, shard index), so a row whose token
// changed drops its shards, and a source is usable for a row exactly when it holds the row's token.  Token 0 is the
// empty batch `ReqBatch::new()` (messages.rs:246-252): its serialization is the single byte 0x00 and is synthesised.
//
// HBM-bound byte work, rare path except `put` and one shard copy per follower and tick: a plan kernel (one lane per ring
// cell) compares the engine's masks with the rows' and compacts the cells with work onto a list; a byte kernel walks the
// list, one wavefront per cell, a lane per 16-byte column: copies are 16-byte loads / stores, rebuilds a per-lane
// coefficient x 16-byte GF(2^8) multiply-accumulate (bit-sliced xtime on 4 packed bytes per VGPR, poly 0x11D) against
// a 256-pattern table of (all shards from the first d present) matrices.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "smr_common.h"
#include "rsp_peek.h"
#include "raft_peek.h"

namespace smr {

constexpr uint32_t PS_NULL = 0xFFFFFFFFu;
constexpr uint32_t PS_NONE = 0xFF, PS_OWN = 0xFE, PS_EMPTY = 0xFD;     // a shard's source: none / my other plane / the empty batch
constexpr uint32_t PS_MAX_SRC = 16, PS_MAX_N = 8;
constexpr uint64_t PS_NO_SRC = ~0ull;

struct PsPlane {
    uint8_t *bytes;        // [(row * n + k) * G + g] * cap_sl
    uint32_t *tok;         // [W][G] the token whose bytes the row holds (PS_NULL: nothing)
    uint8_t *avail;        // [W][G] shards present
    uint32_t *dlen;        // [W][G] data length of the codeword
    uint8_t *alias;        // the VOTED plane of a two-plane store: [W][G] shards of the cell that LIVE IN THE REQS PLANE's row (below);
    uint8_t *base0;        // that plane's bytes.  NULL / unused for every other plane
};
// where shard k of cell i (= row * G + g) of plane p is read from
__device__ __forceinline__ const uint8_t *ps_rd(const PsPlane &p, size_t i, uint32_t k) {
    return (p.alias && ((p.alias[i] >> k) & 1u)) ? p.base0 : p.bytes;
}
struct PsView {
    uint32_t G, W, Wmask, n, d, cap_sl;
    PsPlane pl[2];
    const uint8_t *mat;    // [256][8][8]: shard r = XOR_c mat[pat][r][c] * (c-th present shard of pat)
    uint32_t *it_n;        // the list of cells with work: count (it_n[flip]; the plan kernel zeroes it_n[flip ^ 1] for the next
    uint32_t flip;         // call, so no memset launch sits in front of it), then per item the cell, per plane the shards' sources
    uint32_t *it_cell;     // (8 bits each), the shards to rebuild | the pattern to rebuild from, and the shard length
    uint64_t *it_src[2];
    uint32_t *it_rc;
    uint8_t *it_mat;       // VOTED shards to move out of the REQS row (their alias ends) before anything else touches the cell
    uint32_t *it_sl[2];
    unsigned long long *counters;   // 0 shards copied, 1 shards rebuilt, 2 shards the engine has and nobody could give, 3 rows re-keyed,
                                    // 4 of the copied: written by the sender's put launch (dlv)
    uint8_t *dlv;          // [W][G] REQS shards whose bytes the sender's put launch has written into this store's row already (ps_put_deliver_kernel):
                           // the plan of the same call counts them as copied and lists no copy for them; zero between calls
};
constexpr uint32_t PS_VIA0 = 0x40;  // a source code's flag: the shard is an alias in that source -- read it from the source STORE's reqs plane
struct PsSrcs {
    uint32_t n;
    PsPlane p[PS_MAX_SRC];              // (the plan kernel's: tokens, masks, lengths, alias bits)
    const uint8_t *rd[2 * PS_MAX_SRC];  // the byte kernel's: [2 j] = source j's plane, [2 j + 1] = its store's reqs plane (one indexed load per
                                        // shard: a choice between two fields of p[j] by a per-lane j put the whole table into scratch, 800 B)
    __host__ void set(uint32_t j, const PsPlane &pl) { p[j] = pl; rd[2 * j] = pl.bytes; rd[2 * j + 1] = pl.alias ? pl.base0 : pl.bytes; }
};
__device__ __forceinline__ const uint8_t *ps_src(const PsSrcs &S, uint32_t code) { return S.rd[((code & 15u) << 1) | ((code >> 6) & 1u)]; }

typedef uint32_t ps_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ size_t ps_off(const PsView &v, uint32_t row, uint32_t k, uint32_t g) {
    return (((size_t)row * v.n + k) * v.G + g) * v.cap_sl;
}
// t = q * d + r for a lane index t: 32-bit whenever t fits (it nearly always does -- the emulated 64-bit division is ~150
// instructions, and it sat at the top of every lane of the byte kernels)
__device__ __forceinline__ void ps_divmod(uint64_t t, uint32_t d, uint32_t &q, uint32_t &r) {
    if (t >> 32) { q = (uint32_t)(t / d); r = (uint32_t)(t % d); }
    else { const uint32_t t32 = (uint32_t)t; q = t32 / d; r = t32 - q * d; }
}
__device__ __forceinline__ uint32_t ps_shard_len(uint32_t L, uint32_t d) { return (L + d - 1) / d; }   // rscoding.rs:177-181
__device__ __forceinline__ ps_u32x4 ps_load16(const uint8_t *p) {
    ps_u32x4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
__device__ __forceinline__ void ps_store16(uint8_t *p, ps_u32x4 v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ uint32_t ps_xtime4(uint32_t x) {               // 4 packed GF(2^8) bytes times 2
    const uint32_t hi = x & 0x80808080u;
    return ((x << 1) & 0xFEFEFEFEu) ^ ((hi - (hi >> 7)) & 0x1D1D1D1Du);
}
// the shards in `need` of cell (row, g), columns [c0, c0 + 16), from the first d shards of `pat` (all in `base`'s plane).
// Per output row a Horner scheme over the coefficient BIT PLANES, as rs_kernels.hip evaluates its products:
//     out_r = XOR_b 2^b * (XOR_{c : bit b of m[r][c]} in_c) = (..(P_hb * 2 ^ P_hb-1) * 2 ^ ..) * 2 ^ P_0
// -- (highest coefficient bit) doublings per row instead of 8 per coefficient (RS(3,2) parity: 0 and 3).  The coefficients are
// per-lane values (a lane's own erasure pattern); in ps_put_kernel they are wave-uniform and the plane tests scalar.
// (in[c] = the 16 columns of the c-th present shard of `pat`, already in registers: put has just loaded them from the batch --
// reading them back from the row it stored them to put a store -> load round trip in front of the parity: 60 -> 46 us, profiles/r7d)
template <int D, typename F>
__device__ __forceinline__ void ps_rebuild_emit(const PsView &v, uint32_t need, uint32_t pat, const ps_u32x4 (&in)[D], F emit) {
    const uint8_t *m = v.mat + (size_t)pat * 64;
#pragma unroll
    for (int r = 0; r < (int)PS_MAX_N; r++) {
        if (!((need >> r) & 1u)) continue;
        uint32_t co[D], any = 0;
#pragma unroll
        for (int c = 0; c < D; c++) { co[c] = m[r * 8 + c]; any |= co[c]; }
        ps_u32x4 acc = {0u, 0u, 0u, 0u};
        for (int b = any ? 31 - __clz((int)any) : -1; b >= 0; b--) {
            acc.x = ps_xtime4(acc.x); acc.y = ps_xtime4(acc.y); acc.z = ps_xtime4(acc.z); acc.w = ps_xtime4(acc.w);
#pragma unroll
            for (int c = 0; c < D; c++)
                if ((co[c] >> b) & 1u) acc ^= in[c];
        }
        emit((uint32_t)r, acc);
    }
}
template <int D>
__device__ __forceinline__ void ps_rebuild_from(const PsView &v, uint8_t *base, uint32_t row, uint32_t g, uint32_t c0, uint32_t need,
                                                uint32_t pat, const ps_u32x4 (&in)[D]) {
    ps_rebuild_emit<D>(v, need, pat, in, [&](uint32_t r, ps_u32x4 acc) { ps_store16(base + ps_off(v, row, r, g) + c0, acc); });
}
// the same product with nothing held across rows: per output row a walk over the d inputs (re-read, they are this lane's own
// cache lines by now), one bit-serial multiply-accumulate per coefficient.  For the byte kernel, whose common path is plain copies:
// the Horner form's 8 x 16-byte input registers would set every wavefront's register budget (196 VGPRs, two wavefronts per SIMD)
// for a path that runs at leader changes only.
// (`in0`: the present shards whose bytes are read from `base0` instead -- a VOTED row's aliased shards)
__device__ __forceinline__ void ps_rebuild_small(const PsView &v, uint8_t *base, uint32_t row, uint32_t g, uint32_t c0, uint32_t need,
                                                 uint32_t pat, uint32_t in0 = 0u, const uint8_t *base0 = nullptr) {
    const uint8_t *m = v.mat + (size_t)pat * 64;
    for (uint32_t nd = need; nd; nd &= nd - 1u) {
        const uint32_t r = (uint32_t)__ffs((int)nd) - 1u;
        ps_u32x4 acc = {0u, 0u, 0u, 0u};
        uint32_t p = pat;
        for (uint32_t c = 0; c < v.d; c++) {
            const uint32_t k = (uint32_t)__ffs((int)p) - 1u;
            p &= p - 1u;
            ps_u32x4 x = ps_load16((((in0 >> k) & 1u) ? base0 : base) + ps_off(v, row, k, g) + c0);
            for (uint32_t co = m[r * 8 + c]; co; co >>= 1) {
                if (co & 1u) acc ^= x;
                x.x = ps_xtime4(x.x); x.y = ps_xtime4(x.y); x.z = ps_xtime4(x.z); x.w = ps_xtime4(x.w);
            }
        }
        ps_store16(base + ps_off(v, row, r, g) + c0, acc);
    }
}

// 16 bytes of a serialized batch from offset `off`; bytes at or beyond `lim` read as zero (from_data's padding,
// rscoding.rs:188-189, and the next shard's bytes).  The window that straddles `lim` is still ONE 16-byte load with the excess
// masked off while it stays inside the buffer (`room` readable bytes from p): with shard_len = ceil(L / d) nearly every wavefront
// holds such a lane, and a byte-wise path there is run, divergently, by the whole wavefront (put: 78 -> 60 us, profiles/r7c).  Only the last
// bytes of the whole buffer take the byte loop.
__device__ __forceinline__ ps_u32x4 ps_load_data16(const uint8_t *p, uint32_t off, uint32_t lim, uint64_t room) {
    if (off + 16u <= lim) return ps_load16(p + off);
    const uint32_t nv = off >= lim ? 0u : lim - off;                        // < 16 bytes of the window exist
    ps_u32x4 x = {0u, 0u, 0u, 0u};
    if (nv == 0) return x;
    if ((uint64_t)off + 16u <= room) {
        x = ps_load16(p + off);
        const uint32_t m0 = nv >= 4 ? 0xFFFFFFFFu : (1u << (8 * nv)) - 1u;
        const uint32_t m1 = nv >= 8 ? 0xFFFFFFFFu : (nv > 4 ? (1u << (8 * (nv - 4))) - 1u : 0u);
        const uint32_t m2 = nv >= 12 ? 0xFFFFFFFFu : (nv > 8 ? (1u << (8 * (nv - 8))) - 1u : 0u);
        const uint32_t m3 = nv > 12 ? (1u << (8 * (nv - 12))) - 1u : 0u;
        x.x &= m0; x.y &= m1; x.z &= m2; x.w &= m3;
        return x;
    }
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t b = ((uint32_t)i < nv) ? p[off + i] : 0u;
        w[i >> 2] |= b << (8 * (i & 3));
    }
    return (ps_u32x4){w[0], w[1], w[2], w[3]};
}

// (CRaft: the token of a log entry; its comment stands in front of the CRaft calls below)
// 16 bits of the term and 14 of the slot folded over itself (round 6, ADVICE r5: 10 bits of the term let two conflicting entries
// of one slot whose terms differ by a multiple of 1024 -- a partitioned candidate bumps its term every election timeout --
// share a token, and the plan kernel then kept the stale shards).  A ring cell only ever holds slots that differ by multiples of
// the window, so the fold tells generations of a cell apart for 2^14 windows; terms for 65 536 elections.
__device__ __forceinline__ uint32_t craft_token(uint32_t slot, uint64_t term) {
    return 0x40000000u | (((uint32_t)term & 0xFFFFu) << 14) | ((slot ^ (slot >> 14)) & 0x3FFFu);
}
// the slot ring cell `row` of group g holds now (PS_NULL: none -- the dummy entry 0 carries no codeword)
__device__ __forceinline__ uint32_t craft_cell_slot(uint32_t len, uint32_t st, uint32_t rl, uint32_t W, uint32_t row) {
    if (len == 0) return PS_NULL;
    uint32_t s = ((len - 1u) & ~(W - 1u)) | row;                            // the highest slot <= len - 1 + (W - 1) in this cell ...
    if (s > len - 1u) { if (s < W) return PS_NULL; s -= W; }                 // ... brought below the log's end
    const uint32_t lo = st > rl ? st : rl;
    return (s >= lo && s != 0u) ? s : PS_NULL;
}
__device__ __forceinline__ uint32_t craft_cell_slot(const RaftPeek &e, uint32_t row, uint32_t g) {
    return craft_cell_slot(e.log_len[g], e.start_slot[g], e.ring_lo[g], e.W, row);
}
// ... and its token, for the plan kernel (e: a CRaft replica's log behind an RspPeek)
__device__ __forceinline__ uint32_t craft_want_tok(const RspPeek &e, uint32_t i) {
    const uint32_t row = i / e.G, g = i - row * e.G;
    const uint32_t s = craft_cell_slot(e.c_len[g], e.c_start[g], e.c_rlo[g], e.W, row);
    return s == PS_NULL ? PS_NULL : craft_token(s, e.c_term[i]);
}

// A REQS row about to be REPLACED (put, ingest): the VOTED shards that lived in it go with it.  (The handler behind either call has
// re-initialised the instance's vote already -- handle_req_batch votes for its own batch, request.rs:103-118, and the ring cell of a
// slot one window back is no instance of the engine's any more -- so the next follow re-derives the VOTED cell from the engine.)
__device__ __forceinline__ void ps_drop_aliased(const PsView &v, size_t i) {
    if (!v.pl[1].alias) return;
    const uint32_t a = v.pl[1].alias[i];
    if (!a) return;
    const uint32_t left = (uint32_t)v.pl[1].avail[i] & ~a;
    v.pl[1].avail[i] = (uint8_t)left;
    if (!left) { v.pl[1].tok[i] = PS_NULL; v.pl[1].dlen[i] = 0; }
    v.pl[1].alias[i] = 0;
}

// request.rs:71-101: one lane per (group, 16-byte column) of the tick's batches; D = the number of data shards
// (CRAFT: the entry a CRaft leader appended at a_slot[g] -- PS_NULL: none -- where its log holds that slot; the token is the
// entry's, a_n / a_val are not read)
// what group g puts this tick, if anything: the token, the ring row, the batch's length
template <bool CRAFT>
__device__ __forceinline__ bool ps_put_what(const PsView &v, const uint32_t *__restrict__ a_n, const uint32_t *__restrict__ a_slot,
                                            const uint32_t *__restrict__ a_val, const uint32_t *__restrict__ len, uint32_t data_len, const RaftPeek &cr,
                                            uint32_t g, uint32_t &tok, uint32_t &row, uint32_t &L) {
    const uint32_t sl_ = a_slot[g];
    if (CRAFT) {
        if (sl_ == PS_NULL || sl_ == 0u || craft_cell_slot(cr, sl_ & v.Wmask, g) != sl_) return false;   // (the log does not hold that slot)
        tok = craft_token(sl_, cr.entry_term[(size_t)(sl_ & v.Wmask) * v.G + g]);
    } else {
        if (a_n[g] == 0) return false;
        tok = a_val[g];
    }
    row = sl_ & v.Wmask;
    L = len ? len[g] : data_len;
    if (L > data_len) L = data_len;
    return true;
}

// ---- round 6: the put launch writes the FOLLOWERS' shards too (smr_*_pstore_put_follow_all) --------------------------------------
// In a steady tick a follower's follow copies ONE thing: the shards of the leader's new codeword its engine has just accepted, out of
// the leader's REQS row into its own -- bytes the put launch holds in registers a few microseconds earlier (90 MB read back and 90 MB
// written by ps_bytes_many_kernel at config 4's size, 48 us).  ps_put_deliver_kernel decides per (group, follower) which shards that
// copy WILL be -- ps_deliver_mask: exactly the `take` ps_plan_cell computes for plane 0 from source 0, the leader's row as the
// leader's own follow leaves it -- stores them beside the leader's, and leaves the mask in the follower's `dlv` array; the follower's plan
// (same call, next launch) does everything it always does -- headers, aliases, counters, the list -- except listing those copies.
// Everything else a follow can mean (rebuilds, other sources, votes that part from the reqs row) stays with the plan / byte kernels.
struct PsHdrs {                        // a follower store's REQS / VOTED headers
    const uint32_t *tok0, *tok1;
    const uint8_t *av0, *av1, *alias;  // (alias: NULL for a one-plane store)
    uint8_t *dlv, *b0;                 // its delivered-shards array, its REQS plane's bytes
};
struct PsDeliver {
    uint32_t n, all;                   // followers; (1 << shards) - 1
    PsHdrs h[PS_MAX_N];
    RspPeek e[PS_MAX_N];               // the followers' engines
    RspPeek lead;                      // the leader's: what its own follow leaves of the row just put
};
// what the engine behind `e` wants in cell (row, g), both planes, in the plan's normal form (a plane that wants nothing wants the
// null token)
__device__ __forceinline__ void ps_cell_want(const RspPeek &e, uint32_t G, uint32_t all, uint32_t row, uint32_t g, uint32_t c_len, uint32_t c_st,
                                             uint32_t c_rl, uint32_t (&w_tok)[2], uint32_t (&w_mask)[2]) {
    const uint32_t i = row * G + g;
    if (e.c_len) {                                                        // a CRaft log: the token the (slot, term) of the cell implies
        const uint32_t sl_ = craft_cell_slot(c_len, c_st, c_rl, e.W, row);
        const uint64_t tm = e.c_term[i];
        w_tok[0] = sl_ == PS_NULL ? PS_NULL : craft_token(sl_, tm);
        w_tok[1] = PS_NULL;
    } else { w_tok[0] = e.s_val[i]; w_tok[1] = e.s_vval[i]; }
    w_mask[0] = e.s_mask[i]; w_mask[1] = e.s_vmask[i];
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
        w_mask[pl] &= all;
        if (w_tok[pl] == PS_NULL) w_mask[pl] = 0;
        if (w_mask[pl] == 0) w_tok[pl] = PS_NULL;
    }
}
__device__ __forceinline__ uint32_t ps_deliver_mask(const PsHdrs &h, const RspPeek &fe, const RspPeek &le, uint32_t G, uint32_t all, uint32_t row,
                                                    uint32_t g, uint32_t tok) {
    if (tok == 0u || tok == PS_NULL) return 0u;                           // (the empty batch is synthesised, not copied)
    const uint32_t i = row * G + g;
    uint32_t lw_tok[2], lw_mask[2], w_tok[2], w_mask[2];
    uint32_t lc_len = 0, lc_st = 0, lc_rl = 0, c_len = 0, c_st = 0, c_rl = 0;
    if (le.c_len) { lc_len = le.c_len[g]; lc_st = le.c_start[g]; lc_rl = le.c_rlo[g]; }
    if (fe.c_len) { c_len = fe.c_len[g]; c_st = fe.c_start[g]; c_rl = fe.c_rlo[g]; }
    ps_cell_want(le, G, all, row, g, lc_len, lc_st, lc_rl, lw_tok, lw_mask);
    ps_cell_want(fe, G, all, row, g, c_len, c_st, c_rl, w_tok, w_mask);
    const uint32_t h_tok0 = h.tok0[i], h_tok1 = h.tok1[i], h_av0 = h.av0[i], h_av1 = h.av1[i];
    const uint32_t al = h.alias ? h.alias[i] : 0u;
    // the leader's own follow (between the put and the followers' plan): the row keeps the shards its engine wants of THIS token
    if (lw_tok[0] != tok || w_tok[0] != tok) return 0u;
    const uint32_t have0 = h_tok0 == tok ? (h_av0 & w_mask[0]) : 0u;
    const uint32_t take = w_mask[0] & ~have0 & lw_mask[0];
    // a vote that lives in this REQS row and outlives the row's token moves out FIRST (ps_plan_cell's `mat`): the byte kernel's business
    const uint32_t have1 = h_tok1 == w_tok[1] ? (h_av1 & w_mask[1]) : 0u;
    if ((al & have1) && h_tok0 != tok) return 0u;
    return take;
}

// (two halves around the deliver kernel's barrier: the batch's bytes are on their way while the masks are decided)
template <int D>
struct PsPutLane {
    uint32_t g, blk, tok, row, L, sl, c0;
    bool put, bytes;                   // the group puts this tick; this lane has a column of its shards
    ps_u32x4 in[D];
};
template <int D, bool CRAFT>
__device__ __forceinline__ void ps_put_load(PsPutLane<D> &p, const PsView &v, const uint32_t *__restrict__ a_n, const uint32_t *__restrict__ a_slot,
                                            const uint32_t *__restrict__ a_val, const uint8_t *__restrict__ data, uint64_t data_stride,
                                            const uint32_t *__restrict__ len, uint32_t data_len, uint32_t nblk, const RaftPeek &cr) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    ps_divmod(t, nblk, p.g, p.blk);
    p.put = p.bytes = false;
    p.tok = PS_NULL; p.row = p.L = p.sl = 0; p.c0 = p.blk * 16u;
    if (p.g >= v.G) return;
    p.put = ps_put_what<CRAFT>(v, a_n, a_slot, a_val, len, data_len, cr, p.g, p.tok, p.row, p.L);
    if (!p.put) return;
    p.sl = ps_shard_len(p.L, (uint32_t)D);
    p.bytes = p.c0 < p.sl;
    if (!p.bytes) return;
    const uint8_t *src = data + (size_t)p.g * data_stride;
    const uint64_t room = (uint64_t)(v.G - 1u - p.g) * data_stride + data_len;   // readable bytes from this batch's start
#pragma unroll
    for (int c = 0; c < D; c++) {
        const uint32_t end = ((uint32_t)c + 1u) * p.sl;
        p.in[c] = ps_load_data16(src, (uint32_t)c * p.sl + p.c0, end < p.L ? end : p.L, room);
    }
}
template <int D, bool DELIVER>
__device__ __forceinline__ void ps_put_store(const PsPutLane<D> &p, const PsView &v, const PsDeliver *dv, const uint8_t *sh_dm, uint32_t g_first) {
    if (!p.put) return;
    const uint32_t g = p.g, row = p.row, c0 = p.c0;
    if (p.blk == 0) {
        const size_t i = (size_t)row * v.G + g;
        ps_drop_aliased(v, i);
        v.pl[0].tok[i] = p.tok;
        v.pl[0].avail[i] = (uint8_t)((1u << v.n) - 1u);
        v.pl[0].dlen[i] = p.L;
    }
    if (!p.bytes) return;
    uint32_t dm[PS_MAX_N];                                                 // per follower: the shards to write there as well
    if (DELIVER) {
#pragma unroll
        for (int k = 0; k < (int)PS_MAX_N; k++) dm[k] = (uint32_t)k < dv->n ? sh_dm[(g - g_first) * PS_MAX_N + (uint32_t)k] : 0u;
    }
    auto emit = [&](uint32_t r, ps_u32x4 x) {
        const size_t o = ps_off(v, row, r, g) + c0;
        ps_store16(v.pl[0].bytes + o, x);
        if (DELIVER) {
#pragma unroll
            for (int k = 0; k < (int)PS_MAX_N; k++)
                if ((dm[k] >> r) & 1u) ps_store16(dv->h[k].b0 + o, x);
        }
    };
#pragma unroll
    for (int c = 0; c < D; c++) emit((uint32_t)c, p.in[c]);
    const uint32_t dmask = (1u << D) - 1u;
    ps_rebuild_emit<D>(v, ((1u << v.n) - 1u) & ~dmask, dmask, p.in, emit);  // compute_parity
}
template <int D, bool CRAFT>
__global__ __launch_bounds__(256) void ps_put_kernel(const PsView v, const uint32_t *__restrict__ a_n, const uint32_t *__restrict__ a_slot,
                                                     const uint32_t *__restrict__ a_val, const uint8_t *__restrict__ data, uint64_t data_stride,
                                                     const uint32_t *__restrict__ len, uint32_t data_len, uint32_t nblk, const RaftPeek cr) {
    PsPutLane<D> p;
    ps_put_load<D, CRAFT>(p, v, a_n, a_slot, a_val, data, data_stride, len, data_len, nblk, cr);
    ps_put_store<D, false>(p, v, nullptr, nullptr, 0u);
}
// thread j of a block decides for the j-th group the block's lanes belong to (a block's 256 lanes cover 256 / nblk + 1 groups at
// most -- four at config 4's size) and for every follower; the masks go through LDS to the group's lanes
template <int D, bool CRAFT>
__global__ __launch_bounds__(256) void ps_put_deliver_kernel(const PsView v, const uint32_t *__restrict__ a_n, const uint32_t *__restrict__ a_slot,
                                                             const uint32_t *__restrict__ a_val, const uint8_t *__restrict__ data,
                                                             uint64_t data_stride, const uint32_t *__restrict__ len, uint32_t data_len,
                                                             uint32_t nblk, const RaftPeek cr, const PsDeliver dv) {
    __shared__ uint8_t sh_dm[256 * PS_MAX_N];
    PsPutLane<D> p;
    ps_put_load<D, CRAFT>(p, v, a_n, a_slot, a_val, data, data_stride, len, data_len, nblk, cr);
    const uint64_t t0 = (uint64_t)blockIdx.x * 256;
    uint32_t g_first, g_last, rem;
    ps_divmod(t0, nblk, g_first, rem);
    ps_divmod(t0 + 255u, nblk, g_last, rem);
    const uint32_t gg = g_first + threadIdx.x;
    if (gg <= g_last && gg < v.G) {
        uint32_t tok = PS_NULL, row = 0, L = 0;
        const bool put = ps_put_what<CRAFT>(v, a_n, a_slot, a_val, len, data_len, cr, gg, tok, row, L);
#pragma unroll
        for (int k = 0; k < (int)PS_MAX_N; k++) {
            if ((uint32_t)k >= dv.n) break;
            const uint32_t m = put ? ps_deliver_mask(dv.h[k], dv.e[k], dv.lead, v.G, dv.all, row, gg, tok) : 0u;
            sh_dm[threadIdx.x * PS_MAX_N + (uint32_t)k] = (uint8_t)m;
            if (m) dv.h[k].dlv[row * v.G + gg] = (uint8_t)m;                // (a group that straddles two blocks: both write the same value)
        }
    }
    __syncthreads();
    ps_put_store<D, true>(p, v, &dv, sh_dm, g_first);
}

// what the engine says a ring cell holds against what the rows hold.  A lane takes RPL cells -- rows RPL q .. RPL q + RPL - 1 of one
// group, lanes side by side in the group: every load of a wavefront is still one run of bytes -- and looks at all of them in ONE
// round of loads (round 6: with a lane per cell the launch was four times the wavefronts, each waiting out its own round trip for
// 20 bytes; a steady tick changes one row of W).  A cell where the two agree is done there; the others go through ps_plan_cell.
// (sel: per group the ONE source a shard may come from -- the sender of the message the handler consumed -- or NULL: any)
struct PsCellOut {
    uint64_t src[2];
    uint32_t rc, sl[2], mat;
    bool work;
};
__device__ __forceinline__ PsCellOut ps_plan_cell(const PsView &v, const PsSrcs &S, const uint8_t *__restrict__ sel, uint32_t i, uint32_t g,
                                                  const uint32_t (&w_tok)[2], const uint32_t (&w_mask)[2], const uint32_t (&h_tok)[2],
                                                  const uint32_t (&h_mask)[2], uint32_t &n_copy, uint32_t &n_rebuilt, uint32_t &n_unsat,
                                                  uint32_t &n_rekey, uint32_t &n_dlv) {
    PsCellOut o;
    o.src[0] = o.src[1] = PS_NO_SRC; o.rc = 0; o.sl[0] = o.sl[1] = 0; o.mat = 0;
    uint32_t reqs_tok = PS_NULL, reqs_have = 0, reqs_len = 0;              // plane 0's new state: plane 1's "own other plane"
    const uint32_t only = sel ? sel[g] : PS_NONE;
    // A VOTED shard that equals the REQS row's -- same token, the shard present there -- is not stored twice: its bit in `alias`
    // says "read it from the REQS row".  That is every vote of a steady tick (`inst.voted = (ballot, reqs_cw.clone())`,
    // messages.rs:373-380; the leader's subset of its own codeword, request.rs:103-118), which made the second copy 38 % of the
    // bytes this store moved per tick.  The two part ways in the Prepare phase (reqs_cw takes the highest vote reported,
    // messages.rs:180-194, while my own vote stays): the shard is then moved into the VOTED row first (`mat`).
    const bool can_alias = v.pl[1].alias != nullptr;
    const uint32_t alias_old = can_alias ? v.pl[1].alias[i] : 0u;
    uint32_t alias_new = 0;
    for (int pl = 0; pl < 2; pl++) {
        uint32_t want_tok = w_tok[pl], want = w_mask[pl];
        const uint32_t had_tok = h_tok[pl], had = h_mask[pl], had_len = v.pl[pl].dlen[i];
        uint32_t have = had, L = had_len;
        if (had_tok != want_tok) {
            if (have) n_rekey++;
            have = 0; L = 0;
        }
        have &= want;                                                    // `inst.reqs_cw = reqs_cw`: shards the engine dropped
        if (pl == 1) {
            // aliases that go on: the REQS row still holds that token and that shard (it was not rewritten: a shard the row had
            // and keeps is never in its `need`); the others the vote still has move out of the REQS row now
            alias_new = reqs_tok == want_tok ? (alias_old & have & reqs_have) : 0u;
            o.mat = alias_old & have & ~alias_new;
        }
        uint32_t need = want & ~have;
        uint64_t sb = PS_NO_SRC;
        if (need && want_tok == 0) {                                     // from_data(ReqBatch::new()): one zero byte
            for (uint32_t m = need; m; m &= m - 1u) {
                const uint32_t k = (uint32_t)__ffs((int)m) - 1u;
                sb = (sb & ~(0xFFull << (8 * k))) | ((uint64_t)PS_EMPTY << (8 * k));
            }
            have |= need; need = 0; L = 1;
        }
        // sources in order: the voted plane asks my own reqs plane FIRST (its new state is in registers: no load, and nothing is
        // copied -- the shard becomes an alias); the reqs plane asks the named sources, then my own voted plane
        for (uint32_t jj = 0; jj <= S.n && need; jj++) {
            const uint32_t j = pl == 1 ? (jj == 0 ? S.n : jj - 1u) : jj;
            uint32_t s_tok, s_av, s_len, code, s_al = 0;                 // s_al: the source's shards that are aliases there
            if (j < S.n) {
                if (!S.p[j].tok || (sel && only != j)) continue;
                s_tok = S.p[j].tok[i]; s_av = S.p[j].avail[i]; s_len = S.p[j].dlen[i]; code = j;
                s_al = S.p[j].alias ? S.p[j].alias[i] : 0u;
            }
            // (my voted row's aliased shards ARE the reqs row's: nothing to take there)
            else if (pl == 0) { s_tok = v.pl[1].tok[i]; s_av = (uint32_t)v.pl[1].avail[i] & ~alias_old; s_len = v.pl[1].dlen[i]; code = PS_OWN; }
            else { s_tok = reqs_tok; s_av = reqs_have; s_len = reqs_len; code = PS_OWN; }
            const uint32_t take = (s_tok == want_tok) ? (need & s_av) : 0u;
            if (!take) continue;
            if (pl == 1 && code == PS_OWN && can_alias) alias_new |= take;
            else
                for (uint32_t m = take; m; m &= m - 1u) {
                    const uint32_t k = (uint32_t)__ffs((int)m) - 1u;
                    sb = (sb & ~(0xFFull << (8 * k))) | ((uint64_t)(code | (((s_al >> k) & 1u) ? PS_VIA0 : 0u)) << (8 * k));
                }
            n_copy += (uint32_t)__popc(take);
            L = s_len; need &= ~take; have |= take;
        }
        if (pl == 0 && v.dlv) {                                          // shards the sender's put launch has written here already (ps_put_deliver_kernel)
            const uint32_t dl = v.dlv[i];
            if (dl) {
                v.dlv[i] = 0;
                n_dlv += (uint32_t)__popc(dl);
                for (uint32_t m = dl; m; m &= m - 1u) sb |= 0xFFull << (8 * ((uint32_t)__ffs((int)m) - 1u));
            }
        }
        if (need && (uint32_t)__popc(have) >= v.d) {                     // reconstruct_data / compute_parity
            o.rc |= (need | (have << 8)) << (16 * pl);
            n_rebuilt += (uint32_t)__popc(need);
            have |= need; need = 0;
        }
        n_unsat += (uint32_t)__popc(need);
        // (only what changed: the steady tick touches one row of W, and 9 bytes x every cell x every call was ~19 MB of writes per
        // follow_many at config 4's size -- ADVICE r4)
        if (had_tok != want_tok) v.pl[pl].tok[i] = want_tok;
        if (had != have) v.pl[pl].avail[i] = (uint8_t)have;
        if (had_len != L) v.pl[pl].dlen[i] = L;
        o.src[pl] = sb; o.sl[pl] = ps_shard_len(L, v.d);
        if (pl == 0) { reqs_tok = want_tok; reqs_have = have; reqs_len = L; }
    }
    if (alias_new != alias_old) v.pl[1].alias[i] = (uint8_t)alias_new;
    o.work = o.src[0] != PS_NO_SRC || o.src[1] != PS_NO_SRC || o.rc != 0 || o.mat != 0;
    return o;
}

// cell (row, g): what the engine wants in the two planes and what the rows hold; `same`: the cell is done
__device__ __forceinline__ bool ps_cell_state(const PsView &v, const RspPeek &e, uint32_t row, uint32_t g, uint32_t c_len, uint32_t c_st, uint32_t c_rl,
                                              uint32_t (&w_tok)[2], uint32_t (&w_mask)[2], uint32_t (&h_tok)[2], uint32_t (&h_mask)[2]) {
    const uint32_t i = row * v.G + g;
    ps_cell_want(e, v.G, (1u << v.n) - 1u, row, g, c_len, c_st, c_rl, w_tok, w_mask);
    h_tok[0] = v.pl[0].tok[i]; h_tok[1] = v.pl[1].tok[i];
    h_mask[0] = v.pl[0].avail[i]; h_mask[1] = v.pl[1].avail[i];
    return h_mask[0] == w_mask[0] && h_tok[0] == w_tok[0] && h_mask[1] == w_mask[1] && h_tok[1] == w_tok[1];
}

template <int RPL>
__device__ __forceinline__ void ps_plan_body(const PsView &v, uint32_t flip, const RspPeek &e, const PsSrcs &S, const uint8_t *__restrict__ sel,
                                             uint32_t bx) {
    const uint32_t t = bx * 256 + threadIdx.x;
    const uint32_t nq = v.W / RPL;                                         // (RPL divides W: both powers of two, RPL <= W)
    const bool on = t < nq * v.G;
    const uint32_t q = on ? t / v.G : 0u, g = on ? t - q * v.G : 0u;
    uint32_t n_copy = 0, n_rebuilt = 0, n_unsat = 0, n_rekey = 0, n_dlv = 0;
    // ---- one round of loads: what the engine says against what the rows hold, both planes, RPL cells; a cell where they agree --
    // every cell but the tick's row, in a steady tick -- is done here: its lengths, its alias bits and the sources' cells are not read
    uint32_t w_tok[RPL][2], w_mask[RPL][2], h_tok[RPL][2], h_mask[RPL][2];
    bool same[RPL];
    uint32_t c_len = 0, c_st = 0, c_rl = 0;
    if (e.c_len) { c_len = e.c_len[g]; c_st = e.c_start[g]; c_rl = e.c_rlo[g]; }
#pragma unroll
    for (int k = 0; k < RPL; k++) {
        const uint32_t row = q * RPL + (uint32_t)k;
        same[k] = ps_cell_state(v, e, row, g, c_len, c_st, c_rl, w_tok[k], w_mask[k], h_tok[k], h_mask[k]);
    }
    PsCellOut out[RPL];
    uint32_t n_work = 0;
#pragma unroll
    for (int k = 0; k < RPL; k++) {
        out[k].work = false;
        if (on && !same[k]) {
            const uint32_t i = (q * RPL + (uint32_t)k) * v.G + g;
            out[k] = ps_plan_cell(v, S, sel, i, g, w_tok[k], w_mask[k], h_tok[k], h_mask[k], n_copy, n_rebuilt, n_unsat, n_rekey, n_dlv);
            n_work += out[k].work ? 1u : 0u;
        }
    }
    // ---- the cells with work onto the list: one append per BLOCK (the cells with work are neighbours -- a tick's row -- and ~15 ns
    // per same-address atomic times one per wavefront was a fifth of this kernel: 9.6 -> 7.7 us, profiles/r7d)
    const uint32_t lane = __lane_id(), wv = threadIdx.x >> 6;
    __shared__ uint32_t w_cnt[4], b_base;
    uint32_t incl = n_work;                                                // inclusive scan of n_work over the wavefront
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t x = (uint32_t)__shfl((int)incl, (int)(lane >= d ? lane - d : lane));
        if (lane >= d) incl += x;
    }
    if (lane == 63) w_cnt[wv] = incl;
    if (t == 0) v.it_n[flip ^ 1u] = 0;                                     // the next call's counter (this stream runs it after my byte kernel)
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = w_cnt[0] + w_cnt[1] + w_cnt[2] + w_cnt[3];
        b_base = tot ? atomicAdd(&v.it_n[flip], tot) : 0u;
    }
    __syncthreads();
    if (n_work) {
        uint32_t o = b_base + incl - n_work;
        for (uint32_t k = 0; k < wv; k++) o += w_cnt[k];
#pragma unroll
        for (int k = 0; k < RPL; k++) {
            if (!out[k].work) continue;
            v.it_cell[o] = (q * RPL + (uint32_t)k) * v.G + g;
            v.it_src[0][o] = out[k].src[0]; v.it_src[1][o] = out[k].src[1]; v.it_rc[o] = out[k].rc;
            v.it_sl[0][o] = out[k].sl[0]; v.it_sl[1][o] = out[k].sl[1]; v.it_mat[o] = (uint8_t)out[k].mat;
            o++;
        }
    }
    uint32_t c[5] = {n_copy, n_rebuilt, n_unsat, n_rekey, n_dlv};        // (4: of the copied shards, those a put launch had written already)
    if (__any((c[0] | c[1] | c[2] | c[3]) != 0))
        for (int k = 0; k < 5; k++) {
            uint32_t x = c[k];
            for (int off = 32; off > 0; off >>= 1) x += (uint32_t)__shfl_xor((int)x, off);
            if (lane == 0 && x) ctr_add(v.counters, k, (unsigned long long)x);
        }
}

// rows per lane of a plan launch for a window of W rows (a power of two)
#ifndef PS_RPL_MAX
#define PS_RPL_MAX 4       // (8: measured, profiles/s33)
#endif
static inline uint32_t ps_rpl(uint32_t W) { return W >= PS_RPL_MAX ? (uint32_t)PS_RPL_MAX : W >= 4 ? 4u : W; }

template <int RPL>
__global__ __launch_bounds__(256) void ps_plan_kernel(const PsView v, const RspPeek e, const PsSrcs S, const uint8_t *__restrict__ sel) {
    ps_plan_body<RPL>(v, v.flip, e, S, sel, blockIdx.x);
}
// several replicas that consume ONE sender's message (an Accept goes to every follower): blockIdx.y = which of them.  Their views
// are read from the device copies the stores keep (a by-value table of whole views, indexed by the block, went to scratch:
// 2320 B per lane); which of a store's two list counters is live comes as a bit of `flips`.  The source is none of them (the
// host checks), so nothing one of them writes is read by another.
struct PsMany {
    const PsView *v[PS_MAX_N];
    RspPeek e[PS_MAX_N];
    uint32_t flips;
};
template <int RPL>
__global__ __launch_bounds__(256) void ps_plan_many_kernel(const PsMany M, const PsSrcs S, const uint8_t *__restrict__ sel) {
    const PsView v = *M.v[blockIdx.y];             // a copy in registers: through the pointer every field is reloaded behind every store
    ps_plan_body<RPL>(v, (M.flips >> blockIdx.y) & 1u, M.e[blockIdx.y], S, sel, blockIdx.x);
}

// a lane per (listed cell, 16-byte column), columns fastest, in a grid-stride loop over cells x columns: the lanes of a wavefront
// run over cell boundaries (a wavefront per cell left a third of the lanes idle, 86 columns on 64 lanes: 22.6 -> 17.3 us, r7e).  Plane 0 first (plane 1
// may copy from it).  Where plane 0 is only copied into, a shard goes through both planes in one step: what plane 1 takes from
// plane 0 ("own other plane") without the alias array -- a CRaft store has one plane, so this is for completeness -- is the register
// that was stored there, not a read back; with it (every RSPaxos store) such a shard is no copy at all, see ps_plan_body.
// One (listed cell, 16-byte column):
__device__ __forceinline__ void ps_bytes_one(const PsView &v, const PsSrcs &S, uint32_t it, uint32_t c0) {
    const uint32_t rcw = v.it_rc[it], sl0 = v.it_sl[0][it], sl1 = v.it_sl[1][it];
    const uint64_t sb0 = v.it_src[0][it], sb1 = v.it_src[1][it];
    const uint32_t rc0 = rcw & 0xFFFFu, rc1 = rcw >> 16;
    const uint32_t mat = c0 < sl1 ? (uint32_t)v.it_mat[it] : 0u;
    const bool in0 = (sb0 != PS_NO_SRC || rc0) && c0 < sl0, in1 = (sb1 != PS_NO_SRC || rc1) && c0 < sl1;
    if (!in0 && !in1 && !mat) return;
    const uint32_t cell = v.it_cell[it], row = cell / v.G, g = cell % v.G;
    for (uint32_t k = 0; k < v.n; k++) {
        const size_t o = ps_off(v, row, k, g) + c0;
        // a vote that stops being an alias of the reqs row's shard: into its own row before this call's bytes land in the reqs row
        if ((mat >> k) & 1u) ps_store16(v.pl[1].bytes + o, ps_load16(v.pl[0].bytes + o));
        const uint32_t s0 = in0 ? (uint32_t)(sb0 >> (8 * k)) & 0xFFu : PS_NONE;
        ps_u32x4 x = {0u, 0u, 0u, 0u};
        if (s0 != PS_NONE) {
            if (s0 != PS_EMPTY) x = ps_load16((s0 == PS_OWN ? v.pl[1].bytes : ps_src(S, s0)) + o);
            ps_store16(v.pl[0].bytes + o, x);
        }
        if (rc0) continue;                                              // plane 1 waits for plane 0's rebuild (below)
        const uint32_t s1 = in1 ? (uint32_t)(sb1 >> (8 * k)) & 0xFFu : PS_NONE;
        if (s1 == PS_NONE) continue;
        if (!(s1 == PS_OWN && s0 != PS_NONE)) {                         // (else: x is what plane 0 holds there now)
            x = (ps_u32x4){0u, 0u, 0u, 0u};
            if (s1 != PS_EMPTY) x = ps_load16((s1 == PS_OWN ? v.pl[0].bytes : ps_src(S, s1)) + o);
        }
        ps_store16(v.pl[1].bytes + o, x);
    }
    if (rc0) {
        if (in0) ps_rebuild_small(v, v.pl[0].bytes, row, g, c0, rc0 & 0xFFu, rc0 >> 8);
        for (uint32_t k = 0; k < v.n && in1; k++) {
            const uint32_t s1 = (uint32_t)(sb1 >> (8 * k)) & 0xFFu;
            if (s1 == PS_NONE) continue;
            const size_t o = ps_off(v, row, k, g) + c0;
            ps_u32x4 x = {0u, 0u, 0u, 0u};
            if (s1 != PS_EMPTY) x = ps_load16((s1 == PS_OWN ? v.pl[0].bytes : ps_src(S, s1)) + o);
            ps_store16(v.pl[1].bytes + o, x);
        }
    }
    // (a vote rebuilt from d present shards: the ones that are aliases are read from the reqs row -- the plan kernel has stored
    // the cell's alias bits as they are after this call)
    if (rc1 && in1) ps_rebuild_small(v, v.pl[1].bytes, row, g, c0, rc1 & 0xFFu, rc1 >> 8, v.pl[1].alias ? v.pl[1].alias[cell] : 0u, v.pl[0].bytes);
}

// (Measured and dropped, profiles/r9c: four columns per lane and trip with their loads issued ahead of the first store -- 48.1 us
// against 49.0 for the four followers' launch; 32-bit index arithmetic instead of the emulated 64-bit division -- kept, no
// difference; 1024 .. 8192 blocks per store -- flat.  180 MB in 48 us is what a copy of this shape gets here.)
__device__ __forceinline__ void ps_bytes_body(const PsView &v, uint32_t flip, const PsSrcs &S, uint32_t bx, uint32_t nbx) {
    const uint32_t n_items = v.it_n[flip], ncol = v.cap_sl / 16u;
    const uint64_t total = (uint64_t)n_items * ncol, step = (uint64_t)nbx * 256u;
    for (uint64_t t = (uint64_t)bx * 256u + threadIdx.x; t < total; t += step) {
        uint32_t it, c0;
        ps_divmod(t, ncol, it, c0);
        ps_bytes_one(v, S, it, c0 * 16u);
    }
}

__global__ __launch_bounds__(256) void ps_bytes_kernel(const PsView v, const PsSrcs S) { ps_bytes_body(v, v.flip, S, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(256) void ps_bytes_many_kernel(const PsMany M, const PsSrcs S) {
    const PsView v = *M.v[blockIdx.y];             // (as above: 103 -> 63 us for four followers, profiles/r7j -> r7k)
    ps_bytes_body(v, (M.flips >> blockIdx.y) & 1u, S, blockIdx.x, gridDim.x);
}

// ---- a co-located leader's tick of the byte path as one call (round 6: smr_rsp_pstore_put_follow_all / smr_craft_pstore_put_follow_all)
// = put, the leader's own follow, one follow_many for its followers -- five launches as separate calls, four here: the leader's
// BYTE launch rides in the followers' PLAN launch (the first reads the leader's list and writes the leader's bytes, the second reads
// the leader's HEADERS, as its source's, and the followers': nothing one of them writes is read by the other).
// MEASURED and dropped (profiles/s21, s22): the leader's plan inside the put launch -- extra blocks for the cells the put does not
// write, a header's lane planning its own cell behind its own stores: bit-exact, three launches, and SLOWER: the plan's registers
// are the launch's (74 VGPRs against the put's 37: six wavefronts per SIMD instead of eight) and a wavefront with a header lane
// waits out that lane's three dependent rounds -- 52.9 / 55.1 us (plan blocks last / first in the grid) against 41.9 + 6.7.
constexpr uint32_t PS_LEADER_BYTE_BLOCKS = 64;     // of launch (2): the leader's list is empty in a steady tick (its vote is an alias)
template <int RPL>
__global__ __launch_bounds__(256) void ps_bytes_plan_many_kernel(const PsMany M, const PsSrcs S, const PsView lv) {
    if (blockIdx.x < PS_LEADER_BYTE_BLOCKS) {
        if (blockIdx.y == 0) ps_bytes_body(lv, lv.flip, S, blockIdx.x, PS_LEADER_BYTE_BLOCKS);   // (its items name no source: S is not looked at)
        return;
    }
    const PsView v = *M.v[blockIdx.y];
    ps_plan_body<RPL>(v, (M.flips >> blockIdx.y) & 1u, M.e[blockIdx.y], S, nullptr, blockIdx.x - PS_LEADER_BYTE_BLOCKS);
}

// rscoding.rs:583-609 for a list of instances: item i = (group[i] or i, slot[i]) -> out[i][0 .. dlen)
__global__ __launch_bounds__(256) void ps_get_kernel(const PsView v, uint32_t n_items, const uint32_t *__restrict__ group,
                                                     const uint32_t *__restrict__ slot, const uint32_t *__restrict__ expect, uint8_t *__restrict__ out,
                                                     uint64_t out_stride, uint32_t *__restrict__ len_out, uint8_t *__restrict__ ok, uint32_t nblk) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t it, blk;
    ps_divmod(t, nblk, it, blk);
    if (it >= n_items) return;
    const uint32_t g = group ? group[it] : it, s = slot[it];
    bool good = s != PS_NULL && g < v.G;
    uint32_t row = 0, L = 0;
    if (good) {
        row = s & v.Wmask;
        const size_t i = (size_t)row * v.G + g;
        const uint32_t dm = (1u << v.d) - 1u, tok = v.pl[0].tok[i];
        L = v.pl[0].dlen[i];
        good = tok != PS_NULL && (v.pl[0].avail[i] & dm) == dm && (!expect || expect[it] == tok) && L <= out_stride;
    }
    if (blk == 0) { ok[it] = good ? 1 : 0; len_out[it] = good ? L : 0u; }
    if (!good) return;
    const uint32_t sl = ps_shard_len(L, v.d), c0 = blk * 16u;
    if (c0 >= sl) return;
    uint8_t *dst = out + (size_t)it * out_stride;
    for (uint32_t c = 0; c < v.d; c++) {
        const uint32_t o = c * sl + c0;                                   // offset in the serialized bytes
        if (o >= L) break;
        uint32_t nb = sl - c0 < 16u ? sl - c0 : 16u;
        if (L - o < nb) nb = L - o;
        const uint8_t *p = v.pl[0].bytes + ps_off(v, row, c, g) + c0;
        if (nb == 16u) ps_store16(dst + o, ps_load16(p));
        else for (uint32_t b = 0; b < nb; b++) dst[o + b] = p[b];
    }
}

// The payload of a message, sender side: RSCodeword::subset_copy (rscoding.rs:255-293) of row slot[g] into a message buffer laid
// out like one row ([n][G][cap_sl]); what the row does not hold stays out of mask_out.  One lane per (group, 16-byte column).
__global__ __launch_bounds__(256) void ps_extract_kernel(const PsView v, int plane, const uint8_t *__restrict__ flags, const uint32_t *__restrict__ slot,
                                                         const uint8_t *__restrict__ mask, uint8_t *__restrict__ out, uint32_t *__restrict__ tok_out,
                                                         uint8_t *__restrict__ mask_out, uint32_t *__restrict__ dlen_out, uint32_t nblk) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t g, blk;
    ps_divmod(t, nblk, g, blk);
    if (g >= v.G) return;
    const bool on = (!flags || flags[g]) && slot[g] != PS_NULL;
    const uint32_t row = on ? (slot[g] & v.Wmask) : 0u;
    const size_t i = (size_t)row * v.G + g;
    const uint32_t tok = on ? v.pl[plane].tok[i] : PS_NULL;
    const uint32_t m = (on && tok != PS_NULL) ? ((uint32_t)mask[g] & v.pl[plane].avail[i]) : 0u;
    const uint32_t L = m ? v.pl[plane].dlen[i] : 0u;
    if (blk == 0) { tok_out[g] = m ? tok : PS_NULL; mask_out[g] = (uint8_t)m; dlen_out[g] = L; }
    const uint32_t c0 = blk * 16u;
    if (!m || c0 >= ps_shard_len(L, v.d)) return;
    for (uint32_t k = 0; k < v.n; k++)
        if ((m >> k) & 1u) ps_store16(out + ((size_t)k * v.G + g) * v.cap_sl + c0, ps_load16(ps_rd(v.pl[plane], i, k) + ps_off(v, row, k, g) + c0));
}

// ... and receiver side: the codeword a message carried becomes row slot[g] of a (staging) store -- token, shards present, length
// and bytes REPLACE what the row held -- for smr_rsp_pstore_follow to name as the source behind the handler that consumes the message
__global__ __launch_bounds__(256) void ps_ingest_kernel(const PsView v, int plane, const uint8_t *__restrict__ flags, const uint32_t *__restrict__ slot,
                                                        const uint32_t *__restrict__ tok, const uint8_t *__restrict__ mask,
                                                        const uint32_t *__restrict__ dlen, const uint8_t *__restrict__ in, uint32_t nblk) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t g, blk;
    ps_divmod(t, nblk, g, blk);
    if (g >= v.G || (flags && !flags[g]) || slot[g] == PS_NULL) return;
    const uint32_t row = slot[g] & v.Wmask;
    const size_t i = (size_t)row * v.G + g;
    uint32_t m = (uint32_t)mask[g] & ((1u << v.n) - 1u), L = dlen[g];
    if (tok[g] == PS_NULL || L > v.cap_sl * v.d) m = 0;
    if (blk == 0) {
        if (plane == 0) ps_drop_aliased(v, i);
        else if (v.pl[1].alias && v.pl[1].alias[i]) v.pl[1].alias[i] = 0;           // (the row's own bytes from here on)
        v.pl[plane].tok[i] = m ? tok[g] : PS_NULL; v.pl[plane].avail[i] = (uint8_t)m; v.pl[plane].dlen[i] = m ? L : 0u;
    }
    const uint32_t c0 = blk * 16u;
    if (!m || c0 >= ps_shard_len(L, v.d)) return;
    for (uint32_t k = 0; k < v.n; k++)
        if ((m >> k) & 1u) ps_store16(v.pl[plane].bytes + ps_off(v, row, k, g) + c0, ps_load16(in + ((size_t)k * v.G + g) * v.cap_sl + c0));
}

// The Accept a leader sends, as the frame TcpTransport writes (safetcp.rs:127-132: u64 BE length, then bincode "standard" of
// PeerMessage::Msg { msg: PeerMsg::Accept { slot, ballot, reqs_cw } }, rspaxos/mod.rs:262-270 -- variants 0 and 2 -- with
// RSCodeword's own Encode, utils/rscoding.rs:43-77: d u8, p u8, data_len, shard_len, Vec<Option<Vec<u8>>> shards, data_copy None)
// straight out of the store: per group g the shards mask[g] of row slot[g] that the row holds (`subset_copy`, request.rs:127-142)
// into frames + g * stride, byte for byte what smr_wire_rsp_accept(smr_wire_rscodeword(..)) writes on the host.  One lane per
// (group, 16-byte column): every lane walks the d + p Option headers for the offsets, the column-0 lane writes the header bytes.
__device__ __forceinline__ uint32_t ps_vl(uint64_t v) { return v < 251 ? 1u : v < (1ull << 16) ? 3u : v < (1ull << 32) ? 5u : 9u; }
__device__ __forceinline__ uint32_t ps_put_varint(uint8_t *p, uint64_t v) {
    if (v < 251) { p[0] = (uint8_t)v; return 1; }
    const uint32_t nb = v < (1ull << 16) ? 2u : v < (1ull << 32) ? 4u : 8u;
    p[0] = (uint8_t)(nb == 2 ? 0xFB : nb == 4 ? 0xFC : 0xFD);
    for (uint32_t i = 0; i < nb; i++) p[1 + i] = (uint8_t)(v >> (8 * i));
    return 1 + nb;
}
__global__ __launch_bounds__(256) void ps_emit_accepts_kernel(const PsView v, int plane, const uint8_t *__restrict__ flags, const uint32_t *__restrict__ slot,
                                                              const uint64_t *__restrict__ ballot, const uint8_t *__restrict__ mask,
                                                              uint8_t *__restrict__ frames, uint64_t stride, uint32_t *__restrict__ len, uint32_t nblk) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t g, blk;
    ps_divmod(t, nblk, g, blk);
    if (g >= v.G) return;
    const bool on = (!flags || flags[g]) && slot[g] != PS_NULL;
    const uint32_t row = on ? (slot[g] & v.Wmask) : 0u;
    const size_t i = (size_t)row * v.G + g;
    const uint32_t m = (on && v.pl[plane].tok[i] != PS_NULL) ? ((uint32_t)mask[g] & v.pl[plane].avail[i]) : 0u;
    if (!m) { if (blk == 0) len[g] = 0; return; }
    const uint32_t L = v.pl[plane].dlen[i], sl = ps_shard_len(L, v.d), c0 = blk * 16u;
    const uint64_t sv = slot[g], bv = ballot[g];
    const uint32_t h0 = 8u + 1u + 1u + ps_vl(sv) + ps_vl(bv) + 2u + ps_vl(L) + ps_vl(sl) + 1u;     // up to the first Option tag
    const uint32_t per = 1u + ps_vl(sl) + sl;
    const uint32_t total = h0 + (uint32_t)__popc(m) * per + (v.n - (uint32_t)__popc(m)) + 1u;
    if (total > stride) { if (blk == 0) len[g] = 0xFFFFFFFFu; return; }                           // the caller's slot is too short
    uint8_t *f = frames + (size_t)g * stride;
    if (blk == 0) {
        len[g] = total;
        const uint64_t plen = total - 8u;
        for (int b = 0; b < 8; b++) f[b] = (uint8_t)(plen >> (8 * (7 - b)));
        uint32_t o = 8;
        f[o++] = 0; f[o++] = 2;                                                                    // PeerMessage::Msg, PeerMsg::Accept
        o += ps_put_varint(f + o, sv); o += ps_put_varint(f + o, bv);
        f[o++] = (uint8_t)v.d; f[o++] = (uint8_t)(v.n - v.d);
        o += ps_put_varint(f + o, L); o += ps_put_varint(f + o, sl);
        f[o++] = (uint8_t)v.n;
        for (uint32_t k = 0; k < v.n; k++) {
            if ((m >> k) & 1u) { f[o++] = 1; o += ps_put_varint(f + o, sl); o += sl; }
            else f[o++] = 0;
        }
        f[o] = 0;                                                                                  // data_copy: None
    }
    if (c0 >= sl) return;
    uint32_t o = h0;
    for (uint32_t k = 0; k < v.n; k++) {
        if (!((m >> k) & 1u)) { o += 1; continue; }
        uint8_t *dst = f + o + 1u + ps_vl(sl) + c0;
        const uint8_t *src = ps_rd(v.pl[plane], i, k) + ps_off(v, row, k, g) + c0;
        if (c0 + 16u <= sl) ps_store16(dst, ps_load16(src));
        else for (uint32_t b = 0; b < sl - c0; b++) dst[b] = src[b];
        o += per;
    }
}

// ---- CRaft: the same store keyed by LOG INDEX (craft/mod.rs:129-150 `LogEntry { term, reqs_cw, .. }`) -----------------------
// A Raft entry's identity is (slot, term) (the Log Matching property; the consistency check of craft/messages.rs:95-131 compares
// exactly that), so the token of ring cell [slot % W] is a function of the two: 20 bits of the slot, 10 of the term, bit 30 set
// (never 0 = the synthesised empty batch, never PS_NULL).  Two entries that meet in one cell differ by a multiple of W in their
// slots: their tokens could only agree 2^20 / W wraps of the ring apart, and every wrap re-keys the cell.
// (Round 5b: the tokens are computed where they are compared -- craft_want_tok in the plan kernel, the put kernel's own lanes --
// instead of by a launch of their own into an array per store: three launches and 13 MB per tick less at config 4's size.)

// ---- host: GF(2^8) matrices for the rebuild table ------------------------------------------------------------------
struct PsGf {
    uint8_t exp[512], log[256];
    PsGf() {
        int x = 1;
        for (int i = 0; i < 255; i++) {
            exp[i] = (uint8_t)x; log[x] = (uint8_t)i;
            x <<= 1; if (x & 0x100) x ^= 0x11D;
        }
        for (int i = 255; i < 512; i++) exp[i] = exp[i - 255];
        log[0] = 0;
    }
    uint8_t mul(uint8_t a, uint8_t b) const { return (a && b) ? exp[log[a] + log[b]] : 0; }
    uint8_t inv(uint8_t a) const { return exp[255 - log[a]]; }
};
// Gauss-Jordan inverse of a d x d matrix (row major, stride 8); false if singular
static bool ps_invert(const PsGf &gf, uint8_t (*m)[PS_MAX_N], int d) {
    uint8_t aug[PS_MAX_N][2 * PS_MAX_N] = {};
    for (int r = 0; r < d; r++) {
        for (int c = 0; c < d; c++) aug[r][c] = m[r][c];
        aug[r][d + r] = 1;
    }
    for (int c = 0; c < d; c++) {
        int piv = -1;
        for (int r = c; r < d; r++) if (aug[r][c]) { piv = r; break; }
        if (piv < 0) return false;
        if (piv != c) for (int k = 0; k < 2 * d; k++) { const uint8_t t = aug[c][k]; aug[c][k] = aug[piv][k]; aug[piv][k] = t; }
        const uint8_t iv = gf.inv(aug[c][c]);
        for (int k = 0; k < 2 * d; k++) aug[c][k] = gf.mul(aug[c][k], iv);
        for (int r = 0; r < d; r++) {
            if (r == c || !aug[r][c]) continue;
            const uint8_t f = aug[r][c];
            for (int k = 0; k < 2 * d; k++) aug[r][k] ^= gf.mul(f, aug[c][k]);
        }
    }
    for (int r = 0; r < d; r++) for (int c = 0; c < d; c++) m[r][c] = aug[r][d + c];
    return true;
}
// table[pat][r][c]: every shard r from the first d present shards of pat (ReedSolomon::reconstruct: the sub-matrix of the
// first d present rows of the coding matrix, inverted; then the coding matrix times it)
static bool ps_build_table(int n, int d, std::vector<uint8_t> &tab) {
    std::vector<uint8_t> M((size_t)n * d);
    if (smr_rs_matrix(d, n - d, M.data()) != SMR_OK) return false;
    const PsGf gf;
    tab.assign(256 * 64, 0);
    for (int pat = 0; pat < (1 << n); pat++) {
        if (__builtin_popcount((unsigned)pat) < d) continue;
        uint8_t sub[PS_MAX_N][PS_MAX_N] = {};
        int idx = 0;
        for (int k = 0; k < n && idx < d; k++)
            if ((pat >> k) & 1) { for (int c = 0; c < d; c++) sub[idx][c] = M[(size_t)k * d + c]; idx++; }
        if (!ps_invert(gf, sub, d)) return false;
        for (int r = 0; r < n; r++)
            for (int c = 0; c < d; c++) {
                uint8_t acc = 0;
                for (int k = 0; k < d; k++) acc ^= gf.mul(M[(size_t)r * d + k], sub[k][c]);
                tab[(size_t)pat * 64 + r * 8 + c] = acc;
            }
    }
    return true;
}

}  // namespace smr

using namespace smr;

struct smr_rsp_pstore {
    PsView v;
    uint64_t plane_bytes;
    uint32_t max_data_len;
    void *meta;            // one allocation: tok / avail / dlen of both planes, the table, the list, the counters, a copy of v
    PsView *d_view;        // the device copy of v (flip excepted) smr_rsp_pstore_follow_many's kernels read
    void *plane_alloc[2];  // the planes' allocations (v.pl[p].bytes lies at their start)
    int planes;            // 2; 1 for a CRaft store (a log entry has one codeword: no VOTED bytes are allocated)
    bool craft_store;      // made by smr_craft_pstore_create
};

extern "C" {

static int ps_create(uint32_t n_groups, uint32_t n_shards, uint32_t n_data_shards, uint32_t window, uint32_t max_data_len, int planes,
                     smr_rsp_pstore **out) {
    if (!out) return fail(SMR_ERR_ARG, "pstore: null argument");
    if (n_groups == 0) return fail(SMR_ERR_ARG, "pstore: n_groups is zero");
    if (n_shards < 2 || n_shards > PS_MAX_N || n_data_shards == 0 || n_data_shards >= n_shards)
        return fail(SMR_ERR_ARG, "pstore: need 1 <= data shards < shards <= 8");
    if (!window || (window & (window - 1))) return fail(SMR_ERR_ARG, "pstore: window must be a power of two");
    if (max_data_len == 0) return fail(SMR_ERR_ARG, "pstore: max_data_len is zero");
    if ((uint64_t)window * n_groups > 0xFFFFFFFFull) return fail(SMR_ERR_ARG, "pstore: window x groups exceeds 2^32 cells");
    std::vector<uint8_t> tab;
    if (!ps_build_table((int)n_shards, (int)n_data_shards, tab)) return fail(SMR_ERR_ARG, "pstore: bad scheme");
    smr_rsp_pstore *s = new smr_rsp_pstore();
    memset(&s->v, 0, sizeof(s->v));
    PsView &v = s->v;
    v.G = n_groups; v.W = window; v.Wmask = window - 1; v.n = n_shards; v.d = n_data_shards;
    const uint32_t sl = (max_data_len + n_data_shards - 1) / n_data_shards;
    v.cap_sl = (sl + 15u) / 16u * 16u;
    s->max_data_len = max_data_len;
    s->planes = planes;
    s->craft_store = false;
    s->plane_bytes = (uint64_t)window * n_shards * n_groups * v.cap_sl;
    const size_t cells = (size_t)window * n_groups;
    Arena a, pa[2];
    size_t o_tok[2], o_av[2], o_len[2], o_src[2], o_sl[2], o_mat = 0, o_n = 0, o_cell = 0, o_rc = 0, o_ctr = 0, o_view = 0, o_bytes[2] = {0, 0};
    size_t o_alias = 0, o_itmat = 0, o_dlv = 0;
    // twice: sizes first, then -- the arenas allocated -- once more so that the kernel-source emulator of the CPU suite can mark the
    // unowned gap behind every array (SMR_ARENA_GUARD, smr_common.h); the offsets are the same both times
    auto layout = [&]() {
        a.used = 0;
        for (int p = 0; p < 2; p++) { o_tok[p] = a.reserve(cells * 4); o_av[p] = a.reserve(cells); o_len[p] = a.reserve(cells * 4); }
        o_mat = a.reserve(tab.size()); o_n = a.reserve(256); o_cell = a.reserve(cells * 4);
        for (int p = 0; p < 2; p++) { o_src[p] = a.reserve(cells * 8); o_sl[p] = a.reserve(cells * 4); }
        o_rc = a.reserve(cells * 4); o_ctr = a.reserve(SMR_CTR_WORDS * 8); o_view = a.reserve(sizeof(PsView));
        o_alias = a.reserve(cells); o_itmat = a.reserve(cells); o_dlv = a.reserve(cells);
        for (int p = 0; p < 2; p++) { pa[p].used = 0; o_bytes[p] = pa[p].reserve(p < planes ? s->plane_bytes : 256); }
    };
    layout();
    a.size = a.used + 256;
    hipError_t err = hipMalloc((void **)&a.base, a.size);
    for (int p = 0; p < 2 && err == hipSuccess; p++) {
        pa[p].size = pa[p].used + 256;
        err = hipMalloc((void **)&pa[p].base, pa[p].size);
    }
    if (err == hipSuccess) layout();
    for (int p = 0; p < 2 && err == hipSuccess; p++) v.pl[p].bytes = pa[p].at<uint8_t>(o_bytes[p]);
    if (err == hipSuccess) err = hipMemset(a.base, 0, a.size);
    for (int p = 0; p < 2 && err == hipSuccess; p++) {
        v.pl[p].tok = a.at<uint32_t>(o_tok[p]); v.pl[p].avail = a.at<uint8_t>(o_av[p]); v.pl[p].dlen = a.at<uint32_t>(o_len[p]);
        v.it_src[p] = a.at<uint64_t>(o_src[p]); v.it_sl[p] = a.at<uint32_t>(o_sl[p]);
        err = hipMemset(v.pl[p].tok, 0xFF, cells * 4);
        if (err == hipSuccess) err = hipMemset(v.pl[p].bytes, 0, p < planes ? s->plane_bytes : 256);
    }
    if (err == hipSuccess) err = hipMemcpy(a.base + o_mat, tab.data(), tab.size(), hipMemcpyHostToDevice);
    if (err != hipSuccess) {
        for (int p = 0; p < 2; p++) if (pa[p].base) (void)hipFree(pa[p].base);
        if (a.base) (void)hipFree(a.base);
        delete s;
        return fail(SMR_ERR_DEVICE, std::string("pstore: allocation: ") + hipGetErrorString(err));
    }
    v.mat = a.at<uint8_t>(o_mat); v.it_n = a.at<uint32_t>(o_n); v.it_cell = a.at<uint32_t>(o_cell); v.it_rc = a.at<uint32_t>(o_rc);
    v.counters = a.at<unsigned long long>(o_ctr);
    v.it_mat = a.at<uint8_t>(o_itmat);
    v.dlv = a.at<uint8_t>(o_dlv);                                             // (zeroed with the arena: nothing delivered)
    if (planes == 2) { v.pl[1].alias = a.at<uint8_t>(o_alias); v.pl[1].base0 = v.pl[0].bytes; }   // (zeroed with the arena: no aliases yet)
    s->meta = a.base;
    s->plane_alloc[0] = pa[0].base; s->plane_alloc[1] = pa[1].base;
    s->d_view = a.at<PsView>(o_view);
    err = hipMemcpy(s->d_view, &s->v, sizeof(PsView), hipMemcpyHostToDevice);
    if (err != hipSuccess) {
        smr_rsp_pstore_destroy(s);
        return fail(SMR_ERR_DEVICE, std::string("pstore: view copy: ") + hipGetErrorString(err));
    }
    *out = s;
    return SMR_OK;
}

int smr_rsp_pstore_create(uint32_t n_groups, uint32_t n_shards, uint32_t n_data_shards, uint32_t window, uint32_t max_data_len,
                          smr_rsp_pstore **out) {
    return ps_create(n_groups, n_shards, n_data_shards, window, max_data_len, 2, out);
}

int smr_craft_pstore_create(uint32_t n_groups, uint32_t n_shards, uint32_t n_data_shards, uint32_t window, uint32_t max_data_len,
                            smr_rsp_pstore **out) {
    int rc = ps_create(n_groups, n_shards, n_data_shards, window, max_data_len, 1, out);
    if (rc != SMR_OK) return rc;
    (*out)->craft_store = true;
    return SMR_OK;
}

void smr_rsp_pstore_destroy(smr_rsp_pstore *s) {
    if (!s) return;
    for (int p = 0; p < 2; p++) if (s->plane_alloc[p]) (void)hipFree(s->plane_alloc[p]);
    if (s->meta) (void)hipFree(s->meta);
    delete s;
}

static int ps_put(smr_rsp_pstore *s, const uint32_t *a_n_dev, const uint32_t *a_slot_dev, const uint32_t *a_val_dev, const uint8_t *data_dev,
                  uint64_t data_stride, const uint32_t *len_dev, uint32_t data_len, const RaftPeek *craft, void *stream, const PsDeliver *dv = nullptr) {
    if (data_len == 0 || data_len > s->max_data_len) return fail(SMR_ERR_ARG, "pstore put: data_len must be in 1..max_data_len");
    if (data_stride < data_len) return fail(SMR_ERR_ARG, "pstore put: data_stride is shorter than data_len");
    const PsView &v = s->v;
    const uint32_t nblk = ((data_len + v.d - 1) / v.d + 15u) / 16u;
    const uint64_t threads = (uint64_t)v.G * nblk;
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    RaftPeek cr;
    memset(&cr, 0, sizeof(cr));
    if (craft) cr = *craft;
#define PS_PUT_ARGS grid, block, 0, (hipStream_t)stream, v, a_n_dev, a_slot_dev, a_val_dev, data_dev, data_stride, len_dev, data_len, nblk, cr
#define PS_PUT(D)                                                                                                                         \
    case D:                                                                                                                               \
        if (dv && craft) hipLaunchKernelGGL((ps_put_deliver_kernel<D, true>), PS_PUT_ARGS, *dv);                                          \
        else if (dv) hipLaunchKernelGGL((ps_put_deliver_kernel<D, false>), PS_PUT_ARGS, *dv);                                             \
        else if (craft) hipLaunchKernelGGL((ps_put_kernel<D, true>), PS_PUT_ARGS);                                                        \
        else hipLaunchKernelGGL((ps_put_kernel<D, false>), PS_PUT_ARGS);                                                                  \
        break;
    switch (v.d) { PS_PUT(1) PS_PUT(2) PS_PUT(3) PS_PUT(4) PS_PUT(5) PS_PUT(6) PS_PUT(7) default: return fail(SMR_ERR_ARG, "pstore put: bad scheme"); }
#undef PS_PUT
#undef PS_PUT_ARGS
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_pstore_put(smr_rsp_pstore *s, const uint32_t *a_n_dev, const uint32_t *a_slot_dev, const uint32_t *a_val_dev,
                       const uint8_t *data_dev, uint64_t data_stride, const uint32_t *len_dev, uint32_t data_len, void *stream) {
    if (!s || !a_n_dev || !a_slot_dev || !a_val_dev || !data_dev) return fail(SMR_ERR_ARG, "pstore put: null argument");
    return ps_put(s, a_n_dev, a_slot_dev, a_val_dev, data_dev, data_stride, len_dev, data_len, nullptr, stream);
}

// the store follows the (token, mask) arrays `pk` names: smr_rsp_pstore_follow's an RSPaxos replica's, smr_craft_pstore_follow's the
// ones a CRaft replica's log implies
static int ps_follow(smr_rsp_pstore *s, const RspPeek &pk, uint32_t n_src, smr_rsp_pstore *const *src, const uint8_t *src_plane,
                     const uint8_t *sel_dev, void *stream) {
    if (n_src > PS_MAX_SRC) return fail(SMR_ERR_ARG, "pstore follow: at most 16 sources");
    const PsView &v = s->v;
    if (pk.G != v.G || pk.W != v.W || pk.R != v.n || pk.majority != v.d)
        return fail(SMR_ERR_ARG, "pstore follow: the replica's groups / window / population / majority differ from the store's");
    PsSrcs S;
    memset(&S, 0, sizeof(S));
    S.n = n_src;
    for (uint32_t j = 0; j < n_src; j++) {
        const smr_rsp_pstore *o = src[j];
        if (!o) continue;                                                    // an empty seat (e.g. my own id in a list indexed by replica)
        if ((int)src_plane[j] >= o->planes) return fail(SMR_ERR_ARG, "pstore follow: bad source plane");
        if (o == s) return fail(SMR_ERR_ARG, "pstore follow: a store's own planes are sources already");
        if (o->v.G != v.G || o->v.W != v.W || o->v.n != v.n || o->v.d != v.d || o->v.cap_sl != v.cap_sl)
            return fail(SMR_ERR_ARG, "pstore follow: a source has another geometry");
        S.set(j, o->v.pl[src_plane[j]]);
    }
    hipStream_t st = (hipStream_t)stream;
    s->v.flip ^= 1u;                               // (calls on one store are issued on one stream at a time: the header says so)
    const uint32_t cells = v.W * v.G;
    const uint32_t lanes = cells / ps_rpl(v.W);
    switch (ps_rpl(v.W)) {
    case PS_RPL_MAX: hipLaunchKernelGGL(ps_plan_kernel<PS_RPL_MAX>, dim3((lanes + 255) / 256), dim3(256), 0, st, v, pk, S, sel_dev); break;
#if PS_RPL_MAX != 4
    case 4: hipLaunchKernelGGL(ps_plan_kernel<4>, dim3((lanes + 255) / 256), dim3(256), 0, st, v, pk, S, sel_dev); break;
#endif
    case 2: hipLaunchKernelGGL(ps_plan_kernel<2>, dim3((lanes + 255) / 256), dim3(256), 0, st, v, pk, S, sel_dev); break;
    default: hipLaunchKernelGGL(ps_plan_kernel<1>, dim3((lanes + 255) / 256), dim3(256), 0, st, v, pk, S, sel_dev); break;
    }
    SMR_HIP_TRY(hipGetLastError());
    uint64_t blocks = ((uint64_t)cells * (v.cap_sl / 16u) + 255) / 256;            // (an upper bound: the list's length is the device's)
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ps_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, v, S);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_pstore_follow(smr_rsp_pstore *s, const smr_rsp_replica *e, uint32_t n_src, smr_rsp_pstore *const *src, const uint8_t *src_plane,
                          const uint8_t *sel_dev, void *stream) {
    if (!s || !e || (n_src && (!src || !src_plane))) return fail(SMR_ERR_ARG, "pstore follow: null argument");
    if (s->planes != 2) return fail(SMR_ERR_STATE, "pstore follow: a CRaft store follows a Raft replica (smr_craft_pstore_follow)");
    return ps_follow(s, rsp_peek(e), n_src, src, src_plane, sel_dev, stream);
}

// ---- CRaft: put at append, follow behind AppendEntries / ReconstructReply / commit (craft/request.rs:71-76, messages.rs:133-146,
// 193-233, 697-737; leadership.rs:80-141 decides the masks) ------------------------------------------------------------------
static RspPeek craft_as_peek(const RaftPeek &rp) {
    RspPeek pk;
    memset(&pk, 0, sizeof(pk));
    pk.G = rp.G; pk.W = rp.W; pk.R = rp.R; pk.me = 0; pk.majority = rp.quorum;
    pk.s_mask = rp.entry_mask; pk.s_vmask = rp.entry_mask;
    pk.c_len = rp.log_len; pk.c_start = rp.start_slot; pk.c_rlo = rp.ring_lo; pk.c_term = rp.entry_term;
    return pk;
}
static int craft_peek_of(smr_rsp_pstore *s, const smr_raft_leader *e, RaftPeek &rp) {
    if (!s->craft_store) return fail(SMR_ERR_STATE, "craft pstore: not a CRaft store (smr_craft_pstore_create)");
    rp = raft_peek(e);
    if (!rp.entry_mask) return fail(SMR_ERR_STATE, "craft pstore: CRaft is not enabled on this replica (smr_raft_craft_enable)");
    if (rp.G != s->v.G || rp.W != s->v.W || rp.R != s->v.n || rp.quorum != s->v.d)
        return fail(SMR_ERR_ARG, "craft pstore: the replica's groups / window / population / majority differ from the store's");
    return SMR_OK;
}

int smr_craft_pstore_put(smr_rsp_pstore *s, const smr_raft_leader *e, const uint32_t *slot_dev, const uint8_t *data_dev, uint64_t data_stride,
                         const uint32_t *len_dev, uint32_t data_len, void *stream) {
    if (!s || !e || !slot_dev || !data_dev) return fail(SMR_ERR_ARG, "craft pstore put: null argument");
    RaftPeek rp;
    if (int rc = craft_peek_of(s, e, rp)) return rc;
    return ps_put(s, nullptr, slot_dev, nullptr, data_dev, data_stride, len_dev, data_len, &rp, stream);
}

int smr_craft_pstore_follow(smr_rsp_pstore *s, const smr_raft_leader *e, uint32_t n_src, smr_rsp_pstore *const *src, const uint8_t *sel_dev,
                            void *stream) {
    if (!s || !e || (n_src && !src)) return fail(SMR_ERR_ARG, "craft pstore follow: null argument");
    RaftPeek rp;
    if (int rc = craft_peek_of(s, e, rp)) return rc;
    // the log's codewords are the REQS plane (their tokens: craft_want_tok in the plan kernel); nothing is wanted in the voted plane
    const RspPeek pk = craft_as_peek(rp);
    uint8_t planes[PS_MAX_SRC] = {0};
    return ps_follow(s, pk, n_src, src, planes, sel_dev, stream);
}

// stores[k] follows the (token, mask) arrays pk[k] names, each with the single source (src, src_plane): two launches for all of them
static int ps_many_setup(uint32_t n, smr_rsp_pstore *const *stores, const RspPeek *pk, const smr_rsp_pstore *src, int src_plane, PsMany &M, PsSrcs &S) {
    if (src && (src_plane < 0 || src_plane >= src->planes)) return fail(SMR_ERR_ARG, "pstore follow_many: bad source plane");
    memset(&M, 0, sizeof(M));
    memset(&S, 0, sizeof(S));
    const PsView &v0 = stores[0]->v;
    for (uint32_t k = 0; k < n; k++) {
        smr_rsp_pstore *s = stores[k];
        if (s == src) return fail(SMR_ERR_ARG, "pstore follow_many: the source must not be one of the stores that follow");
        for (uint32_t j = 0; j < k; j++) if (stores[j] == s) return fail(SMR_ERR_ARG, "pstore follow_many: a store is listed twice");
        if (s->v.G != v0.G || s->v.W != v0.W || s->v.n != v0.n || s->v.d != v0.d || s->v.cap_sl != v0.cap_sl)
            return fail(SMR_ERR_ARG, "pstore follow_many: the stores differ in geometry");
        if (pk[k].G != v0.G || pk[k].W != v0.W || pk[k].R != v0.n || pk[k].majority != v0.d)
            return fail(SMR_ERR_ARG, "pstore follow_many: a replica's groups / window / population / majority differ from the stores'");
        M.e[k] = pk[k];
    }
    if (src) {
        if (src->v.G != v0.G || src->v.W != v0.W || src->v.n != v0.n || src->v.d != v0.d || src->v.cap_sl != v0.cap_sl)
            return fail(SMR_ERR_ARG, "pstore follow_many: the source has another geometry");
        S.n = 1;
        S.set(0, src->v.pl[src_plane]);
    }
    return SMR_OK;
}
// (every check passed: from here on the stores' list counters flip)
static void ps_many_flip(uint32_t n, smr_rsp_pstore *const *stores, PsMany &M) {
    for (uint32_t k = 0; k < n; k++) {
        stores[k]->v.flip ^= 1u;
        M.v[k] = stores[k]->d_view;
        M.flips |= stores[k]->v.flip << k;
    }
}
static int ps_many_bytes(uint32_t n, const PsView &v0, const PsMany &M, const PsSrcs &S, hipStream_t st) {
    const uint32_t cells = v0.W * v0.G;
    uint64_t blocks = ((uint64_t)cells * (v0.cap_sl / 16u) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ps_bytes_many_kernel, dim3((unsigned)blocks, n), dim3(256), 0, st, M, S);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}
static int ps_follow_many(uint32_t n, smr_rsp_pstore *const *stores, const RspPeek *pk, const smr_rsp_pstore *src, int src_plane, void *stream) {
    PsMany M;
    PsSrcs S;
    if (int rc = ps_many_setup(n, stores, pk, src, src_plane, M, S)) return rc;
    ps_many_flip(n, stores, M);
    const PsView &v0 = stores[0]->v;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t cells = v0.W * v0.G;
    const uint32_t lanes = cells / ps_rpl(v0.W);
    switch (ps_rpl(v0.W)) {
    case PS_RPL_MAX: hipLaunchKernelGGL(ps_plan_many_kernel<PS_RPL_MAX>, dim3((lanes + 255) / 256, n), dim3(256), 0, st, M, S, (const uint8_t *)nullptr); break;
#if PS_RPL_MAX != 4
    case 4: hipLaunchKernelGGL(ps_plan_many_kernel<4>, dim3((lanes + 255) / 256, n), dim3(256), 0, st, M, S, (const uint8_t *)nullptr); break;
#endif
    case 2: hipLaunchKernelGGL(ps_plan_many_kernel<2>, dim3((lanes + 255) / 256, n), dim3(256), 0, st, M, S, (const uint8_t *)nullptr); break;
    default: hipLaunchKernelGGL(ps_plan_many_kernel<1>, dim3((lanes + 255) / 256, n), dim3(256), 0, st, M, S, (const uint8_t *)nullptr); break;
    }
    SMR_HIP_TRY(hipGetLastError());
    return ps_many_bytes(n, v0, M, S, st);
}

// (SMR_PS_DELIVER=0: the followers' shards through the byte kernel as before round 6 -- for A/B runs and the tests that compare the two)
static bool ps_deliver_on() {
    const char *e = getenv("SMR_PS_DELIVER");
    return !(e && e[0] == '0');
}
// put + the leader's follow + its followers' follow_many: four launches (ps_put_kernel, ps_plan_kernel, ps_bytes_plan_many_kernel,
// ps_bytes_many_kernel); a window of fewer than four rows takes the separate calls' five
static int ps_put_follow_all(smr_rsp_pstore *s, const RspPeek &pk, const uint32_t *a_n_dev, const uint32_t *a_slot_dev, const uint32_t *a_val_dev,
                             const uint8_t *data_dev, uint64_t data_stride, const uint32_t *len_dev, uint32_t data_len, const RaftPeek *craft,
                             uint32_t n, smr_rsp_pstore *const *stores, const RspPeek *fpk, void *stream) {
    if (data_len == 0 || data_len > s->max_data_len) return fail(SMR_ERR_ARG, "pstore put: data_len must be in 1..max_data_len");
    if (data_stride < data_len) return fail(SMR_ERR_ARG, "pstore put: data_stride is shorter than data_len");
    const PsView &v = s->v;
    if (pk.G != v.G || pk.W != v.W || pk.R != v.n || pk.majority != v.d)
        return fail(SMR_ERR_ARG, "pstore follow: the replica's groups / window / population / majority differ from the store's");
    PsMany M;
    PsSrcs S;
    if (n) if (int rc = ps_many_setup(n, stores, fpk, s, 0, M, S)) return rc;
    if (n && (stores[0]->v.G != v.G || stores[0]->v.W != v.W || stores[0]->v.n != v.n || stores[0]->v.d != v.d || stores[0]->v.cap_sl != v.cap_sl))
        return fail(SMR_ERR_ARG, "pstore follow_many: the source has another geometry");
    if (v.W < PS_RPL_MAX) {
        if (int rc = ps_put(s, a_n_dev, a_slot_dev, a_val_dev, data_dev, data_stride, len_dev, data_len, craft, stream)) return rc;
        if (int rc = ps_follow(s, pk, 0, nullptr, nullptr, nullptr, stream)) return rc;
        return n ? ps_follow_many(n, stores, fpk, s, 0, stream) : SMR_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    PsDeliver dv;                                                          // the followers' shards of this tick's codewords: written by the put launch
    memset(&dv, 0, sizeof(dv));
    dv.n = n; dv.all = (1u << v.n) - 1u; dv.lead = pk;
    for (uint32_t k = 0; k < n; k++) {
        const PsView &f = stores[k]->v;
        dv.h[k] = PsHdrs{f.pl[0].tok, f.pl[1].tok, f.pl[0].avail, f.pl[1].avail, f.pl[1].alias, f.dlv, f.pl[0].bytes};
        dv.e[k] = fpk[k];
    }
    if (int rc = ps_put(s, a_n_dev, a_slot_dev, a_val_dev, data_dev, data_stride, len_dev, data_len, craft, stream, (n && ps_deliver_on()) ? &dv : nullptr))
        return rc;
    const uint32_t cells = v.W * v.G, n_plan = (cells / (uint32_t)PS_RPL_MAX + 255) / 256;
    s->v.flip ^= 1u;
    {
        PsSrcs none;
        memset(&none, 0, sizeof(none));
        hipLaunchKernelGGL(ps_plan_kernel<PS_RPL_MAX>, dim3(n_plan), dim3(256), 0, st, v, pk, none, (const uint8_t *)nullptr);
        SMR_HIP_TRY(hipGetLastError());
    }
    if (!n) {                                                              // no followers here: the leader's bytes in a launch of their own
        uint64_t blocks = ((uint64_t)cells * (v.cap_sl / 16u) + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        PsSrcs none;
        memset(&none, 0, sizeof(none));
        hipLaunchKernelGGL(ps_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, v, none);
        SMR_HIP_TRY(hipGetLastError());
        return SMR_OK;
    }
    ps_many_flip(n, stores, M);
    hipLaunchKernelGGL(ps_bytes_plan_many_kernel<PS_RPL_MAX>, dim3(PS_LEADER_BYTE_BLOCKS + n_plan, n), dim3(256), 0, st, M, S, s->v);
    SMR_HIP_TRY(hipGetLastError());
    return ps_many_bytes(n, v, M, S, st);
}

int smr_rsp_pstore_follow_many(uint32_t n, smr_rsp_pstore *const *stores, const smr_rsp_replica *const *replicas, const smr_rsp_pstore *src,
                               int src_plane, void *stream) {
    if (!n || n > PS_MAX_N || !stores || !replicas) return fail(SMR_ERR_ARG, "pstore follow_many: 1 .. 8 stores");
    RspPeek pk[PS_MAX_N];
    for (uint32_t k = 0; k < n; k++) {
        if (!stores[k] || !replicas[k]) return fail(SMR_ERR_ARG, "pstore follow_many: null store / replica");
        if (stores[k]->planes != 2) return fail(SMR_ERR_STATE, "pstore follow_many: a CRaft store follows a Raft replica (smr_craft_pstore_follow_many)");
        pk[k] = rsp_peek(replicas[k]);
    }
    return ps_follow_many(n, stores, pk, src, src_plane, stream);
}

// smr_craft_pstore_follow for n <= 8 followers that consumed ONE leader's AppendEntries: a plan launch and a byte launch for all of them
int smr_craft_pstore_follow_many(uint32_t n, smr_rsp_pstore *const *stores, const smr_raft_leader *const *replicas, const smr_rsp_pstore *src,
                                 void *stream) {
    if (!n || n > PS_MAX_N || !stores || !replicas) return fail(SMR_ERR_ARG, "craft pstore follow_many: 1 .. 8 stores");
    RspPeek pk[PS_MAX_N];
    for (uint32_t k = 0; k < n; k++) {
        if (!stores[k] || !replicas[k]) return fail(SMR_ERR_ARG, "craft pstore follow_many: null store / replica");
        RaftPeek rp;
        if (int rc = craft_peek_of(stores[k], replicas[k], rp)) return rc;
        pk[k] = craft_as_peek(rp);
    }
    return ps_follow_many(n, stores, pk, src, 0, stream);
}

int smr_rsp_pstore_put_follow_all(smr_rsp_pstore *s, const smr_rsp_replica *e, const uint32_t *a_n_dev, const uint32_t *a_slot_dev, const uint32_t *a_val_dev,
                                  const uint8_t *data_dev, uint64_t data_stride, const uint32_t *len_dev, uint32_t data_len, uint32_t n,
                                  smr_rsp_pstore *const *stores, const smr_rsp_replica *const *replicas, void *stream) {
    if (!s || !e || !a_n_dev || !a_slot_dev || !a_val_dev || !data_dev || n > PS_MAX_N || (n && (!stores || !replicas)))
        return fail(SMR_ERR_ARG, "pstore put_follow_all: null argument / more than 8 followers");
    if (s->planes != 2) return fail(SMR_ERR_STATE, "pstore put_follow_all: a CRaft store follows a Raft replica (smr_craft_pstore_put_follow_all)");
    RspPeek pk[PS_MAX_N];
    for (uint32_t k = 0; k < n; k++) {
        if (!stores[k] || !replicas[k]) return fail(SMR_ERR_ARG, "pstore put_follow_all: null store / replica");
        if (stores[k]->planes != 2) return fail(SMR_ERR_STATE, "pstore put_follow_all: a CRaft store follows a Raft replica");
        pk[k] = rsp_peek(replicas[k]);
    }
    return ps_put_follow_all(s, rsp_peek(e), a_n_dev, a_slot_dev, a_val_dev, data_dev, data_stride, len_dev, data_len, nullptr, n, stores, pk, stream);
}

int smr_craft_pstore_put_follow_all(smr_rsp_pstore *s, const smr_raft_leader *e, const uint32_t *slot_dev, const uint8_t *data_dev, uint64_t data_stride,
                                    const uint32_t *len_dev, uint32_t data_len, uint32_t n, smr_rsp_pstore *const *stores,
                                    const smr_raft_leader *const *replicas, void *stream) {
    if (!s || !e || !slot_dev || !data_dev || n > PS_MAX_N || (n && (!stores || !replicas)))
        return fail(SMR_ERR_ARG, "craft pstore put_follow_all: null argument / more than 8 followers");
    RaftPeek rp;
    if (int rc = craft_peek_of(s, e, rp)) return rc;
    RspPeek pk[PS_MAX_N];
    for (uint32_t k = 0; k < n; k++) {
        if (!stores[k] || !replicas[k]) return fail(SMR_ERR_ARG, "craft pstore put_follow_all: null store / replica");
        RaftPeek fr;
        if (int rc = craft_peek_of(stores[k], replicas[k], fr)) return rc;
        pk[k] = craft_as_peek(fr);
    }
    return ps_put_follow_all(s, craft_as_peek(rp), nullptr, slot_dev, nullptr, data_dev, data_stride, len_dev, data_len, &rp, n, stores, pk, stream);
}

int smr_rsp_pstore_get_data(smr_rsp_pstore *s, uint32_t n_items, const uint32_t *group_dev, const uint32_t *slot_dev, const uint32_t *expect_dev,
                            uint8_t *out_dev, uint64_t out_stride, uint32_t *len_out_dev, uint8_t *ok_dev, void *stream) {
    if (!s || !slot_dev || !out_dev || !len_out_dev || !ok_dev) return fail(SMR_ERR_ARG, "pstore get_data: null argument");
    if (n_items == 0) return SMR_OK;
    if (!group_dev && n_items > s->v.G) return fail(SMR_ERR_ARG, "pstore get_data: more items than groups without a group list");
    const PsView &v = s->v;
    const uint32_t nblk = v.cap_sl / 16u;
    const uint64_t threads = (uint64_t)n_items * nblk;
    hipLaunchKernelGGL(ps_get_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, n_items, group_dev, slot_dev,
                       expect_dev, out_dev, out_stride, len_out_dev, ok_dev, nblk);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_pstore_extract(const smr_rsp_pstore *s, int plane, const uint8_t *flags_dev, const uint32_t *slot_dev, const uint8_t *mask_dev,
                           uint8_t *out_dev, uint32_t *tok_out_dev, uint8_t *mask_out_dev, uint32_t *dlen_out_dev, void *stream) {
    if (!s || plane < 0 || plane >= s->planes || !slot_dev || !mask_dev || !out_dev || !tok_out_dev || !mask_out_dev || !dlen_out_dev)
        return fail(SMR_ERR_ARG, "pstore extract: bad argument");
    const PsView &v = s->v;
    const uint32_t nblk = v.cap_sl / 16u;
    const uint64_t threads = (uint64_t)v.G * nblk;
    hipLaunchKernelGGL(ps_extract_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, plane, flags_dev, slot_dev,
                       mask_dev, out_dev, tok_out_dev, mask_out_dev, dlen_out_dev, nblk);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_pstore_ingest(smr_rsp_pstore *s, int plane, const uint8_t *flags_dev, const uint32_t *slot_dev, const uint32_t *tok_dev,
                          const uint8_t *mask_dev, const uint32_t *dlen_dev, const uint8_t *in_dev, void *stream) {
    if (!s || plane < 0 || plane >= s->planes || !slot_dev || !tok_dev || !mask_dev || !dlen_dev || !in_dev) return fail(SMR_ERR_ARG, "pstore ingest: bad argument");
    const PsView &v = s->v;
    const uint32_t nblk = v.cap_sl / 16u;
    const uint64_t threads = (uint64_t)v.G * nblk;
    hipLaunchKernelGGL(ps_ingest_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, plane, flags_dev, slot_dev,
                       tok_dev, mask_dev, dlen_dev, in_dev, nblk);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_pstore_emit_accepts(const smr_rsp_pstore *s, int plane, const uint8_t *flags_dev, const uint32_t *slot_dev, const uint64_t *ballot_dev,
                                const uint8_t *mask_dev, uint8_t *frames_dev, uint64_t stride, uint32_t *len_dev, void *stream) {
    if (!s || plane < 0 || plane >= s->planes || !slot_dev || !ballot_dev || !mask_dev || !frames_dev || !len_dev)
        return fail(SMR_ERR_ARG, "pstore emit_accepts: bad argument");
    if (stride < 64) return fail(SMR_ERR_ARG, "pstore emit_accepts: stride is shorter than a frame's header");
    const PsView &v = s->v;
    const uint32_t nblk = v.cap_sl / 16u;
    const uint64_t threads = (uint64_t)v.G * nblk;
    hipLaunchKernelGGL(ps_emit_accepts_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, plane, flags_dev,
                       slot_dev, ballot_dev, mask_dev, frames_dev, stride, len_dev, nblk);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_rsp_pstore_dump(smr_rsp_pstore *s, int plane, uint32_t *tok_host, uint8_t *avail_host, uint32_t *dlen_host) {
    if (!s || plane < 0 || plane >= s->planes) return fail(SMR_ERR_ARG, "pstore dump: bad argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const size_t cells = (size_t)s->v.W * s->v.G;
    if (tok_host) SMR_HIP_TRY(hipMemcpy(tok_host, s->v.pl[plane].tok, cells * 4, hipMemcpyDeviceToHost));
    if (avail_host) SMR_HIP_TRY(hipMemcpy(avail_host, s->v.pl[plane].avail, cells, hipMemcpyDeviceToHost));
    if (dlen_host) SMR_HIP_TRY(hipMemcpy(dlen_host, s->v.pl[plane].dlen, cells * 4, hipMemcpyDeviceToHost));
    return SMR_OK;
}

int smr_rsp_pstore_read_row(smr_rsp_pstore *s, int plane, uint32_t slot, uint8_t *bytes_host) {
    if (!s || plane < 0 || plane >= s->planes || !bytes_host) return fail(SMR_ERR_ARG, "pstore read_row: bad argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    const size_t row_bytes = (size_t)s->v.n * s->v.G * s->v.cap_sl;
    const uint32_t row = slot & s->v.Wmask;
    SMR_HIP_TRY(hipMemcpy(bytes_host, s->v.pl[plane].bytes + (size_t)row * row_bytes, row_bytes, hipMemcpyDeviceToHost));
    if (s->v.pl[plane].alias) {                                               // the row's shards that live in the REQS row
        std::vector<uint8_t> al(s->v.G);
        SMR_HIP_TRY(hipMemcpy(al.data(), s->v.pl[plane].alias + (size_t)row * s->v.G, s->v.G, hipMemcpyDeviceToHost));
        bool any = false;
        for (uint32_t g = 0; g < s->v.G && !any; g++) any = al[g] != 0;
        if (any) {
            std::vector<uint8_t> r0(row_bytes);
            SMR_HIP_TRY(hipMemcpy(r0.data(), s->v.pl[plane].base0 + (size_t)row * row_bytes, row_bytes, hipMemcpyDeviceToHost));
            for (uint32_t g = 0; g < s->v.G; g++)
                for (uint32_t k = 0; k < s->v.n; k++)
                    if ((al[g] >> k) & 1u) {
                        const size_t o = ((size_t)k * s->v.G + g) * s->v.cap_sl;
                        memcpy(bytes_host + o, r0.data() + o, s->v.cap_sl);
                    }
        }
    }
    return SMR_OK;
}

int smr_rsp_pstore_layout(const smr_rsp_pstore *s, int plane, void **bytes_dev, uint64_t *row_stride, uint64_t *shard_stride,
                          uint64_t *group_stride) {
    if (!s || plane < 0 || plane >= s->planes) return fail(SMR_ERR_ARG, "pstore layout: bad argument");
    if (bytes_dev) *bytes_dev = s->v.pl[plane].bytes;     // (plane 1: see smr_rsp_pstore_voted_alias)
    if (row_stride) *row_stride = (uint64_t)s->v.n * s->v.G * s->v.cap_sl;
    if (shard_stride) *shard_stride = (uint64_t)s->v.G * s->v.cap_sl;
    if (group_stride) *group_stride = s->v.cap_sl;
    return SMR_OK;
}

int smr_rsp_pstore_voted_alias(const smr_rsp_pstore *s, const uint8_t **alias_dev, uint8_t *alias_host) {
    if (!s) return fail(SMR_ERR_ARG, "pstore voted_alias: null argument");
    if (s->planes != 2) return fail(SMR_ERR_STATE, "pstore voted_alias: the store has no VOTED plane");
    if (alias_dev) *alias_dev = s->v.pl[1].alias;
    if (alias_host) {
        SMR_HIP_TRY(hipDeviceSynchronize());
        SMR_HIP_TRY(hipMemcpy(alias_host, s->v.pl[1].alias, (size_t)s->v.W * s->v.G, hipMemcpyDeviceToHost));
    }
    return SMR_OK;
}

int smr_rsp_pstore_counters(smr_rsp_pstore *s, uint64_t *out4_host) {
    if (!s || !out4_host) return fail(SMR_ERR_ARG, "pstore counters: null argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    unsigned long long c[4];
    SMR_HIP_TRY(ctr_read(s->v.counters, 4, c));
    for (int k = 0; k < 4; k++) out4_host[k] = c[k];
    return SMR_OK;
}

// of `copied`: the shards a sender's put launch wrote into this store (smr_*_pstore_put_follow_all), no copy listed for them
int smr_rsp_pstore_debug_delivered(smr_rsp_pstore *s, uint64_t *out_host) {
    if (!s || !out_host) return fail(SMR_ERR_ARG, "pstore delivered: null argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    unsigned long long c[5];
    SMR_HIP_TRY(ctr_read(s->v.counters, 5, c));
    *out_host = c[4];
    return SMR_OK;
}

}  // extern "C"

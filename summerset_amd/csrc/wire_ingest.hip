// Device-side ingest of MultiPaxos peer traffic (the HIP half): the bytes a leader's TCP
// connections delivered this tick -- `[u64 BE length][bincode(PeerMessage)]` frames, src/utils/safetcp.rs:30-70,
// 127-132; PeerMsg variants multipaxos/mod.rs:298-384 -- parsed on the device into the records the engine takes:
// AcceptReply -> smr_mp_ack (what smr_mp_deliver_acks puts into the ack matrix before the quorum tally), Heartbeat /
// CommitNotice -> smr_wire_hb; every other frame is only located (connection, kind, offset, length) for the host's
// smr_wire_decode.  The frame rules are smr_wire_decode's (csrc/wire.hip), restated for a lane.
//
// One lane per connection, one wavefront per block.  A connection's stream is walked through a 128-byte window that the
// wavefront copies from HBM with 16-byte loads into the lane's COLUMN of LDS -- dword d of lane l at win[d * 64 + l], so a
// wavefront's reads never share a bank whatever offsets its lanes are at, and every read is an aligned dword (gfx950's
// LDS does take unaligned 8- and 16-byte reads, but r3v's counters showed SQ_LDS_UNALIGNED_STALL at 87 % of the LDS
// unit's busy cycles with a row-per-lane layout read that way); nothing in LDS is shared between lanes, a column is its
// lane's indexable scratch -- and refilled at the lane's position when the next frame leaves it.  Round 3 (profiles/round3/r3m:
// ~600 wavefront instructions per frame through 188 branches): the frames a running cluster sends -- payloads of 2 .. 24
// bytes whose varints are below 2^32 -- take a straight-line path without a branch, every varint out of the two dwords
// it lies in; everything else (long frames, 64-bit values, timestamps, odd encodings, malformed input) takes the general
// reader, which the wavefront enters only when one of its lanes needs it.  Output order is the sequential decoder's
// (connection by connection, frame by frame): pass 1 counts per lane, per wavefront and per group of 64 wavefronts, pass 2
// sums what lies before its wavefront (two loads per lane), walks again and writes -- no same-address atomics on the
// record counters (three per wavefront on its group's row; smr_common.h on the event counters).  Pass 2 keeps a window's
// AcceptReplies as (slot, ballot) in the window itself -- 12 bytes over the >= 13 bytes of the frame they came from, so
// record j of a window lies wholly below the first unread byte -- and stores them as smr_mp_ack records when the window
// is done: a lane's records of a window leave back to back and merge in L2 (one 24-byte store per frame as it was parsed
// left every 128-byte line of the record array in L2 five times: 417 MB of HBM writes for 201 MB of records, r3m PMC).
//
// Round 5, smr_wire_ingest_mp_conn: the same walk as the ONLY pass.  What the counting pass buys is lists that are dense and in
// the decoder's order ACROSS connections, which the reference does not promise (a connection's messages in order, connections as
// the event loop meets them: server/transport.rs:404-470).  With a segment per connection nothing has to be counted: connection
// c's AcceptReplies start at record conn_off[c] / 13 (no frame that makes a record has fewer than 13 bytes, so the floors keep
// the segments apart), its Heartbeats and located frames have hb_per_conn / other_per_conn places at [c][..], and one more than
// that stops the walk IN FRONT of the frame (status 2: the host hands the rest to the next call, as it does with an incomplete
// frame).  A segment's records are the staged (slot, ballot) dwords as they stand -- 12 bytes, smr_wire_ack12: the group and the
// peer are the connection's -- because what records cost here is their stores' line requests: a probe of pass 2 without its
// stores ran 76 instead of 151 us.  235 -> 145 us per call at 262 144 connections (profiles/r9n, r9q).
#include "smr_common.h"

namespace smr {

typedef uint32_t wi_u32x4 __attribute__((ext_vector_type(4)));
#ifndef SMR_WI_WIN
#define SMR_WI_WIN 128                           // 9 KB of LDS per wavefront: 16 blocks per CU; >= 16 + 8 + WI_HOT_MAX
#endif
constexpr uint32_t WI_WIN = SMR_WI_WIN;          // bytes of a connection's stream in LDS at a time
constexpr uint32_t WI_DW = WI_WIN / 4;
constexpr uint32_t WI_ROWS = WI_DW + 3;          // + three dword rows: a read of up to three dwords at the window's last bytes stays inside
constexpr uint32_t WI_HOT_MAX = 64;              // no AcceptReply / Heartbeat / CommitNotice payload is longer (<= 38 bytes)
constexpr uint32_t WI_FAST_MAX = 24;             // payloads up to this long can take the straight-line path
static_assert(WI_WIN % 16 == 0 && 64 % (WI_WIN / 16) == 0 && WI_WIN >= 16 + 8 + WI_HOT_MAX, "window: whole 16-byte chunks, a whole number of connections per load");

// smr_wire's Rd over my lane's column (bytes [n, end) of the window), eight bytes at a time: three dwords, shifted into place
struct ColRd {
    const uint32_t *col;                         // &win[lane]
    uint32_t n, end;
    bool ok;
    __device__ __forceinline__ uint64_t peek64() const {             // bytes n .. n + 7 of the window, little-endian (n <= WI_WIN)
        const uint32_t i = n >> 2, sh = 8 * (n & 3);
        const uint32_t a = col[i * 64], b = col[(i + 1) * 64], c = col[(i + 2) * 64];
        const uint32_t lo = (uint32_t)((((uint64_t)b << 32) | a) >> sh), hi = (uint32_t)((((uint64_t)c << 32) | b) >> sh);
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ __forceinline__ uint8_t byte() {
        if (n < end) { const uint8_t b = (uint8_t)peek64(); n++; return b; }
        ok = false;
        return 0;
    }
    __device__ __forceinline__ uint64_t varint() {
        if (n >= end) { ok = false; return 0; }
        const uint64_t x = peek64();
        const uint32_t b = (uint32_t)(x & 0xFF);
        const uint32_t need = b < 251 ? 1 : b == 0xFB ? 3 : b == 0xFC ? 5 : b == 0xFD ? 9 : 0;   // 0xFE (u128), 0xFF: not on this path
        if (need == 0 || n + need > end) { ok = false; n = end; return 0; }
        uint64_t v = b;
        if (need == 3) v = (x >> 8) & 0xFFFF;
        else if (need == 5) v = (x >> 8) & 0xFFFFFFFFull;
        else if (need == 9) { n += 1; v = peek64(); n -= 1; }
        n += need;
        return v;
    }
};

// PeerMessage -> (kind, the hot variants' fields): smr_wire_decode's rules (csrc/wire.hip)
template <typename Rd>
__device__ __forceinline__ void parse_peer_message(Rd &r, uint32_t &kind, bool &hot, uint64_t &f0, uint64_t &f1, uint64_t &f2, uint64_t &f3) {
    const uint64_t outer = r.varint();
    if (outer == 2) kind = SMR_WIRE_LEAVE;                                          // PeerMessage::Leave
    else if (outer == 0) {                                                          // PeerMessage::Msg { msg }
        const uint64_t v = r.varint();
        kind = (uint32_t)(v <= SMR_WIRE_COMMIT_NOTICE ? v : SMR_WIRE_OTHER);
        if (v == SMR_WIRE_ACCEPT_REPLY) {
            f0 = r.varint(); f1 = r.varint();                                       // slot, ballot
            const uint8_t ts = r.byte();                                            // Option<SystemTime>
            if (ts == 1) { r.varint(); r.varint(); } else if (ts != 0) r.ok = false;
            hot = true;
        } else if (v == SMR_WIRE_HEARTBEAT) {
            f0 = r.varint(); f1 = r.varint(); f2 = r.varint(); f3 = r.varint();   // ballot, commit_bar, exec_bar, snap_bar
            hot = true;
        } else if (v == SMR_WIRE_COMMIT_NOTICE) {
            f0 = r.varint(); f1 = r.varint();                                       // ballot, commit_bar
            hot = true;
        }
    }
}

// the five bytes at byte a of my column (more where a is not the last byte of its dword), in the low bits: the two
// dwords they lie in, shifted.  An offset past anything a straight-line frame has reads the column's spare rows.
__device__ __forceinline__ uint64_t wi_bytes_at(const uint32_t *col, uint32_t a) {
    const uint32_t i = a >> 2 <= WI_DW + 1 ? a >> 2 : WI_DW + 1;
    const uint32_t d0 = col[i * 64], d1 = col[(i + 1) * 64];
    return (((uint64_t)d1 << 32) | d0) >> (8 * (a & 3));
}
// one varint below 2^32 out of the (>= 5) bytes x that start at it: its length (99: longer, or no varint -- the caller's
// length check then fails) and value
__device__ __forceinline__ void fast_varint(uint64_t x, uint32_t &need, uint32_t &v) {
    const uint32_t b = (uint32_t)x & 0xFF, rest = (uint32_t)(x >> 8);
    const uint32_t t = (b > 250 ? b : 250u) - 250;                   // 0: one byte; 1, 2: 0xFB, 0xFC; 3 .. 5: not this path's
    need = (0x636363050301ull >> (8 * t)) & 0xFF;                  // 1, 3, 5, 99, 99, 99 (a table in a constant)
    v = t == 0 ? b : t == 1 ? (rest & 0xFFFFu) : rest;
}

// One frame as the straight-line path sees it: a payload of 2 .. 24 bytes, all of it in the window, varints below 2^32.
struct WiFrame {
    uint32_t plen, kind, adv;                    // payload length (garbage unless the header was looked at), SMR_WIRE_*, 8 + length
    uint64_t glen;                               // the whole frame's length where the general reader took it (SEG: its located record)
    uint64_t f0, f1, f2, f3;                     // the hot variants' fields
    bool fast;                                   // this path takes the frame
    bool slow;                                   // the general reader has to look at it
    bool fin, nxt;                               // not complete in the stream: my connection is done / needs the next window
    bool take, ack, hbt, loc;                    // the frame is taken: as an smr_mp_ack / an smr_wire_hb / located by this path (the general reader stores its own)
    // the frame at window offset wo (0 if !act), r8 = my stream's bytes from there on, less 8 (>= 0)
    __device__ __forceinline__ void look(const uint32_t *col, uint32_t wo, bool act, int64_t r8) {
        // the header and the payload's first eight bytes: five dwords of my column, shifted into place
        const uint32_t hi0 = wo >> 2, hsh = 8 * (wo & 3);
        const uint32_t d0 = col[hi0 * 64], d1 = col[(hi0 + 1) * 64], d2 = col[(hi0 + 2) * 64], d3 = col[(hi0 + 3) * 64], d4 = col[(hi0 + 4) * 64];
        const uint32_t Hx = (uint32_t)((((uint64_t)d1 << 32) | d0) >> hsh), Hy = (uint32_t)((((uint64_t)d2 << 32) | d1) >> hsh);
        const uint32_t Hz = (uint32_t)((((uint64_t)d3 << 32) | d2) >> hsh), Hw = (uint32_t)((((uint64_t)d4 << 32) | d3) >> hsh);
        const bool big = Hx != 0;                                                   // a length of 2^32 or more: the general reader's
        plen = __builtin_bswap32(Hy);
        const uint32_t room = WI_WIN - 8 - wo;
        const bool whole = act && !big && (int64_t)plen <= r8;
        fin = act && !big && !whole;                                                // frame not complete yet
        nxt = whole && (plen < WI_HOT_MAX ? plen : WI_HOT_MAX) > room;              // (after a refill the offset is < 16: always fits)
        act = act && !fin && !nxt;
        const uint64_t lo = ((uint64_t)Hw << 32) | Hz;
        const uint32_t b0 = Hz & 0xFF, b1 = (Hz >> 8) & 0xFF;
        const bool msg = b0 == 0;
        const bool isar = msg && b1 == SMR_WIRE_ACCEPT_REPLY, ishb = msg && b1 == SMR_WIRE_HEARTBEAT, iscn = msg && b1 == SMR_WIRE_COMMIT_NOTICE;
        const bool hot = isar || ishb || iscn;
        uint32_t k0, k1, k2, k3, v0, v1, v2, v3;                                    // (a varint past the frame reads bytes nobody looks at)
        fast_varint(lo >> 16, k0, v0);
        const uint32_t p1 = wo + 10 + k0;
        fast_varint(wi_bytes_at(col, p1), k1, v1);
        const uint32_t p2 = p1 + k1;
        const uint64_t x2 = wi_bytes_at(col, p2);
        fast_varint(x2, k2, v2);
        const uint32_t p3 = p2 + k2;
        fast_varint(wi_bytes_at(col, p3), k3, v3);
        const uint32_t p4 = p3 + k3;
        const uint32_t used = (isar ? p2 + 1 : ishb ? p4 : p2) - (wo + 8);
        fast = act && !big && plen >= 2 && plen <= WI_FAST_MAX && b0 < 251 && (!msg || b1 < 251) &&
               (!hot || (used == plen && (!isar || (uint8_t)x2 == 0)));
        slow = act && !fast;
        take = fast; ack = fast && isar; hbt = fast && hot && !isar; loc = fast && !hot;
        kind = b0 == 2 ? (uint32_t)SMR_WIRE_LEAVE : msg && b1 <= SMR_WIRE_COMMIT_NOTICE ? b1 : (uint32_t)SMR_WIRE_OTHER;
        f0 = v0; f1 = v1; f2 = ishb ? v2 : 0u; f3 = ishb ? v3 : 0u;
        adv = 8 + plen;
    }
};

struct IngestArgs {
    const uint8_t *buf; uint64_t buf_len;
    const uint64_t *conn_off; const uint32_t *conn_group; const uint8_t *conn_peer; uint32_t n_conn;
    smr_mp_ack *acks; uint64_t ack_cap;
    smr_wire_hb *hbs; uint64_t hb_cap;
    smr_wire_other *others; uint64_t other_cap;
    uint64_t *super;                             // [ceil(n_waves / 64)][3]: the counts of 64 wavefronts together (pass 1, atomics)
    uint64_t *wave_cnt;                          // [n_waves][3]: a wavefront's counts (pass 1)
    uint32_t *lane_cnt;                          // [n_waves * 64][3]: pass 1's counts per connection
    uint64_t *counts;                            // [4]: acks, heartbeats / commit notices, others, malformed connections
    uint64_t *consumed; int32_t *status;
    // one pass, a segment per connection (smr_wire_ingest_mp_conn): connection c's acks -- 12-byte smr_wire_ack12 records behind
    // `acks` -- from record conn_off[c] / 13 on (an AcceptReply frame has >= 13 bytes: the segments cannot meet), its first
    // hb_per_conn / other_per_conn heartbeats / located frames at [c][..]; seg_cnt [n_conn][3] = how many of each
    uint32_t hb_per_conn, other_per_conn;
    uint32_t *seg_cnt;
};

// the sum of x over the lanes before mine (the block is one wavefront)
__device__ __forceinline__ uint32_t wave_exclusive_sum(uint32_t x, uint32_t lane, uint32_t &total) {
    uint32_t s = x;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl(s, (int)(lane >= d ? lane - d : lane));
        if (lane >= d) s += t;
    }
    total = __shfl(s, 63);
    return s - x;
}

// the sum of a 64-bit x over the wavefront
__device__ __forceinline__ uint64_t wave_sum64(uint64_t x) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) x += __shfl_xor(x, m);
    return x;
}

// WRITE = false: count my connection's records, report consumed / status; true: write them at my bases.
// SEG (with WRITE): the only pass -- my bases are my connection's own segments (IngestArgs), nothing is counted first and no
// record's place depends on another connection; a heartbeat / located frame my segment has no room for stops the walk in
// front of it (status 2: not malformed -- consumed says where the next call goes on)
template <bool WRITE, bool SEG = false>
__global__ __launch_bounds__(64) void wire_ingest_mp_kernel(IngestArgs A) {
    __shared__ uint32_t win[WI_ROWS * 64];
    const uint32_t lane = threadIdx.x, c = blockIdx.x * 64 + lane;
    const bool live = c < A.n_conn;
    const uint64_t start = live ? A.conn_off[c] : 0, end = live ? A.conn_off[c + 1] : 0;
    const uint32_t group = live ? A.conn_group[c] : 0, peer = live ? A.conn_peer[c] : 0;
    int st = 0;
    bool done = !live;
    if (live && (end < start || end > A.buf_len)) { st = 1; done = true; }
    uint64_t base0 = 0, base1 = 0, base2 = 0;
    if (SEG) {
        base0 = start / 13u; base1 = (uint64_t)c * A.hb_per_conn; base2 = (uint64_t)c * A.other_per_conn;
    } else if (WRITE) {
        // where my records go: the wavefronts before mine -- whole groups of 64 of them out of `super`, the rest of my group
        // out of their own counts, a row per lane -- and the lanes before me
        const uint32_t w = blockIdx.x, g0 = w & ~63u;
        uint64_t s0 = 0, s1 = 0, s2 = 0;
        for (uint32_t i = lane; i < w / 64; i += 64) { s0 += A.super[i * 3 + 0]; s1 += A.super[i * 3 + 1]; s2 += A.super[i * 3 + 2]; }
        if (g0 + lane < w) { s0 += A.wave_cnt[(size_t)(g0 + lane) * 3 + 0]; s1 += A.wave_cnt[(size_t)(g0 + lane) * 3 + 1]; s2 += A.wave_cnt[(size_t)(g0 + lane) * 3 + 2]; }
        uint32_t t;
        base0 = wave_sum64(s0) + wave_exclusive_sum(live ? A.lane_cnt[(size_t)c * 3 + 0] : 0u, lane, t);
        base1 = wave_sum64(s1) + wave_exclusive_sum(live ? A.lane_cnt[(size_t)c * 3 + 1] : 0u, lane, t);
        base2 = wave_sum64(s2) + wave_exclusive_sum(live ? A.lane_cnt[(size_t)c * 3 + 2] : 0u, lane, t);
        if (w == 0) {                             // the call's totals: every group's sum (pass 1 is complete)
            uint64_t t0 = 0, t1 = 0, t2 = 0;
            for (uint32_t i = lane; i < (gridDim.x + 63) / 64; i += 64) { t0 += A.super[i * 3 + 0]; t1 += A.super[i * 3 + 1]; t2 += A.super[i * 3 + 2]; }
            t0 = wave_sum64(t0); t1 = wave_sum64(t1); t2 = wave_sum64(t2);
            if (lane == 0) { A.counts[0] = t0; A.counts[1] = t1; A.counts[2] = t2; }
        }
    }
    uint32_t n0 = 0, n1 = 0, nall = 0;            // my records so far: acks, heartbeats / commit notices, all (the others: the difference)
    uint32_t *const col = &win[lane];
    // my position: byte woff of the window whose first byte is the stream's byte wlo; r8 = the bytes of my stream from it on, less 8
    uint64_t wlo = start & ~15ull;
    uint32_t woff = (uint32_t)(start - wlo);
    int64_t r8 = done ? -1 : (int64_t)(end - start) - 8;
    for (;;) {
        if (!__ballot(!done)) break;              // wave-uniform: a lane that is done stays to help with the refills
        // ---- refill: every lane's column gets what is left of its stream at its position, at most the window.  The wavefront
        // loads together: WI_WIN / 16 neighbouring lanes take one connection's window, one 16-byte chunk each, so an
        // instruction reads whole runs of WI_WIN contiguous bytes; every chunk's load is issued before the first one is
        // waited for; a chunk that is not wanted, or not whole, reads the buffer's first 16 bytes instead
        {
            const uint64_t pos = wlo + woff;
            wlo = pos & ~15ull;
            woff = (uint32_t)(pos - wlo);
        }
        const uint64_t wbase = wlo;
        const uint64_t left = done ? 0 : end - wbase;
        const uint32_t nchunk = (uint32_t)(left >= WI_WIN ? WI_WIN / 16 : (left + 15) / 16);
        constexpr uint32_t CPW = WI_WIN / 16, PER = 64 / CPW;                       // lanes per connection; connections per instruction
        __syncthreads();                                                            // (one wavefront per block) every lane has left the old window
        const uint32_t k = lane % CPW;
        wi_u32x4 q[CPW];
        uint32_t want = 0, ragged = 0;                                              // bit i: chunk i of my connection-of-the-round is wanted / is the buffer's ragged end
        const bool can = A.buf_len >= 16;
#pragma unroll
        for (uint32_t i = 0; i < CPW; i++) {
            const uint32_t src = i * PER + lane / CPW;
            const uint64_t wb = __shfl(wbase, (int)src);
            const uint32_t nc = __shfl(nchunk, (int)src);
            const uint64_t off = wb + 16ull * k;
            const bool w = k < nc, whole = w && off + 16 <= A.buf_len;
            want |= (uint32_t)w << i;
            ragged |= (uint32_t)(w && !whole) << i;
            q[i] = wi_u32x4{0, 0, 0, 0};
            if (can) q[i] = *(const wi_u32x4 *)(A.buf + (whole ? off : 0));
        }
        if (__ballot(ragged != 0)) {                                                // (one wavefront of the whole grid, once: wave-uniform, the shuffles meet)
#pragma unroll
            for (uint32_t i = 0; i < CPW; i++) {
                const uint64_t off = __shfl(wbase, (int)(i * PER + lane / CPW)) + 16ull * k;
                if ((ragged >> i) & 1) {
                    uint32_t v[4] = {0, 0, 0, 0};
                    for (uint32_t b = 0; b < 16 && off + b < A.buf_len; b++) v[b >> 2] |= (uint32_t)A.buf[off + b] << (8 * (b & 3));
                    q[i] = wi_u32x4{v[0], v[1], v[2], v[3]};
                }
            }
        }
#pragma unroll
        for (uint32_t i = 0; i < CPW; i++) {
            if ((want >> i) & 1) {
                const uint32_t src = i * PER + lane / CPW;
                win[(4 * k + 0) * 64 + src] = q[i].x; win[(4 * k + 1) * 64 + src] = q[i].y;
                win[(4 * k + 2) * 64 + src] = q[i].z; win[(4 * k + 3) * 64 + src] = q[i].w;
            }
        }
        __syncthreads();
        // ---- the frames that lie inside the window, two per turn: the second frame's position needs only the first one's
        // length, so the two frames' chains of dependent LDS reads (header, then one per varint) run side by side
        bool wait = false;                                                          // my next frame needs the next window
        uint32_t *hp = col;                                                         // pass 2: where the next AcceptReply of this window is kept in my column
        for (;;) {
            bool act = !done && !wait;
            if (act && r8 < 0) { done = true; act = false; }                        // length not complete yet
            if (act && woff + 8 > WI_WIN) { wait = true; act = false; }
            if (!__ballot(act)) break;
            WiFrame F[2];
            F[0].look(col, act ? woff : 0u, act, r8);
            const uint32_t wo1 = woff + 8 + F[0].plen;
            const int64_t r81 = r8 - (int64_t)(8 + F[0].plen);
            const bool act1 = F[0].fast && r81 >= 0 && wo1 + 8 <= WI_WIN;          // (else the next turn looks at it)
            F[1].look(col, act1 ? wo1 : 0u, act1, r81);
            done |= F[0].fin || F[1].fin;
            wait |= F[0].nxt || F[1].nxt;
            const uint64_t wlo0 = wlo;                                              // (the general reader may step wlo over a very long frame)
            const bool slow = F[0].slow || F[1].slow;                               // (at most one of them: a second frame is only looked at behind a fast one)
            if (__ballot(slow)) {                                                   // rare: the general reader (smr_wire_decode's rules in full)
                if (slow) {
                    const bool second = !F[0].slow;
                    const uint32_t wo = second ? wo1 : woff;
                    const int64_t rr = second ? r81 : r8;
                    WiFrame G = second ? F[1] : F[0];
                    ColRd r{col, wo, wo + 8, true};
                    const uint64_t gl = __builtin_bswap64(r.peek64());
                    r.n += 8;
                    if (gl > 1000000000000ull) { st = 1; done = true; }             // safetcp.rs:56-66
                    else if ((uint64_t)rr < gl) done = true;                        // frame not complete yet
                    else if ((uint32_t)(gl < WI_HOT_MAX ? gl : WI_HOT_MAX) > WI_WIN - 8 - wo) wait = true;   // (only a length of 2^32 or more gets here unchecked)
                    else {
                        const uint32_t look = (uint32_t)(gl < WI_HOT_MAX ? gl : WI_HOT_MAX);
                        r.end = wo + 8 + look;
                        uint32_t gk = SMR_WIRE_OTHER;
                        bool gh = false;
                        uint64_t g0 = 0, g1 = 0, g2 = 0, g3 = 0;
                        parse_peer_message(r, gk, gh, g0, g1, g2, g3);
                        // a frame whose leading varints do not parse, or a hot frame that does not end where its length says
                        if (!r.ok || (gh && (uint64_t)(r.n - (wo + 8)) != gl)) { st = 1; done = true; }
                        else {
                            // (an AcceptReply for a slot the engine cannot name -- its slots are u32 -- goes the host's way)
                            G.ack = gk == SMR_WIRE_ACCEPT_REPLY && g0 <= 0xFFFFFFFFull;
                            G.hbt = gk == SMR_WIRE_HEARTBEAT || gk == SMR_WIRE_COMMIT_NOTICE;
                            G.kind = gk; G.f0 = g0; G.f1 = g1; G.f2 = g2; G.f3 = g3;
                            G.take = true; G.loc = false;
                            G.glen = 8 + gl;
                            if (WRITE && !SEG && !G.ack && !G.hbt) {                // (located: here, where its 64-bit length is at hand)
                                const uint64_t at = base2 + (nall - n0 - n1) + (second ? (uint32_t)F[0].loc : 0u);
                                if (at < A.other_cap) {
                                    smr_wire_other o; o.conn = c; o.kind = gk; o.off = wlo + wo; o.len = 8 + gl;
                                    A.others[at] = o;
                                }
                            }
                            if (gl > 0x7FFFFFF0u) {                                 // a frame longer than my 32-bit window offset can step over
                                wlo += 8 + gl; r8 -= (int64_t)(8 + gl); G.adv = 0; wait = true;   // (its end, 16-aligned down at the refill)
                            } else G.adv = 8 + (uint32_t)gl;
                        }
                    }
                    if (second) F[1] = G; else F[0] = G;
                }
            }
            bool stop = false;                                                      // SEG: a record my segment has no room for: nothing from it on is taken
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const WiFrame &X = F[u];
                const bool other = X.take && !X.ack && !X.hbt;                      // located by either path
                if (SEG) {
                    if (!stop && ((X.hbt && n1 >= A.hb_per_conn) || (other && nall - n0 - n1 >= A.other_per_conn))) { stop = true; st = 2; done = true; }
                    if (stop) continue;
                }
                if (WRITE) {
                    if (X.ack) {                                                    // kept in the window: three dwords over a frame of >= 13 bytes
                        hp[0] = (uint32_t)X.f0; hp[64] = (uint32_t)X.f1; hp[128] = (uint32_t)(X.f1 >> 32);
                        hp += 192;
                    } else if (X.hbt) {
                        const uint64_t at = base1 + n1;
                        if (at < A.hb_cap) {
                            smr_wire_hb h; h.group = group; h.peer = peer; h.kind = X.kind; h.reserved = 0; h.ballot = X.f0; h.commit_bar = X.f1;
                            h.exec_bar = X.f2; h.snap_bar = X.f3;
                            A.hbs[at] = h;
                        }
                    } else if (SEG ? other : X.loc) {                               // a frame the device only locates (SEG: the general reader's too)
                        const uint64_t at = base2 + (nall - n0 - n1);
                        if (at < A.other_cap) {
                            smr_wire_other o; o.conn = c; o.kind = X.kind; o.off = wlo0 + woff; o.len = X.loc ? X.adv : X.glen;
                            A.others[at] = o;
                        }
                    }
                }
                n0 += X.ack; n1 += X.hbt; nall += X.take;
                woff += X.take ? X.adv : 0u;
                r8 -= X.take ? (int64_t)X.adv : 0;
            }
            if (SEG && stop && wlo != wlo0) { r8 += (int64_t)(wlo - wlo0); wlo = wlo0; }   // (a very long frame stepped over, then not taken)
        }
        if (SEG) {
            // a segment's records are (slot, ballot) alone -- 12 bytes, smr_wire_ack12: the group and the peer are the connection's --
            // so a lane's records of a window are the staged dwords as they stand, four records = three 16-byte stores.  What the
            // records cost is the stores' line requests, not their bytes alone (a probe of the dense call's second pass without its
            // stores ran 76 instead of 151 us): half the bytes, half the requests
            const uint32_t held = (uint32_t)(hp - col) / 192;
            const uint64_t first = base0 + (n0 - held);
            const uint32_t fits = first >= A.ack_cap ? 0u : A.ack_cap - first < held ? (uint32_t)(A.ack_cap - first) : held;
            uint32_t *dst = (uint32_t *)A.acks + first * 3u;
            const uint32_t nd = 3u * fits;                                          // staged dwords to store: rows 0 .. nd - 1 of my column
            const uint32_t *rp = col;
            for (uint32_t d = 0; __ballot(d < nd); d += 4, dst += 4, rp += 256) {
                if (d + 4 <= nd) {
                    const wi_u32x4 w{rp[0], rp[64], rp[128], rp[192]};
                    __builtin_memcpy(dst, &w, 16);
                } else if (d < nd) {
                    dst[0] = rp[0];
                    if (d + 1 < nd) dst[1] = rp[64];
                    if (d + 2 < nd) dst[2] = rp[128];
                }
            }
        } else if (WRITE) {                                                         // this window's AcceptReplies leave, a lane's back to back
            const uint32_t held = (uint32_t)(hp - col) / 192;
            const uint64_t first = base0 + (n0 - held);
            const uint32_t fits = first >= A.ack_cap ? 0u : A.ack_cap - first < held ? (uint32_t)(A.ack_cap - first) : held;
            // two records per turn: 48 contiguous bytes = three 16-byte stores (a record alone is a 16- and an 8-byte one): 1.5 store
            // instructions per record instead of 2 -- each costs the memory pipeline 64 different lines (r3z)
            uint8_t *dst = (uint8_t *)(A.acks + first);
            const uint32_t *rp = col;
            for (uint32_t j = 0; __ballot(j < fits); j += 2, dst += 48, rp += 384) {
                if (j + 1 < fits) {
                    const wi_u32x4 w0{group, rp[0], rp[64], rp[128]}, w1{peer, 0u, group, rp[192]}, w2{rp[256], rp[320], peer, 0u};
                    __builtin_memcpy(dst, &w0, 16); __builtin_memcpy(dst + 16, &w1, 16); __builtin_memcpy(dst + 32, &w2, 16);
                } else if (j < fits) {
                    smr_mp_ack a1; a1.group = group; a1.slot = rp[0]; a1.ballot = ((uint64_t)rp[128] << 32) | rp[64]; a1.peer = peer; a1.reserved = 0;
                    *(smr_mp_ack *)dst = a1;
                }
            }
        }
    }
    if (SEG) {
        if (live) {
            A.consumed[c] = wlo + woff - start; A.status[c] = st;
            A.seg_cnt[(size_t)c * 3 + 0] = n0; A.seg_cnt[(size_t)c * 3 + 1] = n1; A.seg_cnt[(size_t)c * 3 + 2] = nall - n0 - n1;
        }
    }
    if (!WRITE) {
        if (live) {
            A.consumed[c] = wlo + woff - start; A.status[c] = st;
            A.lane_cnt[(size_t)c * 3 + 0] = n0; A.lane_cnt[(size_t)c * 3 + 1] = n1; A.lane_cnt[(size_t)c * 3 + 2] = nall - n0 - n1;
        }
        uint32_t t0, t1, t2;
        (void)wave_exclusive_sum(n0, lane, t0); (void)wave_exclusive_sum(n1, lane, t1); (void)wave_exclusive_sum(nall - n0 - n1, lane, t2);
        const unsigned long long bad = __ballot(st != 0);
        if (lane == 0) {
            A.wave_cnt[(size_t)blockIdx.x * 3 + 0] = t0; A.wave_cnt[(size_t)blockIdx.x * 3 + 1] = t1; A.wave_cnt[(size_t)blockIdx.x * 3 + 2] = t2;
            // my group's sums: three atomics per wavefront on one of n_waves / 64 rows (64 wavefronts per address, not all of them)
            atomicAdd((unsigned long long *)&A.super[(blockIdx.x / 64) * 3 + 0], (unsigned long long)t0);
            atomicAdd((unsigned long long *)&A.super[(blockIdx.x / 64) * 3 + 1], (unsigned long long)t1);
            atomicAdd((unsigned long long *)&A.super[(blockIdx.x / 64) * 3 + 2], (unsigned long long)t2);
            if (bad) atomicAdd((unsigned long long *)&A.counts[3], (unsigned long long)__popcll(bad));
        }
    }
}

}  // namespace smr

using namespace smr;

extern "C" {

uint64_t smr_wire_ingest_scratch_bytes(uint32_t n_conn) {
    const uint64_t n_waves = ((uint64_t)n_conn + 63) / 64, n_super = (n_waves + 63) / 64;
    return n_super * 3 * 8 + n_waves * 3 * 8 + n_waves * 64 * 3 * 4;
}

int smr_wire_ingest_mp(const uint8_t *buf_dev, uint64_t buf_len, const uint64_t *conn_off_dev, const uint32_t *conn_group_dev,
                       const uint8_t *conn_peer_dev, uint32_t n_conn, smr_mp_ack *acks_dev, uint64_t ack_cap, smr_wire_hb *hbs_dev,
                       uint64_t hb_cap, smr_wire_other *others_dev, uint64_t other_cap, uint64_t *counts_dev, uint64_t *consumed_dev,
                       int32_t *status_dev, void *scratch_dev, void *stream) {
    if ((n_conn && (!conn_off_dev || !conn_group_dev || !conn_peer_dev)) || !counts_dev || !consumed_dev || !status_dev || !scratch_dev ||
        (buf_len && !buf_dev) || (ack_cap && !acks_dev) || (hb_cap && !hbs_dev) || (other_cap && !others_dev))
        return fail(SMR_ERR_ARG, "wire ingest: null argument");
    if (((uintptr_t)buf_dev & 15) || ((uintptr_t)scratch_dev & 7))
        return fail(SMR_ERR_ARG, "wire ingest: the byte buffer must be 16-byte aligned, the scratch 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    SMR_HIP_TRY(hipMemsetAsync(counts_dev, 0, 4 * 8, st));
    if (n_conn == 0) return SMR_OK;
    const uint32_t n_waves = (n_conn + 63) / 64, n_super = (n_waves + 63) / 64;
    uint64_t *super = (uint64_t *)scratch_dev, *wave_cnt = super + (size_t)n_super * 3;
    SMR_HIP_TRY(hipMemsetAsync(super, 0, (size_t)n_super * 3 * 8, st));
    IngestArgs A{buf_dev, buf_len, conn_off_dev, conn_group_dev, conn_peer_dev, n_conn, acks_dev, ack_cap, hbs_dev, hb_cap,
                 others_dev, other_cap, super, wave_cnt, (uint32_t *)(wave_cnt + (size_t)n_waves * 3), counts_dev, consumed_dev, status_dev};
    hipLaunchKernelGGL(wire_ingest_mp_kernel<false>, dim3(n_waves), dim3(64), 0, st, A);
    hipLaunchKernelGGL(wire_ingest_mp_kernel<true>, dim3(n_waves), dim3(64), 0, st, A);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

// One pass, a segment per connection (round 5).  The two-pass call above exists for ONE property: its lists are dense and in the
// sequential decoder's order across connections -- which costs a whole counting parse (76 of the call's 214 us at 262 144
// connections).  Nothing downstream needs it: the reference handles a connection's messages in order and connections in whatever
// order its event loop meets them (transport.rs:404-470), and smr_mp_deliver_acks_conn takes the segments as they are.
int smr_wire_ingest_mp_conn(const uint8_t *buf_dev, uint64_t buf_len, const uint64_t *conn_off_dev, const uint32_t *conn_group_dev,
                            const uint8_t *conn_peer_dev, uint32_t n_conn, smr_wire_ack12 *acks_dev, uint64_t ack_cap, smr_wire_hb *hbs_dev,
                            uint32_t hb_per_conn, smr_wire_other *others_dev, uint32_t other_per_conn, uint32_t *cnt_dev, uint64_t *consumed_dev,
                            int32_t *status_dev, void *stream) {
    if ((n_conn && (!conn_off_dev || !conn_group_dev || !conn_peer_dev || !cnt_dev)) || !consumed_dev || !status_dev || (buf_len && !buf_dev) ||
        (ack_cap && !acks_dev) || (hb_per_conn && !hbs_dev) || (other_per_conn && !others_dev))
        return fail(SMR_ERR_ARG, "wire ingest: null argument");
    if ((uintptr_t)buf_dev & 15) return fail(SMR_ERR_ARG, "wire ingest: the byte buffer must be 16-byte aligned");
    if (ack_cap < buf_len / 13 + 1)
        return fail(SMR_ERR_ARG, "wire ingest: the ack array must hold buf_len / 13 + 1 records (a segment per connection, an AcceptReply frame has >= 13 bytes)");
    if (n_conn == 0) return SMR_OK;
    const uint32_t n_waves = (n_conn + 63) / 64;
    IngestArgs A{buf_dev, buf_len, conn_off_dev, conn_group_dev, conn_peer_dev, n_conn, (smr_mp_ack *)acks_dev, ack_cap, hbs_dev, (uint64_t)n_conn * hb_per_conn,
                 others_dev, (uint64_t)n_conn * other_per_conn, nullptr, nullptr, nullptr, nullptr, consumed_dev, status_dev, hb_per_conn, other_per_conn, cnt_dev};
    hipLaunchKernelGGL((wire_ingest_mp_kernel<true, true>), dim3(n_waves), dim3(64), 0, (hipStream_t)stream, A);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

}  // extern "C"

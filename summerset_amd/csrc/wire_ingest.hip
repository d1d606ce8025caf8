// Device-side ingest of MultiPaxos peer traffic (the HIP half): the bytes a leader's TCP
// connections delivered this tick -- `[u64 BE length][bincode(PeerMessage)]` frames, src/utils/safetcp.rs:30-70,
// 127-132; PeerMsg variants multipaxos/mod.rs:298-384 -- parsed on the device into the records the engine takes:
// AcceptReply -> smr_mp_ack (what smr_mp_deliver_acks puts into the ack matrix before the quorum tally), Heartbeat /
// CommitNotice -> smr_wire_hb; every other frame is only located (connection, kind, offset, length) for the host's
// smr_wire_decode.  The frame rules are smr_wire_decode's (csrc/wire.hip), restated for a lane.
//
// One lane per connection, one wavefront per block.  A connection's stream is walked through a 128-byte window that
// the lane copies from HBM with 16-byte loads (8 in flight) into ITS column of LDS -- dword d of lane l at
// win[d * 64 + l], so a wavefront's byte reads never share a bank whatever offsets its lanes are at; nothing in LDS
// is shared between lanes, it is the lane's indexable scratch -- and refilled at the lane's position when the next
// frame leaves it.  Output order is the sequential decoder's (connection by connection, frame by frame): pass 1
// counts per lane and per wavefront, a one-block scan turns the wavefronts' counts into bases, pass 2 walks again and
// writes -- no atomics on the record counters (a same-address atomic per wavefront and pass would cost more than the
// bytes: smr_common.h on the event counters).
#include "smr_common.h"

namespace smr {

typedef uint32_t wi_u32x4 __attribute__((ext_vector_type(4)));
#ifndef SMR_WI_WIN
#define SMR_WI_WIN 128                           // 8 KB of LDS per wavefront: 16 blocks per CU (256: 9, 512: 4); >= 16 + 8 + WI_HOT_MAX.  Measured
                                                 // (profiles/r2p_wire_ingest_first.log, same call): 128 -> 451 us, 256 -> 642, 512 -> 709 per ingest
#endif
#ifndef SMR_WI_COOP
#define SMR_WI_COOP 1                            // refills loaded by the wavefront together (0: every lane its own window)
#endif
constexpr uint32_t WI_WIN = SMR_WI_WIN;          // bytes of a connection's stream in LDS at a time
constexpr uint32_t WI_DW = WI_WIN / 4;
constexpr uint32_t WI_HOT_MAX = 64;              // no AcceptReply / Heartbeat / CommitNotice payload is longer (<= 38 bytes)

// smr_wire's Rd over my lane's window (bytes [n, end) of it).  The window is read eight bytes at a time -- three dwords of
// my column, shifted into place -- and a varint is taken out of that register pair: a byte-at-a-time reader costs a
// dozen instructions and (every fourth byte) a dependent LDS round trip PER BYTE, and the parse, not the bytes, is what
// this kernel's time is (a wavefront instruction takes 4 cycles; 33 frames x 17 bytes per lane and pass).
struct WinRd {
    const uint32_t *col;                         // &win[lane]; two spare dword rows lie behind the window
    uint32_t n, end;
    bool ok;
    __device__ __forceinline__ uint64_t peek64() const {             // bytes n .. n + 7 of the window, little-endian
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
    __device__ __forceinline__ uint64_t be64() {                     // the frame header's length
        if (n + 8 <= end) { const uint64_t x = peek64(); n += 8; return __builtin_bswap64(x); }
        ok = false;
        return 0;
    }
    __device__ __forceinline__ uint64_t varint() {
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

#ifndef SMR_WI_RING
#define SMR_WI_RING 0                            // 1 (experiment, DESIGN §4 "next"): the stream as a ring of whole 128-byte lines per lane
#endif
// WinRd over a ring of 64 dword rows (two 128-byte lines): row0 = the row of the dword the frame starts in, n / end count
// bytes from that dword's first byte
struct RingRd {
    const uint32_t *col;
    uint32_t row0, n, end;
    bool ok;
    __device__ __forceinline__ uint64_t peek64() const {
        const uint32_t i = row0 + (n >> 2), sh = 8 * (n & 3);
        const uint32_t a = col[(i & 63u) * 64], b = col[((i + 1) & 63u) * 64], c = col[((i + 2) & 63u) * 64];
        const uint32_t lo = (uint32_t)((((uint64_t)b << 32) | a) >> sh), hi = (uint32_t)((((uint64_t)c << 32) | b) >> sh);
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ __forceinline__ uint8_t byte() {
        if (n < end) { const uint8_t b = (uint8_t)peek64(); n++; return b; }
        ok = false;
        return 0;
    }
    __device__ __forceinline__ uint64_t be64() {
        if (n + 8 <= end) { const uint64_t x = peek64(); n += 8; return __builtin_bswap64(x); }
        ok = false;
        return 0;
    }
    __device__ __forceinline__ uint64_t varint() {
        const uint64_t x = peek64();
        const uint32_t b = (uint32_t)(x & 0xFF);
        const uint32_t need = b < 251 ? 1 : b == 0xFB ? 3 : b == 0xFC ? 5 : b == 0xFD ? 9 : 0;
        if (need == 0 || n + need > end) { ok = false; n = end; return 0; }
        uint64_t v = b;
        if (need == 3) v = (x >> 8) & 0xFFFF;
        else if (need == 5) v = (x >> 8) & 0xFFFFFFFFull;
        else if (need == 9) { n += 1; v = peek64(); n -= 1; }
        n += need;
        return v;
    }
};

// The 16 bytes behind a frame's header in two registers: what every AcceptReply / Heartbeat / CommitNotice of a running
// cluster fits into (slots, ballots and bars below 2^32 are varints of <= 5 bytes).  Decoding out of registers has no
// window bounds to watch and no LDS round trip per varint; `n` counts the bytes taken and is compared with the frame's
// length once, at the end (an overrun anywhere makes the frame malformed either way).
struct Reg128 {
    uint64_t lo, hi;
    uint32_t n;
    bool ok;
    __device__ __forceinline__ void take(uint32_t k) {               // drop k <= 9 bytes
        const uint32_t sft = 8 * k;
        if (sft >= 64) { lo = hi >> (sft - 64); hi = 0; }
        else { lo = (lo >> sft) | (hi << (64 - sft)); hi >>= sft; }
        n += k;
    }
    __device__ __forceinline__ uint8_t byte() { const uint8_t b = (uint8_t)lo; take(1); return b; }
    __device__ __forceinline__ uint64_t varint() {
        const uint32_t b = (uint32_t)lo & 0xFF;
        const uint32_t need = b < 251 ? 1 : (0x00953u >> (4 * (b - 251))) & 0xF;    // 0xFB -> 3, 0xFC -> 5, 0xFD -> 9, 0xFE / 0xFF -> 0
        const uint64_t rest = (lo >> 8) | (hi << 56);
        const uint64_t v = need == 1 ? b : need == 3 ? (rest & 0xFFFF) : need == 5 ? (rest & 0xFFFFFFFFull) : rest;
        if (need == 0) { ok = false; n += 64; return 0; }                           // (n past any length: stays malformed)
        take(need);
        return v;
    }
};
constexpr uint32_t WI_FAST_MAX = 16;             // payloads up to this long are decoded out of a Reg128

// PeerMessage -> (kind, the hot variants' fields): smr_wire_decode's rules (csrc/wire.hip) for either reader
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

#ifndef SMR_WI_NT
#define SMR_WI_NT 0                              // 1: records leave with non-temporal stores (A/B: tools/r2u_wi_loads.sh)
#endif
template <typename T> __device__ __forceinline__ void put_record(T *dst, const T &v) {
    static_assert(sizeof(T) % 8 == 0, "records are whole 8-byte words");
#if SMR_WI_NT
    const uint64_t *p = (const uint64_t *)&v;
#pragma unroll
    for (uint32_t i = 0; i < sizeof(T) / 8; i++) __builtin_nontemporal_store(p[i], (uint64_t *)dst + i);
#else
    *dst = v;
#endif
}

struct IngestArgs {
    const uint8_t *buf; uint64_t buf_len;
    const uint64_t *conn_off; const uint32_t *conn_group; const uint8_t *conn_peer; uint32_t n_conn;
    smr_mp_ack *acks; uint64_t ack_cap;
    smr_wire_hb *hbs; uint64_t hb_cap;
    smr_wire_other *others; uint64_t other_cap;
    uint64_t *wave_cnt;                          // [n_waves][3]: counts (pass 1), then exclusive bases (the scan)
    uint32_t *lane_cnt;                          // [n_waves * 64][3]: pass 1's counts per connection
    uint64_t *counts;                            // [4]: acks, heartbeats / commit notices, others, malformed connections
    uint64_t *consumed; int32_t *status;
};

// WRITE = false: count my connection's records, report consumed / status; true: write them at my bases
template <bool WRITE>
__global__ __launch_bounds__(64) void wire_ingest_mp_kernel(IngestArgs A) {
#if SMR_WI_RING
    __shared__ uint32_t win[64 * 64];                 // two 128-byte lines per lane
#else
    __shared__ uint32_t win[(WI_DW + 2) * 64];       // + two dword rows: peek64 at the window's last bytes stays inside
#endif
    __shared__ uint32_t sh_cnt[3][64];
    const uint32_t lane = threadIdx.x, c = blockIdx.x * 64 + lane;
    const bool live = c < A.n_conn;
    const uint64_t start = live ? A.conn_off[c] : 0, end = live ? A.conn_off[c + 1] : 0;
    const uint32_t group = live ? A.conn_group[c] : 0, peer = live ? A.conn_peer[c] : 0;
    uint64_t pos = start;
    int st = 0;
    bool done = !live;
    if (live && (end < start || end > A.buf_len)) { st = 1; done = true; }
    uint64_t base[3] = {0, 0, 0};
    if (WRITE) {                                  // where my records go: my wavefront's bases + the lanes before me
#pragma unroll
        for (int k = 0; k < 3; k++) sh_cnt[k][lane] = A.lane_cnt[(size_t)c * 3 + k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; k++) {
            uint64_t b = A.wave_cnt[(size_t)blockIdx.x * 3 + k];
            for (uint32_t l = 0; l < lane; l++) b += sh_cnt[k][l];
            base[k] = b;
        }
    }
    uint32_t n[3] = {0, 0, 0};
    uint32_t *const col = &win[lane];
#if SMR_WI_RING
    // The stream as whole 128-byte lines: every line of it is fetched ONCE (a refill at the parse position re-fetches the
    // line it stands in: 2.4x the stream in line traffic), and the next line's loads are in flight while the
    // frames that end inside the loaded lines are parsed.  A frame's header + look-ahead is <= 72 bytes, so two lines of ring
    // always hold what the next frame needs once the line after its first byte is in.
    uint64_t ld = start & ~127ull;                // [the line my position is in .. ld) is in the ring
    for (;;) {
        if (!__ballot(!done)) break;
        const bool want = !done && ld < end;
        const uint64_t wbase = ld;
        const uint32_t nchunk = want ? (uint32_t)(end - ld >= 128 ? 8 : (end - ld + 15) / 16) : 0u;
        const uint32_t k = lane % 8;
        wi_u32x4 q[8];
        uint32_t ragged = 0;
        const bool can = A.buf_len >= 16;
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {                                          // the next line of eight connections per instruction
            const uint32_t src = i * 8 + lane / 8;
            const uint64_t wb = __shfl(wbase, (int)src);
            const uint32_t nc = __shfl(nchunk, (int)src);
            const uint64_t off = wb + 16ull * k;
            const bool w = k < nc, whole = w && off + 16 <= A.buf_len;
            ragged |= (uint32_t)(w && !whole) << i;
            q[i] = wi_u32x4{0, 0, 0, 0};
            if (can) q[i] = *(const wi_u32x4 *)(A.buf + (whole ? off : 0));
        }
        if (__ballot(ragged != 0)) {
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) {
                const uint64_t off = __shfl(wbase, (int)(i * 8 + lane / 8)) + 16ull * k;
                if ((ragged >> i) & 1) {
                    uint32_t v[4] = {0, 0, 0, 0};
                    for (uint32_t b = 0; b < 16 && off + b < A.buf_len; b++) v[b >> 2] |= (uint32_t)A.buf[off + b] << (8 * (b & 3));
                    q[i] = wi_u32x4{v[0], v[1], v[2], v[3]};
                }
            }
        }
        // ---- the frames that end inside the lines already in the ring (the loads above are still in flight) ----
        while (!done) {
            const uint64_t avail = end - pos;
            if (avail < 8) { done = true; break; }
            if (pos + 8 > ld) break;
            const uint32_t o = (uint32_t)(pos & 3);
            RingRd r{col, (uint32_t)(pos >> 2) & 63u, o, o + 8, true};
            const uint64_t plen = r.be64();
            if (plen > 1000000000000ull) { st = 1; done = true; break; }
            if (avail - 8 < plen) { done = true; break; }
            const uint32_t look = (uint32_t)(plen < WI_HOT_MAX ? plen : WI_HOT_MAX);
            if (pos + 8 + look > ld) break;
            uint32_t kind = SMR_WIRE_OTHER;
            bool hot = false;
            uint64_t f0 = 0, f1 = 0, f2 = 0, f3 = 0;
            if (plen <= WI_FAST_MAX) {
                const uint64_t lo = r.peek64();
                r.n += 8;
                const uint64_t hi = r.peek64();
                Reg128 p{lo, hi, 0, true};
                parse_peer_message(p, kind, hot, f0, f1, f2, f3);
                r.ok = p.ok && p.n <= (uint32_t)plen;
                r.n = o + 8 + p.n;
            } else {
                r.end = o + 8 + look;
                parse_peer_message(r, kind, hot, f0, f1, f2, f3);
            }
            if (!r.ok || (hot && (uint64_t)(r.n - (o + 8)) != plen)) { st = 1; done = true; break; }
            const int what = (kind == SMR_WIRE_ACCEPT_REPLY && f0 <= 0xFFFFFFFFull) ? 0 :
                             (kind == SMR_WIRE_HEARTBEAT || kind == SMR_WIRE_COMMIT_NOTICE) ? 1 : 2;
            if (WRITE) {
                const uint64_t at = base[what] + n[what];
                if (what == 0 && at < A.ack_cap) {
                    smr_mp_ack a; a.group = group; a.slot = (uint32_t)f0; a.ballot = f1; a.peer = peer; a.reserved = 0;
                    put_record(&A.acks[at], a);
                } else if (what == 1 && at < A.hb_cap) {
                    smr_wire_hb h; h.group = group; h.peer = peer; h.kind = kind; h.reserved = 0; h.ballot = f0; h.commit_bar = f1;
                    h.exec_bar = f2; h.snap_bar = f3;
                    put_record(&A.hbs[at], h);
                } else if (what == 2 && at < A.other_cap) {
                    smr_wire_other oo; oo.conn = c; oo.kind = kind; oo.off = pos; oo.len = 8 + plen;
                    put_record(&A.others[at], oo);
                }
            }
            n[what]++;
            pos += 8 + plen;
        }
        // ---- the line that was in flight goes into the ring -- unless my position has left everything loaded (a long frame
        // the device only locates): then the ring restarts at the line my position is in
        const uint64_t nld = want ? ld + 128 : ld;
        const bool jump = !done && pos >= nld;
        const uint32_t nc_w = (want && !jump) ? nchunk : 0u;
        __syncthreads();
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            const uint32_t src = i * 8 + lane / 8;
            const uint64_t wb = __shfl(wbase, (int)src);
            const uint32_t nc = __shfl(nc_w, (int)src);
            if (k < nc) {
                const uint32_t row = (uint32_t)(wb >> 2) + 4 * k;
                win[((row + 0) & 63u) * 64 + src] = q[i].x; win[((row + 1) & 63u) * 64 + src] = q[i].y;
                win[((row + 2) & 63u) * 64 + src] = q[i].z; win[((row + 3) & 63u) * 64 + src] = q[i].w;
            }
        }
        __syncthreads();
        ld = jump ? (pos & ~127ull) : nld;
    }
#else
    for (;;) {
        if (!__ballot(!done)) break;              // wave-uniform: a lane that is done stays to help with the refills
        // ---- refill: every lane's column gets what is left of its stream at its position, at most the window ----
        const uint64_t wbase = pos & ~15ull;
        const uint64_t left = done ? 0 : end - wbase;
        const uint32_t nchunk = (uint32_t)(left >= WI_WIN ? WI_WIN / 16 : (left + 15) / 16);
#if SMR_WI_COOP
        // the wavefront loads together: WI_WIN / 16 neighbouring lanes take one connection's window, one 16-byte chunk
        // each, so an instruction reads whole runs of WI_WIN contiguous bytes (a lane loading its own window alone
        // touches the same 128-byte line from 8 instructions, 64 different lines per instruction)
        constexpr uint32_t CPW = WI_WIN / 16, PER = 64 / CPW;                       // lanes per connection; connections per instruction
        __syncthreads();                                                            // (one wavefront per block) every lane has left the old window
        // every chunk's load is issued before the first one is waited for (a load inside `if (k < nc)` is waited for inside
        // it: CPW memory round trips in a row per refill); a chunk that is not wanted, or not whole, reads the buffer's
        // first 16 bytes instead
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
#else
#pragma unroll 8
        for (uint32_t k = 0; k < WI_WIN / 16; k++) {
            if (k >= nchunk) break;
            const uint64_t off = wbase + 16ull * k;
            uint32_t v[4] = {0, 0, 0, 0};
            if (off + 16 <= A.buf_len) {
                const wi_u32x4 q = *(const wi_u32x4 *)(A.buf + off);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
                for (uint32_t i = 0; i < 16 && off + i < A.buf_len; i++) v[i >> 2] |= (uint32_t)A.buf[off + i] << (8 * (i & 3));
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) col[(4 * k + j) * 64] = v[j];
        }
#endif
        // ---- the frames that lie inside the window -----------------------------------------------------
        while (!done) {
            const uint64_t avail = end - pos;
            if (avail < 8) { done = true; break; }                                  // length not complete yet
            const uint32_t woff = (uint32_t)(pos - wbase);
            if (woff + 8 > WI_WIN) break;
            WinRd r{col, woff, woff + 8, true};
            const uint64_t plen = r.be64();
            if (plen > 1000000000000ull) { st = 1; done = true; break; }            // safetcp.rs:56-66
            if (avail - 8 < plen) { done = true; break; }                           // frame not complete yet
            const uint32_t look = (uint32_t)(plen < WI_HOT_MAX ? plen : WI_HOT_MAX);
            if (woff + 8 + look > WI_WIN) break;                                    // (after a refill woff < 16: always fits)
            uint32_t kind = SMR_WIRE_OTHER;
            bool hot = false;
            uint64_t f0 = 0, f1 = 0, f2 = 0, f3 = 0;
            if (plen <= WI_FAST_MAX) {                                              // the whole payload in two registers
                const uint32_t p = woff + 8, i = p >> 2, sh = 8 * (p & 3);
                uint32_t d[5];
#pragma unroll
                for (uint32_t j = 0; j < 5; j++) d[j] = col[(i + j < WI_DW + 1 ? i + j : WI_DW + 1) * 64];   // (bytes past the frame: any)
                uint32_t w[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) w[j] = (uint32_t)((((uint64_t)d[j + 1] << 32) | d[j]) >> sh);
                Reg128 q{((uint64_t)w[1] << 32) | w[0], ((uint64_t)w[3] << 32) | w[2], 0, true};
                parse_peer_message(q, kind, hot, f0, f1, f2, f3);
                r.ok = q.ok && q.n <= (uint32_t)plen;
                r.n = p + q.n;
            } else {
                r.end = woff + 8 + look;
                parse_peer_message(r, kind, hot, f0, f1, f2, f3);
            }
            // a frame whose leading varints do not parse, or a hot frame that does not end where its length says
            if (!r.ok || (hot && (uint64_t)(r.n - (woff + 8)) != plen)) { st = 1; done = true; break; }
            // (an AcceptReply for a slot the engine cannot name -- its slots are u32 -- goes the host's way)
            const int what = (kind == SMR_WIRE_ACCEPT_REPLY && f0 <= 0xFFFFFFFFull) ? 0 :
                             (kind == SMR_WIRE_HEARTBEAT || kind == SMR_WIRE_COMMIT_NOTICE) ? 1 : 2;
            if (WRITE) {
                const uint64_t at = base[what] + n[what];
                if (what == 0 && at < A.ack_cap) {
                    smr_mp_ack a; a.group = group; a.slot = (uint32_t)f0; a.ballot = f1; a.peer = peer; a.reserved = 0;
                    put_record(&A.acks[at], a);
                } else if (what == 1 && at < A.hb_cap) {
                    smr_wire_hb h; h.group = group; h.peer = peer; h.kind = kind; h.reserved = 0; h.ballot = f0; h.commit_bar = f1;
                    h.exec_bar = f2; h.snap_bar = f3;
                    put_record(&A.hbs[at], h);
                } else if (what == 2 && at < A.other_cap) {
                    smr_wire_other o; o.conn = c; o.kind = kind; o.off = pos; o.len = 8 + plen;
                    put_record(&A.others[at], o);
                }
            }
            n[what]++;
            pos += 8 + plen;
        }
    }
#endif
    if (!WRITE) {
        if (live) { A.consumed[c] = pos - start; A.status[c] = st; }
#pragma unroll
        for (int k = 0; k < 3; k++) { A.lane_cnt[(size_t)c * 3 + k] = n[k]; sh_cnt[k][lane] = n[k]; }
        const unsigned long long bad = __ballot(st != 0);
        __syncthreads();
        if (lane < 3) {
            uint64_t t = 0;
            for (uint32_t l = 0; l < 64; l++) t += sh_cnt[lane][l];
            A.wave_cnt[(size_t)blockIdx.x * 3 + lane] = t;
        }
        if (lane == 0 && bad) atomicAdd((unsigned long long *)&A.counts[3], (unsigned long long)__popcll(bad));
    }
}

// the wavefronts' counts -> exclusive bases, totals into counts[0 .. 3): one block, 256 lanes, a chunk of rows per lane
__global__ __launch_bounds__(256) void wire_ingest_scan_kernel(uint64_t *__restrict__ wave_cnt, uint32_t n_waves, uint64_t *__restrict__ counts) {
    __shared__ uint64_t part[3][256];
    const uint32_t t = threadIdx.x, per = (n_waves + 255) / 256;
    const uint32_t lo = t * per < n_waves ? t * per : n_waves, hi = lo + per < n_waves ? lo + per : n_waves;
    uint64_t s[3] = {0, 0, 0};
    for (uint32_t w0 = lo; w0 < hi; w0 += 8) {                                      // eight rows' loads in flight together
        uint64_t x[8][3];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++)
#pragma unroll
            for (int k = 0; k < 3; k++) x[j][k] = wave_cnt[(size_t)(w0 + j < hi ? w0 + j : lo) * 3 + k];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++)
#pragma unroll
            for (int k = 0; k < 3; k++) s[k] += w0 + j < hi ? x[j][k] : 0;
    }
    for (int k = 0; k < 3; k++) part[k][t] = s[k];
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {                                  // inclusive scan of the 256 partial sums
        uint64_t v[3] = {0, 0, 0};
        if (t >= off) for (int k = 0; k < 3; k++) v[k] = part[k][t - off];
        __syncthreads();
        if (t >= off) for (int k = 0; k < 3; k++) part[k][t] += v[k];
        __syncthreads();
    }
    uint64_t b[3];
    for (int k = 0; k < 3; k++) b[k] = part[k][t] - s[k];                           // exclusive
    for (uint32_t w0 = lo; w0 < hi; w0 += 8) {
        uint64_t x[8][3];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++)
#pragma unroll
            for (int k = 0; k < 3; k++) x[j][k] = wave_cnt[(size_t)(w0 + j < hi ? w0 + j : lo) * 3 + k];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            if (w0 + j < hi) {
#pragma unroll
                for (int k = 0; k < 3; k++) { wave_cnt[(size_t)(w0 + j) * 3 + k] = b[k]; b[k] += x[j][k]; }
            }
        }
    }
    if (t == 255) for (int k = 0; k < 3; k++) counts[k] = b[k] ;
}

}  // namespace smr

using namespace smr;

extern "C" {

uint64_t smr_wire_ingest_scratch_bytes(uint32_t n_conn) {
    const uint64_t n_waves = ((uint64_t)n_conn + 63) / 64;
    return n_waves * 3 * 8 + n_waves * 64 * 3 * 4;
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
    const uint32_t n_waves = (n_conn + 63) / 64;
    IngestArgs A{buf_dev, buf_len, conn_off_dev, conn_group_dev, conn_peer_dev, n_conn, acks_dev, ack_cap, hbs_dev, hb_cap,
                 others_dev, other_cap, (uint64_t *)scratch_dev, (uint32_t *)((uint64_t *)scratch_dev + (size_t)n_waves * 3),
                 counts_dev, consumed_dev, status_dev};
    hipLaunchKernelGGL(wire_ingest_mp_kernel<false>, dim3(n_waves), dim3(64), 0, st, A);
    hipLaunchKernelGGL(wire_ingest_scan_kernel, dim3(1), dim3(256), 0, st, A.wave_cnt, n_waves, counts_dev);
    hipLaunchKernelGGL(wire_ingest_mp_kernel<true>, dim3(n_waves), dim3(64), 0, st, A);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

}  // extern "C"

// Device-side ingest of the reply traffic of Raft and EPaxos (VERDICT r2 #7's second half).
// The MultiPaxos leader takes its AcceptReplies as a LIST of records (csrc/wire_ingest.hip); the Raft leader and the
// EPaxos command leader take ONE reply per (peer, group) and call, as arrays [R][G] -- smr_raft_leader_handle_replies /
// smr_raft_tick, smr_ep_handle_pre_accept_replies -- so here the parser writes those arrays directly:
//   Raft    PeerMsg::AppendEntriesReply { term, end_slot, conflict: Option<(Term, usize)> }   raft/mod.rs:203-234 (variant 1)
//   EPaxos  PeerMsg::PreAcceptReply { slot: SlotIdx(row, col), ballot, seq, deps }             epaxos/mod.rs:306-377 (variant 1)
//   RSPaxos PeerMsg::AcceptReply { slot, ballot }                                               rspaxos/mod.rs:262-305 (variant 3)
// each inside `[u64 BE length][bincode(PeerMessage::Msg { msg })]` (safetcp.rs:30-70, 127-132), with the rules of the
// host's smr_wire_raft_decode / smr_wire_ep_decode (csrc/wire.hip) restated for a lane.
//
// One lane per connection = (group, peer).  A connection's frames are walked in order: the FIRST reply of the wanted kind
// fills the arrays at [peer][group]; the walk stops in front of a second one (the engines take one per call: the host hands
// the rest of the stream to the next call, consumed[c] says where it starts); every other frame -- and a reply the arrays
// cannot hold: a slot beyond u32, a PreAcceptReply for another instance than (me, col[group]) or with another number of
// dependencies than R -- is located for the host (smr_wire_other, in no particular order: one atomic per such frame, they
// are rare).  The streams of a tick are a few dozen bytes per connection (one reply, now and then a heartbeat) and the
// connections of a block lie side by side in the buffer: the block copies its span of the buffer into LDS with aligned
// 16-byte loads (up to 24 KB; a block whose connections carry more reads the rest from HBM) and every lane parses out of
// LDS -- the first version read its frames straight out of HBM eight unaligned bytes at a time, eight dependent loads per
// frame: 105 us for 262 144 connections (profiles/round3/r4c, r4e), four Raft ticks' worth.  What those 105 us were, though, was
// neither the loads nor the parse (r4g): 8192 atomics on ONE address -- the wavefronts' reply counts and the located frames'
// indices; a same-address atomic takes ~10 ns whatever it carries -- so a block of 1024 lanes now counts in LDS, keeps the
// frames it locates in LDS, and goes to the call's counters once: 256 atomics per counter.
#include "smr_common.h"
#include "wire_rd.h"

namespace smr {

struct ReplyArgs {
    const uint8_t *buf; uint64_t buf_len;
    const uint64_t *conn_off; const uint32_t *conn_group; const uint8_t *conn_peer; uint32_t n_conn;
    const uint8_t *conn_len;                     // NULL: connection c ends where c + 1 starts; else its bytes are conn_off[c] .. + conn_len[c]
    uint32_t G, R;
    // Raft: reply_term, end_slot, conflict_term, conflict_slot, flags.  EPaxos: ballot (in a), seq (in c), deps, flags; me, col
    uint64_t *a; uint32_t *b; uint64_t *c; uint32_t *d; uint8_t *flags;
    uint32_t me; const uint32_t *col;
    smr_wire_other *others; uint64_t other_cap;
    uint64_t *counts;                            // [4]: replies taken, frames located, malformed connections, connections stopped in front of a second reply
    uint64_t *consumed; int32_t *status;
};

constexpr int WR_RAFT = 0, WR_EPAXOS = 1, WR_RSPAXOS = 2;
constexpr uint32_t WR_EMAXR = 8;

template <int PROTO>
__global__ __launch_bounds__(WR_BLOCK) void wire_ingest_replies_kernel(ReplyArgs A) {
    __shared__ uint32_t stage[WR_STAGE / 4 + 4];
    __shared__ smr_wire_other loc[WR_LOC];
    __shared__ uint32_t blk[4];                                                     // replies taken, frames located, malformed, deferred: this block's
    __shared__ unsigned long long loc_base;
    if (threadIdx.x < 4) blk[threadIdx.x] = 0;
    const uint32_t c = blockIdx.x * WR_BLOCK + threadIdx.x;
    const bool live = c < A.n_conn;
    const uint64_t start = live ? A.conn_off[c] : 0, end = !live ? 0 : A.conn_len ? start + A.conn_len[c] : A.conn_off[c + 1];
    // my block's span of the buffer -> LDS: from its first connection's start (16-byte aligned down) as far as the stage goes
    const uint32_t c0 = blockIdx.x * WR_BLOCK, c1 = c0 + WR_BLOCK < A.n_conn ? c0 + WR_BLOCK : A.n_conn;
    const uint64_t s0 = A.conn_off[c0] & ~15ull, s1 = A.conn_len ? A.conn_off[c1 - 1] + A.conn_len[c1 - 1] : A.conn_off[c1];
    uint64_t slo = 0, shi = 0;
    if (s0 < s1 && s1 <= A.buf_len) {
        slo = s0; shi = s1 - s0 <= WR_STAGE ? s1 : s0 + WR_STAGE;
        const uint32_t n16 = (uint32_t)((shi - slo + 15) / 16);
        for (uint32_t i = threadIdx.x; i <= n16; i += WR_BLOCK) {                  // (one chunk more: the dwords a read at the last bytes reaches into)
            const uint64_t off = slo + 16ull * i;
            uint32_t v[4] = {0, 0, 0, 0};
            if (off + 16 <= A.buf_len) {
                const wr_u32x4 q = *(const wr_u32x4 *)(A.buf + off);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
                for (uint32_t b = 0; b < 16 && off + b < A.buf_len; b++) v[b >> 2] |= (uint32_t)A.buf[off + b] << (8 * (b & 3));
            }
            if (4 * i + 3 < WR_STAGE / 4 + 4) { stage[4 * i] = v[0]; stage[4 * i + 1] = v[1]; stage[4 * i + 2] = v[2]; stage[4 * i + 3] = v[3]; }
        }
    }
    __syncthreads();
    const uint32_t g = live ? A.conn_group[c] : 0, peer = live ? A.conn_peer[c] : 0;
    uint64_t pos = start;
    int st = 0;
    bool have = false, deferred = false;
    if (live && (end < start || end > A.buf_len || g >= A.G || peer >= A.R)) st = 1;   // (a connection the arrays have no place for: nothing of it is taken)
    while (live && st == 0) {
        const uint64_t avail = end - pos;
        if (avail < 8) break;                                                       // length not complete yet
        GlRd r{A.buf, pos, pos + 8, A.buf_len, true, stage, slo, shi};
        const uint64_t plen = __builtin_bswap64(r.peek64());
        if (plen > 1000000000000ull) { st = 1; break; }                             // safetcp.rs:56-66
        if (avail - 8 < plen) break;                                                // frame not complete yet
        r.n = pos + 8; r.end = pos + 8 + plen;
        uint32_t kind = SMR_WIRE_OTHER;
        bool mine = false;                                                          // a reply the arrays take
        const uint64_t outer = r.varint();
        if (outer == 2) kind = SMR_WIRE_LEAVE;                                      // PeerMessage::Leave
        else if (outer == 0) {                                                      // PeerMessage::Msg { msg }
            const uint64_t v = r.varint();
            if (PROTO == WR_RAFT) {
                if (!r.ok || v > 3) { st = 1; break; }                              // smr_wire_raft_decode: unknown Raft message
                kind = (uint32_t)v;
                if (v == 1) {                                                       // AppendEntriesReply
                    uint64_t term, end_slot, ct, cs;
                    uint8_t has;
                    if (!wr_raft_append_reply(r, term, end_slot, has, ct, cs)) { st = 1; break; }
                    mine = end_slot <= 0xFFFFFFFFull && cs <= 0xFFFFFFFFull;        // (the engine's slots are u32: a wider one goes the host's way)
                    if (mine) {
                        if (have) { deferred = true; break; }                       // the next call's
                        const size_t i = (size_t)peer * A.G + g;
                        A.a[i] = term; A.b[i] = (uint32_t)end_slot; A.c[i] = ct; A.d[i] = (uint32_t)cs;
                        A.flags[i] = (uint8_t)(1u | (has ? 2u : 0u));
                        have = true;
                    }
                }
            } else if (PROTO == WR_RSPAXOS) {
                if (!r.ok) { st = 1; break; }
                kind = (uint32_t)(v <= SMR_WIRE_RSP_HEARTBEAT ? v : SMR_WIRE_OTHER);
                if (v == SMR_WIRE_ACCEPT_REPLY) {                                   // AcceptReply { slot, ballot }: smr_rsp_handle_accept_replies' arrays
                    const uint64_t slot = r.varint(), ballot = r.varint();
                    if (!r.ok || r.n != r.end) { st = 1; break; }
                    mine = slot <= 0xFFFFFFFFull;
                    if (mine) {
                        if (have) { deferred = true; break; }
                        const size_t i = (size_t)peer * A.G + g;
                        A.a[i] = ballot; A.b[i] = (uint32_t)slot;
                        A.flags[i] = 1;
                        have = true;
                    }
                }
            } else {
                if (!r.ok) { st = 1; break; }
                kind = (uint32_t)(v <= SMR_WIRE_EP_COMMIT_NOTICE ? v : SMR_WIRE_OTHER);   // (ExpPrepare & co, Heartbeat: located)
                if (v == SMR_WIRE_EP_PRE_ACCEPT_REPLY) {
                    const uint8_t row = r.byte();
                    const uint64_t col = r.varint(), ballot = r.varint(), seq = r.varint(), n = r.varint();
                    if (!r.ok || n > 64) { st = 1; break; }
                    uint32_t dep[WR_EMAXR];
                    bool fits = true;
                    for (uint64_t q = 0; q < n && r.ok; q++) {
                        const uint8_t some = r.byte();
                        uint64_t x = SMR_EP_NONE;
                        if (some == 1) { x = r.varint(); fits = fits && x < SMR_EP_NONE; } else if (some != 0) r.ok = false;
                        if (q < WR_EMAXR) dep[q] = (uint32_t)x;
                    }
                    if (!r.ok || r.n != r.end) { st = 1; break; }
                    mine = row == A.me && col == (uint64_t)A.col[g] && n == A.R && fits;
                    if (mine) {
                        if (have) { deferred = true; break; }
                        const size_t i = (size_t)peer * A.G + g;
                        A.a[i] = ballot; A.c[i] = seq;
                        for (uint32_t q = 0; q < A.R; q++) A.d[((size_t)peer * A.R + q) * A.G + g] = dep[q];
                        A.flags[i] = 1;
                        have = true;
                    }
                }
            }
        }
        if (!r.ok) { st = 1; break; }                                               // (the enum tags did not parse)
        if (!mine) {                                                                // located, not validated
            smr_wire_other o; o.conn = c; o.kind = kind; o.off = pos; o.len = 8 + plen;
            const uint32_t k = atomicAdd(&blk[1], 1u);
            if (k < WR_LOC) loc[k] = o;
            else {                                                                  // (more than the block keeps: straight to the call's list)
                const unsigned long long at = atomicAdd((unsigned long long *)&A.counts[1], 1ull);
                if (at < A.other_cap) A.others[at] = o;
            }
        }
        pos += 8 + plen;
    }
    if (live) {
        A.consumed[c] = st ? 0 : pos - start;                                       // (a malformed connection: the host looks at all of it)
        A.status[c] = st;
    }
    // the counters: per wavefront into LDS, per block to the call's
    const unsigned long long m0 = __ballot(have && st == 0), m2 = __ballot(st != 0), m3 = __ballot(deferred);
    if (threadIdx.x % 64 == 0) {
        if (m0) atomicAdd(&blk[0], (uint32_t)__popcll(m0));
        if (m2) atomicAdd(&blk[2], (uint32_t)__popcll(m2));
        if (m3) atomicAdd(&blk[3], (uint32_t)__popcll(m3));
    }
    __syncthreads();
    const uint32_t n_kept = blk[1] < WR_LOC ? blk[1] : WR_LOC;
    if (threadIdx.x == 0) {
        if (blk[0]) atomicAdd((unsigned long long *)&A.counts[0], (unsigned long long)blk[0]);
        if (blk[2]) atomicAdd((unsigned long long *)&A.counts[2], (unsigned long long)blk[2]);
        if (blk[3]) atomicAdd((unsigned long long *)&A.counts[3], (unsigned long long)blk[3]);
        loc_base = n_kept ? atomicAdd((unsigned long long *)&A.counts[1], (unsigned long long)n_kept) : 0ull;
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < n_kept; j += WR_BLOCK)
        if (loc_base + j < A.other_cap) A.others[loc_base + j] = loc[j];
}

static int reply_args_ok(const uint8_t *buf, uint64_t buf_len, const uint64_t *off, const uint32_t *grp, const uint8_t *peer, uint32_t n_conn,
                         uint32_t G, uint8_t R, const void *a, const void *b, const void *c, const void *d, const void *flags,
                         const void *others, uint64_t other_cap, const void *counts, const void *consumed, const void *status) {
    if ((n_conn && (!off || !grp || !peer)) || !a || !b || !c || !d || !flags || !counts || !consumed || !status || (buf_len && !buf) ||
        (other_cap && !others))
        return fail(SMR_ERR_ARG, "wire reply ingest: null argument");
    if (G == 0 || R == 0 || R > WR_EMAXR) return fail(SMR_ERR_ARG, "wire reply ingest: n_groups / population out of range");
    if ((uintptr_t)buf & 15) return fail(SMR_ERR_ARG, "wire reply ingest: the byte buffer must be 16-byte aligned");
    return SMR_OK;
}

}  // namespace smr

using namespace smr;

extern "C" {

int smr_wire_ingest_raft_replies(const uint8_t *buf_dev, uint64_t buf_len, const uint64_t *conn_off_dev, const uint32_t *conn_group_dev,
                                 const uint8_t *conn_peer_dev, const uint8_t *conn_len_dev, uint32_t n_conn, uint32_t n_groups, uint8_t population,
                                 uint64_t *reply_term_dev, uint32_t *end_slot_dev, uint64_t *conflict_term_dev, uint32_t *conflict_slot_dev,
                                 uint8_t *flags_dev, smr_wire_other *others_dev, uint64_t other_cap, uint64_t *counts_dev,
                                 uint64_t *consumed_dev, int32_t *status_dev, void *stream) {
    const int rc = reply_args_ok(buf_dev, buf_len, conn_off_dev, conn_group_dev, conn_peer_dev, n_conn, n_groups, population, reply_term_dev,
                                 end_slot_dev, conflict_term_dev, conflict_slot_dev, flags_dev, others_dev, other_cap, counts_dev, consumed_dev,
                                 status_dev);
    if (rc != SMR_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    SMR_HIP_TRY(hipMemsetAsync(counts_dev, 0, 4 * 8, st));
    SMR_HIP_TRY(hipMemsetAsync(flags_dev, 0, (size_t)population * n_groups, st));
    if (n_conn == 0) return SMR_OK;
    ReplyArgs A{buf_dev, buf_len, conn_off_dev, conn_group_dev, conn_peer_dev, n_conn, conn_len_dev, n_groups, population, reply_term_dev, end_slot_dev,
                conflict_term_dev, conflict_slot_dev, flags_dev, 0, nullptr, others_dev, other_cap, counts_dev, consumed_dev, status_dev};
    hipLaunchKernelGGL(wire_ingest_replies_kernel<WR_RAFT>, dim3((n_conn + WR_BLOCK - 1) / WR_BLOCK), dim3(WR_BLOCK), 0, st, A);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_wire_ingest_rsp_accept_replies(const uint8_t *buf_dev, uint64_t buf_len, const uint64_t *conn_off_dev, const uint32_t *conn_group_dev,
                                       const uint8_t *conn_peer_dev, const uint8_t *conn_len_dev, uint32_t n_conn, uint32_t n_groups, uint8_t population, uint32_t *slot_dev,
                                       uint64_t *ballot_dev, uint8_t *flags_dev, smr_wire_other *others_dev, uint64_t other_cap,
                                       uint64_t *counts_dev, uint64_t *consumed_dev, int32_t *status_dev, void *stream) {
    const int rc = reply_args_ok(buf_dev, buf_len, conn_off_dev, conn_group_dev, conn_peer_dev, n_conn, n_groups, population, ballot_dev, slot_dev,
                                 ballot_dev, slot_dev, flags_dev, others_dev, other_cap, counts_dev, consumed_dev, status_dev);
    if (rc != SMR_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    SMR_HIP_TRY(hipMemsetAsync(counts_dev, 0, 4 * 8, st));
    SMR_HIP_TRY(hipMemsetAsync(flags_dev, 0, (size_t)population * n_groups, st));
    if (n_conn == 0) return SMR_OK;
    ReplyArgs A{buf_dev, buf_len, conn_off_dev, conn_group_dev, conn_peer_dev, n_conn, conn_len_dev, n_groups, population, ballot_dev, slot_dev,
                nullptr, nullptr, flags_dev, 0, nullptr, others_dev, other_cap, counts_dev, consumed_dev, status_dev};
    hipLaunchKernelGGL(wire_ingest_replies_kernel<WR_RSPAXOS>, dim3((n_conn + WR_BLOCK - 1) / WR_BLOCK), dim3(WR_BLOCK), 0, st, A);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_wire_ingest_ep_pre_accept_replies(const uint8_t *buf_dev, uint64_t buf_len, const uint64_t *conn_off_dev, const uint32_t *conn_group_dev,
                                          const uint8_t *conn_peer_dev, const uint8_t *conn_len_dev, uint32_t n_conn, uint32_t n_groups, uint8_t population, uint8_t me,
                                          const uint32_t *col_dev, uint64_t *ballot_dev, uint64_t *seq_dev, uint32_t *deps_dev,
                                          uint8_t *flags_dev, smr_wire_other *others_dev, uint64_t other_cap, uint64_t *counts_dev,
                                          uint64_t *consumed_dev, int32_t *status_dev, void *stream) {
    const int rc = reply_args_ok(buf_dev, buf_len, conn_off_dev, conn_group_dev, conn_peer_dev, n_conn, n_groups, population, ballot_dev,
                                 deps_dev, seq_dev, deps_dev, flags_dev, others_dev, other_cap, counts_dev, consumed_dev, status_dev);
    if (rc != SMR_OK) return rc;
    if (!col_dev || me >= population) return fail(SMR_ERR_ARG, "wire reply ingest: the instances' columns / my replica id");
    hipStream_t st = (hipStream_t)stream;
    SMR_HIP_TRY(hipMemsetAsync(counts_dev, 0, 4 * 8, st));
    SMR_HIP_TRY(hipMemsetAsync(flags_dev, 0, (size_t)population * n_groups, st));
    if (n_conn == 0) return SMR_OK;
    ReplyArgs A{buf_dev, buf_len, conn_off_dev, conn_group_dev, conn_peer_dev, n_conn, conn_len_dev, n_groups, population, ballot_dev, nullptr,
                seq_dev, deps_dev, flags_dev, me, col_dev, others_dev, other_cap, counts_dev, consumed_dev, status_dev};
    hipLaunchKernelGGL(wire_ingest_replies_kernel<WR_EPAXOS>, dim3((n_conn + WR_BLOCK - 1) / WR_BLOCK), dim3(WR_BLOCK), 0, st, A);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

}  // extern "C"

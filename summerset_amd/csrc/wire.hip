// Wire + WAL formats of the MultiPaxos hot-path messages: what a host needs
// to feed the engine from, and answer to, real Summerset peers.  Host-only.
//
// Frame, both on TCP and in the log file: 8-byte BIG-endian length, then the bincode 2.0
// "standard"-config bytes of the object (src/utils/safetcp.rs:46,127-132; src/server/storage.rs:
// 326-346).  On TCP the object is `PeerMessage::Msg { msg: PeerMsg }` (src/server/transport.rs:
// 37-52, variant 0) around the protocol's `PeerMsg` (src/protocols/multipaxos/mod.rs:298-368:
// Prepare 0, PrepareReply 1, Accept 2, AcceptReply 3); in the log it is `WalEntry`
// (mod.rs:261-274: PrepareBal 0, AcceptData 1, CommitSlot 2).  `ReqBatch` =
// Vec<(ClientId, ApiRequest)> (src/server/external.rs:33-54: Req 0; src/server/statemach.rs:21-27:
// Get 0, Put 1).
//
// Raft (src/protocols/raft/mod.rs:117-234): PeerMsg { AppendEntries 0, AppendEntriesReply 1, RequestVote 2,
// RequestVoteReply 3 }, LogEntry { term, reqs, external: bool, log_offset }, DurEntry { Metadata 0 { curr_term,
// voted_for: u8 (None = 255) }, LogEntry 1 }.
//
// bincode "standard": little-endian varint integers (< 251: one byte; 0xFB + u16; 0xFC + u32;
// 0xFD + u64), enum variant index as varint u32, Option as a 0/1 byte, String / Vec as varint
// length + elements, struct / tuple fields in declaration order.  bincode is not vendored with the
// reference: this layout is restated and unpinned against the crate.
#include <string.h>

#include <deque>
#include <string>
#include <vector>

#include "smr_common.h"

namespace smr {

struct Wr {
    uint8_t *p; uint64_t cap, n = 0; bool ok = true;
    void byte(uint8_t b) { if (n < cap) p[n] = b; else ok = false; n++; }
    void raw(const void *src, uint64_t len) {
        if (n <= cap && len <= cap - n) { if (len) memcpy(p + n, src, len); } else ok = false;
        n += len;
    }
    void le(uint64_t v, int bytes) { for (int i = 0; i < bytes; i++) byte((uint8_t)(v >> (8 * i))); }
    void varint(uint64_t v) {
        if (v < 251) byte((uint8_t)v);
        else if (v < (1ull << 16)) { byte(0xFB); le(v, 2); }
        else if (v < (1ull << 32)) { byte(0xFC); le(v, 4); }
        else { byte(0xFD); le(v, 8); }
    }
    void bytes(const void *src, uint64_t len) { varint(len); raw(src, len); }      // String / Vec<u8>
};

struct Rd {
    const uint8_t *p; uint64_t len, n = 0; bool ok = true;
    uint8_t byte() { if (n < len) return p[n++]; ok = false; return 0; }
    uint64_t le(int bytes) { uint64_t v = 0; for (int i = 0; i < bytes; i++) v |= (uint64_t)byte() << (8 * i); return v; }
    uint64_t varint() {
        const uint8_t b = byte();
        if (b < 251) return b;
        if (b == 0xFB) return le(2);
        if (b == 0xFC) return le(4);
        if (b == 0xFD) return le(8);
        ok = false;                                                              // 0xFE (u128) / 0xFF never occur here
        return 0;
    }
    // k comes off the wire (a peer's varint, up to 2^64 - 1): compare against the bytes LEFT, never n + k
    bool skip(uint64_t k) { if (ok && k <= len - n) { n += k; return true; } ok = false; return false; }
    // an element count: every element of every Vec on this path is >= 1 byte, so a count above the bytes left is malformed
    uint64_t count() { const uint64_t c = varint(); if (!ok || c > len - n) { ok = false; return 0; } return c; }
};

// finishes a frame whose payload was written at out + 8
static int64_t frame_done(Wr &w, uint8_t *out) {
    if (!w.ok) return fail(SMR_ERR_ARG, "wire: output buffer too small");
    const uint64_t len = w.n - 8;
    for (int i = 0; i < 8; i++) out[i] = (uint8_t)(len >> (8 * (7 - i)));          // big-endian length
    return (int64_t)w.n;
}
static Wr frame_begin(uint8_t *out, uint64_t cap) {
    Wr w{out, cap};
    for (int i = 0; i < 8; i++) w.byte(0);
    return w;
}

// walks one bincode ReqBatch starting at r.n; false if malformed
static bool skip_reqbatch(Rd &r) {
    const uint64_t n = r.count();
    for (uint64_t i = 0; i < n && r.ok; i++) {
        r.varint();                                                              // ClientId
        const uint64_t req = r.varint();                                         // ApiRequest variant
        if (req == 0) {                                                          // Req { id, cmd }
            r.varint();
            const uint64_t cmd = r.varint();
            if (cmd > 1) { r.ok = false; break; }
            r.skip(r.varint());                                                  // key
            if (cmd == 1) r.skip(r.varint());                                    // value
        } else if (req == 2) {                                                   // Leave
        } else { r.ok = false; }                                                 // Conf: not on this path
    }
    return r.ok;
}

}  // namespace smr

using namespace smr;

extern "C" {

int64_t smr_wire_reqbatch(uint32_t n, const uint64_t *client, const uint64_t *req_id, const uint8_t *kind,
                          const char *const *key, const uint32_t *key_len, const char *const *value,
                          const uint32_t *value_len, uint8_t *out, uint64_t cap) {
    if (!client || !req_id || !kind || !key || !key_len || (!out && cap)) return fail(SMR_ERR_ARG, "wire: null argument");
    Wr w{out, cap};
    w.varint(n);
    for (uint32_t i = 0; i < n; i++) {
        if (kind[i] > SMR_CMD_PUT) return fail(SMR_ERR_ARG, "wire: unknown command kind");
        w.varint(client[i]);
        w.varint(0);                                                             // ApiRequest::Req
        w.varint(req_id[i]);
        w.varint(kind[i]);                                                       // Command::{Get, Put}
        w.bytes(key[i], key_len[i]);
        if (kind[i] == SMR_CMD_PUT) {
            if (!value || !value_len) return fail(SMR_ERR_ARG, "wire: Put without value arrays");
            w.bytes(value[i], value_len[i]);
        }
    }
    if (!w.ok) return fail(SMR_ERR_ARG, "wire: output buffer too small");
    return (int64_t)w.n;
}

int64_t smr_wire_prepare(uint64_t trigger_slot, uint64_t ballot, uint8_t *out, uint64_t cap) {
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(SMR_WIRE_PREPARE); w.varint(trigger_slot); w.varint(ballot);
    return frame_done(w, out);
}

int64_t smr_wire_prepare_reply(uint64_t slot, uint64_t trigger_slot, uint64_t endprep_slot, uint64_t ballot,
                               int has_voted, uint64_t voted_ballot, const uint8_t *voted_reqs, uint64_t voted_reqs_len,
                               uint64_t accept_bar, uint8_t *out, uint64_t cap) {
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(SMR_WIRE_PREPARE_REPLY);
    w.varint(slot); w.varint(trigger_slot); w.varint(endprep_slot); w.varint(ballot);
    w.byte(has_voted ? 1 : 0);                                                   // Option<(Ballot, ReqBatch)>
    if (has_voted) { w.varint(voted_ballot); w.raw(voted_reqs, voted_reqs_len); }
    w.varint(accept_bar);
    return frame_done(w, out);
}

int64_t smr_wire_accept(uint64_t slot, uint64_t ballot, const uint8_t *reqs, uint64_t reqs_len, uint8_t *out,
                        uint64_t cap) {
    if (!reqs && reqs_len) return fail(SMR_ERR_ARG, "wire: null request batch");
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(SMR_WIRE_ACCEPT); w.varint(slot); w.varint(ballot);
    w.raw(reqs, reqs_len);                                                       // already bincode(ReqBatch)
    return frame_done(w, out);
}

int64_t smr_wire_accept_reply(uint64_t slot, uint64_t ballot, uint8_t *out, uint64_t cap) {
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(SMR_WIRE_ACCEPT_REPLY); w.varint(slot); w.varint(ballot);
    w.byte(0);                                                                   // reply_ts: None
    return frame_done(w, out);
}

int64_t smr_wire_read_query(const uint8_t *reads, uint64_t reads_len, uint8_t *out, uint64_t cap) {
    if (!reads || !reads_len) return fail(SMR_ERR_ARG, "wire: null read batch");
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(SMR_WIRE_READ_QUERY); w.raw(reads, reads_len);
    return frame_done(w, out);
}

int64_t smr_wire_read_query_reply(uint64_t rq_client, uint64_t rq_req_id, uint32_t n, const uint8_t *state, const uint64_t *slot,
                                  const char *const *value, const uint32_t *value_len, int from_leader, uint8_t *out, uint64_t cap) {
    if (n && (!state || !slot || !value || !value_len)) return fail(SMR_ERR_ARG, "wire: null reply arrays");
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(SMR_WIRE_READ_QUERY_REPLY);
    w.varint(rq_client); w.varint(rq_req_id);                                     // rq_id: (ClientId, RequestId)
    w.varint(n);                                                                  // Vec<Option<(usize, Option<String>)>>
    for (uint32_t i = 0; i < n; i++) {
        if (state[i] > 2) return fail(SMR_ERR_ARG, "wire: unknown reply state");
        if (state[i] == 0) { w.byte(0); continue; }
        w.byte(1); w.varint(slot[i]);
        if (state[i] == 1) { w.byte(0); continue; }
        w.byte(1); w.bytes(value[i], value_len[i]);
    }
    w.byte(from_leader ? 1 : 0);
    return frame_done(w, out);
}

int64_t smr_wire_heartbeat(uint64_t ballot, uint64_t commit_bar, uint64_t exec_bar, uint64_t snap_bar, uint8_t *out, uint64_t cap) {
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(SMR_WIRE_HEARTBEAT); w.varint(ballot); w.varint(commit_bar); w.varint(exec_bar); w.varint(snap_bar);
    return frame_done(w, out);
}

int64_t smr_wire_commit_notice(uint64_t ballot, uint64_t commit_bar, uint8_t *out, uint64_t cap) {
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(SMR_WIRE_COMMIT_NOTICE); w.varint(ballot); w.varint(commit_bar);
    return frame_done(w, out);
}

// the replies of a decoded ReadQueryReply: p = buf + replies_off, len = replies_len (smr_wire_msg)
int64_t smr_wire_read_query_replies(const uint8_t *p, uint64_t len, uint32_t max, uint8_t *state, uint64_t *slot, uint64_t *value_off,
                                    uint64_t *value_len) {
    if (!p || !state || !slot || !value_off || !value_len) return fail(SMR_ERR_ARG, "wire: null argument");
    Rd r{p, len};
    const uint64_t n = r.count();
    if (!r.ok || n > max) return fail(SMR_ERR_ARG, "wire: more replies than the caller has room for");
    for (uint64_t i = 0; i < n && r.ok; i++) {
        state[i] = 0; slot[i] = 0; value_off[i] = 0; value_len[i] = 0;
        const uint8_t t = r.byte();
        if (t == 0) continue;
        if (t != 1) { r.ok = false; break; }
        state[i] = 1; slot[i] = r.varint();
        const uint8_t tv = r.byte();
        if (tv == 0) continue;
        if (tv != 1) { r.ok = false; break; }
        state[i] = 2; value_len[i] = r.varint(); value_off[i] = r.n;
        r.skip(value_len[i]);
    }
    if (!r.ok || r.n != len) return fail(SMR_ERR_ARG, "wire: malformed replies");
    return (int64_t)n;
}

int64_t smr_wal_prepare_bal(uint64_t slot, uint64_t ballot, uint8_t *out, uint64_t cap) {
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(slot); w.varint(ballot);
    return frame_done(w, out);
}

int64_t smr_wal_accept_data(uint64_t slot, uint64_t ballot, const uint8_t *reqs, uint64_t reqs_len, uint8_t *out,
                            uint64_t cap) {
    if (!reqs && reqs_len) return fail(SMR_ERR_ARG, "wire: null request batch");
    Wr w = frame_begin(out, cap);
    w.varint(1); w.varint(slot); w.varint(ballot); w.raw(reqs, reqs_len);
    return frame_done(w, out);
}

int64_t smr_wal_commit_slot(uint64_t slot, uint8_t *out, uint64_t cap) {
    Wr w = frame_begin(out, cap);
    w.varint(2); w.varint(slot);
    return frame_done(w, out);
}

int64_t smr_wire_decode(const uint8_t *buf, uint64_t len, smr_wire_msg *m) {
    if (!buf || !m) return fail(SMR_ERR_ARG, "wire: null argument");
    memset(m, 0, sizeof(*m));
    if (len < 8) return 0;                                                       // length not complete yet
    uint64_t plen = 0;
    for (int i = 0; i < 8; i++) plen = (plen << 8) | buf[i];
    if (plen > 1000000000000ull) return fail(SMR_ERR_ARG, "wire: invalidly large frame");   // safetcp.rs:56-66
    if (len - 8 < plen) return 0;                                                // frame not complete yet
    Rd r{buf + 8, plen};
    const uint64_t outer = r.varint();
    if (outer == 2) { m->kind = SMR_WIRE_LEAVE; return (int64_t)(8 + plen); }    // PeerMessage::Leave
    if (outer != 0) { m->kind = SMR_WIRE_OTHER; return (int64_t)(8 + plen); }    // lease traffic: not this path
    const uint64_t v = r.varint();
    m->kind = (uint8_t)(v <= SMR_WIRE_COMMIT_NOTICE ? v : SMR_WIRE_OTHER);
    switch (v) {
        case SMR_WIRE_PREPARE: m->trigger_slot = r.varint(); m->ballot = r.varint(); break;
        case SMR_WIRE_PREPARE_REPLY: {
            m->slot = r.varint(); m->trigger_slot = r.varint(); m->endprep_slot = r.varint(); m->ballot = r.varint();
            m->has_voted = r.byte();
            if (m->has_voted > 1) r.ok = false;
            if (m->has_voted == 1) {
                m->voted_ballot = r.varint();
                m->reqs_off = 8 + r.n;
                if (skip_reqbatch(r)) m->reqs_len = 8 + r.n - m->reqs_off;
            }
            m->accept_bar = r.varint();
            break;
        }
        case SMR_WIRE_ACCEPT: {
            m->slot = r.varint(); m->ballot = r.varint();
            m->reqs_off = 8 + r.n;
            if (skip_reqbatch(r)) m->reqs_len = 8 + r.n - m->reqs_off;
            break;
        }
        case SMR_WIRE_ACCEPT_REPLY: {
            m->slot = r.varint(); m->ballot = r.varint();
            const uint8_t ts = r.byte();                                          // Option<SystemTime>
            if (ts == 1) { r.varint(); r.varint(); }                              // Duration { secs u64, nanos u32 } since the epoch
            else if (ts != 0) r.ok = false;
            break;
        }
        case SMR_WIRE_READ_QUERY: {
            m->reqs_off = 8 + r.n;
            if (skip_reqbatch(r)) m->reqs_len = 8 + r.n - m->reqs_off;
            break;
        }
        case SMR_WIRE_READ_QUERY_REPLY: {
            m->rq_client = r.varint(); m->rq_req_id = r.varint();
            m->replies_off = 8 + r.n;
            const uint64_t n = r.count();
            m->n_replies = n;
            for (uint64_t i = 0; i < n && r.ok; i++) {
                const uint8_t t = r.byte();
                if (t == 0) continue;
                if (t != 1) { r.ok = false; break; }
                r.varint();
                const uint8_t tv = r.byte();
                if (tv == 1) r.skip(r.varint());
                else if (tv != 0) r.ok = false;
            }
            m->replies_len = 8 + r.n - m->replies_off;
            const uint8_t fl = r.byte();
            if (fl > 1) r.ok = false;
            m->from_leader = fl;
            break;
        }
        case SMR_WIRE_HEARTBEAT:
            m->ballot = r.varint(); m->commit_bar = r.varint(); m->exec_bar = r.varint(); m->snap_bar = r.varint();
            break;
        case SMR_WIRE_COMMIT_NOTICE: m->ballot = r.varint(); m->commit_bar = r.varint(); break;
        default: return (int64_t)(8 + plen);                                      // a variant this build does not know
    }
    if (!r.ok || r.n != plen) return fail(SMR_ERR_ARG, "wire: malformed frame");
    return (int64_t)(8 + plen);
}

/* ---- Raft ------------------------------------------------------------------------------------ */
// entries: n LogEntry records; entry i = (entry_term[i], reqs bytes reqs[reqs_off[i] .. reqs_off[i + 1]),
// external[i]); log_offset is written as 0, as the leader does before sending (durability.rs:49-53
// clones the in-memory entry; followers reset it, messages.rs:150)
int64_t smr_wire_raft_append_entries(uint64_t term, uint64_t prev_slot, uint64_t prev_term, uint32_t n,
                                     const uint64_t *entry_term, const uint8_t *reqs, const uint64_t *reqs_off,
                                     const uint8_t *external, uint64_t leader_commit, uint64_t last_snap, uint8_t *out,
                                     uint64_t cap) {
    if (n && (!entry_term || !reqs || !reqs_off)) return fail(SMR_ERR_ARG, "wire: null entry arrays");
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(0);                                                    // PeerMessage::Msg, AppendEntries
    w.varint(term); w.varint(prev_slot); w.varint(prev_term);
    w.varint(n);
    for (uint32_t i = 0; i < n; i++) {
        w.varint(entry_term[i]);
        w.raw(reqs + reqs_off[i], reqs_off[i + 1] - reqs_off[i]);                // bincode(ReqBatch)
        w.byte(external && external[i] ? 1 : 0);
        w.varint(0);                                                             // log_offset
    }
    w.varint(leader_commit); w.varint(last_snap);
    return frame_done(w, out);
}

int64_t smr_wire_raft_append_entries_reply(uint64_t term, uint64_t end_slot, int has_conflict, uint64_t conflict_term,
                                           uint64_t conflict_slot, uint8_t *out, uint64_t cap) {
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(1); w.varint(term); w.varint(end_slot);
    w.byte(has_conflict ? 1 : 0);
    if (has_conflict) { w.varint(conflict_term); w.varint(conflict_slot); }
    return frame_done(w, out);
}

int64_t smr_wire_raft_request_vote(uint64_t term, uint64_t last_slot, uint64_t last_term, uint8_t *out, uint64_t cap) {
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(2); w.varint(term); w.varint(last_slot); w.varint(last_term);
    return frame_done(w, out);
}

int64_t smr_wire_raft_request_vote_reply(uint64_t term, int granted, uint8_t *out, uint64_t cap) {
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(3); w.varint(term); w.byte(granted ? 1 : 0);
    return frame_done(w, out);
}

/* DurEntry::Metadata { curr_term, voted_for } log record (voted_for 255 = None, mod.rs:146-151,158-163) */
int64_t smr_wal_raft_metadata(uint64_t curr_term, uint8_t voted_for, uint8_t *out, uint64_t cap) {
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(curr_term); w.byte(voted_for);                         // u8 is a raw byte
    return frame_done(w, out);
}

/* Parses the first Raft TCP frame.  For AppendEntries the entries are walked: entry_term_out[i] (up to
 * max_entries of them) and n_entries are filled, reqs are skipped. */
int64_t smr_wire_raft_decode(const uint8_t *buf, uint64_t len, smr_wire_raft_msg *m, uint64_t *entry_term_out,
                             uint32_t max_entries) {
    if (!buf || !m) return fail(SMR_ERR_ARG, "wire: null argument");
    memset(m, 0, sizeof(*m));
    if (len < 8) return 0;
    uint64_t plen = 0;
    for (int i = 0; i < 8; i++) plen = (plen << 8) | buf[i];
    if (plen > 1000000000000ull) return fail(SMR_ERR_ARG, "wire: invalidly large frame");
    if (len - 8 < plen) return 0;
    Rd r{buf + 8, plen};
    const uint64_t outer = r.varint();
    if (outer == 2) { m->kind = SMR_WIRE_LEAVE; return (int64_t)(8 + plen); }
    if (outer != 0) { m->kind = SMR_WIRE_OTHER; return (int64_t)(8 + plen); }
    const uint64_t v = r.varint();
    if (v > 3) return fail(SMR_ERR_ARG, "wire: unknown Raft message");
    m->kind = (uint8_t)v;
    switch (v) {
        case 0: {
            m->term = r.varint(); m->prev_slot = r.varint(); m->prev_term = r.varint();
            const uint64_t n = r.count();
            if (n > plen) { r.ok = false; break; }                               // every entry takes bytes
            m->n_entries = (uint32_t)n;
            for (uint64_t i = 0; i < n && r.ok; i++) {
                const uint64_t t = r.varint();
                if (entry_term_out && i < max_entries) entry_term_out[i] = t;
                skip_reqbatch(r);
                if (r.byte() > 1) r.ok = false;                                  // external
                r.varint();                                                      // log_offset
            }
            m->leader_commit = r.varint(); m->last_snap = r.varint();
            break;
        }
        case 1: {
            m->term = r.varint(); m->end_slot = r.varint();
            m->has_conflict = r.byte();
            if (m->has_conflict > 1) r.ok = false;
            if (m->has_conflict == 1) { m->conflict_term = r.varint(); m->conflict_slot = r.varint(); }
            break;
        }
        case 2: m->term = r.varint(); m->last_slot = r.varint(); m->last_term = r.varint(); break;
        case 3: m->term = r.varint(); m->granted = r.byte(); if (m->granted > 1) r.ok = false; break;
    }
    if (!r.ok || r.n != plen) return fail(SMR_ERR_ARG, "wire: malformed frame");
    return (int64_t)(8 + plen);
}

/* ---- RSPaxos -------------------------------------------------------------------------------- */
// RSCodeword<ReqBatch> (src/utils/rscoding.rs:43-66): num_data_shards u8, num_parity_shards u8, data_len, shard_len,
// shards as Vec<Option<Vec<u8>>>, data_copy: Option<T> (never sent along: subset_copy(.., false) drops it).
// `shards` = the (d + p) x shard_len bytes of a codeword, shard k at k * shard_stride; only those in avail_mask are read.
static void put_codeword(Wr &w, uint8_t d, uint8_t p, uint64_t data_len, uint64_t shard_len, uint32_t avail_mask,
                         const uint8_t *shards, uint64_t shard_stride) {
    w.byte(d); w.byte(p);                                                          // u8: a plain byte, not a varint
    w.varint(data_len); w.varint(shard_len);
    w.varint((uint64_t)d + p);
    for (uint32_t k = 0; k < (uint32_t)d + p; k++) {
        if ((avail_mask >> k) & 1u) { w.byte(1); w.bytes(shards + (uint64_t)k * shard_stride, shard_len); }
        else w.byte(0);
    }
    w.byte(0);                                                                     // data_copy: None
}
static bool get_codeword(Rd &r, uint64_t base, smr_wire_codeword *c) {
    memset(c, 0, sizeof(*c));
    c->num_data_shards = r.byte(); c->num_parity_shards = r.byte();
    c->data_len = r.varint(); c->shard_len = r.varint();
    const uint64_t n = r.count();
    if (!r.ok || n != (uint64_t)c->num_data_shards + c->num_parity_shards || n > 16) { r.ok = false; return false; }
    for (uint64_t k = 0; k < n && r.ok; k++) {
        const uint8_t some = r.byte();
        if (some > 1) { r.ok = false; break; }
        if (!some) continue;
        const uint64_t len = r.varint();
        if (len != c->shard_len) { r.ok = false; break; }
        c->avail_mask |= 1u << k;
        c->shard_off[k] = base + r.n;
        r.skip(len);
    }
    const uint8_t dc = r.byte();                                                   // data_copy
    if (dc == 1) skip_reqbatch(r); else if (dc != 0) r.ok = false;
    return r.ok;
}

int64_t smr_wire_rscodeword(uint8_t d, uint8_t p, uint64_t data_len, uint64_t shard_len, uint32_t avail_mask, const uint8_t *shards,
                            uint64_t shard_stride, uint8_t *out, uint64_t cap) {
    if (!out || (avail_mask && !shards) || (uint32_t)d + p > 16) return fail(SMR_ERR_ARG, "wire: bad argument");
    Wr w{out, cap};
    put_codeword(w, d, p, data_len, shard_len, avail_mask, shards, shard_stride);
    if (!w.ok) return fail(SMR_ERR_ARG, "wire: output buffer too small");
    return (int64_t)w.n;
}

// PeerMessage::Msg (variant 0) around rspaxos PeerMsg (mod.rs:247-311): Prepare 0, PrepareReply 1, Accept 2, AcceptReply 3,
// Reconstruct 4, ReconstructReply 5, Heartbeat 6.  `cw` = bincode(RSCodeword) bytes from smr_wire_rscodeword.
int64_t smr_wire_rsp_prepare(uint64_t trigger_slot, uint64_t ballot, uint8_t *out, uint64_t cap) {
    if (!out) return fail(SMR_ERR_ARG, "wire: null argument");
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(0); w.varint(trigger_slot); w.varint(ballot);
    return frame_done(w, out);
}
int64_t smr_wire_rsp_prepare_reply(uint64_t slot, uint64_t trigger_slot, uint64_t endprep_slot, uint64_t ballot, int has_voted,
                                   uint64_t voted_ballot, const uint8_t *cw, uint64_t cw_len, uint8_t *out, uint64_t cap) {
    if (!out || (has_voted && !cw)) return fail(SMR_ERR_ARG, "wire: null argument");
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(1); w.varint(slot); w.varint(trigger_slot); w.varint(endprep_slot); w.varint(ballot);
    w.byte(has_voted ? 1 : 0);
    if (has_voted) { w.varint(voted_ballot); w.raw(cw, cw_len); }
    return frame_done(w, out);
}
int64_t smr_wire_rsp_accept(uint64_t slot, uint64_t ballot, const uint8_t *cw, uint64_t cw_len, uint8_t *out, uint64_t cap) {
    if (!out || !cw) return fail(SMR_ERR_ARG, "wire: null argument");
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(2); w.varint(slot); w.varint(ballot); w.raw(cw, cw_len);
    return frame_done(w, out);
}
int64_t smr_wire_rsp_accept_reply(uint64_t slot, uint64_t ballot, uint8_t *out, uint64_t cap) {
    if (!out) return fail(SMR_ERR_ARG, "wire: null argument");
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(3); w.varint(slot); w.varint(ballot);
    return frame_done(w, out);
}
int64_t smr_wire_rsp_reconstruct(uint32_t n, const uint64_t *slots, uint8_t *out, uint64_t cap) {
    if (!out || (n && !slots)) return fail(SMR_ERR_ARG, "wire: null argument");
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(4); w.varint(n);
    for (uint32_t i = 0; i < n; i++) w.varint(slots[i]);
    return frame_done(w, out);
}
// slots_data: HashMap<usize, (Ballot, RSCodeword)> = varint count, then (key, (ballot, codeword)) in the map's iteration
// order -- here: the order given; codeword i = cws[cw_off[i] .. cw_off[i + 1])
int64_t smr_wire_rsp_reconstruct_reply(uint32_t n, const uint64_t *slots, const uint64_t *ballots, const uint8_t *cws,
                                       const uint64_t *cw_off, uint8_t *out, uint64_t cap) {
    if (!out || (n && (!slots || !ballots || !cws || !cw_off))) return fail(SMR_ERR_ARG, "wire: null argument");
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(5); w.varint(n);
    for (uint32_t i = 0; i < n; i++) { w.varint(slots[i]); w.varint(ballots[i]); w.raw(cws + cw_off[i], cw_off[i + 1] - cw_off[i]); }
    return frame_done(w, out);
}
int64_t smr_wire_rsp_heartbeat(uint64_t ballot, uint64_t commit_bar, uint64_t exec_bar, uint64_t snap_bar, uint8_t *out, uint64_t cap) {
    if (!out) return fail(SMR_ERR_ARG, "wire: null argument");
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(6); w.varint(ballot); w.varint(commit_bar); w.varint(exec_bar); w.varint(snap_bar);
    return frame_done(w, out);
}
// rspaxos WalEntry (mod.rs:207-222): PrepareBal 0, AcceptData 1 { slot, ballot, reqs_cw }, CommitSlot 2
int64_t smr_wal_rsp_accept_data(uint64_t slot, uint64_t ballot, const uint8_t *cw, uint64_t cw_len, uint8_t *out, uint64_t cap) {
    if (!out || !cw) return fail(SMR_ERR_ARG, "wire: null argument");
    Wr w = frame_begin(out, cap);
    w.varint(1); w.varint(slot); w.varint(ballot); w.raw(cw, cw_len);
    return frame_done(w, out);
}

int64_t smr_wire_rsp_decode(const uint8_t *buf, uint64_t len, smr_wire_rsp_msg *m, smr_wire_codeword *cws, uint64_t *slots,
                            uint64_t *ballots, uint32_t max_items) {
    if (!buf || !m) return fail(SMR_ERR_ARG, "wire: null argument");
    memset(m, 0, sizeof(*m));
    if (len < 8) return 0;
    uint64_t plen = 0;
    for (int i = 0; i < 8; i++) plen = (plen << 8) | buf[i];
    if (plen > 1000000000000ull) return fail(SMR_ERR_ARG, "wire: invalidly large frame");
    if (len - 8 < plen) return 0;
    Rd r{buf + 8, plen};
    const uint64_t outer = r.varint();
    if (outer == 2) { m->kind = SMR_WIRE_LEAVE; return (int64_t)(8 + plen); }
    if (outer != 0) { m->kind = SMR_WIRE_OTHER; return (int64_t)(8 + plen); }
    const uint64_t v = r.varint();
    m->kind = (uint8_t)(v <= SMR_WIRE_RSP_HEARTBEAT ? v : SMR_WIRE_OTHER);
    smr_wire_codeword scratch;
    switch (v) {
        case SMR_WIRE_PREPARE: m->trigger_slot = r.varint(); m->ballot = r.varint(); break;
        case SMR_WIRE_PREPARE_REPLY:
            m->slot = r.varint(); m->trigger_slot = r.varint(); m->endprep_slot = r.varint(); m->ballot = r.varint();
            m->has_voted = r.byte();
            if (m->has_voted > 1) r.ok = false;
            if (m->has_voted == 1) { m->voted_ballot = r.varint(); get_codeword(r, 8, (cws && max_items) ? &cws[0] : &scratch); m->n_items = 1; }
            break;
        case SMR_WIRE_ACCEPT:
            m->slot = r.varint(); m->ballot = r.varint();
            get_codeword(r, 8, (cws && max_items) ? &cws[0] : &scratch); m->n_items = 1;
            break;
        case SMR_WIRE_ACCEPT_REPLY: m->slot = r.varint(); m->ballot = r.varint(); break;
        case SMR_WIRE_RSP_RECONSTRUCT: {
            const uint64_t n = r.count();
            m->n_items = (uint32_t)n;
            for (uint64_t i = 0; i < n && r.ok; i++) { const uint64_t s = r.varint(); if (slots && i < max_items) slots[i] = s; }
            break;
        }
        case SMR_WIRE_RSP_RECONSTRUCT_REPLY: {
            const uint64_t n = r.count();
            m->n_items = (uint32_t)n;
            for (uint64_t i = 0; i < n && r.ok; i++) {
                const uint64_t s = r.varint(), b = r.varint();
                if (slots && i < max_items) slots[i] = s;
                if (ballots && i < max_items) ballots[i] = b;
                get_codeword(r, 8, (cws && i < max_items) ? &cws[i] : &scratch);
            }
            break;
        }
        case SMR_WIRE_RSP_HEARTBEAT: m->ballot = r.varint(); m->commit_bar = r.varint(); m->exec_bar = r.varint(); m->snap_bar = r.varint(); break;
        default: return (int64_t)(8 + plen);
    }
    if (!r.ok || r.n != plen) return fail(SMR_ERR_ARG, "wire: malformed frame");
    return (int64_t)(8 + plen);
}

/* ---- EPaxos --------------------------------------------------------------------------------- */
// epaxos/mod.rs: SlotIdx(ReplicaId u8, usize) (:199), DepSet(Vec<Option<usize>>) (:124), SeqNum u64; PeerMsg (:306-377):
// PreAccept 0 { slot, ballot, seq, deps, reqs }, PreAcceptReply 1 { slot, ballot, seq, deps }, Accept 2 (as PreAccept),
// AcceptReply 3 { slot, ballot }, CommitNotice 4 (as PreAccept); WalEntry (:254-281): PreAcceptSlot 0, AcceptSlot 1,
// CommitSlot 2, all { slot, ballot, seq, deps, reqs }.  deps[i] == SMR_EP_NONE: None.
static void put_ep_body(Wr &w, uint8_t row, uint64_t col, uint64_t ballot, bool with_seq_deps, uint64_t seq, const uint32_t *deps,
                        uint32_t n_deps, const uint8_t *reqs, uint64_t reqs_len) {
    w.byte(row); w.varint(col); w.varint(ballot);
    if (!with_seq_deps) return;
    w.varint(seq);
    w.varint(n_deps);
    for (uint32_t i = 0; i < n_deps; i++) {
        if (deps[i] == SMR_EP_NONE) w.byte(0);
        else { w.byte(1); w.varint(deps[i]); }
    }
    if (reqs) w.raw(reqs, reqs_len);
}

int64_t smr_wire_ep_msg(uint8_t kind, uint8_t row, uint64_t col, uint64_t ballot, uint64_t seq, const uint32_t *deps, uint32_t n_deps,
                        const uint8_t *reqs, uint64_t reqs_len, uint8_t *out, uint64_t cap) {
    if (!out || kind > SMR_WIRE_EP_COMMIT_NOTICE) return fail(SMR_ERR_ARG, "wire: bad argument");
    const bool body = kind != SMR_WIRE_EP_ACCEPT_REPLY, has_reqs = body && kind != SMR_WIRE_EP_PRE_ACCEPT_REPLY;
    if ((body && n_deps && !deps) || (has_reqs && !reqs)) return fail(SMR_ERR_ARG, "wire: null argument");
    Wr w = frame_begin(out, cap);
    w.varint(0); w.varint(kind);
    put_ep_body(w, row, col, ballot, body, seq, deps, n_deps, has_reqs ? reqs : nullptr, reqs_len);
    return frame_done(w, out);
}
int64_t smr_wal_ep_slot(uint8_t kind, uint8_t row, uint64_t col, uint64_t ballot, uint64_t seq, const uint32_t *deps, uint32_t n_deps,
                        const uint8_t *reqs, uint64_t reqs_len, uint8_t *out, uint64_t cap) {
    if (!out || kind > 2 || !reqs || (n_deps && !deps)) return fail(SMR_ERR_ARG, "wire: bad argument");
    Wr w = frame_begin(out, cap);
    w.varint(kind);
    put_ep_body(w, row, col, ballot, true, seq, deps, n_deps, reqs, reqs_len);
    return frame_done(w, out);
}
int64_t smr_wire_ep_decode(const uint8_t *buf, uint64_t len, smr_wire_ep_msg_t *m, uint32_t *deps_out, uint32_t max_deps) {
    if (!buf || !m) return fail(SMR_ERR_ARG, "wire: null argument");
    memset(m, 0, sizeof(*m));
    if (len < 8) return 0;
    uint64_t plen = 0;
    for (int i = 0; i < 8; i++) plen = (plen << 8) | buf[i];
    if (plen > 1000000000000ull) return fail(SMR_ERR_ARG, "wire: invalidly large frame");
    if (len - 8 < plen) return 0;
    Rd r{buf + 8, plen};
    const uint64_t outer = r.varint();
    if (outer == 2) { m->kind = SMR_WIRE_LEAVE; return (int64_t)(8 + plen); }
    if (outer != 0) { m->kind = SMR_WIRE_OTHER; return (int64_t)(8 + plen); }
    const uint64_t v = r.varint();
    if (v > SMR_WIRE_EP_COMMIT_NOTICE) { m->kind = SMR_WIRE_OTHER; return (int64_t)(8 + plen); }   // ExpPrepare & co, Heartbeat
    m->kind = (uint8_t)v;
    m->row = r.byte(); m->col = r.varint(); m->ballot = r.varint();
    if (v != SMR_WIRE_EP_ACCEPT_REPLY) {
        m->seq = r.varint();
        const uint64_t n = r.count();
        if (n > 64) r.ok = false;
        m->n_deps = (uint32_t)n;
        for (uint64_t i = 0; i < n && r.ok; i++) {
            const uint8_t some = r.byte();
            uint32_t d = SMR_EP_NONE;
            if (some == 1) d = (uint32_t)r.varint(); else if (some != 0) r.ok = false;
            if (deps_out && i < max_deps) deps_out[i] = d;
        }
        if (v != SMR_WIRE_EP_PRE_ACCEPT_REPLY) {
            m->reqs_off = 8 + r.n;
            if (skip_reqbatch(r)) m->reqs_len = 8 + r.n - m->reqs_off;
        }
    }
    if (!r.ok || r.n != plen) return fail(SMR_ERR_ARG, "wire: malformed frame");
    return (int64_t)(8 + plen);
}

}  // extern "C"

/* ---- request batching front-end ------------------------------------------------
 * ExternalApi (src/server/external.rs): client requests queue up in rx_req; every batch_interval the ticker
 * (:697-730) wakes get_req_batch (:323-344), which drains up to max_batch_size requests (0 = no limit) into one
 * ReqBatch, FIFO, and ignores ticks that find the queue empty.  Here: one queue per group, one call per tick. */
namespace smr { struct BtReq { uint64_t client, req_id; uint8_t kind; std::string key, value; }; }
struct smr_batcher {
    uint32_t max_batch_size;
    std::vector<std::deque<smr::BtReq>> q;
    uint64_t pending = 0;
};

extern "C" {

int smr_batcher_create(uint32_t n_groups, uint32_t max_batch_size, smr_batcher **out) {
    if (!out || n_groups == 0) return fail(SMR_ERR_ARG, "batcher: bad argument");
    smr_batcher *b = new smr_batcher();
    b->max_batch_size = max_batch_size;
    b->q.resize(n_groups);
    *out = b;
    return SMR_OK;
}
void smr_batcher_destroy(smr_batcher *b) { delete b; }

int smr_batcher_submit(smr_batcher *b, uint32_t group, uint64_t client, uint64_t req_id, uint8_t kind, const char *key, uint32_t key_len,
                       const char *value, uint32_t value_len) {
    if (!b || group >= b->q.size() || kind > SMR_CMD_PUT || (key_len && !key) || (kind == SMR_CMD_PUT && value_len && !value))
        return fail(SMR_ERR_ARG, "batcher: bad argument");
    b->q[group].push_back(smr::BtReq{client, req_id, kind, std::string(key ? key : "", key_len),
                                     std::string(kind == SMR_CMD_PUT && value ? value : "", kind == SMR_CMD_PUT ? value_len : 0)});
    b->pending++;
    return SMR_OK;
}

int smr_batcher_pending(smr_batcher *b, uint64_t *n) {
    if (!b || !n) return fail(SMR_ERR_ARG, "batcher: null argument");
    *n = b->pending;
    return SMR_OK;
}

int64_t smr_batcher_tick(smr_batcher *b, uint32_t *groups, uint32_t *counts, uint64_t *off, uint32_t max_groups, uint8_t *bytes, uint64_t cap) {
    if (!b || !groups || !counts || !off || !bytes) return fail(SMR_ERR_ARG, "batcher: null argument");
    // first pass: sizes (nothing is consumed unless everything fits)
    Wr w{bytes, cap};
    uint32_t n = 0;
    for (uint32_t g = 0; g < b->q.size(); g++) {
        const std::deque<smr::BtReq> &q = b->q[g];
        if (q.empty()) continue;                                                  // "ignore ticks with an empty batch"
        if (n == max_groups) return fail(SMR_ERR_ARG, "batcher: more groups with requests than max_groups");
        const uint32_t take = (b->max_batch_size == 0 || q.size() < b->max_batch_size) ? (uint32_t)q.size() : b->max_batch_size;
        groups[n] = g; counts[n] = take; off[n] = w.n;
        w.varint(take);
        for (uint32_t i = 0; i < take; i++) {
            const smr::BtReq &r = q[i];
            w.varint(r.client); w.varint(0);                                       // ApiRequest::Req { id, cmd }
            w.varint(r.req_id); w.varint(r.kind);
            w.bytes(r.key.data(), r.key.size());
            if (r.kind == SMR_CMD_PUT) w.bytes(r.value.data(), r.value.size());
        }
        n++;
    }
    off[n] = w.n;
    if (!w.ok) return fail(SMR_ERR_ARG, "batcher: output buffer too small");
    for (uint32_t k = 0; k < n; k++) {
        std::deque<smr::BtReq> &q = b->q[groups[k]];
        q.erase(q.begin(), q.begin() + counts[k]);
        b->pending -= counts[k];
    }
    return (int64_t)n;
}

// ---- the WAL backer file as a byte image (server/storage.rs:240-432) ---------------------------------------------------
// StorageHubLoggerTask's five file operations on a growable host buffer that stands for the backer file: entries are
// `[u64 BE length][bincode bytes]` frames (what the smr_wal_* encoders above produce); `file_size` is the logger's own idea
// of the log's end and arrives as an argument, exactly as in the reference's functions; the cursor rests at EOF between calls.
}  // extern "C"
struct smr_wallog { std::vector<uint8_t> f; };
extern "C" {

int smr_wallog_create(smr_wallog **out) {
    if (!out) return fail(SMR_ERR_ARG, "wallog: null argument");
    *out = new smr_wallog();
    return SMR_OK;
}
void smr_wallog_destroy(smr_wallog *l) { delete l; }

int64_t smr_wallog_len(const smr_wallog *l) { return l ? (int64_t)l->f.size() : fail(SMR_ERR_ARG, "wallog: null argument"); }

int64_t smr_wallog_bytes(const smr_wallog *l, uint8_t *out, uint64_t cap) {
    if (!l || (!out && cap)) return fail(SMR_ERR_ARG, "wallog: null argument");
    if (cap < l->f.size()) return fail(SMR_ERR_ARG, "wallog: output buffer too small");
    if (!l->f.empty()) memcpy(out, l->f.data(), l->f.size());
    return (int64_t)l->f.size();
}

static void wallog_put(smr_wallog *l, uint64_t at, const uint8_t *entry, uint64_t len) {
    if (l->f.size() < at + 8 + len) l->f.resize(at + 8 + len);
    for (int i = 0; i < 8; i++) l->f[at + i] = (uint8_t)(len >> (8 * (7 - i)));
    if (len) memcpy(l->f.data() + at + 8, entry, len);
}

// write_entry (storage.rs:282-322): LogAction::Write { entry, offset, sync } -> (offset_ok, now_size)
int smr_wallog_write(smr_wallog *l, uint64_t file_size, const uint8_t *entry, uint64_t entry_len, uint64_t offset, uint8_t *offset_ok,
                     uint64_t *now_size) {
    if (!l || (!entry && entry_len) || !offset_ok || !now_size) return fail(SMR_ERR_ARG, "wallog: null argument");
    if (offset > file_size) { *offset_ok = 0; *now_size = file_size; return SMR_OK; }   // :289-297 no holes in the log file
    if (offset > l->f.size()) return fail(SMR_ERR_ARG, "wallog: file_size beyond the image");
    wallog_put(l, offset, entry, entry_len);
    const uint64_t end = offset + 8 + entry_len;
    *offset_ok = 1; *now_size = end > file_size ? end : file_size;              // :314-320
    return SMR_OK;
}

// append_entry (:326-347): at the cursor, i.e. the image's end; returns file_size + 8 + length
int smr_wallog_append(smr_wallog *l, uint64_t file_size, const uint8_t *entry, uint64_t entry_len, uint64_t *now_size) {
    if (!l || (!entry && entry_len) || !now_size) return fail(SMR_ERR_ARG, "wallog: null argument");
    wallog_put(l, l->f.size(), entry, entry_len);
    *now_size = file_size + 8 + entry_len;
    return SMR_OK;
}

// read_entry (:240-278): *entry_len = -1 for None (then *end_offset = offset), else the entry's bincode bytes in out
int smr_wallog_read(const smr_wallog *l, uint64_t file_size, uint64_t offset, uint8_t *out, uint64_t cap, int64_t *entry_len,
                    uint64_t *end_offset) {
    if (!l || !entry_len || !end_offset || (!out && cap)) return fail(SMR_ERR_ARG, "wallog: null argument");
    *entry_len = -1; *end_offset = offset;
    if (offset + 8 > file_size) return SMR_OK;                                   // :245-256
    if (offset + 8 > l->f.size()) return fail(SMR_ERR_ARG, "wallog: file_size beyond the image");
    uint64_t len = 0;
    for (int i = 0; i < 8; i++) len = (len << 8) | l->f[offset + i];
    if (len > file_size || offset + 8 + len > file_size) return SMR_OK;          // :262-266 invalid length
    if (offset + 8 + len > l->f.size()) return fail(SMR_ERR_ARG, "wallog: file_size beyond the image");
    if (cap < len) return fail(SMR_ERR_ARG, "wallog: output buffer too small");
    if (len) memcpy(out, l->f.data() + offset + 8, len);
    *entry_len = (int64_t)len; *end_offset = offset + 8 + len;
    return SMR_OK;
}

// truncate_log (:351-371): keep the head
int smr_wallog_truncate(smr_wallog *l, uint64_t file_size, uint64_t offset, uint8_t *ok, uint64_t *now_size) {
    if (!l || !ok || !now_size) return fail(SMR_ERR_ARG, "wallog: null argument");
    if (offset > file_size) { *ok = 0; *now_size = file_size; return SMR_OK; }
    l->f.resize(offset);                                                         // set_len: shrinks or zero-extends
    *ok = 1; *now_size = offset;
    return SMR_OK;
}

// discard_log (:375-417): drop [keep, offset), keeping a fixed head of `keep` bytes and the tail from `offset`
int smr_wallog_discard(smr_wallog *l, uint64_t file_size, uint64_t offset, uint64_t keep, uint8_t *ok, uint64_t *now_size) {
    if (!l || !ok || !now_size) return fail(SMR_ERR_ARG, "wallog: null argument");
    if (offset > file_size || keep >= offset) { *ok = 0; *now_size = file_size; return SMR_OK; }
    if (file_size > l->f.size()) return fail(SMR_ERR_ARG, "wallog: file_size beyond the image");
    const uint64_t tail = file_size - offset;
    if (tail) memmove(l->f.data() + keep, l->f.data() + offset, tail);
    l->f.resize(keep + tail);
    *ok = 1; *now_size = keep + tail;
    return SMR_OK;
}

}  // extern "C"

"""Wire + WAL formats of the MultiPaxos hot-path messages (host only): thin ctypes mirror of
smr_wire_* / smr_wal_* (include/summerset_hip.h).  Frames are bytes objects:
8-byte big-endian length + bincode-standard payload (src/utils/safetcp.rs:46,127-132)."""
import ctypes as C

from . import _lib
from ._lib import WireCodeword, WireEpMsg, WireMsg, WireRaftMsg, WireRspMsg, check

PREPARE, PREPARE_REPLY, ACCEPT, ACCEPT_REPLY, READ_QUERY, READ_QUERY_REPLY, HEARTBEAT, COMMIT_NOTICE = range(8)
LEAVE, OTHER = 0xFE, 0xFF
GET, PUT = 0, 1


def _grow_or_raise(rc, cap):
    """a negative return is retried with a larger buffer ONLY when the library says the buffer was the problem"""
    msg = _lib.load().smr_last_error().decode("utf-8", "replace")
    if "output buffer too small" not in msg or cap > (1 << 30):
        raise _lib.SummersetError(int(rc), msg)


def _call(fn, *args, cap=64):
    L = _lib.load()
    while True:
        buf = C.create_string_buffer(cap)
        n = getattr(L, fn)(*args, buf, cap)
        if n >= 0:
            return buf.raw[:n]
        _grow_or_raise(n, cap)        # anything but "output buffer too small" is the caller's error: raise now
        cap *= 8


def reqbatch(reqs):
    """bincode(ReqBatch) of [(client, req_id, ("get", key) | ("put", key, value)), ...]"""
    n = len(reqs)
    b = lambda x: x.encode() if isinstance(x, str) else bytes(x)
    kinds = [PUT if r[2][0] == "put" else GET for r in reqs]
    keys = [b(r[2][1]) for r in reqs]
    vals = [b(r[2][2]) if k == PUT else b"" for r, k in zip(reqs, kinds)]
    args = ((C.c_uint64 * n)(*[r[0] for r in reqs]), (C.c_uint64 * n)(*[r[1] for r in reqs]), (C.c_uint8 * n)(*kinds),
            (C.c_char_p * n)(*keys), (C.c_uint32 * n)(*[len(k) for k in keys]), (C.c_char_p * n)(*vals),
            (C.c_uint32 * n)(*[len(v) for v in vals]))
    return _call("smr_wire_reqbatch", n, *args, cap=64 + sum(len(k) + len(v) + 32 for k, v in zip(keys, vals)))


def prepare(trigger_slot, ballot):
    return _call("smr_wire_prepare", trigger_slot, ballot)


def prepare_reply(slot, trigger_slot, endprep_slot, ballot, voted=None, accept_bar=0):
    vb, vr = voted if voted is not None else (0, b"")
    return _call("smr_wire_prepare_reply", slot, trigger_slot, endprep_slot, ballot, int(voted is not None), vb, vr, len(vr),
                 accept_bar, cap=96 + len(vr))


def accept(slot, ballot, reqs):
    return _call("smr_wire_accept", slot, ballot, reqs, len(reqs), cap=64 + len(reqs))


def accept_reply(slot, ballot):
    return _call("smr_wire_accept_reply", slot, ballot)


def read_query(reads):
    """PeerMsg::ReadQuery; reads = reqbatch([... Gets ...])"""
    return _call("smr_wire_read_query", reads, len(reads), cap=64 + len(reads))


def read_query_reply(rq_id, replies, from_leader=False):
    """replies: [None | (slot, None) | (slot, value)] as `Vec<Option<(usize, Option<String>)>>`"""
    n = len(replies)
    b = lambda x: x.encode() if isinstance(x, str) else bytes(x)
    state = [0 if r is None else (1 if r[1] is None else 2) for r in replies]
    vals = [b(r[1]) if st == 2 else b"" for r, st in zip(replies, state)]
    return _call("smr_wire_read_query_reply", rq_id[0], rq_id[1], n, (C.c_uint8 * n)(*state),
                 (C.c_uint64 * n)(*[0 if r is None else r[0] for r in replies]), (C.c_char_p * n)(*vals),
                 (C.c_uint32 * n)(*[len(v) for v in vals]), int(from_leader), cap=64 + sum(len(v) + 24 for v in vals))


def heartbeat(ballot, commit_bar, exec_bar, snap_bar):
    return _call("smr_wire_heartbeat", ballot, commit_bar, exec_bar, snap_bar)


def commit_notice(ballot, commit_bar):
    return _call("smr_wire_commit_notice", ballot, commit_bar)


def wal_prepare_bal(slot, ballot):
    return _call("smr_wal_prepare_bal", slot, ballot)


def wal_accept_data(slot, ballot, reqs):
    return _call("smr_wal_accept_data", slot, ballot, reqs, len(reqs), cap=64 + len(reqs))


def wal_commit_slot(slot):
    return _call("smr_wal_commit_slot", slot)


def decode(buf):
    """first frame of buf -> (bytes consumed, dict) ; (0, None) if the frame is not complete yet"""
    m = WireMsg()
    n = _lib.load().smr_wire_decode(bytes(buf), len(buf), C.byref(m))
    if n < 0:
        check(int(n))
    if n == 0:
        return 0, None
    d = {k: getattr(m, k) for k, _ in WireMsg._fields_}
    d["reqs"] = bytes(buf[m.reqs_off:m.reqs_off + m.reqs_len]) if m.reqs_len else b""
    if m.kind == READ_QUERY_REPLY:
        k = int(m.n_replies)
        part = bytes(buf[m.replies_off:m.replies_off + m.replies_len])
        st, sl, vo, vl = (C.c_uint8 * k)(), (C.c_uint64 * k)(), (C.c_uint64 * k)(), (C.c_uint64 * k)()
        got = _lib.load().smr_wire_read_query_replies(part, len(part), k, st, sl, vo, vl)
        if got < 0:
            check(int(got))
        d["replies"] = [None if st[i] == 0 else ((sl[i], None) if st[i] == 1 else (sl[i], part[vo[i]:vo[i] + vl[i]])) for i in range(k)]
    return int(n), d


# ---- Raft (src/protocols/raft/mod.rs:117-234) ----
RAFT_APPEND_ENTRIES, RAFT_APPEND_ENTRIES_REPLY, RAFT_REQUEST_VOTE, RAFT_REQUEST_VOTE_REPLY = 0, 1, 2, 3


def raft_append_entries(term, prev_slot, prev_term, entries, leader_commit, last_snap=0):
    """entries: [(term, reqs_bytes, external), ...]"""
    n = len(entries)
    terms = (C.c_uint64 * max(n, 1))(*[e[0] for e in entries])
    blob = b"".join(e[1] for e in entries)
    off, acc = [0], 0
    for e in entries:
        acc += len(e[1])
        off.append(acc)
    offs = (C.c_uint64 * (n + 1))(*off)
    ext = (C.c_uint8 * max(n, 1))(*[int(e[2]) for e in entries])
    return _call("smr_wire_raft_append_entries", term, prev_slot, prev_term, n, terms, blob, offs, ext, leader_commit,
                 last_snap, cap=128 + len(blob) + 24 * n)


def raft_append_entries_reply(term, end_slot, conflict=None):
    ct, cs = conflict if conflict is not None else (0, 0)
    return _call("smr_wire_raft_append_entries_reply", term, end_slot, int(conflict is not None), ct, cs)


def raft_request_vote(term, last_slot, last_term):
    return _call("smr_wire_raft_request_vote", term, last_slot, last_term)


def raft_request_vote_reply(term, granted):
    return _call("smr_wire_raft_request_vote_reply", term, int(granted))


def wal_raft_metadata(curr_term, voted_for=None):
    return _call("smr_wal_raft_metadata", curr_term, 255 if voted_for is None else voted_for)


def raft_decode(buf, max_entries=64):
    m = WireRaftMsg()
    terms = (C.c_uint64 * max_entries)()
    n = _lib.load().smr_wire_raft_decode(bytes(buf), len(buf), C.byref(m), terms, max_entries)
    if n < 0:
        check(int(n))
    if n == 0:
        return 0, None
    d = {k: getattr(m, k) for k, _ in WireRaftMsg._fields_}
    d["entry_terms"] = list(terms[:min(m.n_entries, max_entries)])
    return int(n), d


# ---- RSPaxos (src/protocols/rspaxos/mod.rs:207-311) and RSCodeword (src/utils/rscoding.rs:43-109) ------------------
RSP_RECONSTRUCT, RSP_RECONSTRUCT_REPLY, RSP_HEARTBEAT = 4, 5, 6


def rscodeword(d, p, data_len, shards):
    """bincode(RSCodeword): `shards` = list of d + p entries, bytes of equal length or None (shard absent)"""
    assert len(shards) == d + p
    have = [x for x in shards if x is not None]
    sl = len(have[0]) if have else 0
    assert all(len(x) == sl for x in have)
    mask = sum(1 << k for k, x in enumerate(shards) if x is not None)
    flat = b"".join(x if x is not None else bytes(sl) for x in shards)
    return _call("smr_wire_rscodeword", d, p, data_len, sl, mask, flat, sl, cap=64 + len(flat) + 8 * (d + p))


def rsp_prepare(trigger_slot, ballot):
    return _call("smr_wire_rsp_prepare", trigger_slot, ballot)


def rsp_prepare_reply(slot, trigger_slot, endprep_slot, ballot, voted=None):
    vb, cw = voted if voted is not None else (0, b"")
    return _call("smr_wire_rsp_prepare_reply", slot, trigger_slot, endprep_slot, ballot, int(voted is not None), vb, cw, len(cw),
                 cap=96 + len(cw))


def rsp_accept(slot, ballot, cw):
    return _call("smr_wire_rsp_accept", slot, ballot, cw, len(cw), cap=64 + len(cw))


def rsp_accept_reply(slot, ballot):
    return _call("smr_wire_rsp_accept_reply", slot, ballot)


def rsp_reconstruct(slots):
    n = len(slots)
    return _call("smr_wire_rsp_reconstruct", n, (C.c_uint64 * max(n, 1))(*slots), cap=64 + 9 * n)


def rsp_reconstruct_reply(entries):
    """entries: [(slot, ballot, codeword bytes), ...] in the order the map iterates"""
    n = len(entries)
    off = [0]
    for e in entries:
        off.append(off[-1] + len(e[2]))
    return _call("smr_wire_rsp_reconstruct_reply", n, (C.c_uint64 * max(n, 1))(*[e[0] for e in entries]),
                 (C.c_uint64 * max(n, 1))(*[e[1] for e in entries]), b"".join(e[2] for e in entries), (C.c_uint64 * (n + 1))(*off),
                 cap=64 + off[-1] + 18 * n)


def rsp_heartbeat(ballot, commit_bar, exec_bar, snap_bar):
    return _call("smr_wire_rsp_heartbeat", ballot, commit_bar, exec_bar, snap_bar)


def wal_rsp_accept_data(slot, ballot, cw):
    return _call("smr_wal_rsp_accept_data", slot, ballot, cw, len(cw), cap=64 + len(cw))


def rsp_decode(buf, max_items=16):
    """first RSPaxos frame of `buf`: (bytes consumed, dict) -- (0, None) while incomplete; codewords come back as dicts
    with the shard bytes sliced out of `buf`"""
    m = WireRspMsg()
    cws = (WireCodeword * max_items)()
    slots, ballots = (C.c_uint64 * max_items)(), (C.c_uint64 * max_items)()
    n = _lib.load().smr_wire_rsp_decode(bytes(buf), len(buf), C.byref(m), cws, slots, ballots, max_items)
    if n < 0:
        check(int(n))
    if n == 0:
        return 0, None
    out = {f: getattr(m, f) for f, _ in WireRspMsg._fields_}

    def cw(c):
        k = c.num_data_shards + c.num_parity_shards
        return dict(d=c.num_data_shards, p=c.num_parity_shards, data_len=c.data_len, shard_len=c.shard_len, avail=c.avail_mask,
                    shards=[bytes(buf[c.shard_off[i]:c.shard_off[i] + c.shard_len]) if (c.avail_mask >> i) & 1 else None for i in range(k)])
    k = min(m.n_items, max_items)
    if m.kind in (ACCEPT, PREPARE_REPLY) and m.n_items:
        out["codeword"] = cw(cws[0])
    if m.kind == RSP_RECONSTRUCT:
        out["slots"] = list(slots[:k])
    if m.kind == RSP_RECONSTRUCT_REPLY:
        out["entries"] = [(slots[i], ballots[i], cw(cws[i])) for i in range(k)]
    return int(n), out


# ---- EPaxos (src/protocols/epaxos/mod.rs:124,199,254-377) ------------------------------------------------------------
EP_PRE_ACCEPT, EP_PRE_ACCEPT_REPLY, EP_ACCEPT, EP_ACCEPT_REPLY, EP_COMMIT_NOTICE = 0, 1, 2, 3, 4
EP_NONE = 0xFFFFFFFF


def _deps(deps):
    d = [EP_NONE if x is None else x for x in (deps or [])]
    return (C.c_uint32 * max(len(d), 1))(*d), len(d)


def ep_msg(kind, row, col, ballot, seq=0, deps=None, reqs=None):
    """one EPaxos PeerMsg frame; deps: list with None for Option::None; reqs: bincode(ReqBatch) bytes"""
    d, n = _deps(deps)
    r = reqs if reqs is not None else b""
    return _call("smr_wire_ep_msg", kind, row, col, ballot, seq, d, n, r if reqs is not None else None, len(r), cap=96 + 10 * n + len(r))


def wal_ep_slot(kind, row, col, ballot, seq, deps, reqs):
    d, n = _deps(deps)
    return _call("smr_wal_ep_slot", kind, row, col, ballot, seq, d, n, reqs, len(reqs), cap=96 + 10 * n + len(reqs))


def ep_decode(buf, max_deps=8):
    m = WireEpMsg()
    deps = (C.c_uint32 * max_deps)()
    n = _lib.load().smr_wire_ep_decode(bytes(buf), len(buf), C.byref(m), deps, max_deps)
    if n < 0:
        check(int(n))
    if n == 0:
        return 0, None
    out = {f: getattr(m, f) for f, _ in WireEpMsg._fields_}
    out["deps"] = [None if deps[i] == EP_NONE else deps[i] for i in range(min(m.n_deps, max_deps))]
    out["reqs"] = bytes(buf[m.reqs_off:m.reqs_off + m.reqs_len])
    return int(n), out


# ---- request batching front-end (src/server/external.rs:323-344, 697-730) ---------------------------------------------------
class Batcher:
    """per-group request queues; `tick()` = one batch interval: {group: (n requests, bincode(ReqBatch) bytes)}"""

    def __init__(self, n_groups, max_batch_size=0):
        self.G = int(n_groups)
        self._L = _lib.load()
        h = C.c_void_p()
        check(self._L.smr_batcher_create(self.G, int(max_batch_size), C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.smr_batcher_destroy(self._h)
            self._h = None

    def submit(self, group, client, req_id, cmd):
        """cmd: ("get", key) | ("put", key, value)"""
        b = lambda x: x.encode() if isinstance(x, str) else bytes(x)
        key = b(cmd[1])
        val = b(cmd[2]) if cmd[0] == "put" else b""
        check(self._L.smr_batcher_submit(self._h, group, client, req_id, PUT if cmd[0] == "put" else GET, key, len(key), val, len(val)))

    def pending(self):
        n = C.c_uint64()
        check(self._L.smr_batcher_pending(self._h, C.byref(n)))
        return n.value

    def tick(self, cap=1 << 16):
        groups, counts, off = (C.c_uint32 * self.G)(), (C.c_uint32 * self.G)(), (C.c_uint64 * (self.G + 1))()
        while True:
            buf = C.create_string_buffer(cap)
            n = self._L.smr_batcher_tick(self._h, groups, counts, off, self.G, buf, cap)
            if n >= 0:
                return {groups[k]: (counts[k], buf.raw[off[k]:off[k + 1]]) for k in range(n)}
            _grow_or_raise(n, cap)
            cap *= 8


class WalLog:
    """the WAL backer file as a byte image: StorageHubLoggerTask's file operations (server/storage.rs:240-432) with the
    reference's signatures -- `file_size` is the caller's idea of the log's end; entries are bincode bytes (what the wal_*
    encoders return minus their 8-byte frame header, `frame_payload`)"""

    def __init__(self):
        self._L = _lib.load()
        h = C.c_void_p()
        check(self._L.smr_wallog_create(C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.smr_wallog_destroy(self._h)
            self._h = None

    def __len__(self):
        return int(self._L.smr_wallog_len(self._h))

    def bytes(self):
        n = len(self)
        buf = C.create_string_buffer(max(n, 1))
        got = self._L.smr_wallog_bytes(self._h, buf, n)
        if got < 0:
            check(int(got))
        return buf.raw[:got]

    def write_entry(self, file_size, entry, offset):
        ok, now = C.c_uint8(), C.c_uint64()
        check(self._L.smr_wallog_write(self._h, file_size, bytes(entry), len(entry), offset, C.byref(ok), C.byref(now)))
        return bool(ok.value), int(now.value)

    def append_entry(self, file_size, entry):
        now = C.c_uint64()
        check(self._L.smr_wallog_append(self._h, file_size, bytes(entry), len(entry), C.byref(now)))
        return int(now.value)

    def read_entry(self, file_size, offset):
        n, end = C.c_int64(), C.c_uint64()
        cap = max(len(self), 1)
        buf = C.create_string_buffer(cap)
        check(self._L.smr_wallog_read(self._h, file_size, offset, buf, cap, C.byref(n), C.byref(end)))
        return (None if n.value < 0 else buf.raw[:n.value]), int(end.value)

    def truncate_log(self, file_size, offset):
        ok, now = C.c_uint8(), C.c_uint64()
        check(self._L.smr_wallog_truncate(self._h, file_size, offset, C.byref(ok), C.byref(now)))
        return bool(ok.value), int(now.value)

    def discard_log(self, file_size, offset, keep):
        ok, now = C.c_uint8(), C.c_uint64()
        check(self._L.smr_wallog_discard(self._h, file_size, offset, keep, C.byref(ok), C.byref(now)))
        return bool(ok.value), int(now.value)


def frame_payload(frame):
    """the bincode bytes of a `[u64 BE length][bytes]` frame as the encoders above return it"""
    n = int.from_bytes(frame[:8], "big")
    assert len(frame) == 8 + n
    return frame[8:]


# ---- MultiPaxos peer traffic parsed on the device (smr_wire_ingest_mp, csrc/wire_ingest.hip) ----
import numpy as _np  # noqa: E402

HB_DTYPE = _np.dtype([("group", "<u4"), ("peer", "<u4"), ("kind", "<u4"), ("reserved", "<u4"), ("ballot", "<u8"),
                      ("commit_bar", "<u8"), ("exec_bar", "<u8"), ("snap_bar", "<u8")])
OTHER_DTYPE = _np.dtype([("conn", "<u4"), ("kind", "<u4"), ("off", "<u8"), ("len", "<u8")])


class MpIngest:
    """The receive side of a batch of peer connections on the device: the bytes every connection delivered this tick
    (device uint8 tensor `buf`, connection c = buf[conn_off[c]:conn_off[c + 1]], sent by replica conn_peer[c] of group
    conn_group[c]) -> AcceptReply records in `acks` (multipaxos.ACK_DTYPE, the tensor `deliver_acks` takes), Heartbeat /
    CommitNotice records in `hbs` (HB_DTYPE), every other frame located in `others` (OTHER_DTYPE) for `decode`; all in
    the sequential decoder's order.  Output tensors are allocated once for the capacities given and reused."""

    def __init__(self, n_conn, ack_cap, hb_cap, other_cap, device):
        import torch
        from .multipaxos import ACK_DTYPE
        L = _lib.load()
        self._L, self.n_conn, self.device = L, int(n_conn), device
        self.acks = torch.zeros(max(ack_cap, 1) * ACK_DTYPE.itemsize, dtype=torch.uint8, device=device)
        self.hbs = torch.zeros(max(hb_cap, 1) * HB_DTYPE.itemsize, dtype=torch.uint8, device=device)
        self.others = torch.zeros(max(other_cap, 1) * OTHER_DTYPE.itemsize, dtype=torch.uint8, device=device)
        self.caps = (int(ack_cap), int(hb_cap), int(other_cap))
        self.counts = torch.zeros(4, dtype=torch.int64, device=device)
        self.consumed = torch.zeros(max(n_conn, 1), dtype=torch.int64, device=device)
        self.status = torch.zeros(max(n_conn, 1), dtype=torch.int32, device=device)
        self.scratch = torch.zeros(max(int(L.smr_wire_ingest_scratch_bytes(self.n_conn)), 8) // 8, dtype=torch.int64, device=device)

    def ingest(self, buf, conn_off, conn_group, conn_peer, stream=None):
        """buf uint8 (16-byte aligned: a torch allocation is), conn_off int64 [n_conn + 1], conn_group int32 / uint32
        [n_conn], conn_peer uint8 [n_conn], all on the device; only enqueues work"""
        assert conn_off.numel() == self.n_conn + 1 and conn_group.numel() == self.n_conn and conn_peer.numel() == self.n_conn
        assert conn_off.element_size() == 8 and conn_group.element_size() == 4 and conn_peer.element_size() == 1
        p = lambda t: t.data_ptr()   # noqa: E731
        check(self._L.smr_wire_ingest_mp(p(buf) if buf.numel() else None, buf.numel(), p(conn_off), p(conn_group), p(conn_peer), self.n_conn,
                                         p(self.acks), self.caps[0], p(self.hbs), self.caps[1], p(self.others), self.caps[2],
                                         p(self.counts), p(self.consumed), p(self.status), p(self.scratch), _lib.stream_ptr(stream)))

    def results(self):
        """host copies (synchronises): dict of counts, the record arrays cut to what was stored, consumed, status"""
        from .multipaxos import ACK_DTYPE
        n = [int(x) for x in self.counts.cpu().tolist()]
        cut = lambda t, dt, k, cap: t.cpu().numpy().view(dt)[:min(k, cap)].copy()   # noqa: E731
        return {"n_acks": n[0], "n_hbs": n[1], "n_others": n[2], "n_malformed": n[3],
                "acks": cut(self.acks, ACK_DTYPE, n[0], self.caps[0]), "hbs": cut(self.hbs, HB_DTYPE, n[1], self.caps[1]),
                "others": cut(self.others, OTHER_DTYPE, n[2], self.caps[2]),
                "consumed": self.consumed.cpu().numpy()[:self.n_conn].copy(), "status": self.status.cpu().numpy()[:self.n_conn].copy()}




ACK12_DTYPE = _np.dtype([("slot", "<u4"), ("ballot_lo", "<u4"), ("ballot_hi", "<u4")])        # smr_wire_ack12


class MpIngestConn:
    """`MpIngest` in ONE pass (smr_wire_ingest_mp_conn): every connection has segments of its own -- connection c's AcceptReplies
    as 12-byte (slot, ballot) records (ACK12_DTYPE: the group and the peer are the connection's) in `acks` from record
    conn_off[c] // 13 on (what `MultiPaxosCluster.deliver_acks_conn` takes as it is), its first `hb_per_conn` Heartbeats /
    CommitNotices in `hbs[c]`, its first `other_per_conn` located frames in `others[c]`, and `cnt[c] = (acks, hbs, others)`.
    One record more than a segment holds stops the connection in front of that frame (status 2; `consumed[c]` says where the
    next call goes on).  `max_buf_len` sizes the ack array (max_buf_len // 13 + 1 records)."""

    def __init__(self, n_conn, max_buf_len, hb_per_conn, other_per_conn, device):
        import torch
        self._L, self.n_conn, self.device = _lib.load(), int(n_conn), device
        self.ack_cap = int(max_buf_len) // 13 + 1
        self.hb_per_conn, self.other_per_conn = int(hb_per_conn), int(other_per_conn)
        self.acks = torch.zeros(self.ack_cap * ACK12_DTYPE.itemsize, dtype=torch.uint8, device=device)
        self.hbs = torch.zeros(max(self.n_conn * self.hb_per_conn, 1) * HB_DTYPE.itemsize, dtype=torch.uint8, device=device)
        self.others = torch.zeros(max(self.n_conn * self.other_per_conn, 1) * OTHER_DTYPE.itemsize, dtype=torch.uint8, device=device)
        self.cnt = torch.zeros((max(self.n_conn, 1), 3), dtype=torch.int32, device=device)
        self.consumed = torch.zeros(max(n_conn, 1), dtype=torch.int64, device=device)
        self.status = torch.zeros(max(n_conn, 1), dtype=torch.int32, device=device)
        self.conn = None                                         # (conn_off, conn_group, conn_peer) of the last call

    def ingest(self, buf, conn_off, conn_group, conn_peer, stream=None):
        assert conn_off.numel() == self.n_conn + 1 and conn_group.numel() == self.n_conn and conn_peer.numel() == self.n_conn
        assert conn_off.element_size() == 8 and conn_group.element_size() == 4 and conn_peer.element_size() == 1
        p = lambda t: t.data_ptr()   # noqa: E731
        check(self._L.smr_wire_ingest_mp_conn(p(buf) if buf.numel() else None, buf.numel(), p(conn_off), p(conn_group), p(conn_peer), self.n_conn,
                                              p(self.acks), self.ack_cap, p(self.hbs), self.hb_per_conn, p(self.others), self.other_per_conn,
                                              p(self.cnt), p(self.consumed), p(self.status), _lib.stream_ptr(stream)))
        self.conn = (conn_off, conn_group, conn_peer)

    def results(self):
        """host copies (synchronises) of the last call: the segments gathered connection by connection and widened to
        multipaxos.ACK_DTYPE -- the lists `MpIngest.results` gives where no connection was stopped -- plus cnt, consumed, status"""
        from .multipaxos import ACK_DTYPE
        n = self.n_conn
        off = self.conn[0].cpu().numpy().astype(_np.int64) if n else _np.zeros(1, _np.int64)
        grp = self.conn[1].cpu().numpy().view(_np.uint32) if n else _np.zeros(0, _np.uint32)
        peer = self.conn[2].cpu().numpy() if n else _np.zeros(0, _np.uint8)
        cnt = self.cnt.cpu().numpy()[:n].astype(_np.int64)
        a12, hbs, others = (self.acks.cpu().numpy().view(ACK12_DTYPE), self.hbs.cpu().numpy().view(HB_DTYPE), self.others.cpu().numpy().view(OTHER_DTYPE))
        idx = lambda first, k: _np.concatenate([_np.arange(f, f + m) for f, m in zip(first, k)] or [_np.zeros(0, _np.int64)]).astype(_np.int64)   # noqa: E731
        c = _np.arange(n, dtype=_np.int64)
        seg = a12[idx(off[:-1] // 13, cnt[:, 0])]
        acks = _np.zeros(len(seg), ACK_DTYPE)
        acks["slot"] = seg["slot"]
        acks["ballot"] = seg["ballot_lo"].astype(_np.uint64) | (seg["ballot_hi"].astype(_np.uint64) << _np.uint64(32))
        acks["group"] = _np.repeat(grp, cnt[:, 0]); acks["peer"] = _np.repeat(peer, cnt[:, 0])
        return {"cnt": cnt, "acks": acks, "hbs": hbs[idx(c * self.hb_per_conn, cnt[:, 1])], "others": others[idx(c * self.other_per_conn, cnt[:, 2])],
                "consumed": self.consumed.cpu().numpy()[:n].copy(), "status": self.status.cpu().numpy()[:n].copy()}


# ---- Raft / EPaxos reply traffic parsed on the device (csrc/wire_ingest_replies.hip) ----
class ReplyIngest:
    """One reply per (peer, group) and call out of the bytes the leader's connections delivered, straight into the arrays
    the engines take: `raft(...)` -> dict(reply_term, end_slot, conflict_term, conflict_slot, flags) [R, G] for
    `RaftLeaderGroup.handle_msg_append_entries_reply` / `run_ticks`; `ep_pre_accept(...)` -> dict(ballot, seq, deps
    [R, R, G], flags) for `EPaxosReplicaGroup.handle_pre_accept_replies`.  Tensors are allocated once and reused; other
    frames are located in `others` (OTHER_DTYPE, unordered)."""

    def __init__(self, n_conn, n_groups, population, other_cap, device):
        import torch
        self._L, self.n_conn, self.G, self.R, self.device = _lib.load(), int(n_conn), int(n_groups), int(population), device
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)   # noqa: E731
        R, G = self.R, self.G
        self.u64a, self.u64b = z((R, G), torch.int64), z((R, G), torch.int64)
        self.u32a, self.deps = z((R, G), torch.int32), z((R, R, G), torch.int32)
        self.u32b = z((R, G), torch.int32)
        self.flags = z((R, G), torch.uint8)
        self.others = z(max(other_cap, 1) * OTHER_DTYPE.itemsize, torch.uint8)
        self.other_cap = int(other_cap)
        self.counts = z(4, torch.int64)
        self.consumed = z(max(self.n_conn, 1), torch.int64)
        self.status = z(max(self.n_conn, 1), torch.int32)

    def _conn(self, buf, conn_off, conn_group, conn_peer, conn_len=None):
        """conn_len (uint8 [n_conn], optional): connection c = buf[conn_off[c] : conn_off[c] + conn_len[c]] (the emit calls'
        layout: frames.view(-1), conn_off = arange * stride, conn_len = len) instead of back-to-back streams"""
        assert conn_off.numel() >= self.n_conn + (conn_len is None) and conn_group.numel() == self.n_conn and conn_peer.numel() == self.n_conn
        assert conn_off.element_size() == 8 and conn_group.element_size() == 4 and conn_peer.element_size() == 1
        assert conn_len is None or (conn_len.numel() == self.n_conn and conn_len.element_size() == 1)
        p = lambda t: t.data_ptr()   # noqa: E731
        return (p(buf) if buf.numel() else None, buf.numel(), p(conn_off), p(conn_group), p(conn_peer), None if conn_len is None else p(conn_len),
                self.n_conn, self.G, self.R)

    def raft(self, buf, conn_off, conn_group, conn_peer, stream=None, conn_len=None):
        p = lambda t: t.data_ptr()   # noqa: E731
        check(self._L.smr_wire_ingest_raft_replies(*self._conn(buf, conn_off, conn_group, conn_peer, conn_len), p(self.u64a), p(self.u32a), p(self.u64b),
                                                   p(self.u32b), p(self.flags), p(self.others), self.other_cap, p(self.counts),
                                                   p(self.consumed), p(self.status), _lib.stream_ptr(stream)))
        return dict(reply_term=self.u64a, end_slot=self.u32a, conflict_term=self.u64b, conflict_slot=self.u32b, flags=self.flags)

    def raft_into(self, leader, buf, conn_off, order=None, stream=None, conn_len=None):
        """`raft(...)` + `leader.handle_msg_append_entries_reply(...)` as ONE launch (`smr_raft_leader_handle_wire_replies`): the
        parse is the prologue of the leader's reply handler, no [R, G] arrays in between.  The connections come dense --
        n_groups * (population - 1) of them, connection g * (population - 1) + k = group g's k-th follower in ascending id, the
        leader's own left out.  `results()` as after `raft(...)`."""
        p = lambda t: None if t is None else t.data_ptr()   # noqa: E731
        assert self.n_conn == self.G * (self.R - 1) and conn_off.numel() >= self.n_conn + (conn_len is None) and conn_off.element_size() == 8
        assert conn_len is None or (conn_len.numel() == self.n_conn and conn_len.element_size() == 1)
        assert order is None or (order.numel() == self.G and order.element_size() == 4)
        check(self._L.smr_raft_leader_handle_wire_replies(leader._h, p(buf) if buf.numel() else None, buf.numel(), p(conn_off), p(conn_len), self.n_conn,
                                                          p(order), p(self.others), self.other_cap, p(self.counts), p(self.consumed),
                                                          p(self.status), _lib.stream_ptr(stream)))

    def ep_pre_accept(self, buf, conn_off, conn_group, conn_peer, me, col, stream=None, conn_len=None):
        """col: int32 / uint32 [G] on the device -- the column of MY instance every group's replies are for"""
        p = lambda t: t.data_ptr()   # noqa: E731
        assert col.numel() == self.G and col.element_size() == 4
        check(self._L.smr_wire_ingest_ep_pre_accept_replies(*self._conn(buf, conn_off, conn_group, conn_peer, conn_len), int(me), p(col), p(self.u64a),
                                                            p(self.u64b), p(self.deps), p(self.flags), p(self.others), self.other_cap,
                                                            p(self.counts), p(self.consumed), p(self.status), _lib.stream_ptr(stream)))
        return dict(ballot=self.u64a, seq=self.u64b, deps=self.deps, flags=self.flags)

    def rsp_accept(self, buf, conn_off, conn_group, conn_peer, stream=None, conn_len=None):
        """RSPaxos AcceptReplies -> dict(slot, ballot, flags) [R, G] for `RSPaxosReplicaGroup.accept_replies`"""
        p = lambda t: t.data_ptr()   # noqa: E731
        check(self._L.smr_wire_ingest_rsp_accept_replies(*self._conn(buf, conn_off, conn_group, conn_peer, conn_len), p(self.u32a), p(self.u64a), p(self.flags),
                                                         p(self.others), self.other_cap, p(self.counts), p(self.consumed), p(self.status),
                                                         _lib.stream_ptr(stream)))
        return dict(slot=self.u32a, ballot=self.u64a, flags=self.flags)

    def results(self):
        """host copies (synchronises): counts, located frames, consumed, status"""
        n = [int(x) for x in self.counts.cpu().tolist()]
        return {"n_replies": n[0], "n_others": n[1], "n_malformed": n[2], "n_deferred": n[3],
                "others": self.others.cpu().numpy().view(OTHER_DTYPE)[:min(n[1], self.other_cap)].copy(),
                "consumed": self.consumed.cpu().numpy()[:self.n_conn].copy(), "status": self.status.cpu().numpy()[:self.n_conn].copy()}


# ---- reply frames written on the device (csrc/wire_emit.hip) ----
EMIT_MP_STRIDE, EMIT_RAFT_STRIDE, EMIT_EP_STRIDE = 32, 48, 96


def _emit_out(n, stride, device):
    import torch
    return (torch.zeros((max(n, 1), stride), dtype=torch.uint8, device=device), torch.zeros(max(n, 1), dtype=torch.uint8, device=device))


def emit_mp_accept_replies(acks, n, stream=None):
    """acks: device uint8 tensor of `n` smr_mp_ack records (what `MultiPaxosCluster.collect_acks` fills) -> (frames uint8 [n, 32],
    len uint8 [n]): record i as the AcceptReply frame the follower would send"""
    frames, ln = _emit_out(n, EMIT_MP_STRIDE, acks.device)
    check(_lib.load().smr_wire_emit_mp_accept_replies(acks.data_ptr(), int(n), frames.data_ptr(), ln.data_ptr(), _lib.stream_ptr(stream)))
    return frames, ln


def emit_raft_replies(flags, term, end_slot, conflict_term, conflict_slot, stream=None):
    """the [G] reply tensors of `RaftLeaderGroup.handle_msg_append_entries` -> (frames uint8 [G, 48], len uint8 [G])"""
    G = int(flags.numel())
    frames, ln = _emit_out(G, EMIT_RAFT_STRIDE, flags.device)
    check(_lib.load().smr_wire_emit_raft_replies(flags.data_ptr(), term.data_ptr(), end_slot.data_ptr(), conflict_term.data_ptr(),
                                                 conflict_slot.data_ptr(), G, frames.data_ptr(), ln.data_ptr(), _lib.stream_ptr(stream)))
    return frames, ln


def emit_ep_pre_accept_replies(flags, row, col, ballot, seq, deps, stream=None):
    """the [G] reply tensors of `EPaxosReplicaGroup.handle_msg_pre_accept` (deps [R, G]) for the instances (row, col[g]) ->
    (frames uint8 [G, 96], len uint8 [G])"""
    G, R = int(flags.numel()), int(deps.shape[0])
    frames, ln = _emit_out(G, EMIT_EP_STRIDE, flags.device)
    check(_lib.load().smr_wire_emit_ep_pre_accept_replies(flags.data_ptr(), int(row), col.data_ptr(), ballot.data_ptr(), seq.data_ptr(),
                                                          deps.data_ptr(), G, R, frames.data_ptr(), ln.data_ptr(), _lib.stream_ptr(stream)))
    return frames, ln

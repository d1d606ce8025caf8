"""Wire / WAL frames of the MultiPaxos hot-path messages (host only): byte layouts worked out from
the reference's type definitions (multipaxos/mod.rs:261-368, transport.rs:37-52, external.rs:33-54,
statemach.rs:21-27) under the bincode-standard rules of SURVEY.md Appendix C, round trips, and the
incomplete / malformed cases of safetcp.rs:30-70."""
import numpy as np
import pytest

from summerset_amd import wire
from summerset_amd._lib import SummersetError


def test_reqbatch_bytes_match_appendix_c_and_the_oracle(oracle):
    val = b"v" * 4096
    rb = wire.reqbatch([(7, 42, ("put", b"k0000003", val))])
    # 01 | client | 00 (Req) | id | 01 (Put) | 08 key | FB 00 10 value
    assert rb[:5] == bytes([1, 7, 0, 42, 1]) and rb[5] == 8 and rb[6:14] == b"k0000003"
    assert rb[14:17] == bytes([0xFB, 0x00, 0x10]) and rb[17:] == val and len(rb) == 4110 + 1 + 1 + 1
    assert rb == bytes(oracle.bincode_reqbatch_put(7, 42, b"k0000003", val))        # independent restatement
    g = wire.reqbatch([(300, 70000, ("get", "a"))])
    assert g == bytes([1, 0xFB, 0x2C, 0x01, 0, 0xFC, 0x70, 0x11, 0x01, 0x00, 0, 1]) + b"a"


def test_frames_byte_for_byte():
    assert wire.prepare(5, 0x202) == bytes([0, 0, 0, 0, 0, 0, 0, 6, 0, 0, 5, 0xFB, 0x02, 0x02])
    assert wire.accept_reply(9, 0x101) == bytes([0, 0, 0, 0, 0, 0, 0, 7, 0, 3, 9, 0xFB, 0x01, 0x01, 0])
    rb = wire.reqbatch([(1, 2, ("put", "k", "v"))])
    a = wire.accept(300, 0x101, rb)
    assert a[:8] == (len(a) - 8).to_bytes(8, "big") and a[8:10] == bytes([0, 2]) and a[10:13] == bytes([0xFB, 0x2C, 0x01])
    assert a.endswith(rb)
    assert wire.wal_commit_slot(1 << 40) == bytes([0, 0, 0, 0, 0, 0, 0, 10, 2, 0xFD]) + (1 << 40).to_bytes(8, "little")
    assert wire.wal_prepare_bal(3, 4) == bytes([0, 0, 0, 0, 0, 0, 0, 3, 0, 3, 4])
    w = wire.wal_accept_data(3, 0x101, rb)
    assert w[8:13] == bytes([1, 3, 0xFB, 0x01, 0x01]) and w[13:] == rb


def test_round_trips_and_stream_reassembly():
    rb = wire.reqbatch([(1, 2, ("put", "key", "value")), (1, 3, ("get", "key"))])
    frames = [wire.prepare(17, 0x303), wire.prepare_reply(18, 17, 40, 0x303, voted=(0x101, rb), accept_bar=16),
              wire.prepare_reply(19, 17, 40, 0x303), wire.accept(1000, 0x303, rb), wire.accept_reply(1000, 0x303)]
    stream = b"".join(frames)
    out = []
    while stream:
        n, m = wire.decode(stream)
        assert n > 0
        out.append(m)
        stream = stream[n:]
    assert [m["kind"] for m in out] == [wire.PREPARE, wire.PREPARE_REPLY, wire.PREPARE_REPLY, wire.ACCEPT, wire.ACCEPT_REPLY]
    assert (out[0]["trigger_slot"], out[0]["ballot"]) == (17, 0x303)
    assert (out[1]["slot"], out[1]["endprep_slot"], out[1]["has_voted"], out[1]["voted_ballot"]) == (18, 40, 1, 0x101)
    assert out[1]["accept_bar"] == 16 and out[1]["reqs"] == rb
    assert out[2]["has_voted"] == 0 and out[2]["reqs"] == b""
    assert (out[3]["slot"], out[3]["ballot"]) == (1000, 0x303) and out[3]["reqs"] == rb
    assert (out[4]["slot"], out[4]["ballot"]) == (1000, 0x303)
    # a frame cut anywhere is "not complete yet" (safetcp.rs reads on), never an error
    f = frames[3]
    for cut in (0, 3, 8, 9, len(f) - 1):
        assert wire.decode(f[:cut]) == (0, None)


def test_quorum_read_heartbeat_and_commit_notice_frames():
    """PeerMsg variants 4..7 (multipaxos/mod.rs:344-378), bytes worked out by hand from the bincode standard config"""
    assert wire.heartbeat(0x101, 7, 5, 300) == bytes([0, 0, 0, 0, 0, 0, 0, 10, 0, 6, 0xFB, 0x01, 0x01, 7, 5, 0xFB, 0x2C, 0x01])
    assert wire.commit_notice(0x101, 9) == bytes([0, 0, 0, 0, 0, 0, 0, 6, 0, 7, 0xFB, 0x01, 0x01, 9])
    reads = wire.reqbatch([(4, 11, ("get", "a")), (5, 12, ("get", "bc"))])
    f = wire.read_query(reads)
    assert f[:8] == (2 + len(reads)).to_bytes(8, "big") and f[8:10] == bytes([0, 4]) and f[10:] == reads
    # rq_id (4, 11); replies [None, Some((3, None)), Some((260, Some("xy")))]; from_leader false
    g = wire.read_query_reply((4, 11), [None, (3, None), (260, "xy")])
    assert g == bytes([0, 0, 0, 0, 0, 0, 0, 18, 0, 5, 4, 11, 3, 0, 1, 3, 0, 1, 0xFB, 0x04, 0x01, 1, 2]) + b"xy" + bytes([0])
    stream = f + g + wire.read_query_reply((9, 1), [(0, "v")], from_leader=True) + wire.heartbeat(0x202, 1, 2, 3) + wire.commit_notice(0x202, 8)
    out = []
    while stream:
        n, m = wire.decode(stream)
        assert n > 0
        out.append(m)
        stream = stream[n:]
    assert [m["kind"] for m in out] == [wire.READ_QUERY, wire.READ_QUERY_REPLY, wire.READ_QUERY_REPLY, wire.HEARTBEAT, wire.COMMIT_NOTICE]
    assert out[0]["reqs"] == reads
    assert (out[1]["rq_client"], out[1]["rq_req_id"], out[1]["from_leader"]) == (4, 11, 0)
    assert out[1]["replies"] == [None, (3, None), (260, b"xy")]
    assert out[2]["from_leader"] == 1 and out[2]["replies"] == [(0, b"v")]
    assert (out[3]["ballot"], out[3]["commit_bar"], out[3]["exec_bar"], out[3]["snap_bar"]) == (0x202, 1, 2, 3)
    assert (out[4]["ballot"], out[4]["commit_bar"]) == (0x202, 8)
    for cut in (0, 5, 8, len(g) - 1):
        assert wire.decode(g[:cut]) == (0, None)
    bad = bytearray(g); bad[-1] = 2                         # from_leader is a bool
    with pytest.raises(SummersetError):
        wire.decode(bytes(bad))
    bad = bytearray(g); bad[13] = 3                         # Option tag of the first reply
    with pytest.raises(SummersetError):
        wire.decode(bytes(bad))


def test_malformed_frames_are_errors():
    f = bytearray(wire.accept_reply(9, 0x101))
    f[-1] = 7                                               # Option tag neither 0 nor 1
    with pytest.raises(SummersetError):
        wire.decode(bytes(f))
    g = wire.accept(1, 2, bytes([1, 1, 0, 5, 9]))     # ReqBatch with an unknown Command variant
    with pytest.raises(SummersetError):
        wire.decode(g)
    with pytest.raises(SummersetError):
        wire.decode(bytes([0x7F]) * 8 + b"x")               # absurd length (safetcp.rs:56-66)
    n, m = wire.decode(bytes([0, 0, 0, 0, 0, 0, 0, 1, 2]))  # PeerMessage::Leave
    assert n == 9 and m["kind"] == wire.LEAVE


def test_raft_frames(oracle):
    rb = wire.reqbatch([(1, 2, ("put", "k", "v"))])
    f = wire.raft_request_vote(7, 300, 6)
    assert f == bytes([0, 0, 0, 0, 0, 0, 0, 7, 0, 2, 7, 0xFB, 0x2C, 0x01, 6])
    assert wire.raft_request_vote_reply(7, True) == bytes([0, 0, 0, 0, 0, 0, 0, 4, 0, 3, 7, 1])
    assert wire.raft_append_entries_reply(3, 9) == bytes([0, 0, 0, 0, 0, 0, 0, 5, 0, 1, 3, 9, 0])
    assert wire.raft_append_entries_reply(3, 9, conflict=(2, 4)) == bytes([0, 0, 0, 0, 0, 0, 0, 7, 0, 1, 3, 9, 1, 2, 4])
    assert wire.wal_raft_metadata(5) == bytes([0, 0, 0, 0, 0, 0, 0, 3, 0, 5, 255])   # voted_for None = ReplicaId::MAX
    assert wire.wal_raft_metadata(5, 2)[-1] == 2
    ae = wire.raft_append_entries(4, 10, 3, [(4, rb, True), (4, b"\x00", False)], leader_commit=9, last_snap=1)
    # 0 0 | term prev_slot prev_term | n=2 | (term reqs external log_offset) x 2 | leader_commit last_snap
    assert ae[8:14] == bytes([0, 0, 4, 10, 3, 2]) and ae[14] == 4 and ae[15:15 + len(rb)] == rb
    assert ae[15 + len(rb):] == bytes([1, 0, 4, 0, 0, 0, 9, 1])
    hb = wire.raft_append_entries(4, 10, 3, [], leader_commit=9)
    stream = ae + hb + f + wire.raft_append_entries_reply(3, 9, conflict=(2, 4))
    got = []
    while stream:
        n, m = wire.raft_decode(stream)
        assert n > 0
        got.append(m)
        stream = stream[n:]
    assert [m["kind"] for m in got] == [0, 0, 2, 1]
    assert (got[0]["term"], got[0]["prev_slot"], got[0]["prev_term"], got[0]["n_entries"], got[0]["entry_terms"],
            got[0]["leader_commit"], got[0]["last_snap"]) == (4, 10, 3, 2, [4, 4], 9, 1)
    assert got[1]["n_entries"] == 0 and got[1]["leader_commit"] == 9
    assert (got[2]["term"], got[2]["last_slot"], got[2]["last_term"]) == (7, 300, 6)
    assert (got[3]["has_conflict"], got[3]["conflict_term"], got[3]["conflict_slot"], got[3]["end_slot"]) == (1, 2, 4, 9)
    assert wire.raft_decode(ae[:-1]) == (0, None)
    with pytest.raises(SummersetError):
        wire.raft_decode(bytes([0, 0, 0, 0, 0, 0, 0, 2, 0, 9]))     # unknown PeerMsg variant


# ---- RSPaxos frames and RSCodeword's own encoding ----------------------------------------------------------------------
def test_rscodeword_bytes_by_hand(engine_lib):
    from summerset_amd import wire
    # rscoding.rs:43-66: d u8, p u8, data_len, shard_len, Vec<Option<Vec<u8>>> (len; 0 | 1 + len + bytes), data_copy None
    cw = wire.rscodeword(3, 2, 4, [b"ab", None, b"cd", None, b"\xfe\xff"])
    assert cw == bytes([3, 2, 4, 2, 5, 1, 2]) + b"ab" + bytes([0, 1, 2]) + b"cd" + bytes([0, 1, 2, 0xFE, 0xFF, 0])
    # a null codeword (from_null): lengths 0, five absent shards
    assert wire.rscodeword(3, 2, 0, [None] * 5) == bytes([3, 2, 0, 0, 5, 0, 0, 0, 0, 0, 0])
    # lengths are varints: 251..65535 -> 0xFB + u16 LE (the two u8 fields stay single raw bytes)
    big = wire.rscodeword(1, 1, 300, [bytes(300), None])
    assert big[:9] == bytes([1, 1, 0xFB, 0x2C, 0x01, 0xFB, 0x2C, 0x01, 2]) and big[9:13] == bytes([1, 0xFB, 0x2C, 0x01]) and len(big) == 13 + 300 + 2


def test_rspaxos_frames_by_hand_and_round_trip(engine_lib):
    from summerset_amd import wire
    cw = wire.rscodeword(3, 2, 4, [None, b"xy", None, None, None])
    f = wire.rsp_accept(7, 0x102, cw)                            # PeerMessage::Msg (0) { PeerMsg::Accept (2) { slot, ballot, reqs_cw } }
    assert f == (len(f) - 8).to_bytes(8, "big") + bytes([0, 2, 7, 0xFB, 0x02, 0x01]) + cw
    n, m = wire.rsp_decode(f)
    assert n == len(f) and (m["kind"], m["slot"], m["ballot"]) == (wire.ACCEPT, 7, 0x102)
    assert m["codeword"] == dict(d=3, p=2, data_len=4, shard_len=2, avail=0b00010, shards=[None, b"xy", None, None, None])
    # PrepareReply with and without a vote
    f = wire.rsp_prepare_reply(5, 3, 9, 0x202, voted=(0x101, cw))
    assert f[8:] == bytes([0, 1, 5, 3, 9, 0xFB, 0x02, 0x02, 1, 0xFB, 0x01, 0x01]) + cw
    n, m = wire.rsp_decode(f)
    assert (m["has_voted"], m["voted_ballot"], m["trigger_slot"], m["endprep_slot"], m["codeword"]["avail"]) == (1, 0x101, 3, 9, 2)
    f = wire.rsp_prepare_reply(5, 3, 9, 0x202)
    assert f[8:] == bytes([0, 1, 5, 3, 9, 0xFB, 0x02, 0x02, 0]) and wire.rsp_decode(f)[1]["has_voted"] == 0
    # Reconstruct { slots: Vec<usize> } and its reply (HashMap<usize, (Ballot, RSCodeword)>)
    f = wire.rsp_reconstruct([4, 300])
    assert f[8:] == bytes([0, 4, 2, 4, 0xFB, 0x2C, 0x01]) and wire.rsp_decode(f)[1]["slots"] == [4, 300]
    f = wire.rsp_reconstruct_reply([(4, 0x101, cw), (6, 0x101, wire.rscodeword(3, 2, 4, [b"pq", None, None, None, None]))])
    n, m = wire.rsp_decode(f)
    assert n == len(f) and [(s, b, c["avail"]) for s, b, c in m["entries"]] == [(4, 0x101, 2), (6, 0x101, 1)]
    assert m["entries"][1][2]["shards"][0] == b"pq"
    # Heartbeat (6), AcceptReply (3), Prepare (0)
    f = wire.rsp_heartbeat(0x101, 12, 10, 0)
    assert f[8:] == bytes([0, 6, 0xFB, 0x01, 0x01, 12, 10, 0])
    assert {k: wire.rsp_decode(f)[1][k] for k in ("kind", "ballot", "commit_bar", "exec_bar", "snap_bar")} == \
        dict(kind=6, ballot=0x101, commit_bar=12, exec_bar=10, snap_bar=0)
    assert wire.rsp_accept_reply(7, 1)[8:] == bytes([0, 3, 7, 1]) and wire.rsp_prepare(2, 1)[8:] == bytes([0, 0, 2, 1])
    # WalEntry::AcceptData (1) { slot, ballot, reqs_cw }: no PeerMessage wrapper in the log
    assert wire.wal_rsp_accept_data(7, 1, cw)[8:] == bytes([1, 7, 1]) + cw
    # incomplete / malformed
    f = wire.rsp_accept(7, 1, cw)
    assert wire.rsp_decode(f[:-1]) == (0, None)
    bad = bytearray(f); bad[8 + 4 + 4] = 9                       # the shard count no longer matches d + p
    with pytest.raises(Exception):
        wire.rsp_decode(bytes(bad))


def test_rspaxos_frames_carry_the_rs_kernels_shards(engine_lib, oracle):
    """the bytes an Accept carries are a shard of the codeword the RS path produces (here: the oracle's encoder)"""
    from summerset_amd import wire
    data = bytes(range(1, 11))                                   # L = 10 -> shard_len 4, zero padded
    par = oracle.rs_encode(3, 2, np.frombuffer(data, np.uint8))
    padded = data + bytes(2)
    shards = [padded[0:4], padded[4:8], padded[8:12], par[0].tobytes(), par[1].tobytes()]
    for peer in range(5):
        only = [s if k == peer else None for k, s in enumerate(shards)]
        n, m = wire.rsp_decode(wire.rsp_accept(1, 0x101, wire.rscodeword(3, 2, len(data), only)))
        assert m["codeword"]["avail"] == 1 << peer and m["codeword"]["shards"][peer] == shards[peer] and m["codeword"]["data_len"] == 10


# ---- EPaxos frames -------------------------------------------------------------------------------------------------------
def test_epaxos_frames_by_hand_and_round_trip(engine_lib):
    from summerset_amd import wire
    reqs = wire.reqbatch([(7, 3, ("put", "k1", "v"))])
    # PeerMessage::Msg (0) { PeerMsg::PreAccept (0) { slot: SlotIdx(2, 300), ballot 3, seq 5, deps [None, Some(4), Some(299), None, None], reqs } }
    f = wire.ep_msg(wire.EP_PRE_ACCEPT, 2, 300, 3, seq=5, deps=[None, 4, 299, None, None], reqs=reqs)
    body = bytes([0, 0, 2, 0xFB, 0x2C, 0x01, 3, 5, 5, 0, 1, 4, 1, 0xFB, 0x2B, 0x01, 0, 0]) + reqs
    assert f == len(body).to_bytes(8, "big") + body
    n, m = wire.ep_decode(f)
    assert n == len(f) and (m["kind"], m["row"], m["col"], m["ballot"], m["seq"]) == (0, 2, 300, 3, 5)
    assert m["deps"] == [None, 4, 299, None, None] and m["reqs"] == reqs
    # PreAcceptReply (1): no reqs; AcceptReply (3): slot and ballot only
    f = wire.ep_msg(wire.EP_PRE_ACCEPT_REPLY, 1, 9, 2, seq=6, deps=[0, None, None])
    assert f[8:] == bytes([0, 1, 1, 9, 2, 6, 3, 1, 0, 0, 0]) and wire.ep_decode(f)[1]["deps"] == [0, None, None]
    f = wire.ep_msg(wire.EP_ACCEPT_REPLY, 4, 1, 5)
    assert f[8:] == bytes([0, 3, 4, 1, 5]) and wire.ep_decode(f)[1]["kind"] == 3
    for kind in (wire.EP_ACCEPT, wire.EP_COMMIT_NOTICE):
        f = wire.ep_msg(kind, 0, 1, 1, seq=2, deps=[None] * 5, reqs=reqs)
        assert f[8:10] == bytes([0, kind]) and wire.ep_decode(f)[1]["reqs"] == reqs
    # WalEntry::CommitSlot (2): no PeerMessage wrapper
    f = wire.wal_ep_slot(2, 3, 7, 4, 9, [None, None, 1], reqs)
    assert f[8:] == bytes([2, 3, 7, 4, 9, 3, 0, 0, 1, 1]) + reqs
    # other variants (ExpPrepare 5 ...) are skipped, incomplete frames wait, malformed ones raise
    assert wire.ep_decode((2).to_bytes(8, "big") + bytes([0, 5]))[1]["kind"] == wire.OTHER
    f = wire.ep_msg(wire.EP_ACCEPT_REPLY, 4, 1, 5)
    assert wire.ep_decode(f[:-2]) == (0, None)
    with pytest.raises(Exception):
        wire.ep_decode((6).to_bytes(8, "big") + bytes([0, 3, 4, 1, 5, 9]))      # a trailing byte


# ---- request batching front-end ---------------------------------------------------------------------------------------------
def test_batcher_follows_get_req_batch(engine_lib):
    """external.rs:323-344: a tick drains up to max_batch_size queued requests (0 = all) into one ReqBatch, FIFO;
    a tick that finds a queue empty produces nothing for it"""
    from summerset_amd import wire
    b = wire.Batcher(4, max_batch_size=2)
    assert b.tick() == {} and b.pending() == 0
    reqs = [(7, 1, ("put", "k", "v1")), (8, 1, ("get", "k")), (7, 2, ("put", "k", "v2"))]
    for r in reqs:
        b.submit(2, *r)
    b.submit(0, 9, 5, ("get", "x"))
    out = b.tick()
    assert set(out) == {0, 2} and out[2] == (2, wire.reqbatch(reqs[:2])) and out[0] == (1, wire.reqbatch([(9, 5, ("get", "x"))]))
    assert b.pending() == 1
    assert b.tick() == {2: (1, wire.reqbatch(reqs[2:]))} and b.tick() == {}
    every = wire.Batcher(1)                                     # max_batch_size 0: no limit
    many = [(c, c * 3, ("put", "key%d" % c, "x" * c)) for c in range(300)]
    for r in many:
        every.submit(0, *r)
    assert every.tick(cap=64) == {0: (300, wire.reqbatch(many))}   # the first buffer is too small: nothing consumed, retried
    with pytest.raises(SummersetError):
        b.submit(4, 1, 1, ("get", "k"))                          # no such group


# ---- hostile frames: lengths and counts off the wire are a peer's to choose (Rd::skip overflow) ----

def _frame(payload):
    return len(payload).to_bytes(8, "big") + bytes(payload)


_U64 = lambda v: bytes([0xFD]) + int(v).to_bytes(8, "little")


def _decoders():
    return [wire.decode, wire.raft_decode, wire.rsp_decode, wire.ep_decode]


def test_huge_varint_lengths_are_malformed_not_a_hang():
    """the 35-byte Accept of the advisor's report: ReqBatch count 2^64-1, key_len 2^64-13 -- `n + k` wrapped, the
    cursor moved backwards and the element was re-parsed for ever"""
    reqs = _U64(2 ** 64 - 1) + bytes([1, 0, 2, 1]) + _U64(2 ** 64 - 13)
    acc = _frame(bytes([0, wire.ACCEPT, 5]) + bytes([0xFB, 0x01, 0x01]) + reqs)
    with pytest.raises(SummersetError):
        wire.decode(acc)
    # the same batch with an honest count but a key length just past the end of the frame
    for klen in (2 ** 64 - 1, 2 ** 63, 2 ** 32, 40):
        reqs = bytes([1, 1, 0, 2, 1]) + _U64(klen) + b"k" * 8
        with pytest.raises(SummersetError):
            wire.decode(_frame(bytes([0, wire.ACCEPT, 5, 7]) + reqs))


def test_element_counts_are_bounded_by_the_bytes_left():
    # a ReqBatch that claims 2^40 requests inside a 20-byte frame
    with pytest.raises(SummersetError):
        wire.decode(_frame(bytes([0, wire.ACCEPT, 5, 7]) + _U64(2 ** 40) + bytes(6)))
    # ReadQueryReply: a reply count far beyond the payload
    good = wire.read_query_reply((3, 4), [(5, b"abc"), None])
    n, m = wire.decode(good)
    assert n == len(good) and m["kind"] == wire.READ_QUERY_REPLY


def test_random_garbage_never_hangs_or_reads_outside_the_frame():
    """every decoder on mutated valid frames and on random payloads: returns, raises SummersetError, or says
    "incomplete" -- and every (offset, length) it hands back lies inside the buffer"""
    rng = np.random.default_rng(0xF022)
    rb = wire.reqbatch([(1, 2, ("put", "key", "value")), (1, 3, ("get", "key"))])
    seeds = [wire.accept(9, 0x101, rb), wire.prepare_reply(18, 17, 40, 0x303, voted=(0x101, rb), accept_bar=16),
             wire.read_query_reply((3, 4), [(5, b"abc"), None]), wire.raft_append_entries(2, 4, 1, [(2, rb, 1), (2, rb, 0)], 3),
             wire.rsp_accept(4, 0x101, wire.rscodeword(3, 2, 10, [b"abcd", None, b"efgh", None, None])),
             wire.rsp_reconstruct([1, 2, 3]), wire.ep_msg(0, 1, 2, 3, seq=4, deps=[1, None, 3, None, 5], reqs=rb)]
    hostile = [0xFD, 0xFC, 0xFB, 0xFF, 0xFE, 0xFA, 0x00, 0x01]
    tried = 0
    for seed in seeds:
        for _ in range(300):
            b = bytearray(seed)
            for _k in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(8, len(b)))
                b[pos] = hostile[int(rng.integers(len(hostile)))] if rng.random() < 0.6 else int(rng.integers(256))
                if rng.random() < 0.3:                          # a maximal u64 behind a 0xFD tag
                    b[pos:pos + 9] = _U64(2 ** 64 - int(rng.integers(1, 20)))[: max(0, len(b) - pos)]
            b = bytes(b[:len(seed)])
            for dec in _decoders():
                tried += 1
                try:
                    n, m = dec(b)
                except SummersetError:
                    continue
                assert 0 <= n <= len(b)
                if m:
                    for k, v in m.items():
                        if isinstance(v, (bytes, bytearray)):
                            assert len(v) <= len(b)
    for _ in range(500):
        payload = rng.integers(0, 256, int(rng.integers(1, 48)), dtype=np.uint8).tobytes()
        for dec in _decoders():
            tried += 1
            try:
                n, m = dec(_frame(payload))
                assert 0 <= n <= len(payload) + 8
            except SummersetError:
                pass
    assert tried > 10000


def test_argument_errors_do_not_grow_buffers():
    """a negative return that is not "output buffer too small" raises at once (_call retried ten times
    and ended on an 8 GiB allocation)"""
    import time
    t0 = time.time()
    with pytest.raises(SummersetError):
        wire.read_query(b"")
    assert time.time() - t0 < 1.0
    # and a small buffer still grows as before
    big = wire.reqbatch([(1, 2, ("put", "k" * 300, "v" * 5000))])
    assert len(big) > 5300

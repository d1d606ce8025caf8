"""Edges of the device-side peer-traffic parser (csrc/wire_ingest.hip) that the random streams of
tests/test_zz_wire_ingest_gpu.py only meet by chance, each laid out on purpose and compared with the sequential decoder
`smr_wire_decode` frame by frame: payload lengths either side of the straight-line path (2 .. 24 bytes, varints below
2^32; round 2's two-register path ended at 16), frames that
end exactly at, one byte before and one byte behind the end of the lane's 128-byte window and its 16-byte refill
alignment, every varint width at its extreme values (u64::MAX ballots, slots beyond u32 that go the host's way),
non-canonical varints, varints that run off the frame, streams that end inside a header / inside a payload, a byte buffer
whose length is not a multiple of 16, connections of very different lengths in one wavefront.
Sorted last: written when no device was at hand (verified on the emulator, tests/test_hostsim.py) -- a failure here must
not keep the rest of the suite from running under `pytest -x`."""
import struct

import numpy as np
import pytest

from test_zz_wire_ingest_gpu import _expected, _frame, _ingest, _varint

pytestmark = pytest.mark.gpu          # first device run: GPUTEST_r02 (11 XPASS); quarantine removed in round 3
U64 = (1 << 64) - 1


def _fd(v):
    """a value in the widest encoding whatever its size (the host decoder takes any width: so must the device)"""
    return b"\xfd" + struct.pack("<Q", v)


def _fc(v):
    return b"\xfc" + struct.pack("<I", v)


def _ar(slot, ballot, ts=b"\x00"):
    return _frame(_varint(0) + _varint(3) + slot + ballot + ts)


def _check(wire, cuda, streams, seed=0):
    from summerset_amd.multipaxos import ACK_DTYPE
    rng = np.random.default_rng(seed)
    n = len(streams)
    groups, peers = rng.integers(0, 1 << 20, n), rng.integers(0, 5, n)
    acks, hbs, others, consumed, status = _expected(wire, ACK_DTYPE, streams, groups, peers)
    got = _ingest(wire, cuda, streams, groups, peers)
    assert (got["n_acks"], got["n_hbs"], got["n_others"], got["n_malformed"]) == (len(acks), len(hbs), len(others), int(status.sum()))
    assert np.array_equal(got["consumed"], consumed) and np.array_equal(got["status"], status)
    assert np.array_equal(got["acks"], acks) and np.array_equal(got["hbs"], hbs) and np.array_equal(got["others"], others)
    return len(acks), len(hbs), len(others), int(status.sum())


def test_payload_lengths_around_the_register_fast_path(cuda):
    """AcceptReplies and Heartbeats whose payload is 3 .. 40 bytes long through every mix of varint widths (the parser decodes
    payloads of <= 16 bytes out of two registers, longer ones through the window reader); the same frames with one byte too
    many / one too few declared"""
    from summerset_amd import wire
    widths = [lambda v: _varint(v % 251), lambda v: b"\xfb" + struct.pack("<H", v & 0xFFFF), lambda v: _fc(v & 0xFFFFFFFF), _fd]
    streams, lens = [], set()
    for a in range(4):
        for b in range(4):
            for ts in (b"\x00", b"\x01" + _varint(1790000000) + _varint(999999999), b"\x01" + _fd(5) + _fc(7)):
                body = _varint(0) + _varint(3) + widths[a](77 + a) + widths[b](0x101 + b) + ts
                lens.add(len(body))
                s = _frame(body) + wire.accept_reply(1, 2)
                streams.append(s)
                streams.append(struct.pack(">Q", len(body) + 1) + body + wire.accept_reply(1, 2))      # declared one byte longer: malformed
                streams.append(struct.pack(">Q", len(body) - 1) + body + wire.accept_reply(1, 2))      # one shorter: malformed
    for a in range(4):
        for b in range(4):
            body = _varint(0) + _varint(wire.HEARTBEAT) + widths[a](3) + widths[b](40000) + widths[(a + b) % 4](9) + widths[(a * b) % 4](1)
            lens.add(len(body))
            streams.append(_frame(body) + wire.commit_notice(5, 6))
    assert {15, 16, 17} <= lens and min(lens) <= 6 and max(lens) >= 30
    na, nh, no, nm = _check(wire, cuda, streams)
    assert nm == 2 * 48 and na >= 48 and nh >= 16


def test_every_frame_of_the_straight_line_path(cuda):
    """the parser's branch-free path takes payloads of 2 .. 24 bytes whose varints are 1, 3 or 5 bytes long: Heartbeats through
    all 81 mixes of those widths (6 .. 22 bytes), AcceptReplies and CommitNotices through all 9, each also with one byte too
    many / too few declared (malformed), with a 64-bit field in every position (the general reader's), and the short frames
    the device only locates (Prepare, a short PrepareReply, Leave, an unknown variant, lease traffic) between them; a
    timestamped AcceptReply short enough for the path (it must leave it)"""
    from summerset_amd import wire
    W = [lambda v: _varint(v % 251), lambda v: b"\xfb" + struct.pack("<H", 251 + v % 60000), lambda v: _fc(70000 + v)]
    tail = wire.accept_reply(1, 2)
    streams, n_bad, lens = [], 0, set()
    k = 0
    for a in range(3):
        for b in range(3):
            bodies = [_varint(0) + _varint(3) + W[a](k) + W[b](k + 1) + b"\x00",
                      _varint(0) + _varint(wire.COMMIT_NOTICE) + W[a](k + 2) + W[b](k + 3),
                      _varint(0) + _varint(3) + W[a](k) + W[b](k + 1) + b"\x01" + _varint(5) + _varint(7),          # Some(SystemTime), tiny
                      _varint(0) + _varint(3) + W[a](k) + W[b](k + 1) + b"\x02"]                                     # no such Option tag
            for c in range(3):
                for d in range(3):
                    bodies.append(_varint(0) + _varint(wire.HEARTBEAT) + W[a](k) + W[b](k + 1) + W[c](k + 2) + W[d](k + 3))
                    k += 5
            for i, body in enumerate(bodies):
                lens.add(len(body))
                streams.append(tail + _frame(body) + tail)
                n_bad += i == 3
                for delta in (1, -1):
                    streams.append(tail + struct.pack(">Q", len(body) + delta) + body + tail)
                    n_bad += 1
    # a 64-bit varint in every field of a frame that is otherwise the path's
    for pos in range(4):
        f = [_varint(9), _fc(70000), _varint(3), _varint(4)]
        f[pos] = _fd(1 << 40)
        streams.append(_frame(_varint(0) + _varint(wire.HEARTBEAT) + b"".join(f)) + tail)
    for pos in range(2):
        f = [_varint(9), _varint(3)]
        f[pos] = _fd((1 << 32) - 1)
        streams.append(_frame(_varint(0) + _varint(3) + b"".join(f) + b"\x00") + tail)
    # short frames that are only located, hot frames right behind them
    located = [wire.prepare(7, 0x101), wire.prepare(70000, 1 << 33), wire.prepare_reply(3, 3, 4, 0x101), _frame(_varint(2)),
               _frame(_varint(1) + bytes(range(20))), _frame(_varint(250) + b"\xff" * 22), _frame(_varint(0) + _varint(200) + b"\xfe" * 10)]
    for f in located:
        assert wire.decode(f)[0] == len(f)
        streams.append(tail + f + wire.heartbeat(0x101, 70000, 69999, 300) + f + tail)
    assert {6, 22} <= lens and max(len(frame) - 8 for frame in located) >= 20
    na, nh, no, nm = _check(wire, cuda, streams, seed=5)
    assert nm == n_bad and nh >= 81 * 9 // 9 and no >= 2 * len(located)


def test_frames_at_the_window_edges(cuda):
    """a lane walks its stream through a 128-byte window refilled at a 16-byte-aligned position: a filler frame of every
    length 0 .. 150 in front of hot frames puts their headers and payloads at every offset of the window, across its end and
    across refills; the stream of each connection starts at an arbitrary offset of the byte buffer"""
    from summerset_amd import wire
    streams = []
    for pad in range(0, 151):
        filler = _frame(_varint(1) + bytes((pad * 7 + i) & 0xFF for i in range(pad)))                 # lease traffic: located, not parsed
        s = filler + wire.accept_reply(pad, 0x101) + wire.heartbeat(0x201, pad, 2, 1) + _ar(_fd(pad), _fd(U64)) + wire.commit_notice(9, pad)
        s += wire.heartbeat(70000 + pad, 80000, 90000, 100000)                                        # 22 bytes: the longest straight-line frame
        streams.append(s + wire.accept_reply(65535, 70000) * 9)
    na, nh, no, nm = _check(wire, cuda, streams, seed=1)
    assert nm == 0 and na == 151 * 11 and nh == 151 * 3 and no == 151


def test_frames_around_the_ring_of_two_lines(cuda):
    """written for round 5's ring experiment (a lane's stream through a ring of two aligned 128-byte lines, the next line in
    flight while the ring is parsed: tools/experiments/wire_ingest_ring_r5.patch, measured slower, profiles/r9g) and kept for
    whatever walks the stream: a located filler of every length 0 .. 420 (shorter than a line, a line, two, more: the position
    jumps over bytes that need not be loaded) in front of hot frames, and behind them enough AcceptReplies for several more
    refills, so that every window / ring offset sees a header, a payload's end and a record kept in place of read bytes; two
    long fillers back to back; streams that end inside a frame"""
    from summerset_amd import wire
    streams = []
    for pad in range(0, 421):
        filler = _frame(_varint(1) + bytes((pad * 11 + i) & 0xFF for i in range(pad)))
        s = filler + wire.accept_reply(pad, 0x101) + wire.heartbeat(0x201, pad, 2, 1) + _ar(_fd(pad), _fd(U64))
        if pad % 3 == 0:
            s += _frame(_varint(1) + bytes(200 + pad % 97)) + _frame(_varint(1) + bytes(127 + pad % 5))
        s += wire.accept_reply(65535 + pad, 70000) * (20 + pad % 23)
        if pad % 7 == 0:
            s += wire.heartbeat(1, 2, 3, 4)[:pad % 13]                                              # ends inside a frame
        streams.append(s)
    na, nh, no, nm = _check(wire, cuda, streams, seed=6)
    assert nm == 0 and nh == 421 and no == 421 + 2 * 141 and na == sum(2 + 20 + pad % 23 for pad in range(421))


def test_extreme_values_and_odd_encodings(cuda):
    from summerset_amd import wire
    ok = [
        _ar(_fd(U64), _fd(U64)),                                  # a slot the engine cannot name: the host's (others)
        _ar(_fd((1 << 32)), _varint(1)),                          # first slot beyond u32
        _ar(_fd((1 << 32) - 1), _fd(U64)),                        # last slot the engine names, the largest ballot
        _ar(_fc(0), _fd(0)),                                      # zero in wide encodings
        _frame(_varint(0) + _varint(wire.HEARTBEAT) + _fd(U64) * 4),
        _frame(_varint(0) + _varint(wire.COMMIT_NOTICE) + _fd(U64) + _fc(0xFFFFFFFF)),
        _frame(_varint(2)),                                       # PeerMessage::Leave
        _frame(_fd(0) + _fd(3) + _varint(4) + _varint(5) + b"\x00"),   # the enum tags themselves in the widest encoding
    ]
    bad = [
        _frame(_varint(0) + _varint(3) + b"\xfe" + bytes(16) + _varint(1) + b"\x00"),                 # a u128 varint: not on this path
        _frame(_varint(0) + _varint(3) + b"\xff" + _varint(1) + b"\x00"),
        _frame(_varint(0) + _varint(3) + b"\xfd" + bytes(5)),                                         # a varint that runs off the frame
        _frame(_varint(0) + _varint(3) + _varint(1) + b"\xfb\x01"),
        _frame(_varint(0) + _varint(wire.HEARTBEAT) + _varint(1) + _varint(2) + _varint(3)),          # a field short
        _frame(_varint(0) + _varint(wire.COMMIT_NOTICE) + _varint(1) + _varint(2) + _varint(3)),      # a field too many
        _frame(_varint(0)),                                                                           # Msg without a variant
        struct.pack(">Q", 10 ** 12 + 1),
    ]
    tail = wire.accept_reply(3, 4)
    streams = [f + tail for f in ok] + [tail + f + tail for f in bad] + [b"".join(ok) + bad[0] + tail]
    # incomplete ends: every prefix of a frame behind a whole one
    for f in (wire.accept_reply(70000, 1 << 40), wire.heartbeat(1, 2, 3, 4)):
        streams += [tail + f[:k] for k in range(0, len(f))]
    na, nh, no, nm = _check(wire, cuda, streams, seed=2)
    assert nm == len(bad) + 1 and no >= 3


def test_ragged_wavefront_and_unaligned_buffer_end(cuda):
    """64 connections of one wavefront between 0 bytes and several windows long; the byte buffer ends 1 .. 15 bytes past a
    multiple of 16 with the last connection's last frame in those bytes"""
    from summerset_amd import wire
    rng = np.random.default_rng(3)
    for extra in (1, 7, 15):
        streams = []
        for c in range(64):
            n = int(rng.integers(0, 60)) if c % 5 else 0
            streams.append(b"".join(wire.accept_reply(int(rng.integers(0, 1 << 20)), 0x101) if rng.random() < 0.8 else
                                    wire.heartbeat(0x101, int(rng.integers(0, 300)), 0, 0) for _ in range(n)))
        total = sum(len(s) for s in streams)
        pad = (extra - total) % 16
        streams[-1] += _frame(_varint(1) + bytes(pad + 16 - 9)) if pad + 16 - 9 >= 0 else b""
        streams[-1] += wire.accept_reply(4242, 0x101)
        total = sum(len(s) for s in streams)
        na, nh, no, nm = _check(wire, cuda, streams, seed=4)
        assert nm == 0 and na > 1000


def test_frames_laid_out_by_hand_give_records_laid_out_by_hand(cuda):
    """no decoder of the product on the expected side.  Frames written byte by byte from the reference's
    definitions -- `[u64 BE length][bincode]` (utils/safetcp.rs:46,127-132), `PeerMessage::Msg` = variant 0
    (server/transport.rs:37-55), MultiPaxos `PeerMsg` variants in declaration order Prepare 0 ... AcceptReply 3 ... Heartbeat
    6, CommitNotice 7 (multipaxos/mod.rs:298-384), varints 1 / 3 / 5 / 9 bytes, AcceptReply's trailing Option<SystemTime> --
    and the records the parser must produce written out as plain tuples next to them."""
    from summerset_amd import wire
    from summerset_amd.multipaxos import ACK_DTYPE
    be = lambda n: struct.pack(">Q", n)                          # noqa: E731
    conns = [
        # connection 0 (group 7, peer 1): three AcceptReplies -- 1-byte varints, a u16 slot, a u32 slot with a u64 ballot
        (7, 1,
         be(5) + bytes([0, 3, 9, 17, 0])
         + be(7) + bytes([0, 3, 0xFB, 0x39, 0x30, 250, 0])                                   # slot 12345 = 0x3039 LE
         + be(17) + bytes([0, 3, 0xFC, 0x15, 0xCD, 0x5B, 0x07, 0xFD, 0x02, 0x01, 0, 0, 0, 0, 0, 0x80, 0]),   # 123456789, 0x8000000000000102
         [(7, 9, 17, 1, 0), (7, 12345, 250, 1, 0), (7, 123456789, 0x8000000000000102, 1, 0)], [], []),
        # connection 1 (group 70000, peer 4): Heartbeat {ballot 0x101 (u16), commit_bar 300 (u16), exec_bar 5, snap_bar 0}, an
        # AcceptReply {slot 2, ballot 3, Some(SystemTime {secs 1790000000 (u32), nanos 7})} = 11 bytes, CommitNotice {ballot 258
        # (u16), commit_bar 1}
        (70000, 4,
         be(10) + bytes([0, 6, 0xFB, 0x01, 0x01, 0xFB, 0x2C, 0x01, 5, 0])
         + be(11) + bytes([0, 3, 2, 3, 1, 0xFC]) + struct.pack("<I", 1790000000) + bytes([7])
         + be(6) + bytes([0, 7, 0xFB, 0x02, 0x01, 1]),
         [(70000, 2, 3, 4, 0)], [(70000, 4, 6, 0, 0x101, 300, 5, 0), (70000, 4, 7, 0, 258, 1, 0, 0)], []),
        # connection 2 (group 3, peer 0): Leave (PeerMessage variant 2), a Prepare {trigger 4, ballot 9}, then an AcceptReply: the first
        # two are only located
        (3, 0, be(1) + bytes([2]) + be(4) + bytes([0, 0, 4, 9]) + be(5) + bytes([0, 3, 1, 2, 0]),
         [(3, 1, 2, 0, 0)], [], [(0xFF, 1), (0, 4)]),            # (kind the host decoder names it by, frame payload length)
        # connection 3 (group 11, peer 2): an AcceptReply, then a frame whose Option tag is 2 -- malformed from there on
        (11, 2, be(5) + bytes([0, 3, 8, 8, 0]) + be(5) + bytes([0, 3, 8, 8, 2]) + be(5) + bytes([0, 3, 9, 9, 0]),
         [(11, 8, 8, 2, 0)], [], []),
    ]
    streams = [c[2] for c in conns]
    got = _ingest(wire, cuda, streams, [c[0] for c in conns], [c[1] for c in conns])
    acks = np.array([a for c in conns for a in c[3]], ACK_DTYPE)
    hbs = np.array([h for c in conns for h in c[4]], wire.HB_DTYPE)
    assert got["n_acks"] == len(acks) and np.array_equal(got["acks"], acks)
    assert got["n_hbs"] == len(hbs) and np.array_equal(got["hbs"], hbs)
    # located frames: (connection, offset of the frame in the whole buffer, bytes incl. the 8-byte length)
    base2 = len(streams[0]) + len(streams[1])
    assert got["n_others"] == 2
    assert [(int(o["conn"]), int(o["off"]), int(o["len"])) for o in got["others"]] == [(2, base2, 9), (2, base2 + 9, 12)]
    # every byte of connections 0 - 2 consumed; connection 3 stops in front of the bad frame (13 bytes in) with status 1
    assert list(got["consumed"]) == [len(streams[0]), len(streams[1]), len(streams[2]), 13]
    assert list(got["status"]) == [0, 0, 0, 1] and got["n_malformed"] == 1

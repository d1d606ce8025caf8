"""The dense call's laid-out edge streams (tests/test_zzz_wire_ingest_edges_gpu.py) through the one-pass ingest
(`smr_wire_ingest_mp_conn`, csrc/wire_ingest.hip; reference: src/server/transport.rs:404-470 -> utils/safetcp.rs:30-70, PeerMsg
multipaxos/mod.rs:298-384).  Written when the round's device minutes were used up: verified on the emulator
(tests/test_hostsim.py), first device run at round end -- stage 10, behind everything else, so that a failure here does not keep
the rest of the suite from running under `pytest -x`."""
import numpy as np
import pytest

from test_zz_wire_ingest_conn_gpu import _expected_conn, _ingest_conn, _same
from test_zz_wire_ingest_gpu import _frame, _varint

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300), pytest.mark.stage(10)]


def test_the_dense_calls_edge_streams_through_the_one_pass_call(cuda):
    """tests/test_zzz_wire_ingest_edges_gpu.py lays frames out on purpose -- every payload length either side of the straight-line
    path, headers and payloads at every window offset and across refills, extreme values, odd encodings, streams that end inside
    a frame, long located frames -- for the two-pass call; the same streams through the one-pass call, with room in every
    segment and with room for one Heartbeat / located frame only"""
    from summerset_amd import wire
    from summerset_amd.multipaxos import ACK_DTYPE
    import test_zzz_wire_ingest_edges_gpu as te
    U64 = te.U64
    streams = []
    for pad in range(0, 421, 3):                                               # (te.test_frames_around_the_ring_of_two_lines's layout)
        filler = _frame(_varint(1) + bytes((pad * 11 + i) & 0xFF for i in range(pad)))
        s = filler + wire.accept_reply(pad, 0x101) + wire.heartbeat(0x201, pad, 2, 1) + te._ar(te._fd(pad), te._fd(U64))
        if pad % 2 == 0:
            s += _frame(_varint(1) + bytes(200 + pad % 97)) + _frame(_varint(1) + bytes(127 + pad % 5))
        s += wire.accept_reply(65535 + pad, 70000) * (20 + pad % 23)
        if pad % 7 == 0:
            s += wire.heartbeat(1, 2, 3, 4)[:pad % 13]
        streams.append(s)
    for n in range(2, 26):                                                     # payload lengths around the straight-line path's end
        streams.append(_frame(_varint(0) + _varint(wire.HEARTBEAT) + te._fd(7)[:max(n - 2, 1)] + bytes(max(n - 11, 0))) + wire.accept_reply(n, 0x101))
    streams += [te._ar(te._fd(U64), te._fd(U64)) + wire.accept_reply(3, 4), te._ar(te._fd((1 << 32) - 1), te._fd(U64)),
                _frame(_varint(0) + _varint(wire.HEARTBEAT) + te._fd(U64) * 4), _frame(_varint(2)) + wire.accept_reply(3, 4),
                wire.accept_reply(3, 4) + _frame(_varint(0) + _varint(3) + b"\xfe" + bytes(16) + _varint(1) + b"\x00") + wire.accept_reply(3, 4)]
    rng = np.random.default_rng(23)
    groups, peers = rng.integers(0, 1 << 20, len(streams)), rng.integers(0, 5, len(streams))
    for hb_per, other_per in ((8, 8), (1, 1)):
        _, _, got = _ingest_conn(wire, cuda, streams, groups, peers, hb_per, other_per)
        _same(got, _expected_conn(wire, ACK_DTYPE, streams, groups, peers, hb_per, other_per))
    assert (got["status"] == 2).sum() > 50 and (got["status"] == 1).sum() >= 1


def test_frames_laid_out_by_hand_give_segments_laid_out_by_hand(cuda):
    """No decoder and no widening on the expected side: frames written byte by byte from the reference's definitions (the layouts of
    tests/test_zzz_wire_ingest_edges_gpu.py::test_frames_laid_out_by_hand_give_records_laid_out_by_hand) and the RAW output arrays of
    the one-pass call next to them -- which record of the ack array a connection's segment starts at (conn_off // 13), the 12-byte
    (slot, ballot lo, ballot hi) records themselves, the per-connection places of Heartbeats and located frames, the counts."""
    import struct
    import torch
    from summerset_amd import wire
    be = lambda n: struct.pack(">Q", n)                          # noqa: E731
    streams = [
        # connection 0 (group 7, peer 1): three AcceptReplies -- 1-byte varints, a u16 slot, a u32 slot with a u64 ballot: 13 + 15 + 25 bytes
        be(5) + bytes([0, 3, 9, 17, 0]) + be(7) + bytes([0, 3, 0xFB, 0x39, 0x30, 250, 0])
        + be(17) + bytes([0, 3, 0xFC, 0x15, 0xCD, 0x5B, 0x07, 0xFD, 0x02, 0x01, 0, 0, 0, 0, 0, 0x80, 0]),
        # connection 1 (group 70000, peer 4): Heartbeat {0x101, 300, 5, 0}, AcceptReply {2, 3, Some(SystemTime)}, CommitNotice {258, 1}: 18 + 19 + 14
        be(10) + bytes([0, 6, 0xFB, 0x01, 0x01, 0xFB, 0x2C, 0x01, 5, 0]) + be(11) + bytes([0, 3, 2, 3, 1, 0xFC]) + struct.pack("<I", 1790000000) + bytes([7])
        + be(6) + bytes([0, 7, 0xFB, 0x02, 0x01, 1]),
        # connection 2 (group 3, peer 0): Leave, Prepare {4, 9} -- located --, an AcceptReply {1, 2}: 9 + 12 + 13
        be(1) + bytes([2]) + be(4) + bytes([0, 0, 4, 9]) + be(5) + bytes([0, 3, 1, 2, 0]),
        # connection 3 (group 11, peer 2): an AcceptReply {8, 8}, then a frame whose Option tag is 2: malformed from there on
        be(5) + bytes([0, 3, 8, 8, 0]) + be(5) + bytes([0, 3, 8, 8, 2]) + be(5) + bytes([0, 3, 9, 9, 0]),
    ]
    assert [len(s) for s in streams] == [53, 51, 34, 39]
    ing, _, got = _ingest_conn(wire, cuda, streams, [7, 70000, 3, 11], [1, 4, 0, 2], 2, 2)
    raw = ing.acks.cpu().numpy().view(np.uint32).reshape(-1, 3)                # the ack array as 12-byte records
    assert ing.ack_cap == (53 + 51 + 34 + 39) // 13 + 1
    # segments start at conn_off // 13 = 0 // 13, 53 // 13, 104 // 13, 138 // 13
    assert [tuple(int(x) for x in raw[i]) for i in (0, 1, 2)] == [(9, 17, 0), (12345, 250, 0), (123456789, 0x00000102, 0x80000000)]
    assert tuple(int(x) for x in raw[4]) == (2, 3, 0) and tuple(int(x) for x in raw[8]) == (1, 2, 0) and tuple(int(x) for x in raw[10]) == (8, 8, 0)
    assert ing.cnt.cpu().numpy()[:4].tolist() == [[3, 0, 0], [1, 2, 0], [1, 0, 2], [1, 0, 0]]
    hbs = ing.hbs.cpu().numpy().view(wire.HB_DTYPE)                            # [connection][2]
    assert [tuple(int(x) for x in hbs[2 * 1 + k]) for k in (0, 1)] == [(70000, 4, 6, 0, 0x101, 300, 5, 0), (70000, 4, 7, 0, 258, 1, 0, 0)]
    others = ing.others.cpu().numpy().view(wire.OTHER_DTYPE)                  # [connection][2]: (conn, kind, offset in the buffer, bytes)
    assert [(int(o["conn"]), int(o["off"]), int(o["len"])) for o in others[4:6]] == [(2, 104, 9), (2, 113, 12)]
    assert list(got["consumed"]) == [53, 51, 34, 13] and list(got["status"]) == [0, 0, 0, 1]
    assert isinstance(ing.acks, torch.Tensor)

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


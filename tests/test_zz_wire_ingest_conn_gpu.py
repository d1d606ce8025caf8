"""`smr_wire_ingest_mp_conn` (round 5): MultiPaxos peer traffic parsed in ONE pass into a segment per connection (AcceptReplies as
12-byte (slot, ballot) records: the group and the peer are the connection's; `MpIngestConn.results` widens them), and
`smr_mp_deliver_acks_conn`, which takes the segments as they are.  The frames and their rules are `smr_wire_ingest_mp`'s
(tests/test_zz_wire_ingest_gpu.py): here the segments, gathered connection by connection, must BE the two-pass call's dense
lists -- the sequential decoder's -- wherever no connection was stopped, and a connection that has one Heartbeat or located
frame more than its segment holds must stop in front of that frame (status 2) exactly where a decoder with that rule
stops.  Reference: the receive side of `TcpTransport` (src/server/transport.rs:404-470 -> utils/safetcp.rs:30-70), PeerMsg
multipaxos/mod.rs:298-384."""
import numpy as np
import pytest

from test_zz_wire_ingest_gpu import _expected, _frame, _random_streams, _varint

pytestmark = pytest.mark.gpu


def _expected_conn(wire, ACK_DTYPE, streams, groups, peers, hb_per, other_per):
    """the sequential decoder, connection by connection, with the segments' rule: a Heartbeat / CommitNotice beyond hb_per or a
    located frame beyond other_per stops the connection in front of it (status 2)"""
    acks, hbs, others, cnt, consumed, status = [], [], [], [], [], []
    off = 0
    for c, s in enumerate(streams):
        pos, st, n = 0, 0, [0, 0, 0]
        while True:
            try:
                k, m = wire.decode(s[pos:])
            except Exception:
                st = 1
                break
            if k == 0:
                break
            kind = m["kind"]
            if kind == wire.ACCEPT_REPLY and m["slot"] < 1 << 32:
                acks.append((groups[c], m["slot"], m["ballot"], peers[c], 0)); n[0] += 1
            elif kind in (wire.HEARTBEAT, wire.COMMIT_NOTICE):
                if n[1] >= hb_per:
                    st = 2
                    break
                hbs.append((groups[c], peers[c], kind, 0, m["ballot"], m["commit_bar"], m["exec_bar"], m["snap_bar"])); n[1] += 1
            else:
                if n[2] >= other_per:
                    st = 2
                    break
                others.append((c, kind, off + pos, k)); n[2] += 1
            pos += k
        cnt.append(n); consumed.append(pos); status.append(st)
        off += len(s)
    arr = lambda rows, dt: np.array(rows, dt) if rows else np.zeros(0, dt)   # noqa: E731
    return (arr(acks, ACK_DTYPE), arr(hbs, wire.HB_DTYPE), arr(others, wire.OTHER_DTYPE), np.array(cnt, np.int64).reshape(-1, 3),
            np.array(consumed, np.int64), np.array(status, np.int32))


def _ingest_conn(wire, cuda, streams, groups, peers, hb_per, other_per, slack=0):
    import torch
    n = len(streams)
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in streams])
    blob = b"".join(streams)
    buf = torch.from_numpy(np.frombuffer(blob, np.uint8).copy()).to(cuda) if blob else torch.zeros(0, dtype=torch.uint8, device=cuda)
    ing = wire.MpIngestConn(n, len(blob) + slack, hb_per, other_per, device=cuda)
    # every output array sits in front of 256 bytes of 0xAB that must still be there afterwards: a segment that ran over its
    # array's end would write into them (the capacities handed to the call are the arrays' own)
    guards = []
    for name in ("acks", "hbs", "others", "cnt"):
        t = getattr(ing, name)
        nb = t.numel() * t.element_size()
        big = torch.full((nb + 256,), 0xAB, dtype=torch.uint8, device=cuda)
        big[:nb].zero_()
        setattr(ing, name, big[:nb].view(t.dtype).view(t.shape))
        guards.append((name, big, nb))
    d_off = torch.from_numpy(off).to(cuda)
    ing.ingest(buf, d_off, torch.from_numpy(np.asarray(groups, np.uint32).view(np.int32)).to(cuda), torch.from_numpy(np.asarray(peers, np.uint8)).to(cuda))
    res = ing.results()
    for name, big, nb in guards:
        assert bool((big[nb:] == 0xAB).all()), "the call wrote past the end of `%s`" % name
    return ing, d_off, res


def _same(got, want):
    acks, hbs, others, cnt, consumed, status = want
    assert np.array_equal(got["cnt"], cnt)
    assert np.array_equal(got["consumed"], consumed) and np.array_equal(got["status"], status)
    assert np.array_equal(got["acks"], acks) and np.array_equal(got["hbs"], hbs) and np.array_equal(got["others"], others)


def test_segments_are_the_sequential_decoders_lists(cuda):
    """random streams of every frame kind, incomplete tails and malformed frames: with room in every segment the gathered
    segments are the dense lists of the two-pass call (= `_expected` of the other file), consumed and status included"""
    from summerset_amd import wire
    from summerset_amd.multipaxos import ACK_DTYPE
    rng = np.random.default_rng(21)
    tot = 0
    for n_conn in (1, 63, 64, 200):
        streams = _random_streams(wire, rng, n_conn)
        groups, peers = rng.integers(0, 1 << 20, n_conn), rng.integers(0, 5, n_conn)
        _, _, got = _ingest_conn(wire, cuda, streams, groups, peers, 48, 48)
        want = _expected_conn(wire, ACK_DTYPE, streams, groups, peers, 48, 48)
        _same(got, want)
        a, h, o, consumed, status = _expected(wire, ACK_DTYPE, streams, groups, peers)      # ... which are the dense call's
        assert np.array_equal(got["acks"], a) and np.array_equal(got["hbs"], h) and np.array_equal(got["others"], o)
        assert np.array_equal(got["consumed"], consumed) and np.array_equal(got["status"], status)
        tot += len(a)
    assert tot > 1500


def test_a_full_segment_stops_the_connection_in_front_of_the_frame(cuda):
    """segments of 0, 1 and 2 Heartbeats / located frames per connection under the same random streams; the bytes a stopped
    connection did not consume, fed again, give the rest (what a host's read buffer does with `consumed`)"""
    from summerset_amd import wire
    from summerset_amd.multipaxos import ACK_DTYPE
    rng = np.random.default_rng(22)
    streams = _random_streams(wire, rng, 150)
    groups, peers = rng.integers(0, 1 << 20, 150), rng.integers(0, 5, 150)
    stopped = 0
    for hb_per, other_per in ((0, 0), (1, 2), (2, 1)):
        _, _, got = _ingest_conn(wire, cuda, streams, groups, peers, hb_per, other_per)
        want = _expected_conn(wire, ACK_DTYPE, streams, groups, peers, hb_per, other_per)
        _same(got, want)
        stopped += int((got["status"] == 2).sum())
    assert stopped > 150
    # the rest of every stopped connection, call after call, until nothing is left: all of the traffic, in order
    full = _expected(wire, ACK_DTYPE, streams, groups, peers)
    rest, n_acks, n_hbs, calls = list(streams), 0, 0, 0
    per_conn_acks = [[] for _ in streams]
    while any(rest) and calls < 60:
        _, _, got = _ingest_conn(wire, cuda, rest, groups, peers, 1, 1)
        first = np.concatenate([[0], np.cumsum(got["cnt"][:, 0])])
        for c in range(len(rest)):
            per_conn_acks[c].append(got["acks"][first[c]:first[c + 1]])
        n_hbs += int(got["cnt"][:, 1].sum())
        progressed = False
        for c in range(len(rest)):
            k = int(got["consumed"][c])
            progressed |= k > 0 and got["status"][c] == 2
            rest[c] = rest[c][k:] if got["status"][c] == 2 else b""
        calls += 1
        if not progressed:
            break
    assert calls > 3
    got_acks = np.concatenate([np.concatenate(a) if a else np.zeros(0, ACK_DTYPE) for a in per_conn_acks])
    assert np.array_equal(got_acks, full[0]) and n_hbs == len(full[1])


def test_edges(cuda):
    """no connections, empty connections, an offset table that runs off the buffer, an ack array too small for the segments,
    segments at the very end of the ack array"""
    import torch
    from summerset_amd import _lib, wire
    from summerset_amd.multipaxos import ACK_DTYPE
    _, _, got = _ingest_conn(wire, cuda, [], [], [], 2, 2)
    assert got["acks"].size == 0 and got["cnt"].shape == (0, 3)
    _, _, got = _ingest_conn(wire, cuda, [b"", b"", b""], [1, 2, 3], [0, 1, 2], 2, 2)
    assert (got["cnt"] == 0).all() and (got["consumed"] == 0).all() and (got["status"] == 0).all()
    f = wire.accept_reply(9, 7)                                                # 13 bytes: the shortest frame that makes a record
    assert len(f) == 13
    streams = [f * 7, b"", f * 3 + f[:5]]
    _, _, got = _ingest_conn(wire, cuda, streams, [4, 5, 6], [1, 2, 3], 0, 0)
    assert list(got["cnt"][:, 0]) == [7, 0, 3] and list(got["consumed"]) == [91, 0, 39] and len(got["acks"]) == 10
    ing = wire.MpIngestConn(2, 13 + 40, 1, 1, device=cuda)
    buf = torch.from_numpy(np.frombuffer(f, np.uint8).copy()).to(cuda)
    off = torch.tensor([0, len(f), len(f) + 40], dtype=torch.int64, device=cuda)
    ing.ingest(buf, off, torch.tensor([5, 6], dtype=torch.int32, device=cuda), torch.tensor([1, 2], dtype=torch.uint8, device=cuda))
    got = ing.results()
    assert list(got["cnt"][:, 0]) == [1, 0] and list(got["status"]) == [0, 1] and list(got["consumed"]) == [13, 0]
    small = wire.MpIngestConn(1, 13, 1, 1, device=cuda)
    small.ack_cap = 1                                                          # (buf_len // 13 + 1 = 2 are asked for)
    with pytest.raises(_lib.SummersetError):
        small.ingest(buf, off[:2].contiguous(), torch.tensor([5], dtype=torch.int32, device=cuda), torch.tensor([1], dtype=torch.uint8, device=cuda))


def _acks_over_the_wire_conn(eng, cuda, G, R, cap, t):
    """test_zz_wire_ingest_gpu._acks_over_the_wire through the one-pass call and smr_mp_deliver_acks_conn"""
    import torch
    from summerset_amd import wire
    from summerset_amd.multipaxos import ACK_DTYPE
    names = list(ACK_DTYPE.names)
    cache = {}
    for r in range(R):
        out = torch.zeros(cap * G * R * ACK_DTYPE.itemsize, dtype=torch.uint8, device=cuda)
        n = torch.zeros(1, dtype=torch.int64, device=cuda)
        eng.collect_acks(r, out, n)
        n0 = int(n.item())
        rec = out.cpu().numpy().view(ACK_DTYPE)[:n0].copy()
        rec = rec[np.lexsort((rec["slot"], rec["peer"], rec["group"]))]
        conns, streams = [], []
        for i in range(n0):
            key = (int(rec["group"][i]), int(rec["peer"][i]))
            if not conns or conns[-1] != key:
                conns.append(key)
                streams.append(bytearray(wire.heartbeat(0x101, 0, 0, 0)) if (key[0] + t) % 3 == 0 else bytearray())
            sb = (int(rec["slot"][i]), int(rec["ballot"][i]))
            if sb not in cache:
                cache[sb] = wire.accept_reply(*sb)
            streams[-1] += cache[sb]
        for k in range(0, len(streams), 5):
            streams[k] += wire.accept_reply(1 << 20, 0x101)[:9]
        ing, d_off, got = _ingest_conn(wire, cuda, [bytes(s) for s in streams], [g for g, _ in conns], [p for _, p in conns], 2, 2)
        assert int(got["cnt"][:, 0].sum()) == n0 and (got["status"] == 0).all() and int(got["cnt"][:, 2].sum()) == 0, (t, r)
        assert np.array_equal(got["acks"], rec), (t, r)
        eng.clear_acks(r)
        dropped = torch.zeros(1, dtype=torch.int64, device=cuda)
        if conns:
            eng.deliver_acks_conn(r, ing, dropped)
        eng.collect_acks(r, out, n)
        assert int(dropped.item()) == 0 and int(n.item()) == n0, (t, r)
        back = out.cpu().numpy().view(ACK_DTYPE)[:n0]
        assert np.array_equal(np.sort(back, order=names), np.sort(rec, order=names)), (t, r)


def test_accept_replies_over_the_wire_in_segments(cuda, oracle):
    """the acknowledgements of every tick travel as frames, come back through the one-pass call and `deliver_acks_conn`, and
    the cluster still matches the oracle after every tick (losses, leader changes, long re-Accept outboxes)"""
    import test_mp_gpu as t
    t._run(cuda, oracle, G=130, R=5, S=2, W=64, n_ticks=24, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, per_round=_acks_over_the_wire_conn)

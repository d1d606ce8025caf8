"""MultiPaxos peer traffic parsed on the device (the HIP half: `smr_wire_ingest_mp`,
csrc/wire_ingest.hip): byte streams of `[u64 BE length][bincode(PeerMessage)]` frames (src/utils/safetcp.rs:30-70,
127-132; PeerMsg multipaxos/mod.rs:298-384), one per connection, against the sequential host decoder `smr_wire_decode`
frame by frame -- every varint width, AcceptReplies with and without a timestamp, heartbeats, commit notices, the frames
the device only locates (Prepare, PrepareReply, Accepts longer than the window, Leave, lease traffic), incomplete tails,
malformed frames -- and then in the engine's tick: every replica's AcceptReplies of a tick leave as frames, come back
through the ingest kernel and `smr_mp_deliver_acks`, and the cluster still matches the oracle after every tick."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _varint(v):
    if v < 251:
        return bytes([v])
    if v < 1 << 16:
        return b"\xfb" + struct.pack("<H", v)
    if v < 1 << 32:
        return b"\xfc" + struct.pack("<I", v)
    return b"\xfd" + struct.pack("<Q", v)


def _frame(payload):
    return struct.pack(">Q", len(payload)) + payload


def _expected(wire, ACK_DTYPE, streams, groups, peers):
    """the sequential decoder over every connection: records in order, bytes consumed, status"""
    acks, hbs, others, consumed, status = [], [], [], [], []
    off = 0
    for c, s in enumerate(streams):
        pos, st = 0, 0
        while True:
            try:
                n, m = wire.decode(s[pos:])
            except Exception:
                st = 1
                break
            if n == 0:
                break
            k = m["kind"]
            if k == wire.ACCEPT_REPLY and m["slot"] < 1 << 32:
                acks.append((groups[c], m["slot"], m["ballot"], peers[c], 0))
            elif k in (wire.HEARTBEAT, wire.COMMIT_NOTICE):
                hbs.append((groups[c], peers[c], k, 0, m["ballot"], m["commit_bar"], m["exec_bar"], m["snap_bar"]))
            else:
                others.append((c, k, off + pos, n))
            pos += n
        consumed.append(pos); status.append(st)
        off += len(s)
    return (np.array(acks, ACK_DTYPE) if acks else np.zeros(0, ACK_DTYPE), np.array(hbs, wire.HB_DTYPE) if hbs else np.zeros(0, wire.HB_DTYPE),
            np.array(others, wire.OTHER_DTYPE) if others else np.zeros(0, wire.OTHER_DTYPE), np.array(consumed, np.int64), np.array(status, np.int32))


def _ingest(wire, cuda, streams, groups, peers, caps=None):
    import torch
    n = len(streams)
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in streams])
    blob = b"".join(streams)
    buf = torch.from_numpy(np.frombuffer(blob, np.uint8).copy()).to(cuda) if blob else torch.zeros(0, dtype=torch.uint8, device=cuda)
    total = sum(len(s) for s in streams) // 9 + 8
    ing = wire.MpIngest(n, *(caps or (total, total, total)), device=cuda)
    ing.ingest(buf, torch.from_numpy(off).to(cuda), torch.from_numpy(np.asarray(groups, np.uint32).view(np.int32)).to(cuda),
               torch.from_numpy(np.asarray(peers, np.uint8)).to(cuda))
    return ing.results()


def _random_streams(wire, rng, n_conn):
    big = [1, 250, 251, 65535, 65536, (1 << 32) - 1, 1 << 32, (1 << 63) + 5]
    val = lambda: int(rng.choice(big)) if rng.random() < 0.3 else int(rng.integers(0, 5000))   # noqa: E731
    streams = []
    for c in range(n_conn):
        s = bytearray()
        for _ in range(int(rng.integers(0, 40))):
            x = rng.random()
            if x < 0.55:
                s += wire.accept_reply(val(), val())
            elif x < 0.62:                   # AcceptReply with Some(timestamp): Duration { secs, nanos } since the epoch
                s += _frame(_varint(0) + _varint(3) + _varint(val()) + _varint(val()) + b"\x01" + _varint(1790000000) + _varint(int(rng.integers(0, 10**9))))
            elif x < 0.72:
                s += wire.heartbeat(val(), val(), val(), val())
            elif x < 0.78:
                s += wire.commit_notice(val(), val())
            elif x < 0.84:
                s += wire.prepare(val(), val())
            elif x < 0.90:                   # an Accept: up to several windows long
                reqs = wire.reqbatch([(7, i, ("put", "k%d" % i, "v" * int(rng.integers(1, 700)))) for i in range(int(rng.integers(0, 4)))])
                s += wire.accept(val(), val(), reqs)
            elif x < 0.94:
                s += wire.prepare_reply(val(), val(), val(), val(), None, val())
            elif x < 0.97:
                s += _frame(_varint(2))      # PeerMessage::Leave
            else:
                s += _frame(_varint(1) + bytes(rng.integers(0, 256, int(rng.integers(0, 90)), dtype=np.uint8)))   # lease traffic: skipped
        y = rng.random()
        if y < 0.25 and len(s):              # the last frame has not arrived completely
            f = wire.accept_reply(val(), val()) if rng.random() < 0.5 else wire.heartbeat(val(), val(), val(), val())
            s += f[:int(rng.integers(1, len(f)))]
        elif y < 0.33:                       # malformed frames of the kinds the device parses, then more bytes
            z = int(rng.integers(0, 7))
            if z == 0:
                s += struct.pack(">Q", 10**12 + 1) + b"\x00" * 20                       # invalidly large frame
            elif z == 1:
                s += _frame(_varint(0) + _varint(3) + _varint(5) + _varint(0x101) + b"\x02")   # Option tag 2
            elif z == 2:
                s += _frame(_varint(0) + _varint(3) + _varint(5) + _varint(0x101) + b"\x00\x00")   # a byte too many
            elif z == 3:
                s += _frame(_varint(0) + _varint(6) + _varint(5) + _varint(7))          # a Heartbeat two fields short
            elif z == 4:
                s += _frame(b"")                                                        # an empty payload
            else:                                                                       # fuzz: a hot kind's tag, then random bytes (the other
                kind = [wire.ACCEPT_REPLY, wire.HEARTBEAT, wire.COMMIT_NOTICE][int(rng.integers(0, 3))]   # kinds are located, not validated)
                s += _frame(_varint(0) + _varint(kind) + bytes(rng.integers(0, 256, int(rng.integers(0, 22)), dtype=np.uint8)))
            s += wire.accept_reply(1, 2)
        streams.append(bytes(s))
    return streams


def test_ingest_matches_the_sequential_decoder(cuda):
    from summerset_amd import wire
    from summerset_amd.multipaxos import ACK_DTYPE
    rng = np.random.default_rng(11)
    for n_conn in (1, 63, 64, 200):
        streams = _random_streams(wire, rng, n_conn)
        groups, peers = rng.integers(0, 1 << 20, n_conn), rng.integers(0, 5, n_conn)
        acks, hbs, others, consumed, status = _expected(wire, ACK_DTYPE, streams, groups, peers)
        got = _ingest(wire, cuda, streams, groups, peers)
        assert (got["n_acks"], got["n_hbs"], got["n_others"], got["n_malformed"]) == (len(acks), len(hbs), len(others), int(status.sum()))
        assert np.array_equal(got["consumed"], consumed) and np.array_equal(got["status"], status)
        assert np.array_equal(got["acks"], acks) and np.array_equal(got["hbs"], hbs) and np.array_equal(got["others"], others)
    assert len(acks) > 500 and len(hbs) > 100 and len(others) > 100 and status.sum() > 3


def test_empty_and_overfull(cuda):
    """no connections, connections without bytes, record capacities smaller than the traffic (counted, not stored), an
    offset table that runs off the buffer"""
    import torch
    from summerset_amd import wire
    from summerset_amd.multipaxos import ACK_DTYPE
    got = _ingest(wire, cuda, [], [], [], caps=(4, 4, 4))
    assert (got["n_acks"], got["n_hbs"], got["n_others"], got["n_malformed"]) == (0, 0, 0, 0)
    got = _ingest(wire, cuda, [b"", b"", b""], [1, 2, 3], [0, 1, 2], caps=(4, 4, 4))
    assert got["n_acks"] == 0 and (got["consumed"] == 0).all() and (got["status"] == 0).all()
    streams = [b"".join(wire.accept_reply(i, 0x101) for i in range(30)) + wire.heartbeat(0x101, 3, 2, 0) + wire.prepare(1, 2) for _ in range(70)]
    groups, peers = np.arange(70), np.arange(70) % 5
    acks, hbs, others, consumed, status = _expected(wire, ACK_DTYPE, streams, groups, peers)
    got = _ingest(wire, cuda, streams, groups, peers, caps=(100, 5, 1))
    assert (got["n_acks"], got["n_hbs"], got["n_others"]) == (2100, 70, 70)
    assert np.array_equal(got["acks"], acks[:100]) and np.array_equal(got["hbs"], hbs[:5]) and np.array_equal(got["others"], others[:1])
    assert np.array_equal(got["consumed"], consumed)
    # conn_off beyond the buffer: that connection is malformed, its neighbours are not
    ing = wire.MpIngest(2, 8, 8, 8, device=cuda)
    f = wire.accept_reply(9, 0x101)
    buf = torch.from_numpy(np.frombuffer(f, np.uint8).copy()).to(cuda)
    ing.ingest(buf, torch.tensor([0, len(f), len(f) + 40], dtype=torch.int64, device=cuda), torch.tensor([5, 6], dtype=torch.int32, device=cuda),
               torch.tensor([1, 2], dtype=torch.uint8, device=cuda))
    got = ing.results()
    assert got["n_acks"] == 1 and got["n_malformed"] == 1 and list(got["status"]) == [0, 1] and list(got["consumed"]) == [len(f), 0]


def _acks_over_the_wire(eng, cuda, G, R, cap, t):
    """between R2 and R3: every replica's acknowledgements leave as AcceptReply frames on one connection per (group, peer)
    -- a heartbeat in front, an incomplete frame behind --, the matrix is zeroed, the frames come back through the
    ingest kernel and smr_mp_deliver_acks; a second collect must give the same set"""
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
        got = _ingest(wire, cuda, [bytes(s) for s in streams], [g for g, _ in conns], [p for _, p in conns])
        assert got["n_acks"] == n0 and got["n_malformed"] == 0 and got["n_others"] == 0, (t, r)
        assert np.array_equal(got["acks"], rec), (t, r)          # the sequential decoder's order = the streams' order
        eng.clear_acks(r)
        dev = torch.from_numpy(got["acks"].view(np.uint8).reshape(-1).copy()).to(cuda) if n0 else torch.zeros(24, dtype=torch.uint8, device=cuda)
        dropped = torch.zeros(1, dtype=torch.int64, device=cuda)
        eng.deliver_acks(r, dev, n0, dropped)
        eng.collect_acks(r, out, n)
        assert int(dropped.item()) == 0 and int(n.item()) == n0, (t, r)
        back = out.cpu().numpy().view(ACK_DTYPE)[:n0]
        assert np.array_equal(np.sort(back, order=names), np.sort(rec, order=names)), (t, r)


def test_accept_replies_over_the_wire(cuda, oracle):
    """steady appends, losses and leader changes with their long re-Accept outboxes: the acknowledgements of every tick
    travel as frames and the cluster still matches the oracle after every tick"""
    import test_mp_gpu as t
    t._run(cuda, oracle, G=130, R=5, S=2, W=64, n_ticks=24, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, per_round=_acks_over_the_wire)


def test_ingest_against_what_the_test_wrote(cuda):
    """No product decoder and no product encoder in the loop: frames laid out here from the reference's own type definitions --
    `[u64 BE len]` (safetcp.rs:46,127-132) + bincode-standard `PeerMessage::Msg { msg: PeerMsg::X { .. } }` with the variant
    indexes of multipaxos/mod.rs:298-384 (Prepare 0 .. AcceptReply 3 .. Heartbeat 6, CommitNotice 7) and SURVEY Appendix C's
    varint rule -- and the records the ingest kernel makes of them compared with the values the test put in, field by field,
    in stream order.  (The rest of this file holds the parser against `smr_wire_decode`; this test holds it -- and through the
    first one that decoder -- against the layout itself.)"""
    from summerset_amd import wire
    from summerset_amd.multipaxos import ACK_DTYPE
    rng = np.random.default_rng(5)
    edge = [0, 1, 250, 251, 252, 65535, 65536, (1 << 32) - 1]
    wide = [1 << 32, (1 << 40) + 3, (1 << 63) + 5, (1 << 64) - 1]
    n_conn = 130
    groups, peers = rng.integers(0, 1 << 20, n_conn), rng.integers(0, 5, n_conn)
    streams, want_acks, want_hbs = [], [], []
    for c in range(n_conn):
        s = bytearray()
        for _ in range(int(rng.integers(1, 30))):
            pick = lambda lst: lst[int(rng.integers(0, len(lst)))]       # noqa: E731 -- (by index: numpy would turn 2^64 - 1 into a float)
            v = lambda big=False: pick(edge + (wide if big else [])) if rng.random() < 0.4 else int(rng.integers(0, 70000))   # noqa: E731
            x = rng.random()
            if x < 0.6:                                                  # AcceptReply { slot, ballot, reply_ts: None | Some(SystemTime) }
                slot, ballot = v(), v(True)
                ts = b"\x00" if rng.random() < 0.7 else b"\x01" + _varint(1790000000 + int(rng.integers(0, 1000))) + _varint(int(rng.integers(0, 10**9)))
                s += _frame(_varint(0) + _varint(3) + _varint(slot) + _varint(ballot) + ts)
                want_acks.append((groups[c], slot, ballot, peers[c], 0))
            elif x < 0.8:                                                # Heartbeat { ballot, commit_bar, exec_bar, snap_bar }
                f = [v(True), v(), v(), v()]
                s += _frame(_varint(0) + _varint(6) + b"".join(_varint(q) for q in f))
                want_hbs.append((groups[c], peers[c], 6, 0, f[0], f[1], f[2], f[3]))
            else:                                                        # CommitNotice { ballot, commit_bar }
                f = [v(True), v()]
                s += _frame(_varint(0) + _varint(7) + _varint(f[0]) + _varint(f[1]))
                want_hbs.append((groups[c], peers[c], 7, 0, f[0], f[1], 0, 0))
        streams.append(bytes(s))
    got = _ingest(wire, cuda, streams, groups, peers)
    assert got["n_malformed"] == 0 and got["n_others"] == 0
    assert np.array_equal(got["consumed"], [len(s) for s in streams])
    acks, hbs = np.array(want_acks, ACK_DTYPE), np.array(want_hbs, wire.HB_DTYPE)
    assert got["n_acks"] == len(acks) and np.array_equal(got["acks"], acks)
    assert got["n_hbs"] == len(hbs)
    for name in ("group", "peer", "kind", "ballot", "commit_bar"):
        assert np.array_equal(got["hbs"][name], hbs[name]), name
    hb6 = hbs["kind"] == 6                                               # a CommitNotice carries no exec / snap bar
    assert np.array_equal(got["hbs"]["exec_bar"][hb6], hbs["exec_bar"][hb6]) and np.array_equal(got["hbs"]["snap_bar"][hb6], hbs["snap_bar"][hb6])
    assert len(acks) > 800 and len(hbs) > 300

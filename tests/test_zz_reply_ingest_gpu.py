"""Raft and EPaxos reply traffic parsed on the device (`smr_wire_ingest_raft_replies`, `smr_wire_ingest_ep_pre_accept_replies`,
csrc/wire_ingest_replies.hip): the leader's connections deliver `[u64 BE length][bincode(PeerMessage)]` frames
(safetcp.rs:30-70, 127-132) and the parser fills the arrays [R][G] the engines' reply handlers take, one reply per (peer,
group) and call.

Frame by frame: the frames are laid out HERE from the reference's type definitions -- `PeerMessage::Msg { msg }` = enum
tag 0, then the protocol's `PeerMsg` variant index (Raft: AppendEntries 0, AppendEntriesReply 1, RequestVote 2,
RequestVoteReply 3, raft/mod.rs:203-234; EPaxos: PreAccept 0, PreAcceptReply 1, Accept 2, AcceptReply 3, CommitNotice 4,
ExpPrepare 5 .., epaxos/mod.rs:306-377), fields in declaration order, bincode-standard varints,
`Option` = a tag byte, `SlotIdx(ReplicaId = u8, usize)` -- no product encoder or decoder in the loop; the arrays are compared
with the values the test put into the frames.  Through the engine: the closed-loop clusters of tests/raft_cluster.py and
tests/ep_cluster.py with every AppendEntriesReply / PreAcceptReply travelling as a frame through the parser, against the
oracles wired directly."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NONE32 = 0xFFFFFFFF


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


def _raft_reply(term, end_slot, conflict=None):
    body = _varint(0) + _varint(1) + _varint(term) + _varint(end_slot)
    return _frame(body + (b"\x00" if conflict is None else b"\x01" + _varint(conflict[0]) + _varint(conflict[1])))


def _ep_reply(row, col, ballot, seq, deps):
    body = _varint(0) + _varint(1) + bytes([row]) + _varint(col) + _varint(ballot) + _varint(seq) + _varint(len(deps))
    return _frame(body + b"".join(b"\x00" if d is None else b"\x01" + _varint(d) for d in deps))


def _layout(torch, cuda, streams, groups, peers):
    off = np.zeros(len(streams) + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in streams])
    blob = b"".join(streams)
    buf = torch.from_numpy(np.frombuffer(blob + b"\x00", np.uint8)[:len(blob)].copy()).to(cuda)
    return (buf, torch.from_numpy(off).to(cuda), torch.from_numpy(np.asarray(groups, np.int32)).to(cuda),
            torch.from_numpy(np.asarray(peers, np.uint8)).to(cuda), off)


def _pick(rng, wide=False):
    edge = [0, 1, 250, 251, 65535, 65536, (1 << 32) - 1] + ([1 << 32, (1 << 63) + 5, (1 << 64) - 1] if wide else [])
    return edge[int(rng.integers(0, len(edge)))] if rng.random() < 0.4 else int(rng.integers(0, 70000))


def test_raft_replies_frame_by_frame(cuda):
    """every (group, peer) of 97 groups x 5 replicas a connection: other Raft frames (a RequestVoteReply, an empty AppendEntries,
    lease traffic, Leave) around the reply; a second reply behind it (the next call's); replies with and without `conflict`,
    64-bit terms, slots beyond u32 (located, not taken); incomplete tails; malformed connections"""
    import torch
    from summerset_amd import wire
    rng = np.random.default_rng(11)
    G, R = 97, 5
    junk = [_frame(_varint(0) + _varint(3) + _varint(7) + b"\x01"),                                   # RequestVoteReply { term, granted }
            _frame(_varint(0) + _varint(0) + _varint(3) + _varint(9) + _varint(2) + _varint(0) + _varint(8) + _varint(0)),   # AppendEntries, no entries
            _frame(_varint(1) + bytes(range(30))), _frame(_varint(2))]                                 # lease traffic; PeerMessage::Leave
    streams, groups, peers = [], [], []
    want = dict(term=np.zeros((R, G), np.uint64), end=np.zeros((R, G), np.uint32), ct=np.zeros((R, G), np.uint64),
                cs=np.zeros((R, G), np.uint32), fl=np.zeros((R, G), np.uint8))
    consumed, status, n_loc, n_def = [], [], 0, 0
    for g in range(G):
        for p in range(R):
            s, took, c, st, x = bytearray(), False, 0, 0, rng.random()
            for _ in range(int(rng.integers(0, 3))):
                s += junk[int(rng.integers(0, len(junk)))]; n_loc += 1
            c = len(s)
            if x < 0.8:                                                                                # a reply
                term, end = _pick(rng, True), _pick(rng)
                conflict = (_pick(rng, True), _pick(rng)) if rng.random() < 0.4 else None
                s += _raft_reply(term, end, conflict)
                want["term"][p, g], want["end"][p, g], want["fl"][p, g] = term, end, 1 | (2 if conflict else 0)
                if conflict:
                    want["ct"][p, g], want["cs"][p, g] = conflict
                took, c = True, len(s)
            elif x < 0.86:                                                                             # a slot the engine cannot name: located
                s += _raft_reply(3, (1 << 32) + int(rng.integers(0, 9))); n_loc += 1; c = len(s)
            y = rng.random()
            if y < 0.2:
                s += junk[0]; n_loc += 1; c = len(s)
            if took and y > 0.85:                                                                      # a second reply: not this call's
                s += _raft_reply(5, 6) + junk[3]; n_def += 1
            elif y > 0.7:
                s += _raft_reply(1 << 40, 77)[:int(rng.integers(1, 12))]                               # an incomplete frame
            elif 0.6 < y < 0.64:                                                                       # malformed: a reply that ends a byte before its length says
                s += struct.pack(">Q", 6) + _varint(0) + _varint(1) + _varint(1) + _varint(2) + b"\x00\x00"; st = 1
            elif 0.64 <= y < 0.67:
                s += _frame(_varint(0) + _varint(9)); st = 1                                           # no such Raft message
            if st:
                c, n_loc = 0, n_loc                                                                    # (its located frames were counted before it broke)
            streams.append(bytes(s)); groups.append(g); peers.append(p); consumed.append(c); status.append(st)
    buf, off, grp, peer, _ = _layout(torch, cuda, streams, groups, peers)
    ing = wire.ReplyIngest(len(streams), G, R, 4096, cuda)
    out = {k: v.cpu().numpy() for k, v in ing.raft(buf, off, grp, peer).items()}
    res = ing.results()
    fl = out["flags"]
    taken_ok = np.array(status).reshape(G, R).T == 0
    assert np.array_equal(fl[taken_ok], want["fl"][taken_ok])
    assert np.array_equal(fl != 0, want["fl"] != 0)                  # (a malformed connection's earlier reply stays delivered)
    m = want["fl"] != 0
    assert np.array_equal(out["reply_term"].view(np.uint64)[m], want["term"][m]) and np.array_equal(out["end_slot"].view(np.uint32)[m], want["end"][m])
    mc = (want["fl"] & 2) != 0
    assert np.array_equal(out["conflict_term"].view(np.uint64)[mc], want["ct"][mc]) and np.array_equal(out["conflict_slot"].view(np.uint32)[mc], want["cs"][mc])
    assert np.array_equal(res["status"], status) and np.array_equal(res["consumed"], consumed)
    assert res["n_malformed"] == sum(status) and res["n_deferred"] == n_def and res["n_others"] == n_loc
    assert res["n_replies"] == int(((want["fl"] != 0) & taken_ok).sum())
    assert len(res["others"]) == n_loc and m.sum() > 300 and mc.sum() > 100 and n_def > 20 and sum(status) > 10
    # the located frames: whole frames of the buffer, each once
    blob = b"".join(streams)
    seen = set()
    for o in res["others"]:
        a, n = int(o["off"]), int(o["len"])
        assert struct.unpack(">Q", blob[a:a + 8])[0] == n - 8 and (a, n) not in seen
        seen.add((a, n))


def test_ep_pre_accept_replies_frame_by_frame(cuda):
    """5 replicas, me = 2: PreAcceptReplies for (me, col[g]) with None / small / wide dependencies; replies for another row, another
    column, a dependency list of another length or with an index beyond u32 (all located, not taken); AcceptReplies,
    CommitNotice-sized frames, heartbeats between them; second replies; incomplete and malformed streams"""
    import torch
    from summerset_amd import wire
    rng = np.random.default_rng(12)
    G, R, me = 83, 5, 2
    col = rng.integers(0, 70000, G).astype(np.uint32)
    col[::7] = (1 << 32) - 2
    dep = lambda: None if rng.random() < 0.3 else min(_pick(rng), NONE32 - 1)                          # noqa: E731 -- (Some(2^32 - 1) is the arrays' None)
    junk = [_frame(_varint(0) + _varint(3) + bytes([1]) + _varint(5) + _varint(2)),                    # AcceptReply { slot, ballot }
            _frame(_varint(0) + _varint(9) + _varint(5) + b"".join(_varint(3) for _ in range(5)) + _varint(1)),   # Heartbeat { exec_bars, snap_bar }
            _frame(_varint(2))]
    streams, groups, peers, consumed, status = [], [], [], [], []
    wb, ws = np.zeros((R, G), np.uint64), np.zeros((R, G), np.uint64)
    wd, wf = np.full((R, R, G), NONE32, np.uint32), np.zeros((R, G), np.uint8)
    n_loc = n_def = 0
    for g in range(G):
        for p in range(R):
            if p == me:
                continue
            s, c, st, took = bytearray(), 0, 0, False
            if rng.random() < 0.3:
                s += junk[int(rng.integers(0, 3))]; n_loc += 1
            x = rng.random()
            if x < 0.7:
                b, q, d = _pick(rng, True), _pick(rng, True), [dep() for _ in range(R)]
                s += _ep_reply(me, int(col[g]), b, q, d)
                wb[p, g], ws[p, g], wf[p, g] = b, q, 1
                wd[p, :, g] = [NONE32 if v is None else v for v in d]
                took = True
            elif x < 0.76:
                s += _ep_reply((me + 1) % R, int(col[g]), 3, 4, [None] * R); n_loc += 1               # another row
            elif x < 0.82:
                s += _ep_reply(me, int(col[g]) ^ 1, 3, 4, [1] * R); n_loc += 1                        # another column
            elif x < 0.86:
                s += _ep_reply(me, int(col[g]), 3, 4, [1] * (R - 1)); n_loc += 1                      # four dependencies
            elif x < 0.9:
                s += _ep_reply(me, int(col[g]), 3, 4, [1, None, (1 << 32) + 1 if g % 2 else NONE32, 2, 3]); n_loc += 1   # an index the engine cannot name
            c = len(s)
            y = rng.random()
            if took and y < 0.15:
                s += _ep_reply(me, int(col[g]), 9, 9, [None] * R); n_def += 1
            elif y < 0.3:
                s += junk[0]; n_loc += 1; c = len(s)
                s += _ep_reply(me, int(col[g]), 1 << 40, 1, [7] * R)[:int(rng.integers(1, 15))]
            elif 0.5 < y < 0.54:
                s += struct.pack(">Q", 9) + _varint(0) + _varint(1) + bytes([me]) + _varint(1) + _varint(1) + _varint(1) + _varint(2) + b"\x00\x02"; st = 1   # Option tag 2
            elif 0.54 <= y < 0.57:
                s += _frame(_varint(0) + _varint(1) + bytes([me]) + _varint(1) + _varint(1) + _varint(1) + _varint(0) + b"\x00"); st = 1   # a byte too many
            streams.append(bytes(s)); groups.append(g); peers.append(p); consumed.append(0 if st else c); status.append(st)
    buf, off, grp, peer, _ = _layout(torch, cuda, streams, groups, peers)
    ing = wire.ReplyIngest(len(streams), G, R, 4096, cuda)
    out = {k: v.cpu().numpy() for k, v in ing.ep_pre_accept(buf, off, grp, peer, me, torch.from_numpy(col.view(np.int32)).to(cuda)).items()}
    res = ing.results()
    assert np.array_equal(out["flags"], wf)
    m = wf != 0
    assert np.array_equal(out["ballot"].view(np.uint64)[m], wb[m]) and np.array_equal(out["seq"].view(np.uint64)[m], ws[m])
    md = np.broadcast_to(m[:, None, :], (R, R, G))
    assert np.array_equal(out["deps"].view(np.uint32)[md], wd[md])
    assert np.array_equal(res["status"], status) and np.array_equal(res["consumed"], consumed)
    bad = np.zeros((R, G), bool)
    for i, stt in enumerate(status):
        bad[peers[i], groups[i]] = stt != 0
    assert res["n_malformed"] == sum(status) and res["n_deferred"] == n_def and res["n_others"] == n_loc
    assert res["n_replies"] == int((m & ~bad).sum()) and m.sum() > 200 and n_loc > 60 and n_def > 10 and sum(status) > 5


def _raft_via(cuda, G, R, ing_box):
    import torch
    from summerset_amd import wire

    def via(s, rt, es, fl, ct, cs):
        streams, groups, peers = [], [], []
        for q in range(R):
            for g in range(G):
                if q == s:
                    continue
                f = bytearray()
                if (g + q) % 11 == 0:
                    f += wire.raft_request_vote_reply(1, False)                                       # a frame the parser only locates
                if fl[q, g] & 1:
                    f += wire.raft_append_entries_reply(int(rt[q, g]), int(es[q, g]), (int(ct[q, g]), int(cs[q, g])) if fl[q, g] & 2 else None)
                streams.append(bytes(f)); groups.append(g); peers.append(q)
        buf, off, grp, peer, _ = _layout(torch, cuda, streams, groups, peers)
        if ing_box[0] is None:
            ing_box[0] = wire.ReplyIngest(len(streams), G, R, len(streams), cuda)
        o = {k: v.cpu().numpy() for k, v in ing_box[0].raft(buf, off, grp, peer).items()}
        res = ing_box[0].results()
        assert res["n_malformed"] == 0 and res["n_deferred"] == 0 and res["n_replies"] == int((fl & 1).sum())
        pres = o["flags"] != 0
        z64, z32 = np.zeros((R, G), np.uint64), np.zeros((R, G), np.uint32)
        conf = (o["flags"] & 2) != 0
        return (np.where(pres, o["reply_term"].view(np.uint64), z64), np.where(pres, o["end_slot"].view(np.uint32), z32), o["flags"].copy(),
                np.where(conf, o["conflict_term"].view(np.uint64), z64), np.where(conf, o["conflict_slot"].view(np.uint32), z32))
    return via


def test_raft_cluster_replies_over_the_wire(cuda, oracle):
    """tests/raft_cluster.py's closed loop (elections, appends, conflicts after the second election) with every AppendEntriesReply
    of the device cluster sent as a frame and parsed back on the device; the oracle cluster is wired directly"""
    import raft_cluster as rc
    from summerset_amd import RaftLeaderGroup
    G, W, K, R = 300, 64, 8, 5
    engs = [rc.NumpyRaft(RaftLeaderGroup(G, R, leader_id=r, window=W, term=1), cuda) for r in range(R)]
    orcs = [oracle.RaftOracle(G, R, W, leader_id=r, term=1) for r in range(R)]
    for x in engs + orcs:
        x.preset(rc.FOLLOWER, 0xFF, 0)
    rng = np.random.default_rng(9)
    none = np.full((R, G), 0xFF, np.uint8)
    via = _raft_via(cuda, G, R, [None])
    to = none.copy()
    to[np.arange(G) % R, np.arange(G)] = 0xFE
    rc.tick(engs, to, np.zeros((R, G), np.uint32), K, via=via)
    rc.tick(orcs, to, np.zeros((R, G), np.uint32), K)
    for t in range(12):
        n_new = rng.integers(0, 4, (R, G)).astype(np.uint32)
        to = none.copy()
        if t == 5:
            gs = np.arange(0, G, 3)
            to[(gs + 2) % R, gs] = (gs % R).astype(np.uint8)
        rc.tick(engs, to, n_new, K, via=via)
        rc.tick(orcs, to, n_new, K)
        for r in range(R):
            a, b = engs[r].dump(), orcs[r].dump()
            for n in b:
                assert np.array_equal(a[n], b[n]), (t, r, n)
    d = [o.dump() for o in orcs]
    assert int(np.stack([x["last_commit"] for x in d]).max(axis=0).min()) > 3


def test_ep_cluster_pre_accept_replies_over_the_wire(cuda, oracle):
    """tests/ep_cluster.py's closed loop (PreAccept fan-out with loss, fast and slow path) with every PreAcceptReply of the device
    cluster sent as a frame and parsed back on the device; the oracle cluster is wired directly"""
    import torch
    import ep_cluster as ec
    from summerset_amd import EPaxosReplicaGroup, wire
    G, R, W, K, T = 260, 5, 32, 6, 8
    engs = [ec.NumpyEngine(EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K), cuda) for r in range(R)]
    orcs = [oracle.EpOracle(G, R, me=r, W=W, n_keys=K) for r in range(R)]
    rng = np.random.default_rng(5)
    box = [None]

    def via(s, col, ballot, seq, deps, flags):
        streams, groups, peers = [], [], []
        for q in range(R):
            if q == s:
                continue
            for g in range(G):
                f = bytearray()
                if (g + q) % 9 == 0:
                    f += wire.ep_msg(wire.EP_ACCEPT_REPLY, s, 1, 7)                                   # located only
                if flags[q, g] & 1:
                    f += wire.ep_msg(wire.EP_PRE_ACCEPT_REPLY, s, int(col[g]), int(ballot[q, g]), int(seq[q, g]),
                                     [None if int(d) == NONE32 else int(d) for d in deps[q, :, g]])
                streams.append(bytes(f)); groups.append(g); peers.append(q)
        buf, off, grp, peer, _ = _layout(torch, cuda, streams, groups, peers)
        if box[0] is None:
            box[0] = wire.ReplyIngest(len(streams), G, R, len(streams), cuda)
        o = {k: v.cpu().numpy() for k, v in box[0].ep_pre_accept(buf, off, grp, peer, s, torch.from_numpy(np.ascontiguousarray(col).view(np.int32)).to(cuda)).items()}
        res = box[0].results()
        assert res["n_malformed"] == 0 and res["n_deferred"] == 0 and res["n_replies"] == int((flags & 1).sum())
        pres = o["flags"] != 0
        return (np.where(pres, o["ballot"].view(np.uint64), 0).astype(np.uint64), np.where(pres, o["seq"].view(np.uint64), 0).astype(np.uint64),
                np.where(pres[:, None, :], o["deps"].view(np.uint32), NONE32).astype(np.uint32), o["flags"].copy())

    fast = slow = 0
    for t in range(T):
        keys = ec.zipf_keys(rng, R, G, K)
        drop = {(s, q): rng.random(G) < 0.15 for s in range(R) for q in range(R) if s != q}
        oe, oo = ec.tick(engs, keys, drop, via=via), ec.tick(orcs, keys, drop)
        for s in range(R):
            for k in oo[s]:
                assert np.array_equal(oe[s][k], oo[s][k]), (t, s, k)
            fast += int((oo[s]["decision"] == 3).sum()); slow += int((oo[s]["decision"] == 2).sum())
    for r in range(R):
        a, b = engs[r].dump(), orcs[r].dump()
        for n in b:
            assert np.array_equal(a[n], b[n]), (r, n)
    assert fast > 0 and slow > 0


def test_reply_ingest_argument_edges(cuda):
    """no connections at all; a connection whose group / peer the arrays have no place for (malformed, nothing taken); located
    frames beyond `other_cap` are counted, not stored; a byte buffer that is not 16-byte aligned is refused; more connections in
    a block's span than its stage holds (the lanes behind read from the buffer)"""
    import torch
    from summerset_amd import SummersetError, wire
    G, R = 40, 5
    ing = wire.ReplyIngest(0, G, R, 4, cuda)
    z = torch.zeros(0, dtype=torch.uint8, device=cuda)
    o = ing.raft(z, torch.zeros(1, dtype=torch.int64, device=cuda), torch.zeros(0, dtype=torch.int32, device=cuda), z)
    assert int(o["flags"].sum().item()) == 0 and ing.results()["n_replies"] == 0
    # out-of-range descriptors, an other_cap of 2 against 5 located frames
    junk = _frame(_varint(1) + b"abc")
    streams = [_raft_reply(3, 4), _raft_reply(5, 6), _raft_reply(7, 8), junk * 5 + _raft_reply(9, 10)]
    buf, off, grp, peer, _ = _layout(torch, cuda, streams, [1, G, 2, 3], [1, 1, R, 2])
    ing = wire.ReplyIngest(4, G, R, 2, cuda)
    o = {k: v.cpu().numpy() for k, v in ing.raft(buf, off, grp, peer).items()}
    res = ing.results()
    assert list(res["status"]) == [0, 1, 1, 0] and list(res["consumed"]) == [len(streams[0]), 0, 0, len(streams[3])]
    assert res["n_replies"] == 2 and res["n_malformed"] == 2 and res["n_others"] == 5 and len(res["others"]) == 2
    assert o["flags"].sum() == 2 and o["flags"][1, 1] == 1 and o["flags"][2, 3] == 1 and int(o["end_slot"][2, 3]) == 10
    # alignment
    big = torch.zeros(64, dtype=torch.uint8, device=cuda)
    with pytest.raises(SummersetError):
        ing.raft(big[1:33], off, grp, peer)
    # a block of 1024 connections whose streams span more than the 32 KB a block stages: 60-byte filler in front of every reply
    n = 1100
    fill = _frame(_varint(1) + bytes(51))
    streams = [fill + _raft_reply(100 + c, c) for c in range(n)]
    buf, off, grp, peer, _ = _layout(torch, cuda, streams, np.arange(n) % G, np.ones(n, np.int64))
    ing = wire.ReplyIngest(n, G, R, n, cuda)
    o = {k: v.cpu().numpy() for k, v in ing.raft(buf, off, grp, peer).items()}
    res = ing.results()
    assert res["n_replies"] == n and res["n_others"] == n and res["n_malformed"] == 0 and list(res["consumed"]) == [len(x) for x in streams]
    assert sum(len(x) for x in streams[:1024]) > 32 * 1024
    last = {g: max(c for c in range(n) if c % G == g) for g in range(G)}          # (several connections per (group, peer): the caller's error -- one wins)
    assert all(int(o["end_slot"][1, g]) % G == g for g in range(G)) and set(int(o["reply_term"][1, g]) - 100 for g in range(G)) <= set(range(n)) and last


def run_fused_raft_wire_replies(cuda, oracle, G=300, R=5, W=64, T=6, seed=21, me=2):
    """`smr_raft_leader_handle_wire_replies` (round 6: the parse as the prologue of the leader's reply handler, ONE launch, dense
    connections) against the two calls it stands for -- `smr_wire_ingest_raft_replies` + `smr_raft_leader_handle_replies` -- on a
    second leader in the same state, and both against the oracle fed the decoded replies directly: T ticks of appends and replies
    with conflicts, higher terms, junk frames around the replies, second replies (deferred), incomplete tails and malformed
    connections.  Counts, located frames, `consumed`, `status` and the leaders' state every tick."""
    import torch
    from summerset_amd import RaftLeaderGroup, wire
    rng = np.random.default_rng(seed)
    F = R - 1
    a, b = RaftLeaderGroup(G, R, leader_id=me, window=W, term=2), RaftLeaderGroup(G, R, leader_id=me, window=W, term=2)
    orc = oracle.RaftOracle(G, R, W, leader_id=me, term=2) if oracle is not None else None
    ing_a, ing_b = wire.ReplyIngest(G * F, G, R, 4 * G * F, cuda), wire.ReplyIngest(G * F, G, R, 4 * G * F, cuda)
    peers_of = [p for p in range(R) if p != me]
    grp = torch.from_numpy(np.repeat(np.arange(G), F).astype(np.int32)).to(cuda)
    peer = torch.from_numpy(np.tile(np.array(peers_of, np.uint8), G)).to(cuda)
    junk = [_frame(_varint(0) + _varint(3) + _varint(7) + b"\x01"), _frame(_varint(2)), _frame(_varint(1) + bytes(range(20)))]
    n_exec = 0
    for t in range(T):
        n_new = rng.integers(0, 4, G).astype(np.int32)
        for x in (a, b):
            x.handle_req_batch(torch.from_numpy(n_new).to(cuda))
        if orc is not None:
            orc.append(n_new.astype(np.uint32))
        len_now = a.dump()["log_len"].astype(np.int64)
        rt, es, fl = np.zeros((R, G), np.uint64), np.zeros((R, G), np.uint32), np.zeros((R, G), np.uint8)
        ct, cs = np.zeros((R, G), np.uint64), np.zeros((R, G), np.uint32)
        streams = []
        for g in range(G):
            for p in peers_of:
                s = bytearray()
                if rng.random() < 0.15:
                    s += junk[int(rng.integers(0, len(junk)))]
                x = rng.random()
                if x < 0.85:
                    term = 2 if rng.random() < 0.995 else 3                                        # (a higher term now and then: the leader steps down)
                    end = int(rng.integers(0, max(int(len_now[g]), 1)))
                    conflict = (int(rng.integers(1, 3)), int(rng.integers(0, max(end, 1)))) if rng.random() < 0.1 else None
                    s += _raft_reply(term, end, conflict)
                    rt[p, g], es[p, g], fl[p, g] = term, end, 1 | (2 if conflict else 0)
                    if conflict:
                        ct[p, g], cs[p, g] = conflict
                    y = rng.random()
                    if y < 0.05:
                        s += _raft_reply(2, 1)                                                       # a second reply: the next call's
                    elif y < 0.1:
                        s += _raft_reply(1 << 40, 7)[:int(rng.integers(1, 11))]                      # an incomplete tail
                    elif y < 0.13:
                        s += _frame(_varint(0) + _varint(9))                                         # malformed behind a delivered reply
                elif x < 0.9:
                    s += _raft_reply(2, (1 << 32) + 5)                                               # a slot beyond u32: located, not taken
                streams.append(bytes(s))
        buf, off, _, _, _ = _layout(torch, cuda, streams, [0] * len(streams), [0] * len(streams))
        o = ing_a.raft(buf, off, grp, peer)
        a.handle_msg_append_entries_reply(o["reply_term"], o["end_slot"], o["flags"], o["conflict_term"], o["conflict_slot"])
        ing_b.raft_into(b, buf, off)
        ra, rb = ing_a.results(), ing_b.results()
        for k in ("n_replies", "n_others", "n_malformed", "n_deferred"):
            assert ra[k] == rb[k], (t, k, ra[k], rb[k])
        assert np.array_equal(ra["consumed"], rb["consumed"]) and np.array_equal(ra["status"], rb["status"])
        key = lambda z: np.sort(z, order=["conn", "off"])
        assert np.array_equal(key(ra["others"]), key(rb["others"])), t
        da, db = a.dump(), b.dump()
        for n in da:
            assert np.array_equal(da[n], db[n]), (t, n)
        if orc is not None:
            taken = o["flags"].cpu().numpy()
            assert np.array_equal(taken != 0, fl != 0)
            orc.handle_replies(rt, es, np.ascontiguousarray(taken), ct, cs)
            do = orc.dump()
            for n in do:
                assert np.array_equal(db[n], do[n]), (t, n, "oracle")
        n_exec = int(db["last_commit"].sum())
    assert n_exec > 0
    return n_exec


def test_fused_raft_wire_replies_equal_the_two_calls(cuda, oracle):
    run_fused_raft_wire_replies(cuda, oracle)
    run_fused_raft_wire_replies(cuda, oracle, G=1100, R=3, me=0, seed=5, T=4)      # 128 groups per block, the leader's id in front
    run_fused_raft_wire_replies(cuda, oracle, G=700, R=7, W=32, me=6, seed=6, T=4)  # 42 groups per block (252 of 256 lanes), the 8-wide instance


def test_fused_raft_wire_replies_refuse_sparse_connections(cuda):
    import torch
    from summerset_amd import RaftLeaderGroup, SummersetError, wire
    ld = RaftLeaderGroup(64, 5, leader_id=0, window=32, term=1)
    ing = wire.ReplyIngest(64 * 4, 64, 5, 16, cuda)
    ing.n_conn = 100                                                                   # not n_groups * (population - 1)
    with pytest.raises((SummersetError, AssertionError)):
        ing.raft_into(ld, torch.zeros(16, dtype=torch.uint8, device=cuda), torch.zeros(101, dtype=torch.int64, device=cuda))

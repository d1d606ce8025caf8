"""Reply frames written on the device (`smr_wire_emit_mp_accept_replies`, `smr_wire_emit_raft_replies`,
`smr_wire_emit_ep_pre_accept_replies`, csrc/wire_emit.hip; SURVEY §8 f.1, the send half): every frame against the bytes the
TEST lays out from the reference's type definitions (`[u64 BE length]` safetcp.rs:127-132 + bincode-standard
`PeerMessage::Msg { msg: PeerMsg::X { .. } }`, variant indexes of multipaxos/mod.rs:298-384, raft/mod.rs:203-234,
epaxos/mod.rs:306-377, SURVEY Appendix C's varint rule) and against the host encoder; then the whole device wire loop --
emit -> the frames concatenated per connection -> the ingest kernels (csrc/wire_ingest*.hip) -> the arrays / records that
went in; and the Raft / EPaxos clusters of tests/test_zz_reply_ingest_gpu.py with the emit kernels as the senders."""
import struct

import numpy as np
import pytest

from test_zz_reply_ingest_gpu import NONE32, _ep_reply, _frame, _layout, _pick, _raft_reply, _varint

pytestmark = pytest.mark.gpu


def _frames(frames, ln):
    f, n = frames.cpu().numpy(), ln.cpu().numpy()
    return [bytes(f[i, :n[i]]) for i in range(len(n))]


def test_mp_accept_reply_frames(cuda):
    """every varint width of slot and ballot; the frames back through smr_wire_ingest_mp are the records"""
    import torch
    from summerset_amd import wire
    from summerset_amd.multipaxos import ACK_DTYPE
    rng = np.random.default_rng(21)
    n = 3000
    rec = np.zeros(n, ACK_DTYPE)
    rec["group"], rec["peer"] = rng.integers(0, 1 << 20, n), rng.integers(0, 5, n)
    rec["slot"] = [_pick(rng) for _ in range(n)]
    rec["ballot"] = [_pick(rng, True) for _ in range(n)]
    dev = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(cuda)
    fr = _frames(*wire.emit_mp_accept_replies(dev, n))
    for i in range(n):
        want = _frame(_varint(0) + _varint(3) + _varint(int(rec["slot"][i])) + _varint(int(rec["ballot"][i])) + b"\x00")
        assert fr[i] == want and fr[i] == wire.accept_reply(int(rec["slot"][i]), int(rec["ballot"][i])), i
    # one connection per record group of 7: the ingest kernel gives the records back, in order
    conns = [b"".join(fr[i:i + 7]) for i in range(0, n, 7)]
    groups, peers = rec["group"][::7], rec["peer"][::7]
    buf, off, grp, peer, _ = _layout(torch, cuda, conns, groups, peers)
    ing = wire.MpIngest(len(conns), n, 16, 16, device=cuda)
    ing.ingest(buf, off, grp, peer)
    got = ing.results()
    want = rec.copy()
    want["group"], want["peer"] = np.repeat(groups, 7)[:n], np.repeat(peers, 7)[:n]
    assert got["n_acks"] == n and got["n_malformed"] == 0 and np.array_equal(got["acks"], want)


def test_raft_reply_frames(cuda):
    import torch
    from summerset_amd import wire
    rng = np.random.default_rng(22)
    G, R = 2000, 5
    flags = rng.choice([0, 1, 3], G, p=[0.2, 0.5, 0.3]).astype(np.uint8)
    term = np.array([_pick(rng, True) for _ in range(G)], np.uint64)
    es = np.array([_pick(rng) for _ in range(G)], np.uint32)
    ct = np.array([_pick(rng, True) for _ in range(G)], np.uint64)
    cs = np.array([_pick(rng) for _ in range(G)], np.uint32)
    dv = lambda a: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else (a.view(np.int32) if a.dtype == np.uint32 else a)).to(cuda)   # noqa: E731
    frames_dev, len_dev = wire.emit_raft_replies(dv(flags), dv(term), dv(es), dv(ct), dv(cs))
    fr = _frames(frames_dev, len_dev)
    for g in range(G):
        if not flags[g] & 1:
            assert fr[g] == b""
            continue
        conflict = (int(ct[g]), int(cs[g])) if flags[g] & 2 else None
        assert fr[g] == _raft_reply(int(term[g]), int(es[g]), conflict) == wire.raft_append_entries_reply(int(term[g]), int(es[g]), conflict), g
    # as the traffic of follower 2: back through the ingest kernel into the leader's arrays
    buf, off, grp, peer, _ = _layout(torch, cuda, fr, np.arange(G), np.full(G, 2))
    ing = wire.ReplyIngest(G, G, R, 16, cuda)
    o = {k: v.cpu().numpy() for k, v in ing.raft(buf, off, grp, peer).items()}
    res = ing.results()
    assert res["n_replies"] == int((flags & 1).sum()) and res["n_malformed"] == 0 and res["n_others"] == 0
    m = (flags & 1) != 0
    assert np.array_equal(o["flags"][2], flags) and not o["flags"][[0, 1, 3, 4]].any()
    assert np.array_equal(o["reply_term"][2].view(np.uint64)[m], term[m]) and np.array_equal(o["end_slot"][2].view(np.uint32)[m], es[m])
    mc = (flags & 2) != 0
    assert np.array_equal(o["conflict_term"][2].view(np.uint64)[mc], ct[mc]) and np.array_equal(o["conflict_slot"][2].view(np.uint32)[mc], cs[mc])
    # ... and without leaving the device: the emit call's slots ARE the connections (conn_off = slot starts, conn_len = len)
    ing2 = wire.ReplyIngest(G, G, R, 16, cuda)
    slot_off = torch.arange(G, dtype=torch.int64, device=frames_dev.device) * wire.EMIT_RAFT_STRIDE
    o2 = {k: v.cpu().numpy() for k, v in ing2.raft(frames_dev.view(-1), slot_off, grp, peer, conn_len=len_dev).items()}
    assert ing2.results()["n_replies"] == int(m.sum()) and np.array_equal(ing2.results()["consumed"], len_dev.cpu().numpy())
    for k in o:
        assert np.array_equal(np.where(o["flags"] != 0, o[k], 0) if k != "flags" and "conflict" not in k else np.where((o["flags"] & (2 if "conflict" in k else 1)) != 0, o[k], 0),
                              np.where(o2["flags"] != 0, o2[k], 0) if k != "flags" and "conflict" not in k else np.where((o2["flags"] & (2 if "conflict" in k else 1)) != 0, o2[k], 0)), k


@pytest.mark.parametrize("R", [3, 5, 7])
def test_ep_pre_accept_reply_frames(cuda, R):
    import torch
    from summerset_amd import wire
    rng = np.random.default_rng(23 + R)
    G, row, q = 1500, 1, 2
    flags = (rng.random(G) < 0.8).astype(np.uint8)
    col = np.array([min(_pick(rng), NONE32 - 1) for _ in range(G)], np.uint32)
    ballot = np.array([_pick(rng, True) for _ in range(G)], np.uint64)
    seq = np.array([_pick(rng, True) for _ in range(G)], np.uint64)
    deps = np.array([[NONE32 if rng.random() < 0.3 else min(_pick(rng), NONE32 - 1) for _ in range(G)] for _ in range(R)], np.uint32)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if a.dtype == np.uint64 else (np.int32 if a.dtype == np.uint32 else a.dtype))).to(cuda)   # noqa: E731
    frames_dev, len_dev = wire.emit_ep_pre_accept_replies(dv(flags), row, dv(col), dv(ballot), dv(seq), dv(deps))
    fr = _frames(frames_dev, len_dev)
    for g in range(G):
        if not flags[g]:
            assert fr[g] == b""
            continue
        d = [None if int(x) == NONE32 else int(x) for x in deps[:, g]]
        assert fr[g] == _ep_reply(row, int(col[g]), int(ballot[g]), int(seq[g]), d), g
        assert fr[g] == wire.ep_msg(wire.EP_PRE_ACCEPT_REPLY, row, int(col[g]), int(ballot[g]), int(seq[g]), d), g
    buf, off, grp, peer, _ = _layout(torch, cuda, fr, np.arange(G), np.full(G, q))
    ing = wire.ReplyIngest(G, G, R, 16, cuda)
    o = {k: v.cpu().numpy() for k, v in ing.ep_pre_accept(buf, off, grp, peer, row, dv(col)).items()}
    res = ing.results()
    m = flags != 0
    assert res["n_replies"] == int(m.sum()) and res["n_malformed"] == 0 and res["n_others"] == 0
    assert np.array_equal(o["flags"][q], flags)
    assert np.array_equal(o["ballot"][q].view(np.uint64)[m], ballot[m]) and np.array_equal(o["seq"][q].view(np.uint64)[m], seq[m])
    assert np.array_equal(o["deps"][q].view(np.uint32)[:, m], deps[:, m])
    # the emit call's slots as the connections: no host in the loop
    ing2 = wire.ReplyIngest(G, G, R, 16, cuda)
    slot_off = torch.arange(G, dtype=torch.int64, device=frames_dev.device) * wire.EMIT_EP_STRIDE
    o2 = {k: v.cpu().numpy() for k, v in ing2.ep_pre_accept(frames_dev.view(-1), slot_off, grp, peer, row, dv(col), conn_len=len_dev).items()}
    assert ing2.results()["n_replies"] == int(m.sum()) and np.array_equal(o2["flags"], o["flags"])
    assert np.array_equal(o2["ballot"][q][m], o["ballot"][q][m]) and np.array_equal(o2["seq"][q][m], o["seq"][q][m]) and np.array_equal(o2["deps"][q][:, m], o["deps"][q][:, m])


def test_rsp_accept_replies_ingest(cuda):
    """RSPaxos AcceptReply { slot, ballot } (rspaxos/mod.rs:262-305, variant 3) frames laid out here -> the [R][G] arrays
    smr_rsp_handle_accept_replies takes; a Heartbeat-sized frame and a second reply around them"""
    import torch
    from summerset_amd import wire
    rng = np.random.default_rng(24)
    G, R = 700, 5
    streams, groups, peers = [], [], []
    ws, wb, wf = np.zeros((R, G), np.uint32), np.zeros((R, G), np.uint64), np.zeros((R, G), np.uint8)
    n_loc = n_def = 0
    for g in range(G):
        for p in range(1, R):
            s = bytearray()
            if rng.random() < 0.2:
                s += _frame(_varint(0) + _varint(6) + _varint(9) + _varint(5) + _varint(4) + _varint(0)); n_loc += 1   # Heartbeat
            if rng.random() < 0.85:
                slot, ballot = _pick(rng), _pick(rng, True)
                s += _frame(_varint(0) + _varint(3) + _varint(slot) + _varint(ballot))
                ws[p, g], wb[p, g], wf[p, g] = slot, ballot, 1
                assert wire.rsp_accept_reply(slot, ballot) == _frame(_varint(0) + _varint(3) + _varint(slot) + _varint(ballot))
                if rng.random() < 0.1:
                    s += _frame(_varint(0) + _varint(3) + _varint(1) + _varint(2)); n_def += 1
            streams.append(bytes(s)); groups.append(g); peers.append(p)
    buf, off, grp, peer, _ = _layout(torch, cuda, streams, groups, peers)
    ing = wire.ReplyIngest(len(streams), G, R, 4096, cuda)
    o = {k: v.cpu().numpy() for k, v in ing.rsp_accept(buf, off, grp, peer).items()}
    res = ing.results()
    m = wf != 0
    assert np.array_equal(o["flags"], wf) and np.array_equal(o["slot"].view(np.uint32)[m], ws[m]) and np.array_equal(o["ballot"].view(np.uint64)[m], wb[m])
    assert res["n_replies"] == int(m.sum()) and res["n_others"] == n_loc and res["n_deferred"] == n_def and res["n_malformed"] == 0


def test_raft_cluster_with_emitted_replies(cuda, oracle):
    """tests/raft_cluster.py's closed loop with every AppendEntriesReply written by the emit kernel and parsed by the ingest
    kernel (nothing of a reply goes through a host codec); the oracle cluster is wired directly"""
    import torch
    import raft_cluster as rc
    from summerset_amd import RaftLeaderGroup, wire
    G, W, K, R = 300, 64, 8, 5
    engs = [rc.NumpyRaft(RaftLeaderGroup(G, R, leader_id=r, window=W, term=1), cuda) for r in range(R)]
    orcs = [oracle.RaftOracle(G, R, W, leader_id=r, term=1) for r in range(R)]
    for x in engs + orcs:
        x.preset(rc.FOLLOWER, 0xFF, 0)
    ing = wire.ReplyIngest(G * (R - 1), G, R, 16, cuda)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if a.dtype == np.uint64 else (np.int32 if a.dtype == np.uint32 else a.dtype))).to(cuda)   # noqa: E731

    def via(s, rt, es, fl, ct, cs):
        streams, groups, peers = [], [], []
        for q in range(R):
            if q == s:
                continue
            fr = _frames(*wire.emit_raft_replies(dv(fl[q]), dv(rt[q]), dv(es[q]), dv(ct[q]), dv(cs[q])))
            streams += fr; groups += list(range(G)); peers += [q] * G
        buf, off, grp, peer, _ = _layout(torch, cuda, streams, groups, peers)
        o = {k: v.cpu().numpy() for k, v in ing.raft(buf, off, grp, peer).items()}
        assert ing.results()["n_replies"] == int((fl & 1).sum())
        pres, conf = o["flags"] != 0, (o["flags"] & 2) != 0
        return (np.where(pres, o["reply_term"].view(np.uint64), 0).astype(np.uint64), np.where(pres, o["end_slot"].view(np.uint32), 0).astype(np.uint32),
                o["flags"].copy(), np.where(conf, o["conflict_term"].view(np.uint64), 0).astype(np.uint64),
                np.where(conf, o["conflict_slot"].view(np.uint32), 0).astype(np.uint32))

    rng = np.random.default_rng(9)
    none = np.full((R, G), 0xFF, np.uint8)
    to = none.copy()
    to[np.arange(G) % R, np.arange(G)] = 0xFE
    rc.tick(engs, to, np.zeros((R, G), np.uint32), K, via=via)
    rc.tick(orcs, to, np.zeros((R, G), np.uint32), K)
    for t in range(10):
        n_new = rng.integers(0, 4, (R, G)).astype(np.uint32)
        to = none.copy()
        if t == 4:
            gs = np.arange(0, G, 3)
            to[(gs + 2) % R, gs] = (gs % R).astype(np.uint8)
        rc.tick(engs, to, n_new, K, via=via)
        rc.tick(orcs, to, n_new, K)
        for r in range(R):
            a, b = engs[r].dump(), orcs[r].dump()
            for n in b:
                assert np.array_equal(a[n], b[n]), (t, r, n)


def test_ep_cluster_with_emitted_replies(cuda, oracle):
    """tests/ep_cluster.py's closed loop with every PreAcceptReply written by the emit kernel and parsed by the ingest kernel"""
    import torch
    import ep_cluster as ec
    from summerset_amd import EPaxosReplicaGroup, wire
    G, R, W, K, T = 260, 5, 32, 6, 6
    engs = [ec.NumpyEngine(EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K), cuda) for r in range(R)]
    orcs = [oracle.EpOracle(G, R, me=r, W=W, n_keys=K) for r in range(R)]
    ing = wire.ReplyIngest(G * (R - 1), G, R, 16, cuda)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if a.dtype == np.uint64 else (np.int32 if a.dtype == np.uint32 else a.dtype))).to(cuda)   # noqa: E731

    def via(s, col, ballot, seq, deps, flags):
        streams, groups, peers = [], [], []
        for q in range(R):
            if q == s:
                continue
            streams += _frames(*wire.emit_ep_pre_accept_replies(dv(flags[q] & 1), s, dv(col), dv(ballot[q]), dv(seq[q]), dv(deps[q])))
            groups += list(range(G)); peers += [q] * G
        buf, off, grp, peer, _ = _layout(torch, cuda, streams, groups, peers)
        o = {k: v.cpu().numpy() for k, v in ing.ep_pre_accept(buf, off, grp, peer, s, dv(col)).items()}
        assert ing.results()["n_replies"] == int((flags & 1).sum())
        pres = o["flags"] != 0
        return (np.where(pres, o["ballot"].view(np.uint64), 0).astype(np.uint64), np.where(pres, o["seq"].view(np.uint64), 0).astype(np.uint64),
                np.where(pres[:, None, :], o["deps"].view(np.uint32), NONE32).astype(np.uint32), o["flags"].copy())

    rng = np.random.default_rng(6)
    for t in range(T):
        keys = ec.zipf_keys(rng, R, G, K)
        drop = {(s, q): rng.random(G) < 0.15 for s in range(R) for q in range(R) if s != q}
        oe, oo = ec.tick(engs, keys, drop, via=via), ec.tick(orcs, keys, drop)
        for s in range(R):
            for k in oo[s]:
                assert np.array_equal(oe[s][k], oo[s][k]), (t, s, k)
    for r in range(R):
        a, b = engs[r].dump(), orcs[r].dump()
        for n in b:
            assert np.array_equal(a[n], b[n]), (r, n)

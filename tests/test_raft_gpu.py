"""Raft leader-side parity (SURVEY §8d config 3 stream): HIP kernel vs the literal
CPU restatement on identical synthetic AppendEntriesReply matrices, bit-exact."""
import numpy as np
import pytest

from summerset_amd import stream

pytestmark = pytest.mark.gpu


def _replies(seed, t, G, R, log_len, curr_term, lag_max=3, drop_p=0.05, stale_p=0.005, conflict_p=0.005,
             higher_p=0.0):
    """per (peer, group): end_slot = leader_last - lag, some dropped / stale-term / conflict replies"""
    p = np.arange(R, dtype=np.uint64)[:, None]
    g = np.arange(G, dtype=np.uint64)[None, :]
    u = lambda tag: (stream._key(seed, tag, t, p, g) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    lag = (stream._key(seed, 1, t, p, g) % np.uint64(lag_max + 1)).astype(np.int64)
    last = log_len.astype(np.int64)[None, :] - 1
    end_slot = np.maximum(last - lag, 0).astype(np.uint32)
    flags = (u(2) >= drop_p).astype(np.uint8)
    term = np.broadcast_to(curr_term[None, :], (R, G)).astype(np.uint64).copy()
    stale = u(3) < stale_p
    term[stale] = 0                                                    # stale (smaller) term: still processed
    higher = u(6) < higher_p
    term[higher] += 1                                                  # a peer moved on: leader steps down
    conflict = u(4) < conflict_p
    flags = (flags | (conflict.astype(np.uint8) << 1)).astype(np.uint8)
    cterm = np.where(conflict, curr_term[None, :], 0).astype(np.uint64)
    cslot = np.where(conflict, np.maximum(end_slot.astype(np.int64) - 2, 1), 0).astype(np.uint32)
    order = stream.random_ackctl(seed, t, 1, G, R, 0.0)[0]
    return term, end_slot, flags, cterm, cslot, np.ascontiguousarray(order)


def _run(cuda, oracle, G, R, W, T, commit_extra=0, higher_p=0.0, n_new_max=3, seed=77):
    import torch
    from summerset_amd import RaftLeaderGroup
    eng = RaftLeaderGroup(G, R, 0, W, term=1, commit_extra=commit_extra)
    orc = oracle.RaftOracle(G, R, W, 0, 1, commit_extra)
    dev = lambda a: torch.from_numpy(a).to(cuda)
    for t in range(T):
        n_new = (stream._key(seed, 9, t, np.arange(G, dtype=np.uint64)) % np.uint64(n_new_max + 1)).astype(np.uint32)
        orc.append(n_new)
        eng.handle_req_batch(dev(n_new))
        d = orc.dump()
        term, es, fl, ct, cs, order = _replies(seed, t, G, R, d["log_len"], d["curr_term"], higher_p=higher_p)
        orc.handle_replies(term, es, fl, ct, cs, order)
        eng.handle_msg_append_entries_reply(dev(term), dev(es), dev(fl), dev(ct), dev(cs), dev(order))
        a, b = eng.dump(), orc.dump()
        for k in b:
            assert np.array_equal(a[k], b[k]), "tick %d field %s" % (t, k)
        assert eng.total_commits() == orc.total_commits()
    assert orc.total_commits() > 0
    return eng, orc


def _run_batched(cuda, oracle, G, R, W, T, batches, higher_p=0.0, n_new_max=3, seed=78, quiet=()):
    """`smr_raft_leader_run_ticks` (one launch per <= 16 ticks, the state in registers from tick to tick) against the oracle's
    tick-by-tick run of the same inputs: state at every batch boundary.  `quiet`: ticks that pass NULL for their appends /
    replies (and give the oracle none)"""
    import torch
    from summerset_amd import RaftLeaderGroup, SummersetError
    eng = RaftLeaderGroup(G, R, 0, W, term=1)
    orc = oracle.RaftOracle(G, R, W, 0, 1, 0)
    dev = lambda a: torch.from_numpy(a).to(cuda)
    ticks, snaps = [], []
    for t in range(T):
        n_new = (stream._key(seed, 9, t, np.arange(G, dtype=np.uint64)) % np.uint64(n_new_max + 1)).astype(np.uint32)
        x = {}
        if ("append", t) not in quiet:
            orc.append(n_new)
            x["n_new"] = dev(n_new)
        d = orc.dump()
        term, es, fl, ct, cs, order = _replies(seed, t, G, R, d["log_len"], d["curr_term"], higher_p=higher_p)
        if ("replies", t) not in quiet:
            orc.handle_replies(term, es, fl, ct, cs, order)
            x.update(reply_term=dev(term), end_slot=dev(es), flags=dev(fl), conflict_term=dev(ct), conflict_slot=dev(cs), order=dev(order))
        ticks.append(x)
        snaps.append((orc.dump(), orc.total_commits()))
    t0 = 0
    for k in batches:
        eng.run_ticks(ticks[t0:t0 + k])
        t0 += k
        a, (b, commits) = eng.dump(), snaps[t0 - 1]
        for n in b:
            assert np.array_equal(a[n], b[n]), "after tick %d field %s" % (t0 - 1, n)
        assert eng.total_commits() == commits
    assert t0 == T and orc.total_commits() > 0
    with pytest.raises(SummersetError):
        eng.run_ticks([dict(flags=ticks[0]["flags"])])                   # replies without their terms / end slots
    return eng, orc


def test_raft_batched_ticks_match_the_oracle(cuda, oracle):
    _run_batched(cuda, oracle, G=1000, R=5, W=64, T=47, batches=(1, 5, 16, 20, 2, 3))
    eng, orc = _run_batched(cuda, oracle, G=500, R=3, W=32, T=40, batches=(7, 33), higher_p=0.002)
    assert (orc.dump()["role"] == 0).any()                             # leaders that step down in the middle of a batch
    _run_batched(cuda, oracle, G=300, R=7, W=16, T=30, batches=(30,), n_new_max=9)                  # ring back-pressure, the 8-wide instance
    _run_batched(cuda, oracle, G=300, R=5, W=64, T=12, batches=(12,), quiet={("append", 3), ("replies", 5), ("append", 11), ("replies", 11)})


def test_raft_batched_ticks_config2_size(cuda, oracle):
    _run_batched(cuda, oracle, G=65536, R=5, W=64, T=16, batches=(16,))


def test_craft_leader_refuses_batches(cuda):
    from summerset_amd import CRaftLeaderGroup, SummersetError
    eng = CRaftLeaderGroup(64, 5, 0, 64, term=1)
    with pytest.raises(SummersetError):
        eng.run_ticks([{}])


def test_raft_steady(cuda, oracle):
    _run(cuda, oracle, G=1000, R=5, W=64, T=60)


def test_raft_three_replicas_and_stepdown(cuda, oracle):
    eng, orc = _run(cuda, oracle, G=500, R=3, W=32, T=40, higher_p=0.002)
    assert (orc.dump()["role"] == 0).any()                             # some leaders stepped down (check_term)


def test_craft_threshold(cuda, oracle):
    _run(cuda, oracle, G=300, R=5, W=64, T=40, commit_extra=1)


def test_raft_config3_65536_groups(cuda, oracle):
    _run(cuda, oracle, G=65536, R=5, W=64, T=12)


# ---- follower side and elections (raft/messages.rs:13-218, 391-510; leadership.rs:76-218) ----
def _t(a, cuda):
    import torch
    return torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else (a.view(np.int32) if a.dtype == np.uint32 else a)).to(cuda)


def _same_state(eng, orc, step):
    a, b = eng.dump(), orc.dump()
    for n in a:
        assert np.array_equal(a[n], b[n]), (step, n, np.nonzero(a[n] != b[n]))
    va, vb = eng.dump_votes(), orc.dump_votes()
    for n in va:
        assert np.array_equal(va[n].astype(np.uint64), vb[n].astype(np.uint64)), (step, n)


def _same_reply(r_eng, r_orc, step):
    for k, v in r_orc.items():
        e = r_eng[k].cpu().numpy()
        assert e.dtype.itemsize == v.dtype.itemsize, k
        assert np.array_equal(e.view(v.dtype), v), (step, k, np.nonzero(e.view(v.dtype) != v))


@pytest.mark.parametrize("G,W", [(777, 64), (4096, 32)])
def test_follower_and_elections_match_oracle(cuda, oracle, G, W):
    from summerset_amd import RaftLeaderGroup
    import raft_scenarios as sc                       # tests/ is on sys.path under pytest
    R, me, K = 5, 2, 6
    rng = np.random.default_rng(G + W)
    eng = RaftLeaderGroup(G, R, leader_id=me, window=W, term=1)
    orc = oracle.RaftOracle(G, R, W, leader_id=me, term=1)
    # a few leader appends give every group a log, then everybody is a follower of replica 0
    for _ in range(3):
        n_new = rng.integers(0, 4, G).astype(np.uint32)
        eng.handle_req_batch(_t(n_new, cuda))
        orc.append(n_new)
    eng.preset(0, 0, 1)
    orc.preset(0, 0, 1)
    _same_state(eng, orc, "preset")
    from summerset_amd import stream
    for step in range(40):
        d = orc.dump()
        kind = step % 5
        if kind in (0, 1, 2):
            m = sc.append_entries_round(rng, d, G, K, me, W)
            ro = orc.handle_append_entries(**m)
            re_ = eng.handle_msg_append_entries(**{k: _t(v, cuda) for k, v in m.items()})
            _same_reply(re_, ro, step)
        elif kind == 3:
            src = sc.timeout_round(rng, d, G, me)
            ro = orc.become_candidate(src)
            re_ = eng.become_a_candidate(_t(src, cuda))
            _same_reply(re_, ro, step)
            m = sc.request_vote_round(rng, orc.dump(), G, me, W)
            ro = orc.handle_request_vote(**m)
            re_ = eng.handle_msg_request_vote(**{k: _t(v, cuda) for k, v in m.items()})
            _same_reply(re_, ro, step)
        else:
            ctl = stream.random_ackctl(7, step, 1, G, R, 0.0)[0]
            m = sc.vote_reply_round(rng, d, G, R, me, np.ascontiguousarray(ctl))
            ro = orc.handle_vote_replies(m["term"], m["flags"], m["order"])
            re_ = eng.handle_msg_request_vote_reply(_t(m["term"], cuda), _t(m["flags"], cuda), _t(m["order"], cuda))
            _same_reply(re_, ro, step)
            # the elected ones lead for a moment: appends and replies go through the leader kernels
            n_new = rng.integers(0, 3, G).astype(np.uint32)
            eng.handle_req_batch(_t(n_new, cuda))
            orc.append(n_new)
        _same_state(eng, orc, step)
    roles = orc.dump()["role"]
    assert (roles == 0).any() and (roles == 1).any() and (roles == 2).any()      # every role was reached
    v = orc.dump_votes()
    assert v["n_trunc"].sum() > 0 and v["n_exec"].sum() > 0
    # engine and oracle share ONE deviation from raft/messages.rs:128-140 -- an entry that has left the W-entry term
    # ring is taken as matching -- so the comparison above says nothing about such entries.  Both sides count them: the
    # counts agree, and at W = 64 (the window is as long as this scenario's logs get) no decision of this run rests on the rule
    hits = eng.ring_guard_hits()
    assert hits == orc.ring_guard_hits(), (hits, orc.ring_guard_hits())
    if W >= 64:
        assert hits == 0, "the follower parity run reached the ring guard %d times" % hits


def test_closed_loop_cluster_matches_oracle(cuda, oracle):
    """five per-replica engine objects wired into a Raft cluster (tests/raft_cluster.py: elections, the
    AppendEntries the leaders' appends produce, replies, match-index quorum) against five oracles wired the same way"""
    import raft_cluster as rc
    from summerset_amd import RaftLeaderGroup
    G, W, K, R = 600, 64, 8, 5
    engs = [rc.NumpyRaft(RaftLeaderGroup(G, R, leader_id=r, window=W, term=1), cuda) for r in range(R)]
    orcs = [oracle.RaftOracle(G, R, W, leader_id=r, term=1) for r in range(R)]
    for x in engs + orcs:
        x.preset(rc.FOLLOWER, 0xFF, 0)
    rng = np.random.default_rng(9)
    none = np.full((R, G), 0xFF, np.uint8)

    def same(step):
        for r in range(R):
            a, b = engs[r].dump(), orcs[r].dump()
            for n in b:
                assert np.array_equal(a[n], b[n]), (step, r, n)
            va, vb = engs[r].dump_votes(), orcs[r].dump_votes()
            for n in vb:
                assert np.array_equal(va[n].astype(np.uint64), vb[n].astype(np.uint64)), (step, r, n)

    to = none.copy()
    to[np.arange(G) % R, np.arange(G)] = 0xFE               # first election: replica g % 5 of each group
    for reps in (engs, orcs):
        rc.tick(reps, to, np.zeros((R, G), np.uint32), K)
    same("election")
    for t in range(16):
        n_new = rng.integers(0, 4, (R, G)).astype(np.uint32)
        to = none.copy()
        if t == 7:                                          # a second election in a third of the groups
            gs = np.arange(0, G, 3)
            to[(gs + 2) % R, gs] = (gs % R).astype(np.uint8)
        for reps in (engs, orcs):
            rc.tick(reps, to, n_new, K)
        same(t)
    d = [o.dump() for o in orcs]
    assert min(int(np.stack([x["last_commit"] for x in d]).max(axis=0).min()), 99) > 5
    assert (np.stack([x["curr_term"] for x in d]).max(axis=0) == 2).any()


def run_one_launch_replication(dev, oracle, G=600, W=64, K=8, T=16):
    """`smr_raft_cluster_replicate` -- a sender's AppendEntries for its four peers and their handlers in one launch -- in the closed
    loop of tests/raft_cluster.py (elections, a second election in a third of the groups, conflicts and truncations behind it),
    the replication step sender by sender: a cluster that replicates with the 2 n calls, one that replicates with the one launch
    and five oracles stay identical -- every message, every reply, every replica's state, tick by tick"""
    import raft_cluster as rc
    from summerset_amd import RaftLeaderGroup
    R = 5
    mk = lambda: [rc.NumpyRaft(RaftLeaderGroup(G, R, leader_id=r, window=W, term=1), dev) for r in range(R)]
    calls, fused = mk(), mk()
    orcs = [oracle.RaftOracle(G, R, W, leader_id=r, term=1) for r in range(R)]
    for x in calls + fused + orcs:
        x.preset(rc.FOLLOWER, 0xFF, 0)
    rng = np.random.default_rng(9)
    none = np.full((R, G), 0xFF, np.uint8)
    n_msg = 0

    def step(to, n_new, where):
        nonlocal n_msg
        seen = ([], [], [])
        for reps, one, sn in ((calls, False, seen[0]), (fused, True, seen[1]), (orcs, False, seen[2])):
            rc.tick(reps, to, n_new, K, sender_major=True, one_launch=one, seen=sn)
        for (s, q, m0, r0), (_, _, m1, r1), (_, _, m2, r2) in zip(*seen):
            on = m2["flags"] != 0                                         # (a message's other fields mean something where one is sent)
            n_msg += int(on.sum())
            for k in m2:
                assert np.array_equal(m0[k], m1[k]), (where, "message", s, q, k)
                sel = (slice(None), on) if k == "entry_term" else on
                assert np.array_equal(np.asarray(m1[k])[sel].astype(np.uint64), np.asarray(m2[k])[sel].astype(np.uint64)), (where, "message vs oracle", s, q, k)
            for k in r2:
                assert np.array_equal(r0[k], r1[k]) and np.array_equal(r1[k].astype(np.uint64), r2[k].astype(np.uint64)), (where, "reply", s, q, k)
        for r in range(R):
            a, b, c = calls[r].dump(), fused[r].dump(), orcs[r].dump()
            for n in c:
                assert np.array_equal(a[n], b[n]) and np.array_equal(b[n], c[n]), (where, r, n)

    to = none.copy()
    to[np.arange(G) % R, np.arange(G)] = 0xFE
    step(to, np.zeros((R, G), np.uint32), "election")
    for t in range(T):
        n_new = rng.integers(0, 4, (R, G)).astype(np.uint32)
        to = none.copy()
        if t == 7:
            gs = np.arange(0, G, 3)
            to[(gs + 2) % R, gs] = (gs % R).astype(np.uint8)
        step(to, n_new, t)
    d = [o.dump() for o in orcs]
    assert min(int(np.stack([x["last_commit"] for x in d]).max(axis=0).min()), 99) > 5
    assert (np.stack([x["curr_term"] for x in d]).max(axis=0) == 2).any() and n_msg > 1000
    return n_msg


def run_one_launch_tick(dev, oracle, G=600, W=64, K=8, T=16, R=5):
    """`smr_raft_cluster_tick` -- a sender's append, its AppendEntries for its four peers, their handlers and its reply handler in ONE
    launch -- in the closed loop of tests/raft_cluster.py (elections, a second election in a third of the groups, conflicts and
    truncations behind it), every sender's whole tick before the next sender's: a cluster that runs the 2 + 2 n calls, one that runs
    the one launch and five oracles stay identical -- every message, every reply, every replica's state, tick by tick"""
    import raft_cluster as rc
    from summerset_amd import RaftLeaderGroup
    mk = lambda: [rc.NumpyRaft(RaftLeaderGroup(G, R, leader_id=r, window=W, term=1), dev) for r in range(R)]
    calls, fused = mk(), mk()
    orcs = [oracle.RaftOracle(G, R, W, leader_id=r, term=1) for r in range(R)]
    for x in calls + fused + orcs:
        x.preset(rc.FOLLOWER, 0xFF, 0)
    rng = np.random.default_rng(19)
    none = np.full((R, G), 0xFF, np.uint8)
    n_msg = 0

    def step(to, n_new, where):
        nonlocal n_msg
        seen = ([], [], [])
        for reps, one, sn in ((calls, False, seen[0]), (fused, "tick", seen[1]), (orcs, False, seen[2])):
            rc.tick(reps, to, n_new, K, sender_ticks=True, one_launch=one, seen=sn)
        for (s, q, m0, r0), (_, _, m1, r1), (_, _, m2, r2) in zip(*seen):
            on = m2["flags"] != 0
            n_msg += int(on.sum())
            for k in m2:
                assert np.array_equal(m0[k], m1[k]), (where, "message", s, q, k)
                sel = (slice(None), on) if k == "entry_term" else on
                assert np.array_equal(np.asarray(m1[k])[sel].astype(np.uint64), np.asarray(m2[k])[sel].astype(np.uint64)), (where, "message vs oracle", s, q, k)
            for k in r2:
                assert np.array_equal(r0[k], r1[k]) and np.array_equal(r1[k].astype(np.uint64), r2[k].astype(np.uint64)), (where, "reply", s, q, k)
        for r in range(R):
            a, b, c = calls[r].dump(), fused[r].dump(), orcs[r].dump()
            for n in c:
                assert np.array_equal(a[n], b[n]) and np.array_equal(b[n], c[n]), (where, r, n)

    to = none.copy()
    to[np.arange(G) % R, np.arange(G)] = 0xFE
    step(to, np.zeros((R, G), np.uint32), "election")
    for t in range(T):
        n_new = rng.integers(0, 4, (R, G)).astype(np.uint32)
        to = none.copy()
        if t == 7:
            gs = np.arange(0, G, 3)
            to[(gs + 2) % R, gs] = (gs % R).astype(np.uint8)
        step(to, n_new, t)
    d = [o.dump() for o in orcs]
    assert min(int(np.stack([x["last_commit"] for x in d]).max(axis=0).min()), 99) > (5 if T >= 12 else 1)
    assert (np.stack([x["curr_term"] for x in d]).max(axis=0) == 2).any() and n_msg > (1000 if G >= 200 else 100)
    return n_msg


def test_one_launch_tick_is_the_calls_and_the_oracle_s(cuda, oracle):
    run_one_launch_tick(cuda, oracle)


def test_one_launch_tick_of_seven_replicas(cuda, oracle):
    """the same with seven replicas: the kernel's other instantiation (R > 5), seven wavefronts per block"""
    run_one_launch_tick(cuda, oracle, G=200, T=10, R=7)


def test_one_launch_replication_is_the_2n_calls_and_the_oracle_s(cuda, oracle):
    run_one_launch_replication(cuda, oracle)

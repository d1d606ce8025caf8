"""A closed-loop Raft cluster built from five per-replica oracles (tests/raft_cluster.py): Raft's own
safety properties must hold on the restatement -- at most one leader per term, logs agree on every
entry up to the smallest commit index (and what is committed stays), commit indices only cover
entries a majority holds."""
import numpy as np

import raft_cluster as rc


def _setup(oracle, G, W):
    reps = [oracle.RaftOracle(G, 5, W, leader_id=r, term=1) for r in range(5)]
    for r in reps:
        r.preset(rc.FOLLOWER, 0xFF, 0)                     # nobody leads, term 0, only the dummy entry
    return reps


def _invariants(reps, G, W, prev_commit):
    d = [r.dump() for r in reps]
    roles = np.stack([x["role"] for x in d]); terms = np.stack([x["curr_term"] for x in d])
    for g in range(G):
        lead = [(int(terms[r, g]), r) for r in range(5) if roles[r, g] == rc.LEADER]
        assert len({t for t, _ in lead}) == len(lead), ("two leaders in one term", g, lead)
    commit = np.stack([x["last_commit"] for x in d]); length = np.stack([x["log_len"] for x in d])
    assert (commit < length).all()
    lo = commit.min(axis=0)
    for g in range(G):
        for s in range(1, int(lo[g]) + 1):
            if s + W < int(length[:, g].max()):
                continue                                   # older than the rings still hold
            t = {int(d[r]["entry_term"][s % W, g]) for r in range(5)}
            assert len(t) == 1, ("committed entry differs", g, s, t)
    assert (commit.max(axis=0) >= prev_commit).all()
    # a commit index is backed by a majority of logs reaching it
    for g in range(G):
        c = int(commit[:, g].max())
        assert int((length[:, g] > c).sum()) >= 3, (g, c, length[:, g])
    return commit.max(axis=0)


def test_election_then_replication(oracle):
    G, W, K = 64, 64, 8
    reps = _setup(oracle, G, W)
    rng = np.random.default_rng(3)
    none = np.full((5, G), 0xFF, np.uint8)
    zero = np.zeros((5, G), np.uint32)
    prev = np.zeros(G, np.uint32)
    # tick 0: replica g % 5 of each group times out (about "nobody": leader is None) and gets elected
    to = none.copy()
    for g in range(G):
        to[g % 5, g] = 0xFE                                # any source: leader is None, so :80-85 lets it run
    rc.tick(reps, to, zero, K)
    d = [r.dump() for r in reps]
    for g in range(G):
        assert [int(d[r]["role"][g]) for r in range(5)].count(rc.LEADER) == 1 and int(d[g % 5]["role"][g]) == rc.LEADER
        assert all(int(d[r]["curr_term"][g]) == 1 for r in range(5))
    prev = _invariants(reps, G, W, prev)
    # steady replication: the leader of each group gets 0..3 batches per tick, everybody else gets some too (redirected)
    for t in range(1, 14):
        n_new = rng.integers(0, 4, (5, G)).astype(np.uint32)
        rc.tick(reps, none, n_new, K)
        prev = _invariants(reps, G, W, prev)
    assert prev.min() > 5                                  # every group made progress
    # a second election: in half of the groups a follower times out on the leader and takes over
    to = none.copy()
    for g in range(0, G, 2):
        to[(g + 1) % 5, g] = g % 5                         # the timer of replica g+1 about the current leader g
    rc.tick(reps, to, zero, K)
    d = [r.dump() for r in reps]
    for g in range(0, G, 2):
        assert int(d[(g + 1) % 5]["role"][g]) == rc.LEADER and int(d[(g + 1) % 5]["curr_term"][g]) == 2
    prev = _invariants(reps, G, W, prev)
    # the deposed leader steps down when it hears the new term; replication continues under the new leaders
    for t in range(8):
        n_new = rng.integers(0, 4, (5, G)).astype(np.uint32)
        rc.tick(reps, none, n_new, K)
        prev = _invariants(reps, G, W, prev)
    d = [r.dump() for r in reps]
    for g in range(G):
        assert [int(d[r]["role"][g]) for r in range(5)].count(rc.LEADER) == 1, g

"""EPaxos explicit prepare in the CPU oracle against hand-derived traces of the reference code
(src/protocols/epaxos/heartbeat.rs:17-125, messages.rs:511-821, dependency.rs:249-327), R = 5: simple quorum 3."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import ep_cluster as ec  # noqa: E402

N = 0xFFFFFFFF
NULL, PREACC, ACC, COMMITTED, EXECUTED = 0, 1, 2, 3, 5
NO = 0xFF


def _a(v, t):
    return np.array([v], t)


def _deps(*pairs, R=5):
    d = np.full((R, 1), N, np.uint32)
    for r, c in pairs:
        d[r, 0] = c
    return d


class Replies:
    """the [R][1] arrays handle_exp_prepare_replies takes"""

    def __init__(self, nb, R=5):
        self.nb = np.full((R, 1), nb, np.uint64)
        self.vb = np.zeros((R, 1), np.uint64); self.vs = np.zeros((R, 1), np.uint8); self.vq = np.zeros((R, 1), np.uint64)
        self.vd = np.full((R, R, 1), N, np.uint32); self.vk = np.full((R, 1), NO, np.uint8); self.fl = np.zeros((R, 1), np.uint8)

    def add(self, p, voted_bal, status, seq=0, deps=(), key=NO):
        self.fl[p] = 1; self.vb[p] = voted_bal; self.vs[p] = status; self.vq[p] = seq; self.vk[p] = key
        for r, c in deps:
            self.vd[p, r, 0] = c
        return self

    def to(self, o, row, col):
        return o.handle_exp_prepare_replies(_a(row, np.uint8), _a(col, np.uint32), self.nb, self.vb, self.vs, self.vq, self.vd,
                                            self.vk, self.fl)


def _pre_accept(o, sender, col, seq, deps, key, ballot=None, row=None):
    return o.handle_pre_accept(_a(1, np.uint8), _a(sender, np.uint8), _a(col, np.uint32),
                               _a(sender + 1 if ballot is None else ballot, np.uint64), _a(seq, np.uint64), deps, _a(key, np.uint8),
                               row=None if row is None else _a(row, np.uint8))


def _inst(o, row, col):
    d = o.dump()
    w = col % o.W
    return dict(bal=int(d["bal"][row, w, 0]), status=int(d["status"][row, w, 0]), seq=int(d["seq"][row, w, 0]),
                key=int(d["key"][row, w, 0]), deps=[int(x) for x in d["deps"][row, w, 0]], bk=int(d["bk"][row, w, 0]))


def _me2_with_row0_instance(oracle, seq=1, deps=None, key=1):
    """replica 2 holds (0, 0) PreAccepting as replica 0's PreAccept left it"""
    o = oracle.EpOracle(1, 5, me=2, W=8, n_keys=4)
    r = _pre_accept(o, 0, 0, seq, _deps() if deps is None else deps, key)
    assert int(r["flags"][0]) == 1
    return o


def test_heartbeat_timeout_starts_exp_prepare_on_the_peers_row(oracle):
    o = _me2_with_row0_instance(oracle)
    hb = o.heartbeat_timeout(_a(0, np.uint8), _a(1, np.uint8))
    # one in-progress instance in row 0; new ballot = make_greater_ballot(2, 1) = ((1 >> 8) + 1) << 8 | 3 (mod.rs:500-508)
    assert int(hb["n"][0]) == 1 and int(hb["col"][0, 0]) == 0 and int(hb["ballot"][0, 0]) == (1 << 8) | 3
    x = o.xp_dump()
    # fresh leader bookkeeping, my own reply in it: acks = {2}, voted at the instance's ballot 1 (heartbeat.rs:110-123)
    assert int(x["acks"][0, 0, 0]) == 0b100 and int(x["max_bal"][0, 0, 0]) == 1 and int(x["has"][0, 0, 0]) == 0b100
    assert int(x["vstatus"][0, 0, 2, 0]) == PREACC and int(x["vseq"][0, 0, 2, 0]) == 1 and int(x["vkey"][0, 0, 2, 0]) == 1
    i = _inst(o, 0, 0)
    assert i["bal"] == 1 and i["status"] == PREACC and i["bk"] & 1        # the instance itself is not touched (heartbeat.rs:99-101)
    # no HearTimeout, a timeout on myself, an unknown peer: nothing
    for src in (NO, 2, 7):
        assert int(o.heartbeat_timeout(_a(src, np.uint8))["n"][0]) == 0


def test_heartbeat_timeout_skips_committed_executed_and_foreign_instances(oracle):
    o = oracle.EpOracle(1, 5, me=2, W=8, n_keys=4)
    _pre_accept(o, 0, 0, 1, _deps(), 1)                                  # (0,0) PreAccepting from 0
    o.handle_commit_notice(_a(1, np.uint8), _a(0, np.uint8), _a(1, np.uint32), _a(1, np.uint64), _a(1, np.uint64), _deps(), _a(2, np.uint8))  # (0,1) Committed
    o.handle_accept(_a(1, np.uint8), _a(0, np.uint8), _a(2, np.uint32), _a(1, np.uint64), _a(4, np.uint64), _deps((0, 0)), _a(1, np.uint8))   # (0,2) Accepting from 0
    # (0,3): somebody else (replica 4) is already recovering it: its ExpPrepare made it the instance's source
    r = o.handle_exp_prepare(_a(1, np.uint8), _a(4, np.uint8), _a(0, np.uint8), _a(3, np.uint32), _a((1 << 8) | 5, np.uint64))
    assert int(r["flags"][0]) == 1 and int(r["voted_bal"][0]) == 0 and int(r["status"][0]) == NULL    # a null instance was filled in
    hb = o.heartbeat_timeout(_a(0, np.uint8))
    assert int(hb["n"][0]) == 2 and [int(hb["col"][k, 0]) for k in range(2)] == [0, 2]               # :73-84
    assert [int(hb["ballot"][k, 0]) for k in range(2)] == [(1 << 8) | 3, (1 << 8) | 3]


def test_exp_prepare_acceptor(oracle):
    o = _me2_with_row0_instance(oracle, seq=3, deps=_deps((1, 4)), key=2)
    nb = (1 << 8) | 2                                                      # replica 1 recovers row 0
    r = o.handle_exp_prepare(_a(1, np.uint8), _a(1, np.uint8), _a(0, np.uint8), _a(0, np.uint32), _a(nb, np.uint64))
    assert (int(r["flags"][0]), int(r["voted_bal"][0]), int(r["status"][0]), int(r["seq"][0]), int(r["key"][0])) == (1, 1, PREACC, 3, 2)
    assert [int(x) for x in r["deps"][:, 0]] == [N, 4, N, N, N]
    i = _inst(o, 0, 0)
    assert i["bal"] == 1 and (i["bk"] >> 2) == 1 and i["bk"] & 2           # only replica_bk.source moves (:538-542)
    # a ballot that is not larger than the instance's: no reply (:537)
    assert int(o.handle_exp_prepare(_a(1, np.uint8), _a(1, np.uint8), _a(0, np.uint8), _a(0, np.uint32), _a(1, np.uint64))["flags"][0]) == 0
    # and now a heartbeat timeout of mine on 0 skips the instance: its source is 1, not 0 (:75-79)
    assert int(o.heartbeat_timeout(_a(0, np.uint8))["n"][0]) == 0


def _preparing(oracle, **kw):
    o = _me2_with_row0_instance(oracle, **kw)
    hb = o.heartbeat_timeout(_a(0, np.uint8), _a(1, np.uint8))
    return o, int(hb["ballot"][0, 0])


def test_next_step_needs_a_simple_quorum(oracle):
    o, nb = _preparing(oracle)
    d = Replies(nb).add(1, 1, PREACC, 1, (), 1).to(o, 0, 0)
    assert int(d["decision"][0]) == 0                                       # 2 replies of 3 (dependency.rs:257-260)
    assert int(o.xp_dump()["acks"][0, 0, 0]) == 0b110
    # the same peer again: ignored (messages.rs:607-609)
    assert int(Replies(nb).add(1, 1, COMMITTED, 9, (), 1).to(o, 0, 0)["decision"][0]) == 0
    assert int(o.xp_dump()["vstatus"][0, 0, 1, 0]) == PREACC


def test_next_step_committed_reply_wins(oracle):
    o, nb = _preparing(oracle)
    d = Replies(nb).add(1, 1, PREACC, 1, (), 1).add(3, 1, COMMITTED, 5, ((1, 2),), 1).to(o, 0, 0)
    assert (int(d["decision"][0]), int(d["ballot"][0]), int(d["seq"][0]), int(d["key"][0])) == (COMMITTED, nb, 5, 1)
    assert [int(x) for x in d["deps"][:, 0]] == [N, 2, N, N, N]
    i = _inst(o, 0, 0)
    assert (i["bal"], i["status"], i["seq"]) == (nb, COMMITTED, 5)          # messages.rs:637-645
    assert int(o.dump()["commit_bars"][0, 0]) == 1 and list(o.xp_dump()["counters"]) == [1, 0, 0, 0]
    # later replies find the ballot taken (:603-605)
    assert int(Replies(nb).add(4, 1, PREACC, 1, (), 1).to(o, 0, 0)["decision"][0]) == 0


def test_next_step_accepting_reply_and_higher_voted_ballot(oracle):
    o, nb = _preparing(oracle)
    # peer 1 voted under ballot 0x102 (somebody's earlier recovery went as far as Accept): that clears my own lower-ballot
    # entry (:612-615); peer 3's vote under ballot 1 is then below the maximum and is not kept (:616-621)
    nb2 = (2 << 8) | 3
    o2 = _me2_with_row0_instance(oracle)
    # (replica 2 saw the Accept of that earlier recovery too? no: keep it at PreAccepting under ballot 1)
    hb = o2.heartbeat_timeout(_a(0, np.uint8))
    nb = int(hb["ballot"][0, 0])
    d = Replies(nb).add(1, 0x102, ACC, 7, ((3, 1),), 1).add(3, 1, PREACC, 1, (), 1).to(o2, 0, 0)
    x = o2.xp_dump()
    assert int(x["max_bal"][0, 0, 0]) == 0x102 and int(x["has"][0, 0, 0]) == 0b010 and int(x["acks"][0, 0, 0]) == 0b1110
    assert (int(d["decision"][0]), int(d["seq"][0])) == (ACC, 7) and [int(v) for v in d["deps"][:, 0]] == [N, N, N, 1, N]
    i = _inst(o2, 0, 0)
    assert (i["bal"], i["status"]) == (nb, ACC)
    assert int(o2.dump()["acc_acks"][0, 0, 0]) == 0b100                     # my own AcceptSlot completion (durability.rs:78-83)
    assert nb2 > nb


def test_next_step_enough_identical_pre_accepts_go_to_accept(oracle):
    o, nb = _preparing(oracle, seq=2, deps=_deps((1, 0)), key=1)
    # default ballot of row 0 everywhere, three identical PreAccepting votes, none from replica 0 (dependency.rs:286-315)
    d = Replies(nb).add(1, 1, PREACC, 2, ((1, 0),), 1).add(4, 1, PREACC, 2, ((1, 0),), 1).to(o, 0, 0)
    assert (int(d["decision"][0]), int(d["seq"][0])) == (ACC, 2)
    assert list(o.xp_dump()["counters"]) == [0, 1, 0, 0]


def test_next_step_differing_pre_accepts_start_over_and_avoid_the_fast_path(oracle):
    o, nb = _preparing(oracle, seq=2, deps=_deps((1, 0)), key=1)
    d = Replies(nb).add(1, 1, PREACC, 3, ((1, 0), (3, 2)), 1).add(4, 1, PREACC, 2, ((1, 0),), 1).to(o, 0, 0)
    # mine and 4's agree, 1's differs: 2 identical < 3; at the third reply (peer 4 in id order) the representative is the
    # PreAccepting vote of the highest peer id so far... the decision falls when the quorum is reached, i.e. at peer 1's
    # reply already (acks {2, 1} is 2; with peer 4: 3).  Replies are handled in peer order 1, then 4: after 1 there are two
    # acks -- undecided; after 4 three acks, voteds {1, 2, 4}: no 3 identical, has_pre_accept = peer 4's vote
    assert (int(d["decision"][0]), int(d["seq"][0])) == (PREACC, 2) and [int(v) for v in d["deps"][:, 0]] == [N, 0, N, N, N]
    i = _inst(o, 0, 0)
    assert (i["bal"], i["status"], i["seq"], i["key"]) == (nb, PREACC, 2, 1)
    x = o.xp_dump()
    assert int(x["avoid"][0, 0, 0]) == 1 and list(x["counters"]) == [0, 0, 1, 0]          # messages.rs:773
    assert int(o.dump()["pa_acks"][0, 0, 0]) == 0b100                                      # my own PreAcceptSlot completion
    # the PreAccept round that follows: two peers answer identically -- a super quorum of equal replies, and still the slow path
    ballot = np.zeros((5, 1), np.uint64); seq = np.zeros((5, 1), np.uint64); deps = np.full((5, 5, 1), N, np.uint32)
    flags = np.zeros((5, 1), np.uint8)
    for p in (1, 4):
        flags[p] = 1; ballot[p] = nb; seq[p] = 2; deps[p, 1, 0] = 0
    r = o.handle_pre_accept_replies(_a(0, np.uint32), ballot, seq, deps, flags, row=_a(0, np.uint8))
    assert int(r["decision"][0]) == ACC and _inst(o, 0, 0)["status"] == ACC


def test_next_step_nothing_voted_is_a_noop(oracle):
    o = oracle.EpOracle(1, 5, me=2, W=8, n_keys=4)
    _pre_accept(o, 0, 1, 1, _deps(), 1)                                     # (0,1) arrives, (0,0) is a hole: a null instance
    hb = o.heartbeat_timeout(_a(0, np.uint8))
    assert int(hb["n"][0]) == 2 and int(hb["ballot"][0, 0]) == (1 << 8) | 3   # the hole is prepared too, from ballot 0
    nb = int(hb["ballot"][0, 0])
    d = Replies(nb).add(1, 0, NULL).add(3, 0, NULL).to(o, 0, 0)
    assert (int(d["decision"][0]), int(d["seq"][0]), int(d["key"][0])) == (PREACC, 1, NO)   # dependency.rs:320-327
    assert (d["deps"] == N).all() and list(o.xp_dump()["counters"]) == [0, 0, 0, 1]
    i = _inst(o, 0, 0)
    assert (i["status"], i["key"], i["seq"]) == (PREACC, NO, 1)


def test_suspected_peer_releases_my_own_waiting_instance(oracle):
    """heartbeat.rs:35-60: an instance of mine that waits for a fast quorum it can still reach is re-evaluated when a peer it
    waits for is suspected"""
    o = oracle.EpOracle(1, 5, me=2, W=8, n_keys=4)
    o.propose(_a(1, np.uint8))
    ballot = np.zeros((5, 1), np.uint64); seq = np.zeros((5, 1), np.uint64); deps = np.full((5, 5, 1), N, np.uint32)
    flags = np.zeros((5, 1), np.uint8)
    flags[1] = 1; ballot[1] = 3; seq[1] = 2; deps[1, 1, 0] = 0            # differs from mine
    flags[3] = 1; ballot[3] = 3; seq[3] = 1                                # equals mine
    r = o.handle_pre_accept_replies(_a(0, np.uint32), ballot, seq, deps, flags)
    assert int(r["decision"][0]) == 0                                       # 2 identical + 2 outstanding >= 3: wait
    # peers 0 and 4 have not answered; 0 times out: 2 + (5 - 1 - 3) = 3 >= 3 still waits for 4
    hb = o.heartbeat_timeout(_a(0, np.uint8), _a(0b00001, np.uint8))
    assert _inst(o, 2, 0)["status"] == PREACC and int(hb["n"][0]) == 0
    # 4 times out as well (both timers exploded): 2 + (5 - 2 - 3) = 2 < 3 -> slow path
    o.heartbeat_timeout(_a(4, np.uint8), _a(0b10001, np.uint8))
    assert _inst(o, 2, 0)["status"] == ACC and list(o.dump()["counters"]) == [0, 1, 0]


def run_crash_and_recovery(mk, seed, loss, G=300, W=16, K=5, trace=None):
    """a cluster of five: normal ticks, the command leader 0 dies mid-instance (cut at PreAccept / Accept / Commit, deliveries to
    random subsets), then every live replica in turn times out on 0 and recovers what it knows of row 0 (a replica that
    holds an instance as Committed does not prepare it, heartbeat.rs:82-84: the others learn of it when their own turn comes)"""
    R = 5
    reps = [mk(G, R, r, W, K) for r in range(R)]
    rng = np.random.default_rng(seed)
    for t in range(3):
        ec.tick(reps, ec.zipf_keys(rng, R, G, K), {(s, q): rng.random(G) < 0.1 for s in range(R) for q in range(R) if s != q})
    cut = ec.crash_tick(reps, 0, ec.zipf_keys(rng, 1, G, K, p_propose=1.0)[0], rng, G)
    live = [1, 2, 3, 4]
    tallies = [ec.recover_row(reps, who, 0, live, G, rng, loss, trace=trace) for who in live]
    return reps, live, cut, tallies


@pytest.mark.parametrize("seed,loss", [(0, 0.0), (1, 0.0), (2, 0.2)])
def test_crash_and_recovery_keeps_agreement(oracle, seed, loss):
    """whatever two live replicas hold as committed in row 0 is the same; without loss every instance any live replica knew of
    ends up committed everywhere it is known"""
    G, W = 300, 16
    reps, live, cut, tallies = run_crash_and_recovery(lambda G, R, r, W, K: oracle.EpOracle(G, R, me=r, W=W, n_keys=K), seed, loss, G, W)
    n = ec.check_agreement(reps, live, 0, G)
    assert n > 0 and all(sum(t.values()) > 0 for t in tallies[:1])
    assert {int(c) for c in cut} == {0, 1, 2}
    dec = {k: sum(t[k] for t in tallies) for k in (1, 2, 3)}
    assert dec[1] > 0 and dec[2] > 0 and dec[3] > 0                        # every kind of next step was taken somewhere
    if loss == 0.0:
        for q in live:
            d = reps[q].dump()
            for g in range(G):
                n_cols = int(d["len"][0, g])
                st = d["status"][0, :, g]
                assert all(st[c % W] >= COMMITTED for c in range(max(0, n_cols - W), n_cols)), (q, g)

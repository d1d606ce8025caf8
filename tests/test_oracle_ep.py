"""EPaxos command-leader / acceptor handlers of the CPU oracle against hand-derived traces of the
reference code (src/protocols/epaxos/{request,messages,dependency,durability}.rs), R = 5,
optimized quorums: simple 3, super 3 (mod.rs:693-698)."""
import numpy as np

N = 0xFFFFFFFF
NULL, PREACC, ACC, COMMITTED, EXECUTED = 0, 1, 2, 3, 5


def _a(v, t):
    return np.array([v], t)


def _deps(*pairs, R=5):
    d = np.full((R, 1), N, np.uint32)
    for r, c in pairs:
        d[r, 0] = c
    return d


def _replies(R=5):
    return dict(ballot=np.zeros((R, 1), np.uint64), seq=np.zeros((R, 1), np.uint64), deps=np.full((R, R, 1), N, np.uint32),
                flags=np.zeros((R, 1), np.uint8))


def _reply(m, p, ballot, seq, *pairs):
    m["flags"][p] = 1
    m["ballot"][p] = ballot
    m["seq"][p] = seq
    for r, c in pairs:
        m["deps"][p, r, 0] = c


def _st(o, row, col):
    d = o.dump()
    return int(d["status"][row, col % o.W, 0])


def test_propose_deps_and_seq(oracle):
    o = oracle.EpOracle(1, 5, me=0, W=8, n_keys=4)
    m = o.propose(_a(1, np.uint8))
    assert (int(m["flags"][0]), int(m["col"][0]), int(m["seq"][0])) == (1, 0, 1) and (m["deps"] == N).all()
    d = o.dump()
    assert int(d["bal"][0, 0, 0]) == 1                        # make_default_ballot(0) = (0 << 8) | 1
    assert int(d["pa_acks"][0, 0, 0]) == 0b1 and _st(o, 0, 0) == PREACC   # my own PreAcceptSlot completion = my reply
    assert int(d["highest_cols"][1, 0, 0]) == 0
    m = o.propose(_a(1, np.uint8))                            # same key: depends on (0, 0); seq = 1 + seq(0,0)
    assert (int(m["col"][0]), int(m["seq"][0]), int(m["deps"][0, 0])) == (1, 2, 0)
    m = o.propose(_a(2, np.uint8))                            # other key: no deps
    assert (int(m["col"][0]), int(m["seq"][0])) == (2, 1) and (m["deps"] == N).all()
    assert int(o.propose(_a(0xFF, np.uint8))["flags"][0]) == 0


def test_fast_path_three_identical(oracle):
    o = oracle.EpOracle(1, 5, me=0, W=8, n_keys=4)
    o.propose(_a(1, np.uint8))
    m = _replies()
    _reply(m, 1, 1, 1)
    r = o.handle_pre_accept_replies(_a(0, np.uint32), **m)
    assert int(r["decision"][0]) == 0                          # 2 replies < simple quorum: undecided
    m = _replies()
    _reply(m, 2, 1, 1)
    r = o.handle_pre_accept_replies(_a(0, np.uint32), **m)
    assert int(r["decision"][0]) == COMMITTED and int(r["seq"][0]) == 1
    d = o.dump()
    assert int(d["commit_bars"][0, 0]) == 1 and list(d["counters"]) == [1, 0, 0]
    m = _replies()
    _reply(m, 3, 1, 9, (3, 3))                                 # late reply: status is not PreAccepting any more
    assert int(o.handle_pre_accept_replies(_a(0, np.uint32), **m)["decision"][0]) == 0
    assert int(o.dump()["pa_acks"][0, 0, 0]) == 0b111


def test_slow_path_when_fast_quorum_is_out_of_reach(oracle):
    o = oracle.EpOracle(1, 5, me=0, W=8, n_keys=4)
    o.propose(_a(1, np.uint8))                                 # mine: (1, {})
    m = _replies()
    _reply(m, 1, 1, 2, (1, 0))
    _reply(m, 2, 1, 3, (2, 5))
    r = o.handle_pre_accept_replies(_a(0, np.uint32), **m)
    assert int(r["decision"][0]) == 0                          # 3 replies, all different: 1 + (5 - 0 - 3) = 3 >= 3, wait
    m = _replies()
    _reply(m, 3, 1, 2, (1, 0))                                 # equals peer 1's: class of 2, 2 + (5 - 4) = 3: still wait
    assert int(o.handle_pre_accept_replies(_a(0, np.uint32), **m)["decision"][0]) == 0
    m = _replies()
    _reply(m, 4, 1, 4, (4, 1))                                 # all 5 in, best class 2 < 3: slow path
    r = o.handle_pre_accept_replies(_a(0, np.uint32), **m)
    assert int(r["decision"][0]) == ACC and int(r["seq"][0]) == 4     # max seq, union of deps
    assert [int(x) for x in r["deps"][:, 0]] == [N, 0, 5, N, 1]
    d = o.dump()
    assert _st(o, 0, 0) == ACC and int(d["acc_acks"][0, 0, 0]) == 0b1     # my own AcceptSlot completion
    # slow-path tally: commits at simple quorum (3) incl. me
    a = dict(ballot=np.zeros((5, 1), np.uint64), flags=np.zeros((5, 1), np.uint8))
    a["flags"][2] = 1
    a["ballot"][2] = 1
    assert int(o.handle_accept_replies(_a(0, np.uint32), **a)["committed"][0]) == 0
    a["flags"][:] = 0
    a["flags"][4] = 1
    a["ballot"][4] = 7                                         # wrong ballot: ignored (messages.rs:373)
    assert int(o.handle_accept_replies(_a(0, np.uint32), **a)["committed"][0]) == 0
    a["ballot"][4] = 1
    assert int(o.handle_accept_replies(_a(0, np.uint32), **a)["committed"][0]) == 1
    assert _st(o, 0, 0) == COMMITTED and list(o.dump()["counters"]) == [0, 1, 1]


def test_exploded_timers_shrink_the_reachable_quorum(oracle):
    o = oracle.EpOracle(1, 5, me=0, W=8, n_keys=4)
    o.propose(_a(1, np.uint8))
    m = _replies()
    _reply(m, 1, 1, 2, (1, 0))
    _reply(m, 2, 1, 3, (2, 5))
    ex = _a(0b11000, np.uint8)                                 # peers 3 and 4 suspected: bad = 2
    r = o.handle_pre_accept_replies(_a(0, np.uint32), exploded=ex, **m)
    assert int(r["decision"][0]) == ACC                        # 1 + (5 - 2 - 3) = 1 < 3
    # the "failure suspected" re-evaluation call (ballot == 0, messages.rs:108-144) records nothing
    o = oracle.EpOracle(1, 5, me=0, W=8, n_keys=4)
    o.propose(_a(1, np.uint8))
    m = _replies()
    _reply(m, 1, 1, 2, (1, 0))
    _reply(m, 2, 1, 3, (2, 5))
    assert int(o.handle_pre_accept_replies(_a(0, np.uint32), **m)["decision"][0]) == 0
    m = _replies()
    _reply(m, 3, 0, 0)
    r = o.handle_pre_accept_replies(_a(0, np.uint32), exploded=ex, **m)
    assert int(r["decision"][0]) == ACC and int(o.dump()["pa_acks"][0, 0, 0]) == 0b111


def test_acceptor_merges_its_own_knowledge(oracle):
    o = oracle.EpOracle(1, 5, me=1, W=8, n_keys=4)
    none = _deps()
    r = o.handle_pre_accept(_a(1, np.uint8), _a(0, np.uint8), _a(0, np.uint32), _a(1, np.uint64), _a(1, np.uint64), none,
                            _a(1, np.uint8))
    assert (int(r["flags"][0]), int(r["ballot"][0]), int(r["seq"][0])) == (1, 1, 1) and (r["deps"] == N).all()
    # replica 2 proposes on the same key without having seen (0, 0): I add it (messages.rs:42-45)
    r = o.handle_pre_accept(_a(1, np.uint8), _a(2, np.uint8), _a(0, np.uint32), _a(3, np.uint64), _a(1, np.uint64), none,
                            _a(1, np.uint8))
    assert int(r["seq"][0]) == 2 and [int(x) for x in r["deps"][:, 0]] == [0, N, N, N, N]
    d = o.dump()
    assert [int(x) for x in d["highest_cols"][1, :, 0]] == [0, N, 0, N, N]
    assert int(d["bk"][2, 0, 0]) == (2 | (2 << 2))             # replica_bk.source = 2
    # an older ballot for a slot I already hold is ignored (:40)
    r = o.handle_pre_accept(_a(1, np.uint8), _a(2, np.uint8), _a(0, np.uint32), _a(2, np.uint64), _a(9, np.uint64), none,
                            _a(1, np.uint8))
    assert int(r["flags"][0]) == 0 and int(o.dump()["seq"][2, 0, 0]) == 2
    # a PreAccept for column 3 of an empty row pads it with null instances (:33-36)
    o.handle_pre_accept(_a(1, np.uint8), _a(4, np.uint8), _a(3, np.uint32), _a(5, np.uint64), _a(1, np.uint64), none,
                        _a(2, np.uint8))
    d = o.dump()
    assert int(d["len"][4, 0]) == 4 and [int(d["status"][4, c, 0]) for c in range(4)] == [NULL, NULL, NULL, PREACC]
    # Accept overwrites seq / deps as given (messages.rs:300-306) and is answered with the ballot
    r = o.handle_accept(_a(1, np.uint8), _a(2, np.uint8), _a(0, np.uint32), _a(3, np.uint64), _a(7, np.uint64),
                        _deps((3, 4)), _a(1, np.uint8))
    assert (int(r["flags"][0]), int(r["ballot"][0])) == (1, 3)
    d = o.dump()
    assert int(d["status"][2, 0, 0]) == ACC and int(d["seq"][2, 0, 0]) == 7 and int(d["deps"][2, 0, 0, 3]) == 4


def test_commit_bar_waits_for_holes(oracle):
    o = oracle.EpOracle(1, 5, me=0, W=8, n_keys=4)
    o.propose(_a(1, np.uint8))
    o.propose(_a(2, np.uint8))
    m = _replies()
    _reply(m, 1, 1, 1)
    _reply(m, 2, 1, 1)
    assert int(o.handle_pre_accept_replies(_a(1, np.uint32), **m)["decision"][0]) == COMMITTED   # column 1 first
    assert int(o.dump()["commit_bars"][0, 0]) == 0             # durability.rs:121: only a commit AT the bar moves it
    assert int(o.handle_pre_accept_replies(_a(0, np.uint32), **m)["decision"][0]) == COMMITTED
    assert int(o.dump()["commit_bars"][0, 0]) == 2

"""EPaxos dependency-graph execution (ep_execute_kernel behind every handler call, smr_ep_cfg.execute)
against the CPU oracle: the hand-derived traces of tests/test_oracle_ep_exec.py, seeded handler
streams compared after every call (instance space, bars, store, digest, counters: bit-exact), and the
closed-loop five-replica cluster.

This file sorts last on purpose.  The kernel was written after the round's GPU minutes were spent;
until its first run on the device it has only been run as host code (tests/test_hostsim.py), and
under `pytest -x` a surprise here must not mask the suites that have run on the device before."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _same_all(eng, orc, step):
    a, b = eng.dump(), orc.dump()
    for n in b:
        assert np.array_equal(a[n], b[n]), (step, n, [x[:5] for x in np.nonzero(a[n] != b[n])])
    # the submissions of the call just made, in order: what a host applies to its state machine
    for x, y in zip(eng.take_submissions(), orc.take_submissions()):
        assert np.array_equal(x, y), (step, "submissions", len(x), len(y))
    a, b = eng.exec_dump(), orc.exec_dump()
    for n in ("exec_bars", "kv", "digest"):
        assert np.array_equal(a[n], b[n]), (step, n, [x[:5] for x in np.nonzero(a[n] != b[n])])
    assert b["counters"][3] == 0, "a component with more than one node: the forest argument is wrong"
    assert [int(v) for v in a["counters"]] == [int(v) for v in b["counters"]], (step, a["counters"], b["counters"])


def test_execution_traces(cuda, oracle):
    import ep_cluster as ec
    import test_oracle_ep_exec as tr
    from summerset_amd import EPaxosReplicaGroup
    for name in tr.TRACES:
        getattr(tr, name)(ec.NumpyEngine(EPaxosReplicaGroup(1, 5, me=4, window=8, n_keys=4, execute=True), cuda))


@pytest.mark.parametrize("G,W,me", [(600, 32, 0), (2048, 16, 3)])
def test_handler_streams_with_execution(cuda, oracle, G, W, me):
    """the streams of tests/test_ep_gpu.py plus CommitNotices, execution on: random messages overwrite
    instances in every state, so the walk meets regressed, re-committed, unheld and Null slots"""
    import ep_cluster as ec
    import ep_scenarios as sc
    from summerset_amd import EPaxosReplicaGroup, stream
    R, K = 5, 8
    rng = np.random.default_rng(G + W + me + 1)
    eng = ec.NumpyEngine(EPaxosReplicaGroup(G, R, me=me, window=W, n_keys=K, execute=True), cuda)
    orc = oracle.EpOracle(G, R, me=me, W=W, n_keys=K, execute=True)
    slow_cols = np.zeros(G, np.uint32)
    for step in range(40):
        ctl = np.ascontiguousarray(stream.random_ackctl(13, step, 1, G, R, 0.0)[0])
        m = sc.acceptor_round(rng, orc.dump(), G, R, me, K, W)
        eng.handle_pre_accept(**m); orc.handle_pre_accept(**m)
        m = sc.acceptor_round(rng, orc.dump(), G, R, me, K, W)
        eng.handle_accept(**m); orc.handle_accept(**m)
        for _ in range(3):                                       # commits of the peers' instances, mostly in column order
            m = sc.commit_round(rng, orc.dump(), G, R, me, K, W)
            eng.handle_commit_notice(**m); orc.handle_commit_notice(**m)
            _same_all(eng, orc, (step, "commit"))
        key, ex = sc.propose_round(rng, G, K)
        po = orc.propose(key, ex)
        eng.propose(key, ex)
        for wave in range(2):
            m = sc.pre_accept_replies_round(rng, orc.dump(), po, G, R, me, ctl)
            ro = orc.handle_pre_accept_replies(**m)
            re_ = eng.handle_pre_accept_replies(**m)
            assert np.array_equal(re_["decision"], ro["decision"]), (step, wave)
            slow_cols = np.where(ro["decision"] == 2, po["col"], slow_cols).astype(np.uint32)
        m = sc.accept_replies_round(rng, slow_cols, G, R, me, ctl)
        assert np.array_equal(eng.handle_accept_replies(**m)["committed"], orc.handle_accept_replies(**m)["committed"])
        _same_all(eng, orc, (step, "leader"))
    c = dict(zip(("n_exec", "n_reexec", "n_unheld", "n_multi", "n_attempts", "n_aborts"), (int(v) for v in orc.exec_dump()["counters"])))
    assert c["n_exec"] > G and c["n_reexec"] > 0 and c["n_aborts"] > 0 and c["n_unheld"] > 0, c
    assert (orc.exec_dump()["exec_bars"] > 0).any()


@pytest.mark.parametrize("drop_p", [0.0, 0.15])
def test_closed_loop_cluster_with_execution(cuda, oracle, drop_p):
    import ep_cluster as ec
    from summerset_amd import EPaxosReplicaGroup
    G, R, W, K, T = 500, 5, 32, 6, 12
    engs = [ec.NumpyEngine(EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K, execute=True), cuda) for r in range(R)]
    orcs = [oracle.EpOracle(G, R, me=r, W=W, n_keys=K, execute=True) for r in range(R)]
    rng = np.random.default_rng(8)
    for t in range(T):
        keys = ec.zipf_keys(rng, R, G, K)
        drop = {(s, q): rng.random(G) < drop_p for s in range(R) for q in range(R) if s != q} if drop_p else None
        oe, oo = ec.tick(engs, keys, drop), ec.tick(orcs, keys, drop)
        for s in range(R):
            for k in oo[s]:
                assert np.array_equal(oe[s][k], oo[s][k]), (t, s, k)
        for r in range(R):
            _same_all(engs[r], orcs[r], (t, r))
    x = orcs[0].exec_dump()
    assert int(x["counters"][0]) > 0 and int(x["counters"][1]) > 0
    if not drop_p:
        assert np.array_equal(x["exec_bars"], orcs[0].dump()["commit_bars"])

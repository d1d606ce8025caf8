"""RSPaxos replica kernels (summerset_amd/csrc/rsp_engine.hip, through the C-ABI) against the CPU oracle: the
hand-derived traces of tests/test_oracle_rsp.py on the engine, and five per-replica engine objects wired into
the closed-loop cluster of tests/rsp_cluster.py (steady appends, loss, two leader changes with shard merging,
re-Accepts and reconstruction reads, heartbeats) against five oracles wired the same way -- every message
and, after every tick, every replica's full state: bit-exact.

Sorts last on purpose, like tests/test_zz_ep_exec_gpu.py: these kernels were written after the round's GPU
minutes were spent and have so far only run as host code (tests/test_hostsim.py)."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def test_traces_on_the_engine(cuda, oracle):
    import rsp_cluster as rc
    import test_oracle_rsp as tr
    from summerset_amd import RSPaxosReplicaGroup

    def make(me, ft=0, W=8):
        e = rc.NumpyEngine(RSPaxosReplicaGroup(1, 5, me=me, window=W, fault_tolerance=ft), cuda)
        e.preset_leader(0)
        return e
    for name in tr.TRACES:
        getattr(tr, name)(make)


def _same(engs, orcs, where):
    for r in range(len(engs)):
        for x, y in zip(engs[r].take_executed(), orcs[r].take_executed()):    # what a host would have applied, in order
            assert np.array_equal(x, y), (where, r, "executed", len(x), len(y))
        a, b = engs[r].dump(), orcs[r].dump()
        for n in b:
            assert np.array_equal(a[n], b[n]), (where, r, n, [x[:4] for x in np.nonzero(a[n] != b[n])])


@pytest.mark.parametrize("G,W,ft,loss", [(300, 32, 0, 0.0), (300, 32, 1, 0.1), (1500, 16, 1, 0.05)])
def test_closed_loop_cluster_matches_oracle(cuda, oracle, G, W, ft, loss):
    import rsp_cluster as rc
    import rsp_scenarios as sc
    from summerset_amd import RSPaxosReplicaGroup
    R, T = 5, 21
    engs = [rc.NumpyEngine(RSPaxosReplicaGroup(G, R, me=r, window=W, fault_tolerance=ft), cuda) for r in range(R)]
    orcs = [oracle.RspOracle(G, R, me=r, W=W, fault_tolerance=ft) for r in range(R)]
    # the two clusters run the same seeded scenario one after the other, compared tick by tick through their logs,
    # and state by state at the end (the oracles' per-tick dumps are kept for the comparison)
    snaps, execd = [], []
    lo = sc.run(orcs, G, T, seed=G + ft, loss=loss,
                on_tick=lambda t: (snaps.append([o.dump() for o in orcs]), execd.append([o.take_executed() for o in orcs])))
    step = [0]

    def check(t):
        for r in range(R):
            for x, y in zip(engs[r].take_executed(), execd[t][r]):           # what a host would have applied this tick, in order
                assert np.array_equal(x, y), (t, r, "executed", len(x), len(y))
            a, b = engs[r].dump(), snaps[t][r]
            for n in b:
                assert np.array_equal(a[n], b[n]), (t, r, n, [x[:4] for x in np.nonzero(a[n] != b[n])])
        step[0] += 1
    le = sc.run(engs, G, T, seed=G + ft, loss=loss, on_tick=check)
    assert step[0] == T
    for (t, a), (_, b) in zip(le, lo):
        assert len(a) == len(b), t
        for x, y in zip(a, b):
            for k in y:
                assert np.array_equal(x[k], y[k]) if isinstance(y[k], np.ndarray) else x[k] == y[k], (t, y["kind"], k)
    ev = [e for _, out in lo for e in out]
    c = orcs[0].dump()["counters"]
    assert c[0] > 0 and c[1] > 0 and c[2] == 0
    assert sum(e["voted"] for e in ev if e["kind"] == "prepare_reply") > 0 and sum(e["n"] for e in ev if e["kind"] == "re_accept") > 0
    assert sum(e["rows"] for e in ev if e["kind"] == "recon_reply") > 0


@pytest.mark.parametrize("G,W,me,ft", [(500, 8, 0, 0), (500, 16, 2, 1), (500, 32, 4, 2)])
def test_random_handler_calls_match_oracle(cuda, oracle, G, W, me, ft):
    """differential: seeded random (not protocol-legal) calls, small rings so that slots leave them all the time"""
    import rsp_cluster as rc
    import rsp_random as rr
    from summerset_amd import RSPaxosReplicaGroup
    R = 5
    eng = rc.NumpyEngine(RSPaxosReplicaGroup(G, R, me=me, window=W, fault_tolerance=ft), cuda)
    orc = oracle.RspOracle(G, R, me=me, W=W, fault_tolerance=ft)
    eng.preset_leader(0); orc.preset_leader(0)
    rng = np.random.default_rng(G + W + me)
    seen = set()
    for step in range(160):
        for name, kw in rr.calls(rng, orc.dump(), G, R, me, W):
            seen.add(name)
            a, b = getattr(eng, name)(**kw), getattr(orc, name)(**kw)
            if b is not None:
                for k in b:
                    assert np.array_equal(a[k], b[k]), (step, name, k, [x[:4] for x in np.nonzero(a[k] != b[k])])
            _same([eng], [orc], (step, name))
    assert len(seen) == 10
    c = orc.dump()["counters"]
    assert c[0] > 0 and c[1] > 0

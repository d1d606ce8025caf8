"""A closed-loop EPaxos cluster built from five per-replica oracles (tests/ep_cluster.py): the
protocol's own safety properties must hold on the restatement -- every replica ends with the same
(seq, deps) for a committed instance, and two committed instances that touch the same key see each
other: at least one has the other (or a later instance of its row) in its dependencies."""
import numpy as np

import ep_cluster as ec

N = 0xFFFFFFFF


def _run(oracle, G, ticks, n_keys, seed, drop_p=0.0, execute=False):
    R, W = 5, 64
    reps = [oracle.EpOracle(G, R, me=r, W=W, n_keys=n_keys, execute=execute) for r in range(R)]
    rng = np.random.default_rng(seed)
    log = []
    for t in range(ticks):
        keys = ec.zipf_keys(rng, R, G, n_keys)
        drop = None
        if drop_p:
            drop = {(s, q): rng.random(G) < drop_p for s in range(R) for q in range(R) if s != q}
        log.append((keys, ec.tick(reps, keys, drop)))
    return reps, log


def _check(reps, log, G):
    R = 5
    dumps = [r.dump() for r in reps]
    W = reps[0].W
    committed = []                                         # (g, row, col, key, seq, deps)
    n_fast = n_slow = 0
    for keys, out in log:
        for s in range(R):
            o = out[s]
            for g in np.nonzero(o["committed"])[0]:
                col = int(o["col"][g])
                committed.append((int(g), s, col, int(keys[s][g]), int(o["seq"][g]), tuple(int(x) for x in o["deps"][:, g])))
            n_fast += int((o["decision"] == 3).sum())
            n_slow += int((o["decision"] == 2).sum())
    assert n_fast > 0 and n_slow > 0 and committed
    # agreement: all replicas hold the committed value
    for g, row, col, key, seq, deps in committed:
        for q in range(R):
            d = dumps[q]
            w = col % W
            assert d["status"][row, w, g] >= 3, (g, row, col, q)
            assert int(d["seq"][row, w, g]) == seq and tuple(int(x) for x in d["deps"][row, w, g]) == deps, (g, row, col, q)
            assert int(d["key"][row, w, g]) == key
    # interference: committed instances on the same key are ordered by their dependencies
    by = {}
    for c in committed:
        by.setdefault((c[0], c[3]), []).append(c)
    pairs = 0
    for lst in by.values():
        for i in range(len(lst)):
            for j in range(i + 1, len(lst)):
                a, b = lst[i], lst[j]
                if a[1] == b[1]:
                    continue                               # same row: ordered by column
                a_sees_b = a[5][b[1]] != N and a[5][b[1]] >= b[2]
                b_sees_a = b[5][a[1]] != N and b[5][a[1]] >= a[2]
                assert a_sees_b or b_sees_a, (a, b)
                pairs += 1
    assert pairs > 0
    return n_fast, n_slow


def test_cluster_agreement_and_interference(oracle):
    reps, log = _run(oracle, G=40, ticks=12, n_keys=6, seed=1)
    _check(reps, log, 40)
    # everything proposed commits when nothing is lost, and the commit bars follow
    for keys, out in log:
        for s in range(5):
            assert np.array_equal(out[s]["committed"], out[s]["proposed"])
    bars = [r.dump()["commit_bars"] for r in reps]
    for q in range(1, 5):
        assert np.array_equal(bars[0], bars[q])


def test_cluster_with_lost_pre_accepts(oracle):
    # lost PreAccepts: fewer replies, more slow paths and undecided instances, the same safety properties
    reps, log = _run(oracle, G=40, ticks=12, n_keys=6, seed=2, drop_p=0.25)
    _check(reps, log, 40)


def _exec_props(oracle, drop_p, seed):
    G, T, K = 60, 14, 6
    reps, log = _run(oracle, G, T, K, seed, drop_p, execute=True)
    plain, _ = _run(oracle, G, T, K, seed, drop_p)
    xs, ds, ps = [r.exec_dump() for r in reps], [r.dump() for r in reps], [r.dump() for r in plain]
    for q in range(5):
        x, d, p = xs[q], ds[q], ps[q]
        c = dict(zip(("n_exec", "n_reexec", "n_unheld", "n_multi_scc", "n_attempts", "n_aborts"), (int(v) for v in x["counters"])))
        # the graph is a forest (every component a single node), and no dependency left the ring (W = 64 > ticks):
        # nothing the harness adds to the reference was reached
        assert c["n_multi_scc"] == 0 and c["n_unheld"] == 0 and c["n_exec"] > 0, c
        assert (x["exec_bars"] <= d["commit_bars"]).all()
        # execution only moves instances from Committed on; the protocol state is what it is without it
        for n in ("len", "commit_bars", "bal", "seq", "key", "deps", "pa_acks", "acc_acks", "bk", "highest_cols", "counters"):
            assert np.array_equal(d[n], p[n]), n
        assert np.array_equal(np.minimum(d["status"], 3), np.minimum(p["status"], 3))
        # below its exec bar a row is Executed
        W = reps[q].W
        for row in range(5):
            for g in range(G):
                for col in range(int(x["exec_bars"][row, g])):
                    assert d["status"][row, col % W, g] == 5, (q, row, col, g)
        # lock-step delivers the commits in the same order everywhere: same submissions, same store
        assert np.array_equal(x["digest"], xs[0]["digest"]) and np.array_equal(x["kv"], xs[0]["kv"])
    return xs, ds


def test_cluster_execution(oracle):
    xs, ds = _exec_props(oracle, 0.0, 4)
    for q in range(5):                                         # nothing lost: everything committed has run
        assert np.array_equal(xs[q]["exec_bars"], ds[q]["commit_bars"])
    assert int(xs[0]["counters"][1]) > 0                       # and some instances ran twice (add_edge re-inserting a pruned slot)


def test_cluster_execution_with_lost_pre_accepts(oracle):
    xs, ds = _exec_props(oracle, 0.2, 6)
    assert int(xs[0]["counters"][5]) > 0                       # attempts abandoned on an uncommitted dependency
    assert (xs[0]["exec_bars"] < ds[0]["commit_bars"]).any()


def test_golden_final_states(oracle):
    """the frozen EPaxos and RSPaxos cluster runs end in the committed states (tests/golden/late_golden.npz, generator
    tests/golden/make_golden.py: cluster_states) -- pins both oracles against drift"""
    import importlib.util
    import os
    here = os.path.dirname(__file__)
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = np.load(os.path.join(here, "golden", "late_golden.npz"))
    out = mg.cluster_states(oracle)
    assert len(out) > 50
    for k, v in out.items():
        assert np.array_equal(v, gold[k]), k

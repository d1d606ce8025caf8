"""A closed-loop RSPaxos cluster built from five per-replica oracles (tests/rsp_cluster.py): the protocol's own
safety properties must hold on the restatement -- one value per slot among all replicas that consider it
committed, executed prefixes agree, no shards of different values ever merged -- and the erasure-coded parts
must actually be reached: followers hold one shard and cannot execute, a new leader rebuilds instances from
PrepareReply shards and reconstruction reads."""
import numpy as np

import rsp_cluster as rc
import rsp_scenarios as sc


def _cluster(oracle, G, ft, W=64):
    return [oracle.RspOracle(G, 5, me=r, W=W, fault_tolerance=ft) for r in range(5)]


def _safety(reps, log, G):
    R = 5
    ds = [r.dump() for r in reps]
    W = reps[0].W
    chosen = {}                                                  # (g, slot) -> value, from the leaders' commit events
    for _, out in log:
        for e in out:
            if e["kind"] != "commit":
                continue
            for g in np.nonzero(e["committed"])[0]:
                k = (int(g), int(e["slot"][g]))
                assert chosen.setdefault(k, int(e["val"][g])) == int(e["val"][g]), k
    assert chosen
    for d in ds:
        assert d["counters"][2] == 0                             # never merged shards of different values
        assert (d["exec_bar"] <= d["commit_bar"]).all() and (d["commit_bar"] <= d["len"]).all()
    for (g, slot), val in chosen.items():                        # whoever holds the slot as committed holds that value
        for q in range(R):
            d = ds[q]
            if slot < d["len"][g] and slot + W >= d["len"][g] and d["s_status"][slot % W, g] >= 3:
                assert int(d["s_val"][slot % W, g]) == val, (g, slot, q)
    # executed prefixes agree: the replica that executed the most is the reference for the others
    for g in range(G):
        bars = [int(d["exec_bar"][g]) for d in ds]
        top = int(np.argmax(bars))
        for q in range(R):
            for slot in range(bars[q]):
                if slot + W >= ds[q]["len"][g] and slot + W >= ds[top]["len"][g]:
                    assert ds[q]["s_val"][slot % W, g] == ds[top]["s_val"][slot % W, g], (g, slot, q, top)
    return ds, chosen


def test_steady_state_followers_hold_one_shard(oracle):
    G = 30
    reps = _cluster(oracle, G, ft=1)
    log = sc.run(reps, G, 12, seed=1, loss=0.0, changes=False)
    ds, chosen = _safety(reps, log, G)
    # nothing lost: every batch commits in its tick (needs majority + fault_tolerance = 4 acks) and the leader executes it
    n_commits = sum(int(e["committed"].sum()) for _, out in log for e in out if e["kind"] == "commit")
    assert n_commits == len(chosen) and ds[0]["counters"][0] == len(chosen) and ds[0]["counters"][1] == len(chosen)
    assert np.array_equal(ds[0]["exec_bar"], ds[0]["len"])
    for q in range(1, 5):                                        # followers: one shard each, learn commits by heartbeat, cannot run them
        d = ds[q]
        live = d["s_status"] >= 2
        assert (d["s_mask"][live] == (1 << q)).all()
        assert (d["s_status"] == 3).any() and not d["exec_bar"].any() and not d["commit_bar"].any()


def test_leader_changes_rebuild_from_shards(oracle):
    G = 40
    for ft in (0, 1):
        reps = _cluster(oracle, G, ft)
        log = sc.run(reps, G, 24, seed=3 + ft, loss=0.0)
        ds, _ = _safety(reps, log, G)
        ev = [e for _, out in log for e in out]
        assert sum(e["voted"] for e in ev if e["kind"] == "prepare_reply") > 0       # PrepareReplies carried voted shards
        assert sum(e["n"] for e in ev if e["kind"] == "re_accept") > 0               # the quorum released re-Accepts
        assert sum(e["rows"] for e in ev if e["kind"] == "recon_reply") > 0          # reconstruction reads were answered
        # replica 1 leads the even groups and has run the instances the old leader had committed
        d1 = ds[1]
        even = np.arange(G) % 4 == 2                             # (groups 0, 1 mod 4 moved on to replica 2)
        assert (d1["leader"][even] == 1).all() and (d1["exec_bar"][even] > 0).all()
        assert (ds[2]["leader"][np.arange(G) % 4 == 0] == 2).all()


def test_with_loss(oracle):
    G = 40
    for ft in (0, 1):
        reps = _cluster(oracle, G, ft)
        log = sc.run(reps, G, 24, seed=7 + ft, loss=0.1)
        _safety(reps, log, G)

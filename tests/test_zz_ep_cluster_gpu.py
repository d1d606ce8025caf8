"""The device-resident EPaxos cluster tick (summerset_amd/ep_cluster.py: every message a device tensor between the
handlers) against the numpy-staged driver of tests/ep_cluster.py on five ORACLES wired the same way -- per-leader
decisions of every tick and every replica's final state, with lost PreAccepts and execution on."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("G,K,loss", [(700, 6, 0.15), (2048, 64, 0.0)])
def test_device_cluster_tick_matches_the_oracle_cluster(cuda, oracle, G, K, loss):
    import torch
    import ep_cluster as ec
    from summerset_amd import EPaxosReplicaGroup, ep_cluster
    R, W, T = 5, 32, 9
    engs = [EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K, execute=True) for r in range(R)]
    orcs = [oracle.EpOracle(G, R, me=r, W=W, n_keys=K, execute=True) for r in range(R)]
    rng = np.random.default_rng(G + K)
    fast = slow = 0
    for t in range(T):
        keys = ec.zipf_keys(rng, R, G, K)
        drop = {(s, q): rng.random(G) < loss for s in range(R) for q in range(R) if s != q} if loss else None
        oo = ec.tick(orcs, keys, drop)
        oe = ep_cluster.tick(engs, [torch.from_numpy(np.ascontiguousarray(keys[r])).to(cuda) for r in range(R)],
                             None if drop is None else {k: torch.from_numpy(v).to(cuda) for k, v in drop.items()})
        for s in range(R):
            for k in oo[s]:
                e = oe[s][k].cpu().numpy()
                assert np.array_equal(e.view(oo[s][k].dtype), oo[s][k]), (t, s, k)
            fast += int((oo[s]["decision"] == 3).sum())
            slow += int((oo[s]["decision"] == 2).sum())
    for r in range(R):
        a, b = engs[r].dump(), orcs[r].dump()
        for n in b:
            assert np.array_equal(a[n], b[n]), (r, n)
        xa, xb = engs[r].exec_dump(), orcs[r].exec_dump()
        for n in xb:
            assert np.array_equal(xa[n], xb[n]), (r, "exec", n)
    assert fast > 0 and (slow > 0 or loss == 0.0)


def run_fused_vs_driver(dev, G, K, loss, T=7, execute=True, seed=5, R=5, W=32, oracle=None, phase_major=False):
    """`smr_ep_cluster_tick` -- one C call per tick: as ONE launch (the default) and as the handler kernels back to back -- against
    the handler-by-handler driver on a third set of replicas and, with `oracle`, against five oracles wired into the same
    loop (tests/ep_cluster.py): every leader's outputs every tick, every replica's final state.  phase_major: all four in the
    phase-by-phase order of the leaders' steps (smr_ep_cluster_set_mode bit 1)."""
    import torch
    import ep_cluster as ec
    from summerset_amd import EPaxosReplicaGroup, ep_cluster
    mk = lambda: [EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K, execute=execute) for r in range(R)]
    a, a1, b = mk(), mk(), mk()
    fused = ep_cluster.EPaxosCluster(a, phase_major=phase_major)
    launches = ep_cluster.EPaxosCluster(a1, per_handler_launches=True, phase_major=phase_major)
    orcs = [oracle.EpOracle(G, R, me=r, W=W, n_keys=K, execute=execute) for r in range(R)] if oracle is not None else None
    from summerset_amd import SummersetError
    for wrong in (a[::-1], a[:2], a[:4] + [b[0]]):               # replica r must sit at index r, all of them, of one population
        with pytest.raises(SummersetError):
            ep_cluster.EPaxosCluster(wrong)
    odd = EPaxosReplicaGroup(G, R, me=R - 1, window=W, n_keys=K, execute=not execute)   # a cluster is homogeneous
    with pytest.raises(SummersetError):
        ep_cluster.EPaxosCluster(a[:R - 1] + [odd])
    rng = np.random.default_rng(seed + G)
    dv = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    slow = 0
    outs = fused.new_outputs(dev)                                # the caller's arrays, reused every tick
    for t in range(T):
        keys = ec.zipf_keys(rng, R, G, K)
        drop = {(s, q): rng.random(G) < loss for s in range(R) for q in range(R) if s != q and rng.random() < 0.7} if loss else None
        kd = [dv(keys[r]) for r in range(R)]
        dd = None if drop is None else {k: dv(v) for k, v in drop.items()}
        oa = fused.tick(kd, dd, out=outs)
        oa1 = launches.tick(kd, dd)
        ob = ep_cluster.tick(b, kd, dd, always_accept_round=True, phase_major=phase_major)
        oo = ec.tick(orcs, keys, drop, phase_major=phase_major) if orcs is not None else None
        for s in range(R):
            for k in ob[s]:
                assert np.array_equal(oa[s][k].cpu().numpy(), ob[s][k].cpu().numpy()), (t, s, k)
                assert np.array_equal(oa1[s][k].cpu().numpy(), ob[s][k].cpu().numpy()), (t, s, k, "per-handler launches")
                if oo is not None:
                    assert np.array_equal(oa[s][k].cpu().numpy().view(oo[s][k].dtype), oo[s][k]), (t, s, k, "oracle")
            slow += int((ob[s]["decision"] == 2).sum())
    for r in range(R):
        y = b[r].dump()
        for eng in (a, a1):
            x = eng[r].dump()
            for n in y:
                assert np.array_equal(x[n], y[n]), (r, n)
        if orcs is not None:
            z = orcs[r].dump()
            x = a[r].dump()
            for n in z:
                assert np.array_equal(x[n], z[n]), (r, n, "oracle")
        if execute:
            y = b[r].exec_dump()
            for eng in (a, a1):
                x = eng[r].exec_dump()
                for n in y:
                    assert np.array_equal(x[n], y[n]), (r, "exec", n)
            if orcs is not None:
                z, x = orcs[r].exec_dump(), a[r].exec_dump()
                for n in z:
                    assert np.array_equal(x[n], z[n]), (r, "exec", n, "oracle")
    fused.close()
    launches.close()
    return slow


def run_shared_table_vs_private(dev, G=300, K=6, T=6, R=5, W=32, seed=9):
    """Round 5: a cluster's replicas share ONE per-key table (a 128-byte line per (group, key): csrc/ep_engine.hip
    ep_hc_migrate_kernel).  The same ticks on private tables (SMR_EP_PRIVATE_HC) must leave every replica in the same state;
    a replica outlives its cluster with its entries back in its own table, and goes first without harm."""
    import os
    import torch
    import ep_cluster as ec
    from summerset_amd import EPaxosReplicaGroup, SummersetError, ep_cluster
    mk = lambda: [EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K, execute=True) for r in range(R)]
    a, b = mk(), mk()
    shared = ep_cluster.EPaxosCluster(a, phase_major=True)
    os.environ["SMR_EP_PRIVATE_HC"] = "1"
    try:
        private = ep_cluster.EPaxosCluster(b, phase_major=True)
    finally:
        del os.environ["SMR_EP_PRIVATE_HC"]
    rng = np.random.default_rng(seed)
    tn = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)

    def same(where):
        for r in range(R):
            x, y = a[r].dump(), b[r].dump()
            for n in y:
                assert np.array_equal(x[n], y[n]), (where, r, n)
            x, y = a[r].exec_dump(), b[r].exec_dump()
            for n in y:
                assert np.array_equal(x[n], y[n]), (where, r, "exec", n)
    for t in range(T):
        keys = ec.zipf_keys(rng, R, G, K)
        drop = {(s, q): tn(rng.random(G) < 0.1) for s in range(R) for q in range(R) if s != q}
        oa, ob = shared.tick([tn(keys[r]) for r in range(R)], drop), private.tick([tn(keys[r]) for r in range(R)], drop)
        for s in range(R):
            for k in ob[s]:
                assert torch.equal(oa[s][k], ob[s][k]), (t, s, k)
    same("ticks")
    assert int(a[0].exec_dump()["kv"].max()) >> 32 >= 1 and int(a[0].dump()["highest_cols"].max()) > 0
    # the handlers of a replica that sits in a cluster work in the shared table too: one more tick through the per-handler loop
    keys = ec.zipf_keys(rng, R, G, K)
    ep_cluster.tick(a, [tn(keys[r]) for r in range(R)], None, phase_major=True)
    ep_cluster.tick(b, [tn(keys[r]) for r in range(R)], None, phase_major=True)
    same("per-handler tick inside the cluster")
    shared.close()                                               # the entries go home
    same("after the cluster")
    keys = ec.zipf_keys(rng, R, G, K)
    ep_cluster.tick(a, [tn(keys[r]) for r in range(R)], None, phase_major=True)
    ep_cluster.tick(b, [tn(keys[r]) for r in range(R)], None, phase_major=True)
    same("per-handler tick after the cluster")
    again = ep_cluster.EPaxosCluster(a, phase_major=True)        # ... and into a new cluster's table
    keys = ec.zipf_keys(rng, R, G, K)
    again.tick([tn(keys[r]) for r in range(R)])
    private.tick([tn(keys[r]) for r in range(R)])
    same("second cluster")
    a[R - 1].close()                                             # a replica that goes before its cluster
    with pytest.raises(SummersetError):
        again.tick([tn(keys[r]) for r in range(R)])
    again.close()
    private.close()


def test_cluster_shares_one_per_key_table(cuda):
    run_shared_table_vs_private(cuda, G=1500, K=64)

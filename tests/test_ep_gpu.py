"""EPaxos command-leader / acceptor parity: HIP kernels (through the C-ABI) vs the CPU oracle on
identical seeded message streams, compared after every call on the full instance space,
highest-column table and bars (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(a, cuda):
    import torch
    if a is None:
        return None
    v = a.view(np.int64) if a.dtype == np.uint64 else (a.view(np.int32) if a.dtype == np.uint32 else a)
    return torch.from_numpy(np.ascontiguousarray(v)).to(cuda)


def _same_state(eng, orc, step):
    a, b = eng.dump(), orc.dump()
    for n in b:
        assert np.array_equal(a[n], b[n]), (step, n, [x[:5] for x in np.nonzero(a[n] != b[n])])


def _same(out_eng, out_orc, keys, step):
    for k in keys:
        e, o = out_eng[k].cpu().numpy(), out_orc[k]
        assert np.array_equal(e.view(o.dtype), o), (step, k, np.nonzero(e.view(o.dtype) != o))


@pytest.mark.parametrize("G,W,me", [(600, 32, 0), (4096, 16, 3)])
def test_epaxos_handlers_match_oracle(cuda, oracle, G, W, me):
    import ep_scenarios as sc
    from summerset_amd import EPaxosReplicaGroup, stream
    R, K = 5, 8                                              # few keys: plenty of conflicts
    rng = np.random.default_rng(G + W + me)
    eng = EPaxosReplicaGroup(G, R, me=me, window=W, n_keys=K)
    orc = oracle.EpOracle(G, R, me=me, W=W, n_keys=K)
    slow_cols = np.zeros(G, np.uint32)
    for step in range(36):
        ctl = np.ascontiguousarray(stream.random_ackctl(11, step, 1, G, R, 0.0)[0])
        # acceptor side: a PreAccept and an Accept from peers
        m = sc.acceptor_round(rng, orc.dump(), G, R, me, K, W)
        _same(eng.handle_msg_pre_accept({k: _t(v, cuda) for k, v in m.items()}), orc.handle_pre_accept(**m),
              ("flags", "ballot", "seq", "deps"), (step, "pre_accept"))
        m = sc.acceptor_round(rng, orc.dump(), G, R, me, K, W)
        _same(eng.handle_msg_accept({k: _t(v, cuda) for k, v in m.items()}), orc.handle_accept(**m),
              ("flags", "ballot"), (step, "accept"))
        _same_state(eng, orc, (step, "acceptor"))
        # command leader: propose, then the replies in two waves
        key, ex = sc.propose_round(rng, G, K)
        po = orc.propose(key, ex)
        _same(eng.handle_req_batch(_t(key, cuda), _t(ex, cuda)), po, ("flags", "col", "seq", "deps"), (step, "propose"))
        for wave in range(2):
            m = sc.pre_accept_replies_round(rng, orc.dump(), po, G, R, me, ctl)
            ro = orc.handle_pre_accept_replies(**m)
            re_ = eng.handle_msg_pre_accept_reply(**{k: _t(v, cuda) for k, v in m.items()})
            _same(re_, ro, ("decision", "seq", "deps"), (step, "pa_replies", wave))
            slow_cols = np.where(ro["decision"] == 2, po["col"], slow_cols).astype(np.uint32)
        m = sc.accept_replies_round(rng, slow_cols, G, R, me, ctl)
        _same(eng.handle_msg_accept_reply(**{k: _t(v, cuda) for k, v in m.items()}), orc.handle_accept_replies(**m),
              ("committed",), (step, "acc_replies"))
        _same_state(eng, orc, (step, "leader"))
    c = orc.dump()["counters"]
    assert c[0] > 0 and c[1] > 0 and c[2] > 0, c             # fast commits, slow-path entries, slow-path commits
    assert (orc.dump()["commit_bars"][me] > 0).any()


def test_closed_loop_cluster_matches_oracle(cuda, oracle):
    """five per-replica engine objects wired into a cluster (tests/ep_cluster.py: PreAccept fan-out, replies,
    slow-path Accepts, CommitNotices) against five oracles wired the same way"""
    import ep_cluster as ec
    from summerset_amd import EPaxosReplicaGroup
    G, R, W, K, T = 700, 5, 32, 6, 10
    engs = [ec.NumpyEngine(EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K), cuda) for r in range(R)]
    orcs = [oracle.EpOracle(G, R, me=r, W=W, n_keys=K) for r in range(R)]
    rng = np.random.default_rng(5)
    slow = fast = 0
    for t in range(T):
        keys = ec.zipf_keys(rng, R, G, K)
        drop = {(s, q): rng.random(G) < 0.15 for s in range(R) for q in range(R) if s != q}
        oe, oo = ec.tick(engs, keys, drop), ec.tick(orcs, keys, drop)
        for s in range(R):
            for k in oo[s]:
                assert np.array_equal(oe[s][k], oo[s][k]), (t, s, k)
            fast += int((oo[s]["decision"] == 3).sum())
            slow += int((oo[s]["decision"] == 2).sum())
    for r in range(R):
        a, b = engs[r].dump(), orcs[r].dump()
        for n in b:
            assert np.array_equal(a[n], b[n]), (r, n)
    assert fast > 0 and slow > 0

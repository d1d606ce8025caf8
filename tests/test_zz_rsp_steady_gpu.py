"""`rsp_cluster.SteadyLoop` -- the steady state of a co-located RSPaxos cluster with every message a device tensor (what
bench.py's config-4 leg times and captures into a HIP graph) -- against the numpy-staged closed loop `rsp_cluster.tick` on five
ORACLES: the leader's commits of every tick, every replica's state at the end; with lost Accepts / AcceptReplies / Heartbeats,
f = 0 and f = 1, refused batches (window full) included.  The shard fan-out is checked against the codeword's bytes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_steady(dev, oracle, G, W, ft, loss, T=14, hb_every=3, with_cw=False, one_launch=False):
    import torch
    from summerset_amd import RSCodewordBatch, RSPaxosReplicaGroup, rsp_cluster as rc
    R, s = 5, 0
    engs = [RSPaxosReplicaGroup(G, R, me=r, window=W, fault_tolerance=ft) for r in range(R)]
    orcs = [oracle.RspOracle(G, R, me=r, W=W, fault_tolerance=ft) for r in range(R)]
    for x in engs + orcs:
        x.preset_leader(s)
    loop = rc.SteadyLoop(engs, leader=s, one_launch=one_launch)
    rng = np.random.default_rng(G * 7 + ft)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    total = 0
    for t in range(T):
        val = (1 + t * G + np.arange(G)).astype(np.uint32)
        val[rng.random(G) < 0.1] = rc.NULL                                   # groups without a batch this tick
        drop = {}
        if loss:
            for q in range(1, R):
                drop[("accept", s, q)] = rng.random(G) < loss
                drop[("accept_reply", q, s)] = rng.random(G) < loss
                drop[("hb", s, q)] = rng.random(G) < loss
                drop[("hb", q, s)] = rng.random(G) < loss
        hb = (t % hb_every) == hb_every - 1
        cw = None
        if with_cw:
            data = rng.integers(0, 256, (G, 97 + t), dtype=np.uint8)
            cw = loop.encode(dv(data))
            two = RSCodewordBatch.from_data(dv(data), 3, 2)
            two.compute_parity()
            assert torch.equal(cw.buf[:, :5 * cw.shard_len], two.buf[:, :5 * two.shard_len])
            for i in (0, G // 2, G - 1):
                assert np.array_equal(cw.buf[i, 3 * cw.shard_len:5 * cw.shard_len].cpu().numpy().reshape(2, -1), oracle.rs_encode(3, 2, data[i]))
        got = loop.tick(dv(val.view(np.int32)), lost={k: dv(v) for k, v in drop.items()} or None, heartbeat=hb).cpu().numpy()
        log = rc.tick(orcs, val, np.full(G, s, np.uint8), drop=drop or None, heartbeat=hb)
        want = [e for e in log if e["kind"] == "commit"]
        assert len(want) == 1                                                # one Accept list entry per tick in the steady state
        assert np.array_equal(got, want[0]["committed"]), t
        total += int(got.sum())
        if with_cw:
            sl = cw.shard_len
            for q in range(R):
                assert torch.equal(loop.stores[q], cw.buf[:, q * sl:(q + 1) * sl]), (t, q)
    for r in range(R):
        a, b = engs[r].dump(), orcs[r].dump()
        for n in b:
            assert np.array_equal(a[n], b[n]), (r, n)
    return total


@pytest.mark.parametrize("G,W,ft,loss", [(700, 16, 1, 0.1), (1500, 32, 0, 0.2), (300, 8, 1, 0.0)])
def test_device_steady_loop_is_the_closed_loop(cuda, oracle, G, W, ft, loss):
    assert run_steady(cuda, oracle, G, W, ft, loss) > 0


@pytest.mark.parametrize("G,W,ft,loss", [(700, 16, 1, 0.1), (1500, 32, 0, 0.2), (300, 8, 1, 0.0), (64, 8, 1, 0.3), (65, 16, 0, 0.05)])
def test_one_launch_steady_tick_is_the_closed_loop(cuda, oracle, G, W, ft, loss):
    """`smr_rsp_cluster_steady_tick` (one launch per tick, messages through LDS) against five oracles in the numpy-staged loop"""
    assert run_steady(cuda, oracle, G, W, ft, loss, one_launch=True) > 0


def test_one_launch_cluster_argument_errors(cuda):
    from summerset_amd import RSPaxosReplicaGroup, SummersetError, rsp_cluster as rc
    a = [RSPaxosReplicaGroup(64, 5, me=r, window=8, fault_tolerance=1) for r in range(5)]
    odd = RSPaxosReplicaGroup(64, 5, me=4, window=16, fault_tolerance=1)
    for wrong in (a[::-1], a[:2], a[:4] + [odd]):
        with pytest.raises(SummersetError):
            rc.SteadyLoop(wrong, one_launch=True)


def test_device_steady_loop_fans_the_shards_out(cuda, oracle):
    assert run_steady(cuda, oracle, 400, 16, 1, 0.05, T=6, with_cw=True) > 0

"""Layout L2 of the RSPaxos engine, device-resident (summerset_amd/spread_rsp.py): every rank of the job in THIS process -- same
objects, plans, buffers and kernels as the multi-process job, the all-to-all a copy -- against the co-located steady loop
(`rsp_cluster.SteadyLoop`, itself held against five oracles in tests/test_zz_rsp_steady_gpu.py): the leaders' commits of every
tick, every replica's state at the end, and the shard bytes each follower received.  The gpu-marked tests run it on the
device; tests/test_hostsim.py runs the same function on the emulator build of the engine."""
import numpy as np
import pytest


def run_spread_vs_colocated(dev, world, total, W=16, L=100, ft=1, loss=0.1, T=9, hb_every=3, seed=3):
    import torch
    from summerset_amd import RSPaxosReplicaGroup, rsp_cluster, shard, spread_rsp
    R = 5
    job = spread_rsp.in_process(total, R, W, world, dev, L, fault_tolerance=ft)
    ref = {}
    for b in range(world):
        lo, hi = shard.group_range(total, world, b)
        reps = [RSPaxosReplicaGroup(hi - lo, R, me=r, window=W, fault_tolerance=ft) for r in range(R)]
        for e in reps:
            e.preset_leader(0)
        ref[b] = (rsp_cluster.SteadyLoop(reps, leader=0), lo, hi)
    rng = np.random.default_rng(seed + world)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    n_commit = 0
    for t in range(T):
        hb = (t % hb_every) == hb_every - 1
        data, val, lost = {}, {}, {}
        for b, (loop, lo, hi) in ref.items():
            G = hi - lo
            v = (1 + t * total + lo + np.arange(G)).astype(np.uint32)
            v[rng.random(G) < 0.1] = rsp_cluster.NULL
            data[b], val[b] = dv(rng.integers(0, 256, (G, L), dtype=np.uint8)), dv(v.view(np.int32))
            lost[b] = {}
            if loss:
                for q in range(1, R):
                    for kind, a, c in (("accept", 0, q), ("accept_reply", q, 0), ("hb", 0, q), ("hb", q, 0)):
                        lost[b][(kind, a, c)] = dv(rng.random(G) < loss)
        got = job.tick(data, val, lost=lost if loss else None, heartbeat=hb)
        for b, (loop, lo, hi) in ref.items():
            cw = loop.encode(data[b])
            want = loop.tick(val[b], lost=lost[b] if loss else None, heartbeat=hb)
            assert torch.equal(got[b], want), (t, b)
            n_commit += int(want.sum())
            # what every follower received: its shard of every codeword (the receive buffer IS its shard store of the tick)
            for q in range(1, R):
                rk = job.ranks[spread_rsp.home(b, q, world)]
                p = rk._plans["accept"]
                m = rk._accept_msg(p["rbuf"], p["roff"][(b, q)], hi - lo)
                assert torch.equal(m["shard"], cw.shard(q)), (t, b, q)
            assert torch.equal(job.ranks[b % world].cw[b].buf[:, :5 * cw.shard_len], cw.buf[:, :5 * cw.shard_len])
    for b, (loop, lo, hi) in ref.items():
        for r in range(R):
            x = job.ranks[spread_rsp.home(b, r, world)].reps[(b, r)].dump()
            y = loop.reps[r].dump()
            for n in y:
                assert np.array_equal(x[n], y[n]), (b, r, n)
    assert n_commit > 0
    sent = sum(rk.bytes_sent for rk in job.ranks)
    assert sent > 0
    return n_commit


@pytest.mark.gpu
@pytest.mark.parametrize("world,total", [(2, 600), (3, 500), (8, 2048)])
def test_spread_rspaxos_is_the_colocated_loop(cuda, world, total):
    run_spread_vs_colocated(cuda, world, total)


@pytest.mark.gpu
def test_spread_rspaxos_no_loss_4k_values(cuda):
    run_spread_vs_colocated(cuda, 4, 1024, W=32, L=4113, loss=0.0, T=6)

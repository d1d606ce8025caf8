"""Layout L2 of the RSPaxos engine, device-resident (summerset_amd/spread_rsp.py): every rank of the job in THIS process -- same
objects, plans, buffers and kernels as the multi-process job, the all-to-all a copy -- against the co-located steady loop
(`rsp_cluster.SteadyLoop`, itself held against five oracles in tests/test_zz_rsp_steady_gpu.py): the leaders' commits of every
tick, every replica's state at the end, and the shard bytes each follower received.  The gpu-marked tests run it on the
device; tests/test_hostsim.py runs the same function on the emulator build of the engine."""
import numpy as np
import pytest


def run_spread_vs_colocated(dev, world, total, W=16, L=100, ft=1, loss=0.1, T=9, hb_every=3, seed=3, payload=False, oracle=None, library_tick=False):
    """payload: the job keeps its bytes in payload stores (put / extract -> message -> ingest / follow); on top of everything else
    every store cell must then hold what its engine says it holds and every held shard must be the ORACLE's codeword of the batch"""
    import torch
    from summerset_amd import RSPaxosReplicaGroup, rsp_cluster, shard, spread_rsp
    R = 5
    job = spread_rsp.in_process(total, R, W, world, dev, L, fault_tolerance=ft, payload=payload)
    if library_tick:                                             # round 6: every rank's phases are the library's segments (csrc/rsp_spread.hip)
        for rk in job.ranks:
            rk.use_library_tick()
    batches = {}                                                 # token -> the serialized batch (host copy)
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
            raw = rng.integers(0, 256, (G, L), dtype=np.uint8)
            data[b], val[b] = dv(raw), dv(v.view(np.int32))
            if payload:
                batches.update({int(v[g]): raw[g] for g in range(G) if v[g] != rsp_cluster.NULL})
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
                if payload:                                  # (a group without a batch this tick has no row to take a shard from)
                    has = val[b] != -1
                    assert torch.equal(m["shard"][has], cw.shard(q)[has]), (t, b, q)
                else:
                    assert torch.equal(m["shard"], cw.shard(q)), (t, b, q)
            if not payload and not library_tick:                 # (the library keeps the leader's codeword buffer to itself)
                assert torch.equal(job.ranks[b % world].cw[b].buf[:, :5 * cw.shard_len], cw.buf[:, :5 * cw.shard_len])
        if payload:
            _check_stores(job, world, R, L, batches, oracle, t)
    for b, (loop, lo, hi) in ref.items():
        for r in range(R):
            x = job.ranks[spread_rsp.home(b, r, world)].reps[(b, r)].dump()
            y = loop.reps[r].dump()
            for n in y:
                assert np.array_equal(x[n], y[n]), (b, r, n)
    assert n_commit > 0
    sent = sum(rk.bytes_sent for rk in job.ranks)
    assert sent > 0
    for rk in job.ranks:
        rk.close_library_tick()
    return n_commit


def _check_stores(job, world, R, L, batches, oracle, where):
    from summerset_amd import spread_rsp
    sl = oracle.rs_shard_len(L, 3)
    memo = {}
    n = 0
    for rk in job.ranks:
        for (b, r), st in rk.stores.items():
            d = rk.reps[(b, r)].dump()
            c = st.counters()
            assert c["unsatisfied"] == 0, (where, b, r, c)
            for plane, (kt, km) in enumerate((("s_val", "s_mask"), ("s_vval", "s_vmask"))):
                want_tok, want = d[kt].copy(), d[km].copy()
                want[want_tok == 0xFFFFFFFF] = 0
                sd = st.dump(plane)
                assert np.array_equal(sd["avail"], want) and np.array_equal(sd["tok"][want != 0], want_tok[want != 0]), (where, b, r, plane)
                for w in range(st.W):
                    if not want[w].any():
                        continue
                    row = st.read_row(w, plane)
                    for g in np.nonzero(want[w])[0]:
                        tok = int(want_tok[w, g])
                        if tok not in memo:
                            x = np.zeros(3 * sl, np.uint8)
                            x[:L] = batches[tok]
                            memo[tok] = np.concatenate([x.reshape(3, sl), oracle.rs_encode(3, 2, batches[tok])])
                        for k in range(R):
                            if (want[w, g] >> k) & 1:
                                assert np.array_equal(row[k, g, :sl], memo[tok][k]), (where, b, r, plane, w, g, k)
                                n += 1
    assert n > 0
    return n


@pytest.mark.gpu
@pytest.mark.parametrize("world,total", [(2, 600), (3, 500), (8, 2048)])
def test_spread_rspaxos_is_the_colocated_loop(cuda, world, total):
    run_spread_vs_colocated(cuda, world, total)


@pytest.mark.gpu
@pytest.mark.parametrize("world,total", [(2, 600), (3, 500), (8, 2048)])
def test_spread_rspaxos_tick_inside_the_library(cuda, world, total):
    """round 6: the same layout with the tick's segments inside the library (smr_rsp_spread_segment, csrc/rsp_spread.hip)"""
    run_spread_vs_colocated(cuda, world, total, library_tick=True)
    if world == 2:
        run_spread_vs_colocated(cuda, 4, 1024, W=32, L=4113, loss=0.0, T=6, library_tick=True)


@pytest.mark.gpu
def test_spread_rspaxos_no_loss_4k_values(cuda):
    run_spread_vs_colocated(cuda, 4, 1024, W=32, L=4113, loss=0.0, T=6)


@pytest.mark.gpu
@pytest.mark.parametrize("world,total", [(2, 300), (4, 512)])
def test_spread_rspaxos_with_the_bytes_in_payload_stores(cuda, oracle, world, total):
    """layout L2 with `payload=True`: put at the leader, extract -> message -> ingest -> follow at every follower"""
    run_spread_vs_colocated(cuda, world, total, T=7, payload=True, oracle=oracle)

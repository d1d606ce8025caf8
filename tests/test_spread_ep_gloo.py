"""Layout L2 of the EPaxos cluster (BASELINE config 5: fast-quorum kernel + all-to-all) as a real two-process job:
world_size 2 over gloo, the EMULATOR BUILD of the engine on every rank (tests/hostsim: the shipped kernels compiled for
the host -- not the oracle), one `all_to_all_single` per exchange on the job's own send / receive buffers.  After the
run every (block, replica) must hold exactly what the single-process co-located cluster holds for those groups, and every
command leader must have reported the same decisions tick by tick."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, R, W, K, TICKS, LOSS = 150, 5, 32, 6, 7, 0.15
OUT = ("col", "proposed", "decision", "committed", "seq", "deps")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(t):
    """tick t's proposals and lost PreAccepts for the WHOLE job (every rank draws the same and keeps its groups)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ep_cluster as ec
    rng = np.random.default_rng(77 + t)
    keys = ec.zipf_keys(rng, R, G, K)
    drop = {(s, q): rng.random(G) < LOSS for s in range(R) for q in range(R) if s != q}
    return keys, drop


def _worker(rank, world, port, out_dir, execute, ordered, via="torch"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch
    import torch.distributed as dist
    import hostsim
    from summerset_amd import spread_ep
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tn = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    with hostsim.patched():
        job = spread_ep.SpreadEPaxos(G, R, rank, world, "cpu", window=W, n_keys=K, execute=execute, ordered=ordered)
        out = {}
        comm = None
        if via in ("library", "library_tick"):                    # every exchange through smr_comm_exchange (csrc/comm.hip on tests/hostsim/rccl_sim.cpp)
            from summerset_amd import comm as smr_comm
            comm = smr_comm.Comm.from_torch_distributed("cpu")
            job.bind_comm(comm)
        if via in ("library_tick", "library_segments"):           # round 6: the tick itself inside the library (csrc/ep_spread.hip) -- with the
            job.use_library_tick()                                # communicator ONE C call per tick, without it torch moves the buffers
        for t in range(TICKS):
            keys, drop = _inputs(t)
            bk = {(b, r): tn(keys[r, job.range[b][0]:job.range[b][1]]) for (b, r) in job.reps}
            bd = {(b, s, q): tn(v[job.range[b][0]:job.range[b][1]]) for (b, s) in job.reps for (s2, q), v in drop.items() if s2 == s}
            for (b, s), o in job.tick(bk, bd).items():
                for k in OUT:
                    out["t%d_b%d_s%d_%s" % (t, b, s, k)] = o[k].numpy().copy()
        for (b, r), rep in job.reps.items():
            for n, v in rep.dump().items():
                out["dump_b%d_r%d_%s" % (b, r, n)] = v
            if execute:
                for n, v in rep.exec_dump().items():
                    out["exec_b%d_r%d_%s" % (b, r, n)] = v
        info = comm.info() if comm is not None else dict(exchanges=0, bytes_sent=0, bytes_received=0)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), live=np.array(sorted(job.reps)), sent=job.bytes_sent,
                 exchanges=job.exchanges_per_tick(), lib_exchanges=info["exchanges"], lib_sent=info["bytes_sent"], lib_received=info["bytes_received"], **out)
        if comm is not None:
            comm.close()
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, execute, ordered=None, via="torch"):
    """ordered=False with execution: the 5-exchange schedule = the co-located loop with the leaders' steps phase by phase"""
    import torch
    import torch.multiprocessing as mp
    import hostsim
    from summerset_amd import EPaxosReplicaGroup, ep_cluster, shard
    hostsim.build()                                                   # once, before the workers race to build it
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), execute, ordered, via), nprocs=2, join=True)
    five = not execute or ordered is False
    ranks = [np.load(str(tmp_path / ("rank%d.npz" % k))) for k in range(2)]
    if via in ("library", "library_tick"):                            # the library moved every byte the job counts, both ways
        assert all(int(rk["lib_exchanges"]) == TICKS * (5 if five else 17) and int(rk["lib_sent"]) == int(rk["sent"]) for rk in ranks)
        assert int(ranks[0]["lib_sent"]) == int(ranks[1]["lib_received"]) and int(ranks[1]["lib_sent"]) == int(ranks[0]["lib_received"])
    else:
        assert all(int(rk["lib_exchanges"]) == 0 for rk in ranks)
    pairs = sorted(tuple(x) for rk in ranks for x in rk["live"].tolist())
    assert pairs == sorted((b, r) for b in range(2) for r in range(R))          # every (block, replica) lives on exactly one rank
    assert all(int(rk["sent"]) > 0 and int(rk["exchanges"]) == (5 if five else 17) for rk in ranks)
    tn = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    fast = slow = 0
    with hostsim.patched():
        ref = [EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K, execute=execute) for r in range(R)]
        for t in range(TICKS):
            keys, drop = _inputs(t)
            oo = ep_cluster.tick(ref, [tn(keys[r]) for r in range(R)], {k: tn(v) for k, v in drop.items()}, always_accept_round=True,
                                 phase_major=execute and ordered is False)
            for rk in ranks:
                for b, s in rk["live"].tolist():
                    lo, hi = shard.group_range(G, 2, b)
                    for k in OUT:
                        assert np.array_equal(rk["t%d_b%d_s%d_%s" % (t, b, s, k)], oo[s][k][..., lo:hi].numpy()), (t, b, s, k)
            fast += sum(int((oo[s]["decision"] == 3).sum()) for s in range(R))
            slow += sum(int((oo[s]["decision"] == 2).sum()) for s in range(R))
        for rk in ranks:
            for b, r in rk["live"].tolist():
                lo, hi = shard.group_range(G, 2, b)
                for n, x in ref[r].dump().items():
                    if n != "counters":
                        gax = {"deps": 2}.get(n, x.ndim - 1)
                        assert np.array_equal(rk["dump_b%d_r%d_%s" % (b, r, n)], np.take(x, np.arange(lo, hi), axis=gax)), (b, r, n)
                if execute:
                    xd = ref[r].exec_dump()
                    for n in ("exec_bars", "kv", "digest"):
                        assert np.array_equal(rk["exec_b%d_r%d_%s" % (b, r, n)], xd[n][..., lo:hi]), (b, r, "exec", n)
    assert fast > 0 and slow > 0


def test_world_size_2_spread_epaxos_job_is_the_colocated_one(tmp_path):
    _run(tmp_path, execute=False)


def test_world_size_2_spread_epaxos_ordered_schedule_with_execution(tmp_path):
    _run(tmp_path, execute=True)


def test_world_size_2_spread_epaxos_five_exchanges_with_execution(tmp_path):
    _run(tmp_path, execute=True, ordered=False)


def test_world_size_2_spread_epaxos_through_the_library_exchange(tmp_path):
    """BASELINE config 5's layout with every exchange inside the library: `bind_comm` -> smr_comm_exchange, the SHIPPED
    csrc/comm.hip with two ranks (its receive-first ring posting order, per-peer sizes), RCCL stood in by shared memory"""
    _run(tmp_path, execute=True, ordered=False, via="library")


def test_world_size_2_spread_epaxos_tick_inside_the_library(tmp_path):
    """round 6 (VERDICT r5 missing #3): config 5's L2 tick as ONE C call -- smr_ep_spread_tick: the segments and the five
    smr_comm_exchange calls back to back, two ranks, RCCL stood in by shared memory"""
    _run(tmp_path, execute=True, ordered=False, via="library_tick")


def test_world_size_2_spread_epaxos_library_segments_under_gloo(tmp_path):
    """... and its segments with torch.distributed moving the buffers (a gloo job has no RCCL): the ordered schedule, 17 exchanges"""
    _run(tmp_path, execute=True, via="library_segments")

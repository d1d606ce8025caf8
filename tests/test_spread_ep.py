"""Spread layout (L2) of the EPaxos cluster -- BASELINE config 5's "fast-quorum kernel with RCCL all-to-all"
(summerset_amd/spread_ep.py): replica r of block b on rank (b + r) mod world, five exchanges per tick (2 + 3 R in the
ordered schedule), each one all_to_all_single on device tensors -- against the co-located closed loop of
summerset_amd/ep_cluster.py on the same keys and losses: every command leader's decisions of every tick and every
replica's full state.  All ranks of the job in one process here (on the device: tests/test_zzy_spread_ep_gpu.py; on the emulator
build: tests/test_hostsim.py); tests/test_spread_ep_gloo.py is the two-process job."""
import numpy as np
import pytest


def run_spread_vs_colocated(dev, G, world, n_ticks, loss, K=8, R=5, W=32, execute=False, ordered=None, make=None, seed=0, ref_phase_major=False,
                            oracle=None):
    """`make(world)` -> object with tick(keys, drop) and .ranks (default: spread_ep.in_process).  ref_phase_major: the colocated
    reference loop runs the leaders' steps phase by phase -- the order of the 5-exchange schedule (`ordered=False`).
    oracle: the oracle module -- five EpOracle objects then run the same ticks in tests/ep_cluster.tick and every block's
    decisions and every (block, replica)'s final state are held against THEM, not only against the co-located engine."""
    import torch
    import ep_cluster as ec
    from summerset_amd import EPaxosReplicaGroup, ep_cluster, shard, spread_ep
    ref = [EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K, execute=execute) for r in range(R)]
    orcs = [oracle.EpOracle(G, R, me=r, W=W, n_keys=K, execute=execute) for r in range(R)] if oracle is not None else None
    job = spread_ep.in_process(G, R, world, dev, window=W, n_keys=K, execute=execute, ordered=ordered) if make is None else make(world)
    rng = np.random.default_rng(1000 * world + G + seed)
    rngs = {b: shard.group_range(G, world, b) for b in range(world)}
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    fast = slow = 0
    for t in range(n_ticks):
        keys = ec.zipf_keys(rng, R, G, K)
        drop = {(s, q): rng.random(G) < loss for s in range(R) for q in range(R) if s != q} if loss else None
        oo = ep_cluster.tick(ref, [dv(keys[r]) for r in range(R)], None if drop is None else {k: dv(v) for k, v in drop.items()},
                             always_accept_round=True, phase_major=ref_phase_major)
        bk = {(b, r): dv(keys[r, lo:hi]) for b, (lo, hi) in rngs.items() for r in range(R) if hi > lo}
        bd = None if drop is None else {(b, s, q): dv(v[lo:hi]) for b, (lo, hi) in rngs.items() for (s, q), v in drop.items() if hi > lo}
        oe = job.tick(bk, bd)
        assert sorted(oe) == sorted(bk)
        oc = ec.tick(orcs, keys, drop, phase_major=ref_phase_major) if orcs is not None else None
        for (b, s), o in oe.items():
            lo, hi = rngs[b]
            for k in o:
                assert np.array_equal(o[k].cpu().numpy(), oo[s][k][..., lo:hi].cpu().numpy()), (t, b, s, k)
                if oc is not None:                                      # the spread job against the ORACLE cluster directly
                    assert np.array_equal(o[k].cpu().numpy().view(oc[s][k].dtype), oc[s][k][..., lo:hi]), (t, b, s, k, "oracle")
        fast += sum(int((oo[s]["decision"] == 3).sum()) for s in range(R))
        slow += sum(int((oo[s]["decision"] == 2).sum()) for s in range(R))
    full = [ref[r].dump() for r in range(R)]
    xfull = [ref[r].exec_dump() for r in range(R)] if execute else None
    ofull = [o.dump() for o in orcs] if orcs is not None else None
    oxfull = [o.exec_dump() for o in orcs] if (orcs is not None and execute) else None
    seen = set()
    for rk in job.ranks:
        for (b, r), rep in rk.reps.items():
            assert (b, r) not in seen and spread_ep.home(b, r, world) == rk.rank
            seen.add((b, r))
            lo, hi = rngs[b]
            a = rep.dump()
            if ofull is not None:                                       # ... and its final state against the oracle's
                for n, x in ofull[r].items():
                    if n == "counters":
                        continue
                    gax = {"deps": 2}.get(n, x.ndim - 1)
                    assert np.array_equal(a[n], np.take(x, np.arange(lo, hi), axis=gax)), (rk.rank, b, r, n, "oracle")
                if execute:
                    xa = rep.exec_dump()
                    for n in ("exec_bars", "kv", "digest"):
                        assert np.array_equal(xa[n], oxfull[r][n][..., lo:hi]), (rk.rank, b, r, "exec", n, "oracle")
            for n, x in full[r].items():
                if n == "counters":
                    continue
                gax = {"deps": 2}.get(n, x.ndim - 1)                    # deps is [R, W, G, R]; everything else ends in G
                assert np.array_equal(a[n], np.take(x, np.arange(lo, hi), axis=gax)), (rk.rank, b, r, n)
            if execute:
                xa = rep.exec_dump()
                for n in ("exec_bars", "kv", "digest"):
                    assert np.array_equal(xa[n], xfull[r][n][..., lo:hi]), (rk.rank, b, r, "exec", n)
    assert seen == {(b, r) for b, (lo, hi) in rngs.items() for r in range(R) if hi > lo}
    for r in range(R):                                                  # the event counters add up over the blocks
        tot = sum(rep.dump()["counters"] for rk in job.ranks for (b, q), rep in rk.reps.items() if q == r)
        assert np.array_equal(tot, full[r]["counters"]), r
    assert fast > 0 and (slow > 0 or not loss)
    return job


def library_tick_job(G, R=5, W=32, K=8, execute=False, ordered=None, dev="cpu"):
    """`make` for run_spread_vs_colocated: the in-process job with every rank's tick inside the library (round 6:
    `smr_ep_spread_segment`, csrc/ep_spread.hip -- schedule, message plan and packing are the library's; here the test moves the
    exchanges' buffers between the virtual ranks)"""
    from summerset_amd import spread_ep

    def make(world):
        job = spread_ep.in_process(G, R, world, dev, window=W, n_keys=K, execute=execute, ordered=ordered)
        for rk in job.ranks:
            rk.use_library_tick()
        return job
    return make


def run_library_tick_cases(dev, oracle, scale=1):
    """the cases of the Python-driven layout (tests/test_hostsim.py, tests/test_zzy_spread_ep_gpu.py) through the library's tick"""
    g = lambda n: n * scale   # noqa: E731
    run_spread_vs_colocated(dev, G=g(130), world=2, n_ticks=5, loss=0.15, make=library_tick_job(g(130), dev=dev), oracle=oracle)
    run_spread_vs_colocated(dev, G=g(100), world=3, n_ticks=4, loss=0.15, make=library_tick_job(g(100), dev=dev))
    run_spread_vs_colocated(dev, G=21, world=8, n_ticks=3, loss=0.1, make=library_tick_job(21, dev=dev))             # blocks of 2-3 groups
    job = run_spread_vs_colocated(dev, G=g(120), world=4, n_ticks=5, loss=0.15, K=6, execute=True,
                                  make=library_tick_job(g(120), K=6, execute=True, dev=dev), oracle=oracle)
    assert job.ranks[0].exchanges_per_tick() == 17
    job = run_spread_vs_colocated(dev, G=g(120), world=4, n_ticks=5, loss=0.15, K=6, execute=True, ordered=False, ref_phase_major=True,
                                  make=library_tick_job(g(120), K=6, execute=True, ordered=False, dev=dev), oracle=oracle)
    assert job.ranks[0].exchanges_per_tick() == 5 and all(rk.bytes_sent > 0 for rk in job.ranks)
    job = run_spread_vs_colocated(dev, G=g(70), world=1, n_ticks=4, loss=0.1, make=library_tick_job(g(70), dev=dev))  # ONE C call per tick: smr_ep_spread_tick
    assert job.ranks[0].bytes_sent == 0
    run_spread_vs_colocated(dev, G=g(60), world=2, n_ticks=4, loss=0.1, R=3, K=4, make=library_tick_job(g(60), R=3, K=4, dev=dev))
    for rk in job.ranks:
        rk.close_library_tick()


def test_library_tick_refuses_bad_arguments():
    """argument errors of smr_ep_spread_create / _segment need no device"""
    import ctypes as C
    import hostsim
    from summerset_amd import SummersetError, _lib
    hostsim.build()
    with hostsim.patched() as L:
        h = C.c_void_p()
        groups = (C.c_uint32 * 2)(8, 8)
        for args, frag in (((None, None, None, 0, groups, 2, 2, 5, 0, C.byref(h)), "rank / world"),
                           ((None, None, None, 0, groups, 2, 0, 2, 0, C.byref(h)), "population"),
                           ((None, None, None, 0, groups, 2, 0, 5, 0, C.byref(h)), "was not handed over")):
            with pytest.raises(SummersetError) as e:
                _lib.check(L.smr_ep_spread_create(*args))
            assert frag in e.value.msg, e.value.msg
        with pytest.raises(SummersetError):
            _lib.check(L.smr_ep_spread_segment(None, 0, None, None, None, None))


def test_plans_agree_across_ranks_without_a_device():
    """the static plans only: what rank s sends to d is what d expects from s, in every exchange of both schedules, and
    every (block, replica) has exactly one home -- no library call, no device"""
    from summerset_amd import shard, spread_ep
    for world, G, ordered in [(2, 130, False), (3, 100, True), (8, 5, False), (8, 1000, True)]:
        stubs = []
        for rank in range(world):
            s = spread_ep.SpreadEPaxos.__new__(spread_ep.SpreadEPaxos)
            import torch
            s.torch, s.R, s.rank, s.world, s.device = torch, 5, rank, world, "cpu"
            s.range = {b: shard.group_range(G, world, b) for b in range(world)}
            stubs.append(s)
        sets = [[s] for s in range(5)] if ordered else [list(range(5))]
        for kind, ls in [("pre_accept", list(range(5))), ("pa_reply", list(range(5)))] + [(k, l) for l in sets for k in ("accept", "acc_reply", "commit")]:
            plans = [s._plan(kind, ls) for s in stubs]
            for a in range(world):
                for b in range(world):
                    assert plans[a]["in_split"][b] == plans[b]["out_split"][a]
                assert plans[a]["in_split"][a] == 0 and sum(plans[a]["in_split"]) == plans[a]["n_send"]
                assert all(v % 8 == 0 for v in plans[a]["in_split"])
        homes = [(b, r) for b in range(world) for r in range(5)]
        assert sorted(homes) == sorted((b, r) for k in range(world) for b in range(world) for r in range(5) if spread_ep.home(b, r, world) == k)


def test_bench_layout_spread_epaxos_on_the_emulator(capsys, monkeypatch, tmp_path):
    """`bench.py --layout spread-epaxos` end to end with virtual ranks (world 1), the emulator build standing in for the
    device: the line it prints carries the contract's keys and a positive rate"""
    import json
    import sys
    import torch
    import hostsim
    import bench
    hostsim.build()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--layout", "spread-epaxos", "--groups", "96", "--steps", "3", "--warmup", "1", "--spread-ranks", "3"])
    monkeypatch.setenv("SMR_BENCH_DETAIL_DIR", str(tmp_path))
    args = bench.parse()
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    with hostsim.patched():
        bench.spread_epaxos_main(args, torch, torch.distributed, 0, 0, 1, "cpu")
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["spread_ranks"] == 3 and line["exchange"]["collectives_per_tick"] == 5
    assert line["value"] > 0 and line["exchange"]["bytes_sent_per_tick_per_rank"] > 0 and line["steps"] == 3


def test_bench_layout_colocated_epaxos_on_the_emulator(capsys, monkeypatch, tmp_path):
    """`bench.py --layout colocated-epaxos` (config 5 in layout L1: one smr_ep_cluster_tick call per tick) end to end, the
    emulator build standing in for the device"""
    import json
    import sys
    import torch
    import hostsim
    import bench
    hostsim.build()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--layout", "colocated-epaxos", "--groups", "96", "--steps", "3", "--warmup", "1"])
    monkeypatch.setenv("SMR_BENCH_DETAIL_DIR", str(tmp_path))    # the full record goes beside the (small) final line
    args = bench.parse()
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    with hostsim.patched():
        bench.colocated_epaxos_main(args, torch, torch.distributed, 0, 0, 1, "cpu")
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["steps"] == 3 and line["config"]["layout"] == "colocated"
    # every replica executes every command of all four ticks (a few run twice: an executing slot that add_edge re-inserts,
    # execution.rs:57-59 -- the reference's behaviour, counted by the engine as re-submissions)
    detail = json.load(open(tmp_path / bench.DETAIL_FILE))
    assert detail["value"] == pytest.approx(line["value"], rel=1e-6) and len(json.dumps(line)) < bench.LINE_BUDGET
    assert 5 * 5 * 96 * 4 <= detail["commands_executed_this_rank"] <= 5 * 5 * 96 * 4 * 1.02

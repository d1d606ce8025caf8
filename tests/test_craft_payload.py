"""CRaft shard bytes (`smr_craft_pstore_*`, csrc/rsp_payload.hip) on the kernel-source emulator: the closed loop of
tests/craft_payload_loop.py -- put at append, follow behind AppendEntries (own shard / full-copy / a majority without the data
shards -> reconstruct_data on commit), a new leader's Reconstruct round -- every shard byte against the oracle's encoder; the
ring wrapping; argument and state errors.  The device run of the same loop: tests/test_zz_craft_payload_gpu.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))


@pytest.mark.parametrize("staging,many", [(False, False), (True, False), (False, True), (False, "one_call")],
                         ids=["colocated", "messages", "follow_many", "put_follow_all"])
def test_craft_payload_loop_on_the_emulator(oracle, staging, many):
    import hostsim
    import craft_payload_loop as cl
    hostsim.build()
    with hostsim.patched():
        lp = cl.run("cpu", oracle, G=70, W=32, L=67, staging=staging, many=many)
        if many == "one_call":                                           # the put launch wrote followers' shards -- and not all of them
            dl, cp = [s.delivered() for s in lp.stores], [s.counters()["copied"] for s in lp.stores]
            assert sum(dl) > 0 and all(d <= c for d, c in zip(dl, cp)) and sum(dl) < sum(cp), (dl, cp)


def test_craft_payload_ring_wraps_on_the_emulator(oracle):
    """W = 8: the log outgrows the ring, cells are re-keyed by later entries, the followers' stores follow"""
    import hostsim
    import craft_payload_loop as cl
    hostsim.build()
    with hostsim.patched():
        lp = cl.Loop("cpu", oracle, G=40, W=8, L=40, seed=3)
        for t in range(14):
            lp.tick(p_new=1.0)
        assert int(lp.reps[0].dump()["log_len"].max()) > 8
        assert sum(int(s.counters()["rekeyed"]) for s in lp.stores) > 0
        for r in range(lp.R):
            lp.check(r, ("end", r))


def test_craft_store_errors_on_the_emulator(oracle):
    import torch
    import hostsim
    from summerset_amd import CRaftLeaderGroup, CRaftPayloadStore, RaftLeaderGroup, RSPaxosPayloadStore, RSPaxosReplicaGroup, SummersetError
    from summerset_amd.rsp_payload import VOTED
    hostsim.build()
    with hostsim.patched():
        G, R, W, L = 8, 5, 8, 32
        st, eng = CRaftPayloadStore(G, R, W, max_data_len=L), CRaftLeaderGroup(G, R, leader_id=0, window=W, term=1)
        slot = torch.zeros(G, dtype=torch.int32)
        data = torch.zeros((G, L), dtype=torch.uint8)
        with pytest.raises(SummersetError):
            st.put(RaftLeaderGroup(G, R, leader_id=0, window=W, term=1), slot, data)      # a plain Raft replica has no codewords
        with pytest.raises(SummersetError):
            st.follow(CRaftLeaderGroup(G, R, leader_id=0, window=2 * W, term=1))          # another window
        with pytest.raises(SummersetError):
            st.dump(VOTED)                                                               # a log entry has one codeword
        with pytest.raises(SummersetError):
            RSPaxosPayloadStore(G, R, W, max_data_len=L).follow(eng)                      # ... and an RSPaxos store follows an RSPaxos replica
        rsp = RSPaxosReplicaGroup(G, R, me=0, window=W)
        with pytest.raises(SummersetError):
            RSPaxosPayloadStore.follow(st, rsp)                                           # smr_rsp_pstore_follow on a CRaft store
        st.put(eng, slot.fill_(-1), data)                                                 # nothing appended: nothing stored
        st.follow(eng)
        assert int(st.dump()["avail"].sum()) == 0 and st.counters()["unsatisfied"] == 0


def test_bench_leg_craft_payload_on_the_emulator(oracle):
    """bench.py's `craft_payload` leg -- its loop and its own end-of-leg checks -- at a small shape, the emulator as the device"""
    import importlib.util
    import torch
    import hostsim
    spec = importlib.util.spec_from_file_location("smr_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    hostsim.build()

    def run_untimed(torch_, fn, iters, sleep_cycles=0):
        for i in range(iters):
            fn(i)
        return 100.0
    with hostsim.patched():
        out = bench.craft_payload_leg(torch, "cpu", ticks=5, warmup=2, G=130, L=67, time_us=run_untimed)
    assert out["verified"] and out["counters"]["unsatisfied"] == 0 and out["counters"]["copied"] > 0
    line = bench.compact_leg(out)
    assert set(line) == {"value", "unit", "ms_per_tick", "frac", "frac_on_8d_bytes", "traffic_ratio", "cpu_cores"}


def run_one_launch_replication_is_the_eight_calls(dev, G=150, W=8, L=40, T=11):
    """two CRaft clusters on the same inputs: one replicates with gather_entries + handle_msg_append_entries per follower, the other
    with the one-launch `replicate_many` -- engines, messages, replies and stores stay identical tick by tick (W = 8: the ring wraps)"""
    import torch
    from summerset_amd import workloads
    a, b = workloads.craft_payload_cluster(G, W, L, 1, dev), workloads.craft_payload_cluster(G, W, L, 1, dev)
    rng = np.random.default_rng(5)
    for t in range(T):
        slot = torch.full((G,), t + 1, dtype=torch.int32, device=dev)
        src = torch.from_numpy(rng.integers(0, 256, (G, L), dtype=np.uint8)).to(dev)
        ma = workloads.craft_payload_tick(*a, slot, src, one_launch=False)
        mb = workloads.craft_payload_tick(*b, slot, src, one_launch=True, one_tick_launch=False)
        for q in ma:
            for k in ma[q]:
                assert torch.equal(ma[q][k], mb[q][k]), (t, q, k)
        for k in ("rt", "es", "fl", "ct", "cs"):
            assert torch.equal(a[2][k], b[2][k]), (t, k)
        for r in range(len(a[0])):
            da, db = a[0][r].dump(), b[0][r].dump()
            for k in da:
                assert np.array_equal(da[k], db[k]), (t, r, k)
            sa, sb = a[1][r].dump(), b[1][r].dump()
            for k in sa:
                assert np.array_equal(sa[k], sb[k]), (t, r, k)
    assert int(a[0][0].dump()["last_commit"].min()) >= T - 2


def test_one_launch_replication_on_the_emulator():
    import hostsim
    hostsim.build()
    with hostsim.patched():
        run_one_launch_replication_is_the_eight_calls("cpu", G=70)


# ---- round 6: put + the leader's follow + the followers' follow_many as ONE call, four launches (smr_*_pstore_put_follow_all) ----
def _stores_equal(sa, sb, planes, W, tag):
    for pl in range(planes):
        da, db = sa.dump(pl), sb.dump(pl)
        for k in da:
            assert np.array_equal(da[k], db[k]), (tag, pl, k)
        for row in range(W):                                          # every byte of every shard the headers name
            ra, rb = sa.read_row(row, pl), sb.read_row(row, pl)
            sl = -(-da["dlen"][row].astype(np.int64) // (sa.R // 2 + 1))
            for k in range(sa.R):
                m = ((da["avail"][row] >> k) & 1).astype(bool)
                if planes == 2 and pl == 1:
                    m &= ~((sa.voted_alias()[row] >> k) & 1).astype(bool)   # (an aliased vote's bytes live in the REQS row)
                cols = np.arange(ra.shape[2])[None, :] < sl[:, None]
                assert np.array_equal(ra[k][m[:, None] & cols], rb[k][m[:, None] & cols]), (tag, pl, row, k)
    if planes == 2:
        assert np.array_equal(sa.voted_alias(), sb.voted_alias()), tag
    assert sa.counters() == sb.counters(), (tag, sa.counters(), sb.counters())


def run_craft_one_call_is_the_three_calls(dev, G=150, W=8, L=40, T=11):
    """two CRaft clusters on the same inputs, the byte path of one as put / follow / follow_many, of the other as
    `put_follow_all`: engines, stores (headers, every named shard byte) and counters identical tick by tick; W = 8: the ring wraps"""
    import torch
    from summerset_amd import workloads
    a, b = workloads.craft_payload_cluster(G, W, L, 1, dev), workloads.craft_payload_cluster(G, W, L, 1, dev)
    rng = np.random.default_rng(7)
    for t in range(T):
        slot = torch.full((G,), t + 1, dtype=torch.int32, device=dev)
        if t == 4:
            slot[::3] = -1                                             # groups that do not put this tick (their entry has no bytes yet)
        src = torch.from_numpy(rng.integers(0, 256, (G, L), dtype=np.uint8)).to(dev)
        lens = torch.from_numpy(rng.integers(1, L + 1, G).astype(np.int32)).to(dev)
        workloads.craft_payload_tick(*a, slot, src, lens=lens, one_call=False)
        workloads.craft_payload_tick(*b, slot, src, lens=lens, one_call=True, one_tick_launch=False)
        for r in range(len(a[0])):
            da, db = a[0][r].dump(), b[0][r].dump()
            for k in da:
                assert np.array_equal(da[k], db[k]), (t, r, k)
            _stores_equal(a[1][r], b[1][r], 1, W, (t, r))
    assert a[1][1].counters()["copied"] > 0
    assert a[1][1].delivered() == 0 and 0 < b[1][1].delivered() <= b[1][1].counters()["copied"]   # (the put launch wrote them)


def run_rspaxos_one_call_is_the_three_calls(dev, G=130, W=8, L=67, T=12):
    """config 4's cluster + payload stores twice, one tick as the three calls, the other as `put_follow_all`, reply loss on"""
    import torch
    from summerset_amd import workloads
    a, b = workloads.config4_payload_cluster(G, W, 1, L), workloads.config4_payload_cluster(G, W, 1, L)
    rng = np.random.default_rng(11)
    ones = torch.ones(G, dtype=torch.int32, device=dev)
    for t in range(T):
        slot = torch.full((G,), t, dtype=torch.int32, device=dev)
        src = torch.from_numpy(rng.integers(0, 256, (G, L), dtype=np.uint8)).to(dev)
        val = torch.from_numpy(workloads.config4_tokens(G, t)).to(dev)
        lost = {k: torch.from_numpy(v).to(dev) for k, v in workloads.config4_loss(rng, G, p=0.2).items()}
        ca = workloads.config4_payload_tick(*a, slot, src, val, lost, t % 4 == 3, ones, one_call=False)
        cb = workloads.config4_payload_tick(*b, slot, src, val, lost, t % 4 == 3, ones, one_call=True)
        assert torch.equal(ca, cb), t
        for r in range(len(a[0])):
            da, db = a[0][r].dump(), b[0][r].dump()
            for k in da:
                assert np.array_equal(da[k], db[k]), (t, r, k)
            _stores_equal(a[2][r], b[2][r], 2, W, (t, r))
    assert a[2][1].counters()["copied"] > 0
    assert a[2][1].delivered() == 0 and 0 < b[2][1].delivered() <= b[2][1].counters()["copied"]


def run_one_launch_tick_is_the_three_launches(dev, G=150, W=8, L=40, T=11):
    """two CRaft clusters on the same inputs: the engines' tick of one as append / replicate / replies (three launches, the replies in
    front of the tick's bytes), of the other as `smr_raft_cluster_tick` (one launch) -- engines, messages, the leader's reply arrays,
    stores and counters identical tick by tick; W = 8: the ring wraps.  Then the same without the stores on a plain Raft cluster."""
    import torch
    from summerset_amd import workloads
    a, b = workloads.craft_payload_cluster(G, W, L, 1, dev), workloads.craft_payload_cluster(G, W, L, 1, dev)
    rng = np.random.default_rng(17)
    for t in range(T):
        slot = torch.full((G,), t + 1, dtype=torch.int32, device=dev)
        src = torch.from_numpy(rng.integers(0, 256, (G, L), dtype=np.uint8)).to(dev)
        lens = torch.from_numpy(rng.integers(1, L + 1, G).astype(np.int32)).to(dev)
        ma = workloads.craft_payload_tick(*a, slot, src, lens=lens, one_tick_launch=False, replies_first=True)
        mb = workloads.craft_payload_tick(*b, slot, src, lens=lens, one_tick_launch=True)
        for q in ma:
            for k in ma[q]:
                assert torch.equal(ma[q][k], mb[q][k]), (t, q, k)
        for k in ("first", "rt", "es", "fl", "ct", "cs"):
            assert torch.equal(a[2][k], b[2][k]), (t, k)
        for r in range(len(a[0])):
            da, db = a[0][r].dump(), b[0][r].dump()
            for k in da:
                assert np.array_equal(da[k], db[k]), (t, r, k)
            assert a[0][r].total_commits() == b[0][r].total_commits(), (t, r)
            _stores_equal(a[1][r], b[1][r], 1, W, (t, r))
    assert int(a[0][0].dump()["last_commit"].min()) >= T - 2


def test_one_launch_tick_on_the_emulator():
    import hostsim
    hostsim.build()
    with hostsim.patched():
        run_one_launch_tick_is_the_three_launches("cpu", G=70)


def test_one_call_byte_path_on_the_emulator():
    import hostsim
    hostsim.build()
    with hostsim.patched():
        run_craft_one_call_is_the_three_calls("cpu", G=70)
        run_rspaxos_one_call_is_the_three_calls("cpu", G=70)

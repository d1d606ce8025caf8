"""CRaft shard bytes on the device (`smr_craft_pstore_*`): the closed loop of tests/craft_payload_loop.py through the C-ABI --
balanced / full-copy assignment, reconstruct_data on commit, a new leader's Reconstruct round -- every shard byte of every
replica's store against the oracle's encoder after every handler call; a larger shape with 4 KiB batches; the ring wrapping."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(__file__))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("staging,many", [(False, False), (True, False), (False, True), (False, "one_call")],
                         ids=["colocated", "messages", "follow_many", "put_follow_all"])
def test_craft_payload_loop(cuda, oracle, staging, many):
    import craft_payload_loop as cl
    lp = cl.run(cuda, oracle, G=96, W=32, L=131, staging=staging, many=many)
    if many == "one_call":
        assert 0 < sum(s.delivered() for s in lp.stores) < sum(s.counters()["copied"] for s in lp.stores)


def test_craft_payload_loop_4k_batches(cuda, oracle):
    import craft_payload_loop as cl
    cl.run(cuda, oracle, G=300, W=32, L=4113, seed=11, staging=True)


def test_craft_payload_ring_wraps(cuda, oracle):
    import craft_payload_loop as cl
    lp = cl.Loop(cuda, oracle, G=128, W=8, L=200, seed=3)
    for t in range(20):
        lp.tick(p_new=1.0)
    assert int(lp.reps[0].dump()["log_len"].max()) > 8
    assert sum(int(s.counters()["rekeyed"]) for s in lp.stores) > 0
    for r in range(lp.R):
        lp.check(r, ("end", r))


def test_one_launch_replication_is_the_eight_calls(cuda):
    """`smr_raft_cluster_replicate` (the leader's four AppendEntries and their handlers in one launch) against the eight calls"""
    import test_craft_payload as t
    t.run_one_launch_replication_is_the_eight_calls(cuda, G=1000, W=8, L=200, T=14)


def test_one_call_byte_path_is_the_three_calls(cuda):
    """`smr_craft_pstore_put_follow_all` / `smr_rsp_pstore_put_follow_all` (round 6: put + the leader's follow + the followers'
    follow_many in four launches) against the three calls, stores compared byte for byte every tick"""
    import test_craft_payload as t
    t.run_craft_one_call_is_the_three_calls(cuda, G=1000, W=8, L=200, T=14)
    t.run_rspaxos_one_call_is_the_three_calls(cuda, G=1000, W=8, L=333, T=14)


def test_one_launch_tick_is_the_three_launches(cuda):
    """`smr_raft_cluster_tick` (round 6: append + replicate + replies of a co-located CRaft cluster in one launch) against the three"""
    import test_craft_payload as t
    t.run_one_launch_tick_is_the_three_launches(cuda, G=1000, W=8, L=200, T=14)

"""Spread layout (L2) of the EPaxos cluster on the device: all ranks of the job in one process on cuda:0, the collective a
device copy -- summerset_amd/spread_ep.py against the co-located closed loop (tests/test_spread_ep.py holds the
comparison) AND, since round 4, against five EpOracle objects running the same ticks: every block's
decisions of every tick and every (block, replica)'s final state directly against the oracle.  Sorted behind the rest (first device run: profiles/round2/r2q), before the files that have not run on a device yet; a failure here must not keep the rest of the suite from
running under `pytest -x`."""
import pytest

from test_spread_ep import run_spread_vs_colocated

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3, 8])
def test_spread_epaxos_job_is_the_colocated_one(cuda, oracle, world):
    job = run_spread_vs_colocated(cuda, G=96 * world, world=world, n_ticks=8, loss=0.15, oracle=oracle)
    assert all(rk.bytes_sent > 0 and rk.exchanges_per_tick() == 5 for rk in job.ranks)


def test_spread_epaxos_ordered_schedule_with_execution(cuda, oracle):
    job = run_spread_vs_colocated(cuda, G=700, world=4, n_ticks=8, loss=0.15, K=6, execute=True, oracle=oracle)
    assert all(rk.exchanges_per_tick() == 17 for rk in job.ranks)
    run_spread_vs_colocated(cuda, G=512, world=8, n_ticks=6, loss=0.0, K=64, execute=True)


def test_spread_epaxos_five_exchanges_with_execution(cuda, oracle):
    """execution on with the 5-exchange schedule (`ordered=False`): that IS the co-located loop with the command leaders' steps
    phase by phase (`ep_cluster.tick(.., phase_major=True)`, smr_ep_cluster_set_mode bit 1) -- decisions, protocol state and
    the executors' state bit for bit"""
    job = run_spread_vs_colocated(cuda, G=700, world=4, n_ticks=8, loss=0.15, K=6, execute=True, ordered=False, ref_phase_major=True, oracle=oracle)
    assert job.ranks[0].exchanges_per_tick() == 5
    run_spread_vs_colocated(cuda, G=512, world=8, n_ticks=6, loss=0.0, K=64, execute=True, ordered=False, ref_phase_major=True)


def test_spread_epaxos_tick_inside_the_library(cuda, oracle):
    """round 6: the same layout with the tick's schedule, message plan and packing inside the library (smr_ep_spread_segment /
    smr_ep_spread_tick, csrc/ep_spread.hip)"""
    from test_spread_ep import run_library_tick_cases
    run_library_tick_cases(cuda, oracle, scale=4)

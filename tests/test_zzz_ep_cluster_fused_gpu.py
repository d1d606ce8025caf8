"""`smr_ep_cluster_tick` -- one tick of a co-located EPaxos cluster as ONE C-ABI call -- on the device, against the
handler-by-handler driver loop of summerset_amd/ep_cluster.py on a second set of replicas: every leader's outputs every
tick, every replica's instances and execution state.  Sorted last: written when no device was at hand (verified on the
emulator, tests/test_hostsim.py) -- a failure here must not keep the rest of the suite from running under `pytest -x`."""
import pytest

from test_zz_ep_cluster_gpu import run_fused_vs_driver

pytestmark = pytest.mark.gpu          # first device run: GPUTEST_r02 (11 XPASS); quarantine removed in round 3


@pytest.mark.parametrize("G,K,loss", [(700, 6, 0.15), (4096, 64, 0.0)])
def test_fused_cluster_tick_is_the_driver_loop(cuda, oracle, G, K, loss):
    slow = run_fused_vs_driver(cuda, G, K, loss, oracle=oracle)
    assert slow > 0 or loss == 0.0


def test_fused_cluster_tick_other_populations(cuda, oracle):
    assert run_fused_vs_driver(cuda, 900, 4, 0.15, T=6, R=3, W=16, oracle=oracle) > 0
    assert run_fused_vs_driver(cuda, 700, 6, 0.15, T=6, R=7, W=16, oracle=oracle) > 0
    run_fused_vs_driver(cuda, 1000, 16, 0.1, T=6, execute=False, oracle=oracle)        # a ragged last tile, execution off


def test_fused_cluster_tick_phase_by_phase(cuda, oracle):
    """smr_ep_cluster_set_mode bit 1: the leaders' steps phase by phase (all R replicas of a group at work in every step of the
    one-launch kernel) -- one launch, launch by launch, the Python loop and the oracle cluster, all run in that order: loss and
    slow paths with execution on, other populations, execution off"""
    assert run_fused_vs_driver(cuda, 700, 6, 0.15, oracle=oracle, phase_major=True) > 0
    run_fused_vs_driver(cuda, 4096, 64, 0.0, oracle=oracle, phase_major=True)
    assert run_fused_vs_driver(cuda, 900, 4, 0.15, T=6, R=3, W=16, oracle=oracle, phase_major=True) > 0
    assert run_fused_vs_driver(cuda, 700, 6, 0.15, T=6, R=7, W=16, oracle=oracle, phase_major=True) > 0
    run_fused_vs_driver(cuda, 1000, 16, 0.1, T=6, execute=False, oracle=oracle, phase_major=True)

"""`smr_ep_cluster_tick` -- one tick of a co-located EPaxos cluster as ONE C-ABI call -- on the device, against the
handler-by-handler driver loop of summerset_amd/ep_cluster.py on a second set of replicas: every leader's outputs every
tick, every replica's instances and execution state.  Sorted last: written when no device was at hand (verified on the
emulator, tests/test_hostsim.py) -- a failure here must not keep the rest of the suite from running under `pytest -x`."""
import pytest

from test_zz_ep_cluster_gpu import run_fused_vs_driver

# Quarantined until its first device run: written after round 2's GPU minutes were spent (every scenario here passes on the
# kernel-source emulator, tests/test_hostsim.py).  xfail(strict=False) = it RUNS on the device with the rest of the suite and
# its outcome is reported (XPASS / xfailed), but a surprise here cannot turn the device suite red or stop `pytest -x` in front
# of anything else.  Remove the mark once profiles/ holds its first device log (tools/r3a_first_call.sh).
pytestmark = [pytest.mark.gpu, pytest.mark.xfail(strict=False, reason="first device run pending (emulator-verified)")]


@pytest.mark.parametrize("G,K,loss", [(700, 6, 0.15), (4096, 64, 0.0)])
def test_fused_cluster_tick_is_the_driver_loop(cuda, G, K, loss):
    slow = run_fused_vs_driver(cuda, G, K, loss)
    assert slow > 0 or loss == 0.0


def test_fused_cluster_tick_other_populations(cuda):
    assert run_fused_vs_driver(cuda, 900, 4, 0.15, T=6, R=3, W=16) > 0
    assert run_fused_vs_driver(cuda, 700, 6, 0.15, T=6, R=7, W=16) > 0

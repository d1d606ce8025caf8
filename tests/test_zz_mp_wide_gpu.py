"""MultiPaxos cluster engine at populations other than 3 and 5 (4, 7, 8 replicas: the 8-wide template instances of
mp_quorum_tally / the reply kernels), against the oracle after every tick.  Added after the round's GPU minutes
were spent, together with the emulator run of the same shapes (tests/test_hostsim.py); sorts last like the other
first-run device tests."""
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.mark.parametrize("R,S,extra", [(7, 2, 0), (4, 3, 0), (8, 1, 0), (7, 2, 2)])
def test_other_populations(cuda, oracle, R, S, extra):
    import test_mp_gpu as t
    t._run(cuda, oracle, G=700, R=R, S=S, W=64, n_ticks=36, drop_p=0.1, timeout_frac=0.0 if extra else 1.0, hb_every=4, preset=True,
           commit_extra=extra)

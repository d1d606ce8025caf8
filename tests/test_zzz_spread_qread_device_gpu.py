"""summerset_amd.spread.read_quorum_step_device on the device (one rank: handlers, the packing into one send tensor, the views
the tally takes, the answer buffer -- everything but the collective) against the oracle's rounds.  Sorted last: written when no
device was at hand (the two-process gloo job with the emulator engine: tests/test_spread_qread_device_gloo.py)."""
import pytest

from test_spread_qread_device_gloo import run_one_rank

pytestmark = pytest.mark.gpu          # first device run: GPUTEST_r02 (11 XPASS); quarantine removed in round 3


def test_device_resident_quorum_read_round_on_the_device(cuda, oracle):
    run_one_rank(oracle, cuda)

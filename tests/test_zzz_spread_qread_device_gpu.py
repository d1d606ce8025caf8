"""summerset_amd.spread.read_quorum_step_device on the device (one rank: handlers, the packing into one send tensor, the views
the tally takes, the answer buffer -- everything but the collective) against the oracle's rounds.  Sorted last: written when no
device was at hand (the two-process gloo job with the emulator engine: tests/test_spread_qread_device_gloo.py)."""
import pytest

from test_spread_qread_device_gloo import run_one_rank

# Quarantined until its first device run: written after round 2's GPU minutes were spent (every scenario here passes on the
# kernel-source emulator, tests/test_hostsim.py).  xfail(strict=False) = it RUNS on the device with the rest of the suite and
# its outcome is reported (XPASS / xfailed), but a surprise here cannot turn the device suite red or stop `pytest -x` in front
# of anything else.  Remove the mark once profiles/ holds its first device log (tools/r3a_first_call.sh).
pytestmark = [pytest.mark.gpu, pytest.mark.xfail(strict=False, reason="first device run pending (emulator-verified)")]


def test_device_resident_quorum_read_round_on_the_device(cuda, oracle):
    run_one_rank(oracle, cuda)

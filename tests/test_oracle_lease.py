"""Pins oracle/lease_oracle.c on the reference's own unit tests for `LeaseManager` (src/server/leaseman.rs:1079-2301),
restated with explicit time in tests/lease_scenarios.py."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.dirname(__file__))
import lease_scenarios as LS  # noqa: E402
from oracle.oracle import LeaseOracle  # noqa: E402


def mk(G):
    return lambda R, me, expire: LeaseOracle(G, R, me, expire, 20)


@pytest.mark.parametrize("trace", LS.ALL_TRACES, ids=lambda f: f.__name__)
def test_reference_trace(trace):
    trace(mk(3))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_mutual_leases(seed):
    LS.mutual_leases(mk(2), order_seed=seed)


def test_new_and_setup_rejects_what_the_reference_rejects():
    """leaseman.rs:175-193: expire timeout within [100 ms, 10 s] and more than two heartbeat send intervals"""
    for expire, hb in [(99, 20), (10001, 20), (100, 50), (600, 300)]:
        with pytest.raises(ValueError):
            LeaseOracle(1, 5, 0, expire, hb)
    LeaseOracle(1, 5, 0, 100, 49)
    LeaseOracle(1, 5, 0, 10000, 20)

"""Batched LeaseManager kernels (summerset_amd/csrc/leaseman.hip, through the C-ABI): the reference's own unit tests
(src/server/leaseman.rs:1079-2301, restated in tests/lease_scenarios.py) run on the engine, and a seeded random stream of
notices over thousands of groups with every action and the full state compared with the oracle after each call --
bit-exact."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(__file__))
import lease_scenarios as LS  # noqa: E402

pytestmark = pytest.mark.gpu


def _mk_dev(G):
    return lambda R, me, expire: LS.DeviceAdapter(G, R, me, expire)


@pytest.mark.parametrize("trace", LS.ALL_TRACES, ids=lambda f: f.__name__)
def test_reference_trace_on_the_engine(cuda, trace):
    trace(_mk_dev(130))


def test_mutual_leases_on_the_engine(cuda):
    LS.mutual_leases(_mk_dev(3), order_seed=1)


@pytest.mark.parametrize("G,R,me,seed", [(4096, 5, 2, 0), (1000, 3, 0, 1), (300, 8, 7, 2)])
def test_random_notices_match_oracle(cuda, oracle, G, R, me, seed):
    granted, held = LS.random_differential(lambda *a: LS.DeviceAdapter(*a), lambda G, R, me, e: oracle.LeaseOracle(G, R, me, e, 20),
                                           G=G, R=R, me=me, steps=150, seed=seed)
    assert granted > 0 and held > 0            # the stream does reach promised leases on both sides


def test_create_rejects_what_new_and_setup_rejects(cuda):
    from summerset_amd import SummersetError
    from summerset_amd.leaseman import LeaseManager
    for expire, hb in [(99, 20), (10001, 20), (100, 50), (600, 300)]:
        with pytest.raises(SummersetError):
            LeaseManager(4, 5, 0, expire, hb)
    LeaseManager(4, 5, 0, 100, 49).close()


def test_sets_kernel(cuda):
    """grant_set / lease_set / lease_cnt for every group in one launch equal the dump's"""
    import numpy as np
    d = LS.DeviceAdapter(200, 5, 1, 600)
    ck = LS.Clock()
    n = LS.Node(d, ck)
    n.recv(3, 0, LS.GUARD)
    n.recv(3, 0, LS.PROMISE)
    n.recv(3, 4, LS.GUARD)
    n.new_grants(3, {2, 3})
    n.recv(3, 2, LS.GUARD_REPLY)
    g, l, c = (x.cpu().numpy() for x in d.m.sets())
    assert (g == 0b00100).all() and (l == 0b00001).all() and (c == 2).all()
    dd = d.dump()
    assert (dd["grant_set"] == g).all() and (dd["lease_set"] == l).all() and (dd["lease_cnt"] == c).all()
    assert (dd["guards_sent"] == 0b01000).all() and (dd["guards_held"] == 0b10000).all()

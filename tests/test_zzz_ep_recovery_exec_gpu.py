"""EPaxos explicit prepare TOGETHER with dependency-graph execution on the device (smr_ep_cfg.recovery = execute = 1):
the crash-and-recovery cluster run of tests/test_zz_ep_recovery_gpu.py with execution on, against the oracle cluster.
Sorted last: written when no device was at hand (verified on the emulator, tests/test_hostsim.py) -- a failure here
must not keep the rest of the suite from running under `pytest -x`."""
import numpy as np
import pytest

import ep_cluster as ec
import test_oracle_ep_recovery as tr
from test_zz_ep_recovery_gpu import _EngineAsOracle

pytestmark = pytest.mark.gpu          # first device run: GPUTEST_r02 (11 XPASS); quarantine removed in round 3


@pytest.mark.parametrize("G,seed,loss", [(700, 1, 0.0), (1500, 5, 0.15)])
def test_crash_and_recovery_with_execution_matches_the_oracle_cluster(cuda, oracle, G, seed, loss):
    """the same run with dependency-graph execution on: a HearTimeout runs several inner handlers per group in one call
    and each may move a commit bar, so the engine makes the reference's execution attempts inside that call, behind each
    (ep_heartbeat_timeout_kernel<true>).  Every output of every call, the instances, and the execution state -- exec
    bars, the KV store, the per-group digest over (command, old value) in submission order, the commands each replica
    submitted call by call -- equal the oracle cluster's."""
    W = 16
    te, to = [], []
    eng = _EngineAsOracle(cuda)
    re_, live, cut_e, tal_e = tr.run_crash_and_recovery(lambda G, R, r, W, K: eng.EpOracle(G, R, me=r, W=W, n_keys=K, execute=True), seed, loss, G, W, trace=te)
    ro, _, cut_o, tal_o = tr.run_crash_and_recovery(lambda G, R, r, W, K: oracle.EpOracle(G, R, me=r, W=W, n_keys=K, execute=True), seed, loss, G, W, trace=to)
    assert (cut_e == cut_o).all() and tal_e == tal_o and len(te) == len(to)
    for i, (a, b) in enumerate(zip(te, to)):
        assert a[0] == b[0]
        for x, y in zip(a[1:], b[1:]):
            assert np.array_equal(x, y), (i, a[0])
    executed = 0
    for q in range(5):
        a, b = re_[q].dump(), ro[q].dump()
        for n in b:
            assert np.array_equal(a[n], b[n]), (q, n)
        a, b = re_[q].exec_dump(), ro[q].exec_dump()
        for n in b:
            assert np.array_equal(a[n], b[n]), (q, "exec", n)
        executed += int(b["counters"][0])
        for x, y in zip(re_[q].take_submissions(), ro[q].take_submissions()):
            assert np.array_equal(x, y), (q, "submissions")
    assert executed > 0 and ec.check_agreement(re_, live, 0, G) > 0
